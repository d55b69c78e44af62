// Backward of the alpha compositing, Gaussian-per-lane ("skewed") form.  Replaces renderCUDA, cuda_rasterizer/backward.cu:399-557.
//
// The reference (and render_backward.hip) give every pixel a lane and must therefore SUM nine per-Gaussian partial gradients
// over the pixels: 9 global atomics per pair there, a 64-lane DPP reduce-scatter per (list entry, wave) here in
// render_backward.hip -- which is half of that kernel's instructions.  This kernel turns the loop around:
//
//   * the unit of work is an ITEM = (8x8 pixel block, list entry that blended at least one of its pixels); the forward blend
//     leaves the items behind as per-block lists (render_forward.hip, LISTS);
//   * one wave walks a stream of items, back to front within a block, block after block.  Lane j holds item 64*e + j of the
//     stream (e = "epoch") in registers and visits the 64 pixels of that item's block one per step, skewed by its lane
//     index: at step t it is on pixel (t - j) & 63.  So at any step the 64 lanes work on 64 different items and 64
//     different pixels, every pixel meets the items of its block in list order, and
//   * the per-pixel state of the back-to-front recurrence (T and the colour accumulated behind, backward.cu:502-515)
//     travels from lane j-1 to lane j between steps through DPP wave_ror:1 reads fused into the instructions that consume it; a
//     select injects the initial state where a lane starts the first item of a new block;
//   * the nine sums of an item never leave their lane: they are nine accumulator registers, stored as one 48-byte row
//     when the lane moves on to its next item (one lane per step does so).  No cross-lane reduction, no atomics, no LDS
//     combine, no barriers (workgroup = one wave), and the pipeline never drains between blocks.
//
// Per-pixel constants (pixel centre, last contributor, final T, dL/dpixel) sit in LDS, three blocks' worth per wave; the
// items of the next epoch are gathered (list entry -> 48-byte splat record -> row index) while the current epoch runs and
// staged in LDS, from where a lane picks up its next item with three 16-byte reads.
//
// Rows hold the raw moments S w, S w dx, S w dy, S w dx^2, S w dx dy, S w dy^2 (w = G dL/dalpha) and S alpha T dL/dpixel;
// the per-Gaussian factors of backward.cu:538-554 are applied by the per-Gaussian kernel after it has summed a Gaussian's rows.
// Row r of Gaussian g's rows: the (tile, quadrant) pairs marked in bmask, numbered by row_scan_kernel below in emission order.
//
// alpha / G are recomputed with the hardware exp2 and redone with the forward's gs_expf whenever a lane is within rounding of the
// alpha >= 1/255 threshold, so the set of blended pairs is exactly the forward's (as in render_backward.hip).
// VALU-issue bound; integer/fp32 vector work, no MFMA.
#include "common.h"
#include <stdio.h>
#include <algorithm>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Row numbering: rowinfo[slot] = (number of marked (slot', quadrant) pairs with slot' < slot) << 4 | quadrant mask of slot,
// rowinfo[R] = total << 4.  The rows of a Gaussian (its slots are contiguous) are therefore contiguous, in emission order.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int RS_PER_THREAD = 8, RS_PER_BLOCK = 256 * RS_PER_THREAD;
constexpr int QUEUE_SHARDS = 8;  // block queues of the blend kernel, one 128-byte line each

__device__ __forceinline__ uint32_t quad_mask(uint32_t w) { return (w & 1u) | ((w >> 7) & 2u) | ((w >> 14) & 4u) | ((w >> 21) & 8u); }

__global__ void __launch_bounds__(256) row_count_kernel(uint32_t R, const uint32_t* __restrict__ bmask4, uint32_t* __restrict__ sums)
{
    __shared__ uint32_t s_w[4];
    const uint32_t base = blockIdx.x * RS_PER_BLOCK + threadIdx.x * RS_PER_THREAD;
    uint32_t n = 0;
    if (base + RS_PER_THREAD <= R) {
        const uint4 a = *reinterpret_cast<const uint4*>(bmask4 + base), b = *reinterpret_cast<const uint4*>(bmask4 + base + 4);
        n = __popc(a.x & 0x01010101u) + __popc(a.y & 0x01010101u) + __popc(a.z & 0x01010101u) + __popc(a.w & 0x01010101u) +
            __popc(b.x & 0x01010101u) + __popc(b.y & 0x01010101u) + __popc(b.z & 0x01010101u) + __popc(b.w & 0x01010101u);
    } else {
        for (uint32_t i = base; i < R && i < base + RS_PER_THREAD; i++) n += __popc(bmask4[i] & 0x01010101u);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) n += __shfl_xor(n, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ void __launch_bounds__(256) row_write_kernel(uint32_t R, const uint32_t* __restrict__ bmask4, const uint32_t* __restrict__ offsets,
                                                        uint32_t* __restrict__ total, uint32_t* __restrict__ rowinfo,
                                                        uint32_t* __restrict__ queue_heads)
{
    __shared__ uint32_t s_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * RS_PER_BLOCK + threadIdx.x * RS_PER_THREAD;
    uint32_t w[RS_PER_THREAD], n = 0;
#pragma unroll
    for (int i = 0; i < RS_PER_THREAD; i++) {
        w[i] = base + i < R ? bmask4[base + i] : 0u;
        n += __popc(w[i] & 0x01010101u);
    }
    uint32_t incl = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t run = offsets[blockIdx.x] + incl - n;
    for (int k = 0; k < wave; k++) run += s_w[k];
#pragma unroll
    for (int i = 0; i < RS_PER_THREAD; i++) {
        if (base + i < R) rowinfo[base + i] = (run << 4) | quad_mask(w[i]);
        run += __popc(w[i] & 0x01010101u);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) rowinfo[R] = total[0] << 4;
    if (blockIdx.x == 0 && threadIdx.x < QUEUE_SHARDS) queue_heads[32 * threadIdx.x] = 0;  // the blend kernel's block queues
}

// ---------------------------------------------------------------------------------------------------------------------
// The blend backward
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PIX_SLOTS = 3;              // blocks whose pixel constants are resident per wave
constexpr uint32_t PIX_SLOT_BYTES = 64 * 32;
constexpr uint32_t NO_ROW = 0xFFFFFFFFu;

// State pass: lane j takes what lane j-1 (lane 0: lane 63) left behind in the previous step, except where `first` is set (the lane
// is on the first item of a block: initial state).  The wave_ror:1 read is fused into the first instruction that consumes the
// value (DPP on src0 of v_subrev_f32 / v_mul_f32) and the initial state is selected afterwards with an SGPR-mask v_cndmask:
// 8 instructions of 4.4 cycles.  (A v_cndmask_b32_dpp selects and rotates in one instruction, but its mask is VCC and that form
// issues in 23.5 cycles on gfx950 -- tools/valu_cost.hip.)
//   d_c = colour_c - accum_c   (accum_c = 0 on a first item)
//   T   = T_prev * inv         (T_prev = T_final of the pixel on a first item)
// accumulated colour itself (for the reference's form of the recurrence, accum' = alpha c + (1 - alpha) accum)
__device__ __forceinline__ void pass_accum(float& A0, float& A1, float& A2, float A0_o, float A1_o, float A2_o, unsigned long long first)
{
    asm volatile("s_nop 1\n"
                 "v_mov_b32_dpp %0, %3 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %1, %4 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %2, %5 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32 %0, %0, 0, %6\n"
                 "v_cndmask_b32 %1, %1, 0, %6\n"
                 "v_cndmask_b32 %2, %2, 0, %6\n"
                 : "=&v"(A0), "=&v"(A1), "=&v"(A2)
                 : "v"(A0_o), "v"(A1_o), "v"(A2_o), "s"(first));
}

__device__ __forceinline__ void pass_state(float& d0, float& d1, float& d2, float& T, float A0_o, float A1_o, float A2_o, float T_o,
                                           float cr, float cg, float cb, float inv, float T_final, unsigned long long first)
{
    float tf;
    asm volatile("s_nop 1\n"
                 "v_subrev_f32_dpp %0, %5, %9 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_subrev_f32_dpp %1, %6, %10 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_subrev_f32_dpp %2, %7, %11 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mul_f32_dpp %3, %8, %12 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mul_f32 %4, %13, %12\n"
                 "v_cndmask_b32 %0, %0, %9, %14\n"
                 "v_cndmask_b32 %1, %1, %10, %14\n"
                 "v_cndmask_b32 %2, %2, %11, %14\n"
                 "v_cndmask_b32 %3, %3, %4, %14\n"
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(T), "=&v"(tf)
                 : "v"(A0_o), "v"(A1_o), "v"(A2_o), "v"(T_o), "v"(cr), "v"(cg), "v"(cb), "v"(inv), "v"(T_final), "s"(first));
}

__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// One block of the stream: its list and where its pixels are
struct BlockRef {
    uint32_t tile, q, cnt;     // cnt = entries in the block's list
    uint32_t list_lo, list_hi; // 64-bit index of the list's first entry in blist
};

// What the lanes of one epoch hold: lanes [0, nA) continue block A (lane 0 on list position posA, descending),
// lanes [nA, nA + nB) start block B (lane nA on its LAST list entry = the first item of the block), the rest idle.
struct Plan {
    int nA, nB;
    uint32_t posA, posB;
    uint32_t pixA, pixB;       // byte offsets of the blocks' pixel slots in s_pix
    BlockRef A, B;
};

template <bool FAST_EXP, bool REF_ACCUM, int XV>
__global__ void __launch_bounds__(256) render_backward_skew_kernel(int W, int H, int gx, int gy, uint32_t nblocks4, const uint32_t* __restrict__ order,
                                                                 const uint32_t* __restrict__ ranges, const uint2* __restrict__ blist,
                                                                 const uint32_t* __restrict__ bcount, const uint32_t* __restrict__ rowinfo,
                                                                 const float* __restrict__ splat, const float* __restrict__ bg,
                                                                 const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
                                                                 const float* __restrict__ dL_dpixels, float* __restrict__ grad_rows,
                                                                 uint32_t* __restrict__ queue_heads, unsigned long long* __restrict__ dbg)
{
    const unsigned long long dbg_t0 = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    uint32_t dbg_epochs = 0, dbg_items = 0;
    // A workgroup is four INDEPENDENT waves (no barrier, nothing shared): its only purpose is to put one wave on each SIMD of a CU
    // -- single-wave workgroups were placed unevenly and the waves of crowded SIMDs ran up to 1.7x longer than the rest.
    __shared__ float4 s_pix_all[4][PIX_SLOTS * 128];  // per wave and slot: 64 x {px, py, last contributor (uint bits), T_final | dL0, dL1, dL2, bg . dL}
    __shared__ float4 s_sa_all[4][64], s_sb_all[4][64], s_sc_all[4][64];  // staged items: {mx, my, cx, cy | cz, opacity, r, g | b, row, list position, -}
    const int lane = threadIdx.x & 63;
    const int wv = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: everything derived from it stays in SGPRs
    const uint32_t wave_id = blockIdx.x * 4u + (uint32_t)wv;
    float4* const s_pix = s_pix_all[wv];
    float4* const s_sa = s_sa_all[wv];
    float4* const s_sb = s_sb_all[wv];
    float4* const s_sc = s_sc_all[wv];
    const size_t plane = (size_t)H * W;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

    // ---- the wave's stream of blocks: (tiles in launch order, longest lists first) x (4 quadrants), handed out by QUEUE_SHARDS
    // atomic counters -- shard s owns the sequence positions == s (mod QUEUE_SHARDS); a wave starts on shard wave_id % 8 and moves
    // to the next one when its shard is exhausted.  Waves run at different speeds (the SIMD's arbiter favours the older wave), so a
    // fixed split of equal work finished 40 % after the average wave; one counter for everybody would saturate (a single word
    // serves ~88 returning atomics per us, 33k blocks at 1080p).  The next ticket is always in flight: a wave never waits for one
    // unless it runs into several empty blocks in a row.
    uint32_t shard = wave_id % QUEUE_SHARDS, shards_left = QUEUE_SHARDS;
    uint32_t ticket = 0;  // VGPR (lane 0): result of the pending atomic
    auto request = [&]() { if (lane == 0) ticket = atomicAdd(queue_heads + 32 * shard, 1u); };
    request();
    auto fetch_block = [&](BlockRef& b) -> bool {
        while (shards_left > 0) {
            const uint32_t i = sgpr(ticket) * QUEUE_SHARDS + shard;
            if (i >= nblocks4) {  // this shard is exhausted: go on with the next one
                shards_left--;
                shard = shard + 1 == QUEUE_SHARDS ? 0 : shard + 1;
                if (shards_left > 0) request();
                continue;
            }
            request();
            const uint32_t tile = order ? sgpr(order[i >> 2]) : (i >> 2);
            const uint32_t q = i & 3u;
            const uint32_t cnt = sgpr(bcount[4 * tile + q]);
            if (cnt == 0) continue;
            const uint32_t r0 = sgpr(ranges[2 * tile]), r1 = sgpr(ranges[2 * tile + 1]);
            const unsigned long long lo = 4ull * r0 + (unsigned long long)q * (r1 - r0);
            b.tile = tile; b.q = q; b.cnt = cnt; b.list_lo = (uint32_t)lo; b.list_hi = (uint32_t)(lo >> 32);
            return true;
        }
        return false;
    };

    BlockRef cur;            // block being issued
    uint32_t cur_rem = 0;    // its entries not yet handed to a lane (the next one is list position cur_rem - 1)
    uint32_t cur_pix = 0;    // its pixel slot (byte offset)
    uint32_t slot_rr = 0;    // round robin over the pixel slots
    cur.tile = 0; cur.q = 0; cur.cnt = 0; cur.list_lo = 0; cur.list_hi = 0;

    // plans an epoch: at most the tail of the current block and ONE new block (so a block's pixel slot is reused only after
    // two later blocks have started, i.e. at least two epochs after its last item was handed out: that item is long done)
    auto plan_epoch = [&](Plan& p) {
        p.nA = (int)min(cur_rem, 64u);
        p.posA = cur_rem - 1u;
        p.pixA = cur_pix;
        p.A = cur;
        cur_rem -= (uint32_t)p.nA;
        p.nB = 0; p.posB = 0; p.pixB = 0; p.B = cur;
        if (cur_rem == 0 && p.nA < 64) {
            BlockRef nb;
            if (fetch_block(nb)) {
                slot_rr = slot_rr + 1 == PIX_SLOTS ? 0 : slot_rr + 1;
                cur = nb;
                cur_pix = slot_rr * PIX_SLOT_BYTES;
                p.nB = (int)min(nb.cnt, (uint32_t)(64 - p.nA));
                p.posB = nb.cnt - 1u;
                p.pixB = cur_pix;
                p.B = nb;
                cur_rem = nb.cnt - (uint32_t)p.nB;
            }
        }
    };

    // ---- gathering an epoch's items (one per lane) ----
    // Three dependent loads (list entry -> 48-byte splat record -> row numbering) plus the new block's pixel constants.  They are
    // issued one stage per 8-step group (gather_stage 0..3) and each result is first touched a group later: a load has ~2000
    // cycles to land, and the registers it lands in stay live in between, so the steps never wait on them.
    // the lane's item of the NEXT epoch, until its stage slot is free: {mx, my, cx, cy | cz, opacity, r, g | b, row, list position, -}
    struct Held { float4 a, b, c; } held;
    const float4 pixh0 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint2 g_it = make_uint2(0u, 0u);                      // in flight: list entry {Gaussian id, list position}
    float4 g_q2 = pixh0;                                  // in flight: third quarter of the splat record (the first two land in held.a / held.b)
    uint32_t g_ri = 0, g_q = 0;                           // in flight: row numbering word; the item's quadrant
    bool g_valid = false;
    float g_Tf = 0.f, g_d0 = 0.f, g_d1 = 0.f, g_d2 = 0.f; // in flight: the new block's pixel constants
    uint32_t g_lc = 0, g_px = 0, g_py = 0;
    auto gather_stage = [&](const Plan& p, int stage) {
        if (stage == 0) {
            const bool isA = lane < p.nA, isB = !isA && lane < p.nA + p.nB;
            g_valid = isA || isB;
            g_q = isA ? p.A.q : p.B.q;
            if (g_valid) {
                const uint32_t pos = isA ? p.posA - (uint32_t)lane : p.posB - (uint32_t)(lane - p.nA);
                const unsigned long long li = (((unsigned long long)(isA ? p.A.list_hi : p.B.list_hi)) << 32 | (isA ? p.A.list_lo : p.B.list_lo)) + pos;
                g_it = blist[li];
            }
            if (p.nB != 0) {  // pixel constants of the block that starts in this epoch (lane = pixel)
                const uint32_t tile = p.B.tile, q = p.B.q;
                const uint32_t bx = tile % (uint32_t)gx, by = tile / (uint32_t)gx;
                g_px = bx * GSR_TILE_X + (q & 1u) * 8u + ((uint32_t)lane & 7u);
                g_py = by * GSR_TILE_Y + (q >> 1) * 8u + ((uint32_t)lane >> 3);
                g_Tf = 0.f; g_d0 = 0.f; g_d1 = 0.f; g_d2 = 0.f; g_lc = 0;
                if (g_px < (uint32_t)W && g_py < (uint32_t)H) {
                    const size_t pid = (size_t)W * g_py + g_px;
                    g_Tf = final_Ts[pid]; g_lc = n_contrib[pid];
                    g_d0 = dL_dpixels[pid]; g_d1 = dL_dpixels[plane + pid]; g_d2 = dL_dpixels[2 * plane + pid];
                }
            }
        } else if (stage == 1) {
            if (g_valid) {
                const float4* rec = reinterpret_cast<const float4*>(splat + (size_t)g_it.x * GSR_SPLAT_F);
                held.a = rec[0]; held.b = rec[1]; g_q2 = rec[2];
            }
        } else if (stage == 2) {
            if (g_valid) {
                const bool isA = lane < p.nA;
                const uint32_t tile = isA ? p.A.tile : p.B.tile;
                const uint32_t bx = tile % (uint32_t)gx, by = tile / (uint32_t)gx;
                uint32_t mnx, mny, mxx, mxy;
                gsr_get_rect(held.a.x, held.a.y, __float_as_int(g_q2.z), gx, gy, mnx, mny, mxx, mxy);
                g_ri = rowinfo[__float_as_uint(g_q2.w) + (by - mny) * (mxx - mnx) + (bx - mnx)];
            }
        } else {
            if (g_valid) {
                const uint32_t row = (g_ri >> 4) + (uint32_t)__popc(g_ri & ((1u << g_q) - 1u));
                held.c = make_float4(g_q2.x, __uint_as_float(row), __uint_as_float(g_it.y), 0.f);
            } else {  // idle lane: opacity 0 and behind every last contributor -- never blends; no row
                held.a = make_float4(0.f, 0.f, 1.f, 0.f);
                held.b = make_float4(1.f, 0.f, 0.f, 0.f);
                held.c = make_float4(0.f, __uint_as_float(NO_ROW), __uint_as_float(0xFFFFFFFFu), 0.f);
            }
        }
    };
    auto commit_items = [&](bool low_half) {
        if ((lane < 32) == low_half) { s_sa[lane] = held.a; s_sb[lane] = held.b; s_sc[lane] = held.c; }
    };
    // finished rows wait in the lanes' row buffers; the 32 lanes of a half store theirs together once all of them have switched
    float rb0 = 0.f, rb1 = 0.f, rb2 = 0.f, rb3 = 0.f, rb4 = 0.f, rb5 = 0.f, rb6 = 0.f, rb7 = 0.f, rb8 = 0.f;
    uint32_t rb_row = NO_ROW;
    auto flush_rows = [&](bool low_half) {
        if ((lane < 32) == low_half && rb_row != NO_ROW) {
            float4* o = reinterpret_cast<float4*>(grad_rows + (size_t)rb_row * GSR_GROW_F);
            o[0] = make_float4(rb1, rb2, rb3, rb4);
            o[1] = make_float4(rb5, rb0, rb6, rb7);
            reinterpret_cast<float*>(o)[8] = rb8;  // the row's last three floats are never read
        }
    };
    // a group later (the stores have read their data by then, so nothing waits on them): the buffers go back to zero
    auto flush_done = [&](bool low_half) {
        if ((lane < 32) == low_half) { rb0 = 0.f; rb1 = 0.f; rb2 = 0.f; rb3 = 0.f; rb4 = 0.f; rb5 = 0.f; rb6 = 0.f; rb7 = 0.f; rb8 = 0.f; rb_row = NO_ROW; }
    };
    auto commit_pixels = [&](const Plan& p) {
        if (p.nB == 0) return;
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<char*>(s_pix) + p.pixB);
        const float bg_dot = ((0.f + bg0 * g_d0) + bg1 * g_d1) + bg2 * g_d2;
        dst[2 * lane] = make_float4((float)g_px, (float)g_py, __uint_as_float(g_lc), g_Tf);
        dst[2 * lane + 1] = make_float4(g_d0, g_d1, g_d2, bg_dot);
    };

    // ---- the lanes' running state ----
    float mx = 0.f, my = 0.f, cx = 1.f, cy = 0.f, cz = 1.f, op = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
    uint32_t kidx = 0xFFFFFFFFu, row = NO_ROW, pa = 0;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f, acc4 = 0.f, acc5 = 0.f, acc6 = 0.f, acc7 = 0.f, acc8 = 0.f;
    float T_o = 0.f, A0_o = 0.f, A1_o = 0.f, A2_o = 0.f;
    unsigned long long first_mask = 0ull;

    // ---- prologue: epoch 0 ----
    Plan planc, plann;   // current epoch's plan, next epoch's plan
    plan_epoch(planc);
    if (planc.nA + planc.nB == 0) return;  // nothing for this wave
    for (int st = 0; st < 4; st++) gather_stage(planc, st);
    commit_items(true);
    commit_items(false);
    commit_pixels(planc);
    __builtin_amdgcn_wave_barrier();

    bool more = true;  // the epoch being run holds at least one real item (then one more epoch flushes it)
    int last_items = 0;
    while (true) {
        const bool real = planc.nA + planc.nB > 0;
        if (!real && !more) break;
        more = real;
        dbg_epochs++; dbg_items += (uint32_t)(planc.nA + planc.nB);
        // an epoch: 64 steps in 8 groups; the next epoch is prepared between the groups.  The last epoch only flushes: lane j of
        // the epoch before it is done (and stores its row) at step j, so it runs as many steps as that epoch had items.
        const int groups = real ? 8 : (last_items + 7) / 8;
        last_items = planc.nA + planc.nB;
        for (int c = 0; c < groups; c++) {
            if (c == 0) {
                flush_done(false);    // (the rows of lanes 32..63 were stored at the end of the previous epoch)
                plan_epoch(plann);
                gather_stage(plann, 0);
            } else if (c < 4) {
                gather_stage(plann, c);
            } else if (c == 4) {
                flush_rows(true);     // lanes 0..31 have all switched (steps 0..31): their stage slots are free, their rows complete
                commit_items(true);
            } else if (c == 5) {
                flush_done(true);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {  // unrolled: a taken branch costs a wave ~40 cycles, the step body has none on its common path
                const int k = 8 * c + i;
                // lane k moves on to its item of this epoch: store the finished item's row, pick up the new one
                if (!(XV & 2) && lane == k) {
                    const float4 a = s_sa[k], b = s_sb[k], cc = s_sc[k];
                    // the finished sums move to the lane's row buffer, which holds zeros (flush_done): one v_swap_b32 per value both
                    // saves the row and clears the accumulator.  The row leaves for HBM with those of 31 other lanes (flush_rows).
                    // (A store straight from the accumulators must be waited for before they can be cleared -- one memory round trip
                    // per step; parking the row in LDS costs the LDS store path as much as a full-wave write, every step.)
                    if (!(XV & 1)) {
                        asm volatile("v_swap_b32 %0, %9\n v_swap_b32 %1, %10\n v_swap_b32 %2, %11\n v_swap_b32 %3, %12\n v_swap_b32 %4, %13\n"
                                     "v_swap_b32 %5, %14\n v_swap_b32 %6, %15\n v_swap_b32 %7, %16\n v_swap_b32 %8, %17\n"
                                     : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7), "+v"(acc8),
                                       "+v"(rb0), "+v"(rb1), "+v"(rb2), "+v"(rb3), "+v"(rb4), "+v"(rb5), "+v"(rb6), "+v"(rb7), "+v"(rb8));
                        rb_row = row;
                    } else { acc0 = 0.f; acc1 = 0.f; acc2 = 0.f; acc3 = 0.f; acc4 = 0.f; acc5 = 0.f; acc6 = 0.f; acc7 = 0.f; acc8 = 0.f; }
                    mx = a.x; my = a.y; cx = a.z; cy = a.w;
                    cz = b.x; op = b.y; cr = b.z; cg = b.w;
                    cb = cc.x; row = __float_as_uint(cc.y); kidx = __float_as_uint(cc.z);
                    pa = k < planc.nA ? planc.pixA : planc.pixB;
                }
                {
                    const unsigned long long bit = 1ull << k;
                    const unsigned long long set = (planc.nB > 0 && k == planc.nA) ? bit : 0ull;  // s_cselect, no branch
                    first_mask = (first_mask & ~bit) | set;
                }
                // this step's pixel
                float4 p0, p1;
                if (XV & 8) { p0 = make_float4(mx + 1.f, my - 2.f, __uint_as_float(100000u), 0.5f); p1 = make_float4(0.1f, 0.2f, 0.3f, 0.f); }
                else {
                    p0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_pix) + pa);
                    p1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_pix) + pa + 16);
                }
                pa += 32;
                const float dx = mx - p0.x, dy = my - p0.y;
                // identical to the forward (CONTRACT-FMA)
                const float q = __builtin_fmaf(cx * dx, dx, (cz * dy) * dy);
                const float power = __builtin_fmaf(-0.5f, q, -((cy * dx) * dy));
                float G = FAST_EXP ? __builtin_amdgcn_exp2f(power * 1.44269504088896341f) : gsr_expf(fminf(power, 0.0f));
                float alpha = fminf(0.99f, op * G);
                // entries at or behind the pixel's last contributor were never blended into it (backward.cu:480-482)
                const unsigned long long m_pre = __builtin_amdgcn_ballot_w64(kidx < __float_as_uint(p0.z)) & __builtin_amdgcn_ballot_w64(power <= 0.0f);
                if (FAST_EXP && !(XV & 4) && __builtin_expect((m_pre & __builtin_amdgcn_ballot_w64(fabsf(alpha - 1.0f / 255.0f) < 3e-8f)) != 0ull, 0)) {
                    G = gsr_expf(fminf(power, 0.0f));  // the forward's exp for the decision (see render_backward.hip)
                    alpha = fminf(0.99f, op * G);
                }
                const unsigned long long m_c = m_pre & __builtin_amdgcn_ballot_w64(alpha >= 1.0f / 255.0f);
                // a lane that does not blend gets alpha = G = 0: T / (1 - 0) = T, its sums get zeros, the colour recurrence passes through
                float Ge, ae;
                asm("v_cndmask_b32 %0, 0, %2, %4\n v_cndmask_b32 %1, 0, %3, %4" : "=&v"(Ge), "=v"(ae) : "v"(G), "v"(alpha), "s"(m_c));
                const float oma = 1.f - ae;
                const float inv = __builtin_amdgcn_rcpf(oma);
                float d0, d1, d2, T;  // colour - accumulated colour behind (backward.cu:515-520), T / (1 - alpha) (backward.cu:502)
                pass_state(d0, d1, d2, T, A0_o, A1_o, A2_o, T_o, cr, cg, cb, inv, p0.w, first_mask);
                float dL_dalpha = d0 * p1.x;
                dL_dalpha = __builtin_fmaf(d1, p1.y, dL_dalpha);
                dL_dalpha = __builtin_fmaf(d2, p1.z, dL_dalpha);
                dL_dalpha *= T;
                dL_dalpha = __builtin_fmaf(-p0.w * inv, p1.w, dL_dalpha);  // background term, backward.cu:531-534
                const float w = Ge * dL_dalpha;
                const float aT = ae * T;
                const float wdx = w * dx, wdy = w * dy;
                acc0 += w; acc1 += wdx; acc2 += wdy;
                acc3 = __builtin_fmaf(wdx, dx, acc3); acc4 = __builtin_fmaf(wdx, dy, acc4); acc5 = __builtin_fmaf(wdy, dy, acc5);
                acc6 = __builtin_fmaf(aT, p1.x, acc6); acc7 = __builtin_fmaf(aT, p1.y, acc7); acc8 = __builtin_fmaf(aT, p1.z, acc8);
                // state handed to the next item of this pixel, the colour recurrence of backward.cu:515 folded in:
                // accum' = alpha c + (1 - alpha) accum = c - (1 - alpha)(c - accum)
                T_o = T;
                if (REF_ACCUM) {
                    float A0, A1, A2;
                    pass_accum(A0, A1, A2, A0_o, A1_o, A2_o, first_mask);
                    A0_o = __builtin_fmaf(ae, cr, oma * A0);
                    A1_o = __builtin_fmaf(ae, cg, oma * A1);
                    A2_o = __builtin_fmaf(ae, cb, oma * A2);
                } else {
                    A0_o = __builtin_fmaf(-oma, d0, cr);
                    A1_o = __builtin_fmaf(-oma, d1, cg);
                    A2_o = __builtin_fmaf(-oma, d2, cb);
                }
            }
        }
        // epoch boundary: stage slots 32..63 and the new block's pixel slot are free now
        if (groups > 4) flush_rows(false);
        if (!real) {  // that was the flushing epoch: whatever it parked in the low half (all of it if it was short) leaves now
            if (groups <= 4) { flush_rows(true); flush_rows(false); }
            break;
        }
        commit_items(false);
        commit_pixels(plann);
        planc = plann;
        __builtin_amdgcn_wave_barrier();
    }
    if (dbg && lane == 0) {
        dbg[4 * wave_id] = dbg_t0; dbg[4 * wave_id + 1] = __builtin_amdgcn_s_memrealtime();
        dbg[4 * wave_id + 2] = dbg_epochs; dbg[4 * wave_id + 3] = dbg_items;
    }
}

}  // namespace

size_t gsr_row_scan_blocks(int64_t R) { return (size_t)(((R > 0 ? R : 1) + RS_PER_BLOCK - 1) / RS_PER_BLOCK); }

// rowinfo [R + 1]; sums / offsets [gsr_row_scan_blocks(R)] each; total [1]; queue_heads: QUEUE_SHARDS counters, 128 bytes apart
size_t gsr_skew_queue_bytes() { return (size_t)QUEUE_SHARDS * 128; }
void gsr_launch_row_scan(uint32_t R, const uint8_t* bmask, uint32_t* sums, uint32_t* offsets, uint32_t* total, uint32_t* rowinfo,
                         uint32_t* queue_heads, hipStream_t s)
{
    const int nb = (int)gsr_row_scan_blocks(R);
    hipLaunchKernelGGL(row_count_kernel, dim3(nb), dim3(256), 0, s, R, reinterpret_cast<const uint32_t*>(bmask), sums);
    gsr_launch_scan_blocks(nb, sums, offsets, total, nullptr, 0, s);
    hipLaunchKernelGGL(row_write_kernel, dim3(nb), dim3(256), 0, s, R, reinterpret_cast<const uint32_t*>(bmask), offsets, total, rowinfo, queue_heads);
}

void gsr_launch_render_backward_skew(int W, int H, const uint32_t* order, const uint32_t* ranges, const uint2* blist, const uint32_t* bcount,
                                     const uint32_t* rowinfo, const float* splat, const float* bg, const float* final_T,
                                     const uint32_t* n_contrib, const float* dL_dpix, float* grad_rows, uint32_t* queue_heads, hipStream_t s)
{
    const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    const uint32_t nblocks4 = 4u * (uint32_t)(gx * gy);
    // workgroups of four independent waves (one per SIMD); a fixed number of them pulls blocks from the queues
    int wgs = gsr_variant("SKEW_WAVES", 16) * 256 / 4;
    if ((uint32_t)wgs * 4u > nblocks4) wgs = (int)((nblocks4 + 3) / 4);
    const int waves = wgs * 4;
    if (!gsr_variant("LPT", 1)) order = nullptr;
    const bool fast = gsr_variant("BWD_FASTEXP", 1), ref = gsr_variant("SKEW_REFACC", 1);
    unsigned long long* dbg = nullptr;
    if (gsr_variant("SKEW_DEBUG", 0)) hipMalloc(&dbg, (size_t)waves * 32);  // diagnostic build only: per-wave start / end / epochs / items
#define GSR_SKEW_LAUNCH(F, A, X)                                                                                                            \
    hipLaunchKernelGGL((render_backward_skew_kernel<F, A, X>), dim3(wgs), dim3(256), 0, s, W, H, gx, gy, nblocks4, order, ranges, blist, bcount, \
                       rowinfo, splat, bg, final_T, n_contrib, dL_dpix, grad_rows, queue_heads, dbg)
    const int xv = gsr_variant("SKEW_XV", 0);  // timing experiments (results are wrong for xv != 0)
    if (xv == 1) GSR_SKEW_LAUNCH(true, false, 1);
    else if (xv == 2) GSR_SKEW_LAUNCH(true, false, 2);
    else if (xv == 4) GSR_SKEW_LAUNCH(true, false, 4);
    else if (xv == 8) GSR_SKEW_LAUNCH(true, false, 8);
    else if (xv == 15) GSR_SKEW_LAUNCH(true, false, 15);
    else if (fast && ref) GSR_SKEW_LAUNCH(true, true, 0);
    else if (fast) GSR_SKEW_LAUNCH(true, false, 0);
    else if (ref) GSR_SKEW_LAUNCH(false, true, 0);
    else GSR_SKEW_LAUNCH(false, false, 0);
#undef GSR_SKEW_LAUNCH
    if (dbg) {
        std::vector<unsigned long long> h((size_t)waves * 4);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        hipFree(dbg);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int w = 0; w < waves; w++) { t0 = std::min(t0, h[4 * w]); t1 = std::max(t1, h[4 * w + 1]); }
        std::vector<double> dur, st, it;
        for (int w = 0; w < waves; w++) { dur.push_back((h[4 * w + 1] - h[4 * w]) * 0.01); st.push_back((h[4 * w] - t0) * 0.01); it.push_back((double)h[4 * w + 3]); }
        auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
        double sum_items = 0, sum_ep = 0;
        for (int w = 0; w < waves; w++) { sum_items += h[4 * w + 3]; sum_ep += h[4 * w + 2]; }
        fprintf(stderr, "[skew] waves %d span %.1f us | wave duration us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f | start us: p50 %.1f p90 %.1f max %.1f | items/wave: min %.0f p50 %.0f max %.0f | items %.0f epochs %.0f (%.1f items/epoch)\n",
                waves, (t1 - t0) * 0.01, pct(dur, 0), pct(dur, 0.1), pct(dur, 0.5), pct(dur, 0.9), pct(dur, 1), pct(st, 0.5), pct(st, 0.9), pct(st, 1),
                pct(it, 0), pct(it, 0.5), pct(it, 1), sum_items, sum_ep, sum_items / sum_ep);
    }
}
