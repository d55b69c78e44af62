// Backward of the alpha compositing (replaces renderCUDA, cuda_rasterizer/backward.cu:399-557).
//
// The reference issues 9 global float atomicAdds per contributing (pixel, Gaussian) pair (backward.cu:523,545-554).  On MI355X global
// float atomics top out near 1.3 TB/s chip-wide and collapse when lanes scatter, so this kernel never uses them: the 9 partial sums
// of a list entry are reduced across the 64 lanes of a wave with DPP adds (fixed tree), the waves of a tile are combined in LDS (float
// atomics with exactly two addends per row: order-independent),
// and every reachable (Gaussian, tile) instance writes ONE 36-byte gradient row (nine floats; 48-byte rows with three padding floats until round 4) with plain stores.  Rows are numbered per Gaussian
// (first row + rank of the tile among the tiles its splat can reach, gsr_reach_mask), so the rows of one Gaussian are contiguous and
// the per-Gaussian kernel sums them in a fixed order: gradients are bitwise reproducible run to run, unlike the reference's.
//
// What is reduced are MOMENTS of w = G*dL/dalpha (S w, S w*dx, S w*dy, S w*dx^2, S w*dx*dy, S w*dy^2) and S alpha*T*dL/dpixel; the
// per-Gaussian factors of backward.cu:538-554 (opacity, conic, 0.5*W, 0.5*H) are applied once per list entry after the reduction.
//
// Work split.  A tile is 16x16 pixels = four 8x8 quadrants.  A wave owns NQ quadrants (NQ pixels per lane, state in registers):
//   NQ = 1: four waves per tile (256 threads).  One wave reduction per (entry, quadrant) that blends: 2.90M on the benchmark scene.
//   NQ = 2: two waves per tile (128 threads), a 16x8 half each.  A lane's two pixels lie in the same image row, so they share dy: each
//           blending quadrant leaves (w, w*dx, dx, alpha*T), the two are combined as S w, S w*dx, S w*dx^2, S alpha*T*dL/dpixel, and the y
//           moments are S w*dy = (S w)*dy, S w*dx*dy = (S w*dx)*dy, S w*dy^2 = (S w*dy)*dy -- 14 instructions for a pair, 8 for a single
//           quadrant -- and reduced once per (entry, half): 1.87M reductions (18 DPP adds + 3 selects + 2 permlane swaps).  A wave walks twice
//           the pixels, so a tile's critical path is twice as long: faster where lists are short and tiles many (500k Gaussians at
//           1080p: 0.325 vs 0.348 ms), slower where every tile's list is long (2M: 0.745 vs 0.636 ms) or the image has few tiles
//           (800x800: 0.339 vs 0.192).  The launcher picks by the mean list length and the tile count.  (NQ = 4, one wave
//           per tile and 1.22M reductions, was built and measured in round 3: fewest instructions, but the longest tiles alone set the
//           kernel's duration.)
// The quadrants of an entry are visited under SCALAR branches: which staged entries can reach which quadrant (gsr_quad_mask) is a
// ballot, i.e. a wave-uniform 64-bit mask in SGPRs, and a wave picks its entries with scalar bit scans.
//
// Traversal is back to front from the tile's largest n_contrib (entries behind it contribute to no pixel, backward.cu:480-482);
// alpha / G are recomputed so that the set of blended pairs is exactly the forward's (see FAST_EXP below).
//
// TWO kernels live in this file.  render_backward_kernel (rounds 1-4) re-derives every blend decision from the geometry, as the reference
// does.  render_backward_log_kernel (round 5, the default) reads them from the BLEND LOG the forward leaves (api.hip,
// gsr_blend_log_for; render_forward.hip): one 64-bit lane mask per (list entry, quadrant) visit -- see its own header below.  Staging, the
// wave reduction, the LDS sums, the gradient rows and the work split are the same in both.
#include "common.h"

namespace {

// Sums 9 registers over the 64 lanes of a wave by reduce-scatter: at every level a lane adds its partner's copy of the values it keeps
// and drops the others.  The in-row levels are ordered by what "keep by bit b of the lane" costs in DPP:
//   lane^4, lane^8 first (eight, then four values): a DPP bank is four CONSECUTIVE lanes, so a bank mask enables lanes by bit 2 or by
//                  bit 3 and a pair of values is two masked DPP adds -- no select;
//   lane^1         (two values) bits 0 and 1 cannot be masked: two selects + one DPP add;
//   lane^2         one value left: plain butterfly.
// The 9th value is reduced along its row on its own (4 DPP adds slotted between dependent steps) and parked in lane 2, and the two
// cross-row levels (v_permlane16_swap / v_permlane32_swap, two VALU instructions + a copy each) are shared by all nine.
// 18 DPP adds + 3 selects; round 2's order (lane^1, ^2, ^4, ^8) needed 13 + 15.  Written as ONE asm block: every DPP read is far enough
// behind the write it depends on (hipcc pads nothing in asm), and the compiler cannot re-pack the adds into v_pk_add_f32 + v_mov_dpp
// pairs (what it emits for the builtin form: 2.5x the instructions).
// Result (lanes 0..15): lane l with bit 1 clear holds the wave total of value (l&1) + 2*((l>>3)&1) + 4*((l>>2)&1), lane 2 that of v8.
// m0/m2: lane masks (l&1), ((l&15)==2).  All 64 lanes must be active.
// Measured alternatives (round 3, cfg2 two waves per tile / 2M Gaussians four waves per tile, ms): cross-row levels FIRST with pair-wise
// permlane swaps (8 swaps + 7 DPP, no select at all): 0.349 / 0.668 against 0.338 / 0.637 -- a swap costs far more than its 8.5 cycles
// back to back suggest; cross-row levels through ds_bpermute_b32: 0.347 / 0.676 (the LDS round trip lengthens every entry's chain).
// wave_sum9_rows: the in-row levels only -- lane l of EVERY row then holds its row's partial of the value the lane keeps.
__device__ __forceinline__ float wave_sum9_rows(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                                float v8, unsigned long long m0, unsigned long long m2)
{
    float t0;
    asm volatile(
        "s_nop 1\n"  // the first DPP reads may follow the instruction that produced their operand (the compiler sees no DPP in here)
        // lane^4: lanes with bit 2 clear keep v0..v3, the others v4..v7
        "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n"
        "v_add_f32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %1, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %2, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %3, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"  // 9th value
        // lane^8: lanes 0..7 of a row keep registers 0, 1, lanes 8..15 registers 2, 3
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n"
        "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %1, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n"
        "v_add_f32_dpp %8, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n"
        // lane^1: lanes with bit 0 keep register 1
        "v_cndmask_b32 %9, %1, %0, %10\n"   // what the partner needs from me
        "v_cndmask_b32 %0, %0, %1, %10\n"   // what I keep
        "v_add_f32_dpp %8, %8, %8 row_mirror row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %0, %9, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        // lane^2
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        // park the 9th value's row total in lane 2 of each row
        "v_cndmask_b32 %0, %0, %8, %11\n"
        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "=&v"(t0)
        : "s"(m0), "s"(m2));
    return v0;
}
// XROW: the two cross-row levels through the LDS crossbar (ds_bpermute_b32 of lane + 16, lane + 32: no VALU slot but the add, the same sum
// tree, bit-identical) instead of v_permlane16_swap / v_permlane32_swap (3.6 ns of VALU issue each + add + copy).  Measured with the log
// walk (round 5, two alternations / one): two waves per tile (cfg2) 0.2829 / 0.2836 -> 0.2759 / 0.2761 ms; four waves per tile (2 M
// Gaussians) 0.5858 -> 0.6063 (twice the waves on the LDS pipe) -- so the log walk uses it with two quadrants per wave only
// (GSR_BWD_XROW: 0 never, 1 that rule, 2 always).  Round 3 had measured it slower in the kernel of the time (header above).
#ifndef GSR_BWD_XROW
#define GSR_BWD_XROW 1
#endif
#ifndef GSR_BWD_EARLYMASK
#define GSR_BWD_EARLYMASK 0  // 1: the log walk requests the next entry's masks right behind the first use of this entry's operands (pinned by scheduling
                             // barriers) instead of with the LDS operands before the reduction -- meant to keep the scalar load out of the cross-row
                             // exchange's lgkmcnt(0) wait; measured 0.2806 / 0.2819 against 0.2765 / 0.2779 ms: slower, off
#endif
#ifndef GSR_BWD_SMULADDR
#define GSR_BWD_SMULADDR 1   // the entry's offset into the LDS sums by s_mul_i32 (0: the compiler's choice, a v_mad_u64_u32 with a register pair on either
                             // side); measured 0.2742 / 0.2748 against 0.2765 / 0.2779 ms
#endif
template <bool XROW = false>
__device__ __forceinline__ float wave_sum9_scatter(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                                   float v8, unsigned long long m0, unsigned long long m2)
{
    v0 = wave_sum9_rows(v0, v1, v2, v3, v4, v5, v6, v7, v8, m0, m2);
    if (XROW) {
        const int lane4 = (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) << 2);
        // (lane + 16, lane + 32 mod 64 rather than lane ^ 16, lane ^ 32: the constant goes into the instruction's offset field, one address
        // register for both; the lanes of row 0 -- the only ones whose totals are used -- form the same tree (r0 + r1) + (r2 + r3))
        v0 += __int_as_float(__builtin_amdgcn_ds_bpermute(lane4 + 64, __float_as_int(v0)));
        v0 += __int_as_float(__builtin_amdgcn_ds_bpermute(lane4 + 128, __float_as_int(v0)));
        return v0;
    }
    // levels 5, 6: lane-wise across the four rows with gfx950's v_permlane16_swap / v_permlane32_swap
    {
        const uint32_t u = __float_as_uint(v0);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // (row0,row0,row2,row2), (row1,row1,row3,row3)
        v0 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const uint32_t u = __float_as_uint(v0);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // (lo,lo), (hi,hi)
        v0 = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v0;
}
// the lanes that hold a total after wave_sum9_scatter, and which
__device__ __forceinline__ bool gsr_sum9_writer(int lane) { return lane < 16 && ((lane & 2) == 0 || lane == 2); }
__device__ __forceinline__ int gsr_sum9_slot(int lane) { return lane == 2 ? 8 : (lane & 1) + 2 * ((lane >> 3) & 1) + 4 * ((lane >> 2) & 1); }

// The two cross-row levels on the MATRIX core (round 5): v_mfma_f32_16x16x4_f32 with A = the row partials (lane l supplies A[l % 16][l / 16])
// and B = ones gives D[i][j] = sum over the four rows of position i, for every j -- the sum the two permlane swaps + two adds + two
// copies (~27 VALU cycles) formed, for one VALU issue slot; the 8 passes run beside the VALU.  Lane (g = l / 16, any j) then holds positions
// 4g .. 4g+3 in its four result registers: value 0, 1, 8 in registers 0, 1, 2 of row 0, values 4, 5 / 2, 3 / 6, 7 in registers 0, 1 of rows
// 1 / 2 / 3 (the positions wave_sum9_rows leaves them at).  Lanes 0, 16, 32, 48 add registers 0 and 1 to the entry's LDS sums, lane 0 also
// register 2: three LDS atomics of 4 + 4 + 1 lanes instead of one of 9.  The sum of four fp32 terms in the core's fixed order: deterministic.
// MEASURED (cfg2, one call, three alternations): 8 % fewer VALU instructions (178.6 M -> 164.9 M) and the kernel 1.7 % SLOWER (0.2960 vs
// 0.2911 ms): the wave waits ~17 cycles on the matrix core per reduction, and it is the length of a wave's own chain, not the VALU issue
// count, that the kernel is short of.  With the next entry's operand requests between the MFMA and the first read of its result instead
// of the s_nops: 0.3135 ms -- the matrix core does not run beside the VALU here, an 8-pass MFMA costs the SIMD ~27 cycles of issue.
// Parity-green; off by default (make EXTRA=-DGSR_BWD_MFMA=1).
#ifndef GSR_BWD_MFMA
#define GSR_BWD_MFMA 0
#endif
typedef float gsr_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int gsr_sum9_mfma_slot0(int lane) { const int g = lane >> 4; return g == 0 ? 0 : g == 1 ? 4 : g == 2 ? 2 : 6; }

constexpr int NV = 9;  // partial sums per list entry
constexpr uint32_t GSR_NO_ROW = 0x0FFFFFFFu;  // 28 bits of gradient-row index (the host refuses more instances), 4 of quadrant mask

// Timing ablations (results are WRONG on purpose; never defined in the shipped build): make EXTRA=-DGSR_ABLATE=n
//   1: plain sum instead of the wave reduction   2: 1 + nothing written to LDS   3: no blend math after the alpha decision
//   4: nothing after the range test              5: no walk at all (staging + flush only)
//   6: (alone) no zero rows for the entries behind the tile's last contributor
#ifndef GSR_BWD_WPE
#define GSR_BWD_WPE 0  // experiment: waves per SIMD the register allocation aims at (0: 8)
#endif
// When an entry's operands are read from LDS: 1 = requested during the previous entry (after its quadrants, before its reduction), 0 = at
// the top of its own iteration, -1 = by the work split.  Measured in one call, ms: two waves per tile (cfg2) 0.332 / 0.333 piped vs
// 0.335 / 0.337 not; four waves per tile (2M Gaussians) 0.634 / 0.635 piped vs 0.604 / 0.603 not.
#ifndef GSR_BWD_BATCH1
#define GSR_BWD_BATCH1 128  // entries per round with four waves per tile (64: half the LDS, 8 instead of 6 waves per SIMD, twice the rounds)
#endif
#ifndef GSR_BWD_PIPE
#define GSR_BWD_PIPE -1
#endif
#ifndef GSR_ABLATE
#define GSR_ABLATE 0
#endif
#define GSR_ABL (GSR_ABLATE == 6 ? 0 : GSR_ABLATE)  // the walk's ablation level

// FAST_EXP: G = exp(power) from the hardware exp2 (v_exp_f32, <= 1e-6 relative off gsr_expf_blend on the range that can pass), 2 issue slots
// instead of the 9 instructions of the forward's fully specified gsr_expf_blend.  The alpha >= 1/255 decision must be the forward's: when
// any lane's alpha lies within 3e-8 of the threshold the wave redoes the entry with gsr_expf_blend, so the blended set is exactly the forward's.
template <int NQ, bool FAST_EXP>
__global__ void __launch_bounds__(64 * (4 / NQ)) __attribute__((amdgpu_waves_per_eu(GSR_BWD_WPE ? GSR_BWD_WPE : 8)))
render_backward_kernel(int W, int H, int gx, int gy, const uint32_t* __restrict__ order, const uint32_t* __restrict__ ranges,
                       const uint32_t* __restrict__ point_list, const float* __restrict__ splat, const float* __restrict__ bg,
                       const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
                       const float* __restrict__ dL_dpixels, float* __restrict__ grad_rows)
{
    constexpr int WPT = 4 / NQ;      // waves per tile
    constexpr int NT = 64 * WPT;     // threads
    constexpr int BATCH = NQ == 1 ? GSR_BWD_BATCH1 : 128;  // list entries staged per round
    constexpr int SW = BATCH / 64;                          // staging waves
    constexpr bool PIPE = GSR_BWD_PIPE < 0 ? NQ == 2 : GSR_BWD_PIPE != 0;
    __shared__ float4 s_a[BATCH];    // mx, my, conic.x, conic.y
    __shared__ float4 s_b[BATCH];    // conic.z, opacity, pthr, r
    __shared__ float2 s_c[BATCH];    // g, b
    // gradient row index (GSR_NO_ROW: nothing to flush) | quadrants the splat can reach << 28.  44 bytes per entry + 36 of sums: with two
    // waves per tile the workgroup's LDS is exactly 10,240 bytes = 1/16 of the CU's 160 KB, eight waves per SIMD
    __shared__ uint32_t s_row[BATCH];
    // wave totals of the batch's entries: zero-initialised rows that TWO waves add into with LDS float atomics (two addends: the sum does
    // not depend on their order).  NQ = 2: one row per entry; NQ = 1: one per wave pair, the two added in fixed order when the batch is
    // flushed (a row per wave took 24 KB of LDS per workgroup: 6 waves per SIMD; now 15 KB: 8)
    constexpr int NACC = NQ == 2 ? 1 : 2;
    __shared__ float s_acc[NACC * BATCH * NV];
    uint32_t* s_max = reinterpret_cast<uint32_t*>(s_acc);  // [WPT], only before the first round

    // 1-D launch in `order` (tiles by descending list length, binning.hip); NULL = row-major
    const uint32_t tile = order ? order[blockIdx.x] : blockIdx.x;
    const uint32_t bx = tile % (uint32_t)gx, by = tile / (uint32_t)gx;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(ranges[2 * tile]);
    const uint32_t r1 = __builtin_amdgcn_readfirstlane(ranges[2 * tile + 1]);
    const uint32_t total = r1 - r0;
    if (total == 0) return;

    // pixels of this lane: (lx, ly) of each of the wave's quadrants; quadrant q = 2 * (y half) + (x half)
    const uint32_t lx = lane & 7, ly = lane >> 3;
    const int q_first = wave * NQ;
    const float tile_x0 = (float)(bx * GSR_TILE_X), tile_y0 = (float)(by * GSR_TILE_Y);
    const float pxf0 = tile_x0 + (float)(8 * (q_first & 1)) + (float)lx, pxf1 = pxf0 + 8.0f;  // NQ = 2: x halves 0 and 1
    const float pyf = tile_y0 + (float)(8 * (q_first >> 1)) + (float)ly;
    const size_t plane = (size_t)H * W;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    float T[NQ], tfbg[NQ], behind[NQ], dL0[NQ], dL1[NQ], dL2[NQ];
    uint32_t lastc[NQ];
    unsigned long long possible[NQ], act[NQ];
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const uint32_t px = bx * GSR_TILE_X + (uint32_t)(8 * ((q_first + q) & 1)) + lx;
        const uint32_t py = by * GSR_TILE_Y + (uint32_t)(8 * ((q_first + q) >> 1)) + ly;
        const bool inside = px < (uint32_t)W && py < (uint32_t)H;
        const size_t pix_id = (size_t)W * py + px;
        const float Tf = inside ? final_Ts[pix_id] : 0.f;
        T[q] = Tf;
        lastc[q] = inside ? n_contrib[pix_id] : 0u;
        dL0[q] = dL1[q] = dL2[q] = 0.f;
        if (inside) { dL0[q] = dL_dpixels[pix_id]; dL1[q] = dL_dpixels[plane + pix_id]; dL2[q] = dL_dpixels[2 * plane + pix_id]; }
        tfbg[q] = Tf * (((0.f + bg0 * dL0[q]) + bg1 * dL1[q]) + bg2 * dL2[q]);  // T_final * (bg . dL/dpixel), backward.cu:526-530
        behind[q] = 0.f;
        possible[q] = __builtin_amdgcn_ballot_w64(lastc[q] > 0u);  // lanes that become active at some point of the walk
        act[q] = 0ull;
        m = max(m, lastc[q]);
    }
    // largest n_contrib of the tile = number of list entries that matter to any pixel
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    if (lane == 0) s_max[wave] = m;
    __syncthreads();
    uint32_t hi_ = s_max[0];
#pragma unroll
    for (int w = 1; w < WPT; w++) hi_ = max(hi_, s_max[w]);
    const uint32_t hi = __builtin_amdgcn_readfirstlane(hi_);

    // reduce-scatter constants: lane masks and the accumulator slot this lane writes
    const unsigned long long M0 = 0xAAAAAAAAAAAAAAAAull, M2 = 0x0004000400040004ull;
    const bool writer = gsr_sum9_writer(lane);
    const int my_slot = gsr_sum9_slot(lane);
    float* my_acc = s_acc + (NQ == 2 ? 0 : (wave >> 1) * BATCH * NV);
    const int rounds = (int)((hi + BATCH - 1) / BATCH);

    for (int i = 0; i < rounds; i++) {
        // ---- stage this batch (back to front): one record per thread; instances whose tile the splat cannot reach own no
        // gradient row (gsr_reach_mask, preprocess) and are dropped ----
        const uint32_t progress = (uint32_t)i * BATCH + tid;
        uint32_t qm = 0, row = GSR_NO_ROW;
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
        float pthr = 0.f;
        if (tid < BATCH && progress < hi) {
            const uint32_t k = hi - 1 - progress;
            const uint32_t id = point_list[r0 + k];
            const float4* rec = reinterpret_cast<const float4*>(splat + (size_t)id * GSR_SPLAT_F);
            q0 = rec[0]; q1 = rec[1]; q2 = rec[2];
            const float4 q3 = rec[3];  // gradient-row bookkeeping: first row, reachable tiles of the rectangle (common.h)
            uint32_t mnx, mny, mxx, mxy;
            gsr_get_rect(q0.x, q0.y, __float_as_int(q2.z), gx, gy, mnx, mny, mxx, mxy);
            const uint32_t tilepos = (by - mny) * (mxx - mnx) + (bx - mnx);
            pthr = q1.y > 0.f ? __logf(1.0f / (255.0f * q1.y)) - 1e-3f : -INFINITY;
            bool keep;
            if (__float_as_uint(q3.w) == 0u) {  // reach mask over the rectangle: the row number is the tile's rank among the reachable ones
                const unsigned long long mask = ((unsigned long long)__float_as_uint(q3.z) << 32) | __float_as_uint(q3.y);
                keep = (mask >> tilepos) & 1ull;
                row = __float_as_uint(q3.x) + (uint32_t)__popcll(mask & ((1ull << tilepos) - 1ull));
            } else {  // rectangle of more than 64 tiles: every instance owns a row; unreachable ones get their zero row right here
                row = __float_as_uint(q3.x) + tilepos;
                keep = !gsr_tile_cannot_touch(q0.x, q0.y, q0.z, q0.w, q1.x, pthr, tile_x0, tile_y0);
                if (!keep) {
                    const float z[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    gsr_row_store(grad_rows, row, z);
                }
            }
            // an instance that owns a row but reaches no quadrant (the two tests round differently) is flushed as a zero row
            if (keep) qm = gsr_quad_mask(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tile_x0, tile_y0);
            if (!keep) row = GSR_NO_ROW;
        }
        __syncthreads();  // previous batch flushed
        if (tid < BATCH) {
            s_a[tid] = q0;
            s_b[tid] = make_float4(q1.x, q1.y, fmaxf(pthr, -80.0f), q1.z);  // clamp: see the forward
            s_c[tid] = make_float2(q1.w, q2.x);
            s_row[tid] = row | (qm << 28);
#pragma unroll
            for (int t = 0; t < NACC * NV; t++) s_acc[(t / NV) * BATCH * NV + tid * NV + (t % NV)] = 0.f;
        }
        __syncthreads();

        // ---- walk the staged entries that can reach this wave's quadrants ----
        const uint32_t kbase = hi - 1u - (uint32_t)i * BATCH;  // list position of staged entry j = kbase - j (scalar)
#pragma unroll
        for (int w = 0; w < SW; w++) {
            unsigned long long bq[NQ];
            unsigned long long U = 0ull;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                // the staged entries that can reach quadrant q: a ballot over the bits the staging threads left beside the row index
                bq[q] = __builtin_amdgcn_ballot_w64(((s_row[64 * w + lane] >> (28 + q_first + q)) & 1u) != 0u);
                U |= bq[q];
            }
            if (GSR_ABL >= 5) U = 0ull;
            if (U == 0ull) continue;
            // PIPE: the operands of the NEXT entry are requested from LDS after this entry's quadrants have been visited and before its
            // wave reduction: the registers are free by then and the round trip hides behind the reduction
            int jj = (int)__builtin_ctzll(U);
            U &= U - 1ull;
            int j = 64 * w + jj;
            // (two register sets taking turns -- the loop body exists twice -- so that nothing is copied between iterations)
            float4 opA0 = s_a[j], opB0 = s_b[j], opA1, opB1;
            float2 opC0 = s_c[j], opC1;
            auto entry = [&](const float4& pa, const float4& pb, const float2& pc, float4& na, float4& nb, float2& nc)
                             __attribute__((always_inline)) -> bool {
                float4 a, b;
                float2 c;
                if (PIPE) { a = pa; b = pb; c = pc; } else { a = s_a[j]; b = s_b[j]; c = s_c[j]; }
                const uint32_t kidx = kbase - (uint32_t)j;
                const float dy = a.y - pyf;  // the wave's quadrants lie side by side: one dy per lane
                // what a blending quadrant leaves for the sums: w = G * dL/dalpha, dx, alpha * T (NQ = 2: combined below)
                // (separate scalars, not arrays: an array of two would be kept as a 64-bit register pair and copied into place)
                float W0, WX0, DX0, AT0, W1, WX1, DX1, AT1;  // defined where bl0 / bl1 say so
                auto visit = [&](auto Q, float& Wq, float& WXq, float& DXq, float& ATq) -> bool {
                    constexpr int q = decltype(Q)::value;
                    if (NQ > 1 && !((bq[q] >> jj) & 1ull)) return false;  // scalar: the splat cannot reach this quadrant
                    // lanes behind whose last contributor this entry lies never see it (backward.cu:480-482).  The walk runs back to
                    // front, so a lane once active stays active: the compare is only needed until every possible lane is
                    if (act[q] != possible[q]) act[q] = __builtin_amdgcn_ballot_w64(kidx < lastc[q]);
                    if (act[q] == 0ull) return false;
                    float dx;  // a.x - pixel x; as asm so that the result is born in the register it stays in until the sums are formed
                    asm("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(a.x), "v"((NQ > 1 && q) ? pxf1 : pxf0));
                    // identical to the forward (CONTRACT-FMA)
                    const float qq = __builtin_fmaf(a.z * dx, dx, (b.x * dy) * dy);
                    const float power = __builtin_fmaf(-0.5f, qq, -((a.w * dx) * dy));
                    // Lane predicates are kept as 64-bit lane masks in scalar registers (each ballot of a plain compare IS the
                    // compare's result mask): combining them and testing "any lane" is scalar work
                    const unsigned long long m_maybe = act[q] & __builtin_amdgcn_ballot_w64(power >= b.z);  // power > 0: below
                    if (m_maybe == 0ull) return false;
                    if (GSR_ABL >= 4) { T[q] += power; return false; }
                    unsigned long long m_c = m_maybe & __builtin_amdgcn_ballot_w64(power <= 0.0f);
                    float G = FAST_EXP ? __builtin_amdgcn_exp2f(power * 1.44269504088896341f) : gsr_expf_blend(power);
                    float alpha = fminf(0.99f, b.y * G);
                    if (FAST_EXP && (m_c & __builtin_amdgcn_ballot_w64(fabsf(alpha - 1.0f / 255.0f) < 3e-8f)) != 0ull) {
                        asm volatile("" ::: "memory");  // a real (rarely taken) branch: keeps the compiler from speculating the 15 instructions
                        G = gsr_expf_blend(power);
                        alpha = fminf(0.99f, b.y * G);
                    }
                    m_c &= __builtin_amdgcn_ballot_w64(alpha >= 1.0f / 255.0f);
                    if (m_c == 0ull) return false;
                    if (GSR_ABL >= 3) { T[q] += alpha; return false; }
                    // From here all 64 lanes run the same straight-line code.  A lane that does not contribute is given alpha = G = 0:
                    // then T/(1-0) = T, its gradient terms vanish, and the colour recurrence (backward.cu:515) leaves accum unchanged
                    float Ge, ae;
                    asm("v_cndmask_b32 %0, 0, %2, %4\n v_cndmask_b32 %1, 0, %3, %4" : "=&v"(Ge), "=v"(ae) : "v"(G), "v"(alpha), "s"(m_c));
                    const float oma = 1.f - ae;
                    const float inv = __builtin_amdgcn_rcpf(oma);
                    T[q] = T[q] * inv;  // T / (1 - alpha), backward.cu:502
                    // The reference carries the colour accumulated behind this entry as a vector, accum = last_alpha * last_color +
                    // (1 - last_alpha) * accum, and forms sum_ch (c - accum) * dL/dpixel (backward.cu:505-520).  Only accum . dL/dpixel
                    // is ever used, and it obeys the same recurrence with the scalar c . dL/dpixel in place of the colour: one
                    // register per pixel instead of three, 6 instructions instead of 12.  `behind` = accum . dL/dpixel of the
                    // entries behind this one (the recurrence already applied for the previous entry)
                    const float cdl = __builtin_fmaf(c.y, dL2[q], __builtin_fmaf(c.x, dL1[q], b.w * dL0[q]));
                    float dL_dalpha = cdl - behind[q];
                    behind[q] = __builtin_fmaf(ae, cdl, oma * behind[q]);
                    dL_dalpha *= T[q];
                    dL_dalpha = __builtin_fmaf(-tfbg[q], inv, dL_dalpha);  // (-T_final / (1 - alpha)) * (bg . dL/dpixel)
                    Wq = Ge * dL_dalpha;
                    WXq = Wq * dx;
                    DXq = dx;
                    ATq = ae * T[q];
                    return true;
                };
                const bool bl0 = visit(std::integral_constant<int, 0>{}, W0, WX0, DX0, AT0);
                const bool bl1 = NQ > 1 ? visit(std::integral_constant<int, NQ - 1>{}, W1, WX1, DX1, AT1) : false;
                const bool more = U != 0ull;
                const int jjn = more ? (int)__builtin_ctzll(U) : jj;
                U &= U - 1ull;
                const int jn = 64 * w + jjn;
                if (PIPE) { na = s_a[jn]; nb = s_b[jn]; nc = s_c[jn]; }
                if constexpr (NQ == 1) {  // one quadrant per wave
                    if (bl0) {
                        const float wdx = WX0, wdy = W0 * dy;
                        const float tot = wave_sum9_scatter(W0, wdx, wdy, wdx * DX0, wdx * dy, wdy * dy, AT0 * dL0[0], AT0 * dL1[0],
                                                            AT0 * dL2[0], M0, M2);
                        if (writer) atomicAdd(&my_acc[j * NV + my_slot], tot);  // LDS
                    }
                } else if (bl0 || bl1) {  // wave-uniform
                    // the lane's two pixels share dy: the y moments of their sum are products of the x moments with dy
                    float sw, swx, sxx, c0, c1, c2;
                    if (bl0 && bl1) {
                        sw = W0 + W1;
                        swx = WX0 + WX1;
                        sxx = __builtin_fmaf(WX1, DX1, WX0 * DX0);
                        c0 = __builtin_fmaf(AT1, dL0[1], AT0 * dL0[0]);
                        c1 = __builtin_fmaf(AT1, dL1[1], AT0 * dL1[0]);
                        c2 = __builtin_fmaf(AT1, dL2[1], AT0 * dL2[0]);
                    } else if (bl0) {
                        sw = W0; swx = WX0; sxx = swx * DX0;
                        c0 = AT0 * dL0[0]; c1 = AT0 * dL1[0]; c2 = AT0 * dL2[0];
                    } else {
                        sw = W1; swx = WX1; sxx = swx * DX1;
                        c0 = AT1 * dL0[1]; c1 = AT1 * dL1[1]; c2 = AT1 * dL2[1];
                    }
                    const float swy = sw * dy;
#if GSR_ABL >= 1
                    const float tot = ((sw + swx) + (swy + sxx)) + ((swx * dy + swy * dy) + (c0 + c1)) + c2;
                    if (GSR_ABL < 2 ? writer : tot == 12345.678f) atomicAdd(&my_acc[j * NV + my_slot], tot);
#else
                    const float tot = wave_sum9_scatter(sw, swx, swy, sxx, swx * dy, swy * dy, c0, c1, c2, M0, M2);
                    if (writer) atomicAdd(&my_acc[j * NV + my_slot], tot);  // LDS
#endif
                }
                jj = jjn; j = jn;
                return more;
            };
            while (entry(opA0, opB0, opC0, opA1, opB1, opC1) && entry(opA1, opB1, opC1, opA0, opB0, opC0)) {}
        }
        __syncthreads();
        // ---- flush: entry `tid` of the batch -> one gradient row ----
        if (tid < BATCH) {
            const uint32_t frow = s_row[tid] & GSR_NO_ROW;
            if (frow != GSR_NO_ROW) {
                float r[NV];
#pragma unroll
                for (int t = 0; t < NV; t++) r[t] = NACC == 1 ? s_acc[tid * NV + t] : s_acc[tid * NV + t] + s_acc[(BATCH + tid) * NV + t];
                // per-Gaussian factors of backward.cu:538-554 applied once to the reduced moments
                const float4 a = s_a[tid];
                const float4 b = s_b[tid];
                const float op = b.y, cx = a.z, cy = a.w, cz = b.x;
                const float gmx = op * (-cx * r[1] - cy * r[2]) * ddelx_dx;
                const float gmy = op * (-cz * r[2] - cy * r[1]) * ddely_dy;
                const float hop = -0.5f * op;
                const float out9[9] = {gmx, gmy, hop * r[3], hop * r[4], hop * r[5], r[0], r[6], r[7], r[8]};
                gsr_row_store(grad_rows, frow, out9);
            }
        }
    }
#if GSR_ABL >= 3
    {   // keeps the ablated walk's arithmetic alive
        float keep = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; q++) keep += T[q];
        if (keep == 12345.678f) grad_rows[0] = keep;
    }
#endif
    // list entries behind every pixel's last contributor: zero rows (for those that own a row)
    for (uint32_t e = hi + tid; e < total && GSR_ABLATE != 6; e += NT) {  // (ablation 6: what the zero rows cost)
        const uint32_t id = point_list[r0 + e];
        const float4* rec = reinterpret_cast<const float4*>(splat + (size_t)id * GSR_SPLAT_F);
        const float4 q0 = rec[0], q2 = rec[2], q3 = rec[3];
        uint32_t mnx, mny, mxx, mxy;
        gsr_get_rect(q0.x, q0.y, __float_as_int(q2.z), gx, gy, mnx, mny, mxx, mxy);
        const uint32_t tilepos = (by - mny) * (mxx - mnx) + (bx - mnx);
        uint32_t row;
        if (__float_as_uint(q3.w) == 0u) {
            const unsigned long long mask = ((unsigned long long)__float_as_uint(q3.z) << 32) | __float_as_uint(q3.y);
            if (!((mask >> tilepos) & 1ull)) continue;
            row = __float_as_uint(q3.x) + (uint32_t)__popcll(mask & ((1ull << tilepos) - 1ull));
        } else
            row = __float_as_uint(q3.x) + tilepos;
        const float z[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        gsr_row_store(grad_rows, row, z);
    }
}


// Profiling builds of the log walk (never shipped): -DGSR_LOG_PROFILE=2 counts entries / visits / blended pixels into the same counters
// (tools/r5_counts.py), =3 time-stamps every wave (tools/r5_wave_times.py), and
// Phase profile of the log walk (make EXTRA=-DGSR_LOG_PROFILE=1, never in the shipped build): wave-cycles (s_memtime) per phase, summed over all
// waves of all launches since the last gsr_debug_blend_profile call: [0] whole wave, [1] prologue, [2] staging loads + tests, [3] wait at the
// "previous batch flushed" barrier, [4] LDS writes + barrier, [5] walk, [6] wait at the end-of-walk barrier (the other wave is still walking),
// [7] flush + zero rows.
#ifndef GSR_LOG_PROFILE
#define GSR_LOG_PROFILE 0
#endif
#if GSR_LOG_PROFILE == 3
__device__ unsigned long long g_wave_times[4 * 32768];  // per wave of the last launch: start, end (s_memrealtime, 100 MHz), HW_ID, XCC_ID
#endif
__device__ unsigned long long g_blend_prof[8];  // ([4] holds the zero rows + end of the wave in profiling builds: the LDS writes are folded into [3])
#define GSR_PROF_MARK(slot) do { if (GSR_LOG_PROFILE == 1) { const long long t_ = clock64(); prof[slot] += (unsigned long long)(t_ - t_last); t_last = t_; } } while (0)

// ---- the log walk (round 5) ----------------------------------------------------------------------------------------------------------
// The same kernel fed with the forward's blend log (render_forward.hip): for every (list entry, quadrant) visit the forward made, the
// 64-bit mask of the pixels that blended the entry.  Everything render_backward_kernel re-derives per visit -- is the entry in front of
// the pixel's last contributor (n_contrib), does power pass the threshold, is power <= 0, is alpha >= 1/255 (with an exact second exp when
// the fast one lands near the threshold) -- is ONE scalar load here, and the set of blended pairs is the forward's by construction.
// Why it pays (tools/smem_probe.hip, profiles/README.md round 5): a SALU instruction costs 1.74 ns per SIMD on gfx950 -- more than a
// plain VALU instruction (1.1 ns) -- and the decision chains were ~27 scalar instructions per visit (ballot, s_and, s_cmp, s_cbranch, four
// times over) beside ~10 of the visit's 44 VALU issue slots; both pipes of the old kernel ran at 75-90 %.
// Which quadrants an entry is walked for also comes from the log: the forward stores whole 64-entry windows (zeros where nothing
// blended), so a staging thread reads its entry's four masks and keeps a quadrant bit where the mask is non-zero and the slot is one the
// forward wrote (k < the quadrant's last contributor, quad_hi: behind the forward wave's exit nothing was stored) -- no reach test per
// quadrant (gsr_quad_mask: ~100 instructions, six square roots), and the walk visits exactly the (entry, quadrant) pairs that blend.
// The kernel is VALU-throughput-bound (an instruction added per entry costs exactly its issue time, profiles/README.md): 25 instructions
// per visit, 38 per wave reduction, ~9 per entry.
template <int NQ>
__global__ void __launch_bounds__(64 * (4 / NQ)) __attribute__((amdgpu_waves_per_eu(GSR_BWD_WPE ? GSR_BWD_WPE : 8)))
render_backward_log_kernel(int W, int H, int gx, int gy, const uint32_t* __restrict__ order, const uint32_t* __restrict__ ranges,
                           const uint32_t* __restrict__ point_list, const float* __restrict__ splat, const float* __restrict__ bg,
                           const float* __restrict__ final_Ts, const uint32_t* __restrict__ quad_hi,
                           const float* __restrict__ dL_dpixels, const unsigned long long* __restrict__ blend_log,
                           float* __restrict__ grad_rows)
{
    constexpr int WPT = 4 / NQ;      // waves per tile
    constexpr int NT = 64 * WPT;     // threads
    constexpr int BATCH = NQ == 1 ? GSR_BWD_BATCH1 : 128;  // list entries staged per round
    constexpr int SW = BATCH / 64;                          // staging waves
    __shared__ float4 s_a[BATCH];    // mx, my, conic.x, conic.y
    __shared__ float4 s_b[BATCH];    // conic.z, opacity, -, r
    __shared__ float2 s_c[BATCH];    // g, b
    __shared__ uint32_t s_row[BATCH];  // gradient row index | (quadrants that blended the entry, by the log) << 28
    constexpr int NACC = NQ == 2 ? 1 : 2;
    __shared__ float s_acc[NACC * BATCH * NV];

    const uint32_t tile = order ? order[blockIdx.x] : blockIdx.x;
    const uint32_t bx = tile % (uint32_t)gx, by = tile / (uint32_t)gx;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(ranges[2 * tile]);
    const uint32_t r1 = __builtin_amdgcn_readfirstlane(ranges[2 * tile + 1]);
    const uint32_t total = r1 - r0;
    if (total == 0) return;
    // largest n_contrib of each quadrant (stored by the forward beside its log): how far the quadrant's log goes.  Scalar loads: the
    // tile's bounds are known before anything per pixel has been read
    const uint32_t h0 = __builtin_amdgcn_readfirstlane(quad_hi[4 * tile]), h1 = __builtin_amdgcn_readfirstlane(quad_hi[4 * tile + 1]);
    const uint32_t h2 = __builtin_amdgcn_readfirstlane(quad_hi[4 * tile + 2]), h3 = __builtin_amdgcn_readfirstlane(quad_hi[4 * tile + 3]);
    const uint32_t hi = max(max(h0, h1), max(h2, h3));  // number of list entries that matter to any pixel of the tile
#if GSR_LOG_PROFILE == 3
    const unsigned long long wt_begin = __builtin_amdgcn_s_memrealtime();
#endif
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_last = GSR_LOG_PROFILE ? clock64() : 0;
    const long long t_begin = t_last;

    const uint32_t lx = lane & 7, ly = lane >> 3;
    const int q_first = wave * NQ;
    const float tile_x0 = (float)(bx * GSR_TILE_X), tile_y0 = (float)(by * GSR_TILE_Y);
    const float pxf0 = tile_x0 + (float)(8 * (q_first & 1)) + (float)lx, pxf1 = pxf0 + 8.0f;  // NQ = 2: x halves 0 and 1
    const float pyf = tile_y0 + (float)(8 * (q_first >> 1)) + (float)ly;
    const size_t plane = (size_t)H * W;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    float T[NQ], tfbg[NQ], behind[NQ], dL0[NQ], dL1[NQ], dL2[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const uint32_t px = bx * GSR_TILE_X + (uint32_t)(8 * ((q_first + q) & 1)) + lx;
        const uint32_t py = by * GSR_TILE_Y + (uint32_t)(8 * ((q_first + q) >> 1)) + ly;
        const bool inside = px < (uint32_t)W && py < (uint32_t)H;
        const size_t pix_id = (size_t)W * py + px;
        const float Tf = inside ? final_Ts[pix_id] : 0.f;
        T[q] = Tf;
        dL0[q] = dL1[q] = dL2[q] = 0.f;
        if (inside) { dL0[q] = dL_dpixels[pix_id]; dL1[q] = dL_dpixels[plane + pix_id]; dL2[q] = dL_dpixels[2 * plane + pix_id]; }
        tfbg[q] = Tf * (((0.f + bg0 * dL0[q]) + bg1 * dL1[q]) + bg2 * dL2[q]);  // T_final * (bg . dL/dpixel), backward.cu:526-530
        behind[q] = 0.f;
    }
    // list entries behind every pixel's last contributor: zero rows (for those that own a row).  At the END of the kernel: written first
    // (measured, round 5: +0.027 ms) their stores sit in the memory pipeline in front of the first batch's loads
    auto zero_rows = [&]() __attribute__((always_inline)) {
        for (uint32_t e = hi + tid; e < total; e += NT) {
            const uint32_t id = point_list[r0 + e];
            const float4* rec = reinterpret_cast<const float4*>(splat + (size_t)id * GSR_SPLAT_F);
            const float4 q0 = rec[0], q2 = rec[2], q3 = rec[3];
            uint32_t mnx, mny, mxx, mxy;
            gsr_get_rect(q0.x, q0.y, __float_as_int(q2.z), gx, gy, mnx, mny, mxx, mxy);
            const uint32_t tilepos = (by - mny) * (mxx - mnx) + (bx - mnx);
            uint32_t row;
            if (__float_as_uint(q3.w) == 0u) {
                const unsigned long long mask = ((unsigned long long)__float_as_uint(q3.z) << 32) | __float_as_uint(q3.y);
                if (!((mask >> tilepos) & 1ull)) continue;
                row = __float_as_uint(q3.x) + (uint32_t)__popcll(mask & ((1ull << tilepos) - 1ull));
            } else
                row = __float_as_uint(q3.x) + tilepos;
            const float z[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            gsr_row_store(grad_rows, row, z);
        }
    };
    GSR_PROF_MARK(1);

    unsigned long long M0 = 0xAAAAAAAAAAAAAAAAull, M2 = 0x0004000400040004ull;
    asm volatile("" : "+s"(M0), "+s"(M2));  // kept as two register PAIRS (the halves are equal: the compiler would rebuild the pair per use)
    const bool writer = gsr_sum9_writer(lane);
    const int my_slot = gsr_sum9_slot(lane);
    float* my_acc = s_acc + (NQ == 2 ? 0 : (wave >> 1) * BATCH * NV);
    // this lane's slot of entry 0 as an INDEX into s_acc (32-bit arithmetic per reduction: through the pointer the compiler formed the
    // address with a 64-bit v_mad_u64_u32 per reduction)
    const uint32_t acc_slot0 = (uint32_t)((NQ == 2 ? 0 : (wave >> 1) * BATCH * NV) + my_slot);
    float* acc_lane = my_acc + gsr_sum9_mfma_slot0(lane);  // GSR_BWD_MFMA: where this lane's result registers 0, 1 (+ 8: register 2 of lane 0) go
    const bool w4 = (lane & 15) == 0, w1 = lane == 0;
    const int rounds = (int)((hi + BATCH - 1) / BATCH);
    // the masks of (list position 0 of the tile, this wave's first quadrant); position k lies 4 k words further
    const unsigned long long* logw = blend_log + gsr_blend_log_slot(r0, (uint32_t)q_first);

    // (the next round's Gaussian ids loaded a round ahead -- one of the two dependent load latencies of a staging round behind the
    // previous walk -- measured, round 5: 0.2913 vs 0.2917 ms, not kept)
    for (int i = 0; i < rounds; i++) {
        // ---- stage this batch (back to front): as render_backward_kernel, but the quadrant bits come from the log ----
        const uint32_t progress = (uint32_t)i * BATCH + tid;
        uint32_t qm = 0, row = GSR_NO_ROW;
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
        if (tid < BATCH && progress < hi) {
            const uint32_t k = hi - 1 - progress;
            const uint32_t id = point_list[r0 + k];
            const float4* rec = reinterpret_cast<const float4*>(splat + (size_t)id * GSR_SPLAT_F);
            q0 = rec[0]; q1 = rec[1]; q2 = rec[2];
            const float4 q3 = rec[3];
            uint32_t mnx, mny, mxx, mxy;
            gsr_get_rect(q0.x, q0.y, __float_as_int(q2.z), gx, gy, mnx, mny, mxx, mxy);
            const uint32_t tilepos = (by - mny) * (mxx - mnx) + (bx - mnx);
            bool keep;
            if (__float_as_uint(q3.w) == 0u) {
                const unsigned long long mask = ((unsigned long long)__float_as_uint(q3.z) << 32) | __float_as_uint(q3.y);
                keep = (mask >> tilepos) & 1ull;
                row = __float_as_uint(q3.x) + (uint32_t)__popcll(mask & ((1ull << tilepos) - 1ull));
            } else {
                const float pthr = q1.y > 0.f ? __logf(1.0f / (255.0f * q1.y)) - 1e-3f : -INFINITY;
                row = __float_as_uint(q3.x) + tilepos;
                keep = !gsr_tile_cannot_touch(q0.x, q0.y, q0.z, q0.w, q1.x, pthr, tile_x0, tile_y0);
                if (!keep) {
                    const float z[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    gsr_row_store(grad_rows, row, z);
                }
            }
            if (keep) {
                // which quadrants blended the entry: its four masks in the forward's log (32 bytes per list position, coalesced over the
                // staging threads) instead of the reach test per quadrant (gsr_quad_mask: ~100 instructions, six square roots).  A slot
                // is valid for k < the quadrant's last contributor (the forward stores whole windows, zeros where nothing blended; behind
                // its exit nothing was stored).  The walk then visits exactly the (entry, quadrant) pairs that blend.
                const ulonglong2* mp = reinterpret_cast<const ulonglong2*>(blend_log + gsr_blend_log_slot(r0 + k, 0u));
                const ulonglong2 m01 = mp[0], m23 = mp[1];
                qm = ((m01.x != 0ull && k < h0) ? 1u : 0u) | ((m01.y != 0ull && k < h1) ? 2u : 0u) |
                     ((m23.x != 0ull && k < h2) ? 4u : 0u) | ((m23.y != 0ull && k < h3) ? 8u : 0u);
            }
            if (!keep) row = GSR_NO_ROW;
        }
        GSR_PROF_MARK(2);
        if (GSR_LOG_PROFILE == 2) {  // counting build: [6] staged entries some quadrant blended, [7] their quadrant bits
            prof[6] += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(qm != 0u));
            prof[7] += (unsigned long long)(__popcll(__builtin_amdgcn_ballot_w64((qm & 1u) != 0u)) + __popcll(__builtin_amdgcn_ballot_w64((qm & 2u) != 0u)) +
                                            __popcll(__builtin_amdgcn_ballot_w64((qm & 4u) != 0u)) + __popcll(__builtin_amdgcn_ballot_w64((qm & 8u) != 0u)));
        }
        __syncthreads();  // previous batch flushed
        GSR_PROF_MARK(3);
        if (tid < BATCH) {
            s_a[tid] = q0;
            s_b[tid] = make_float4(q1.x, q1.y, 0.f, q1.z);
            s_c[tid] = make_float2(q1.w, q2.x);
            s_row[tid] = row | (qm << 28);
#pragma unroll
            for (int t = 0; t < NACC * NV; t++) s_acc[(t / NV) * BATCH * NV + tid * NV + (t % NV)] = 0.f;
        }
        __syncthreads();
        GSR_PROF_MARK(3);

        // ---- walk the staged entries whose log slots exist for this wave's quadrants ----
        const uint32_t kbase = hi - 1u - (uint32_t)i * BATCH;  // list position of staged entry j = kbase - j (scalar)
#pragma unroll
        for (int w = 0; w < SW; w++) {
            unsigned long long bq[NQ];
            unsigned long long U = 0ull;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                bq[q] = __builtin_amdgcn_ballot_w64(((s_row[64 * w + lane] >> (28 + q_first + q)) & 1u) != 0u);
                U |= bq[q];
            }
            if (U == 0ull) continue;
            int jj = (int)__builtin_ctzll(U);
            U &= U - 1ull;
            int j = 64 * w + jj;
            // an entry's masks: NQ consecutive 64-bit words, read with a scalar load (uniform address, read-only memory)
            struct Masks { unsigned long long m[NQ]; };
            auto load_masks = [&](int jx) -> Masks {
                Masks r;
                const unsigned long long* p = logw + 4 * (size_t)(kbase - (uint32_t)jx);
#pragma unroll
                for (int q = 0; q < NQ; q++) r.m[q] = p[q];
                return r;
            };
            // (two register sets taking turns, as in render_backward_kernel)
            float4 opA0 = s_a[j], opB0 = s_b[j], opA1, opB1;
            float2 opC0 = s_c[j], opC1;
            Masks opM0 = load_masks(j), opM1;
            auto entry = [&](const float4& a, const float4& b, const float2& c, const Masks& mk, float4& na, float4& nb, float2& nc,
                             Masks& nm) __attribute__((always_inline)) -> bool {
                // this entry's masks: the forward's where the quadrant's bit was staged (valid slot, non-zero mask), else nothing (the slot may
                // never have been written)
                unsigned long long mq[NQ];
#pragma unroll
                for (int q = 0; q < NQ; q++) mq[q] = ((bq[q] >> jj) & 1ull) ? mk.m[q] : 0ull;
                const int jcur = j;
                const bool more = U != 0ull;
                // The NEXT entry's operands (LDS) are requested after this entry's quadrants and before its reduction: the registers are free
                // by then and the round trip hides behind the reduction.  Its masks (a scalar load from L2 / HBM) are requested EARLIER, right
                // behind the first use of this entry's operands (dy, cyy below): scalar loads return out of order, so while one is
                // outstanding every wait for LDS data is a wait for it as well (lgkmcnt(0)) -- the wait for this entry's operands must lie
                // in front of it, and the wait of the reduction's cross-row exchange as far behind it as possible (the visits and the
                // in-row levels).  The scheduling barriers pin it there.
                const int jjn = more ? (int)__builtin_ctzll(U) : jj;
                U &= U - 1ull;
                const int jn = 64 * w + jjn;
                auto request_next = [&]() __attribute__((always_inline)) {
                    na = s_a[jn]; nb = s_b[jn]; nc = s_c[jn];
                    if (!GSR_BWD_EARLYMASK) nm = load_masks(jn);
                    jj = jjn; j = jn;
                };
                const float dy = a.y - pyf;  // the wave's quadrants lie side by side: one dy per lane
                const float cyy = (b.x * dy) * dy;  // shared by the wave's quadrants (same operation order as the forward's)
                if (GSR_BWD_EARLYMASK) {
                    __builtin_amdgcn_sched_barrier(0);
                    nm = load_masks(jn);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // one (entry, quadrant) visit with the forward's mask m_c != 0: what the quadrant leaves for the sums
                auto visit = [&](auto Q, unsigned long long m_c, float& Wq, float& WXq, float& DXq, float& ATq) __attribute__((always_inline)) {
                    constexpr int q = decltype(Q)::value;
                    float dx;
                    asm("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(a.x), "v"((NQ > 1 && q) ? pxf1 : pxf0));
                    // as the forward (CONTRACT-FMA); G from the hardware exp2: <= 1e-6 relative off the forward's gsr_expf_blend
                    const float qq = __builtin_fmaf(a.z * dx, dx, cyy);
                    const float power = __builtin_fmaf(-0.5f, qq, -((a.w * dx) * dy));
                    const float G = __builtin_amdgcn_exp2f(power * 1.44269504088896341f);
                    const float alpha = fminf(0.99f, b.y * G);
                    // all 64 lanes run the same straight-line code; a lane outside the mask is given alpha = G = 0 (whatever its own
                    // power gave: the select discards it): T / (1 - 0) = T, its terms vanish, `behind` stays
                    float Ge, ae;
                    asm("v_cndmask_b32 %0, 0, %2, %4\n v_cndmask_b32 %1, 0, %3, %4" : "=&v"(Ge), "=v"(ae) : "v"(G), "v"(alpha), "s"(m_c));
                    const float oma = 1.f - ae;
                    const float inv = __builtin_amdgcn_rcpf(oma);
                    T[q] = T[q] * inv;  // T / (1 - alpha), backward.cu:502
                    // accum . dL/dpixel carried as one scalar per pixel (see render_backward_kernel); recurrence accum' = accum +
                    // alpha (c . dL - accum), backward.cu:515 regrouped: one fma on the difference dL/dalpha needs anyway
                    const float cdl = __builtin_fmaf(c.y, dL2[q], __builtin_fmaf(c.x, dL1[q], b.w * dL0[q]));
                    const float diff = cdl - behind[q];
                    behind[q] = __builtin_fmaf(ae, diff, behind[q]);
                    float dL_dalpha = diff * T[q];
                    dL_dalpha = __builtin_fmaf(-tfbg[q], inv, dL_dalpha);  // (-T_final / (1 - alpha)) * (bg . dL/dpixel)
                    Wq = Ge * dL_dalpha;
                    WXq = Wq * dx;
                    DXq = dx;
                    ATq = ae * T[q];
                };
                auto reduce = [&](float sw, float swx, float sxx, float c0, float c1, float c2) __attribute__((always_inline)) {
                    // the lane's pixels share dy: the y moments of their sum are products of the x moments with dy
                    const float swy = sw * dy;
                    if (GSR_BWD_MFMA) {
                        const float rows = wave_sum9_rows(sw, swx, swy, sxx, swx * dy, swy * dy, c0, c1, c2, M0, M2);
                        // (as asm: the builtin let the register allocator put the result tuple over the A operand -- v_mfma v[10:13], v11,
                        // v48, 0 -- and the sums came out wrong; "=&v" keeps the four result registers apart from both sources.  The
                        // wait states before the result may be read -- 8 passes -- are in the block: the compiler knows nothing of it.  And
                        // the s_nop in FRONT: the A operand was written by the asm block above, invisible to the hazard recognizer; without
                        // it the value parked last -- the ninth -- arrived stale.)
                        gsr_f4v d;
                        asm volatile("s_nop 4\n v_mfma_f32_16x16x4_f32 %0, %1, %2, 0\n s_nop 7\n s_nop 3" : "=&v"(d) : "v"(rows), "v"(1.0f));
                        float* o = acc_lane + jcur * NV;
                        if (w4) { atomicAdd(o, d[0]); atomicAdd(o + 1, d[1]); }  // LDS
                        if (w1) atomicAdd(o + 8, d[2]);
                    } else {
                        const float tot = wave_sum9_scatter<(GSR_BWD_XROW == 2 || (GSR_BWD_XROW == 1 && NQ == 2))>(sw, swx, swy, sxx, swx * dy, swy * dy, c0, c1,
                                                                                                                c2, M0, M2);
                        // (an LDS-typed pointer: 32-bit address arithmetic; through the generic pointer the compiler formed the address with a
                        // 64-bit v_mad_u64_u32 per reduction)
                        typedef __attribute__((address_space(3))) float lds_f32;
                        uint32_t eoff = (uint32_t)jcur * NV;
                        if (GSR_BWD_SMULADDR) asm("s_mul_i32 %0, %1, %2" : "=s"(eoff) : "s"((uint32_t)jcur), "n"(NV));  // (on the scalar side: the compiler picks a v_mad_u64_u32)
                        if (writer)
                            __hip_atomic_fetch_add((lds_f32*)s_acc + (acc_slot0 + eoff), tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                };
                // Each case is ONE straight-line block (two quadrants: their two chains -- sub, 6 mul/fma, exp, select, rcp, 10 mul/fma --
                // are independent and interleave) and ends in its own copy of the reduction: nothing is merged behind the branches.
                float W0, WX0, DX0, AT0, W1, WX1, DX1, AT1;
                if constexpr (NQ == 1) {
                    if (GSR_LOG_PROFILE == 2) { prof[1]++; prof[mq[0] != 0ull ? 3 : 4]++; prof[5] += (unsigned long long)__popcll(mq[0]); }
                    if (mq[0] != 0ull) {
                        visit(std::integral_constant<int, 0>{}, mq[0], W0, WX0, DX0, AT0);
                        request_next();
                        reduce(W0, WX0, WX0 * DX0, AT0 * dL0[0], AT0 * dL1[0], AT0 * dL2[0]);
                    } else {
                        request_next();
                    }
                } else {
                    const bool v0 = mq[0] != 0ull, v1 = mq[NQ - 1] != 0ull;
                    if (GSR_LOG_PROFILE == 2) {  // counting build: [1] entries walked, [2] both quadrants, [3] one, [4] none, [5] blended pixels
                        prof[1]++; prof[v0 && v1 ? 2 : (v0 || v1) ? 3 : 4]++;
                        prof[5] += (unsigned long long)(__popcll(mq[0]) + __popcll(mq[NQ - 1]));
                    }
                    if (v0 && v1) {
                        visit(std::integral_constant<int, 0>{}, mq[0], W0, WX0, DX0, AT0);
                        visit(std::integral_constant<int, NQ - 1>{}, mq[NQ - 1], W1, WX1, DX1, AT1);
                        request_next();
                        reduce(W0 + W1, WX0 + WX1, __builtin_fmaf(WX1, DX1, WX0 * DX0), __builtin_fmaf(AT1, dL0[NQ - 1], AT0 * dL0[0]),
                               __builtin_fmaf(AT1, dL1[NQ - 1], AT0 * dL1[0]), __builtin_fmaf(AT1, dL2[NQ - 1], AT0 * dL2[0]));
                    } else if (v0) {
                        visit(std::integral_constant<int, 0>{}, mq[0], W0, WX0, DX0, AT0);
                        request_next();
                        reduce(W0, WX0, WX0 * DX0, AT0 * dL0[0], AT0 * dL1[0], AT0 * dL2[0]);
                    } else if (v1) {
                        visit(std::integral_constant<int, NQ - 1>{}, mq[NQ - 1], W1, WX1, DX1, AT1);
                        request_next();
                        reduce(W1, WX1, WX1 * DX1, AT1 * dL0[NQ - 1], AT1 * dL1[NQ - 1], AT1 * dL2[NQ - 1]);
                    } else {
                        request_next();
                    }
                }
                return more;
            };
            while (entry(opA0, opB0, opC0, opM0, opA1, opB1, opC1, opM1) && entry(opA1, opB1, opC1, opM1, opA0, opB0, opC0, opM0)) {}
        }
        GSR_PROF_MARK(5);
        __syncthreads();
        GSR_PROF_MARK(6);
        // ---- flush: entry `tid` of the batch -> one gradient row ----
        if (tid < BATCH) {
            const uint32_t frow = s_row[tid] & GSR_NO_ROW;
            if (frow != GSR_NO_ROW) {
                float r[NV];
#pragma unroll
                for (int t = 0; t < NV; t++) r[t] = NACC == 1 ? s_acc[tid * NV + t] : s_acc[tid * NV + t] + s_acc[(BATCH + tid) * NV + t];
                const float4 a = s_a[tid];
                const float4 b = s_b[tid];
                const float op = b.y, cx = a.z, cy = a.w, cz = b.x;
                const float gmx = op * (-cx * r[1] - cy * r[2]) * ddelx_dx;
                const float gmy = op * (-cz * r[2] - cy * r[1]) * ddely_dy;
                const float hop = -0.5f * op;
                const float out9[9] = {gmx, gmy, hop * r[3], hop * r[4], hop * r[5], r[0], r[6], r[7], r[8]};
                gsr_row_store(grad_rows, frow, out9);
            }
        }
        GSR_PROF_MARK(7);
    }
    zero_rows();
#if GSR_LOG_PROFILE == 3
    if (lane == 0) {  // time-stamp build: this wave's start, end, list length | HW_ID, entries walked | XCC_ID
        const size_t wi = ((size_t)blockIdx.x * WPT + (size_t)wave) & 32767u;
        g_wave_times[4 * wi] = wt_begin;
        g_wave_times[4 * wi + 1] = __builtin_amdgcn_s_memrealtime();
        g_wave_times[4 * wi + 2] = ((unsigned long long)total << 32) | (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_wave_times[4 * wi + 3] = ((unsigned long long)hi << 32) | (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    if (GSR_LOG_PROFILE == 1 || GSR_LOG_PROFILE == 2) {
        GSR_PROF_MARK(4);
        prof[0] = (unsigned long long)(clock64() - t_begin);
        if (lane == 0)
            for (int t = 0; t < 8; t++) atomicAdd(&g_blend_prof[t], prof[t]);
    }
}

}  // namespace

// mean_list: mean length of the reference's tile lists (num_rendered / tiles), known to the host.  BWD_NQ: 1 / 2 force a work split,
// 0 (default) picks two quadrants per wave when the lists are short (see the header).
void gsr_launch_render_backward(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* splat,
                                const float* bg, const float* final_T, const uint32_t* n_contrib, const float* dL_dpix,
                                const uint32_t* order, float* grad_rows, float mean_list, const unsigned long long* blend_log,
                                const uint32_t* quad_hi, hipStream_t s)
{
    if (!gsr_variant("LPT", 1)) order = nullptr;
    const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    int nq = gsr_variant("BWD_NQ", 0);
    // two quadrants per wave pay when the lists are short AND the image has enough tiles to fill the chip with half as many waves
    // (measured at 1080p, ms, NQ = 2 vs 1: 500k Gaussians, mean list 467: 0.374 vs 0.397; 1M, 930: 0.461 vs 0.434; 1.5M, 1400: 0.609 vs 0.521;
    // 2M, 1866: 0.745 vs 0.636; and 10k Gaussians at 256x256, 256 tiles: 0.121 vs 0.064; with the final kernels, mean list ~400: 300k Gaussians at
    // 800x800, 2,500 tiles: 0.339 vs 0.192; 150k at 640x360, 920 tiles: 0.394 vs 0.204 -- few tiles leave the longest list's waves alone on the chip)
    if (nq != 1 && nq != 2) nq = (mean_list < (float)gsr_variant("BWD_NQ_LIST", 700) && gx * gy >= gsr_variant("BWD_NQ_TILES", 4096)) ? 2 : 1;
    if (blend_log) {  // the forward logged its blend decisions (render_forward.hip): the walk reads them instead of re-deriving them
        if (nq == 2)
            hipLaunchKernelGGL((render_backward_log_kernel<2>), dim3(gx * gy), dim3(128), (size_t)gsr_variant("BWD_PADLDS", 0), s, W, H, gx, gy, order, ranges,
                               point_list, splat, bg, final_T, quad_hi, dL_dpix, blend_log, grad_rows);
        else
            hipLaunchKernelGGL((render_backward_log_kernel<1>), dim3(gx * gy), dim3(256), (size_t)gsr_variant("BWD_PADLDS", 0), s, W, H, gx, gy, order, ranges,
                               point_list, splat, bg, final_T, quad_hi, dL_dpix, blend_log, grad_rows);
        return;
    }
    const bool fast = gsr_variant("BWD_FASTEXP", 1) != 0;
#define GSR_BWD_LAUNCH(NQ_, F_)                                                                                                  \
    hipLaunchKernelGGL((render_backward_kernel<NQ_, F_>), dim3(gx * gy), dim3(64 * (4 / NQ_)), (size_t)gsr_variant("BWD_PADLDS", 0), s, W, H, gx, gy, order, ranges, \
                       point_list, splat, bg, final_T, n_contrib, dL_dpix, grad_rows)
    if (nq == 2) { if (fast) GSR_BWD_LAUNCH(2, true); else GSR_BWD_LAUNCH(2, false); }
    else { if (fast) GSR_BWD_LAUNCH(1, true); else GSR_BWD_LAUNCH(1, false); }
#undef GSR_BWD_LAUNCH
}

int gsr_render_backward_ablate() { return GSR_ABLATE; }

#if GSR_LOG_PROFILE == 3
// time-stamp builds only (tools/r5_wave_times.py; not in the header, not in the shipped library): copies and clears the per-wave time stamps
// of the last log-walk launch
extern "C" __attribute__((visibility("default"))) int gsr_debug_wave_times(unsigned long long* out, int words)
{
    if (words > 4 * 32768) words = 4 * 32768;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), (size_t)words * 8) != hipSuccess) return -1;
    static unsigned long long zeros[4 * 32768];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wave_times), zeros, sizeof(zeros)) != hipSuccess) return -1;
    return 0;
}
#endif

// copies and clears the phase profile of the log walk (all zero unless built with -DGSR_LOG_PROFILE=1)
int gsr_blend_profile_fetch(unsigned long long* out8)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_blend_prof), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_blend_prof), z, sizeof(z)) != hipSuccess) return -1;
    return GSR_LOG_PROFILE;
}
