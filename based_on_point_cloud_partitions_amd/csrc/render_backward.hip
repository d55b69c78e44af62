// Backward of the alpha compositing, one 16x16 tile per 256-thread workgroup (4 wave64s, each a 16x4 pixel strip).
// Replaces renderCUDA, cuda_rasterizer/backward.cu:399-557.
//
// The reference issues 9 global float atomicAdds per contributing (pixel, Gaussian) pair
// (backward.cu:523,545-554).  On MI355X global float atomics top out near 1.3 TB/s chip-wide and collapse
// when lanes scatter, so this kernel never uses them: the 9 partial sums of a list entry are reduced across
// the 64 pixels of a wave with DPP adds (fixed tree), the four waves' results are combined through LDS in
// wave order, and every list entry writes ONE 48-byte gradient row with plain stores.  Rows are addressed
// by the Gaussian's emission slot (exclusive instance offset + position of this tile in its rectangle), so
// the rows of one Gaussian are contiguous and the per-Gaussian kernel sums them in a fixed order:
// gradients are bitwise reproducible run to run, unlike the reference's.
//
// What is reduced are MOMENTS of w = G*dL/dalpha (S w, S w*dx, S w*dy, S w*dx^2, S w*dx*dy, S w*dy^2) and
// S alpha*T*dL/dpixel; the per-Gaussian factors of backward.cu:538-554 (opacity, conic, 0.5*W, 0.5*H) are applied
// once per list entry after the reduction instead of once per pixel.
//
// Traversal is back to front from the tile's largest n_contrib (entries behind it contribute to no pixel,
// backward.cu:480-482); alpha / G are recomputed so that the set of blended pairs is exactly the forward's (see FAST_EXP).
// The kernel is VALU-issue bound; see DESIGN.md §4 for the instruction budget.
#include "common.h"

namespace {

// Sums 9 registers over the 64 lanes of a wave; totals are valid in lane 63.  One v_add_f32_dpp per value and level,
// written as ONE asm block in level-major order: every DPP read is 8 instructions behind the write it depends on, so
// no wait states are needed inside (hipcc pads nothing in asm), and the compiler cannot SLP-pack the adds into
// v_pk_add_f32 + v_mov_dpp pairs (which is what it emits for the builtin form: 2.5x the instructions).
// Must be executed with all 64 lanes active (it is only called under a wave-uniform condition).
#define GSR_DPP_LEVEL(ctrl)                    \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n"      \
    "v_add_f32_dpp %1, %1, %1 " ctrl "\n"      \
    "v_add_f32_dpp %2, %2, %2 " ctrl "\n"      \
    "v_add_f32_dpp %3, %3, %3 " ctrl "\n"      \
    "v_add_f32_dpp %4, %4, %4 " ctrl "\n"      \
    "v_add_f32_dpp %5, %5, %5 " ctrl "\n"      \
    "v_add_f32_dpp %6, %6, %6 " ctrl "\n"      \
    "v_add_f32_dpp %7, %7, %7 " ctrl "\n"      \
    "v_add_f32_dpp %8, %8, %8 " ctrl "\n"
__device__ __forceinline__ void wave_sum9_to_lane63(float& v0, float& v1, float& v2, float& v3, float& v4, float& v5, float& v6,
                                                    float& v7, float& v8)
{
    asm volatile("s_nop 1\n"
                 GSR_DPP_LEVEL("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 GSR_DPP_LEVEL("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 GSR_DPP_LEVEL("row_half_mirror row_mask:0xf bank_mask:0xf")
                 GSR_DPP_LEVEL("row_mirror row_mask:0xf bank_mask:0xf")
                 GSR_DPP_LEVEL("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 GSR_DPP_LEVEL("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8));
}

// Reduce-scatter form of the same sum: after each exchange a lane keeps only half of the values, so 8 values cost
// 12 + 6 + 4 + 1 instructions for the four in-row levels instead of 32; the 9th value is reduced along its row
// (4 DPP adds) and parked in lane 8 of each row, and the two cross-row levels are shared by all nine.
// Result: lane l < 8 holds the wave total of value (4*(l&1) + 2*((l>>1)&1) + ((l>>2)&1)), lane 8 holds the total of v8.
// m0/m1/m2/m8: lane masks (l&1), (l&2), (l&4), ((l&15)==8).  All 64 lanes must be active.
__device__ __forceinline__ float wave_sum9_scatter(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                                   float v8, unsigned long long m0, unsigned long long m1,
                                                   unsigned long long m2, unsigned long long m8)
{
    float t0, t1, t2, t3;
    asm volatile(
        // level 1: partner lane^1; lanes with bit0 keep v4..v7, the others v0..v3
        "v_cndmask_b32 %9, %4, %0, %13\n"   // t = what the partner needs from me
        "v_cndmask_b32 %10, %5, %1, %13\n"
        "v_cndmask_b32 %11, %6, %2, %13\n"
        "v_cndmask_b32 %12, %7, %3, %13\n"
        "v_cndmask_b32 %0, %0, %4, %13\n"   // v = what I keep
        "v_cndmask_b32 %1, %1, %5, %13\n"
        "v_cndmask_b32 %2, %2, %6, %13\n"
        "v_cndmask_b32 %3, %3, %7, %13\n"
        "v_add_f32_dpp %0, %9, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %1, %10, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %2, %11, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %3, %12, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        // 9th value, row levels interleaved as independent work between dependent DPP steps
        "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        // level 2: partner lane^2; lanes with bit1 keep slots 2,3
        "v_cndmask_b32 %9, %2, %0, %14\n"
        "v_cndmask_b32 %10, %3, %1, %14\n"
        "v_cndmask_b32 %0, %0, %2, %14\n"
        "v_cndmask_b32 %1, %1, %3, %14\n"
        "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %0, %9, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %1, %10, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %8, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        // level 3: partner lane^4; lanes with bit2 keep slot 1
        "v_cndmask_b32 %9, %1, %0, %15\n"
        "v_cndmask_b32 %0, %0, %1, %15\n"
        "v_add_f32_dpp %8, %8, %8 row_mirror row_mask:0xf bank_mask:0xf\n"
        "s_nop 0\n"
        "v_add_f32_dpp %0, %9, %0 row_shr:4 row_mask:0xf bank_mask:0xa\n"
        "v_add_f32_dpp %0, %9, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n"
        "s_nop 1\n"
        // level 4: partner lane^8
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
        // park the 9th value's row total in lane 8 of each row
        "v_cndmask_b32 %0, %0, %8, %16\n"
        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "=&v"(t0), "=&v"(t1),
          "=&v"(t2), "=&v"(t3)
        : "s"(m0), "s"(m1), "s"(m2), "s"(m8));
    // levels 5, 6: lane-wise across the four rows with gfx950's v_permlane16_swap / v_permlane32_swap (two VALU
    // instructions per level; __shfl_xor would take two dependent trips through the LDS crossbar)
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

constexpr int NV = 9;  // partial sums per list entry

template <int BATCH, bool RS, bool FAST_EXP, bool QUAD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) render_backward_kernel(int W, int H, int gx, int gy, int wave8x8, const uint32_t* __restrict__ order,
                                                              const uint32_t* __restrict__ ranges,
                                                              const uint32_t* __restrict__ point_list,
                                                              const float* __restrict__ splat, const float* __restrict__ bg,
                                                              const float* __restrict__ final_Ts,
                                                              const uint32_t* __restrict__ n_contrib,
                                                              const float* __restrict__ dL_dpixels, float* __restrict__ grad_rows)
{
    constexpr int ACC_STRIDE = BATCH * NV;  // floats per wave
    __shared__ float4 s_a[BATCH];  // mx, my, conic.x, conic.y
    __shared__ float4 s_b[BATCH];  // conic.z, opacity, pthr, index of the entry in the tile's list (uint bits)
    __shared__ float4 s_c[BATCH];  // r, g, b, gradient row index (uint bits)
    __shared__ int s_wcnt[4];      // entries each wave keeps after the tile-level rejection
    // QUAD: per blend wave (quadrant), per 64 staged entries: the entries that can reach the wave's 8x8 block (gsr_quad_mask);
    // entries stay at their staged position and a wave picks its own with scalar bit scans (as in the forward)
    __shared__ unsigned long long s_qm[QUAD ? 4 : 1][(BATCH + 63) / 64];
    __shared__ __attribute__((aligned(16))) float s_acc[4 * ACC_STRIDE];
    __shared__ uint32_t s_max[4];

    // 1-D launch in `order` (tiles by descending list length, binning.hip); NULL = row-major
    const uint32_t tile = order ? order[blockIdx.x] : blockIdx.x;
    const uint32_t bx = tile % (uint32_t)gx, by = tile / (uint32_t)gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pixel of this lane: each wave owns an 8x8 block of the tile (wave8x8) or a 16x4 strip
    const uint32_t in_x = wave8x8 ? ((tid >> 6) & 1) * 8 + (tid & 7) : (tid & 15);
    const uint32_t in_y = wave8x8 ? (tid >> 7) * 8 + ((tid >> 3) & 7) : (tid >> 4);
    const uint32_t pix_x = bx * GSR_TILE_X + in_x, pix_y = by * GSR_TILE_Y + in_y;
    const bool inside = pix_x < (uint32_t)W && pix_y < (uint32_t)H;
    const size_t pix_id = (size_t)W * pix_y + pix_x;
    const size_t plane = (size_t)H * W;
    const float pixf_x = (float)pix_x, pixf_y = (float)pix_y;
    const float tile_x0 = (float)(bx * GSR_TILE_X), tile_y0 = (float)(by * GSR_TILE_Y);
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(ranges[2 * tile]);
    const uint32_t r1 = __builtin_amdgcn_readfirstlane(ranges[2 * tile + 1]);
    const uint32_t total = r1 - r0;
    if (total == 0) return;

    const float T_final = inside ? final_Ts[pix_id] : 0.f;
    float T = T_final;
    const uint32_t last_contributor = inside ? n_contrib[pix_id] : 0u;
    float dL0 = 0.f, dL1 = 0.f, dL2 = 0.f;
    if (inside) { dL0 = dL_dpixels[pix_id]; dL1 = dL_dpixels[plane + pix_id]; dL2 = dL_dpixels[2 * plane + pix_id]; }
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const float bg_dot = ((0.f + bg0 * dL0) + bg1 * dL1) + bg2 * dL2;
    const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

    // largest n_contrib of the tile = number of list entries that matter to any pixel
    uint32_t m = last_contributor;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
    if (lane == 0) s_max[wave] = m;
    __syncthreads();
    const uint32_t hi = __builtin_amdgcn_readfirstlane(max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));

    float accum0 = 0.f, accum1 = 0.f, accum2 = 0.f;
    float* my_acc = s_acc + wave * ACC_STRIDE;
    const int rounds = (int)((hi + BATCH - 1) / BATCH);
    // reduce-scatter constants: lane masks and the accumulator slot this lane writes
    const unsigned long long M0 = 0xAAAAAAAAAAAAAAAAull, M1 = 0xCCCCCCCCCCCCCCCCull, M2 = 0xF0F0F0F0F0F0F0F0ull,
                             M8 = 0x0100010001000100ull;
    const int my_slot = lane < 8 ? (4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1)) : 8;

    for (int i = 0; i < rounds; i++) {
        // gather this batch's records (back to front); instances whose tile the splat cannot reach own no gradient row
        // (gsr_reach_mask, preprocess) and are dropped from the walk
        const uint32_t progress = (uint32_t)i * BATCH + tid;
        bool keep = false;
        float4 q0, q1, q2;
        float pthr = 0.f;
        uint32_t k = 0, row = 0;
        if (tid < BATCH && progress < hi) {
            k = hi - 1 - progress;
            const uint32_t id = point_list[r0 + k];
            const float4* rec = reinterpret_cast<const float4*>(splat + (size_t)id * GSR_SPLAT_F);
            q0 = rec[0]; q1 = rec[1]; q2 = rec[2];
            const float4 q3 = rec[3];  // gradient-row bookkeeping: first row, reachable tiles of the rectangle (common.h)
            uint32_t mnx, mny, mxx, mxy;
            gsr_get_rect(q0.x, q0.y, __float_as_int(q2.z), gx, gy, mnx, mny, mxx, mxy);
            const uint32_t tilepos = (by - mny) * (mxx - mnx) + (bx - mnx);
            pthr = q1.y > 0.f ? __logf(1.0f / (255.0f * q1.y)) - 1e-3f : -INFINITY;
            if (__float_as_uint(q3.w) == 0u) {
                // the preprocess kernel has applied gsr_reach_mask to the rectangle: an unreachable instance owns no row and is
                // dropped here; a reachable one is numbered by its rank among the reachable tiles
                const unsigned long long mask = ((unsigned long long)__float_as_uint(q3.z) << 32) | __float_as_uint(q3.y);
                keep = (mask >> tilepos) & 1ull;
                row = __float_as_uint(q3.x) + (uint32_t)__popcll(mask & ((1ull << tilepos) - 1ull));
            } else {  // rectangle of more than 64 tiles: every instance owns a row; unreachable ones get their zero row right here
                row = __float_as_uint(q3.x) + tilepos;
                keep = !gsr_tile_cannot_touch(q0.x, q0.y, q0.z, q0.w, q1.x, pthr, tile_x0, tile_y0);
                if (!keep) {
                    float4* o = reinterpret_cast<float4*>(grad_rows + (size_t)row * GSR_GROW_F);
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    o[0] = z; o[1] = z; o[2] = z;
                }
            }
        }
        int nb = 0;
        if constexpr (QUAD) {
            uint32_t qm = 0;
            if (keep) {
                qm = gsr_quad_mask(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tile_x0, tile_y0);
                if (qm == 0u) {  // owns a row, but no block of the tile can reach it (the two tests round differently): zero row
                    float4* o = reinterpret_cast<float4*>(grad_rows + (size_t)row * GSR_GROW_F);
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    o[0] = z; o[1] = z; o[2] = z;
                }
            }
            const unsigned long long b0 = __ballot(qm & 1u), b1 = __ballot(qm & 2u), b2 = __ballot(qm & 4u), b3 = __ballot(qm & 8u);
            __syncthreads();  // previous batch fully flushed
            if (lane == 0 && wave < (BATCH + 63) / 64) { s_qm[0][wave] = b0; s_qm[1][wave] = b1; s_qm[2][wave] = b2; s_qm[3][wave] = b3; }
            if (tid < BATCH) {
                if (qm != 0u) {
                    s_a[tid] = q0;
                    s_b[tid] = make_float4(q1.x, q1.y, fmaxf(pthr, -80.0f), __uint_as_float(k));  // clamp: see the forward
                }
                s_c[tid] = make_float4(q1.z, q1.w, q2.x, __uint_as_float(qm != 0u ? row : 0xFFFFFFFFu));  // 0xFFFFFFFF: nothing to flush
            }
        } else {
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wcnt[wave] = __popcll(bal);
        __syncthreads();  // previous batch fully flushed; counts visible
        int base = 0;
        for (int w = 0; w < wave; w++) base += s_wcnt[w];
        nb = __builtin_amdgcn_readfirstlane(s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3]);
        if (keep) {  // order-preserving compaction
            const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
            s_a[pos] = q0;
            s_b[pos] = make_float4(q1.x, q1.y, fmaxf(pthr, -80.0f), __uint_as_float(k));  // clamp: see the forward
            s_c[pos] = make_float4(q1.z, q1.w, q2.x, __uint_as_float(row));
        }
        }
        {   // zero this wave's accumulators
            float4* z = reinterpret_cast<float4*>(my_acc);
#pragma unroll
            for (int t = 0; t < (ACC_STRIDE / 4 + 63) / 64; t++)
                if (t * 64 + lane < ACC_STRIDE / 4) z[t * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        auto walk_entry = [&](const int j) {
            const float4 a = s_a[j];
            const float4 b = s_b[j];
            const uint32_t kidx = __float_as_uint(b.w);  // position of this entry in the tile's list
            // entries behind this pixel's last contributor never reach it (backward.cu:480-482): their lanes are masked out of
            // every lane mask below (no per-lane select on the pixel position -- a select costs 4.4 cycles, a scalar AND nothing)
            const unsigned long long m_act = __builtin_amdgcn_ballot_w64(kidx < last_contributor);
            const float dx = a.x - pixf_x, dy = a.y - pixf_y;
            // identical to the forward (CONTRACT-FMA)
            const float q = __builtin_fmaf(a.z * dx, dx, (b.x * dy) * dy);
            const float power = __builtin_fmaf(-0.5f, q, -((a.w * dx) * dy));
            // Lane predicates are kept as 64-bit lane masks in scalar registers (each ballot of a plain compare IS the
            // compare's result mask): combining them and testing "any lane" is scalar work, and the two selects below take
            // the mask directly.  Written with bools, the compiler re-materialises every combined predicate through
            // v_cndmask + v_cmp_ne (two half-rate VALU instructions per test).
            const unsigned long long m_maybe = m_act & __builtin_amdgcn_ballot_w64(power >= b.z);  // power > 0 is tested below
            if (m_maybe == 0ull) return;  // wave-uniform
            unsigned long long m_c = m_maybe & __builtin_amdgcn_ballot_w64(power <= 0.0f);
            // G = exp(power) from the hardware exp2 (v_exp_f32, <= 1e-6 relative off gsr_expf on the range that can pass),
            // 2 issue slots instead of the 17 instructions of the forward's fully specified gsr_expf.  The alpha >= 1/255
            // decision must be the forward's: when any lane's alpha lies within 8e-6 (relative) of the threshold the wave
            // redoes this entry with gsr_expf, so the set of blended pairs is exactly the forward's.
            float G = FAST_EXP ? __builtin_amdgcn_exp2f(power * 1.44269504088896341f) : gsr_expf_inrange(power);
            float alpha = fminf(0.99f, b.y * G);
            if (FAST_EXP && (m_c & __builtin_amdgcn_ballot_w64(fabsf(alpha - 1.0f / 255.0f) < 3e-8f)) != 0ull) {
                G = gsr_expf_inrange(power);
                alpha = fminf(0.99f, b.y * G);
            }
            m_c &= __builtin_amdgcn_ballot_w64(alpha >= 1.0f / 255.0f);
            if (m_c == 0ull) return;  // wave-uniform
            // From here all 64 lanes run the same straight-line code.  A lane that does not contribute is given alpha = G = 0:
            // then T/(1-0) = T, its gradient terms vanish, and the colour recurrence (backward.cu:515) leaves accum unchanged
            // (0 * c + 1 * accum) -- so no per-variable select is needed.
            float Ge, ae;
            asm("v_cndmask_b32 %0, 0, %2, %4\n v_cndmask_b32 %1, 0, %3, %4" : "=&v"(Ge), "=v"(ae) : "v"(G), "v"(alpha), "s"(m_c));
            const float4 c = s_c[j];
            const float oma = 1.f - ae;
            const float inv = __builtin_amdgcn_rcpf(oma);
            T = T * inv;  // T / (1 - alpha), backward.cu:502
            // accum0..2 hold the colour accumulated behind this entry (backward.cu:515 already applied for the previous one)
            float dL_dalpha = (c.x - accum0) * dL0;
            dL_dalpha = __builtin_fmaf(c.y - accum1, dL1, dL_dalpha);
            dL_dalpha = __builtin_fmaf(c.z - accum2, dL2, dL_dalpha);
            // ... and the recurrence for the next one right away, same operands and order as the reference evaluates it there:
            // accum = last_alpha * last_color + (1 - last_alpha) * accum (no last_alpha / last_color registers to carry)
            accum0 = __builtin_fmaf(ae, c.x, oma * accum0);
            accum1 = __builtin_fmaf(ae, c.y, oma * accum1);
            accum2 = __builtin_fmaf(ae, c.z, oma * accum2);
            dL_dalpha *= T;
            dL_dalpha = __builtin_fmaf(-T_final * inv, bg_dot, dL_dalpha);
            const float w = Ge * dL_dalpha;
            const float aT = ae * T;
            const float wdx = w * dx, wdy = w * dy;
            float v0 = w, v1 = wdx, v2 = wdy, v3 = wdx * dx, v4 = wdx * dy, v5 = wdy * dy, v6 = aT * dL0, v7 = aT * dL1, v8 = aT * dL2;
            if constexpr (RS) {
                const float tot = wave_sum9_scatter(v0, v1, v2, v3, v4, v5, v6, v7, v8, M0, M1, M2, M8);
                if (lane < 9) my_acc[j * NV + my_slot] = tot;
            } else {
                wave_sum9_to_lane63(v0, v1, v2, v3, v4, v5, v6, v7, v8);
                if (lane == 63) {
                    float* o = my_acc + j * NV;
                    o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3; o[4] = v4; o[5] = v5; o[6] = v6; o[7] = v7; o[8] = v8;
                }
            }
        };
        if constexpr (QUAD) {
#pragma unroll
            for (int w = 0; w < (BATCH + 63) / 64; w++) {
                unsigned long long m = s_qm[wave][w];
                m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(m >> 32)) << 32) |
                    (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)m);
                while (m != 0ull) {
                    walk_entry(64 * w + (int)__builtin_ctzll(m));
                    m &= m - 1ull;
                }
            }
        } else {
#pragma unroll 2
            for (int j = 0; j < nb; j++) walk_entry(j);
        }
        __syncthreads();
        // flush: entry `tid` of the batch -> one 48-byte row, waves summed in fixed order
        if (QUAD ? (tid < BATCH && __float_as_uint(s_c[tid].w) != 0xFFFFFFFFu) : (tid < nb)) {
            float r[NV];
#pragma unroll
            for (int t = 0; t < NV; t++) {
                r[t] = ((s_acc[tid * NV + t] + s_acc[ACC_STRIDE + tid * NV + t]) + s_acc[2 * ACC_STRIDE + tid * NV + t]) +
                       s_acc[3 * ACC_STRIDE + tid * NV + t];
            }
            // per-Gaussian factors of backward.cu:538-554 applied once to the reduced moments
            const float4 a = s_a[tid];
            const float4 b = s_b[tid];
            const float op = b.y, cx = a.z, cy = a.w, cz = b.x;
            const float gmx = op * (-cx * r[1] - cy * r[2]) * ddelx_dx;
            const float gmy = op * (-cz * r[2] - cy * r[1]) * ddely_dy;
            const float hop = -0.5f * op;
            const uint32_t row = __float_as_uint(s_c[tid].w);
            float4* o = reinterpret_cast<float4*>(grad_rows + (size_t)row * GSR_GROW_F);
            o[0] = make_float4(gmx, gmy, hop * r[3], hop * r[4]);
            o[1] = make_float4(hop * r[5], r[0], r[6], r[7]);
            o[2] = make_float4(r[8], 0.f, 0.f, 0.f);
        }
    }
    // list entries behind every pixel's last contributor: zero rows (for those that own a row)
    for (uint32_t e = hi + tid; e < total; e += 256) {
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
        float4* o = reinterpret_cast<float4*>(grad_rows + (size_t)row * GSR_GROW_F);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        o[0] = z; o[1] = z; o[2] = z;
    }
}

}  // namespace

void gsr_launch_render_backward(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* splat,
                                const float* bg, const float* final_T, const uint32_t* n_contrib, const float* dL_dpix,
                                const uint32_t* order, float* grad_rows, hipStream_t s)
{
    if (!gsr_variant("LPT", 1)) order = nullptr;
    const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    const int batch = gsr_variant("BWD_BATCH", 128), rs = gsr_variant("BWD_RS", 1), wave8x8 = gsr_variant("WAVE8X8", 1);
#define GSR_BWD_LAUNCH(B, R_, F_, Q_)                                                                                           \
    hipLaunchKernelGGL((render_backward_kernel<B, R_, F_, Q_>), dim3(gx * gy), dim3(256), 0, s, W, H, gx, gy, wave8x8, order, ranges, point_list, splat, \
                       bg, final_T, n_contrib, dL_dpix, grad_rows)
    const int fast = gsr_variant("BWD_FASTEXP", 1);
    const bool quad = wave8x8 && gsr_variant("QUAD", 1);
    if (batch == 256) GSR_BWD_LAUNCH(256, true, true, false);
    else if (batch == 64) GSR_BWD_LAUNCH(64, true, true, false);
    else if (!rs) GSR_BWD_LAUNCH(128, false, false, false);
    else if (fast && quad) GSR_BWD_LAUNCH(128, true, true, true);
    else if (fast) GSR_BWD_LAUNCH(128, true, true, false);
    else GSR_BWD_LAUNCH(128, true, false, false);
#undef GSR_BWD_LAUNCH
}
