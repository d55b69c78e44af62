// Front-to-back alpha compositing, one 16x16 tile per 256-thread workgroup (4 wave64s, each a 16x4 pixel strip).
// Replaces renderCUDA, cuda_rasterizer/forward.cu:261-374.
//
// Per batch of 256 list entries the workgroup gathers each splat's packed 48-byte record into LDS
// (three 16-byte loads per lane) and every wave then walks the batch reading the entry with wave-uniform
// (broadcast) LDS reads.  A wave leaves the walk as soon as all of its 64 pixels are done, the workgroup
// stops when all four have (the reference's __syncthreads_count early exit, forward.cu:301-304).
//
// Arithmetic contract with the CPU oracle: power / colour accumulation use the explicit fmaf pairs
// below ("CONTRACT-FMA"), exp is gsr_expf_blend -> final_T, n_contrib and out_color are bit-identical.
// The `pthr` test is a pure shortcut: power < ln(1/(255*opacity)) - 1e-3 implies alpha < 1/255 for any
// exp accurate to 1e-6, so skipping the exp there cannot change a decision.
#include "common.h"

namespace {

#ifndef GSR_FWD_GROUP
#define GSR_FWD_GROUP 4  // entries per turn of the walk (the all-pixels-done test runs once per turn)
#endif

template <bool QUAD>
__global__ void __launch_bounds__(256) render_forward_kernel(int W, int H, int gx, int gy, int wave8x8, const uint32_t* __restrict__ order,
                                                             const uint32_t* __restrict__ ranges,
                                                             const uint32_t* __restrict__ point_list,
                                                             const float* __restrict__ splat, const float* __restrict__ bg,
                                                             float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                             float* __restrict__ out_color)
{
    __shared__ float4 s_a[256 + (GSR_FWD_GROUP > 4 ? GSR_FWD_GROUP : 4)];  // mx, my, conic.x, conic.y          (+4: padding entries for the unrolled walk)
    __shared__ float4 s_b[256 + (GSR_FWD_GROUP > 4 ? GSR_FWD_GROUP : 4)];  // conic.z, opacity, pthr, index of the entry in the tile's list (uint bits)
    __shared__ float4 s_c[256];  // r, g, b, Gaussian id (uint bits)
    __shared__ int s_done[4];
    __shared__ int s_wcnt[4];    // entries each wave keeps after the tile-level rejection
    __shared__ unsigned long long s_qm[QUAD ? 4 : 1][4];  // QUAD: per blend wave (quadrant), per staging wave: the staged entries that can reach it

    // 1-D launch in `order` (tiles by descending list length, binning.hip); NULL = row-major
    const uint32_t tile = order ? order[blockIdx.x] : blockIdx.x;
    const uint32_t bx = tile % (uint32_t)gx, by = tile / (uint32_t)gx;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float tile_x0 = (float)(bx * GSR_TILE_X), tile_y0 = (float)(by * GSR_TILE_Y);
    // pixel of this lane: each wave owns an 8x8 block of the tile (wave8x8) or a 16x4 strip
    const uint32_t in_x = wave8x8 ? ((tid >> 6) & 1) * 8 + (tid & 7) : (tid & 15);
    const uint32_t in_y = wave8x8 ? (tid >> 7) * 8 + ((tid >> 3) & 7) : (tid >> 4);
    const uint32_t pix_x = bx * GSR_TILE_X + in_x, pix_y = by * GSR_TILE_Y + in_y;
    const bool inside = pix_x < (uint32_t)W && pix_y < (uint32_t)H;
    const float pixf_x = (float)pix_x, pixf_y = (float)pix_y;
    // tile range is workgroup-uniform: keep it (and every loop bound derived from it) in scalar registers
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(ranges[2 * tile]);
    const uint32_t r1 = __builtin_amdgcn_readfirstlane(ranges[2 * tile + 1]);
    const int total = (int)(r1 - r0);
    const int rounds = (total + 255) / 256;

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last_contributor = 0;
    // `done` is encoded in px: a finished (or outside) pixel is parked at x = PARKED, which drives its `power` to
    // -inf/NaN so the single range test of the hot loop rejects it -- no `done` mask, no EXEC branch per entry.
    constexpr float PARKED = 1e30f;
    float px = inside ? pixf_x : PARKED;
    bool wave_done = !gsr_any(px != PARKED);

    if (QUAD && tid < (GSR_FWD_GROUP > 4 ? GSR_FWD_GROUP : 4)) {  // padding entries that can never pass the range test (the walk takes entries four at a time)
        s_a[256 + tid] = make_float4(-PARKED, -PARKED, 1.f, 0.f);
        s_b[256 + tid] = make_float4(1.f, 0.f, INFINITY, 0.f);
    }
    for (int i = 0; QUAD && i < rounds; i++) {
        // QUAD: every staged entry is tested against the four 8x8 quadrants of the tile (gsr_quad_mask) and stays at its staged
        // position; a wave walks only the entries that can reach ITS block, picking them from the ballot masks with scalar bit
        // scans -- 40 % of the (entry, wave) visits of a walk over the whole tile list blend nothing.
        const int progress = i * 256 + tid;
        uint32_t qm = 0;
        float4 q0, q1, q2;
        uint32_t id = 0;
        if (progress < total) {
            id = point_list[r0 + progress];
            const float4* rec = reinterpret_cast<const float4*>(splat + (size_t)id * GSR_SPLAT_F);
            q0 = rec[0]; q1 = rec[1]; q2 = rec[2];
            qm = gsr_quad_mask(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tile_x0, tile_y0);
            if (wave8x8 == 2 && qm) qm = 15u;  // debugging aid (GSR_QUAD=2): every kept entry visits every wave
        }
        const unsigned long long b0 = __ballot(qm & 1u), b1 = __ballot(qm & 2u), b2 = __ballot(qm & 4u), b3 = __ballot(qm & 8u);
        __syncthreads();  // previous batch fully consumed
        if (lane == 0) { s_done[wave] = wave_done; s_qm[0][wave] = b0; s_qm[1][wave] = b1; s_qm[2][wave] = b2; s_qm[3][wave] = b3; }
        if (qm != 0u) {
            // conservative skip threshold on `power` (see header); opacity <= 0 or NaN never skips; clamped to -80 (see below)
            const float pthr = q1.y > 0.f ? __logf(1.0f / (255.0f * q1.y)) - 1e-3f : -INFINITY;
            s_a[tid] = q0;
            s_b[tid] = make_float4(q1.x, q1.y, fmaxf(pthr, -80.0f), __uint_as_float((uint32_t)progress));
            s_c[tid] = make_float4(q1.z, q1.w, q2.x, __uint_as_float(id));
        }
        __syncthreads();
        // workgroup-wide early exit (forward.cu:301-304)
        if (s_done[0] & s_done[1] & s_done[2] & s_done[3]) break;
        if (wave_done) continue;
        for (int w = 0; w < 4 && !wave_done; w++) {
            unsigned long long m = s_qm[wave][w];
            // (readfirstlane returns int: without the uint32_t casts the low half would be sign-extended over the high half)
            m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(m >> 32)) << 32) |
                (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)m);
            while (m != 0ull) {
                // the next (up to) four entries of this wave; missing ones are the padding entry
                int jj[GSR_FWD_GROUP];
#pragma unroll
                for (int u = 0; u < GSR_FWD_GROUP; u++) {
                    jj[u] = m != 0ull ? 64 * w + (int)__builtin_ctzll(m) : 256 + u;
                    m &= m - 1ull;
                }
#pragma unroll
                for (int u = 0; u < GSR_FWD_GROUP; u++) {
                    const int j = jj[u];
                    const float4 a = s_a[j];
                    const float4 b = s_b[j];
                    const float dx = a.x - px, dy = a.y - pixf_y;
                    // CONTRACT-FMA: power = -0.5*(cx*dx*dx + cz*dy*dy) - cy*dx*dy
                    const float q = __builtin_fmaf(a.z * dx, dx, (b.x * dy) * dy);
                    const float power = __builtin_fmaf(-0.5f, q, -((a.w * dx) * dy));
                    const bool maybe = power >= b.z;
                    if (gsr_any(maybe)) {  // wave-uniform
                        if (maybe && px != PARKED && power <= 0.0f) {
                            const float alpha = fminf(0.99f, b.y * gsr_expf_blend(power));
                            if (alpha >= 1.0f / 255.0f) {
                                const float test_T = T * (1.0f - alpha);
                                if (test_T < 0.0001f) {
                                    px = PARKED;  // done
                                } else {
                                    const float4 c = s_c[j];
                                    // CONTRACT-FMA: C += (f*alpha)*T
                                    C0 = __builtin_fmaf(c.x * alpha, T, C0);
                                    C1 = __builtin_fmaf(c.y * alpha, T, C1);
                                    C2 = __builtin_fmaf(c.z * alpha, T, C2);
                                    T = test_T;
                                    last_contributor = __float_as_uint(b.w) + 1u;  // = `contributor` of forward.cu:322
                                }
                            }
                        }
                    }
                }
                if (!gsr_any(px != PARKED)) { wave_done = true; break; }
            }
        }
    }
    for (int i = 0; !QUAD && i < rounds; i++) {
        // gather this batch's records and decide, per entry, whether the tile can see it at all
        const int progress = i * 256 + tid;
        bool keep = false;
        float4 q0, q1, q2;
        float pthr = 0.f;
        uint32_t id = 0;
        if (progress < total) {
            id = point_list[r0 + progress];
            const float4* rec = reinterpret_cast<const float4*>(splat + (size_t)id * GSR_SPLAT_F);
            q0 = rec[0]; q1 = rec[1]; q2 = rec[2];
            // conservative skip threshold on `power` (see header); opacity <= 0 or NaN never skips
            pthr = q1.y > 0.f ? __logf(1.0f / (255.0f * q1.y)) - 1e-3f : -INFINITY;
            keep = !gsr_tile_cannot_touch(q0.x, q0.y, q0.z, q0.w, q1.x, pthr, tile_x0, tile_y0);
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) { s_done[wave] = wave_done; s_wcnt[wave] = __popcll(bal); }
        __syncthreads();  // previous batch fully consumed; counts and done flags visible
        // workgroup-wide early exit (forward.cu:301-304)
        if (s_done[0] & s_done[1] & s_done[2] & s_done[3]) break;
        int base = 0;
        for (int w = 0; w < wave; w++) base += s_wcnt[w];
        const int kept = __builtin_amdgcn_readfirstlane(s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3]);
        if (keep) {  // order-preserving compaction (the walk must stay front to back)
            const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
            s_a[pos] = q0;
            // walk threshold clamped to -80: below it alpha < op * e^-80, which no opacity turns into 1/255, and the blend's
            // exp then never sees an argument outside [-80, 0]
            s_b[pos] = make_float4(q1.x, q1.y, fmaxf(pthr, -80.0f), __uint_as_float((uint32_t)progress));
            s_c[pos] = make_float4(q1.z, q1.w, q2.x, __uint_as_float(id));
        }
        if (tid < 4) {  // padding entries that can never pass the range test (walk runs in groups of 4, no bounds checks)
            s_a[kept + tid] = make_float4(-PARKED, -PARKED, 1.f, 0.f);
            s_b[kept + tid] = make_float4(1.f, 0.f, INFINITY, 0.f);
        }
        __syncthreads();
        if (!wave_done) {
            for (int j0 = 0; j0 < kept; j0 += 4) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = j0 + u;
                    const float4 a = s_a[j];
                    const float4 b = s_b[j];
                    const float dx = a.x - px, dy = a.y - pixf_y;
                    // CONTRACT-FMA: power = -0.5*(cx*dx*dx + cz*dy*dy) - cy*dx*dy
                    const float q = __builtin_fmaf(a.z * dx, dx, (b.x * dy) * dy);
                    const float power = __builtin_fmaf(-0.5f, q, -((a.w * dx) * dy));
                    // one compare (parked pixels and padding entries give -inf / NaN / pthr = inf and fail it); power > 0 is
                    // left to the exact path
                    const bool maybe = power >= b.z;
                    if (gsr_any(maybe)) {  // wave-uniform; rare relative to the skip
                        if (maybe && px != PARKED && power <= 0.0f) {
                            const float alpha = fminf(0.99f, b.y * gsr_expf_blend(power));
                            if (alpha >= 1.0f / 255.0f) {
                                const float test_T = T * (1.0f - alpha);
                                if (test_T < 0.0001f) {
                                    px = PARKED;  // done
                                } else {
                                    const float4 c = s_c[j];
                                    // CONTRACT-FMA: C += (f*alpha)*T
                                    C0 = __builtin_fmaf(c.x * alpha, T, C0);
                                    C1 = __builtin_fmaf(c.y * alpha, T, C1);
                                    C2 = __builtin_fmaf(c.z * alpha, T, C2);
                                    T = test_T;
                                    last_contributor = __float_as_uint(b.w) + 1u;  // = `contributor` of forward.cu:322
                                }
                            }
                        }
                    }
                }
                if (!gsr_any(px != PARKED)) { wave_done = true; break; }
            }
        }
    }
    if (inside) {
        const size_t pix_id = (size_t)W * pix_y + pix_x;
        const size_t plane = (size_t)H * W;
        final_T[pix_id] = T;
        n_contrib[pix_id] = last_contributor;
        out_color[pix_id] = C0 + T * bg[0];
        out_color[plane + pix_id] = C1 + T * bg[1];
        out_color[2 * plane + pix_id] = C2 + T * bg[2];
    }
}

}  // namespace

void gsr_launch_render_forward(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* splat,
                               const float* bg, float* final_T, uint32_t* n_contrib, float* out_color, const uint32_t* order,
                               hipStream_t s)
{
    if (!gsr_variant("LPT", 1)) order = nullptr;
    const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    const int wave8x8 = gsr_variant("WAVE8X8", 1);
    if (wave8x8 && gsr_variant("QUAD", 1))
        hipLaunchKernelGGL((render_forward_kernel<true>), dim3(gx * gy), dim3(256), (size_t)gsr_variant("FWD_PADLDS", 0), s, W, H, gx, gy, gsr_variant("QUAD", 1) == 2 ? 2 : 1, order, ranges, point_list, splat, bg,
                           final_T, n_contrib, out_color);
    else
        hipLaunchKernelGGL((render_forward_kernel<false>), dim3(gx * gy), dim3(256), 0, s, W, H, gx, gy, wave8x8, order, ranges, point_list, splat, bg,
                           final_T, n_contrib, out_color);
}
