// Front-to-back alpha compositing, one 16x16 tile per 256-thread workgroup (4 wave64s, each an 8x8 quadrant of the tile; 16x4 strips with
// GSR_WAVE8X8=0).
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

// Timing ablations of the blend log (results WRONG on purpose, never defined in the shipped build): -DGSR_FWD_LOG_ABLATE=n
//   1: no window stores (the masks are collected, not written)   2: no v_writelane (zeros are written)
#ifndef GSR_FWD_LOG_ABLATE
#define GSR_FWD_LOG_ABLATE 0
#endif
// Phase profile (make EXTRA=-DGSR_LOG_PROFILE=1, never in the shipped build): wave-cycles per phase of the quadrant walk, summed over all waves:
// [0] whole wave, [1] staging loads + reach tests, [2] barriers + LDS writes, [3] walk (+ window stores), [4] prologue + epilogue
#ifndef GSR_LOG_PROFILE
#define GSR_LOG_PROFILE 0
#endif
__device__ unsigned long long g_fwd_prof[8];
#define GSR_FPROF_MARK(slot) do { if (GSR_LOG_PROFILE) { const long long t_ = clock64(); fprof[slot] += (unsigned long long)(t_ - ft_last); ft_last = t_; } } while (0)
#ifndef GSR_FWD_GROUP
#define GSR_FWD_GROUP 4  // entries per turn of the walk (the all-pixels-done test runs once per turn)
#endif

// LOG (QUAD only): the blend log.  Every (list entry, 8x8 quadrant) visit of the walk that blends leaves the 64-bit mask of the lanes
// (pixels) that blended the entry -- the outcome of every decision of forward.cu:332-352 for 64 pixels -- at
// blend_log[gsr_blend_log_slot(list position, quadrant)].  The backward (render_backward.hip, log walk) reads the masks instead of re-deriving
// them: no range test, no threshold test, no alpha test, no n_contrib bookkeeping, no reach test per quadrant, and the decisions are the
// forward's by construction.
// How the masks get to memory (round 5, designs measured: profiles/README.md): a mask is a ballot, i.e. it lives in scalar registers.
// Scalar stores (s_store_dwordx2) work on gfx950 and cost no VALU slot, but one per visit made the forward 18 % slower (the scalar cache
// takes ~4 bytes per clock per CU, every store counts on lgkmcnt like the LDS operand reads, and its data registers must stay untouched
// until it completes).  So a wave keeps the masks of the 64-entry window it is walking in a VGPR pair -- lane j holds the mask of the
// window's entry j, written with v_writelane_b32 when a visit blends -- and stores the pair once per window, one coalesced 8-byte store
// per lane; entries it did not visit, or that blended nowhere, read as zero.  Windows the wave never walks (behind its exit) are never
// stored: the backward knows the exit from the quadrant's largest n_contrib, which the wave leaves in quad_hi[4 tile + quadrant].
template <bool QUAD, bool LOG>
__global__ void __launch_bounds__(256) render_forward_kernel(int W, int H, int gx, int gy, int wave8x8, const uint32_t* __restrict__ order,
                                                             const uint32_t* __restrict__ ranges,
                                                             const uint32_t* __restrict__ point_list,
                                                             const float* __restrict__ splat, const float* __restrict__ bg,
                                                             float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                             float* __restrict__ out_color, unsigned long long* blend_log,
                                                             uint32_t* __restrict__ quad_hi)
{
    // One 48-byte record per staged entry (round 5; three separate arrays before): a = mx, my, conic.x, conic.y; b = conic.z, opacity, pthr,
    // index of the entry in the tile's list (uint bits); c = r, g, b, Gaussian id (uint bits).  The walk reads an entry with wave-uniform
    // addresses: with one stride, ONE address register serves a, the two halves of b and c through the instructions' offsets -- the slow
    // path re-made two addresses (v_mov from an SGPR: 1.75 ns each) per visit.  (+4: padding entries for the unrolled walk)
    __shared__ float4 s_rec[3 * (256 + (GSR_FWD_GROUP > 4 ? GSR_FWD_GROUP : 4))];
#define s_a(j) s_rec[3 * (j)]
#define s_b(j) s_rec[3 * (j) + 1]
#define s_c(j) s_rec[3 * (j) + 2]
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

    unsigned long long fprof[5] = {0, 0, 0, 0, 0};
    long long ft_last = GSR_LOG_PROFILE ? clock64() : 0;
    const long long ft_begin = ft_last;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last_contributor = 0;
    // `done` is encoded in px: a finished (or outside) pixel is parked at x = PARKED, which drives its `power` to
    // -inf/NaN so the single range test of the hot loop rejects it -- no `done` mask, no EXEC branch per entry.
    constexpr float PARKED = 1e30f;
    float px = inside ? pixf_x : PARKED;
    bool wave_done = !gsr_any(px != PARKED);

    if (QUAD && tid < (GSR_FWD_GROUP > 4 ? GSR_FWD_GROUP : 4)) {  // padding entries that can never pass the range test (the walk takes entries four at a time)
        s_a(256 + tid) = make_float4(-PARKED, -PARKED, 1.f, 0.f);
        s_b(256 + tid) = make_float4(1.f, 0.f, INFINITY, 0.f);
    }
    // LOG: the masks of the window being walked (lane j: entry j of the window) and the slot of (list position `lane` of the tile, this
    // wave's quadrant): the window starting at list position p stores at log_lane + 4 p
    const uint32_t wave_s = __builtin_amdgcn_readfirstlane((uint32_t)wave);
    uint32_t log_lo = 0u, log_hi = 0u;
    unsigned long long* log_lane = LOG ? blend_log + gsr_blend_log_slot(r0 + (uint32_t)lane, wave_s) : nullptr;
    GSR_FPROF_MARK(4);
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
        GSR_FPROF_MARK(1);
        __syncthreads();  // previous batch fully consumed
        if (lane == 0) { s_done[wave] = wave_done; s_qm[0][wave] = b0; s_qm[1][wave] = b1; s_qm[2][wave] = b2; s_qm[3][wave] = b3; }
        if (qm != 0u) {
            // conservative skip threshold on `power` (see header); opacity <= 0 or NaN never skips; clamped to -80 (see below)
            const float pthr = q1.y > 0.f ? __logf(1.0f / (255.0f * q1.y)) - 1e-3f : -INFINITY;
            s_a(tid) = q0;
            // (.w = the entry's index in the list + 1, i.e. `contributor` of forward.cu:322 itself: the blend copies it)
            s_b(tid) = make_float4(q1.x, q1.y, fmaxf(pthr, -80.0f), __uint_as_float((uint32_t)progress + 1u));
            s_c(tid) = make_float4(q1.z, q1.w, q2.x, __uint_as_float(id));
        }
        __syncthreads();
        GSR_FPROF_MARK(2);
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
                    const bool real = m != 0ull;
                    jj[u] = real ? 64 * w + (int)__builtin_ctzll(m) : 256 + u;
                    m &= m - 1ull;
                }
#pragma unroll
                for (int u = 0; u < GSR_FWD_GROUP; u++) {
                    const int j = jj[u];
                    // the entry's record through ONE address register for the whole visit: the byte offset is moved into a VGPR by an asm
                    // the compiler cannot see through -- a wave-uniform address it would rather re-make from the scalar register (a second
                    // v_mov, 1.75 ns) in the slow path than keep
                    uint32_t rec_off;
                    asm("v_mov_b32 %0, %1" : "=v"(rec_off) : "s"((uint32_t)j * 48u));
                    const float4* rec_j = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_rec) + rec_off);
                    const float4 a = rec_j[0];
                    const float4 b = rec_j[1];
                    const float dx = a.x - px, dy = a.y - pixf_y;
                    // CONTRACT-FMA: power = -0.5*(cx*dx*dx + cz*dy*dy) - cy*dx*dy
                    const float q = __builtin_fmaf(a.z * dx, dx, (b.x * dy) * dy);
                    const float power = __builtin_fmaf(-0.5f, q, -((a.w * dx) * dy));
                    const bool maybe = power >= b.z;
                    {
                        // The decisions of forward.cu:332-352 as 64-bit LANE MASKS in scalar registers (round 5; nested per-lane branches
                        // before: three exec save / restore pairs and a second address per visit) -- each compare is its own
                        // ballot, the conjunctions are scalar ANDs -- so the mask of the lanes that blend is there (for the log, LOG) without
                        // a VALU slot, and the blend itself runs under EXEC = that mask (one asm block: the compiler cannot express "these
                        // lanes" from a scalar mask without rebuilding it per lane).  alpha / test_T are evaluated by all lanes; a lane
                        // outside the masks keeps nothing of them.  Same arithmetic per blending lane as the nested form: bit-identical.
                        // (The block leaves EXEC = all ones: every lane of the 256-thread workgroup is active here -- the walk is under
                        // wave-uniform control flow from the kernel's entry, finished and outside pixels are parked, not masked off.)
                        unsigned long long blended = 0ull;
                        const unsigned long long m_maybe = __builtin_amdgcn_ballot_w64(maybe);
                        if (m_maybe != 0ull) {  // wave-uniform
                            // (no `px != PARKED` term: a parked lane has dx = -1e30, so conic.x dx^2 -- conic.x > 0 for every listed splat -- is +inf
                            // and power -inf or NaN: it fails `maybe`, which is what parking a pixel means; the compare was a ninth of the slow path's
                            // non-fp32 instructions)
                            const unsigned long long m1 = m_maybe & __builtin_amdgcn_ballot_w64(power <= 0.0f);
                            const float alpha = fminf(0.99f, b.y * gsr_expf_blend(power));
                            const unsigned long long m2 = m1 & __builtin_amdgcn_ballot_w64(alpha >= 1.0f / 255.0f);
                            const float test_T = T * (1.0f - alpha);
                            const unsigned long long m_done = m2 & __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
                            blended = m2 & ~m_done;
                            const float4 c = rec_j[2];
                            float t0, t1, t2;
                            asm volatile(
                                "v_cndmask_b32 %[px], %[px], %[parked], %[md]\n"   // done: parked
                                "s_mov_b64 exec, %[mb]\n"
                                "v_mul_f32 %[t0], %[cx], %[al]\n"                  // CONTRACT-FMA: C += (f*alpha)*T
                                "v_mul_f32 %[t1], %[cy], %[al]\n"
                                "v_mul_f32 %[t2], %[cz], %[al]\n"
                                "v_fmac_f32 %[C0], %[t0], %[T]\n"
                                "v_fmac_f32 %[C1], %[t1], %[T]\n"
                                "v_fmac_f32 %[C2], %[t2], %[T]\n"
                                "v_mov_b32 %[T], %[tT]\n"
                                "v_mov_b32 %[lc], %[bw]\n"                         // = `contributor` of forward.cu:322 (staged as index + 1)
                                "s_mov_b64 exec, -1\n"
                                : [px] "+v"(px), [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [T] "+v"(T), [lc] "+v"(last_contributor),
                                  [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
                                : [parked] "v"(PARKED), [md] "s"(m_done), [mb] "s"(blended), [cx] "v"(c.x), [cy] "v"(c.y), [cz] "v"(c.z),
                                  [al] "v"(alpha), [tT] "v"(test_T), [bw] "v"(b.w));
                            // the entry's mask into lane (entry's index in the window) of the window's pair.  (A padding entry of the
                            // unrolled walk never gets here: its threshold is +inf.)
                            if (LOG && GSR_FWD_LOG_ABLATE != 2)
                                // (M0 is on the clobber list although the compiler warns that it "may not preserve" reserved registers: on gfx9 it
                                // loads M0 right in front of each instruction that reads it and keeps nothing there across other code)
                                asm volatile("s_mov_b32 m0, %4\n s_nop 0\n v_writelane_b32 %0, %2, m0\n v_writelane_b32 %1, %3, m0"  // (one SGPR + m0 per VALU)
                                             : "+v"(log_lo), "+v"(log_hi)
                                             : "s"((uint32_t)blended), "s"((uint32_t)(blended >> 32)), "s"((uint32_t)j & 63u)
                                             : "m0");
                        }
                    }
                }
                if (!gsr_any(px != PARKED)) { wave_done = true; break; }
            }
            if (LOG && GSR_FWD_LOG_ABLATE != 1) {
                // the window's masks (whatever the wave walked of it): list positions 256 i + 64 w + lane, as far as the list goes
                const uint32_t p0 = (uint32_t)(256 * i + 64 * w);
                if (p0 + (uint32_t)lane < (uint32_t)total) log_lane[4 * (size_t)p0] = ((unsigned long long)log_hi << 32) | log_lo;
                log_lo = 0u; log_hi = 0u;
            }
        }
        GSR_FPROF_MARK(3);
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
            s_a(pos) = q0;
            // walk threshold clamped to -80: below it alpha < op * e^-80, which no opacity turns into 1/255, and the blend's
            // exp then never sees an argument outside [-80, 0]
            s_b(pos) = make_float4(q1.x, q1.y, fmaxf(pthr, -80.0f), __uint_as_float((uint32_t)progress));
            s_c(pos) = make_float4(q1.z, q1.w, q2.x, __uint_as_float(id));
        }
        if (tid < 4) {  // padding entries that can never pass the range test (walk runs in groups of 4, no bounds checks)
            s_a(kept + tid) = make_float4(-PARKED, -PARKED, 1.f, 0.f);
            s_b(kept + tid) = make_float4(1.f, 0.f, INFINITY, 0.f);
        }
        __syncthreads();
        if (!wave_done) {
            for (int j0 = 0; j0 < kept; j0 += 4) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int j = j0 + u;
                    const float4 a = s_a(j);
                    const float4 b = s_b(j);
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
                                    const float4 c = s_c(j);
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
    if (LOG) {
        // the quadrant's largest n_contrib = how far its log goes (the backward stages nothing of this quadrant behind it)
        uint32_t m = last_contributor;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
        if (lane == 0) quad_hi[4 * tile + wave_s] = m;
    }
    if (GSR_LOG_PROFILE && QUAD) {
        GSR_FPROF_MARK(4);
        fprof[0] = (unsigned long long)(clock64() - ft_begin);
        if (lane == 0)
            for (int t = 0; t < 5; t++) atomicAdd(&g_fwd_prof[t], fprof[t]);
    }
}

#undef s_a
#undef s_b
#undef s_c

}  // namespace

// blend_log: the R-sized mask array of the binning buffer (BinLayout::blend_log), or NULL (no log: the backward then re-derives every
// decision).  Written only by the quadrant walk (GSR_QUAD=1, GSR_WAVE8X8=1, the default): gsr_blend_log_for (api.hip) says whether it is.
int gsr_forward_profile_fetch(unsigned long long* out8)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_fwd_prof), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_prof), z, sizeof(z)) != hipSuccess) return -1;
    return GSR_LOG_PROFILE;
}

void gsr_launch_render_forward(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* splat,
                               const float* bg, float* final_T, uint32_t* n_contrib, float* out_color, const uint32_t* order,
                               unsigned long long* blend_log, uint32_t* quad_hi, hipStream_t s)
{
    if (!gsr_variant("LPT", 1)) order = nullptr;
    const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    const int wave8x8 = gsr_variant("WAVE8X8", 1);
    if (wave8x8 && gsr_variant("QUAD", 1)) {
        const int mode = gsr_variant("QUAD", 1) == 2 ? 2 : 1;
        if (blend_log && mode == 1)
            hipLaunchKernelGGL((render_forward_kernel<true, true>), dim3(gx * gy), dim3(256), (size_t)gsr_variant("FWD_PADLDS", 0), s, W, H, gx, gy, mode, order, ranges,
                               point_list, splat, bg, final_T, n_contrib, out_color, blend_log, quad_hi);
        else
            hipLaunchKernelGGL((render_forward_kernel<true, false>), dim3(gx * gy), dim3(256), (size_t)gsr_variant("FWD_PADLDS", 0), s, W, H, gx, gy, mode, order, ranges,
                               point_list, splat, bg, final_T, n_contrib, out_color, (unsigned long long*)nullptr, (uint32_t*)nullptr);
    } else
        hipLaunchKernelGGL((render_forward_kernel<false, false>), dim3(gx * gy), dim3(256), 0, s, W, H, gx, gy, wave8x8, order, ranges, point_list, splat, bg,
                           final_T, n_contrib, out_color, (unsigned long long*)nullptr, (uint32_t*)nullptr);
}
