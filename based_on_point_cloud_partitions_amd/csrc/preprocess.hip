// Per-Gaussian forward stages for MI355X (gfx950): frustum cull + 3D->2D covariance projection + SH colour
// (replaces preprocessCUDA, cuda_rasterizer/forward.cu:156-256), the tiles_touched prefix sum
// (cub::DeviceScan::InclusiveSum, rasterizer_impl.cu:277), key emission (duplicateWithKeys,
// rasterizer_impl.cu:70-111), tile ranges (identifyTileRanges, rasterizer_impl.cu:116-138) and
// checkFrustum (rasterizer_impl.cu:54-66).
//
// Arithmetic contract: plain IEEE fp32 evaluated in the reference's expression order with NO fma
// contraction (file is compiled with -ffp-contract=off) so radii / tiles_touched / depth bits -- and
// therefore the sort keys -- are bit-identical to the CPU oracle.
#include "common.h"

namespace {

__constant__ float kSH_C0 = 0.28209479177387814f;
__constant__ float kSH_C1 = 0.4886025119029199f;
__constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f};
__constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f};

struct F3 { float x, y, z; };

__device__ __forceinline__ F3 sh_to_rgb(int deg, const float* __restrict__ sh, F3 dir, bool* cl)
{
    const float x = dir.x, y = dir.y, z = dir.z;
    float res[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float result = kSH_C0 * sh[ch];
        if (deg > 0) {
            result = result - kSH_C1 * y * sh[3 + ch] + kSH_C1 * z * sh[6 + ch] - kSH_C1 * x * sh[9 + ch];
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                result = result + kSH_C2[0] * xy * sh[12 + ch] + kSH_C2[1] * yz * sh[15 + ch] +
                         kSH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + ch] + kSH_C2[3] * xz * sh[21 + ch] +
                         kSH_C2[4] * (xx - yy) * sh[24 + ch];
                if (deg > 2) {
                    result = result + kSH_C3[0] * y * (3.0f * xx - yy) * sh[27 + ch] + kSH_C3[1] * xy * z * sh[30 + ch] +
                             kSH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + ch] +
                             kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + ch] +
                             kSH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + ch] + kSH_C3[5] * z * (xx - yy) * sh[42 + ch] +
                             kSH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + ch];
                }
            }
        }
        result += 0.5f;
        cl[ch] = result < 0.0f;
        res[ch] = fmaxf(result, 0.0f);
    }
    return F3{res[0], res[1], res[2]};
}

// One thread per Gaussian, 256 per workgroup (4 waves).  Streams xyz/scale/rot/opacity/SH once,
// writes one 48-byte splat record + cov3D + clamped + tiles_touched, and the workgroup's
// tiles_touched sum (feeds the two-level prefix sum without an extra pass over P).
// Kernel-side copy of PreprocessArgs with __restrict__ pointers: lets the compiler hoist the input loads over the output
// stores and read the camera matrices with scalar loads.
struct PreprocessPtrs {
    int P, D, M, W, H, sh_direct;
    const float* __restrict__ means3D; const float* __restrict__ scales; const float* __restrict__ rotations;
    const float* __restrict__ opacities; const float* __restrict__ shs; const float* __restrict__ cov3D_precomp;
    const float* __restrict__ colors_precomp; const float* __restrict__ viewmatrix; const float* __restrict__ projmatrix;
    const float* __restrict__ campos; const float* __restrict__ features_dc; const float* __restrict__ features_rest;
    float scale_modifier, tan_fovx, tan_fovy;
    int* __restrict__ radii; float* __restrict__ splat; float* __restrict__ cov3D; uint8_t* __restrict__ clamped;
    uint32_t* __restrict__ tiles_touched; uint32_t* __restrict__ block_sums; uint32_t* __restrict__ row_block_sums;
    uint32_t* zero64;
    uint32_t* trap_flag; uint32_t trap_tag;
};

// RAW: scales / rotations / opacities are the pre-activation parameters (exp / normalize / sigmoid applied here) and the
// SH rows are assembled in LDS from features_dc + features_rest -- the four activation passes and the 96 MB torch.cat of
// scene/gaussian_model.py:102-122 never touch HBM.
#ifndef GSR_PRE_WPE_MAX
#define GSR_PRE_WPE_MAX 8
#endif
#ifndef GSR_PRE_WPE
#define GSR_PRE_WPE 0  // experiment (with GSR_PRE_WPE_MAX): waves per SIMD the register allocation aims at (0: the compiler's choice, 5 at 91 VGPRs)
#endif
template <bool RAW>
__global__ void __launch_bounds__(256)
#if GSR_PRE_WPE
__attribute__((amdgpu_waves_per_eu(GSR_PRE_WPE, GSR_PRE_WPE_MAX)))
#endif
preprocess_kernel(PreprocessPtrs a, float focal_x, float focal_y, int gx, int gy)
{
    // SH rows are 192 B per Gaussian: the workgroup's 256 rows come in through LDS with coalesced 16-byte loads and are read
    // per thread with an odd row stride (see gaussian_backward.hip)
    extern __shared__ float s_sh[];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int n3 = 3 * a.M, sh_stride = n3 + 1;
    if (RAW && !a.sh_direct) {
        const int rows = min(256, a.P - (int)blockIdx.x * 256);
        sh_cols_load(a.features_dc + (size_t)blockIdx.x * 256 * 3, s_sh, rows, 3, sh_stride, 0);
        if (n3 > 3) sh_cols_load(a.features_rest + (size_t)blockIdx.x * 256 * (n3 - 3), s_sh, rows, n3 - 3, sh_stride, 3);
        __syncthreads();
    } else if (a.shs && !a.sh_direct) {
        sh_tile_load(a.shs + (size_t)blockIdx.x * 256 * n3, s_sh, min(256, a.P - (int)blockIdx.x * 256), n3, sh_stride);
        __syncthreads();
    }
    uint32_t touched = 0, rows = 0;
    if (idx < a.P) {
        float4 rec0 = make_float4(0.f, 0.f, 0.f, 0.f), rec1 = rec0, rec2 = rec0, rec3 = rec0;  // culled: radius 0
        int radius_out = 0;
        const float px = a.means3D[3 * idx], py = a.means3D[3 * idx + 1], pz = a.means3D[3 * idx + 2];
        const float* pm = a.projmatrix;
        const float* vm = a.viewmatrix;
        // in_frustum (auxiliary.h:139-164)
        const float hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
        const float hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
        const float hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float ndc_x = hx * p_w, ndc_y = hy * p_w;
        float tx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
        float ty = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
        const float tz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
        if (tz > 0.2f) {
            float c0, c1, c2, c3, c4, c5;
            if (a.cov3D_precomp) {
                const float* c = a.cov3D_precomp + 6 * (size_t)idx;
                c0 = c[0]; c1 = c[1]; c2 = c[2]; c3 = c[3]; c4 = c[4]; c5 = c[5];
            } else {
                // computeCov3D (forward.cu:118-152): Sigma = Rq S^2 Rq^T with GLM's summation order
                float sc0 = a.scales[3 * idx], sc1 = a.scales[3 * idx + 1], sc2 = a.scales[3 * idx + 2];
                float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];
                if (RAW) {
                    sc0 = gsr_expf(sc0); sc1 = gsr_expf(sc1); sc2 = gsr_expf(sc2);
                    const float d = fmaxf(gsr_quat_norm(q.x, q.y, q.z, q.w), 1e-12f);
                    q.x = q.x / d; q.y = q.y / d; q.z = q.z / d; q.w = q.w / d;
                }
                const float s0 = a.scale_modifier * sc0, s1 = a.scale_modifier * sc1, s2 = a.scale_modifier * sc2;
                const float r = q.x, x = q.y, y = q.z, z = q.w;
                const float m00 = s0 * (1.f - 2.f * (y * y + z * z)), m01 = s1 * (2.f * (x * y - r * z)), m02 = s2 * (2.f * (x * z + r * y));
                const float m10 = s0 * (2.f * (x * y + r * z)), m11 = s1 * (1.f - 2.f * (x * x + z * z)), m12 = s2 * (2.f * (y * z - r * x));
                const float m20 = s0 * (2.f * (x * z - r * y)), m21 = s1 * (2.f * (y * z + r * x)), m22 = s2 * (1.f - 2.f * (x * x + y * y));
                c0 = m00 * m00 + m01 * m01 + m02 * m02;
                c1 = m10 * m00 + m11 * m01 + m12 * m02;
                c2 = m20 * m00 + m21 * m01 + m22 * m02;
                c3 = m10 * m10 + m11 * m11 + m12 * m12;
                c4 = m20 * m10 + m21 * m11 + m22 * m12;
                c5 = m20 * m20 + m21 * m21 + m22 * m22;
                float* o = a.cov3D + 6 * (size_t)idx;
                o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3; o[4] = c4; o[5] = c5;
            }
            // computeCov2D (forward.cu:74-113)
            const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
            const float txtz = tx / tz, tytz = ty / tz;
            tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
            ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
            const float J00 = focal_x / tz, J02 = -(focal_x * tx) / (tz * tz);
            const float J11 = focal_y / tz, J12 = -(focal_y * ty) / (tz * tz);
            // T[c][r] (GLM): T0r = row r of first column, T1r second column; third column is zero
            const float T00 = vm[0] * J00 + vm[1] * 0.0f + vm[2] * J02;
            const float T01 = vm[4] * J00 + vm[5] * 0.0f + vm[6] * J02;
            const float T02 = vm[8] * J00 + vm[9] * 0.0f + vm[10] * J02;
            const float T10 = vm[0] * 0.0f + vm[1] * J11 + vm[2] * J12;
            const float T11 = vm[4] * 0.0f + vm[5] * J11 + vm[6] * J12;
            const float T12 = vm[8] * 0.0f + vm[9] * J11 + vm[10] * J12;
            // A[c][r] = T[r][0]*Vrk[0][c] + T[r][1]*Vrk[1][c] + T[r][2]*Vrk[2][c]
            const float A00 = T00 * c0 + T01 * c1 + T02 * c2;
            const float A10 = T00 * c1 + T01 * c3 + T02 * c4;
            const float A20 = T00 * c2 + T01 * c4 + T02 * c5;
            const float A01 = T10 * c0 + T11 * c1 + T12 * c2;
            const float A11 = T10 * c1 + T11 * c3 + T12 * c4;
            const float A21 = T10 * c2 + T11 * c4 + T12 * c5;
            const float covx = (A00 * T00 + A10 * T01 + A20 * T02) + 0.3f;
            const float covy = A01 * T00 + A11 * T01 + A21 * T02;
            const float covz = (A01 * T10 + A11 * T11 + A21 * T12) + 0.3f;
            const float det = covx * covz - covy * covy;
            if (det != 0.0f) {
                const float det_inv = 1.f / det;
                const float conx = covz * det_inv, cony = -covy * det_inv, conz = covx * det_inv;
                const float mid = 0.5f * (covx + covz);
                const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
                const float lambda1 = mid + sq, lambda2 = mid - sq;
                const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
                // ndc2Pix in double (auxiliary.h:41-44)
                const float pix_x = (float)((((double)ndc_x + 1.0) * (double)a.W - 1.0) * 0.5);
                const float pix_y = (float)((((double)ndc_y + 1.0) * (double)a.H - 1.0) * 0.5);
                const int iradius = (int)my_radius;
                uint32_t mnx, mny, mxx, mxy;
                gsr_get_rect(pix_x, pix_y, iradius, gx, gy, mnx, mny, mxx, mxy);
                const uint32_t area = (mxx - mnx) * (mxy - mny);
                if (area != 0) {
                    float cr, cg, cb;
                    if (a.colors_precomp) {
                        cr = a.colors_precomp[3 * idx]; cg = a.colors_precomp[3 * idx + 1]; cb = a.colors_precomp[3 * idx + 2];
                    } else {
                        // computeColorFromSH (forward.cu:20-71)
                        float dx = px - a.campos[0], dy = py - a.campos[1], dz = pz - a.campos[2];
                        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                        F3 dir{dx / len, dy / len, dz / len};
                        bool cl[3];
                        F3 c;
                        if (RAW && a.sh_direct) {
                            // the same from the two raw tensors: 3 + 45 floats (rows of 12 and 180 bytes, dword-aligned pieces)
                            float sh[48];
                            const gsr_f3u d = *reinterpret_cast<const gsr_f3u*>(a.features_dc + (size_t)idx * 3);
                            sh[0] = d.x; sh[1] = d.y; sh[2] = d.z;
                            const float* row = a.features_rest + (size_t)idx * 45;
#pragma unroll
                            for (int q = 0; q < 11; q++) {
                                const gsr_f4u v = *reinterpret_cast<const gsr_f4u*>(row + 4 * q);
                                sh[3 + 4 * q] = v.x; sh[4 + 4 * q] = v.y; sh[5 + 4 * q] = v.z; sh[6 + 4 * q] = v.w;
                            }
                            sh[47] = row[44];
                            c = sh_to_rgb(3, sh, dir, cl);
                        } else if (!RAW && a.sh_direct) {
                            // degree 3, 16 coefficients: the visible Gaussian reads its own 192-byte row with twelve 16-byte loads (no
                            // LDS tile, so nothing but registers limits the waves in flight; culled Gaussians read nothing)
                            float sh[48];
                            const float4* row = reinterpret_cast<const float4*>(a.shs + (size_t)idx * 48);
#pragma unroll
                            for (int q = 0; q < 12; q++) {
                                const float4 v = row[q];
                                sh[4 * q] = v.x; sh[4 * q + 1] = v.y; sh[4 * q + 2] = v.z; sh[4 * q + 3] = v.w;
                            }
                            c = sh_to_rgb(3, sh, dir, cl);
                        } else {
                            c = sh_to_rgb(a.D, s_sh + threadIdx.x * sh_stride, dir, cl);
                        }
                        cr = c.x; cg = c.y; cb = c.z;
                        a.clamped[3 * idx] = cl[0]; a.clamped[3 * idx + 1] = cl[1]; a.clamped[3 * idx + 2] = cl[2];
                    }
                    rec0 = make_float4(pix_x, pix_y, conx, cony);
                    const float opa = RAW ? gsr_sigmoidf(a.opacities[idx]) : a.opacities[idx];
                    rec1 = make_float4(conz, opa, cr, cg);
                    rec2 = make_float4(cb, tz, __int_as_float(iradius), 0.0f);
                    radius_out = iradius;
                    touched = area;
                    // which tiles of the rectangle can the alpha >= 1/255 ellipse reach at all?  Decides which instances own a
                    // gradient row (common.h, gsr_reach_mask); the backward blend drops the others from its walk
                    if (area <= 64) {
                        const unsigned long long mask = gsr_reach_mask(pix_x, pix_y, conx, cony, conz, opa, mnx, mny, mxx, mxy);
                        rec3 = make_float4(0.f, __uint_as_float((uint32_t)mask), __uint_as_float((uint32_t)(mask >> 32)), __uint_as_float(0u));
                        rows = (uint32_t)__popcll(mask);
                    } else {
                        rec3 = make_float4(0.f, 0.f, 0.f, __uint_as_float(1u));
                        rows = area;
                    }
                }
            }
        } else if (a.trap_flag && tz <= 0.2f) {
            *a.trap_flag = a.trap_tag;  // "Point is filtered although prefiltered is set" (auxiliary.h:154-160); the host raises
        }
        float4* rec = reinterpret_cast<float4*>(a.splat + (size_t)idx * GSR_SPLAT_F);
        rec[0] = rec0; rec[1] = rec1; rec[2] = rec2; rec[3] = rec3;
        a.radii[idx] = radius_out;
        a.tiles_touched[idx] = touched;
    }
    // workgroup sums of tiles_touched and of the gradient-row counts
    uint32_t v = touched, r = rows;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { v += __shfl_down(v, off, 64); r += __shfl_down(r, off, 64); }
    __shared__ uint32_t wsum[4], rsum[4];
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = v; rsum[threadIdx.x >> 6] = r; }
    __syncthreads();
    // (Folding the block scan in here -- the workgroup that finishes last scans, found with a device-scope ticket counter -- was built
    // and measured in round 4: with a release fence per workgroup this kernel went 0.039 -> 0.166 ms (every fence writes back the
    // XCD's whole L2), with atomic stores of the two sums + s_waitcnt 0.039 + 0.008 -> 0.051 ms: 1,954 tickets on one address cost
    // more than the launch boundary they save.  profiles/README.md.)
    if (threadIdx.x == 0) {
        a.block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        a.row_block_sums[blockIdx.x] = rsum[0] + rsum[1] + rsum[2] + rsum[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < GSR_BAND_COUNTERS && a.zero64) a.zero64[threadIdx.x] = 0;  // filled by emit_partition_kernel's atomics
}

// Exclusive scan of the per-workgroup sums (single workgroup, 1024 threads, chunked with a carry).
// `mailbox` (optional) is a word of host-mapped pinned memory: the total is published there, tagged with `seq`, so the
// host can pick it up by polling instead of queueing a copy and synchronising the stream.
__global__ void __launch_bounds__(1024) scan_blocks_kernel(int n, const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                            uint32_t* __restrict__ total, const uint32_t* __restrict__ in2,
                                                            uint32_t* __restrict__ out2, unsigned long long* mailbox, uint32_t seq,
                                                            uint32_t* __restrict__ zero64)
{
    __shared__ uint32_t wtot[16], wtot2[16];
    __shared__ uint32_t carry_s, carry2_s;
    if (threadIdx.x == 0) { carry_s = 0; carry2_s = 0; }
    if (zero64 && threadIdx.x < 64) zero64[threadIdx.x] = 0;  // counters the next stage fills with atomics (a memset launch would cost more)
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < n ? in[i] : 0u;
        const uint32_t v2 = (in2 && i < n) ? in2[i] : 0u;
        uint32_t incl = v, incl2 = v2;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64), t2 = __shfl_up(incl2, off, 64);
            if (lane >= off) { incl += t; incl2 += t2; }
        }
        if (lane == 63) { wtot[wave] = incl; wtot2[wave] = incl2; }
        __syncthreads();
        uint32_t wbase = 0, wbase2 = 0;
        for (int w = 0; w < wave; w++) { wbase += wtot[w]; wbase2 += wtot2[w]; }
        const uint32_t carry = carry_s, carry2 = carry2_s;
        if (i < n) {
            out[i] = carry + wbase + incl - v;
            if (in2) out2[i] = carry2 + wbase2 + incl2 - v2;
        }
        __syncthreads();
        if (threadIdx.x == 1023) { carry_s = carry + wbase + incl; carry2_s = carry2 + wbase2 + incl2; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *total = carry_s;
        if (in2) out2[n] = carry2_s;
        if (mailbox)
            __hip_atomic_store(mailbox, ((unsigned long long)seq << 32) | carry_s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Key emission.  Workgroup-local exclusive scan of tiles_touched + the scanned workgroup offset gives
// each Gaussian its first instance slot; keys are (tile id << 32) | depth bits, values the Gaussian index
// (rasterizer_impl.cu:70-111).  Also records the inclusive offsets (point_offsets) and the exclusive
// offset inside the splat record (used by the backward pass to address per-instance gradient rows).
// (The banded tile binning does this inside its partition pass: emit_partition_kernel, binning.hip.)
__global__ void __launch_bounds__(256) emit_keys_kernel(int P, int gx, int gy, float* __restrict__ splat,
                                                        const uint32_t* __restrict__ tiles_touched,
                                                        const uint32_t* __restrict__ block_offsets,
                                                        uint32_t* __restrict__ point_offsets,
                                                        const uint32_t* __restrict__ row_block_offsets, uint32_t* __restrict__ row_offsets,
                                                        uint32_t* __restrict__ wg_first, uint64_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = idx < P ? tiles_touched[idx] : 0u;
    // gradient rows of this Gaussian (fourth quarter of its record, written by the preprocess kernel)
    uint32_t nr = 0;
    if (n != 0) {
        const float4 q3 = *reinterpret_cast<const float4*>(splat + (size_t)idx * GSR_SPLAT_F + 12);
        nr = gsr_splat_rows(__float_as_uint(q3.y), __float_as_uint(q3.z), __float_as_uint(q3.w), n);
    }
    uint32_t incl = n, incl_r = nr;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64), tr = __shfl_up(incl_r, off, 64);
        if (lane >= off) { incl += t; incl_r += tr; }
    }
    __shared__ uint32_t wtot[4], wtot_r[4];
    if (lane == 63) { wtot[wave] = incl; wtot_r[wave] = incl_r; }
    __syncthreads();
    uint32_t base = block_offsets[blockIdx.x], base_r = row_block_offsets[blockIdx.x];
    for (int w = 0; w < wave; w++) { base += wtot[w]; base_r += wtot_r[w]; }
    if (idx >= P) return;
    const uint32_t end = base + incl;
    uint32_t off = end - n;
    point_offsets[idx] = end;
    row_offsets[idx] = base_r + incl_r;
    gsr_part_store(wg_first, P, idx, base_r + incl_r - nr, base_r + incl_r, row_block_offsets[(P + 255) >> 8]);
    if (n == 0) return;
    float* rec = splat + (size_t)idx * GSR_SPLAT_F;
    rec[SP_ROWBASE] = __uint_as_float(base_r + incl_r - nr);
    const float mx = rec[SP_MX], my = rec[SP_MY];
    const uint32_t dbits = __float_as_uint(rec[SP_DEPTH]);
    const int radius = __float_as_int(rec[SP_RADIUS]);
    rec[SP_OFFS] = __uint_as_float(off);
    if (keys == nullptr) return;  // offsets only: the tile-binning path (binning.hip) builds the lists itself
    uint32_t mnx, mny, mxx, mxy;
    gsr_get_rect(mx, my, radius, gx, gy, mnx, mny, mxx, mxy);
    for (uint32_t y = mny; y < mxy; y++)
        for (uint32_t x = mnx; x < mxx; x++) {
            keys[off] = ((uint64_t)(y * (uint32_t)gx + x) << 32) | dbits;
            vals[off] = (uint32_t)idx;
            off++;
        }
}

// identifyTileRanges (rasterizer_impl.cu:116-138); ranges must be zeroed first (:310).
__global__ void __launch_bounds__(256) tile_ranges_kernel(uint32_t L, const uint64_t* __restrict__ keys, uint32_t* __restrict__ ranges)
{
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= L) return;
    const uint32_t cur = (uint32_t)(keys[idx] >> 32);
    if (idx == 0) ranges[2 * cur] = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (cur != prev) { ranges[2 * prev + 1] = idx; ranges[2 * cur] = idx; }
    }
    if (idx == L - 1) ranges[2 * cur + 1] = L;
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                           const float* __restrict__ vm, uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
    const float tz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    present[idx] = tz <= 0.2f ? 0 : 1;
}

}  // namespace

void gsr_launch_preprocess(const PreprocessArgs& a, hipStream_t s)
{
    const float focal_y = a.H / (2.0f * a.tan_fovy);  // rasterizer_impl.cu:222-223
    const float focal_x = a.W / (2.0f * a.tan_fovx);
    const int gx = (a.W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (a.H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    PreprocessPtrs p;
    p.P = a.P; p.D = a.D; p.M = a.M; p.W = a.W; p.H = a.H;
    p.means3D = a.means3D; p.scales = a.scales; p.rotations = a.rotations; p.opacities = a.opacities; p.shs = a.shs;
    p.cov3D_precomp = a.cov3D_precomp; p.colors_precomp = a.colors_precomp; p.viewmatrix = a.viewmatrix;
    p.projmatrix = a.projmatrix; p.campos = a.campos;
    p.scale_modifier = a.scale_modifier; p.tan_fovx = a.tan_fovx; p.tan_fovy = a.tan_fovy;
    p.radii = a.radii; p.splat = a.splat; p.cov3D = a.cov3D; p.clamped = a.clamped; p.tiles_touched = a.tiles_touched;
    p.block_sums = a.block_sums; p.row_block_sums = a.row_block_sums; p.features_dc = a.features_dc; p.features_rest = a.features_rest;
    p.zero64 = a.zero64;
    p.trap_flag = a.trap_flag; p.trap_tag = a.trap_tag;
    p.sh_direct = ((a.raw ? a.features_rest != nullptr : a.shs != nullptr) && a.M == 16 && a.D == 3 && gsr_variant("PRE_DIRECT", 1)) ? 1 : 0;
    const size_t lds = (((a.shs || a.raw) && !p.sh_direct) ? (size_t)256 * (3 * a.M + 1) * 4 : 0) + (size_t)gsr_variant("LDS_PAD", 0);
    if (a.raw)
        hipLaunchKernelGGL(preprocess_kernel<true>, dim3((a.P + 255) / 256), dim3(256), lds, s, p, focal_x, focal_y, gx, gy);
    else
        hipLaunchKernelGGL(preprocess_kernel<false>, dim3((a.P + 255) / 256), dim3(256), lds, s, p, focal_x, focal_y, gx, gy);
}

void gsr_launch_scan_blocks(int nblocks, const uint32_t* block_sums, uint32_t* block_offsets, uint32_t* total,
                            unsigned long long* mailbox, uint32_t seq, hipStream_t s)
{
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, s, nblocks, block_sums, block_offsets, total, nullptr, nullptr, mailbox, seq,
                       (uint32_t*)nullptr);
}

void gsr_launch_scan_blocks2(int nblocks, const uint32_t* block_sums, uint32_t* block_offsets, uint32_t* total, const uint32_t* in2,
                             uint32_t* out2, unsigned long long* mailbox, uint32_t seq, uint32_t* zero64, hipStream_t s)
{
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, s, nblocks, block_sums, block_offsets, total, in2, out2, mailbox, seq, zero64);
}

void gsr_launch_emit(int P, int W, int H, float* splat, const uint32_t* tiles_touched, const uint32_t* block_offsets,
                     uint32_t* point_offsets, const uint32_t* row_block_offsets, uint32_t* row_offsets, uint32_t* wg_first,
                     uint64_t* keys, uint32_t* vals, hipStream_t s)
{
    const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    hipLaunchKernelGGL(emit_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, gx, gy, splat, tiles_touched, block_offsets,
                       point_offsets, row_block_offsets, row_offsets, wg_first, keys, vals);
}

void gsr_launch_tile_ranges(uint32_t R, const uint64_t* keys_sorted, uint32_t* ranges, uint32_t ntiles, hipStream_t s)
{
    hipMemsetAsync(ranges, 0, (size_t)ntiles * 8, s);
    if (R > 0) hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, s, R, keys_sorted, ranges);
}

void gsr_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
}
