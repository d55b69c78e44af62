// Loss block of the training step for MI355X: loss = (1 - lambda) * L1(img, gt) + lambda * (1 - SSIM(img, gt)) and its
// gradient w.r.t. img, the first consumer of the rendered image and the producer of dL/dcolor for the rasterizer's
// backward (train.py:116-118; utils/loss_utils.py:17-18 `l1_loss`, :23-31 window, :33-63 `ssim`/`_ssim`).
//
// The reference evaluates SSIM with five grouped 11x11 conv2d calls plus a dozen elementwise kernels and lets autograd
// replay them backwards (~70 passes over 3xHxW tensors per step).  Here it is two kernels:
//   forward : a 32x32 output tile per workgroup (4 outputs per thread); img and gt tiles (with a 5-pixel halo, zero padded like conv2d's
//             padding=5) go to LDS once, the Gaussian window is applied separably (11 + 11 taps instead of 121) to
//             x, y, x^2, y^2, x*y; the SSIM map value, |x - y| and the three partial derivatives of the map w.r.t. the
//             window sums (d/dmu1, d/dE[x^2], d/dE[xy]) are produced in registers; per-workgroup partial sums of the
//             map and of |x - y| are stored and reduced by one more workgroup in a fixed order (deterministic loss);
//   backward: the three derivative maps are filtered with the same separable window (the adjoint of a symmetric
//             zero-padded convolution is the same convolution) and combined into dL/dimg, written CHW.
// fp32 throughout; results match the reference's loss_utils.py within 1e-6 relative (different summation order).
#include "common.h"

namespace {

// utils/loss_utils.py:23-25: exp(-(x-5)^2 / (2*1.5^2)) normalised in float32
__constant__ float kW[11] = {0x1.0d956cp-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f, 0x1.b43c3ep-3f, 0x1.10656p-2f,
                             0x1.b43c3ep-3f,  0x1.bff0fep-4f, 0x1.26eb18p-5f, 0x1.f1fe02p-8f, 0x1.0d956cp-10f};

constexpr int TB = 32;               // output tile width; the height TBY (32 or 16) is a template parameter: 256 threads, TBY/8 outputs each
constexpr int HALO = 5;
constexpr int TIN = TB + 2 * HALO;   // 42
constexpr int SX = TIN + 1;          // LDS row stride of the input tiles: 43 (the 4-wide horizontal tasks hit 32 distinct banks)
constexpr int SH = TB + 1;           // LDS row stride of the horizontally filtered tiles

// Register-blocked separable filter.  Horizontal pass: a task = (input row, group of 4 adjacent outputs) reads 14 values per
// image once and produces 4 x 5 filtered values; vertical pass: a thread = (column, group of 4 adjacent rows) reads 14 rows
// x 5 quantities once and produces 4 outputs.  Per pixel that is ~37 LDS operations and ~240 VALU instructions instead of
// ~100 and ~450 for one output per thread (the first version, 16x16 tiles); every sum still runs over the taps in
// ascending order; the taps are accumulated with fused multiply-adds (the file is compiled without contraction, and the
// loss is compared with the reference at 1e-6, not bit for bit).
template <int TBY>
__global__ void __launch_bounds__(256) ssim_forward_kernel(int C, int H, int W, const float* __restrict__ img1,
                                                           const float* __restrict__ img2, const float* __restrict__ img_l1,
                                                           float* __restrict__ maps, float* __restrict__ partials)
{
    constexpr int TINY = TBY + 2 * HALO, RPT = TBY / 8;
    __shared__ float s_x[TINY][SX], s_y[TINY][SX];
    __shared__ float s_h[5][TINY][SH];  // horizontally filtered x, y, xx, yy, xy
    __shared__ float s_red[2][4];
    const int x0 = blockIdx.x * TB, y0 = blockIdx.y * TBY, c = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const float* a = img1 + (size_t)c * plane;
    const float* b = img2 + (size_t)c * plane;
    // the separate L1 image's values for this thread's outputs: requested now, used in the epilogue (a load issued there
    // would leave its whole latency exposed at the end of the workgroup)
    float xl_pre[RPT];
    if (img_l1) {
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            const int gy = y0 + (int)(threadIdx.x >> 5) * RPT + j, gx = x0 + (int)(threadIdx.x & 31);
            xl_pre[j] = (gx < W && gy < H) ? img_l1[(size_t)c * plane + (size_t)gy * W + gx] : 0.f;
        }
    }
    for (int i = threadIdx.x; i < TINY * TIN; i += 256) {
        const int ly = i / TIN, lx = i - ly * TIN;
        const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        s_x[ly][lx] = in ? a[(size_t)gy * W + gx] : 0.f;
        s_y[ly][lx] = in ? b[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < TINY * (TB / 4); t += 256) {  // horizontal pass: 42 rows x 8 groups of 4 columns
        const int r = t >> 3, cg = (t & 7) * 4;
        float p[14], q[14], pp[14], qq[14], pq[14];
#pragma unroll
        for (int k = 0; k < 14; k++) {
            p[k] = s_x[r][cg + k]; q[k] = s_y[r][cg + k];
            pp[k] = p[k] * p[k]; qq[k] = q[k] * q[k]; pq[k] = p[k] * q[k];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float hx = 0.f, hy = 0.f, hxx = 0.f, hyy = 0.f, hxy = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = kW[k];
                hx = __builtin_fmaf(w, p[j + k], hx); hy = __builtin_fmaf(w, q[j + k], hy); hxx = __builtin_fmaf(w, pp[j + k], hxx);
                hyy = __builtin_fmaf(w, qq[j + k], hyy); hxy = __builtin_fmaf(w, pq[j + k], hxy);
            }
            s_h[0][r][cg + j] = hx; s_h[1][r][cg + j] = hy; s_h[2][r][cg + j] = hxx; s_h[3][r][cg + j] = hyy; s_h[4][r][cg + j] = hxy;
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, rg = (threadIdx.x >> 5) * RPT;  // column, first of this thread's 4 rows
    float acc[RPT][5];
#pragma unroll
    for (int j = 0; j < RPT; j++)
#pragma unroll
        for (int u = 0; u < 5; u++) acc[j][u] = 0.f;
#pragma unroll
    for (int rr = 0; rr < RPT + 10; rr++) {
        float v[5];
#pragma unroll
        for (int u = 0; u < 5; u++) v[u] = s_h[u][rg + rr][tx];
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            const int k = rr - j;  // tap of output row rg + j that input row rg + rr feeds (ascending in rr, as before)
            if (k >= 0 && k < 11) {
#pragma unroll
                for (int u = 0; u < 5; u++) acc[j][u] = __builtin_fmaf(kW[k], v[u], acc[j][u]);
            }
        }
    }
    const int gx = x0 + tx;
    float m_sum = 0.f, l1_sum = 0.f;
    const size_t CP = (size_t)C * plane;
#pragma unroll
    for (int j = 0; j < RPT; j++) {
        const int gy = y0 + rg + j;
        if (gx < W && gy < H) {
            const float mu1 = acc[j][0], mu2 = acc[j][1], e11 = acc[j][2], e22 = acc[j][3], e12 = acc[j][4];
            const float C1 = 0.0001f, C2 = 0.0009f;  // 0.01**2, 0.03**2 (loss_utils.py:56-57)
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
            const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu1_mu2;
            const float A = 2.f * mu1_mu2 + C1, B = 2.f * s12 + C2, Cc = mu1_sq + mu2_sq + C1, D = s1 + s2 + C2;
            // v_rcp_f32 (1 ulp) instead of two IEEE divisions (~12 instructions each); the loss is compared at 1e-6
            const float inv = __builtin_amdgcn_rcpf(Cc * D);
            const float m = (A * B) * inv;
            const size_t id = (size_t)c * plane + (size_t)gy * W + gx;
            maps[id] = (2.f * mu2 * B - 2.f * mu2 * A) * inv - m * (2.f * mu1 * D - 2.f * mu1 * Cc) * inv;  // d m / d mu1
            maps[CP + id] = -(A * B) * (inv * __builtin_amdgcn_rcpf(D));                                     // d m / d E[x^2] = -m / D
            maps[2 * CP + id] = 2.f * A * inv;                                                               // d m / d E[xy]
            m_sum += m;
            // L1 on the rendered image itself, or on a separate image (the appearance-decoupled one of train.py:112-116)
            const float xl = img_l1 ? xl_pre[j] : s_x[rg + j + HALO][tx + HALO];
            l1_sum += fabsf(xl - s_y[rg + j + HALO][tx + HALO]);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { m_sum += __shfl_down(m_sum, off, 64); l1_sum += __shfl_down(l1_sum, off, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = m_sum; s_red[1][threadIdx.x >> 6] = l1_sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[2 * blk] = ((s_red[0][0] + s_red[0][1]) + s_red[0][2]) + s_red[0][3];
        partials[2 * blk + 1] = ((s_red[1][0] + s_red[1][1]) + s_red[1][2]) + s_red[1][3];
    }
}

// Fixed-order reduction of the per-workgroup partial sums -> out[0] = loss, out[1] = L1, out[2] = SSIM.
__global__ void __launch_bounds__(1024) loss_finalize_kernel(int nblocks, const float* __restrict__ partials, float lambda, float invN,
                                                             float* __restrict__ out)
{
    __shared__ double s_m[16], s_l[16];
    double m = 0.0, l = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 1024) { m += (double)partials[2 * i]; l += (double)partials[2 * i + 1]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { m += __shfl_down(m, off, 64); l += __shfl_down(l, off, 64); }
    if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = m; s_l[threadIdx.x >> 6] = l; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double M = 0.0, L = 0.0;
        for (int w = 0; w < 16; w++) { M += s_m[w]; L += s_l[w]; }
        const float ssim = (float)(M * (double)invN), l1 = (float)(L * (double)invN);
        out[0] = (1.0f - lambda) * l1 + lambda * (1.0f - ssim);
        out[1] = l1;
        out[2] = ssim;
    }
}

template <int TBY>
__global__ void __launch_bounds__(256) ssim_backward_kernel(int C, int H, int W, const float* __restrict__ img1,
                                                            const float* __restrict__ img2, const float* __restrict__ maps,
                                                            const float* __restrict__ img_l1, float lambda, float invN,
                                                            const float* __restrict__ grad_loss, float* __restrict__ dL_dimg1,
                                                            float* __restrict__ dL_dimg_l1)
{
    constexpr int TINY = TBY + 2 * HALO, RPT = TBY / 8;
    __shared__ float s_m[3][TINY][SX];
    __shared__ float s_h[3][TINY][SH];
    const int x0 = blockIdx.x * TB, y0 = blockIdx.y * TBY, c = blockIdx.z;
    const size_t plane = (size_t)H * W, CP = (size_t)C * plane;
    const float* mp = maps + (size_t)c * plane;
    // the epilogue's per-pixel inputs, requested before the filtering instead of after it
    float x_pre[RPT], y_pre[RPT], xl_pre[RPT];
#pragma unroll
    for (int j = 0; j < RPT; j++) {
        const int gy = y0 + (int)(threadIdx.x >> 5) * RPT + j, gx = x0 + (int)(threadIdx.x & 31);
        const bool in = gx < W && gy < H;
        const size_t id = in ? (size_t)c * plane + (size_t)gy * W + gx : 0;
        x_pre[j] = in ? img1[id] : 0.f;
        y_pre[j] = in ? img2[id] : 0.f;
        xl_pre[j] = (in && img_l1) ? img_l1[id] : 0.f;
    }
    for (int i = threadIdx.x; i < TINY * TIN; i += 256) {
        const int ly = i / TIN, lx = i - ly * TIN;
        const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = (size_t)gy * W + gx;
        s_m[0][ly][lx] = in ? mp[o] : 0.f;
        s_m[1][ly][lx] = in ? mp[CP + o] : 0.f;
        s_m[2][ly][lx] = in ? mp[2 * CP + o] : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < TINY * (TB / 4); t += 256) {  // horizontal pass, 4 outputs per task (see the forward)
        const int r = t >> 3, cg = (t & 7) * 4;
#pragma unroll
        for (int u = 0; u < 3; u++) {
            float p[14];
#pragma unroll
            for (int k = 0; k < 14; k++) p[k] = s_m[u][r][cg + k];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float h = 0.f;
#pragma unroll
                for (int k = 0; k < 11; k++) h = __builtin_fmaf(kW[k], p[j + k], h);
                s_h[u][r][cg + j] = h;
            }
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, rg = (threadIdx.x >> 5) * RPT;
    float acc[RPT][3];
#pragma unroll
    for (int j = 0; j < RPT; j++) { acc[j][0] = 0.f; acc[j][1] = 0.f; acc[j][2] = 0.f; }
#pragma unroll
    for (int rr = 0; rr < RPT + 10; rr++) {
        const float v0 = s_h[0][rg + rr][tx], v1 = s_h[1][rg + rr][tx], v2 = s_h[2][rg + rr][tx];
#pragma unroll
        for (int j = 0; j < RPT; j++) {
            const int k = rr - j;
            if (k >= 0 && k < 11) {
                acc[j][0] = __builtin_fmaf(kW[k], v0, acc[j][0]); acc[j][1] = __builtin_fmaf(kW[k], v1, acc[j][1]);
                acc[j][2] = __builtin_fmaf(kW[k], v2, acc[j][2]);
            }
        }
    }
    const int gx = x0 + tx;
    const float g = grad_loss ? grad_loss[0] : 1.0f;
#pragma unroll
    for (int j = 0; j < RPT; j++) {
        const int gy = y0 + rg + j;
        if (gx < W && gy < H) {
            const size_t id = (size_t)c * plane + (size_t)gy * W + gx;
            const float x = x_pre[j], y = y_pre[j];
            const float d = (img_l1 ? xl_pre[j] : x) - y;
            const float sgn = d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f);
            const float g_ssim = (-lambda * invN) * (acc[j][0] + 2.f * x * acc[j][1] + y * acc[j][2]);
            const float g_l1 = ((1.0f - lambda) * invN) * sgn;
            if (img_l1) {
                dL_dimg1[id] = g * g_ssim;
                dL_dimg_l1[id] = g * g_l1;
            } else {
                dL_dimg1[id] = g * (g_ssim + g_l1);
            }
        }
    }
}

}  // namespace

size_t gsr_loss_scratch_layout(int C, int H, int W, size_t* maps_off, size_t* partials_off)
{
    const size_t n = (size_t)C * H * W;
    const size_t nblk = (size_t)((W + TB - 1) / TB) * ((H + 15) / 16) * C;  // enough for either tile height
    size_t o = 0;
    *maps_off = o; o = gsr_align(o + 3 * n * 4);
    *partials_off = o; o = gsr_align(o + 2 * nblk * 4);
    return o;
}

void gsr_launch_loss_forward(int C, int H, int W, const float* img1, const float* img2, const float* img_l1, float lambda,
                             float* maps, float* partials, float* out3, hipStream_t s)
{
    const int tby = gsr_variant("LOSS_TBY", 32) == 16 ? 16 : 32;
    const dim3 grid((W + TB - 1) / TB, (H + tby - 1) / tby, C);
    if (tby == 16) hipLaunchKernelGGL(ssim_forward_kernel<16>, grid, dim3(256), 0, s, C, H, W, img1, img2, img_l1, maps, partials);
    else hipLaunchKernelGGL(ssim_forward_kernel<32>, grid, dim3(256), 0, s, C, H, W, img1, img2, img_l1, maps, partials);
    const int nblk = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, s, nblk, partials, lambda, 1.0f / ((float)C * H * W), out3);
}

void gsr_launch_loss_backward(int C, int H, int W, const float* img1, const float* img2, const float* img_l1, const float* maps,
                              float lambda, const float* grad_loss, float* dL_dimg1, float* dL_dimg_l1, hipStream_t s)
{
    const int tby = gsr_variant("LOSS_TBY", 32) == 16 ? 16 : 32;
    const dim3 grid((W + TB - 1) / TB, (H + tby - 1) / tby, C);
    if (tby == 16)
        hipLaunchKernelGGL(ssim_backward_kernel<16>, grid, dim3(256), 0, s, C, H, W, img1, img2, maps, img_l1, lambda,
                           1.0f / ((float)C * H * W), grad_loss, dL_dimg1, dL_dimg_l1);
    else
        hipLaunchKernelGGL(ssim_backward_kernel<32>, grid, dim3(256), 0, s, C, H, W, img1, img2, maps, img_l1, lambda,
                           1.0f / ((float)C * H * W), grad_loss, dL_dimg1, dL_dimg_l1);
}
