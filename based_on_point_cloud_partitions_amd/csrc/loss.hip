// Loss block of the training step for MI355X: loss = (1 - lambda) * L1(img, gt) + lambda * (1 - SSIM(img, gt)) and its
// gradient w.r.t. img, the first consumer of the rendered image and the producer of dL/dcolor for the rasterizer's
// backward (train.py:116-118; utils/loss_utils.py:17-18 `l1_loss`, :23-31 window, :33-63 `ssim`/`_ssim`).
//
// The reference evaluates SSIM with five grouped 11x11 conv2d calls plus a dozen elementwise kernels and lets autograd
// replay them backwards (~70 passes over 3xHxW tensors per step).  Here it is two kernels:
//   forward : a 16x16 output tile per workgroup; img and gt tiles (with a 5-pixel halo, zero padded like conv2d's
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

constexpr int TB = 16;          // output tile edge
constexpr int HALO = 5;
constexpr int TIN = TB + 2 * HALO;  // 26

__global__ void __launch_bounds__(256) ssim_forward_kernel(int C, int H, int W, const float* __restrict__ img1,
                                                           const float* __restrict__ img2, float* __restrict__ maps,
                                                           float* __restrict__ partials)
{
    __shared__ float s_x[TIN][TIN + 1], s_y[TIN][TIN + 1];
    __shared__ float s_h[5][TIN][TB + 1];  // horizontally filtered x, y, xx, yy, xy
    __shared__ float s_red[2][4];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * TB, y0 = blockIdx.y * TB, c = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const float* a = img1 + (size_t)c * plane;
    const float* b = img2 + (size_t)c * plane;
    for (int i = threadIdx.x; i < TIN * TIN; i += 256) {
        const int ly = i / TIN, lx = i - ly * TIN;
        const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        s_x[ly][lx] = in ? a[(size_t)gy * W + gx] : 0.f;
        s_y[ly][lx] = in ? b[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TIN * TB; i += 256) {  // horizontal pass: 26 rows x 16 columns
        const int ly = i / TB, lx = i - ly * TB;
        float hx = 0.f, hy = 0.f, hxx = 0.f, hyy = 0.f, hxy = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float p = s_x[ly][lx + k], q = s_y[ly][lx + k], w = kW[k];
            hx += w * p; hy += w * q; hxx += w * (p * p); hyy += w * (q * q); hxy += w * (p * q);
        }
        s_h[0][ly][lx] = hx; s_h[1][ly][lx] = hy; s_h[2][ly][lx] = hxx; s_h[3][ly][lx] = hyy; s_h[4][ly][lx] = hxy;
    }
    __syncthreads();
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = kW[k];
        mu1 += w * s_h[0][ty + k][tx]; mu2 += w * s_h[1][ty + k][tx]; e11 += w * s_h[2][ty + k][tx];
        e22 += w * s_h[3][ty + k][tx]; e12 += w * s_h[4][ty + k][tx];
    }
    const int gx = x0 + tx, gy = y0 + ty;
    const bool inside = gx < W && gy < H;
    float m = 0.f, l1 = 0.f;
    if (inside) {
        const float C1 = 0.0001f, C2 = 0.0009f;  // 0.01**2, 0.03**2 (loss_utils.py:56-57)
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
        const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu1_mu2;
        const float A = 2.f * mu1_mu2 + C1, B = 2.f * s12 + C2, Cc = mu1_sq + mu2_sq + C1, D = s1 + s2 + C2;
        const float inv = 1.0f / (Cc * D);
        m = (A * B) * inv;
        const size_t id = (size_t)c * plane + (size_t)gy * W + gx;
        const size_t CP = (size_t)C * plane;
        maps[id] = (2.f * mu2 * B - 2.f * mu2 * A) * inv - m * (2.f * mu1 * D - 2.f * mu1 * Cc) * inv;  // d m / d mu1
        maps[CP + id] = -m / D;                                                                          // d m / d E[x^2]
        maps[2 * CP + id] = 2.f * A * inv;                                                               // d m / d E[xy]
        l1 = fabsf(s_x[ty + HALO][tx + HALO] - s_y[ty + HALO][tx + HALO]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { m += __shfl_down(m, off, 64); l1 += __shfl_down(l1, off, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = m; s_red[1][threadIdx.x >> 6] = l1; }
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

__global__ void __launch_bounds__(256) ssim_backward_kernel(int C, int H, int W, const float* __restrict__ img1,
                                                            const float* __restrict__ img2, const float* __restrict__ maps,
                                                            float lambda, float invN, const float* __restrict__ grad_loss,
                                                            float* __restrict__ dL_dimg1)
{
    __shared__ float s_m[3][TIN][TIN + 1];
    __shared__ float s_h[3][TIN][TB + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * TB, y0 = blockIdx.y * TB, c = blockIdx.z;
    const size_t plane = (size_t)H * W, CP = (size_t)C * plane;
    const float* mp = maps + (size_t)c * plane;
    for (int i = threadIdx.x; i < TIN * TIN; i += 256) {
        const int ly = i / TIN, lx = i - ly * TIN;
        const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = (size_t)gy * W + gx;
        s_m[0][ly][lx] = in ? mp[o] : 0.f;
        s_m[1][ly][lx] = in ? mp[CP + o] : 0.f;
        s_m[2][ly][lx] = in ? mp[2 * CP + o] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TIN * TB; i += 256) {
        const int ly = i / TB, lx = i - ly * TB;
        float h0 = 0.f, h1 = 0.f, h2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kW[k];
            h0 += w * s_m[0][ly][lx + k]; h1 += w * s_m[1][ly][lx + k]; h2 += w * s_m[2][ly][lx + k];
        }
        s_h[0][ly][lx] = h0; s_h[1][ly][lx] = h1; s_h[2][ly][lx] = h2;
    }
    __syncthreads();
    const int gx = x0 + tx, gy = y0 + ty;
    if (gx < W && gy < H) {
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kW[k];
            v0 += w * s_h[0][ty + k][tx]; v1 += w * s_h[1][ty + k][tx]; v2 += w * s_h[2][ty + k][tx];
        }
        const size_t id = (size_t)c * plane + (size_t)gy * W + gx;
        const float x = img1[id], y = img2[id];
        const float g = grad_loss ? grad_loss[0] : 1.0f;
        const float d = x - y;
        const float sgn = d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f);
        dL_dimg1[id] = g * ((-lambda * invN) * (v0 + 2.f * x * v1 + y * v2) + ((1.0f - lambda) * invN) * sgn);
    }
}

}  // namespace

size_t gsr_loss_scratch_layout(int C, int H, int W, size_t* maps_off, size_t* partials_off)
{
    const size_t n = (size_t)C * H * W;
    const size_t nblk = (size_t)((W + TB - 1) / TB) * ((H + TB - 1) / TB) * C;
    size_t o = 0;
    *maps_off = o; o = gsr_align(o + 3 * n * 4);
    *partials_off = o; o = gsr_align(o + 2 * nblk * 4);
    return o;
}

void gsr_launch_loss_forward(int C, int H, int W, const float* img1, const float* img2, float lambda, float* maps, float* partials,
                             float* out3, hipStream_t s)
{
    const dim3 grid((W + TB - 1) / TB, (H + TB - 1) / TB, C);
    hipLaunchKernelGGL(ssim_forward_kernel, grid, dim3(256), 0, s, C, H, W, img1, img2, maps, partials);
    const int nblk = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, s, nblk, partials, lambda, 1.0f / ((float)C * H * W), out3);
}

void gsr_launch_loss_backward(int C, int H, int W, const float* img1, const float* img2, const float* maps, float lambda,
                              const float* grad_loss, float* dL_dimg1, hipStream_t s)
{
    const dim3 grid((W + TB - 1) / TB, (H + TB - 1) / TB, C);
    hipLaunchKernelGGL(ssim_backward_kernel, grid, dim3(256), 0, s, C, H, W, img1, img2, maps, lambda, 1.0f / ((float)C * H * W),
                       grad_loss, dL_dimg1);
}
