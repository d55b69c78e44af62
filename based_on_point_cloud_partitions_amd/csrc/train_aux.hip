// Per-iteration bookkeeping of the training loop around the rasterizer, fused:
//  * densification statistics (train.py:140-142 + GaussianModel.add_densification_stats, scene/gaussian_model.py:433-435):
//        max_radii2D[vis] = max(max_radii2D[vis], radii[vis]); xyz_gradient_accum[vis] += ||viewspace_grad[vis, :2]||;
//        denom[vis] += 1          with vis = radii > 0
//    three masked read-modify-writes in torch (each a nonzero + gather + scatter, with a device sync) -> one streaming pass;
//  * order-preserving row compaction by a mask for ALL tensors of the model at once (the `t[mask]` of prune_points /
//    _prune_optimizer, gaussian_model.py:297-331: 6 parameters, their 12 Adam moments and 3 statistics = 21 boolean
//    indexings in the reference).
// Integer / copy work, HBM-bound.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) densification_stats_kernel(int P, const int* __restrict__ radii, const float* __restrict__ vgrad,
                                                                  float* __restrict__ max_radii2D, float* __restrict__ accum,
                                                                  float* __restrict__ denom)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
    const float gx = vgrad[3 * (size_t)i], gy = vgrad[3 * (size_t)i + 1];
    accum[i] = accum[i] + sqrtf(gx * gx + gy * gy);
    denom[i] = denom[i] + 1.0f;
}

// number of kept rows per 256-row block
__global__ void __launch_bounds__(256) compact_count_kernel(int P, const uint8_t* __restrict__ keep, uint32_t* __restrict__ block_sums)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool k = i < P && keep[i] != 0;
    const uint32_t c = __popcll(__builtin_amdgcn_ballot_w64(k));
    __shared__ uint32_t w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

constexpr int COMPACT_MAX_TENSORS = 24;
struct CompactTensor { const float* __restrict__ src; float* __restrict__ dst; int row_floats; };
struct CompactArgs {
    CompactTensor t[COMPACT_MAX_TENSORS];
    int n_tensors;
};

// one workgroup per 256 source rows: destination row of every kept row, then the kept rows of each tensor are copied with
// the element index running over the block's [256][row_floats] slab (coalesced reads, row-contiguous writes)
__global__ void __launch_bounds__(256) compact_rows_kernel(int P, const uint8_t* __restrict__ keep, const uint32_t* __restrict__ block_offsets,
                                                           CompactArgs args)
{
    __shared__ uint32_t s_pos[256];  // destination row, or 0xFFFFFFFF
    __shared__ uint32_t w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = blockIdx.x * 256;
    const int i = first + threadIdx.x;
    const bool k = i < P && keep[i] != 0;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(k);
    if (lane == 0) w[wave] = __popcll(m);
    __syncthreads();
    uint32_t base = block_offsets[blockIdx.x];
    for (int q = 0; q < wave; q++) base += w[q];
    s_pos[threadIdx.x] = k ? base + __popcll(m & ((1ull << lane) - 1ull)) : 0xFFFFFFFFu;
    __syncthreads();
    const int rows = min(256, P - first);
    for (int ti = 0; ti < args.n_tensors; ti++) {
        const CompactTensor t = args.t[ti];
        const int rf = t.row_floats;
        const float* src = t.src + (size_t)first * rf;
        const int count = rows * rf;
        for (int e = threadIdx.x; e < count; e += 256) {
            const int r = e / rf, c = e - r * rf;
            const uint32_t d = s_pos[r];
            if (d != 0xFFFFFFFFu) t.dst[(size_t)d * rf + c] = src[e];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// mse of utils/image_utils.py:17-18: (((img1 - img2)) ** 2).view(rows, -1).mean(1) -- one value per leading index (per
// channel for a [3,H,W] image as train.py:227 passes it, per image for the [1,3,H,W] tensors of metrics.py:93).  Two fixed
// order stages (256 partial sums per row in double, then their sum): bitwise reproducible.
// ------------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------------
// Bias + ReLU after a convolution, and their backward with the bias gradient (the `+ bias`, nn.ReLU, threshold_backward and the
// bias-gradient reduction that torch runs as four kernels around each Conv2d of the appearance head,
// scene/ptgs/appearance_network.py:20-29,48): y [C][n] = max(x + b[c], 0) in place; dz = [y > 0] g, db[c] = sum_n dz[c][n]
// (two fixed-order stages, sums in double).
// ------------------------------------------------------------------------------------------------------------------
constexpr int BR_PARTS = 32;

__global__ void __launch_bounds__(256) bias_relu_fwd_kernel(long long n, float* __restrict__ x, const float* __restrict__ bias)
{
    const int c = blockIdx.y;
    const float b = bias[c];
    float* p = x + (size_t)c * n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = p[i] + b;
        p[i] = v > 0.0f ? v : (v != v ? v : 0.0f);  // torch's ReLU: NaN stays NaN
    }
}

__global__ void __launch_bounds__(256) bias_relu_bwd_kernel(long long n, const float* __restrict__ g, const float* __restrict__ y,
                                                            float* __restrict__ dz, double* __restrict__ partials)
{
    __shared__ double s_w[4];
    const int c = blockIdx.y, part = blockIdx.x;
    const long long per = (n + BR_PARTS - 1) / BR_PARTS, lo = part * per, hi = lo + per < n ? lo + per : n;
    const size_t base = (size_t)c * n;
    double acc = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const float yv = y[base + i];
        const float v = yv <= 0.0f ? 0.0f : g[base + i];  // threshold_backward: a NaN activation lets the gradient through
        dz[base + i] = v;
        acc += (double)v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[(size_t)c * BR_PARTS + part] = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}

__global__ void __launch_bounds__(64) bias_relu_bwd_final_kernel(int C, const double* __restrict__ partials, float* __restrict__ db)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    double acc = 0.0;
    for (int p = 0; p < BR_PARTS; p++) acc += partials[(size_t)c * BR_PARTS + p];
    db[c] = (float)acc;
}

constexpr int MSE_PARTS = 256;

__global__ void __launch_bounds__(256) mse_partial_kernel(long long n, const float* __restrict__ a, const float* __restrict__ b,
                                                          double* __restrict__ partials)
{
    __shared__ double s_w[4];
    const int row = blockIdx.y, part = blockIdx.x;
    const long long per = (n + MSE_PARTS - 1) / MSE_PARTS, lo = part * per, hi = lo + per < n ? lo + per : n;
    const float* pa = a + (size_t)row * n;
    const float* pb = b + (size_t)row * n;
    double acc = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const float d = pa[i] - pb[i];
        acc += (double)(d * d);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[(size_t)row * MSE_PARTS + part] = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}

__global__ void __launch_bounds__(256) mse_final_kernel(long long n, const double* __restrict__ partials, float* __restrict__ out)
{
    __shared__ double s_w[4];
    double acc = partials[(size_t)blockIdx.x * MSE_PARTS + threadIdx.x];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)((((s_w[0] + s_w[1]) + s_w[2]) + s_w[3]) / (double)n);
}

}  // namespace

void gsr_launch_densification_stats(int P, const int* radii, const float* vgrad, float* max_radii2D, float* accum, float* denom, hipStream_t s)
{
    hipLaunchKernelGGL(densification_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, radii, vgrad, max_radii2D, accum, denom);
}

void gsr_launch_compact_count(int P, const uint8_t* keep, uint32_t* block_sums, hipStream_t s)
{
    hipLaunchKernelGGL(compact_count_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, keep, block_sums);
}

void gsr_launch_compact_rows(int P, const uint8_t* keep, const uint32_t* block_offsets, int n_tensors, const GsrCompactTensorHost* t,
                             hipStream_t s)
{
    int k = 0;  // runs over the input once; empty descriptors take no slot of a chunk
    while (k < n_tensors) {
        CompactArgs a;
        a.n_tensors = 0;
        for (; k < n_tensors && a.n_tensors < COMPACT_MAX_TENSORS; k++) {
            if (t[k].row_floats <= 0) continue;
            a.t[a.n_tensors].src = t[k].src; a.t[a.n_tensors].dst = t[k].dst; a.t[a.n_tensors].row_floats = (int)t[k].row_floats;
            a.n_tensors++;
        }
        if (a.n_tensors > 0)
            hipLaunchKernelGGL(compact_rows_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, keep, block_offsets, a);
    }
}

size_t gsr_mse_scratch_bytes_impl(int rows) { return (size_t)rows * MSE_PARTS * sizeof(double); }

void gsr_launch_mse(int rows, long long n, const float* a, const float* b, float* out, double* partials, hipStream_t s)
{
    mse_partial_kernel<<<dim3(MSE_PARTS, rows), 256, 0, s>>>(n, a, b, partials);
    mse_final_kernel<<<rows, 256, 0, s>>>(n, partials, out);
}

size_t gsr_bias_relu_scratch_bytes_impl(int C) { return (size_t)C * BR_PARTS * sizeof(double); }

void gsr_launch_bias_relu_forward(int C, long long n, float* x, const float* bias, hipStream_t s)
{
    const long long blocks = (n + 1023) / 1024;
    bias_relu_fwd_kernel<<<dim3((unsigned)(blocks < 256 ? (blocks > 0 ? blocks : 1) : 256), C), 256, 0, s>>>(n, x, bias);
}

void gsr_launch_bias_relu_backward(int C, long long n, const float* g, const float* y, float* dz, float* db, double* partials, hipStream_t s)
{
    bias_relu_bwd_kernel<<<dim3(BR_PARTS, C), 256, 0, s>>>(n, g, y, dz, partials);
    bias_relu_bwd_final_kernel<<<(C + 63) / 64, 64, 0, s>>>(C, partials, db);
}

