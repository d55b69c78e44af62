// Adam step for all parameter tensors of a GaussianModel in ONE launch (replaces the per-tensor op chains of
// torch.optim.Adam(l, lr=0.0, eps=1e-15).step(), scene/gaussian_model.py:164-175, train.py:155-157).
// Dense semantics, exactly torch's `_single_tensor_adam` with weight_decay = 0, amsgrad = False, maximize = False:
//   exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2);
//   denom = exp_avg_sq.sqrt() / sqrt(1 - beta2^step) + eps; param.addcdiv_(exp_avg, denom, value = -lr / (1 - beta1^step))
// in the arithmetic torch's own kernels use (pinned by tests/test_adam.py against torch on CPU):
//   m = fma(w, g - m, m);  v = fma((1-beta2)*g, g, v*beta2);  p = p + (ss*m) / (sqrt(v)/bs + eps)
// with w, 1-beta2, ss, bs rounded to fp32 from the double-precision Python scalars.
// Pure streaming, HBM-bound: 16 B read + 12 B written per element, 16-byte accesses.
#include "common.h"
#include <math.h>

namespace {

constexpr int ADAM_MAX_TENSORS = 16;
constexpr int ADAM_ELEMS_PER_BLOCK = 256 * 4 * 4;  // 256 threads x 4 float4

struct AdamTensor {
    float* __restrict__ p; const float* __restrict__ g; float* __restrict__ m; float* __restrict__ v;
    long n;
    float w, omb2, b2, ss, bs, eps;  // lerp weight, 1-beta2, beta2, -lr/bc1, sqrt(bc2), eps
    int first_block;
};
struct AdamArgs {
    AdamTensor t[ADAM_MAX_TENSORS];
    int n_tensors;
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamTensor& a)
{
    m = __builtin_fmaf(a.w, g - m, m);
    v = __builtin_fmaf(a.omb2 * g, g, v * a.b2);
    const float denom = sqrtf(v) / a.bs + a.eps;
    p = p + (a.ss * m) / denom;
}

__global__ void __launch_bounds__(256) adam_kernel(AdamArgs args)
{
    int ti = 0;
#pragma unroll
    for (int k = 1; k < ADAM_MAX_TENSORS; k++)
        if (k < args.n_tensors && (int)blockIdx.x >= args.t[k].first_block) ti = k;
    const AdamTensor& a = args.t[ti];
    const long base = (long)((int)blockIdx.x - a.first_block) * ADAM_ELEMS_PER_BLOCK;
    const long n = a.n;
    const bool vec = (((uintptr_t)a.p | (uintptr_t)a.g | (uintptr_t)a.m | (uintptr_t)a.v) & 15) == 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const long i = base + ((long)r * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        if (vec && i + 4 <= n) {
            float4 p = *reinterpret_cast<float4*>(a.p + i);
            const float4 g = *reinterpret_cast<const float4*>(a.g + i);
            float4 m = *reinterpret_cast<float4*>(a.m + i);
            float4 v = *reinterpret_cast<float4*>(a.v + i);
            adam_elem(p.x, g.x, m.x, v.x, a); adam_elem(p.y, g.y, m.y, v.y, a);
            adam_elem(p.z, g.z, m.z, v.z, a); adam_elem(p.w, g.w, m.w, v.w, a);
            *reinterpret_cast<float4*>(a.p + i) = p;
            *reinterpret_cast<float4*>(a.m + i) = m;
            *reinterpret_cast<float4*>(a.v + i) = v;
        } else {
            for (long e = i; e < min(i + 4, n); e++) {
                float p = a.p[e], m = a.m[e], v = a.v[e];
                adam_elem(p, a.g[e], m, v, a);
                a.p[e] = p; a.m[e] = m; a.v[e] = v;
            }
        }
    }
}

}  // namespace

int gsr_launch_adam(int n_tensors, const GsrAdamTensorHost* t, double beta1, double beta2, double eps, hipStream_t s)
{
    // `k` runs over the input once: a chunk ends when ADAM_MAX_TENSORS non-empty tensors are packed, and the next chunk
    // continues at the first entry not yet consumed (empty tensors -- e.g. features_rest at sh_degree 0 -- take no slot).
    int k = 0;
    while (k < n_tensors) {
        AdamArgs a;
        a.n_tensors = 0;
        int blocks = 0;
        for (; k < n_tensors && a.n_tensors < ADAM_MAX_TENSORS; k++) {
            if (t[k].n <= 0) continue;
            AdamTensor& d = a.t[a.n_tensors++];
            d.p = t[k].param; d.g = t[k].grad; d.m = t[k].exp_avg; d.v = t[k].exp_avg_sq; d.n = (long)t[k].n;
            // the Python scalars of torch/optim/adam.py, in double, rounded to fp32 where torch hands them to its kernels
            const double bc1 = 1.0 - pow(beta1, (double)t[k].step), bc2 = 1.0 - pow(beta2, (double)t[k].step);
            d.w = (float)(1.0 - beta1); d.omb2 = (float)(1.0 - beta2); d.b2 = (float)beta2;
            d.ss = (float)(-(t[k].lr / bc1)); d.bs = (float)sqrt(bc2); d.eps = (float)eps;
            d.first_block = blocks;
            blocks += (int)((t[k].n + ADAM_ELEMS_PER_BLOCK - 1) / ADAM_ELEMS_PER_BLOCK);
        }
        if (blocks > 0) hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, a);
    }
    return 0;
}
