// Micro-benchmark: sustained rate of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 with the whole chip loaded, `waves`
// waves per SIMD and `nacc` independent accumulators, in the issue pattern of csrc/appearance.hip.  Prints cycles per MFMA
// per SIMD at 2.4 GHz and the TFLOP/s.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k16(float* out, int iters)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k32(float* out, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++)
        for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++)
        for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int nacc, int flop_per, int waves_per_simd, float* out)
{
    const int iters = 400, blocks = 256 * waves_per_simd;  // 256 threads = one wave per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma_per_wave = (double)iters * 16 * nacc;
    const double cyc = ms * 1e-3 * 2.4e9 / (n_mfma_per_wave * waves_per_simd);
    const double tf = n_mfma_per_wave * blocks * 4 * flop_per / (ms * 1e-3) / 1e12;
    printf("%-10s nacc=%d waves/SIMD=%d  %.3f ms  %.1f cycles/MFMA/SIMD (at 2.4 GHz)  %.1f TFLOP/s\n", name, nacc, waves_per_simd, ms, cyc, tf);
}

int main()
{
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w = 1; w <= 2; w++) {
        run("16x16x4", k16<1>, 1, 2048, w, out);
        run("16x16x4", k16<2>, 2, 2048, w, out);
        run("16x16x4", k16<4>, 4, 2048, w, out);
        run("16x16x4", k16<9>, 9, 2048, w, out);
        run("32x32x2", k32<1>, 1, 4096, w, out);
        run("32x32x2", k32<2>, 2, 4096, w, out);
        run("32x32x2", k32<4>, 4, 4096, w, out);
    }
    // long run: does the rate hold for ~100 ms?
    for (int rep = 0; rep < 3; rep++) run("16x16x4", k16<4>, 4, 2048, 2, out);
    return 0;
}
