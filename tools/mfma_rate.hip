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

// the issue pattern of app_conv16_kernel: per tap 16 MFMAs (4 accumulators x 4 weight registers), the next tap's 16 B
// operands read from LDS meanwhile.  MODE 0: as in the kernel; 1: no sched_group_barrier; 2: B operands reused (one LDS
// read per tap only); 3: ds_read_b64 pairs
template <int MODE>
__global__ void __launch_bounds__(256) kconv(float* out, int iters)
{
    __shared__ float lds[16 * 688];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, k = lane >> 4;
    for (int i = tid; i < 16 * 688; i += 256) lds[i] = 0.001f * i;
    __syncthreads();
    float wreg[36];
    for (int i = 0; i < 36; i++) wreg[i] = 0.01f * (i + lane);
    const float* bbase = lds + k * 688 + n;
    f32x4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bq[2][16];
    auto load_tap = [&](float(&b)[16], int row, int t) {
        const int dy = t / 3, dx = t % 3;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int s = 0; s < 4; s++) b[j * 4 + s] = bbase[4 * j * 688 + (row + dy) * 66 + s * 16 + dx];
    };
    for (int it = 0; it < iters; it++) {
        load_tap(bq[0], it & 7, 0);
#pragma unroll
        for (int t = 0; t < 9; t++) {
            if (t + 1 < 9) {
                if (MODE == 2) bq[(t + 1) & 1][0] = bbase[(it & 7) * 66 + t];
                else load_tap(bq[(t + 1) & 1], it & 7, t + 1);
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int s = 0; s < 4; s++)
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[t * 4 + j], bq[t & 1][MODE == 2 ? 0 : j * 4 + s], acc[s], 0, 0, 0);
            if (MODE != 1) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MFMAs with independent streaming traffic beside them, in the proportion of the data-gradient conv kernel: per 144 MFMAs a
// wave loads 24 x 256 B and stores 16 x 256 B.  MODE 0: both; 1: MFMAs only; 2: traffic only.  Do they overlap?
template <int MODE>
__global__ void __launch_bounds__(256) kmix(float* out, const float* __restrict__ src, float* __restrict__ dst, int iters)
{
    f32x4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    const size_t wave_id = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    float carry = 0.f;
    for (int it = 0; it < iters; it++) {
        float v[24];
        if (MODE != 1) {
            const float* p = src + ((wave_id * iters + it) * 24) * 64 + lane;
#pragma unroll
            for (int i = 0; i < 24; i++) v[i] = p[i * 64];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep [loads][MFMAs][stores]: the compiler otherwise pulls the stores (and the wait) up
        if (MODE != 2) {
#pragma unroll
            for (int r = 0; r < 36; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 1) {
            float* q = dst + ((wave_id * iters + it) * 16) * 64 + lane;
#pragma unroll
            for (int i = 0; i < 16; i++) q[i * 64] = v[i] + v[(i + 8) % 24] + carry;
            carry = v[23];
        }
    }
    float s = carry;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run_mix(float* out, const float* src, float* dst)
{
    const int iters = 60, blocks = 512;  // 2 waves per SIMD, ~8 "tiles" of 288 MFMAs x 2 per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kmix<MODE><<<blocks, 256>>>(out, src, dst, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kmix<MODE><<<blocks, 256>>>(out, src, dst, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 4 * iters * (24 + 16) * 256;
    printf("mix mode %d: %.3f ms  (%.1f MB moved -> %.2f TB/s;  %.1f cycles per MFMA per SIMD at 2.4 GHz)\n", MODE, ms, bytes / 1e6,
           MODE == 1 ? 0.0 : bytes / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (iters * 144.0 * 2));
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
    for (int w = 1; w <= 2; w++) {
        run("conv m0", kconv<0>, 9, 2048, w, out);  // 144 MFMAs per iteration = 16 x 9
        run("conv m1", kconv<1>, 9, 2048, w, out);
        run("conv m2", kconv<2>, 9, 2048, w, out);
    }
    {
        float *src, *dst;
        const size_t n_src = (size_t)512 * 4 * 60 * 24 * 64, n_dst = (size_t)512 * 4 * 60 * 16 * 64;
        hipMalloc(&src, n_src * sizeof(float));
        hipMalloc(&dst, n_dst * sizeof(float));
        hipMemset(src, 0, n_src * sizeof(float));
        run_mix<1>(out, src, dst);
        run_mix<2>(out, src, dst);
        run_mix<0>(out, src, dst);
        run_mix<0>(out, src, dst);
    }
    // long run: does the rate hold for ~100 ms?
    for (int rep = 0; rep < 3; rep++) run("16x16x4", k16<4>, 4, 2048, 2, out);
    return 0;
}
