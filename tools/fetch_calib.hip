// Calibration of rocprofv3's FETCH_SIZE for the access shapes of the blend kernels (MI355X_MICROARCH.md §HBM: on gfx950 the
// counter reads exactly half of the bytes of a wide coalesced streaming read; "other access widths are uncalibrated: calibrate
// on a known byte count in your own access pattern").  Three kernels over a table far larger than L2 + Infinity Cache, every
// byte / record touched exactly once:
//   stream16   16 B per lane, consecutive lanes consecutive addresses (the calibrated case)
//   gather48   one 48-byte record per lane as three 16-byte loads, records in random order (the blend kernels' splat gathers)
//   gather16   one 16-byte piece per lane at a random 48-byte record (a single quarter of the record)
// Run under `rocprofv3 --pmc FETCH_SIZE` and compare the counter with the byte counts printed here.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) stream16(const float4* __restrict__ t, size_t n4, float* __restrict__ out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = t[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

__global__ void __launch_bounds__(256) gather48(const float* __restrict__ t, const uint32_t* __restrict__ idx, size_t n, float* __restrict__ out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4* r = reinterpret_cast<const float4*>(t + (size_t)idx[i] * 12);
        const float4 a = r[0], b = r[1], c = r[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 123.456f) out[0] = acc;
}

__global__ void __launch_bounds__(256) gather16(const float* __restrict__ t, const uint32_t* __restrict__ idx, size_t n, float* __restrict__ out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 a = *reinterpret_cast<const float4*>(t + (size_t)idx[i] * 12);
        acc += a.x;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main()
{
    const size_t n_rec = 32u << 20;            // 32 Mi records x 48 B = 1.5 GiB
    const size_t bytes = n_rec * 48;
    float* t; uint32_t* idx; float* out;
    CHECK(hipMalloc(&t, bytes));
    CHECK(hipMalloc(&idx, n_rec * 4));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(t, 0, bytes));
    std::vector<uint32_t> h(n_rec);
    std::iota(h.begin(), h.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(h.begin(), h.end(), rng);
    CHECK(hipMemcpy(idx, h.data(), n_rec * 4, hipMemcpyHostToDevice));
    // distinct 64-byte and 128-byte lines under the records: all of them (the records tile the table), so a gather of all
    // records in any order touches every byte of the table once -- what differs between the kernels is how often a line is fetched
    printf("table %.1f MiB = %zu records of 48 B; index list %.1f MiB\n", bytes / 1048576.0, n_rec, n_rec * 4 / 1048576.0);
    printf("known bytes: stream16 %.1f MiB | gather48: %.1f MiB of records + %.1f MiB of indices | gather16: %.1f MiB requested (16 of every 48 B) + indices\n",
           bytes / 1048576.0, bytes / 1048576.0, n_rec * 4 / 1048576.0, n_rec * 16 / 1048576.0);
    // a 48-byte record at offset 48*i touches these many 64-B / 128-B lines on average (period 384 B = 8 records)
    int l64 = 0, l128 = 0, q64 = 0;
    for (int i = 0; i < 8; i++) {
        const int o = 48 * i;
        l64 += (o + 47) / 64 - o / 64 + 1;
        l128 += (o + 47) / 128 - o / 128 + 1;
        q64 += 1;  // the first 16-byte piece lies in one 64-byte line
    }
    printf("lines per record: %.3f x 64 B = %.1f B, %.3f x 128 B = %.1f B; first 16-byte piece: 1 x 64 B\n", l64 / 8.0, l64 * 8.0, l128 / 8.0, l128 * 16.0);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
        float ms;
        hipEventRecord(a); hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const float4*)t, bytes / 16, out); hipEventRecord(b);
        CHECK(hipEventSynchronize(b)); hipEventElapsedTime(&ms, a, b);
        printf("stream16 %.3f ms (%.2f TB/s)\n", ms, bytes / ms / 1e9);
        hipEventRecord(a); hipLaunchKernelGGL(gather48, dim3(4096), dim3(256), 0, 0, t, idx, n_rec, out); hipEventRecord(b);
        CHECK(hipEventSynchronize(b)); hipEventElapsedTime(&ms, a, b);
        printf("gather48 %.3f ms (%.2f TB/s of record bytes)\n", ms, bytes / ms / 1e9);
        hipEventRecord(a); hipLaunchKernelGGL(gather16, dim3(4096), dim3(256), 0, 0, t, idx, n_rec, out); hipEventRecord(b);
        CHECK(hipEventSynchronize(b)); hipEventElapsedTime(&ms, a, b);
        printf("gather16 %.3f ms\n", ms);
    }
    return 0;
}
