// Micro-benchmark (round 5): scalar-memory stores and SALU issue on gfx950.
//  A. do s_store_dwordx4 + s_dcache_wb work, and what do they cost beside VALU work?  A wave writes `nrec` 16-byte records to its own
//     region, one per `gap` FMA groups -- the pattern of a blend forward that logs (entry, lane mask) per contributing visit.
//  B. the same records read back with s_load_dwordx4 in a dependent walk (the blend backward's consumption pattern).
//  C. SALU issue rate next to VALU work: cycles per s_and_b64 with 1..8 waves per SIMD, alone and interleaved with v_fma_f32.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/smem_probe tools/smem_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 4: the s_store of MODE 1 with the records of all waves INTERLEAVED (record i of wave w at rec[i * waves + w], 8 bytes each half by
// two s_store_dwordx2): neighbouring 8-byte slots of one 64-byte line are written by waves of different CUs -- checks that the scalar
// cache writes back dirty BYTES, not lines.
// MODE 0: FMA work only; 1: + one s_store_dwordx4 per gap; 2: + the same record by a one-lane vector store (global_store_dwordx4 under
// exec = lane 0); 3: + the same record as a vector store by ALL lanes to the same address
template <int MODE>
__global__ void __launch_bounds__(256) k_store(u32x4* __restrict__ rec, float* __restrict__ sink, int nrec, int gap)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
    u32x4* mine = rec + (size_t)wave * nrec;
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < nrec; i++) {
        for (int g = 0; g < gap; g++)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        const unsigned long long m = __builtin_amdgcn_ballot_w64(a > -1e30f);  // some lane mask (all ones here unless NaN)
        u32x4 r;
        r.x = (uint32_t)i; r.y = (uint32_t)m; r.z = (uint32_t)(m >> 32) ^ wave; r.w = 0x5EC0DE00u + (uint32_t)i;
        if (MODE == 1) {
            const uint32_t off = (uint32_t)i * 16u;
            asm volatile("s_store_dwordx4 %0, %1, %2" ::"s"(r), "s"(mine), "s"(off) : "memory");
        } else if (MODE == 4) {
            const uint32_t waves = gridDim.x * 4u;
            const uint32_t off = ((uint32_t)i * waves + wave) * 16u;
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            u32x2 lo, hi;
            lo.x = r.x; lo.y = r.y; hi.x = r.z; hi.y = r.w;
            const uint32_t off2 = off + 8u;
            asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(lo), "s"(rec), "s"(off) : "memory");
            asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(hi), "s"(rec), "s"(off2) : "memory");
        } else if (MODE == 2) {
            if ((threadIdx.x & 63) == 0) mine[i] = r;
        } else if (MODE == 3) {
            mine[i] = r;
        }
    }
    if (MODE == 1 || MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)\n s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");
    if (a + b + c + d == 12345.678f) sink[0] = a;
}

// reads the records back with scalar loads in a walk whose next address depends on nothing but the counter (prefetchable), sums a field
__global__ void __launch_bounds__(256) k_load(const u32x4* __restrict__ rec, uint32_t* __restrict__ out, int nrec, int gap)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
    const u32x4* mine = rec + (size_t)wave * nrec;
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    uint32_t acc = 0;
    u32x4 cur;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(cur) : "s"(mine) : "memory");
    for (int i = 0; i < nrec; i++) {
        u32x4 nxt;
        const uint32_t off = (uint32_t)(i + 1 < nrec ? i + 1 : i) * 16u;
        asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(nxt) : "s"(mine), "s"(off) : "memory");
        for (int g = 0; g < gap; g++)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        acc += cur.x + (cur.w & 0xFFu);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        cur = nxt;
    }
    if ((threadIdx.x & 63) == 0) out[wave] = acc + (a + b + c + d == 12345.678f ? 1u : 0u);
}

// C. SALU issue: KIND 0: 64 dependent-free s_and_b64 per iteration; 1: 32 v_fma_f32; 2: 32 v_fma + 32 s_and interleaved; 3: 32 v_fma + 64 s_and
template <int KIND>
__global__ void __launch_bounds__(256) k_salu(float* out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    unsigned long long s0 = 0x1234567ull + blockIdx.x, s1 = 0xFFFF0000FFFFull, s2 = 0xF0F0F0F0ull, s3 = 0xFF00FF00FFull;
#define V4 "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n"
#define S4 "s_and_b64 %4, %4, %5\n s_or_b64 %5, %5, %6\n s_and_b64 %6, %6, %7\n s_or_b64 %7, %7, %4\n"
#define VS4 "v_fma_f32 %0, %0, %1, %2\n s_and_b64 %4, %4, %5\n v_fma_f32 %1, %1, %2, %3\n s_or_b64 %5, %5, %6\n v_fma_f32 %2, %2, %3, %0\n s_and_b64 %6, %6, %7\n v_fma_f32 %3, %3, %0, %1\n s_or_b64 %7, %7, %4\n"
#define VSS4 "v_fma_f32 %0, %0, %1, %2\n s_and_b64 %4, %4, %5\n s_or_b64 %5, %5, %6\n v_fma_f32 %1, %1, %2, %3\n s_and_b64 %6, %6, %7\n s_or_b64 %7, %7, %4\n v_fma_f32 %2, %2, %3, %0\n s_and_b64 %4, %4, %5\n s_or_b64 %5, %5, %6\n v_fma_f32 %3, %3, %0, %1\n s_and_b64 %6, %6, %7\n s_or_b64 %7, %7, %4\n"
#define R8(x) x x x x x x x x
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) asm volatile(R8(S4) R8(S4) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        else if (KIND == 1) asm volatile(R8(V4) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        else if (KIND == 2) asm volatile(R8(VS4) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        else asm volatile(R8(VSS4) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
    }
    if (a + b + c + d == 12345.678f || (s0 ^ s1 ^ s2 ^ s3) == 0x123456789ull) out[0] = a;
}

template <typename F>
static float time_ms(F launch, int reps = 5)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const int CUS = 256, nrec = 256;
    float* sink;
    CK(hipMalloc(&sink, 64));
    for (int wps = 2; wps <= 8; wps += 2) {   // waves per SIMD = workgroups of 4 waves per CU
        const int blocks = CUS * wps, waves = blocks * 4;
        u32x4* rec;
        uint32_t* out;
        CK(hipMalloc(&rec, (size_t)waves * nrec * 16));
        CK(hipMalloc(&out, (size_t)waves * 4));
        for (int gap = 1; gap <= 4; gap *= 2) {
            CK(hipMemset(rec, 0, (size_t)waves * nrec * 16));
            const float t0 = time_ms([&] { hipLaunchKernelGGL(k_store<0>, dim3(blocks), dim3(256), 0, 0, rec, sink, nrec, gap); });
            const float t1 = time_ms([&] { hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(256), 0, 0, rec, sink, nrec, gap); });
            // verify the scalar stores from the host
            std::vector<uint32_t> h((size_t)waves * nrec * 4);
            CK(hipMemcpy(h.data(), rec, h.size() * 4, hipMemcpyDeviceToHost));
            long bad = 0;
            for (int w = 0; w < waves; w++)
                for (int i = 0; i < nrec; i++) {
                    const uint32_t* r = &h[((size_t)w * nrec + i) * 4];
                    if (r[0] != (uint32_t)i || r[1] != 0xFFFFFFFFu || r[2] != (0xFFFFFFFFu ^ (uint32_t)w) || r[3] != 0x5EC0DE00u + (uint32_t)i) bad++;
                }
            const float tl = time_ms([&] { hipLaunchKernelGGL(k_load, dim3(blocks), dim3(256), 0, 0, rec, out, nrec, gap); });
            std::vector<uint32_t> ho(waves);
            CK(hipMemcpy(ho.data(), out, (size_t)waves * 4, hipMemcpyDeviceToHost));
            long badl = 0;
            uint32_t want = 0;
            for (int i = 0; i < nrec; i++) want += (uint32_t)i + (uint32_t)(i & 0xFF);
            for (int w = 0; w < waves; w++) if (ho[w] != want) badl++;
            CK(hipMemset(rec, 0xEE, (size_t)waves * nrec * 16));
            const float t4 = time_ms([&] { hipLaunchKernelGGL(k_store<4>, dim3(blocks), dim3(256), 0, 0, rec, sink, nrec, gap); }, 2);
            CK(hipMemcpy(h.data(), rec, h.size() * 4, hipMemcpyDeviceToHost));
            long bad4 = 0;
            for (int w = 0; w < waves; w++)
                for (int i = 0; i < nrec; i++) {
                    const uint32_t* r = &h[((size_t)i * waves + w) * 4];
                    if (r[0] != (uint32_t)i || r[1] != 0xFFFFFFFFu || r[2] != (0xFFFFFFFFu ^ (uint32_t)w) || r[3] != 0x5EC0DE00u + (uint32_t)i) bad4++;
                }
            printf("   interleaved 2 x s_store_dwordx2 per record: %.4f ms, bad records %ld of %ld\n", t4, bad4, (long)waves * nrec);
            const float t2 = time_ms([&] { hipLaunchKernelGGL(k_store<2>, dim3(blocks), dim3(256), 0, 0, rec, sink, nrec, gap); });
            const float t3 = time_ms([&] { hipLaunchKernelGGL(k_store<3>, dim3(blocks), dim3(256), 0, 0, rec, sink, nrec, gap); });
            printf("waves/SIMD %d  fma groups per record %d (%d v_fma): fma only %.4f ms | + s_store_dwordx4 %.4f ms (bad records %ld of %ld) | + one-lane vector store %.4f | "
                   "+ all-lane vector store %.4f | s_load walk %.4f ms (bad sums %ld)  -> s_store costs %.1f ns per record per CU-wave-slot\n",
                   wps, gap, 8 * gap, t0, t1, bad, (long)waves * nrec, t2, t3, tl, badl, 1e6 * (t1 - t0) / nrec);
            fflush(stdout);
        }
        CK(hipFree(rec)); CK(hipFree(out));
    }
    const int iters = 2000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = CUS * wps;
        const float a = time_ms([&] { hipLaunchKernelGGL(k_salu<0>, dim3(blocks), dim3(256), 0, 0, sink, iters); });
        const float b = time_ms([&] { hipLaunchKernelGGL(k_salu<1>, dim3(blocks), dim3(256), 0, 0, sink, iters); });
        const float c = time_ms([&] { hipLaunchKernelGGL(k_salu<2>, dim3(blocks), dim3(256), 0, 0, sink, iters); });
        const float d = time_ms([&] { hipLaunchKernelGGL(k_salu<3>, dim3(blocks), dim3(256), 0, 0, sink, iters); });
        // per SIMD: wps waves, each iters * n instructions
        printf("waves/SIMD %d: 64 s_and/or per iter %.4f ms = %.2f ns per SALU instr per SIMD | 32 v_fma %.4f ms = %.2f ns per VALU | 32 v_fma + 32 salu interleaved %.4f ms | "
               "32 v_fma + 64 salu %.4f ms\n", wps, a, 1e6 * a / (iters * 64.0 * wps), b, 1e6 * b / (iters * 32.0 * wps), c, d);
        fflush(stdout);
    }
    return 0;
}
