// Micro-benchmark (round 5): issue cost of scalar instructions and of the VALU instructions that move data between the scalar and the
// vector side, per SIMD with 8 waves resident (the occupancy of the blend kernels).  ns per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/salu_probe tools/salu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    float2 p0 = {a, 1.f}, p1 = {1.0001f, 0.999f}, p2 = {0.5f, 0.25f}, p3 = {0.125f, 0.3f};
    unsigned long long q0 = threadIdx.x, q1 = threadIdx.x * 7ull;
    uint32_t va = threadIdx.x, vb = threadIdx.x * 3u;
    unsigned long long s0 = 0x1234567ull + blockIdx.x, s1 = 0xFFFF0000FFFFull, s2 = 0xF0F0F0F0ull, s3 = 0xFF00FF00FFull;
    uint32_t t0 = blockIdx.x + 1, t1 = 77, t2 = 5, t3 = 9;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) asm volatile(R16("s_and_b32 %0, %0, %1\n s_or_b32 %1, %1, %2\n s_and_b32 %2, %2, %3\n s_or_b32 %3, %3, %0\n") : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(t3) : : "scc");
        else if (KIND == 1) asm volatile(R16("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n") : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(t3) : : "scc");
        else if (KIND == 2) asm volatile(R16("s_cmp_eq_u64 %0, %1\n s_cmp_eq_u64 %1, %2\n s_cmp_eq_u64 %2, %3\n s_cmp_eq_u64 %3, %0\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        else if (KIND == 3) asm volatile(R16("s_ff1_i32_b64 %0, %4\n s_ff1_i32_b64 %1, %5\n s_ff1_i32_b64 %2, %6\n s_ff1_i32_b64 %3, %7\n") : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(t3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        else if (KIND == 4) asm volatile(R16("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n 1:\n s_cmp_eq_u32 %1, 0\n s_cbranch_scc1 2f\n 2:\n") : "+s"(t0), "+s"(t1) : : "scc");      // not-taken branches (t != 0): 2 cmp + 2 branch
        else if (KIND == 5) asm volatile(R16("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n 1:\n s_cmp_lg_u32 %1, 0\n s_cbranch_scc1 2f\n 2:\n") : "+s"(t0), "+s"(t1) : : "scc");      // taken branches (to the next instruction)
        else if (KIND == 6) asm volatile(R16("s_mov_b64 %0, %1\n s_mov_b64 %1, %2\n s_mov_b64 %2, %3\n s_mov_b64 %3, %0\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
        else if (KIND == 7) asm volatile(R16("s_cselect_b64 %0, %1, %2\n s_cselect_b64 %1, %2, %3\n s_cselect_b64 %2, %3, %0\n s_cselect_b64 %3, %0, %1\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
        else if (KIND == 8) asm volatile(R16("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"));
        else if (KIND == 9) asm volatile(R16("v_readfirstlane_b32 %0, %4\n v_readfirstlane_b32 %1, %5\n v_readfirstlane_b32 %2, %4\n v_readfirstlane_b32 %3, %5\n") : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(t3), "+v"(va), "+v"(vb));
        else if (KIND == 10) asm volatile(R16("v_writelane_b32 %0, %2, 3\n v_writelane_b32 %1, %3, 5\n v_writelane_b32 %0, %4, 7\n v_writelane_b32 %1, %5, 9\n") : "+v"(va), "+v"(vb) : "s"(t0), "s"(t1), "s"(t2), "s"(t3));
        else if (KIND == 11) asm volatile(R16("v_or_b32 %0, %2, %0\n v_or_b32 %1, %3, %1\n v_or_b32 %0, %4, %0\n v_or_b32 %1, %5, %1\n") : "+v"(va), "+v"(vb) : "s"(t0), "s"(t1), "s"(t2), "s"(t3));
        else if (KIND == 12) asm volatile(R16("v_mov_b32 %0, %2\n v_mov_b32 %1, %3\n v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n") : "+v"(va), "+v"(vb) : "s"(t0), "s"(t1), "s"(t2), "s"(t3));
        else if (KIND == 13) asm volatile(R16("v_addc_co_u32 %0, vcc, %0, %0, %2\n v_addc_co_u32 %1, vcc, %1, %1, %3\n v_addc_co_u32 %0, vcc, %0, %0, %3\n v_addc_co_u32 %1, vcc, %1, %1, %2\n") : "+v"(va), "+v"(vb) : "s"(s0), "s"(s1) : "vcc");
        else if (KIND == 14) asm volatile(R16("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0\n v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n") : "+v"(va), "+v"(vb));
        else if (KIND == 15) asm volatile(R16("v_cmp_ne_u32 %2, 0, %0\n v_cmp_ne_u32 %3, 0, %1\n v_cmp_ne_u32 %2, 1, %0\n v_cmp_ne_u32 %3, 1, %1\n") : "+v"(va), "+v"(vb), "+s"(s0), "+s"(s1));
        else if (KIND == 16) asm volatile(R16("s_and_saveexec_b64 %0, %1\n s_or_b64 exec, exec, %0\n s_and_saveexec_b64 %0, %1\n s_or_b64 exec, exec, %0\n") : "+s"(s0) : "s"(~0ull) : "scc");
        else if (KIND == 17) asm volatile(R16("s_bitcmp1_b64 %0, 3\n s_bitcmp1_b64 %1, 5\n s_bitset0_b64 %2, 7\n s_bitset0_b64 %3, 9\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        else if (KIND == 18) asm volatile(R16("s_lshl_b64 %0, %1, 3\n s_lshl_b64 %1, %2, 3\n s_lshl_b64 %2, %3, 3\n s_lshl_b64 %3, %0, 3\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        else if (KIND == 19) asm volatile(R16("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (KIND == 20) asm volatile(R16("v_cndmask_b32 %0, 0, %1, %4\n v_cndmask_b32 %1, 0, %2, %4\n v_cndmask_b32 %2, 0, %3, %4\n v_cndmask_b32 %3, 0, %0, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s1));
        else if (KIND == 21) asm volatile(R16("v_bfe_u32 %0, %1, %2, 1\n v_bfe_u32 %1, %0, %3, 1\n v_and_b32 %0, %2, %1\n v_and_b32 %1, %3, %0\n") : "+v"(va), "+v"(vb) : "s"(t0), "s"(t1));
        else if (KIND == 22) asm volatile(R16("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        else if (KIND == 23) asm volatile(R16("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        else if (KIND == 24) asm volatile(R16("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        else if (KIND == 25) asm volatile(R16("v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xa\n v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (KIND == 26) asm volatile(R16("v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shl:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (KIND == 27) asm volatile(R16("v_add_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (KIND == 28) asm volatile(R16("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (KIND == 30) asm volatile(R16("v_mad_u64_u32 %0, vcc, %2, 36, %0\n v_mad_u64_u32 %1, vcc, %3, 36, %1\n v_mad_u64_u32 %0, vcc, %2, 36, %0\n v_mad_u64_u32 %1, vcc, %3, 36, %1\n") : "+v"(q0), "+v"(q1) : "s"(t0), "s"(t1) : "vcc");
        else if (KIND == 31) asm volatile(R16("v_mul_legacy_f32 %0, %0, %1\n v_mul_legacy_f32 %1, %1, %2\n v_mul_legacy_f32 %2, %2, %3\n v_mul_legacy_f32 %3, %3, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (KIND == 32) asm volatile(R16("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (KIND == 33) asm volatile(R16("v_fmaak_f32 %0, %0, %1, 0x3d635a47\n v_fmaak_f32 %1, %1, %2, 0x3d635a47\n v_fmaak_f32 %2, %2, %3, 0x3d635a47\n v_fmaak_f32 %3, %3, %0, 0x3d635a47\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (KIND == 34) asm volatile(R16("v_mul_f32_e64 %0, %0, -%1\n v_mul_f32_e64 %1, %1, -%2\n v_mul_f32_e64 %2, %2, -%3\n v_mul_f32_e64 %3, %3, -%0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (KIND == 35) asm volatile(R16("v_mul_f32 %0, 0x3fb8aa3b, %1\n v_mul_f32 %1, 0x3fb8aa3b, %2\n v_mul_f32 %2, 0x3fb8aa3b, %3\n v_mul_f32 %3, 0x3fb8aa3b, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (KIND == 36) asm volatile(R16("v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %1, %2, %3, %0\n v_fma_f32 %2, %3, %0, %1\n v_fma_f32 %3, %0, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (KIND == 37) asm volatile(R16("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (KIND == 29) asm volatile(R16("v_pk_mov_b32 %0, %1, %2\n v_pk_mov_b32 %1, %2, %3\n v_pk_mov_b32 %2, %3, %0\n v_pk_mov_b32 %3, %0, %1\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    }
    if (p0.x + p1.y + p2.x + p3.y == 777.5f) out[1] = p0.x;
    if (q0 + q1 == 0x123456789abcull) out[2] = 1.f;
    if (a + b + c + d == 12345.678f || (s0 ^ s1 ^ s2 ^ s3) == 0x123456789ull || t0 + t1 + t2 + t3 == 0x7777777u || va + vb == 0x3333333u) out[0] = a;
}

template <int KIND>
static void run(const char* name, int per_iter, float* sink)
{
    const int iters = 500, wps = 8, blocks = 256 * wps;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, sink, iters);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, sink, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-44s %8.4f ms  %6.2f ns per wave-instruction per SIMD (8 waves/SIMD)\n", name, best, 1e6 * best / ((double)iters * per_iter * wps));
    fflush(stdout);
}

int main()
{
    float* sink;
    CK(hipMalloc(&sink, 64));
    run<19>("v_fma_f32 (reference)", 64, sink);
    run<0>("s_and_b32 / s_or_b32", 64, sink);
    run<1>("s_add_u32", 64, sink);
    run<2>("s_cmp_eq_u64", 64, sink);
    run<3>("s_ff1_i32_b64", 64, sink);
    run<4>("s_cmp + s_cbranch not taken (per instr)", 64, sink);
    run<5>("s_cmp + s_cbranch taken (per instr)", 64, sink);
    run<6>("s_mov_b64", 64, sink);
    run<7>("s_cselect_b64", 64, sink);
    run<8>("s_nop 0", 64, sink);
    run<16>("s_and_saveexec_b64 / s_or_b64 exec", 64, sink);
    run<17>("s_bitcmp1_b64 / s_bitset0_b64", 64, sink);
    run<18>("s_lshl_b64", 64, sink);
    run<9>("v_readfirstlane_b32", 64, sink);
    run<10>("v_writelane_b32", 64, sink);
    run<11>("v_or_b32 sgpr operand", 64, sink);
    run<12>("v_mov_b32 from sgpr", 64, sink);
    run<13>("v_addc_co_u32 carry-in sgpr pair", 64, sink);
    run<14>("v_add_u32 / v_lshlrev_b32", 64, sink);
    run<15>("v_cmp_ne_u32 -> sgpr pair", 64, sink);
    run<20>("v_cndmask_b32 0, v, sgpr mask", 64, sink);
    run<21>("v_bfe_u32 sgpr pos / v_and_b32 sgpr", 64, sink);
    run<22>("v_pk_fma_f32", 64, sink);
    run<23>("v_pk_mul_f32", 64, sink);
    run<24>("v_pk_add_f32", 64, sink);
    run<25>("v_add_f32_dpp (bank-masked, row shifts)", 64, sink);
    run<26>("v_mov_b32_dpp", 64, sink);
    run<27>("v_add_f32 / v_mul_f32", 64, sink);
    run<28>("v_mov_b32 vgpr", 64, sink);
    run<29>("v_pk_mov_b32", 64, sink);
    run<30>("v_mad_u64_u32 (sgpr x const + vgpr pair)", 64, sink);
    run<31>("v_mul_legacy_f32", 64, sink);
    run<19>("v_fma_f32 (again, warm clocks)", 64, sink);
    run<32>("v_fmac_f32 (VOP2)", 64, sink);
    run<33>("v_fmaak_f32 (VOP2 + 32-bit literal)", 64, sink);
    run<34>("v_mul_f32_e64 (VOP3, neg modifier)", 64, sink);
    run<35>("v_mul_f32 with a 32-bit literal", 64, sink);
    run<36>("v_fma_f32 three distinct sources", 64, sink);
    run<37>("v_fma_f32 two distinct sources", 64, sink);
    run<27>("v_add_f32 / v_mul_f32 (again)", 64, sink);
    return 0;
}
