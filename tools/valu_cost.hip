// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the cross-lane instructions the blend backward
// uses, measured with all SIMDs loaded by `waves` waves each.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters)
{
    float a = threadIdx.x * 1.0f, b = a + 1.f, c = a + 2.f, d = a + 3.f, e = a + 4.f, f = a + 5.f, g = a + 6.f, h = a + 7.f;
    unsigned long long m = 0xAAAAAAAAAAAAAAAAull;
    int mi = 0;
    unsigned long long dd0 = threadIdx.x, dd1 = threadIdx.x + 7;
    uint32_t mi0 = threadIdx.x, mi1 = threadIdx.x * 3;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {
            asm volatile(REP16("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 1) {
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 2) {
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                               "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 3) {
            asm volatile(REP16("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n"
                               "v_permlane32_swap_b32 %6, %7\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        } else if (KIND == 4) {
            asm volatile(REP16("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n"
                               "v_permlane16_swap_b32 %6, %7\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        } else if (KIND == 5) {
            asm volatile(REP16("v_cndmask_b32 %0, %1, %2, %4\n v_cndmask_b32 %1, %2, %3, %4\n v_cndmask_b32 %2, %3, %0, %4\n"
                               "v_cndmask_b32 %3, %0, %1, %4\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(m));
        } else if (KIND == 6) {
            asm volatile(REP16("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 7) {
            asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 8) {
            asm volatile(REP16("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 9) {
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xa\n"
                               "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa\n"
                               "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n"
                               "v_add_f32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xa\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 10) {
            asm volatile(REP16("v_cndmask_b32_e32 %0, %1, %2, vcc\n v_cndmask_b32_e32 %1, %2, %3, vcc\n v_cndmask_b32_e32 %2, %3, %0, vcc\n"
                               "v_cndmask_b32_e32 %3, %0, %1, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if (KIND == 11) {
            asm volatile(REP16("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %0\n v_med3_f32 %3, %3, %0, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 12) {
            asm volatile(REP16("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cmp_lt_f32_e32 vcc, %1, %2\n v_cmp_lt_f32_e32 vcc, %2, %3\n v_cmp_lt_f32_e32 vcc, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if (KIND == 13) {
            asm volatile(REP16("v_cmp_lt_f32_e64 %4, %0, %1\n v_cmp_lt_f32_e64 %4, %1, %2\n v_cmp_lt_f32_e64 %4, %2, %3\n v_cmp_lt_f32_e64 %4, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(m));
        } else if (KIND == 14) {
            asm volatile(REP16("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 15) {
            asm volatile(REP16("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(1.0001f));
        } else if (KIND == 16) {
            asm volatile(REP16("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 17) {
            asm volatile(REP16("v_mul_f32_e64 %0, %0, -%1\n v_mul_f32_e64 %1, %1, -%2\n v_mul_f32_e64 %2, %2, -%3\n v_mul_f32_e64 %3, %3, -%0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 18) {
            asm volatile(REP16("v_fma_f32 %0, %0, %4, %2\n v_fma_f32 %1, %1, %4, %3\n v_fma_f32 %2, %2, %4, %0\n v_fma_f32 %3, %3, %4, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(1.0001f));
        } else if (KIND == 19) {
            asm volatile(REP16("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 20) {
            asm volatile(REP16("v_cmp_lt_u32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %1, %2, vcc\n v_cmp_lt_u32_e32 vcc, %2, %3\n v_cndmask_b32_e32 %2, %3, %1, vcc\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if (KIND == 21) {
            asm volatile(REP16("v_min_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 22) {
            asm volatile(REP16("v_readfirstlane_b32 %4, %0\n v_readfirstlane_b32 %4, %1\n v_readfirstlane_b32 %4, %2\n v_readfirstlane_b32 %4, %3\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(mi));
        } else if (KIND == 23) {
            asm volatile(REP16("v_mul_f32 %0, 0x3fb8aa3b, %0\n v_mul_f32 %1, 0x3fb8aa3b, %1\n v_mul_f32 %2, 0x3fb8aa3b, %2\n v_mul_f32 %3, 0x3fb8aa3b, %3\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 24) {
            asm volatile(REP16("v_fmaak_f32 %0, %0, %1, 0x3c088908\n v_fmaak_f32 %1, %1, %2, 0x3c088908\n v_fmaak_f32 %2, %2, %3, 0x3c088908\n v_fmaak_f32 %3, %3, %0, 0x3c088908\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 25) {
            asm volatile(REP16("v_mul_f32 %0, 0.5, %0\n v_mul_f32 %1, 2.0, %1\n v_add_f32 %2, 1.0, %2\n v_fma_f32 %3, %3, %0, -1.0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 26) {
            asm volatile(REP16("v_rndne_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_lshl_add_u32 %2, %2, 23, %3\n v_cvt_f32_i32 %3, %3\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 27) {
            asm volatile(REP16("v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n s_nop 0\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0\n s_nop 0\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 28) {
            asm volatile(REP16("v_bfi_b32 %0, %4, %1, %2\n v_bfi_b32 %1, %4, %2, %3\n v_bfi_b32 %2, %4, %3, %0\n v_bfi_b32 %3, %4, %0, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
        } else if (KIND == 29) {
            asm volatile(REP16("v_and_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_and_or_b32 %3, %3, %0, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 30) {
            asm volatile(REP16("v_max3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 31) {
            asm volatile(REP16("v_cmp_lt_f32_e32 vcc, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if (KIND == 32) {
            asm volatile(REP16("v_cndmask_b32 %0, %0, %1, %4\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(m));
        } else if (KIND == 33) {
            asm volatile(REP16("v_cndmask_b32 %0, %0, %1, %4\n v_add_f32 %1, %1, %2\n v_cndmask_b32 %2, %2, %3, %4\n v_add_f32 %3, %3, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(m));
        } else if (KIND == 34) {
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32 %1, %1, %2\n"
                               "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32 %3, %3, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 35) {
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32 %1, %1, %2, %4\n"
                               "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_cndmask_b32 %3, %3, %1, %4\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(m));
        } else if (KIND == 36) {
            asm volatile(REP16("v_exp_f32 %0, %0\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 37) {
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32 %1, %1, %2\n"
                               "v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 38) {
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32 %0, %0, %1\nv_add_f32 %1, %1, %2\nv_add_f32 %2, %2, %3\nv_add_f32 %3, %3, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 39) {
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32 %2, %2, %3\nv_add_f32 %3, %3, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 40) {
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\ns_nop 0\nv_add_f32 %1, %1, %2\nv_add_f32 %2, %2, %3\nv_add_f32 %3, %3, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 41) {
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32 %0, %0, %1\nv_add_f32 %1, %1, %2\nv_add_f32 %2, %2, %3\nv_add_f32 %3, %3, %0\nv_add_f32 %0, %0, %1\nv_add_f32 %1, %1, %2\nv_add_f32 %2, %2, %3\nv_add_f32 %3, %3, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 42) {
            asm volatile(REP16("v_exp_f32 %0, %0\n s_nop 0\nv_add_f32 %1, %1, %2\nv_add_f32 %2, %2, %3\nv_add_f32 %3, %3, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 43) {
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\ns_nop 1\nv_add_f32 %1, %1, %2\nv_add_f32 %2, %2, %3\nv_add_f32 %3, %3, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 44) {
            asm volatile(REP16("v_add_f32 %1, %1, %2\nv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\ns_nop 0\nv_add_f32 %2, %2, %3\nv_add_f32 %3, %3, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 45) {
            asm volatile(REP16("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\ns_nop 0\nv_add_f32 %2, %2, %3\nv_add_f32 %3, %3, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 46) {
            asm volatile(REP16("v_rcp_f32 %0, %0\n s_nop 0\nv_add_f32 %1, %1, %2\nv_add_f32 %2, %2, %3\nv_add_f32 %3, %3, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 48) {
            asm volatile(REP16("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %3 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 49) {
            asm volatile(REP16("v_cndmask_b32_dpp %0, %1, %2, vcc wave_ror:1 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %1, %2, %3, vcc wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                               "v_cndmask_b32_dpp %2, %3, %0, vcc wave_ror:1 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %3, %0, %1, vcc wave_ror:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if (KIND == 50) {
            asm volatile(REP16("v_cndmask_b32_dpp %0, %1, %2, vcc row_ror:1 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %1, %2, %3, vcc row_ror:1 row_mask:0xf bank_mask:0xf\n"
                               "v_cndmask_b32_dpp %2, %3, %0, vcc row_ror:1 row_mask:0xf bank_mask:0xf\n v_cndmask_b32_dpp %3, %0, %1, vcc row_ror:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        } else if (KIND == 51) {
            asm volatile(REP16("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 52) {
            asm volatile(REP16("v_mad_u64_u32 %0, %4, %1, 48, %0\n v_mad_u64_u32 %2, %4, %3, 48, %2\n v_mad_u64_u32 %0, %4, %1, 48, %0\n v_mad_u64_u32 %2, %4, %3, 48, %2\n")
                         : "+v"(dd0), "+v"(mi0), "+v"(dd1), "+v"(mi1), "+s"(m));
        } else if (KIND == 53) {
            // single active lane: one wave-instruction costs the same issue time as with 64 lanes?
            asm volatile("s_mov_b64 %4, exec\n s_mov_b64 exec, 1\n" REP16("v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n v_mov_b32 %2, 0\n v_mov_b32 %3, 0\n") "s_mov_b64 exec, %4\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(m));
        } else if (KIND == 54) {
            asm volatile(REP16("v_mov_b64 %0, 0\n v_mov_b64 %1, 0\n v_mov_b64 %0, 0\n v_mov_b64 %1, 0\n")
                         : "+v"(dd0), "+v"(dd1));
        } else if (KIND == 55) {
            asm volatile(REP16("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0\n v_pk_fma_f32 %0, %0, %1, %0\n v_pk_add_f32 %1, %1, %0\n")
                         : "+v"(dd0), "+v"(dd1));
        } else if (KIND == 56) {
            asm volatile(REP16("v_exp_f32 %0, %0\n v_mul_f32 %1, %1, %2\n v_rcp_f32 %2, %2\n v_mul_f32 %3, %3, %1\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (KIND == 47) {
            asm volatile(REP16("v_exp_f32 %0, %0\nv_add_f32 %1, %1, %2\nv_add_f32 %2, %2, %3\nv_add_f32 %3, %3, %1\nv_add_f32 %1, %1, %2\nv_add_f32 %2, %2, %3\nv_add_f32 %3, %3, %1\nv_add_f32 %1, %1, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + (float)mi + (float)(m & 1) + (float)(dd0 + dd1) + (float)(mi0 + mi1);
}

template <int KIND>
void run(const char* name, float* out, int wgs, int per_rep = 4)
{
    const int iters = 2000;  // 16 * per_rep instructions per iteration per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // every SIMD runs wgs*4/1024 waves one after / beside another
    const double waves_per_simd = wgs * 4.0 / 1024.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * 16.0 * per_rep * waves_per_simd);
    printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at 2.4 GHz, %g waves/SIMD)\n", name, ms, cyc, waves_per_simd);
}

int main()
{
    float* out;
    hipMalloc(&out, 8192 * 256 * 4);
    for (int wgs : {1024, 1536}) {
        run<48>("v_mov_b32_dpp wave_ror:1", out, wgs);
        run<49>("v_cndmask_b32_dpp wave_ror:1", out, wgs);
        run<50>("v_cndmask_b32_dpp row_ror:1", out, wgs);
        run<51>("v_mov_b32_dpp wave_shr:1", out, wgs);
        run<52>("v_mad_u64_u32", out, wgs);
        run<53>("v_mov_b32, one lane active", out, wgs);
        run<54>("v_mov_b64", out, wgs);
        run<55>("v_pk_mul/fma/add_f32", out, wgs);
        run<56>("exp, mul, rcp, mul", out, wgs);
        run<0>("v_add_f32", out, wgs);
        run<7>("v_fma_f32", out, wgs);
        run<5>("v_cndmask_b32 (sgpr mask)", out, wgs);
        run<1>("v_add_f32_dpp quad_perm", out, wgs);
        run<2>("v_add_f32_dpp row_ror:8", out, wgs);
        run<9>("v_add_f32_dpp row_shr bank", out, wgs);
        run<6>("v_mov_b32_dpp quad_perm", out, wgs);
        run<3>("v_permlane32_swap_b32", out, wgs);
        run<4>("v_permlane16_swap_b32", out, wgs);
        run<8>("v_exp_f32", out, wgs);
        run<10>("v_cndmask_b32_e32 (vcc)", out, wgs);
        run<20>("v_cmp_u32 + v_cndmask e32", out, wgs);
        run<11>("v_med3_f32", out, wgs);
        run<12>("v_cmp_lt_f32_e32", out, wgs);
        run<13>("v_cmp_lt_f32_e64 (sgpr dst)", out, wgs);
        run<14>("v_rcp_f32", out, wgs);
        run<15>("v_mul_f32 sgpr operand", out, wgs);
        run<18>("v_fma_f32 sgpr operand", out, wgs);
        run<16>("v_fmac_f32", out, wgs);
        run<17>("v_mul_f32_e64 neg", out, wgs);
        run<19>("v_mov_b32", out, wgs);
        run<21>("min/max/sub/mul mix", out, wgs);
        run<22>("v_readfirstlane_b32", out, wgs);
        run<23>("v_mul_f32 literal", out, wgs);
        run<24>("v_fmaak_f32 (literal K)", out, wgs);
        run<25>("inline constants 0.5/2/1/-1", out, wgs);
        run<26>("rndne/cvt/lshl_add/cvt", out, wgs);
        run<27>("4 v_mul + 2 s_nop 0 (per 4 instr)", out, wgs);
        run<28>("v_bfi_b32 (vgpr mask select)", out, wgs);
        run<29>("and/or/xor/and_or", out, wgs);
        run<30>("max3/min3/mad_u32_u24/add3", out, wgs);
        run<31>("1 v_cmp + 3 v_add", out, wgs);
        run<32>("1 cndmask + 3 add", out, wgs);
        run<33>("2 cndmask + 2 add", out, wgs);
        run<34>("2 dpp + 2 add", out, wgs);
        run<35>("2 dpp + 2 cndmask", out, wgs);
        run<36>("1 exp + 3 add", out, wgs);
        run<37>("1 dpp + 3 add", out, wgs);
        run<38>("4 dpp + 4 add, grouped", out, wgs, 8);
        run<39>("2 dpp + 2 add, grouped", out, wgs, 4);
        run<40>("1 dpp + s_nop + 3 add", out, wgs, 4);
        run<41>("8 dpp + 8 add, grouped", out, wgs, 16);
        run<42>("1 exp + s_nop + 3 add", out, wgs, 4);
        run<43>("1 dpp + s_nop 1 + 3 add", out, wgs, 4);
        run<44>("add, dpp, s_nop, 2 add", out, wgs, 4);
        run<45>("2 dpp + s_nop + 2 add", out, wgs, 4);
        run<46>("1 rcp + s_nop + 3 add", out, wgs, 4);
        run<47>("1 exp + 7 add", out, wgs, 8);
    }
    return 0;
}
