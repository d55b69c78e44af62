// Micro-benchmark for the Gaussian-per-lane ("skewed") backward blend: cycles per pipeline step.
//
// One wave = one stream of (8x8 block, list entry) items.  Lane j holds item 64*e + j of the stream and walks the 64 pixels
// of that item's block, one pixel per step, skewed by its lane index: at step t it is on pixel (t - j) & 63.  The per-pixel
// state (T, accumulated colour behind) moves from lane j-1 to lane j between steps with DPP wave_ror:1, so the pixel meets
// the items in list order; the 9 gradient sums of an item stay in ITS lane's registers (no cross-lane reduction) and are
// stored as one row when the lane switches to its next item (one lane per step).
//
// Synthetic data, fixed list length per block (multiple of 64) -- this measures the step cost only.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/skew_probe.hip -o tools/skew_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Item {  // 64 B staged record
    float mx, my, cx, cy;
    float cz, op, pthr; uint32_t kidx;
    float r, g, b; uint32_t row;
    uint32_t pixb, first, pad0, pad1;
};

// state pass: lane j takes lane j-1's value (lane 0 takes lane 63's), or the init value where first_mask is set (first item of
// a block).  One v_cndmask_b32_dpp per value (VOP2 form: the select mask is VCC).
template <int ROWROR>
__device__ __forceinline__ void pass_state(float& T_i, float& A0, float& A1, float& A2, float T_o, float A0_o, float A1_o, float A2_o,
                                           float T_init, unsigned long long first_mask)
{
    const float zero = 0.f;
    if (ROWROR)
    asm volatile("s_mov_b64 vcc, %9\n s_nop 1\n"
                 "v_cndmask_b32_dpp %0, %4, %8, vcc row_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_dpp %1, %5, %10, vcc row_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_dpp %2, %6, %10, vcc row_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_dpp %3, %7, %10, vcc row_ror:1 row_mask:0xf bank_mask:0xf\n"
                 : "=&v"(T_i), "=&v"(A0), "=&v"(A1), "=&v"(A2)
                 : "v"(T_o), "v"(A0_o), "v"(A1_o), "v"(A2_o), "v"(T_init), "s"(first_mask), "v"(zero)
                 : "vcc");
    else
    asm volatile("s_mov_b64 vcc, %9\n s_nop 1\n"
                 "v_cndmask_b32_dpp %0, %4, %8, vcc wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_dpp %1, %5, %10, vcc wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_dpp %2, %6, %10, vcc wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_dpp %3, %7, %10, vcc wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                 : "=&v"(T_i), "=&v"(A0), "=&v"(A1), "=&v"(A2)
                 : "v"(T_o), "v"(A0_o), "v"(A1_o), "v"(A2_o), "v"(T_init), "s"(first_mask), "v"(zero)
                 : "vcc");
}

template <bool STORE, int V>
__global__ void __launch_bounds__(64) skew_kernel(const Item* __restrict__ items, const float* __restrict__ pixdata, int blocks_per_wave,
                                                  int L, float* __restrict__ rows, int* __restrict__ check)
{
    __shared__ float4 s_pix[3 * 64 * 2];  // 3 slots x 64 pixels x {rx, ry, lastc, Tfin | dL0, dL1, dL2, Tfb}
    __shared__ Item s_stage[64];
    const int lane = threadIdx.x;
    const int wave = blockIdx.x;
    const Item* my_items = items + (size_t)wave * blocks_per_wave * L;
    const int n_items = blocks_per_wave * L;
    const int n_epochs = n_items / 64;

    // params of the lane's current item
    float mx = 1e30f, my = 0.f, cx = 1.f, cy = 0.f, cz = 1.f, op = 0.f, pthr = INFINITY, cr = 0.f, cg = 0.f, cb = 0.f;
    uint32_t kidx = 0, row = 0xFFFFFFFFu, pa = 0;
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; i++) acc[i] = 0.f;
    float T_o = 1.f, A0_o = 0.f, A1_o = 0.f, A2_o = 0.f;
    unsigned long long first_mask = 0ull;
    float4 pn0 = make_float4(0.f, 0.f, 0.f, 0.f), pn1 = pn0;

    // epoch 0 prologue: pixel data of block 0 and items of epoch 0
    {
        const float4* pd = reinterpret_cast<const float4*>(pixdata) + (size_t)(wave * blocks_per_wave) * 128;
        s_pix[lane * 2] = pd[lane * 2];
        s_pix[lane * 2 + 1] = pd[lane * 2 + 1];
        s_stage[lane] = my_items[lane];
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    const int total_steps = n_epochs * 64 + 64;  // + drain
    for (int t = 0; t < total_steps; t++) {
        const int k = t & 63;
        const int e = t >> 6;
        // refill halves (items of epoch e+1) and the next block's pixel data
        if (k == 32 && e + 1 < n_epochs) {
            if (lane < 32) s_stage[lane] = my_items[(e + 1) * 64 + lane];
        }
        if (k == 0 && e > 0 && e < n_epochs) {
            if (lane >= 32) s_stage[lane] = my_items[e * 64 + lane];
            // pixel data of the block that starts in the NEXT epoch (if any)
            const int nb = ((e + 1) * 64) / L;
            if (((e + 1) * 64) % L == 0 && nb < blocks_per_wave) {
                const float4* pd = reinterpret_cast<const float4*>(pixdata) + (size_t)(wave * blocks_per_wave + nb) * 128;
                const int slot = nb % 3;
                s_pix[slot * 128 + lane * 2] = pd[lane * 2];
                s_pix[slot * 128 + lane * 2 + 1] = pd[lane * 2 + 1];
            }
        }
        // lane k switches to its next item (stream item 64*e + k): flush the finished one, load the new one
        if (lane == k) {
            if (STORE && row != 0xFFFFFFFFu) {
                float4* o = reinterpret_cast<float4*>(rows + (size_t)row * 12);
                o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
                o[2] = make_float4(acc[8], 0.f, 0.f, 0.f);
            }
            if (e < n_epochs) {
                const Item it = s_stage[k];
                mx = it.mx; my = it.my; cx = it.cx; cy = it.cy; cz = it.cz; op = it.op; pthr = it.pthr; kidx = it.kidx;
                cr = it.r; cg = it.g; cb = it.b; row = it.row; pa = it.pixb;
                if (!(V & 4)) first_mask = it.first ? (first_mask | (1ull << k)) : (first_mask & ~(1ull << k));
            } else {
                pthr = INFINITY; row = 0xFFFFFFFFu;
            }
            if (!(V & 8)) {
#pragma unroll
            for (int i = 0; i < 9; i++) acc[i] = 0.f;
            }
        }
        if (V & 4) {  // scalar bookkeeping of the first-item mask (block boundaries are known to the scalar unit)
            const unsigned long long bit = 1ull << k;
            first_mask = ((t % L) == 0 && e < n_epochs) ? (first_mask | bit) : (first_mask & ~bit);
        } else
        first_mask = __builtin_amdgcn_readfirstlane((uint32_t)first_mask) |
                     ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(first_mask >> 32)) << 32);
        // pixel of this step
        float4 p0, p1;
        if (V & 2) {  // prefetched during the previous step; the switching lane re-reads its own
            p0 = pn0; p1 = pn1;
            if (lane == k) {
                p0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_pix) + pa);
                p1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_pix) + pa + 16);
            }
            pa += 32;
            pn0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_pix) + pa);
            pn1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_pix) + pa + 16);
        } else {
            p0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_pix) + pa);
            p1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_pix) + pa + 16);
            pa += 32;
        }
        float T_i, A0, A1, A2;
        pass_state<V & 1>(T_i, A0, A1, A2, T_o, A0_o, A1_o, A2_o, p0.w, first_mask);
        const bool act = kidx < __float_as_uint(p0.z);
        const float px = act ? p0.x : 1e30f;
        const float dx = mx - px, dy = my - p0.y;
        const float q = __builtin_fmaf(cx * dx, dx, (cz * dy) * dy);
        const float power = __builtin_fmaf(-0.5f, q, -((cy * dx) * dy));
        float G = __builtin_amdgcn_exp2f(power * 1.44269504088896341f);
        float alpha = fminf(0.99f, op * G);
        const bool c = (power >= pthr) && (power <= 0.0f) && (alpha >= 1.0f / 255.0f);
        const float Ge = c ? G : 0.f, ae = c ? alpha : 0.f;
        const float oma = 1.f - ae;
        const float inv = __builtin_amdgcn_rcpf(oma);
        const float T = T_i * inv;
        float dLa = (cr - A0) * p1.x;
        dLa = __builtin_fmaf(cg - A1, p1.y, dLa);
        dLa = __builtin_fmaf(cb - A2, p1.z, dLa);
        dLa *= T;
        dLa = __builtin_fmaf(-p1.w, inv, dLa);
        const float w = Ge * dLa;
        const float aT = ae * T;
        const float wdx = w * dx, wdy = w * dy;
        acc[0] += w; acc[1] += wdx; acc[2] += wdy;
        acc[3] = __builtin_fmaf(wdx, dx, acc[3]); acc[4] = __builtin_fmaf(wdx, dy, acc[4]); acc[5] = __builtin_fmaf(wdy, dy, acc[5]);
        acc[6] = __builtin_fmaf(aT, p1.x, acc[6]); acc[7] = __builtin_fmaf(aT, p1.y, acc[7]); acc[8] = __builtin_fmaf(aT, p1.z, acc[8]);
        T_o = T;
        A0_o = __builtin_fmaf(ae, cr, oma * A0);
        A1_o = __builtin_fmaf(ae, cg, oma * A1);
        A2_o = __builtin_fmaf(ae, cb, oma * A2);
    }
    if (check) {
        float s = T_o + A0_o + A1_o + A2_o;
        for (int i = 0; i < 9; i++) s += acc[i];
        if (s == 12345.678f) check[0] = 1;
    }
}

int main(int argc, char** argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 128;          // entries per block (multiple of 64)
    const int bpw = argc > 2 ? atoi(argv[2]) : 8;          // blocks per wave
    const int waves_per_simd = argc > 3 ? atoi(argv[3]) : 4;
    const int nwaves = 256 * 4 * waves_per_simd;
    const size_t n_items = (size_t)nwaves * bpw * L;
    std::vector<Item> h(n_items);
    std::vector<float> pix((size_t)nwaves * bpw * 64 * 8);
    srand(1);
    auto rnd = [] { return (float)(rand() & 0xFFFFFF) / 16777216.0f; };
    for (size_t w = 0; w < (size_t)nwaves; w++)
        for (int b = 0; b < bpw; b++) {
            for (int p = 0; p < 64; p++) {
                float* d = &pix[((w * bpw + b) * 64 + p) * 8];
                d[0] = (float)(p & 7); d[1] = (float)(p >> 3);
                uint32_t lc = (uint32_t)(L * (0.5f + 0.5f * rnd()));
                memcpy(&d[2], &lc, 4);
                d[3] = 0.05f * rnd();                      // T_final
                d[4] = rnd() - 0.5f; d[5] = rnd() - 0.5f; d[6] = rnd() - 0.5f; d[7] = 0.f;
            }
            for (int i = 0; i < L; i++) {
                Item& it = h[(w * bpw + b) * L + i];
                it.mx = 12.f * rnd() - 2.f; it.my = 12.f * rnd() - 2.f;
                const float s = 1.5f + 3.f * rnd();
                it.cx = 1.f / (s * s); it.cz = 1.f / (s * s); it.cy = 0.2f * it.cx * (rnd() - 0.5f);
                it.op = 0.05f + 0.9f * rnd();
                it.pthr = fmaxf(logf(1.f / (255.f * it.op)) - 1e-3f, -80.f);
                it.kidx = (uint32_t)(L - 1 - i);            // back to front
                it.r = rnd(); it.g = rnd(); it.b = rnd();
                it.row = (uint32_t)((w * bpw + b) * L + i);
                it.pixb = (uint32_t)((b % 3) * 64 * 32);
                it.first = i == 0;
            }
        }
    Item* d_items; float* d_pix; float* d_rows; int* d_check;
    CHECK(hipMalloc(&d_items, n_items * sizeof(Item)));
    CHECK(hipMalloc(&d_pix, pix.size() * 4));
    CHECK(hipMalloc(&d_rows, n_items * 48));
    CHECK(hipMalloc(&d_check, 4));
    CHECK(hipMemcpy(d_items, h.data(), n_items * sizeof(Item), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_pix, pix.data(), pix.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_rows, 0, n_items * 48));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int variant = 0; variant < 8; variant++) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(a);
#define LAUNCH(S, V) hipLaunchKernelGGL((skew_kernel<S, V>), dim3(nwaves), dim3(64), 0, 0, d_items, d_pix, bpw, L, d_rows, d_check)
            switch (variant) {
                case 0: LAUNCH(true, 0); break;
                case 1: LAUNCH(false, 0); break;
                case 2: LAUNCH(true, 1); break;
                case 3: LAUNCH(true, 4); break;
                case 4: LAUNCH(true, 6); break;
                case 5: LAUNCH(true, 14); break;
                case 6: LAUNCH(true, 7); break;
                case 7: LAUNCH(false, 15); break;
            }
            hipEventRecord(b);
            CHECK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        const char* names[8] = {"baseline (store rows)", "no stores", "row_ror instead of wave_ror", "scalar first-mask", "scalar first-mask + pixel prefetch",
                                "... + no acc zeroing", "scalar mask + prefetch + row_ror", "everything off (no store, no zero, row_ror, prefetch)"};
        const double steps_per_simd = (double)waves_per_simd * (bpw * L + 64);
        printf("%-55s L=%d blocks/wave=%d waves/SIMD=%d: %.3f ms, %.1f cycles per step per SIMD at 2.4 GHz\n", names[variant], L, bpw, waves_per_simd,
               best, best * 1e-3 * 2.4e9 / steps_per_simd);
    }
    // sanity: a few rows must be non-zero and finite
    std::vector<float> r(12 * 1024);
    CHECK(hipMemcpy(r.data(), d_rows, r.size() * 4, hipMemcpyDeviceToHost));
    int nz = 0, bad = 0;
    for (float v : r) { nz += v != 0.f; bad += !std::isfinite(v); }
    printf("rows sample: %d non-zero of %zu, %d non-finite\n", nz, r.size(), bad);
    return 0;
}
