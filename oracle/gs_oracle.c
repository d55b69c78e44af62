/*
 * gs_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle), never the product path.
 *
 * Plain-C restatement of the reference's differentiable Gaussian rasterizer
 * (submodules/diff-gaussian-rasterization, "DGR") for use as the parity checker
 * of the MI355X HIP path and as bench.py's `cpu_baseline` ("port").  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY STATUS: the reference ships no golden vectors and its CUDA sources
 * cannot be compiled in this image (no nvcc / CUDA / CUB headers), so this
 * oracle is "parity unpinned" against the CUDA binary.  It is pinned instead by
 * (i) the reference's importable Python helpers (utils/sh_utils.eval_sh,
 * utils/graphics_utils) through tests/golden fixtures, (ii) an independent
 * float64 autograd re-derivation in tests/, (iii) analytic micro-cases.
 *
 * Arithmetic contract (shared with the HIP kernels, which are written
 * independently): everything is IEEE fp32, evaluated left to right exactly as
 * typed in the reference, NO fused multiply-add contraction (-ffp-contract=off)
 * except the explicit fmaf() calls in the pixel/Gaussian blend pair (marked
 * "CONTRACT-FMA"), and exp() in the blend is the fully specified gs_expf_blend()
 * below, so the forward pass is reproducible bit for bit on CPU and GPU.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/submodules/diff-gaussian-rasterization unless noted).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16 /* cuda_rasterizer/config.h:15 */
#define BLOCK_Y 16 /* cuda_rasterizer/config.h:16 */

/* cuda_rasterizer/auxiliary.h:22-39 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct { float x, y, z; } v3;

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }

/* float -> int conversion as the GPU does it (cvt.rzi saturates, NaN -> 0);
 * used where the reference writes (int)(float) in device code. */
static inline int f2i_sat(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT_MAX;
    if (f <= -2147483648.0f) return INT_MIN;
    return (int)f;
}

/* ------------------------------------------------------------------------
 * gs_expf: exp() of the parameter activations (orc_activate), fully specified so CPU and GPU agree bit for bit.
 * Cephes-style: n = rint(x*log2e); r = x - n*ln2 (two fmaf); degree-5 Horner
 * in fmaf; scale by 2^n through the exponent bits.  x < -80 -> 0, x > 80 -> 80.
 * ------------------------------------------------------------------------ */
float gs_expf(float x)
{
    if (x < -80.0f) return 0.0f;
    if (x > 80.0f) x = 80.0f;
    float t = x * 1.44269504088896341f;
    float n = rintf(t);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float z = r * r;
    float y = fmaf(p, z, r);
    y = y + 1.0f;
    union { float f; int32_t i; } u;
    u.f = y;
    u.i += ((int32_t)n) * (1 << 23);
    return u.f;
}

/* ------------------------------------------------------------------------
 * gs_expf_blend: the blend's exp() (the reference calls CUDA's exp(float), forward.cu:343 / backward.cu:497, whose
 * last-ulp behaviour is not reproducible anywhere else), fully specified and built for the MI355X's vector pipe: nine
 * instructions, seven of them full-rate fmaf (round 3; gs_expf above costs 16, and the exp was a third of the forward
 * blend kernel).  The blend only asks for x in [ln(1/255) - eps, 0] (alpha >= 1/255 needs G >= 1/255):
 *   m = fmaf(x, log2e, 1.5*2^23)   -- adding the magic number rounds x*log2e to the nearest integer n in m's low bits
 *   f = fmaf(x, log2e, -(m - 1.5*2^23))                       -- the exact product minus n: f in [-0.5, 0.5]
 *   p = 1 + f*(c1 + f*(c2 + f*(c3 + f*(c4 + f*c5))))          -- 2^f, Horner in fmaf (minimax, relative error 1.0e-7)
 *   result = bits(p) + (bits(m) << 23)                         -- * 2^n through the exponent bits
 * Max relative error 2.5e-7 on [-6, 0] (the same as gs_expf), 1.3e-6 at -80; monotone; exp(0) = 1.  x < -80 -> 0.
 * ------------------------------------------------------------------------ */
float gs_expf_blend(float x)
{
    if (x < -80.0f) return 0.0f;
    const float L = 1.44269504088896341f, MAGIC = 12582912.0f;
    float m = fmaf(x, L, MAGIC);
    float nf = m - MAGIC;
    float f = fmaf(x, L, -nf);
    float p = 0x1.5cbc2cp-10f;
    p = fmaf(p, f, 0x1.3cea8ap-7f);
    p = fmaf(p, f, 0x1.c6b48ep-5f);
    p = fmaf(p, f, 0x1.ebf9bcp-3f);
    p = fmaf(p, f, 0x1.62e42cp-1f);
    p = fmaf(p, f, 1.0f);
    union { float f; uint32_t u; } a, b;
    a.f = p;
    b.f = m;
    a.u += b.u << 23;
    return a.f;
}

/* auxiliary.h:58-77 */
static inline v3 transformPoint4x3(v3 p, const float* m)
{
    v3 o;
    o.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    o.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    o.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    return o;
}
static inline void transformPoint4x4(v3 p, const float* m, float o[4])
{
    o[0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    o[1] = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    o[2] = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    o[3] = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
}

/* auxiliary.h:41-44: evaluated in double, rounded to float on return */
static inline float ndc2Pix(float v, int S)
{
    return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

/* auxiliary.h:46-56 */
static inline void getRect(float px, float py, int max_radius, int gx, int gy,
                           uint32_t* minx, uint32_t* miny, uint32_t* maxx, uint32_t* maxy)
{
    float r = (float)max_radius;
    int a;
    a = f2i_sat((px - r) / (float)BLOCK_X); if (a < 0) a = 0; if (a > gx) a = gx; *minx = (uint32_t)a;
    a = f2i_sat((py - r) / (float)BLOCK_Y); if (a < 0) a = 0; if (a > gy) a = gy; *miny = (uint32_t)a;
    a = f2i_sat((px + r + (float)(BLOCK_X - 1)) / (float)BLOCK_X); if (a < 0) a = 0; if (a > gx) a = gx; *maxx = (uint32_t)a;
    a = f2i_sat((py + r + (float)(BLOCK_Y - 1)) / (float)BLOCK_Y); if (a < 0) a = 0; if (a > gy) a = gy; *maxy = (uint32_t)a;
}

/* forward.cu:118-152 (GLM column-major algebra expanded):
 * M[c][k] = s_k*Rq[c][k];  Sigma[c][r] = sum_k M[r][k]*M[c][k], k = 0,1,2 left to right. */
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D)
{
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    float Rq[3][3] = {
        {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
        {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
        {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    float M[3][3];
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 3; k++)
            M[c][k] = s[k] * Rq[c][k];
#define SIG(c, r) (M[r][0] * M[c][0] + M[r][1] * M[c][1] + M[r][2] * M[c][2])
    cov3D[0] = SIG(0, 0);
    cov3D[1] = SIG(0, 1);
    cov3D[2] = SIG(0, 2);
    cov3D[3] = SIG(1, 1);
    cov3D[4] = SIG(1, 2);
    cov3D[5] = SIG(2, 2);
#undef SIG
}

/* Shared by forward.cu:74-113 and backward.cu:160-195: builds the clamped t,
 * T = W*J (GLM [col][row]) and the 2D covariance before the +0.3 filter. */
typedef struct {
    float T[3][3]; /* GLM indexing T[c][r] */
    float cov00, cov01, cov11;
    float tx, ty, tz; /* clamped t */
    float txtz, tytz, limx, limy;
} Cov2DInter;

static void cov2D_core(v3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                       const float* c3, const float* v, Cov2DInter* o)
{
    v3 t = transformPoint4x3(mean, v);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z;
    const float tytz = t.y / t.z;
    t.x = fminf_(limx, fmaxf_(-limx, txtz)) * t.z;
    t.y = fminf_(limy, fmaxf_(-limy, tytz)) * t.z;
    const float J00 = fx / t.z, J02 = -(fx * t.x) / (t.z * t.z);
    const float J11 = fy / t.z, J12 = -(fy * t.y) / (t.z * t.z);
    float (*T)[3] = o->T;
    for (int r = 0; r < 3; r++) {
        /* T[c][r] = W[0][r]*J[c][0] + W[1][r]*J[c][1] + W[2][r]*J[c][2], W[k][r] = v[k+4r] */
        T[0][r] = v[4 * r] * J00 + v[4 * r + 1] * 0.0f + v[4 * r + 2] * J02;
        T[1][r] = v[4 * r] * 0.0f + v[4 * r + 1] * J11 + v[4 * r + 2] * J12;
        T[2][r] = v[4 * r] * 0.0f + v[4 * r + 1] * 0.0f + v[4 * r + 2] * 0.0f;
    }
    const float Vrk[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
    float A[3][3]; /* A = transpose(T)*transpose(Vrk): A[c][r] = sum_k T[r][k]*Vrk[k][c] */
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            A[c][r] = T[r][0] * Vrk[0][c] + T[r][1] * Vrk[1][c] + T[r][2] * Vrk[2][c];
    /* cov = A*T: cov[c][r] = A[0][r]*T[c][0] + A[1][r]*T[c][1] + A[2][r]*T[c][2] */
    o->cov00 = A[0][0] * T[0][0] + A[1][0] * T[0][1] + A[2][0] * T[0][2];
    o->cov01 = A[0][1] * T[0][0] + A[1][1] * T[0][1] + A[2][1] * T[0][2];
    o->cov11 = A[0][1] * T[1][0] + A[1][1] * T[1][1] + A[2][1] * T[1][2];
    o->tx = t.x; o->ty = t.y; o->tz = t.z;
    o->txtz = txtz; o->tytz = tytz; o->limx = limx; o->limy = limy;
}

/* forward.cu:20-71 */
static void computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                               const float* shs, uint8_t* clamped, float* out)
{
    float dx = means[3 * idx] - campos[0], dy = means[3 * idx + 1] - campos[1], dz = means[3 * idx + 2] - campos[2];
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    float x = dx / len, y = dy / len, z = dz / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    for (int ch = 0; ch < 3; ch++) {
#define S(i) sh[3 * (i) + ch]
        float result = SH_C0 * S(0);
        if (deg > 0) {
            result = result - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z;
                float xy = x * y, yz = y * z, xz = x * z;
                result = result + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) +
                         SH_C2[2] * (2.0f * zz - xx - yy) * S(6) + SH_C2[3] * xz * S(7) +
                         SH_C2[4] * (xx - yy) * S(8);
                if (deg > 2) {
                    result = result + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
                             SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) +
                             SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) +
                             SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14) +
                             SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
                }
            }
        }
#undef S
        result += 0.5f;
        clamped[3 * idx + ch] = (result < 0);
        out[ch] = fmaxf_(result, 0.0f);
    }
}

/* ------------------------------------------------------------------------
 * K1 preprocess + K2 scan.  forward.cu:156-256, rasterizer_impl.cu:222-281.
 * All per-Gaussian outputs are zero-filled first (the reference leaves culled
 * entries uninitialised; parity is only defined where radii > 0).
 * Returns num_rendered (= point_offsets[P-1]).
 * ------------------------------------------------------------------------ */
long orc_preprocess(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                    const float* rotations, const float* opacities, const float* shs, const float* cov3D_precomp,
                    const float* colors_precomp, const float* viewmatrix, const float* projmatrix,
                    const float* campos, int W, int H, float tan_fovx, float tan_fovy,
                    /* out */ int* radii, float* means2D, float* depths, float* cov3Ds, float* rgb,
                    float* conic_opacity, uint8_t* clamped, uint32_t* tiles_touched, uint32_t* point_offsets)
{
    const float focal_y = H / (2.0f * tan_fovy); /* rasterizer_impl.cu:222-223 */
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    memset(radii, 0, sizeof(int) * (size_t)P);
    memset(means2D, 0, sizeof(float) * 2 * (size_t)P);
    memset(depths, 0, sizeof(float) * (size_t)P);
    memset(cov3Ds, 0, sizeof(float) * 6 * (size_t)P);
    memset(rgb, 0, sizeof(float) * 3 * (size_t)P);
    memset(conic_opacity, 0, sizeof(float) * 4 * (size_t)P);
    memset(clamped, 0, 3 * (size_t)P);
    memset(tiles_touched, 0, sizeof(uint32_t) * (size_t)P);
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        v3 p_orig = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        /* in_frustum, auxiliary.h:139-164 */
        float p_hom[4];
        transformPoint4x4(p_orig, projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj_x = p_hom[0] * p_w, p_proj_y = p_hom[1] * p_w;
        v3 p_view = transformPoint4x3(p_orig, viewmatrix);
        if (p_view.z <= 0.2f) continue;
        const float* cov3D;
        float c3[6];
        if (cov3D_precomp) {
            cov3D = cov3D_precomp + 6 * (size_t)idx;
        } else {
            computeCov3D(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, c3);
            memcpy(cov3Ds + 6 * (size_t)idx, c3, sizeof(c3));
            cov3D = c3;
        }
        Cov2DInter ci;
        cov2D_core(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &ci);
        float covx = ci.cov00 + 0.3f, covy = ci.cov01, covz = ci.cov11 + 0.3f; /* forward.cu:110-112 */
        float det = (covx * covz - covy * covy);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {covz * det_inv, -covy * det_inv, covx * det_inv};
        float mid = 0.5f * (covx + covz);
        float lambda1 = mid + sqrtf(fmaxf_(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf_(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf_(lambda1, lambda2)));
        float pix = ndc2Pix(p_proj_x, W), piy = ndc2Pix(p_proj_y, H);
        uint32_t mnx, mny, mxx, mxy;
        getRect(pix, piy, f2i_sat(my_radius), gx, gy, &mnx, &mny, &mxx, &mxy);
        if ((mxx - mnx) * (mxy - mny) == 0) continue;
        if (!colors_precomp) {
            float col[3];
            computeColorFromSH(idx, D, M, means3D, campos, shs, clamped, col);
            rgb[3 * idx] = col[0]; rgb[3 * idx + 1] = col[1]; rgb[3 * idx + 2] = col[2];
        }
        depths[idx] = p_view.z;
        radii[idx] = f2i_sat(my_radius);
        means2D[2 * idx] = pix; means2D[2 * idx + 1] = piy;
        conic_opacity[4 * idx] = conic[0]; conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2]; conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (mxy - mny) * (mxx - mnx);
    }
    /* InclusiveSum, rasterizer_impl.cu:277 */
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += tiles_touched[i]; point_offsets[i] = acc; }
    return (long)acc;
}

/* rasterizer_impl.cu:35-50 */
uint32_t orc_getHigherMsb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* Stable LSD radix sort of (u64 key, u32 value) on bits [0, end_bit) -- the
 * contract of cub::DeviceRadixSort::SortPairs at rasterizer_impl.cu:303-308.
 * OpenMP: the input is cut into one contiguous chunk per thread; per digit pass every thread counts its chunk, the
 * (digit-major, chunk-minor) exclusive scan of the counts gives every chunk its first slot per digit, and every thread
 * scatters its chunk in order -- equal digits keep their input order, so each pass (and the whole sort) is stable. */
static void radix_sort_pairs(long n, const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout, int end_bit)
{
    uint64_t* ka = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    uint32_t* va = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    uint64_t* kb = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    uint32_t* vb = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    memcpy(ka, kin, sizeof(uint64_t) * (size_t)n);
    memcpy(va, vin, sizeof(uint32_t) * (size_t)n);
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
#endif
    if (n < 65536) nt = 1;
    size_t* hist = (size_t*)malloc(sizeof(size_t) * 256 * (size_t)nt);
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint32_t mask = (1u << bits) - 1u;
        memset(hist, 0, sizeof(size_t) * 256 * (size_t)nt);
#pragma omp parallel num_threads(nt)
        {
            int t = 0;
#ifdef _OPENMP
            t = omp_get_thread_num();
#endif
            const long lo = n * t / nt, hi = n * (t + 1) / nt;
            size_t* h = hist + 256 * (size_t)t;
            for (long i = lo; i < hi; i++) h[(ka[i] >> shift) & mask]++;
#pragma omp barrier
#pragma omp single
            {
                size_t run = 0;
                for (int d = 0; d < 256; d++)
                    for (int c = 0; c < nt; c++) { size_t v = hist[256 * (size_t)c + d]; hist[256 * (size_t)c + d] = run; run += v; }
            }
            for (long i = lo; i < hi; i++) {
                size_t pos = h[(ka[i] >> shift) & mask]++;
                kb[pos] = ka[i];
                vb[pos] = va[i];
            }
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    memcpy(kout, ka, sizeof(uint64_t) * (size_t)n);
    memcpy(vout, va, sizeof(uint32_t) * (size_t)n);
    free(hist); free(ka); free(va); free(kb); free(vb);
}

/* ------------------------------------------------------------------------
 * K4 duplicateWithKeys (rasterizer_impl.cu:70-111), K5 sort (:300-308),
 * K6 identifyTileRanges (:116-138, :310-318).
 * ------------------------------------------------------------------------ */
void orc_binning(int P, long R, int W, int H, const int* radii, const float* means2D, const float* depths,
                 const uint32_t* point_offsets,
                 /* out */ uint64_t* keys_unsorted, uint32_t* values_unsorted, uint64_t* keys_sorted,
                 uint32_t* point_list, uint32_t* ranges /* [T][2] */)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
            uint32_t mnx, mny, mxx, mxy;
            getRect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, &mnx, &mny, &mxx, &mxy);
            uint32_t dbits;
            memcpy(&dbits, &depths[idx], 4);
            for (uint32_t y = mny; y < mxy; y++)
                for (uint32_t x = mnx; x < mxx; x++) {
                    uint64_t key = (uint64_t)(y * (uint32_t)gx + x);
                    key <<= 32;
                    key |= dbits;
                    keys_unsorted[off] = key;
                    values_unsorted[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
    int bit = (int)orc_getHigherMsb((uint32_t)(gx * gy));
    radix_sort_pairs(R, keys_unsorted, values_unsorted, keys_sorted, point_list, 32 + bit);
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    for (long i = 0; i < R; i++) {
        uint32_t cur = (uint32_t)(keys_sorted[i] >> 32);
        if (i == 0) ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(keys_sorted[i - 1] >> 32);
            if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
    }
}

/* Blend pair evaluation shared by forward and backward (forward.cu:332-343,
 * backward.cu:484-500).  CONTRACT-FMA: the two mul-add pairs are fused.
 * Returns 0 if the pair is skipped. */
static inline int blend_pair(const float* xy, const float* con_o, float pixx, float pixy,
                             float* dxo, float* dyo, float* Go, float* alphao)
{
    float dx = xy[0] - pixx, dy = xy[1] - pixy;
    float q = fmaf(con_o[0] * dx, dx, (con_o[2] * dy) * dy);
    float power = fmaf(-0.5f, q, -((con_o[1] * dx) * dy));
    if (power > 0.0f) return 0;
    float G = gs_expf_blend(power);
    float alpha = fminf(0.99f, con_o[3] * G);
    if (alpha < 1.0f / 255.0f) return 0;
    *dxo = dx; *dyo = dy; *Go = G; *alphao = alpha;
    return 1;
}

/* K7 renderCUDA forward, forward.cu:261-374.  Per-pixel loops (the block-wide
 * early exit of :301-304 does not change any per-pixel result).
 * stats (optional, 4 x int64; [3] = list entries with at least one blended pixel in their tile): [0] visited pairs, [1] blended pairs, [2] (list entry, 16x4 pixel strip) instances with
 * at least one blended pixel -- the unit of work of the GPU blend kernels' slow path. */
void orc_render_forward(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D,
                        const float* features, const float* conic_opacity, const float* bg,
                        /* out */ float* final_T, uint32_t* n_contrib, float* out_color, long long* stats)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    long long visited = 0, blended = 0, strips = 0, touched = 0, heavy8 = 0, qmax = 0, qsum = 0, hmax = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : visited, blended, strips, touched, heavy8, qmax, qsum, hmax)
    for (int tile = 0; tile < gx * gy; tile++) {
        int ty = tile / gx, tx = tile % gx;
        uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        uint8_t* sflag = stats ? (uint8_t*)calloc((size_t)(r1 - r0) + 1, 1) : NULL;
        uint16_t* qflag = stats ? (uint16_t*)calloc((size_t)(r1 - r0) + 1, 2) : NULL; /* 4x4-pixel quarters that blend the entry */
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (px >= W || py >= H) continue;
                float pixx = (float)px, pixy = (float)py;
                float T = 1.0f, C[3] = {0, 0, 0};
                uint32_t contributor = 0, last_contributor = 0;
                for (uint32_t k = r0; k < r1; k++) {
                    contributor++;
                    visited++;
                    uint32_t id = point_list[k];
                    float dx, dy, G, alpha;
                    if (!blend_pair(means2D + 2 * (size_t)id, conic_opacity + 4 * (size_t)id, pixx, pixy, &dx, &dy, &G, &alpha))
                        continue;
                    float test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) break; /* done = true */
                    for (int ch = 0; ch < 3; ch++) /* CONTRACT-FMA: C += (f*alpha)*T */
                        C[ch] = fmaf(features[3 * (size_t)id + ch] * alpha, T, C[ch]);
                    T = test_T;
                    last_contributor = contributor;
                    blended++;
                    if (sflag) { sflag[k - r0] |= (uint8_t)(1u << (ly / 4)); qflag[k - r0] |= (uint16_t)(1u << ((ly / 4) * 4 + lx / 4)); }
                }
                size_t pix_id = (size_t)W * py + px;
                final_T[pix_id] = T;
                n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < 3; ch++)
                    out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * bg[ch];
            }
        if (sflag) {
            for (uint32_t k = 0; k < r1 - r0; k++) { strips += __builtin_popcount(sflag[k]); if (sflag[k]) touched++; }
            free(sflag);
            /* what-if statistics for sub-wave work lists: a wave owns an 8x8 block = 4 quarters of 4x4 pixels */
            for (int wv = 0; wv < 4; wv++) {
                const int wy = wv / 2, wx = wv % 2;
                int qb[4];
                for (int j = 0; j < 4; j++) qb[j] = (2 * wy + j / 2) * 4 + 2 * wx + j % 2;
                long long len[4] = {0, 0, 0, 0}, lenh[2] = {0, 0}, hv = 0;
                for (uint32_t k = 0; k < r1 - r0; k++) {
                    int any = 0, h0 = 0, h1 = 0;
                    for (int j = 0; j < 4; j++)
                        if (qflag[k] >> qb[j] & 1) { len[j]++; any = 1; if (j < 2) h0 = 1; else h1 = 1; }
                    hv += any; lenh[0] += h0; lenh[1] += h1;
                }
                heavy8 += hv;
                long long m = len[0];
                for (int j = 1; j < 4; j++) if (len[j] > m) m = len[j];
                qmax += m;
                qsum += len[0] + len[1] + len[2] + len[3];
                hmax += lenh[0] > lenh[1] ? lenh[0] : lenh[1];
            }
            free(qflag);
        }
    }
    if (stats) {
        stats[0] = visited; stats[1] = blended; stats[2] = strips; stats[3] = touched;
        stats[4] = heavy8; stats[5] = qmax; stats[6] = qsum; stats[7] = hmax;
    }
}

/* Per-8x8-block work lists of the backward (no reference counterpart: the reference's backward re-walks whole tile lists,
 * backward.cu:470-482; this restates which (list entry, 8x8 block) pairs it finds work for).  flags[k - ranges[0]] gets bit
 * q = 2*(ly/8) + lx/8 set iff list entry k is BLENDED into at least one pixel of quadrant q of its tile, with the forward's
 * own predicates (forward.cu:332-357: power <= 0, alpha >= 1/255, pixel not yet done and not saturating at this entry). */
void orc_block_flags(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D,
                     const float* conic_opacity, /* out, one byte per list entry */ uint8_t* flags)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; tile++) {
        int ty = tile / gx, tx = tile % gx;
        uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (uint32_t k = r0; k < r1; k++) flags[k] = 0;
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (px >= W || py >= H) continue;
                float T = 1.0f;
                for (uint32_t k = r0; k < r1; k++) {
                    uint32_t id = point_list[k];
                    float dx, dy, G, alpha;
                    if (!blend_pair(means2D + 2 * (size_t)id, conic_opacity + 4 * (size_t)id, (float)px, (float)py, &dx, &dy, &G, &alpha))
                        continue;
                    float test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) break;
                    T = test_T;
                    flags[k] |= (uint8_t)(1u << (2 * (ly / 8) + lx / 8));
                }
            }
    }
}

/* K8 renderCUDA backward, backward.cu:399-557.  The reference accumulates the 9
 * per-Gaussian sums with float atomicAdd in nondeterministic order; the oracle
 * computes every per-pair term in fp32 exactly as typed and accumulates the sums
 * in double (the order-free ideal of those atomics), rounding once at the end.
 * dL_dmean2D is [P][3] (z untouched = 0), dL_dconic is [P][4] (x,y,w used). */
static void render_backward_impl(int abs_mode, int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* bg,
                         const float* means2D, const float* conic_opacity, const float* colors, const float* final_Ts,
                         const uint32_t* n_contrib, const float* dL_dpixels,
                         /* out */ float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    double* acc = (double*)calloc((size_t)(P > 0 ? P : 1) * 9, sizeof(double));
    const float ddelx_dx = 0.5 * W; /* backward.cu:460-461 */
    const float ddely_dy = 0.5 * H;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; tile++) {
        int ty = tile / gx, tx = tile % gx;
        uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        uint32_t len = r1 - r0;
        if (len == 0) continue;
        double* loc = (double*)calloc((size_t)len * 9, sizeof(double));
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (px >= W || py >= H) continue;
                size_t pix_id = (size_t)W * py + px;
                float pixx = (float)px, pixy = (float)py;
                const float T_final = final_Ts[pix_id];
                float T = T_final;
                const uint32_t last_contributor = n_contrib[pix_id];
                float accum_rec[3] = {0, 0, 0}, dL_dpixel[3], last_color[3] = {0, 0, 0};
                for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[(size_t)i * H * W + pix_id];
                float last_alpha = 0;
                /* entries with index >= last_contributor are skipped (:480-482) */
                for (long kk = (long)last_contributor - 1; kk >= 0; kk--) {
                    uint32_t id = point_list[r0 + kk];
                    const float* con_o = conic_opacity + 4 * (size_t)id;
                    float dx, dy, G, alpha;
                    if (!blend_pair(means2D + 2 * (size_t)id, con_o, pixx, pixy, &dx, &dy, &G, &alpha)) continue;
                    double* a = loc + (size_t)kk * 9;
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    float dL_dalpha = 0.0f;
                    for (int ch = 0; ch < 3; ch++) {
                        const float c = colors[3 * (size_t)id + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        const float dL_dchannel = dL_dpixel[ch];
                        dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                        a[6 + ch] += (double)(abs_mode ? fabsf(dchannel_dcolor * dL_dchannel) : dchannel_dcolor * dL_dchannel);
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    float bg_dot_dpixel = 0;
                    for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                    if (abs_mode) {
                        /* Condition-number pass (orc_render_backward_abs): every product keeps its factors' magnitudes and
                         * every sum adds magnitudes, so a[j] bounds the sum of |terms| that ANY summation of this gradient --
                         * per-pixel gradients as here, or moments of G*dL_dalpha as the HIP kernel forms -- adds up. */
                        float mag = 0.0f;
                        for (int ch = 0; ch < 3; ch++) {
                            /* accum_rec[ch] is already the value this pair sees; last_color[ch] = this pair's colour */
                            mag += fabsf(last_color[ch] - accum_rec[ch]) * fabsf(dL_dpixel[ch]);
                        }
                        mag *= T;
                        float bgm = 0;
                        for (int i = 0; i < 3; i++) bgm += fabsf(bg[i] * dL_dpixel[i]);
                        mag += fabsf(T_final / (1.f - alpha)) * bgm;
                        const float dLdG = fabsf(con_o[3]) * mag;
                        const float agdx = fabsf(G * dx), agdy = fabsf(G * dy);
                        a[0] += (double)(dLdG * (agdx * fabsf(con_o[0]) + agdy * fabsf(con_o[1])) * ddelx_dx);
                        a[1] += (double)(dLdG * (agdy * fabsf(con_o[2]) + agdx * fabsf(con_o[1])) * ddely_dy);
                        a[2] += (double)(0.5f * agdx * fabsf(dx) * dLdG);
                        a[3] += (double)(0.5f * agdx * fabsf(dy) * dLdG);
                        a[4] += (double)(0.5f * agdy * fabsf(dy) * dLdG);
                        a[5] += (double)(G * mag);
                        continue;
                    }
                    const float dL_dG = con_o[3] * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
                    const float dG_ddely = -gdy * con_o[2] - gdx * con_o[1];
                    a[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                    a[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                    a[2] += (double)(-0.5f * gdx * dx * dL_dG);
                    a[3] += (double)(-0.5f * gdx * dy * dL_dG);
                    a[4] += (double)(-0.5f * gdy * dy * dL_dG);
                    a[5] += (double)(G * dL_dalpha);
                }
            }
        for (uint32_t k = 0; k < len; k++) {
            uint32_t id = point_list[r0 + k];
            for (int j = 0; j < 9; j++) {
                double v = loc[(size_t)k * 9 + j];
                if (v != 0.0) {
#pragma omp atomic
                    acc[(size_t)id * 9 + j] += v;
                }
            }
        }
        free(loc);
    }
    for (int i = 0; i < P; i++) {
        const double* a = acc + (size_t)i * 9;
        dL_dmean2D[3 * i] = (float)a[0]; dL_dmean2D[3 * i + 1] = (float)a[1]; dL_dmean2D[3 * i + 2] = 0.0f;
        dL_dconic[4 * i] = (float)a[2]; dL_dconic[4 * i + 1] = (float)a[3]; dL_dconic[4 * i + 2] = 0.0f;
        dL_dconic[4 * i + 3] = (float)a[4];
        dL_dopacity[i] = (float)a[5];
        dL_dcolors[3 * i] = (float)a[6]; dL_dcolors[3 * i + 1] = (float)a[7]; dL_dcolors[3 * i + 2] = (float)a[8];
    }
    free(acc);
}

void orc_render_backward(int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* bg,
                         const float* means2D, const float* conic_opacity, const float* colors, const float* final_Ts,
                         const uint32_t* n_contrib, const float* dL_dpixels,
                         /* out */ float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors)
{
    render_backward_impl(0, P, W, H, ranges, point_list, bg, means2D, conic_opacity, colors, final_Ts, n_contrib, dL_dpixels,
                         dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors);
}

/* The same traversal with every term replaced by its magnitude: out[j] = sum over the blended (pixel, Gaussian) pairs of
 * |term_j|.  Not a reference function: the tests use it as the computed condition number of each gradient sum
 * (an fp32 summation of n terms in any order is off by at most ~n_levels * 2^-24 * sum |terms|). */
void orc_render_backward_abs(int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* bg,
                             const float* means2D, const float* conic_opacity, const float* colors, const float* final_Ts,
                             const uint32_t* n_contrib, const float* dL_dpixels,
                             /* out */ float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors)
{
    render_backward_impl(1, P, W, H, ranges, point_list, bg, means2D, conic_opacity, colors, final_Ts, n_contrib, dL_dpixels,
                         dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors);
}

/* auxiliary.h:107-117 */
static inline v3 dnormvdv3(v3 v, v3 dv)
{
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    v3 o;
    o.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    o.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    o.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return o;
}

/* backward.cu:20-139.  dL_dmeans[idx] += ... */
static void computeColorFromSH_bw(int idx, int deg, int max_coeffs, const float* means, const float* campos,
                                  const float* shs, const uint8_t* clamped, const float* dL_dcolor,
                                  float* dL_dmeans, float* dL_dshs)
{
    v3 dir_orig = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    const float* sh = shs + (size_t)idx * max_coeffs * 3;
    float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
    float dL_dRGB[3];
    for (int ch = 0; ch < 3; ch++) dL_dRGB[ch] = dL_dcolor[3 * idx + ch] * (clamped[3 * idx + ch] ? 0.f : 1.f);
    float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
#define SH(i) sh[3 * (i) + ch]
#define OUT(i, w) for (int ch = 0; ch < 3; ch++) dL_dsh[3 * (i) + ch] = (w) * dL_dRGB[ch]
    float dRGBdsh0 = SH_C0;
    OUT(0, dRGBdsh0);
    if (deg > 0) {
        float dRGBdsh1 = -SH_C1 * y, dRGBdsh2 = SH_C1 * z, dRGBdsh3 = -SH_C1 * x;
        OUT(1, dRGBdsh1); OUT(2, dRGBdsh2); OUT(3, dRGBdsh3);
        for (int ch = 0; ch < 3; ch++) {
            dRGBdx[ch] = -SH_C1 * SH(3);
            dRGBdy[ch] = -SH_C1 * SH(1);
            dRGBdz[ch] = SH_C1 * SH(2);
        }
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            float dRGBdsh4 = SH_C2[0] * xy, dRGBdsh5 = SH_C2[1] * yz, dRGBdsh6 = SH_C2[2] * (2.f * zz - xx - yy);
            float dRGBdsh7 = SH_C2[3] * xz, dRGBdsh8 = SH_C2[4] * (xx - yy);
            OUT(4, dRGBdsh4); OUT(5, dRGBdsh5); OUT(6, dRGBdsh6); OUT(7, dRGBdsh7); OUT(8, dRGBdsh8);
            for (int ch = 0; ch < 3; ch++) {
                dRGBdx[ch] += SH_C2[0] * y * SH(4) + SH_C2[2] * 2.f * -x * SH(6) + SH_C2[3] * z * SH(7) + SH_C2[4] * 2.f * x * SH(8);
                dRGBdy[ch] += SH_C2[0] * x * SH(4) + SH_C2[1] * z * SH(5) + SH_C2[2] * 2.f * -y * SH(6) + SH_C2[4] * 2.f * -y * SH(8);
                dRGBdz[ch] += SH_C2[1] * y * SH(5) + SH_C2[2] * 2.f * 2.f * z * SH(6) + SH_C2[3] * x * SH(7);
            }
            if (deg > 2) {
                float dRGBdsh9 = SH_C3[0] * y * (3.f * xx - yy);
                float dRGBdsh10 = SH_C3[1] * xy * z;
                float dRGBdsh11 = SH_C3[2] * y * (4.f * zz - xx - yy);
                float dRGBdsh12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                float dRGBdsh13 = SH_C3[4] * x * (4.f * zz - xx - yy);
                float dRGBdsh14 = SH_C3[5] * z * (xx - yy);
                float dRGBdsh15 = SH_C3[6] * x * (xx - 3.f * yy);
                OUT(9, dRGBdsh9); OUT(10, dRGBdsh10); OUT(11, dRGBdsh11); OUT(12, dRGBdsh12);
                OUT(13, dRGBdsh13); OUT(14, dRGBdsh14); OUT(15, dRGBdsh15);
                for (int ch = 0; ch < 3; ch++) {
                    dRGBdx[ch] += (SH_C3[0] * SH(9) * 3.f * 2.f * xy + SH_C3[1] * SH(10) * yz + SH_C3[2] * SH(11) * -2.f * xy +
                                   SH_C3[3] * SH(12) * -3.f * 2.f * xz + SH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                                   SH_C3[5] * SH(14) * 2.f * xz + SH_C3[6] * SH(15) * 3.f * (xx - yy));
                    dRGBdy[ch] += (SH_C3[0] * SH(9) * 3.f * (xx - yy) + SH_C3[1] * SH(10) * xz +
                                   SH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12) * -3.f * 2.f * yz +
                                   SH_C3[4] * SH(13) * -2.f * xy + SH_C3[5] * SH(14) * -2.f * yz +
                                   SH_C3[6] * SH(15) * -3.f * 2.f * xy);
                    dRGBdz[ch] += (SH_C3[1] * SH(10) * xy + SH_C3[2] * SH(11) * 4.f * 2.f * yz +
                                   SH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13) * 4.f * 2.f * xz +
                                   SH_C3[5] * SH(14) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef OUT
    v3 dL_ddir;
    dL_ddir.x = dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2];
    dL_ddir.y = dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2];
    dL_ddir.z = dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2];
    v3 d = dnormvdv3(dir_orig, dL_ddir);
    dL_dmeans[3 * idx] += d.x; dL_dmeans[3 * idx + 1] += d.y; dL_dmeans[3 * idx + 2] += d.z;
}

/* backward.cu:278-341 */
static void computeCov3D_bw(int idx, const float* scale, float mod, const float* rot, const float* dL_dcov3Ds,
                            float* dL_dscales, float* dL_drots)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    float Rq[3][3] = {
        {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
        {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
        {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    float M[3][3]; /* GLM M[c][r] = s_r * Rq[c][r] */
    for (int c = 0; c < 3; c++)
        for (int k = 0; k < 3; k++) M[c][k] = s[k] * Rq[c][k];
    const float* g = dL_dcov3Ds + 6 * (size_t)idx;
    /* GLM dL_dSigma[c][r] (symmetric) */
    float dS[3][3] = {{g[0], 0.5f * g[1], 0.5f * g[2]}, {0.5f * g[1], g[3], 0.5f * g[4]}, {0.5f * g[2], 0.5f * g[4], g[5]}};
    /* dL_dM = (2.0f*M)*dL_dSigma : dL_dM[c][r] = sum_k (2M)[k][r]*dS[c][k] */
    float dM[3][3];
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++)
            dM[c][rr] = (2.0f * M[0][rr]) * dS[c][0] + (2.0f * M[1][rr]) * dS[c][1] + (2.0f * M[2][rr]) * dS[c][2];
    /* Rt[c][r] = R[r][c] = Rq[r][c];  dL_dMt[c][r] = dM[r][c] */
    float dMt[3][3];
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) dMt[c][rr] = dM[rr][c];
    float* dsc = dL_dscales + 3 * (size_t)idx;
    for (int c = 0; c < 3; c++)
        dsc[c] = Rq[0][c] * dMt[c][0] + Rq[1][c] * dMt[c][1] + Rq[2][c] * dMt[c][2];
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) dMt[c][rr] *= s[c];
    float* dq = dL_drots + 4 * (size_t)idx;
    dq[0] = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
    dq[1] = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
    dq[2] = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
    dq[3] = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
}

/* K9 computeCov2DCUDA (backward.cu:144-274) + K10 preprocessCUDA backward
 * (backward.cu:346-396).  Gradient outputs must be zero-initialised by the
 * caller exactly as rasterize_points.cu:151-159 does (torch::zeros). */
static void preprocess_backward_one(int idx, int D, int M, const float* means3D, const float* shs,
                             const uint8_t* clamped, const float* scales, const float* rotations, float scale_modifier,
                             const float* cov3Ds, const float* viewmatrix, const float* projmatrix, int W, int H,
                             float tan_fovx, float tan_fovy, const float* campos, const float* dL_dmean2D,
                             const float* dL_dconics, float* dL_dmeans, const float* dL_dcolor, float* dL_dcov3D,
                             float* dL_dsh, float* dL_dscale, float* dL_drot)
{
    const float h_y = H / (2.0f * tan_fovy); /* rasterizer_impl.cu:381-382 */
    const float h_x = W / (2.0f * tan_fovx);
    const float* v = viewmatrix;
    const float* proj = projmatrix;
    {
        /* ---- computeCov2DCUDA ---- */
        v3 mean = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        const float* cov3D = cov3Ds + 6 * (size_t)idx;
        float dLc[3] = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
        Cov2DInter ci;
        cov2D_core(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, v, &ci);
        const float x_grad_mul = (ci.txtz < -ci.limx || ci.txtz > ci.limx) ? 0.f : 1.f;
        const float y_grad_mul = (ci.tytz < -ci.limy || ci.tytz > ci.limy) ? 0.f : 1.f;
        float (*T)[3] = ci.T;
        const float Vrk[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
        float a = ci.cov00 + 0.3f, b = ci.cov01, c = ci.cov11 + 0.3f;
        float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* dcv = dL_dcov3D + 6 * (size_t)idx;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dLc[0] + 2 * b * c * dLc[1] + (denom - a * c) * dLc[2]);
            dL_dc = denom2inv * (-a * a * dLc[2] + 2 * a * b * dLc[1] + (denom - a * c) * dLc[0]);
            dL_db = denom2inv * 2 * (b * c * dLc[0] - (denom + 2 * b * b) * dLc[1] + a * b * dLc[2]);
            dcv[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
            dcv[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
            dcv[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
            dcv[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
            dcv[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
            dcv[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) dcv[i] = 0;
        }
        float dL_dT00 = 2 * (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_da +
                        (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_db;
        float dL_dT01 = 2 * (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_da +
                        (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_db;
        float dL_dT02 = 2 * (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_da +
                        (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_db;
        float dL_dT10 = 2 * (T[1][0] * Vrk[0][0] + T[1][1] * Vrk[0][1] + T[1][2] * Vrk[0][2]) * dL_dc +
                        (T[0][0] * Vrk[0][0] + T[0][1] * Vrk[0][1] + T[0][2] * Vrk[0][2]) * dL_db;
        float dL_dT11 = 2 * (T[1][0] * Vrk[1][0] + T[1][1] * Vrk[1][1] + T[1][2] * Vrk[1][2]) * dL_dc +
                        (T[0][0] * Vrk[1][0] + T[0][1] * Vrk[1][1] + T[0][2] * Vrk[1][2]) * dL_db;
        float dL_dT12 = 2 * (T[1][0] * Vrk[2][0] + T[1][1] * Vrk[2][1] + T[1][2] * Vrk[2][2]) * dL_dc +
                        (T[0][0] * Vrk[2][0] + T[0][1] * Vrk[2][1] + T[0][2] * Vrk[2][2]) * dL_db;
        /* W[c][r] = v[c + 4r] */
#define Wm(c, r) v[(c) + 4 * (r)]
        float dL_dJ00 = Wm(0, 0) * dL_dT00 + Wm(0, 1) * dL_dT01 + Wm(0, 2) * dL_dT02;
        float dL_dJ02 = Wm(2, 0) * dL_dT00 + Wm(2, 1) * dL_dT01 + Wm(2, 2) * dL_dT02;
        float dL_dJ11 = Wm(1, 0) * dL_dT10 + Wm(1, 1) * dL_dT11 + Wm(1, 2) * dL_dT12;
        float dL_dJ12 = Wm(2, 0) * dL_dT10 + Wm(2, 1) * dL_dT11 + Wm(2, 2) * dL_dT12;
#undef Wm
        float tz = 1.f / ci.tz;
        float tz2 = tz * tz;
        float tz3 = tz2 * tz;
        float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * ci.tx) * tz3 * dL_dJ02 + (2 * h_y * ci.ty) * tz3 * dL_dJ12;
        /* transformVec4x3Transpose, auxiliary.h:89-97 */
        float dmx = v[0] * dL_dtx + v[1] * dL_dty + v[2] * dL_dtz;
        float dmy = v[4] * dL_dtx + v[5] * dL_dty + v[6] * dL_dtz;
        float dmz = v[8] * dL_dtx + v[9] * dL_dty + v[10] * dL_dtz;
        dL_dmeans[3 * idx] = dmx; dL_dmeans[3 * idx + 1] = dmy; dL_dmeans[3 * idx + 2] = dmz;

        /* ---- preprocessCUDA backward ---- */
        float m_hom[4];
        transformPoint4x4(mean, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
        float g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
        float gx_ = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        float gy_ = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        float gz_ = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        dL_dmeans[3 * idx] += gx_; dL_dmeans[3 * idx + 1] += gy_; dL_dmeans[3 * idx + 2] += gz_;
        if (shs)
            computeColorFromSH_bw(idx, D, M, means3D, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
        if (scales)
            computeCov3D_bw(idx, scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, dL_dcov3D, dL_dscale, dL_drot);
    }
}

void orc_preprocess_backward(int P, int D, int M, const float* means3D, const int* radii, const float* shs,
                             const uint8_t* clamped, const float* scales, const float* rotations, float scale_modifier,
                             const float* cov3Ds, const float* viewmatrix, const float* projmatrix, int W, int H,
                             float tan_fovx, float tan_fovy, const float* campos, const float* dL_dmean2D,
                             const float* dL_dconics, float* dL_dmeans, float* dL_dcolor, float* dL_dcov3D,
                             float* dL_dsh, float* dL_dscale, float* dL_drot)
{
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        preprocess_backward_one(idx, D, M, means3D, shs, clamped, scales, rotations, scale_modifier, cov3Ds, viewmatrix, projmatrix,
                                W, H, tan_fovx, tan_fovy, campos, dL_dmean2D, dL_dconics, dL_dmeans, dL_dcolor, dL_dcov3D, dL_dsh,
                                dL_dscale, dL_drot);
    }
}

/* Stand-alone entry points for the pieces of K1 / K10 that the reference's own PYTHON can pin (tests/golden/ref_cov3d.npz,
 * ref_sh_grad.npz): computeCov3D (forward.cu:118-152) and its backward (backward.cu:278-341) for a batch, and the SH colour
 * backward (backward.cu:20-139) alone -- dL/dSH and the part of dL/dmean that flows through the view direction. */
void orc_cov3d(int P, const float* scales, float mod, const float* rots, float* cov3D)
{
    for (int i = 0; i < P; i++) computeCov3D(scales + 3 * (size_t)i, mod, rots + 4 * (size_t)i, cov3D + 6 * (size_t)i);
}

void orc_cov3d_backward(int P, const float* scales, float mod, const float* rots, const float* dL_dcov3D, float* dL_dscales,
                        float* dL_drots)
{
    for (int i = 0; i < P; i++)
        computeCov3D_bw(i, scales + 3 * (size_t)i, mod, rots + 4 * (size_t)i, dL_dcov3D, dL_dscales, dL_drots);
}

void orc_sh_backward(int P, int D, int M, const float* means3D, const float* campos, const float* shs, const uint8_t* clamped,
                     const float* dL_dcolor, float* dL_dmeans, float* dL_dsh)
{
    for (int i = 0; i < P; i++) computeColorFromSH_bw(i, D, M, means3D, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
}

/* ------------------------------------------------------------------------
 * Parameter activations of GaussianModel (/root/reference/scene/gaussian_model.py:33-40,102-122), the inputs
 * gaussian_renderer/__init__.py:52-93 hands the rasterizer:
 *   get_scaling = torch.exp(_scaling); get_opacity = torch.sigmoid(_opacity);
 *   get_rotation = torch.nn.functional.normalize(_rotation) = x / max(||x||_2, 1e-12);
 *   get_features = torch.cat((_features_dc, _features_rest), dim=1).
 * Arithmetic contract for the fused raw-parameter operator (gsr_forward_params / gsr_backward_params): exp is gs_expf,
 * sigmoid(x) = 1 / (1 + gs_expf(-x)), ||q|| = sqrtf(((q0^2 + q1^2) + q2^2) + q3^2).  torch's own exp / sigmoid differ
 * from these by <= 2 ulp; tests/test_params_operator.py pins this restatement against torch on CPU.
 * ---------------------------------------------------------------------- */
void orc_activate(int P, int Mrest, const float* f_dc, const float* f_rest, const float* opacity_raw, const float* scaling_raw,
                  const float* rotation_raw, float* shs, float* opacities, float* scales, float* rotations)
{
    const int M = 1 + Mrest;
    for (int i = 0; i < P; i++) {
        for (int c = 0; c < 3; c++) shs[(size_t)i * M * 3 + c] = f_dc[(size_t)i * 3 + c];
        for (int c = 0; c < 3 * Mrest; c++) shs[(size_t)i * M * 3 + 3 + c] = f_rest[(size_t)i * Mrest * 3 + c];
        opacities[i] = 1.0f / (1.0f + gs_expf(-opacity_raw[i]));
        for (int c = 0; c < 3; c++) scales[3 * i + c] = gs_expf(scaling_raw[3 * i + c]);
        const float* q = rotation_raw + 4 * (size_t)i;
        const float n = sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
        const float d = fmaxf_(n, 1e-12f);
        for (int c = 0; c < 4; c++) rotations[4 * i + c] = q[c] / d;
    }
}

/* autograd of the above (torch: exp' = y, sigmoid' = (1 - y) y, normalize' = (g - q (q.g)) / ||x||, or g / eps when the
 * norm sits below the clamp; cat' = split) */
void orc_activate_backward(int P, int Mrest, const float* opacity_raw, const float* scaling_raw, const float* rotation_raw,
                           const float* dL_dsh, const float* dL_dopacity, const float* dL_dscales, const float* dL_drots,
                           float* dL_df_dc, float* dL_df_rest, float* dL_dopacity_raw, float* dL_dscaling_raw,
                           float* dL_drotation_raw)
{
    const int M = 1 + Mrest;
    for (int i = 0; i < P; i++) {
        for (int c = 0; c < 3; c++) dL_df_dc[(size_t)i * 3 + c] = dL_dsh[(size_t)i * M * 3 + c];
        for (int c = 0; c < 3 * Mrest; c++) dL_df_rest[(size_t)i * Mrest * 3 + c] = dL_dsh[(size_t)i * M * 3 + 3 + c];
        const float y = 1.0f / (1.0f + gs_expf(-opacity_raw[i]));
        dL_dopacity_raw[i] = dL_dopacity[i] * ((1.0f - y) * y);
        for (int c = 0; c < 3; c++) dL_dscaling_raw[3 * i + c] = dL_dscales[3 * i + c] * gs_expf(scaling_raw[3 * i + c]);
        const float* v = rotation_raw + 4 * (size_t)i;
        const float* g = dL_drots + 4 * (size_t)i;
        const float n = sqrtf(((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + v[3] * v[3]);
        if (n > 1e-12f) {
            float q[4];
            for (int c = 0; c < 4; c++) q[c] = v[c] / n;
            const float dot = ((q[0] * g[0] + q[1] * g[1]) + q[2] * g[2]) + q[3] * g[3];
            for (int c = 0; c < 4; c++) dL_drotation_raw[4 * i + c] = (g[c] - q[c] * dot) / n;
        } else {
            for (int c = 0; c < 4; c++) dL_drotation_raw[4 * i + c] = g[c] / 1e-12f;
        }
    }
}

/* ------------------------------------------------------------------------
 * Optimizer step: torch.optim.Adam(l, lr=0.0, eps=1e-15).step() as the reference runs it
 * (/root/reference/scene/gaussian_model.py:164-175, train.py:155-157): torch/optim/adam.py `_single_tensor_adam` with
 * weight_decay = 0, amsgrad = False, maximize = False.  The reference's arithmetic lives in its dependency torch (installed
 * here: 2.10); restated with the fused multiply-adds torch's kernels use -- exp_avg.lerp_ = fma(w, g - m, m),
 * addcmul_ = fma((1-beta2) g, g, v beta2), addcdiv_ = p + (value m) / denom -- and pinned against torch.optim.Adam on
 * CPU by tests/test_adam.py.  Scalars follow the Python doubles of adam.py and are rounded to fp32 where torch's
 * kernels receive them.
 * ---------------------------------------------------------------------- */
void orc_adam_step(long n, float* p, const float* g, float* m, float* v, double lr, double beta1, double beta2, double eps,
                   long step)
{
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float w = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2), b2 = (float)beta2;
    const float ss = (float)(-(lr / bc1)), bs = (float)sqrt(bc2), epsf = (float)eps;
    for (long i = 0; i < n; i++) {
        const float mi = fmaf(w, g[i] - m[i], m[i]);            /* CONTRACT-FMA (torch lerp) */
        const float vi = fmaf(omb2 * g[i], g[i], v[i] * b2);    /* CONTRACT-FMA (torch addcmul) */
        const float denom = sqrtf(vi) / bs + epsf;
        m[i] = mi; v[i] = vi;
        p[i] = p[i] + (ss * mi) / denom;
    }
}

/* checkFrustum, rasterizer_impl.cu:54-66 (+ auxiliary.h:139-164) */
void orc_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present)
{
    (void)projmatrix;
    for (int idx = 0; idx < P; idx++) {
        v3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        v3 pv = transformPoint4x3(p, viewmatrix);
        present[idx] = pv.z <= 0.2f ? 0 : 1;
    }
}

/* ========================================================================
 * simple-knn (submodules/simple-knn/simple_knn.cu) -- distCUDA2
 * ======================================================================== */
#define BOX_SIZE 1024 /* simple_knn.cu:12 */

/* simple_knn.cu:45-52 */
static inline uint32_t prepMorton(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
static inline uint32_t f2u_sat(float f)
{ /* device float -> uint32 conversion: saturating, NaN -> 0 */
    if (f != f || f <= 0.0f) return 0;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
/* simple_knn.cu:54-61 */
static inline uint32_t coord2Morton(v3 c, v3 mn, v3 mx)
{
    uint32_t x = prepMorton(f2u_sat(((c.x - mn.x) / (mx.x - mn.x)) * ((1 << 10) - 1)));
    uint32_t y = prepMorton(f2u_sat(((c.y - mn.y) / (mx.y - mn.y)) * ((1 << 10) - 1)));
    uint32_t z = prepMorton(f2u_sat(((c.z - mn.z) / (mx.z - mn.z)) * ((1 << 10) - 1)));
    return x | (y << 1) | (z << 2);
}

typedef struct { v3 minn, maxx; } MinMax;

/* simple_knn.cu:119-129 */
static inline float distBoxPoint(const MinMax* box, v3 p)
{
    float dx = 0, dy = 0, dz = 0;
    if (p.x < box->minn.x || p.x > box->maxx.x) dx = fminf_(fabsf(p.x - box->minn.x), fabsf(p.x - box->maxx.x));
    if (p.y < box->minn.y || p.y > box->maxx.y) dy = fminf_(fabsf(p.y - box->minn.y), fabsf(p.y - box->maxx.y));
    if (p.z < box->minn.z || p.z > box->maxx.z) dz = fminf_(fabsf(p.z - box->minn.z), fabsf(p.z - box->maxx.z));
    return dx * dx + dy * dy + dz * dz;
}
/* simple_knn.cu:131-145, K = 3 */
static inline void updateKBest3(v3 ref, v3 point, float* knn)
{
    float dx = point.x - ref.x, dy = point.y - ref.y, dz = point.z - ref.z;
    float dist = dx * dx + dy * dy + dz * dz;
    for (int j = 0; j < 3; j++)
        if (knn[j] > dist) { float t = knn[j]; knn[j] = dist; dist = t; }
}

/* SimpleKNN::knn, simple_knn.cu:185-221; distCUDA2, spatial.cu:15-26. */
void orc_knn_mean_dist2(int P, const float* pts, float* meanDists)
{
    const v3* points = (const v3*)pts;
    for (int i = 0; i < P; i++) meanDists[i] = 0.0f; /* torch::full({P}, 0.0) */
    if (P <= 0) return;
    v3 mn = {0, 0, 0}, mx = {0, 0, 0}; /* init = (0,0,0), simple_knn.cu:191-200 */
    for (int i = 0; i < P; i++) {
        mn.x = fminf_(mn.x, points[i].x); mn.y = fminf_(mn.y, points[i].y); mn.z = fminf_(mn.z, points[i].z);
        mx.x = fmaxf_(mx.x, points[i].x); mx.y = fmaxf_(mx.y, points[i].y); mx.z = fmaxf_(mx.z, points[i].z);
    }
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)P);
    uint32_t* idxs = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)P);
    uint64_t* keys_s = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)P);
    uint32_t* indices = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)P);
    for (int i = 0; i < P; i++) { keys[i] = coord2Morton(points[i], mn, mx); idxs[i] = (uint32_t)i; }
    radix_sort_pairs(P, keys, idxs, keys_s, indices, 32); /* stable, full 32 bits: simple_knn.cu:206-213 */
    uint32_t num_boxes = ((uint32_t)P + BOX_SIZE - 1) / BOX_SIZE;
    MinMax* boxes = (MinMax*)malloc(sizeof(MinMax) * num_boxes);
    for (uint32_t b = 0; b < num_boxes; b++) { /* boxMinMax, simple_knn.cu:78-117 */
        MinMax me = {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
        for (uint32_t i = b * BOX_SIZE; i < (uint32_t)P && i < (b + 1) * BOX_SIZE; i++) {
            v3 p = points[indices[i]];
            me.minn.x = fminf_(me.minn.x, p.x); me.minn.y = fminf_(me.minn.y, p.y); me.minn.z = fminf_(me.minn.z, p.z);
            me.maxx.x = fmaxf_(me.maxx.x, p.x); me.maxx.y = fmaxf_(me.maxx.y, p.y); me.maxx.z = fmaxf_(me.maxx.z, p.z);
        }
        boxes[b] = me;
    }
#pragma omp parallel for schedule(dynamic, 256)
    for (int idx = 0; idx < P; idx++) { /* boxMeanDist, simple_knn.cu:147-183 */
        v3 point = points[indices[idx]];
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        int lo = idx - 3 < 0 ? 0 : idx - 3, hi = idx + 3 > P - 1 ? P - 1 : idx + 3;
        for (int i = lo; i <= hi; i++) {
            if (i == idx) continue;
            updateKBest3(point, points[indices[i]], best);
        }
        float reject = best[2];
        best[0] = FLT_MAX; best[1] = FLT_MAX; best[2] = FLT_MAX;
        for (uint32_t b = 0; b < num_boxes; b++) {
            float dist = distBoxPoint(&boxes[b], point);
            if (dist > reject || dist > best[2]) continue;
            int e = (int)((b + 1) * BOX_SIZE) < P ? (int)((b + 1) * BOX_SIZE) : P;
            for (int i = (int)(b * BOX_SIZE); i < e; i++) {
                if (i == idx) continue;
                updateKBest3(point, points[indices[i]], best);
            }
        }
        meanDists[indices[idx]] = (best[0] + best[1] + best[2]) / 3.0f;
    }
    free(keys); free(idxs); free(keys_s); free(indices); free(boxes);
}

/* Brute-force 3-NN mean squared distance (no reference counterpart; used only
 * to cross-check orc_knn_mean_dist2 on small inputs). */
void orc_knn_bruteforce(int P, const float* pts, float* meanDists)
{
    const v3* points = (const v3*)pts;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            updateKBest3(points[i], points[j], best);
        }
        meanDists[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}

/* ========================================================================
 * Loss block of the training step: utils/loss_utils.py of the reference --
 * l1_loss (:17-18), gaussian/create_window (:23-31), ssim/_ssim (:33-63), combined as in train.py:116-117
 *     loss = (1 - lambda) * L1(img1, img2) + lambda * (1 - ssim(img1, img2)).
 * PINNED: the reference's loss_utils.py is importable; tests/golden/ref_loss.npz holds its outputs and autograd
 * gradients, and this restatement is checked against them.
 * The window is the reference's: 1-D Gaussian (sigma 1.5, 11 taps) normalised in float32, 2-D = outer product in
 * float32, zero padding (F.conv2d(padding=5, groups=C)).
 * ======================================================================== */
static const float SSIM_W1[11] = {0x1.0d956cp-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f, 0x1.b43c3ep-3f, 0x1.10656p-2f,
                                  0x1.b43c3ep-3f,  0x1.bff0fep-4f, 0x1.26eb18p-5f, 0x1.f1fe02p-8f, 0x1.0d956cp-10f};

/* out[0] = loss, out[1] = L1, out[2] = ssim.  dL_dimg1 (optional) = d loss / d img1.  Images are [C][H][W]. */
void orc_l1_ssim_loss(int C, int H, int W, const float* img1, const float* img2, float lambda, float* out, float* dL_dimg1)
{
    const size_t N = (size_t)C * H * W;
    float w2[11][11];
    for (int i = 0; i < 11; i++)
        for (int j = 0; j < 11; j++) w2[i][j] = SSIM_W1[i] * SSIM_W1[j]; /* _1D_window.mm(_1D_window.t()) */
    const float C1 = (float)(0.01 * 0.01), C2 = (float)(0.03 * 0.03);
    float* dmu1 = (float*)calloc(N, sizeof(float));
    float* de11 = (float*)calloc(N, sizeof(float));
    float* de12 = (float*)calloc(N, sizeof(float));
    double ssim_sum = 0.0, l1_sum = 0.0;
#pragma omp parallel for collapse(2) schedule(static) reduction(+ : ssim_sum, l1_sum)
    for (int c = 0; c < C; c++)
        for (int y = 0; y < H; y++) {
            const float* a = img1 + (size_t)c * H * W;
            const float* b = img2 + (size_t)c * H * W;
            for (int x = 0; x < W; x++) {
                float mu1 = 0, mu2 = 0, e11 = 0, e22 = 0, e12 = 0;
                for (int i = 0; i < 11; i++) {
                    const int yy = y + i - 5;
                    if (yy < 0 || yy >= H) continue;
                    for (int j = 0; j < 11; j++) {
                        const int xx = x + j - 5;
                        if (xx < 0 || xx >= W) continue;
                        const float p = a[(size_t)yy * W + xx], q = b[(size_t)yy * W + xx], w = w2[i][j];
                        mu1 += w * p; mu2 += w * q; e11 += w * (p * p); e22 += w * (q * q); e12 += w * (p * q);
                    }
                }
                const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
                const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu1_mu2;
                const float A = 2 * mu1_mu2 + C1, B = 2 * s12 + C2, Cc = mu1_sq + mu2_sq + C1, D = s1 + s2 + C2;
                const float m = (A * B) / (Cc * D);
                ssim_sum += m;
                const size_t id = (size_t)c * H * W + (size_t)y * W + x;
                l1_sum += fabsf(a[(size_t)y * W + x] - b[(size_t)y * W + x]);
                /* partial derivatives of the map value w.r.t. the five window sums (e11, e12, mu1 treated as inputs) */
                const float inv = 1.0f / (Cc * D);
                dmu1[id] = (2 * mu2 * B - 2 * mu2 * A) * inv - m * (2 * mu1 * D - 2 * mu1 * Cc) * inv;
                de11[id] = -m / D;
                de12[id] = 2 * A * inv;
            }
        }
    const float l1 = (float)(l1_sum / (double)N), ssim = (float)(ssim_sum / (double)N);
    out[0] = (1.0f - lambda) * l1 + lambda * (1.0f - ssim);
    out[1] = l1;
    out[2] = ssim;
    if (dL_dimg1) {
        const float gs = -lambda / (float)N, gl = (1.0f - lambda) / (float)N;
#pragma omp parallel for collapse(2) schedule(static)
        for (int c = 0; c < C; c++)
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    const size_t base = (size_t)c * H * W;
                    float s_mu = 0, s_11 = 0, s_12 = 0;
                    for (int i = 0; i < 11; i++) {
                        const int yy = y + i - 5;
                        if (yy < 0 || yy >= H) continue;
                        for (int j = 0; j < 11; j++) {
                            const int xx = x + j - 5;
                            if (xx < 0 || xx >= W) continue;
                            const size_t id = base + (size_t)yy * W + xx;
                            const float w = w2[i][j];
                            s_mu += w * dmu1[id]; s_11 += w * de11[id]; s_12 += w * de12[id];
                        }
                    }
                    const size_t id = base + (size_t)y * W + x;
                    const float d = img1[id] - img2[id];
                    const float sgn = d > 0 ? 1.0f : (d < 0 ? -1.0f : 0.0f);
                    dL_dimg1[id] = gs * (s_mu + 2 * img1[id] * s_11 + img2[id] * s_12) + gl * sgn;
                }
    }
    free(dmu1); free(de11); free(de12);
}

/* Thread count of the OpenMP loops above (bench.py's cpu_baseline times one thread next to all of them). */
void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

