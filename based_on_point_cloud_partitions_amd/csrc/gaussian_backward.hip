// Per-Gaussian backward for MI355X: sums the per-instance gradient rows written by the blend backward, then
// back-propagates through the conic/2D covariance (computeCov2DCUDA, cuda_rasterizer/backward.cu:144-274),
// the projection of the mean, the SH colour and the scale/rotation -> 3D covariance
// (preprocessCUDA + computeColorFromSH + computeCov3D, backward.cu:20-139,278-396) in ONE pass.
// Every output element is written exactly once (zeros for culled Gaussians), so the 300 B/Gaussian of
// torch::zeros the reference needs per call (rasterize_points.cu:151-159) disappear.
//
// SH rows are 192 B per Gaussian (AoS, [P][16][3]): read or written per thread they cost one memory instruction per
// dword with 64 distinct cache lines each.  The workgroup therefore moves its 256 rows through LDS: coalesced 16-byte
// loads of the SH tile in, per-thread reads with an odd row stride (bank-conflict free), the dL/dSH rows written back
// into the same LDS tile and streamed out with coalesced 16-byte stores.
#include "common.h"

namespace {

__constant__ float bSH_C0 = 0.28209479177387814f;
__constant__ float bSH_C1 = 0.4886025119029199f;
__constant__ float bSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f};
__constant__ float bSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                -0.5900435899266435f};

// The argument struct cannot carry __restrict__; without it every output store would fence the loads behind it
// (the compiler must assume dL_d* may alias shs / viewmatrix / ...), serialising ~10 HBM round trips per wave.
struct GaussBackwardPtrs {
    const float* __restrict__ means3D; const float* __restrict__ shs; const float* __restrict__ scales;
    const float* __restrict__ rotations; const float* __restrict__ cov3D_in; const float* __restrict__ viewmatrix;
    const float* __restrict__ projmatrix; const float* __restrict__ campos; const int* __restrict__ radii;
    const uint8_t* __restrict__ clamped; const float* __restrict__ splat; const uint32_t* __restrict__ tiles_touched;
    const float* __restrict__ grad_rows; const uint32_t* __restrict__ point_offsets; const uint32_t* __restrict__ row_offsets;
    float* __restrict__ dL_dmeans2D; float* __restrict__ dL_dcolors; float* __restrict__ dL_dopacity;
    float* __restrict__ dL_dmeans3D; float* __restrict__ dL_dcov3D; float* __restrict__ dL_dsh; float* __restrict__ dL_dscales;
    float* __restrict__ dL_drots;
    const float* __restrict__ features_dc; const float* __restrict__ features_rest;
    float* __restrict__ dL_dfeatures_dc; float* __restrict__ dL_dfeatures_rest;
    float* __restrict__ sums_out;
    const uint32_t* __restrict__ wg_first; const uint32_t* __restrict__ row_total;
    int P, D, M, has_shs, has_scales, lds_floats;
    float scale_modifier, tan_fovx, tan_fovy;
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_read(float v)  // the DPP-selected lane's v; 0 where the selection has no source or the row is masked
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}

// Balanced summation of the gradient rows of one pass (`cnt` <= BS consecutive Gaussians whose rows are the contiguous range
// [s_off[0], s_off[cnt]) of grad_rows; s_off[k] = first row of Gaussian k, both in LDS).  One LANE per ROW: the waves take equal
// shares of the rows (cut at Gaussian boundaries), walk them 64 at a time -- coalesced 16-byte loads, the next step's rows requested
// before this step is reduced -- and run a SEGMENTED inclusive scan over the 64 lanes (segments = Gaussians): four row_shr levels
// inside the rows of 16 lanes, row_bcast:15 and row_bcast:31 across them, a scalar carry from step to step.  The lane of a Gaussian's
// last row then holds its nine sums and stores them to s_sum[k][9].  Every Gaussian's rows are added in a tree whose shape depends
// only on the row numbering (bitwise reproducible), and the time of a pass is its number of rows / 64, however they are distributed
// over the Gaussians: the earlier scheme (every thread adds its own Gaussian's rows) took as long as the Gaussian with most rows,
// per chunk -- 2.2x slower on a spatially sorted model, whose workgroups hold either many-row or few-row Gaussians.
template <int BS>
__device__ __forceinline__ void reduce_rows_balanced(const float* __restrict__ grad_rows, const uint32_t* s_off, float* s_sum, int cnt)
{
    constexpr int NW = BS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t rb = s_off[0], re = s_off[cnt];
    if (rb == re) return;
    auto first_at = [&](uint32_t row) -> int {  // first k in [0, cnt] with s_off[k] >= row (wave-uniform)
        int lo = 0, hi = cnt;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_off[mid] >= row) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    const uint32_t share = (re - rb + NW - 1) / NW;
    const int ka = wave == 0 ? 0 : first_at(rb + wave * share);
    const int kb = wave == NW - 1 ? cnt : first_at(rb + (wave + 1) * share);
    const uint32_t ws = s_off[ka], we = s_off[kb];
    if (ws >= we) return;
    float carry[9];
#pragma unroll
    for (int i = 0; i < 9; i++) carry[i] = 0.f;
    bool cont = false;  // the segment of the previous step's last row goes on in this step
    float nx[9];
    auto request = [&](uint32_t base) {
        const uint32_t r = base + lane;
#pragma unroll
        for (int i = 0; i < 9; i++) nx[i] = 0.f;
        if (r < we) gsr_row_load(grad_rows, r, nx);
    };
    request(ws);
    const uint32_t l15 = lane & 15, l31 = lane & 31;
    for (uint32_t base = ws; base < we; base += 64) {
        float v[9];
#pragma unroll
        for (int i = 0; i < 9; i++) v[i] = nx[i];
        if (base + 64 < we) request(base + 64);
        const uint32_t r = base + lane;
        const bool valid = r < we;
        // owner: the last Gaussian k in [ka, kb) with s_off[k] <= r (Gaussians without rows share their successor's offset)
        int lo = ka, hi = kb;  // invariant: s_off[lo] <= r < s_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_off[mid] <= r) lo = mid; else hi = mid;
        }
        const uint32_t dist = valid ? r - s_off[lo] : 0u;  // rows back to the first row of this lane's Gaussian
        const bool ends = valid && r + 1 == s_off[lo + 1];
        if (lane == 0 && cont) {
#pragma unroll
            for (int i = 0; i < 9; i++) v[i] += carry[i];
        }
#pragma unroll
        for (int i = 0; i < 9; i++) {
            float t;
            t = dpp_read<0x111, 0xf>(v[i]); v[i] = dist >= 1u ? v[i] + t : v[i];    // row_shr:1
            t = dpp_read<0x112, 0xf>(v[i]); v[i] = dist >= 2u ? v[i] + t : v[i];    // row_shr:2
            t = dpp_read<0x114, 0xf>(v[i]); v[i] = dist >= 4u ? v[i] + t : v[i];    // row_shr:4
            t = dpp_read<0x118, 0xf>(v[i]); v[i] = dist >= 8u ? v[i] + t : v[i];    // row_shr:8
            t = dpp_read<0x142, 0xa>(v[i]); v[i] = dist > l15 ? v[i] + t : v[i];    // row_bcast:15 -> rows 1, 3
            t = dpp_read<0x143, 0xc>(v[i]); v[i] = dist > l31 ? v[i] + t : v[i];    // row_bcast:31 -> rows 2, 3
        }
        if (ends) {
            float* o = s_sum + lo * 9;
#pragma unroll
            for (int i = 0; i < 9; i++) o[i] = v[i];
        }
        cont = __builtin_amdgcn_readlane((int)(valid && !ends), 63) != 0;
#pragma unroll
        for (int i = 0; i < 9; i++) carry[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), 63));
    }
}

// RAW (gsr_backward_params): scales / rotations are the pre-activation parameters; the chain rule through exp / sigmoid /
// normalize is applied here and dL/dSH leaves LDS as the two tensors dL/dfeatures_dc, dL/dfeatures_rest -- the autograd
// passes of scene/gaussian_model.py:102-122 (exp, sigmoid, normalize, cat backward) never touch HBM.
// DIRECT (degree-3 storage, 16 coefficients; activated parameters or the raw features_dc / features_rest pair): a visible Gaussian reads its 192-byte SH row with twelve
// 16-byte loads at the start of its chain and stores its dL/dSH row the same way -- no SH tile in LDS, so the workgroup's LDS is
// the small gradient-row buffer alone and twice as many waves are in flight (the kernel is latency-bound: with half the waves
// it takes 1.5x as long).  Culled Gaussians read no SH at all.
// BAL: the cost-balanced work split and the lane-per-row summation (the default); !BAL: BS consecutive Gaussians per workgroup, every
// thread adds its own Gaussian's rows (GSR_GB_BALANCED=0, and scenes without any instance).
#ifndef GSR_GB_WPE_MAX
#define GSR_GB_WPE_MAX 8
#endif
#ifndef GSR_GB_WPE
#define GSR_GB_WPE 0  // experiment: waves per SIMD the register allocation aims at (0: the compiler's choice, 4 at 120 VGPRs)
#endif
template <bool RAW, int BS, bool DIRECT, int RK, bool BAL>
__global__ void __launch_bounds__(BS)
#if GSR_GB_WPE
__attribute__((amdgpu_waves_per_eu(GSR_GB_WPE, GSR_GB_WPE_MAX)))
#endif
gaussian_backward_kernel(GaussBackwardPtrs a, float h_x, float h_y)
{
    extern __shared__ float s_tile[];  // [BS][3M + 1]
    // this workgroup's Gaussians: the cost-balanced split stored by the forward (gsr_part_*, common.h), or BS consecutive ones
    int w_first, w_last;
    if (BAL) {
        const uint32_t total = a.row_total[0];
        const uint32_t nw = (uint32_t)(((unsigned long long)GSR_PART_W * (unsigned long long)a.P + total) >> gsr_part_shift(a.P, total)) + 1u;
        if (blockIdx.x >= nw) return;
        w_first = (int)a.wg_first[blockIdx.x];
        w_last = blockIdx.x + 1 < nw ? (int)a.wg_first[blockIdx.x + 1] : a.P;
    } else {
        w_first = blockIdx.x * BS;
        w_last = min(a.P, w_first + BS);
    }
    const int M = a.M, n3 = 3 * M, stride = n3 + 1;
    // blockIdx.y: a workgroup of the split holds more than BS Gaussians only when the scene has more than ~8 gradient rows per Gaussian
    // (gsr_part_shift > 10); its Gaussians are then shared among the workgroups of its grid column, BS each
    const int gbase = w_first + (int)blockIdx.y * BS;
    if (gbase >= w_last) return;
    {
    const int idx = gbase + threadIdx.x;
    const int rows = min(BS, w_last - gbase);
    const size_t tile_base = (size_t)gbase * n3;
    const bool valid = (int)threadIdx.x < rows;
    const bool visible = valid && a.radii[idx] > 0;

    // the Gaussian's own parameters: requested here so that they arrive while the gradient rows are summed (the chain below would
    // otherwise start with one more exposed round trip per workgroup)
    float mx = 0.f, my = 0.f, mz = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, c4 = 0.f, c5 = 0.f;
    float4 q_in = make_float4(0.f, 0.f, 0.f, 0.f);
    float sc_in[3] = {0.f, 0.f, 0.f};
    if (visible) {
        mx = a.means3D[3 * idx]; my = a.means3D[3 * idx + 1]; mz = a.means3D[3 * idx + 2];
        const float* cov3D = a.cov3D_in + 6 * (size_t)idx;
        c0 = cov3D[0]; c1 = cov3D[1]; c2 = cov3D[2]; c3 = cov3D[3]; c4 = cov3D[4]; c5 = cov3D[5];
        if (a.has_scales) {
            q_in = reinterpret_cast<const float4*>(a.rotations)[idx];
            sc_in[0] = a.scales[3 * idx]; sc_in[1] = a.scales[3 * idx + 1]; sc_in[2] = a.scales[3 * idx + 2];
        }
    }

    // ---- sum each Gaussian's gradient rows ----
    // The rows of this workgroup's Gaussians are one contiguous range of grad_rows (emission order).  They are streamed
    // through LDS in chunks with coalesced 16-byte loads; every thread then adds the rows of its own Gaussian from LDS in
    // ascending row order (fixed order -> bitwise reproducible).  Reading them per thread straight from HBM serialises
    // one dependent round trip per row and was 60 % of this kernel.
    float g[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (BAL) {
        // s_off[k] = first gradient row of Gaussian gbase + k (exclusive prefix of the row counts), s_off[rows] = the end
        uint32_t* s_off = reinterpret_cast<uint32_t*>(s_tile);
        float* s_sum = s_tile + BS + 1;
        if ((int)threadIdx.x <= rows) {
            const int i = gbase + (int)threadIdx.x;
            s_off[threadIdx.x] = i == 0 ? 0u : a.row_offsets[i - 1];
        }
        if (BS + threadIdx.x <= (unsigned)rows) s_off[BS] = a.row_offsets[gbase + BS - 1];  // (thread 0 when the pass is full)
        __syncthreads();
        reduce_rows_balanced<BS>(a.grad_rows, s_off, s_sum, rows);
        __syncthreads();
        if (valid && s_off[threadIdx.x + 1] != s_off[threadIdx.x]) {
#pragma unroll
            for (int i = 0; i < 9; i++) g[i] = s_sum[threadIdx.x * 9 + i];
        }
        __syncthreads();  // s_tile is reused below
    } else {
        const int first = gbase, last = gbase + rows;
        uint32_t begin, end, my_off = 0, my_n = 0;
        // one row per reachable instance, numbered by the preprocess kernel's tile masks (render_backward.hip)
        begin = first == 0 ? 0u : a.row_offsets[first - 1];
        end = a.row_offsets[last - 1];
        if (visible) {
            const float4 q3 = *reinterpret_cast<const float4*>(a.splat + (size_t)idx * GSR_SPLAT_F + 12);
            my_off = __float_as_uint(q3.x);
            my_n = gsr_splat_rows(__float_as_uint(q3.y), __float_as_uint(q3.z), __float_as_uint(q3.w), a.tiles_touched[idx]);
        }
        if (GSR_GROW_F != 12) {  // 36-byte rows: plain dword copies of a chunk, every thread adds its own Gaussian's rows
            const uint32_t CHg = (uint32_t)a.lds_floats / GSR_GROW_F;
            for (uint32_t cb = begin; cb < end; cb += CHg) {
                const uint32_t nrows = min(CHg, end - cb);
                for (uint32_t f = threadIdx.x; f < nrows * GSR_GROW_F; f += BS) s_tile[f] = a.grad_rows[(size_t)cb * GSR_GROW_F + f];
                __syncthreads();
                const uint32_t lo = max(my_off, cb), hi = min(my_off + my_n, cb + nrows);
                for (uint32_t r = lo; r < hi; r++)
#pragma unroll
                    for (int i = 0; i < 9; i++) g[i] += s_tile[(r - cb) * GSR_GROW_F + i];
                __syncthreads();
            }
        } else {
        const uint32_t CH = DIRECT ? (uint32_t)(RK * BS) : (uint32_t)a.lds_floats / GSR_GROW_F;  // rows per chunk
        float4* s_rows = reinterpret_cast<float4*>(s_tile);
        const float4* g4 = reinterpret_cast<const float4*>(a.grad_rows);
        // DIRECT: chunks of RK*BS rows = 3*RK float4 per thread, the next chunk requested into registers before this one is summed
        constexpr int NPRE = 3 * RK;
        float4 pre[DIRECT ? NPRE : 1];
        auto request = [&](uint32_t cb) {
            const uint32_t n3r = min(CH, end - cb) * 3;
#pragma unroll
            for (int t = 0; t < NPRE; t++) {
                const uint32_t f = threadIdx.x + t * BS;
                pre[DIRECT ? t : 0] = f < n3r ? g4[(size_t)cb * 3 + f] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        if (DIRECT && begin < end) request(begin);
        for (uint32_t cb = begin; cb < end; cb += CH) {
            const uint32_t nrows = min(CH, end - cb);
            if (DIRECT) {
#pragma unroll
                for (int t = 0; t < NPRE; t++) {
                    const uint32_t f = threadIdx.x + t * BS;
                    if (f < nrows * 3) s_rows[f] = pre[DIRECT ? t : 0];
                }
            } else {
                for (uint32_t f = threadIdx.x; f < nrows * 3; f += BS) s_rows[f] = g4[(size_t)cb * 3 + f];
            }
            __syncthreads();
            if (DIRECT && cb + CH < end) request(cb + CH);
            const uint32_t lo = max(my_off, cb), hi = min(my_off + my_n, cb + nrows);
            for (uint32_t r = lo; r < hi; r++) {
                const float4 r0 = s_rows[(r - cb) * 3], r1 = s_rows[(r - cb) * 3 + 1], r2 = s_rows[(r - cb) * 3 + 2];
                g[0] += r0.x; g[1] += r0.y; g[2] += r0.z; g[3] += r0.w;
                g[4] += r1.x; g[5] += r1.y; g[6] += r1.z; g[7] += r1.w;
                g[8] += r2.x;
            }
            __syncthreads();
        }
        }
    }

    if (RAW && !DIRECT) {
        sh_cols_load(a.features_dc + (size_t)gbase * 3, s_tile, rows, 3, stride, 0);
        if (n3 > 3) sh_cols_load(a.features_rest + (size_t)gbase * (n3 - 3), s_tile, rows, n3 - 3, stride, 3);
    } else if (!DIRECT && a.has_shs) sh_tile_load(a.shs + tile_base, s_tile, rows, n3, stride);
    if (!DIRECT) __syncthreads();
    float shr[DIRECT ? 48 : 1];
    float* __restrict__ my_row = s_tile + threadIdx.x * stride;  // SH in, dL/dSH out (not DIRECT)
    // DIRECT: dL/dSH = basis weight x dL/dRGB, zero for a culled Gaussian; the rows leave through LDS after the chain (below)
    float w[DIRECT ? 16 : 1], dRGBo[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (DIRECT ? 16 : 1); i++) w[i] = 0.f;

    if (visible) {

    float* __restrict__ o_m2 = a.dL_dmeans2D + 3 * (size_t)idx;
    float* __restrict__ o_col = a.dL_dcolors + 3 * (size_t)idx;
    float* __restrict__ o_m3 = a.dL_dmeans3D + 3 * (size_t)idx;
    float* __restrict__ o_cov = a.dL_dcov3D + 6 * (size_t)idx;
    float* __restrict__ o_sh = my_row;
    float* __restrict__ o_sc = a.dL_dscales + 3 * (size_t)idx;
    float* __restrict__ o_rot = a.dL_drots + 4 * (size_t)idx;

    const float dL_dmean2D_x = g[GR_MX], dL_dmean2D_y = g[GR_MY];
    const float dLc_x = g[GR_CX], dLc_y = g[GR_CY], dLc_w = g[GR_CW];
    float dL_dcol[3] = {g[GR_R], g[GR_G], g[GR_B]};
    if (a.sums_out) {  // test hook: exactly the sums the chain below consumes
        float* so = a.sums_out + (size_t)idx * GSR_SUMS_F;
#pragma unroll
        for (int i = 0; i < 9; i++) so[i] = g[i];
    }
    o_m2[0] = dL_dmean2D_x; o_m2[1] = dL_dmean2D_y; o_m2[2] = 0.f;
    if (!RAW || a.dL_dcolors) { o_col[0] = dL_dcol[0]; o_col[1] = dL_dcol[1]; o_col[2] = dL_dcol[2]; }
    if (RAW) {  // d sigmoid: y (1 - y), y = the activated opacity kept in the splat record
        const float opa = a.splat[(size_t)idx * GSR_SPLAT_F + SP_OPA];
        a.dL_dopacity[idx] = g[GR_OPA] * ((1.0f - opa) * opa);
    } else
        a.dL_dopacity[idx] = g[GR_OPA];

    const float* v = a.viewmatrix;
    const float* proj = a.projmatrix;

    // ---- computeCov2DCUDA (backward.cu:144-274) ----
    float tx = v[0] * mx + v[4] * my + v[8] * mz + v[12];
    float ty = v[1] * mx + v[5] * my + v[9] * mz + v[13];
    const float tz_ = v[2] * mx + v[6] * my + v[10] * mz + v[14];
    const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
    const float txtz = tx / tz_, tytz = ty / tz_;
    tx = fminf(limx, fmaxf(-limx, txtz)) * tz_;
    ty = fminf(limy, fmaxf(-limy, tytz)) * tz_;
    const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float J00 = h_x / tz_, J02 = -(h_x * tx) / (tz_ * tz_);
    const float J11 = h_y / tz_, J12 = -(h_y * ty) / (tz_ * tz_);
    // GLM T[c][r]: T0r / T1r
    const float T00 = v[0] * J00 + v[1] * 0.0f + v[2] * J02;
    const float T01 = v[4] * J00 + v[5] * 0.0f + v[6] * J02;
    const float T02 = v[8] * J00 + v[9] * 0.0f + v[10] * J02;
    const float T10 = v[0] * 0.0f + v[1] * J11 + v[2] * J12;
    const float T11 = v[4] * 0.0f + v[5] * J11 + v[6] * J12;
    const float T12 = v[8] * 0.0f + v[9] * J11 + v[10] * J12;
    const float A00 = T00 * c0 + T01 * c1 + T02 * c2;
    const float A10 = T00 * c1 + T01 * c3 + T02 * c4;
    const float A20 = T00 * c2 + T01 * c4 + T02 * c5;
    const float A01 = T10 * c0 + T11 * c1 + T12 * c2;
    const float A11 = T10 * c1 + T11 * c3 + T12 * c4;
    const float A21 = T10 * c2 + T11 * c4 + T12 * c5;
    const float ca = (A00 * T00 + A10 * T01 + A20 * T02) + 0.3f;
    const float cb = A01 * T00 + A11 * T01 + A21 * T02;
    const float cc = (A01 * T10 + A11 * T11 + A21 * T12) + 0.3f;
    const float denom = ca * cc - cb * cb;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dcv[6];
    if (denom2inv != 0) {
        dL_da = denom2inv * (-cc * cc * dLc_x + 2 * cb * cc * dLc_y + (denom - ca * cc) * dLc_w);
        dL_dc = denom2inv * (-ca * ca * dLc_w + 2 * ca * cb * dLc_y + (denom - ca * cc) * dLc_x);
        dL_db = denom2inv * 2 * (cb * cc * dLc_x - (denom + 2 * cb * cb) * dLc_y + ca * cb * dLc_w);
        dcv[0] = (T00 * T00 * dL_da + T00 * T10 * dL_db + T10 * T10 * dL_dc);
        dcv[3] = (T01 * T01 * dL_da + T01 * T11 * dL_db + T11 * T11 * dL_dc);
        dcv[5] = (T02 * T02 * dL_da + T02 * T12 * dL_db + T12 * T12 * dL_dc);
        dcv[1] = 2 * T00 * T01 * dL_da + (T00 * T11 + T01 * T10) * dL_db + 2 * T10 * T11 * dL_dc;
        dcv[2] = 2 * T00 * T02 * dL_da + (T00 * T12 + T02 * T10) * dL_db + 2 * T10 * T12 * dL_dc;
        dcv[4] = 2 * T02 * T01 * dL_da + (T01 * T12 + T02 * T11) * dL_db + 2 * T11 * T12 * dL_dc;
    } else {
        for (int i = 0; i < 6; i++) dcv[i] = 0;
    }
    if (!RAW || a.dL_dcov3D)
        for (int i = 0; i < 6; i++) o_cov[i] = dcv[i];
    // Vrk is symmetric: Vrk[0]=(c0,c1,c2), Vrk[1]=(c1,c3,c4), Vrk[2]=(c2,c4,c5)
    const float dL_dT00 = 2 * (T00 * c0 + T01 * c1 + T02 * c2) * dL_da + (T10 * c0 + T11 * c1 + T12 * c2) * dL_db;
    const float dL_dT01 = 2 * (T00 * c1 + T01 * c3 + T02 * c4) * dL_da + (T10 * c1 + T11 * c3 + T12 * c4) * dL_db;
    const float dL_dT02 = 2 * (T00 * c2 + T01 * c4 + T02 * c5) * dL_da + (T10 * c2 + T11 * c4 + T12 * c5) * dL_db;
    const float dL_dT10 = 2 * (T10 * c0 + T11 * c1 + T12 * c2) * dL_dc + (T00 * c0 + T01 * c1 + T02 * c2) * dL_db;
    const float dL_dT11 = 2 * (T10 * c1 + T11 * c3 + T12 * c4) * dL_dc + (T00 * c1 + T01 * c3 + T02 * c4) * dL_db;
    const float dL_dT12 = 2 * (T10 * c2 + T11 * c4 + T12 * c5) * dL_dc + (T00 * c2 + T01 * c4 + T02 * c5) * dL_db;
    // W[c][r] = v[c + 4r]
    const float dL_dJ00 = v[0] * dL_dT00 + v[4] * dL_dT01 + v[8] * dL_dT02;
    const float dL_dJ02 = v[2] * dL_dT00 + v[6] * dL_dT01 + v[10] * dL_dT02;
    const float dL_dJ11 = v[1] * dL_dT10 + v[5] * dL_dT11 + v[9] * dL_dT12;
    const float dL_dJ12 = v[2] * dL_dT10 + v[6] * dL_dT11 + v[10] * dL_dT12;
    const float tz = 1.f / tz_;
    const float tz2 = tz * tz;
    const float tz3 = tz2 * tz;
    const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * tx) * tz3 * dL_dJ02 + (2 * h_y * ty) * tz3 * dL_dJ12;
    float dmean_x = v[0] * dL_dtx + v[1] * dL_dty + v[2] * dL_dtz;
    float dmean_y = v[4] * dL_dtx + v[5] * dL_dty + v[6] * dL_dtz;
    float dmean_z = v[8] * dL_dtx + v[9] * dL_dty + v[10] * dL_dtz;

    // ---- preprocessCUDA backward (backward.cu:346-396) ----
    const float m_homw = proj[3] * mx + proj[7] * my + proj[11] * mz + proj[15];
    const float m_w = 1.0f / (m_homw + 0.0000001f);
    const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * m_w * m_w;
    dmean_x += (proj[0] * m_w - proj[3] * mul1) * dL_dmean2D_x + (proj[1] * m_w - proj[3] * mul2) * dL_dmean2D_y;
    dmean_y += (proj[4] * m_w - proj[7] * mul1) * dL_dmean2D_x + (proj[5] * m_w - proj[7] * mul2) * dL_dmean2D_y;
    dmean_z += (proj[8] * m_w - proj[11] * mul1) * dL_dmean2D_x + (proj[9] * m_w - proj[11] * mul2) * dL_dmean2D_y;

    if (DIRECT || a.has_shs) {
        // computeColorFromSH backward (backward.cu:20-139)
        if (DIRECT && RAW) {  // 3 + 45 floats from the two raw tensors (rows of 12 and 180 bytes, dword-aligned pieces)
            const gsr_f3u d = *reinterpret_cast<const gsr_f3u*>(a.features_dc + (size_t)idx * 3);
            shr[0] = d.x; shr[1] = d.y; shr[2] = d.z;
            const float* row_in = a.features_rest + (size_t)idx * 45;
#pragma unroll
            for (int q = 0; q < 11; q++) {
                const gsr_f4u t = *reinterpret_cast<const gsr_f4u*>(row_in + 4 * q);
                shr[3 + 4 * q] = t.x; shr[4 + 4 * q] = t.y; shr[5 + 4 * q] = t.z; shr[6 + 4 * q] = t.w;
            }
            shr[DIRECT ? 47 : 0] = row_in[44];
        } else if (DIRECT) {
            const float4* row_in = reinterpret_cast<const float4*>(a.shs + (size_t)idx * 48);
#pragma unroll
            for (int q = 0; q < 12; q++) {
                const float4 t = row_in[q];
                shr[4 * q] = t.x; shr[4 * q + 1] = t.y; shr[4 * q + 2] = t.z; shr[4 * q + 3] = t.w;
            }
        }
        const int deg = a.D;
        const float dox = mx - a.campos[0], doy = my - a.campos[1], doz = mz - a.campos[2];
        const float len = sqrtf(dox * dox + doy * doy + doz * doz);
        const float x = dox / len, y = doy / len, z = doz / len;
        const float* sh = DIRECT ? shr : my_row;
        float dRGB[3];
        for (int ch = 0; ch < 3; ch++) dRGB[ch] = dL_dcol[ch] * (a.clamped[3 * idx + ch] ? 0.f : 1.f);
        float dx_[3] = {0, 0, 0}, dy_[3] = {0, 0, 0}, dz_[3] = {0, 0, 0};
#define SH(i) sh[3 * (i) + ch]
#define OUT(i, w) { const float w_ = (w); o_sh[3 * (i)] = w_ * dRGB[0]; o_sh[3 * (i) + 1] = w_ * dRGB[1]; o_sh[3 * (i) + 2] = w_ * dRGB[2]; }
        // first every read of the SH row (it is overwritten in place by dL/dSH below)
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        if (deg > 0) {
            for (int ch = 0; ch < 3; ch++) {
                dx_[ch] = -bSH_C1 * SH(3);
                dy_[ch] = -bSH_C1 * SH(1);
                dz_[ch] = bSH_C1 * SH(2);
            }
            if (deg > 1) {
                for (int ch = 0; ch < 3; ch++) {
                    dx_[ch] += bSH_C2[0] * y * SH(4) + bSH_C2[2] * 2.f * -x * SH(6) + bSH_C2[3] * z * SH(7) + bSH_C2[4] * 2.f * x * SH(8);
                    dy_[ch] += bSH_C2[0] * x * SH(4) + bSH_C2[1] * z * SH(5) + bSH_C2[2] * 2.f * -y * SH(6) + bSH_C2[4] * 2.f * -y * SH(8);
                    dz_[ch] += bSH_C2[1] * y * SH(5) + bSH_C2[2] * 2.f * 2.f * z * SH(6) + bSH_C2[3] * x * SH(7);
                }
                if (deg > 2) {
                    for (int ch = 0; ch < 3; ch++) {
                        dx_[ch] += (bSH_C3[0] * SH(9) * 3.f * 2.f * xy + bSH_C3[1] * SH(10) * yz + bSH_C3[2] * SH(11) * -2.f * xy +
                                    bSH_C3[3] * SH(12) * -3.f * 2.f * xz + bSH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                                    bSH_C3[5] * SH(14) * 2.f * xz + bSH_C3[6] * SH(15) * 3.f * (xx - yy));
                        dy_[ch] += (bSH_C3[0] * SH(9) * 3.f * (xx - yy) + bSH_C3[1] * SH(10) * xz +
                                    bSH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) + bSH_C3[3] * SH(12) * -3.f * 2.f * yz +
                                    bSH_C3[4] * SH(13) * -2.f * xy + bSH_C3[5] * SH(14) * -2.f * yz +
                                    bSH_C3[6] * SH(15) * -3.f * 2.f * xy);
                        dz_[ch] += (bSH_C3[1] * SH(10) * xy + bSH_C3[2] * SH(11) * 4.f * 2.f * yz +
                                    bSH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) + bSH_C3[4] * SH(13) * 4.f * 2.f * xz +
                                    bSH_C3[5] * SH(14) * (xx - yy));
                    }
                }
            }
        }
        if (DIRECT) {
            // dL/dSH = dRGB/dSH * dL/dRGB: the 16 basis weights (zero above the active degree)
            dRGBo[0] = dRGB[0]; dRGBo[1] = dRGB[1]; dRGBo[2] = dRGB[2];
            w[0] = bSH_C0;
            if (deg > 0) {
                w[1] = -bSH_C1 * y; w[2] = bSH_C1 * z; w[3] = -bSH_C1 * x;
                if (deg > 1) {
                    w[4] = bSH_C2[0] * xy; w[5] = bSH_C2[1] * yz; w[6] = bSH_C2[2] * (2.f * zz - xx - yy);
                    w[7] = bSH_C2[3] * xz; w[8] = bSH_C2[4] * (xx - yy);
                    if (deg > 2) {
                        w[9] = bSH_C3[0] * y * (3.f * xx - yy); w[10] = bSH_C3[1] * xy * z;
                        w[11] = bSH_C3[2] * y * (4.f * zz - xx - yy); w[12] = bSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                        w[13] = bSH_C3[4] * x * (4.f * zz - xx - yy); w[14] = bSH_C3[5] * z * (xx - yy);
                        w[15] = bSH_C3[6] * x * (xx - 3.f * yy);
                    }
                }
            }
        } else {
        // then dL/dSH = dRGB/dSH * dL/dRGB into the same row
        OUT(0, bSH_C0);
        int written = 1;
        if (deg > 0) {
            OUT(1, -bSH_C1 * y); OUT(2, bSH_C1 * z); OUT(3, -bSH_C1 * x);
            written = 4;
            if (deg > 1) {
                OUT(4, bSH_C2[0] * xy); OUT(5, bSH_C2[1] * yz); OUT(6, bSH_C2[2] * (2.f * zz - xx - yy));
                OUT(7, bSH_C2[3] * xz); OUT(8, bSH_C2[4] * (xx - yy));
                written = 9;
                if (deg > 2) {
                    OUT(9, bSH_C3[0] * y * (3.f * xx - yy)); OUT(10, bSH_C3[1] * xy * z);
                    OUT(11, bSH_C3[2] * y * (4.f * zz - xx - yy)); OUT(12, bSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                    OUT(13, bSH_C3[4] * x * (4.f * zz - xx - yy)); OUT(14, bSH_C3[5] * z * (xx - yy));
                    OUT(15, bSH_C3[6] * x * (xx - 3.f * yy));
                    written = 16;
                }
            }
        }
#undef SH
#undef OUT
        if (!DIRECT)
            for (int i = 3 * written; i < 3 * M; i++) o_sh[i] = 0.f;  // coefficients above the active degree
        }
        const float ddx = dx_[0] * dRGB[0] + dx_[1] * dRGB[1] + dx_[2] * dRGB[2];
        const float ddy = dy_[0] * dRGB[0] + dy_[1] * dRGB[1] + dy_[2] * dRGB[2];
        const float ddz = dz_[0] * dRGB[0] + dz_[1] * dRGB[1] + dz_[2] * dRGB[2];
        // dnormvdv (auxiliary.h:107-117)
        const float sum2 = dox * dox + doy * doy + doz * doz;
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmean_x += ((+sum2 - dox * dox) * ddx - doy * dox * ddy - doz * dox * ddz) * invsum32;
        dmean_y += (-dox * doy * ddx + (sum2 - doy * doy) * ddy - doz * doy * ddz) * invsum32;
        dmean_z += (-dox * doz * ddx - doy * doz * ddy + (sum2 - doz * doz) * ddz) * invsum32;
    } else if (!DIRECT) {
        for (int i = 0; i < 3 * M; i++) o_sh[i] = 0.f;
    }
    o_m3[0] = dmean_x; o_m3[1] = dmean_y; o_m3[2] = dmean_z;

    if (a.has_scales) {
        // computeCov3D backward (backward.cu:278-341)
        float4 q = q_in;
        float sc[3] = {sc_in[0], sc_in[1], sc_in[2]};
        float qn = 1.0f;
        if (RAW) {
            sc[0] = gsr_expf(sc[0]); sc[1] = gsr_expf(sc[1]); sc[2] = gsr_expf(sc[2]);
            qn = gsr_quat_norm(q.x, q.y, q.z, q.w);
            const float d = fmaxf(qn, 1e-12f);
            q.x = q.x / d; q.y = q.y / d; q.z = q.z / d; q.w = q.w / d;
        }
        const float r = q.x, x = q.y, y = q.z, z = q.w;
        const float Rq[3][3] = {
            {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
            {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
            {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
        const float s[3] = {a.scale_modifier * sc[0], a.scale_modifier * sc[1], a.scale_modifier * sc[2]};
        float Mm[3][3];
        for (int c = 0; c < 3; c++)
            for (int k = 0; k < 3; k++) Mm[c][k] = s[k] * Rq[c][k];
        const float dS[3][3] = {{dcv[0], 0.5f * dcv[1], 0.5f * dcv[2]}, {0.5f * dcv[1], dcv[3], 0.5f * dcv[4]},
                                {0.5f * dcv[2], 0.5f * dcv[4], dcv[5]}};
        float dMt[3][3];  // dL_dMt[c][r] = dL_dM[r][c];  dL_dM[c][r] = sum_k (2M)[k][r]*dS[c][k]
        for (int c = 0; c < 3; c++)
            for (int rr = 0; rr < 3; rr++)
                dMt[rr][c] = (2.0f * Mm[0][rr]) * dS[c][0] + (2.0f * Mm[1][rr]) * dS[c][1] + (2.0f * Mm[2][rr]) * dS[c][2];
        for (int c = 0; c < 3; c++) {
            const float ds = Rq[0][c] * dMt[c][0] + Rq[1][c] * dMt[c][1] + Rq[2][c] * dMt[c][2];
            o_sc[c] = RAW ? ds * sc[c] : ds;  // d exp: y
        }
        for (int c = 0; c < 3; c++)
            for (int rr = 0; rr < 3; rr++) dMt[c][rr] *= s[c];
        float4 dq;
        dq.x = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
        dq.y = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
        dq.z = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
        dq.w = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
        if (RAW) {  // d normalize: (g - q (q.g)) / ||v||, or g / eps below the clamp
            if (qn > 1e-12f) {
                const float dot = ((q.x * dq.x + q.y * dq.y) + q.z * dq.z) + q.w * dq.w;
                dq.x = (dq.x - q.x * dot) / qn; dq.y = (dq.y - q.y * dot) / qn;
                dq.z = (dq.z - q.z * dot) / qn; dq.w = (dq.w - q.w * dot) / qn;
            } else {
                dq.x = dq.x / 1e-12f; dq.y = dq.y / 1e-12f; dq.z = dq.z / 1e-12f; dq.w = dq.w / 1e-12f;
            }
        }
        reinterpret_cast<float4*>(o_rot)[0] = dq;
    } else {
        o_sc[0] = 0.f; o_sc[1] = 0.f; o_sc[2] = 0.f;
        reinterpret_cast<float4*>(o_rot)[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    } else if (valid) {  // culled: every output is zero
        float* __restrict__ o_m2 = a.dL_dmeans2D + 3 * (size_t)idx;
        float* __restrict__ o_col = a.dL_dcolors + 3 * (size_t)idx;
        float* __restrict__ o_m3 = a.dL_dmeans3D + 3 * (size_t)idx;
        float* __restrict__ o_cov = a.dL_dcov3D + 6 * (size_t)idx;
        float* __restrict__ o_sc = a.dL_dscales + 3 * (size_t)idx;
        o_m2[0] = 0.f; o_m2[1] = 0.f; o_m2[2] = 0.f;
        if (!RAW || a.dL_dcolors) { o_col[0] = 0.f; o_col[1] = 0.f; o_col[2] = 0.f; }
        a.dL_dopacity[idx] = 0.f;
        o_m3[0] = 0.f; o_m3[1] = 0.f; o_m3[2] = 0.f;
        if (!RAW || a.dL_dcov3D)
            for (int i = 0; i < 6; i++) o_cov[i] = 0.f;
        if (!DIRECT)
            for (int i = 0; i < n3; i++) my_row[i] = 0.f;
        o_sc[0] = 0.f; o_sc[1] = 0.f; o_sc[2] = 0.f;
        reinterpret_cast<float4*>(a.dL_drots + 4 * (size_t)idx)[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (DIRECT) {
        // the dL/dSH rows in three pieces of 16 floats: products into an LDS tile of [BS][17] (the gradient-row buffer is free by
        // now), out with coalesced 16-byte stores (64 contiguous bytes per row and piece)
        float* g_out = RAW ? nullptr : a.dL_dsh + tile_base;
#pragma unroll
        for (int c = 0; c < 3; c++) {
#pragma unroll
            for (int k = 0; k < 16; k++) s_tile[threadIdx.x * 17 + k] = w[(16 * c + k) / 3] * dRGBo[(16 * c + k) % 3];
            __syncthreads();
            if (RAW) {  // the row splits into dL/dfeatures_dc (floats 0..2) and dL/dfeatures_rest (3..47): coalesced dword stores
                float* o_dc = a.dL_dfeatures_dc + (size_t)gbase * 3;
                float* o_rest = a.dL_dfeatures_rest + (size_t)gbase * 45;
                for (int f = threadIdx.x; f < rows * 16; f += BS) {
                    const int r = f >> 4, j = 16 * c + (f & 15);
                    const float v = s_tile[r * 17 + (f & 15)];
                    if (j < 3) o_dc[r * 3 + j] = v;
                    else o_rest[r * 45 + j - 3] = v;
                }
            } else {
                for (int f = threadIdx.x; f < rows * 4; f += BS) {
                    const int r = f >> 2, q = f & 3;
                    const float* sp = s_tile + r * 17 + 4 * q;
                    *reinterpret_cast<float4*>(g_out + (size_t)r * 48 + 16 * c + 4 * q) = make_float4(sp[0], sp[1], sp[2], sp[3]);
                }
            }
            __syncthreads();
        }
        return;
    }
    __syncthreads();
    if (RAW) {
        sh_cols_store(a.dL_dfeatures_dc + (size_t)gbase * 3, s_tile, rows, 3, stride, 0);
        if (n3 > 3) sh_cols_store(a.dL_dfeatures_rest + (size_t)gbase * (n3 - 3), s_tile, rows, n3 - 3, stride, 3);
    } else if (n3 > 0) sh_tile_store(a.dL_dsh + tile_base, s_tile, rows, n3, stride);
    }
}

}  // namespace

void gsr_launch_gaussian_backward(const GaussBackwardArgs& a, hipStream_t s)
{
    const float h_y = a.H / (2.0f * a.tan_fovy);  // rasterizer_impl.cu:381-382
    const float h_x = a.W / (2.0f * a.tan_fovx);
    GaussBackwardPtrs p;
    p.means3D = a.means3D; p.shs = a.shs; p.scales = a.scales; p.rotations = a.rotations; p.cov3D_in = a.cov3D_in;
    p.viewmatrix = a.viewmatrix; p.projmatrix = a.projmatrix; p.campos = a.campos; p.radii = a.radii; p.clamped = a.clamped;
    p.splat = a.splat; p.tiles_touched = a.tiles_touched; p.grad_rows = a.grad_rows; p.point_offsets = a.point_offsets;
    p.row_offsets = a.row_offsets;
    p.dL_dmeans2D = a.dL_dmeans2D; p.dL_dcolors = a.dL_dcolors; p.dL_dopacity = a.dL_dopacity; p.dL_dmeans3D = a.dL_dmeans3D;
    p.dL_dcov3D = a.dL_dcov3D; p.dL_dsh = a.dL_dsh; p.dL_dscales = a.dL_dscales; p.dL_drots = a.dL_drots;
    p.features_dc = a.features_dc; p.features_rest = a.features_rest;
    p.dL_dfeatures_dc = a.dL_dfeatures_dc; p.dL_dfeatures_rest = a.dL_dfeatures_rest;
    p.sums_out = a.sums_out;
    p.wg_first = a.wg_first; p.row_total = a.row_total;
    p.P = a.P; p.D = a.D; p.M = a.M; p.has_shs = a.has_shs; p.has_scales = a.has_scales;
    p.scale_modifier = a.scale_modifier; p.tan_fovx = a.tan_fovx; p.tan_fovy = a.tan_fovy;
    const int bs = gsr_variant("GB_BLOCK", 128);
    const bool direct = (a.raw ? (a.features_rest != nullptr && a.dL_dfeatures_rest != nullptr) : a.has_shs != 0) && a.M == 16 && gsr_variant("GB_DIRECT", 1);
    // LDS: the SH tile and room for 4 gradient rows per Gaussian and chunk, or (DIRECT) the row chunk / the dL/dSH pieces alone
    // DIRECT: chunks of GB_RK gradient rows per thread (500k Gaussians: 1 -> 82 us, 2 or 4 -> 86 us)
    const int rk = gsr_variant("GB_RK", 1) == 2 ? 2 : 1;
    size_t lds = direct ? (size_t)rk * bs * 12 * 4 : (size_t)bs * (3 * a.M + 1) * 4;
    if (direct && lds < (size_t)bs * 17 * 4) lds = (size_t)bs * 17 * 4;  // the dL/dSH pieces
    if (!direct && lds < (size_t)4 * bs * 12 * 4) lds = (size_t)4 * bs * 12 * 4;
    p.lds_floats = (int)(lds / 4);
    lds += (size_t)gsr_variant("LDS_PAD", 0);
    // balanced split: gsr_part_cap(P) workgroups cover every scene (the ones beyond the split's end leave at once); a second grid
    // dimension only when the split's workgroups may hold more than `bs` Gaussians (rows <= instances: an upper bound of the shift)
    // (a workgroup of the split holds at most 2^shift / GSR_PART_W Gaussians)
    const uint32_t cols = a.wg_first ? (uint32_t)((((1u << gsr_part_shift(a.P, a.rows_bound)) / GSR_PART_W) + bs - 1) / bs) : 1u;
    const dim3 grid(a.wg_first ? gsr_part_cap(a.P) : (uint32_t)((a.P + bs - 1) / bs), cols), block(bs);
#define GB_LAUNCH(RAW_, BS_, DIRECT_, RK_)                                                                                      \
    do {                                                                                                                        \
        if (a.wg_first) hipLaunchKernelGGL((gaussian_backward_kernel<RAW_, BS_, DIRECT_, RK_, true>), grid, block, lds, s, p, h_x, h_y);  \
        else hipLaunchKernelGGL((gaussian_backward_kernel<RAW_, BS_, DIRECT_, RK_, false>), grid, block, lds, s, p, h_x, h_y);  \
    } while (0)
    if (bs == 256) {
        if (a.raw && direct) GB_LAUNCH(true, 256, true, 1);
        else if (a.raw) GB_LAUNCH(true, 256, false, 1);
        else if (direct && rk == 2) GB_LAUNCH(false, 256, true, 2);
        else if (direct) GB_LAUNCH(false, 256, true, 1);
        else GB_LAUNCH(false, 256, false, 1);
    } else {
        if (a.raw && direct) GB_LAUNCH(true, 128, true, 1);
        else if (a.raw) GB_LAUNCH(true, 128, false, 1);
        else if (direct && rk == 2) GB_LAUNCH(false, 128, true, 2);
        else if (direct) GB_LAUNCH(false, 128, true, 1);
        else GB_LAUNCH(false, 128, false, 1);
    }
#undef GB_LAUNCH
}
