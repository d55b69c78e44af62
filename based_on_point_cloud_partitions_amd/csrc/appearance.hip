// Full-resolution tail of the appearance-decoupling network: scene/ptgs/appearance_network.py:54-60
// from the bilinear upsampling of `up4`'s output on, and `transformed_image = mapping_image * image` (:15), forward and
// backward.  As stock torch ops this tail is 2.6 ms of the 3.7 ms the network costs per training iteration at 1080p
// (tools/appearance_probe.py) -- three times the rasterizer step.
//
//   u  = bilinear(feat [16,h,w] -> [16,H,W], align_corners)         built per tile in LDS from the patch of feat under it;
//                                                                   also written out once: the weight gradient reads it
//   a  = relu(conv3x3(u, w2) + b2)          [16,H,W]               saved for backward
//   m  = sigmoid(conv3x3(a, w3) + b3)       [3,H,W]                mapping_image
//   t  = m * image                                                  transformed_image
//
// The 16 -> 16 convolutions (forward, data gradient, weight gradient) are implicit GEMMs on v_mfma_f32_16x16x4_f32: f32 in,
// f32 accumulate, exactly an fmaf chain, so the results carry fp32 rounding only (no TF32/bf16 step).  Operand roles are
// chosen so that the pixel index is the MFMA column wherever a result is stored (coalesced CHW rows) and the reduction
// index wherever weights are accumulated.  LDS plane strides are chosen per access pattern (see `plane_base`).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AC = 16;  // feature channels: conv2 in/out, conv3 in (appearance_network.py:47-48)
constexpr int AO = 3;   // mapping channels
constexpr int TW = 64;  // tile width of every kernel here
constexpr int RS = TW + 2;  // LDS row stride of a tile with a one-pixel halo

// Tiles with a halo, read as MFMA B operands "4 channel planes x 16 consecutive pixels" (ds_read_b32 serves lanes 0-31 and
// 32-63 separately, bank = dword address mod 32): planes 16 banks apart (PS8 below; PS_BIG for the 16-row tile of conv3 forward).
constexpr int TH_BIG = 16;
constexpr int PS_BIG = 1200;  // >= (TH_BIG + 2) * RS = 1188, = 16 mod 32
// Tiles read as "16 channel planes x 2 consecutive pixels" per half-wave (the reduction index is the pixel): plane i starts
// at bank 16 (i & 1) + 2 (i >> 1), which also keeps planes 4j+k, k = 0..1 / 2..3, 16 banks apart for the pattern above.
__device__ __forceinline__ int plane_base(int i, int plane_stride) { return i * plane_stride + 16 * (i & 1) + 2 * (i >> 1); }
constexpr int TH_SMALL = 8;
constexpr int PS_CENTER = 544;  // >= 8 * 64 + 30, = 0 mod 32
constexpr int PS_HALO = 704;    // >= 10 * 66 + 30, = 0 mod 32
constexpr int PS_DZ3 = (TH_SMALL + 2) * RS;

struct Lerp {
    int i0, i1;
    float l0, l1;
};
// at::native upsample_bilinear2d, align_corners = True: src = scale * dst, scale = (in - 1) / (out - 1) in float
__device__ __forceinline__ Lerp lerp_coord(int dst, int n_in, float scale)
{
    Lerp r;
    const float src = scale * (float)dst;
    r.i0 = (int)src;
    r.i1 = r.i0 + (r.i0 < n_in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.0f - r.l1;
    return r;
}

// u tile (halo 1) of `rows` x 64 centre pixels at (ty0, tx0) into LDS planes at plane_off(c); zero outside the image
template <typename PlaneOff>
__device__ __forceinline__ void fill_upsampled_tile(float* lds, PlaneOff plane_off, int rows, int ty0, int tx0, int H, int W, int Hh, int Wh,
                                                    float sy, float sx, const float* __restrict__ feat, float* __restrict__ u_out, int tid)
{
    const int n = (rows + 2) * RS;
    const size_t plane = (size_t)Hh * Wh, oplane = (size_t)H * W;
    for (int pos = tid; pos < n; pos += 256) {
        const int ry = pos / RS, rx = pos - ry * RS;
        const int y = ty0 - 1 + ry, x = tx0 - 1 + rx;
        float* dst = lds + ry * RS + rx;
        if (y < 0 || y >= H || x < 0 || x >= W) {
#pragma unroll
            for (int c = 0; c < AC; c++) dst[plane_off(c)] = 0.0f;
            continue;
        }
        const Lerp ly = lerp_coord(y, Hh, sy), lx = lerp_coord(x, Wh, sx);
        const float* p = feat + (size_t)ly.i0 * Wh + lx.i0;
        const int o01 = lx.i1 - lx.i0, o10 = (ly.i1 - ly.i0) * Wh;
        const bool center = u_out && ry >= 1 && ry <= rows && rx >= 1 && rx <= TW;
#pragma unroll
        for (int c = 0; c < AC; c++) {
            const float top = lx.l0 * p[0] + lx.l1 * p[o01];
            const float bot = lx.l0 * p[o10] + lx.l1 * p[o10 + o01];
            const float u = ly.l0 * top + ly.l1 * bot;
            dst[plane_off(c)] = u;
            if (center) u_out[c * oplane + (size_t)y * W + x] = u;
            p += plane;
        }
    }
}

// PixelShuffle(2) of the block that produces feat (appearance_network.py:20-29): ps[c][Y][X] = x[4c + 2 (Y & 1) + (X & 1)][Y/2][X/2]
constexpr int UC = 8;  // channels after the shuffle (up4: 32 -> 8 -> conv -> 16)
__device__ __forceinline__ size_t ps_index(int c, int Y, int X, int Wh, size_t hplane)
{
    return (size_t)(4 * c + 2 * (Y & 1) + (X & 1)) * hplane + (size_t)(Y >> 1) * Wh + (X >> 1);
}

// A tile in flight: the 16 channel values of NPOS positions per thread, loaded from a CHW tensor while the previous tile is
// being computed on (tile_fetch), stored to LDS once that is done (tile_commit).  Zero outside the image.
template <int WIDTH, int NPOS>
__device__ __forceinline__ void tile_fetch(float (&v)[NPOS][AC], int rows, int y_first, int x_first, int H, int W,
                                           const float* __restrict__ src, int tid)
{
    const size_t plane = (size_t)H * W;
#pragma unroll
    for (int h = 0; h < NPOS; h++) {
        const int pos = tid + 256 * h;
        const int ry = pos / WIDTH, rx = pos - ry * WIDTH;
        const int y = y_first + ry, x = x_first + rx;
        const bool in = pos < rows * WIDTH && y >= 0 && y < H && x >= 0 && x < W;
        const float* p = src + (size_t)(in ? y : 0) * W + (in ? x : 0);
#pragma unroll
        for (int c = 0; c < AC; c++) v[h][c] = in ? p[c * plane] : 0.0f;
    }
}

template <int WIDTH, int NPOS, typename PlaneOff>
__device__ __forceinline__ void tile_commit(float* lds, PlaneOff plane_off, const float (&v)[NPOS][AC], int rows, int row_stride, int tid)
{
#pragma unroll
    for (int h = 0; h < NPOS; h++) {
        const int pos = tid + 256 * h;
        const int ry = pos / WIDTH, rx = pos - ry * WIDTH;
        if (pos < rows * WIDTH) {
#pragma unroll
            for (int c = 0; c < AC; c++) lds[plane_off(c) + ry * row_stride + rx] = v[h][c];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 16 -> 16 convolution as 36 MFMAs per 16 pixels: D[out channel][pixel] += W[out][in group of 4] * X[in group of 4][pixel].
//   DATA_GRAD = false: a = relu(conv(u, w2) + b2), u rebuilt from feat and also written out for the backward's weight
//                      gradient                                                     (appearance_network.py:54-57)
//   DATA_GRAD = true : du = conv^T(dz2, w2)                                        (its autograd)
// Persistent workgroups over 8 x 64 tiles.  While a tile's 288 MFMAs per wave run, the next tile's input is already on its
// way into registers: for the data gradient the halo tile itself, for the forward the patch of `feat` under the tile (8 x 36
// source pixels when the scale is about 1/2: `staged`), which is then parked in LDS and interpolated from there.
// ------------------------------------------------------------------------------------------------------------------
constexpr int PS8 = 688;  // >= 10 * 66, = 16 mod 32: planes 16 banks apart for "4 channel planes x 16 pixels" operand reads
constexpr int ST_ROWS = 8, ST_COLS = 36, ST_POS = ST_ROWS * ST_COLS, ST_PER_THREAD = AC * ST_POS / 256;
inline bool upsample_staged_ok(float sy, float sx) { return sy <= 0.55f && sx <= 0.51f; }  // 9 sy + 2 <= 8, 65 sx + 2 <= 36

template <int MODE>
__global__ __launch_bounds__(256) void app_conv16_kernel(int H, int W, int Hh, int Wh, float sy, float sx, int tiles_x, int n_tiles,
                                                         int staged, const float* __restrict__ src, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, float* __restrict__ out,
                                                         float* __restrict__ u_out)
{
    constexpr bool DATA_GRAD = MODE != 0, UP = MODE == 2;
    __shared__ float lds[AC * PS8];
    __shared__ float stage[DATA_GRAD ? 1 : AC * ST_POS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, k = lane >> 4;
    auto plane_off = [](int c) { return c * PS8; };
    const size_t plane = (size_t)H * W, hplane = (size_t)Hh * Wh;
    // A operands: lane (row n, k) of step (tap t, group j) holds W[n][4j + k] for that tap
    float wreg[36];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            wreg[t * 4 + j] = UP          ? (n < UC ? w2[((4 * j + k) * UC + n) * 9 + t] : 0.0f)  // dps[ci = n < 8] += w4[co][ci][t] * dz[co]
                              : DATA_GRAD ? w2[((4 * j + k) * AC + n) * 9 + t]   // du[ci = n] += w2[co = 4j+k][ci][t] * dz2[co]
                                          : w2[(n * AC + 4 * j + k) * 9 + t];   // a[co = n]  += w2[co][ci = 4j+k][t] * u[ci]
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; r++) bias[r] = DATA_GRAD ? 0.0f : b2[4 * k + r];

    float pre[DATA_GRAD ? 3 : 1][AC];        // data gradient: the next halo tile (660 positions)
    float pre_f[DATA_GRAD ? 1 : ST_PER_THREAD];  // forward: the next patch of feat
    auto tile_origin = [&](int tile, int& ty0, int& tx0) { ty0 = (tile / tiles_x) * TH_SMALL; tx0 = (tile % tiles_x) * TW; };
    auto patch_origin = [&](int ty0, int tx0, int& ry0, int& rx0) {
        ry0 = lerp_coord(max(ty0 - 1, 0), Hh, sy).i0;
        rx0 = lerp_coord(max(tx0 - 1, 0), Wh, sx).i0;
    };
    auto fetch = [&](int tile) {
        int ty0, tx0;
        tile_origin(tile, ty0, tx0);
        if constexpr (DATA_GRAD) {
            tile_fetch<RS, 3>(pre, TH_SMALL + 2, ty0 - 1, tx0 - 1, H, W, src, tid);
        } else if (staged) {
            int ry0, rx0;
            patch_origin(ty0, tx0, ry0, rx0);
#pragma unroll
            for (int i = 0; i < ST_PER_THREAD; i++) {
                const int idx = tid + 256 * i, c = idx / ST_POS, p = idx - c * ST_POS, ry = p / ST_COLS, rx = p - ry * ST_COLS;
                pre_f[i] = src[c * hplane + (size_t)min(ry0 + ry, Hh - 1) * Wh + min(rx0 + rx, Wh - 1)];
            }
        }
    };
    auto commit = [&](int tile) {
        int ty0, tx0;
        tile_origin(tile, ty0, tx0);
        if constexpr (DATA_GRAD) {
            tile_commit<RS, 3>(lds, plane_off, pre, TH_SMALL + 2, RS, tid);
        } else if (!staged) {
            fill_upsampled_tile(lds, plane_off, TH_SMALL, ty0, tx0, H, W, Hh, Wh, sy, sx, src, u_out, tid);
        } else {
            int ry0, rx0;
            patch_origin(ty0, tx0, ry0, rx0);
#pragma unroll
            for (int i = 0; i < ST_PER_THREAD; i++) stage[tid + 256 * i] = pre_f[i];
            __syncthreads();
            for (int pos = tid; pos < (TH_SMALL + 2) * RS; pos += 256) {
                const int ry = pos / RS, rx = pos - ry * RS;
                const int y = ty0 - 1 + ry, x = tx0 - 1 + rx;
                float* dst = lds + pos;
                if (y < 0 || y >= H || x < 0 || x >= W) {
#pragma unroll
                    for (int c = 0; c < AC; c++) dst[c * PS8] = 0.0f;
                    continue;
                }
                const Lerp ly = lerp_coord(y, Hh, sy), lx = lerp_coord(x, Wh, sx);
                const float* p = stage + (ly.i0 - ry0) * ST_COLS + (lx.i0 - rx0);
                const int o01 = lx.i1 - lx.i0, o10 = (ly.i1 - ly.i0) * ST_COLS;
                const bool center = ry >= 1 && ry <= TH_SMALL && rx >= 1 && rx <= TW;
#pragma unroll
                for (int c = 0; c < AC; c++) {
                    const float top = lx.l0 * p[0] + lx.l1 * p[o01];
                    const float bot = lx.l0 * p[o10] + lx.l1 * p[o10 + o01];
                    const float u = ly.l0 * top + ly.l1 * bot;
                    dst[c * PS8] = u;
                    if (center) u_out[c * plane + (size_t)y * W + x] = u;
                    p += ST_POS;
                }
            }
        }
    };
    // one 16-pixel segment of finished accumulators -> bias, ReLU (forward), CHW store
    auto store_segment = [&](const f32x4& acc, int y, int x) {
        if (y < H && x < W) {
            if constexpr (UP) {  // rows 0..7 are the shuffled channels: undo the pixel shuffle (x is [32][H/2][W/2])
                if (k < 2) {
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) out[ps_index(4 * k + reg, y, x, Wh, hplane)] = acc[reg];
                }
            } else {
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    float v = acc[reg] + bias[reg];
                    if (!DATA_GRAD) v = fmaxf(v, 0.0f);
                    out[(4 * k + reg) * plane + (size_t)y * W + x] = v;
                }
            }
        }
    };

    const float* bbase = lds + k * PS8 + n;
    // B operands of one tap (4 channel groups x 4 segments of 16 pixels) are read one tap ahead of the MFMAs using them
    auto load_tap = [&](float(&b)[16], int row, int t) {
        const int dy = t / 3, dx = t % 3;
        // forward reads x[q + (dy-1, dx-1)]; the data gradient reads dz[q - (dy-1, dx-1)]; tile origin is (-1, -1)
        const int oy = DATA_GRAD ? 2 - dy : dy, ox = DATA_GRAD ? 2 - dx : dx;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int s = 0; s < 4; s++) b[j * 4 + s] = bbase[4 * j * PS8 + (row + oy) * RS + s * 16 + ox];
    };

    if ((int)blockIdx.x < n_tiles) {
        fetch(blockIdx.x);
        commit(blockIdx.x);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int ty0, tx0;
        tile_origin(tile, ty0, tx0);
        const int next = tile + gridDim.x;
        if (next < n_tiles) fetch(next);
        float bq[2][16];
        f32x4 acc[2][4];
        load_tap(bq[0], wave * 2, 0);
#pragma unroll
        for (int r = 0; r < 2; r++) {
#pragma unroll
            for (int s = 0; s < 4; s++) acc[r][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int step = r * 9 + t;
                if (step + 1 < 18) load_tap(bq[(step + 1) & 1], wave * 2 + (step + 1) / 9, (step + 1) % 9);
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int s = 0; s < 4; s++)
                        acc[r][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[t * 4 + j], bq[step & 1][j * 4 + s], acc[r][s], 0, 0, 0);
                // the first row's results leave while the second row's MFMAs run
                if (r == 1 && t >= 1 && t <= 4) store_segment(acc[0][t - 1], ty0 + wave * 2, tx0 + (t - 1) * 16 + n);
                // keep the schedule as written: this tap's MFMAs interleaved with the next tap's LDS reads
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; s++) store_segment(acc[1][s], ty0 + wave * 2 + 1, tx0 + s * 16 + n);
        __syncthreads();  // every wave is done reading this tile
        if (next < n_tiles) commit(next);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// m = sigmoid(conv3x3(a, w3) + b3), t = m * image   (appearance_network.py:58-60, :15).  3 output channels: VALU, 4 pixels
// per thread, weights broadcast from LDS.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void app_conv3_fwd_kernel(int H, int W, int tiles_x, int n_tiles, const float* __restrict__ a,
                                                            const float* __restrict__ w3, const float* __restrict__ b3,
                                                            const float* __restrict__ image, float* __restrict__ mapping,
                                                            float* __restrict__ transformed)
{
    __shared__ float lds[AC * PS_BIG];
    __shared__ float4 wl[AC * 9];  // [ci][dy][dx] -> the three output channels' weights
    const int tid = threadIdx.x;
    auto plane_off = [](int c) { return c * PS_BIG; };
    if (tid < AC * 9) wl[tid] = float4{w3[tid], w3[AC * 9 + tid], w3[2 * AC * 9 + tid], 0.0f};
    const float bias[AO] = {b3[0], b3[1], b3[2]};
    const int row = tid >> 4, xs = (tid & 15) * 4;
    const size_t plane = (size_t)H * W;
    // persistent over 16 x 64 tiles; the next tile (18 x 66 with its halo) travels in registers during the FMAs of this one
    float pre[5][AC];
    auto fetch = [&](int tile) {
        tile_fetch<RS, 5>(pre, TH_BIG + 2, (tile / tiles_x) * TH_BIG - 1, (tile % tiles_x) * TW - 1, H, W, a, tid);
    };
    if ((int)blockIdx.x < n_tiles) {
        fetch(blockIdx.x);
        tile_commit<RS, 5>(lds, plane_off, pre, TH_BIG + 2, RS, tid);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int ty0 = (tile / tiles_x) * TH_BIG, tx0 = (tile % tiles_x) * TW;
        const int next = tile + gridDim.x;
        if (next < n_tiles) fetch(next);
        // this thread's 4 x 3 image values: requested now, used after the FMAs
        const int y = ty0 + row;
        float img[4][AO];
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int co = 0; co < AO; co++) {
                const bool in = y < H && tx0 + xs + p < W;
                img[p][co] = in ? image[co * plane + (size_t)y * W + tx0 + xs + p] : 0.0f;
            }
        float acc[4][AO];
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int co = 0; co < AO; co++) acc[p][co] = 0.0f;
#pragma unroll 2
        for (int c = 0; c < AC; c++) {
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                const float2* p2 = reinterpret_cast<const float2*>(lds + c * PS_BIG + (row + dy) * RS + xs);
                const float2 v01 = p2[0], v23 = p2[1], v45 = p2[2];
                const float v[6] = {v01.x, v01.y, v23.x, v23.y, v45.x, v45.y};
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {
                    const float4 w = wl[(c * 3 + dy) * 3 + dx];
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        acc[p][0] = __builtin_fmaf(w.x, v[p + dx], acc[p][0]);
                        acc[p][1] = __builtin_fmaf(w.y, v[p + dx], acc[p][1]);
                        acc[p][2] = __builtin_fmaf(w.z, v[p + dx], acc[p][2]);
                    }
                }
            }
        }
#pragma unroll
        for (int co = 0; co < AO; co++) {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int x = tx0 + xs + p;
                const float m = gsr_sigmoidf(acc[p][co] + bias[co]);
                if (y < H && x < W) {
                    const size_t o = co * plane + (size_t)y * W + x;
                    mapping[o] = m;
                    transformed[o] = m * img[p][co];
                }
            }
        }
        __syncthreads();  // every wave is done reading this tile
        if (next < n_tiles) tile_commit<RS, 5>(lds, plane_off, pre, TH_BIG + 2, RS, tid);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward through t = m * image, the sigmoid and conv3, per 8 x 64 tile:
//   dimage = g_t * m ; dm = g_t * image (+ g_m) ; dz3 = dm * (1 - m) * m                       (elementwise, while staging)
//   dz2[ci] = [a > 0] * sum_{co,tap} w3[co][ci][tap] dz3[co][q - tap]     7 MFMAs / 16 px   (K = 27 padded to 28)
//   dW3[co][ci][tap] += sum_q a[ci][q] dz3[co][q - tap]                   8 MFMAs / 16 px   (N = 27 padded to 32)
//   db3 += sum dz3 ; db2 += sum dz2
// Each workgroup walks tiles blockIdx.x, + gridDim.x, ... and leaves one row of partial sums (452 floats).
// ------------------------------------------------------------------------------------------------------------------
constexpr int B2_PARTIAL = 27 * AC + AO + AC + 1;  // dW3 [27 cols][16 ci] | db3 | db2 | pad  = 452

__global__ __launch_bounds__(256) void app_conv3_bwd_kernel(int H, int W, int tiles_x, int n_tiles, const float* __restrict__ a,
                                                            const float* __restrict__ w3, const float* __restrict__ image,
                                                            const float* __restrict__ mapping, const float* __restrict__ g_t,
                                                            const float* __restrict__ g_m, float* __restrict__ dimage,
                                                            float* __restrict__ dz2, float* __restrict__ partials)
{
    __shared__ float a_lds[AC * PS_CENTER];
    __shared__ float z_lds[AO * PS_DZ3];
    __shared__ float red[4 * 512 + 4 * 256 + 4 * AO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, k = lane >> 4;
    const size_t plane = (size_t)H * W;
    // da: A[ci = n][kk = 4s + k] = w3[co][ci][tap], kk = 3 tap + co (28th column zero); B[kk][px] = dz3[co][q - tap]
    float wa[7];
    int offB[7];
#pragma unroll
    for (int s = 0; s < 7; s++) {
        const int kk = 4 * s + k, kc = kk < 27 ? kk : 0;
        const int tap = kc / 3, co = kc % 3, dy = tap / 3, dx = tap % 3;
        wa[s] = kk < 27 ? w3[(co * AC + n) * 9 + tap] : 0.0f;
        offB[s] = co * PS_DZ3 + (2 - dy) * RS + (2 - dx) + n;
    }
    // dW3: A[ci = n][q = k] = a[ci][q]; B[q = k][col = 16 nb + n] = dz3[co][q - tap], col = 3 tap + co (cols 27..31 unused)
    int offW[2];
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
        const int col = nb * 16 + n, cc = col < 27 ? col : 26;
        const int tap = cc / 3, co = cc % 3, dy = tap / 3, dx = tap % 3;
        offW[nb] = co * PS_DZ3 + (2 - dy) * RS + (2 - dx) + k;
    }
    const int a_base_n = plane_base(n, PS_CENTER);
    int a_base_d[4];
#pragma unroll
    for (int reg = 0; reg < 4; reg++) a_base_d[reg] = plane_base(4 * k + reg, PS_CENTER);
    f32x4 accW[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float bsum2[4] = {0.f, 0.f, 0.f, 0.f};
    float bsum3[AO] = {0.f, 0.f, 0.f};

    // the next tile travels in registers while this one is computed: relu(conv2) for the centre pixels, and the three
    // per-pixel inputs of dz3 (upstream gradient(s), image, mapping) for the centre and its halo
    auto a_off = [](int c) { return plane_base(c, PS_CENTER); };
    float pre_a[2][AC];
    float pre_g[3][AO], pre_i[3][AO], pre_m[3][AO], pre_gm[3][AO];
    auto fetch = [&](int tile) {
        const int ty0 = (tile / tiles_x) * TH_SMALL, tx0 = (tile % tiles_x) * TW;
        tile_fetch<TW, 2>(pre_a, TH_SMALL, ty0, tx0, H, W, a, tid);
#pragma unroll
        for (int h = 0; h < 3; h++) {
            const int pos = tid + 256 * h, ry = pos / RS, rx = pos - ry * RS;
            const int y = ty0 - 1 + ry, x = tx0 - 1 + rx;
            const bool in = pos < PS_DZ3 && y >= 0 && y < H && x >= 0 && x < W;
            const size_t o = in ? (size_t)y * W + x : 0;
#pragma unroll
            for (int co = 0; co < AO; co++) {
                pre_g[h][co] = in ? g_t[co * plane + o] : 0.0f;
                pre_i[h][co] = in ? image[co * plane + o] : 0.0f;
                pre_m[h][co] = in ? mapping[co * plane + o] : 0.0f;
                pre_gm[h][co] = (in && g_m) ? g_m[co * plane + o] : 0.0f;
            }
        }
    };
    auto commit = [&](int tile) {
        const int ty0 = (tile / tiles_x) * TH_SMALL, tx0 = (tile % tiles_x) * TW;
        tile_commit<TW, 2>(a_lds, a_off, pre_a, TH_SMALL, TW, tid);
#pragma unroll
        for (int h = 0; h < 3; h++) {
            const int pos = tid + 256 * h, ry = pos / RS, rx = pos - ry * RS;
            const int y = ty0 - 1 + ry, x = tx0 - 1 + rx;
            if (pos >= PS_DZ3) continue;
            const bool in = y >= 0 && y < H && x >= 0 && x < W;
            const bool center = in && ry >= 1 && ry <= TH_SMALL && rx >= 1 && rx <= TW;
#pragma unroll
            for (int co = 0; co < AO; co++) {
                const float g = pre_g[h][co], m = pre_m[h][co];
                const float dm = g * pre_i[h][co] + pre_gm[h][co];  // outside the image every term is 0
                const float dz = (dm * (1.0f - m)) * m;
                if (center) {
                    dimage[co * plane + (size_t)y * W + x] = g * m;
                    bsum3[co] += dz;
                }
                z_lds[co * PS_DZ3 + pos] = dz;
            }
        }
    };
    if ((int)blockIdx.x < n_tiles) {
        fetch(blockIdx.x);
        commit(blockIdx.x);
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int ty0 = (tile / tiles_x) * TH_SMALL, tx0 = (tile % tiles_x) * TW;
        const int next = tile + gridDim.x;
        if (next < n_tiles) fetch(next);
        // operands of one 16-pixel segment (19 LDS reads + 4 mask reads) are fetched one segment ahead of its 15 MFMAs
        auto load_seg = [&](float(&zd)[7], float(&av)[4], float(&zw)[8], float(&am)[4], int seg) {
            const int r = wave * 2 + (seg >> 2), x0 = (seg & 3) * 16;
            const float* zb = z_lds + r * RS + x0;
#pragma unroll
            for (int s = 0; s < 7; s++) zd[s] = zb[offB[s]];
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                av[ks] = a_lds[a_base_n + r * TW + x0 + 4 * ks + k];
#pragma unroll
                for (int nb = 0; nb < 2; nb++) zw[ks * 2 + nb] = zb[offW[nb] + 4 * ks];
            }
#pragma unroll
            for (int reg = 0; reg < 4; reg++) am[reg] = a_lds[a_base_d[reg] + r * TW + x0 + n];
        };
        float zd[2][7], av[2][4], zw[2][8], am[2][4];
        load_seg(zd[0], av[0], zw[0], am[0], 0);
#pragma unroll
        for (int seg = 0; seg < 8; seg++) {
            const int r = wave * 2 + (seg >> 2), x0 = (seg & 3) * 16, cur = seg & 1;
            if (seg + 1 < 8) load_seg(zd[cur ^ 1], av[cur ^ 1], zw[cur ^ 1], am[cur ^ 1], seg + 1);
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            // the 7 MFMAs of the dz2 chain depend on each other: alternate them with the 8 independent dW3 ones
#pragma unroll
            for (int s = 0; s < 7; s++) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], zd[cur][s], acc, 0, 0, 0);
                accW[s & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][s >> 1], zw[cur][s], accW[s & 1], 0, 0, 0);
            }
            accW[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][3], zw[cur][7], accW[1], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 15; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int y = ty0 + r, x = tx0 + x0 + n;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const float v = am[cur][reg] > 0.0f ? acc[reg] : 0.0f;
                bsum2[reg] += v;
                if (y < H && x < W) dz2[(4 * k + reg) * plane + (size_t)y * W + x] = v;
            }
        }
        __syncthreads();  // every wave is done reading this tile
        if (next < n_tiles) commit(next);
        __syncthreads();
    }
    // workgroup totals: dW3 D layout = row ci = 4k + reg, column col = 16 nb + n
    float* redW = red;             // [wave][nb][reg][lane]
    float* red2 = red + 4 * 512;   // [wave][reg][lane]
    float* red3 = red2 + 4 * 256;  // [wave][co]
#pragma unroll
    for (int nb = 0; nb < 2; nb++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) redW[wave * 512 + (nb * 4 + reg) * 64 + lane] = accW[nb][reg];
#pragma unroll
    for (int reg = 0; reg < 4; reg++) red2[wave * 256 + reg * 64 + lane] = bsum2[reg];
#pragma unroll
    for (int co = 0; co < AO; co++) {
        float v = bsum3[co];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red3[wave * AO + co] = v;
    }
    __syncthreads();
    float* out = partials + (size_t)blockIdx.x * B2_PARTIAL;
    for (int e = tid; e < 512; e += 256) {
        const float v = ((redW[e] + redW[512 + e]) + redW[1024 + e]) + redW[1536 + e];
        const int nb = e >> 8, reg = (e >> 6) & 3, l = e & 63, col = nb * 16 + (l & 15), ci = 4 * (l >> 4) + reg;
        if (col < 27) out[col * AC + ci] = v;
    }
    if (tid < AO) out[27 * AC + tid] = ((red3[tid] + red3[AO + tid]) + red3[2 * AO + tid]) + red3[3 * AO + tid];
    if (tid < AC) {  // db2[ci = 4k + reg]: 16 columns x 4 waves
        const int kk = tid >> 2, reg = tid & 3;
        float v = 0.0f;
        for (int w = 0; w < 4; w++)
            for (int c = 0; c < 16; c++) v += red2[w * 256 + reg * 64 + kk * 16 + c];
        out[27 * AC + AO + tid] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// dW2[co][ci][tap] += sum_q dz2[co][q] u[ci][q + tap]: A[co = n][q = k], B[q = k][ci = n] per tap, 9 MFMAs per 4 pixels,
// u as the forward saved it.  One row of 2304 partial sums per workgroup (layout [tap][co][ci]).
// ------------------------------------------------------------------------------------------------------------------
constexpr int B3_PARTIAL = 9 * AC * AC;
constexpr int UP_PARTIAL = B3_PARTIAL + AC;  // + the bias gradient

template <bool UP>  // UP: the same for up4's conv (8 shuffled input channels read through ps_index; + the bias gradient)
__global__ __launch_bounds__(256) void app_conv2_bwd_weight_kernel(int H, int W, int tiles_x, int n_tiles, const float* __restrict__ u,
                                                                   const float* __restrict__ dz2, float* __restrict__ partials)
{
    __shared__ float d_lds[AC * PS_CENTER];
    __shared__ float u_lds[AC * PS_HALO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, k = lane >> 4;
    const int d_base = plane_base(n, PS_CENTER) + k, u_base = plane_base(UP ? (n & 7) : n, PS_HALO) + k;  // UP: columns 8..15 unused
    auto d_off = [](int c) { return plane_base(c, PS_CENTER); };
    auto u_off = [](int c) { return plane_base(c, PS_HALO); };
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the next tile (dz2 centre, u with halo) travels in registers while this one's 288 MFMAs per wave run
    float pre_d[2][AC], pre_u[3][UP ? UC : AC];
    float bsum = 0.0f;  // UP: sum of this lane's A operands = its share of the bias gradient of channel n
    auto fetch = [&](int tile) {
        const int ty0 = (tile / tiles_x) * TH_SMALL, tx0 = (tile % tiles_x) * TW;
        tile_fetch<TW, 2>(pre_d, TH_SMALL, ty0, tx0, H, W, dz2, tid);
        if constexpr (UP) {
            const size_t hplane = (size_t)(H / 2) * (W / 2);
#pragma unroll
            for (int h = 0; h < 3; h++) {
                const int pos = tid + 256 * h, ry = pos / RS, rx = pos - ry * RS;
                const int y = ty0 - 1 + ry, x = tx0 - 1 + rx;
                const bool in = pos < (TH_SMALL + 2) * RS && y >= 0 && y < H && x >= 0 && x < W;
#pragma unroll
                for (int c = 0; c < UC; c++) pre_u[h][c] = in ? u[ps_index(c, y, x, W / 2, hplane)] : 0.0f;
            }
        } else {
            tile_fetch<RS, 3>(pre_u, TH_SMALL + 2, ty0 - 1, tx0 - 1, H, W, u, tid);
        }
    };
    auto commit = [&]() {
        tile_commit<TW, 2>(d_lds, d_off, pre_d, TH_SMALL, TW, tid);
        if constexpr (UP) {
#pragma unroll
            for (int h = 0; h < 3; h++) {
                const int pos = tid + 256 * h;
                if (pos < (TH_SMALL + 2) * RS) {
#pragma unroll
                    for (int c = 0; c < UC; c++) u_lds[u_off(c) + pos] = pre_u[h][c];
                }
            }
        } else {
            tile_commit<RS, 3>(u_lds, u_off, pre_u, TH_SMALL + 2, RS, tid);
        }
    };
    if ((int)blockIdx.x < n_tiles) {
        fetch(blockIdx.x);
        commit();
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if (next < n_tiles) fetch(next);
        // the 10 operands of one 4-pixel step are fetched one step ahead of its 9 MFMAs
        auto load_step = [&](float& a_op, float(&b_op)[9], int step) {
            const int r = wave * 2 + (step >> 4), x0 = (step & 15) * 4;
            a_op = d_lds[d_base + r * TW + x0];
#pragma unroll
            for (int t = 0; t < 9; t++) b_op[t] = u_lds[u_base + (r + t / 3) * RS + x0 + t % 3];
        };
        float aq[2], bq[2][9];
        load_step(aq[0], bq[0], 0);
#pragma unroll
        for (int step = 0; step < 32; step++) {
            const int cur = step & 1;
            if (step + 1 < 32) load_step(aq[cur ^ 1], bq[cur ^ 1], step + 1);
#pragma unroll
            for (int t = 0; t < 9; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[cur], bq[cur][t], acc[t], 0, 0, 0);
            if (UP) bsum += aq[cur];
#pragma unroll
            for (int i = 0; i < 10; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // every wave is done reading this tile
        if (next < n_tiles) commit();
        __syncthreads();
    }
    float* red = u_lds;  // [wave][t][reg][lane] = 4 x 2304 floats <= AC * PS_HALO
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) red[wave * B3_PARTIAL + (t * 4 + reg) * 64 + lane] = acc[t][reg];
    __syncthreads();
    float* out = partials + (size_t)blockIdx.x * (UP ? UP_PARTIAL : B3_PARTIAL);
    for (int e = tid; e < B3_PARTIAL; e += 256) {
        const float v = ((red[e] + red[B3_PARTIAL + e]) + red[2 * B3_PARTIAL + e]) + red[3 * B3_PARTIAL + e];
        const int t = e >> 8, reg = (e >> 6) & 3, l = e & 63, co = 4 * (l >> 4) + reg, ci = l & 15;
        out[(t * AC + co) * AC + ci] = v;
    }
    if constexpr (UP) {  // bias gradient: channel n, summed over the 4 pixel groups k and the 4 waves
        __syncthreads();
        red[tid] = bsum;
        __syncthreads();
        if (tid < AC) {
            float v = 0.0f;
            for (int w = 0; w < 4; w++)
                for (int kk = 0; kk < 4; kk++) v += red[w * 64 + kk * 16 + tid];
            out[B3_PARTIAL + tid] = v;
        }
    }
}

// up4 forward: feat = relu(conv3x3(ps(x), w) + b), 8 -> 16 channels: 18 MFMAs per 16 pixels, same pipeline as app_conv16_kernel
__global__ __launch_bounds__(256) void app_up_fwd_kernel(int H, int W, int tiles_x, int n_tiles, const float* __restrict__ x,
                                                         const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ feat)
{
    __shared__ float lds[UC * PS8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, k = lane >> 4;
    const size_t plane = (size_t)H * W, hplane = (size_t)(H / 2) * (W / 2);
    float wreg[18];  // A[co = n][ci = 4j + k] per tap
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int j = 0; j < 2; j++) wreg[t * 2 + j] = w[(n * UC + 4 * j + k) * 9 + t];
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; r++) bias[r] = b[4 * k + r];
    float pre[3][UC];
    auto fetch = [&](int tile) {
        const int ty0 = (tile / tiles_x) * TH_SMALL, tx0 = (tile % tiles_x) * TW;
#pragma unroll
        for (int h = 0; h < 3; h++) {
            const int pos = tid + 256 * h, ry = pos / RS, rx = pos - ry * RS;
            const int y = ty0 - 1 + ry, xx = tx0 - 1 + rx;
            const bool in = pos < (TH_SMALL + 2) * RS && y >= 0 && y < H && xx >= 0 && xx < W;
#pragma unroll
            for (int c = 0; c < UC; c++) pre[h][c] = in ? x[ps_index(c, y, xx, W / 2, hplane)] : 0.0f;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int h = 0; h < 3; h++) {
            const int pos = tid + 256 * h;
            if (pos < (TH_SMALL + 2) * RS) {
#pragma unroll
                for (int c = 0; c < UC; c++) lds[c * PS8 + pos] = pre[h][c];
            }
        }
    };
    auto store_segment = [&](const f32x4& acc, int y, int xx) {
        if (y < H && xx < W) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) feat[(4 * k + reg) * plane + (size_t)y * W + xx] = fmaxf(acc[reg] + bias[reg], 0.0f);
        }
    };
    const float* bbase = lds + k * PS8 + n;
    auto load_tap = [&](float(&bv)[8], int row, int t) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int s = 0; s < 4; s++) bv[j * 4 + s] = bbase[4 * j * PS8 + (row + t / 3) * RS + s * 16 + t % 3];
    };
    if ((int)blockIdx.x < n_tiles) {
        fetch(blockIdx.x);
        commit();
    }
    __syncthreads();
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int ty0 = (tile / tiles_x) * TH_SMALL, tx0 = (tile % tiles_x) * TW;
        const int next = tile + gridDim.x;
        if (next < n_tiles) fetch(next);
        float bq[2][8];
        f32x4 acc[2][4];
        load_tap(bq[0], wave * 2, 0);
#pragma unroll
        for (int r = 0; r < 2; r++) {
#pragma unroll
            for (int s = 0; s < 4; s++) acc[r][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int step = r * 9 + t;
                if (step + 1 < 18) load_tap(bq[(step + 1) & 1], wave * 2 + (step + 1) / 9, (step + 1) % 9);
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int s = 0; s < 4; s++)
                        acc[r][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[t * 2 + j], bq[step & 1][j * 4 + s], acc[r][s], 0, 0, 0);
                if (r == 1 && t >= 1 && t <= 4) store_segment(acc[0][t - 1], ty0 + wave * 2, tx0 + (t - 1) * 16 + n);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; s++) store_segment(acc[1][s], ty0 + wave * 2 + 1, tx0 + s * 16 + n);
        __syncthreads();
        if (next < n_tiles) commit();
        __syncthreads();
    }
}

// one wave per element of dW4 [16][8][3][3] and db4 [16]: sums the weight-gradient kernel's partial rows
__global__ __launch_bounds__(256) void app_up_reduce_kernel(const float* __restrict__ partials, int rows, float* __restrict__ dw,
                                                            float* __restrict__ db)
{
    const int lane = threadIdx.x & 63, e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= AC * UC * 9 + AC) return;
    int src;
    if (e < AC * UC * 9) {  // dw[co][ci][t] <- partial [t][co][ci16]
        const int co = e / (UC * 9), ci = (e / 9) % UC, t = e % 9;
        src = (t * AC + co) * AC + ci;
    } else {
        src = B3_PARTIAL + (e - AC * UC * 9);
    }
    float v = 0.0f;
    for (int r = lane; r < rows; r += 64) v += partials[(size_t)r * UP_PARTIAL + src];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) (e < AC * UC * 9 ? dw[e] : db[e - AC * UC * 9]) = v;
}

// One wave per output element: sums `rows` partial rows (row stride `stride`), lanes striding over the rows.
//   job 0: dW3/db3/db2 from conv3_bwd partials;  job 1: dW2 from conv2_bwd_weight partials
__global__ __launch_bounds__(256) void app_reduce_partials_kernel(const float* __restrict__ p3, int rows3, const float* __restrict__ p2,
                                                                  int rows2, float* __restrict__ dw3, float* __restrict__ db3,
                                                                  float* __restrict__ db2, float* __restrict__ dw2)
{
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n3 = 27 * AC + AO + AC;
    if (e >= n3 + B3_PARTIAL) return;
    const bool first = e < n3;
    const float* src = first ? p3 + e : p2 + (e - n3);
    const int rows = first ? rows3 : rows2, stride = first ? B2_PARTIAL : B3_PARTIAL;
    float v = 0.0f;
    for (int r = lane; r < rows; r += 64) v += src[(size_t)r * stride];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane != 0) return;
    if (first) {
        if (e < 27 * AC) {  // partial layout [col = 3 tap + co][ci] -> w3 [co][ci][tap]
            const int col = e / AC, ci = e % AC;
            dw3[((col % 3) * AC + ci) * 9 + col / 3] = v;
        } else if (e < 27 * AC + AO)
            db3[e - 27 * AC] = v;
        else
            db2[e - 27 * AC - AO] = v;
    } else {  // partial layout [tap][co][ci] -> w2 [co][ci][tap]
        const int j = e - n3, t = j >> 8, co = (j >> 4) & 15, ci = j & 15;
        dw2[(co * AC + ci) * 9 + t] = v;
    }
}

// dfeat[c][hy][hx] = sum over the output pixels whose bilinear footprint contains (hy, hx): the adjoint of the upsampling,
// as a gather (deterministic).  One thread per (hy, hx), all 16 channels.
__global__ __launch_bounds__(256) void app_upsample_bwd_kernel(int H, int W, int Hh, int Wh, float sy, float sx,
                                                               const float* __restrict__ du, const float* __restrict__ relu_of,
                                                               float* __restrict__ dfeat)
{
    const int hx = blockIdx.x * 64 + (threadIdx.x & 63), hy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (hx >= Wh || hy >= Hh) return;
    // candidate destination range: src = scale * dst in (h - 1, h + 1)
    const int ylo = sy > 0.0f ? max(0, (int)floorf((float)(hy - 1) / sy) - 1) : 0;
    const int yhi = sy > 0.0f ? min(H - 1, (int)ceilf((float)(hy + 1) / sy) + 1) : H - 1;
    const int xlo = sx > 0.0f ? max(0, (int)floorf((float)(hx - 1) / sx) - 1) : 0;
    const int xhi = sx > 0.0f ? min(W - 1, (int)ceilf((float)(hx + 1) / sx) + 1) : W - 1;
    float acc[AC];
#pragma unroll
    for (int c = 0; c < AC; c++) acc[c] = 0.0f;
    const size_t plane = (size_t)H * W;
    for (int y = ylo; y <= yhi; y++) {
        const Lerp ly = lerp_coord(y, Hh, sy);
        const float wy = (ly.i0 == hy ? ly.l0 : 0.0f) + (ly.i1 == hy ? ly.l1 : 0.0f);
        if (wy == 0.0f) continue;
        for (int x = xlo; x <= xhi; x++) {
            const Lerp lx = lerp_coord(x, Wh, sx);
            const float wx = (lx.i0 == hx ? lx.l0 : 0.0f) + (lx.i1 == hx ? lx.l1 : 0.0f);
            if (wx == 0.0f) continue;
            const float w = wy * wx;
            const float* p = du + (size_t)y * W + x;
#pragma unroll
            for (int c = 0; c < AC; c++) acc[c] = __builtin_fmaf(w, p[c * plane], acc[c]);
        }
    }
    const size_t hplane = (size_t)Hh * Wh;
#pragma unroll
    for (int c = 0; c < AC; c++) {
        const size_t o = c * hplane + (size_t)hy * Wh + hx;
        dfeat[o] = (!relu_of || relu_of[o] > 0.0f) ? acc[c] : 0.0f;
    }
}

// The same adjoint for the scales the network produces (about 1/2): one workgroup per 8 x 32 tile of feat.  The output-space
// region that can touch the tile is staged in LDS once per channel and reduced separably, first along x with the tile
// columns' weights, then along y -- each du element is read once (plus the region overlap) instead of ~6 times.
constexpr int UB_HT = 8, UB_WT = 32, UB_ROWS = 26, UB_COLS = 80, UB_TAPS = 5;
inline bool upsample_bwd_tiled_ok(float sy, float sx) { return sy >= 0.41f && sx >= 0.435f; }  // regions and windows fit

__global__ __launch_bounds__(256) void app_upsample_bwd_tiled_kernel(int H, int W, int Hh, int Wh, float sy, float sx,
                                                                     const float* __restrict__ du, const float* __restrict__ relu_of,
                                                                     float* __restrict__ dfeat)
{
    __shared__ float region[UB_ROWS * UB_COLS];
    __shared__ float rowsum[UB_ROWS * UB_WT];
    const int tid = threadIdx.x, cx = tid & 31, cy = tid >> 5;
    const int hx0 = blockIdx.x * UB_WT, hy0 = blockIdx.y * UB_HT;
    const int hx = hx0 + cx, hy = hy0 + cy;
    // first output pixel that can have source index >= h - 1 (two pixels of slack for the float division)
    auto first_dst = [](int h, float s) { return max(0, (int)floorf((float)(h - 1) / s) - 1); };
    const int y_lo = first_dst(hy0, sy), x_lo = first_dst(hx0, sx);
    // this thread's tap windows inside the region: at these scales a source pixel is touched by at most 5 consecutive
    // destination pixels (2 / s < 5); the window starts on the first of them, found among the first candidates
    auto weight = [](int dst, int n_dst, int h, int n_in, float s) {
        const Lerp l = lerp_coord(min(dst, n_dst - 1), n_in, s);
        return dst < n_dst ? (l.i0 == h ? l.l0 : 0.0f) + (l.i1 == h ? l.l1 : 0.0f) : 0.0f;
    };
    int xw = first_dst(hx, sx) - x_lo, yw = first_dst(hy, sy) - y_lo;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (weight(x_lo + xw, W, hx, Wh, sx) == 0.0f) xw++;
        if (weight(y_lo + yw, H, hy, Hh, sy) == 0.0f) yw++;
    }
    float wx[UB_TAPS], wy[UB_TAPS];
#pragma unroll
    for (int j = 0; j < UB_TAPS; j++) {
        wx[j] = xw + j < UB_COLS ? weight(x_lo + xw + j, W, hx, Wh, sx) : 0.0f;
        wy[j] = yw + j < UB_ROWS ? weight(y_lo + yw + j, H, hy, Hh, sy) : 0.0f;
    }
    const size_t plane = (size_t)H * W, hplane = (size_t)Hh * Wh;
    constexpr int PER_THREAD = (UB_ROWS * UB_COLS + 255) / 256;
    float pre[PER_THREAD];
    auto fetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; i++) {
            const int idx = tid + 256 * i, ry = idx / UB_COLS, rx = idx - ry * UB_COLS;
            const int y = y_lo + ry, x = x_lo + rx;
            pre[i] = (idx < UB_ROWS * UB_COLS && y < H && x < W) ? du[c * plane + (size_t)y * W + x] : 0.0f;
        }
    };
    fetch(0);
    for (int c = 0; c < AC; c++) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; i++)
            if (tid + 256 * i < UB_ROWS * UB_COLS) region[tid + 256 * i] = pre[i];
        __syncthreads();
        if (c + 1 < AC) fetch(c + 1);
#pragma unroll
        for (int i = 0; i < (UB_ROWS + 7) / 8; i++) {
            const int ry = cy + 8 * i;
            if (ry < UB_ROWS) {
                float v = 0.0f;
#pragma unroll
                for (int j = 0; j < UB_TAPS; j++) v = __builtin_fmaf(wx[j], region[ry * UB_COLS + min(xw + j, UB_COLS - 1)], v);
                rowsum[ry * UB_WT + cx] = v;
            }
        }
        __syncthreads();
        float v = 0.0f;
#pragma unroll
        for (int j = 0; j < UB_TAPS; j++) v = __builtin_fmaf(wy[j], rowsum[min(yw + j, UB_ROWS - 1) * UB_WT + cx], v);
        if (hx < Wh && hy < Hh) {
            const size_t o = c * hplane + (size_t)hy * Wh + hx;
            dfeat[o] = (!relu_of || relu_of[o] > 0.0f) ? v : 0.0f;
        }
    }
}

inline float up_scale(int n_in, int n_out) { return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.0f; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int MAX_PARTIAL_ROWS = 1024;
constexpr int CONV3_BWD_GRID = 512;    // 2 workgroups (55 KB of LDS each, ~200 VGPRs) per CU
constexpr int CONV3_FWD_GRID = 512;    // 2 workgroups (79 KB of LDS each) per CU
constexpr int UP_FWD_GRID = 1024;      // 22 KB of LDS: 4 workgroups per CU
constexpr int WGRAD_GRID = 512;        // 2 workgroups (80 KB of LDS each) per CU
constexpr int CONV16_FWD_GRID = 512;  // 2 workgroups (62 KB of LDS each) per CU x 256 CUs
constexpr int CONV16_BWD_GRID = 512;  // 2 per CU (188 VGPRs; forcing 3 waves per SIMD spills and measured slower)

}  // namespace

size_t gsr_appearance_layout(int H, int W, size_t* dz2_off, size_t* du_off, size_t* p3_off, size_t* p2_off)
{
    const size_t act = ((size_t)AC * H * W * sizeof(float) + 255) & ~(size_t)255;
    size_t off = 0;
    *dz2_off = off; off += act;
    *du_off = off; off += act;
    *p3_off = off; off += ((size_t)MAX_PARTIAL_ROWS * B2_PARTIAL * sizeof(float) + 255) & ~(size_t)255;
    *p2_off = off; off += (size_t)MAX_PARTIAL_ROWS * B3_PARTIAL * sizeof(float);
    return off;
}

void gsr_launch_appearance_forward(int H, int W, int Hh, int Wh, const float* feat, const float* w2, const float* b2, const float* w3,
                                   const float* b3, const float* image, float* mapping, float* transformed, float* saved_a, hipStream_t s)
{
    const int tx = cdiv(W, TW), tiles = tx * cdiv(H, TH_BIG), tiles8 = tx * cdiv(H, TH_SMALL);
    const float sy = up_scale(Hh, H), sx = up_scale(Wh, W);
    float* saved_u = saved_a + (size_t)AC * H * W;
    app_conv16_kernel<0><<<tiles8 < CONV16_FWD_GRID ? tiles8 : CONV16_FWD_GRID, 256, 0, s>>>(
        H, W, Hh, Wh, sy, sx, tx, tiles8, upsample_staged_ok(sy, sx) ? 1 : 0, feat, w2, b2, saved_a, saved_u);
    app_conv3_fwd_kernel<<<tiles < CONV3_FWD_GRID ? tiles : CONV3_FWD_GRID, 256, 0, s>>>(H, W, tx, tiles, saved_a, w3, b3, image, mapping, transformed);
}

void gsr_launch_appearance_backward(int H, int W, int Hh, int Wh, const float* feat, const float* w2, const float* w3, const float* image,
                                    const float* mapping, const float* saved_a, const float* g_transformed, const float* g_mapping,
                                    float* dimage, float* dfeat, float* dw2, float* db2, float* dw3, float* db3, char* scratch,
                                    int mask_relu, hipStream_t s)
{
    size_t o_dz2, o_du, o_p3, o_p2;
    gsr_appearance_layout(H, W, &o_dz2, &o_du, &o_p3, &o_p2);
    float* dz2 = (float*)(scratch + o_dz2);
    float* du = (float*)(scratch + o_du);
    float* p3 = (float*)(scratch + o_p3);
    float* p2 = (float*)(scratch + o_p2);
    const float sy = up_scale(Hh, H), sx = up_scale(Wh, W);
    const int tx = cdiv(W, TW), tiles_small = tx * cdiv(H, TH_SMALL);
    const int rows = tiles_small < CONV3_BWD_GRID ? tiles_small : CONV3_BWD_GRID;
    const int rows2 = tiles_small < WGRAD_GRID ? tiles_small : WGRAD_GRID;
    const float* saved_u = saved_a + (size_t)AC * H * W;
    app_conv3_bwd_kernel<<<rows, 256, 0, s>>>(H, W, tx, tiles_small, saved_a, w3, image, mapping, g_transformed, g_mapping, dimage, dz2, p3);
    app_conv2_bwd_weight_kernel<false><<<rows2, 256, 0, s>>>(H, W, tx, tiles_small, saved_u, dz2, p2);
    app_conv16_kernel<1><<<tiles_small < CONV16_BWD_GRID ? tiles_small : CONV16_BWD_GRID, 256, 0, s>>>(H, W, Hh, Wh, sy, sx, tx, tiles_small, 0,
                                                                                                 dz2, w2, nullptr, du, nullptr);
    app_reduce_partials_kernel<<<cdiv(27 * AC + AO + AC + B3_PARTIAL, 4), 256, 0, s>>>(p3, rows, p2, rows2, dw3, db3, db2, dw2);
    if (upsample_bwd_tiled_ok(sy, sx))
        app_upsample_bwd_tiled_kernel<<<dim3(cdiv(Wh, UB_WT), cdiv(Hh, UB_HT)), 256, 0, s>>>(H, W, Hh, Wh, sy, sx, du, mask_relu ? feat : nullptr, dfeat);
    else
        app_upsample_bwd_kernel<<<dim3(cdiv(Wh, 64), cdiv(Hh, 4)), 256, 0, s>>>(H, W, Hh, Wh, sy, sx, du, mask_relu ? feat : nullptr, dfeat);
}

size_t gsr_appearance_up_scratch_bytes_impl() { return (size_t)WGRAD_GRID * UP_PARTIAL * sizeof(float); }

void gsr_launch_appearance_up_forward(int Hf, int Wf, const float* x, const float* w, const float* b, float* feat, hipStream_t s)
{
    const int tx = cdiv(Wf, TW), tiles = tx * cdiv(Hf, TH_SMALL);
    app_up_fwd_kernel<<<tiles < UP_FWD_GRID ? tiles : UP_FWD_GRID, 256, 0, s>>>(Hf, Wf, tx, tiles, x, w, b, feat);
}

void gsr_launch_appearance_up_backward(int Hf, int Wf, const float* x, const float* w, const float* dz, float* dx, float* dw, float* db,
                                       char* scratch, hipStream_t s)
{
    const int tx = cdiv(Wf, TW), tiles = tx * cdiv(Hf, TH_SMALL);
    const int rows = tiles < WGRAD_GRID ? tiles : WGRAD_GRID;
    float* partials = (float*)scratch;
    app_conv2_bwd_weight_kernel<true><<<rows, 256, 0, s>>>(Hf, Wf, tx, tiles, x, dz, partials);
    app_conv16_kernel<2><<<tiles < CONV16_BWD_GRID ? tiles : CONV16_BWD_GRID, 256, 0, s>>>(Hf, Wf, Hf / 2, Wf / 2, 0.0f, 0.0f, tx, tiles, 0, dz, w,
                                                                                           nullptr, dx, nullptr);
    app_up_reduce_kernel<<<cdiv(AC * UC * 9 + AC, 4), 256, 0, s>>>(partials, rows, dw, db);
}

