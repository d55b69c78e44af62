// Shared declarations of the MI355X (gfx950) Gaussian-rasterizer library.
// Internal header: the public C ABI is include/gsrast_mi355x.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define GSR_TILE_X 16
#define GSR_TILE_Y 16
#define GSR_TILE_PIX 256
#define GSR_SPLAT_F 16   // floats per packed per-Gaussian record (64 B, 64-byte aligned: a gather never straddles a line)
#ifndef GSR_GROW_F
#define GSR_GROW_F 9     // floats per per-instance gradient row: the nine sums, 36 B (12 = the 48-byte rows of rounds 1-3, three aligned float4)
#endif
#define GSR_SUMS_F 12    // row stride of the gsr_debug_capture_sums test hook

// Packed per-Gaussian record written by the preprocess kernel and gathered by the blend kernels.
// One 64-byte row (16 B quarters) instead of the reference's four separate arrays
// (means2D / conic_opacity / rgb / depths, rasterizer_impl.h:29-45).  The forward blend reads the first three quarters.
// Fourth quarter = gradient-row bookkeeping of the backward: which tiles of the Gaussian's rectangle its alpha >= 1/255 ellipse
// can reach at all (gsr_tile_cannot_touch; 2/3 of the instances of a typical frame cannot be reached), as a bit mask over the
// rectangle in row-major order, and the number of the Gaussian's first gradient row.  Only reachable instances own a row.
// Rectangles of more than 64 tiles set SP_BIG instead: every instance owns a row (numbered by its position in the rectangle).
enum SplatField {
    SP_MX = 0, SP_MY = 1, SP_CX = 2, SP_CY = 3, SP_CZ = 4, SP_OPA = 5,
    SP_R = 6, SP_G = 7, SP_B = 8, SP_DEPTH = 9, SP_RADIUS = 10 /* int bits */, SP_OFFS = 11 /* uint bits: exclusive instance offset */,
    SP_ROWBASE = 12 /* uint bits: first gradient row */, SP_MASK_LO = 13, SP_MASK_HI = 14 /* reachable tiles of the rectangle */,
    SP_BIG = 15 /* uint bits: 1 = rectangle of more than 64 tiles, no mask */
};

// Per-instance gradient row written by the backward blend kernel, summed per Gaussian afterwards.
enum GradRowField { GR_MX = 0, GR_MY = 1, GR_CX = 2, GR_CY = 3, GR_CW = 4, GR_OPA = 5, GR_R = 6, GR_G = 7, GR_B = 8 };

struct GeomLayout {
    size_t splat, cov3D, clamped, tiles_touched, point_offsets, block_sums, block_offsets, total, radii,
        row_offsets /* inclusive prefix of the Gaussians' gradient-row counts */, row_block_sums, row_block_offsets,
        band_count /* ids per band of the banded tile binning: zeroed by the scan kernel, filled by emit_partition_kernel */,
        wg_first /* work split of the per-Gaussian backward: first Gaussian of every workgroup (gsr_part_*) */, bytes;
};
struct ImageLayout {
    size_t final_T, n_contrib, ranges, quad_hi, bytes;
};
#define GSR_BAND_COUNTERS 64  // most bands; band_mem = [64 unused words][nb * P ids]
// order[] = [T tile ids by descending list length][64 + 64 counts, one per band, of the tiles whose lists a workgroup sorts
// ("long") and of those only the 1024-thread kernel can sort ("mid")][T + T tile ids: those tiles, the segment of a band at
// its first tile id] (binning.hip)
#define GSR_LONG_COUNT(T) (T)
#define GSR_MID_COUNT(T) ((T) + GSR_BAND_COUNTERS)
#define GSR_LONG_LIST(T) ((T) + 2 * GSR_BAND_COUNTERS)
#define GSR_MID_LIST(T) (2 * (T) + 2 * GSR_BAND_COUNTERS)
// ... followed by [64 bands][256 length classes]: tiles of the band per class (the band's share of the counting sort behind order[])
#define GSR_CLASS_HIST(T) (3 * (T) + 2 * GSR_BAND_COUNTERS)

struct BinLayout {
    size_t keys_a, keys_b, vals_a, vals_b, hist_global, hist_blocks, table, totals, order, band, blend_log, fixed_bytes, bytes;
    uint32_t sort_blocks;
};

// 4-byte-aligned vector loads (a features_rest row is 45 floats: its 16-byte pieces are only dword-aligned; gfx950 global
// loads of 2-4 dwords need no more than that)
typedef float gsr_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float gsr_f3u __attribute__((ext_vector_type(3), aligned(4)));

// One gradient row in HBM: nine floats.  36-byte rows (GSR_GROW_F = 9): two dword-aligned 16-byte accesses + one dword; 48-byte rows (12):
// three aligned float4 with three padding floats.
__device__ __forceinline__ void gsr_row_store(float* __restrict__ rows, size_t row, const float* v)
{
    float* o = rows + row * GSR_GROW_F;
    if (GSR_GROW_F == 12) {
        float4* o4 = reinterpret_cast<float4*>(o);
        o4[0] = make_float4(v[0], v[1], v[2], v[3]);
        o4[1] = make_float4(v[4], v[5], v[6], v[7]);
        o4[2] = make_float4(v[8], 0.f, 0.f, 0.f);
    } else {
        gsr_f4u a, b;
        a.x = v[0]; a.y = v[1]; a.z = v[2]; a.w = v[3];
        b.x = v[4]; b.y = v[5]; b.z = v[6]; b.w = v[7];
        *reinterpret_cast<gsr_f4u*>(o) = a;
        *reinterpret_cast<gsr_f4u*>(o + 4) = b;
        o[8] = v[8];
    }
}
__device__ __forceinline__ void gsr_row_load(const float* __restrict__ rows, size_t row, float* v)
{
    const float* o = rows + row * GSR_GROW_F;
    if (GSR_GROW_F == 12) {
        const float4* o4 = reinterpret_cast<const float4*>(o);
        const float4 a = o4[0], b = o4[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        v[8] = o[8];
    } else {
        const gsr_f4u a = *reinterpret_cast<const gsr_f4u*>(o), b = *reinterpret_cast<const gsr_f4u*>(o + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        v[8] = o[8];
    }
}

static inline size_t gsr_align(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- work split of the per-Gaussian backward (gaussian_backward.hip) ---------------------------------------------------------
// A workgroup's cost is its Gaussians (the chain: ~600 B and ~300 instructions each) plus their gradient rows (48 B each): Gaussian
// i starts at cost f(i) = 8 i + (gradient rows before i) and belongs to workgroup f(i) >> sh.  sh = 10 (1024 cost units = 128
// Gaussians without rows, or 1024 rows) unless that needs more than gsr_part_cap(P) workgroups -- a scene with more than ~8 rows per
// Gaussian -- in which case sh grows and a workgroup walks its Gaussians in passes.  The kernels that compute the row offsets
// (emit_partition_kernel / emit_keys_kernel) store every workgroup's first Gaussian in GeomLayout::wg_first, so the split costs the
// backward one load per workgroup and its duration no longer depends on the order of the Gaussians in memory (a spatially sorted
// model puts all the many-row Gaussians of a close-up region into the same workgroups).
#ifndef GSR_PART_W
#define GSR_PART_W 8        // cost of a Gaussian in gradient rows (experiment switch; see profiles/README.md)
#endif
#ifndef GSR_PART_SHIFT
#define GSR_PART_SHIFT 10   // log2 of the cost a workgroup takes
#endif
__host__ __device__ static inline uint32_t gsr_part_cap(int P)
{
    return (2u * (((uint32_t)(P > 0 ? P : 1) + 127u) / 128u) + 8u) << (GSR_PART_SHIFT < 10 ? 10 - GSR_PART_SHIFT : 0);
}
__host__ __device__ static inline int gsr_part_shift(int P, uint32_t total_rows)
{
    const unsigned long long f = (unsigned long long)GSR_PART_W * (unsigned long long)(P > 0 ? P : 0) + total_rows;
    int sh = GSR_PART_SHIFT;
    while ((f >> sh) + 1ull > (unsigned long long)gsr_part_cap(P)) sh++;
    return sh;
}

static inline GeomLayout gsr_geom_layout(int P)
{
    GeomLayout L;
    size_t o = 0, p = (size_t)(P > 0 ? P : 1);
    size_t nb = (p + 255) / 256;
    L.splat = o; o = gsr_align(o + p * GSR_SPLAT_F * 4);
    L.cov3D = o; o = gsr_align(o + p * 6 * 4);
    L.clamped = o; o = gsr_align(o + p * 3);
    L.tiles_touched = o; o = gsr_align(o + p * 4);
    L.point_offsets = o; o = gsr_align(o + p * 4);
    L.block_sums = o; o = gsr_align(o + nb * 4);
    L.block_offsets = o; o = gsr_align(o + nb * 4);
    L.total = o; o = gsr_align(o + 16);
    L.radii = o; o = gsr_align(o + p * 4);
    L.row_offsets = o; o = gsr_align(o + p * 4);
    L.row_block_sums = o; o = gsr_align(o + nb * 4);
    L.row_block_offsets = o; o = gsr_align(o + nb * 4 + 16);  // + the total
    L.band_count = o; o = gsr_align(o + GSR_BAND_COUNTERS * 4);
    L.wg_first = o; o = gsr_align(o + ((size_t)gsr_part_cap(P) + 1) * 4);
    L.bytes = o;
    return L;
}
static inline ImageLayout gsr_image_layout(int W, int H)
{
    ImageLayout L;
    size_t o = 0, n = (size_t)W * H;
    size_t t = (size_t)((W + GSR_TILE_X - 1) / GSR_TILE_X) * ((H + GSR_TILE_Y - 1) / GSR_TILE_Y);
    if (n == 0) n = 1;
    if (t == 0) t = 1;
    L.final_T = o; o = gsr_align(o + n * 4);
    L.n_contrib = o; o = gsr_align(o + n * 4);
    L.ranges = o; o = gsr_align(o + t * 8);
    L.quad_hi = o; o = gsr_align(o + t * 16);  // per tile: largest n_contrib of each 8x8 quadrant (written with the blend log, render_forward.hip)
    L.bytes = o;
    return L;
}

#define GSR_SORT_THREADS 256
#define GSR_SORT_ITEMS 16
#define GSR_SORT_TILE (GSR_SORT_THREADS * GSR_SORT_ITEMS)

int gsr_tile_slices(int P, int* gpb_out);
size_t gsr_band_table_bytes(int W, int H);
size_t gsr_band_list_bytes(int P, int W, int H);

// The part of the binning buffer whose size does not depend on R comes first: the stages that only need it (instance
// offsets, band partition, tile histograms) can be enqueued before the host knows R (gsr_forward).
static inline BinLayout gsr_bin_layout(long R, int P, int W, int H)
{
    BinLayout L;
    size_t o = 0, r = (size_t)(R > 0 ? R : 1);
    L.sort_blocks = (uint32_t)((r + GSR_SORT_TILE - 1) / GSR_SORT_TILE);
    {   // tile binning (binning.hip): [slice][tile] counters, tile totals, launch order, band lists
        int gpb;
        const size_t ns = (size_t)gsr_tile_slices(P, &gpb);
        size_t t = (size_t)((W + GSR_TILE_X - 1) / GSR_TILE_X) * ((H + GSR_TILE_Y - 1) / GSR_TILE_Y);
        if (t == 0) t = 1;
        size_t tb = ns * t * 4;  // whole-image slices; the banded binning needs [bands * 64 slices][band tiles]
        if (gsr_band_table_bytes(W, H) > tb) tb = gsr_band_table_bytes(W, H);
        L.table = o; o = gsr_align(o + tb);
        L.totals = o; o = gsr_align(o + t * 4);
        L.order = o; o = gsr_align(o + (3 * t + 2 * GSR_BAND_COUNTERS + 256 * GSR_BAND_COUNTERS) * 4);  // tiles by descending list length (launch order of the blend kernels) + the long lists
        L.band = o; o = gsr_align(o + gsr_band_list_bytes(P, W, H));  // band counters + per-band Gaussian id lists
    }
    L.hist_global = o; o = gsr_align(o + 8 * 256 * 4);
    L.fixed_bytes = o;  // everything above is independent of R
    L.keys_a = o; o = gsr_align(o + r * 8);
    L.keys_b = o; o = gsr_align(o + r * 8);
    L.vals_a = o; o = gsr_align(o + r * 4);
    L.vals_b = o; o = gsr_align(o + r * 4);
    L.hist_blocks = o; o = gsr_align(o + (size_t)L.sort_blocks * 256 * 4);
    L.blend_log = o; o = gsr_align(o + r * 32);  // blend log (render_forward.hip -> render_backward.hip): four 64-bit lane masks per list position
    L.bytes = o;
    return L;
}

// Blend log: index (in 64-bit words) of the mask of (list position p = first position of the tile + k, quadrant q)
__host__ __device__ __forceinline__ size_t gsr_blend_log_slot(uint32_t p, uint32_t q) { return (size_t)p * 4 + q; }

// rasterizer_impl.cu:35-50 (number of key bits that hold the tile id)
static inline uint32_t gsr_higher_msb(uint32_t n)
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

// ---- launchers (one per stage), all asynchronous on `stream` ----
struct PreprocessArgs {
    int P, D, M, W, H;
    const float *means3D, *scales, *rotations, *opacities, *shs, *cov3D_precomp, *colors_precomp;
    const float *viewmatrix, *projmatrix, *campos;
    float scale_modifier, tan_fovx, tan_fovy;
    int* radii;
    float* splat; float* cov3D; uint8_t* clamped; uint32_t* tiles_touched; uint32_t* block_sums; uint32_t* row_block_sums;
    // raw-parameter mode (gsr_forward_params): scales / rotations / opacities hold the PRE-activation parameters of
    // scene/gaussian_model.py:102-122 and the SH rows come from the two tensors features_dc [P][1][3] / features_rest [P][M-1][3]
    int raw;
    const float *features_dc, *features_rest;
    uint32_t* zero64;  // the 64 band counters of the tile binning (zeroed here so that no kernel has to run in between just for that)
    // prefiltered (auxiliary.h:156-160: the reference traps when a point is culled although the caller promised none would be): a point that
    // fails the near-plane test stores trap_tag -- this call's non-zero tag, so the word never needs zeroing -- into *trap_flag (NULL: off)
    int prefiltered = 0;  // host side: launch_geometry (api.hip) turns it into the two fields below
    uint32_t* trap_flag = nullptr; uint32_t trap_tag = 0;
};
void gsr_launch_preprocess(const PreprocessArgs& a, hipStream_t s);
void gsr_launch_scan_blocks(int nblocks, const uint32_t* block_sums, uint32_t* block_offsets, uint32_t* total,
                            unsigned long long* mailbox, uint32_t seq, hipStream_t s);
// the same with a second, independent array scanned in the same launch (in2 may be NULL); total2 = out2 + nblocks
// zero64 (may be NULL): 64 words the kernel zeroes (the band counters of the tile binning, filled by the next stage)
void gsr_launch_scan_blocks2(int nblocks, const uint32_t* block_sums, uint32_t* block_offsets, uint32_t* total, const uint32_t* in2,
                             uint32_t* out2, unsigned long long* mailbox, uint32_t seq, uint32_t* zero64, hipStream_t s);
// row_block_offsets[ceil(P / 256)] is the total number of gradient rows (scan_blocks2); wg_first: GeomLayout::wg_first
void gsr_launch_emit(int P, int W, int H, float* splat, const uint32_t* tiles_touched, const uint32_t* block_offsets,
                     uint32_t* point_offsets, const uint32_t* row_block_offsets, uint32_t* row_offsets, uint32_t* wg_first,
                     uint64_t* keys, uint32_t* vals, hipStream_t s);
// banded tile binning: the instance / gradient-row offsets of gsr_launch_emit (no keys) and the per-band id lists in one pass
// over the splat records; false (nothing launched) when the banded binning does not apply to this tile grid
// (it scans the preprocess kernel's block sums itself; mailbox != NULL: workgroup 0 also publishes R there, tagged with seq)
bool gsr_launch_emit_partition(int P, int W, int H, float* splat, const uint32_t* tiles_touched, const uint32_t* block_sums,
                               const uint32_t* row_block_sums, uint32_t* total, uint32_t* row_total, unsigned long long* mailbox,
                               uint32_t seq, uint32_t* point_offsets, uint32_t* row_offsets, uint32_t* wg_first,
                               uint32_t* band_count, uint32_t* band_mem, hipStream_t s);
// true (and the band count / height in tile rows) when the banded tile binning applies to this tile grid
bool gsr_band_geometry(int gx, int gy, int* nb, int* bh);
// returns 0 if the sorted pairs end in (ka,va), 1 if in (kb,vb)
int gsr_launch_radix_sort(uint64_t* ka, uint64_t* kb, uint32_t* va, uint32_t* vb, uint32_t n, int end_bit,
                          uint32_t* hist_global, uint32_t* hist_blocks, uint32_t nblocks, hipStream_t s);
static inline int gsr_sort_result_in_b(int end_bit) { return (((end_bit + 7) / 8) & 1); }
// head: everything that does not need the R-sized arrays (band partition, tile histograms, column scan -> totals);
// tail: scatter into the tile segments + per-tile sorts
// band_count: the band counters in the geometry buffer (GeomLayout::band_count); band_mem: the per-band id lists
void gsr_launch_tile_binning_head(int P, int W, int H, const float* splat, uint32_t* table, uint32_t* totals, const uint32_t* band_count,
                                  uint32_t* band_mem, hipStream_t s);
void gsr_launch_tile_binning_tail(int P, int W, int H, const float* splat, uint32_t* table, uint32_t* totals, uint32_t* order,
                                  uint32_t* ranges, const uint32_t* band_count, uint32_t* band_mem, uint64_t* bins, uint32_t* point_list,
                                  uint64_t* keys_sorted, uint64_t* scratch /* R keys: lists beyond the LDS budget are partitioned through it (may be keys_sorted) */,
                                  uint32_t* hint, uint32_t hint_seq, bool launch_mid, hipStream_t s);
void gsr_launch_tile_ranges(uint32_t R, const uint64_t* keys_sorted, uint32_t* ranges, uint32_t ntiles, hipStream_t s);
void gsr_launch_render_forward(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* splat,
                               const float* bg, float* final_T, uint32_t* n_contrib, float* out_color, const uint32_t* order,
                               unsigned long long* blend_log, uint32_t* quad_hi, hipStream_t s);
void gsr_launch_render_backward(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* splat,
                                const float* bg, const float* final_T, const uint32_t* n_contrib, const float* dL_dpix,
                                const uint32_t* order, float* grad_rows, float mean_list, const unsigned long long* blend_log,
                                const uint32_t* quad_hi, hipStream_t s);
int gsr_blend_profile_fetch(unsigned long long* out8);
int gsr_forward_profile_fetch(unsigned long long* out8);  // render_forward.hip: the same for the forward's quadrant walk  // render_backward.hip: phase profile of the log walk (profiling builds)
bool gsr_blend_log_for(int64_t R, int W, int H);  // does this frame's forward log its blend decisions for the backward? (api.hip)
int gsr_variant(const char* name, int dflt);  // experiment switch read from the environment (GSR_<name>)
bool gsr_cull_enabled();                      // tile lists without the instances the splat's alpha >= 1/255 ellipse cannot reach
struct GaussBackwardArgs {
    int P, D, M, W, H;
    const float *means3D, *shs, *scales, *rotations, *cov3D_in /* precomp or computed */, *viewmatrix, *projmatrix, *campos;
    const int* radii; const uint8_t* clamped; const float* splat; const uint32_t* tiles_touched; const uint32_t* point_offsets;
    const uint32_t* row_offsets;
    float scale_modifier, tan_fovx, tan_fovy;
    int has_shs, has_scales;
    const float* grad_rows;
    float *dL_dmeans2D, *dL_dcolors, *dL_dopacity, *dL_dmeans3D, *dL_dcov3D, *dL_dsh, *dL_dscales, *dL_drots;
    // raw-parameter mode (gsr_backward_params): scales / rotations are pre-activation, gradients are w.r.t. the raw
    // parameters, dL_dsh is split into dL_dfeatures_dc / dL_dfeatures_rest; dL_dcolors / dL_dcov3D may be NULL
    int raw;
    const float *features_dc, *features_rest;
    float *dL_dfeatures_dc, *dL_dfeatures_rest;
    const uint32_t* wg_first;   // GeomLayout::wg_first (NULL: every workgroup takes GB_BLOCK consecutive Gaussians)
    const uint32_t* row_total;  // total number of gradient rows (row_block_offsets[ceil(P / 256)])
    uint32_t rows_bound;        // what the host knows about it: <= num_rendered
    float* sums_out;  // test hook (gsr_debug_capture_sums): the nine reduced blend sums per Gaussian, [P][GSR_GROW_F], or NULL
};
void gsr_launch_gaussian_backward(const GaussBackwardArgs& a, hipStream_t s);
int gsr_render_backward_ablate();  // the compile-time timing ablations of the two kernels (0 in every shipped build)
int gsr_binning_ablate();
void gsr_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);

size_t gsr_appearance_layout(int H, int W, size_t* dz2_off, size_t* du_off, size_t* p3_off, size_t* p2_off);
void gsr_launch_appearance_forward(int H, int W, int Hh, int Wh, const float* feat, const float* w2, const float* b2, const float* w3,
                                   const float* b3, const float* image, float* mapping, float* transformed, float* saved_a, hipStream_t s);
void gsr_launch_appearance_backward(int H, int W, int Hh, int Wh, const float* feat, const float* w2, const float* w3, const float* image,
                                    const float* mapping, const float* saved_a, const float* g_transformed, const float* g_mapping,
                                    float* dimage, float* dfeat, float* dw2, float* db2, float* dw3, float* db3, char* scratch,
                                    int mask_relu, hipStream_t s);
size_t gsr_appearance_up_scratch_bytes_impl();
size_t gsr_mse_scratch_bytes_impl(int rows);
size_t gsr_bias_relu_scratch_bytes_impl(int C);
void gsr_launch_bias_relu_forward(int C, long long n, float* x, const float* bias, hipStream_t s);
void gsr_launch_bias_relu_backward(int C, long long n, const float* g, const float* y, float* dz, float* db, double* partials, hipStream_t s);
void gsr_launch_mse(int rows, long long n, const float* a, const float* b, float* out, double* partials, hipStream_t s);
void gsr_launch_appearance_up_forward(int Hf, int Wf, const float* x, const float* w, const float* b, float* feat, hipStream_t s);
void gsr_launch_appearance_up_backward(int Hf, int Wf, const float* x, const float* w, const float* dz, float* dx, float* dw, float* db,
                                       char* scratch, hipStream_t s);
size_t gsr_loss_scratch_layout(int C, int H, int W, size_t* maps_off, size_t* partials_off);
void gsr_launch_loss_forward(int C, int H, int W, const float* img1, const float* img2, const float* img_l1, float lambda,
                             float* maps, float* partials, float* out3, hipStream_t s);
void gsr_launch_loss_backward(int C, int H, int W, const float* img1, const float* img2, const float* img_l1, const float* maps,
                              float lambda, const float* grad_loss, float* dL_dimg1, float* dL_dimg_l1, hipStream_t s);

struct GsrAdamTensorHost {  // = gsr_adam_tensor of the C ABI
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t n; double lr; int64_t step;
};
int gsr_launch_adam(int n_tensors, const GsrAdamTensorHost* t, double beta1, double beta2, double eps, hipStream_t s);

struct GsrCompactTensorHost { const float* src; float* dst; int64_t row_floats; };  // = gsr_compact_tensor of the C ABI
void gsr_launch_densification_stats(int P, const int* radii, const float* vgrad, float* max_radii2D, float* accum, float* denom,
                                    hipStream_t s);
void gsr_launch_compact_count(int P, const uint8_t* keep, uint32_t* block_sums, hipStream_t s);
void gsr_launch_compact_rows(int P, const uint8_t* keep, const uint32_t* block_offsets, int n_tensors, const GsrCompactTensorHost* t,
                             hipStream_t s);

size_t gsr_knn_scratch_layout_bytes(int P);
int gsr_knn_run(int P, const float* points, float* out, void* scratch, size_t scratch_bytes, unsigned long long* evals, hipStream_t s);

// ---- device helpers ----
#ifdef __HIPCC__
// exp() of the parameter activations: same fully specified algorithm as the oracle's gs_expf (independently written).
__device__ __forceinline__ float gsr_expf(float x)
{
    // branch-free form of: if (x < -80) return 0; if (x > 80) x = 80; ...
    const float xc = __builtin_amdgcn_fmed3f(x, -80.0f, 80.0f);
    const float n = __builtin_rintf(xc * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, xc);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = __builtin_fmaf(p, r, 1.3981999507E-3f);
    p = __builtin_fmaf(p, r, 8.3334519073E-3f);
    p = __builtin_fmaf(p, r, 4.1665795894E-2f);
    p = __builtin_fmaf(p, r, 1.6666665459E-1f);
    p = __builtin_fmaf(p, r, 5.0000001201E-1f);
    const float y = __builtin_fmaf(p, r * r, r) + 1.0f;
    const float e = __int_as_float(__float_as_int(y) + (int)n * (1 << 23));
    return x < -80.0f ? 0.0f : e;
}

// The blend's exp(): the same fully specified algorithm as the oracle's gs_expf_blend (independently written), for arguments in
// [-80, 0] (pthr >= -80 is enforced when a list entry is staged and power <= 0 is tested before): nine instructions, seven of them
// full-rate fma -- the magic-number add rounds x * log2e to an integer inside an fma, the same fma without the integer gives the
// fraction exactly, a degree-5 Horner gives 2^f, and the integer goes into the exponent with one shift-add.
__device__ __forceinline__ float gsr_expf_blend(float x)
{
    const float L = 1.44269504088896341f, MAGIC = 12582912.0f;
    const float m = __builtin_fmaf(x, L, MAGIC);
    // (-(m - MAGIC) formed as MAGIC - m, exactly the same number: the fma then takes it without a negation modifier and the compiler can
    // use the VOP2 literal form -- with the modifier it put log2e into an SGPR, and a VALU instruction with an SGPR operand issues in 1.75 ns
    // instead of 1.06)
    const float nnf = MAGIC - m;
    const float f = __builtin_fmaf(x, L, nnf);
    float p = 0x1.5cbc2cp-10f;
    p = __builtin_fmaf(p, f, 0x1.3cea8ap-7f);
    p = __builtin_fmaf(p, f, 0x1.c6b48ep-5f);
    p = __builtin_fmaf(p, f, 0x1.ebf9bcp-3f);
    p = __builtin_fmaf(p, f, 0x1.62e42cp-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __uint_as_float(__float_as_uint(p) + (__float_as_uint(m) << 23));
}
// Wave-uniform "any lane" test.  __ballot() != 0 compiles to v_cndmask + v_cmp_ne on top of the compare that produced the
// predicate (two half-rate VALU instructions); the ballot intrinsic folds into a scalar test of the compare's own lane mask.
__device__ __forceinline__ bool gsr_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// Moves the workgroup's [rows][3M] tile between global memory and LDS (row stride `stride` floats in LDS).
__device__ __forceinline__ void sh_tile_load(const float* __restrict__ g, float* __restrict__ s, int rows, int n3, int stride)
{
    const int count = rows * n3;
    if ((n3 & 3) == 0) {
        const float4* g4 = reinterpret_cast<const float4*>(g);
        for (int f = threadIdx.x; f < (count >> 2); f += blockDim.x) {
            const float4 v = g4[f];
            const int e = f << 2, r = e / n3, c = e - r * n3;
            float* d = s + r * stride + c;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
        for (int e = threadIdx.x; e < count; e += blockDim.x) {
            const int r = e / n3, c = e - r * n3;
            s[r * stride + c] = g[e];
        }
    }
}
__device__ __forceinline__ void sh_tile_store(float* __restrict__ g, const float* __restrict__ s, int rows, int n3, int stride)
{
    const int count = rows * n3;
    if ((n3 & 3) == 0) {
        float4* g4 = reinterpret_cast<float4*>(g);
        for (int f = threadIdx.x; f < (count >> 2); f += blockDim.x) {
            const int e = f << 2, r = e / n3, c = e - r * n3;
            const float* d = s + r * stride + c;
            g4[f] = make_float4(d[0], d[1], d[2], d[3]);
        }
    } else {
        for (int e = threadIdx.x; e < count; e += blockDim.x) {
            const int r = e / n3, c = e - r * n3;
            g[e] = s[r * stride + c];
        }
    }
}

// Same for a column range [col0, col0 + ncols) of the LDS rows whose global source/destination is its own dense
// [rows][ncols] array (features_dc / features_rest and their gradients).  `g` must be 16-byte aligned (it is: a
// workgroup's base offset is a multiple of 256 rows); a float4 may straddle rows, so the row/column is stepped per element.
__device__ __forceinline__ void sh_cols_load(const float* __restrict__ g, float* __restrict__ s, int rows, int ncols, int stride, int col0)
{
    const int count = rows * ncols, n4 = count >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int f = threadIdx.x; f < n4; f += blockDim.x) {
        const float4 v = g4[f];
        const float e4[4] = {v.x, v.y, v.z, v.w};
        int r = (f << 2) / ncols, c = (f << 2) - r * ncols;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            s[r * stride + col0 + c] = e4[k];
            if (++c == ncols) { c = 0; r++; }
        }
    }
    for (int e = (n4 << 2) + threadIdx.x; e < count; e += blockDim.x) {
        const int r = e / ncols, c = e - r * ncols;
        s[r * stride + col0 + c] = g[e];
    }
}
__device__ __forceinline__ void sh_cols_store(float* __restrict__ g, const float* __restrict__ s, int rows, int ncols, int stride, int col0)
{
    const int count = rows * ncols, n4 = count >> 2;
    float4* g4 = reinterpret_cast<float4*>(g);
    for (int f = threadIdx.x; f < n4; f += blockDim.x) {
        float e4[4];
        int r = (f << 2) / ncols, c = (f << 2) - r * ncols;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            e4[k] = s[r * stride + col0 + c];
            if (++c == ncols) { c = 0; r++; }
        }
        g4[f] = make_float4(e4[0], e4[1], e4[2], e4[3]);
    }
    for (int e = (n4 << 2) + threadIdx.x; e < count; e += blockDim.x) {
        const int r = e / ncols, c = e - r * ncols;
        g[e] = s[r * stride + col0 + c];
    }
}

// Activations of scene/gaussian_model.py:33-40 in the arithmetic the oracle restates (oracle/gs_oracle.c, orc_activate):
// scaling = exp, opacity = sigmoid, rotation = x / max(||x||, 1e-12) (torch.nn.functional.normalize).
__device__ __forceinline__ float gsr_sigmoidf(float x) { return 1.0f / (1.0f + gsr_expf(-x)); }
__device__ __forceinline__ float gsr_quat_norm(float r, float x, float y, float z) { return sqrtf(((r * r + x * x) + y * y) + z * z); }

// Conservative tile-level rejection used while staging a tile's list: returns true only if NO pixel centre of the
// 16x16 tile whose first pixel is (x0, y0) can reach power >= pthr, i.e. alpha >= 1/255 (power = -0.5*q,
// q = cx*dx^2 + 2*cy*dx*dy + cz*dy^2).  q is convex, so for a mean outside the tile its minimum over the tile lies on
// one of the four edges; each edge minimum is closed-form.  The margin covers the rounding of both this estimate and
// the per-pixel evaluation (relative 1e-5 of the term magnitudes), so a rejected entry can never pass the exact
// per-pixel tests: dropping it from the walk cannot change n_contrib, final_T or the colour.  On the 500k-Gaussian
// benchmark scene 2/3 of all (tile, Gaussian) list entries are rejected here (the lists are built from the square
// around the 3-sigma circle of the LARGER axis, auxiliary.h:46-56, regardless of opacity and anisotropy).
__device__ __forceinline__ bool gsr_tile_cannot_touch(float mx, float my, float cx, float cy, float cz, float pthr, float x0, float y0)
{
    if (!(pthr > -INFINITY)) return false;  // opacity <= 0 or NaN: leave everything to the exact path
    const float t = -2.0f * pthr;           // need q <= t somewhere in the tile (t < 0: impossible)
    const float dxl = x0 - mx, dxh = x0 + 15.0f - mx, dyl = y0 - my, dyh = y0 + 15.0f - my;
    if (dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f) return false;  // mean inside the tile
    float qmin = INFINITY, qabs = 0.f;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        {   // vertical edge dx = const
            const float dx = e ? dxh : dxl;
            const float dy = fminf(fmaxf(-(cy * dx) / cz, dyl), dyh);
            const float a = cx * dx * dx, b = cz * dy * dy, c = 2.0f * cy * dx * dy;
            const float q = a + b + c;
            if (!(q >= qmin)) { qmin = q; qabs = a + b + fabsf(c); }
        }
        {   // horizontal edge dy = const
            const float dy = e ? dyh : dyl;
            const float dx = fminf(fmaxf(-(cy * dy) / cx, dxl), dxh);
            const float a = cx * dx * dx, b = cz * dy * dy, c = 2.0f * cy * dx * dy;
            const float q = a + b + c;
            if (!(q >= qmin)) { qmin = q; qabs = a + b + fabsf(c); }
        }
    }
    return qmin > t + (1e-3f * fabsf(t) + 1e-5f * qabs + 1e-3f);  // NaN anywhere -> false (keep)
}

// Gradient rows a Gaussian owns: the reachable tiles of its rectangle, or all of them for a rectangle too large for the mask
__device__ __forceinline__ uint32_t gsr_splat_rows(uint32_t mask_lo, uint32_t mask_hi, uint32_t big, uint32_t tiles_touched)
{
    return big ? tiles_touched : (uint32_t)(__popc(mask_lo) + __popc(mask_hi));
}

// Called by the thread that has just computed Gaussian idx's inclusive gradient-row prefix: records idx + 1 as the first Gaussian of
// every workgroup of the per-Gaussian backward that starts between Gaussian idx and idx + 1 (gsr_part_shift above).
__device__ __forceinline__ void gsr_part_store(uint32_t* __restrict__ wg_first, int P, int idx, uint32_t rows_excl, uint32_t rows_incl,
                                               uint32_t total_rows)
{
    const int sh = gsr_part_shift(P, total_rows);
    const uint32_t w0 = (uint32_t)(((unsigned long long)GSR_PART_W * (unsigned long long)idx + rows_excl) >> sh);
    const uint32_t w1 = (uint32_t)(((unsigned long long)GSR_PART_W * (unsigned long long)(idx + 1) + rows_incl) >> sh);
    for (uint32_t w = w0 + 1; w <= w1; w++) wg_first[w] = (uint32_t)(idx + 1);
    if (idx == 0) wg_first[0] = 0u;
}

// relative margin of the reach tests as a function of the conic's conditioning kappa = cx cz / det (see gsr_reach_mask):
// 1e-3 for a round splat (the value every measurement of rounds 1-2 was taken with), +2e-6 per unit of kappa
#define GSR_REACH_REL(kappa) (1e-3f + 2e-6f * (kappa))

// Which tiles of a splat's rectangle [mnx, mxx) x [mny, mxy) (at most 64 tiles) can its alpha >= 1/255 ellipse reach?  Bit
// (ty - mny) * (mxx - mnx) + (tx - mnx) is set if any pixel centre of tile (tx, ty) may satisfy
// q = cx u^2 + 2 cy u v + cz v^2 <= t, t = 2 ln(255 opacity), (u, v) = pixel - mean.  CONSERVATIVE (margins far above the fp32
// rounding of the blend kernels' own evaluation): a cleared bit guarantees that no pixel of the tile blends the splat, so the
// instance needs no gradient row.  Per tile ROW the reachable tiles are an interval -- the ellipse cut by the row's band of pixel
// centres is convex -- whose ends are the band's extreme chord points: u(v) = -(cy/cx) v +- sqrt(cx t - det v^2) / cx is concave /
// convex in v, so each end is one evaluation at the clamped unconstrained optimum.  ~40 instructions per tile row instead of a
// ~100-instruction test per tile.
__device__ __forceinline__ unsigned long long gsr_reach_mask(float mx, float my, float cx, float cy, float cz, float opacity,
                                                             uint32_t mnx, uint32_t mny, uint32_t mxx, uint32_t mxy)
{
    const uint32_t w = mxx - mnx, h = mxy - mny, n = w * h;
    const unsigned long long all = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    if (!(opacity > 0.f)) return all;                       // opacity <= 0 or NaN: everything is left to the exact path
    const float t0 = 2.0f * __logf(255.0f * opacity);
    if (t0 + (2e-3f * fabsf(t0) + 5e-3f) < 0.f) return 0ull;  // alpha < 1/255 even at the centre
    const float det = cx * cz - cy * cy;
    if (!(cx > 0.f) || !(cz > 0.f) || !(det > 0.f) || !(det < INFINITY)) return all;  // not an ellipse: no statement
    // hardware reciprocal / square root (1 ulp): their error is orders of magnitude below the margins
    const float inv_cx = __builtin_amdgcn_rcpf(cx), inv_det = __builtin_amdgcn_rcpf(det);
    // Conditioning.  kappa = cx cz / det >= 1 grows with the square of the aspect ratio of a diagonal splat.  The fp32 products behind
    // det cancel (relative error ~1.2e-7 kappa), and so do the three terms of q in the blend kernels' own per-pixel evaluation
    // (cx dx^2 + cz dy^2 <= 2 kappa q): every margin below is therefore relative to GSR_REACH_REL(kappa), and beyond kappa = 4096
    // (a 130:1 needle on the diagonal) no statement is made at all.
    const float kappa = (cx * cz) * inv_det;
    if (!(kappa <= 4096.0f)) return all;
    const float rel = GSR_REACH_REL(kappa);
    const float t = t0 + (2.0f * rel * fabsf(t0) + 5e-3f);  // margin of the blend kernels' own skip threshold and then some
    const float vmax = __builtin_amdgcn_sqrtf(cx * t * inv_det) * (1.0f + rel) + 0.02f;   // |v| extent of the ellipse
    const float umax_g = __builtin_amdgcn_sqrtf(cz * t * inv_det);                  // |u| extent; it is reached at v = -+(cy / cz) * umax_g
    const float vstar = -(cy * __builtin_amdgcn_rcpf(cz)) * umax_g;                 // v of the right-most point (left-most: -vstar)
    const float slope = -cy * inv_cx;
    unsigned long long mask = 0ull;
    // tile rows whose band of pixel centres [16 ty, 16 ty + 15] meets [my - vmax, my + vmax]
    const int ty0 = max((int)mny, (int)fmaxf(ceilf((my - vmax - (float)(GSR_TILE_Y - 1)) * (1.0f / GSR_TILE_Y)), -1.0f));
    const int ty1 = min((int)mxy - 1, (int)fminf(floorf((my + vmax) * (1.0f / GSR_TILE_Y)), 1.0e6f));
    for (int ty = ty0; ty <= ty1; ty++) {
        const float vl = (float)(ty * GSR_TILE_Y) - my, vh = vl + (float)(GSR_TILE_Y - 1);
        // right end: maximise the concave u_max(v) over the band; left end: minimise the convex u_min(v)
        const float vr = fminf(fmaxf(vstar, vl), vh), vq = fminf(fmaxf(-vstar, vl), vh);
        const float dr = fmaxf(cx * t - det * vr * vr, 0.f), dq = fmaxf(cx * t - det * vq * vq, 0.f);
        const float ur = slope * vr + __builtin_amdgcn_sqrtf(dr) * inv_cx, ul = slope * vq - __builtin_amdgcn_sqrtf(dq) * inv_cx;
        const float m = 0.02f + rel * (fabsf(ur) + fabsf(ul));
        // tiles whose pixel centres [16 tx, 16 tx + 15] meet [mx + ul - m, mx + ur + m]
        const float lo_f = ceilf((mx + ul - m - (float)(GSR_TILE_X - 1)) * (1.0f / GSR_TILE_X));
        const float hi_f = floorf((mx + ur + m) * (1.0f / GSR_TILE_X));
        const int lo = max((int)mnx, (int)fmaxf(lo_f, -1.0f)), hi = min((int)mxx - 1, (int)fminf(hi_f, 1.0e6f));
        if (lo > hi) continue;
        const uint32_t cnt = (uint32_t)(hi - lo + 1);
        const unsigned long long run = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
        mask |= run << (((uint32_t)ty - mny) * w + ((uint32_t)lo - mnx));
    }
    return mask & all;
}

// The same question one level down: which of the four 8x8 quadrants of the tile whose first pixel is (x0, y0) can the splat's
// alpha >= 1/255 ellipse reach?  Bit q = 2 * (y half) + (x half) -- the pixel block of wave q of the blend kernels.  Two bands
// of 8 pixel rows, each cut like a tile row in gsr_reach_mask; conservative in the same way (NaN anywhere: bit set).
__device__ __forceinline__ uint32_t gsr_quad_mask(float mx, float my, float cx, float cy, float cz, float opacity, float x0, float y0)
{
    if (!(opacity > 0.f)) return 15u;
    const float t0 = 2.0f * __logf(255.0f * opacity);
    if (t0 + (2e-3f * fabsf(t0) + 5e-3f) < 0.f) return 0u;
    const float det = cx * cz - cy * cy;
    if (!(cx > 0.f) || !(cz > 0.f) || !(det > 0.f) || !(det < INFINITY)) return 15u;
    const float inv_cx = __builtin_amdgcn_rcpf(cx), inv_det = __builtin_amdgcn_rcpf(det);
    const float kappa = (cx * cz) * inv_det;  // conditioning: see gsr_reach_mask
    if (!(kappa <= 4096.0f)) return 15u;
    const float rel = GSR_REACH_REL(kappa);
    const float t = t0 + (2.0f * rel * fabsf(t0) + 5e-3f);
    const float vmax = __builtin_amdgcn_sqrtf(cx * t * inv_det) * (1.0f + rel) + 0.02f;
    const float vstar = -(cy * __builtin_amdgcn_rcpf(cz)) * __builtin_amdgcn_sqrtf(cz * t * inv_det);
    const float slope = -cy * inv_cx;
    uint32_t mask = 0;
#pragma unroll
    for (int band = 0; band < 2; band++) {
        const float vl = y0 + (float)(8 * band) - my, vh = vl + 7.0f;
        if (vl > vmax || vh < -vmax) continue;
        const float vr = fminf(fmaxf(vstar, vl), vh), vq = fminf(fmaxf(-vstar, vl), vh);
        const float dr = fmaxf(cx * t - det * vr * vr, 0.f), dq = fmaxf(cx * t - det * vq * vq, 0.f);
        const float ur = slope * vr + __builtin_amdgcn_sqrtf(dr) * inv_cx, ul = slope * vq - __builtin_amdgcn_sqrtf(dq) * inv_cx;
        const float m = 0.02f + rel * (fabsf(ur) + fabsf(ul));
        const float A = mx + ul - m, B = mx + ur + m;  // pixel centres the band's cut of the ellipse spans
        if (!(A > x0 + 7.0f) && !(B < x0)) mask |= 1u << (2 * band);
        if (!(A > x0 + 15.0f) && !(B < x0 + 8.0f)) mask |= 2u << (2 * band);
    }
    return mask;
}

// auxiliary.h:46-56 -- tile rectangle of a splat (saturating float->int as the hardware converts)
__device__ __forceinline__ void gsr_get_rect(float px, float py, int max_radius, int gx, int gy,
                                             uint32_t& minx, uint32_t& miny, uint32_t& maxx, uint32_t& maxy)
{
    const float r = (float)max_radius;
    minx = (uint32_t)min(gx, max(0, (int)((px - r) / (float)GSR_TILE_X)));
    miny = (uint32_t)min(gy, max(0, (int)((py - r) / (float)GSR_TILE_Y)));
    maxx = (uint32_t)min(gx, max(0, (int)((px + r + (float)(GSR_TILE_X - 1)) / (float)GSR_TILE_X)));
    maxy = (uint32_t)min(gy, max(0, (int)((py + r + (float)(GSR_TILE_Y - 1)) / (float)GSR_TILE_Y)));
}
#endif
