/*
 * gsrast_mi355x.h -- C ABI of the MI355X (gfx950) Gaussian-splat rasterizer + kNN-init library
 * (libgsrast_mi355x.so).  This is the drop-in boundary for the ONE hot path of
 * 92401/Based-on-point-cloud-partitions: it replaces the pybind11/libtorch functions of
 * submodules/diff-gaussian-rasterization (rasterize_points.h:18-67, ext.cpp:15-19) and
 * submodules/simple-knn (spatial.h, ext.cpp:15-17).  Plain pointers and sizes only, no torch types.
 *
 * Conventions (identical to the reference):
 *  - every pointer is a DEVICE pointer to contiguous float32 / int32 / uint8 data unless marked "host";
 *  - optional inputs are passed as NULL (the reference passes empty tensors, rasterize_points.cu:94-111);
 *    exactly one of {shs, colors_precomp} and one of {scales+rotations, cov3D_precomp} must be non-NULL;
 *  - shs is [P][M][3], out_color is [3][H][W], viewmatrix / projmatrix are the transposed 4x4 matrices
 *    flattened row-major (element [4*c + r] = row r, column c), campos / bg are 3 floats on the device;
 *  - `stream` is a hipStream_t (NULL = default stream); all work is enqueued on it;
 *  - the three state buffers are opaque, owned by the caller, and must be kept alive between forward and
 *    backward (the reference does the same through ctx.save_for_backward, __init__.py:92-97).
 *  - return value: 0 on success, negative on error; gsr_last_error() describes the last error of the
 *    calling thread (the reference throws std::runtime_error / AT_ERROR).
 *
 * The forward is two-phase because the number of (Gaussian, tile) instances R is only known after the
 * per-Gaussian stage (the reference resizes its binning buffer through a std::function callback instead,
 * rasterize_points.cu:27-33, rasterizer_impl.cu:281-285).
 */
#ifndef GSRAST_MI355X_H
#define GSRAST_MI355X_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_ABI_VERSION 16

int gsr_abi_version(void);
const char* gsr_last_error(void);

/* Sizes in bytes of the opaque state buffers (required<GeometryState/ImageState/BinningState>,
 * rasterizer_impl.h:67-73) and of the backward scratch (one 36-byte gradient row -- nine floats -- per instance).
 * Binning: 24 bytes per instance as the reference (keys + values, double-buffered) + 4.25 for the radix path + 32 for the blend log
 * (four 64-bit lane masks per list position: which pixels of each 8x8 quadrant blended the entry, written by the forward for the
 * backward).  Image: 8 bytes per pixel + 8 per tile as the reference + 16 per tile (last contributor of
 * each quadrant). */
size_t gsr_geometry_bytes(int P);
size_t gsr_image_bytes(int W, int H);
size_t gsr_binning_bytes(int64_t R, int P, int W, int H);
size_t gsr_backward_scratch_bytes(int64_t R, int W, int H);
/* The one size limit below the reference's (whose 32-bit offsets allow 2^31 instances, rasterizer_impl.cu:280-285): the backward numbers
 * its gradient rows -- one per (Gaussian, tile) instance the splat can reach, so rows <= num_rendered -- in 28 bits.
 * gsr_gradient_row_limit() = 2^28 - 1; gsr_gradient_rows reads the row count a finished forward left in its geometry buffer (a 4-byte
 * copy + stream synchronisation: callers only need it when num_rendered itself is at or above the limit).  The Python operators check at
 * FORWARD time when a gradient will be asked for; gsr_backward / gsr_backward_params check again and return -1. */
int64_t gsr_gradient_row_limit(void);
int gsr_gradient_rows(int P, const void* geom_buffer, size_t geom_bytes, int64_t* rows_host, void* stream);

/* Phase 1 of CudaRasterizer::Rasterizer::forward (rasterizer_impl.cu:198-285): frustum cull, covariance
 * projection, SH colour, tiles_touched and its prefix sum.  Writes radii[P] and *num_rendered_host (host
 * pointer; the call waits for the prefix sum to read it, like the reference's cudaMemcpy at :281).
 * prefiltered != 0 (also in gsr_forward): the caller promises that no point fails the near-plane test; where one does, the
 * reference's kernel prints "Point is filtered although prefiltered is set. This shouldn't happen!" and traps
 * (auxiliary.h:154-160) -- here the call returns -6 with that message in gsr_last_error() (one more stream synchronisation per
 * forward in this mode; the reference's callers pass false, gaussian_renderer/__init__.py:47). */
int gsr_forward_geometry(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                         const float* rotations, const float* opacities, const float* shs, const float* cov3D_precomp,
                         const float* colors_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                         int W, int H, float tan_fovx, float tan_fovy, int prefiltered, int* radii, void* geom_buffer,
                         size_t geom_bytes, int64_t* num_rendered_host, void* stream);

/* Phase 2 (rasterizer_impl.cu:287-336): tile binning (per-tile lists sorted by depth, bit-identical to the reference's
 * key emission + 64-bit radix sort + tile ranges) and alpha compositing. */
int gsr_forward_render(int P, int64_t R, int W, int H, const float* background, void* geom_buffer, size_t geom_bytes,
                       void* binning_buffer, size_t binning_bytes, void* image_buffer, size_t image_bytes,
                       float* out_color, void* stream);

/* Both phases in one call, for callers that bring a binning buffer sized from an estimate (e.g. the previous step's R):
 * runs phase 1, reads R, and continues with phase 2 without returning to the caller if `binning_bytes` suffices.
 * Returns 0 when the image is complete, 1 when the binning buffer is too small (then *num_rendered_host is valid, phase 1
 * is done, and the caller allocates gsr_binning_bytes(R, P, W, H) bytes and calls gsr_forward_render), negative on error. */
int gsr_forward(int P, int D, int M, const float* background, const float* means3D, const float* scales, float scale_modifier,
                const float* rotations, const float* opacities, const float* shs, const float* cov3D_precomp,
                const float* colors_precomp, const float* viewmatrix, const float* projmatrix, const float* campos, int W, int H,
                float tan_fovx, float tan_fovy, int prefiltered, int* radii, void* geom_buffer, size_t geom_bytes,
                void* binning_buffer, size_t binning_bytes, void* image_buffer, size_t image_bytes, float* out_color,
                int64_t* num_rendered_host, void* stream);

/* CudaRasterizer::Rasterizer::backward (rasterizer_impl.cu:340-434) + the zero-initialised outputs of
 * RasterizeGaussiansBackwardCUDA (rasterize_points.cu:151-159): every output element is written.
 * Outputs: dL_dmeans2D[P][3], dL_dcolors[P][3], dL_dopacity[P], dL_dmeans3D[P][3], dL_dcov3D[P][6],
 * dL_dsh[P][M][3], dL_dscales[P][3], dL_drotations[P][4].
 * A reduction from the reference: the number of gradient rows (<= num_rendered; see gsr_gradient_row_limit above) must be below
 * 2^28 - 1 = 268,435,455 -- the blend backward keeps the row index in 28 bits beside a 4-bit quadrant mask -- or the call returns -1
 * (BASELINE.json's largest configuration has 1.5e7 instances).  The forward accepts any R below 2^32. */
int gsr_backward(int P, int D, int M, int64_t R, int W, int H, const float* background, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* campos, float tan_fovx, float tan_fovy, const int* radii, const void* geom_buffer,
                 size_t geom_bytes, const void* binning_buffer, size_t binning_bytes, const void* image_buffer,
                 size_t image_bytes, void* scratch, size_t scratch_bytes, const float* dL_dpix, float* dL_dmeans2D,
                 float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscales, float* dL_drotations, void* stream);

/* Raw-parameter operator: the rasterizer fed directly with the parameters GaussianModel stores
 * (scene/gaussian_model.py:48-56) -- xyz [P][3], features_dc [P][1][3], features_rest [P][M-1][3], opacity_raw [P][1],
 * scaling_raw [P][3], rotation_raw [P][4] -- instead of the activated copies gaussian_renderer/__init__.py:52-93 builds
 * per render with get_scaling = exp, get_rotation = normalize (x / max(||x||, 1e-12)), get_opacity = sigmoid and
 * get_features = cat(features_dc, features_rest) (gaussian_model.py:102-122).  Activations are evaluated inside the
 * per-Gaussian kernels, and the backward returns gradients w.r.t. the RAW parameters (what autograd hands the optimizer
 * in train.py:155-157), so the four activation passes, the 96 MB concat and their autograd mirrors never run.
 * gsr_forward_params has gsr_forward's protocol (returns 1 when the binning buffer is too small: allocate and call
 * gsr_forward_render).  The state buffers are interchangeable with gsr_forward's (gsr_debug_export works on both). */
int gsr_forward_params(int P, int D, int M, const float* background, const float* xyz, const float* features_dc,
                       const float* features_rest, const float* opacity_raw, const float* scaling_raw, const float* rotation_raw,
                       float scale_modifier, const float* viewmatrix, const float* projmatrix, const float* campos, int W, int H,
                       float tan_fovx, float tan_fovy, int* radii, void* geom_buffer, size_t geom_bytes, void* binning_buffer,
                       size_t binning_bytes, void* image_buffer, size_t image_bytes, float* out_color, int64_t* num_rendered_host,
                       void* stream);
/* Outputs: dL_dmeans2D[P][3], dL_dxyz[P][3], dL_dfeatures_dc[P][1][3], dL_dfeatures_rest[P][M-1][3] (NULL iff M == 1),
 * dL_dopacity_raw[P], dL_dscaling_raw[P][3], dL_drotation_raw[P][4]; every element is written. */
int gsr_backward_params(int P, int D, int M, int64_t R, int W, int H, const float* background, const float* xyz,
                        const float* features_dc, const float* features_rest, const float* scaling_raw, const float* rotation_raw,
                        float scale_modifier, const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                        float tan_fovy, const int* radii, const void* geom_buffer, size_t geom_bytes, const void* binning_buffer,
                        size_t binning_bytes, const void* image_buffer, size_t image_bytes, void* scratch, size_t scratch_bytes,
                        const float* dL_dpix, float* dL_dmeans2D, float* dL_dxyz, float* dL_dfeatures_dc, float* dL_dfeatures_rest,
                        float* dL_dopacity_raw, float* dL_dscaling_raw, float* dL_drotation_raw, void* stream);

/* markVisible (rasterize_points.cu:198-217): present[i] = 1 iff the point passes the near-plane test. */
int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream);

/* distCUDA2 (submodules/simple-knn/spatial.cu:15-26): out[i] = mean squared distance from point i to its 3
 * nearest other points.  points is [P][3]. */
size_t gsr_knn_scratch_bytes(int P);
int gsr_knn_mean_dist2(int P, const float* points, float* out, void* scratch, size_t scratch_bytes, void* stream);

/* Loss block of the training step: loss = (1 - lambda) * L1(img_l1, img2) + lambda * (1 - SSIM(img1, img2)),
 * the reference's utils/loss_utils.py:17-18,23-63 combined as in train.py:112-117 (11-tap Gaussian window, sigma 1.5, zero
 * padding, mean over C*H*W).  img1 = rendered image, img2 = ground truth; img_l1 = the image the L1 term is taken on:
 * NULL means img1 (plain 3DGS), the reference's train.py passes the appearance-decoupled image there (train.py:112,116).
 * Images are [C][H][W] float32 on the device.  gsr_loss_forward writes out3 = {loss, L1, SSIM} (device) and keeps the per-pixel
 * partial derivatives in `scratch` (gsr_loss_scratch_bytes); gsr_loss_backward turns them into dL/dimg1 (the dL/dcolor the
 * rasterizer's backward consumes) and, when img_l1 is given, dL/dimg_l1, both scaled by *grad_loss (device scalar, NULL = 1). */
size_t gsr_loss_scratch_bytes(int C, int H, int W);
int gsr_loss_forward(int C, int H, int W, const float* img1, const float* img2, const float* img_l1, float lambda_dssim,
                     void* scratch, size_t scratch_bytes, float* out3, void* stream);
int gsr_loss_backward(int C, int H, int W, const float* img1, const float* img2, const float* img_l1, float lambda_dssim,
                      const void* scratch, size_t scratch_bytes, const float* grad_loss, float* dL_dimg1, float* dL_dimg_l1,
                      void* stream);

/* Appearance decoupling of the reference's training loop (train.py:110-111): the full-resolution tail of
 * AppearanceNetwork.forward (scene/ptgs/appearance_network.py:54-60) and `transformed_image = mapping_image * image` (:15):
 *   u = bilinear(feat -> H x W, align_corners=True); a = relu(conv2(u)); mapping = sigmoid(conv3(a)); transformed = mapping * image
 * feat = output of `up4` [16][feat_h][feat_w]; conv2 is Conv2d(16, 16, 3, padding=1) (weight [16][16][3][3], bias [16]), conv3
 * Conv2d(16, 3, 3, padding=1) (weight [3][16][3][3], bias [3]); image, mapping, transformed are [3][H][W].  Everything is
 * float32 on the device.  The forward keeps `a` in `saved` (gsr_appearance_saved_bytes) for the backward, which returns the
 * gradients of feat, the four parameter tensors and the image -- for the image the direct `mapping * dL_dtransformed` term;
 * the path through the network's down-sampled input is upstream of feat and stays with the caller's autograd.
 * dL_dmapping may be NULL (train.py uses only the transformed image).  mask_relu_of_feat != 0 zeroes dL_dfeat wherever
 * feat <= 0, i.e. returns the gradient of the pre-activation of the ReLU that produced feat (what gsr_appearance_up_backward
 * takes), saving the separate masking pass. */
size_t gsr_appearance_saved_bytes(int H, int W);
size_t gsr_appearance_scratch_bytes(int H, int W);
int gsr_appearance_tail_forward(int H, int W, int feat_h, int feat_w, const float* feat, const float* conv2_weight,
                                const float* conv2_bias, const float* conv3_weight, const float* conv3_bias, const float* image,
                                float* mapping, float* transformed, void* saved, size_t saved_bytes, void* stream);
int gsr_appearance_tail_backward(int H, int W, int feat_h, int feat_w, const float* feat, const float* conv2_weight,
                                 const float* conv3_weight, const float* image, const float* mapping, const void* saved,
                                 size_t saved_bytes, const float* dL_dtransformed, const float* dL_dmapping, float* dL_dimage,
                                 float* dL_dfeat, float* dL_dconv2_weight, float* dL_dconv2_bias, float* dL_dconv3_weight,
                                 float* dL_dconv3_bias, void* scratch, size_t scratch_bytes, int mask_relu_of_feat, void* stream);

/* The block that produces feat: `up4` of the same network, UpsampleBlock(32, 16) = PixelShuffle(2) -> Conv2d(8, 16, 3, padding=1)
 * -> ReLU (scene/ptgs/appearance_network.py:20-29,45).  x is [32][feat_h/2][feat_w/2], weight [16][8][3][3], bias [16], feat
 * [16][feat_h][feat_w] (feat_h, feat_w even).  The backward takes dz = dL/dfeat already masked by feat > 0
 * (gsr_appearance_tail_backward with mask_relu_of_feat) and returns the gradients of x, weight and bias. */
size_t gsr_appearance_up_scratch_bytes(int feat_h, int feat_w);
int gsr_appearance_up_forward(int feat_h, int feat_w, const float* x, const float* weight, const float* bias, float* feat, void* stream);
int gsr_appearance_up_backward(int feat_h, int feat_w, const float* x, const float* weight, const float* dz, float* dL_dx,
                               float* dL_dweight, float* dL_dbias, void* scratch, size_t scratch_bytes, void* stream);

/* The `+ bias` and ReLU that follow each Conv2d of the appearance network's head (scene/ptgs/appearance_network.py:20-29,48), and
 * their backward together with the bias gradient -- four torch kernels per layer otherwise.  x / y / dL_dy / dL_dx are [C][n]
 * (n = H * W): forward y = max(x + bias[c], 0) IN PLACE on x; backward dL_dx = [y > 0] dL_dy, dL_dbias[c] = sum_n dL_dx[c][n]
 * (fixed order, in double). */
size_t gsr_bias_relu_scratch_bytes(int C);
int gsr_bias_relu_forward(int C, int64_t n, float* x, const float* bias, void* stream);
int gsr_bias_relu_backward(int C, int64_t n, const float* dL_dy, const float* y, float* dL_dx, float* dL_dbias, void* scratch,
                           size_t scratch_bytes, void* stream);

/* Evaluation metric of train.py:227 / metrics.py:93: mse of utils/image_utils.py:17-18,
 * out[r] = mean over n elements of (img1[r][i] - img2[r][i])^2 for r < rows; psnr = 20 log10(1 / sqrt(mse)) (:20-22) is left to
 * the caller.  Sums are taken in double in a fixed order. */
size_t gsr_mse_scratch_bytes(int rows);
int gsr_mse(int rows, int64_t n, const float* img1, const float* img2, float* out, void* scratch, size_t scratch_bytes, void* stream);

/* Optimizer step of the training loop: torch.optim.Adam(l, lr=0.0, eps=1e-15).step() of
 * scene/gaussian_model.py:164-175 / train.py:155-157 for ALL parameter tensors in one launch -- dense Adam, weight_decay = 0,
 * amsgrad = False, in the arithmetic of torch's own kernels (csrc/adam.hip).  `tensors` is a HOST array; param / grad /
 * exp_avg / exp_avg_sq are device pointers to n contiguous floats; lr is the tensor's param-group learning rate and `step`
 * its step count AFTER this update (torch keeps one per parameter).  Parameters without a gradient are simply not listed. */
typedef struct {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t n; double lr; int64_t step;
} gsr_adam_tensor;
int gsr_adam_step(int n_tensors, const gsr_adam_tensor* tensors, double beta1, double beta2, double eps, void* stream);

/* Per-iteration densification statistics (train.py:140-142 + GaussianModel.add_densification_stats,
 * scene/gaussian_model.py:433-435) in one pass, with vis = radii > 0:
 *   max_radii2D[vis] = max(max_radii2D[vis], radii[vis]); xyz_gradient_accum[vis] += ||viewspace_grad[vis, :2]||; denom[vis] += 1.
 * viewspace_grad is the [P][3] gradient of the screen-space points (gsr_backward's dL_dmeans2D). */
int gsr_densification_stats(int P, const int* radii, const float* viewspace_grad, float* max_radii2D, float* xyz_gradient_accum,
                            float* denom, void* stream);

/* Order-preserving compaction of the rows of many [P][row_floats] float tensors by one byte mask (keep != 0): the boolean
 * indexing `t[mask]` that prune_points / _prune_optimizer (scene/gaussian_model.py:297-331) apply to 21 tensors one by one.
 * gsr_compact_plan scans the mask and returns the number of kept rows (it waits for it: the caller sizes the outputs);
 * gsr_compact_rows then copies the kept rows of every listed tensor (host array of descriptors, device pointers). */
typedef struct { const float* src; float* dst; int64_t row_floats; } gsr_compact_tensor;
size_t gsr_compact_scratch_bytes(int P);
int gsr_compact_plan(int P, const uint8_t* keep_mask, void* scratch, size_t scratch_bytes, int64_t* count_host, void* stream);
int gsr_compact_rows(int P, const uint8_t* keep_mask, const void* scratch, size_t scratch_bytes, int n_tensors,
                     const gsr_compact_tensor* tensors, void* stream);

/* Test/inspection hook: copies the intermediates held in the opaque buffers into plain arrays (any output may
 * be NULL).  means2D[P][2], depths[P], conic_opacity[P][4], rgb[P][3], cov3D[P][6], clamped[P][3],
 * tiles_touched[P], point_offsets[P], keys_sorted[R], point_list[R], ranges[T][2], final_T[H*W], n_contrib[H*W]. */
int gsr_debug_export(int P, int64_t R, int W, int H, const void* geom_buffer, const void* binning_buffer,
                     const void* image_buffer, float* means2D, float* depths, float* conic_opacity, float* rgb, float* cov3D,
                     uint8_t* clamped, uint32_t* tiles_touched, uint32_t* point_offsets, uint64_t* keys_sorted,
                     uint32_t* point_list, uint32_t* ranges, float* final_T, uint32_t* n_contrib, void* stream);

/* Test hook for the two-stage gradient check (tests/helpers.py): while `sums` is non-NULL every gsr_backward / gsr_backward_params
 * call of this process also stores, per visible Gaussian, the nine blend-level sums its per-Gaussian chain consumes -- rows of 12
 * floats [P][12]: dL_dmean2D.xy, dL_dconic.xyw, dL_dopacity, dL_dcolor.rgb (the reference keeps them in dL_dmean2D / dL_dconic /
 * dL_dopacity / dL_dcolor, rasterize_points.cu:151-159); rows of culled Gaussians are not written.  NULL switches it off. */
int gsr_debug_capture_sums(float* sums);

/* Measurement hook (bench.py's roofline_knn): while `counter` (a device uint64, zeroed by the caller) is non-NULL, gsr_knn_mean_dist2
 * adds to it the number of candidate distances its queries evaluate.  NULL switches it off. */
int gsr_debug_knn_counter(uint64_t* counter);
/* wave-cycle sums per phase of the blend kernels since the last call: [0..7] backward log walk, [8..15] forward quadrant walk (a library built
 * with -DGSR_LOG_PROFILE=1; else zeros): returns 1 / 0 */
int gsr_debug_blend_profile(uint64_t* counters16);

/* What this binary was built with: "abi=<n> ablate=<GSR_ABLATE> bin_ablate=<GSR_BIN_ABLATE> extra=[<make EXTRA=...>]".  The two
 * ablation macros are timing experiments that produce wrong results on purpose; a shipped library reports 0 for both and an
 * empty extra (tests/test_abi.py checks the built .so; bench.py prints the string). */
const char* gsr_build_info(void);

/* Overrides an experiment switch (environment GSR_<NAME>) for this process: A/B measurements and the tests
 * that keep the alternative kernels covered.  No switch changes a result beyond the documented tolerances.
 * Returns 0, or -1 when the switch cannot be stored (NULL / name of 24+ characters / table of 64 switches full). */
int gsr_set_variant(const char* name, int value);
int gsr_get_variant(const char* name, int default_value);

/* Stand-alone stable LSD radix sort of (u64,u32) pairs on bits [0,end_bit) (the binning sort, exported for
 * tests).  Returns 0 if the result is in (keys_a, vals_a), 1 if in (keys_b, vals_b), negative on error.
 * scratch must hold gsr_sort_scratch_bytes(n) bytes. */
size_t gsr_sort_scratch_bytes(int64_t n);
int gsr_sort_pairs(uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, int64_t n, int end_bit,
                   void* scratch, size_t scratch_bytes, void* stream);

/* Optional per-stage timing (bench.py): when enabled, every stage launched through this library is bracketed by
 * HIP events on the caller's stream; gsr_profile_collect() waits for them, sums elapsed ms and launch counts per
 * stage (arrays of nstages >= GSR_NUM_STAGES) and resets the log. */
enum {
    GSR_STAGE_PREPROCESS = 0, GSR_STAGE_SCAN = 1, GSR_STAGE_EMIT = 2, GSR_STAGE_SORT = 3, GSR_STAGE_RANGES = 4,
    GSR_STAGE_RENDER_FWD = 5, GSR_STAGE_RENDER_BWD = 6, GSR_STAGE_GAUSS_BWD = 7, GSR_NUM_STAGES = 8
};
void gsr_profile_enable(int on);
/* Restricts the events to the stages whose bit is set (bit i = stage i; default all).  Every pair of events costs a few
 * microseconds of stream time, so a timed run that only needs the dominant stage selects that one. */
void gsr_profile_select(uint32_t stage_mask);
int gsr_profile_collect(double* ms_per_stage, int64_t* launches_per_stage, int nstages);

#ifdef __cplusplus
}
#endif
#endif
