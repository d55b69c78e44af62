// extern "C" entry points of libgsrast_mi355x.so (declared in include/gsrast_mi355x.h).
// Host-side orchestration only: buffer layouts, argument checks, stage launches on the caller's stream.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/gsrast_mi355x.h"
#include "common.h"

namespace {

thread_local char g_err[512] = "";
float* g_capture_sums = nullptr;  // gsr_debug_capture_sums
unsigned long long* g_knn_evals = nullptr;  // gsr_debug_knn_counter

int fail(int code, const char* fmt, const char* a = "")
{
    snprintf(g_err, sizeof(g_err), fmt, a);
    return code;
}

int check_hip(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return -100;
    }
    return 0;
}

// ---- optional per-stage timing with HIP events on the caller's stream (bench.py's roofline figures) ----
struct ProfRec { int stage; hipEvent_t a, b; bool count; };
bool g_prof = false;
uint32_t g_prof_mask = 0xFFFFFFFFu;  // stages that get events while g_prof is set
std::vector<ProfRec> g_recs;
std::mutex g_prof_mu;

struct StageTimer {
    int stage; hipStream_t s; hipEvent_t a, b; bool on, count;  // count = false: a second segment of a stage already counted
    StageTimer(int stage_, hipStream_t s_, bool count_ = true)
        : stage(stage_), s(s_), on(g_prof && ((g_prof_mask >> stage_) & 1u)), count(count_)
    {
        if (on) {
            // timing-only events: without the system-scope fence a default event carries (a write-back and invalidation of the caches
            // when it completes -- paid by the kernel behind it).  GSR_EVENT_FENCE=1 brings the default events back (A/B).
            const unsigned flags = gsr_variant("EVENT_FENCE", 0) ? hipEventDefault : hipEventDisableSystemFence;
            hipEventCreateWithFlags(&a, flags);
            hipEventCreateWithFlags(&b, flags);
            hipEventRecord(a, s);
        }
    }
    ~StageTimer()
    {
        if (on) {
            hipEventRecord(b, s);
            std::lock_guard<std::mutex> lk(g_prof_mu);
            g_recs.push_back({stage, a, b, count});
        }
    }
};

// ---- host mailbox for num_rendered -------------------------------------------------------------------------------
// The forward needs R on the host to size the binning buffer (the reference synchronises at the same point,
// rasterizer_impl.cu:281).  Instead of hipMemcpyAsync + hipStreamSynchronize (a copy command plus a blocking wait,
// ~50 us of GPU idle time per step) the prefix-sum kernel publishes R, tagged with a sequence number, in host-mapped
// pinned memory and the host polls that word.
constexpr int kMailSlots = 64;
// slots kMailSlots, kMailSlots + 1 of the pinned words: the long-list hint of the tile binning (binning.hip, mid_hint_publish)
struct Mailbox { unsigned long long* host = nullptr; unsigned long long* dev = nullptr; };
constexpr int kMailDevices = 64;  // more devices than that: the forward falls back to memcpy + stream sync
Mailbox g_mail[kMailDevices];
std::mutex g_mail_mu;
uint32_t g_mail_seq = 0;

bool mailbox_acquire(unsigned long long** host_slot, unsigned long long** dev_slot, uint32_t* seq)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMailDevices) return false;
    std::lock_guard<std::mutex> lk(g_mail_mu);
    Mailbox& m = g_mail[dev];
    if (!m.host) {
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, (kMailSlots + 1) * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return false; }
        memset(h, 0, (kMailSlots + 1) * sizeof(unsigned long long));
        m.host = (unsigned long long*)h;
        m.dev = (unsigned long long*)d;
    }
    const uint32_t s = ++g_mail_seq;
    *host_slot = m.host + (s % kMailSlots);
    *dev_slot = m.dev + (s % kMailSlots);
    *seq = s;
    return true;
}

// The long-list hint of this device (two 32-bit words behind the mailbox slots) and this call's frame number (counted per device: the
// words are per device).  launch_mid = false only when the hint is there, a frame of THIS problem shape (P, W, H) has reported, none of the
// last 2^20 frames had a list of more than 2048 entries (a scene that shows such a list once keeps the kernel for good: the in-kernel
// fallback costs milliseconds, the empty launch ~2 us) and R has not grown by more than a quarter since the kernel was first skipped
// (densification, a closer camera: longer lists are likely, and the frames the host has already enqueued would all take the fallback).
struct MidState { uint32_t seq = 0, since = 0, R_ref = 0; int P = -1, W = -1, H = -1; };
MidState g_mid[kMailDevices];
bool mid_hint_acquire(int P, int W, int H, uint32_t R, uint32_t** hint_dev, uint32_t* seq, bool* launch_mid)
{
    *hint_dev = nullptr; *seq = 0; *launch_mid = true;
    int dev = 0;
    const int mode = gsr_variant("MID_HINT", 1);  // 0: always launch the kernel, 1: by the hint, 2: never (tests of the in-kernel fallback)
    if (mode == 2) *launch_mid = false;
    if (mode != 1 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMailDevices) return false;
    std::lock_guard<std::mutex> lk(g_mail_mu);
    const Mailbox& m = g_mail[dev];
    if (!m.host) return false;  // (created by the geometry phase's mailbox_acquire)
    MidState& st = g_mid[dev];
    const volatile uint32_t* h = reinterpret_cast<const volatile uint32_t*>(m.host + kMailSlots);
    const uint32_t s = ++st.seq;
    if (P != st.P || W != st.W || H != st.H) { st.P = P; st.W = W; st.H = H; st.since = s; st.R_ref = 0; }
    const uint32_t done = h[0], mid = h[1];
    *hint_dev = reinterpret_cast<uint32_t*>(m.dev + kMailSlots);
    *seq = s;
    bool launch = done == 0u || done - st.since > (1u << 31) /* done < since: no frame of this shape has reported */ ||
                  (mid != 0u && s - mid <= (1u << 20));
    if (!launch) {
        if (st.R_ref == 0u) st.R_ref = R;
        if ((unsigned long long)R * 4ull > (unsigned long long)st.R_ref * 5ull) { launch = true; st.since = s; st.R_ref = 0; }
    }
    *launch_mid = launch;
    return true;
}

// Waits for the tagged value.  Every 2^17 spins (~a millisecond) the stream is queried: if it has drained (hipSuccess) without the word
// showing up, or reports an error (a faulted kernel), the wait ends at once and the caller falls back to memcpy + stream
// synchronisation, which surfaces the error -- a dead kernel stalls the host for a millisecond, not for seconds.
bool mailbox_wait(const unsigned long long* host_slot, uint32_t seq, uint32_t* value, hipStream_t s, hipError_t* stream_error)
{
    *stream_error = hipSuccess;
    const volatile unsigned long long* p = host_slot;
    for (long spin = 0; spin < 400000000L; spin++) {
        const unsigned long long v = *p;
        if ((uint32_t)(v >> 32) == seq) { *value = (uint32_t)v; return true; }
        __builtin_ia32_pause();
        if ((spin & 0x1FFFF) == 0x1FFFF) {
            const hipError_t q = hipStreamQuery(s);
            if (q != hipErrorNotReady) {  // drained or failed: one last look at the word, then give up
                const unsigned long long v2 = *p;
                if (q == hipSuccess && (uint32_t)(v2 >> 32) == seq) { *value = (uint32_t)v2; return true; }
                if (q != hipSuccess) *stream_error = q;  // handed to the caller: a failed launch must not end as a stale R with status 0
                return false;
            }
        }
    }
    return false;
}

// Sorted 64-bit keys of the tile-binning path, rebuilt on demand for inspection: (tile << 32) | depth bits of each list entry
// (the production path keeps only point_list + ranges; the reference's keys are a function of those and the depths).
__global__ void __launch_bounds__(256) export_keys_kernel(const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                                                          const float* __restrict__ splat, unsigned long long* __restrict__ keys)
{
    const uint32_t tile = blockIdx.x;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    for (uint32_t i = r0 + threadIdx.x; i < r1; i += 256)
        keys[i] = ((unsigned long long)tile << 32) | __float_as_uint(splat[(size_t)point_list[i] * GSR_SPLAT_F + SP_DEPTH]);
}

__global__ void __launch_bounds__(256) export_geom_kernel(int P, const float* __restrict__ splat, float* means2D, float* depths,
                                                          float* conic_opacity, float* rgb)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float* r = splat + (size_t)i * GSR_SPLAT_F;
    if (means2D) { means2D[2 * i] = r[SP_MX]; means2D[2 * i + 1] = r[SP_MY]; }
    if (depths) depths[i] = r[SP_DEPTH];
    if (conic_opacity) {
        conic_opacity[4 * i] = r[SP_CX]; conic_opacity[4 * i + 1] = r[SP_CY];
        conic_opacity[4 * i + 2] = r[SP_CZ]; conic_opacity[4 * i + 3] = r[SP_OPA];
    }
    if (rgb) { rgb[3 * i] = r[SP_R]; rgb[3 * i + 1] = r[SP_G]; rgb[3 * i + 2] = r[SP_B]; }
}

}  // namespace

// Experiment switches: GSR_<NAME> is read from the environment ONCE per name (first use) and cached, so launches from
// several host threads never race on getenv and a switch cannot change between the stages of one step.  gsr_set_variant
// (tests, A/B runs inside one process) overrides an entry.
namespace {
struct VariantEntry { char name[24]; bool set; int value; };
VariantEntry g_variants[64];
int g_n_variants = 0;
std::mutex g_variant_mu;
}  // namespace

int gsr_variant(const char* name, int dflt)
{
    std::lock_guard<std::mutex> lk(g_variant_mu);
    for (int i = 0; i < g_n_variants; i++)
        if (strcmp(g_variants[i].name, name) == 0) return g_variants[i].set ? g_variants[i].value : dflt;
    char key[64];
    snprintf(key, sizeof(key), "GSR_%s", name);
    const char* v = getenv(key);
    if (g_n_variants < 64 && strlen(name) < sizeof(g_variants[0].name)) {
        VariantEntry& e = g_variants[g_n_variants++];
        strcpy(e.name, name);
        e.set = v != nullptr;
        e.value = v ? atoi(v) : 0;
    }
    return v ? atoi(v) : dflt;
}

static bool gsr_variant_store(const char* name, int value)
{
    std::lock_guard<std::mutex> lk(g_variant_mu);
    for (int i = 0; i < g_n_variants; i++)
        if (strcmp(g_variants[i].name, name) == 0) { g_variants[i].set = true; g_variants[i].value = value; return true; }
    return false;
}

// Tile lists without the (Gaussian, tile) instances the splat cannot reach (GSR_CULL, default on; needs the tile binning).  The
// reference lists every tile of the square around the 3-sigma circle of the larger axis (auxiliary.h:46-56): on config 2, 40 % of
// those instances lie outside the alpha >= 1/255 ellipse (gsr_reach_mask) and cannot contribute to any pixel.  Leaving them out
// changes no pixel and no gradient; GSR_CULL=0 keeps the reference's lists bit for bit (sort keys, point_list, ranges, n_contrib).
bool gsr_cull_enabled() { return gsr_variant("CULL", 1) && gsr_variant("BIN", 1); }
// The blend log (render_forward.hip) is written by the quadrant walk of the forward only.  GSR_BLEND_LOG: 0 never, 1 (default) / 2 always;
// GSR_BLEND_LOG_LIST = n > 0 restricts it to frames whose mean reference tile list is shorter than n (0, the default: no restriction).
// Measured with the final kernels of round 5 (profiles/README.md): the log pays at every density -- 500k Gaussians at 1080p (mean list 467)
// 0.689 -> 0.649 ms per step, 10k at 256x256 0.160 -> 0.139, 1 M (934) kernels 0.545 -> 0.538, 1.5 M (1,400) 0.647 -> 0.636, 2 M (1,866)
// 0.812 -> 0.793 (the forward pays ~0.012-0.02 ms for writing it, the backward gains 0.02-0.04).  (An earlier build of the round wrote it
// only below a mean list of 700: its four-waves-per-tile backward gained nothing then.)  Forward and backward of one frame get the same
// answer because both ask with the frame's own R and image size; the switches are read at launch time, so change them between frames,
// not between a forward and its backward.
bool gsr_blend_log_for(int64_t R, int W, int H)
{
    const int mode = gsr_variant("BLEND_LOG", 1);
    if (mode == 0 || R <= 0 || gsr_variant("QUAD", 1) != 1 || gsr_variant("WAVE8X8", 1) == 0) return false;
    const int limit = gsr_variant("BLEND_LOG_LIST", 0);
    if (limit <= 0) return true;
    const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    return (float)R / (float)(gx * gy) < (float)limit;
}

extern "C" {

int gsr_abi_version(void) { return GSR_ABI_VERSION; }

int gsr_get_variant(const char* name, int default_value) { return name ? gsr_variant(name, default_value) : default_value; }

int gsr_set_variant(const char* name, int value)
{
    if (!name) return fail(-1, "gsr_set_variant: NULL name");
    gsr_variant(name, 0);  // make sure the entry exists
    if (!gsr_variant_store(name, value)) return fail(-1, "gsr_set_variant: switch %s cannot be stored (name too long or table full)", name);
    return 0;
}
const char* gsr_last_error(void) { return g_err; }

size_t gsr_geometry_bytes(int P) { return gsr_geom_layout(P).bytes; }
size_t gsr_image_bytes(int W, int H) { return gsr_image_layout(W, H).bytes; }
size_t gsr_binning_bytes(int64_t R, int P, int W, int H) { return gsr_bin_layout((long)R, P, W, H).bytes; }
// Backward scratch: one gradient row (GSR_GROW_F = 9 floats) per (Gaussian, tile) instance at most (only reachable instances own one).
size_t gsr_backward_scratch_bytes(int64_t R, int W, int H)
{
    (void)W; (void)H;
    return gsr_align((size_t)(R > 0 ? R : 1) * GSR_GROW_F * 4);
}

// The blend backward keeps a gradient-row index in 28 bits beside the 4-bit quadrant mask.  Rows <= instances (only reachable instances
// own one: ~60 % of R on the benchmark scene), so R below the limit needs no look at the device; above it the row count the forward's
// offset kernels left in the geometry buffer decides (one 4-byte copy + stream sync, only in that case).
int64_t gsr_gradient_row_limit(void) { return (int64_t)gsr_variant("ROW_LIMIT", 0x0FFFFFFF); }  // (lowered by tests only)
int gsr_gradient_rows(int P, const void* geom_buffer, size_t geom_bytes, int64_t* rows_host, void* stream)
{
    if (!rows_host) return fail(-1, "rows_host is NULL");
    *rows_host = 0;
    if (P <= 0) return 0;
    const GeomLayout G = gsr_geom_layout(P);
    if (!geom_buffer || geom_bytes < G.bytes) return fail(-3, "geometry buffer too small");
    uint32_t rows = 0;
    const uint32_t* src = (const uint32_t*)((const char*)geom_buffer + G.row_block_offsets) + (P + 255) / 256;
    hipError_t e = hipMemcpyAsync(&rows, src, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail(-4, "gsr_gradient_rows: %s", hipGetErrorString(e));
    *rows_host = (int64_t)rows;
    return 0;
}

// Phase 1 shared by gsr_forward_geometry, gsr_forward and gsr_forward_params: `a` carries the inputs, the outputs inside the
// geometry buffer are filled in here.  launch_geometry enqueues the two kernels, wait_geometry returns R.
struct GeometryWait {
    bool mail = false; unsigned long long* mail_host = nullptr; uint32_t seq = 0; char* total_dev = nullptr;
    unsigned long long* mail_dev = nullptr;  // handed to emit_partition_kernel when no scan kernel was launched (with_scan = false)
    const uint32_t* trap_dev = nullptr; uint32_t trap_tag = 0;  // prefiltered: the word a culled point stamps with this call's tag
};
std::atomic<uint32_t> g_trap_tag{0};

// with_scan = false: the caller enqueues the banded head right behind this (launch_render_head), whose first kernel scans the block sums
// itself and publishes R -- no scan_blocks launch (one ~5 us kernel less on the critical path)
static int launch_geometry(PreprocessArgs a, void* geom_buffer, size_t geom_bytes, GeometryWait* w, hipStream_t s, bool with_scan = true)
{
    const int P = a.P;
    const GeomLayout L = gsr_geom_layout(P);
    if (geom_bytes < L.bytes) return fail(-3, "geometry buffer too small");
    char* g = (char*)geom_buffer;
    a.splat = (float*)(g + L.splat); a.cov3D = (float*)(g + L.cov3D); a.clamped = (uint8_t*)(g + L.clamped);
    a.tiles_touched = (uint32_t*)(g + L.tiles_touched); a.block_sums = (uint32_t*)(g + L.block_sums);
    a.row_block_sums = (uint32_t*)(g + L.row_block_sums);
    a.zero64 = (uint32_t*)(g + L.band_count);
    if (a.prefiltered) {  // the third of the four words at L.total; tags are non-zero and differ from call to call: no zeroing needed
        uint32_t tag = ++g_trap_tag;
        if (tag == 0) tag = ++g_trap_tag;
        a.trap_flag = (uint32_t*)(g + L.total) + 2; a.trap_tag = tag;
        w->trap_dev = a.trap_flag; w->trap_tag = tag;
    }
    { StageTimer t(GSR_STAGE_PREPROCESS, s); gsr_launch_preprocess(a, s); }
    const int nb = (P + 255) / 256;
    unsigned long long* mail_dev = nullptr;
    w->mail = gsr_variant("MAILBOX", 1) && mailbox_acquire(&w->mail_host, &mail_dev, &w->seq);
    w->mail_dev = w->mail ? mail_dev : nullptr;
    w->total_dev = g + L.total;
    if (with_scan) {
        StageTimer t(GSR_STAGE_SCAN, s);
        gsr_launch_scan_blocks2(nb, a.block_sums, (uint32_t*)(g + L.block_offsets), (uint32_t*)(g + L.total), a.row_block_sums,
                                (uint32_t*)(g + L.row_block_offsets), w->mail ? mail_dev : nullptr, w->seq, (uint32_t*)(g + L.band_count), s);
    }
    return 0;
}

static int wait_geometry(const GeometryWait& w, int64_t* num_rendered_host, hipStream_t s)
{
    uint32_t total = 0;
    hipError_t qe = hipSuccess;
    if (!w.mail || !mailbox_wait(w.mail_host, w.seq, &total, s, &qe)) {
        if (qe != hipSuccess) { (void)hipGetLastError(); return fail(-100, "forward_geometry: %s", hipGetErrorString(qe)); }
        hipError_t e = hipMemcpyAsync(&total, w.total_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(-100, "forward_geometry: %s", hipGetErrorString(e));
    }
    *num_rendered_host = (int64_t)total;
    if (w.trap_dev) {
        // prefiltered = true: the reference's kernel prints this message and traps when the frustum test rejects a point
        // (auxiliary.h:154-160), which ends the process's CUDA context; here the call fails with the same words and the context lives
        uint32_t stamp = 0;
        hipError_t e = hipMemcpyAsync(&stamp, w.trap_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(-100, "forward_geometry: %s", hipGetErrorString(e));
        if (stamp == w.trap_tag) return fail(-6, "Point is filtered although prefiltered is set. This shouldn't happen!");
    }
    return check_hip("forward_geometry");
}

static int run_geometry(PreprocessArgs a, void* geom_buffer, size_t geom_bytes, int64_t* num_rendered_host, hipStream_t s)
{
    GeometryWait w;
    const int rc = launch_geometry(a, geom_buffer, geom_bytes, &w, s);
    return rc != 0 ? rc : wait_geometry(w, num_rendered_host, s);
}

// argument checks of RasterizeGaussiansCUDA (rasterize_points.cu:57-113) + the launch descriptor; 1 = nothing to do (P == 0)
static int make_std_args(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier, const float* rotations,
                         const float* opacities, const float* shs, const float* cov3D_precomp, const float* colors_precomp,
                         const float* viewmatrix, const float* projmatrix, const float* campos, int W, int H, float tan_fovx,
                         float tan_fovy, int* radii, void* geom_buffer, int64_t* num_rendered_host, PreprocessArgs* out)
{
    if (!num_rendered_host) return fail(-1, "num_rendered_host is NULL");
    *num_rendered_host = 0;
    if (P < 0 || W <= 0 || H <= 0) return fail(-1, "invalid sizes");
    if (P == 0) return 1;
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !campos || !radii || !geom_buffer)
        return fail(-1, "NULL required pointer");
    if ((shs == nullptr) == (colors_precomp == nullptr)) return fail(-2, "provide exactly one of shs / colors_precomp");
    if (cov3D_precomp == nullptr && (scales == nullptr || rotations == nullptr))
        return fail(-2, "provide scales+rotations or cov3D_precomp");
    if (shs && (D < 0 || D > 3 || M < (D + 1) * (D + 1))) return fail(-2, "SH degree / coefficient count mismatch");
    PreprocessArgs& a = *out;
    a.P = P; a.D = D; a.M = M; a.W = W; a.H = H;
    a.means3D = means3D; a.scales = scales; a.rotations = rotations; a.opacities = opacities; a.shs = shs;
    a.cov3D_precomp = cov3D_precomp; a.colors_precomp = colors_precomp;
    a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; a.campos = campos;
    a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
    a.radii = radii;
    a.raw = 0; a.features_dc = nullptr; a.features_rest = nullptr;
    return 0;
}

int gsr_forward_geometry(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                         const float* rotations, const float* opacities, const float* shs, const float* cov3D_precomp,
                         const float* colors_precomp, const float* viewmatrix, const float* projmatrix, const float* campos,
                         int W, int H, float tan_fovx, float tan_fovy, int prefiltered, int* radii, void* geom_buffer,
                         size_t geom_bytes, int64_t* num_rendered_host, void* stream)
{
    PreprocessArgs a;
    const int rc = make_std_args(P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp, colors_precomp,
                                 viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, radii, geom_buffer, num_rendered_host, &a);
    if (rc != 0) return rc < 0 ? rc : 0;
    a.prefiltered = prefiltered != 0;
    return run_geometry(a, geom_buffer, geom_bytes, num_rendered_host, (hipStream_t)stream);
}

// The stages of phase 2 that need neither R nor the R-sized arrays: instance offsets (emit) and the head of the tile binning.
// gsr_forward / gsr_forward_params enqueue them BEFORE waiting for R, so the host round trip overlaps with GPU work.
static void launch_render_head(int P, int W, int H, void* geom_buffer, void* binning_buffer, hipStream_t s,
                               unsigned long long* mailbox = nullptr, uint32_t seq = 0)
{
    const GeomLayout G = gsr_geom_layout(P);
    const BinLayout B = gsr_bin_layout(0, P, W, H);  // the offsets used here do not depend on R
    char* g = (char*)geom_buffer;
    char* b = (char*)binning_buffer;
    {   // per-Gaussian instance offsets (needed by the backward's gradient-row addressing), with the band partition when it applies
        StageTimer t(GSR_STAGE_EMIT, s);
        if (!gsr_launch_emit_partition(P, W, H, (float*)(g + G.splat), (const uint32_t*)(g + G.tiles_touched),
                                       (const uint32_t*)(g + G.block_sums), (const uint32_t*)(g + G.row_block_sums),
                                       (uint32_t*)(g + G.total), (uint32_t*)(g + G.row_block_offsets) + (P + 255) / 256, mailbox, seq,
                                       (uint32_t*)(g + G.point_offsets), (uint32_t*)(g + G.row_offsets),
                                       (uint32_t*)(g + G.wg_first), (uint32_t*)(g + G.band_count), (uint32_t*)(b + B.band), s))
            gsr_launch_emit(P, W, H, (float*)(g + G.splat), (const uint32_t*)(g + G.tiles_touched), (const uint32_t*)(g + G.block_offsets),
                            (uint32_t*)(g + G.point_offsets), (const uint32_t*)(g + G.row_block_offsets), (uint32_t*)(g + G.row_offsets),
                            (uint32_t*)(g + G.wg_first), nullptr, nullptr, s);
    }
    StageTimer t(GSR_STAGE_SORT, s);
    gsr_launch_tile_binning_head(P, W, H, (const float*)(g + G.splat), (uint32_t*)(b + B.table), (uint32_t*)(b + B.totals),
                                 (const uint32_t*)(g + G.band_count), (uint32_t*)(b + B.band), s);
}

static bool can_launch_head_early(int P, int W, int H, const void* binning_buffer, size_t binning_bytes)
{
    return P > 0 && gsr_variant("BIN", 1) && gsr_variant("EARLY", 1) && binning_buffer &&
           binning_bytes >= gsr_bin_layout(0, P, W, H).fixed_bytes;
}

static int forward_render_impl(int P, int64_t R, int W, int H, const float* background, void* geom_buffer, size_t geom_bytes,
                               void* binning_buffer, size_t binning_bytes, void* image_buffer, size_t image_bytes, float* out_color,
                               bool head_done, hipStream_t s)
{
    if (P == 0) return 0;
    if (!background || !geom_buffer || !image_buffer || !out_color) return fail(-1, "NULL required pointer");
    if (R < 0 || R > 0xFFFFFFFFll) return fail(-1, "num_rendered out of range");
    const GeomLayout G = gsr_geom_layout(P);
    const ImageLayout I = gsr_image_layout(W, H);
    const BinLayout B = gsr_bin_layout((long)R, P, W, H);
    if (geom_bytes < G.bytes) return fail(-3, "geometry buffer too small");
    if (image_bytes < I.bytes) return fail(-3, "image buffer too small");
    if (R > 0 && (!binning_buffer || binning_bytes < B.bytes)) return fail(-3, "binning buffer too small");
    char* g = (char*)geom_buffer;
    char* b = (char*)binning_buffer;
    char* im = (char*)image_buffer;
    const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    uint32_t* ranges = (uint32_t*)(im + I.ranges);
    const uint32_t* point_list = nullptr;
    if (R > 0) {
        uint64_t* ka = (uint64_t*)(b + B.keys_a);
        uint64_t* kb = (uint64_t*)(b + B.keys_b);
        uint32_t* va = (uint32_t*)(b + B.vals_a);
        uint32_t* vb = (uint32_t*)(b + B.vals_b);
        if (gsr_variant("BIN", 1)) {
            if (!head_done) {
                // the scan kernel zeroed the band counters for ONE head; this one may be a second (the early head ran, then the
                // caller had to re-allocate the binning buffer and came back through gsr_forward_render)
                hipMemsetAsync(g + G.band_count, 0, GSR_BAND_COUNTERS * 4, s);
                launch_render_head(P, W, H, geom_buffer, binning_buffer, s);
            }
            uint32_t* hint = nullptr;
            uint32_t hint_seq = 0;
            bool launch_mid = true;
            mid_hint_acquire(P, W, H, (uint32_t)R, &hint, &hint_seq, &launch_mid);
            StageTimer t(GSR_STAGE_SORT, s, false);
            gsr_launch_tile_binning_tail(P, W, H, (const float*)(g + G.splat), (uint32_t*)(b + B.table), (uint32_t*)(b + B.totals),
                                         (uint32_t*)(b + B.order), ranges, (const uint32_t*)(g + G.band_count), (uint32_t*)(b + B.band), ka, va,
                                         gsr_variant("KEYS", 0) ? kb : nullptr, kb, hint, hint_seq, launch_mid, s);
            point_list = va;
        } else {
        {
            StageTimer t(GSR_STAGE_EMIT, s);
            gsr_launch_emit(P, W, H, (float*)(g + G.splat), (const uint32_t*)(g + G.tiles_touched),
                            (const uint32_t*)(g + G.block_offsets), (uint32_t*)(g + G.point_offsets),
                            (const uint32_t*)(g + G.row_block_offsets), (uint32_t*)(g + G.row_offsets), (uint32_t*)(g + G.wg_first), ka, va, s);
        }
        const int end_bit = 32 + (int)gsr_higher_msb((uint32_t)(gx * gy));  // rasterizer_impl.cu:300-308
        int in_b;
        {
            StageTimer t(GSR_STAGE_SORT, s);
            in_b = gsr_launch_radix_sort(ka, kb, va, vb, (uint32_t)R, end_bit, (uint32_t*)(b + B.hist_global),
                                         (uint32_t*)(b + B.hist_blocks), B.sort_blocks, s);
        }
        { StageTimer t(GSR_STAGE_RANGES, s); gsr_launch_tile_ranges((uint32_t)R, in_b ? kb : ka, ranges, (uint32_t)(gx * gy), s); }
        point_list = in_b ? vb : va;
        }
    } else {
        // point_offsets must still be valid for the backward pass; no instances -> all ranges empty
        // (an early head, if any, wrote the same zeros)
        hipMemsetAsync(g + G.point_offsets, 0, (size_t)P * 4, s);
        hipMemsetAsync(g + G.row_offsets, 0, (size_t)P * 4, s);
        gsr_launch_tile_ranges(0, nullptr, ranges, (uint32_t)(gx * gy), s);
    }
    {
        StageTimer t(GSR_STAGE_RENDER_FWD, s);
        const uint32_t* order = (R > 0 && gsr_variant("BIN", 1)) ? (const uint32_t*)(b + B.order) : nullptr;
        unsigned long long* blend_log = gsr_blend_log_for(R, W, H) ? (unsigned long long*)(b + B.blend_log) : nullptr;
        gsr_launch_render_forward(W, H, ranges, point_list, (const float*)(g + G.splat), background, (float*)(im + I.final_T),
                                  (uint32_t*)(im + I.n_contrib), out_color, order, blend_log, (uint32_t*)(im + I.quad_hi), s);
    }
    return check_hip("forward_render");
}

int gsr_forward_render(int P, int64_t R, int W, int H, const float* background, void* geom_buffer, size_t geom_bytes,
                       void* binning_buffer, size_t binning_bytes, void* image_buffer, size_t image_bytes,
                       float* out_color, void* stream)
{
    return forward_render_impl(P, R, W, H, background, geom_buffer, geom_bytes, binning_buffer, binning_bytes, image_buffer,
                               image_bytes, out_color, false, (hipStream_t)stream);
}

// phases 1 + 2 with the R round trip hidden behind the R-independent head of phase 2
static int forward_fused(const PreprocessArgs& a, const float* background, void* geom_buffer, size_t geom_bytes, void* binning_buffer,
                         size_t binning_bytes, void* image_buffer, size_t image_bytes, float* out_color, int64_t* num_rendered_host,
                         hipStream_t s)
{
    GeometryWait w;
    const bool early = can_launch_head_early(a.P, a.W, a.H, binning_buffer, binning_bytes);
    // EMIT_SCAN (default): with the banded head enqueued right behind the preprocess kernel, its first kernel does the block scan and
    // publishes R; the separate scan kernel is launched only where R must reach the host without that head
    int nb_, bh_;
    const bool emit_scan = early && gsr_variant("EMIT_SCAN", 1) && gsr_variant("MAILBOX", 1) &&
                           gsr_band_geometry((a.W + GSR_TILE_X - 1) / GSR_TILE_X, (a.H + GSR_TILE_Y - 1) / GSR_TILE_Y, &nb_, &bh_);
    int rc = launch_geometry(a, geom_buffer, geom_bytes, &w, s, !emit_scan);
    if (rc != 0) return rc;
    // (without a mailbox slot wait_geometry copies the total that emit_partition_kernel's first workgroup stores)
    if (early) launch_render_head(a.P, a.W, a.H, geom_buffer, binning_buffer, s, emit_scan ? w.mail_dev : nullptr, w.seq);
    rc = wait_geometry(w, num_rendered_host, s);
    if (rc != 0) return rc;
    const int64_t R = *num_rendered_host;
    if (R > 0 && gsr_bin_layout((long)R, a.P, a.W, a.H).bytes > binning_bytes) return 1;  // caller re-allocates, then gsr_forward_render
    return forward_render_impl(a.P, R, a.W, a.H, background, geom_buffer, geom_bytes, binning_buffer, binning_bytes, image_buffer,
                               image_bytes, out_color, early, s);
}

int gsr_forward(int P, int D, int M, const float* background, const float* means3D, const float* scales, float scale_modifier,
                const float* rotations, const float* opacities, const float* shs, const float* cov3D_precomp,
                const float* colors_precomp, const float* viewmatrix, const float* projmatrix, const float* campos, int W, int H,
                float tan_fovx, float tan_fovy, int prefiltered, int* radii, void* geom_buffer, size_t geom_bytes,
                void* binning_buffer, size_t binning_bytes, void* image_buffer, size_t image_bytes, float* out_color,
                int64_t* num_rendered_host, void* stream)
{
    PreprocessArgs a;
    const int rc = make_std_args(P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, cov3D_precomp, colors_precomp,
                                 viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, radii, geom_buffer, num_rendered_host, &a);
    if (rc != 0) return rc < 0 ? rc : 0;
    a.prefiltered = prefiltered != 0;
    return forward_fused(a, background, geom_buffer, geom_bytes, binning_buffer, binning_bytes, image_buffer, image_bytes, out_color,
                         num_rendered_host, (hipStream_t)stream);
}

int gsr_forward_params(int P, int D, int M, const float* background, const float* xyz, const float* features_dc,
                       const float* features_rest, const float* opacity_raw, const float* scaling_raw, const float* rotation_raw,
                       float scale_modifier, const float* viewmatrix, const float* projmatrix, const float* campos, int W, int H,
                       float tan_fovx, float tan_fovy, int* radii, void* geom_buffer, size_t geom_bytes, void* binning_buffer,
                       size_t binning_bytes, void* image_buffer, size_t image_bytes, float* out_color, int64_t* num_rendered_host,
                       void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!num_rendered_host) return fail(-1, "num_rendered_host is NULL");
    *num_rendered_host = 0;
    if (P < 0 || W <= 0 || H <= 0) return fail(-1, "invalid sizes");
    if (P == 0) return 0;
    if (!xyz || !features_dc || !opacity_raw || !scaling_raw || !rotation_raw || !viewmatrix || !projmatrix || !campos || !radii ||
        !geom_buffer)
        return fail(-1, "NULL required pointer");
    if (D < 0 || D > 3 || M < (D + 1) * (D + 1)) return fail(-2, "SH degree / coefficient count mismatch");
    if (M > 1 && !features_rest) return fail(-1, "features_rest is NULL");
    PreprocessArgs a;
    a.P = P; a.D = D; a.M = M; a.W = W; a.H = H;
    a.means3D = xyz; a.scales = scaling_raw; a.rotations = rotation_raw; a.opacities = opacity_raw; a.shs = nullptr;
    a.cov3D_precomp = nullptr; a.colors_precomp = nullptr;
    a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; a.campos = campos;
    a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
    a.radii = radii;
    a.raw = 1; a.features_dc = features_dc; a.features_rest = features_rest;
    return forward_fused(a, background, geom_buffer, geom_bytes, binning_buffer, binning_bytes, image_buffer, image_bytes, out_color,
                         num_rendered_host, s);
}

// Shared by gsr_backward and gsr_backward_params: `a` carries the per-Gaussian inputs and the output pointers.
static int run_backward(GaussBackwardArgs a, int64_t R, const float* background, const void* geom_buffer, size_t geom_bytes,
                        const void* binning_buffer, size_t binning_bytes, const void* image_buffer, size_t image_bytes,
                        void* scratch, size_t scratch_bytes, const float* dL_dpix, const float* cov3D_precomp, hipStream_t s)
{
    const int P = a.P, W = a.W, H = a.H;
    const GeomLayout G = gsr_geom_layout(P);
    const ImageLayout I = gsr_image_layout(W, H);
    const BinLayout B = gsr_bin_layout((long)R, P, W, H);
    if (geom_bytes < G.bytes || image_bytes < I.bytes) return fail(-3, "state buffer too small");
    // render_backward keeps a gradient-row index in 28 bits beside the 4-bit quadrant mask: the ROW count is what is limited (rows <=
    // instances; the Python operators make the same check at forward time when a gradient will be asked for)
    if (R >= gsr_gradient_row_limit()) {
        int64_t rows = 0;
        const int rc = gsr_gradient_rows(P, geom_buffer, geom_bytes, &rows, s);
        if (rc != 0) return rc;
        if (rows >= gsr_gradient_row_limit()) {
            snprintf(g_err, sizeof(g_err), "%lld gradient rows: the backward keeps gradient-row indices in 28 bits (limit %lld)",
                     (long long)rows, (long long)gsr_gradient_row_limit());
            return -1;
        }
    }
    if (R > 0 && (!binning_buffer || binning_bytes < B.bytes)) return fail(-3, "binning buffer too small");
    if (R > 0 && (!scratch || scratch_bytes < gsr_backward_scratch_bytes(R, W, H))) return fail(-3, "backward scratch too small");
    const char* g = (const char*)geom_buffer;
    const char* b = (const char*)binning_buffer;
    const char* im = (const char*)image_buffer;
    const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    if (R > 0) {
        const int end_bit = 32 + (int)gsr_higher_msb((uint32_t)(gx * gy));
        const bool in_b = !gsr_variant("BIN", 1) && gsr_sort_result_in_b(end_bit);
        const uint32_t* point_list = (const uint32_t*)(b + (in_b ? B.vals_b : B.vals_a));
        StageTimer t(GSR_STAGE_RENDER_BWD, s);
        gsr_launch_render_backward(W, H, (const uint32_t*)(im + I.ranges), point_list, (const float*)(g + G.splat), background,
                                   (const float*)(im + I.final_T), (const uint32_t*)(im + I.n_contrib), dL_dpix,
                                   gsr_variant("BIN", 1) ? (const uint32_t*)(b + B.order) : nullptr, (float*)scratch,
                                   (float)R / (float)(gx * gy),
                                   gsr_blend_log_for(R, W, H) ? (const unsigned long long*)(b + B.blend_log) : nullptr,
                                   (const uint32_t*)(im + I.quad_hi), s);
    }
    a.cov3D_in = cov3D_precomp ? cov3D_precomp : (const float*)(g + G.cov3D);
    a.clamped = (const uint8_t*)(g + G.clamped); a.splat = (const float*)(g + G.splat);
    a.tiles_touched = (const uint32_t*)(g + G.tiles_touched);
    a.point_offsets = (const uint32_t*)(g + G.point_offsets);
    a.row_offsets = (const uint32_t*)(g + G.row_offsets);
    // the work split stored by the forward's offset kernels; with no instance at all they did not run (and the split is the plain one)
    a.wg_first = (R > 0 && gsr_variant("GB_BALANCED", 1)) ? (const uint32_t*)(g + G.wg_first) : nullptr;
    a.row_total = (const uint32_t*)(g + G.row_block_offsets) + (P + 255) / 256;
    a.rows_bound = (uint32_t)R;
    a.grad_rows = (const float*)scratch;
    a.sums_out = g_capture_sums;
    { StageTimer t(GSR_STAGE_GAUSS_BWD, s); gsr_launch_gaussian_backward(a, s); }
    return check_hip("backward");
}

int gsr_backward(int P, int D, int M, int64_t R, int W, int H, const float* background, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* campos, float tan_fovx, float tan_fovy, const int* radii, const void* geom_buffer,
                 size_t geom_bytes, const void* binning_buffer, size_t binning_bytes, const void* image_buffer,
                 size_t image_bytes, void* scratch, size_t scratch_bytes, const float* dL_dpix, float* dL_dmeans2D,
                 float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscales, float* dL_drotations, void* stream)
{
    (void)colors_precomp;
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) return 0;
    if (!background || !means3D || !viewmatrix || !projmatrix || !campos || !radii || !geom_buffer || !image_buffer || !dL_dpix ||
        !dL_dmeans2D || !dL_dcolors || !dL_dopacity || !dL_dmeans3D || !dL_dcov3D || !dL_dscales || !dL_drotations)
        return fail(-1, "NULL required pointer");
    if (M > 0 && !dL_dsh) return fail(-1, "dL_dsh is NULL");
    GaussBackwardArgs a;
    a.P = P; a.D = D; a.M = M; a.W = W; a.H = H;
    a.means3D = means3D; a.shs = shs; a.scales = scales; a.rotations = rotations;
    a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; a.campos = campos;
    a.radii = radii;
    a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
    a.has_shs = shs != nullptr; a.has_scales = scales != nullptr;
    a.dL_dmeans2D = dL_dmeans2D; a.dL_dcolors = dL_dcolors; a.dL_dopacity = dL_dopacity; a.dL_dmeans3D = dL_dmeans3D;
    a.dL_dcov3D = dL_dcov3D; a.dL_dsh = dL_dsh; a.dL_dscales = dL_dscales; a.dL_drots = dL_drotations;
    a.raw = 0; a.features_dc = nullptr; a.features_rest = nullptr; a.dL_dfeatures_dc = nullptr; a.dL_dfeatures_rest = nullptr;
    return run_backward(a, R, background, geom_buffer, geom_bytes, binning_buffer, binning_bytes, image_buffer, image_bytes, scratch,
                        scratch_bytes, dL_dpix, cov3D_precomp, s);
}

int gsr_backward_params(int P, int D, int M, int64_t R, int W, int H, const float* background, const float* xyz,
                        const float* features_dc, const float* features_rest, const float* scaling_raw, const float* rotation_raw,
                        float scale_modifier, const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                        float tan_fovy, const int* radii, const void* geom_buffer, size_t geom_bytes, const void* binning_buffer,
                        size_t binning_bytes, const void* image_buffer, size_t image_bytes, void* scratch, size_t scratch_bytes,
                        const float* dL_dpix, float* dL_dmeans2D, float* dL_dxyz, float* dL_dfeatures_dc, float* dL_dfeatures_rest,
                        float* dL_dopacity_raw, float* dL_dscaling_raw, float* dL_drotation_raw, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) return 0;
    if (!background || !xyz || !features_dc || !scaling_raw || !rotation_raw || !viewmatrix || !projmatrix || !campos || !radii ||
        !geom_buffer || !image_buffer || !dL_dpix || !dL_dmeans2D || !dL_dxyz || !dL_dfeatures_dc || !dL_dopacity_raw ||
        !dL_dscaling_raw || !dL_drotation_raw)
        return fail(-1, "NULL required pointer");
    if (M < 1 || (M > 1 && (!features_rest || !dL_dfeatures_rest))) return fail(-1, "features_rest / dL_dfeatures_rest is NULL");
    GaussBackwardArgs a;
    a.P = P; a.D = D; a.M = M; a.W = W; a.H = H;
    a.means3D = xyz; a.shs = nullptr; a.scales = scaling_raw; a.rotations = rotation_raw;
    a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; a.campos = campos;
    a.radii = radii;
    a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;
    a.has_shs = 1; a.has_scales = 1;
    a.dL_dmeans2D = dL_dmeans2D; a.dL_dcolors = nullptr; a.dL_dopacity = dL_dopacity_raw; a.dL_dmeans3D = dL_dxyz;
    a.dL_dcov3D = nullptr; a.dL_dsh = nullptr; a.dL_dscales = dL_dscaling_raw; a.dL_drots = dL_drotation_raw;
    a.raw = 1; a.features_dc = features_dc; a.features_rest = features_rest;
    a.dL_dfeatures_dc = dL_dfeatures_dc; a.dL_dfeatures_rest = dL_dfeatures_rest;
    return run_backward(a, R, background, geom_buffer, geom_bytes, binning_buffer, binning_bytes, image_buffer, image_bytes, scratch,
                        scratch_bytes, dL_dpix, nullptr, s);
}

int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream)
{
    (void)projmatrix;
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !present) return fail(-1, "NULL required pointer");
    gsr_launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
    return check_hip("mark_visible");
}

size_t gsr_loss_scratch_bytes(int C, int H, int W)
{
    size_t a, b;
    return gsr_loss_scratch_layout(C, H, W, &a, &b);
}

int gsr_loss_forward(int C, int H, int W, const float* img1, const float* img2, const float* img_l1, float lambda_dssim,
                     void* scratch, size_t scratch_bytes, float* out3, void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0) return fail(-1, "invalid image size");
    if (!img1 || !img2 || !scratch || !out3) return fail(-1, "NULL required pointer");
    size_t mo, po;
    if (scratch_bytes < gsr_loss_scratch_layout(C, H, W, &mo, &po)) return fail(-3, "loss scratch too small");
    gsr_launch_loss_forward(C, H, W, img1, img2, img_l1, lambda_dssim, (float*)((char*)scratch + mo), (float*)((char*)scratch + po), out3,
                            (hipStream_t)stream);
    return check_hip("loss_forward");
}

int gsr_loss_backward(int C, int H, int W, const float* img1, const float* img2, const float* img_l1, float lambda_dssim,
                      const void* scratch, size_t scratch_bytes, const float* grad_loss, float* dL_dimg1, float* dL_dimg_l1,
                      void* stream)
{
    if (C <= 0 || H <= 0 || W <= 0) return fail(-1, "invalid image size");
    if (!img1 || !img2 || !scratch || !dL_dimg1 || (img_l1 && !dL_dimg_l1)) return fail(-1, "NULL required pointer");
    size_t mo, po;
    if (scratch_bytes < gsr_loss_scratch_layout(C, H, W, &mo, &po)) return fail(-3, "loss scratch too small");
    gsr_launch_loss_backward(C, H, W, img1, img2, img_l1, (const float*)((const char*)scratch + mo), lambda_dssim, grad_loss,
                             dL_dimg1, dL_dimg_l1, (hipStream_t)stream);
    return check_hip("loss_backward");
}

size_t gsr_appearance_saved_bytes(int H, int W) { return H > 0 && W > 0 ? (size_t)2 * 16 * H * W * sizeof(float) : 0; }  // relu(conv2), u

size_t gsr_appearance_scratch_bytes(int H, int W)
{
    if (H <= 0 || W <= 0) return 0;
    size_t a, b, c, d;
    return gsr_appearance_layout(H, W, &a, &b, &c, &d);
}

int gsr_appearance_tail_forward(int H, int W, int feat_h, int feat_w, const float* feat, const float* conv2_weight,
                                const float* conv2_bias, const float* conv3_weight, const float* conv3_bias, const float* image,
                                float* mapping, float* transformed, void* saved, size_t saved_bytes, void* stream)
{
    if (H <= 0 || W <= 0 || feat_h <= 0 || feat_w <= 0) return fail(-1, "invalid image / feature size");
    if (!feat || !conv2_weight || !conv2_bias || !conv3_weight || !conv3_bias || !image || !mapping || !transformed || !saved)
        return fail(-1, "NULL required pointer");
    if (saved_bytes < gsr_appearance_saved_bytes(H, W)) return fail(-3, "appearance saved buffer too small");
    gsr_launch_appearance_forward(H, W, feat_h, feat_w, feat, conv2_weight, conv2_bias, conv3_weight, conv3_bias, image, mapping,
                                  transformed, (float*)saved, (hipStream_t)stream);
    return check_hip("appearance_tail_forward");
}

int gsr_appearance_tail_backward(int H, int W, int feat_h, int feat_w, const float* feat, const float* conv2_weight,
                                 const float* conv3_weight, const float* image, const float* mapping, const void* saved,
                                 size_t saved_bytes, const float* dL_dtransformed, const float* dL_dmapping, float* dL_dimage,
                                 float* dL_dfeat, float* dL_dconv2_weight, float* dL_dconv2_bias, float* dL_dconv3_weight,
                                 float* dL_dconv3_bias, void* scratch, size_t scratch_bytes, int mask_relu_of_feat, void* stream)
{
    if (H <= 0 || W <= 0 || feat_h <= 0 || feat_w <= 0) return fail(-1, "invalid image / feature size");
    if (!feat || !conv2_weight || !conv3_weight || !image || !mapping || !saved || !dL_dtransformed || !dL_dimage || !dL_dfeat ||
        !dL_dconv2_weight || !dL_dconv2_bias || !dL_dconv3_weight || !dL_dconv3_bias || !scratch)
        return fail(-1, "NULL required pointer");
    if (saved_bytes < gsr_appearance_saved_bytes(H, W)) return fail(-3, "appearance saved buffer too small");
    if (scratch_bytes < gsr_appearance_scratch_bytes(H, W)) return fail(-3, "appearance scratch too small");
    gsr_launch_appearance_backward(H, W, feat_h, feat_w, feat, conv2_weight, conv3_weight, image, mapping, (const float*)saved,
                                   dL_dtransformed, dL_dmapping, dL_dimage, dL_dfeat, dL_dconv2_weight, dL_dconv2_bias,
                                   dL_dconv3_weight, dL_dconv3_bias, (char*)scratch, mask_relu_of_feat, (hipStream_t)stream);
    return check_hip("appearance_tail_backward");
}

size_t gsr_appearance_up_scratch_bytes(int feat_h, int feat_w) { return feat_h > 0 && feat_w > 0 ? gsr_appearance_up_scratch_bytes_impl() : 0; }

int gsr_appearance_up_forward(int feat_h, int feat_w, const float* x, const float* weight, const float* bias, float* feat, void* stream)
{
    if (feat_h <= 0 || feat_w <= 0 || (feat_h & 1) || (feat_w & 1)) return fail(-1, "feature size must be positive and even (PixelShuffle(2))");
    if (!x || !weight || !bias || !feat) return fail(-1, "NULL required pointer");
    gsr_launch_appearance_up_forward(feat_h, feat_w, x, weight, bias, feat, (hipStream_t)stream);
    return check_hip("appearance_up_forward");
}

int gsr_appearance_up_backward(int feat_h, int feat_w, const float* x, const float* weight, const float* dz, float* dL_dx,
                               float* dL_dweight, float* dL_dbias, void* scratch, size_t scratch_bytes, void* stream)
{
    if (feat_h <= 0 || feat_w <= 0 || (feat_h & 1) || (feat_w & 1)) return fail(-1, "feature size must be positive and even (PixelShuffle(2))");
    if (!x || !weight || !dz || !dL_dx || !dL_dweight || !dL_dbias || !scratch) return fail(-1, "NULL required pointer");
    if (scratch_bytes < gsr_appearance_up_scratch_bytes(feat_h, feat_w)) return fail(-3, "appearance up-block scratch too small");
    gsr_launch_appearance_up_backward(feat_h, feat_w, x, weight, dz, dL_dx, dL_dweight, dL_dbias, (char*)scratch, (hipStream_t)stream);
    return check_hip("appearance_up_backward");
}

size_t gsr_bias_relu_scratch_bytes(int C) { return C > 0 ? gsr_bias_relu_scratch_bytes_impl(C) : 0; }

int gsr_bias_relu_forward(int C, int64_t n, float* x, const float* bias, void* stream)
{
    if (C <= 0 || C > 65535 || n <= 0) return fail(-1, "bias_relu: C must be in 1..65535 and n positive");
    if (!x || !bias) return fail(-1, "NULL required pointer");
    gsr_launch_bias_relu_forward(C, (long long)n, x, bias, (hipStream_t)stream);
    return check_hip("bias_relu_forward");
}

int gsr_bias_relu_backward(int C, int64_t n, const float* dL_dy, const float* y, float* dL_dx, float* dL_dbias, void* scratch,
                           size_t scratch_bytes, void* stream)
{
    if (C <= 0 || C > 65535 || n <= 0) return fail(-1, "bias_relu: C must be in 1..65535 and n positive");
    if (!dL_dy || !y || !dL_dx || !dL_dbias || !scratch) return fail(-1, "NULL required pointer");
    if (scratch_bytes < gsr_bias_relu_scratch_bytes(C)) return fail(-3, "bias_relu scratch too small");
    gsr_launch_bias_relu_backward(C, (long long)n, dL_dy, y, dL_dx, dL_dbias, (double*)scratch, (hipStream_t)stream);
    return check_hip("bias_relu_backward");
}

size_t gsr_mse_scratch_bytes(int rows) { return rows > 0 ? gsr_mse_scratch_bytes_impl(rows) : 0; }

int gsr_mse(int rows, int64_t n, const float* img1, const float* img2, float* out, void* scratch, size_t scratch_bytes, void* stream)
{
    if (rows <= 0 || n <= 0 || rows > 65535) return fail(-1, "mse: rows must be in 1..65535 and n positive");
    if (!img1 || !img2 || !out || !scratch) return fail(-1, "NULL required pointer");
    if (scratch_bytes < gsr_mse_scratch_bytes(rows)) return fail(-3, "mse scratch too small");
    gsr_launch_mse(rows, (long long)n, img1, img2, out, (double*)scratch, (hipStream_t)stream);
    return check_hip("mse");
}

int gsr_adam_step(int n_tensors, const gsr_adam_tensor* tensors, double beta1, double beta2, double eps, void* stream)
{
    static_assert(sizeof(gsr_adam_tensor) == sizeof(GsrAdamTensorHost), "ABI struct mismatch");
    if (n_tensors < 0 || (n_tensors > 0 && !tensors)) return fail(-1, "invalid tensor list");
    for (int k = 0; k < n_tensors; k++) {
        const gsr_adam_tensor& t = tensors[k];
        if (t.n < 0 || t.step < 1) return fail(-1, "adam: n must be >= 0 and step >= 1");
        if (t.n > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq)) return fail(-1, "adam: NULL tensor pointer");
    }
    gsr_launch_adam(n_tensors, reinterpret_cast<const GsrAdamTensorHost*>(tensors), beta1, beta2, eps, (hipStream_t)stream);
    return check_hip("adam_step");
}

int gsr_densification_stats(int P, const int* radii, const float* viewspace_grad, float* max_radii2D, float* xyz_gradient_accum,
                            float* denom, void* stream)
{
    if (P < 0) return fail(-1, "invalid P");
    if (P == 0) return 0;
    if (!radii || !viewspace_grad || !max_radii2D || !xyz_gradient_accum || !denom) return fail(-1, "NULL required pointer");
    gsr_launch_densification_stats(P, radii, viewspace_grad, max_radii2D, xyz_gradient_accum, denom, (hipStream_t)stream);
    return check_hip("densification_stats");
}

// scratch: block sums, block offsets (one u32 per 256 rows each), total
static size_t compact_layout(int P, size_t* offsets_off, size_t* total_off)
{
    const size_t nb = ((size_t)(P > 0 ? P : 1) + 255) / 256;
    size_t o = gsr_align(nb * 4);
    *offsets_off = o; o = gsr_align(o + nb * 4);
    *total_off = o; o = gsr_align(o + 16);
    return o;
}
size_t gsr_compact_scratch_bytes(int P) { size_t a, b; return compact_layout(P, &a, &b); }

int gsr_compact_plan(int P, const uint8_t* keep_mask, void* scratch, size_t scratch_bytes, int64_t* count_host, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!count_host) return fail(-1, "count_host is NULL");
    *count_host = 0;
    if (P < 0) return fail(-1, "invalid P");
    if (P == 0) return 0;
    size_t off, tot;
    if (!keep_mask || !scratch || scratch_bytes < compact_layout(P, &off, &tot)) return fail(-3, "compact scratch too small");
    char* c = (char*)scratch;
    const int nb = (P + 255) / 256;
    gsr_launch_compact_count(P, keep_mask, (uint32_t*)c, s);
    gsr_launch_scan_blocks(nb, (const uint32_t*)c, (uint32_t*)(c + off), (uint32_t*)(c + tot), nullptr, 0, s);
    uint32_t total = 0;
    hipError_t e = hipMemcpyAsync(&total, c + tot, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(-100, "compact_plan: %s", hipGetErrorString(e));
    *count_host = (int64_t)total;
    return check_hip("compact_plan");
}

int gsr_compact_rows(int P, const uint8_t* keep_mask, const void* scratch, size_t scratch_bytes, int n_tensors,
                     const gsr_compact_tensor* tensors, void* stream)
{
    static_assert(sizeof(gsr_compact_tensor) == sizeof(GsrCompactTensorHost), "ABI struct mismatch");
    if (P < 0 || n_tensors < 0 || (n_tensors > 0 && !tensors)) return fail(-1, "invalid arguments");
    if (P == 0 || n_tensors == 0) return 0;
    size_t off, tot;
    if (!keep_mask || !scratch || scratch_bytes < compact_layout(P, &off, &tot)) return fail(-3, "compact scratch too small");
    for (int k = 0; k < n_tensors; k++)
        if (tensors[k].row_floats < 0 || (tensors[k].row_floats > 0 && (!tensors[k].src || !tensors[k].dst)))
            return fail(-1, "compact: bad tensor descriptor");
    gsr_launch_compact_rows(P, keep_mask, (const uint32_t*)((const char*)scratch + off), n_tensors,
                            reinterpret_cast<const GsrCompactTensorHost*>(tensors), (hipStream_t)stream);
    return check_hip("compact_rows");
}

size_t gsr_knn_scratch_bytes(int P) { return gsr_knn_scratch_layout_bytes(P); }

int gsr_knn_mean_dist2(int P, const float* points, float* out, void* scratch, size_t scratch_bytes, void* stream)
{
    if (P == 0) return 0;
    if (!points || !out || !scratch) return fail(-1, "NULL required pointer");
    if (gsr_knn_run(P, points, out, scratch, scratch_bytes, g_knn_evals, (hipStream_t)stream) != 0) return fail(-3, "knn scratch too small");
    return check_hip("knn_mean_dist2");
}

int gsr_debug_export(int P, int64_t R, int W, int H, const void* geom_buffer, const void* binning_buffer,
                     const void* image_buffer, float* means2D, float* depths, float* conic_opacity, float* rgb, float* cov3D,
                     uint8_t* clamped, uint32_t* tiles_touched, uint32_t* point_offsets, uint64_t* keys_sorted,
                     uint32_t* point_list, uint32_t* ranges, float* final_T, uint32_t* n_contrib, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (P == 0) return 0;
    const GeomLayout G = gsr_geom_layout(P);
    const ImageLayout I = gsr_image_layout(W, H);
    const BinLayout B = gsr_bin_layout((long)R, P, W, H);
    const char* g = (const char*)geom_buffer;
    const char* b = (const char*)binning_buffer;
    const char* im = (const char*)image_buffer;
    const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    const size_t N = (size_t)W * H;
    if (g) {
        hipLaunchKernelGGL(export_geom_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, (const float*)(g + G.splat), means2D, depths,
                           conic_opacity, rgb);
        if (cov3D) hipMemcpyAsync(cov3D, g + G.cov3D, (size_t)P * 24, hipMemcpyDeviceToDevice, s);
        if (clamped) hipMemcpyAsync(clamped, g + G.clamped, (size_t)P * 3, hipMemcpyDeviceToDevice, s);
        if (tiles_touched) hipMemcpyAsync(tiles_touched, g + G.tiles_touched, (size_t)P * 4, hipMemcpyDeviceToDevice, s);
        if (point_offsets) hipMemcpyAsync(point_offsets, g + G.point_offsets, (size_t)P * 4, hipMemcpyDeviceToDevice, s);
    }
    if (b && R > 0) {
        const int end_bit = 32 + (int)gsr_higher_msb((uint32_t)(gx * gy));
        const int bin = gsr_variant("BIN", 1);
        const int in_b = bin ? 0 : gsr_sort_result_in_b(end_bit);
        if (keys_sorted && bin && gsr_variant("KEYS", 0))  // the forward wrote the sorted keys itself (GSR_KEYS=1): hand those out
            hipMemcpyAsync(keys_sorted, b + B.keys_b, (size_t)R * 8, hipMemcpyDeviceToDevice, s);
        else if (keys_sorted && bin && im && g)
            hipLaunchKernelGGL(export_keys_kernel, dim3(gx * gy), dim3(256), 0, s, (const uint32_t*)(im + I.ranges),
                               (const uint32_t*)(b + B.vals_a), (const float*)(g + G.splat), (unsigned long long*)keys_sorted);
        else if (keys_sorted)
            hipMemcpyAsync(keys_sorted, b + (in_b ? B.keys_b : B.keys_a), (size_t)R * 8, hipMemcpyDeviceToDevice, s);
        if (point_list) hipMemcpyAsync(point_list, b + (in_b ? B.vals_b : B.vals_a), (size_t)R * 4, hipMemcpyDeviceToDevice, s);
    }
    if (im) {
        if (ranges) hipMemcpyAsync(ranges, im + I.ranges, (size_t)gx * gy * 8, hipMemcpyDeviceToDevice, s);
        if (final_T) hipMemcpyAsync(final_T, im + I.final_T, N * 4, hipMemcpyDeviceToDevice, s);
        if (n_contrib) hipMemcpyAsync(n_contrib, im + I.n_contrib, N * 4, hipMemcpyDeviceToDevice, s);
    }
    return check_hip("debug_export");
}

int gsr_debug_capture_sums(float* sums)
{
    g_capture_sums = sums;
    return 0;
}

// Phase profile of the blend kernels: sixteen wave-cycle sums, [0..7] the backward's log walk (render_backward.hip), [8..15] the forward's
// quadrant walk (render_forward.hip), GSR_LOG_PROFILE; returns 1 if the library was built with the profile, 0 if not (all zeros),
// negative on error.  Synchronises the device.
int gsr_debug_blend_profile(uint64_t* counters16)
{
    if (!counters16) return fail(-1, "counters16 is NULL");
    if (hipDeviceSynchronize() != hipSuccess) return check_hip("gsr_debug_blend_profile");
    const int rc = gsr_blend_profile_fetch((unsigned long long*)counters16);
    const int rf = gsr_forward_profile_fetch((unsigned long long*)counters16 + 8);
    return (rc < 0 || rf < 0) ? fail(-4, "gsr_debug_blend_profile: symbol copy failed") : rc;
}

int gsr_debug_knn_counter(uint64_t* counter)
{
    g_knn_evals = reinterpret_cast<unsigned long long*>(counter);
    return 0;
}

const char* gsr_build_info(void)
{
#ifndef GSR_EXTRA_FLAGS
#define GSR_EXTRA_FLAGS ""
#endif
    static char info[256];
    snprintf(info, sizeof(info), "abi=%d ablate=%d bin_ablate=%d extra=[%s]", GSR_ABI_VERSION, gsr_render_backward_ablate(),
             gsr_binning_ablate(), GSR_EXTRA_FLAGS);
    return info;
}

void gsr_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof = on != 0;
}

void gsr_profile_select(uint32_t stage_mask)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_mask = stage_mask;
}

int gsr_profile_collect(double* ms_per_stage, int64_t* launches_per_stage, int nstages)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < nstages; i++) { ms_per_stage[i] = 0.0; launches_per_stage[i] = 0; }
    int rc = 0;
    for (const ProfRec& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) rc = -100;
        else if (r.stage < nstages) { ms_per_stage[r.stage] += ms; launches_per_stage[r.stage] += r.count ? 1 : 0; }
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    g_recs.clear();
    return rc;
}

size_t gsr_sort_scratch_bytes(int64_t n)
{
    const BinLayout B = gsr_bin_layout((long)n, 1, 16, 16);
    return gsr_align(8 * 256 * 4) + gsr_align((size_t)B.sort_blocks * 256 * 4);
}

int gsr_sort_pairs(uint64_t* keys_a, uint64_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, int64_t n, int end_bit,
                   void* scratch, size_t scratch_bytes, void* stream)
{
    if (n < 0 || n > 0xFFFFFFFFll || end_bit < 1 || end_bit > 64) return fail(-1, "invalid sort arguments");
    if (n == 0) return ((end_bit + 7) / 8) & 1;
    if (!keys_a || !keys_b || !vals_a || !vals_b || !scratch) return fail(-1, "NULL required pointer");
    if (scratch_bytes < gsr_sort_scratch_bytes(n)) return fail(-3, "sort scratch too small");
    const BinLayout B = gsr_bin_layout((long)n, 1, 16, 16);
    uint32_t* hg = (uint32_t*)scratch;
    uint32_t* hb = (uint32_t*)((char*)scratch + gsr_align(8 * 256 * 4));
    const int r = gsr_launch_radix_sort(keys_a, keys_b, vals_a, vals_b, (uint32_t)n, end_bit, hg, hb, B.sort_blocks, (hipStream_t)stream);
    const int c = check_hip("sort_pairs");
    return c ? c : r;
}

}  // extern "C"
