// distCUDA2 for MI355X: mean squared distance to the 3 nearest other points (exact), used once per
// sub-scene to initialise Gaussian scales (scene/gaussian_model.py:144).  Replaces SimpleKNN::knn,
// submodules/simple-knn/simple_knn.cu:185-221.
//
// Same pruning idea as the reference (Morton order + AABBs of 1024-point boxes, reject radius from the +-3
// Morton neighbours) but restructured for wave64 / LDS: a workgroup owns 256 consecutive Morton-sorted queries;
// for every box that ANY of them cannot reject, the box's points are loaded once, coalesced, into LDS and the
// interested lanes scan them with broadcast reads.  The reference walks boxes with one divergent thread per
// point and gathers every candidate from global memory through the index array (simple_knn.cu:147-183).
// No host round trips (the reference copies min/max back to the host twice, simple_knn.cu:196-200).
// The result is the exact 3-NN set, so it does not depend on the pruning order; distances are evaluated as
// dx*dx + dy*dy + dz*dz without contraction, matching the CPU oracle bit for bit.
#include <float.h>
#include "common.h"

namespace {

// the reference uses boxes of 1024 points (simple_knn.cu:12); see gsr_knn_run for the sizes used here
constexpr int BOX_MIN = 64;    // smallest box size selectable here (sizes the scratch)

struct KnnLayout {
    size_t partial, minmax, keys_a, keys_b, vals_a, vals_b, hist_global, hist_blocks, sorted, boxes, supers, subs, tops, bytes;
    uint32_t sort_blocks, num_boxes, red_blocks;
};

KnnLayout knn_layout(int P)
{
    KnnLayout L;
    size_t o = 0, p = (size_t)(P > 0 ? P : 1);
    L.sort_blocks = (uint32_t)((p + GSR_SORT_TILE - 1) / GSR_SORT_TILE);
    L.num_boxes = (uint32_t)((p + BOX_MIN - 1) / BOX_MIN);
    L.red_blocks = (uint32_t)((p + 1023) / 1024);
    if (L.red_blocks > 1024) L.red_blocks = 1024;
    L.partial = o; o = gsr_align(o + (size_t)L.red_blocks * 6 * 4);
    L.minmax = o; o = gsr_align(o + 6 * 4);
    L.keys_a = o; o = gsr_align(o + p * 8);
    L.keys_b = o; o = gsr_align(o + p * 8);
    L.vals_a = o; o = gsr_align(o + p * 4);
    L.vals_b = o; o = gsr_align(o + p * 4);
    L.hist_global = o; o = gsr_align(o + 8 * 256 * 4);
    L.hist_blocks = o; o = gsr_align(o + (size_t)L.sort_blocks * 256 * 4);
    L.sorted = o; o = gsr_align(o + p * 16);
    L.boxes = o; o = gsr_align(o + (size_t)L.num_boxes * 6 * 4);
    L.supers = o; o = gsr_align(o + (size_t)L.num_boxes * 6 * 4);  // group AABBs (at most one per box)
    L.subs = o; o = gsr_align(o + (size_t)L.num_boxes * 4 * 6 * 4);  // AABBs of the four 16-point quarters of every 64-point box
    L.tops = o; o = gsr_align(o + ((size_t)L.num_boxes / 32 + 1) * 6 * 4);  // AABBs of 32 groups each (the wave kernel's top level)
    L.bytes = o;
    return L;
}

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// component-wise min / max with init (0,0,0): simple_knn.cu:191-200
__global__ void __launch_bounds__(256) minmax_partial_kernel(int P, const float* __restrict__ pts, float* __restrict__ partial)
{
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
        for (int c = 0; c < 3; c++) {
            const float v = pts[3 * (size_t)i + c];
            mn[c] = fminf(mn[c], v);
            mx[c] = fmaxf(mx[c], v);
        }
    }
    __shared__ float red[4][6];
    for (int c = 0; c < 3; c++) { mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]); }
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; c++) { red[threadIdx.x >> 6][c] = mn[c]; red[threadIdx.x >> 6][3 + c] = mx[c]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[0][threadIdx.x];
        for (int w = 1; w < 4; w++) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
        partial[blockIdx.x * 6 + threadIdx.x] = v;
    }
}
// one wave: every lane takes every 64th partial row, then six wave reductions (round 3: six threads walked all rows one dependent
// load after the other -- 51 us for a kilobyte-sized reduce)
__global__ void __launch_bounds__(64) minmax_final_kernel(int nb, const float* __restrict__ partial, float* __restrict__ minmax)
{
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nb; b += 64)
        for (int c = 0; c < 3; c++) { mn[c] = fminf(mn[c], partial[b * 6 + c]); mx[c] = fmaxf(mx[c], partial[b * 6 + 3 + c]); }
    for (int c = 0; c < 3; c++) { mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]); }
    if (threadIdx.x == 0)
        for (int c = 0; c < 3; c++) { minmax[c] = mn[c]; minmax[3 + c] = mx[c]; }
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x)  // simple_knn.cu:45-52
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

__global__ void __launch_bounds__(256) morton_kernel(int P, const float* __restrict__ pts, const float* __restrict__ minmax,
                                                     uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    // simple_knn.cu:54-61
    const uint32_t mxc = prep_morton((uint32_t)(((x - minmax[0]) / (minmax[3] - minmax[0])) * 1023.0f));
    const uint32_t myc = prep_morton((uint32_t)(((y - minmax[1]) / (minmax[4] - minmax[1])) * 1023.0f));
    const uint32_t mzc = prep_morton((uint32_t)(((z - minmax[2]) / (minmax[5] - minmax[2])) * 1023.0f));
    keys[i] = (uint64_t)(mxc | (myc << 1) | (mzc << 2));
    vals[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256) gather_sorted_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                            float4* __restrict__ sorted)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint32_t o = order[i];
    sorted[i] = make_float4(pts[3 * (size_t)o], pts[3 * (size_t)o + 1], pts[3 * (size_t)o + 2], __uint_as_float(o));
}

// AABB of each run of 1024 sorted points (boxMinMax, simple_knn.cu:78-117)
template <int BOX>
__global__ void __launch_bounds__(256) box_minmax_kernel(int P, const float4* __restrict__ sorted, float* __restrict__ boxes)
{
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const int base = blockIdx.x * BOX;
    for (int k = threadIdx.x; k < BOX; k += 256) {
        const int i = base + k;
        if (i < P) {
            const float4 p = sorted[i];
            mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
            mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
        }
    }
    __shared__ float red[4][6];
    for (int c = 0; c < 3; c++) { mn[c] = wave_min(mn[c]); mx[c] = wave_max(mx[c]); }
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; c++) { red[threadIdx.x >> 6][c] = mn[c]; red[threadIdx.x >> 6][3 + c] = mx[c]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[0][threadIdx.x];
        for (int w = 1; w < 4; w++) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
        boxes[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// AABB of each run of 16 sorted points (the quarters of the 64-point boxes): one thread per run, 256 contiguous bytes each
__global__ void __launch_bounds__(256) sub_minmax_kernel(int P, const float4* __restrict__ sorted, float* __restrict__ subs)
{
    const int sidx = blockIdx.x * 256 + threadIdx.x;
    const int base = sidx * 16;
    if (base >= P) return;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const int cnt = min(16, P - base);
    for (int k = 0; k < cnt; k++) {
        const float4 p = sorted[base + k];
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
    for (int c = 0; c < 3; c++) { subs[sidx * 6 + c] = mn[c]; subs[sidx * 6 + 3 + c] = mx[c]; }
}

// AABB of each group of GROUP consecutive boxes (a Morton run of GROUP * BOX points): the first level of the two-level cull
__global__ void __launch_bounds__(256) super_minmax_kernel(uint32_t num_boxes, int group, const float* __restrict__ boxes,
                                                           float* __restrict__ supers)
{
    const uint32_t sidx = blockIdx.x * 256 + threadIdx.x;
    const uint32_t first = sidx * (uint32_t)group;
    if (first >= num_boxes) return;
    const uint32_t last = min(num_boxes, first + (uint32_t)group);
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t b = first; b < last; b++)
        for (int c = 0; c < 3; c++) { mn[c] = fminf(mn[c], boxes[b * 6 + c]); mx[c] = fmaxf(mx[c], boxes[b * 6 + 3 + c]); }
    for (int c = 0; c < 3; c++) { supers[sidx * 6 + c] = mn[c]; supers[sidx * 6 + 3 + c] = mx[c]; }
}

// distBoxPoint, simple_knn.cu:119-129
__device__ __forceinline__ float dist_box_point(const float* __restrict__ bx, float qx, float qy, float qz)
{
    const float mnx = bx[0], mny = bx[1], mnz = bx[2], mxx = bx[3], mxy = bx[4], mxz = bx[5];
    float ddx = 0.f, ddy = 0.f, ddz = 0.f;
    if (qx < mnx || qx > mxx) ddx = fminf(fabsf(qx - mnx), fabsf(qx - mxx));
    if (qy < mny || qy > mxy) ddy = fminf(fabsf(qy - mny), fabsf(qy - mxy));
    if (qz < mnz || qz > mxz) ddz = fminf(fabsf(qz - mnz), fabsf(qz - mxz));
    return ddx * ddx + ddy * ddy + ddz * ddz;
}

__device__ __forceinline__ void update3(float qx, float qy, float qz, float cx, float cy, float cz, float& b0, float& b1, float& b2)
{
    // updateKBest<3>, simple_knn.cu:131-145
    const float dx = cx - qx, dy = cy - qy, dz = cz - qz;
    float dist = dx * dx + dy * dy + dz * dz;
    if (b0 > dist) { const float t = b0; b0 = dist; dist = t; }
    if (b1 > dist) { const float t = b1; b1 = dist; dist = t; }
    if (b2 > dist) { b2 = dist; }
}

template <int BOX>
__global__ void __launch_bounds__(256) knn_kernel(int P, uint32_t num_boxes, int group, const float4* __restrict__ sorted,
                                                  const float* __restrict__ boxes, const float* __restrict__ supers,
                                                  float* __restrict__ out, unsigned long long* __restrict__ evals)
{
    __shared__ float s_x[BOX], s_y[BOX], s_z[BOX];
    uint32_t my_evals = 0;  // candidate distances this query evaluated (only summed up when a counter was handed in: bench.py)
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const bool valid = idx < P;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    uint32_t orig = 0;
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    if (valid) {
        const float4 q = sorted[idx];
        qx = q.x; qy = q.y; qz = q.z; orig = __float_as_uint(q.w);
        // seed the reject radius from the +-3 Morton neighbours (simple_knn.cu:156-163)
        const int lo = max(0, idx - 3), hi = min(P - 1, idx + 3);
        for (int i = lo; i <= hi; i++) {
            if (i == idx) continue;
            const float4 c = sorted[i];
            update3(qx, qy, qz, c.x, c.y, c.z, b0, b1, b2);
        }
    }
    const float reject = b2;
    b0 = FLT_MAX; b1 = FLT_MAX; b2 = FLT_MAX;
    const uint32_t num_supers = (num_boxes + (uint32_t)group - 1) / (uint32_t)group;
    for (uint32_t sb = 0; sb < num_supers; sb++) {
        // level 1: a group of boxes nobody in the workgroup can reach is skipped as a whole (the same test as for a box, on the
        // union AABB: it can only pass more often, never drop a box the per-box test would keep)
        const float sdist = dist_box_point(supers + sb * 6, qx, qy, qz);
        if (!__syncthreads_or(valid && !(sdist > reject || sdist > b2))) continue;
        const uint32_t bend = min(num_boxes, (sb + 1) * (uint32_t)group);
        for (uint32_t b = sb * (uint32_t)group; b < bend; b++) {
            const float dist = dist_box_point(boxes + b * 6, qx, qy, qz);
            const bool want = valid && !(dist > reject || dist > b2);
            if (!__syncthreads_or(want)) continue;
            const int base = (int)b * BOX;
            const int cnt = min(BOX, P - base);
            for (int k = threadIdx.x; k < cnt; k += 256) {
                const float4 c = sorted[base + k];
                s_x[k] = c.x; s_y[k] = c.y; s_z[k] = c.z;
            }
            __syncthreads();
            if (want) {
                const int self = idx - base;  // position of this query inside the box, if it is in it
                my_evals += (uint32_t)cnt;
                for (int k = 0; k < cnt; k++) {
                    if (k == self) continue;
                    update3(qx, qy, qz, s_x[k], s_y[k], s_z[k], b0, b1, b2);
                }
            }
            __syncthreads();
        }
    }
    if (valid) out[orig] = (b0 + b1 + b2) / 3.0f;
    if (evals) {
        unsigned long long e = my_evals;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) e += __shfl_xor(e, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(evals, e);
    }
}

// Wave-autonomous variant (round 4, the default): a wave is one 64-point box of Morton-sorted queries and never waits for the other
// waves of its workgroup.  (1) Its own box first: the 64 points go to the wave's LDS slice and every lane scans them -- an exact 3-NN
// inside the box, i.e. a much tighter reject radius than the +-3 Morton neighbours of the reference (simple_knn.cu:156-163).  (2) The
// groups of boxes are visited outward from the wave's own group (own, +1, -1, +2, ...: Morton neighbours are spatial neighbours more often
// than not, so the radius shrinks early); a group, then a box, is entered when ANY lane cannot reject its AABB -- a ballot, no barrier --
// and a box that is entered is loaded once (lane k loads point k, coalesced), parked in the wave's LDS slice and scanned with broadcast
// reads by the lanes that want it.  Exactness: a box is skipped for a lane only when its AABB is farther than the lane's current third-best
// distance, so the result is the exact 3-NN set whatever the order.  500k points: 796 -> ~250 candidate distances per point, no
// __syncthreads in the kernel (the workgroup-cooperative kernel above spends its time in 245 + ~400 workgroup votes per workgroup).
__global__ void __launch_bounds__(256) knn_wave_kernel(int P, uint32_t num_boxes, int group, const float4* __restrict__ sorted,
                                                       const float* __restrict__ boxes, const float* __restrict__ supers,
                                                       const float* __restrict__ subs, const float* __restrict__ tops,
                                                       float* __restrict__ out, unsigned long long* __restrict__ evals)
{
    __shared__ float s_pts[4][3][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t my_box = blockIdx.x * 4u + (uint32_t)wave;  // wave-uniform
    if (my_box >= num_boxes) return;
    float* sx = s_pts[wave][0];
    float* sy = s_pts[wave][1];
    float* sz = s_pts[wave][2];
    const int idx = (int)my_box * 64 + lane;
    const bool valid = idx < P;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    uint32_t orig = 0, my_evals = 0;
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    // loads box `b` (lane k: point k) into the wave's LDS slice; returns the number of points in it
    auto stage = [&](uint32_t b) -> int {
        const int base = (int)b * 64;
        const int cnt = min(64, P - base);
        if (lane < cnt) {
            const float4 c = sorted[base + lane];
            sx[lane] = c.x; sy[lane] = c.y; sz[lane] = c.z;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return cnt;
    };
    auto release = [&]() {  // every lane is done reading the slice before the next box overwrites it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    if (valid) {
        const float4 q = sorted[idx];
        qx = q.x; qy = q.y; qz = q.z; orig = __float_as_uint(q.w);
    }
    {   // (1) the wave's own box
        const int cnt = stage(my_box);
        if (valid) {
            my_evals += (uint32_t)cnt;
            for (int k = 0; k < cnt; k++) {
                if (k == lane) continue;
                update3(qx, qy, qz, sx[k], sy[k], sz[k], b0, b1, b2);
            }
        }
        release();
    }
    // (2) the other boxes: three levels of AABBs above them -- 32 groups ("top"), a group of `group` boxes, a box, and below a box its
    // four quarters.  The tops are visited outward from the wave's own (own, +1, -1, +2, ...), the groups of an entered top likewise
    // outward from the own group's position, so the reject radius has shrunk before most of the tree is looked at.
    const uint32_t num_supers = (num_boxes + (uint32_t)group - 1) / (uint32_t)group;
    const uint32_t num_tops = (num_supers + 31u) / 32u;
    const uint32_t own_super = my_box / (uint32_t)group, own_top = own_super / 32u;
    auto visit_super = [&](uint32_t sb) {
        const float sdist = dist_box_point(supers + sb * 6, qx, qy, qz);
        if (__builtin_amdgcn_ballot_w64(valid && !(sdist > b2)) == 0ull) return;
        const uint32_t bend = min(num_boxes, (sb + 1) * (uint32_t)group);
        for (uint32_t b = sb * (uint32_t)group; b < bend; b++) {
            if (b == my_box) continue;
            const float dist = dist_box_point(boxes + b * 6, qx, qy, qz);
            const bool want = valid && !(dist > b2);
            if (__builtin_amdgcn_ballot_w64(want) == 0ull) continue;
            const int cnt = stage(b);
            // the box's four 16-point quarters have AABBs of their own: a quarter is scanned only by the lanes that cannot reject it
            // (Morton boxes are rarely compact: on uniform points this takes 617 candidate distances per point down to 218)
            for (int j = 0; 16 * j < cnt; j++) {
                const float dj = dist_box_point(subs + ((size_t)b * 4 + j) * 6, qx, qy, qz);
                const bool wj = want && !(dj > b2);
                if (__builtin_amdgcn_ballot_w64(wj) == 0ull) continue;
                if (wj) {
                    const int k1 = min(cnt, 16 * j + 16);
                    my_evals += (uint32_t)(k1 - 16 * j);
                    for (int k = 16 * j; k < k1; k++) update3(qx, qy, qz, sx[k], sy[k], sz[k], b0, b1, b2);
                }
            }
            release();
        }
    };
    for (uint32_t tstep = 0; tstep < 2 * num_tops; tstep++) {
        const long long tp_ = (long long)own_top + ((tstep & 1u) ? (long long)((tstep + 1) >> 1) : -(long long)(tstep >> 1));
        if (tp_ < 0 || tp_ >= (long long)num_tops) continue;
        const uint32_t tp = (uint32_t)tp_;
        if (tp != own_top) {
            const float tdist = dist_box_point(tops + tp * 6, qx, qy, qz);
            if (__builtin_amdgcn_ballot_w64(valid && !(tdist > b2)) == 0ull) continue;
        }
        const uint32_t s0 = tp * 32u, s1 = min(num_supers, s0 + 32u);
        // outward from the own group inside the own top, from the near end inside the others
        const uint32_t centre = tp == own_top ? own_super : (tp < own_top ? s1 - 1 : s0);
        for (uint32_t step = 0; step < 64; step++) {
            const long long sb_ = (long long)centre + ((step & 1u) ? (long long)((step + 1) >> 1) : -(long long)(step >> 1));
            if (sb_ < (long long)s0 || sb_ >= (long long)s1) continue;
            visit_super((uint32_t)sb_);
        }
    }
    if (valid) out[orig] = (b0 + b1 + b2) / 3.0f;
    if (evals) {
        unsigned long long e = my_evals;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) e += __shfl_xor(e, off, 64);
        if (lane == 0) atomicAdd(evals, e);
    }
}

}  // namespace

size_t gsr_knn_scratch_layout_bytes(int P) { return knn_layout(P).bytes; }

int gsr_knn_run(int P, const float* points, float* out, void* scratch, size_t scratch_bytes, unsigned long long* evals, hipStream_t s)
{
    if (P <= 0) return 0;
    const KnnLayout L = knn_layout(P);
    if (scratch_bytes < L.bytes) return -1;
    char* base = (char*)scratch;
    float* partial = (float*)(base + L.partial);
    float* minmax = (float*)(base + L.minmax);
    uint64_t* ka = (uint64_t*)(base + L.keys_a);
    uint64_t* kb = (uint64_t*)(base + L.keys_b);
    uint32_t* va = (uint32_t*)(base + L.vals_a);
    uint32_t* vb = (uint32_t*)(base + L.vals_b);
    float4* sorted = (float4*)(base + L.sorted);
    float* boxes = (float*)(base + L.boxes);
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(L.red_blocks), dim3(256), 0, s, P, points, partial);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, s, (int)L.red_blocks, partial, minmax);
    hipLaunchKernelGGL(morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, minmax, ka, va);
    const int in_b = gsr_launch_radix_sort(ka, kb, va, vb, (uint32_t)P, 32, (uint32_t*)(base + L.hist_global),
                                           (uint32_t*)(base + L.hist_blocks), L.sort_blocks, s);
    const uint32_t* order = in_b ? vb : va;
    hipLaunchKernelGGL(gather_sorted_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, order, sorted);
    // The result (exact mean squared distance to the 3 nearest other points) does not depend on the box size; the reference
    // uses 1024 (simple_knn.cu:12).  Smaller boxes have tighter AABBs and far fewer candidates per query; groups of `group`
    // consecutive boxes get a union AABB so the per-query box loop stays short (two-level cull).  500k points: 8.0 ms with
    // one level of 1024-point boxes, 2.4 ms with 64-point boxes in groups of 32; 2M points: 23.3 -> 8.2 ms.
#define GSR_KNN_LAUNCH(B)                                                                                                   \
    {                                                                                                                       \
        const uint32_t nb = (uint32_t)((P + (B) - 1) / (B));                                                                \
        hipLaunchKernelGGL(box_minmax_kernel<B>, dim3(nb), dim3(256), 0, s, P, sorted, boxes);                              \
        hipLaunchKernelGGL(super_minmax_kernel, dim3((nb / group + 256) / 256), dim3(256), 0, s, nb, group, boxes, supers);  \
        hipLaunchKernelGGL(knn_kernel<B>, dim3((P + 255) / 256), dim3(256), 0, s, P, nb, group, sorted, boxes, supers, out, evals); \
    }
    const int group = max(1, gsr_variant("KNN_GROUP", 32));
    float* supers = (float*)(base + L.supers);
    if (gsr_variant("KNN_WAVE", 1)) {  // wave-autonomous kernel: 64-point boxes, no workgroup barriers
        const uint32_t nb = (uint32_t)((P + 63) / 64);
        hipLaunchKernelGGL(box_minmax_kernel<64>, dim3(nb), dim3(256), 0, s, P, sorted, boxes);
        hipLaunchKernelGGL(super_minmax_kernel, dim3((nb / group + 256) / 256), dim3(256), 0, s, nb, group, boxes, supers);
        float* subs = (float*)(base + L.subs);
        hipLaunchKernelGGL(sub_minmax_kernel, dim3((4 * nb + 255) / 256), dim3(256), 0, s, P, sorted, subs);
        const uint32_t ns = (nb + (uint32_t)group - 1) / (uint32_t)group;
        float* tops = (float*)(base + L.tops);
        hipLaunchKernelGGL(super_minmax_kernel, dim3((ns / 32 + 256) / 256), dim3(256), 0, s, ns, 32, supers, tops);
        hipLaunchKernelGGL(knn_wave_kernel, dim3((nb + 3) / 4), dim3(256), 0, s, P, nb, group, sorted, boxes, supers, subs, tops, out, evals);
        return 0;
    }
    switch (gsr_variant("KNN_BOX", 64)) {
        case 1024: GSR_KNN_LAUNCH(1024); break;
        case 512: GSR_KNN_LAUNCH(512); break;
        case 256: GSR_KNN_LAUNCH(256); break;
        case 128: GSR_KNN_LAUNCH(128); break;
        default: GSR_KNN_LAUNCH(64); break;
    }
#undef GSR_KNN_LAUNCH
    return 0;
}
