// Tile binning for MI355X: produces, for every 16x16 tile, the list of Gaussian ids that touch it sorted by
// (depth bits, Gaussian id) -- bit-identical to what the reference obtains with one global stable radix sort of
// 64-bit (tile | depth) keys over all R instances (duplicateWithKeys + cub::DeviceRadixSort::SortPairs +
// identifyTileRanges, cuda_rasterizer/rasterizer_impl.cu:70-138,287-318).
//
// A 45-bit LSD sort streams 24 B/instance six times through HBM.  Here the tile id never enters a sort:
//   1. tile_count:   each workgroup takes a slice of the Gaussians and histograms the tiles of their rectangles in LDS
//                    (one LDS atomic per instance), then stores its row of the [workgroup][tile] table;
//   2. tile_colscan: per tile, exclusive prefix over the workgroups (in place) + the tile's total;
//   3. tile_scatter: same slicing as (1); every workgroup derives the tile starts from the totals (the first slice stores them
//                    as the reference's `ranges`), then an LDS cursor per tile hands out slots and the (depth bits, id) pairs
//                    go straight into their tile's segment (8 B/instance, written once);
//   4. tile_sort:    every tile's segment is sorted by (depth, id) in LDS -- a bucket sort whose entries rank themselves inside
//                    their depth bucket; ONE wave per list of fewer than 512 entries, a 256-thread workgroup up to 2048, a
//                    1024-thread workgroup up to 8192, a normalized bitonic network (all comparators ascending, tail padded
//                    virtually) beyond that and when a bucket overflows -- and written out as point_list (+ keys); the same
//                    launch orders the tiles by list length for the blend kernels.
// The default path runs (1)-(3) per band of tile rows (see "Banded variant" below), fed by emit_partition_kernel.
// The order in which (4) fills a segment is arbitrary (LDS atomics), but ids are unique within a tile, so the sorted
// result is unique: ascending depth bits, ties by ascending id = the stable sort's emission order.
// Integer/byte work, HBM + LDS bound; no MFMA.
#include "common.h"

namespace {

constexpr int TCHUNK = 16384;       // tiles histogrammed per pass (64 KB of LDS counters)
constexpr int BUCKET_CAP = 2048;      // lists sorted by a 256-thread workgroup of tile_sort_small_kernel
constexpr int BUCKET_CAP_TINY = 512;  // lists of fewer entries than this are sorted by a single wave of it
constexpr int BUCKET_CAP_MID = 8192;  // lists sorted by a 1024-thread workgroup (tile_sort_mid_kernel)

// The valid lanes of the wave that hold the same 8-bit class as this lane (meaningless for an invalid lane).  Neighbouring tiles
// have similar list lengths, so a plain LDS atomic per tile on its length-class counter serialises on a few hot counters: the
// first lane of each group adds the group's size instead.
__device__ __forceinline__ unsigned long long class_peers(uint32_t cls, bool valid)
{
    const unsigned long long v = __ballot(valid);
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        const uint32_t set = (cls >> bit) & 1u;
        const unsigned long long bal = __ballot(set != 0u);
        const uint32_t keep = set - 1u;  // 0 if this lane has the bit (keep the lanes of `bal`), ~0 if not (keep the others)
        lo &= (uint32_t)bal ^ keep;
        hi &= (uint32_t)(bal >> 32) ^ keep;
    }
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ void class_count(uint32_t total, bool valid, uint32_t* s_cls)
{
    const uint32_t cls = min(total >> 3, 255u);  // length class: 8 entries per class up to 2040, everything longer in the top class
    const unsigned long long peers = class_peers(cls, valid);
    if (valid && (threadIdx.x & 63) == (uint32_t)__builtin_ctzll(peers)) atomicAdd(&s_cls[cls], (uint32_t)__builtin_popcountll(peers));
}

// Lists of BUCKET_CAP_TINY..BUCKET_CAP entries ("long" lists: a 256-thread workgroup sorts them, not a wave) and longer ones ("mid":
// a 1024-thread workgroup) are announced to the sort kernels by the scatter kernels, which hold the tile totals: one segment of
// tile ids per band behind order[] (common.h GSR_LONG_* / GSR_MID_*), no global atomics.  list_append: aggregated append of
// this lane's tile to the workgroup's list, one LDS atomic per wave.
__device__ __forceinline__ void list_append(bool is_long, uint32_t tile, uint32_t* s_count, uint32_t* __restrict__ list)
{
    const unsigned long long m = __ballot(is_long);
    if (m == 0) return;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(s_count, (uint32_t)__builtin_popcountll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
    if (is_long) list[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = tile;
}

// Timing ablations of the scatter (results WRONG on purpose, never defined in the shipped build): make EXTRA=-DGSR_BIN_ABLATE=n
//   1: the (depth, id) pairs are not stored   2: no instance is processed at all (prologue: tile starts, ranges, long-list announcements)
#ifndef GSR_BIN_ABLATE
#define GSR_BIN_ABLATE 0
#endif

// What the host learns about a frame WITHOUT waiting for it (two words of host-mapped pinned memory, read at the start of a later
// call): hint[0] = number of the last frame whose scatter ran, hint[1] = number of the last frame that had a list of more than
// BUCKET_CAP entries.  A frame whose recent predecessors had none does not launch tile_sort_mid_kernel (an empty launch still
// costs its ~5 us boundary); should such a list turn up anyway, tile_sort_small_kernel sorts it with the slow in-place network.
__device__ __forceinline__ void mid_hint_publish(uint32_t* hint, uint32_t seq, bool first_band, bool has_mid)
{
    if (!hint) return;
    if (first_band) __hip_atomic_store(hint, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (has_mid) __hip_atomic_store(hint + 1, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct SliceRect {
    uint32_t mnx, mny, mxx, mxy, dbits;
    unsigned long long mask;  // reachable tiles of the rectangle (bit = row-major position); ~0 = all (no culling, or > 64 tiles)
    bool visible;
    // is tile (x, y) of the rectangle listed?  With `cull` the instances whose tile the splat's alpha >= 1/255 ellipse cannot
    // reach (gsr_reach_mask, computed by the preprocess kernel) never enter a list: they could not contribute to any pixel.
    __device__ __forceinline__ bool listed(uint32_t x, uint32_t y) const
    {
        return mask == ~0ull || ((mask >> ((y - mny) * (mxx - mnx) + (x - mnx))) & 1ull);
    }
};

__device__ __forceinline__ SliceRect load_rect(const float* __restrict__ splat, int idx, int gx, int gy, int cull)
{
    SliceRect r;
    const float4* rec = reinterpret_cast<const float4*>(splat + (size_t)idx * GSR_SPLAT_F);
    const float4 q0 = rec[0], q2 = rec[2];
    const int radius = __float_as_int(q2.z);
    r.visible = radius > 0;
    r.dbits = __float_as_uint(q2.y);
    gsr_get_rect(q0.x, q0.y, radius, gx, gy, r.mnx, r.mny, r.mxx, r.mxy);
    r.mask = ~0ull;
    if (cull && r.visible && (r.mxx - r.mnx) * (r.mxy - r.mny) < 64u) {  // (a full 64-tile mask is ~0 = "all" anyway)
        const float4 q3 = rec[3];
        if (__float_as_uint(q3.w) == 0u) {
            r.mask = ((unsigned long long)__float_as_uint(q3.z) << 32) | __float_as_uint(q3.y);
            // a mask of zero leaves `listed` false everywhere; keep ~0 reserved for "all"
        }
    }
    return r;
}

// (1) per-slice tile histogram.  grid = nslices, block = SLICE_THREADS, dynamic LDS = min(T, TCHUNK) * 4 bytes.
// 1024 threads per workgroup: with ~500 slices there are only ~2 workgroups per CU, and each instance is a dependent
// LDS-atomic -> global-store chain, so the latency is hidden by waves, not by work per thread.
constexpr int SLICE_THREADS = 1024;
__global__ void __launch_bounds__(SLICE_THREADS) tile_count_kernel(int P, int gpb, int gx, int gy, int T, int cull, const float* __restrict__ splat,
                                                         uint32_t* __restrict__ table)
{
    extern __shared__ uint32_t s_hist[];
    const int first = blockIdx.x * gpb, last = min(P, first + gpb);
    for (int tbase = 0; tbase < T; tbase += TCHUNK) {
        const int tn = min(TCHUNK, T - tbase);
        for (int t = threadIdx.x; t < tn; t += SLICE_THREADS) s_hist[t] = 0;
        __syncthreads();
        for (int idx = first + threadIdx.x; idx < last; idx += SLICE_THREADS) {
            const SliceRect r = load_rect(splat, idx, gx, gy, cull);
            if (!r.visible) continue;
            for (uint32_t y = r.mny; y < r.mxy; y++)
                for (uint32_t x = r.mnx; x < r.mxx; x++) {
                    const int t = (int)(y * (uint32_t)gx + x) - tbase;
                    if (t >= 0 && t < tn && r.listed(x, y)) atomicAdd(&s_hist[t], 1u);
                }
        }
        __syncthreads();
        uint32_t* row = table + (size_t)blockIdx.x * T + tbase;
        for (int t = threadIdx.x; t < tn; t += SLICE_THREADS) row[t] = s_hist[t];
        __syncthreads();
    }
}

// (2) per tile: exclusive prefix over the slices (in place) and the tile total.  A workgroup owns 32 adjacent tiles
// (128-byte rows of the table) and splits the slices into 8 groups, so the walk over the slices is 8x shorter than one
// thread per tile and the grid (T/32 workgroups) still fills the chip.
__global__ void __launch_bounds__(256) tile_colscan_kernel(int T, int nslices, uint32_t* __restrict__ table, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t part[8][32];
    const int tl = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int t = blockIdx.x * 32 + tl;
    const int per = (nslices + 7) / 8;
    const int b0 = sg * per, b1 = min(nslices, b0 + per);
    uint32_t sum = 0;
    if (t < T) {
#pragma unroll 8
        for (int b = b0; b < b1; b++) sum += table[(size_t)b * T + t];
    }
    part[sg][tl] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int g = 0; g < sg; g++) run += part[g][tl];
    if (t < T) {
#pragma unroll 8
        for (int b = b0; b < b1; b++) {
            const size_t i = (size_t)b * T + t;
            const uint32_t c = table[i];
            table[i] = run;
            run += c;
        }
        if (sg == 7) totals[t] = run;  // the last group ends on the tile total (empty groups pass `run` through)
    }
}

// (3)+(4) scatter (depth bits, id) into the tile segments.  Same slicing and LDS size as (1).
// Every workgroup first turns the tile totals into the tiles' start offsets itself (an exclusive scan of T numbers costs a
// 1024-thread workgroup about a microsecond; a separate single-workgroup kernel for it was 11 us of serialised latency on the
// critical path).  Workgroup 0 also stores the reference's ranges[] (uint2 per tile; empty tiles stay (0,0) like
// identifyTileRanges after its memset); workgroup 1 also announces the long lists to the sort kernels (list_append).
__global__ void __launch_bounds__(SLICE_THREADS) tile_scatter_kernel(int P, int gpb, int gx, int gy, int T, int cull, const float* __restrict__ splat,
                                                           const uint32_t* __restrict__ table, const uint32_t* __restrict__ totals,
                                                           uint32_t* __restrict__ ranges, uint32_t* __restrict__ order,
                                                           uint2* __restrict__ bins, uint32_t* hint, uint32_t hint_seq)
{
    extern __shared__ uint32_t s_cur[];
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t s_nlong[2], s_cls[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == (gridDim.x > 1 ? 1u : 0u)) {  // the long lists (one segment for the whole image here)
        if (threadIdx.x < 2) s_nlong[threadIdx.x] = 0;
        if (threadIdx.x < 256) s_cls[threadIdx.x] = 0;
        __syncthreads();
        for (int base = 0; base < T; base += SLICE_THREADS) {
            const int i = base + (int)threadIdx.x;
            const uint32_t n = i < T ? totals[i] : 0u;
            class_count(n, i < T, s_cls);
            list_append(n >= (uint32_t)BUCKET_CAP_TINY && n <= (uint32_t)BUCKET_CAP, (uint32_t)i, &s_nlong[0], order + GSR_LONG_LIST(T));
            list_append(n > (uint32_t)BUCKET_CAP, (uint32_t)i, &s_nlong[1], order + GSR_MID_LIST(T));
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            order[GSR_LONG_COUNT(T)] = s_nlong[0]; order[GSR_MID_COUNT(T)] = s_nlong[1];
            mid_hint_publish(hint, hint_seq, true, s_nlong[1] != 0u);
        }
        if (threadIdx.x < 256) order[GSR_CLASS_HIST(T) + threadIdx.x] = s_cls[threadIdx.x];
    }
    const int first = blockIdx.x * gpb, last = min(P, first + gpb);
    uint32_t chunk_base = 0;  // instances in the tiles before this chunk
    constexpr int EPT = TCHUNK / SLICE_THREADS;  // tiles per thread in the scan
    for (int tbase = 0; tbase < T; tbase += TCHUNK) {
        const int tn = min(TCHUNK, T - tbase);
        const uint32_t* row = table + (size_t)blockIdx.x * T + tbase;
        {   // cursor of tile t = start of the tile (exclusive scan of the totals) + this slice's offset inside it
            const int i0 = threadIdx.x * EPT;
            uint32_t v[EPT], tsum = 0;
#pragma unroll
            for (int e = 0; e < EPT; e++) { v[e] = i0 + e < tn ? totals[tbase + i0 + e] : 0u; tsum += v[e]; }
            uint32_t incl = tsum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t u = __shfl_up(incl, off, 64);
                if (lane >= off) incl += u;
            }
            if (lane == 63) wtot[wave] = incl;
            __syncthreads();
            uint32_t wbase = 0, chunk_total = 0;
#pragma unroll
            for (int w = 0; w < SLICE_THREADS / 64; w++) { if (w < wave) wbase += wtot[w]; chunk_total += wtot[w]; }
            uint32_t st = chunk_base + wbase + incl - tsum;
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int i = i0 + e;
                if (i < tn) {
                    s_cur[i] = st + row[i];
                    if (blockIdx.x == 0) {
                        ranges[2 * (tbase + i)] = v[e] ? st : 0u;
                        ranges[2 * (tbase + i) + 1] = v[e] ? st + v[e] : 0u;
                    }
                }
                st += v[e];
            }
            chunk_base += chunk_total;
        }
        __syncthreads();
        for (int idx = first + threadIdx.x; idx < last; idx += SLICE_THREADS) {
            const SliceRect r = load_rect(splat, idx, gx, gy, cull);
            if (!r.visible) continue;
            for (uint32_t y = r.mny; y < r.mxy; y++)
                for (uint32_t x = r.mnx; x < r.mxx; x++) {
                    const int t = (int)(y * (uint32_t)gx + x) - tbase;
                    if (t >= 0 && t < tn && r.listed(x, y)) {
                        const uint32_t slot = atomicAdd(&s_cur[t], 1u);
                        bins[slot] = make_uint2((uint32_t)idx, r.dbits);  // little-endian u64 = depth << 32 | id
                    }
                }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Banded variant of (1)-(3).  With slices that cover the whole image, a slice contributes only ~7 entries to any one tile:
// the scatter then writes 8-byte pairs into ~60-byte runs spread over all tile segments, every store a partial line
// (measured: 244 MB of write traffic for 30 MB of pairs).  Here the tile rows are cut into NB horizontal bands and the
// visible Gaussians are first partitioned into per-band id lists (emit_partition_kernel; a Gaussian whose rectangle crosses
// a band boundary is listed in each band it touches).  A slice then works on ONE band: its runs per tile are NB times longer, the
// [slice][tile] table shrinks NB-fold (band tiles only), and the LDS histograms cover a band instead of the image.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BAND_SLICES = 64;   // slices per band
constexpr int NB_MAX = GSR_BAND_COUNTERS;  // bands

struct BandGeom { int gx, gy, T, nb, bh, tmax, cull; };  // bh = tile rows per band, tmax = bh * gx; cull: see SliceRect::listed

// Instance / gradient-row offsets of every Gaussian (emit_keys_kernel of preprocess.hip without the keys: point_offsets, row_offsets
// and the two offsets kept in the splat record for the backward pass) AND the per-band id lists, in one pass over the records --
// the two used to be separate launches that each read all of them.
// PART_ITEMS Gaussians per thread (measured on 500k Gaussians: 1 -> 15.7 us, 2 -> 16.0 us, 4 -> 17.6 us; the workgroup's returning
// global atomics on the band counters favour few workgroups, filling 256 CUs many).  Each 256-Gaussian quarter of an item is one
// block of the scanned block offsets (scan_blocks_kernel works on 256-Gaussian blocks).
constexpr int PART_ITEMS = 2;
__global__ void __launch_bounds__(SLICE_THREADS) emit_partition_kernel(int P, BandGeom G, float* __restrict__ splat,
                                                                        const uint32_t* __restrict__ tiles_touched,
                                                                        const uint32_t* __restrict__ block_sums,
                                                                        const uint32_t* __restrict__ row_block_sums,
                                                                        uint32_t* __restrict__ total_out, uint32_t* __restrict__ row_total_out,
                                                                        unsigned long long* mailbox, uint32_t seq,
                                                                        uint32_t* __restrict__ point_offsets,
                                                                        uint32_t* __restrict__ row_offsets,
                                                                        uint32_t* __restrict__ wg_first,
                                                                        uint32_t* __restrict__ band_count, uint32_t* __restrict__ band_list)
{
    __shared__ uint32_t cnt[NB_MAX], base[NB_MAX];
    __shared__ uint32_t wtot[PART_ITEMS][16], wtot_r[PART_ITEMS][16];
    __shared__ uint32_t s_red[4][16];
    if (threadIdx.x < NB_MAX) cnt[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Round 4: the block scan is done HERE -- every workgroup adds up the preprocess kernel's per-256-Gaussian sums (instances and
    // gradient rows) in front of its own Gaussians and all of them (a few KB out of L2, two loads per thread at 500k Gaussians), so the
    // single-workgroup scan_blocks_kernel, a launch of its own between preprocess and this kernel, is only needed where the host must
    // know R before this kernel can run (api.hip).  Workgroup 0 publishes R in the host's mailbox word and stores the two totals.
    uint32_t before = 0, before_r = 0, total = 0, total_r = 0;
    {
        const int nblk = (P + 255) >> 8, first_blk = (int)blockIdx.x * (PART_ITEMS * SLICE_THREADS / 256);
        for (int i = threadIdx.x; i < nblk; i += SLICE_THREADS) {
            const uint32_t v = block_sums[i], vr = row_block_sums[i];
            total += v; total_r += vr;
            if (i < first_blk) { before += v; before_r += vr; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            before += __shfl_xor(before, off, 64); before_r += __shfl_xor(before_r, off, 64);
            total += __shfl_xor(total, off, 64); total_r += __shfl_xor(total_r, off, 64);
        }
        if (lane == 0) { s_red[0][wave] = before; s_red[1][wave] = before_r; s_red[2][wave] = total; s_red[3][wave] = total_r; }
    }
    int b0[PART_ITEMS], b1[PART_ITEMS];
    uint32_t n[PART_ITEMS], nr[PART_ITEMS], incl[PART_ITEMS], incl_r[PART_ITEMS];
#pragma unroll
    for (int e = 0; e < PART_ITEMS; e++) {
        const int idx = (blockIdx.x * PART_ITEMS + e) * SLICE_THREADS + threadIdx.x;
        b0[e] = 0; b1[e] = -1;
        n[e] = idx < P ? tiles_touched[idx] : 0u;
        nr[e] = 0;
        if (n[e] != 0) {  // visible with a non-empty rectangle
            const float4* rec = reinterpret_cast<const float4*>(splat + (size_t)idx * GSR_SPLAT_F);
            const float4 q0 = rec[0], q2 = rec[2], q3 = rec[3];
            nr[e] = gsr_splat_rows(__float_as_uint(q3.y), __float_as_uint(q3.z), __float_as_uint(q3.w), n[e]);
            uint32_t mnx, mny, mxx, mxy;
            gsr_get_rect(q0.x, q0.y, __float_as_int(q2.z), G.gx, G.gy, mnx, mny, mxx, mxy);
            b0[e] = (int)mny / G.bh; b1[e] = (int)(mxy - 1) / G.bh;
        }
        uint32_t in = n[e], ir = nr[e];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(in, off, 64), tr = __shfl_up(ir, off, 64);
            if (lane >= off) { in += t; ir += tr; }
        }
        incl[e] = in; incl_r[e] = ir;
        if (lane == 63) { wtot[e][wave] = in; wtot_r[e][wave] = ir; }
    }
    __syncthreads();
    before = before_r = total = total_r = 0;
#pragma unroll
    for (int w = 0; w < SLICE_THREADS / 64; w++) { before += s_red[0][w]; before_r += s_red[1][w]; total += s_red[2][w]; total_r += s_red[3][w]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *total_out = total;
        *row_total_out = total_r;
        if (mailbox) __hip_atomic_store(mailbox, ((unsigned long long)seq << 32) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#pragma unroll
    for (int e = 0; e < PART_ITEMS; e++) {
        const int idx = (blockIdx.x * PART_ITEMS + e) * SLICE_THREADS + threadIdx.x;
        for (int b = b0[e]; b <= b1[e]; b++) atomicAdd(&cnt[b], 1u);
        if (idx < P) {
            // instances / rows before this wave: the workgroup's base + the items before this one + the waves before this one
            uint32_t bs = before, bs_r = before_r;
            for (int e2 = 0; e2 < e; e2++)
                for (int w = 0; w < SLICE_THREADS / 64; w++) { bs += wtot[e2][w]; bs_r += wtot_r[e2][w]; }
            for (int w = 0; w < wave; w++) { bs += wtot[e][w]; bs_r += wtot_r[e][w]; }
            const uint32_t end = bs + incl[e], end_r = bs_r + incl_r[e];
            point_offsets[idx] = end;
            row_offsets[idx] = end_r;
            gsr_part_store(wg_first, P, idx, end_r - nr[e], end_r, total_r);
            if (n[e] != 0) {
                float* rec = splat + (size_t)idx * GSR_SPLAT_F;
                rec[SP_ROWBASE] = __uint_as_float(end_r - nr[e]);
                rec[SP_OFFS] = __uint_as_float(end - n[e]);
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < G.nb) {
        const uint32_t c = cnt[threadIdx.x];
        base[threadIdx.x] = c ? atomicAdd(&band_count[threadIdx.x], c) : 0u;
        cnt[threadIdx.x] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PART_ITEMS; e++) {
        const int idx = (blockIdx.x * PART_ITEMS + e) * SLICE_THREADS + threadIdx.x;
        for (int b = b0[e]; b <= b1[e]; b++) band_list[(size_t)b * P + base[b] + atomicAdd(&cnt[b], 1u)] = (uint32_t)idx;
    }
}

__device__ __forceinline__ void band_slice_range(const uint32_t* __restrict__ band_count, int b, int sidx, uint32_t& first, uint32_t& last)
{
    const uint32_t n = band_count[b];
    const uint32_t per = (n + BAND_SLICES - 1) / BAND_SLICES;
    first = min(n, (uint32_t)sidx * per);
    last = min(n, first + per);
}

__global__ void __launch_bounds__(SLICE_THREADS) tile_count_banded_kernel(int P, BandGeom G, const float* __restrict__ splat,
                                                                           const uint32_t* __restrict__ band_count,
                                                                           const uint32_t* __restrict__ band_list, uint32_t* __restrict__ table)
{
    extern __shared__ uint32_t s_hist[];
    const int b = blockIdx.x / BAND_SLICES, sidx = blockIdx.x % BAND_SLICES;
    const uint32_t row0 = (uint32_t)(b * G.bh), row1 = min((uint32_t)G.gy, row0 + (uint32_t)G.bh);
    const int tn = (int)(row1 - row0) * G.gx;
    for (int t = threadIdx.x; t < tn; t += SLICE_THREADS) s_hist[t] = 0;
    __syncthreads();
    uint32_t first, last;
    band_slice_range(band_count, b, sidx, first, last);
    for (uint32_t i = first + threadIdx.x; i < last; i += SLICE_THREADS) {
        const SliceRect r = load_rect(splat, (int)band_list[(size_t)b * P + i], G.gx, G.gy, G.cull);
        const uint32_t y0 = max(r.mny, row0), y1 = min(r.mxy, row1);
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = r.mnx; x < r.mxx; x++)
                if (r.listed(x, y)) atomicAdd(&s_hist[(y - row0) * (uint32_t)G.gx + x], 1u);
    }
    __syncthreads();
    uint32_t* row = table + (size_t)blockIdx.x * G.tmax;
    for (int t = threadIdx.x; t < tn; t += SLICE_THREADS) row[t] = s_hist[t];
}

// per band and tile: exclusive prefix over the band's slices (in place) and the tile total; blockIdx.y = band
__global__ void __launch_bounds__(256) tile_colscan_banded_kernel(BandGeom G, uint32_t* __restrict__ table, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t part[8][32];
    const int b = blockIdx.y;
    const int row0 = b * G.bh, row1 = min(G.gy, row0 + G.bh);
    const int tn = (row1 - row0) * G.gx;
    const int tl = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int t = blockIdx.x * 32 + tl;
    if (blockIdx.x * 32 >= tn) return;
    constexpr int per = BAND_SLICES / 8;
    uint32_t* col = table + (size_t)b * BAND_SLICES * G.tmax + t;
    // Order of the slices' runs inside a tile segment.  A run is ~4 entries (35 bytes), so nearly every 64-byte piece of a segment is shared
    // by the runs of two or three slices.  GSR_COLSCAN_XCD (default): group sg takes the slices sg, sg + 8, sg + 16, ... -- the workgroups
    // of the scatter that run on ONE XCD (workgroup index mod 8; a band's first workgroup is a multiple of 64) -- so the slices that share
    // a piece write it through the same L2, which merges their partial writes; with consecutive slices per group (round 1-3) neighbours
    // sat on different XCDs and each L2 wrote its part of the piece back on its own (62 MiB written for 18 MB of pairs).
#ifndef GSR_COLSCAN_XCD
#define GSR_COLSCAN_XCD 1
#endif
    auto slice_of = [&](int k) { return GSR_COLSCAN_XCD ? sg + 8 * k : sg * per + k; };
    uint32_t sum = 0;
    if (t < tn) {
#pragma unroll
        for (int k = 0; k < per; k++) sum += col[(size_t)slice_of(k) * G.tmax];
    }
    part[sg][tl] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int g = 0; g < sg; g++) run += part[g][tl];
    if (t < tn) {
#pragma unroll
        for (int k = 0; k < per; k++) {
            uint32_t* e = col + (size_t)slice_of(k) * G.tmax;
            const uint32_t c = *e;
            *e = run;
            run += c;
        }
        if (sg == 7) totals[row0 * G.gx + t] = run;
    }
}

__global__ void __launch_bounds__(SLICE_THREADS) tile_scatter_banded_kernel(int P, BandGeom G, const float* __restrict__ splat,
                                                                             const uint32_t* __restrict__ band_count,
                                                                             const uint32_t* __restrict__ band_list,
                                                                             const uint32_t* __restrict__ table, const uint32_t* __restrict__ totals,
                                                                             uint32_t* __restrict__ ranges, uint32_t* __restrict__ order,
                                                                             uint2* __restrict__ bins, uint32_t* hint, uint32_t hint_seq)
{
    extern __shared__ uint32_t s_cur[];
    __shared__ uint32_t wtot[16], woth[16];
    __shared__ uint32_t s_nlong[2], s_cls[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int T = G.T;
    const int b = blockIdx.x / BAND_SLICES, sidx = blockIdx.x % BAND_SLICES;
    const uint32_t row0 = (uint32_t)(b * G.bh), row1 = min((uint32_t)G.gy, row0 + (uint32_t)G.bh);
    const int t0 = (int)row0 * G.gx, tn = (int)(row1 - row0) * G.gx;
    const uint32_t* row = table + (size_t)blockIdx.x * G.tmax;
    // this thread's first Gaussian: its loads (id -> record) are in flight during the prologue
    uint32_t first, last;
    band_slice_range(band_count, b, sidx, first, last);
    uint32_t i = first + threadIdx.x;
    bool have = i < last;
    uint32_t idx = have ? band_list[(size_t)b * P + i] : 0u;
    SliceRect r{};
    if (have) r = load_rect(splat, (int)idx, G.gx, G.gy, G.cull);
    // tile starts = instances in the bands before this one + exclusive scan of the band's own tile totals (one load round trip:
    // scanning all T totals in every workgroup, with the table row read behind the scan, cost 12 us of the kernel's 44);
    // slice 0 of the band stores them as the reference's ranges[] (empty tiles (0,0) like identifyTileRanges after its memset)
    // and announces the band's long lists to the sort kernels (list_append)
    uint32_t other = 0;
    for (int k = threadIdx.x; k < t0; k += SLICE_THREADS) other += totals[k];
    if (threadIdx.x < 2) s_nlong[threadIdx.x] = 0;
    if (sidx == 0 && threadIdx.x < 256) s_cls[threadIdx.x] = 0;
    uint32_t chunk_base = 0;
    for (int tbase = 0; tbase < tn; tbase += SLICE_THREADS) {
        const int lt = tbase + (int)threadIdx.x;
        const bool in = lt < tn;
        const uint32_t v = in ? totals[t0 + lt] : 0u, rw = in ? row[lt] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off, 64);
            if (lane >= off) incl += u;
        }
        if (tbase == 0) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) other += __shfl_xor(other, off, 64);
        }
        if (lane == 63) { wtot[wave] = incl; if (tbase == 0) woth[wave] = other; }
        __syncthreads();
        uint32_t wbase = 0, chunk_total = 0;
#pragma unroll
        for (int w = 0; w < SLICE_THREADS / 64; w++) {
            if (w < wave) wbase += wtot[w];
            chunk_total += wtot[w];
            if (tbase == 0) chunk_base += woth[w];
        }
        const uint32_t st = chunk_base + wbase + incl - v;
        if (in) {
            s_cur[lt] = st + rw;
            if (sidx == 0) {
                ranges[2 * (t0 + lt)] = v ? st : 0u;
                ranges[2 * (t0 + lt) + 1] = v ? st + v : 0u;
            }
        }
        if (sidx == 0) {
            class_count(v, in, s_cls);
            list_append(v >= (uint32_t)BUCKET_CAP_TINY && v <= (uint32_t)BUCKET_CAP, (uint32_t)(t0 + lt), &s_nlong[0], order + GSR_LONG_LIST(T) + t0);
            list_append(v > (uint32_t)BUCKET_CAP, (uint32_t)(t0 + lt), &s_nlong[1], order + GSR_MID_LIST(T) + t0);
        }
        chunk_base += chunk_total;
        __syncthreads();
    }
    if (sidx == 0 && threadIdx.x == 0) {
        order[GSR_LONG_COUNT(T) + b] = s_nlong[0]; order[GSR_MID_COUNT(T) + b] = s_nlong[1];
        mid_hint_publish(hint, hint_seq, b == 0, s_nlong[1] != 0u);
    }
    if (sidx == 0 && threadIdx.x < 256) order[GSR_CLASS_HIST(T) + b * 256 + threadIdx.x] = s_cls[threadIdx.x];
    if (GSR_BIN_ABLATE >= 2 && !(have && r.dbits == 0x12345678u)) have = false;  // (keeps the first loads alive)
    while (have) {
        const uint32_t y0 = max(r.mny, row0), y1 = min(r.mxy, row1);
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = r.mnx; x < r.mxx; x++) {
                if (!r.listed(x, y)) continue;
                const uint32_t slot = atomicAdd(&s_cur[(y - row0) * (uint32_t)G.gx + x], 1u);
                if (GSR_BIN_ABLATE != 1 || slot == 0xFFFFFFFFu) bins[slot] = make_uint2(idx, r.dbits);  // little-endian u64 = depth << 32 | id
            }
        i += SLICE_THREADS;
        have = i < last && GSR_BIN_ABLATE < 2;
        if (have) {
            idx = band_list[(size_t)b * P + i];
            r = load_rect(splat, (int)idx, G.gx, G.gy, G.cull);
        }
    }
}

// Normalized bitonic network on n keys held in `a` (LDS or HBM), virtually padded with +inf up to N = pow2 >= n.
// Every comparator is ascending (lo < hi), so a comparator whose upper index is >= n is a no-op.
// Comparator i of a stage is handled by wave (i / (N/2 / waves)): each of the `waves` active waves owns one contiguous
// segment of S = N / waves positions, so every stage whose comparators stay inside a segment (stride 2j <= S, mirror
// block k <= S) needs no workgroup barrier -- a wave's LDS operations execute in order.  Only the top log2(waves)
// merge levels cross segments: 3 barrier stages for any N instead of log2(N)*(log2(N)+1)/2.
template <bool LOCAL_OK, typename Ptr>
__device__ __forceinline__ void bitonic_sort_block(Ptr a, uint32_t n, uint32_t N)
{
    const uint32_t half = N >> 1;
    const uint32_t waves = half >= 256 ? 4u : (half >= 64 ? half / 64 : 1u);   // active waves
    const uint32_t per_wave = half / waves;                                     // comparators per wave and stage
    const uint32_t S = N / waves;                                               // positions per wave segment
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool active = wave < waves;
    for (uint32_t k = 2, lk = 1; k <= N; k <<= 1, lk++) {  // k = 1 << lk (strides are powers of two: shifts, no division)
        const uint32_t hk = k >> 1;
        if (active)
            for (uint32_t c = lane; c < per_wave; c += 64) {  // mirror step
                const uint32_t i = wave * per_wave + c;
                const uint32_t blk = i >> (lk - 1), off = i & (hk - 1);
                const uint32_t lo = (blk << lk) + off, hi = (blk << lk) + k - 1 - off;
                if (hi < n) {
                    const unsigned long long x = a[lo], y = a[hi];
                    if (x > y) { a[lo] = y; a[hi] = x; }
                }
            }
        if (LOCAL_OK && k <= S) __builtin_amdgcn_wave_barrier(); else __syncthreads();
        for (uint32_t j = k >> 2, lj = lk - 2; j >= 1; j >>= 1, lj--) {  // j = 1 << lj
            if (active)
                for (uint32_t c = lane; c < per_wave; c += 64) {
                    const uint32_t i = wave * per_wave + c;
                    const uint32_t lo = ((i >> lj) << (lj + 1)) + (i & (j - 1)), hi = lo + j;
                    if (hi < n) {
                        const unsigned long long x = a[lo], y = a[hi];
                        if (x > y) { a[lo] = y; a[hi] = x; }
                    }
                }
            // the NEXT stage (stride j/2, or the next merge's mirror) decides which sync is needed before it
            const uint32_t next_span = j > 1 ? j : (k << 1);  // 2*(j/2) = j, or the next mirror block size
            if (LOCAL_OK && next_span <= S && 2 * j <= S) __builtin_amdgcn_wave_barrier(); else __syncthreads();
        }
    }
    __syncthreads();
}

// The same network driven by one wave alone (tile_sort_tiny_kernel's degenerate lists): a wave's LDS operations execute in
// order, so no barrier is needed between stages.
__device__ __forceinline__ void bitonic_sort_wave(unsigned long long* a, uint32_t n, uint32_t N)
{
    const uint32_t half = N >> 1, lane = threadIdx.x & 63;
    for (uint32_t k = 2, lk = 1; k <= N; k <<= 1, lk++) {
        const uint32_t hk = k >> 1;
        for (uint32_t i = lane; i < half; i += 64) {
            const uint32_t blk = i >> (lk - 1), off = i & (hk - 1);
            const uint32_t lo = (blk << lk) + off, hi = (blk << lk) + k - 1 - off;
            if (hi < n) {
                const unsigned long long x = a[lo], y = a[hi];
                if (x > y) { a[lo] = y; a[hi] = x; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j = k >> 2, lj = lk - 2; j >= 1; j >>= 1, lj--) {
            for (uint32_t i = lane; i < half; i += 64) {
                const uint32_t lo = ((i >> lj) << (lj + 1)) + (i & (j - 1)), hi = lo + j;
                if (hi < n) {
                    const unsigned long long x = a[lo], y = a[hi];
                    if (x > y) { a[lo] = y; a[hi] = x; }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Bucket sort of one tile's list in LDS (by a workgroup, or by one wave: THREADS == 64).
// Depth bits are mapped monotonically onto N buckets (N = pow2 >= n, so ~1 entry per bucket), a counting sort places
// the keys bucket by bucket, and every entry then ranks itself against the few keys of its own bucket on the full 64-bit
// (depth, id) key and is written straight to its final position.  ~10 LDS operations per entry instead of the ~100 of a
// bitonic network (which is what limited the first version: it was LDS-bandwidth bound).  A bucket that grows past
// BUCKET_LIMIT (many equal depths) sends the tile to the bitonic path, so the worst case stays n log^2 n.
constexpr int BUCKET_LIMIT = 16;
template <int CAP, int THREADS>
struct BucketSortLds {
    unsigned long long keys[CAP];
    uint32_t cnt[CAP + 1];
    uint32_t red[2 * THREADS / 64];
    uint32_t wsum[THREADS / 64];
    int overflow;
};

// BUCKET_EPT = entries (and buckets) per thread: 2, 4 or 8 for lists of up to 2, 4 or 8 x THREADS entries
// barrier of the group that sorts one list: the workgroup, or a single wave (whose LDS operations execute in order: only the
// compiler has to be kept from moving them)
template <int THREADS>
__device__ __forceinline__ void bucket_sync()
{
    if constexpr (THREADS == 64) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

template <int BUCKET_EPT, int THREADS, int CAP>
__device__ __forceinline__ void bucket_sort_tile(BucketSortLds<CAP, THREADS>& L, uint32_t tile, uint32_t r0, uint32_t n, uint32_t N,
                                                 const unsigned long long* __restrict__ bins, uint32_t* __restrict__ point_list,
                                                 unsigned long long* __restrict__ keys_sorted)
{
    unsigned long long* s_keys = L.keys;
    uint32_t* s_cnt = L.cnt;
    uint32_t* s_red = L.red;
    uint32_t* s_wsum = L.wsum;
    int& s_overflow = L.overflow;
    // THREADS == 64: one wave of a larger workgroup sorts the list on its own (tid = lane, barriers at wave scope)
    const int tid = THREADS == 64 ? (int)(threadIdx.x & 63) : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long* seg = bins + r0;
    unsigned long long key[BUCKET_EPT];
    uint32_t dmin = 0xFFFFFFFFu, dmax = 0u;
#pragma unroll
    for (int e = 0; e < BUCKET_EPT; e++) {
        const uint32_t i = tid + THREADS * e;
        key[e] = i < n ? seg[i] : ~0ull;
        if (i < n) { const uint32_t d = (uint32_t)(key[e] >> 32); dmin = min(dmin, d); dmax = max(dmax, d); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, off, 64));
        dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, off, 64));
    }
    constexpr int WAVES = THREADS / 64;
    if (lane == 0) { s_red[wave] = dmin; s_red[WAVES + wave] = dmax; }
    for (uint32_t b = tid; b <= N; b += THREADS) s_cnt[b] = 0;
    if (tid == 0) s_overflow = 0;
    bucket_sync<THREADS>();
#pragma unroll
    for (int w = 0; w < WAVES; w++) { dmin = min(dmin, s_red[w]); dmax = max(dmax, s_red[WAVES + w]); }
    // monotone map depth bits -> bucket (float rounding is monotone, so equal or larger depth never gets a smaller bucket)
    const float inv = (float)N / ((float)(dmax - dmin) + 1.0f);
    uint32_t bucket[BUCKET_EPT];
#pragma unroll
    for (int e = 0; e < BUCKET_EPT; e++) {
        const uint32_t i = tid + THREADS * e;
        bucket[e] = 0;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[e] >> 32);
            bucket[e] = min(N - 1, (uint32_t)((float)(d - dmin) * inv));
            atomicAdd(&s_cnt[bucket[e]], 1u);
        }
    }
    bucket_sync<THREADS>();
    // exclusive scan of the N bucket counts: thread t owns buckets [t*EPT, t*EPT + EPT)
    uint32_t c[BUCKET_EPT], tsum = 0;
#pragma unroll
    for (int e = 0; e < BUCKET_EPT; e++) {
        const uint32_t b = tid * BUCKET_EPT + e;
        c[e] = b < N ? s_cnt[b] : 0u;
        tsum += c[e];
    }
    uint32_t incl = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) s_wsum[wave] = incl;
    bucket_sync<THREADS>();
    uint32_t start = incl - tsum;
    for (int w = 0; w < wave; w++) start += s_wsum[w];
    bool big = false;
#pragma unroll
    for (int e = 0; e < BUCKET_EPT; e++) {
        const uint32_t b = tid * BUCKET_EPT + e;
        if (b < N) s_cnt[b] = start;  // becomes the bucket's write cursor
        start += c[e];
        big |= c[e] > BUCKET_LIMIT;
    }
    if (big) s_overflow = 1;
    bucket_sync<THREADS>();
    if (s_overflow) {
        // degenerate depth distribution: sort the whole list with the bitonic network instead
        bucket_sync<THREADS>();
#pragma unroll
        for (int e = 0; e < BUCKET_EPT; e++) {
            const uint32_t i = tid + THREADS * e;
            if (i < n) s_keys[i] = key[e];
        }
        bucket_sync<THREADS>();
        if (n > 1) {
            if constexpr (THREADS == 64) bitonic_sort_wave(s_keys, n, N);
            else bitonic_sort_block<true>(s_keys, n, N);
        }
        for (uint32_t i = tid; i < n; i += THREADS) {
            const unsigned long long k = s_keys[i];
            point_list[r0 + i] = (uint32_t)k;
            if (keys_sorted) keys_sorted[r0 + i] = ((unsigned long long)tile << 32) | (k >> 32);
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < BUCKET_EPT; e++) {
        const uint32_t i = tid + THREADS * e;
        if (i < n) s_keys[atomicAdd(&s_cnt[bucket[e]], 1u)] = key[e];
    }
    bucket_sync<THREADS>();
    // every entry ranks itself inside its bucket (the cursors now hold the bucket ends; keys are unique, the id is part of
    // them) and goes straight to its place in the list: balanced over the entries, no dependent chain of LDS round trips as
    // in a per-thread insertion sort, and no second pass over the sorted keys
#pragma unroll
    for (int e = 0; e < BUCKET_EPT; e++) {
        const uint32_t i = tid + THREADS * e;
        if (i < n) {
            const uint32_t b = bucket[e];
            const uint32_t st = b ? s_cnt[b - 1] : 0u, en = s_cnt[b];
            uint32_t rank = st;
            for (uint32_t j = st; j < en; j++) rank += s_keys[j] < key[e] ? 1u : 0u;
            point_list[r0 + rank] = (uint32_t)key[e];
            if (keys_sorted) keys_sorted[r0 + rank] = ((unsigned long long)tile << 32) | (key[e] >> 32);
        }
    }
}

// Launch order of the blend kernels: order[] = the tile ids by descending list length (256 length classes of 8 entries,
// everything from 2040 up in the top class; counting sort): the blend kernels take workgroup b -> tile order[b], so the long
// lists start first and the short ones fill the tail of the launch (longest-processing-time-first).  Which tile a workgroup
// renders has no effect on any result.  The counting half is done by the scatter kernels (class_count, per band), the placing
// half by the first workgroups of tile_sort_small_kernel, one per band, beside the sorting.  Neighbouring tiles have similar
// lengths, so a wave first groups its lanes by class (8 ballots) and only the first lane of a group touches the LDS counter:
// a plain atomic per tile serialises on a few hot counters.
// One 256-thread workgroup per band places the band's tiles [t0, t0 + tn): the per-band class counts (scatter kernels) give the
// start of every class in order[] and the number of tiles the bands before this one put into it.
__device__ __forceinline__ void emit_band_order(int T, int nbl, int band, int t0, int tn, const uint32_t* __restrict__ ranges,
                                                uint32_t* __restrict__ order, uint32_t* cls_cur)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint2* rg = reinterpret_cast<const uint2*>(ranges) + t0;
    constexpr int CH = 4;  // tiles per thread whose ranges are loaded together (one round trip, then the ballots)
    uint2 q[CH];
#pragma unroll
    for (int e = 0; e < CH; e++) {
        const int i = e * 256 + (int)threadIdx.x;
        q[e] = i < tn ? rg[i] : make_uint2(0u, 0u);
    }
    uint32_t total = 0, before = 0;
    for (int b = 0; b < nbl; b++) {
        const uint32_t h = order[GSR_CLASS_HIST(T) + b * 256 + threadIdx.x];
        total += h;
        if (b < band) before += h;
    }
    cls_cur[threadIdx.x] = total;
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive scan from the top class down (one wave, 4 classes per lane)
        const int c0 = 255 - 4 * (int)threadIdx.x;
        const uint32_t n0 = cls_cur[c0], n1 = cls_cur[c0 - 1], n2 = cls_cur[c0 - 2], n3 = cls_cur[c0 - 3];
        const uint32_t tsum = n0 + n1 + n2 + n3;
        uint32_t incl = tsum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off, 64);
            if ((int)threadIdx.x >= off) incl += u;
        }
        const uint32_t b = incl - tsum;
        cls_cur[c0] = b; cls_cur[c0 - 1] = b + n0; cls_cur[c0 - 2] = b + n0 + n1; cls_cur[c0 - 3] = b + n0 + n1 + n2;
    }
    __syncthreads();
    cls_cur[threadIdx.x] += before;
    __syncthreads();
    for (int base = 0; base < tn; base += 256 * CH) {
        if (base) {
#pragma unroll
            for (int e = 0; e < CH; e++) {
                const int i = base + e * 256 + (int)threadIdx.x;
                q[e] = i < tn ? rg[i] : make_uint2(0u, 0u);
            }
        }
#pragma unroll
        for (int e = 0; e < CH; e++) {
            const int i = base + e * 256 + (int)threadIdx.x;
            if (base + e * 256 >= tn) break;  // uniform
            const bool valid = i < tn;
            const uint32_t cls = min((q[e].y - q[e].x) >> 3, 255u);
            const unsigned long long peers = class_peers(cls, valid);
            const int leader = valid ? __builtin_ctzll(peers) : (int)lane;
            uint32_t slot = 0;
            if (valid && (int)lane == leader) slot = atomicAdd(&cls_cur[cls], (uint32_t)__builtin_popcountll(peers));
            slot = (uint32_t)__shfl((int)slot, leader, 64);
            if (valid) order[slot + (uint32_t)__builtin_popcountll(peers & ((1ull << lane) - 1ull))] = (uint32_t)(t0 + i);
        }
    }
}

// k-th tile of the per-band lists taken together (counts[nbl], band segments `seg` tiles apart); false past the end
__device__ __forceinline__ bool listed_tile(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ lists, int nbl, int seg, uint32_t k,
                                            uint32_t& tile)
{
    for (int band = 0; band < nbl; band++) {
        const uint32_t c = counts[band];
        if (k < c) { tile = lists[(size_t)band * seg + k]; return true; }
        k -= c;
    }
    return false;
}

// (5a) lists of up to BUCKET_CAP entries, one launch, three kinds of workgroup:
//  * workgroups [0, nbl): emit_band_order;
//  * the next n_long: a 256-thread workgroup per list of BUCKET_CAP_TINY..BUCKET_CAP entries, striding over the long
//    lists of the bands (list_append; `nbl` segments `seg` tiles apart, n_long is a multiple of nbl);
//  * the others: ONE wave per list of fewer than BUCKET_CAP_TINY entries (tile = 4 * (workgroup - nbl - n_long) + wave) -- most
//    tiles of the 500k-Gaussian configuration (mean 280).  A whole workgroup per short list spent its time waiting (ranges ->
//    bins -> four workgroup barriers -> store); a wave per list keeps four times as many lists in flight and needs no barrier.
// The long lists are dispatched first and sort while the short ones stream through the rest of the chip.
union TileSortLds {
    BucketSortLds<BUCKET_CAP, 256> whole;
    BucketSortLds<BUCKET_CAP_TINY, 64> wave[4];
};

// (waves_per_eu: the in-place fallback for lists beyond BUCKET_CAP raised the allocation from 80 to 87 VGPRs -- 5 instead of 6 waves per
// SIMD, 19 -> 26 us for the whole kernel; pinned to 6 the allocator keeps the hot paths at 80 without spilling)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) tile_sort_small_kernel(int T, int n_long, int nbl, int seg, uint32_t* __restrict__ order,
                                                              const uint32_t* __restrict__ ranges,
                                                              const unsigned long long* bins,
                                                              uint32_t* __restrict__ point_list,
                                                              unsigned long long* __restrict__ keys_sorted, int mid_fallback)
{
    __shared__ TileSortLds L;
    if ((int)blockIdx.x < nbl) {
        const int t0 = (int)blockIdx.x * seg;
        emit_band_order(T, nbl, (int)blockIdx.x, t0, nbl == 1 ? T : min(seg, T - t0), ranges, order, L.whole.cnt);
        return;
    }
    if ((int)blockIdx.x < nbl + n_long) {
        for (uint32_t k = blockIdx.x - (uint32_t)nbl;; k += (uint32_t)n_long) {
            uint32_t tile;
            if (!listed_tile(order + GSR_LONG_COUNT(T), order + GSR_LONG_LIST(T), nbl, seg, k, tile)) break;
            const uint32_t r0 = __builtin_amdgcn_readfirstlane(ranges[2 * tile]);
            const uint32_t r1 = __builtin_amdgcn_readfirstlane(ranges[2 * tile + 1]);
            const uint32_t n = r1 - r0;
            uint32_t N = 1;
            while (N < n) N <<= 1;
            if (N <= 512) bucket_sort_tile<2, 256, BUCKET_CAP>(L.whole, tile, r0, n, N, bins, point_list, keys_sorted);
            else if (N <= 1024) bucket_sort_tile<4, 256, BUCKET_CAP>(L.whole, tile, r0, n, N, bins, point_list, keys_sorted);
            else bucket_sort_tile<8, 256, BUCKET_CAP>(L.whole, tile, r0, n, N, bins, point_list, keys_sorted);
            __syncthreads();  // the LDS image is reused by the next list
        }
        // tile_sort_mid_kernel was not launched (no list that long in the recent frames, see mid_hint_publish) and this frame has
        // one after all: the normalized bitonic network in place in HBM -- slow (milliseconds on a dense scene), one frame long
        if (mid_fallback)
            for (uint32_t k = blockIdx.x - (uint32_t)nbl;; k += (uint32_t)n_long) {
                uint32_t tile;
                if (!listed_tile(order + GSR_MID_COUNT(T), order + GSR_MID_LIST(T), nbl, seg, k, tile)) break;
                const uint32_t r0 = __builtin_amdgcn_readfirstlane(ranges[2 * tile]);
                const uint32_t r1 = __builtin_amdgcn_readfirstlane(ranges[2 * tile + 1]);
                const uint32_t n = r1 - r0;
                uint32_t N = 1;
                while (N < n) N <<= 1;
                unsigned long long* sg = const_cast<unsigned long long*>(bins) + r0;
                bitonic_sort_block<false>(sg, n, N);
                for (uint32_t q = threadIdx.x; q < n; q += 256) {
                    const unsigned long long key = sg[q];
                    point_list[r0 + q] = (uint32_t)key;
                    if (keys_sorted) keys_sorted[r0 + q] = ((unsigned long long)tile << 32) | (key >> 32);
                }
                __syncthreads();
            }
        return;
    }
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile = 4u * (blockIdx.x - (uint32_t)(nbl + n_long)) + wv;
    if (tile >= (uint32_t)T) return;
    const uint32_t r0 = __builtin_amdgcn_readfirstlane(ranges[2 * tile]);
    const uint32_t r1 = __builtin_amdgcn_readfirstlane(ranges[2 * tile + 1]);
    const uint32_t n = r1 - r0;
    if (n == 0 || n >= BUCKET_CAP_TINY) return;
    uint32_t N = 1;
    while (N < n) N <<= 1;
    if (N <= 128) bucket_sort_tile<2, 64, BUCKET_CAP_TINY>(L.wave[wv], tile, r0, n, N, bins, point_list, keys_sorted);
    else if (N <= 256) bucket_sort_tile<4, 64, BUCKET_CAP_TINY>(L.wave[wv], tile, r0, n, N, bins, point_list, keys_sorted);
    else bucket_sort_tile<8, 64, BUCKET_CAP_TINY>(L.wave[wv], tile, r0, n, N, bins, point_list, keys_sorted);
}

// (5a') lists of more than 2048 entries: up to 8192 (dense scenes: the 2M-Gaussian configuration has a mean list length of ~1900): the same
// bucket sort with 1024 threads and 96 KB of LDS; the bitonic network took 0.8 ms per frame there.
// A list longer than the LDS budget (n > BUCKET_CAP_MID; a dense or distant region behind one tile: every second tile of a 6 M-Gaussian
// scene at 1080p).  Two levels: the depth bits are mapped monotonically onto NB coarse buckets of ~1-2 k entries (counting sort through LDS
// counters: three reads of the list and one write, into `scratch` -- the second key array of the binning buffer, which this path does not
// use otherwise), then every coarse bucket is a list of its own for bucket_sort_tile, which ranks on the full (depth, id) key and writes
// its entries at their final positions; the buckets are in depth order, so their concatenation is the sorted list.  Rounds 1-3 ran the
// bitonic network in place in HBM here (four active waves, log^2 n passes: 9 ms for a list of 125 k entries, 4.6 of the 8 ms of a step on
// 6 M Gaussians); that network is now only the fallback for a coarse bucket that is itself too long for LDS (a run of equal depths).
// In KEYS mode keys_sorted IS scratch: a coarse bucket's sort loads all its keys before it stores any (bucket_sort_tile), and its stores
// stay inside its own segment.
__device__ __forceinline__ void giant_list_sort(BucketSortLds<BUCKET_CAP_MID, 1024>& L, uint32_t* s_start, uint32_t tile, uint32_t r0, uint32_t n,
                                                const unsigned long long* __restrict__ bins, unsigned long long* scratch,
                                                uint32_t* __restrict__ point_list, unsigned long long* keys_sorted)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long* seg = bins + r0;
    uint32_t dmin = 0xFFFFFFFFu, dmax = 0u;
    for (uint32_t i = tid; i < n; i += 1024) {
        const uint32_t d = (uint32_t)(seg[i] >> 32);
        dmin = min(dmin, d); dmax = max(dmax, d);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, off, 64));
        dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, off, 64));
    }
    uint32_t NB = 2;
    while (NB < (n + 2047u) / 2048u) NB <<= 1;
    if (NB > (uint32_t)BUCKET_CAP_MID) NB = (uint32_t)BUCKET_CAP_MID;
    if (lane == 0) { L.red[wave] = dmin; L.red[16 + wave] = dmax; }
    for (uint32_t b = tid; b <= NB; b += 1024) L.cnt[b] = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; w++) { dmin = min(dmin, L.red[w]); dmax = max(dmax, L.red[16 + w]); }
    const float inv = (float)NB / ((float)(dmax - dmin) + 1.0f);  // monotone, as in bucket_sort_tile
    for (uint32_t i = tid; i < n; i += 1024) {
        const uint32_t d = (uint32_t)(seg[i] >> 32);
        atomicAdd(&L.cnt[min(NB - 1, (uint32_t)((float)(d - dmin) * inv))], 1u);
    }
    __syncthreads();
    // exclusive scan of the NB counts: thread t owns buckets [t * E, t * E + E), E = max(1, NB / 1024) <= 8
    const uint32_t E = NB >= 1024u ? NB / 1024u : 1u;
    uint32_t c[8], tsum = 0;
#pragma unroll
    for (uint32_t e = 0; e < 8; e++) {
        const uint32_t b = (uint32_t)tid * E + e;
        c[e] = (e < E && b < NB) ? L.cnt[b] : 0u;
        tsum += c[e];
    }
    uint32_t incl = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) L.wsum[wave] = incl;
    __syncthreads();
    uint32_t start = incl - tsum;
    for (int w = 0; w < wave; w++) start += L.wsum[w];
#pragma unroll
    for (uint32_t e = 0; e < 8; e++) {
        const uint32_t b = (uint32_t)tid * E + e;
        if (e < E && b < NB) { L.cnt[b] = start; s_start[b] = start; }  // cnt: the bucket's write cursor
        start += c[e];
    }
    if (tid == 0) s_start[NB] = n;
    __syncthreads();
    unsigned long long* out = scratch + r0;
    for (uint32_t i = tid; i < n; i += 1024) {
        const unsigned long long k = seg[i];
        const uint32_t d = (uint32_t)(k >> 32);
        out[atomicAdd(&L.cnt[min(NB - 1, (uint32_t)((float)(d - dmin) * inv))], 1u)] = k;
    }
    __syncthreads();  // (workgroup-scope visibility of the stores above, like the in-place network's stages)
    for (uint32_t b = 0; b < NB; b++) {
        const uint32_t s0 = s_start[b], nb_ = s_start[b + 1] - s0;
        if (nb_ == 0) continue;
        uint32_t Nb = 1;
        while (Nb < nb_) Nb <<= 1;
        if (nb_ > (uint32_t)BUCKET_CAP_MID) {
            unsigned long long* sg = out + s0;
            bitonic_sort_block<false>(sg, nb_, Nb);
            for (uint32_t q = tid; q < nb_; q += 1024) {
                const unsigned long long key = sg[q];
                point_list[r0 + s0 + q] = (uint32_t)key;
                if (keys_sorted) keys_sorted[r0 + s0 + q] = ((unsigned long long)tile << 32) | (key >> 32);
            }
        } else if (Nb <= 4096) bucket_sort_tile<4, 1024, BUCKET_CAP_MID>(L, tile, r0 + s0, nb_, Nb, scratch, point_list, keys_sorted);
        else bucket_sort_tile<8, 1024, BUCKET_CAP_MID>(L, tile, r0 + s0, nb_, Nb, scratch, point_list, keys_sorted);
        __syncthreads();  // the LDS image is reused by the next bucket
    }
}

__global__ void __launch_bounds__(1024) tile_sort_mid_kernel(int T, int nbl, int seg, uint32_t* __restrict__ order,
                                                             const uint32_t* __restrict__ ranges,
                                                             const unsigned long long* bins, unsigned long long* scratch,
                                                             uint32_t* __restrict__ point_list,
                                                             unsigned long long* keys_sorted)
{
    __shared__ BucketSortLds<BUCKET_CAP_MID, 1024> L;
    __shared__ uint32_t s_start[BUCKET_CAP_MID + 1];
    for (uint32_t k = blockIdx.x;; k += gridDim.x) {
        uint32_t tile;
        if (!listed_tile(order + GSR_MID_COUNT(T), order + GSR_MID_LIST(T), nbl, seg, k, tile)) break;
        const uint32_t r0 = __builtin_amdgcn_readfirstlane(ranges[2 * tile]);
        const uint32_t r1 = __builtin_amdgcn_readfirstlane(ranges[2 * tile + 1]);
        const uint32_t n = r1 - r0;
        uint32_t N = 1;
        while (N < n) N <<= 1;
        if (n > BUCKET_CAP_MID) {
            giant_list_sort(L, s_start, tile, r0, n, bins, scratch, point_list, keys_sorted);
        } else if (N <= 4096) bucket_sort_tile<4, 1024, BUCKET_CAP_MID>(L, tile, r0, n, N, bins, point_list, keys_sorted);
        else bucket_sort_tile<8, 1024, BUCKET_CAP_MID>(L, tile, r0, n, N, bins, point_list, keys_sorted);
        __syncthreads();  // the LDS image is reused by the next list
    }
}

}  // namespace

int gsr_tile_slices(int P, int* gpb_out)
{
    int ns = (P + 1023) / 1024;
    if (ns > 512) ns = 512;
    if (ns < 1) ns = 1;
    int gpb = ((P + ns - 1) / ns + 255) / 256 * 256;
    if (gpb < 256) gpb = 256;
    ns = (P + gpb - 1) / gpb;
    if (ns < 1) ns = 1;
    *gpb_out = gpb;
    return ns;
}

// Band geometry for an image: at least 8 bands, more when a band of the image would not fit the LDS histogram.
static BandGeom band_geom(int gx, int gy)
{
    BandGeom G;
    G.gx = gx; G.gy = gy; G.T = gx * gy;
    int nb = gsr_variant("BANDS", 8);
    if ((G.T + 8191) / 8192 > nb) nb = (G.T + 8191) / 8192;
    if (nb > gy) nb = gy;
    if (nb > NB_MAX) nb = NB_MAX;
    if (nb < 1) nb = 1;
    G.bh = (gy + nb - 1) / nb;
    G.nb = (gy + G.bh - 1) / G.bh;
    G.tmax = G.bh * gx;
    G.cull = gsr_cull_enabled() ? 1 : 0;
    return G;
}

bool gsr_band_geometry(int gx, int gy, int* nb, int* bh)
{
    const BandGeom G = band_geom(gx, gy);
    *nb = G.nb; *bh = G.bh;
    return gsr_variant("BIN", 2) == 2 && G.tmax <= TCHUNK && G.nb <= GSR_BAND_COUNTERS;
}

size_t gsr_band_table_bytes(int W, int H)
{
    const BandGeom G = band_geom((W + GSR_TILE_X - 1) / GSR_TILE_X, (H + GSR_TILE_Y - 1) / GSR_TILE_Y);
    return (size_t)G.nb * BAND_SLICES * G.tmax * 4;
}
size_t gsr_band_list_bytes(int P, int W, int H)
{
    const BandGeom G = band_geom((W + GSR_TILE_X - 1) / GSR_TILE_X, (H + GSR_TILE_Y - 1) / GSR_TILE_Y);
    return (size_t)G.nb * (size_t)(P > 0 ? P : 1) * 4 + GSR_BAND_COUNTERS * 4;  // id lists (+ 64 unused words in front)
}

bool gsr_launch_emit_partition(int P, int W, int H, float* splat, const uint32_t* tiles_touched, const uint32_t* block_sums,
                               const uint32_t* row_block_sums, uint32_t* total, uint32_t* row_total, unsigned long long* mailbox,
                               uint32_t seq, uint32_t* point_offsets, uint32_t* row_offsets, uint32_t* wg_first,
                               uint32_t* band_count, uint32_t* band_mem, hipStream_t s)
{
    const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    int nb_, bh_;
    if (!band_mem || !gsr_band_geometry(gx, gy, &nb_, &bh_)) return false;
    const BandGeom G = band_geom(gx, gy);
    // the band counters were zeroed by scan_blocks_kernel (preprocess.hip)
    hipLaunchKernelGGL(emit_partition_kernel, dim3((P + SLICE_THREADS * PART_ITEMS - 1) / (SLICE_THREADS * PART_ITEMS)), dim3(SLICE_THREADS), 0, s, P,
                       G, splat, tiles_touched, block_sums, row_block_sums, total, row_total, mailbox, seq, point_offsets, row_offsets, wg_first,
                       band_count, band_mem + GSR_BAND_COUNTERS);
    return true;
}

// after gsr_launch_emit_partition (banded) or gsr_launch_emit (whole-image slices)
void gsr_launch_tile_binning_head(int P, int W, int H, const float* splat, uint32_t* table, uint32_t* totals, const uint32_t* band_count,
                                  uint32_t* band_mem, hipStream_t s)
{
    const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    const int T = gx * gy;
    const BandGeom G = band_geom(gx, gy);
    int nb_, bh_;
    if (band_mem && gsr_band_geometry(gx, gy, &nb_, &bh_)) {
        const uint32_t* band_list = band_mem + GSR_BAND_COUNTERS;
        hipLaunchKernelGGL(tile_count_banded_kernel, dim3(G.nb * BAND_SLICES), dim3(SLICE_THREADS), (size_t)G.tmax * 4, s, P, G, splat,
                           band_count, band_list, table);
        hipLaunchKernelGGL(tile_colscan_banded_kernel, dim3((G.tmax + 31) / 32, G.nb), dim3(256), 0, s, G, table, totals);
    } else {
        int gpb;
        const int ns = gsr_tile_slices(P, &gpb);
        const size_t lds = (size_t)(T < TCHUNK ? T : TCHUNK) * 4;
        hipLaunchKernelGGL(tile_count_kernel, dim3(ns), dim3(SLICE_THREADS), lds, s, P, gpb, gx, gy, T, G.cull, splat, table);
        hipLaunchKernelGGL(tile_colscan_kernel, dim3((T + 31) / 32), dim3(256), 0, s, T, ns, table, totals);
    }
}

void gsr_launch_tile_binning_tail(int P, int W, int H, const float* splat, uint32_t* table, uint32_t* totals, uint32_t* order,
                                  uint32_t* ranges, const uint32_t* band_count, uint32_t* band_mem, uint64_t* bins, uint32_t* point_list,
                                  uint64_t* keys_sorted, uint64_t* scratch, uint32_t* hint, uint32_t hint_seq, bool launch_mid, hipStream_t s)
{
    const int gx = (W + GSR_TILE_X - 1) / GSR_TILE_X, gy = (H + GSR_TILE_Y - 1) / GSR_TILE_Y;
    const int T = gx * gy;
    const BandGeom G = band_geom(gx, gy);
    int nb_, bh_;
    if (band_mem && gsr_band_geometry(gx, gy, &nb_, &bh_)) {
        const uint32_t* band_list = band_mem + GSR_BAND_COUNTERS;
        hipLaunchKernelGGL(tile_scatter_banded_kernel, dim3(G.nb * BAND_SLICES), dim3(SLICE_THREADS), (size_t)G.tmax * 4, s, P, G, splat,
                           band_count, band_list, table, totals, ranges, order, reinterpret_cast<uint2*>(bins), hint, hint_seq);
    } else {
        int gpb;
        const int ns = gsr_tile_slices(P, &gpb);
        const size_t lds = (size_t)(T < TCHUNK ? T : TCHUNK) * 4;
        hipLaunchKernelGGL(tile_scatter_kernel, dim3(ns), dim3(SLICE_THREADS), lds, s, P, gpb, gx, gy, T, G.cull, splat, table, totals, ranges,
                           order, reinterpret_cast<uint2*>(bins), hint, hint_seq);
    }
    // long-list segments: one per band (banded scatter) or one for the image
    const bool banded = band_mem && gsr_band_geometry(gx, gy, &nb_, &bh_);
    const int nbl = banded ? G.nb : 1, seg = banded ? G.tmax : 0;
    const int n_long = T < 1024 ? T : 1024;
    hipLaunchKernelGGL(tile_sort_small_kernel, dim3(nbl + n_long + (T + 3) / 4), dim3(256), 0, s, T, n_long, nbl, seg, order, ranges,
                       reinterpret_cast<const unsigned long long*>(bins), point_list, reinterpret_cast<unsigned long long*>(keys_sorted),
                       launch_mid ? 0 : 1);
    if (launch_mid)
        hipLaunchKernelGGL(tile_sort_mid_kernel, dim3(T < 256 ? T : 256), dim3(1024), 0, s, T, nbl, seg, order, ranges,
                       reinterpret_cast<const unsigned long long*>(bins), reinterpret_cast<unsigned long long*>(scratch), point_list,
                       reinterpret_cast<unsigned long long*>(keys_sorted));
}

int gsr_binning_ablate() { return GSR_BIN_ABLATE; }
