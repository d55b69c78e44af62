// Stable LSD radix sort of (u64 key, u32 value) pairs on bits [0, end_bit), 8 bits per pass, for wave64.
// Replaces cub::DeviceRadixSort::SortPairs at rasterizer_impl.cu:303-308 (45-bit keys at 1080p -> 6 passes)
// and simple_knn.cu:206-213.  Integer-only, HBM-bound: each pass reads 12 B and writes 12 B per pair, plus
// one extra 8 B key read for the per-workgroup digit histogram.
//
// Pass structure (3 launches): (1) per-workgroup 256-bin histogram, stored bin-major [bin][workgroup];
// (2) 256 workgroups, one per bin, turn their row into exclusive offsets and add the bin's base taken
// from the global digit histogram (computed once for all passes up front); (3) stable scatter: every wave
// owns a contiguous slice of the workgroup's tile and ranks its keys with 8 ballots per key (match-any),
// keeping running per-digit counters in LDS, so the order (wave, round, lane) equals the input order.
#include "common.h"

namespace {

constexpr int THREADS = GSR_SORT_THREADS;   // 256 = 4 waves
constexpr int ITEMS = GSR_SORT_ITEMS;       // keys per thread
constexpr int TILE = GSR_SORT_TILE;         // keys per workgroup
constexpr int WAVES = THREADS / 64;
constexpr int WAVE_SLICE = TILE / WAVES;    // contiguous keys per wave

// All passes' global digit histograms in one sweep over the keys: hist[pass][256].
__global__ void __launch_bounds__(THREADS) global_hist_kernel(const uint64_t* __restrict__ keys, uint32_t n, int passes, int end_bit,
                                                              uint32_t* __restrict__ hist)
{
    __shared__ uint32_t h[8 * 256];
    for (int i = threadIdx.x; i < passes * 256; i += THREADS) h[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * TILE;
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = base + k * THREADS + threadIdx.x;
        if (i < n) {
            const uint64_t key = keys[i];
            for (int p = 0; p < passes; p++) {
                const int shift = 8 * p;
                const int bits = min(8, end_bit - shift);
                const uint32_t d = (uint32_t)(key >> shift) & ((1u << bits) - 1u);
                atomicAdd(&h[p * 256 + d], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < passes * 256; i += THREADS) {
        const uint32_t c = h[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

__global__ void __launch_bounds__(THREADS) block_hist_kernel(const uint64_t* __restrict__ keys, uint32_t n, int shift, uint32_t mask,
                                                             uint32_t nblocks, uint32_t* __restrict__ hist_blocks)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * TILE;
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = base + k * THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    hist_blocks[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// One workgroup per bin: exclusive scan of that bin's per-workgroup counts + the bin's global base.
__global__ void __launch_bounds__(THREADS) bin_scan_kernel(uint32_t nblocks, const uint32_t* __restrict__ hist_global_pass,
                                                           uint32_t* __restrict__ hist_blocks)
{
    __shared__ uint32_t red[THREADS];
    __shared__ uint32_t wtot[WAVES];
    __shared__ uint32_t carry_s;
    const uint32_t bin = blockIdx.x;
    // base of this bin = sum of the global counts of all smaller digits
    red[threadIdx.x] = threadIdx.x < bin ? hist_global_pass[threadIdx.x] : 0u;
    __syncthreads();
    for (int off = THREADS / 2; off >= 1; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) carry_s = red[0];
    __syncthreads();
    uint32_t* row = hist_blocks + (size_t)bin * nblocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += THREADS) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < nblocks ? row[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < wave; w++) wbase += wtot[w];
        const uint32_t carry = carry_s;
        if (i < nblocks) row[i] = carry + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == THREADS - 1) carry_s = carry + wbase + incl;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(THREADS) scatter_kernel(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                          uint64_t* __restrict__ kout, uint32_t* __restrict__ vout, uint32_t n,
                                                          int shift, uint32_t mask, uint32_t nblocks,
                                                          const uint32_t* __restrict__ hist_blocks)
{
    __shared__ uint32_t run[WAVES][256];   // running per-digit counts of each wave's slice
    __shared__ uint32_t gbase[256];        // global offset of (digit, this workgroup)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = 0; w < WAVES; w++) run[w][threadIdx.x] = 0;
    gbase[threadIdx.x] = hist_blocks[(size_t)threadIdx.x * nblocks + blockIdx.x];
    __syncthreads();
    const uint32_t wbase = blockIdx.x * TILE + wave * WAVE_SLICE;
    uint64_t key[ITEMS];
    uint32_t val[ITEMS];
    uint32_t rank[ITEMS];
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = wbase + k * 64 + lane;
        const bool valid = i < n;
        key[k] = valid ? kin[i] : ~0ull;
        val[k] = valid ? vin[i] : 0u;
        const uint32_t d = (uint32_t)(key[k] >> shift) & mask;
        // match-any over the 8 digit bits (invalid lanes form their own group and are never written)
        uint64_t m = __ballot(valid);
        m = valid ? m : ~m;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t before = __popcll(m & lt);
        const uint32_t count = __popcll(m);
        uint32_t r = 0;
        if (valid) {
            r = run[wave][d] + before;          // all lanes of a group read the same counter ...
        }
        // ... before the group's leader bumps it (LDS ops of one wave execute in program order)
        __builtin_amdgcn_wave_barrier();
        if (valid && before == 0) run[wave][d] += count;
        __builtin_amdgcn_wave_barrier();
        rank[k] = r;
    }
    __syncthreads();
    // digit base of this wave inside the workgroup = counts of the preceding waves
    // (computed redundantly per key; 4 waves -> at most 3 LDS reads)
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = wbase + k * 64 + lane;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[k] >> shift) & mask;
            uint32_t pos = gbase[d] + rank[k];
            for (int w = 0; w < wave; w++) pos += run[w][d];
            kout[pos] = key[k];
            vout[pos] = val[k];
        }
    }
}

}  // namespace

int gsr_launch_radix_sort(uint64_t* ka, uint64_t* kb, uint32_t* va, uint32_t* vb, uint32_t n, int end_bit,
                          uint32_t* hist_global, uint32_t* hist_blocks, uint32_t nblocks, hipStream_t s)
{
    const int passes = (end_bit + 7) / 8;
    if (n == 0) return passes & 1;
    hipMemsetAsync(hist_global, 0, 8 * 256 * 4, s);
    hipLaunchKernelGGL(global_hist_kernel, dim3(nblocks), dim3(THREADS), 0, s, ka, n, passes, end_bit, hist_global);
    uint64_t *kin = ka, *kout = kb;
    uint32_t *vin = va, *vout = vb;
    for (int p = 0; p < passes; p++) {
        const int shift = 8 * p;
        const int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        const uint32_t mask = (1u << bits) - 1u;
        hipLaunchKernelGGL(block_hist_kernel, dim3(nblocks), dim3(THREADS), 0, s, kin, n, shift, mask, nblocks, hist_blocks);
        hipLaunchKernelGGL(bin_scan_kernel, dim3(256), dim3(THREADS), 0, s, nblocks, hist_global + p * 256, hist_blocks);
        hipLaunchKernelGGL(scatter_kernel, dim3(nblocks), dim3(THREADS), 0, s, kin, vin, kout, vout, n, shift, mask, nblocks, hist_blocks);
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    return passes & 1;
}
