// gsr_tile_order.h — launch order of the blend backward (K7): the tiles of the slab by the work the forward measured,
// longest first.  All of a frame's tiles are resident or queued at once (one wave each, 4 per SIMD) and a wave lives for a third
// of the kernel, so the launch ends with the SIMDs draining: in tile order that tail was 45 % of the kernel at cfg3
// (tools/bwd_trace.py) and the XCDs, which owned contiguous bands of tiles, finished up to 60 us apart.  Longest first, dealt
// round-robin over the XCDs (block b runs on XCD b % 8), evens the XCDs out and leaves the shortest tiles for the end.
// One block: counting sort on 9 bits of the work (the longest tile -> 511) and 3 bits of the tile index, descending.  Tiles of
// equal work land in the order their atomics did: the order decides which block runs a tile, never a value.
// A device function, so that two kernels can carry it: k_tile_order (gsr_render.hip, 1024 threads) and block 0 of the
// backward's zero fill (gsr_geom.hip, 256 threads x 32 tiles in registers: the sort then costs no launch and no stream time).
#pragma once
#include "gsr_internal.h"

namespace gsr {

constexpr int kOrderBins = 4096;
constexpr int kOrderThreads = 1024;

// PER: tiles a thread keeps in registers (THREADS x PER of them; more tiles spill over to a second pass of LDS atomics)
template <int THREADS, int kOrderPer>
__device__ __forceinline__ void tile_order_block(int n_tiles, int tile_base, const uint32_t *__restrict__ tile_work,
                                                 uint32_t *__restrict__ tile_order)
{
    __shared__ uint32_t hist[kOrderBins];
    __shared__ uint32_t sh_wave[THREADS / kWave];
    __shared__ uint32_t sh_max;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = threadIdx.x; j < kOrderBins; j += THREADS) hist[j] = 0;
    if (threadIdx.x == 0) sh_max = 0;
    uint32_t w[kOrderPer], mx = 0;
#pragma unroll
    for (int i = 0; i < kOrderPer; ++i) {
        const int t = threadIdx.x + i * THREADS;
        w[i] = t < n_tiles ? tile_work[tile_base + t] : 0u;
        mx = w[i] > mx ? w[i] : mx;
    }
    for (int t = threadIdx.x + kOrderPer * THREADS; t < n_tiles; t += THREADS) { const uint32_t v = tile_work[tile_base + t]; mx = v > mx ? v : mx; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const uint32_t u = (uint32_t)__shfl_xor((int)mx, off); mx = u > mx ? u : mx; }
    __syncthreads();
    if (lane == 0) atomicMax(&sh_max, mx);
    __syncthreads();
    // bin = work scaled to 9 bits (the longest tile -> 511), three low bits from the tile index: equal work spreads over
    // eight counters instead of queueing on one LDS address
    int shift = 0;
    while ((sh_max >> shift) > 511u) ++shift;
    auto bin_of = [&](uint32_t work, int t) { return (kOrderBins - 1) - (int)(((work >> shift) << 3) | (uint32_t)(7 - (t & 7))); };
    uint32_t slot[kOrderPer];
#pragma unroll
    for (int i = 0; i < kOrderPer; ++i) {
        const int t = threadIdx.x + i * THREADS;
        if (t < n_tiles) slot[i] = atomicAdd(&hist[bin_of(w[i], t)], 1u);
    }
    __syncthreads();                                   // the register-held tiles own the first slots of their bins
    for (int t = threadIdx.x + kOrderPer * THREADS; t < n_tiles; t += THREADS) atomicAdd(&hist[bin_of(tile_work[tile_base + t], t)], 1u);
    __syncthreads();
    {   // exclusive scan of the 4096 bins, kBinsPer consecutive ones per thread
        constexpr int kBinsPer = kOrderBins / THREADS;
        uint32_t c[kBinsPer], mine = 0;
#pragma unroll
        for (int i = 0; i < kBinsPer; ++i) { c[i] = hist[kBinsPer * threadIdx.x + i]; mine += c[i]; }
        uint32_t inc = mine;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)inc, off);
            if (lane >= off) inc += u;
        }
        if (lane == 63) sh_wave[wv] = inc;
        __syncthreads();
        uint32_t run = inc - mine;
        for (int i = 0; i < wv; ++i) run += sh_wave[i];
#pragma unroll
        for (int i = 0; i < kBinsPer; ++i) { hist[kBinsPer * threadIdx.x + i] = run; run += c[i]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kOrderPer; ++i) {
        const int t = threadIdx.x + i * THREADS;
        if (t < n_tiles) tile_order[hist[bin_of(w[i], t)] + slot[i]] = (uint32_t)t;
    }
    // more tiles than the registers hold: the rest goes behind its bin's register-held tiles, in the order the atomics land
    __syncthreads();
    if (n_tiles > kOrderPer * THREADS) {
#pragma unroll
        for (int i = 0; i < kOrderPer; ++i) {           // advance every bin past the slots handed out above
            const int t = threadIdx.x + i * THREADS;
            if (t < n_tiles) atomicAdd(&hist[bin_of(w[i], t)], 1u);
        }
        __syncthreads();
        for (int t = threadIdx.x + kOrderPer * THREADS; t < n_tiles; t += THREADS)
            tile_order[atomicAdd(&hist[bin_of(tile_work[tile_base + t], t)], 1u)] = (uint32_t)t;
    }
}

}  // namespace gsr
