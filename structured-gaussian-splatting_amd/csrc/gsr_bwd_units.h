// gsr_bwd_units.h — the work-unit list of the blend backward (K7), longest unit first.
//
// A unit is (tile, depth chunk, segment): kSeg consecutive entries of the tile's list in that chunk, walked front to back from
// the forward's checkpoint (gsr_render.hip).  Only entries in front of the tile's deepest contributor in the chunk are walked
// (ImageWS::tile_walk, written by K6), so a (tile, chunk) pair with w walked entries gives floor(w / kSeg) FULL units and one
// partial unit of w % kSeg entries (also when w == 0, so that every non-empty pair has a last unit: the flag in bit 31).
// One wave per unit, all units resident or queued at once: the launch ends when the last-started units finish, so the full
// units go first and the partial ones follow by descending length (a counting sort on the length, kSeg bins; units of equal
// length land in the order their LDS atomics did — the order only decides which block runs a unit, never a value).
// One block builds the list; a device function so that two kernels can carry it: k_bwd_units (gsr_render.hip) and block 0 of the
// backward's zero fill (gsr_geom.hip: no launch, no stream time of its own).
#pragma once
#include "gsr_internal.h"

namespace gsr {

// kPer (pair, walked) records per thread are loaded up front, all loads in flight together: the block is latency-bound (8 160
// pairs at 1080p; two dependent global loads per pair cost 2 us a round), so rounds of THREADS x kPer pairs, not of THREADS.
template <int THREADS, int kPer>
__device__ __forceinline__ void bwd_units_block(const BwdUnitArgs a)
{
    __shared__ uint32_t hist[kSeg];              // partial units by length: bin kSeg - 1 - length (descending)
    __shared__ uint32_t sh_full, sh_cursor;
    const int lane = threadIdx.x & 63;
    for (int j = threadIdx.x; j < kSeg; j += THREADS) hist[j] = 0;
    if (threadIdx.x == 0) { sh_full = 0; sh_cursor = 0; }
    __syncthreads();
    const long long pairs = (long long)a.n_tiles * a.chunks_run;
    // walked entries of pair q (0xFFFFFFFF: the pair's range is empty, no unit) and its (tile | chunk << 24) word
    auto load_round = [&](long long q0, uint32_t (&w)[kPer], uint32_t (&head)[kPer]) {
        uint2 r[kPer];
        uint32_t wk[kPer];
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const long long q = q0 + (long long)i * THREADS + threadIdx.x;
            const uint32_t c = q < pairs ? (uint32_t)(q / a.n_tiles) : 0u;
            const uint32_t tile = (uint32_t)a.tile_base + (q < pairs ? (uint32_t)(q - (long long)c * a.n_tiles) : 0u);
            head[i] = tile | (c << kUnitTileBits);
            r[i] = q < pairs ? a.ranges[(size_t)c * a.Tn + tile] : make_uint2(0u, 0u);
            wk[i] = q < pairs ? a.tile_walk[(size_t)c * a.Tn + tile] : 0u;
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const uint32_t n = r[i].y - r[i].x;
            w[i] = n == 0u ? 0xFFFFFFFFu : (wk[i] < n ? wk[i] : n);
        }
    };
    uint32_t my_full = 0;
    for (long long q0 = 0; q0 < pairs; q0 += (long long)THREADS * kPer) {
        uint32_t w[kPer], head[kPer];
        load_round(q0, w, head);
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            if (w[i] == 0xFFFFFFFFu) continue;
            const uint32_t full = w[i] / kSeg, rest = w[i] - full * kSeg;
            my_full += full;
            if (rest > 0u || full == 0u) atomicAdd(&hist[kSeg - 1 - rest], 1u);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) my_full += (uint32_t)__shfl_xor((int)my_full, off);
    if (lane == 0 && my_full) atomicAdd(&sh_full, my_full);
    __syncthreads();
    const uint32_t n_full = sh_full;
    if (threadIdx.x < kWave) {                   // exclusive scan of the kSeg bins by the first wave
        uint32_t run = 0;
        for (int b0 = 0; b0 < kSeg; b0 += kWave) {
            const uint32_t v = hist[b0 + lane];
            uint32_t inc = v;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const uint32_t u = (uint32_t)__shfl_up((int)inc, off);
                if (lane >= off) inc += u;
            }
            hist[b0 + lane] = n_full + run + inc - v;
            run += (uint32_t)__shfl((int)inc, kWave - 1);
        }
        if (lane == 0) *a.n_units = n_full + run < a.capacity ? n_full + run : a.capacity;
    }
    __syncthreads();
    for (long long q0 = 0; q0 < pairs; q0 += (long long)THREADS * kPer) {          // (every lane runs every round: a wave shares one cursor bump)
        uint32_t w[kPer], head[kPer];
        load_round(q0, w, head);
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const bool have = w[i] != 0xFFFFFFFFu;
            const uint32_t full = have ? w[i] / kSeg : 0u, rest = have ? w[i] - full * kSeg : 0u;
            const bool partial = have && (rest > 0u || full == 0u);
            // full units: one LDS atomic per wave (8 160 lanes queueing on one address cost 30 us), a shuffle scan inside it
            uint32_t inc = full;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)inc, off);
                if (lane >= off) inc += v;
            }
            uint32_t wave_at = 0;
            const uint32_t wave_total = (uint32_t)__shfl((int)inc, kWave - 1);
            if (lane == kWave - 1 && wave_total) wave_at = atomicAdd(&sh_cursor, wave_total);
            wave_at = (uint32_t)__shfl((int)wave_at, kWave - 1);
            const uint32_t at_full = wave_at + inc - full;
            for (uint32_t sgm = 0; sgm < full; ++sgm)
                if (at_full + sgm < a.capacity) a.units[at_full + sgm] = make_uint2(head[i], sgm | ((!partial && sgm == full - 1) ? kUnitLast : 0u));
            if (partial) {
                const uint32_t at = atomicAdd(&hist[kSeg - 1 - rest], 1u);
                if (at < a.capacity) a.units[at] = make_uint2(head[i], full | kUnitLast);
            }
        }
    }
}

}  // namespace gsr
