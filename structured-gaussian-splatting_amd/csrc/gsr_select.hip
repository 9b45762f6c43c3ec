// gsr_select.hip — the depth order of a frame by SELECTION: only what gets binned is ever sorted.
//
// The reference sorts every (tile, depth) duplicate of the frame (cub radix sort, rasterizer_impl.cu K4); round 1 of
// this design sorted the P Gaussians by depth first (100 us at 1e6) and then binned depth chunks until every tile
// had closed.  A frame whose tiles saturate closes after the first chunk — 0.6 % of the Gaussians at cfg3 — so 99 %
// of that sort ordered Gaussians nobody looked at.  Here:
//   1. a two-level histogram of the depth keys (2048 bins of the key's top bits, then 2048 sub-bins inside the bins
//      the first chunk boundaries fall into) carries, per bin, the Gaussian count, the tiles touched and the optical
//      mass (gsr_math.h optical_mass); a one-block scan turns the chunk rule of gsr_binning.hip (chunk c ends where the
//      running mass passes first_mass * 4^c and the running tile count passes kMinFirstChunk * 4^c) into KEY
//      thresholds, with the exact Gaussian and tile counts of every chunk — no sorted order needed;
//   2. a stable partition by chunk (one pass: 8-way, so per-wave ballots instead of digit histograms) puts chunk c's
//      Gaussians at order[bnd[c] .. bnd[c+1]) in index order;
//   3. a chunk is sorted by (depth, index) only when it is about to be binned: its keys are gathered relative to the
//      chunk's lowest key and sorted in LDS by one block (chunks up to 16 K Gaussians: the nearest chunk of a
//      saturating frame) or by the library's radix sort on the bits that actually vary inside the chunk.
// Equal keys always share a chunk and keep index order inside it, so the (depth, index) order of the binned prefix
// is exactly the stable full sort's; the image does not depend on where chunks end.
#include "gsr_internal.h"

namespace gsr {

constexpr int kSelThreads = 512;
constexpr int kSelHistBlocks = 256;
constexpr int kSelWaves = kSelThreads / kWave;

__device__ __forceinline__ uint32_t sel_bin1(uint32_t key) { const uint32_t b = key >> kSelShift1; return b < (uint32_t)kSelBins ? b : (uint32_t)kSelBins - 1u; }
__device__ __forceinline__ uint32_t sel_bin2(uint32_t key) { return (key >> kSelShift2) & (uint32_t)(kSelBins - 1); }

// inclusive scan of (count, tiles, mass) over the 2048 bins of a table by one block of kSelThreads threads (4 consecutive bins
// each); results into LDS arrays.
struct Triple { uint32_t n; unsigned long long t, m; };
__device__ __forceinline__ Triple operator+(const Triple &a, const Triple &b) { return Triple{a.n + b.n, a.t + b.t, a.m + b.m}; }
constexpr int kSelPerThread = kSelBins / kSelThreads;

__device__ __forceinline__ void scan_table(const SelTables &tab, Triple base, uint32_t *cum_n, unsigned long long *cum_t, unsigned long long *cum_m,
                                           Triple *sh_wave /* [kSelWaves] */)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b0 = kSelPerThread * threadIdx.x;
    Triple v[kSelPerThread], mine{0u, 0ull, 0ull};
#pragma unroll
    for (int i = 0; i < kSelPerThread; ++i) {
        v[i] = Triple{tab.cnt[b0 + i], tab.tiles[b0 + i], tab.mass[b0 + i]};
        mine = mine + v[i];
    }
    Triple inc = mine;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        Triple u{(uint32_t)__shfl_up((int)inc.n, off), (unsigned long long)__shfl_up((long long)inc.t, off),
                 (unsigned long long)__shfl_up((long long)inc.m, off)};
        if (lane >= off) inc = inc + u;
    }
    __syncthreads();                                             // (sh_wave and the cum arrays may still be read from a previous use)
    if (lane == 63) sh_wave[wv] = inc;
    __syncthreads();
    Triple run = base;
    for (int i = 0; i < wv; ++i) run = run + sh_wave[i];
    run = Triple{run.n + inc.n - mine.n, run.t + inc.t - mine.t, run.m + inc.m - mine.m};
#pragma unroll
    for (int i = 0; i < kSelPerThread; ++i) {
        run = run + v[i];
        cum_n[b0 + i] = run.n; cum_t[b0 + i] = run.t; cum_m[b0 + i] = run.m;
    }
    __syncthreads();
}

// the chunk rule: boundary c lies behind the first bin whose running mass exceeds first_mass * 4^c AND whose running
// tile count exceeds kMinFirstChunk * 4^c (the same two conditions as round 1's plan on the sorted order)
__device__ __forceinline__ unsigned long long sel_mass_target(unsigned long long first_mass, int c) { return first_mass << (kChunkGrowthLog2 * c); }
__device__ __forceinline__ unsigned long long sel_tile_floor(int c) { return (unsigned long long)kMinFirstChunk << (kChunkGrowthLog2 * c); }

// first bin of the scanned table at which both running sums have passed their thresholds (kSelBins: none)
__device__ __forceinline__ void first_crossing(const unsigned long long *cum_t, const unsigned long long *cum_m, unsigned long long T,
                                               unsigned long long F, uint32_t *out)
{
#pragma unroll
    for (int k = 0; k < kSelPerThread; ++k) {
        const int b = kSelPerThread * threadIdx.x + k;
        const bool here = cum_m[b] > T && cum_t[b] > F;
        const bool prev = b > 0 && cum_m[b - 1] > T && cum_t[b - 1] > F;
        if (here && !prev) *out = (uint32_t)b;                  // both sums are monotone: exactly one such bin, or none
    }
}

// ---- 1b. level-1 scan: V, R, the bin of every boundary; the first
// kSelRefine boundaries get refined at level 2
// (every block of the level-2 histogram runs this on its own — 40 KB of table out of L2 and a scan, cheaper than a one-block
// launch in between; block 0 stores the result for the plan step, the others only use sh_first)
__device__ void sel_plan1(unsigned long long first_mass, SelState *st, uint32_t *cum_n, unsigned long long *cum_t, unsigned long long *cum_m,
                          Triple *sh_wave, uint32_t *sh_first /* [GSR_MAX_CHUNKS] */, bool write)
{
    if (threadIdx.x < GSR_MAX_CHUNKS) sh_first[threadIdx.x] = (uint32_t)kSelBins;
    scan_table(st->t1, Triple{0u, 0ull, 0ull}, cum_n, cum_t, cum_m, sh_wave);
    const uint32_t V = cum_n[kSelBins - 1];
    const unsigned long long R = cum_t[kSelBins - 1];
    for (int c = 0; c < GSR_MAX_CHUNKS - 1; ++c) {
        if (sel_tile_floor(c) >= R) continue;                   // a chunk may not be smaller than the floor: it takes the rest
        first_crossing(cum_t, cum_m, sel_mass_target(first_mass, c), sel_tile_floor(c), &sh_first[c]);
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (!write) return;
    // the highest occupied bin: the largest key of the frame lies in it (the last chunk's sort needs no bits above that)
#pragma unroll
    for (int k = 0; k < kSelPerThread; ++k) {
        const int b = kSelPerThread * threadIdx.x + k;
        if (V > 0u && cum_n[b] >= V && (b == 0 || cum_n[b - 1] < V)) st->max_bin = (uint32_t)b;
    }
    if (c == 0) { st->V = V; st->R = R; if (V == 0u) st->max_bin = 0u; }
    if (c < GSR_MAX_CHUNKS - 1) {
        const uint32_t b = sh_first[c];                         // kSelBins: no boundary, the chunk takes everything left
        st->coarse_bin[c] = b;
        if (b < (uint32_t)kSelBins) {
            st->coarse_cnt[c] = cum_n[b]; st->coarse_tiles[c] = cum_t[b];
            if (c < kSelRefine) {
                st->bin[c] = b;
                st->base_cnt[c] = b ? cum_n[b - 1] : 0u; st->base_tiles[c] = b ? cum_t[b - 1] : 0ull; st->base_mass[c] = b ? cum_m[b - 1] : 0ull;
            }
        } else if (c < kSelRefine) st->bin[c] = (uint32_t)kSelBins;
    }
}

// ---- 1d. level-2 scan of the refined boundaries, then the plan itself:
// key thresholds, rank boundaries, instance bounds
// (every block of the partition's count pass runs this on its own; block 0 stores the plan, all of them leave the chunk
// count and the key thresholds in out_n / out_ends)
__device__ void sel_plan2(unsigned long long first_mass, const SelState *st, Ctrl *ctrl, uint32_t *cum_n, unsigned long long *cum_t,
                          unsigned long long *cum_m, Triple *sh_wave, uint32_t *sh_small /* [4 * GSR_MAX_CHUNKS] */, bool write,
                          uint32_t *out_n, uint32_t *out_ends /* [GSR_MAX_CHUNKS] */)
{
    uint32_t *sh_sub = sh_small, *sh_key = sh_small + GSR_MAX_CHUNKS, *sh_cnt = sh_small + 2 * GSR_MAX_CHUNKS;
    __shared__ unsigned long long sh_tiles[GSR_MAX_CHUNKS];
    const unsigned long long R = st->R;
    for (int c = 0; c < kSelRefine; ++c) {
        const uint32_t b1 = st->bin[c];
        if (b1 >= (uint32_t)kSelBins) continue;                 // block-uniform
        if (threadIdx.x == 0) sh_sub[0] = (uint32_t)kSelBins;
        scan_table(st->t2[c], Triple{st->base_cnt[c], st->base_tiles[c], st->base_mass[c]}, cum_n, cum_t, cum_m, sh_wave);
        first_crossing(cum_t, cum_m, sel_mass_target(first_mass, c), sel_tile_floor(c), &sh_sub[0]);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t sub = sh_sub[0] < (uint32_t)kSelBins ? sh_sub[0] : (uint32_t)kSelBins - 1u;      // (the level-1 bin does cross)
            sh_key[c] = (b1 << kSelShift1) | (sub << kSelShift2) | ((1u << kSelShift2) - 1u);
            sh_cnt[c] = cum_n[sub]; sh_tiles[c] = cum_t[sub];
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;                               // (the caller's next barrier publishes out_n / out_ends)
    // the plan: chunks in depth order; an empty one (two boundaries inside one sub-bin) is merged into its successor
    const uint32_t V = st->V;
    uint32_t n = 0, begin = 0;
    uint32_t p_key[GSR_MAX_CHUNKS], p_cnt[GSR_MAX_CHUNKS], p_full[GSR_MAX_CHUNKS];
    unsigned long long begin_tiles = 0;
    for (int k = 0; k < GSR_MAX_CHUNKS && begin < V; ++k) {
        uint32_t key = 0xFFFFFFFEu, cnt = V;
        unsigned long long tiles = R;
        if (k < GSR_MAX_CHUNKS - 1 && st->coarse_bin[k] < (uint32_t)kSelBins) {
            if (k < kSelRefine) { key = sh_key[k]; cnt = sh_cnt[k]; tiles = sh_tiles[k]; }
            else {
                const unsigned long long edge = ((unsigned long long)(st->coarse_bin[k] + 1u) << kSelShift1) - 1ull;
                key = edge > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)edge; cnt = st->coarse_cnt[k]; tiles = st->coarse_tiles[k];
            }
        }
        if (cnt <= begin) continue;
        const unsigned long long full = tiles - begin_tiles;
        p_key[n] = key; p_cnt[n] = cnt; p_full[n] = full > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)full;
        begin = cnt; begin_tiles = tiles;
        ++n;
        if (cnt >= V) break;
    }
    for (uint32_t k = n; k < GSR_MAX_CHUNKS; ++k) { p_key[k] = 0xFFFFFFFEu; p_cnt[k] = V; p_full[k] = 0; }
    *out_n = n;
    for (int k = 0; k < GSR_MAX_CHUNKS; ++k) out_ends[k] = p_key[k];
    if (!write) return;
    for (int k = 0; k < GSR_MAX_CHUNKS; ++k) { ctrl->key_end[k] = p_key[k]; ctrl->bnd[k + 1] = p_cnt[k]; ctrl->chunk_full[k] = p_full[k]; }
    for (int k = 0; k < GSR_MAX_CHUNKS; ++k) { ctrl->chunk_R[k] = 0; ctrl->chunk_base[k + 1] = 0; ctrl->chunk_live[k] = 0xFFFFFFFFu; }
    ctrl->bnd[0] = 0; ctrl->chunk_base[0] = 0;
    {
        const unsigned long long edge = ((unsigned long long)(st->max_bin + 1u) << kSelShift1) - 1ull;
        ctrl->key_max = edge > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)edge;
    }
    ctrl->V = V;
    ctrl->R_total = R > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)R;
    ctrl->overflow = R > 0xFFFFFFFFull ? 1u : 0u;
    ctrl->open_count = 0;
    ctrl->num_chunks = n;
}

// ---- 1a/1c. histograms (LEVEL 1: all visible Gaussians; LEVEL 2: those inside the bins picked by level 1's plan)
template <int LEVEL>
__global__ __launch_bounds__(kSelThreads) void k_sel_hist(int P, const uint32_t *__restrict__ keys, const uint2 *__restrict__ tiles_mass,
                                                          unsigned long long first_mass, SelState *st)
{
    constexpr int NT = LEVEL == 1 ? 1 : kSelRefine;
    __shared__ uint32_t sh_cnt[NT][kSelBins];
    __shared__ unsigned long long sh_tiles[NT][kSelBins], sh_mass[NT][kSelBins];
    __shared__ Triple sh_wave[kSelWaves];
    __shared__ uint32_t sh_first[GSR_MAX_CHUNKS];
    uint32_t want[NT];
    bool any = true;
    if (LEVEL == 2) {
        // the level-1 plan step, in every block (the histogram arrays hold its running sums first)
        sel_plan1(first_mass, st, sh_cnt[0], sh_tiles[0], sh_mass[0], sh_wave, sh_first, blockIdx.x == 0);
        any = false;
#pragma unroll
        for (int c = 0; c < NT; ++c) { want[c] = sh_first[c]; any |= want[c] < (uint32_t)kSelBins; }
        __syncthreads();
    }
    if (any) {                                                   // (level 2: some boundary needs refining)
        for (int j = threadIdx.x; j < NT * kSelBins; j += kSelThreads) { (&sh_cnt[0][0])[j] = 0; (&sh_tiles[0][0])[j] = 0; (&sh_mass[0][0])[j] = 0; }
        __syncthreads();
        for (int i = blockIdx.x * kSelThreads + threadIdx.x; i < P; i += gridDim.x * kSelThreads) {
            const uint32_t key = keys[i];
            if (key == 0xFFFFFFFFu) continue;
            if (LEVEL == 1) {
                const uint2 tm = tiles_mass[i];
                const uint32_t b = sel_bin1(key);
                atomicAdd(&sh_cnt[0][b], 1u); atomicAdd(&sh_tiles[0][b], (unsigned long long)tm.x); atomicAdd(&sh_mass[0][b], (unsigned long long)tm.y);
            } else {
                const uint32_t b1 = sel_bin1(key);
                bool hit = false;
#pragma unroll
                for (int c = 0; c < NT; ++c) hit |= b1 == want[c];
                if (!hit) continue;
                const uint2 tm = tiles_mass[i];
                const uint32_t b = sel_bin2(key);
#pragma unroll
                for (int c = 0; c < NT; ++c)
                    if (b1 == want[c]) {
                        atomicAdd(&sh_cnt[c][b], 1u); atomicAdd(&sh_tiles[c][b], (unsigned long long)tm.x);
                        atomicAdd(&sh_mass[c][b], (unsigned long long)tm.y);
                    }
            }
        }
        __syncthreads();
        SelTables *out = LEVEL == 1 ? &st->t1 : st->t2;
        for (int j = threadIdx.x; j < NT * kSelBins; j += kSelThreads) {
            const int c = j / kSelBins, b = j - c * kSelBins;
            const uint32_t n = sh_cnt[c][b];
            if (n) { atomicAdd(&out[c].cnt[b], n); atomicAdd(&out[c].tiles[b], sh_tiles[c][b]); atomicAdd(&out[c].mass[b], sh_mass[c][b]); }
        }
    }
}

// ---- 2. stable partition of the visible Gaussians by chunk: order[bnd[c] + j] = the j-th (by index) Gaussian of chunk c
__device__ __forceinline__ int sel_chunk_of(uint32_t key, const uint32_t *ends, int n)
{
    int c = 0;
    for (int k = 0; k + 1 < n; ++k) c += key > ends[k] ? 1 : 0;
    return c;
}

__device__ __forceinline__ void sel_block_range(int P, int &lo, int &hi)
{
    int per = (P + (int)gridDim.x - 1) / (int)gridDim.x;
    per = (per + kSelThreads - 1) / kSelThreads * kSelThreads;
    const long long l = (long long)blockIdx.x * per;
    lo = l < P ? (int)l : P;
    hi = l + per < P ? (int)(l + per) : P;
}

__global__ __launch_bounds__(kSelThreads) void k_part_count(int P, const uint32_t *__restrict__ keys, unsigned long long first_mass, Ctrl *ctrl,
                                                            SelState *st)
{
    __shared__ uint32_t sh_cnt[GSR_MAX_CHUNKS];
    __shared__ uint32_t ends[GSR_MAX_CHUNKS];
    __shared__ uint32_t cum_n[kSelBins];
    __shared__ unsigned long long cum_t[kSelBins], cum_m[kSelBins];
    __shared__ Triple sh_wave[kSelWaves];
    __shared__ uint32_t sh_small[4 * GSR_MAX_CHUNKS];
    __shared__ uint32_t sh_n;
    // the plan step (level-2 scan + chunk table), in every block; block 0 stores it for the scatter pass and the host
    sel_plan2(first_mass, st, ctrl, cum_n, cum_t, cum_m, sh_wave, sh_small, blockIdx.x == 0, &sh_n, ends);
    if (threadIdx.x < GSR_MAX_CHUNKS) sh_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int n = (int)sh_n;
    int lo, hi;
    sel_block_range(P, lo, hi);
    uint32_t mine[GSR_MAX_CHUNKS] = {};
    for (int i = lo + (int)threadIdx.x; i < hi; i += kSelThreads) {
        const uint32_t key = keys[i];
        const int c = key == 0xFFFFFFFFu ? -1 : sel_chunk_of(key, ends, n);
        for (int k = 0; k < n; ++k) {
            const unsigned long long m = __ballot(c == k);
            if ((threadIdx.x & 63) == 0) mine[k] += (uint32_t)__popcll(m);
        }
    }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < n; ++k)
            if (mine[k]) atomicAdd(&sh_cnt[k], mine[k]);
    __syncthreads();
    if ((int)threadIdx.x < n) st->blk_cnt[threadIdx.x][blockIdx.x] = sh_cnt[threadIdx.x];
}

// relative key of a Gaussian inside its chunk: the chunk's keys lie in (lower, key_end]; visible depths exceed the near cut,
// whose bits bound the first chunk from below.  The chunk sort works on these (fewer varying bits = fewer passes).
constexpr uint32_t kNearBits = 0x3E4CCCCDu;                      // 0.2f (GSR_NEAR_CUT): in_frustum() keeps view depth > 0.2
__host__ __device__ __forceinline__ uint32_t sel_key_base(uint32_t prev_end, bool first) { return first || prev_end < kNearBits ? kNearBits : prev_end; }

__global__ __launch_bounds__(kSelThreads) void k_part_scatter(int P, const uint32_t *__restrict__ keys, const uint2 *__restrict__ tiles_mass,
                                                              const Ctrl *__restrict__ ctrl, const SelState *__restrict__ st,
                                                              uint32_t *__restrict__ order, uint32_t *__restrict__ pos_key,
                                                              uint32_t *__restrict__ pos_tiles, uint4 *__restrict__ clear16, int clear16_n)
{
    // on the side: the frame's per-chunk tile ranges and per-tile counters start at zero (no memset launch of their own)
    for (int z = (int)(blockIdx.x * kSelThreads + threadIdx.x); z < clear16_n; z += (int)gridDim.x * kSelThreads)
        clear16[z] = make_uint4(0u, 0u, 0u, 0u);
    __shared__ uint32_t ends[GSR_MAX_CHUNKS];
    __shared__ uint32_t sh_run[GSR_MAX_CHUNKS];                  // next free position of chunk k for this block
    __shared__ uint32_t sh_w[GSR_MAX_CHUNKS][kSelWaves], sh_pre[GSR_MAX_CHUNKS][kSelWaves];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = (int)ctrl->num_chunks;
    if (threadIdx.x < GSR_MAX_CHUNKS) ends[threadIdx.x] = ctrl->key_end[threadIdx.x];
    if (wv < n) {                                                // wave k: chunk k's Gaussians in the blocks before this one
        uint32_t s = 0;
        for (int b = lane; b < (int)blockIdx.x; b += kWave) s += st->blk_cnt[wv][b];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += (uint32_t)__shfl_xor((int)s, off);
        if (lane == 0) sh_run[wv] = ctrl->bnd[wv] + s;
    }
    __syncthreads();
    int lo, hi;
    sel_block_range(P, lo, hi);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int i0 = lo; i0 < hi; i0 += kSelThreads) {
        const int i = i0 + (int)threadIdx.x;
        const uint32_t key = i < hi ? keys[i] : 0xFFFFFFFFu;
        const int c = key == 0xFFFFFFFFu ? -1 : sel_chunk_of(key, ends, n);
        uint32_t rank = 0;
        for (int k = 0; k < n; ++k) {
            const unsigned long long m = __ballot(c == k);
            if (c == k) rank = (uint32_t)__popcll(m & below);
            if (lane == 0) sh_w[k][wv] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if ((int)threadIdx.x < n) {
            uint32_t run = sh_run[threadIdx.x];
#pragma unroll
            for (int w = 0; w < kSelWaves; ++w) { sh_pre[threadIdx.x][w] = run; run += sh_w[threadIdx.x][w]; }
            sh_run[threadIdx.x] = run;
        }
        __syncthreads();
        if (c >= 0) {
            const uint32_t pos = sh_pre[c][wv] + rank, base = sel_key_base(c > 0 ? ends[c - 1] : 0u, c == 0);
            order[pos] = (uint32_t)i;
            pos_key[pos] = key > base ? key - base : 0u;
            pos_tiles[pos] = tiles_mass[i].x;
        }
    }
}

int launch_depth_select(const FrameK &f, GeomWS &ws, bool debug, hipStream_t s, void *clear16, size_t clear16_n)
{
    if (f.P == 0) return GSR_OK;
    const double slab_px = (double)(f.ty1 - f.ty0) * GSR_TILE * (double)f.Gx * GSR_TILE;
    const unsigned long long first_mass =
        (unsigned long long)((double)kChunkOpticalDepths * kCutoffOpticalDepth * slab_px * (double)kMassUnitsPerPixelNeper) + 1ull;
    // grid sizes, swept at 1e6 and 5e6 Gaussians: the histogram kernels are fastest with one block per CU (every block flushes
    // the bins it touched with global atomics: 11 / 10.5 us at 256 blocks, 14.6 / 16.6 at 1024), the partition's scatter with
    // four (12.3 -> 9.7 us; 53 -> 25 us at 5e6); the count pass does not care and must match the scatter
    int blocks = (f.P + kSelThreads * 4 - 1) / (kSelThreads * 4);
    if (blocks > kSelBlocks) blocks = kSelBlocks;
    const int hist_blocks = blocks > kSelHistBlocks ? kSelHistBlocks : blocks;
    {
        ProfileScope prof("depth_hist", s);
        hipLaunchKernelGGL(k_sel_hist<1>, dim3(hist_blocks), dim3(kSelThreads), 0, s, f.P, ws.sort_keys[0], ws.tiles_mass, first_mass, ws.sel);
        hipLaunchKernelGGL(k_sel_hist<2>, dim3(hist_blocks), dim3(kSelThreads), 0, s, f.P, ws.sort_keys[0], ws.tiles_mass, first_mass, ws.sel);
        GSR_LAUNCH_CHECK("depth_hist", debug, s);
    }
    {
        ProfileScope prof("depth_partition", s);
        hipLaunchKernelGGL(k_part_count, dim3(blocks), dim3(kSelThreads), 0, s, f.P, ws.sort_keys[0], first_mass, ws.ctrl, ws.sel);
        hipLaunchKernelGGL(k_part_scatter, dim3(blocks), dim3(kSelThreads), 0, s, f.P, ws.sort_keys[0], ws.tiles_mass, ws.ctrl, ws.sel, ws.order,
                           ws.sort_keys[1], ws.sort_vals[1], reinterpret_cast<uint4 *>(clear16), (int)clear16_n);
        GSR_LAUNCH_CHECK("depth_partition", debug, s);
    }
    return GSR_OK;
}


// ---- 2b. multi-GPU gradient exchange (sharded.py): the Gaussians whose depth key is <= key_max — every Gaussian some rank
// binned — in INDEX order, and their 48-byte screen-space gradient rows packed behind one another for the all-reduce.  Keys do
// not depend on a rank's slab, so every rank builds the same list.  Two launches (count per block, then scatter with the
// blocks before it summed in the prologue) instead of eight torch kernels (two compares, and, nonzero's scan, gather, ...).
constexpr int kRowsThreads = 1024;
__device__ __forceinline__ void rows_block_range(int P, int &lo, int &hi)
{
    int per = (P + (int)gridDim.x - 1) / (int)gridDim.x;
    per = (per + kRowsThreads - 1) / kRowsThreads * kRowsThreads;
    const long long l = (long long)blockIdx.x * per;
    lo = l < P ? (int)l : P;
    hi = l + per < P ? (int)(l + per) : P;
}

__global__ __launch_bounds__(kRowsThreads) void k_rows_count(int P, const uint32_t *__restrict__ keys, uint32_t key_max, uint32_t *__restrict__ blk_cnt)
{
    __shared__ uint32_t sh_n;
    if (threadIdx.x == 0) sh_n = 0;
    __syncthreads();
    int lo, hi;
    rows_block_range(P, lo, hi);
    uint32_t mine = 0;
    for (int i = lo + (int)threadIdx.x; i < hi; i += kRowsThreads) mine += keys[i] <= key_max ? 1u : 0u;      // invisible = 0xFFFFFFFF
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine += (uint32_t)__shfl_xor((int)mine, off);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&sh_n, mine);
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = sh_n;
}

__global__ __launch_bounds__(kRowsThreads) void k_rows_gather(int P, const uint32_t *__restrict__ keys, uint32_t key_max,
                                                              const uint32_t *__restrict__ blk_cnt, const float4 *__restrict__ screen,
                                                              int n_rows, int32_t *__restrict__ rows, float4 *__restrict__ packed)
{
    __shared__ uint32_t sh_base, sh_w[kRowsThreads / kWave], sh_pre[kRowsThreads / kWave], sh_tot;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv == 0) {                                               // rows of the blocks in front of this one
        uint32_t s = 0;
        for (int b = lane; b < (int)blockIdx.x; b += kWave) s += blk_cnt[b];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += (uint32_t)__shfl_xor((int)s, off);
        if (lane == 0) sh_base = s;
    }
    __syncthreads();
    uint32_t run = sh_base;
    int lo, hi;
    rows_block_range(P, lo, hi);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int i0 = lo; i0 < hi; i0 += kRowsThreads) {
        const int i = i0 + (int)threadIdx.x;
        const bool take = i < hi && keys[i] <= key_max;
        const unsigned long long m = __ballot(take);
        if (lane == 0) sh_w[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int w = 0; w < kRowsThreads / kWave; ++w) { sh_pre[w] = t; t += sh_w[w]; }
            sh_tot = t;
        }
        __syncthreads();
        if (take) {
            const uint32_t r = run + sh_pre[wv] + (uint32_t)__popcll(m & below);
            if ((int)r < n_rows) {
                rows[r] = i;
                packed[3 * (size_t)r] = screen[3 * (size_t)i]; packed[3 * (size_t)r + 1] = screen[3 * (size_t)i + 1];
                packed[3 * (size_t)r + 2] = screen[3 * (size_t)i + 2];
            }
        }
        run += sh_tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_rows_scatter(int n_rows, int P, const int32_t *__restrict__ rows, const float4 *__restrict__ packed,
                                                      float4 *__restrict__ screen)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    // (entries the gather did not fill stay -1: ranks that disagree on the depth keys must not write out of range)
    if (rows[r] < 0 || rows[r] >= P) return;
    const size_t i = (size_t)rows[r];
    screen[3 * i] = packed[3 * (size_t)r]; screen[3 * i + 1] = packed[3 * (size_t)r + 1]; screen[3 * i + 2] = packed[3 * (size_t)r + 2];
}

int launch_rows_gather(const FrameK &f, GeomWS &ws, uint32_t key_max, const float *screen, int n_rows, int32_t *rows, float *packed,
                       bool debug, hipStream_t s)
{
    if (f.P == 0 || n_rows <= 0) return GSR_OK;
    int blocks = (f.P + kRowsThreads * 4 - 1) / (kRowsThreads * 4);
    if (blocks > kSelBlocks) blocks = kSelBlocks;              // blk_cnt[0] of the frame's selection scratch: idle after the forward
    uint32_t *blk_cnt = &ws.sel->blk_cnt[0][0];
    ProfileScope prof("exchange_rows", s);
    hipLaunchKernelGGL(k_rows_count, dim3(blocks), dim3(kRowsThreads), 0, s, f.P, ws.sort_keys[0], key_max, blk_cnt);
    hipLaunchKernelGGL(k_rows_gather, dim3(blocks), dim3(kRowsThreads), 0, s, f.P, ws.sort_keys[0], key_max, blk_cnt,
                       reinterpret_cast<const float4 *>(screen), n_rows, rows, reinterpret_cast<float4 *>(packed));
    GSR_LAUNCH_CHECK("exchange_rows(gather)", debug, s);
    return GSR_OK;
}

int launch_rows_scatter(int n_rows, int P, const int32_t *rows, const float *packed, float *screen, bool debug, hipStream_t s)
{
    if (n_rows <= 0) return GSR_OK;
    ProfileScope prof("exchange_rows", s);
    hipLaunchKernelGGL(k_rows_scatter, dim3((n_rows + 255) / 256), dim3(256), 0, s, n_rows, P, rows, reinterpret_cast<const float4 *>(packed),
                       reinterpret_cast<float4 *>(screen));
    GSR_LAUNCH_CHECK("exchange_rows(scatter)", debug, s);
    return GSR_OK;
}

// ---- 3. sort of ONE chunk by (depth, index), when it is about to be binned.  Input: order[r0 .. r0 + n) = the chunk's
// Gaussians in index order (the partition above); output: the same range in depth order and, beside it, the inclusive
// scan of their tile counts (offs_full, relative to the chunk's first rank).
//
// Small chunks (the nearest chunk of a saturating frame: 5 K Gaussians at cfg3) are sorted by ONE block in LDS.  One CU
// issues about one wave instruction per SIMD every four cycles, so instructions are what counts: the keys are dealt into
// 4096 buckets by their top bits (one LDS atomic each), the bucket sizes are scanned, and the keys are rank-sorted inside their
// buckets (1.3 keys per bucket on average at cfg3, 3.8 at cfg2's 15 k) by (key, position in the partition = Gaussian index
// order).  The result does not depend on the order the atomics landed in.  Keys that crowd into one bucket (> kBucketMax:
// equal depths) take the 4-bit radix passes below instead (stable, any distribution, ~8x slower).
constexpr int kSmallSortMax = 16384;                          // two instantiations: 8192 keys (8 per thread) and 16384 (16 per thread,
constexpr int kSmallThreads = 1024;                           // some of them spilled: only chunks that need it take that one)
constexpr int kBuckets = 4096;
constexpr int kBucketsPer = kBuckets / kSmallThreads;
constexpr int kBucketBits = 12;
constexpr int kBucketMax = 32;

template <int CAP>
__global__ __launch_bounds__(kSmallThreads) void k_chunk_sort_small(int n, uint32_t r0, int bits, int force_radix,
                                                                    const uint32_t *__restrict__ pos_key,
                                                                    const uint32_t *__restrict__ pos_tiles, uint32_t *__restrict__ order,
                                                                    uint32_t *__restrict__ offs_full)
{
    constexpr int kSmallPer = CAP / kSmallThreads;               // keys per thread
    constexpr int kSmallGroups = CAP / kWave;                    // groups of 64 keys
    constexpr int kSmallPerWave = kSmallGroups / (kSmallThreads / kWave);
    constexpr int kSmallCntPer = 16 * kSmallGroups / kSmallThreads;      // radix fallback: counters per thread in its scan
    __shared__ unsigned long long sc[CAP];                       // fast path: (relative key << 32 | original slot) by bucket order
    uint32_t *sk = reinterpret_cast<uint32_t *>(sc), *sv = sk + CAP;      // fallback: the same memory as two arrays
    constexpr bool kStage = CAP <= 8192;                         // room to keep (Gaussian, tiles) by original slot in LDS
    __shared__ uint32_t sg[kStage ? CAP : 1], st[kStage ? CAP : 1];
    __shared__ uint32_t bcnt[kBuckets];
    __shared__ unsigned short bofs[kBuckets];
    unsigned short *cnt = reinterpret_cast<unsigned short *>(bcnt);      // radix fallback: [digit][group], over the bucket counters
    __shared__ uint32_t sh_wave[kSmallThreads / kWave];
    __shared__ uint32_t sh_max;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // (a key's bucket is recomputed where it is needed, and the keys' places inside their buckets - used only when no bucket holds more
    // than kBucketMax = 32 keys - are kept as bytes, four to a register: 28 registers fewer than three full arrays)
    uint32_t key[kSmallPer], slot4[(kSmallPer + 3) / 4];
    for (int j = threadIdx.x; j < kBuckets; j += kSmallThreads) bcnt[j] = 0;
    if (threadIdx.x == 0) sh_max = force_radix ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int j = 0; j < kSmallPer; ++j) {                        // coalesced: the partition left keys and tile counts by position
        const int e = threadIdx.x + j * kSmallThreads;
        key[j] = 0xFFFFFFFFu;
        if ((j & 3) == 0) slot4[j >> 2] = 0;
        if (e < n) {
            key[j] = pos_key[r0 + e];
            if constexpr (kStage) { sg[e] = order[r0 + e]; st[e] = pos_tiles[r0 + e]; }
        }
    }
    __syncthreads();
    const int bshift = bits > kBucketBits ? bits - kBucketBits : 0;
    auto bucket_of = [&](uint32_t k) { const uint32_t b = k >> bshift; return b < (uint32_t)kBuckets ? b : (uint32_t)kBuckets - 1u; };
#pragma unroll
    for (int j = 0; j < kSmallPer; ++j) {
        const int e = threadIdx.x + j * kSmallThreads;
        if (e < n) {
            slot4[j >> 2] |= min(atomicAdd(&bcnt[bucket_of(key[j])], 1u), 255u) << (8 * (j & 3));
        }
    }
    __syncthreads();
    {   // exclusive scan of the bucket sizes (4 per thread) and their maximum
        uint32_t c[kBucketsPer], mine = 0, mx = 0;
#pragma unroll
        for (int i = 0; i < kBucketsPer; ++i) { c[i] = bcnt[kBucketsPer * threadIdx.x + i]; mine += c[i]; mx = c[i] > mx ? c[i] : mx; }
        uint32_t inc = mine;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)inc, off);
            if (lane >= off) inc += u;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const uint32_t u = (uint32_t)__shfl_xor((int)mx, off); mx = u > mx ? u : mx; }
        if (lane == 63) sh_wave[wv] = inc;
        if (lane == 0) atomicMax(&sh_max, mx);
        __syncthreads();
        uint32_t run = inc - mine;
        for (int i = 0; i < wv; ++i) run += sh_wave[i];
#pragma unroll
        for (int i = 0; i < kBucketsPer; ++i) { bofs[kBucketsPer * threadIdx.x + i] = (unsigned short)run; run += c[i]; }
    }
    __syncthreads();
    const bool fast = sh_max <= (uint32_t)kBucketMax;             // block-uniform
    const uint32_t *sorted_slot = fast ? sk : sv;
    if (fast) {
#pragma unroll
        for (int j = 0; j < kSmallPer; ++j) {
            const int e = threadIdx.x + j * kSmallThreads;
            if (e < n) sc[(uint32_t)bofs[bucket_of(key[j])] + ((slot4[j >> 2] >> (8 * (j & 3))) & 255u)] = ((unsigned long long)key[j] << 32) | (unsigned long long)e;
        }
        __syncthreads();
        // RANK sort inside the buckets, one position per lane: a position's bucket spans at most kBucketMax neighbours on
        // either side, so its rank (= members of its bucket that sort before it) comes from a walk over the offsets
        // -m .. +m (m = the largest bucket among the wave's 64 positions, minus one) with coalesced, independent LDS reads and
        // no divergence.  (Per-thread insertion / rank sorts of whole buckets measured 23 - 39 us here: their dependent,
        // bank-conflicting LDS round trips cost ~200 cycles per step.)
        uint32_t fin[kSmallPerWave];
#pragma unroll
        for (int j = 0; j < kSmallPerWave; ++j) {
            const int grp = wv + (kSmallThreads / kWave) * j;
            fin[j] = 0xFFFFFFFFu;
            if (grp * kWave < n) {                               // wave-uniform
                const int p = grp * kWave + lane;
                const bool mine = p < n;
                const unsigned long long me = mine ? sc[p] : 0ull;
                uint32_t b = (uint32_t)(me >> 32) >> bshift;
                b = b < (uint32_t)kBuckets ? b : (uint32_t)kBuckets - 1u;
                const int lo = mine ? (int)bofs[b] : 0;
                const uint32_t len = mine ? bcnt[b] : 0u;
                int m = mine ? (int)len - 1 : 0;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) { const int u = __shfl_xor(m, off); m = u > m ? u : m; }
                uint32_t r = 0;
                for (int d = -m; d <= m; d += 4) {               // four independent neighbours per step
                    unsigned long long nb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int q = p + d + u;
                        nb[u] = sc[q < 0 ? 0 : (q >= CAP ? CAP - 1 : q)];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        r += ((uint32_t)(p + d + u - lo) < len && d + u <= m && nb[u] < me) ? 1u : 0u;
                }
                if (mine) fin[j] = ((uint32_t)lo + r) | ((uint32_t)me << 16);       // final position (< 8192) | slot (< 8192)
            }
        }
        __syncthreads();                                         // every read of the bucket order is done: the sorted slots go
#pragma unroll                                                   // into the first half of the array (as 32-bit words)
        for (int j = 0; j < kSmallPerWave; ++j)
            if (fin[j] != 0xFFFFFFFFu) sk[fin[j] & 0xFFFFu] = fin[j] >> 16;
        __syncthreads();
    } else {
        // fallback: least-significant-digit radix sort, 4 bits per pass; per pass every 64-key group ranks its keys per digit
        // with four ballots, the (digit, group) counters are scanned in digit-major order, the keys move to their new slots
        const int ngroups = (n + kWave - 1) / kWave;
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int j = 0; j < kSmallPer; ++j) {
            const int e = threadIdx.x + j * kSmallThreads;
            if (e < ngroups * kWave) { sk[e] = key[j]; sv[e] = e < n ? (uint32_t)e : 0xFFFFFFFFu; }      // padding sorts last, stays last
        }
        __syncthreads();
        for (int shift = 0; shift < bits; shift += 4) {
            for (int j = threadIdx.x; j < 16 * kSmallGroups; j += kSmallThreads) cnt[j] = 0;
            __syncthreads();
            uint32_t rk[kSmallPerWave], rv[kSmallPerWave], dr[kSmallPerWave];
#pragma unroll
            for (int j = 0; j < kSmallPerWave; ++j) {
                const int grp = wv + (kSmallThreads / kWave) * j;
                if (grp < ngroups) {                             // wave-uniform
                    const int e = grp * kWave + lane;
                    const uint32_t k = sk[e], d = (k >> shift) & 15u;
                    unsigned long long m = ~0ull;
#pragma unroll
                    for (int bit = 0; bit < 4; ++bit) {
                        const unsigned long long bb = __ballot((d >> bit) & 1u);
                        m &= ((d >> bit) & 1u) ? bb : ~bb;
                    }
                    const uint32_t rank = (uint32_t)__popcll(m & below);
                    if (rank == 0) cnt[d * kSmallGroups + grp] = (unsigned short)__popcll(m);
                    rk[j] = k; rv[j] = sv[e]; dr[j] = d | (rank << 4);
                }
            }
            __syncthreads();
            {   // exclusive scan of the 4096 counters in [digit][group] order, 4 per thread
                uint32_t c[kSmallCntPer], mine = 0;
#pragma unroll
                for (int i = 0; i < kSmallCntPer; ++i) { c[i] = cnt[kSmallCntPer * threadIdx.x + i]; mine += c[i]; }
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
                for (int i = 0; i < kSmallCntPer; ++i) { cnt[kSmallCntPer * threadIdx.x + i] = (unsigned short)run; run += c[i]; }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kSmallPerWave; ++j) {
                const int grp = wv + (kSmallThreads / kWave) * j;
                if (grp < ngroups) {
                    const uint32_t pos = (uint32_t)cnt[(dr[j] & 15u) * kSmallGroups + grp] + (dr[j] >> 4);
                    sk[pos] = rk[j]; sv[pos] = rv[j];
                }
            }
            __syncthreads();
        }
    }
    // the chunk in depth order + the inclusive scan of its tile counts (16 consecutive ranks per thread): Gaussian and tile
    // count come from where the partition left them (by original slot); all reads before the first write
    uint32_t t[kSmallPer], gs[kSmallPer], sum = 0;
#pragma unroll
    for (int i = 0; i < kSmallPer; ++i) {
        const int e = threadIdx.x * kSmallPer + i;
        t[i] = 0; gs[i] = 0;
        if (e < n) {
            const uint32_t src = sorted_slot[e];
            if constexpr (kStage) { gs[i] = sg[src]; t[i] = st[src]; }
            else { gs[i] = order[r0 + src]; t[i] = pos_tiles[r0 + src]; }
        }
        sum += t[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kSmallPer; ++i) {
        const int e = threadIdx.x * kSmallPer + i;
        if (e < n) order[r0 + e] = gs[i];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += v;
    }
    __syncthreads();
    if (lane == 63) sh_wave[wv] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int i = 0; i < wv; ++i) run += sh_wave[i];
#pragma unroll
    for (int i = 0; i < kSmallPer; ++i) {
        const int e = threadIdx.x * kSmallPer + i;
        run += t[i];
        if (e < n) offs_full[r0 + e] = run;
    }
}

static int bits_of(uint32_t v)
{
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

// live_count (device): only the first *live_count Gaussians of the range need sorting (launch_live_filter put the ones that
// can still reach an open tile in front); the rest keep their place behind them
int launch_chunk_order(const FrameK &f, int r0, int r1, uint32_t key_lo, uint32_t key_hi, bool first, GeomWS &ws, bool debug, hipStream_t s,
                       const uint32_t *live_count)
{
    const int n = r1 - r0;
    if (n <= 0) return GSR_OK;
    // the partition left, by position, the Gaussian (order), its key relative to the chunk's lower end (sort_keys[1]) and
    // its tile count (sort_vals[1])
    const uint32_t base = sel_key_base(key_lo, first);
    int bits = bits_of(key_hi > base ? key_hi - base : 0u);
    if (bits < 1) bits = 1;
    if (n <= kSmallSortMax && !live_count) {
        const int force_radix = getenv("GSR_SORT_FORCE_RADIX") ? 1 : 0;               // test hook (tests set it per case): the fallback path of the LDS sort
        ProfileScope prof("chunk_sort", s);
        if (n <= kSmallSortMax / 2)
            hipLaunchKernelGGL(k_chunk_sort_small<kSmallSortMax / 2>, dim3(1), dim3(kSmallThreads), 0, s, n, (uint32_t)r0, bits, force_radix,
                               ws.sort_keys[1], ws.sort_vals[1], ws.order, ws.offs_full);
        else
            hipLaunchKernelGGL(k_chunk_sort_small<kSmallSortMax>, dim3(1), dim3(kSmallThreads), 0, s, n, (uint32_t)r0, bits, force_radix,
                               ws.sort_keys[1], ws.sort_vals[1], ws.order, ws.offs_full);
        GSR_LAUNCH_CHECK("chunk_sort", debug, s);
        return GSR_OK;
    }
    int rc, result = 0;
    // ping-pong between (sort_keys[1], order) and (cnt_open, sort_vals[1]); an even number of passes ends in the first pair
    uint32_t *kbuf[2] = {ws.sort_keys[1] + r0, ws.cnt_open + r0};
    uint32_t *vbuf[2] = {ws.order + r0, ws.sort_vals[1] + r0};
    if ((rc = launch_radix_sort<uint32_t>(kbuf, vbuf, live_count, (uint32_t)n, (uint64_t)n, nullptr, 0, bits, ws.radix_temp, &result, "chunk_sort",
                                          debug, s, /*even_passes=*/true)))
        return rc;
    if (result != 0) { set_error("internal: chunk sort result buffer %d", result); return GSR_ERR_HIP; }
    return launch_scan_inclusive(nullptr, ws.offs_full + r0, n, ws.scan_temp, nullptr, nullptr, nullptr, "scan_tiles", debug, s, nullptr,
                                 ws.order + r0, ws.tiles_mass, ws.mass_blocks);
}

}  // namespace gsr
