// gsr_binning.hip — PROGRESSIVE tile binning (spec: SURVEY A.7; same per-tile order as a global stable
// sort on (tile << 32 | depth bits), i.e. (tile, depth, Gaussian index)).
//
// The reference duplicates every visible Gaussian into every tile of its rectangle and radix-sorts all
// R = sum(tiles touched) 64-bit keys.  On MI355X that sort is pure HBM traffic (24 B x R x passes) and
// most of it is wasted whenever tiles saturate: at the BASELINE cfg3 scene R = 43.8 M but every tile has
// all 256 pixels below the 1e-4 transmittance cut-off after the nearest ~0.3 % of the Gaussians; only
// 1.4 M of the 43.8 M instances can ever touch a pixel.  So:
//
//   1. sort the P Gaussians ONCE by depth (32-bit keys, invisible ones last)           [P-sized]
//   2. inclusive scan of tiles touched in depth order; plan depth chunks by cumulative instance count:
//      the first holds ~384 instances per tile of the slab (at least 1 M, at most R/8), each further one
//      4x more: every chunk costs ~16 small launches (~100 us of fixed time) while an instance costs
//      ~0.1 ns, so few and large chunks win, but tiles saturate after a few hundred splats    [P-sized]
//   3. per chunk: count the instances each Gaussian emits (tiles of its rectangle that are OPEN and that its
//      alpha >= 1/255 ellipse can reach: exact tile culling), scan, emit (tile id, slot) pairs in depth
//      order, stable radix sort on the tile id only (2 passes of 8 bits), tile ranges, blend
//      (gsr_render.hip), recount the open tiles.
//      The host reads back ONE word per chunk (open tiles left) and stops when it is zero.
//
// A tile is closed only when every pixel has taken the A.8 cut-off, after which no further splat can
// change any of its pixels (forward) or receive gradient from them (backward): pixels are identical.
// Payload = slot (absolute emission index, Gaussian-major inside a chunk): the backward writes one
// private gradient row per instance and reduces per Gaussian over a contiguous slot range in fixed
// order - no float atomics, bitwise-reproducible gradients.
#include "gsr_internal.h"

namespace gsr {

constexpr uint32_t kMinFirstChunk = 1u << 20;    // instances in the first depth chunk (at least)
constexpr int kFirstChunkDiv = 8;                // first chunk <= R / 8 ...
constexpr int kFirstChunkPerTile = 384;          // ... and ~384 instances per tile: tiles saturate after a few hundred splats
constexpr int kChunkGrowthLog2 = 2;              // then x4 per chunk

GeomWS carve_geom(void *base, int P)
{
    GeomWS w;
    size_t o = 0;
    char *b = (char *)base;
    const size_t Pn = (size_t)(P > 0 ? P : 1);
    w.records = (float4 *)(b + o); o += align_up(Pn * 48);
    w.tiles_touched = (uint32_t *)(b + o); o += align_up(Pn * 4);
    w.clamped = (uint8_t *)(b + o); o += align_up(Pn);
    for (int i = 0; i < 2; ++i) { w.sort_keys[i] = (uint32_t *)(b + o); o += align_up(Pn * 4); }
    for (int i = 0; i < 2; ++i) { w.sort_vals[i] = (uint32_t *)(b + o); o += align_up(Pn * 4); }
    w.order = w.sort_vals[0];                    // 4 passes of 8 bits: the result lands back in buffer 0
    w.tiles_sorted = (uint32_t *)(b + o); o += align_up(Pn * 4);
    w.offs_full = (uint32_t *)(b + o); o += align_up(Pn * 4);
    w.cnt_open = (uint32_t *)(b + o); o += align_up(Pn * 4);
    w.offs_open = (uint32_t *)(b + o); o += align_up(Pn * 4);
    w.row_begin = (uint32_t *)(b + o); o += align_up(Pn * 4);
    w.ctrl = (Ctrl *)(b + o); o += align_up(sizeof(Ctrl));
    w.scan_temp = b + o; o += scan_temp_bytes((int)Pn);
    w.radix_temp = b + o; o += radix_temp_bytes();
    w.total = o;
    return w;
}

ImageWS carve_image(void *base, const FrameK &f)
{
    ImageWS w;
    size_t o = 0;
    char *b = (char *)base;
    const size_t N = (size_t)f.W * f.H, Tn = (size_t)f.Gx * f.Gy;
    w.T_state = (float *)(b + o); o += align_up((N ? N : 1) * 4);
    w.last_enc = (int32_t *)(b + o); o += align_up((N ? N : 1) * 4);
    w.ranges = (uint2 *)(b + o); o += align_up((Tn ? Tn : 1) * 8 * GSR_MAX_CHUNKS);
    w.open = (uint32_t *)(b + o); o += align_up((Tn ? Tn : 1) * 4);
    w.ctrl_scratch = (Ctrl *)(b + o); o += align_up(sizeof(Ctrl));
    w.total = o;
    return w;
}

BinningWS carve_binning(void *base, int64_t R)
{
    BinningWS w;
    size_t o = 0;
    char *b = (char *)base;
    const size_t Rn = (size_t)(R > 0 ? R : 1);
    for (int i = 0; i < 2; ++i) { w.keys[i] = (uint32_t *)(b + o); o += align_up(Rn * 4); }
    for (int i = 0; i < 2; ++i) { w.vals[i] = (uint32_t *)(b + o); o += align_up(Rn * 4); }
    w.inst_gid = (uint32_t *)(b + o); o += align_up(Rn * 4);
    w.sorted_gid = (uint32_t *)(b + o); o += align_up(Rn * 4);
    w.grad_rows = nullptr;                          // backward-only, sized from instances_emitted: gsr_backward_rows_size
    w.total = o;
    return w;
}

constexpr int kBinBlock = 256;

// ---- depth order: sort (depth bits, Gaussian) pairs written by the preprocess kernel, then gather the
// tile counts into depth order and scan them.
__global__ __launch_bounds__(kBinBlock) void k_gather_tiles(int P, const uint32_t *__restrict__ order,
                                                            const uint32_t *__restrict__ tiles, uint32_t *__restrict__ tiles_sorted,
                                                            uint32_t *__restrict__ cnt_open)
{
    const int r = blockIdx.x * kBinBlock + threadIdx.x;
    if (r >= P) return;
    tiles_sorted[r] = tiles[order[r]];
    cnt_open[r] = 0;
}

int launch_depth_order(const FrameK &f, GeomWS &ws, bool debug, hipStream_t s)
{
    if (f.P == 0) return GSR_OK;
    int result = 0, rc;
    if ((rc = launch_radix_sort<uint32_t>(ws.sort_keys, ws.sort_vals, nullptr, (uint32_t)f.P, (uint64_t)f.P, nullptr, 0, 32,
                                          ws.radix_temp, &result, "depth_sort", debug, s)))
        return rc;
    if (result != 0) { set_error("internal: depth sort result buffer %d", result); return GSR_ERR_HIP; }
    {
        ProfileScope prof("gather_tiles", s);
        hipLaunchKernelGGL(k_gather_tiles, dim3((f.P + kBinBlock - 1) / kBinBlock), dim3(kBinBlock), 0, s, f.P, ws.order,
                           ws.tiles_touched, ws.tiles_sorted, ws.cnt_open);
        GSR_LAUNCH_CHECK("gather_tiles", debug, s);
    }
    return launch_scan_inclusive(ws.tiles_sorted, ws.offs_full, f.P, ws.scan_temp, &ws.ctrl->R_total, nullptr, nullptr, "scan_tiles",
                                 debug, s, &ws.ctrl->overflow);
}

// ---- chunk plan (one block of 9 waves): wave 0 finds V (first rank whose key is 0xFFFFFFFF), wave 1+c the end
// of chunk c (first rank whose inclusive tile count exceeds the chunk's cumulative target: first, 4 first,
// 16 first, ... with the last chunk taking everything that is left).  Each wave runs a 64-ary search, so the
// dependent-load chain is 4 deep at a million Gaussians instead of 20.
template <typename Pred>
__device__ __forceinline__ uint32_t wave_lower_bound(uint32_t n, Pred pred)   // first i in [0,n) with pred(i), else n
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t step = (hi - lo + 63u) >> 6;
        const uint32_t seg = lo + lane * step;
        const bool valid = seg < hi;
        const uint32_t last = valid ? min(hi, seg + step) - 1u : 0u;            // last element of this lane's segment
        const unsigned long long m = __ballot(valid && pred(last));
        if (m == 0ull) return hi;
        const uint32_t k = (uint32_t)__ffsll((long long)m) - 1u;
        hi = __shfl(last, (int)k);
        lo = lo + k * step;
    }
    return lo;
}

__global__ __launch_bounds__(kWave *(GSR_MAX_CHUNKS + 1)) void k_chunk_plan(int P, int n_tiles, const uint32_t *__restrict__ sorted_keys,
                                                                            const uint32_t *__restrict__ offs_full, Ctrl *ctrl)
{
    __shared__ uint32_t sh_V, sh_end[GSR_MAX_CHUNKS];
    const uint32_t R = ctrl->R_total;
    const int w = threadIdx.x >> 6;
    uint32_t first = (uint32_t)kFirstChunkPerTile * (uint32_t)n_tiles;
    if (first > R / (uint32_t)kFirstChunkDiv) first = R / (uint32_t)kFirstChunkDiv;
    if (first < kMinFirstChunk) first = kMinFirstChunk;
    if (w == 0) {
        const uint32_t V = wave_lower_bound((uint32_t)P, [&](uint32_t i) { return sorted_keys[i] == 0xFFFFFFFFu; });
        if ((threadIdx.x & 63) == 0) sh_V = V;
    } else {
        // invisible Gaussians have a tile count of 0, so the scan is flat beyond V: searching [0,P) and clamping
        // to V gives the same answer as searching [0,V)
        const int c = w - 1;
        const uint64_t target = (uint64_t)first << (kChunkGrowthLog2 * c);
        uint32_t end = (uint32_t)P;
        if (c < GSR_MAX_CHUNKS - 1 && target < (uint64_t)R)
            end = wave_lower_bound((uint32_t)P, [&](uint32_t i) { return (uint64_t)offs_full[i] > target; });
        if ((threadIdx.x & 63) == 0) sh_end[c] = end;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const uint32_t V = sh_V;
    ctrl->V = V;
    ctrl->open_count = 0;
    ctrl->bnd[0] = 0;
    ctrl->chunk_base[0] = 0;
    for (int c = 0; c < GSR_MAX_CHUNKS; ++c) {
        ctrl->chunk_full[c] = 0; ctrl->chunk_R[c] = 0; ctrl->bnd[c + 1] = V; ctrl->chunk_base[c + 1] = 0;
    }
    uint32_t begin = 0, nchunks = 0;
    for (int c = 0; c < GSR_MAX_CHUNKS && begin < V; ++c) {
        uint32_t end = sh_end[c];
        if (end <= begin) end = begin + 1;               // every chunk makes progress
        if (end > V) end = V;
        ctrl->bnd[c + 1] = end;
        ctrl->chunk_full[c] = offs_full[end - 1] - (begin ? offs_full[begin - 1] : 0u);
        begin = end;
        nchunks = (uint32_t)c + 1;
    }
    ctrl->num_chunks = nchunks;
}

int launch_chunk_plan(const FrameK &f, GeomWS &ws, bool debug, hipStream_t s)
{
    ProfileScope prof("chunk_plan", s);
    hipLaunchKernelGGL(k_chunk_plan, dim3(1), dim3(kWave * (GSR_MAX_CHUNKS + 1)), 0, s, f.P, (f.ty1 - f.ty0) * f.Gx, ws.sort_keys[0], ws.offs_full, ws.ctrl);
    GSR_LAUNCH_CHECK("chunk_plan", debug, s);
    return GSR_OK;
}

// ---- open flags: (re)initialise for the slab, count the tiles that are still open (one block).
__global__ __launch_bounds__(1024) void k_open_count(FrameK f, int init, uint32_t *__restrict__ open, Ctrl *ctrl)
{
    __shared__ uint32_t sh_count;
    if (threadIdx.x == 0) sh_count = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (int t = threadIdx.x; t < f.Gx * f.Gy; t += blockDim.x) {
        if (init) {
            const int ty = t / f.Gx;
            open[t] = (ty >= f.ty0 && ty < f.ty1) ? 1u : 0u;
        }
        mine += open[t];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sh_count, mine);
    __syncthreads();
    if (threadIdx.x == 0) ctrl->open_count = sh_count;
}

int launch_binning_init(const FrameK &f, GeomWS &gw, ImageWS &iw, bool debug, hipStream_t s)
{
    const size_t Tn = (size_t)f.Gx * f.Gy;
    GSR_HIP_CHECK(hipMemsetAsync(iw.ranges, 0, Tn * sizeof(uint2) * GSR_MAX_CHUNKS, s));
    ProfileScope prof("open_count", s);
    hipLaunchKernelGGL(k_open_count, dim3(1), dim3(1024), 0, s, f, 1, iw.open, gw.ctrl);
    GSR_LAUNCH_CHECK("open_count(init)", debug, s);
    return GSR_OK;
}

int launch_open_update(const FrameK &f, GeomWS &gw, ImageWS &iw, bool debug, hipStream_t s)
{
    ProfileScope prof("open_count", s);
    hipLaunchKernelGGL(k_open_count, dim3(1), dim3(1024), 0, s, f, 0, iw.open, gw.ctrl);
    GSR_LAUNCH_CHECK("open_count", debug, s);
    return GSR_OK;
}

// ---- per chunk: count the instances each Gaussian will emit: tiles of its rectangle that are still open AND
// that its alpha >= 1/255 ellipse can reach (tile_may_contribute: exact culling, no pixel changes).  One WAVE
// per Gaussian, 64 tiles per step, so screen-filling splats do not serialise on a lane.
__device__ __forceinline__ bool instance_wanted(const FrameK &f, const TileRect &t, int w, int i, int total,
                                                const float4 &a, const float4 &b, const uint32_t *__restrict__ open,
                                                uint32_t &tile)
{
    if (i >= total) return false;
    const int tx = t.x0 + i % w, ty = t.y0 + i / w;
    tile = (uint32_t)(ty * f.Gx + tx);
    float A, B, C, op;
    unscale_conic(a.z, a.w, b.x, b.y, A, B, C, op);
    return open[tile] != 0u && tile_may_contribute(a.x, a.y, A, B, C, op, tx, ty);
}

__global__ __launch_bounds__(kBinBlock) void k_count_open(FrameK f, int r0, int r1, const uint32_t *__restrict__ order,
                                                          const float4 *__restrict__ records, const uint32_t *__restrict__ open,
                                                          const Ctrl *__restrict__ ctrl, int chunk, uint32_t *__restrict__ cnt_open)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * kBinBlock + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * kBinBlock) >> 6;
    const bool nothing_open = chunk > 0 && ctrl->open_count == 0u;
    for (int r = r0 + wave; r < r1; r += n_waves) {
        uint32_t cnt = 0;
        if (!nothing_open) {
            const uint32_t g = order[r];
            const float4 a = records[3 * (size_t)g], b = records[3 * (size_t)g + 1], cc = records[3 * (size_t)g + 2];
            const TileRect t = unpack_rect(cc.z, cc.w);
            const int w = t.x1 - t.x0, total = w * (t.y1 - t.y0);
            for (int i0 = 0; i0 < total; i0 += kWave) {
                uint32_t tile;
                cnt += (uint32_t)__popcll(__ballot(instance_wanted(f, t, w, i0 + lane, total, a, b, open, tile)));
            }
        }
        if (lane == 0) cnt_open[r] = cnt;
    }
}

// ---- emit (tile, slot) pairs of the chunk's Gaussians in depth order, open tiles only.  One WAVE walks one
// Gaussian's rectangle cooperatively (ballot + popcount gives each open tile its ordinal), so a splat that
// covers thousands of tiles does not serialise on one lane; 4 Gaussians per 256-thread block per step.
__global__ __launch_bounds__(kBinBlock) void k_emit(FrameK f, int c, int r0, int r1, const uint32_t *__restrict__ order,
                                                    const float4 *__restrict__ records, const uint32_t *__restrict__ open,
                                                    const uint32_t *__restrict__ cnt_open, const uint32_t *__restrict__ offs_open,
                                                    const Ctrl *__restrict__ ctrl, uint32_t *__restrict__ keys,
                                                    uint32_t *__restrict__ vals, uint32_t *__restrict__ inst_gid,
                                                    uint32_t *__restrict__ row_begin)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * kBinBlock + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * kBinBlock) >> 6;
    const uint32_t base = ctrl->chunk_base[c];
    if (ctrl->chunk_R[c] == 0u) {                       // nothing to emit (e.g. a speculatively enqueued chunk)
        for (int r = r0 + blockIdx.x * kBinBlock + threadIdx.x; r < r1; r += gridDim.x * kBinBlock) row_begin[r] = base;
        return;
    }
    for (int r = r0 + wave; r < r1; r += n_waves) {
        const uint32_t cnt = cnt_open[r];
        const uint32_t first = base + offs_open[r] - cnt;
        if (lane == 0) row_begin[r] = first;
        if (cnt == 0) continue;
        const uint32_t g = order[r];
        const float4 a = records[3 * (size_t)g], b = records[3 * (size_t)g + 1], cc = records[3 * (size_t)g + 2];
        const TileRect t = unpack_rect(cc.z, cc.w);
        const int w = t.x1 - t.x0, total = w * (t.y1 - t.y0);
        uint32_t emitted = 0;
        for (int i0 = 0; i0 < total; i0 += kWave) {
            uint32_t tile = 0;
            const bool is_open = instance_wanted(f, t, w, i0 + lane, total, a, b, open, tile);
            const unsigned long long m = __ballot(is_open);
            if (is_open) {
                const uint32_t slot = first + emitted + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                keys[slot] = tile;
                vals[slot] = slot;
                inst_gid[slot] = g;
            }
            emitted += (uint32_t)__popcll(m);
        }
    }
}

// ---- ranges of the chunk's sorted list (grid-stride, element count on the device)
__global__ __launch_bounds__(kBinBlock) void k_ranges(int c, const Ctrl *__restrict__ ctrl, const uint32_t *__restrict__ keys,
                                                      const uint32_t *__restrict__ slots, const uint32_t *__restrict__ inst_gid,
                                                      uint2 *__restrict__ ranges, uint32_t *__restrict__ sorted_gid)
{
    const uint32_t n = ctrl->chunk_R[c], base = ctrl->chunk_base[c];
    for (uint32_t i = blockIdx.x * kBinBlock + threadIdx.x; i < n; i += gridDim.x * kBinBlock) {
        const uint32_t a = base + i;
        const uint32_t tile = keys[a];
        if (i == 0) ranges[tile].x = a;
        else {
            const uint32_t prev = keys[a - 1];
            if (prev != tile) { ranges[prev].y = a; ranges[tile].x = a; }
        }
        if (i == n - 1) ranges[tile].y = a + 1;
        sorted_gid[a] = inst_gid[slots[a]];
    }
}

static int msb_plus1(uint32_t n)
{
    int b = 0;
    while (n) { ++b; n >>= 1; }
    return b;
}

int launch_chunk_binning(const FrameK &f, int c, int r0, int r1, uint64_t n_max, GeomWS &gw, BinningWS &bw, ImageWS &iw,
                         int *sort_result, bool debug, hipStream_t s)
{
    int rc;
    const int n = r1 - r0;
    if (n <= 0) return GSR_OK;
    const size_t Tn = (size_t)f.Gx * f.Gy;
    {
        ProfileScope prof("count_open", s);
        int cblocks = (n + 3) / 4;                    // one wave per Gaussian per step
        if (cblocks > 4096) cblocks = 4096;
        hipLaunchKernelGGL(k_count_open, dim3(cblocks), dim3(kBinBlock), 0, s, f, r0, r1, gw.order, gw.records, iw.open,
                           gw.ctrl, c, gw.cnt_open);
        GSR_LAUNCH_CHECK("count_open", debug, s);
    }
    if ((rc = launch_scan_inclusive(gw.cnt_open + r0, gw.offs_open + r0, n, gw.scan_temp, &gw.ctrl->chunk_R[c],
                                    &gw.ctrl->chunk_base[c], &gw.ctrl->chunk_base[c + 1], "scan_open", debug, s)))
        return rc;
    {
        ProfileScope prof("emit", s);
        int blocks = (n + 3) / 4;                     // one wave per Gaussian per step
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_emit, dim3(blocks), dim3(kBinBlock), 0, s, f, c, r0, r1, gw.order, gw.records, iw.open, gw.cnt_open,
                           gw.offs_open, gw.ctrl, bw.keys[0], bw.vals[0], bw.inst_gid, gw.row_begin);
        GSR_LAUNCH_CHECK("emit", debug, s);
    }
    const int tile_bits = msb_plus1((uint32_t)(Tn ? Tn - 1 : 0));
    if ((rc = launch_radix_sort<uint32_t>(bw.keys, bw.vals, &gw.ctrl->chunk_R[c], 0, n_max, &gw.ctrl->chunk_base[c], 0,
                                          tile_bits > 0 ? tile_bits : 1, gw.radix_temp, sort_result, "tile_sort", debug, s)))
        return rc;
    {
        ProfileScope prof("ranges", s);
        uint64_t blocks = (n_max + kBinBlock - 1) / kBinBlock;
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_ranges, dim3((unsigned)blocks), dim3(kBinBlock), 0, s, c, gw.ctrl, bw.keys[*sort_result],
                           bw.vals[*sort_result], bw.inst_gid, iw.ranges + (size_t)c * Tn, bw.sorted_gid);
        GSR_LAUNCH_CHECK("ranges", debug, s);
    }
    return GSR_OK;
}

}  // namespace gsr
