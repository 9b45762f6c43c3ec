// gsr_binning.hip — PROGRESSIVE tile binning (spec: SURVEY A.7; same per-tile order as a global stable
// sort on (tile << 32 | depth bits), i.e. (tile, depth, Gaussian index)).
//
// The reference duplicates every visible Gaussian into every tile of its rectangle and radix-sorts all
// R = sum(tiles touched) 64-bit keys.  On MI355X that sort is pure HBM traffic (24 B x R x passes) and
// most of it is wasted whenever tiles saturate: at the BASELINE cfg3 scene R = 43.8 M but every tile has
// all 256 pixels below the 1e-4 transmittance cut-off after the nearest ~0.3 % of the Gaussians; only
// 1.4 M of the 43.8 M instances can ever touch a pixel.  So:
//
//   1. select depth chunks WITHOUT sorting (gsr_select.hip): a histogram of the depth keys carries tile counts and
//      OPTICAL MASS per bin; a pixel takes the transmittance cut-off once the optical depths in front of it sum to
//      ln(1e4) = 9.21, so the first chunk ends at the key where the frame's mean optical depth (running sum of the
//      splats' optical masses / slab pixels, gsr_math.h optical_mass) reaches kChunkOpticalDepths x 9.21, each further
//      one at 4x more; a frame whose total never gets there (a scene that does not saturate) is ONE chunk.  Every
//      chunk costs ~16 small launches (~100 us of fixed time) while an instance costs ~0.1 ns, so few and large
//      chunks win.  A stable partition groups the Gaussians by chunk                                       [P-sized]
//   2. per chunk: sort ITS Gaussians by (depth, index) (one block in LDS, or the radix sort), count the instances
//      each Gaussian emits (tiles of its rectangle that are OPEN and that its alpha >= 1/255 ellipse can reach: exact
//      tile culling), scan, then the per-tile lists in depth order: by GATHER for chunks of few large splats (one
//      wave per tile collects its ranks), else emit (tile id, slot) pairs in depth order + stable radix sort on the
//      tile id only + tile ranges; blend (gsr_render.hip), recount the open tiles.  Late chunks of frames that never
//      close first go through the live filter (below).
//      The host reads back ONE word per chunk (open tiles left) and stops when it is zero.
//
// A tile is closed only when every pixel has taken the A.8 cut-off, after which no further splat can
// change any of its pixels (forward) or receive gradient from them (backward): pixels are identical.
// Payload = slot (absolute emission index, Gaussian-major inside a chunk): the backward writes one
// private gradient row per instance and reduces per Gaussian over a contiguous slot range in fixed
// order - no float atomics, bitwise-reproducible gradients.
#include "gsr_internal.h"

namespace gsr {

// Chunk boundaries come from the frame itself.  Chunk c ends at the first depth rank where the mean optical depth per
// slab pixel exceeds kChunkOpticalDepths x ln(1e4) x 4^c.  Where pixels take the cut-off relative to that mean is a
// property of the blend rule, not of a scene: with splats scattered uniformly over the frame, half the pixels have
// closed at a mean of 1.9 cut-off depths, 99 % at 3.3 and the last one at 4.0-4.3 (measured with the oracle on cfg2,
// cfg3, a second seed and off-axis cameras: the mass counts the faint skirts below alpha = 1/255 that the blend
// skips, hence > 1).  5 leaves a margin; regions the splats cover unevenly simply close in the next, 4x larger, chunk.
// (kCutoffOpticalDepth, kChunkOpticalDepths, kMinFirstChunk — a chunk's fixed cost is worth a few hundred thousand
// instances — and kChunkGrowthLog2 live in gsr_internal.h: the plan is made in gsr_select.hip.)

GeomWS carve_geom(void *base, int P)
{
    GeomWS w;
    size_t o = 0;
    char *b = (char *)base;
    const size_t Pn = (size_t)(P > 0 ? P : 1);
    w.records = (float4 *)(b + o); o += align_up(Pn * 48);
    w.tiles_mass = (uint2 *)(b + o); o += align_up(Pn * 8);
    w.mass_blocks = (unsigned long long *)(b + o); o += align_up(((Pn + kScanTileElems - 1) / kScanTileElems + 1) * 8);
    w.sel = (SelState *)(b + o); o += align_up(sizeof(SelState));
    w.clamped = (uint8_t *)(b + o); o += align_up(Pn);
    for (int i = 0; i < 2; ++i) { w.sort_keys[i] = (uint32_t *)(b + o); o += align_up(Pn * 4); }
    for (int i = 0; i < 2; ++i) { w.sort_vals[i] = (uint32_t *)(b + o); o += align_up(Pn * 4); }
    w.order = w.sort_vals[0];
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
    w.tile_cnt = (uint32_t *)(b + o); o += align_up((Tn ? Tn : 1) * 4);      // directly behind the ranges: one memset clears both
    w.unit_count = (uint32_t *)(b + o); o += align_up(kUnitShards * kUnitClasses * 4);      // ... and these
    w.open = (uint32_t *)(b + o); o += align_up((Tn ? Tn : 1) * 4);
    w.open_bits = (unsigned long long *)(b + o); o += align_up((size_t)(f.Gy > 0 ? f.Gy : 1) * (size_t)((f.Gx + 63) / 64 + 1) * 8);
    w.ctrl_scratch = (Ctrl *)(b + o); o += align_up(sizeof(Ctrl));
    w.tile_walk = (uint32_t *)(b + o); o += align_up((Tn ? Tn : 1) * 4 * GSR_MAX_CHUNKS);
    w.ckpt_start = (float *)(b + o); o += align_up((Tn ? Tn : 1) * (size_t)(GSR_MAX_CHUNKS - 1) * kCkptFloats * sizeof(float));
    w.total = o;
    return w;
}

BinningWS carve_binning(void *base, int64_t R, const FrameK &f)
{
    BinningWS w;
    size_t o = 0;
    char *b = (char *)base;
    const size_t Rn = (size_t)(R > 0 ? R : 1), Tn = (size_t)f.Gx * f.Gy;
    for (int i = 0; i < 2; ++i) { w.keys[i] = (uint32_t *)(b + o); o += align_up(Rn * 4); }
    for (int i = 0; i < 2; ++i) { w.vals[i] = (uint32_t *)(b + o); o += align_up(Rn * 4); }
    for (int i = 0; i < 2; ++i) { w.gids[i] = (uint32_t *)(b + o); o += align_up(Rn * 4); }
    w.row_valid = (uint8_t *)(b + o); o += align_up(Rn);
    w.ckpt = (float *)(b + o); o += align_up((Rn / kSeg + 2) * (size_t)kCkptFloats * sizeof(float));
    {
        const size_t cf = Rn / kSeg + 1, cp = (size_t)GSR_MAX_CHUNKS * (Tn / kUnitShards + 1);
        w.units.cap_full = (uint32_t)(cf > 0xFFFFFFFFull ? 0xFFFFFFFFull : cf);
        w.units.cap_part = (uint32_t)cp;
        w.units.units = (uint2 *)(b + o); o += align_up(kUnitShards * w.units.shard_stride() * sizeof(uint2));
    }
    w.grad_rows = nullptr;                          // backward-only, sized from instances_emitted: gsr_backward_rows_size
    w.total = o;
    return w;
}

// i / w and i % w for tile counts (both < 2^24: gsr_forward_preprocess rejects frames with more tiles): the float quotient with the
// precomputed reciprocal is off by at most one; ~10 instructions instead of the ~45 of the 32-bit division sequence, per candidate
__device__ __forceinline__ void divmod_tiles(uint32_t i, uint32_t w, float inv_w, int &q, int &r)
{
    int qq = (int)((float)i * inv_w);
    int rr = (int)i - qq * (int)w;
    if (rr < 0) { --qq; rr += (int)w; }
    else if (rr >= (int)w) { ++qq; rr -= (int)w; }
    q = qq; r = rr;
}

constexpr int kBinBlock = 256;

// ---- open flags: (re)initialise for the slab, count the tiles that are still open and pack the flags into one
// bit per tile (row-major, ceil(Gx/64) words per tile row) for the count / emit kernels (one block).
__global__ __launch_bounds__(1024) void k_open_count(FrameK f, int init, uint32_t *__restrict__ open,
                                                     unsigned long long *__restrict__ open_bits, Ctrl *ctrl, CtrlMirror mirror)
{
    __shared__ uint32_t sh_count, sh_stuck;
    if (threadIdx.x == 0) { sh_count = 0; sh_stuck = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int W64 = (f.Gx + 63) >> 6;
    uint32_t mine = 0, stuck = 0;
    for (int j = wave; j < f.Gy * W64; j += n_waves) {        // one wave per (tile row, 64-tile word)
        const int ty = j / W64, tx = (j - ty * W64) * 64 + lane;
        uint32_t o = 0;
        if (tx < f.Gx) {
            const int t = ty * f.Gx + tx;
            if (init) { o = (ty >= f.ty0 && ty < f.ty1) ? 1u : 0u; open[t] = o; }
            else o = open[t];
        }
        const unsigned long long m = __ballot(o != 0u), m2 = __ballot(o == 2u);
        if (lane == 0) { open_bits[j] = m; mine += (uint32_t)__popcll(m); stuck += (uint32_t)__popcll(m2); }
    }
    if (lane == 0 && mine) atomicAdd(&sh_count, mine);
    if (lane == 0 && stuck) atomicAdd(&sh_stuck, stuck);
    __syncthreads();
    if (threadIdx.x == 0) { ctrl->open_count = sh_count; ctrl->open_stuck = sh_stuck; }
    // This kernel is the last one in front of both host decisions of a frame (the plan; "did the frame close"): it hands the
    // control block to the host itself, through host-coherent memory, instead of a device-to-host copy + event behind it.
    // Every field but the two above was written by earlier kernels; the flag word goes last, released at system scope.
    if (mirror.words) {
        constexpr int kWords = (int)(sizeof(Ctrl) / 4);
        static_assert(sizeof(Ctrl) % 4 == 0 && kWords <= 1024, "one thread per word");
        if ((int)threadIdx.x < kWords) {
            uint32_t v = reinterpret_cast<const uint32_t *>(ctrl)[threadIdx.x];
            if (threadIdx.x == offsetof(Ctrl, open_count) / 4) v = sh_count;
            if (threadIdx.x == offsetof(Ctrl, open_stuck) / 4) v = sh_stuck;
            mirror.words[threadIdx.x] = v;
            __threadfence_system();
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&mirror.words[kWords], mirror.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

size_t binning_clear_bytes(const FrameK &f, const ImageWS &iw)
{
    // the per-chunk tile ranges and, directly behind them, the per-tile counters with their padding (carve_image: both are
    // 256-byte aligned blocks, so the size is a multiple of 16)
    const size_t Tn = (size_t)f.Gx * f.Gy;
    (void)Tn;
    return (size_t)((char *)iw.unit_count - (char *)iw.ranges) + align_up(kUnitShards * kUnitClasses * sizeof(uint32_t));
}

int launch_binning_init(const FrameK &f, GeomWS &gw, ImageWS &iw, bool debug, hipStream_t s, bool ranges_cleared, CtrlMirror mirror)
{
    if (!ranges_cleared) GSR_HIP_CHECK(hipMemsetAsync(iw.ranges, 0, binning_clear_bytes(f, iw), s));
    ProfileScope prof("open_count", s);
    hipLaunchKernelGGL(k_open_count, dim3(1), dim3(1024), 0, s, f, 1, iw.open, iw.open_bits, gw.ctrl, mirror);
    GSR_LAUNCH_CHECK("open_count(init)", debug, s);
    return GSR_OK;
}

int launch_open_update(const FrameK &f, GeomWS &gw, ImageWS &iw, bool debug, hipStream_t s, CtrlMirror mirror)
{
    ProfileScope prof("open_count", s);
    hipLaunchKernelGGL(k_open_count, dim3(1), dim3(1024), 0, s, f, 0, iw.open, iw.open_bits, gw.ctrl, mirror);
    GSR_LAUNCH_CHECK("open_count", debug, s);
    return GSR_OK;
}

// ---- variant A (chunks of few, large splats — the nearest ones): a TEAM of W waves (one block) per Gaussian walks
// its rectangle 64 tiles per wave step.  A tile takes an instance iff it is still open AND the splat's
// alpha >= 1/255 ellipse can reach it (tile_may_contribute: exact culling, no pixel changes).  The count pass
// keeps each step's 64-bit acceptance mask (scratch in the idle radix buffer), so the emit pass evaluates nothing:
// it scans the masks' popcounts and expands them into (tile, slot, Gaussian) triples in rectangle order.
__device__ __forceinline__ uint32_t mask_base(const uint32_t *__restrict__ offs_full, int r0, int r, uint32_t total)
{
    const uint32_t start = offs_full[r] - total;               // first candidate of rank r inside the chunk (the scan restarts per chunk)
    return (start >> 6) + (uint32_t)(r - r0);                  // disjoint ranges of ceil(total / 64) words per rank
}

// GATHER: the chunk's tile lists are built by k_tile_gather (below) instead of emit + sort: the count pass then also
// counts the instances per tile (atomics on consecutive counters: a wave step covers runs of consecutive tiles),
// keeps the running popcount in front of every mask word (the instance's ordinal inside its Gaussian = its gradient
// row) and one 32-byte metadata row per depth rank (rectangle, mask base, Gaussian, alpha bounding box).
template <int W, bool GATHER>
__global__ __launch_bounds__(W *kWave) void k_count_team(FrameK f, int c, int r0, const uint32_t *__restrict__ order,
                                                         const float4 *__restrict__ records,
                                                         const unsigned long long *__restrict__ open_bits,
                                                         const Ctrl *__restrict__ ctrl, const uint32_t *__restrict__ offs_full,
                                                         unsigned long long *__restrict__ masks, uint32_t *__restrict__ cnt_open,
                                                         uint32_t *__restrict__ tile_cnt, uint32_t *__restrict__ wprefix,
                                                         uint4 *__restrict__ meta_a, float4 *__restrict__ meta_b, int n_total)
{
    constexpr int T = W * kWave;
    __shared__ uint32_t sh_cnt[W];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // one block per Gaussian; a chunk the live filter went through is walked by a fixed grid instead (n_total blocks would mostly
    // be Gaussians that cannot emit: their counts were cleared by the launcher, and the walk stops at the last live one)
    const uint32_t live = ctrl->chunk_live[c];
    for (int rk = (int)blockIdx.x; rk < n_total; rk += (int)gridDim.x) {
    const int r = r0 + rk;
    if ((uint32_t)rk >= live) break;
    if (c > 0 && ctrl->open_count == 0u) {                     // a speculatively enqueued chunk: nothing is open
        if (threadIdx.x == 0) cnt_open[r] = 0;
        continue;
    }
    const uint32_t g = order[r];
    const float4 a = records[3 * (size_t)g], b = records[3 * (size_t)g + 1], cc = records[3 * (size_t)g + 2];
    const TileRect t = unpack_rect(cc.z, cc.w);
    const int w = t.x1 - t.x0, total = w * (t.y1 - t.y0), ns = (total + kWave - 1) >> 6;
    const float inv_w = 1.f / (float)w;
    const uint32_t mb = mask_base(offs_full, r0, r, (uint32_t)total);
    const int W64 = (f.Gx + 63) >> 6;
    float A, B, C, op;
    unscale_conic(a.z, a.w, b.x, b.y, A, B, C, op);
    uint32_t local = 0;
    for (int s = wv; s < ns; s += W) {
        const int i = s * kWave + lane;
        bool want = false;
        if (i < total) {
            int qy, rx;
            divmod_tiles((uint32_t)i, (uint32_t)w, inv_w, qy, rx);
            const int tx = t.x0 + rx, ty = t.y0 + qy;
            want = ((open_bits[ty * W64 + (tx >> 6)] >> (tx & 63)) & 1ull) && tile_may_contribute(a.x, a.y, A, B, C, op, tx, ty);
            if (GATHER && want) atomicAdd(&tile_cnt[ty * f.Gx + tx], 1u);
        }
        const unsigned long long m = __ballot(want);
        if (lane == 0) masks[mb + (uint32_t)s] = m;
        local += (uint32_t)__popcll(m);
    }
    if (W > 1) {
        if (lane == 0) sh_cnt[wv] = local;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t sum = 0;
#pragma unroll
            for (int i = 0; i < W; ++i) sum += sh_cnt[i];
            cnt_open[r] = sum;
        }
    } else if (lane == 0) cnt_open[r] = local;
    if constexpr (GATHER) {
        if (threadIdx.x == 0) {
            float xe, ye;
            splat_extent_q(a.z, a.w, b.x, b.y, xe, ye);
            meta_a[r - r0] = make_uint4((uint32_t)t.x0 | ((uint32_t)t.y0 << 16), (uint32_t)w | ((uint32_t)(t.y1 - t.y0) << 16), mb, g);
            meta_b[r - r0] = make_float4(a.x, a.y, xe, ye);
        }
        __syncthreads();                                       // the block's mask words are visible to all its threads
        uint32_t carry = 0;
        for (int s0 = 0; s0 < ns; s0 += T) {                   // exclusive scan of the popcounts of steps [s0, s0 + T)
            const int s = s0 + (int)threadIdx.x;
            const uint32_t pc = s < ns ? (uint32_t)__popcll(masks[mb + (uint32_t)s]) : 0u;
            uint32_t inc = pc;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const uint32_t v = __shfl_up(inc, off);
                if (lane >= off) inc += v;
            }
            uint32_t wbase = 0, wtot = inc;
            if (W > 1) {
                __syncthreads();
                if (lane == 63) sh_cnt[wv] = inc;
                __syncthreads();
                wtot = 0;
#pragma unroll
                for (int i = 0; i < W; ++i) { const uint32_t v = sh_cnt[i]; if (i < wv) wbase += v; wtot += v; }
            } else wtot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            if (s < ns) wprefix[mb + (uint32_t)s] = carry + wbase + inc - pc;
            carry += wtot;
        }
    }
    if (W > 1 || GATHER) __syncthreads();                   // (LDS of this Gaussian is done with before the next one)
    }
}

template <int W>
__global__ __launch_bounds__(W *kWave) void k_emit_team(FrameK f, int c, int r0, const uint32_t *__restrict__ order,
                                                        const float4 *__restrict__ records, const Ctrl *__restrict__ ctrl,
                                                        const uint32_t *__restrict__ offs_full,
                                                        const unsigned long long *__restrict__ masks,
                                                        const uint32_t *__restrict__ cnt_open, const uint32_t *__restrict__ offs_open,
                                                        uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                        uint32_t *__restrict__ inst_gid, uint32_t *__restrict__ row_begin, int n_total)
{
    constexpr int T = W * kWave;
    __shared__ unsigned long long sh_m[T];
    __shared__ uint32_t sh_off[T];
    __shared__ uint32_t sh_wave[W];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t base = ctrl->chunk_base[c];
    const bool nothing = ctrl->chunk_R[c] == 0u;
    const uint32_t live = ctrl->chunk_live[c];                   // (see k_count_team: a fixed grid walks a filtered chunk)
    for (int rk = (int)blockIdx.x; rk < n_total; rk += (int)gridDim.x) {
    if ((uint32_t)rk >= live) break;
    const int r = r0 + rk;
    const uint32_t cnt = nothing ? 0u : cnt_open[r];
    const uint32_t first = base + (nothing ? 0u : offs_open[r] - cnt);
    if (threadIdx.x == 0) row_begin[r] = first;
    if (cnt == 0u) continue;
    const uint32_t g = order[r];
    const float4 ra = records[3 * (size_t)g], rb = records[3 * (size_t)g + 1], cc = records[3 * (size_t)g + 2];
    const TileRect t = unpack_rect(cc.z, cc.w);
    const int w = t.x1 - t.x0, total = w * (t.y1 - t.y0), ns = (total + kWave - 1) >> 6;
    const float inv_w = 1.f / (float)w;
    const uint32_t mb = mask_base(offs_full, r0, r, (uint32_t)total);
    float xe, ye;                                  // large splats: the quadrant mask from the alpha >= 1/255 bounding box (cheap;
    splat_extent_q(ra.z, ra.w, rb.x, rb.y, xe, ye);    // their tiles are nearly all fully covered); small ones get the exact test
    uint32_t carry = 0;
    for (int s0 = 0; s0 < ns; s0 += T) {
        // exclusive scan of the popcounts of steps [s0, s0 + T)
        const int s = s0 + (int)threadIdx.x;
        const unsigned long long m = s < ns ? masks[mb + (uint32_t)s] : 0ull;
        const uint32_t pc = (uint32_t)__popcll(m);
        uint32_t inc = pc;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t v = __shfl_up(inc, off);
            if (lane >= off) inc += v;
        }
        if (lane == 63) sh_wave[wv] = inc;
        __syncthreads();
        uint32_t wbase = 0, wtot = 0;
#pragma unroll
        for (int i = 0; i < W; ++i) { const uint32_t v = sh_wave[i]; if (i < wv) wbase += v; wtot += v; }
        sh_m[threadIdx.x] = m;
        sh_off[threadIdx.x] = carry + wbase + inc - pc;
        __syncthreads();
        // expansion: one wave per step
        const int nq = min(T, ns - s0);
        for (int q = wv; q < nq; q += W) {
            const unsigned long long mq = sh_m[q];
            if ((mq >> lane) & 1ull) {
                const int i = (s0 + q) * kWave + lane;
                int qy, rx;
                divmod_tiles((uint32_t)i, (uint32_t)w, inv_w, qy, rx);
                const int tx = t.x0 + rx, ty = t.y0 + qy;
                const uint32_t slot = first + sh_off[q] + (uint32_t)__popcll(mq & ((1ull << lane) - 1ull));
                keys[slot] = (uint32_t)(ty * f.Gx + tx);
                vals[slot] = slot;
                // sub-tile culling for the blend kernels: which 8x8 quadrants of this tile the splat can reach
                inst_gid[slot] = g | (quadrant_mask_bbox(ra.x, ra.y, xe, ye, (float)(tx * GSR_TILE), (float)(ty * GSR_TILE)) << kQuadMaskShift);
            }
        }
        carry += wtot;
        __syncthreads();
    }
    }
}

// ---- variant B (chunks of many, small splats): count, then emit, the instances of the chunk's Gaussians: tiles of a Gaussian's rectangle that
// are still open AND that its alpha >= 1/255 ellipse can reach (tile_may_contribute: exact culling, no pixel
// changes).  One wave takes 64 consecutive depth ranks.
//   Phase 1, one LANE per Gaussian: coalesced rank metadata, the record gathered into LDS (64 gathers in flight),
//   quick reject of Gaussians whose rectangle holds no open tile (bit table staged in LDS).
//   Phase 2, FLATTENED: the (Gaussian, tile) candidates of the 64 Gaussians form one list (prefix sum of the
//   rectangle sizes); every step hands 64 consecutive candidates to the 64 lanes (binary search in the prefix),
//   so a 2x2-tile splat costs 4 lanes, not a whole wave step, and a screen-filling one spreads over many steps.
//   A Gaussian's candidates occupy consecutive lanes of a step: ballot + popcount over that lane segment give
//   each accepted instance its ordinal; the segment's first lane carries the running count in LDS.
constexpr int kBitsMaxWords = 2048;          // LDS copy of the open-bit table: up to 16 KB (e.g. 256 tile rows x 8 words)
constexpr int kBinWaves = kBinBlock / kWave;

__device__ __forceinline__ bool rect_has_open_tile(const TileRect &t, const unsigned long long *bits, int W64)
{
    if (t.x1 <= t.x0) return false;
    const int w0 = t.x0 >> 6, w1 = (t.x1 - 1) >> 6;
    for (int y = t.y0; y < t.y1; ++y)
        for (int w = w0; w <= w1; ++w) {
            unsigned long long m = ~0ull;
            if (w == w0) m &= ~0ull << (t.x0 & 63);
            if (w == w1) m &= ~0ull >> (63 - ((t.x1 - 1) & 63));
            if (bits[y * W64 + w] & m) return true;
        }
    return false;
}

template <bool EMIT>
__global__ __launch_bounds__(kBinBlock) void k_bin_chunk(FrameK f, int c, int r0, int r1, const uint32_t *__restrict__ order,
                                                         const float4 *__restrict__ records,
                                                         const unsigned long long *__restrict__ open_bits,
                                                         const Ctrl *__restrict__ ctrl, uint32_t *__restrict__ cnt_open,
                                                         const uint32_t *__restrict__ offs_open, uint32_t *__restrict__ keys,
                                                         uint32_t *__restrict__ vals, uint32_t *__restrict__ inst_gid,
                                                         uint32_t *__restrict__ row_begin)
{
    __shared__ unsigned long long sh_bits[kBitsMaxWords];
    __shared__ float4 sh_a[kBinWaves][kWave];          // x, y, qA, qB
    __shared__ float4 sh_b[kBinWaves][kWave];          // qC, lop, (x0 | y0 << 16), rectangle width
    __shared__ uint32_t sh_end[kBinWaves][kWave];      // inclusive prefix of the rectangle sizes
    __shared__ uint32_t sh_cnt[kBinWaves][kWave];      // accepted instances so far, per Gaussian
    __shared__ uint32_t sh_first[kBinWaves][kWave];    // EMIT: first slot of the Gaussian
    __shared__ uint32_t sh_gid[kBinWaves][kWave];      // EMIT: Gaussian index
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int W64 = (f.Gx + 63) >> 6, n_words = f.Gy * W64;
    const bool in_lds = n_words <= kBitsMaxWords;
    if (in_lds) {
        for (int j = threadIdx.x; j < n_words; j += kBinBlock) sh_bits[j] = open_bits[j];
        __syncthreads();
    }
    const unsigned long long *bits = in_lds ? sh_bits : open_bits;
    const int r = r0 + ((blockIdx.x * kBinBlock + threadIdx.x) >> 6) * kWave + lane;       // this lane's depth rank
    const bool mine = r < r1;
    if (r - lane >= r1) return;                                                            // wave-uniform
    // ---- phase 1
    uint32_t total = 0, first = 0, g = 0;
    bool alive = false;
    if constexpr (EMIT) {
        const uint32_t base = ctrl->chunk_base[c];
        if (mine) {
            const uint32_t cnt = ctrl->chunk_R[c] == 0u ? 0u : cnt_open[r];
            first = base + (ctrl->chunk_R[c] == 0u ? 0u : offs_open[r] - cnt);
            row_begin[r] = first;
            alive = cnt != 0u;
        }
    } else {
        alive = mine && !(c > 0 && ctrl->open_count == 0u) && (uint32_t)(r - r0) < ctrl->chunk_live[c];
    }
    if (alive) {
        g = order[r];
        const float4 cc = records[3 * (size_t)g + 2];
        const TileRect t = unpack_rect(cc.z, cc.w);
        if constexpr (!EMIT) alive = rect_has_open_tile(t, bits, W64);
        if (alive) {
            const float4 b = records[3 * (size_t)g + 1];
            const int w = t.x1 - t.x0;
            total = (uint32_t)(w * (t.y1 - t.y0));
            sh_a[wv][lane] = records[3 * (size_t)g];
            sh_b[wv][lane] = make_float4(b.x, b.y, __uint_as_float((uint32_t)t.x0 | ((uint32_t)t.y0 << 16)), __uint_as_float((uint32_t)w));
        }
    }
    uint32_t end = total;                              // wave inclusive prefix sum
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t v = __shfl_up(end, off);
        if (lane >= off) end += v;
    }
    sh_end[wv][lane] = end;
    sh_cnt[wv][lane] = 0;
    if constexpr (EMIT) { sh_first[wv][lane] = first; sh_gid[wv][lane] = g; }
    const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)end, 63);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- phase 2
    for (uint32_t k0 = 0; k0 < T; k0 += kWave) {
        const uint32_t k = k0 + (uint32_t)lane;
        const bool valid = k < T;
        int j = 0;                                     // smallest j with end[j] > k
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1)
            if (sh_end[wv][j + step - 1] <= k) j += step;
        if (!valid) j = 63;
        const uint32_t ej = sh_end[wv][j];
        const float4 a = sh_a[wv][j], b = sh_b[wv][j];
        const uint32_t xy = __float_as_uint(b.z), w = __float_as_uint(b.w);
        // candidate index inside its Gaussian: its rectangle holds (ej - start) tiles, start = end[j-1]
        const uint32_t start = j > 0 ? sh_end[wv][j - 1] : 0u;
        const uint32_t i = k - start;
        bool want = false;
        uint32_t tile = 0;
        int tx = 0, ty = 0;
        if (valid) {
            int qy, rx;
            divmod_tiles(i, w, 1.f / (float)w, qy, rx);            // (w < 2^16: the reciprocal is one v_rcp away, cheaper than another LDS word)
            tx = (int)(xy & 0xFFFFu) + rx; ty = (int)(xy >> 16) + qy;
            tile = (uint32_t)(ty * f.Gx + tx);
            if ((bits[ty * W64 + (tx >> 6)] >> (tx & 63)) & 1ull) {
                float A, B, C, op;
                unscale_conic(a.z, a.w, b.x, b.y, A, B, C, op);
                want = tile_may_contribute(a.x, a.y, A, B, C, op, tx, ty);
            }
        }
        const unsigned long long m = __ballot(want);
        // this Gaussian's candidates sit on lanes [seg_lo, seg_hi) of this step
        const uint32_t seg_lo = i < (uint32_t)lane ? (uint32_t)lane - i : 0u;
        const uint32_t left = ej - k;                                      // candidates of j from this one on (>= 1)
        const uint32_t seg_hi = min(64u, (uint32_t)lane + left);
        const uint32_t before = sh_cnt[wv][j];
        if constexpr (EMIT) {
            if (want) {
                const unsigned long long below = m & ((1ull << lane) - 1ull) & (~0ull << seg_lo);
                const uint32_t slot = sh_first[wv][j] + before + (uint32_t)__popcll(below);
                keys[slot] = tile;
                vals[slot] = slot;
                inst_gid[slot] = sh_gid[wv][j] |
                                 (quadrant_mask_q(a.x, a.y, a.z, a.w, b.x, b.y, (float)(tx * GSR_TILE), (float)(ty * GSR_TILE)) << kQuadMaskShift);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (valid && (uint32_t)lane == seg_lo) {                            // one lane per Gaussian per step
            const unsigned long long seg = (seg_hi >= 64u ? ~0ull : ((1ull << seg_hi) - 1ull)) & (~0ull << seg_lo);
            sh_cnt[wv][j] = before + (uint32_t)__popcll(m & seg);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if constexpr (!EMIT) {
        if (mine) cnt_open[r] = sh_cnt[wv][lane];
    }
}

// ---- variant A': the tile lists of a chunk of few, large splats by GATHER, not by sort.  The count pass left the
// per-tile instance counts; k_tile_ranges turns them into the tiles' ranges (exclusive scan in tile order: exactly
// where the stable sort by tile id would put them) and k_tile_gather lets one wave per open tile walk the chunk's
// depth ranks, 64 x kGatherUnroll per step, test "is this tile inside the rank's rectangle, and did the count pass
// accept it" (one bit of the rank's mask words) and append the accepted ranks to the tile's list in rank order =
// depth order.  The result is bit for bit what emit + stable sort + ranges produce; the work is ranks x open tiles
// bit tests (44 M for cfg3's first chunk) instead of a 2 M-key radix sort.
// scan_n > 0 (chunks of up to kRankScanMax Gaussians): the same block first scans the chunk's per-rank counts (offs_open, the chunk's
// emitted total and the next chunk's base), which saves the launch of a one-block scan in front of this one.
constexpr int kRankScanMax = 16384;
__global__ __launch_bounds__(1024) void k_tile_ranges(FrameK f, int c, Ctrl *__restrict__ ctrl, uint32_t *__restrict__ tile_cnt,
                                                      uint2 *__restrict__ ranges_c, int scan_n, const uint32_t *__restrict__ cnt_open,
                                                      uint32_t *__restrict__ offs_open)
{
    __shared__ uint32_t sh_wave[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t_begin = f.ty0 * f.Gx, t_end = f.ty1 * f.Gx;
    if (scan_n > 0) {
        constexpr int kPer = kRankScanMax / 1024;                // 16 consecutive ranks per thread
        uint32_t v[kPer], mine = 0;
#pragma unroll
        for (int i = 0; i < kPer; ++i) { const int r = (int)threadIdx.x * kPer + i; v[i] = r < scan_n ? cnt_open[r] : 0u; mine += v[i]; }
        uint32_t inc = mine;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t u = __shfl_up(inc, off);
            if (lane >= off) inc += u;
        }
        if (lane == 63) sh_wave[wv] = inc;
        __syncthreads();
        uint32_t run = inc - mine, total = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const uint32_t u = sh_wave[i]; if (i < wv) run += u; total += u; }
#pragma unroll
        for (int i = 0; i < kPer; ++i) { const int r = (int)threadIdx.x * kPer + i; run += v[i]; if (r < scan_n) offs_open[r] = run; }
        if (threadIdx.x == 0) { ctrl->chunk_R[c] = total; ctrl->chunk_base[c + 1] = ctrl->chunk_base[c] + total; }
        __syncthreads();
    }
    uint32_t carry = ctrl->chunk_base[c];
    for (int t0 = t_begin; t0 < t_end; t0 += 1024) {
        const int t = t0 + (int)threadIdx.x;
        const uint32_t v = t < t_end ? tile_cnt[t] : 0u;
        if (v) tile_cnt[t] = 0u;                               // ready for the next chunk / frame
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t u = __shfl_up(inc, off);
            if (lane >= off) inc += u;
        }
        __syncthreads();
        if (lane == 63) sh_wave[wv] = inc;
        __syncthreads();
        uint32_t wbase = 0, wtot = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const uint32_t u = sh_wave[i]; if (i < wv) wbase += u; wtot += u; }
        if (v) { const uint32_t b = carry + wbase + inc - v; ranges_c[t] = make_uint2(b, b + v); }
        carry += wtot;
    }
}

constexpr int kGatherBlock = 512;            // 8 tiles per block share the ranks' metadata through LDS

// the heavy half of k_tile_gather: `cnt` (<= 64) queued (rank, tile-in-rectangle) candidates of one tile, one per lane
__device__ __forceinline__ uint32_t gather_flush(const uint2 *q, int cnt, int lane, unsigned long long below, uint32_t out,
                                                 uint32_t base, int r0, int tx, int ty, const uint4 *__restrict__ meta_a,
                                                 const float4 *__restrict__ meta_b, const unsigned long long *__restrict__ masks,
                                                 const uint32_t *__restrict__ wprefix, const uint32_t *__restrict__ cnt_open,
                                                 const uint32_t *__restrict__ offs_open, uint32_t *__restrict__ sorted_gid,
                                                 uint32_t *__restrict__ sorted_slot)
{
    bool acc = false;
    unsigned long long word = 0ull;
    uint32_t pre = 0, first = 0, g = 0, bit = 0;
    float4 mb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < cnt) {                                          // one round of loads, all independent
        const uint2 e = q[lane];
        const uint32_t r = e.y >> 6;
        bit = e.y & 63u;
        word = masks[e.x];
        pre = wprefix[e.x];
        g = meta_a[r].w;
        mb = meta_b[r];
        first = offs_open[r0 + r] - cnt_open[r0 + r];
        acc = (word >> bit) & 1ull;
    }
    const unsigned long long m = __ballot(acc);
    if (acc) {
        const uint32_t k = out + (uint32_t)__popcll(m & below);
        const uint32_t ordinal = pre + (uint32_t)__popcll(word & ((1ull << bit) - 1ull));
        sorted_gid[k] = g | (quadrant_mask_bbox(mb.x, mb.y, mb.z, mb.w, (float)(tx * GSR_TILE), (float)(ty * GSR_TILE)) << kQuadMaskShift);
        sorted_slot[k] = base + first + ordinal;
    }
    return out + (uint32_t)__popcll(m);
}

__global__ __launch_bounds__(kGatherBlock) void k_tile_gather(FrameK f, int c, int r0, int n, const Ctrl *__restrict__ ctrl,
                                                              const uint2 *__restrict__ ranges_c, const uint4 *__restrict__ meta_a,
                                                              const float4 *__restrict__ meta_b,
                                                              const unsigned long long *__restrict__ masks,
                                                              const uint32_t *__restrict__ wprefix, const uint32_t *__restrict__ cnt_open,
                                                              const uint32_t *__restrict__ offs_open, uint32_t *__restrict__ sorted_gid,
                                                              uint32_t *__restrict__ sorted_slot, uint32_t *__restrict__ row_begin)
{
    __shared__ uint4 sh_meta[kGatherBlock];
    __shared__ uint32_t sh_rank[kGatherBlock];
    __shared__ uint32_t sh_wcnt[kGatherBlock / kWave];
    __shared__ uint2 sh_q[kGatherBlock / kWave][2 * kWave];    // per tile: ranks whose rectangle holds it, waiting for a full wave
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t base = ctrl->chunk_base[c];
    const bool nothing = ctrl->chunk_R[c] == 0u;
    {                                                          // the ranks' first gradient rows
        const int r = (int)blockIdx.x * kGatherBlock + (int)threadIdx.x;
        if (r < n) {
            const uint32_t cnt = nothing ? 0u : cnt_open[r0 + r];
            row_begin[r0 + r] = base + (nothing ? 0u : offs_open[r0 + r] - cnt);
        }
    }
    if (nothing) return;
    constexpr int kTiles = kGatherBlock / kWave;
    const int slab_tiles = (f.ty1 - f.ty0) * f.Gx;
    const int t_first = (int)blockIdx.x * kTiles;
    if (t_first >= slab_tiles) return;                           // block-uniform
    const int t_local = t_first + wv;
    const int tile = f.ty0 * f.Gx + t_local;
    uint2 rng = make_uint2(0u, 0u);
    if (t_local < slab_tiles) rng = ranges_c[tile];
    if (__syncthreads_or(rng.y != rng.x) == 0) return;           // none of the block's tiles takes anything in this chunk
    const int ty = tile / f.Gx, tx = tile - ty * f.Gx;
    // bounding box of the block's tiles (consecutive in row-major order: one row, or the end of one and the start of the next):
    // a rank whose rectangle misses it is dropped for all eight tiles at once
    const int t_last = min(t_first + kTiles, slab_tiles) - 1;
    const int by0 = f.ty0 + t_first / f.Gx, by1 = f.ty0 + t_last / f.Gx;
    const int bx0 = by0 == by1 ? t_first % f.Gx : 0, bx1 = by0 == by1 ? t_last % f.Gx : f.Gx - 1;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint2 *q = sh_q[wv];
    uint32_t out = rng.x;
    int qn = 0;                                                // wave-uniform
    uint4 next = (int)threadIdx.x < n ? meta_a[threadIdx.x] : make_uint4(0u, 0u, 0u, 0u);
    for (int k0 = 0; k0 < n; k0 += kGatherBlock) {
        const uint4 mine = next;                                 // w = h = 0 (past the end): never inside
        {
            const int r = k0 + kGatherBlock + (int)threadIdx.x;
            next = r < n ? meta_a[r] : make_uint4(0u, 0u, 0u, 0u);
        }
        const int x0 = (int)(mine.x & 0xFFFFu), y0 = (int)(mine.x >> 16), w = (int)(mine.y & 0xFFFFu), h = (int)(mine.y >> 16);
        const bool keep = w > 0 && x0 <= bx1 && x0 + w > bx0 && y0 <= by1 && y0 + h > by0;
        const unsigned long long mk = __ballot(keep);
        __syncthreads();                                         // (the previous batch's survivors have been read)
        if (lane == 0) sh_wcnt[wv] = (uint32_t)__popcll(mk);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < kTiles; ++i) { const uint32_t v = sh_wcnt[i]; if (i < wv) before += v; total += v; }
        if (keep) {
            const uint32_t p = before + (uint32_t)__popcll(mk & below);
            sh_meta[p] = mine; sh_rank[p] = (uint32_t)(k0 + (int)threadIdx.x);
        }
        __syncthreads();
        if (out >= rng.y) continue;                            // this tile's list is complete (or empty); keep the barriers
        for (uint32_t u = 0; u < total; u += kWave) {
            const bool have = u + (uint32_t)lane < total;
            const uint4 ma = have ? sh_meta[u + lane] : make_uint4(0u, 0u, 0u, 0u);
            const uint32_t dx = (uint32_t)tx - (ma.x & 0xFFFFu), dy = (uint32_t)ty - (ma.x >> 16), ww = ma.y & 0xFFFFu;
            const bool in = dx < ww && dy < (ma.y >> 16);
            const unsigned long long m_in = __ballot(in);
            if (m_in == 0ull) continue;
            if (in) {
                const uint32_t idx = dy * ww + dx;
                q[qn + __popcll(m_in & below)] = make_uint2(ma.z + (idx >> 6), (sh_rank[u + lane] << 6) | (idx & 63u));
            }
            qn += __popcll(m_in);
            if (qn >= kWave) {
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                out = gather_flush(q, kWave, lane, below, out, base, r0, tx, ty, meta_a, meta_b, masks, wprefix, cnt_open, offs_open,
                                   sorted_gid, sorted_slot);
                const uint2 e = q[kWave + lane];
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                q[lane] = e;
                qn -= kWave;
            }
        }
    }
    if (qn > 0 && out < rng.y) {
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        gather_flush(q, qn, lane, below, out, base, r0, tx, ty, meta_a, meta_b, masks, wprefix, cnt_open, offs_open, sorted_gid, sorted_slot);
    }
}

// ---- late chunks: most tiles have closed, so most of the chunk's Gaussians can no longer emit anything, yet sorting,
// scanning and counting would still walk all of them (the last chunk of a frame with an uncovered corner is most of the
// scene).  Before such a chunk is sorted, a stable two-way partition of its range puts the Gaussians whose rectangle still
// holds an open tile in front; ctrl->chunk_live[c] = how many.  The sort and the binning (one wave per Gaussian: the
// survivors all have work, and there are few of them) touch the front part only; the others stay in the depth order
// behind them, unsorted, with an instance count of zero.
constexpr int kLiveThreads = 512;
constexpr int kLiveWaves = kLiveThreads / kWave;

__device__ __forceinline__ void live_block_range(int n, int &lo, int &hi)
{
    int per = (n + (int)gridDim.x - 1) / (int)gridDim.x;
    per = (per + kLiveThreads - 1) / kLiveThreads * kLiveThreads;
    const long long l = (long long)blockIdx.x * per;
    lo = l < n ? (int)l : n;
    hi = l + per < n ? (int)(l + per) : n;
}

// pass 1: one flag byte per position + the live count of every block
__global__ __launch_bounds__(kLiveThreads) void k_live_flags(FrameK f, int r0, int n, const uint32_t *__restrict__ order,
                                                             const float4 *__restrict__ records,
                                                             const unsigned long long *__restrict__ open_bits, SelState *st,
                                                             uint8_t *__restrict__ flags)
{
    __shared__ unsigned long long sh_bits[kBitsMaxWords];
    __shared__ uint32_t sh_cnt;
    const int W64 = (f.Gx + 63) >> 6, n_words = f.Gy * W64;
    const bool in_lds = n_words <= kBitsMaxWords;
    if (in_lds)
        for (int j = threadIdx.x; j < n_words; j += kLiveThreads) sh_bits[j] = open_bits[j];
    if (threadIdx.x == 0) sh_cnt = 0;
    __syncthreads();
    const unsigned long long *bits = in_lds ? sh_bits : open_bits;
    int lo, hi;
    live_block_range(n, lo, hi);
    uint32_t mine = 0;
    for (int i = lo + (int)threadIdx.x; i < hi; i += kLiveThreads) {
        const float4 cc = records[3 * (size_t)order[r0 + i] + 2];
        const bool live = rect_has_open_tile(unpack_rect(cc.z, cc.w), bits, W64);
        flags[i] = live ? 1 : 0;
        mine += live ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine += (uint32_t)__shfl_xor((int)mine, off);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&sh_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) st->blk_cnt[0][blockIdx.x] = sh_cnt;
}

// pass 2: live Gaussians to the front, the others behind them, both in their old order; (Gaussian, relative key, tiles) move together
__global__ __launch_bounds__(kLiveThreads) void k_live_scatter(int c, int r0, int n, const uint32_t *__restrict__ order,
                                                               const uint32_t *__restrict__ pos_key, const uint32_t *__restrict__ pos_tiles,
                                                               const uint8_t *__restrict__ flags, const SelState *__restrict__ st, Ctrl *ctrl,
                                                               uint32_t *__restrict__ t_order, uint32_t *__restrict__ t_key,
                                                               uint32_t *__restrict__ t_tiles, LiveParts parts)
{
    __shared__ uint32_t sh_w[2][kLiveWaves], sh_pre[2][kLiveWaves], sh_run[2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv < 2) {                                                // wave 0: live Gaussians in the blocks before this one; wave 1: all live
        uint32_t s = 0;
        const int upto = wv == 0 ? (int)blockIdx.x : (int)gridDim.x;
        for (int b = lane; b < upto; b += kWave) s += st->blk_cnt[0][b];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += (uint32_t)__shfl_xor((int)s, off);
        if (lane == 0) sh_run[wv] = s;
    }
    __syncthreads();
    int lo, hi;
    live_block_range(n, lo, hi);
    const uint32_t live_total = sh_run[1];
    uint32_t live_before = sh_run[0], dead_before = (uint32_t)lo - live_before;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctrl->chunk_live[c] = live_total;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int i0 = lo; i0 < hi; i0 += kLiveThreads) {
        const int i = i0 + (int)threadIdx.x;
        const bool valid = i < hi;
        const bool live = valid && flags[i] != 0;
        const unsigned long long m = __ballot(live), mv = __ballot(valid);
        __syncthreads();                                         // (sh_pre / sh_run of the previous round have been read)
        if (lane == 0) { sh_w[0][wv] = (uint32_t)__popcll(m); sh_w[1][wv] = (uint32_t)__popcll(mv & ~m); }
        __syncthreads();
        if (threadIdx.x < 2) {
            uint32_t run = threadIdx.x == 0 ? live_before : live_total + dead_before;
#pragma unroll
            for (int w = 0; w < kLiveWaves; ++w) { sh_pre[threadIdx.x][w] = run; run += sh_w[threadIdx.x][w]; }
            sh_run[threadIdx.x] = run;
        }
        __syncthreads();
        if (valid) {
            const uint32_t pos = live ? sh_pre[0][wv] + (uint32_t)__popcll(m & below) : sh_pre[1][wv] + (uint32_t)__popcll(mv & ~m & below);
            uint32_t delta = 0;                                  // (a range merged from several planned chunks: one key base for all)
            for (int j = 0; j + 1 < parts.n; ++j) delta = (uint32_t)i >= parts.end[j] ? parts.delta[j + 1] : delta;
            t_order[pos] = order[r0 + i]; t_key[pos] = pos_key[r0 + i] + delta; t_tiles[pos] = pos_tiles[r0 + i];
        }
        live_before = sh_run[0]; dead_before = sh_run[1] - live_total;
    }
}

__global__ __launch_bounds__(256) void k_live_copyback(int n, const uint32_t *__restrict__ t_order, const uint32_t *__restrict__ t_key,
                                                       const uint32_t *__restrict__ t_tiles, uint32_t *__restrict__ order,
                                                       uint32_t *__restrict__ pos_key, uint32_t *__restrict__ pos_tiles)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    order[i] = t_order[i]; pos_key[i] = t_key[i]; pos_tiles[i] = t_tiles[i];
}

int launch_live_filter(const FrameK &f, int c, int r0, int r1, const LiveParts &parts, GeomWS &gw, ImageWS &iw, bool debug, hipStream_t s)
{
    const int n = r1 - r0;
    if (n <= 0) return GSR_OK;
    ProfileScope prof("live_filter", s);
    int blocks = (n + kLiveThreads * 4 - 1) / (kLiveThreads * 4);
    if (blocks > kSelBlocks) blocks = kSelBlocks;
    // scratch: the chunk's own by-rank arrays, unused until its binning starts; the flags in the radix table area
    uint32_t *t_order = gw.offs_open + r0, *t_key = gw.row_begin + r0, *t_tiles = gw.cnt_open + r0;
    uint8_t *flags = reinterpret_cast<uint8_t *>(gw.offs_full + r0);          // n bytes of the chunk's (not yet written) tile-count scan
    hipLaunchKernelGGL(k_live_flags, dim3(blocks), dim3(kLiveThreads), 0, s, f, r0, n, gw.order, gw.records, iw.open_bits, gw.sel, flags);
    hipLaunchKernelGGL(k_live_scatter, dim3(blocks), dim3(kLiveThreads), 0, s, c, r0, n, gw.order, gw.sort_keys[1], gw.sort_vals[1], flags,
                       gw.sel, gw.ctrl, t_order, t_key, t_tiles, parts);
    hipLaunchKernelGGL(k_live_copyback, dim3((n + 255) / 256), dim3(256), 0, s, n, t_order, t_key, t_tiles, gw.order + r0, gw.sort_keys[1] + r0,
                       gw.sort_vals[1] + r0);
    GSR_LAUNCH_CHECK("live_filter", debug, s);
    return GSR_OK;
}

// ---- ranges of the chunk's sorted list (grid-stride, element count on the device)
// GATHER: also sorted_gid[a] = inst_gid[slot at a] (sorts below 4 M instances, whose slot -> Gaussian table stays in the L2 / Infinity
// Cache; above, the word travels through the radix passes as a second payload: the gather read 128 B per 4-byte word, 3.1 GB at 23 M)
template <bool GATHER>
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
        if constexpr (GATHER) sorted_gid[a] = inst_gid[slots[a]];
    }
}

static int msb_plus1(uint32_t n)
{
    int b = 0;
    while (n) { ++b; n >>= 1; }
    return b;
}

int launch_chunk_binning(const FrameK &f, int c, int r0, int r1, uint64_t n_max, uint64_t emitted_before, GeomWS &gw, BinningWS &bw,
                         ImageWS &iw,
                         int *sort_result, bool debug, hipStream_t s, bool filtered)
{
    int rc;
    const int n = r1 - r0;
    if (n <= 0) return GSR_OK;
    const size_t Tn = (size_t)f.Gx * f.Gy;
    // variant B packs the (Gaussian, tile) candidates of 64 Gaussians into full wave steps: right when rectangles are
    // small; chunks of few large splats (the nearest ones) get a team of 1, 4 or 16 waves per Gaussian (variant A)
    const uint64_t avg = n_max / (uint64_t)n;
    // a chunk the live filter went through: few survivors, every one of them with work -> one wave per Gaussian whatever the
    // rectangle size (64 ranks per wave would leave a handful of long-running waves); its mask scratch fits (the caller checked)
    const bool flat = avg < 24 && !filtered;
    const int team = filtered ? 1 : avg >= 1024 ? 16 : avg >= 96 ? 4 : 1;
    const int bin_blocks = (n + kBinBlock - 1) / kBinBlock;      // B: 64 depth ranks per wave, 4 waves per block
    // A's mask scratch: the idle half of the radix double buffer beyond everything earlier chunks have written (their exact
    // emitted count; ceil(n_max / 64) + n words of 8 bytes fit in the n_max slots the caller has checked are there: a team
    // chunk averages >= 24 tiles per Gaussian)
    unsigned long long *masks = reinterpret_cast<unsigned long long *>(bw.keys[1] + ((emitted_before + 1) & ~(uint64_t)1));
    // A' (gather instead of emit + sort) when the ranks x tiles bit tests are cheaper than sorting the instances; its
    // scratch (mask prefixes, rank metadata: n_max / 64 + 9 n + 4 words) sits in the other idle key buffer
    const uint64_t slab_tiles = (uint64_t)(f.ty1 - f.ty0) * (uint64_t)f.Gx;
    const uint64_t scratch_words = n_max / 64 + 1 + 9 * (uint64_t)n + 4;
    static const bool no_gather = getenv("GSR_NO_GATHER") != nullptr;      // experiment switch, read once
    const bool gather = !flat && !filtered && !no_gather && n < (1 << 26) && (uint64_t)n * slab_tiles <= 24 * n_max + (1ull << 22) &&
                        scratch_words <= n_max;
    uint32_t *scratch = bw.keys[0] + ((emitted_before + 3) & ~(uint64_t)3);
    uint4 *meta_a = reinterpret_cast<uint4 *>(scratch);
    float4 *meta_b = reinterpret_cast<float4 *>(scratch + 4 * (size_t)n);
    uint32_t *wprefix = scratch + 8 * (size_t)n;
    const int tile_bits = msb_plus1((uint32_t)(Tn ? Tn - 1 : 0));
    // The blend kernels find a sorted entry's Gaussian (| quadrant mask) in gids[1], always.  Large sorts carry the word through the
    // radix passes (the emit kernels then write it into the buffer the passes leave in [1]); small ones gather it behind the sort.
    const int sort_passes = ((tile_bits > 0 ? tile_bits : 1) + 7) / 8;
    const bool carry_gid = n_max >= (4ull << 20);
    uint32_t *const gid_emit = bw.gids[carry_gid && !(sort_passes & 1) ? 1 : 0];
    uint32_t *const gid_pingpong[2] = {gid_emit, gid_emit == bw.gids[0] ? bw.gids[1] : bw.gids[0]};
    // a filtered chunk: a fixed grid walks the live front part; everybody else's count is zero from the start
    const int team_grid = filtered ? (n < 65536 ? n : 65536) : n;      // (4 k: +35 us per launch, 256 k: +20 us; measured)
    if (filtered) GSR_HIP_CHECK(hipMemsetAsync(gw.cnt_open + r0, 0, (size_t)n * sizeof(uint32_t), s));
    {
        ProfileScope prof("count_open", s);
#define GSR_CT(W, G)                                                                                                       \
    hipLaunchKernelGGL((k_count_team<W, G>), dim3(team_grid), dim3(W * kWave), 0, s, f, c, r0, gw.order, gw.records, iw.open_bits, gw.ctrl, \
                       gw.offs_full, masks, gw.cnt_open, iw.tile_cnt, wprefix, meta_a, meta_b, n)
        if (flat)
            hipLaunchKernelGGL(k_bin_chunk<false>, dim3(bin_blocks), dim3(kBinBlock), 0, s, f, c, r0, r1, gw.order, gw.records,
                               iw.open_bits, gw.ctrl, gw.cnt_open, gw.offs_open, bw.keys[0], bw.vals[0], gid_emit, gw.row_begin);
        else if (team == 16) { if (gather) GSR_CT(16, true); else GSR_CT(16, false); }
        else if (team == 4) { if (gather) GSR_CT(4, true); else GSR_CT(4, false); }
        else { if (gather) GSR_CT(1, true); else GSR_CT(1, false); }
#undef GSR_CT
        GSR_LAUNCH_CHECK("count_open", debug, s);
    }
    const bool fused_scan = gather && n <= kRankScanMax;         // the rank scan rides in k_tile_ranges
    if (!fused_scan && (rc = launch_scan_inclusive(gw.cnt_open + r0, gw.offs_open + r0, n, gw.scan_temp, &gw.ctrl->chunk_R[c],
                                                   &gw.ctrl->chunk_base[c], &gw.ctrl->chunk_base[c + 1], "scan_open", debug, s)))
        return rc;
    if (gather) {
        // the list ends where the radix passes would have left it, so that all chunks of a frame agree on the buffer
        const int res = (((tile_bits > 0 ? tile_bits : 1) + 7) / 8) & 1;
        *sort_result = res;
        {
            ProfileScope prof("tile_ranges", s);
            hipLaunchKernelGGL(k_tile_ranges, dim3(1), dim3(1024), 0, s, f, c, gw.ctrl, iw.tile_cnt, iw.ranges + (size_t)c * Tn,
                               fused_scan ? n : 0, gw.cnt_open + r0, gw.offs_open + r0);
            GSR_LAUNCH_CHECK("tile_ranges", debug, s);
        }
        ProfileScope prof("tile_gather", s);
        const uint64_t tile_blocks = (slab_tiles + kGatherBlock / kWave - 1) / (kGatherBlock / kWave);
        const uint64_t rank_blocks = ((uint64_t)n + kGatherBlock - 1) / kGatherBlock;
        const uint64_t blocks = tile_blocks > rank_blocks ? tile_blocks : rank_blocks;
        hipLaunchKernelGGL(k_tile_gather, dim3((unsigned)blocks), dim3(kGatherBlock), 0, s, f, c, r0, n, gw.ctrl, iw.ranges + (size_t)c * Tn,
                           meta_a, meta_b, masks, wprefix, gw.cnt_open, gw.offs_open, bw.gids[1], bw.vals[res], gw.row_begin);
        GSR_LAUNCH_CHECK("tile_gather", debug, s);
        return GSR_OK;
    }
    {
        ProfileScope prof("emit", s);
#define GSR_ET(W)                                                                                                          \
    hipLaunchKernelGGL(k_emit_team<W>, dim3(team_grid), dim3(W * kWave), 0, s, f, c, r0, gw.order, gw.records, gw.ctrl, gw.offs_full,  \
                       masks, gw.cnt_open, gw.offs_open, bw.keys[0], bw.vals[0], gid_emit, gw.row_begin, n)
        if (flat)
            hipLaunchKernelGGL(k_bin_chunk<true>, dim3(bin_blocks), dim3(kBinBlock), 0, s, f, c, r0, r1, gw.order, gw.records,
                               iw.open_bits, gw.ctrl, gw.cnt_open, gw.offs_open, bw.keys[0], bw.vals[0], gid_emit, gw.row_begin);
        else if (team == 16) GSR_ET(16);
        else if (team == 4) GSR_ET(4);
        else GSR_ET(1);
#undef GSR_ET
        GSR_LAUNCH_CHECK("emit", debug, s);
    }
    if ((rc = launch_radix_sort<uint32_t>(bw.keys, bw.vals, &gw.ctrl->chunk_R[c], 0, n_max, &gw.ctrl->chunk_base[c], 0,
                                          tile_bits > 0 ? tile_bits : 1, gw.radix_temp, sort_result, "tile_sort", debug, s, false,
                                          carry_gid ? gid_pingpong : nullptr)))
        return rc;
    {
        ProfileScope prof("ranges", s);
        uint64_t blocks = (n_max + kBinBlock - 1) / kBinBlock;
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        if (carry_gid)
            hipLaunchKernelGGL(k_ranges<false>, dim3((unsigned)blocks), dim3(kBinBlock), 0, s, c, gw.ctrl, bw.keys[*sort_result],
                               nullptr, nullptr, iw.ranges + (size_t)c * Tn, nullptr);
        else
            hipLaunchKernelGGL(k_ranges<true>, dim3((unsigned)blocks), dim3(kBinBlock), 0, s, c, gw.ctrl, bw.keys[*sort_result],
                               bw.vals[*sort_result], bw.gids[0], iw.ranges + (size_t)c * Tn, bw.gids[1]);
        GSR_LAUNCH_CHECK("ranges", debug, s);
    }
    return GSR_OK;
}

}  // namespace gsr
