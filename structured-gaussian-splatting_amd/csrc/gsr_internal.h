// gsr_internal.h — workspace layouts, launch helpers and kernel-launcher prototypes shared by the
// translation units of libgsrast.so.  Not part of the public ABI (that is include/gsrast.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gsrast.h"
#include "gsr_math.h"

namespace gsr {

constexpr int kWave = 64;                 // CDNA wavefront
constexpr int kRowFloats = 12;            // per-instance gradient row (48 B, 3 x float4; 9 used)
constexpr size_t kAlign = 256;

void set_error(const char *fmt, ...);     // thread-local message behind gsr_last_error()

#define GSR_HIP_CHECK(expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            gsr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GSR_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

// After every kernel launch: catches launch-configuration errors always, and (debug=1, the
// reference's `debug` flag: README.md:147-150) synchronises so a faulting kernel is named.
#define GSR_LAUNCH_CHECK(name, debug, stream)                                                     \
    do {                                                                                          \
        hipError_t _e = hipGetLastError();                                                        \
        if (_e == hipSuccess && (debug)) _e = hipStreamSynchronize(stream);                       \
        if (_e != hipSuccess) {                                                                   \
            gsr::set_error("kernel %s failed: %s", name, hipGetErrorString(_e));                  \
            return GSR_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

// ---- optional per-kernel timing (gsr_profile_enable / gsr_profile_read)
bool profile_on();
int profile_begin(const char *name, hipStream_t s);
void profile_end(int idx, hipStream_t s);
struct ProfileScope {
    hipStream_t s; int idx;
    ProfileScope(const char *name, hipStream_t st) : s(st), idx(profile_on() ? profile_begin(name, st) : -1) {}
    ~ProfileScope() { if (idx >= 0) profile_end(idx, s); }
};

inline size_t align_up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

inline FrameK make_frame(const gsr_frame_desc &d)
{
    FrameK f;
    f.P = d.P; f.D = d.sh_degree; f.M = d.sh_coeffs; f.W = d.width; f.H = d.height;
    f.Gx = (d.width + GSR_TILE - 1) / GSR_TILE;
    f.Gy = (d.height + GSR_TILE - 1) / GSR_TILE;
    f.ty0 = d.tile_row_begin < 0 ? 0 : (d.tile_row_begin > f.Gy ? f.Gy : d.tile_row_begin);
    f.ty1 = (d.tile_row_end <= 0 || d.tile_row_end > f.Gy) ? f.Gy : d.tile_row_end;
    if (f.ty1 < f.ty0) f.ty1 = f.ty0;
    f.tanfovx = d.tanfovx; f.tanfovy = d.tanfovy;
    f.focal_x = (float)d.width / (2.f * d.tanfovx);
    f.focal_y = (float)d.height / (2.f * d.tanfovy);
    f.scale_modifier = d.scale_modifier;
    return f;
}

// ---- workspace carving (all sub-arrays 256-B aligned; the caller's base is torch-allocated, 512-B aligned)
// ---- depth selection (gsr_select.hip): two-level histogram of the depth keys, chunk rule, stable partition by chunk
constexpr int kSelBins = 2048;           // bins per level
constexpr int kSelShift1 = 20;           // level 1: key >> 20 (positive float bits < 2^31: 2040 bins in use, 8 per binade)
constexpr int kSelShift2 = 9;            // level 2: the next 11 bits -> sub-bins of 512 key codes (6e-5 relative depth)
constexpr int kSelRefine = 2;            // chunk boundaries refined at level 2: the ends of chunks 0 and 1
constexpr int kSelBlocks = 1024;         // blocks of the partition kernels (the histogram kernels run 256: fewer flushes)
// the chunk rule (see gsr_binning.hip's header): chunk c ends where the running optical mass passes
// kChunkOpticalDepths x ln(1e4) per slab pixel x 4^c and the running tile count passes kMinFirstChunk x 4^c
constexpr float kCutoffOpticalDepth = 9.2103404f;   // -ln(GSR_T_CUTOFF)
constexpr float kChunkOpticalDepths = 5.f;
constexpr uint32_t kMinFirstChunk = 1u << 18;
constexpr int kChunkGrowthLog2 = 2;              // x4 per chunk
struct SelTables { uint32_t cnt[kSelBins]; unsigned long long tiles[kSelBins], mass[kSelBins]; };
struct SelState {               // device scratch of the selection (zeroed at the start of every frame)
    SelTables t1, t2[kSelRefine];
    uint32_t bin[kSelRefine], base_cnt[kSelRefine];
    unsigned long long base_tiles[kSelRefine], base_mass[kSelRefine];
    uint32_t coarse_bin[GSR_MAX_CHUNKS], coarse_cnt[GSR_MAX_CHUNKS];
    unsigned long long coarse_tiles[GSR_MAX_CHUNKS];
    uint32_t V, max_bin;         // max_bin: the highest occupied level-1 bin
    unsigned long long R;
    uint32_t done[2];           // tickets of the two histogram kernels: their last block runs the plan step
    uint32_t blk_cnt[GSR_MAX_CHUNKS][kSelBlocks];
    uint32_t pad16_[2];
};
static_assert(sizeof(SelState) % 16 == 0, "k_preprocess clears SelState with 16-byte stores");

struct Ctrl {                   // small device-side control block of one frame
    uint32_t R_total, V, num_chunks, open_count;
    uint32_t bnd[GSR_MAX_CHUNKS + 1];          // boundaries of the chunks in the depth order
    uint32_t key_end[GSR_MAX_CHUNKS];          // chunk c holds the visible Gaussians with depth bits in (key_end[c-1], key_end[c]]
    uint32_t chunk_full[GSR_MAX_CHUNKS];       // instances of the chunk if every tile were open
    uint32_t chunk_R[GSR_MAX_CHUNKS];          // instances actually emitted
    uint32_t chunk_base[GSR_MAX_CHUNKS + 1];   // first absolute instance index of the chunk
    uint32_t chunk_live[GSR_MAX_CHUNKS];       // Gaussians at the front of the chunk's range that can still reach an open tile
                                               // (0xFFFFFFFF: the chunk was not filtered, all of them)
    uint32_t open_stuck;                       // open tiles with a pixel that is still more than half transparent (nothing covers it yet)
    uint32_t overflow;                         // 1 = the sum of tiles touched does not fit 32 bits
    uint32_t prefilter_violation;              // 1 = prefiltered was set and a Gaussian failed the frustum test (A.1)
    uint32_t key_max;                          // no visible Gaussian's depth key exceeds it (upper edge of the highest occupied histogram bin)
};
struct GeomWS {                 // O(P): the reference's geomBuffer
    float4 *records;            // [P,3]  Splat records, by Gaussian
    uint2 *tiles_mass;          // [P]    by Gaussian: .x = tiles touched, .y = optical mass inside the slab in 1/64 pixel-neper
                                //        units (gsr_math.h optical_mass): one 8-byte gather in the depth-order scan
    unsigned long long *mass_blocks;   // [ceil(P / 2048) + 1] scratch of the tile-count scan (per-block sums of the pairs' second half)
    SelState *sel;              // depth selection scratch
    uint8_t *clamped;           // [P]    by Gaussian
    uint32_t *sort_keys[2];     // [P] x2 [0]: depth bits by Gaussian (0xFFFFFFFF = invisible), kept; [1]: chunk-sort scratch
    uint32_t *sort_vals[2];     // [P] x2 [0] = order; [1]: chunk-sort scratch
    uint32_t *order;            // [P]    the depth order: chunk c at [bnd[c], bnd[c+1]), in (depth, index) order once the
                                //        chunk has been binned, in index order before; beyond V: unspecified
    uint32_t *offs_full;        // [P]    by position: inclusive scan of the tiles touched, restarting at every chunk
    uint32_t *cnt_open;         // [P]    by rank: instances emitted for this Gaussian
    uint32_t *offs_open;        // [P]    by rank: inclusive scan of cnt_open inside its chunk
    uint32_t *row_begin;        // [P]    by rank: absolute index of the Gaussian's first instance
    Ctrl *ctrl;
    void *scan_temp;
    void *radix_temp;
    size_t total;
};
// ---- work units of the blend backward (K7).  A tile's list is walked FRONT TO BACK in segments of kSeg entries; the forward
// leaves the per-pixel state (T, colour) at every segment boundary (a checkpoint), so the segments of one tile are independent
// work units: (tile, chunk, segment).  kSeg list entries are ~25 us of one wave; a 400-entry tile as ONE unit was 100 us and the
// launch ended with a third of its time draining (two rounds of 4096 resident waves, tail = the last-started tiles).
// The forward's wave of a (tile, chunk) appends the pair's units itself when it is done — one returning atomic on one of
// kUnitShards counters (shard = tile % 8), so that the list costs no kernel and no block of its own — and the backward reads
// every shard's list BACKWARDS: the tiles that finish the forward last are the long ones, and the launch should start with them.
#ifndef GSR_BWD_SEG
#define GSR_BWD_SEG 128
#endif
constexpr int kSeg = GSR_BWD_SEG;
static_assert(kSeg % kWave == 0, "segments are whole 64-entry batches");
constexpr int kCkptFloats = 16 * kWave;   // one checkpoint = (T, r, g, b) of the tile's 256 pixels: [quadrant][field][lane], 4 KB
constexpr int kUnitTileBits = 24;         // BwdUnit.x = tile | chunk << 24; .y = segment
constexpr int kUnitShards = 8;
// ... in kUnitClasses lists per shard, longest first: full segments, then the pairs' last, partial segments in 16 length classes of
// kSeg / 16 entries.  The launch ends when the last-started units finish: started in arbitrary order (any unit last) K7 took 618 us
// at cfg3n, longest first 564; cfg3's 8 160 units are nearly all partial (two rounds of 4 096 waves): 4 classes cost it 17 us.
constexpr int kUnitPartClasses = 16;
constexpr int kUnitClasses = 1 + kUnitPartClasses;
static_assert(kSeg % kUnitPartClasses == 0, "length classes of whole entries");
__host__ __device__ inline int unit_class(uint32_t rest) { return kUnitPartClasses - (int)(rest / (kSeg / kUnitPartClasses)); }      // rest in 1 .. kSeg - 1
struct UnitLists {             // where a shard's five lists live inside BinningWS::units
    uint2 *units; uint32_t cap_full, cap_part;
    __host__ __device__ size_t shard_stride() const { return (size_t)cap_full + (size_t)(kUnitClasses - 1) * cap_part; }
    __host__ __device__ size_t list_begin(int shard, int cls) const
    {
        return (size_t)shard * shard_stride() + (cls == 0 ? 0 : (size_t)cap_full + (size_t)(cls - 1) * cap_part);
    }
    __host__ __device__ uint32_t list_cap(int cls) const { return cls == 0 ? cap_full : cap_part; }
};

struct ImageWS {                // O(N + Tn): the reference's imgBuffer
    float *T_state;             // [N]  running / final transmittance; negative = pixel hit the cut-off
    int32_t *last_enc;          // [N]  (chunk + 1) << 26 | contributor position in that chunk's range
    uint2 *ranges;              // [GSR_MAX_CHUNKS][Tn]
    uint32_t *tile_cnt;         // [Tn] instances per tile of the chunk being binned (gather variant; zero between chunks)
    uint32_t *unit_count;       // [kUnitShards][kUnitClasses] work units of the blend backward appended so far (cleared with the ranges and counters)
    uint32_t *open;             // [Tn] 1 = tile still has an unsaturated pixel (0 outside the slab)
    unsigned long long *open_bits;   // [Gy][ceil(Gx/64)] the same flags, one bit per tile (rebuilt at chunk boundaries)
    uint32_t *tile_walk;        // [GSR_MAX_CHUNKS][Tn] entries of chunk c's range the backward walks: the tile's deepest contributor
                                //      there (written by K6 for every tile that blends in chunk c)
    float *ckpt_start;          // [GSR_MAX_CHUNKS - 1][Tn][kCkptFloats] the pixels' state when chunk c >= 1 starts on a tile
    Ctrl *ctrl_scratch;         // stand-in control block for frames without a geometry workspace (P == 0)
    size_t total;
};
struct BinningWS {              // O(R): the reference's binningBuffer
    uint32_t *keys[2];          // [R] x2 tile id (radix double buffer)
    uint32_t *vals[2];          // [R] x2 payload = instance slot (absolute index in emission order)
    uint32_t *gids[2];          // [R] x2 Gaussian | quadrant mask << 28 (gsr_math.h quadrant_mask_q of the instance's tile): [1] by sorted
                                //      position — what the blend kernels gather records by; the other side holds the word by slot as emitted
                                //      (sorts of >= 4 M instances carry it through the radix passes as a second payload; smaller ones
                                //      gather [1][p] = [0][slot at p] behind the sort)
    uint8_t *row_valid;         // [R] by slot: 1 = the blend backward wrote the instance's gradient row (cleared ahead of it: by the
                                //      forward's zero fill, or by gsr_backward_render itself)
    float *ckpt;                // [R / kSeg + 2][kCkptFloats] the pixels' state in front of sorted position p = range start + k kSeg
                                //      (k >= 1), at slot p / kSeg: ranges are disjoint, so the slots are
    UnitLists units;            // [kUnitShards] x (full segments [R / kSeg + 1], 16 classes of partial ones [GSR_MAX_CHUNKS (Tn / 8 + 1)] each):
                                // work units of the blend backward, each list in the order the forward's waves finished
    float *grad_rows;           // [instances emitted, kRowFloats] per-instance screen-space gradient rows, by slot: the caller's
                                // backward-time allocation (gsr_backward_rows_size), not part of the carved block
    size_t total;
};
constexpr int kLastShift = 26;  // last_enc = (chunk + 1) << kLastShift | position
constexpr int kQuadMaskShift = 28;                // BinningWS::gids = quadrant mask << 28 | Gaussian (P < 2^28)
constexpr uint32_t kGidMask = (1u << kQuadMaskShift) - 1u;
constexpr float kMassUnitsPerPixelNeper = 64.f;   // GeomWS::mass fixed point
constexpr int kScanTileElems = 2048;              // elements per block of the prefix sums (= granularity of mass_blocks)

size_t scan_temp_bytes(int n);
size_t radix_temp_bytes();
GeomWS carve_geom(void *base, int P);
ImageWS carve_image(void *base, const FrameK &f);
BinningWS carve_binning(void *base, int64_t R, const FrameK &f);

// ---- primitives (gsr_sort.hip)
int launch_scan_inclusive(const uint32_t *in, uint32_t *out, int n, void *temp, uint32_t *grand_total, const uint32_t *acc_in,
                          uint32_t *acc_out, const char *name, bool debug, hipStream_t s, uint32_t *overflow = nullptr,
                          const uint32_t *gather = nullptr, const uint2 *aux = nullptr,
                          unsigned long long *aux_block_prefix = nullptr);
// vals2: an optional second payload, carried along with vals
template <typename K>
int launch_radix_sort(K *const keys[2], uint32_t *const vals[2], const uint32_t *n_ptr, uint32_t n_host, uint64_t n_max,
                      const uint32_t *base_ptr, int begin_bit, int end_bit, void *temp, int *result, const char *name,
                      bool debug, hipStream_t s, bool even_passes = false, uint32_t *const *vals2 = nullptr);

// ---- kernel launchers (each returns a gsr_status)
int launch_preprocess(const FrameK &f, const gsr_camera &cam, const gsr_gaussians &g, GeomWS &ws, int32_t *radii,
                      bool prefiltered, bool debug, hipStream_t s);
// clear16 / clear16_n: 16-byte words the partition's scatter kernel zeroes on the side (the frame's tile ranges and counters)
int launch_depth_select(const FrameK &f, GeomWS &ws, bool debug, hipStream_t s, void *clear16 = nullptr, size_t clear16_n = 0);
int launch_rows_gather(const FrameK &f, GeomWS &ws, uint32_t key_max, const float *screen, int n_rows, int32_t *rows, float *packed,
                       bool debug, hipStream_t s);
int launch_rows_scatter(int n_rows, int P, const int32_t *rows, const float *packed, float *screen, bool debug, hipStream_t s);
int launch_chunk_order(const FrameK &f, int r0, int r1, uint32_t key_lo, uint32_t key_hi, bool first, GeomWS &ws, bool debug, hipStream_t s,
                       const uint32_t *live_count = nullptr);
// parts: the range may span several planned chunks (merged by the caller); their relative keys are re-based to the first one's
struct LiveParts { int n; uint32_t end[GSR_MAX_CHUNKS]; uint32_t delta[GSR_MAX_CHUNKS]; };      // end: position in the range; delta: added to the key
int launch_live_filter(const FrameK &f, int c, int r0, int r1, const LiveParts &parts, GeomWS &gw, ImageWS &iw, bool debug, hipStream_t s);
size_t binning_clear_bytes(const FrameK &f, const ImageWS &iw);      // ranges + tile counters: a multiple of 16
// Where k_open_count publishes the control block for the host: sizeof(Ctrl) / 4 words of host-coherent pinned memory (device
// address) followed by a flag word that receives `seq` last, with system-scope release.  words == nullptr: not wanted.
struct CtrlMirror { uint32_t *words = nullptr; uint32_t seq = 0; };
int launch_binning_init(const FrameK &f, GeomWS &gw, ImageWS &iw, bool debug, hipStream_t s, bool ranges_cleared = false,
                        CtrlMirror mirror = CtrlMirror());
int launch_chunk_colors(const FrameK &f, const gsr_camera &cam, const gsr_gaussians &g, int r0, int r1, int num_visible, GeomWS &ws,
                        bool debug, hipStream_t s);
int launch_chunk_binning(const FrameK &f, int c, int r0, int r1, uint64_t n_max, uint64_t emitted_before, GeomWS &gw, BinningWS &bw,
                         ImageWS &iw,
                         int *sort_result, bool debug, hipStream_t s, bool filtered = false);
int launch_open_update(const FrameK &f, GeomWS &gw, ImageWS &iw, bool debug, hipStream_t s, CtrlMirror mirror = CtrlMirror());
// Zero fill of up to eleven arrays in one go (the sparse path's "memset"): the segments are laid end to end in a virtual float index
// space (lengths rounded up to 4 floats).
struct ZeroSegs {
    float *ptr[11];
    size_t len[11];      // floats to clear in segment i
    size_t end[11];      // exclusive end of segment i in the virtual index space
    int n;
};
ZeroSegs zero_segments(const FrameK &f, const gsr_gaussians &g, float *screen, const gsr_grads &out, uint8_t *row_valid = nullptr,
                       size_t valid_bytes = 0);
// groups: the chunk's splats are small (fewer than 4.5 tiles per Gaussian): the quadrant-group kernel (gsr_render.hip)
int launch_render_fwd(const FrameK &f, const gsr_camera &cam, int c, bool last_chunk, int sort_result, const GeomWS &gw, const BinningWS &bw,
                      ImageWS &iw, float *out_color, bool debug, hipStream_t s, bool groups = false);
// rows_upper: bound of the instances the chunks that ran emitted (sizes the launch: the unit count lives on the device)
int launch_render_bwd(const FrameK &f, int chunks_run, int sort_result, long long rows_upper, const GeomWS &gw, BinningWS &bw,
                      const ImageWS &iw, const float *out_color, const float *dL_dcolor, bool debug, hipStream_t s);
// row_valid / valid_bytes: also clear that many bytes of the blend backward's row-valid flags
int launch_zero_segments(const ZeroSegs &z, hipStream_t s);
int launch_zero_outputs(const FrameK &f, const gsr_gaussians &g, float *screen, const gsr_grads &out, hipStream_t s,
                        uint8_t *row_valid = nullptr, size_t valid_bytes = 0);
int launch_reduce_rows(const FrameK &f, const gsr_frame_plan &plan, const GeomWS &gw, const BinningWS &bw, float *screen_grads,
                       int prezeroed, bool debug, hipStream_t s);
// Depth ranks that can own a gradient, as far as the host knows: the chunks that ran, a chunk that went through the live filter
// counted as nothing (only the Gaussians that still reached an open tile were binned: few, and how few is the device's knowledge).
// Decides "sparse geometry backward + zero fill" against "dense geometry backward" (sparse below P / 4).
inline long long effective_binned_ranks(const gsr_frame_plan &plan)
{
    long long n = 0;
    const int chunks = (plan.num_rendered > 0 && plan.chunks_run > 0) ? plan.chunks_run : 0;
    for (int c = 0; c < chunks && c < GSR_MAX_CHUNKS; ++c)
        if (!((plan.chunks_filtered >> c) & 1)) n += plan.chunk_rank_begin[c + 1] - plan.chunk_rank_begin[c];
    return n;
}
// own_frame_sparse: the gradients are this frame's own (gsr_backward_render of the same plan) and the sparse path was chosen from
// effective_binned_ranks: ranks that emitted no instance (GeomWS::cnt_open) are skipped before anything of theirs is read
int launch_geom_bwd(const FrameK &f, const gsr_camera &cam, const gsr_gaussians &g, const int32_t *radii, const GeomWS &gw,
                    const float *screen_grads, int g0, int g1, int n_ranks, const gsr_grads &out, bool debug, hipStream_t s,
                    const uint32_t *rows = nullptr, bool own_frame_sparse = false);
int launch_mark_visible(int P, const float *means3D, const float *view, uint8_t *present, hipStream_t s);

}  // namespace gsr
