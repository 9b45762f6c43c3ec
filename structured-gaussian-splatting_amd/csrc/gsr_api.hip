// gsr_api.hip — the extern "C" entry points of libgsrast.so (include/gsrast.h).
// Host-side only: argument validation, workspace carving, kernel sequencing on the caller's stream.
#include <stdarg.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "gsr_internal.h"
#ifndef GSR_FILL_BEFORE_WAIT
#define GSR_FILL_BEFORE_WAIT 1
#endif

namespace gsr {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ---- per-kernel timing: process-wide (the backward runs on the autograd worker thread) list of
// (name, start, stop) event pairs behind a mutex; events pooled.  Opt-in, off by default.
struct ProfEntry { char name[GSR_PROFILE_NAME_LEN]; hipEvent_t a, b; };
struct ProfState {
    bool on = false;
    std::vector<ProfEntry> pending;
    std::vector<hipEvent_t> pool;
    struct Acc { char name[GSR_PROFILE_NAME_LEN]; double ms; int n; };
    std::vector<Acc> acc;
};
static ProfState g_prof;
static std::mutex g_prof_mu;

bool profile_on() { return g_prof.on; }

static hipEvent_t prof_event()
{
    if (!g_prof.pool.empty()) { hipEvent_t e = g_prof.pool.back(); g_prof.pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

int profile_begin(const char *name, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfEntry e;
    strncpy(e.name, name, GSR_PROFILE_NAME_LEN - 1);
    e.name[GSR_PROFILE_NAME_LEN - 1] = 0;
    e.a = prof_event(); e.b = prof_event();
    (void)hipEventRecord(e.a, s);
    g_prof.pending.push_back(e);
    return (int)g_prof.pending.size() - 1;
}

void profile_end(int idx, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (idx >= 0 && idx < (int)g_prof.pending.size()) (void)hipEventRecord(g_prof.pending[idx].b, s);
}

static void profile_drain()
{
    for (auto &e : g_prof.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(e.b) == hipSuccess && hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            bool found = false;
            for (auto &a : g_prof.acc)
                if (!strcmp(a.name, e.name)) { a.ms += ms; a.n += 1; found = true; break; }
            if (!found) { ProfState::Acc a; strcpy(a.name, e.name); a.ms = ms; a.n = 1; g_prof.acc.push_back(a); }
        }
        g_prof.pool.push_back(e.a); g_prof.pool.push_back(e.b);
    }
    g_prof.pending.clear();
}

static int validate(const gsr_frame_desc *d)
{
    if (!d) { set_error("frame descriptor is NULL"); return GSR_ERR_INVALID_ARGUMENT; }
    if (d->P >= (1 << kQuadMaskShift)) {
        set_error("P = %d: at most 2^%d - 1 Gaussians per frame (the sorted instance list packs a 4-bit quadrant mask above the index)",
                  d->P, kQuadMaskShift);
        return GSR_ERR_INVALID_ARGUMENT;
    }
    if (d->P < 0 || d->width <= 0 || d->height <= 0) {
        set_error("bad frame: P=%d width=%d height=%d", d->P, d->width, d->height);
        return GSR_ERR_INVALID_ARGUMENT;
    }
    if (((long long)(d->width + GSR_TILE - 1) / GSR_TILE) * ((long long)(d->height + GSR_TILE - 1) / GSR_TILE) >= (1ll << kUnitTileBits)) {
        set_error("%d x %d pixels: at most 2^%d - 1 tiles per frame (the blend backward's work units pack the tile index)", d->width,
                  d->height, kUnitTileBits);
        return GSR_ERR_INVALID_ARGUMENT;
    }
    if (d->sh_degree < 0 || d->sh_degree > 3) { set_error("sh_degree %d outside 0..3", d->sh_degree); return GSR_ERR_INVALID_ARGUMENT; }
    if (d->sh_coeffs < 0 || d->sh_coeffs > 16) { set_error("sh_coeffs %d outside 0..16", d->sh_coeffs); return GSR_ERR_INVALID_ARGUMENT; }
    if (!(d->tanfovx > 0.f) || !(d->tanfovy > 0.f)) { set_error("tanfov must be positive"); return GSR_ERR_INVALID_ARGUMENT; }
    return GSR_OK;
}

static int validate_inputs(const gsr_frame_desc *d, const gsr_camera *c, const gsr_gaussians *g)
{
    if (!c || !c->bg || !c->viewmatrix || !c->projmatrix || !c->campos) { set_error("camera tensors missing"); return GSR_ERR_INVALID_ARGUMENT; }
    if (!g) { set_error("gaussians missing"); return GSR_ERR_INVALID_ARGUMENT; }
    if (d->P == 0) return GSR_OK;
    if (!g->means3D || !g->opacities) { set_error("means3D / opacities missing"); return GSR_ERR_INVALID_ARGUMENT; }
    if (g->raw) {
        // raw-parameter mode (a14): log-scales, raw quaternions, opacity logits; SH as _features_dc + _features_rest (raw == 1) or
        // as one interleaved table (raw == 2)
        const bool common = g->shs && g->scales && g->rotations && !g->colors_precomp && !g->cov3D_precomp && d->sh_coeffs >= 1;
        if (g->raw == 1 && (!common || ((g->shs_rest == nullptr) != (d->sh_coeffs == 1)))) {
            set_error("raw mode needs shs (features_dc), shs_rest (features_rest, NULL iff sh_coeffs == 1), scales and rotations, "
                      "and neither colors_precomp nor cov3D_precomp");
            return GSR_ERR_INVALID_ARGUMENT;
        }
        if (g->raw == 2 && (!common || g->shs_rest)) {
            set_error("raw mode 2 needs shs (the interleaved [P,sh_coeffs,3] table), no shs_rest, scales and rotations, and neither "
                      "colors_precomp nor cov3D_precomp");
            return GSR_ERR_INVALID_ARGUMENT;
        }
        if (g->raw != 1 && g->raw != 2) { set_error("raw must be 0, 1 or 2"); return GSR_ERR_INVALID_ARGUMENT; }
    } else if (g->shs_rest) {
        set_error("shs_rest is only meaningful with raw != 0");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    if ((g->shs == nullptr) == (g->colors_precomp == nullptr)) {
        set_error("provide exactly one of shs / colors_precomp");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    const bool sr = g->scales && g->rotations;
    if (sr == (g->cov3D_precomp != nullptr) || (!sr && (g->scales || g->rotations))) {
        set_error("provide exactly one of (scales, rotations) / cov3D_precomp");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    if (g->shs && (d->sh_degree + 1) * (d->sh_degree + 1) > d->sh_coeffs) {
        set_error("active SH degree %d needs %d coefficients, shs holds %d", d->sh_degree,
                  (d->sh_degree + 1) * (d->sh_degree + 1), d->sh_coeffs);
        return GSR_ERR_INVALID_ARGUMENT;
    }
    return GSR_OK;
}

// Control-block readback: the two host decisions of a frame (R / chunk plan, open-tile count) wait for 250 bytes.  The kernel
// in front of each decision (k_open_count) writes the block straight into host-coherent pinned memory and releases a sequence
// number behind it (CtrlMirror); the host polls that word.  (Round 1 copied the block with hipMemcpyAsync and polled an event:
// the copy is a 5 us kernel of its own and the event another packet, both on the critical path of a 0.8 ms step.)  A frame
// whose geometry workspace is absent, or a stream that drains without the word arriving, falls back to the copy.
constexpr double kReadbackTimeoutS = 30.0;
// experiment switches (tools/): read ONCE, when the library is loaded — a re-run of gsr_forward_render after GSR_ERR_WORKSPACE must
// take the decisions of the first run, whatever the environment does in between
static const bool kNoLiveFilter = getenv("GSR_NO_LIVE_FILTER") != nullptr, kNoChunkMerge = getenv("GSR_NO_CHUNK_MERGE") != nullptr;
static const int kFwdGroups = getenv("GSR_FWD_GROUPS") ? atoi(getenv("GSR_FWD_GROUPS")) : -1;       // 0 / 1: force the blend forward's kernel

static inline void cpu_pause()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#endif
}

struct Readback {            // one per host thread, never freed (pinned words + an event; the HIP runtime may already be gone
    uint32_t *words = nullptr;   // when thread-local destructors run).  words: Ctrl, then the sequence word
    uint32_t *words_dev = nullptr;
    uint32_t seq = 0;
    hipEvent_t ev = nullptr;
};
static thread_local Readback tl_readback;
constexpr int kCtrlWords = (int)(sizeof(Ctrl) / 4);

static int readback_init(Readback &rb)
{
    if (rb.words) return GSR_OK;
    void *p = nullptr;
    GSR_HIP_CHECK(hipHostMalloc(&p, sizeof(Ctrl) + 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
    memset(p, 0, sizeof(Ctrl) + 64);
    void *d = nullptr;
    GSR_HIP_CHECK(hipHostGetDevicePointer(&d, p, 0));
    GSR_HIP_CHECK(hipEventCreateWithFlags(&rb.ev, hipEventDisableTiming));
    rb.words = static_cast<uint32_t *>(p);
    rb.words_dev = static_cast<uint32_t *>(d);
    return GSR_OK;
}

// the mirror to hand to the kernel in front of the next wait_ctrl() of this thread
static int next_mirror(CtrlMirror *m)
{
    Readback &rb = tl_readback;
    int rc = readback_init(rb);
    if (rc) return rc;
    if (++rb.seq == 0) rb.seq = 1;
    m->words = rb.words_dev; m->seq = rb.seq;
    return GSR_OK;
}

static int read_ctrl_by_copy(const Ctrl *dev, Ctrl *out, hipStream_t s)
{
    Readback &rb = tl_readback;
    int rc = readback_init(rb);
    if (rc) return rc;
    GSR_HIP_CHECK(hipMemcpyAsync(rb.words, dev, sizeof(Ctrl), hipMemcpyDeviceToHost, s));
    GSR_HIP_CHECK(hipStreamSynchronize(s));
    memcpy(out, rb.words, sizeof(Ctrl));
    return GSR_OK;
}

// Wait for the block the kernel given next_mirror()'s mirror publishes.  Polls, politely and not for ever: a few thousand
// loads back to back (the word normally lands within microseconds of the kernel's end), then with a pause instruction, now and
// then a look at the stream (an error, or a drained stream without the word, ends the wait), and after kReadbackTimeoutS an
// error instead of hanging the host on a wedged stream.
static int wait_ctrl(const Ctrl *dev, Ctrl *out, hipStream_t s)
{
    Readback &rb = tl_readback;
    const uint32_t want = rb.seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long long spins = 0;; ++spins) {
        if (__atomic_load_n(&rb.words[kCtrlWords], __ATOMIC_ACQUIRE) == want) break;
        if (spins > 4096) {
            cpu_pause();
            if ((spins & 4095) == 0) {
                const hipError_t e = hipStreamQuery(s);
                if (e == hipSuccess) {                               // drained: the word is there now, or it is not coming
                    if (__atomic_load_n(&rb.words[kCtrlWords], __ATOMIC_ACQUIRE) == want) break;
                    return read_ctrl_by_copy(dev, out, s);
                }
                if (e != hipErrorNotReady) { set_error("control-block readback: %s", hipGetErrorString(e)); return GSR_ERR_HIP; }
                if (spins > (1ull << 20)) std::this_thread::yield();         // a long wait (debugger, huge frame): give the core away
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (waited > kReadbackTimeoutS) {
                    set_error("control-block readback did not complete within %.0f s: the stream is stuck (a kernel of this frame "
                              "hangs or the device was lost)", kReadbackTimeoutS);
                    return GSR_ERR_HIP;
                }
            }
        }
    }
    memcpy(out, rb.words, sizeof(Ctrl));
    return GSR_OK;
}

// Instances the backward's scratch must hold: exact when the forward stopped early (the count was read back with the open-tile
// count); when the LAST chunk ran its count stayed on the device and the bound is what the chunks that ran could have emitted.
static long long rows_bound(const gsr_frame_plan *plan)
{
    long long n = plan->instances_emitted;
    if (n < 0) {
        n = 0;
        for (int c = 0; c < plan->chunks_run && c < GSR_MAX_CHUNKS; ++c) n += plan->chunk_instances_max[c];
    }
    return n < 1 ? 1 : n;
}

// row-valid flags to clear ahead of the blend backward: one byte per instance the chunks that ran can have emitted (never more
// than the binning workspace holds), rounded up to whole 16-byte words of its padded block
static size_t valid_bytes(const gsr_frame_plan *plan)
{
    long long n = rows_bound(plan);
    const long long cap = plan->binning_capacity > 0 ? plan->binning_capacity : plan->num_rendered;
    if (n > cap) n = cap;
    return ((size_t)(n < 1 ? 1 : n) + 15) & ~(size_t)15;
}

}  // namespace gsr

using namespace gsr;

extern "C" {

int gsr_version(void) { return GSR_VERSION; }

const char *gsr_last_error(void) { return g_err; }

int gsr_workspace_sizes(const gsr_frame_desc *desc, size_t *geom_bytes, size_t *image_bytes)
{
    int rc = validate(desc);
    if (rc) return rc;
    const FrameK f = make_frame(*desc);
    if (geom_bytes) *geom_bytes = carve_geom(nullptr, f.P).total;
    if (image_bytes) *image_bytes = carve_image(nullptr, f).total;
    return GSR_OK;
}

int gsr_binning_size(const gsr_frame_desc *desc, int64_t num_rendered, size_t *binning_bytes)
{
    int rc = validate(desc);
    if (rc) return rc;
    if (num_rendered < 0 || !binning_bytes) { set_error("bad num_rendered / NULL out"); return GSR_ERR_INVALID_ARGUMENT; }
    *binning_bytes = carve_binning(nullptr, num_rendered, make_frame(*desc)).total;
    return GSR_OK;
}

int gsr_binning_first_chunk_capacity(const gsr_frame_plan *plan, int64_t *instances)
{
    if (!plan || !instances) { set_error("gsr_binning_first_chunk_capacity: NULL argument"); return GSR_ERR_INVALID_ARGUMENT; }
    // what the first depth chunk can emit at most, plus headroom for a small straggler chunk; never more than R
    int64_t n = plan->num_chunks > 1 ? plan->chunk_instances_max[0] + plan->chunk_instances_max[0] / 4 + (1 << 20) : plan->num_rendered;
    if (n > plan->num_rendered) n = plan->num_rendered;
    *instances = n;
    return GSR_OK;
}

int gsr_forward_preprocess(const gsr_frame_desc *desc, const gsr_camera *cam, const gsr_gaussians *g, void *geom_ws,
                           void *image_ws, int32_t *radii, gsr_frame_plan *plan, void *stream)
{
    int rc = validate(desc);
    if (rc) return rc;
    if ((rc = validate_inputs(desc, cam, g))) return rc;
    if (!plan || (desc->P > 0 && (!geom_ws || !radii))) { set_error("NULL output"); return GSR_ERR_INVALID_ARGUMENT; }
    hipStream_t s = (hipStream_t)stream;
    const bool dbg = desc->debug != 0;
    const FrameK f = make_frame(*desc);
    memset(plan, 0, sizeof *plan);
    plan->num_chunks = 1;
    if (f.P == 0) return GSR_OK;
    GeomWS gw = carve_geom(geom_ws, f.P);
    // (the selection's histograms are cleared by the preprocess kernel, the tile ranges by the partition's scatter kernel)
    if ((rc = launch_preprocess(f, *cam, *g, gw, radii, desc->prefiltered != 0, dbg, s))) return rc;
    if (image_ws) {                                  // stage 2's reset of ranges / open flags, in the shadow of the readback
        ImageWS iw = carve_image(image_ws, f);
        if ((rc = launch_depth_select(f, gw, dbg, s, iw.ranges, binning_clear_bytes(f, iw) / 16))) return rc;      // chunk plan + partition by chunk
        CtrlMirror mirror;
        if ((rc = next_mirror(&mirror))) return rc;
        if ((rc = launch_binning_init(f, gw, iw, dbg, s, true, mirror))) return rc;
        plan->binning_initialised = 1;
    } else if ((rc = launch_depth_select(f, gw, dbg, s))) return rc;
    // The one host synchronisation of this stage: the plan (R sizes the binning workspace).
    Ctrl h;
    if ((rc = image_ws ? wait_ctrl(gw.ctrl, &h, s) : read_ctrl_by_copy(gw.ctrl, &h, s))) return rc;
    if (desc->prefiltered && h.prefilter_violation) {
        set_error("prefiltered is set but at least one Gaussian fails the frustum test (view depth <= 0.2): the caller's "
                  "pre-filter and the rasterizer disagree");
        return GSR_ERR_PREFILTERED;
    }
    if (h.overflow) {
        set_error("the frame's tile instances (sum of tiles touched over %d visible Gaussians) exceed 2^32 - 1: "
                  "split the frame into tile-row slabs (gsr_frame_desc.tile_row_begin/end)", (int)h.V);
        return GSR_ERR_INVALID_ARGUMENT;
    }
    plan->num_rendered = (int64_t)h.R_total;
    plan->num_visible = (int32_t)h.V;
    plan->num_chunks = h.num_chunks > 0 ? (int32_t)h.num_chunks : 1;
    for (int c = 0; c <= GSR_MAX_CHUNKS; ++c) plan->chunk_rank_begin[c] = (int32_t)h.bnd[c];
    for (int c = 0; c < GSR_MAX_CHUNKS; ++c) plan->chunk_instances_max[c] = (int64_t)h.chunk_full[c];
    for (int c = 0; c < GSR_MAX_CHUNKS; ++c) plan->chunk_key_end[c] = h.key_end[c];
    plan->key_max = h.key_max;
    if (h.num_chunks == 0)
        for (int c = 0; c <= GSR_MAX_CHUNKS; ++c) plan->chunk_rank_begin[c] = 0;
    return GSR_OK;
}

// fill (optional): a zero fill to enqueue behind the FIRST chunk's blend and readback kernels, ahead of the wait for that readback
// (gsr_forward's early fill); *fill_done reports whether it was enqueued
static int forward_render_impl(const gsr_frame_desc *desc, const gsr_camera *cam, const gsr_gaussians *g, void *geom_ws, void *binning_ws,
                               void *image_ws, gsr_frame_plan *plan, float *out_color, void *stream, const ZeroSegs *fill, bool *fill_done)
{
    int rc = validate(desc);
    if (rc) return rc;
    if ((rc = validate_inputs(desc, cam, g))) return rc;
    if (!cam || !cam->bg || !image_ws || !out_color || !plan || plan->num_rendered < 0 ||
        (desc->P > 0 && !geom_ws) || (plan->num_rendered > 0 && !binning_ws)) {
        set_error("gsr_forward_render: NULL argument");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    if (plan->num_chunks < 1 || plan->num_chunks > GSR_MAX_CHUNKS) { set_error("bad plan (num_chunks %d)", plan->num_chunks); return GSR_ERR_INVALID_ARGUMENT; }
    hipStream_t s = (hipStream_t)stream;
    const bool dbg = desc->debug != 0;
    const FrameK f = make_frame(*desc);
    ImageWS iw = carve_image(image_ws, f);
    plan->chunks_run = 0; plan->instances_emitted = 0; plan->sort_result = 0; plan->tile_order_ready = 0;
    if (f.P == 0 || plan->num_rendered == 0) {
        // nothing to bin: one blend pass over empty ranges writes the background
        GeomWS gw0 = carve_geom(geom_ws, f.P);
        BinningWS bw0 = carve_binning(binning_ws, 0, f);
        if (f.P == 0 || !geom_ws) gw0.ctrl = iw.ctrl_scratch;   // no geometry workspace at all
        if ((rc = launch_binning_init(f, gw0, iw, dbg, s))) return rc;
        if ((rc = launch_render_fwd(f, *cam, 0, true, 0, gw0, bw0, iw, out_color, dbg, s))) return rc;
        plan->chunks_run = 1;
        return GSR_OK;
    }
    GeomWS gw = carve_geom(geom_ws, f.P);
    // The binning workspace holds `binning_capacity` instances (0: the upper bound R, as the reference sizes its
    // binningBuffer).  A frame whose tiles saturate runs one small depth chunk and touches a few per cent of R, so
    // callers may size it for the first chunk only; a later chunk that does not fit stops the frame with
    // GSR_ERR_WORKSPACE before anything of that chunk is written (re-run this stage with a workspace for R instances).
    const int64_t capacity = plan->binning_capacity > 0 ? plan->binning_capacity : plan->num_rendered;
    BinningWS bw = carve_binning(binning_ws, capacity, f);
    if (!plan->binning_initialised && (rc = launch_binning_init(f, gw, iw, dbg, s))) return rc;
    plan->binning_initialised = 0;                  // a re-run of this stage must reset the tile ranges / open flags itself
    int sort_result = 0;
    uint64_t emitted_before = 0;                    // instances earlier chunks emitted (exact: read back with the open-tile count)
    uint32_t open_now = 0xFFFFFFFFu, stuck_now = 0; // tiles still open before the current chunk, and those nothing covers yet (known from chunk 1 on)
    constexpr int kLiveFilterMin = 1 << 16;         // the filter's launches pay off from this many Gaussians on
    // Early stop: one control-block readback per chunk (~10 us of stream idle, measured); the chunk plan keeps
    // the number of chunks at three or fewer.
    for (int c = 0; c < plan->num_chunks; ++c) {
        // A frame that is not going to close (an uncovered corner: some open tile still has a pixel that nothing has covered after
        // the chunks so far, k_render_fwd marks those): every further chunk costs its fixed ~250 us for little.  All remaining chunks then become ONE, which goes
        // through the live filter below (its Gaussians' relative keys are re-based there).
        LiveParts parts{1, {0}, {0}};
        const long long slab_tiles_now = (long long)(f.ty1 - f.ty0) * f.Gx;
        if (c > 0 && c >= plan->chunks_sorted && c < plan->num_chunks - 1 && stuck_now > 0 && (long long)open_now * 2 < slab_tiles_now &&
            !kNoLiveFilter && !kNoChunkMerge) {
            const int last_c = plan->num_chunks - 1;
            uint64_t m_max = 0;
            for (int j = c; j <= last_c; ++j) m_max += (uint64_t)plan->chunk_instances_max[j];
            const uint64_t m_n = (uint64_t)(plan->chunk_rank_begin[last_c + 1] - plan->chunk_rank_begin[c]);
            if (m_n >= (uint64_t)kLiveFilterMin && m_max / 32 + 2 * m_n + 4 <= m_max && m_max <= 0xFFFFFFFFull) {
                if (emitted_before + m_max > (uint64_t)capacity) {          // (before the plan is touched: the re-run decides the same)
                    set_error("binning workspace holds %lld instances, the remaining depth chunks may need %llu: re-run "
                              "gsr_forward_render with binning_capacity = num_rendered (%lld)", (long long)capacity,
                              (unsigned long long)(emitted_before + m_max), (long long)plan->num_rendered);
                    return GSR_ERR_WORKSPACE;
                }
                auto base_of = [&](int j) { const uint32_t e = plan->chunk_key_end[j - 1]; return e < 0x3E4CCCCDu ? 0x3E4CCCCDu : e; };
                parts.n = last_c - c + 1;
                for (int j = c; j <= last_c; ++j) {
                    parts.end[j - c] = (uint32_t)(plan->chunk_rank_begin[j + 1] - plan->chunk_rank_begin[c]);
                    parts.delta[j - c] = base_of(j) - base_of(c);
                }
                plan->chunk_rank_begin[c + 1] = plan->chunk_rank_begin[last_c + 1];
                plan->chunk_key_end[c] = plan->chunk_key_end[last_c];
                plan->chunk_instances_max[c] = (int64_t)m_max;
                plan->num_chunks = c + 1;
            }
        }
        const bool last = c == plan->num_chunks - 1;
        const int r0 = plan->chunk_rank_begin[c], r1 = plan->chunk_rank_begin[c + 1];
        if (emitted_before + (uint64_t)plan->chunk_instances_max[c] > (uint64_t)capacity) {
            set_error("binning workspace holds %lld instances, depth chunk %d may need %llu: re-run gsr_forward_render with "
                      "binning_capacity = num_rendered (%lld)", (long long)capacity, c,
                      (unsigned long long)(emitted_before + (uint64_t)plan->chunk_instances_max[c]), (long long)plan->num_rendered);
            return GSR_ERR_WORKSPACE;
        }
        // the chunk's Gaussians were selected by depth; now that it is needed, put them in (depth, index) order
        // (once: a re-run of this stage after GSR_ERR_WORKSPACE finds the earlier chunks sorted, and the sort's inputs consumed)
        // A late, large chunk with most tiles closed (a frame with a corner no splat covers runs every chunk, and the last one is
        // most of the scene): first move the Gaussians whose rectangle still holds an open tile to the front of its range
        // (launch_live_filter) — only those get sorted and binned, one wave each.
        const uint64_t chunk_n = (uint64_t)(r1 - r0), chunk_max = (uint64_t)plan->chunk_instances_max[c];
        const bool filtered = c > 0 && r1 - r0 >= kLiveFilterMin && (long long)open_now * 2 < (long long)(f.ty1 - f.ty0) * f.Gx &&
                              chunk_max / 32 + 2 * chunk_n + 4 <= chunk_max && !kNoLiveFilter;
        if (filtered) plan->chunks_filtered |= 1 << c;
        if (c >= plan->chunks_sorted) {
            if (filtered && (rc = launch_live_filter(f, c, r0, r1, parts, gw, iw, dbg, s))) return rc;
            // (the sort's key span: up to the chunk's end, and never beyond the frame's largest key — the last chunk's end is "everything")
            const uint32_t key_hi = plan->key_max != 0u && plan->key_max < plan->chunk_key_end[c] ? plan->key_max : plan->chunk_key_end[c];
            if ((rc = launch_chunk_order(f, r0, r1, c > 0 ? plan->chunk_key_end[c - 1] : 0u, key_hi, c == 0, gw, dbg, s,
                                         filtered ? &gw.ctrl->chunk_live[c] : nullptr)))
                return rc;
            plan->chunks_sorted = c + 1;
        }
        if ((rc = launch_chunk_binning(f, c, r0, r1, chunk_max, emitted_before, gw, bw, iw, &sort_result, dbg, s,
                                       (plan->chunks_filtered >> c) & 1)))
            return rc;
        if ((rc = launch_chunk_colors(f, *cam, *g, r0, r1, plan->num_visible, gw, dbg, s))) return rc;      // A.6 for this chunk's Gaussians only
        // small splats (fewer than 4.5 tiles per Gaussian on average; filtered chunks bin a small, unknown part of their bound):
        // the blend forward with one 16-lane group per quadrant
        const bool small_splats = kFwdGroups >= 0 ? kFwdGroups != 0 : (!((plan->chunks_filtered >> c) & 1) && chunk_n > 0 && chunk_max * 2 < chunk_n * 9);
        if ((rc = launch_render_fwd(f, *cam, c, last, sort_result, gw, bw, iw, out_color, dbg, s, small_splats))) return rc;
        plan->chunks_run = c + 1;
        plan->instances_emitted = -1;                 // the last chunk's count stays on the device
        if (last) break;
        CtrlMirror mirror;
        if ((rc = next_mirror(&mirror))) return rc;
        if ((rc = launch_open_update(f, gw, iw, dbg, s, mirror))) return rc;
        if (fill) {                                   // behind the kernel that posts the readback, ahead of the wait for it
            ProfileScope prof("zero_outputs", s);
            if ((rc = launch_zero_segments(*fill, s))) return rc;
            *fill_done = true; fill = nullptr;
        }
        Ctrl h;
        if ((rc = wait_ctrl(gw.ctrl, &h, s))) return rc;
        plan->instances_emitted = (int64_t)h.chunk_base[c + 1];
        emitted_before = (uint64_t)h.chunk_base[c + 1];
        open_now = h.open_count;
        stuck_now = h.open_stuck;
        if (h.open_count == 0) break;
    }
    plan->sort_result = sort_result;
    return GSR_OK;
}

int gsr_forward_render(const gsr_frame_desc *desc, const gsr_camera *cam, const gsr_gaussians *g, void *geom_ws, void *binning_ws,
                       void *image_ws, gsr_frame_plan *plan, float *out_color, void *stream)
{
    bool unused = false;
    return forward_render_impl(desc, cam, g, geom_ws, binning_ws, image_ws, plan, out_color, stream, nullptr, &unused);
}

int gsr_forward(const gsr_frame_desc *desc, const gsr_camera *cam, const gsr_gaussians *g, void *geom_ws, void *image_ws, int32_t *radii,
                gsr_frame_plan *plan, void *binning_ws, int64_t binning_capacity, float *out_color, gsr_grads *early_fill, void *stream)
{
    int rc = gsr_forward_preprocess(desc, cam, g, geom_ws, image_ws, radii, plan, stream);
    if (rc) return rc;
    int64_t need = 0;
    if ((rc = gsr_binning_first_chunk_capacity(plan, &need))) return rc;
    if (plan->num_rendered > 0 && (!binning_ws || binning_capacity < need)) {
        set_error("gsr_forward: the binning workspace holds %lld instances, the first depth chunk may need %lld: allocate and call "
                  "gsr_forward_render", (long long)binning_capacity, (long long)need);
        return GSR_ERR_WORKSPACE;
    }
    plan->binning_capacity = binning_capacity < plan->num_rendered ? binning_capacity : plan->num_rendered;
    if (plan->binning_capacity <= 0) plan->binning_capacity = 0;
    // The early fill: the zeros of a depth-complex frame's gradient tensors + the blend backward's row flags, 50 us of HBM writes at
    // 1M Gaussians.  It is enqueued behind the first chunk's blend and the kernel that posts its readback, BEFORE the host waits, so that it
    // runs while the host is idle in the wait and then busy getting back to its caller and on to the next launch (~60 us in which the
    // GPU has nothing else to do).  Whether the frame ends up sparse is only known after the readback: the fill is enqueued when the
    // first planned chunk alone is sparse - a frame whose first chunk holds a quarter of the Gaussians takes the dense path whatever
    // follows - and is wasted bandwidth, nothing more, when later chunks make the frame dense after all.  (Measured and dropped: the
    // fill on a side stream beside the blend - the cross-queue waits cost more than it hid; the fill as the first job of every wave
    // of the blend forward - 17 us there instead of 47, but the gap after the readback was empty again: 0.694 ms per step, not 0.678.)
    const bool fill_wanted = early_fill && !early_fill->prezeroed && desc->P > 0 && plan->num_rendered > 0;
    const FrameK f = make_frame(*desc);
    ZeroSegs fill;
    fill.n = 0;
    bool filled = false;
    if (GSR_FILL_BEFORE_WAIT && fill_wanted && plan->num_chunks > 1 && (long long)plan->chunk_rank_begin[1] * 4 < (long long)desc->P) {
        const long long cap = plan->binning_capacity > 0 ? plan->binning_capacity : plan->num_rendered;
        const BinningWS bw = carve_binning(binning_ws, cap, f);
        fill = zero_segments(f, *g, nullptr, *early_fill, bw.row_valid, ((size_t)(cap < 1 ? 1 : cap) + 15) & ~(size_t)15);
    }
    if ((rc = forward_render_impl(desc, cam, g, geom_ws, binning_ws, image_ws, plan, out_color, stream, fill.n > 0 ? &fill : nullptr, &filled)))
        return rc;
    if (fill_wanted && plan->chunks_run > 0 && effective_binned_ranks(*plan) * 4 < (long long)desc->P) {
        if (!filled) {
            const BinningWS bw = carve_binning(binning_ws, plan->binning_capacity > 0 ? plan->binning_capacity : plan->num_rendered, f);
            ProfileScope prof("zero_outputs", (hipStream_t)stream);
            if ((rc = launch_zero_outputs(f, *g, nullptr, *early_fill, (hipStream_t)stream, bw.row_valid, valid_bytes(plan)))) return rc;
        }
        early_fill->prezeroed = 1;
        plan->tile_order_ready = 1;                      // the fill also cleared the blend backward's row flags
    }
    return GSR_OK;
}

int gsr_backward_rows_size(const gsr_frame_desc *desc, const gsr_frame_plan *plan, size_t *rows_bytes)
{
    int rc = validate(desc);
    if (rc) return rc;
    if (!plan || !rows_bytes) { set_error("gsr_backward_rows_size: NULL argument"); return GSR_ERR_INVALID_ARGUMENT; }
    *rows_bytes = align_up((size_t)rows_bound(plan) * kRowFloats * sizeof(float));
    return GSR_OK;
}

int gsr_backward_prepare(const gsr_frame_desc *desc, const gsr_gaussians *g, gsr_frame_plan *plan, float *screen_grads,
                         gsr_grads *grads, void *stream)
{
    int rc = validate(desc);
    if (rc) return rc;
    if (!g || !plan || !grads) { set_error("gsr_backward_prepare: NULL argument"); return GSR_ERR_INVALID_ARGUMENT; }
    if (desc->P == 0) return GSR_OK;
    const FrameK f = make_frame(*desc);
    ProfileScope prof("zero_outputs", (hipStream_t)stream);
    if ((rc = launch_zero_outputs(f, *g, screen_grads, *grads, (hipStream_t)stream))) return rc;
    if (screen_grads) plan->screen_prezeroed = 1;
    grads->prezeroed = 1;
    return GSR_OK;
}

int gsr_bwd_segment_entries(void) { return kSeg; }

int gsr_backward_render(const gsr_frame_desc *desc, const gsr_camera *cam, const void *geom_ws, void *binning_ws,
                        const void *image_ws, void *rows_ws, const gsr_frame_plan *plan, const float *out_color, const float *dL_dcolor,
                        float *screen_grads, void *stream)
{
    int rc = validate(desc);
    if (rc) return rc;
    if (!cam || !cam->bg || !dL_dcolor || !out_color || !plan || (desc->P > 0 && (!screen_grads || !geom_ws)) || !image_ws ||
        (plan->num_rendered > 0 && (!binning_ws || !rows_ws))) {
        set_error("gsr_backward_render: NULL argument");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = (hipStream_t)stream;
    const bool dbg = desc->debug != 0;
    const FrameK f = make_frame(*desc);
    if (f.P == 0) return GSR_OK;
    GeomWS gw = carve_geom(const_cast<void *>(geom_ws), f.P);
    ImageWS iw = carve_image(const_cast<void *>(image_ws), f);
    BinningWS bw = carve_binning(binning_ws, plan->binning_capacity > 0 ? plan->binning_capacity : plan->num_rendered, f);
    bw.grad_rows = (float *)rows_ws;
    long long rows_upper = 0;
    for (int c = 0; c < plan->chunks_run && c < GSR_MAX_CHUNKS; ++c) rows_upper += plan->chunk_instances_max[c];
    if (plan->num_rendered > 0) {
        if (!plan->tile_order_ready) GSR_HIP_CHECK(hipMemsetAsync(bw.row_valid, 0, valid_bytes(plan), s));
        if ((rc = launch_render_bwd(f, plan->chunks_run, plan->sort_result, plan->instances_emitted >= 0 ? (long long)plan->instances_emitted : rows_upper,
                                    gw, bw, iw, out_color, dL_dcolor, dbg, s)))
            return rc;
    }
    // only the depth ranks of chunks that ran can own gradient rows
    if ((rc = launch_reduce_rows(f, *plan, gw, bw, screen_grads, plan->screen_prezeroed, dbg, s))) return rc;
    return GSR_OK;
}

int gsr_backward_geom(const gsr_frame_desc *desc, const gsr_camera *cam, const gsr_gaussians *g, const int32_t *radii,
                      const void *geom_ws, const float *screen_grads, int32_t g_begin, int32_t g_end, int32_t binned_ranks,
                      const gsr_frame_plan *own_plan, const gsr_grads *out, void *stream)
{
    int rc = validate(desc);
    if (rc) return rc;
    if ((rc = validate_inputs(desc, cam, g))) return rc;
    if (!out) { set_error("gsr_backward_geom: NULL grads"); return GSR_ERR_INVALID_ARGUMENT; }
    if (g_begin < 0 || g_end > desc->P || g_begin > g_end) { set_error("bad Gaussian range [%d, %d)", g_begin, g_end); return GSR_ERR_INVALID_ARGUMENT; }
    if (g_end == g_begin) return GSR_OK;
    if (!radii || !geom_ws || !screen_grads) { set_error("gsr_backward_geom: NULL argument"); return GSR_ERR_INVALID_ARGUMENT; }
    const FrameK f = make_frame(*desc);
    GeomWS gw = carve_geom(const_cast<void *>(geom_ws), f.P);
    // own_plan: the gradients come from gsr_backward_render of THIS frame: the ranks of the chunks that ran, and among them only
    // those that emitted an instance, can be non-zero; sparse (fill + visit those) unless unfiltered chunks hold P / 4 Gaussians or more
    bool own_sparse = false;
    if (own_plan && g_begin == 0 && g_end == desc->P && own_plan->num_rendered > 0 && own_plan->chunks_run > 0) {
        binned_ranks = own_plan->chunk_rank_begin[own_plan->chunks_run];
        own_sparse = effective_binned_ranks(*own_plan) * 4 < (long long)desc->P;
    }
    return launch_geom_bwd(f, *cam, *g, radii, gw, screen_grads, g_begin, g_end, binned_ranks, *out, desc->debug != 0,
                           (hipStream_t)stream, nullptr, own_sparse);
}

int gsr_backward_geom_rows(const gsr_frame_desc *desc, const gsr_camera *cam, const gsr_gaussians *g, const int32_t *radii,
                           const void *geom_ws, const float *screen_grads, const int32_t *rows, int32_t n_rows, const gsr_grads *out,
                           void *stream)
{
    int rc = validate(desc);
    if (rc) return rc;
    if ((rc = validate_inputs(desc, cam, g))) return rc;
    if (!out || n_rows < 0 || n_rows > desc->P || (n_rows > 0 && !rows)) { set_error("gsr_backward_geom_rows: bad argument"); return GSR_ERR_INVALID_ARGUMENT; }
    if (desc->P == 0) return GSR_OK;
    if (!radii || !geom_ws || !screen_grads) { set_error("gsr_backward_geom_rows: NULL argument"); return GSR_ERR_INVALID_ARGUMENT; }
    const FrameK f = make_frame(*desc);
    GeomWS gw = carve_geom(const_cast<void *>(geom_ws), f.P);
    return launch_geom_bwd(f, *cam, *g, radii, gw, screen_grads, 0, f.P, n_rows, *out, desc->debug != 0, (hipStream_t)stream,
                           reinterpret_cast<const uint32_t *>(rows));
}

int gsr_frame_arrays(const gsr_frame_desc *desc, const void *geom_ws, const uint32_t **depth_keys, const uint32_t **depth_order)
{
    int rc = validate(desc);
    if (rc) return rc;
    if (!geom_ws) { set_error("gsr_frame_arrays: NULL workspace"); return GSR_ERR_INVALID_ARGUMENT; }
    GeomWS gw = carve_geom(const_cast<void *>(geom_ws), desc->P);
    if (depth_keys) *depth_keys = gw.sort_keys[0];
    if (depth_order) *depth_order = gw.order;
    return GSR_OK;
}

int gsr_exchange_rows_gather(const gsr_frame_desc *desc, void *geom_ws, uint32_t key_max, const float *screen_grads, int32_t n_rows,
                             int32_t *rows, float *packed, void *stream)
{
    int rc = validate(desc);
    if (rc) return rc;
    if (n_rows < 0 || !geom_ws || (n_rows > 0 && (!screen_grads || !rows || !packed))) {
        set_error("gsr_exchange_rows_gather: bad argument");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    const FrameK f = make_frame(*desc);
    GeomWS gw = carve_geom(geom_ws, f.P);
    return launch_rows_gather(f, gw, key_max, screen_grads, n_rows, rows, packed, desc->debug != 0, (hipStream_t)stream);
}

int gsr_exchange_rows_scatter(const gsr_frame_desc *desc, int32_t n_rows, const int32_t *rows, const float *packed, float *screen_grads,
                              void *stream)
{
    int rc = validate(desc);
    if (rc) return rc;
    if (n_rows < 0 || (n_rows > 0 && (!rows || !packed || !screen_grads))) { set_error("gsr_exchange_rows_scatter: bad argument"); return GSR_ERR_INVALID_ARGUMENT; }
    return launch_rows_scatter(n_rows, desc->P, rows, packed, screen_grads, desc->debug != 0, (hipStream_t)stream);
}

int gsr_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix, uint8_t *present,
                     void *stream)
{
    (void)projmatrix;
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) { set_error("gsr_mark_visible: bad argument"); return GSR_ERR_INVALID_ARGUMENT; }
    return launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
}

int gsr_debug_sort_temp_bytes(size_t *bytes)
{
    if (!bytes) return GSR_ERR_INVALID_ARGUMENT;
    *bytes = radix_temp_bytes() + 256;
    return GSR_OK;
}

int gsr_debug_sort_pairs(uint32_t *keys0, uint32_t *keys1, uint32_t *vals0, uint32_t *vals1, int64_t n, int32_t end_bit,
                         int32_t count_on_device, void *temp, int32_t *result_buffer, void *stream)
{
    if (!keys0 || !keys1 || !vals0 || !vals1 || !temp || !result_buffer || n < 0 || n > 0xFFFFFFFFll || end_bit < 1 || end_bit > 32) {
        set_error("gsr_debug_sort_pairs: bad argument");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = (hipStream_t)stream;
    uint32_t *keys[2] = {keys0, keys1}, *vals[2] = {vals0, vals1};
    uint32_t *n_dev = reinterpret_cast<uint32_t *>((char *)temp + radix_temp_bytes());
    const uint32_t n32 = (uint32_t)n;
    if (count_on_device) GSR_HIP_CHECK(hipMemcpyAsync(n_dev, &n32, 4, hipMemcpyHostToDevice, s));
    int result = 0;
    int rc = launch_radix_sort<uint32_t>(keys, vals, count_on_device ? n_dev : nullptr, n32, (uint64_t)n, nullptr, 0, end_bit, temp,
                                         &result, "debug_sort", false, s);
    *result_buffer = result;
    if (count_on_device) GSR_HIP_CHECK(hipStreamSynchronize(s));     // n32 lives on this stack frame
    return rc;
}

int gsr_profile_enable(int enable)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    profile_drain();
    g_prof.acc.clear();
    g_prof.on = enable != 0;
    return GSR_OK;
}

int gsr_profile_read(int max_entries, char (*names)[GSR_PROFILE_NAME_LEN], float *total_ms, int32_t *launches)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    profile_drain();
    int n = 0;
    for (auto &a : g_prof.acc) {
        if (n >= max_entries) break;
        if (names) strcpy(names[n], a.name);
        if (total_ms) total_ms[n] = (float)a.ms;
        if (launches) launches[n] = a.n;
        ++n;
    }
    return n;
}

int gsr_debug_get_views(const gsr_frame_desc *desc, const void *geom_ws, const void *binning_ws, const void *image_ws,
                        const gsr_frame_plan *plan, gsr_debug_views *v)
{
    int rc = validate(desc);
    if (rc) return rc;
    if (!v || !plan) { set_error("NULL views / plan"); return GSR_ERR_INVALID_ARGUMENT; }
    memset(v, 0, sizeof *v);
    const FrameK f = make_frame(*desc);
    if (geom_ws) {
        GeomWS gw = carve_geom(const_cast<void *>(geom_ws), f.P);
        v->splat_records = reinterpret_cast<const float *>(gw.records);
        v->tiles_touched = reinterpret_cast<const uint32_t *>(gw.tiles_mass); v->depth_order = gw.order; v->point_offsets = gw.offs_full;
        v->clamped = gw.clamped;
    }
    if (binning_ws) {
        BinningWS bw = carve_binning(const_cast<void *>(binning_ws), plan->binning_capacity > 0 ? plan->binning_capacity : plan->num_rendered, f);
        v->sorted_gaussian = bw.gids[1];
        v->bwd_units = reinterpret_cast<const uint32_t *>(bw.units.units); v->bwd_unit_cap_full = bw.units.cap_full; v->bwd_unit_cap_part = bw.units.cap_part;
    }
    if (image_ws) {
        ImageWS iw = carve_image(const_cast<void *>(image_ws), f);
        v->ranges = reinterpret_cast<const uint32_t *>(iw.ranges);
        v->final_T = iw.T_state; v->n_contrib = iw.last_enc;
        v->tile_walk = iw.tile_walk; v->bwd_unit_count = iw.unit_count;
    }
    return GSR_OK;
}

}  // extern "C"
