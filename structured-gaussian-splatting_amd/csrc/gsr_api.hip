// gsr_api.hip — the extern "C" entry points of libgsrast.so (include/gsrast.h).
// Host-side only: argument validation, workspace carving, kernel sequencing on the caller's stream.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "gsr_internal.h"

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
    if (d->P < 0 || d->width <= 0 || d->height <= 0) {
        set_error("bad frame: P=%d width=%d height=%d", d->P, d->width, d->height);
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
    *binning_bytes = carve_binning(nullptr, num_rendered).total;
    return GSR_OK;
}

int gsr_forward_preprocess(const gsr_frame_desc *desc, const gsr_camera *cam, const gsr_gaussians *g, void *geom_ws,
                           int32_t *radii, int64_t *num_rendered_host, void *stream)
{
    int rc = validate(desc);
    if (rc) return rc;
    if ((rc = validate_inputs(desc, cam, g))) return rc;
    if (!num_rendered_host || (desc->P > 0 && (!geom_ws || !radii))) { set_error("NULL output"); return GSR_ERR_INVALID_ARGUMENT; }
    hipStream_t s = (hipStream_t)stream;
    const FrameK f = make_frame(*desc);
    *num_rendered_host = 0;
    if (f.P == 0) return GSR_OK;
    GeomWS gw = carve_geom(geom_ws, f.P);
    if ((rc = launch_preprocess(f, *cam, *g, gw, radii, desc->debug != 0, s))) return rc;
    if ((rc = launch_scan(gw, f.P, desc->debug != 0, s))) return rc;
    // The one host synchronisation of the path: R sizes the binning workspace.
    uint32_t total = 0;
    GSR_HIP_CHECK(hipMemcpyAsync(&total, gw.offsets + (f.P - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GSR_HIP_CHECK(hipStreamSynchronize(s));
    *num_rendered_host = (int64_t)total;
    return GSR_OK;
}

int gsr_forward_render(const gsr_frame_desc *desc, const gsr_camera *cam, void *geom_ws, void *binning_ws, void *image_ws,
                       int64_t num_rendered, float *out_color, void *stream)
{
    int rc = validate(desc);
    if (rc) return rc;
    if (!cam || !cam->bg || !image_ws || !out_color || num_rendered < 0 || (num_rendered > 0 && (!binning_ws || !geom_ws))) {
        set_error("gsr_forward_render: NULL argument");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = (hipStream_t)stream;
    const bool dbg = desc->debug != 0;
    const FrameK f = make_frame(*desc);
    GeomWS gw = carve_geom(geom_ws, f.P);
    ImageWS iw = carve_image(image_ws, f);
    BinningWS bw = carve_binning(binning_ws, num_rendered);
    int result = 0;
    // the duplicate kernel takes the integer radius from the splat record, so this stage depends on the
    // workspaces alone
    if ((rc = launch_duplicate(f, gw, bw, num_rendered, dbg, s))) return rc;
    if ((rc = launch_sort(f, bw, num_rendered, &result, dbg, s))) return rc;
    if ((rc = launch_ranges(f, bw, result, iw, num_rendered, dbg, s))) return rc;
    if ((rc = launch_render_fwd(f, *cam, gw, bw, iw, out_color, dbg, s))) return rc;
    return GSR_OK;
}

int gsr_backward_render(const gsr_frame_desc *desc, const gsr_camera *cam, const void *geom_ws, void *binning_ws,
                        const void *image_ws, int64_t num_rendered, const float *dL_dcolor, float *screen_grads, void *stream)
{
    int rc = validate(desc);
    if (rc) return rc;
    if (!cam || !cam->bg || !dL_dcolor || (desc->P > 0 && (!screen_grads || !geom_ws)) || !image_ws ||
        (num_rendered > 0 && !binning_ws)) {
        set_error("gsr_backward_render: NULL argument");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = (hipStream_t)stream;
    const bool dbg = desc->debug != 0;
    const FrameK f = make_frame(*desc);
    if (f.P == 0) return GSR_OK;
    GeomWS gw = carve_geom(const_cast<void *>(geom_ws), f.P);
    ImageWS iw = carve_image(const_cast<void *>(image_ws), f);
    BinningWS bw = carve_binning(binning_ws, num_rendered);
    if (num_rendered > 0 && (rc = launch_render_bwd(f, *cam, gw, bw, iw, dL_dcolor, dbg, s))) return rc;
    if ((rc = launch_reduce_rows(f, gw, bw, screen_grads, dbg, s))) return rc;
    return GSR_OK;
}

int gsr_backward_geom(const gsr_frame_desc *desc, const gsr_camera *cam, const gsr_gaussians *g, const int32_t *radii,
                      const void *geom_ws, const float *screen_grads, int32_t g_begin, int32_t g_end, const gsr_grads *out,
                      void *stream)
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
    return launch_geom_bwd(f, *cam, *g, radii, gw, screen_grads, g_begin, g_end, *out, desc->debug != 0, (hipStream_t)stream);
}

int gsr_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix, uint8_t *present,
                     void *stream)
{
    (void)projmatrix;
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) { set_error("gsr_mark_visible: bad argument"); return GSR_ERR_INVALID_ARGUMENT; }
    return launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
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
                        int64_t num_rendered, gsr_debug_views *v)
{
    int rc = validate(desc);
    if (rc) return rc;
    if (!v) { set_error("NULL views"); return GSR_ERR_INVALID_ARGUMENT; }
    memset(v, 0, sizeof *v);
    const FrameK f = make_frame(*desc);
    if (geom_ws) {
        GeomWS gw = carve_geom(const_cast<void *>(geom_ws), f.P);
        v->splat_records = reinterpret_cast<const float *>(gw.records);
        v->tiles_touched = gw.tiles_touched; v->point_offsets = gw.offsets; v->clamped = gw.clamped;
    }
    if (binning_ws) {
        BinningWS bw = carve_binning(const_cast<void *>(binning_ws), num_rendered);
        v->sorted_keys = nullptr;     // key buffers are scratch once the ranges are known
        v->sorted_gaussian = bw.sorted_gid;
    }
    if (image_ws) {
        ImageWS iw = carve_image(const_cast<void *>(image_ws), f);
        v->ranges = reinterpret_cast<const uint32_t *>(iw.ranges);
        v->final_T = iw.final_T; v->n_contrib = iw.n_contrib;
    }
    return GSR_OK;
}

}  // extern "C"
