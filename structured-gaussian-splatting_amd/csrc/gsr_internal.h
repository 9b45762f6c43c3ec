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
struct GeomWS {                 // O(P): the reference's geomBuffer
    float4 *records;            // [P,3]  Splat records
    uint32_t *tiles_touched;    // [P]
    uint32_t *offsets;          // [P]    inclusive scan of tiles_touched
    uint8_t *clamped;           // [P]
    void *scan_temp; size_t scan_temp_bytes;
    size_t total;
};
struct ImageWS {                // O(N + Tn): the reference's imgBuffer
    float *final_T;             // [N]
    int32_t *n_contrib;         // [N]
    uint2 *ranges;              // [Tn]
    size_t total;
};
struct BinningWS {              // O(R): the reference's binningBuffer
    uint64_t *keys[2];          // [R] x2 (radix double buffer)
    uint32_t *vals[2];          // [R] x2   payload = instance slot (index in duplicate order)
    uint32_t *inst_gid;         // [R] slot -> Gaussian
    uint32_t *sorted_gid;       // [R] sorted position -> Gaussian
    uint32_t *sorted_slot;      // [R] sorted position -> slot (the sorted payload, kept for the backward)
    float *grad_rows;           // [R, kRowFloats] per-instance screen-space gradient rows (backward)
    void *sort_temp; size_t sort_temp_bytes;
    size_t total;
};

size_t scan_temp_bytes(int P);
size_t sort_temp_bytes(int64_t R);
GeomWS carve_geom(void *base, int P);
ImageWS carve_image(void *base, const FrameK &f);
BinningWS carve_binning(void *base, int64_t R);

// ---- kernel launchers (each returns a gsr_status)
int launch_preprocess(const FrameK &f, const gsr_camera &cam, const gsr_gaussians &g, GeomWS &ws, int32_t *radii,
                      bool debug, hipStream_t s);
int launch_scan(GeomWS &ws, int P, bool debug, hipStream_t s);
int launch_duplicate(const FrameK &f, const GeomWS &gw, BinningWS &bw, int64_t R, bool debug, hipStream_t s);
int launch_sort(const FrameK &f, BinningWS &bw, int64_t R, int *result_buffer, bool debug, hipStream_t s);
int launch_ranges(const FrameK &f, BinningWS &bw, int result_buffer, ImageWS &iw, int64_t R, bool debug, hipStream_t s);
int launch_render_fwd(const FrameK &f, const gsr_camera &cam, const GeomWS &gw, const BinningWS &bw, ImageWS &iw,
                      float *out_color, bool debug, hipStream_t s);
int launch_render_bwd(const FrameK &f, const gsr_camera &cam, const GeomWS &gw, BinningWS &bw, const ImageWS &iw,
                      const float *dL_dcolor, bool debug, hipStream_t s);
int launch_reduce_rows(const FrameK &f, const GeomWS &gw, const BinningWS &bw, float *screen_grads, bool debug,
                       hipStream_t s);
int launch_geom_bwd(const FrameK &f, const gsr_camera &cam, const gsr_gaussians &g, const int32_t *radii, const GeomWS &gw,
                    const float *screen_grads, int g0, int g1, const gsr_grads &out, bool debug, hipStream_t s);
int launch_mark_visible(int P, const float *means3D, const float *view, uint8_t *present, hipStream_t s);

}  // namespace gsr
