// host_harness.cpp — TEST INFRASTRUCTURE.  Compiles the product's per-Gaussian maths header
// (structured-gaussian-splatting_amd/csrc/gsr_math.h, the functions the HIP kernels wrap) with g++
// so tests/test_host_math.py can check the formulas against the oracle without a GPU.
// Nothing in the product loads this library.
#include <cstdint>
#include <cstring>

#include "../structured-gaussian-splatting_amd/csrc/gsr_math.h"

using namespace gsr;

static FrameK make_frame(int P, int D, int M, int W, int H, float tanfovx, float tanfovy, float mod, int ty0, int ty1)
{
    FrameK f;
    f.P = P; f.D = D; f.M = M; f.W = W; f.H = H;
    f.Gx = (W + GSR_TILE - 1) / GSR_TILE; f.Gy = (H + GSR_TILE - 1) / GSR_TILE;
    f.ty0 = ty0 < 0 ? 0 : ty0;
    f.ty1 = (ty1 <= 0 || ty1 > f.Gy) ? f.Gy : ty1;
    f.tanfovx = tanfovx; f.tanfovy = tanfovy;
    f.focal_x = (float)W / (2.f * tanfovx); f.focal_y = (float)H / (2.f * tanfovy);
    f.scale_modifier = mod;
    return f;
}

extern "C" void hh_preprocess(int P, int D, int M, int W, int H, float tanfovx, float tanfovy, float mod, int ty0, int ty1,
                              const float *V, const float *PV, const float *campos, const float *means, const float *scales,
                              const float *rots, const float *covpre, const float *opac, const float *shs,
                              const float *colpre, int32_t *radii, uint32_t *tiles, uint8_t *clamped, float *records)
{
    FrameK f = make_frame(P, D, M, W, H, tanfovx, tanfovy, mod, ty0, ty1);
    for (int i = 0; i < P; ++i) {
        PreOut o;
        preprocess_one(f, V, PV, campos, means + 3 * i, scales ? scales + 3 * i : nullptr, rots ? rots + 4 * i : nullptr,
                       covpre ? covpre + 6 * i : nullptr, opac[i], shs ? shs + (size_t)i * M * 3 : nullptr,
                       colpre ? colpre + 3 * i : nullptr, o);
        radii[i] = o.radius; tiles[i] = o.tiles; clamped[i] = (uint8_t)o.clamped;
        std::memcpy(records + 12 * i, &o.s, sizeof(Splat));
    }
}

extern "C" void hh_geom_backward(int P, int D, int M, int W, int H, float tanfovx, float tanfovy, float mod,
                                 const float *V, const float *PV, const float *campos, const float *means,
                                 const float *scales, const float *rots, const float *covpre, const float *shs,
                                 int has_colpre, const int32_t *radii, const uint8_t *clamped, const float *screen9,
                                 float *dmeans3D, float *dmeans2D, float *dsh, float *dcolors, float *dopac,
                                 float *dscales, float *drots, float *dcov)
{
    FrameK f = make_frame(P, D, M, W, H, tanfovx, tanfovy, mod, 0, 0);
    for (int i = 0; i < P; ++i) {
        if (radii[i] <= 0) continue;
        GeomGrad g;
        geom_backward_one(f, V, PV, campos, means + 3 * i, scales ? scales + 3 * i : nullptr, rots ? rots + 4 * i : nullptr,
                          covpre ? covpre + 6 * i : nullptr, shs ? shs + (size_t)i * M * 3 : nullptr, has_colpre != 0,
                          clamped[i], screen9 + 9 * i, g, (shs && dsh) ? dsh + (size_t)i * M * 3 : nullptr, shs && dsh);
        for (int k = 0; k < 3; ++k) { dmeans3D[3 * i + k] = g.dmean[k]; dcolors[3 * i + k] = g.dcolor[k]; dscales[3 * i + k] = g.dscale[k]; }
        dmeans2D[3 * i] = g.dmean2D[0]; dmeans2D[3 * i + 1] = g.dmean2D[1]; dmeans2D[3 * i + 2] = 0.f;
        dopac[i] = g.dopacity;
        for (int k = 0; k < 4; ++k) drots[4 * i + k] = g.drot[k];
        for (int k = 0; k < 6; ++k) dcov[6 * i + k] = g.dcov[k];
    }
}

extern "C" void hh_tile_may_contribute(int n, const float *sx, const float *sy, const float *A, const float *B, const float *C,
                                       const float *op, const int32_t *tx, const int32_t *ty, uint8_t *out)
{
    for (int i = 0; i < n; ++i) out[i] = tile_may_contribute(sx[i], sy[i], A[i], B[i], C[i], op[i], tx[i], ty[i]) ? 1 : 0;
}

// A.3: packed covariance from scale * modifier and a (normalised) quaternion
extern "C" void hh_cov3d(int n, const float *scales, const float *quats, float mod, float *cov6)
{
    for (int i = 0; i < n; ++i) cov3d_from_scale_rot(scales + 3 * i, mod, quats + 4 * i, cov6 + 6 * i);
}

// sub-tile culling on the pre-scaled record: 4-bit quadrant mask per (splat, tile)
extern "C" void hh_quadrant_mask(int n, const float *rec12, const int32_t *tx, const int32_t *ty, uint8_t *out)
{
    for (int i = 0; i < n; ++i) {
        const float *r = rec12 + 12 * (size_t)i;
        out[i] = (uint8_t)quadrant_mask_q(r[0], r[1], r[2], r[3], r[4], r[5], (float)(tx[i] * GSR_TILE), (float)(ty[i] * GSR_TILE));
    }
}

extern "C" void hh_quadrant_mask_bbox(int n, const float *rec12, const int32_t *tx, const int32_t *ty, uint8_t *out)
{
    for (int i = 0; i < n; ++i) {
        const float *r = rec12 + 12 * (size_t)i;
        float xe, ye;
        splat_extent_q(r[2], r[3], r[4], r[5], xe, ye);
        out[i] = (uint8_t)quadrant_mask_bbox(r[0], r[1], xe, ye, (float)(tx[i] * GSR_TILE), (float)(ty[i] * GSR_TILE));
    }
}

// a14: activations of the raw parameters and their chain rule (gsr_math.h activate_raw / activate_raw_backward).
// in: log_scales[n,3], raw_q[n,4], logits[n]; upstream d_scale[n,3], d_q[n,4], d_op[n]
// out: scale[n,3], q[n,4], op[n] and the gradients w.r.t. the raw values in g_ls[n,3], g_rq[n,4], g_logit[n]
extern "C" void hh_activate_raw(int n, const float *log_scales, const float *raw_q, const float *logits, const float *d_scale,
                                const float *d_q, const float *d_op, float *scale, float *q, float *op, float *g_ls,
                                float *g_rq, float *g_logit)
{
    for (int i = 0; i < n; ++i) {
        RawAct a;
        activate_raw(log_scales + 3 * i, raw_q + 4 * i, logits[i], a);
        for (int k = 0; k < 3; ++k) scale[3 * i + k] = a.scale[k];
        for (int k = 0; k < 4; ++k) q[4 * i + k] = a.q[k];
        op[i] = a.opacity;
        GeomGrad g;
        for (int k = 0; k < 3; ++k) g.dscale[k] = d_scale[3 * i + k];
        for (int k = 0; k < 4; ++k) g.drot[k] = d_q[4 * i + k];
        g.dopacity = d_op[i];
        activate_raw_backward(a, g);
        for (int k = 0; k < 3; ++k) g_ls[3 * i + k] = g.dscale[k];
        for (int k = 0; k < 4; ++k) g_rq[4 * i + k] = g.drot[k];
        g_logit[i] = g.dopacity;
    }
}
