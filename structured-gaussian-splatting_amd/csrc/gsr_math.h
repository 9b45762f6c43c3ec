// gsr_math.h — per-Gaussian fp32 maths of the rasterizer (forward preprocess and its backward),
// written once as GSR_HD inline functions.  The HIP kernels (gsr_geom.hip) are thin load/store
// wrappers around these; tests/host_harness.cpp compiles the same functions with g++ so the
// formulas can be checked against the oracle without a GPU (test infrastructure, not a CPU path:
// nothing in the product loads that harness).
//
// Spec: SURVEY.md Appendix A.1-A.6 (forward), A.10 (backward).  Reference twins are cited inline
// (paths relative to /root/reference).
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/gsr_constants.h"

#if defined(__HIPCC__)
#define GSR_HD __host__ __device__ __forceinline__
#else
#define GSR_HD static inline
#endif

namespace gsr {

struct FrameK {            // kernel-side scalars derived from gsr_frame_desc
    int P, D, M, W, H, Gx, Gy, ty0, ty1;
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
};

// 48-byte record consumed by the binning and blend kernels (3 x float4).  The conic and the opacity are stored
// PRE-SCALED so that a pixel's alpha is ONE exp2 of a two-FMA quadratic:
//   qA = -0.5 log2(e) A,  qB = -log2(e) B,  qC = -0.5 log2(e) C,  lop = log2(opacity)
//   p(d) = qA dx^2 + qB dx dy + qC dy^2 + lop  =  log2(opacity * exp(power(d)))      =>  alpha = min(0.99, 2^p)
//   power(d) > 0  <=>  p > lop.
constexpr float kLog2e = 1.4426950408889634f;
struct Splat {
    float x, y, cA, cB;    // pixel-space mean, qA, qB
    float cC, op, r, g;    // qC, lop, colour r g
    float b, depth, rect_x, rect_y; // rect_x / rect_y: bit patterns of (x0 | x1 << 16), (y0 | y1 << 16): the tile
                                    // rectangle the binning walks (tight_rect below), slab-clipped
};

struct TileRect { int x0, y0, x1, y1; };

// ---- A.5: tile rectangle of a splat (slab-clipped in y).  int() truncates toward zero.
GSR_HD TileRect tile_rect(float px, float py, float radius, const FrameK &f)
{
    TileRect r;
    r.x0 = (int)((px - radius) / (float)GSR_TILE);
    r.y0 = (int)((py - radius) / (float)GSR_TILE);
    r.x1 = (int)((px + radius + (float)(GSR_TILE - 1)) / (float)GSR_TILE);
    r.y1 = (int)((py + radius + (float)(GSR_TILE - 1)) / (float)GSR_TILE);
    r.x0 = r.x0 < 0 ? 0 : (r.x0 > f.Gx ? f.Gx : r.x0);
    r.x1 = r.x1 < 0 ? 0 : (r.x1 > f.Gx ? f.Gx : r.x1);
    r.y0 = r.y0 < 0 ? 0 : (r.y0 > f.Gy ? f.Gy : r.y0);
    r.y1 = r.y1 < 0 ? 0 : (r.y1 > f.Gy ? f.Gy : r.y1);
    return r;
}

// ---- exact tile culling.  Returns false only when NO pixel of tile (tile_x, tile_y) can pass the blend's
// acceptance test (power <= 0 and alpha = min(0.99, op * exp(power)) >= 1/255, A.8) for this splat, so dropping
// the (tile, splat) instance changes no pixel and no gradient.  The maximum of the concave quadratic
// power(d) = -1/2 (A dx^2 + C dy^2) - B dx dy over the tile's pixel-centre rectangle (a superset of its 256
// pixels) is compared with the power at which alpha reaches 1/255, minus a margin that covers the blend
// kernels' fp32 evaluation error (few 1e-7 of the term magnitudes) and their exp2-based exponential (1e-6).
// A, B, C, opacity back from the pre-scaled record fields.
GSR_HD void unscale_conic(float qA, float qB, float qC, float lop, float &A, float &B, float &C, float &op)
{
    A = qA * (-2.f / kLog2e); B = qB * (-1.f / kLog2e); C = qC * (-2.f / kLog2e); op = exp2f(lop);
}

GSR_HD bool tile_may_contribute(float sx, float sy, float A, float B, float C, float op, int tile_x, int tile_y)
{
    if (op < (float)GSR_ALPHA_MIN) return false;             // op * G <= op < 1/255 for every G <= 1
    if (!(A > 0.f) || !(C > 0.f) || !(A * C - B * B > 0.f)) return true;     // not a proper ellipse: keep
    // d = splat - pixel, pixel in [16 t, 16 t + 15]
    const float dx0 = sx - (float)(tile_x * GSR_TILE + GSR_TILE - 1), dx1 = sx - (float)(tile_x * GSR_TILE);
    const float dy0 = sy - (float)(tile_y * GSR_TILE + GSR_TILE - 1), dy1 = sy - (float)(tile_y * GSR_TILE);
    if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return true;   // centre inside the rectangle
    float qmax = -3.0e38f;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int e = 0; e < 2; ++e) {
        const float ex = e ? dx1 : dx0;                         // edge dx = ex, dy free
        float dy = fminf(dy1, fmaxf(dy0, -B * ex / C));
        qmax = fmaxf(qmax, -0.5f * (A * ex * ex + C * dy * dy) - B * ex * dy);
        const float ey = e ? dy1 : dy0;                         // edge dy = ey, dx free
        float dx = fminf(dx1, fmaxf(dx0, -B * ey / A));
        qmax = fmaxf(qmax, -0.5f * (A * dx * dx + C * ey * ey) - B * dx * ey);
    }
    const float mx = fmaxf(fabsf(dx0), fabsf(dx1)), my = fmaxf(fabsf(dy0), fabsf(dy1));
    const float S = 0.5f * A * mx * mx + 0.5f * C * my * my + fabsf(B) * mx * my;
    const float need = -logf(255.f * op);                       // power at which op * exp(power) = 1/255
    return qmax >= need - (0.01f + 1e-5f * S);
}

// ---- sub-tile culling, same bound as tile_may_contribute on an arbitrary pixel rectangle [x0, x1] x [y0, y1] (pixel
// centres, inclusive), written on the PRE-SCALED record fields (Splat): p(d) = qA dx^2 + qB dx dy + qC dy^2 + lop is
// log2(opacity * exp(power(d))), a pixel accepts the splat iff log2(1/255) <= p and p <= lop.  Returns false only when
// no pixel of the rectangle can accept it (margins as in tile_may_contribute, in log2 units).
GSR_HD bool rect_may_contribute_q(float sx, float sy, float qA, float qB, float qC, float lop, float x0, float x1, float y0,
                                  float y1)
{
    const float kLog2AlphaMin = -7.994353437f;                  // log2(1/255)
    if (lop < kLog2AlphaMin) return false;
    if (!(qA < 0.f) || !(qC < 0.f) || !(4.f * qA * qC - qB * qB > 0.f)) return true;     // not a proper ellipse: keep
    const float dx0 = sx - x1, dx1 = sx - x0, dy0 = sy - y1, dy1 = sy - y0;         // d = splat - pixel
    if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return true;          // centre inside the rectangle
    const float hC = -0.5f * qB / qC, hA = -0.5f * qB / qA;     // vertex of the 1-D quadratic along an edge
    float qmax = -3.0e38f;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int e = 0; e < 2; ++e) {
        const float ex = e ? dx1 : dx0;                             // edge dx = ex, dy free
        const float dy = fminf(dy1, fmaxf(dy0, hC * ex));
        qmax = fmaxf(qmax, (qA * ex + qB * dy) * ex + qC * dy * dy);
        const float ey = e ? dy1 : dy0;                             // edge dy = ey, dx free
        const float dx = fminf(dx1, fmaxf(dx0, hA * ey));
        qmax = fmaxf(qmax, (qC * ey + qB * dx) * ey + qA * dx * dx);
    }
    const float mx = fmaxf(fabsf(dx0), fabsf(dx1)), my = fmaxf(fabsf(dy0), fabsf(dy1));
    const float S = -qA * mx * mx - qC * my * my + fabsf(qB) * mx * my;
    return qmax >= (kLog2AlphaMin - lop) - (0.0145f + 1e-5f * S);
}

// Bit k (k = qx + 2 qy) set <=> the 8x8 quadrant (qx, qy) of the 16x16 tile whose first pixel is (px0, py0) may hold a
// pixel that accepts the splat.  The blend kernels map one quadrant to one pixel per lane and skip clear quadrants.
GSR_HD unsigned quadrant_mask_q(float sx, float sy, float qA, float qB, float qC, float lop, float px0, float py0)
{
    unsigned m = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; ++k) {
        const float x0 = px0 + (float)((k & 1) * (GSR_TILE / 2)), y0 = py0 + (float)((k >> 1) * (GSR_TILE / 2));
        if (rect_may_contribute_q(sx, sy, qA, qB, qC, lop, x0, x0 + (float)(GSR_TILE / 2 - 1), y0, y0 + (float)(GSR_TILE / 2 - 1)))
            m |= 1u << k;
    }
    return m;
}

// Cheaper, looser form of quadrant_mask_q for the emit kernels: the axis-aligned bounding box of the region where the
// splat's alpha can reach 1/255 (half-widths xe, ye: two square roots per SPLAT, see splat_extent_q) against each
// quadrant's pixel range: a handful of compares per tile.  Conservative (a superset of quadrant_mask_q's bits).
GSR_HD void splat_extent_q(float qA, float qB, float qC, float lop, float &xe, float &ye)
{
    const float kLog2AlphaMin = -7.994353437f;
    const float L = (lop - kLog2AlphaMin) * 1.001f + 0.02f;    // log2 units; covers rect_may_contribute_q's margins
    const float det = 4.f * qA * qC - qB * qB;
    if (!(qA < 0.f) || !(qC < 0.f) || !(det > 0.f) || !(L > 0.f)) { xe = L > 0.f ? 3.0e38f : -1.f; ye = xe; return; }
    // p(d) - lop = qA dx^2 + qB dx dy + qC dy^2 >= -L  <=>  a dx^2 + 2 b dx dy + c dy^2 <= L with a = -qA, b = -qB/2, c = -qC;
    // extents of that ellipse: |dx| <= sqrt(L c / (a c - b^2)), |dy| <= sqrt(L a / (a c - b^2)); a c - b^2 = det / 4
    const float k = 4.f * L / det;
    xe = sqrtf(k * -qC) * 1.001f + 0.01f;
    ye = sqrtf(k * -qA) * 1.001f + 0.01f;
}

GSR_HD unsigned quadrant_mask_bbox(float sx, float sy, float xe, float ye, float px0, float py0)
{
    if (xe < 0.f) return 0u;
    const float h = (float)(GSR_TILE / 2);
    // quadrant columns [px0, px0 + 7] and [px0 + 8, px0 + 15] against [sx - xe, sx + xe]; rows likewise
    const unsigned mx = ((sx - xe <= px0 + h - 1.f && sx + xe >= px0) ? 1u : 0u) | ((sx - xe <= px0 + 2.f * h - 1.f && sx + xe >= px0 + h) ? 2u : 0u);
    const unsigned my = ((sy - ye <= py0 + h - 1.f && sy + ye >= py0) ? 1u : 0u) | ((sy - ye <= py0 + 2.f * h - 1.f && sy + ye >= py0 + h) ? 2u : 0u);
    // bit k = qx + 2 qy
    return ((my & 1u) ? mx : 0u) | ((my & 2u) ? (mx << 2) : 0u);
}

GSR_HD void slab_clip(TileRect &r, const FrameK &f)
{
    if (r.y0 < f.ty0) r.y0 = f.ty0;
    if (r.y1 > f.ty1) r.y1 = f.ty1;
    if (r.y1 < r.y0) r.y1 = r.y0;
}

// ---- the rectangle the binning actually walks: the reference's A.5 rectangle (a pixel outside it never sees
// the splat, whatever its alpha) INTERSECTED with the bounding box of the region where alpha can reach 1/255:
// |dx| <= sqrt(2 L cov_xx), |dy| <= sqrt(2 L cov_yy), L = ln(255 opacity), cov = 2D covariance incl. dilation.
// Outside that box op * exp(power) < 1/255 for every pixel, so no tile there can take the splat: exact.
// 2 % + 0.5 px of margin covers the blend kernels' fp32 arithmetic.  L <= 0: the splat is never accepted.
GSR_HD TileRect tight_rect(const TileRect &ref, float px, float py, float cov_xx, float cov_yy, float opacity)
{
    TileRect r = ref;
    const float L = logf(255.f * opacity);
    if (!(L > 0.f)) { r.x1 = r.x0; r.y1 = r.y0; return r; }
    const float xe = 1.02f * sqrtf(2.f * L * cov_xx) + 0.5f, ye = 1.02f * sqrtf(2.f * L * cov_yy) + 0.5f;
    const float inv = 1.f / (float)GSR_TILE;
    const int x0 = (int)floorf((px - xe) * inv), x1 = (int)floorf((px + xe) * inv) + 1;
    const int y0 = (int)floorf((py - ye) * inv), y1 = (int)floorf((py + ye) * inv) + 1;
    if (x0 > r.x0) r.x0 = x0;
    if (x1 < r.x1) r.x1 = x1;
    if (y0 > r.y0) r.y0 = y0;
    if (y1 < r.y1) r.y1 = y1;
    if (r.x1 < r.x0) r.x1 = r.x0;
    if (r.y1 < r.y0) r.y1 = r.y0;
    return r;
}

GSR_HD float pack_u16x2(int lo, int hi)
{
    const uint32_t u = ((uint32_t)lo & 0xFFFFu) | (((uint32_t)hi & 0xFFFFu) << 16);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

GSR_HD TileRect unpack_rect(float rx, float ry)
{
    uint32_t ux, uy;
    memcpy(&ux, &rx, 4);
    memcpy(&uy, &ry, 4);
    TileRect r;
    r.x0 = (int)(ux & 0xFFFFu); r.x1 = (int)(ux >> 16);
    r.y0 = (int)(uy & 0xFFFFu); r.y1 = (int)(uy >> 16);
    return r;
}

// ---- A.3: rotation from quaternion (r,x,y,z) used as given; utils/general_utils.py:90-98.
GSR_HD void quat_to_rot(const float q[4], float R[9])
{
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// ---- A.3: Sigma = R diag(mod*s)^2 R^T packed [xx,xy,xz,yy,yz,zz]; scene/gaussian_model.py:25-29.
GSR_HD void cov3d_from_scale_rot(const float s[3], float mod, const float q[4], float cov[6])
{
    float R[9];
    quat_to_rot(q, R);
    const float s0 = mod * s[0], s1 = mod * s[1], s2 = mod * s[2];
    const float v0 = s0 * s0, v1 = s1 * s1, v2 = s2 * s2;
    cov[0] = R[0] * R[0] * v0 + R[1] * R[1] * v1 + R[2] * R[2] * v2;
    cov[1] = R[0] * R[3] * v0 + R[1] * R[4] * v1 + R[2] * R[5] * v2;
    cov[2] = R[0] * R[6] * v0 + R[1] * R[7] * v1 + R[2] * R[8] * v2;
    cov[3] = R[3] * R[3] * v0 + R[4] * R[4] * v1 + R[5] * R[5] * v2;
    cov[4] = R[3] * R[6] * v0 + R[4] * R[7] * v1 + R[5] * R[8] * v2;
    cov[5] = R[6] * R[6] * v0 + R[7] * R[7] * v1 + R[8] * R[8] * v2;
}

// Shared by forward and backward: everything the EWA projection (A.4) derives from p_view and Sigma.
struct Ewa {
    float tx, ty, tz;            // clamped view-space point
    float xmul, ymul;            // 0 where tx/tz (ty/tz) was clamped, else 1
    float T[6];                  // T = J * R_w2c (2x3, row-major)
    float S0[3], S1[3];          // rows of T * Sigma
    float a, b, c;               // 2D covariance entries incl. the 0.3 dilation
};

GSR_HD void ewa_project(const float pv[3], const float cov[6], const float *V, const FrameK &f, Ewa &e)
{
    const float limx = (float)GSR_FOV_CLAMP * f.tanfovx, limy = (float)GSR_FOV_CLAMP * f.tanfovy;
    const float txtz = pv[0] / pv[2], tytz = pv[1] / pv[2];
    e.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    e.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    e.tx = fminf(limx, fmaxf(-limx, txtz)) * pv[2];
    e.ty = fminf(limy, fmaxf(-limy, tytz)) * pv[2];
    e.tz = pv[2];
    const float J00 = f.focal_x / e.tz, J02 = -(f.focal_x * e.tx) / (e.tz * e.tz);
    const float J11 = f.focal_y / e.tz, J12 = -(f.focal_y * e.ty) / (e.tz * e.tz);
    // R_w2c[i][j] = V[j][i] with V the transposed (row-vector) view matrix, scene/cameras.py:54
    e.T[0] = J00 * V[0] + J02 * V[2];  e.T[1] = J00 * V[4] + J02 * V[6];  e.T[2] = J00 * V[8] + J02 * V[10];
    e.T[3] = J11 * V[1] + J12 * V[2];  e.T[4] = J11 * V[5] + J12 * V[6];  e.T[5] = J11 * V[9] + J12 * V[10];
    e.S0[0] = cov[0] * e.T[0] + cov[1] * e.T[1] + cov[2] * e.T[2];
    e.S0[1] = cov[1] * e.T[0] + cov[3] * e.T[1] + cov[4] * e.T[2];
    e.S0[2] = cov[2] * e.T[0] + cov[4] * e.T[1] + cov[5] * e.T[2];
    e.S1[0] = cov[0] * e.T[3] + cov[1] * e.T[4] + cov[2] * e.T[5];
    e.S1[1] = cov[1] * e.T[3] + cov[3] * e.T[4] + cov[4] * e.T[5];
    e.S1[2] = cov[2] * e.T[3] + cov[4] * e.T[4] + cov[5] * e.T[5];
    e.a = e.T[0] * e.S0[0] + e.T[1] * e.S0[1] + e.T[2] * e.S0[2] + (float)GSR_COV2D_DILATE;
    e.b = e.T[0] * e.S1[0] + e.T[1] * e.S1[1] + e.T[2] * e.S1[2];
    e.c = e.T[3] * e.S1[0] + e.T[4] * e.S1[1] + e.T[5] * e.S1[2] + (float)GSR_COV2D_DILATE;
}

// ---- A.6: SH basis (utils/sh_utils.py:74-100) and its derivatives wrt the unit direction.
template <bool WITH_GRAD>
GSR_HD void sh_basis(int D, float x, float y, float z, float *b, float *bx, float *by, float *bz)
{
    b[0] = (float)GSR_SH_C0;
    if (WITH_GRAD) { bx[0] = by[0] = bz[0] = 0.f; }
    if (D > 0) {
        b[1] = -(float)GSR_SH_C1 * y; b[2] = (float)GSR_SH_C1 * z; b[3] = -(float)GSR_SH_C1 * x;
        if (WITH_GRAD) {
            bx[1] = 0.f; by[1] = -(float)GSR_SH_C1; bz[1] = 0.f;
            bx[2] = 0.f; by[2] = 0.f; bz[2] = (float)GSR_SH_C1;
            bx[3] = -(float)GSR_SH_C1; by[3] = 0.f; bz[3] = 0.f;
        }
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = (float)GSR_SH_C2_0 * xy;
            b[5] = (float)GSR_SH_C2_1 * yz;
            b[6] = (float)GSR_SH_C2_2 * (2.f * zz - xx - yy);
            b[7] = (float)GSR_SH_C2_3 * xz;
            b[8] = (float)GSR_SH_C2_4 * (xx - yy);
            if (WITH_GRAD) {
                bx[4] = (float)GSR_SH_C2_0 * y; by[4] = (float)GSR_SH_C2_0 * x; bz[4] = 0.f;
                bx[5] = 0.f; by[5] = (float)GSR_SH_C2_1 * z; bz[5] = (float)GSR_SH_C2_1 * y;
                bx[6] = (float)GSR_SH_C2_2 * -2.f * x; by[6] = (float)GSR_SH_C2_2 * -2.f * y; bz[6] = (float)GSR_SH_C2_2 * 4.f * z;
                bx[7] = (float)GSR_SH_C2_3 * z; by[7] = 0.f; bz[7] = (float)GSR_SH_C2_3 * x;
                bx[8] = (float)GSR_SH_C2_4 * 2.f * x; by[8] = (float)GSR_SH_C2_4 * -2.f * y; bz[8] = 0.f;
            }
            if (D > 2) {
                b[9]  = (float)GSR_SH_C3_0 * y * (3.f * xx - yy);
                b[10] = (float)GSR_SH_C3_1 * xy * z;
                b[11] = (float)GSR_SH_C3_2 * y * (4.f * zz - xx - yy);
                b[12] = (float)GSR_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
                b[13] = (float)GSR_SH_C3_4 * x * (4.f * zz - xx - yy);
                b[14] = (float)GSR_SH_C3_5 * z * (xx - yy);
                b[15] = (float)GSR_SH_C3_6 * x * (xx - 3.f * yy);
                if (WITH_GRAD) {
                    bx[9]  = (float)GSR_SH_C3_0 * 6.f * xy;  by[9]  = (float)GSR_SH_C3_0 * (3.f * xx - 3.f * yy); bz[9] = 0.f;
                    bx[10] = (float)GSR_SH_C3_1 * yz;        by[10] = (float)GSR_SH_C3_1 * xz;  bz[10] = (float)GSR_SH_C3_1 * xy;
                    bx[11] = (float)GSR_SH_C3_2 * -2.f * xy; by[11] = (float)GSR_SH_C3_2 * (4.f * zz - xx - 3.f * yy); bz[11] = (float)GSR_SH_C3_2 * 8.f * yz;
                    bx[12] = (float)GSR_SH_C3_3 * -6.f * xz; by[12] = (float)GSR_SH_C3_3 * -6.f * yz; bz[12] = (float)GSR_SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
                    bx[13] = (float)GSR_SH_C3_4 * (4.f * zz - 3.f * xx - yy); by[13] = (float)GSR_SH_C3_4 * -2.f * xy; bz[13] = (float)GSR_SH_C3_4 * 8.f * xz;
                    bx[14] = (float)GSR_SH_C3_5 * 2.f * xz;  by[14] = (float)GSR_SH_C3_5 * -2.f * yz; bz[14] = (float)GSR_SH_C3_5 * (xx - yy);
                    bx[15] = (float)GSR_SH_C3_6 * (3.f * xx - 3.f * yy); by[15] = (float)GSR_SH_C3_6 * -6.f * xy; bz[15] = 0.f;
                }
            }
        }
    }
}

struct PreOut {
    int radius;            // 0 = invisible
    unsigned tiles;        // tiles touched inside the slab
    unsigned clamped;      // bit c: channel c clamped at 0
    float mass;            // optical mass inside the slab, pixel x nepers (optical_mass below); 0 when no tile is touched
    Splat s;
};

// ---- optical mass of a splat: the integral over the image plane of its optical depth tau(x) = -ln(1 - alpha(x)),
// alpha(x) = opacity * exp(power(x)).  For a 2D Gaussian of covariance Sigma' that integral is
// 2 pi sqrt(det Sigma') * Li2(opacity) (dilogarithm; = opacity * 2 pi sqrt(det) for faint splats).  A pixel takes the
// blend's transmittance cut-off (T < 1e-4, A.8) once the optical depths in front of it sum to ln(1e4) = 9.21, so
// the running sum of masses in depth order, divided by the pixel count, is the frame's mean optical depth: the
// chunk plan (gsr_binning.hip) cuts the depth order where that mean reaches a fixed multiple of 9.21 instead of at
// a swept instance count.  Capped by what the splat's binning rectangle can hold, scaled to the slab's share of it.
GSR_HD float optical_mass(float det_cov2d, float opacity, int rect_tiles_full, int rect_tiles_slab)
{
    if (rect_tiles_full <= 0 || rect_tiles_slab <= 0) return 0.f;
    const float op = fminf(fmaxf(opacity, 0.f), (float)GSR_ALPHA_MAX);
    const float li2 = op * (1.f + op * (0.25f + 0.39f * op * op));             // Li2 on [0, 1] to 2 %
    float m = 6.2831853f * sqrtf(fmaxf(det_cov2d, 0.f)) * li2;
    const float cap = (float)rect_tiles_full * (float)(GSR_TILE * GSR_TILE) * -logf(1.f - op);
    m = fminf(m, cap);
    return m * ((float)rect_tiles_slab / (float)rect_tiles_full);
}

// ---- A.6: colour of one Gaussian from its [K,3] SH coefficients (+0.5, clamp at 0; bit c of `clamped` = channel c
// clamped).  Its own function because the forward evaluates it LAZILY: only for the depth chunks that are binned.
template <int DEG = -1>
GSR_HD void sh_color_one(const FrameK &f, const float *campos, const float p[3], const float *sh, float rgb[3], unsigned &clamped)
{
    float dx = p[0] - campos[0], dy = p[1] - campos[1], dz = p[2] - campos[2];
    const float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
    dx *= inv; dy *= inv; dz *= inv;
    float bas[16];
    const int D = DEG >= 0 ? DEG : f.D;          // DEG >= 0: compile-time degree, loops unroll
    sh_basis<false>(D, dx, dy, dz, bas, nullptr, nullptr, nullptr);
    const int K = (D + 1) * (D + 1);
    clamped = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int ch = 0; ch < 3; ++ch) {
        float acc = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 0; k < K; ++k) acc += bas[k] * sh[3 * k + ch];
        acc += (float)GSR_SH_OFFSET;
        if (acc < 0.f) { clamped |= 1u << ch; acc = 0.f; }
        rgb[ch] = acc;
    }
}

// ---- A.1-A.6 for one Gaussian.  `sh` points at this Gaussian's [M,3] coefficients (or null: then, without a
// precomputed colour, the colour is left at 0 for a later sh_color_one), `colpre` at its precomputed colour (or
// null), `covpre` at its precomputed covariance (or null).
template <int DEG = -1>
GSR_HD void preprocess_one(const FrameK &f, const float *V, const float *PV, const float *campos,
                           const float p[3], const float *scale, const float *quat, const float *covpre,
                           float opacity, const float *sh, const float *colpre, PreOut &o)
{
    o.radius = 0; o.tiles = 0; o.clamped = 0; o.mass = 0.f;
    o.s = Splat{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float pv[3];
    pv[0] = p[0] * V[0] + p[1] * V[4] + p[2] * V[8] + V[12];
    pv[1] = p[0] * V[1] + p[1] * V[5] + p[2] * V[9] + V[13];
    pv[2] = p[0] * V[2] + p[1] * V[6] + p[2] * V[10] + V[14];
    if (pv[2] <= (float)GSR_NEAR_CUT) return;                                   // A.1
    const float hx = p[0] * PV[0] + p[1] * PV[4] + p[2] * PV[8] + PV[12];
    const float hy = p[0] * PV[1] + p[1] * PV[5] + p[2] * PV[9] + PV[13];
    const float hw = p[0] * PV[3] + p[1] * PV[7] + p[2] * PV[11] + PV[15];
    const float pw = 1.f / (hw + (float)GSR_HOM_EPS);                           // A.2
    const float ndcx = hx * pw, ndcy = hy * pw;

    float cov[6];
    if (covpre) { for (int k = 0; k < 6; ++k) cov[k] = covpre[k]; }
    else cov3d_from_scale_rot(scale, f.scale_modifier, quat, cov);

    Ewa e;
    ewa_project(pv, cov, V, f, e);
    const float det = e.a * e.c - e.b * e.b;
    if (det == 0.f) return;
    const float det_inv = 1.f / det;
    const float mid = 0.5f * (e.a + e.c);
    const float disc = fmaxf((float)GSR_LAMBDA_FLOOR, mid * mid - det);
    const float sq = sqrtf(disc);
    const float lmax = fmaxf(mid + sq, mid - sq);
    const float my_radius = ceilf((float)GSR_RADIUS_SIGMAS * sqrtf(lmax));
    const float px = ((ndcx + 1.f) * (float)f.W - 1.f) * 0.5f;
    const float py = ((ndcy + 1.f) * (float)f.H - 1.f) * 0.5f;
    TileRect r = tile_rect(px, py, my_radius, f);
    if ((r.x1 - r.x0) * (r.y1 - r.y0) == 0) return;

    float rgb[3] = {0.f, 0.f, 0.f};
    if (colpre) { rgb[0] = colpre[0]; rgb[1] = colpre[1]; rgb[2] = colpre[2]; }
    else if (sh) sh_color_one<DEG>(f, campos, p, sh, rgb, o.clamped);
    o.radius = (int)my_radius;
    o.s.x = px; o.s.y = py;
    o.s.cA = (-0.5f * kLog2e) * (e.c * det_inv); o.s.cB = -kLog2e * (-e.b * det_inv); o.s.cC = (-0.5f * kLog2e) * (e.a * det_inv);
    o.s.op = log2f(opacity);
    o.s.r = rgb[0]; o.s.g = rgb[1]; o.s.b = rgb[2];
    o.s.depth = pv[2];
    TileRect t = tight_rect(r, px, py, e.a, e.c, opacity);
    const int full_tiles = (t.x1 - t.x0) * (t.y1 - t.y0);
    slab_clip(t, f);
    o.s.rect_x = pack_u16x2(t.x0, t.x1);
    o.s.rect_y = pack_u16x2(t.y0, t.y1);
    o.tiles = (unsigned)((t.x1 - t.x0) * (t.y1 - t.y0));
    o.mass = optical_mass(det, opacity, full_tiles, (int)o.tiles);
}

struct GeomGrad {
    float dmean[3], dmean2D[2], dopacity, dcolor[3], dscale[3], drot[4], dcov[6];
};

// ---- A.10 for one visible Gaussian.  sg = (dmean2D.x, dmean2D.y, gA, gB, gC, dopacity, drgb[3]).
// dsh (this Gaussian's [K,3] row, written when want_dsh) receives basis_k * dRGB for k < K; the caller zeroes
// k >= K.  It may point anywhere (registers, an LDS staging row, the output tensor itself) but never depends on a
// run-time select, so that a register array stays in registers.  clamped: bit c set <=> channel c was clamped.
template <int DEG = -1>
GSR_HD void geom_backward_one(const FrameK &f, const float *V, const float *PV, const float *campos,
                              const float p[3], const float *scale, const float *quat, const float *covpre,
                              const float *sh, bool has_colpre, unsigned clamped, const float sg[9],
                              GeomGrad &g, float *dsh, bool want_dsh = true)
{
    float pv[3];
    pv[0] = p[0] * V[0] + p[1] * V[4] + p[2] * V[8] + V[12];
    pv[1] = p[0] * V[1] + p[1] * V[5] + p[2] * V[9] + V[13];
    pv[2] = p[0] * V[2] + p[1] * V[6] + p[2] * V[10] + V[14];
    float cov[6];
    if (covpre) { for (int k = 0; k < 6; ++k) cov[k] = covpre[k]; }
    else cov3d_from_scale_rot(scale, f.scale_modifier, quat, cov);
    Ewa e;
    ewa_project(pv, cov, V, f, e);

    g.dmean2D[0] = sg[0]; g.dmean2D[1] = sg[1];
    g.dopacity = sg[5];

    // conic -> 2D covariance (stored gB is half the derivative wrt the scalar B, A.9 note)
    const float den = e.a * e.c - e.b * e.b;
    const float k2 = 1.f / (den * den + (float)GSR_CONIC_BWD_EPS);
    const float gA = sg[2], gB = sg[3], gC = sg[4];
    const float dL_da = k2 * (-e.c * e.c * gA + 2.f * e.b * e.c * gB + (den - e.a * e.c) * gC);
    const float dL_dc = k2 * (-e.a * e.a * gC + 2.f * e.a * e.b * gB + (den - e.a * e.c) * gA);
    const float dL_db = k2 * 2.f * (e.b * e.c * gA - (den + 2.f * e.b * e.b) * gB + e.a * e.b * gC);
    const float *T = e.T;
    g.dcov[0] = T[0] * T[0] * dL_da + T[0] * T[3] * dL_db + T[3] * T[3] * dL_dc;
    g.dcov[3] = T[1] * T[1] * dL_da + T[1] * T[4] * dL_db + T[4] * T[4] * dL_dc;
    g.dcov[5] = T[2] * T[2] * dL_da + T[2] * T[5] * dL_db + T[5] * T[5] * dL_dc;
    g.dcov[1] = 2.f * T[0] * T[1] * dL_da + (T[0] * T[4] + T[1] * T[3]) * dL_db + 2.f * T[3] * T[4] * dL_dc;
    g.dcov[2] = 2.f * T[0] * T[2] * dL_da + (T[0] * T[5] + T[2] * T[3]) * dL_db + 2.f * T[3] * T[5] * dL_dc;
    g.dcov[4] = 2.f * T[2] * T[1] * dL_da + (T[1] * T[5] + T[2] * T[4]) * dL_db + 2.f * T[4] * T[5] * dL_dc;
    // dL/dT = 2 G2 T Sigma
    const float dT00 = 2.f * e.S0[0] * dL_da + e.S1[0] * dL_db, dT01 = 2.f * e.S0[1] * dL_da + e.S1[1] * dL_db,
                dT02 = 2.f * e.S0[2] * dL_da + e.S1[2] * dL_db;
    const float dT10 = 2.f * e.S1[0] * dL_dc + e.S0[0] * dL_db, dT11 = 2.f * e.S1[1] * dL_dc + e.S0[1] * dL_db,
                dT12 = 2.f * e.S1[2] * dL_dc + e.S0[2] * dL_db;
    // dL/dJ (non-zeros of J) = dL/dT R_w2c^T
    const float dJ00 = V[0] * dT00 + V[4] * dT01 + V[8] * dT02;
    const float dJ02 = V[2] * dT00 + V[6] * dT01 + V[10] * dT02;
    const float dJ11 = V[1] * dT10 + V[5] * dT11 + V[9] * dT12;
    const float dJ12 = V[2] * dT10 + V[6] * dT11 + V[10] * dT12;
    const float tzi = 1.f / e.tz, tz2 = tzi * tzi, tz3 = tz2 * tzi;
    const float dtx = e.xmul * -f.focal_x * tz2 * dJ02;
    const float dty = e.ymul * -f.focal_y * tz2 * dJ12;
    const float dtz = -f.focal_x * tz2 * dJ00 - f.focal_y * tz2 * dJ11 + (2.f * f.focal_x * e.tx) * tz3 * dJ02 +
                      (2.f * f.focal_y * e.ty) * tz3 * dJ12;
    g.dmean[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
    g.dmean[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
    g.dmean[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;

    // projection path
    const float hx = p[0] * PV[0] + p[1] * PV[4] + p[2] * PV[8] + PV[12];
    const float hy = p[0] * PV[1] + p[1] * PV[5] + p[2] * PV[9] + PV[13];
    const float hw = p[0] * PV[3] + p[1] * PV[7] + p[2] * PV[11] + PV[15];
    const float mw = 1.f / (hw + (float)GSR_HOM_EPS);
    const float mul1 = hx * mw * mw, mul2 = hy * mw * mw;
    g.dmean[0] += (PV[0] * mw - PV[3] * mul1) * sg[0] + (PV[1] * mw - PV[3] * mul2) * sg[1];
    g.dmean[1] += (PV[4] * mw - PV[7] * mul1) * sg[0] + (PV[5] * mw - PV[7] * mul2) * sg[1];
    g.dmean[2] += (PV[8] * mw - PV[11] * mul1) * sg[0] + (PV[9] * mw - PV[11] * mul2) * sg[1];

    // colour path
    g.dcolor[0] = sg[6]; g.dcolor[1] = sg[7]; g.dcolor[2] = sg[8];
    if (!has_colpre) {
        const float ox = p[0] - campos[0], oy = p[1] - campos[1], oz = p[2] - campos[2];
        const float inv = 1.f / sqrtf(ox * ox + oy * oy + oz * oz);
        const float dxn = ox * inv, dyn = oy * inv, dzn = oz * inv;
        float bas[16], bx[16], by[16], bz[16];
        const int D = DEG >= 0 ? DEG : f.D;
        sh_basis<true>(D, dxn, dyn, dzn, bas, bx, by, bz);
        const int K = (D + 1) * (D + 1);
        float ddx = 0.f, ddy = 0.f, ddz = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int ch = 0; ch < 3; ++ch) {
            const float dRGB = ((clamped >> ch) & 1u) ? 0.f : sg[6 + ch];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int k = 0; k < K; ++k) {
                const float c = sh[3 * k + ch] * dRGB;
                if (want_dsh) dsh[3 * k + ch] = bas[k] * dRGB;
                ddx += bx[k] * c; ddy += by[k] * c; ddz += bz[k] * c;
            }
        }
        const float dot = dxn * ddx + dyn * ddy + dzn * ddz;           // through normalize()
        g.dmean[0] += (ddx - dxn * dot) * inv;
        g.dmean[1] += (ddy - dyn * dot) * inv;
        g.dmean[2] += (ddz - dzn * dot) * inv;
    }

    // 3D covariance path
    for (int k = 0; k < 3; ++k) g.dscale[k] = 0.f;
    for (int k = 0; k < 4; ++k) g.drot[k] = 0.f;
    if (!covpre) {
        float R[9];
        quat_to_rot(quat, R);
        const float s[3] = {f.scale_modifier * scale[0], f.scale_modifier * scale[1], f.scale_modifier * scale[2]};
        const float G3[9] = {g.dcov[0], 0.5f * g.dcov[1], 0.5f * g.dcov[2], 0.5f * g.dcov[1], g.dcov[3],
                             0.5f * g.dcov[4], 0.5f * g.dcov[2], 0.5f * g.dcov[4], g.dcov[5]};
        float gR[9];
        for (int k = 0; k < 3; ++k) {
            float ds = 0.f;
            for (int j = 0; j < 3; ++j) {
                // dM[k][j] = 2 * sum_l s_k R[l][k] G3[l][j]
                const float dM = 2.f * s[k] * (R[k] * G3[j] + R[3 + k] * G3[3 + j] + R[6 + k] * G3[6 + j]);
                ds += R[3 * j + k] * dM;
                gR[3 * j + k] = s[k] * dM;
            }
            g.dscale[k] = ds;                     // scale_modifier factor omitted (A.10)
        }
        const float r = quat[0], x = quat[1], y = quat[2], z = quat[3];
        g.drot[0] = 2.f * (-z * gR[1] + y * gR[2] + z * gR[3] - x * gR[5] - y * gR[6] + x * gR[7]);
        g.drot[1] = 2.f * (y * gR[1] + z * gR[2] + y * gR[3] - 2.f * x * gR[4] - r * gR[5] + z * gR[6] + r * gR[7] - 2.f * x * gR[8]);
        g.drot[2] = 2.f * (-2.f * y * gR[0] + x * gR[1] + r * gR[2] + x * gR[3] + z * gR[5] - r * gR[6] + z * gR[7] - 2.f * y * gR[8]);
        g.drot[3] = 2.f * (-2.f * z * gR[0] - r * gR[1] + x * gR[2] + r * gR[3] - 2.f * z * gR[4] + y * gR[5] + x * gR[6] + y * gR[7]);
    }
}

// ---- the activations of the reference's parameter store, for the raw-parameter mode
// (scene/gaussian_model.py:47-60: scaling exp, opacity sigmoid, rotation torch.nn.functional.normalize with
// eps 1e-12; :108-127 getters).  activate_raw_backward turns the gradients w.r.t. the activated values into
// gradients w.r.t. the raw parameters, in place.
struct RawAct {
    float scale[3], q[4], opacity, inv_norm;
    bool clamped_norm;     // |raw quaternion| < eps: normalize() divided by eps, not by the norm
};

GSR_HD void activate_raw(const float log_scale[3], const float raw_q[4], float logit, RawAct &a)
{
    for (int k = 0; k < 3; ++k) a.scale[k] = expf(log_scale[k]);
    const float n = sqrtf(raw_q[0] * raw_q[0] + raw_q[1] * raw_q[1] + raw_q[2] * raw_q[2] + raw_q[3] * raw_q[3]);
    a.clamped_norm = n < 1e-12f;
    a.inv_norm = 1.f / fmaxf(n, 1e-12f);
    for (int k = 0; k < 4; ++k) a.q[k] = raw_q[k] * a.inv_norm;
    a.opacity = 1.f / (1.f + expf(-logit));
}

GSR_HD void activate_raw_backward(const RawAct &a, GeomGrad &g)
{
    for (int k = 0; k < 3; ++k) g.dscale[k] *= a.scale[k];
    const float dot = a.clamped_norm ? 0.f : a.q[0] * g.drot[0] + a.q[1] * g.drot[1] + a.q[2] * g.drot[2] + a.q[3] * g.drot[3];
    for (int k = 0; k < 4; ++k) g.drot[k] = (g.drot[k] - a.q[k] * dot) * a.inv_norm;
    g.dopacity *= a.opacity * (1.f - a.opacity);
}

// ---- A.1 alone (markVisible).
GSR_HD bool in_frustum(const float p[3], const float *V)
{
    return (p[0] * V[2] + p[1] * V[6] + p[2] * V[10] + V[14]) > (float)GSR_NEAR_CUT;
}

}  // namespace gsr
