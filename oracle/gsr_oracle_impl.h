/* gsr_oracle_impl.h — body of the CPU oracle, instantiated twice (float, double)
 * by gsr_oracle.c.  TEST INFRASTRUCTURE ONLY (see gsr_oracle.c header).
 *
 * Every function cites the section of SURVEY.md Appendix A it restates and, where one
 * exists, the in-tree Python twin of the reference (path:line relative to /root/reference).
 *
 * Required macros: REAL (float|double), FN(name) (symbol suffixing), R_SQRT, R_EXP, R_CEIL,
 * R_FABS.
 */

typedef struct FN(ctx) {
    /* configuration */
    int P, D, M, W, H, Gx, Gy, ty0, ty1;
    REAL tanfovx, tanfovy, focal_x, focal_y, scale_modifier, fragile_eps, fragile_pos;
    REAL bg[3], view[16], proj[16], campos[3];
    int has_cov_precomp, has_colors_precomp;
    /* inputs (owned copies) */
    REAL *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    /* per-Gaussian forward state (A.1-A.6) */
    int32_t *radii;
    REAL *xy, *depth, *cov3D, *conic_op, *rgb;
    float *depth_lo, *depth_hi;  /* [P]: the binary32 view depth as evaluated without FMA, with FMA, and rounded from binary64: min / max */
    REAL *kappa;                 /* [P]: (a c + b^2) / det of the 2D covariance: how much binary32 rounding of a, b, c is amplified in the conic */
    uint8_t *clamped;
    int32_t *rect;               /* xmin, ymin, xmax, ymax (tile units, slab-clipped in y) */
    uint32_t *tiles_touched;
    /* binning (A.7) */
    int64_t R;
    uint64_t *keys;
    uint32_t *vals;
    int64_t *ranges;             /* 2 * Tn : [start, end) */
    /* per-pixel forward state (A.8) */
    REAL *color;                 /* [3, H, W] */
    REAL *final_T;               /* [H*W] */
    int32_t *n_contrib;          /* [H*W] */
    uint8_t *fragile_px;         /* [H*W]: a skip/stop decision fell within fragile_eps of its threshold */
    uint8_t *fragile_g;          /* [P]: Gaussian contributes to (or was decided at) a fragile pixel */
    int64_t n_pairs;             /* (pixel, splat) pairs evaluated by the forward blend */
} FN(ctx);

static void *FN(dup)(const REAL *src, size_t n)
{
    if (!src || n == 0) return NULL;
    REAL *d = (REAL *)malloc(n * sizeof(REAL));
    memcpy(d, src, n * sizeof(REAL));
    return d;
}

void FN(free)(FN(ctx) *c)
{
    if (!c) return;
    free(c->means3D); free(c->shs); free(c->colors_precomp); free(c->opacities);
    free(c->scales); free(c->rotations); free(c->cov3D_precomp);
    free(c->radii); free(c->xy); free(c->depth); free(c->cov3D); free(c->conic_op);
    free(c->rgb); free(c->kappa); free(c->depth_lo); free(c->depth_hi); free(c->clamped); free(c->rect); free(c->tiles_touched);
    free(c->keys); free(c->vals); free(c->ranges);
    free(c->color); free(c->final_T); free(c->n_contrib); free(c->fragile_px); free(c->fragile_g);
    free(c);
}

/* ---- A.3: quaternion (r,x,y,z) -> rotation, used as given.
 * Twin: utils/general_utils.py:90-98 (build_rotation, after its own normalisation). */
static void FN(quat_to_rot)(const REAL *q, REAL Rm[9])
{
    REAL r = q[0], x = q[1], y = q[2], z = q[3];
    Rm[0] = 1 - 2 * (y * y + z * z); Rm[1] = 2 * (x * y - r * z);     Rm[2] = 2 * (x * z + r * y);
    Rm[3] = 2 * (x * y + r * z);     Rm[4] = 1 - 2 * (x * x + z * z); Rm[5] = 2 * (y * z - r * x);
    Rm[6] = 2 * (x * z - r * y);     Rm[7] = 2 * (y * z + r * x);     Rm[8] = 1 - 2 * (x * x + y * y);
}

/* ---- A.3: Sigma = R diag(s)^2 R^T, packed [xx,xy,xz,yy,yz,zz].
 * Twins: scene/gaussian_model.py:25-29, utils/general_utils.py:64-73,101-110. */
static void FN(cov3d)(const REAL *scale, REAL mod, const REAL *q, REAL cov[6])
{
    REAL Rm[9];
    FN(quat_to_rot)(q, Rm);
    REAL s0 = mod * scale[0], s1 = mod * scale[1], s2 = mod * scale[2];
    REAL v0 = s0 * s0, v1 = s1 * s1, v2 = s2 * s2;
    cov[0] = Rm[0] * Rm[0] * v0 + Rm[1] * Rm[1] * v1 + Rm[2] * Rm[2] * v2;
    cov[1] = Rm[0] * Rm[3] * v0 + Rm[1] * Rm[4] * v1 + Rm[2] * Rm[5] * v2;
    cov[2] = Rm[0] * Rm[6] * v0 + Rm[1] * Rm[7] * v1 + Rm[2] * Rm[8] * v2;
    cov[3] = Rm[3] * Rm[3] * v0 + Rm[4] * Rm[4] * v1 + Rm[5] * Rm[5] * v2;
    cov[4] = Rm[3] * Rm[6] * v0 + Rm[4] * Rm[7] * v1 + Rm[5] * Rm[8] * v2;
    cov[5] = Rm[6] * Rm[6] * v0 + Rm[7] * Rm[7] * v1 + Rm[8] * Rm[8] * v2;
}

/* ---- A.6: SH basis values for a unit direction; twin utils/sh_utils.py:74-100.
 * basis[k] multiplies sh[k] (coefficient index k, k < (D+1)^2 <= 16). */
static void FN(sh_basis)(int D, REAL x, REAL y, REAL z, REAL b[16])
{
    b[0] = (REAL)GSR_SH_C0;
    if (D > 0) {
        b[1] = -(REAL)GSR_SH_C1 * y;
        b[2] = (REAL)GSR_SH_C1 * z;
        b[3] = -(REAL)GSR_SH_C1 * x;
        if (D > 1) {
            REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = (REAL)GSR_SH_C2_0 * xy;
            b[5] = (REAL)GSR_SH_C2_1 * yz;
            b[6] = (REAL)GSR_SH_C2_2 * (2 * zz - xx - yy);
            b[7] = (REAL)GSR_SH_C2_3 * xz;
            b[8] = (REAL)GSR_SH_C2_4 * (xx - yy);
            if (D > 2) {
                b[9]  = (REAL)GSR_SH_C3_0 * y * (3 * xx - yy);
                b[10] = (REAL)GSR_SH_C3_1 * xy * z;
                b[11] = (REAL)GSR_SH_C3_2 * y * (4 * zz - xx - yy);
                b[12] = (REAL)GSR_SH_C3_3 * z * (2 * zz - 3 * xx - 3 * yy);
                b[13] = (REAL)GSR_SH_C3_4 * x * (4 * zz - xx - yy);
                b[14] = (REAL)GSR_SH_C3_5 * z * (xx - yy);
                b[15] = (REAL)GSR_SH_C3_6 * x * (xx - 3 * yy);
            }
        }
    }
}

/* d basis[k] / d(x,y,z) for the same polynomials (needed by A.10's view-direction path). */
static void FN(sh_basis_grad)(int D, REAL x, REAL y, REAL z, REAL dx[16], REAL dy[16], REAL dz[16])
{
    for (int k = 0; k < 16; ++k) dx[k] = dy[k] = dz[k] = 0;
    if (D > 0) {
        dy[1] = -(REAL)GSR_SH_C1; dz[2] = (REAL)GSR_SH_C1; dx[3] = -(REAL)GSR_SH_C1;
        if (D > 1) {
            REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y;
            dx[4] = (REAL)GSR_SH_C2_0 * y;  dy[4] = (REAL)GSR_SH_C2_0 * x;
            dy[5] = (REAL)GSR_SH_C2_1 * z;  dz[5] = (REAL)GSR_SH_C2_1 * y;
            dx[6] = (REAL)GSR_SH_C2_2 * -2 * x; dy[6] = (REAL)GSR_SH_C2_2 * -2 * y; dz[6] = (REAL)GSR_SH_C2_2 * 4 * z;
            dx[7] = (REAL)GSR_SH_C2_3 * z;  dz[7] = (REAL)GSR_SH_C2_3 * x;
            dx[8] = (REAL)GSR_SH_C2_4 * 2 * x; dy[8] = (REAL)GSR_SH_C2_4 * -2 * y;
            if (D > 2) {
                dx[9]  = (REAL)GSR_SH_C3_0 * 6 * xy;            dy[9]  = (REAL)GSR_SH_C3_0 * (3 * xx - 3 * yy);
                dx[10] = (REAL)GSR_SH_C3_1 * y * z;             dy[10] = (REAL)GSR_SH_C3_1 * x * z;  dz[10] = (REAL)GSR_SH_C3_1 * xy;
                dx[11] = (REAL)GSR_SH_C3_2 * -2 * xy;           dy[11] = (REAL)GSR_SH_C3_2 * (4 * zz - xx - 3 * yy); dz[11] = (REAL)GSR_SH_C3_2 * 8 * y * z;
                dx[12] = (REAL)GSR_SH_C3_3 * -6 * x * z;        dy[12] = (REAL)GSR_SH_C3_3 * -6 * y * z; dz[12] = (REAL)GSR_SH_C3_3 * (6 * zz - 3 * xx - 3 * yy);
                dx[13] = (REAL)GSR_SH_C3_4 * (4 * zz - 3 * xx - yy); dy[13] = (REAL)GSR_SH_C3_4 * -2 * xy; dz[13] = (REAL)GSR_SH_C3_4 * 8 * x * z;
                dx[14] = (REAL)GSR_SH_C3_5 * 2 * x * z;         dy[14] = (REAL)GSR_SH_C3_5 * -2 * y * z; dz[14] = (REAL)GSR_SH_C3_5 * (xx - yy);
                dx[15] = (REAL)GSR_SH_C3_6 * (3 * xx - 3 * yy); dy[15] = (REAL)GSR_SH_C3_6 * -6 * xy;
            }
        }
    }
}

/* ---- A.1-A.6: per-Gaussian preprocess. */
static void FN(preprocess)(FN(ctx) *c)
{
    const int P = c->P, K = (c->D + 1) * (c->D + 1);
    const REAL *V = c->view, *PV = c->proj;
    for (int i = 0; i < P; ++i) {
        c->radii[i] = 0;
        c->tiles_touched[i] = 0;
        const REAL *p = c->means3D + 3 * i;
        /* A.0 row-vector convention; twin utils/graphics_utils.py:22-29 */
        REAL tvx = p[0] * V[0] + p[1] * V[4] + p[2] * V[8] + V[12];
        REAL tvy = p[0] * V[1] + p[1] * V[5] + p[2] * V[9] + V[13];
        REAL tvz = p[0] * V[2] + p[1] * V[6] + p[2] * V[10] + V[14];
        if (tvz <= (REAL)GSR_NEAR_CUT) continue;                                    /* A.1 */
        REAL hx = p[0] * PV[0] + p[1] * PV[4] + p[2] * PV[8] + PV[12];
        REAL hy = p[0] * PV[1] + p[1] * PV[5] + p[2] * PV[9] + PV[13];
        REAL hw = p[0] * PV[3] + p[1] * PV[7] + p[2] * PV[11] + PV[15];
        REAL pw = 1 / (hw + (REAL)GSR_HOM_EPS);                                      /* A.2 */
        REAL ndcx = hx * pw, ndcy = hy * pw;

        REAL cov[6];
        if (c->has_cov_precomp) memcpy(cov, c->cov3D_precomp + 6 * i, sizeof cov);
        else FN(cov3d)(c->scales + 3 * i, c->scale_modifier, c->rotations + 4 * i, cov);
        memcpy(c->cov3D + 6 * i, cov, sizeof cov);

        /* A.4 EWA projection */
        REAL limx = (REAL)GSR_FOV_CLAMP * c->tanfovx, limy = (REAL)GSR_FOV_CLAMP * c->tanfovy;
        REAL txtz = tvx / tvz, tytz = tvy / tvz;
        REAL tx = (txtz < -limx ? -limx : (txtz > limx ? limx : txtz)) * tvz;
        REAL ty = (tytz < -limy ? -limy : (tytz > limy ? limy : tytz)) * tvz;
        REAL tz = tvz;
        REAL J00 = c->focal_x / tz, J02 = -(c->focal_x * tx) / (tz * tz);
        REAL J11 = c->focal_y / tz, J12 = -(c->focal_y * ty) / (tz * tz);
        /* Wm = R_w2c, Wm[i][j] = V[j][i] */
        REAL W00 = V[0], W01 = V[4], W02 = V[8];
        REAL W10 = V[1], W11 = V[5], W12 = V[9];
        REAL W20 = V[2], W21 = V[6], W22 = V[10];
        REAL T00 = J00 * W00 + J02 * W20, T01 = J00 * W01 + J02 * W21, T02 = J00 * W02 + J02 * W22;
        REAL T10 = J11 * W10 + J12 * W20, T11 = J11 * W11 + J12 * W21, T12 = J11 * W12 + J12 * W22;
        REAL S0x = cov[0] * T00 + cov[1] * T01 + cov[2] * T02;   /* (Sigma T0^T) */
        REAL S0y = cov[1] * T00 + cov[3] * T01 + cov[4] * T02;
        REAL S0z = cov[2] * T00 + cov[4] * T01 + cov[5] * T02;
        REAL S1x = cov[0] * T10 + cov[1] * T11 + cov[2] * T12;
        REAL S1y = cov[1] * T10 + cov[3] * T11 + cov[4] * T12;
        REAL S1z = cov[2] * T10 + cov[4] * T11 + cov[5] * T12;
        REAL a = T00 * S0x + T01 * S0y + T02 * S0z + (REAL)GSR_COV2D_DILATE;
        REAL b = T00 * S1x + T01 * S1y + T02 * S1z;
        REAL cc = T10 * S1x + T11 * S1y + T12 * S1z + (REAL)GSR_COV2D_DILATE;
        REAL det = a * cc - b * b;
        if (det == 0) continue;
        REAL det_inv = 1 / det;
        REAL cA = cc * det_inv, cB = -b * det_inv, cC = a * det_inv;
        REAL mid = (REAL)0.5 * (a + cc);
        REAL disc = mid * mid - det;
        if (disc < (REAL)GSR_LAMBDA_FLOOR) disc = (REAL)GSR_LAMBDA_FLOOR;
        REAL l1 = mid + R_SQRT(disc), l2 = mid - R_SQRT(disc);
        REAL lmax = l1 > l2 ? l1 : l2;
        REAL my_radius = R_CEIL((REAL)GSR_RADIUS_SIGMAS * R_SQRT(lmax));
        REAL px = ((ndcx + 1) * c->W - 1) * (REAL)0.5;                               /* A.2 */
        REAL py = ((ndcy + 1) * c->H - 1) * (REAL)0.5;
        /* A.5 tile rect; int() truncates toward zero */
        int rx0 = (int)((px - my_radius) / GSR_TILE), ry0 = (int)((py - my_radius) / GSR_TILE);
        int rx1 = (int)((px + my_radius + GSR_TILE - 1) / GSR_TILE), ry1 = (int)((py + my_radius + GSR_TILE - 1) / GSR_TILE);
        rx0 = rx0 < 0 ? 0 : (rx0 > c->Gx ? c->Gx : rx0); rx1 = rx1 < 0 ? 0 : (rx1 > c->Gx ? c->Gx : rx1);
        ry0 = ry0 < 0 ? 0 : (ry0 > c->Gy ? c->Gy : ry0); ry1 = ry1 < 0 ? 0 : (ry1 > c->Gy ? c->Gy : ry1);
        if ((rx1 - rx0) * (ry1 - ry0) == 0) continue;

        /* A.6 colour */
        REAL rgb[3]; uint8_t cl[3] = {0, 0, 0};
        if (c->has_colors_precomp) {
            rgb[0] = c->colors_precomp[3 * i]; rgb[1] = c->colors_precomp[3 * i + 1]; rgb[2] = c->colors_precomp[3 * i + 2];
        } else {
            REAL dx = p[0] - c->campos[0], dy = p[1] - c->campos[1], dz = p[2] - c->campos[2];
            REAL inv = 1 / R_SQRT(dx * dx + dy * dy + dz * dz);
            dx *= inv; dy *= inv; dz *= inv;
            REAL bas[16];
            FN(sh_basis)(c->D, dx, dy, dz, bas);
            const REAL *sh = c->shs + (size_t)i * c->M * 3;
            for (int ch = 0; ch < 3; ++ch) {
                REAL acc = 0;
                for (int k = 0; k < K; ++k) acc += bas[k] * sh[3 * k + ch];
                acc += (REAL)GSR_SH_OFFSET;
                cl[ch] = acc < 0;
                rgb[ch] = acc < 0 ? 0 : acc;
            }
        }
        c->radii[i] = (int32_t)my_radius;
        c->depth[i] = tvz;
        c->xy[2 * i] = px; c->xy[2 * i + 1] = py;
        c->conic_op[4 * i] = cA; c->conic_op[4 * i + 1] = cB; c->conic_op[4 * i + 2] = cC;
        c->conic_op[4 * i + 3] = c->opacities[i];
        c->kappa[i] = (a * cc + b * b) * R_FABS(det_inv);
        {   /* the sort key is the binary32 view depth: three legitimate binary32 evaluations of it */
            const float p0 = (float)p[0], p1 = (float)p[1], p2 = (float)p[2];
            const float v2 = (float)V[2], v6 = (float)V[6], v10 = (float)V[10], v14 = (float)V[14];
            volatile float m0 = p0 * v2, m1 = p1 * v6, m2 = p2 * v10;
            volatile float s01 = m0 + m1;
            volatile float s012 = s01 + m2;
            const float dA = s012 + v14;
            const float dB = fmaf(p2, v10, fmaf(p1, v6, m0)) + v14;
            const float dC = (float)((double)p0 * v2 + (double)p1 * v6 + (double)p2 * v10 + (double)v14);
            float lo_ = dA < dB ? dA : dB, hi_ = dA < dB ? dB : dA;
            c->depth_lo[i] = lo_ < dC ? lo_ : dC; c->depth_hi[i] = hi_ > dC ? hi_ : dC;
        }
        memcpy(c->rgb + 3 * i, rgb, sizeof rgb);
        memcpy(c->clamped + 3 * i, cl, 3);
        /* slab clip in tile rows (multi-GPU): radii stay those of the full image */
        if (ry0 < c->ty0) ry0 = c->ty0;
        if (ry1 > c->ty1) ry1 = c->ty1;
        if (ry1 < ry0) ry1 = ry0;
        c->rect[4 * i] = rx0; c->rect[4 * i + 1] = ry0; c->rect[4 * i + 2] = rx1; c->rect[4 * i + 3] = ry1;
        c->tiles_touched[i] = (uint32_t)((rx1 - rx0) * (ry1 - ry0));
    }
}

/* stable merge sort of (key, val) pairs by key (A.7 requires stability). */
static void FN(msort)(uint64_t *k, uint32_t *v, uint64_t *tk, uint32_t *tv, int64_t lo, int64_t hi)
{
    if (hi - lo < 2) return;
    if (hi - lo <= 16) {                       /* insertion sort, stable */
        for (int64_t i = lo + 1; i < hi; ++i) {
            uint64_t kk = k[i]; uint32_t vv = v[i]; int64_t j = i - 1;
            while (j >= lo && k[j] > kk) { k[j + 1] = k[j]; v[j + 1] = v[j]; --j; }
            k[j + 1] = kk; v[j + 1] = vv;
        }
        return;
    }
    int64_t mid = lo + (hi - lo) / 2;
    FN(msort)(k, v, tk, tv, lo, mid);
    FN(msort)(k, v, tk, tv, mid, hi);
    if (k[mid - 1] <= k[mid]) return;
    int64_t i = lo, j = mid, o = lo;
    while (i < mid && j < hi) {
        if (k[j] < k[i]) { tk[o] = k[j]; tv[o++] = v[j++]; }
        else             { tk[o] = k[i]; tv[o++] = v[i++]; }
    }
    while (i < mid) { tk[o] = k[i]; tv[o++] = v[i++]; }
    while (j < hi)  { tk[o] = k[j]; tv[o++] = v[j++]; }
    memcpy(k + lo, tk + lo, (size_t)(hi - lo) * sizeof *k);
    memcpy(v + lo, tv + lo, (size_t)(hi - lo) * sizeof *v);
}

/* ---- A.7: duplicate with keys, stable sort, tile ranges.  The sort key's depth half is the
 * IEEE-754 binary32 pattern of the depth in BOTH instantiations (the fp64 build rounds the depth
 * to binary32 for the key only), so the order matches what a 64-bit (tile<<32 | depth) key gives. */
static void FN(binning)(FN(ctx) *c)
{
    int64_t R = 0;
    for (int i = 0; i < c->P; ++i) R += c->tiles_touched[i];
    c->R = R;
    c->keys = (uint64_t *)malloc((size_t)(R ? R : 1) * sizeof(uint64_t));
    c->vals = (uint32_t *)malloc((size_t)(R ? R : 1) * sizeof(uint32_t));
    int64_t o = 0;
    for (int i = 0; i < c->P; ++i) {
        if (c->radii[i] <= 0 || c->tiles_touched[i] == 0) continue;
        float df = (float)c->depth[i];
        uint32_t dbits; memcpy(&dbits, &df, 4);
        const int32_t *r = c->rect + 4 * i;
        for (int y = r[1]; y < r[3]; ++y)
            for (int x = r[0]; x < r[2]; ++x) {
                c->keys[o] = ((uint64_t)(uint32_t)(y * c->Gx + x) << 32) | dbits;
                c->vals[o] = (uint32_t)i;
                ++o;
            }
    }
    uint64_t *tk = (uint64_t *)malloc((size_t)(R ? R : 1) * sizeof(uint64_t));
    uint32_t *tv = (uint32_t *)malloc((size_t)(R ? R : 1) * sizeof(uint32_t));
    FN(msort)(c->keys, c->vals, tk, tv, 0, R);
    free(tk); free(tv);
    int Tn = c->Gx * c->Gy;
    for (int t = 0; t < 2 * Tn; ++t) c->ranges[t] = 0;
    for (int64_t i = 0; i < R; ++i) {
        uint32_t t = (uint32_t)(c->keys[i] >> 32);
        if (i == 0 || (uint32_t)(c->keys[i - 1] >> 32) != t) c->ranges[2 * t] = i;
        if (i == R - 1 || (uint32_t)(c->keys[i + 1] >> 32) != t) c->ranges[2 * t + 1] = i + 1;
    }
}

/* ---- A.8: forward blend of one tile (all its pixels), front to back. */
static void FN(render_tile)(FN(ctx) *c, int tx, int ty, int64_t *pairs_out)
{
    const int W = c->W, H = c->H;
    const int64_t start = c->ranges[2 * (ty * c->Gx + tx)], end = c->ranges[2 * (ty * c->Gx + tx) + 1];
    const REAL eps = c->fragile_eps;
    int64_t pairs = 0;
    for (int py = ty * GSR_TILE; py < (ty + 1) * GSR_TILE && py < H; ++py)
        for (int px = tx * GSR_TILE; px < (tx + 1) * GSR_TILE && px < W; ++px) {
            REAL T = 1, C[3] = {0, 0, 0};
            int32_t contributor = 0, last = 0;
            uint8_t fragile = c->fragile_px[(size_t)py * W + px];      /* tile membership of a splat that reaches it is uncertain */
            REAL terr = 0;
            float prev_lo = 0, prev_hi = 0;
            int prev_reaches = 0;
            for (int64_t j = start; j < end; ++j) {
                ++contributor; ++pairs;
                uint32_t g = c->vals[j];
                REAL dx = c->xy[2 * g] - (REAL)px, dy = c->xy[2 * g + 1] - (REAL)py;
                const REAL *co = c->conic_op + 4 * g;
                REAL qa = (REAL)0.5 * co[0] * dx * dx, qc = (REAL)0.5 * co[2] * dy * dy, qb = co[1] * dx * dy;
                REAL power = -(REAL)0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                REAL mag = R_FABS(qa) + R_FABS(qc) + R_FABS(qb);
                /* how far a binary32 evaluation of `power` may sit from this one: rounding of the 2D covariance amplified
                 * through its determinant (kappa), and the splat's pixel position -- xy ~ W/2 carries an absolute error of a
                 * few ulps of the image size, which dx = xy - pix inherits in full -- times d power / d xy */
                const REAL band = eps * c->kappa[g], pband = c->fragile_pos * (R_FABS(co[0] * dx + co[1] * dy) + R_FABS(co[2] * dy + co[1] * dx));
                if (R_FABS(power) <= band * mag + pband && mag > 0) fragile = 1;   /* sign of power uncertain */
                if (power > 0) continue;
                REAL alpha = co[3] * R_EXP(power);
                if (alpha > (REAL)GSR_ALPHA_MAX) alpha = (REAL)GSR_ALPHA_MAX;
                if (R_FABS(alpha * 255 - 1) <= band * (1 + mag) + pband) fragile = 1;
                /* the sort key is the BINARY32 view depth: two list neighbours whose depths sit within a few ulps of each other
                 * may come in either order in a binary32 evaluation; the blend of two splats does not commute (the colour moves
                 * by alpha_1 alpha_2 (c_1 - c_2) T, n_contrib by one) when both reach this pixel */
                {
                    const float dlo = c->depth_lo[g], dhi = c->depth_hi[g];
                    const int reaches = alpha * 255 - 1 >= -(band * (1 + mag) + pband);
                    /* (depths that every evaluation finds bit-identical are ordered by index everywhere: A.7's sort is stable) */
                    if (reaches && prev_reaches && prev_hi >= dlo && !(prev_lo == prev_hi && dlo == dhi && prev_lo == dlo)) fragile = 1;
                    /* a neighbour that does not reach the pixel hides nothing: keep the last one that does */
                    if (reaches) { prev_lo = dlo; prev_hi = dhi > prev_hi || !prev_reaches ? dhi : prev_hi; prev_reaches = 1; }
                }
                if (alpha < (REAL)GSR_ALPHA_MIN) continue;
                REAL test_T = T * (1 - alpha);
                /* first-order bound on the relative error of T = prod(1 - alpha_i) in binary32: every factor inherits its
                 * alpha's uncertainty (the bands above) times alpha / (1 - alpha); 64 eps covers the roundings of the product */
                /* (an alpha that sits on its 0.99 clamp with room to spare does not move with its inputs) */
                const REAL rho = band * (1 + mag) + pband;
                const REAL da = co[3] * R_EXP(power) * (1 - rho) > (REAL)GSR_ALPHA_MAX ? 0 : rho * alpha / (1 - alpha);
                if (R_FABS(test_T - (REAL)GSR_T_CUTOFF) <= (64 * eps + terr + da) * (REAL)GSR_T_CUTOFF) fragile = 1;
                if (test_T < (REAL)GSR_T_CUTOFF) break;
                terr += da;
                const REAL *col = c->rgb + 3 * g;
                C[0] += col[0] * alpha * T; C[1] += col[1] * alpha * T; C[2] += col[2] * alpha * T;
                T = test_T;
                last = contributor;
            }
            size_t pix = (size_t)py * W + px;
            c->final_T[pix] = T;
            c->n_contrib[pix] = last;
            c->fragile_px[pix] = fragile;
            for (int ch = 0; ch < 3; ++ch) c->color[(size_t)ch * H * W + pix] = C[ch] + T * c->bg[ch];
            if (fragile) {
                /* Only Gaussians that can REACH this pixel are affected by a flipped decision: those whose
                 * alpha here is >= 1/255 (within the band), anywhere in the tile's list -- a flipped stop
                 * decision lets later splats in.  A race on the byte store is benign (all writers store 1). */
                for (int64_t j = start; j < end; ++j) {
                    uint32_t g = c->vals[j];
                    REAL dx = c->xy[2 * g] - (REAL)px, dy = c->xy[2 * g + 1] - (REAL)py;
                    const REAL *co = c->conic_op + 4 * g;
                    REAL qa = (REAL)0.5 * co[0] * dx * dx, qc = (REAL)0.5 * co[2] * dy * dy, qb = co[1] * dx * dy;
                    REAL power = -(REAL)0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    REAL mag = R_FABS(qa) + R_FABS(qc) + R_FABS(qb);
                    const REAL band = eps * c->kappa[g], pband = c->fragile_pos * (R_FABS(co[0] * dx + co[1] * dy) + R_FABS(co[2] * dy + co[1] * dx));
                    if (power > band * mag + pband) continue;
                    REAL alpha = co[3] * R_EXP(power > 0 ? 0 : power);
                    if (alpha * 255 - 1 < -(band * (1 + mag) + pband)) continue;
                    c->fragile_g[g] = 1;
                }
            }
        }
    *pairs_out = pairs;
}

/* ---- fragile tile membership.  A.5 puts a Gaussian into tile t iff floor((p - r) / 16) <= t < floor((p + r + 15) / 16): where
 * (p -+ r) sits within the position uncertainty of a multiple of 16 px, a binary32 evaluation may put the Gaussian into one
 * tile column / row more or less.  The 3-sigma box cuts INSIDE the alpha >= 1/255 ellipse for opacities above ~0.35, so the
 * pixels of such a tile that the splat reaches are decided by that rounding: they are marked fragile (and the Gaussian with
 * them) before the blend runs. */
static void FN(mark_rect_fragile)(FN(ctx) *c)
{
    const REAL d = c->fragile_pos, eps = c->fragile_eps;
    if (!(d > 0)) return;
    for (int i = 0; i < c->P; ++i) {
        if (c->radii[i] <= 0) continue;
        const REAL px = c->xy[2 * i], py = c->xy[2 * i + 1], r = (REAL)c->radii[i];
        const REAL lo[2] = {px - r, py - r}, hi[2] = {px + r + (GSR_TILE - 1), py + r + (GSR_TILE - 1)};
        const int G[2] = {c->Gx, c->Gy};
        int in0[2], in1[2], out0[2], out1[2], any = 0;
        for (int a = 0; a < 2; ++a) {
            /* inner rect: surely inside; outer rect: possibly inside */
            int i0 = (int)((lo[a] + d) / GSR_TILE), o0 = (int)((lo[a] - d) / GSR_TILE);
            int i1 = (int)((hi[a] - d) / GSR_TILE), o1 = (int)((hi[a] + d) / GSR_TILE);
            if (lo[a] - d < 0 && lo[a] + d >= 0) o0 = 0;      /* truncation toward zero: (-1, 0] and [0, 1) share tile 0 */
#define GSO_CLAMP(v, m) ((v) < 0 ? 0 : ((v) > (m) ? (m) : (v)))
            in0[a] = GSO_CLAMP(i0, G[a]); out0[a] = GSO_CLAMP(o0, G[a]); in1[a] = GSO_CLAMP(i1, G[a]); out1[a] = GSO_CLAMP(o1, G[a]);
#undef GSO_CLAMP
            if (in0[a] != out0[a] || in1[a] != out1[a]) any = 1;
        }
        if (!any) continue;
        if (out0[1] < c->ty0) out0[1] = c->ty0;
        if (out1[1] > c->ty1) out1[1] = c->ty1;
        const REAL *co = c->conic_op + 4 * i;
        for (int ty = out0[1]; ty < out1[1]; ++ty)
            for (int tx = out0[0]; tx < out1[0]; ++tx) {
                if (tx >= in0[0] && tx < in1[0] && ty >= in0[1] && ty < in1[1]) continue;      /* membership certain */
                for (int y = ty * GSR_TILE; y < (ty + 1) * GSR_TILE && y < c->H; ++y)
                    for (int x = tx * GSR_TILE; x < (tx + 1) * GSR_TILE && x < c->W; ++x) {
                        REAL dx = px - (REAL)x, dy = py - (REAL)y;
                        REAL qa = (REAL)0.5 * co[0] * dx * dx, qc = (REAL)0.5 * co[2] * dy * dy, qb = co[1] * dx * dy;
                        REAL power = -(qa + qc) - qb;
                        REAL mag = R_FABS(qa) + R_FABS(qc) + R_FABS(qb);
                        const REAL band = eps * c->kappa[i], pband = d * (R_FABS(co[0] * dx + co[1] * dy) + R_FABS(co[2] * dy + co[1] * dx));
                        if (power > band * mag + pband) continue;
                        REAL alpha = co[3] * R_EXP(power > 0 ? 0 : power);
                        if (alpha * 255 - 1 < -(band * (1 + mag) + pband)) continue;
                        c->fragile_px[(size_t)y * c->W + x] = 1;
                        c->fragile_g[i] = 1;
                    }
            }
    }
}

FN(ctx) *FN(forward)(int P, int D, int M, int W, int H, double tanfovx, double tanfovy, double scale_modifier,
                     int tile_row_begin, int tile_row_end, double fragile_eps, double fragile_pos, int parallel,
                     const REAL *bg, const REAL *view, const REAL *proj, const REAL *campos,
                     const REAL *means3D, const REAL *shs, const REAL *colors_precomp, const REAL *opacities,
                     const REAL *scales, const REAL *rotations, const REAL *cov3D_precomp)
{
    FN(ctx) *c = (FN(ctx) *)calloc(1, sizeof *c);
    c->P = P; c->D = D; c->M = M; c->W = W; c->H = H;
    c->Gx = (W + GSR_TILE - 1) / GSR_TILE; c->Gy = (H + GSR_TILE - 1) / GSR_TILE;
    c->ty0 = tile_row_begin < 0 ? 0 : tile_row_begin;
    c->ty1 = (tile_row_end < 0 || tile_row_end > c->Gy) ? c->Gy : tile_row_end;
    c->tanfovx = (REAL)tanfovx; c->tanfovy = (REAL)tanfovy; c->scale_modifier = (REAL)scale_modifier;
    c->focal_x = (REAL)W / (2 * c->tanfovx); c->focal_y = (REAL)H / (2 * c->tanfovy);
    c->fragile_eps = (REAL)fragile_eps; c->fragile_pos = (REAL)fragile_pos;
    memcpy(c->bg, bg, 3 * sizeof(REAL)); memcpy(c->view, view, 16 * sizeof(REAL));
    memcpy(c->proj, proj, 16 * sizeof(REAL)); memcpy(c->campos, campos, 3 * sizeof(REAL));
    c->has_cov_precomp = cov3D_precomp != NULL; c->has_colors_precomp = colors_precomp != NULL;
    c->means3D = FN(dup)(means3D, (size_t)3 * P);
    c->shs = FN(dup)(shs, (size_t)3 * M * P);
    c->colors_precomp = FN(dup)(colors_precomp, (size_t)3 * P);
    c->opacities = FN(dup)(opacities, (size_t)P);
    c->scales = FN(dup)(scales, (size_t)3 * P);
    c->rotations = FN(dup)(rotations, (size_t)4 * P);
    c->cov3D_precomp = FN(dup)(cov3D_precomp, (size_t)6 * P);
    size_t Pn = P ? P : 1, N = (size_t)W * H;
    c->radii = (int32_t *)calloc(Pn, 4);
    c->xy = (REAL *)calloc(2 * Pn, sizeof(REAL)); c->depth = (REAL *)calloc(Pn, sizeof(REAL));
    c->cov3D = (REAL *)calloc(6 * Pn, sizeof(REAL)); c->conic_op = (REAL *)calloc(4 * Pn, sizeof(REAL));
    c->rgb = (REAL *)calloc(3 * Pn, sizeof(REAL)); c->clamped = (uint8_t *)calloc(3 * Pn, 1);
    c->kappa = (REAL *)calloc(Pn, sizeof(REAL));
    c->depth_lo = (float *)calloc(Pn, sizeof(float)); c->depth_hi = (float *)calloc(Pn, sizeof(float));
    c->rect = (int32_t *)calloc(4 * Pn, 4); c->tiles_touched = (uint32_t *)calloc(Pn, 4);
    c->ranges = (int64_t *)calloc((size_t)2 * c->Gx * c->Gy, sizeof(int64_t));
    c->color = (REAL *)calloc(3 * N, sizeof(REAL)); c->final_T = (REAL *)calloc(N, sizeof(REAL));
    c->n_contrib = (int32_t *)calloc(N, 4); c->fragile_px = (uint8_t *)calloc(N, 1);
    c->fragile_g = (uint8_t *)calloc(Pn, 1);

    FN(preprocess)(c);
    FN(binning)(c);
    FN(mark_rect_fragile)(c);
    int64_t total_pairs = 0;
    const int Tn_slab = (c->ty1 - c->ty0) * c->Gx;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : total_pairs) if (parallel)
    for (int t = 0; t < Tn_slab; ++t) {
        int64_t pr = 0;
        FN(render_tile)(c, t % c->Gx, c->ty0 + t / c->Gx, &pr);
        total_pairs += pr;
    }
    c->n_pairs = total_pairs;
    return c;
}

/* ---- A.9: backward blend of one tile; accumulates per-splat partial sums over the tile's pixels
 * (raster order) into `acc` (9 REALs per list entry: dmean2D.xy, gA, gB, gC, dopacity, drgb[3]). */
static void FN(render_tile_bwd)(const FN(ctx) *c, int tx, int ty, const REAL *dL_dpix, REAL *acc)
{
    const int W = c->W, H = c->H;
    const int64_t start = c->ranges[2 * (ty * c->Gx + tx)];
    const REAL ddelx_dx = (REAL)0.5 * W, ddely_dy = (REAL)0.5 * H;
    for (int py = ty * GSR_TILE; py < (ty + 1) * GSR_TILE && py < H; ++py)
        for (int px = tx * GSR_TILE; px < (tx + 1) * GSR_TILE && px < W; ++px) {
            size_t pix = (size_t)py * W + px;
            const REAL T_final = c->final_T[pix];
            REAL T = T_final;
            const int32_t last = c->n_contrib[pix];
            REAL dpix[3], accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0;
            REAL bg_dot = 0;
            for (int ch = 0; ch < 3; ++ch) { dpix[ch] = dL_dpix[(size_t)ch * H * W + pix]; bg_dot += c->bg[ch] * dpix[ch]; }
            for (int64_t j = start + last - 1; j >= start; --j) {
                uint32_t g = c->vals[j];
                REAL dx = c->xy[2 * g] - (REAL)px, dy = c->xy[2 * g + 1] - (REAL)py;
                const REAL *co = c->conic_op + 4 * g;
                REAL power = -(REAL)0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0) continue;
                REAL G = R_EXP(power);
                REAL alpha = co[3] * G;
                if (alpha > (REAL)GSR_ALPHA_MAX) alpha = (REAL)GSR_ALPHA_MAX;
                if (alpha < (REAL)GSR_ALPHA_MIN) continue;
                T = T / (1 - alpha);
                REAL dchannel_dcolor = alpha * T;
                REAL dL_dalpha = 0;
                const REAL *col = c->rgb + 3 * g;
                REAL *a = acc + 9 * (j - start);
                for (int ch = 0; ch < 3; ++ch) {
                    accum_rec[ch] = last_alpha * last_color[ch] + (1 - last_alpha) * accum_rec[ch];
                    last_color[ch] = col[ch];
                    dL_dalpha += (col[ch] - accum_rec[ch]) * dpix[ch];
                    a[6 + ch] += dchannel_dcolor * dpix[ch];
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1 - alpha)) * bg_dot;
                REAL dL_dG = co[3] * dL_dalpha;
                REAL gdx = G * dx, gdy = G * dy;
                REAL dG_ddelx = -gdx * co[0] - gdy * co[1];
                REAL dG_ddely = -gdy * co[2] - gdx * co[1];
                a[0] += dL_dG * dG_ddelx * ddelx_dx;
                a[1] += dL_dG * dG_ddely * ddely_dy;
                a[2] += -(REAL)0.5 * gdx * dx * dL_dG;
                a[3] += -(REAL)0.5 * gdx * dy * dL_dG;
                a[4] += -(REAL)0.5 * gdy * dy * dL_dG;
                a[5] += G * dL_dalpha;
            }
        }
}

/* ---- A.10: per-Gaussian backward from the screen-space gradients
 * sg[9] = (dmean2D.x, dmean2D.y, gA, gB, gC, dopacity, drgb[3]).  Gaussians in [g0, g1). */
static void FN(geom_bwd)(const FN(ctx) *c, const REAL *screen, int g0, int g1,
                         REAL *dmeans3D, REAL *dmeans2D, REAL *dsh, REAL *dcolors, REAL *dopac,
                         REAL *dscales, REAL *drot, REAL *dcov3D)
{
    const int K = (c->D + 1) * (c->D + 1), M = c->M;
    const REAL *V = c->view, *PV = c->proj;
    for (int i = g0; i < g1; ++i) {
        if (c->radii[i] <= 0) continue;
        const REAL *sg = screen + 9 * i;
        const REAL *p = c->means3D + 3 * i;
        const REAL *cov = c->cov3D + 6 * i;
        dmeans2D[3 * i] = sg[0]; dmeans2D[3 * i + 1] = sg[1]; dmeans2D[3 * i + 2] = 0;
        dopac[i] = sg[5];
        REAL dmean[3] = {0, 0, 0};

        /* ---- conic -> cov2D -> (cov3D, t) */
        REAL tvx = p[0] * V[0] + p[1] * V[4] + p[2] * V[8] + V[12];
        REAL tvy = p[0] * V[1] + p[1] * V[5] + p[2] * V[9] + V[13];
        REAL tvz = p[0] * V[2] + p[1] * V[6] + p[2] * V[10] + V[14];
        REAL limx = (REAL)GSR_FOV_CLAMP * c->tanfovx, limy = (REAL)GSR_FOV_CLAMP * c->tanfovy;
        REAL txtz = tvx / tvz, tytz = tvy / tvz;
        REAL tx = (txtz < -limx ? -limx : (txtz > limx ? limx : txtz)) * tvz;
        REAL ty = (tytz < -limy ? -limy : (tytz > limy ? limy : tytz)) * tvz;
        REAL tz = tvz;
        REAL x_grad_mul = (txtz < -limx || txtz > limx) ? 0 : 1;
        REAL y_grad_mul = (tytz < -limy || tytz > limy) ? 0 : 1;
        REAL J00 = c->focal_x / tz, J02 = -(c->focal_x * tx) / (tz * tz);
        REAL J11 = c->focal_y / tz, J12 = -(c->focal_y * ty) / (tz * tz);
        REAL W00 = V[0], W01 = V[4], W02 = V[8], W10 = V[1], W11 = V[5], W12 = V[9], W20 = V[2], W21 = V[6], W22 = V[10];
        REAL T00 = J00 * W00 + J02 * W20, T01 = J00 * W01 + J02 * W21, T02 = J00 * W02 + J02 * W22;
        REAL T10 = J11 * W10 + J12 * W20, T11 = J11 * W11 + J12 * W21, T12 = J11 * W12 + J12 * W22;
        REAL S0x = cov[0] * T00 + cov[1] * T01 + cov[2] * T02, S0y = cov[1] * T00 + cov[3] * T01 + cov[4] * T02, S0z = cov[2] * T00 + cov[4] * T01 + cov[5] * T02;
        REAL S1x = cov[0] * T10 + cov[1] * T11 + cov[2] * T12, S1y = cov[1] * T10 + cov[3] * T11 + cov[4] * T12, S1z = cov[2] * T10 + cov[4] * T11 + cov[5] * T12;
        REAL a = T00 * S0x + T01 * S0y + T02 * S0z + (REAL)GSR_COV2D_DILATE;
        REAL b = T00 * S1x + T01 * S1y + T02 * S1z;
        REAL cc = T10 * S1x + T11 * S1y + T12 * S1z + (REAL)GSR_COV2D_DILATE;
        REAL den = a * cc - b * b;
        REAL k2 = 1 / (den * den + (REAL)GSR_CONIC_BWD_EPS);
        REAL gA = sg[2], gB = sg[3], gC = sg[4];
        REAL dL_da = 0, dL_db = 0, dL_dc = 0;
        REAL dcov[6] = {0, 0, 0, 0, 0, 0};
        {
            dL_da = k2 * (-cc * cc * gA + 2 * b * cc * gB + (den - a * cc) * gC);
            dL_dc = k2 * (-a * a * gC + 2 * a * b * gB + (den - a * cc) * gA);
            dL_db = k2 * 2 * (b * cc * gA - (den + 2 * b * b) * gB + a * b * gC);
            dcov[0] = T00 * T00 * dL_da + T00 * T10 * dL_db + T10 * T10 * dL_dc;
            dcov[3] = T01 * T01 * dL_da + T01 * T11 * dL_db + T11 * T11 * dL_dc;
            dcov[5] = T02 * T02 * dL_da + T02 * T12 * dL_db + T12 * T12 * dL_dc;
            dcov[1] = 2 * T00 * T01 * dL_da + (T00 * T11 + T01 * T10) * dL_db + 2 * T10 * T11 * dL_dc;
            dcov[2] = 2 * T00 * T02 * dL_da + (T00 * T12 + T02 * T10) * dL_db + 2 * T10 * T12 * dL_dc;
            dcov[4] = 2 * T02 * T01 * dL_da + (T01 * T12 + T02 * T11) * dL_db + 2 * T11 * T12 * dL_dc;
        }
        /* dL/dT = 2 G2 T Sigma, with (T Sigma)_0 = S0, (T Sigma)_1 = S1 */
        REAL dT00 = 2 * S0x * dL_da + S1x * dL_db, dT01 = 2 * S0y * dL_da + S1y * dL_db, dT02 = 2 * S0z * dL_da + S1z * dL_db;
        REAL dT10 = 2 * S1x * dL_dc + S0x * dL_db, dT11 = 2 * S1y * dL_dc + S0y * dL_db, dT12 = 2 * S1z * dL_dc + S0z * dL_db;
        /* dL/dJ = dL/dT Wm^T (only J's non-zeros) */
        REAL dJ00 = W00 * dT00 + W01 * dT01 + W02 * dT02;
        REAL dJ02 = W20 * dT00 + W21 * dT01 + W22 * dT02;
        REAL dJ11 = W10 * dT10 + W11 * dT11 + W12 * dT12;
        REAL dJ12 = W20 * dT10 + W21 * dT11 + W22 * dT12;
        REAL tzi = 1 / tz, tz2 = tzi * tzi, tz3 = tz2 * tzi;
        REAL dtx = x_grad_mul * -c->focal_x * tz2 * dJ02;
        REAL dty = y_grad_mul * -c->focal_y * tz2 * dJ12;
        REAL dtz = -c->focal_x * tz2 * dJ00 - c->focal_y * tz2 * dJ11 + (2 * c->focal_x * tx) * tz3 * dJ02 + (2 * c->focal_y * ty) * tz3 * dJ12;
        /* back through the view transform (assigned) */
        dmean[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
        dmean[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
        dmean[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;

        /* ---- projection path (added) */
        REAL hx = p[0] * PV[0] + p[1] * PV[4] + p[2] * PV[8] + PV[12];
        REAL hy = p[0] * PV[1] + p[1] * PV[5] + p[2] * PV[9] + PV[13];
        REAL hw = p[0] * PV[3] + p[1] * PV[7] + p[2] * PV[11] + PV[15];
        REAL mw = 1 / (hw + (REAL)GSR_HOM_EPS);
        REAL mul1 = hx * mw * mw, mul2 = hy * mw * mw;
        dmean[0] += (PV[0] * mw - PV[3] * mul1) * sg[0] + (PV[1] * mw - PV[3] * mul2) * sg[1];
        dmean[1] += (PV[4] * mw - PV[7] * mul1) * sg[0] + (PV[5] * mw - PV[7] * mul2) * sg[1];
        dmean[2] += (PV[8] * mw - PV[11] * mul1) * sg[0] + (PV[9] * mw - PV[11] * mul2) * sg[1];

        /* ---- colour path */
        if (c->has_colors_precomp) {
            dcolors[3 * i] = sg[6]; dcolors[3 * i + 1] = sg[7]; dcolors[3 * i + 2] = sg[8];
        } else {
            REAL ox = p[0] - c->campos[0], oy = p[1] - c->campos[1], oz = p[2] - c->campos[2];
            REAL len2 = ox * ox + oy * oy + oz * oz, inv = 1 / R_SQRT(len2);
            REAL dxn = ox * inv, dyn = oy * inv, dzn = oz * inv;
            REAL bas[16], bx[16], by[16], bz[16];
            FN(sh_basis)(c->D, dxn, dyn, dzn, bas);
            FN(sh_basis_grad)(c->D, dxn, dyn, dzn, bx, by, bz);
            const REAL *sh = c->shs + (size_t)i * M * 3;
            REAL *dshi = dsh + (size_t)i * M * 3;
            REAL ddir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ++ch) {
                REAL dRGB = c->clamped[3 * i + ch] ? 0 : sg[6 + ch];
                for (int k = 0; k < K; ++k) {
                    dshi[3 * k + ch] = bas[k] * dRGB;
                    ddir[0] += bx[k] * sh[3 * k + ch] * dRGB;
                    ddir[1] += by[k] * sh[3 * k + ch] * dRGB;
                    ddir[2] += bz[k] * sh[3 * k + ch] * dRGB;
                }
            }
            /* through normalize(): (I - n n^T)/|v| */
            REAL dot = dxn * ddir[0] + dyn * ddir[1] + dzn * ddir[2];
            dmean[0] += (ddir[0] - dxn * dot) * inv;
            dmean[1] += (ddir[1] - dyn * dot) * inv;
            dmean[2] += (ddir[2] - dzn * dot) * inv;
        }
        dmeans3D[3 * i] = dmean[0]; dmeans3D[3 * i + 1] = dmean[1]; dmeans3D[3 * i + 2] = dmean[2];

        /* ---- cov3D path */
        if (c->has_cov_precomp) {
            for (int k = 0; k < 6; ++k) dcov3D[6 * i + k] = dcov[k];
        } else {
            REAL Rm[9];
            const REAL *q = c->rotations + 4 * i, *sc = c->scales + 3 * i;
            FN(quat_to_rot)(q, Rm);
            REAL s[3] = {c->scale_modifier * sc[0], c->scale_modifier * sc[1], c->scale_modifier * sc[2]};
            /* full symmetric gradient matrix: off-diagonals halve the stored-parameter gradients */
            REAL G3[9] = {dcov[0], (REAL)0.5 * dcov[1], (REAL)0.5 * dcov[2],
                          (REAL)0.5 * dcov[1], dcov[3], (REAL)0.5 * dcov[4],
                          (REAL)0.5 * dcov[2], (REAL)0.5 * dcov[4], dcov[5]};
            /* Mm[k][j] = s_k R[j][k];  dL/dMm = 2 Mm G3 */
            REAL dM[9];
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < 3; ++j) {
                    REAL accm = 0;
                    for (int l = 0; l < 3; ++l) accm += s[k] * Rm[3 * l + k] * G3[3 * l + j];
                    dM[3 * k + j] = 2 * accm;
                }
            REAL gR[9];     /* dL/dR[j][k] = s_k dM[k][j] */
            for (int k = 0; k < 3; ++k) {
                REAL ds = 0;
                for (int j = 0; j < 3; ++j) { ds += Rm[3 * j + k] * dM[3 * k + j]; gR[3 * j + k] = s[k] * dM[3 * k + j]; }
                dscales[3 * i + k] = ds;        /* scale_modifier factor omitted, as A.10 records */
            }
            REAL r = q[0], x = q[1], y = q[2], z = q[3];
            drot[4 * i + 0] = 2 * (-z * gR[1] + y * gR[2] + z * gR[3] - x * gR[5] - y * gR[6] + x * gR[7]);
            drot[4 * i + 1] = 2 * (y * gR[1] + z * gR[2] + y * gR[3] - 2 * x * gR[4] - r * gR[5] + z * gR[6] + r * gR[7] - 2 * x * gR[8]);
            drot[4 * i + 2] = 2 * (-2 * y * gR[0] + x * gR[1] + r * gR[2] + x * gR[3] + z * gR[5] - r * gR[6] + z * gR[7] - 2 * y * gR[8]);
            drot[4 * i + 3] = 2 * (-2 * z * gR[0] - r * gR[1] + x * gR[2] + r * gR[3] - 2 * z * gR[4] + y * gR[5] + x * gR[6] + y * gR[7]);
        }
    }
}

/* Screen-space stage of the backward (A.9): screen[9*P], zero for Gaussians that touch no tile. */
void FN(backward_screen)(const FN(ctx) *c, const REAL *dL_dpix, int parallel, REAL *screen)
{
    memset(screen, 0, (size_t)9 * (c->P ? c->P : 1) * sizeof(REAL));
    const int Tn_slab = (c->ty1 - c->ty0) * c->Gx;
#pragma omp parallel for schedule(dynamic, 4) if (parallel)
    for (int t = 0; t < Tn_slab; ++t) {
        int tx = t % c->Gx, ty = c->ty0 + t / c->Gx;
        int64_t start = c->ranges[2 * (ty * c->Gx + tx)], end = c->ranges[2 * (ty * c->Gx + tx) + 1];
        if (end <= start) continue;
        REAL *acc = (REAL *)calloc((size_t)9 * (end - start), sizeof(REAL));
        FN(render_tile_bwd)(c, tx, ty, dL_dpix, acc);
#pragma omp critical
        for (int64_t j = start; j < end; ++j) {
            REAL *dst = screen + 9 * (size_t)c->vals[j];
            for (int k = 0; k < 9; ++k) dst[k] += acc[9 * (j - start) + k];
        }
        free(acc);
    }
}

/* Geometry stage of the backward (A.10) on Gaussians [g0, g1); outputs must be zero-initialised. */
void FN(backward_geom)(const FN(ctx) *c, const REAL *screen, int g0, int g1,
                       REAL *dmeans3D, REAL *dmeans2D, REAL *dsh, REAL *dcolors, REAL *dopac,
                       REAL *dscales, REAL *drot, REAL *dcov3D)
{
    FN(geom_bwd)(c, screen, g0, g1, dmeans3D, dmeans2D, dsh, dcolors, dopac, dscales, drot, dcov3D);
}

/* ---- accessors (ctypes copies state out through these) */
int64_t FN(num_rendered)(const FN(ctx) *c) { return c->R; }
int64_t FN(num_pairs)(const FN(ctx) *c) { return c->n_pairs; }
void FN(get_image)(const FN(ctx) *c, REAL *color, REAL *final_T, int32_t *n_contrib, uint8_t *fragile_px)
{
    size_t N = (size_t)c->W * c->H;
    if (color) memcpy(color, c->color, 3 * N * sizeof(REAL));
    if (final_T) memcpy(final_T, c->final_T, N * sizeof(REAL));
    if (n_contrib) memcpy(n_contrib, c->n_contrib, N * 4);
    if (fragile_px) memcpy(fragile_px, c->fragile_px, N);
}
void FN(get_geom)(const FN(ctx) *c, int32_t *radii, REAL *xy, REAL *depth, REAL *cov3D, REAL *conic_op, REAL *rgb,
                  uint8_t *clamped, int32_t *rect, uint32_t *tiles_touched, uint8_t *fragile_g)
{
    size_t P = c->P;
    if (radii) memcpy(radii, c->radii, P * 4);
    if (xy) memcpy(xy, c->xy, 2 * P * sizeof(REAL));
    if (depth) memcpy(depth, c->depth, P * sizeof(REAL));
    if (cov3D) memcpy(cov3D, c->cov3D, 6 * P * sizeof(REAL));
    if (conic_op) memcpy(conic_op, c->conic_op, 4 * P * sizeof(REAL));
    if (rgb) memcpy(rgb, c->rgb, 3 * P * sizeof(REAL));
    if (clamped) memcpy(clamped, c->clamped, 3 * P);
    if (rect) memcpy(rect, c->rect, 4 * P * 4);
    if (tiles_touched) memcpy(tiles_touched, c->tiles_touched, P * 4);
    if (fragile_g) memcpy(fragile_g, c->fragile_g, P);
}
void FN(get_binning)(const FN(ctx) *c, uint64_t *keys, uint32_t *vals, int64_t *ranges)
{
    if (keys) memcpy(keys, c->keys, (size_t)c->R * 8);
    if (vals) memcpy(vals, c->vals, (size_t)c->R * 4);
    if (ranges) memcpy(ranges, c->ranges, (size_t)2 * c->Gx * c->Gy * 8);
}
