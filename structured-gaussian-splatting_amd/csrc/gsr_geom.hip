// gsr_geom.hip — per-Gaussian kernels: forward preprocess (K1), geometry backward (K8+K9 fused),
// frustum test (K10).  One thread per Gaussian; the maths lives in gsr_math.h.
//
// HBM traffic per Gaussian: K1 reads 44 + 12K B (means 12, scale 12, quat 16, opacity 4,
// K SH triples) and writes 48 (record) + 4 (radii) + 4 (tiles) + 1 (clamp mask); K8+K9 reads
// 48 (screen grads) + 44 + 12K + 5 and writes 40 + 12M.  Camera matrices are wave-uniform loads that
// the compiler scalarises (s_load) — they never cost vector memory bandwidth.
#include "gsr_internal.h"

namespace gsr {

constexpr int kGeomBlock = 256;

template <int DEG>
__global__ __launch_bounds__(kGeomBlock) void k_preprocess(FrameK f, const float *__restrict__ view,
                                                           const float *__restrict__ proj, const float *__restrict__ campos,
                                                           const float *__restrict__ means, const float *__restrict__ scales,
                                                           const float *__restrict__ rots, const float *__restrict__ covpre,
                                                           const float *__restrict__ opac, const float *__restrict__ shs,
                                                           const float *__restrict__ colpre, float4 *__restrict__ records,
                                                           uint32_t *__restrict__ tiles, uint8_t *__restrict__ clamped,
                                                           int32_t *__restrict__ radii, uint32_t *__restrict__ sort_keys,
                                                           uint32_t *__restrict__ sort_vals)
{
    const int i = blockIdx.x * kGeomBlock + threadIdx.x;
    if (i >= f.P) return;
    float V[16], PV[16], cp[3];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = view[k]; PV[k] = proj[k]; }
    cp[0] = campos[0]; cp[1] = campos[1]; cp[2] = campos[2];
    const float p[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    float sc[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, cv[6];
    if (covpre) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cv[k] = covpre[6 * (size_t)i + k];
    } else {
        sc[0] = scales[3 * i]; sc[1] = scales[3 * i + 1]; sc[2] = scales[3 * i + 2];
        const float4 qq = reinterpret_cast<const float4 *>(rots)[i];
        q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
    }
    PreOut o;
    preprocess_one<DEG>(f, V, PV, cp, p, sc, q, covpre ? cv : nullptr, opac[i], shs ? shs + (size_t)i * f.M * 3 : nullptr,
                   colpre ? colpre + 3 * (size_t)i : nullptr, o);
    radii[i] = o.radius;
    tiles[i] = o.tiles;
    clamped[i] = (uint8_t)o.clamped;
    records[3 * (size_t)i + 0] = make_float4(o.s.x, o.s.y, o.s.cA, o.s.cB);
    records[3 * (size_t)i + 1] = make_float4(o.s.cC, o.s.op, o.s.r, o.s.g);
    records[3 * (size_t)i + 2] = make_float4(o.s.b, o.s.depth, o.s.rect_x, o.s.rect_y);
    // depth-sort key: binary32 pattern of the (positive) view depth; invisible Gaussians sort last.  Visibility
    // is that of the FULL image (radius > 0), not of this rank's slab, so that the depth order is identical on
    // every rank of a sharded render (the multi-GPU gradient exchange indexes by depth rank).
    sort_keys[i] = o.radius > 0 ? __float_as_uint(o.s.depth) : 0xFFFFFFFFu;
    sort_vals[i] = (uint32_t)i;
}

int launch_preprocess(const FrameK &f, const gsr_camera &cam, const gsr_gaussians &g, GeomWS &ws, int32_t *radii,
                      bool debug, hipStream_t s)
{
    if (f.P == 0) return GSR_OK;
    const int grid = (f.P + kGeomBlock - 1) / kGeomBlock;
    ProfileScope prof("preprocess", s);
#define GSR_PRE(DEG)                                                                                              \
    hipLaunchKernelGGL(k_preprocess<DEG>, dim3(grid), dim3(kGeomBlock), 0, s, f, cam.viewmatrix, cam.projmatrix,  \
                       cam.campos, g.means3D, g.scales, g.rotations, g.cov3D_precomp, g.opacities, g.shs,         \
                       g.colors_precomp, ws.records, ws.tiles_touched, ws.clamped, radii, ws.sort_keys[0],  \
                       ws.sort_vals[0])
    switch (g.shs ? f.D : 0) {
        case 0: GSR_PRE(0); break;
        case 1: GSR_PRE(1); break;
        case 2: GSR_PRE(2); break;
        default: GSR_PRE(3); break;
    }
#undef GSR_PRE
    GSR_LAUNCH_CHECK("preprocess", debug, s);
    return GSR_OK;
}

// ---- K8 + K9: dL/d(screen-space quantities) -> dL/d(inputs) for Gaussians [g0, g1).
template <int DEG>
__global__ __launch_bounds__(kGeomBlock) void k_geom_bwd(FrameK f, int g0, int g1, const float *__restrict__ view,
                                                         const float *__restrict__ proj, const float *__restrict__ campos,
                                                         const float *__restrict__ means, const float *__restrict__ scales,
                                                         const float *__restrict__ rots, const float *__restrict__ covpre,
                                                         const float *__restrict__ shs, int has_colpre,
                                                         const int32_t *__restrict__ radii, const uint8_t *__restrict__ clamped,
                                                         const float4 *__restrict__ screen, gsr_grads out)
{
    __shared__ float sh_stage[(kGeomBlock / 64) * 64 * 49];
    const int i = g0 + blockIdx.x * kGeomBlock + threadIdx.x;
    const bool in_range = i < g1;
    const int M = f.M;
    const bool visible = in_range && radii[i] > 0;
    GeomGrad g;
#pragma unroll
    for (int k = 0; k < 3; ++k) { g.dmean[k] = 0.f; g.dcolor[k] = 0.f; g.dscale[k] = 0.f; }
    g.dmean2D[0] = g.dmean2D[1] = 0.f; g.dopacity = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) g.drot[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) g.dcov[k] = 0.f;
    float dsh[48];
    const int K = (DEG + 1) * (DEG + 1);
    // A Gaussian that no pixel accepted (occluded behind saturated tiles, or just too faint everywhere) has an
    // all-zero screen-space gradient; every output of A.10 is linear in it, so its rows are exact zeros and none
    // of its inputs need to be read.  In depth-complex scenes that is the vast majority of the visible set.
    bool live = visible;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0;
    if (visible) {
        s0 = screen[3 * (size_t)i]; s1 = screen[3 * (size_t)i + 1]; s2 = screen[3 * (size_t)i + 2];
        live = (s0.x != 0.f) | (s0.y != 0.f) | (s0.z != 0.f) | (s0.w != 0.f) | (s1.x != 0.f) | (s1.y != 0.f) |
               (s1.z != 0.f) | (s1.w != 0.f) | (s2.x != 0.f);
    }
    if (live) {
        float V[16], PV[16], cp[3];
#pragma unroll
        for (int k = 0; k < 16; ++k) { V[k] = view[k]; PV[k] = proj[k]; }
        cp[0] = campos[0]; cp[1] = campos[1]; cp[2] = campos[2];
        const float p[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
        float sc[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, cv[6];
        if (covpre) {
#pragma unroll
            for (int k = 0; k < 6; ++k) cv[k] = covpre[6 * (size_t)i + k];
        } else {
            sc[0] = scales[3 * i]; sc[1] = scales[3 * i + 1]; sc[2] = scales[3 * i + 2];
            const float4 qq = reinterpret_cast<const float4 *>(rots)[i];
            q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
        }
        const float sg[9] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, s2.x};
        geom_backward_one<DEG>(f, V, PV, cp, p, sc, q, covpre ? cv : nullptr, shs ? shs + (size_t)i * M * 3 : nullptr,
                          has_colpre != 0, clamped[i], sg, g, (shs && out.shs) ? dsh : nullptr);
    }
    if (!in_range) { /* lanes past the end only help with the cooperative SH store below */ }
    else {
    if (out.means3D) { out.means3D[3 * i] = g.dmean[0]; out.means3D[3 * i + 1] = g.dmean[1]; out.means3D[3 * i + 2] = g.dmean[2]; }
    if (out.means2D) { out.means2D[3 * i] = g.dmean2D[0]; out.means2D[3 * i + 1] = g.dmean2D[1]; out.means2D[3 * i + 2] = 0.f; }
    if (out.opacities) out.opacities[i] = g.dopacity;
    if (out.colors_precomp && has_colpre) {
        out.colors_precomp[3 * i] = g.dcolor[0]; out.colors_precomp[3 * i + 1] = g.dcolor[1]; out.colors_precomp[3 * i + 2] = g.dcolor[2];
    }
    if (out.scales && !covpre) { out.scales[3 * i] = g.dscale[0]; out.scales[3 * i + 1] = g.dscale[1]; out.scales[3 * i + 2] = g.dscale[2]; }
    if (out.rotations && !covpre) reinterpret_cast<float4 *>(out.rotations)[i] = make_float4(g.drot[0], g.drot[1], g.drot[2], g.drot[3]);
    if (out.cov3D_precomp && covpre) {
#pragma unroll
        for (int k = 0; k < 6; ++k) out.cov3D_precomp[6 * (size_t)i + k] = g.dcov[k];
    }
    }
    if (out.shs && shs) {
        // dL/dsh rows are 12*M bytes per Gaussian: a wave's 64 rows form one contiguous run, written with
        // lane-contiguous stores.  Fast path (no live Gaussian in the wave): zeros straight from registers.
        // Otherwise the rows go through LDS (row stride 3M+1 dwords: conflict-free transposition).
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int row = 3 * M;                                    // floats per Gaussian
        const int wave_first = i - lane;                           // first Gaussian of this wave
        const int n_rows = min(64, g1 - wave_first);
        float *dst = out.shs + (size_t)wave_first * row;
        const int total = n_rows * row;
        if (__ballot(live) == 0ull) {
            for (int e = lane; e < total; e += 64) dst[e] = 0.f;
        } else {
            float *stage = sh_stage + w * (64 * 49);
            const int nlive = live ? 3 * K : 0;
#pragma unroll
            for (int k = 0; k < 48; ++k)
                if (k < row) stage[lane * (row + 1) + k] = k < nlive ? dsh[k] : 0.f;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int e = lane; e < total; e += 64) dst[e] = stage[(e / row) * (row + 1) + e % row];
        }
    }
}

// ---- sparse variant for depth-complex frames: every output was zero-filled by memset; only the Gaussians of the
// binned depth prefix (rank < n_ranks) can have a non-zero screen-space gradient.  One thread per rank, rows
// written individually (they are few).
template <int DEG>
__global__ __launch_bounds__(kGeomBlock) void k_geom_bwd_sparse(FrameK f, int n_ranks, const uint32_t *__restrict__ order,
                                                                const float *__restrict__ view, const float *__restrict__ proj,
                                                                const float *__restrict__ campos, const float *__restrict__ means,
                                                                const float *__restrict__ scales, const float *__restrict__ rots,
                                                                const float *__restrict__ covpre, const float *__restrict__ shs,
                                                                int has_colpre, const int32_t *__restrict__ radii,
                                                                const uint8_t *__restrict__ clamped,
                                                                const float4 *__restrict__ screen, gsr_grads out)
{
    const int r = blockIdx.x * kGeomBlock + threadIdx.x;
    if (r >= n_ranks) return;
    const int i = (int)order[r];
    if (radii[i] <= 0) return;
    const float4 s0 = screen[3 * (size_t)i], s1 = screen[3 * (size_t)i + 1], s2 = screen[3 * (size_t)i + 2];
    const bool live = (s0.x != 0.f) | (s0.y != 0.f) | (s0.z != 0.f) | (s0.w != 0.f) | (s1.x != 0.f) | (s1.y != 0.f) |
                      (s1.z != 0.f) | (s1.w != 0.f) | (s2.x != 0.f);
    if (!live) return;
    const int M = f.M;
    float V[16], PV[16], cp[3];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = view[k]; PV[k] = proj[k]; }
    cp[0] = campos[0]; cp[1] = campos[1]; cp[2] = campos[2];
    const float p[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    float sc[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, cv[6];
    if (covpre) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cv[k] = covpre[6 * (size_t)i + k];
    } else {
        sc[0] = scales[3 * i]; sc[1] = scales[3 * i + 1]; sc[2] = scales[3 * i + 2];
        const float4 qq = reinterpret_cast<const float4 *>(rots)[i];
        q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
    }
    const float sg[9] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, s2.x};
    GeomGrad g;
    float dsh[48];
    geom_backward_one<DEG>(f, V, PV, cp, p, sc, q, covpre ? cv : nullptr, shs ? shs + (size_t)i * M * 3 : nullptr, has_colpre != 0,
                           clamped[i], sg, g, (shs && out.shs) ? dsh : nullptr);
    if (out.means3D) { out.means3D[3 * i] = g.dmean[0]; out.means3D[3 * i + 1] = g.dmean[1]; out.means3D[3 * i + 2] = g.dmean[2]; }
    if (out.means2D) { out.means2D[3 * i] = g.dmean2D[0]; out.means2D[3 * i + 1] = g.dmean2D[1]; }
    if (out.opacities) out.opacities[i] = g.dopacity;
    if (out.colors_precomp && has_colpre) {
        out.colors_precomp[3 * i] = g.dcolor[0]; out.colors_precomp[3 * i + 1] = g.dcolor[1]; out.colors_precomp[3 * i + 2] = g.dcolor[2];
    }
    if (out.scales && !covpre) { out.scales[3 * i] = g.dscale[0]; out.scales[3 * i + 1] = g.dscale[1]; out.scales[3 * i + 2] = g.dscale[2]; }
    if (out.rotations && !covpre) reinterpret_cast<float4 *>(out.rotations)[i] = make_float4(g.drot[0], g.drot[1], g.drot[2], g.drot[3]);
    if (out.cov3D_precomp && covpre) {
#pragma unroll
        for (int k = 0; k < 6; ++k) out.cov3D_precomp[6 * (size_t)i + k] = g.dcov[k];
    }
    if (out.shs && shs) {
        float *dst = out.shs + (size_t)i * M * 3;
        constexpr int K3 = 3 * (DEG + 1) * (DEG + 1);
#pragma unroll
        for (int k = 0; k < 48; ++k)
            if (k < K3 && k < 3 * M) dst[k] = dsh[k];
    }
}

int launch_geom_bwd(const FrameK &f, const gsr_camera &cam, const gsr_gaussians &g, const int32_t *radii, const GeomWS &gw,
                    const float *screen_grads, int g0, int g1, int n_ranks, const gsr_grads &out, bool debug, hipStream_t s)
{
    if (g1 <= g0) return GSR_OK;
    if (n_ranks >= 0 && g0 == 0 && g1 == f.P && (long long)n_ranks * 4 < (long long)f.P) {
        // depth-complex frame: almost every gradient row is zero -> memset the outputs, then visit the binned prefix only
        ProfileScope prof("geom_bwd", s);
        const size_t P = (size_t)f.P;
        if (out.means3D) GSR_HIP_CHECK(hipMemsetAsync(out.means3D, 0, P * 12, s));
        if (out.means2D) GSR_HIP_CHECK(hipMemsetAsync(out.means2D, 0, P * 12, s));
        if (out.opacities) GSR_HIP_CHECK(hipMemsetAsync(out.opacities, 0, P * 4, s));
        if (out.colors_precomp && g.colors_precomp) GSR_HIP_CHECK(hipMemsetAsync(out.colors_precomp, 0, P * 12, s));
        if (out.scales && !g.cov3D_precomp) GSR_HIP_CHECK(hipMemsetAsync(out.scales, 0, P * 12, s));
        if (out.rotations && !g.cov3D_precomp) GSR_HIP_CHECK(hipMemsetAsync(out.rotations, 0, P * 16, s));
        if (out.cov3D_precomp && g.cov3D_precomp) GSR_HIP_CHECK(hipMemsetAsync(out.cov3D_precomp, 0, P * 24, s));
        if (out.shs && g.shs) GSR_HIP_CHECK(hipMemsetAsync(out.shs, 0, P * 12 * (size_t)f.M, s));
        if (n_ranks > 0) {
            const int sgrid = (n_ranks + kGeomBlock - 1) / kGeomBlock;
#define GSR_GS(DEG)                                                                                                  \
    hipLaunchKernelGGL(k_geom_bwd_sparse<DEG>, dim3(sgrid), dim3(kGeomBlock), 0, s, f, n_ranks, gw.order, cam.viewmatrix, \
                       cam.projmatrix, cam.campos, g.means3D, g.scales, g.rotations, g.cov3D_precomp, g.shs,         \
                       g.colors_precomp ? 1 : 0, radii, gw.clamped, reinterpret_cast<const float4 *>(screen_grads), out)
            switch (g.shs ? f.D : 0) {
                case 0: GSR_GS(0); break;
                case 1: GSR_GS(1); break;
                case 2: GSR_GS(2); break;
                default: GSR_GS(3); break;
            }
#undef GSR_GS
        }
        GSR_LAUNCH_CHECK("geom_bwd_sparse", debug, s);
        return GSR_OK;
    }
    const int grid = (g1 - g0 + kGeomBlock - 1) / kGeomBlock;
    ProfileScope prof("geom_bwd", s);
#define GSR_GB(DEG)                                                                                               \
    hipLaunchKernelGGL(k_geom_bwd<DEG>, dim3(grid), dim3(kGeomBlock), 0, s, f, g0, g1, cam.viewmatrix, cam.projmatrix, \
                       cam.campos, g.means3D, g.scales, g.rotations, g.cov3D_precomp, g.shs, g.colors_precomp ? 1 : 0, \
                       radii, gw.clamped, reinterpret_cast<const float4 *>(screen_grads), out)
    switch (g.shs ? f.D : 0) {
        case 0: GSR_GB(0); break;
        case 1: GSR_GB(1); break;
        case 2: GSR_GB(2); break;
        default: GSR_GB(3); break;
    }
#undef GSR_GB
    GSR_LAUNCH_CHECK("geom_bwd", debug, s);
    return GSR_OK;
}

__global__ __launch_bounds__(kGeomBlock) void k_mark_visible(int P, const float *__restrict__ means,
                                                             const float *__restrict__ view, uint8_t *__restrict__ present)
{
    const int i = blockIdx.x * kGeomBlock + threadIdx.x;
    if (i >= P) return;
    float V[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) V[k] = view[k];
    const float p[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    present[i] = in_frustum(p, V) ? 1 : 0;
}

int launch_mark_visible(int P, const float *means3D, const float *view, uint8_t *present, hipStream_t s)
{
    if (P == 0) return GSR_OK;
    hipLaunchKernelGGL(k_mark_visible, dim3((P + kGeomBlock - 1) / kGeomBlock), dim3(kGeomBlock), 0, s, P, means3D, view, present);
    GSR_LAUNCH_CHECK("mark_visible", false, s);
    return GSR_OK;
}

}  // namespace gsr
