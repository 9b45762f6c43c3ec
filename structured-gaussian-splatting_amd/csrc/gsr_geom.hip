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

// One Gaussian's inputs in registers.  RAW != 0: the tensors are the optimizer's raw parameters and
// the activations (exp / normalize / sigmoid) happen here, in registers.  RAW == 1: the SH coefficients arrive as the reference's
// two tensors (features_dc [P,1,3], features_rest [P,M-1,3]) and are joined in registers (its torch.cat); RAW == 2: they are ONE
// interleaved table [P,M,3] (scene.GaussianModel's packed leaf), read and differentiated exactly like activated input.
template <int DEG, int RAW>
struct GaussIn {
    float p[3], sc[3], q[4], cv[6], opacity;
    float shl[RAW == 1 ? 3 * (DEG + 1) * (DEG + 1) : 1];     // RAW == 1: the [K,3] coefficients, features_dc ++ features_rest
    const float *sh_global;   // otherwise: this Gaussian's [M,3] coefficients in the shs tensor (or null)
    RawAct act;
    // what preprocess_one / geom_backward_one read; no pointer member aliases shl, so the array stays in registers
    __device__ __forceinline__ const float *sh() const { if constexpr (RAW == 1) return shl; else return sh_global; }
};

template <int DEG, int RAW>
__device__ __forceinline__ void load_gaussian(int i, int M, const float *__restrict__ means, const float *__restrict__ scales,
                                              const float *__restrict__ rots, const float *__restrict__ covpre,
                                              const float *__restrict__ opac, const float *__restrict__ shs,
                                              const float *__restrict__ shs_rest, bool with_sh, GaussIn<DEG, RAW> &in)
{
    in.p[0] = means[3 * i]; in.p[1] = means[3 * i + 1]; in.p[2] = means[3 * i + 2];
    in.sc[0] = in.sc[1] = in.sc[2] = 0.f;
    in.q[0] = 1.f; in.q[1] = in.q[2] = in.q[3] = 0.f;
    if constexpr (RAW) {
        const float ls[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        const float4 qq = reinterpret_cast<const float4 *>(rots)[i];
        const float rq[4] = {qq.x, qq.y, qq.z, qq.w};
        activate_raw(ls, rq, opac[i], in.act);
#pragma unroll
        for (int k = 0; k < 3; ++k) in.sc[k] = in.act.scale[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) in.q[k] = in.act.q[k];
        in.opacity = in.act.opacity;
        if constexpr (RAW == 1) {
            constexpr int K = (DEG + 1) * (DEG + 1);
            if (with_sh) {                       // culled Gaussians never read their coefficients
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) in.shl[ch] = shs[3 * (size_t)i + ch];
                const float *rest = shs_rest + (size_t)i * (M - 1) * 3;
#pragma unroll
                for (int k = 3; k < 3 * K; ++k) in.shl[k] = rest[k - 3];
            } else {
#pragma unroll
                for (int k = 0; k < 3 * K; ++k) in.shl[k] = 0.f;
            }
            in.sh_global = nullptr;
        } else {
            in.sh_global = shs ? shs + (size_t)i * M * 3 : nullptr;
        }
    } else {
        if (covpre) {
#pragma unroll
            for (int k = 0; k < 6; ++k) in.cv[k] = covpre[6 * (size_t)i + k];
        } else {
            in.sc[0] = scales[3 * i]; in.sc[1] = scales[3 * i + 1]; in.sc[2] = scales[3 * i + 2];
            const float4 qq = reinterpret_cast<const float4 *>(rots)[i];
            in.q[0] = qq.x; in.q[1] = qq.y; in.q[2] = qq.z; in.q[3] = qq.w;
        }
        in.opacity = opac ? opac[i] : 0.f;
        in.sh_global = shs ? shs + (size_t)i * M * 3 : nullptr;
    }
}

template <int DEG, int RAW, bool LAZY>
__global__ __launch_bounds__(kGeomBlock) void k_preprocess(FrameK f, const float *__restrict__ view,
                                                           const float *__restrict__ proj, const float *__restrict__ campos,
                                                           const float *__restrict__ means, const float *__restrict__ scales,
                                                           const float *__restrict__ rots, const float *__restrict__ covpre,
                                                           const float *__restrict__ opac, const float *__restrict__ shs,
                                                           const float *__restrict__ shs_rest,
                                                           const float *__restrict__ colpre, float4 *__restrict__ records,
                                                           uint2 *__restrict__ tiles_mass, uint8_t *__restrict__ clamped,
                                                           int32_t *__restrict__ radii, uint32_t *__restrict__ sort_keys,
                                                           uint32_t *__restrict__ sort_vals, uint32_t *__restrict__ prefilter_flag,
                                                           uint4 *__restrict__ zero16, int zero16_n)
{
    const int i = blockIdx.x * kGeomBlock + threadIdx.x;
    // the selection's histograms (SelState, 130 KB) start every frame at zero: cleared here, one 16-byte store per thread,
    // instead of by a memset launch of its own in front of this kernel (the histogram kernels run after this one)
    for (int z = i; z < zero16_n; z += (int)gridDim.x * kGeomBlock) zero16[z] = make_uint4(0u, 0u, 0u, 0u);
    if (i >= f.P) return;
    float V[16], PV[16], cp[3];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = view[k]; PV[k] = proj[k]; }
    cp[0] = campos[0]; cp[1] = campos[1]; cp[2] = campos[2];
    GaussIn<DEG, RAW> in;
    const float p0[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    const bool near_ok = in_frustum(p0, V);
    // prefiltered = "the caller guarantees every point passes the frustum test" (the upstream kernel traps when one
    // does not); here the frame is refused with GSR_ERR_PREFILTERED after the plan readback
    if (prefilter_flag && !near_ok) *prefilter_flag = 1u;
    // LAZY: SH colours are evaluated per depth chunk (k_chunk_colors), only for the chunks that get binned; this kernel then
    // reads no coefficient at all (two thirds of its bytes at degree 3)
    load_gaussian<DEG, RAW>(i, f.M, means, scales, rots, covpre, opac, shs, shs_rest, near_ok && !LAZY, in);
    PreOut o;
    preprocess_one<DEG>(f, V, PV, cp, in.p, in.sc, in.q, (!RAW && covpre) ? in.cv : nullptr, in.opacity, LAZY ? nullptr : in.sh(),
                        (!RAW && colpre) ? colpre + 3 * (size_t)i : nullptr, o);
    radii[i] = o.radius;
    // mass in fixed point: integer sums are order-independent
    tiles_mass[i] = make_uint2(o.tiles, (uint32_t)fminf(o.mass * kMassUnitsPerPixelNeper, 4294967040.f));
    clamped[i] = (uint8_t)o.clamped;
    records[3 * (size_t)i + 0] = make_float4(o.s.x, o.s.y, o.s.cA, o.s.cB);
    records[3 * (size_t)i + 1] = make_float4(o.s.cC, o.s.op, o.s.r, o.s.g);
    records[3 * (size_t)i + 2] = make_float4(o.s.b, o.s.depth, o.s.rect_x, o.s.rect_y);
    // depth-sort key: binary32 pattern of the (positive) view depth; invisible Gaussians sort last.  Visibility
    // is that of the FULL image (radius > 0), not of this rank's slab, so that the depth order is identical on
    // every rank of a sharded render (the multi-GPU gradient exchange selects by depth key).
    sort_keys[i] = o.radius > 0 ? __float_as_uint(o.s.depth) : 0xFFFFFFFFu;
    (void)sort_vals;                           // the depth order is built by selection (gsr_select.hip)
}

int launch_preprocess(const FrameK &f, const gsr_camera &cam, const gsr_gaussians &g, GeomWS &ws, int32_t *radii,
                      bool prefiltered, bool debug, hipStream_t s)
{
    if (f.P == 0) return GSR_OK;
    const int grid = (f.P + kGeomBlock - 1) / kGeomBlock;
    uint32_t *prefilter_flag = nullptr;
    if (prefiltered) {
        prefilter_flag = &ws.ctrl->prefilter_violation;
        GSR_HIP_CHECK(hipMemsetAsync(prefilter_flag, 0, 4, s));
    }
    ProfileScope prof("preprocess", s);
#define GSR_PRE(DEG, RAW, LAZY)                                                                                     \
    hipLaunchKernelGGL((k_preprocess<DEG, RAW, LAZY>), dim3(grid), dim3(kGeomBlock), 0, s, f, cam.viewmatrix, cam.projmatrix,  \
                       cam.campos, g.means3D, g.scales, g.rotations, g.cov3D_precomp, g.opacities, g.shs, g.shs_rest,   \
                       g.colors_precomp, ws.records, ws.tiles_mass, ws.clamped, radii, ws.sort_keys[0],                 \
                       ws.sort_vals[0], prefilter_flag, reinterpret_cast<uint4 *>(ws.sel), (int)(sizeof(SelState) / 16))
    if (g.shs) {                  // colours from SH: lazily, per binned chunk
        if (g.raw) GSR_PRE(0, 1, true);            // lazy colours: no coefficient is read here, the SH layout does not matter
        else GSR_PRE(0, 0, true);
    } else {
        GSR_PRE(0, 0, false);  // precomputed colours are copied into the record here
    }
#undef GSR_PRE
    GSR_LAUNCH_CHECK("preprocess", debug, s);
    return GSR_OK;
}

// raw mode with the reference's two SH tensors (features_dc + features_rest): raw == 1.  raw == 2: one interleaved [P,M,3] table
static inline bool raw_split_sh(const FrameK &f, const gsr_gaussians &g) { (void)f; return g.raw == 1; }

// ---- A.6, lazily: SH colour (+ clamp flags) of the Gaussians of depth ranks [r0, r1) — one binned chunk — patched into
// their splat records right before the chunk is blended.  On a depth-complex frame that is a few thousand Gaussians
// instead of all the visible ones.
template <int DEG, int RAW>
__global__ __launch_bounds__(kGeomBlock) void k_chunk_colors(FrameK f, int r0, int r1, const uint32_t *__restrict__ order,
                                                             const uint32_t *__restrict__ cnt_open,
                                                             const float *__restrict__ campos, const float *__restrict__ means,
                                                             const float *__restrict__ shs, const float *__restrict__ shs_rest,
                                                             float4 *__restrict__ records, uint8_t *__restrict__ clamped)
{
    const int r = r0 + blockIdx.x * kGeomBlock + threadIdx.x;
    if (r >= r1) return;
    if (cnt_open[r] == 0u) return;               // no tile took this Gaussian (a late chunk: most tiles are closed): nobody reads its colour
    const int i = (int)order[r];
    const float cp[3] = {campos[0], campos[1], campos[2]};
    const float p[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    constexpr int K = (DEG + 1) * (DEG + 1);
    float shl[3 * K];
    if constexpr (RAW == 1) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) shl[ch] = shs[3 * (size_t)i + ch];
        const float *rest = shs_rest + (size_t)i * (f.M - 1) * 3;
#pragma unroll
        for (int k = 3; k < 3 * K; ++k) shl[k] = rest[k - 3];
    } else {
        const float *src = shs + (size_t)i * f.M * 3;
        if constexpr ((3 * K) % 4 == 0) {
            if ((f.M * 3) % 4 == 0) {              // the row starts on a 16-byte boundary: a gathered row is 12 x 16 B, not 48 x 4 B
                const float4 *src4 = reinterpret_cast<const float4 *>(src);
#pragma unroll
                for (int k = 0; k < 3 * K / 4; ++k) {
                    const float4 v = src4[k];
                    shl[4 * k] = v.x; shl[4 * k + 1] = v.y; shl[4 * k + 2] = v.z; shl[4 * k + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 3 * K; ++k) shl[k] = src[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3 * K; ++k) shl[k] = src[k];
        }
    }
    float rgb[3];
    unsigned cl;
    sh_color_one<DEG>(f, cp, p, shl, rgb, cl);
    float *rec = reinterpret_cast<float *>(records + 3 * (size_t)i);
    rec[6] = rgb[0]; rec[7] = rgb[1]; rec[8] = rgb[2];          // record = {x, y, qA, qB | qC, lop, r, g | b, depth, rect, rect}
    clamped[i] = (uint8_t)cl;
}

// The same for a chunk that holds EVERY visible Gaussian (a frame that does not saturate: one chunk): Gaussians are taken in
// index order, so a wave's 64 coefficient rows are one contiguous run, loaded with full 16-byte-per-lane coalescing into
// LDS rows (a gather by depth rank reads 192-byte rows scattered over the tensor).  Invisible Gaussians (zeroed record)
// are skipped.
template <int DEG, int RAW>
__global__ __launch_bounds__(kGeomBlock) void k_chunk_colors_all(FrameK f, const float *__restrict__ campos, const float *__restrict__ means,
                                                                 const float *__restrict__ shs, const float *__restrict__ shs_rest,
                                                                 float4 *__restrict__ records, uint8_t *__restrict__ clamped)
{
    constexpr int K = (DEG + 1) * (DEG + 1);
    constexpr int kRow = 3 * K + 1;                            // +1: lanes read their rows conflict-free
    __shared__ float sh_stage[(kGeomBlock / 64) * 64 * kRow];
    const int i = blockIdx.x * kGeomBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave_first = i - lane;
    const int n_rows = min(64, f.P - wave_first);
    if (n_rows <= 0) return;
    const bool visible = i < f.P && reinterpret_cast<const float *>(records + 3 * (size_t)i)[9] > 0.f;      // record.depth
    if (__ballot(visible) == 0ull) return;
    float *stage = sh_stage + wv * (64 * kRow);
    const int rowf = 3 * f.M;                                  // stored floats per Gaussian (>= 3 K)
    if constexpr (RAW == 1) {
        for (int e = lane; e < n_rows * 3; e += 64) stage[(e / 3) * kRow + e % 3] = shs[(size_t)wave_first * 3 + e];
        if constexpr (K > 1) {
            const int row = rowf - 3;
            const float *src = shs_rest + (size_t)wave_first * row;
            for (int e = lane; e < n_rows * row; e += 64) {
                const int c = e % row;
                if (c < 3 * K - 3) stage[(e / row) * kRow + 3 + c] = src[e];
            }
        }
    } else {
        const float *src = shs + (size_t)wave_first * rowf;
        const int total = n_rows * rowf;
        if ((((uintptr_t)src) & 15) == 0 && rowf == 3 * K && K == 16 && n_rows == 64) {
            // the common shape (degree 3 stored and active, a full wave): all twelve 1-KB loads in flight before the first LDS store
            const float4 *src4 = reinterpret_cast<const float4 *>(src);
            float4 v[12];
#pragma unroll
            for (int it = 0; it < 12; ++it) v[it] = src4[lane + 64 * it];
#pragma unroll
            for (int it = 0; it < 12; ++it) {
                const int e = 4 * (lane + 64 * it), r = e / 48, c = e - r * 48;
                float *d = stage + r * kRow + c;
                d[0] = v[it].x; d[1] = v[it].y; d[2] = v[it].z; d[3] = v[it].w;
            }
        } else if ((((uintptr_t)src) & 15) == 0 && rowf % 4 == 0) {
            const float4 *src4 = reinterpret_cast<const float4 *>(src);
            for (int e4 = lane; e4 < total / 4; e4 += 64) {
                const float4 v = src4[e4];
                const int e = 4 * e4, r = e / rowf, c = e - r * rowf;          // rowf % 4 == 0: the four floats share a row
                if (c < 3 * K) {                                               // (3 K) % 4 == 0 whenever K > 1; K == 1 has rowf % 4 != 0 unless M % 4 == 0
                    float *d = stage + r * kRow + c;
                    d[0] = v.x;
                    if (c + 1 < 3 * K) d[1] = v.y;
                    if (c + 2 < 3 * K) d[2] = v.z;
                    if (c + 3 < 3 * K) d[3] = v.w;
                }
            }
        } else {
            for (int e = lane; e < total; e += 64) {
                const int r = e / rowf, c = e - r * rowf;
                if (c < 3 * K) stage[r * kRow + c] = src[e];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (!visible) return;
    const float cp[3] = {campos[0], campos[1], campos[2]};
    const float p[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    float shl[3 * K];
#pragma unroll
    for (int k = 0; k < 3 * K; ++k) shl[k] = stage[lane * kRow + k];
    float rgb[3];
    unsigned cl;
    sh_color_one<DEG>(f, cp, p, shl, rgb, cl);
    float *rec = reinterpret_cast<float *>(records + 3 * (size_t)i);
    rec[6] = rgb[0]; rec[7] = rgb[1]; rec[8] = rgb[2];
    clamped[i] = (uint8_t)cl;
}

int launch_chunk_colors(const FrameK &f, const gsr_camera &cam, const gsr_gaussians &g, int r0, int r1, int num_visible, GeomWS &ws,
                        bool debug, hipStream_t s)
{
    if (!g.shs || r1 <= r0) return GSR_OK;           // precomputed colours went into the records in the preprocess
    ProfileScope prof("chunk_colors", s);
    if (r0 == 0 && r1 >= num_visible && (long long)num_visible * 2 >= (long long)f.P) {
        // the chunk is every visible Gaussian and most Gaussians are visible: walk the tensor in index order
        const int grid_all = (f.P + kGeomBlock - 1) / kGeomBlock;
#define GSR_CA(DEG, RAW)                                                                                                \
    hipLaunchKernelGGL((k_chunk_colors_all<DEG, RAW>), dim3(grid_all), dim3(kGeomBlock), 0, s, f, cam.campos, g.means3D, g.shs,  \
                       g.shs_rest, ws.records, ws.clamped)
        if (raw_split_sh(f, g)) {              // (no activations in this kernel: RAW only names the SH layout)
            switch (f.D) {
                case 0: GSR_CA(0, 1); break;
                case 1: GSR_CA(1, 1); break;
                case 2: GSR_CA(2, 1); break;
                default: GSR_CA(3, 1); break;
            }
        } else {
            switch (f.D) {
                case 0: GSR_CA(0, 0); break;
                case 1: GSR_CA(1, 0); break;
                case 2: GSR_CA(2, 0); break;
                default: GSR_CA(3, 0); break;
            }
        }
#undef GSR_CA
        GSR_LAUNCH_CHECK("chunk_colors_all", debug, s);
        return GSR_OK;
    }
    const int grid = (r1 - r0 + kGeomBlock - 1) / kGeomBlock;
#define GSR_CC(DEG, RAW)                                                                                             \
    hipLaunchKernelGGL((k_chunk_colors<DEG, RAW>), dim3(grid), dim3(kGeomBlock), 0, s, f, r0, r1, ws.order, ws.cnt_open, cam.campos, \
                       g.means3D, g.shs, g.shs_rest, ws.records, ws.clamped)
    if (raw_split_sh(f, g)) {
        switch (f.D) {
            case 0: GSR_CC(0, 1); break;
            case 1: GSR_CC(1, 1); break;
            case 2: GSR_CC(2, 1); break;
            default: GSR_CC(3, 1); break;
        }
    } else {
        switch (f.D) {
            case 0: GSR_CC(0, 0); break;
            case 1: GSR_CC(1, 0); break;
            case 2: GSR_CC(2, 0); break;
            default: GSR_CC(3, 0); break;
        }
    }
#undef GSR_CC
    GSR_LAUNCH_CHECK("chunk_colors", debug, s);
    return GSR_OK;
}

// ---- K8 + K9: dL/d(screen-space quantities) -> dL/d(inputs) for Gaussians [g0, g1).
__device__ __forceinline__ bool sh_wanted_or_read(const float *shs, int has_colpre) { return shs != nullptr && !has_colpre; }

template <int DEG, int RAW>
__global__ __launch_bounds__(kGeomBlock) void k_geom_bwd(FrameK f, int g0, int g1, const float *__restrict__ view,
                                                         const float *__restrict__ proj, const float *__restrict__ campos,
                                                         const float *__restrict__ means, const float *__restrict__ scales,
                                                         const float *__restrict__ rots, const float *__restrict__ covpre,
                                                         const float *__restrict__ opac, const float *__restrict__ shs,
                                                         const float *__restrict__ shs_rest, int has_colpre,
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
    constexpr int K = (DEG + 1) * (DEG + 1);
    // dL/dsh leaves through LDS: every lane owns one staging row of 3M (+1 pad) floats; geom_backward_one writes the
    // 3K live entries straight into it, the rest is zero-filled, and the wave then stores its 64 rows as one
    // contiguous, lane-coalesced run
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int rowf = 3 * M;                                       // floats per Gaussian incl. the DC triple
    float *my_row = sh_stage + wv * (64 * 49) + lane * (rowf + 1);
    const bool sh_wanted = shs && (out.shs || (RAW == 1 && out.shs_rest));
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
    // The wave's SH rows ([M,3] per Gaussian, consecutive Gaussians contiguous in memory) come in through the same LDS
    // staging rows the gradients leave by: 64 lanes x 16 B per load instruction instead of 64 single dwords 192 B apart (the
    // per-lane strided form re-fetches every 64-B sector sixteen times).  geom_backward_one reads coefficient k and then
    // overwrites it with its gradient, so one row serves both directions.
    const int wave_first = i - lane;                           // first Gaussian of this wave
    const int n_rows = min(64, g1 - wave_first);
    const unsigned long long live_mask = __ballot(live);
    const bool wave_live = live_mask != 0ull;
    // few live lanes (a frame whose back Gaussians sit behind saturated pixels: 19 % live at 5e6 Gaussians / 4K, in index order
    // every wave holds some): each live lane fetches its own row, three whole 64-byte sectors, instead of the wave staging all 64
    // rows for a dozen readers (0.96 GB of the kernel's 2.6 GB there)
    const bool own_rows = RAW != 1 && __popcll(live_mask) * 3 < 64 && rowf % 4 == 0 && (((uintptr_t)shs) & 15) == 0;
    if (sh_wanted_or_read(shs, has_colpre) && wave_live && rowf > 0 && n_rows > 0 && own_rows) {
        if (live) {
            const float4 *src4 = reinterpret_cast<const float4 *>(shs + (size_t)i * rowf);
            for (int e4 = 0; e4 < rowf / 4; ++e4) {
                const float4 v = src4[e4];
                my_row[4 * e4] = v.x; my_row[4 * e4 + 1] = v.y; my_row[4 * e4 + 2] = v.z; my_row[4 * e4 + 3] = v.w;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    } else if (sh_wanted_or_read(shs, has_colpre) && wave_live && rowf > 0 && n_rows > 0) {
        float *stage_w = sh_stage + wv * (64 * 49);
        if constexpr (RAW == 1) {
            for (int e = lane; e < n_rows * 3; e += 64) stage_w[(e / 3) * (rowf + 1) + e % 3] = shs[(size_t)wave_first * 3 + e];
            const int row = rowf - 3;
            if (row > 0) {
                const float *src = shs_rest + (size_t)wave_first * row;
                for (int e = lane; e < n_rows * row; e += 64) stage_w[(e / row) * (rowf + 1) + 3 + e % row] = src[e];
            }
        } else {
            const float *src = shs + (size_t)wave_first * rowf;
            const int total = n_rows * rowf;
            if ((((uintptr_t)src) & 15) == 0 && rowf == 48 && n_rows == 64) {
                // the common shape (16 coefficients, a full wave): all twelve 1-KB loads in flight before the first LDS store
                const float4 *src4 = reinterpret_cast<const float4 *>(src);
                float4 v[12];
#pragma unroll
                for (int it = 0; it < 12; ++it) v[it] = src4[lane + 64 * it];
#pragma unroll
                for (int it = 0; it < 12; ++it) {
                    const int e = 4 * (lane + 64 * it), r = e / 48, c = e - r * 48;
                    float *d = stage_w + r * 49 + c;
                    d[0] = v[it].x; d[1] = v[it].y; d[2] = v[it].z; d[3] = v[it].w;
                }
            } else if ((((uintptr_t)src) & 15) == 0) {
                const float4 *src4 = reinterpret_cast<const float4 *>(src);
                for (int e4 = lane; e4 < total / 4; e4 += 64) {
                    const float4 v = src4[e4];
                    const int e = 4 * e4;
                    stage_w[(e / rowf) * (rowf + 1) + e % rowf] = v.x;
                    stage_w[((e + 1) / rowf) * (rowf + 1) + (e + 1) % rowf] = v.y;
                    stage_w[((e + 2) / rowf) * (rowf + 1) + (e + 2) % rowf] = v.z;
                    stage_w[((e + 3) / rowf) * (rowf + 1) + (e + 3) % rowf] = v.w;
                }
                for (int e = (total / 4) * 4 + lane; e < total; e += 64) stage_w[(e / rowf) * (rowf + 1) + e % rowf] = src[e];
            } else {
                for (int e = lane; e < total; e += 64) stage_w[(e / rowf) * (rowf + 1) + e % rowf] = src[e];
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (live) {
        float V[16], PV[16], cp[3];
#pragma unroll
        for (int k = 0; k < 16; ++k) { V[k] = view[k]; PV[k] = proj[k]; }
        cp[0] = campos[0]; cp[1] = campos[1]; cp[2] = campos[2];
        GaussIn<DEG, RAW> in;
        load_gaussian<DEG, RAW>(i, M, means, scales, rots, covpre, opac, nullptr, nullptr, false, in);      // SH: from the LDS row
        const float sg[9] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, s2.x};
        // The clamp mask of an SH colour is recomputed from the coefficients (the forward's own function), not read: colours are
        // evaluated lazily, only for Gaussians some tile of THIS frame took, and a rank of a sharded render back-propagates
        // Gaussians that only the other ranks binned.
        unsigned clamp_bits = clamped[i];
        if (shs && !has_colpre) { float rgb_[3]; sh_color_one<DEG>(f, cp, in.p, my_row, rgb_, clamp_bits); }
        geom_backward_one<DEG>(f, V, PV, cp, in.p, in.sc, in.q, (!RAW && covpre) ? in.cv : nullptr, shs ? my_row : nullptr, has_colpre != 0,
                               clamp_bits, sg, g, my_row, sh_wanted);
        if constexpr (RAW) activate_raw_backward(in.act, g);
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
    if (sh_wanted && rowf > 0) {
        const bool any_live = wave_live;
        if (any_live) {
            const int nlive = live ? 3 * K : 0;
            for (int k = nlive; k < rowf; ++k) my_row[k] = 0.f;    // rows of dead lanes, coefficients above the active degree
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        const float *stage = sh_stage + wv * (64 * 49);
        if constexpr (RAW == 1) {
            if (in_range && out.shs) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) out.shs[3 * (size_t)i + ch] = any_live ? my_row[ch] : 0.f;
            }
            const int row = rowf - 3;                              // _features_rest: floats per Gaussian
            if (out.shs_rest && row > 0 && n_rows > 0) {
                float *dst = out.shs_rest + (size_t)wave_first * row;
                const int total = n_rows * row;
                if (!any_live) { for (int e = lane; e < total; e += 64) dst[e] = 0.f; }
                else { for (int e = lane; e < total; e += 64) dst[e] = stage[(e / row) * (rowf + 1) + 3 + e % row]; }
            }
        } else if (n_rows > 0) {
            float *dst = out.shs + (size_t)wave_first * rowf;
            const int total = n_rows * rowf;
            if (rowf == 48 && n_rows == 64 && (((uintptr_t)dst) & 15) == 0) {      // twelve 1-KB stores instead of forty-eight 256-B ones
                float4 *dst4 = reinterpret_cast<float4 *>(dst);
#pragma unroll
                for (int it = 0; it < 12; ++it) {
                    const int e = 4 * (lane + 64 * it), r = e / 48, c = e - r * 48;
                    const float *q = stage + r * 49 + c;
                    dst4[lane + 64 * it] = any_live ? make_float4(q[0], q[1], q[2], q[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            else if (!any_live) { for (int e = lane; e < total; e += 64) dst[e] = 0.f; }
            else { for (int e = lane; e < total; e += 64) dst[e] = stage[(e / rowf) * (rowf + 1) + e % rowf]; }
        }
    }
}

// ---- sparse variant for depth-complex frames: every output was zero-filled by memset; only the Gaussians of the
// binned depth prefix (rank < n_ranks) can have a non-zero screen-space gradient.  One thread per rank, rows
// written individually (they are few).
template <int DEG, int RAW>
__global__ __launch_bounds__(kGeomBlock) void k_geom_bwd_sparse(FrameK f, int n_ranks, const uint32_t *__restrict__ order,
                                                                const float *__restrict__ view, const float *__restrict__ proj,
                                                                const float *__restrict__ campos, const float *__restrict__ means,
                                                                const float *__restrict__ scales, const float *__restrict__ rots,
                                                                const float *__restrict__ covpre, const float *__restrict__ opac,
                                                                const float *__restrict__ shs, const float *__restrict__ shs_rest,
                                                                int has_colpre, const int32_t *__restrict__ radii,
                                                                const uint8_t *__restrict__ clamped,
                                                                const float4 *__restrict__ screen, gsr_grads out,
                                                                const uint32_t *__restrict__ cnt_open)
{
    const int r = blockIdx.x * kGeomBlock + threadIdx.x;
    if (r >= n_ranks) return;
    if (cnt_open && cnt_open[r] == 0u) return;   // this frame binned no instance of the rank: its screen-space row is zero (or was never written)
    const int i = (int)order[r];
    if (i < 0 || i >= f.P) return;               // (an exchange list entry nobody filled: -1)
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
    GaussIn<DEG, RAW> in;
    load_gaussian<DEG, RAW>(i, M, means, scales, rots, covpre, opac, shs, shs_rest, true, in);
    const float sg[9] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, s2.x};
    GeomGrad g;
    float dsh[3 * (DEG + 1) * (DEG + 1)];          // constant indices only: stays in registers
    unsigned clamp_bits = clamped[i];               // (recomputed for SH colours: see k_geom_bwd)
    if (shs && !has_colpre) { float rgb_[3]; sh_color_one<DEG>(f, cp, in.p, in.sh(), rgb_, clamp_bits); }
    geom_backward_one<DEG>(f, V, PV, cp, in.p, in.sc, in.q, (!RAW && covpre) ? in.cv : nullptr, in.sh(), has_colpre != 0, clamp_bits,
                           sg, g, dsh, shs != nullptr);
    if constexpr (RAW) activate_raw_backward(in.act, g);
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
    constexpr int K3 = 3 * (DEG + 1) * (DEG + 1);
    if constexpr (RAW == 1) {
        if (out.shs) { out.shs[3 * (size_t)i] = dsh[0]; out.shs[3 * (size_t)i + 1] = dsh[1]; out.shs[3 * (size_t)i + 2] = dsh[2]; }
        if (out.shs_rest) {
            float *dst = out.shs_rest + (size_t)i * (M - 1) * 3;
#pragma unroll
            for (int k = 3; k < K3; ++k) dst[k - 3] = dsh[k];
        }
    } else if (out.shs && shs) {
        float *dst = out.shs + (size_t)i * M * 3;
#pragma unroll
        for (int k = 0; k < K3; ++k)
            if (k < 3 * M) dst[k] = dsh[k];
    }
}

// ---- one launch that zero-fills up to nine output tensors (the sparse path's "memset"): the segments are laid end to
// end in a virtual float index space; each thread clears a float4 where the 16 bytes lie inside one segment.
__global__ __launch_bounds__(kGeomBlock) void k_zero_segments(ZeroSegs z)
{
    const unsigned block = blockIdx.x, blocks = gridDim.x;
    const size_t total = z.end[z.n - 1];
    const size_t stride = (size_t)blocks * kGeomBlock * 4;
    for (size_t v = ((size_t)block * kGeomBlock + threadIdx.x) * 4; v < total; v += stride) {
        int sgm = 0;
        while (v >= z.end[sgm]) ++sgm;
        const size_t off = v - (sgm ? z.end[sgm - 1] : 0);           // multiple of 4 inside the segment
        float *p = z.ptr[sgm] + off;
        if (off + 4 <= z.len[sgm] && (((uintptr_t)p) & 15) == 0) *reinterpret_cast<float4 *>(p) = make_float4(0.f, 0.f, 0.f, 0.f);
        else
            for (int e = 0; e < 4; ++e)
                if (off + e < z.len[sgm]) p[e] = 0.f;
    }
}

ZeroSegs zero_segments(const FrameK &f, const gsr_gaussians &g, float *screen, const gsr_grads &out, uint8_t *row_valid, size_t valid_bytes)
{
    const size_t P = (size_t)f.P;
    ZeroSegs z;
    z.n = 0;
    auto add = [&](float *ptr, size_t floats) {
        if (ptr && floats) { z.ptr[z.n] = ptr; z.len[z.n] = floats; z.end[z.n] = (z.n ? z.end[z.n - 1] : 0) + ((floats + 3) & ~(size_t)3); ++z.n; }
    };
    add(screen, P * kRowFloats);
    add(out.means3D, P * 3);
    add(out.means2D, P * 3);
    add(out.opacities, P);
    if (g.colors_precomp) add(out.colors_precomp, P * 3);
    if (!g.cov3D_precomp) { add(out.scales, P * 3); add(out.rotations, P * 4); }
    if (g.cov3D_precomp) add(out.cov3D_precomp, P * 6);
    const bool split = raw_split_sh(f, g);
    if (g.shs) add(out.shs, P * 3 * (size_t)(split ? 1 : f.M));
    if (split && f.M > 1) add(out.shs_rest, P * 3 * (size_t)(f.M - 1));
    if (row_valid && valid_bytes) add(reinterpret_cast<float *>(row_valid), (valid_bytes + 3) / 4);     // (a 256-byte aligned, padded block)
    return z;
}

// zero-fill of the backward's outputs in ONE launch: screen-space gradients (optional) + every wanted parameter gradient
int launch_zero_outputs(const FrameK &f, const gsr_gaussians &g, float *screen, const gsr_grads &out, hipStream_t s,
                        uint8_t *row_valid, size_t valid_bytes)
{
    return launch_zero_segments(zero_segments(f, g, screen, out, row_valid, valid_bytes), s);
}

int launch_zero_segments(const ZeroSegs &z, hipStream_t s)
{
    if (z.n == 0) return GSR_OK;
    const size_t total = z.end[z.n - 1];
    size_t blocks = (total / 4 + kGeomBlock - 1) / kGeomBlock / 4 + 1;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_zero_segments, dim3((unsigned)blocks), dim3(kGeomBlock), 0, s, z);
    GSR_LAUNCH_CHECK("zero_outputs", false, s);
    return GSR_OK;
}

int launch_geom_bwd(const FrameK &f, const gsr_camera &cam, const gsr_gaussians &g, const int32_t *radii, const GeomWS &gw,
                    const float *screen_grads, int g0, int g1, int n_ranks, const gsr_grads &out, bool debug, hipStream_t s,
                    const uint32_t *rows, bool own_frame_sparse)
{
    if (g1 <= g0) return GSR_OK;
    if (!rows) rows = gw.order;                  // the frame's own binned prefix
    if (n_ranks >= 0 && g0 == 0 && g1 == f.P && (own_frame_sparse || (long long)n_ranks * 4 < (long long)f.P)) {
        // depth-complex frame: almost every gradient row is zero -> memset the outputs, then visit the binned prefix only
        ProfileScope prof("geom_bwd", s);
        int rc0;
        if (!out.prezeroed && (rc0 = launch_zero_outputs(f, g, nullptr, out, s))) return rc0;
        if (n_ranks > 0) {
            const int sgrid = (n_ranks + kGeomBlock - 1) / kGeomBlock;
#define GSR_GS(DEG, RAW)                                                                                                     \
    hipLaunchKernelGGL((k_geom_bwd_sparse<DEG, RAW>), dim3(sgrid), dim3(kGeomBlock), 0, s, f, n_ranks, rows, cam.viewmatrix, \
                       cam.projmatrix, cam.campos, g.means3D, g.scales, g.rotations, g.cov3D_precomp, g.opacities, g.shs,      \
                       g.shs_rest, g.colors_precomp ? 1 : 0, radii, gw.clamped, reinterpret_cast<const float4 *>(screen_grads),   \
                       out, own_frame_sparse ? gw.cnt_open : nullptr)
            switch ((!g.raw ? 0 : (raw_split_sh(f, g) ? 1 : 2)) * 4 + (f.D > 3 ? 3 : (f.D < 0 ? 0 : f.D))) {
                case 0: GSR_GS(0, 0); break;  case 1: GSR_GS(1, 0); break;  case 2: GSR_GS(2, 0); break;  case 3: GSR_GS(3, 0); break;
                case 4: GSR_GS(0, 1); break;  case 5: GSR_GS(1, 1); break;  case 6: GSR_GS(2, 1); break;  case 7: GSR_GS(3, 1); break;
                case 8: GSR_GS(0, 2); break;  case 9: GSR_GS(1, 2); break;  case 10: GSR_GS(2, 2); break; default: GSR_GS(3, 2); break;
            }
#undef GSR_GS
        }
        GSR_LAUNCH_CHECK("geom_bwd_sparse", debug, s);
        return GSR_OK;
    }
    const int grid = (g1 - g0 + kGeomBlock - 1) / kGeomBlock;
    ProfileScope prof("geom_bwd", s);
#define GSR_GB(DEG, RAW)                                                                                                  \
    hipLaunchKernelGGL((k_geom_bwd<DEG, RAW>), dim3(grid), dim3(kGeomBlock), 0, s, f, g0, g1, cam.viewmatrix, cam.projmatrix, \
                       cam.campos, g.means3D, g.scales, g.rotations, g.cov3D_precomp, g.opacities, g.shs, g.shs_rest,         \
                       g.colors_precomp ? 1 : 0, radii, gw.clamped, reinterpret_cast<const float4 *>(screen_grads), out)
    switch ((!g.raw ? 0 : (raw_split_sh(f, g) ? 1 : 2)) * 4 + (f.D > 3 ? 3 : (f.D < 0 ? 0 : f.D))) {
        case 0: GSR_GB(0, 0); break;  case 1: GSR_GB(1, 0); break;  case 2: GSR_GB(2, 0); break;  case 3: GSR_GB(3, 0); break;
        case 4: GSR_GB(0, 1); break;  case 5: GSR_GB(1, 1); break;  case 6: GSR_GB(2, 1); break;  case 7: GSR_GB(3, 1); break;
        case 8: GSR_GB(0, 2); break;  case 9: GSR_GB(1, 2); break;  case 10: GSR_GB(2, 2); break; default: GSR_GB(3, 2); break;
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
