// gsr_render.hip — per-tile alpha blending, forward (K6) and backward (K7), plus the deterministic
// per-Gaussian reduction of the backward's instance rows.  Spec: SURVEY A.8 / A.9.
//
// MI355X mapping ("one wave, one tile"):
//   * a 16x16 binning tile is blended by ONE wave64; lane l owns column x = l & 15 and the four rows
//     y = (l >> 4) + 4k, k = 0..3 ("strips").  dx, A*dx^2 and B*dx are shared by a lane's four pixels,
//     no workgroup barrier exists anywhere in the blend loop, and early termination is a wave ballot.
//   * splat records (48 B: xy, conic, opacity, rgb) are gathered 64 at a time, one per lane, staged in
//     LDS and read back as wave-uniform broadcasts (ds_read_b128, conflict-free by construction).
//   * backward: each lane first sums a splat's nine partial gradients over its own four pixels in
//     registers, then ONE DPP row_shr / row_bcast reduction per value crosses the 64 lanes; the wave's
//     result goes to the splat's private 48-B row (no atomics).  Splats that no pixel of the tile
//     accepts skip the reduction (wave-uniform ballot).
//   * workgroup = one wave (64 threads): blockIdx -> tile goes through an XCD-aware bijective remap so
//     the 8 XCDs each walk a contiguous run of tiles and neighbouring tiles share an L2.
#include "gsr_internal.h"

namespace gsr {

__device__ __forceinline__ float fast_exp(float x)
{
    // v_exp_f32 computes 2^x; same function in forward and backward (SURVEY 7 "expf accuracy").
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Bijective XCD-aware remap: workgroups are dealt round-robin over the 8 XCDs (b % 8 names the group);
// give group x the contiguous run [start(x), start(x) + count(x)) of the n work items.
__device__ __forceinline__ int xcd_remap(int b, int n)
{
    const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
    return x * q + (x < r ? x : r) + i;
}

// dst = src + dpp_move(src) with out-of-range / masked-off lanes contributing 0.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false);
    return v + __int_as_float(t);
}

// Nine independent wave sums in ONE asm block, step-major: v_add_f32_dpp v, v, v <ctrl> computes
// v = dpp(v) + v (lanes whose DPP source does not exist are disabled and keep v, i.e. add 0).  The same
// register's next step is nine instructions later, which covers the VALU-write -> DPP-read wait states; the
// leading s_nop covers the instruction that produced the inputs.  Totals land in lane 63.
#define GSR_DPP9(ctrl)                                                                                   \
    "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                                   \
    "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                                   \
    "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                                   \
    "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                                   \
    "v_add_f32_dpp %4, %4, %4 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                                   \
    "v_add_f32_dpp %5, %5, %5 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                                   \
    "v_add_f32_dpp %6, %6, %6 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                                   \
    "v_add_f32_dpp %7, %7, %7 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                                   \
    "v_add_f32_dpp %8, %8, %8 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void wave_sum9_to_lane63(float &a, float &b, float &c, float &d, float &e, float &f_, float &g,
                                                    float &h, float &i)
{
    asm volatile("s_nop 1\n\t" GSR_DPP9("row_shr:1") GSR_DPP9("row_shr:2") GSR_DPP9("row_shr:4") GSR_DPP9("row_shr:8")
                     GSR_DPP9("row_bcast:15") GSR_DPP9("row_bcast:31") "s_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f_), "+v"(g), "+v"(h), "+v"(i));
}
#undef GSR_DPP9

// Sum over the 64 lanes; the total lands in lane 63 (other lanes hold partial sums).
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<0x111, 0xf, 0xf>(v);   // row_shr:1
    v = dpp_add<0x112, 0xf, 0xf>(v);   // row_shr:2
    v = dpp_add<0x114, 0xf, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf, 0xf>(v);   // row_shr:8   -> lane 15 of each row = row total
    v = dpp_add<0x142, 0xf, 0xf>(v);   // row_bcast:15: row r += lane 15 of row r-1 (row 0: no source -> +0)
    v = dpp_add<0x143, 0xf, 0xf>(v);   // row_bcast:31: rows 2,3 += lane 31          -> lane 63 = total
    return v;
}

__device__ __forceinline__ float bcast_lane63(float v)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

constexpr int kStrips = 4;
constexpr int kPairs = kStrips / 2;
typedef float v2f __attribute__((ext_vector_type(2)));      // packed-fp32 operand (two pixels of one lane)

// ------------------------------------------------------------------------------------------- K6
// One launch per depth chunk.  A tile's wave resumes the pixels' state (T, colour, last contributor) where
// the previous chunk left it, blends the chunk's range, and closes the tile once all 256 pixels have
// taken the cut-off.  State lives in the image workspace: T_state (negative = done), last_enc, and the
// un-finalised colour in out_color itself; the background term is added exactly once, when the tile
// closes or after the last chunk.
__global__ __launch_bounds__(kWave) void k_render_fwd(FrameK f, int n_tiles, int c, int finalize_all,
                                                      const uint2 *__restrict__ ranges_c, uint32_t *__restrict__ open,
                                                      const uint32_t *__restrict__ sorted_gid,
                                                      const float4 *__restrict__ records, const float *__restrict__ bg,
                                                      float *__restrict__ out_color, float *__restrict__ T_state,
                                                      int32_t *__restrict__ last_enc)
{
    __shared__ float4 sh_rec[kWave * 3];
    const int t = xcd_remap(blockIdx.x, n_tiles);
    const int tx = t % f.Gx, ty = f.ty0 + t / f.Gx;
    const int tile = ty * f.Gx + tx;
    if (open[tile] == 0u) return;                       // closed by an earlier chunk: pixels are final
    const uint2 rng = ranges_c[tile];
    const int lane = threadIdx.x;
    const int px = tx * GSR_TILE + (lane & 15);
    const int py0 = ty * GSR_TILE + (lane >> 4);
    const float fx = (float)px;
    const size_t N = (size_t)f.W * f.H;

    // Per pixel: Tl = live transmittance (0 once the pixel has taken the cut-off), Tf = transmittance to report
    // (frozen at the cut-off), colour, last contributor.  A rejected splat runs the same arithmetic with
    // alpha = 0, which leaves everything unchanged, so the only selects are on alpha, on the stop decision and
    // on the contributor index.
    float fy[kStrips], Tl[kStrips], Tf[kStrips], Cr[kStrips], Cg[kStrips], Cb[kStrips];
    int last[kStrips];
#pragma unroll
    for (int k = 0; k < kStrips; ++k) {
        const int py = py0 + 4 * k;
        fy[k] = (float)py;
        const bool inside = px < f.W && py < f.H;
        Tf[k] = 1.f; Cr[k] = Cg[k] = Cb[k] = 0.f; last[k] = 0;
        Tl[k] = inside ? 1.f : 0.f;
        if (c > 0 && inside) {
            const size_t pix = (size_t)py * f.W + px;
            const float ts = T_state[pix];
            Tf[k] = fabsf(ts);
            Tl[k] = ts < 0.f ? 0.f : ts;
            Cr[k] = out_color[pix]; Cg[k] = out_color[N + pix]; Cb[k] = out_color[2 * N + pix];
            last[k] = last_enc[pix];
        }
    }

    const int n_total = (int)(rng.y - rng.x);
    const int enc_base = (c + 1) << kLastShift;
    for (int base = 0; base < n_total; base += kWave) {
        const bool any_live = (Tl[0] != 0.f) || (Tl[1] != 0.f) || (Tl[2] != 0.f) || (Tl[3] != 0.f);
        if (__ballot(any_live) == 0ull) break;
        const int n = min(kWave, n_total - base);
        __syncthreads();
        if (lane < n) {
            const uint32_t gid = sorted_gid[rng.x + base + lane];
            const float4 *r = records + 3 * (size_t)gid;
            sh_rec[3 * lane + 0] = r[0];
            sh_rec[3 * lane + 1] = r[1];
            sh_rec[3 * lane + 2] = r[2];
        }
        __syncthreads();
        // Strip pairs (rows 0-7 and 8-15 of the tile) whose 128 pixels have all taken the cut-off are skipped for
        // the whole batch: a wave-uniform flag per pair, evaluated once per 64 splats.
        bool pair_live0 = __ballot((Tl[0] != 0.f) || (Tl[1] != 0.f)) != 0ull;
        bool pair_live1 = __ballot((Tl[2] != 0.f) || (Tl[3] != 0.f)) != 0ull;
        for (int j = 0; j < n; ++j) {
            if ((j & 7) == 0 && j) {            // every 8 splats: a tile that saturates mid-batch stops there, not 30 splats later
                pair_live0 = pair_live0 && __ballot((Tl[0] != 0.f) || (Tl[1] != 0.f)) != 0ull;
                pair_live1 = pair_live1 && __ballot((Tl[2] != 0.f) || (Tl[3] != 0.f)) != 0ull;
                if (!pair_live0 && !pair_live1) break;
            }
            const float4 a = sh_rec[3 * j], b = sh_rec[3 * j + 1];
            const float cb = sh_rec[3 * j + 2].x;
            const float dx = a.x - fx;
            const float axx = a.z * dx * dx + b.y, bx = a.w * dx;         // record is pre-scaled: p = log2(op exp(power))
            const int contributor = enc_base | (base + j + 1);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (!(p == 0 ? pair_live0 : pair_live1)) continue;
#pragma unroll
                for (int k = 2 * p; k < 2 * p + 2; ++k) {
                    const float dy = a.y - fy[k];
                    const float lp = (b.x * dy + bx) * dy + axx;
                    const float alpha = fminf((float)GSR_ALPHA_MAX, __builtin_amdgcn_exp2f(lp));
                    const bool keep = !(lp > b.y) && !(alpha < (float)GSR_ALPHA_MIN);       // power > 0  <=>  lp > lop
                    const float ae = keep ? alpha : 0.f;
                    const float test_T = Tl[k] * (1.f - ae);              // == Tl when rejected, 0 when already done
                    const bool stop = test_T < (float)GSR_T_CUTOFF;        // live + accepted + below the cut-off, or done
                    const float w = stop ? 0.f : ae * Tl[k];               // the stopping splat is NOT composited (A.8)
                    Cr[k] += b.z * w; Cg[k] += b.w * w; Cb[k] += cb * w;
                    Tf[k] = stop ? Tf[k] : test_T;
                    Tl[k] = stop ? 0.f : test_T;
                    last[k] = (keep && !stop) ? contributor : last[k];
                }
            }
        }
    }
    const bool any_live = (Tl[0] != 0.f) || (Tl[1] != 0.f) || (Tl[2] != 0.f) || (Tl[3] != 0.f);
    const bool closing = __ballot(any_live) == 0ull;
    const bool finalize = closing || finalize_all != 0;
    const float bg0 = finalize ? bg[0] : 0.f, bg1 = finalize ? bg[1] : 0.f, bg2 = finalize ? bg[2] : 0.f;
#pragma unroll
    for (int k = 0; k < kStrips; ++k) {
        const int py = py0 + 4 * k;
        if (px < f.W && py < f.H) {
            const size_t pix = (size_t)py * f.W + px;
            out_color[pix] = Cr[k] + Tf[k] * bg0;
            out_color[N + pix] = Cg[k] + Tf[k] * bg1;
            out_color[2 * N + pix] = Cb[k] + Tf[k] * bg2;
            T_state[pix] = Tl[k] == 0.f ? -Tf[k] : Tf[k];
            last_enc[pix] = last[k];
        }
    }
    if (lane == 0) open[tile] = closing ? 0u : 1u;
}

int launch_render_fwd(const FrameK &f, const gsr_camera &cam, int c, bool last_chunk, const GeomWS &gw, const BinningWS &bw,
                      ImageWS &iw, float *out_color, bool debug, hipStream_t s)
{
    const int n_tiles = (f.ty1 - f.ty0) * f.Gx;
    if (n_tiles <= 0) return GSR_OK;
    ProfileScope prof("render_fwd", s);
    const size_t Tn = (size_t)f.Gx * f.Gy;
    hipLaunchKernelGGL(k_render_fwd, dim3(n_tiles), dim3(kWave), 0, s, f, n_tiles, c, last_chunk ? 1 : 0,
                       iw.ranges + (size_t)c * Tn, iw.open, bw.sorted_gid, gw.records, cam.bg, out_color, iw.T_state,
                       iw.last_enc);
    GSR_LAUNCH_CHECK("render_fwd", debug, s);
    return GSR_OK;
}

// ------------------------------------------------------------------------------------------- K7
// One launch for the whole frame: a tile's wave walks its chunks last to first, each chunk's range back to
// front.  A pixel takes part in chunk c up to its own last contributor (all of the range for chunks before
// the one that holds it, nothing after).
__global__ __launch_bounds__(kWave, 4) void k_render_bwd(FrameK f, int n_tiles, int chunks_run, const uint2 *__restrict__ ranges,
                                                      const uint32_t *__restrict__ sorted_gid,
                                                      const uint32_t *__restrict__ sorted_slot,
                                                      const float4 *__restrict__ records, const float *__restrict__ bg,
                                                      const float *__restrict__ T_state, const int32_t *__restrict__ last_enc,
                                                      const float *__restrict__ dL_dpix, float4 *__restrict__ grad_rows)
{
    __shared__ float4 sh_rec[kWave * 3];
    const int t = xcd_remap(blockIdx.x, n_tiles);
    const int tx = t % f.Gx, ty = f.ty0 + t / f.Gx;
    const int tile = ty * f.Gx + tx;
    const size_t Tn = (size_t)f.Gx * f.Gy;
    const int lane = threadIdx.x;
    const int px = tx * GSR_TILE + (lane & 15);
    const int py0 = ty * GSR_TILE + (lane >> 4);
    const float fx = (float)px;
    const size_t N = (size_t)f.W * f.H;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

    // Per-pixel state in PAIRS of strips (k = 2p, 2p + 1) as 2-vectors: the arithmetic below then compiles to packed
    // fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), two pixels per VALU issue.
    v2f fy[kPairs], T[kPairs], bgterm[kPairs], dpr[kPairs], dpg[kPairs], dpb[kPairs];
    v2f ar[kPairs], ag[kPairs], ab[kPairs];                    // colour behind the current splat, per pixel
    int c_last[kStrips], n_last[kStrips];
#pragma unroll
    for (int k = 0; k < kStrips; ++k) {
        const int py = py0 + 4 * k;
        const bool inside = px < f.W && py < f.H;
        const size_t pix = inside ? (size_t)py * f.W + px : 0;
        const float Tk = inside ? fabsf(T_state[pix]) : 0.f;
        const int enc = inside ? last_enc[pix] : 0;
        c_last[k] = (enc >> kLastShift) - 1;                              // -1: no contributor at all
        n_last[k] = enc & ((1 << kLastShift) - 1);
        const float r_ = inside ? dL_dpix[pix] : 0.f, g_ = inside ? dL_dpix[N + pix] : 0.f, b_ = inside ? dL_dpix[2 * N + pix] : 0.f;
        fy[k >> 1][k & 1] = (float)py;
        T[k >> 1][k & 1] = Tk;
        dpr[k >> 1][k & 1] = r_; dpg[k >> 1][k & 1] = g_; dpb[k >> 1][k & 1] = b_;
        bgterm[k >> 1][k & 1] = -Tk * (bg0 * r_ + bg1 * g_ + bg2 * b_);     // -T_final * <bg, dL/dpix>
    }
#pragma unroll
    for (int p = 0; p < kPairs; ++p) { ar[p] = v2f{0.f, 0.f}; ag[p] = ar[p]; ab[p] = ar[p]; }
    const float half_w = 0.5f * (float)f.W, half_h = 0.5f * (float)f.H;

    for (int c = chunks_run - 1; c >= 0; --c) {
        const uint2 rng = ranges[(size_t)c * Tn + tile];
        const int n_total = (int)(rng.y - rng.x);
        if (n_total == 0) continue;
        int limit[kStrips];
        int max_contrib = 0;
#pragma unroll
        for (int k = 0; k < kStrips; ++k) {
            limit[k] = c < c_last[k] ? n_total : (c == c_last[k] ? n_last[k] : 0);
            max_contrib = max(max_contrib, limit[k]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) max_contrib = max(max_contrib, __shfl_xor(max_contrib, off));

        const int n_batches = (n_total + kWave - 1) / kWave;
        for (int bi = n_batches - 1; bi >= 0; --bi) {
            const int base = bi * kWave;
            const int n = min(kWave, n_total - base);
            uint32_t slot = 0;
            if (lane < n) slot = sorted_slot[rng.x + base + lane];
            unsigned long long written = 0ull;                 // wave-uniform: bit j = splat j's row stored
            if (base < max_contrib) {
                __syncthreads();
                if (lane < n) {
                    const uint32_t gid = sorted_gid[rng.x + base + lane];
                    const float4 *r = records + 3 * (size_t)gid;
                    sh_rec[3 * lane + 0] = r[0];
                    sh_rec[3 * lane + 1] = r[1];
                    sh_rec[3 * lane + 2] = r[2];
                }
                __syncthreads();
                for (int j = n - 1; j >= 0; --j) {
                    const int pos = base + j;                 // contributor index of this splat is pos + 1
                    if (pos >= max_contrib) continue;         // wave-uniform
                    const float4 a = sh_rec[3 * j], b = sh_rec[3 * j + 1];
                    const float cb = sh_rec[3 * j + 2].x;
                    const float dx = a.x - fx;
                    const float axx = a.z * dx * dx + b.y, bx = a.w * dx;       // pre-scaled record: lp = log2(op exp(power))
                    float cA, cB, cC, opac;                                      // the unscaled conic / opacity (per splat)
                    unscale_conic(a.z, a.w, b.x, b.y, cA, cB, cC, opac);
                    const float inv_op = fast_rcp(opac);
                    v2f S0 = {0.f, 0.f}, S1 = S0, S2 = S0, S3 = S0, S4 = S0, S5 = S0, S6 = S0, S7 = S0, S8 = S0;
                    bool any_valid = false;
#pragma unroll
                    for (int p = 0; p < kPairs; ++p) {
                        const v2f dy = a.y - fy[p];
                        const v2f lp = (b.x * dy + bx) * dy + axx;
                        v2f araw, ae, Ge;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int k = 2 * p + e;
                            araw[e] = __builtin_amdgcn_exp2f(fminf(lp[e], b.y));     // = op * G; power > 0 lanes are rejected
                            const float alpha = fminf((float)GSR_ALPHA_MAX, araw[e]);
                            const bool valid = (pos < limit[k]) && !(lp[e] > b.y) && !(alpha < (float)GSR_ALPHA_MIN);
                            any_valid = any_valid || valid;
                            // Rejected pixels run the same arithmetic with alpha = 0 and G = 0: T, the colour behind and
                            // every partial sum then stay exactly unchanged, so only these two values need a select.
                            ae[e] = valid ? alpha : 0.f;
                            Ge[e] = valid ? araw[e] * inv_op : 0.f;
                        }
                        const v2f one_m = 1.f - ae;
                        const v2f inv1ma = {fast_rcp(one_m[0]), fast_rcp(one_m[1])};
                        const v2f Tn_ = T[p] * inv1ma;                        // T before this splat
                        const v2f w = ae * Tn_;                               // d colour / d rgb
                        // colour behind this splat (A.9's accum_rec), updated as soon as the splat is processed:
                        // B <- alpha c + (1 - alpha) B
                        const v2f dr = b.z - ar[p], dg = b.w - ag[p], db = cb - ab[p];
                        v2f dL_dalpha = dr * dpr[p] + dg * dpg[p] + db * dpb[p];
                        ar[p] += ae * dr; ag[p] += ae * dg; ab[p] += ae * db;
                        dL_dalpha = dL_dalpha * Tn_ + bgterm[p] * inv1ma;
                        const v2f gdx = Ge * dx, gdy = Ge * dy;
                        const v2f tG = opac * dL_dalpha;                      // dL/dG (times G through gdx, gdy)
                        const v2f tx = tG * gdx, ty = tG * gdy;
                        S0 -= tx * cA + ty * cB;                              // dL/dG * dG/ddelx
                        S1 -= ty * cC + tx * cB;
                        S2 += tx * dx;
                        S3 += tx * dy;
                        S4 += ty * dy;
                        S5 += Ge * dL_dalpha;
                        S6 += w * dpr[p]; S7 += w * dpg[p]; S8 += w * dpb[p];
                        T[p] = Tn_;
                    }
                    float s0 = S0[0] + S0[1], s1 = S1[0] + S1[1], s2 = S2[0] + S2[1], s3 = S3[0] + S3[1], s4 = S4[0] + S4[1],
                          s5 = S5[0] + S5[1], s6 = S6[0] + S6[1], s7 = S7[0] + S7[1], s8 = S8[0] + S8[1];
                    if (__ballot(any_valid) == 0ull) continue;               // nobody accepted this splat: row stays 0
                    wave_sum9_to_lane63(s0, s1, s2, s3, s4, s5, s6, s7, s8);
                    // lane 63 holds the nine totals: it stores the splat's row itself (three 16-B stores)
                    const uint32_t slot_j = (uint32_t)__builtin_amdgcn_readlane((int)slot, j);
                    written |= 1ull << j;
                    if (lane == kWave - 1) {
                        float4 *row = grad_rows + 3 * (size_t)slot_j;
                        row[0] = make_float4(s0 * half_w, s1 * half_h, s2 * -0.5f, s3 * -0.5f);
                        row[1] = make_float4(s4 * -0.5f, s5, s6, s7);
                        row[2] = make_float4(s8, 0.f, 0.f, 0.f);
                    }
                }
            }
            // rows of splats that were never reduced (past every pixel's last contributor, or accepted by nobody)
            if (lane < n && !((written >> lane) & 1ull)) {
                float4 *row = grad_rows + 3 * (size_t)slot;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                row[0] = z; row[1] = z; row[2] = z;
            }
        }
    }
}

int launch_render_bwd(const FrameK &f, const gsr_camera &cam, int chunks_run, int sort_result, const GeomWS &gw, BinningWS &bw,
                      const ImageWS &iw, const float *dL_dcolor, bool debug, hipStream_t s)
{
    const int n_tiles = (f.ty1 - f.ty0) * f.Gx;
    if (n_tiles <= 0 || chunks_run <= 0) return GSR_OK;
    ProfileScope prof("render_bwd", s);
    hipLaunchKernelGGL(k_render_bwd, dim3(n_tiles), dim3(kWave), 0, s, f, n_tiles, chunks_run, iw.ranges, bw.sorted_gid,
                       bw.vals[sort_result], gw.records, cam.bg, iw.T_state, iw.last_enc, dL_dcolor,
                       reinterpret_cast<float4 *>(bw.grad_rows));
    GSR_LAUNCH_CHECK("render_bwd", debug, s);
    return GSR_OK;
}

// ---- per-Gaussian reduction of the instance rows (bitwise reproducible: fixed lane assignment and a fixed
// shuffle tree).  EIGHT lanes cooperate on one depth rank (8 ranks per wave): lane i of the group sums rows
// i, i+8, ... and three xor-shuffles combine the group.  The near, screen-filling Gaussians own hundreds of
// rows each; one thread per Gaussian left a tail of a few thousand threads walking them serially.
// Only ranks of chunks that actually ran can own rows; every other Gaussian's gradient row is zero (memset).
constexpr int kRedBlock = 256;
template <int kRedGroup>
__global__ __launch_bounds__(kRedBlock) void k_reduce_rows(int n_ranks, const uint32_t *__restrict__ order,
                                                           const uint32_t *__restrict__ cnt_open,
                                                           const uint32_t *__restrict__ row_begin,
                                                           const float4 *__restrict__ grad_rows, float4 *__restrict__ screen)
{
    const int sub = threadIdx.x & (kRedGroup - 1);
    const int r = (blockIdx.x * kRedBlock + threadIdx.x) / kRedGroup;
    const bool live = r < n_ranks;
    const uint32_t cnt = live ? cnt_open[r] : 0u;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    float a8 = 0.f;
    if (cnt) {
        const uint32_t begin = row_begin[r];
        for (uint32_t sl = begin + sub; sl < begin + cnt; sl += kRedGroup) {
            const float4 r0 = grad_rows[3 * (size_t)sl], r1 = grad_rows[3 * (size_t)sl + 1];
            const float r2 = grad_rows[3 * (size_t)sl + 2].x;
            a0.x += r0.x; a0.y += r0.y; a0.z += r0.z; a0.w += r0.w;
            a1.x += r1.x; a1.y += r1.y; a1.z += r1.z; a1.w += r1.w;
            a8 += r2;
        }
    }
#pragma unroll
    for (int off = kRedGroup / 2; off >= 1; off >>= 1) {
        a0.x += __shfl_xor(a0.x, off); a0.y += __shfl_xor(a0.y, off); a0.z += __shfl_xor(a0.z, off); a0.w += __shfl_xor(a0.w, off);
        a1.x += __shfl_xor(a1.x, off); a1.y += __shfl_xor(a1.y, off); a1.z += __shfl_xor(a1.z, off); a1.w += __shfl_xor(a1.w, off);
        a8 += __shfl_xor(a8, off);
    }
    if (live && cnt && sub < 3) {
        const uint32_t g = order[r];
        screen[3 * (size_t)g + sub] = sub == 0 ? a0 : (sub == 1 ? a1 : make_float4(a8, 0.f, 0.f, 0.f));
    }
}

int launch_reduce_rows(const FrameK &f, int n_ranks, long long rows_upper, const GeomWS &gw, const BinningWS &bw,
                       float *screen_grads, bool prezeroed, bool debug, hipStream_t s)
{
    if (f.P == 0) return GSR_OK;
    ProfileScope prof("reduce_rows", s);
    if (!prezeroed) GSR_HIP_CHECK(hipMemsetAsync(screen_grads, 0, (size_t)f.P * kRowFloats * sizeof(float), s));
    if (n_ranks > 0) {
        // lanes per Gaussian: a whole wave when the processed Gaussians own many rows each (depth-complex scenes:
        // a few thousand screen-filling splats), eight otherwise
        const bool wide = rows_upper / (long long)n_ranks >= 48;
        const long long threads = (long long)n_ranks * (wide ? 64 : 8);
        const dim3 grid((unsigned)((threads + kRedBlock - 1) / kRedBlock));
        if (wide)
            hipLaunchKernelGGL(k_reduce_rows<64>, grid, dim3(kRedBlock), 0, s, n_ranks, gw.order, gw.cnt_open, gw.row_begin,
                               reinterpret_cast<const float4 *>(bw.grad_rows), reinterpret_cast<float4 *>(screen_grads));
        else
            hipLaunchKernelGGL(k_reduce_rows<8>, grid, dim3(kRedBlock), 0, s, n_ranks, gw.order, gw.cnt_open, gw.row_begin,
                               reinterpret_cast<const float4 *>(bw.grad_rows), reinterpret_cast<float4 *>(screen_grads));
    }
    GSR_LAUNCH_CHECK("reduce_rows", debug, s);
    return GSR_OK;
}

}  // namespace gsr
