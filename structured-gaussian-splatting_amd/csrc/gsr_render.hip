// gsr_render.hip — per-tile alpha blending, forward (K6) and backward (K7), plus the deterministic
// per-Gaussian reduction of the backward's instance rows.  Spec: SURVEY A.8 / A.9.
//
// MI355X mapping ("one wave, one tile"):
//   * forward: a 16x16 binning tile is blended by ONE wave64.  Chunks of large splats (k_render_fwd): lane l owns the pixel
//     (l & 7, l >> 3) of each of the tile's four 8x8 QUADRANTS, so "can this splat reach quadrant k at all" is wave-uniform and
//     clear quadrants are skipped with scalar branches (sub-tile culling); early termination is a wave ballot per quadrant.
//     Chunks of small splats (k_render_fwd_groups): the backward's mapping, one 16-lane group per quadrant, each walking its own
//     entries.  The two render the same bits.  No workgroup barrier exists anywhere in the blend loops.
//   * splat records (48 B: xy, conic, opacity, rgb) are gathered 64 at a time, one per lane, staged in
//     LDS and read back as broadcasts (ds_read_b128).
//   * backward: FRONT TO BACK, in independent work units of kSeg list entries (gsr_internal.h: kSeg, UnitLists) that start from the
//     per-pixel state the forward left at the segment boundary.  The wave is four GROUPS of 16 lanes, one per quadrant, each
//     walking the entries that reach ITS quadrant: up to four different splats per pass, one DPP butterfly over the rows of 16
//     lanes for all four (k_render_bwd).  The sums of a batch of 64 entries meet in LDS and leave as whole 48-B rows, one per
//     list entry (no global atomics; one valid byte per row says whether it was written).
//   * workgroup = one wave (64 threads); forward: tile = block (consecutive tiles run on different XCDs).
#include "gsr_internal.h"

namespace gsr {

__device__ __forceinline__ float fast_exp(float x)
{
    // v_exp_f32 computes 2^x; same function in forward and backward (SURVEY 7 "expf accuracy").
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Nine sums over each ROW of 16 lanes, all four rows at once (k_render_bwd's quadrant groups): a transposing butterfly.  The first
// two steps fold two values per instruction (bank_mask picks the half / the quads that keep a value), the two steps inside a quad
// run on the three registers that are left: 20 DPP adds for four splats.  On return every lane of a quad holds, with b = the quad's
// number in the row:   a[0]: value (0, 1, 4, 5)[b]     a[2]: value (2, 3, 6, 7)[b]     a[8]: value 8
// DPP hazard (a VALU write followed by a DPP read of the same register needs two wait states) is covered by the order and s_nops.
__device__ __forceinline__ void row_sum9_transpose(float (&a)[9])
{
#define GSR_DPP(d, s_, ctrl, bank) "v_add_f32_dpp %" #d ", %" #s_ ", %" #s_ " " ctrl " row_mask:0xf bank_mask:" bank "\n\t"
    asm volatile("s_nop 1\n\t"
                 GSR_DPP(0, 0, "row_ror:8", "0x3") GSR_DPP(1, 1, "row_ror:8", "0x3") GSR_DPP(2, 2, "row_ror:8", "0x3")
                 GSR_DPP(3, 3, "row_ror:8", "0x3") GSR_DPP(8, 8, "row_ror:8", "0xf")
                 GSR_DPP(0, 4, "row_ror:8", "0xc") GSR_DPP(1, 5, "row_ror:8", "0xc") GSR_DPP(2, 6, "row_ror:8", "0xc")
                 GSR_DPP(3, 7, "row_ror:8", "0xc")
                 "s_nop 1\n\t"
                 GSR_DPP(0, 0, "row_half_mirror", "0x5") GSR_DPP(2, 2, "row_half_mirror", "0x5") GSR_DPP(8, 8, "row_half_mirror", "0xf")
                 GSR_DPP(0, 1, "row_half_mirror", "0xa") GSR_DPP(2, 3, "row_half_mirror", "0xa")
                 "s_nop 1\n\t"
                 GSR_DPP(0, 0, "quad_perm:[1,0,3,2]", "0xf") GSR_DPP(2, 2, "quad_perm:[1,0,3,2]", "0xf")
                 GSR_DPP(8, 8, "quad_perm:[1,0,3,2]", "0xf")
                 "s_nop 0\n\t"
                 GSR_DPP(0, 0, "quad_perm:[2,3,0,1]", "0xf") GSR_DPP(2, 2, "quad_perm:[2,3,0,1]", "0xf")
                 GSR_DPP(8, 8, "quad_perm:[2,3,0,1]", "0xf")
                 "s_nop 1"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]));
#undef GSR_DPP
}

__device__ __forceinline__ int wave_max_uniform(int v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off));
    return __builtin_amdgcn_readfirstlane(v);
}

#ifndef GSR_FWD_WAVES
#define GSR_FWD_WAVES 5        // resident waves per SIMD the register allocation of k_render_fwd aims for
#endif
#ifndef GSR_BWD_WAVES
#define GSR_BWD_WAVES 4        // ... and of k_render_bwd
#endif
typedef float v2f __attribute__((ext_vector_type(2)));      // packed-fp32 operand: the lane's two pixels of a pair

// Pixel mapping of k_render_fwd ("one wave, one tile", QUADRANT-major; the group kernels have their own, see k_render_bwd): lane l owns
// the pixel (l & 7, l >> 3) of each of the tile's four 8x8 quadrants.  Whether a splat can reach a quadrant at all (quadrant_mask_q: the
// exact alpha >= 1/255 bound on the 8x8 pixel rectangle) is then WAVE-UNIFORM: the 4-bit mask travels with the tile list's entry, and the
// blend loop skips clear quadrants with scalar branches.  A 2-3 px splat reaches one or two quadrants of a tile, not four.
// The two quadrants of a pair share dy and differ by 8 in dx: they run as packed fp32 (v_pk_fma_f32 issues two FMAs in the cycles
// of 1.2 plain ones on gfx950: profiles/valu_microbench).
__device__ __forceinline__ int lane_px(int lane) { return lane & 7; }
__device__ __forceinline__ int lane_py(int lane) { return lane >> 3; }

// lp = log2(opacity exp(power)) of a pair of pixels from the pre-scaled record (qA, qB, qC = a.z, a.w, b.x; log2 opacity = b.y):
//     lp = qA dx^2 + qB dx dy + qC dy^2 + lop = ((qC dy + qB dx) dy) + (qA dx) dx + lop
// ONE definition for the three blend kernels, with every rounding spelled out (no contraction left to the compiler): the forward
// kernels and the backward then agree on every pixel's lp - and with it on alpha, on the accept / stop decisions and on the
// transmittance - bit for bit, whichever kernel rendered the frame.
struct LpTerms { v2f axx, bx; };
__device__ __forceinline__ LpTerms lp_terms(float qA, float qB, float lop, v2f dx)
{
#pragma clang fp contract(off)
    LpTerms t;
    const v2f adx = qA * dx;                                               // rounded
    t.axx = __builtin_elementwise_fma(adx, dx, v2f{lop, lop});             // (qA dx) dx + lop
    t.bx = qB * dx;                                                        // rounded
    return t;
}
__device__ __forceinline__ v2f lp_at(const LpTerms &t, float qC, float dy)
{
#pragma clang fp contract(off)
    const v2f s = __builtin_elementwise_fma(v2f{qC, qC}, v2f{dy, dy}, t.bx);        // qC dy + qB dx
    return __builtin_elementwise_fma(s, v2f{dy, dy}, t.axx);
}

// Stage up to 64 records (one per lane) of a tile's list into LDS; returns the lane's quadrant mask (bits 28..31 of the
// sorted list's entry, put there by the emit kernels: gsr_binning.hip).
__device__ __forceinline__ unsigned stage_batch(float4 *sh_rec, int lane, int n, const uint32_t *__restrict__ sorted_gid, uint32_t first,
                                                const float4 *__restrict__ records)
{
    unsigned m = 0;
    if (lane < n) {
        const uint32_t v = sorted_gid[first + lane];
        m = v >> kQuadMaskShift;
        const float4 *r = records + 3 * (size_t)(v & kGidMask);
        sh_rec[3 * lane + 0] = r[0];
        sh_rec[3 * lane + 1] = r[1];
        sh_rec[3 * lane + 2] = r[2];
    }
    return m;
}

// ------------------------------------------------------------------------------------------- K6
// One launch per depth chunk.  A tile's wave resumes the pixels' state (T, colour, last contributor) where
// the previous chunk left it, blends the chunk's range, and closes the tile once all 256 pixels have
// taken the cut-off.  State lives in the image workspace: T_state (negative = done), last_enc, and the
// un-finalised colour in out_color itself; the background term is added exactly once, when the tile
// closes or after the last chunk.
//
// The alpha >= 1/255 test (A.8) is taken on the exponent: lp = log2(opacity exp(power)) < log2(1/255).  Forward and backward use
// the same constant on the same lp, so they agree on every pixel; against alpha = opacity * exp(power) < 1/255 in exact
// arithmetic the two rules differ only where alpha sits within rounding of the threshold (the oracle's fragile band).
constexpr float kLog2AlphaMin = -7.99435343685886f;      // log2(1 / 255)

// Per pixel: T (signed, see FwdPair), colour, last contributor.  A rejected splat runs the same arithmetic with alpha = 0, which leaves
// everything unchanged, so the only selects are on alpha, on the stop decision and on the contributor index.
struct FwdPair {             // state of a lane's two pixels in one pair of quadrants (left, right)
    v2f T, Cr, Cg, Cb;       // T > 0: live transmittance; T < 0: the pixel has taken the cut-off (or lies outside the image), |T| = the
    int last0, last1;        // transmittance to report (frozen in front of the stopping splat): exactly what T_state holds
};

// One register per pixel carries both "live transmittance" and "final transmittance of a done pixel": for a done pixel
// test_T = T (1 - alpha) is negative, so the stop test fires by itself, nothing is composited and T keeps its value.
// ONE definition for both forward kernels (active: the lane's group has a splat in this pass - always, in the lock-step kernel).
__device__ __forceinline__ void fwd_pair(bool active, v2f lp, float lop, float cr, float cg, float cb, int contributor, FwdPair &P)
{
    const bool keep0 = active && !(lp[0] > lop) && !(lp[0] < kLog2AlphaMin);      // power > 0  <=>  lp > lop;  alpha < 1/255  <=>  lp < log2(1/255)
    const bool keep1 = active && !(lp[1] > lop) && !(lp[1] < kLog2AlphaMin);
    // a rejected pixel gets lp = -inf: exp2 gives alpha = 0 by itself (one select per pixel, in front of the exponential)
    const v2f ae = {fminf((float)GSR_ALPHA_MAX, __builtin_amdgcn_exp2f(keep0 ? lp[0] : -INFINITY)),
                    fminf((float)GSR_ALPHA_MAX, __builtin_amdgcn_exp2f(keep1 ? lp[1] : -INFINITY))};
    const v2f test_T = P.T * (1.f - ae);                       // == T when rejected, negative when already done
    const v2f aT = ae * P.T;
    const bool stop0 = test_T[0] < (float)GSR_T_CUTOFF;        // live + accepted + below the cut-off, or done
    const bool stop1 = test_T[1] < (float)GSR_T_CUTOFF;
    const v2f w = {stop0 ? 0.f : aT[0], stop1 ? 0.f : aT[1]};  // the stopping splat is NOT composited (A.8)
    P.Cr = __builtin_elementwise_fma(v2f{cr, cr}, w, P.Cr);
    P.Cg = __builtin_elementwise_fma(v2f{cg, cg}, w, P.Cg);
    P.Cb = __builtin_elementwise_fma(v2f{cb, cb}, w, P.Cb);
    P.T = v2f{stop0 ? -fabsf(P.T[0]) : test_T[0], stop1 ? -fabsf(P.T[1]) : test_T[1]};
    P.last0 = (keep0 && !stop0) ? contributor : P.last0;
    P.last1 = (keep1 && !stop1) ? contributor : P.last1;
}

// wave-uniform 4-bit mask: quadrant k still has a live pixel
__device__ __forceinline__ unsigned live_quadrants(const FwdPair &A, const FwdPair &B)
{
    unsigned m = 0;
    if (__ballot(A.T[0] > 0.f) != 0ull) m |= 1u;
    if (__ballot(A.T[1] > 0.f) != 0ull) m |= 2u;
    if (__ballot(B.T[0] > 0.f) != 0ull) m |= 4u;
    if (__ballot(B.T[1] > 0.f) != 0ull) m |= 8u;
    return m;
}

#if defined(GSR_BWD_TRACE) || defined(GSR_FWD_TRACE)
// debug builds only (tools/bwd_trace.sh): per block {start, end (100 MHz clock), HW_ID, XCC_ID} of the last launch of the
// traced kernel
__device__ unsigned long long g_bwd_trace[4 * 65536];
extern "C" int gsr_debug_bwd_trace(unsigned long long *host_out, int n_blocks)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_bwd_trace), sizeof(unsigned long long) * 4 * (size_t)n_blocks);
}
struct TraceEnd {
    unsigned long long t0; int b;
    __device__ ~TraceEnd() {
        if (threadIdx.x == 0 && b < 65536) {
            g_bwd_trace[4 * b] = t0; g_bwd_trace[4 * b + 1] = wall_clock64();
            g_bwd_trace[4 * b + 2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
            g_bwd_trace[4 * b + 3] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
        }
    }
};
#endif

// the end of a (tile, chunk)'s blend: the tile's open flag, how deep the backward will walk, and the backward's work units
__device__ __forceinline__ void fwd_tile_epilogue(int t, int tile, int c, int lane, bool closing, bool stuck, int walked, uint32_t *__restrict__ open,
                                                  uint32_t *__restrict__ tile_walk_c, const UnitLists &units, uint32_t *__restrict__ unit_count)
{
    if (lane == 0) {
        open[tile] = closing ? 0u : (stuck ? 2u : 1u);
        tile_walk_c[tile] = (uint32_t)walked;
    }
    // the blend backward's work units of this (tile, chunk): one per kSeg walked entries, appended to the lists of the tile's shard
    // (full segments; the last, partial one by its length class)
    const uint32_t n_full = (uint32_t)walked / kSeg, rest = (uint32_t)walked - n_full * kSeg;
    if (walked > 0) {
        const int shard = t & (kUnitShards - 1);
        const uint32_t head = (uint32_t)tile | ((uint32_t)c << kUnitTileBits);
        const int cls = unit_class(rest);
        uint32_t at_full = 0, at_part = 0;
        if (lane == 0) {
            if (n_full) at_full = atomicAdd(&unit_count[shard * kUnitClasses], n_full);
            if (rest) at_part = atomicAdd(&unit_count[shard * kUnitClasses + cls], 1u);
        }
        at_full = (uint32_t)__builtin_amdgcn_readfirstlane((int)at_full);
        uint2 *full = units.units + units.list_begin(shard, 0);
        for (uint32_t sgm = (uint32_t)lane; sgm < n_full; sgm += kWave)
            if (at_full + sgm < units.cap_full) full[at_full + sgm] = make_uint2(head, sgm);
        if (lane == 0 && rest && at_part < units.cap_part) units.units[units.list_begin(shard, cls) + at_part] = make_uint2(head, n_full);
    }
}

__global__ __launch_bounds__(kWave, GSR_FWD_WAVES) void k_render_fwd(FrameK f, int n_tiles, int c, int finalize_all,
                                                      const uint2 *__restrict__ ranges_c, uint32_t *__restrict__ open,
                                                      const uint32_t *__restrict__ sorted_gid,
                                                      const float4 *__restrict__ records, const float *__restrict__ bg,
                                                      float *__restrict__ out_color, float *__restrict__ T_state,
                                                      int32_t *__restrict__ last_enc, uint32_t *__restrict__ tile_walk_c,
                                                      float *__restrict__ ckpt, float *__restrict__ ckpt_start_c,
                                                      UnitLists units, uint32_t *__restrict__ unit_count)
{
    __shared__ float4 sh_rec[kWave * 3];
#ifdef GSR_FWD_TRACE
    TraceEnd trace_end{(unsigned long long)wall_clock64(), (int)blockIdx.x};
#endif
    // tile = block: consecutive tiles run on different XCDs (block b -> XCD b % 8), which balances them; contiguous bands
    // per XCD (round 1) finished up to 20 % apart at cfg3 and bought nothing measurable in L2 hits
    const int t = (int)blockIdx.x;
    const int tx = t % f.Gx, ty = f.ty0 + t / f.Gx;
    const int tile = ty * f.Gx + tx;
    if (open[tile] == 0u) return;                       // closed by an earlier chunk: pixels are final
    const uint2 rng = ranges_c[tile];
    const int lane = threadIdx.x;
    const int px0 = tx * GSR_TILE + lane_px(lane), py0 = ty * GSR_TILE + lane_py(lane);       // the lane's pixel in quadrant 0
    const float fy0 = (float)py0, fy1 = fy0 + 8.f;
    const v2f fxv = {(float)px0, (float)(px0 + 8)};
    const size_t N = (size_t)f.W * f.H;

    FwdPair P0, P1;
    auto load_px = [&](int k, float &t_, float &r_, float &g_, float &b_, int &last) {
        const int px = px0 + (k & 1) * 8, py = py0 + (k >> 1) * 8;
        const bool inside = px < f.W && py < f.H;
        t_ = inside ? 1.f : -1.f; r_ = g_ = b_ = 0.f; last = 0;
        if (c > 0 && inside) {
            const size_t pix = (size_t)py * f.W + px;
            t_ = T_state[pix];
            r_ = out_color[pix]; g_ = out_color[N + pix]; b_ = out_color[2 * N + pix];
            last = last_enc[pix];
        }
    };
    {
        float t0, r0, g0, b0, t1, r1, g1, b1;
        load_px(0, t0, r0, g0, b0, P0.last0); load_px(1, t1, r1, g1, b1, P0.last1);
        P0.T = v2f{t0, t1}; P0.Cr = v2f{r0, r1}; P0.Cg = v2f{g0, g1}; P0.Cb = v2f{b0, b1};
        load_px(2, t0, r0, g0, b0, P1.last0); load_px(3, t1, r1, g1, b1, P1.last1);
        P1.T = v2f{t0, t1}; P1.Cr = v2f{r0, r1}; P1.Cg = v2f{g0, g1}; P1.Cb = v2f{b0, b1};
    }

    const int n_total = (int)(rng.y - rng.x);
    const int enc_base = (c + 1) << kLastShift;
    // The blend backward walks this list front to back in independent segments of kSeg entries (gsr_internal.h): it starts a
    // segment from the pixels' state (live transmittance, colour so far) in front of the segment's first entry, kept here.
    auto checkpoint = [&](float *dst) {                 // [quadrant][T, r, g, b][lane]: sixteen coalesced 256-byte stores
        dst[0 * kWave + lane] = P0.T[0]; dst[1 * kWave + lane] = P0.Cr[0]; dst[2 * kWave + lane] = P0.Cg[0]; dst[3 * kWave + lane] = P0.Cb[0];
        dst[4 * kWave + lane] = P0.T[1]; dst[5 * kWave + lane] = P0.Cr[1]; dst[6 * kWave + lane] = P0.Cg[1]; dst[7 * kWave + lane] = P0.Cb[1];
        dst[8 * kWave + lane] = P1.T[0]; dst[9 * kWave + lane] = P1.Cr[0]; dst[10 * kWave + lane] = P1.Cg[0]; dst[11 * kWave + lane] = P1.Cb[0];
        dst[12 * kWave + lane] = P1.T[1]; dst[13 * kWave + lane] = P1.Cr[1]; dst[14 * kWave + lane] = P1.Cg[1]; dst[15 * kWave + lane] = P1.Cb[1];
    };
    if (c > 0 && n_total > 0) checkpoint(ckpt_start_c + (size_t)tile * kCkptFloats);
    for (int base = 0; base < n_total; base += kWave) {
        unsigned live = live_quadrants(P0, P1);
        if (live == 0u) break;
        if (base > 0 && base % kSeg == 0) checkpoint(ckpt + (size_t)((rng.x + (uint32_t)base) / kSeg) * kCkptFloats);
        const int n = min(kWave, n_total - base);
        __syncthreads();
        const unsigned mymask = stage_batch(sh_rec, lane, n, sorted_gid, rng.x + base, records);
        __syncthreads();
        // splats of the batch that can reach a live quadrant, front to back (bit j = splat j): the others cost nothing
        unsigned long long act = __ballot((mymask & live) != 0u);
        int since_refresh = 0;
        while (act != 0ull) {
            const int j = __ffsll((long long)act) - 1;
            act &= act - 1ull;
            const unsigned m = (unsigned)__builtin_amdgcn_readlane((int)mymask, j) & live;
            const float4 a = sh_rec[3 * j], b = sh_rec[3 * j + 1];
            const float cbl = sh_rec[3 * j + 2].x;
            const int contributor = enc_base | (base + j + 1);
            const v2f dx = a.x - fxv;                                 // one subtraction from the pixel's x (all three blend kernels)
            const LpTerms lt = lp_terms(a.z, a.w, b.y, dx);
            if (m & 3u) {
                const float dy = a.y - fy0;
                fwd_pair(true, lp_at(lt, b.x, dy), b.y, b.z, b.w, cbl, contributor, P0);
            }
            if (m & 12u) {
                const float dy = a.y - fy1;
                fwd_pair(true, lp_at(lt, b.x, dy), b.y, b.z, b.w, cbl, contributor, P1);
            }
            if (++since_refresh == 8) {       // every 8 blended splats: quadrants (and tiles) that saturate mid-batch stop there
                since_refresh = 0;
                live = live_quadrants(P0, P1);
                act &= __ballot((mymask & live) != 0u);
            }
        }
    }
    const bool closing = live_quadrants(P0, P1) == 0u;
    const bool finalize = closing || finalize_all != 0;
    const float bg0 = finalize ? bg[0] : 0.f, bg1 = finalize ? bg[1] : 0.f, bg2 = finalize ? bg[2] : 0.f;
    auto store_px = [&](int k, float t_, float r_, float g_, float b_, int last) {
        const int px = px0 + (k & 1) * 8, py = py0 + (k >> 1) * 8;
        if (px < f.W && py < f.H) {
            const size_t pix = (size_t)py * f.W + px;
            const float tf = fabsf(t_);
            out_color[pix] = r_ + tf * bg0;
            out_color[N + pix] = g_ + tf * bg1;
            out_color[2 * N + pix] = b_ + tf * bg2;
            T_state[pix] = t_;
            last_enc[pix] = last;
        }
    };
    store_px(0, P0.T[0], P0.Cr[0], P0.Cg[0], P0.Cb[0], P0.last0);
    store_px(1, P0.T[1], P0.Cr[1], P0.Cg[1], P0.Cb[1], P0.last1);
    store_px(2, P1.T[0], P1.Cr[0], P1.Cg[0], P1.Cb[0], P1.last0);
    store_px(3, P1.T[1], P1.Cr[1], P1.Cg[1], P1.Cb[1], P1.last1);
    // 2 = open AND some pixel is still more than half transparent after everything so far: a tile no splat has covered yet
    // (the chunk plan merges the remaining chunks when such tiles exist: the frame is not going to close, gsr_api.hip)
    auto clear_px = [&](int k, float t_) { return px0 + (k & 1) * 8 < f.W && py0 + (k >> 1) * 8 < f.H && t_ > 0.5f; };
    const bool stuck = __ballot(clear_px(0, P0.T[0]) || clear_px(1, P0.T[1]) || clear_px(2, P1.T[0]) || clear_px(3, P1.T[1])) != 0ull;
    // what the backward's wave will walk of this chunk's range: up to the tile's deepest contributor (launch order of K7)
    auto depth_here = [&](int last) { return (last >> kLastShift) == c + 1 ? (last & ((1 << kLastShift) - 1)) : 0; };
    const int walked = wave_max_uniform(max(max(depth_here(P0.last0), depth_here(P0.last1)), max(depth_here(P1.last0), depth_here(P1.last1))));
    fwd_tile_epilogue(t, tile, c, lane, closing, stuck, walked, open, tile_walk_c, units, unit_count);
}

// ---- The same blend with the wave split into four GROUPS of 16 lanes, one per quadrant (the mapping and the walk of k_render_bwd):
// for chunks of SMALL splats (launch_render_fwd's `groups`: fewer than 4.5 tiles per Gaussian).  A splat that reaches one or two
// quadrants costs the lock-step kernel a pass of the whole wave; here up to four different splats share a pass.  Where splats reach
// most quadrants the lock-step kernel is faster (wave-uniform records, whole pairs skipped): cfg3n 258 -> 230 us with groups, but
// cfg3 105 -> 116, cfg5n 696 -> 706 - hence two kernels.  The checkpoints and every per-pixel array are the same either way.
// per 16-lane group (= quadrant): does it still hold a live pixel?  One ballot; bit g of the result = group g
__device__ __forceinline__ unsigned live_groups(const FwdPair &A, const FwdPair &B)
{
    const unsigned long long bal = __ballot(A.T[0] > 0.f || A.T[1] > 0.f || B.T[0] > 0.f || B.T[1] > 0.f);
    return ((bal & 0xFFFFull) ? 1u : 0u) | ((bal & 0xFFFF0000ull) ? 2u : 0u) | ((bal & 0xFFFF00000000ull) ? 4u : 0u) |
           ((bal & 0xFFFF000000000000ull) ? 8u : 0u);
}

__global__ __launch_bounds__(kWave, GSR_FWD_WAVES) void k_render_fwd_groups(FrameK f, int n_tiles, int c, int finalize_all,
                                                      const uint2 *__restrict__ ranges_c, uint32_t *__restrict__ open,
                                                      const uint32_t *__restrict__ sorted_gid,
                                                      const float4 *__restrict__ records, const float *__restrict__ bg,
                                                      float *__restrict__ out_color, float *__restrict__ T_state,
                                                      int32_t *__restrict__ last_enc, uint32_t *__restrict__ tile_walk_c,
                                                      float *__restrict__ ckpt, float *__restrict__ ckpt_start_c,
                                                      UnitLists units, uint32_t *__restrict__ unit_count)
{
    __shared__ float4 sh_rec[kWave * 3];
    const int t = (int)blockIdx.x;
    const int tx = t % f.Gx, ty = f.ty0 + t / f.Gx;
    const int tile = ty * f.Gx + tx;
    if (open[tile] == 0u) return;                       // closed by an earlier chunk: pixels are final
    const uint2 rng = ranges_c[tile];
    const int lane = threadIdx.x, grp = lane >> 4, gi = lane & 15;
    const int lx = gi & 3, ly = gi >> 2;                                                  // inside the quadrant; the other pixels are 4 further
    const int px0 = tx * GSR_TILE + (grp & 1) * 8 + lx, py0 = ty * GSR_TILE + (grp >> 1) * 8 + ly;
    const v2f fxv = {(float)px0, (float)(px0 + 4)};
    const float fy0 = (float)py0, fy1 = fy0 + 4.f;
    const size_t N = (size_t)f.W * f.H;
    const unsigned grp_shift = 8u * (unsigned)grp;

    FwdPair P0, P1;
    auto load_px = [&](int e, float &t_, float &r_, float &g_, float &b_, int &last) {
        const int px = px0 + (e & 1) * 4, py = py0 + (e >> 1) * 4;
        const bool inside = px < f.W && py < f.H;
        t_ = inside ? 1.f : -1.f; r_ = g_ = b_ = 0.f; last = 0;
        if (c > 0 && inside) {
            const size_t pix = (size_t)py * f.W + px;
            t_ = T_state[pix];
            r_ = out_color[pix]; g_ = out_color[N + pix]; b_ = out_color[2 * N + pix];
            last = last_enc[pix];
        }
    };
    {
        float t0, r0, g0, b0, t1, r1, g1, b1;
        load_px(0, t0, r0, g0, b0, P0.last0); load_px(1, t1, r1, g1, b1, P0.last1);
        P0.T = v2f{t0, t1}; P0.Cr = v2f{r0, r1}; P0.Cg = v2f{g0, g1}; P0.Cb = v2f{b0, b1};
        load_px(2, t0, r0, g0, b0, P1.last0); load_px(3, t1, r1, g1, b1, P1.last1);
        P1.T = v2f{t0, t1}; P1.Cr = v2f{r0, r1}; P1.Cg = v2f{g0, g1}; P1.Cb = v2f{b0, b1};
    }

    const int n_total = (int)(rng.y - rng.x);
    const int enc_base = (c + 1) << kLastShift;
    // checkpoints in the layout k_render_fwd writes: [quadrant][T, r, g, b][y << 3 | x inside the quadrant]
    auto checkpoint = [&](float *dst) {
        float *q = dst + 4 * grp * kWave + (ly << 3 | lx);
        q[0] = P0.T[0]; q[kWave] = P0.Cr[0]; q[2 * kWave] = P0.Cg[0]; q[3 * kWave] = P0.Cb[0];
        q[4] = P0.T[1]; q[kWave + 4] = P0.Cr[1]; q[2 * kWave + 4] = P0.Cg[1]; q[3 * kWave + 4] = P0.Cb[1];
        q[32] = P1.T[0]; q[kWave + 32] = P1.Cr[0]; q[2 * kWave + 32] = P1.Cg[0]; q[3 * kWave + 32] = P1.Cb[0];
        q[36] = P1.T[1]; q[kWave + 36] = P1.Cr[1]; q[2 * kWave + 36] = P1.Cg[1]; q[3 * kWave + 36] = P1.Cb[1];
    };
    if (c > 0 && n_total > 0) checkpoint(ckpt_start_c + (size_t)tile * kCkptFloats);
    for (int base = 0; base < n_total; base += kWave) {
        unsigned live = live_groups(P0, P1);
        if (live == 0u) break;
        if (base > 0 && base % kSeg == 0) checkpoint(ckpt + (size_t)((rng.x + (uint32_t)base) / kSeg) * kCkptFloats);
        const int n = min(kWave, n_total - base);
        __syncthreads();
        const unsigned mymask = stage_batch(sh_rec, lane, n, sorted_gid, rng.x + base, records);
        __syncthreads();
        // each group walks the batch's entries that reach ITS quadrant, front to back, while the quadrant holds a live pixel
        unsigned long long act0 = (live & 1u) ? __ballot((mymask & 1u) != 0u) : 0ull, act1 = (live & 2u) ? __ballot((mymask & 2u) != 0u) : 0ull,
                           act2 = (live & 4u) ? __ballot((mymask & 4u) != 0u) : 0ull, act3 = (live & 8u) ? __ballot((mymask & 8u) != 0u) : 0ull;
        int since_refresh = 0;
        while ((act0 | act1 | act2 | act3) != 0ull) {
            const unsigned j0 = act0 ? (unsigned)__ffsll((long long)act0) - 1u : 255u, j1 = act1 ? (unsigned)__ffsll((long long)act1) - 1u : 255u,
                           j2 = act2 ? (unsigned)__ffsll((long long)act2) - 1u : 255u, j3 = act3 ? (unsigned)__ffsll((long long)act3) - 1u : 255u;
            act0 &= act0 - 1ull; act1 &= act1 - 1ull; act2 &= act2 - 1ull; act3 &= act3 - 1ull;
            const unsigned j = __builtin_amdgcn_ubfe(j0 | j1 << 8 | j2 << 16 | j3 << 24, grp_shift, 8u);      // 255: the group idles
            const bool active = j < (unsigned)kWave;
            const unsigned jj = active ? j : 0u;                       // (an idle group reads the batch's first record: staged for sure)
            const float4 a = sh_rec[3u * jj], b = sh_rec[3u * jj + 1u];
            const float cbl = sh_rec[3u * jj + 2u].x;
            const int contributor = enc_base | (int)((unsigned)base + jj + 1u);
            const v2f dx = a.x - fxv;
            const LpTerms lt = lp_terms(a.z, a.w, b.y, dx);
            {
                const float dy = a.y - fy0;
                fwd_pair(active, lp_at(lt, b.x, dy), b.y, b.z, b.w, cbl, contributor, P0);
            }
            {
                const float dy = a.y - fy1;
                fwd_pair(active, lp_at(lt, b.x, dy), b.y, b.z, b.w, cbl, contributor, P1);
            }
            if (++since_refresh == 8) {       // every 8 passes: quadrants (and tiles) that saturate mid-batch stop there
                since_refresh = 0;
                live = live_groups(P0, P1);
                if (!(live & 1u)) act0 = 0ull;
                if (!(live & 2u)) act1 = 0ull;
                if (!(live & 4u)) act2 = 0ull;
                if (!(live & 8u)) act3 = 0ull;
            }
        }
    }
    const bool closing = live_groups(P0, P1) == 0u;
    const bool finalize = closing || finalize_all != 0;
    const float bg0 = finalize ? bg[0] : 0.f, bg1 = finalize ? bg[1] : 0.f, bg2 = finalize ? bg[2] : 0.f;
    auto store_px = [&](int e, float t_, float r_, float g_, float b_, int last) {
        const int px = px0 + (e & 1) * 4, py = py0 + (e >> 1) * 4;
        if (px < f.W && py < f.H) {
            const size_t pix = (size_t)py * f.W + px;
            const float tf = fabsf(t_);
            out_color[pix] = r_ + tf * bg0;
            out_color[N + pix] = g_ + tf * bg1;
            out_color[2 * N + pix] = b_ + tf * bg2;
            T_state[pix] = t_;
            last_enc[pix] = last;
        }
    };
    store_px(0, P0.T[0], P0.Cr[0], P0.Cg[0], P0.Cb[0], P0.last0);
    store_px(1, P0.T[1], P0.Cr[1], P0.Cg[1], P0.Cb[1], P0.last1);
    store_px(2, P1.T[0], P1.Cr[0], P1.Cg[0], P1.Cb[0], P1.last0);
    store_px(3, P1.T[1], P1.Cr[1], P1.Cg[1], P1.Cb[1], P1.last1);
    auto clear_px = [&](int e, float t_) { return px0 + (e & 1) * 4 < f.W && py0 + (e >> 1) * 4 < f.H && t_ > 0.5f; };
    const bool stuck = __ballot(clear_px(0, P0.T[0]) || clear_px(1, P0.T[1]) || clear_px(2, P1.T[0]) || clear_px(3, P1.T[1])) != 0ull;
    auto depth_here = [&](int last) { return (last >> kLastShift) == c + 1 ? (last & ((1 << kLastShift) - 1)) : 0; };
    const int walked = wave_max_uniform(max(max(depth_here(P0.last0), depth_here(P0.last1)), max(depth_here(P1.last0), depth_here(P1.last1))));
    fwd_tile_epilogue(t, tile, c, lane, closing, stuck, walked, open, tile_walk_c, units, unit_count);
}

int launch_render_fwd(const FrameK &f, const gsr_camera &cam, int c, bool last_chunk, int sort_result, const GeomWS &gw, const BinningWS &bw,
                      ImageWS &iw, float *out_color, bool debug, hipStream_t s, bool groups)
{
    const int n_tiles = (f.ty1 - f.ty0) * f.Gx;
    if (n_tiles <= 0) return GSR_OK;
    ProfileScope prof("render_fwd", s);
    const size_t Tn = (size_t)f.Gx * f.Gy;
    hipLaunchKernelGGL(groups ? k_render_fwd_groups : k_render_fwd, dim3(n_tiles), dim3(kWave), 0, s, f, n_tiles, c, last_chunk ? 1 : 0,
                       iw.ranges + (size_t)c * Tn, iw.open, bw.gids[1], gw.records, cam.bg, out_color, iw.T_state,
                       iw.last_enc, iw.tile_walk + (size_t)c * Tn, bw.ckpt,
                       c > 0 ? iw.ckpt_start + (size_t)(c - 1) * Tn * kCkptFloats : nullptr, bw.units, iw.unit_count);
    GSR_LAUNCH_CHECK("render_fwd", debug, s);
    return GSR_OK;
}

// ------------------------------------------------------------------------------------------- K7
// One launch for the whole frame, one wave per WORK UNIT (tile, chunk, segment of kSeg list entries), FRONT TO BACK.
//
// A.9 walks a pixel's list back to front with the colour behind the current splat as its state.  The same derivative in
// front-to-back form: with w_j = alpha_j T_j (T_j = transmittance in front of splat j), D_i = sum_{j <= i} w_j <c_j, dL/dpix>
// (the prefix of the pixel's colour, dotted with the upstream gradient) and Q = <out_color, dL/dpix> (the FINAL pixel,
// background term T_final bg included),
//     dL/dalpha_i = T_i <c_i, dL/dpix> - (Q - D_i) / (1 - alpha_i)
// since (Q - D_i) is what all splats behind i and the background contribute.  The state per pixel is (T, D): T is
// advanced by the multiplication the forward used (no division by 1 - alpha to recover it), and a walk may start anywhere
// the forward left (T, colour): a checkpoint every kSeg entries, D = <colour so far, dL/dpix>.  So the segments of a tile
// are independent units — 21 k units of <= 128 entries instead of 8 160 tiles of ~400 at cfg3n, where the launch used to end
// with a third of its time draining.
//
// Per pixel and splat, written so that a rejected pixel runs the same arithmetic with alpha = 0 and G = 0: T, D and every
// partial sum then stay exactly unchanged, and only alpha and G need a select.
//
// Algebra: with ga = opacity * G (the alpha before its 0.99 clamp; 0 where the pixel rejects the splat) every term of A.9
// that carries G * dL/dG = G * opacity * dL/dalpha is dL/dalpha * ga, so neither the opacity nor its reciprocal is needed
// per pixel (dL/dopacity = sum(G dL/dalpha) = sum(ga dL/dalpha) / opacity: one division per splat, after the reduction), and
// the conic enters through the pre-scaled record fields directly: cA = -2 ln2 qA, cB = -ln2 qB, cC = -2 ln2 qC, so
// -(tx cA + ty cB) = ln2 (2 qA tx + qB ty): the factor ln2 goes into the row store.
struct BwdSplat {            // per-splat values (uniform over a 16-lane group)
    float lop, cr, cg, cb;
};
struct BwdPair {             // state of a lane's pair of pixels (4 apart in x, the same row)
    v2f T, E, dpr, dpg, dpb;           // transmittance in front of the current splat; E = Q - D: what everything behind the current
                                       // splat (and the background) still adds to <pixel, dL/dpix>; dL/dpix
    int limit0, limit1;                // contributors of the current chunk each pixel takes part in
};
// the lane's partial sums of one splat, per pair element: X = sum tA dx, Y = sum tA dy (dL/dmean2D = ln2 (2 qA X + qB Y, 2 qC Y + qB X):
// formed once per splat by the lane that stores the row), S2..S4 the conic's, S5 the opacity's, S6..S8 the colour's
struct BwdAcc { v2f X, Y, S2, S3, S4, S5, S6, S7, S8; };
__device__ __forceinline__ float add_halves(v2f v)
{
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(v[0]), "v"(v[1]));
    return r;
}

template <bool INIT>         // INIT: the splat's first pair, the sums start here (A comes in undefined)
__device__ __forceinline__ void bwd_pair(const BwdSplat sp, v2f lp, v2f dx, float dy, int pos, BwdPair &P, BwdAcc &A)
{
    const bool valid0 = (pos < P.limit0) && !(lp[0] > sp.lop) && !(lp[0] < kLog2AlphaMin);      // power > 0 <=> lp > lop
    const bool valid1 = (pos < P.limit1) && !(lp[1] > sp.lop) && !(lp[1] < kLog2AlphaMin);
    // a rejected pixel gets lp = -inf: ga = opacity * G and alpha come out 0 by themselves
    const v2f ga = {__builtin_amdgcn_exp2f(valid0 ? lp[0] : -INFINITY), __builtin_amdgcn_exp2f(valid1 ? lp[1] : -INFINITY)};
    const v2f ae = {fminf((float)GSR_ALPHA_MAX, ga[0]), fminf((float)GSR_ALPHA_MAX, ga[1])};
    const v2f one_m = 1.f - ae;
    const v2f inv1ma = {fast_rcp(one_m[0]), fast_rcp(one_m[1])};
    const v2f cdp = sp.cr * P.dpr + sp.cg * P.dpg + sp.cb * P.dpb;      // <c_i, dL/dpix>
    const v2f w = ae * P.T;                                              // d colour / d rgb
    P.E -= w * cdp;                                                      // this splat's own share leaves the remainder
    const v2f dL_dalpha = P.T * cdp - P.E * inv1ma;
    P.T = P.T * one_m;                                                   // exactly the forward's update
    const v2f tA = dL_dalpha * ga;
    const v2f tx = tA * dx, ty = tA * dy;
    if constexpr (INIT) {
        A.X = tx; A.Y = ty;
        A.S2 = tx * dx;
        A.S3 = tx * dy;
        A.S4 = ty * dy;
        A.S5 = tA;
        A.S6 = w * P.dpr; A.S7 = w * P.dpg; A.S8 = w * P.dpb;
    } else {
        A.X += tx; A.Y += ty;
        A.S2 += tx * dx;
        A.S3 += tx * dy;
        A.S4 += ty * dy;
        A.S5 += tA;
        A.S6 += w * P.dpr; A.S7 += w * P.dpg; A.S8 += w * P.dpb;
    }
}

#ifndef GSR_BWD_ORDERED_ADDS
#define GSR_BWD_ORDERED_ADDS 1     // groups that meet on an entry in one pass add to its LDS sums one group after the other: the order of
#endif                             // every floating-point sum is fixed by the program (0: one LDS instruction for all four, 2 % faster)
// The kernel: one wave per work unit, front to back from the forward's checkpoint, with the wave split into four GROUPS of 16 lanes,
// one per 8x8 quadrant of the tile.  A small splat reaches
// one or two quadrants; walked in lock step the whole wave spends a pass (and a wave-wide reduction) on it.  Here lane l belongs
// to quadrant g = l >> 4 and owns the pixels (i & 3 [+ 4], i >> 2 [+ 4]), i = l & 15, of it (two packed pairs), and each group walks
// only the batch's entries whose mask has ITS bit: in one pass the wave works on up to four different splats, and one transposing
// butterfly over the rows of 16 lanes (row_sum9_transpose) reduces all four.  The order inside a quadrant is the list's, which is
// all the blend needs.  A group's totals are added to the entry's nine sums in LDS; when the batch is through, lane j forms entry
// j's gradient row and stores it whole (48 B) - rows are written for every entry some group attempted (row_valid).
// A batch ends when its slowest group does (measured imbalance over a frame: 1.01 .. 1.14, tools/quad_stats.py).
__global__ __launch_bounds__(kWave, GSR_BWD_WAVES) void k_render_bwd(FrameK f, const uint2 *__restrict__ ranges,
                                                      const uint32_t *__restrict__ tile_walk,
                                                      const uint32_t *__restrict__ sorted_gid, const uint32_t *__restrict__ sorted_slot,
                                                      const float4 *__restrict__ records, const float *__restrict__ out_color,
                                                      const float *__restrict__ T_state, const int32_t *__restrict__ last_enc,
                                                      const float *__restrict__ dL_dpix, const float *__restrict__ ckpt,
                                                      const float *__restrict__ ckpt_start, float4 *__restrict__ grad_rows,
                                                      uint8_t *__restrict__ row_valid, UnitLists units,
                                                      const uint32_t *__restrict__ unit_count)
{
    __shared__ float4 sh_rec[kWave * 3];
    __shared__ float sh_acc[kWave * 9];          // the batch's sums, [entry][value]: stride 9 words, conflict-free by lane
#ifdef GSR_BWD_TRACE
    TraceEnd trace_end{(unsigned long long)wall_clock64(), (int)blockIdx.x};
#endif
    const int shard = (int)(blockIdx.x & (kUnitShards - 1));
    uint32_t list_end[kUnitClasses];
    {
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < kUnitClasses; ++k) { run += min(unit_count[shard * kUnitClasses + k], units.list_cap(k)); list_end[k] = run; }
    }
    const uint32_t n_units = list_end[kUnitClasses - 1];
    const size_t Tn = (size_t)f.Gx * f.Gy;
    const int lane = threadIdx.x, grp = lane >> 4, gi = lane & 15;
    const size_t N = (size_t)f.W * f.H;
    const float half_w = 0.5f * (float)f.W, half_h = 0.5f * (float)f.H;
    // where the butterfly leaves this lane's share of a group's totals: which value (of which register), if any
    const int quad = (lane >> 2) & 3, in_quad = lane & 3;
    const unsigned acc_idx = (unsigned)(in_quad == 0 ? (quad < 2 ? quad : quad + 2) : in_quad == 1 ? (quad < 2 ? quad + 2 : quad + 4) : 8);
    const unsigned grp_shift = 8u * (unsigned)grp;
    const bool acc_on = in_quad < 2 || (in_quad == 2 && quad == 0);
#pragma unroll
    for (int k = 0; k < 9; ++k) sh_acc[lane * 9 + k] = 0.f;
    for (uint32_t u = blockIdx.x / kUnitShards; u < n_units; u += gridDim.x / kUnitShards) {
        int cls = 0;
        for (int k = 0; k < kUnitClasses - 1; ++k) cls += u >= list_end[k] ? 1 : 0;
        const uint2 unit = units.units[units.list_begin(shard, cls) + (u - (cls ? list_end[cls - 1] : 0u))];
        const int tile = (int)(unit.x & ((1u << kUnitTileBits) - 1u)), c = (int)(unit.x >> kUnitTileBits);
        const int sgm = (int)unit.y;
        const int ty = tile / f.Gx, tx = tile - ty * f.Gx;
        const int lx = gi & 3, ly = gi >> 2;                                    // inside the quadrant; the other pixels are 4 further
        const int px0 = tx * GSR_TILE + (grp & 1) * 8 + lx, py0 = ty * GSR_TILE + (grp >> 1) * 8 + ly;
        // dx and dy are one subtraction from the pixel's coordinate, as in the forward kernels: every pixel's lp, alpha and transmittance
        // are the forward's, bit for bit
        const v2f fxv = {(float)px0, (float)(px0 + 4)};
        const float fy0 = (float)py0, fy1 = fy0 + 4.f;
        const uint2 rng = ranges[(size_t)c * Tn + tile];
        const int n_total = (int)(rng.y - rng.x);
        const int seg_begin = sgm * kSeg, seg_end = min(n_total, seg_begin + kSeg);
        const int walk_end = min(seg_end, (int)tile_walk[(size_t)c * Tn + tile]);

        BwdPair P0, P1;
        const float *chk = (sgm > 0) ? ckpt + (size_t)((rng.x + (uint32_t)seg_begin) / kSeg) * kCkptFloats
                                     : (c > 0 ? ckpt_start + ((size_t)(c - 1) * Tn + tile) * kCkptFloats : nullptr);
        auto load_px = [&](int e, float &Tk, float &Ek, float &r_, float &g_, float &b_, int &limit) {
            const int ox = (e & 1) * 4, oy = (e >> 1) * 4;
            const int px = px0 + ox, py = py0 + oy;
            const bool inside = px < f.W && py < f.H;
            const size_t pix = inside ? (size_t)py * f.W + px : 0;
            const int enc = inside ? last_enc[pix] : 0;
            const int c_last = (enc >> kLastShift) - 1;
            limit = c < c_last ? n_total : (c == c_last ? (enc & ((1 << kLastShift) - 1)) : 0);
            r_ = inside ? dL_dpix[pix] : 0.f; g_ = inside ? dL_dpix[N + pix] : 0.f; b_ = inside ? dL_dpix[2 * N + pix] : 0.f;
            float er = inside ? out_color[pix] : 0.f, eg = inside ? out_color[N + pix] : 0.f, eb = inside ? out_color[2 * N + pix] : 0.f;
            Tk = 1.f;
            if (chk) {
                // the forward's checkpoint layout: [quadrant][T, r, g, b][its lane = (y << 3 | x) inside the quadrant]
                const float *q = chk + 4 * grp * kWave + ((ly + oy) << 3 | (lx + ox));
                Tk = q[0];
                er -= q[kWave]; eg -= q[2 * kWave]; eb -= q[3 * kWave];
            }
            Ek = er * r_ + eg * g_ + eb * b_;
        };
        {
            float Ta, Ea, ra, ga, bla, Tb, Eb, rb, gb, blb;
            load_px(0, Ta, Ea, ra, ga, bla, P0.limit0); load_px(1, Tb, Eb, rb, gb, blb, P0.limit1);
            P0.T = v2f{Ta, Tb}; P0.E = v2f{Ea, Eb}; P0.dpr = v2f{ra, rb}; P0.dpg = v2f{ga, gb}; P0.dpb = v2f{bla, blb};
            load_px(2, Ta, Ea, ra, ga, bla, P1.limit0); load_px(3, Tb, Eb, rb, gb, blb, P1.limit1);
            P1.T = v2f{Ta, Tb}; P1.E = v2f{Ea, Eb}; P1.dpr = v2f{ra, rb}; P1.dpg = v2f{ga, gb}; P1.dpb = v2f{bla, blb};
        }
        // per quadrant (= per group): its last participating contributor
        int gmax = max(max(P0.limit0, P0.limit1), max(P1.limit0, P1.limit1));
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) gmax = max(gmax, __shfl_xor(gmax, off));
        const int qmax0 = min(walk_end, __builtin_amdgcn_readlane(gmax, 0)), qmax1 = min(walk_end, __builtin_amdgcn_readlane(gmax, 16)),
                  qmax2 = min(walk_end, __builtin_amdgcn_readlane(gmax, 32)), qmax3 = min(walk_end, __builtin_amdgcn_readlane(gmax, 48));
        const int max_contrib = max(max(qmax0, qmax1), max(qmax2, qmax3));

        for (int base = seg_begin; base < seg_end; base += kWave) {
            if (base >= max_contrib) break;
            const int n = min(kWave, seg_end - base);
            const uint32_t slot = lane < n ? sorted_slot[rng.x + base + lane] : 0u;
            __syncthreads();
            unsigned mymask = stage_batch(sh_rec, lane, n, sorted_gid, rng.x + base, records);
            __syncthreads();
            const int mypos = base + lane;
            mymask &= (mypos < qmax0 ? 1u : 0u) | (mypos < qmax1 ? 2u : 0u) | (mypos < qmax2 ? 4u : 0u) | (mypos < qmax3 ? 8u : 0u);
            // the entries each group has to walk (wave-uniform bit sets: scalar registers)
            unsigned long long act0 = __ballot((mymask & 1u) != 0u), act1 = __ballot((mymask & 2u) != 0u),
                               act2 = __ballot((mymask & 4u) != 0u), act3 = __ballot((mymask & 8u) != 0u);
            // a pixel's limit relative to the batch, clamped to [0, 64]: an idle group's entry number (255) fails it by itself
            const int L00 = P0.limit0, L01 = P0.limit1, L10 = P1.limit0, L11 = P1.limit1;
            P0.limit0 = min(max(L00 - base, 0), kWave); P0.limit1 = min(max(L01 - base, 0), kWave);
            P1.limit0 = min(max(L10 - base, 0), kWave); P1.limit1 = min(max(L11 - base, 0), kWave);
            while ((act0 | act1 | act2 | act3) != 0ull) {
                // each group's next entry (255: it is through with the batch), one byte per group in a scalar register
                const unsigned j0 = act0 ? (unsigned)__ffsll((long long)act0) - 1u : 255u, j1 = act1 ? (unsigned)__ffsll((long long)act1) - 1u : 255u,
                               j2 = act2 ? (unsigned)__ffsll((long long)act2) - 1u : 255u, j3 = act3 ? (unsigned)__ffsll((long long)act3) - 1u : 255u;
                act0 &= act0 - 1ull; act1 &= act1 - 1ull; act2 &= act2 - 1ull; act3 &= act3 - 1ull;      // (0 & anything = 0)
                const unsigned jw = j0 | j1 << 8 | j2 << 16 | j3 << 24;
                const unsigned j = __builtin_amdgcn_ubfe(jw, grp_shift, 8u);
                const unsigned jj = j < (unsigned)kWave ? j : 0u;       // (an idle group reads the batch's first record: staged for sure,
                                                                         // where the slots behind a short batch's end hold whatever LDS held)
                const float4 a = sh_rec[3u * jj], b = sh_rec[3u * jj + 1u];
                const BwdSplat sp{b.y, b.z, b.w, sh_rec[3u * jj + 2u].x};
                const v2f dx = a.x - fxv;
                const LpTerms lt = lp_terms(a.z, a.w, b.y, dx);
                BwdAcc A;
                {
                    const float dy = a.y - fy0;
                    bwd_pair<true>(sp, lp_at(lt, b.x, dy), dx, dy, (int)j, P0, A);
                }
                {
                    const float dy = a.y - fy1;
                    bwd_pair<false>(sp, lp_at(lt, b.x, dy), dx, dy, (int)j, P1, A);
                }
                float s[9] = {add_halves(A.X), add_halves(A.Y), add_halves(A.S2), add_halves(A.S3), add_halves(A.S4), add_halves(A.S5),
                              add_halves(A.S6), add_halves(A.S7), add_halves(A.S8)};
                row_sum9_transpose(s);
                const float mine = in_quad == 0 ? s[0] : in_quad == 1 ? s[2] : s[8];
#if GSR_BWD_ORDERED_ADDS
                // groups that meet on an entry in the same pass add in group order: one LDS instruction per group
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (acc_on && j < (unsigned)kWave && grp == k) atomicAdd(&sh_acc[jj * 9u + acc_idx], mine);
#else
                // (one instruction: the order in which the LDS serves lanes of different groups that meet on an entry is the hardware's)
                if (acc_on && j < (unsigned)kWave) atomicAdd(&sh_acc[jj * 9u + acc_idx], mine);
#endif
            }
            P0.limit0 = L00; P0.limit1 = L01; P1.limit0 = L10; P1.limit1 = L11;
            __syncthreads();
            if (mymask != 0u) {
                float v[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) { v[k] = sh_acc[lane * 9 + k]; sh_acc[lane * 9 + k] = 0.f; }
                const float4 a = sh_rec[3 * lane], b = sh_rec[3 * lane + 1];
                // -(X cA + Y cB) = ln2 (2 qA X + qB Y), likewise for y; gA, gB, gC carry the -1/2 of A.9; dL/dopacity = sum / opacity
                const float gx = 2.f * a.z * v[0] + a.w * v[1], gy = 2.f * b.x * v[1] + a.w * v[0];
                float4 *row = grad_rows + 3 * (size_t)slot;
                row[0] = make_float4(gx * (0.69314718f * half_w), gy * (0.69314718f * half_h), v[2] * -0.5f, v[3] * -0.5f);
                row[1] = make_float4(v[4] * -0.5f, v[5] * __builtin_amdgcn_exp2f(-b.y), v[6], v[7]);
                row[2] = make_float4(v[8], 0.f, 0.f, 0.f);
                row_valid[slot] = 1;
            }
        }
    }
}

int launch_render_bwd(const FrameK &f, int chunks_run, int sort_result, long long rows_upper, const GeomWS &gw, BinningWS &bw,
                      const ImageWS &iw, const float *out_color, const float *dL_dcolor, bool debug, hipStream_t s)
{
    const int n_tiles = (f.ty1 - f.ty0) * f.Gx;
    if (n_tiles <= 0 || chunks_run <= 0) return GSR_OK;
    // one block per unit while they fit the grid; the counts are the device's (the forward appended the units), the bound the
    // host's: a (tile, chunk) pair has at most n / kSeg + 1 units.  Blocks are dealt to the shards round-robin.
    long long per_shard = (rows_upper / kSeg + (long long)n_tiles * chunks_run) / kUnitShards + 1;
    if (per_shard > (long long)bw.units.shard_stride()) per_shard = (long long)bw.units.shard_stride();
    if (per_shard > (1 << 13)) per_shard = 1 << 13;
    ProfileScope prof("render_bwd", s);
    hipLaunchKernelGGL(k_render_bwd, dim3((unsigned)(per_shard * kUnitShards)), dim3(kWave), 0, s, f, iw.ranges, iw.tile_walk, bw.gids[1],
                       bw.vals[sort_result], gw.records, out_color, iw.T_state, iw.last_enc, dL_dcolor, bw.ckpt, iw.ckpt_start,
                       reinterpret_cast<float4 *>(bw.grad_rows), bw.row_valid, bw.units, iw.unit_count);
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
__global__ __launch_bounds__(kRedBlock) void k_reduce_rows(int r_begin, int n_ranks, const uint32_t *__restrict__ order,
                                                           const uint32_t *__restrict__ cnt_open,
                                                           const uint32_t *__restrict__ row_begin,
                                                           const uint8_t *__restrict__ row_valid,
                                                           const float4 *__restrict__ grad_rows, float4 *__restrict__ screen,
                                                           int write_empty)
{
    const int sub = threadIdx.x & (kRedGroup - 1);
    const int r = r_begin + (blockIdx.x * kRedBlock + threadIdx.x) / kRedGroup;
    const bool live = r < n_ranks;
    // the rank's three words in one round trip (the chain was count -> first row -> rows -> Gaussian: four)
    const uint32_t cnt = live ? cnt_open[r] : 0u, begin = live ? row_begin[r] : 0u, g = live ? order[r] : 0u;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    float a8 = 0.f;
    if (cnt) {
        for (uint32_t sl = begin + sub; sl < begin + cnt; sl += kRedGroup) {
            // a clear valid byte: nobody walked that far into the tile's list, or no pixel accepted the splat — the row was never
            // written.  Wide groups (big splats: most rows of a saturating frame are invalid) test the byte first; narrow ones (small
            // splats, nearly every row valid) load the row beside its byte — one memory round trip instead of two — and drop it after
            float4 r0, r1;
            float r2;
            if constexpr (kRedGroup >= 64) {
                if (!row_valid[sl]) continue;
                r0 = grad_rows[3 * (size_t)sl]; r1 = grad_rows[3 * (size_t)sl + 1]; r2 = grad_rows[3 * (size_t)sl + 2].x;
            } else {
                const uint8_t ok = row_valid[sl];
                r0 = grad_rows[3 * (size_t)sl]; r1 = grad_rows[3 * (size_t)sl + 1]; r2 = grad_rows[3 * (size_t)sl + 2].x;
                if (!ok) continue;                      // (whatever the unwritten row holds — NaN patterns included — is never added)
            }
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
    if (live && (cnt || write_empty) && sub < 3)
        screen[3 * (size_t)g + sub] = sub == 0 ? a0 : (sub == 1 ? a1 : make_float4(a8, 0.f, 0.f, 0.f));
}

// One launch per depth chunk that ran, each with its own lanes-per-Gaussian: a whole wave where the chunk's Gaussians own many
// rows each (the nearest, screen-filling splats: a few thousand rows), eight otherwise — and always eight for a chunk that went
// through the live filter, whose instance bound says nothing about what it emitted (a training frame's last chunk is most of the
// scene with a bound of tens of millions: 64 lanes for each of its 1e6 ranks was 100 us of idle threads).
int launch_reduce_rows(const FrameK &f, const gsr_frame_plan &plan, const GeomWS &gw, const BinningWS &bw, float *screen_grads,
                       int prezeroed, bool debug, hipStream_t s)
{
    // prezeroed: 0 = clear the whole tensor first; 1 = the caller already has; 2 = only the rows of the binned prefix will ever
    // be read (the sparse geometry backward of the same frame): every prefix row is written, zeros included, nothing else
    if (f.P == 0) return GSR_OK;
    ProfileScope prof("reduce_rows", s);
    if (prezeroed == 0) GSR_HIP_CHECK(hipMemsetAsync(screen_grads, 0, (size_t)f.P * kRowFloats * sizeof(float), s));
    const int write_empty = prezeroed == 2 ? 1 : 0;
    const int chunks = (plan.num_rendered > 0 && plan.chunks_run > 0) ? plan.chunks_run : 0;
    for (int c = 0; c < chunks && c < GSR_MAX_CHUNKS; ++c) {
        const int r0 = plan.chunk_rank_begin[c], r1 = plan.chunk_rank_begin[c + 1];
        if (r1 <= r0) continue;
        const bool filtered = (plan.chunks_filtered >> c) & 1;
        const long long avg = plan.chunk_instances_max[c] / (long long)(r1 - r0);
        const bool wide = !filtered && avg >= 48;
        // (two lanes per Gaussian for small splats, measured: 89 us against 81 at cfg3n, 352 against 374 at cfg5n — not kept)
        const long long threads = (long long)(r1 - r0) * (wide ? 64 : 8);
        const dim3 grid((unsigned)((threads + kRedBlock - 1) / kRedBlock));
        if (wide)
            hipLaunchKernelGGL(k_reduce_rows<64>, grid, dim3(kRedBlock), 0, s, r0, r1, gw.order, gw.cnt_open, gw.row_begin,
                               bw.row_valid, reinterpret_cast<const float4 *>(bw.grad_rows), reinterpret_cast<float4 *>(screen_grads), write_empty);
        else
            hipLaunchKernelGGL(k_reduce_rows<8>, grid, dim3(kRedBlock), 0, s, r0, r1, gw.order, gw.cnt_open, gw.row_begin,
                               bw.row_valid, reinterpret_cast<const float4 *>(bw.grad_rows), reinterpret_cast<float4 *>(screen_grads), write_empty);
    }
    GSR_LAUNCH_CHECK("reduce_rows", debug, s);
    return GSR_OK;
}

}  // namespace gsr
