// gsr_loss.hip — fused training loss of the reference's timed window (train.py:104-105):
//     loss = (1 - lambda) * mean|a - b| + lambda * (1 - mean(SSIM_map(a, b)))
// with the reference's SSIM (utils/loss_utils.py:33-63): 11x11 Gaussian window (sigma 1.5), depthwise,
// zero padding 5, C1 = 0.01^2, C2 = 0.03^2.  SURVEY 8f row f3: in torch this is 5 depthwise 11x11
// convolutions forward + their backward (MIOpen: ~10 ms per step at 1080p on MI355X, half of the whole
// train step); here it is two kernels.
//
// Forward: one 256-thread block per (32x32 tile, channel).  The tile + 5-px halo of a and b is staged in
// LDS once; the 11-tap window is applied separably (rows into LDS, then columns) to the five moments
// a, b, a^2, b^2, ab; the SSIM map and its three partial derivatives (wrt mu1, E[a^2], E[ab]) are formed
// in registers; the derivatives go to HBM (12 B/px) for the backward; the per-block sums of SSIM and
// |a-b| go to a partials array that a one-block kernel adds up in a fixed order (deterministic).
// Backward: same tiling, separable window over the three derivative maps, combined with a, b and
// sign(a - b).  HBM traffic: forward reads 8 B/px, writes 12; backward reads 20, writes 4.
#include "gsr_internal.h"

namespace gsr {

constexpr int kLT = 32;                 // output tile edge
constexpr int kLH = 5;                  // halo (window radius)
constexpr int kLI = kLT + 2 * kLH;      // 42: staged input edge
constexpr int kLBlock = 256;
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

struct Win { float g[11]; };
typedef float v2f __attribute__((ext_vector_type(2)));     // packed-fp32 operand: two moments of one pixel per VALU issue

// 1-D grid, XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so XCD x gets the contiguous run of
// (channel, tile row, tile column) items [start(x), start(x) + count(x)): neighbouring tiles, whose 5-px halos
// overlap, then share one L2 instead of fetching the halo from HBM once per XCD.
// Row ranges of a launch (multi-GPU slabs; the whole image is {0, H, 0, H, 0}): `sum` rows enter the L1 / SSIM
// sums and receive gradients, `map` rows get their SSIM derivative maps written (the slab plus the 5 rows either side
// that the backward's window reaches), tile rows start at ty0.
struct LossRows { int sum_b, sum_e, map_b, map_e, ty0; };

struct LossTile { int ch, x0, y0, bid; };
__device__ __forceinline__ LossTile loss_tile(int tiles_x, int tiles_y, int channels, int ty0 = 0)
{
    const int n = tiles_x * tiles_y * channels;
    const int b = (int)blockIdx.x;
    const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
    const int t = x * q + (x < r ? x : r) + i;
    LossTile o;
    o.bid = t;
    o.ch = t / (tiles_x * tiles_y);
    const int rem = t - o.ch * tiles_x * tiles_y;
    o.y0 = (ty0 + rem / tiles_x) * kLT;
    o.x0 = (rem % tiles_x) * kLT;
    return o;
}

static Win make_window()
{
    // exactly utils/loss_utils.py:20-22: exp(-(x - 5)^2 / (2 * 1.5^2)) in fp32, normalised by its fp32 sum
    Win w;
    float sum = 0.f;
    for (int i = 0; i < 11; ++i) { w.g[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += w.g[i]; }
    for (int i = 0; i < 11; ++i) w.g[i] /= sum;
    return w;
}

__device__ __forceinline__ float block_sum(float v, float *sh /*[4]*/)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// Register-blocked separable window.  Row pass: 168 threads, each one row x 8 adjacent columns (18 staged values of a
// and of b -> 8 x 5 moments); column pass: 256 threads, each one column x 4 adjacent rows (14 row-filtered values per
// moment).  ~115 LDS reads per thread instead of ~350 for one-output-at-a-time.
constexpr int kLX = 8;                  // staged columns start at x0 - 8 (16-byte aligned), the window needs x0 - 5
constexpr int kLSW = kLT + 2 * kLX;     // 48 staged columns = 12 float4 per row
constexpr int kLP = kLSW + 4;           // 52: staged row pitch (16-byte aligned, rows spread over 8 bank offsets)
constexpr int kLO = kLX - kLH;          // 3: column of the staged row that holds x0 - 5

// Stage rows y0-5 .. y0+36, columns x0-8 .. x0+39 of NP planes into LDS.  VEC (W % 4 == 0): one float4 per load, every
// float4 is either fully inside or fully outside the image; all loads of a thread are issued before the first LDS write.
template <int NP, bool VEC>
__device__ __forceinline__ void stage_planes(const float *const (&src)[NP], float *lds /* [NP][kLI][kLP] */, size_t plane, int H, int W,
                                             int x0, int y0)
{
    if constexpr (VEC) {
        constexpr int kPerPlane = kLI * (kLSW / 4);                 // 504 float4
        constexpr int kIters = (NP * kPerPlane + kLBlock - 1) / kLBlock;
        float4 v[kIters];
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int idx = threadIdx.x + it * kLBlock;
            v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < NP * kPerPlane) {
                const int pl = idx / kPerPlane, rem = idx % kPerPlane, r = rem / (kLSW / 4), c4 = rem % (kLSW / 4);
                const int y = y0 - kLH + r, x = x0 - kLX + 4 * c4;
                if (y >= 0 && y < H && x >= 0 && x < W)
                    v[it] = *reinterpret_cast<const float4 *>(src[pl] + plane + (size_t)y * W + x);
            }
        }
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            const int idx = threadIdx.x + it * kLBlock;
            if (idx < NP * kPerPlane) {
                const int pl = idx / kPerPlane, rem = idx % kPerPlane, r = rem / (kLSW / 4), c4 = rem % (kLSW / 4);
                *reinterpret_cast<float4 *>(lds + ((size_t)pl * kLI + r) * kLP + 4 * c4) = v[it];
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < kLI * kLI; idx += kLBlock) {
            const int r = idx / kLI, c = idx % kLI;
            const int y = y0 - kLH + r, x = x0 - kLH + c;
            const bool in = y >= 0 && y < H && x >= 0 && x < W;
            const size_t p = plane + (size_t)y * W + x;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) lds[((size_t)pl * kLI + r) * kLP + kLO + c] = in ? src[pl][p] : 0.f;
        }
    }
}

constexpr int kRowW = 8;                // outputs per thread in the row pass
constexpr int kRowItems = kLI * (kLT / kRowW);     // 168
constexpr int kColH = 4;                // outputs per thread in the column pass

template <bool VEC>
__global__ __launch_bounds__(kLBlock) void k_loss_fwd(int C, int H, int W, LossRows rows, int tiles_y, Win win, const float *__restrict__ a,
                                                      const float *__restrict__ b,
                                                      float *__restrict__ d_mu, float *__restrict__ d_eaa, float *__restrict__ d_eab,
                                                      float *__restrict__ partial_ssim, float *__restrict__ partial_l1)
{
    // LDS is used twice: first the staged tile + halo of a and b, then (after every thread holds its row-filtered
    // values in registers) the five row-filtered moment planes OVER the same bytes.
    constexpr int kStageFloats = 2 * kLI * kLP;
    constexpr int kHzFloats = 5 * kLI * (kLT + 1);
    __shared__ __attribute__((aligned(16))) float lds[kHzFloats > kStageFloats ? kHzFloats : kStageFloats];
    __shared__ float red[4];
    float (*sa)[kLP] = reinterpret_cast<float (*)[kLP]>(lds);
    float (*sb)[kLP] = reinterpret_cast<float (*)[kLP]>(lds + kLI * kLP);
    float (*hz)[kLI][kLT + 1] = reinterpret_cast<float (*)[kLI][kLT + 1]>(lds);
    const LossTile lt = loss_tile((W + kLT - 1) / kLT, tiles_y, C, rows.ty0);
    const int ch = lt.ch, x0 = lt.x0, y0 = lt.y0;
    const size_t plane = (size_t)ch * H * W;
    {
        const float *const src[2] = {a, b};
        stage_planes<2, VEC>(src, lds, plane, H, W, x0, y0);
    }
    __syncthreads();
    // column-pass mapping of this thread: column cc, rows rr0 .. rr0 + 3; centre values for the L1 term are read now,
    // before the staging area is overwritten
    const int cc = threadIdx.x % kLT, rr0 = (threadIdx.x / kLT) * kColH;
    float ca[kColH], cbv[kColH];
#pragma unroll
    for (int k = 0; k < kColH; ++k) { ca[k] = sa[rr0 + k + kLH][kLO + cc + kLH]; cbv[k] = sb[rr0 + k + kLH][kLO + cc + kLH]; }
    // row pass
    float m[5][kRowW];
    const bool row_item = threadIdx.x < kRowItems;
    const int hr = threadIdx.x / (kLT / kRowW), hc0 = (threadIdx.x % (kLT / kRowW)) * kRowW;
    if (row_item) {
        v2f ab[kRowW + 10];                                   // (a, b) of one staged pixel: the moments pair up as
                                                               // (E[a], E[b]) and (E[a^2], E[b^2]) -> v_pk_fma_f32
        // The 18 staged values start at column 3 + 8 q of the row: read as SIX aligned 16-byte words per plane (columns 8 q ..
        // 8 q + 23).  Single-float reads put a wave's 64 addresses (row pitch and 8 q are multiples of 4 floats) on 8 of the 32
        // banks: an 8-way conflict on every one of them.
        float ra[kRowW + 16], rb[kRowW + 16];
#pragma unroll
        for (int j = 0; j < (kRowW + 16) / 4; ++j) {
            const float4 va = *reinterpret_cast<const float4 *>(&sa[hr][hc0 + 4 * j]);
            const float4 vb = *reinterpret_cast<const float4 *>(&sb[hr][hc0 + 4 * j]);
            ra[4 * j] = va.x; ra[4 * j + 1] = va.y; ra[4 * j + 2] = va.z; ra[4 * j + 3] = va.w;
            rb[4 * j] = vb.x; rb[4 * j + 1] = vb.y; rb[4 * j + 2] = vb.z; rb[4 * j + 3] = vb.w;
        }
        v2f sq[kRowW + 10];                                   // (a^2, b^2) and a b of every staged pixel, once: three FMAs per tap
        float pr[kRowW + 10];                                 // (four instructions per tap when the products are formed inside)
#pragma unroll
        for (int i = 0; i < kRowW + 10; ++i) { ab[i] = v2f{ra[kLO + i], rb[kLO + i]}; sq[i] = ab[i] * ab[i]; pr[i] = ab[i][0] * ab[i][1]; }
#pragma unroll
        for (int o = 0; o < kRowW; ++o) {
            v2f s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
            float s4 = 0.f;
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                const float g = win.g[i];
                s01 += g * ab[o + i];
                s23 += g * sq[o + i];
                s4 += g * pr[o + i];
            }
            m[0][o] = s01[0]; m[1][o] = s01[1]; m[2][o] = s23[0]; m[3][o] = s23[1]; m[4][o] = s4;
        }
    }
    __syncthreads();
    if (row_item) {
#pragma unroll
        for (int pl = 0; pl < 5; ++pl)
#pragma unroll
            for (int o = 0; o < kRowW; ++o) hz[pl][hr][hc0 + o] = m[pl][o];
    }
    __syncthreads();
    // column pass, moments in pairs again
    float acc[5][kColH];
    {
        v2f v01[kColH + 10], v23[kColH + 10];
        float v4[kColH + 10];
#pragma unroll
        for (int i = 0; i < kColH + 10; ++i) {
            v01[i] = v2f{hz[0][rr0 + i][cc], hz[1][rr0 + i][cc]};
            v23[i] = v2f{hz[2][rr0 + i][cc], hz[3][rr0 + i][cc]};
            v4[i] = hz[4][rr0 + i][cc];
        }
#pragma unroll
        for (int k = 0; k < kColH; ++k) {
            v2f t01 = {0.f, 0.f}, t23 = {0.f, 0.f};
            float t4 = 0.f;
#pragma unroll
            for (int i = 0; i < 11; ++i) { const float g = win.g[i]; t01 += g * v01[k + i]; t23 += g * v23[k + i]; t4 += g * v4[k + i]; }
            acc[0][k] = t01[0]; acc[1][k] = t01[1]; acc[2][k] = t23[0]; acc[3][k] = t23[1]; acc[4][k] = t4;
        }
    }
    float s_ssim = 0.f, s_l1 = 0.f;
#pragma unroll
    for (int k = 0; k < kColH; ++k) {
        const int y = y0 + rr0 + k, x = x0 + cc;
        const float mu1 = acc[0][k], mu2 = acc[1][k], eaa = acc[2][k], ebb = acc[3][k], eab = acc[4][k];
        if (y < H && x < W && y >= rows.map_b && y < rows.map_e) {
            const float s1 = eaa - mu1 * mu1, s2 = ebb - mu2 * mu2, s12 = eab - mu1 * mu2;
            const float A1 = 2.f * mu1 * mu2 + kC1, A2 = 2.f * s12 + kC2;
            const float B1 = mu1 * mu1 + mu2 * mu2 + kC1, B2 = s1 + s2 + kC2;
            const float inv = 1.f / (B1 * B2);
            const float mm = A1 * A2 * inv;
            if (y >= rows.sum_b && y < rows.sum_e) {
                s_ssim += mm;
                s_l1 += fabsf(ca[k] - cbv[k]);
            }
            if (d_mu) {
                const size_t p = plane + (size_t)y * W + x;
                d_mu[p] = 2.f * mu2 * (A2 - A1) * inv - 2.f * mu1 * mm * (B2 - B1) * inv;
                d_eaa[p] = -mm / B2;
                d_eab[p] = 2.f * A1 * inv;
            }
        }
    }
    const float t_ssim = block_sum(s_ssim, red);
    const float t_l1 = block_sum(s_l1, red);
    if (threadIdx.x == 0) {
        partial_ssim[lt.bid] = t_ssim;
        partial_l1[lt.bid] = t_l1;
    }
}

__global__ __launch_bounds__(kLBlock) void k_loss_finish(int nblocks, float inv_count, float lambda, int raw_sums,
                                                         const float *__restrict__ partial_ssim,
                                                         const float *__restrict__ partial_l1, float *__restrict__ out /*[3]*/)
{
    __shared__ float red[4];
    float s = 0.f, l = 0.f;
    for (int i0 = threadIdx.x; i0 < nblocks; i0 += 8 * kLBlock) {        // eight independent loads in flight per thread
        float ps[8], pl[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * kLBlock;
            ps[u] = i < nblocks ? partial_ssim[i] : 0.f;
            pl[u] = i < nblocks ? partial_l1[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s += ps[u]; l += pl[u]; }                // fixed order: deterministic
    }
    const float ts = block_sum(s, red) * (raw_sums ? 1.f : inv_count);
    const float tl = block_sum(l, red) * (raw_sums ? 1.f : inv_count);
    if (threadIdx.x == 0) {
        if (raw_sums) { out[0] = tl; out[1] = ts; }      // slab: un-normalised sums, the caller adds the ranks' pairs
        else {
            out[0] = (1.f - lambda) * tl + lambda * (1.f - ts);
            out[1] = tl;
            out[2] = ts;
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(kLBlock) void k_loss_bwd(int C, int H, int W, LossRows rows, int tiles_y, Win win, float inv_count, float lambda,
                                                      const float *__restrict__ upstream, const float *__restrict__ a,
                                                      const float *__restrict__ b, const float *__restrict__ d_mu,
                                                      const float *__restrict__ d_eaa, const float *__restrict__ d_eab,
                                                      float *__restrict__ grad_a)
{
    constexpr int kStageFloats = 3 * kLI * kLP;
    __shared__ __attribute__((aligned(16))) float lds[kStageFloats];   // staged maps, then (over the same bytes) row-filtered maps
    float (*sm)[kLI][kLP] = reinterpret_cast<float (*)[kLI][kLP]>(lds);
    float (*hz)[kLI][kLT + 1] = reinterpret_cast<float (*)[kLI][kLT + 1]>(lds);
    const LossTile lt = loss_tile((W + kLT - 1) / kLT, tiles_y, C, rows.ty0);
    const int ch = lt.ch, x0 = lt.x0, y0 = lt.y0;
    const size_t plane = (size_t)ch * H * W;
    {
        const float *const src[3] = {d_mu, d_eaa, d_eab};
        stage_planes<3, VEC>(src, lds, plane, H, W, x0, y0);
    }
    __syncthreads();
    float m[3][kRowW];
    const bool row_item = threadIdx.x < kRowItems;
    const int hr = threadIdx.x / (kLT / kRowW), hc0 = (threadIdx.x % (kLT / kRowW)) * kRowW;
    if (row_item) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            float v[kRowW + 16];                              // six aligned 16-byte reads (see k_loss_fwd): columns 8 q .. 8 q + 23
#pragma unroll
            for (int j = 0; j < (kRowW + 16) / 4; ++j) {
                const float4 q4 = *reinterpret_cast<const float4 *>(&sm[pl][hr][hc0 + 4 * j]);
                v[4 * j] = q4.x; v[4 * j + 1] = q4.y; v[4 * j + 2] = q4.z; v[4 * j + 3] = q4.w;
            }
#pragma unroll
            for (int o = 0; o < kRowW; ++o) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < 11; ++i) t += win.g[i] * v[kLO + o + i];
                m[pl][o] = t;
            }
        }
    }
    __syncthreads();
    if (row_item) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int o = 0; o < kRowW; ++o) hz[pl][hr][hc0 + o] = m[pl][o];
    }
    __syncthreads();
    const float up = upstream ? upstream[0] : 1.f;
    const float k_ssim = -lambda * inv_count * up, k_l1 = (1.f - lambda) * inv_count * up;
    const int cc = threadIdx.x % kLT, rr0 = (threadIdx.x / kLT) * kColH;
    float acc[3][kColH];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        float v[kColH + 10];
#pragma unroll
        for (int i = 0; i < kColH + 10; ++i) v[i] = hz[pl][rr0 + i][cc];
#pragma unroll
        for (int k = 0; k < kColH; ++k) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 11; ++i) t += win.g[i] * v[k + i];
            acc[pl][k] = t;
        }
    }
#pragma unroll
    for (int k = 0; k < kColH; ++k) {
        const int y = y0 + rr0 + k, x = x0 + cc;
        if (y >= H || x >= W || y < rows.sum_b || y >= rows.sum_e) continue;
        const size_t p = plane + (size_t)y * W + x;
        const float va = a[p], vb = b[p];
        const float d = va - vb;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        grad_a[p] = k_ssim * (acc[0][k] + 2.f * va * acc[1][k] + vb * acc[2][k]) + k_l1 * sgn;
    }
}

}  // namespace gsr

using namespace gsr;

// d mean|a - b| / da = sign(a - b) / n (sign(0) = 0, as torch.abs): the backward of the reference's l1_loss on its own
__global__ __launch_bounds__(kLBlock) void k_l1_bwd(long long n, float inv_n, const float *__restrict__ upstream, const float *__restrict__ a,
                                                    const float *__restrict__ b, float *__restrict__ g)
{
    const float s = (upstream ? upstream[0] : 1.f) * inv_n;
    for (long long i = (long long)blockIdx.x * kLBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kLBlock) {
        const float d = a[i] - b[i];
        g[i] = d > 0.f ? s : (d < 0.f ? -s : 0.f);
    }
}

extern "C" {

int gsr_loss_l1_backward(int64_t n, const float *upstream, const float *image, const float *target, float *grad_image, void *stream)
{
    if (n < 0 || (n > 0 && (!image || !target || !grad_image))) { set_error("gsr_loss_l1_backward: bad argument"); return GSR_ERR_INVALID_ARGUMENT; }
    if (n == 0) return GSR_OK;
    long long blocks = (n + kLBlock * 4 - 1) / (kLBlock * 4);
    if (blocks > 4096) blocks = 4096;
    ProfileScope prof("loss_bwd", (hipStream_t)stream);
    hipLaunchKernelGGL(k_l1_bwd, dim3((unsigned)blocks), dim3(kLBlock), 0, (hipStream_t)stream, (long long)n, 1.f / (float)n, upstream, image,
                       target, grad_image);
    GSR_LAUNCH_CHECK("l1_bwd", false, (hipStream_t)stream);
    return GSR_OK;
}

int gsr_loss_workspace_size(int32_t channels, int32_t height, int32_t width, size_t *bytes)
{
    if (channels <= 0 || height <= 0 || width <= 0 || !bytes) { set_error("gsr_loss_workspace_size: bad argument"); return GSR_ERR_INVALID_ARGUMENT; }
    const size_t n = (size_t)channels * height * width;
    const size_t blocks = (size_t)((width + kLT - 1) / kLT) * ((height + kLT - 1) / kLT) * channels;
    *bytes = 3 * align_up(n * 4) + 2 * align_up(blocks * 4);
    return GSR_OK;
}

static void carve_loss(void *ws, int C, int H, int W, float **d_mu, float **d_eaa, float **d_eab, float **ps, float **pl)
{
    char *b = (char *)ws;
    const size_t n = (size_t)C * H * W;
    const size_t blocks = (size_t)((W + kLT - 1) / kLT) * ((H + kLT - 1) / kLT) * C;
    *d_mu = (float *)b; b += align_up(n * 4);
    *d_eaa = (float *)b; b += align_up(n * 4);
    *d_eab = (float *)b; b += align_up(n * 4);
    *ps = (float *)b; b += align_up(blocks * 4);
    *pl = (float *)b;
}

// rows = NULL: whole image, out = (loss, l1, ssim).  rows given: slab, out = (sum |a - b|, sum SSIM) over the slab's rows.
static int loss_forward_impl(int32_t channels, int32_t height, int32_t width, float lambda_dssim, const float *image,
                             const float *target, void *workspace, float *out, const int32_t *rows, hipStream_t s)
{
    if (channels <= 0 || height <= 0 || width <= 0 || !image || !target || !workspace || !out) {
        set_error("gsr_loss_l1_ssim_forward: bad argument");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    LossRows lr{0, height, 0, height, 0};
    if (rows) {
        if (rows[0] < 0 || rows[1] > height || rows[0] > rows[1]) { set_error("loss rows [%d, %d) outside the image", rows[0], rows[1]); return GSR_ERR_INVALID_ARGUMENT; }
        lr.sum_b = rows[0]; lr.sum_e = rows[1];
        lr.map_b = rows[0] - kLH > 0 ? rows[0] - kLH : 0;
        lr.map_e = rows[1] + kLH < height ? rows[1] + kLH : height;
        lr.ty0 = lr.map_b / kLT;
    }
    float *d_mu, *d_eaa, *d_eab, *ps, *pl;
    carve_loss(workspace, channels, height, width, &d_mu, &d_eaa, &d_eab, &ps, &pl);
    const int tiles_y = lr.map_e > lr.map_b ? (lr.map_e + kLT - 1) / kLT - lr.ty0 : 0;
    const int nblocks = ((width + kLT - 1) / kLT) * tiles_y * channels;
    const float inv_count = 1.f / ((float)channels * (float)height * (float)width);
    const Win win = make_window();
    ProfileScope prof("loss_fwd", s);
    if (nblocks > 0) {
        const dim3 grid(nblocks);
        const bool vec = width % 4 == 0 && ((uintptr_t)image % 16 == 0) && ((uintptr_t)target % 16 == 0);
        if (vec) hipLaunchKernelGGL(k_loss_fwd<true>, grid, dim3(kLBlock), 0, s, channels, height, width, lr, tiles_y, win, image, target, d_mu, d_eaa, d_eab, ps, pl);
        else hipLaunchKernelGGL(k_loss_fwd<false>, grid, dim3(kLBlock), 0, s, channels, height, width, lr, tiles_y, win, image, target, d_mu, d_eaa, d_eab, ps, pl);
    }
    hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(kLBlock), 0, s, nblocks, inv_count, lambda_dssim, rows ? 1 : 0, ps, pl, out);
    GSR_LAUNCH_CHECK("loss_fwd", false, s);
    return GSR_OK;
}

static int loss_backward_impl(int32_t channels, int32_t height, int32_t width, float lambda_dssim, const float *upstream,
                              const float *image, const float *target, const void *workspace, float *grad_image,
                              const int32_t *rows, hipStream_t s)
{
    if (channels <= 0 || height <= 0 || width <= 0 || !image || !target || !workspace || !grad_image) {
        set_error("gsr_loss_l1_ssim_backward: bad argument");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    LossRows lr{0, height, 0, height, 0};
    if (rows) {
        if (rows[0] < 0 || rows[1] > height || rows[0] > rows[1]) { set_error("loss rows [%d, %d) outside the image", rows[0], rows[1]); return GSR_ERR_INVALID_ARGUMENT; }
        lr.sum_b = rows[0]; lr.sum_e = rows[1];
        lr.map_b = lr.sum_b; lr.map_e = lr.sum_e;
        lr.ty0 = lr.sum_b / kLT;
    }
    const int tiles_y = lr.sum_e > lr.sum_b ? (lr.sum_e + kLT - 1) / kLT - lr.ty0 : 0;
    if (tiles_y == 0) return GSR_OK;
    float *d_mu, *d_eaa, *d_eab, *ps, *pl;
    carve_loss(const_cast<void *>(workspace), channels, height, width, &d_mu, &d_eaa, &d_eab, &ps, &pl);
    const dim3 grid(((width + kLT - 1) / kLT) * tiles_y * channels);
    const float inv_count = 1.f / ((float)channels * (float)height * (float)width);
    const Win win = make_window();
    ProfileScope prof("loss_bwd", s);
    // the derivative maps are 256-byte aligned planes of the workspace: float4 rows whenever the width allows
    const bool vec = width % 4 == 0 && ((uintptr_t)workspace % 16 == 0) && (((size_t)channels * height * width * 4) % 16 == 0);
    if (vec)
        hipLaunchKernelGGL(k_loss_bwd<true>, grid, dim3(kLBlock), 0, s, channels, height, width, lr, tiles_y, win, inv_count, lambda_dssim,
                           upstream, image, target, d_mu, d_eaa, d_eab, grad_image);
    else
        hipLaunchKernelGGL(k_loss_bwd<false>, grid, dim3(kLBlock), 0, s, channels, height, width, lr, tiles_y, win, inv_count, lambda_dssim,
                           upstream, image, target, d_mu, d_eaa, d_eab, grad_image);
    GSR_LAUNCH_CHECK("loss_bwd", false, s);
    return GSR_OK;
}

int gsr_loss_l1_ssim_forward(int32_t channels, int32_t height, int32_t width, float lambda_dssim, const float *image,
                             const float *target, void *workspace, float *out3, void *stream)
{
    return loss_forward_impl(channels, height, width, lambda_dssim, image, target, workspace, out3, nullptr, (hipStream_t)stream);
}

int gsr_loss_l1_ssim_backward(int32_t channels, int32_t height, int32_t width, float lambda_dssim, const float *upstream,
                              const float *image, const float *target, const void *workspace, float *grad_image, void *stream)
{
    return loss_backward_impl(channels, height, width, lambda_dssim, upstream, image, target, workspace, grad_image, nullptr,
                              (hipStream_t)stream);
}

int gsr_loss_l1_ssim_forward_rows(int32_t channels, int32_t height, int32_t width, const float *image, const float *target,
                                  void *workspace, float *out2, int32_t row_begin, int32_t row_end, void *stream)
{
    const int32_t rows[2] = {row_begin, row_end};
    return loss_forward_impl(channels, height, width, 0.f, image, target, workspace, out2, rows, (hipStream_t)stream);
}

int gsr_loss_l1_ssim_backward_rows(int32_t channels, int32_t height, int32_t width, float lambda_dssim, const float *upstream,
                                   const float *image, const float *target, const void *workspace, float *grad_image,
                                   int32_t row_begin, int32_t row_end, void *stream)
{
    const int32_t rows[2] = {row_begin, row_end};
    return loss_backward_impl(channels, height, width, lambda_dssim, upstream, image, target, workspace, grad_image, rows,
                              (hipStream_t)stream);
}

}  // extern "C"
