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

__global__ __launch_bounds__(kLBlock) void k_loss_fwd(int H, int W, Win win, const float *__restrict__ a, const float *__restrict__ b,
                                                      float *__restrict__ d_mu, float *__restrict__ d_eaa, float *__restrict__ d_eab,
                                                      float *__restrict__ partial_ssim, float *__restrict__ partial_l1)
{
    // LDS is used twice: first the staged tile + halo of a and b, then (after every thread holds its row-filtered
    // values in registers) the five row-filtered moment planes OVER the same bytes: 27.7 KB per block instead of
    // 42 KB, i.e. 5 resident blocks per CU instead of 3.
    constexpr int kStageFloats = 2 * kLI * (kLI + 1);
    constexpr int kHzFloats = 5 * kLI * (kLT + 1);
    __shared__ float lds[kHzFloats > kStageFloats ? kHzFloats : kStageFloats];
    __shared__ float red[4];
    float (*sa)[kLI + 1] = reinterpret_cast<float (*)[kLI + 1]>(lds);
    float (*sb)[kLI + 1] = reinterpret_cast<float (*)[kLI + 1]>(lds + kLI * (kLI + 1));
    float (*hz)[kLI][kLT + 1] = reinterpret_cast<float (*)[kLI][kLT + 1]>(lds);
    const int ch = blockIdx.z, x0 = blockIdx.x * kLT, y0 = blockIdx.y * kLT;
    const size_t plane = (size_t)ch * H * W;
    for (int idx = threadIdx.x; idx < kLI * kLI; idx += kLBlock) {
        const int r = idx / kLI, c = idx % kLI;
        const int y = y0 - kLH + r, x = x0 - kLH + c;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        sa[r][c] = in ? a[plane + (size_t)y * W + x] : 0.f;
        sb[r][c] = in ? b[plane + (size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    // this thread's four output pixels: centre values for the L1 term, before the staging area is overwritten
    float ca[(kLT * kLT) / kLBlock], cbv[(kLT * kLT) / kLBlock];
#pragma unroll
    for (int k = 0; k < (kLT * kLT) / kLBlock; ++k) {
        const int idx = threadIdx.x + k * kLBlock;
        ca[k] = sa[idx / kLT + kLH][idx % kLT + kLH];
        cbv[k] = sb[idx / kLT + kLH][idx % kLT + kLH];
    }
    constexpr int kHzIters = (kLI * kLT + kLBlock - 1) / kLBlock;       // 6
    float m0[kHzIters], m1[kHzIters], m2[kHzIters], m3[kHzIters], m4[kHzIters];
#pragma unroll
    for (int it = 0; it < kHzIters; ++it) {
        const int idx = threadIdx.x + it * kLBlock;
        m0[it] = m1[it] = m2[it] = m3[it] = m4[it] = 0.f;
        if (idx < kLI * kLT) {
            const int r = idx / kLT, c = idx % kLT;
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                const float g = win.g[i], va = sa[r][c + i], vb = sb[r][c + i];
                m0[it] += g * va; m1[it] += g * vb; m2[it] += g * va * va; m3[it] += g * vb * vb; m4[it] += g * va * vb;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kHzIters; ++it) {
        const int idx = threadIdx.x + it * kLBlock;
        if (idx < kLI * kLT) {
            const int r = idx / kLT, c = idx % kLT;
            hz[0][r][c] = m0[it]; hz[1][r][c] = m1[it]; hz[2][r][c] = m2[it]; hz[3][r][c] = m3[it]; hz[4][r][c] = m4[it];
        }
    }
    __syncthreads();
    float s_ssim = 0.f, s_l1 = 0.f;
#pragma unroll
    for (int k = 0; k < (kLT * kLT) / kLBlock; ++k) {
        const int idx = threadIdx.x + k * kLBlock;
        const int r = idx / kLT, c = idx % kLT;
        const int y = y0 + r, x = x0 + c;
        float mu1 = 0.f, mu2 = 0.f, eaa = 0.f, ebb = 0.f, eab = 0.f;
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const float g = win.g[i];
            mu1 += g * hz[0][r + i][c]; mu2 += g * hz[1][r + i][c];
            eaa += g * hz[2][r + i][c]; ebb += g * hz[3][r + i][c]; eab += g * hz[4][r + i][c];
        }
        if (y < H && x < W) {
            const float s1 = eaa - mu1 * mu1, s2 = ebb - mu2 * mu2, s12 = eab - mu1 * mu2;
            const float A1 = 2.f * mu1 * mu2 + kC1, A2 = 2.f * s12 + kC2;
            const float B1 = mu1 * mu1 + mu2 * mu2 + kC1, B2 = s1 + s2 + kC2;
            const float inv = 1.f / (B1 * B2);
            const float m = A1 * A2 * inv;
            s_ssim += m;
            s_l1 += fabsf(ca[k] - cbv[k]);
            if (d_mu) {
                const size_t p = plane + (size_t)y * W + x;
                d_mu[p] = 2.f * mu2 * (A2 - A1) * inv - 2.f * mu1 * m * (B2 - B1) * inv;
                d_eaa[p] = -m / B2;
                d_eab[p] = 2.f * A1 * inv;
            }
        }
    }
    const float t_ssim = block_sum(s_ssim, red);
    const float t_l1 = block_sum(s_l1, red);
    if (threadIdx.x == 0) {
        const int bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial_ssim[bid] = t_ssim;
        partial_l1[bid] = t_l1;
    }
}

__global__ __launch_bounds__(kLBlock) void k_loss_finish(int nblocks, float inv_count, float lambda, const float *__restrict__ partial_ssim,
                                                         const float *__restrict__ partial_l1, float *__restrict__ out /*[3]*/)
{
    __shared__ float red[4];
    float s = 0.f, l = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += kLBlock) { s += partial_ssim[i]; l += partial_l1[i]; }
    const float ts = block_sum(s, red) * inv_count;
    const float tl = block_sum(l, red) * inv_count;
    if (threadIdx.x == 0) {
        out[0] = (1.f - lambda) * tl + lambda * (1.f - ts);
        out[1] = tl;
        out[2] = ts;
    }
}

__global__ __launch_bounds__(kLBlock) void k_loss_bwd(int H, int W, Win win, float inv_count, float lambda,
                                                      const float *__restrict__ upstream, const float *__restrict__ a,
                                                      const float *__restrict__ b, const float *__restrict__ d_mu,
                                                      const float *__restrict__ d_eaa, const float *__restrict__ d_eab,
                                                      float *__restrict__ grad_a)
{
    constexpr int kStageFloats = 3 * kLI * (kLI + 1);
    __shared__ float lds[kStageFloats];                              // staged maps, then (over the same bytes) row-filtered maps
    float (*sm)[kLI][kLI + 1] = reinterpret_cast<float (*)[kLI][kLI + 1]>(lds);
    float (*hz)[kLI][kLT + 1] = reinterpret_cast<float (*)[kLI][kLT + 1]>(lds);
    const int ch = blockIdx.z, x0 = blockIdx.x * kLT, y0 = blockIdx.y * kLT;
    const size_t plane = (size_t)ch * H * W;
    for (int idx = threadIdx.x; idx < kLI * kLI; idx += kLBlock) {
        const int r = idx / kLI, c = idx % kLI;
        const int y = y0 - kLH + r, x = x0 - kLH + c;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const size_t p = plane + (size_t)y * W + x;
        sm[0][r][c] = in ? d_mu[p] : 0.f;
        sm[1][r][c] = in ? d_eaa[p] : 0.f;
        sm[2][r][c] = in ? d_eab[p] : 0.f;
    }
    __syncthreads();
    constexpr int kHzIters = (kLI * kLT + kLBlock - 1) / kLBlock;
    float m0[kHzIters], m1[kHzIters], m2[kHzIters];
#pragma unroll
    for (int it = 0; it < kHzIters; ++it) {
        const int idx = threadIdx.x + it * kLBlock;
        m0[it] = m1[it] = m2[it] = 0.f;
        if (idx < kLI * kLT) {
            const int r = idx / kLT, c = idx % kLT;
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                const float g = win.g[i];
                m0[it] += g * sm[0][r][c + i]; m1[it] += g * sm[1][r][c + i]; m2[it] += g * sm[2][r][c + i];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kHzIters; ++it) {
        const int idx = threadIdx.x + it * kLBlock;
        if (idx < kLI * kLT) {
            const int r = idx / kLT, c = idx % kLT;
            hz[0][r][c] = m0[it]; hz[1][r][c] = m1[it]; hz[2][r][c] = m2[it];
        }
    }
    __syncthreads();
    const float up = upstream ? upstream[0] : 1.f;
    const float k_ssim = -lambda * inv_count * up, k_l1 = (1.f - lambda) * inv_count * up;
#pragma unroll
    for (int k = 0; k < (kLT * kLT) / kLBlock; ++k) {
        const int idx = threadIdx.x + k * kLBlock;
        const int r = idx / kLT, c = idx % kLT;
        const int y = y0 + r, x = x0 + c;
        if (y >= H || x >= W) continue;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const float g = win.g[i];
            c0 += g * hz[0][r + i][c]; c1 += g * hz[1][r + i][c]; c2 += g * hz[2][r + i][c];
        }
        const size_t p = plane + (size_t)y * W + x;
        const float va = a[p], vb = b[p];
        const float d = va - vb;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        grad_a[p] = k_ssim * (c0 + 2.f * va * c1 + vb * c2) + k_l1 * sgn;
    }
}

}  // namespace gsr

using namespace gsr;

extern "C" {

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

int gsr_loss_l1_ssim_forward(int32_t channels, int32_t height, int32_t width, float lambda_dssim, const float *image,
                             const float *target, void *workspace, float *out3, void *stream)
{
    if (channels <= 0 || height <= 0 || width <= 0 || !image || !target || !workspace || !out3) {
        set_error("gsr_loss_l1_ssim_forward: bad argument");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = (hipStream_t)stream;
    float *d_mu, *d_eaa, *d_eab, *ps, *pl;
    carve_loss(workspace, channels, height, width, &d_mu, &d_eaa, &d_eab, &ps, &pl);
    const dim3 grid((width + kLT - 1) / kLT, (height + kLT - 1) / kLT, channels);
    const int nblocks = (int)(grid.x * grid.y * grid.z);
    const float inv_count = 1.f / ((float)channels * (float)height * (float)width);
    const Win win = make_window();
    {
        ProfileScope prof("loss_fwd", s);
        hipLaunchKernelGGL(k_loss_fwd, grid, dim3(kLBlock), 0, s, height, width, win, image, target, d_mu, d_eaa, d_eab, ps, pl);
        hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(kLBlock), 0, s, nblocks, inv_count, lambda_dssim, ps, pl, out3);
        GSR_LAUNCH_CHECK("loss_fwd", false, s);
    }
    return GSR_OK;
}

int gsr_loss_l1_ssim_backward(int32_t channels, int32_t height, int32_t width, float lambda_dssim, const float *upstream,
                              const float *image, const float *target, const void *workspace, float *grad_image, void *stream)
{
    if (channels <= 0 || height <= 0 || width <= 0 || !image || !target || !workspace || !grad_image) {
        set_error("gsr_loss_l1_ssim_backward: bad argument");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    hipStream_t s = (hipStream_t)stream;
    float *d_mu, *d_eaa, *d_eab, *ps, *pl;
    carve_loss(const_cast<void *>(workspace), channels, height, width, &d_mu, &d_eaa, &d_eab, &ps, &pl);
    const dim3 grid((width + kLT - 1) / kLT, (height + kLT - 1) / kLT, channels);
    const float inv_count = 1.f / ((float)channels * (float)height * (float)width);
    const Win win = make_window();
    {
        ProfileScope prof("loss_bwd", s);
        hipLaunchKernelGGL(k_loss_bwd, grid, dim3(kLBlock), 0, s, height, width, win, inv_count, lambda_dssim, upstream, image,
                           target, d_mu, d_eaa, d_eab, grad_image);
        GSR_LAUNCH_CHECK("loss_bwd", false, s);
    }
    return GSR_OK;
}

}  // extern "C"
