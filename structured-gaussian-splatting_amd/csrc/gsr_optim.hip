// gsr_optim.hip — SURVEY 8f row f1: the optimizer step of the reference's training loop (train.py:140-142,
// torch.optim.Adam with eps = 1e-15, scene/gaussian_model.py:159-177) as ONE pass over (param, grad, m, v):
// 16 B read + 12 B written per element, where the foreach implementation makes ~8 passes.
// Semantics of torch.optim.Adam (no weight decay, no amsgrad):
//   m <- m + (g - m)(1 - b1);  v <- b2 v + (1 - b2) g^2;  p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "gsr_internal.h"

namespace gsr {

constexpr int kAdamBlock = 256;

// SPLIT: rows of `row_len` elements (>= 4) whose first `split` elements step with `step_size` and the rest
// with `step_tail`: the interleaved SH table [P, 16, 3] of this package's GaussianModel, where the reference keeps
// f_dc (lr = feature_lr) and f_rest (lr = feature_lr / 20) as two tensors (scene/gaussian_model.py:166-168).
template <bool SPLIT>
__global__ __launch_bounds__(kAdamBlock) void k_adam(size_t n, float *__restrict__ p, const float *__restrict__ g,
                                                     float *__restrict__ m, float *__restrict__ v, float one_minus_b1,
                                                     float b2, float one_minus_b2, float step_size, float inv_bc2_sqrt, float eps,
                                                     uint32_t row_len, uint32_t split, float step_tail)
{
    const size_t n4 = n / 4;
    const size_t stride = (size_t)gridDim.x * kAdamBlock;
    for (size_t i = (size_t)blockIdx.x * kAdamBlock + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4 *>(p)[i], mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
        const float4 gg = reinterpret_cast<const float4 *>(g)[i];
        float4 st = make_float4(step_size, step_size, step_size, step_size);
        if (SPLIT) {
            const uint32_t col = (uint32_t)((i * 4) % row_len);
            auto pick = [&](uint32_t c) { c = c >= row_len ? c - row_len : c; return c < split ? step_size : step_tail; };
            st = make_float4(pick(col), pick(col + 1), pick(col + 2), pick(col + 3));
        }
#define GSR_ADAM1(c)                                                          \
        mm.c = mm.c + (gg.c - mm.c) * one_minus_b1;                           \
        vv.c = vv.c * b2 + one_minus_b2 * gg.c * gg.c;                        \
        pp.c = pp.c - st.c * (mm.c / (sqrtf(vv.c) * inv_bc2_sqrt + eps));
        GSR_ADAM1(x) GSR_ADAM1(y) GSR_ADAM1(z) GSR_ADAM1(w)
        reinterpret_cast<float4 *>(p)[i] = pp; reinterpret_cast<float4 *>(m)[i] = mm; reinterpret_cast<float4 *>(v)[i] = vv;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * kAdamBlock + threadIdx.x; i < n; i += stride) {
        float mm = m[i], vv = v[i];
        const float gg = g[i];
        mm = mm + (gg - mm) * one_minus_b1;
        vv = vv * b2 + one_minus_b2 * gg * gg;
        const float st = SPLIT && (uint32_t)(i % row_len) >= split ? step_tail : step_size;
        p[i] = p[i] - st * (mm / (sqrtf(vv) * inv_bc2_sqrt + eps));
        m[i] = mm; v[i] = vv;
    }
#undef GSR_ADAM1
}

// ---- every tensor of the optimizer in one launch.  Work item = kMultiChunk consecutive float4 of one tensor (a block walks
// items blockIdx.x, + gridDim.x, ...; each thread takes kMultiPer float4 of the item 256 apart: all of a thread's 4 kMultiPer
// loads are issued before the first store).  Five launches of the one-tensor kernel left four kernel boundaries and four tails
// in a 0.35 ms step.
constexpr int kMultiPer = 4;
constexpr int kMultiChunk = kAdamBlock * kMultiPer;          // float4 per work item
struct AdamMulti {
    float *p[GSR_ADAM_MAX_TENSORS]; const float *g[GSR_ADAM_MAX_TENSORS]; float *m[GSR_ADAM_MAX_TENSORS]; float *v[GSR_ADAM_MAX_TENSORS];
    unsigned long long n[GSR_ADAM_MAX_TENSORS];
    unsigned long long item_end[GSR_ADAM_MAX_TENSORS];       // running count of work items
    float step_head[GSR_ADAM_MAX_TENSORS], step_tail[GSR_ADAM_MAX_TENSORS], inv_bc2_sqrt[GSR_ADAM_MAX_TENSORS];
    uint32_t row_len[GSR_ADAM_MAX_TENSORS], split[GSR_ADAM_MAX_TENSORS];
    int count;
};

__global__ __launch_bounds__(kAdamBlock) void k_adam_multi(AdamMulti a, float one_minus_b1, float b2, float one_minus_b2, float eps)
{
    const unsigned long long items = a.item_end[a.count - 1];
    for (unsigned long long it = blockIdx.x; it < items; it += gridDim.x) {
        int t = 0;
        while (it >= a.item_end[t]) ++t;                               // block-uniform, <= 8 steps
        const unsigned long long first = (it - (t ? a.item_end[t - 1] : 0ull)) * kMultiChunk;      // float4 index inside tensor t
        const unsigned long long n = a.n[t], n4 = n / 4;
        float4 *p4 = reinterpret_cast<float4 *>(a.p[t]), *m4 = reinterpret_cast<float4 *>(a.m[t]), *v4 = reinterpret_cast<float4 *>(a.v[t]);
        const float4 *g4 = reinterpret_cast<const float4 *>(a.g[t]);
        const float sh = a.step_head[t], stl = a.step_tail[t], ib = a.inv_bc2_sqrt[t];
        const uint32_t row_len = a.row_len[t], split = a.split[t];
        float4 pp[kMultiPer], mm[kMultiPer], vv[kMultiPer], gg[kMultiPer];
#pragma unroll
        for (int k = 0; k < kMultiPer; ++k) {
            const unsigned long long i = first + threadIdx.x + (unsigned long long)k * kAdamBlock;
            if (i < n4) { pp[k] = p4[i]; mm[k] = m4[i]; vv[k] = v4[i]; gg[k] = g4[i]; }
        }
#pragma unroll
        for (int k = 0; k < kMultiPer; ++k) {
            const unsigned long long i = first + threadIdx.x + (unsigned long long)k * kAdamBlock;
            if (i >= n4) continue;
            float4 st = make_float4(sh, sh, sh, sh);
            if (row_len) {
                const uint32_t col = (uint32_t)((i * 4) % row_len);
                auto pick = [&](uint32_t c) { c = c >= row_len ? c - row_len : c; return c < split ? sh : stl; };
                st = make_float4(pick(col), pick(col + 1), pick(col + 2), pick(col + 3));
            }
#define GSR_ADAM1(c)                                                                      \
            mm[k].c = mm[k].c + (gg[k].c - mm[k].c) * one_minus_b1;                       \
            vv[k].c = vv[k].c * b2 + one_minus_b2 * gg[k].c * gg[k].c;                    \
            pp[k].c = pp[k].c - st.c * (mm[k].c / (sqrtf(vv[k].c) * ib + eps));
            GSR_ADAM1(x) GSR_ADAM1(y) GSR_ADAM1(z) GSR_ADAM1(w)
#undef GSR_ADAM1
            p4[i] = pp[k]; m4[i] = mm[k]; v4[i] = vv[k];
        }
        // the tensor's last item also takes the n % 4 trailing elements
        if (first <= n4 && first + kMultiChunk > n4 && threadIdx.x < (unsigned)(n - n4 * 4)) {
            const unsigned long long i = n4 * 4 + threadIdx.x;
            float m1 = a.m[t][i], v1 = a.v[t][i];
            const float g1 = a.g[t][i];
            m1 = m1 + (g1 - m1) * one_minus_b1;
            v1 = v1 * b2 + one_minus_b2 * g1 * g1;
            const float st = row_len && (uint32_t)(i % row_len) >= split ? stl : sh;
            a.p[t][i] = a.p[t][i] - st * (m1 / (sqrtf(v1) * ib + eps));
            a.m[t][i] = m1; a.v[t][i] = v1;
        }
    }
}

// ---- densification bookkeeping of one training iteration (train.py:127-130 + scene/gaussian_model.py:415-417):
//   max_radii2D[vis] = max(max_radii2D[vis], radii[vis]);  xyz_gradient_accum[vis] += |viewspace grad.xy|;  denom[vis] += 1
// with vis = radii > 0.  One pass and no host synchronisation, where boolean-mask indexing costs four nonzero()
// compactions (each a device->host size readback) and ~20 kernels.
__global__ __launch_bounds__(256) void k_densify_stats(int P, const int32_t *__restrict__ radii, const float *__restrict__ vs_grad,
                                                       float *__restrict__ max_radii2D, float *__restrict__ grad_accum,
                                                       float *__restrict__ denom)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
    const float gx = vs_grad[3 * (size_t)i], gy = vs_grad[3 * (size_t)i + 1];
    grad_accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.f;
}

}  // namespace gsr

using namespace gsr;

extern "C" int gsr_densify_stats(int32_t P, const int32_t *radii, const float *viewspace_grad, float *max_radii2D,
                                 float *xyz_gradient_accum, float *denom, void *stream)
{
    if (P < 0 || (P > 0 && (!radii || !viewspace_grad || !max_radii2D || !xyz_gradient_accum || !denom))) {
        set_error("gsr_densify_stats: bad argument");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return GSR_OK;
    hipStream_t s = (hipStream_t)stream;
    ProfileScope prof("densify_stats", s);
    hipLaunchKernelGGL(k_densify_stats, dim3((P + 255) / 256), dim3(256), 0, s, P, radii, viewspace_grad, max_radii2D,
                       xyz_gradient_accum, denom);
    GSR_LAUNCH_CHECK("densify_stats", false, s);
    return GSR_OK;
}

extern "C" int gsr_adam_step(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float lr, float beta1,
                             float beta2, float eps, int64_t step, void *stream)
{
    if (n < 0 || step < 1 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) {
        set_error("gsr_adam_step: bad argument");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    if (n == 0) return GSR_OK;
    hipStream_t s = (hipStream_t)stream;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    size_t blocks = ((size_t)n / 4 + kAdamBlock - 1) / kAdamBlock;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    ProfileScope prof("adam", s);
    hipLaunchKernelGGL(k_adam<false>, dim3((unsigned)blocks), dim3(kAdamBlock), 0, s, (size_t)n, param, grad, exp_avg, exp_avg_sq, 1.f - beta1,
                       beta2, 1.f - beta2, step_size, inv_bc2_sqrt, eps, 0u, 0u, 0.f);
    GSR_LAUNCH_CHECK("adam", false, s);
    return GSR_OK;
}

extern "C" int gsr_adam_step_split(int64_t rows, int32_t row_len, int32_t split, float *param, const float *grad, float *exp_avg,
                                   float *exp_avg_sq, float lr_head, float lr_tail, float beta1, float beta2, float eps, int64_t step,
                                   void *stream)
{
    if (rows < 0 || step < 1 || row_len < 4 || split < 0 || split > row_len ||
        (rows > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) {
        set_error("gsr_adam_step_split: bad argument (row_len >= 4, 0 <= split <= row_len)");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    if (rows == 0) return GSR_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)rows * (size_t)row_len;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_head = (float)((double)lr_head / bc1), step_tail = (float)((double)lr_tail / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    size_t blocks = (n / 4 + kAdamBlock - 1) / kAdamBlock;
    if (blocks > 4096) blocks = 4096;
    ProfileScope prof("adam", s);
    hipLaunchKernelGGL(k_adam<true>, dim3((unsigned)blocks), dim3(kAdamBlock), 0, s, n, param, grad, exp_avg, exp_avg_sq, 1.f - beta1, beta2,
                       1.f - beta2, step_head, inv_bc2_sqrt, eps, (uint32_t)row_len, (uint32_t)split, step_tail);
    GSR_LAUNCH_CHECK("adam", false, s);
    return GSR_OK;
}

extern "C" int gsr_adam_step_multi(int32_t count, const gsr_adam_tensor *tensors, float beta1, float beta2, float eps, void *stream)
{
    if (count < 0 || count > GSR_ADAM_MAX_TENSORS || (count > 0 && !tensors)) {
        set_error("gsr_adam_step_multi: 0 .. %d tensors", GSR_ADAM_MAX_TENSORS);
        return GSR_ERR_INVALID_ARGUMENT;
    }
    AdamMulti a;
    a.count = 0;
    unsigned long long items = 0;
    for (int i = 0; i < count; ++i) {
        const gsr_adam_tensor &t = tensors[i];
        if (t.n < 0 || t.step < 1 || (t.n > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq)) ||
            (t.row_len != 0 && (t.row_len < 4 || t.split < 0 || t.split > t.row_len || t.n % t.row_len != 0))) {
            set_error("gsr_adam_step_multi: bad tensor %d (row_len 0 or >= 4 and dividing n, 0 <= split <= row_len, step >= 1)", i);
            return GSR_ERR_INVALID_ARGUMENT;
        }
        if (t.n == 0) continue;
        const int k = a.count++;
        const double bc1 = 1.0 - pow((double)beta1, (double)t.step), bc2 = 1.0 - pow((double)beta2, (double)t.step);
        a.p[k] = t.param; a.g[k] = t.grad; a.m[k] = t.exp_avg; a.v[k] = t.exp_avg_sq;
        a.n[k] = (unsigned long long)t.n;
        a.step_head[k] = (float)((double)t.lr / bc1); a.step_tail[k] = (float)((double)t.lr_tail / bc1);
        a.inv_bc2_sqrt[k] = (float)(1.0 / sqrt(bc2));
        a.row_len[k] = (uint32_t)t.row_len; a.split[k] = (uint32_t)t.split;
        items += ((unsigned long long)t.n / 4 + kMultiChunk - 1) / kMultiChunk;
        if (((unsigned long long)t.n / 4) % kMultiChunk == 0 && t.n % 4 != 0) items += 1;      // an item for the trailing elements alone
        a.item_end[k] = items;
    }
    if (a.count == 0) return GSR_OK;
    for (int k = a.count; k < GSR_ADAM_MAX_TENSORS; ++k) { a.item_end[k] = items; a.n[k] = 0; a.p[k] = nullptr; a.g[k] = nullptr; a.m[k] = nullptr; a.v[k] = nullptr; }
    hipStream_t s = (hipStream_t)stream;
    unsigned long long blocks = items < 16384 ? items : 16384;
    ProfileScope prof("adam", s);
    hipLaunchKernelGGL(k_adam_multi, dim3((unsigned)blocks), dim3(kAdamBlock), 0, s, a, 1.f - beta1, beta2, 1.f - beta2, eps);
    GSR_LAUNCH_CHECK("adam", false, s);
    return GSR_OK;
}
