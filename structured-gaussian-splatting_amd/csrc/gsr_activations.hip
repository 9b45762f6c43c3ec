// gsr_activations.hip — SURVEY 8a row a14: the activations between the optimizer's raw parameters and the
// rasterizer's inputs (scene/gaussian_model.py:47-60 setup_functions, :101-125 the getters):
//   get_scaling  = exp(_scaling)            [P,3]
//   get_rotation = normalize(_rotation)     [P,4]   v / max(|v|_2, 1e-12)   (torch.nn.functional.normalize defaults)
//   get_opacity  = sigmoid(_opacity)        [P,1]
// and their backward, each as ONE pass over the three tensors (8 floats per Gaussian in, 8 out) where torch runs
// ~8 kernels forward and ~12 backward.  HBM-bound: 64 B per Gaussian forward, 112 B backward.
// (get_features = torch.cat needs no kernel: the parameter store keeps f_dc and f_rest interleaved, see
// scene/gaussian_model.py in this package.)
#include "gsr_internal.h"

namespace gsr {

constexpr int kActBlock = 256;
constexpr float kNormalizeEps = 1e-12f;

// The three tensors are walked as flat float4 streams by three block ranges of one grid: [0, nb_s) the 3P scale
// elements, [nb_s, nb_s + nb_o) the P opacities, the rest the P quaternions (one float4 each).
struct ActGrid { unsigned nb_s, nb_o, nb_r; };

__host__ ActGrid act_grid(int64_t P)
{
    auto blocks = [](int64_t n4) { int64_t b = (n4 + kActBlock - 1) / kActBlock; return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); };
    ActGrid g;
    g.nb_s = blocks((3 * P + 3) / 4); g.nb_o = blocks((P + 3) / 4); g.nb_r = blocks(P);
    return g;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(kActBlock) void k_act_fwd(int64_t P, ActGrid grid, const float *__restrict__ scaling, const float *__restrict__ rotation,
                                                       const float *__restrict__ opacity, float *__restrict__ scales,
                                                       float *__restrict__ rotations, float *__restrict__ opacities)
{
    unsigned b = blockIdx.x;
    if (b < grid.nb_s) {
        if (!scaling) return;
        const int64_t n = 3 * P, n4 = n / 4, stride = (int64_t)grid.nb_s * kActBlock;
        for (int64_t i = (int64_t)b * kActBlock + threadIdx.x; i < n4; i += stride) {
            const float4 v = reinterpret_cast<const float4 *>(scaling)[i];
            reinterpret_cast<float4 *>(scales)[i] = make_float4(expf(v.x), expf(v.y), expf(v.z), expf(v.w));
        }
        if (b == 0 && (int64_t)threadIdx.x < n - n4 * 4) scales[n4 * 4 + threadIdx.x] = expf(scaling[n4 * 4 + threadIdx.x]);
        return;
    }
    b -= grid.nb_s;
    if (b < grid.nb_o) {
        if (!opacity) return;
        const int64_t n = P, n4 = n / 4, stride = (int64_t)grid.nb_o * kActBlock;
        for (int64_t i = (int64_t)b * kActBlock + threadIdx.x; i < n4; i += stride) {
            const float4 v = reinterpret_cast<const float4 *>(opacity)[i];
            reinterpret_cast<float4 *>(opacities)[i] = make_float4(sigmoidf(v.x), sigmoidf(v.y), sigmoidf(v.z), sigmoidf(v.w));
        }
        if (b == 0 && (int64_t)threadIdx.x < n - n4 * 4) opacities[n4 * 4 + threadIdx.x] = sigmoidf(opacity[n4 * 4 + threadIdx.x]);
        return;
    }
    b -= grid.nb_o;
    if (!rotation) return;
    const int64_t stride = (int64_t)grid.nb_r * kActBlock;
    for (int64_t i = (int64_t)b * kActBlock + threadIdx.x; i < P; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(rotation)[i];
        const float nc = fmaxf(sqrtf(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w), kNormalizeEps);
        reinterpret_cast<float4 *>(rotations)[i] = make_float4(v.x / nc, v.y / nc, v.z / nc, v.w / nc);      // a division, as torch does
    }
}

// backward:  d exp = g * s;   d sigmoid = g * o (1 - o);   d normalize = g / n_c - v (v . g) / (n n_c^2) for n >= eps
// (n_c = max(n, eps); below eps the clamp has no gradient and only g / eps remains), as autograd derives it from
// norm -> clamp_min -> expand -> div.
__global__ __launch_bounds__(kActBlock) void k_act_bwd(int64_t P, ActGrid grid, const float *__restrict__ scales, const float *__restrict__ rotation,
                                                       const float *__restrict__ opacities, const float *__restrict__ g_scales,
                                                       const float *__restrict__ g_rotations, const float *__restrict__ g_opacities,
                                                       float *__restrict__ d_scaling, float *__restrict__ d_rotation,
                                                       float *__restrict__ d_opacity)
{
    unsigned b = blockIdx.x;
    if (b < grid.nb_s) {
        if (!d_scaling) return;
        const int64_t n = 3 * P, n4 = n / 4, stride = (int64_t)grid.nb_s * kActBlock;
        for (int64_t i = (int64_t)b * kActBlock + threadIdx.x; i < n4; i += stride) {
            const float4 s = reinterpret_cast<const float4 *>(scales)[i], g = reinterpret_cast<const float4 *>(g_scales)[i];
            reinterpret_cast<float4 *>(d_scaling)[i] = make_float4(g.x * s.x, g.y * s.y, g.z * s.z, g.w * s.w);
        }
        if (b == 0 && (int64_t)threadIdx.x < n - n4 * 4) {
            const int64_t i = n4 * 4 + threadIdx.x;
            d_scaling[i] = g_scales[i] * scales[i];
        }
        return;
    }
    b -= grid.nb_s;
    if (b < grid.nb_o) {
        if (!d_opacity) return;
        const int64_t n = P, n4 = n / 4, stride = (int64_t)grid.nb_o * kActBlock;
        for (int64_t i = (int64_t)b * kActBlock + threadIdx.x; i < n4; i += stride) {
            const float4 o = reinterpret_cast<const float4 *>(opacities)[i], g = reinterpret_cast<const float4 *>(g_opacities)[i];
            reinterpret_cast<float4 *>(d_opacity)[i] = make_float4(g.x * (1.f - o.x) * o.x, g.y * (1.f - o.y) * o.y, g.z * (1.f - o.z) * o.z,
                                                                   g.w * (1.f - o.w) * o.w);
        }
        if (b == 0 && (int64_t)threadIdx.x < n - n4 * 4) {
            const int64_t i = n4 * 4 + threadIdx.x;
            d_opacity[i] = g_opacities[i] * (1.f - opacities[i]) * opacities[i];
        }
        return;
    }
    b -= grid.nb_o;
    if (!d_rotation) return;
    const int64_t stride = (int64_t)grid.nb_r * kActBlock;
    for (int64_t i = (int64_t)b * kActBlock + threadIdx.x; i < P; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(rotation)[i], g = reinterpret_cast<const float4 *>(g_rotations)[i];
        const float n = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w), nc = fmaxf(n, kNormalizeEps), inv = 1.f / nc;
        const float dot = v.x * g.x + v.y * g.y + v.z * g.z + v.w * g.w;
        const float k = n >= kNormalizeEps ? dot * inv * inv / n : 0.f;       // clamp_min passes the gradient where n >= eps
        reinterpret_cast<float4 *>(d_rotation)[i] = make_float4(g.x * inv - v.x * k, g.y * inv - v.y * k, g.z * inv - v.z * k, g.w * inv - v.w * k);
    }
}

}  // namespace gsr

using namespace gsr;

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" int gsr_activations_forward(int64_t P, const float *scaling_raw, const float *rotation_raw, const float *opacity_raw,
                                       float *scales, float *rotations, float *opacities, void *stream)
{
    if (P < 0 || (!scaling_raw) != (!scales) || (!rotation_raw) != (!rotations) || (!opacity_raw) != (!opacities)) {
        set_error("gsr_activations_forward: every input needs its output (and only those)");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    if (!aligned16(scaling_raw) || !aligned16(rotation_raw) || !aligned16(opacity_raw) || !aligned16(scales) || !aligned16(rotations) ||
        !aligned16(opacities)) {
        set_error("gsr_activations_forward: tensors must be 16-byte aligned");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return GSR_OK;
    hipStream_t s = (hipStream_t)stream;
    const ActGrid g = act_grid(P);
    ProfileScope prof("activations_fwd", s);
    hipLaunchKernelGGL(k_act_fwd, dim3(g.nb_s + g.nb_o + g.nb_r), dim3(kActBlock), 0, s, P, g, scaling_raw, rotation_raw, opacity_raw, scales,
                       rotations, opacities);
    GSR_LAUNCH_CHECK("activations_fwd", false, s);
    return GSR_OK;
}

extern "C" int gsr_activations_backward(int64_t P, const float *scales, const float *rotation_raw, const float *opacities,
                                        const float *dL_dscales, const float *dL_drotations, const float *dL_dopacities,
                                        float *dL_dscaling_raw, float *dL_drotation_raw, float *dL_dopacity_raw, void *stream)
{
    if (P < 0 || (dL_dscaling_raw && (!scales || !dL_dscales)) || (dL_drotation_raw && (!rotation_raw || !dL_drotations)) ||
        (dL_dopacity_raw && (!opacities || !dL_dopacities))) {
        set_error("gsr_activations_backward: a wanted gradient needs the forward tensor and the incoming gradient");
        return GSR_ERR_INVALID_ARGUMENT;
    }
    const void *all[] = {scales, rotation_raw, opacities, dL_dscales, dL_drotations, dL_dopacities, dL_dscaling_raw, dL_drotation_raw,
                         dL_dopacity_raw};
    for (const void *p : all)
        if (!aligned16(p)) { set_error("gsr_activations_backward: tensors must be 16-byte aligned"); return GSR_ERR_INVALID_ARGUMENT; }
    if (P == 0) return GSR_OK;
    hipStream_t s = (hipStream_t)stream;
    const ActGrid g = act_grid(P);
    ProfileScope prof("activations_bwd", s);
    hipLaunchKernelGGL(k_act_bwd, dim3(g.nb_s + g.nb_o + g.nb_r), dim3(kActBlock), 0, s, P, g, scales, rotation_raw, opacities, dL_dscales,
                       dL_drotations, dL_dopacities, dL_dscaling_raw, dL_drotation_raw, dL_dopacity_raw);
    GSR_LAUNCH_CHECK("activations_bwd", false, s);
    return GSR_OK;
}
