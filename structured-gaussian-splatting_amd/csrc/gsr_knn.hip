// gsr_knn.hip — SURVEY 8f row f2: the `distCUDA2` of the reference's second native dependency (simple_knn,
// call sites scene/gaussian_model.py:144-145, scene/latent_gaussian_model.py:219-220): for every point the
// MEAN of the squared distances to its 3 nearest OTHER points (A.12).  Used once, to initialise scales.
//
// Exact (not approximate) 3-NN:
//   1. bounding box of the cloud (block reduction + atomics on ordered-int floats)
//   2. 30-bit Morton code per point (10 bits per axis), stable radix sort (gsr_sort.hip) -> Morton order
//   3. axis-aligned bounds of every run of kBox = 512 consecutive points in Morton order
//   4. one thread per point: seed the best-3 list from its +-3 neighbours in Morton order, then visit every
//      box whose distance to the point is below the current 3rd-best distance and scan its points.
// Fewer than 4 points: the missing neighbours contribute 0 and the sum is still divided by 3.
#include "gsr_internal.h"

namespace gsr {

constexpr int kKnnBlock = 256;
constexpr int kBox = 512;

__device__ __forceinline__ int float_to_ordered(float f)
{
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ordered_to_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ __launch_bounds__(kKnnBlock) void k_knn_bounds_init(int *mm)
{
    if (threadIdx.x < 3) { mm[threadIdx.x] = 0x7FFFFFFF; mm[3 + threadIdx.x] = (int)0x80000000; }
}

__global__ __launch_bounds__(kKnnBlock) void k_knn_bounds(int P, const float *__restrict__ xyz, int *__restrict__ mm)
{
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = blockIdx.x * kKnnBlock + threadIdx.x; i < P; i += gridDim.x * kKnnBlock)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = xyz[3 * (size_t)i + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&mm[a], float_to_ordered(lo[a])); atomicMax(&mm[3 + a], float_to_ordered(hi[a])); }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(kKnnBlock) void k_knn_morton(int P, const float *__restrict__ xyz, const int *__restrict__ mm,
                                                          uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const int i = blockIdx.x * kKnnBlock + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = ordered_to_float(mm[a]), hi = ordered_to_float(mm[3 + a]);
        const float ext = hi - lo;
        const float t = ext > 0.f ? (xyz[3 * (size_t)i + a] - lo) / ext : 0.f;
        const uint32_t q = (uint32_t)fminf(1023.f, fmaxf(0.f, t * 1023.f));
        code |= spread10(q) << (2 - a);
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kKnnBlock) void k_knn_gather_boxes(int P, const float *__restrict__ xyz, const uint32_t *__restrict__ order,
                                                                float4 *__restrict__ sorted_pts, float *__restrict__ boxes)
{
    __shared__ float sh[6][kKnnBlock / kWave];
    const int box = blockIdx.x;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int j = threadIdx.x; j < kBox; j += kKnnBlock) {
        const int r = box * kBox + j;
        if (r < P) {
            const uint32_t i = order[r];
            const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
            sorted_pts[r] = make_float4(x, y, z, 0.f);
            lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
            hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
        if ((threadIdx.x & 63) == 0) { sh[a][threadIdx.x >> 6] = lo[a]; sh[3 + a][threadIdx.x >> 6] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sh[threadIdx.x][0];
        for (int w = 1; w < kKnnBlock / kWave; ++w) v = threadIdx.x < 3 ? fminf(v, sh[threadIdx.x][w]) : fmaxf(v, sh[threadIdx.x][w]);
        boxes[6 * box + threadIdx.x] = v;
    }
}

__device__ __forceinline__ void knn_insert(float d, float best[3])
{
    if (d < best[2]) {
        best[2] = d;
        if (best[2] < best[1]) { const float t = best[1]; best[1] = best[2]; best[2] = t; }
        if (best[1] < best[0]) { const float t = best[0]; best[0] = best[1]; best[1] = t; }
    }
}

__global__ __launch_bounds__(kKnnBlock) void k_knn_search(int P, int n_boxes, const float4 *__restrict__ pts,
                                                          const uint32_t *__restrict__ order, const float *__restrict__ boxes,
                                                          float *__restrict__ out)
{
    const int r = blockIdx.x * kKnnBlock + threadIdx.x;
    if (r >= P) return;
    const float4 p = pts[r];
    float best[3] = {3.0e38f, 3.0e38f, 3.0e38f};
    for (int j = max(0, r - 3); j <= min(P - 1, r + 3); ++j) {
        if (j == r) continue;
        const float4 q = pts[j];
        const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
        knn_insert(dx * dx + dy * dy + dz * dz, best);
    }
    for (int b = 0; b < n_boxes; ++b) {
        const float *bx = boxes + 6 * b;
        const float ex = fmaxf(0.f, fmaxf(bx[0] - p.x, p.x - bx[3]));
        const float ey = fmaxf(0.f, fmaxf(bx[1] - p.y, p.y - bx[4]));
        const float ez = fmaxf(0.f, fmaxf(bx[2] - p.z, p.z - bx[5]));
        if (ex * ex + ey * ey + ez * ez > best[2]) continue;
        const int j0 = b * kBox, j1 = min(P, j0 + kBox);
        for (int j = j0; j < j1; ++j) {
            if (j == r || (j >= r - 3 && j <= r + 3)) continue;     // the seeds were inserted already
            const float4 q = pts[j];
            const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
            knn_insert(dx * dx + dy * dy + dz * dz, best);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) sum += best[k] < 3.0e38f ? best[k] : 0.f;
    out[order[r]] = sum / 3.f;
}

}  // namespace gsr

using namespace gsr;

extern "C" {

int gsr_dist2_workspace_size(int32_t P, size_t *bytes)
{
    if (P < 0 || !bytes) { set_error("gsr_dist2_workspace_size: bad argument"); return GSR_ERR_INVALID_ARGUMENT; }
    const size_t Pn = P > 0 ? (size_t)P : 1, nb = (Pn + kBox - 1) / kBox;
    *bytes = 4 * align_up(Pn * 4) + align_up(Pn * 16) + align_up(nb * 24) + align_up(64) + radix_temp_bytes();
    return GSR_OK;
}

int gsr_dist2_knn3(int32_t P, const float *xyz, float *mean_dist2, void *workspace, void *stream)
{
    if (P < 0 || (P > 0 && (!xyz || !mean_dist2 || !workspace))) { set_error("gsr_dist2_knn3: bad argument"); return GSR_ERR_INVALID_ARGUMENT; }
    if (P == 0) return GSR_OK;
    hipStream_t s = (hipStream_t)stream;
    char *b = (char *)workspace;
    const size_t Pn = (size_t)P, nb = (Pn + kBox - 1) / kBox;
    uint32_t *keys[2], *vals[2];
    for (int i = 0; i < 2; ++i) { keys[i] = (uint32_t *)b; b += align_up(Pn * 4); }
    for (int i = 0; i < 2; ++i) { vals[i] = (uint32_t *)b; b += align_up(Pn * 4); }
    float4 *pts = (float4 *)b; b += align_up(Pn * 16);
    float *boxes = (float *)b; b += align_up(nb * 24);
    int *mm = (int *)b; b += align_up(64);
    void *radix_temp = b;
    ProfileScope prof("dist2_knn3", s);
    hipLaunchKernelGGL(k_knn_bounds_init, dim3(1), dim3(kKnnBlock), 0, s, mm);
    int grid = (P + kKnnBlock - 1) / kKnnBlock;
    hipLaunchKernelGGL(k_knn_bounds, dim3(grid > 1024 ? 1024 : grid), dim3(kKnnBlock), 0, s, P, xyz, mm);
    hipLaunchKernelGGL(k_knn_morton, dim3(grid), dim3(kKnnBlock), 0, s, P, xyz, mm, keys[0], vals[0]);
    GSR_LAUNCH_CHECK("knn_morton", false, s);
    int result = 0, rc;
    if ((rc = launch_radix_sort<uint32_t>(keys, vals, nullptr, (uint32_t)P, (uint64_t)P, nullptr, 0, 30, radix_temp, &result,
                                          "knn_sort", false, s)))
        return rc;
    hipLaunchKernelGGL(k_knn_gather_boxes, dim3((unsigned)nb), dim3(kKnnBlock), 0, s, P, xyz, vals[result], pts, boxes);
    hipLaunchKernelGGL(k_knn_search, dim3(grid), dim3(kKnnBlock), 0, s, P, (int)nb, pts, vals[result], boxes, mean_dist2);
    GSR_LAUNCH_CHECK("knn_search", false, s);
    return GSR_OK;
}

}  // extern "C"
