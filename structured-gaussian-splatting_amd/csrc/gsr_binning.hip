// gsr_binning.hip — tile binning: prefix sum of tiles touched (K2), duplicate-with-keys (K3),
// stable radix tile-sort restricted to the live key bits (K4), tile ranges (K5).  Spec: SURVEY A.7.
//
// Key = (tile_id << 32) | binary32 bits of the view-space depth; payload = instance slot (the index of
// the instance in duplicate order = Gaussian-major).  Sorting slots instead of Gaussian ids lets the
// backward write one private gradient row per instance and reduce them per Gaussian in a fixed
// order — no float atomics, bitwise-reproducible gradients (MI355X float atomics run at ~1.3 TB/s
// when well shaped and ~17x slower when scattered one dword per row: MI355X_MICROARCH "Global float
// atomics").  Ties in (tile, depth) resolve by slot = ascending Gaussian index, as A.7 requires.
#include "gsr_internal.h"

#include <cstring>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace gsr {

size_t scan_temp_bytes(int P)
{
    size_t bytes = 0;
    if (P > 0)
        (void)rocprim::inclusive_scan(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)P,
                                      rocprim::plus<uint32_t>());
    return bytes;
}

size_t sort_temp_bytes(int64_t R)
{
    size_t bytes = 0;
    if (R > 0) {
        rocprim::double_buffer<uint64_t> k(nullptr, nullptr);
        rocprim::double_buffer<uint32_t> v(nullptr, nullptr);
        (void)rocprim::radix_sort_pairs(nullptr, bytes, k, v, (size_t)R, 0, 64);
    }
    return bytes;
}

GeomWS carve_geom(void *base, int P)
{
    GeomWS w;
    size_t o = 0;
    char *b = (char *)base;
    const size_t Pn = (size_t)(P > 0 ? P : 1);
    w.records = (float4 *)(b + o); o += align_up(Pn * 48);
    w.tiles_touched = (uint32_t *)(b + o); o += align_up(Pn * 4);
    w.offsets = (uint32_t *)(b + o); o += align_up(Pn * 4);
    w.clamped = (uint8_t *)(b + o); o += align_up(Pn);
    w.scan_temp_bytes = scan_temp_bytes(P);
    w.scan_temp = b + o; o += align_up(w.scan_temp_bytes + 16);
    w.total = o;
    return w;
}

ImageWS carve_image(void *base, const FrameK &f)
{
    ImageWS w;
    size_t o = 0;
    char *b = (char *)base;
    const size_t N = (size_t)f.W * f.H, Tn = (size_t)f.Gx * f.Gy;
    w.final_T = (float *)(b + o); o += align_up((N ? N : 1) * 4);
    w.n_contrib = (int32_t *)(b + o); o += align_up((N ? N : 1) * 4);
    w.ranges = (uint2 *)(b + o); o += align_up((Tn ? Tn : 1) * 8);
    w.total = o;
    return w;
}

BinningWS carve_binning(void *base, int64_t R)
{
    BinningWS w;
    size_t o = 0;
    char *b = (char *)base;
    const size_t Rn = (size_t)(R > 0 ? R : 1);
    for (int i = 0; i < 2; ++i) { w.keys[i] = (uint64_t *)(b + o); o += align_up(Rn * 8); }
    for (int i = 0; i < 2; ++i) { w.vals[i] = (uint32_t *)(b + o); o += align_up(Rn * 4); }
    w.inst_gid = (uint32_t *)(b + o); o += align_up(Rn * 4);
    w.sorted_gid = (uint32_t *)(b + o); o += align_up(Rn * 4);
    w.sorted_slot = (uint32_t *)(b + o); o += align_up(Rn * 4);
    w.grad_rows = (float *)(b + o); o += align_up(Rn * kRowFloats * 4);
    w.sort_temp_bytes = sort_temp_bytes(R);
    w.sort_temp = b + o; o += align_up(w.sort_temp_bytes + 16);
    w.total = o;
    return w;
}

int launch_scan(GeomWS &ws, int P, bool debug, hipStream_t s)
{
    if (P == 0) return GSR_OK;
    size_t bytes = ws.scan_temp_bytes;
    ProfileScope prof("scan", s);
    GSR_HIP_CHECK(rocprim::inclusive_scan(ws.scan_temp, bytes, ws.tiles_touched, ws.offsets, (size_t)P,
                                          rocprim::plus<uint32_t>(), s));
    GSR_LAUNCH_CHECK("scan", debug, s);
    return GSR_OK;
}

constexpr int kBinBlock = 256;

// ---- K3: one thread per Gaussian walks its (slab-clipped) tile rectangle in row-major order.
__global__ __launch_bounds__(kBinBlock) void k_duplicate(FrameK f, const float4 *__restrict__ records,
                                                         const uint32_t *__restrict__ tiles, const uint32_t *__restrict__ offsets,
                                                         uint64_t *__restrict__ keys,
                                                         uint32_t *__restrict__ vals, uint32_t *__restrict__ inst_gid)
{
    const int i = blockIdx.x * kBinBlock + threadIdx.x;
    if (i >= f.P) return;
    const uint32_t cnt = tiles[i];
    if (cnt == 0) return;
    const float4 r0 = records[3 * (size_t)i];
    const float4 r2 = records[3 * (size_t)i + 2];
    const float depth = r2.y;
    TileRect r = tile_rect(r0.x, r0.y, r2.z, f);
    slab_clip(r, f);
    uint32_t off = offsets[i] - cnt;
    const uint64_t dbits = (uint64_t)__float_as_uint(depth);
    for (int y = r.y0; y < r.y1; ++y)
        for (int x = r.x0; x < r.x1; ++x) {
            keys[off] = ((uint64_t)(uint32_t)(y * f.Gx + x) << 32) | dbits;
            vals[off] = off;
            inst_gid[off] = (uint32_t)i;
            ++off;
        }
}

int launch_duplicate(const FrameK &f, const GeomWS &gw, BinningWS &bw, int64_t R, bool debug, hipStream_t s)
{
    if (f.P == 0 || R == 0) return GSR_OK;
    ProfileScope prof("duplicate", s);
    hipLaunchKernelGGL(k_duplicate, dim3((f.P + kBinBlock - 1) / kBinBlock), dim3(kBinBlock), 0, s, f, gw.records,
                       gw.tiles_touched, gw.offsets, bw.keys[0], bw.vals[0], bw.inst_gid);
    GSR_LAUNCH_CHECK("duplicate", debug, s);
    return GSR_OK;
}

static int msb_plus1(uint32_t n)
{
    int b = 0;
    while (n) { ++b; n >>= 1; }
    return b;
}

// ---- K4: stable LSD radix sort over the live bits only: 32 depth bits + ceil(log2(Tn)) tile bits.
int launch_sort(const FrameK &f, BinningWS &bw, int64_t R, int *result_buffer, bool debug, hipStream_t s)
{
    *result_buffer = 0;
    if (R == 0) return GSR_OK;
    const int end_bit = 32 + msb_plus1((uint32_t)(f.Gx * f.Gy));
    rocprim::double_buffer<uint64_t> k(bw.keys[0], bw.keys[1]);
    rocprim::double_buffer<uint32_t> v(bw.vals[0], bw.vals[1]);
    size_t bytes = bw.sort_temp_bytes;
    ProfileScope prof("radix_sort", s);
    GSR_HIP_CHECK(rocprim::radix_sort_pairs(bw.sort_temp, bytes, k, v, (size_t)R, 0, (unsigned)end_bit, s));
    *result_buffer = (k.current() == bw.keys[0]) ? 0 : 1;
    GSR_LAUNCH_CHECK("radix_sort", debug, s);
    return GSR_OK;
}

// ---- K5: tile boundaries in the sorted keys -> ranges; also materialises sorted position -> Gaussian.
__global__ __launch_bounds__(kBinBlock) void k_ranges(int64_t R, const uint64_t *__restrict__ keys,
                                                      const uint32_t *__restrict__ slots, const uint32_t *__restrict__ inst_gid,
                                                      uint2 *__restrict__ ranges, uint32_t *__restrict__ sorted_gid,
                                                      uint32_t *__restrict__ sorted_slot)
{
    const int64_t i = (int64_t)blockIdx.x * kBinBlock + threadIdx.x;
    if (i >= R) return;
    const uint32_t tile = (uint32_t)(keys[i] >> 32);
    if (i == 0) ranges[tile].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
        if (prev != tile) { ranges[prev].y = (uint32_t)i; ranges[tile].x = (uint32_t)i; }
    }
    if (i == R - 1) ranges[tile].y = (uint32_t)R;
    const uint32_t slot = slots[i];
    sorted_slot[i] = slot;
    sorted_gid[i] = inst_gid[slot];
}

int launch_ranges(const FrameK &f, BinningWS &bw, int result_buffer, ImageWS &iw, int64_t R, bool debug, hipStream_t s)
{
    const size_t Tn = (size_t)f.Gx * f.Gy;
    GSR_HIP_CHECK(hipMemsetAsync(iw.ranges, 0, Tn * sizeof(uint2), s));
    if (R == 0) return GSR_OK;
    ProfileScope prof("ranges", s);
    hipLaunchKernelGGL(k_ranges, dim3((unsigned)((R + kBinBlock - 1) / kBinBlock)), dim3(kBinBlock), 0, s, R,
                       bw.keys[result_buffer], bw.vals[result_buffer], bw.inst_gid, iw.ranges, bw.sorted_gid,
                       bw.sorted_slot);
    GSR_LAUNCH_CHECK("ranges", debug, s);
    return GSR_OK;
}

}  // namespace gsr
