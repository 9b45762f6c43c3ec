// gsr_sort.hip — hand-written device primitives for the binning stage:
//   * inclusive prefix sum of u32 (three small kernels, host-known length)
//   * stable LSD radix sort of (key, u32 value) pairs, 8-bit digits, FIXED grid, element count read from
//     DEVICE memory (the progressive binning never tells the host how many instances a chunk emitted).
//
// Sort structure per pass (three launches): per-block digit histogram -> exclusive scan of the
// [256][B] table -> stable scatter.  Block b owns the contiguous key range [b*per, (b+1)*per), walked in
// sub-tiles of 1024 keys; inside a sub-tile wave w owns keys [256w, 256w+256) in 4 rounds of 64
// consecutive keys, so (sub-tile, wave, round, lane) order == key order and the sort is stable.
// Ranking inside a round is a wave64 match-any built from 8 ballots (one per digit bit) + popcount;
// no LDS atomics are needed in the scatter, and the histogram uses one LDS atomic per wave when a
// wave's 64 keys share a digit (the common case for the high tile-id digit) instead of 64.
#include "gsr_internal.h"

namespace gsr {

// ------------------------------------------------------------------------------------ prefix sum
constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanBlock * kScanItems;      // 2048 elements per block
static_assert(kScanTile == kScanTileElems, "mass_blocks granularity");

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// block-wide inclusive scan of one value per thread (256 threads); returns inclusive value, total in *total
__device__ __forceinline__ uint32_t block_incl_scan(uint32_t v, uint32_t *sh_wave /*[4]*/, uint32_t *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v, lane);
    if (lane == 63) sh_wave[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kScanBlock / kWave; ++i) {
        const uint32_t s = sh_wave[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return inc + base;
}

// 64-bit flavour for the auxiliary (optical mass) block sums
__device__ __forceinline__ unsigned long long block_incl_scan64(unsigned long long v, unsigned long long *sh_wave /*[4]*/,
                                                               unsigned long long *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const unsigned long long t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) sh_wave[w] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kScanBlock / kWave; ++i) {
        const unsigned long long s = sh_wave[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return inc + base;
}

// gather != nullptr: the scanned sequence is in[gather[i]] (tiles touched, read in depth order)
// aux != nullptr: the input is (value, auxiliary) PAIRS and replaces `in`; the auxiliary quantity (optical mass) is summed
// per block alongside: aux_blocks[b] = sum of aux[gather[i]].y over the block's elements (k_scan_tops turns the block
// sums into their exclusive prefix)
__global__ __launch_bounds__(kScanBlock) void k_scan_local(const uint32_t *__restrict__ in, const uint32_t *__restrict__ gather,
                                                           uint32_t *__restrict__ out, uint32_t *__restrict__ block_sums, int n,
                                                           const uint2 *__restrict__ aux, unsigned long long *__restrict__ aux_blocks)
{
    __shared__ uint32_t sh_wave[kScanBlock / kWave];
    __shared__ unsigned long long sh_wave64[kScanBlock / kWave];
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems], sum = 0;
    unsigned long long asum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const bool ok = base + i < n;
        const uint32_t src = ok ? (gather ? gather[base + i] : (uint32_t)(base + i)) : 0u;
        if (aux) {                                             // (value, auxiliary) pairs: one 8-byte gather
            const uint2 pr = ok ? aux[src] : make_uint2(0u, 0u);
            v[i] = pr.x; asum += pr.y;
        } else {
            v[i] = ok ? in[src] : 0u;
        }
        sum += v[i];
    }
    uint32_t total;
    uint32_t run = block_incl_scan(sum, sh_wave, &total) - sum;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) { run += v[i]; if (base + i < n) out[base + i] = run; }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
    if (aux) {
        unsigned long long atotal;
        block_incl_scan64(asum, sh_wave64, &atotal);
        if (threadIdx.x == 0) aux_blocks[blockIdx.x] = atotal;
    }
}

// exclusive scan of the block sums, any count, one block
__global__ __launch_bounds__(kScanBlock) void k_scan_tops(uint32_t *__restrict__ block_sums, int nb, uint32_t *__restrict__ grand_total,
                                                          const uint32_t *__restrict__ acc_in, uint32_t *__restrict__ acc_out,
                                                          uint32_t *__restrict__ overflow, unsigned long long *__restrict__ aux_blocks)
{
    __shared__ uint32_t sh_wave[kScanBlock / kWave];
    __shared__ unsigned long long sh_wide[kScanBlock / kWave];
    if (aux_blocks) {                                          // block sums -> exclusive prefix, [nb] = grand total
        unsigned long long acarry = 0;
        for (int base = 0; base < nb; base += kScanBlock) {
            const int i = base + threadIdx.x;
            const unsigned long long v = i < nb ? aux_blocks[i] : 0ull;
            unsigned long long total;
            const unsigned long long inc = block_incl_scan64(v, sh_wide, &total);
            if (i < nb) aux_blocks[i] = acarry + inc - v;
            acarry += total;
        }
        if (threadIdx.x == 0) aux_blocks[nb] = acarry;
        __syncthreads();
    }
    uint32_t carry = 0;
    unsigned long long wide = 0;                               // the same sum in 64 bits, to detect wrap-around
    for (int base = 0; base < nb; base += kScanBlock) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nb ? block_sums[i] : 0u;
        wide += v;
        uint32_t total;
        const uint32_t inc = block_incl_scan(v, sh_wave, &total);
        if (i < nb) block_sums[i] = carry + inc - v;
        carry += total;
    }
    if (overflow) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) wide += __shfl_xor(wide, off);
        if ((threadIdx.x & 63) == 0) sh_wide[threadIdx.x >> 6] = wide;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (int i = 0; i < kScanBlock / kWave; ++i) t += sh_wide[i];
            *overflow = t > 0xFFFFFFFFull ? 1u : 0u;
        }
    }
    if (threadIdx.x == 0) {
        if (grand_total) *grand_total = carry;
        if (acc_out) *acc_out = (acc_in ? *acc_in : 0u) + carry;      // running base of the next chunk
    }
}

__global__ __launch_bounds__(kScanBlock) void k_scan_add(uint32_t *__restrict__ out, const uint32_t *__restrict__ block_sums, int n)
{
    const uint32_t add = block_sums[blockIdx.x];
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
        if (base + i < n) out[base + i] += add;
}

// short inputs (a chunk of a few thousand large splats): one block, one launch instead of three
constexpr int kScanSmallMax = 8 * kScanTile;
__global__ __launch_bounds__(kScanBlock) void k_scan_small(const uint32_t *__restrict__ in, const uint32_t *__restrict__ gather,
                                                           uint32_t *__restrict__ out, int n,
                                                           uint32_t *__restrict__ grand_total, const uint32_t *__restrict__ acc_in,
                                                           uint32_t *__restrict__ acc_out, uint32_t *__restrict__ overflow,
                                                           const uint2 *__restrict__ aux, unsigned long long *__restrict__ aux_blocks)
{
    __shared__ uint32_t sh_wave[kScanBlock / kWave];
    __shared__ unsigned long long sh_wave64[kScanBlock / kWave];
    uint32_t carry = 0;
    unsigned long long wide = 0, acarry = 0;
    for (int t0 = 0; t0 < n; t0 += kScanTile) {
        const int base = t0 + threadIdx.x * kScanItems;
        uint32_t v[kScanItems], sum = 0;
        unsigned long long asum = 0;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) {
            const bool ok = base + i < n;
            const uint32_t src = ok ? (gather ? gather[base + i] : (uint32_t)(base + i)) : 0u;
            if (aux) {
                const uint2 pr = ok ? aux[src] : make_uint2(0u, 0u);
                v[i] = pr.x; asum += pr.y;
            } else {
                v[i] = ok ? in[src] : 0u;
            }
            sum += v[i];
        }
        uint32_t total;
        uint32_t run = carry + block_incl_scan(sum, sh_wave, &total) - sum;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) { run += v[i]; if (base + i < n) out[base + i] = run; }
        wide += total;
        carry += total;
        if (aux) {                                             // exclusive prefix of the per-block sums, written as we go
            unsigned long long atotal;
            block_incl_scan64(asum, sh_wave64, &atotal);
            if (threadIdx.x == 0) aux_blocks[t0 / kScanTile] = acarry;
            acarry += atotal;
        }
    }
    if (aux && threadIdx.x == 0) aux_blocks[(n + kScanTile - 1) / kScanTile] = acarry;
    if (threadIdx.x == 0) {
        if (grand_total) *grand_total = carry;
        if (acc_out) *acc_out = (acc_in ? *acc_in : 0u) + carry;
        if (overflow) *overflow = wide > 0xFFFFFFFFull ? 1u : 0u;
    }
}

size_t scan_temp_bytes(int n) { return align_up((size_t)((n + kScanTile - 1) / kScanTile + 1) * 4); }

// out[i] = in[0] + ... + in[i]; optional *grand_total (device) = sum of all.
int launch_scan_inclusive(const uint32_t *in, uint32_t *out, int n, void *temp, uint32_t *grand_total, const uint32_t *acc_in,
                          uint32_t *acc_out, const char *name, bool debug, hipStream_t s, uint32_t *overflow, const uint32_t *gather,
                          const uint2 *aux, unsigned long long *aux_block_prefix)
{
    if (n <= 0) {
        if (overflow) GSR_HIP_CHECK(hipMemsetAsync(overflow, 0, 4, s));
        if (grand_total) GSR_HIP_CHECK(hipMemsetAsync(grand_total, 0, 4, s));
        if (acc_out && acc_in) GSR_HIP_CHECK(hipMemcpyAsync(acc_out, acc_in, 4, hipMemcpyDeviceToDevice, s));
        return GSR_OK;
    }
    ProfileScope prof(name, s);
    if (n <= kScanSmallMax) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(kScanBlock), 0, s, in, gather, out, n, grand_total, acc_in, acc_out, overflow,
                           aux, aux_block_prefix);
        GSR_LAUNCH_CHECK(name, debug, s);
        return GSR_OK;
    }
    const int nb = (n + kScanTile - 1) / kScanTile;
    uint32_t *sums = (uint32_t *)temp;
    hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(kScanBlock), 0, s, in, gather, out, sums, n, aux, aux_block_prefix);
    hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(kScanBlock), 0, s, sums, nb, grand_total, acc_in, acc_out, overflow,
                       aux ? aux_block_prefix : nullptr);
    if (nb > 1) hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(kScanBlock), 0, s, out, sums, n);
    GSR_LAUNCH_CHECK(name, debug, s);
    return GSR_OK;
}

// ------------------------------------------------------------------------------------ radix sort
constexpr int kRadixBlock = 256;
constexpr int kRadixBins = 256;
constexpr int kRadixRounds = 4;                                  // 64-key rounds per wave per sub-tile
constexpr int kRadixSubTile = kRadixBlock * kRadixRounds;        // 1024 keys

__device__ __forceinline__ void block_range(uint32_t n, int nblocks, uint32_t subtile, uint32_t &lo, uint32_t &hi)
{
    uint32_t per = (n + nblocks - 1) / nblocks;
    per = (per + subtile - 1) / subtile * subtile;
    const uint64_t l = (uint64_t)blockIdx.x * per, h = l + per;
    lo = l < n ? (uint32_t)l : n;
    hi = h < n ? (uint32_t)h : n;
}

template <typename K>
__global__ __launch_bounds__(kRadixBlock) void k_radix_hist(const K *__restrict__ keys, const uint32_t *__restrict__ n_ptr,
                                                            uint32_t n_host, const uint32_t *__restrict__ base_ptr, int shift, uint32_t dmask,
                                                            uint32_t subtile, uint32_t *__restrict__ table)
{
    __shared__ uint32_t hist[kRadixBins];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    if (base_ptr) keys += *base_ptr;
    uint32_t lo, hi;
    block_range(n, gridDim.x, subtile, lo, hi);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += kRadixBlock) {
        const uint32_t d = (uint32_t)(keys[i] >> shift) & dmask;
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
        const unsigned long long same = __ballot(d == d0);
        const unsigned long long act = __ballot(1);
        if (same == act) {                                         // whole wave on one digit: one atomic
            if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicAdd(&hist[d0], (uint32_t)__popcll(act));
        } else {
            atomicAdd(&hist[d], 1u);
        }
    }
    __syncthreads();
    table[threadIdx.x * gridDim.x + blockIdx.x] = hist[threadIdx.x];
}

// Per-digit exclusive scan over the blocks: block d owns row d of the digit-major [256][B] table
// (B <= 2048 = 256 threads x 8 entries) and writes the row total to totals[d].  The scatter kernel adds the
// exclusive prefix of the 256 totals itself, so a pass stays at three launches.
constexpr int kRadixMaxBlocks = 2048;
__global__ __launch_bounds__(kRadixBlock) void k_radix_scan(uint32_t *__restrict__ table, int B, uint32_t *__restrict__ totals)
{
    __shared__ uint32_t sh_wave[kRadixBlock / kWave];
    uint32_t *row = table + (size_t)blockIdx.x * B;
    const int per = (B + kRadixBlock - 1) / kRadixBlock;            // <= 8
    const int b0 = threadIdx.x * per;
    uint32_t v[8], sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = (i < per && b0 + i < B) ? row[b0 + i] : 0u; sum += v[i]; }
    uint32_t total;
    uint32_t run = block_incl_scan(sum, sh_wave, &total) - sum;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < per && b0 + i < B) { row[b0 + i] = run; run += v[i]; }
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

template <typename K>
__global__ __launch_bounds__(kRadixBlock) void k_radix_scatter(const K *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                               K *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                               const uint32_t *__restrict__ n_ptr, uint32_t n_host,
                                                               const uint32_t *__restrict__ base_ptr, int shift, uint32_t dmask,
                                                               const uint32_t *__restrict__ table,
                                                               const uint32_t *__restrict__ totals,
                                                               const uint32_t *__restrict__ vals2_in, uint32_t *__restrict__ vals2_out)
{
    if (base_ptr) {
        const uint32_t bo = *base_ptr;
        keys_in += bo; vals_in += bo; keys_out += bo; vals_out += bo;
        if (vals2_in) { vals2_in += bo; vals2_out += bo; }
    }
    __shared__ uint32_t base[kRadixBins];                          // next global position per digit for this block
    __shared__ volatile uint32_t wave_cnt[kRadixBlock / kWave][kRadixBins];
    __shared__ uint32_t offs[kRadixBlock / kWave][kRadixBins];
    __shared__ uint32_t sh_wave[kRadixBlock / kWave];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        const uint32_t tot = totals[threadIdx.x];
        uint32_t all;
        const uint32_t digit_excl = block_incl_scan(tot, sh_wave, &all) - tot;
        base[threadIdx.x] = digit_excl + table[threadIdx.x * gridDim.x + blockIdx.x];
    }
#pragma unroll
    for (int i = 0; i < kRadixBlock / kWave; ++i) wave_cnt[i][threadIdx.x] = 0;
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    uint32_t lo, hi;
    block_range(n, gridDim.x, kRadixSubTile, lo, hi);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    __syncthreads();
    for (uint32_t tile = lo; tile < hi; tile += kRadixSubTile) {
        K key[kRadixRounds];
        uint32_t local[kRadixRounds];
#pragma unroll
        for (int r = 0; r < kRadixRounds; ++r) {
            const uint32_t idx = tile + w * (kWave * kRadixRounds) + r * kWave + lane;
            const bool valid = idx < hi;
            key[r] = valid ? keys_in[idx] : (K)0;
            const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const unsigned long long m = __ballot((d >> bit) & 1u);
                peers &= ((d >> bit) & 1u) ? m : ~m;
            }
            const uint32_t old = wave_cnt[w][d];
            local[r] = old + (uint32_t)__popcll(peers & lt_mask);
            if (valid && (peers & lt_mask) == 0ull) wave_cnt[w][d] = old + (uint32_t)__popcll(peers);
        }
        __syncthreads();
        {                                                          // thread d: digit d's wave offsets
            uint32_t run = base[threadIdx.x];
#pragma unroll
            for (int i = 0; i < kRadixBlock / kWave; ++i) {
                const uint32_t c = wave_cnt[i][threadIdx.x];
                offs[i][threadIdx.x] = run;
                run += c;
                wave_cnt[i][threadIdx.x] = 0;
            }
            base[threadIdx.x] = run;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRadixRounds; ++r) {
            const uint32_t idx = tile + w * (kWave * kRadixRounds) + r * kWave + lane;
            if (idx < hi) {
                const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
                const uint32_t dst = offs[w][d] + local[r];
                keys_out[dst] = key[r];
                vals_out[dst] = vals_in[idx];
                if (vals2_in) vals2_out[dst] = vals2_in[idx];
            }
        }
        // the next sub-tile's first barrier orders these reads of offs[] before its rewrite
    }
}

// ---- large-n scatter: same stable ranking, but the sub-tile (4096 keys) is first written to LDS in sorted
// order and then streamed out, so that consecutive threads store consecutive keys of a digit's run (~16 keys =
// 64 B per run and sub-tile) instead of 64 unrelated dwords per wave-instruction.  The first tile-sort pass sees
// keys whose low byte walks through all 256 values (consecutive tiles of a rectangle): without the reorder
// every store instruction touches 64 different cache lines.
constexpr int kBigRounds = 16;
constexpr int kBigSubTile = kRadixBlock * kBigRounds;            // 4096 keys
template <typename K, bool V2>          // V2: a second payload array travels with vals (its staging LDS only then)
__global__ __launch_bounds__(kRadixBlock) void k_radix_scatter_big(const K *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                                   K *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                                   const uint32_t *__restrict__ n_ptr, uint32_t n_host,
                                                                   const uint32_t *__restrict__ base_ptr, int shift, uint32_t dmask,
                                                                   const uint32_t *__restrict__ table,
                                                                   const uint32_t *__restrict__ totals,
                                                                   const uint32_t *__restrict__ vals2_in, uint32_t *__restrict__ vals2_out)
{
    if (base_ptr) {
        const uint32_t bo = *base_ptr;
        keys_in += bo; vals_in += bo; keys_out += bo; vals_out += bo;
        if (vals2_in) { vals2_in += bo; vals2_out += bo; }
    }
    __shared__ uint32_t sv2[V2 ? kBigSubTile : 1];
    __shared__ uint32_t base[kRadixBins];                          // next global position per digit for this block
    __shared__ uint32_t delta[kRadixBins];                         // global position minus position in the sorted sub-tile
    __shared__ volatile uint32_t wave_cnt[kRadixBlock / kWave][kRadixBins];
    __shared__ uint32_t offs[kRadixBlock / kWave][kRadixBins];
    __shared__ uint32_t sh_wave[kRadixBlock / kWave];
    __shared__ K sk[kBigSubTile];
    __shared__ uint32_t sv[kBigSubTile];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        const uint32_t tot = totals[threadIdx.x];
        uint32_t all;
        const uint32_t digit_excl = block_incl_scan(tot, sh_wave, &all) - tot;
        base[threadIdx.x] = digit_excl + table[threadIdx.x * gridDim.x + blockIdx.x];
    }
#pragma unroll
    for (int i = 0; i < kRadixBlock / kWave; ++i) wave_cnt[i][threadIdx.x] = 0;
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    uint32_t lo, hi;
    block_range(n, gridDim.x, kBigSubTile, lo, hi);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    __syncthreads();
    for (uint32_t tile = lo; tile < hi; tile += kBigSubTile) {
        K key[kBigRounds];
        uint32_t local[kBigRounds];
#pragma unroll
        for (int r = 0; r < kBigRounds; ++r) {
            const uint32_t idx = tile + w * (kWave * kBigRounds) + r * kWave + lane;
            const bool valid = idx < hi;
            key[r] = valid ? keys_in[idx] : (K)0;
            const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const unsigned long long m = __ballot((d >> bit) & 1u);
                peers &= ((d >> bit) & 1u) ? m : ~m;
            }
            const uint32_t old = wave_cnt[w][d];
            local[r] = old + (uint32_t)__popcll(peers & lt_mask);
            if (valid && (peers & lt_mask) == 0ull) wave_cnt[w][d] = old + (uint32_t)__popcll(peers);
        }
        __syncthreads();
        {                                                          // thread d: digit d of this sub-tile
            uint32_t c[kRadixBlock / kWave], tot = 0;
#pragma unroll
            for (int i = 0; i < kRadixBlock / kWave; ++i) { c[i] = wave_cnt[i][threadIdx.x]; tot += c[i]; wave_cnt[i][threadIdx.x] = 0; }
            uint32_t all;
            uint32_t run = block_incl_scan(tot, sh_wave, &all) - tot;   // start of digit d inside the sorted sub-tile
            delta[threadIdx.x] = base[threadIdx.x] - run;
            base[threadIdx.x] += tot;
#pragma unroll
            for (int i = 0; i < kRadixBlock / kWave; ++i) { offs[i][threadIdx.x] = run; run += c[i]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kBigRounds; ++r) {
            const uint32_t idx = tile + w * (kWave * kBigRounds) + r * kWave + lane;
            if (idx < hi) {
                const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
                const uint32_t p = offs[w][d] + local[r];
                sk[p] = key[r];
                sv[p] = vals_in[idx];
                if constexpr (V2) sv2[p] = vals2_in[idx];
            }
        }
        __syncthreads();
        const uint32_t count = min((uint32_t)kBigSubTile, hi - tile);
        for (uint32_t p = threadIdx.x; p < count; p += kRadixBlock) {
            const K kk = sk[p];
            const uint32_t dst = delta[(uint32_t)(kk >> shift) & dmask] + p;
            keys_out[dst] = kk;
            vals_out[dst] = sv[p];
            if constexpr (V2) vals2_out[dst] = sv2[p];
        }
        __syncthreads();
    }
}

// keys / vals of buffer 1 back into buffer 0 (a caller that needs its result in place, after an odd number of passes): one
// streaming launch instead of a fourth radix pass (three launches)
template <typename K>
__global__ __launch_bounds__(kRadixBlock) void k_radix_copy_back(const K *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                                 K *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                                 const uint32_t *__restrict__ n_ptr, uint32_t n_host,
                                                                 const uint32_t *__restrict__ base_ptr)
{
    const uint32_t n = n_ptr ? *n_ptr : n_host;
    const uint32_t bo = base_ptr ? *base_ptr : 0u;
    for (uint32_t i = blockIdx.x * kRadixBlock + threadIdx.x; i < n; i += gridDim.x * kRadixBlock) {
        keys_out[bo + i] = keys_in[bo + i];
        vals_out[bo + i] = vals_in[bo + i];
    }
}

int radix_blocks(uint64_t n_max)
{
    uint64_t b = (n_max + 2047) / 2048;          // ~2 sub-tiles per block: enough blocks to fill 256 CUs early (1024 and 4096
                                                 // keys per block measured 3-15 % slower at 1-2 M keys: tools/sort_bench.py)
    if (b < 64) b = 64;
    if (b > kRadixMaxBlocks) b = kRadixMaxBlocks;
    return (int)b;
}

size_t radix_temp_bytes() { return align_up((size_t)kRadixBins * kRadixMaxBlocks * 4) + align_up(kRadixBins * 4); }

// Sorts n (device *n_ptr if non-null, else n_host) pairs on key bits [begin_bit, end_bit).  buffers[0] holds
// the input; *result = index of the buffer holding the output.  n_max bounds n on the host (grid sizing).
template <typename K>
int launch_radix_sort(K *const keys[2], uint32_t *const vals[2], const uint32_t *n_ptr, uint32_t n_host, uint64_t n_max,
                      const uint32_t *base_ptr, int begin_bit, int end_bit, void *temp, int *result, const char *name,
                      bool debug, hipStream_t s, bool even_passes, uint32_t *const *vals2)
{
    *result = 0;
    if (n_max == 0 || end_bit <= begin_bit) return GSR_OK;
    ProfileScope prof(name, s);
    const bool big = n_max >= (4ull << 20);          // large sorts: LDS-reordering scatter on 4096-key sub-tiles
    int B = radix_blocks(n_max);
    if (big) { uint64_t b = (n_max + 2 * kBigSubTile - 1) / (2 * kBigSubTile); B = (int)(b > kRadixMaxBlocks ? kRadixMaxBlocks : b); }
    const uint32_t subtile = big ? kBigSubTile : kRadixSubTile;
    uint32_t *table = (uint32_t *)temp;
    uint32_t *totals = (uint32_t *)((char *)temp + align_up((size_t)kRadixBins * kRadixMaxBlocks * 4));
    int cur = 0;
    // digits: as few passes as 8-bit digits need, the key bits spread evenly over them (13 tile-id bits sort as
    // 7 + 6, not 8 + 5: fewer bins in the first pass means longer contiguous runs in its scatter)
    const int total_bits = end_bit - begin_bit;
    const int passes = (total_bits + 7) / 8;
    const bool copy_back = even_passes && (passes & 1);          // the caller wants the result back in buffer 0
    int shift = begin_bit;
    for (int p = 0; p < passes; ++p) {
        const int left = end_bit - shift, bits = (left + (passes - p) - 1) / (passes - p);
        const uint32_t dmask = (1u << bits) - 1u;
        hipLaunchKernelGGL(k_radix_hist<K>, dim3(B), dim3(kRadixBlock), 0, s, keys[cur], n_ptr, n_host, base_ptr, shift, dmask, subtile,
                           table);
        hipLaunchKernelGGL(k_radix_scan, dim3(kRadixBins), dim3(kRadixBlock), 0, s, table, B, totals);
        if (big && vals2)
            hipLaunchKernelGGL((k_radix_scatter_big<K, true>), dim3(B), dim3(kRadixBlock), 0, s, keys[cur], vals[cur], keys[cur ^ 1],
                               vals[cur ^ 1], n_ptr, n_host, base_ptr, shift, dmask, table, totals, vals2[cur], vals2[cur ^ 1]);
        else if (big)
            hipLaunchKernelGGL((k_radix_scatter_big<K, false>), dim3(B), dim3(kRadixBlock), 0, s, keys[cur], vals[cur], keys[cur ^ 1],
                               vals[cur ^ 1], n_ptr, n_host, base_ptr, shift, dmask, table, totals, nullptr, nullptr);
        else
            hipLaunchKernelGGL(k_radix_scatter<K>, dim3(B), dim3(kRadixBlock), 0, s, keys[cur], vals[cur], keys[cur ^ 1],
                               vals[cur ^ 1], n_ptr, n_host, base_ptr, shift, dmask, table, totals, vals2 ? vals2[cur] : nullptr,
                               vals2 ? vals2[cur ^ 1] : nullptr);
        cur ^= 1;
        shift += bits;
    }
    if (copy_back) {
        uint64_t blocks = (n_max + 4 * kRadixBlock - 1) / (4 * kRadixBlock);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_radix_copy_back<K>, dim3((unsigned)(blocks ? blocks : 1)), dim3(kRadixBlock), 0, s, keys[cur], vals[cur], keys[cur ^ 1],
                           vals[cur ^ 1], n_ptr, n_host, base_ptr);
        cur ^= 1;
    }
    *result = cur;
    GSR_LAUNCH_CHECK(name, debug, s);
    return GSR_OK;
}

template int launch_radix_sort<uint32_t>(uint32_t *const[2], uint32_t *const[2], const uint32_t *, uint32_t, uint64_t,
                                         const uint32_t *, int, int, void *, int *, const char *, bool, hipStream_t, bool, uint32_t *const *);
template int launch_radix_sort<uint64_t>(uint64_t *const[2], uint32_t *const[2], const uint32_t *, uint32_t, uint64_t,
                                         const uint32_t *, int, int, void *, int *, const char *, bool, hipStream_t, bool, uint32_t *const *);

}  // namespace gsr
