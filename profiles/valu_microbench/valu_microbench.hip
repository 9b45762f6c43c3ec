// valu_microbench.hip — VALU issue rates on gfx950 for the instruction classes of the blend kernels
// (csrc/gsr_render.hip): wave-instructions per cycle per SIMD at 1, 2, 4 and 8 resident waves per SIMD.
//
// Why: rocprofv3 ships no gfx950 section in its derived-counter tables, so "VALUBusy" falls back to the gfx94x formula
// (SQ_ACTIVE_INST_VALU x 4 / ...), which assumes a 4-cycle wave64 issue.  bench.py's `roofline.valu` prices a kernel's
// instruction mix with the rates measured HERE (profiles/valu_microbench/r02_valu_rates.json) instead.
//
// Method: every wave runs `iters` x 64 independent instructions of one class (8 accumulators round-robin, inline asm so
// the compiler cannot fuse, reorder or drop them) between two pairs of stamps: s_memtime (shader-clock ticks) and
// s_memrealtime (constant 100 MHz), so cycles AND the clock the chip actually held are known per wave.  A block is 256
// threads = 4 waves = one wave per SIMD; W blocks per CU give W waves per SIMD (grid = CUs x W; registers/LDS never limit
// residency here).  rate = W x instructions / median over waves of the stamped cycles.  The chip is kept busy for ~1 s
// before the first measurement and every measured launch runs for milliseconds (DVFS settles).
//
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 -o valu_microbench valu_microbench.hip && ./valu_microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define X8(s) s s s s s s s s
// one group = the instruction applied to accumulators %0..%7; I(n) expands to the instruction text for accumulator n
#define G8(I) I("0") I("1") I("2") I("3") I("4") I("5") I("6") I("7")

#define I_FMA(n) "v_fma_f32 %" n ", %" n ", %8, %9\n"
#define I_FMAC(n) "v_fmac_f32 %" n ", %8, %9\n"
#define I_MUL(n) "v_mul_f32 %" n ", %" n ", %8\n"
#define I_ADD(n) "v_add_f32 %" n ", %" n ", %8\n"
#define I_SUB(n) "v_sub_f32 %" n ", %8, %" n "\n"
#define I_MAX(n) "v_max_f32 %" n ", %" n ", %8\n"
#define I_MIN(n) "v_min_f32 %" n ", %" n ", %8\n"
#define I_MOV(n) "v_mov_b32 %" n ", %8\n"
#define I_MED3(n) "v_med3_f32 %" n ", %" n ", %8, %9\n"
#define I_EXP(n) "v_exp_f32 %" n ", %" n "\n"
#define I_RCP(n) "v_rcp_f32 %" n ", %" n "\n"
#define I_DPP(n) "v_add_f32_dpp %" n ", %" n ", %" n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_CND_VCC(n) "v_cndmask_b32 %" n ", %" n ", %8, vcc\n"
#define I_CND_SGPR(n) "v_cndmask_b32_e64 %" n ", %" n ", %8, %10\n"
#define I_CMP_VCC(n) "v_cmp_gt_f32 vcc, %" n ", %8\n"
#define I_CMP_SGPR(n) "v_cmp_gt_f32_e64 %8, %" n ", %9\n"
#define I_CMP_CND(n) "v_cmp_gt_f32 vcc, %" n ", %8\n v_cndmask_b32 %" n ", %" n ", %9, vcc\n"
#define I_PKFMA(n) "v_pk_fma_f32 %" n ", %" n ", %8, %9\n"
#define I_PKMUL(n) "v_pk_mul_f32 %" n ", %" n ", %8\n"
#define I_PKADD(n) "v_pk_add_f32 %" n ", %" n ", %8\n"

enum Op { FMA, FMAC, MUL, ADD, SUB, MAX_, MIN_, MOV, MED3, EXP, RCP, DPP, CND_VCC, CND_SGPR, CMP_VCC, CMP_SGPR, CMP_CND, PKFMA, PKMUL, PKADD,
          LDS_B128, N_OPS };
static const char *kNames[N_OPS] = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_max_f32", "v_min_f32",
                                    "v_mov_b32", "v_med3_f32", "v_exp_f32", "v_rcp_f32", "v_add_f32_dpp", "v_cndmask_b32(vcc)",
                                    "v_cndmask_b32(sgpr)", "v_cmp_gt_f32(vcc)", "v_cmp_gt_f32(sgpr)", "v_cmp+v_cndmask(pair)",
                                    "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "ds_read_b128(broadcast)"};
static const int kInstPerGroupItem[N_OPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1};

template <int OP>
__global__ __launch_bounds__(256) void k_rate(float *out, unsigned long long *stamps, int iters)
{
    __shared__ float4 lds[64];
    if (threadIdx.x < 64) lds[threadIdx.x] = make_float4(threadIdx.x, 1.f, 2.f, 3.f);
    __syncthreads();
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    const float b = 0.999f, c = 1e-4f;
    const v2f b2 = {0.999f, 1.001f}, c2 = {1e-4f, -1e-4f};
    unsigned long long sm = 0x5555555555555555ull;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#define ACC8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define PACC8 "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
#define RUN(I) asm volatile(X8(G8(I)) : ACC8 : "v"(b), "v"(c), "s"(sm) : "vcc")
#define RUNW(I) asm volatile(X8(G8(I)) : ACC8, "+s"(sm) : "v"(b), "v"(c) : "vcc")
#define PRUN(I) asm volatile(X8(G8(I)) : PACC8 : "v"(b2), "v"(c2))
        if constexpr (OP == FMA) RUN(I_FMA);
        else if constexpr (OP == FMAC) RUN(I_FMAC);
        else if constexpr (OP == MUL) RUN(I_MUL);
        else if constexpr (OP == ADD) RUN(I_ADD);
        else if constexpr (OP == SUB) RUN(I_SUB);
        else if constexpr (OP == MAX_) RUN(I_MAX);
        else if constexpr (OP == MIN_) RUN(I_MIN);
        else if constexpr (OP == MOV) RUN(I_MOV);
        else if constexpr (OP == MED3) RUN(I_MED3);
        else if constexpr (OP == EXP) RUN(I_EXP);
        else if constexpr (OP == RCP) RUN(I_RCP);
        else if constexpr (OP == DPP) RUN(I_DPP);
        else if constexpr (OP == CND_VCC) RUN(I_CND_VCC);
        else if constexpr (OP == CND_SGPR) RUN(I_CND_SGPR);
        else if constexpr (OP == CMP_VCC) RUN(I_CMP_VCC);
        else if constexpr (OP == CMP_CND) RUN(I_CMP_CND);
        else if constexpr (OP == PKFMA) PRUN(I_PKFMA);
        else if constexpr (OP == PKMUL) PRUN(I_PKMUL);
        else if constexpr (OP == PKADD) PRUN(I_PKADD);
        else if constexpr (OP == CMP_SGPR)
            asm volatile(X8(G8(I_CMP_SGPR)) : ACC8, "+s"(sm) : "v"(b), "v"(c) : "vcc");
        else if constexpr (OP == LDS_B128) {
            // 64 wave-uniform 16-byte reads (all lanes the same address: the blend kernels' record broadcast), 8 in flight
            typedef float v4f __attribute__((ext_vector_type(4)));
            const unsigned addr = (unsigned)(size_t)(&lds[i & 31]) & 0xFFFFu;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v4f q0, q1, q2, q3, q4, q5, q6, q7;
                asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                             "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n"
                             "ds_read_b128 %7, %8 offset:112\n s_waitcnt lgkmcnt(0)\n"
                             : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(addr));
                a0 += q0[0]; a1 += q1[1]; a2 += q2[2]; a3 += q3[3]; a4 += q4[0]; a5 += q5[1]; a6 += q6[2]; a7 += q7[3];
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    const float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1] + (float)(sm & 1);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

struct Result { double per_tick, ticks_per_realtick, wall_G; };

template <int OP>
static Result run(int W, int iters, float *out, unsigned long long *st_dev, int n_cu)
{
    const int blocks = n_cu * W;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, st_dev, iters);                    // warm-up, same length
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, st_dev, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> s((size_t)blocks * 8);
    CHECK(hipMemcpy(s.data(), st_dev, s.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ticks, ratio;
    for (size_t w = 0; w < (size_t)blocks * 4; ++w) { ticks.push_back((double)s[2 * w]); ratio.push_back((double)s[2 * w] / (double)s[2 * w + 1]); }
    std::sort(ticks.begin(), ticks.end()); std::sort(ratio.begin(), ratio.end());
    const double instr = 64.0 * iters * kInstPerGroupItem[OP];
    Result r;
    r.per_tick = (double)W * instr / ticks[ticks.size() / 2];
    r.ticks_per_realtick = ratio[ratio.size() / 2];
    r.wall_G = (double)blocks * 4 * instr / ((double)n_cu * 4) / (ms * 1e-3) * 1e-9;
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return r;
}

int main(int argc, char **argv)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float *out; unsigned long long *st;
    CHECK(hipMalloc(&out, (size_t)n_cu * 8 * 256 * 4));
    CHECK(hipMalloc(&st, (size_t)n_cu * 8 * 4 * 16));
    for (int i = 0; i < 12; ++i) run<FMA>(8, iters, out, st, n_cu);          // ~1 s of load before the first measurement
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"simds\": %d, \"clock_khz_max\": %d, \"instructions_per_wave\": %d,\n"
           " \"method\": \"64 independent wave64 instructions per loop trip, 8 accumulators, inline asm; cycles = s_memtime ticks, clock = "
           "ticks per s_memrealtime tick x 100 MHz; median over all waves\",\n \"rates\": {\n",
           prop.gcnArchName, n_cu, n_cu * 4, prop.clockRate, 64 * iters);
    const int Ws[4] = {1, 2, 4, 8};
    for (int op = 0; op < N_OPS; ++op) {
        printf("  \"%s\": {", kNames[op]);
        for (int wi = 0; wi < 4; ++wi) {
            const int W = Ws[wi];
            Result r{};
            switch (op) {
#define CASE(O) case O: r = run<O>(W, iters, out, st, n_cu); break;
                CASE(FMA) CASE(FMAC) CASE(MUL) CASE(ADD) CASE(SUB) CASE(MAX_) CASE(MIN_) CASE(MOV) CASE(MED3) CASE(EXP) CASE(RCP) CASE(DPP)
                CASE(CND_VCC) CASE(CND_SGPR) CASE(CMP_VCC) CASE(CMP_SGPR) CASE(CMP_CND) CASE(PKFMA) CASE(PKMUL) CASE(PKADD) CASE(LDS_B128)
            }
            printf("%s\"w%d\": {\"per_cycle_per_simd\": %.4f, \"clock_mhz\": %.0f, \"wall_G_per_s_per_simd\": %.4f}", wi ? ", " : "", W,
                   r.per_tick, r.ticks_per_realtick * 100.0, r.wall_G);
            fflush(stdout);
        }
        printf("}%s\n", op + 1 < N_OPS ? "," : "");
    }
    printf(" }\n}\n");
    return 0;
}
