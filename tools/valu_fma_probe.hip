// VALU issue probe for one question: how many SIMD cycles does a wave64 v_fma_f32 cost on gfx950 (MI355X)?
// MI355X_MICROARCH.md lists 2 cycles ("SIMD-32"); tools/valu_peak.hip measured 4.2 in round 2.  This probe pins every register by
// number so that the source operands' VGPR banks (register index mod 4) are known, and runs the same 64-instruction block
//   * `free`:     v_fma_f32 vK, vK, v17, v18   K = 8, 12, ..., 36 (bank 0), the two constants in banks 1 and 2: no two sources share a bank
//   * `conflict`: v_fma_f32 vK, vK, v40, v44   all three sources in bank 0
//   * `pk`:       v_pk_fma_f32 v[K:K+1], v[K:K+1], v[48:49], v[50:51]   (two FMAs per lane and instruction)
//   * `mul`:      v_mul_f32 vK, vK, v17        (the plain-f32 class the r02 table has at 2.3 cycles)
// at 1, 2, 4 and 8 waves per SIMD (workgroups of 256 lanes, LDS-sized so that W fit a CU), 8 independent chains per wave.
// Output: one JSON line per (variant, W): cycles per wave instruction per SIMD from the HIP event time over the whole chip at the
// measured shader clock (s_memtime / s_memrealtime), and TFLOP/s.   tools/valu_fma_probe.sh adds the SQ counters and the ISA.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define REP8(X) X(8) X(12) X(16) X(20) X(24) X(28) X(32) X(36)
#define FMA_FREE(K) "v_fma_f32 v" #K ", v" #K ", v17, v18\n"
#define FMA_CONF(K) "v_fma_f32 v" #K ", v" #K ", v40, v44\n"
#define MUL_F32(K) "v_mul_f32 v" #K ", v" #K ", v17\n"
#define MUL_CONF(K) "v_mul_f32 v" #K ", v" #K ", v40\n"
#define MUL_IND_FREE(K) "v_mul_f32 v" #K ", v17, v18\n"
#define MUL_IND_CONF(K) "v_mul_f32 v" #K ", v40, v44\n"
#define SUB_CHAIN(K) "v_sub_f32 v" #K ", v17, v" #K "\n"
#define CND_FREE(K) "v_cndmask_b32 v" #K ", v" #K ", v17, vcc\n"
#define CND_CONF(K) "v_cndmask_b32 v" #K ", v" #K ", v40, vcc\n"
#define MIN_FREE(K) "v_min_f32 v" #K ", v" #K ", v17\n"
#define MAD24_FREE(K) "v_mad_i32_i24 v" #K ", v" #K ", v17, v18\n"
#define MAD24_CONF(K) "v_mad_i32_i24 v" #K ", v" #K ", v40, v44\n"
#define FMA_LIT(K) "v_fma_f32 v" #K ", v" #K ", v17, 1.0\n"
#define PK2(K, K1) "v_pk_fma_f32 v[" #K ":" #K1 "], v[" #K ":" #K1 "], v[48:49], v[50:51]\n"
#define PK_ALL PK2(8, 9) PK2(12, 13) PK2(16, 17) PK2(20, 21) PK2(24, 25) PK2(28, 29) PK2(32, 33) PK2(36, 37)
#define BLOCK64(B) B B B B B B B B
#define CLOB "v8", "v9", "v12", "v13", "v16", "v17", "v18", "v20", "v21", "v24", "v25", "v28", "v29", "v32", "v33", "v36", "v37", "v40", "v44", "v48", "v49", "v50", "v51"

template <int V>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *clk, int iters)
{
    extern __shared__ char pad[];
    unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("v_mov_b32 v17, 1.0\n v_mov_b32 v18, 0\n v_mov_b32 v40, 1.0\n v_mov_b32 v44, 0\n v_mov_b32 v48, 1.0\n v_mov_b32 v49, 1.0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n"
                 "v_mov_b32 v8, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v16, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v24, 1.0\n v_mov_b32 v28, 1.0\n v_mov_b32 v32, 1.0\n v_mov_b32 v36, 1.0\n"
                 "v_mov_b32 v9, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v37, 1.0\n" ::: CLOB);
    for (int i = 0; i < iters; i++)
    {
        if (V == 0) asm volatile(BLOCK64(REP8(FMA_FREE)) ::: CLOB);
        if (V == 1) asm volatile(BLOCK64(REP8(FMA_CONF)) ::: CLOB);
        if (V == 2) asm volatile(BLOCK64(PK_ALL) ::: CLOB);
        if (V == 3) asm volatile(BLOCK64(REP8(MUL_F32)) ::: CLOB);
        if (V == 4) asm volatile(BLOCK64(REP8(MUL_CONF)) ::: CLOB);
        if (V == 5) asm volatile(BLOCK64(REP8(MUL_IND_FREE)) ::: CLOB);
        if (V == 6) asm volatile(BLOCK64(REP8(MUL_IND_CONF)) ::: CLOB);
        if (V == 7) asm volatile(BLOCK64(REP8(SUB_CHAIN)) ::: CLOB);
        if (V == 8) asm volatile(BLOCK64(REP8(CND_FREE)) ::: CLOB, "vcc");
        if (V == 9) asm volatile(BLOCK64(REP8(CND_CONF)) ::: CLOB, "vcc");
        if (V == 10) asm volatile(BLOCK64(REP8(MIN_FREE)) ::: CLOB);
        if (V == 11) asm volatile(BLOCK64(REP8(MAD24_FREE)) ::: CLOB);
        if (V == 12) asm volatile(BLOCK64(REP8(MAD24_CONF)) ::: CLOB);
        if (V == 13) asm volatile(BLOCK64(REP8(FMA_LIT)) ::: CLOB);
    }
    float r;
    asm volatile("v_add_f32 %0, v8, v36" : "=v"(r)::CLOB);
    unsigned long long t1 = __builtin_readcyclecounter();
    if (r == 123456.0f)
        out[0] = r + pad[0];
    if (threadIdx.x == 0 && blockIdx.x == 0)
        clk[0] = t1 - t0;
}

static double shaderGHz()
{
    // s_memtime ticks per s_memrealtime tick (100 MHz) over a spin: done on the host side with two events instead -- the event time of
    // a known instruction count at the measured rate would be circular, so the clock is read from rocm-smi by the script; here the
    // nominal value is only a fallback
    const char *e = getenv("PROBE_GHZ");
    return e ? atof(e) : 2.4;
}

int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    int only = argc > 2 ? atoi(argv[2]) : -1;
    float *d_out;
    unsigned long long *d_clk;
    hipMalloc(&d_out, 64);
    hipMalloc(&d_clk, 64);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipFuncSetAttribute((const void *)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define ATTR(N) hipFuncSetAttribute((const void *)probe<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    ATTR(4) ATTR(5) ATTR(6) ATTR(7) ATTR(8) ATTR(9) ATTR(10) ATTR(11) ATTR(12) ATTR(13)
#undef ATTR
    const char *names[14] = {"v_fma_f32 bank-conflict-free", "v_fma_f32 all sources in one bank", "v_pk_fma_f32", "v_mul_f32",
                             "v_mul_f32 both sources in one bank", "v_mul_f32 independent, sources in two banks", "v_mul_f32 independent, two registers of one bank",
                             "v_sub_f32", "v_cndmask_b32 sources in two banks", "v_cndmask_b32 sources in one bank", "v_min_f32", "v_mad_i32_i24 bank-conflict-free",
                             "v_mad_i32_i24 all sources in one bank", "v_fma_f32 two registers + literal"};
    const double ghz = shaderGHz();
    for (int v = 0; v < 14; v++)
        for (int W = (only < 0 && v >= 4) ? 8 : 1; W <= 8; W *= 2)
        {
            if (only >= 0 && only != v * 10 + W)
                continue;
            const size_t lds = (size_t)(160 * 1024) / W - 512; // W workgroups of 4 waves per CU = W waves per SIMD
            const int grid = cus * W;
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++)
            {
                hipEventRecord(a);
                switch (v)
                {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), lds, 0, d_out, d_clk, iters); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), lds, 0, d_out, d_clk, iters); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), lds, 0, d_out, d_clk, iters); break;
#define CASE(N) case N: hipLaunchKernelGGL(probe<N>, dim3(grid), dim3(256), lds, 0, d_out, d_clk, iters); break;
                CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
#undef CASE
                default: break;
                }
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (rep > 0 && ms < best)
                    best = ms;
            }
            unsigned long long clk = 0;
            hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost);
            const double instsPerSimd = (double)iters * 64.0 * W; // wave instructions one SIMD issues
            const double cyc = best * 1e-3 * ghz * 1e9 / instsPerSimd;
            const double flopPerInst = (v == 2 ? 256.0 : (v == 0 || v == 1 || v == 13) ? 128.0 : 64.0);
            const double tflops = instsPerSimd * flopPerInst * cus * 4 / (best * 1e-3) / 1e12;
            printf("{\"variant\": \"%s\", \"waves_per_simd\": %d, \"iters\": %d, \"event_ms\": %.4f, \"assumed_ghz\": %.3f, \"simd_cycles_per_wave_inst\": %.3f, "
                   "\"one_wave_cycles_per_inst(s_memtime)\": %.3f, \"tflops\": %.1f, \"cus\": %d}\n",
                   names[v], W, iters, best, ghz, cyc, (double)clk / ((double)iters * 64.0), tflops, cus);
        }
    return 0;
}
