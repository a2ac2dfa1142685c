#!/usr/bin/env python3
"""Writes tools/valu_peak.hip, the VALU issue-rate microbenchmark for gfx950 (MI355X), from the op table below.

    python tools/gen_valu_peak.py && hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tools/build/valu_peak tools/valu_peak.hip
    gpurun -- bash tools/valu_peak.sh ; python tools/summarize_valu_peak.py gpurun_out/valu_peak profiles/r02/valu_peak.json

Every case is an asm block of 64 instructions over 8 independent register chains, repeated `iters` times per wave.
A workgroup is 256 lanes = 4 waves = one per SIMD of its CU; W workgroups per CU are forced through the LDS size
(160 KiB / W each) and the grid is 256 * W workgroups, so each SIMD holds W waves.  Clocks: s_memtime per wave (shader
clock), s_memrealtime (100 MHz; gives the frequency), hipEvents around the launch.  The event time over the whole chip is
the figure to trust (workgroups are not spread perfectly evenly, so per-wave medians under-state a loaded SIMD); the
rocprofv3 pass of tools/valu_peak.sh adds SQ_INSTS_VALU / GRBM_GUI_ACTIVE for the same launches.
"""
import os

OPS32 = [
    # name, template: {k} = chain register 0..7, %[b] = b (vgpr u32), %[c] = c (vgpr, 1.0f), %[m] = 64-bit sgpr mask, %[s] = sgpr
    ("v_add_u32", "v_add_u32 %{k}, %{k}, %[b]"),
    ("v_sub_u32", "v_sub_u32 %{k}, %{k}, %[b]"),
    ("v_and_b32", "v_and_b32 %{k}, %{k}, %[b]"),
    ("v_or_b32", "v_or_b32 %{k}, %{k}, %[b]"),
    ("v_xor_b32", "v_xor_b32 %{k}, %{k}, %[b]"),
    ("v_mov_b32", "v_mov_b32 %{k}, %[b]"),
    ("v_lshlrev_b32", "v_lshlrev_b32 %{k}, 1, %{k}"),
    ("v_lshrrev_b32", "v_lshrrev_b32 %{k}, 1, %{k}"),
    ("v_ashrrev_i32", "v_ashrrev_i32 %{k}, 1, %{k}"),
    ("v_min_u32", "v_min_u32 %{k}, %{k}, %[b]"),
    ("v_max_i32", "v_max_i32 %{k}, %{k}, %[b]"),
    ("v_mul_f32", "v_mul_f32 %{k}, %{k}, %[c]"),
    ("v_add_f32", "v_add_f32 %{k}, %{k}, %[c]"),
    ("v_sub_f32", "v_sub_f32 %{k}, %{k}, %[c]"),
    ("v_max_f32", "v_max_f32 %{k}, %{k}, %[c]"),
    ("v_min_f32", "v_min_f32 %{k}, %{k}, %[c]"),
    ("v_fmac_f32", "v_fmac_f32 %{k}, %[c], %[c]"),
    ("v_fma_f32", "v_fma_f32 %{k}, %{k}, %[c], %[c]"),
    ("v_mul_i32_i24", "v_mul_i32_i24 %{k}, %{k}, %[b]"),
    ("v_mul_u32_u24", "v_mul_u32_u24 %{k}, %{k}, %[b]"),
    ("v_mad_i32_i24", "v_mad_i32_i24 %{k}, %{k}, %[b], %[c]"),
    ("v_mad_u32_u24", "v_mad_u32_u24 %{k}, %{k}, %[b], %[c]"),
    ("v_mul_lo_u32", "v_mul_lo_u32 %{k}, %{k}, %[b]"),
    ("v_add3_u32", "v_add3_u32 %{k}, %{k}, %[b], %[c]"),
    ("v_lshl_add_u32", "v_lshl_add_u32 %{k}, %{k}, 1, %[c]"),
    ("v_or3_b32", "v_or3_b32 %{k}, %{k}, %[b], %[c]"),
    ("v_and_or_b32", "v_and_or_b32 %{k}, %{k}, %[b], %[c]"),
    ("v_bfi_b32", "v_bfi_b32 %{k}, %[b], %{k}, %[c]"),
    ("v_bfe_u32", "v_bfe_u32 %{k}, %{k}, 1, 31"),
    ("v_perm_b32", "v_perm_b32 %{k}, %{k}, %[b], %[c]"),
    ("v_alignbit_b32", "v_alignbit_b32 %{k}, %{k}, %[b], 3"),
    ("v_dot4_u32_u8", "v_dot4_u32_u8 %{k}, %[b], %[c], %{k}"),
    ("v_dot4c_i32_i8", "v_dot4c_i32_i8 %{k}, %[b], %[c]"),
    ("v_dot2_i32_i16", "v_dot2_i32_i16 %{k}, %[b], %[c], %{k}"),
    ("v_pk_mad_u16", "v_pk_mad_u16 %{k}, %{k}, %[b], %[c]"),
    ("v_pk_add_u16", "v_pk_add_u16 %{k}, %{k}, %[b]"),
    ("v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 %{k}, %{k}"),
    ("v_cvt_f32_ubyte2", "v_cvt_f32_ubyte2 %{k}, %{k}"),
    ("v_cvt_f32_i32", "v_cvt_f32_i32 %{k}, %{k}"),
    ("v_cvt_i32_f32", "v_cvt_i32_f32 %{k}, %{k}"),
    ("v_rndne_f32", "v_rndne_f32 %{k}, %{k}"),
    ("v_min3_f32", "v_min3_f32 %{k}, %{k}, %[c], %[b]"),
    ("v_med3_f32", "v_med3_f32 %{k}, %{k}, %[c], %[b]"),
    ("v_rcp_f32", "v_rcp_f32 %{k}, %{k}"),
    ("v_sqrt_f32", "v_sqrt_f32 %{k}, %{k}"),
    ("v_add_u32_sdwa", "v_add_u32_sdwa %{k}, %{k}, %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2"),
    ("v_mov_b32_dpp_quad", "v_mov_b32_dpp %{k}, %{k} quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf"),
    ("v_add_u32_dpp_row_shr", "v_add_u32_dpp %{k}, %{k}, %{k} row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_cmp_lt_f32_vcc", "v_cmp_lt_f32 vcc, %{k}, %[c]"),
    ("v_cmp_lt_u32_sgpr", "v_cmp_lt_u32_e64 %[m], %{k}, %[b]"),
    ("v_cndmask_b32_vcc_chain", "v_cndmask_b32 %{k}, %{k}, %[b], vcc"),
    ("v_cndmask_b32_vcc_indep", "v_cndmask_b32 %{k}, %[c], %[b], vcc"),
    ("v_cndmask_b32_sgpr_chain", "v_cndmask_b32_e64 %{k}, %{k}, %[b], %[m]"),
    ("v_cndmask_b32_sgpr_indep", "v_cndmask_b32_e64 %{k}, %[c], %[b], %[m]"),
    ("v_readlane_b32", "v_readlane_b32 %[s], %{k}, 3"),
    ("v_readfirstlane_b32", "v_readfirstlane_b32 %[s], %{k}"),
    ("v_writelane_b32", "v_writelane_b32 %{k}, %[s], 5"),
    ("v_mbcnt_lo", "v_mbcnt_lo_u32_b32 %{k}, %[b], %{k}"),
    ("ds_bpermute_b32", "ds_bpermute_b32 %{k}, %[b], %{k}"),
    ("ds_swizzle_b32", "ds_swizzle_b32 %{k}, %{k} offset:swizzle(QUAD_PERM,1,2,3,0)"),
    ("s_and_b32(salu)", "s_and_b32 %[s], %[s], 0x7fff"),
]
MIX = [
    ("mix_3mul_1cndmask_vcc", ["v_mul_f32 %{k}, %{k}, %[c]"] * 3 + ["v_cndmask_b32 %{k}, %{k}, %[b], vcc"]),
    ("mix_3lshl_1cndmask_vcc", ["v_lshlrev_b32 %{k}, 1, %{k}"] * 3 + ["v_cndmask_b32 %{k}, %{k}, %[b], vcc"]),
    ("mix_1mulf32_1lshl", ["v_mul_f32 %{k}, %{k}, %[c]", "v_lshlrev_b32 %{k}, 1, %{k}"]),
    ("mix_1addu32_1perm", ["v_add_u32 %{k}, %{k}, %[b]", "v_perm_b32 %{k}, %{k}, %[b], %[c]"]),
    ("mix_cmp_cndmask_pairs", ["v_cmp_lt_u32 vcc, %{k}, %[b]", "v_cndmask_b32 %{k}, %{k}, %[b], vcc"]),
    ("mix_cmp_2cndmask_vcc", ["v_cmp_lt_u32 vcc, %{k}, %[b]"] + ["v_cndmask_b32 %{k}, %{k}, %[b], vcc"] * 2),
    ("mix_cmp_4cndmask_vcc", ["v_cmp_lt_u32 vcc, %{k}, %[b]"] + ["v_cndmask_b32 %{k}, %{k}, %[b], vcc"] * 4),
    ("mix_lshl_4cndmask_vcc", ["v_lshlrev_b32 %{k}, 1, %{k}"] + ["v_cndmask_b32 %{k}, %{k}, %[b], vcc"] * 4),
    ("mix_lshl_7cndmask_vcc", ["v_lshlrev_b32 %{k}, 1, %{k}"] + ["v_cndmask_b32 %{k}, %{k}, %[b], vcc"] * 7),
    ("mix_1pkmul_2mulf32", ["v_mul_f32 %{k}, %{k}, %[c]", "v_mul_f32 %{k}, %{k}, %[c]", "v_perm_b32 %{k}, %{k}, %[b], %[c]"]),
    ("mix_valu_salu", ["v_lshlrev_b32 %{k}, 1, %{k}", "s_and_b32 %[s], %[s], 0x7fff"]),
]
# pair matrix: even chains run X, odd chains run Y (which pairs of kinds does a SIMD overlap between waves?)
PAIR_OPS = [
    ("add_u32", "v_add_u32 %{k}, %{k}, %[b]"),
    ("mul_f32", "v_mul_f32 %{k}, %{k}, %[c]"),
    ("and_b32", "v_and_b32 %{k}, %{k}, %[b]"),
    ("mov_b32", "v_mov_b32 %{k}, %[b]"),
    ("and_lit", "v_and_b32 %{k}, 0xff00ff, %{k}"),
    ("mul_sgpr", "v_mul_f32 %{k}, %[s], %{k}"),
    ("lshl", "v_lshlrev_b32 %{k}, 1, %{k}"),
    ("max_f32", "v_max_f32 %{k}, %{k}, %[c]"),
    ("fma_f32", "v_fma_f32 %{k}, %{k}, %[c], %[c]"),
    ("mad_i24", "v_mad_i32_i24 %{k}, %{k}, %[b], %[c]"),
    ("perm", "v_perm_b32 %{k}, %{k}, %[b], %[c]"),
    ("cvt_ub0", "v_cvt_f32_ubyte0 %{k}, %{k}"),
    ("cndmask_s", "v_cndmask_b32_e64 %{k}, %{k}, %[b], %[m]"),
    ("cmp_vcc", "v_cmp_lt_f32 vcc, %{k}, %[c]"),
    ("sdwa", "v_add_u32_sdwa %{k}, %{k}, %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2"),
    ("dpp", "v_mov_b32_dpp %{k}, %{k} quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf"),
    ("rcp", "v_rcp_f32 %{k}, %{k}"),
]
for _i, (_nx, _tx) in enumerate(PAIR_OPS):
    for _ny, _ty in PAIR_OPS[_i:]:
        MIX.append(("pair:%s+%s" % (_nx, _ny), [_tx, _ty]))

# class probes: which ops overlap with a float multiply / an integer add of another wave?
PROBE_OPS = [
    ("add_f32", "v_add_f32 %{k}, %{k}, %[c]"),
    ("fmac_f32", "v_fmac_f32 %{k}, %[c], %[c]"),
    ("lshr", "v_lshrrev_b32 %{k}, 1, %{k}"),
    ("dot4_u8", "v_dot4_u32_u8 %{k}, %[b], %[c], %{k}"),
    ("cvt_i32_f32", "v_cvt_i32_f32 %{k}, %{k}"),
    ("cvt_f32_i32", "v_cvt_f32_i32 %{k}, %{k}"),
    ("rndne", "v_rndne_f32 %{k}, %{k}"),
    ("mul_i24", "v_mul_i32_i24 %{k}, %{k}, %[b]"),
    ("mul_lo", "v_mul_lo_u32 %{k}, %{k}, %[b]"),
    ("readlane", "v_readlane_b32 %[s], %{k}, 3"),
    ("min_u32", "v_min_u32 %{k}, %{k}, %[b]"),
    ("bfe", "v_bfe_u32 %{k}, %{k}, 1, 31"),
    ("or3", "v_or3_b32 %{k}, %{k}, %[b], %[c]"),
    ("lshl_add", "v_lshl_add_u32 %{k}, %{k}, 1, %[c]"),
    ("add3", "v_add3_u32 %{k}, %{k}, %[b], %[c]"),
    ("cmp_sgpr", "v_cmp_lt_u32_e64 %[m], %{k}, %[b]"),
    ("cndmask_vcc", "v_cndmask_b32 %{k}, %{k}, %[b], vcc"),
    ("pk_mad_u16", "v_pk_mad_u16 %{k}, %{k}, %[b], %[c]"),
    ("med3", "v_med3_f32 %{k}, %{k}, %[c], %[b]"),
    ("ds_swizzle", "ds_swizzle_b32 %{k}, %{k} offset:swizzle(QUAD_PERM,1,2,3,0)"),
]
for _nx, _tx in PROBE_OPS:
    MIX.append(("probe:mul_f32+%s" % _nx, ["v_mul_f32 %{k}, %{k}, %[c]", _tx]))
    MIX.append(("probe:add_u32+%s" % _nx, ["v_add_u32 %{k}, %{k}, %[b]", _tx]))
# dependent-issue latency: one chain only (every instruction waits for the previous one)
for _nx, _tx in [("add_u32", "v_add_u32 %0, %0, %[b]"), ("mul_f32", "v_mul_f32 %0, %0, %[c]"), ("lshl", "v_lshlrev_b32 %0, 1, %0"),
                 ("fma_f32", "v_fma_f32 %0, %0, %[c], %[c]"), ("mad_i24", "v_mad_i32_i24 %0, %0, %[b], %[c]"), ("perm", "v_perm_b32 %0, %0, %[b], %[c]"),
                 ("cvt_ub0", "v_cvt_f32_ubyte0 %0, %0"), ("rcp", "v_rcp_f32 %0, %0"), ("dpp", "v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf")]:
    MIX.append(("dep1:%s" % _nx, [_tx]))

# does the ORDER of a wave's instructions matter?  The same multiset of 64 instructions, clustered by kind (what a compiler
# emits for an unrolled loop) and interleaved (a plain f32 instruction between any two of another kind)
_CVT, _MUL, _ADD, _SUB = "v_cvt_f32_ubyte0 %{k}, %{k}", "v_mul_f32 %{k}, %{k}, %[c]", "v_add_f32 %{k}, %{k}, %[c]", "v_sub_f32 %{k}, %{k}, %[c]"
_MIN, _MAX, _RND, _CVI = "v_min_f32 %{k}, %{k}, %[c]", "v_max_f32 %{k}, %{k}, %[c]", "v_rndne_f32 %{k}, %{k}", "v_cvt_i32_f32 %{k}, %{k}"
_DOT, _PRM = "v_dot4_u32_u8 %{k}, %[b], %[c], %{k}", "v_perm_b32 %{k}, %{k}, %[b], %[c]"
MIX.append(("seq:32cvt_then_32mul", [_CVT] * 32 + [_MUL] * 32))
MIX.append(("seq:8cvt_8mul_x4", ([_CVT] * 8 + [_MUL] * 8) * 4))
MIX.append(("seq:4cvt_4mul_x8", ([_CVT] * 4 + [_MUL] * 4) * 8))
MIX.append(("seq:2cvt_2mul_x16", ([_CVT] * 2 + [_MUL] * 2) * 16))
MIX.append(("seq:1cvt_3mul_x16", ([_CVT] + [_MUL] * 3) * 16))
MIX.append(("seq:4cvt_12mul_x4", ([_CVT] * 4 + [_MUL] * 12) * 4))
MIX.append(("seq:16cvt_48mul", [_CVT] * 16 + [_MUL] * 48))
# the pixel step of the BC7 dual-plane search (bc7_kernel.hip evalDualFast), 1.6 pixels: as compiled, and interleaved
_PIX = [_CVT] * 4 + [_SUB] * 4 + [_MUL] * 4 + [_ADD] * 3 + [_MIN, _MAX, _RND, _MIN, _MAX, _RND] + [_MUL] * 11 + [_ADD] * 7 + [_CVI] * 1
_PIXI = [_CVT, _SUB, _MUL, _ADD, _CVT, _SUB, _MUL, _ADD, _CVT, _SUB, _MUL, _ADD, _CVT, _SUB, _MUL, _MUL, _MIN, _MUL, _MAX, _MUL, _RND, _MUL, _MIN,
         _MUL, _MAX, _MUL, _RND, _MUL, _CVI, _MUL, _ADD, _MUL, _ADD, _MUL, _ADD, _MUL, _ADD, _ADD, _ADD, _ADD]
assert sorted(_PIX) == sorted(_PIXI) and len(_PIX) == 40
MIX.append(("seq:pixel_as_compiled", (_PIX * 2)[:64]))
MIX.append(("seq:pixel_interleaved", (_PIXI * 2)[:64]))
MIX.append(("seq:dot4_perm_block_then_mul", [_PRM] * 8 + [_DOT] * 16 + [_MUL] * 40))
MIX.append(("seq:dot4_perm_spread_in_mul", ([_DOT, _MUL, _MUL, _PRM, _MUL, _DOT, _MUL, _MUL] * 8)))

OPS64 = [
    ("v_pk_mul_f32", "v_pk_mul_f32 %{k}, %{k}, %[b]"),
    ("v_pk_add_f32", "v_pk_add_f32 %{k}, %{k}, %[b]"),
    ("v_pk_fma_f32", "v_pk_fma_f32 %{k}, %{k}, %[b], %[b]"),
    ("v_mov_b64", "v_mov_b64 %{k}, %[b]"),
    ("v_lshlrev_b64", "v_lshlrev_b64 %{k}, 1, %{k}"),
    ("v_fma_f64", "v_fma_f64 %{k}, %{k}, %[b], %[b]"),
]

HEAD = r"""// GENERATED by tools/gen_valu_peak.py -- edit the op table there.
// VALU issue-rate microbenchmark for gfx950 (MI355X): how many cycles one SIMD needs per wave64 instruction, per
// instruction kind.  See the generator's docstring for the method.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

"""

KERNEL = r"""
template <int OP>
__global__ void __launch_bounds__(256) valu_kernel(uint32_t *sink, uint64_t *times, int iters)
{
    extern __shared__ uint32_t lds[];
    uint32_t a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = blockIdx.x | 0x01020304u, c = 0x3f800000u;
    double d0 = 1.0, d1 = 1.0, d2 = 1.0, d3 = 1.0, d4 = 1.0, d5 = 1.0, d6 = 1.0, d7 = 1.0; // 64-bit register pairs
    double e = 1.0;
    uint64_t mask = __builtin_amdgcn_readfirstlane(iters) * 0x9E3779B97F4A7C15ull;
    uint32_t s0 = __builtin_amdgcn_readfirstlane(iters);
    if (threadIdx.x == 0x7fffffff)
        lds[0] = 0;
    __syncthreads();
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a3), "v"(b) : "vcc");
    uint64_t t0 = __builtin_readcyclecounter();
    uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++)
    {
@CASES@    }
    uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    uint64_t t1 = __builtin_readcyclecounter();
    uint32_t acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ (uint32_t)mask;
    acc ^= (uint32_t)__double_as_longlong(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (acc == 0x12345678u)
        sink[0] = acc + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0)
    {
        size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        times[w * 2] = t1 - t0;
        times[w * 2 + 1] = r1 - r0;
    }
}

"""

HOST = r"""typedef void (*kern_t)(uint32_t *, uint64_t *, int);

template <int OP>
static kern_t getKernel() { return valu_kernel<OP>; }

template <int... I>
static void fillTable(kern_t *t, std::integer_sequence<int, I...>) { ((t[I] = getKernel<I>()), ...); }

int main(int argc, char **argv)
{
    int iters = 2000;
    int onlyOp = -1;
    int onlyW = 0;
    for (int i = 1; i < argc; i++)
    {
        if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--op") && i + 1 < argc)
        {
            const char *want = argv[++i];
            onlyOp = -2;
            for (int k = 0; k < OP_COUNT; k++)
                if (!strcmp(kOpNames[k], want))
                    onlyOp = k;
            if (onlyOp == -2 && want[0] >= '0' && want[0] <= '9')
                onlyOp = atoi(want);
        }
        else if (!strcmp(argv[i], "--waves") && i + 1 < argc) onlyW = atoi(argv[++i]);
    }
    kern_t table[OP_COUNT];
    fillTable(table, std::make_integer_sequence<int, OP_COUNT>());

    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int numCU = prop.multiProcessorCount;
    uint32_t *sink;
    uint64_t *times;
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&times, sizeof(uint64_t) * 2 * 4 * numCU * 8));
    hipEvent_t ev0, ev1;
    CHECK(hipEventCreate(&ev0));
    CHECK(hipEventCreate(&ev1));
    const int wlist[5] = {1, 2, 3, 4, 8};
    for (int op = 0; op < OP_COUNT; op++)
    {
        if (onlyOp >= 0 && op != onlyOp)
            continue;
        for (int wi = 0; wi < 5; wi++)
        {
            const int W = wlist[wi];
            if (onlyW && W != onlyW)
                continue;
            // 160 KiB of LDS per CU: W workgroups fit, W + 1 do not (W = 8: the wave slots cap it anyway)
            const size_t ldsBytes = W == 8 ? 16 * 1024 : (size_t)(160 * 1024 / W) - (W == 1 ? 0 : 1024);
            CHECK(hipFuncSetAttribute((const void *)table[op], hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
            const int grid = numCU * W;
            for (int rep = 0; rep < 3; rep++)
            {
                CHECK(hipEventRecord(ev0, 0));
                hipLaunchKernelGGL(table[op], dim3(grid), dim3(256), ldsBytes, 0, sink, times, iters);
                CHECK(hipEventRecord(ev1, 0));
                CHECK(hipDeviceSynchronize());
                if (rep < 2)
                    continue;
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, ev0, ev1));
                std::vector<uint64_t> h(2 * 4 * (size_t)grid);
                CHECK(hipMemcpy(h.data(), times, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
                std::vector<uint64_t> cyc, rt;
                for (size_t w = 0; w < (size_t)grid * 4; w++)
                {
                    cyc.push_back(h[2 * w]);
                    rt.push_back(h[2 * w + 1]);
                }
                std::sort(cyc.begin(), cyc.end());
                std::sort(rt.begin(), rt.end());
                const double insts = (double)iters * 64.0;
                const double medCyc = (double)cyc[cyc.size() / 2], medRt = (double)rt[rt.size() / 2];
                // s_memrealtime ticks at 100 MHz
                const double clockGhz = medRt > 0 ? medCyc / (medRt * 10.0) : 0.0;
                const double totalWaveInsts = insts * grid * 4.0;
                printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"wave_insts\": %.0f, \"memtime_cycles_median\": %.0f, \"memtime_cycles_max\": %.0f, "
                       "\"realtime_ticks_median\": %.0f, \"memtime_ghz\": %.4f, \"cycles_per_wave_inst_per_wave\": %.4f, "
                       "\"simd_cycles_per_wave_inst(memtime)\": %.4f, \"event_ms\": %.4f, \"wave_insts_per_s_per_simd(event)\": %.4e, "
                       "\"wave_insts_per_s_chip(event)\": %.4e}\n",
                       kOpNames[op], W, insts, medCyc, (double)cyc.back(), medRt, clockGhz, medCyc / insts, medCyc / insts / W, ms,
                       totalWaveInsts / (ms * 1e-3) / (numCU * 4.0), totalWaveInsts / (ms * 1e-3));
                fflush(stdout);
            }
        }
    }
    return 0;
}
"""


def block(templates):
    lines, n = [], 0
    for _ in range(8):
        for k in range(8):
            lines.append(templates[n % len(templates)].replace("{k}", str(k)))
            n += 1
    return "\\n".join(lines) + "\\n"


def main():
    names, cases = [], []
    for name, t in OPS32:
        names.append(name)
        cases.append(("32", block([t])))
    for name, tl in MIX:
        names.append(name)
        cases.append(("32", block(tl)))
    for name, t in OPS64:
        names.append(name)
        cases.append(("64", block([t])))
    body = []
    for i, (kind, b) in enumerate(cases):
        pre = "else " if i else ""
        if kind == "32":
            body.append('        %sif constexpr (OP == %d)\n            asm volatile("%s" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), '
                        '"+v"(a6), "+v"(a7), [m] "+s"(mask), [s] "+s"(s0) : [b] "v"(b), [c] "v"(c) : "vcc");\n' % (pre, i, b))
        else:
            body.append('        %sif constexpr (OP == %d)\n            asm volatile("%s" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), '
                        '"+v"(d6), "+v"(d7) : [b] "v"(e));\n' % (pre, i, b))
    table = "#define OP_COUNT %d\nstatic const char *kOpNames[OP_COUNT] = {\n%s};\n" % (len(names), "".join('    "%s",\n' % n for n in names))
    out = HEAD + table + KERNEL.replace("@CASES@", "".join(body)) + HOST
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_peak.hip")
    open(path, "w").write(out)
    print("wrote", path, len(names), "cases")


if __name__ == "__main__":
    main()
