// BC7 endpoint / partition / index search for gfx950 (MI355X), hand-written HIP.
//
// What it replaces: cvtt::Internal::BC7Computer::Pack and everything under it
// (reference ConvectionKernels_BC67.cpp:1975-2204 -> TrySinglePlane 1042-1662,
// TryDualPlane 1664-1965, CompressEndpoints* 829-938; EndpointSelector.h, EndpointRefiner.h,
// IndexSelector.h, AggregatedError.h).  Results are bit-identical to the reference's SSE2
// lanes; the arithmetic contract is kept by compiling this file with
// -ffp-contract=off and IEEE divide/sqrt, f32 denormals on.
//
// Mapping (this is not the reference's 8-blocks-per-SSE-register layout; DESIGN.md 4.1 has the measurements):
//   * one wavefront = 16 blocks = two reference "groups" of 8; lanes [0,32) and [32,64) each
//     form one group, whose two alpha-derived booleans (BC67.cpp:1069, 1072) are wave ballots.
//   * exact branch-and-bound: a rigorous lower bound on the error of ANY trial of a shape (its
//     total-least-squares residual minus the allowance for rounding, cvtt_kernel_common.h) is compared with the
//     block's best error so far; a candidate that cannot pass the reference's commit test is skipped, which leaves
//     the output bit-identical.  Whole-block bounds order and prune the four rotations of modes 4/5 and mode 6;
//     bounds of all 64 partitions (2-D projection, v_dot4 / v_dot2 masked sums) go to LDS per mode family, and the
//     partitions they leave alive get the same bound in all channels of each subset (raw integer sums, second tier).
//   * dual-plane modes 4/5: a block is owned by a lane QUAD, sub-lane = seed point; the block is held channel-major,
//     the float index selection and refiner sums are the reference's operations, the integer error comes from level
//     tables + v_perm_b32 + v_dot4_u32_u8 (evalDualFast; evalDual for slow indexing).
//   * single-plane modes 0-3, 6, 7: work pooled over the wave.  Per round the blocks offer their cheapest-bound
//     partitions, one lane per (item, subset) runs the PCA seed search on LDS-staged pixels, then one lane per CHAIN
//     (p-bit combination x seed point, with its serial refine rounds: evalChain) evaluates it; an order-preserving
//     argmin reproduces the reference's first-minimum rule and the owning block commits by (error, position in the
//     reference's candidate order), so the evaluation order is free.
//   * BC7_RespectPunchThrough (separate instantiation): the commit rule couples the 8 blocks of a group per trial;
//     every trial's error is recorded in LDS and the rule is replayed over 8-lane ballot slices.
//   * modes 4 / 5 are packed by one lane at constant bit positions; for the other modes the quad packs the (up to 66)
//     bit fields of its block in parallel; sub-lane 0 stores the 16 bytes.
//   * large inputs: a wave that would still have many mode-7 partitions to search near the end of the grid hands those
//     blocks to a second launch (HARD instantiation, 16 waves per block) and a commit launch picks the winner.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <type_traits>
#include <stdlib.h>

#include "cvtt_device.h"

// BC7_RespectPunchThrough: refine rounds whose trial errors are kept in LDS (2 KB per round; the reference's default is 2
// rounds).  With more rounds the table of a wave lives in HBM (CvttBc7Args::ptTrial).  6 rounds of LDS held this
// instantiation at 6 workgroups per CU; 2 rounds: 10 (2.5 waves per SIMD).
constexpr int kMaxPTRefine = 2;
// half-range of the integer grid the projected points are rounded to: 12-bit coordinates summed with v_dot2_i32_i16, or
// 8-bit ones summed with v_dot4_i32_i8 (half the instructions, bounds looser by the coarser rounding)
constexpr float kBoundGrid16 = 2000.0f, kBoundLimit16 = 2040.0f;
constexpr float kBoundGrid8 = 124.0f, kBoundLimit8 = 127.0f;
// which grid the first-tier bounds of the RGBA partitions (mode 7) and of the RGB partitions (modes 0-3) use.  With the
// second tier behind them the cheaper bounds win everywhere (RGBA noise +3 %, photo-like opaque content +6 %, opaque
// noise equal); the 12-bit grid stays selectable for experiments.
#ifndef CVTT_GRID8_RGBA
#define CVTT_GRID8_RGBA true
#endif
#ifndef CVTT_GRID8_RGB
#define CVTT_GRID8_RGB true
#endif
// Waves per SIMD the register allocator must leave room for (512 VGPRs / waves).  4 since round 2: the kernel fits 8 LDS
// granules (16-bit bound table, 64 result slots), and the dual-plane search reads the block's channel-major pixels, the
// rotation's seeds and the per-block invariants from LDS where it uses them, which brought the 128-register build from
// 43 spilled registers to 33.  Measured 3 -> 4 waves: RGBA noise 630 -> 672 Mblocks/s at 4096^2, opaque noise 136 -> 150,
// smooth / photo-like / two-colour content +13 % (chain rounds are plain f32 arithmetic, which an EVEN number of resident
// waves overlaps: profiles/r02/valu_order.txt).
#ifndef CVTT_BC7_WAVES
#define CVTT_BC7_WAVES 4
#endif
// the same for the slow-indexing instantiation (Flags::Better / Ultra), whose three probes per pixel need more registers
// (4 since round 3: with the pixel loop of its dual-plane search rolled into four trips the instantiation fits 128
// registers; Flags::Better on 4096^2 RGBA noise 292 -> 306 Mblocks/s, opaque 75 -> 78)
#ifndef CVTT_BC7_WAVES_SLOW
#define CVTT_BC7_WAVES_SLOW 4
#endif
// power iterations per principal axis of the projection the first-tier bounds are taken in (tightness only, never validity)
#ifndef CVTT_EIG_ITERS
#define CVTT_EIG_ITERS 6
#endif

#include "cvtt_kernel_common.h"

// candidates a block may offer per round when 1 / 2 / <=4 / <=8 blocks of the wave offer at all
#ifndef CVTT_SPEC_1
#define CVTT_SPEC_1 32
#define CVTT_SPEC_2 16
#define CVTT_SPEC_4 8
#define CVTT_SPEC_8 4
#endif
#ifndef CVTT_SPEC_16
#define CVTT_SPEC_16 2
#endif
// probe survivors a wave collects before it searches them in full (one full-search chunk holds 32 / 21 partitions)
#ifndef CVTT_FILTER_STRIKES
#define CVTT_FILTER_STRIKES 2
#endif
#ifndef CVTT_SHARP_COST4
#define CVTT_SHARP_COST4 64
#endif
#ifndef CVTT_SHARP_COST8
#define CVTT_SHARP_COST8 20
#endif
#ifndef CVTT_PEND_MIN
#define CVTT_PEND_MIN 20
#endif

// Developer-only phase profile (-DCVTT_BC7_PROFILE): wave cycles per phase, summed over waves.
#ifdef CVTT_BC7_PROFILE
__device__ unsigned long long g_bc7Prof[48];
__device__ unsigned long long g_bc7Dup[8]; // chain rounds: total, same endpoints as a lower seed point this round, seen in an earlier round of the group
// per single-plane stage (6, 7, 1, 3, 0, 2): chain batches, units searched, partitions alive when the stage starts (after
// both bound tiers), offer rounds, partitions committed as the block's new best, active chain lanes, wave-stages entered
__device__ unsigned long long g_bc7Stage[6][8];
__device__ unsigned long long g_bc7Stage2[6][8]; // spare
extern "C" int cvttmi_bc7_stage_read(unsigned long long *out)
{
    unsigned long long zero[48] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bc7Stage), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bc7Stage), zero, sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out + 48, HIP_SYMBOL(g_bc7Stage2), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bc7Stage2), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#define PROF_STAGE(stage, slot, n) { const unsigned long long v_ = (unsigned long long)(n); if (threadIdx.x == 0 && v_) atomicAdd(&g_bc7Stage[stage][slot], v_); }
#define PROF_STAGE_LANES(stage, slot, pred) { const unsigned long long v_ = (unsigned long long)__popcll(__ballot(pred)); if (threadIdx.x == 0 && v_) atomicAdd(&g_bc7Stage[stage][slot], v_); }
#define PROF_STAGE2(stage, slot, n) { const unsigned long long v_ = (unsigned long long)(n); if (threadIdx.x == 0 && v_) atomicAdd(&g_bc7Stage2[stage][slot], v_); }
extern "C" int cvttmi_bc7_dup_read(unsigned long long *out)
{
    unsigned long long zero[8] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bc7Dup), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bc7Dup), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#define PROF_DECL unsigned long long profT = __builtin_readcyclecounter(); unsigned long long profAcc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long profCnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PROF_MARK(slot) { const unsigned long long now = __builtin_readcyclecounter(); profAcc[slot] += now - profT; profT = now; }
#define PROF_FLUSH if (threadIdx.x == 0) { unsigned long long tot = 0; for (int i = 0; i < 16; i++) { atomicAdd(&g_bc7Prof[i], profAcc[i]); tot += profAcc[i]; } \
    int bucket = 63 - __builtin_clzll(tot | 1ull) - 12; bucket = bucket < 0 ? 0 : (bucket > 15 ? 15 : bucket); atomicAdd(&g_bc7Prof[16 + bucket], 1ull); \
    } { unsigned long long wsum[8]; for (int i = 0; i < 8; i++) { unsigned long long v = profCnt[i]; if (i >= 4) { for (int st = 1; st < 64; st <<= 1) v += __shfl_xor(v, st); } wsum[i] = v; } \
    if (threadIdx.x == 0) { for (int i = 0; i < 8; i++) atomicAdd(&g_bc7Prof[32 + i], wsum[i]); \
    const unsigned long long passes = profCnt[2] / 64ull; int pb = 63 - __builtin_clzll(passes | 1ull); pb = pb > 7 ? 7 : pb; atomicAdd(&g_bc7Prof[40 + pb], 1ull); } }
#define PROF_COUNT(slot, n) { profCnt[slot] += (unsigned long long)(n); }
extern "C" int cvttmi_bc7_prof_read(unsigned long long *out)
{
    unsigned long long zero[48] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bc7Prof), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bc7Prof), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#else
#define PROF_DECL
#define PROF_MARK(slot)
#define PROF_COUNT(slot, n)
#define PROF_FLUSH
#define PROF_STAGE(stage, slot, n)
#define PROF_STAGE2(stage, slot, n)
#define PROF_STAGE_LANES(stage, slot, pred)
#endif

// Developer-only trial trace (-DCVTT_BC7_DEBUG): per-round results of the dual-plane search of one block.
#ifdef CVTT_BC7_DEBUG
__device__ float g_bc7Dbg[2048];
__device__ unsigned g_bc7DbgBlock = 0xffffffffu;
extern "C" int cvttmi_bc7_debug_set(unsigned block)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bc7DbgBlock), &block, sizeof(block)) == hipSuccess ? 0 : -1;
}
extern "C" int cvttmi_bc7_debug_read(float *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bc7Dbg), sizeof(float) * 2048) == hipSuccess ? 0 : -1;
}
#endif

namespace
{
// ---- BC7 endpoint quantisation (reference BC67.cpp:829-860); all values fit 16 bits ----
__device__ __forceinline__ int quantizeNoP(int v, int bits) { return ((v << bits) - v + (127 + (1 << (7 - bits)))) >> 8; }
__device__ __forceinline__ int quantizeP(int v, int bits, int p)
{
    const int addend = p ? ((1 << (8 - bits)) - 1) : 255;
    const int q = ((v << (bits + 1)) - v + addend) >> 9;
    return (q << 1) | p;
}
__device__ __forceinline__ int unquantize(int v, int bits)
{
    const int t = v << (8 - bits);
    return t | (t >> bits);
}

// Static description of a single-plane mode (BC7 format + reference BC67.cpp:862-938).
struct ModeDesc
{
    int mode;
    int indexBits;
    int numP;     // parity combinations: 4 per-endpoint, 2 per-subset, 1 none
    int quantBits;
    int unquantBits; // 0 = value is already 8 bit
};

__device__ __forceinline__ void compressEndpoints(const ModeDesc &md, int (&ep)[2][4], int pIter, bool isRGB)
{
    const int nch = isRGB ? 3 : 4;
#pragma unroll
    for (int j = 0; j < 2; j++)
    {
        // p-bit of endpoint j: per-endpoint modes use bit j of pIter, per-subset modes share
        // bit 0 (reference BC67.cpp:1307-1309, 872-880)
        const int p = (md.numP == 4) ? ((pIter >> j) & 1) : (pIter & 1);
#pragma unroll
        for (int ch = 0; ch < 4; ch++)
        {
            if (ch < nch)
            {
                int v = ep[j][ch];
                v = (md.numP == 1) ? quantizeNoP(v, md.quantBits) : quantizeP(v, md.quantBits, p);
                if (md.unquantBits)
                    v = unquantize(v, md.unquantBits);
                ep[j][ch] = v;
            }
        }
        if (isRGB)
            ep[j][3] = 255;
    }
}

struct ShapeBest
{
    float err;
    u32 ep0, ep1; // packed RGBA endpoints
    u32 idxLo, idxHi; // 4 bits per pixel, by pixel position
};

__device__ __forceinline__ u32 packEP(const int (&e)[4])
{
    return (u32)e[0] | ((u32)e[1] << 8) | ((u32)e[2] << 16) | ((u32)e[3] << 24);
}

// Order-preserving argmin over the 4 lanes of a quad; ties go to the lower sub-lane, which
// is the earlier candidate in the reference's sequential commit order.
__device__ __forceinline__ void quadArgminBroadcast(ShapeBest &b, int lane)
{
    int who = lane & 3;
    float err = b.err;
#pragma unroll
    for (int step = 1; step <= 2; step <<= 1)
    {
        const float oErr = xorLane(err, step);
        const int oWho = xorLane(who, step);
        const bool take = (oErr < err) || (oErr == err && oWho < who);
        err = take ? oErr : err;
        who = take ? oWho : who;
    }
    const int src = (lane & ~3) | who;
    b.err = err;
    b.ep0 = __shfl(b.ep0, src);
    b.ep1 = __shfl(b.ep1, src);
    b.idxLo = __shfl(b.idxLo, src);
    b.idxHi = __shfl(b.idxHi, src);
}

// One CHAIN of a single-plane shape: a (p-bit combination, seed point) pair of the reference's
// pIter x tweak loops (BC67.cpp:1298-1434) with its refine rounds, run by one lane.  The
// pixels of the block are read from LDS (`lp`, 16 packed RGBA8 words) in ascending order of
// the shape's members.  NRC = numRealChannels (3 for modes 0-3).  maxCount = wave-uniform
// upper bound on popcount(mask).
// TRACE (BC7_RespectPunchThrough only): every round's error is also written to trialErr[round], and with
// captureRound >= 0 the result is that round's instead of the best one.
// `list`: the member pixels in ascending order, 4 bits each (UnitRec::listLo / listHi); best.idxLo / idxHi come back COMPACT: the
// index of member i in bits [4i, 4i + 4) (expandIndexes puts them at their pixel positions when a block is packed).
// wantPayload (wave-uniform) = false: only best.err is wanted (probes).
// A chain cut into pieces (the probes run round 0 of every chain first and the later rounds only once per distinct set of
// end points): rounds [firstRound, endRound) are run.  firstRound > 0: ep0 / ep1 bring that round's (compressed) end points.
// endRound < numRefine: ep0 / ep1 take the compressed end points of round endRound away.  r0ep0 / r0ep1: the compressed
// end points the first executed round used.  (Wave-uniform except for the end points.)
struct ChainSplit
{
    int firstRound, endRound;
    u32 ep0, ep1;
    u32 r0ep0, r0ep1;
};

template <int NRC, bool FAST, bool TRACE>
__device__ __forceinline__ void evalChain(const u32 *lp, u32 mask, u64 list, int maxCount, const ModeDesc md, const Unfinished &u,
                                          int pIter, int tweak, bool active, const CvttBc7Args &A,
                                          const CvttDeviceTables *__restrict__ T, int numRefine, ShapeBest &best,
                                          const float (&vs)[4], bool wantPayload = true, float *trialErr = nullptr, int captureRound = -1, float *trialErrHbm = nullptr,
                                          ChainSplit *split = nullptr)
{
    const bool isRGB = (NRC == 3);
    const int range = 1 << md.indexBits;
    // (literals: the clamp below then needs no canonicalising v_max_f32 per pixel)
    const float maxValue = md.indexBits == 2 ? 3.0f : md.indexBits == 3 ? 7.0f : 15.0f;
    const float rcpMaxIndex = T->rcpMaxIndex[md.indexBits];
    const int weightRcp = (65536 + (range - 1)) / (2 * (range - 1)); // g_weightReciprocals, IndexSelector.cpp:43-62
    const int count = __popc(mask);
    const float wRcp = T->rcpTable[count];
    const float wCount = (float)count;
    const bool uniformErr = (A.flags & CVTTMI_FLAG_UNIFORM) != 0;

    best.err = FLT_MAX;
    best.ep0 = best.ep1 = 0;
    best.idxLo = best.idxHi = 0;
    const int myCount = active ? count : 0; // member i exists for i < myCount
    const u32 listLo = (u32)list, listHi = (u32)(list >> 32);
    // pixel id of member i (i is wave-uniform: the halves of the list are told apart by a scalar branch)
    auto memberOf = [&](int i) -> int {
        return (int)(i < 8 ? __builtin_amdgcn_ubfe(listLo, (u32)(4 * i), 4u) : __builtin_amdgcn_ubfe(listHi, (u32)(4 * i - 32), 4u));
    };
#ifdef CVTT_BC7_PROFILE
    u32 profHist[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#endif

    // static alpha error of RGB modes (reference BC67.cpp:1250-1264); zero on opaque blocks
    float staticAlphaError = 0.0f;
    if (isRGB)
    {
        u32 acc = 0;
        for (int i = 0; i < maxCount; i++)
        {
            if (i < myCount)
            {
                const int d = 255 - byteI(lp[memberOf(i)], 3);
                acc = (u32)mad24(d, d, (int)acc);
            }
        }
        staticAlphaError = uniformErr ? (float)(int)acc : (float)(int)acc * A.wSq[3];
    }

    // UnfinishedEndpoints::FinishLDR (reference UnfinishedEndpoints.h:77-91)
    const float tf0 = T->tweakFactors[md.indexBits - 2][tweak][0];
    const float tf1 = T->tweakFactors[md.indexBits - 2][tweak][1];
    int ep[2][4];
    const int firstRound = split ? split->firstRound : 0, endRound = split ? split->endRound : numRefine;
    if (firstRound > 0)
    {
#pragma unroll
        for (int ch = 0; ch < 4; ch++)
        {
            ep[0][ch] = byteI(split->ep0, ch);
            ep[1][ch] = byteI(split->ep1, ch);
        }
    }
    else
    {
#pragma unroll
        for (int ch = 0; ch < 4; ch++)
        {
            if (ch < NRC)
            {
                ep[0][ch] = (int)clampRound(u.base[ch] + u.offset[ch] * tf0, 255.0f);
                ep[1][ch] = (int)clampRound(u.base[ch] + u.offset[ch] * tf1, 255.0f);
            }
            else
                ep[0][ch] = ep[1][ch] = 255;
        }
    }

    for (int refine = firstRound; refine < endRound; refine++)
    {
        const bool last = (refine == numRefine - 1);
        if (refine > firstRound || firstRound == 0)
            compressEndpoints(md, ep, pIter, isRGB);
        if (split && refine == firstRound)
        {
            split->r0ep0 = packEP(ep[0]);
            split->r0ep1 = packEP(ep[1]);
        }
#ifdef CVTT_BC7_PROFILE
        {
            const u32 k0 = packEP(ep[0]), k1 = packEP(ep[1]);
            const int ln = (int)threadIdx.x;
            bool dupNow = false, dupOld = false;
            for (int o = 0; o < 4; o++)
            {
                const int src = (ln & ~3) | o;
                const u32 a0 = __shfl(k0, src), a1 = __shfl(k1, src);
                const bool oAct = __shfl((int)active, src) != 0;
                if (oAct && o < (ln & 3) && a0 == k0 && a1 == k1) dupNow = true;
                for (int h = 0; h < refine && h < 4; h++)
                {
                    const u32 b0 = __shfl(profHist[h][0], src), b1 = __shfl(profHist[h][1], src);
                    if (oAct && b0 == k0 && b1 == k1) dupOld = true;
                }
            }
            if (refine < 4) { profHist[refine][0] = k0; profHist[refine][1] = k1; }
            const u64 mA = __ballot(active), mN = __ballot(active && dupNow), mO = __ballot(active && !dupNow && dupOld);
            if (ln == 0)
            {
                atomicAdd(&g_bc7Dup[0], (unsigned long long)__popcll(mA));
                atomicAdd(&g_bc7Dup[1], (unsigned long long)__popcll(mN));
                atomicAdd(&g_bc7Dup[2], (unsigned long long)__popcll(mO));
                atomicAdd(&g_bc7Dup[3 + (refine < 3 ? refine : 3)], (unsigned long long)__popcll(mN | mO));
            }
        }
#endif

        // IndexSelector<4>::Init (reference IndexSelector.h:27-77)
        float origin[4], axis[4];
        int recBase[4], recDelta[4];
        {
            float epDW[4];
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
            {
                origin[ch] = (float)ep[0][ch];
                epDW[ch] = ((float)ep[1][ch] - origin[ch]) * A.w[ch];
                // scaled by 4: the reconstructed value (x >> 6) is then byte 1 of the sum
                recBase[ch] = ((ep[0][ch] << 6) + 32) << 2;
                recDelta[ch] = (ep[1][ch] - ep[0][ch]) << 2;
            }
            float lenSq = epDW[0] * epDW[0];
#pragma unroll
            for (int ch = 1; ch < 4; ch++)
                lenSq = lenSq + epDW[ch] * epDW[ch];
            lenSq = safeDenom(lenSq);
            const float mvdls = maxValue / lenSq;
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
                axis[ch] = epDW[ch] * A.w[ch] * mvdls;
        }

        const v2f org01 = {origin[0], origin[1]}, org23 = {origin[2], origin[3]};
        const v2f ax01 = {axis[0], axis[1]}, ax23 = {axis[2], axis[3]};
        const v2f w01 = {A.w[0], A.w[1]}, w23 = {A.w[2], A.w[3]};
        u32 err[4] = {0, 0, 0, 0};
        float slowErr = 0.0f;
        v2f tv01 = {0.0f, 0.0f}, tv23 = {0.0f, 0.0f};
        float tt = 0.0f, ts = 0.0f;
        u32 idxLo = 0, idxHi = 0; // compact: member i in bits [4i, 4i + 4)

        for (int i = 0; i < maxCount; i++)
        {
            if (i < myCount)
            {
                const u32 pk = lp[memberOf(i)];
                // SelectIndexLDR (reference IndexSelector.h:124-131)
                const v2f x01 = {byteF(pk, 0), byteF(pk, 1)}, x23 = {byteF(pk, 2), byteF(pk, 3)};
                const v2f p01 = (x01 - org01) * ax01, p23 = (x23 - org23) * ax23;
                float dist = p01.x + p01.y;
                dist = dist + p23.x;
                dist = dist + p23.y;
                float fidx = clampRound(dist, maxValue);
                int index = (int)fidx;

                if (FAST)
                {
                    // ReconstructLDR_BC7 + ComputeErrorLDR (IndexSelector.h:90-100, BCCommon.h:24-29)
                    const int wgt = mad24(weightRcp, index, 256) >> 9;
                    err[0] = accumulateChannelError<0>(wgt, recDelta[0], recBase[0], pk, err[0]);
                    err[1] = accumulateChannelError<1>(wgt, recDelta[1], recBase[1], pk, err[1]);
                    err[2] = accumulateChannelError<2>(wgt, recDelta[2], recBase[2], pk, err[2]);
                    if (NRC == 4)
                        err[3] = accumulateChannelError<3>(wgt, recDelta[3], recBase[3], pk, err[3]);
                }
                else
                {
                    // slow indexing: also probe index-1 / index+1 (reference BC67.cpp:1367-1386)
                    float bestE = 0.0f;
                    const int index0 = index;
#pragma unroll
                    for (int probe = 0; probe < 3; probe++)
                    {
                        int cand = index0;
                        if (probe == 1) cand = (index0 > 1 ? index0 : 1) - 1;
                        if (probe == 2) cand = (index0 + 1 < range - 1) ? index0 + 1 : range - 1;
                        const int wgt = mad24(weightRcp, cand, 256) >> 9;
                        // the reconstructed channel is byte 1 of w * delta4 + base4; its difference to the pixel and the
                        // square are taken as floats (exact on these integers, see evalDual: plain f32 instructions)
                        const float xs[4] = {x01.x, x01.y, x23.x, x23.y};
                        float e4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int ch = 0; ch < NRC; ch++)
                        {
                            const float d = byteF((u32)madI24(wgt, recDelta[ch], recBase[ch]), 1) - xs[ch];
                            e4[ch] = d * d;
                        }
                        float e;
                        if (uniformErr)
                            e = ((e4[0] + e4[1]) + e4[2]) + e4[3];
                        else
                        {
                            e = e4[0] * A.wSq[0];
                            e = e + e4[1] * A.wSq[1];
                            e = e + e4[2] * A.wSq[2];
                            e = e + e4[3] * A.wSq[3];
                        }
                        if (probe == 0)
                            bestE = e;
                        else
                        {
                            // both alternatives derive from the initial index (BC67.cpp:1367-1386); the winner is tracked as an
                            // integer and converted once, and the comparison that moves it also serves the minimum
                            const bool better = e < bestE;
                            bestE = better ? e : bestE;
                            index = better ? cand : index;
                        }
                    }
                    slowErr = slowErr + bestE;
                    fidx = (float)index;
                }

                if (!last)
                {
                    // EndpointRefiner::ContributeUnweightedPW (EndpointRefiner.h:78-92)
                    const float t = fidx * rcpMaxIndex;
                    const v2f t2 = {t, t};
                    const v2f v01 = x01 * w01, v23 = x23 * w23; // channel 3 is unused when NRC == 3
                    tv01 = tv01 + t2 * v01;
                    tv23 = tv23 + t2 * v23;
                    tt = tt + t * t;
                    ts = ts + t;
                }
                if (wantPayload)
                {
                    if (i < 8)
                        idxLo |= (u32)index << (4 * i);
                    else
                        idxHi |= (u32)index << (4 * i - 32);
                }
            }
        }

        // AggregatedError<4>::Finalize (reference AggregatedError.h:29-46)
        float shapeError;
        if (FAST)
        {
            if (uniformErr)
                shapeError = (float)(int)(err[0] + err[1] + err[2] + err[3]);
            else
            {
                shapeError = (float)(int)err[0] * A.wSq[0];
                shapeError = shapeError + (float)(int)err[1] * A.wSq[1];
                shapeError = shapeError + (float)(int)err[2] * A.wSq[2];
                shapeError = shapeError + (float)(int)err[3] * A.wSq[3];
            }
        }
        else
            shapeError = slowErr;
        if (isRGB)
            shapeError = shapeError + staticAlphaError;

        if (TRACE && active)
        {
            // (two pointers, not one: the LDS stores stay ds_write instead of turning into flat stores)
            if (trialErrHbm)
                trialErrHbm[refine] = shapeError;
            else if (trialErr)
                trialErr[refine] = shapeError;
        }
        const bool take = (TRACE && captureRound >= 0) ? (refine == captureRound) : (shapeError < best.err);
        if (active && take)
        {
            best.err = shapeError;
            best.ep0 = packEP(ep[0]);
            best.ep1 = packEP(ep[1]);
            best.idxLo = idxLo;
            best.idxHi = idxHi;
        }

        if (!last)
        {
            // EndpointRefiner::GetRefinedEndpointsLDR (EndpointRefiner.h:99-152)
            const float tv[4] = {tv01.x, tv01.y, tv23.x, tv23.y};
            float adenom = (tt * wCount - ts * ts) * wRcp;
            const bool adenomZero = (adenom == 0.0f);
            if (adenomZero)
                adenom = 1.0f;
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
            {
                if (ch < NRC)
                {
                    const float a = (tv[ch] - ts * vs[ch] * wRcp) / adenom;
                    const float b = (vs[ch] - a * ts) * wRcp;
                    float p1 = b;
                    float p2 = a + b;
                    if (adenomZero)
                    {
                        p1 = vs[ch] * wRcp;
                        p2 = p1;
                    }
                    ep[0][ch] = (int)clampRound(p1 * A.rcpW[ch], 255.0f);
                    ep[1][ch] = (int)clampRound(p2 * A.rcpW[ch], 255.0f);
                }
                else
                    ep[0][ch] = ep[1][ch] = 0; // overwritten with 255 by compressEndpoints
            }
        }
    }
    if (split && endRound < numRefine)
    {
        compressEndpoints(md, ep, pIter, isRGB);
        split->ep0 = packEP(ep[0]);
        split->ep1 = packEP(ep[1]);
    }
}

// Order-preserving argmin over aligned groups of `width` lanes (4, 8 or 16): ties go to the
// lower lane, i.e. the earlier chain in the reference's sequential commit order.
__device__ __forceinline__ void groupArgminBroadcast(ShapeBest &b, int lane, int width)
{
    int who = lane;
    float err = b.err;
    // (width is 4, 8 or 16, wave-uniform: the steps are spelled out so that each is a ds_swizzle with a literal pattern)
#define CVTT_GROUP_STEP(STEP)                                             \
    {                                                                     \
        const float oErr = xorLane(err, STEP);                            \
        const int oWho = xorLane(who, STEP);                              \
        const bool take = (oErr < err) || (oErr == err && oWho < who);    \
        err = take ? oErr : err;                                          \
        who = take ? oWho : who;                                          \
    }
    CVTT_GROUP_STEP(1)
    CVTT_GROUP_STEP(2)
    if (width > 4)
        CVTT_GROUP_STEP(4)
    if (width > 8)
        CVTT_GROUP_STEP(8)
#undef CVTT_GROUP_STEP
    b.err = err;
    b.ep0 = __shfl(b.ep0, who);
    b.ep1 = __shfl(b.ep1, who);
    b.idxLo = __shfl(b.idxLo, who);
    b.idxHi = __shfl(b.idxHi, who);
}

// Pixel source for the PCA passes of a block staged in LDS.
struct FetchLDS
{
    const u32 *lp;
    const float (&w)[4];
    template <int N>
    __device__ __forceinline__ void get(int px, float (&v)[N]) const
    {
        const u32 pk = lp[px];
#pragma unroll
        for (int ch = 0; ch < N; ch++)
            v[ch] = byteF(pk, ch) * w[ch];
    }
};

template <int N>
__device__ __forceinline__ void pcaEndpointsLDS(const u32 *lp, u32 mask, const float (&w)[4], Unfinished &u, float *sums)
{
    const FetchLDS F = {lp, w};
    Moments<N> m;
    pcaMomentsT<N>(F, mask, m, sums);
    pcaFinishT<N>(F, mask, w, m, u);
}

// What one lane of the PCA pass leaves for the chain lanes of its (block, partition, subset).
struct UnitRec
{
    float base[4];
    float offset[4];
    u32 packed;  // mask | numTweak << 16 | blk << 20 | slot << 24  (slot = item * subsets + subset: where the subset's result goes)
    float scErr; // BC7_TrySingleColor: error of the fixed candidate (FLT_MAX when not tried)
    // the refiner's m_v of the subset: sum of the pre-weighted member pixels in ascending order (EndpointRefiner.h:78-92).
    // It does not depend on the indexes, so the seed lane takes it once for all chains and rounds of the unit.
    float vs[4];
    // the member pixels of the subset in ascending order, 4 bits each: the chain lanes read pixel i of their subset with one
    // bit-field extract at a wave-uniform position instead of popping a per-lane bit mask
    u32 listLo, listHi;
    __device__ __forceinline__ u32 mask() const { return packed & 0xffffu; }
    __device__ __forceinline__ int numTweak() const { return (int)((packed >> 16) & 7u); }
    __device__ __forceinline__ int blk() const { return (int)((packed >> 20) & 15u); }
    __device__ __forceinline__ int slot() const { return (int)(packed >> 24); }
};

struct WorkState
{
    float err;
    int mode;
    int partOrIS; // partition, or index selector for modes 4/5 (union in the reference, BC67.cpp:67-75)
    int rotation;
    u32 ep[3][2];
    u32 idxLo, idxHi;   // primary indexes, 4 bits per pixel
    u32 idx2Lo, idx2Hi; // secondary indexes (modes 4/5)
};

// Fix-ups + bit packing of a mode 4 / mode 5 block (reference BC67.cpp:2003-2203) by one lane: with the mode a template
// parameter every field sits at a constant bit position, so the 66 fields cost a shift-or each instead of the generic
// packer's per-lane field arithmetic (the kernel keeps that one for the other modes).
template <int MODE>
__device__ __forceinline__ void packDualPlane(const WorkState &work, u32 &w0, u32 &w1, u32 &w2, u32 &w3)
{
    constexpr int rgbBits = (MODE == 4) ? 5 : 7, alphaBits = (MODE == 4) ? 6 : 8, alphaIndexBits = (MODE == 4) ? 3 : 2;
    constexpr u32 idxOnes = 0x33333333u, idx2Ones = (MODE == 4) ? 0x77777777u : 0x33333333u;
    u32 iLo = work.idxLo, iHi = work.idxHi, jLo = work.idx2Lo, jHi = work.idx2Hi;
    // the anchor (pixel 0) of either index set must have its top bit clear: complement the set and exchange the
    // endpoints it interpolates (every 4-bit slot holds at most the set's maximum, so the complement is an XOR)
    bool flipRGB = ((iLo >> 1) & 1u) != 0;
    bool flipAlpha = ((jLo >> (alphaIndexBits - 1)) & 1u) != 0;
    if (flipRGB)
    {
        iLo ^= idxOnes;
        iHi ^= idxOnes;
    }
    if (flipAlpha)
    {
        jLo ^= idx2Ones;
        jHi ^= idx2Ones;
    }
    const int indexSelector = (MODE == 4) ? work.partOrIS : 0;
    if (indexSelector)
    {
        const bool t = flipRGB;
        flipRGB = flipAlpha;
        flipAlpha = t;
    }
    u32 e0 = work.ep[0][0], e1 = work.ep[0][1];
    if (flipRGB)
    {
        const u32 a = e0, b = e1;
        e0 = (a & 0xff000000u) | (b & 0x00ffffffu);
        e1 = (b & 0xff000000u) | (a & 0x00ffffffu);
    }
    if (flipAlpha)
    {
        const u32 a = e0, b = e1;
        e0 = (a & 0x00ffffffu) | (b & 0xff000000u);
        e1 = (b & 0x00ffffffu) | (a & 0xff000000u);
    }
    u64 lo = 0, hi = 0;
    auto put = [&](u32 value, int off) {
        const u64 v = (u64)value;
        if (off < 64)
        {
            lo |= v << off;
            if (off > 56)
                hi |= v >> (64 - off); // fields are at most 8 bits wide
        }
        else
            hi |= v << (off - 64);
    };
    put(1u << MODE, 0);
    put((u32)work.rotation, MODE + 1);
    if (MODE == 4)
        put((u32)indexSelector, 7);
    constexpr int epBase = 8;
#pragma unroll
    for (int g = 0; g < 6; g++)
    {
        const u32 e = (g & 1) ? e1 : e0;
        put(((e >> (8 * (g >> 1))) & 0xffu) >> (8 - rgbBits), epBase + g * rgbBits);
    }
    constexpr int alphaBase = epBase + 6 * rgbBits;
    put((e0 >> 24) >> (8 - alphaBits), alphaBase);
    put((e1 >> 24) >> (8 - alphaBits), alphaBase + alphaBits);
    constexpr int idxBase = alphaBase + 2 * alphaBits;
#pragma unroll
    for (int px = 0; px < 16; px++)
        put(((px < 8 ? iLo : iHi) >> (4 * (px & 7))) & 0xfu, idxBase + 2 * px - (px > 0 ? 1 : 0));
    constexpr int idx2Base = idxBase + 31;
#pragma unroll
    for (int px = 0; px < 16; px++)
        put(((px < 8 ? jLo : jHi) >> (4 * (px & 7))) & 0xfu, idx2Base + alphaIndexBits * px - (px > 0 ? 1 : 0));
    w0 = (u32)lo;
    w1 = (u32)(lo >> 32);
    w2 = (u32)hi;
    w3 = (u32)(hi >> 32);
}

// In-place channel rotation of the packed pixels: rotation r > 0 exchanges channel r-1 with
// alpha (reference BC67.cpp:1695-1698).  The exchange is an involution.
__device__ __forceinline__ u32 rotatePixel(u32 pk, int rotation)
{
    if (rotation == 1) return (pk & 0x00ffff00u) | (pk >> 24) | (pk << 24);
    if (rotation == 2) return (pk & 0x00ff00ffu) | ((pk >> 16) & 0x0000ff00u) | ((pk << 16) & 0xff000000u);
    if (rotation == 3) return (pk & 0x0000ffffu) | ((pk >> 8) & 0x00ff0000u) | ((pk << 8) & 0xff000000u);
    return pk;
}

__device__ __forceinline__ void levelTable(int e0, int e1, bool threeBit, u32 &tabLo, u32 &tabHi);

struct DualInv
{
    const u32 *words; // s_raw + block index
    int rowSq[4], rowVs[4];
    u32 minMax;
};

// One (mode, rotation, index selector) configuration of the dual-plane modes 4/5
// (reference BC67.cpp:1725-1940): sub-lane c runs seed point c.  `pix` is already rotated so
// that byte 3 is the separately coded channel; w/wSq/rcpW are rotated the same way.
template <bool FAST>
// `sP`: the block's 16 pixels in LDS, already rotated (byte 3 = the separately coded channel), read four at a time in every
// round instead of living in 16 registers; `minMax` = min | max << 8 of the separately coded channel.
__device__ __forceinline__ void evalDual(const u32 *sP, const DualInv &inv, int mode, int indexSelector, const Unfinished &uRGB,
                                         int numTweak, const float (&rw)[4], const float (&rwSq)[4],
                                         const float (&rrcpW)[4], u32 flags, const CvttDeviceTables *__restrict__ T,
                                         int numRefine, int lane, ShapeBest &bestRGB, ShapeBest &bestA)
{
    const int c = lane & 3;
    int rgbPrec, alphaPrec;
    if (mode == 4)
    {
        rgbPrec = indexSelector ? 3 : 2;
        alphaPrec = indexSelector ? 2 : 3;
    }
    else
        rgbPrec = alphaPrec = 2;
    const int rgbRange = 1 << rgbPrec, alphaRange = 1 << alphaPrec;
    const float rgbMax = (float)(rgbRange - 1), alphaMaxV = (float)(alphaRange - 1);
    const int rgbWR = (65536 + (rgbRange - 1)) / (2 * (rgbRange - 1));
    const int alphaWR = (65536 + (alphaRange - 1)) / (2 * (alphaRange - 1));
    const float rgbRcpMax = T->rcpMaxIndex[rgbPrec], alphaRcpMax = T->rcpMaxIndex[alphaPrec];
    const float wRcp16 = T->rcpTable[16];
    const bool uniformErr = (flags & CVTTMI_FLAG_UNIFORM) != 0;

    bestRGB.err = bestA.err = FLT_MAX;
    bestRGB.ep0 = bestRGB.ep1 = bestRGB.idxLo = bestRGB.idxHi = 0;
    bestA.ep0 = bestA.ep1 = bestA.idxLo = bestA.idxHi = 0;

    const u32 minMax = inv.minMax;
    const int alphaMin = (int)(minMax & 0xffu), alphaMax = (int)(minMax >> 8);

    const int tweak = c;
    if (tweak < numTweak)
    {
        int ep[2][4];
        {
            const float tf0 = T->tweakFactors[rgbPrec - 2][tweak][0];
            const float tf1 = T->tweakFactors[rgbPrec - 2][tweak][1];
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                ep[0][ch] = (int)clampRound(uRGB.base[ch] + uRGB.offset[ch] * tf0, 255.0f);
                ep[1][ch] = (int)clampRound(uRGB.base[ch] + uRGB.offset[ch] * tf1, 255.0f);
            }
            // TweakAlpha (reference BC67.cpp:815-827)
            const float af0 = T->tweakFactors[alphaPrec - 2][tweak][0];
            const float af1 = T->tweakFactors[alphaPrec - 2][tweak][1];
            const float base = (float)alphaMin;
            const float offs = (float)alphaMax - base;
            ep[0][3] = (int)clampRound(base + offs * af0, 255.0f);
            ep[1][3] = (int)clampRound(base + offs * af1, 255.0f);
        }

        for (int refine = 0; refine < numRefine; refine++)
        {
            const bool last = (refine == numRefine - 1);
            // CompressEndpoints4 / 5 (reference BC67.cpp:901-923)
            const int cb = (mode == 4) ? 5 : 7;
#pragma unroll
            for (int j = 0; j < 2; j++)
            {
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    ep[j][ch] = unquantize(quantizeNoP(ep[j][ch], cb), cb);
                if (mode == 4)
                    ep[j][3] = unquantize(quantizeNoP(ep[j][3], 6), 6);
            }

            // IndexSelector<3> (rotated weights) and IndexSelector<1> (weight 1.0)
            float origin[4], axis[4];
            int recBase[4], recDelta[4];
            {
                float epDW[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    origin[ch] = (float)ep[0][ch];
                    epDW[ch] = ((float)ep[1][ch] - origin[ch]) * rw[ch];
                }
                float lenSq = epDW[0] * epDW[0];
                lenSq = lenSq + epDW[1] * epDW[1];
                lenSq = lenSq + epDW[2] * epDW[2];
                lenSq = safeDenom(lenSq);
                const float mvdls = rgbMax / lenSq;
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    axis[ch] = epDW[ch] * rw[ch] * mvdls;
                origin[3] = (float)ep[0][3];
                const float aDW = (float)ep[1][3] - origin[3];
                const float aLen = safeDenom(aDW * aDW);
                axis[3] = aDW * (alphaMaxV / aLen);
#pragma unroll
                for (int ch = 0; ch < 4; ch++)
                {
                    // scaled by 4: the reconstructed value (x >> 6) is then byte 1 of the sum
                    recBase[ch] = ((ep[0][ch] << 6) + 32) << 2;
                    recDelta[ch] = (ep[1][ch] - ep[0][ch]) << 2;
                }
            }
            const v2f org01 = {origin[0], origin[1]}, org23 = {origin[2], origin[3]};
            const v2f ax01 = {axis[0], axis[1]}, ax23 = {axis[2], axis[3]};
            const v2f w01 = {rw[0], rw[1]}, w23 = {rw[2], 1.0f};
            const v2f rcpMax2 = {rgbRcpMax, alphaRcpMax};
            // slow indexing: the 4 / 8 values an index can reconstruct, per channel, as bytes (the fast path's level tables):
            // the three candidates of a pixel are then one v_perm_b32 per channel instead of a weight and a multiply-add each
            u32 tLo[4] = {0, 0, 0, 0}, tHi[4] = {0, 0, 0, 0};
            if (!FAST)
            {
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    levelTable(ep[0][ch], ep[1][ch], rgbPrec == 3, tLo[ch], tHi[ch]);
                levelTable(ep[0][3], ep[1][3], alphaPrec == 3, tLo[3], tHi[3]);
            }

            u32 err[4] = {0, 0, 0, 0};
            float slowRGB = 0.0f, slowA = 0.0f;
            v2f tv01 = {0.0f, 0.0f}, tv23 = {0.0f, 0.0f};
            v2f tt2 = {0.0f, 0.0f}, ts2 = {0.0f, 0.0f}; // {RGB plane, alpha plane}
            u32 rgbLo = 0, rgbHi = 0, aLo = 0, aHi = 0;

            const u32 *sPr = sP; // opaque to the optimiser: the loads stay inside the round (see evalDualFast)
            asm volatile("" : "+v"(sPr));
            // four pixels per trip, the trips not unrolled: unrolled sixteen times, the three probes of every pixel kept so
            // much in flight that this instantiation could not be held in the 128 registers of four waves per SIMD
#pragma unroll 1
            for (int g = 0; g < 4; g++)
            {
            const uint4 L = *reinterpret_cast<const uint4 *>(sPr + 4 * g);
            u32 r4 = 0, a4 = 0; // the group's indexes, 4 bits per pixel
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const u32 pk = k == 0 ? L.x : k == 1 ? L.y : k == 2 ? L.z : L.w;
                const v2f x01 = {byteF(pk, 0), byteF(pk, 1)}, x23 = {byteF(pk, 2), byteF(pk, 3)};
                const v2f p01 = (x01 - org01) * ax01, p23 = (x23 - org23) * ax23;
                float dist = p01.x + p01.y;
                dist = dist + p23.x;
                float fRGB = clampRound(dist, rgbMax);
                float fA = clampRound(p23.y, alphaMaxV);
                int iRGB = (int)fRGB, iA = (int)fA;

                if (FAST)
                {
                    const int wgt = mad24(rgbWR, iRGB, 256) >> 9;
                    err[0] = accumulateChannelError<0>(wgt, recDelta[0], recBase[0], pk, err[0]);
                    err[1] = accumulateChannelError<1>(wgt, recDelta[1], recBase[1], pk, err[1]);
                    err[2] = accumulateChannelError<2>(wgt, recDelta[2], recBase[2], pk, err[2]);
                    const int wa = mad24(alphaWR, iA, 256) >> 9;
                    err[3] = accumulateChannelError<3>(wa, recDelta[3], recBase[3], pk, err[3]);
                }
                else
                {
                    // reference BC67.cpp:1834-1880
                    float eRGB = 0.0f, eA = 0.0f;
                    const int c1R = (iRGB > 1 ? iRGB : 1) - 1, c2R = (iRGB + 1 < rgbRange - 1) ? iRGB + 1 : rgbRange - 1;
                    const int c1A = (iA > 1 ? iA : 1) - 1, c2A = (iA + 1 < alphaRange - 1) ? iA + 1 : alphaRange - 1;
                    const u32 selR = (u32)iRGB | ((u32)c1R << 8) | ((u32)c2R << 16) | 0x0c000000u;
                    const u32 selA = (u32)iA | ((u32)c1A << 8) | ((u32)c2A << 16) | 0x0c000000u;
                    // byte k = the value candidate k reconstructs
                    const u32 rec0 = __builtin_amdgcn_perm(tHi[0], tLo[0], selR), rec1 = __builtin_amdgcn_perm(tHi[1], tLo[1], selR),
                              rec2 = __builtin_amdgcn_perm(tHi[2], tLo[2], selR), rec3 = __builtin_amdgcn_perm(tHi[3], tLo[3], selA);
                    auto probeOne = [&](auto kTag, int candRGB, int candA) {
                        constexpr int K = decltype(kTag)::value;
                        // (rec - px)^2 as floats: byte -> float conversions, a subtract and a multiply are exact on these
                        // integers (|d| <= 255, d^2 < 2^24), so the squares are the numbers the integer form converts -- and
                        // they are plain f32 instructions, which a second wave can issue alongside (SDWA and 24-bit
                        // multiplies cannot)
                        const float d0 = byteF(rec0, K) - x01.x, d1 = byteF(rec1, K) - x01.y, d2 = byteF(rec2, K) - x23.x, d3 = byteF(rec3, K) - x23.y;
                        const float e3[3] = {d0 * d0, d1 * d1, d2 * d2};
                        const float e1 = d3 * d3;
                        float er, ea;
                        if (uniformErr)
                        {
                            er = (e3[0] + e3[1]) + e3[2];
                            ea = e1;
                        }
                        else
                        {
                            er = e3[0] * rwSq[0];
                            er = er + e3[1] * rwSq[1];
                            er = er + e3[2] * rwSq[2];
                            ea = e1 * rwSq[3];
                        }
                        if (K == 0)
                        {
                            eRGB = er;
                            eA = ea;
                        }
                        else
                        {
                            // (the winner is tracked as an integer and converted once; equal errors are equal bits, so the
                            // comparison that moves the index also serves the minimum)
                            const bool bR = er < eRGB, bA = ea < eA;
                            eRGB = bR ? er : eRGB;
                            eA = bA ? ea : eA;
                            iRGB = bR ? candRGB : iRGB;
                            iA = bA ? candA : iA;
                        }
                    };
                    probeOne(std::integral_constant<int, 0>{}, iRGB, iA);
                    probeOne(std::integral_constant<int, 1>{}, c1R, c1A);
                    probeOne(std::integral_constant<int, 2>{}, c2R, c2A);
                    fRGB = (float)iRGB;
                    fA = (float)iA;
                    slowRGB = slowRGB + eRGB;
                    slowA = slowA + eA;
                }

                if (!last)
                {
                    // EndpointRefiner<3> with the rotated weights and EndpointRefiner<1> with weight 1
                    const v2f f2 = {fRGB, fA};
                    const v2f t2 = f2 * rcpMax2;
                    const v2f tt = {t2.x, t2.x};
                    const v2f v01 = x01 * w01, v23 = x23 * w23;
                    tv01 = tv01 + tt * v01;
                    tv23 = tv23 + t2 * v23;
                    tt2 = tt2 + t2 * t2;
                    ts2 = ts2 + t2;
                }
                r4 |= (u32)iRGB << (4 * k);
                a4 |= (u32)iA << (4 * k);
            }
            if (g < 2)
            {
                rgbLo |= r4 << (16 * g);
                aLo |= a4 << (16 * g);
            }
            else
            {
                rgbHi |= r4 << (16 * (g - 2));
                aHi |= a4 << (16 * (g - 2));
            }
            }

            float errorRGB, errorA;
            if (FAST)
            {
                if (uniformErr)
                {
                    errorRGB = (float)(int)(err[0] + err[1] + err[2]);
                    errorA = (float)(int)err[3];
                }
                else
                {
                    errorRGB = (float)(int)err[0] * rwSq[0];
                    errorRGB = errorRGB + (float)(int)err[1] * rwSq[1];
                    errorRGB = errorRGB + (float)(int)err[2] * rwSq[2];
                    errorA = (float)(int)err[3] * rwSq[3];
                }
            }
            else
            {
                errorRGB = slowRGB;
                errorA = slowA;
            }

#ifdef CVTT_BC7_DEBUG
            if (blockIdx.x * 16u + (u32)(lane >> 2) == g_bc7DbgBlock)
            {
                // slot: ((mode-4)*8 + rotationHint*2 + indexSelector) is not known here; the caller's config order is
                // recovered from the write counter kept in element 0
                const int cfgSlot = (int)g_bc7Dbg[0];
                float *d = &g_bc7Dbg[16 + ((cfgSlot * 4 + c) * 3 + refine) * 12];
                d[0] = (float)mode; d[1] = (float)indexSelector; d[2] = (float)c; d[3] = (float)refine;
                d[4] = errorRGB; d[5] = errorA;
                d[6] = (float)ep[0][0]; d[7] = (float)ep[0][1]; d[8] = (float)ep[0][2];
                d[9] = (float)ep[1][0]; d[10] = (float)ep[1][1]; d[11] = (float)ep[1][2];
            }
#endif
            if (errorRGB < bestRGB.err)
            {
                bestRGB.err = errorRGB;
                bestRGB.ep0 = (u32)ep[0][0] | ((u32)ep[0][1] << 8) | ((u32)ep[0][2] << 16);
                bestRGB.ep1 = (u32)ep[1][0] | ((u32)ep[1][1] << 8) | ((u32)ep[1][2] << 16);
                bestRGB.idxLo = rgbLo;
                bestRGB.idxHi = rgbHi;
            }
            if (errorA < bestA.err)
            {
                bestA.err = errorA;
                bestA.ep0 = (u32)ep[0][3] << 24;
                bestA.ep1 = (u32)ep[1][3] << 24;
                bestA.idxLo = aLo;
                bestA.idxHi = aHi;
            }

            if (!last)
            {
                // EndpointRefiner<3> / <1>::GetRefinedEndpointsLDR, 16 contributions each
                // (the sums of the pre-weighted pixels do not depend on the indexes: taken once per block, see evalDualFast)
                const float vs[4] = {__builtin_bit_cast(float, inv.words[inv.rowVs[0]]), __builtin_bit_cast(float, inv.words[inv.rowVs[1]]),
                                     __builtin_bit_cast(float, inv.words[inv.rowVs[2]]), __builtin_bit_cast(float, inv.words[inv.rowVs[3]])};
                {
                    const float tv[4] = {tv01.x, tv01.y, tv23.x, tv23.y};
                    const float ttRGB = tt2.x, tsRGB = ts2.x;
                    float adenom = (ttRGB * 16.0f - tsRGB * tsRGB) * wRcp16;
                    const bool z = (adenom == 0.0f);
                    if (z) adenom = 1.0f;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const float a = (tv[ch] - tsRGB * vs[ch] * wRcp16) / adenom;
                        const float b = (vs[ch] - a * tsRGB) * wRcp16;
                        float p1 = b, p2 = a + b;
                        if (z)
                        {
                            p1 = vs[ch] * wRcp16;
                            p2 = p1;
                        }
                        ep[0][ch] = (int)clampRound(p1 * rrcpW[ch], 255.0f);
                        ep[1][ch] = (int)clampRound(p2 * rrcpW[ch], 255.0f);
                    }
                }
                {
                    const float tv[4] = {tv01.x, tv01.y, tv23.x, tv23.y};
                    const float ttA = tt2.y, tsA = ts2.y;
                    float adenom = (ttA * 16.0f - tsA * tsA) * wRcp16;
                    const bool z = (adenom == 0.0f);
                    if (z) adenom = 1.0f;
                    const float a = (tv[3] - tsA * vs[3] * wRcp16) / adenom;
                    const float b = (vs[3] - a * tsA) * wRcp16;
                    float p1 = b, p2 = a + b;
                    if (z)
                    {
                        p1 = vs[3] * wRcp16;
                        p2 = p1;
                    }
                    ep[0][3] = (int)clampRound(p1, 255.0f);
                    ep[1][3] = (int)clampRound(p2, 255.0f);
                }
            }
        }
    }
    quadArgminBroadcast(bestRGB, lane);
    quadArgminBroadcast(bestA, lane);
}


typedef unsigned short v2us __attribute__((ext_vector_type(2)));

// The 4 or 8 reconstructed values of one channel, ((64 - w) * e0 + w * e1 + 32) >> 6 for the BC7 weights w of a 2- or
// 3-bit index (reference IndexSelector.h:90-100), as bytes: tabLo = levels 0..3, tabHi = levels 4..7 (3-bit only).
// Two levels per instruction in packed 16-bit lanes (every intermediate fits 16 bits).
__device__ __forceinline__ void levelTable(int e0, int e1, bool threeBit, u32 &tabLo, u32 &tabHi)
{
    const u32 base = (u32)((e0 << 6) + 32), delta = (u32)(e1 - e0) & 0xffffu;
    const v2us BB = __builtin_bit_cast(v2us, base | (base << 16)), DD = __builtin_bit_cast(v2us, delta | (delta << 16));
    const v2us six = {6, 6};
    // weight pairs: 2-bit {0, 21}, {43, 64}; 3-bit {0, 9}, {18, 27}, {37, 46}, {55, 64}
    const u32 w01 = threeBit ? (0u | (9u << 16)) : (0u | (21u << 16));
    const u32 w23 = threeBit ? (18u | (27u << 16)) : (43u | (64u << 16));
    const v2us r01 = (__builtin_bit_cast(v2us, w01) * DD + BB) >> six;
    const v2us r23 = (__builtin_bit_cast(v2us, w23) * DD + BB) >> six;
    tabLo = __builtin_amdgcn_perm(__builtin_bit_cast(u32, r23), __builtin_bit_cast(u32, r01), 0x06040200u);
    tabHi = 0;
    if (threeBit)
    {
        const v2us r45 = (__builtin_bit_cast(v2us, 37u | (46u << 16)) * DD + BB) >> six;
        const v2us r67 = (__builtin_bit_cast(v2us, 55u | (64u << 16)) * DD + BB) >> six;
        tabHi = __builtin_amdgcn_perm(__builtin_bit_cast(u32, r67), __builtin_bit_cast(u32, r45), 0x06040200u);
    }
}

// four byte-wide indexes -> four nibbles in the low 16 bits
__device__ __forceinline__ u32 nibblesOf(u32 b4)
{
    const u32 t = b4 | (b4 >> 4);
    return __builtin_amdgcn_perm(0u, t, 0x0c0c0200u);
}

// evalDual for fast indexing, on channel-major pixels: P[ch][g] holds channel ch of pixels 4g..4g+3 (one byte each),
// channels already in rotated order (P[3] = the separately coded channel).  The float index selection and the refiner
// are the reference's operation for operation; the integer error sum(rec - px)^2 is evaluated per channel as
// sum(rec^2) - 2 sum(rec * px) + sum(px^2) with the reconstructed bytes looked up by v_perm_b32 from the level table
// and the sums taken by v_dot4_u32_u8 over four pixels at a time -- integers, so the result is the same number.
// Invariants of the block in rotated channel order (DualInv): sum(px^2) per channel, min / max of the separately coded
// channel, and the refiner's sums of the pre-weighted pixels sum(x * w) -- ContributeUnweightedPW adds the same 16 values
// in the same order whatever the indexes are (EndpointRefiner.h:78-92), so the sum is taken once per block.
// They wait in LDS (rows of 16 words, one word per block of the wave) and are read where they are used -- the pixel loop of
// the search is the place with the fewest registers to spare: `rowSq[ch]` / `rowVs[ch]` = word offset of channel position
// ch's sum(px^2) / refiner sum, `minMax` = min | max << 8 of the separately coded channel.
// `sP`: the block's pixels in LDS, channel-major and already in rotated channel order: four words (channel positions 0..3)
// per group g of four pixels, read again in every round (one 128-bit load per group) instead of living in 16 registers.
__device__ __forceinline__ void evalDualFast(const u32 *sP, const DualInv &inv, int mode, int indexSelector, const Unfinished &uRGB,
                                             int numTweak, const float (&rw)[4], const float (&rwSq)[4],
                                             const float (&rrcpW)[4], u32 flags, const CvttDeviceTables *__restrict__ T,
                                             int numRefine, int lane, ShapeBest &bestRGB, ShapeBest &bestA)
{
    const int c = lane & 3;
    int rgbPrec, alphaPrec;
    if (mode == 4)
    {
        rgbPrec = indexSelector ? 3 : 2;
        alphaPrec = indexSelector ? 2 : 3;
    }
    else
        rgbPrec = alphaPrec = 2;
    const float rgbMax = (float)((1 << rgbPrec) - 1), alphaMaxV = (float)((1 << alphaPrec) - 1);
    const float rgbRcpMax = T->rcpMaxIndex[rgbPrec], alphaRcpMax = T->rcpMaxIndex[alphaPrec];
    const float wRcp16 = T->rcpTable[16];
    const bool uniformErr = (flags & CVTTMI_FLAG_UNIFORM) != 0;

    bestRGB.err = bestA.err = FLT_MAX;
    bestRGB.ep0 = bestRGB.ep1 = bestRGB.idxLo = bestRGB.idxHi = 0;
    bestA.ep0 = bestA.ep1 = bestA.idxLo = bestA.idxHi = 0;

    const int alphaMin = (int)(inv.minMax & 0xffu), alphaMax = (int)(inv.minMax >> 8);

    const int tweak = c;
    if (tweak < numTweak)
    {
        int ep[2][4];
        {
            const float tf0 = T->tweakFactors[rgbPrec - 2][tweak][0];
            const float tf1 = T->tweakFactors[rgbPrec - 2][tweak][1];
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                ep[0][ch] = (int)clampRound(uRGB.base[ch] + uRGB.offset[ch] * tf0, 255.0f);
                ep[1][ch] = (int)clampRound(uRGB.base[ch] + uRGB.offset[ch] * tf1, 255.0f);
            }
            // TweakAlpha (reference BC67.cpp:815-827)
            const float af0 = T->tweakFactors[alphaPrec - 2][tweak][0];
            const float af1 = T->tweakFactors[alphaPrec - 2][tweak][1];
            const float base = (float)alphaMin;
            const float offs = (float)alphaMax - base;
            ep[0][3] = (int)clampRound(base + offs * af0, 255.0f);
            ep[1][3] = (int)clampRound(base + offs * af1, 255.0f);
        }

        for (int refine = 0; refine < numRefine; refine++)
        {
            const bool last = (refine == numRefine - 1);
            // CompressEndpoints4 / 5 (reference BC67.cpp:901-923)
            const int cb = (mode == 4) ? 5 : 7;
#pragma unroll
            for (int j = 0; j < 2; j++)
            {
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    ep[j][ch] = unquantize(quantizeNoP(ep[j][ch], cb), cb);
                if (mode == 4)
                    ep[j][3] = unquantize(quantizeNoP(ep[j][3], 6), 6);
            }

            // IndexSelector<3> (rotated weights) and IndexSelector<1> (weight 1.0)
            float origin[4], axis[4];
            u32 tabLo[4], tabHi[4];
            {
                float epDW[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    origin[ch] = (float)ep[0][ch];
                    epDW[ch] = ((float)ep[1][ch] - origin[ch]) * rw[ch];
                }
                float lenSq = epDW[0] * epDW[0];
                lenSq = lenSq + epDW[1] * epDW[1];
                lenSq = lenSq + epDW[2] * epDW[2];
                lenSq = safeDenom(lenSq);
                const float mvdls = rgbMax / lenSq;
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    axis[ch] = epDW[ch] * rw[ch] * mvdls;
                origin[3] = (float)ep[0][3];
                const float aDW = (float)ep[1][3] - origin[3];
                const float aLen = safeDenom(aDW * aDW);
                axis[3] = aDW * (alphaMaxV / aLen);
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    levelTable(ep[0][ch], ep[1][ch], rgbPrec == 3, tabLo[ch], tabHi[ch]);
                levelTable(ep[0][3], ep[1][3], alphaPrec == 3, tabLo[3], tabHi[3]);
            }
            const v2f org01 = {origin[0], origin[1]}, org23 = {origin[2], origin[3]};
            const v2f ax01 = {axis[0], axis[1]}, ax23 = {axis[2], axis[3]};
            const v2f w01 = {rw[0], rw[1]}, w23 = {rw[2], 1.0f};
            const v2f rcpMax2 = {rgbRcpMax, alphaRcpMax};

            u32 s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
            v2f tv01 = {0.0f, 0.0f}, tv23 = {0.0f, 0.0f};
            v2f tt2 = {0.0f, 0.0f}, ts2 = {0.0f, 0.0f}; // {RGB plane, alpha plane}
            u32 rgbLo = 0, rgbHi = 0, aLo = 0, aHi = 0; // 4 bits per pixel

            // an address the optimiser cannot see through: the loads stay inside the round (hoisted, they would hold 16
            // registers again, and the 64 conversions with them)
            const u32 *sPr = sP;
            asm volatile("" : "+v"(sPr));
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
                u32 iR4 = 0, iA4 = 0;
                const uint4 L = *reinterpret_cast<const uint4 *>(sPr + 4 * g);
                const u32 Pg[4] = {L.x, L.y, L.z, L.w};
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const v2f x01 = {byteF(Pg[0], k), byteF(Pg[1], k)}, x23 = {byteF(Pg[2], k), byteF(Pg[3], k)};
                    const v2f p01 = (x01 - org01) * ax01, p23 = (x23 - org23) * ax23;
                    float dist = p01.x + p01.y;
                    dist = dist + p23.x;
                    const float fRGB = clampRound(dist, rgbMax);
                    const float fA = clampRound(p23.y, alphaMaxV);
                    // the indexes are small integers: v_cvt_pk_u8_f32 converts and files one as byte k in one instruction
                    iR4 = __builtin_amdgcn_cvt_pk_u8_f32(fRGB, (u32)k, iR4);
                    iA4 = __builtin_amdgcn_cvt_pk_u8_f32(fA, (u32)k, iA4);
                    if (!last)
                    {
                        // EndpointRefiner<3> with the rotated weights and EndpointRefiner<1> with weight 1; the sums of the
                        // pre-weighted pixels themselves (vs) do not depend on the indexes: DualInv
                        const v2f f2 = {fRGB, fA};
                        const v2f t2 = f2 * rcpMax2;
                        const v2f tt = {t2.x, t2.x};
                        const v2f v01 = x01 * w01, v23 = x23 * w23;
                        tv01 = tv01 + tt * v01;
                        tv23 = tv23 + t2 * v23;
                        tt2 = tt2 + t2 * t2;
                        ts2 = ts2 + t2;
                    }
                }
                if (g < 2)
                {
                    rgbLo |= nibblesOf(iR4) << (16 * g);
                    aLo |= nibblesOf(iA4) << (16 * g);
                }
                else
                {
                    rgbHi |= nibblesOf(iR4) << (16 * (g - 2));
                    aHi |= nibblesOf(iA4) << (16 * (g - 2));
                }
#pragma unroll
                for (int ch = 0; ch < 4; ch++)
                {
                    const u32 R = __builtin_amdgcn_perm(tabHi[ch], tabLo[ch], ch == 3 ? iA4 : iR4);
                    s1[ch] = __builtin_amdgcn_udot4(R, Pg[ch], s1[ch], false);
                    s2[ch] = __builtin_amdgcn_udot4(R, R, s2[ch], false);
                }
            }

            u32 err[4];
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
                err[ch] = s2[ch] + inv.words[inv.rowSq[ch]] - 2u * s1[ch];
            float errorRGB, errorA;
            if (uniformErr)
            {
                errorRGB = (float)(int)(err[0] + err[1] + err[2]);
                errorA = (float)(int)err[3];
            }
            else
            {
                errorRGB = (float)(int)err[0] * rwSq[0];
                errorRGB = errorRGB + (float)(int)err[1] * rwSq[1];
                errorRGB = errorRGB + (float)(int)err[2] * rwSq[2];
                errorA = (float)(int)err[3] * rwSq[3];
            }

            if (errorRGB < bestRGB.err)
            {
                bestRGB.err = errorRGB;
                bestRGB.ep0 = (u32)ep[0][0] | ((u32)ep[0][1] << 8) | ((u32)ep[0][2] << 16);
                bestRGB.ep1 = (u32)ep[1][0] | ((u32)ep[1][1] << 8) | ((u32)ep[1][2] << 16);
                bestRGB.idxLo = rgbLo;
                bestRGB.idxHi = rgbHi;
            }
            if (errorA < bestA.err)
            {
                bestA.err = errorA;
                bestA.ep0 = (u32)ep[0][3] << 24;
                bestA.ep1 = (u32)ep[1][3] << 24;
                bestA.idxLo = aLo;
                bestA.idxHi = aHi;
            }

            if (!last)
            {
                // EndpointRefiner<3> / <1>::GetRefinedEndpointsLDR, 16 contributions each
                const float tv[4] = {tv01.x, tv01.y, tv23.x, tv23.y};
                const float vs[4] = {__builtin_bit_cast(float, inv.words[inv.rowVs[0]]), __builtin_bit_cast(float, inv.words[inv.rowVs[1]]),
                                     __builtin_bit_cast(float, inv.words[inv.rowVs[2]]), __builtin_bit_cast(float, inv.words[inv.rowVs[3]])};
                {
                    const float ttRGB = tt2.x, tsRGB = ts2.x;
                    float adenom = (ttRGB * 16.0f - tsRGB * tsRGB) * wRcp16;
                    const bool z = (adenom == 0.0f);
                    if (z) adenom = 1.0f;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const float a = (tv[ch] - tsRGB * vs[ch] * wRcp16) / adenom;
                        const float b = (vs[ch] - a * tsRGB) * wRcp16;
                        float p1 = b, p2 = a + b;
                        if (z)
                        {
                            p1 = vs[ch] * wRcp16;
                            p2 = p1;
                        }
                        ep[0][ch] = (int)clampRound(p1 * rrcpW[ch], 255.0f);
                        ep[1][ch] = (int)clampRound(p2 * rrcpW[ch], 255.0f);
                    }
                }
                {
                    const float ttA = tt2.y, tsA = ts2.y;
                    float adenom = (ttA * 16.0f - tsA * tsA) * wRcp16;
                    const bool z = (adenom == 0.0f);
                    if (z) adenom = 1.0f;
                    const float a = (tv[3] - tsA * vs[3] * wRcp16) / adenom;
                    const float b = (vs[3] - a * tsA) * wRcp16;
                    float p1 = b, p2 = a + b;
                    if (z)
                    {
                        p1 = vs[3] * wRcp16;
                        p2 = p1;
                    }
                    ep[0][3] = (int)clampRound(p1, 255.0f);
                    ep[1][3] = (int)clampRound(p2, 255.0f);
                }
            }
        }
    }
    quadArgminBroadcast(bestRGB, lane);
    quadArgminBroadcast(bestA, lane);
}

// =====================================================================================
// Exact branch-and-bound, part 2: cheap bounds for every partition before anything is
// searched (part 1, the bound itself, is shapeErrorLowerBound in cvtt_kernel_common.h).
//
// The weighted pixels of a block are projected on an orthonormal pair (e1, e2) -- the two
// leading principal axes of the whole block, so that little of any subset's residual is lost --
// scaled and rounded to 12-bit integers.  An orthogonal projection never increases distances,
// so the total-least-squares residual of the projected, rounded points (minus the allowance
// for the rounding radius) still bounds the error of every trial from below; in 2-D the
// residual is the smaller eigenvalue of a 2x2 matrix whose entries are exact integers
// accumulated with v_dot2_i32_i16 over a per-lane pixel mask.  Sub-lane c of a quad walks
// partitions c, c+4, ...; the bounds land in LDS (64 partitions x 16 blocks).
// =====================================================================================
typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int dot2(u32 a, u32 b, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b), acc, false);
}

__device__ __forceinline__ constexpr int tri(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }

struct BlockScatter
{
    float S[10];     // weighted scatter matrix of the 16 pixels (lower triangle, row-major)
    float meanW[4];  // weighted centroid
};

// raw sums of the 16 pixels: s[ch] = sum x, p[tri(r, c)] = sum x_r x_c
__device__ __forceinline__ void blockRawSums(const u32 (&pix)[16], int (&s)[4], int (&p)[10])
{
#pragma unroll
    for (int i = 0; i < 4; i++)
        s[i] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++)
        p[i] = 0;
#pragma unroll
    for (int px = 0; px < 16; px++)
    {
        const u32 pk = fetchPixel(pix[px]);
        int x[4];
#pragma unroll
        for (int ch = 0; ch < 4; ch++)
        {
            x[ch] = byteI(pk, ch);
            s[ch] += x[ch];
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c <= r; c++)
                p[tri(r, c)] = mad24(x[r], x[c], p[tri(r, c)]);
    }
}

// the same, by the four lanes of the quad that holds the block: sub-lane c sums pixels c, c+4, c+8, c+12 and the quad adds up
__device__ __forceinline__ void blockRawSumsQuad(const u32 (&pix)[16], int c, int (&s)[4], int (&p)[10])
{
#pragma unroll
    for (int i = 0; i < 4; i++)
        s[i] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++)
        p[i] = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
        const u32 p0 = pix[4 * j], p1 = pix[4 * j + 1], p2 = pix[4 * j + 2], p3 = pix[4 * j + 3];
        const u32 pk = fetchPixel(c == 0 ? p0 : c == 1 ? p1 : c == 2 ? p2 : p3);
        int x[4];
#pragma unroll
        for (int ch = 0; ch < 4; ch++)
        {
            x[ch] = byteI(pk, ch);
            s[ch] += x[ch];
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int cc = 0; cc <= r; cc++)
                p[tri(r, cc)] = mad24(x[r], x[cc], p[tri(r, cc)]);
    }
#pragma unroll
    for (int step = 1; step <= 2; step <<= 1)
    {
#pragma unroll
        for (int i = 0; i < 4; i++)
            s[i] += xorLane(s[i], step);
#pragma unroll
        for (int i = 0; i < 10; i++)
            p[i] += xorLane(p[i], step);
    }
}

__device__ __forceinline__ void scatterFromRaw(const int (&s)[4], const int (&p)[10], const CvttBc7Args &A, BlockScatter &bs)
{
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        bs.meanW[r] = (float)s[r] * A.w[r] * 0.0625f;
#pragma unroll
        for (int c = 0; c <= r; c++)
        {
            // 16*sum(xy) - sum(x)*sum(y): an exact integer below 2^24
            const int v = 16 * p[tri(r, c)] - __mul24(s[r], s[c]);
            // (no product of two weights on its own: the optimiser would keep all ten of them in registers for the whole kernel)
            bs.S[tri(r, c)] = (((float)v * 0.0625f) * A.w[r]) * A.w[c];
        }
    }
}

// Exact branch-and-bound, second tier: the bound of a subset in all of its channels.  The 2-D projection of the first
// tier keeps one of the three (two) residual directions of a subset; on low-variance content with noise in every
// channel that leaves most partitions alive.  Raw integer sums of the member pixels of a subset -> scatter matrix ->
// shapeErrorLowerBound, for the partitions that survived the first tier only.
struct RawSums
{
    int n;
    int s[4];
    int p[10];
};

__device__ __forceinline__ void maskedRawSums(const u32 (&pix)[16], u32 mask, RawSums &r)
{
    r.n = __popc(mask & 0xffffu);
#pragma unroll
    for (int i = 0; i < 4; i++)
        r.s[i] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++)
        r.p[i] = 0;
#pragma unroll
    for (int px = 0; px < 16; px++)
    {
        // a pixel that is not a member counts as (0, 0, 0, 0)
        const u32 pk = fetchPixel(pix[px]) & (u32)__builtin_amdgcn_sbfe(mask, px, 1);
        int x[4];
#pragma unroll
        for (int ch = 0; ch < 4; ch++)
        {
            x[ch] = byteI(pk, ch);
            r.s[ch] += x[ch];
        }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b <= a; b++)
                r.p[tri(a, b)] = mad24(x[a], x[b], r.p[tri(a, b)]);
    }
}

// channel-major copy of a block: P[ch][g] = channel ch of pixels 4g .. 4g+3 (the layout of the dual-plane search)
__device__ __forceinline__ void channelMajor(const u32 (&pix)[16], u32 (&P)[4][4])
{
#pragma unroll
    for (int g = 0; g < 4; g++)
    {
        const u32 a = fetchPixel(pix[4 * g]), b = fetchPixel(pix[4 * g + 1]), cc = fetchPixel(pix[4 * g + 2]), d = fetchPixel(pix[4 * g + 3]);
        const u32 ab02 = __builtin_amdgcn_perm(b, a, 0x06020400u); // a0 b0 a2 b2
        const u32 ab13 = __builtin_amdgcn_perm(b, a, 0x07030501u); // a1 b1 a3 b3
        const u32 cd02 = __builtin_amdgcn_perm(d, cc, 0x06020400u);
        const u32 cd13 = __builtin_amdgcn_perm(d, cc, 0x07030501u);
        P[0][g] = __builtin_amdgcn_perm(cd02, ab02, 0x05040100u); // a0 b0 c0 d0
        P[2][g] = __builtin_amdgcn_perm(cd02, ab02, 0x07060302u); // a2 b2 c2 d2
        P[1][g] = __builtin_amdgcn_perm(cd13, ab13, 0x05040100u);
        P[3][g] = __builtin_amdgcn_perm(cd13, ab13, 0x07060302u);
    }
}

// the same sums from the channel-major copy: four pixels per v_dot4_u32_u8
__device__ __forceinline__ void maskedRawSumsCM(const u32 (&P)[4][4], u32 mask, RawSums &r)
{
    r.n = __popc(mask & 0xffffu);
#pragma unroll
    for (int i = 0; i < 4; i++)
        r.s[i] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++)
        r.p[i] = 0;
#pragma unroll
    for (int g = 0; g < 4; g++)
    {
        // four mask bits -> four 0x00 / 0xff bytes
        const u32 bits = (((mask >> (4 * g)) & 0xfu) * 0x00204081u) & 0x01010101u;
        const u32 sel = (bits << 8) - bits;
        u32 m[4];
#pragma unroll
        for (int ch = 0; ch < 4; ch++)
        {
            m[ch] = P[ch][g] & sel;
            r.s[ch] = (int)__builtin_amdgcn_udot4(m[ch], 0x01010101u, (u32)r.s[ch], false);
        }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b <= a; b++)
                r.p[tri(a, b)] = (int)__builtin_amdgcn_udot4(m[a], P[b][g], (u32)r.p[tri(a, b)], false);
    }
}

__device__ __forceinline__ void rawSumsSub(RawSums &d, const RawSums &a)
{
    d.n -= a.n;
#pragma unroll
    for (int i = 0; i < 4; i++)
        d.s[i] -= a.s[i];
#pragma unroll
    for (int i = 0; i < 10; i++)
        d.p[i] -= a.p[i];
}

// n*sum(xy) - sum(x)*sum(y) is an exact integer below 2^24, so the matrix entries carry the rounding of two float
// products only, which the margins of shapeErrorLowerBound cover
template <int N>
__device__ __forceinline__ float subsetBoundFull(const RawSums &r, const float (&w)[4], float delta)
{
    if (r.n < 2)
        return 0.0f;
    const float n = (float)r.n;
    const float inv = __builtin_amdgcn_rcpf(n);
    Moments<N> m;
#pragma unroll
    for (int a = 0; a < N; a++)
#pragma unroll
        for (int b = 0; b <= a; b++)
        {
            const int v = __mul24(r.n, r.p[tri(a, b)]) - __mul24(r.s[a], r.s[b]);
            m.cov[tri(a, b)] = (((float)v * inv) * w[a]) * w[b]; // see scatterFromRaw
        }
    return shapeErrorLowerBound<N>(m, n, delta);
}

// dominant eigenvector of a symmetric PSD 4x4 (power iteration from the column of the
// largest diagonal entry).  Accuracy only affects how tight the bounds are, never their
// validity: the caller orthonormalises whatever comes back.
__device__ __forceinline__ void topEigenvector(const float (&M)[10], float (&e)[4])
{
    int k = 0;
    float d = M[tri(0, 0)];
#pragma unroll
    for (int i = 1; i < 4; i++)
        if (M[tri(i, i)] > d)
        {
            d = M[tri(i, i)];
            k = i;
        }
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        v[i] = (k == 0) ? M[tri(i, 0)] : (k == 1) ? M[tri(i, 1)] : (k == 2) ? M[tri(i, 2)] : M[tri(i, 3)];
    for (int it = 0; it < CVTT_EIG_ITERS; it++)
    {
        float nv[4];
        float big = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            float a = M[tri(r, 0)] * v[0];
#pragma unroll
            for (int c = 1; c < 4; c++)
                a = __fmaf_rn(M[tri(r, c)], v[c], a);
            nv[r] = a;
            big = fmaxf(big, fabsf(a));
        }
        const float sc = (big > 0.0f) ? __builtin_amdgcn_rcpf(big) : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++)
            v[r] = nv[r] * sc;
    }
    float len = v[0] * v[0];
#pragma unroll
    for (int i = 1; i < 4; i++)
        len = __fmaf_rn(v[i], v[i], len);
    const bool ok = len > 1e-12f && len < 1e12f; // false for 0 and NaN
    const float inv = ok ? __builtin_amdgcn_rsqf(len) : 0.0f;
    e[0] = ok ? v[0] * inv : 1.0f;
#pragma unroll
    for (int i = 1; i < 4; i++)
        e[i] = v[i] * inv;
}

// Second tier, sharpened ("Soundness of the bounds", part 3).  The bound above treats a trial as ANY line plus
// the worst rounding of the reconstructed colours (delta = 0.5 * |w|, as if every rounding pointed at the pixel).  Two facts
// about the trials tighten it, both stated for an arbitrary unit vector d (here: the subset's principal direction from a few
// power iterations -- its accuracy affects only how tight the result is):
//  * Rounding moves a reconstructed colour by at most 0.5 * w_ch per channel, and only the part of that ACROSS the trial's
//    line lowers the error.  For a line along u the largest such part is s(u) = 0.5 * sqrt(|w|^2 - min_sigma (u . (sigma w))^2)
//    (sigma = sign patterns).  With the default channel weights (green and alpha dominate) and a line that follows green, s is
//    0.16 instead of 0.52.
//  * The reconstructed colours of a subset are only 2^indexBits points of the line.  Projected on d the pixels are therefore
//    approximated by that many values: Q_d = the cost of the best clustering of the projections into 2^indexBits groups
//    (sum of squared distances to the group means).  It is computed for the two-bit modes, exactly: the projections are
//    sorted (the groups of an optimal clustering in one dimension are runs) and a dynamic programme places the boundaries.
// A trial whose line u makes the angle asin(beta) with d (u = alpha d + beta v, v _|_ d) has, before rounding,
//    error >= alpha^2 R_d + beta^2 L_d - 2 alpha beta rho + max(0, alpha sqrt(Q_d) - beta sqrt(R_d))^2,
// L_d = d^T S d, R_d = trace(S) - L_d (>= v^T S v), rho = |S d - L_d d| (>= |d^T S v|); the last term is the distance of the
// projections on u from the (cone of) vectors with few distinct values, which differs from that of the projections on d by at
// most beta sqrt(v^T S v).  And |u . (sigma w)| >= alpha m_d - beta sqrt(|w|^2 - m_d^2) with m_d = min_sigma |d . (sigma w)|
// (>= 2 max_ch - sum_ch of |d_ch| w_ch).  [0, 1] is cut into a few intervals of beta; on each, every term is replaced by its worst
// value there, G - 2 s sqrt(n G) is formed as in the first bound, and the minimum over the intervals is the result.  Margins:
// every quantity that comes from a cancellation is moved by 1e-5 * trace(S) in the safe direction (the float error of S, of d's
// length and of the products is below 2e-6 of it), the projections by 2e-3, the result is scaled by 0.9999.
// (lp: the block's sixteen pixels in LDS.  They are read twice, for the sums and for the projections, rather than kept in
// registers in between: this function is the register peak of the bounds.)
template <int N>
__device__ __forceinline__ float subsetBoundSharp(const u32 *lp, u32 mask, const float (&w)[4], float delta, float wSq, bool fourLevels)
{
    RawSums r;
    {
        asm volatile("" : "+v"(lp));
        u32 px[16], CM[4][4];
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const uint4 v = *reinterpret_cast<const uint4 *>(lp + 4 * i);
            px[4 * i + 0] = v.x;
            px[4 * i + 1] = v.y;
            px[4 * i + 2] = v.z;
            px[4 * i + 3] = v.w;
        }
        channelMajor(px, CM);
        maskedRawSumsCM(CM, mask, r);
    }
    if (r.n < 2)
        return 0.0f;
    const float n = (float)r.n;
    const float inv = __builtin_amdgcn_rcpf(n);
    Moments<N> m;
#pragma unroll
    for (int a = 0; a < N; a++)
#pragma unroll
        for (int b = 0; b <= a; b++)
        {
            const int v = __mul24(r.n, r.p[tri(a, b)]) - __mul24(r.s[a], r.s[b]);
            m.cov[tri(a, b)] = (((float)v * inv) * w[a]) * w[b]; // see scatterFromRaw
        }
    float rTrue;
    const float cur = shapeErrorLowerBound<N>(m, n, delta, &rTrue);
    if (r.n < 3 || !(rTrue >= 0.0f))
        return cur;
    float T = 0.0f;
#pragma unroll
    for (int i = 0; i < N; i++)
        T += m.cov[tri(i, i)];
    float e[4];
    {
        float M[10];
#pragma unroll
        for (int i = 0; i < 10; i++)
            M[i] = (N == 4 || i < 6) ? m.cov[i < N * (N + 1) / 2 ? i : 0] : 0.0f;
        topEigenvector(M, e);
    }
    float L = 0.0f, Sd[N];
#pragma unroll
    for (int a = 0; a < N; a++)
    {
        float acc = 0.0f;
#pragma unroll
        for (int b = 0; b < N; b++)
            acc = __fmaf_rn(m.cov[a >= b ? tri(a, b) : tri(b, a)], e[b], acc);
        Sd[a] = acc;
        L = __fmaf_rn(e[a], acc, L);
    }
    float rho2 = 0.0f, aMax = 0.0f, aSum = 0.0f;
#pragma unroll
    for (int a = 0; a < N; a++)
    {
        const float t = Sd[a] - L * e[a];
        rho2 = __fmaf_rn(t, t, rho2);
        const float ac = fabsf(e[a]) * w[a];
        aMax = fmaxf(aMax, ac);
        aSum += ac;
    }
    const float eta = 1e-5f * T;
    const float Llo = fmaxf(L - eta, 0.0f);
    const float Rlo = fmaxf(T - L - eta, 0.0f);
    const float sqrtRhi = __builtin_amdgcn_sqrtf(fmaxf(T - L, 0.0f) + eta) * 1.000001f;
    const float rhoHi = __builtin_amdgcn_sqrtf(rho2) * 1.000001f + eta;
    const float W2 = wSq * 1.000001f;
    const float md = fmaxf(2.0f * aMax - aSum, 0.0f) * 0.99999f;
    const float perp = __builtin_amdgcn_sqrtf(fmaxf(W2 - md * md, 0.0f)) * 1.000001f;

    float sqrtQ = 0.0f;
    constexpr int kMaxN = 13; // the largest subset of any BC7 partition (a larger one would simply go without Q_d)
    if (fourLevels && r.n > 4 && r.n <= kMaxN)
    {
        float g[4];
#pragma unroll
        for (int ch = 0; ch < 4; ch++)
            g[ch] = (ch < N) ? e[ch] * w[ch] : 0.0f;
        float t[16];
        asm volatile("" : "+v"(lp));
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            const uint4 v = *reinterpret_cast<const uint4 *>(lp + 4 * q);
            const u32 pw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                float a = 0.0f;
#pragma unroll
                for (int ch = 0; ch < N; ch++)
                    a = __fmaf_rn(g[ch], byteF(pw[k], ch), a);
                t[4 * q + k] = ((mask >> (4 * q + k)) & 1u) ? a : 1e30f;
            }
        }
        // Batcher's merge exchange, 63 comparators: the members come first, in ascending order
#define CE(a, b) { const float lo_ = fminf(t[a], t[b]); t[b] = fmaxf(t[a], t[b]); t[a] = lo_; }
        CE(0,1) CE(2,3) CE(4,5) CE(6,7) CE(8,9) CE(10,11) CE(12,13) CE(14,15) CE(0,2) CE(1,3) CE(4,6) CE(5,7) CE(8,10) CE(9,11) CE(12,14) CE(13,15)
        CE(1,2) CE(5,6) CE(9,10) CE(13,14) CE(0,4) CE(1,5) CE(2,6) CE(3,7) CE(8,12) CE(9,13) CE(10,14) CE(11,15) CE(2,4) CE(3,5) CE(10,12) CE(11,13)
        CE(1,2) CE(3,4) CE(5,6) CE(9,10) CE(11,12) CE(13,14) CE(0,8) CE(1,9) CE(2,10) CE(3,11) CE(4,12) CE(5,13) CE(6,14) CE(7,15) CE(4,8) CE(5,9)
        CE(6,10) CE(7,11) CE(2,4) CE(3,5) CE(6,8) CE(7,9) CE(10,12) CE(11,13) CE(1,2) CE(3,4) CE(5,6) CE(7,8) CE(9,10) CE(11,12) CE(13,14)
#undef CE
        // The exact cost of the best clustering of the sorted projections into four groups, by dynamic programming over the
        // group boundaries: D_k[j] = min_i D_(k-1)[i] + cost(i, j), cost(i, j) = sum t^2 - (sum t)^2 / (j - i) over t[i .. j-1]
        // (empty groups allowed: D_k[0] = 0).  Everything is unrolled over the kMaxN slots a subset can fill, the slots past the
        // members hold zeros after the shift by t[0], and the last group is taken from suffix sums so that no step depends on
        // the (per-lane) member count.
        const int cnt = r.n;
        const float t0 = t[0];
        float range = 0.0f;
#pragma unroll
        for (int j = 0; j < kMaxN; j++)
        {
            t[j] = (j < cnt) ? t[j] - t0 : 0.0f;
            range = fmaxf(range, t[j]);
        }
        constexpr float kRcp[17] = {0.0f, 1.0f, 1.0f / 2, 1.0f / 3, 1.0f / 4, 1.0f / 5, 1.0f / 6, 1.0f / 7, 1.0f / 8, 1.0f / 9, 1.0f / 10, 1.0f / 11, 1.0f / 12,
                                    1.0f / 13, 1.0f / 14, 1.0f / 15, 1.0f / 16};
        float D2[kMaxN + 1], D3[kMaxN + 1];
        D2[0] = D3[0] = 0.0f;
        {
            float su = 0.0f, sq = 0.0f;
#pragma unroll
            for (int j = 1; j <= kMaxN; j++)
            {
                su += t[j - 1];
                sq = __fmaf_rn(t[j - 1], t[j - 1], sq);
                D2[j] = D3[j] = __fmaf_rn(-su * su, kRcp[j], sq); // one group: D_1[j]
            }
        }
        float fin = D3[kMaxN]; // (zeros past the members: an over-estimate of the one-group cost of all of them, harmless in a minimum)
        {
            float su0 = 0.0f, sq0 = 0.0f; // prefix sums: D_1[i]
#pragma unroll
            for (int i = 1; i < kMaxN; i++)
            {
                su0 += t[i - 1];
                sq0 = __fmaf_rn(t[i - 1], t[i - 1], sq0);
                const float d1 = __fmaf_rn(-su0 * su0, kRcp[i], sq0);
                const float d2 = D2[i], d3 = D3[i]; // final: every start below i has been through
                float su = 0.0f, sq = 0.0f;
#pragma unroll
                for (int j = i + 1; j <= kMaxN; j++)
                {
                    su += t[j - 1];
                    sq = __fmaf_rn(t[j - 1], t[j - 1], sq);
                    const float cst = __fmaf_rn(-su * su, kRcp[j - i], sq);
                    D2[j] = fminf(D2[j], d1 + cst);
                    D3[j] = fminf(D3[j], d2 + cst);
                }
                // the fourth group, t[i .. cnt-1]: su / sq now hold the sums of everything from i on (zeros past the members)
                const float m = (float)(cnt - i);
                const float last = __fmaf_rn(-su * su, __builtin_amdgcn_rcpf(m), sq); // (v_rcp_f32: 1 ulp, inside the margin below)
                fin = (i < cnt) ? fminf(fin, d3 + last) : fin;
            }
        }
        // float error of a cost: a few ulps of (count * range^2); four of them and the running minima
        const float Q = fmaxf(fin - 2e-5f * range * range, 0.0f);
        sqrtQ = fmaxf(__builtin_amdgcn_sqrtf(fmaxf(Q * 0.9999f, 0.0f)) - 2e-3f, 0.0f);
    }

    // beta = 0, 0.01, 0.02, 0.03, 0.04, 0.055, 0.07, 0.085, 0.1, 0.125, 0.15, 0.175, 0.2, 0.25, 0.3, 0.375, 0.45, 0.55, 0.7, 0.85, 1: dense
    // where the minimum usually is (a small angle buys a lot of rounding slack when d is close to the dominant channel)
    constexpr int kIntervals = 20;
    constexpr float kB0sq[20] = {0.0f, 0.0001f, 0.0004f, 0.0009f, 0.0016f, 0.003025f, 0.0049f, 0.007225f, 0.01f, 0.015625f, 0.0225f, 0.030625f, 0.04f, 0.0625f, 0.09f, 0.140625f, 0.2025f, 0.3025f, 0.49f, 0.7225f};
    constexpr float kB1sq[20] = {0.0001f, 0.0004f, 0.0009f, 0.0016f, 0.003025f, 0.0049f, 0.007225f, 0.01f, 0.015625f, 0.0225f, 0.030625f, 0.04f, 0.0625f, 0.09f, 0.140625f, 0.2025f, 0.3025f, 0.49f, 0.7225f, 1.0f};
    constexpr float kB1[20] = {0.01f, 0.02f, 0.03f, 0.04f, 0.055f, 0.07f, 0.085f, 0.1f, 0.125f, 0.15f, 0.175f, 0.2f, 0.25f, 0.3f, 0.375f, 0.45f, 0.55f, 0.7f, 0.85f, 1.0f};
    // sqrt(1 - beta1^2), rounded down
    constexpr float kA1[20] = {0.999948f, 0.999798f, 0.9995479f, 0.9991977f, 0.9984844f, 0.997545f, 0.996379f, 0.9949854f, 0.9921548f, 0.988684f, 0.9845665f, 0.9797939f, 0.9682439f, 0.9539373f, 0.927023f, 0.8930268f, 0.835163f, 0.7141414f, 0.5267816f, 0.0f};
    // the largest alpha * beta of the interval, rounded up
    constexpr float kCross[20] = {0.00999952f, 0.01999604f, 0.02998656f, 0.03996807f, 0.05491686f, 0.06982843f, 0.08469255f, 0.09949894f, 0.1240198f, 0.1483032f, 0.1722998f, 0.1959596f, 0.2420619f, 0.2861823f, 0.347635f, 0.4018637f, 0.4593415f, 0.499901f, 0.500001f, 0.4477662f};
    float best = FLT_MAX;
#pragma unroll
    for (int j = 0; j < kIntervals; j++)
    {
        const float e0 = __fmaf_rn(kB0sq[j], Llo - Rlo, Rlo), e1 = __fmaf_rn(kB1sq[j], Llo - Rlo, Rlo);
        float base = fminf(e0, e1) * 0.999999f - 2.0f * kCross[j] * rhoHi;
        base = fmaxf(base, rTrue);
        const float qd = fmaxf(kA1[j] * sqrtQ - kB1[j] * sqrtRhi, 0.0f);
        const float G = __fmaf_rn(qd, qd, base);
        const float inner = fmaxf(kA1[j] * md - kB1[j] * perp, 0.0f);
        const float s2 = 0.25f * (W2 - inner * inner) * 1.000001f;
        float v = 0.0f;
        if (G > 4.0001f * n * s2)
            v = G - 2.0f * __builtin_amdgcn_sqrtf(s2 * n * G) * 1.000001f;
        best = fminf(best, v);
    }
    return fmaxf(cur, best * 0.9999f);
}

__device__ __forceinline__ int dot4s(u32 a, u32 b, int acc) { return __builtin_amdgcn_sdot4((int)a, (int)b, acc, false); }
template <bool G8>
struct Proj2D
{
    // G8: int8 quadruples, pixel 4k + j in byte j; else int16 pairs, pixel 2k in the low half, 2k+1 in the high half
    u32 U[G8 ? 4 : 8], V[G8 ? 4 : 8];
    int tU, tV, tUU, tVV, tUV; // sums over all 16 pixels
};

// Project the block on its two leading principal axes (channels 0..2 only when !use4).
template <bool G8>
__device__ __forceinline__ void makeProjection(const u32 (&pix)[16], const BlockScatter &bs, const CvttBc7Args &A,
                                               bool use4, float scale, Proj2D<G8> &P)
{
    float M[10];
#pragma unroll
    for (int i = 0; i < 10; i++)
        M[i] = (use4 || i < 6) ? bs.S[i] : 0.0f;
    float e1[4], e2[4];
    topEigenvector(M, e1);
    {
        // deflate: M -= lambda e1 e1^T
        float Me[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            float a = M[tri(r, 0)] * e1[0];
#pragma unroll
            for (int c = 1; c < 4; c++)
                a = __fmaf_rn(M[tri(r, c)], e1[c], a);
            Me[r] = a;
        }
        float lam = Me[0] * e1[0];
#pragma unroll
        for (int r = 1; r < 4; r++)
            lam = __fmaf_rn(Me[r], e1[r], lam);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c <= r; c++)
                M[tri(r, c)] = __fmaf_rn(-lam * e1[r], e1[c], M[tri(r, c)]);
    }
    topEigenvector(M, e2);
    {
        // Gram-Schmidt against e1; when nothing is left take the coordinate axis e1 leans on least
        float d = e2[0] * e1[0];
#pragma unroll
        for (int i = 1; i < 4; i++)
            d = __fmaf_rn(e2[i], e1[i], d);
        float len = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            e2[i] = __fmaf_rn(-d, e1[i], e2[i]);
            len = __fmaf_rn(e2[i], e2[i], len);
        }
        if (!(len > 1e-4f))
        {
            int k = 0;
            float small = fabsf(e1[0]);
#pragma unroll
            for (int i = 1; i < 4; i++)
                if ((use4 || i < 3) && fabsf(e1[i]) < small)
                {
                    small = fabsf(e1[i]);
                    k = i;
                }
            const float ek = (k == 0) ? e1[0] : (k == 1) ? e1[1] : (k == 2) ? e1[2] : e1[3];
            len = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                e2[i] = ((i == k) ? 1.0f : 0.0f) - ek * e1[i];
                len = __fmaf_rn(e2[i], e2[i], len);
            }
        }
        const float inv = __builtin_amdgcn_rsqf(len);
#pragma unroll
        for (int i = 0; i < 4; i++)
            e2[i] *= inv;
    }
    if (!use4)
        e1[3] = e2[3] = 0.0f; // already zero up to rounding; keep alpha out exactly

    float g1[4], g2[4];
    float o1 = 0.0f, o2 = 0.0f;
#pragma unroll
    for (int ch = 0; ch < 4; ch++)
    {
        g1[ch] = e1[ch] * A.w[ch] * scale;
        g2[ch] = e2[ch] * A.w[ch] * scale;
        o1 = __fmaf_rn(e1[ch] * scale, bs.meanW[ch], o1);
        o2 = __fmaf_rn(e2[ch] * scale, bs.meanW[ch], o2);
    }
    P.tU = P.tV = P.tUU = P.tVV = P.tUV = 0;
    if constexpr (G8)
    {
    // the four lanes of a quad hold the same block: sub-lane c projects pixels 4c .. 4c+3 and the quad exchanges the words
    int tid = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); // = threadIdx.x (one wave per workgroup), recomputed:
    asm volatile("" : "+v"(tid)); // the optimiser otherwise keeps the prologue's thread id (or `lane & ~3`) alive, spilled, for this
    const int c = tid & 3;
    u32 myU = 0, myV = 0;
    {
        u32 ub = 0, vb = 0;
#pragma unroll
        for (int h = 0; h < 4; h++)
        {
            const u32 p0 = pix[h], p1 = pix[4 + h], p2 = pix[8 + h], p3 = pix[12 + h];
            const u32 pk = fetchPixel(c == 0 ? p0 : c == 1 ? p1 : c == 2 ? p2 : p3);
            float fu = -o1, fv = -o2;
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
            {
                const float x = byteF(pk, ch);
                fu = __fmaf_rn(g1[ch], x, fu);
                fv = __fmaf_rn(g2[ch], x, fv);
            }
            // clamping is a projection on a box: it never increases distances either
            fu = fminf(fmaxf(fu, -kBoundLimit8), kBoundLimit8);
            fv = fminf(fmaxf(fv, -kBoundLimit8), kBoundLimit8);
            ub |= ((u32)(int)rintf(fu) & 0xffu) << (8 * h);
            vb |= ((u32)(int)rintf(fv) & 0xffu) << (8 * h);
        }
        myU = ub;
        myV = vb;
    }
    const int quadBase = tid & ~3;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const u32 ub = (u32)__shfl((int)myU, quadBase + k), vb = (u32)__shfl((int)myV, quadBase + k);
        P.U[k] = ub;
        P.V[k] = vb;
        P.tU = dot4s(ub, 0x01010101u, P.tU);
        P.tV = dot4s(vb, 0x01010101u, P.tV);
        P.tUU = dot4s(ub, ub, P.tUU);
        P.tVV = dot4s(vb, vb, P.tVV);
        P.tUV = dot4s(ub, vb, P.tUV);
    }
    }
    else
    {
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
        int uu[2], vv[2];
#pragma unroll
        for (int h = 0; h < 2; h++)
        {
            const u32 pk = fetchPixel(pix[2 * k + h]);
            float fu = -o1, fv = -o2;
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
            {
                const float x = byteF(pk, ch);
                fu = __fmaf_rn(g1[ch], x, fu);
                fv = __fmaf_rn(g2[ch], x, fv);
            }
            // clamping is a projection on a box: it never increases distances either
            fu = fminf(fmaxf(fu, -kBoundLimit16), kBoundLimit16);
            fv = fminf(fmaxf(fv, -kBoundLimit16), kBoundLimit16);
            uu[h] = (int)rintf(fu);
            vv[h] = (int)rintf(fv);
        }
        P.U[k] = ((u32)uu[0] & 0xffffu) | ((u32)uu[1] << 16);
        P.V[k] = ((u32)vv[0] & 0xffffu) | ((u32)vv[1] << 16);
        P.tU = dot2(P.U[k], 0x00010001u, P.tU);
        P.tV = dot2(P.V[k], 0x00010001u, P.tV);
        P.tUU = dot2(P.U[k], P.U[k], P.tUU);
        P.tVV = dot2(P.V[k], P.V[k], P.tVV);
        P.tUV = dot2(P.U[k], P.V[k], P.tUV);
    }
    }
}

struct Sums2D
{
    int n, u, v, uu, vv, uv;
};

// 8-bit grid, the subset given as the byte masks of CvttDeviceTables::subsetByteMask (one 16-byte load instead of four
// nibble -> byte-mask expansions per subset: those were a third of the instructions of a first-tier bound, a quarter-rate
// v_mul_lo_u32 among them)
__device__ __forceinline__ void maskedSumsSel(const Proj2D<true> &P, int n, const uint4 &sel, Sums2D &m)
{
    m.n = n;
    m.u = m.v = m.uu = m.vv = m.uv = 0;
    const u32 s4[4] = {sel.x, sel.y, sel.z, sel.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const u32 um = P.U[k] & s4[k], vm = P.V[k] & s4[k];
        m.u = dot4s(um, 0x01010101u, m.u);
        m.v = dot4s(vm, 0x01010101u, m.v);
        m.uu = dot4s(um, P.U[k], m.uu);
        m.vv = dot4s(vm, P.V[k], m.vv);
        m.uv = dot4s(um, P.V[k], m.uv);
    }
}

template <bool G8>
__device__ __forceinline__ void maskedSums(const Proj2D<G8> &P, u32 mask, Sums2D &m)
{
    m.n = __popc(mask);
    m.u = m.v = m.uu = m.vv = m.uv = 0;
    if constexpr (G8)
    {
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        // four mask bits -> four 0x00 / 0xff bytes
        const u32 bits = (((mask >> (4 * k)) & 0xfu) * 0x00204081u) & 0x01010101u;
        const u32 sel = (bits << 8) - bits;
        const u32 um = P.U[k] & sel, vm = P.V[k] & sel;
        m.u = dot4s(um, 0x01010101u, m.u);
        m.v = dot4s(vm, 0x01010101u, m.v);
        m.uu = dot4s(um, P.U[k], m.uu);
        m.vv = dot4s(vm, P.V[k], m.vv);
        m.uv = dot4s(um, P.V[k], m.uv);
    }
    }
    else
    {
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
        const u32 lo = (u32)__builtin_amdgcn_sbfe(mask, 2 * k, 1);
        const u32 hi = (u32)__builtin_amdgcn_sbfe(mask, 2 * k + 1, 1);
        const u32 sel = (lo & 0x0000ffffu) | (hi & 0xffff0000u);
        const u32 um = P.U[k] & sel, vm = P.V[k] & sel;
        m.u = dot2(um, 0x00010001u, m.u);
        m.v = dot2(vm, 0x00010001u, m.v);
        m.uu = dot2(um, P.U[k], m.uu);
        m.vv = dot2(vm, P.V[k], m.vv);
        m.uv = dot2(um, P.V[k], m.uv);
    }
    }
}

// Lower bound on the error of any trial of the subset with sums `m`.  |coordinates| <= 2040 and
// n <= 16 keep n*sum(uu) and sum(u)^2 below 2^31, so A, B, C are exact; the float steps that
// follow are protected by relative margins on the large terms.
template <bool G8 = false>
__device__ __forceinline__ float subsetBound2D(const Sums2D &m, float invScaleSq, float delta)
{
    if (m.n < 2)
        return 0.0f;
    // |sum u|, |sum v| <= 16 * 2040 fit 24 bits: v_mul_i32_i24 is a full-rate instruction, v_mul_lo_u32 is not; on the 8-bit
    // grid the sums of squares (<= 16 * 128^2) fit as well
    const int a = (G8 ? __mul24(m.n, m.uu) : m.n * m.uu) - __mul24(m.u, m.u);
    const int c = (G8 ? __mul24(m.n, m.vv) : m.n * m.vv) - __mul24(m.v, m.v);
    const int b = (G8 ? __mul24(m.n, m.uv) : m.n * m.uv) - __mul24(m.u, m.v);
    const float fa = (float)a, fc = (float)c, fb = (float)b;
    const float half = (fa + fc) * 0.5f;
    const float diff = (fa - fc) * 0.5f;
    // v_sqrt_f32 / v_rcp_f32 are accurate to 1 ulp; the margins below cover that
    const float rad = __builtin_amdgcn_sqrtf(__fmaf_rn(diff, diff, fb * fb));
    const float lam = half * 0.999998f - rad * 1.000002f; // n * scale^2 * (smaller eigenvalue), rounded down
    const float n = (float)m.n;
    const float L = lam * invScaleSq;                      // n * R
    float lb = 0.0f;
    if (L > n * n * delta * delta)
        lb = (L * __builtin_amdgcn_rcpf(n) - 2.0f * delta * __builtin_amdgcn_sqrtf(L)) * 0.9999f;
    return lb > 0.0f ? lb : 0.0f;
}

} // namespace

// bit k of the result = bit 4k + c of m: the partitions sub-lane c of a quad looks after
__device__ __forceinline__ u32 everyFourth(u64 m, int c)
{
    u64 x = (m >> c) & 0x1111111111111111ull;
    x = (x | (x >> 3)) & 0x0303030303030303ull;
    x = (x | (x >> 6)) & 0x000f000f000f000full;
    x = (x | (x >> 12)) & 0x000000ff000000ffull;
    x = (x | (x >> 24)) & 0xffffull;
    return (u32)x;
}

// HARD = false: the encoder.  A block whose mode-7 stage starts with at least A.hardMin partitions alive hands them to
// the second launch (A.hardCap slots; none left: it searches them itself) and records its best without them.
// HARD = true: the second launch.  kHardWaves (16) wavefronts per recorded block; each reloads the block's original wave
// (the group-wide booleans need its neighbours), searches one slice of the 64 mode-7 partitions against the recorded
// best and leaves its best candidate, packed, in A.hardCand; cvttmi_bc7_hard_commit_kernel picks the winner.
// Every candidate is compared by (error, position in the reference's order), so the split cannot change the result.
template <bool FAST, bool PT, bool HARD>
// (the body of the kernel as a function of its work item: the encoder's launches run it once per workgroup, the HARD launch
// loops a fixed-size grid over its items -- cvttmi_bc7_kernel below)
__device__ __forceinline__ void bc7Body(const uint8_t *__restrict__ blocks, uint8_t *__restrict__ out,
                                        const CvttBc7Args &A, const CvttDeviceTables *__restrict__ T,
                                        const CvttBc7DevicePlan *__restrict__ dplan, const u32 vBlock, const u32 vGrid)
{
    // error lower bound of every partition of the current mode, per block: the upper 16 bits of the binary32 value
    // (truncated, i.e. rounded down -- a bound may always be smaller).  Half the bytes of a float table: with that and 64
    // result slots the kernel fits 8 LDS granules = 16 workgroups per CU = 4 waves per SIMD.
    __shared__ unsigned short s_bound[64][16];
    // A bound is never negative, so its sign bit is free: it marks the entries that hold a second-tier (full-dimension)
    // bound already (lbStoreTier2), and readers take the absolute value (a source modifier, no instruction).
    auto lbLoadRaw = [&](int partition, int b) -> float { return __builtin_bit_cast(float, (u32)s_bound[partition][b] << 16); };
    auto lbLoad = [&](int partition, int b) -> float { return __builtin_fabsf(lbLoadRaw(partition, b)); };
    auto lbStore = [&](int partition, int b, float v) { s_bound[partition][b] = (unsigned short)(__builtin_bit_cast(u32, v) >> 16); };
    auto lbStoreTier2 = [&](int partition, int b, float v) { s_bound[partition][b] = (unsigned short)((__builtin_bit_cast(u32, v) >> 16) | 0x8000u); };
    // the same bytes as 32-bit words, [row][block], rows 1..16: where the dual-plane search parks its per-block invariants
    // (row 0 of the 16-bit table, the mode-6 bound, lies below them)
    u32 *const s_raw = reinterpret_cast<u32 *>(&s_bound[0][0]);
    __shared__ u32 s_pix[16][16];     // the 16 blocks of this wave
    __shared__ u32 s_blkFlags[16];    // wantPCA4 of the block's group
    __shared__ int s_scatter[16][14]; // raw sums (sum x per channel, sum x_r x_c) of every block: the scatter matrix of the bounds and the totals of the second tier come from here
    __shared__ u32 s_item[64];        // items of the current phase of a round: block | partition << 8 (up to 64 probes)
    __shared__ uint8_t s_myItems[PT ? 16 : 1][32]; // punch-through stages only: the items a block offered this round (elsewhere a block's items are consecutive)
    __shared__ UnitRec s_unit[64];    // PCA seeds per (item, subset)
    __shared__ u32 s_parked[16][2];   // the secondary index set of every block's best mode 4 / 5 candidate, out of the registers until the block is packed
    // the pixel bitmaps of the partitions (subset 1 of a two-subset partition; subsets 1 | 2 << 16 of a three-subset one): the
    // search looks them up per lane many times per round, and a per-lane lookup in the HBM tables is a dependent global load
    __shared__ float s_static[16];    // static alpha error of the RGB modes, whole block (BC67.cpp:1250-1264), times 0.9999
    __shared__ unsigned short s_pm2[64];
    __shared__ u32 s_pm3[64];
    // best of every (item, subset) = unit of the round: error, endpoints, indexes.  (During the dual-plane search the same
    // bytes hold the channel-major copy of the blocks, read with 128-bit loads.)
    __shared__ __attribute__((aligned(16))) u32 s_res[64][5];
    // BC7_RespectPunchThrough: the error of every trial (chain x refine round) of every unit of the round
    // [unit][chain][round], 32 x 16 x numRefine floats (punch-through instantiation only; up to kMaxPTRefine rounds)
    __shared__ float s_trialErr[PT ? 32 * 16 * kMaxPTRefine : 1];
    // ... and with more rounds than that (the reference only clamps refineRoundsBC7 from below, BC67.cpp:1044-1045) the
    // table of this wave lives in HBM; it is written and read inside the wave, between barriers
    float *const trialHbm = (PT && A.ptTrial) ? A.ptTrial + (size_t)vBlock * (size_t)(32 * 16) * (size_t)(A.refineRounds < 1 ? 1 : A.refineRounds) : nullptr;
    const cvttmi_bc7_plan *__restrict__ plan = &dplan->plan;
    // (not const: REFRESH_LANE() below hands the optimiser the same values as "new" ones at the start of every section of the
    // single-plane search, so that the LDS addresses and masks it derives from them are computed where they are used instead
    // of being hoisted in front of the stage loop and kept -- i.e. spilled -- across the chain rounds)
    int lane = threadIdx.x;
    int c = lane & 3;
    int blk = lane >> 2;
// (the lane number comes from v_mbcnt -- a workgroup is one wave -- so that not even threadIdx.x has to stay alive)
#define REFRESH_LANE() do { lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); asm volatile("" : "+v"(lane)); c = lane & 3; blk = lane >> 2; } while (0)
    u32 hardIndex = 0, hardBlock = 0;
    u64 hardMine = 0;
    float hardErr = FLT_MAX;
    int hardSeq = -1;
    if (HARD)
    {
        hardIndex = vBlock / kHardWaves;
        const u32 count = *A.hardCount;
        if (hardIndex >= (count < A.hardCap ? count : A.hardCap))
            return;
        const CvttBc7HardRec rec = A.hardRec[hardIndex];
        hardBlock = rec.blockIndex;
        hardErr = rec.err;
        hardSeq = rec.seq;
        // this wavefront's share: every kHardWaves-th of the partitions that were alive when the block was handed over
        u64 m = ((u64)rec.aliveHi << 32) | rec.aliveLo;
        const int q = (int)(vBlock % kHardWaves);
        for (int r = 0; m != 0; r++)
        {
            const u64 low = m & (0ull - m);
            if ((r % kHardWaves) == q)
                hardMine |= low;
            m ^= low;
        }
        if (hardMine == 0)
        {
            if (lane == 0)
                A.hardCand[vBlock].err = FLT_MAX;
            return;
        }
    }
    const u32 blockIndex = (HARD ? (hardBlock & ~15u) : vBlock * 16u) + (u32)(lane >> 2);
    const bool inRange = blockIndex < A.numBlocks;
    // The per-lane predicates of the search live as bits of one register (`lf`) rather than as one lane mask each: the
    // scalar registers a mask takes (two each, for the whole kernel) are what the register allocator runs out of first.
    enum : u32 { LF_VALID = 1u, LF_ANY_ALPHA = 2u, LF_ALLOW_RGB = 4u, LF_ALLOW_M7 = 8u, LF_NONMAX_ALPHA = 16u };
    // HARD: the neighbours are loaded for the group-wide booleans only
    u32 lf = (inRange && (!HARD || blockIndex == hardBlock)) ? LF_VALID : 0u;
#define valid ((lf & LF_VALID) != 0)
#define anyBlockHasAlpha ((lf & LF_ANY_ALPHA) != 0)
#define allowRGBModes ((lf & LF_ALLOW_RGB) != 0)
#define allowMode7 ((lf & LF_ALLOW_M7) != 0)
#define blockHasNonMaxAlpha ((lf & LF_NONMAX_ALPHA) != 0)

    PROF_DECL
    s_pm2[lane] = T->partition2[lane];
    s_pm3[lane] = (u32)T->subsetMask3[lane][0] | ((u32)T->subsetMask3[lane][1] << 16);
    __syncthreads();
    u32 pix[16];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(blocks + (size_t)(inRange ? blockIndex : 0u) * 64u);
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const uint4 v = src[i];
            pix[4 * i + 0] = v.x;
            pix[4 * i + 1] = v.y;
            pix[4 * i + 2] = v.z;
            pix[4 * i + 3] = v.w;
        }
        if (c == 0)
        {
#pragma unroll
            for (int i = 0; i < 4; i++)
                *reinterpret_cast<uint4 *>(&s_pix[lane >> 2][4 * i]) = make_uint4(pix[4 * i], pix[4 * i + 1], pix[4 * i + 2], pix[4 * i + 3]);
        }
    }

    REFRESH_LANE();
    // ---- per-block alpha statistics and the two group-wide booleans (BC67.cpp:1054-1078) ----
    int minAlpha = 255;
#pragma unroll
    for (int px = 0; px < 16; px++)
    {
        const int a = byteI(pix[px], 3);
        minAlpha = a < minAlpha ? a : minAlpha;
    }
    lf |= (minAlpha < 255) ? LF_NONMAX_ALPHA : 0u;
    const u64 ballotA = __ballot(minAlpha < 255);
    const u64 ballotR = __ballot(250 < minAlpha);
    const u32 groupA = (lane < 32) ? (u32)ballotA : (u32)(ballotA >> 32);
    const u32 groupR = (lane < 32) ? (u32)ballotR : (u32)(ballotR >> 32);
    lf |= (groupA != 0) ? LF_ANY_ALPHA : 0u;
    lf |= (groupR != 0) ? LF_ALLOW_RGB : 0u;
    const u64 mode7RGB = plan->mode7RGBPartitionEnabled;
    lf |= (groupA != 0 || mode7RGB != 0) ? LF_ALLOW_M7 : 0u;
    // RGBA seeds: PCA over 4 channels when the group has alpha or no RGB modes, otherwise the
    // RGB seeds extended with alpha = 255 (reference BC67.cpp:1113-1144)
    const bool wantPCA4 = anyBlockHasAlpha || !allowRGBModes;
    if (c == 0)
    {
        u32 f = wantPCA4 ? 1u : 0u;
        if (PT)
        {
            // BC67.cpp:1054-1067: punch-through = every alpha is 0 or 255
            bool isPunchThrough = true;
            int maxAlpha = 0;
#pragma unroll
            for (int px = 0; px < 16; px++)
            {
                const int a = byteI(pix[px], 3);
                isPunchThrough = isPunchThrough && (a == 0 || a == 255);
                maxAlpha = a > maxAlpha ? a : maxAlpha;
            }
            f |= (isPunchThrough ? 2u : 0u) | (maxAlpha > 0 ? 4u : 0u) | (blockHasNonMaxAlpha ? 8u : 0u);
        }
        s_blkFlags[lane >> 2] = f;
    }

    int numRefine = A.refineRounds;
    if (numRefine < 1)
        numRefine = 1;

    // Running best.  The reference walks candidates in a fixed order (single-plane modes
    // 0,1,2,3,6,7 by partition, then mode 4 / mode 5 by rotation and index selector) and
    // commits on a strict '<', i.e. it keeps the FIRST candidate that reaches the minimum.
    // We evaluate in a different order, so every candidate carries its position `seq` in the
    // reference's order and the commit compares (error, seq) lexicographically.
    // What is carried through the search is small: the error, `seq` -- which also names the mode, the partition or index
    // selector and the rotation (decoded when the block is packed) -- and the payload of the best candidate (three endpoint
    // pairs, 64 index bits), of which every lane of the quad keeps a quarter: sub-lane s < 3 the endpoints of subset s,
    // sub-lane 3 the indexes.  All four lanes take the same commit decisions, so the quarters always belong together.
    struct { float err; } work;
    work.err = HARD ? hardErr : FLT_MAX;
    int workSeq = HARD ? hardSeq : -1; // nothing committed yet: a candidate must beat FLT_MAX strictly
    u32 workPay0 = 0, workPay1 = 0;

    REFRESH_LANE();
    // ------------- whole-block scatter matrix: bounds for mode 6 and the four rotations -------------
    const bool prune = !HARD && A.prune != 0; // HARD: the bounds have been applied by the first launch
    BlockScatter bs;
    float lbRotMine = 0.0f; // sub-lane r: the bound of rotation r
    auto lbRotOf = [&](int rotation) -> float { return __shfl(lbRotMine, (lane & ~3) | rotation); };
    int rotOrder[4] = {0, 1, 2, 3};
    if (prune)
    {
        {
            int rs[4], rp[10];
            blockRawSumsQuad(pix, c, rs, rp);
            if (c == 0)
            {
#pragma unroll
                for (int i = 0; i < 10; i++)
                    s_scatter[lane >> 2][i] = rp[i];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    s_scatter[lane >> 2][10 + i] = rs[i];
            }
            scatterFromRaw(rs, rp, A, bs);
        }
        if (!HARD)
        {
            Moments<4> m;
#pragma unroll
            for (int i = 0; i < 10; i++)
                m.cov[i] = bs.S[i];
            const float lbMode6 = shapeErrorLowerBound<4>(m, 16.0f, A.delta4);
            // mode 6 is the first single-plane stage: its bound waits where that stage looks for it
            if (c == 0)
                lbStore(0, lane >> 2, lbMode6);
        }
        // rotation r codes channel r-1 (alpha for r = 0) on its own; the other three share a line
        if (!HARD)
        {
            const float s012 = A.wSqSum3;
            const float d0 = A.delta3;
            const float d1 = 0.5000005f * __builtin_amdgcn_sqrtf(s012 - A.wSq[0] + A.wSq[3]) * 1.000001f;
            const float d2 = 0.5000005f * __builtin_amdgcn_sqrtf(s012 - A.wSq[1] + A.wSq[3]) * 1.000001f;
            const float d3 = 0.5000005f * __builtin_amdgcn_sqrtf(s012 - A.wSq[2] + A.wSq[3]) * 1.000001f;
            // sub-lane r bounds rotation r (its three channels picked by selects, so that the quad runs the bound once)
            Moments<3> m;
#define pick(i0, i1, i2, i3) (c == 0 ? bs.S[i0] : c == 1 ? bs.S[i1] : c == 2 ? bs.S[i2] : bs.S[i3]) /* literal indexes: bs.S stays in registers */
            // channel triples (a < b < c'): rotation 0: 0 1 2, 1: 1 2 3, 2: 0 2 3, 3: 0 1 3
            m.cov[0] = pick(tri(0, 0), tri(1, 1), tri(0, 0), tri(0, 0)); // (a, a)
            m.cov[1] = pick(tri(1, 0), tri(2, 1), tri(2, 0), tri(1, 0)); // (b, a)
            m.cov[2] = pick(tri(1, 1), tri(2, 2), tri(2, 2), tri(1, 1)); // (b, b)
            m.cov[3] = pick(tri(2, 0), tri(3, 1), tri(3, 0), tri(3, 0)); // (c', a)
            m.cov[4] = pick(tri(2, 1), tri(3, 2), tri(3, 2), tri(3, 1)); // (c', b)
            m.cov[5] = pick(tri(2, 2), tri(3, 3), tri(3, 3), tri(3, 3)); // (c', c')
#undef pick
            lbRotMine = shapeErrorLowerBound<3>(m, 16.0f, c == 0 ? d0 : c == 1 ? d1 : c == 2 ? d2 : d3);
        }
        // search the rotations in the order of their bounds summed over the wave: the likely
        // winner first, so that the others meet a tight best error
        float tot[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < 4 && !HARD; r++)
        {
            float t = valid ? lbRotOf(r) : 0.0f;
#pragma unroll
            for (int step = 1; step < 64; step <<= 1)
                t += xorLane(t, step);
            tot[r] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t)));
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            int rank = 0;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (q != r && (tot[q] < tot[r] || (tot[q] == tot[r] && q < r)))
                    rank++;
#pragma unroll
            for (int slot = 0; slot < 4; slot++)
                if (rank == slot)
                    rotOrder[slot] = r;
        }
    }
    PROF_MARK(6)

    // ================================ dual-plane modes 4,5 ================================
    // reference TryDualPlane, BC67.cpp:1678-1963.  The RGB seeds depend only on the rotation,
    // so sub-lane r computes them for rotation r once (the reference recomputes them for each
    // mode / index selector).
    REFRESH_LANE();
    if constexpr (!HARD)
    {
        Unfinished uRot;
        {
            u32 rpix[16];
#pragma unroll
            for (int px = 0; px < 16; px++)
                rpix[px] = rotatePixel(pix[px], c);
            float lw[4];
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
                lw[ch] = A.w[ch];
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                if (c == ch + 1)
                {
                    lw[ch] = A.w[3];
                    lw[3] = A.w[ch];
                }
            pcaEndpoints<3>(rpix, 0xffffu, lw, -1, uRot);
        }
        // the seeds of rotation r wait in s_unit[quad | r] (idle until the single-plane stages): a step reads the ones of its
        // rotation from there instead of every lane keeping its own through all twelve steps
#pragma unroll
        for (int ch = 0; ch < 4; ch++)
        {
            s_unit[lane].base[ch] = uRot.base[ch];
            s_unit[lane].offset[ch] = uRot.offset[ch];
        }
        PROF_MARK(0)

        // fast indexing: the search reads a channel-major copy of the block -- per group g of four pixels one word per channel
        // position, in the channel order of the rotation being searched -- from s_P (= s_res, idle until the single-plane
        // stages; 20 words per block so that the 128-bit loads of a wave spread over the banks).  Sub-lane g builds group g
        // from the pixel-major copy whenever the rotation changes (at most four times per wave).
        u32 *const s_P = &s_res[0][0] + (lane >> 2) * 20;
        auto writeRotated = [&](int rotation) {
            const uint4 q = *reinterpret_cast<const uint4 *>(&s_pix[lane >> 2][4 * c]);
            const u32 ab02 = __builtin_amdgcn_perm(q.y, q.x, 0x06020400u); // a0 b0 a2 b2
            const u32 ab13 = __builtin_amdgcn_perm(q.y, q.x, 0x07030501u); // a1 b1 a3 b3
            const u32 cd02 = __builtin_amdgcn_perm(q.w, q.z, 0x06020400u);
            const u32 cd13 = __builtin_amdgcn_perm(q.w, q.z, 0x07030501u);
            u32 w0 = __builtin_amdgcn_perm(cd02, ab02, 0x05040100u); // a0 b0 c0 d0
            u32 w2 = __builtin_amdgcn_perm(cd02, ab02, 0x07060302u); // a2 b2 c2 d2
            u32 w1 = __builtin_amdgcn_perm(cd13, ab13, 0x05040100u);
            u32 w3 = __builtin_amdgcn_perm(cd13, ab13, 0x07060302u);
            // rotation r > 0 exchanges channel r - 1 with alpha (reference BC67.cpp:1695-1698)
            if (rotation == 1) { const u32 t = w0; w0 = w3; w3 = t; }
            if (rotation == 2) { const u32 t = w1; w1 = w3; w3 = t; }
            if (rotation == 3) { const u32 t = w2; w2 = w3; w3 = t; }
            *reinterpret_cast<uint4 *>(s_P + 4 * c) = make_uint4(w0, w1, w2, w3);
        };
        // what every configuration of a rotation would otherwise recompute (DualInv), per original channel: sub-lane ch
        // takes channel ch and parks the five values in rows 1..16 of s_raw (= s_bound), which is idle until the partition bounds
        // (row 0 of the 16-bit table holds the mode-6 bound); a step reads the rows its rotation needs
        {
            __syncthreads(); // s_pix is complete
            writeRotated(0);
            __syncthreads();
            u32 mine[4];
#pragma unroll
            for (int g = 0; g < 4; g++)
                mine[g] = s_P[4 * g + c];
            const float wc = (c == 0) ? A.w[0] : (c == 1) ? A.w[1] : (c == 2) ? A.w[2] : A.w[3];
            u32 sq = 0, su = 0;
            int mn = 255, mx = 0;
            float sw = 0.0f; // sum of x * w in pixel order: the refiner's m_v
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
                sq = __builtin_amdgcn_udot4(mine[g], mine[g], sq, false);
                su = __builtin_amdgcn_udot4(mine[g], 0x01010101u, su, false);
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const int a = byteI(mine[g], k);
                    mn = a < mn ? a : mn;
                    mx = a > mx ? a : mx;
                    sw = sw + byteF(mine[g], k) * wc;
                }
            }
            s_raw[(1 + c) * 16 + (lane >> 2)] = sq;
            s_raw[(5 + c) * 16 + (lane >> 2)] = (u32)mn | ((u32)mx << 8);
            s_raw[(9 + c) * 16 + (lane >> 2)] = __builtin_bit_cast(u32, sw);
            s_raw[(13 + c) * 16 + (lane >> 2)] = __builtin_bit_cast(u32, (float)(int)su); // exact
        }
        __syncthreads();
        // slow indexing reads the pixels pixel-major, rotated, from the same LDS bytes (written before its first step)
        auto writeRotatedPixels = [&](int rotation) {
            const uint4 q = *reinterpret_cast<const uint4 *>(&s_pix[lane >> 2][4 * c]);
            *reinterpret_cast<uint4 *>(s_P + 4 * c) = make_uint4(rotatePixel(q.x, rotation), rotatePixel(q.y, rotation),
                                                                 rotatePixel(q.z, rotation), rotatePixel(q.w, rotation));
        };
        int curRotation = FAST ? 0 : -1;
#ifdef CVTT_BC7_PROFILE
        float simKey[12], simErr[12];
        int simN = 0;
#endif
        for (int step = 0; step < 12; step++)
        {
            REFRESH_LANE();
            const int slot = step / 3, which = step - slot * 3;
            const int rotation = (slot == 0) ? rotOrder[0] : (slot == 1) ? rotOrder[1] : (slot == 2) ? rotOrder[2] : rotOrder[3];
            // position in the reference's commit order: mode 4 (rot 0: is 0,1; rot 1: ...), then mode 5 rot 0..3
            const int mode = which < 2 ? 4 : 5;
            const int indexSelector = which < 2 ? which : 0;
            const int cfg = which < 2 ? rotation * 2 + which : 8 + rotation;
            int numTweak = (mode == 4) ? plan->mode4SP[rotation][indexSelector] : plan->mode5SP[rotation];
            if (numTweak <= 0)
                continue;
            if (numTweak > 4)
                numTweak = 4;
            if (prune)
            {
                // the second plane costs >= 0, the first at least the bound of its three channels
                const float lb = lbRotOf(rotation);
                if (__ballot(valid && !(lb > work.err)) == 0)
                    continue;
            }

            if (rotation != curRotation)
            {
                if (FAST)
                {
                    __syncthreads(); // everybody is done with the previous rotation's copy
                    writeRotated(rotation);
                    __syncthreads();
                }
                else
                {
                    __syncthreads();
                    writeRotatedPixels(rotation);
                    __syncthreads();
                }
                curRotation = rotation;
            }
            float rw[4], rwSq[4], rrcpW[4];
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
            {
                rw[ch] = A.w[ch];
                rwSq[ch] = A.wSq[ch];
                rrcpW[ch] = A.rcpW[ch];
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                if (rotation == ch + 1)
                {
                    rw[ch] = A.w[3];
                    rwSq[ch] = A.wSq[3];
                    rrcpW[ch] = A.rcpW[3];
                    rw[3] = A.w[ch];
                    rwSq[3] = A.wSq[ch];
                    rrcpW[3] = A.rcpW[ch];
                }

            PROF_COUNT(0, 16)
            PROF_COUNT(1, __popcll(__ballot(valid && c == 0 && !(lbRotOf(rotation) > work.err))))
            Unfinished u;
            {
                const UnitRec &ur = s_unit[(lane & ~3) | rotation];
#pragma unroll
                for (int ch = 0; ch < 4; ch++)
                {
                    u.base[ch] = ur.base[ch];
                    u.offset[ch] = ur.offset[ch];
                }
            }
            ShapeBest b, bA;
            {
                DualInv inv;
                const int sepCh = (rotation == 0) ? 3 : rotation - 1; // the original channel that is coded on its own
                inv.words = s_raw + (lane >> 2);
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    const int orig = (rotation == ch + 1) ? 3 : ch;
                    inv.rowSq[ch] = (1 + orig) * 16;
                    inv.rowVs[ch] = (9 + orig) * 16;
                }
                inv.rowSq[3] = (1 + sepCh) * 16;
                inv.rowVs[3] = (13 + sepCh) * 16;
                inv.minMax = s_raw[(5 + sepCh) * 16 + (lane >> 2)];
                if (FAST)
                    evalDualFast(s_P, inv, mode, indexSelector, u, numTweak, rw, rwSq, rrcpW, A.flags, T, numRefine, lane, b, bA);
                else
                    evalDual<FAST>(s_P, inv, mode, indexSelector, u, numTweak, rw, rwSq, rrcpW, A.flags, T, numRefine, lane, b, bA);
            }

#ifdef CVTT_BC7_DEBUG
            if (vBlock * 16u + (u32)(lane >> 2) == g_bc7DbgBlock && c == 0)
            {
                const int cfgSlot = (int)g_bc7Dbg[0];
                g_bc7Dbg[1 + cfgSlot] = (float)(rotation * 100 + cfg);
                g_bc7Dbg[0] = (float)(cfgSlot + 1);
            }
#endif
            const float combined = b.err + bA.err; // reference BC67.cpp:1942
            const int seq = 384 + cfg;
#ifdef CVTT_BC7_PROFILE
            {
                const float lbp = lbRotOf(rotation);
                const bool needed = valid && c == 0 && !(lbp > work.err);
                PROF_COUNT(6, (needed && (bA.err + lbp > work.err)) ? 1 : 0)
                if (needed && simN < 12)
                {
                    simKey[simN] = lbp + bA.err;
                    simErr[simN] = combined;
                    simN++;
                }
            }
#endif
            if (combined < work.err || (combined == work.err && seq < workSeq))
            {
                work.err = combined;
                workSeq = seq;
                // index selector 1: the 2-bit set is the alpha plane (BC67.cpp:1953-1957)
                const u32 iLo = indexSelector ? bA.idxLo : b.idxLo, iHi = indexSelector ? bA.idxHi : b.idxHi;
                workPay0 = (c == 0) ? (b.ep0 | bA.ep0) : (c == 3) ? iLo : 0u;
                workPay1 = (c == 0) ? (b.ep1 | bA.ep1) : (c == 3) ? iHi : 0u;
                s_parked[lane >> 2][0] = indexSelector ? b.idxLo : bA.idxLo;
                s_parked[lane >> 2][1] = indexSelector ? b.idxHi : bA.idxHi;
            }
        }
#ifdef CVTT_BC7_PROFILE
        {
            // how many colour planes an alpha-plane-first, best-key-first order would have to evaluate
            float best = FLT_MAX;
            int evald = 0;
            u32 left = (1u << simN) - 1u;
            while (left)
            {
                int pick = -1;
                float pk = FLT_MAX;
                for (int i = 0; i < 12; i++)
                    if (((left >> i) & 1u) && (pick < 0 || simKey[i] < pk))
                    {
                        pick = i;
                        pk = simKey[i];
                    }
                if (pk > best)
                    break;
                evald++;
                best = simErr[pick] < best ? simErr[pick] : best;
                left &= ~(1u << pick);
            }
            PROF_COUNT(7, evald)
        }
#endif
        __syncthreads();
        PROF_MARK(1)
    }

    // =========================== single-plane modes 0,1,2,3,6,7 ===========================
    // reference TrySinglePlane, BC67.cpp:1146-1660.  Per mode: the partitions that can still
    // win (bounds in LDS vs the best error so far) form a wave-uniform survivor list, which is
    // consumed in batches of four (partition, subset) items: sub-lane c runs the PCA seed search
    // of item c (per-lane shape mask), then the items are searched one after the other with the
    // shape wave-uniform.
    // pixel bitmap of subset `sub` of partition `partition` of a mode with numSub subsets
    auto subsetMaskOf = [&](int numSub, int partition, int sub) -> u32 {
        if (numSub == 1)
            return 0xffffu;
        if (numSub == 2)
        {
            const u32 m1 = s_pm2[partition];
            return sub ? m1 : (~m1 & 0xffffu);
        }
        const u32 m = s_pm3[partition];
        const u32 m1 = m & 0xffffu, m2 = m >> 16;
        return sub == 1 ? m1 : sub == 2 ? m2 : (~(m1 | m2) & 0xffffu);
    };
    // From here on the pixel-major block is read from LDS where it is needed (the bounds): 16 registers that the chain
    // rounds, which have the fewest to spare, do not have to carry.
    auto pixFromLds = [&](u32 (&px)[16]) {
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const uint4 v = *reinterpret_cast<const uint4 *>(&s_pix[blk][4 * i]);
            px[4 * i + 0] = v.x;
            px[4 * i + 1] = v.y;
            px[4 * i + 2] = v.z;
            px[4 * i + 3] = v.w;
        }
    };
    // Wave-uniform bookkeeping of the bounds, packed into one word of 3-bit fields (six separate ints were six VGPRs to the
    // register allocator, which cannot tell that they are uniform: they were the first to be spilled):
    //   WS_BOUNDS_FOR  which bound set s_bound holds, plus one: 1 = two subsets RGBA, 2 = two subsets RGB, 3 = three subsets RGB,
    //                  4 = mode 6; 0 = none.  (Which of its entries have had the second-tier bound merged in is marked in the
    //                  entries themselves: lbStoreTier2.)
    //   WS_T2_NOPAY    the second tier removed nothing in a stage with this many index bits: not again, unless a mode has fewer
    //   WS_SHARP_NOPAY the same for its second, sharper pass;  WS_STRIKES: and for the sharper bound in the filter of the probes
    //   WS_BOUND_BITS  The second-tier bounds know how many index levels the mode has, and a set of bounds is shared by modes with
    //                  three and with two index bits: what is in the table holds for modes with at most this many index bits
    //                  (4 = first tier only, any mode).  The stage orders run the three-bit mode of a set first; its bounds are
    //                  valid for the two-bit mode that follows, which computes its own, sharper ones for what is still alive then.
    //   WS_SHARP_RAN   The marks in the table say "has the plain second-tier bound" (which knows no levels); that the sharper
    //                  pass has been over the set for a number of index bits is remembered here (the partitions alive in a later
    //                  stage of the same set were alive then, too).
    enum : int { WS_BOUNDS_FOR = 0, WS_BOUND_BITS = 3, WS_SHARP_RAN = 6, WS_SHARP_NOPAY = 9, WS_T2_NOPAY = 12, WS_STRIKES = 15 };
    u32 ws = (4u << WS_BOUND_BITS) | (4u << WS_SHARP_RAN);
#define WS_GET(f) ((int)((ws >> (f)) & 7u))
#define WS_SET(f, v) do { ws = (u32)__builtin_amdgcn_readfirstlane((int)((ws & ~(7u << (f))) | ((u32)(v) << (f)))); } while (0)
    // static alpha error of the RGB modes, whole block (BC67.cpp:1250-1264)
    // (kept in LDS, one float per block: in a register it lives through every chain round for the sake of two additions per
    // stage -- and gets spilled)
    if (c == 0)
        s_static[blk] = 0.0f;
    if (prune)
    {
        u32 acc = 0;
        u32 lpix[16];
        pixFromLds(lpix);
#pragma unroll
        for (int px = 0; px < 16; px++)
        {
            const int d = 255 - byteI(lpix[px], 3);
            acc = (u32)mad24(d, d, (int)acc);
        }
        if (c == 0)
            s_static[blk] = ((A.flags & CVTTMI_FLAG_UNIFORM) ? (float)(int)acc : (float)(int)acc * A.wSq[3]) * 0.9999f;
    }

    for (int stageOrder = 0; stageOrder < 6; stageOrder++)
    {
        // The order of the stages is free (every commit compares (error, position in the reference's order)).  A wave without a
        // single translucent pixel searches modes 1 and 3 before mode 7: there mode 7 -- the same partitions with coarser end
        // points -- almost never wins, and what the other two find first prunes its partitions (smooth opaque gradients: 44
        // -> ... partitions per block alive in mode 7); the three share one set of bounds.
        REFRESH_LANE();
        const int stageIter = (ballotA == 0 && !HARD) ? ((stageOrder == 1) ? 2 : (stageOrder == 2) ? 3 : (stageOrder == 3) ? 1 : stageOrder) : stageOrder;
        ModeDesc md;
        int numSubsets, numPartitions, stage, boundSet;
        switch (stageIter)
        {
        case 0: stage = 4; md = {6, 4, 4, 7, 0}; numSubsets = 1; numPartitions = 1; boundSet = 3; break;
        case 1: stage = 5; md = {7, 2, 4, 5, 6}; numSubsets = 2; numPartitions = 64; boundSet = 0; break;
        case 2: stage = 1; md = {1, 3, 2, 6, 7}; numSubsets = 2; numPartitions = 64; boundSet = 1; break;
        case 3: stage = 3; md = {3, 2, 4, 7, 0}; numSubsets = 2; numPartitions = 64; boundSet = 1; break;
        case 4: stage = 0; md = {0, 3, 4, 4, 5}; numSubsets = 3; numPartitions = 16; boundSet = 2; break;
        default: stage = 2; md = {2, 2, 1, 5, 5}; numSubsets = 3; numPartitions = 64; boundSet = 2; break;
        }
        if (HARD && stageIter != 1)
            continue; // the second launch searches mode-7 partitions only
        // A wave without a single translucent pixel: the RGBA bounds of mode 7 would equal the RGB bounds of modes 1 / 3
        // (no variance in alpha; the colour part of a trial's error is bounded by the three-channel bound whatever alpha
        // does), so mode 7 takes the RGB set and the next stage finds its bounds already there.
        if (stageIter == 1 && ballotA == 0)
            boundSet = 1;
        const int mode = md.mode;
        const bool isRGB = mode < 4;
        // does this mode run for my group?  (wave-uniform skip when it runs for nobody -- before the plan is read: a
        // scalar load from memory per stage that a wave of alpha blocks would wait on four times for nothing)
        const bool laneRuns = isRGB ? allowRGBModes : (mode == 7 ? allowMode7 : true);
        if (__ballot(laneRuns) == 0)
            continue;
        u64 enabled;
        switch (stageIter)
        {
        case 0: enabled = plan->mode6Enabled ? 1 : 0; break;
        case 1: enabled = ~0ull; break; // dead mask in the reference (BC67.cpp:1592-1597)
        case 2: enabled = plan->mode1PartitionEnabled; break;
        case 3: enabled = plan->mode3PartitionEnabled; break;
        case 4: enabled = plan->mode0PartitionEnabled & 0xffffull; break;
        default: enabled = plan->mode2PartitionEnabled; break;
        }
        if (HARD)
            enabled = hardMine; // this wavefront's slice of the mode-7 partitions
        if (enabled == 0)
            continue;

        // ---- bounds of every partition of this mode (shared by the modes with the same subsets/channels) ----
        u32 freshAlive = 0;      // bit k: the bound of partition 4k + c, computed just now, does not exceed the block's best error
        bool freshBounds = false;
        if (prune)
        {
            if (WS_GET(WS_BOUNDS_FOR) == boundSet + 1 && WS_GET(WS_BOUND_BITS) < md.indexBits)
                WS_SET(WS_BOUNDS_FOR, 0); // (no stage order does this: two-bit bounds in the table and a three-bit mode to search)
            if (WS_GET(WS_BOUNDS_FOR) != boundSet + 1)
            {
                WS_SET(WS_BOUND_BITS, 4);
                WS_SET(WS_SHARP_RAN, 4);
                __syncthreads();
                if (boundSet == 3)
                {
                    // written when the block statistics were computed
                }
                else
                {
                    const bool use4 = (boundSet == 0);
                    auto boundsOnGrid = [&](auto g8) {
                    constexpr bool G8 = decltype(g8)::value;
                    const float wsum = A.wSqSum3 + (use4 ? A.wSq[3] : 0.0f);
                    const float scale = (G8 ? kBoundGrid8 : kBoundGrid16) / (255.0f * __builtin_amdgcn_sqrtf(wsum)); // wave-uniform
                    const float invScaleSq = 1.0f / (scale * scale);
                    // rounding the projected points moves each by at most sqrt(2)/2 grid units
                    const float delta = (use4 ? A.delta4 : A.delta3) + 0.7072f / scale;
                    Proj2D<G8> P;
                    PROF_MARK(2)
                    {
                        // take the block statistics back from LDS (they were not kept in registers across the search)
                        BlockScatter bsL;
                        {
                            int rs[4], rp[10];
#pragma unroll
                            for (int i = 0; i < 10; i++)
                                rp[i] = s_scatter[blk][i];
#pragma unroll
                            for (int i = 0; i < 4; i++)
                                rs[i] = s_scatter[blk][10 + i];
                            scatterFromRaw(rs, rp, A, bsL);
                        }
                        u32 lpix[16];
                        pixFromLds(lpix);
                        makeProjection(lpix, bsL, A, use4, scale, P);
                    }
                    PROF_MARK(7)
                    // the byte masks of the next partition are loaded while this one is computed
                    const bool two = boundSet < 2;
                    auto selOf = [&](int partition, int which) -> uint4 {
                        return *reinterpret_cast<const uint4 *>(T->subsetByteMask[two ? partition : 64 + 2 * partition + which]);
                    };
                    uint4 selA = make_uint4(0, 0, 0, 0), selB = make_uint4(0, 0, 0, 0);
                    if constexpr (G8)
                    {
                        selA = selOf(c, 0);
                        if (!two)
                            selB = selOf(c, 1);
                    }
                    for (int k = 0; k < 16; k++)
                    {
                        const int partition = 4 * k + c;
                        float lb;
                        uint4 selA1 = selA, selB1 = selB;
                        if constexpr (G8)
                        {
                            const int nextPartition = (k < 15 ? partition + 4 : partition);
                            selA1 = selOf(nextPartition, 0);
                            if (!two)
                                selB1 = selOf(nextPartition, 1);
                        }
                        if (boundSet < 2)
                        {
                            Sums2D s1, s0;
                            if constexpr (G8)
                                maskedSumsSel(P, __popc((u32)s_pm2[partition]), selA, s1);
                            else
                                maskedSums(P, s_pm2[partition], s1);
                            s0.n = 16 - s1.n;
                            s0.u = P.tU - s1.u;
                            s0.v = P.tV - s1.v;
                            s0.uu = P.tUU - s1.uu;
                            s0.vv = P.tVV - s1.vv;
                            s0.uv = P.tUV - s1.uv;
                            lb = subsetBound2D<G8>(s0, invScaleSq, delta) + subsetBound2D<G8>(s1, invScaleSq, delta);
                        }
                        else
                        {
                            Sums2D s1, s2, s0;
                            if constexpr (G8)
                            {
                                maskedSumsSel(P, __popc(s_pm3[partition] & 0xffffu), selA, s1);
                                maskedSumsSel(P, __popc(s_pm3[partition] >> 16), selB, s2);
                            }
                            else
                            {
                            maskedSums(P, s_pm3[partition] & 0xffffu, s1);
                            maskedSums(P, s_pm3[partition] >> 16, s2);
                            }
                            s0.n = 16 - s1.n - s2.n;
                            s0.u = P.tU - s1.u - s2.u;
                            s0.v = P.tV - s1.v - s2.v;
                            s0.uu = P.tUU - s1.uu - s2.uu;
                            s0.vv = P.tVV - s1.vv - s2.vv;
                            s0.uv = P.tUV - s1.uv - s2.uv;
                            lb = subsetBound2D<G8>(s0, invScaleSq, delta) + subsetBound2D<G8>(s1, invScaleSq, delta) +
                                 subsetBound2D<G8>(s2, invScaleSq, delta);
                        }
                        if (!use4)
                            lb += s_static[blk];
                        lbStore(partition, blk, lb);
                        freshAlive |= (lb > work.err) ? 0u : (1u << k);
                        selA = selA1;
                        selB = selB1;
                    }
                    freshBounds = true;
                    };
                    if (use4 ? CVTT_GRID8_RGBA : CVTT_GRID8_RGB)
                        boundsOnGrid(std::true_type{});
                    else
                        boundsOnGrid(std::false_type{});
                }
                __syncthreads();
                WS_SET(WS_BOUNDS_FOR, boundSet + 1);
            }
        }
        PROF_MARK(2)

        // ---- every block offers its cheapest-bound candidate partition per round, until the bound
        // of the next one exceeds its best error.  Sub-lane c keeps the candidates 4k+c.
        // The offers of a round are searched by the whole wave: one lane per chain. ----
        const int CP = md.numP * 4;         // chains per subset: p-bit combinations x seed points
        const int UPB = 64 / CP;            // subsets searched at once
        // bit k: partition 4k + c is enabled for this block and its bound does not rule it out
        u32 aliveBits = (valid && laneRuns) ? everyFourth(enabled, c) : 0u; // `enabled` has no bits past the mode's partitions
        if (mode == 7 && anyBlockHasAlpha && !blockHasNonMaxAlpha)
            aliveBits &= everyFourth(mode7RGB, c); // BC67.cpp:1625-1635: this lane may not take the other partitions
        u32 tier2Done = 0; // bit k: partition 4k + c has its second-tier bound (none has when the bounds are fresh)
        if (prune)
        {
            if (freshBounds)
                aliveBits &= freshAlive; // compared while the bounds were computed: the best error has not moved since
            else
            {
                for (int k = 0; k * 4 < numPartitions; k++)
                {
                    const float raw = lbLoadRaw(4 * k + c, blk);
                    if (__builtin_fabsf(raw) > work.err)
                        aliveBits &= ~(1u << k);
                    tier2Done |= (__builtin_bit_cast(u32, raw) >> 31) << k;
                }
            }
        }

        if (prune && numSubsets >= 2)
        {
            // second tier: full-dimension bounds of the partitions that are still alive and have none yet.  A partition
            // costs about a fifth of a chain pass here and a seed pass plus chain passes if it stays.  Two passes: the plain
            // bound (subsetBoundFull) for all of them; then the sharper one (subsetBoundSharp, three times the work) for what the
            // first pass left -- on noise-like content that is little, and nothing of it goes, so a wave that sees the second
            // pass remove nothing does not run it again for modes with as many index bits.
#ifdef CVTT_NO_SHARP_TIER2
            const bool sharpPays = false;
#else
            const bool sharpPays = WS_GET(WS_SHARP_NOPAY) == 0 || md.indexBits < WS_GET(WS_SHARP_NOPAY);
#endif
            const u32 todo = aliveBits & ~tier2Done;
            const bool tier2Pays = WS_GET(WS_T2_NOPAY) == 0 || md.indexBits < WS_GET(WS_T2_NOPAY);
            const bool runPlain = tier2Pays && __ballot(todo != 0) != 0;
            const bool maySharp = sharpPays && WS_GET(WS_SHARP_RAN) > md.indexBits;
            if (runPlain || (maySharp && __ballot(aliveBits != 0) != 0))
            {
                const u32 aliveBefore = aliveBits;
                const bool use4 = (boundSet == 0);
                float lw[4];
#pragma unroll
                for (int ch = 0; ch < 4; ch++)
                    lw[ch] = A.w[ch];
                if (runPlain)
                {
                    u32 CM[4][4];
                    {
                        u32 lpix[16];
                        pixFromLds(lpix);
                        channelMajor(lpix, CM);
                    }
                    // the rest of the block: its raw sums are in LDS since the block bounds
                    auto rest = [&](RawSums &d, const RawSums &a) {
                        d.n = 16 - a.n;
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            d.s[i] = s_scatter[blk][10 + i] - a.s[i];
#pragma unroll
                        for (int i = 0; i < 10; i++)
                            d.p[i] = s_scatter[blk][i] - a.p[i];
                    };
                    // every lane walks its own list: as many rounds as the longest list, not as many as there are slots in use
                    u32 rem = todo;
                    while (__ballot(rem != 0) != 0)
                    {
                        if (rem != 0)
                        {
                            const int k = __ffs((int)rem) - 1;
                            rem &= rem - 1u;
                            const int partition = 4 * k + c;
                            float lb;
                            RawSums s0, s1;
                            if (numSubsets == 2)
                            {
                                maskedRawSumsCM(CM, s_pm2[partition], s1);
                                rest(s0, s1);
                                lb = use4 ? subsetBoundFull<4>(s1, lw, A.delta4) : subsetBoundFull<3>(s1, lw, A.delta3);
                                lb += use4 ? subsetBoundFull<4>(s0, lw, A.delta4) : subsetBoundFull<3>(s0, lw, A.delta3);
                            }
                            else
                            {
                                maskedRawSumsCM(CM, s_pm3[partition] & 0xffffu, s1);
                                rest(s0, s1);
                                lb = subsetBoundFull<3>(s1, lw, A.delta3);
                                maskedRawSumsCM(CM, s_pm3[partition] >> 16, s1);
                                rawSumsSub(s0, s1);
                                lb += subsetBoundFull<3>(s1, lw, A.delta3);
                                lb += subsetBoundFull<3>(s0, lw, A.delta3);
                            }
                            if (!use4)
                                lb += s_static[blk];
                            lbStoreTier2(partition, blk, fmaxf(lb, lbLoad(partition, blk)));
                            if (lb > work.err)
                                aliveBits &= ~(1u << k);
                        }
                    }
                }
                u32 rem2 = maySharp ? aliveBits : 0u;
                const bool fourLevels = (md.indexBits == 2);
                const int sharpNeed = (numSubsets * (fourLevels ? CVTT_SHARP_COST4 : CVTT_SHARP_COST8) + md.numP * 4 - 1) / (md.numP * 4);
                // (a wave with fewer candidates than two rounds would have to remove -- noise-like content -- does not start)
                int sharpCand = __popc(rem2);
#pragma unroll
                for (int step = 1; step < 64; step <<= 1)
                    sharpCand += xorLane(sharpCand, step);
                if (maySharp && sharpCand >= 2 * sharpNeed)
                {
                    const u32 aliveMid = aliveBits;
                    const float wSq3 = A.wSqSum3, wSq4 = A.wSqSum3 + A.wSq[3];
                    WS_SET(WS_SHARP_RAN, md.indexBits);
                    WS_SET(WS_BOUND_BITS, WS_GET(WS_BOUND_BITS) < md.indexBits ? WS_GET(WS_BOUND_BITS) : md.indexBits);
                    bool firstRound = true;
                    while (__ballot(rem2 != 0) != 0)
                    {
                        const bool did = rem2 != 0;
                        bool killed = false;
                        if (did)
                        {
                            const int k = __ffs((int)rem2) - 1;
                            rem2 &= rem2 - 1u;
                            const int partition = 4 * k + c;
                            // (subset by subset in a rolled loop: the bound of one subset needs some sixty registers)
                            float lb = 0.0f;
#pragma unroll 1
                            for (int sub = 0; sub < numSubsets; sub++)
                            {
                                const u32 sm = subsetMaskOf(numSubsets, partition, sub);
                                lb += use4 ? subsetBoundSharp<4>(&s_pix[blk][0], sm, lw, A.delta4, wSq4, fourLevels) : subsetBoundSharp<3>(&s_pix[blk][0], sm, lw, A.delta3, wSq3, fourLevels);
                            }
                            if (!use4)
                                lb += s_static[blk];
                            lbStoreTier2(partition, blk, fmaxf(lb, lbLoad(partition, blk)));
                            if (lb > work.err)
                            {
                                aliveBits &= ~(1u << k);
                                killed = true;
                            }
                        }
                        // One round of this loop costs about numSubsets chain passes (0.6 of that without the sort), and a partition
                        // it removes saves a probe: 4 * numP chain lanes for a round and a half.  A round that removes fewer
                        // partitions than it costs ends the loop (the lists of the lanes run out one by one), and when that is
                        // the very first round the wave does not try again in the stages with as many index bits.
                        PROF_STAGE2(stageIter, 0, 1)
                        PROF_STAGE2(stageIter, 1, __popcll(__ballot(did)))
                        PROF_STAGE2(stageIter, 2, __popcll(__ballot(killed)))
                        if (__popcll(__ballot(killed)) < sharpNeed)
                        {
                            if (firstRound)
                                WS_SET(WS_SHARP_NOPAY, md.indexBits);
                            break;
                        }
                        firstRound = false;
                    }
                }
                // content on which these bounds remove nothing (errors dominated by quantisation, not by the fit of
                // a line) does not get them again in the later stages of this wave
                if (runPlain && __ballot(aliveBits != aliveBefore) == 0)
                    WS_SET(WS_T2_NOPAY, md.indexBits);
            }
        }
#ifdef CVTT_BC7_PROFILE_SPLIT
        PROF_MARK(4)
#endif
        // nothing left for any block of the wave (the rule on RGBA noise): on to the next mode
        if (__ballot(aliveBits != 0) == 0)
            continue;
        if (!PT && !HARD && mode == 7 && A.hardCap != 0)
        {
            // A wave with many partitions to search runs for tens of chain passes.  That only matters when it starts
            // near the end of the launch (workgroups are dispatched in order), where it would run on alone: the number
            // of partitions a wave may keep shrinks with the number of waves still to be dispatched after it.
            int cnt = __popc(aliveBits);
            cnt += xorLane(cnt, 1);
            cnt += xorLane(cnt, 2);
            int waveCnt = cnt;
#pragma unroll
            for (int step = 4; step < 64; step <<= 1)
                waveCnt += xorLane(waveCnt, step);
            u32 allowed = (vGrid - vBlock) / A.hardDiv;
            allowed = allowed > A.hardMin ? allowed : A.hardMin;
            // the blocks that make up most of the excess go; an eighth of the allowance each may stay
            if ((u32)waveCnt >= allowed && cnt >= 2 && (u32)cnt >= allowed / 8u)
            {
                u32 slot = 0;
                if (c == 0)
                    slot = atomicAdd(A.hardCount, 1u);
                slot = __shfl(slot, lane & ~3);
                if (slot < A.hardCap)
                {
                    // bit k of sub-lane c is partition 4k + c
                    u64 m = aliveBits;
                    m = (m | (m << 24)) & 0x000000ff000000ffull;
                    m = (m | (m << 12)) & 0x000f000f000f000full;
                    m = (m | (m << 6)) & 0x0303030303030303ull;
                    m = (m | (m << 3)) & 0x1111111111111111ull;
                    m <<= c;
                    u32 lo = (u32)m, hi = (u32)(m >> 32);
                    lo |= xorLane(lo, 1);
                    hi |= xorLane(hi, 1);
                    lo |= xorLane(lo, 2);
                    hi |= xorLane(hi, 2);
                    aliveBits = 0;
                    if (c == 0)
                    {
                        A.hardRec[slot].aliveLo = lo;
                        A.hardRec[slot].aliveHi = hi;
                        s_blkFlags[blk] |= (slot + 1u) << 8; // remembered in LDS until the block's best is final
                    }
                }
            }
        }
        PROF_COUNT(4, __popc(aliveBits))
        PROF_COUNT(5, (c == 0 && valid) ? 1 : 0)
#ifdef CVTT_BC7_PROFILE
        {
            int cntAlive = __popc(aliveBits);
            for (int st = 1; st < 64; st <<= 1) cntAlive += xorLane(cntAlive, st);
            PROF_STAGE(stageIter, 2, cntAlive)
            PROF_STAGE(stageIter, 6, 1)
        }
#endif
        const int itemCap = (numSubsets == 3) ? 21 : 32; // items * subsets <= 64 lanes of the seed pass
        // BC7_RespectPunchThrough couples the 8 blocks of a group in modes 6 and 7 (BC67.cpp:1283-1428): a
        // partition one block wants is searched by its whole group, trial by trial in lock-step
        const bool ptStage = PT && (mode == 6 || mode == 7);
        // Everything else takes its partitions SUBSET BY SUBSET.  The offers of a round (a set of partitions per block, kept
        // as a 64-bit mask) are first PROBED: one unit per partition, its largest subset, searched exactly; the subset's
        // exact error plus the full-dimension bound of the other subset(s) is again a lower bound of the partition's total,
        // and on smooth, photo-like and two-colour content -- where the geometric bounds say little because the errors are
        // quantisation errors -- it rules out 76-99 % of the probed partitions (profiles/r03/bc7_stage_whatif.txt).  Only
        // the survivors are then searched in full (all subsets, payload kept) and committed as before.  The items of a
        // phase are numbered block by block (a block's items are consecutive), in chunks of what the 64 result slots hold.
        const bool staged = !ptStage;
        const bool probing = staged && prune && numSubsets >= 2 && (A.prune & 2u) == 0; // (bit 1 of A.prune: developer knob, probes off)
        const int offerCap = probing ? 64 : itemCap;
        // the subset of a partition that is probed: the one with the most pixels (the first of them)
        auto probeSubOf = [&](int partition) -> int {
            if (numSubsets == 2)
                return __popc((u32)s_pm2[partition]) > 8 ? 1 : 0;
            const u32 m = s_pm3[partition];
            const int n1 = __popc(m & 0xffffu), n2 = __popc(m >> 16), n0 = 16 - n1 - n2;
            return (n2 > n0 && n2 > n1) ? 2 : (n1 > n0) ? 1 : 0;
        };
        // survivors of the probes wait here (per block) until the wave has enough of them for a well-filled full search
        // (CVTT_PEND_MIN partitions) or the stage has nothing left to offer
        u64 pend = 0;
        for (;;)
        {
            // ---- offers.  Pass 0: every block offers its cheapest-bound candidate.  When few
            // blocks offer, the idle lanes take further candidates of the same blocks (they might have been
            // pruned by the first result, but waiting for it would cost a whole round). ----
            REFRESH_LANE();
            int numItems = 0, myCount = 0, maxPasses = 1;
            u64 offerMask = 0; // staged: the partitions this block offers in this round (the same in the four lanes of the quad)
            if (probing)
            {
                // When the partitions the whole wave has left fit one probe phase (64) -- noise-like content, where the bounds
                // leave few -- there is nothing to order and the offer passes are skipped: all of them are probed at once
                // (opaque noise +13 %, near-opaque alpha +22 %).  With more than that the order matters: cheapest bound first, a
                // few per block and round, so that what the first survivors achieve prunes the rest (probing a stage block by
                // block instead measured -12 % on smooth content).
                u64 m = aliveBits; // bit k of sub-lane c is partition 4k + c
                m = (m | (m << 24)) & 0x000000ff000000ffull;
                m = (m | (m << 12)) & 0x000f000f000f000full;
                m = (m | (m << 6)) & 0x0303030303030303ull;
                m = (m | (m << 3)) & 0x1111111111111111ull;
                m <<= c;
                u32 lo = (u32)m, hi = (u32)(m >> 32);
                lo |= xorLane(lo, 1);
                hi |= xorLane(hi, 1);
                lo |= xorLane(lo, 2);
                hi |= xorLane(hi, 2);
                const int kb = __popc(lo) + __popc(hi);
                int allTotal = 0;
#pragma unroll
                for (int b = 0; b < 16; b++)
                    allTotal += __builtin_amdgcn_readlane(kb, 4 * b); // (a scalar: no lane-address register per block to keep alive)
                allTotal = __builtin_amdgcn_readfirstlane(allTotal);
                if (allTotal <= 64)
                {
                    offerMask = ((u64)hi << 32) | lo;
                    aliveBits = 0;
                    numItems = allTotal;
                    maxPasses = 0;
                }
            }
            // Probing stages: every LANE offers the cheapest of the sixteen partitions it looks after (a block offers up to four
            // per pass, nearly its four cheapest), so a full wave fills a probe phase in one pass over the bounds instead of four
            // with a quad-wide argmin each.  The order only steers the pruning, never the result.
            const bool laneOffers = probing;
            for (int pass = 0; pass < maxPasses; pass++)
            {
                // argmin of the bounds still alive (ties: lowest partition), over the quad unless every lane offers for itself
                float pickLb = FLT_MAX;
                int pick = 255;
                for (int k = 0; k * 4 < numPartitions; k++)
                {
                    if ((aliveBits >> k) & 1u)
                    {
                        const float lb = prune ? lbLoad(4 * k + c, blk) : 0.0f;
                        if (lb < pickLb)
                        {
                            pickLb = lb;
                            pick = 4 * k + c;
                        }
                    }
                }
                if (!laneOffers)
                {
#pragma unroll
                    for (int step = 1; step <= 2; step <<= 1)
                    {
                        const float oLb = xorLane(pickLb, step);
                        const int oPick = xorLane(pick, step);
                        const bool take = (oLb < pickLb) || (oLb == pickLb && oPick < pick);
                        pickLb = take ? oLb : pickLb;
                        pick = take ? oPick : pick;
                    }
                }
                bool offer = pick < 64 && !(pickLb > work.err);
                if (!offer)
                    aliveBits = 0; // everything left costs even more
                if (ptStage)
                {
                    // the cheapest offer inside the group becomes the offer of all its blocks
                    if (!offer)
                    {
                        pickLb = FLT_MAX;
                        pick = 255;
                    }
#pragma unroll
                    for (int step = 4; step <= 16; step <<= 1)
                    {
                        const float oLb = xorLane(pickLb, step);
                        const int oPick = xorLane(pick, step);
                        const bool take = (oLb < pickLb) || (oLb == pickLb && oPick < pick);
                        pickLb = take ? oLb : pickLb;
                        pick = take ? oPick : pick;
                    }
                    offer = pick < 64;
                }
                const u64 offers = __ballot(offer && (laneOffers || c == 0));
                const int numOffers = __popcll(offers);
                if (numOffers == 0 || numItems + numOffers > offerCap)
                    break;
                if (pass == 0 && !ptStage)
                {
                    if (laneOffers)
                        maxPasses = numOffers >= 33 ? 1 : numOffers >= 17 ? 2 : numOffers >= 9 ? 4 : numOffers >= 5 ? 8 : 16; // about 64 probes per round
                    else
                        maxPasses = (numOffers <= 1) ? CVTT_SPEC_1 : (numOffers <= 2) ? CVTT_SPEC_2 : (numOffers <= 4) ? CVTT_SPEC_4 : (numOffers <= 8) ? CVTT_SPEC_8 : CVTT_SPEC_16;
                }
                if (offer)
                {
                    if (staged)
                        offerMask |= 1ull << pick;
                    else
                    {
                        const int item = numItems + __popcll(offers & ((1ull << (lane & ~3)) - 1ull));
                        if (c == 0)
                        {
                            s_item[item] = (u32)blk | ((u32)pick << 8);
                            s_myItems[PT ? blk : 0][pass] = (uint8_t)item;
                        }
                    }
                    if ((pick & 3) == c)
                        aliveBits &= ~(1u << (pick >> 2));
                    myCount = pass + 1;
                }
                numItems += numOffers;
            }
            if (laneOffers)
            {
                u32 lo = (u32)offerMask, hi = (u32)(offerMask >> 32);
                lo |= xorLane(lo, 1);
                hi |= xorLane(hi, 1);
                lo |= xorLane(lo, 2);
                hi |= xorLane(hi, 2);
                offerMask = ((u64)hi << 32) | lo;
            }
            PROF_MARK(8)
            const bool flush = (numItems == 0); // nothing left to offer: the waiting survivors are searched, then the stage ends
            if (flush && __ballot(pend != 0) == 0)
                break;
            PROF_STAGE(stageIter, 3, 1)
            bool isProbe = probing && !flush;
            u64 curMask = offerMask; // staged: this block's partitions of the current phase
            if (!isProbe)
            {
                curMask |= pend;
                pend = 0;
            }
            int chunkLo = 0;
            for (;;) // phases of the round: [probe,] then the full search of the survivors, chunk by chunk
            {
            PROF_MARK(14)
            REFRESH_LANE();
            int iLo = 0, iHi = 0, phaseTotal = numItems;
            if (staged)
            {
                // number the items block by block and write this chunk's part of the list
                const int kb = __popcll(curMask);
                int base = 0;
                phaseTotal = 0;
#pragma unroll
                for (int b = 0; b < 16; b++)
                {
                    const int o = __builtin_amdgcn_readlane(kb, 4 * b);
                    base += (b < blk) ? o : 0;
                    phaseTotal += o;
                }
                phaseTotal = __builtin_amdgcn_readfirstlane(phaseTotal);
                const int cap = isProbe ? 64 : itemCap;
                numItems = phaseTotal - chunkLo < cap ? phaseTotal - chunkLo : cap;
                iLo = base - chunkLo;
                iHi = iLo + kb;
                iLo = iLo < 0 ? 0 : (iLo > numItems ? numItems : iLo);
                iHi = iHi < 0 ? 0 : (iHi > numItems ? numItems : iHi);
                // sub-lane c files the partitions 16c .. 16c+15 of the block's mask
                u32 part = (u32)(curMask >> (16 * c)) & 0xffffu;
                int rnk = base - chunkLo + __popcll(curMask & ((1ull << (16 * c)) - 1ull));
                while (part)
                {
                    const int bit = __ffs((int)part) - 1;
                    part &= part - 1u;
                    if (rnk >= 0 && rnk < numItems)
                        s_item[rnk] = (u32)blk | ((u32)(16 * c + bit) << 8);
                    rnk++;
                }
            }
            const int numUnits = isProbe ? numItems : numItems * numSubsets;
            PROF_STAGE(stageIter, isProbe ? 7 : 1, numUnits)
            __syncthreads();

            // ---- PCA seed search: lane l takes unit l = (item, subset) ----
            // The chain lanes of a batch walk the member pixels of their subsets in lock-step, i.e. as many steps as the
            // largest subset of the batch has pixels: the units are therefore filed in ascending order of their pixel
            // counts (a counting sort on ballots), so that a batch holds subsets of (nearly) one size.  Results are routed
            // by UnitRec::slot, so the order of the units is free.  (The punch-through replay indexes units by item.)
            int unitPos = lane;
            {
                int cnt = 0;
                if (lane < numUnits && numSubsets >= 2 && !ptStage)
                {
                    const int item = isProbe ? lane : (numSubsets == 2) ? (lane >> 1) : (lane / 3);
                    const int upart = (int)(s_item[item] >> 8);
                    const int sub = isProbe ? probeSubOf(upart) : lane - item * numSubsets;
                    cnt = __popc(subsetMaskOf(numSubsets, upart, sub));
                }
                if (numSubsets >= 2 && !ptStage)
                {
                    int base = 0;
                    for (int v = 1; v < 16; v++)
                    {
                        const u64 m = __ballot(cnt == v);
                        if (m == 0)
                            continue;
                        if (cnt == v)
                            unitPos = base + (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                        base += __popcll(m);
                    }
                }
            }
            PROF_MARK(9)
            REFRESH_LANE();
            if (lane < numUnits)
            {
                const int item = (isProbe || numSubsets == 1) ? lane : (numSubsets == 2) ? (lane >> 1) : (lane / 3);
                const u32 it = s_item[item];
                const int ublk = (int)(it & 255u), upart = (int)(it >> 8);
                const int sub = isProbe ? probeSubOf(upart) : lane - item * numSubsets;
                int shape = 0;
                if (numSubsets == 2)
                    shape = T->shapes2[upart][sub];
                else if (numSubsets == 3)
                    shape = T->shapes3[upart][sub];
                const u32 uMask = subsetMaskOf(numSubsets, upart, sub);
                int seeds = isRGB ? plan->seedPointsForShapeRGB[shape] : plan->seedPointsForShapeRGBA[shape];
                if (seeds > 4)
                    seeds = 4;
                const bool rgbListed = ((dplan->rgbListed[shape >> 5] >> (shape & 31)) & 1u) != 0;
                const bool rgbaListed = isRGB ? true : (((dplan->rgbaListed[shape >> 5] >> (shape & 31)) & 1u) != 0);
                const bool uWantPCA4 = (s_blkFlags[ublk] & 1u) != 0;
                const bool wanted = seeds > 0;
                // which PCA does this unit need?  (BC67.cpp:1085-1144; unlisted shapes keep zero seeds)
                const bool do4 = wanted && !isRGB && uWantPCA4 && rgbaListed;
                const bool do3 = wanted && rgbListed && (isRGB || (!uWantPCA4 && rgbaListed));
                const bool expandAlpha = !isRGB && wanted && !uWantPCA4 && rgbaListed;
                Unfinished uu;
#pragma unroll
                for (int ch = 0; ch < 4; ch++)
                    uu.base[ch] = uu.offset[ch] = 0.0f;
                const u32 *lp = &s_pix[ublk][0];
                float vsum[4] = {0.0f, 0.0f, 0.0f, 0.0f}; // the refiner's m_v (UnitRec::vs); the PCA's first pass forms the same sums
                if (do3)
                {
                    Unfinished u3;
                    pcaEndpointsLDS<3>(lp, uMask, A.w, u3, vsum);
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        uu.base[ch] = u3.base[ch];
                        uu.offset[ch] = u3.offset[ch];
                    }
                }
                if (expandAlpha)
                {
                    uu.base[3] = 255.0f; // ExpandTo<4>(255), UnfinishedEndpoints.h:93-114
                    uu.offset[3] = 0.0f;
                }
                if (do4)
                    pcaEndpointsLDS<4>(lp, uMask, A.w, uu, vsum);
                if (wanted && !do4)
                {
                    // no four-channel pass ran: alpha of an RGBA mode whose seeds come from the RGB pass, or every channel
                    // of a shape the plan does not list (zero seeds, but its chains still run)
#pragma unroll
                    for (int ch = 0; ch < 4; ch++)
                        if (ch == 3 ? !isRGB : !do3)
                        {
                            float a = 0.0f;
#pragma unroll
                            for (int px = 0; px < 16; px++)
                                if ((uMask >> px) & 1u)
                                    a = a + byteF(lp[px], ch) * A.w[ch];
                            vsum[ch] = a;
                        }
                }
                UnitRec &r = s_unit[unitPos];
#pragma unroll
                for (int ch = 0; ch < 4; ch++)
                {
                    r.base[ch] = uu.base[ch];
                    r.offset[ch] = uu.offset[ch];
                }
#pragma unroll
                for (int ch = 0; ch < 4; ch++)
                    r.vs[ch] = vsum[ch];
                // where the unit's result goes: a probe's by item, a full search's by (item, subset)
                r.packed = uMask | ((u32)(seeds < 0 ? 0 : seeds) << 16) | ((u32)ublk << 20) | ((u32)(isProbe ? item : item * numSubsets + sub) << 24);
                float scErr = FLT_MAX;
                if ((A.flags & CVTTMI_FLAG_BC7_TRY_SINGLE_COLOR) && seeds > 0)
                {
                    // BC67.cpp:1436-1570 + 940-1040.  With the reference's AndNot operand order no
                    // single-colour table entry is ever taken (finite weights), so what reaches the
                    // error test is endpoints (0,0,0[,255]) with index 0 for every pixel.
                    u32 e[4] = {0, 0, 0, 0}, st = 0;
#pragma unroll
                    for (int px = 0; px < 16; px++)
                        if ((uMask >> px) & 1u)
                        {
                            const u32 pk = lp[px];
                            e[0] = (u32)mad24(byteI(pk, 0), byteI(pk, 0), (int)e[0]);
                            e[1] = (u32)mad24(byteI(pk, 1), byteI(pk, 1), (int)e[1]);
                            e[2] = (u32)mad24(byteI(pk, 2), byteI(pk, 2), (int)e[2]);
                            const int d = 255 - byteI(pk, 3);
                            if (isRGB)
                                st = (u32)mad24(d, d, (int)st);
                            else
                                e[3] = (u32)mad24(d, d, (int)e[3]);
                        }
                    const bool uniformErr = (A.flags & CVTTMI_FLAG_UNIFORM) != 0;
                    if (uniformErr)
                        scErr = (float)(int)(e[0] + e[1] + e[2] + e[3]);
                    else
                    {
                        scErr = (float)(int)e[0] * A.wSq[0];
                        scErr = scErr + (float)(int)e[1] * A.wSq[1];
                        scErr = scErr + (float)(int)e[2] * A.wSq[2];
                        scErr = scErr + (float)(int)e[3] * A.wSq[3];
                    }
                    if (isRGB)
                        scErr = scErr + (uniformErr ? (float)(int)st : (float)(int)st * A.wSq[3]);
                }
                r.scErr = scErr;
                {
                    u32 lo = 0, hi = 0;
                    int k = 0;
#pragma unroll
                    for (int px = 0; px < 16; px++)
                        if ((uMask >> px) & 1u)
                        {
                            if (k < 8)
                                lo |= (u32)px << (4 * k);
                            else
                                hi |= (u32)px << (4 * k - 32);
                            k++;
                        }
                    r.listLo = lo;
                    r.listHi = hi;
                }
            }
            __syncthreads();
            PROF_MARK(10)

            REFRESH_LANE();
            // ---- probes with two refine rounds (the default): round 1 only once per distinct set of end points.  The four seed
            // points of a (unit, p-bits) often refine to the same end points, or to end points round 0 has already tried
            // (24-47 % of all chain rounds on smooth / photo-like / two-colour content, tools/bc7_dup_profile.py) -- the same
            // end points on the same pixels give the same error, and a probe wants nothing but the minimum.  So a batch runs
            // round 0 of its 64 chains, the chains whose round-1 end points are new file them (unit, end points) in a ring in
            // LDS (the payload words of s_res, idle during a probe), and whenever 64 are waiting one batch runs them; the
            // minima are taken with ds_min_u32 on the error bits (errors are >= 0). ----
            const bool dedup = isProbe && numRefine == 2 && (A.prune & 4u) == 0;
            if (dedup)
            {
                constexpr int kRing = 85; // items of three dwords in words 1..4 of the 64 result rows
                auto ringWord = [&](int item, int j) -> u32 * {
                    const int flat = item * 3 + j;
                    return &s_res[flat >> 2][1 + (flat & 3)];
                };
                if (lane < numUnits)
                {
                    const UnitRec &ru = s_unit[lane];
                    s_res[ru.slot()][0] = __builtin_bit_cast(u32, ru.scErr); // the single-colour try (FLT_MAX when there is none)
                }
                __syncthreads();
                int ringHead = 0, ringCount = 0;
                auto runRound1 = [&](int n) {
                    // lane l < n takes the l-th waiting item
                    const bool have = lane < n;
                    int it = ringHead + lane;
                    it -= it >= kRing ? kRing : 0;
                    const u32 w0 = have ? *ringWord(it, 0) : 0u;
                    ChainSplit sp;
                    sp.firstRound = 1;
                    sp.endRound = 2;
                    sp.ep0 = have ? *ringWord(it, 1) : 0u;
                    sp.ep1 = have ? *ringWord(it, 2) : 0u;
                    sp.r0ep0 = sp.r0ep1 = 0;
                    const UnitRec &r = s_unit[w0 & 63u];
                    const u32 uMask = r.mask();
                    int maxCount = have ? __popc(uMask) : 0;
#pragma unroll
                    for (int step = 1; step < 64; step <<= 1)
                    {
                        const int o = xorLane(maxCount, step);
                        maxCount = o > maxCount ? o : maxCount;
                    }
                    maxCount = __builtin_amdgcn_readfirstlane(maxCount);
                    Unfinished uu;
#pragma unroll
                    for (int ch = 0; ch < 4; ch++)
                        uu.base[ch] = uu.offset[ch] = 0.0f; // not used: the end points come with the item
                    const float uvs[4] = {0.0f, 0.0f, 0.0f, 0.0f}; // the last round refines nothing
                    const u64 uList = ((u64)r.listHi << 32) | r.listLo;
                    ShapeBest b;
                    if (isRGB)
                        evalChain<3, FAST, false>(&s_pix[r.blk()][0], uMask, uList, maxCount, md, uu, 0, 0, have, A, T, numRefine, b, uvs, false, nullptr, -1, nullptr, &sp);
                    else
                        evalChain<4, FAST, false>(&s_pix[r.blk()][0], uMask, uList, maxCount, md, uu, 0, 0, have, A, T, numRefine, b, uvs, false, nullptr, -1, nullptr, &sp);
                    if (have)
                        atomicMin(&s_res[r.slot()][0], __builtin_bit_cast(u32, b.err));
                    ringHead += n;
                    ringHead -= ringHead >= kRing ? kRing : 0;
                    ringCount -= n;
                    __syncthreads();
                };
                for (int u0 = 0; u0 < numUnits; u0 += UPB)
                {
                    const int unit = u0 + lane / CP;
                    const int chain = lane & (CP - 1);
                    const bool inRange = unit < numUnits;
                    const UnitRec &r = s_unit[inRange ? unit : 0];
                    const int tweak = chain & 3, pIter = chain >> 2;
                    const u32 uMask = r.mask();
                    const bool active = inRange && tweak < r.numTweak();
                    int maxCount = active ? __popc(uMask) : 0;
#pragma unroll
                    for (int step = 1; step < 64; step <<= 1)
                    {
                        const int o = xorLane(maxCount, step);
                        maxCount = o > maxCount ? o : maxCount;
                    }
                    maxCount = __builtin_amdgcn_readfirstlane(maxCount);
                    Unfinished uu;
#pragma unroll
                    for (int ch = 0; ch < 4; ch++)
                    {
                        uu.base[ch] = r.base[ch];
                        uu.offset[ch] = r.offset[ch];
                    }
                    const float uvs[4] = {r.vs[0], r.vs[1], r.vs[2], r.vs[3]};
                    const u64 uList = ((u64)r.listHi << 32) | r.listLo;
                    PROF_STAGE(stageIter, 0, 1)
                    PROF_STAGE_LANES(stageIter, 5, active)
                    ChainSplit sp;
                    sp.firstRound = 0;
                    sp.endRound = 1;
                    sp.ep0 = sp.ep1 = sp.r0ep0 = sp.r0ep1 = 0;
                    ShapeBest b;
                    if (isRGB)
                        evalChain<3, FAST, false>(&s_pix[r.blk()][0], uMask, uList, maxCount, md, uu, pIter, tweak, active, A, T, numRefine, b, uvs, false, nullptr, -1, nullptr, &sp);
                    else
                        evalChain<4, FAST, false>(&s_pix[r.blk()][0], uMask, uList, maxCount, md, uu, pIter, tweak, active, A, T, numRefine, b, uvs, false, nullptr, -1, nullptr, &sp);
                    if (active)
                        atomicMin(&s_res[r.slot()][0], __builtin_bit_cast(u32, b.err));
                    // is my round 1 new?  Not if a lower seed point of my (unit, p-bits) goes to the same end points, and not
                    // if some seed point of it has just tried them in round 0
                    bool needed = active;
#pragma unroll
                    for (int o = 0; o < 4; o++)
                    {
                        const int src = (lane & ~3) | o;
                        const bool oAct = __shfl((int)active, src) != 0;
                        const u32 n0 = __shfl(sp.ep0, src), n1 = __shfl(sp.ep1, src);
                        const u32 q0 = __shfl(sp.r0ep0, src), q1 = __shfl(sp.r0ep1, src);
                        if (oAct && o < (lane & 3) && n0 == sp.ep0 && n1 == sp.ep1)
                            needed = false;
                        if (oAct && q0 == sp.ep0 && q1 == sp.ep1)
                            needed = false;
                    }
                    const u64 nm = __ballot(needed);
                    const int nNew = __popcll(nm);
                    // room for them?  (only when the ring is nearly full: a short batch then)
                    if (ringCount + nNew > kRing)
                        runRound1(ringCount < 64 ? ringCount : 64);
                    if (needed)
                    {
                        int it = ringHead + ringCount + (int)__builtin_amdgcn_mbcnt_hi((u32)(nm >> 32), __builtin_amdgcn_mbcnt_lo((u32)nm, 0u));
                        it -= it >= kRing ? kRing : 0;
                        *ringWord(it, 0) = (u32)unit;
                        *ringWord(it, 1) = sp.ep0;
                        *ringWord(it, 2) = sp.ep1;
                    }
                    ringCount += nNew;
                    __syncthreads();
                    while (ringCount >= 64)
                        runRound1(64);
                }
                while (ringCount > 0)
                    runRound1(ringCount < 64 ? ringCount : 64);
            }
            else
            // ---- chains: lane l of a batch = (unit l / CP, p-bits (l % CP) / 4, seed point l % 4) ----
            for (int u0 = 0; u0 < numUnits; u0 += UPB)
            {
                const int unit = u0 + lane / CP;
                const int chain = lane & (CP - 1);
                const bool inRange = unit < numUnits;
                const UnitRec &r = s_unit[inRange ? unit : 0];
                const int tweak = chain & 3, pIter = chain >> 2;
                const u32 uMask = r.mask();
                const bool active = inRange && tweak < r.numTweak();
                int maxCount = active ? __popc(uMask) : 0;
#pragma unroll
                for (int step = 1; step < 64; step <<= 1)
                {
                    const int o = xorLane(maxCount, step);
                    maxCount = o > maxCount ? o : maxCount;
                }
                maxCount = __builtin_amdgcn_readfirstlane(maxCount);
                Unfinished uu;
#pragma unroll
                for (int ch = 0; ch < 4; ch++)
                {
                    uu.base[ch] = r.base[ch];
                    uu.offset[ch] = r.offset[ch];
                }
                const u32 *lp = &s_pix[r.blk()][0];
                const float uvs[4] = {r.vs[0], r.vs[1], r.vs[2], r.vs[3]};
                const u64 uList = ((u64)r.listHi << 32) | r.listLo;
                PROF_COUNT(2, 64)
                PROF_COUNT(3, __popcll(__ballot(active)))
                PROF_STAGE(stageIter, 0, 1)
                PROF_STAGE_LANES(stageIter, 5, active)
                ShapeBest b;
                if (ptStage)
                {
                    // record every trial; the lock-step commit rule is applied below
                    if (PT)
                        evalChain<4, FAST, true>(lp, uMask, uList, maxCount, md, uu, pIter, tweak, active, A, T, numRefine, b, uvs, true,
                                                 trialHbm ? nullptr : &s_trialErr[((inRange ? unit : 0) * 16 + chain) * numRefine], -1,
                                                 trialHbm ? trialHbm + ((inRange ? unit : 0) * 16 + chain) * numRefine : nullptr);
                    continue;
                }
                if (isRGB)
                    evalChain<3, FAST, false>(lp, uMask, uList, maxCount, md, uu, pIter, tweak, active, A, T, numRefine, b, uvs, !isProbe);
                else
                    evalChain<4, FAST, false>(lp, uMask, uList, maxCount, md, uu, pIter, tweak, active, A, T, numRefine, b, uvs, !isProbe);
                groupArgminBroadcast(b, lane, CP);
                if (inRange && chain == 0)
                {
                    if (r.scErr < b.err)
                    {
                        // the single-colour try comes after every chain of the shape (strict '<')
                        b.err = r.scErr;
                        b.ep0 = b.ep1 = isRGB ? 0u : 0xff000000u;
                        b.idxLo = b.idxHi = 0;
                    }
                    // with no seed points the shape keeps its reset error FLT_MAX (BC67.cpp:1228-1242)
                    u32 *dst = &s_res[r.slot()][0];
                    dst[0] = __builtin_bit_cast(u32, b.err);
                    dst[1] = b.ep0;
                    dst[2] = b.ep1;
                    dst[3] = b.idxLo;
                    dst[4] = b.idxHi;
                }
            }
            __syncthreads();
            if (isProbe) { PROF_MARK(11) } else { PROF_MARK(13) }
            REFRESH_LANE();

            if (isProbe)
            {
                // ---- which of the probed partitions can still win?  Exact error of the probed subset + full-dimension
                // bound of the other subset(s) <= the partition's total in the reference: for two subsets the total is the
                // float sum of the same two kinds of terms (rounding is monotone); for three the association differs, which
                // the factor below covers (the total is >= the real sum * (1 - 2u), this sum <= the real one * (1 + 2u)).
                // Sub-lane c takes the block's items c, c + 4, ... ----
                u64 surv = 0;
                if (__ballot(iLo + c < iHi) != 0)
                {
                    const bool use4 = (boundSet == 0);
                    const float lwf[4] = {A.w[0], A.w[1], A.w[2], A.w[3]};
                    {
                        u32 CMf[4][4];
                        {
                            u32 lpix[16];
                            pixFromLds(lpix);
                            channelMajor(lpix, CMf);
                        }
                        for (int i = iLo + c; __ballot(i < iHi) != 0; i += 4)
                        {
                            if (i < iHi)
                            {
                                const int partition = (int)(s_item[i] >> 8);
                                const float eProbed = __builtin_bit_cast(float, s_res[i][0]);
                                const int ps = probeSubOf(partition);
                                float lbRest = 0.0f;
                                for (int sub = 0; sub < numSubsets; sub++)
                                {
                                    if (sub == ps)
                                        continue;
                                    RawSums rs;
                                    maskedRawSumsCM(CMf, subsetMaskOf(numSubsets, partition, sub), rs);
                                    lbRest += use4 ? subsetBoundFull<4>(rs, lwf, A.delta4) : subsetBoundFull<3>(rs, lwf, A.delta3);
                                }
                                const float lbTotal = (eProbed + lbRest) * 0.9999995f;

                                if (!(lbTotal > work.err))
                                    surv |= 1ull << partition;
                            }
                        }
                    }
                    // what the plain bound of the rest leaves gets the sharper one (in a wave where that one pays: see the second tier)
                    // (a partition it removes saves a full search -- numSubsets * 4 numP chain lanes for two rounds --, a round of
                    // this loop costs about numSubsets - 1 chain passes: it runs as long as a round removes what it costs, and a
                    // wave gives it up after CVTT_FILTER_STRIKES calls in a row whose first round did not)
                    const int filterNeed = ((numSubsets - 1) * (md.indexBits == 2 ? 40 : 26) + numSubsets * md.numP * 4 - 1) / (numSubsets * md.numP * 4);
                    // (surv is per lane here: sub-lane c has filed the items c, c + 4, ... of its block)
                    int filterCand = __popcll(surv);
#pragma unroll
                    for (int step = 1; step < 64; step <<= 1)
                        filterCand += xorLane(filterCand, step);
#ifdef CVTT_NO_SHARP_FILTER
                    if (false)
#else
                    if (WS_GET(WS_STRIKES) < CVTT_FILTER_STRIKES && filterCand >= 2 * filterNeed)
#endif
                    {
                        bool firstRound = true;
                        for (int i = iLo + c; __ballot(i < iHi) != 0; i += 4)
                        {
                            const int partition = i < iHi ? (int)(s_item[i] >> 8) : 0;
                            bool killed = false;
                            if (i < iHi && ((surv >> partition) & 1ull))
                            {
                                const float eProbed = __builtin_bit_cast(float, s_res[i][0]);
                                const int ps = probeSubOf(partition);
                                float lbRest = 0.0f;
#pragma unroll 1
                                for (int sub = 0; sub < numSubsets; sub++)
                                {
                                    if (sub == ps)
                                        continue;
                                    const u32 sm = subsetMaskOf(numSubsets, partition, sub);
                                    lbRest += use4 ? subsetBoundSharp<4>(&s_pix[blk][0], sm, lwf, A.delta4, A.wSqSum3 + A.wSq[3], md.indexBits == 2)
                                                   : subsetBoundSharp<3>(&s_pix[blk][0], sm, lwf, A.delta3, A.wSqSum3, md.indexBits == 2);
                                }
                                const float lbTotal = (eProbed + lbRest) * 0.9999995f;
                                if (lbTotal > work.err)
                                {
                                    surv &= ~(1ull << partition);
                                    killed = true;
                                }
                            }
                            PROF_STAGE2(stageIter, 3, 1)
                            PROF_STAGE2(stageIter, 4, __popcll(__ballot(i < iHi && ((surv >> partition) & 1ull))) + __popcll(__ballot(killed)))
                            PROF_STAGE2(stageIter, 5, __popcll(__ballot(killed)))
                            if (__popcll(__ballot(killed)) < filterNeed)
                            {
                                if (firstRound)
                                    WS_SET(WS_STRIKES, WS_GET(WS_STRIKES) + 1);
                                break;
                            }
                            WS_SET(WS_STRIKES, 0);
                            firstRound = false;
                        }
                    }
                }
                {
                    u32 lo = (u32)surv, hi = (u32)(surv >> 32);
                    lo |= xorLane(lo, 1);
                    hi |= xorLane(hi, 1);
                    lo |= xorLane(lo, 2);
                    hi |= xorLane(hi, 2);
                    surv = ((u64)hi << 32) | lo;
                }
                pend |= surv;
                int pendTotal = 0;
                {
                    const int kb = __popcll(pend);
#pragma unroll
                    for (int b = 0; b < 16; b++)
                        pendTotal += __builtin_amdgcn_readlane(kb, 4 * b);
                    pendTotal = __builtin_amdgcn_readfirstlane(pendTotal);
                }
                __syncthreads(); // everybody has read its items before the next list is written
                PROF_MARK(12)
                if (pendTotal < CVTT_PEND_MIN)
                    break; // they wait for more
                curMask = pend;
                pend = 0;
                isProbe = false;
                chunkLo = 0;
                continue;
            }

            if (PT && ptStage)
            {
                __syncthreads();
                // ---- the reference's commit rule, trial by trial, for the 8 blocks of a group together
                // (BC67.cpp:1283-1428): lane = (group of the round, subset, block of the group) ----
                const int perGroup = 8 * numSubsets;
                const int gi = lane / perGroup, sub = (lane - gi * perGroup) >> 3, l8 = lane & 7;
                const int item = gi * 8 + l8;
                const bool scanActive = item < numItems && lane < (numItems / 8) * perGroup;
                const int unit = scanActive ? item * numSubsets + sub : 0;
                const UnitRec &r = s_unit[unit];
                const u32 bf = s_blkFlags[r.blk()];
                const bool isPT = (bf & 2u) != 0, nonZeroA = (bf & 4u) != 0, nonMaxA = (bf & 8u) != 0;
                const int slice = lane & ~7;
                float held = FLT_MAX; // shapeBestError
                int last = -1;        // the trial whose result the shape holds
                for (int pI = 0; pI < 4; pI++)
                {
                    const bool invalid = scanActive && ((pI == 0) ? (isPT && nonZeroA) : (pI == 3) ? (isPT && nonMaxA) : isPT);
                    const u32 invSlice = (u32)(__ballot(invalid) >> slice) & 0xffu;
                    const bool allInvalid = invSlice == 0xffu;
                    const bool needCheck = invSlice != 0;
                    for (int tw = 0; tw < 4; tw++)
                        for (int rf = 0; rf < numRefine; rf++)
                        {
                            const int t = (pI * 4 + tw) * numRefine + rf;
                            const bool run = scanActive && !allInvalid && tw < r.numTweak();
                            const float e = run ? (trialHbm ? trialHbm[unit * 16 * numRefine + t] : s_trialErr[unit * 16 * numRefine + t]) : FLT_MAX;
                            const bool better = run && e < held;
                            const bool anyBetter = ((u32)(__ballot(better) >> slice) & 0xffu) != 0;
                            bool commit = better;
                            if (needCheck)
                            {
                                // AndNot(punchThroughInvalid, better) = invalid & ~better (ParallelMath.h:900-905)
                                commit = run && invalid && !better;
                                if (((u32)(__ballot(commit) >> slice) & 0xffu) == 0)
                                    commit = false;
                            }
                            if (anyBetter && commit)
                            {
                                held = e;
                                last = t;
                            }
                        }
                }
                // the payload of the held trial: run that chain again up to its round
                const bool have = scanActive && last >= 0;
                const int hChain = have ? last / numRefine : 0, hRound = have ? last - hChain * numRefine : 0;
                int maxCount = have ? __popc(r.mask()) : 0;
#pragma unroll
                for (int step = 1; step < 64; step <<= 1)
                {
                    const int o = xorLane(maxCount, step);
                    maxCount = o > maxCount ? o : maxCount;
                }
                maxCount = __builtin_amdgcn_readfirstlane(maxCount);
                Unfinished uu;
#pragma unroll
                for (int ch = 0; ch < 4; ch++)
                {
                    uu.base[ch] = r.base[ch];
                    uu.offset[ch] = r.offset[ch];
                }
                ShapeBest b;
                const float uvs[4] = {r.vs[0], r.vs[1], r.vs[2], r.vs[3]};
                evalChain<4, FAST, true>(&s_pix[r.blk()][0], r.mask(), ((u64)r.listHi << 32) | r.listLo, maxCount, md, uu, hChain >> 2, hChain & 3, have, A, T, numRefine, b, uvs,
                                         true, nullptr, hRound);
                if (scanActive)
                {
                    if (!have)
                    {
                        b.err = FLT_MAX;
                        b.ep0 = b.ep1 = b.idxLo = b.idxHi = 0;
                    }
                    if (r.scErr < b.err)
                    {
                        b.err = r.scErr;
                        b.ep0 = b.ep1 = 0xff000000u;
                        b.idxLo = b.idxHi = 0;
                    }
                    u32 *dst = &s_res[r.slot()][0];
                    dst[0] = __builtin_bit_cast(u32, b.err);
                    dst[1] = b.ep0;
                    dst[2] = b.ep1;
                    dst[3] = b.idxLo;
                    dst[4] = b.idxHi;
                }
            }
            REFRESH_LANE();
            // ---- every offering block adds up the subsets of its items and commits ----
            for (int j = 0; j < 64; j++)
            {
                const bool mine = staged ? (iLo + j < iHi) : (j < myCount);
                if (__ballot(mine) == 0)
                    break;
                if (mine)
                {
                    const int item = staged ? iLo + j : (int)s_myItems[PT ? blk : 0][j];
                    const int partition = (int)(s_item[item] >> 8);
                    float totalError = 0.0f;
                    u32 pe[3][2] = {{0, 0}, {0, 0}, {0, 0}};
                    // the index sets come compact (member i of a subset in bits [4i, 4i + 4)): subset after subset, still
                    // compact -- expandIndexes sorts them to their pixels when the block is packed
                    u64 pIdx = 0;
                    int nibbles = 0;
                    for (int sub = 0; sub < numSubsets; sub++)
                    {
                        const u32 *src = &s_res[item * numSubsets + sub][0];
                        totalError = totalError + __builtin_bit_cast(float, src[0]);
                        const u32 e0 = src[1], e1 = src[2];
                        if (sub == 0) { pe[0][0] = e0; pe[0][1] = e1; }
                        else if (sub == 1) { pe[1][0] = e0; pe[1][1] = e1; }
                        else { pe[2][0] = e0; pe[2][1] = e1; }
                        const u64 part = ((u64)src[4] << 32) | src[3];
                        pIdx |= nibbles < 16 ? part << (4 * nibbles) : 0ull;
                        nibbles += __popc(subsetMaskOf(numSubsets, partition, sub));
                    }
                    const u32 pIdxLo = (u32)pIdx, pIdxHi = (u32)(pIdx >> 32);
                    const int seq = stage * 64 + partition;
                    bool mayTake = laneRuns;
                    if (ptStage && mode == 7 && anyBlockHasAlpha && !blockHasNonMaxAlpha && ((mode7RGB >> partition) & 1ull) == 0)
                        mayTake = false; // searched for the group's sake only (BC67.cpp:1625-1635)
                    PROF_STAGE_LANES(stageIter, 4, c == 0 && mayTake && (totalError < work.err || (totalError == work.err && seq < workSeq)))
                    if (mayTake && (totalError < work.err || (totalError == work.err && seq < workSeq)))
                    {
                        work.err = totalError;
                        workSeq = seq;
                        workPay0 = (c == 0) ? pe[0][0] : (c == 1) ? pe[1][0] : (c == 2) ? pe[2][0] : pIdxLo;
                        workPay1 = (c == 0) ? pe[0][1] : (c == 1) ? pe[1][1] : (c == 2) ? pe[2][1] : pIdxHi;
                    }
                }
            }
            if (!staged)
                break;
            chunkLo += itemCap;
            if (chunkLo >= phaseTotal)
                break;
            __syncthreads(); // the commits have read this chunk's items and results
            } // phases of the round
            PROF_MARK(14)
            if (flush)
                break;
        }
    }

#ifdef CVTT_BC7_PROFILE_SPLIT
    PROF_MARK(3)
#endif
    // ===================== fix-ups + bit packing (reference BC67.cpp:2003-2203) ==========
    REFRESH_LANE();
    {
        // the winner: seq = stage * 64 + partition for the single-plane modes (stages 0..5 = modes 0, 1, 2, 3, 6, 7),
        // 384 + rotation * 2 + index selector for mode 4, 392 + rotation for mode 5; nothing committed: mode 0, all zero
        WorkState packed;
        {
            const int sq = workSeq < 0 ? 0 : workSeq;
            const int cfg = sq - 384;
            const int stage = sq >> 6;
            packed.mode = (sq >= 392) ? 5 : (sq >= 384) ? 4 : (stage >= 4) ? stage + 2 : stage;
            packed.partOrIS = (sq >= 392) ? 0 : (sq >= 384) ? (cfg & 1) : (sq & 63);
            packed.rotation = (sq >= 392) ? cfg - 8 : (sq >= 384) ? (cfg >> 1) : 0;
            const int q = lane & ~3;
            packed.ep[0][0] = __shfl(workPay0, q);
            packed.ep[0][1] = __shfl(workPay1, q);
            packed.ep[1][0] = __shfl(workPay0, q | 1);
            packed.ep[1][1] = __shfl(workPay1, q | 1);
            packed.ep[2][0] = __shfl(workPay0, q | 2);
            packed.ep[2][1] = __shfl(workPay1, q | 2);
            packed.idxLo = __shfl(workPay0, q | 3);
            packed.idxHi = __shfl(workPay1, q | 3);
            packed.idx2Lo = packed.idx2Hi = 0;
            if (packed.mode < 4 || packed.mode == 7)
            {
                // single-plane results carry their indexes compact, subset after subset in member order: sort them to their
                // pixels (sub-lane c places pixels 4c .. 4c + 3; modes 4 / 5 are positional already, mode 6 is one subset of 16)
                const int p = packed.partOrIS;
                const bool three = (packed.mode == 0 || packed.mode == 2);
                const u32 m1 = three ? (s_pm3[p] & 0xffffu) : (u32)s_pm2[p];
                const u32 m2 = three ? (s_pm3[p] >> 16) : 0u;
                const u32 m0 = ~(m1 | m2) & 0xffffu;
                const int n0 = __popc(m0), n1 = __popc(m1);
                const u64 cat = ((u64)packed.idxHi << 32) | packed.idxLo;
                u64 pos = 0;
#pragma unroll
                for (int t = 0; t < 4; t++)
                {
                    const int px = 4 * c + t;
                    const u32 below = (1u << px) - 1u;
                    const bool in1 = (m1 >> px) & 1u, in2 = (m2 >> px) & 1u;
                    const int k = in2 ? n0 + n1 + __popc(m2 & below) : in1 ? n0 + __popc(m1 & below) : __popc(m0 & below);
                    pos |= ((cat >> (4 * k)) & 0xfull) << (4 * px);
                }
                u32 lo = (u32)pos, hi = (u32)(pos >> 32);
                lo |= xorLane(lo, 1);
                hi |= xorLane(hi, 1);
                lo |= xorLane(lo, 2);
                hi |= xorLane(hi, 2);
                packed.idxLo = lo;
                packed.idxHi = hi;
            }
        }
        const int mode = packed.mode;
        u32 w0, w1, w2, w3;
        if (mode == 4 || mode == 5)
        {
            packed.idx2Lo = s_parked[lane >> 2][0];
            packed.idx2Hi = s_parked[lane >> 2][1];
        }
        if (mode == 4)
            packDualPlane<4>(packed, w0, w1, w2, w3);
        else if (mode == 5)
            packDualPlane<5>(packed, w0, w1, w2, w3);
        else
        {
        // mode description bit-fields (BC7 format)
        const int numSubsetsTab[8] = {3, 2, 3, 2, 1, 1, 1, 2};
        const int partitionBitsTab[8] = {4, 6, 6, 6, 0, 0, 0, 6};
        const int rgbBitsTab[8] = {4, 6, 5, 7, 5, 7, 7, 5};
        const int alphaBitsTab[8] = {0, 0, 0, 0, 6, 8, 7, 5};
        const int indexBitsTab[8] = {3, 3, 2, 2, 2, 2, 4, 2};
        const int alphaIndexBitsTab[8] = {0, 0, 0, 0, 3, 2, 0, 0};
        const int pBitModeTab[8] = {0, 1, 2, 0, 2, 2, 0, 0}; // 0 per endpoint, 1 per subset, 2 none
        int numSubsets = 0, partitionBits = 0, rgbBits = 0, alphaBits = 0, indexBits = 0, alphaIndexBits = 0, pBitMode = 0;
#pragma unroll
        for (int m = 0; m < 8; m++)
            if (mode == m)
            {
                numSubsets = numSubsetsTab[m];
                partitionBits = partitionBitsTab[m];
                rgbBits = rgbBitsTab[m];
                alphaBits = alphaBitsTab[m];
                indexBits = indexBitsTab[m];
                alphaIndexBits = alphaIndexBitsTab[m];
                pBitMode = pBitModeTab[m];
            }
        constexpr bool separateAlpha = false; // modes 4 and 5 have gone to packDualPlane; their branches below fold away
        const bool combinedAlpha = (mode == 6 || mode == 7);
        const int partition = packed.partOrIS;
        const int indexSelector = packed.partOrIS;

        u64 idx = ((u64)packed.idxHi << 32) | packed.idxLo;
        u64 idx2 = ((u64)packed.idx2Hi << 32) | packed.idx2Lo;
        u32 ep[3][2];
#pragma unroll
        for (int s = 0; s < 3; s++)
        {
            ep[s][0] = packed.ep[s][0];
            ep[s][1] = packed.ep[s][1];
        }
        const u64 ones = 0x1111111111111111ull;
        int fix1 = 0, fix2 = 0;

        if (separateAlpha)
        {
            bool flipRGB = ((idx >> (indexBits - 1)) & 1ull) != 0;
            bool flipAlpha = ((idx2 >> (alphaIndexBits - 1)) & 1ull) != 0;
            if (flipRGB)
                idx = ones * (u64)((1 << indexBits) - 1) - idx;
            if (flipAlpha)
                idx2 = ones * (u64)((1 << alphaIndexBits) - 1) - idx2;
            if (indexSelector)
            {
                const bool t = flipRGB;
                flipRGB = flipAlpha;
                flipAlpha = t;
            }
            if (flipRGB)
            {
                const u32 a = ep[0][0], b = ep[0][1];
                ep[0][0] = (a & 0xff000000u) | (b & 0x00ffffffu);
                ep[0][1] = (b & 0xff000000u) | (a & 0x00ffffffu);
            }
            if (flipAlpha)
            {
                const u32 a = ep[0][0], b = ep[0][1];
                ep[0][0] = (a & 0x00ffffffu) | (b & 0xff000000u);
                ep[0][1] = (b & 0x00ffffffu) | (a & 0xff000000u);
            }
        }
        else
        {
            u32 subsetMap = 0; // 2 bits per pixel
            if (numSubsets == 2)
            {
                fix1 = T->anchor2[partition & 63];
                const u32 bits = T->partition2[partition & 63];
#pragma unroll
                for (int px = 0; px < 16; px++)
                    subsetMap |= ((bits >> px) & 1u) << (2 * px);
            }
            else if (numSubsets == 3)
            {
                fix1 = T->anchor3[partition & 63][0];
                fix2 = T->anchor3[partition & 63][1];
                subsetMap = T->partition3[partition & 63];
            }
            const int hiBit = indexBits - 1;
            bool flip[3];
            flip[0] = ((idx >> hiBit) & 1ull) != 0;
            flip[1] = (numSubsets >= 2) && (((idx >> (4 * fix1 + hiBit)) & 1ull) != 0);
            flip[2] = (numSubsets >= 3) && (((idx >> (4 * fix2 + hiBit)) & 1ull) != 0);
            u64 flipMask = 0;
#pragma unroll
            for (int px = 0; px < 16; px++)
            {
                const int s = (subsetMap >> (2 * px)) & 3;
                const bool f = (s == 0) ? flip[0] : ((s == 1) ? flip[1] : flip[2]);
                if (f)
                    flipMask |= 0xfull << (4 * px);
            }
            const u64 inverted = ones * (u64)((1 << indexBits) - 1) - idx;
            idx = (idx & ~flipMask) | (inverted & flipMask);
            const u32 chMask = combinedAlpha ? 0xffffffffu : 0x00ffffffu;
#pragma unroll
            for (int s = 0; s < 3; s++)
                if (flip[s])
                {
                    const u32 a = ep[s][0], b = ep[s][1];
                    ep[s][0] = (a & ~chMask) | (b & chMask);
                    ep[s][1] = (b & ~chMask) | (a & chMask);
                }
        }

        // ---- bit packing: the (up to 66) fields of the block are dealt round-robin to the four
        // lanes of the quad, each lane shifts its fields to their absolute bit positions, and
        // the quad ORs the four partial blocks together ----
        u64 lo = 0, hi = 0;
        auto put = [&](u32 value, int off, int bits) {
            const u64 v = (u64)value;
            (void)bits;
            if (off < 64)
            {
                lo |= v << off;
                if (off > 56)
                    hi |= v >> (64 - off); // fields are at most 8 bits wide
            }
            else
                hi |= v << (off - 64);
        };
        // header
        int epBase = mode + 1;
        if (c == 0)
        {
            put(1u << mode, 0, mode + 1);
            if (partitionBits)
                put((u32)partition, epBase, partitionBits);
        }
        epBase += partitionBits;
        if (separateAlpha)
        {
            if (c == 1)
            {
                put((u32)packed.rotation, epBase, 2);
                if (mode == 4)
                    put((u32)indexSelector, epBase + 2, 1);
            }
            epBase += (mode == 4) ? 3 : 2;
        }
        // endpoints: channel-major, then subset, then endpoint 0/1; then alpha; then p-bits
        const int perChannel = 2 * numSubsets;
        const int numRGB = 3 * perChannel;
        const int numAlpha = alphaBits ? perChannel : 0;
        const int numPBits = (pBitMode == 0) ? perChannel : (pBitMode == 1) ? numSubsets : 0;
        const int alphaBase = epBase + numRGB * rgbBits;
        const int pBase = alphaBase + numAlpha * alphaBits;
        const int idxBase = pBase + numPBits;
        for (int i = 0; i < 8; i++)
        {
            const int f = 4 * i + c;
            if (__ballot(f < numRGB + numAlpha + numPBits) == 0)
                break;
            if (f < numRGB + numAlpha)
            {
                // colour or alpha endpoint field
                const bool isAlpha = f >= numRGB;
                const int g = isAlpha ? f - numRGB : f;
                const int ch = isAlpha ? 3 : ((numSubsets == 1) ? (g >> 1) : (numSubsets == 2) ? (g >> 2) : (g / 6));
                const int rem = isAlpha ? g : g - ch * perChannel;
                const int sub = rem >> 1, e = rem & 1;
                const u32 e0 = (sub == 0) ? ep[0][0] : (sub == 1) ? ep[1][0] : ep[2][0];
                const u32 e1 = (sub == 0) ? ep[0][1] : (sub == 1) ? ep[1][1] : ep[2][1];
                const u32 val = ((e ? e1 : e0) >> (8 * ch)) & 0xffu;
                const int bits = isAlpha ? alphaBits : rgbBits;
                put(val >> (8 - bits), isAlpha ? alphaBase + g * alphaBits : epBase + g * rgbBits, bits);
            }
            else if (f < numRGB + numAlpha + numPBits)
            {
                const int g = f - numRGB - numAlpha;
                const int sub = (pBitMode == 1) ? g : (g >> 1);
                const int e = (pBitMode == 1) ? 0 : (g & 1);
                const u32 e0 = (sub == 0) ? ep[0][0] : (sub == 1) ? ep[1][0] : ep[2][0];
                const u32 e1 = (sub == 0) ? ep[0][1] : (sub == 1) ? ep[1][1] : ep[2][1];
                put((((e ? e1 : e0) & 0xffu) >> (7 - rgbBits)) & 1u, pBase + g, 1);
            }
        }
        // indexes: pixel px at idxBase + px*indexBits minus one bit per anchor before it
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int px = 4 * i + c;
            const bool anchor = (px == 0 || px == fix1 || px == fix2);
            const int before = (px > 0 ? 1 : 0) + ((fix1 != 0 && px > fix1) ? 1 : 0) + ((fix2 != 0 && px > fix2) ? 1 : 0);
            put((u32)((idx >> (4 * px)) & 0xfull), idxBase + px * indexBits - before, indexBits - (anchor ? 1 : 0));
        }
        if (separateAlpha)
        {
            const int idx2Base = idxBase + 16 * indexBits - 1;
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const int px = 4 * i + c;
                put((u32)((idx2 >> (4 * px)) & 0xfull), idx2Base + px * alphaIndexBits - (px > 0 ? 1 : 0), alphaIndexBits - (px == 0 ? 1 : 0));
            }
        }
        w0 = (u32)lo;
        w1 = (u32)(lo >> 32);
        w2 = (u32)hi;
        w3 = (u32)(hi >> 32);
#pragma unroll
        for (int step = 1; step <= 2; step <<= 1)
        {
            w0 |= xorLane(w0, step);
            w1 |= xorLane(w1, step);
            w2 |= xorLane(w2, step);
            w3 |= xorLane(w3, step);
        }
        }

        if (HARD)
        {
            if (valid && c == 0)
            {
                // nothing committed: no candidate from this slice
                const bool improved = (work.err != hardErr) || (workSeq != hardSeq);
                CvttBc7HardCand cand;
                cand.packed[0] = w0;
                cand.packed[1] = w1;
                cand.packed[2] = w2;
                cand.packed[3] = w3;
                cand.err = improved ? work.err : FLT_MAX;
                cand.seq = workSeq;
                cand.pad[0] = cand.pad[1] = 0;
                A.hardCand[vBlock] = cand;
            }
        }
        else if (valid && c == 0)
        {
            uint4 o;
            o.x = w0;
            o.y = w1;
            o.z = w2;
            o.w = w3;
            *reinterpret_cast<uint4 *>(out + (size_t)blockIndex * 16u) = o;
            if (!PT && A.hardCap != 0)
            {
                const u32 slot = s_blkFlags[blk] >> 8;
                if (slot != 0)
                {
                    CvttBc7HardRec &rec = A.hardRec[slot - 1u];
                    rec.blockIndex = blockIndex;
                    rec.err = work.err;
                    rec.seq = workSeq;
                }
            }
        }
    }
    PROF_MARK(5)
    PROF_FLUSH
#undef REFRESH_LANE
#undef valid
#undef anyBlockHasAlpha
#undef allowRGBModes
#undef allowMode7
#undef blockHasNonMaxAlpha
}

template <bool FAST, bool PT, bool HARD>
// (the punch-through instantiation holds 22 KB of LDS: two waves per SIMD is all that fits, so it may use their registers)
__global__ __launch_bounds__(64, PT ? 3 : (FAST || HARD) ? CVTT_BC7_WAVES : CVTT_BC7_WAVES_SLOW) void cvttmi_bc7_kernel(const uint8_t *__restrict__ blocks, uint8_t *__restrict__ out,
                                                        const CvttBc7Args A, const CvttDeviceTables *__restrict__ T,
                                                        const CvttBc7DevicePlan *__restrict__ dplan)
{
    if constexpr (!HARD)
    {
        // the counter of the NEXT encode's hand-over list (the two alternate, shim.cpp): nothing that read it is still running
        // (launches that use the list are ordered, orderAfterPrevious), and nothing of this encode touches it -- no memset launch
        if (!PT && A.hardCap != 0 && blockIdx.x == 0 && threadIdx.x == 0)
            *A.hardCountNext = 0u;
        bc7Body<FAST, PT, HARD>(blocks, out, A, T, dplan, blockIdx.x, gridDim.x);
    }
    else
    {
        // The second launch is a FIXED grid (cvttmi_launch_bc7: at most 4 096 waves) whose waves walk the items -- kHardWaves per
        // recorded block -- in strides of the grid.  On content that hands nothing over (RGBA noise) it used to be 16 waves per
        // SLOT, every one dispatched only to read the count and leave: 24 us of a 1.5 ms encode.
        const u32 count = *A.hardCount;
        const u32 total = (count < A.hardCap ? count : A.hardCap) * (u32)kHardWaves;
        for (u32 item = blockIdx.x; item < total; item += gridDim.x)
        {
            bc7Body<FAST, PT, HARD>(blocks, out, A, T, dplan, item, total);
            __syncthreads(); // the next item reuses the LDS of this one
        }
    }
}

// Third launch: the winner among the recorded best of a handed-over block and the candidates of its partition slices.
__global__ __launch_bounds__(64) void cvttmi_bc7_hard_commit_kernel(uint8_t *__restrict__ out, const CvttBc7Args A)
{
    const u32 h = blockIdx.x * 64u + threadIdx.x;
    const u32 count = *A.hardCount;
    if (h >= (count < A.hardCap ? count : A.hardCap))
        return;
    const CvttBc7HardRec rec = A.hardRec[h];
    float bestErr = rec.err;
    int bestSeq = rec.seq, best = -1;
    for (int q = 0; q < kHardWaves; q++)
    {
        const CvttBc7HardCand &cand = A.hardCand[h * kHardWaves + q];
        const float e = cand.err;
        const int sq = cand.seq;
        if (e == FLT_MAX)
            continue; // nothing from this slice
        if (e < bestErr || (e == bestErr && sq < bestSeq))
        {
            bestErr = e;
            bestSeq = sq;
            best = q;
        }
    }
    if (best >= 0)
    {
        const CvttBc7HardCand &cand = A.hardCand[h * kHardWaves + best];
        *reinterpret_cast<uint4 *>(out + (size_t)rec.blockIndex * 16u) = make_uint4(cand.packed[0], cand.packed[1], cand.packed[2], cand.packed[3]);
    }
}

extern "C" hipError_t cvttmi_launch_bc7(const void *d_blocks, void *d_out, const CvttBc7Args *args,
                                        const CvttDeviceTables *d_tables, const CvttBc7DevicePlan *d_plan,
                                        hipStream_t stream)
{
    const uint32_t waves = (args->numBlocks + 15u) / 16u;
    if (waves == 0)
        return hipSuccess;
    const bool fast = (args->flags & CVTTMI_FLAG_BC7_FAST_INDEXING) != 0;
    const bool pt = (args->flags & CVTTMI_FLAG_BC7_RESPECT_PUNCHTHROUGH) != 0;
    // developer knob: extra (unused) dynamic LDS per workgroup, to pin the number of resident waves per SIMD in experiments
    static const unsigned ldsPad = getenv("CVTTMI_BC7_LDS_PAD") ? (unsigned)atoi(getenv("CVTTMI_BC7_LDS_PAD")) : 0u;
#define CVTT_LAUNCH(F, P, H, GRID) hipLaunchKernelGGL((cvttmi_bc7_kernel<F, P, H>), dim3(GRID), dim3(64), (H) ? 0u : ldsPad, stream, (const uint8_t *)d_blocks, (uint8_t *)d_out, *args, d_tables, d_plan)
    if (pt)
    {
        if (fast) CVTT_LAUNCH(true, true, false, waves); else CVTT_LAUNCH(false, true, false, waves);
        return hipGetLastError();
    }
    const bool split = args->hardCap != 0; // (the hand-over counter is zero: the previous encode's first launch cleared it, the context's creation the first two)
    if (fast) CVTT_LAUNCH(true, false, false, waves); else CVTT_LAUNCH(false, false, false, waves);
    if (split)
    {
        // a fixed grid that walks the items (kHardWaves per recorded block): one resident generation at most
        const uint32_t hardItems = args->hardCap * (uint32_t)kHardWaves;
        const uint32_t hardGrid = hardItems < 4096u ? hardItems : 4096u;
        if (fast) CVTT_LAUNCH(true, false, true, hardGrid); else CVTT_LAUNCH(false, false, true, hardGrid);
        hipLaunchKernelGGL(cvttmi_bc7_hard_commit_kernel, dim3((args->hardCap + 63u) / 64u), dim3(64), 0, stream, (uint8_t *)d_out, *args);
    }
#undef CVTT_LAUNCH
    return hipGetLastError();
}
