// BC6H (unsigned / signed half-float) endpoint search for gfx950.
//
// Replaces cvtt::Internal::BC6HComputer::Pack as reached from cvtt::Kernels::EncodeBC6HU /
// EncodeBC6HS (reference ConvectionKernels_API.cpp:56-84, ConvectionKernels_BC67.cpp:2665-3051;
// QuantizeEndpoints* 2503-2595, Evaluate*Legality 2597-2663, quantise/unquantise 2425-2501,
// IndexSelectorHDR.h, ParallelMath.h:996-1066).  Bit-identical to the reference's SSE2 lanes
// in its canonical build (zero-initialised automatics, program-order rounding
// scopes).
//
// Mapping (round 6): a lane QUAD owns one block, sub-lane t of the quad runs the chain of seed point ("tweak") t -- its up to
// three refine rounds -- so a wave owns 16 blocks = two reference groups, and the four chains a subset has per block advance
// side by side instead of one after the other.  (Rounds 1-5: one lane per block, 64 blocks per wave, 236 registers and 18 KB
// of LDS per wave = two waves per SIMD, VALU issue at 0.50 of its peak.)
//   * the pixels of the 16 blocks live in LDS ([pixel][block]: the quad's four lanes read one address, the 16 blocks
//     consecutive words): 2CL integers and their linear values (TwosCLHalfToFloat, BC67.cpp:2711), 5 KB; nothing of a block
//     occupies registers, so a lane needs about a hundred and the wave 10 KB of LDS: four waves per SIMD;
//   * the history of a partition's 2 x 12 "meta rounds" (quantised end points, errors) is in LDS too, [subset][round][block],
//     each lane writing the rounds of its own chain: the commit loop (run redundantly by the four lanes of a quad, which keeps
//     the block's running best in all four) and the legality pass read it from there;
//   * the reference couples the 8 lanes of a group in two places -- the duplicate-round skip (BC67.cpp:2853-2877, AllSet) and the
//     mode-commit loop (2936-2984, AnySet); both are 32-lane slices of a wave ballot (8 blocks x 4 identical sub-lanes).
//   * THE DUPLICATE TEST AND THE PARALLEL CHAINS.  Round (t, r) is dropped when in all eight blocks of the group its quantised end
//     points repeat those of an EARLIER meta round (t' < t with any r', or t' = t with r' < r), and a dropped round leaves the
//     refiner of (t, r + 1) empty.  With the four chains in lock-step over r, the rounds (t' < t, r' <= r) and (t, r' < r) are
//     known when round (t, r) has quantised its end points (the neighbours' words come over DPP quad_perm): a group-wide match
//     among those is a drop, decided where the reference decides it.  What is not known yet are the rounds (t' < t, r' > r).  They
//     are compared after the last pass; a group-wide match that only they produce (the first such round in the reference's
//     order is certain: everything before it is final) is recorded as a forced drop and the subset's three passes run again --
//     which noise does about never (8e-4 per partition search) and content whose chains converge to the same end points often
//     (smooth ramps 0.15, a narrow value range 0.92 per search: tools/bc6h_rerun_stats.py, profiles/r06/ab_bc6h.txt).  The
//     state this converges to is the reference's: by induction over the meta-round
//     order every round's end points follow from its chain's earlier rounds and their validity, every validity from the end
//     points of the earlier rounds.
//   * THE COMMIT.  With one mode at a precision no block looks at its group mates: the loop leaves the first legal pair of rounds
//     with the smallest combined error, found by the quad (sub-lane t: the pairs of its chain's subset-0 rounds).  With three
//     modes (8, 11 bits) the loop commits a block at every legal mode up to E = max over S of (first legal mode), S = the mates
//     that beat their best at that pair; the pair is found the same way and the mode from the mates at that pair, and the wave
//     walks the pairs like the reference only when a mate's membership of S depends on the pairs before (see the commit code).
//   * slow indexing: the weighted linear colours of the 8 / 16 interpolants of a round sit in registers and every pixel scans
//     them in order (strict '<', IndexSelectorHDR.h:125-139); the anchor pixel goes first because its index decides the
//     endpoint inversion; indexes are not kept (the winner's are selected again, once, after the search).
//
// What is NOT searched: nine of the ten two-subset modes delta-code three of their four end points, the
// reference's commit loop skips a (round, round, mode) triple whose deltas do not fit with a `continue` that changes no state
// (BC67.cpp:2954-2955), and whether a triple fits depends on the rounds' quantised end points alone -- known before a round
// looks at a pixel.  So
//   * a round whose end points fit no mode in any lane of the wave runs without errors (`needError`), and not at all in the
//     last refine pass; a partition without a usable subset-0 round skips subset 1 and the commit loop (`usable0`) -- with 16
//     blocks per wave that is decided for a quarter of the blocks it used to be decided for;
//   * precisions of 8 bits and more are searched lazily (`lazy`): chains of both subsets without errors, then the exact set of
//     round pairs some block could commit -- per block for the single-mode precisions, per GROUP for the three-mode ones,
//     whose mode loop couples the group -- then a replay of just those rounds with errors, each by the lane whose chain it is.
// The output is bit-identical because every skipped piece is one the reference computes and then cannot use.
#include "cvtt_kernel_common.h"
#include <hip/hip_fp16.h>
#include <type_traits>

// Developer-only counters (-DCVTT_BC6H_PROFILE): [0] partition searches (per wave), [1] subset passes run again after a late
// duplicate, [2] lazy partitions, [3] lazy partitions with a replay, [4] replayed round slots, [5] eager partitions,
// [6] several-mode commits worked out without the pair loop, [7] with it
#ifdef CVTT_BC6H_PROFILE
__device__ unsigned long long g_bc6hProf[32];
#define PROF_COUNT(slot, n) { if (threadIdx.x == 0) atomicAdd(&g_bc6hProf[slot], (unsigned long long)(n)); }
extern "C" int cvttmi_bc6h_prof_read(unsigned long long *out)
{
    unsigned long long zero[32] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bc6hProf), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bc6hProf), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#else
#define PROF_COUNT(slot, n) {}
#endif
// Developer-only trace (-DCVTT_BC6H_TRACE=<block index>): the round history of that block at 6 bits, per partition:
// roundValid0, roundValid1, then [subset][round]{word a, word b, error bits} (tools/bc6h_trace.py compares it with the oracle's)
#ifdef CVTT_BC6H_TRACE
__device__ unsigned g_bc6hTrace[32 * 74];
__device__ unsigned long long g_bc6hTrace2[16];
extern "C" int cvttmi_bc6h_trace2_read(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bc6hTrace2), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
extern "C" int cvttmi_bc6h_trace_read(unsigned *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bc6hTrace), sizeof(unsigned) * 32 * 74) == hipSuccess ? 0 : -1;
}
#endif

namespace
{
// Waves per SIMD the register allocator may assume (128 registers; the LDS footprint of 10 KB per wave allows four).
#ifndef CVTT_BC6H_WAVES
#define CVTT_BC6H_WAVES 4
#endif
// Lowest two-subset precision that is searched lazily (chains without errors, then only the rounds of pairs that can be
// committed are evaluated).  8: where a round fits its own delta with probability 2^-8 or less on content without structure.
// 6 is never lazy: its mode stores no deltas.
#ifndef CVTT_BC6H_LAZY_MIN
#define CVTT_BC6H_LAZY_MIN 8
#endif
// a partition whose replay needs more than this many round slots switches the rest of the precision to the eager search
#ifndef CVTT_BC6H_LAZY_SWITCH
#define CVTT_BC6H_LAZY_SWITCH 2
#endif

// ceil(n / 31) for 0 <= n < 2^23: one v_mul_hi_u32.  2216757579 = (2^36 + 8213) / 31 (NOT ceil(2^36 / 31) = 2216757315):
// x * 2216757579 / 2^36 = x / 31 + x * 8213 / (31 * 2^36), and x / 31 lies at least 1 / 31 below the next integer, so the
// floor is that of x / 31 while x * 8213 < 2^36, i.e. for x < 8.3 million (tools/check_bc6h_quantize.py compares every value)
__device__ __forceinline__ int ceilDiv31(int n)
{
    return (int)(__umulhi((u32)(n + 30), 2216757579u) >> 4);
}

// QuantizeSingleEndpointElementUnsigned, BC67.cpp:2441-2445.  The reference divides elem * 64 by 31 in binary32 rounded
// up (RoundUpForScope, BC67.cpp:2556), subtracts 32768, takes the ceiling and re-biases.  For every elem the colour-space
// clamp lets through (0 ... 31743) that is the integer ceil(elem * 64 / 31): the quotient rounded up never passes the next
// integer (integers below 2^24 are representable), and its distance from the integer below is at least 1/31, far above
// half an ulp of the difference -- checked exhaustively against the float sequence by tools/check_bc6h_quantize.py.
__device__ __forceinline__ int quantizeUnsigned(int elem, int precision)
{
    return ceilDiv31(elem * 64) >> (16 - precision);
}

// QuantizeSingleEndpointElementSigned, BC67.cpp:2425-2439: ceil(|elem| * 32 / 31), by the same argument (|elem| <= 31743)
__device__ __forceinline__ int quantizeSigned(int elem, int precision)
{
    const bool neg = elem < 0;
    int a = neg ? -elem : elem;
    int i = ceilDiv31(a * 32);
    i = i > 32767 ? 32767 : i;
    a = (int)(((u32)i & 0xffffu) >> (16 - precision));
    return neg ? -a : a;
}

// Unquantize*, BC67.cpp:2447-2501: returns the interpolation endpoint, `finished` = colour-space value
__device__ __forceinline__ int unquantizeUnsigned(int comp, int precision, int &finished)
{
    u32 unq = (u32)comp & 0xffffu;
    if (precision < 15)
    {
        unq = (((u32)comp << (16 - precision)) + (0x8000u >> precision)) & 0xffffu;
        if (comp == 0) unq = 0;
        if (((1 << precision) - 2) < comp) unq = 0xffffu;
    }
    finished = (int)((unq * 31u) >> 6);
    return (int)unq;
}

__device__ __forceinline__ int unquantizeSigned(int comp, int precision, int &finished)
{
    const bool neg = comp < 0;
    const int absComp = neg ? -comp : comp;
    int unq, absUnq;
    if (precision >= 16)
    {
        unq = comp;
        absUnq = absComp;
    }
    else
    {
        absUnq = (int)(short)(unsigned short)((absComp << (16 - precision)) + (0x4000 >> (precision - 1)));
        if (comp == 0) absUnq = 0;
        if (((1 << (precision - 1)) - 2) < comp) absUnq = 0x7fff;
        unq = neg ? -absUnq : absUnq;
    }
    int funq = (int)((((u32)absUnq & 0xffffu) * 31u) >> 5);
    funq = funq > 32767 ? 32767 : funq;
    finished = (int)(short)(neg ? -funq : funq);
    return (int)(short)unq;
}

// TwosCLHalfToFloat, ParallelMath.h:1012-1041 -- literal bit manipulation (needed for the
// signed format, whose negative 2CL pixel values are not valid half patterns)
__device__ __forceinline__ float twosCLHalfToFloatBits(int v16)
{
    const u32 v = (u32)v16 & 0xffffu;
    const u32 signBits = v & 0x8000u;
    const u32 mantissa = v & 0x03ffu;
    u32 exponent = v & 0x7c00u;
    const bool isDenormal = exponent == 0;
    exponent = ((exponent >> 3) + 14336u) & 0xffffu;
    const u32 corrHigh = isDenormal ? (signBits | 14336u) : 0u;
    const u32 highBits = signBits | exponent | (mantissa >> 3);
    const u32 lowBits = (mantissa << 13) & 0xffffu;
    return __uint_as_float((highBits << 16) | lowBits) - __uint_as_float(corrHigh << 16);
}

// For every finite non-negative half pattern (all the unsigned format ever sees: inputs are
// clamped to [0, 0x7BFF], reconstructions to <= 31743) the function above equals the hardware
// conversion, except that it halves denormals (exponent field 0) -- checked exhaustively.
template <bool SIGNED>
__device__ __forceinline__ float twosCLHalfToFloat(int v16)
{
    if (SIGNED)
        return twosCLHalfToFloatBits(v16);
    const float f = __half2float(__ushort_as_half((unsigned short)v16));
    return (v16 & 0x7c00) ? f : f * 0.5f;
}

// ReconstructHDR{Signed,Unsigned}Uninverted for one channel, IndexSelectorHDR.h:34-66
// (64 - w) * e0 + w * e1 = 64 * e0 + w * (e1 - e0): one 24-bit multiply-add per interpolant instead of two 32-bit
// multiplications (quarter rate on gfx950) -- |e| < 2^16, w <= 64, so every term fits 24 bits
// reconstructBase: 64 * e0 + 32, with the difference e1 - e0 computed once per round and channel
__device__ __forceinline__ u32 mulU24(u32 a, u32 b)
{
    u32 r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// DYN: the weight is a run-time value (forced v_mad_i32_i24); otherwise a literal the compiler may fold
template <bool SIGNED, bool DYN = true>
__device__ __forceinline__ int reconstructFrom(int base, int diff, int weight)
{
    if (SIGNED)
    {
        int p = (DYN ? madI24(weight, diff, base) : mad24(weight, diff, base)) >> 6;
        p = p > 32767 ? 32767 : (p < -32768 ? -32768 : p);
        const bool neg = p < 0;
        const int a = neg ? -p : p;
        int scaled = (int)(__umul24((u32)a & 0xffffu, 31u) >> 5); // UnscaleHDRValueSigned, BC67.cpp:766-782
        scaled = scaled > 32767 ? 32767 : scaled;
        return (int)(short)(unsigned short)((u32)scaled | (neg ? 0x8000u : 0u));
    }
    else
    {
        // e < 2^16, weight <= 64: 0 <= 64 e0 + w (e1 - e0) + 32 < 2^22 + 32, so ">> 6 & 0xffff" is one v_bfe_u32 (and the mask never bites)
        const u32 p = __builtin_amdgcn_ubfe((u32)(DYN ? madI24(weight, diff, base) : mad24(weight, diff, base)), 6u, 16u);
        return (int)(mulU24(p, 31u) >> 6);   // UnscaleHDRValueUnsigned
    }
}
template <bool SIGNED>
__device__ __forceinline__ int reconstructChannel(int e0, int e1, int weight)
{
    return reconstructFrom<SIGNED, false>(e0 * 64 + 32, e1 - e0, weight);
}

// the 32-lane slice of a wave ballot that belongs to the lane's reference group (8 blocks x 4 sub-lanes)
__device__ __forceinline__ u32 groupBits(u64 ballot, int lane) { return (u32)(ballot >> (lane & 32)); }
// the value sub-lane Q of the lane's quad holds: v_mov_b32 with a DPP quad_perm (no LDS traffic, no address register)
template <int Q>
__device__ __forceinline__ u32 quadBcast(u32 v)
{
    return (u32)__builtin_amdgcn_mov_dpp((int)v, Q * 0x55, 0xf, 0xf, true);
}
// meta round m = 3 * tweak + refine pass: its tweak (m = 0 ... 11)
__device__ __forceinline__ int tweakOf(int m) { return (m * 11) >> 5; }
} // namespace

template <bool SIGNED, bool FAST>
__global__ __launch_bounds__(64, CVTT_BC6H_WAVES) void cvttmi_bc6h_kernel(const uint8_t *__restrict__ blocks, uint8_t *__restrict__ out,
                                                                         const CvttBc6hArgs A, const CvttDeviceTables *__restrict__ T)
{
    // ---- everything a wave knows about its 16 blocks and the partition it is searching: 10 KB of LDS ----
    // [pixel][block]: .w = red | green << 16 (2CL integers); .xyz = unsigned format: the three PRE-WEIGHTED values ((float)value *
    // channel weight: what the PCA and the refiner add up -- the linear value is one v_cvt_f32_f16 away); signed format: the three
    // LINEAR values (TwosCLHalfToFloat of a negative 2CL value is ten instructions, the pre-weighted value three)
    __shared__ float4 s_px[16][16];
    __shared__ u32 s_pk2[16][16];          // [pixel][block]: blue (2CL)
    __shared__ u32 s_epq[2][12][2][16];    // [subset][meta round][word][block]: the quantised end points of a round (layout: packEPQ)
    __shared__ float s_err[2][12][16];     // [subset][meta round][block]
    __shared__ uint8_t s_xb[2][16][4];     // [subset][block][tweak]: bits 2r, 2r+1 = the two spare bits of round (tweak, r) (two-subset form)
    __shared__ uint8_t s_inv[2][16][4];    // [subset][block][tweak]: bit r = round (tweak, r) exchanged its end points (anchor index in the upper half)

    // (not const: REFRESH_LANE() hands the optimiser the same values as "new" ones at the top of every partition, so that the lane
    // masks (tw > 0, tw == 1, ...) and LDS addresses it derives from them are computed where they are used instead of being
    // hoisted in front of the whole search and kept -- i.e. spilled -- across it; the lane number comes from v_mbcnt, a
    // workgroup being one wave)
    int lane = threadIdx.x;
    int blk = lane >> 2;
    int tw = lane & 3;
#define REFRESH_LANE() do { lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); asm volatile("" : "+v"(lane)); blk = lane >> 2; tw = lane & 3; } while (0)
    const u32 blockIndex = blockIdx.x * 16u + (u32)blk;
    const bool valid = blockIndex < A.numBlocks;

    // Does any pixel value of the wave have a zero exponent field (the patterns TwosCLHalfToFloat halves)?  Almost never, and then
    // the three fix-ups per pixel are skipped (unsigned format; wave-uniform)
    bool pixelFixup = true;
    // ---- load + clamp to the "2CL" domain (BC67.cpp:2691-2715); sub-lane t takes pixels t, t + 4, t + 8, t + 12 ----
    {
        const uint2 *src = reinterpret_cast<const uint2 *>(blocks + (size_t)(valid ? blockIndex : 0u) * 128u);
        bool zeroExp = false;
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const int px = 4 * j + tw;
            const uint2 raw = src[px];
            int v[3] = {(int)(short)(raw.x & 0xffffu), (int)(short)(raw.x >> 16), (int)(short)(raw.y & 0xffffu)};
            float l[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                int x = v[ch];
                if (SIGNED)
                {
                    if (x < 0)
                        x = -(x & 32767);
                    x = x < -31743 ? -31743 : x;
                }
                else
                    x = x < 0 ? 0 : x;
                x = x > 31743 ? 31743 : x;
                v[ch] = x;
                l[ch] = SIGNED ? twosCLHalfToFloat<SIGNED>(x) : (float)x * A.w[ch];
                zeroExp = zeroExp | ((x & 0x7c00) == 0);
            }
            s_px[px][blk] = make_float4(l[0], l[1], l[2], __uint_as_float(((u32)v[0] & 0xffffu) | ((u32)v[1] << 16)));
            s_pk2[px][blk] = (u32)v[2] & 0xffffu;
        }
        pixelFixup = __ballot(zeroExp) != 0;
    }
    __syncthreads();
    // pixel px of the lane's block: a = red | green << 16, b = blue (2CL integers), lf = the three linear values (BC67.cpp:2711),
    // pw = the three pre-weighted values (BCCommon PreWeightPixelsHDR)
    auto pixLoad = [&](int px, u32 &a, u32 &b, float (&lf)[3], float (&pw)[3]) {
        const float4 v = s_px[px][blk];
        a = __float_as_uint(v.w);
        b = s_pk2[px][blk];
        if (SIGNED)
        {
            lf[0] = v.x;
            lf[1] = v.y;
            lf[2] = v.z;
            pw[0] = (float)(int)(short)(a & 0xffffu) * A.w[0];
            pw[1] = (float)(int)(short)(a >> 16) * A.w[1];
            pw[2] = (float)(int)(short)(b & 0xffffu) * A.w[2];
        }
        else
        {
            pw[0] = v.x;
            pw[1] = v.y;
            pw[2] = v.z;
            const u32 c[3] = {a & 0xffffu, a >> 16, b & 0xffffu};
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                lf[ch] = __half2float(__ushort_as_half((unsigned short)c[ch]));
            if (pixelFixup)
            {
                asm volatile("" ::: "memory"); // keep this ONE branch: as selects it costs what it is meant to save
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    lf[ch] = (c[ch] & 0x7c00u) ? lf[ch] : lf[ch] * 0.5f;
            }
        }
    };
    // ... the pre-weighted values alone (PCA)
    auto pixPW = [&](int px, float (&pw)[3]) {
        const float4 v = s_px[px][blk];
        if (SIGNED)
        {
            const u32 a = __float_as_uint(v.w), b = s_pk2[px][blk];
            pw[0] = (float)(int)(short)(a & 0xffffu) * A.w[0];
            pw[1] = (float)(int)(short)(a >> 16) * A.w[1];
            pw[2] = (float)(int)(short)(b & 0xffffu) * A.w[2];
        }
        else
        {
            pw[0] = v.x;
            pw[1] = v.y;
            pw[2] = v.z;
        }
    };

    const int numTweakRounds = A.seedPoints < 1 ? 1 : (A.seedPoints > 4 ? 4 : A.seedPoints);
    const int numRefineRounds = A.refineRounds < 1 ? 1 : (A.refineRounds > 3 ? 3 : A.refineRounds);
    bool twActive = tw < numTweakRounds;
    // the meta rounds the option values leave out (BC67.cpp:2819-2827): never valid, their end points stay the zeros of the
    // canonical build's fresh automatics -- which later rounds are still compared with
    u32 abortMask = 0;
    for (int t = 0; t < 4; t++)
        for (int r = 0; r < 3; r++)
            if (t >= numTweakRounds || r >= numRefineRounds)
                abortMask |= 1u << (3 * t + r);
    abortMask = (u32)__builtin_amdgcn_readfirstlane((int)abortMask);

    // running best (BC67.cpp:2721-2733), the same in the four lanes of a quad
    float bestError = FLT_MAX;
    // The rest of the block's best -- seven words: the encoded end points [subset][3 dwords: (e0r|e0g<<16),(e0b|e1r<<16),(e1g|e1b<<16)]
    // and mode | partition << 4 | swap << 9 (swap bit s: the winning round of subset s exchanged its end points) -- is computed by
    // all four lanes of the quad at a commit and KEPT two words per sub-lane (words 2t, 2t+1 in sub-lane t): it is written a few
    // times per block and read once, and as nine registers per lane it was what the allocator spilled (56 B of scratch).
    u32 bestShare0 = 0, bestShare1 = 0;
    auto keepBest = [&](const int (&enc)[2][2][3], int mode, int partition, u32 swap, bool two) {
        const u32 w0 = ((u32)enc[0][0][0] & 0xffffu) | ((u32)enc[0][0][1] << 16);
        const u32 w1 = ((u32)enc[0][0][2] & 0xffffu) | ((u32)enc[0][1][0] << 16);
        const u32 w2 = ((u32)enc[0][1][1] & 0xffffu) | ((u32)enc[0][1][2] << 16);
        const u32 w3 = two ? (((u32)enc[1][0][0] & 0xffffu) | ((u32)enc[1][0][1] << 16)) : 0u;
        const u32 w4 = two ? (((u32)enc[1][0][2] & 0xffffu) | ((u32)enc[1][1][0] << 16)) : 0u;
        const u32 w5 = two ? (((u32)enc[1][1][1] & 0xffffu) | ((u32)enc[1][1][2] << 16)) : 0u;
        const u32 w6 = (u32)mode | ((u32)partition << 4) | (swap << 9);
        bestShare0 = tw == 0 ? w0 : tw == 1 ? w2 : tw == 2 ? w4 : w6;
        bestShare1 = tw == 0 ? w1 : tw == 1 ? w3 : tw == 2 ? w5 : 0u;
    };

    // A mode of the current precision as one wave-uniform word (a scalar register): mode | transformed << 4 | the bits a
    // delta loses per channel (16 - bPrec) << 8, 16, 24.
    auto modeWord = [&](int mode) -> u32 {
        const u32 w = (u32)mode | (T->bc6hModeInfo[mode][2] != 0 ? 16u : 0u) | ((16u - T->bc6hModeInfo[mode][4]) << 8) |
                      ((16u - T->bc6hModeInfo[mode][5]) << 16) | ((16u - T->bc6hModeInfo[mode][6]) << 24);
        return (u32)__builtin_amdgcn_readfirstlane((int)w);
    };
    // does the delta end point 1 - end point 0 of one subset fit the mode (BC67.cpp:2597-2663 restricted to that delta)?
    auto ownDeltaFits = [&](const int (&e)[2][3], u32 mw, int aPrec) -> bool {
        if ((mw & 16u) == 0)
            return true;
        const int mask = (1 << aPrec) - 1;
        bool ok = true;
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
        {
            const int lost = (int)((mw >> (8 + 8 * ch)) & 31u);
            const int bReduced = e[1][ch] & mask & 0xffff;
            const int d16 = (int)(short)(unsigned short)(e[1][ch] - e[0][ch]);
            const int delta = (int)(short)(unsigned short)((u32)d16 << lost) >> lost;
            ok = ok & (((delta + e[0][ch]) & mask & 0xffff) == bReduced);
        }
        return ok;
    };

    // PCA seeds of a subset and the sums of its pre-weighted pixels (EndpointSelector.h; the reference precomputes them for
    // every partition, BC67.cpp:2738-2774; per precision here).  `subsetMask` is PER LANE: the four sub-lanes of a quad compute the
    // seeds of four different subsets of their block at once (two partitions x two subsets; the power iteration with its 24
    // correctly rounded divisions is two thirds of a PCA and would otherwise be done four times over).  Every lane walks its own
    // member pixels in ascending order (the sums' order of the reference), as many steps as the wave's largest subset has pixels.
    auto pcaSeeds = [&](u32 subsetMask, Unfinished &u, float (&sums)[3]) {
        float cen[3] = {0.0f, 0.0f, 0.0f};
        float count = 0.0f;
        for (u32 rem = subsetMask; __ballot(rem != 0) != 0;)
        {
            if (rem != 0)
            {
                const int px = __builtin_ctz(rem);
                rem &= rem - 1u;
                float pw[3];
                pixPW(px, pw);
                cen[0] = cen[0] + pw[0];
                cen[1] = cen[1] + pw[1];
                cen[2] = cen[2] + pw[2];
                count = count + 1.0f;
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
            sums[ch] = cen[ch];
        const float denom = safeDenom(count);
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
            cen[ch] = cen[ch] / denom;
        float cov[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (u32 rem = subsetMask; __ballot(rem != 0) != 0;)
        {
            if (rem != 0)
            {
                const int px = __builtin_ctz(rem);
                rem &= rem - 1u;
                float pw[3];
                pixPW(px, pw);
                const float d0 = pw[0] - cen[0];
                const float d1 = pw[1] - cen[1];
                const float d2 = pw[2] - cen[2];
                cov[0] = cov[0] + d0 * d0;
                cov[1] = cov[1] + d1 * d0;
                cov[2] = cov[2] + d1 * d1;
                cov[3] = cov[3] + d2 * d0;
                cov[4] = cov[4] + d2 * d1;
                cov[5] = cov[5] + d2 * d2;
            }
        }
        float approx[3] = {1.0f, 1.0f, 1.0f};
        for (int it = 0; it < 8; it++)
        {
            float product[3];
#pragma unroll
            for (int row = 0; row < 3; row++)
            {
                float sum = 0.0f;
#pragma unroll
                for (int col = 0; col < 3; col++)
                {
                    const int hi = row > col ? row : col;
                    const int lo = row > col ? col : row;
                    sum = sum + approx[col] * cov[hi * (hi + 1) / 2 + lo];
                }
                product[row] = sum;
            }
            float largest = product[0];
            largest = sseMax(largest, product[1]);
            largest = sseMax(largest, product[2]);
            largest = safeDenom(largest);
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                approx[ch] = product[ch] / largest;
        }
        float approxLen = approx[0] * approx[0];
        approxLen = approxLen + approx[1] * approx[1];
        approxLen = approxLen + approx[2] * approx[2];
        approxLen = safeDenom(sqrtExact(approxLen));
        float direction[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
            direction[ch] = approx[ch] / approxLen;
        float minDist = FLT_MAX, maxDist = -FLT_MAX;
        for (u32 rem = subsetMask; __ballot(rem != 0) != 0;)
        {
            if (rem != 0)
            {
                const int px = __builtin_ctz(rem);
                rem &= rem - 1u;
                float pw[3];
                pixPW(px, pw);
                float dist = 0.0f;
                dist = dist + direction[0] * (pw[0] - cen[0]);
                dist = dist + direction[1] * (pw[1] - cen[1]);
                dist = dist + direction[2] * (pw[2] - cen[2]);
                minDist = sseMin(minDist, dist);
                maxDist = sseMax(maxDist, dist);
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
        {
            const float mn = cen[ch] + direction[ch] * minDist;
            const float mx = cen[ch] + direction[ch] * maxDist;
            u.base[ch] = mn / A.w[ch];
            u.offset[ch] = (mx - mn) / A.w[ch];
        }
    };
    // the value sub-lane `src` (wave-uniform) of the quad holds
    auto quadFrom = [&](float v, int src) -> float {
        const u32 x = __float_as_uint(v);
        u32 r;
        if (src == 0) r = quadBcast<0>(x);
        else if (src == 1) r = quadBcast<1>(x);
        else if (src == 2) r = quadBcast<2>(x);
        else r = quadBcast<3>(x);
        return __uint_as_float(r);
    };

    // the single-subset (4-bit indexes) and the partitioned (3-bit) search are two instantiations of the same body, so
    // that the interpolant table is 16 or 8 entries of registers
    auto searchAll = [&](auto partitionedTag) {
        constexpr bool partitioned = decltype(partitionedTag)::value;
        constexpr int numPartitions = partitioned ? 32 : 1;
        constexpr int numSubsets = partitioned ? 2 : 1;
        constexpr int indexBits = partitioned ? 3 : 4;
        constexpr int indexRange = 1 << indexBits;
        const float maxValue = (float)(indexRange - 1);
        const int weightRcp = partitioned ? 4681 : 2185; // g_weightReciprocals[8], [16]
        // (= CvttDeviceTables::rcpMaxIndex[indexBits]: the correctly rounded quotient, folded at compile time -- a literal instead of a
        // register held from here to the end of the search)
        constexpr float rcpMaxIndex = 1.0f / (float)(indexRange - 1);

        // ---- the quantised end points of a round as kept in LDS ----
        // two-subset precisions (<= 11 bits): two words per round -- r | g << 11 | b[9:0] << 22 per end point -- plus bit 10 of the
        // two blue values in s_xb; single-subset precisions (<= 16 bits): three words per round (6 x int16), the third in the
        // words of subset 1, which that search does not have
        auto packEPQ = [&](const int (&q)[2][3], u32 &wa, u32 &wb, u32 &wc) {
            if (partitioned)
            {
                wa = ((u32)q[0][0] & 0x7ffu) | (((u32)q[0][1] & 0x7ffu) << 11) | (((u32)q[0][2] & 0x3ffu) << 22);
                wb = ((u32)q[1][0] & 0x7ffu) | (((u32)q[1][1] & 0x7ffu) << 11) | (((u32)q[1][2] & 0x3ffu) << 22);
                wc = (((u32)q[0][2] >> 10) & 1u) | ((((u32)q[1][2] >> 10) & 1u) << 1);
            }
            else
            {
                wa = ((u32)q[0][0] & 0xffffu) | ((u32)q[0][1] << 16);
                wb = ((u32)q[0][2] & 0xffffu) | ((u32)q[1][0] << 16);
                wc = ((u32)q[1][1] & 0xffffu) | ((u32)q[1][2] << 16);
            }
        };
        auto unpackEPQ = [&](u32 wa, u32 wb, u32 wc, int (&e)[2][3]) {
            if (partitioned)
            {
                const u32 v[2][3] = {{wa & 0x7ffu, (wa >> 11) & 0x7ffu, (wa >> 22) | ((wc & 1u) << 10)},
                                     {wb & 0x7ffu, (wb >> 11) & 0x7ffu, (wb >> 22) | ((wc & 2u) << 9)}};
#pragma unroll
                for (int epi = 0; epi < 2; epi++)
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                        e[epi][ch] = SIGNED ? ((int)(v[epi][ch] << 21) >> 21) : (int)v[epi][ch];
            }
            else
            {
                e[0][0] = (int)(short)(wa & 0xffffu); e[0][1] = (int)(short)(wa >> 16); e[0][2] = (int)(short)(wb & 0xffffu);
                e[1][0] = (int)(short)(wb >> 16); e[1][1] = (int)(short)(wc & 0xffffu); e[1][2] = (int)(short)(wc >> 16);
            }
        };
        // the stored words of round m of subset s of the lane's block (m may differ from lane to lane)
        auto loadWords = [&](int s, int m, u32 &wa, u32 &wb, u32 &wc) {
            wa = s_epq[s][m][0][blk];
            wb = s_epq[s][m][1][blk];
            if (partitioned)
            {
                const u32 xw = *reinterpret_cast<const u32 *>(&s_xb[s][blk][0]);
                wc = (xw >> (2 * m + 2 * tweakOf(m))) & 3u; // byte t, bits 2r: 8 t + 2 r = 2 m + 2 t
            }
            else
                wc = s_epq[1][m][0][blk];
        };
        auto loadEPQ = [&](int s, int m, int (&e)[2][3]) {
            u32 wa, wb, wc;
            loadWords(s, m, wa, wb, wc);
            unpackEPQ(wa, wb, wc, e);
        };
        auto wasSwapped = [&](int s, int m) -> bool {
            const u32 iw = *reinterpret_cast<const u32 *>(&s_inv[s][blk][0]);
            return ((iw >> (m + 5 * tweakOf(m))) & 1u) != 0; // byte t, bit r: 8 t + r = m + 5 t
        };
        auto errAt = [&](int m, int s) -> float & { return s_err[s][m][blk]; };
        // Do the three words (wa, wb, wc) of a round of the lane's chain equal those of one of the meta rounds (t' < tw, r2 <= rOther)
        // or (tw, r2 < rOwn) of subset s -- as the history in LDS has them?  (The exact form of the duplicate test; the rounds'
        // words were stored by the lanes that own them earlier in this pass, and a wave's LDS operations complete in order.)
        auto earlierRoundHolds = [&](int s, u32 wa, u32 wb, u32 wc, int rOther, int rOwn) -> bool {
            bool hit = false;
            for (int t2 = 0; t2 < 4; t2++)
                for (int r2 = 0; r2 < 3; r2++)
                {
                    const bool wanted = (t2 < tw && r2 <= rOther) || (t2 == tw && r2 < rOwn);
                    u32 oa, ob, oc;
                    loadWords(s, 3 * t2 + r2, oa, ob, oc);
                    hit = hit | (wanted & (oa == wa) & (ob == wb) & (oc == wc));
                }
            return hit;
        };

        // ---- one round's index selection set up from its (un-swapped) quantised end points ----
        struct Selector
        {
            float iw[indexRange][3]; // slow: weighted linear colour of every interpolant
            float origin[3], axis[3]; // fast: projection axis
            int recBase[3], recDiff[3];
            bool interpFixup;
        };
        auto setupSelector = [&](const int (&unq)[2][3], const int (&fin)[2][3], Selector &S) {
            S.interpFixup = true;
            if (FAST)
            {
                // IndexSelector::Init on the colour-space endpoints + SelectIndexLDR
                float epDW[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    S.origin[ch] = (float)fin[0][ch];
                    epDW[ch] = ((float)fin[1][ch] - S.origin[ch]) * A.w[ch];
                }
                float lenSq = epDW[0] * epDW[0];
                lenSq = lenSq + epDW[1] * epDW[1];
                lenSq = lenSq + epDW[2] * epDW[2];
                lenSq = safeDenom(lenSq);
                const float mvdls = maxValue / lenSq;
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    S.axis[ch] = epDW[ch] * A.w[ch] * mvdls;
            }
            else
            {
                // An interpolant lies between the two finished end points (the interpolation, the unscaling and the conversion
                // are monotone), so when no lane's end point has a zero exponent field no interpolant has one and the halving of
                // TwosCLHalfToFloat is skipped (unsigned format; decided for the lanes that run this round)
                if (!SIGNED)
                {
                    int lowest = fin[0][0] < fin[1][0] ? fin[0][0] : fin[1][0];
#pragma unroll
                    for (int ch = 1; ch < 3; ch++)
                    {
                        lowest = fin[0][ch] < lowest ? fin[0][ch] : lowest;
                        lowest = fin[1][ch] < lowest ? fin[1][ch] : lowest;
                    }
                    S.interpFixup = __ballot(lowest < 0x400) != 0;
#ifdef CVTT_BC6H_DBG_FIXUP
                    S.interpFixup = true;
#endif
                }
                if (S.interpFixup)
                {
#pragma unroll
                    for (int i = 0; i < indexRange; i++)
                    {
                        const int weight = mad24(weightRcp, i, 256) >> 9;
#pragma unroll
                        for (int ch = 0; ch < 3; ch++)
                            S.iw[i][ch] = twosCLHalfToFloat<SIGNED>(reconstructChannel<SIGNED>(unq[0][ch], unq[1][ch], weight)) * A.w[ch];
                    }
                }
                else
                {
#pragma unroll
                    for (int i = 0; i < indexRange; i++)
                    {
                        const int weight = mad24(weightRcp, i, 256) >> 9;
#pragma unroll
                        for (int ch = 0; ch < 3; ch++)
                            S.iw[i][ch] = __half2float(__ushort_as_half((unsigned short)reconstructChannel<SIGNED>(unq[0][ch], unq[1][ch], weight))) * A.w[ch];
                    }
                }
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                S.recBase[ch] = unq[0][ch] * 64 + 32;
                S.recDiff[ch] = unq[1][ch] - unq[0][ch];
            }
        };
        // raw (un-inverted) index of the pixel (a, b) with linear values lf; SelectIndexHDRSlow keeps the FIRST minimum
        auto rawIndexOf = [&](const Selector &S, u32 a, u32 b, const float (&lf)[3]) -> int {
            if (FAST)
            {
                const int c0 = (int)(short)(a & 0xffffu), c1 = (int)(short)(a >> 16), c2 = (int)(short)(b & 0xffffu);
                float dist = ((float)c0 - S.origin[0]) * S.axis[0];
                dist = dist + ((float)c1 - S.origin[1]) * S.axis[1];
                dist = dist + ((float)c2 - S.origin[2]) * S.axis[2];
                return (int)clampRound(dist, maxValue);
            }
            const float l0 = lf[0] * A.w[0];
            const float l1 = lf[1] * A.w[1];
            const float l2 = lf[2] * A.w[2];
            float be = 0.0f;
            int bi = 0;
#ifdef CVTT_BC6H_PK_SCAN
            // two interpolants per packed instruction: the same IEEE operations on the same values (v_pk_add_f32 / v_pk_mul_f32
            // round each half like the scalar instruction; no contraction), so the errors are bit-identical
            typedef float pk2 __attribute__((ext_vector_type(2)));
            const pk2 L0 = {l0, l0}, L1 = {l1, l1}, L2 = {l2, l2};
#pragma unroll
            for (int i = 0; i < indexRange; i += 2)
            {
                const pk2 I0 = {S.iw[i][0], S.iw[i + 1][0]}, I1 = {S.iw[i][1], S.iw[i + 1][1]}, I2 = {S.iw[i][2], S.iw[i + 1][2]};
                pk2 d = L0 - I0;
                pk2 e = d * d;
                d = L1 - I1;
                e = e + d * d;
                d = L2 - I2;
                e = e + d * d;
                if (i == 0)
                    be = e.x;
                else
                {
                    const bool lt = e.x < be;
                    bi = lt ? i : bi;
                    be = __builtin_fminf(be, e.x);
                }
                const bool lt2 = e.y < be;
                bi = lt2 ? i + 1 : bi;
                be = __builtin_fminf(be, e.y);
            }
#else
#pragma unroll
            for (int i = 0; i < indexRange; i++)
            {
                float d = l0 - S.iw[i][0];
                float e = d * d;
                d = l1 - S.iw[i][1];
                e = e + d * d;
                d = l2 - S.iw[i][2];
                e = e + d * d;
                if (i == 0)
                    be = e;
                else
                {
                    // be = sseMin(be, e) with the index taken along on the same comparison (the errors are sums of squares: no
                    // NaN, no -0, so equal values are equal bits; the minimum as v_min_f32, the comparison only for the index)
                    const bool lt = e < be;
                    bi = lt ? i : bi;
                    be = __builtin_fminf(be, e);
                }
            }
#endif
            return bi;
        };
        // error of the pixel reconstructed with raw index `raw` (BCCommon.h:45-79)
        auto pixelError = [&](const Selector &S, int raw, u32 a, u32 b, const float (&lf)[3]) -> float {
            const int weight = (int)((mulU24((u32)raw, (u32)weightRcp) + 256u) >> 9);
            const int orig[3] = {(int)(short)(a & 0xffffu), (int)(short)(a >> 16), (int)(short)(b & 0xffffu)};
            float err = 0.0f;
            int rec[3];
            float recF[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                rec[ch] = reconstructFrom<SIGNED>(S.recBase[ch], S.recDiff[ch], weight);
                if (!FAST)
                    recF[ch] = SIGNED ? twosCLHalfToFloat<SIGNED>(rec[ch]) : __half2float(__ushort_as_half((unsigned short)rec[ch]));
            }
            if (!FAST && !SIGNED && S.interpFixup)
            {
                asm volatile("" ::: "memory"); // one wave-uniform branch around the three rare halvings
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    recF[ch] = (rec[ch] & 0x7c00) ? recF[ch] : recF[ch] * 0.5f;
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                float sq;
                if (FAST)
                {
                    // SqDiffSInt16, ParallelMath.h:996-1010
                    const int r16 = (int)(short)rec[ch];
                    const u32 du = (u32)((r16 > orig[ch] ? r16 : orig[ch]) - (r16 > orig[ch] ? orig[ch] : r16)) & 0xffffu;
                    sq = (float)(int)(du * du);
                }
                else
                {
                    const float d = recF[ch] - lf[ch];
                    sq = d * d;
                }
                err = err + sq * A.wSq[ch]; // (Flags::Uniform: the weights are 1.0 and the product is exact, no second form needed)
            }
            return err;
        };

        for (int aPrec = 16; aPrec >= 0; aPrec--)
        {
            // g_hdrModesExistForPrecision, BC67.cpp:144-149
            const u32 exists = partitioned ? 0x0fc0u : 0x11c00u;
            if (((exists >> aPrec) & 1u) == 0)
                continue;

            // the (at most three) modes of this precision, in table order (BC67.cpp:2936-2942)
            int numModesHere = 0;
            u32 modeW0 = 0, modeW1 = 0, modeW2 = 0;
            for (int mode = 0; mode < 14; mode++)
                if ((T->bc6hModeInfo[mode][1] != 0) == partitioned && T->bc6hModeInfo[mode][3] == aPrec)
                {
                    if (numModesHere == 0) modeW0 = modeWord(mode);
                    else if (numModesHere == 1) modeW1 = modeWord(mode);
                    else modeW2 = modeWord(mode);
                    numModesHere++;
                }
            numModesHere = __builtin_amdgcn_readfirstlane(numModesHere);
            auto fitsAnyMode = [&](const int (&e)[2][3]) -> bool {
                bool fits = ownDeltaFits(e, modeW0, aPrec);
                if (numModesHere > 1)
                    fits = fits | ownDeltaFits(e, modeW1, aPrec);
                if (numModesHere > 2)
                    fits = fits | ownDeltaFits(e, modeW2, aPrec);
                return fits;
            };
            // Precisions whose modes all delta-code their end points tightly are searched LAZILY (file header).  The first
            // partition of a precision whose replay is long switches the rest of the precision back to the eager search
            // (content whose deltas do fit).
            bool eagerNow = !(partitioned && aPrec >= CVTT_BC6H_LAZY_MIN);
            // A precision with ONE mode (7, 9, 10 bits) has no coupling between the blocks of a group in its commit loop: a block
            // commits a pair iff the pair beats its best and is legal, whatever its group mates do.  So each block needs the errors
            // of the rounds of ITS legal pairs only.  With three modes (8, 11 bits) a block that is better but illegal keeps the
            // mode loop of its group going, so every block of the GROUP needs the errors of every round in the group's set.
            const bool perBlockReplay = numModesHere == 1;

            // the seeds of four subsets at a time: sub-lane t holds those of subset (t & 1) of partition (p & ~1) + (t >> 1)
            Unfinished quadU;
            float quadVs[3] = {0.0f, 0.0f, 0.0f};
            for (int p = 0; p < numPartitions; p++)
            {
                const u32 partitionMask = partitioned ? (u32)__builtin_amdgcn_readfirstlane((int)T->partition2[p]) : 0u;
                // (the single-subset search computes its seeds inside the pass loop: with its sixteen-entry interpolant table it has no
                // registers to hold them across the passes)
                if (partitioned && (p & 1) == 0)
                {
                    const u32 pm = T->partition2[p + (tw >> 1)];
                    pcaSeeds((tw & 1) ? pm : (~pm & 0xffffu), quadU, quadVs);
                }
                const bool lazy = !eagerNow;
                __syncthreads(); // the previous partition's history has been read by everybody
                PROF_COUNT(0, 1)
                if (partitioned) { if (lazy) { PROF_COUNT(2, 1) } else { PROF_COUNT(5, 1) } }
                // per group (identical in its 32 lanes): bit m = meta round m of the subset takes part in the commit
                u32 roundValid0 = 0xfffu & ~abortMask, roundValid1 = 0xfffu & ~abortMask;
                u32 cand0 = 0;   // lazy: bit r = round (tw, r) of subset 0 fits its own delta in a mode (this lane's chain)
                // does a round of subset 0 fit the delta coding of a mode of this precision in some lane of the wave?  A block can
                // only be committed with such a round; a partition without one needs neither subset 1 nor the commit loop.
                bool usable0 = false;

                for (int subset = 0; subset < numSubsets; subset++)
                {
                    const u32 subsetMask = partitioned ? (subset ? partitionMask : (~partitionMask & 0xffffu)) : 0xffffu;
                    if (subset == 1 && !usable0)
                        break;
                    const int fixupIndex = __builtin_amdgcn_readfirstlane((subset == 0) ? 0 : (int)T->anchor2[p]);
                    // (the anchor pixel of the subset -- its index decides the swap of a round's end points, BC67.cpp:2525-2547 -- is
                    // loaded where a round scans it: five registers less across the passes)
                    const int count = __popc(subsetMask);
                    const float wRcp = T->rcpTable[count];
                    const float wCount = (float)count;

                    // the seeds (and the refiner's sums of the pre-weighted member pixels, EndpointRefiner.h:78-92, which do not depend on
                    // the indexes: the PCA's first pass forms the same sums in the same order) sit in sub-lane `seedSrc` of the quad
                    const int seedSrc = partitioned ? (((p & 1) << 1) | subset) : tw;
                    // meta rounds of this subset that are dropped although the rounds known at their time did not say so (see the
                    // file header): per group, found after the last pass
                    u32 forcedDrop = 0;
                    for (;;)
                    {
                        // EndpointRefiner<3> refiners[2]: fresh (zero in the canonical build) per tweak
                        float tv[3] = {0, 0, 0}, vs[3] = {0, 0, 0}, tt = 0.0f, ts = 0.0f;
                        int refCount = 0; // contributions of the previous round
                        // one word that mixes the three words of a round's end points (equal rounds have equal fingerprints); the words
                        // themselves are read back from the history in LDS in the rare cases that need them
                        u32 myFp[3] = {0, 0, 0};
                        u32 lateBits = 0; // bit r: the fingerprint of round (tw, r) equals that of a LATER pass of a lower sub-lane
                        u32 myXb = 0, myInv = 0;
                        if (partitioned)
                            s_xb[subset][blk][tw] = 0; // (rounds that have not run yet read as zeros, like the words of the rounds left out)
                        u32 rv = 0xfffu & ~abortMask;
                        bool usableNow = false;
                        u32 candNow = 0;

#pragma unroll 1
                        for (int refinePass = 0; refinePass < 3; refinePass++)
                        {
                            const int metaRound = tw * 3 + refinePass;
                            const bool act = twActive && refinePass < numRefineRounds;
                            if (refinePass >= numRefineRounds)
                                break; // (every lane's round is left out: the zero end points below are written once, before the loop)

                            // the seeds of this subset from the sub-lane that computed them (DPP: executed by all lanes, before any branch)
                            float seedBase[3] = {0.0f, 0.0f, 0.0f}, seedOffset[3] = {0.0f, 0.0f, 0.0f};
                            if (refinePass == 0)
                            {
                                if (!partitioned)
                                    pcaSeeds(0xffffu, quadU, quadVs); // (every lane its own copy: four units per block)
#pragma unroll
                                for (int ch = 0; ch < 3; ch++)
                                {
                                    seedBase[ch] = partitioned ? quadFrom(quadU.base[ch], seedSrc) : quadU.base[ch];
                                    seedOffset[ch] = partitioned ? quadFrom(quadU.offset[ch], seedSrc) : quadU.offset[ch];
                                }
                            }
                            u32 qa = 0, qb = 0, qc = 0;
                            Selector S;
                            int fixRaw = 0;
                            bool invert = false;
                            bool fits = false;
                            if (act)
                            {
                                // ---- endpoints in colour space ----
                                int epCS[2][3];
                                if (refinePass == 0)
                                {
                                    const float tf0 = T->tweakFactors[indexBits - 2][tw][0];
                                    const float tf1 = T->tweakFactors[indexBits - 2][tw][1];
#pragma unroll
                                    for (int ch = 0; ch < 3; ch++)
                                    {
                                        // FinishHDRSigned / Unsigned, UnfinishedEndpoints.h:39-75
                                        const float lo = SIGNED ? -31743.0f : 0.0f;
                                        const float f0 = sseMax(sseMin(seedBase[ch] + seedOffset[ch] * tf0, 31743.0f), lo);
                                        const float f1 = sseMax(sseMin(seedBase[ch] + seedOffset[ch] * tf1, 31743.0f), lo);
                                        epCS[0][ch] = (int)rintf(f0);
                                        epCS[1][ch] = (int)rintf(f1);
                                    }
                                }
                                else
                                {
                                    // EndpointRefiner::GetRefinedEndpointsHDR (EndpointRefiner.h:99-175) from the previous round's
                                    // sums (empty when that round was skipped as a duplicate)
                                    const float w = (refCount == 0) ? 1.0f : wCount;
                                    const float wr = (refCount == 0) ? T->rcpTable[1] : wRcp;
                                    float adenom = (tt * w - ts * ts) * wr;
                                    const bool z = (adenom == 0.0f);
                                    if (z) adenom = 1.0f;
#pragma unroll
                                    for (int ch = 0; ch < 3; ch++)
                                    {
                                        const float a = (tv[ch] - ts * vs[ch] * wr) / adenom;
                                        const float b = (vs[ch] - a * ts) * wr;
                                        float p1 = b, p2 = a + b;
                                        if (z)
                                        {
                                            p1 = vs[ch] * wr;
                                            p2 = p1;
                                        }
                                        const float lo = SIGNED ? -31743.0f : 0.0f;
                                        epCS[0][ch] = (int)rintf(sseMax(sseMin(p1 * A.rcpW[ch], 31743.0f), lo));
                                        epCS[1][ch] = (int)rintf(sseMax(sseMin(p2 * A.rcpW[ch], 31743.0f), lo));
                                    }
                                }
                                // refiners[subset].Init(...)
#pragma unroll
                                for (int ch = 0; ch < 3; ch++)
                                    tv[ch] = vs[ch] = 0.0f;
                                tt = ts = 0.0f;
                                refCount = 0;

                                // ---- QuantizeEndpoints{Signed,Unsigned}, BC67.cpp:2503-2595 ----
                                int q[2][3], unq[2][3], fin[2][3];
#pragma unroll
                                for (int epi = 0; epi < 2; epi++)
#pragma unroll
                                    for (int ch = 0; ch < 3; ch++)
                                    {
                                        if (SIGNED)
                                        {
                                            q[epi][ch] = quantizeSigned(epCS[epi][ch], aPrec);
                                            unq[epi][ch] = unquantizeSigned(q[epi][ch], aPrec, fin[epi][ch]);
                                        }
                                        else
                                        {
                                            q[epi][ch] = quantizeUnsigned(epCS[epi][ch] & 0xffff, aPrec);
                                            unq[epi][ch] = unquantizeUnsigned(q[epi][ch], aPrec, fin[epi][ch]);
                                        }
                                    }
                                setupSelector(unq, fin, S);
                                {
                                    u32 fa, fb;
                                    float fixLf[3], fixPw[3];
                                    pixLoad(fixupIndex, fa, fb, fixLf, fixPw);
                                    fixRaw = rawIndexOf(S, fa, fb, fixLf);
                                }
                                invert = (indexRange / 2 - 1) < fixRaw;
                                if (invert)
                                {
#pragma unroll
                                    for (int ch = 0; ch < 3; ch++)
                                    {
                                        const int t = q[0][ch];
                                        q[0][ch] = q[1][ch];
                                        q[1][ch] = t;
                                    }
                                }
                                packEPQ(q, qa, qb, qc);
                                if (subset == 0)
                                    fits = fitsAnyMode(q);
                                s_epq[subset][metaRound][0][blk] = qa;
                                s_epq[subset][metaRound][1][blk] = qb;
                                if (!partitioned)
                                    s_epq[1][metaRound][0][blk] = qc;
                                if (partitioned)
                                {
                                    myXb |= qc << (2 * refinePass);
                                    s_xb[subset][blk][tw] = (uint8_t)myXb;
                                }
                                myInv |= (invert ? 1u : 0u) << refinePass;
                            }
                            // (zeros -- a round the options leave out -- have the fingerprint zero)
                            const u32 fp = qa ^ ((qb << 13) | (qb >> 19)) ^ ((qc << 7) | (qc >> 25)) ^ (qc << 21);
#pragma unroll
                            for (int k = 0; k < 3; k++) // (no dynamic index: the words stay registers)
                                if (k == refinePass)
                                {
                                    myFp[k] = fp;
                                }

                            bool needError = true; // wave-uniform
                            if (subset == 0)
                            {
                                needError = __ballot(fits) != 0;
                                usableNow = usableNow || needError;
                                if (lazy && fits)
                                    candNow |= 1u << refinePass;
                            }
                            if (lazy)
                                needError = false;
#ifdef CVTT_BC6H_DBG_NOSKIP
                            needError = true;
                            usableNow = true;
#endif

                            // ---- duplicate-round test against the meta rounds that are known now: the earlier tweaks' rounds up to
                            // this pass (DPP from the quad's lower sub-lanes) and this chain's earlier rounds.  Only a group whose
                            // eight blocks ALL repeat an earlier round skips the round, so a fingerprint of the three words is
                            // compared, and the words themselves only if some group matches in it everywhere.
                            bool same = false;
#ifndef CVTT_BC6H_X_NODUP // (timing experiment: invalid output)
#pragma unroll
                            for (int r2 = 0; r2 < 3; r2++)
                            {
                                if (r2 > refinePass)
                                    continue;
                                const u32 o0 = quadBcast<0>(myFp[r2]), o1 = quadBcast<1>(myFp[r2]), o2 = quadBcast<2>(myFp[r2]);
                                same = same | ((tw > 0) & (o0 == fp)) | ((tw > 1) & (o1 == fp)) | ((tw > 2) & (o2 == fp));
                                if (r2 == refinePass)
                                {
                                    // ... and while the lower sub-lanes' fingerprints of THIS pass are here: do they repeat an EARLIER
                                    // round of this chain?  That round ran without knowing (the late duplicates, after the passes)
#pragma unroll
                                    for (int r0 = 0; r0 < 2; r0++)
                                        if (r0 < refinePass)
                                        {
                                            const bool l = ((tw > 0) & (o0 == myFp[r0])) | ((tw > 1) & (o1 == myFp[r0])) | ((tw > 2) & (o2 == myFp[r0]));
                                            lateBits |= l ? (1u << r0) : 0u;
                                        }
                                }
                                if (r2 < refinePass)
                                    same = same | (myFp[r2] == fp);
                            }
#endif
                            bool dropped = false;
                            {
                                u64 g = __ballot(same && act);
                                g &= g >> 4;
                                g &= g >> 8;
                                g &= g >> 16;
#ifdef CVTT_BC6H_DBG_NODROP
                                g = 0;
#endif
                                if ((g & 0x0000000f0000000full) != 0)
                                {
                                    __syncthreads(); // (the other sub-lanes' words of this pass are in LDS)
                                    same = earlierRoundHolds(subset, qa, qb, qc, refinePass, refinePass);
                                    same = same && act;
                                    // which rounds (tweak t2, this pass) does the lane's group drop?
                                    const u32 gb = groupBits(__ballot(same), lane);
#pragma unroll
                                    for (int t2 = 0; t2 < 4; t2++)
                                        if (((gb >> t2) & 0x11111111u) == 0x11111111u)
                                        {
                                            rv &= ~(1u << (3 * t2 + refinePass));
                                            if (t2 == tw)
                                                dropped = true;
                                        }
                                }
                            }
                            // a late duplicate found by an earlier run of these passes
                            if (forcedDrop != 0)
                            {
#pragma unroll
                                for (int t2 = 0; t2 < 4; t2++)
                                    if ((forcedDrop >> (3 * t2 + refinePass)) & 1u)
                                    {
                                        rv &= ~(1u << (3 * t2 + refinePass));
                                        if (t2 == tw)
                                            dropped = true;
                                    }
                            }

                            const bool run = act && !dropped;
                            if (!needError && refinePass == numRefineRounds - 1)
                            {
                                // nobody can use this round and no refine pass follows it
                                if (run)
                                    errAt(metaRound, subset) = FLT_MAX;
                            }
                            else if (__ballot(run) != 0)
                            {
                                // ---- error and refiner sums in pixel order (BC67.cpp:2879-2909); the indexes themselves are not kept ----
                                float subsetError = needError ? 0.0f : FLT_MAX;
                                if (run)
                                {
                                    for (u32 rem = subsetMask; rem != 0; rem &= rem - 1u)
                                    {
                                        const int px = __builtin_ctz(rem);
                                        u32 a, b;
                                        float lfp[3], pwp[3];
                                        pixLoad(px, a, b, lfp, pwp);
                                        // (the anchor's scan has been done: its index decided the inversion)
                                        const int raw = (px == fixupIndex) ? fixRaw : rawIndexOf(S, a, b, lfp);
                                        if (needError)
                                            subsetError = subsetError + pixelError(S, raw, a, b, lfp);
                                        if (refinePass != numRefineRounds - 1)
                                        {
                                            const int index = invert ? (indexRange - 1) - raw : raw;
                                            const float t = (float)index * rcpMaxIndex;
                                            tv[0] = tv[0] + t * pwp[0];
                                            tv[1] = tv[1] + t * pwp[1];
                                            tv[2] = tv[2] + t * pwp[2];
                                            tt = tt + t * t;
                                            ts = ts + t;
                                            refCount++;
                                        }
                                    }
                                    errAt(metaRound, subset) = subsetError;
                                }
                                if (refinePass != numRefineRounds - 1)
                                {
                                    // (DPP outside the `if (run)`: every lane executes it)
#pragma unroll
                                    for (int ch = 0; ch < 3; ch++)
                                    {
                                        const float v = partitioned ? quadFrom(quadVs[ch], seedSrc) : quadVs[ch];
                                        if (run)
                                            vs[ch] = v;
                                    }
                                }
                            }
                        }
                        // the spare bits and the swap flags of this chain's rounds, and -- once per run -- zeros for the rounds the
                        // options leave out
                        if (partitioned && !twActive)
                            s_xb[subset][blk][tw] = 0;
                        s_inv[subset][blk][tw] = (uint8_t)myInv;
                        if (abortMask != 0)
                        {
#pragma unroll
                            for (int r = 0; r < 3; r++)
                                if (!twActive || r >= numRefineRounds)
                                {
                                    s_epq[subset][tw * 3 + r][0][blk] = 0;
                                    s_epq[subset][tw * 3 + r][1][blk] = 0;
                                    if (!partitioned)
                                        s_epq[1][tw * 3 + r][0][blk] = 0;
                                }
                        }

                        // ---- the rounds that were not known in time: (t' < tw, r' > r) ----
                        {
                            // (fingerprints: gathered during the passes; the passes the options leave out have zero words, fingerprint 0)
                            if (numRefineRounds < 3 && tw > 0)
                            {
#pragma unroll
                                for (int r = 0; r < 2; r++)
                                    lateBits |= (myFp[r] == 0u) ? (1u << r) : 0u;
                            }
                            bool anyLate = twActive && (lateBits & ((1u << numRefineRounds) - 1u) & 3u) != 0;
                            u32 newDrop = 0; // per group: meta rounds that ran although all eight blocks repeat an earlier round
#ifdef CVTT_BC6H_DBG_NOLATE
                            anyLate = false;
#endif
#ifdef CVTT_BC6H_TRACE
                            if (partitioned && aPrec == 6 && p == 11 && subset == 1 && blockIdx.x == (u32)(CVTT_BC6H_TRACE) / 16u)
                            {
                                const u64 b0 = __ballot(anyLate);
                                if (lane == 0) { g_bc6hTrace2[0] += 1; g_bc6hTrace2[1] = b0; g_bc6hTrace2[2] = forcedDrop; }
                            }
#endif
                            if (__ballot(anyLate) != 0)
                            {
                                // the complete test of rounds 0 and 1 of every chain: all three words, every earlier meta round
                                __syncthreads();
#pragma unroll 1
                                for (int r = 0; r < 2; r++)
                                {
                                    u32 wa, wb, wc;
                                    loadWords(subset, 3 * tw + r, wa, wb, wc);
                                    bool full = earlierRoundHolds(subset, wa, wb, wc, 2, r);
                                    full = full && twActive && r < numRefineRounds;
                                    const u32 gb = groupBits(__ballot(full), lane);
#ifdef CVTT_BC6H_TRACE
                                    if (partitioned && aPrec == 6 && p == 11 && subset == 1 && blockIdx.x == (u32)(CVTT_BC6H_TRACE) / 16u)
                                    {
                                        const u64 b1 = __ballot(full);
                                        if (lane == 0) { g_bc6hTrace2[3 + r] = b1; g_bc6hTrace2[5 + r] = rv; }
                                    }
#endif
#pragma unroll
                                    for (int t2 = 1; t2 < 4; t2++)
                                        if (((gb >> t2) & 0x11111111u) == 0x11111111u && ((rv >> (3 * t2 + r)) & 1u))
                                            newDrop |= 1u << (3 * t2 + r);
                                }
                            }
                            if (__ballot(newDrop != 0) == 0)
                            {
                                if (subset == 0) { roundValid0 = rv; usable0 = usableNow; cand0 = candNow; } else roundValid1 = rv;
                                break;
                            }
                            // the first of them in the reference's order is certain (everything before it is final): the passes run again
                            // with it dropped.  (A group without a new drop repeats its passes with the same result.)
                            if (newDrop != 0)
                                forcedDrop |= 1u << __builtin_ctz(newDrop);
                            PROF_COUNT(1, 1)
#ifdef CVTT_BC6H_PROFILE
                            if ((lane & 31) == 0 && newDrop != 0)
                            {
                                atomicAdd(&g_bc6hProf[8 + __builtin_ctz(newDrop)], 1ull); // which round is forced
                                atomicAdd(&g_bc6hProf[20 + (__popc(newDrop) > 7 ? 7 : __popc(newDrop))], 1ull); // how many new drops the run found
                            }
#endif
                            __syncthreads();
                        }
                    }
                }
                __syncthreads(); // the history of this partition is complete: every lane may read every round of its block
#ifdef CVTT_BC6H_TRACE
                if (partitioned && aPrec == 6 && blockIndex == (u32)(CVTT_BC6H_TRACE) && tw == 0)
                {
                    unsigned *d = &g_bc6hTrace[p * 74];
                    d[0] = roundValid0;
                    d[1] = usable0 ? roundValid1 : 0xdeadu;
                    for (int s2 = 0; s2 < 2; s2++)
                        for (int m = 0; m < 12; m++)
                        {
                            d[2 + (s2 * 12 + m) * 3 + 0] = s_epq[s2][m][0][blk];
                            d[2 + (s2 * 12 + m) * 3 + 1] = s_epq[s2][m][1][blk];
                            d[2 + (s2 * 12 + m) * 3 + 2] = __float_as_uint(s_err[s2][m][blk]);
                        }
                }
#endif

                if (lazy && usable0)
                {
                    // ---- which pairs of rounds could the block commit? (Evaluate*Legality, BC67.cpp:2597-2663, on end points alone) ----
                    // lane (block, t) takes the subset-0 rounds of its own chain that fit their own delta and pairs them with all
                    // twelve rounds of subset 1
                    u32 need0 = 0, need1 = 0; // bit m: the error of meta round m of subset 0 / 1 is needed (by this block)
                    u32 c0 = cand0 & (roundValid0 >> (3 * tw)) & 7u;
                    while (__ballot(c0 != 0) != 0)
                    {
                        const bool actL = c0 != 0;
                        const int r0 = actL ? __builtin_ctz(c0) : 0;
                        c0 &= c0 - 1u;
                        const int m0 = 3 * tw + r0;
                        int e0[2][3];
                        loadEPQ(0, m0, e0);
                        const bool own0 = ownDeltaFits(e0, modeW0, aPrec);
                        const bool own1 = numModesHere > 1 && ownDeltaFits(e0, modeW1, aPrec);
                        const bool own2 = numModesHere > 2 && ownDeltaFits(e0, modeW2, aPrec);
                        if (__ballot(actL && (own0 || own1 || own2)) == 0)
                            continue;
                        for (int m1 = 0; m1 < 12; m1++)
                        {
                            int x[2][3];
                            loadEPQ(1, m1, x);
                            const bool in = actL && ((roundValid1 >> m1) & 1u);
                            bool ok0 = in && own0, ok1 = in && own1, ok2 = in && own2;
                            const int mask = (1 << aPrec) - 1;
                            auto fitsOne = [&](int v, int base, u32 mw, int ch) -> bool {
                                const int lost = (int)((mw >> (8 + 8 * ch)) & 31u);
                                const int d16 = (int)(short)(unsigned short)(v - base);
                                const int delta = (int)(short)(unsigned short)((u32)d16 << lost) >> lost;
                                return ((delta + base) & mask & 0xffff) == (v & mask & 0xffff);
                            };
                            bool alive = true;
#pragma unroll
                            for (int epi = 0; epi < 2 && alive; epi++)
#pragma unroll
                                for (int ch = 0; ch < 3 && alive; ch++)
                                {
                                    ok0 = ok0 && fitsOne(x[epi][ch], e0[0][ch], modeW0, ch);
                                    if (numModesHere > 1)
                                        ok1 = ok1 && fitsOne(x[epi][ch], e0[0][ch], modeW1, ch);
                                    if (numModesHere > 2)
                                        ok2 = ok2 && fitsOne(x[epi][ch], e0[0][ch], modeW2, ch);
                                    alive = __ballot(ok0 || ok1 || ok2) != 0;
                                }
                            if (ok0 || ok1 || ok2)
                            {
                                need0 |= 1u << m0;
                                need1 |= 1u << m1;
                            }
                        }
                    }
                    if (__ballot(need0 != 0) == 0)
                        continue; // nobody can commit anything with this partition at this precision
                    PROF_COUNT(3, 1)
                    // the block's sets (the quad's lanes looked at different rounds of subset 0) ...
                    need0 |= (u32)__builtin_amdgcn_mov_dpp((int)need0, 0xb1, 0xf, 0xf, true); // quad_perm [1,0,3,2]
                    need0 |= (u32)__builtin_amdgcn_mov_dpp((int)need0, 0x4e, 0xf, 0xf, true); // quad_perm [2,3,0,1]
                    need1 |= (u32)__builtin_amdgcn_mov_dpp((int)need1, 0xb1, 0xf, 0xf, true);
                    need1 |= (u32)__builtin_amdgcn_mov_dpp((int)need1, 0x4e, 0xf, 0xf, true);
                    if (!perBlockReplay)
                    {
                        // ... and with three modes the group's: a block that is better but illegal keeps its group's mode loop going
#pragma unroll
                        for (int step = 4; step < 32; step <<= 1)
                        {
                            need0 |= xorLane(need0, step);
                            need1 |= xorLane(need1, step);
                        }
                    }
                    // ---- replay: the rounds of the set again, from their quantised end points, this time with errors; each by the
                    // lane whose chain the round belongs to ----
                    int slots = 0;
#pragma unroll 1
                    for (int rs = 0; rs < 2; rs++)
                    {
#pragma unroll 1
                        for (int r = 0; r < 3; r++)
                        {
                            const int rm = 3 * tw + r;
                            const bool actL = (((rs ? need1 : need0) >> rm) & 1u) != 0;
                            if (__ballot(actL) == 0)
                                continue;
                            slots++;
                            if (actL)
                            {
                                const u32 rmask = rs ? partitionMask : (~partitionMask & 0xffffu);
                                // the end points as the round had them before it swapped them (the scan's first-minimum rule sees the order)
                                const bool was = wasSwapped(rs, rm);
                                int rq[2][3];
                                loadEPQ(rs, rm, rq);
                                int unq[2][3], fin[2][3];
#pragma unroll
                                for (int ch = 0; ch < 3; ch++)
                                {
                                    const int q0 = was ? rq[1][ch] : rq[0][ch], q1 = was ? rq[0][ch] : rq[1][ch];
                                    unq[0][ch] = SIGNED ? unquantizeSigned(q0, aPrec, fin[0][ch]) : unquantizeUnsigned(q0, aPrec, fin[0][ch]);
                                    unq[1][ch] = SIGNED ? unquantizeSigned(q1, aPrec, fin[1][ch]) : unquantizeUnsigned(q1, aPrec, fin[1][ch]);
                                }
                                Selector S;
                                setupSelector(unq, fin, S);
                                float subsetError = 0.0f;
                                for (u32 rem = rmask; rem != 0; rem &= rem - 1u)
                                {
                                    const int px = __builtin_ctz(rem);
                                    u32 a, b;
                                    float lfp[3], pwp[3];
                                    pixLoad(px, a, b, lfp, pwp);
                                    const int raw = rawIndexOf(S, a, b, lfp);
                                    subsetError = subsetError + pixelError(S, raw, a, b, lfp);
                                }
                                errAt(rm, rs) = subsetError;
                            }
                        }
                    }
                    PROF_COUNT(4, slots)
                    if (slots > CVTT_BC6H_LAZY_SWITCH)
                        eagerNow = true;
                    __syncthreads();
                }

                // ---- delta-coding legality + commit, BC67.cpp:2914-2986 (the four lanes of a quad do the same) ----
                if (!usable0)
                    continue;
#ifdef CVTT_BC6H_X_NOCOMMIT // (timing experiment: invalid output)
                if (p != 31)
                    continue;
#endif
                const int numMeta1 = partitioned ? 12 : 1;
                // cheapest valid subset-1 round: no combination with meta0 can beat the best unless this one does.  Every sub-lane
                // looks at the three rounds of its own chain, the quad takes the minimum (sseMin-free: no NaN among errors that count,
                // and a NaN never passes the `<` of the commit either way).
                float minErr1 = 0.0f;
                if (partitioned)
                {
                    minErr1 = FLT_MAX;
#pragma unroll
                    for (int r = 0; r < 3; r++)
                    {
                        const float e = errAt(3 * tw + r, 1);
                        if (((roundValid1 >> (3 * tw + r)) & 1u) && e < minErr1)
                            minErr1 = e;
                    }
                    float o = __uint_as_float((u32)__builtin_amdgcn_mov_dpp((int)__float_as_uint(minErr1), 0xb1, 0xf, 0xf, true));
                    minErr1 = o < minErr1 ? o : minErr1;
                    o = __uint_as_float((u32)__builtin_amdgcn_mov_dpp((int)__float_as_uint(minErr1), 0x4e, 0xf, 0xf, true));
                    minErr1 = o < minErr1 ? o : minErr1;
                }
                // Can ANY round of subset 0 of any block of the wave beat its block's best with that?  Each sub-lane asks for its own
                // three rounds; when nobody can, the loop below would pass every meta0 without touching anything (its first test).
                {
                    bool mine = false;
#pragma unroll
                    for (int r = 0; r < 3; r++)
                    {
                        const float e0 = errAt(3 * tw + r, 0);
                        mine = mine | ((((roundValid0 >> (3 * tw + r)) & 1u) != 0) & ((partitioned ? e0 + minErr1 : e0) < bestError));
                    }
                    if (__ballot(mine) == 0)
                        continue;
                }
                if (numModesHere == 1)
                {
                    // ---- ONE mode at this precision (6, 7, 9, 10 bits; every single-subset precision): no block looks at its group
                    // mates in the commit loop -- a pair is committed iff it beats the block's best and is legal -- so what the loop
                    // leaves behind is the legal pair with the smallest combined error below the best, the first such pair in
                    // (meta0, meta1) order (every commit needs a strictly smaller error).  Sub-lane t looks for it among the pairs
                    // of ITS chain's subset-0 rounds (meta0 = 3t ... 3t+2: consecutive in that order), the quad takes the smallest,
                    // the lower sub-lane on equal errors. ----
                    const u32 mw = modeW0;
                    const bool transformed = (mw & 16u) != 0;
                    const int mask = (1 << aPrec) - 1;
                    auto fitsOne = [&](int v, int base, int ch) -> bool {
                        const int lost = (int)((mw >> (8 + 8 * ch)) & 31u);
                        const int d16 = (int)(short)(unsigned short)(v - base);
                        const int delta = (int)(short)(unsigned short)((u32)d16 << lost) >> lost;
                        return ((delta + base) & mask & 0xffff) == (v & mask & 0xffff);
                    };
                    float localBest = bestError;
                    int localPair = -1; // meta0 * 16 + meta1
#pragma unroll 1
                    for (int r = 0; r < 3; r++)
                    {
                        const int m0 = 3 * tw + r;
                        const float err0 = errAt(m0, 0);
                        bool cand = (((roundValid0 >> m0) & 1u) != 0) & ((partitioned ? err0 + minErr1 : err0) < localBest);
                        if (__ballot(cand) == 0)
                            continue;
                        int e0[2][3];
                        loadEPQ(0, m0, e0);
                        cand = cand & ownDeltaFits(e0, mw, aPrec);
                        if (!partitioned)
                        {
                            if (cand)
                            {
                                localBest = err0;
                                localPair = m0 * 16;
                            }
                            continue;
                        }
                        if (__ballot(cand) == 0)
                            continue;
#pragma unroll 1
                        for (int m1 = 0; m1 < 12; m1++)
                        {
                            const float c = err0 + errAt(m1, 1);
                            bool ok = cand & (((roundValid1 >> m1) & 1u) != 0) & (c < localBest);
                            if (__ballot(ok) == 0)
                                continue;
                            if (transformed)
                            {
                                int x[2][3];
                                loadEPQ(1, m1, x);
#pragma unroll
                                for (int epi = 0; epi < 2; epi++)
#pragma unroll
                                    for (int ch = 0; ch < 3; ch++)
                                        ok = ok & fitsOne(x[epi][ch], e0[0][ch], ch);
                            }
                            if (ok)
                            {
                                localBest = c;
                                localPair = m0 * 16 + m1;
                            }
                        }
                    }
                    // the quad's winner (all four lanes end up with the same)
                    float wb = __uint_as_float(quadBcast<0>(__float_as_uint(localBest)));
                    int wp = (int)quadBcast<0>((u32)localPair);
#define CVTT_QUAD_STEP(Q) { const float ob = __uint_as_float(quadBcast<Q>(__float_as_uint(localBest))); const int op = (int)quadBcast<Q>((u32)localPair); \
                            const bool lt = ob < wb; wb = lt ? ob : wb; wp = lt ? op : wp; }
                    CVTT_QUAD_STEP(1) CVTT_QUAD_STEP(2) CVTT_QUAD_STEP(3)
#undef CVTT_QUAD_STEP
                    const bool improved = (wp >= 0) & (wb < bestError);
                    if (__ballot(improved) != 0)
                    {
                        if (improved)
                        {
                            const int m0 = wp >> 4, m1 = wp & 15;
                            int e0[2][3], e1[2][3] = {{0, 0, 0}, {0, 0, 0}};
                            loadEPQ(0, m0, e0);
                            if (partitioned)
                                loadEPQ(1, m1, e1);
                            int enc[2][2][3];
#pragma unroll
                            for (int ch = 0; ch < 3; ch++)
                            {
                                enc[0][0][ch] = e0[0][ch];
                                enc[0][1][ch] = e0[1][ch];
                                enc[1][0][ch] = partitioned ? e1[0][ch] : 0;
                                enc[1][1][ch] = partitioned ? e1[1][ch] : 0;
                                if (transformed)
                                {
                                    // the deltas the mode stores (they fit: the pair is legal)
                                    const int lost = (int)((mw >> (8 + 8 * ch)) & 31u);
#pragma unroll
                                    for (int sb = 0; sb < 2; sb++)
#pragma unroll
                                        for (int epi = 0; epi < 2; epi++)
                                        {
                                            if ((sb == 0 && epi == 0) || (sb == 1 && !partitioned))
                                                continue;
                                            const int d16 = (int)(short)(unsigned short)(enc[sb][epi][ch] - enc[0][0][ch]);
                                            enc[sb][epi][ch] = (int)(short)(unsigned short)((u32)d16 << lost) >> lost;
                                        }
                                }
                            }
                            bestError = wb;
                            keepBest(enc, (int)(mw & 15u), p, (wasSwapped(0, m0) ? 1u : 0u) | ((partitioned && wasSwapped(1, m1)) ? 2u : 0u), partitioned);
                        }
                    }
                    continue;
                }
#ifdef CVTT_BC6H_X_NOCOMMIT3 // (timing experiment: invalid output)
                if (p != 31)
                    continue;
#endif
#ifndef CVTT_BC6H_X_SEQ3
                if (partitioned)
                {
                    // ---- SEVERAL modes at this precision (8 and 11 bits).  What the mode loop below leaves behind, worked out
                    // without walking the pairs one at a time.  For a pair Y of rounds let S(Y) = the blocks of the group whose
                    // combined error beats their best just before Y, L_i(Y) = the modes block i's deltas fit at Y, f_i = min L_i.
                    // The loop commits i in S at EVERY legal mode up to the mode E(Y) where it stops, and it stops at the first mode at
                    // which somebody commits and nobody of S is left without a commit: E(Y) = max over S of f_i, the last mode when
                    // somebody of S fits none.  So (1) a block's best error moves exactly as with one mode -- down to the combined
                    // error of every pair that beats it and is legal in SOME mode -- and its final pair X is the first pair with the
                    // smallest such error; (2) its mode is the highest one of L(X) that is <= E(X).  (1) is the quad search of the
                    // one-mode path with a mode mask.  (2) needs the group mates at X only when L(X) holds more than one mode: a mate
                    // with error(X) >= its best before the search is not in S, one with error(X) < its best AFTER the search is, and
                    // for one in between S depends on the pairs before X -- when such a mate could raise the mode, the wave takes the
                    // pair-by-pair loop below instead (nothing has been written by then). ----
                    const int mask = (1 << aPrec) - 1;
                    auto fitsOne = [&](u32 mw, int v, int base, int ch) -> bool {
                        const int lost = (int)((mw >> (8 + 8 * ch)) & 31u);
                        const int d16 = (int)(short)(unsigned short)(v - base);
                        const int delta = (int)(short)(unsigned short)((u32)d16 << lost) >> lost;
                        return ((delta + base) & mask & 0xffff) == (v & mask & 0xffff);
                    };
                    auto ownMask = [&](const int (&e)[2][3]) -> u32 {
                        u32 m = ownDeltaFits(e, modeW0, aPrec) ? 1u : 0u;
                        if (numModesHere > 1)
                            m |= ownDeltaFits(e, modeW1, aPrec) ? 2u : 0u;
                        if (numModesHere > 2)
                            m |= ownDeltaFits(e, modeW2, aPrec) ? 4u : 0u;
                        return m;
                    };
                    // the modes of `own` that subset 1's end points fit too (deltas against subset 0's first end point)
                    auto pairMask = [&](u32 own, const int (&e0)[2][3], const int (&x)[2][3]) -> u32 {
                        u32 m = own;
#pragma unroll
                        for (int mi = 0; mi < 3; mi++)
                        {
                            if (mi >= numModesHere)
                                break;
                            const u32 mw = (mi == 0) ? modeW0 : (mi == 1) ? modeW1 : modeW2;
                            if ((mw & 16u) == 0)
                                continue;
                            bool ok = true;
#pragma unroll
                            for (int epi = 0; epi < 2; epi++)
#pragma unroll
                                for (int ch = 0; ch < 3; ch++)
                                    ok = ok & fitsOne(mw, x[epi][ch], e0[0][ch], ch);
                            m &= ok ? ~0u : ~(1u << mi);
                        }
                        return m;
                    };
                    float localBest = bestError;
                    int localPair = -1; // meta0 * 16 + meta1
                    u32 localModes = 0;
#pragma unroll 1
                    for (int r = 0; r < 3; r++)
                    {
                        const int m0 = 3 * tw + r;
                        const float err0 = errAt(m0, 0);
                        bool cand = (((roundValid0 >> m0) & 1u) != 0) & (err0 + minErr1 < localBest);
                        if (__ballot(cand) == 0)
                            continue;
                        int e0[2][3];
                        loadEPQ(0, m0, e0);
                        const u32 own = ownMask(e0);
                        cand = cand & (own != 0);
                        if (__ballot(cand) == 0)
                            continue;
#pragma unroll 1
                        for (int m1 = 0; m1 < 12; m1++)
                        {
                            const float c = err0 + errAt(m1, 1);
                            bool ok = cand & (((roundValid1 >> m1) & 1u) != 0) & (c < localBest);
                            if (__ballot(ok) == 0)
                                continue;
                            int x[2][3];
                            loadEPQ(1, m1, x);
                            const u32 L = pairMask(own, e0, x);
                            if (ok & (L != 0))
                            {
                                localBest = c;
                                localPair = m0 * 16 + m1;
                                localModes = L;
                            }
                        }
                    }
                    float wb = __uint_as_float(quadBcast<0>(__float_as_uint(localBest)));
                    int wp = (int)quadBcast<0>((u32)localPair);
                    u32 wL = quadBcast<0>(localModes);
#define CVTT_QUAD_STEP(Q) { const float ob = __uint_as_float(quadBcast<Q>(__float_as_uint(localBest))); const int op = (int)quadBcast<Q>((u32)localPair); \
                            const u32 ol = quadBcast<Q>(localModes); const bool lt = ob < wb; wb = lt ? ob : wb; wp = lt ? op : wp; wL = lt ? ol : wL; }
                    CVTT_QUAD_STEP(1) CVTT_QUAD_STEP(2) CVTT_QUAD_STEP(3)
#undef CVTT_QUAD_STEP
                    const bool improved = (wp >= 0) & (wb < bestError);
                    if (__ballot(improved) == 0)
                        continue;
                    const float bestAfter = improved ? wb : bestError;
                    int modeIdx = improved ? (int)__builtin_ctz(wL | 8u) : 0;
                    bool walkPairs = false;
                    u64 toResolve = __ballot(improved & ((wL & (wL - 1u)) != 0) & (tw == 0));
                    while (toResolve != 0)
                    {
                        const int src = (int)__builtin_ctzll(toResolve);
                        toResolve &= toResolve - 1;
                        const int X = __builtin_amdgcn_readlane(wp, src);
                        const u32 Lq = (u32)__builtin_amdgcn_readlane((int)wL, src);
                        const int k1 = (int)__builtin_ctz(Lq);
                        const int m0 = X >> 4, m1 = X & 15;
                        const bool mate = (((lane ^ src) & 32) == 0) & ((lane >> 2) != (src >> 2));
                        const float ci = errAt(m0, 0) + errAt(m1, 1);
                        const bool maybeIn = mate & (((roundValid0 >> m0) & 1u) != 0) & (((roundValid1 >> m1) & 1u) != 0) & (ci < bestError);
                        if (__ballot(maybeIn) == 0)
                            continue;
                        int y0[2][3], y1[2][3];
                        loadEPQ(0, m0, y0);
                        loadEPQ(1, m1, y1);
                        const u32 Li = pairMask(ownMask(y0), y0, y1);
                        const int fi = Li != 0 ? (int)__builtin_ctz(Li) : numModesHere - 1;
                        // would this mate, if it is in S, carry the loop to a legal mode of the block above k1?
                        const bool raises = maybeIn & ((Lq & ((2u << fi) - 1u) & ~((2u << k1) - 1u)) != 0);
                        // in S for certain: the error beats even the best AFTER the search, or this is the pair the mate commits
                        // itself; a mate whose own pair comes before X has its final best at X (decided by the first test), and
                        // so has one that does not improve.  Open: a mate whose own pair comes after X, error in between.
                        const bool inS = (ci < bestAfter) | (improved & (wp == X));
                        const bool sure = raises & inS;
                        if (__ballot(raises & !inS & improved & (wp > X)) != 0)
                        {
                            walkPairs = true;
                            break;
                        }
                        const u64 sureBits = __ballot(sure);
                        if (sureBits == 0)
                            continue;
                        int E = k1;
                        if (__ballot(sure & (fi >= 1)) != 0)
                            E = E > 1 ? E : 1;
                        if (__ballot(sure & (fi >= 2)) != 0)
                            E = 2;
                        const int top = 31 - (int)__builtin_clz(Lq & ((2u << E) - 1u));
                        if ((lane >> 2) == (src >> 2))
                            modeIdx = top;
                    }
                    if (!walkPairs)
                    {
                        if (improved)
                        {
                            const u32 mw = (modeIdx == 0) ? modeW0 : (modeIdx == 1) ? modeW1 : modeW2;
                            const bool transformed = (mw & 16u) != 0;
                            const int m0 = wp >> 4, m1 = wp & 15;
                            int e0[2][3], e1[2][3];
                            loadEPQ(0, m0, e0);
                            loadEPQ(1, m1, e1);
                            int enc[2][2][3];
#pragma unroll
                            for (int ch = 0; ch < 3; ch++)
                            {
                                enc[0][0][ch] = e0[0][ch];
                                enc[0][1][ch] = e0[1][ch];
                                enc[1][0][ch] = e1[0][ch];
                                enc[1][1][ch] = e1[1][ch];
                                if (transformed)
                                {
                                    const int lost = (int)((mw >> (8 + 8 * ch)) & 31u);
#pragma unroll
                                    for (int sb = 0; sb < 2; sb++)
#pragma unroll
                                        for (int epi = 0; epi < 2; epi++)
                                        {
                                            if (sb == 0 && epi == 0)
                                                continue;
                                            const int d16 = (int)(short)(unsigned short)(enc[sb][epi][ch] - enc[0][0][ch]);
                                            enc[sb][epi][ch] = (int)(short)(unsigned short)((u32)d16 << lost) >> lost;
                                        }
                                }
                            }
                            bestError = wb;
                            keepBest(enc, (int)(mw & 15u), p, (wasSwapped(0, m0) ? 1u : 0u) | (wasSwapped(1, m1) ? 2u : 0u), true);
                        }
                        PROF_COUNT(6, 1)
                        continue;
                    }
                    PROF_COUNT(7, 1)
                }
#endif
                for (int meta0 = 0; meta0 < 12; meta0++)
                {
                    const bool valid0 = ((roundValid0 >> meta0) & 1u) != 0;
                    const float err0 = errAt(meta0, 0);
                    const bool canBeat = valid0 && ((partitioned ? err0 + minErr1 : err0) < bestError);
                    if (__ballot(canBeat) == 0)
                        continue;
                    // quantised endpoints of subset 0's round, and whether its own delta fits each mode of this precision
                    int e0[2][3];
                    bool legal0[3] = {true, true, true};
                    {
                        loadEPQ(0, meta0, e0);
                        legal0[0] = ownDeltaFits(e0, modeW0, aPrec);
                        if (numModesHere > 1)
                            legal0[1] = ownDeltaFits(e0, modeW1, aPrec);
                        if (numModesHere > 2)
                            legal0[2] = ownDeltaFits(e0, modeW2, aPrec);
                    }
                    // a block whose subset-0 delta fits no mode cannot commit with this meta0 whatever meta1 is
                    if (__ballot(canBeat && (legal0[0] || (numModesHere > 1 && legal0[1]) || (numModesHere > 2 && legal0[2]))) == 0)
                        continue;
                    for (int meta1 = 0; meta1 < numMeta1; meta1++)
                    {
                        const bool roundsOk = valid0 && (!partitioned || ((roundValid1 >> meta1) & 1u));
                        float combined = err0;
                        if (partitioned)
                            combined = combined + errAt(meta1, 1);
                        const bool errorBetter = roundsOk && (combined < bestError);
                        if (__ballot(errorBetter) == 0)
                            continue;
                        const bool groupAny = groupBits(__ballot(errorBetter), lane) != 0;
                        bool needsCommit = errorBetter;
                        bool groupDone = !groupAny; // this group's mode loop has ended (or never started)

                        int e1[2][3] = {{0, 0, 0}, {0, 0, 0}};
                        if (partitioned)
                            loadEPQ(1, meta1, e1);

                        for (int mi = 0; mi < numModesHere; mi++)
                        {
                            const u32 mw = (mi == 0) ? modeW0 : (mi == 1) ? modeW1 : modeW2;
                            const int mode = (int)(mw & 15u);
                            // nobody can commit in this mode: the mode changes no state (BC67.cpp:2954-2955 `continue`)
                            const bool l0 = (mi == 0) ? legal0[0] : (mi == 1) ? legal0[1] : legal0[2];
                            if (__ballot(errorBetter && l0 && !groupDone) == 0)
                                continue;
                            const bool transformed = (mw & 16u) != 0;
                            const int lostBits[3] = {(int)((mw >> 8) & 31u), (int)((mw >> 16) & 31u), (int)((mw >> 24) & 31u)};

                            // Evaluate{Partitioned,Single}Legality, BC67.cpp:2597-2663
                            int enc[2][2][3];
                            bool legal = true;
                            const int mask = (1 << aPrec) - 1;
#pragma unroll
                            for (int ch = 0; ch < 3; ch++)
                            {
                                enc[0][0][ch] = e0[0][ch];
                                enc[0][1][ch] = e0[1][ch];
                                enc[1][0][ch] = partitioned ? e1[0][ch] : 0;
                                enc[1][1][ch] = partitioned ? e1[1][ch] : 0;
                                if (transformed)
                                {
                                    const int lost = lostBits[ch];
#pragma unroll
                                    for (int s = 0; s < 2; s++)
#pragma unroll
                                        for (int epi = 0; epi < 2; epi++)
                                        {
                                            if ((s == 0 && epi == 0) || (s == 1 && !partitioned))
                                                continue;
                                            const int bReduced = enc[s][epi][ch] & mask & 0xffff;
                                            const int d16 = (int)(short)(unsigned short)(enc[s][epi][ch] - enc[0][0][ch]);
                                            const int delta = (int)(short)(unsigned short)((u32)d16 << lost) >> lost;
                                            enc[s][epi][ch] = delta;
                                            const int reconstructed = (delta + enc[0][0][ch]) & mask & 0xffff;
                                            legal = legal && (reconstructed == bReduced);
                                        }
                                }
                            }

                            const bool commit = errorBetter && legal && !groupDone;
                            const u32 gCommit = groupBits(__ballot(commit), lane);
                            if (__ballot(commit) != 0)
                            {
                                if (commit)
                                {
                                    bestError = combined;
                                    // (the indexes of the two rounds are selected again after the search, from these end
                                    // points in the order the rounds had them)
                                    keepBest(enc, mode, p, (wasSwapped(0, meta0) ? 1u : 0u) | ((partitioned && wasSwapped(1, meta1)) ? 2u : 0u), partitioned);
                                    needsCommit = false;
                                }
                            }
                            // `continue` when nobody of the group commits skips the needsCommit test (BC67.cpp:2954-2955)
                            if (gCommit != 0 && groupBits(__ballot(needsCommit && !groupDone), lane) == 0)
                                groupDone = true;
                            if (__ballot(!groupDone) == 0)
                                break;
                        }
                    }
                }
            }
        }
    };
#ifndef CVTT_BC6H_X_NOSINGLE // (timing experiment: invalid output)
    searchAll(std::false_type{});
#endif
    searchAll(std::true_type{});

    // ---- the winner's indexes, selected again from its end points.  A round's indexes are a function of its quantised end
    // points in the order it had them before the anchor swap, of the precision and of the pixels (QuantizeEndpoints* +
    // SelectIndexHDR*, BC67.cpp:2503-2595, 2879-2893): the same operations on the same values here, once per block.  Sub-lane t
    // selects pixels 4t ... 4t+3; the quad then puts the words together. ----
    u32 bestIdxLo = 0, bestIdxHi = 0;
    REFRESH_LANE();
    // the block's best back from the four sub-lanes
    const u32 bestEP[6] = {quadBcast<0>(bestShare0), quadBcast<0>(bestShare1), quadBcast<1>(bestShare0),
                           quadBcast<1>(bestShare1), quadBcast<2>(bestShare0), quadBcast<2>(bestShare1)};
    const u32 bestW6 = quadBcast<3>(bestShare0);
    const int bestMode = (int)(bestW6 & 15u), bestPartition = (int)((bestW6 >> 4) & 31u);
    const u32 bestSwap = bestW6 >> 9;
    {
        const bool partitionedB = T->bc6hModeInfo[bestMode][1] != 0;
        const int aPrecB = (int)T->bc6hModeInfo[bestMode][3];
        const bool transformedB = T->bc6hModeInfo[bestMode][2] != 0;
        const u32 pmask = partitionedB ? T->partition2[bestPartition & 31] : 0u;
        const int rangeB = partitionedB ? 8 : 16;
        const int weightRcpB = partitionedB ? 4681 : 2185;
        const float maxValueB = (float)(rangeB - 1);
        const int enc[2][2][3] = {
            {{(int)(short)(bestEP[0] & 0xffffu), (int)(short)(bestEP[0] >> 16), (int)(short)(bestEP[1] & 0xffffu)},
             {(int)(short)(bestEP[1] >> 16), (int)(short)(bestEP[2] & 0xffffu), (int)(short)(bestEP[2] >> 16)}},
            {{(int)(short)(bestEP[3] & 0xffffu), (int)(short)(bestEP[3] >> 16), (int)(short)(bestEP[4] & 0xffffu)},
             {(int)(short)(bestEP[4] >> 16), (int)(short)(bestEP[5] & 0xffffu), (int)(short)(bestEP[5] >> 16)}}};
        // the quantised end points: a transformed mode stores deltas whose sum with the base equals the end point in its low
        // aPrec bits (the legality test, BC67.cpp:2597-2663), and an end point has no other bits (sign-extended when signed)
        int unqB[2][2][3], finB[2][2][3];
        const int maskB = (1 << aPrecB) - 1;
#pragma unroll
        for (int sb = 0; sb < 2; sb++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                int qv[2];
#pragma unroll
                for (int epi = 0; epi < 2; epi++)
                {
                    int v = enc[sb][epi][ch];
                    if (transformedB && (sb != 0 || epi != 0))
                    {
                        v = (v + enc[0][0][ch]) & maskB;
                        if (SIGNED)
                            v = (int)((u32)v << (32 - aPrecB)) >> (32 - aPrecB);
                    }
                    qv[epi] = v;
                }
                const bool sw = ((bestSwap >> sb) & 1u) != 0; // back to the order the round selected its indexes in
                const int q0 = sw ? qv[1] : qv[0], q1 = sw ? qv[0] : qv[1];
                unqB[sb][0][ch] = SIGNED ? unquantizeSigned(q0, aPrecB, finB[sb][0][ch]) : unquantizeUnsigned(q0, aPrecB, finB[sb][0][ch]);
                unqB[sb][1][ch] = SIGNED ? unquantizeSigned(q1, aPrecB, finB[sb][1][ch]) : unquantizeUnsigned(q1, aPrecB, finB[sb][1][ch]);
            }
        float originB[2][3], axisB[2][3];
        if (FAST)
        {
#pragma unroll
            for (int sb = 0; sb < 2; sb++)
            {
                float epDW[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    originB[sb][ch] = (float)finB[sb][0][ch];
                    epDW[ch] = ((float)finB[sb][1][ch] - originB[sb][ch]) * A.w[ch];
                }
                float lenSq = epDW[0] * epDW[0];
                lenSq = lenSq + epDW[1] * epDW[1];
                lenSq = lenSq + epDW[2] * epDW[2];
                lenSq = safeDenom(lenSq);
                const float mvdls = maxValueB / lenSq;
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    axisB[sb][ch] = epDW[ch] * A.w[ch] * mvdls;
            }
        }
        u32 myIdx = 0; // four indexes, 4 bits each: pixels 4 tw ... 4 tw + 3
#pragma unroll 1
        for (int j = 0; j < 4; j++)
        {
            const int px = 4 * tw + j;
            const bool s1 = ((pmask >> px) & 1u) != 0;
            u32 a, b;
            float lfp[3], pwp[3];
            pixLoad(px, a, b, lfp, pwp);
            const int c[3] = {(int)(short)(a & 0xffffu), (int)(short)(a >> 16), (int)(short)(b & 0xffffu)};
            int raw = 0;
            if (FAST)
            {
                float dist = ((float)c[0] - (s1 ? originB[1][0] : originB[0][0])) * (s1 ? axisB[1][0] : axisB[0][0]);
                dist = dist + ((float)c[1] - (s1 ? originB[1][1] : originB[0][1])) * (s1 ? axisB[1][1] : axisB[0][1]);
                dist = dist + ((float)c[2] - (s1 ? originB[1][2] : originB[0][2])) * (s1 ? axisB[1][2] : axisB[0][2]);
                raw = (int)clampRound(dist, maxValueB);
            }
            else
            {
                int e0[3], e1[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    e0[ch] = s1 ? unqB[1][0][ch] : unqB[0][0][ch];
                    e1[ch] = s1 ? unqB[1][1][ch] : unqB[0][1][ch];
                }
                const float l0 = lfp[0] * A.w[0], l1 = lfp[1] * A.w[1], l2 = lfp[2] * A.w[2];
                float be = 0.0f;
#pragma unroll 1
                for (int i = 0; i < 16; i++)
                {
                    const int weight = mad24(weightRcpB, i, 256) >> 9;
                    float d = l0 - twosCLHalfToFloat<SIGNED>(reconstructChannel<SIGNED>(e0[0], e1[0], weight)) * A.w[0];
                    float e = d * d;
                    d = l1 - twosCLHalfToFloat<SIGNED>(reconstructChannel<SIGNED>(e0[1], e1[1], weight)) * A.w[1];
                    e = e + d * d;
                    d = l2 - twosCLHalfToFloat<SIGNED>(reconstructChannel<SIGNED>(e0[2], e1[2], weight)) * A.w[2];
                    e = e + d * d;
                    const bool lt = (i == 0) || (i < rangeB && e < be);
                    raw = lt ? i : raw;
                    be = lt ? e : be;
                }
            }
            const bool sw = ((bestSwap >> (s1 ? 1 : 0)) & 1u) != 0;
            const u32 index = (u32)(sw ? (rangeB - 1) - raw : raw);
            myIdx |= index << (4 * j);
        }
        // pixels 0-7 are sub-lanes 0 and 1, 8-15 sub-lanes 2 and 3
        const u32 i0 = quadBcast<0>(myIdx), i1 = quadBcast<1>(myIdx), i2 = quadBcast<2>(myIdx), i3 = quadBcast<3>(myIdx);
        bestIdxLo = i0 | (i1 << 16);
        bestIdxHi = i2 | (i3 << 16);
    }
    // ---- header scatter + indexes (BC67.cpp:2992-3050, BC6H_IO: table from tools/gen_bc6h_layout.py) ----
    // (the block index again, from the refreshed lane number: held since the load it was spilled)
    const u32 blockIndexOut = blockIdx.x * 16u + (u32)blk;
    if (blockIndexOut < A.numBlocks && tw == 0)
    {
        const bool partitioned = T->bc6hModeInfo[bestMode][1] != 0;
        const int headerBits = partitioned ? 82 : 65;
        u32 fields[14];
        fields[0] = T->bc6hModeInfo[bestMode][0];
        fields[1] = (u32)bestPartition;
        // fields: rw rx ry rz | gw gx gy gz | bw bx by bz  (w,x = subset 0 ep 0,1; y,z = subset 1)
        const u32 e[2][2][3] = {
            {{bestEP[0] & 0xffffu, bestEP[0] >> 16, bestEP[1] & 0xffffu}, {bestEP[1] >> 16, bestEP[2] & 0xffffu, bestEP[2] >> 16}},
            {{bestEP[3] & 0xffffu, bestEP[3] >> 16, bestEP[4] & 0xffffu}, {bestEP[4] >> 16, bestEP[5] & 0xffffu, bestEP[5] >> 16}}};
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
        {
            fields[2 + ch * 4 + 0] = e[0][0][ch];
            fields[2 + ch * 4 + 1] = e[0][1][ch];
            fields[2 + ch * 4 + 2] = e[1][0][ch];
            fields[2 + ch * 4 + 3] = e[1][1][ch];
        }
        u64 lo = 0, hi = 0;
        for (int bit = 0; bit < headerBits; bit++)
        {
            const u32 code = T->bc6hLayout[bestMode][bit];
            u32 fv = 0;
#pragma unroll
            for (int f = 0; f < 14; f++)
                if ((code >> 4) == (u32)f)
                    fv = fields[f];
            const u64 b = (u64)((fv >> (code & 15u)) & 1u);
            if (bit < 64)
                lo |= b << bit;
            else
                hi |= b << (bit - 64);
        }
        int off = headerBits;
        const int fixupIndex1 = partitioned ? (int)T->anchor2[bestPartition & 31] : 0;
        const int ib = partitioned ? 3 : 4;
        const u64 idx = ((u64)bestIdxHi << 32) | bestIdxLo;
#pragma unroll
        for (int px = 0; px < 16; px++)
        {
            const int bits = (px == 0 || px == fixupIndex1) ? ib - 1 : ib;
            const u64 v = (idx >> (4 * px)) & 0xfull;
            if (off < 64)
            {
                lo |= v << off;
                if (off + bits > 64)
                    hi |= v >> (64 - off);
            }
            else
                hi |= v << (off - 64);
            off += bits;
        }
        uint4 o;
        o.x = (u32)lo;
        o.y = (u32)(lo >> 32);
        o.z = (u32)hi;
        o.w = (u32)(hi >> 32);
        *reinterpret_cast<uint4 *>(out + (size_t)blockIndexOut * 16u) = o;
    }
}

extern "C" hipError_t cvttmi_launch_bc6h(const void *d_blocks, void *d_out, const CvttBc6hArgs *args,
                                         const CvttDeviceTables *d_tables, int isSigned, hipStream_t stream)
{
    const uint32_t waves = (args->numBlocks + 15u) / 16u;
    if (waves == 0)
        return hipSuccess;
    const bool fast = (args->flags & CVTTMI_FLAG_BC6H_FAST_INDEXING) != 0;
#define CVTT_LAUNCH(S, Fq) hipLaunchKernelGGL((cvttmi_bc6h_kernel<S, Fq>), dim3(waves), dim3(64), 0, stream, (const uint8_t *)d_blocks, (uint8_t *)d_out, *args, d_tables)
    if (isSigned)
    {
        if (fast) CVTT_LAUNCH(true, true); else CVTT_LAUNCH(true, false);
    }
    else
    {
        if (fast) CVTT_LAUNCH(false, true); else CVTT_LAUNCH(false, false);
    }
#undef CVTT_LAUNCH
    return hipGetLastError();
}
#undef REFRESH_LANE
