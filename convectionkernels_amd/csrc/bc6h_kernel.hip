// BC6H (unsigned / signed half-float) endpoint search for gfx950.
//
// Replaces cvtt::Internal::BC6HComputer::Pack as reached from cvtt::Kernels::EncodeBC6HU /
// EncodeBC6HS (reference ConvectionKernels_API.cpp:56-84, ConvectionKernels_BC67.cpp:2665-3051;
// QuantizeEndpoints* 2503-2595, Evaluate*Legality 2597-2663, quantise/unquantise 2425-2501,
// IndexSelectorHDR.h, ParallelMath.h:996-1066).  Bit-identical to the reference's SSE2 lanes
// in its canonical build (zero-initialised automatics, program-order rounding
// scopes).
//
// Mapping: one LANE owns one block, a wave owns 8 reference groups (64 consecutive blocks).
// The reference couples the 8 lanes of a group in two places -- the duplicate-round skip
// (BC67.cpp:2853-2877, AllSet) and the mode-commit loop (2936-2984, AnySet) -- and both are
// evaluated in the reference's exact candidate order, so every loop here is wave-uniform and
// the two group predicates are 8-lane slices of a wave ballot.
//   * pixels: 2CL integers packed in 32 VGPRs; their weighted linear values are re-derived per
//     use (one v_cvt_f32_f16 + multiply) instead of living in 48 more registers;
//   * slow indexing: the weighted linear colours of the 8 / 16 interpolants of a round sit in
//     registers and every pixel scans them in order (strict '<', IndexSelectorHDR.h:125-139);
//     the anchor pixel goes first because its index decides the endpoint inversion, and a
//     round that turns out to be a duplicate never looks at the other pixels;
//   * the "meta round" results of a partition (errors and quantised end points of both subsets) live in LDS, [entry][lane]
//     so every access is conflict free; indexes are not kept (the winner's are selected again after the search) and nothing
//     goes to memory between the load of the block and the store of its 16 bytes.
//
// What is NOT searched ("Round 3"): nine of the ten two-subset modes delta-code three of their four end
// points, the reference's commit loop skips a (round, round, mode) triple whose deltas do not fit with a `continue` that
// changes no state (BC67.cpp:2954-2955), and whether a triple fits depends on the rounds' quantised end points alone --
// known before a round looks at a pixel.  So
//   * a round whose end points fit no mode in any lane of the wave runs without errors (`needError`), and not at all in the
//     last refine pass; a partition without a usable subset-0 round skips subset 1 and the commit loop (`usable0`);
//   * precisions of 8 bits and more are searched lazily (`lazy`): chains of both subsets without errors, then the exact set of
//     round pairs some lane could commit, then a replay of just those rounds with errors (a separate block of code);
// On content without structure that is two thirds of the reference's work; the output is bit-identical because every
// skipped piece is one the reference computes and then cannot use.
#include "cvtt_kernel_common.h"
#include <hip/hip_fp16.h>
#include <type_traits>

// Developer-only phase profile (-DCVTT_BC6H_PROFILE): wave cycles per phase, summed over waves.
#ifdef CVTT_BC6H_PROFILE
__device__ unsigned long long g_bc6hProf[16];
#define PROF_DECL unsigned long long profT = __builtin_readcyclecounter(); unsigned long long profAcc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PROF_MARK(slot) { const unsigned long long now = __builtin_readcyclecounter(); profAcc[slot] += now - profT; profT = now; }
#define PROF_FLUSH if (threadIdx.x == 0) { for (int i = 0; i < 8; i++) atomicAdd(&g_bc6hProf[i], profAcc[i]); }
#define PROF_COUNT(slot, n) { if (threadIdx.x == 0) atomicAdd(&g_bc6hProf[slot], (unsigned long long)(n)); }
extern "C" int cvttmi_bc6h_prof_read(unsigned long long *out)
{
    unsigned long long zero[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bc6hProf), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bc6hProf), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#else
#define PROF_DECL
#define PROF_MARK(slot)
#define PROF_FLUSH
#define PROF_COUNT(slot, n) {}
#endif

namespace
{
// ---- per-partition "meta round" state: everything stays on the chip (LDS, [entry][lane], conflict free) ----
//   EPQ : the quantised end points of every round of BOTH subsets (duplicate-round test, legality pass, commit).
//         two-subset precisions (<= 11 bits): two words per round -- r | g << 11 | b[9:0] << 22 per end point -- plus bit 10 of
//         the two blue values in a register (24 bits per subset); single-subset precisions (<= 16 bits): three words per
//         round (6 x int16).  2 x 12 x 2 = 48 entries (36 for the single-subset search)
//   ERR : 12 rounds x 2 subsets, binary32                                                        = 24 entries
// 72 entries x 256 B = 18 KB per wave: two waves per SIMD (8 per CU, 144 of 160 KB).  Indexes are not kept at all: the
// winner's are selected again, once, from its end points after the search (same deterministic scan), and the PCA seeds are
// computed per precision.  Round 3 kept the subset-0 history, the indexes of every round and the seeds in a 169 KB per wave
// HBM work buffer and 24 errors + spills in 256 B of private scratch: 104 GB of HBM traffic per 4096^2 image (FETCH_SIZE /
// WRITE_SIZE, profiles/r04) against 151 MB of algorithmic bytes.
constexpr int kEpqBase = 0, kEpqDwords = 48, kErrBase = 48, kMetaDwords = 72;
// Waves per SIMD the register allocator may assume.  The LDS footprint allows two; with 256 registers nothing spills and the
// pixels' weighted linear values stay in registers.  Measured with round 3's code: 2 waves 7.17, 3 waves 7.25, 4 waves
// (128 registers, 256 B of scratch) 7.70 Mblocks/s on noise -- the fourth wave bought 7 % and cost the HBM traffic above.
#ifndef CVTT_BC6H_WAVES
#define CVTT_BC6H_WAVES 2
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
        const u32 p = (u32)(DYN ? madI24(weight, diff, base) : mad24(weight, diff, base)) >> 6; // e < 2^16, weight <= 64: 0 <= 64 e0 + w (e1 - e0) + 32 < 2^22 + 32
        return (int)(mulU24(p & 0xffffu, 31u) >> 6);   // UnscaleHDRValueUnsigned
    }
}
template <bool SIGNED>
__device__ __forceinline__ int reconstructChannel(int e0, int e1, int weight)
{
    return reconstructFrom<SIGNED, false>(e0 * 64 + 32, e1 - e0, weight);
}

struct FetchHDR
{
    const u32 (&pk01)[16];
    const u32 (&pk2)[16];
    const float (&w)[4];
    template <int N>
    __device__ __forceinline__ void get(int px, float (&v)[N]) const
    {
        const u32 a = fetchPixel(pk01[px]);
        const u32 b = fetchPixel(pk2[px]);
        v[0] = (float)(int)(short)(a & 0xffffu) * w[0];
        v[1] = (float)(int)(short)(a >> 16) * w[1];
        v[2] = (float)(int)(short)(b & 0xffffu) * w[2];
    }
};

__device__ __forceinline__ u32 groupBits(u64 ballot, int lane) { return (u32)(ballot >> (lane & 56)) & 0xffu; }
} // namespace

template <bool SIGNED, bool FAST>
// Developer variant for the mapping A/B VERDICT r3 asks for (make VARIANT=b6_gw EXTRA=-DCVTT_BC6H_GROUPWAVE=1; never the shipped
// form): a wave is ONE reference group -- lane = candidate * 8 + block: eight partitions of each of the eight blocks are searched
// side by side (partition 8 j + candidate in step j), the single-subset search runs redundantly in the eight candidate lanes.
// The two group predicates stay the 8-lane ballot slices they are.  Per-lane partitions mean per-lane subset masks (every pixel
// step is predicated per lane), and because the mode loop of the commit depends on the group mates' running best at that point of
// the reference's candidate order (BC67.cpp:2936-2984), the eight candidates of a step still commit ONE AFTER THE OTHER, each
// followed by a broadcast of the block's best to its eight lanes.  profiles/r04/ab_bc6h_mapping.txt has the measurement.
#ifndef CVTT_BC6H_GROUPWAVE
#define CVTT_BC6H_GROUPWAVE 0
#endif
// Lowest two-subset precision that is searched lazily (chains without errors, then only the rounds of pairs that can be
// committed are evaluated).  8: where a round fits its own delta with probability 2^-8 or less on content without structure.
// 7 (the 7-bit mode, 6-bit deltas: one round in nine fits its own delta) was built and measured in round 5 with the per-lane
// replay below: 157.3 ms against 156.2 ms for config 3 (profiles/r05/ab_bc6h.txt) -- a lane that has one legal pair has many
// (neighbouring refine rounds quantise to neighbouring end points), so the wave's busiest lane replays about ten rounds per
// partition and the pair search + replay cost what the skipped errors saved.  6 is never lazy: its mode stores no deltas.
#ifndef CVTT_BC6H_LAZY_MIN
#define CVTT_BC6H_LAZY_MIN 8
#endif
// a partition whose per-lane replay needs more than this many rounds (both subsets, the wave's busiest lane each) switches the rest
// of the precision to the eager search: content whose deltas do fit
#ifndef CVTT_BC6H_LAZY_SWITCH
#define CVTT_BC6H_LAZY_SWITCH 6
#endif
#ifndef CVTT_BC6H_WG_WAVES
#define CVTT_BC6H_WG_WAVES 1 // waves per workgroup (independent of each other: each has its own 18 KB of the LDS block)
#endif
__global__ __launch_bounds__(64 * CVTT_BC6H_WG_WAVES, (CVTT_BC6H_WAVES + CVTT_BC6H_WG_WAVES - 1) / CVTT_BC6H_WG_WAVES) void cvttmi_bc6h_kernel(const uint8_t *__restrict__ blocks, uint8_t *__restrict__ out,
                                                         const CvttBc6hArgs A, const CvttDeviceTables *__restrict__ T)
{
    __shared__ u32 metaAll[CVTT_BC6H_WG_WAVES][kMetaDwords][64];
    u32 (&meta)[kMetaDwords][64] = metaAll[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    // error of round m of subset s of the current partition (indexed by wave-uniform loop counters)
    auto errAt = [&](int m, int s) -> float & { return reinterpret_cast<float &>(meta[kErrBase + m * 2 + s][lane]); };
    constexpr bool GW = CVTT_BC6H_GROUPWAVE != 0;
    static_assert(!GW || CVTT_BC6H_WG_WAVES == 1, "the group-wave variant is one wave per workgroup");
    const u32 blockIndex = GW ? blockIdx.x * 8u + (threadIdx.x & 7u) : blockIdx.x * (64u * CVTT_BC6H_WG_WAVES) + threadIdx.x;
    const bool valid = blockIndex < A.numBlocks;

    PROF_DECL
    // ---- load + clamp to the "2CL" domain (BC67.cpp:2691-2715) ----
    u32 pk01[16], pk2[16];
    {
        const uint2 *src = reinterpret_cast<const uint2 *>(blocks + (size_t)(valid ? blockIndex : 0u) * 128u);
#pragma unroll
        for (int px = 0; px < 16; px++)
        {
            const uint2 raw = src[px];
            int v[3] = {(int)(short)(raw.x & 0xffffu), (int)(short)(raw.x >> 16), (int)(short)(raw.y & 0xffffu)};
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
            }
            pk01[px] = ((u32)v[0] & 0xffffu) | ((u32)v[1] << 16);
            pk2[px] = (u32)v[2] & 0xffffu;
        }
    }
    const FetchHDR F = {pk01, pk2, A.w};
    // Does any pixel value of the wave have a zero exponent field (the patterns TwosCLHalfToFloat halves)?  Almost never,
    // and then the three fix-ups per pixel and round are skipped (unsigned format; wave-uniform).
    bool pixelFixup = true;
    if (!SIGNED)
    {
        bool z = false;
#pragma unroll
        for (int px = 0; px < 16; px++)
            z = z | ((pk01[px] & 0x7c00u) == 0) | ((pk01[px] & 0x7c000000u) == 0) | ((pk2[px] & 0x7c00u) == 0);
        pixelFixup = __ballot(z) != 0;
    }
    auto pixelToFloat = [&](int v16) -> float {
        if (SIGNED)
            return twosCLHalfToFloat<SIGNED>(v16);
        float f = __half2float(__ushort_as_half((unsigned short)v16));
        if (pixelFixup)
        {
            asm volatile("" ::: "memory"); // keep this a branch: as a select it costs what it is meant to save
            f = (v16 & 0x7c00) ? f : f * 0.5f;
        }
        return f;
    };

    // slow indexing: the linear value of every pixel channel (TwosCLHalfToFloat of the pixel, BC67.cpp:2711), converted once:
    // the error takes it as it is, the interpolant scan its product with the channel weight (floatPixelsLinearWeighted; a
    // plain multiplication per use).  48 of the 256 registers two waves per SIMD leave a lane.
    // (filled between the single-subset search, whose sixteen-entry interpolant table needs the registers, and the
    // partitioned one; the single-subset search converts per use)
    float lf[FAST ? 1 : 16][3];
    // the three linear values of a pixel converted on the spot, with ONE wave-uniform branch around the rare halvings
    auto pixelLinear = [&](u32 a, u32 b, float (&o)[3]) {
        const int v[3] = {(int)(short)(a & 0xffffu), (int)(short)(a >> 16), (int)(short)(b & 0xffffu)};
        if (SIGNED)
        {
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                o[ch] = twosCLHalfToFloat<SIGNED>(v[ch]);
            return;
        }
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
            o[ch] = __half2float(__ushort_as_half((unsigned short)v[ch]));
        if (pixelFixup)
        {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                o[ch] = (v[ch] & 0x7c00) ? o[ch] : o[ch] * 0.5f;
        }
    };

    int numTweakRounds = A.seedPoints < 1 ? 1 : (A.seedPoints > 4 ? 4 : A.seedPoints);
    int numRefineRounds = A.refineRounds < 1 ? 1 : (A.refineRounds > 3 ? 3 : A.refineRounds);

    // running best (BC67.cpp:2721-2733)
    float bestError = FLT_MAX;
    int bestMode = 0, bestPartition = 0;
    u32 bestEP[6] = {0, 0, 0, 0, 0, 0}; // [subset][3 dwords: (e0r|e0g<<16),(e0b|e1r<<16),(e1g|e1b<<16)]
    u32 bestSwap = 0; // bit s: the winning round of subset s exchanged its end points (anchor index in the upper half)

    // A mode of the current precision as one wave-uniform word (a scalar register): mode | transformed << 4 | the bits a
    // delta loses per channel (16 - bPrec) << 8, 16, 24.  Read from the table once per precision: a table lookup inside a
    // round is a per-lane memory load the round then waits for.
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

    // PCA seeds of a subset and the sums of its pre-weighted pixels (the reference precomputes them for every partition,
    // BC67.cpp:2738-2774; per precision here: 64 x 9 floats per block have no place on the chip, and the six passes cost
    // 4 % of the kernel where a 147 KB per wave work buffer cost 14 GB of HBM traffic per 4096^2 image)
    auto pcaSeeds = [&](u32 subsetMask, Unfinished &u, float (&sums)[3]) {
        Moments<3> m;
        pcaMomentsT<3>(F, subsetMask, m, sums);
        pcaFinishT<3>(F, subsetMask, A.w, m, u);
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
        const float rcpMaxIndex = T->rcpMaxIndex[indexBits];

        // ---- the quantised end points of a round as kept in LDS (layout at the top of the file) ----
        // wc: the two-subset form's two spare bits (bit 10 of the blue values), kept in a register per subset
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
        constexpr int epqStride = partitioned ? 2 : 3;
        // LDS entry of word 0 of round `m` of subset `s`
        auto epqEntry = [&](int s, int m) -> int { return kEpqBase + (s * 12 + m) * epqStride; };
        // the stored words of a round (xb: the subset's spare-bit register)
        auto loadEPQ = [&](int s, int m, u32 xb, int (&e)[2][3]) {
            const int at = epqEntry(s, m);
            const u32 wa = meta[at][lane], wb = meta[at + 1][lane];
            const u32 wc = partitioned ? ((xb >> (2 * m)) & 3u) : meta[at + 2][lane];
            unpackEPQ(wa, wb, wc, e);
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
            // Precisions whose modes all delta-code their end points tightly (8 bits and more: a lane's round fits a mode
            // with probability 2^-8 ... 2^-18 on content without structure) are searched LAZILY: first the chains of both
            // subsets without any error (indexes only as far as the refiners need them), then the exact set of
            // (subset-0 round, subset-1 round) pairs some lane could commit, and only the rounds of that set are evaluated
            // again, with errors (the replay below: code of its own, so that the chains' registers and branches do not know
            // about it).  The first partition of a precision that needs a replay switches the rest of the precision back to
            // the eager search (content whose deltas do fit).
            bool eagerNow = !(partitioned && aPrec >= CVTT_BC6H_LAZY_MIN);
            // A precision with ONE mode (7, 9, 10 bits) has no coupling between the lanes of a group in its commit loop: a lane
            // commits a pair iff the pair beats its best and is legal, whatever its group mates do (the mode loop of
            // BC67.cpp:2936-2984 has a single iteration).  So each lane needs the errors of the rounds of ITS legal pairs only,
            // and the replay below evaluates a different round in every lane.  With three modes (8, 11 bits) a lane that is
            // better but illegal keeps the mode loop of its group going, so every lane needs the errors of every round in the
            // wave's set: there the replay stays wave-uniform.
            const bool perLaneReplay = !GW && numModesHere == 1;

            for (int pStep = 0; pStep < ((GW && partitioned) ? 4 : numPartitions); pStep++)
            {
                const int p = (GW && partitioned) ? pStep * 8 + (lane >> 3) : pStep; // GW: a partition per candidate lane
                const u32 partitionMask = partitioned ? T->partition2[p] : 0u;
                const bool lazy = !eagerNow;
                if (partitioned && aPrec >= 8) { if (lazy) { PROF_COUNT(13, 1) } else { PROF_COUNT(11, 1) } }
                u32 cand0 = 0;   // lazy: rounds of subset 0 whose own delta fits a mode in THIS lane
                u32 invBits = 0; // bit subset * 12 + round = the round swapped its end points (anchor index in the upper half)
                u32 xbits0 = 0, xbits1 = 0; // the spare bits of the packed end points (two per round)
                u32 roundValid0 = 0xfffu, roundValid1 = 0xfffu; // per group (identical in its 8 lanes)
                // Rounds of subset 0 whose quantised end points fit the delta coding of a mode of this precision in at
                // least one lane of the wave.  A block can only be committed with such a round (the legality test of
                // BC67.cpp:2597-2663 includes subset 0's own delta), the commit loop changes no state without a commit,
                // and the legality is known before a round's pixels are looked at.  So a round nobody can use needs its
                // indexes only as far as the next refine pass needs them (no error; nothing at all in the last pass), and
                // a partition in which no round of subset 0 is usable needs neither subset 1 nor the commit loop.
                u32 usable0 = 0;

                PROF_MARK(6)
                for (int subset = 0; subset < numSubsets; subset++)
                {
                    const u32 subsetMask = partitioned ? (subset ? partitionMask : (~partitionMask & 0xffffu)) : 0xffffu;
                    if (subset == 1 && usable0 == 0)
                        break;
                    const int fixupIndex = GW ? ((subset == 0) ? 0 : (int)T->anchor2[p]) : __builtin_amdgcn_readfirstlane((subset == 0) ? 0 : (int)T->anchor2[p]);
                    // the anchor pixel of the subset (its index decides the swap of a round's end points, BC67.cpp:2525-2547)
                    u32 fa = 0, fb = 0;
                    float fixLw[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int px = 0; px < 16; px++)
                        if (px == fixupIndex)
                        {
                            fa = pk01[px];
                            fb = pk2[px];
                        }
                    if (!FAST)
                    {
                        float fl[3];
                        pixelLinear(fa, fb, fl);
                        fixLw[0] = fl[0] * A.w[0];
                        fixLw[1] = fl[1] * A.w[1];
                        fixLw[2] = fl[2] * A.w[2];
                    }
                    const int count = __popc(subsetMask);
                    const float wRcp = T->rcpTable[count];
                    const float wCount = (float)count;

                    Unfinished ufep;
                    // the refiner's sums of the pre-weighted member pixels (EndpointRefiner.h:78-92) do not depend on the indexes:
                    // the PCA's first pass forms the same sums in the same order, so a round takes them from here instead of
                    // adding them up pixel by pixel
                    float vsSubset[3];
                    pcaSeeds(subsetMask, ufep, vsSubset);

                    PROF_MARK(0)
                    for (int tweak = 0; tweak < 4; tweak++)
                    {
                        // EndpointRefiner<3> refiners[2]: fresh (zero in the canonical build) per tweak
                        float tv[3] = {0, 0, 0}, vs[3] = {0, 0, 0}, tt = 0.0f, ts = 0.0f;
                        int refCount = 0; // contributions of the previous round
                        bool abortRemaining = false;

                        for (int refinePass = 0; refinePass < 3; refinePass++)
                        {
                            const int metaRound = tweak * 3 + refinePass;
                            if (tweak >= numTweakRounds || refinePass >= numRefineRounds)
                                abortRemaining = true;
                            if (abortRemaining)
                            {
                                if (subset == 0) roundValid0 &= ~(1u << metaRound); else roundValid1 &= ~(1u << metaRound);
                                // never written by the reference: the canonical build's zero-initialised automatics
                                // (fresh for every partition, BC67.cpp:2797-2801), which later rounds compare against
                                // (the spare bits of such a round stay zero in xbits0 / xbits1)
                                meta[epqEntry(subset, metaRound)][lane] = 0;
                                meta[epqEntry(subset, metaRound) + 1][lane] = 0;
                                if (!partitioned)
                                    meta[epqEntry(subset, metaRound) + 2][lane] = 0;
                                continue;
                            }

                            // ---- endpoints in colour space ----
                            int epCS[2][3];
                            if (refinePass == 0)
                            {
                                const float tf0 = T->tweakFactors[indexBits - 2][tweak][0];
                                const float tf1 = T->tweakFactors[indexBits - 2][tweak][1];
#pragma unroll
                                for (int ch = 0; ch < 3; ch++)
                                {
                                    // FinishHDRSigned / Unsigned, UnfinishedEndpoints.h:39-75
                                    const float lo = SIGNED ? -31743.0f : 0.0f;
                                    const float f0 = sseMax(sseMin(ufep.base[ch] + ufep.offset[ch] * tf0, 31743.0f), lo);
                                    const float f1 = sseMax(sseMin(ufep.base[ch] + ufep.offset[ch] * tf1, 31743.0f), lo);
                                    epCS[0][ch] = (int)rintf(f0);
                                    epCS[1][ch] = (int)rintf(f1);
                                }
                            }
                            else
                            {
                                // EndpointRefiner::GetRefinedEndpointsHDR (EndpointRefiner.h:99-175) from the
                                // previous round's sums (empty when that round was skipped as a duplicate)
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

                            PROF_MARK(1)
                            // ---- index selection, one pixel at a time: IndexSelectorHDR.h:100-144 ----
                            const u32 sm = GW ? subsetMask : opaqueUniform(subsetMask);
                            bool interpFixup = true; // may an interpolant of this round have a zero exponent field? (wave-uniform)
                            float iw[indexRange][3]; // slow: weighted linear colour of every interpolant
                            float origin[3], axis[3]; // fast: projection axis
                            if (FAST)
                            {
                                // IndexSelector::Init on the colour-space endpoints + SelectIndexLDR
                                float epDW[3];
#pragma unroll
                                for (int ch = 0; ch < 3; ch++)
                                {
                                    origin[ch] = (float)fin[0][ch];
                                    epDW[ch] = ((float)fin[1][ch] - origin[ch]) * A.w[ch];
                                }
                                float lenSq = epDW[0] * epDW[0];
                                lenSq = lenSq + epDW[1] * epDW[1];
                                lenSq = lenSq + epDW[2] * epDW[2];
                                lenSq = safeDenom(lenSq);
                                const float mvdls = maxValue / lenSq;
#pragma unroll
                                for (int ch = 0; ch < 3; ch++)
                                    axis[ch] = epDW[ch] * A.w[ch] * mvdls;
                            }
                            else
                            {
                                // An interpolant lies between the two finished end points (the interpolation, the
                                // unscaling and the conversion are monotone), so when no lane's end point has a zero exponent
                                // field no interpolant has one and the halving of TwosCLHalfToFloat is skipped (unsigned format)
                                if (!SIGNED)
                                {
                                    int lowest = fin[0][0] < fin[1][0] ? fin[0][0] : fin[1][0];
#pragma unroll
                                    for (int ch = 1; ch < 3; ch++)
                                    {
                                        lowest = fin[0][ch] < lowest ? fin[0][ch] : lowest;
                                        lowest = fin[1][ch] < lowest ? fin[1][ch] : lowest;
                                    }
                                    interpFixup = __ballot(lowest < 0x400) != 0;
                                }
                                if (interpFixup)
                                {
#pragma unroll
                                    for (int i = 0; i < indexRange; i++)
                                    {
                                        const int weight = mad24(weightRcp, i, 256) >> 9;
#pragma unroll
                                        for (int ch = 0; ch < 3; ch++)
                                            iw[i][ch] = twosCLHalfToFloat<SIGNED>(reconstructChannel<SIGNED>(unq[0][ch], unq[1][ch], weight)) * A.w[ch];
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
                                            iw[i][ch] = __half2float(__ushort_as_half((unsigned short)reconstructChannel<SIGNED>(unq[0][ch], unq[1][ch], weight))) * A.w[ch];
                                    }
                                }
                            }
                            int recBase[3], recDiff[3];
#pragma unroll
                            for (int ch = 0; ch < 3; ch++)
                            {
                                recBase[ch] = unq[0][ch] * 64 + 32;
                                recDiff[ch] = unq[1][ch] - unq[0][ch];
                            }
                            // raw (un-inverted) index of the pixel packed in (a, b); SelectIndexHDRSlow keeps the FIRST minimum
                            // lwp: the pixel's weighted linear values (slow indexing only)
                            auto rawIndexOf = [&](u32 a, u32 b, const float (&lwp)[3]) -> int {
                                if (FAST)
                                {
                                    const int c0 = (int)(short)(a & 0xffffu), c1 = (int)(short)(a >> 16), c2 = (int)(short)(b & 0xffffu);
                                    float dist = ((float)c0 - origin[0]) * axis[0];
                                    dist = dist + ((float)c1 - origin[1]) * axis[1];
                                    dist = dist + ((float)c2 - origin[2]) * axis[2];
                                    return (int)clampRound(dist, maxValue);
                                }
                                const float l0 = lwp[0];
                                const float l1 = lwp[1];
                                const float l2 = lwp[2];
                                float be = 0.0f;
                                int bi = 0;
#pragma unroll
                                for (int i = 0; i < indexRange; i++)
                                    {
                                        float d = l0 - iw[i][0];
                                        float e = d * d;
                                        d = l1 - iw[i][1];
                                        e = e + d * d;
                                        d = l2 - iw[i][2];
                                        e = e + d * d;
                                        if (i == 0)
                                            be = e;
                                        else
                                        {
                                            // be = sseMin(be, e) with the index taken along on the same comparison (the errors
                                            // are sums of squares: no NaN, no -0, so equal values are equal bits)
                                            // (the minimum as v_min_f32, the comparison only for the index: the select form made
                                            // every step wait for the previous one's compare -> select, with a hazard nop between)
                                            const bool lt = e < be;
                                            bi = lt ? i : bi;
                                            be = __builtin_fminf(be, e);
                                        }
                                    }
                                return bi;
                            };

                            const int fixRaw = rawIndexOf(fetchPixel(fa), fetchPixel(fb), fixLw);
                            PROF_MARK(2)
                            const bool invert = (indexRange / 2 - 1) < fixRaw;
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
                            u32 qa, qb, qc;
                            packEPQ(q, qa, qb, qc);
                            bool needError = true; // wave-uniform
                            if (subset == 0)
                            {
                                bool fits = ownDeltaFits(q, modeW0, aPrec);
                                if (numModesHere > 1)
                                    fits = fits | ownDeltaFits(q, modeW1, aPrec);
                                if (numModesHere > 2)
                                    fits = fits | ownDeltaFits(q, modeW2, aPrec);
                                needError = __ballot(fits) != 0;
                                if (needError)
                                    usable0 |= 1u << metaRound;
                                if (lazy && fits)
                                    cand0 |= 1u << metaRound;
                            }
                            if (lazy)
                                needError = false;
                            if (invert)
                                invBits |= 1u << (subset * 12 + metaRound);
                            // ---- duplicate-round test against every earlier meta round of this subset (group-wide) ----
                            // Only a group whose eight lanes ALL repeat an earlier round skips the round, so the first of the three
                            // words is compared alone, and the other two only if some group matches in it everywhere.
                            const int epq0 = epqEntry(subset, 0), epqNow = epqEntry(subset, metaRound);
                            const u32 xbPrev = subset == 0 ? xbits0 : xbits1;
                            bool anySame = false;
                            for (int prev = 0; prev < metaRound; prev++)
                                anySame = anySame || (meta[epq0 + prev * epqStride][lane] == qa);
                            meta[epqNow][lane] = qa;
                            meta[epqNow + 1][lane] = qb;
                            if (partitioned)
                            {
                                if (subset == 0) xbits0 |= qc << (2 * metaRound); else xbits1 |= qc << (2 * metaRound);
                            }
                            else
                                meta[epqNow + 2][lane] = qc;
                            bool groupAllSame = false;
                            if (metaRound > 0)
                            {
                                u64 g = __ballot(anySame);
                                g &= g >> 1;
                                g &= g >> 2;
                                g &= g >> 4;
                                if ((g & 0x0101010101010101ull) != 0)
                                {
                                    anySame = false;
                                    for (int prev = 0; prev < metaRound; prev++)
                                    {
                                        const u32 pc = partitioned ? ((xbPrev >> (2 * prev)) & 3u) : meta[epq0 + prev * epqStride + 2][lane];
                                        anySame = anySame || (meta[epq0 + prev * epqStride][lane] == qa && meta[epq0 + prev * epqStride + 1][lane] == qb && pc == qc);
                                    }
                                    groupAllSame = groupBits(__ballot(anySame), lane) == 0xffu;
                                }
                            }
                            PROF_MARK(3)
                            if (groupAllSame)
                            {
                                if (subset == 0) roundValid0 &= ~(1u << metaRound); else roundValid1 &= ~(1u << metaRound);
                                // (its indexes are never read: only valid rounds reach the commit)
                            }
                            else if (!needError && refinePass == numRefineRounds - 1)
                            {
                                // nobody can use this round and no refine pass follows it
                                errAt(metaRound, subset) = FLT_MAX;
                            }
                            else
                            {
                                // ---- error and refiner sums in pixel order (BC67.cpp:2879-2909); the indexes themselves are not kept ----
                                float subsetError = needError ? 0.0f : FLT_MAX;
#pragma unroll
                                for (int px = 0; px < 16; px++)
                                    if ((sm >> px) & 1u)
                                    {
                                        const u32 a = fetchPixel(pk01[px]), b = fetchPixel(pk2[px]);
                                        float lfp[3] = {0.0f, 0.0f, 0.0f};
                                        if (!FAST)
                                        {
                                            if (partitioned)
                                            {
                                                lfp[0] = lf[FAST ? 0 : px][0];
                                                lfp[1] = lf[FAST ? 0 : px][1];
                                                lfp[2] = lf[FAST ? 0 : px][2];
                                            }
                                            else
                                                pixelLinear(a, b, lfp);
                                        }
                                        // (the anchor's scan has been done: its index decided the inversion)
                                        int raw;
                                        if (px == fixupIndex)
                                            raw = fixRaw;
                                        else
                                        {
                                            const float lwp[3] = {lfp[0] * A.w[0], lfp[1] * A.w[1], lfp[2] * A.w[2]};
                                            raw = rawIndexOf(a, b, lwp);
                                        }
                                        const int index = invert ? (indexRange - 1) - raw : raw;

                                        const int weight = (int)((mulU24((u32)raw, (u32)weightRcp) + 256u) >> 9);
                                        const int orig[3] = {(int)(short)(a & 0xffffu), (int)(short)(a >> 16), (int)(short)(b & 0xffffu)};
                                        if (needError)
                                        {
                                        float err = 0.0f;
                                        int rec[3];
                                        float recF[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                                        for (int ch = 0; ch < 3; ch++)
                                        {
                                            rec[ch] = reconstructFrom<SIGNED>(recBase[ch], recDiff[ch], weight);
                                            if (!FAST)
                                                recF[ch] = SIGNED ? twosCLHalfToFloat<SIGNED>(rec[ch]) : __half2float(__ushort_as_half((unsigned short)rec[ch]));
                                        }
                                        if (!FAST && !SIGNED && interpFixup)
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
                                                const float d = recF[ch] - lfp[ch];
                                                sq = d * d;
                                            }
                                            err = err + sq * A.wSq[ch]; // (Flags::Uniform: the weights are 1.0 and the product is exact, no second form needed)
                                        }
                                        subsetError = subsetError + err;
                                        }

                                        if (refinePass != numRefineRounds - 1)
                                        {
                                            const float t = (float)index * rcpMaxIndex;
#pragma unroll
                                            for (int ch = 0; ch < 3; ch++)
                                            {
                                                const float v = (float)orig[ch] * A.w[ch];
                                                tv[ch] = tv[ch] + t * v;
                                            }
                                            tt = tt + t * t;
                                            ts = ts + t;
                                            refCount++;
                                        }
                                    }
                                if (refinePass != numRefineRounds - 1)
                                {
#pragma unroll
                                    for (int ch = 0; ch < 3; ch++)
                                        vs[ch] = vsSubset[ch];
                                }
                                errAt(metaRound, subset) = subsetError;
                            }
                            PROF_MARK(4)
                        }
                    }
                }

                if (lazy && usable0 != 0)
                {
                    // ---- which pairs of rounds could some lane commit? (Evaluate*Legality, BC67.cpp:2597-2663, on end points alone) ----
                    u32 need0 = 0, need1 = 0;
                    u32 c0 = cand0 & roundValid0;
                    PROF_COUNT(8, 1)
                    PROF_COUNT(12, __popcll(__ballot(c0 != 0)))
                    while (__ballot(c0 != 0) != 0)
                    {
                        PROF_COUNT(9, 1)
                        const bool act = c0 != 0;
                        const int m0 = act ? __builtin_ctz(c0) : 0;
                        c0 &= c0 - 1u;
                        int e0[2][3];
                        loadEPQ(0, m0, xbits0, e0); // (m0 differs from lane to lane: the entry index is per lane, the bank is the lane's)
                        const bool own0 = ownDeltaFits(e0, modeW0, aPrec);
                        const bool own1 = numModesHere > 1 && ownDeltaFits(e0, modeW1, aPrec);
                        const bool own2 = numModesHere > 2 && ownDeltaFits(e0, modeW2, aPrec);
                        // (the lanes with a candidate are few: one or two of 64 on content without structure, and each fits one
                        // mode as a rule, so every test is followed by a wave-wide "is anybody still in?")
                        if (__ballot(act && (own0 || own1 || own2)) == 0)
                            continue;
                        for (int m1 = 0; m1 < 12; m1++)
                        {
                            int x[2][3];
                            loadEPQ(1, m1, xbits1, x);
                            const bool in = act && ((roundValid1 >> m1) & 1u);
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
                    // ---- replay: the rounds of the set again, from their quantised end points, this time with errors ----
                    // `rm`: the round of subset `rs` (wave-uniform) this lane evaluates -- the same in every lane (three-mode
                    // precisions) or each lane's own (single-mode precisions); `act`: the lane has one
                    auto replayRound = [&](int rs, int rm, bool act) {
                        const u32 rmask = GW ? (rs ? partitionMask : (~partitionMask & 0xffffu)) : opaqueUniform(rs ? partitionMask : (~partitionMask & 0xffffu));
                        // the end points as the round had them before it swapped them (the scan's first-minimum rule sees the order)
                        const bool was = ((invBits >> (rs * 12 + rm)) & 1u) != 0;
                        int rq[2][3];
                        loadEPQ(rs, rm, rs ? xbits1 : xbits0, rq);
                        const int (&s0)[3] = rq[0];
                        const int (&s1)[3] = rq[1];
                        int unq[2][3], fin[2][3];
#pragma unroll
                        for (int ch = 0; ch < 3; ch++)
                        {
                            const int q0 = was ? s1[ch] : s0[ch], q1 = was ? s0[ch] : s1[ch];
                            unq[0][ch] = SIGNED ? unquantizeSigned(q0, aPrec, fin[0][ch]) : unquantizeUnsigned(q0, aPrec, fin[0][ch]);
                            unq[1][ch] = SIGNED ? unquantizeSigned(q1, aPrec, fin[1][ch]) : unquantizeUnsigned(q1, aPrec, fin[1][ch]);
                        }
                        float iw[indexRange][3];
                        float origin[3], axis[3];
                        if (FAST)
                        {
                            float epDW[3];
#pragma unroll
                            for (int ch = 0; ch < 3; ch++)
                            {
                                origin[ch] = (float)fin[0][ch];
                                epDW[ch] = ((float)fin[1][ch] - origin[ch]) * A.w[ch];
                            }
                            float lenSq = epDW[0] * epDW[0];
                            lenSq = lenSq + epDW[1] * epDW[1];
                            lenSq = lenSq + epDW[2] * epDW[2];
                            lenSq = safeDenom(lenSq);
                            const float mvdls = maxValue / lenSq;
#pragma unroll
                            for (int ch = 0; ch < 3; ch++)
                                axis[ch] = epDW[ch] * A.w[ch] * mvdls;
                        }
                        else
                        {
#pragma unroll
                            for (int i = 0; i < indexRange; i++)
                            {
                                const int weight = mad24(weightRcp, i, 256) >> 9;
#pragma unroll
                                for (int ch = 0; ch < 3; ch++)
                                    iw[i][ch] = twosCLHalfToFloat<SIGNED>(reconstructChannel<SIGNED>(unq[0][ch], unq[1][ch], weight)) * A.w[ch];
                            }
                        }
                        float subsetError = 0.0f;
#pragma unroll 1
                        for (int px = 0; px < 16; px++)
                        {
                            if (((rmask >> px) & 1u) == 0)
                                continue;
                            u32 a = 0, b = 0;
#pragma unroll
                            for (int k = 0; k < 16; k++)
                                if (k == px)
                                {
                                    a = pk01[k];
                                    b = pk2[k];
                                }
                            float lf[3] = {0.0f, 0.0f, 0.0f};
                            int raw;
                            if (FAST)
                            {
                                const int c0 = (int)(short)(a & 0xffffu), c1 = (int)(short)(a >> 16), c2 = (int)(short)(b & 0xffffu);
                                float dist = ((float)c0 - origin[0]) * axis[0];
                                dist = dist + ((float)c1 - origin[1]) * axis[1];
                                dist = dist + ((float)c2 - origin[2]) * axis[2];
                                raw = (int)clampRound(dist, maxValue);
                            }
                            else
                            {
                                lf[0] = pixelToFloat((int)(short)(a & 0xffffu));
                                lf[1] = pixelToFloat((int)(short)(a >> 16));
                                lf[2] = pixelToFloat((int)(short)(b & 0xffffu));
                                const float l0 = lf[0] * A.w[0], l1 = lf[1] * A.w[1], l2 = lf[2] * A.w[2];
                                float be = 0.0f;
                                raw = 0;
#pragma unroll
                                for (int i = 0; i < indexRange; i++)
                                {
                                    float d = l0 - iw[i][0];
                                    float e = d * d;
                                    d = l1 - iw[i][1];
                                    e = e + d * d;
                                    d = l2 - iw[i][2];
                                    e = e + d * d;
                                    const bool lt = (i == 0) || (e < be);
                                    raw = lt ? i : raw;
                                    be = (i == 0) ? e : __builtin_fminf(be, e);
                                }
                            }
                            const int weight = (int)((mulU24((u32)raw, (u32)weightRcp) + 256u) >> 9);
                            const int orig[3] = {(int)(short)(a & 0xffffu), (int)(short)(a >> 16), (int)(short)(b & 0xffffu)};
                            float err = 0.0f;
#pragma unroll
                            for (int ch = 0; ch < 3; ch++)
                            {
                                const int rec = reconstructChannel<SIGNED>(unq[0][ch], unq[1][ch], weight);
                                float sq;
                                if (FAST)
                                {
                                    const int r16 = (int)(short)rec;
                                    const u32 du = (u32)((r16 > orig[ch] ? r16 : orig[ch]) - (r16 > orig[ch] ? orig[ch] : r16)) & 0xffffu;
                                    sq = (float)(int)(du * du);
                                }
                                else
                                {
                                    const float d = twosCLHalfToFloat<SIGNED>(rec) - lf[ch];
                                    sq = d * d;
                                }
                                err = err + sq * A.wSq[ch]; // (Flags::Uniform: the weights are 1.0 and the product is exact, no second form needed)
                            }
                            subsetError = subsetError + err;
                        }
                        if (act)
                            errAt(rm, rs) = subsetError;
                    };
                    if (__ballot(need0 != 0) == 0)
                        continue; // nobody can commit anything with this partition at this precision
                    PROF_COUNT(10, 1)
                    if (perLaneReplay)
                    {
                        // every lane its own rounds: as many passes as the wave's busiest lane has rounds to evaluate
                        int passes = 0;
#pragma unroll 1
                        for (int rs = 0; rs < 2; rs++)
                        {
                            u32 todo = rs ? need1 : need0;
                            while (__ballot(todo != 0) != 0)
                            {
                                const bool act = todo != 0;
                                const int rm = act ? __builtin_ctz(todo) : 0;
                                todo &= todo - 1u;
                                replayRound(rs, rm, act);
                                passes++;
                            }
                        }
                        if (passes > CVTT_BC6H_LAZY_SWITCH)
                            eagerNow = true;
                    }
                    else
                    {
                        u32 replayRounds = 0; // wave-uniform: bit subset * 12 + round
#pragma unroll
                        for (int m = 0; m < 12; m++)
                        {
                            if (__ballot((need0 >> m) & 1u) != 0) replayRounds |= 1u << m;
                            if (__ballot((need1 >> m) & 1u) != 0) replayRounds |= 1u << (12 + m);
                        }
                        eagerNow = true;
                        for (u32 todo = replayRounds; todo != 0; todo &= todo - 1u)
                        {
                            const int bit = __builtin_ctz(todo);
                            const int rs = bit >= 12 ? 1 : 0;
                            replayRound(rs, bit - 12 * rs, true);
                        }
                    }
                }

                // ---- delta-coding legality + commit, BC67.cpp:2914-2986 ----
                if (usable0 == 0)
                    continue;
                const int numMeta1 = partitioned ? 12 : 1;
                // cheapest valid subset-1 round: no combination with meta0 can beat the best unless this one does
                float minErr1 = 0.0f;
                if (partitioned)
                {
                    minErr1 = FLT_MAX;
                    for (int m = 0; m < 12; m++)
                    {
                        const float e = errAt(m, 1);
                        if (((roundValid1 >> m) & 1u) && e < minErr1)
                            minErr1 = e;
                    }
                }
                for (int cand = 0; cand < ((GW && partitioned) ? 8 : 1); cand++)
                {
                const bool act = !(GW && partitioned) || (lane >> 3) == cand; // GW: the candidates commit in the reference's order
                for (int meta0 = 0; meta0 < 12; meta0++)
                {
                    const bool valid0 = act && ((roundValid0 >> meta0) & 1u) != 0;
                    const float err0 = errAt(meta0, 0);
                    const bool canBeat = valid0 && ((partitioned ? err0 + minErr1 : err0) < bestError);
                    if (__ballot(canBeat) == 0)
                        continue;
                    // quantised endpoints of subset 0's round, and whether its own delta fits each mode of this precision
                    int e0[2][3];
                    bool legal0[3] = {true, true, true};
                    {
                        loadEPQ(0, meta0, xbits0, e0);
                        legal0[0] = ownDeltaFits(e0, modeW0, aPrec);
                        if (numModesHere > 1)
                            legal0[1] = ownDeltaFits(e0, modeW1, aPrec);
                        if (numModesHere > 2)
                            legal0[2] = ownDeltaFits(e0, modeW2, aPrec);
                    }
                    // a lane whose subset-0 delta fits no mode cannot commit with this meta0 whatever meta1 is
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
                            loadEPQ(1, meta1, xbits1, e1);

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
                                    bestMode = mode;
                                    bestPartition = p;
                                    bestEP[0] = ((u32)enc[0][0][0] & 0xffffu) | ((u32)enc[0][0][1] << 16);
                                    bestEP[1] = ((u32)enc[0][0][2] & 0xffffu) | ((u32)enc[0][1][0] << 16);
                                    bestEP[2] = ((u32)enc[0][1][1] & 0xffffu) | ((u32)enc[0][1][2] << 16);
                                    if (partitioned)
                                    {
                                        bestEP[3] = ((u32)enc[1][0][0] & 0xffffu) | ((u32)enc[1][0][1] << 16);
                                        bestEP[4] = ((u32)enc[1][0][2] & 0xffffu) | ((u32)enc[1][1][0] << 16);
                                        bestEP[5] = ((u32)enc[1][1][1] & 0xffffu) | ((u32)enc[1][1][2] << 16);
                                    }
                                    // (the indexes of the two rounds are selected again after the search, from these end
                                    // points in the order the rounds had them)
                                    bestSwap = ((invBits >> meta0) & 1u) | (partitioned ? ((invBits >> (12 + meta1)) & 1u) << 1 : 0u);
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
                if (GW && partitioned)
                {
                    // the block's best after this candidate's partition, to all eight lanes of the block
                    const int src = cand * 8 + (lane & 7);
                    bestError = __shfl(bestError, src);
                    bestMode = __shfl(bestMode, src);
                    bestPartition = __shfl(bestPartition, src);
                    bestSwap = (u32)__shfl((int)bestSwap, src);
#pragma unroll
                    for (int i = 0; i < 6; i++)
                        bestEP[i] = (u32)__shfl((int)bestEP[i], src);
                }
                } // candidates
            }
        }
    };
    searchAll(std::false_type{});
    if (!FAST)
    {
#pragma unroll
        for (int px = 0; px < 16; px++)
            pixelLinear(pk01[px], pk2[px], lf[px]);
    }
    searchAll(std::true_type{});

    PROF_MARK(5)
    PROF_FLUSH
    // ---- the winner's indexes, selected again from its end points.  A round's indexes are a function of its quantised end
    // points in the order it had them before the anchor swap, of the precision and of the pixels (QuantizeEndpoints* +
    // SelectIndexHDR*, BC67.cpp:2503-2595, 2879-2893): the same operations on the same values here, once per block, instead of
    // 48 words of index history per partition and precision in memory. ----
    u32 bestIdxLo = 0, bestIdxHi = 0;
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
#pragma unroll
        for (int px = 0; px < 16; px++)
        {
            const bool s1 = ((pmask >> px) & 1u) != 0;
            const u32 a = fetchPixel(pk01[px]), b = fetchPixel(pk2[px]);
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
                const float l0 = lf[FAST ? 0 : px][0] * A.w[0], l1 = lf[FAST ? 0 : px][1] * A.w[1], l2 = lf[FAST ? 0 : px][2] * A.w[2];
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
            if (px < 8)
                bestIdxLo |= index << (4 * px);
            else
                bestIdxHi |= index << (4 * (px - 8));
        }
    }
    // ---- header scatter + indexes (BC67.cpp:2992-3050, BC6H_IO: table from tools/gen_bc6h_layout.py) ----
    if (valid && (!GW || (threadIdx.x >> 3) == 0))
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
        *reinterpret_cast<uint4 *>(out + (size_t)blockIndex * 16u) = o;
    }
}

extern "C" hipError_t cvttmi_launch_bc6h(const void *d_blocks, void *d_out, const CvttBc6hArgs *args,
                                         const CvttDeviceTables *d_tables, int isSigned, hipStream_t stream)
{
    const uint32_t waves = CVTT_BC6H_GROUPWAVE ? (args->numBlocks + 7u) / 8u : (args->numBlocks + 63u) / 64u;
    if (waves == 0)
        return hipSuccess;
    const bool fast = (args->flags & CVTTMI_FLAG_BC6H_FAST_INDEXING) != 0;
#define CVTT_LAUNCH(S, Fq) hipLaunchKernelGGL((cvttmi_bc6h_kernel<S, Fq>), dim3((waves + CVTT_BC6H_WG_WAVES - 1) / CVTT_BC6H_WG_WAVES), dim3(64 * CVTT_BC6H_WG_WAVES), 0, stream, (const uint8_t *)d_blocks, (uint8_t *)d_out, *args, d_tables)
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
