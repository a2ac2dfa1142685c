// ETC2 RGB / RGBA (EAC alpha) encoders for gfx950.
//
// Replaces cvtt::Internal::ETCComputer::CompressETC2Block (punchthroughAlpha = false) and
// CompressETC2AlphaBlock as reached from cvtt::Kernels::EncodeETC2 / EncodeETC2RGBA /
// EncodeETC2Alpha (reference ConvectionKernels_API.cpp:216-229, 246-256, 270-286;
// ConvectionKernels_ETC.cpp: EncodePlanar 1274-1662, sector split 1723-1848, EncodeTMode 396-647,
// EncodeHMode 649-885, CompressETC1BlockInternal 2624-2882 (differential only), TestHalfBlock
// 94-149, FindBestDifferentialCombination 219-362, emitters 2414-2622, alpha 1902-2085,
// 2366-2411).  Bit-identical to the reference's SSE2 lanes in its canonical build (SURVEY App. C,
// hazard H2: the T-mode candidate slot just past a lane's unique colours reads as zero).
//
// Colour kernel mapping: one WAVE (= workgroup) = one block, one LANE = one candidate (a T-mode line colour, an H-mode colour pair, an ETC1
// half-block base colour, a planar coefficient combination).  Every candidate is evaluated over
// its pixels sequentially in registers (the reference's float sums are order dependent); the
// wave then reduces (error, candidate id) with ties going to the lowest id, which is the first
// candidate the reference's strict '<' would have kept.  Pixels, candidate lists and the
// differential attempt lists live in LDS.  The quantities that couple the eight blocks of a reference group (maxima of
// the T-mode unique-colour counts and line-pixel counts, the punch-through predicates) are recomputed by every wave
// from the group's 512 bytes of pixels, lane l standing in for group member l & 7: no workgroup barrier.
// Alpha kernel: integer-only and lane independent, one lane per block.
#include "cvtt_kernel_common.h"
#include <type_traits>

namespace
{
#define WAVE_SYNC()                                          \
    do                                                       \
    {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                     \
    } while (0)

constexpr int kMaxAttempts = 624; // 57 + 7 * 81 (reference ETC.h:38)

struct EtcWaveShared
{
    int pix[16][4];   // r, g, b, unused
    float pw[16][4];  // pre-weighted pixels (ExtractBlocks, ETC.cpp:2128-2155)
    float isoErr[16]; // T mode: error of the isolated colour per pixel
    unsigned short tColors[8][36];
    int tCount[8];
    union
    {
        // the group's 512 bytes of pixels (EncodeETC2 / EncodeETC1): read by pixJ / pwJ until the T modes are done, i.e. strictly
        // before the first store to `h` or `a` below (the hand-over point is marked in the kernel)
        u32 group[8][16];
        struct
        {
            float err[16][72];       // H mode: min error of colour ci +/- modifier per pixel, two tables at a time; [pixel][colour]: the
                                     // lanes of a pass hold consecutive colours, so every access is one run of consecutive words
            unsigned short color[2][36];
            unsigned short sign[72];
        } h;
        struct
        {
            // ETC1 differential attempts per half block: only the error is kept (5 KB); an attempt's
            // colour is looked up again in dColors and the winners' selectors are recomputed, which
            // keeps the workgroup at 72 KB of LDS = two workgroups per CU
            float err[2][kMaxAttempts];
        } a;
    } u;
    unsigned short dColors[16][82];      // de-duplicated base colours per (sector, table)
    int dCount[16];
    int planarRange[3][3][2];
};

// v_min_f32 of two values known not to be NaN (sums of squares read back from LDS): __builtin_fminf on loaded values makes
// the compiler canonicalise both operands first (two v_max_f32 per minimum)
__device__ __forceinline__ float minLoaded(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ int udivSmall(int n, int d)
{
    // exact n / d for 0 <= n < 2^16, 0 < d < 2^12 (integer divisions of ETC.cpp:446, 546, 717).  v_rcp_f32 (1 ulp) is enough:
    // the product is within 0.02 of n / d, so the truncated value is off by at most one, which the two corrections undo
    // (__frcp_rn is a full IEEE division sequence under -fhip-fp32-correctly-rounded-divide-sqrt: twenty instructions)
    int q = (int)((float)n * __builtin_amdgcn_rcpf((float)d));
    const int r = n - q * d;
    if (r < 0) q--;
    if (r >= d) q++;
    return q;
}

// ceil(2^20 / d) for 1 <= d <= 128 (one float division with the two possible corrections, per EAC candidate)
__device__ __forceinline__ int udivSmall20(int d)
{
    int q = (int)(1048576.0f * __builtin_amdgcn_rcpf((float)d));
    int r = 1048576 - q * d;
    if (r < 0) { q--; r += d; }
    if (r >= d) { q++; r -= d; }
    return r == 0 ? q : q + 1;
}

__device__ __forceinline__ u32 bswap32(u32 v) { return __builtin_bswap32(v); }

// A value that is the same in every lane (a wave-wide winner, a best-so-far) moved to a scalar register: the compiler cannot
// see the uniformity through shuffles and LDS reads and would keep one copy per lane alive across the search phases.
__device__ __forceinline__ u32 uniU(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int uniI(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uniF(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// the value lane `lane` holds, `lane` the same in every lane: v_readlane_b32 (a scalar result) instead of a ds_bpermute_b32
__device__ __forceinline__ int laneI(int v, int lane) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(lane)); }
__device__ __forceinline__ float laneF(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), __builtin_amdgcn_readfirstlane(lane))); }
// The value of lane (l ^ M), M a power of two: ds_swizzle_b32 in bit mode inside the 32-lane halves (no address register, a
// third of the LDS-crossbar time of ds_bpermute_b32), ds_bpermute_b32 across them.  __shfl_xor computes its lane index with the
// wave-size arithmetic of amd_warp_functions.h, whose common subexpression the compiler kept alive -- and spilled -- through the
// whole kernel.
template <int M>
__device__ __forceinline__ int xorLaneI(int v)
{
    if (M < 32)
        return __builtin_amdgcn_ds_swizzle(v, (M << 10) | 0x1f);
    return __builtin_amdgcn_ds_bpermute((int)((threadIdx.x ^ 32u) << 2), v);
}
template <int M>
__device__ __forceinline__ float xorLaneF(float v) { return __int_as_float(xorLaneI<M>(__float_as_int(v))); }

// wave-wide argmin of (err, id); ties -> lowest id.  All lanes receive the winner (as scalars: what is derived from it is scalar
// arithmetic).  The errors are sums of squares or FLT_MAX -- no NaN, no -0 -- so the lexicographic minimum of (err, id) is the
// minimum of the errors (six ds_swizzle + v_min_f32) followed by the lowest id among the lanes that hold it: nearly always ONE
// lane (a ballot, s_ff1, v_readlane), otherwise a second reduction over the ids.  (Round 5: the pair-at-a-time form -- two
// swizzles, two compares, two selects per step -- was a third of the pair walk's instructions.)
// An error that IS NaN (infinite weights: 0 * inf) would leave the lanes with different minima -- minLoaded keeps its second
// operand on an unordered compare -- and the id without a holder: such errors count as FLT_MAX, so the id returned is always
// one of the candidates (one v_cmp_u_f32 and a scalar branch in the common path).
__device__ __forceinline__ void waveArgmin(float &err, int &id)
{
    if (__ballot(err != err) != 0)
        err = (err == err) ? err : FLT_MAX;
    float m = err;
    m = minLoaded(m, xorLaneF<1>(m));
    m = minLoaded(m, xorLaneF<2>(m));
    m = minLoaded(m, xorLaneF<4>(m));
    m = minLoaded(m, xorLaneF<8>(m));
    m = minLoaded(m, xorLaneF<16>(m));
    m = minLoaded(m, xorLaneF<32>(m));
    const bool mine = err == m;
    const u64 holders = __ballot(mine);
    int best;
    if (__popcll(holders) == 1)
        best = __builtin_amdgcn_readlane(id, __ffsll((long long)holders) - 1);
    else
    {
        int c = mine ? id : 0x7fffffff;
#define ETC_IDMIN_STEP(M) { const int o = xorLaneI<M>(c); c = o < c ? o : c; }
        ETC_IDMIN_STEP(1) ETC_IDMIN_STEP(2) ETC_IDMIN_STEP(4) ETC_IDMIN_STEP(8) ETC_IDMIN_STEP(16) ETC_IDMIN_STEP(32)
#undef ETC_IDMIN_STEP
        best = __builtin_amdgcn_readfirstlane(c);
    }
    err = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m)));
    id = best;
}

// ConvertToFakeBT709, ETC.cpp:2343-2352
__device__ __forceinline__ void toFake709(float (&yuv)[3], float r, float g, float b)
{
    yuv[0] = r * 0.368233989135369f + g * 1.23876274963149f + b * 0.125054068802017f;
    yuv[1] = r * 0.5f - g * 0.4541529f - b * 0.04584709f;
    yuv[2] = r * -0.081014709086133f - g * 0.272538676238785f + b * 0.353553390593274f;
}

// the octant search shared by ResolveTHFakeBT709Rounding and ResolveHalfBlockFakeBT709RoundingAccurate
// (ETC.cpp:2199-2231, 2304-2326): which of the 8 cell corners is nearest to `target`; the reference's error adds the
// chroma-U difference twice instead of squaring it
__device__ __forceinline__ int fakeOctant(const float (&lowF)[3], const float (&highF)[3], const float (&target)[3])
{
    float cum[3];
    toFake709(cum, target[0], target[1], target[2]);
    float bestError = FLT_MAX;
    int bestOctant = 0;
#pragma unroll
    for (int octant = 0; octant < 8; octant++)
    {
        float oy[3];
        toFake709(oy, (octant & 1) ? highF[0] : lowF[0], (octant & 2) ? highF[1] : lowF[1], (octant & 4) ? highF[2] : lowF[2]);
        const float d0 = oy[0] - cum[0], d1 = oy[1] - cum[1], d2 = oy[2] - cum[2];
        const float error = d0 * d0 + d1 + d1 + d2 * d2;
        if (error < bestError)
            bestOctant = octant;
        bestError = sseMin(error, bestError);
    }
    return bestOctant;
}

// ResolveTHFakeBT709Rounding, ETC.cpp:2286-2327
__device__ __forceinline__ void resolveTHFake(int (&quantized)[3], const int (&targets)[3], int granularity)
{
    float lowF[3], highF[3], tf[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
    {
        const int unq = (quantized[ch] << 4) | quantized[ch];
        const int next = unq + 17 < 255 ? unq + 17 : 255;
        lowF[ch] = (float)(int)(short)((int)(short)(unq * granularity) << 1);
        highF[ch] = (float)(int)(short)((int)(short)(next * granularity) << 1);
        tf[ch] = (float)targets[ch];
    }
    const int octant = fakeOctant(lowF, highF, tf);
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
        quantized[ch] += (octant >> ch) & 1;
}

// ResolveHalfBlockFakeBT709RoundingAccurate / Fast, ETC.cpp:2157-2284
__device__ __forceinline__ void resolveHalfFake(int (&quantized)[3], const int (&cumulative)[3], bool isDifferential, bool accurate,
                                                const CvttDeviceTables *__restrict__ T)
{
    if (accurate)
    {
        float lowF[3], highF[3], tf[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
        {
            const u32 cu = (u32)cumulative[ch];
            int unq, next;
            if (isDifferential)
            {
                quantized[ch] = (int)(((((cu << 5) - cu) + (cu >> 3)) & 0xffffu) >> 11);
                unq = (quantized[ch] << 3) | (quantized[ch] >> 2);
                const int qn = quantized[ch] + 1 < 31 ? quantized[ch] + 1 : 31;
                next = (qn << 3) | (qn >> 2);
            }
            else
            {
                quantized[ch] = (int)(((((cu << 5) - ((cu << 1) & 0xffffu)) + (cu >> 3)) & 0xffffu) >> 12);
                unq = (quantized[ch] << 4) | quantized[ch];
                next = unq + 17 < 255 ? unq + 17 : 255;
            }
            lowF[ch] = (float)(unq << 3);
            highF[ch] = (float)(next << 3);
            tf[ch] = (float)cumulative[ch];
        }
        const int octant = fakeOctant(lowF, highF, tf);
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
            quantized[ch] += (octant >> ch) & 1;
        return;
    }
    int fill[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
        fill[ch] = cumulative[ch] + (cumulative[ch] >> 8);
    const int index = isDifferential ? (((fill[0] << 6) & 0xf00) | ((fill[1] << 4) & 0x0f0) | ((fill[2] >> 2) & 0x00f))
                                     : (((fill[0] << 5) & 0xf00) | ((fill[1] << 1) & 0x0f0) | ((fill[2] >> 3) & 0x00f));
    const int octant = T->fake709Rounding[index];
    const int upper = isDifferential ? 31 : 15, shift = isDifferential ? 6 : 7;
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
    {
        const int q = (fill[ch] >> shift) + ((octant >> ch) & 1);
        quantized[ch] = q < upper ? q : upper;
    }
}

// Two f32 values that go through the same operations: two plain instructions each (default), or, with CVTT_ETC_PACKED_F32, one
// v_pk_mul_f32 / v_pk_add_f32.  Same operations on the same values in the same order either way.  A packed instruction does two
// values in one 4.3-cycle issue slot but never shares the slot with another wave's instruction; a plain f32 instruction shares it
// with nearly anything (tools/valu_mix.py), and this kernel has more compares, minima and selects looking for such a partner than
// plain f32 work.  Round 2 measured packed ahead (3.7 waves per SIMD, 24 spilled registers); round 5, with 4.2 waves and no
// scratch: plain EncodeETC2RGBA 41.1 -> 42.6, EncodeETC2 44.1 -> 46.2, EncodeETC1 42.2 -> 44.0 Mblocks/s (profiles/r05/ab_etc2.txt).
#ifndef CVTT_ETC_PACKED_F32
struct EtcPair
{
    float x, y;
};
__device__ __forceinline__ EtcPair operator-(const EtcPair &a, const EtcPair &b) { return EtcPair{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ EtcPair operator+(const EtcPair &a, const EtcPair &b) { return EtcPair{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ EtcPair operator*(const EtcPair &a, const EtcPair &b) { return EtcPair{a.x * b.x, a.y * b.y}; }
#else
typedef float EtcPair __attribute__((ext_vector_type(2)));
#endif

struct EtcErr
{
    bool uniform;
    float rw, gw, bw;
    bool fake;
    // the metric of most call sites: ComputeErrorFakeBT709 (ETC.cpp:82-92) first, then uniform, then weighted
    __device__ __forceinline__ float operator()(int r, int g, int b, const int *px, const float *pw) const
    {
        if (fake)
        {
            float yuv[3];
            toFake709(yuv, (float)r, (float)g, (float)b);
            const float dy = yuv[0] - pw[0], du = yuv[1] - pw[1], dv = yuv[2] - pw[2];
            return dy * dy + du * du + dv * dv;
        }
        return wu(r, g, b, px, pw);
    }
    // The weighted metric for TWO colours at once (an EtcPair: plain or packed f32, see above): each of the two does
    // ComputeErrorWeighted's operations in its order (product, difference, squares added left to right), so both results are
    // the numbers the scalar form gives.  Only valid when !uniform (and, where the caller would use operator(), !fake).
    typedef EtcPair f32x2;
    __device__ __forceinline__ void weigh2(f32x2 (&mw)[3], const int (&a)[3], const int (&b)[3]) const
    {
        mw[0] = f32x2{(float)a[0] * rw, (float)b[0] * rw};
        mw[1] = f32x2{(float)a[1] * gw, (float)b[1] * gw};
        mw[2] = f32x2{(float)a[2] * bw, (float)b[2] * bw};
    }
    __device__ __forceinline__ f32x2 err2(const f32x2 (&mw)[3], const float *pw) const
    {
        f32x2 d = mw[0] - f32x2{pw[0], pw[0]};
        f32x2 e = d * d;
        d = mw[1] - f32x2{pw[1], pw[1]};
        e = e + d * d;
        d = mw[2] - f32x2{pw[2], pw[2]};
        e = e + d * d;
        return e;
    }
    // ComputeErrorUniform / ComputeErrorWeighted, ETC.cpp:59-80.  The line colours of the T modes are measured with
    // this even under ETC_UseFakeBT709 (ETC.cpp:600, 1133, 1150), against pre-weighted pixels that then hold luma/chroma
    __device__ __forceinline__ float wu(int r, int g, int b, const int *px, const float *pw) const
    {
        if (uniform)
        {
            const float d0 = (float)(r - px[0]), d1 = (float)(g - px[1]), d2 = (float)(b - px[2]);
            float e = d0 * d0;
            e = e + d1 * d1;
            e = e + d2 * d2;
            return e;
        }
        const float dr = (float)r * rw - pw[0];
        const float dg = (float)g * gw - pw[1];
        const float db = (float)b * bw - pw[2];
        return dr * dr + dg * dg + db * db;
    }
};

__device__ __forceinline__ int planarDecode(int coeff, int ch)
{
    return (ch == 1) ? (((coeff << 1) | (coeff >> 6)) & 0xffff) : (((coeff << 2) | (coeff >> 4)) & 0xffff);
}

// EmitTModeBlock, ETC.cpp:2414-2460
__device__ __forceinline__ void emitT(u32 &hi, u32 &lo, const int (&lineColor)[3], const int (&iso)[3], u32 packedSelectors, int table, bool opaque = true)
{
    hi = 0;
    lo = 0;
    const int rh = (iso[0] >> 2) & 3, rl = iso[0] & 3;
    if (rh + rl < 4) hi |= 1u << (58 - 32); else hi |= 7u << (61 - 32);
    hi |= (u32)rh << (59 - 32);
    hi |= (u32)rl << (56 - 32);
    hi |= (u32)iso[1] << (52 - 32);
    hi |= (u32)iso[2] << (48 - 32);
    hi |= (u32)lineColor[0] << (44 - 32);
    hi |= (u32)lineColor[1] << (40 - 32);
    hi |= (u32)lineColor[2] << (36 - 32);
    hi |= (u32)((table >> 1) & 3) << (34 - 32);
    if (opaque)
        hi |= 1u << (33 - 32);
    hi |= (u32)(table & 1);
#pragma unroll
    for (int px = 0; px < 16; px++)
    {
        const int src = ((px & 3) << 2) | (px >> 2); // selectorOrder
        const u32 sel = (packedSelectors >> (2 * src)) & 3u;
        lo |= (sel & 1u) << px;
        lo |= ((sel >> 1) & 1u) << (16 + px);
    }
}

// EmitHModeBlock, ETC.cpp:2462-2563.  bc0 / bc1: 4:4:4 colours packed r << 10 | g << 5 | b.
__device__ __forceinline__ void emitH(u32 &outHi, u32 &outLo, int bc0, int bc1, u32 sectorBits, u32 signBits, int table, bool opaque)
{
    if (bc0 == bc1)
    {
        const int lineColor[3] = {(bc0 >> 10) & 0x1f, (bc0 >> 5) & 0x1f, bc0 & 0x1f};
        u32 packedSelectors = 0x55555555u;
#pragma unroll
        for (int px = 0; px < 16; px++)
            packedSelectors |= ((signBits >> px) & 1u) << ((px * 2) + 1);
        emitT(outHi, outLo, lineColor, lineColor, packedSelectors, table, opaque);
        return;
    }
    int colors[2][3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
    {
        colors[0][ch] = (bc0 >> ((2 - ch) * 5)) & 15;
        colors[1][ch] = (bc1 >> ((2 - ch) * 5)) & 15;
    }
    if (((table & 1) == 1) != (bc0 > bc1))
    {
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
        {
            const int t = colors[0][ch];
            colors[0][ch] = colors[1][ch];
            colors[1][ch] = t;
        }
        sectorBits ^= 0xffffu;
    }
    const int r1 = colors[0][0], g1a = colors[0][1] >> 1, g1b = colors[0][1] & 1, b1a = colors[0][2] >> 3, b1b = colors[0][2] & 7;
    const int r2 = colors[1][0], g2 = colors[1][1], b2 = colors[1][2];
    u32 hi = 0, lo = 0;
    if ((g1a & 4) != 0 && r1 + g1a < 8) hi |= 1u << (63 - 32);
    const int fakeDG = b1b >> 1, fakeG = b1a | (g1b << 1);
    if (fakeG + fakeDG < 4) hi |= 1u << (50 - 32); else hi |= 7u << (53 - 32);
    hi |= (u32)r1 << (59 - 32);
    hi |= (u32)g1a << (56 - 32);
    hi |= (u32)g1b << (52 - 32);
    hi |= (u32)b1a << (51 - 32);
    hi |= (u32)b1b << (47 - 32);
    hi |= (u32)r2 << (43 - 32);
    hi |= (u32)g2 << (39 - 32);
    hi |= (u32)b2 << (35 - 32);
    hi |= (u32)((table >> 2) & 1) << (34 - 32);
    if (opaque)
        hi |= 1u << (33 - 32);
    hi |= (u32)((table >> 1) & 1);
#pragma unroll
    for (int px = 0; px < 16; px++)
    {
        const int src2 = ((px & 3) << 2) | (px >> 2);
        lo |= ((signBits >> src2) & 1u) << px;
        lo |= ((sectorBits >> src2) & 1u) << (16 + px);
    }
    outHi = hi;
    outLo = lo;
}
} // namespace

#ifdef CVTT_ETC_PROFILE
// developer-only: wave cycles per stage (0 planar, 1/2 T mode calls, 3 H mode, 4 cluster fit), summed over waves
__device__ unsigned long long g_etcProf[16];
extern "C" int cvttmi_etc_prof_read(unsigned long long *out)
{
    unsigned long long zero[16] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_etcProf), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_etcProf), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif
// ------------------------------------------------------------------------------------------
// MODE 0: EncodeETC2 (RGB), 1: EncodeETC1, 2: EncodeETC2PunchthroughAlpha; FAKE: ETC_UseFakeBT709
#ifndef CVTT_ETC2_WAVES
#define CVTT_ETC2_WAVES 5
#endif
struct EtcGroupPixels { u32 w[8][16]; };
struct EtcNoGroupPixels { u32 unused; };
template <int MODE, bool FAKE>
// (the punch-through instantiation keeps its group pixels apart -- 9.4 KB, 16 workgroups per CU -- so a tighter register budget
// would only make it spill)
__global__ __launch_bounds__(64, MODE == 2 ? 4 : CVTT_ETC2_WAVES) void cvttmi_etc2_color_kernel(const uint8_t *__restrict__ blocks, uint8_t *__restrict__ out,
                                                              const CvttEtcArgs A, const CvttDeviceTables *__restrict__ T)
{
    // One wave = one workgroup = one block.  What a block needs from the other seven of its reference group (the maxima of
    // the unique-colour counts and of the line-pixel counts, the two punch-through predicates) it computes itself: lane l
    // works for group member l & 7 from the 512 bytes of the group's pixels.  No workgroup barrier, no waiting for the
    // slowest block of the group, and 9 KB of LDS per wave instead of 72 KB per eight.
    __shared__ EtcWaveShared shared1;
    // The group's pixels (512 B).  EncodeETC2 / EncodeETC1 need them for the sector split and the T modes only, which are over
    // before the H mode and the cluster fit use the union `u`: there they live in it (8 940 B of LDS = 7 allocation granules =
    // 18 workgroups per CU instead of 16).  The punch-through modes look at the group again after the cluster fit.
    __shared__ typename std::conditional<MODE == 2, EtcGroupPixels, EtcNoGroupPixels>::type groupSeparate;
    u32 (*const gpix)[16] = (MODE == 2) ? reinterpret_cast<u32 (*)[16]>(&groupSeparate) : shared1.u.group;

    // (not const: ETC_REFRESH_LANE() hands the optimiser the same number as a NEW value at the start of the later stages, so that
    // what is derived from it -- lane & 7, lane & 15, lane >> 3 ... -- is computed where it is used instead of being kept alive,
    // and spilled, from the sector split to the last stage)
    int lane = threadIdx.x;
#define ETC_REFRESH_LANE() do { lane = (int)threadIdx.x; asm volatile("" : "+v"(lane)); } while (0)
    // The eight waves of a reference group read the same 512 bytes.  Workgroup b runs on XCD b % 8 (observed, for speed only;
    // nothing depends on it), each XCD with an L2 of its own: with block = workgroup number the group's pixels were fetched from
    // HBM eight times (851 MB per 4096^2 image against 75 MB of algorithmic bytes).  So XCD x takes the x-th eighth of the
    // groups, and the waves of a group are the workgroups b, b + 8, ... b + 56 of one XCD, dispatched within a few microseconds
    // of each other.
    const u32 xcdChunk = ((A.numBlocks / 8u + 7u) / 8u) * 8u; // blocks per XCD: whole groups
    const u32 blockIndex = (blockIdx.x & 7u) * xcdChunk + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= xcdChunk || blockIndex >= A.numBlocks)
        return;
    const int own = (int)(blockIndex & 7u), jb = lane & 7;
    EtcWaveShared &S = shared1;
    const EtcErr E = {(A.flags & CVTTMI_FLAG_UNIFORM) != 0, A.rw, A.gw, A.bw, FAKE};
    const bool fakeAccurate = (A.flags & CVTTMI_FLAG_ETC_FAKE_BT709_ACCURATE) != 0;
    constexpr bool ETC1 = MODE == 1, PUNCH = MODE == 2;
    // The eight tables' small and large ETC1 modifiers and T / H distances as bytes of wave-uniform words: a candidate's table
    // is a per-lane value, and looking it up in memory was up to four loads (and their latency) at the head of every pass of
    // 64 candidates; one v_perm_b32 picks the byte instead.
    u32 mSLo = 0, mSHi = 0, mLLo = 0, mLHi = 0, thLo = 0, thHi = 0;
#pragma unroll
    for (int t = 0; t < 4; t++)
    {
        mSLo |= (u32)T->etc1Modifiers[t][2] << (8 * t);
        mSHi |= (u32)T->etc1Modifiers[4 + t][2] << (8 * t);
        mLLo |= (u32)T->etc1Modifiers[t][3] << (8 * t);
        mLHi |= (u32)T->etc1Modifiers[4 + t][3] << (8 * t);
        thLo |= (u32)T->thDistance[t] << (8 * t);
        thHi |= (u32)T->thDistance[4 + t] << (8 * t);
    }
    mSLo = (u32)__builtin_amdgcn_readfirstlane((int)mSLo); mSHi = (u32)__builtin_amdgcn_readfirstlane((int)mSHi);
    mLLo = (u32)__builtin_amdgcn_readfirstlane((int)mLLo); mLHi = (u32)__builtin_amdgcn_readfirstlane((int)mLHi);
    thLo = (u32)__builtin_amdgcn_readfirstlane((int)thLo); thHi = (u32)__builtin_amdgcn_readfirstlane((int)thHi);
    auto smallMod = [&](int table) -> int { return (int)__builtin_amdgcn_perm(mSHi, mSLo, (u32)table | 0x0c0c0c00u); };
    auto largeMod = [&](int table) -> int { return (int)__builtin_amdgcn_perm(mLHi, mLLo, (u32)table | 0x0c0c0c00u); };
    auto thDist = [&](int table) -> int { return (int)__builtin_amdgcn_perm(thHi, thLo, (u32)table | 0x0c0c0c00u); };

    // ---- load: pixel px by lane px.  Punch-through: pixels whose alpha is below the threshold are transparent and
    // count as black from here on (ETC.cpp:1670-1720) ----
    bool pxTransparent = false;
    u32 alphaLane = 0; // EncodeETC2RGBA: the alpha of pixel `lane`
    if (lane < 16)
    {
        const u32 pk = reinterpret_cast<const u32 *>(blocks + (size_t)blockIndex * 64u)[lane];
        alphaLane = pk >> 24;
        int r = (int)(pk & 0xffu), g = (int)((pk >> 8) & 0xffu), b = (int)((pk >> 16) & 0xffu);
        pxTransparent = PUNCH && (pk >> 24) < A.alphaThreshold;
        if (pxTransparent)
            r = g = b = 0;
        S.pix[lane][0] = r;
        S.pix[lane][1] = g;
        S.pix[lane][2] = b;
        float pw3[3] = {E.uniform ? (float)r : (float)r * A.rw, E.uniform ? (float)g : (float)g * A.gw, E.uniform ? (float)b : (float)b * A.bw};
        if (FAKE) // ExtractBlocks, ETC.cpp:2141-2142 (takes precedence over Uniform)
            toFake709(pw3, (float)r, (float)g, (float)b);
        S.pw[lane][0] = pxTransparent ? 0.0f : pw3[0];
        S.pw[lane][1] = pxTransparent ? 0.0f : pw3[1];
        S.pw[lane][2] = pxTransparent ? 0.0f : pw3[2];
    }
    const u32 transMask = PUNCH ? ((u32)__ballot(pxTransparent) & 0xffffu) : 0u;

    // ---- EncodeETC2RGBA (outStride 16): the EAC alpha half of the block, here, from the pixels this wave has just loaded ----
    // CompressETC2AlphaBlockInternal, ETC.cpp:1902-2085 -- the search of cvttmi_eac_alpha_kernel<0> below with the 320
    // (table, range, multiplier) candidates dealt to the lanes: candidate c = table * 20 + r * 2 + mo is the reference's loop
    // order, lane l takes c = l, l + 64, ... in ascending order with the reference's strict '<', and the wave-wide minimum of
    // (error << 9 | c) is the first minimum of that order.  The sixteen alpha values are wave-uniform (scalar registers), so a
    // candidate costs what it costs in the one-lane-per-block kernel, the same instruction count per block without a second
    // pass over the PixelBlocks -- and the block goes out as ONE 16-byte store (round 4: two kernels, both reading the 64 MB of
    // a 4096^2 image, 8-byte stores into 16-byte slots: 2.36 x the algorithmic HBM traffic).
    u32 alphaW0 = 0, alphaW1 = 0; // wave-uniform
    if (MODE == 0 && A.outStride == 16u)
    {
        int al[16];
#pragma unroll
        for (int px = 0; px < 16; px++)
            al[px] = __builtin_amdgcn_readlane((int)alphaLane, px);
        int minAlpha = 255, maxAlpha = 0;
#pragma unroll
        for (int px = 0; px < 16; px++)
        {
            minAlpha = al[px] < minAlpha ? al[px] : minAlpha;
            maxAlpha = al[px] > maxAlpha ? al[px] : maxAlpha;
        }
        const int alphaSpan = maxAlpha - minAlpha, midTimes2 = maxAlpha + minAlpha;
        u32 bestKey = 0xffffffffu;
        int bestBase = 0, bestMultiplier = 0;
#pragma unroll 1
        for (int c = lane; c < 320; c += 64)
        {
            const int tableIndex = (int)(((u32)c * 3277u) >> 16); // c / 20 for c < 320
            const int rem = c - tableIndex * 20;
            const int r = rem >> 1, mo = rem & 1;
            const u32 posWord = T->eacPosWord[tableIndex], roundBits = T->eacRoundBits[tableIndex];
            const int mainRange = (int)(((u32)r * 11u) >> 5); // r / 3 for r < 10
            const int subrange = r - mainRange * 3;
            const int maxOffset = (int)((posWord >> (8 * (3 - mainRange - (subrange & 1)))) & 0xffu);
            const int minOffset = -(int)((posWord >> (8 * (3 - mainRange - ((subrange >> 1) & 1)))) & 0xffu) - 1;
            int minMultiplier = udivSmall(alphaSpan, maxOffset - minOffset);
            minMultiplier = minMultiplier > 14 ? 14 : minMultiplier;
            minMultiplier = minMultiplier < 1 ? 1 : minMultiplier;
            const int multiplier = minMultiplier + mo;
            int base2 = midTimes2 - multiplier * maxOffset - multiplier * minOffset;
            base2 = base2 < 0 ? 0 : (base2 > 510 ? 510 : base2);
            const int baseAlpha = (base2 + 1) >> 1;
            const u32 magic = (u32)udivSmall20(multiplier);
            u32 totalError = 0;
#pragma unroll
            for (int px = 0; px < 16; px++)
            {
                const int a = al[px];
                const int refl2 = (a - baseAlpha) * 2 + multiplier;
                const int absv = refl2 < 0 ? -refl2 : refl2;
                int li = (int)(__umul24((u32)(absv >> 1), magic) >> 20);
                li = li >= 13 ? 12 : li;
                const int index = (int)((roundBits >> (2 * li)) & 3u);
                const int pOff = (int)((posWord >> (8 * index)) & 0xffu);
                const int sign = refl2 < 0 ? -1 : 0;
                int q = baseAlpha + (pOff ^ sign) * multiplier;
                q = q < 0 ? 0 : (q > 255 ? 255 : q);
                const int dq = q - a;
                totalError += (u32)(dq * dq);
            }
            const u32 key = (totalError << 9) | (u32)c; // 16 x 255^2 < 2^20
            if (key < bestKey)
            {
                bestKey = key;
                bestBase = baseAlpha;
                bestMultiplier = multiplier;
            }
        }
        u32 key = bestKey;
#define ETC_KEYMIN_STEP(M) { const u32 o = (u32)xorLaneI<M>((int)key); key = o < key ? o : key; }
        ETC_KEYMIN_STEP(1) ETC_KEYMIN_STEP(2) ETC_KEYMIN_STEP(4) ETC_KEYMIN_STEP(8) ETC_KEYMIN_STEP(16) ETC_KEYMIN_STEP(32)
#undef ETC_KEYMIN_STEP
        const int bestC = (int)(key & 511u);
        const int bestTable = (int)(((u32)bestC * 3277u) >> 16);
        bestBase = laneI(bestBase, bestC & 63);
        bestMultiplier = laneI(bestMultiplier, bestC & 63);
        // the winner's indexes: lane px < 16 its pixel's code, at the place the column-major emission order gives it
        // (16 x 3 bits, MSB first, ETC.cpp:2049-2084: the s-th code emitted is the pixel ((s & 3) << 2) | (s >> 2))
        u32 bitsLo = 0, bitsHi = 0; // bits 0..31 / 32..47 of the 48-bit field
        {
            const u32 magic = (u32)udivSmall20(bestMultiplier);
            const int refl2 = ((int)alphaLane - bestBase) * 2 + bestMultiplier;
            const int absv = refl2 < 0 ? -refl2 : refl2;
            int li = (int)(__umul24((u32)(absv >> 1), magic) >> 20);
            li = li >= 13 ? 12 : li;
            const int index = (int)((T->eacRoundBits[bestTable] >> (2 * li)) & 3u);
            const u64 code = (u64)(u32)(index + 4 - ((refl2 < 0 ? -1 : 0) & 4));
            const int s = ((lane & 3) << 2) | ((lane >> 2) & 3);
            const u64 placed = lane < 16 ? code << (3 * (15 - s)) : 0ull;
            bitsLo = (u32)placed;
            bitsHi = (u32)(placed >> 32);
#define ETC_OR_STEP(M) { bitsLo |= (u32)xorLaneI<M>((int)bitsLo); bitsHi |= (u32)xorLaneI<M>((int)bitsHi); }
            ETC_OR_STEP(1) ETC_OR_STEP(2) ETC_OR_STEP(4) ETC_OR_STEP(8)
#undef ETC_OR_STEP
        }
        const u32 w0 = ((u32)bestBase & 0xffu) | ((u32)(((bestMultiplier << 4) | bestTable) & 0xff) << 8) | (((bitsHi >> 8) & 0xffu) << 16) | ((bitsHi & 0xffu) << 24);
        alphaW0 = (u32)__builtin_amdgcn_readfirstlane((int)w0);
        alphaW1 = (u32)__builtin_amdgcn_readfirstlane((int)bswap32(bitsLo));
    }
    // the group's pixels: 128 words, two per lane
    {
        const u32 *gsrc = reinterpret_cast<const u32 *>(blocks + (size_t)(blockIndex & ~7u) * 64u);
        (&gpix[0][0])[lane] = gsrc[lane];
        (&gpix[0][0])[lane + 64] = gsrc[lane + 64];
    }
    WAVE_SYNC();
    // member jb of the group as this lane sees it: transparency mask, and pixel accessors that apply ExtractBlocks and
    // the punch-through blackening exactly as the block's own wave does
    u32 transJ = 0;
    if (PUNCH)
    {
#pragma unroll
        for (int px = 0; px < 16; px++)
            transJ |= ((gpix[jb][px] >> 24) < A.alphaThreshold ? 1u : 0u) << px;
    }
    const int numOpaqueJ = 16 - __popc(transJ);
    auto pixJ = [&](int px, int (&c)[3]) {
        const u32 pk = gpix[jb][px];
        const bool tr = ((transJ >> px) & 1u) != 0;
        c[0] = tr ? 0 : (int)(pk & 0xffu);
        c[1] = tr ? 0 : (int)((pk >> 8) & 0xffu);
        c[2] = tr ? 0 : (int)((pk >> 16) & 0xffu);
    };
    auto pwJ = [&](int px, float (&w3)[3]) {
        int c[3];
        pixJ(px, c);
        w3[0] = E.uniform ? (float)c[0] : (float)c[0] * A.rw;
        w3[1] = E.uniform ? (float)c[1] : (float)c[1] * A.gw;
        w3[2] = E.uniform ? (float)c[2] : (float)c[2] * A.bw;
        if (FAKE)
            toFake709(w3, (float)c[0], (float)c[1], (float)c[2]);
        if ((transJ >> px) & 1u)
            w3[0] = w3[1] = w3[2] = 0.0f;
    };
    // the two group-wide predicates of CompressETC2Block: does ANY block have a transparent pixel (then every block
    // also goes through the punch-through modes), are ALL blocks fully transparent (then the opaque modes are skipped)
    const bool groupAny = PUNCH && (__ballot(transJ != 0) & 0xffull) != 0;
    const bool groupAll = PUNCH && (__ballot(transJ == 0xffffu) & 0xffull) == 0xffull;
    // maximum over the eight members (lanes that differ in their low three bits)
    auto groupMax = [](int v) {
        int o = xorLaneI<1>(v);
        v = o > v ? o : v;
        o = xorLaneI<2>(v);
        v = o > v ? o : v;
        o = xorLaneI<4>(v);
        v = o > v ? o : v;
        return v;
    };

    float bestError = FLT_MAX;
    u32 outHi = 0, outLo = 0;
#ifdef CVTT_ETC_PROFILE
    unsigned long long profT = __builtin_readcyclecounter();
#define DBG_TAP(i) do { const unsigned long long now = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&g_etcProf[i], now - profT); profT = now; } while (0)
#define DBG_COUNT(i, n) do { if (lane == 0) atomicAdd(&g_etcProf[i], (unsigned long long)(n)); } while (0)
#elif defined(CVTT_ETC_DEBUG)
    float *dbg = reinterpret_cast<float *>(A.debug) + (size_t)blockIndex * 8;
#define DBG_TAP(i) do { if (lane == 0 && A.debug) dbg[i] = bestError; } while (0)
#define DBG_COUNT(i, n) do {} while (0)
#else
#define DBG_TAP(i) do {} while (0)
#define DBG_COUNT(i, n) do {} while (0)
#endif

    u32 isolatedMask = 0; // bit px: pixel is "isolated" / sector 1
    u32 isoJ = 0;         // the same for group member jb
    // line-pixel totals of member jb under a sector assignment (the T modes' candidate lists come from them)
    auto lineTotalsJ = [&](u32 lineMask, int (&tot)[3]) {
        tot[0] = tot[1] = tot[2] = 0;
#pragma unroll
        for (int px = 0; px < 16; px++)
            if ((lineMask >> px) & 1u)
            {
                int c[3];
                pixJ(px, c);
                tot[0] += c[0];
                tot[1] += c[1];
                tot[2] += c[2];
            }
    };

    if (!ETC1) // EncodeETC1 is the cluster fit alone (CompressETC1Block, ETC.cpp:2116-2126)
    {
    // =================================== planar ===================================
    if (!groupAll)
    {
        // closed-form least squares per channel (lanes 0..2), ETC.cpp:1291-1413
        const int ch = lane < 3 ? lane : 0;
        float fhh = 0.f, fho = 0.f, fhv = 0.f, foo = 0.f, fov = 0.f, fvv = 0.f;
        float fh = 0.0f, fv = 0.0f, fo = 0.0f;
        for (int px = 0; px < 16; px++)
        {
            const float x = (float)(px & 3), y = (float)(px >> 2);
            const float c = FAKE ? S.pw[px][ch] : (float)S.pix[px][ch];
            fhh += x * x; fhv += x * y; fho += x; fh = fh - c * x;
            fhv += y * x; fvv += y * y; fov += y; fv = fv - c * y;
            fho += x; fov += y; foo += 1.0f; fo = fo - c;
            fh = fh - c * x; fv = fv - c * y; fo = fo - c;
        }
        const float d = 2.0f * fhh, e = fho, f = fhv, gD = fh;
        const float i = fhv, j = fov, k = 2.0f * fvv, lD = fv;
        const float m = fho, n = 2.0f * foo, p = fov, qD = fo;
        const float r0to1 = -i / d, r0to2 = -m / d;
        const float j1 = j + r0to1 * e, k1 = k + r0to1 * f, l1D = lD + gD * r0to1;
        const float n1 = n + r0to2 * e, p1 = p + r0to2 * f, q1D = qD + gD * r0to2;
        const float r1to2 = -p1 / k1;
        const float n2 = n1 + r1to2 * j1, q2D = q1D + l1D * r1to2;
        const float oc = -q2D / n2;
        const float r2to1 = -j1 / n2;
        const float l2D = l1D + q2D * r2to1;
        const float elim2 = -f / k1, elim1 = -e / n2;
        const float g2D = gD + l2D * elim2 + q2D * elim1;
        float hc = -g2D / d, vc = -l2D / k1;
        hc = hc * 4.0f + oc;
        vc = vc * 4.0f + oc;
        float totalError = 0.0f;
        int coeffs[3][3];
        if (FAKE)
        {
            // ETC.cpp:1415-1474: the plane was fitted in luma/chroma; back to RGB, round to nearest, one candidate
            float o3[3], h3[3], v3[3];
#pragma unroll
            for (int c3 = 0; c3 < 3; c3++)
            {
                o3[c3] = laneF(oc, c3);
                h3[c3] = laneF(hc, c3);
                v3[c3] = laneF(vc, c3);
            }
            auto fromFake = [](float (&rgb)[3], const float (&yuv)[3]) { // ConvertFromFakeBT709, ETC.cpp:2354-2364
                const float yy = yuv[0] * 0.57735026466774571071f;
                rgb[0] = yy + yuv[1] * 1.5748000207960953486f;
                rgb[1] = yy - yuv[1] * 0.46812425854364753669f - yuv[2] * 0.26491652528157560861f;
                rgb[2] = yy + yuv[2] * 2.6242146882856944069f;
            };
            float oRGB[3], hRGB[3], vRGB[3];
            fromFake(oRGB, o3);
            fromFake(hRGB, h3);
            fromFake(vRGB, v3);
            int dO[3], hMinusO[3], vMinusO[3];
#pragma unroll
            for (int c3 = 0; c3 < 3; c3++)
            {
                const float fcoeffs[3] = {oRGB[c3], hRGB[c3], vRGB[c3]};
#pragma unroll
                for (int c = 0; c < 3; c++)
                {
                    float coeff = sseMax(0.0f, fcoeffs[c]);
                    coeff = (c3 == 1) ? sseMin(127.0f, coeff * (127.0f / 255.0f)) : sseMin(63.0f, coeff * (63.0f / 255.0f));
                    coeffs[c3][c] = (int)rintf(coeff);
                }
                dO[c3] = planarDecode(coeffs[c3][0], c3);
                hMinusO[c3] = (int)(short)(planarDecode(coeffs[c3][1], c3) - dO[c3]);
                vMinusO[c3] = (int)(short)(planarDecode(coeffs[c3][2], c3) - dO[c3]);
            }
            for (int px = 0; px < 16; px++)
            {
                int rec[3];
#pragma unroll
                for (int c3 = 0; c3 < 3; c3++)
                {
                    const int dec = (int)(short)((px & 3) * hMinusO[c3] + (px >> 2) * vMinusO[c3] + (int)(short)((dO[c3] << 2) + 2)) >> 2;
                    rec[c3] = dec < 0 ? 0 : (dec > 255 ? 255 : dec);
                }
                totalError = totalError + E(rec[0], rec[1], rec[2], S.pix[px], S.pw[px]);
            }
        }
        else
        {
        if (lane < 3)
        {
            const float fcoeffs[3] = {oc, hc, vc};
#pragma unroll
            for (int c = 0; c < 3; c++)
            {
                float coeff = sseMax(0.0f, fcoeffs[c]);
                coeff = (ch == 1) ? sseMin(127.0f, coeff * (127.0f / 255.0f)) : sseMin(63.0f, coeff * (63.0f / 255.0f));
                S.planarRange[ch][c][0] = (int)floorf(coeff); // RoundDownForScope
                S.planarRange[ch][c][1] = (int)ceilf(coeff);  // RoundUpForScope
            }
        }
        WAVE_SYNC();

        // the 8 floor/ceil combinations per channel: lanes 0..23 (ETC.cpp:1507-1552)
        const int pch = (lane >> 3) < 3 ? (lane >> 3) : 0;
        const int combo = lane & 7;
        const int cO = S.planarRange[pch][0][(combo >> 2) & 1];
        const int cH = S.planarRange[pch][1][(combo >> 1) & 1];
        const int cV = S.planarRange[pch][2][combo & 1];
        float error = 0.0f;
        {
            const int dO = planarDecode(cO, pch), dH = planarDecode(cH, pch), dV = planarDecode(cV, pch);
            const int hMinusO = (int)(short)(dH - dO), vMinusO = (int)(short)(dV - dO);
            const int addend = (int)(short)((dO << 2) + 2);
            for (int px = 0; px < 16; px++)
            {
                const int x = px & 3, y = px >> 2;
                int dec = (int)(short)(x * hMinusO + y * vMinusO + addend) >> 2;
                dec = dec < 0 ? 0 : (dec > 255 ? 255 : dec);
                const float deltaF = (float)(S.pix[px][pch] - dec);
                error = error + deltaF * deltaF;
            }
        }
        // argmin over the 8 lanes of a channel, ties -> lowest combination
        float cErr = lane < 24 ? error : FLT_MAX;
        int cId = combo;
#define ETC_CH_STEP(M) { const float oe = xorLaneF<M>(cErr); const int oi = xorLaneI<M>(cId); const bool take = (oe < cErr) || (oe == cErr && oi < cId); cErr = take ? oe : cErr; cId = take ? oi : cId; }
        ETC_CH_STEP(1) ETC_CH_STEP(2) ETC_CH_STEP(4)
#undef ETC_CH_STEP
        float chErr[3];
#pragma unroll
        for (int c3 = 0; c3 < 3; c3++)
        {
            const float eBest = laneF(cErr, c3 * 8);
            const int win = laneI(cId, c3 * 8);
            // bestChannelError starts at FLT_MAX and is replaced on a strict '<'
            chErr[c3] = eBest < FLT_MAX ? eBest : FLT_MAX;
            coeffs[c3][0] = S.planarRange[c3][0][(win >> 2) & 1];
            coeffs[c3][1] = S.planarRange[c3][1][(win >> 1) & 1];
            coeffs[c3][2] = S.planarRange[c3][2][win & 1];
            if (!E.uniform)
            {
                const float w = c3 == 0 ? A.rw : (c3 == 1 ? A.gw : A.bw);
                chErr[c3] = chErr[c3] * (w * w);
            }
        }
        totalError = totalError + chErr[0];
        totalError = totalError + chErr[1];
        totalError = totalError + chErr[2];
        }
        if (totalError < bestError)
        {
            bestError = totalError;
            const int ro = coeffs[0][0], rh = coeffs[0][1], rv = coeffs[0][2];
            const int go = coeffs[1][0], gh = coeffs[1][1], gv = coeffs[1][2];
            const int bo = coeffs[2][0], bh = coeffs[2][1], bv = coeffs[2][2];
            const int go1 = go >> 6, go2 = go & 63;
            const int bo1 = bo >> 5, bo2 = (bo >> 3) & 3, bo3 = bo & 7;
            const int rh1 = rh >> 1, rh2 = rh & 1;
            const int fakeR = ro >> 2, fakeDR = go1 | ((ro & 3) << 1);
            const int fakeG = go2 >> 2, fakeDG = ((go2 & 3) << 1) | bo1;
            const int fakeB = bo2, fakeDB = bo3 >> 1;
            u32 hi = 0, lo = 0;
            if ((fakeDR & 4) != 0 && fakeR + fakeDR < 8) hi |= 1u << (63 - 32);
            if ((fakeDG & 4) != 0 && fakeG + fakeDG < 8) hi |= 1u << (55 - 32);
            if (fakeB + fakeDB < 4) hi |= 1u << (42 - 32); else hi |= 7u << (45 - 32);
            hi |= (u32)ro << (57 - 32);
            hi |= (u32)go1 << (56 - 32);
            hi |= (u32)go2 << (49 - 32);
            hi |= (u32)bo1 << (48 - 32);
            hi |= (u32)bo2 << (43 - 32);
            hi |= (u32)bo3 << (39 - 32);
            hi |= (u32)rh1 << (34 - 32);
            hi |= 1u << (33 - 32);
            hi |= (u32)rh2;
            lo |= (u32)gh << 25;
            lo |= (u32)bh << 19;
            lo |= (u32)rv << 13;
            lo |= (u32)gv << 6;
            lo |= (u32)bv;
            outHi = hi;
            outLo = lo;
        }
    }

    DBG_TAP(0);
    bestError = uniF(bestError); outHi = uniU(outHi); outLo = uniU(outLo);
    // ============ sector split along the chroma principal axis (ETC.cpp:1723-1848) ============
    // every lane does it for group member jb; the block's own split is lane `own`'s
    {
        float cdx[16], cdy[16];
        if (E.uniform)
        {
            int cenX = 0, cenY = 0;
            int ccx[16], ccy[16];
#pragma unroll
            for (int px = 0; px < 16; px++)
            {
                int c[3];
                pixJ(px, c);
                ccx[px] = (int)(short)(c[0] - c[2]);
                ccy[px] = (int)(short)(c[0] - (c[1] << 1) + c[2]);
                cenX = (int)(short)(cenX + ccx[px]);
                cenY = (int)(short)(cenY + ccy[px]);
            }
#pragma unroll
            for (int px = 0; px < 16; px++)
            {
                // x 16, or x the number of opaque pixels in the punch-through encoder (ETC.cpp:1746-1765)
                cdx[px] = (float)(int)(short)((int)(short)(ccx[px] * numOpaqueJ) - cenX);
                cdy[px] = (float)(int)(short)((int)(short)(ccy[px] * numOpaqueJ) - cenY) * 0.57735026918962576450914878050196f;
            }
        }
        else
        {
            float ccx[16], ccy[16], cenX = 0.0f, cenY = 0.0f;
#pragma unroll
            for (int px = 0; px < 16; px++)
            {
                float w3[3];
                pwJ(px, w3);
                ccx[px] = w3[0] * A.axis0[0] + w3[1] * A.axis0[1] + w3[2] * A.axis0[2];
                ccy[px] = w3[0] * A.axis1[0] + w3[1] * A.axis1[1] + w3[2] * A.axis1[2];
            }
#pragma unroll
            for (int px = 0; px < 16; px++)
            {
                cenX = cenX + ccx[px];
                cenY = cenY + ccy[px];
            }
#pragma unroll
            for (int px = 0; px < 16; px++)
            {
                cdx[px] = ccx[px] * (float)numOpaqueJ - cenX;
                cdy[px] = ccy[px] * (float)numOpaqueJ - cenY;
            }
        }
        float covXX = 0.0f, covYY = 0.0f, covXY = 0.0f;
#pragma unroll
        for (int px = 0; px < 16; px++)
        {
            covXX = covXX + cdx[px] * cdx[px];
            covYY = covYY + cdy[px] * cdy[px];
            covXY = covXY + cdx[px] * cdy[px];
        }
        const float halfTrace = (covXX + covYY) * 0.5f;
        const float det = covXX * covYY - covXY * covXY;
        const float mm = sqrtExact(sseMax(0.0f, halfTrace * halfTrace - det));
        const float ev = halfTrace + mm;
        float dx = (covYY - ev + covXY);
        const float dy = -(covXX - ev + covXY);
        if (dx == 0.0f && dy == 0.0f)
            dx = 1.0f;
#pragma unroll
        for (int px = 0; px < 16; px++)
            if ((cdx[px] * dx + cdy[px] * dy) < 0.0f)
                isoJ |= 1u << px;
    }
    isolatedMask = (u32)__shfl((int)isoJ, own);
    // =================================== T mode x2 ===================================
    for (int call = 0; call < 2 && !groupAll; call++)
    {
        const u32 iso = call == 0 ? isolatedMask : (~isolatedMask & 0xffffu);
        int isolatedTotal[3] = {0, 0, 0}, lineTotal[3] = {0, 0, 0};
        const int numIsolated = __popc(iso);
#pragma unroll
        for (int px = 0; px < 16; px++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                const int v = S.pix[px][ch];
                lineTotal[ch] += v;
                if ((iso >> px) & 1u)
                    isolatedTotal[ch] += v;
            }
        const int numLine = 16 - numIsolated;
        int isoQ[3], isoColor[3], isoTargets[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
        {
            lineTotal[ch] -= isolatedTotal[ch];
            const int numerator = isolatedTotal[ch] + isolatedTotal[ch] + (FAKE ? 0 : ((numIsolated << 4) | numIsolated));
            isoQ[ch] = numIsolated == 0 ? 0 : udivSmall(numerator, numIsolated * 34);
            isoTargets[ch] = numerator;
        }
        if (FAKE)
            resolveTHFake(isoQ, isoTargets, numIsolated); // may push a channel to 16 (ETC.cpp:453-454)
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
            isoColor[ch] = isoQ[ch] | (isoQ[ch] << 4);
        {
            const int px = lane & 15;
            S.isoErr[px] = E(isoColor[0], isoColor[1], isoColor[2], S.pix[px], S.pw[px]);
        }
        // unique line colours of this block (ETC.cpp:494-560): all 8 x (2 numLine + 1) (table, premultiplier) candidates in
        // parallel, then the order-dependent removal of consecutive duplicates per table (in place, by ballot prefix)
        {
            const int span = 2 * numLine + 1; // <= 33
            const int lineDivisor = numLine * 34;
            const int lineAddend = (numLine << 4) | numLine;
            for (int id = lane; id < 8 * span; id += 64)
            {
                const int tbl = udivSmall(id, span), kk = id - tbl * span; // (id < 2^16, span <= 33: the float form, not a 32-bit division)
                const int modifierAddend = (int)(short)((kk - numLine) * (thDist(tbl) * 2));
                int packed = 0, q3[3], targets[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    int numerator = (int)(short)((int)(short)(lineTotal[ch] + lineTotal[ch] + (FAKE ? 0 : lineAddend)) + modifierAddend);
                    numerator = numerator < 0 ? 0 : numerator;
                    const int divided = lineDivisor == 0 ? 0 : udivSmall(numerator, lineDivisor);
                    q3[ch] = divided < 15 ? divided : 15;
                    targets[ch] = numerator;
                }
                if (FAKE)
                    resolveTHFake(q3, targets, numLine);
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    packed |= q3[ch] << (ch * 5);
                S.tColors[tbl][kk] = (unsigned short)packed;
            }
            WAVE_SYNC();
            for (int tbl = 0; tbl < 8; tbl++)
            {
                const bool in = lane < span;
                const int cur = in ? (int)S.tColors[tbl][lane] : -1;
                const int prev = (in && lane > 0) ? (int)S.tColors[tbl][lane - 1] : -1;
                const bool keep = in && (lane == 0 || cur != prev);
                const u64 bal = __ballot(keep);
                WAVE_SYNC(); // every lane has read before anyone writes
                if (keep)
                    S.tColors[tbl][__popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)cur;
                if (lane == 0)
                    S.tCount[tbl] = __popcll(bal);
            }
            WAVE_SYNC();
        }
        // Candidates in the reference's order: per table its unique colours, then -- hazard H2, only when another block of
        // the group has MORE unique colours for that table -- one slot that reads as colour 0 (the copies of colour 0
        // behind it repeat candidate 0 and can never pass the strict '<').  Whether the slot exists takes the other seven
        // blocks' counts, which cost more than this block's whole candidate generation, and the black line colour almost
        // never wins; so the slot is evaluated as "tentative" and the counts are only computed when a tentative candidate
        // would be the winner.
        // Work order: the real candidates table by table, then the eight zero slots (they usually fit in the idle lanes of
        // the last pass).  Reference order, for ties: key = table << 8 | position in the table's list.
        int prefix[9];
        prefix[0] = 0;
#pragma unroll
        for (int t = 0; t < 8; t++)
            prefix[t + 1] = __builtin_amdgcn_readfirstlane(prefix[t] + S.tCount[t]); // wave-uniform: scalar registers
        // attempt 0: zero slots tracked apart (tentative); attempt 1 (rare): the slots of the tables in presentMask are
        // candidates like the others, the rest is skipped.  One copy of the evaluation loop serves both.
        float wErr = FLT_MAX;
        int wId = 0x7fffffff;
        u32 presentMask = 0;
        for (int attempt = 0; attempt < 2; attempt++)
        {
            const bool exact = attempt == 1;
            float rErr = FLT_MAX, zErr = FLT_MAX;
            int rId = 0x7fffffff, zId = 0x7fffffff;
            for (int base = 0; base < prefix[8] + 8; base += 64)
            {
                const int e = base + lane;
                if (e < prefix[8] + 8)
                {
                    const bool zeroSlot = e >= prefix[8];
                    int table = 0, start = 0;
#pragma unroll
                    for (int t = 1; t < 8; t++)
                    {
                        const bool ge = e >= prefix[t];
                        table = ge ? t : table;
                        start = ge ? prefix[t] : start;
                    }
                    if (zeroSlot)
                        table = e - prefix[8];
                    const int ci = zeroSlot ? S.tCount[table] : e - start;
                    const int id = (table << 8) | ci;
                    if (zeroSlot && exact && !((presentMask >> table) & 1u))
                        continue;
                    const int packed = zeroSlot ? 0 : (int)S.tColors[table][ci];
                    const int modifier = thDist(table);
                    int lc[3][3];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const int q = (packed >> (ch * 5)) & 15;
                        const int u = (q << 4) | q;
                        lc[0][ch] = u + modifier < 255 ? u + modifier : 255;
                        lc[1][ch] = u;
                        lc[2][ch] = u - modifier > 0 ? u - modifier : 0;
                    }
                    float error = 0.0f;
                    EtcErr::f32x2 lw01[3];
                    E.weigh2(lw01, lc[0], lc[1]);
                    // the metric is a run-time flag: decided once per pass, not per pixel (as a test inside the loop it was four
                    // branches per pixel and kept the sixteen trips from being unrolled)
                    auto tPixels = [&](auto uniTag) {
                        constexpr bool UNI = decltype(uniTag)::value;
                        const EtcErr EE = {UNI, E.rw, E.gw, E.bw, E.fake};
                        EtcWaveShared &S = shared1; // (static storage: named again here, a generic lambda does not capture the outer reference)
#pragma unroll 4
                        for (int px = 0; px < 16; px++)
                        {
                            float pixelError = S.isoErr[px];
                            float e3[3];
                            if (!UNI)
                            {
                                const EtcErr::f32x2 e01 = EE.err2(lw01, S.pw[px]);
                                e3[0] = e01.x;
                                e3[1] = e01.y;
                            }
                            else
                            {
                                e3[0] = EE.wu(lc[0][0], lc[0][1], lc[0][2], S.pix[px], S.pw[px]);
                                e3[1] = EE.wu(lc[1][0], lc[1][1], lc[1][2], S.pix[px], S.pw[px]);
                            }
                            e3[2] = EE.wu(lc[2][0], lc[2][1], lc[2][2], S.pix[px], S.pw[px]); // sic: never the fake metric
                            // only the candidate's error (sums of squares: three v_min_f32); the selectors are worked out again
                            // for the one candidate that wins
                            pixelError = __builtin_fminf(minLoaded(pixelError, e3[0]), __builtin_fminf(e3[1], e3[2]));
                            error = error + pixelError;
                        }
                    };
                    if (E.uniform)
                        tPixels(std::true_type{});
                    else
                        tPixels(std::false_type{});
                    if (zeroSlot && !exact)
                    {
                        if (error < zErr || (error == zErr && id < zId))
                        {
                            zErr = error;
                            zId = id;
                        }
                    }
                    else if (error < rErr || (error == rErr && id < rId)) // the reference keeps the first in ITS order
                    {
                        rErr = error;
                        rId = id;
                    }
                }
            }
            wErr = rErr;
            wId = rId;
            waveArgmin(wErr, wId);
            if (exact)
                break;
            float wzErr = zErr;
            int wzId = zId;
            waveArgmin(wzErr, wzId);
            if (!(wzErr < bestError && (wzErr < wErr || (wzErr == wErr && wzId < wId))))
                break;
            // a zero slot would win: now the group's counts are needed (lane = (group member jb, table), counting only)
#ifdef CVTT_ETC_PROFILE
            if (lane == 0)
                atomicAdd(&g_etcProf[6], 1ull);
#endif
            int nMember;
            {
                const int tbl = lane >> 3;
                const u32 isoM = call == 0 ? isoJ : (~isoJ & 0xffffu);
                int lineTotalM[3];
                lineTotalsJ(~isoM & 0xffffu, lineTotalM);
                const int numLineM = 16 - __popc(isoM);
                const int modifierOffset = thDist(tbl) * 2;
                const int lineDivisor = numLineM * 34;
                const int lineAddend = (numLineM << 4) | numLineM;
                int n = 0, last = -1;
                const int kMax = __builtin_amdgcn_readfirstlane(groupMax(numLineM));
                for (int k = -kMax; k <= kMax; k++)
                {
                    if (k < -numLineM || k > numLineM)
                        continue;
                    const int modifierAddend = (int)(short)(k * modifierOffset);
                    int packed = 0, q3[3], targets[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        int numerator = (int)(short)((int)(short)(lineTotalM[ch] + lineTotalM[ch] + (FAKE ? 0 : lineAddend)) + modifierAddend);
                        numerator = numerator < 0 ? 0 : numerator;
                        const int divided = lineDivisor == 0 ? 0 : udivSmall(numerator, lineDivisor);
                        q3[ch] = divided < 15 ? divided : 15;
                        targets[ch] = numerator;
                    }
                    if (FAKE)
                        resolveTHFake(q3, targets, numLineM);
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                        packed |= q3[ch] << (ch * 5);
                    if (n == 0 || packed != last)
                    {
                        n++;
                        last = packed;
                    }
                }
                nMember = groupMax(n);
            }
            presentMask = 0;
#pragma unroll
            for (int t = 0; t < 8; t++)
                if (laneI(nMember, t * 8) > S.tCount[t])
                    presentMask |= 1u << t;
        }
        if (wErr < bestError)
        {
            bestError = wErr;
            const int table = wId >> 8, ci = wId & 255;
            const int packed = ci < S.tCount[table] ? (int)S.tColors[table][ci] : 0;
            const int lineColor[3] = {packed & 15, (packed >> 5) & 15, (packed >> 10) & 15};
            // the winner's selectors: lane px works out pixel px again (the operations of the candidate loop: first minimum
            // among the isolated colour and the three line colours, strict '<')
            u32 selectors = 0;
            {
                const int px = lane & 15;
                const int modifier = thDist(table);
                int lc[3][3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    const int u = (lineColor[ch] << 4) | lineColor[ch];
                    lc[0][ch] = u + modifier < 255 ? u + modifier : 255;
                    lc[1][ch] = u;
                    lc[2][ch] = u - modifier > 0 ? u - modifier : 0;
                }
                float e3[3];
                if (!E.uniform)
                {
                    EtcErr::f32x2 lw01[3];
                    E.weigh2(lw01, lc[0], lc[1]);
                    const EtcErr::f32x2 e01 = E.err2(lw01, S.pw[px]);
                    e3[0] = e01.x;
                    e3[1] = e01.y;
                }
                else
                {
                    e3[0] = E.wu(lc[0][0], lc[0][1], lc[0][2], S.pix[px], S.pw[px]);
                    e3[1] = E.wu(lc[1][0], lc[1][1], lc[1][2], S.pix[px], S.pw[px]);
                }
                e3[2] = E.wu(lc[2][0], lc[2][1], lc[2][2], S.pix[px], S.pw[px]);
                float pixelError = S.isoErr[px];
                u32 sel = 0;
#pragma unroll
                for (int i = 0; i < 3; i++)
                    if (e3[i] < pixelError)
                    {
                        sel = (u32)(i + 1);
                        pixelError = e3[i];
                    }
                const u64 b0 = __ballot(lane < 16 && (sel & 1u)), b1 = __ballot(lane < 16 && (sel & 2u));
#pragma unroll
                for (int i = 0; i < 16; i++)
                    selectors |= (u32)(((b0 >> i) & 1ull) | (((b1 >> i) & 1ull) << 1)) << (2 * i);
            }
            emitT(outHi, outLo, lineColor, isoQ, selectors, table);
        }
        DBG_TAP(1 + call);
        bestError = uniF(bestError); outHi = uniU(outHi); outLo = uniU(outLo);
        WAVE_SYNC();
    }

    // =================================== H mode ===================================
    // HAND-OVER POINT of the stage union: up to here `S.u.group` (the group's pixels, read through pixJ / pwJ by the sector
    // split and the T modes' group counts) was alive; the H mode's `S.u.h` and the cluster fit's `S.u.a` overwrite it.  The
    // WAVE_SYNC at the end of the T-mode loop above (a wavefront-scope fence) orders every read of the group before the stores
    // below; no pixJ / pwJ call may be added after this line for MODE 0 / 1 (the punch-through instantiation keeps its group
    // pixels in LDS of their own).
    // groupings = the flipped sector assignment (ETC.cpp:1855-1860)
    if (!groupAll)
    {
        const u32 grp = ~isolatedMask & 0xffffu;
        int counts[2], totals[2][3] = {{0, 0, 0}, {0, 0, 0}};
        counts[1] = __popc(grp);
        counts[0] = 16 - counts[1];
#pragma unroll
        for (int px = 0; px < 16; px++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                const int v = S.pix[px][ch];
                totals[0][ch] += v;
                if ((grp >> px) & 1u)
                    totals[1][ch] += v;
            }
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
            totals[0][ch] -= totals[1][ch];

        float hBestErr = FLT_MAX; // best H candidate so far (only committed if it beats bestError)
        int hBestId = 0x7fffffff;
        int hBestC0 = 0, hBestC1 = 0;

        // base colours of all eight tables and both sectors (ETC.cpp:691-737), one (table, sector, premultiplier) per lane
        // and pass: 8 x 34 candidates; then the order-dependent removal of consecutive duplicates, one lane per
        // (table, sector), in place.  Lists: S.dColors[table * 2 + sector][], counts: S.dCount[table * 2 + sector]
        // (both free until the cluster fit).
        {
            const int perTable = 2 * counts[0] + 1 + 2 * counts[1] + 1; // 34
            for (int id = lane; id < 8 * perTable; id += 64)
            {
                const int table = udivSmall(id, perTable), j = id - table * perTable;
                const int sector = j < 2 * counts[0] + 1 ? 0 : 1;
                const int kk = sector ? j - (2 * counts[0] + 1) : j;
                const int cnt = sector ? counts[1] : counts[0];
                const int k = kk - cnt;
                const int modifier = thDist(table);
                int q[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    const int tt = sector ? totals[1][ch] : totals[0][ch];
                    if (cnt == 0)
                        q[ch] = 0;
                    else
                    {
                        int v = (int)(short)(tt * 2 + cnt * 17 + modifier * 2 * k);
                        v = v < 0 ? 0 : v;
                        v = udivSmall(v, cnt * 34);
                        q[ch] = v < 15 ? v : 15;
                    }
                }
                S.dColors[table * 2 + sector][kk] = (unsigned short)((q[0] << 10) | (q[1] << 5) | q[2]);
            }
            WAVE_SYNC();
            if (lane < 16)
            {
                const int cnt = (lane & 1) ? counts[1] : counts[0];
                int n = 0, last = -1;
                for (int kk = 0; kk <= 2 * cnt; kk++)
                {
                    const int packed = S.dColors[lane][kk];
                    if (n == 0 || packed != last)
                    {
                        S.dColors[lane][n++] = (unsigned short)packed; // n <= kk: in place
                        last = packed;
                    }
                }
                S.dCount[lane] = n;
            }
            WAVE_SYNC();
        }
        // two tables per step, so that the (at most 34-entry) error-row pass and the last pair pass of a table do not leave
        // most lanes idle
        for (int tp = 0; tp < 4; tp++)
        {
            const int tabA = 2 * tp, tabB = tabA + 1;
            const int nA0 = __builtin_amdgcn_readfirstlane(S.dCount[tabA * 2]), nA1 = __builtin_amdgcn_readfirstlane(S.dCount[tabA * 2 + 1]),
                      nB0 = __builtin_amdgcn_readfirstlane(S.dCount[tabB * 2]), nB1 = __builtin_amdgcn_readfirstlane(S.dCount[tabB * 2 + 1]); // wave-uniform
            const int rowsA = nA0 + nA1, rowsB = nB0 + nB1; // table B's rows follow table A's
            // per-colour error rows: lane = colour (ETC.cpp:752-787)
            for (int base = 0; base < rowsA + rowsB; base += 64)
            {
                const int r = base + lane;
                if (r < rowsA + rowsB)
                {
                    const bool isB = r >= rowsA;
                    const int table = isB ? tabB : tabA;
                    const int li = isB ? r - rowsA : r;
                    const int n0 = isB ? nB0 : nA0;
                    const int modifier = thDist(table);
                    const int packed = li < n0 ? S.dColors[table * 2][li] : S.dColors[table * 2 + 1][li - n0];
                    int c0[3], c1[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const int q = (packed >> ((2 - ch) * 5)) & 15;
                        const int u = (q << 4) | q;
                        c0[ch] = u + modifier < 255 ? u + modifier : 255;
                        c1[ch] = u - modifier > 0 ? u - modifier : 0;
                    }
                    EtcErr::f32x2 cw[3];
                    E.weigh2(cw, c0, c1);
                    // (the metric flag is decided once per pass, as in the T modes)
                    auto hPixels = [&](auto uniTag) {
                        constexpr bool UNI = decltype(uniTag)::value;
                        const EtcErr EE = {UNI, E.rw, E.gw, E.bw, E.fake};
                        EtcWaveShared &S = shared1;
#pragma unroll 4
                        for (int px = 0; px < 16; px++)
                        {
                            float e0, e1;
                            if (!FAKE && !UNI)
                            {
                                const EtcErr::f32x2 e01 = EE.err2(cw, S.pw[px]);
                                e0 = e01.x;
                                e1 = e01.y;
                            }
                            else
                            {
                                e0 = EE(c0[0], c0[1], c0[2], S.pix[px], S.pw[px]);
                                e1 = EE(c1[0], c1[1], c1[2], S.pix[px], S.pw[px]);
                            }
                            S.u.h.err[px][r] = __builtin_fminf(e0, e1); // (which of the two it was is worked out again for the winning pair)
                        }
                    };
                    if (E.uniform)
                        hPixels(std::true_type{});
                    else
                        hPixels(std::false_type{});
                }
            }
            WAVE_SYNC();
            // colour pairs in the reference's odometer order (ETC.cpp:797-812): step s = 1..n0*n1 visits
            // (s % n0, min(n1 - 1, s / n0)); the last step wraps to (0, n1 - 1), which is a repeat
            // unless n1 == 1, where it is the only visit of (0, 0).  Table A's steps, then table B's.
            const int pairsA = nA0 * nA1, pairsB = nB0 * nB1;
            for (int base = 1; base <= pairsA + pairsB; base += 64)
            {
                const int kk = base + lane;
                if (kk <= pairsA + pairsB)
                {
                    const bool isB = kk > pairsA;
                    const int table = isB ? tabB : tabA;
                    const int k = isB ? kk - pairsA : kk;
                    const int n0 = isB ? nB0 : nA0, n1 = isB ? nB1 : nA1;
                    const int rowBase = isB ? rowsA : 0;
                    const int kq = udivSmall(k, n0); // (k < 2^11, n0 <= 18)
                    const int i0 = k - kq * n0;
                    const int i1 = kq < n1 - 1 ? kq : n1 - 1;
                    const int ci0 = rowBase + i0, ci1 = rowBase + n0 + i1;
                    // only the pair's error: which colour and which sign every pixel takes is worked out again for the one pair
                    // that wins (below), instead of being carried as two bit masks through every pixel of every pair
                    float totalError = 0.0f;
#pragma unroll
                    for (int px = 0; px < 16; px++)
                        totalError = totalError + minLoaded(S.u.h.err[px][ci0], S.u.h.err[px][ci1]); // sums of squares: no NaN, no -0
                    const int id = table * 1024 + k;
                    if (totalError < hBestErr || (totalError == hBestErr && id < hBestId))
                    {
                        hBestErr = totalError;
                        hBestId = id;
                        hBestC0 = S.dColors[table * 2][i0];
                        hBestC1 = S.dColors[table * 2 + 1][i1];
                    }
                }
            }
            WAVE_SYNC();
        }
        float wErr = hBestErr;
        int wId = hBestId;
        waveArgmin(wErr, wId);
        if (wErr < bestError)
        {
            bestError = wErr;
            // the winner's data sits in the lane that evaluated it
            const bool mine = (hBestId == wId) && (hBestErr == wErr);
            const u64 who = __ballot(mine);
            const int src = __ffsll((long long)who) - 1;
            const int bc0 = laneI(hBestC0, src), bc1 = laneI(hBestC1, src);
            const int table = wId >> 10;
            // the winner's sector and sign bits: lane px works out pixel px again (the operations of the colour rows above)
            u32 sectorBits, signBits;
            {
                const int px = lane & 15;
                const int modifier = thDist(table);
                float rowErr[2];
                bool rowSign[2];
#pragma unroll
                for (int w = 0; w < 2; w++)
                {
                    const int packed = w ? bc1 : bc0;
                    int c0[3], c1[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const int q = (packed >> ((2 - ch) * 5)) & 15;
                        const int u = (q << 4) | q;
                        c0[ch] = u + modifier < 255 ? u + modifier : 255;
                        c1[ch] = u - modifier > 0 ? u - modifier : 0;
                    }
                    float e0, e1;
                    if (!FAKE && !E.uniform)
                    {
                        EtcErr::f32x2 cw[3];
                        E.weigh2(cw, c0, c1);
                        const EtcErr::f32x2 e01 = E.err2(cw, S.pw[px]);
                        e0 = e01.x;
                        e1 = e01.y;
                    }
                    else
                    {
                        e0 = E(c0[0], c0[1], c0[2], S.pix[px], S.pw[px]);
                        e1 = E(c1[0], c1[1], c1[2], S.pix[px], S.pw[px]);
                    }
                    rowSign[w] = e1 < e0;
                    rowErr[w] = rowSign[w] ? e1 : e0;
                }
                const bool oneBetter = rowErr[1] < rowErr[0];
                sectorBits = (u32)(__ballot(lane < 16 && oneBetter) & 0xffffull);
                signBits = (u32)(__ballot(lane < 16 && (oneBetter ? rowSign[1] : rowSign[0])) & 0xffffull);
            }
            emitH(outHi, outLo, bc0, bc1, sectorBits, signBits, table, true);
        }
        WAVE_SYNC();
    }

    DBG_TAP(3);
    bestError = uniF(bestError); outHi = uniU(outHi); outLo = uniU(outLo);
    } // !ETC1
    // ====================== ETC1 cluster fit ======================
    // ETC2 reaches it through CompressETC2Block, which asks for the differential mode only (d = 1, ETC.cpp:1862);
    // EncodeETC1 also tries the individual 4:4:4 + 4:4:4 mode (d = 0) of each flip first.  punch = true is
    // CompressETC1PunchthroughBlockInternal (ETC.cpp:2885-3082): three paint colours + "transparent" per half block
    auto clusterFit = [&](const bool punch)
    {
        ETC_REFRESH_LANE();
        bool etcBest = false;
        int bFlip = 0, bD = 1;
        u32 bPacked0 = 0, bPacked1 = 0; // selectors | colour << 16
        int bTable0 = 0, bTable1 = 0;

        for (int flip = 0; flip < 2; flip++)
        for (int d = ETC1 ? 0 : 1; d < 2; d++)
        {
            // half-block membership: flip 0 = left/right 2x4 columns, flip 1 = top/bottom (g_flipTables, ETC.cpp:47-57)
            // pixel list of (flip, sector): computed on the fly
            // ---- base colours of the cluster fit (ETC.cpp:2690-2760): all 2 x 624 (sector, table, offset) candidates in
            // parallel, with the order-dependent removal of consecutive duplicates folded in ----
            if (!punch)
            {
                int cum[2][3] = {{0, 0, 0}, {0, 0, 0}};
#pragma unroll
                for (int px = 0; px < 16; px++)
                {
                    const int inSector = flip == 0 ? ((px >> 1) & 1) : (px >> 3);
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const int v = S.pix[px][ch];
                        cum[0][ch] += inSector ? 0 : v;
                        cum[1][ch] += inSector ? v : 0;
                    }
                }
                // One pass over the 624 (table, offset) pairs, 64 per trip, both sectors per lane (they share table and offset).
                // A colour stays iff it differs from its predecessor in its list (ETC.cpp:2762-2775): the predecessor is the
                // neighbouring lane's colour (the last lane's of the previous trip for lane 0), so the kept colours are filed
                // straight into their lists at ballot-prefix positions -- no raw array, no second pass of LDS round trips.  A trip
                // covers at most two tables (the shortest list has 57 entries): `A` = the table of lane 0, `B` = the next one.
                // (table, offset and "first entry of its table" of entry f come as one word of a table made on the host: derived
                // from f with a comparison chain they were 21 instructions per lane and trip)
                int keptA[2] = {0, 0}, prevLast[2] = {-1, -1}, curT = 0; // wave-uniform
                u32 entryNext = T->clusterEntry[lane];
                for (int base = 0; base < kMaxAttempts; base += 64)
                {
                    const int f = base + lane;
                    const bool in = f < kMaxAttempts;
                    const u32 entry = entryNext;
                    if (base + 64 < kMaxAttempts)
                        entryNext = T->clusterEntry[f + 64];
                    const int off = (int)(short)(entry & 0xffffu);
                    const int table = (int)((entry >> 16) & 7u);
                    const bool firstOfTable = (entry & (1u << 19)) != 0;
                    const int tLo = __builtin_amdgcn_readfirstlane(table);
                    if (tLo != curT)
                    {
                        // the previous table ended exactly at the end of the previous trip
                        if (lane == 0)
                        {
                            S.dCount[curT] = keptA[0];
                            S.dCount[8 + curT] = keptA[1];
                        }
                        keptA[0] = keptA[1] = 0;
                        curT = tLo;
                    }
                    const bool isA = table == tLo;
                    const bool crossing = __ballot(in && !isA) != 0;
#pragma unroll
                    for (int sector = 0; sector < 2; sector++)
                    {
                        int packed = 0;
                        if (FAKE)
                        {
                            int offsetCumulative[3], q3[3];
#pragma unroll
                            for (int ch = 0; ch < 3; ch++)
                            {
                                const int cu = (int)(short)(cum[sector][ch] + off);
                                offsetCumulative[ch] = cu < 0 ? 0 : (cu > 2040 ? 2040 : cu);
                            }
                            resolveHalfFake(q3, offsetCumulative, d == 1, fakeAccurate, T);
                            packed = q3[0] | (q3[1] << 5) | (q3[2] << 10);
                        }
                        else
                        {
#pragma unroll
                            for (int ch = 0; ch < 3; ch++)
                            {
                                int cu = (int)(short)(cum[sector][ch] + off);
                                cu = cu < 0 ? 0 : (cu > 2040 ? 2040 : cu);
                                const u32 q = d == 1 ? ((((u32)cu << 5) - (u32)cu + ((u32)cu >> 3) + 1024u) & 0xffffu) >> 11
                                                     : ((((u32)cu << 5) - ((u32)cu << 1) + ((u32)cu >> 3) + 2048u) & 0xffffu) >> 12;
                                packed |= (int)q << (ch * 5);
                            }
                        }
                        // the neighbouring lane's colour: a DPP wave shift (no LDS round trip in the middle of the chain that decides `keep`)
                        int prevCol = __builtin_amdgcn_update_dpp(packed, packed, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                        if (lane == 0)
                            prevCol = prevLast[sector];
                        const bool keep = in && (firstOfTable || packed != prevCol);
                        const u64 balA = __ballot(keep && isA), balB = __ballot(keep && !isA);
                        const u64 below = (1ull << lane) - 1ull;
                        const int pos = isA ? keptA[sector] + __popcll(balA & below) : __popcll(balB & below);
                        if (keep)
                            S.dColors[sector * 8 + table][pos] = (unsigned short)packed;
                        if (crossing)
                        {
                            if (lane == 0)
                                S.dCount[sector * 8 + tLo] = keptA[sector] + __popcll(balA);
                            keptA[sector] = __popcll(balB);
                        }
                        else
                            keptA[sector] += __popcll(balA);
                        prevLast[sector] = __builtin_amdgcn_readlane(packed, 63);
                    }
                    if (crossing)
                        curT = tLo + 1;
                }
                if (lane == 0)
                {
                    S.dCount[curT] = keptA[0];
                    S.dCount[8 + curT] = keptA[1];
                }
            }
            if (punch && lane < 16)
            {
                const int sector = lane >> 3, table = lane & 7;
                int cumulative[3] = {0, 0, 0};
#pragma unroll
                for (int px = 0; px < 16; px++)
                {
                    const int inSector = flip == 0 ? ((px >> 1) & 1) : (px >> 3);
                    if (inSector == sector)
                    {
                        cumulative[0] += S.pix[px][0];
                        cumulative[1] += S.pix[px][1];
                        cumulative[2] += S.pix[px][2];
                    }
                }
                const int numOffsets = T->clusterCount[table];
                const int start = T->clusterStart[table];
                int n = 0, last = -1;
                if (punch)
                {
                    // base colours: cumulative + om * modifier, divided by the number of TRANSPARENT pixels of the half
                    // (sic, ETC.cpp:2949-2951), om = -count..count
                    const u32 sectorMask = flip == 0 ? (sector ? 0xccccu : 0x3333u) : (sector ? 0xff00u : 0x00ffu);
                    const int count = __popc(transMask & sectorMask);
                    const int denominator = (count > 1 ? count : 1) << 8, addend = count << 7, cumulativeMax = 255 * count;
                    const int modifier = largeMod(table);
                    for (int om = -count; om <= count; om++)
                    {
                        const int off = (int)(short)(om * modifier);
                        int packed = 0;
#pragma unroll
                        for (int ch = 0; ch < 3; ch++)
                        {
                            int cu = (int)(short)(cumulative[ch] + off);
                            cu = cu < 0 ? 0 : (cu > cumulativeMax ? cumulativeMax : cu);
                            const u32 numerator = ((((u32)cu << 5) - (u32)cu) + (((u32)cu >> 3) + (u32)addend)) & 0xffffu;
                            packed |= udivSmall((int)numerator, denominator) << (ch * 5);
                        }
                        if (n == 0 || packed != last)
                        {
                            S.dColors[lane][n++] = (unsigned short)packed;
                            last = packed;
                        }
                    }
                }
                S.dCount[lane] = n;
            }
            WAVE_SYNC();
#ifdef CVTT_ETC_PROFILE
            DBG_TAP(7); // base colours + duplicate removal
#endif
            int prefix[17];
            prefix[0] = 0;
#pragma unroll
            for (int i = 0; i < 16; i++)
                prefix[i + 1] = __builtin_amdgcn_readfirstlane(prefix[i] + S.dCount[i]); // wave-uniform: scalar registers, not seventeen per lane
            const int numA0 = prefix[8], numA1 = prefix[16] - prefix[8];

            // ---- TestHalfBlock per candidate: lane = candidate (ETC.cpp:94-149, 2793-2828) ----
            // A half-block candidate only ever matters through a pair whose sum is below the block's best so far
            // (FindBestDifferentialCombination, ETC.cpp:219-362: every use of an attempt's error is a '<' against that best or
            // against best - partner).  Its error is a sum of non-negative terms, so once the running sum of EVERY candidate of
            // a pass has reached the limit the rest of their pixels cannot change anything: the pass stops and leaves the
            // partial sums (all >= the limit) as the errors.  Limit: the best so far for sector 0; for sector 1, once sector 0
            // is complete, best - (cheapest sector-0 attempt) -- with a margin of 4 ulp of the best, so that even the float sum
            // of that pair cannot come out below the best.
            const float flipBest = bestError;
            float limit1 = flipBest;
            bool limit1Ready = false;
            for (int base = 0; base < prefix[16]; base += 64)
            {
                if (!limit1Ready && base >= prefix[8])
                {
                    WAVE_SYNC();
                    float mn = FLT_MAX;
                    for (int i = lane; i < numA0; i += 64)
                        mn = fminf(mn, S.u.a.err[0][i]);
                    mn = fminf(mn, xorLaneF<1>(mn)); mn = fminf(mn, xorLaneF<2>(mn)); mn = fminf(mn, xorLaneF<4>(mn));
                    mn = fminf(mn, xorLaneF<8>(mn)); mn = fminf(mn, xorLaneF<16>(mn)); mn = fminf(mn, xorLaneF<32>(mn));
                    const float l = (flipBest - mn) + flipBest * 2.4e-7f;
                    limit1 = (l < flipBest) ? l : flipBest; // (also when nothing of sector 0 is below the best: the difference is <= 0, every pass stops at once)
                    limit1Ready = true;
                }
                const int id = base + lane;
                if (id < prefix[16])
                {
                    // (list and first entry from the same comparisons: prefix[slot] afterwards is a second select chain)
                    int slot = 0, start = 0;
#pragma unroll
                    for (int i = 1; i < 16; i++)
                    {
                        const bool ge = id >= prefix[i];
                        slot = ge ? i : slot;
                        start = ge ? prefix[i] : start;
                    }
                    const int sector = slot >> 3, table = slot & 7;
                    const int packed = S.dColors[slot][id - start];
                    int modified[4][3];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const int q = (packed >> (ch * 5)) & 31;
                        const int u = d == 1 ? ((q << 3) | (q >> 2)) : ((q << 4) | q);
#pragma unroll
                        for (int s = 0; s < 4; s++)
                        {
                            const int v = u + (s == 0 ? -largeMod(table) : s == 1 ? -smallMod(table) : s == 2 ? smallMod(table) : largeMod(table));
                            modified[s][ch] = v < 0 ? 0 : (v > 255 ? 255 : v);
                        }
                    }
                    if (punch)
                    {
                        // TestHalfBlockPunchthrough, ETC.cpp:151-217: colour - m, colour, colour + m
                        const int m = largeMod(table);
#pragma unroll
                        for (int ch = 0; ch < 3; ch++)
                        {
                            const int q = (packed >> (ch * 5)) & 31;
                            const int u = (q << 3) | (q >> 2);
                            modified[0][ch] = (u > m ? u : m) - m;
                            modified[1][ch] = u;
                            modified[2][ch] = u + m < 255 ? u + m : 255;
                            modified[3][ch] = 0;
                        }
                    }
                    u32 selectors = 0;
                    float totalError = 0.0f;
                    if (!punch && !FAKE && !E.uniform)
                    {
                        // the weighted metric (ComputeErrorWeighted, ETC.cpp:70-80), the four paint colours of a candidate two
                        // at a time (EtcPair): each value goes through the reference's operations in the reference's order
                        // (product, difference, squares added left to right), so the sums are the same numbers; 16 packed
                        // instructions per pixel instead of 32 plain ones
                        typedef EtcPair pk2;
                        pk2 mw01[3], mw23[3];
#pragma unroll
                        for (int ch = 0; ch < 3; ch++)
                        {
                            const float w = ch == 0 ? E.rw : (ch == 1 ? E.gw : E.bw);
                            mw01[ch] = pk2{(float)modified[0][ch] * w, (float)modified[1][ch] * w};
                            mw23[ch] = pk2{(float)modified[2][ch] * w, (float)modified[3][ch] * w};
                        }
                        // two pixels per trip (the pair of a half-block row for flip 0, neighbours for flip 1): the loop is not
                        // unrolled because of its exit, and per pixel its bookkeeping was a fifth of the body
                        for (int sp2 = 0; sp2 < 4; sp2++)
                        {
#pragma unroll
                            for (int h = 0; h < 2; h++)
                            {
                            const int spx = sp2 * 2 + h;
                            const int px = flip == 0 ? ((spx >> 1) * 4 + (spx & 1) + sector * 2) : (spx + sector * 8);
                            const float *pw = S.pw[px];
                            const pk2 p0 = pk2{pw[0], pw[0]}, p1 = pk2{pw[1], pw[1]}, p2 = pk2{pw[2], pw[2]};
                            pk2 dd = mw01[0] - p0;
                            pk2 e01 = dd * dd;
                            dd = mw01[1] - p1;
                            e01 = e01 + dd * dd;
                            dd = mw01[2] - p2;
                            e01 = e01 + dd * dd;
                            dd = mw23[0] - p0;
                            pk2 e23 = dd * dd;
                            dd = mw23[1] - p1;
                            e23 = e23 + dd * dd;
                            dd = mw23[2] - p2;
                            e23 = e23 + dd * dd;
                            // only the candidate's error is kept (the winners' selectors are recomputed), and the four values are
                            // sums of squares -- no NaN, no -0 -- so the SSE minimum chain is three v_min_f32
                            const float be = __builtin_fminf(__builtin_fminf(__builtin_fminf(e01.x, e01.y), e23.x), e23.y);
                            totalError = totalError + be;
                            }
                            if (__ballot(totalError < (sector == 0 ? flipBest : limit1)) == 0)
                                break;
                        }
                    }
                    else
                    for (int spx = 0; spx < 8; spx++)
                    {
                        // g_flipTables[flip][sector][spx]
                        const int px = flip == 0 ? ((spx >> 1) * 4 + (spx & 1) + sector * 2) : (spx + sector * 8);
                        float be = FLT_MAX;
                        u32 bs = 0;
#pragma unroll
                        for (int s = 0; s < 4; s++)
                        {
                            if (punch && s == 3)
                                continue;
                            const float e = E(modified[s][0], modified[s][1], modified[s][2], S.pix[px], S.pw[px]);
                            if (e < be)
                                bs = (u32)s;
                            be = sseMin(e, be);
                        }
                        if (punch && ((transMask >> px) & 1u))
                            be = 0.0f; // a transparent pixel costs nothing
                        totalError = totalError + be;
                        selectors |= bs << (spx * 2);
                        if (__ballot(totalError < (sector == 0 ? flipBest : limit1)) == 0)
                            break;
                    }
                    const int pos = sector == 0 ? id : id - prefix[8];
                    S.u.a.err[sector][pos] = totalError;
                    (void)selectors;
                }
            }
            WAVE_SYNC();
#ifdef CVTT_ETC_PROFILE
            DBG_TAP(5); // TestHalfBlock over all candidates
#endif

            // ---- FindBestDifferentialCombination (ETC.cpp:219-362), wave-parallel scans ----
            const float blockBest0 = bestError;
            // cheapest attempt per sector (strict '<' in index order -> lowest index on ties)
            float m0 = FLT_MAX, m1 = FLT_MAX;
            int i0 = 0x7fffffff, i1 = 0x7fffffff;
            for (int i = lane; i < numA0; i += 64)
            {
                const float e = S.u.a.err[0][i];
                if (e < m0) { m0 = e; i0 = i; }
            }
            for (int i = lane; i < numA1; i += 64)
            {
                const float e = S.u.a.err[1][i];
                if (e < m1) { m1 = e; i1 = i; }
            }
            waveArgmin(m0, i0);
            waveArgmin(m1, i1);
            if (i0 == 0x7fffffff) i0 = 0;
            if (i1 == 0x7fffffff) i1 = 0;
            auto tableOf = [&](int sector, int pos) {
                int t = 0;
#pragma unroll
                for (int i = 1; i < 8; i++)
                    if (pos + (sector ? prefix[8] : 0) >= prefix[sector * 8 + i])
                        t = i;
                return t;
            };
            // colour (5:5:5) of attempt `pos` of a sector: which (table) list it falls in, then the list entry
            auto colorOf = [&](int sector, int pos) -> u32 {
                const int id = pos + (sector ? prefix[8] : 0);
                int slot = sector * 8, base = sector ? prefix[8] : 0;
#pragma unroll
                for (int i = 1; i < 8; i++)
                {
                    const int start = sector ? prefix[8 + i] : prefix[i];
                    if (id >= start)
                    {
                        slot = sector * 8 + i;
                        base = start;
                    }
                }
                return (u32)S.dColors[slot][id - base];
            };
            auto legal = [](u32 a, u32 b) {
                const int d2 = (int)(b >> 10) - (int)(a >> 10);
                const int d1 = (int)((b >> 5) & 31u) - (int)((a >> 5) & 31u);
                const int d0 = (int)(b & 31u) - (int)(a & 31u);
                return d2 >= -4 && d2 <= 3 && d1 >= -4 && d1 <= 3 && d0 >= -4 && d0 <= 3;
            };
            // punch-through: a fully transparent half 0 takes the colour of half 1 and is always legal; half 1 never gets
            // that treatment because the reference starts its flag as false (ETC.cpp:2938)
            const bool canIgnore0 = punch && (transMask & (flip == 0 ? 0x3333u : 0x00ffu)) == (flip == 0 ? 0x3333u : 0x00ffu);
            if (m0 + m1 < blockBest0)
            {
                const u32 p1 = colorOf(1, i1) << 16, p0 = canIgnore0 ? p1 : (colorOf(0, i0) << 16);
                if (d == 0 || canIgnore0 || legal(p0 >> 16, p1 >> 16)) // individual mode: the halves are unconstrained (ETC.cpp:2831-2851)
                {
                    etcBest = true;
                    bestError = m0 + m1;
                    bFlip = flip;
                    bD = d;
                    bPacked0 = p0;
                    bPacked1 = p1;
                    bTable0 = tableOf(0, i0);
                    bTable1 = tableOf(1, i1);
                }
                else
                {
                    // slow path: walk sector 0's attempts in (error, index) order without sorting;
                    // for each, the cheapest legal partner (lowest (error, index)) of sector 1.
                    // A step of the walk is two scans over all ~460 attempts of a sector, and a walk takes about twenty steps, most of
                    // them without a commit.  What the walk can still touch is small: sector-0 attempts after the current one
                    // that are below the best and pass the walk's own `best - e0 < m1` exit (a prefix of the sorted order, both
                    // tests being monotone in e0), and sector-1 attempts below best - m0 (the bound on every later maxError1).
                    // If both sets fit the wave they are dealt to the lanes (one entry each) and the remaining steps are two
                    // wave-wide minima instead of two scans; otherwise the walk goes on as before.  Same steps, same tests,
                    // same order: the sets only leave out what the walk's exits and the `pE < maxError1` test rule out.
                    float prevE = -1.0f;
                    int prevI = -1;
                    float blockBest = blockBest0;
                    bool compact = false;
                    DBG_COUNT(12, 1);
                    constexpr int kPerLane = 2, kCap = 64 * kPerLane; // entries of a set per lane / in all
                    float ce0[kPerLane], ce1[kPerLane];
                    int ci0[kPerLane], cj1[kPerLane];
                    u32 cc0[kPerLane], cc1[kPerLane];
#pragma unroll
                    for (int k = 0; k < kPerLane; k++)
                    {
                        ce0[k] = ce1[k] = FLT_MAX;
                        ci0[k] = cj1[k] = 0x7fffffff;
                        cc0[k] = cc1[k] = 0;
                    }
                    unsigned short *const stage = &S.tColors[0][0]; // 2 x kCap attempt numbers: the T modes' colour lists are dead by now
                    // deal what the walk can still touch to the lanes (see above); called before the first step and, while the sets
                    // are too large for the wave, again after every commit (each tightens them)
                    auto tryCompact = [&]() {
                        WAVE_SYNC();
                        int cnt0 = 0, cnt1 = 0;
                        const float bound1 = bestError - m0;
                        const u64 below = (1ull << lane) - 1ull;
                        for (int base = 0; base < numA0; base += 64)
                        {
                            const int i = base + lane;
                            const float e = i < numA0 ? S.u.a.err[0][i] : FLT_MAX;
                            const bool after = (e > prevE) || (e == prevE && i > prevI);
                            const bool alive = i < numA0 && after && e < blockBest0 && e < blockBest && !((bestError - e) < m1);
                            const u64 bal = __ballot(alive);
                            const int slot = cnt0 + __popcll(bal & below);
                            if (alive && slot < kCap)
                                stage[slot] = (unsigned short)i;
                            cnt0 += __popcll(bal);
                        }
                        for (int base = 0; base < numA1; base += 64)
                        {
                            const int j = base + lane;
                            const float e = j < numA1 ? S.u.a.err[1][j] : FLT_MAX;
                            const bool alive = j < numA1 && e < blockBest0 && e < bound1;
                            const u64 bal = __ballot(alive);
                            const int slot = cnt1 + __popcll(bal & below);
                            if (alive && slot < kCap)
                                stage[kCap + slot] = (unsigned short)j;
                            cnt1 += __popcll(bal);
                        }
                        WAVE_SYNC();
                        DBG_COUNT(9, 1);
                        DBG_COUNT(13, cnt0);
                        DBG_COUNT(14, cnt1);
                        if (cnt0 <= kCap && cnt1 <= kCap)
                        {
                            DBG_COUNT(10, 1);
                            compact = true;
#pragma unroll
                            for (int k = 0; k < kPerLane; k++)
                            {
                                if (lane + 64 * k < cnt0)
                                {
                                    ci0[k] = (int)stage[lane + 64 * k];
                                    ce0[k] = S.u.a.err[0][ci0[k]];
                                    cc0[k] = colorOf(0, ci0[k]);
                                }
                                if (lane + 64 * k < cnt1)
                                {
                                    cj1[k] = (int)stage[kCap + lane + 64 * k];
                                    ce1[k] = S.u.a.err[1][cj1[k]];
                                    cc1[k] = colorOf(1, cj1[k]);
                                }
                            }
                        }
                    };
                    tryCompact();
                    for (;;)
                    {
                        // next attempt of sector 0 in sorted order among those with error < blockBest0
                        float nE = FLT_MAX;
                        int nI = 0x7fffffff;
                        if (!compact)
                        {
                            for (int i = lane; i < numA0; i += 64)
                            {
                                const float e = S.u.a.err[0][i];
                                const bool after = (e > prevE) || (e == prevE && i > prevI);
                                if (e < blockBest0 && after && ((e < nE) || (e == nE && i < nI)))
                                {
                                    nE = e;
                                    nI = i;
                                }
                            }
                        }
                        else
                        {
                            // (FLT_MAX / 0x7fffffff where a lane has no entry or the entry has had its turn)
#pragma unroll
                            for (int k = 0; k < kPerLane; k++)
                                if (ce0[k] < nE || (ce0[k] == nE && ci0[k] < nI))
                                {
                                    nE = ce0[k];
                                    nI = ci0[k];
                                }
                        }
                        waveArgmin(nE, nI);
                        if (nI == 0x7fffffff)
                            break;
                        DBG_COUNT(compact ? 11 : 8, 1);
                        prevE = nE;
                        prevI = nI;
                        const float error0 = nE;
                        if (error0 >= blockBest)
                            break;
                        const float maxError1 = bestError - error0;
                        if (maxError1 < m1)
                            break;
                        u32 c0;
                        if (!compact)
                            c0 = colorOf(0, nI);
                        else
                        {
                            u32 mine = 0;
                            bool have = false;
#pragma unroll
                            for (int k = 0; k < kPerLane; k++)
                                if (ci0[k] == nI)
                                {
                                    mine = cc0[k];
                                    have = true;
                                    ce0[k] = FLT_MAX; // this entry has had its turn
                                    ci0[k] = 0x7fffffff;
                                }
                            const u64 who = __ballot(have);
                            c0 = laneI(mine, __ffsll((long long)who) - 1);
                        }
                        // the sorted scan of sector 1 stops at the first entry with error >= maxError1,
                        // so the partner is the cheapest LEGAL entry provided it is below maxError1
                        float pE = FLT_MAX;
                        int pI = 0x7fffffff;
                        if (!compact)
                        {
                            for (int j = lane; j < numA1; j += 64)
                            {
                                const float e = S.u.a.err[1][j];
                                if (e < blockBest0 && legal(c0, colorOf(1, j)) && ((e < pE) || (e == pE && j < pI)))
                                {
                                    pE = e;
                                    pI = j;
                                }
                            }
                        }
                        else
                        {
#pragma unroll
                            for (int k = 0; k < kPerLane; k++)
                                if (cj1[k] != 0x7fffffff && legal(c0, cc1[k]) && (ce1[k] < pE || (ce1[k] == pE && cj1[k] < pI)))
                                {
                                    pE = ce1[k];
                                    pI = cj1[k];
                                }
                        }
                        waveArgmin(pE, pI);
                        if (pI != 0x7fffffff && pE < maxError1)
                        {
                            blockBest = error0 + pE;
                            etcBest = true;
                            bestError = blockBest;
                            bFlip = flip;
                            bD = 1;
                            bPacked0 = c0 << 16;
                            bPacked1 = colorOf(1, pI) << 16;
                            bTable0 = tableOf(0, nI);
                            bTable1 = tableOf(1, pI);
                            if (!compact)
                                tryCompact();
                        }
                    }
                }
            }
            WAVE_SYNC();
            // the block's best after this (flip, mode): wave-uniform
            bestError = uniF(bestError);
            etcBest = uniI(etcBest ? 1 : 0) != 0;
            bFlip = uniI(bFlip); bD = uniI(bD);
            bPacked0 = uniU(bPacked0); bPacked1 = uniU(bPacked1);
            bTable0 = uniI(bTable0); bTable1 = uniI(bTable1);
        }

        if (etcBest)
        {
            // the winners' selectors again (TestHalfBlock, ETC.cpp:94-149): lane = sector * 8 + pixel of the half block
            {
                const int sector = (lane >> 3) & 1, spx = lane & 7;
                const u32 colr = (sector ? bPacked1 : bPacked0) >> 16;
                const int table = sector ? bTable1 : bTable0;
                const int px = bFlip == 0 ? ((spx >> 1) * 4 + (spx & 1) + sector * 2) : (spx + sector * 8);
                float be = FLT_MAX;
                u32 bs = 0;
#pragma unroll
                for (int sel = 0; sel < 4; sel++)
                {
                    if (punch && sel == 3)
                        continue;
                    int m[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const int q = (int)((colr >> (ch * 5)) & 31u);
                        const int u = bD == 1 ? ((q << 3) | (q >> 2)) : ((q << 4) | q);
                        int v = u + (sel == 0 ? -largeMod(table) : sel == 1 ? -smallMod(table) : sel == 2 ? smallMod(table) : largeMod(table));
                        if (punch)
                        {
                            const int md = largeMod(table);
                            v = sel == 0 ? (u > md ? u : md) - md : (sel == 1 ? u : u + md);
                        }
                        m[ch] = v < 0 ? 0 : (v > 255 ? 255 : v);
                    }
                    const float e = E(m[0], m[1], m[2], S.pix[px], S.pw[px]);
                    if (e < be)
                        bs = (u32)sel;
                    be = sseMin(e, be);
                }
                if (punch)
                {
                    // selector codes in table order: 0, 2, 3 are the colours, 1 is "transparent" (ETC.cpp:199-207)
                    bs = (bs << 1) < 3u ? (bs << 1) : 3u;
                    if ((transMask >> px) & 1u)
                        bs = 1u;
                }
                const u64 bit0 = __ballot(lane < 16 && (bs & 1u)), bit1 = __ballot(lane < 16 && (bs & 2u));
                u32 sel0 = 0, sel1 = 0;
#pragma unroll
                for (int i = 0; i < 8; i++)
                {
                    sel0 |= (u32)(((bit0 >> i) & 1ull) | (((bit1 >> i) & 1ull) << 1)) << (2 * i);
                    sel1 |= (u32)(((bit0 >> (8 + i)) & 1ull) | (((bit1 >> (8 + i)) & 1ull) << 1)) << (2 * i);
                }
                bPacked0 = (bPacked0 & 0xffff0000u) | sel0;
                bPacked1 = (bPacked1 & 0xffff0000u) | sel1;
            }
            // EmitETC1Block, ETC.cpp:2565-2622 (opaque)
            const u32 col0 = bPacked0 >> 16, col1 = bPacked1 >> 16;
            int colors[2][3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                colors[0][ch] = (int)((col0 >> (ch * 5)) & 31u);
                colors[1][ch] = (int)((col1 >> (ch * 5)) & 31u);
            }
            u32 hi = 0, lo = 0;
            if (bD == 0)
            {
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    hi |= (u32)colors[0][ch] << (28 - 8 * ch);
                    hi |= (u32)colors[1][ch] << (24 - 8 * ch);
                }
            }
            else
            {
                hi |= (u32)colors[0][0] << 27;
                hi |= (u32)((colors[1][0] - colors[0][0]) & 7) << 24;
                hi |= (u32)colors[0][1] << 19;
                hi |= (u32)((colors[1][1] - colors[0][1]) & 7) << 16;
                hi |= (u32)colors[0][2] << 11;
                hi |= (u32)((colors[1][2] - colors[0][2]) & 7) << 8;
            }
            hi |= (u32)bTable0 << 5;
            hi |= (u32)bTable1 << 2;
            if (!punch)
                hi |= (u32)bD << 1; // the opaque bit of the punch-through format stays clear
            hi |= (u32)bFlip;
            // selector -> modifier code {3, 2, 0, 1}, scattered through the flip table then column-major
            u32 codes = 0; // 2 bits per pixel position
#pragma unroll
            for (int sector = 0; sector < 2; sector++)
            {
                const u32 sels = (sector == 0 ? bPacked0 : bPacked1) & 0xffffu;
#pragma unroll
                for (int spx = 0; spx < 8; spx++)
                {
                    const int px = bFlip == 0 ? ((spx >> 1) * 4 + (spx & 1) + sector * 2) : (spx + sector * 8);
                    const u32 sel = (sels >> (2 * spx)) & 3u;
                    const u32 code = (0x4Bu >> (2 * sel)) & 3u; // {3,2,0,1} packed little-endian: 3 | 2<<2 | 0<<4 | 1<<6
                    codes |= code << (2 * px);
                }
            }
#pragma unroll
            for (int px = 0; px < 16; px++)
            {
                const int src = ((px & 3) << 2) | (px >> 2);
                const u32 code = (codes >> (2 * src)) & 3u;
                lo |= (code & 1u) << px;
                lo |= ((code >> 1) & 1u) << (16 + px);
            }
            outHi = hi;
            outLo = lo;
        }
    };

    if (!(PUNCH && groupAll))
        clusterFit(false);

    // ============ punch-through: "virtual T mode" x2 and the punch-through cluster fit (ETC.cpp:1865-1886) ============
    if (PUNCH && groupAny)
    {
        if (transMask != 0)
            bestError = FLT_MAX; // blocks with a transparent pixel start over; opaque blocks keep their opaque result
        for (int call = 0; call < 2; call++)
        {
            // EncodeVirtualTModePunchthrough, ETC.cpp:887-1264: T mode (isolated colour, line colour +/- m) and H mode
            // (line colour +/- m, second colour - m) with the remaining paint colour replaced by "transparent"
            const u32 base = call == 0 ? isolatedMask : (~isolatedMask & 0xffffu);
            const u32 iso = base & ~transMask, line = ~base & ~transMask & 0xffffu;
            int isolatedTotal[3] = {0, 0, 0}, lineTotal[3] = {0, 0, 0};
#pragma unroll
            for (int px = 0; px < 16; px++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    const int v = S.pix[px][ch];
                    if ((iso >> px) & 1u)
                        isolatedTotal[ch] += v;
                    if ((line >> px) & 1u)
                        lineTotal[ch] += v;
                }
            const int numIsolated = __popc(iso), numLine = __popc(line);
            const int isoAddend = (numIsolated << 4) | numIsolated;
            int isoQ[3], isoColor[3], isoTargets[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                isoTargets[ch] = isolatedTotal[ch] + isolatedTotal[ch] + (FAKE ? 0 : isoAddend);
                isoQ[ch] = numIsolated == 0 ? 0 : udivSmall(isoTargets[ch], numIsolated * 34);
            }
            if (FAKE)
                resolveTHFake(isoQ, isoTargets, numIsolated);
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                isoColor[ch] = isoQ[ch] | (isoQ[ch] << 4);
            {
                const int px = lane & 15;
                const float e = E(isoColor[0], isoColor[1], isoColor[2], S.pix[px], S.pw[px]);
                S.isoErr[px] = ((transMask >> px) & 1u) ? 0.0f : e;
            }
            // H-mode second colour per table and its error per pixel: lane = (table & 3) * 16 + pixel, two rounds
#pragma unroll
            for (int half = 0; half < 2; half++)
            {
                const int table = half * 4 + (lane >> 4), px = lane & 15;
                const int modifier = thDist(table);
                int hq[3], hc[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    const int offsetTotal = isolatedTotal[ch] + modifier * numIsolated;
                    int q = numIsolated == 0 ? 0 : udivSmall(offsetTotal + offsetTotal + isoAddend, numIsolated * 34);
                    q = q < 15 ? q : 15;
                    hq[ch] = q;
                    const int u = (q << 4) | q;
                    hc[ch] = u - modifier > 0 ? u - modifier : 0;
                }
                const float e = E.wu(hc[0], hc[1], hc[2], S.pix[px], S.pw[px]);
                S.u.h.err[px][table] = ((transMask >> px) & 1u) ? 0.0f : e;
                if (px == 0)
                    S.u.h.color[0][table] = (unsigned short)((hq[0] << 10) | (hq[1] << 5) | hq[2]);
            }
            // the premultiplier of the line colours walks the GROUP's range in steps of two, clamped to the block's own
            // range (ETC.cpp:1025-1040): the candidates depend on the group maximum of the line-pixel counts.
            // lane = (group member jb, table), as in the opaque T mode
            int nUnique = 0;
            {
                const int tbl = lane >> 3;
                const u32 baseM = call == 0 ? isoJ : (~isoJ & 0xffffu);
                const u32 lineM = ~baseM & ~transJ & 0xffffu;
                int lineTotalM[3];
                lineTotalsJ(lineM, lineTotalM);
                const int numLineM = __popc(lineM);
                const int clusterMaxLine = __builtin_amdgcn_readfirstlane(groupMax(numLineM));
                const bool mine = jb == own;
                const int modifierOffset = thDist(tbl) * 2;
                const int lineDivisor = numLineM * 34;
                const int lineAddend = (numLineM << 4) | numLineM;
                int n = 0, last = -1;
                for (int k = -clusterMaxLine; k <= clusterMaxLine; k += 2)
                {
                    int kc = k < numLineM ? k : numLineM;
                    kc = kc > -numLineM ? kc : -numLineM;
                    const int modifierAddend = (int)(short)(kc * modifierOffset);
                    int q[3], targets[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        int numerator = (int)(short)((int)(short)(lineTotalM[ch] + lineTotalM[ch] + (FAKE ? 0 : lineAddend)) + modifierAddend);
                        numerator = numerator < 0 ? 0 : numerator;
                        const int divided = lineDivisor == 0 ? 0 : udivSmall(numerator, lineDivisor);
                        q[ch] = divided < 15 ? divided : 15;
                        targets[ch] = numerator;
                    }
                    if (FAKE)
                        resolveTHFake(q, targets, numLineM);
                    const int packed = (q[0] << 10) | (q[1] << 5) | q[2];
                    if (n == 0 || packed != last)
                    {
                        if (mine)
                            S.tColors[tbl][n] = (unsigned short)packed;
                        n++;
                        last = packed;
                    }
                }
                if (mine)
                    S.tCount[tbl] = n;
                nUnique = groupMax(n);
            }
            WAVE_SYNC();
            int prefix[9];
            prefix[0] = 0;
#pragma unroll
            for (int t = 0; t < 8; t++)
                prefix[t + 1] = prefix[t] + laneI(nUnique, t * 8);

            // candidate `id` = (table, ci) in the reference's order; own colours, then (hazard H2) one zero slot, then
            // copies of colour 0
            auto candidateOf = [&](int id, int &table) -> int {
                table = 0;
#pragma unroll
                for (int t = 1; t < 8; t++)
                    if (id >= prefix[t])
                        table = t;
                const int ci = id - prefix[table];
                const int n = S.tCount[table];
                return ci < n ? S.tColors[table][ci] : (ci == n ? 0 : S.tColors[table][0]);
            };
            float candErr = FLT_MAX;
            int candId = 0x7fffffff;
            bool candH = false;
            for (int idBase = 0; idBase < prefix[8]; idBase += 64)
            {
                const int id = idBase + lane;
                if (id < prefix[8])
                {
                    int table;
                    const int packed = candidateOf(id, table);
                    const int modifier = thDist(table);
                    int lc[2][3];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const int q = (packed >> (10 - ch * 5)) & 15;
                        const int u = (q << 4) | q;
                        lc[0][ch] = u + modifier < 255 ? u + modifier : 255;
                        lc[1][ch] = u - modifier > 0 ? u - modifier : 0;
                    }
                    float tErr = 0.0f, hErr = 0.0f;
                    for (int px = 0; px < 16; px++)
                    {
                        const float e0 = E.wu(lc[0][0], lc[0][1], lc[0][2], S.pix[px], S.pw[px]);
                        const float e1 = E.wu(lc[1][0], lc[1][1], lc[1][2], S.pix[px], S.pw[px]);
                        const float le = ((transMask >> px) & 1u) ? 0.0f : sseMin(e0, e1);
                        tErr = tErr + sseMin(le, S.isoErr[px]);
                        hErr = hErr + sseMin(le, S.u.h.err[px][table]);
                    }
                    // H mode stores the order of its two colours in the low table bit and cannot swap them here
                    const bool hLegal = (packed < (int)S.u.h.color[0][table]) == ((table & 1) == 0);
                    const bool useH = (hErr < tErr) && hLegal;
                    const float roundBest = useH ? hErr : tErr;
                    if (roundBest < candErr)
                    {
                        candErr = roundBest;
                        candId = id;
                        candH = useH;
                    }
                }
            }
            float wErr = candErr;
            int wId = candId;
            waveArgmin(wErr, wId);
            if (wErr < bestError)
            {
                bestError = wErr;
                const bool useH = laneI((int)candH, wId & 63) != 0;
                int table;
                const int packed = candidateOf(wId, table);
                const int modifier = thDist(table);
                const int packedH2 = (int)S.u.h.color[0][table];
                int lc[2][3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    const int q = (packed >> (10 - ch * 5)) & 15;
                    const int u = (q << 4) | q;
                    lc[0][ch] = u + modifier < 255 ? u + modifier : 255;
                    lc[1][ch] = u - modifier > 0 ? u - modifier : 0;
                }
                u32 selectors = 0;
                for (int px = 0; px < 16; px++)
                {
                    const float e0 = E.wu(lc[0][0], lc[0][1], lc[0][2], S.pix[px], S.pw[px]);
                    const float e1 = E.wu(lc[1][0], lc[1][1], lc[1][2], S.pix[px], S.pw[px]);
                    const bool tr = ((transMask >> px) & 1u) != 0;
                    const float le = tr ? 0.0f : sseMin(e0, e1);
                    u32 sel = (e0 <= e1) ? 1u : 3u;
                    if ((useH ? S.u.h.err[px][table] : S.isoErr[px]) < le)
                        sel = 0u;
                    if (tr)
                        sel = 2u;
                    selectors |= sel << (px * 2);
                }
                if (useH)
                {
                    // T: C1, C2+M, transparent, C2-M  ->  H: C1+M, C1-M, transparent, C2-M (ETC.cpp:1232-1251)
                    u32 sectorBits = 0, signBits = 0;
#pragma unroll
                    for (int px = 0; px < 16; px++)
                    {
                        const u32 sel = (selectors >> (px * 2)) & 3u;
                        sectorBits |= ((sel & 1u) ^ 1u) << px;             // {1, 0, 1, 0}
                        signBits |= ((0x9u >> sel) & 1u) << px;            // {1, 0, 0, 1}
                    }
                    emitH(outHi, outLo, packed, packedH2, sectorBits, signBits, table, false);
                }
                else
                {
                    const int lineColor[3] = {(packed >> 10) & 15, (packed >> 5) & 15, packed & 15};
                    emitT(outHi, outLo, lineColor, isoQ, selectors, table, false);
                }
            }
            WAVE_SYNC();
        }
        clusterFit(true);
    }

    DBG_TAP(4);
    if (lane == 0)
    {
        if (MODE == 0 && A.outStride == 16u)
        {
            // [EAC alpha | colour]: one 16-byte store per block.  (Built and measured in round 5: two waves per workgroup that
            // file their results in LDS, the last finisher storing both blocks as one 32-byte sector -- 27.1 instead of 25.8 ms
            // at 4096^2: the per-wave LDS base costs registers the kernel does not have at five waves per SIMD.  What WRITE_SIZE
            // shows above the 16 MB of payload is the kernel's scratch, not the store width.)
            uint4 o;
            o.x = alphaW0;
            o.y = alphaW1;
            o.z = bswap32(outHi);
            o.w = bswap32(outLo);
            *reinterpret_cast<uint4 *>(out + (size_t)blockIndex * 16u) = o;
        }
        else
        {
            uint2 o;
            o.x = bswap32(outHi);
            o.y = bswap32(outLo);
            *reinterpret_cast<uint2 *>(out + (size_t)blockIndex * A.outStride + A.outOffset) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// EAC 8-bit alpha: CompressETC2AlphaBlockInternal (ETC.cpp:1902-2085), QuantizeETC2Alpha 2366-2411.
// KIND 0: 8-bit alpha of PixelBlockU8 (EncodeETC2Alpha / the alpha half of EncodeETC2RGBA);
// KIND 1 / 2: unsigned / signed 11-bit EAC of PixelBlockScalarS16 (EncodeETC2Alpha11, CompressEACBlock ETC.cpp:2087-2114)
// SPREAD = false: one lane per block (the throughput form: a wave's 64 blocks share every table word).
// SPREAD = true: sixteen lanes per block, lane t searches modifier table t, and a 16-lane minimum of (error, table) picks the
// winner -- the reference walks the tables in ascending order and keeps the first minimum, i.e. the lowest table among equal
// errors.  The same instructions per block, but a sixteenth of the latency: what a call with a handful of blocks (the
// reference's 8-block convention, ConvectionKernels_API.cpp:246-256, 270-286) waits for (213 -> 25 us at 16 blocks).
template <int KIND, bool SPREAD>
__global__ __launch_bounds__(64) void cvttmi_eac_alpha_kernel(const uint8_t *__restrict__ blocks, uint8_t *__restrict__ out,
                                                             const CvttEtcArgs A, const CvttDeviceTables *__restrict__ T)
{
    constexpr bool is11 = KIND != 0, isSigned = KIND == 2;
    const u32 blockIndex = SPREAD ? blockIdx.x * 4u + (threadIdx.x >> 4) : blockIdx.x * 64u + threadIdx.x;
    const bool valid = blockIndex < A.numBlocks;
    int pixel[16];
    if (!is11)
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(blocks + (size_t)(valid ? blockIndex : 0u) * 64u);
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const uint4 v = src[i];
            pixel[4 * i + 0] = (int)(v.x >> 24);
            pixel[4 * i + 1] = (int)(v.y >> 24);
            pixel[4 * i + 2] = (int)(v.z >> 24);
            pixel[4 * i + 3] = (int)(v.w >> 24);
        }
    }
    else
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(blocks + (size_t)(valid ? blockIndex : 0u) * 32u);
#pragma unroll
        for (int i = 0; i < 2; i++)
        {
            const uint4 v = src[i];
            const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 8; j++)
            {
                int x = (int)(short)((w[j >> 1] >> (16 * (j & 1))) & 0xffffu);
                // shifted ranges: signed 1..2047, unsigned 0..2047
                if (isSigned)
                {
                    x = (x > 1023 ? 1023 : x) + 1024;
                    x = x < 1 ? 1 : x;
                }
                else
                    x = x > 2047 ? 2047 : (x < 0 ? 0 : x);
                pixel[8 * i + j] = x;
            }
        }
    }
    int minAlpha = is11 ? 2047 : 255, maxAlpha = 0;
#pragma unroll
    for (int px = 0; px < 16; px++)
    {
        minAlpha = pixel[px] < minAlpha ? pixel[px] : minAlpha;
        maxAlpha = pixel[px] > maxAlpha ? pixel[px] : maxAlpha;
    }
    const int alphaSpan = maxAlpha - minAlpha;
    const int midTimes2 = maxAlpha + minAlpha;

    u32 bestTotalError = 0x7fffffffu;
    int bestTable = 0, bestBase = 0, bestMultiplier = 0;
    u32 bestIdxLo = 0, bestIdxHi = 0; // 3 bits per pixel

    for (int tableIter = 0; tableIter < (SPREAD ? 1 : 16); tableIter++)
    {
        const int tableIndex = SPREAD ? (int)(threadIdx.x & 15u) : tableIter;
        const int pos[4] = {T->eacPositive[tableIndex][0], T->eacPositive[tableIndex][1], T->eacPositive[tableIndex][2], T->eacPositive[tableIndex][3]};
        // the table's row of the rounding table (13 entries of 2 bits) and its four positive modifiers as two wave-uniform
        // words: a pixel's lookups are bit-field extracts instead of a per-lane byte load and a select chain
        u32 roundBits = 0;
        for (int i = 0; i < 13; i++)
            roundBits |= (u32)T->eacRounding[tableIndex][i] << (2 * i);
        u32 posWord = (u32)(pos[0] | (pos[1] << 8) | (pos[2] << 16) | (pos[3] << 24));
        if (!SPREAD)
        {
            roundBits = (u32)__builtin_amdgcn_readfirstlane((int)roundBits);
            posWord = (u32)__builtin_amdgcn_readfirstlane((int)posWord);
        }
        for (int r = 0; r < 10; r++)
        {
            const int subrange = r % 3, mainRange = r / 3;
            const int maxOffset = T->eacPositive[tableIndex][3 - mainRange - (subrange & 1)];
            const int minOffset = -(int)T->eacPositive[tableIndex][3 - mainRange - ((subrange >> 1) & 1)] - 1;
            const int offsetSpan = maxOffset - minOffset;
            int minMultiplier = udivSmall(alphaSpan, offsetSpan);
            if (is11)
            {
                minMultiplier = minMultiplier > 112 ? 112 : minMultiplier;
                minMultiplier &= 120;
            }
            else
            {
                minMultiplier = minMultiplier > 14 ? 14 : minMultiplier;
                minMultiplier = minMultiplier < 1 ? 1 : minMultiplier;
            }
            for (int mo = 0; mo < 2; mo++)
            {
                int multiplier = minMultiplier;
                if (is11)
                {
                    if (mo == 1)
                        multiplier += 8;
                    else
                        multiplier = multiplier < 1 ? 1 : multiplier;
                }
                else
                    multiplier += mo;
                int base2 = midTimes2 - multiplier * maxOffset - multiplier * minOffset; // all lanes stay inside 16 bits
                int baseAlpha;
                if (is11)
                {
                    if (isSigned)
                        base2 += 8;
                    const int lo2 = isSigned ? 16 : 0;
                    base2 = base2 < lo2 ? lo2 : (base2 > 4095 ? 4095 : base2);
                    baseAlpha = (base2 >> 1) & 2040;
                    if (!isSigned)
                        baseAlpha += 4;
                }
                else
                {
                    base2 = base2 < 0 ? 0 : (base2 > 510 ? 510 : base2);
                    baseAlpha = (base2 + 1) >> 1;
                }
                // lookup / multiplier for lookup < 2^12, multiplier <= 128: (lookup * ceil(2^20 / multiplier)) >> 20 is exact
                // (the excess of the magic number adds less than lookup / 2^20 < 1/256 to a quotient whose fraction is at most
                // 127/128), one 24-bit multiply and a shift per pixel; the magic number is per candidate
                const u32 magic = (u32)udivSmall20(multiplier);
                u32 totalError = 0; // (only the error: the winner's indexes are worked out again after the search)
#pragma unroll
                for (int px = 0; px < 16; px++)
                {
                    const int a = pixel[px];
                    const int refl2 = (a - baseAlpha) * 2 + multiplier;
                    const int absv = refl2 < 0 ? -refl2 : refl2;
                    const int lookup = absv >> 1;
                    int li = (int)(__umul24((u32)lookup, magic) >> 20);
                    li = li >= 13 ? 12 : li;
                    const int index = (int)((roundBits >> (2 * li)) & 3u);
                    const int pOff = (int)((posWord >> (8 * index)) & 0xffu);
                    const int sign = refl2 < 0 ? -1 : 0;
                    const int quantizedOffset = (pOff ^ sign) * multiplier;
                    int q = baseAlpha + quantizedOffset;
                    const int qLo = (is11 && isSigned) ? 1 : 0, qHi = is11 ? 2047 : 255;
                    q = q < qLo ? qLo : (q > qHi ? qHi : q);
                    const int d = q - a;
                    totalError += (u32)(d * d);
                }
                if (totalError < bestTotalError)
                {
                    bestTotalError = totalError;
                    bestTable = tableIndex;
                    bestBase = baseAlpha;
                    bestMultiplier = multiplier;
                }
            }
        }
    }
    if (SPREAD)
    {
        // (error, table) as one word: the error of 16 pixels stays below 2^27 (16 x 2047^2)
        u32 key = (bestTotalError << 4) | (u32)bestTable;
#pragma unroll
        for (int step = 1; step < 16; step <<= 1)
        {
            const u32 o = (u32)__shfl_xor((int)key, step);
            key = o < key ? o : key;
        }
        const int src = (int)((threadIdx.x & ~15u) | (key & 15u));
        bestTable = (int)(key & 15u);
        bestBase = __shfl(bestBase, src);
        bestMultiplier = __shfl(bestMultiplier, src);
    }
    // the winner's indexes (the operations of the candidate loop, once)
    {
        const u32 magic = (u32)udivSmall20(bestMultiplier);
#pragma unroll
        for (int px = 0; px < 16; px++)
        {
            const int refl2 = (pixel[px] - bestBase) * 2 + bestMultiplier;
            const int absv = refl2 < 0 ? -refl2 : refl2;
            int li = (int)(__umul24((u32)(absv >> 1), magic) >> 20);
            li = li >= 13 ? 12 : li;
            const int index = T->eacRounding[bestTable][li];
            const u32 code = (u32)(index + 4 - ((refl2 < 0 ? -1 : 0) & 4));
            if (px < 8)
                bestIdxLo |= code << (3 * px);
            else
                bestIdxHi |= code << (3 * (px - 8));
        }
    }
    if (is11)
    {
        bestMultiplier >>= 3;
        if (isSigned)
            bestBase ^= 0x80;
    }
    if (valid && (!SPREAD || (threadIdx.x & 15u) == 0))
    {
        // 16 x 3-bit indexes, column-major pixel order, MSB first (ETC.cpp:2049-2084)
        u64 bits = 0;
#pragma unroll
        for (int s = 0; s < 16; s++)
        {
            const int px = ((s & 3) << 2) | (s >> 2); // s-th emitted = pixel with selectorOrder[px] == s
            const u32 code = px < 8 ? (bestIdxLo >> (3 * px)) & 7u : (bestIdxHi >> (3 * (px - 8))) & 7u;
            bits = (bits << 3) | code;
        }
        uint8_t *o = out + (size_t)blockIndex * A.outStride + A.outOffset;
        const u32 w0 = ((u32)bestBase & 0xffu) | ((u32)(((bestMultiplier << 4) | bestTable) & 0xff) << 8) | ((u32)((bits >> 40) & 0xff) << 16) | ((u32)((bits >> 32) & 0xff) << 24);
        const u32 w1 = bswap32((u32)bits);
        uint2 v;
        v.x = w0;
        v.y = w1;
        *reinterpret_cast<uint2 *>(o) = v;
    }
}

// Launches of at most this many blocks use the sixteen-lanes-per-block form of the EAC search: the chip is not full there
// (256 CUs x 4 SIMDs x 4 waves x 64 lanes = 262 144 blocks resident in the one-lane form), so latency is what counts.
// CVTTMI_EAC_SPREAD_MAX in the environment overrides it (developer knob for A/B runs).
static const uint32_t kEacSpreadMax = getenv("CVTTMI_EAC_SPREAD_MAX") ? (uint32_t)atol(getenv("CVTTMI_EAC_SPREAD_MAX")) : 65536u;

extern "C" hipError_t cvttmi_launch_etc2(const void *d_blocks, void *d_out, const CvttEtcArgs *args,
                                         const CvttDeviceTables *d_tables, int mode, hipStream_t stream)
{
    // mode 0: RGB (8 B), 1: RGBA = [alpha | colour] (16 B), 2: alpha only (8 B), 3: ETC1 (8 B), 4: punch-through alpha (8 B)
    if (args->numBlocks == 0)
        return hipSuccess;
    CvttEtcArgs a = *args;
    a.outStride = (mode == 1) ? 16u : 8u;
    const bool fake = (a.flags & CVTTMI_FLAG_ETC_USE_FAKE_BT709) != 0;
    const uint32_t colourGrid = ((a.numBlocks / 8u + 7u) / 8u) * 64u; // eight XCDs x whole groups (see the kernel's block map)
#define CVTT_LAUNCH_COLOR(M)                                                                                                      \
    do                                                                                                                            \
    {                                                                                                                             \
        if (fake)                                                                                                                 \
            hipLaunchKernelGGL((cvttmi_etc2_color_kernel<M, true>), dim3(colourGrid), dim3(64), 0, stream,                 \
                               (const uint8_t *)d_blocks, (uint8_t *)d_out, a, d_tables);                                         \
        else                                                                                                                      \
            hipLaunchKernelGGL((cvttmi_etc2_color_kernel<M, false>), dim3(colourGrid), dim3(64), 0, stream,                 \
                               (const uint8_t *)d_blocks, (uint8_t *)d_out, a, d_tables);                                         \
    } while (0)
    if (mode == 3 || mode == 4)
    {
        a.outOffset = 0u;
        if (mode == 3)
            CVTT_LAUNCH_COLOR(1);
        else
            CVTT_LAUNCH_COLOR(2);
        return hipGetLastError();
    }
    if (mode != 2)
    {
        a.outOffset = (mode == 1) ? 8u : 0u;
        CVTT_LAUNCH_COLOR(0);
    }
#undef CVTT_LAUNCH_COLOR
    if (mode == 2) // EncodeETC2Alpha alone (RGBA: the colour kernel searches the alpha half itself and stores the whole block)
    {
        a.outOffset = 0u;
        if (a.numBlocks <= kEacSpreadMax)
            hipLaunchKernelGGL((cvttmi_eac_alpha_kernel<0, true>), dim3((a.numBlocks + 3u) / 4u), dim3(64), 0, stream, (const uint8_t *)d_blocks,
                               (uint8_t *)d_out, a, d_tables);
        else
            hipLaunchKernelGGL((cvttmi_eac_alpha_kernel<0, false>), dim3((a.numBlocks + 63u) / 64u), dim3(64), 0, stream, (const uint8_t *)d_blocks,
                               (uint8_t *)d_out, a, d_tables);
    }
    return hipGetLastError();
}

extern "C" hipError_t cvttmi_launch_eac11(const void *d_blocksS16, void *d_out, uint32_t numBlocks, int isSigned,
                                          const CvttDeviceTables *d_tables, hipStream_t stream)
{
    if (numBlocks == 0)
        return hipSuccess;
    CvttEtcArgs a = {};
    a.numBlocks = numBlocks;
    a.outStride = 8u;
    a.outOffset = 0u;
    const bool spread = numBlocks <= kEacSpreadMax;
    const dim3 grid(spread ? (numBlocks + 3u) / 4u : (numBlocks + 63u) / 64u);
    if (isSigned)
    {
        if (spread) hipLaunchKernelGGL((cvttmi_eac_alpha_kernel<2, true>), grid, dim3(64), 0, stream, (const uint8_t *)d_blocksS16, (uint8_t *)d_out, a, d_tables);
        else hipLaunchKernelGGL((cvttmi_eac_alpha_kernel<2, false>), grid, dim3(64), 0, stream, (const uint8_t *)d_blocksS16, (uint8_t *)d_out, a, d_tables);
    }
    else
    {
        if (spread) hipLaunchKernelGGL((cvttmi_eac_alpha_kernel<1, true>), grid, dim3(64), 0, stream, (const uint8_t *)d_blocksS16, (uint8_t *)d_out, a, d_tables);
        else hipLaunchKernelGGL((cvttmi_eac_alpha_kernel<1, false>), grid, dim3(64), 0, stream, (const uint8_t *)d_blocksS16, (uint8_t *)d_out, a, d_tables);
    }
    return hipGetLastError();
}
