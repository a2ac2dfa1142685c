// Device helpers shared by the encoder kernels (lane arithmetic of the reference's SSE2 path,
// PCA seed search, branch-and-bound bound).  Included by the .hip files only.
#ifndef CVTTMI_KERNEL_COMMON_H
#define CVTTMI_KERNEL_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "cvtt_device.h"

namespace
{
typedef uint32_t u32;
typedef uint64_t u64;
// Two-element float values.  Deliberately NOT v_pk_add_f32 / v_pk_mul_f32: on gfx950 a packed instruction occupies both
// halves of the SIMD's float datapath for its 4 cycles, while two plain v_mul_f32 / v_add_f32 / v_sub_f32 take the same
// 2 x ~2.3 cycles AND can each be overlapped with an instruction of another wave (conversions, min/max, v_perm, compares
// ... see profiles/r02/valu_peak.json: the pair matrix).  Measured on BASELINE config 2: scalar + -fno-slp-vectorize
// (the Makefile default; LLVM's SLP pass otherwise re-packs) 542 vs 515 Mblocks/s.  -DCVTT_PACKED_F32 restores the packed
// form for A/B runs.
#ifdef CVTT_PACKED_F32
typedef float v2f __attribute__((ext_vector_type(2)));
#else
struct v2f
{
    float x, y;
};
__device__ __forceinline__ v2f operator+(v2f a, v2f b) { return v2f{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ v2f operator-(v2f a, v2f b) { return v2f{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ v2f operator*(v2f a, v2f b) { return v2f{a.x * b.x, a.y * b.y}; }
#endif

// The value lane (l ^ step) holds, step a power of two known at compile time (after unrolling): ds_swizzle_b32 in bit mode inside
// the 32-lane halves -- no address register and a third of the LDS-crossbar time of the ds_bpermute_b32 that __shfl_xor becomes,
// whose lane arithmetic (amd_warp_functions.h) the compiler also keeps alive across whole kernels -- and ds_bpermute_b32 across
// the halves.  One workgroup = one wave in every kernel that uses it (threadIdx.x is the lane).  Round 5, ETC2: +4.5 %.
__device__ __forceinline__ int xorLane(int v, int step)
{
#ifdef CVTT_XORLANE_BPERMUTE // A/B: the __shfl_xor form
    return __shfl_xor(v, step);
#endif
    switch (step)
    {
    case 1: return __builtin_amdgcn_ds_swizzle(v, (1 << 10) | 0x1f);
    case 2: return __builtin_amdgcn_ds_swizzle(v, (2 << 10) | 0x1f);
    case 4: return __builtin_amdgcn_ds_swizzle(v, (4 << 10) | 0x1f);
    case 8: return __builtin_amdgcn_ds_swizzle(v, (8 << 10) | 0x1f);
    case 16: return __builtin_amdgcn_ds_swizzle(v, (16 << 10) | 0x1f);
    default: return __builtin_amdgcn_ds_bpermute((int)((((unsigned)threadIdx.x & 63u) ^ (unsigned)step) << 2), v);
    }
}
__device__ __forceinline__ unsigned xorLane(unsigned v, int step) { return (unsigned)xorLane((int)v, step); }
__device__ __forceinline__ float xorLane(float v, int step) { return __int_as_float(xorLane(__float_as_int(v), step)); }

// ---- lane arithmetic helpers -------------------------------------------------------
// MINPS/MAXPS operand order (reference ParallelMath.h:522-559): second operand wins on
// NaN / equal.
__device__ __forceinline__ float sseMin(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float sseMax(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float safeDenom(float v) { return v == 0.0f ? 1.0f : v; }
// SQRTPS: correctly rounded.  NOT __fsqrt_rn -- in this toolchain that is the 1-ulp native
// v_sqrt_f32 (__clang_hip_math.h); __builtin_sqrtf under -fhip-fp32-correctly-rounded-divide-sqrt
// expands to the exact sequence.  (One block in two million lands on a rounding boundary
// where the difference shows: tests/golden/regress_bc7_group_683856.npy.)
__device__ __forceinline__ float sqrtExact(float v) { return __builtin_sqrtf(v); }

// Clamp then CVTPS2DQ under round-to-nearest-even (reference ParallelMath.h:561-567,
// 936-946).  fminf/fmaxf match MINPS/MAXPS here: a NaN input yields `hi`, and the sign of
// a zero result is irrelevant once converted to an integer.  Result stays a float
// (integral value) so callers can use it as both index and refiner weight.
__device__ __forceinline__ float clampRound(float v, float hi)
{
    return rintf(fmaxf(fminf(v, hi), 0.0f));
}

__device__ __forceinline__ float byteF(u32 pk, int ch) { return (float)((pk >> (8 * ch)) & 0xffu); }
__device__ __forceinline__ int byteI(u32 pk, int ch) { return (int)((pk >> (8 * ch)) & 0xffu); }

// Fetch a packed pixel through an empty asm so the optimiser treats it as a fresh value:
// without it LICM hoists all 64 byte->float conversions (and their weighted products) out of
// the trial loops and keeps ~190 extra VGPRs alive for the whole kernel.
__device__ __forceinline__ u32 fetchPixel(u32 pk)
{
    asm volatile("" : "+v"(pk));
    return pk;
}

// Same trick for wave-uniform values (shape masks): keeps the 16 per-pixel bit tests as
// s_bitcmp on one SGPR instead of 16 hoisted 64-bit lane masks.
__device__ __forceinline__ u32 opaqueUniform(u32 v)
{
    asm volatile("" : "+s"(v));
    return v;
}

// 24-bit integer multiply-add: full-rate v_mad_i32_i24 (operands here are < 2^16)
__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }

// Forced instruction selection for the reconstruct-and-compare step (the optimiser otherwise
// rewrites it into 64-bit multiply-adds and mask/shift sequences):
//   sum = w * delta4 + base4            one v_mad_i32_i24; the reconstructed channel is byte 1
//   d   = byte1(sum) - byteCH(pixel)    one SDWA subtract
//   acc = d * d + acc                   one v_mad_i32_i24
__device__ __forceinline__ int madI24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int CH>
__device__ __forceinline__ int subByte1(int sum, u32 pk)
{
    int r;
    if (CH == 0)
        asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0" : "=v"(r) : "v"(sum), "v"(pk));
    else if (CH == 1)
        asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(r) : "v"(sum), "v"(pk));
    else if (CH == 2)
        asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "=v"(r) : "v"(sum), "v"(pk));
    else
        asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_3" : "=v"(r) : "v"(sum), "v"(pk));
    return r;
}
// byte K of `recs` (a reconstructed channel value) minus channel CH of the pixel: one SDWA subtract
template <int K, int CH>
__device__ __forceinline__ int subByteK(u32 recs, u32 pk)
{
    int r;
#define CVTT_SUBK(k, c) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #k " src1_sel:BYTE_" #c : "=v"(r) : "v"(recs), "v"(pk))
    if (K == 0)
    {
        if (CH == 0) CVTT_SUBK(0, 0); else if (CH == 1) CVTT_SUBK(0, 1); else if (CH == 2) CVTT_SUBK(0, 2); else CVTT_SUBK(0, 3);
    }
    else if (K == 1)
    {
        if (CH == 0) CVTT_SUBK(1, 0); else if (CH == 1) CVTT_SUBK(1, 1); else if (CH == 2) CVTT_SUBK(1, 2); else CVTT_SUBK(1, 3);
    }
    else
    {
        if (CH == 0) CVTT_SUBK(2, 0); else if (CH == 1) CVTT_SUBK(2, 1); else if (CH == 2) CVTT_SUBK(2, 2); else CVTT_SUBK(2, 3);
    }
#undef CVTT_SUBK
    return r;
}
// squared error of channel CH for interpolation weight w, added to acc
template <int CH>
__device__ __forceinline__ u32 accumulateChannelError(int w, int delta4, int base4, u32 pk, u32 acc)
{
    const int d = subByte1<CH>(madI24(w, delta4, base4), pk);
    return (u32)madI24(d, d, (int)acc);
}
template <int CH>
__device__ __forceinline__ u32 channelError(int w, int delta4, int base4, u32 pk)
{
    const int d = subByte1<CH>(madI24(w, delta4, base4), pk);
    return (u32)__mul24(d, d);
}

struct Unfinished
{
    float base[4];
    float offset[4];
};

// ---- EndpointSelector<N,8> (reference EndpointSelector.h:33-149,
// PackedCovarianceMatrix.h:29-59), one lane, pixels of `mask` in ascending order.  Split in
// two so that the branch-and-bound test below can run between the passes. -----------------
template <int N>
struct Moments
{
    float centroid[N];
    float cov[N * (N + 1) / 2]; // lower triangle, row-major: (row,col) at row*(row+1)/2+col
};

// Pixel source for the PCA passes: pre-weighted channel values of pixel px.
struct FetchLDR
{
    const u32 (&pix)[16];
    const float (&w)[4];
    template <int N>
    __device__ __forceinline__ void get(int px, float (&v)[N]) const
    {
        const u32 pk = fetchPixel(pix[px]);
#pragma unroll
        for (int ch = 0; ch < N; ch++)
            v[ch] = byteF(pk, ch) * w[ch];
    }
};

// passes 0 and 1: centroid and scatter matrix of the pre-weighted pixels
// `sums` (optional): the plain sums of the pre-weighted pixels before the division, i.e. what EndpointRefiner::
// ContributeUnweightedPW accumulates as m_v over the same pixels in the same order
template <int N, class Fetch>
__device__ __forceinline__ void pcaMomentsT(const Fetch &F, u32 mask, Moments<N> &m, float *sums = nullptr)
{
#pragma unroll
    for (int ch = 0; ch < N; ch++)
        m.centroid[ch] = 0.0f;
    float count = 0.0f;
#pragma unroll
    for (int px = 0; px < 16; px++)
    {
        if ((mask >> px) & 1u)
        {
            float v[N];
            F.template get<N>(px, v);
#pragma unroll
            for (int ch = 0; ch < N; ch++)
                m.centroid[ch] = m.centroid[ch] + v[ch];
            count = count + 1.0f;
        }
    }
    if (sums)
    {
#pragma unroll
        for (int ch = 0; ch < N; ch++)
            sums[ch] = m.centroid[ch];
    }
    const float denom = safeDenom(count);
#pragma unroll
    for (int ch = 0; ch < N; ch++)
        m.centroid[ch] = m.centroid[ch] / denom;

#pragma unroll
    for (int i = 0; i < N * (N + 1) / 2; i++)
        m.cov[i] = 0.0f;
#pragma unroll
    for (int px = 0; px < 16; px++)
    {
        if ((mask >> px) & 1u)
        {
            float v[N];
            F.template get<N>(px, v);
            float diff[N];
#pragma unroll
            for (int ch = 0; ch < N; ch++)
                diff[ch] = v[ch] - m.centroid[ch];
            int index = 0;
#pragma unroll
            for (int row = 0; row < N; row++)
#pragma unroll
                for (int col = 0; col <= row; col++)
                {
                    m.cov[index] = m.cov[index] + diff[row] * diff[col];
                    index++;
                }
        }
    }
}

template <int N>
__device__ __forceinline__ void pcaMoments(const u32 (&pix)[16], u32 mask, const float (&w)[4], Moments<N> &m)
{
    const FetchLDR F = {pix, w};
    pcaMomentsT<N>(F, mask, m);
}

// power iteration, pass 2 and GetEndpoints
template <int N, class Fetch>
__device__ __forceinline__ void pcaFinishT(const Fetch &F, u32 mask, const float (&w)[4], const Moments<N> &m, Unfinished &u)
{
    float approx[N];
#pragma unroll
    for (int ch = 0; ch < N; ch++)
        approx[ch] = 1.0f;
    for (int it = 0; it < 8; it++)
    {
        float product[N];
#pragma unroll
        for (int row = 0; row < N; row++)
        {
            float sum = 0.0f;
#pragma unroll
            for (int col = 0; col < N; col++)
            {
                const int hi = row > col ? row : col;
                const int lo = row > col ? col : row;
                sum = sum + approx[col] * m.cov[hi * (hi + 1) / 2 + lo];
            }
            product[row] = sum;
        }
        float largest = product[0];
#pragma unroll
        for (int ch = 1; ch < N; ch++)
            largest = sseMax(largest, product[ch]);
        largest = safeDenom(largest);
#pragma unroll
        for (int ch = 0; ch < N; ch++)
            approx[ch] = product[ch] / largest;
    }
    float approxLen = 0.0f;
#pragma unroll
    for (int ch = 0; ch < N; ch++)
        approxLen = approxLen + approx[ch] * approx[ch];
    approxLen = safeDenom(sqrtExact(approxLen));
    float direction[N];
#pragma unroll
    for (int ch = 0; ch < N; ch++)
        direction[ch] = approx[ch] / approxLen;

    float minDist = FLT_MAX, maxDist = -FLT_MAX;
#pragma unroll
    for (int px = 0; px < 16; px++)
    {
        if ((mask >> px) & 1u)
        {
            float v[N];
            F.template get<N>(px, v);
            float dist = 0.0f;
#pragma unroll
            for (int ch = 0; ch < N; ch++)
                dist = dist + direction[ch] * (v[ch] - m.centroid[ch]);
            minDist = sseMin(minDist, dist);
            maxDist = sseMax(maxDist, dist);
        }
    }
#pragma unroll
    for (int ch = 0; ch < N; ch++)
    {
        const float mn = m.centroid[ch] + direction[ch] * minDist;
        const float mx = m.centroid[ch] + direction[ch] * maxDist;
        u.base[ch] = mn / w[ch];
        u.offset[ch] = (mx - mn) / w[ch];
    }
}

template <int N>
__device__ __forceinline__ void pcaFinish(const u32 (&pix)[16], u32 mask, const float (&w)[4], const Moments<N> &m,
                                          Unfinished &u)
{
    const FetchLDR F = {pix, w};
    pcaFinishT<N>(F, mask, w, m, u);
}

template <int N>
__device__ __forceinline__ void pcaEndpoints(const u32 (&pix)[16], u32 mask, const float (&w)[4], int, Unfinished &u)
{
    Moments<N> m;
    pcaMoments<N>(pix, mask, w, m);
    pcaFinish<N>(pix, mask, w, m, u);
}

// ---- exact branch-and-bound: a rigorous lower bound on the error of ANY trial of a shape --
// Every reconstructed colour of a trial is floor(I + 0.5) per channel for a point I on the
// segment between the two (quantised) endpoints (IndexSelector.h:90-100), so in the weighted
// metric it lies within delta = 0.5*sqrt(sum w_ch^2) of some line L.  Hence for every pixel
// the trial's error is >= max(0, d - delta)^2 >= d^2 - 2*delta*d, d = weighted distance to L,
// and with D2 = sum d^2 >= R (R = total-least-squares residual of the shape = trace(S) -
// lambda_max(S), S the scatter matrix that PCA pass 1 accumulates anyway) and sum d <=
// sqrt(n*D2):   error >= R - 2*delta*sqrt(n*R)   whenever R >= n*delta^2.
// lambda_max is over-estimated by trace(S^4)^(1/4), so R is under-estimated; the result is
// scaled down by 1e-4 to absorb the float rounding of S, of this computation and of the
// reference's own error sums.  A candidate whose bound exceeds the current best can never be
// committed (the commit needs error <= best), so skipping it leaves the output bit-identical.
// (rOut, optional: the under-estimated residual itself, or -1 when there is none)
template <int N>
__device__ __forceinline__ float shapeErrorLowerBound(const Moments<N> &m, float n, float delta, float *rOut = nullptr)
{
    float trace = 0.0f;
#pragma unroll
    for (int i = 0; i < N; i++)
        trace += m.cov[i * (i + 1) / 2 + i];
    // S2 = S*S (symmetric), trace(S^4) = ||S2||_F^2
    float t4 = 0.0f;
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
        {
            float e = 0.0f;
#pragma unroll
            for (int k = 0; k < N; k++)
            {
                const int a0 = i > k ? i : k, a1 = i > k ? k : i;
                const int b0 = j > k ? j : k, b1 = j > k ? k : j;
                e = __fmaf_rn(m.cov[a0 * (a0 + 1) / 2 + a1], m.cov[b0 * (b0 + 1) / 2 + b1], e);
            }
            t4 = __fmaf_rn((i == j ? 1.0f : 2.0f) * e, e, t4);
        }
    // v_sqrt_f32 is accurate to 1 ulp; the 1e-4 margins cover it ("Soundness of the bounds": the computed
    // lambdaUp exceeds lambda_max by at least 0.98e-4 * lambda_max, every rounding of S, of the trace and of this function
    // moves r by less than 2e-6 * lambda_max, and `delta` carries its own 1e-6 margin for the sqrt / products below)
    const float lambdaUp = __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(t4)) * 1.0001f;
    const float r = trace - lambdaUp;
    float lb = 0.0f;
    // t4 below 1e-30 (lambda_max < 2e-8: absurdly small channel weights) would be summed from denormal products: no bound
    if (rOut)
        *rOut = (t4 > 1e-30f) ? r : -1.0f;
    if (r > n * delta * delta && t4 > 1e-30f)
        lb = (r - 2.0f * delta * __builtin_amdgcn_sqrtf(n * r)) * 0.9999f;
    return lb > 0.0f ? lb : 0.0f; // NaN / inf inputs end up as "no bound"
}


// Order-preserving argmin helpers and 128-bit writer live with the kernels that use them.
} // namespace

#endif
