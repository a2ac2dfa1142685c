// Interpolated (BC3 alpha, BC4, BC5) and explicit (BC2) alpha blocks for gfx950 -- SURVEY.md 8f row 4.
//
// Replaces cvtt::Internal::S3TCComputer::PackInterpolatedAlpha (reference ConvectionKernels_S3TC.cpp:
// 343-715) and PackExplicitAlpha (306-341) as reached from cvtt::Kernels::EncodeBC2 / BC3 / BC4U / BC4S /
// BC5U / BC5S (ConvectionKernels_API.cpp:101-199).  One lane owns one block and one channel: the path has no
// cross-lane coupling (its AnySet / AllSet only skip duplicate work, and a weighted refiner contribution
// with weight 1.0 equals an unweighted one), every loop is wave-uniform.
#include "cvtt_kernel_common.h"

namespace
{
struct AlphaArgs
{
    uint32_t numBlocks;
    uint32_t inStride;   // bytes between input blocks (64)
    uint32_t channel;    // byte of the RGBA pixel to encode
    uint32_t outStride;  // bytes between output blocks
    uint32_t outOffset;
    uint32_t isSigned;   // input is PixelBlockS8: bias like Util::BiasSignedInput (Util.cpp:47-60)
    int32_t maxTweakRounds; // Options::seedPoints
    int32_t numRefineRounds; // Options::refineRoundsIIC
};

// one (end points -> indexes, error, refiner sums) evaluation; RANGE = 8 (full) or 6 (reduced + terminals)
template <int RANGE>
__device__ __forceinline__ float alphaTrial(const int (&pixel)[16], const int (&ep)[2], int highTerminal, bool feedRefiner,
                                            float rcpMaxIndex, u64 &idxOut, float &tv, float &vs, float &tt, float &ts, int &count)
{
    const float maxValue = (float)(RANGE - 1);
    const int weightRcp = (65536 + (RANGE - 1)) / (2 * (RANGE - 1)); // g_weightReciprocals, IndexSelector.cpp:43-62
    // IndexSelector<1>::Init, weight 1.0 (IndexSelector.h:27-77)
    const float origin = (float)ep[0];
    const float epDW = ((float)ep[1] - origin) * 1.0f;
    const float lenSq = safeDenom(epDW * epDW);
    const float axis = epDW * 1.0f * (maxValue / lenSq);
    const int recBase = (ep[0] << 8) + 128, recDelta = ep[1] - ep[0];
    tv = vs = tt = ts = 0.0f;
    count = 0;
    u64 idx = 0;
    u32 agg = 0;
    float error = 0.0f;
#pragma unroll
    for (int px = 0; px < 16; px++)
    {
        const float f = (float)pixel[px];
        const float fidx = clampRound((f - origin) * axis, maxValue);
        const int selected = (int)fidx;
        // ReconstructLDRPrecise (IndexSelector.h:102-112)
        const int wgt = mad24(weightRcp, selected, 64) >> 7;
        const int rec = mad24(wgt, recDelta, recBase) >> 8;
        const int d = rec - pixel[px];
        int index = selected;
        bool contribute = true;
        if (RANGE == 8)
            agg += (u32)(d * d);
        else
        {
            // the two reserved values 0 and highTerminal compete with the interpolated one (S3TC.cpp:583-611)
            const float zeroError = (float)(pixel[px] * pixel[px]);
            const int dh = highTerminal - pixel[px];
            const float highError = (float)(dh * dh);
            const float selectedError = (float)(d * d);
            float bestPixelError = zeroError;
            index = 6;
            if (highError < bestPixelError)
                index = 7;
            bestPixelError = sseMin(bestPixelError, highError);
            contribute = selectedError < bestPixelError;
            if (contribute)
                index = selected;
            bestPixelError = sseMin(bestPixelError, selectedError);
            error = error + bestPixelError;
        }
        if (feedRefiner && contribute)
        {
            // EndpointRefiner<1>::ContributeUnweightedPW (EndpointRefiner.h:78-92)
            const float t = fidx * rcpMaxIndex;
            tv = tv + t * f;
            vs = vs + f;
            tt = tt + t * t;
            ts = ts + t;
            count++;
        }
        idx |= (u64)(u32)index << (4 * px);
    }
    idxOut = idx;
    return (RANGE == 8) ? (float)(int)agg : error;
}

__global__ __launch_bounds__(64) void cvttmi_s3tc_alpha_kernel(const uint8_t *__restrict__ blocks, uint8_t *__restrict__ out,
                                                               const AlphaArgs A, const CvttDeviceTables *__restrict__ T)
{
    const u32 blockIndex = blockIdx.x * 64u + threadIdx.x;
    const bool valid = blockIndex < A.numBlocks;
    const int highTerminal = A.isSigned ? 254 : 255;
    int pixel[16], sorted[16];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(blocks + (size_t)(valid ? blockIndex : 0u) * A.inStride);
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const uint4 v = src[i];
            const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                int x = (int)((w[j] >> (8 * A.channel)) & 0xffu);
                if (A.isSigned)
                {
                    x = (int)(signed char)x;
                    x = (x < -127 ? -127 : x) + 127;  // BiasSignedInput
                    x = x > highTerminal ? highTerminal : x;
                }
                pixel[4 * i + j] = x;
                sorted[4 * i + j] = x;
            }
        }
    }
    // the reference's bubble network (S3TC.cpp:372-382)
#pragma unroll
    for (int sortEnd = 15; sortEnd > 0; sortEnd--)
#pragma unroll
        for (int o = 0; o < sortEnd; o++)
        {
            const int a = sorted[o], b = sorted[o + 1];
            sorted[o] = a < b ? a : b;
            sorted[o + 1] = a < b ? b : a;
        }

    const int maxTweak = A.maxTweakRounds < 1 ? 1 : A.maxTweakRounds;
    const int numRefine = A.numRefineRounds < 1 ? 1 : A.numRefineRounds;
    int tweakRounds = 4; // TweakRoundsForRange(8) = TweakRoundsForRange(6) = 4
    if (tweakRounds > maxTweak)
        tweakRounds = maxTweak;

    float bestError = FLT_MAX;
    int bestFull = 0, bestEP0 = 0, bestEP1 = 0;
    u64 bestIdx = 0;

    // candidate end-point pairs: the full range, then the four reduced-precision pairs (S3TC.cpp:470-563)
    int minEPs[2], maxEPs[2];
    {
        int heurMin = sorted[0], heurMax = sorted[15];
        const int largest = heurMax - heurMin;
        const int hiClear = highTerminal - heurMax;
        const int lowest = heurMin < hiClear ? heurMin : hiClear;
        // ParallelMath::LessOrEqual(UInt15) is '<' (ParallelMath.h:740-745)
        const bool canTryClipping = ((lowest << 2) + (lowest << 4)) < largest;
        if (__ballot(canTryClipping) != 0)
        {
            for (int firstIndex = 0; firstIndex < 16; firstIndex++)
            {
                int lowClearance = 0, sFirst = sorted[0];
#pragma unroll
                for (int i = 1; i < 16; i++)
                    if (i == firstIndex)
                    {
                        lowClearance = sorted[i - 1];
                        sFirst = sorted[i];
                    }
                for (int lastIndex = firstIndex; lastIndex < 16; lastIndex++)
                {
                    const int numSkippedHigh = 15 - lastIndex, numSkipped = firstIndex + numSkippedHigh;
                    if (!(0 < numSkipped)) // bestSkipCount stays 0 in the reference
                        continue;
                    int sLast = sorted[0], highClearance = 0;
#pragma unroll
                    for (int i = 0; i < 16; i++)
                    {
                        if (i == lastIndex)
                            sLast = sorted[i];
                        if (i == 16 - numSkippedHigh && numSkippedHigh > 0)
                            highClearance = highTerminal - sorted[i];
                    }
                    const int clearance = highClearance > lowClearance ? highClearance : lowClearance;
                    const bool better = canTryClipping && (((clearance << 2) + (clearance << 4)) < (sLast - sFirst));
                    if (better)
                    {
                        heurMin = sFirst;
                        heurMax = sLast;
                    }
                }
            }
        }
        int simpleMin = 1, simpleMax = highTerminal - 1;
#pragma unroll
        for (int px = 0; px < 16; px++)
        {
            if (0 < sorted[15 - px]) simpleMin = sorted[15 - px];
            if (sorted[px] < highTerminal) simpleMax = sorted[px];
        }
        minEPs[0] = simpleMin; minEPs[1] = heurMin;
        maxEPs[0] = simpleMax; maxEPs[1] = heurMax;
    }

    for (int cand = 0; cand < 5; cand++)
    {
        const bool full = cand == 0;
        float base, offset;
        if (full)
        {
            base = (float)sorted[0];
            offset = (float)(sorted[15] - sorted[0]);
        }
        else
        {
            const int mi = (cand - 1) >> 1, ma = (cand - 1) & 1;
            const int lo = mi ? minEPs[1] : minEPs[0], hi = ma ? maxEPs[1] : maxEPs[0];
            base = (float)lo;
            offset = (float)((hi - lo) & 0xffff);
        }
        const float rcpMaxIndex = full ? T->rcpMaxIndex[3] : 0.2f; // 1/7 (IEEE divide on the host), 1/5
        for (int tweak = 0; tweak < tweakRounds; tweak++)
        {
            // UnfinishedEndpoints<1>::FinishLDR(tweak, 8) in both precisions (S3TC.cpp:409, 576)
            int ep[2];
            ep[0] = (int)clampRound(base + offset * T->tweakFactors[1][tweak][0], 255.0f);
            ep[1] = (int)clampRound(base + offset * T->tweakFactors[1][tweak][1], 255.0f);
            for (int refinePass = 0; refinePass < numRefine; refinePass++)
            {
                if (A.isSigned)
                {
                    ep[0] = ep[0] > highTerminal ? highTerminal : ep[0];
                    ep[1] = ep[1] > highTerminal ? highTerminal : ep[1];
                }
                const bool feed = refinePass != numRefine - 1;
                u64 idx;
                float tv, vs, tt, ts;
                int count;
                const float error = full ? alphaTrial<8>(pixel, ep, highTerminal, feed, rcpMaxIndex, idx, tv, vs, tt, ts, count)
                                         : alphaTrial<6>(pixel, ep, highTerminal, feed, rcpMaxIndex, idx, tv, vs, tt, ts, count);
                if (error < bestError)
                {
                    bestError = error;
                    bestFull = full ? 1 : 0;
                    bestEP0 = ep[0];
                    bestEP1 = ep[1];
                    bestIdx = idx;
                }
                if (feed)
                {
                    // EndpointRefiner<1>::GetRefinedEndpointsLDR (EndpointRefiner.h:99-152); w = contributions, 0 -> 1
                    const int wi = count == 0 ? 1 : count;
                    const float w = (float)wi, wRcp = T->rcpTable[wi];
                    float adenom = (tt * w - ts * ts) * wRcp;
                    const bool z = adenom == 0.0f;
                    if (z)
                        adenom = 1.0f;
                    const float a = (tv - ts * vs * wRcp) / adenom;
                    const float b = (vs - a * ts) * wRcp;
                    float p1 = b, p2 = a + b;
                    if (z)
                    {
                        p1 = vs * wRcp;
                        p2 = p1;
                    }
                    ep[0] = (int)clampRound(p1 * 1.0f, 255.0f);
                    ep[1] = (int)clampRound(p2 * 1.0f, 255.0f);
                }
            }
        }
    }

    // S3TC.cpp:648-713
    int ep0 = bestEP0, ep1 = bestEP1;
    if (A.isSigned)
    {
        ep0 -= 127;
        ep1 -= 127;
    }
    const bool swapEndpoints = (bestFull != 0) != (ep0 > ep1);
    if (swapEndpoints)
    {
        const int t = ep0;
        ep0 = ep1;
        ep1 = t;
    }
    const int maxValue = bestFull ? 7 : 5;
    u64 bits = 0;
#pragma unroll
    for (int px = 0; px < 16; px++)
    {
        int index = (int)((bestIdx >> (4 * px)) & 0xfull);
        if (swapEndpoints && index <= maxValue)
            index = maxValue - index;
        if (index != 0)
        {
            if (index == maxValue)
                index = 1;
            else if (index < maxValue)
                index++;
        }
        bits |= (u64)(u32)index << (3 * px);
    }
    if (valid)
    {
        uint2 o;
        o.x = ((u32)ep0 & 0xffu) | (((u32)ep1 & 0xffu) << 8) | ((u32)(bits & 0xffffull) << 16);
        o.y = (u32)(bits >> 16);
        *reinterpret_cast<uint2 *>(out + (size_t)blockIndex * A.outStride + A.outOffset) = o;
    }
}

// PackExplicitAlpha (BC2): 4-bit alpha = IndexSelector<1> over [0, 255] with 16 levels
__global__ __launch_bounds__(64) void cvttmi_s3tc_explicit_alpha_kernel(const uint8_t *__restrict__ blocks, uint8_t *__restrict__ out,
                                                                        const AlphaArgs A)
{
    const u32 blockIndex = blockIdx.x * 64u + threadIdx.x;
    if (blockIndex >= A.numBlocks)
        return;
    const uint4 *src = reinterpret_cast<const uint4 *>(blocks + (size_t)blockIndex * A.inStride);
    // Init: origin 0, axis = 255 * 1 * (15 / 255^2)
    const float epDW = (255.0f - 0.0f) * 1.0f;
    const float axis = epDW * 1.0f * (15.0f / (epDW * epDW));
    u64 bits = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const uint4 v = src[i];
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const float f = (float)((w[j] >> (8 * A.channel)) & 0xffu);
            const u32 index = (u32)clampRound((f - 0.0f) * axis, 15.0f);
            bits |= (u64)index << (4 * (4 * i + j));
        }
    }
    uint2 o;
    o.x = (u32)bits;
    o.y = (u32)(bits >> 32);
    *reinterpret_cast<uint2 *>(out + (size_t)blockIndex * A.outStride + A.outOffset) = o;
}
} // namespace

extern "C" hipError_t cvttmi_launch_s3tc_alpha(const void *d_blocks, void *d_out, uint32_t numBlocks, uint32_t channel, uint32_t outStride,
                                               uint32_t outOffset, int isSigned, int explicitAlpha, int seedPoints, int refineRounds,
                                               const CvttDeviceTables *d_tables, hipStream_t stream)
{
    if (numBlocks == 0)
        return hipSuccess;
    AlphaArgs a;
    a.numBlocks = numBlocks;
    a.inStride = 64u;
    a.channel = channel;
    a.outStride = outStride;
    a.outOffset = outOffset;
    a.isSigned = isSigned ? 1u : 0u;
    a.maxTweakRounds = seedPoints;
    a.numRefineRounds = refineRounds;
    const dim3 grid((numBlocks + 63u) / 64u), block(64);
    if (explicitAlpha)
        hipLaunchKernelGGL(cvttmi_s3tc_explicit_alpha_kernel, grid, block, 0, stream, (const uint8_t *)d_blocks, (uint8_t *)d_out, a);
    else
        hipLaunchKernelGGL(cvttmi_s3tc_alpha_kernel, grid, block, 0, stream, (const uint8_t *)d_blocks, (uint8_t *)d_out, a, d_tables);
    return hipGetLastError();
}
