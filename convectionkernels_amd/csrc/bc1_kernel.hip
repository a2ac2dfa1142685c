// BC1 (DXT1, alpha-test variant) endpoint search for gfx950; also the colour half of BC2 / BC3 (alphaTest = 0).
//
// Replaces cvtt::Internal::S3TCComputer::PackRGB(alphaTest = true) as reached from
// cvtt::Kernels::EncodeBC1 (reference ConvectionKernels_API.cpp:86-99,
// ConvectionKernels_S3TC.cpp:717-1052; TestEndpoints 190-258, QuantizeTo565 52-69,
// ParanoidDiff 71-81), non-exhaustive search.  Bit-identical to the reference's SSE2 lanes.
//
// Mapping: the search per block is small (<= 7 seed chains x refine rounds x 16 pixels), so one
// LANE owns one block: a wave encodes 64 consecutive blocks (8 reference groups; the
// non-exhaustive path has no cross-lane coupling), every loop is wave-uniform
// and the reference's sequential strict-'<' commit order is simply program order.
#include "cvtt_kernel_common.h"

namespace
{
__device__ __forceinline__ int quantize5(int v)
{
    const int r = (__mul24(v, 249) + 1024) >> 11; // S3TC.cpp:58-62, fits 16 bits
    return (r << 3) | (r >> 2);
}
__device__ __forceinline__ int quantize6(int v)
{
    const int r = (__mul24(v, 253) + 512) >> 10; // S3TC.cpp:52-56
    return (r << 2) | (r >> 4);
}
} // namespace

template <bool PARANOID>
__global__ __launch_bounds__(64) void cvttmi_bc1_kernel(const uint8_t *__restrict__ blocks, uint8_t *__restrict__ out,
                                                        const CvttBc1Args A, const CvttDeviceTables *__restrict__ T)
{
    const u32 blockIndex = blockIdx.x * 64u + threadIdx.x;
    const bool valid = blockIndex < A.numBlocks;

    u32 pix[16];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(blocks + (size_t)(valid ? blockIndex : 0u) * 64u);
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const uint4 v = src[i];
            pix[4 * i + 0] = v.x;
            pix[4 * i + 1] = v.y;
            pix[4 * i + 2] = v.z;
            pix[4 * i + 3] = v.w;
        }
    }

    // alpha test (S3TC.cpp:746-755): pixels below the threshold get PCA weight 0
    u32 opaqueMask = 0;
#pragma unroll
    for (int px = 0; px < 16; px++)
        if (!A.alphaTest || !(byteI(pix[px], 3) < A.threshold))
            opaqueMask |= 1u << px;

    // weighted PCA (S3TC.cpp:764-784): passes 0/1 over the opaque pixels, min/max pass over all
    Unfinished ufep;
    {
        Moments<3> m;
        pcaMoments<3>(pix, opaqueMask, A.w, m);
        pcaFinish<3>(pix, 0xffffu, A.w, m, ufep);
    }

    int numRefine = A.refineRounds < 1 ? 1 : A.refineRounds;
    int maxTweak = A.seedPoints < 1 ? 1 : A.seedPoints;
    const bool uniformErr = (A.flags & CVTTMI_FLAG_UNIFORM) != 0;
    const float wRcp16 = T->rcpTable[16];

    float bestError = FLT_MAX;
    u32 bestEP0 = 0, bestEP1 = 0, bestIdx = 0; // indexes: 2 bits per pixel
    int bestRange = 0;

    for (int range = A.alphaTest ? 3 : 4; range <= 4; range++) // S3TC.cpp:939
    {
        int tweakRounds = (range == 3) ? 3 : 4; // BCCommon::TweakRoundsForRange
        if (tweakRounds > maxTweak)
            tweakRounds = maxTweak;
        const float maxValue = (float)(range - 1);
        const float rcpMaxIndex = (range == 3) ? 0.5f : T->rcpMaxIndex[2];
        const int weightRcp = (range == 3) ? 16384 : 10923; // g_weightReciprocals, IndexSelector.cpp:43-62

        for (int tweak = 0; tweak < tweakRounds; tweak++)
        {
            const float tf0 = T->tweakFactors3[range - 3][tweak][0];
            const float tf1 = T->tweakFactors3[range - 3][tweak][1];
            int ep[2][3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                ep[0][ch] = (int)clampRound(ufep.base[ch] + ufep.offset[ch] * tf0, 255.0f);
                ep[1][ch] = (int)clampRound(ufep.base[ch] + ufep.offset[ch] * tf1, 255.0f);
            }

            for (int refine = 0; refine < numRefine; refine++)
            {
                // TestEndpoints, S3TC.cpp:190-258
                int q[2][3];
#pragma unroll
                for (int j = 0; j < 2; j++)
                {
                    q[j][0] = quantize5(ep[j][0]);
                    q[j][1] = quantize6(ep[j][1]);
                    q[j][2] = quantize5(ep[j][2]);
                }
                float origin[3], axis[3], paranoid[3];
                int recBase[3], recDelta[3];
                {
                    float epDW[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        origin[ch] = (float)q[0][ch];
                        epDW[ch] = ((float)q[1][ch] - origin[ch]) * A.w[ch];
                        recBase[ch] = (q[0][ch] << 8) + 128;
                        recDelta[ch] = q[1][ch] - q[0][ch];
                        paranoid[ch] = fabsf((float)(q[0][ch] - q[1][ch])) * 0.03f; // ParanoidFactorForSpan
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

                float error = 0.0f;
                u32 err[3] = {0, 0, 0};
                float tv[3] = {0, 0, 0}, vs[3] = {0, 0, 0};
                float tt = 0.0f, ts = 0.0f;
                u32 idxBits = 0;
#pragma unroll
                for (int px = 0; px < 16; px++)
                {
                    const u32 pk = fetchPixel(pix[px]);
                    float dist = (byteF(pk, 0) - origin[0]) * axis[0];
                    dist = dist + (byteF(pk, 1) - origin[1]) * axis[1];
                    dist = dist + (byteF(pk, 2) - origin[2]) * axis[2];
                    const float fidx = clampRound(dist, maxValue);
                    const int index = (int)fidx;
                    idxBits |= (u32)index << (2 * px);

                    // the refiner is fed in every round (S3TC.cpp:223-224)
                    const float t = fidx * rcpMaxIndex;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const float v = byteF(pk, ch) * A.w[ch];
                        tv[ch] = tv[ch] + t * v;
                        vs[ch] = vs[ch] + v;
                    }
                    tt = tt + t * t;
                    ts = ts + t;

                    // ReconstructLDRPrecise (IndexSelector.h:102-112)
                    const int wgt = mad24(weightRcp, index, 64) >> 7;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const int rec = mad24(wgt, recDelta[ch], recBase[ch]) >> 8;
                        const int d = rec - byteI(pk, ch);
                        if (PARANOID)
                        {
                            float ad = fabsf((float)d) + paranoid[ch]; // ParanoidDiff, S3TC.cpp:76-81
                            error = error + ad * ad * A.wSq[ch];
                        }
                        else
                            err[ch] = (u32)mad24(d, d, (int)err[ch]);
                    }
                }
                if (!PARANOID)
                {
                    if (uniformErr)
                        error = (float)(int)(err[0] + err[1] + err[2]);
                    else
                    {
                        error = (float)(int)err[0] * A.wSq[0];
                        error = error + (float)(int)err[1] * A.wSq[1];
                        error = error + (float)(int)err[2] * A.wSq[2];
                    }
                }

                if (error < bestError)
                {
                    bestError = error;
                    bestEP0 = (u32)q[0][0] | ((u32)q[0][1] << 8) | ((u32)q[0][2] << 16);
                    bestEP1 = (u32)q[1][0] | ((u32)q[1][1] << 8) | ((u32)q[1][2] << 16);
                    bestIdx = idxBits;
                    bestRange = range;
                }

                if (refine != numRefine - 1)
                {
                    // EndpointRefiner<3>::GetRefinedEndpointsLDR, 16 contributions (EndpointRefiner.h:99-152)
                    float adenom = (tt * 16.0f - ts * ts) * wRcp16;
                    const bool z = (adenom == 0.0f);
                    if (z)
                        adenom = 1.0f;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const float a = (tv[ch] - ts * vs[ch] * wRcp16) / adenom;
                        const float b = (vs[ch] - a * ts) * wRcp16;
                        float p1 = b, p2 = a + b;
                        if (z)
                        {
                            p1 = vs[ch] * wRcp16;
                            p2 = p1;
                        }
                        ep[0][ch] = (int)clampRound(p1 * A.rcpW[ch], 255.0f);
                        ep[1][ch] = (int)clampRound(p2 * A.rcpW[ch], 255.0f);
                    }
                }
            }
        }
    }

    // colour ordering + index remap (S3TC.cpp:967-1051)
    u32 c0 = ((bestEP0 & 0xf8u) << 8) | (((bestEP0 >> 8) & 0xfcu) << 3) | (((bestEP0 >> 16) & 0xf8u) >> 3);
    u32 c1 = ((bestEP1 & 0xf8u) << 8) | (((bestEP1 >> 8) & 0xfcu) << 3) | (((bestEP1 >> 16) & 0xf8u) >> 3);
    u32 order; // 2 bits per source index
    if (bestRange == 4)
    {
        if (c0 == c1)
            order = 0;
        else if (c0 < c1)
        {
            const u32 t = c0; c0 = c1; c1 = t;
            order = 1u | (3u << 2) | (2u << 4) | (0u << 6);
        }
        else
            order = 0u | (2u << 2) | (3u << 4) | (1u << 6);
    }
    else
    {
        if (c0 > c1)
        {
            const u32 t = c0; c0 = c1; c1 = t;
            order = 1u | (2u << 2) | (0u << 4) | (3u << 6);
        }
        else
            order = 0u | (2u << 2) | (1u << 4) | (3u << 6);
    }
    u32 packedIdx = 0;
#pragma unroll
    for (int px = 0; px < 16; px++)
    {
        const u32 index = (bestIdx >> (2 * px)) & 3u;
        packedIdx |= ((order >> (2 * index)) & 3u) << (2 * px);
    }
    if (valid)
    {
        uint2 o;
        o.x = c0 | (c1 << 16);
        o.y = packedIdx;
        *reinterpret_cast<uint2 *>(out + (size_t)blockIndex * A.outStride + A.outOffset) = o;
    }
}

extern "C" hipError_t cvttmi_launch_bc1(const void *d_blocks, void *d_out, const CvttBc1Args *args,
                                        const CvttDeviceTables *d_tables, hipStream_t stream)
{
    const uint32_t waves = (args->numBlocks + 63u) / 64u;
    if (waves == 0)
        return hipSuccess;
    if (args->flags & CVTTMI_FLAG_S3TC_PARANOID)
        hipLaunchKernelGGL(cvttmi_bc1_kernel<true>, dim3(waves), dim3(64), 0, stream, (const uint8_t *)d_blocks,
                           (uint8_t *)d_out, *args, d_tables);
    else
        hipLaunchKernelGGL(cvttmi_bc1_kernel<false>, dim3(waves), dim3(64), 0, stream, (const uint8_t *)d_blocks,
                           (uint8_t *)d_out, *args, d_tables);
    return hipGetLastError();
}
