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

template <bool PARANOID, bool EXHAUSTIVE>
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

    float bestError = FLT_MAX;
    u32 bestEP0 = 0, bestEP1 = 0, bestIdx = 0; // indexes: 2 bits per pixel
    int bestRange = 0;

    // TestEndpoints (S3TC.cpp:190-258): quantise to 5:6:5, select indexes, error; optionally feed the refiner sums
    auto testEndpoints = [&](const int (&ep)[2][3], int range, bool feed, float (&tv)[3], float (&vs)[3], float &tt, float &ts) {
        const float maxValue = (float)(range - 1);
        const float rcpMaxIndex = (range == 3) ? 0.5f : T->rcpMaxIndex[2];
        const int weightRcp = (range == 3) ? 16384 : 10923; // g_weightReciprocals, IndexSelector.cpp:43-62
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
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
            tv[ch] = vs[ch] = 0.0f;
        tt = ts = 0.0f;
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

            if (feed)
            {
                // the refiner is fed in every round of the seeded search (S3TC.cpp:223-224)
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
            }

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
    };

    // EndpointRefiner<3>::GetRefinedEndpointsLDR (EndpointRefiner.h:99-152) from sums over `count` contributions (0 -> 1)
    auto refinedEndpoints = [&](const float (&tv)[3], const float (&vs)[3], float tt, float ts, int count, int (&ep)[2][3]) {
        const int wi = count == 0 ? 1 : count;
        const float w = (float)wi, wRcp = T->rcpTable[wi];
        float adenom = (tt * w - ts * ts) * wRcp;
        const bool z = (adenom == 0.0f);
        if (z)
            adenom = 1.0f;
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
        {
            const float a = (tv[ch] - ts * vs[ch] * wRcp) / adenom;
            const float b = (vs[ch] - a * ts) * wRcp;
            float p1 = b, p2 = a + b;
            if (z)
            {
                p1 = vs[ch] * wRcp;
                p2 = p1;
            }
            ep[0][ch] = (int)clampRound(p1 * A.rcpW[ch], 255.0f);
            ep[1][ch] = (int)clampRound(p2 * A.rcpW[ch], 255.0f);
        }
    };

    if (!EXHAUSTIVE)
    {
        for (int range = A.alphaTest ? 3 : 4; range <= 4; range++) // S3TC.cpp:939
        {
            int tweakRounds = (range == 3) ? 3 : 4; // BCCommon::TweakRoundsForRange
            if (tweakRounds > maxTweak)
                tweakRounds = maxTweak;
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
                    float tv[3], vs[3], tt, ts;
                    testEndpoints(ep, range, true, tv, vs, tt, ts);
                    if (refine != numRefine - 1)
                        refinedEndpoints(tv, vs, tt, ts, 16, ep);
                }
            }
        }
    }
    else
    {
        // ---- S3TC_Exhaustive (S3TC.cpp:798-936): every split of the pixels, sorted along the PCA axis, into 4 (and 3)
        // clusters of consecutive pixels; least-squares end points per split (TestCounts, 260-304); single-colour tables ----
        // sort keys: 11-bit position along the axis << 4 | pixel; transparent pixels (alpha test) sort first as -16 + pixel
        int bins[16];
        {
            int sortEP[2][3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                // FinishLDR(tweak 0, range 11): factors (-0.0, 1.0), Util.cpp:75-84
                sortEP[0][ch] = (int)clampRound(ufep.base[ch] + ufep.offset[ch] * -0.0f, 255.0f);
                sortEP[1][ch] = (int)clampRound(ufep.base[ch] + ufep.offset[ch] * 1.0f, 255.0f);
            }
            float origin[3], axis[3], epDW[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                origin[ch] = (float)sortEP[0][ch];
                epDW[ch] = ((float)sortEP[1][ch] - origin[ch]) * A.w[ch];
            }
            float lenSq = epDW[0] * epDW[0];
            lenSq = lenSq + epDW[1] * epDW[1];
            lenSq = lenSq + epDW[2] * epDW[2];
            lenSq = safeDenom(lenSq);
            const float mvdls = 2047.0f / lenSq;
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                axis[ch] = epDW[ch] * A.w[ch] * mvdls;
#pragma unroll
            for (int px = 0; px < 16; px++)
            {
                const u32 pk = pix[px];
                float dist = (byteF(pk, 0) - origin[0]) * axis[0];
                dist = dist + (byteF(pk, 1) - origin[1]) * axis[1];
                dist = dist + (byteF(pk, 2) - origin[2]) * axis[2];
                int bin = (int)clampRound(dist, 2047.0f) << 4;
                if (A.alphaTest && !((opaqueMask >> px) & 1u))
                    bin = -16;
                bins[px] = bin + px;
            }
        }
#pragma unroll
        for (int sortEnd = 1; sortEnd < 16; sortEnd++)
#pragma unroll
            for (int loc = sortEnd; loc > 0; loc--)
            {
                const int a = bins[loc], b = bins[loc - 1];
                bins[loc] = a > b ? a : b;
                bins[loc - 1] = a > b ? b : a;
            }
        int firstElement = 0;
#pragma unroll
        for (int e = 0; e < 16; e++)
            if (bins[e] < 0)
                firstElement = e + 1;
        const int numElements = 16 - firstElement;
        // TestCounts stops contributing where NO lane of the group has elements left (AnySet, S3TC.cpp:275-281)
        int groupMaxElements = numElements;
#pragma unroll
        for (int step = 1; step <= 4; step <<= 1)
        {
            const int o = xorLane(groupMaxElements, step);
            groupMaxElements = o > groupMaxElements ? o : groupMaxElements;
        }
        // pre-weighted pixels in descending key order; slots past numElements stay zero
        float pwSorted[16][3];
#pragma unroll
        for (int slot = 0; slot < 16; slot++)
        {
            const int e = 15 - slot;
            u32 pk = 0;
#pragma unroll
            for (int px = 0; px < 16; px++)
                if ((bins[e] & 15) == px)
                    pk = pix[px];
            const bool live = e >= firstElement;
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                pwSorted[slot][ch] = live ? byteF(pk, ch) * A.w[ch] : 0.0f;
        }

        auto testCounts = [&](int c0, int c1, int c2, int nCounts) {
            // cluster i covers sorted slots [start_i, start_i + counts_i); slot e of cluster i is its n-th element
            const float rcpMaxIndex = (nCounts == 3) ? 0.5f : T->rcpMaxIndex[2];
            float tv[3] = {0, 0, 0}, vs[3] = {0, 0, 0}, tt = 0.0f, ts = 0.0f;
            int count = 0;
            bool escaped = false;
            const int s1 = c0, s2 = c0 + c1, s3 = c0 + c1 + c2;
#pragma unroll
            for (int e = 0; e < 16; e++)
            {
                const int i = (e >= s1 ? 1 : 0) + (e >= s2 ? 1 : 0) + ((nCounts == 4 && e >= s3) ? 1 : 0);
                const int start = (i == 0) ? 0 : (i == 1) ? s1 : (i == 2) ? s2 : s3;
                const int n = e - start;
                if (!(n < groupMaxElements))
                    escaped = true; // wave-uniform inside a group; lanes of other groups carry their own flag
                if (!escaped && n < numElements)
                {
                    const float t = (float)i * rcpMaxIndex;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        tv[ch] = tv[ch] + t * pwSorted[e][ch];
                        vs[ch] = vs[ch] + pwSorted[e][ch];
                    }
                    tt = tt + t * t;
                    ts = ts + t;
                    count++;
                }
            }
            int ep[2][3];
            refinedEndpoints(tv, vs, tt, ts, count, ep);
            float tv2[3], vs2[3], tt2, ts2;
            testEndpoints(ep, nCounts, false, tv2, vs2, tt2, ts2);
        };

        auto testSingleColor = [&](int range) {
            // TestSingleColor, S3TC.cpp:83-188
            u32 total[3] = {0, 0, 0};
#pragma unroll
            for (int px = 0; px < 16; px++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    total[ch] += (u32)byteI(pix[px], ch);
            const int t = (PARANOID ? 4 : 0) + (range == 3 ? 2 : 0);
            int eps[2][3], interpolated[3];
            float spanFactor[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
            {
                const int avg = (int)((total[ch] + 8u) >> 4);
                const uint8_t *e = T->s3tcSingleColor[t + (ch == 1 ? 1 : 0)][avg];
                eps[0][ch] = e[0];
                eps[1][ch] = e[1];
                interpolated[ch] = e[2];
                spanFactor[ch] = fabsf((float)(int)e[3]) * 0.03f;
            }
            float error = 0.0f;
#pragma unroll
            for (int px = 0; px < 16; px++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                {
                    const int d = interpolated[ch] - byteI(pix[px], ch);
                    if (PARANOID)
                    {
                        const float ad = fabsf((float)d) + spanFactor[ch];
                        error = error + ad * ad * A.wSq[ch];
                    }
                    else
                        error = error + (float)(d * d) * A.wSq[ch];
                }
            if (error < bestError)
            {
                bestError = error;
                bestEP0 = (u32)eps[0][0] | ((u32)eps[0][1] << 8) | ((u32)eps[0][2] << 16);
                bestEP1 = (u32)eps[1][0] | ((u32)eps[1][1] << 8) | ((u32)eps[1][2] << 16);
                bestIdx = 0x55555555u; // every index 1
                bestRange = range;
            }
        };

        for (int n0 = 0; n0 <= 15; n0++)
        {
            const int remainingFor1 = (16 - n0 == 16) ? 15 : 16 - n0;
            for (int n1 = 0; n1 <= remainingFor1; n1++)
            {
                const int remainingFor2 = (16 - n1 - n0 == 16) ? 15 : 16 - n1 - n0;
                for (int n2 = 0; n2 <= remainingFor2; n2++)
                {
                    if (16 - n2 - n1 - n0 == 16)
                        continue;
                    testCounts(n0, n1, n2, 4);
                }
            }
        }
        testSingleColor(4);
        if (A.alphaTest)
        {
            for (int n0 = 0; n0 <= 15; n0++)
            {
                const int remainingFor1 = (16 - n0 == 16) ? 15 : 16 - n0;
                for (int n1 = 0; n1 <= remainingFor1; n1++)
                {
                    if (16 - n1 - n0 == 16)
                        continue;
                    testCounts(n0, n1, 16 - n1 - n0, 3);
                }
            }
            testSingleColor(3);
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
    const bool paranoid = (args->flags & CVTTMI_FLAG_S3TC_PARANOID) != 0, exhaustive = (args->flags & CVTTMI_FLAG_S3TC_EXHAUSTIVE) != 0;
#define CVTT_LAUNCH(P, E) hipLaunchKernelGGL((cvttmi_bc1_kernel<P, E>), dim3(waves), dim3(64), 0, stream, (const uint8_t *)d_blocks, (uint8_t *)d_out, *args, d_tables)
    if (paranoid)
    {
        if (exhaustive) CVTT_LAUNCH(true, true); else CVTT_LAUNCH(true, false);
    }
    else
    {
        if (exhaustive) CVTT_LAUNCH(false, true); else CVTT_LAUNCH(false, false);
    }
#undef CVTT_LAUNCH
    return hipGetLastError();
}
