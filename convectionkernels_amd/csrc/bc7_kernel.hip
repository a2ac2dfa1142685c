// BC7 endpoint / partition / index search for gfx950 (MI355X), hand-written HIP.
//
// What it replaces: cvtt::Internal::BC7Computer::Pack and everything under it
// (reference ConvectionKernels_BC67.cpp:1975-2204 -> TrySinglePlane 1042-1662,
// TryDualPlane 1664-1965, CompressEndpoints* 829-938; EndpointSelector.h, EndpointRefiner.h,
// IndexSelector.h, AggregatedError.h).  Results are bit-identical to the reference's SSE2
// lanes; the arithmetic contract is kept by compiling this file with
// -ffp-contract=off and IEEE divide/sqrt, f32 denormals on.
//
// Mapping (this is not the reference's 8-blocks-per-SSE-register layout):
//   * one wavefront = 16 blocks = two reference "groups" of 8; lanes [0,32) and [32,64) each
//     form one group, whose two alpha-derived booleans (BC67.cpp:1069, 1072) are wave ballots.
//   * a block is owned by a lane QUAD.  Sub-lane c of the quad walks a quarter of the
//     candidate chains of the current shape -- the (p-bit, seed-point) pairs of the
//     reference's pIter x tweak loops (BC67.cpp:1298-1305) -- and runs each chain's refine
//     rounds sequentially in registers (the refinement chain is inherently serial).
//   * the 16 pixels of the block live packed (RGBA8) in 16 VGPRs per lane; the shape being
//     searched is wave-uniform, so the pixel loop is a fully unrolled sequence of scalar
//     (SGPR) bit tests: no divergence, no LDS, no scratch.
//   * per shape the quad reduces (error, chain id) with an order-preserving argmin -- the
//     reference commits with a strict '<' in p -> tweak -> refine order, so ties go to the
//     lowest chain id -- and broadcasts the winner's endpoints/indexes with quad shuffles.
//   * partition totals and the mode/partition commit are per-lane scalars (identical in
//     the four lanes of a quad); sub-lane 0 packs and stores the 16-byte block.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "cvtt_device.h"

// minimum waves per SIMD the register allocator must leave room for (512 VGPR+AGPR / waves)
#ifndef CVTT_BC7_WAVES
#define CVTT_BC7_WAVES 3
#endif

#include "cvtt_kernel_common.h"

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
        const float oErr = __shfl_xor(err, step);
        const int oWho = __shfl_xor(who, step);
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

// One shape of a single-plane mode: the reference's pIter x tweak x refine loops
// (BC67.cpp:1298-1434) spread over the quad.  NRC = numRealChannels (3 for modes 0-3).
template <int NRC, bool FAST>
__device__ __forceinline__ void evalShape(const u32 (&pix)[16], u32 mask, const ModeDesc md, const Unfinished &u,
                                          int numTweak, const CvttBc7Args &A, const CvttDeviceTables *__restrict__ T,
                                          int numRefine, int lane, ShapeBest &best)
{
    const bool isRGB = (NRC == 3);
    const int c = lane & 3;
    const int range = 1 << md.indexBits;
    const float maxValue = (float)(range - 1);
    const float rcpMaxIndex = T->rcpMaxIndex[md.indexBits];
    const int weightRcp = (65536 + (range - 1)) / (2 * (range - 1)); // g_weightReciprocals, IndexSelector.cpp:43-62
    const int count = __popc(mask);
    const float wRcp = T->rcpTable[count];
    const float wCount = (float)count;
    const bool uniformErr = (A.flags & CVTTMI_FLAG_UNIFORM) != 0;

    // static alpha error of RGB modes (reference BC67.cpp:1250-1264); zero on opaque groups
    float staticAlphaError = 0.0f;
    if (isRGB)
    {
        u32 acc = 0;
#pragma unroll
        for (int px = 0; px < 16; px++)
            if ((mask >> px) & 1u)
            {
                const int d = 255 - byteI(fetchPixel(pix[px]), 3);
                acc = (u32)mad24(d, d, (int)acc);
            }
        staticAlphaError = uniformErr ? (float)(int)acc : (float)(int)acc * A.wSq[3];
    }

    best.err = FLT_MAX;
    best.ep0 = best.ep1 = 0;
    best.idxLo = best.idxHi = 0;

    // chains of this sub-lane: (pIter, tweak) pairs in increasing sequential order
    int chainsPerLane, pIter, tweak0;
    if (md.numP == 4) { chainsPerLane = 4; pIter = c; tweak0 = 0; }
    else if (md.numP == 2) { chainsPerLane = 2; pIter = c >> 1; tweak0 = (c & 1) * 2; }
    else { chainsPerLane = 1; pIter = 0; tweak0 = c; }

    for (int k = 0; k < chainsPerLane; k++)
    {
        const int tweak = tweak0 + k;
        if (tweak < numTweak)
        {
            // UnfinishedEndpoints::FinishLDR (reference UnfinishedEndpoints.h:77-91)
            const float tf0 = T->tweakFactors[md.indexBits - 2][tweak][0];
            const float tf1 = T->tweakFactors[md.indexBits - 2][tweak][1];
            int ep[2][4];
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

            for (int refine = 0; refine < numRefine; refine++)
            {
                const bool last = (refine == numRefine - 1);
                compressEndpoints(md, ep, pIter, isRGB);

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
                        recBase[ch] = (ep[0][ch] << 6) + 32;
                        recDelta[ch] = ep[1][ch] - ep[0][ch];
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

                const u32 m = opaqueUniform(mask);
                u32 err[4] = {0, 0, 0, 0};
                float slowErr = 0.0f;
                float tv[4] = {0, 0, 0, 0}, vs[4] = {0, 0, 0, 0};
                float tt = 0.0f, ts = 0.0f;
                u32 idxLo = 0, idxHi = 0;

#pragma unroll
                for (int px = 0; px < 16; px++)
                {
                    if ((m >> px) & 1u)
                    {
                        const u32 pk = fetchPixel(pix[px]);
                        // SelectIndexLDR (reference IndexSelector.h:124-131)
                        float dist = (byteF(pk, 0) - origin[0]) * axis[0];
#pragma unroll
                        for (int ch = 1; ch < 4; ch++)
                            dist = dist + (byteF(pk, ch) - origin[ch]) * axis[ch];
                        float fidx = clampRound(dist, maxValue);
                        int index = (int)fidx;

                        if (FAST)
                        {
                            // ReconstructLDR_BC7 + ComputeErrorLDR (IndexSelector.h:90-100, BCCommon.h:24-29)
                            const int wgt = mad24(weightRcp, index, 256) >> 9;
#pragma unroll
                            for (int ch = 0; ch < NRC; ch++)
                            {
                                const int rec = mad24(wgt, recDelta[ch], recBase[ch]) >> 6;
                                const int d = rec - byteI(pk, ch);
                                err[ch] = (u32)mad24(d, d, (int)err[ch]);
                            }
                        }
                        else
                        {
                            // slow indexing: also probe index-1 / index+1 (reference BC67.cpp:1367-1386)
                            float bestE = 0.0f;
#pragma unroll
                            for (int probe = 0; probe < 3; probe++)
                            {
                                int cand = index;
                                if (probe == 1) cand = (index > 1 ? index : 1) - 1;
                                if (probe == 2) cand = (index + 1 < range - 1) ? index + 1 : range - 1;
                                if (probe == 0) cand = index;
                                const int baseIndex = index;
                                (void)baseIndex;
                                const int wgt = mad24(weightRcp, cand, 256) >> 9;
                                u32 e4[4] = {0, 0, 0, 0};
#pragma unroll
                                for (int ch = 0; ch < NRC; ch++)
                                {
                                    const int rec = mad24(wgt, recDelta[ch], recBase[ch]) >> 6;
                                    const int d = rec - byteI(pk, ch);
                                    e4[ch] = (u32)__mul24(d, d);
                                }
                                float e;
                                if (uniformErr)
                                    e = (float)(int)(e4[0] + e4[1] + e4[2] + e4[3]);
                                else
                                {
                                    e = (float)(int)e4[0] * A.wSq[0];
                                    e = e + (float)(int)e4[1] * A.wSq[1];
                                    e = e + (float)(int)e4[2] * A.wSq[2];
                                    e = e + (float)(int)e4[3] * A.wSq[3];
                                }
                                if (probe == 0)
                                    bestE = e;
                                else
                                {
                                    // alternatives are derived from the index chosen so far
                                    // (the reference computes both from the initial index)
                                    const bool better = e < bestE;
                                    bestE = sseMin(bestE, e);
                                    if (better)
                                        fidx = (float)cand;
                                }
                            }
                            slowErr = slowErr + bestE;
                            index = (int)fidx;
                        }

                        if (!last)
                        {
                            // EndpointRefiner::ContributeUnweightedPW (EndpointRefiner.h:78-92)
                            const float t = fidx * rcpMaxIndex;
#pragma unroll
                            for (int ch = 0; ch < NRC; ch++)
                            {
                                const float v = byteF(pk, ch) * A.w[ch];
                                tv[ch] = tv[ch] + t * v;
                                vs[ch] = vs[ch] + v;
                            }
                            tt = tt + t * t;
                            ts = ts + t;
                        }
                        if (px < 8)
                            idxLo |= (u32)index << (4 * px);
                        else
                            idxHi |= (u32)index << (4 * (px - 8));
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

                if (shapeError < best.err)
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
        }
    }
    quadArgminBroadcast(best, lane);
}

// 128-bit little-endian bit writer (reference PackingVector, BC67.cpp:652-698)
struct BitWriter
{
    u64 lo, hi;
    int off;
    __device__ __forceinline__ void put(u32 value, int bits)
    {
        const u64 v = (u64)value;
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

// In-place channel rotation of the packed pixels: rotation r > 0 exchanges channel r-1 with
// alpha (reference BC67.cpp:1695-1698).  The exchange is an involution.
__device__ __forceinline__ u32 rotatePixel(u32 pk, int rotation)
{
    if (rotation == 1) return (pk & 0x00ffff00u) | (pk >> 24) | (pk << 24);
    if (rotation == 2) return (pk & 0x00ff00ffu) | ((pk >> 16) & 0x0000ff00u) | ((pk << 16) & 0xff000000u);
    if (rotation == 3) return (pk & 0x0000ffffu) | ((pk >> 8) & 0x00ff0000u) | ((pk << 8) & 0xff000000u);
    return pk;
}

// One (mode, rotation, index selector) configuration of the dual-plane modes 4/5
// (reference BC67.cpp:1725-1940): sub-lane c runs seed point c.  `pix` is already rotated so
// that byte 3 is the separately coded channel; w/wSq/rcpW are rotated the same way.
template <bool FAST>
__device__ __forceinline__ void evalDual(const u32 (&pix)[16], int mode, int indexSelector, const Unfinished &uRGB,
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

    int alphaMin = byteI(pix[0], 3), alphaMax = alphaMin;
#pragma unroll
    for (int px = 1; px < 16; px++)
    {
        const int a = byteI(pix[px], 3);
        alphaMin = a < alphaMin ? a : alphaMin;
        alphaMax = a > alphaMax ? a : alphaMax;
    }

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
                    recBase[ch] = (ep[0][ch] << 6) + 32;
                    recDelta[ch] = ep[1][ch] - ep[0][ch];
                }
            }

            u32 err[4] = {0, 0, 0, 0};
            float slowRGB = 0.0f, slowA = 0.0f;
            float tv[4] = {0, 0, 0, 0}, vs[4] = {0, 0, 0, 0};
            float ttRGB = 0.0f, tsRGB = 0.0f, ttA = 0.0f, tsA = 0.0f;
            u32 rgbLo = 0, rgbHi = 0, aLo = 0, aHi = 0;

#pragma unroll
            for (int px = 0; px < 16; px++)
            {
                const u32 pk = fetchPixel(pix[px]);
                float dist = (byteF(pk, 0) - origin[0]) * axis[0];
                dist = dist + (byteF(pk, 1) - origin[1]) * axis[1];
                dist = dist + (byteF(pk, 2) - origin[2]) * axis[2];
                float fRGB = clampRound(dist, rgbMax);
                float fA = clampRound((byteF(pk, 3) - origin[3]) * axis[3], alphaMaxV);
                int iRGB = (int)fRGB, iA = (int)fA;

                if (FAST)
                {
                    const int wgt = mad24(rgbWR, iRGB, 256) >> 9;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const int rec = mad24(wgt, recDelta[ch], recBase[ch]) >> 6;
                        const int d = rec - byteI(pk, ch);
                        err[ch] = (u32)mad24(d, d, (int)err[ch]);
                    }
                    const int wa = mad24(alphaWR, iA, 256) >> 9;
                    const int recA = mad24(wa, recDelta[3], recBase[3]) >> 6;
                    const int dA = recA - byteI(pk, 3);
                    err[3] = (u32)mad24(dA, dA, (int)err[3]);
                }
                else
                {
                    // reference BC67.cpp:1834-1880
                    float eRGB = 0.0f, eA = 0.0f;
#pragma unroll
                    for (int probe = 0; probe < 3; probe++)
                    {
                        int candRGB = iRGB, candA = iA;
                        if (probe == 1)
                        {
                            candRGB = (iRGB > 1 ? iRGB : 1) - 1;
                            candA = (iA > 1 ? iA : 1) - 1;
                        }
                        if (probe == 2)
                        {
                            candRGB = (iRGB + 1 < rgbRange - 1) ? iRGB + 1 : rgbRange - 1;
                            candA = (iA + 1 < alphaRange - 1) ? iA + 1 : alphaRange - 1;
                        }
                        const int wgt = mad24(rgbWR, candRGB, 256) >> 9;
                        u32 e3[3];
#pragma unroll
                        for (int ch = 0; ch < 3; ch++)
                        {
                            const int rec = mad24(wgt, recDelta[ch], recBase[ch]) >> 6;
                            const int d = rec - byteI(pk, ch);
                            e3[ch] = (u32)__mul24(d, d);
                        }
                        const int wa = mad24(alphaWR, candA, 256) >> 9;
                        const int recA = mad24(wa, recDelta[3], recBase[3]) >> 6;
                        const int dA = recA - byteI(pk, 3);
                        const u32 e1 = (u32)__mul24(dA, dA);
                        float er, ea;
                        if (uniformErr)
                        {
                            er = (float)(int)(e3[0] + e3[1] + e3[2]);
                            ea = (float)(int)e1;
                        }
                        else
                        {
                            er = (float)(int)e3[0] * rwSq[0];
                            er = er + (float)(int)e3[1] * rwSq[1];
                            er = er + (float)(int)e3[2] * rwSq[2];
                            ea = (float)(int)e1 * rwSq[3];
                        }
                        if (probe == 0)
                        {
                            eRGB = er;
                            eA = ea;
                        }
                        else
                        {
                            const bool bR = er < eRGB, bA = ea < eA;
                            eRGB = sseMin(er, eRGB);
                            eA = sseMin(ea, eA);
                            if (bR) fRGB = (float)candRGB;
                            if (bA) fA = (float)candA;
                        }
                    }
                    slowRGB = slowRGB + eRGB;
                    slowA = slowA + eA;
                    iRGB = (int)fRGB;
                    iA = (int)fA;
                }

                if (!last)
                {
                    const float t = fRGB * rgbRcpMax;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                    {
                        const float v = byteF(pk, ch) * rw[ch];
                        tv[ch] = tv[ch] + t * v;
                        vs[ch] = vs[ch] + v;
                    }
                    ttRGB = ttRGB + t * t;
                    tsRGB = tsRGB + t;
                    const float ta = fA * alphaRcpMax;
                    const float va = byteF(pk, 3);
                    tv[3] = tv[3] + ta * va;
                    vs[3] = vs[3] + va;
                    ttA = ttA + ta * ta;
                    tsA = tsA + ta;
                }
                if (px < 8)
                {
                    rgbLo |= (u32)iRGB << (4 * px);
                    aLo |= (u32)iA << (4 * px);
                }
                else
                {
                    rgbHi |= (u32)iRGB << (4 * (px - 8));
                    aHi |= (u32)iA << (4 * (px - 8));
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
                {
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

} // namespace

// Broadcast the seeds computed by sub-lane `srcSub` of every quad to the whole quad.
__device__ __forceinline__ void quadBroadcast(Unfinished &dst, const Unfinished &src, int lane, int srcSub)
{
    const int from = (lane & ~3) | srcSub;
#pragma unroll
    for (int ch = 0; ch < 4; ch++)
    {
        dst.base[ch] = __shfl(src.base[ch], from);
        dst.offset[ch] = __shfl(src.offset[ch], from);
    }
}

template <bool FAST>
__global__ __launch_bounds__(64, CVTT_BC7_WAVES) void cvttmi_bc7_kernel(const uint8_t *__restrict__ blocks, uint8_t *__restrict__ out,
                                                        const CvttBc7Args A, const CvttDeviceTables *__restrict__ T,
                                                        const CvttBc7DevicePlan *__restrict__ dplan)
{
    const cvttmi_bc7_plan *__restrict__ plan = &dplan->plan;
    const int lane = threadIdx.x;
    const u32 blockIndex = blockIdx.x * 16u + (u32)(lane >> 2);
    const bool valid = blockIndex < A.numBlocks;
    const int c = lane & 3;

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

    // ---- per-block alpha statistics and the two group-wide booleans (BC67.cpp:1054-1078) ----
    int minAlpha = 255;
#pragma unroll
    for (int px = 0; px < 16; px++)
    {
        const int a = byteI(pix[px], 3);
        minAlpha = a < minAlpha ? a : minAlpha;
    }
    const bool blockHasNonMaxAlpha = minAlpha < 255;
    const u64 ballotA = __ballot(blockHasNonMaxAlpha);
    const u64 ballotR = __ballot(250 < minAlpha);
    const u32 groupA = (lane < 32) ? (u32)ballotA : (u32)(ballotA >> 32);
    const u32 groupR = (lane < 32) ? (u32)ballotR : (u32)(ballotR >> 32);
    const bool anyBlockHasAlpha = groupA != 0;
    const bool allowRGBModes = groupR != 0;
    const u64 mode7RGB = plan->mode7RGBPartitionEnabled;
    const bool allowMode7 = anyBlockHasAlpha || (mode7RGB != 0);
    // RGBA seeds: PCA over 4 channels when the group has alpha or no RGB modes, otherwise the
    // RGB seeds extended with alpha = 255 (reference BC67.cpp:1113-1144)
    const bool wantPCA4 = anyBlockHasAlpha || !allowRGBModes;
    const bool anyWantsPCA4 = __ballot(wantPCA4) != 0;
    const bool anyWantsExpand = __ballot(!wantPCA4) != 0;

    int numRefine = A.refineRounds;
    if (numRefine < 1)
        numRefine = 1;

    // Running best.  The reference walks candidates in a fixed order (single-plane modes
    // 0,1,2,3,6,7 by partition, then mode 4 / mode 5 by rotation and index selector) and
    // commits on a strict '<', i.e. it keeps the FIRST candidate that reaches the minimum.
    // We evaluate in a different order, so every candidate carries its position `seq` in the
    // reference's order and the commit compares (error, seq) lexicographically.
    WorkState work;
    work.err = FLT_MAX;
    int workSeq = -1; // nothing committed yet: a candidate must beat FLT_MAX strictly
    work.mode = 0;
    work.partOrIS = 0;
    work.rotation = 0;
#pragma unroll
    for (int s = 0; s < 3; s++)
        work.ep[s][0] = work.ep[s][1] = 0;
    work.idxLo = work.idxHi = work.idx2Lo = work.idx2Hi = 0;

    // ================================ dual-plane modes 4,5 ================================
    // reference TryDualPlane, BC67.cpp:1678-1963.  The RGB seeds depend only on the rotation,
    // so sub-lane r computes them for rotation r once (the reference recomputes them for each
    // mode / index selector).
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

        int curRotation = 0;
        for (int cfg = 0; cfg < 12; cfg++)
        {
            // cfg order = reference commit order: mode 4 (rot 0: is 0,1; rot 1: ...), then mode 5 rot 0..3
            const int mode = cfg < 8 ? 4 : 5;
            const int rotation = cfg < 8 ? (cfg >> 1) : (cfg - 8);
            const int indexSelector = cfg < 8 ? (cfg & 1) : 0;
            int numTweak = (mode == 4) ? plan->mode4SP[rotation][indexSelector] : plan->mode5SP[rotation];
            if (numTweak <= 0)
                continue;
            if (numTweak > 4)
                numTweak = 4;

            if (rotation != curRotation)
            {
#pragma unroll
                for (int px = 0; px < 16; px++)
                    pix[px] = rotatePixel(rotatePixel(pix[px], curRotation), rotation);
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

            Unfinished u;
            quadBroadcast(u, uRot, lane, rotation);
            ShapeBest b, bA;
            evalDual<FAST>(pix, mode, indexSelector, u, numTweak, rw, rwSq, rrcpW, A.flags, T, numRefine, lane, b, bA);

            const float combined = b.err + bA.err; // reference BC67.cpp:1942
            const int seq = 384 + cfg;
            if (combined < work.err || (combined == work.err && seq < workSeq))
            {
                work.err = combined;
                workSeq = seq;
                work.mode = mode;
                work.rotation = rotation;
                work.partOrIS = indexSelector;
                work.ep[0][0] = b.ep0 | bA.ep0;
                work.ep[0][1] = b.ep1 | bA.ep1;
                if (indexSelector)
                {
                    // index selector 1: the 2-bit set is the alpha plane (BC67.cpp:1953-1957)
                    work.idxLo = bA.idxLo;
                    work.idxHi = bA.idxHi;
                    work.idx2Lo = b.idxLo;
                    work.idx2Hi = b.idxHi;
                }
                else
                {
                    work.idxLo = b.idxLo;
                    work.idxHi = b.idxHi;
                    work.idx2Lo = bA.idxLo;
                    work.idx2Hi = bA.idxHi;
                }
            }
        }
        if (curRotation != 0)
        {
#pragma unroll
            for (int px = 0; px < 16; px++)
                pix[px] = rotatePixel(pix[px], curRotation);
        }
    }

    // =========================== single-plane modes 0,1,2,3,6,7 ===========================
    // reference TrySinglePlane, BC67.cpp:1146-1660.  The (partition, subset) pairs of a mode
    // form a flat list of shape evaluations; it is consumed in batches of four: sub-lane c
    // runs the PCA seed search for item 4*batch + c (per-lane shape mask), then the four
    // items are searched one after the other with the shape wave-uniform.
    for (int stageIter = 0; stageIter < 6; stageIter++)
    {
        ModeDesc md;
        int numSubsets, numPartitions, stage;
        u64 enabled;
        switch (stageIter)
        {
        case 0: stage = 4; md = {6, 4, 4, 7, 0}; numSubsets = 1; numPartitions = 1; enabled = plan->mode6Enabled ? 1 : 0; break;
        case 1: stage = 5; md = {7, 2, 4, 5, 6}; numSubsets = 2; numPartitions = 64; enabled = ~0ull; break; // dead mask in the reference (BC67.cpp:1592-1597)
        case 2: stage = 1; md = {1, 3, 2, 6, 7}; numSubsets = 2; numPartitions = 64; enabled = plan->mode1PartitionEnabled; break;
        case 3: stage = 3; md = {3, 2, 4, 7, 0}; numSubsets = 2; numPartitions = 64; enabled = plan->mode3PartitionEnabled; break;
        case 4: stage = 0; md = {0, 3, 4, 4, 5}; numSubsets = 3; numPartitions = 16; enabled = plan->mode0PartitionEnabled; break;
        default: stage = 2; md = {2, 2, 1, 5, 5}; numSubsets = 3; numPartitions = 64; enabled = plan->mode2PartitionEnabled; break;
        }
        const int mode = md.mode;
        const bool isRGB = mode < 4;
        // does this mode run for my group?  (wave-uniform skip when it runs for nobody)
        const bool laneRuns = isRGB ? allowRGBModes : (mode == 7 ? allowMode7 : true);
        if (__ballot(laneRuns) == 0 || enabled == 0)
            continue;

        const int numItems = numPartitions * numSubsets;
        float totalError = 0.0f;
        u32 pe00 = 0, pe01 = 0, pe10 = 0, pe11 = 0, pe20 = 0, pe21 = 0;
        u32 pIdxLo = 0, pIdxHi = 0;

        int deadPartition = -1; // partition proven unable to win (skipped from here on)

        for (int batch = 0; batch * 4 < numItems; batch++)
        {
            // ---- phase A: moments (and the error lower bound) of up to four items, one per sub-lane ----
            Moments<3> m3;
            Moments<4> m4;
            u32 myMask = 0;
            bool do3 = false, do4 = false, expandAlpha = false;
            float lbMine = 0.0f;
            {
                const int item = batch * 4 + c;
                int partition, sub;
                if (numSubsets == 1) { partition = item; sub = 0; }
                else if (numSubsets == 2) { partition = item >> 1; sub = item & 1; }
                else { partition = item / 3; sub = item - partition * 3; }
                const bool live = item < numItems && ((enabled >> partition) & 1ull) != 0;
                int shape = 0;
                if (live)
                {
                    if (numSubsets == 2)
                        shape = T->shapes2[partition][sub];
                    else if (numSubsets == 3)
                        shape = T->shapes3[partition][sub];
                }
                myMask = T->shapeMask[shape];
                const int seeds = isRGB ? plan->seedPointsForShapeRGB[shape] : plan->seedPointsForShapeRGBA[shape];
                const bool rgbListed = ((dplan->rgbListed[shape >> 5] >> (shape & 31)) & 1u) != 0;
                const bool rgbaListed = isRGB ? true : (((dplan->rgbaListed[shape >> 5] >> (shape & 31)) & 1u) != 0);
                const bool wanted = live && seeds != 0;
                // which PCA does this lane need?  (BC67.cpp:1085-1144; unlisted shapes keep zero seeds)
                do4 = wanted && !isRGB && wantPCA4 && rgbaListed;
                do3 = wanted && rgbListed && (isRGB || (!wantPCA4 && rgbaListed));
                expandAlpha = !isRGB && wanted && !wantPCA4 && rgbaListed;
                const float n = (float)__popc(myMask);
                if (__ballot(do3) != 0)
                {
                    pcaMoments<3>(pix, do3 ? myMask : 0u, A.w, m3);
                    if (do3 && A.prune)
                        lbMine = shapeErrorLowerBound<3>(m3, n, isRGB ? A.delta3 : A.delta4);
                }
                if (__ballot(do4) != 0)
                {
                    pcaMoments<4>(pix, do4 ? myMask : 0u, A.w, m4);
                    if (do4 && A.prune)
                        lbMine = shapeErrorLowerBound<4>(m4, n, A.delta4);
                }
                if (isRGB && wanted && A.prune)
                {
                    // the RGB modes add the exact error of replacing alpha by 255 (BC67.cpp:1250-1264)
                    u32 acc = 0;
#pragma unroll
                    for (int px = 0; px < 16; px++)
                        if ((myMask >> px) & 1u)
                        {
                            const int d = 255 - byteI(pix[px], 3);
                            acc = (u32)mad24(d, d, (int)acc);
                        }
                    const float st = (A.flags & CVTTMI_FLAG_UNIFORM) ? (float)(int)acc : (float)(int)acc * A.wSq[3];
                    lbMine = lbMine + st;
                }
                if (live && seeds == 0)
                    lbMine = FLT_MAX; // its error stays FLT_MAX (BC67.cpp:1228-1242)
                if (!live)
                    lbMine = 0.0f;
            }

            // ---- phase B: which items can still win?  bound = errors already known for the
            // partition + lower bounds of its items in this batch; wave-uniform decisions ----
            bool pruneItem[4];
            float lbItem[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
                lbItem[j] = __shfl(lbMine, (lane & ~3) | j);
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                pruneItem[j] = false;
                const int item = batch * 4 + j;
                if (A.prune && item < numItems)
                {
                    int partition, sub;
                    if (numSubsets == 1) { partition = item; sub = 0; }
                    else if (numSubsets == 2) { partition = item >> 1; sub = item & 1; }
                    else { partition = item / 3; sub = item - partition * 3; }
                    // items of the same partition inside this batch: [jLo, jHi]
                    int jLo = j - sub;
                    int jHi = jLo + numSubsets - 1;
                    float bound = (jLo < 0) ? totalError : 0.0f; // earlier subsets came with previous batches
                    if (jLo < 0) jLo = 0;
                    if (jHi > 3) jHi = 3;
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (i >= jLo && i <= jHi)
                            bound = bound + lbItem[i];
                    bound = bound * 0.999999f;
                    const bool cannotWin = bound > work.err;
                    pruneItem[j] = __ballot(valid && laneRuns && !cannotWin) == 0;
                }
            }

            // ---- phase C: finish the seed search of the items that survive ----
            Unfinished uMine;
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
                uMine.base[ch] = uMine.offset[ch] = 0.0f;
            {
                const bool myPruned = (c == 0) ? pruneItem[0] : (c == 1) ? pruneItem[1] : (c == 2) ? pruneItem[2] : pruneItem[3];
                const bool fin3 = do3 && !myPruned, fin4 = do4 && !myPruned;
                if (__ballot(fin3) != 0)
                {
                    Unfinished u3;
                    pcaFinish<3>(pix, fin3 ? myMask : 0u, A.w, m3, u3);
                    if (fin3)
                    {
#pragma unroll
                        for (int ch = 0; ch < 3; ch++)
                        {
                            uMine.base[ch] = u3.base[ch];
                            uMine.offset[ch] = u3.offset[ch];
                        }
                    }
                }
                if (expandAlpha)
                {
                    uMine.base[3] = 255.0f; // ExpandTo<4>(255), UnfinishedEndpoints.h:93-114
                    uMine.offset[3] = 0.0f;
                }
                if (__ballot(fin4) != 0)
                {
                    Unfinished u4;
                    pcaFinish<4>(pix, fin4 ? myMask : 0u, A.w, m4, u4);
                    if (fin4)
                        uMine = u4;
                }
            }

            // ---- phase D: search the four items ----
            for (int j = 0; j < 4; j++)
            {
                const int item = batch * 4 + j;
                if (item >= numItems)
                    break;
                int partition, sub;
                if (numSubsets == 1) { partition = item; sub = 0; }
                else if (numSubsets == 2) { partition = item >> 1; sub = item & 1; }
                else { partition = item / 3; sub = item - partition * 3; }
                if (((enabled >> partition) & 1ull) == 0)
                    continue;
                const bool prunedNow = (j == 0) ? pruneItem[0] : (j == 1) ? pruneItem[1] : (j == 2) ? pruneItem[2] : pruneItem[3];
                if (prunedNow)
                    deadPartition = partition;
                if (partition == deadPartition)
                    continue;

                int shape = 0;
                if (numSubsets == 2)
                    shape = T->shapes2[partition][sub];
                else if (numSubsets == 3)
                    shape = T->shapes3[partition][sub];
                const u32 mask = T->shapeMask[shape];
                int numTweak = isRGB ? plan->seedPointsForShapeRGB[shape] : plan->seedPointsForShapeRGBA[shape];
                if (numTweak > 4)
                    numTweak = 4;

                if (sub == 0)
                {
                    totalError = 0.0f;
                    pIdxLo = pIdxHi = 0;
                }
                ShapeBest b;
                if (numTweak <= 0)
                {
                    b.err = FLT_MAX; // shapeBestError stays at its reset value (BC67.cpp:1228-1242)
                    b.ep0 = b.ep1 = b.idxLo = b.idxHi = 0;
                }
                else
                {
                    Unfinished u;
                    quadBroadcast(u, uMine, lane, j);
#ifndef CVTT_EXP_NO_RGB
                    if (isRGB)
                        evalShape<3, FAST>(pix, mask, md, u, numTweak, A, T, numRefine, lane, b);
                    else
#endif
                        evalShape<4, FAST>(pix, mask, md, u, numTweak, A, T, numRefine, lane, b);
                }
                totalError = totalError + b.err;
                if (sub == 0) { pe00 = b.ep0; pe01 = b.ep1; }
                else if (sub == 1) { pe10 = b.ep0; pe11 = b.ep1; }
                else { pe20 = b.ep0; pe21 = b.ep1; }
                pIdxLo |= b.idxLo;
                pIdxHi |= b.idxHi;

                if (sub == numSubsets - 1)
                {
                    const int seq = stage * 64 + partition;
                    bool better = laneRuns && (totalError < work.err || (totalError == work.err && seq < workSeq));
                    if (mode == 7 && anyBlockHasAlpha)
                    {
                        // lanes without alpha may only take partitions enabled for RGB (BC67.cpp:1625-1635)
                        const bool rgbAllowed = ((mode7RGB >> partition) & 1ull) != 0;
                        if (!rgbAllowed)
                            better = better && blockHasNonMaxAlpha;
                    }
                    if (better)
                    {
                        work.err = totalError;
                        workSeq = seq;
                        work.mode = mode;
                        work.partOrIS = partition;
                        work.ep[0][0] = pe00;
                        work.ep[0][1] = pe01;
                        work.ep[1][0] = pe10;
                        work.ep[1][1] = pe11;
                        work.ep[2][0] = pe20;
                        work.ep[2][1] = pe21;
                        work.idxLo = pIdxLo;
                        work.idxHi = pIdxHi;
                    }
                }
                else if (A.prune)
                {
                    // the partition already costs more than the best: its remaining subsets
                    // cannot bring it back (errors are >= 0)
                    const bool cannotWin = totalError * 0.999999f > work.err;
                    if (__ballot(valid && laneRuns && !cannotWin) == 0)
                        deadPartition = partition;
                }
            }
        }
    }

    // ===================== fix-ups + bit packing (reference BC67.cpp:2003-2203) ==========
    {
        const int mode = work.mode;
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
        const bool separateAlpha = (mode == 4 || mode == 5);
        const bool combinedAlpha = (mode == 6 || mode == 7);
        const int partition = work.partOrIS;
        const int indexSelector = work.partOrIS;

        u64 idx = ((u64)work.idxHi << 32) | work.idxLo;
        u64 idx2 = ((u64)work.idx2Hi << 32) | work.idx2Lo;
        u32 ep[3][2];
#pragma unroll
        for (int s = 0; s < 3; s++)
        {
            ep[s][0] = work.ep[s][0];
            ep[s][1] = work.ep[s][1];
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

        BitWriter bw;
        bw.lo = bw.hi = 0;
        bw.off = 0;
        bw.put(1u << mode, mode + 1);
        if (partitionBits)
            bw.put((u32)partition, partitionBits);
        if (separateAlpha)
            bw.put((u32)work.rotation, 2);
        if (mode == 4)
            bw.put((u32)indexSelector, 1);
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
#pragma unroll
            for (int s = 0; s < 3; s++)
                if (s < numSubsets)
                {
                    bw.put(((ep[s][0] >> (8 * ch)) & 0xffu) >> (8 - rgbBits), rgbBits);
                    bw.put(((ep[s][1] >> (8 * ch)) & 0xffu) >> (8 - rgbBits), rgbBits);
                }
        if (alphaBits)
        {
#pragma unroll
            for (int s = 0; s < 3; s++)
                if (s < numSubsets)
                {
                    bw.put((ep[s][0] >> 24) >> (8 - alphaBits), alphaBits);
                    bw.put((ep[s][1] >> 24) >> (8 - alphaBits), alphaBits);
                }
        }
        if (pBitMode == 1)
        {
#pragma unroll
            for (int s = 0; s < 3; s++)
                if (s < numSubsets)
                    bw.put(((ep[s][0] & 0xffu) >> (7 - rgbBits)) & 1u, 1);
        }
        else if (pBitMode == 0)
        {
#pragma unroll
            for (int s = 0; s < 3; s++)
                if (s < numSubsets)
                {
                    bw.put(((ep[s][0] & 0xffu) >> (7 - rgbBits)) & 1u, 1);
                    bw.put(((ep[s][1] & 0xffu) >> (7 - rgbBits)) & 1u, 1);
                }
        }
#pragma unroll
        for (int px = 0; px < 16; px++)
        {
            int bits = indexBits;
            if (px == 0 || px == fix1 || px == fix2)
                bits--;
            bw.put((u32)((idx >> (4 * px)) & 0xfull), bits);
        }
        if (separateAlpha)
        {
#pragma unroll
            for (int px = 0; px < 16; px++)
            {
                int bits = alphaIndexBits;
                if (px == 0)
                    bits--;
                bw.put((u32)((idx2 >> (4 * px)) & 0xfull), bits);
            }
        }

        if (valid && c == 0)
        {
            uint4 o;
            o.x = (u32)bw.lo;
            o.y = (u32)(bw.lo >> 32);
            o.z = (u32)bw.hi;
            o.w = (u32)(bw.hi >> 32);
            *reinterpret_cast<uint4 *>(out + (size_t)blockIndex * 16u) = o;
        }
    }
}

extern "C" hipError_t cvttmi_launch_bc7(const void *d_blocks, void *d_out, const CvttBc7Args *args,
                                        const CvttDeviceTables *d_tables, const CvttBc7DevicePlan *d_plan,
                                        hipStream_t stream)
{
    const uint32_t waves = (args->numBlocks + 15u) / 16u;
    if (waves == 0)
        return hipSuccess;
    if (args->flags & CVTTMI_FLAG_BC7_FAST_INDEXING)
        hipLaunchKernelGGL(cvttmi_bc7_kernel<true>, dim3(waves), dim3(64), 0, stream, (const uint8_t *)d_blocks,
                           (uint8_t *)d_out, *args, d_tables, d_plan);
    else
        hipLaunchKernelGGL(cvttmi_bc7_kernel<false>, dim3(waves), dim3(64), 0, stream, (const uint8_t *)d_blocks,
                           (uint8_t *)d_out, *args, d_tables, d_plan);
    return hipGetLastError();
}
