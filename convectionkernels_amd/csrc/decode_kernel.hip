// BC7 and BC6H decoders: cvtt::Kernels::DecodeBC7 / DecodeBC6HU / DecodeBC6HS
// (reference ConvectionKernels_API.cpp:288-310 -> BC7Computer::UnpackOne, ConvectionKernels_BC67.cpp:
// 2206-2423, and BC6HComputer::UnpackOne, 3058-3289).  Integer-only, one lane per block: 16 bytes
// in, one PixelBlockU8 (64 B) or PixelBlockF16 (128 B) out; they let quality be checked on the
// device (decode + PSNR) independently of bit-exactness.  Blocks with a reserved mode decode to
// zeros (BC7) / zeros with alpha 1.0 (BC6H), like the reference.
#include "cvtt_kernel_common.h"

namespace
{
struct BitReader
{
    u64 lo, hi;
    int pos;
    __device__ __forceinline__ u32 get(int bits)
    {
        if (bits <= 0)
            return 0;
        u64 v;
        if (pos < 64)
        {
            v = lo >> pos;
            if (pos + bits > 64)
                v |= hi << (64 - pos);
        }
        else
            v = hi >> (pos - 64);
        pos += bits;
        return (u32)(v & ((1ull << bits) - 1ull));
    }
};

__device__ __forceinline__ int bc7Weight(int indexBits, int index)
{
    // g_weightTables as g_weightReciprocals (IndexSelector.cpp:43-62): identical values
    const int range = 1 << indexBits;
    const int rcp = (65536 + (range - 1)) / (2 * (range - 1));
    return (rcp * index + 256) >> 9;
}

__global__ void cvttmi_decode_bc7_kernel(const uint8_t *__restrict__ bc, uint8_t *__restrict__ out, u32 numBlocks,
                                         const CvttDeviceTables *__restrict__ T)
{
    const u32 block = blockIdx.x * blockDim.x + threadIdx.x;
    if (block >= numBlocks)
        return;
    const uint4 raw = *reinterpret_cast<const uint4 *>(bc + (size_t)block * 16u);
    BitReader br = {((u64)raw.y << 32) | raw.x, ((u64)raw.w << 32) | raw.z, 0};
    uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)block * 64u);

    int mode = 8;
    for (int i = 0; i < 8; i++)
        if (br.get(1) == 1)
        {
            mode = i;
            break;
        }
    if (mode > 7)
    {
        for (int i = 0; i < 4; i++)
            dst[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    // mode table of the format (reference BC67.cpp:108-123)
    const int numSubsetsTab[8] = {3, 2, 3, 2, 1, 1, 1, 2};
    const int partitionBitsTab[8] = {4, 6, 6, 6, 0, 0, 0, 6};
    const int rgbBitsTab[8] = {4, 6, 5, 7, 5, 7, 7, 5};
    const int alphaBitsTab[8] = {0, 0, 0, 0, 6, 8, 7, 5};
    const int indexBitsTab[8] = {3, 3, 2, 2, 2, 2, 4, 2};
    const int alphaIndexBitsTab[8] = {0, 0, 0, 0, 3, 2, 0, 0};
    const int pBitModeTab[8] = {0, 1, 2, 0, 2, 2, 0, 0}; // 0 per endpoint, 1 per subset, 2 none
    int numSubsets = 1, partitionBits = 0, rgbBits = 0, alphaBits = 0, indexBits = 0, alphaIndexBits = 0, pBitMode = 2;
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
    const bool hasAlpha = alphaBits != 0;

    const int partition = (int)br.get(partitionBits);
    const int rotation = separateAlpha ? (int)br.get(2) : 0;
    const int indexSelector = (mode == 4) ? (int)br.get(1) : 0;

    int fix1 = 0, fix2 = 0;
    if (!separateAlpha)
    {
        if (numSubsets == 2)
            fix1 = T->anchor2[partition & 63];
        else if (numSubsets == 3)
        {
            fix1 = T->anchor3[partition & 63][0];
            fix2 = T->anchor3[partition & 63][1];
        }
    }

    int ep[3][2][4];
    for (int ch = 0; ch < 3; ch++)
        for (int s = 0; s < 3; s++)
            for (int e = 0; e < 2; e++)
                ep[s][e][ch] = (s < numSubsets) ? (int)(br.get(rgbBits) << (8 - rgbBits)) : 0;
    for (int s = 0; s < 3; s++)
        for (int e = 0; e < 2; e++)
            ep[s][e][3] = (s < numSubsets && hasAlpha) ? (int)(br.get(alphaBits) << (8 - alphaBits)) : 255;

    int parityBits = 0;
    if (pBitMode != 2)
    {
        for (int s = 0; s < 3; s++)
            if (s < numSubsets)
            {
                int p = 0;
                for (int e = 0; e < 2; e++)
                {
                    if (pBitMode == 0 || e == 0)
                        p = (int)br.get(1);
                    for (int ch = 0; ch < 3; ch++)
                        ep[s][e][ch] |= p << (7 - rgbBits);
                    if (hasAlpha)
                        ep[s][e][3] |= p << (7 - alphaBits);
                }
            }
        parityBits = 1;
    }
    for (int s = 0; s < 3; s++)
        for (int e = 0; e < 2; e++)
        {
            for (int ch = 0; ch < 3; ch++)
                ep[s][e][ch] |= ep[s][e][ch] >> (rgbBits + parityBits);
            if (hasAlpha)
                ep[s][e][3] |= ep[s][e][3] >> (alphaBits + parityBits);
        }

    int idx[16], idx2[16];
    for (int px = 0; px < 16; px++)
    {
        const bool anchor = (px == 0) || (px == fix1) || (px == fix2);
        idx[px] = (int)br.get(indexBits - (anchor ? 1 : 0));
    }
    for (int px = 0; px < 16; px++)
        idx2[px] = separateAlpha ? (int)br.get(alphaIndexBits - (px == 0 ? 1 : 0)) : 0;

    const u32 map2 = T->partition2[partition & 63], map3 = T->partition3[partition & 63];
    u32 pixels[16];
    for (int px = 0; px < 16; px++)
    {
        int rgbWeight = bc7Weight(indexBits, idx[px]);
        int alphaWeight = 0;
        if (mode == 6 || mode == 7)
            alphaWeight = rgbWeight;
        else if (separateAlpha)
            alphaWeight = bc7Weight(alphaIndexBits, idx2[px]);
        if (indexSelector == 1)
        {
            const int t = rgbWeight;
            rgbWeight = alphaWeight;
            alphaWeight = t;
        }
        int subset = 0;
        if (numSubsets == 2)
            subset = (int)((map2 >> px) & 1u);
        else if (numSubsets == 3)
            subset = (int)((map3 >> (2 * px)) & 3u);
        int e0[4], e1[4];
        for (int ch = 0; ch < 4; ch++)
        {
            e0[ch] = (subset == 0) ? ep[0][0][ch] : (subset == 1) ? ep[1][0][ch] : ep[2][0][ch];
            e1[ch] = (subset == 0) ? ep[0][1][ch] : (subset == 1) ? ep[1][1][ch] : ep[2][1][ch];
        }
        int pixel[4] = {0, 0, 0, 255};
        for (int ch = 0; ch < 3; ch++)
            pixel[ch] = ((64 - rgbWeight) * e0[ch] + rgbWeight * e1[ch] + 32) >> 6;
        if (hasAlpha)
            pixel[3] = ((64 - alphaWeight) * e0[3] + alphaWeight * e1[3] + 32) >> 6;
        if (rotation != 0)
        {
            const int a = pixel[3];
            if (rotation == 1) { pixel[3] = pixel[0]; pixel[0] = a; }
            else if (rotation == 2) { pixel[3] = pixel[1]; pixel[1] = a; }
            else { pixel[3] = pixel[2]; pixel[2] = a; }
        }
        pixels[px] = ((u32)pixel[0] & 0xffu) | (((u32)pixel[1] & 0xffu) << 8) | (((u32)pixel[2] & 0xffu) << 16) | (((u32)pixel[3] & 0xffu) << 24);
    }
    for (int i = 0; i < 4; i++)
        dst[i] = make_uint4(pixels[4 * i], pixels[4 * i + 1], pixels[4 * i + 2], pixels[4 * i + 3]);
}

__device__ __forceinline__ int signExtend(int v, int bits)
{
    if (v & (1 << (bits - 1)))
        v |= -(1 << bits);
    return v;
}

template <bool SIGNED>
__global__ void cvttmi_decode_bc6h_kernel(const uint8_t *__restrict__ bc, uint8_t *__restrict__ out, u32 numBlocks,
                                          const CvttDeviceTables *__restrict__ T)
{
    const u32 block = blockIdx.x * blockDim.x + threadIdx.x;
    if (block >= numBlocks)
        return;
    const uint4 raw = *reinterpret_cast<const uint4 *>(bc + (size_t)block * 16u);
    BitReader br = {((u64)raw.y << 32) | raw.x, ((u64)raw.w << 32) | raw.z, 0};
    uint2 *dst = reinterpret_cast<uint2 *>(out + (size_t)block * 128u);

    int modeBits = (int)(raw.x & 3u);
    if (modeBits != 0 && modeBits != 1)
        modeBits = (int)(raw.x & 0x1fu);
    int mode = -1;
    for (int m = 0; m < 14; m++)
        if (mode < 0 && T->bc6hModeInfo[m][0] == modeBits)
            mode = m;
    if (mode < 0)
    {
        for (int px = 0; px < 16; px++)
            dst[px] = make_uint2(0u, 0x3c000000u);
        return;
    }
    const bool partitioned = T->bc6hModeInfo[mode][1] != 0;
    const bool transformed = T->bc6hModeInfo[mode][2] != 0;
    const int aPrec = T->bc6hModeInfo[mode][3];
    const int bPrec[3] = {T->bc6hModeInfo[mode][4], T->bc6hModeInfo[mode][5], T->bc6hModeInfo[mode][6]};
    const int headerBits = partitioned ? 82 : 65;

    // header bits -> fields (m d rw rx ry rz gw gx gy gz bw bx by bz), BC6H_IO.cpp via tools/gen_bc6h_layout.py
    u32 fields[14];
    for (int f = 0; f < 14; f++)
        fields[f] = 0;
    for (int bit = 0; bit < headerBits; bit++)
    {
        const u32 b = br.get(1);
        const u32 code = T->bc6hLayout[mode][bit];
        if (code != 255u)
        {
#pragma unroll
            for (int f = 0; f < 14; f++)
                if ((code >> 4) == (u32)f)
                    fields[f] |= b << (code & 15u);
        }
    }
    const int partition = (int)(fields[1] & 31u);
    int eps[2][2][3];
    for (int ch = 0; ch < 3; ch++)
    {
        eps[0][0][ch] = (int)fields[2 + ch * 4 + 0];
        eps[0][1][ch] = (int)fields[2 + ch * 4 + 1];
        eps[1][0][ch] = (int)fields[2 + ch * 4 + 2];
        eps[1][1][ch] = (int)fields[2 + ch * 4 + 3];
    }

    const int fixupIndex1 = partitioned ? (int)T->anchor2[partition] : 0;
    const int indexBits = partitioned ? 3 : 4;
    const int numSubsets = partitioned ? 2 : 1;
    int idx[16];
    for (int px = 0; px < 16; px++)
        idx[px] = (int)br.get((px == 0 || px == fixupIndex1) ? indexBits - 1 : indexBits);

    for (int ch = 0; ch < 3; ch++)
    {
        if (SIGNED)
            eps[0][0][ch] = signExtend(eps[0][0][ch], aPrec);
        if (transformed || SIGNED)
        {
            eps[0][1][ch] = signExtend(eps[0][1][ch], bPrec[ch]);
            if (partitioned)
            {
                eps[1][0][ch] = signExtend(eps[1][0][ch], bPrec[ch]);
                eps[1][1][ch] = signExtend(eps[1][1][ch], bPrec[ch]);
            }
        }
    }
    if (transformed)
    {
        const int wrapMask = (1 << aPrec) - 1;
        for (int ch = 0; ch < 3; ch++)
        {
            eps[0][1][ch] = (eps[0][0][ch] + eps[0][1][ch]) & wrapMask;
            if (SIGNED)
                eps[0][1][ch] = signExtend(eps[0][1][ch], aPrec);
            if (partitioned)
            {
                eps[1][0][ch] = (eps[0][0][ch] + eps[1][0][ch]) & wrapMask;
                eps[1][1][ch] = (eps[0][0][ch] + eps[1][1][ch]) & wrapMask;
                if (SIGNED)
                {
                    eps[1][0][ch] = signExtend(eps[1][0][ch], aPrec);
                    eps[1][1][ch] = signExtend(eps[1][1][ch], aPrec);
                }
            }
        }
    }
    // unquantise
    for (int s = 0; s < 2; s++)
        for (int e = 0; e < 2; e++)
            for (int ch = 0; ch < 3; ch++)
            {
                if (s >= numSubsets)
                    continue;
                int v = eps[s][e][ch];
                if (SIGNED)
                {
                    if (aPrec < 16)
                    {
                        const bool neg = v < 0;
                        const int comp = neg ? -v : v;
                        int unq;
                        if (comp == 0)
                            unq = 0;
                        else if (comp >= ((1 << (aPrec - 1)) - 1))
                            unq = 0x7fff;
                        else
                            unq = ((comp << 15) + 0x4000) >> (aPrec - 1);
                        v = neg ? -unq : unq;
                    }
                }
                else
                {
                    if (aPrec < 15 && v != 0)
                        v = (v == ((1 << aPrec) - 1)) ? 0xffff : (((v << 16) + 0x8000) >> aPrec);
                }
                eps[s][e][ch] = v;
            }

    const u32 map2 = T->partition2[partition];
    for (int px = 0; px < 16; px++)
    {
        const int subset = partitioned ? (int)((map2 >> px) & 1u) : 0;
        const int w = bc7Weight(indexBits, idx[px]);
        u32 c[3];
        for (int ch = 0; ch < 3; ch++)
        {
            const int e0 = subset ? eps[1][0][ch] : eps[0][0][ch], e1 = subset ? eps[1][1][ch] : eps[0][1][ch];
            int comp = ((64 - w) * e0 + w * e1 + 32) >> 6;
            if (SIGNED)
            {
                comp = (comp < 0) ? -(((-comp) * 31) >> 5) : ((comp * 31) >> 5);
                u32 sgn = 0;
                if (comp < 0)
                {
                    sgn = 0x8000u;
                    comp = -comp;
                }
                c[ch] = (sgn | (u32)comp) & 0xffffu;
            }
            else
                c[ch] = (u32)((comp * 31) >> 6) & 0xffffu;
        }
        dst[px] = make_uint2(c[0] | (c[1] << 16), c[2] | 0x3c000000u);
    }
}
} // namespace

extern "C" hipError_t cvttmi_launch_decode(const void *d_bc, void *d_out, uint32_t numBlocks, int format,
                                           const CvttDeviceTables *d_tables, hipStream_t stream)
{
    if (numBlocks == 0)
        return hipSuccess;
    const dim3 grid((numBlocks + 63u) / 64u), block(64);
    if (format == 0)
        hipLaunchKernelGGL(cvttmi_decode_bc7_kernel, grid, block, 0, stream, (const uint8_t *)d_bc, (uint8_t *)d_out, numBlocks, d_tables);
    else if (format == 1)
        hipLaunchKernelGGL(cvttmi_decode_bc6h_kernel<false>, grid, block, 0, stream, (const uint8_t *)d_bc, (uint8_t *)d_out, numBlocks, d_tables);
    else
        hipLaunchKernelGGL(cvttmi_decode_bc6h_kernel<true>, grid, block, 0, stream, (const uint8_t *)d_bc, (uint8_t *)d_out, numBlocks, d_tables);
    return hipGetLastError();
}
