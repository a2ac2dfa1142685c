// Image -> PixelBlock tiling and packed-row compaction on the device.
//
// What it replaces: the caller-side loop that cuts a linear image into groups of eight
// horizontally adjacent 4x4 blocks with edge clamping (reference etc2packer/etc2packer.cpp:
// 215-247: clampedX = min(x, w-1), clampedY = min(y, h-1); the blocks that pad the last group
// of a block row are encoded but not written, 275-281).  Pure HBM copies: one thread moves one
// block row of four pixels (16 B of RGBA8, 32 B of RGBA16F), neighbouring threads take
// neighbouring blocks, so reads are contiguous along the image row.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace
{
template <int BYTES_PER_PIXEL>
__global__ void cvttmi_tile_kernel(const uint8_t *__restrict__ image, uint8_t *__restrict__ blocks, uint32_t width,
                                   uint32_t height, size_t rowPitch, uint32_t blocksPerRow, uint32_t blockRows)
{
    const uint32_t bx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t subY = threadIdx.y;
    const uint32_t by = blockIdx.y;
    if (bx >= blocksPerRow || by >= blockRows)
        return;
    const uint32_t y = min(by * 4u + subY, height - 1u);
    const uint8_t *row = image + (size_t)y * rowPitch;
    uint8_t *dst = blocks + ((size_t)by * blocksPerRow + bx) * (16u * BYTES_PER_PIXEL) + subY * (4u * BYTES_PER_PIXEL);
    typedef typename std::conditional<BYTES_PER_PIXEL == 4, uint32_t, uint2>::type Pixel;
    const Pixel *src = reinterpret_cast<const Pixel *>(row);
    Pixel px[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        px[i] = src[min(bx * 4u + (uint32_t)i, width - 1u)];
    Pixel *out = reinterpret_cast<Pixel *>(dst);
#pragma unroll
    for (int i = 0; i < 4; i++)
        out[i] = px[i];
}

// keep the first `realPerRow` packed blocks of every padded block row
__global__ void cvttmi_compact_rows_kernel(const uint8_t *__restrict__ packed, uint8_t *__restrict__ out, uint32_t bytesPerBlock,
                                           uint32_t blocksPerRow, uint32_t realPerRow, uint32_t blockRows)
{
    const uint32_t words = bytesPerBlock / 4u;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; // word inside the real part of a row
    const uint32_t by = blockIdx.y;
    if (i >= realPerRow * words || by >= blockRows)
        return;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(packed) + (size_t)by * blocksPerRow * words;
    uint32_t *dst = reinterpret_cast<uint32_t *>(out) + (size_t)by * realPerRow * words;
    dst[i] = src[i];
}
} // namespace

extern "C" hipError_t cvttmi_launch_tile(const void *d_image, void *d_blocks, uint32_t width, uint32_t height, size_t rowPitch,
                                         uint32_t bytesPerPixel, hipStream_t stream)
{
    const uint32_t realPerRow = (width + 3u) / 4u;
    const uint32_t blocksPerRow = (realPerRow + 7u) / 8u * 8u;
    const uint32_t blockRows = (height + 3u) / 4u;
    if (blockRows == 0 || realPerRow == 0)
        return hipSuccess;
    const dim3 block(64, 4);
    const dim3 grid((blocksPerRow + 63u) / 64u, blockRows);
    if (bytesPerPixel == 4)
        hipLaunchKernelGGL(cvttmi_tile_kernel<4>, grid, block, 0, stream, (const uint8_t *)d_image, (uint8_t *)d_blocks, width,
                           height, rowPitch, blocksPerRow, blockRows);
    else
        hipLaunchKernelGGL(cvttmi_tile_kernel<8>, grid, block, 0, stream, (const uint8_t *)d_image, (uint8_t *)d_blocks, width,
                           height, rowPitch, blocksPerRow, blockRows);
    return hipGetLastError();
}

extern "C" hipError_t cvttmi_launch_compact_rows(const void *d_packed, void *d_out, uint32_t width, uint32_t height,
                                                 uint32_t bytesPerBlock, hipStream_t stream)
{
    const uint32_t realPerRow = (width + 3u) / 4u;
    const uint32_t blocksPerRow = (realPerRow + 7u) / 8u * 8u;
    const uint32_t blockRows = (height + 3u) / 4u;
    if (blockRows == 0 || realPerRow == 0)
        return hipSuccess;
    const uint32_t words = realPerRow * (bytesPerBlock / 4u);
    hipLaunchKernelGGL(cvttmi_compact_rows_kernel, dim3((words + 255u) / 256u, blockRows), dim3(256), 0, stream,
                       (const uint8_t *)d_packed, (uint8_t *)d_out, bytesPerBlock, blocksPerRow, realPerRow, blockRows);
    return hipGetLastError();
}
