// Arithmetic self-test: the kernels promise the reference's SSE2 lane arithmetic (SURVEY.md
// App. A): IEEE-754 binary32 divide and square root, correctly rounded, denormals kept.  This
// kernel evaluates both on pseudo-random bit patterns with exactly the expressions the encoders
// use (operator/ and sqrtExact, same compiler flags); the host compares with DIVSS / SQRTSS.
#include "cvtt_kernel_common.h"

namespace
{
__device__ __forceinline__ u64 mix64(u64 z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void cvttmi_selftest_kernel(u64 seed, u64 first, u32 count, u32 *__restrict__ operands, u32 *__restrict__ results)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    const u64 r = mix64(seed + (first + i) * 0x9E3779B97F4A7C15ull);
    u32 a = (u32)r, b = (u32)(r >> 32);
    // keep the operands finite; every other bit pattern (zeros, denormals, both signs) stays in
    if ((a & 0x7f800000u) == 0x7f800000u) a ^= 0x00800000u;
    if ((b & 0x7f800000u) == 0x7f800000u) b ^= 0x00800000u;
    const float fa = __uint_as_float(a), fb = __uint_as_float(b);
    operands[2 * i] = a;
    operands[2 * i + 1] = b;
    results[2 * i] = __float_as_uint(fa / fb);
    results[2 * i + 1] = __float_as_uint(sqrtExact(__uint_as_float(a & 0x7fffffffu)));
}
} // namespace

extern "C" hipError_t cvttmi_launch_selftest(uint64_t seed, uint64_t first, uint32_t count, void *d_operands, void *d_results,
                                             hipStream_t stream)
{
    if (count == 0)
        return hipSuccess;
    hipLaunchKernelGGL(cvttmi_selftest_kernel, dim3((count + 255u) / 256u), dim3(256), 0, stream, seed, first, count,
                       (u32 *)d_operands, (u32 *)d_results);
    return hipGetLastError();
}
