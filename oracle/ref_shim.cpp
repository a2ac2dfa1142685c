// TEST INFRASTRUCTURE ONLY -- never linked or imported by the product path.
//
// Thin C-ABI harness around the *unmodified* reference library.  It is compiled
// together with /root/reference/ConvectionKernels_SingleFile.cpp (sources stay
// where they are; see oracle/Makefile) into oracle/_ref/libcvtt_ref.so, which is
// git-ignored.  Tests use it to pin the C restatement in oracle/cvtt_oracle.c and
// bench.py may time it as the "reference" CPU baseline.
//
// Entry points take flat arrays of blocks; group g = blocks [8g, 8g+8), exactly
// what one call of cvtt::Kernels::Encode* consumes (ConvectionKernels.h:241).
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "ConvectionKernels.h"
#include "ConvectionKernels_BC6H_IO.h"

namespace
{
    void *shimAlloc(void *, size_t size)
    {
        void *p = NULL;
        if (posix_memalign(&p, 64, size) != 0)
            return NULL;
        // canonical oracle: scratch starts zeroed (hazard H2)
        memset(p, 0, size);
        return p;
    }

    void shimFree(void *, void *ptr, size_t)
    {
        free(ptr);
    }

    cvtt::Options makeOptions(uint32_t flags, const float *weights4, float threshold, int refineBC7, int refineBC6H, int refineS3TC, int seedPoints)
    {
        cvtt::Options o;
        o.flags = flags;
        o.threshold = threshold;
        o.redWeight = weights4[0];
        o.greenWeight = weights4[1];
        o.blueWeight = weights4[2];
        o.alphaWeight = weights4[3];
        o.refineRoundsBC7 = refineBC7;
        o.refineRoundsBC6H = refineBC6H;
        o.refineRoundsS3TC = refineS3TC;
        o.seedPoints = seedPoints;
        return o;
    }
}

extern "C"
{
    // sizeof checks used by the boundary tests
    size_t ref_sizeof_options() { return sizeof(cvtt::Options); }
    size_t ref_sizeof_bc7_plan() { return sizeof(cvtt::BC7EncodingPlan); }
    size_t ref_sizeof_bc7_finetune() { return sizeof(cvtt::BC7FineTuningParams); }

    // Default-constructed PODs, as raw bytes.
    void ref_default_options(void *out) { cvtt::Options o; memcpy(out, &o, sizeof(o)); }
    void ref_default_bc7_plan(void *out) { cvtt::BC7EncodingPlan p; memset(out, 0, sizeof(p)); memcpy(out, &p, sizeof(p)); }
    void ref_bc7_plan_from_quality(void *out, int quality)
    {
        cvtt::BC7EncodingPlan p;
        cvtt::Kernels::ConfigureBC7EncodingPlanFromQuality(p, quality);
        memcpy(out, &p, sizeof(p));
    }
    void ref_bc7_plan_from_finetune(void *out, const void *params)
    {
        cvtt::BC7EncodingPlan p;
        cvtt::BC7FineTuningParams ft;
        memcpy(&ft, params, sizeof(ft));
        cvtt::Kernels::ConfigureBC7EncodingPlanFromFineTuningParams(p, ft);
        memcpy(out, &p, sizeof(p));
    }

    // rcpps of 0..16 on this host (index 0 unused) -- SURVEY App. A.
    void ref_probe_rcp(float *out17)
    {
        for (int i = 0; i <= 16; i++)
        {
            float v = (float)(i == 0 ? 1 : i);
            __m128 r = _mm_rcp_ps(_mm_set1_ps(v));
            out17[i] = _mm_cvtss_f32(r);
        }
    }

    void ref_encode_bc7(uint8_t *out, const uint8_t *blocks, size_t numBlocks, const void *optionsBytes, const void *planBytes)
    {
        cvtt::Options o;
        memcpy(&o, optionsBytes, sizeof(o));
        cvtt::BC7EncodingPlan plan;
        memcpy(&plan, planBytes, sizeof(plan));
        const cvtt::PixelBlockU8 *in = reinterpret_cast<const cvtt::PixelBlockU8 *>(blocks);
        for (size_t b = 0; b + cvtt::NumParallelBlocks <= numBlocks; b += cvtt::NumParallelBlocks)
            cvtt::Kernels::EncodeBC7(out + b * 16, in + b, o, plan);
    }

    void ref_encode_bc1(uint8_t *out, const uint8_t *blocks, size_t numBlocks, const void *optionsBytes)
    {
        cvtt::Options o;
        memcpy(&o, optionsBytes, sizeof(o));
        const cvtt::PixelBlockU8 *in = reinterpret_cast<const cvtt::PixelBlockU8 *>(blocks);
        for (size_t b = 0; b + cvtt::NumParallelBlocks <= numBlocks; b += cvtt::NumParallelBlocks)
            cvtt::Kernels::EncodeBC1(out + b * 8, in + b, o);
    }

    void ref_encode_bc6h(uint8_t *out, const uint8_t *blocks, size_t numBlocks, const void *optionsBytes, int isSigned)
    {
        cvtt::Options o;
        memcpy(&o, optionsBytes, sizeof(o));
        const cvtt::PixelBlockF16 *in = reinterpret_cast<const cvtt::PixelBlockF16 *>(blocks);
        for (size_t b = 0; b + cvtt::NumParallelBlocks <= numBlocks; b += cvtt::NumParallelBlocks)
        {
            if (isSigned)
                cvtt::Kernels::EncodeBC6HS(out + b * 16, in + b, o);
            else
                cvtt::Kernels::EncodeBC6HU(out + b * 16, in + b, o);
        }
    }

    // mode: 0 = ETC2 RGB (8 B/block), 1 = ETC2 RGBA (16 B/block), 2 = EAC alpha only (8 B/block), 3 = ETC1 (8 B/block), 4 = ETC2 punch-through alpha (8 B/block)
    int ref_encode_etc2_alloc(uint8_t *out, const uint8_t *blocks, size_t numBlocks, const void *optionsBytes, const void *allocOptionsBytes, int mode);
    int ref_encode_etc2(uint8_t *out, const uint8_t *blocks, size_t numBlocks, const void *optionsBytes, int mode)
    {
        return ref_encode_etc2_alloc(out, blocks, numBlocks, optionsBytes, optionsBytes, mode);
    }
    // allocOptionsBytes: the Options handed to AllocETC2Data (may differ from those of the Encode calls)
    int ref_encode_etc2_alloc(uint8_t *out, const uint8_t *blocks, size_t numBlocks, const void *optionsBytes, const void *allocOptionsBytes, int mode)
    {
        cvtt::Options o, ao;
        memcpy(&o, optionsBytes, sizeof(o));
        memcpy(&ao, allocOptionsBytes, sizeof(ao));
        const cvtt::PixelBlockU8 *in = reinterpret_cast<const cvtt::PixelBlockU8 *>(blocks);
        if (mode == 3)
        {
            cvtt::ETC1CompressionData *data1 = cvtt::Kernels::AllocETC1Data(shimAlloc, NULL);
            if (!data1)
                return -1;
            for (size_t b = 0; b + cvtt::NumParallelBlocks <= numBlocks; b += cvtt::NumParallelBlocks)
                cvtt::Kernels::EncodeETC1(out + b * 8, in + b, o, data1);
            cvtt::Kernels::ReleaseETC1Data(data1, shimFree);
            return 0;
        }
        cvtt::ETC2CompressionData *data = NULL;
        if (mode != 2)
        {
            data = cvtt::Kernels::AllocETC2Data(shimAlloc, NULL, ao);
            if (!data)
                return -1;
        }
        for (size_t b = 0; b + cvtt::NumParallelBlocks <= numBlocks; b += cvtt::NumParallelBlocks)
        {
            if (mode == 0)
                cvtt::Kernels::EncodeETC2(out + b * 8, in + b, o, data);
            else if (mode == 1)
                cvtt::Kernels::EncodeETC2RGBA(out + b * 16, in + b, o, data);
            else if (mode == 4)
                cvtt::Kernels::EncodeETC2PunchthroughAlpha(out + b * 8, in + b, o, data);
            else
                cvtt::Kernels::EncodeETC2Alpha(out + b * 8, in + b, o);
        }
        if (data)
            cvtt::Kernels::ReleaseETC2Data(data, shimFree);
        return 0;
    }

    // BC6H header bit scatter of mode `modeIndex` (0..13, order of the BC6H mode table):
    // fields = { m, d, rw, rx, ry, rz, gw, gx, gy, gz, bw, bx, by, bz }.  Used by
    // tools/gen_bc6h_layout.py to tabulate the (public) BC6H bit layout by probing.
    void ref_bc6h_write_header(int modeIndex, const uint16_t *f, uint32_t *out3)
    {
        out3[0] = out3[1] = out3[2] = 0;
        cvtt::BC6H_IO::g_writeFuncs[modeIndex](out3, f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8], f[9], f[10], f[11], f[12], f[13]);
    }

    // format: 2 = BC2, 3 = BC3, 4 = BC4U, 5 = BC4S, 6 = BC5U, 7 = BC5S
    void ref_encode_s3tc(uint8_t *out, const uint8_t *blocks, size_t numBlocks, const void *optionsBytes, int format)
    {
        cvtt::Options o;
        memcpy(&o, optionsBytes, sizeof(o));
        const cvtt::PixelBlockU8 *in = reinterpret_cast<const cvtt::PixelBlockU8 *>(blocks);
        const cvtt::PixelBlockS8 *ins = reinterpret_cast<const cvtt::PixelBlockS8 *>(blocks);
        const size_t per = (format == 4 || format == 5) ? 8 : 16;
        for (size_t b = 0; b + cvtt::NumParallelBlocks <= numBlocks; b += cvtt::NumParallelBlocks)
        {
            uint8_t *dst = out + b * per;
            switch (format)
            {
            case 2: cvtt::Kernels::EncodeBC2(dst, in + b, o); break;
            case 3: cvtt::Kernels::EncodeBC3(dst, in + b, o); break;
            case 4: cvtt::Kernels::EncodeBC4U(dst, in + b, o); break;
            case 5: cvtt::Kernels::EncodeBC4S(dst, ins + b, o); break;
            case 6: cvtt::Kernels::EncodeBC5U(dst, in + b, o); break;
            default: cvtt::Kernels::EncodeBC5S(dst, ins + b, o); break;
            }
        }
    }

    void ref_encode_eac11(uint8_t *out, const int16_t *blocksS16, size_t numBlocks, const void *optionsBytes, int isSigned)
    {
        cvtt::Options o;
        memcpy(&o, optionsBytes, sizeof(o));
        const cvtt::PixelBlockScalarS16 *in = reinterpret_cast<const cvtt::PixelBlockScalarS16 *>(blocksS16);
        for (size_t b = 0; b + cvtt::NumParallelBlocks <= numBlocks; b += cvtt::NumParallelBlocks)
            cvtt::Kernels::EncodeETC2Alpha11(out + b * 8, in + b, isSigned != 0, o);
    }

    void ref_decode_bc6h(uint8_t *outBlocksF16, const uint8_t *bc, size_t numBlocks, int isSigned)
    {
        cvtt::PixelBlockF16 *o = reinterpret_cast<cvtt::PixelBlockF16 *>(outBlocksF16);
        for (size_t b = 0; b + cvtt::NumParallelBlocks <= numBlocks; b += cvtt::NumParallelBlocks)
        {
            if (isSigned)
                cvtt::Kernels::DecodeBC6HS(o + b, bc + b * 16);
            else
                cvtt::Kernels::DecodeBC6HU(o + b, bc + b * 16);
        }
    }

    void ref_decode_bc7(uint8_t *outBlocks, const uint8_t *bc, size_t numBlocks)
    {
        cvtt::PixelBlockU8 *o = reinterpret_cast<cvtt::PixelBlockU8 *>(outBlocks);
        for (size_t b = 0; b + cvtt::NumParallelBlocks <= numBlocks; b += cvtt::NumParallelBlocks)
            cvtt::Kernels::DecodeBC7(o + b, bc + b * 16);
    }
    // Time-bounded multi-threaded run for bench.py's cpu_baseline: `numThreads` std::threads claim chunks of
    // `chunkBlocks` blocks (a multiple of 8) in ascending order -- the reference's own one-worker-per-group caller
    // pattern (etc2packer.cpp:215-281) -- until the input is exhausted or `budgetSeconds` have passed; every claimed chunk
    // is finished, so on return exactly the prefix [0, *blocksDone) of `out` is valid.
    // format: 0 = BC7 (plan used), 1 = BC1, 2 = BC6HU, 3 = BC6HS, 4 = ETC2 RGB, 5 = ETC2 RGBA.
    int ref_encode_mt(int format, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const void *optionsBytes, const void *planBytes,
                      int numThreads, double budgetSeconds, size_t chunkBlocks, uint64_t *blocksDone, double *seconds)
    {
        if (format < 0 || format > 5 || numThreads < 1 || chunkBlocks == 0 || chunkBlocks % cvtt::NumParallelBlocks != 0)
            return -1;
        cvtt::Options o;
        memcpy(&o, optionsBytes, sizeof(o));
        cvtt::BC7EncodingPlan plan;
        if (planBytes)
            memcpy(&plan, planBytes, sizeof(plan));
        const size_t inBytes = (format == 2 || format == 3) ? 128 : 64;
        const size_t outBytes = (format == 1 || format == 4) ? 8 : 16;
        const size_t numChunks = numBlocks / chunkBlocks;
        std::atomic<size_t> next(0);
        std::atomic<int> failed(0);
        const auto t0 = std::chrono::steady_clock::now();
        auto worker = [&]() {
            cvtt::ETC2CompressionData *etc = NULL;
            if (format >= 4)
            {
                etc = cvtt::Kernels::AllocETC2Data(shimAlloc, NULL, o);
                if (!etc)
                {
                    failed = 1;
                    return;
                }
            }
            for (;;)
            {
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > budgetSeconds)
                    break;
                const size_t c = next.fetch_add(1);
                if (c >= numChunks)
                    break;
                for (size_t b = c * chunkBlocks; b < (c + 1) * chunkBlocks; b += cvtt::NumParallelBlocks)
                {
                    const uint8_t *in = blocks + b * inBytes;
                    uint8_t *dst = out + b * outBytes;
                    switch (format)
                    {
                    case 0: cvtt::Kernels::EncodeBC7(dst, reinterpret_cast<const cvtt::PixelBlockU8 *>(in), o, plan); break;
                    case 1: cvtt::Kernels::EncodeBC1(dst, reinterpret_cast<const cvtt::PixelBlockU8 *>(in), o); break;
                    case 2: cvtt::Kernels::EncodeBC6HU(dst, reinterpret_cast<const cvtt::PixelBlockF16 *>(in), o); break;
                    case 3: cvtt::Kernels::EncodeBC6HS(dst, reinterpret_cast<const cvtt::PixelBlockF16 *>(in), o); break;
                    case 4: cvtt::Kernels::EncodeETC2(dst, reinterpret_cast<const cvtt::PixelBlockU8 *>(in), o, etc); break;
                    default: cvtt::Kernels::EncodeETC2RGBA(dst, reinterpret_cast<const cvtt::PixelBlockU8 *>(in), o, etc); break;
                    }
                }
            }
            if (etc)
                cvtt::Kernels::ReleaseETC2Data(etc, shimFree);
        };
        std::vector<std::thread> threads;
        for (int t = 1; t < numThreads; t++)
            threads.emplace_back(worker);
        worker();
        for (size_t t = 0; t < threads.size(); t++)
            threads[t].join();
        *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const size_t claimed = next.load() < numChunks ? next.load() : numChunks;
        *blocksDone = (uint64_t)(claimed * chunkBlocks);
        return failed.load() ? -2 : 0;
    }
}
