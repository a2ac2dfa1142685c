// C++ face (include/cvtt/ConvectionKernels.h) on top of the C ABI: same names and call
// convention as the reference's cvtt::Kernels (reference ConvectionKernels_API.cpp:41-99,
// 216-286).  One context PER CALLING THREAD on HIP device CVTTMI_DEVICE (default 0), created on the thread's first call
// and destroyed when the thread ends: the reference is called by one worker thread per group of blocks
// (etc2packer.cpp:215-281), and contexts are independent (own stream, staging and work buffers), so such callers run
// side by side instead of queueing on one lock.
#include "../../include/cvtt/ConvectionKernels.h"
#include "../../include/cvtt_mi355x.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

static_assert(sizeof(cvtt::Options) == sizeof(cvttmi_options), "cvtt::Options layout");
static_assert(sizeof(cvtt::BC7EncodingPlan) == sizeof(cvttmi_bc7_plan), "cvtt::BC7EncodingPlan layout");
static_assert(sizeof(cvtt::BC7FineTuningParams) == sizeof(cvttmi_bc7_fine_tuning), "cvtt::BC7FineTuningParams layout");
static_assert(sizeof(cvtt::PixelBlockU8) == 64 && sizeof(cvtt::PixelBlockF16) == 128, "pixel block layout");

namespace
{
    struct ThreadContext
    {
        cvttmi_context *ctx;
        ThreadContext() : ctx(NULL) {}
        ~ThreadContext()
        {
            if (ctx)
                cvttmi_destroy(ctx);
        }
    };
    thread_local ThreadContext t_ctx;

    cvttmi_context *context()
    {
        if (!t_ctx.ctx)
        {
            const char *dev = getenv("CVTTMI_DEVICE");
            const int rc = cvttmi_create(&t_ctx.ctx, dev ? atoi(dev) : 0);
            if (rc != CVTTMI_OK)
            {
                fprintf(stderr, "cvtt (MI355X): no usable gfx950 device (cvttmi_create = %d); there is no CPU fallback\n", rc);
                abort();
            }
        }
        return t_ctx.ctx;
    }

    void check(int rc, const char *what)
    {
        if (rc != CVTTMI_OK)
        {
            fprintf(stderr, "cvtt (MI355X): %s failed (%d): %s\n", what, rc, cvttmi_last_error(t_ctx.ctx));
            abort();
        }
    }

    const cvttmi_options *opt(const cvtt::Options &o) { return reinterpret_cast<const cvttmi_options *>(&o); }

    // What survives of the reference's 136 KB ETC2 scratch: the allocator context (for ReleaseETC2Data) and the Options of
    // the allocation -- the reference derives the chroma axes of the sector split from THOSE (ETC.cpp:3117-3145), and from
    // the Options of the Encode call everything else.
    struct Etc2Marker : public cvtt::ETC2CompressionData
    {
        void *context;
        cvttmi_options allocOptions;
    };
    const cvttmi_options *allocOpt(cvtt::ETC2CompressionData *data) { return data ? &static_cast<Etc2Marker *>(data)->allocOptions : NULL; }
    struct Etc1Marker : public cvtt::ETC1CompressionData
    {
        void *context;
    };
}

cvtt::Options::Options()
{
    cvttmi_default_options(reinterpret_cast<cvttmi_options *>(this));
}

cvtt::BC7EncodingPlan::BC7EncodingPlan()
{
    cvttmi_default_bc7_plan(reinterpret_cast<cvttmi_bc7_plan *>(this));
}

cvtt::BC7FineTuningParams::BC7FineTuningParams()
{
    cvttmi_default_bc7_fine_tuning(reinterpret_cast<cvttmi_bc7_fine_tuning *>(this));
}

namespace cvtt
{
    namespace Kernels
    {
        void ConfigureBC7EncodingPlanFromQuality(BC7EncodingPlan &encodingPlan, int quality)
        {
            cvttmi_bc7_plan_from_quality(reinterpret_cast<cvttmi_bc7_plan *>(&encodingPlan), quality);
        }

        bool ConfigureBC7EncodingPlanFromFineTuningParams(BC7EncodingPlan &encodingPlan, const BC7FineTuningParams &params)
        {
            cvttmi_bc7_plan_from_fine_tuning(reinterpret_cast<cvttmi_bc7_plan *>(&encodingPlan), reinterpret_cast<const cvttmi_bc7_fine_tuning *>(&params));
            return true; // like the reference (BC67.cpp:3482)
        }

        void EncodeBC7Batch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options, const BC7EncodingPlan &plan)
        {
            check(cvttmi_encode_bc7(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options),
                                    reinterpret_cast<const cvttmi_bc7_plan *>(&plan)), "EncodeBC7");
        }
        void EncodeBC1Batch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options)
        {
            check(cvttmi_encode_bc1(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options)), "EncodeBC1");
        }
#define CVTT_S3TC_BATCH(NAME, PIXELTYPE, CALL)                                                                                  \
        void NAME##Batch(uint8_t *pBC, const PIXELTYPE *pBlocks, size_t numBlocks, const Options &options)                      \
        {                                                                                                                       \
            check(CALL, #NAME);                                                                                                 \
        }                                                                                                                       \
        void NAME(uint8_t *pBC, const PIXELTYPE *pBlocks, const Options &options) { NAME##Batch(pBC, pBlocks, NumParallelBlocks, options); }
        CVTT_S3TC_BATCH(EncodeBC2, PixelBlockU8, cvttmi_encode_bc2(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options)))
        CVTT_S3TC_BATCH(EncodeBC3, PixelBlockU8, cvttmi_encode_bc3(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options)))
        CVTT_S3TC_BATCH(EncodeBC4U, PixelBlockU8, cvttmi_encode_bc4(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), 0))
        CVTT_S3TC_BATCH(EncodeBC4S, PixelBlockS8, cvttmi_encode_bc4(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), 1))
        CVTT_S3TC_BATCH(EncodeBC5U, PixelBlockU8, cvttmi_encode_bc5(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), 0))
        CVTT_S3TC_BATCH(EncodeBC5S, PixelBlockS8, cvttmi_encode_bc5(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), 1))
#undef CVTT_S3TC_BATCH

        void EncodeBC6HUBatch(uint8_t *pBC, const PixelBlockF16 *pBlocks, size_t numBlocks, const Options &options)
        {
            check(cvttmi_encode_bc6h(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), 0), "EncodeBC6HU");
        }
        void EncodeBC6HSBatch(uint8_t *pBC, const PixelBlockF16 *pBlocks, size_t numBlocks, const Options &options)
        {
            check(cvttmi_encode_bc6h(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), 1), "EncodeBC6HS");
        }
        void EncodeETC2PunchthroughAlphaBatch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options, ETC2CompressionData *data)
        {
            check(cvttmi_encode_etc2_with_data(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), allocOpt(data), CVTTMI_ETC2_PUNCHTHROUGH),
                  "EncodeETC2PunchthroughAlpha");
        }
        void EncodeETC1Batch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options)
        {
            check(cvttmi_encode_etc1(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options)), "EncodeETC1");
        }
        void EncodeETC2Batch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options, ETC2CompressionData *data)
        {
            check(cvttmi_encode_etc2_with_data(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), allocOpt(data), CVTTMI_ETC2_RGB), "EncodeETC2");
        }
        void EncodeETC2RGBABatch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options, ETC2CompressionData *data)
        {
            check(cvttmi_encode_etc2_with_data(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), allocOpt(data), CVTTMI_ETC2_RGBA), "EncodeETC2RGBA");
        }
        void EncodeETC2AlphaBatch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options)
        {
            check(cvttmi_encode_etc2_alpha(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options)), "EncodeETC2Alpha");
        }

        void EncodeETC2Alpha11Batch(uint8_t *pBC, const PixelBlockScalarS16 *pBlocks, size_t numBlocks, bool isSigned, const Options &options)
        {
            check(cvttmi_encode_etc2_alpha11(context(), pBC, reinterpret_cast<const int16_t *>(pBlocks), numBlocks, isSigned ? 1 : 0, opt(options)),
                  "EncodeETC2Alpha11");
        }
        void EncodeETC2Alpha11(uint8_t *pBC, const PixelBlockScalarS16 *pBlocks, bool isSigned, const Options &options)
        {
            EncodeETC2Alpha11Batch(pBC, pBlocks, NumParallelBlocks, isSigned, options);
        }
        void DecodeBC7Batch(PixelBlockU8 *pBlocks, const uint8_t *pBC, size_t numBlocks)
        {
            check(cvttmi_decode_bc7(context(), reinterpret_cast<uint8_t *>(pBlocks), pBC, numBlocks), "DecodeBC7");
        }
        void DecodeBC6HUBatch(PixelBlockF16 *pBlocks, const uint8_t *pBC, size_t numBlocks)
        {
            check(cvttmi_decode_bc6h(context(), reinterpret_cast<uint8_t *>(pBlocks), pBC, numBlocks, 0), "DecodeBC6HU");
        }
        void DecodeBC6HSBatch(PixelBlockF16 *pBlocks, const uint8_t *pBC, size_t numBlocks)
        {
            check(cvttmi_decode_bc6h(context(), reinterpret_cast<uint8_t *>(pBlocks), pBC, numBlocks, 1), "DecodeBC6HS");
        }
        void DecodeBC7(PixelBlockU8 *pBlocks, const uint8_t *pBC) { DecodeBC7Batch(pBlocks, pBC, NumParallelBlocks); }
        void DecodeBC6HU(PixelBlockF16 *pBlocks, const uint8_t *pBC) { DecodeBC6HUBatch(pBlocks, pBC, NumParallelBlocks); }
        void DecodeBC6HS(PixelBlockF16 *pBlocks, const uint8_t *pBC) { DecodeBC6HSBatch(pBlocks, pBC, NumParallelBlocks); }

        void EncodeBC7(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, const BC7EncodingPlan &plan) { EncodeBC7Batch(pBC, pBlocks, NumParallelBlocks, options, plan); }
        void EncodeBC1(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options) { EncodeBC1Batch(pBC, pBlocks, NumParallelBlocks, options); }
        void EncodeBC6HU(uint8_t *pBC, const PixelBlockF16 *pBlocks, const Options &options) { EncodeBC6HUBatch(pBC, pBlocks, NumParallelBlocks, options); }
        void EncodeBC6HS(uint8_t *pBC, const PixelBlockF16 *pBlocks, const Options &options) { EncodeBC6HSBatch(pBC, pBlocks, NumParallelBlocks, options); }
        void EncodeETC2PunchthroughAlpha(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, ETC2CompressionData *data) { EncodeETC2PunchthroughAlphaBatch(pBC, pBlocks, NumParallelBlocks, options, data); }
        void EncodeETC1(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, ETC1CompressionData *) { EncodeETC1Batch(pBC, pBlocks, NumParallelBlocks, options); }
        void EncodeETC2(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, ETC2CompressionData *data) { EncodeETC2Batch(pBC, pBlocks, NumParallelBlocks, options, data); }
        void EncodeETC2RGBA(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, ETC2CompressionData *data) { EncodeETC2RGBABatch(pBC, pBlocks, NumParallelBlocks, options, data); }
        void EncodeETC2Alpha(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options) { EncodeETC2AlphaBatch(pBC, pBlocks, NumParallelBlocks, options); }

        // The reference places 136 KB of scratch in caller memory (ETC.cpp:3100-3115); here the allocator context survives
        // so that ReleaseETC2Data can hand the block back, and the Options, whose colour weights fix the chroma axes.
        ETC2CompressionData *AllocETC2Data(allocFunc_t allocFunc, void *ctx, const Options &options)
        {
            void *buffer = allocFunc(ctx, sizeof(Etc2Marker));
            if (!buffer)
                return NULL;
            Etc2Marker *m = new (buffer) Etc2Marker();
            m->context = ctx;
            memcpy(&m->allocOptions, &options, sizeof(m->allocOptions));
            return m;
        }
        void ReleaseETC2Data(ETC2CompressionData *data, freeFunc_t freeFunc)
        {
            Etc2Marker *m = static_cast<Etc2Marker *>(data);
            void *ctx = m->context;
            freeFunc(ctx, data, sizeof(Etc2Marker));
        }
        ETC1CompressionData *AllocETC1Data(allocFunc_t allocFunc, void *ctx) // ETC.cpp:3083-3098
        {
            void *buffer = allocFunc(ctx, sizeof(Etc1Marker));
            if (!buffer)
                return NULL;
            Etc1Marker *m = new (buffer) Etc1Marker();
            m->context = ctx;
            return m;
        }
        void ReleaseETC1Data(ETC1CompressionData *data, freeFunc_t freeFunc)
        {
            Etc1Marker *m = static_cast<Etc1Marker *>(data);
            void *ctx = m->context;
            freeFunc(ctx, data, sizeof(Etc1Marker));
        }
    }
}
