// C++ face (include/cvtt/ConvectionKernels.h) on top of the C ABI: same names and call
// convention as the reference's cvtt::Kernels (reference ConvectionKernels_API.cpp:41-99,
// 216-286).  One context PER CALLING THREAD on HIP device CVTTMI_DEVICE (default 0), created on the thread's first call
// and destroyed when the thread ends: the reference is called by one worker thread per group of blocks
// (etc2packer.cpp:215-281), and contexts are independent (own stream, staging and work buffers), so such callers run
// side by side instead of queueing on one lock.
//
// The reference's own calls take ONE group of 8 blocks (ConvectionKernels_API.cpp:41-99); on a GPU such a call is a PCIe
// round trip and a one-wave launch, and sixteen caller threads doing that side by side queue on the device.  They are
// therefore coalesced: calls of the same kind (format, Options, plan) that arrive while a launch of that kind is in flight
// wait for it, and the first of them then encodes all of them with ONE launch (coalescer.h: one slot -- context and queue --
// per kind, so callers of different kinds run side by side; the groups are independent, so the bytes are the ones separate
// calls give).  A single caller thread never waits: nothing is in flight when its call arrives.
// CVTTMI_DROPIN_COALESCE=0 turns it off (every call on its thread's own context).
#include "../../include/cvtt/ConvectionKernels.h"
#include "../../include/cvtt_mi355x.h"
#include "coalescer.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

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

    // ---- coalescing of concurrent one-group calls (coalescer.h) ----
    struct CallKey
    {
        int kind;                 // which entry point (and its integer argument)
        cvttmi_options options;
        cvttmi_options allocOptions; // ETC2: the Options of AllocETC2Data
        bool hasAlloc;
        bool hasPlan;
        cvttmi_bc7_plan plan;
        bool same(const CallKey &o) const
        {
            return kind == o.kind && hasAlloc == o.hasAlloc && hasPlan == o.hasPlan && memcmp(&options, &o.options, sizeof(options)) == 0 &&
                   (!hasAlloc || memcmp(&allocOptions, &o.allocOptions, sizeof(allocOptions)) == 0) && (!hasPlan || memcmp(&plan, &o.plan, sizeof(plan)) == 0);
        }
    };

    CallKey makeKey(int kind, const cvtt::Options &o, const cvttmi_options *alloc = NULL, const cvtt::BC7EncodingPlan *plan = NULL)
    {
        CallKey k;
        memset(&k, 0, sizeof(k));
        k.kind = kind;
        memcpy(&k.options, &o, sizeof(k.options));
        k.hasAlloc = alloc != NULL;
        if (alloc)
            k.allocOptions = *alloc;
        k.hasPlan = plan != NULL;
        if (plan)
            memcpy(&k.plan, plan, sizeof(k.plan));
        return k;
    }

    enum Kind { K_BC7, K_BC1, K_BC2, K_BC3, K_BC4U, K_BC4S, K_BC5U, K_BC5S, K_BC6HU, K_BC6HS, K_ETC1, K_ETC2, K_ETC2RGBA, K_ETC2PT, K_ETC2A, K_A11U, K_A11S };
    // encode `n` contiguous blocks with the given key on `ctx`
    int encodeBatch(cvttmi_context *ctx, const CallKey &k, uint8_t *out, const uint8_t *in, size_t n)
    {
        const cvttmi_options *o = &k.options;
        const cvttmi_options *ao = k.hasAlloc ? &k.allocOptions : NULL;
        switch (k.kind)
        {
        case K_BC7: return cvttmi_encode_bc7(ctx, out, in, n, o, &k.plan);
        case K_BC1: return cvttmi_encode_bc1(ctx, out, in, n, o);
        case K_BC2: return cvttmi_encode_bc2(ctx, out, in, n, o);
        case K_BC3: return cvttmi_encode_bc3(ctx, out, in, n, o);
        case K_BC4U: return cvttmi_encode_bc4(ctx, out, in, n, o, 0);
        case K_BC4S: return cvttmi_encode_bc4(ctx, out, in, n, o, 1);
        case K_BC5U: return cvttmi_encode_bc5(ctx, out, in, n, o, 0);
        case K_BC5S: return cvttmi_encode_bc5(ctx, out, in, n, o, 1);
        case K_BC6HU: return cvttmi_encode_bc6h(ctx, out, in, n, o, 0);
        case K_BC6HS: return cvttmi_encode_bc6h(ctx, out, in, n, o, 1);
        case K_ETC1: return cvttmi_encode_etc1(ctx, out, in, n, o);
        case K_ETC2: return cvttmi_encode_etc2_with_data(ctx, out, in, n, o, ao, CVTTMI_ETC2_RGB);
        case K_ETC2RGBA: return cvttmi_encode_etc2_with_data(ctx, out, in, n, o, ao, CVTTMI_ETC2_RGBA);
        case K_ETC2PT: return cvttmi_encode_etc2_with_data(ctx, out, in, n, o, ao, CVTTMI_ETC2_PUNCHTHROUGH);
        case K_ETC2A: return cvttmi_encode_etc2_alpha(ctx, out, in, n, o);
        case K_A11U: return cvttmi_encode_etc2_alpha11(ctx, out, reinterpret_cast<const int16_t *>(in), n, 0, o);
        case K_A11S: return cvttmi_encode_etc2_alpha11(ctx, out, reinterpret_cast<const int16_t *>(in), n, 1, o);
        }
        return CVTTMI_E_INVALID;
    }

    // the device side of a coalescer slot
    void *slotCreate()
    {
        cvttmi_context *c = NULL;
        const char *dev = getenv("CVTTMI_DEVICE");
        const int rc = cvttmi_create(&c, dev ? atoi(dev) : 0);
        if (rc != CVTTMI_OK)
        {
            fprintf(stderr, "cvtt (MI355X): no usable gfx950 device (cvttmi_create = %d); there is no CPU fallback\n", rc);
            abort();
        }
        return c;
    }
    void slotDestroy(void *ctx) { cvttmi_destroy(static_cast<cvttmi_context *>(ctx)); }
    void *slotHostAlloc(void *ctx, size_t bytes)
    {
        void *p = NULL;
        return cvttmi_host_alloc(static_cast<cvttmi_context *>(ctx), &p, bytes) == CVTTMI_OK ? p : NULL;
    }
    void slotHostFree(void *ctx, void *p) { cvttmi_host_free(static_cast<cvttmi_context *>(ctx), p); }
    int slotEncode(void *ctx, const CallKey &key, uint8_t *out, const uint8_t *in, size_t numGroups)
    {
        return encodeBatch(static_cast<cvttmi_context *>(ctx), key, out, in, numGroups * cvtt::NumParallelBlocks);
    }
    typedef cvttmi_dropin::Coalescer<CallKey> DropinCoalescer;
    int envInt(const char *name, int dflt) { return getenv(name) ? atoi(getenv(name)) : dflt; }
    DropinCoalescer &coalescer()
    {
        static const cvttmi_dropin::Backend<CallKey> backend = {slotCreate, slotDestroy, slotHostAlloc, slotHostFree, slotEncode};
        // developer knobs: groups per launch (tests: more callers than a launch takes), gather window
        static DropinCoalescer c(backend, static_cast<size_t>(envInt("CVTTMI_DROPIN_MAX_GROUPS", 256)), envInt("CVTTMI_DROPIN_WINDOW_US", 100),
                                 sizeof(cvtt::PixelBlockF16) * cvtt::NumParallelBlocks, 16 * cvtt::NumParallelBlocks);
        return c;
    }
    bool coalesceEnabled()
    {
        static const bool on = envInt("CVTTMI_DROPIN_COALESCE", 1) != 0;
        return on;
    }

    // one reference-style call: a group of 8 blocks, `inBytes` in, `outBytes` out
    void oneGroup(int kind, uint8_t *pBC, const void *pBlocks, size_t inBytes, size_t outBytes, const cvtt::Options &options, const char *what,
                  const cvttmi_options *alloc = NULL, const cvtt::BC7EncodingPlan *plan = NULL)
    {
        const CallKey key = makeKey(kind, options, alloc, plan);
        if (coalesceEnabled())
        {
            void *ran = NULL;
            const int rc = coalescer().call(key, pBC, static_cast<const uint8_t *>(pBlocks), inBytes, outBytes, &ran);
            if (rc == CVTTMI_OK)
                return;
            if (rc != DropinCoalescer::kNoSlot) // (kNoSlot: more kinds in flight than slots -- this thread's own context below)
            {
                fprintf(stderr, "cvtt (MI355X): %s failed (%d): %s\n", what, rc, ran ? cvttmi_last_error(static_cast<cvttmi_context *>(ran)) : "no staging memory");
                abort();
            }
        }
        check(encodeBatch(context(), key, pBC, static_cast<const uint8_t *>(pBlocks), cvtt::NumParallelBlocks), what);
    }

    // ---- the device list of the *Batch entry points (cvtt_mi355x.h, cvttmi_dropin_set_devices) ----
    const size_t kMultiMinBlocks = 65536; // below this a second device's launch and transfers cost more than they save
    std::mutex g_devMu;
    bool g_devInit = false;
    std::vector<int> g_devList;
    cvttmi_multi *g_multi = NULL;
    // the handle when the list has several entries and the call is large enough, else NULL (single device: context())
    cvttmi_multi *multiFor(size_t numBlocks)
    {
        std::lock_guard<std::mutex> lock(g_devMu);
        if (!g_devInit)
        {
            g_devInit = true;
            if (const char *e = getenv("CVTTMI_DEVICES"))
                for (const char *p = e; *p;)
                {
                    char *end = NULL;
                    const long v = strtol(p, &end, 10);
                    if (end == p)
                        break;
                    g_devList.push_back((int)v);
                    p = (*end == ',') ? end + 1 : end;
                }
        }
        if (g_devList.size() < 2 || numBlocks < kMultiMinBlocks)
            return NULL;
        if (!g_multi)
        {
            const int rc = cvttmi_multi_create(&g_multi, g_devList.data(), (int)g_devList.size());
            if (rc != CVTTMI_OK)
            {
                fprintf(stderr, "cvtt (MI355X): device list not usable (cvttmi_multi_create = %d); there is no CPU fallback\n", rc);
                abort();
            }
        }
        return g_multi;
    }
    void checkMulti(cvttmi_multi *m, int rc, const char *what)
    {
        if (rc != CVTTMI_OK)
        {
            fprintf(stderr, "cvtt (MI355X): %s failed (%d): %s\n", what, rc, cvttmi_multi_last_error(m));
            abort();
        }
    }

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
            if (cvttmi_multi *m = multiFor(numBlocks))
                return checkMulti(m, cvttmi_multi_encode(m, CVTTMI_FMT_BC7, pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, 0, opt(options),
                                                         reinterpret_cast<const cvttmi_bc7_plan *>(&plan)), "EncodeBC7");
            check(cvttmi_encode_bc7(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options),
                                    reinterpret_cast<const cvttmi_bc7_plan *>(&plan)), "EncodeBC7");
        }
        void EncodeBC1Batch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options)
        {
            if (cvttmi_multi *m = multiFor(numBlocks))
                return checkMulti(m, cvttmi_multi_encode(m, CVTTMI_FMT_BC1, pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, 0, opt(options), NULL), "EncodeBC1");
            check(cvttmi_encode_bc1(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options)), "EncodeBC1");
        }
#define CVTT_S3TC_BATCH(NAME, PIXELTYPE, CALL)                                                                                  \
        void NAME##Batch(uint8_t *pBC, const PIXELTYPE *pBlocks, size_t numBlocks, const Options &options)                      \
        {                                                                                                                       \
            check(CALL, #NAME);                                                                                                 \
        }                                                                                                                       \
        void NAME(uint8_t *pBC, const PIXELTYPE *pBlocks, const Options &options) { oneGroup(K_##NAME, pBC, pBlocks, sizeof(PIXELTYPE) * NumParallelBlocks, OUTBYTES * NumParallelBlocks, options, #NAME); }
#define K_EncodeBC2 K_BC2
#define K_EncodeBC3 K_BC3
#define K_EncodeBC4U K_BC4U
#define K_EncodeBC4S K_BC4S
#define K_EncodeBC5U K_BC5U
#define K_EncodeBC5S K_BC5S
#define OUTBYTES 16
        CVTT_S3TC_BATCH(EncodeBC2, PixelBlockU8, cvttmi_encode_bc2(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options)))
        CVTT_S3TC_BATCH(EncodeBC3, PixelBlockU8, cvttmi_encode_bc3(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options)))
#undef OUTBYTES
#define OUTBYTES 8
        CVTT_S3TC_BATCH(EncodeBC4U, PixelBlockU8, cvttmi_encode_bc4(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), 0))
        CVTT_S3TC_BATCH(EncodeBC4S, PixelBlockS8, cvttmi_encode_bc4(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), 1))
#undef OUTBYTES
#define OUTBYTES 16
        CVTT_S3TC_BATCH(EncodeBC5U, PixelBlockU8, cvttmi_encode_bc5(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), 0))
        CVTT_S3TC_BATCH(EncodeBC5S, PixelBlockS8, cvttmi_encode_bc5(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), 1))
#undef CVTT_S3TC_BATCH
#undef OUTBYTES

        void EncodeBC6HUBatch(uint8_t *pBC, const PixelBlockF16 *pBlocks, size_t numBlocks, const Options &options)
        {
            if (cvttmi_multi *m = multiFor(numBlocks))
                return checkMulti(m, cvttmi_multi_encode(m, CVTTMI_FMT_BC6HU, pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, 0, opt(options), NULL), "EncodeBC6HU");
            check(cvttmi_encode_bc6h(context(), pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, opt(options), 0), "EncodeBC6HU");
        }
        void EncodeBC6HSBatch(uint8_t *pBC, const PixelBlockF16 *pBlocks, size_t numBlocks, const Options &options)
        {
            if (cvttmi_multi *m = multiFor(numBlocks))
                return checkMulti(m, cvttmi_multi_encode(m, CVTTMI_FMT_BC6HS, pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, 0, opt(options), NULL), "EncodeBC6HS");
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
            // (sharded only when the scratch was allocated with the Options of this call: the sharded form has one Options argument)
            if (!allocOpt(data) || memcmp(allocOpt(data), &options, sizeof(cvttmi_options)) == 0)
                if (cvttmi_multi *m = multiFor(numBlocks))
                    return checkMulti(m, cvttmi_multi_encode(m, CVTTMI_FMT_ETC2_RGBA, pBC, reinterpret_cast<const uint8_t *>(pBlocks), numBlocks, 0, opt(options), NULL), "EncodeETC2RGBA");
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
            oneGroup(isSigned ? K_A11S : K_A11U, pBC, pBlocks, sizeof(PixelBlockScalarS16) * NumParallelBlocks, 8 * NumParallelBlocks, options, "EncodeETC2Alpha11");
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

        // The reference's own entry points: one group of NumParallelBlocks blocks per call (coalesced across caller threads)
        void EncodeBC7(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, const BC7EncodingPlan &plan) { oneGroup(K_BC7, pBC, pBlocks, 512, 128, options, "EncodeBC7", NULL, &plan); }
        void EncodeBC1(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options) { oneGroup(K_BC1, pBC, pBlocks, 512, 64, options, "EncodeBC1"); }
        void EncodeBC6HU(uint8_t *pBC, const PixelBlockF16 *pBlocks, const Options &options) { oneGroup(K_BC6HU, pBC, pBlocks, 1024, 128, options, "EncodeBC6HU"); }
        void EncodeBC6HS(uint8_t *pBC, const PixelBlockF16 *pBlocks, const Options &options) { oneGroup(K_BC6HS, pBC, pBlocks, 1024, 128, options, "EncodeBC6HS"); }
        void EncodeETC2PunchthroughAlpha(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, ETC2CompressionData *data) { oneGroup(K_ETC2PT, pBC, pBlocks, 512, 64, options, "EncodeETC2PunchthroughAlpha", allocOpt(data)); }
        void EncodeETC1(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, ETC1CompressionData *) { oneGroup(K_ETC1, pBC, pBlocks, 512, 64, options, "EncodeETC1"); }
        void EncodeETC2(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, ETC2CompressionData *data) { oneGroup(K_ETC2, pBC, pBlocks, 512, 64, options, "EncodeETC2", allocOpt(data)); }
        void EncodeETC2RGBA(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, ETC2CompressionData *data) { oneGroup(K_ETC2RGBA, pBC, pBlocks, 512, 128, options, "EncodeETC2RGBA", allocOpt(data)); }
        void EncodeETC2Alpha(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options) { oneGroup(K_ETC2A, pBC, pBlocks, 512, 64, options, "EncodeETC2Alpha"); }

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

extern "C" int cvttmi_dropin_set_devices(const int *devices, int numDevices)
{
    if (numDevices < 0 || (numDevices > 0 && !devices))
        return CVTTMI_E_INVALID;
    std::lock_guard<std::mutex> lock(g_devMu);
    g_devInit = true;
    g_devList.assign(devices, devices + numDevices);
    if (g_multi)
    {
        cvttmi_multi_destroy(g_multi);
        g_multi = NULL;
    }
    return CVTTMI_OK;
}
