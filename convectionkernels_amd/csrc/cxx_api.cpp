// C++ face (include/cvtt/ConvectionKernels.h) on top of the C ABI: same names and call
// convention as the reference's cvtt::Kernels (reference ConvectionKernels_API.cpp:41-99,
// 216-286).  One context PER CALLING THREAD on HIP device CVTTMI_DEVICE (default 0), created on the thread's first call
// and destroyed when the thread ends: the reference is called by one worker thread per group of blocks
// (etc2packer.cpp:215-281), and contexts are independent (own stream, staging and work buffers), so such callers run
// side by side instead of queueing on one lock.
//
// The reference's own calls take ONE group of 8 blocks (ConvectionKernels_API.cpp:41-99); on a GPU such a call is a PCIe
// round trip and a one-wave launch, and sixteen caller threads doing that side by side queue on the device.  They are
// therefore coalesced: calls of the same kind (format, Options, plan) that arrive while a launch is in flight wait for it,
// and the first of them then encodes all of them with ONE launch on a shared context (`Coalescer` below; the groups are
// independent, so the bytes are the ones separate calls give).  A single caller thread never waits: nothing is in flight
// when its call arrives.  CVTTMI_DROPIN_COALESCE=0 turns it off (every call on its thread's own context).
#include "../../include/cvtt/ConvectionKernels.h"
#include "../../include/cvtt_mi355x.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
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

    // ---- coalescing of concurrent one-group calls ----
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
    struct Request
    {
        const CallKey *key;
        const uint8_t *in;
        uint8_t *out;
        int rc;
        std::atomic<bool> done;
    };
    // encode `numBlocks` contiguous blocks with the given key on `ctx`
    typedef int (*BatchFn)(cvttmi_context *ctx, const CallKey &key, uint8_t *out, const uint8_t *in, size_t numBlocks);

    class Coalescer
    {
    public:
        static const size_t kMaxGroups = 256; // groups per launch (256 x 8 blocks: 128 KiB of PixelBlockU8)
        Coalescer() : ctx_(NULL), busy_(false), recent_(1), busyFlag_(false), stageIn_(NULL), stageOut_(NULL) {}
        ~Coalescer()
        {
            if (ctx_)
            {
                if (stageIn_) cvttmi_host_free(ctx_, stageIn_);
                if (stageOut_) cvttmi_host_free(ctx_, stageOut_);
                cvttmi_destroy(ctx_);
            }
        }
        static bool enabled()
        {
            static const bool on = !(getenv("CVTTMI_DROPIN_COALESCE") && atoi(getenv("CVTTMI_DROPIN_COALESCE")) == 0);
            return on;
        }
        // one group: 8 blocks of inBytes / 8 bytes each in, outBytes out
        int call(const CallKey &key, uint8_t *out, const uint8_t *in, size_t inBytes, size_t outBytes, BatchFn fn)
        {
            Request me;
            me.key = &key;
            me.in = in;
            me.out = out;
            me.rc = CVTTMI_OK;
            me.done.store(false, std::memory_order_relaxed);
            std::unique_lock<std::mutex> lock(mu_);
            pending_.push_back(&me);
            for (;;)
            {
                if (me.done.load(std::memory_order_acquire))
                    return me.rc;
                if (!busy_)
                    break; // nothing in flight: this thread runs the next launch
                // A launch is in flight (tens of microseconds): poll for a while without the lock -- a futex wake-up costs
                // about as much as the launch itself -- and only then sleep on the condition variable.
                lock.unlock();
                const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                bool turn = false;
                while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(400))
                {
                    if (me.done.load(std::memory_order_acquire) || !busyFlag_.load(std::memory_order_acquire))
                    {
                        turn = true;
                        break;
                    }
                    std::this_thread::yield();
                }
                lock.lock();
                if (!turn && busy_ && !me.done.load(std::memory_order_acquire))
                    cv_.wait_for(lock, std::chrono::milliseconds(2));
            }
            busy_ = true;
            busyFlag_.store(true, std::memory_order_release);
            // The callers of the previous launch return, prepare their next group and arrive here within a few microseconds of
            // each other; the first one to arrive would otherwise leave with a launch of its own and make the others wait for
            // it.  So when recent launches carried more calls than are waiting now, give the others a moment (bounded: 100 us,
            // about one one-wave launch; measured with 16 callers, BC7 / ETC2 RGBA calls per second in total: no wait 96 k / 30 k,
            // 40 us 129 k / 46 k, 80 us 137 k / 51 k, 150 us 147 k / 56 k) -- a lone caller thread (recent_ == 1) never waits, and a
            // pool that shrinks pays the wait once per lost thread (recent_ falls by one per launch).
            if (pending_.size() < recent_)
            {
                const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                static const int windowUs = getenv("CVTTMI_DROPIN_WINDOW_US") ? atoi(getenv("CVTTMI_DROPIN_WINDOW_US")) : 100; // developer knob
                while (pending_.size() < recent_ && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(windowUs))
                {
                    lock.unlock();
                    std::this_thread::yield();
                    lock.lock();
                }
            }
            // every waiting call of this kind, in arrival order, goes into this launch
            std::vector<Request *> batch;
            for (size_t i = 0; i < pending_.size();)
            {
                if (batch.size() < kMaxGroups && pending_[i]->key->same(key))
                {
                    batch.push_back(pending_[i]);
                    pending_.erase(pending_.begin() + i);
                }
                else
                    i++;
            }
            // what "recently" means: this launch, or one less than before when it carried fewer calls (a lone caller is back at 1 after a few calls)
            recent_ = batch.size() >= recent_ ? batch.size() : recent_ - 1;
            lock.unlock();
            int rc = ensure(inBytes, outBytes);
            if (rc == CVTTMI_OK)
            {
                if (batch.size() == 1)
                    rc = fn(ctx_, key, out, in, cvtt::NumParallelBlocks);
                else
                {
                    for (size_t i = 0; i < batch.size(); i++)
                        memcpy(static_cast<uint8_t *>(stageIn_) + i * inBytes, batch[i]->in, inBytes);
                    rc = fn(ctx_, key, static_cast<uint8_t *>(stageOut_), static_cast<const uint8_t *>(stageIn_), batch.size() * cvtt::NumParallelBlocks);
                    if (rc == CVTTMI_OK)
                        for (size_t i = 0; i < batch.size(); i++)
                            memcpy(batch[i]->out, static_cast<uint8_t *>(stageOut_) + i * outBytes, outBytes);
                }
            }
            lock.lock();
            for (size_t i = 0; i < batch.size(); i++)
            {
                Request *r = batch[i];
                r->rc = rc;
                r->done.store(true, std::memory_order_release); // (a request other than `me` may be gone right after this)
            }
            busy_ = false;
            busyFlag_.store(false, std::memory_order_release);
            cv_.notify_all();
            return rc;
        }
        const char *lastError() { return ctx_ ? cvttmi_last_error(ctx_) : "no context"; }

    private:
        int ensure(size_t inBytes, size_t outBytes)
        {
            if (!ctx_)
            {
                const char *dev = getenv("CVTTMI_DEVICE");
                const int rc = cvttmi_create(&ctx_, dev ? atoi(dev) : 0);
                if (rc != CVTTMI_OK)
                {
                    fprintf(stderr, "cvtt (MI355X): no usable gfx950 device (cvttmi_create = %d); there is no CPU fallback\n", rc);
                    abort();
                }
            }
            // page-locked staging for the gathered groups (the largest block type: PixelBlockF16 / 16-byte outputs)
            if (!stageIn_ && cvttmi_host_alloc(ctx_, &stageIn_, kMaxGroups * 8 * 128) != CVTTMI_OK)
                return CVTTMI_E_HIP;
            if (!stageOut_ && cvttmi_host_alloc(ctx_, &stageOut_, kMaxGroups * 8 * 64) != CVTTMI_OK)
                return CVTTMI_E_HIP;
            (void)inBytes;
            (void)outBytes;
            return CVTTMI_OK;
        }
        cvttmi_context *ctx_;
        std::mutex mu_;
        std::condition_variable cv_;
        std::vector<Request *> pending_;
        bool busy_;
        size_t recent_; // calls per launch, recently
        std::atomic<bool> busyFlag_; // == busy_, readable without the lock
        void *stageIn_, *stageOut_;
    };
    Coalescer g_coalescer;

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
    void checkShared(int rc, const char *what)
    {
        if (rc != CVTTMI_OK)
        {
            fprintf(stderr, "cvtt (MI355X): %s failed (%d): %s\n", what, rc, g_coalescer.lastError());
            abort();
        }
    }

    enum Kind { K_BC7, K_BC1, K_BC2, K_BC3, K_BC4U, K_BC4S, K_BC5U, K_BC5S, K_BC6HU, K_BC6HS, K_ETC1, K_ETC2, K_ETC2RGBA, K_ETC2PT, K_ETC2A, K_A11U, K_A11S };
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
    // one reference-style call: a group of 8 blocks, `inBytes` in, `outBytes` out
    void oneGroup(int kind, uint8_t *pBC, const void *pBlocks, size_t inBytes, size_t outBytes, const cvtt::Options &options, const char *what,
                  const cvttmi_options *alloc = NULL, const cvtt::BC7EncodingPlan *plan = NULL)
    {
        const CallKey key = makeKey(kind, options, alloc, plan);
        if (Coalescer::enabled())
            checkShared(g_coalescer.call(key, pBC, static_cast<const uint8_t *>(pBlocks), inBytes, outBytes, encodeBatch), what);
        else
            check(encodeBatch(context(), key, pBC, static_cast<const uint8_t *>(pBlocks), cvtt::NumParallelBlocks), what);
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
