// Host side of the C ABI declared in include/cvtt_mi355x.h.
//
// Mirrors what the reference's API glue does around its per-format computers
// (reference ConvectionKernels_API.cpp:41-54: Util::FillWeights, then one Pack per group),
// except that a call covers a whole batch of groups and the work happens in HIP kernels.
// There is deliberately no CPU implementation here: without a gfx950 device every encode
// entry point fails with CVTTMI_E_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <xmmintrin.h>
#include <math.h>

#include <mutex>
#include <string>

#include "cvtt_device.h"
#include "bc7_tables.h"
#include "s3tc_sc_tables.h"
#include "bc7_quality_events.h"
#include "fake709_rounding.h"
#include "bc6h_layout.h"
#include "etc_tables.h"

extern "C" hipError_t cvttmi_launch_bc7(const void *d_blocks, void *d_out, const CvttBc7Args *args,
                                        const CvttDeviceTables *d_tables, const CvttBc7DevicePlan *d_plan,
                                        hipStream_t stream);

extern "C" hipError_t cvttmi_launch_tile(const void *d_image, void *d_blocks, uint32_t width, uint32_t height, size_t rowPitch,
                                         uint32_t bytesPerPixel, hipStream_t stream);
extern "C" hipError_t cvttmi_launch_compact_rows(const void *d_packed, void *d_out, uint32_t width, uint32_t height,
                                                 uint32_t bytesPerBlock, hipStream_t stream);
extern "C" hipError_t cvttmi_launch_selftest(uint64_t seed, uint64_t first, uint32_t count, void *d_operands, void *d_results,
                                             hipStream_t stream);
extern "C" hipError_t cvttmi_launch_decode(const void *d_bc, void *d_out, uint32_t numBlocks, int format,
                                           const CvttDeviceTables *d_tables, hipStream_t stream);
extern "C" hipError_t cvttmi_launch_eac11(const void *d_blocksS16, void *d_out, uint32_t numBlocks, int isSigned,
                                          const CvttDeviceTables *d_tables, hipStream_t stream);
extern "C" hipError_t cvttmi_launch_s3tc_alpha(const void *d_blocks, void *d_out, uint32_t numBlocks, uint32_t channel, uint32_t outStride,
                                               uint32_t outOffset, int isSigned, int explicitAlpha, int seedPoints, int refineRounds,
                                               const CvttDeviceTables *d_tables, hipStream_t stream);
extern "C" hipError_t cvttmi_launch_bc1(const void *d_blocks, void *d_out, const CvttBc1Args *args,
                                        const CvttDeviceTables *d_tables, hipStream_t stream);

extern "C" hipError_t cvttmi_launch_bc6h(const void *d_blocks, void *d_out, const CvttBc6hArgs *args,
                                         const CvttDeviceTables *d_tables, int isSigned, hipStream_t stream);

extern "C" hipError_t cvttmi_launch_etc2(const void *d_blocks, void *d_out, const CvttEtcArgs *args,
                                         const CvttDeviceTables *d_tables, int mode, hipStream_t stream);

static_assert(sizeof(cvttmi_options) == 44, "cvtt::Options layout");
static_assert(sizeof(cvttmi_bc7_plan) == 808, "cvtt::BC7EncodingPlan layout");

struct cvttmi_context
{
    int device;
    CvttDeviceTables hostTables;
    CvttDeviceTables *dTables;
    // plan staging: a small ring of device plan slots so back-to-back launches with
    // different plans never race with an in-flight kernel
    static const int kPlanSlots = 8;
    CvttBc7DevicePlan *dPlans;
    CvttBc7DevicePlan *pinnedPlans; // the staged plans in pinned memory: sources of the asynchronous uploads
    cvttmi_bc7_plan lastPlan[kPlanSlots];
    bool planValid[kPlanSlots];
    // launches that read a slot since it was last written: one event per stream that launched one (a slot is rewritten
    // only when every one of them has finished; more than kPlanUsers distinct streams: the oldest entry is waited for)
    static const int kPlanUsers = 4;
    struct PlanUse
    {
        hipStream_t stream;
        hipEvent_t ev; // created on first use
        bool valid;
    } planUse[kPlanSlots][kPlanUsers];
    hipEvent_t evPlanUp[kPlanSlots]; // recorded after the slot's upload, on the stream that did it
    hipStream_t planUpStream[kPlanSlots];
    int nextPlanSlot;
    // host-pointer entry points: a pipeline of chunks, each slot with its own staging and stream, so that the uploads of
    // the next chunks and the downloads of the previous ones overlap the search of the current one, and the searches of
    // neighbouring chunks (small launches: a chunk is two or three waves per SIMD) fill the device together
    static const int kPipeSlots = 4;
    struct PipeSlot
    {
        void *pinnedIn, *pinnedOut, *dIn, *dOut;
        hipStream_t stream;
        hipEvent_t done;
    } pipe[kPipeSlots];
    size_t pipeInBytes, pipeOutBytes;
    hipStream_t stream; // = pipe[0].stream: the context's private stream
    // calls may arrive from several host threads and on several streams, but the work buffers below exist once:
    // `mu` serialises the host side of a call, `evLast` orders its launches after the previous call's
    std::recursive_mutex mu;
    hipEvent_t evLast;
    hipStream_t lastStream;
    bool lastValid;
    float *dPtTrial;      // BC7_RespectPunchThrough with more than 2 refine rounds (kMaxPTRefine of bc7_kernel.hip): the trial-error table of a launch's waves
    size_t dPtTrialBytes;
    size_t ptTableBytes;  // cap of that table per launch (256 MB; CVTTMI_BC7_PT_TABLE_KB, read when the context is created)
    bool exhaustive; // search every candidate even when it provably cannot win
    // BC7: blocks with many live mode-7 partitions are finished by a second launch (bc7_kernel.hip, HARD).
    // One set of buffers per context: launches of one context are expected on one stream at a time.
    static const uint32_t kHardSlots = 8192;
    uint32_t *dHardCount;
    CvttBc7HardRec *dHardRec;
    CvttBc7HardCand *dHardCand;
    int hardMin;         // live partitions of a wave from which its blocks are handed over; 0 = never
    int hardDiv;
    int hardCapOverride; // > 0: slots per launch (experiments)
    uint32_t lastHardCap;
    // timing
    bool timing;
    hipEvent_t evStart, evStop;
    double totalMs;
    uint64_t launches;
    std::string lastError;
};

namespace
{
    int fail(cvttmi_context *ctx, int code, const char *what, hipError_t e = hipSuccess)
    {
        if (ctx)
        {
            std::lock_guard<std::recursive_mutex> lock(ctx->mu); // callers on several threads may fail at the same time
            ctx->lastError = what;
            if (e != hipSuccess)
            {
                ctx->lastError += ": ";
                ctx->lastError += hipGetErrorString(e);
            }
        }
        return code;
    }

    // Util::ComputeTweakFactors, reference ConvectionKernels_Util.cpp:75-84 (binary32 on the host)
    void tweakFactors(int tweak, int range, float out[2])
    {
        const int totalUnits = range - 1;
        const int minOutsideUnits = (tweak >> 1) & 1;
        const int maxOutsideUnits = tweak & 1;
        const int insideUnits = totalUnits - minOutsideUnits - maxOutsideUnits;
        volatile float a = -static_cast<float>(minOutsideUnits) / static_cast<float>(insideUnits);
        volatile float b = static_cast<float>(maxOutsideUnits) / static_cast<float>(insideUnits) + 1.0f;
        out[0] = a;
        out[1] = b;
    }

    void fillTables(CvttDeviceTables &t)
    {
        memset(&t, 0, sizeof(t));
        for (int i = 0; i < 243; i++)
            t.shapeMask[i] = k_shape_mask[i];
        for (int i = 0; i < 64; i++)
        {
            t.partition2[i] = k_partition2[i];
            t.partition3[i] = k_partition3[i];
            t.shapes2[i][0] = k_shapes2[i * 2 + 0];
            t.shapes2[i][1] = k_shapes2[i * 2 + 1];
            for (int s = 0; s < 3; s++)
                t.shapes3[i][s] = k_shapes3[i * 3 + s];
            t.subsetMask3[i][0] = k_shape_mask[k_shapes3[i * 3 + 1]];
            t.subsetMask3[i][1] = k_shape_mask[k_shapes3[i * 3 + 2]];
            for (int which = 0; which < 3; which++)
            {
                const uint32_t bits = which == 0 ? t.partition2[i] : t.subsetMask3[i][which - 1];
                uint32_t *dst = which == 0 ? t.subsetByteMask[i] : t.subsetByteMask[64 + 2 * i + (which - 1)];
                for (int w = 0; w < 4; w++)
                {
                    dst[w] = 0;
                    for (int b = 0; b < 4; b++)
                        if ((bits >> (4 * w + b)) & 1u)
                            dst[w] |= 0xffu << (8 * b);
                }
            }
            t.anchor2[i] = k_anchor2[i];
            t.anchor3[i][0] = k_anchor3[i * 2 + 0];
            t.anchor3[i][1] = k_anchor3[i * 2 + 1];
        }
        memcpy(t.s3tcSingleColor, k_s3tcsc, sizeof(t.s3tcSingleColor));
        memcpy(t.fake709Rounding, k_fake709Rounding16, sizeof(t.fake709Rounding));
        for (int r = 0; r < 3; r++)
            for (int tw = 0; tw < 4; tw++)
                tweakFactors(tw, 4 << r, t.tweakFactors[r][tw]);
        for (int i = 0; i <= 16; i++)
        {
            const float v = static_cast<float>(i == 0 ? 1 : i);
            t.rcpTable[i] = _mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(v)));
        }
        for (int r = 0; r < 2; r++)
            for (int tw = 0; tw < 4; tw++)
                tweakFactors(tw, 3 + r, t.tweakFactors3[r][tw]); // range 3, tweak 3 divides by zero: never read
        for (int m = 0; m < 14; m++)
        {
            for (int i = 0; i < 7; i++)
                t.bc6hModeInfo[m][i] = k_bc6h_mode_info[m][i];
            for (int b = 0; b < 82; b++)
                t.bc6hLayout[m][b] = k_bc6h_header_layout[m][b];
        }
        for (int i = 0; i < 8; i++)
        {
            for (int s = 0; s < 4; s++)
                t.etc1Modifiers[i][s] = k_etc1_modifiers[i * 4 + s];
            t.thDistance[i] = k_th_distance[i];
            t.clusterCount[i] = k_cluster_count[i];
            t.clusterStart[i] = k_cluster_start[i];
        }
        for (int i = 0; i < 16; i++)
        {
            for (int s = 0; s < 4; s++)
                t.eacPositive[i][s] = k_eac_positive[i * 4 + s];
            for (int s = 0; s < 13; s++)
                t.eacRounding[i][s] = k_eac_rounding[i * 13 + s];
        }
        for (int i = 0; i < 632; i++)
            t.clusterOffsets[i] = k_cluster_offsets[i];
        t.rcpMaxIndex[0] = 0.0f;
        for (int bits = 1; bits <= 4; bits++)
        {
            volatile float r = 1.0f / static_cast<float>((1 << bits) - 1);
            t.rcpMaxIndex[bits] = r;
        }
    }

    int uploadTables(cvttmi_context *ctx)
    {
        hipError_t e = hipMemcpy(ctx->dTables, &ctx->hostTables, sizeof(CvttDeviceTables), hipMemcpyHostToDevice);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_HIP, "hipMemcpy(tables)", e);
        return CVTTMI_OK;
    }

    // The launches of this call use the context's one set of work buffers: make `stream` wait for the previous call's
    // launches when those went to a different stream (same stream: already ordered).
    void orderAfterPrevious(cvttmi_context *ctx, hipStream_t stream)
    {
        if (ctx->lastValid && ctx->lastStream != stream)
            (void)hipStreamWaitEvent(stream, ctx->evLast, 0);
    }
    void markLaunch(cvttmi_context *ctx, hipStream_t stream)
    {
        if (hipEventRecord(ctx->evLast, stream) == hipSuccess)
        {
            ctx->lastStream = stream;
            ctx->lastValid = true;
        }
    }

    // Upload of a plan into the ring of device slots, asynchronously on the launch stream.  A slot is rewritten only
    // after the last launch that read it has finished (its event; normally long past), never with a device-wide sync.
    int stagePlan(cvttmi_context *ctx, const cvttmi_bc7_plan *plan, hipStream_t stream, const CvttBc7DevicePlan **dPlanOut, int *slotOut)
    {
        for (int i = 0; i < cvttmi_context::kPlanSlots; i++)
        {
            if (ctx->planValid[i] && memcmp(&ctx->lastPlan[i], plan, sizeof(*plan)) == 0)
            {
                // uploaded on another stream: that copy must have landed before this stream's kernel reads the slot
                if (ctx->planUpStream[i] != stream)
                    (void)hipStreamWaitEvent(stream, ctx->evPlanUp[i], 0);
                *dPlanOut = ctx->dPlans + i;
                *slotOut = i;
                return CVTTMI_OK;
            }
        }
        const int slot = ctx->nextPlanSlot;
        ctx->nextPlanSlot = (slot + 1) % cvttmi_context::kPlanSlots;
        hipError_t e;
        // every launch that read the slot, on whatever stream, must have finished (normally long past)
        for (int u = 0; u < cvttmi_context::kPlanUsers; u++)
        {
            cvttmi_context::PlanUse &use = ctx->planUse[slot][u];
            if (use.valid && (e = hipEventSynchronize(use.ev)) != hipSuccess)
                return fail(ctx, CVTTMI_E_HIP, "hipEventSynchronize(plan slot)", e);
            use.valid = false;
        }
        ctx->planValid[slot] = false; // until the upload below has been queued
        CvttBc7DevicePlan &staged = ctx->pinnedPlans[slot];
        memset(&staged, 0, sizeof(staged));
        staged.plan = *plan;
        // the counts index fixed-size lists: never read past them, whatever the caller wrote
        const int numRGB = plan->rgbNumShapesToEvaluate > 243 ? 243 : plan->rgbNumShapesToEvaluate;
        const int numRGBA = plan->rgbaNumShapesToEvaluate > 129 ? 129 : plan->rgbaNumShapesToEvaluate;
        staged.plan.rgbNumShapesToEvaluate = static_cast<uint8_t>(numRGB);
        staged.plan.rgbaNumShapesToEvaluate = static_cast<uint8_t>(numRGBA);
        for (int i = 0; i < numRGB; i++)
        {
            const int shape = plan->rgbShapeList[i];
            if (shape < 243)
                staged.rgbListed[shape >> 5] |= 1u << (shape & 31);
        }
        for (int i = 0; i < numRGBA; i++)
        {
            const int shape = plan->rgbaShapeList[i];
            if (shape < 129)
                staged.rgbaListed[shape >> 5] |= 1u << (shape & 31);
        }
        e = hipMemcpyAsync(ctx->dPlans + slot, &staged, sizeof(staged), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_HIP, "hipMemcpyAsync(plan)", e);
        if ((e = hipEventRecord(ctx->evPlanUp[slot], stream)) != hipSuccess)
            return fail(ctx, CVTTMI_E_HIP, "hipEventRecord(plan upload)", e);
        ctx->planUpStream[slot] = stream;
        memcpy(&ctx->lastPlan[slot], plan, sizeof(*plan));
        ctx->planValid[slot] = true;
        *dPlanOut = ctx->dPlans + slot;
        *slotOut = slot;
        return CVTTMI_OK;
    }

    // A launch on `stream` that reads plan slot `slot` has just been queued.
    void markPlanUse(cvttmi_context *ctx, int slot, hipStream_t stream)
    {
        cvttmi_context::PlanUse *use = NULL;
        for (int u = 0; u < cvttmi_context::kPlanUsers && !use; u++)
            if (ctx->planUse[slot][u].valid && ctx->planUse[slot][u].stream == stream)
                use = &ctx->planUse[slot][u];
        for (int u = 0; u < cvttmi_context::kPlanUsers && !use; u++)
            if (!ctx->planUse[slot][u].valid)
                use = &ctx->planUse[slot][u];
        if (!use)
        {
            // more distinct streams than entries: the first entry's launch is waited for and the entry taken over
            use = &ctx->planUse[slot][0];
            if (hipEventSynchronize(use->ev) != hipSuccess)
                (void)hipDeviceSynchronize(); // the entry's launch cannot be waited for by its event: wait for everything
        }
        if (!use->ev && hipEventCreateWithFlags(&use->ev, hipEventDisableTiming) != hipSuccess)
        {
            use->ev = NULL;
            (void)hipStreamSynchronize(stream); // no event to remember the launch by: wait for it now
            use->valid = false;
            return;
        }
        use->stream = stream;
        use->valid = hipEventRecord(use->ev, stream) == hipSuccess;
        if (!use->valid)
            (void)hipStreamSynchronize(stream);
    }

    void freePipe(cvttmi_context *ctx)
    {
        for (int i = 0; i < cvttmi_context::kPipeSlots; i++)
        {
            cvttmi_context::PipeSlot &S = ctx->pipe[i];
            if (S.pinnedIn) (void)hipHostFree(S.pinnedIn);
            if (S.pinnedOut) (void)hipHostFree(S.pinnedOut);
            if (S.dIn) (void)hipFree(S.dIn);
            if (S.dOut) (void)hipFree(S.dOut);
            S.pinnedIn = S.pinnedOut = S.dIn = S.dOut = NULL;
        }
        ctx->pipeInBytes = ctx->pipeOutBytes = 0;
    }

    int ensurePipe(cvttmi_context *ctx, size_t inBytes, size_t outBytes)
    {
        if (ctx->pipeInBytes >= inBytes && ctx->pipeOutBytes >= outBytes)
            return CVTTMI_OK;
        for (int i = 0; i < cvttmi_context::kPipeSlots; i++)
            (void)hipStreamSynchronize(ctx->pipe[i].stream);
        inBytes = inBytes > ctx->pipeInBytes ? inBytes : ctx->pipeInBytes;
        outBytes = outBytes > ctx->pipeOutBytes ? outBytes : ctx->pipeOutBytes;
        freePipe(ctx);
        hipError_t e;
        for (int i = 0; i < cvttmi_context::kPipeSlots; i++)
        {
            cvttmi_context::PipeSlot &S = ctx->pipe[i];
            if ((e = hipHostMalloc(&S.pinnedIn, inBytes)) != hipSuccess || (e = hipHostMalloc(&S.pinnedOut, outBytes)) != hipSuccess ||
                (e = hipMalloc(&S.dIn, inBytes)) != hipSuccess || (e = hipMalloc(&S.dOut, outBytes)) != hipSuccess)
            {
                freePipe(ctx);
                return fail(ctx, CVTTMI_E_HIP, "staging allocation", e);
            }
        }
        ctx->pipeInBytes = inBytes;
        ctx->pipeOutBytes = outBytes;
        return CVTTMI_OK;
    }

    // page-locked (hipHostMalloc / hipHostRegister) host memory can be the source / target of an asynchronous copy
    bool isPinnedHost(const void *p)
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, p) != hipSuccess)
        {
            (void)hipGetLastError(); // pageable memory: not an error for us
            return false;
        }
        return attr.type == hipMemoryTypeHost;
    }
    // ... the WHOLE range [p, p + bytes): walked allocation by allocation.  Every piece must be page-locked memory for which
    // the runtime reports the covering range (hipMemGetAddressRange on its device alias); the walk continues at the end of
    // that range.  A byte the runtime gives no range for ends the walk with "not pinned", and the caller's buffer goes
    // through the staged copies instead of being read in place by the kernel (a pageable hole inside a buffer registered
    // in parts would otherwise be a GPU page fault).  One or two runtime calls per registration, whatever the size.
    bool isPinnedHostRange(const void *p, size_t bytes)
    {
        if (bytes == 0)
            return false;
        const uint8_t *b = static_cast<const uint8_t *>(p);
        size_t off = 0;
        for (int pieces = 0; off < bytes; pieces++)
        {
            if (pieces > 4096 || !isPinnedHost(b + off))
                return false;
            hipDeviceptr_t base = NULL;
            size_t size = 0;
            void *dev = NULL;
            if (hipHostGetDevicePointer(&dev, const_cast<uint8_t *>(b + off), 0) != hipSuccess || hipMemGetAddressRange(&base, &size, dev) != hipSuccess)
            {
                (void)hipGetLastError();
                return false;
            }
            const uint8_t *lo = static_cast<const uint8_t *>(static_cast<void *>(base));
            const uint8_t *d = static_cast<const uint8_t *>(dev);
            if (d < lo || d >= lo + size)
                return false;
            off += (size_t)(lo + size - d);
        }
        return true;
    }

    // The host-pointer entry points: `numBlocks` blocks of `inBpb` bytes in host memory -> `outBpb` bytes each in host
    // memory through `launch(dOut, dIn, n, stream)`, in chunks of 2^17 blocks (8 MiB of PixelBlockU8) dealt to pipeline slots with their own
    // stream, so the PCIe transfers of neighbouring chunks run beside the kernels of the current one (launches that use the
    // context's work buffers stay ordered through them).  Page-locked caller memory (cvttmi_host_alloc /
    // cvttmi_host_register) is transferred in place; pageable memory goes through the slots' pinned staging, the CPU copy
    // of the next chunk overlapping the device work of the current one.  Measured on 4096^2 BC7 with the 4-wave kernel,
    // median of 25 calls, page-locked / pageable, Mblocks/s: 2 slots x 2^17: 440 / 278; 4 x 2^17: 385 / 346; 4 x 2^16:
    // 323 / 326 (best 471, unstable); 4 x 2^18: 359 / 317.  CVTTMI_HOST_SLOTS / CVTTMI_HOST_CHUNK_LOG2 select other settings.
    // zeroCopyOk: the kernel reads every input byte once (BC7, where the in-place path was measured); kernels that read a
    // block several times (ETC2: every wave loads its whole group) would re-read it over PCIe and keep the staged copies.
    template <class Launch>
    int hostPipeline(cvttmi_context *ctx, uint8_t *out, const uint8_t *in, size_t numBlocks, size_t inBpb, size_t outBpb, Launch launch, bool zeroCopyOk = false)
    {
        static const int chunkLog2 = getenv("CVTTMI_HOST_CHUNK_LOG2") ? atoi(getenv("CVTTMI_HOST_CHUNK_LOG2")) : 17;
        const size_t kChunk = (size_t)1 << (chunkLog2 < 10 ? 10 : chunkLog2 > 24 ? 24 : chunkLog2);
        static const int slotsEnv = getenv("CVTTMI_HOST_SLOTS") ? atoi(getenv("CVTTMI_HOST_SLOTS")) : 0;
        const bool inPinned = isPinnedHostRange(in, numBlocks * inBpb);
        const bool outPinned = isPinnedHostRange(out, numBlocks * outBpb);
        // page-locked caller memory: two slots keep the link and the device busy; pageable memory: four, so that the CPU
        // copies into and out of the staging buffers overlap as well
        const int slotsWanted = slotsEnv > 0 ? slotsEnv : (inPinned && outPinned ? 2 : cvttmi_context::kPipeSlots);
        const size_t kSlots = (size_t)(slotsWanted > cvttmi_context::kPipeSlots ? cvttmi_context::kPipeSlots : slotsWanted);
        // Page-locked memory on both sides: the kernels read the input straight from host memory and write the result
        // straight back over PCIe -- no staging, no chunking, one launch over the whole batch (BC7: a wave reads 1 KiB
        // contiguous and writes 256 B; 4096^2 host to host 642 Mblocks/s against 440 with pipelined copies).
        static const bool zeroCopy = !(getenv("CVTTMI_HOST_ZEROCOPY") && atoi(getenv("CVTTMI_HOST_ZEROCOPY")) == 0);
        if (zeroCopy && zeroCopyOk && inPinned && outPinned)
        {
            void *dIn = NULL, *dOut = NULL;
            if (hipHostGetDevicePointer(&dIn, const_cast<uint8_t *>(in), 0) == hipSuccess && hipHostGetDevicePointer(&dOut, out, 0) == hipSuccess)
            {
                const int zrc = launch(dOut, dIn, numBlocks, ctx->stream);
                if (zrc != CVTTMI_OK)
                    return zrc;
                const hipError_t ze = hipStreamSynchronize(ctx->stream);
                return ze == hipSuccess ? CVTTMI_OK : fail(ctx, CVTTMI_E_HIP, "kernel execution", ze);
            }
            (void)hipGetLastError();
        }
        const size_t chunk = numBlocks < kChunk ? numBlocks : kChunk;
        int rc = ensurePipe(ctx, chunk * inBpb, chunk * outBpb);
        if (rc != CVTTMI_OK)
            return rc;
        const size_t numChunks = (numBlocks + chunk - 1) / chunk;
        hipError_t e;
        // an error leaves copies of earlier chunks in flight that still write into `out` / read `in`: nothing returns
        // before every pipeline stream has drained, so the caller may free its buffers as soon as it has the error code
        auto drained = [&](int code) -> int {
            for (size_t sl = 0; sl < kSlots; sl++)
                (void)hipStreamSynchronize(ctx->pipe[sl].stream);
            return code;
        };
        auto finish = [&](size_t c) -> int {
            cvttmi_context::PipeSlot &S = ctx->pipe[c % kSlots];
            hipError_t fe = hipEventSynchronize(S.done);
            if (fe != hipSuccess)
                return fail(ctx, CVTTMI_E_HIP, "kernel execution", fe);
            if (!outPinned)
            {
                const size_t n = (numBlocks - c * chunk) < chunk ? (numBlocks - c * chunk) : chunk;
                memcpy(out + c * chunk * outBpb, S.pinnedOut, n * outBpb);
            }
            return CVTTMI_OK;
        };
        for (size_t c = 0; c < numChunks; c++)
        {
            cvttmi_context::PipeSlot &S = ctx->pipe[c % kSlots];
            if (c >= kSlots && (rc = finish(c - kSlots)) != CVTTMI_OK)
                return drained(rc);
            const size_t first = c * chunk;
            const size_t n = (numBlocks - first) < chunk ? (numBlocks - first) : chunk;
            const void *src = in + first * inBpb;
            if (!inPinned)
            {
                memcpy(S.pinnedIn, src, n * inBpb);
                src = S.pinnedIn;
            }
            if ((e = hipMemcpyAsync(S.dIn, src, n * inBpb, hipMemcpyHostToDevice, S.stream)) != hipSuccess)
                return drained(fail(ctx, CVTTMI_E_HIP, "H2D", e));
            if ((rc = launch(S.dOut, S.dIn, n, S.stream)) != CVTTMI_OK)
                return drained(rc);
            void *dst = outPinned ? static_cast<void *>(out + first * outBpb) : S.pinnedOut;
            if ((e = hipMemcpyAsync(dst, S.dOut, n * outBpb, hipMemcpyDeviceToHost, S.stream)) != hipSuccess)
                return drained(fail(ctx, CVTTMI_E_HIP, "D2H", e));
            if ((e = hipEventRecord(S.done, S.stream)) != hipSuccess)
                return drained(fail(ctx, CVTTMI_E_HIP, "hipEventRecord", e));
        }
        for (size_t c = numChunks >= kSlots ? numChunks - kSlots : 0; c < numChunks; c++)
            if ((rc = finish(c)) != CVTTMI_OK)
                return drained(rc);
        return CVTTMI_OK;
    }

    // Util::FillWeights (reference ConvectionKernels_Util.cpp:62-73) + the derived per-call
    // constants the reference computes in scalar float on the host.
    void fillWeightArgs(const cvttmi_options *o, float w[4], float wSq[4], float rcpW[4])
    {
        if (o->flags & CVTTMI_FLAG_UNIFORM)
            w[0] = w[1] = w[2] = w[3] = 1.0f;
        else
        {
            w[0] = o->redWeight;
            w[1] = o->greenWeight;
            w[2] = o->blueWeight;
            w[3] = o->alphaWeight;
        }
        for (int ch = 0; ch < 4; ch++)
        {
            volatile float sq = w[ch] * w[ch];
            wSq[ch] = sq;
            volatile float r = 1.0f;
            if (w[ch] != 0.0f)
                r = 1.0f / w[ch];
            rcpW[ch] = r;
        }
    }
}

extern "C"
{
    void cvttmi_default_options(cvttmi_options *out)
    {
        // cvtt::Options::Options(), reference ConvectionKernels.h:89-102
        out->flags = CVTTMI_FLAGS_DEFAULT;
        out->threshold = 0.5f;
        out->redWeight = 0.2125f / 0.7154f;
        out->greenWeight = 1.0f;
        out->blueWeight = 0.0721f / 0.7154f;
        out->alphaWeight = 1.0f;
        out->refineRoundsBC7 = 2;
        out->refineRoundsBC6H = 3;
        out->refineRoundsIIC = 8;
        out->refineRoundsS3TC = 2;
        out->seedPoints = 4;
    }

    void cvttmi_default_bc7_plan(cvttmi_bc7_plan *p)
    {
        // cvtt::BC7EncodingPlan::BC7EncodingPlan(), reference ConvectionKernels.h:166-198:
        // every shape, every partition, four seed points
        memset(p, 0, sizeof(*p));
        p->mode0PartitionEnabled = 0xffff;
        p->mode1PartitionEnabled = p->mode2PartitionEnabled = p->mode3PartitionEnabled = ~0ull;
        p->mode7RGBAPartitionEnabled = p->mode7RGBPartitionEnabled = ~0ull;
        p->mode6Enabled = 1;
        memset(p->mode4SP, 4, sizeof(p->mode4SP));
        memset(p->mode5SP, 4, sizeof(p->mode5SP));
        memset(p->seedPointsForShapeRGB, 4, sizeof(p->seedPointsForShapeRGB));
        memset(p->seedPointsForShapeRGBA, 4, sizeof(p->seedPointsForShapeRGBA));
        for (int i = 0; i < 243; i++)
            p->rgbShapeList[i] = static_cast<uint8_t>(i);
        for (int i = 0; i < 129; i++)
            p->rgbaShapeList[i] = static_cast<uint8_t>(i);
        p->rgbNumShapesToEvaluate = 243;
        p->rgbaNumShapesToEvaluate = 129;
    }

    void cvttmi_default_bc7_fine_tuning(cvttmi_bc7_fine_tuning *p)
    {
        memset(p, 4, sizeof(*p)); // cvtt::BC7FineTuningParams(), ConvectionKernels.h:117-139: four seed points everywhere
    }

    // Shape lists, counts and the mode-7 RGB mask follow from the enables and per-shape seed points
    // (tail of ConfigureBC7EncodingPlanFromFineTuningParams, ConvectionKernels_BC67.cpp:3460-3480).
    static void finishPlan(cvttmi_bc7_plan *plan)
    {
        plan->rgbNumShapesToEvaluate = plan->rgbaNumShapesToEvaluate = 0;
        memset(plan->rgbShapeList, 0, sizeof(plan->rgbShapeList));
        memset(plan->rgbaShapeList, 0, sizeof(plan->rgbaShapeList));
        for (int shape = 0; shape < 243; shape++)
            if (plan->seedPointsForShapeRGB[shape])
                plan->rgbShapeList[plan->rgbNumShapesToEvaluate++] = static_cast<uint8_t>(shape);
        for (int shape = 0; shape < 129; shape++)
            if (plan->seedPointsForShapeRGBA[shape])
                plan->rgbaShapeList[plan->rgbaNumShapesToEvaluate++] = static_cast<uint8_t>(shape);
        plan->mode7RGBPartitionEnabled = plan->mode7RGBAPartitionEnabled & ~plan->mode3PartitionEnabled;
    }

    int cvttmi_bc7_plan_from_fine_tuning(cvttmi_bc7_plan *plan, const cvttmi_bc7_fine_tuning *params)
    {
        if (!plan || !params)
            return CVTTMI_E_INVALID;
        memset(plan, 0, sizeof(*plan));
        // one row per partitioned mode: seed points per partition, number of partitions, subsets, where the enable bit
        // and the per-shape seed points live (modes 0-3 feed the RGB shape set, mode 7 the RGBA set)
        uint64_t m0 = 0;
        const struct
        {
            const uint8_t *sp;
            int numPartitions, numSubsets;
            uint64_t *enable;
            uint8_t *shapeSeeds;
        } rows[5] = {
            {params->mode0SP, 16, 3, &m0, plan->seedPointsForShapeRGB},
            {params->mode1SP, 64, 2, &plan->mode1PartitionEnabled, plan->seedPointsForShapeRGB},
            {params->mode2SP, 64, 3, &plan->mode2PartitionEnabled, plan->seedPointsForShapeRGB},
            {params->mode3SP, 64, 2, &plan->mode3PartitionEnabled, plan->seedPointsForShapeRGB},
            {params->mode7SP, 64, 2, &plan->mode7RGBAPartitionEnabled, plan->seedPointsForShapeRGBA},
        };
        for (const auto &row : rows)
            for (int partition = 0; partition < row.numPartitions; partition++)
            {
                const uint8_t sp = row.sp[partition];
                if (!sp)
                    continue;
                *row.enable |= 1ull << partition;
                for (int subset = 0; subset < row.numSubsets; subset++)
                {
                    const int shape = (row.numSubsets == 3) ? k_shapes3[partition * 3 + subset] : k_shapes2[partition * 2 + subset];
                    if (row.shapeSeeds[shape] < sp)
                        row.shapeSeeds[shape] = sp;
                }
            }
        plan->mode0PartitionEnabled = static_cast<uint16_t>(m0);
        memcpy(plan->mode4SP, params->mode4SP, sizeof(plan->mode4SP));
        memcpy(plan->mode5SP, params->mode5SP, sizeof(plan->mode5SP));
        if (params->mode6SP)
        {
            plan->mode6Enabled = 1;
            if (plan->seedPointsForShapeRGBA[0] < params->mode6SP) // the whole block is shape 0
                plan->seedPointsForShapeRGBA[0] = params->mode6SP;
        }
        finishPlan(plan);
        return CVTTMI_OK;
    }

    int cvttmi_bc7_plan_from_quality(cvttmi_bc7_plan *plan, int quality)
    {
        if (!plan)
            return CVTTMI_E_INVALID;
        quality = quality < 1 ? 1 : quality > 100 ? 100 : quality; // BC67.cpp:3295-3298
        memset(plan, 0, sizeof(*plan));
        uint64_t *masks[5] = {NULL, &plan->mode1PartitionEnabled, &plan->mode2PartitionEnabled, &plan->mode3PartitionEnabled,
                              &plan->mode7RGBAPartitionEnabled};
        uint64_t m0 = 0;
        masks[0] = &m0;
        for (int i = 0; i < CVTT_BC7_NUM_QUALITY_EVENTS; i++)
        {
            const unsigned e = k_bc7QualityEvents[i];
            if (static_cast<int>(e >> 24) > quality)
                break;
            const unsigned kind = (e >> 16) & 255u, index = (e >> 8) & 255u;
            const uint8_t value = static_cast<uint8_t>(e & 255u);
            if (kind < 5)
                *masks[kind] = (*masks[kind] & ~(1ull << index)) | (static_cast<uint64_t>(value & 1u) << index);
            else if (kind == 5)
                plan->mode4SP[index >> 1][index & 1] = value;
            else if (kind == 6)
                plan->mode5SP[index] = value;
            else if (kind == 7)
                plan->mode6Enabled = value;
            else if (kind == 8)
                plan->seedPointsForShapeRGB[index] = value;
            else
                plan->seedPointsForShapeRGBA[index] = value;
        }
        plan->mode0PartitionEnabled = static_cast<uint16_t>(m0);
        finishPlan(plan);
        return CVTTMI_OK;
    }

    int cvttmi_create(cvttmi_context **out, int device)
    {
        if (!out)
            return CVTTMI_E_INVALID;
        *out = NULL;
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
            return CVTTMI_E_NO_DEVICE;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess)
            return CVTTMI_E_NO_DEVICE;
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return CVTTMI_E_NO_DEVICE; // kernels are built for gfx950 only
        if (hipSetDevice(device) != hipSuccess)
            return CVTTMI_E_NO_DEVICE;

        cvttmi_context *ctx = new cvttmi_context();
        ctx->device = device;
        ctx->dTables = NULL;
        ctx->dPlans = NULL;
        ctx->pinnedPlans = NULL;
        ctx->nextPlanSlot = 0;
        for (int i = 0; i < cvttmi_context::kPlanSlots; i++)
        {
            ctx->planValid[i] = false;
            ctx->evPlanUp[i] = NULL;
            ctx->planUpStream[i] = NULL;
        }
        memset(ctx->planUse, 0, sizeof(ctx->planUse));
        memset(ctx->pipe, 0, sizeof(ctx->pipe));
        ctx->pipeInBytes = ctx->pipeOutBytes = 0;
        ctx->stream = NULL;
        ctx->evLast = ctx->evStart = ctx->evStop = NULL;
        ctx->lastStream = NULL;
        ctx->lastValid = false;
        ctx->dPtTrial = NULL;
        ctx->dPtTrialBytes = 0;
        ctx->timing = false;
        ctx->exhaustive = getenv("CVTTMI_EXHAUSTIVE") != NULL && atoi(getenv("CVTTMI_EXHAUSTIVE")) != 0;
        ctx->totalMs = 0.0;
        ctx->launches = 0;
        ctx->dHardCount = NULL;
        ctx->dHardRec = NULL;
        ctx->dHardCand = NULL;
        ctx->lastHardCap = 0;
        ctx->ptTableBytes = getenv("CVTTMI_BC7_PT_TABLE_KB") ? (size_t)atol(getenv("CVTTMI_BC7_PT_TABLE_KB")) << 10 : (size_t)256 << 20;
        ctx->hardMin = getenv("CVTTMI_BC7_HARD_MIN") ? atoi(getenv("CVTTMI_BC7_HARD_MIN")) : 8;
        ctx->hardDiv = getenv("CVTTMI_BC7_HARD_DIV") ? atoi(getenv("CVTTMI_BC7_HARD_DIV")) : 128;
        if (ctx->hardDiv < 1)
            ctx->hardDiv = 1;
        ctx->hardCapOverride = getenv("CVTTMI_BC7_HARD_CAP") ? atoi(getenv("CVTTMI_BC7_HARD_CAP")) : 0;
        fillTables(ctx->hostTables);
        hipError_t e;
        bool ok =
            (e = hipMalloc(reinterpret_cast<void **>(&ctx->dHardCount), 256)) == hipSuccess &&
            (e = hipMalloc(reinterpret_cast<void **>(&ctx->dHardRec), sizeof(CvttBc7HardRec) * cvttmi_context::kHardSlots)) == hipSuccess &&
            (e = hipMalloc(reinterpret_cast<void **>(&ctx->dHardCand), sizeof(CvttBc7HardCand) * cvttmi_context::kHardSlots * kHardWaves)) == hipSuccess &&
            (e = hipMalloc(reinterpret_cast<void **>(&ctx->dTables), sizeof(CvttDeviceTables))) == hipSuccess &&
            (e = hipMalloc(reinterpret_cast<void **>(&ctx->dPlans), sizeof(CvttBc7DevicePlan) * cvttmi_context::kPlanSlots)) == hipSuccess &&
            (e = hipHostMalloc(reinterpret_cast<void **>(&ctx->pinnedPlans), sizeof(CvttBc7DevicePlan) * cvttmi_context::kPlanSlots)) == hipSuccess &&

            (e = hipEventCreateWithFlags(&ctx->evLast, hipEventDisableTiming)) == hipSuccess &&
            (e = hipEventCreate(&ctx->evStart)) == hipSuccess && (e = hipEventCreate(&ctx->evStop)) == hipSuccess;
        for (int i = 0; ok && i < cvttmi_context::kPipeSlots; i++)
            ok = (e = hipStreamCreate(&ctx->pipe[i].stream)) == hipSuccess &&
                 (e = hipEventCreateWithFlags(&ctx->pipe[i].done, hipEventDisableTiming)) == hipSuccess;
        for (int i = 0; ok && i < cvttmi_context::kPlanSlots; i++)
            ok = (e = hipEventCreateWithFlags(&ctx->evPlanUp[i], hipEventDisableTiming)) == hipSuccess;
        ctx->stream = ctx->pipe[0].stream;
        if (!ok || uploadTables(ctx) != CVTTMI_OK)
        {
            cvttmi_destroy(ctx); // frees whatever was created: every handle above starts out NULL
            return CVTTMI_E_HIP;
        }
        *out = ctx;
        return CVTTMI_OK;
    }

    void cvttmi_destroy(cvttmi_context *ctx)
    {
        if (!ctx)
            return;
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
        if (ctx->dTables) (void)hipFree(ctx->dTables);
        if (ctx->dPlans) (void)hipFree(ctx->dPlans);
        if (ctx->pinnedPlans) (void)hipHostFree(ctx->pinnedPlans);
        freePipe(ctx);
        if (ctx->dPtTrial) (void)hipFree(ctx->dPtTrial);
        if (ctx->dHardCount) (void)hipFree(ctx->dHardCount);
        if (ctx->dHardRec) (void)hipFree(ctx->dHardRec);
        if (ctx->dHardCand) (void)hipFree(ctx->dHardCand);
        for (int i = 0; i < cvttmi_context::kPipeSlots; i++)
        {
            if (ctx->pipe[i].stream) (void)hipStreamDestroy(ctx->pipe[i].stream);
            if (ctx->pipe[i].done) (void)hipEventDestroy(ctx->pipe[i].done);
        }
        for (int i = 0; i < cvttmi_context::kPlanSlots; i++)
        {
            for (int u = 0; u < cvttmi_context::kPlanUsers; u++)
                if (ctx->planUse[i][u].ev) (void)hipEventDestroy(ctx->planUse[i][u].ev);
            if (ctx->evPlanUp[i]) (void)hipEventDestroy(ctx->evPlanUp[i]);
        }
        if (ctx->evLast) (void)hipEventDestroy(ctx->evLast);
        if (ctx->evStart) (void)hipEventDestroy(ctx->evStart);
        if (ctx->evStop) (void)hipEventDestroy(ctx->evStop);
        delete ctx;
    }

    // Page-locked host memory for the host-pointer entry points: they transfer such buffers in place (no staging copy).
    int cvttmi_host_alloc(cvttmi_context *ctx, void **ptr, size_t bytes)
    {
        if (!ctx || !ptr)
            return CVTTMI_E_INVALID;
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess)
            e = hipHostMalloc(ptr, bytes ? bytes : 1);
        return e == hipSuccess ? CVTTMI_OK : fail(ctx, CVTTMI_E_HIP, "hipHostMalloc", e);
    }
    // ctx may be NULL (the memory may outlive the context that allocated it; no error text is kept then)
    int cvttmi_host_free(cvttmi_context *ctx, void *ptr)
    {
        hipError_t e = ptr ? hipHostFree(ptr) : hipSuccess;
        return e == hipSuccess ? CVTTMI_OK : fail(ctx, CVTTMI_E_HIP, "hipHostFree", e);
    }
    int cvttmi_host_register(cvttmi_context *ctx, void *ptr, size_t bytes)
    {
        if (!ctx || !ptr || !bytes)
            return CVTTMI_E_INVALID;
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess)
            e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
        return e == hipSuccess ? CVTTMI_OK : fail(ctx, CVTTMI_E_HIP, "hipHostRegister", e);
    }
    int cvttmi_host_unregister(cvttmi_context *ctx, void *ptr)
    {
        if (!ctx || !ptr)
            return CVTTMI_E_INVALID;
        hipError_t e = hipHostUnregister(ptr);
        return e == hipSuccess ? CVTTMI_OK : fail(ctx, CVTTMI_E_HIP, "hipHostUnregister", e);
    }

    const char *cvttmi_last_error(const cvttmi_context *ctx)
    {
        return ctx ? ctx->lastError.c_str() : "no context";
    }

    int cvttmi_set_rcp_table(cvttmi_context *ctx, const float lut[17])
    {
        if (!ctx || !lut)
            return CVTTMI_E_INVALID;
        hipSetDevice(ctx->device);
        hipDeviceSynchronize();
        for (int i = 1; i <= 16; i++)
            ctx->hostTables.rcpTable[i] = lut[i];
        ctx->hostTables.rcpTable[0] = lut[1];
        return uploadTables(ctx);
    }

    int cvttmi_get_rcp_table(const cvttmi_context *ctx, float lut[17])
    {
        if (!ctx || !lut)
            return CVTTMI_E_INVALID;
        for (int i = 0; i <= 16; i++)
            lut[i] = ctx->hostTables.rcpTable[i];
        return CVTTMI_OK;
    }

    size_t cvttmi_tiled_block_count(uint32_t width, uint32_t height)
    {
        const size_t perRow = (((size_t)width + 3) / 4 + 7) / 8 * 8;
        return perRow * (((size_t)height + 3) / 4);
    }

    int cvttmi_tile_image_device(cvttmi_context *ctx, void *d_blocks, const void *d_image, uint32_t width, uint32_t height,
                                 size_t rowPitchBytes, int pixelFormat, void *hipStream)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        const uint32_t bpp = (pixelFormat == CVTTMI_PIXELS_RGBA8) ? 4u : (pixelFormat == CVTTMI_PIXELS_RGBA16F) ? 8u : 0u;
        if (!d_blocks || !d_image || bpp == 0 || width == 0 || height == 0 || rowPitchBytes < (size_t)width * bpp ||
            (rowPitchBytes % bpp) != 0)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu); // lastError, timing state and work buffers are per context
        e = cvttmi_launch_tile(d_image, d_blocks, width, height, rowPitchBytes, bpp, static_cast<hipStream_t>(hipStream));
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_HIP, "tile kernel launch", e);
        return CVTTMI_OK;
    }

    int cvttmi_compact_rows_device(cvttmi_context *ctx, void *d_out, const void *d_packed, uint32_t width, uint32_t height,
                                   uint32_t bytesPerBlock, void *hipStream)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!d_out || !d_packed || width == 0 || height == 0 || (bytesPerBlock != 8 && bytesPerBlock != 16))
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu); // lastError, timing state and work buffers are per context
        e = cvttmi_launch_compact_rows(d_packed, d_out, width, height, bytesPerBlock, static_cast<hipStream_t>(hipStream));
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_HIP, "compact kernel launch", e);
        return CVTTMI_OK;
    }

    // format: 0 = BC7 -> PixelBlockU8 (64 B), 1 = BC6HU, 2 = BC6HS -> PixelBlockF16 (128 B)
    static int decodeDevice(cvttmi_context *ctx, void *d_blocks, const void *d_bc, size_t numBlocks, int format, void *hipStream)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!d_blocks || !d_bc || (numBlocks % 8) != 0 || numBlocks > 0xfffffff0u)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu); // lastError, timing state and work buffers are per context
        e = cvttmi_launch_decode(d_bc, d_blocks, static_cast<uint32_t>(numBlocks), format, ctx->dTables, static_cast<hipStream_t>(hipStream));
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_HIP, "decode kernel launch", e);
        return CVTTMI_OK;
    }

    static int decodeHost(cvttmi_context *ctx, uint8_t *blocks, const uint8_t *bc, size_t numBlocks, int format)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!blocks || !bc || (numBlocks % 8) != 0)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        return hostPipeline(ctx, reinterpret_cast<uint8_t *>(blocks), reinterpret_cast<const uint8_t *>(bc), numBlocks, 16, (format == 0 ? 64 : 128),
                            [&](void *dOut, const void *dIn, size_t n, hipStream_t st) { return decodeDevice(ctx, dOut, dIn, n, format, st); });
    }

    int cvttmi_decode_bc7_device(cvttmi_context *ctx, void *d_blocks, const void *d_bc, size_t numBlocks, void *hipStream)
    {
        return decodeDevice(ctx, d_blocks, d_bc, numBlocks, 0, hipStream);
    }
    int cvttmi_decode_bc6h_device(cvttmi_context *ctx, void *d_blocksF16, const void *d_bc, size_t numBlocks, int isSigned, void *hipStream)
    {
        return decodeDevice(ctx, d_blocksF16, d_bc, numBlocks, isSigned ? 2 : 1, hipStream);
    }
    int cvttmi_decode_bc7(cvttmi_context *ctx, uint8_t *blocks, const uint8_t *bc, size_t numBlocks)
    {
        return decodeHost(ctx, blocks, bc, numBlocks, 0);
    }
    int cvttmi_decode_bc6h(cvttmi_context *ctx, uint8_t *blocksF16, const uint8_t *bc, size_t numBlocks, int isSigned)
    {
        return decodeHost(ctx, blocksF16, bc, numBlocks, isSigned ? 2 : 1);
    }

    int cvttmi_selftest_arith(cvttmi_context *ctx, uint64_t count, uint64_t seed, uint64_t *divMismatches, uint64_t *sqrtMismatches)
    {
        if (!ctx || !divMismatches || !sqrtMismatches)
            return CVTTMI_E_INVALID;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        const uint32_t chunk = 1u << 20;
        uint32_t *dOps = NULL, *dRes = NULL;
        if (hipMalloc(&dOps, chunk * 8) != hipSuccess || hipMalloc(&dRes, chunk * 8) != hipSuccess)
        {
            hipFree(dOps);
            return fail(ctx, CVTTMI_E_HIP, "hipMalloc (selftest)");
        }
        std::vector<uint32_t> ops(chunk * 2), res(chunk * 2);
        uint64_t badDiv = 0, badSqrt = 0;
        for (uint64_t first = 0; first < count; first += chunk)
        {
            const uint32_t n = static_cast<uint32_t>(count - first < chunk ? count - first : chunk);
            e = cvttmi_launch_selftest(seed, first, n, dOps, dRes, NULL);
            if (e == hipSuccess) e = hipMemcpy(ops.data(), dOps, (size_t)n * 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(res.data(), dRes, (size_t)n * 8, hipMemcpyDeviceToHost);
            if (e != hipSuccess)
            {
                hipFree(dOps);
                hipFree(dRes);
                return fail(ctx, CVTTMI_E_HIP, "selftest kernel", e);
            }
            for (uint32_t i = 0; i < n; i++)
            {
                float a, b;
                memcpy(&a, &ops[2 * i], 4);
                memcpy(&b, &ops[2 * i + 1], 4);
                const float q = _mm_cvtss_f32(_mm_div_ss(_mm_set_ss(a), _mm_set_ss(b)));
                uint32_t absBits = ops[2 * i] & 0x7fffffffu;
                float absA;
                memcpy(&absA, &absBits, 4);
                const float r = _mm_cvtss_f32(_mm_sqrt_ss(_mm_set_ss(absA)));
                uint32_t qb, rb;
                memcpy(&qb, &q, 4);
                memcpy(&rb, &r, 4);
                const bool qNan = (qb & 0x7fffffffu) > 0x7f800000u, gNan = (res[2 * i] & 0x7fffffffu) > 0x7f800000u;
                if (!(qNan && gNan) && qb != res[2 * i])
                    badDiv++;
                if (rb != res[2 * i + 1])
                    badSqrt++;
            }
        }
        hipFree(dOps);
        hipFree(dRes);
        *divMismatches = badDiv;
        *sqrtMismatches = badSqrt;
        return CVTTMI_OK;
    }

    int cvttmi_set_exhaustive(cvttmi_context *ctx, int exhaustive)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        ctx->exhaustive = exhaustive != 0;
        return CVTTMI_OK;
    }

    // Diagnostic (tools/, not part of include/): blocks the last BC7 launch of this context handed to its second launch,
    // and how many of them found a slot.  Synchronises the device.
    int cvttmi_bc7_hard_stats(cvttmi_context *ctx, uint32_t *handedOver, uint32_t *slots)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        hipSetDevice(ctx->device);
        hipDeviceSynchronize();
        uint32_t v = 0;
        if (hipMemcpy(&v, ctx->dHardCount, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(ctx, CVTTMI_E_HIP, "hipMemcpy(hard count)");
        if (handedOver) *handedOver = v;
        if (slots) *slots = ctx->lastHardCap;
        return CVTTMI_OK;
    }

    int cvttmi_timing_enable(cvttmi_context *ctx, int enable)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        ctx->timing = enable != 0;
        ctx->totalMs = 0.0;
        ctx->launches = 0;
        return CVTTMI_OK;
    }

    int cvttmi_timing_read(cvttmi_context *ctx, double *totalMs, uint64_t *launches)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (totalMs) *totalMs = ctx->totalMs;
        if (launches) *launches = ctx->launches;
        return CVTTMI_OK;
    }

    int cvttmi_encode_bc7_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks,
                                 const cvttmi_options *options, const cvttmi_bc7_plan *plan, void *hipStream)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!d_out || !d_blocks || !options || !plan || (numBlocks % 8) != 0 || numBlocks > 0xfffffff0u)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        hipStream_t stream = static_cast<hipStream_t>(hipStream);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);

        const CvttBc7DevicePlan *dPlan = NULL;
        int planSlot = 0;
        int rc = stagePlan(ctx, plan, stream, &dPlan, &planSlot);
        if (rc != CVTTMI_OK)
            return rc;

        CvttBc7Args args;
        fillWeightArgs(options, args.w, args.wSq, args.rcpW);
        args.flags = options->flags;
        args.refineRounds = options->refineRoundsBC7;
        args.numBlocks = static_cast<uint32_t>(numBlocks);
        static const bool noProbe = getenv("CVTTMI_BC7_NOPROBE") && atoi(getenv("CVTTMI_BC7_NOPROBE")) != 0; // developer knob (A/B runs)
        static const bool noDedup = getenv("CVTTMI_BC7_NODEDUP") && atoi(getenv("CVTTMI_BC7_NODEDUP")) != 0; // developer knob (A/B runs)
        args.prune = ctx->exhaustive ? 0u : ((noProbe ? 3u : 1u) | (noDedup ? 4u : 0u));
        {
            // Slots for one block in a thousand: with the second-tier bounds few blocks of ordinary content keep many
            // partitions; the slots bound the extra wavefronts where every block does, and the second launch is sized
            // for all of them.  Below half a million blocks the two extra launches (about 25 us) cost more than the
            // tail they remove (measured: 1024^2 -5 %, 2048^2 -1 %, 4096^2 +1 %, near-opaque alpha noise +7 %).
            uint32_t cap = numBlocks >= (1u << 19) ? static_cast<uint32_t>(numBlocks / 1024) : 0u;
            if (ctx->hardCapOverride > 0)
                cap = static_cast<uint32_t>(ctx->hardCapOverride);
            cap = cap > cvttmi_context::kHardSlots ? cvttmi_context::kHardSlots : cap;
            args.hardCap = (args.prune && ctx->hardMin > 0) ? cap : 0u;
            args.hardMin = static_cast<uint32_t>(ctx->hardMin);
            args.hardDiv = static_cast<uint32_t>(ctx->hardDiv);
            args.hardCount = ctx->dHardCount;
            args.hardRec = ctx->dHardRec;
            args.hardCand = ctx->dHardCand;
            ctx->lastHardCap = args.hardCap;
        }
        {
            const double s3 = (double)args.wSq[0] + (double)args.wSq[1] + (double)args.wSq[2];
            const double s4 = s3 + (double)args.wSq[3];
            args.wSqSum3 = (args.wSq[0] + args.wSq[1]) + args.wSq[2];
            args.delta3 = static_cast<float>(0.5 * sqrt(s3) * 1.000001);
            args.delta4 = static_cast<float>(0.5 * sqrt(s4) * 1.000001);
        }

        // BC7_RespectPunchThrough records the error of every trial of a round of units; up to 2 refine rounds (the reference's
        // default) that table is 2 KB of LDS per round, beyond (the reference clamps refineRoundsBC7 only from below,
        // BC67.cpp:1044-1045) it lives in HBM: 2 KB per round and wave, so such a call goes in launches of as many waves as
        // 256 MB of table hold
        args.ptTrial = NULL;
        size_t blocksPerLaunch = numBlocks;
        if ((options->flags & CVTTMI_FLAG_BC7_RESPECT_PUNCHTHROUGH) && options->refineRoundsBC7 > 2) // kMaxPTRefine of bc7_kernel.hip
        {
            const size_t perWave = (size_t)32 * 16 * sizeof(float) * (size_t)options->refineRoundsBC7;
            size_t wavesPerLaunch = ctx->ptTableBytes / perWave; // 256 MB; developer knob CVTTMI_BC7_PT_TABLE_KB (tests: several launches)
            wavesPerLaunch = wavesPerLaunch < 1 ? 1 : wavesPerLaunch;
            const size_t wavesNeeded = (numBlocks + 15) / 16;
            wavesPerLaunch = wavesPerLaunch > wavesNeeded ? wavesNeeded : wavesPerLaunch;
            const size_t need = wavesPerLaunch * perWave;
            if (ctx->dPtTrialBytes < need)
            {
                if (ctx->dPtTrial)
                {
                    hipDeviceSynchronize();
                    hipFree(ctx->dPtTrial);
                    ctx->dPtTrial = NULL;
                    ctx->dPtTrialBytes = 0;
                }
                if ((e = hipMalloc(reinterpret_cast<void **>(&ctx->dPtTrial), need)) != hipSuccess)
                    return fail(ctx, CVTTMI_E_HIP, "hipMalloc(punch-through trial table)", e);
                ctx->dPtTrialBytes = need;
            }
            args.ptTrial = ctx->dPtTrial;
            blocksPerLaunch = wavesPerLaunch * 16;
        }

        // the hand-over list and the punch-through table exist once per context: launches that use them are ordered after
        // the previous one that did (the plan ring is protected by its own per-slot events)
        if (args.hardCap || args.ptTrial)
            orderAfterPrevious(ctx, stream);
        if (ctx->timing)
            hipEventRecord(ctx->evStart, stream);
        for (size_t first = 0; first < numBlocks; first += blocksPerLaunch)
        {
            const size_t n = (numBlocks - first) < blocksPerLaunch ? (numBlocks - first) : blocksPerLaunch;
            args.numBlocks = static_cast<uint32_t>(n);
            e = cvttmi_launch_bc7(static_cast<const uint8_t *>(d_blocks) + first * 64, static_cast<uint8_t *>(d_out) + first * 16, &args, ctx->dTables, dPlan, stream);
            if (e != hipSuccess)
            {
                // launches queued before this one still read the plan slot, the trial table and the hand-over list: nothing
                // may rewrite those before they have finished
                if (first != 0)
                    (void)hipStreamSynchronize(stream);
                return fail(ctx, CVTTMI_E_HIP, "bc7 kernel launch", e);
            }
        }
        if (args.hardCap || args.ptTrial)
            markLaunch(ctx, stream);
        markPlanUse(ctx, planSlot, stream);
        if (ctx->timing)
        {
            hipEventRecord(ctx->evStop, stream);
            hipEventSynchronize(ctx->evStop);
            float ms = 0.0f;
            hipEventElapsedTime(&ms, ctx->evStart, ctx->evStop);
            ctx->totalMs += ms;
            ctx->launches += 1;
        }
        return CVTTMI_OK;
    }

    // mode 0: EncodeETC2, 1: EncodeETC2RGBA, 2: EncodeETC2Alpha
    static int etc2Device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks,
                          const cvttmi_options *options, int mode, void *hipStream, const cvttmi_options *allocOptions = NULL)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!d_out || !d_blocks || !options || (numBlocks % 8) != 0 || numBlocks > 0xfffffff0u)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu); // lastError, timing state and work buffers are per context
        hipStream_t stream = static_cast<hipStream_t>(hipStream);
        CvttEtcArgs args;
        memset(&args, 0, sizeof(args));
        args.rw = options->redWeight;
        args.gw = options->greenWeight;
        args.bw = options->blueWeight;
        {
            // ETC2CompressionDataInternal constructor, reference ETC.cpp:3117-3145 (scalar binary32): the axes belong to
            // the Options of AllocETC2Data, not to those of the Encode call
            const cvttmi_options *ao = allocOptions ? allocOptions : options;
            volatile float cd[3] = {ao->redWeight, ao->greenWeight, ao->blueWeight};
            volatile float rotCD[3] = {cd[1], cd[2], cd[0]};
            volatile float offs = -(rotCD[0] * cd[0] + rotCD[1] * cd[1] + rotCD[2] * cd[2]) / (cd[0] * cd[0] + cd[1] * cd[1] + cd[2] * cd[2]);
            volatile float a0[3] = {rotCD[0] + cd[0] * offs, rotCD[1] + cd[1] * offs, rotCD[2] + cd[2] * offs};
            volatile float a1u[3] = {a0[1] * cd[2] - a0[2] * cd[1], a0[2] * cd[0] - a0[0] * cd[2], a0[0] * cd[1] - a0[1] * cd[0]};
            volatile float l0 = (a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2]);
            volatile float l1 = (a1u[0] * a1u[0] + a1u[1] * a1u[1] + a1u[2] * a1u[2]);
            volatile float ratio = static_cast<float>(sqrt(static_cast<double>(l0 / l1)));
            for (int i = 0; i < 3; i++)
            {
                args.axis0[i] = a0[i];
                args.axis1[i] = a1u[i] * ratio;
            }
        }
        args.flags = options->flags;
        args.numBlocks = static_cast<uint32_t>(numBlocks);
        {
            // punch-through: alpha below floor(clamp(threshold, 0, 1) * 255 + 1) is transparent (reference ETC.cpp:1672-1675)
            float t = options->threshold;
            t = 1.0f < t ? 1.0f : t;
            t = t > 0.0f ? t : 0.0f;
            args.alphaThreshold = static_cast<uint16_t>(floorf(t * 255.0f + 1.0f));
        }
        args.debug = getenv("CVTTMI_ETC_DEBUG_PTR") ? strtoull(getenv("CVTTMI_ETC_DEBUG_PTR"), NULL, 0) : 0;
        if (ctx->timing)
            hipEventRecord(ctx->evStart, stream);
        e = cvttmi_launch_etc2(d_blocks, d_out, &args, ctx->dTables, mode, stream);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_HIP, "etc2 kernel launch", e);
        if (ctx->timing)
        {
            hipEventRecord(ctx->evStop, stream);
            hipEventSynchronize(ctx->evStop);
            float ms = 0.0f;
            hipEventElapsedTime(&ms, ctx->evStart, ctx->evStop);
            ctx->totalMs += ms;
            ctx->launches += 1;
        }
        return CVTTMI_OK;
    }

    static int etc2Host(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                        const cvttmi_options *options, int mode, const cvttmi_options *allocOptions = NULL)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!out || !blocks || !options || (numBlocks % 8) != 0)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        return hostPipeline(ctx, reinterpret_cast<uint8_t *>(out), reinterpret_cast<const uint8_t *>(blocks), numBlocks, 64, (mode == 1 ? 16 : 8),
                            [&](void *dOut, const void *dIn, size_t n, hipStream_t st) { return etc2Device(ctx, dOut, dIn, n, options, mode, st, allocOptions); });
    }

    int cvttmi_encode_etc2_with_data_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options,
                                            const cvttmi_options *allocOptions, int kind, void *hipStream)
    {
        if (kind != CVTTMI_ETC2_RGB && kind != CVTTMI_ETC2_RGBA && kind != CVTTMI_ETC2_PUNCHTHROUGH)
            return ctx ? fail(ctx, CVTTMI_E_INVALID, "invalid argument") : CVTTMI_E_INVALID;
        return etc2Device(ctx, d_out, d_blocks, numBlocks, options, kind, hipStream, allocOptions);
    }
    int cvttmi_encode_etc2_with_data(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options,
                                     const cvttmi_options *allocOptions, int kind)
    {
        if (kind != CVTTMI_ETC2_RGB && kind != CVTTMI_ETC2_RGBA && kind != CVTTMI_ETC2_PUNCHTHROUGH)
            return ctx ? fail(ctx, CVTTMI_E_INVALID, "invalid argument") : CVTTMI_E_INVALID;
        return etc2Host(ctx, out, blocks, numBlocks, options, kind, allocOptions);
    }

    int cvttmi_encode_etc2_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options, void *hipStream)
    { return etc2Device(ctx, d_out, d_blocks, numBlocks, options, 0, hipStream); }
    int cvttmi_encode_etc2_rgba_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options, void *hipStream)
    { return etc2Device(ctx, d_out, d_blocks, numBlocks, options, 1, hipStream); }
    int cvttmi_encode_etc2_alpha_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options, void *hipStream)
    { return etc2Device(ctx, d_out, d_blocks, numBlocks, options, 2, hipStream); }
    int cvttmi_encode_etc2_punchthrough_alpha_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options, void *hipStream)
    { return etc2Device(ctx, d_out, d_blocks, numBlocks, options, 4, hipStream); }
    int cvttmi_encode_etc2_punchthrough_alpha(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options)
    { return etc2Host(ctx, out, blocks, numBlocks, options, 4); }
    int cvttmi_encode_etc1_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options, void *hipStream)
    { return etc2Device(ctx, d_out, d_blocks, numBlocks, options, 3, hipStream); }
    int cvttmi_encode_etc1(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options)
    { return etc2Host(ctx, out, blocks, numBlocks, options, 3); }
    int cvttmi_encode_etc2(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options)
    { return etc2Host(ctx, out, blocks, numBlocks, options, 0); }
    int cvttmi_encode_etc2_rgba(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options)
    { return etc2Host(ctx, out, blocks, numBlocks, options, 1); }
    int cvttmi_encode_etc2_alpha(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options)
    { return etc2Host(ctx, out, blocks, numBlocks, options, 2); }

    // EncodeETC2Alpha11 (reference ConvectionKernels_API.cpp:258-268): PixelBlockScalarS16 (32 B) -> 8 B; `options` is unused by
    // the reference's integer search but kept in the signature
    int cvttmi_encode_etc2_alpha11_device(cvttmi_context *ctx, void *d_out, const void *d_blocksS16, size_t numBlocks, int isSigned,
                                          const cvttmi_options *options, void *hipStream)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!d_out || !d_blocksS16 || !options || (numBlocks % 8) != 0 || numBlocks > 0xfffffff0u)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu); // lastError, timing state and work buffers are per context
        e = cvttmi_launch_eac11(d_blocksS16, d_out, static_cast<uint32_t>(numBlocks), isSigned, ctx->dTables, static_cast<hipStream_t>(hipStream));
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_HIP, "eac11 kernel launch", e);
        return CVTTMI_OK;
    }

    int cvttmi_encode_etc2_alpha11(cvttmi_context *ctx, uint8_t *out, const int16_t *blocksS16, size_t numBlocks, int isSigned,
                                   const cvttmi_options *options)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!out || !blocksS16 || !options || (numBlocks % 8) != 0)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        return hostPipeline(ctx, reinterpret_cast<uint8_t *>(out), reinterpret_cast<const uint8_t *>(blocksS16), numBlocks, 32, 8,
                            [&](void *dOut, const void *dIn, size_t n, hipStream_t st) { return cvttmi_encode_etc2_alpha11_device(ctx, dOut, dIn, n, isSigned, options, st); });
    }

    int cvttmi_encode_bc6h_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks,
                                  const cvttmi_options *options, int isSigned, void *hipStream)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!d_out || !d_blocks || !options || (numBlocks % 8) != 0 || numBlocks > 0xfffffff0u)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        hipStream_t stream = static_cast<hipStream_t>(hipStream);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        // (the search keeps its whole state on the chip: no work buffer, nothing shared between calls or streams)
        CvttBc6hArgs args;
        fillWeightArgs(options, args.w, args.wSq, args.rcpW);
        args.flags = options->flags;
        args.refineRounds = options->refineRoundsBC6H;
        args.seedPoints = options->seedPoints;
        // one launch per 2^28 blocks (the grid is blocks / 64 workgroups)
        const size_t kChunk = (size_t)1 << 28;
        if (ctx->timing)
            hipEventRecord(ctx->evStart, stream);
        for (size_t first = 0; first < numBlocks; first += kChunk)
        {
            const size_t n = (numBlocks - first) < kChunk ? (numBlocks - first) : kChunk;
            args.numBlocks = static_cast<uint32_t>(n);
            e = cvttmi_launch_bc6h(static_cast<const uint8_t *>(d_blocks) + first * 128, static_cast<uint8_t *>(d_out) + first * 16,
                                   &args, ctx->dTables, isSigned, stream);
            if (e != hipSuccess)
                return fail(ctx, CVTTMI_E_HIP, "bc6h kernel launch", e);
        }
        if (ctx->timing)
        {
            hipEventRecord(ctx->evStop, stream);
            hipEventSynchronize(ctx->evStop);
            float ms = 0.0f;
            hipEventElapsedTime(&ms, ctx->evStart, ctx->evStop);
            ctx->totalMs += ms;
            ctx->launches += 1;
        }
        return CVTTMI_OK;
    }

    int cvttmi_encode_bc6h(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                           const cvttmi_options *options, int isSigned)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!out || !blocks || !options || (numBlocks % 8) != 0)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        return hostPipeline(ctx, reinterpret_cast<uint8_t *>(out), reinterpret_cast<const uint8_t *>(blocks), numBlocks, 128, 16,
                            [&](void *dOut, const void *dIn, size_t n, hipStream_t st) { return cvttmi_encode_bc6h_device(ctx, dOut, dIn, n, options, isSigned, st); });
    }

    int cvttmi_encode_bc1_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks,
                                 const cvttmi_options *options, void *hipStream)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!d_out || !d_blocks || !options || (numBlocks % 8) != 0 || numBlocks > 0xfffffff0u)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu); // lastError, timing state and work buffers are per context
        hipStream_t stream = static_cast<hipStream_t>(hipStream);
        CvttBc1Args args;
        fillWeightArgs(options, args.w, args.wSq, args.rcpW);
        args.flags = options->flags;
        args.refineRounds = options->refineRoundsS3TC;
        args.seedPoints = options->seedPoints;
        // S3TC.cpp:748: MakeUInt15(static_cast<uint16_t>(floor(alphaThreshold * 255.0f + 0.5f)))
        args.threshold = static_cast<int16_t>(static_cast<uint16_t>(static_cast<int32_t>(floor(options->threshold * 255.0f + 0.5f))));
        args.numBlocks = static_cast<uint32_t>(numBlocks);
        args.alphaTest = 1u;
        args.outStride = 8u;
        args.outOffset = 0u;
        if (ctx->timing)
            hipEventRecord(ctx->evStart, stream);
        e = cvttmi_launch_bc1(d_blocks, d_out, &args, ctx->dTables, stream);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_HIP, "bc1 kernel launch", e);
        if (ctx->timing)
        {
            hipEventRecord(ctx->evStop, stream);
            hipEventSynchronize(ctx->evStop);
            float ms = 0.0f;
            hipEventElapsedTime(&ms, ctx->evStart, ctx->evStop);
            ctx->totalMs += ms;
            ctx->launches += 1;
        }
        return CVTTMI_OK;
    }

    // BC2 / BC3 / BC4 / BC5 (reference ConvectionKernels_API.cpp:101-199).  format: 2 = BC2, 3 = BC3, 4 = BC4U, 5 = BC4S,
    // 6 = BC5U, 7 = BC5S; signed formats read PixelBlockS8.
    static int s3tcDevice(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options,
                          int format, void *hipStream)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!d_out || !d_blocks || !options || (numBlocks % 8) != 0 || numBlocks > 0xfffffff0u || format < 2 || format > 7)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu); // lastError, timing state and work buffers are per context
        hipStream_t stream = static_cast<hipStream_t>(hipStream);
        const uint32_t n = static_cast<uint32_t>(numBlocks);
        if (format == 2 || format == 3)
        {
            CvttBc1Args args;
            fillWeightArgs(options, args.w, args.wSq, args.rcpW);
            args.flags = options->flags;
            args.refineRounds = options->refineRoundsS3TC;
            args.seedPoints = options->seedPoints;
            args.threshold = 0;
            args.numBlocks = n;
            args.alphaTest = 0u;
            args.outStride = 16u;
            args.outOffset = 8u;
            if ((e = cvttmi_launch_bc1(d_blocks, d_out, &args, ctx->dTables, stream)) != hipSuccess)
                return fail(ctx, CVTTMI_E_HIP, "s3tc colour kernel launch", e);
            e = cvttmi_launch_s3tc_alpha(d_blocks, d_out, n, 3, 16, 0, 0, format == 2, options->seedPoints, options->refineRoundsIIC, ctx->dTables, stream);
        }
        else if (format == 4 || format == 5)
            e = cvttmi_launch_s3tc_alpha(d_blocks, d_out, n, 0, 8, 0, format == 5, 0, options->seedPoints, options->refineRoundsIIC, ctx->dTables, stream);
        else
        {
            e = cvttmi_launch_s3tc_alpha(d_blocks, d_out, n, 0, 16, 0, format == 7, 0, options->seedPoints, options->refineRoundsIIC, ctx->dTables, stream);
            if (e == hipSuccess)
                e = cvttmi_launch_s3tc_alpha(d_blocks, d_out, n, 1, 16, 8, format == 7, 0, options->seedPoints, options->refineRoundsIIC, ctx->dTables, stream);
        }
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_HIP, "s3tc alpha kernel launch", e);
        return CVTTMI_OK;
    }

    static int s3tcHost(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options, int format)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!out || !blocks || !options || (numBlocks % 8) != 0)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        return hostPipeline(ctx, reinterpret_cast<uint8_t *>(out), reinterpret_cast<const uint8_t *>(blocks), numBlocks, 64, ((format == 4 || format == 5) ? 8 : 16),
                            [&](void *dOut, const void *dIn, size_t n, hipStream_t st) { return s3tcDevice(ctx, dOut, dIn, n, options, format, st); });
    }

    int cvttmi_encode_bc2_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options, void *hipStream)
    { return s3tcDevice(ctx, d_out, d_blocks, numBlocks, options, 2, hipStream); }
    int cvttmi_encode_bc3_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options, void *hipStream)
    { return s3tcDevice(ctx, d_out, d_blocks, numBlocks, options, 3, hipStream); }
    int cvttmi_encode_bc4_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options, int isSigned, void *hipStream)
    { return s3tcDevice(ctx, d_out, d_blocks, numBlocks, options, isSigned ? 5 : 4, hipStream); }
    int cvttmi_encode_bc5_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options, int isSigned, void *hipStream)
    { return s3tcDevice(ctx, d_out, d_blocks, numBlocks, options, isSigned ? 7 : 6, hipStream); }
    int cvttmi_encode_bc2(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options)
    { return s3tcHost(ctx, out, blocks, numBlocks, options, 2); }
    int cvttmi_encode_bc3(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options)
    { return s3tcHost(ctx, out, blocks, numBlocks, options, 3); }
    int cvttmi_encode_bc4(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options, int isSigned)
    { return s3tcHost(ctx, out, blocks, numBlocks, options, isSigned ? 5 : 4); }
    int cvttmi_encode_bc5(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options, int isSigned)
    { return s3tcHost(ctx, out, blocks, numBlocks, options, isSigned ? 7 : 6); }

    int cvttmi_encode_bc1(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                          const cvttmi_options *options)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!out || !blocks || !options || (numBlocks % 8) != 0)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        return hostPipeline(ctx, reinterpret_cast<uint8_t *>(out), reinterpret_cast<const uint8_t *>(blocks), numBlocks, 64, 8,
                            [&](void *dOut, const void *dIn, size_t n, hipStream_t st) { return cvttmi_encode_bc1_device(ctx, dOut, dIn, n, options, st); });
    }

    int cvttmi_encode_bc7(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                          const cvttmi_options *options, const cvttmi_bc7_plan *plan)
    {
        if (!ctx)
            return CVTTMI_E_INVALID;
        if (!out || !blocks || !options || !plan || (numBlocks % 8) != 0)
            return fail(ctx, CVTTMI_E_INVALID, "invalid argument");
        if (numBlocks == 0)
            return CVTTMI_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess)
            return fail(ctx, CVTTMI_E_NO_DEVICE, "hipSetDevice", e);
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        return hostPipeline(ctx, reinterpret_cast<uint8_t *>(out), reinterpret_cast<const uint8_t *>(blocks), numBlocks, 64, 16,
                            [&](void *dOut, const void *dIn, size_t n, hipStream_t st) { return cvttmi_encode_bc7_device(ctx, dOut, dIn, n, options, plan, st); },
                            true /* reads every PixelBlock once: page-locked caller memory is used in place */);
    }
}
