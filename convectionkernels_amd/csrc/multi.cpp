// Several devices behind the C ABI (include/cvtt_mi355x.h, "one job on several devices"): the north-star's "large images
// shard by block row across the GPUs of one node" for callers of the C / C++ interface, who have no torch.distributed.
// The unit of the path is the reference's group of 8 blocks (ConvectionKernels.h:71, 241); groups are independent, so the
// search needs no exchange between devices: a job is cut into contiguous, group-aligned ranges of whole block rows (the rule
// of convectionkernels_amd/sharding.py, shard_block_rows), every range is encoded by its own context on its own device
// from its own host thread through the host-pointer entry points (pinned staging, PCIe pipelined with the search), and the
// packed blocks land in the caller's output buffer at the range's offset -- that copy is the gather.  (Between PROCESSES,
// one per GPU, the packed output is gathered with RCCL send/recv over xGMI: sharding.py / bench.py --gpus N.)
// The device list may name a device more than once (a context each): that is how a one-GPU box exercises every line here.
//
// Device-resident callers (cvttmi_multi_encode_device): every shard's PixelBlocks already sit in the HBM of the device that
// searches them; shard r's packed blocks go to the root buffer on devices[0] -- written there directly by the shards that run on
// the root device, copied with hipMemcpyPeerAsync (xGMI, peer access enabled where the pair allows it) from a buffer on their own
// device by the others.  That copy is the north-star's "gather over xGMI of the packed output" for one process.
//
// No exception leaves an extern "C" function: every body that can allocate is wrapped (GUARD_BEGIN / GUARD_END).
#include "../../include/cvtt_mi355x.h"

#include <hip/hip_runtime_api.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct cvttmi_multi
{
    std::vector<int> devices;
    std::vector<cvttmi_context *> ctx;
    std::mutex mu; // one job at a time per handle; also guards lastError / lastFirst / lastLast and the setters
    std::string lastError;
    std::vector<size_t> lastFirst, lastLast; // the shard table of the most recent job (tests, logs)
    // cvttmi_multi_encode_device: per shard a stream on its device and, for shards away from the root device, a buffer there
    std::vector<hipStream_t> streams;
    std::vector<void *> stage;
    std::vector<size_t> stageBytes;
    bool forceStage; // CVTTMI_MULTI_FORCE_STAGE=1: shards on the root device take the staged + peer-copy route too (one-GPU tests)
};

namespace
{
    // the error text of the calling thread's most recent failed call (the stateless forms have no handle to ask)
    thread_local char t_lastError[256] = "";
    void noteError(cvttmi_multi *m, const std::string &text)
    {
        snprintf(t_lastError, sizeof(t_lastError), "%s", text.c_str());
        if (m)
            m->lastError = text;
    }
    void noteErrorNoThrow(const char *text) { snprintf(t_lastError, sizeof(t_lastError), "%s", text); }
}
#define GUARD_BEGIN try {
#define GUARD_END(code) } catch (const std::exception &e) { noteErrorNoThrow(e.what()); return (code); } catch (...) { noteErrorNoThrow("exception"); return (code); }

extern "C"
{
    int cvttmi_shard_block_rows(size_t blockRows, size_t blocksPerRow, int rank, int world, size_t *first, size_t *last)
    {
        if (!first || !last || world < 1 || rank < 0 || rank >= world)
            return CVTTMI_E_INVALID;
        const size_t G = 8;
        const size_t total = blockRows * blocksPerRow;
        size_t lo = (blockRows * (size_t)rank / (size_t)world) * blocksPerRow;
        size_t hi = (blockRows * ((size_t)rank + 1) / (size_t)world) * blocksPerRow;
        lo = (lo + G - 1) / G * G;
        hi = (hi + G - 1) / G * G;
        lo = lo > total ? total : lo;
        hi = (rank == world - 1 || hi > total) ? total : hi;
        *first = lo;
        *last = hi < lo ? lo : hi;
        return CVTTMI_OK;
    }

    int cvttmi_multi_create(cvttmi_multi **out, const int *devices, int numDevices)
    {
        if (!out || !devices || numDevices < 1 || numDevices > 64)
            return CVTTMI_E_INVALID;
        *out = NULL;
        cvttmi_multi *m = NULL;
        try
        {
            m = new cvttmi_multi();
            m->devices.reserve(numDevices);
            m->ctx.reserve(numDevices);
            m->streams.assign(numDevices, (hipStream_t)NULL);
            m->stage.assign(numDevices, (void *)NULL);
            m->stageBytes.assign(numDevices, 0);
            m->lastFirst.assign(numDevices, 0);
            m->lastLast.assign(numDevices, 0);
            const char *fs = getenv("CVTTMI_MULTI_FORCE_STAGE");
            m->forceStage = fs && fs[0] == '1';
            for (int i = 0; i < numDevices; i++)
            {
                cvttmi_context *c = NULL;
                const int rc = cvttmi_create(&c, devices[i]);
                if (rc != CVTTMI_OK)
                {
                    cvttmi_multi_destroy(m);
                    return rc;
                }
                m->devices.push_back(devices[i]); // (reserved above: no allocation)
                m->ctx.push_back(c);
            }
        }
        catch (...)
        {
            if (m)
                cvttmi_multi_destroy(m);
            noteErrorNoThrow("out of memory");
            return CVTTMI_E_HIP;
        }
        *out = m;
        return CVTTMI_OK;
    }

    void cvttmi_multi_destroy(cvttmi_multi *m)
    {
        if (!m)
            return;
        for (size_t k = 0; k < m->streams.size(); k++)
            if (m->streams[k] || m->stage[k])
            {
                if (k < m->devices.size() && hipSetDevice(m->devices[k]) == hipSuccess)
                {
                    if (m->streams[k]) (void)hipStreamDestroy(m->streams[k]);
                    if (m->stage[k]) (void)hipFree(m->stage[k]);
                }
            }
        for (size_t k = 0; k < m->ctx.size(); k++)
            cvttmi_destroy(m->ctx[k]);
        delete m;
    }

    // the text of the most recent failure: of the handle when one is given (copied under its lock into the calling thread's
    // buffer, so the pointer stays valid while other threads run jobs), else of the calling thread's last failed multi-device
    // call -- which is how the stateless *_multi forms report
    const char *cvttmi_multi_last_error(const cvttmi_multi *m)
    {
        if (m)
        {
            cvttmi_multi *mm = const_cast<cvttmi_multi *>(m);
            std::lock_guard<std::mutex> lock(mm->mu);
            snprintf(t_lastError, sizeof(t_lastError), "%s", mm->lastError.c_str());
        }
        return t_lastError;
    }
    int cvttmi_multi_num_devices(const cvttmi_multi *m) { return m ? (int)m->ctx.size() : 0; }
    cvttmi_context *cvttmi_multi_context(cvttmi_multi *m, int index) { return (m && index >= 0 && index < (int)m->ctx.size()) ? m->ctx[index] : NULL; }

    int cvttmi_multi_last_shard(const cvttmi_multi *m, int index, size_t *first, size_t *last)
    {
        if (!m || !first || !last || index < 0 || index >= (int)m->lastFirst.size())
            return CVTTMI_E_INVALID;
        *first = m->lastFirst[index];
        *last = m->lastLast[index];
        return CVTTMI_OK;
    }

    int cvttmi_multi_set_rcp_table(cvttmi_multi *m, const float lut[17])
    {
        if (!m)
            return CVTTMI_E_INVALID;
        std::lock_guard<std::mutex> lock(m->mu);
        for (size_t k = 0; k < m->ctx.size(); k++)
        {
            const int rc = cvttmi_set_rcp_table(m->ctx[k], lut);
            if (rc != CVTTMI_OK)
                return rc;
        }
        return CVTTMI_OK;
    }

    int cvttmi_multi_set_exhaustive(cvttmi_multi *m, int exhaustive)
    {
        if (!m)
            return CVTTMI_E_INVALID;
        std::lock_guard<std::mutex> lock(m->mu);
        for (size_t k = 0; k < m->ctx.size(); k++)
            cvttmi_set_exhaustive(m->ctx[k], exhaustive);
        return CVTTMI_OK;
    }

}

namespace
{
    struct ErrText { char t[160]; };

    bool formatSizes(int format, size_t &inBpb, size_t &outBpb)
    {
        inBpb = 64;
        outBpb = 16;
        switch (format)
        {
        case CVTTMI_FMT_BC7: return true;
        case CVTTMI_FMT_BC1: outBpb = 8; return true;
        case CVTTMI_FMT_BC6HU: case CVTTMI_FMT_BC6HS: inBpb = 128; return true;
        case CVTTMI_FMT_ETC2_RGB: outBpb = 8; return true;
        case CVTTMI_FMT_ETC2_RGBA: return true;
        }
        return false;
    }

    // one shard, host buffers: the host-pointer entry point of its context (pinned staging, PCIe pipelined with the search)
    int encodeHostShard(cvttmi_context *c, int format, uint8_t *o, const uint8_t *b, size_t n, const cvttmi_options *options, const cvttmi_bc7_plan *plan)
    {
        switch (format)
        {
        case CVTTMI_FMT_BC7: return cvttmi_encode_bc7(c, o, b, n, options, plan);
        case CVTTMI_FMT_BC1: return cvttmi_encode_bc1(c, o, b, n, options);
        case CVTTMI_FMT_BC6HU: return cvttmi_encode_bc6h(c, o, b, n, options, 0);
        case CVTTMI_FMT_BC6HS: return cvttmi_encode_bc6h(c, o, b, n, options, 1);
        case CVTTMI_FMT_ETC2_RGB: return cvttmi_encode_etc2(c, o, b, n, options);
        case CVTTMI_FMT_ETC2_RGBA: return cvttmi_encode_etc2_rgba(c, o, b, n, options);
        }
        return CVTTMI_E_INVALID;
    }

    int encodeDeviceShard(cvttmi_context *c, int format, void *o, const void *b, size_t n, const cvttmi_options *options, const cvttmi_bc7_plan *plan, hipStream_t st)
    {
        switch (format)
        {
        case CVTTMI_FMT_BC7: return cvttmi_encode_bc7_device(c, o, b, n, options, plan, st);
        case CVTTMI_FMT_BC1: return cvttmi_encode_bc1_device(c, o, b, n, options, st);
        case CVTTMI_FMT_BC6HU: return cvttmi_encode_bc6h_device(c, o, b, n, options, 0, st);
        case CVTTMI_FMT_BC6HS: return cvttmi_encode_bc6h_device(c, o, b, n, options, 1, st);
        case CVTTMI_FMT_ETC2_RGB: return cvttmi_encode_etc2_device(c, o, b, n, options, st);
        case CVTTMI_FMT_ETC2_RGBA: return cvttmi_encode_etc2_rgba_device(c, o, b, n, options, st);
        }
        return CVTTMI_E_INVALID;
    }

    // Shard r of a device-resident job: search on devices[r] (launch on the shard's own stream), then -- unless the shard runs on
    // the root device and wrote its slice of the root buffer directly -- one peer copy of the packed blocks to the root buffer.
    // `err` receives the failing HIP call's text (a fixed buffer: this runs on a worker thread and must not throw).
    int deviceShard(cvttmi_multi *m, int r, int format, uint8_t *rootOut, const void *d_in, size_t n, size_t outBpb,
                    const cvttmi_options *options, const cvttmi_bc7_plan *plan, char (&err)[160])
    {
        err[0] = 0;
        const int dev = m->devices[r], root = m->devices[0];
        hipError_t e = hipSetDevice(dev);
        if (e != hipSuccess) { snprintf(err, sizeof(err), "hipSetDevice(%d): %s", dev, hipGetErrorString(e)); return CVTTMI_E_NO_DEVICE; }
        if (!m->streams[r])
        {
            e = hipStreamCreateWithFlags(&m->streams[r], hipStreamNonBlocking);
            if (e != hipSuccess) { snprintf(err, sizeof(err), "hipStreamCreate: %s", hipGetErrorString(e)); return CVTTMI_E_HIP; }
        }
        const hipStream_t st = m->streams[r];
        const size_t bytes = n * outBpb;
        const bool direct = dev == root && !m->forceStage;
        void *target = rootOut;
        if (!direct)
        {
            if (m->stageBytes[r] < bytes)
            {
                if (m->stage[r]) (void)hipFree(m->stage[r]);
                m->stage[r] = NULL;
                m->stageBytes[r] = 0;
                e = hipMalloc(&m->stage[r], bytes);
                if (e != hipSuccess) { snprintf(err, sizeof(err), "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return CVTTMI_E_HIP; }
                m->stageBytes[r] = bytes;
            }
            target = m->stage[r];
            if (dev != root)
            {
                // direct xGMI writes where the pair allows them; where not (or already enabled) the copy below still works
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, dev, root) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(root, 0);
                (void)hipGetLastError();
            }
        }
        const int rc = encodeDeviceShard(m->ctx[r], format, target, d_in, n, options, plan, st);
        if (rc != CVTTMI_OK) { snprintf(err, sizeof(err), "%s", cvttmi_last_error(m->ctx[r])); return rc; }
        if (!direct)
        {
            e = hipMemcpyPeerAsync(rootOut, root, target, dev, bytes, st);
            if (e != hipSuccess) { snprintf(err, sizeof(err), "hipMemcpyPeerAsync(%d -> %d): %s", dev, root, hipGetErrorString(e)); return CVTTMI_E_HIP; }
        }
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) { snprintf(err, sizeof(err), "hipStreamSynchronize: %s", hipGetErrorString(e)); return CVTTMI_E_HIP; }
        return CVTTMI_OK;
    }

    // host job (blocks != NULL) or device-resident job (d_shards != NULL); may throw std::bad_alloc / std::system_error: the
    // extern "C" callers catch
    int runJob(cvttmi_multi *m, int format, uint8_t *out, const uint8_t *blocks, const void *const *d_shards, size_t numBlocks, size_t blocksPerRow,
               const cvttmi_options *options, const cvttmi_bc7_plan *plan)
    {
        if (!m)
        {
            noteErrorNoThrow("no handle");
            return CVTTMI_E_INVALID;
        }
        std::lock_guard<std::mutex> lock(m->mu);
        size_t inBpb = 64, outBpb = 16;
        if (!formatSizes(format, inBpb, outBpb)) { noteError(m, "unknown format"); return CVTTMI_E_INVALID; }
        if (format == CVTTMI_FMT_BC7 && !plan) { noteError(m, "BC7 needs a plan"); return CVTTMI_E_INVALID; }
        if (!out || (!blocks && !d_shards) || !options || (numBlocks % 8) != 0)
        {
            noteError(m, "invalid argument");
            return CVTTMI_E_INVALID;
        }
        // rows: the caller's block rows when it names them (a tiled image), else one group per row
        if (blocksPerRow == 0 || numBlocks % blocksPerRow != 0)
            blocksPerRow = 8;
        const size_t rows = numBlocks / blocksPerRow;
        const int world = (int)m->ctx.size();
        std::vector<int> rcs(world, CVTTMI_OK);
        std::vector<ErrText> errs(d_shards ? world : 0);
        std::vector<std::thread> workers;
        workers.reserve(world);
        bool spawnFailed = false;
        for (int r = 0; r < world && !spawnFailed; r++)
        {
            size_t lo = 0, hi = 0;
            cvttmi_shard_block_rows(rows, blocksPerRow, r, world, &lo, &hi);
            m->lastFirst[r] = lo;
            m->lastLast[r] = hi;
            if (hi <= lo)
                continue;
            if (d_shards && !d_shards[r])
            {
                rcs[r] = CVTTMI_E_INVALID;
                snprintf(errs[r].t, sizeof(errs[r].t), "d_shards[%d] is NULL for a shard of %zu blocks", r, hi - lo);
                continue;
            }
            try
            {
                workers.push_back(std::thread([=, &rcs, &errs]() {
                    if (d_shards)
                        rcs[r] = deviceShard(m, r, format, out + lo * outBpb, d_shards[r], hi - lo, outBpb, options, plan, errs[r].t);
                    else
                        rcs[r] = encodeHostShard(m->ctx[r], format, out + lo * outBpb, blocks + lo * inBpb, hi - lo, options, plan);
                }));
            }
            catch (...) // no thread to be had: finish what was started, then report
            {
                spawnFailed = true;
            }
        }
        for (size_t i = 0; i < workers.size(); i++)
            workers[i].join();
        if (spawnFailed)
        {
            noteError(m, "could not start a worker thread");
            return CVTTMI_E_HIP;
        }
        for (int r = 0; r < world; r++)
            if (rcs[r] != CVTTMI_OK)
            {
                char head[64];
                snprintf(head, sizeof(head), "shard %d (device %d): ", r, m->devices[r]);
                noteError(m, std::string(head) + (d_shards ? errs[r].t : cvttmi_last_error(m->ctx[r])));
                return rcs[r];
            }
        return CVTTMI_OK;
    }
}

extern "C"
{
    int cvttmi_multi_encode(cvttmi_multi *m, int format, uint8_t *out, const uint8_t *blocks, size_t numBlocks, size_t blocksPerRow,
                            const cvttmi_options *options, const cvttmi_bc7_plan *plan)
    {
        GUARD_BEGIN
        return runJob(m, format, out, blocks, NULL, numBlocks, blocksPerRow, options, plan);
        GUARD_END(CVTTMI_E_HIP)
    }

    int cvttmi_multi_encode_device(cvttmi_multi *m, int format, void *d_out, const void *const *d_shards, size_t numBlocks, size_t blocksPerRow,
                                   const cvttmi_options *options, const cvttmi_bc7_plan *plan)
    {
        GUARD_BEGIN
        return runJob(m, format, (uint8_t *)d_out, NULL, d_shards, numBlocks, blocksPerRow, options, plan);
        GUARD_END(CVTTMI_E_HIP)
    }
}

namespace
{
    // the handles behind the stateless *_multi calls: one per device list, kept for the life of the process
    std::mutex g_mu;
    std::map<std::vector<int>, cvttmi_multi *> g_handles;

    int statelessEncode(const int *devices, int numDevices, int format, uint8_t *out, const uint8_t *blocks, size_t numBlocks, size_t blocksPerRow,
                        const cvttmi_options *options, const cvttmi_bc7_plan *plan)
    {
        if (!devices || numDevices < 1)
            return CVTTMI_E_INVALID;
        GUARD_BEGIN
        cvttmi_multi *m = NULL;
        {
            std::lock_guard<std::mutex> lock(g_mu);
            const std::vector<int> key(devices, devices + numDevices);
            std::map<std::vector<int>, cvttmi_multi *>::iterator it = g_handles.find(key);
            if (it == g_handles.end())
            {
                const int rc = cvttmi_multi_create(&m, devices, numDevices);
                if (rc != CVTTMI_OK)
                    return rc;
                try
                {
                    g_handles[key] = m;
                }
                catch (...)
                {
                    cvttmi_multi_destroy(m);
                    throw;
                }
            }
            else
                m = it->second;
        }
        return cvttmi_multi_encode(m, format, out, blocks, numBlocks, blocksPerRow, options, plan);
        GUARD_END(CVTTMI_E_HIP)
    }
}

extern "C"
{
    int cvttmi_encode_bc7_multi(const int *devices, int numDevices, uint8_t *out, const uint8_t *blocks, size_t numBlocks, size_t blocksPerRow,
                                const cvttmi_options *options, const cvttmi_bc7_plan *plan)
    {
        return statelessEncode(devices, numDevices, CVTTMI_FMT_BC7, out, blocks, numBlocks, blocksPerRow, options, plan);
    }
    int cvttmi_encode_bc1_multi(const int *devices, int numDevices, uint8_t *out, const uint8_t *blocks, size_t numBlocks, size_t blocksPerRow,
                                const cvttmi_options *options)
    {
        return statelessEncode(devices, numDevices, CVTTMI_FMT_BC1, out, blocks, numBlocks, blocksPerRow, options, NULL);
    }
    int cvttmi_encode_bc6h_multi(const int *devices, int numDevices, uint8_t *out, const uint8_t *blocks, size_t numBlocks, size_t blocksPerRow,
                                 const cvttmi_options *options, int isSigned)
    {
        return statelessEncode(devices, numDevices, isSigned ? CVTTMI_FMT_BC6HS : CVTTMI_FMT_BC6HU, out, blocks, numBlocks, blocksPerRow, options, NULL);
    }
    int cvttmi_encode_etc2_rgba_multi(const int *devices, int numDevices, uint8_t *out, const uint8_t *blocks, size_t numBlocks, size_t blocksPerRow,
                                      const cvttmi_options *options)
    {
        return statelessEncode(devices, numDevices, CVTTMI_FMT_ETC2_RGBA, out, blocks, numBlocks, blocksPerRow, options, NULL);
    }
}
