// Several devices behind the C ABI (include/cvtt_mi355x.h, "one job on several devices"): the north-star's "large images
// shard by block row across the GPUs of one node" for callers of the C / C++ interface, who have no torch.distributed.
// The unit of the path is the reference's group of 8 blocks (ConvectionKernels.h:71, 241); groups are independent, so the
// search needs no exchange between devices: a job is cut into contiguous, group-aligned ranges of whole block rows (the rule
// of convectionkernels_amd/sharding.py, shard_block_rows), every range is encoded by its own context on its own device
// from its own host thread through the host-pointer entry points (pinned staging, PCIe pipelined with the search), and the
// packed blocks land in the caller's output buffer at the range's offset -- that copy is the gather.  (Between PROCESSES,
// one per GPU, the packed output is gathered with RCCL send/recv over xGMI: sharding.py / bench.py --gpus N.)
// The device list may name a device more than once (a context each): that is how a one-GPU box exercises every line here.
#include "../../include/cvtt_mi355x.h"

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
    std::mutex mu; // one job at a time per handle
    std::string lastError;
    std::vector<size_t> lastFirst, lastLast; // the shard table of the most recent job (tests, logs)
};

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
        cvttmi_multi *m = new cvttmi_multi();
        for (int i = 0; i < numDevices; i++)
        {
            cvttmi_context *c = NULL;
            const int rc = cvttmi_create(&c, devices[i]);
            if (rc != CVTTMI_OK)
            {
                for (size_t k = 0; k < m->ctx.size(); k++)
                    cvttmi_destroy(m->ctx[k]);
                delete m;
                *out = NULL;
                return rc;
            }
            m->devices.push_back(devices[i]);
            m->ctx.push_back(c);
        }
        *out = m;
        return CVTTMI_OK;
    }

    void cvttmi_multi_destroy(cvttmi_multi *m)
    {
        if (!m)
            return;
        for (size_t k = 0; k < m->ctx.size(); k++)
            cvttmi_destroy(m->ctx[k]);
        delete m;
    }

    const char *cvttmi_multi_last_error(const cvttmi_multi *m) { return m ? m->lastError.c_str() : "no handle"; }
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
        for (size_t k = 0; k < m->ctx.size(); k++)
            cvttmi_set_exhaustive(m->ctx[k], exhaustive);
        return CVTTMI_OK;
    }

    int cvttmi_multi_encode(cvttmi_multi *m, int format, uint8_t *out, const uint8_t *blocks, size_t numBlocks, size_t blocksPerRow,
                            const cvttmi_options *options, const cvttmi_bc7_plan *plan)
    {
        if (!m)
            return CVTTMI_E_INVALID;
        std::lock_guard<std::mutex> lock(m->mu);
        size_t inBpb = 64, outBpb = 16;
        switch (format)
        {
        case CVTTMI_FMT_BC7: if (!plan) { m->lastError = "BC7 needs a plan"; return CVTTMI_E_INVALID; } break;
        case CVTTMI_FMT_BC1: outBpb = 8; break;
        case CVTTMI_FMT_BC6HU: case CVTTMI_FMT_BC6HS: inBpb = 128; break;
        case CVTTMI_FMT_ETC2_RGB: outBpb = 8; break;
        case CVTTMI_FMT_ETC2_RGBA: break;
        default: m->lastError = "unknown format"; return CVTTMI_E_INVALID;
        }
        if (!out || !blocks || !options || (numBlocks % 8) != 0)
        {
            m->lastError = "invalid argument";
            return CVTTMI_E_INVALID;
        }
        // rows: the caller's block rows when it names them (a tiled image), else one group per row
        if (blocksPerRow == 0 || numBlocks % blocksPerRow != 0)
            blocksPerRow = 8;
        const size_t rows = numBlocks / blocksPerRow;
        const int world = (int)m->ctx.size();
        m->lastFirst.assign(world, 0);
        m->lastLast.assign(world, 0);
        std::vector<int> rcs(world, CVTTMI_OK);
        std::vector<std::thread> workers;
        bool spawnFailed = false;
        for (int r = 0; r < world && !spawnFailed; r++)
        {
            size_t lo = 0, hi = 0;
            cvttmi_shard_block_rows(rows, blocksPerRow, r, world, &lo, &hi);
            m->lastFirst[r] = lo;
            m->lastLast[r] = hi;
            if (hi <= lo)
                continue;
            try
            {
            workers.push_back(std::thread([=, &rcs]() {
                cvttmi_context *c = m->ctx[r];
                uint8_t *o = out + lo * outBpb;
                const uint8_t *b = blocks + lo * inBpb;
                const size_t n = hi - lo;
                int rc = CVTTMI_E_INVALID;
                switch (format)
                {
                case CVTTMI_FMT_BC7: rc = cvttmi_encode_bc7(c, o, b, n, options, plan); break;
                case CVTTMI_FMT_BC1: rc = cvttmi_encode_bc1(c, o, b, n, options); break;
                case CVTTMI_FMT_BC6HU: rc = cvttmi_encode_bc6h(c, o, b, n, options, 0); break;
                case CVTTMI_FMT_BC6HS: rc = cvttmi_encode_bc6h(c, o, b, n, options, 1); break;
                case CVTTMI_FMT_ETC2_RGB: rc = cvttmi_encode_etc2(c, o, b, n, options); break;
                case CVTTMI_FMT_ETC2_RGBA: rc = cvttmi_encode_etc2_rgba(c, o, b, n, options); break;
                }
                rcs[r] = rc;
            }));
            }
            catch (...) // no thread to be had: finish what was started, then report (a C interface must not throw)
            {
                spawnFailed = true;
            }
        }
        for (size_t i = 0; i < workers.size(); i++)
            workers[i].join();
        if (spawnFailed)
        {
            m->lastError = "could not start a worker thread";
            return CVTTMI_E_HIP;
        }
        for (int r = 0; r < world; r++)
            if (rcs[r] != CVTTMI_OK)
            {
                char head[64];
                snprintf(head, sizeof(head), "shard %d (device %d): ", r, m->devices[r]);
                m->lastError = std::string(head) + cvttmi_last_error(m->ctx[r]);
                return rcs[r];
            }
        return CVTTMI_OK;
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
                g_handles[key] = m;
            }
            else
                m = it->second;
        }
        return cvttmi_multi_encode(m, format, out, blocks, numBlocks, blocksPerRow, options, plan);
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
