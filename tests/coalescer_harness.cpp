// Host-only harness for csrc/coalescer.h (tests/test_coalescer.py): the queueing logic of the drop-in face with a stand-in
// backend -- "encode" XORs every byte of a group with a key byte after a short sleep that plays the launch -- so that the
// > maxGroups path, mixed kinds and slot re-keying run under ASan / TSan on a machine without a GPU.
//   harness <threads> <callsPerThread> <maxGroups> <kinds> [launchUs]
// prints "ok launches=L largest=B" or the first mismatch; exit code 0 / 1.
#include "../convectionkernels_amd/csrc/coalescer.h"

#include <stdio.h>
#include <stdlib.h>

namespace
{
    struct Key
    {
        int kind;
        bool same(const Key &o) const { return kind == o.kind; }
    };
    const size_t kIn = 512, kOut = 128;
    int g_launchUs = 30;
    std::atomic<int> g_contexts(0), g_buffers(0);

    void *create() { g_contexts++; return new int(0); }
    void destroy(void *c) { g_contexts--; delete static_cast<int *>(c); }
    void *hostAlloc(void *, size_t n) { g_buffers++; return malloc(n); }
    void hostFree(void *, void *p) { g_buffers--; free(p); }
    int encode(void *ctx, const Key &k, uint8_t *out, const uint8_t *in, size_t groups)
    {
        int *inFlight = static_cast<int *>(ctx);
        if (++*inFlight != 1) // one launch at a time per slot (TSan sees a race here as well if the coalescer allows two)
            return -7;
        std::this_thread::sleep_for(std::chrono::microseconds(g_launchUs));
        for (size_t g = 0; g < groups; g++)
            for (size_t i = 0; i < kOut; i++)
                out[g * kOut + i] = static_cast<uint8_t>(in[g * kIn + i * 4] ^ (0x5a + k.kind));
        --*inFlight;
        return 0;
    }
}

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 300;
    const int calls = argc > 2 ? atoi(argv[2]) : 20;
    const size_t maxGroups = argc > 3 ? static_cast<size_t>(atoi(argv[3])) : 256;
    const int kinds = argc > 4 ? atoi(argv[4]) : 1;
    g_launchUs = argc > 5 ? atoi(argv[5]) : 30;
    std::atomic<int> bad(0), noSlot(0);
    uint64_t launches = 0, largest = 0;
    {
        const cvttmi_dropin::Backend<Key> be = {create, destroy, hostAlloc, hostFree, encode};
        cvttmi_dropin::Coalescer<Key> co(be, maxGroups, 100, kIn, kOut);
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++)
            pool.push_back(std::thread([&, t]() {
                std::vector<uint8_t> in(kIn), out(kOut);
                for (int c = 0; c < calls; c++)
                {
                    Key k;
                    k.kind = kinds <= 1 ? 0 : (t + c / 3) % kinds; // a thread changes its kind every three calls: slots are re-keyed
                    for (size_t i = 0; i < kIn; i++)
                        in[i] = static_cast<uint8_t>(t * 131 + c * 17 + i * 7);
                    std::fill(out.begin(), out.end(), 0xEE);
                    int rc = co.call(k, out.data(), in.data(), kIn, kOut);
                    if (rc == cvttmi_dropin::Coalescer<Key>::kNoSlot)
                    {
                        noSlot++; // the caller's own context: nothing to check here
                        continue;
                    }
                    bool ok = rc == 0;
                    for (size_t i = 0; ok && i < kOut; i++)
                        ok = out[i] == static_cast<uint8_t>(in[i * 4] ^ (0x5a + k.kind));
                    if (!ok && bad++ == 0)
                        fprintf(stderr, "mismatch: thread %d call %d rc %d\n", t, c, rc);
                }
            }));
        for (size_t i = 0; i < pool.size(); i++)
            pool[i].join();
        launches = co.launches();
        largest = co.largestBatch();
    }
    if (g_contexts != 0 || g_buffers != 0)
    {
        fprintf(stderr, "leak: %d contexts, %d buffers\n", g_contexts.load(), g_buffers.load());
        return 1;
    }
    if (largest > maxGroups)
    {
        fprintf(stderr, "a launch carried %llu groups, more than maxGroups\n", (unsigned long long)largest);
        return 1;
    }
    printf("%s launches=%llu largest=%llu noslot=%d\n", bad ? "BAD" : "ok", (unsigned long long)launches, (unsigned long long)largest, noSlot.load());
    return bad ? 1 : 0;
}
