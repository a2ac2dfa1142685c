// Cost of the UNMODIFIED caller's convention on the drop-in (developer / bench tool): one cvtt::Kernels::Encode* call
// per 8 blocks (reference ConvectionKernels_API.cpp:41-54; caller loop etc2packer/etc2packer.cpp:215-281), through
// include/cvtt/ConvectionKernels.h and the shipped library -- every call is a PCIe round trip plus a one-wave launch.
// Prints one JSON object: microseconds per call and calls per second from 1 and 16 caller threads (a context each).
//   dropin_bench [seconds per measurement, default 0.5]
#include "cvtt/ConvectionKernels.h"

#include <atomic>
#include <chrono>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

namespace
{
    typedef std::chrono::steady_clock Clock;
    const int kGroups = 256; // distinct inputs a thread cycles through

    template <class Call>
    double callsPerSecond(int threads, double seconds, Call call)
    {
        std::atomic<long> total(0);
        std::atomic<int> ready(0);
        std::atomic<bool> go(false);
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++)
            th.emplace_back([&, t] {
                call(t, 0); // creates the thread's context, loads the code object
                call(t, 1);
                ready++;
                while (!go.load())
                    std::this_thread::yield();
                const Clock::time_point t0 = Clock::now();
                long n = 0;
                while (std::chrono::duration<double>(Clock::now() - t0).count() < seconds)
                {
                    for (int k = 0; k < 8; k++)
                        call(t, (int)((n + k) % kGroups));
                    n += 8;
                }
                total += n;
            });
        while (ready.load() < threads)
            std::this_thread::yield();
        const Clock::time_point t0 = Clock::now();
        go = true;
        for (auto &x : th)
            x.join();
        return (double)total.load() / std::chrono::duration<double>(Clock::now() - t0).count();
    }
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 0.5;
    const int maxThreads = 16;
    static cvtt::PixelBlockU8 in[kGroups][cvtt::NumParallelBlocks];
    uint64_t s = 0x1234567ull;
    for (size_t i = 0; i < sizeof(in); i++)
    {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        reinterpret_cast<uint8_t *>(in)[i] = (uint8_t)(z >> 56);
    }
    static uint8_t out[maxThreads][128];
    cvtt::Options options;
    cvtt::BC7EncodingPlan plan;
    auto bc7 = [&](int t, int g) { cvtt::Kernels::EncodeBC7(out[t], in[g], options, plan); };
    auto bc1 = [&](int t, int g) { cvtt::Kernels::EncodeBC1(out[t], in[g], options); };
    auto etc = [&](int t, int g) { cvtt::Kernels::EncodeETC2RGBA(out[t], in[g], options, NULL); };
    printf("{");
    const char *names[3] = {"bc7", "bc1", "etc2rgba"};
    for (int f = 0; f < 3; f++)
    {
        double c1, c16;
        if (f == 0) { c1 = callsPerSecond(1, seconds, bc7); c16 = callsPerSecond(maxThreads, seconds, bc7); }
        else if (f == 1) { c1 = callsPerSecond(1, seconds, bc1); c16 = callsPerSecond(maxThreads, seconds, bc1); }
        else { c1 = callsPerSecond(1, seconds, etc); c16 = callsPerSecond(maxThreads, seconds, etc); }
        printf("%s\"%s\": {\"us_per_call_1_thread\": %.2f, \"calls_per_s_1_thread\": %.0f, \"mblocks_s_1_thread\": %.5f, "
               "\"us_per_call_16_threads\": %.2f, \"calls_per_s_16_threads\": %.0f, \"mblocks_s_16_threads\": %.5f}",
               f ? ", " : "", names[f], 1e6 / c1, c1, c1 * 8 / 1e6, 16e6 / c16, c16, c16 * 8 / 1e6);
    }
    // a pool of mixed kinds: sixteen threads, four different Options (per-texture weights) -- one coalescer slot per kind, so
    // the four kinds launch side by side (round 4: one busy flag serialised them)
    {
        static cvtt::Options mixed[4];
        for (int k = 0; k < 4; k++)
            mixed[k].redWeight = 0.25f + 0.25f * (float)k;
        auto bc7m = [&](int t, int g) { cvtt::Kernels::EncodeBC7(out[t], in[g], mixed[t & 3], plan); };
        const double c16 = callsPerSecond(maxThreads, seconds, bc7m);
        printf(", \"bc7_mixed_4_kinds\": {\"us_per_call_16_threads\": %.2f, \"calls_per_s_16_threads\": %.0f, \"mblocks_s_16_threads\": %.5f}", 16e6 / c16, c16, c16 * 8 / 1e6);
    }
    printf(", \"blocks_per_call\": 8, \"caller_threads\": [1, %d]}\n", maxThreads);
    return 0;
}
