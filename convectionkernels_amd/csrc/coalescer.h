// Coalescing of concurrent one-group calls of the drop-in C++ face (cxx_api.cpp).
//
// The reference's own entry points take ONE group of 8 blocks (ConvectionKernels_API.cpp:41-99) and are called by one worker
// thread per group (etc2packer.cpp:215-281).  On a GPU such a call is a PCIe round trip and a one-wave launch, so calls of the
// same kind (format, Options, plan) that arrive while a launch of that kind is in flight wait for it, and the first of them
// then encodes all of them with ONE launch (groups are independent: the bytes are the ones separate calls give).
//
// One SLOT per kind: a slot owns a context (stream, staging) and its own queue, so pools of callers with different kinds --
// per-texture weights, several formats -- run side by side instead of queueing behind each other's launches; when more kinds
// are live than there are slots the call runs on the caller thread's own context (`call` returns kNoSlot).  A single caller
// thread never waits: nothing is in flight when its call arrives.
//
// The device side is behind `Backend` (plain function pointers) so that the queueing logic is testable on a machine without
// a GPU, under ASan / TSan, with a stand-in backend (tests/test_coalescer.py).
#ifndef CVTTMI_COALESCER_H
#define CVTTMI_COALESCER_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace cvttmi_dropin
{
    // what makes two calls "the same kind": K = a trivially copyable key compared bytewise by `same`
    template<class K>
    struct Backend
    {
        void *(*create)();                                // a context; NULL = failure
        void (*destroy)(void *ctx);
        void *(*hostAlloc)(void *ctx, size_t bytes);      // page-locked staging; NULL = failure
        void (*hostFree)(void *ctx, void *p);
        int (*encode)(void *ctx, const K &key, uint8_t *out, const uint8_t *in, size_t numGroups); // 0 = ok
    };

    template<class K>
    class Coalescer
    {
    public:
        static const int kNoSlot = -1000;    // every slot is busy with another kind: the caller encodes on its own context
        static const int kStagingFailed = -1001;
        static const size_t kSlots = 8;
        // maxGroups: groups per launch (256 x 8 blocks = 128 KiB of PixelBlockU8); maxInBytes / maxOutBytes: the largest group
        Coalescer(const Backend<K> &backend, size_t maxGroups, int windowUs, size_t maxInBytes, size_t maxOutBytes)
            : be_(backend), maxGroups_(maxGroups < 1 ? 1 : maxGroups), windowUs_(windowUs), maxIn_(maxInBytes), maxOut_(maxOutBytes), clock_(0)
        {
        }
        ~Coalescer()
        {
            for (size_t i = 0; i < kSlots; i++)
            {
                Slot &s = slots_[i];
                if (s.ctx)
                {
                    if (s.stageIn) be_.hostFree(s.ctx, s.stageIn);
                    if (s.stageOut) be_.hostFree(s.ctx, s.stageOut);
                    be_.destroy(s.ctx);
                }
            }
        }

        // one group: inBytes in, outBytes out.  Returns the backend's code, kNoSlot or kStagingFailed.
        // lastCtx (optional): the context the launch ran on, for error texts.
        int call(const K &key, uint8_t *out, const uint8_t *in, size_t inBytes, size_t outBytes, void **lastCtx = NULL)
        {
            Slot *s = acquire(key);
            if (!s)
                return kNoSlot;
            const int rc = callOnSlot(*s, key, out, in, inBytes, outBytes);
            if (lastCtx)
                *lastCtx = s->ctx;
            release(s);
            return rc;
        }

        // launches so far and the largest number of groups one of them carried (tests, tuning)
        uint64_t launches() const { return launches_.load(); }
        uint64_t largestBatch() const { return largest_.load(); }

    private:
        struct Request
        {
            const uint8_t *in;
            uint8_t *out;
            int rc;
            std::atomic<bool> done;
        };
        struct Slot
        {
            Slot() : keyed(false), users(0), lastUse(0), ctx(NULL), stageIn(NULL), stageOut(NULL), busy(false), recent(1), busyFlag(false) {}
            // under slotsMu_
            bool keyed;
            K key;
            size_t users; // callers inside callOnSlot: a slot changes its kind only at 0
            uint64_t lastUse;
            // under mu
            void *ctx;
            void *stageIn, *stageOut;
            std::mutex mu;
            std::condition_variable cv;
            std::vector<Request *> pending;
            bool busy;
            size_t recent;              // calls per launch, recently
            std::atomic<bool> busyFlag; // == busy, readable without the lock
        };

        Slot *acquire(const K &key)
        {
            std::lock_guard<std::mutex> lock(slotsMu_);
            Slot *idle = NULL;
            for (size_t i = 0; i < kSlots; i++)
            {
                Slot &s = slots_[i];
                if (s.keyed && s.key.same(key))
                {
                    s.users++;
                    s.lastUse = ++clock_;
                    return &s;
                }
                // a slot nobody is in may change its kind: never-used ones first, then the least recently used
                if (s.users == 0 && (!idle || (!s.keyed && idle->keyed) || (s.keyed == idle->keyed && s.lastUse < idle->lastUse)))
                    idle = &s;
            }
            if (!idle)
                return NULL;
            idle->keyed = true;
            idle->key = key;
            idle->users = 1;
            idle->lastUse = ++clock_;
            return idle;
        }
        void release(Slot *s)
        {
            std::lock_guard<std::mutex> lock(slotsMu_);
            s->users--;
        }

        int callOnSlot(Slot &s, const K &key, uint8_t *out, const uint8_t *in, size_t inBytes, size_t outBytes)
        {
            Request me;
            me.in = in;
            me.out = out;
            me.rc = 0;
            me.done.store(false, std::memory_order_relaxed);
            std::unique_lock<std::mutex> lock(s.mu);
            s.pending.push_back(&me);
            for (;;)
            {
                if (me.done.load(std::memory_order_acquire))
                    return me.rc;
                if (!s.busy)
                    break; // nothing in flight: this thread runs the next launch
                // A launch is in flight (tens of microseconds): poll for a while without the lock -- a futex wake-up costs
                // about as much as the launch itself -- and only then sleep on the condition variable.
                lock.unlock();
                const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                bool turn = false;
                while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(400))
                {
                    if (me.done.load(std::memory_order_acquire) || !s.busyFlag.load(std::memory_order_acquire))
                    {
                        turn = true;
                        break;
                    }
                    std::this_thread::yield();
                }
                lock.lock();
                if (!turn && s.busy && !me.done.load(std::memory_order_acquire))
                    s.cv.wait_until(lock, std::chrono::system_clock::now() + std::chrono::milliseconds(2)); // (system clock: pthread_cond_timedwait, which gcc 11's TSan intercepts)
            }
            // Leader.  `me` is not done and no launch is in flight, so it is still queued (only a leader takes requests out,
            // and it marks them done before it clears `busy`).
            s.busy = true;
            s.busyFlag.store(true, std::memory_order_release);
            // The callers of the previous launch return, prepare their next group and arrive here within a few microseconds of
            // each other; the first one to arrive would otherwise leave with a launch of its own and make the others wait for
            // it.  So when recent launches carried more calls than are waiting now, give the others a moment (bounded: 100 us,
            // about one one-wave launch; measured with 16 callers, BC7 / ETC2 RGBA calls per second in total: no wait 96 k / 30 k,
            // 40 us 129 k / 46 k, 80 us 137 k / 51 k, 150 us 147 k / 56 k) -- a lone caller thread (recent == 1) never waits, and a
            // pool that shrinks pays the wait once per lost thread (recent falls by one per launch).
            const size_t want = s.recent < maxGroups_ ? s.recent : maxGroups_;
            if (s.pending.size() < want)
            {
                const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                while (s.pending.size() < want && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(windowUs_))
                {
                    lock.unlock();
                    std::this_thread::yield();
                    lock.lock();
                }
            }
            // This launch: the leader's own request FIRST (with more callers queued than a launch takes it would otherwise
            // launch without itself and return with nothing written), then the others in arrival order.
            std::vector<Request *> batch;
            batch.reserve(s.pending.size() < maxGroups_ ? s.pending.size() : maxGroups_);
            batch.push_back(&me);
            {
                size_t keep = 0;
                for (size_t i = 0; i < s.pending.size(); i++)
                {
                    Request *r = s.pending[i];
                    if (r == &me)
                        continue; // already first in the batch
                    if (batch.size() < maxGroups_)
                        batch.push_back(r);
                    else
                        s.pending[keep++] = r;
                }
                s.pending.resize(keep);
            }
            // what "recently" means: this launch, or one less than before when it carried fewer calls (a lone caller is back at 1 after a few calls)
            s.recent = batch.size() >= s.recent ? batch.size() : s.recent - 1;
            lock.unlock();
            int rc = ensure(s);
            if (rc == 0)
            {
                if (batch.size() == 1)
                    rc = be_.encode(s.ctx, key, out, in, 1);
                else
                {
                    for (size_t i = 0; i < batch.size(); i++)
                        memcpy(static_cast<uint8_t *>(s.stageIn) + i * inBytes, batch[i]->in, inBytes);
                    rc = be_.encode(s.ctx, key, static_cast<uint8_t *>(s.stageOut), static_cast<const uint8_t *>(s.stageIn), batch.size());
                    if (rc == 0)
                        for (size_t i = 0; i < batch.size(); i++)
                            memcpy(batch[i]->out, static_cast<uint8_t *>(s.stageOut) + i * outBytes, outBytes);
                }
            }
            launches_.fetch_add(1);
            for (uint64_t seen = largest_.load(); batch.size() > seen && !largest_.compare_exchange_weak(seen, batch.size());)
            {
            }
            lock.lock();
            for (size_t i = 1; i < batch.size(); i++)
            {
                Request *r = batch[i];
                r->rc = rc;
                r->done.store(true, std::memory_order_release); // (the request's owner may return, and the object die, right after this)
            }
            s.busy = false;
            s.busyFlag.store(false, std::memory_order_release);
            s.cv.notify_all();
            return rc;
        }

        // the slot's context and staging, created by its first leader (only a leader gets here, one at a time per slot)
        int ensure(Slot &s)
        {
            if (!s.ctx)
            {
                s.ctx = be_.create();
                if (!s.ctx)
                    return kStagingFailed;
            }
            if (!s.stageIn && !(s.stageIn = be_.hostAlloc(s.ctx, maxGroups_ * maxIn_)))
                return kStagingFailed;
            if (!s.stageOut && !(s.stageOut = be_.hostAlloc(s.ctx, maxGroups_ * maxOut_)))
                return kStagingFailed;
            return 0;
        }

        Backend<K> be_;
        size_t maxGroups_;
        int windowUs_;
        size_t maxIn_, maxOut_;
        std::mutex slotsMu_;
        uint64_t clock_;
        Slot slots_[kSlots];
        std::atomic<uint64_t> launches_{0}, largest_{0};
    };
}
#endif
