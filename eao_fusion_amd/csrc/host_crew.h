// host_crew.h -- the process-wide crew of host threads behind eao_local_ba_batch (window set-up workers, group leaders) and behind the set-up of a map-scale
// BundleAdjustment (sessions).  Plain C++17, no HIP: tests/cpp/host_crew_test.cpp drives it under ThreadSanitizer on a machine without a GPU.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>

namespace eao {
namespace lm {

// The host threads of a batch call (window set-up workers, group leaders) are PERSISTENT: a call hands `count` tasks to the crew and
// joins them.  Starting eight std::threads per call cost ~0.3 ms of a 2.9 ms batch (clone + first-touch of the thread's HIP state,
// one after the other) -- which is why more set-up threads used to make a call slower.  The crew is process-wide, grows on
// demand and is never torn down (its threads sleep on a condition variable between calls and die with the process).
inline thread_local bool t_inCrew = false;      // this thread belongs to the crew: a task must not hand work to the crew itself (one call at a time owns it)
struct HostCrew {
    std::mutex m;
    std::condition_variable wake, finished;
    std::function<void(int)> fn;
    int generation = 0, next = 0, count = 0, running = 0, threads = 0;
    // ---- SESSIONS (round 6): the set-up of a map-scale BundleAdjustment is a dozen short parallel passes (0.05 - 0.3 ms each) with serial joints in between.  Handing each
    //      pass over through the condition variable cost ~0.1 ms per pass -- as much as the pass saved on the 200-keyframe map.  A session wakes the crew ONCE; its threads
    //      then poll one word for the passes of that set-up (claiming chunks with a compare-and-swap on it: sequence number, chunk count and next chunk in one 64-bit
    //      word, so a thread that is late can never take a chunk of a later pass) and go back to sleep when the session closes.  The caller works on every pass itself
    //      and waits only for chunks somebody claimed: a crew thread that never wakes costs nothing but its share.
    int sessionSeq = 0;                          // (under m)
    std::atomic<int> sessionOpen{0};
    std::atomic<uint64_t> passWord{0};           // seq << 40 | chunks << 20 | next chunk
    std::atomic<int> passDone{0};
    const std::function<void(int)>* passFn = nullptr;
    uint64_t passSeq = 0;
    static void cpu_relax() { __builtin_ia32_pause(); }
    bool claim_chunks(uint64_t seq) {            // chunks of pass `seq` until none is left (or the pass is over); true if the word still belongs to that pass
        for (;;) {
            uint64_t w = passWord.load(std::memory_order_acquire);
            if ((w >> 40) != seq) return false;
            const int nx = (int)(w & 0xFFFFF), n = (int)((w >> 20) & 0xFFFFF);
            if (nx >= n) return true;
            if (!passWord.compare_exchange_weak(w, w + 1, std::memory_order_acq_rel)) continue;
            (*passFn)(nx);
            passDone.fetch_add(1, std::memory_order_release);
        }
    }
    void spin_session() {
        const auto t0 = std::chrono::steady_clock::now();
        dbgJoined++;
        for (int it = 0; sessionOpen.load(std::memory_order_acquire); it++) {
            const uint64_t w = passWord.load(std::memory_order_acquire);
            if ((int)(w & 0xFFFFF) < (int)((w >> 20) & 0xFFFFF)) claim_chunks(w >> 40);
            else cpu_relax();
            if ((it & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;      // (a set-up takes a few milliseconds)
        }
    }
    void body() {
        t_inCrew = true;
        int seen = 0, seenSession = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            wake.wait(lk, [&] { return (generation != seen && next < count) || sessionSeq != seenSession; });
            if (sessionSeq != seenSession) {
                seenSession = sessionSeq;
                lk.unlock();
                spin_session();
                continue;
            }
            while (next < count) {
                const int i = next++;
                running++;
                lk.unlock();
                fn(i);
                lk.lock();
                running--;
            }
            seen = generation;
            if (running == 0) finished.notify_all();
        }
    }
    // runs fn(0 .. n-1) on the crew (at least n threads, so tasks that wait for each other cannot starve) and fn0() on the caller
    std::mutex callMu;      // one batch call at a time uses the crew (calls from several host threads queue up here)
    void run(int n, const std::function<void(int)>& f, const std::function<void()>& fn0) {
        std::lock_guard<std::mutex> oneCall(callMu);
        {
            std::unique_lock<std::mutex> lk(m);
            while (threads < n) { std::thread(&HostCrew::body, this).detach(); threads++; }
            fn = f; next = 0; count = n; generation++;
        }
        wake.notify_all();
        fn0();
        std::unique_lock<std::mutex> lk(m);
        finished.wait(lk, [&] { return next >= count && running == 0; });
        count = 0;
    }
    // a session: false when another call owns the crew (the caller then runs its passes alone)
    bool session_begin(int nThreads) {
        if (!callMu.try_lock()) return false;
        {
            std::unique_lock<std::mutex> lk(m);
            while (threads < nThreads) { std::thread(&HostCrew::body, this).detach(); threads++; }
            sessionOpen.store(1, std::memory_order_release);
            sessionSeq++;
            dbgJoined = 0; dbgT0 = std::chrono::steady_clock::now();
        }
        wake.notify_all();
        return true;
    }
    void session_end() {
        sessionOpen.store(0, std::memory_order_release);
        callMu.unlock();
    }
    std::atomic<int> dbgJoined{0};
    std::chrono::steady_clock::time_point dbgT0;
    void session_pass(int nChunks, const std::function<void(int)>& chunk) {      // (the session's owner only)
        static const bool dbg = getenv("EAO_DEBUG_CREW") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        std::atomic<int> mine{0};
        std::function<void(int)> wrapped;
        if (dbg) {
            wrapped = [&](int q) { if (!t_inCrew) mine++; chunk(q); };
            session_pass_(nChunks, wrapped);
            fprintf(stderr, "[crew] pass of %d chunks: %.3f ms (at %.3f since session start), caller ran %d, %d crew threads in the session so far\n", nChunks,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), std::chrono::duration<double, std::milli>(t0 - dbgT0).count(), mine.load(), dbgJoined.load());
            return;
        }
        session_pass_(nChunks, chunk);
    }
    void session_pass_(int nChunks, const std::function<void(int)>& chunk) {
        passFn = &chunk;
        passDone.store(0, std::memory_order_relaxed);
        passSeq = (passSeq + 1) & 0xFFFFFF;
        if (passSeq == 0) passSeq = 1;
        passWord.store((passSeq << 40) | ((uint64_t)nChunks << 20), std::memory_order_release);
        claim_chunks(passSeq);
        while (passDone.load(std::memory_order_acquire) < nChunks) cpu_relax();
    }
};
inline HostCrew& host_crew() {
    // ONE crew per process (round 4; it was one per calling thread: a pool of short-lived caller threads grew the process by ~19 sleeping threads per
    // caller).  Its size is the largest thread count a call ever asked for (the set-up threads + group leaders of eao_local_ba_batch: about
    // nineteen with the defaults); the threads sleep on a condition variable between calls and end with the process (detached: a static destructor
    // that joined them would run after the HIP runtime's own teardown).
    static HostCrew* crew = new HostCrew();
    return *crew;
}

}  // namespace lm
}  // namespace eao
