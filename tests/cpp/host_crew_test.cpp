// host_crew_test.cpp -- the process-wide host crew (eao_fusion_amd/csrc/host_crew.h) without a GPU: sessions (passes claimed chunk by chunk through one polled word),
// batch-style runs through the condition variable, and two callers competing for the crew.  Built by tests/test_host_crew_cpu.py with -fsanitize=thread: every
// chunk of every pass must run exactly once, a pass must not return before its chunks have, and ThreadSanitizer must stay silent.
#include <cstdio>
#include <random>
#include <vector>

#include "../../eao_fusion_amd/csrc/host_crew.h"

using eao::lm::host_crew;

static int fail(const char* what, int a, int b) { std::fprintf(stderr, "FAILED: %s (%d, %d)\n", what, a, b); return 1; }

int main() {
    std::mt19937 rng(12345);
    std::vector<int> hits(4096, 0), data(4096, 0);
    long long passes = 0, chunks = 0;
    // 1. sessions of a few passes each, crew sizes 1 .. 7 (the container has eight cores; more threads than cores on purpose at the top end)
    for (int s = 0; s < 300; s++) {
        const int nT = 1 + (int)(rng() % 7);
        if (!host_crew().session_begin(nT)) return fail("session_begin refused with nobody else around", s, nT);
        const int nPass = 1 + (int)(rng() % 8);
        for (int p = 0; p < nPass; p++) {
            const int n = 1 + (int)(rng() % (p % 3 == 0 ? 250 : 48));
            for (int q = 0; q < n; q++) hits[q] = 0;      // (plain writes: the pass's release store publishes them)
            const int salt = (int)(rng() & 0xFFFF);
            host_crew().session_pass(n, [&](int q) { hits[q]++; data[q] = q ^ salt; });
            for (int q = 0; q < n; q++) {
                if (hits[q] != 1) return fail("a chunk ran a number of times other than once", q, hits[q]);
                if (data[q] != (q ^ salt)) return fail("a chunk's write is not visible behind the pass", q, data[q]);
            }
            passes++; chunks += n;
        }
        host_crew().session_end();
        // 2. now and then a batch-style run between two sessions (the crew's other mode)
        if (s % 7 == 3) {
            std::atomic<int> sum(0);
            const int n = 2 + (int)(rng() % 6);
            host_crew().run(n, [&](int i) { sum += i + 1; }, [&] { sum += 100; });
            if (sum.load() != 100 + n * (n + 1) / 2) return fail("batch-style run lost a task", n, sum.load());
        }
    }
    // 3. two callers: whoever gets the session runs its passes on the crew, the other one learns it must work alone; both finish
    std::atomic<int> got(0), refused(0), bad(0);
    auto caller = [&](int id) {
        std::vector<int> mine(64, 0);
        for (int k = 0; k < 200; k++) {
            if (host_crew().session_begin(4)) {
                got++;
                host_crew().session_pass(64, [&](int q) { mine[q]++; });
                host_crew().session_end();
            } else {
                refused++;
                for (int q = 0; q < 64; q++) mine[q]++;
            }
        }
        for (int q = 0; q < 64; q++) if (mine[q] != 200) bad++;
        (void)id;
    };
    std::thread a(caller, 0), b(caller, 1);
    a.join(); b.join();
    if (bad.load()) return fail("a caller lost chunks while competing for the crew", bad.load(), 0);
    std::printf("host crew: %lld passes, %lld chunks, every chunk once; competing callers: %d sessions, %d refusals\n", passes, chunks, got.load(), refused.load());
    return 0;
}
