// api_common.hip -- status/error plumbing and the device gate of the C-ABI (include/eao_fusion.h)
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.h"

namespace eao {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

eao_status require_device() {
    static std::once_flag once;
    static eao_status cached = EAO_ERR_NO_DEVICE;
    static char cached_msg[256] = "";
    std::call_once(once, [] {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) {
            snprintf(cached_msg, sizeof(cached_msg),
                     "no HIP device available (%s); libeaofusion_hip has no CPU fallback",
                     e != hipSuccess ? hipGetErrorString(e) : "device count 0");
            (void)hipGetLastError();
            return;
        }
        int dev = 0;
        (void)hipGetDevice(&dev);
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            snprintf(cached_msg, sizeof(cached_msg), "hipGetDeviceProperties failed");
            return;
        }
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            snprintf(cached_msg, sizeof(cached_msg), "device arch %s is not gfx950 (code objects are built for MI355X only)",
                     prop.gcnArchName);
            return;
        }
        cached = EAO_OK;
    });
    if (cached != EAO_OK) set_error("%s", cached_msg);
    return cached;
}

hipError_t create_stream(hipStream_t* s, StreamClass c) {
    static std::once_flag once;
    static int prio[3] = {0, 0, 0};
    static bool on = true;
    std::call_once(once, [] {
        const char* e = getenv("EAO_STREAM_PRIORITY");
        if (e && !atoi(e)) { on = false; return; }
        int least = 0, greatest = 0;      // numerically: greatest priority <= least priority
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); on = false; return; }
        if (least == greatest) { on = false; return; }
        prio[(int)StreamClass::Latency] = greatest;
        prio[(int)StreamClass::Bulk] = least;
        prio[(int)StreamClass::Background] = least - greatest >= 2 ? (least + greatest) / 2 : least;
    });
    // (Measured and not kept, profiles/r06_mixed_load_experiments.txt: leaving one or two compute units per XCD to the latency class with a CU mask on the
    //  Background / Bulk streams -- hipExtStreamCreateWithCUMask -- made every class slower: LocalBundleAdjustment 2.14 -> 2.67 ms, the 25-window batch 2.8 -> 5.0 ms,
    //  the tracked frame's p99 beside a looping LBA 0.87 -> 2.1 ms.)
    if (!on) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio[(int)c]);
}

namespace {
std::atomic<long long> g_latencyStampNs{0};
long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace
void note_latency_call() { g_latencyStampNs.store(now_ns(), std::memory_order_relaxed); }
bool latency_caller_alive() {
    const long long t = g_latencyStampNs.load(std::memory_order_relaxed);
    return t != 0 && now_ns() - t < 100000000ll;
}

hipError_t wait_latency(hipStream_t s) {
    note_latency_call();
    static const bool spin = !(getenv("EAO_SPIN_WAIT") && !atoi(getenv("EAO_SPIN_WAIT")));
    if (spin) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned it = 0;; it++) {
            const hipError_t e = hipStreamQuery(s);
            if (e != hipErrorNotReady) return e;
            if ((it & 15) == 15 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(3)) break;
        }
    }
    return hipStreamSynchronize(s);
}

// roctx ranges around the stages of the hot path (SURVEY.md s5 "Tracing"): EAO_ROCTX=1 loads libroctx64 at the first range and
// every later range shows up under `rocprofv3 --marker-trace`; without the variable a range is one predictable branch.
namespace {
typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)();
roctx_push_t g_push = nullptr;
roctx_pop_t g_pop = nullptr;
bool roctx_ready() {
    static std::once_flag once;
    static bool on = false;
    std::call_once(once, [] {
        const char* e = getenv("EAO_ROCTX");
        if (!e || !atoi(e)) return;
        // rocprofv3 listens to the rocprofiler-sdk flavour of the library; roctracer's libroctx64 is the fallback (rocprof v1/v2)
        void* lib = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        g_push = (roctx_push_t)dlsym(lib, "roctxRangePushA");
        g_pop = (roctx_pop_t)dlsym(lib, "roctxRangePop");
        on = g_push && g_pop;
    });
    return on;
}
}  // namespace
void range_push(const char* name) { if (roctx_ready()) g_push(name); }
void range_pop() { if (roctx_ready()) g_pop(); }

}  // namespace eao

extern "C" {

const char* eao_last_error(void) { return eao::g_err; }
eao_status eao_device_check(void) { return eao::require_device(); }
const char* eao_version(void) { return "eaofusion-hip 0.1 (gfx950)"; }

}  // extern "C"
