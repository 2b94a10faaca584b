// common.h -- error plumbing shared by the HIP translation units of libeaofusion_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/eao_fusion.h"
#include "ref_constants.inc"   // GENERATED from the reference text (tools/gen_ref_constants.py): namespace refc

namespace eao {

void set_error(const char* fmt, ...);

#define EAO_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            eao::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return EAO_ERR_NO_DEVICE;                                                                   \
        }                                                                                               \
    } while (0)

#define EAO_REQUIRE(cond, ...)           \
    do {                                 \
        if (!(cond)) {                   \
            eao::set_error(__VA_ARGS__); \
            return EAO_ERR_INVALID;      \
        }                                \
    } while (0)

// Fails loudly (EAO_ERR_NO_DEVICE) when no HIP device can run gfx950 code objects.  There is no CPU fallback.
eao_status require_device();

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    // grow-only allocation
    eao_status reserve(size_t count) {
        if (count <= n) return EAO_OK;
        release();
        EAO_HIP(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
        return EAO_OK;
    }
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Stream classes (round 6).  Upstream runs three threads at once on ONE device -- Tracking (per frame, real time), LocalMapping
// (LocalBundleAdjustment, src/LocalMapping.cc:75) and LoopClosing's global BundleAdjustment thread (src/LoopClosing.cc:594; all started
// in src/System.cc:98-138) -- so every stream the library creates says which of them it serves, and the HIP stream priority follows:
//   Latency    -- extractor handles, the tracked-frame chain, PoseOptimization, the guided searches (a frame waits for each of them),
//   Background -- LocalBundleAdjustment (one window and batches),
//   Bulk       -- map-scale BundleAdjustment.
// A priority class also has its own pool of hardware queues in the runtime, so a tracker stream no longer shares a queue with an LM
// stream (beyond four streams of one class they do share: measured in round 2 on the batch groups).  EAO_STREAM_PRIORITY=0: every
// stream at the default priority (the rounds 1-5 behaviour, for A/B runs).
enum class StreamClass { Latency = 0, Background = 1, Bulk = 2 };
hipError_t create_stream(hipStream_t* s, StreamClass c);
// The host's wait for a Latency-class stream: hipStreamSynchronize parks the thread on an interrupt, whose wake-up costs tens of microseconds and now and then
// milliseconds (tests/cpp/mixed_load.cpp: the 7-10 ms maxima of the tracked frame beside a looping bundle adjustment disappear under HSA_ENABLE_INTERRUPT=0).  A frame
// waits for each of these calls, so the caller's thread polls hipStreamQuery for up to 3 ms first (a Tracking thread has nothing else to do meanwhile; upstream's
// computes the features on that core) and only then blocks.  EAO_SPIN_WAIT=0: always block.
hipError_t wait_latency(hipStream_t s);
// A latency-class call is in progress / was made moments ago (stamped by wait_latency and by the tracked-frame chain's poll): eao_local_ba_batch deals its windows to TWO
// stream groups instead of four while a frame-rate caller is alive in the process.  Measured (tests/cpp/mixed_load.cpp, gpurun_out/r06n): four groups keep every CU's
// register file occupied without a gap, and the tracker's full-register-file workgroups wait for the batch's whole busy period (tracked frame p50 1.65 / p99 3.2 ms against
// 0.77 idle); with two groups 1.08 / 1.64 ms, the batch 2.87 -> 3.13 ms per call.  No latency-class call within the last 100 ms: four groups, as in the benchmark.
void note_latency_call();
bool latency_caller_alive();

#if defined(__HIPCC__)
// THE hand-over point between the lanes of ONE wavefront through LDS (or through memory the wave alone touches): the
// lanes' earlier stores are visible to the other lanes' later loads.  The hardware needs nothing for that -- a wave's
// LDS / memory instructions execute in order -- so a wavefront-scope fence lowers to NO instruction; what has to be
// stopped is the COMPILER: round 4 found the scheduler moving LDS reads above writes across a bare fence when the two
// sides used different access types (8-byte pairs in, 16-byte rows out: no alias in its view).  Hence: memory clobber +
// fence + wave_barrier (a scheduling barrier) + memory clobber.  Every such point in csrc/ goes through this one
// function (VERDICT r4 next #7a); the ISA of the kernels is unchanged by it (tools/isa_census.py, profiles/r05_wave_sync_isa.txt).
__device__ __forceinline__ void wave_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
#endif

// roctx range around a stage of the hot path (host side: the enqueue of its kernels); active only under EAO_ROCTX=1
void range_push(const char* name);
void range_pop();
struct Range {
    explicit Range(const char* name) { range_push(name); }
    ~Range() { range_pop(); }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};

}  // namespace eao
