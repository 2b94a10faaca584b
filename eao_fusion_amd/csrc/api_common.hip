// api_common.hip -- status/error plumbing and the device gate of the C-ABI (include/eao_fusion.h)
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

}  // namespace eao

extern "C" {

const char* eao_last_error(void) { return eao::g_err; }
eao_status eao_device_check(void) { return eao::require_device(); }
const char* eao_version(void) { return "eaofusion-hip 0.1 (gfx950)"; }

}  // extern "C"
