// Error reporting + ABI version of libsenas_hip.so.
#include "common.h"
#include <string.h>
#include <mutex>
#include <set>
#include <utility>

namespace senas {
static thread_local char g_err[512] = "";

void set_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}
void set_error_msg(const char* what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
}

// hipFuncSetAttribute applies to the CURRENT device: remember (kernel, device ordinal) pairs, not one flag per process --
// a process that drives a second GPU (or a second thread) must raise the limit there too.
int raise_lds_limit(const void* kernel, int bytes, const char* what) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { set_error(what, e); return SENAS_ELAUNCH; }
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return SENAS_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { set_error(what, e); return SENAS_ELAUNCH; }
    done.insert({kernel, dev});
    return SENAS_OK;
}
}  // namespace senas

extern "C" const char* senas_last_error(void) { return senas::g_err; }
extern "C" int senas_abi_version(void) { return 35; }

