// Error reporting + ABI version of libsenas_hip.so.
#include "common.h"
#include <string.h>

namespace senas {
static thread_local char g_err[512] = "";

void set_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}
void set_error_msg(const char* what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
}
}  // namespace senas

extern "C" const char* senas_last_error(void) { return senas::g_err; }
extern "C" int senas_abi_version(void) { return 22; }

