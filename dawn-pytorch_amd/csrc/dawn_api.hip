// Error state + ABI version for libdawn_hip.so.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" int dawn_set_error(hipError_t e, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "%s:%d: HIP error %d (%s)", file, line, (int)e, hipGetErrorString(e));
    return -(int)e;
}
extern "C" int dawn_set_error_msg(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
extern "C" const char* dawn_last_error(void) { return g_err; }
extern "C" int dawn_abi_version(void) { return 8; }
