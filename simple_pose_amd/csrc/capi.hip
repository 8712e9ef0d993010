// capi.hip - library-level entry points and error plumbing of libsimple_pose_hip.so.
#include "sp_common.h"

static thread_local char g_err[512] = "";

void sp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int sp_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        sp_set_error("%s: %s", what, hipGetErrorString(e));
        return SP_ELAUNCH;
    }
    return SP_OK;
}

extern "C" int sp_abi_version(void) { return SP_ABI_VERSION; }
extern "C" const char* sp_last_error(void) { return g_err; }

static thread_local char g_kname[256] = "";
static thread_local bool g_kname_active = false;
bool sp_name_query_active() { return g_kname_active; }
void sp_name_query_begin() { g_kname_active = true; g_kname[0] = 0; }
const char* sp_name_query_end() { g_kname_active = false; return g_kname; }
void sp_name_query_set(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kname, sizeof(g_kname), fmt, ap);
    va_end(ap);
}
