// Error reporting and ABI version of libhgr.so.
#include "hgr_common.h"

static thread_local char g_err[512] = "";

int hgr_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int hgr_abi_version(void) { return HGR_ABI_VERSION; }
extern "C" const char *hgr_last_error(void) { return g_err; }
