// Thread-local error text + version for the nfdp C ABI.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/nfdp.h"

static thread_local char g_err[512] = "";

void nfdp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* nfdp_last_error(void) { return g_err; }
extern "C" int nfdp_version(void) { return 100; }
