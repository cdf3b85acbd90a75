// Error reporting and version query of the C ABI.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mk_last_error(void) { return g_err; }
extern "C" int mk_version(void) { return 100; }
