// Error reporting and version entry points of libanihip.
#include <stdarg.h>
#include <stdio.h>

#include "anihip_common.h"

namespace anihip {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace anihip

extern "C" const char *anihip_last_error(void) { return anihip::g_err; }
extern "C" int anihip_abi_version(void) { return ANIHIP_ABI_VERSION; }
