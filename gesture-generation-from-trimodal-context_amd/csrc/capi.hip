// ABI bookkeeping: version, thread-local error message.
#include <stdarg.h>

#include "common.hpp"

namespace tg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace tg

extern "C" int tg_version(void) { return TG_ABI_VERSION; }
extern "C" const char* tg_last_error(void) { return tg::g_err; }
