// Library-level entry points: version and error text.
#include <stdarg.h>

#include "pin_common.h"

namespace pin {
static thread_local char g_err[512] = "";
char* last_error_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace pin

extern "C" int pin_version(void) { return PIN_ABI_VERSION; }
extern "C" const char* pin_last_error(void) { return pin::last_error_buf(); }
