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

namespace pin {
int pin_warm_knn(); int pin_warm_pool(); int pin_warm_prep(); int pin_warm_comm(); int pin_warm_train(); int pin_warm_maint();
int pin_warm_dp(); int pin_warm_sdf(); int pin_warm_brick();
}  // namespace pin

extern "C" int pin_warmup(void) {
    // HIP loads a code object when the first of its kernels is launched (tens of milliseconds for the larger translation units):
    // inside a SLAM loop that is a frame of 100 ms the first time a stage runs.  One attribute query per translation unit
    // loads them all up front.
    int (*const warm[])() = {pin::pin_warm_knn, pin::pin_warm_pool, pin::pin_warm_prep, pin::pin_warm_comm, pin::pin_warm_train,
                             pin::pin_warm_maint, pin::pin_warm_dp, pin::pin_warm_sdf, pin::pin_warm_brick};
    for (auto f : warm)
        if (f() != 0) return pin::fail(-2, "pin_warmup: a code object could not be loaded: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

extern "C" int pin_version(void) { return PIN_ABI_VERSION; }
extern "C" const char* pin_last_error(void) { return pin::last_error_buf(); }
