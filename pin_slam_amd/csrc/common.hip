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

// Sticky status flags on the device (PIN_STATUS_*): one word per device, allocated on first use and never freed.  Kernels that
// detect a condition the host must hear about OR a bit in; the registration loop's solve kernel copies the word into its
// state read-back (PIN_GN_STATE_STATUS), pin_status reads it synchronously.
int* status_word() {
    static int* words[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (words[dev] == nullptr) {
        int* p = nullptr;
        if (hipMalloc(&p, sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, sizeof(int)) != hipSuccess) return nullptr;
        words[dev] = p;
    }
    return words[dev];
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

extern "C" int pin_status(int32_t* flags_out, int32_t clear, void* stream) {
    PIN_ENTER();
    PIN_CHECK_ARG(flags_out, "NULL pointer");
    int* w = pin::status_word();
    if (w == nullptr) return pin::fail(-2, "pin_status: no status word on this device");
    hipStream_t s = pin::as_stream(stream);
    PIN_CHECK_HIP(hipMemcpyAsync(flags_out, w, sizeof(int), hipMemcpyDeviceToHost, s));
    if (clear) PIN_CHECK_HIP(hipMemsetAsync(w, 0, sizeof(int), s));
    PIN_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
}

extern "C" int pin_version(void) { return PIN_ABI_VERSION; }
extern "C" const char* pin_last_error(void) { return pin::last_error_buf(); }
