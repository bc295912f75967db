// Library-level entry points: version, last-error text, device count.
#include "ia_common.h"

namespace ia {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ia

extern "C" int ia_version(void) { return IA_HIP_ABI_VERSION; }

extern "C" size_t ia_last_error(char* h_buf, size_t n) {
    const char* msg = ia::error_buffer();
    size_t len = strlen(msg);
    if (h_buf && n) {
        size_t k = len < n - 1 ? len : n - 1;
        memcpy(h_buf, msg, k);
        h_buf[k] = 0;
    }
    return len;
}

extern "C" int ia_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return ia::fail(IA_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}
