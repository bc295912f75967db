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

static SatProbe* g_sat_probes = nullptr;      // filled by static initialisers of the translation units (before any call)
void register_sat_probe(SatProbe* p) { p->next = g_sat_probes; g_sat_probes = p; }

}  // namespace ia

extern "C" int ia_split_saturation_poll(unsigned int* h_flagged, int reset, void* stream) {
    IA_REQUIRE(h_flagged || reset, "nothing to do: no output pointer and no reset");
    hipStream_t s = (hipStream_t)stream;
    unsigned int words[64];
    int n = 0;
    for (ia::SatProbe* p = ia::g_sat_probes; p && n < 64; p = p->next, ++n) {
        words[n] = 0;
        const hipError_t e = p->read(h_flagged ? &words[n] : nullptr, reset, s);
        if (e != hipSuccess) return ia::fail(IA_ERR_LAUNCH, "ia_split_saturation_poll: %s", hipGetErrorString(e));
    }
    if (!h_flagged) return IA_OK;          // clear only: stream-ordered, the host does not wait
    const hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return ia::fail(IA_ERR_LAUNCH, "ia_split_saturation_poll: %s", hipGetErrorString(e));
    unsigned int any = 0;
    for (int i = 0; i < n; ++i) any |= words[i];
    *h_flagged = any;
    return IA_OK;
}

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
