// librtsync.so: error reporting + device discovery (rts_last_error, rts_version, rts_device_count).
#include "common.h"

#include <string.h>

namespace rts {

char *last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace rts

extern "C" {

const char *rts_last_error(void) { return rts::last_error_buf(); }

int rts_version(void) { return 100; }

int rts_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        rts::set_error(RTS_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return e == hipErrorNoDevice ? 0 : RTS_ERR_HIP;
    }
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

}  // extern "C"
