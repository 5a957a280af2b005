// Shared host-side helpers for librtsync.so (error reporting, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/rtsync.h"

namespace rts {

char *last_error_buf();  // thread-local, 512 bytes
int set_error(int code, const char *fmt, ...);

#define RTS_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ::rts::set_error(RTS_ERR_HIP, "%s failed: %s (%s:%d)", #call,              \
                                    hipGetErrorString(e_), __FILE__, __LINE__);               \
    } while (0)

}  // namespace rts
