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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global store
// (a full release fence), which costs a memory round trip per call in loops that stream results to HBM while the
// threads talk to each other through LDS alone.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

}  // namespace rts
