// Two-workgroup ping-pong over the strip DP's hand-over primitive (csrc/sdp.h): a naturally aligned 8-byte `sc1` store
// whose value is its own flag (the buffer is pre-filled with a sentinel), polled with `sc1` loads.  Reports the one-way
// latency = round trip / 2, for partners that are neighbours in dispatch order (blocks b, b^1: different XCDs under
// round-robin placement) and for partners 8 blocks apart (b, b^8: same XCD), idle chip.
//
//   hipcc --offload-arch=gfx950 -O3 -o handoff tools/microbench/handoff.hip && ./handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr unsigned long long kSentinel = 0x7FF4DEAD7FF4DEADull;

__device__ __forceinline__ unsigned long long load_sc1(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_sc1(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// blocks come in pairs (b, b ^ stride): the lower one pings, the other pongs.  words: [pair][iter][2]
__global__ void pingpong(unsigned long long *words, long long *cycles, int iters, int stride, int sleep) {
    const int b = blockIdx.x;
    const int partner = b ^ stride;
    const int pair = (b < partner ? b : partner);
    unsigned long long *w = words + (size_t)pair * iters * 2;
    if (threadIdx.x != 0) return;
    const bool pinger = b < partner;
    long long total = 0;
    for (int k = 0; k < iters; k++) {
        unsigned long long *mine = w + 2 * k + (pinger ? 0 : 1), *theirs = w + 2 * k + (pinger ? 1 : 0);
        long long spins = 0;
        if (pinger) {
            const long long t0 = (long long)__builtin_amdgcn_s_memtime();
            store_sc1(mine, (unsigned long long)k + 1);
            while (load_sc1(theirs) == kSentinel && ++spins < (1 << 22))
                if (sleep) __builtin_amdgcn_s_sleep(2);
            total += (long long)__builtin_amdgcn_s_memtime() - t0;
        } else {
            while (load_sc1(theirs) == kSentinel && ++spins < (1 << 22))
                if (sleep) __builtin_amdgcn_s_sleep(2);
            store_sc1(mine, (unsigned long long)k + 1);
        }
    }
    if (pinger) cycles[pair] = total;
}

int main() {
    const int iters = 2000, blocks = 16;
    unsigned long long *words;
    long long *cycles;
    hipMalloc(&words, sizeof(unsigned long long) * blocks * iters * 2);
    hipMalloc(&cycles, sizeof(long long) * blocks);
    unsigned long long *fill = (unsigned long long *)malloc(sizeof(unsigned long long) * blocks * iters * 2);
    for (int i = 0; i < blocks * iters * 2; i++) fill[i] = kSentinel;
    long long host[blocks];
    for (int sleep = 0; sleep < 2; sleep++)
        for (int stride : {1, 8}) {
            for (int rep = 0; rep < 3; rep++) {
                hipMemcpy(words, fill, sizeof(unsigned long long) * blocks * iters * 2, hipMemcpyHostToDevice);
                hipMemset(cycles, 0, sizeof(long long) * blocks);
                hipLaunchKernelGGL(pingpong, dim3(blocks), dim3(64), 0, 0, words, cycles, iters, stride, sleep);
                hipDeviceSynchronize();
            }
            hipMemcpy(host, cycles, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
            double sum = 0, mx = 0, mn = 1e30;
            int n = 0;
            for (int b = 0; b < blocks; b++)
                if (host[b] > 0) {
                    const double c = (double)host[b] / iters / 2.0;  // one way, s_memtime ticks
                    sum += c;
                    n++;
                    mx = c > mx ? c : mx;
                    mn = c < mn ? c : mn;
                }
            // s_memtime ticks at the shader clock here (the WTW stamps of this round: 24.5 k ticks per 10.1 us)
            printf("{\"partner\": \"b^%d\", \"s_sleep\": %d, \"pairs\": %d, \"one_way_ticks_mean\": %.2f, \"min\": %.2f, \"max\": %.2f, "
                   "\"one_way_us_at_2.4GHz\": %.3f}\n", stride, sleep, n, sum / n, mn, mx, sum / n / 2400.0);
        }
    return 0;
}
