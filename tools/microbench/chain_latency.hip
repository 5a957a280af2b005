// Dependent-issue latency of the instructions the DP chains of csrc/otw.hip, csrc/sdp.h and csrc/wtw.hip are made of, one
// wave alone on its SIMD (what the DP wave of a strip / the chain wave of an OTW step is): N dependent instances of each
// pattern, timed with HIP events (ns per instance) and with s_memtime (ticks per instance).  A pattern of known cost --
// dependent v_add_f32: 4 core cycles per pass of a 64-lane wave on a 16-lane SIMD, back-to-back issue -- is measured
// beside the others; it gives the core clock of the run, and everything is also reported in core cycles through it.
//
//   hipcc --offload-arch=gfx950 -O3 -o chain_latency tools/microbench/chain_latency.hip && ./chain_latency
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// pattern id -> 64 dependent instances per loop iteration
template <int P>
__global__ void chain(double *out, long long *ticks, int iters, double seed) {
    double a = seed + threadIdx.x, b = seed * 0.5, c = 1.0, d2 = 2.0, d3 = 3.0;
    float fa = (float)seed + threadIdx.x, fb = 1.5f;
    int ia = threadIdx.x, lo = 0, hi = 0;
    __shared__ double lds[128];
    lds[threadIdx.x] = a;
    lds[threadIdx.x + 64] = b;
    __syncthreads();
    double *lp = lds + threadIdx.x;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int k = 0; k < iters; k++) {
        if (P == 0) { REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));) }
        if (P == 1) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (P == 2) { REP64(asm volatile("v_min_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (P == 3) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(b));) }
        // the strip DP's / window DP's lane-to-lane hop: result -> two DPP moves (wave_shr:1) -> add -> next result
        if (P == 4) {
            REP64(asm volatile("s_nop 1\n\tv_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                               "v_add_u32 %0, %1, %2" : "+v"(ia), "+v"(lo), "+v"(hi));)
        }
        // one full DTW cell on the chain, as the compiler schedules it: lane hop of the value, v_add_f64, 2 x v_min_f64
        if (P == 5) {
#pragma unroll
            for (int u = 0; u < 64; u++) {
                const int l2 = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x138, 0xf, 0xf, false);
                const int h2 = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x138, 0xf, 0xf, false);
                double up = __hiloint2double(h2, l2), o0 = a + b, o1 = up + b, m, r;
                asm volatile("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(o0), "v"(o1));
                asm volatile("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(m), "v"(c));
                a = r;
            }
        }
        // LDS round trip: store then dependent load of the same word
        if (P == 6) {
            REP64(asm volatile("ds_write_b64 %1, %0\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(a) : "v"((unsigned)(size_t)lp) : "memory");)
        }
        // v_readlane -> SALU -> v_mov (the scalar detour of a walk step / of OTW's control decisions)
        if (P == 7) {
            REP64(asm volatile("v_readlane_b32 s20, %0, 5\n\ts_add_u32 s20, s20, 1\n\tv_mov_b32 %0, s20" : "+v"(ia) : : "s20", "scc");)
        }
        // v_cmp -> v_cndmask (a select on the chain, 64-bit: two cndmasks)
        if (P == 8) {
            REP64(asm volatile("v_cmp_lt_f64 vcc, %0, %2\n\tv_cndmask_b32 %1, %1, %3, vcc\n\tv_add_f64 %0, %0, %2"
                               : "+v"(a), "+v"(lo) : "v"(b), "v"(hi) : "vcc");)
        }
        // two independent chains interleaved in one wave (what cost_pair and the two-cells-in-flight helpers do)
        if (P == 9) { REP64(asm volatile("v_add_f64 %0, %0, %2\n\tv_add_f64 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
        // four independent chains
        if (P == 10) {
            REP64(asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4"
                               : "+v"(a), "+v"(c), "+v"(d2), "+v"(d3) : "v"(b));)
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a + fa + ia + lo + hi + c + d2 + d3;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int P>
static double run(const char *name, double *out, long long *ticks, double base_ns) {
    fprintf(stderr, "running %s\n", name);
    const int iters = 2000;
    long long h = 0;
    float best_ms = 1e30f;
    long long best_ticks = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(chain<P>, dim3(1), dim3(64), 0, 0, out, ticks, iters, 1.25);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, ticks, sizeof(h), hipMemcpyDeviceToHost);
        if (ms < best_ms) best_ms = ms, best_ticks = h;
    }
    const double n = iters * 64.0;
    const double ns = best_ms * 1e6 / n;  // includes ~10 us of launch overhead in ~ms of run time
    if (base_ns > 0)
        printf("{\"pattern\": \"%s\", \"ns_per_instance\": %.3f, \"memtime_ticks_per_instance\": %.2f, \"core_cycles_if_dependent_v_add_f32_is_4\": %.1f}\n",
               name, ns, best_ticks / n, 4.0 * ns / base_ns);
    else
        printf("{\"pattern\": \"%s\", \"ns_per_instance\": %.3f, \"memtime_ticks_per_instance\": %.2f, \"implied_core_clock_GHz_if_4_cycles\": %.3f}\n",
               name, ns, best_ticks / n, 4.0 / ns);
    fflush(stdout);
    return ns;
}

int main() {
    double *out;
    long long *ticks;
    hipMalloc(&out, sizeof(double) * 64);
    hipMalloc(&ticks, sizeof(long long));
    const double base = run<0>("dependent v_add_f32", out, ticks, 0);
    run<1>("dependent v_add_f64", out, ticks, base);
    run<2>("dependent v_min_f64", out, ticks, base);
    run<3>("dependent v_fma_f64", out, ticks, base);
    run<4>("s_nop 1 + 2 x v_mov_b32_dpp wave_shr:1 + v_add_u32 (lane hop)", out, ticks, base);
    run<5>("DTW cell chain: lane hop + v_add_f64 + 2 x v_min_f64", out, ticks, base);
    run<6>("LDS round trip: ds_write_b64 + ds_read_b64 + wait", out, ticks, base);
    run<7>("v_readlane -> s_add -> v_mov", out, ticks, base);
    run<8>("v_cmp_lt_f64 + v_cndmask + v_add_f64", out, ticks, base);
    run<9>("2 independent v_add_f64 chains, per pair", out, ticks, base);
    run<10>("4 independent v_add_f64 chains, per group of 4", out, ticks, base);
    return 0;
}
