// Offline DTW for gfx950: dtw.DTW(seq_a, seq_b) -> (cost, acc_cost, path)   (/root/reference/dtw.py:5-53)
//
//   dtw_cost_kernel   cost[i][j] = 1 - <a_i, b_j>  (the only GEMM-shaped op on this path, K = 12:
//                     far too thin for MFMA to matter; one fma chain per element in dgemm's
//                     k-order, coalesced stores), whole chip.
//   dtw_dp_kernel     one workgroup per (a, b) pair sweeps the anti-diagonals d = i + j.  Every
//                     cell of a diagonal depends only on the two previous diagonals, which live in
//                     three rotating LDS rows indexed by i; so each cell performs exactly the
//                     reference's three float64 adds and first-minimum argmin (dtw.py:35-40) and
//                     the result is bit-identical to the serial double loop.  The next diagonal's
//                     costs are fetched before the barrier to hide the strided global read.
//                     The same launch ends with the backtrack (dtw.py:43-52): one lane walks the
//                     back-pointers, then all threads reverse the path in place.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace rts {

constexpr int kDtwF = 12;

struct DtwArgs {
    const void *a;  // [B][M][F] (a_stride frames between pairs; 0 = shared)
    const void *b;  // [B][N][F]
    double *cost;   // [B][M][N]
    double *acc;    // [B][M][N]
    int8_t *back;   // [B][M][N]
    int32_t *path;  // [B][M+N][2]
    int32_t *path_len;  // [B]
    long long a_stride, b_stride;
    int M, N, a_f64, b_f64;
    double *diag_ws;  // [B][3][M] doubles in HBM when the three diagonals do not fit LDS, else NULL
};

__device__ __forceinline__ double dtw_load(const void *p, int f64, long long idx) {
    return f64 ? reinterpret_cast<const double *>(p)[idx] : (double)reinterpret_cast<const float *>(p)[idx];
}

__global__ void __launch_bounds__(256) dtw_cost_kernel(DtwArgs g) {
    const int pair = blockIdx.z;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= g.M || j >= g.N) return;
    const long long ao = ((long long)pair * g.a_stride + i) * kDtwF;
    const long long bo = ((long long)pair * g.b_stride + j) * kDtwF;
    double s = 0.0;
#pragma unroll
    for (int f = 0; f < kDtwF; f++) s = fma(dtw_load(g.a, g.a_f64, ao + f), dtw_load(g.b, g.b_f64, bo + f), s);
    g.cost[((size_t)pair * g.M + i) * g.N + j] = 1.0 - s;
}

// WS: the three diagonals live in an HBM workspace (long sequences) instead of LDS -- a compile-time choice so that
// the common case addresses them with LDS instructions, not through generic (flat) pointers.
template <int NT, bool WS>
__global__ void __launch_bounds__(NT) dtw_dp_kernel(DtwArgs g) {
    extern __shared__ __align__(16) unsigned char dtw_smem[];
    const int pair = blockIdx.x;
    // [3][M]: LDS, or (long sequences) an HBM workspace -- workgroup-scope visibility through the barrier
    // is all a single-workgroup sweep needs
    auto diag_row = [&](int which) {
        if constexpr (WS)
            return g.diag_ws + ((size_t)pair * 3 + which) * g.M;
        else
            return reinterpret_cast<double *>(dtw_smem) + (size_t)which * g.M;
    };
    const int tid = threadIdx.x;
    const int M = g.M, N = g.N;
    const size_t base = (size_t)pair * M * N;
    const double *cost = g.cost + base;
    double *acc = g.acc + base;
    int8_t *back = g.back + base;

    // rows handled by this thread: i = tid + r*NT.  The register prefetch covers r < kPf rows and runs two
    // diagonals ahead (one diagonal is well under a microsecond of work, a cost read from HBM/L2 takes longer); the
    // three register sets rotate by name (the loop is unrolled by three), so no copy forces an early wait.  Any
    // further rows (M > kPf*NT) read their cost directly.
    constexpr int kPf = (NT >= 1024) ? 2 : 4;  // 1024 threads leave 128 registers each: three sets of two
    double ca[kPf], cb[kPf], cc[kPf];
    auto fetch = [&](double (&dst)[kPf], int dd) {
#pragma unroll
        for (int r = 0; r < kPf; r++) {
            const int i = tid + r * NT;
            const int j = dd - i;
            dst[r] = (i < M && j >= 0 && j < N) ? cost[(size_t)i * N + j] : 0.0;
        }
    };
    const int n_diag = M + N - 1;
    auto step = [&](const double (&pre)[kPf], int d) {
        auto cur = diag_row(d % 3);
        const auto p1 = diag_row((d + 2) % 3);  // diagonal d-1
        const auto p2 = diag_row((d + 1) % 3);  // diagonal d-2
        for (int r = 0, i = tid; i < M; r++, i += NT) {
            const int j = d - i;
            if (j < 0 || j >= N) continue;
            double c;
            if (r < kPf) {
                c = pre[0];
#pragma unroll
                for (int q = 1; q < kPf; q++) c = (r == q) ? pre[q] : c;
            } else {
                c = cost[(size_t)i * N + j];
            }
            double best;
            int s;
            if (i == 0 && j == 0) {
                best = c;
                s = 2;  // dtw.py:20-21
            } else if (j == 0) {
                best = c + p1[i - 1];  // dtw.py:24
                s = 1;
            } else if (i == 0) {
                best = c + p1[0];  // dtw.py:27
                s = 0;
            } else {
                const double o0 = p1[i] + c;          // (i, j-1)
                const double o1 = p1[i - 1] + c;      // (i-1, j)
                const double o2 = p2[i - 1] + 2 * c;  // (i-1, j-1)
                best = o0;
                s = 0;
                if (o1 < best) {
                    best = o1;
                    s = 1;
                }
                if (o2 < best) {
                    best = o2;
                    s = 2;
                }
            }
            cur[i] = best;
            acc[(size_t)i * N + j] = best;
            back[(size_t)i * N + j] = (int8_t)s;
        }
        // the diagonals are the only cross-thread traffic: LDS-only barrier when they live in LDS, so that the acc /
        // back-pointer stores of this diagonal stay in flight
        if constexpr (WS)
            __syncthreads();
        else
            lds_barrier();
    };
    fetch(ca, 0);
    fetch(cb, 1);
    for (int d = 0; d < n_diag; d += 3) {
        fetch(cc, d + 2);
        step(ca, d);
        if (d + 1 < n_diag) {
            fetch(ca, d + 3);
            step(cb, d + 1);
        }
        if (d + 2 < n_diag) {
            fetch(cb, d + 4);
            step(cc, d + 2);
        }
    }
    __syncthreads();  // back-pointers visible to the lane that walks them

    // ---- backtrack (dtw.py:43-52).  One lane walks the back-pointers, but not through HBM one dependent load at a
    // time: the workgroup stages the 64 x 64 tile whose bottom-right corner is the walk's position in LDS, the lane
    // walks until it leaves the tile (at least 64 steps), and so on.
    constexpr int kT = 64;
    __shared__ int s_len, s_i, s_j;
    __shared__ int8_t s_tile[kT][kT + 4];
    int32_t *path = g.path + (size_t)pair * (M + N) * 2;
    if (tid == 0) {
        path[0] = M - 1;
        path[1] = N - 1;
        s_len = 1;
        s_i = M - 1;
        s_j = N - 1;
    }
    __syncthreads();
    while (s_i > 0 || s_j > 0) {  // uniform: shared values only change behind barriers
        const int i0 = s_i, j0 = s_j;
        const int ti = (i0 - kT + 1 > 0) ? i0 - kT + 1 : 0, tj = (j0 - kT + 1 > 0) ? j0 - kT + 1 : 0;
        const int th = i0 - ti + 1, tw = j0 - tj + 1;
        for (int idx = tid; idx < th * tw; idx += NT) {
            const int r = idx / tw, q = idx - r * tw;
            s_tile[r][q] = back[(size_t)(ti + r) * N + (tj + q)];
        }
        __syncthreads();
        if (tid == 0) {
            int i = i0, j = j0, len = s_len;
            while (i >= ti && j >= tj && (i > 0 || j > 0)) {
                const int s = s_tile[i - ti][j - tj];
                if (s == 0)
                    j -= 1;
                else if (s == 1)
                    i -= 1;
                else {
                    i -= 1;
                    j -= 1;
                }
                path[2 * len] = i;
                path[2 * len + 1] = j;
                len++;
            }
            s_len = len;
            s_i = i;
            s_j = j;
        }
        __syncthreads();
    }
    if (tid == 0) g.path_len[pair] = s_len;
    __syncthreads();
    const int len = s_len;
    for (int p = tid; p < len / 2; p += NT) {  // path.reverse()
        const int q = len - 1 - p;
        const int x0 = path[2 * p], y0 = path[2 * p + 1];
        const int x1 = path[2 * q], y1 = path[2 * q + 1];
        path[2 * p] = x1;
        path[2 * p + 1] = y1;
        path[2 * q] = x0;
        path[2 * q + 1] = y0;
    }
}

}  // namespace rts

extern "C" {

int rts_dtw_workspace_bytes(int M, int N, int B, size_t *back_bytes) {
    using namespace rts;
    if (!back_bytes) return set_error(RTS_ERR_INVALID, "back_bytes is NULL");
    if (M < 1 || N < 1 || B < 1) return set_error(RTS_ERR_INVALID, "M, N, B must be >= 1");
    *back_bytes = (size_t)B * M * N;
    return RTS_OK;
}

int rts_dtw(const void *a_dev, int a_dtype, long long a_stride, const void *b_dev, int b_dtype,
            long long b_stride, int F, int M, int N, int B, double *cost_dev, double *acc_dev,
            int8_t *back_dev, int32_t *path_dev, int32_t *path_len_dev, void *stream) {
    return rts_dtw_ws(a_dev, a_dtype, a_stride, b_dev, b_dtype, b_stride, F, M, N, B, cost_dev, acc_dev, back_dev,
                      path_dev, path_len_dev, nullptr, stream);
}

int rts_dtw_ws(const void *a_dev, int a_dtype, long long a_stride, const void *b_dev, int b_dtype,
               long long b_stride, int F, int M, int N, int B, double *cost_dev, double *acc_dev,
               int8_t *back_dev, int32_t *path_dev, int32_t *path_len_dev, double *diag_ws_dev, void *stream) {
    using namespace rts;
    if (!a_dev || !b_dev || !cost_dev || !acc_dev || !back_dev || !path_dev || !path_len_dev)
        return set_error(RTS_ERR_INVALID, "NULL device buffer");
    if (F != kDtwF) return set_error(RTS_ERR_UNSUPPORTED, "F must be 12 chroma bins (got %d)", F);
    if (M < 1 || N < 1 || B < 1) return set_error(RTS_ERR_INVALID, "M, N, B must be >= 1 (got %d %d %d)", M, N, B);
    if ((a_dtype != RTS_F32 && a_dtype != RTS_F64) || (b_dtype != RTS_F32 && b_dtype != RTS_F64))
        return set_error(RTS_ERR_INVALID, "bad dtype");
    size_t smem = sizeof(double) * 3 * (size_t)M;
    const bool in_lds = smem <= 150 * 1024;
    if (!in_lds && !diag_ws_dev)
        return set_error(RTS_ERR_UNSUPPORTED,
                         "M=%d rows exceed the %d the LDS-resident DP sweep holds; call rts_dtw_ws with a "
                         "B*3*M-double workspace", M, (int)(150 * 1024 / 24));
    if (!in_lds) smem = 64;
    if ((long long)M * N > 0x7fffffffLL * 4) return set_error(RTS_ERR_INVALID, "M*N too large");
    hipStream_t s = (hipStream_t)stream;
    DtwArgs g;
    g.a = a_dev;
    g.b = b_dev;
    g.cost = cost_dev;
    g.acc = acc_dev;
    g.back = back_dev;
    g.path = path_dev;
    g.path_len = path_len_dev;
    g.a_stride = a_stride;
    g.b_stride = b_stride;
    g.M = M;
    g.N = N;
    g.a_f64 = a_dtype == RTS_F64;
    g.b_f64 = b_dtype == RTS_F64;
    g.diag_ws = in_lds ? nullptr : diag_ws_dev;
    hipLaunchKernelGGL(dtw_cost_kernel, dim3((N + 63) / 64, (M + 3) / 4, B), dim3(256), 0, s, g);
    RTS_HIP(hipGetLastError());
    // one row per thread up to 1024 rows; fewer waves for small M keeps the per-diagonal barrier cheap
    if (!in_lds) {
        hipLaunchKernelGGL((dtw_dp_kernel<1024, true>), dim3(B), dim3(1024), smem, s, g);
    } else if (M <= 256) {
        static bool done = false;
        if (!done) {
            RTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dtw_dp_kernel<256, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            done = true;
        }
        hipLaunchKernelGGL((dtw_dp_kernel<256, false>), dim3(B), dim3(256), smem, s, g);
    } else if (M <= 512) {
        static bool done = false;
        if (!done) {
            RTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dtw_dp_kernel<512, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            done = true;
        }
        hipLaunchKernelGGL((dtw_dp_kernel<512, false>), dim3(B), dim3(512), smem, s, g);
    } else {
        static bool done = false;
        if (!done) {
            RTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dtw_dp_kernel<1024, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            done = true;
        }
        hipLaunchKernelGGL((dtw_dp_kernel<1024, false>), dim3(B), dim3(1024), smem, s, g);
    }
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

}  // extern "C"
