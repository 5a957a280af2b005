// Offline DTW for gfx950: dtw.DTW(seq_a, seq_b) -> (cost, acc_cost, path)   (/root/reference/dtw.py:5-53)
//
//   dtw_cost_kernel       cost[i][j] = 1 - <a_i, b_j>  (K = 12: far too thin for MFMA to matter; one fma
//                         chain per element in dgemm's k-order, coalesced stores), whole chip.
//   dtw_sdp_kernel        the accumulated-cost recurrence as a strip DP (sdp.h): a wave owns 64 rows, its
//                         lanes are skewed in time and exchange predecessors by DPP; the waves of a
//                         workgroup and the workgroups of a launch form one pipeline down the matrix, so a
//                         single long pair spreads over many CUs and a batch of short pairs fills the chip.
//                         Every cell does the reference's three float64 adds and first-minimum argmin
//                         (dtw.py:35-40): acc_cost is bit-identical to the serial double loop.
//   dtw_hops_kernel /     dtw.py:43-52 over the packed step codes: the columns at which the path crosses the strip
//   dtw_segment_kernel    boundaries (one dependent load per strip), then every strip's segment walked by its own wave.
//   dtw_back_decode_kernel optional: the reference's `back` matrix as int8 [M][N].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"
#include "sdp.h"

namespace rts {

constexpr int kDtwF = 12;

struct DtwArgs {
    const void *a;  // [B][M][F] (a_stride frames between pairs; 0 = shared)
    const void *b;  // [B][N][F]
    double *cost;   // [B][M][N]
    double *acc;    // [B][M][N]
    int8_t *back;   // [B][M][N] or NULL
    int32_t *path;  // [B][M+N][2]
    int32_t *path_len;  // [B]
    long long a_stride, b_stride;
    int M, N, a_f64, b_f64;
    uint32_t *codes;           // [B][codes_words]
    unsigned long long *bnd;   // [B][n_strips][N]
    int32_t *entb;             // [B][n_strips][N]
    int32_t *cross, *lens;     // [B][n_strips]
    int32_t *pscr;             // [B][scratch_pairs(M, N)][2] path segments as walked (sdp::path_segment)
    double *yrec;              // [B][N][14] prepared column records
    int32_t *err;
    int32_t *ticket;           // [B] next row group of each pair (sdp::for_each_rowgroup)
    int n_rg, n_strips_wg;
};

// One thread per column j and kCostRows consecutive rows: b_j stays in registers, the a rows are wave-uniform
// (broadcast) loads, every store instruction writes 512 contiguous bytes of one row.
constexpr int kCostRows = 32;

template <bool A64, bool B64>
__global__ void __launch_bounds__(256) dtw_cost_kernel(DtwArgs g) {
    const int pair = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * kCostRows;
    if (j >= g.N) return;
    double b[kDtwF];
    sdp::load_frame(g.b, B64, (long long)pair * g.b_stride + j, b);
    double *out = g.cost + ((size_t)pair * g.M + i0) * g.N + j;
    const int rows = (g.M - i0 < kCostRows) ? g.M - i0 : kCostRows;
    for (int r = 0; r < rows; r++) {
        double a[kDtwF];
        sdp::load_frame(g.a, A64, (long long)pair * g.a_stride + i0 + r, a);
        double s = 0.0;
#pragma unroll
        for (int f = 0; f < kDtwF; f++) s = fma(a[f], b[f], s);
        out[(size_t)r * g.N] = 1.0 - s;
    }
}


__global__ void __launch_bounds__(256) dtw_prep_kernel(DtwArgs g) {
    const int pair = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= g.N) return;
    const void *b = g.b_f64 ? (const void *)(reinterpret_cast<const double *>(g.b) + (long long)pair * g.b_stride * kDtwF)
                            : (const void *)(reinterpret_cast<const float *>(g.b) + (long long)pair * g.b_stride * kDtwF);
    sdp::prep_column<sdp::DtwPolicy>(b, g.b_f64, j, g.yrec + (size_t)pair * g.N * sdp::kYRec);
}

// H helper waves per strip.  <2>: two strips per workgroup (6 waves); <3>: one strip per workgroup (4 waves, the DP wave
// has a SIMD to itself) -- see sdp::pick_config.
template <int H>
__global__ void __launch_bounds__(H == 2 ? 384 : 256) dtw_sdp_kernel(DtwArgs g) {
    extern __shared__ __align__(16) unsigned char dtw_smem[];
    const int pair = blockIdx.y;
    sdp::Problem pb;
    pb.x = g.a_f64 ? (const void *)(reinterpret_cast<const double *>(g.a) + (long long)pair * g.a_stride * kDtwF)
                   : (const void *)(reinterpret_cast<const float *>(g.a) + (long long)pair * g.a_stride * kDtwF);
    pb.x_f64 = g.a_f64;
    pb.yrec = g.yrec + (size_t)pair * g.N * sdp::kYRec;
    pb.M = g.M;
    pb.N = g.N;
    pb.D = g.acc + (size_t)pair * g.M * g.N;
    pb.ldD = g.N;
    pb.codes = g.codes + (size_t)pair * sdp::codes_words(g.M, g.N);
    pb.bnd = g.bnd + (size_t)pair * sdp::n_strips(g.M) * g.N;
    pb.entb = g.entb + (size_t)pair * sdp::n_strips(g.M) * g.N;
    pb.err = g.err;
    sdp::for_each_rowgroup(g.ticket + pair, g.n_rg, g.n_strips_wg, dtw_smem, [&](int rg) {
        sdp::run_rowgroup<sdp::DtwPolicy, true, H>(pb, rg, g.n_rg, g.n_strips_wg, dtw_smem);
    });
}

__global__ void __launch_bounds__(64) dtw_hops_kernel(DtwArgs g) {
    __shared__ uint32_t win[2 * sdp::kBtChunks * 64];
    const int pair = blockIdx.x, S = sdp::n_strips(g.M);
    sdp::path_hops(g.codes + (size_t)pair * sdp::codes_words(g.M, g.N), g.entb + (size_t)pair * S * g.N, g.M, g.N,
                   g.cross + (size_t)pair * S, win);
}

template <int PASS>
__global__ void __launch_bounds__(64) dtw_segment_kernel(DtwArgs g) {
    __shared__ uint32_t win[2 * sdp::kBtChunks * 64];
    const int pair = blockIdx.y, s = blockIdx.x, S = sdp::n_strips(g.M);
    int32_t *path = g.path + (size_t)pair * (g.M + g.N) * 2;
    sdp::path_segment(g.codes + (size_t)pair * sdp::codes_words(g.M, g.N), g.M, g.N, s, g.cross + (size_t)pair * S,
                      g.lens + (size_t)pair * S, PASS, path, g.path_len + pair, win,
                      g.pscr + (size_t)pair * 2 * sdp::scratch_pairs(g.M, g.N));
    if (PASS == 1 && s == 0 && threadIdx.x == 0 && *g.err != 0) g.path_len[pair] = -1;
}

// hops + both segment passes of a short pair in one launch (at most sdp::kTailStrips strips: one wave each)
__global__ void __launch_bounds__(64 * sdp::kTailStrips) dtw_tail_kernel(DtwArgs g) {
    extern __shared__ __align__(16) unsigned char dtw_smem[];
    const int pair = blockIdx.x, S = sdp::n_strips(g.M);
    sdp::path_tail(g.codes + (size_t)pair * sdp::codes_words(g.M, g.N), g.entb + (size_t)pair * S * g.N, g.M, g.N,
                   g.cross + (size_t)pair * S, g.lens + (size_t)pair * S, g.path + (size_t)pair * (g.M + g.N) * 2,
                   g.path_len + pair, reinterpret_cast<uint32_t *>(dtw_smem),
                   g.pscr + (size_t)pair * 2 * sdp::scratch_pairs(g.M, g.N));
    __syncthreads();
    if (threadIdx.x == 0 && *g.err != 0) g.path_len[pair] = -1;
}

__global__ void __launch_bounds__(256) dtw_back_decode_kernel(DtwArgs g) {
    const int pair = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= g.N) return;
    const uint32_t *codes = g.codes + (size_t)pair * sdp::codes_words(g.M, g.N);
    const int l = i & 63, t = j + l;
    const uint32_t w = codes[((size_t)(i >> 6) * sdp::n_chunks(g.N) + (t >> 4)) * 64 + l];
    g.back[((size_t)pair * g.M + i) * g.N + j] = (int8_t)((w >> (2 * (t & 15))) & 3);
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Workgroups of dtw_sdp_kernel<3> (one strip each) / <2> (two strips each) the current device holds at once;
// queried once per device and LDS padding (sdp::pick_config, "Residency").
static int dtw_residency(int &r1, int &r2) {
    static int cache[16][3];  // [device]: pad + 1, r1, r2
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t pad = sdp::lds_pad();
    int *c = (dev >= 0 && dev < 16) ? cache[dev] : nullptr;
    if (c && c[0] == (int)pad + 1) {
        r1 = c[1];
        r2 = c[2];
        return RTS_OK;
    }
    RTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dtw_sdp_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dtw_sdp_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    r1 = sdp::resident_blocks(dtw_sdp_kernel<3>, 256, sdp::lds_bytes(1) + pad);
    r2 = sdp::resident_blocks(dtw_sdp_kernel<2>, 384, sdp::lds_bytes(2) + pad);
    if (c) {
        c[1] = r1;
        c[2] = r2;
        c[0] = (int)pad + 1;
    }
    return RTS_OK;
}

}  // namespace rts

extern "C" {

int rts_dtw_workspace_bytes(int M, int N, int B, size_t *bytes) {
    using namespace rts;
    if (!bytes) return set_error(RTS_ERR_INVALID, "bytes is NULL");
    if (M < 1 || N < 1 || B < 1) return set_error(RTS_ERR_INVALID, "M, N, B must be >= 1");
    const size_t strips = (size_t)B * sdp::n_strips(M);
    *bytes = 256 + align256(sizeof(unsigned long long) * strips * N) + align256(sizeof(int32_t) * strips * N) +
             2 * align256(sizeof(int32_t) * strips) + align256(sizeof(double) * (size_t)B * N * sdp::kYRec) +
             align256(sizeof(uint32_t) * (size_t)B * sdp::codes_words(M, N)) +
             align256(sizeof(int32_t) * 2 * (size_t)B * sdp::scratch_pairs(M, N)) + align256(sizeof(int32_t) * (size_t)B);
    return RTS_OK;
}

int rts_dtw(const void *a_dev, int a_dtype, long long a_stride, const void *b_dev, int b_dtype,
            long long b_stride, int F, int M, int N, int B, double *cost_dev, double *acc_dev,
            int8_t *back_dev, int32_t *path_dev, int32_t *path_len_dev, void *ws_dev, size_t ws_bytes,
            void *stream) {
    using namespace rts;
    if (!a_dev || !b_dev || !cost_dev || !acc_dev || !path_dev || !path_len_dev || !ws_dev)
        return set_error(RTS_ERR_INVALID, "NULL device buffer");
    if (F != kDtwF) return set_error(RTS_ERR_UNSUPPORTED, "F must be 12 chroma bins (got %d)", F);
    if (M < 1 || N < 1 || B < 1) return set_error(RTS_ERR_INVALID, "M, N, B must be >= 1 (got %d %d %d)", M, N, B);
    if ((a_dtype != RTS_F32 && a_dtype != RTS_F64) || (b_dtype != RTS_F32 && b_dtype != RTS_F64))
        return set_error(RTS_ERR_INVALID, "bad dtype");
    if ((long long)M * N > 0x7fffffffLL * 4) return set_error(RTS_ERR_INVALID, "M*N too large");
    if (B > 65535) return set_error(RTS_ERR_INVALID, "at most 65535 pairs per call");
    size_t need = 0;
    rts_dtw_workspace_bytes(M, N, B, &need);
    if (ws_bytes < need)
        return set_error(RTS_ERR_INVALID, "workspace of %zu bytes is smaller than rts_dtw_workspace_bytes = %zu", ws_bytes, need);
    if (((uintptr_t)ws_dev & 15) != 0) return set_error(RTS_ERR_INVALID, "workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int strips = sdp::n_strips(M);
    int NS, H, G, res1 = 0, res2 = 0;
    if (int rc = dtw_residency(res1, res2); rc != RTS_OK) return rc;
    if (res1 < 1 && res2 < 1)
        return set_error(RTS_ERR_HIP, "the occupancy query reports no resident workgroup for the strip-DP kernel on this device");
    sdp::pick_config(strips, B, res1, res2, NS, H, G);
    const int n_rg = (strips + NS - 1) / NS;
    unsigned char *ws = reinterpret_cast<unsigned char *>(ws_dev);
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
    g.err = reinterpret_cast<int32_t *>(ws);
    {
        unsigned char *p = ws + 256;
        g.bnd = reinterpret_cast<unsigned long long *>(p);
        p += align256(sizeof(unsigned long long) * (size_t)B * strips * N);
        g.entb = reinterpret_cast<int32_t *>(p);
        p += align256(sizeof(int32_t) * (size_t)B * strips * N);
        g.cross = reinterpret_cast<int32_t *>(p);
        p += align256(sizeof(int32_t) * (size_t)B * strips);
        g.lens = reinterpret_cast<int32_t *>(p);
        p += align256(sizeof(int32_t) * (size_t)B * strips);
        g.yrec = reinterpret_cast<double *>(p);
        p += align256(sizeof(double) * (size_t)B * N * sdp::kYRec);
        g.codes = reinterpret_cast<uint32_t *>(p);
        p += align256(sizeof(uint32_t) * (size_t)B * sdp::codes_words(M, N));
        g.pscr = reinterpret_cast<int32_t *>(p);
        p += align256(sizeof(int32_t) * 2 * (size_t)B * sdp::scratch_pairs(M, N));
        g.ticket = reinterpret_cast<int32_t *>(p);
    }
    g.n_rg = n_rg;
    g.n_strips_wg = NS;
    RTS_HIP(hipMemsetAsync(g.err, 0, 16, s));
    RTS_HIP(hipMemsetAsync(g.ticket, 0, sizeof(int32_t) * (size_t)B, s));
    if (n_rg > 1) RTS_HIP(hipMemsetD32Async((hipDeviceptr_t)g.bnd, (int)sdp::kSentinel32, (size_t)2 * B * strips * N, s));
    {
        const dim3 grid((N + 255) / 256, (M + kCostRows - 1) / kCostRows, B);
        if (g.a_f64 && g.b_f64)
            hipLaunchKernelGGL((dtw_cost_kernel<true, true>), grid, dim3(256), 0, s, g);
        else if (g.a_f64)
            hipLaunchKernelGGL((dtw_cost_kernel<true, false>), grid, dim3(256), 0, s, g);
        else if (g.b_f64)
            hipLaunchKernelGGL((dtw_cost_kernel<false, true>), grid, dim3(256), 0, s, g);
        else
            hipLaunchKernelGGL((dtw_cost_kernel<false, false>), grid, dim3(256), 0, s, g);
    }
    RTS_HIP(hipGetLastError());
    hipLaunchKernelGGL(dtw_prep_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, g);
    const size_t smem = sdp::lds_bytes(NS) + sdp::lds_pad();
    if (H == 2)
        hipLaunchKernelGGL((dtw_sdp_kernel<2>), dim3(G, B), dim3(64 * NS * 3), smem, s, g);
    else
        hipLaunchKernelGGL((dtw_sdp_kernel<3>), dim3(G, B), dim3(64 * NS * 4), smem, s, g);
    RTS_HIP(hipGetLastError());
    if (strips <= sdp::kTailStrips) {
        hipLaunchKernelGGL(dtw_tail_kernel, dim3(B), dim3(64 * strips), sdp::tail_lds_bytes(strips), s, g);
    } else {
        hipLaunchKernelGGL(dtw_hops_kernel, dim3(B), dim3(64), 0, s, g);
        hipLaunchKernelGGL((dtw_segment_kernel<0>), dim3(strips, B), dim3(64), 0, s, g);
        hipLaunchKernelGGL((dtw_segment_kernel<1>), dim3(strips, B), dim3(64), 0, s, g);
    }
    RTS_HIP(hipGetLastError());
    if (back_dev) {
        hipLaunchKernelGGL(dtw_back_decode_kernel, dim3((N + 255) / 256, M, B), dim3(256), 0, s, g);
        RTS_HIP(hipGetLastError());
    }
    return RTS_OK;
}

}  // extern "C"
