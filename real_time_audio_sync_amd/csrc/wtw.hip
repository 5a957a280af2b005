// Windowed time warping for gfx950: the chroma-level half of wtw.WTW.insert
// (/root/reference/wtw.py:92-128) with get_cost_matrix (:162-171), run_dtw (:173-217) and
// find_path (:219-240), batched over B independent live streams against one reference chroma.
//
// One workgroup per stream walks its newly appended live chroma columns; whenever a full window
// of W live frames is available it
//   1. computes the W x W normalised-cosine cost with the reference's dot orders (norms: fma
//      chain = x.dot(x) on a contiguous copy; cross term: OpenBLAS strided ddot order),
//   2. sweeps the anti-diagonals of the unit-weight DP (candidates (i-1,j), (i,j-1), (i-1,j-1),
//      strict '<' in that order, codes 3/1/2) with three rotating float64 diagonals in LDS -- all
//      cells of a diagonal are independent, so D and B are bit-identical to the serial loops,
//   3. backtracks B and applies the hand-over rule of wtw.py:107-128 (append sub-path points with
//      l <= dtw_hop/hop, move (live_ptr, ref_ptr) to the last appended point) on one lane.
// The back-pointer matrix lives in LDS for W <= 128 and in an HBM workspace above that.
//
// Windows of more than 64 frames (wtw_live.py's W = 100 up to BASELINE configs[4]'s W = 10 000, where one window is
// 1e8 cells) run as a strip DP (sdp.h) spread over many workgroups: per window one launch of
// wtw_big_dp_kernel (a pipeline of row groups down the W x W matrix, step codes packed 2 bits per cell), the
// backtrack kernels (wtw_big_hops_kernel, wtw_big_segment_kernel: every strip's path segment by its own wave) and
// wtw_big_ctl_kernel (hand-over, then the column bookkeeping of
// wtw.py:92-100 up to the next window).  The host enqueues as many such rounds as the pushed columns can
// possibly complete windows; rounds with nothing pending return at once.  Everything stays asynchronous.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "sdp.h"

namespace rts {

constexpr int kWF = 12;
constexpr int kWtwNT = 256;
constexpr int kWtwLdsB = 128;  // largest W whose back-pointers stay in LDS
constexpr int kWtwLdsW = 512;  // largest W whose window (features, norms, diagonals) stays in LDS
constexpr int kWtwStripFrom = 64;  // windows above this many frames use the strip DP (sdp.h)
constexpr int kWtwMaxW = 16384;

struct WtwArgs {
    const double *ref;    // [M][F]
    double *live;         // [B][N][F]  (N = 2M)
    int32_t *appended;    // [B] columns written to `live` so far
    int32_t *state;       // [B][8]: chroma_ptr, live_ptr, ref_ptr, status, n_path, n_windows, cells_lo, cells_hi
    int32_t *path;        // [B][path_cap][2]
    int8_t *bwork;        // [B][W][W] or NULL when W <= kWtwLdsB
    double *dlast;        // [B][W][W] last window's D (optional, NULL = not stored)
    // W > kWtwLdsW (strip DP):
    int32_t *ws_sub;           // [B][4W] ints: the window's sub-path, reversed
    int32_t *ws_scr;           // [B][scratch_pairs(W, W)][2]: its segments as walked (sdp::path_segment)
    int32_t *ctl;              // [B][8]: pending, live_ptr, ref_ptr, n, m of the window being computed
    uint32_t *codes;           // [B][codes_words(W, W)] packed step codes
    unsigned long long *bnd;   // [B][n_strips(W)][W] rows handed between row groups
    int32_t *entb;             // [B][n_strips(W)][W] entry columns of the strips' bottom rows (sdp.h)
    int32_t *cross, *lens;     // [B][n_strips(W)] strip-boundary crossings / segment lengths of the window's path
    double *yrec;              // [B][W][14] prepared records of the window's reference columns
    int32_t *err;
    int32_t *ticket;           // [B] next row group of the pending window (sdp::for_each_rowgroup); zeroed by the control step
    int n_rg, n_strips_wg;
    int M, N, W, hopf, path_cap;
    int fill_separate;         // long windows: the hand-over's fill and column records by wtw_big_fill_kernel
};

__device__ __forceinline__ double wtw_dot_chain(const double *x, const double *y) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < kWF; i++) s = fma(x[i], y[i], s);
    return s;
}

__device__ __forceinline__ double wtw_dot_strided(const double *x, const double *y) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int i = 0; i < kWF; i += 4) {
        const double m3 = y[i + 2] * x[i + 2];
        const double m4 = y[i + 3] * x[i + 3];
        const double a = fma(y[i], x[i], m3);
        const double b = fma(y[i + 1], x[i + 1], m4);
        t1 = t1 + a;
        t2 = t2 + b;
    }
    return t1 + t2;
}

// Windows of up to kWtwLdsW frames: window state in LDS, 256 threads.  BL: back-pointers in LDS (W <= kWtwLdsB)
// rather than HBM.  Compile-time, so that every pointer has a known address space.
template <bool BL>
__global__ void __launch_bounds__(256) wtw_advance_kernel(WtwArgs g) {
    extern __shared__ __align__(16) unsigned char wtw_smem[];
    const int W = g.W;
    const int b = blockIdx.x, tid = threadIdx.x;
    double *xs_l = reinterpret_cast<double *>(wtw_smem);  // [W][F] live window
    double *ys_l = xs_l + (size_t)W * kWF;                // [W][F] ref window
    double *nx = ys_l + (size_t)W * kWF;                  // [W]
    double *ny = nx + W;                                  // [W]
    double *diag = ny + W;                                // [3][W]
    int32_t *sub = reinterpret_cast<int32_t *>(diag + 3 * (size_t)W);  // [2W][2]
    int8_t *bl = reinterpret_cast<int8_t *>(sub + 4 * (size_t)W);  // [W][W] when W <= kWtwLdsB
    __shared__ int s_chroma_ptr, s_live_ptr, s_ref_ptr, s_status, s_n_path, s_n_windows, s_go;
    __shared__ long long s_cells;

    int32_t *st = g.state + (size_t)b * 8;
    const double *live = g.live + (size_t)b * g.N * kWF;
    auto Bm = [&]() {
        if constexpr (BL)
            return bl;
        else
            return g.bwork + (size_t)b * W * W;
    }();
    // `appended` may exceed the capacity N: the excess columns were dropped by the append kernel and mean
    // "the next column does not fit" (wtw.py:92 would raise IndexError) once the stored ones are consumed
    const int appended_raw = g.appended[b];
    const int appended = appended_raw < g.N ? appended_raw : g.N;

    if (tid == 0) {
        s_chroma_ptr = st[0];
        s_live_ptr = st[1];
        s_ref_ptr = st[2];
        s_status = st[3];
        s_n_path = st[4];
        s_n_windows = st[5];
        s_cells = ((long long)(uint32_t)st[7] << 32) | (uint32_t)st[6];
    }
    __syncthreads();

    for (;;) {
        // ---- one new column (wtw.py:92-97), decided by lane 0, broadcast through s_go
        if (tid == 0) {
            int go = 0;
            if (s_status == RTS_RUNNING && s_chroma_ptr < appended) {
                s_chroma_ptr += 1;
                if (s_ref_ptr >= (g.M - 1 - W) || s_live_ptr >= (g.N - 1 - W))
                    s_status = RTS_STOP_REF_END;
                else
                    go = 1;
            }
            s_go = go;
        }
        __syncthreads();
        if (!s_go) break;
        // ---- windows (wtw.py:100-128)
        while (s_chroma_ptr - s_live_ptr >= W) {  // uniform: shared values only change behind barriers
            const int lp = s_live_ptr, rp = s_ref_ptr;
            const int n = W;
            int m = W;
            if (rp + m > g.M) m = g.M - rp;  // numpy slice truncation of chroma_ref[:, rp:rp+W]
            if (m <= 0) break;
            for (int idx = tid; idx < n * kWF; idx += kWtwNT) xs_l[idx] = live[(size_t)lp * kWF + idx];
            for (int idx = tid; idx < m * kWF; idx += kWtwNT) ys_l[idx] = g.ref[(size_t)rp * kWF + idx];
            const double *xs = xs_l, *ys = ys_l;
            __syncthreads();
            for (int i = tid; i < n; i += kWtwNT) nx[i] = sqrt(wtw_dot_chain(xs + i * kWF, xs + i * kWF));
            for (int j = tid; j < m; j += kWtwNT) ny[j] = sqrt(wtw_dot_chain(ys + j * kWF, ys + j * kWF));
            __syncthreads();
            const int n_diag = n + m - 1;
            for (int d = 0; d < n_diag; d++) {
                double *cur = diag + (size_t)(d % 3) * W;
                const double *p1 = diag + (size_t)((d + 2) % 3) * W;
                const double *p2 = diag + (size_t)((d + 1) % 3) * W;
                for (int i = tid; i < n; i += kWtwNT) {
                    const int j = d - i;
                    if (j < 0 || j >= m) continue;
                    const double dot = wtw_dot_strided(xs + i * kWF, ys + j * kWF);
                    const double c = 1.0 - dot / (nx[i] * ny[j]);  // wtw.py:169
                    double dv;
                    int8_t code;
                    if (i == 0 && j == 0) {
                        dv = c;
                        code = 0;
                    } else if (j == 0) {
                        dv = p1[i - 1] + c;  // wtw.py:187-191
                        code = 3;
                    } else if (i == 0) {
                        dv = p1[0] + c;  // wtw.py:194-198
                        code = 1;
                    } else {
                        double mc = p1[i - 1];  // (i-1, j)
                        code = 3;
                        const double v1 = p1[i];  // (i, j-1)
                        if (v1 < mc) {
                            mc = v1;
                            code = 1;
                        }
                        const double v2 = p2[i - 1];  // (i-1, j-1)
                        if (v2 < mc) {
                            mc = v2;
                            code = 2;
                        }
                        dv = mc + c;
                    }
                    cur[i] = dv;
                    Bm[(size_t)i * W + j] = code;
                    if (g.dlast) g.dlast[((size_t)b * W + i) * W + j] = dv;
                }
                lds_barrier();  // diagonals in LDS: leave the back-pointer / D stores in flight
            }
            __syncthreads();  // back-pointers (HBM for W > 128) visible to the lane that walks them
            if (tid == 0) {
                // find_path (wtw.py:219-240): walk back from (n-1, m-1); sub[] holds it reversed
                int i = n - 1, j = m - 1, len = 0;
                sub[0] = i;
                sub[1] = j;
                len = 1;
                while (!(i == 0 && j == 0) && len < 2 * W) {
                    const int8_t p = Bm[(size_t)i * W + j];
                    if (p == 1)
                        j -= 1;
                    else if (p == 2) {
                        i -= 1;
                        j -= 1;
                    } else
                        i -= 1;
                    sub[2 * len] = i;
                    sub[2 * len + 1] = j;
                    len++;
                }
                // hand-over (wtw.py:107-128), iterating the sub-path forwards
                int change = 0, idx_l = 0, idx_r = 0;
                int32_t *path = g.path + (size_t)b * g.path_cap * 2;
                for (int q = len - 1; q >= 0; q--) {
                    const int l = sub[2 * q], r = sub[2 * q + 1];
                    if (l <= g.hopf) {
                        if (s_n_path < g.path_cap) {
                            path[2 * s_n_path] = l + lp;
                            path[2 * s_n_path + 1] = r + rp;
                        }
                        s_n_path += 1;
                        idx_l = l;
                        idx_r = r;
                    } else {
                        change = 1;
                        break;
                    }
                }
                if (change) {
                    s_live_ptr = lp + idx_l;
                    s_ref_ptr = rp + idx_r;
                } else {
                    s_live_ptr = lp + g.hopf;
                    s_ref_ptr = rp + g.hopf;
                }
                s_n_windows += 1;
                s_cells += (long long)n * m;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (s_status == RTS_RUNNING && s_chroma_ptr >= g.N && appended_raw > g.N) s_status = RTS_LIVE_OVERFLOW;
        st[0] = s_chroma_ptr;
        st[1] = s_live_ptr;
        st[2] = s_ref_ptr;
        st[3] = s_status;
        st[4] = s_n_path;
        st[5] = s_n_windows;
        st[6] = (int32_t)(uint32_t)(s_cells & 0xffffffffLL);
        st[7] = (int32_t)(uint32_t)((unsigned long long)s_cells >> 32);
    }
}

// ---- windows of at most kWinMaxW frames: every window of a push in ONE launch, one workgroup per stream -------------
//
// wtw_live.py runs W = 100 / hop = 50, tests.py:174 W = 20 / hop = 10: thousands of small windows per stream, each one
// depending on the hand-over of the one before.  wtw_win_kernel<R, STAGE> keeps a stream's whole window loop on the device.
// Per window k, between workgroup barriers:
//   A. waves 1.. : the n x m cost matrix (1 - x.y / (|x| |y|), wtw.py:169, the reference's dot orders) into LDS -- lanes
//      own columns (reference frame and norm in registers), the waves share the rows (each stages its own rows' live
//      frames and their norms in a wave-private slice of LDS: no barrier inside the phase);
//      wave 0, at the same time: find_path + hand-over of window k - 1 (below).
//   B. row 0 and column 0 of D are running sums (wtw.py:187-198): two lanes accumulate them in the reference's order.
//   C. the interior on R waves, one matrix row per lane, lanes skewed in time (lane l of wave r owns row 1 + 64 r + l and
//      handles column 1 + t - l at step t): the three predecessors of a cell are registers -- the lane's own previous
//      value, the previous value of the lane above (DPP wave_shr:1; lane 0 receives row 0, or wave 0's bottom row, from
//      LDS) and what that move delivered one step earlier.  Same float64 operations and candidate order as
//      wtw.py:201-215 (sdp::WtwPolicy::cell), so D and the path are bit-identical.  The block's 16 costs are fetched
//      into registers before its 16 dependent steps.  With R = 2 (65 < W <= 128) wave 1 runs 5 blocks behind wave 0; one
//      LDS barrier per 16 steps.  Step codes: 2 bits per cell, 16 steps of a lane per dword.
//      Beside the values the lanes propagate, for every cell below row h = dtw_hop / hop, the column at which the cell's
//      best path leaves row h (the same recurrence driven by the step codes; row h + 1 seeds it).  At the last cell that
//      is the column r* of the last point the hand-over will append (wtw.py:113: l <= h), hence the next window's
//      pointers (live_ptr + h, ref_ptr + r*; wtw.py:118-124) are known the moment the DP ends -- without the path.
//   wave 0 in phase A of the next iteration walks the path back from (h, r*) -- only the part that is handed over; from
//      (n-1, m-1) when the window is no taller than the hop -- with the position in SGPRs and the row's code word one
//      v_readlane away (find_path, wtw.py:219-240), and appends it (wtw.py:107-117).
// The column bookkeeping up to the next window is in closed form (the same rules as wtw_ctl_body below).
constexpr int kWinMaxW = 128;   // what the kernel can do (RTS_WTW_WIN=1 forces it up to here)
constexpr int kWinAutoW = 128;  // what it is chosen for by default
constexpr int kWinKW = 12;       // code words per lane: 64 + 127 - 1 steps at most
constexpr int kWinPadFront = 64;  // doubles in front of / behind the cost matrix: lanes that are not on a valid cell read
constexpr int kWinPadBack = 208;  // (and ignore) whatever their row pointer + step lands on

// wave 0's bottom row (bot / botx): 64 slots in front and 32 behind for the steps at which lane 63 is not on a valid cell,
// and 80 more where every other lane's share of the same unconditional store lands
constexpr int kWinBotFront = 64, kWinBotBack = 32, kWinBotDummy = 80;
constexpr int kWinBotPad = kWinBotFront + kWinBotBack + kWinBotDummy;

// Windows of at most kWinPrefW frames (tests.py:174: 20): while the DP of window k runs, the idle cost waves fetch what the
// cost phase of window k + 1 will need -- its live rows (live_ptr + hop is known) into xs / nx, and the 128 reference frames
// from ref_ptr on (the next window starts at ref_ptr + r*, r* < W, so its columns are among them) into ypre / nypre -- so
// that phase A reads LDS instead of waiting for L2 (3 k of the 5 k cycles of a 20-frame window's cost phase).
constexpr int kWinPrefW = 64;   // (the one-wave kernel's whole range)
constexpr int kWinPrefN = 128;  // reference frames fetched: two per lane of the fetching waves

__host__ __device__ inline int win_ldc(int W) { return (W | 1) + 1; }  // even > W: lanes a row apart hit different LDS banks
__host__ __device__ inline size_t win_lds_bytes(int W) {
    const size_t feat = sizeof(double) * ((size_t)W * kWF + (size_t)W);                    // xs, nx
    const size_t walk = sizeof(uint32_t) * 2 * kWinKW * 64 + sizeof(double) * (3 * (size_t)W + 32 + kWinBotPad) +  // codes; row 0, column 0 (padded), bottom row of wave 0 (padded)
                        sizeof(int32_t) * ((size_t)W + kWinBotPad) + sizeof(int32_t) * 4 * (size_t)W;  // its crossing columns (padded); sub-path
    const size_t pref = (W <= kWinPrefW) ? sizeof(double) * (kWinPrefN * kWF + kWinPrefN) : 0;  // ypre, nypre
    return sizeof(double) * ((size_t)W * win_ldc(W) + kWinPadFront + kWinPadBack) + feat + walk + pref + 128;
}

#ifdef RTS_WIN_STAMPS
// diagnostic build only, stream 0: [0] phase A as wave 0 sees it (walk + hand-over + wait for the cost waves), [1] running
// sums, [2] DP, [3] wave 0's own walk + hand-over, [4] wave 1's own cost phase, [5] windows
__device__ long long g_win_stamps[8];
#define RTS_WIN_STAMP(slot)                                                       \
    do {                                                                          \
        const long long now_ = (long long)__builtin_amdgcn_s_memtime();           \
        if (b == 0 && tid == 0) g_win_stamps[slot] += now_ - stamp_t;             \
        stamp_t = now_;                                                           \
    } while (0)
#else
#define RTS_WIN_STAMP(slot) \
    do {                    \
    } while (0)
#endif

// One block of 16 DP steps of wtw_win_kernel, as a macro so that each DP wave gets its own straight-line copy (W1: this is the
// second wave; NSTEP: 16, or 8 for a short last block -- it fetches the crossing columns of wave 0's bottom row for its lane 0 and has no row to hand on; the first
// wave does the opposite).  Not a lambda: inside a generic lambda the compiler lowers the step's selects to exec-mask branch
// ladders (135 branches per block), and a branch inside the block costs the overlap between consecutive steps.  The block's
// 16 costs and 16 values of the row above lane 0 are fetched first (one LDS round trip per block instead of one per step);
// every lane stores every step -- lane 63 to bot[column] (column = t - 62 is hit exactly once per window; steps at which the
// lane is not on a valid cell land in the padding), the others to a scratch strip -- because a conditional store would be a
// branch again (228 cycles per step with it, 139 in the one-wave kernel that has none).
#define RTS_WIN_DP_BLOCK(W1, NSTEP)                                                                                \
    {                                                                                                              \
                double cbuf[16], ubuf[16]; \
                int uxbuf[16]; \
_Pragma("unroll") \
                for (int q = 0; q < (NSTEP); q++) { \
                    const int t = 16 * kb + q; \
                    cbuf[q] = crow[t]; \
                    ubuf[q] = upin[(t + 1 < m) ? t + 1 : 0]; \
                    uxbuf[q] = (W1) ? botx[(t + 1 < m) ? t + 1 : 0] : 0; \
                } \
                double *bw = (lane == 63) ? bot + (16 * kb - 62) : bot_dummy + lane; \
                int32_t *bxw = (lane == 63) ? botx + (16 * kb - 62) : botx_dummy + lane; \
                word = 0; \
_Pragma("unroll") \
                for (int q = 0; q < (NSTEP); q++) { \
                    const int t = 16 * kb + q; \
                    const unsigned jj = (unsigned)(t - lane); \
                    const bool incol = jj < (unsigned)mint; \
                    const bool valid = incol && lane < rows; \
                    const double du = sdp::shr1(dlast_v, ubuf[q]); \
                    const int xu = sdp::shr1_i(xl, uxbuf[q]); \
                    double dv; \
                    int code; \
                    sdp::WtwPolicy::cell(false, false, du, dlast_v, du_prev, cbuf[q], dv, code); \
                    int xn = (code == sdp::kUp) ? xu : ((code == sdp::kDiag) ? xu_prev : xl); \
                    const int xh = (code == sdp::kUp) ? (int)jj + 1 : ((code == sdp::kDiag) ? (int)jj : xl); \
                    xn = is_h1 ? xh : xn; \
                    du_prev = du; \
                    xu_prev = xu; \
                    word |= (uint32_t)code << (2 * q); \
                    if (incol) { \
                        dlast_v = dv; \
                        xl = xn; \
                    } \
                    if constexpr (R > 1 && !(W1)) { /* only the first DP wave has a row to hand on */ \
                        bw[q] = dv; \
                        bxw[q] = xn; \
                    } \
                    if (STAGE && valid) dout[jj + 1] = dv; \
                } \
    }

template <int R, bool STAGE>
__global__ void __launch_bounds__(R == 1 ? 256 : 512) wtw_win_kernel(WtwArgs g) {
    constexpr int NT = (R == 1) ? 256 : 512;  // two waves per SIMD for the wide windows: their cost phase is fp64-bound
    constexpr int NCW = NT / 64 - 1;          // cost waves: 1 .. NCW (wave 0 walks the previous window's path meanwhile)
    extern __shared__ __align__(16) unsigned char wtw_smem[];
    const int W = g.W, ldc = win_ldc(W);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int32_t *sh = reinterpret_cast<int32_t *>(wtw_smem);                          // [32] control words
    double *C = reinterpret_cast<double *>(wtw_smem + 128) + kWinPadFront;        // [W][ldc]
    double *xs = C + (size_t)W * ldc + kWinPadBack;                               // [W][F] live window (row i staged by wave 1 + i % NCW)
    double *nx = xs + (size_t)W * kWF;                                            // [W]
    uint32_t *codes = reinterpret_cast<uint32_t *>(nx + W);                       // [2][kWinKW][64]
    double *row0 = reinterpret_cast<double *>(codes + 2 * kWinKW * 64);           // [W + 16] D[0][:]
    double *col0 = row0 + W + 16;                                                 // [W + 16] D[:][0]
    double *bot = col0 + W + 16 + kWinBotFront;                                   // [W] D[64][:] (wave 0's last row), padded
    double *bot_dummy = bot + W + kWinBotBack;                                    // [80]
    int32_t *botx = reinterpret_cast<int32_t *>(bot_dummy + kWinBotDummy) + kWinBotFront;  // [W] ... and its crossing columns, padded
    int32_t *botx_dummy = botx + W + kWinBotBack;                                 // [80]
    int32_t *sub = botx_dummy + kWinBotDummy;                                     // [2W][2], reversed
    double *ypre = reinterpret_cast<double *>(sub + 4 * (size_t)W);              // [kWinPrefN][F] reference frames pref_rp .. (W <= kWinPrefW)
    double *nypre = ypre + kWinPrefN * kWF;                                       // [kWinPrefN] their norms

    int32_t *st = g.state + (size_t)b * 8;
    const double *live = g.live + (size_t)b * g.N * kWF;
    int32_t *path = g.path + (size_t)b * g.path_cap * 2;
    const int appended_raw = g.appended[b];
    const int appended = appended_raw < g.N ? appended_raw : g.N;
    // state in registers (uniform: every thread computes the same values; n_path is kept by wave 0 only)
    int chroma_ptr = st[0], live_ptr = st[1], ref_ptr = st[2], status = st[3], n_path = st[4], n_windows = st[5];
    long long cells = ((long long)(uint32_t)st[7] << 32) | (uint32_t)st[6];
    // the window whose path wave 0 still has to walk: its pointers and where the walk starts
    int have_prev = 0, plp = 0, prp = 0, pwi = 0, pwj = 0;
    int have_pref = 0, pref_lp = 0, pref_rp = 0;  // what the cost waves fetched during the last DP phase
    const int h = g.hopf;

    for (;;) {
        // ---- wtw.py:92-100 in closed form up to the next window (see wtw_ctl_body)
        int pending = 0, m = 0;
        const int n = W;
        if (status == RTS_RUNNING && chroma_ptr < appended) {
            if (ref_ptr >= (g.M - 1 - W) || live_ptr >= (g.N - 1 - W)) {
                chroma_ptr += 1;
                status = RTS_STOP_REF_END;
            } else if (live_ptr + W <= appended) {
                chroma_ptr = live_ptr + W;
                m = W;
                if (ref_ptr + m > g.M) m = g.M - ref_ptr;
                pending = m > 0 ? 1 : 0;
                if (!pending) chroma_ptr = appended;
            } else {
                chroma_ptr = appended;
            }
        }
        if (!pending && !have_prev) break;
        const int lp = live_ptr, rp = ref_ptr;
#ifdef RTS_WIN_STAMPS
        long long stamp_t = (long long)__builtin_amdgcn_s_memtime();
#endif
        // ---- A. wave 0: path + hand-over of the previous window; waves 1..: this window's cost matrix
        if (wave == 0) {
            if (have_prev) {
#ifdef RTS_WIN_STAMPS
                const long long w0_ = (long long)__builtin_amdgcn_s_memtime();
#endif
                // find_path (wtw.py:219-240) from the last point that is handed over back to (0, 0); sub[] holds it reversed
                int i = pwi, j = pwj, len = 1;
                if (lane == 0) {
                    sub[0] = i;
                    sub[1] = j;
                }
                int cur = -1;
                uint32_t cw = 0;
                while (i > 0 && j > 0) {  // interior cells: the code of (i, j) is bit pair t & 15 of word t >> 4 of lane (i-1) & 63
                    const int l = (i - 1) & 63;
                    const int t = l + j - 1;
                    const int key = ((i - 1) >> 6) * kWinKW + (t >> 4);
                    if (key != cur) {  // uniform
                        cw = codes[(size_t)key * 64 + lane];
                        cur = key;
                    }
                    const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)cw, l);
                    const int code = (int)((w >> (2 * (t & 15))) & 3u);
                    i -= (code != sdp::kLeft) ? 1 : 0;   // kUp, kDiag
                    j -= (code != sdp::kUp) ? 1 : 0;     // kLeft, kDiag
                    if (lane == 0) {
                        sub[2 * len] = i;
                        sub[2 * len + 1] = j;
                    }
                    len++;
                }
                // on row 0 the path runs left, on column 0 up (wtw.py:187-198): the rest is a straight line to (0, 0)
                const int rest = i + j;  // one of them is 0
                for (int q = lane; q < rest; q += 64) {
                    sub[2 * (len + q)] = i > 0 ? i - 1 - q : 0;
                    sub[2 * (len + q) + 1] = j > 0 ? j - 1 - q : 0;
                }
                len += rest;
                // hand-over (wtw.py:107-117): every walked point has l <= dtw_hop / hop; forwards
                for (int f = lane; f < len; f += 64) {
                    if (n_path + f < g.path_cap) {
                        path[2 * (size_t)(n_path + f)] = sub[2 * (len - 1 - f)] + plp;
                        path[2 * (size_t)(n_path + f) + 1] = sub[2 * (len - 1 - f) + 1] + prp;
                    }
                }
                n_path += len;
#ifdef RTS_WIN_STAMPS
                if (b == 0 && lane == 0) g_win_stamps[3] += (long long)__builtin_amdgcn_s_memtime() - w0_;
#endif
            }
        } else if (pending) {
#ifdef RTS_WIN_STAMPS
            const long long c0_ = (long long)__builtin_amdgcn_s_memtime();
#endif
            const int cwv = wave - 1;
            double y0[kWF], y1[kWF];
            double ny0;
            const int nrows = (n - cwv + NCW - 1) / NCW;
            // (uniform) everything this window needs was fetched during the previous window's DP phase
            const bool pref = (R == 1) && W <= kWinPrefW && have_pref && lp == pref_lp && rp >= pref_rp && rp - pref_rp + m <= kWinPrefN;
            if (pref) {
                const int off = rp - pref_rp + (lane < m ? lane : m - 1);
#pragma unroll
                for (int f = 0; f < kWF; f++) y0[f] = ypre[off * kWF + f];
                ny0 = nypre[off];
            } else {
                // my columns' reference frames first (their latency hides behind the staging of the rows)
                sdp::load_frame(g.ref, 1, (long long)rp + (lane < m ? lane : m - 1), y0);
                if (m > 64) sdp::load_frame(g.ref, 1, (long long)rp + (64 + lane < m ? 64 + lane : m - 1), y1);
                // my rows (cwv, cwv + NCW, ...) into my slice of xs / nx: wave-private, no barrier
                for (int q = lane; q < nrows * kWF; q += 64) {
                    const int row = cwv + NCW * (q / kWF);
                    xs[row * kWF + q % kWF] = live[(size_t)(lp + row) * kWF + q % kWF];
                }
                for (int q = lane; q < nrows; q += 64) {
                    const int row = cwv + NCW * q;
                    double x[kWF];
#pragma unroll
                    for (int f = 0; f < kWF; f++) x[f] = xs[row * kWF + f];
                    nx[row] = sdp::WtwPolicy::norm(x);
                }
                ny0 = sdp::WtwPolicy::norm(y0);
            }
            const double ny1 = (m > 64) ? sdp::WtwPolicy::norm(y1) : 1.0;
            for (int i = cwv; i < n; i += NCW) {
                double x[kWF];
#pragma unroll
                for (int f = 0; f < kWF; f++) x[f] = xs[i * kWF + f];  // wave-uniform: a broadcast
                const double nxi = nx[i];
                const double c0 = sdp::WtwPolicy::cost(x, nxi, y0, ny0);  // wtw.py:169
                if (lane < m) C[(size_t)i * ldc + lane] = c0;
                if (m > 64) {
                    const double c1 = sdp::WtwPolicy::cost(x, nxi, y1, ny1);
                    if (64 + lane < m) C[(size_t)i * ldc + 64 + lane] = c1;
                }
            }
#ifdef RTS_WIN_STAMPS
            if (b == 0 && tid == 64) g_win_stamps[4] += (long long)__builtin_amdgcn_s_memtime() - c0_;
#endif
        }
        __syncthreads();
        RTS_WIN_STAMP(0);
        if (!pending) {  // that was the last window's path
            have_prev = 0;
            continue;
        }
        // ---- B. row 0 (lane 0) and column 0 (lane 1) of D: running sums in the reference's order (wtw.py:183-198).  16
        // costs per LDS round trip, then 16 dependent adds; the stores run up to 15 elements past the end (the arrays
        // are padded for it), so nothing in the loop depends on the element count
        if (wave == 0 && lane < 2) {
            const int cnt = lane == 0 ? m : n;
            const double *src = C;
            const int stride = lane == 0 ? 1 : ldc;
            double *out = lane == 0 ? row0 : col0;
            double acc = 0.0;
            for (int q0 = 0; q0 < cnt; q0 += 16) {
                double v[16];
#pragma unroll
                for (int q = 0; q < 16; q++) v[q] = src[q * stride];  // (past the end: the padding behind C)
                src += 16 * stride;
                if (q0 == 0) {
                    acc = v[0];
                    out[0] = acc;
#pragma unroll
                    for (int q = 1; q < 16; q++) {
                        acc = acc + v[q];
                        out[q] = acc;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        acc = acc + v[q];
                        out[q] = acc;
                    }
                }
                out += 16;
            }
        }
        __syncthreads();
        if (STAGE) {
            double *dl = g.dlast + (size_t)b * W * W;
            for (int q = tid; q < m; q += NT) dl[q] = row0[q];
            for (int q = tid; q < n; q += NT) dl[(size_t)q * W] = col0[q];
        }
        RTS_WIN_STAMP(1);
        // ---- C. interior DP: lane l of wave r owns row 1 + 64 r + l, column 1 + t - l at local step t
        const bool has_cross = h + 1 <= n - 1;  // the path goes below row h: wtw.py:118-124's "change"
        if (R == 1 && W <= kWinPrefW && wave >= 1) {  // the idle cost waves: what the next window's cost phase will read (see kWinPrefW)
            const int cwv = wave - 1, nlp = lp + h;
            const int nrows = (n - cwv + NCW - 1) / NCW;
            for (int q = lane; q < nrows * kWF; q += 64) {
                const int row = cwv + NCW * (q / kWF);
                const int fr = (nlp + row < g.N) ? nlp + row : g.N - 1;
                xs[row * kWF + q % kWF] = live[(size_t)fr * kWF + q % kWF];
            }
            for (int q = lane; q < nrows; q += 64) {
                const int row = cwv + NCW * q;
                double x[kWF];
#pragma unroll
                for (int f = 0; f < kWF; f++) x[f] = xs[row * kWF + f];
                nx[row] = sdp::WtwPolicy::norm(x);
            }
            if (wave <= 2) {  // waves 1 and 2: 64 reference frames each
                const int slot = 64 * (wave - 1) + lane;
                double yv[kWF];
                sdp::load_frame(g.ref, 1, (long long)((rp + slot < g.M) ? rp + slot : g.M - 1), yv);
#pragma unroll
                for (int f = 0; f < kWF; f++) ypre[slot * kWF + f] = yv[f];
                nypre[slot] = sdp::WtwPolicy::norm(yv);
            }
        }
        {
            const int r = wave;
            const int nint = n - 1, mint = m - 1;                       // interior rows / columns
            const int rows = (r < R) ? ((nint - 64 * r) < 64 ? (nint - 64 * r) : 64) : 0;
            const int T = (rows > 0 && mint > 0) ? rows + mint - 1 : 0;
            const int blocks = (T + 15) >> 4;
            const int T0 = (nint > 0 && mint > 0) ? (nint < 64 ? nint : 64) + mint - 1 : 0;
            const int T1 = (nint > 64 && mint > 0) ? (nint - 64) + mint - 1 : 0;
            const int rounds0 = (T0 + 15) >> 4, rounds1 = T1 > 0 ? 5 + ((T1 + 15) >> 4) : 0;
            const int rounds_all = (R == 1 || rounds0 > rounds1) ? rounds0 : rounds1;
            const int i = 1 + 64 * r + lane;
            const int ic = i < n ? i : n - 1;
            // D[i][0] and D[i-1][0]: what the first interior step of this lane sees as `left` and `diag`
            double dlast_v = col0[ic], du_prev = col0[ic - 1 >= 0 ? ic - 1 : 0];
            // crossing columns: column 0 is left through (h, 0)
            int xl = 0, xu_prev = 0;
            const bool is_h1 = (i == h + 1);
            const double *upin = (r == 0) ? row0 : bot;                  // the row above this wave's lane 0
            double *dout = STAGE ? g.dlast + ((size_t)b * W + ic) * W : nullptr;
            const double *crow = C + (size_t)ic * ldc + 1 - lane;         // crow[t] = C[i][1 + t - lane]
            const bool short_tail = (R == 1) && blocks > 0 && (T - 16 * (blocks - 1) <= 8);  // wave-uniform
            const int loop_rounds = short_tail ? rounds_all - 1 : rounds_all;
            for (int k = 0; k < loop_rounds; k++) {
                if (R > 1) lds_barrier();  // wave 1 reads bottom-row values wave 0 wrote at least one round ago
                const int kb = k - 5 * r;
                if (kb < 0 || kb >= blocks) continue;  // wave-uniform
#ifdef RTS_WIN_STAMPS
                const long long blk_t0_ = (long long)__builtin_amdgcn_s_memtime();
#endif
                uint32_t word = 0;
                if (R > 1 && r > 0) RTS_WIN_DP_BLOCK(true, 16) else RTS_WIN_DP_BLOCK(false, 16)
                codes[((size_t)r * kWinKW + kb) * 64 + lane] = word;
#ifdef RTS_WIN_STAMPS
                if (b == 0 && lane == 0 && r < 2) g_win_stamps[6 + r] += (long long)__builtin_amdgcn_s_memtime() - blk_t0_;  // a DP wave's own block time
#endif
            }
            if (short_tail) {  // one DP wave: a last block of at most 8 steps runs as 8 (tests.py:174's 20-frame windows sweep 37 steps)
                const int kb = blocks - 1;
                uint32_t word = 0;
                RTS_WIN_DP_BLOCK(false, 8)
                codes[((size_t)r * kWinKW + kb) * 64 + lane] = word;
            }
            if (r < R && lane < rows && i == n - 1) sh[2] = xl;  // the last cell's crossing column
        }
        __syncthreads();
        RTS_WIN_STAMP(2);
        // ---- the next window's pointers (wtw.py:118-128), and what wave 0 walks during the next phase A
        const int xstar = (has_cross && m > 1) ? sh[2] : 0;  // (m == 1: the path is column 0)
        have_pref = (R == 1 && W <= kWinPrefW) ? 1 : 0;
        pref_lp = lp + h;
        pref_rp = rp;
        plp = lp;
        prp = rp;
        pwi = has_cross ? h : n - 1;
        pwj = has_cross ? xstar : m - 1;
        have_prev = 1;
        live_ptr = lp + h;
        ref_ptr = rp + (has_cross ? xstar : h);
        n_windows += 1;
        cells += (long long)n * m;
#ifdef RTS_WIN_STAMPS
        if (b == 0 && tid == 0) g_win_stamps[5] += 1;
#endif
    }
    if (status == RTS_RUNNING && chroma_ptr >= g.N && appended_raw > g.N) status = RTS_LIVE_OVERFLOW;
    if (tid == 0) {
        st[0] = chroma_ptr;
        st[1] = live_ptr;
        st[2] = ref_ptr;
        st[3] = status;
        st[4] = n_path;
        st[5] = n_windows;
        st[6] = (int32_t)(uint32_t)(cells & 0xffffffffLL);
        st[7] = (int32_t)(uint32_t)((unsigned long long)cells >> 32);
    }
}

// ---- windows of more than kWtwLdsW frames: strip DP over many workgroups (sdp.h) ------------------------------

// H helper waves per strip: <.., 2> two strips per workgroup, <.., 3> one strip per workgroup (sdp::pick_config).
template <bool STAGE, int H>
__global__ void __launch_bounds__(H == 2 ? 384 : 256) wtw_big_dp_kernel(WtwArgs g) {
    extern __shared__ __align__(16) unsigned char wtw_smem[];
    const int b = blockIdx.y;
    const int32_t *ctl = g.ctl + (size_t)b * 8;
    if (ctl[0] == 0) return;  // no window pending for this stream
    const int lp = ctl[1];
    sdp::Problem pb;
    pb.x = g.live + ((size_t)b * g.N + lp) * kWF;  // rows: the live window (wtw.py:101)
    pb.x_f64 = 1;
    pb.yrec = g.yrec + (size_t)b * g.W * sdp::kYRec;  // columns: the reference window (wtw.py:102), prepared by ctl
    pb.M = ctl[3];
    pb.N = ctl[4];
    pb.D = STAGE ? g.dlast + (size_t)b * g.W * g.W : nullptr;
    pb.ldD = g.W;
    pb.codes = g.codes + (size_t)b * sdp::codes_words(g.W, g.W);
    pb.bnd = g.bnd + (size_t)b * sdp::n_strips(g.W) * g.W;
    pb.entb = g.entb + (size_t)b * sdp::n_strips(g.W) * g.W;
    pb.err = g.err;
    sdp::for_each_rowgroup(g.ticket + b, g.n_rg, g.n_strips_wg, wtw_smem, [&](int rg) {
        sdp::run_rowgroup<sdp::WtwPolicy, STAGE, H>(pb, rg, g.n_rg, g.n_strips_wg, wtw_smem);
    });
}

// find_path (wtw.py:219-240) for the pending window of each stream, over the packed step codes (sdp.h): the
// strip-boundary crossings, then every strip's segment by its own wave -- count, then write -- into ws_sub,
// forward-ordered; ctl[5] receives the sub-path's length.
__global__ void __launch_bounds__(64) wtw_big_hops_kernel(WtwArgs g) {
    __shared__ uint32_t win[2 * sdp::kBtChunks * 64];
    const int b = blockIdx.x, S = sdp::n_strips(g.W);
    const int32_t *ctl = g.ctl + (size_t)b * 8;
    if (ctl[0] == 0) return;
    sdp::path_hops(g.codes + (size_t)b * sdp::codes_words(g.W, g.W), g.entb + (size_t)b * S * g.W, ctl[3], ctl[4],
                   g.cross + (size_t)b * S, win);
}

template <int PASS>
__global__ void __launch_bounds__(64) wtw_big_segment_kernel(WtwArgs g) {
    __shared__ uint32_t win[2 * sdp::kBtChunks * 64];
    const int b = blockIdx.y, s = blockIdx.x, S = sdp::n_strips(g.W);
    int32_t *ctl = g.ctl + (size_t)b * 8;
    if (ctl[0] == 0 || s >= sdp::n_strips(ctl[3])) return;
    sdp::path_segment(g.codes + (size_t)b * sdp::codes_words(g.W, g.W), ctl[3], ctl[4], s, g.cross + (size_t)b * S,
                      g.lens + (size_t)b * S, PASS, g.ws_sub + (size_t)b * 4 * g.W, ctl + 5, win,
                      g.ws_scr + (size_t)b * 2 * sdp::scratch_pairs(g.W, g.W));
}

// Hands a window to the DP launch that follows: its boundary words start as "not written", and its reference columns
// become float64 records with their norms.  `tid` of `nt` threads share the work.
__device__ __forceinline__ void wtw_window_handover(const WtwArgs &g, int b, int ref_ptr, int nm, size_t tid, size_t nt) {
    const int W = g.W;
    unsigned long long *bnd = g.bnd + (size_t)b * sdp::n_strips(W) * W;
    const size_t words = (size_t)(g.n_rg > 1 ? g.n_rg - 1 : 0) * nm;
    for (size_t k = tid; k < words; k += nt) bnd[k] = sdp::kSentinel;
    for (size_t col = tid; col < (size_t)nm; col += nt)
        sdp::prep_column<sdp::WtwPolicy>(g.ref, 1, (long long)ref_ptr + (long long)col,
                                         g.yrec + (size_t)b * W * sdp::kYRec - (size_t)ref_ptr * sdp::kYRec);
}

// Long windows (W = 10 000: 12.5 MB of boundary words): the hand-over by many workgroups instead of the control
// kernel's one -- after it, reading the window it has just announced in ctl[].
__global__ void __launch_bounds__(256) wtw_big_fill_kernel(WtwArgs g) {
    const int b = blockIdx.y;
    const int32_t *ctl = g.ctl + (size_t)b * 8;
    if (ctl[0] == 0) return;
    wtw_window_handover(g, b, ctl[2], ctl[4], (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

// One workgroup per stream.  If a window is pending (its sub-path was just written by the kernels above): the
// hand-over (wtw.py:107-128).  Then the column bookkeeping of wtw.py:92-100 in closed
// form up to the next event: between two windows the stop test (wtw.py:96) sees constant pointers, and a window
// fires exactly when chroma_ptr reaches live_ptr + W.
__device__ __forceinline__ void wtw_ctl_body(const WtwArgs &g) {
    __shared__ int s_cnt;
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    const int W = g.W;
    int32_t *st = g.state + (size_t)b * 8;
    int32_t *ctl = g.ctl + (size_t)b * 8;
    int32_t *sub = g.ws_sub + (size_t)b * 4 * W;
    int32_t *path = g.path + (size_t)b * g.path_cap * 2;
    const int pending = ctl[0];
    int lp = ctl[1], rp = ctl[2];
    const int n = ctl[3], m = ctl[4];
    int live_ptr = st[1], ref_ptr = st[2], n_path = st[4];
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (pending) {
        // sub[] holds the path from (0, 0) to (n-1, m-1); l is non-decreasing along it, so the points handed over
        // (l <= dtw_hop / hop, wtw.py:113) are a prefix of it
        // (written by another wave of this workgroup when the backtrack ran in the same launch: not through L1)
        int len = __hip_atomic_load(ctl + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        len = len < 1 ? 1 : (len > 2 * W ? 2 * W : len);
        int local = 0;
        for (int q = tid; q < len; q += NT) local += (sub[2 * q] <= g.hopf) ? 1 : 0;
        if (local) atomicAdd(&s_cnt, local);
        __syncthreads();
        const int cnt = s_cnt;
        for (int f = tid; f < cnt; f += NT) {
            if (n_path + f < g.path_cap) {
                path[2 * (size_t)(n_path + f)] = sub[2 * f] + lp;
                path[2 * (size_t)(n_path + f) + 1] = sub[2 * f + 1] + rp;
            }
        }
        if (cnt < len && cnt >= 1) {  // "change": the path went past the hop (wtw.py:118-124)
            live_ptr = lp + sub[2 * (cnt - 1)];  // the last appended point
            ref_ptr = rp + sub[2 * (cnt - 1) + 1];
        } else {
            live_ptr = lp + g.hopf;
            ref_ptr = rp + g.hopf;
        }
        n_path += cnt;
    }
    __syncthreads();
    // ---- bookkeeping up to the next event (uniform across the workgroup: everything below depends on state only)
    const int appended_raw = g.appended[b];
    const int appended = appended_raw < g.N ? appended_raw : g.N;
    int chroma_ptr = st[0], status = st[3], n_windows = st[5];
    long long cells = ((long long)(uint32_t)st[7] << 32) | (uint32_t)st[6];
    if (pending) {
        n_windows += 1;
        cells += (long long)n * m;
    }
    int next_pending = 0, nn = 0, nm = 0;
    if (status == RTS_RUNNING && chroma_ptr < appended) {
        if (ref_ptr >= (g.M - 1 - W) || live_ptr >= (g.N - 1 - W)) {
            chroma_ptr += 1;  // wtw.py:92-97: the next column trips the boundary check
            status = RTS_STOP_REF_END;
        } else if (live_ptr + W <= appended) {
            chroma_ptr = live_ptr + W;  // the column that completes the window (wtw.py:100)
            nn = W;
            nm = W;
            if (ref_ptr + nm > g.M) nm = g.M - ref_ptr;  // numpy slice truncation of chroma_ref[:, rp:rp+W]
            next_pending = nm > 0 ? 1 : 0;
            if (!next_pending) chroma_ptr = appended;  // unreachable (wtw.py:96 stops first); never stall
        } else {
            chroma_ptr = appended;
        }
    }
    if (!next_pending && status == RTS_RUNNING && chroma_ptr >= g.N && appended_raw > g.N) status = RTS_LIVE_OVERFLOW;
    if (next_pending && !g.fill_separate) wtw_window_handover(g, b, ref_ptr, nm, tid, NT);
    __syncthreads();
    if (tid == 0) {
        st[0] = chroma_ptr;
        st[1] = live_ptr;
        st[2] = ref_ptr;
        st[3] = (*g.err != 0 && status == RTS_RUNNING) ? RTS_DEVICE_FAULT : status;
        st[4] = n_path;
        st[5] = n_windows;
        st[6] = (int32_t)(uint32_t)(cells & 0xffffffffLL);
        st[7] = (int32_t)(uint32_t)((unsigned long long)cells >> 32);
        g.ticket[b] = 0;  // the DP launch that follows hands its row groups out from 0 again
        ctl[0] = next_pending;
        ctl[1] = live_ptr;
        ctl[2] = ref_ptr;
        ctl[3] = nn;
        ctl[4] = nm;
    }
}

__global__ void __launch_bounds__(1024) wtw_big_ctl_kernel(WtwArgs g) { wtw_ctl_body(g); }

// Backtrack (if a window is pending) and control step in one launch, for windows of at most sdp::kTailStrips strips.
__global__ void __launch_bounds__(64 * sdp::kTailStrips) wtw_big_tail_ctl_kernel(WtwArgs g) {
    extern __shared__ __align__(16) unsigned char wtw_smem[];
    const int b = blockIdx.x, S = sdp::n_strips(g.W);
    int32_t *ctl = g.ctl + (size_t)b * 8;
    if (ctl[0] != 0)  // uniform over the workgroup
        sdp::path_tail(g.codes + (size_t)b * sdp::codes_words(g.W, g.W), g.entb + (size_t)b * S * g.W, ctl[3], ctl[4],
                       g.cross + (size_t)b * S, g.lens + (size_t)b * S, g.ws_sub + (size_t)b * 4 * g.W, ctl + 5,
                       reinterpret_cast<uint32_t *>(wtw_smem), g.ws_scr + (size_t)b * 2 * sdp::scratch_pairs(g.W, g.W));
    __syncthreads();
    wtw_ctl_body(g);
}

// wtw.py:76-77: the check made at the top of insert(), before any column is processed.
__global__ void wtw_precheck_kernel(int32_t *state, int B, int M, int N) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int32_t *st = state + (size_t)b * 8;
    if (st[3] == RTS_RUNNING && (st[2] >= M - 1 || st[1] >= N - 1)) st[3] = RTS_STOP_REF_END;
}

// Append n_new[b] columns from cols [B][n_max][F] to the live history; columns beyond the 2M capacity
// are dropped (the reference would raise IndexError at that column) and reported as status
// LIVE_OVERFLOW by the advance kernel once it has walked the columns that did fit.
// grid (B, slices): a push of a whole recording is 200 KB per stream, so several workgroups share a stream's copy.  Every
// slice reads the old count from `appended`; slice 0 writes the new one to `appended_next`, and the host swaps the two
// arrays after the launch (the kernels that follow read the new one).
constexpr int kWtwAppendSlices = 8;
__global__ void wtw_append_kernel(double *live, int32_t *appended, int32_t *appended_next, int32_t *state, const void *cols,
                                  int cols_f64, const int32_t *n_new, int n_uniform, int n_max, int B, int N) {
    const int b = blockIdx.x;
    if (b >= B) return;
    int nn = n_new ? n_new[b] : n_uniform;
    nn = nn < 0 ? 0 : (nn > n_max ? n_max : nn);  // never read a neighbouring stream's columns
    const int base_raw = appended[b];
    const int base = base_raw < N ? base_raw : N;
    const int running = state[(size_t)b * 8 + 3] == RTS_RUNNING;
    int take = nn;
    if (base + take > N) take = N - base > 0 ? N - base : 0;
    if (blockIdx.y == 0 && threadIdx.x == 0)
        appended_next[b] = (!running || nn <= 0) ? base_raw : ((take < nn) ? N + 1 : base + take);  // N+1: a column was dropped
    if (!running) return;  // sticky stop: later columns are ignored
    const int total = take * kWF;
    const int per = (total + gridDim.y - 1) / gridDim.y;
    const int lo = blockIdx.y * per, hi = (lo + per < total) ? lo + per : total;
    for (int idx = lo + threadIdx.x; idx < hi; idx += blockDim.x) {
        const size_t src = ((size_t)b * n_max) * kWF + idx;
        const double v = cols_f64 ? reinterpret_cast<const double *>(cols)[src]
                                  : (double)reinterpret_cast<const float *>(cols)[src];
        live[((size_t)b * N + base) * kWF + idx] = v;
    }
}

}  // namespace rts

struct rts_wtw {
    const double *ref;
    int M, N, B, W, hopf, path_cap;
    double *live;
    int32_t *appended, *appended_next, *state, *path;
    int8_t *bwork;
    double *dlast;
    int32_t *ws_sub, *ws_scr, *ctl, *err, *ticket, *entb, *cross, *lens;
    double *yrec;
    uint32_t *codes;
    unsigned long long *bnd;
    int big_waves, big_helpers, n_rg, big_grid, use_big;
    int use_win;  // windows of at most kWinMaxW frames: wtw_win_kernel (0: off, 1 / 2: DP waves)
    int device;  // the HIP device the handle's buffers live on
    size_t smem;
};

namespace rts {
// A handle belongs to the device that was current at rts_wtw_create (like rts_otw handles).
static int wtw_check_device(const rts_wtw *h) {
    int d = -1;
    RTS_HIP(hipGetDevice(&d));
    if (d != h->device)
        return set_error(RTS_ERR_INVALID, "handle was created on device %d but device %d is current "
                                          "(one process per GPU, or hipSetDevice before the call)", h->device, d);
    return RTS_OK;
}
}  // namespace rts

extern "C" {

int rts_wtw_create(const double *chroma_ref_dev, int F, int M, int B, int win_frames, int hop_frames, int keep_last_d,
                   rts_wtw **out) {
    using namespace rts;
    if (!out) return set_error(RTS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!chroma_ref_dev) return set_error(RTS_ERR_INVALID, "chroma_ref_dev is NULL");
    if (F < 1) return set_error(RTS_ERR_INVALID, "F must be >= 1 (got %d)", F);
    if (F != kWF) return set_error(RTS_ERR_UNSUPPORTED, "F must be 12 chroma bins (got %d)", F);
    if (M < 1 || B < 1) return set_error(RTS_ERR_INVALID, "M and B must be >= 1");
    if (win_frames < 1) return set_error(RTS_ERR_INVALID, "dtw_win_size / hop_size must be >= 1 frame");
    if (hop_frames < 1)
        return set_error(RTS_ERR_INVALID, "dtw_hop_size / hop_size must be >= 1 frame (the reference loops forever at 0)");
    if (win_frames > kWtwMaxW)
        return set_error(RTS_ERR_UNSUPPORTED, "window of %d frames exceeds the supported %d", win_frames, kWtwMaxW);
    rts_wtw *h = (rts_wtw *)calloc(1, sizeof(rts_wtw));
    if (!h) return set_error(RTS_ERR_INVALID, "out of host memory");
    h->ref = chroma_ref_dev;
    if (hipError_t ed = hipGetDevice(&h->device); ed != hipSuccess) {
        free(h);
        return set_error(RTS_ERR_HIP, "hipGetDevice failed: %s", hipGetErrorString(ed));
    }
    h->M = M;
    h->N = 2 * M;  // wtw.py:52
    h->B = B;
    h->W = win_frames;
    h->hopf = hop_frames;
    h->path_cap = (h->N / hop_frames + 2) * (win_frames + hop_frames + 2);
    const int W = win_frames;
    // Windows of more than one strip (64 rows) take the strip-DP path: measured on 64 streams at wtw_live.py's W = 100 /
    // hop = 50 it is twice as fast as the single-workgroup sweep despite its five launches per window.
    // RTS_WTW_BIG_FROM overrides the threshold (tuning and tests; results do not depend on it).
    int big_from = kWtwStripFrom;
    if (const char *e = getenv("RTS_WTW_BIG_FROM")) big_from = atoi(e) < kWtwLdsW ? atoi(e) : kWtwLdsW;
    // Windows of at most 128 frames: the one-launch window kernel (RTS_WTW_WIN=0 or an explicit RTS_WTW_BIG_FROM select the
    // older paths: tests and A/B runs; results are identical).
    // (measured, 64 streams, hop = W / 2: 3.2x faster than the anti-diagonal sweep at W = 20, 3.8x at W = 64, on a par with
    // the strip DP's launch-per-window rounds at W = 100, behind them from ~110 frames on: kWinAutoW)
    bool win = W <= kWinAutoW;
    if (const char *e = getenv("RTS_WTW_WIN")) win = W <= kWinMaxW && atoi(e) != 0;
    if (getenv("RTS_WTW_BIG_FROM")) win = false;
    h->use_win = win ? ((W <= 65 && !getenv("RTS_WIN_FORCE_R2")) ? 1 : 2) : 0;  // interior rows 1 .. W-1: one wave up to 65 frames (RTS_WIN_FORCE_R2: tests)
    const bool big = !win && W > big_from;
    h->use_big = big;
    if (big) {
        int nw, nh, grid;
        // workgroups of the one-strip / two-strip DP kernel this device holds at once (sdp::pick_config, "Residency")
        const size_t pad = sdp::lds_pad();
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_big_dp_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_big_dp_kernel<false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_big_dp_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_big_dp_kernel<true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const int res1 = keep_last_d ? sdp::resident_blocks(wtw_big_dp_kernel<true, 3>, 256, sdp::lds_bytes(1) + pad)
                                     : sdp::resident_blocks(wtw_big_dp_kernel<false, 3>, 256, sdp::lds_bytes(1) + pad);
        const int res2 = keep_last_d ? sdp::resident_blocks(wtw_big_dp_kernel<true, 2>, 384, sdp::lds_bytes(2) + pad)
                                     : sdp::resident_blocks(wtw_big_dp_kernel<false, 2>, 384, sdp::lds_bytes(2) + pad);
        if (res1 < 1 && res2 < 1) {
            free(h);
            return set_error(RTS_ERR_HIP, "the occupancy query reports no resident workgroup for the strip-DP kernel on this device");
        }
        sdp::pick_config(sdp::n_strips(W), B, res1, res2, nw, nh, grid);
        h->big_waves = nw;
        h->big_helpers = nh;
        h->n_rg = (sdp::n_strips(W) + nw - 1) / nw;
        h->big_grid = grid;
        h->smem = sdp::lds_bytes(nw) + pad;
    } else if (win) {
        h->smem = win_lds_bytes(W);
    } else {
        h->smem = sizeof(double) * ((size_t)2 * W * kWF + 2 * W + 3 * W) + sizeof(int32_t) * 4 * W +
                  (W <= kWtwLdsB ? (size_t)W * W : 0) + 64;
    }
    hipError_t e;
    if ((e = hipMalloc((void **)&h->live, sizeof(double) * kWF * (size_t)h->N * B)) != hipSuccess ||
        (e = hipMalloc((void **)&h->appended, sizeof(int32_t) * (size_t)B)) != hipSuccess ||
        (e = hipMalloc((void **)&h->appended_next, sizeof(int32_t) * (size_t)B)) != hipSuccess ||
        (e = hipMalloc((void **)&h->state, sizeof(int32_t) * 8 * (size_t)B)) != hipSuccess ||
        (e = hipMalloc((void **)&h->path, sizeof(int32_t) * 2 * (size_t)h->path_cap * B)) != hipSuccess ||
        (!big && !win && W > kWtwLdsB && (e = hipMalloc((void **)&h->bwork, (size_t)B * W * W)) != hipSuccess) ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_win_kernel<1, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_win_kernel<2, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_win_kernel<1, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_win_kernel<2, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
        (big && (e = hipMalloc((void **)&h->ws_sub, sizeof(int32_t) * 4 * (size_t)W * B)) != hipSuccess) ||
        (big && (e = hipMalloc((void **)&h->ws_scr, sizeof(int32_t) * 2 * sdp::scratch_pairs(W, W) * B)) != hipSuccess) ||
        (big && (e = hipMalloc((void **)&h->ctl, sizeof(int32_t) * 8 * (size_t)B)) != hipSuccess) ||
        (big && (e = hipMalloc((void **)&h->err, 16)) != hipSuccess) ||
        (big && (e = hipMalloc((void **)&h->ticket, sizeof(int32_t) * (size_t)B)) != hipSuccess) ||
        (big && (e = hipMalloc((void **)&h->codes, sizeof(uint32_t) * sdp::codes_words(W, W) * B)) != hipSuccess) ||
        (big && (e = hipMalloc((void **)&h->bnd, sizeof(unsigned long long) * (size_t)sdp::n_strips(W) * W * B)) != hipSuccess) ||
        (big && (e = hipMalloc((void **)&h->entb, sizeof(int32_t) * (size_t)sdp::n_strips(W) * W * B)) != hipSuccess) ||
        (big && (e = hipMalloc((void **)&h->cross, sizeof(int32_t) * (size_t)sdp::n_strips(W) * B)) != hipSuccess) ||
        (big && (e = hipMalloc((void **)&h->yrec, sizeof(double) * sdp::kYRec * (size_t)W * B)) != hipSuccess) ||
        (big && (e = hipMalloc((void **)&h->lens, sizeof(int32_t) * (size_t)sdp::n_strips(W) * B)) != hipSuccess) ||
        (keep_last_d && (e = hipMalloc((void **)&h->dlast, sizeof(double) * (size_t)B * W * W)) != hipSuccess) ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_advance_kernel<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_advance_kernel<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_big_dp_kernel<true, 2>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_big_dp_kernel<false, 2>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_big_dp_kernel<true, 3>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess ||
        (e = hipFuncSetAttribute(reinterpret_cast<const void *>(&wtw_big_dp_kernel<false, 3>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) {
        rts_wtw_destroy(h);
        return set_error(RTS_ERR_HIP, "WTW allocation failed: %s", hipGetErrorString(e));
    }
    int rc = rts_wtw_reset(h, nullptr);
    if (rc != RTS_OK) {
        rts_wtw_destroy(h);
        return rc;
    }
    if (hipError_t es = hipStreamSynchronize(nullptr); es != hipSuccess) {
        rts_wtw_destroy(h);
        return set_error(RTS_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(es));
    }
    *out = h;
    return RTS_OK;
}

int rts_wtw_destroy(rts_wtw *h) {
    if (!h) return RTS_OK;
    if (h->live) (void)hipFree(h->live);
    if (h->appended) (void)hipFree(h->appended);
    if (h->appended_next) (void)hipFree(h->appended_next);
    if (h->state) (void)hipFree(h->state);
    if (h->path) (void)hipFree(h->path);
    if (h->bwork) (void)hipFree(h->bwork);
    if (h->dlast) (void)hipFree(h->dlast);
    if (h->ws_sub) (void)hipFree(h->ws_sub);
    if (h->ws_scr) (void)hipFree(h->ws_scr);
    if (h->ctl) (void)hipFree(h->ctl);
    if (h->err) (void)hipFree(h->err);
    if (h->ticket) (void)hipFree(h->ticket);
    if (h->codes) (void)hipFree(h->codes);
    if (h->bnd) (void)hipFree(h->bnd);
    if (h->entb) (void)hipFree(h->entb);
    if (h->cross) (void)hipFree(h->cross);
    if (h->yrec) (void)hipFree(h->yrec);
    if (h->lens) (void)hipFree(h->lens);
    free(h);
    return RTS_OK;
}

int rts_wtw_reset(rts_wtw *h, void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (h->state) {  // (rts_wtw_create resets before the handle is complete; the device is current there by construction)
        if (int rc = wtw_check_device(h); rc != RTS_OK) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    RTS_HIP(hipMemsetAsync(h->appended, 0, sizeof(int32_t) * (size_t)h->B, s));
    RTS_HIP(hipMemsetAsync(h->state, 0, sizeof(int32_t) * 8 * (size_t)h->B, s));
    // wtw.py:55: chroma_live starts as zeros
    RTS_HIP(hipMemsetAsync(h->live, 0, sizeof(double) * kWF * (size_t)h->N * h->B, s));
    if (h->ctl) RTS_HIP(hipMemsetAsync(h->ctl, 0, sizeof(int32_t) * 8 * (size_t)h->B, s));
    if (h->err) RTS_HIP(hipMemsetAsync(h->err, 0, 16, s));
    if (h->ticket) RTS_HIP(hipMemsetAsync(h->ticket, 0, sizeof(int32_t) * (size_t)h->B, s));
    return RTS_OK;
}

int rts_wtw_push(rts_wtw *h, const void *cols_dev, int cols_dtype, int n_max, const int32_t *n_new_dev, int precheck,
                 void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (n_max < 0) return set_error(RTS_ERR_INVALID, "n_max < 0");
    if (n_max > 0 && !cols_dev) return set_error(RTS_ERR_INVALID, "cols_dev is NULL");
    if (cols_dtype != RTS_F32 && cols_dtype != RTS_F64) return set_error(RTS_ERR_INVALID, "bad cols_dtype %d", cols_dtype);
    if (int rc = wtw_check_device(h); rc != RTS_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (precheck) {
        hipLaunchKernelGGL(wtw_precheck_kernel, dim3((h->B + 63) / 64), dim3(64), 0, s, h->state, h->B, h->M, h->N);
        RTS_HIP(hipGetLastError());
    }
    if (n_max == 0) return RTS_OK;
    hipLaunchKernelGGL(wtw_append_kernel, dim3(h->B, n_max >= 64 ? kWtwAppendSlices : 1), dim3(256), 0, s, h->live, h->appended,
                       h->appended_next, h->state, cols_dev, cols_dtype == RTS_F64, n_new_dev, n_max, n_max, h->B, h->N);
    {
        int32_t *t = h->appended;
        h->appended = h->appended_next;
        h->appended_next = t;
    }
    RTS_HIP(hipGetLastError());
    WtwArgs g;
    g.ref = h->ref;
    g.live = h->live;
    g.appended = h->appended;
    g.state = h->state;
    g.path = h->path;
    g.bwork = h->bwork;
    g.dlast = h->dlast;
    g.ws_sub = h->ws_sub;
    g.ws_scr = h->ws_scr;
    g.ctl = h->ctl;
    g.codes = h->codes;
    g.bnd = h->bnd;
    g.entb = h->entb;
    g.cross = h->cross;
    g.lens = h->lens;
    g.yrec = h->yrec;
    g.err = h->err;
    g.ticket = h->ticket;
    g.n_rg = h->n_rg;
    g.fill_separate = ((size_t)(h->n_rg > 1 ? h->n_rg - 1 : 0) * h->W > (1u << 16)) ? 1 : 0;  // more than 0.5 MB of boundary words
    g.n_strips_wg = h->big_waves;
    g.M = h->M;
    g.N = h->N;
    g.W = h->W;
    g.hopf = h->hopf;
    g.path_cap = h->path_cap;
    if (h->use_big) {
        // one (dp, ctl) round per window the new columns can complete: the first needs at least one column, every
        // further one dtw_hop / hop more (the live pointer advances by exactly that per window, wtw.py:118-128)
        const int rounds = n_max / h->hopf + 1;
        hipLaunchKernelGGL(wtw_big_ctl_kernel, dim3(h->B), dim3(1024), 0, s, g);
        if (g.fill_separate) hipLaunchKernelGGL(wtw_big_fill_kernel, dim3(128, h->B), dim3(256), 0, s, g);
        for (int r = 0; r < rounds; r++) {
            const dim3 grid(h->big_grid, h->B), block(64 * h->big_waves * (1 + h->big_helpers));
            if (h->big_helpers == 2) {
                if (h->dlast)
                    hipLaunchKernelGGL((wtw_big_dp_kernel<true, 2>), grid, block, h->smem, s, g);
                else
                    hipLaunchKernelGGL((wtw_big_dp_kernel<false, 2>), grid, block, h->smem, s, g);
            } else {
                if (h->dlast)
                    hipLaunchKernelGGL((wtw_big_dp_kernel<true, 3>), grid, block, h->smem, s, g);
                else
                    hipLaunchKernelGGL((wtw_big_dp_kernel<false, 3>), grid, block, h->smem, s, g);
            }
            if (sdp::n_strips(h->W) <= sdp::kTailStrips) {
                hipLaunchKernelGGL(wtw_big_tail_ctl_kernel, dim3(h->B), dim3(64 * sdp::n_strips(h->W)),
                                   sdp::tail_lds_bytes(sdp::n_strips(h->W)), s, g);
            } else {
                hipLaunchKernelGGL(wtw_big_hops_kernel, dim3(h->B), dim3(64), 0, s, g);
                hipLaunchKernelGGL((wtw_big_segment_kernel<0>), dim3(sdp::n_strips(h->W), h->B), dim3(64), 0, s, g);
                hipLaunchKernelGGL((wtw_big_segment_kernel<1>), dim3(sdp::n_strips(h->W), h->B), dim3(64), 0, s, g);
                hipLaunchKernelGGL(wtw_big_ctl_kernel, dim3(h->B), dim3(1024), 0, s, g);
            }
            if (g.fill_separate) hipLaunchKernelGGL(wtw_big_fill_kernel, dim3(128, h->B), dim3(256), 0, s, g);
        }
    } else if (h->use_win == 1) {
        if (h->dlast)
            hipLaunchKernelGGL((wtw_win_kernel<1, true>), dim3(h->B), dim3(256), h->smem, s, g);
        else
            hipLaunchKernelGGL((wtw_win_kernel<1, false>), dim3(h->B), dim3(256), h->smem, s, g);
    } else if (h->use_win == 2) {
        if (h->dlast)
            hipLaunchKernelGGL((wtw_win_kernel<2, true>), dim3(h->B), dim3(512), h->smem, s, g);
        else
            hipLaunchKernelGGL((wtw_win_kernel<2, false>), dim3(h->B), dim3(512), h->smem, s, g);
    } else if (h->W > kWtwLdsB) {
        hipLaunchKernelGGL((wtw_advance_kernel<false>), dim3(h->B), dim3(kWtwNT), h->smem, s, g);
    } else {
        hipLaunchKernelGGL((wtw_advance_kernel<true>), dim3(h->B), dim3(kWtwNT), h->smem, s, g);
    }
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

int rts_wtw_read_states(rts_wtw *h, int32_t *states, void *stream) {
    using namespace rts;
    if (!h || !states) return set_error(RTS_ERR_INVALID, "NULL argument");
    if (int rc = wtw_check_device(h); rc != RTS_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    RTS_HIP(hipMemcpyAsync(states, h->state, sizeof(int32_t) * 8 * (size_t)h->B, hipMemcpyDeviceToHost, s));
    RTS_HIP(hipStreamSynchronize(s));
    return RTS_OK;
}

int rts_wtw_read_path(rts_wtw *h, int b, int32_t *pairs, int cap_pairs, int *n, void *stream) {
    using namespace rts;
    if (!h || !n) return set_error(RTS_ERR_INVALID, "NULL argument");
    if (b < 0 || b >= h->B) return set_error(RTS_ERR_INVALID, "stream index %d out of range [0, %d)", b, h->B);
    if (int rc = wtw_check_device(h); rc != RTS_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    int32_t np = 0;
    RTS_HIP(hipMemcpyAsync(&np, h->state + (size_t)b * 8 + 4, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    RTS_HIP(hipStreamSynchronize(s));
    *n = np;
    int m = np < h->path_cap ? np : h->path_cap;
    if (m > cap_pairs) m = cap_pairs;
    if (m > 0 && pairs) {
        RTS_HIP(hipMemcpyAsync(pairs, h->path + (size_t)b * h->path_cap * 2, sizeof(int32_t) * 2 * (size_t)m,
                               hipMemcpyDeviceToHost, s));
        RTS_HIP(hipStreamSynchronize(s));
    }
    return RTS_OK;
}

int rts_wtw_read_last_d(rts_wtw *h, int b, double *d_host, void *stream) {
    using namespace rts;
    if (!h || !d_host) return set_error(RTS_ERR_INVALID, "NULL argument");
    if (b < 0 || b >= h->B) return set_error(RTS_ERR_INVALID, "stream index %d out of range [0, %d)", b, h->B);
    if (!h->dlast) return set_error(RTS_ERR_INVALID, "handle was created with keep_last_d = 0");
    if (int rc = wtw_check_device(h); rc != RTS_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    RTS_HIP(hipMemcpyAsync(d_host, h->dlast + (size_t)b * h->W * h->W, sizeof(double) * (size_t)h->W * h->W,
                           hipMemcpyDeviceToHost, s));
    RTS_HIP(hipStreamSynchronize(s));
    return RTS_OK;
}

#ifdef RTS_WIN_STAMPS
/* Diagnostic build only: reads and clears the per-phase cycle sums of wtw_win_kernel (stream 0, wave 0). */
int rts_wtw_read_win_stamps(long long *out) {
    static const long long zero[8] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rts::g_win_stamps), sizeof(long long) * 8) != hipSuccess) return -3;
    return hipMemcpyToSymbol(HIP_SYMBOL(rts::g_win_stamps), zero, sizeof(zero)) == hipSuccess ? 0 : -3;
}
#endif

int rts_wtw_state_view(rts_wtw *h, int32_t **state_dev) {
    using namespace rts;
    if (!h || !state_dev) return set_error(RTS_ERR_INVALID, "NULL argument");
    *state_dev = h->state;
    return RTS_OK;
}

int rts_wtw_device_views(rts_wtw *h, double **live_chroma_dev, int *live_capacity, double **last_d_dev) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (live_chroma_dev) *live_chroma_dev = h->live;
    if (live_capacity) *live_capacity = h->N;
    if (last_d_dev) *last_d_dev = h->dlast;
    return RTS_OK;
}

}  // extern "C"
