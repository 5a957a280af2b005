// Batched online time warping for gfx950 (MI355X): OnlineTimeWarping / LiveNote / LiveNoteV2.
//
// Reference semantics: /root/reference/otw_eran.py:38-239, livenote.py:37-226,
// livenote_v2.py:43-236 (restated on the CPU, for tests only, in oracle/rtsync_oracle.c).
//
// Design (DESIGN.md "OTW kernel"):
//   * one workgroup (NW waves) per live stream; the whole per-stream state lives in LDS for the
//     duration of a launch: the two live accumulated-cost bands (row t over columns [j-c, j] and
//     column j over rows [t-c, t]) and ring windows of the last W reference / live chroma frames;
//   * a strip (<= c cells of one row or one column) is evaluated in two phases:
//       cost phase   one thread per cell: 12-term cost d, and a = min(up + d, diag + 2d) from
//                    the previous band -> LDS scratch;
//       chain phase  acc_k = min(a_k, acc_{k-1} + d_k) along the strip.  This is a serial
//                    float64 recurrence whose rounding must not change, so it is solved by
//                    *chunked speculative carry propagation*: every lane scans its L = W/64
//                    consecutive cells, then lanes repeatedly re-scan with the neighbour's last
//                    value as carry-in (DPP wave shift) until no carry changes.  Because
//                    x -> fl(x + d) and min are monotone, the fixed point is bit-identical to the
//                    sequential scan; it is reached after (longest carry run / L) + 1 rounds
//                    (2-7 on chroma data at c = 500, against 500 dependent steps);
//   * in a "Both" step the row strip (wave 0) and the column strip (wave 1) run concurrently;
//     only the corner cell depends on both and is finished by one lane;
//   * best_point's two argmins are wave-level reductions (value min, then lowest lane holding
//     it), direction / run-count / path logic runs on one lane.
// All arithmetic is float64 in the oracle's operation order (build with -ffp-contract=off), so
// accumulated costs are bit-identical to the CPU restatement, not merely close.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace rts {

constexpr int kF = 12;
constexpr int kFetch = 8;  // frames fetched per ring refill

struct OtwArgs {
    const void *ref;          // [N][F]
    const void *live;         // [B][live_stride][F]
    const int32_t *live_len;  // [B]
    int32_t *state;           // [B][RTS_STATE_LEN]
    int32_t *path;            // [B][path_cap][2]
    double *bands;            // [B][2][c+1]
    long long live_stride;    // frames between consecutive streams in `live`
    int N, c, max_run_count, variant, cost_kind, mode;
    int path_cap, live_cap;   // live_cap = 2N (otw_eran.py:14)
    int ref_f64, live_f64;
    int clamp_len;            // run mode: never read past live_stride frames
    long long *debug;         // diagnostic builds only (-DRTS_OTW_STAMPS): [B][16] cycle sums
};

template <int W>
struct OtwLds {
    static constexpr int L = W / 64;     // cells per lane in the chain phase
    static constexpr int SWZ = L * 65;   // swizzled band length (one pad slot per row of 64)
    double R[SWZ];   // acc[t][.]  row band
    double C[SWZ];   // acc[.][j]  column band
    double Dr[SWZ];  // row strip: cell costs
    double Ar[SWZ];  // row strip: min over the two out-of-strip predecessors
    double Dc[SWZ];
    double Ac[SWZ];
    double refw[kF][W];   // feature-major ring of reference frames (index y & (W-1))
    double livew[kF][W];  // feature-major ring of live frames      (index x & (W-1))
    double row_last, col_last;
    double rb_min, cb_min;        // np.argmin state of the two bands at the last decide()
    double rfresh_min, cfresh_min;  // argmin of the strip just computed (chain waves -> decide)
    long long cells;
    int t, j, dir, prev, run_count, status, first, n_path, consumed, rows, cols;
    int pending_col, truncated, pend_dir, last_x, last_y;
    int rb_idx, cb_idx, rfresh_idx, cfresh_idx, recomputes;
};

// Band position -> LDS slot.  Cell k lives at row (k mod L), column (k / L mod 64): the chain
// phase (lane owns L consecutive cells) and the cost phase (consecutive threads own consecutive
// cells) both touch distinct banks.
template <int W>
__device__ __forceinline__ int swz(int k) {
    constexpr int L = W / 64;
    const int p = k & (W - 1);
    return (p % L) * 65 + (p / L);
}

__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; }

// np.dot on two strided column views == OpenBLAS ddot with inc != 1 (oracle: orc_dot_strided).
__device__ __forceinline__ double dot_strided12(const double (&x)[kF], const double (&y)[kF]) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int i = 0; i < kF; i += 4) {
        const double m3 = y[i + 2] * x[i + 2];
        const double m4 = y[i + 3] * x[i + 3];
        const double a = fma(y[i], x[i], m3);
        const double b = fma(y[i + 1], x[i + 1], m4);
        t1 = t1 + a;
        t2 = t2 + b;
    }
    return t1 + t2;
}

// np.sqrt(np.sum((a-b)**2)) with numpy's pairwise order for 12 terms (oracle: orc_euclid).
__device__ __forceinline__ double euclid12(const double (&a)[kF], const double (&b)[kF]) {
    double sq[kF];
#pragma unroll
    for (int i = 0; i < kF; i++) {
        const double d = a[i] - b[i];
        sq[i] = d * d;
    }
    double res = ((sq[0] + sq[1]) + (sq[2] + sq[3])) + ((sq[4] + sq[5]) + (sq[6] + sq[7]));
    res = res + sq[8];
    res = res + sq[9];
    res = res + sq[10];
    res = res + sq[11];
    return sqrt(res);
}

__device__ __forceinline__ double cell_cost(const double (&lf)[kF], const double (&rf)[kF], int euclid) {
    return euclid ? euclid12(lf, rf) : (1.0 - dot_strided12(lf, rf));
}

// v_min_f64 without the canonicalising v_max pair hipcc adds around fmin().  Operands are never NaN
// on this path (costs of finite chroma; the +inf sentinel is handled exactly by the instruction).
__device__ __forceinline__ double vmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// lane i receives lane i-1's value; lane 0 receives `lane0`.  The s_nop covers the VALU-write ->
// DPP-read hazard for a source produced by the preceding (asm or compiler) instruction.
__device__ __forceinline__ double wave_shift_up(double v, double lane0) {
    int lo = __double2loint(lane0), hi = __double2hiint(lane0);
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %3 wave_shr:1 row_mask:0xf bank_mask:0xf"
        : "+v"(lo), "+v"(hi)
        : "v"(__double2loint(v)), "v"(__double2hiint(v)));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_bcast(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// Minimum over the 64 lanes, returned in every lane.  Six DPP steps (row_shr 1/2/4/8, then
// row_bcast 15 and 31) leave the total in lane 63; lanes without a DPP source keep their own value.
#define RTS_DPP_MIN_STEP(CTRL)                                                   \
    do {                                                                         \
        int tlo = __double2loint(x), thi = __double2hiint(x);                    \
        asm volatile("s_nop 1\n\t"                                               \
                     "v_mov_b32_dpp %0, %2 " CTRL "\n\t"                         \
                     "v_mov_b32_dpp %1, %3 " CTRL                                \
                     : "+v"(tlo), "+v"(thi)                                      \
                     : "v"(__double2loint(x)), "v"(__double2hiint(x)));          \
        x = vmin(x, __hiloint2double(thi, tlo));                                 \
    } while (0)

__device__ __forceinline__ double wave_min(double x) {
    RTS_DPP_MIN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf");
    RTS_DPP_MIN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf");
    RTS_DPP_MIN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf");
    RTS_DPP_MIN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf");
    RTS_DPP_MIN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf");
    RTS_DPP_MIN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf");
    return wave_bcast(x, 63);
}

// Exact solution of acc_i = min(A_i, acc_{i-1} + D_i), i in [0, n), acc_{-1} = x_in, for the strip
// whose cell i sits at band position k1 + i.  One wave; writes out[swz(k1+i)], returns acc_{n-1}
// (x_in if n == 0) in every lane.  Also returns np.argmin (first minimum) of the new strip
// restricted to band positions >= lo_arg: (fmin, fidx), fidx = 0x7fffffff if that range is empty.
template <int W>
__device__ __forceinline__ double strip_chain(const double *__restrict__ Dv, const double *__restrict__ Av,
                                              double *__restrict__ out, int k1, int n, double x_in, int lane,
                                              int lo_arg, double &fmin_out, int &fidx_out) {
    constexpr int L = W / 64;
    const double inf = INFINITY;
    double A[L], D[L], v[L];
#pragma unroll
    for (int m = 0; m < L; m++) {
        const int i = L * lane + m;
        const bool valid = i < n;
        const int s = swz<W>(k1 + i);
        A[m] = valid ? Av[s] : inf;
        D[m] = valid ? Dv[s] : inf;
    }
    // round 0: every lane scans its own cells; only lane 0 knows its true carry-in
    double p = (lane == 0) ? x_in : inf;
#pragma unroll
    for (int m = 0; m < L; m++) {
        p = p + D[m];
        v[m] = vmin(A[m], p);
        p = v[m];
    }
    // further rounds: carry-in = left neighbour's current last value.  Values only ever decrease
    // and min(A, chain of rounded adds from the carry) is exactly what the serial scan computes
    // once the carry is final, so the fixed point is the serial result.
    // Lanes 0..r are final after round r, so 64 rounds always suffice; the bound also keeps a NaN
    // (NaN != NaN) from spinning forever.
    for (int round = 0; round < 64; round++) {
        double q = wave_shift_up(v[L - 1], x_in);
        // a carry that is already beaten at the lane's first cell can never win further right
        // (monotonicity), so if that holds in every lane the strip is final
        q = q + D[0];
        if (!__any(q < v[0])) break;
        v[0] = vmin(v[0], q);
#pragma unroll
        for (int m = 1; m < L; m++) {
            q = q + D[m];
            v[m] = vmin(v[m], q);
        }
    }
    double lm = inf;
#pragma unroll
    for (int m = 0; m < L; m++) {
        const int i = L * lane + m;
        if (i < n) {
            out[swz<W>(k1 + i)] = v[m];
            if (k1 + i >= lo_arg) lm = vmin(lm, v[m]);
        }
    }
    // first minimum: lowest lane holding the wave minimum, then its first matching cell
    const double g = wave_min(lm);
    int cand = 0x7fffffff;
#pragma unroll
    for (int m = L - 1; m >= 0; m--) {
        const int i = L * lane + m;
        if (i < n && k1 + i >= lo_arg && v[m] == g) cand = k1 + i;
    }
    const unsigned long long mask = __ballot(cand != 0x7fffffff);
    fmin_out = g;
    fidx_out = mask ? __builtin_amdgcn_readlane(cand, (int)__builtin_ctzll(mask)) : 0x7fffffff;
    if (n == 0) return x_in;
    const int li = n - 1;
    double mine = v[0];
#pragma unroll
    for (int m = 1; m < L; m++) mine = ((li % L) == m) ? v[m] : mine;
    return wave_bcast(mine, li / L);
}

// np.argmin over band[lo..hi] (first minimum); (inf, 0x7fffffff) for an empty range.  One wave;
// results uniform.  Only used when an incrementally maintained band minimum has left the window.
template <int W>
__device__ __forceinline__ void band_argmin(const double *__restrict__ band, int lo, int hi, int lane,
                                            double &vmin_out, int &imin) {
    constexpr int L = W / 64;
    double v[L];
    double lm = INFINITY;
#pragma unroll
    for (int m = 0; m < L; m++) {
        const int k = lo + L * lane + m;
        v[m] = (k <= hi) ? band[swz<W>(k)] : (double)INFINITY;
        if (k <= hi) lm = vmin(lm, v[m]);
    }
    const double g = wave_min(lm);
    int cand = 0x7fffffff;
#pragma unroll
    for (int m = L - 1; m >= 0; m--) {
        const int k = lo + L * lane + m;
        if (k <= hi && v[m] == g) cand = k;
    }
    const unsigned long long mask = __ballot(cand != 0x7fffffff);
    vmin_out = g;
    imin = mask ? __builtin_amdgcn_readlane(cand, (int)__builtin_ctzll(mask)) : 0x7fffffff;
}

// In-kernel cycle stamps exist only in the diagnostic build (tools/otw_phase_profile.py); the
// shipped library compiles them away.
#ifdef RTS_OTW_STAMPS
#define RTS_STAMP(slot)                                                   \
    do {                                                                  \
        const long long now_ = (long long)__builtin_amdgcn_s_memtime();   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                               \
        stamp_sum[slot] += now_ - stamp_last;                             \
        stamp_last = now_;                                                \
    } while (0)
#else
#define RTS_STAMP(slot) \
    do {                \
    } while (0)
#endif

template <int W, int NW>
__global__ void __launch_bounds__(64 * NW) otw_advance_kernel(OtwArgs a) {
    constexpr int NT = 64 * NW;
#ifdef RTS_OTW_STAMPS
    long long stamp_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long stamp_last = (long long)__builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    extern __shared__ __align__(16) unsigned char smem_raw[];
    OtwLds<W> &S = *reinterpret_cast<OtwLds<W> *>(smem_raw);

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c = a.c, N = a.N;
    const int euclid = a.cost_kind == RTS_COST_EUCLID;
    const double sentinel = (a.variant == RTS_VARIANT_OTW) ? 1e10 : (double)INFINITY;
    const double inf = INFINITY;
    int32_t *st = a.state + (size_t)b * RTS_STATE_LEN;
    int live_len_raw = a.live_len[b];
    if (a.clamp_len && live_len_raw > (int)a.live_stride) live_len_raw = (int)a.live_stride;
    if (live_len_raw < 0) live_len_raw = 0;
    const int live_len = live_len_raw;
    // LiveNote's set_live applies the run-count update at the bottom of its loop (livenote_v2.py:149-155)
    const bool deferred_update = (a.mode == RTS_MODE_SET_LIVE) && (a.variant != RTS_VARIANT_OTW);

    if (tid == 0) {
        S.t = st[RTS_ST_T];
        S.j = st[RTS_ST_J];
        S.dir = st[RTS_ST_DIRECTION];
        S.prev = st[RTS_ST_PREVIOUS];
        S.run_count = st[RTS_ST_RUN_COUNT];
        S.status = st[RTS_ST_STATUS];
        S.first = st[RTS_ST_FIRST_INSERT];
        S.n_path = st[RTS_ST_N_PATH];
        S.consumed = st[RTS_ST_CONSUMED];
        S.rows = st[RTS_ST_ROW_STRIPS];
        S.cols = st[RTS_ST_COL_STRIPS];
        S.cells = ((long long)(uint32_t)st[RTS_ST_CELLS_HI] << 32) | (uint32_t)st[RTS_ST_CELLS_LO];
        S.truncated = st[RTS_ST_PATH_TRUNCATED];
        S.pending_col = 0;
        S.recomputes = st[RTS_ST_BAND_RECOMPUTES];
        S.pend_dir = st[14];
        S.last_x = -1;
        S.last_y = -1;
        if (S.n_path > 0 && S.n_path <= a.path_cap) {
            const int32_t *pp = a.path + ((size_t)b * a.path_cap + (S.n_path - 1)) * 2;
            S.last_x = pp[0];
            S.last_y = pp[1];
        }
    }
    __syncthreads();
    if (S.status == RTS_STOP_REF_END) return;  // sticky; the reference's callers stop inserting
    if (S.status == RTS_LIVE_OVERFLOW) {       // otw_eran.py:50-55: t keeps counting inserts
        if (tid == 0 && live_len > S.consumed) {
            st[RTS_ST_T] = live_len - 1;
            st[RTS_ST_CONSUMED] = live_len;
        }
        return;
    }
    if (live_len <= S.consumed) return;  // nothing new

    // ---- ring refills (uniform control flow: live_hi / ref_hi are identical in every thread)
    int live_hi = -1, ref_hi = -1;
    auto load_feat = [&](const void *base, int is_f64, long long idx) -> double {
        return is_f64 ? reinterpret_cast<const double *>(base)[idx]
                      : (double)reinterpret_cast<const float *>(base)[idx];
    };
    const long long live_base = (long long)b * a.live_stride * kF;
    auto fill_live = [&](int lo, int hi) {  // frames lo..hi -> ring
        for (int idx = tid; idx < (hi - lo + 1) * kF; idx += NT) {
            const int fr = lo + idx / kF, f = idx % kF;
            S.livew[f][fr & (W - 1)] = load_feat(a.live, a.live_f64, live_base + (long long)fr * kF + f);
        }
    };
    auto fill_ref = [&](int lo, int hi) {
        for (int idx = tid; idx < (hi - lo + 1) * kF; idx += NT) {
            const int fr = lo + idx / kF, f = idx % kF;
            S.refw[f][fr & (W - 1)] = load_feat(a.ref, a.ref_f64, (long long)fr * kF + f);
        }
    };
    auto ensure_live = [&](int need) {
        if (need > live_hi) {
            int hi = live_hi + kFetch;
            if (hi < need) hi = need;
            if (hi > live_len - 1) hi = live_len - 1;
            fill_live(live_hi + 1, hi);
            live_hi = hi;
            __syncthreads();
        }
    };
    auto ensure_ref = [&](int need) {
        if (need > ref_hi) {
            int hi = ref_hi + kFetch;
            if (hi < need) hi = need;
            if (hi > N - 1) hi = N - 1;
            fill_ref(ref_hi + 1, hi);
            ref_hi = hi;
            __syncthreads();
        }
    };

    // ---- decide(): best_point + path + direction (otw_eran.py:153-211, livenote_v2.py:193-236).
    // Called by every lane of wave 0 with the post-strip (t, j).  The two band argmins are kept
    // incrementally: a strip computed this step brings its own argmin from the chain wave; a band
    // that merely slid by one cell keeps its minimum unless that cell left the window (then a full
    // wave reduction over the band recomputes it); the one cell appended at the top index wins only
    // if strictly smaller (np.argmin returns the first minimum).
    auto decide = [&](int t, int j, bool row_fresh, bool col_fresh, bool row_corner, bool col_corner,
                      bool full) {
        const int j1 = (j - c + 1 > 0) ? j - c + 1 : 0;
        const int t1 = (t - c + 1 > 0) ? t - c + 1 : 0;
        double rmin, cmin;
        int ridx, cidx;
        if (full) {
            band_argmin<W>(S.R, j1, j, lane, rmin, ridx);
            band_argmin<W>(S.C, t1, t, lane, cmin, cidx);
        } else {
            // row band: positions [j1, j]; the top cell j is a corner appended by a column strip
            if (row_fresh) {
                rmin = S.rfresh_min;
                ridx = S.rfresh_idx;
            } else {
                rmin = S.rb_min;
                ridx = S.rb_idx;
                if (ridx < j1) {  // uniform: the old minimum slid out of the window
                    band_argmin<W>(S.R, j1, row_corner ? j - 1 : j, lane, rmin, ridx);
                    if (lane == 0) S.recomputes += 1;
                }
            }
            if (row_corner) {
                const double rc = S.R[swz<W>(j)];
                if (rc < rmin) {
                    rmin = rc;
                    ridx = j;
                }
            }
            // column band: positions [t1, t]; the top cell t is appended by a row strip or is the corner
            if (col_fresh) {
                cmin = S.cfresh_min;
                cidx = S.cfresh_idx;
            } else {
                cmin = S.cb_min;
                cidx = S.cb_idx;
                if (cidx < t1) {
                    band_argmin<W>(S.C, t1, col_corner ? t - 1 : t, lane, cmin, cidx);
                    if (lane == 0) S.recomputes += 1;
                }
            }
            if (col_corner) {
                const double cc = S.C[swz<W>(t)];
                if (cc < cmin) {
                    cmin = cc;
                    cidx = t;
                }
            }
        }
        if (lane == 0) {
            S.rb_min = rmin;
            S.rb_idx = ridx;
            S.cb_min = cmin;
            S.cb_idx = cidx;
            int x, y;
            if (rmin < cmin) {
                x = t;
                y = ridx;
            } else {
                x = cidx;
                y = j;
            }
            bool append = true;
            if (a.variant == RTS_VARIANT_LIVENOTE_V2)  // livenote_v2.py:198
                append = (S.n_path == 0) || (x > S.last_x && y >= S.last_y);
            if (append) {
                if (S.n_path < a.path_cap) {
                    int2 *pp = reinterpret_cast<int2 *>(a.path) + ((size_t)b * a.path_cap + S.n_path);
                    *pp = make_int2(x, y);
                } else {
                    S.truncated = 1;
                }
                S.n_path += 1;
                S.last_x = x;
                S.last_y = y;
            }
            int nd;
            if (t < c)
                nd = RTS_DIR_BOTH;
            else if (S.run_count >= a.max_run_count)
                nd = (S.prev == RTS_DIR_ROW) ? RTS_DIR_COLUMN : RTS_DIR_ROW;
            else if (x < t)
                nd = RTS_DIR_COLUMN;
            else if (y < j)
                nd = RTS_DIR_ROW;
            else
                nd = RTS_DIR_BOTH;
            if (deferred_update) {
                S.pend_dir = nd;
            } else {
                S.run_count = (nd == S.prev) ? S.run_count + 1 : 1;
                if (nd != RTS_DIR_BOTH) S.prev = nd;
            }
            S.dir = nd;
            S.pending_col = (nd == RTS_DIR_COLUMN);
            S.t = t;
            S.j = j;
        }
    };
    auto apply_pending = [&]() {  // lane 0 of wave 0 only
        if (deferred_update && S.pend_dir != -2) {
            const int nd = S.pend_dir;
            S.run_count = (nd == S.prev) ? S.run_count + 1 : 1;
            if (nd != RTS_DIR_BOTH) S.prev = nd;
            S.pend_dir = -2;
        }
    };

    // ---- prologue: first frame, or reload of the persisted bands / windows
    if (S.first) {
        ensure_live(0);
        ensure_ref(0);
        if (tid == 0) {
            double lf[kF], rf[kF];
#pragma unroll
            for (int f = 0; f < kF; f++) {
                lf[f] = S.livew[f][0];
                rf[f] = S.refw[f][0];
            }
            const double d = cell_cost(lf, rf, euclid);
            S.R[swz<W>(0)] = d;
            S.C[swz<W>(0)] = d;
            S.first = 0;
            S.consumed = 1;
            S.cells += 1;
            S.t = 0;
            S.j = 0;
            S.pend_dir = -2;
        }
        __syncthreads();
        if (a.mode == RTS_MODE_SET_LIVE) {
            if (wave == 0) decide(0, 0, false, false, false, false, true);
            __syncthreads();
        } else {
            if (tid == 0) {  // both bands hold the single cell (0,0)
                S.rb_min = S.R[swz<W>(0)];
                S.cb_min = S.rb_min;
                S.rb_idx = 0;
                S.cb_idx = 0;
            }
            __syncthreads();
        }
    } else {
        const int t = S.t, j = S.j;
        const double *bb = a.bands + (size_t)b * 2 * (c + 1);
        for (int i = tid; i <= c; i += NT) {
            const int y = j - c + i, x = t - c + i;
            if (y >= 0) S.R[swz<W>(y)] = bb[i];
            if (x >= 0) S.C[swz<W>(x)] = bb[(c + 1) + i];
        }
        const int lo_l = (t - c + 1 > 0) ? t - c + 1 : 0;
        const int lo_r = (j - c + 1 > 0) ? j - c + 1 : 0;
        fill_live(lo_l, t);
        fill_ref(lo_r, j);
        live_hi = t;
        ref_hi = j;
        __syncthreads();
        if (wave == 0) {  // band minima are not persisted: rebuild them from the reloaded bands
            double rmin, cmin;
            int ridx, cidx;
            band_argmin<W>(S.R, (j - c + 1 > 0) ? j - c + 1 : 0, j, lane, rmin, ridx);
            band_argmin<W>(S.C, (t - c + 1 > 0) ? t - c + 1 : 0, t, lane, cmin, cidx);
            if (lane == 0) {
                S.rb_min = rmin;
                S.rb_idx = ridx;
                S.cb_min = cmin;
                S.cb_idx = cidx;
            }
        }
        __syncthreads();
    }

    // ---- step loop: one iteration = one row strip and/or one column strip + one decide()
    for (;;) {
        RTS_STAMP(0);
        const int t0 = S.t, j0 = S.j, dir = S.dir, pending_col = S.pending_col;
        bool do_row, do_col;
        int t = t0;
        if (pending_col) {
            do_row = false;
            do_col = true;
        } else {
            if (t0 + 1 >= live_len) {  // live sequence exhausted
                if (tid == 0 && a.mode == RTS_MODE_SET_LIVE) S.t = t0 + 1;  // otw_eran.py:111-115
                break;
            }
            t = t0 + 1;
            if (t >= a.live_cap) {  // otw_eran.py:53-55
                if (tid == 0) {
                    S.status = RTS_LIVE_OVERFLOW;
                    S.t = live_len - 1;
                    S.consumed = live_len;
                }
                break;
            }
            do_row = true;
            do_col = (dir != RTS_DIR_ROW);
        }
        const int jn = j0 + (do_col ? 1 : 0);
        const bool stop = do_col && (jn >= N);  // otw_eran.py:67-71
        if (do_row) ensure_live(t);
        ensure_ref(stop ? j0 : jn);

        const int k1r = (j0 - c + 1 > 0) ? j0 - c + 1 : 0, nr = j0 - k1r + 1;  // row strip: columns
        const int k1c = (t - c + 1 > 0) ? t - c + 1 : 0, nc = t - k1c + 1;     // column strip: rows
        const bool col_active = do_col && !stop;
        RTS_STAMP(1);

        // -- cost phase
        if (do_row) {
            double lf[kF];
#pragma unroll
            for (int f = 0; f < kF; f++) lf[f] = S.livew[f][t & (W - 1)];
            for (int i = tid; i < nr; i += NT) {
                const int k = k1r + i;
                double rf[kF];
#pragma unroll
                for (int f = 0; f < kF; f++) rf[f] = S.refw[f][k & (W - 1)];
                const double d = cell_cost(lf, rf, euclid);
                double av = S.R[swz<W>(k)] + d;  // (t-1, k): always present
                if (k > 0) av = dmin(av, S.R[swz<W>(k - 1)] + 2 * d);
                S.Dr[swz<W>(k)] = d;
                S.Ar[swz<W>(k)] = av;
            }
        }
        if (col_active) {
            double rf[kF];
#pragma unroll
            for (int f = 0; f < kF; f++) rf[f] = S.refw[f][jn & (W - 1)];
            for (int i = tid; i < nc; i += NT) {
                const int k = k1c + i;
                double lf[kF];
#pragma unroll
                for (int f = 0; f < kF; f++) lf[f] = S.livew[f][k & (W - 1)];
                const double d = cell_cost(lf, rf, euclid);
                // (k, jn-1) is C[k]; for the corner cell of a Both step it is this step's row
                // result, so only the diagonal term is formed here and the rest in the fix-up.
                const bool corner = do_row && (k == t);
                double av = corner ? inf : S.C[swz<W>(k)] + d;
                if (k > 0) av = dmin(av, S.C[swz<W>(k - 1)] + 2 * d);
                S.Dc[swz<W>(k)] = d;
                S.Ac[swz<W>(k)] = av;
            }
        }
        RTS_STAMP(2);
        __syncthreads();
        RTS_STAMP(3);

        // -- chain phase: row strip on wave 0, column strip on wave 1 when both exist
        const int col_wave = (NW > 1 && do_row) ? 1 : 0;
        if (do_row && wave == 0) {
            const double x_in = (k1r > 0) ? sentinel : inf;  // (t, k1r-1) was never evaluated
            const int lo_arg = (jn - c + 1 > 0) ? jn - c + 1 : 0;  // row band's lower end at decide()
            double fm;
            int fi;
            const double last = strip_chain<W>(S.Dr, S.Ar, S.R, k1r, nr, x_in, lane, lo_arg, fm, fi);
            if (lane == 0) {
                if (k1r > 0) S.R[swz<W>(k1r - 1)] = sentinel;
                S.row_last = last;
                S.rfresh_min = fm;
                S.rfresh_idx = fi;
            }
        }
        if (col_active && wave == col_wave) {
            const double x_in = (k1c > 0) ? sentinel : inf;  // (k1c-1, jn) was never evaluated
            const int ncc = nc - (do_row ? 1 : 0);            // corner cell waits for the row strip
            double fm;
            int fi;
            const double last = strip_chain<W>(S.Dc, S.Ac, S.C, k1c, ncc, x_in, lane, k1c, fm, fi);
            if (lane == 0) {
                if (k1c > 0) S.C[swz<W>(k1c - 1)] = sentinel;
                S.col_last = last;
                S.cfresh_min = fm;
                S.cfresh_idx = fi;
            }
        }
        RTS_STAMP(4);
        __syncthreads();
        RTS_STAMP(5);

        // -- corner fix-up + decide (wave 0)
        if (wave == 0) {
            if (lane == 0) {
                if (!stop) apply_pending();  // livenote_v2.py:139-142 breaks before the update
                if (do_row) {
                    S.rows += 1;
                    S.cells += nr;
                    S.consumed = t + 1;
                }
                if (do_row && !col_active) S.C[swz<W>(t)] = S.row_last;  // column j0 gains row t
                if (col_active) {
                    double cl = S.col_last;
                    if (do_row) {
                        const double d = S.Dc[swz<W>(t)];
                        const double av = dmin(S.row_last + d, S.Ac[swz<W>(t)]);
                        cl = dmin(av, cl + d);  // cl = value of (t-1, jn), or the sentinel carry
                        S.C[swz<W>(t)] = cl;
                    }
                    S.R[swz<W>(jn)] = cl;  // row t gains column jn
                    S.cols += 1;
                    S.cells += nc;
                }
                if (stop) {
                    S.status = RTS_STOP_REF_END;
                    S.t = t;
                    S.j = jn;
                    S.pending_col = 0;
                }
            }
            __builtin_amdgcn_wave_barrier();
            RTS_STAMP(6);
            // row band: fresh from this step's row strip, plus the corner a column strip appended;
            // column band: fresh from this step's column strip (its corner cell was finished by the
            // fix-up, outside the chain), or the old band plus the row strip's last cell
            if (!stop) decide(t, jn, do_row, col_active, col_active, do_row, false);
            RTS_STAMP(7);
        }
        __syncthreads();
        RTS_STAMP(8);
        if (S.status != RTS_RUNNING) break;
    }
    __syncthreads();

    // ---- epilogue: persist bands + state
    {
        int t = S.t, j = S.j;
        if (t > a.live_cap - 1) t = a.live_cap - 1;
        if (j > N - 1) j = N - 1;
        double *bb = a.bands + (size_t)b * 2 * (c + 1);
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        const bool have = !S.first;
        for (int i = tid; i <= c; i += NT) {
            const int y = j - c + i, x = t - c + i;
            bb[i] = (have && y >= 0) ? S.R[swz<W>(y)] : qnan;
            bb[(c + 1) + i] = (have && x >= 0 && x <= S.t) ? S.C[swz<W>(x)] : qnan;
        }
    }
#ifdef RTS_OTW_STAMPS
    if (tid == 0 && a.debug)
        for (int i = 0; i < 12; i++) a.debug[(size_t)b * 16 + i] = stamp_sum[i];
#endif
    if (tid == 0) {
        st[RTS_ST_T] = S.t;
        st[RTS_ST_J] = S.j;
        // LiveNote's set_live keeps the direction in a local; self.direction stays "both"
        st[RTS_ST_DIRECTION] = deferred_update ? RTS_DIR_BOTH : S.dir;
        st[RTS_ST_PREVIOUS] = S.prev;
        st[RTS_ST_RUN_COUNT] = S.run_count;
        st[RTS_ST_STATUS] = S.status;
        st[RTS_ST_FIRST_INSERT] = S.first;
        st[RTS_ST_N_PATH] = S.n_path;
        st[RTS_ST_CONSUMED] = S.consumed;
        st[RTS_ST_ROW_STRIPS] = S.rows;
        st[RTS_ST_COL_STRIPS] = S.cols;
        st[RTS_ST_CELLS_LO] = (int32_t)(uint32_t)(S.cells & 0xffffffffLL);
        st[RTS_ST_CELLS_HI] = (int32_t)(uint32_t)((unsigned long long)S.cells >> 32);
        st[RTS_ST_PATH_TRUNCATED] = S.truncated;
        st[14] = S.pend_dir;
        st[RTS_ST_BAND_RECOMPUTES] = S.recomputes;
    }
}

// Fresh per-stream state (otw_eran.py:29-36 / livenote_v2.py:31-37).
__global__ void otw_reset_kernel(int32_t *state, int B, int variant) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int32_t *st = state + (size_t)b * RTS_STATE_LEN;
    for (int i = 0; i < RTS_STATE_LEN; i++) st[i] = 0;
    st[RTS_ST_DIRECTION] = RTS_DIR_BOTH;
    st[RTS_ST_PREVIOUS] = RTS_DIR_NONE;
    st[RTS_ST_RUN_COUNT] = (variant == RTS_VARIANT_OTW) ? 1 : 0;
    st[RTS_ST_STATUS] = RTS_RUNNING;
    st[RTS_ST_FIRST_INSERT] = 1;
    st[14] = -2;
}

// Append one frame per (active) stream to the handle-owned history and bump its length.
__global__ void otw_append_kernel(double *hist, int32_t *hist_len, const void *frames, int frames_f64,
                                  const uint8_t *active, int B, int cap) {
    const int b = blockIdx.x;
    const int f = threadIdx.x;
    if (b >= B || f >= kF) return;
    if (active && !active[b]) return;
    const int n = hist_len[b];
    if (n < cap) {
        const double v = frames_f64 ? reinterpret_cast<const double *>(frames)[b * kF + f]
                                    : (double)reinterpret_cast<const float *>(frames)[b * kF + f];
        hist[((size_t)b * cap + n) * kF + f] = v;
    }
    __syncthreads();
    if (f == 0) hist_len[b] = n + 1;  // may exceed cap: the kernel reports LIVE_OVERFLOW at t >= 2N
}

}  // namespace rts

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct rts_otw {
    const void *ref;
    int ref_dtype, F, N, B, c, max_run_count, variant, cost_kind;
    int W, waves, path_cap, live_cap;
    int32_t *state;     // [B][16]
    int32_t *path;      // [B][path_cap][2]
    double *bands;      // [B][2][c+1]
    double *hist;       // [B][live_cap][F], allocated on first insert
    int32_t *hist_len;  // [B]
    long long *debug;   // diagnostic builds only
};

namespace rts {

template <int W, int NW>
static int launch_advance(const OtwArgs &args, int B, hipStream_t s) {
    const size_t smem = sizeof(OtwLds<W>);
    static bool attr_done = false;  // per instantiation
    if (!attr_done) {
        RTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&otw_advance_kernel<W, NW>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done = true;
    }
    hipLaunchKernelGGL((otw_advance_kernel<W, NW>), dim3(B), dim3(64 * NW), smem, s, args);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

template <int W>
static int launch_w(const OtwArgs &args, int B, int waves, hipStream_t s) {
    switch (waves) {
        case 1: return launch_advance<W, 1>(args, B, s);
        case 2: return launch_advance<W, 2>(args, B, s);
        case 4: return launch_advance<W, 4>(args, B, s);
        case 8: return launch_advance<W, 8>(args, B, s);
    }
    return set_error(RTS_ERR_INVALID, "waves must be 1, 2, 4 or 8 (got %d)", waves);
}

static int launch(const rts_otw *h, const OtwArgs &args, hipStream_t s) {
    switch (h->W) {
        case 64: return launch_w<64>(args, h->B, h->waves, s);
        case 128: return launch_w<128>(args, h->B, h->waves, s);
        case 256: return launch_w<256>(args, h->B, h->waves, s);
        case 512: return launch_w<512>(args, h->B, h->waves, s);
    }
    return set_error(RTS_ERR_UNSUPPORTED, "no kernel for window %d", h->W);
}

static OtwArgs base_args(const rts_otw *h) {
    OtwArgs a;
    memset(&a, 0, sizeof(a));
    a.ref = h->ref;
    a.state = h->state;
    a.path = h->path;
    a.bands = h->bands;
    a.N = h->N;
    a.c = h->c;
    a.max_run_count = h->max_run_count;
    a.variant = h->variant;
    a.cost_kind = h->cost_kind;
    a.path_cap = h->path_cap;
    a.live_cap = h->live_cap;
    a.ref_f64 = h->ref_dtype == RTS_F64;
    a.debug = h->debug;
    return a;
}

}  // namespace rts

extern "C" {

int rts_otw_create(const void *ref_dev, int ref_dtype, int F, int N, int B, int c, int max_run_count,
                   int variant, int cost_kind, rts_otw **out) {
    using namespace rts;
    if (!out) return set_error(RTS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!ref_dev) return set_error(RTS_ERR_INVALID, "ref_dev is NULL");
    if (F != kF) return set_error(RTS_ERR_UNSUPPORTED, "F must be 12 chroma bins (got %d)", F);
    if (N < 1 || B < 1) return set_error(RTS_ERR_INVALID, "N and B must be >= 1 (got N=%d B=%d)", N, B);
    if (ref_dtype != RTS_F32 && ref_dtype != RTS_F64) return set_error(RTS_ERR_INVALID, "bad ref_dtype %d", ref_dtype);
    if (c < 1) return set_error(RTS_ERR_INVALID, "c must be >= 1 (got %d)", c);
    if (c > 500)
        return set_error(RTS_ERR_UNSUPPORTED, "band width c=%d exceeds the 500 cells the LDS-resident kernel holds", c);
    if (max_run_count < 1) return set_error(RTS_ERR_INVALID, "max_run_count must be >= 1");
    if (variant < RTS_VARIANT_OTW || variant > RTS_VARIANT_LIVENOTE_V2)
        return set_error(RTS_ERR_INVALID, "bad variant %d", variant);
    if (cost_kind != RTS_COST_DOT && cost_kind != RTS_COST_EUCLID)
        return set_error(RTS_ERR_INVALID, "bad cost_kind %d", cost_kind);
    if ((long long)N * 3 + 8 > 0x3fffffffLL) return set_error(RTS_ERR_INVALID, "N too large");

    rts_otw *h = (rts_otw *)calloc(1, sizeof(rts_otw));
    if (!h) return set_error(RTS_ERR_INVALID, "out of host memory");
    h->ref = ref_dev;
    h->ref_dtype = ref_dtype;
    h->F = F;
    h->N = N;
    h->B = B;
    h->c = c;
    h->max_run_count = max_run_count;
    h->variant = variant;
    h->cost_kind = cost_kind;
    h->W = 64;
    while (h->W < c + 12) h->W *= 2;
    h->waves = 4;
    h->live_cap = 2 * N;
    h->path_cap = 3 * N + 8;  // one point per decide(); decides <= row strips + column strips <= 2N + N
    hipError_t e;
    if ((e = hipMalloc((void **)&h->state, sizeof(int32_t) * RTS_STATE_LEN * (size_t)B)) != hipSuccess ||
        (e = hipMalloc((void **)&h->path, sizeof(int32_t) * 2 * (size_t)h->path_cap * B)) != hipSuccess ||
        (e = hipMalloc((void **)&h->bands, sizeof(double) * 2 * (size_t)(c + 1) * B)) != hipSuccess ||
        (e = hipMalloc((void **)&h->hist_len, sizeof(int32_t) * (size_t)B)) != hipSuccess) {
        rts_otw_destroy(h);
        return set_error(RTS_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    int rc = rts_otw_reset(h, nullptr);
    if (rc != RTS_OK) {
        rts_otw_destroy(h);
        return rc;
    }
    RTS_HIP(hipStreamSynchronize(nullptr));
    *out = h;
    return RTS_OK;
}

int rts_otw_destroy(rts_otw *h) {
    if (!h) return RTS_OK;
    if (h->state) (void)hipFree(h->state);
    if (h->path) (void)hipFree(h->path);
    if (h->bands) (void)hipFree(h->bands);
    if (h->hist) (void)hipFree(h->hist);
    if (h->hist_len) (void)hipFree(h->hist_len);
    free(h);
    return RTS_OK;
}

int rts_otw_reset(rts_otw *h, void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(otw_reset_kernel, dim3((h->B + 63) / 64), dim3(64), 0, s, h->state, h->B, h->variant);
    RTS_HIP(hipGetLastError());
    RTS_HIP(hipMemsetAsync(h->hist_len, 0, sizeof(int32_t) * (size_t)h->B, s));
    return RTS_OK;
}

int rts_otw_set_waves(rts_otw *h, int waves) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (waves != 1 && waves != 2 && waves != 4 && waves != 8)
        return set_error(RTS_ERR_INVALID, "waves must be 1, 2, 4 or 8 (got %d)", waves);
    h->waves = waves;
    return RTS_OK;
}

const char *rts_otw_kernel_name(const rts_otw *) { return "otw_advance_kernel"; }

int rts_otw_run(rts_otw *h, const void *live_dev, int live_dtype, int T_max, const int32_t *live_len_dev,
                int mode, void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (!live_dev || !live_len_dev) return set_error(RTS_ERR_INVALID, "live_dev / live_len_dev is NULL");
    if (live_dtype != RTS_F32 && live_dtype != RTS_F64) return set_error(RTS_ERR_INVALID, "bad live_dtype %d", live_dtype);
    if (T_max < 0) return set_error(RTS_ERR_INVALID, "T_max < 0");
    if (mode != RTS_MODE_INSERT_LOOP && mode != RTS_MODE_SET_LIVE) return set_error(RTS_ERR_INVALID, "bad mode %d", mode);
    hipStream_t s = (hipStream_t)stream;
    int rc = rts_otw_reset(h, stream);
    if (rc != RTS_OK) return rc;
    OtwArgs a = base_args(h);
    a.live = live_dev;
    a.live_len = live_len_dev;
    a.live_stride = T_max;
    a.live_f64 = live_dtype == RTS_F64;
    a.mode = mode;
    a.clamp_len = 1;
    return launch(h, a, s);
}

int rts_otw_insert(rts_otw *h, const void *frames_dev, int frames_dtype, const uint8_t *active_dev, void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (!frames_dev) return set_error(RTS_ERR_INVALID, "frames_dev is NULL");
    if (frames_dtype != RTS_F32 && frames_dtype != RTS_F64) return set_error(RTS_ERR_INVALID, "bad frames_dtype %d", frames_dtype);
    hipStream_t s = (hipStream_t)stream;
    if (!h->hist) {
        RTS_HIP(hipMalloc((void **)&h->hist, sizeof(double) * kF * (size_t)h->live_cap * h->B));
    }
    hipLaunchKernelGGL(otw_append_kernel, dim3(h->B), dim3(64), 0, s, h->hist, h->hist_len, frames_dev,
                       frames_dtype == RTS_F64, active_dev, h->B, h->live_cap);
    RTS_HIP(hipGetLastError());
    OtwArgs a = base_args(h);
    a.live = h->hist;
    a.live_len = h->hist_len;
    a.live_stride = h->live_cap;
    a.live_f64 = 1;
    a.mode = RTS_MODE_INSERT_LOOP;
    return launch(h, a, s);
}

int rts_otw_read_states(rts_otw *h, int32_t *states, void *stream) {
    using namespace rts;
    if (!h || !states) return set_error(RTS_ERR_INVALID, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    RTS_HIP(hipMemcpyAsync(states, h->state, sizeof(int32_t) * RTS_STATE_LEN * (size_t)h->B, hipMemcpyDeviceToHost, s));
    RTS_HIP(hipStreamSynchronize(s));
    for (int b = 0; b < h->B; b++) states[b * RTS_STATE_LEN + 14] = 0;  // slot 14 is kernel-private
    return RTS_OK;
}

int rts_otw_read_state(rts_otw *h, int b, int32_t *state, void *stream) {
    using namespace rts;
    if (!h || !state) return set_error(RTS_ERR_INVALID, "NULL argument");
    if (b < 0 || b >= h->B) return set_error(RTS_ERR_INVALID, "stream index %d out of range [0, %d)", b, h->B);
    hipStream_t s = (hipStream_t)stream;
    RTS_HIP(hipMemcpyAsync(state, h->state + (size_t)b * RTS_STATE_LEN, sizeof(int32_t) * RTS_STATE_LEN,
                           hipMemcpyDeviceToHost, s));
    RTS_HIP(hipStreamSynchronize(s));
    state[14] = 0;
    return RTS_OK;
}

int rts_otw_read_path(rts_otw *h, int b, int32_t *pairs, int cap_pairs, int *n, void *stream) {
    using namespace rts;
    if (!h || !n) return set_error(RTS_ERR_INVALID, "NULL argument");
    if (b < 0 || b >= h->B) return set_error(RTS_ERR_INVALID, "stream index %d out of range [0, %d)", b, h->B);
    hipStream_t s = (hipStream_t)stream;
    int32_t np = 0;
    RTS_HIP(hipMemcpyAsync(&np, h->state + (size_t)b * RTS_STATE_LEN + RTS_ST_N_PATH, sizeof(int32_t),
                           hipMemcpyDeviceToHost, s));
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

int rts_otw_read_bands(rts_otw *h, int b, double *row_band, double *col_band, void *stream) {
    using namespace rts;
    if (!h || !row_band || !col_band) return set_error(RTS_ERR_INVALID, "NULL argument");
    if (b < 0 || b >= h->B) return set_error(RTS_ERR_INVALID, "stream index %d out of range [0, %d)", b, h->B);
    hipStream_t s = (hipStream_t)stream;
    const double *bb = h->bands + (size_t)b * 2 * (h->c + 1);
    RTS_HIP(hipMemcpyAsync(row_band, bb, sizeof(double) * (h->c + 1), hipMemcpyDeviceToHost, s));
    RTS_HIP(hipMemcpyAsync(col_band, bb + (h->c + 1), sizeof(double) * (h->c + 1), hipMemcpyDeviceToHost, s));
    RTS_HIP(hipStreamSynchronize(s));
    return RTS_OK;
}

#ifdef RTS_OTW_STAMPS
/* Diagnostic build only: caller-provided [B][16] int64 device buffer receiving per-phase cycle sums. */
int rts_otw_set_debug(rts_otw *h, long long *debug_dev) {
    if (!h) return rts::set_error(RTS_ERR_INVALID, "handle is NULL");
    h->debug = debug_dev;
    return RTS_OK;
}
#endif

int rts_otw_device_views(rts_otw *h, int32_t **path_dev, int *path_cap, int32_t **state_dev) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (path_dev) *path_dev = h->path;
    if (path_cap) *path_cap = h->path_cap;
    if (state_dev) *state_dev = h->state;
    return RTS_OK;
}

}  // extern "C"
