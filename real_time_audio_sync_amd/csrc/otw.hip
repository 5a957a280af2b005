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
    double R[SWZ];      // acc[t][.]  row band
    double C[SWZ];      // acc[.][j]  column band
    double Dr[2][SWZ];  // row strip cell costs: [buf] = this step's, [buf^1] = being pre-computed for the next
    double Dc[2][SWZ];
    double Ar[SWZ];     // row strip: min over the two out-of-strip predecessors
    double Ac[SWZ];
    double refw[kF][W];   // feature-major ring of reference frames (index y & (W-1))
    double livew[kF][W];  // feature-major ring of live frames      (index x & (W-1))
    double col_last, cfresh_min;  // column chain wave -> wave 0
    int cfresh_idx;
    int plan_t, plan_j0, plan_flags;  // wave 0 -> everyone: the next step
    int t, j;                         // final position, published at exit for the epilogue
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
    // Waves >= HW0 pre-compute the next step's cell costs while waves 0/1 run the chains.  With fewer
    // than 4 waves there are no spare ones and every wave does its share after its chain.
    constexpr int HW0 = (NW >= 4) ? 2 : 0;
    constexpr int NHELP = 64 * (NW - HW0);
    constexpr int kPlanRow = 1, kPlanCol = 2, kPlanStop = 4, kPlanExit = 8;
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

    // ---- control state.  Every thread loads it (uniform), but only wave 0 keeps it current: it lives
    // in wave 0's registers for the whole launch and reaches the other waves as a 3-word plan in LDS.
    int t = st[RTS_ST_T], j = st[RTS_ST_J], dir = st[RTS_ST_DIRECTION], prev = st[RTS_ST_PREVIOUS];
    int run_count = st[RTS_ST_RUN_COUNT], status = st[RTS_ST_STATUS], first = st[RTS_ST_FIRST_INSERT];
    int n_path = st[RTS_ST_N_PATH], consumed = st[RTS_ST_CONSUMED], rows = st[RTS_ST_ROW_STRIPS];
    int cols = st[RTS_ST_COL_STRIPS], truncated = st[RTS_ST_PATH_TRUNCATED], pend_dir = st[14];
    int recomputes = st[RTS_ST_BAND_RECOMPUTES];
    long long cells = ((long long)(uint32_t)st[RTS_ST_CELLS_HI] << 32) | (uint32_t)st[RTS_ST_CELLS_LO];
    int pending_col = 0, last_x = -1, last_y = -1;
    if (n_path > 0 && n_path <= a.path_cap) {
        const int32_t *pp = a.path + ((size_t)b * a.path_cap + (n_path - 1)) * 2;
        last_x = pp[0];
        last_y = pp[1];
    }
    double rb_min = inf, cb_min = inf;
    int rb_idx = 0, cb_idx = 0;

    if (status == RTS_STOP_REF_END) return;  // sticky; the reference's callers stop inserting
    if (status == RTS_LIVE_OVERFLOW) {       // otw_eran.py:50-55: t keeps counting inserts
        if (tid == 0 && live_len > consumed) {
            st[RTS_ST_T] = live_len - 1;
            st[RTS_ST_CONSUMED] = live_len;
        }
        return;
    }
    if (live_len <= consumed) return;  // nothing new

    // ---- feature rings
    auto load_feat = [&](const void *base, int is_f64, long long idx) -> double {
        return is_f64 ? reinterpret_cast<const double *>(base)[idx]
                      : (double)reinterpret_cast<const float *>(base)[idx];
    };
    const long long live_base = (long long)b * a.live_stride * kF;
    auto fill_live = [&](int lo, int hi) {  // synchronous, all threads (prologue only)
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
    // Asynchronous refills (wave 0): the next kFetch frames sit in two registers per lane, loaded a
    // refill period (~8 steps) before they are written into the ring.
    int live_hi = -1, ref_hi = -1;
    double pfl0 = 0.0, pfl1 = 0.0, pfr0 = 0.0, pfr1 = 0.0;
    auto prefetch_live = [&]() {  // frames live_hi+1 .. live_hi+kFetch
        const int i0 = lane, i1 = lane + 64;
        const int f0 = live_hi + 1 + i0 / kF, f1 = live_hi + 1 + i1 / kF;
        if (f0 < live_len) pfl0 = load_feat(a.live, a.live_f64, live_base + (long long)f0 * kF + i0 % kF);
        if (i1 < kFetch * kF && f1 < live_len)
            pfl1 = load_feat(a.live, a.live_f64, live_base + (long long)f1 * kF + i1 % kF);
    };
    auto prefetch_ref = [&]() {
        const int i0 = lane, i1 = lane + 64;
        const int f0 = ref_hi + 1 + i0 / kF, f1 = ref_hi + 1 + i1 / kF;
        if (f0 < N) pfr0 = load_feat(a.ref, a.ref_f64, (long long)f0 * kF + i0 % kF);
        if (i1 < kFetch * kF && f1 < N) pfr1 = load_feat(a.ref, a.ref_f64, (long long)f1 * kF + i1 % kF);
    };
    auto commit_live = [&]() {  // registers -> ring, then start fetching the following block
        const int i0 = lane, i1 = lane + 64;
        const int f0 = live_hi + 1 + i0 / kF, f1 = live_hi + 1 + i1 / kF;
        if (f0 < live_len) S.livew[i0 % kF][f0 & (W - 1)] = pfl0;
        if (i1 < kFetch * kF && f1 < live_len) S.livew[i1 % kF][f1 & (W - 1)] = pfl1;
        live_hi = (live_hi + kFetch < live_len - 1) ? live_hi + kFetch : live_len - 1;
        prefetch_live();
    };
    auto commit_ref = [&]() {
        const int i0 = lane, i1 = lane + 64;
        const int f0 = ref_hi + 1 + i0 / kF, f1 = ref_hi + 1 + i1 / kF;
        if (f0 < N) S.refw[i0 % kF][f0 & (W - 1)] = pfr0;
        if (i1 < kFetch * kF && f1 < N) S.refw[i1 % kF][f1 & (W - 1)] = pfr1;
        ref_hi = (ref_hi + kFetch < N - 1) ? ref_hi + kFetch : N - 1;
        prefetch_ref();
    };

    // ---- speculative costs for the step after the one that leaves the state at (t_now, j_now):
    // row t_now+1 over columns [j_now-c+1, j_now] and column j_now+1 over rows [t_now-c+1, t_now+1]
    // cover every possible next step (Row, Both or Column).  hidx/hn: index and count of the
    // threads sharing the work.
    auto precompute = [&](int t_now, int j_now, double *Drn, double *Dcn, int hidx, int hn) {
        const int tn = t_now + 1, jn1 = j_now + 1;
        const bool row_ok = (tn < live_len) && (tn < a.live_cap);
        const bool col_ok = jn1 < N;
        const int k1 = (j_now - c + 1 > 0) ? j_now - c + 1 : 0;
        const int nrow = row_ok ? j_now - k1 + 1 : 0;
        const int r1 = (t_now - c + 1 > 0) ? t_now - c + 1 : 0;
        const int rtop = row_ok ? tn : t_now;
        const int ncol = col_ok ? rtop - r1 + 1 : 0;
        if (nrow > 0) {
            double lf[kF];
#pragma unroll
            for (int f = 0; f < kF; f++) lf[f] = S.livew[f][tn & (W - 1)];
            for (int i = hidx; i < nrow; i += hn) {
                const int k = k1 + i;
                double rf[kF];
#pragma unroll
                for (int f = 0; f < kF; f++) rf[f] = S.refw[f][k & (W - 1)];
                Drn[swz<W>(k)] = cell_cost(lf, rf, euclid);
            }
        }
        if (ncol > 0) {
            double rf[kF];
#pragma unroll
            for (int f = 0; f < kF; f++) rf[f] = S.refw[f][jn1 & (W - 1)];
            for (int i = hidx; i < ncol; i += hn) {
                const int r = r1 + i;
                double lf[kF];
#pragma unroll
                for (int f = 0; f < kF; f++) lf[f] = S.livew[f][r & (W - 1)];
                Dcn[swz<W>(r)] = cell_cost(lf, rf, euclid);
            }
        }
    };

    // ---- decide(): best_point + path + direction (otw_eran.py:153-211, livenote_v2.py:193-236).
    // Wave 0, all lanes, register state.  The two band argmins are kept incrementally: a strip
    // computed this step brings its own argmin from its chain; a band that merely slid by one cell
    // keeps its minimum unless that cell left the window (then a full wave reduction recomputes
    // it); the one cell appended at the top index wins only if strictly smaller (np.argmin returns
    // the first minimum).
    auto decide = [&](int tt, int jj, bool row_fresh, double rf_min, int rf_idx, bool col_fresh, double cf_min,
                      int cf_idx, bool row_corner, double rc, bool col_corner, double cc, bool full) {
        const int j1 = (jj - c + 1 > 0) ? jj - c + 1 : 0;
        const int t1 = (tt - c + 1 > 0) ? tt - c + 1 : 0;
        double rmin, cmin;
        int ridx, cidx;
        if (full) {
            band_argmin<W>(S.R, j1, jj, lane, rmin, ridx);
            band_argmin<W>(S.C, t1, tt, lane, cmin, cidx);
        } else {
            if (row_fresh) {
                rmin = rf_min;
                ridx = rf_idx;
            } else {
                rmin = rb_min;
                ridx = rb_idx;
                if (ridx < j1) {  // the old minimum slid out of the window
                    band_argmin<W>(S.R, j1, row_corner ? jj - 1 : jj, lane, rmin, ridx);
                    recomputes += 1;
                }
            }
            if (row_corner && rc < rmin) {
                rmin = rc;
                ridx = jj;
            }
            if (col_fresh) {
                cmin = cf_min;
                cidx = cf_idx;
            } else {
                cmin = cb_min;
                cidx = cb_idx;
                if (cidx < t1) {
                    band_argmin<W>(S.C, t1, col_corner ? tt - 1 : tt, lane, cmin, cidx);
                    recomputes += 1;
                }
            }
            if (col_corner && cc < cmin) {
                cmin = cc;
                cidx = tt;
            }
        }
        rb_min = rmin;
        rb_idx = ridx;
        cb_min = cmin;
        cb_idx = cidx;
        int x, y;
        if (rmin < cmin) {
            x = tt;
            y = ridx;
        } else {
            x = cidx;
            y = jj;
        }
        bool append = true;
        if (a.variant == RTS_VARIANT_LIVENOTE_V2)  // livenote_v2.py:198
            append = (n_path == 0) || (x > last_x && y >= last_y);
        if (append) {
            if (n_path < a.path_cap) {
                if (lane == 0) {
                    int2 *pp = reinterpret_cast<int2 *>(a.path) + ((size_t)b * a.path_cap + n_path);
                    *pp = make_int2(x, y);
                }
            } else {
                truncated = 1;
            }
            n_path += 1;
            last_x = x;
            last_y = y;
        }
        int nd;
        if (tt < c)
            nd = RTS_DIR_BOTH;
        else if (run_count >= a.max_run_count)
            nd = (prev == RTS_DIR_ROW) ? RTS_DIR_COLUMN : RTS_DIR_ROW;
        else if (x < tt)
            nd = RTS_DIR_COLUMN;
        else if (y < jj)
            nd = RTS_DIR_ROW;
        else
            nd = RTS_DIR_BOTH;
        if (deferred_update) {
            pend_dir = nd;
        } else {
            run_count = (nd == prev) ? run_count + 1 : 1;
            if (nd != RTS_DIR_BOTH) prev = nd;
        }
        dir = nd;
        pending_col = (nd == RTS_DIR_COLUMN);
        t = tt;
        j = jj;
    };

    // ---- plan for the next step from the current register state (wave 0); lane 0 publishes it
    auto make_plan = [&]() {
        int flags = 0, pt = t;
        if (status != RTS_RUNNING) {
            flags = kPlanExit;
        } else if (pending_col) {
            flags = kPlanCol;
        } else if (t + 1 >= live_len) {  // live sequence exhausted
            if (a.mode == RTS_MODE_SET_LIVE) t = t + 1;  // otw_eran.py:111-115
            flags = kPlanExit;
        } else if (t + 1 >= a.live_cap) {  // otw_eran.py:53-55
            status = RTS_LIVE_OVERFLOW;
            t = live_len - 1;
            consumed = live_len;
            flags = kPlanExit;
        } else {
            pt = t + 1;
            flags = kPlanRow | ((dir != RTS_DIR_ROW) ? kPlanCol : 0);
        }
        if ((flags & kPlanCol) && j + 1 >= N) flags |= kPlanStop;  // otw_eran.py:67-71
        if (lane == 0) {
            S.plan_t = pt;
            S.plan_j0 = j;
            S.plan_flags = flags;
            if (flags & kPlanExit) {
                S.t = t;
                S.j = j;
            }
        }
        // keep the rings one frame ahead of what this step's cost pre-computation will read
        if (!(flags & kPlanExit)) {
            const int jn_p = j + ((flags & kPlanCol) ? 1 : 0);
            const int need_l = (pt + 1 < live_len - 1) ? pt + 1 : live_len - 1;
            const int need_r = (jn_p + 1 < N - 1) ? jn_p + 1 : N - 1;
            if (need_l > live_hi) commit_live();
            if (need_r > ref_hi) commit_ref();
        }
    };

    // ---- prologue: first frame, or reload of the persisted bands / windows
    {
        const int lo_l = (t - c + 1 > 0) ? t - c + 1 : 0;
        const int lo_r = (j - c + 1 > 0) ? j - c + 1 : 0;
        live_hi = (t + 2 < live_len - 1) ? t + 2 : live_len - 1;
        ref_hi = (j + 2 < N - 1) ? j + 2 : N - 1;
        fill_live(lo_l, live_hi);
        fill_ref(lo_r, ref_hi);
        if (!first) {
            const double *bb = a.bands + (size_t)b * 2 * (c + 1);
            for (int i = tid; i <= c; i += NT) {
                const int y = j - c + i, x = t - c + i;
                if (y >= 0) S.R[swz<W>(y)] = bb[i];
                if (x >= 0) S.C[swz<W>(x)] = bb[(c + 1) + i];
            }
        }
        __syncthreads();
        if (wave == 0) {
            prefetch_live();
            prefetch_ref();
        }
    }
    if (first) {
        if (wave == 0) {
            double lf[kF], rf[kF];
#pragma unroll
            for (int f = 0; f < kF; f++) {
                lf[f] = S.livew[f][0];
                rf[f] = S.refw[f][0];
            }
            const double d = cell_cost(lf, rf, euclid);
            if (lane == 0) {
                S.R[swz<W>(0)] = d;
                S.C[swz<W>(0)] = d;
            }
            first = 0;
            consumed = 1;
            cells += 1;
            t = 0;
            j = 0;
            pend_dir = -2;
            rb_min = d;
            cb_min = d;
            rb_idx = 0;
            cb_idx = 0;
            __builtin_amdgcn_wave_barrier();
            if (a.mode == RTS_MODE_SET_LIVE) decide(0, 0, false, 0.0, 0, false, 0.0, 0, false, 0.0, false, 0.0, true);
        }
    } else if (wave == 0) {  // band minima are not persisted: rebuild them from the reloaded bands
        band_argmin<W>(S.R, (j - c + 1 > 0) ? j - c + 1 : 0, j, lane, rb_min, rb_idx);
        band_argmin<W>(S.C, (t - c + 1 > 0) ? t - c + 1 : 0, t, lane, cb_min, cb_idx);
    }
    if (wave == 0) {
        if (lane == 0) {
            S.t = t;
            S.j = j;
        }
    }
    __syncthreads();
    // prime the cost buffers for the first step (all threads), then publish its plan
    int buf = 0;
    precompute(S.t, S.j, S.Dr[0], S.Dc[0], tid, NT);
    if (wave == 0) make_plan();
    __syncthreads();

    // ---- step loop: one iteration = one row strip and/or one column strip + one decide()
    for (;;) {
        RTS_STAMP(0);
        const int pt = S.plan_t, j0 = S.plan_j0, pflags = S.plan_flags;
        if (pflags & kPlanExit) break;
        const bool do_row = (pflags & kPlanRow) != 0, do_col = (pflags & kPlanCol) != 0;
        const bool stop = (pflags & kPlanStop) != 0;
        const int jn = j0 + (do_col ? 1 : 0);
        const int k1r = (j0 - c + 1 > 0) ? j0 - c + 1 : 0, nr = j0 - k1r + 1;  // row strip: columns
        const int k1c = (pt - c + 1 > 0) ? pt - c + 1 : 0, nc = pt - k1c + 1;  // column strip: rows
        const bool col_active = do_col && !stop;
        double *Dr = S.Dr[buf], *Dc = S.Dc[buf];

        // -- predecessor phase: a = min over the two out-of-strip predecessors, costs are ready
        if (do_row) {
            for (int i = tid; i < nr; i += NT) {
                const int k = k1r + i;
                const double d = Dr[swz<W>(k)];
                double av = S.R[swz<W>(k)] + d;  // (t-1, k): always present
                if (k > 0) av = vmin(av, S.R[swz<W>(k - 1)] + 2 * d);
                S.Ar[swz<W>(k)] = av;
            }
        }
        if (col_active) {
            for (int i = tid; i < nc; i += NT) {
                const int k = k1c + i;
                const double d = Dc[swz<W>(k)];
                // (k, jn-1) is C[k]; for the corner cell of a Both step it is this step's row
                // result, so only the diagonal term is formed here and the rest in the fix-up.
                const bool corner = do_row && (k == pt);
                double av = corner ? inf : S.C[swz<W>(k)] + d;
                if (k > 0) av = vmin(av, S.C[swz<W>(k - 1)] + 2 * d);
                S.Ac[swz<W>(k)] = av;
            }
        }
        RTS_STAMP(1);
        __syncthreads();
        RTS_STAMP(2);

        // -- chain phase: row strip on wave 0, column strip on wave 1; spare waves pre-compute the
        //    next step's costs meanwhile
        const int col_wave = (NW > 1 && do_row) ? 1 : 0;
        double row_last = 0.0, rf_min = inf;
        int rf_idx = 0x7fffffff;
        if (do_row && wave == 0) {
            const double x_in = (k1r > 0) ? sentinel : inf;  // (t, k1r-1) was never evaluated
            const int lo_arg = (jn - c + 1 > 0) ? jn - c + 1 : 0;  // row band's lower end at decide()
            row_last = strip_chain<W>(Dr, S.Ar, S.R, k1r, nr, x_in, lane, lo_arg, rf_min, rf_idx);
            if (lane == 0 && k1r > 0) S.R[swz<W>(k1r - 1)] = sentinel;
        }
        if (col_active && wave == col_wave) {
            const double x_in = (k1c > 0) ? sentinel : inf;  // (k1c-1, jn) was never evaluated
            const int ncc = nc - (do_row ? 1 : 0);            // corner cell waits for the row strip
            double fm;
            int fi;
            const double last = strip_chain<W>(Dc, S.Ac, S.C, k1c, ncc, x_in, lane, k1c, fm, fi);
            if (lane == 0) {
                if (k1c > 0) S.C[swz<W>(k1c - 1)] = sentinel;
                S.col_last = last;
                S.cfresh_min = fm;
                S.cfresh_idx = fi;
            }
        }
        RTS_STAMP(3);
        if (!stop && wave >= HW0) precompute(pt, jn, S.Dr[buf ^ 1], S.Dc[buf ^ 1], tid - 64 * HW0, NHELP);
        RTS_STAMP(4);
        __syncthreads();
        RTS_STAMP(5);

        // -- corner fix-up, decide, next plan (wave 0, register state)
        if (wave == 0) {
            if (!stop && deferred_update && pend_dir != -2) {  // livenote_v2.py:149-155
                run_count = (pend_dir == prev) ? run_count + 1 : 1;
                if (pend_dir != RTS_DIR_BOTH) prev = pend_dir;
                pend_dir = -2;
            }
            if (do_row) {
                rows += 1;
                cells += nr;
                consumed = pt + 1;
            }
            double cl = 0.0, cf_min = inf;
            int cf_idx = 0x7fffffff;
            if (do_row && !col_active && lane == 0) S.C[swz<W>(pt)] = row_last;  // column j0 gains row t
            if (col_active) {
                cl = S.col_last;
                cf_min = S.cfresh_min;
                cf_idx = S.cfresh_idx;
                if (do_row) {
                    const double d = Dc[swz<W>(pt)];
                    const double av = vmin(row_last + d, S.Ac[swz<W>(pt)]);
                    cl = vmin(av, cl + d);  // cl was the value of (t-1, jn), or the sentinel carry
                    if (lane == 0) S.C[swz<W>(pt)] = cl;
                }
                if (lane == 0) S.R[swz<W>(jn)] = cl;  // row t gains column jn
                cols += 1;
                cells += nc;
            }
            RTS_STAMP(6);
            if (stop) {
                status = RTS_STOP_REF_END;
                t = pt;
                j = jn;
                pending_col = 0;
            } else {
                // row band: fresh from this step's row strip, plus the corner a column strip appended;
                // column band: fresh from this step's column strip (its corner cell is outside the
                // chain), or the old band plus the row strip's last cell
                decide(pt, jn, do_row, rf_min, rf_idx, col_active, cf_min, cf_idx, col_active, cl, do_row,
                       col_active ? cl : row_last, false);
            }
            RTS_STAMP(7);
            make_plan();
        }
        RTS_STAMP(8);
        __syncthreads();
        buf ^= 1;
    }
    __syncthreads();

    // ---- epilogue: persist bands + state
    {
        int te = S.t, je = S.j;
        const int t_state = te;
        if (te > a.live_cap - 1) te = a.live_cap - 1;
        if (je > N - 1) je = N - 1;
        double *bb = a.bands + (size_t)b * 2 * (c + 1);
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int i = tid; i <= c; i += NT) {
            const int y = je - c + i, x = te - c + i;
            bb[i] = (y >= 0) ? S.R[swz<W>(y)] : qnan;
            bb[(c + 1) + i] = (x >= 0 && x <= t_state) ? S.C[swz<W>(x)] : qnan;
        }
    }
#ifdef RTS_OTW_STAMPS
    if (tid == 0 && a.debug)
        for (int i = 0; i < 12; i++) a.debug[(size_t)b * 16 + i] = stamp_sum[i];
#endif
    if (tid == 0) {
        st[RTS_ST_T] = t;
        st[RTS_ST_J] = j;
        // LiveNote's set_live keeps the direction in a local; self.direction stays "both"
        st[RTS_ST_DIRECTION] = deferred_update ? RTS_DIR_BOTH : dir;
        st[RTS_ST_PREVIOUS] = prev;
        st[RTS_ST_RUN_COUNT] = run_count;
        st[RTS_ST_STATUS] = status;
        st[RTS_ST_FIRST_INSERT] = first;
        st[RTS_ST_N_PATH] = n_path;
        st[RTS_ST_CONSUMED] = consumed;
        st[RTS_ST_ROW_STRIPS] = rows;
        st[RTS_ST_COL_STRIPS] = cols;
        st[RTS_ST_CELLS_LO] = (int32_t)(uint32_t)(cells & 0xffffffffLL);
        st[RTS_ST_CELLS_HI] = (int32_t)(uint32_t)((unsigned long long)cells >> 32);
        st[RTS_ST_PATH_TRUNCATED] = truncated;
        st[14] = pend_dir;
        st[RTS_ST_BAND_RECOMPUTES] = recomputes;
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
