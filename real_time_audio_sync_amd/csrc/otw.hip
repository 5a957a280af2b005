// Batched online time warping for gfx950 (MI355X): OnlineTimeWarping / LiveNote / LiveNoteV2.
//
// Reference semantics: /root/reference/otw_eran.py:38-239, livenote.py:37-226,
// livenote_v2.py:43-236 (restated on the CPU, for tests only, in oracle/rtsync_oracle.c).
//
// Design:
//   * one workgroup (NW waves) per live stream; the whole per-stream state lives in LDS for the
//     duration of a launch: the two live accumulated-cost bands (row t over columns [j-c, j] and
//     column j over rows [t-c, t]), ring windows of the last W live (and, plain kernel, reference)
//     chroma frames, and the cost strips the helper waves prepare ahead of time;
//   * a strip (<= c cells of one row or one column) is one wave's job (strip_chain): with the
//     pre-computed costs d it forms a = min(up + d, diag + 2d) from the previous band and solves
//     acc_k = min(a_k, acc_{k-1} + d_k) along the strip.  This is a serial float64 recurrence whose
//     rounding must not change, so it is solved by *chunked speculative carry propagation*: every
//     lane scans its L = W/64 consecutive cells, then lanes repeatedly re-scan with the neighbour's
//     last value as carry-in (DPP wave shift) until no carry changes.  Because x -> fl(x + d) and
//     min are monotone, the fixed point is bit-identical to the sequential scan; it is reached after
//     (longest carry run / L) + 1 rounds (1.8 extra rounds on average at c = 500, against 500
//     dependent steps).  The same wave reduces the strip's np.argmin while the values are in registers;
//   * wave 0 owns the control state in registers: corner cell, best_point / direction / run-count /
//     path logic (decide), and the plan for the next step;
//   * plain kernel (SPEC = false): chain phase (row strip on wave 0, column strip on wave 1 in a
//     "Both" step), barrier, control phase, barrier; helper waves compute the next step's costs meanwhile;
//   * pipelined kernel (SPEC = true, the default): while wave 0 runs the control work of a step, waves 1
//     and 2 already run the row / column strip the *next* step needs if it is Row-only / Column-only
//     (minus the last cell) into shadow bands; that step is then a "hit" -- the shadow becomes the
//     band, wave 0 finishes one cell and decides -- with a single barrier.  See the comment at the
//     step loops.
// All arithmetic is float64 in the oracle's operation order (build with -ffp-contract=off), so
// accumulated costs are bit-identical to the CPU restatement, not merely close.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

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
    int spec;                 // 1: the pipelined kernel was selected (host-side choice; the kernel does not read it)
    double *dense_acc;        // optional [B][2N][N]: the reference's dense acc_cost (otw_eran.py:27), NULL = off
    double *dense_cost;       // optional [B][2N][N]: the reference's dense cost (otw_eran.py:23)
};

// RT = element type of the feature rings: double, or float when both inputs are float32 (their values
// widen back exactly, so results are identical while the plain kernel's LDS drops from 131 KB to 74 KB at
// W = 512 -- 58 KB for the pipelined kernel, which keeps no reference ring -- and two workgroups share a CU:
// what matters beyond 256 concurrent streams).
// Ring element type tag of the pipelined kernel's second flavour: no live ring in LDS at all, the helper waves read
// live frames from global memory like reference frames (otw_live_frame).  Taken for float64 features (a float64 ring
// costs 49 KB at W = 512 and with it three of four workgroups per CU: B = 4096 75.2 -> 60.7 ms, B = 512 9.35 -> 7.52 ms)
// and for the 1024-cell window (no ring of either type fits beside its bands).  With float32 features at W <= 512 the
// ring stays: 4.50 vs 4.62 ms at B = 64, 46.9 vs 48.4 ms at B = 4096 (same-call A/B, one box).
struct LiveFromGlobal {};
// The same without a live ring, built for residency (batches of two streams per CU or more): 73 VGPRs instead of
// 111, so three workgroups share a CU.  What it gives up for that: the helper waves keep one cost cell in flight
// instead of two, and the chains of the rare steps that are not hits run on wave 2 (idle in such a step) instead of
// wave 0, whose registers then hold the control state only.  Same arithmetic, same results.
struct LiveFromGlobalLean {};
template <typename RT> struct RingElem { using type = RT; };
template <> struct RingElem<LiveFromGlobal> { using type = float; };
template <> struct RingElem<LiveFromGlobalLean> { using type = float; };
template <typename RT> constexpr bool kThroughput = std::is_same<RT, LiveFromGlobalLean>::value;
template <typename RT> constexpr bool kHasLiveRing = !std::is_same<RT, LiveFromGlobal>::value && !kThroughput<RT>;

template <int W, typename RT>
struct OtwLds {
    static constexpr int L = W / 64;     // cells per lane in the chain phase
    static constexpr int SWZ = L * 65;   // swizzled band length (one pad slot per row of 64)
    double R[SWZ];      // acc[t][.]  row band
    double C[SWZ];      // acc[.][j]  column band
    double Dr[2][SWZ];  // row strip cell costs: [buf] = this step's, [buf^1] = being pre-computed for the next
    double Dc[2][SWZ];
    using E = typename RingElem<RT>::type;
    E livew[kF][kHasLiveRing<RT> ? W : 1];  // feature-major ring of live frames (index x & (W-1))
    // column chain wave -> wave 0, read back in one go
    double cfresh_min;
    double corner_pa;  // Both step: acc[t-1][jn-1] + 2 d(t, jn), stashed before column jn-1 is overwritten
    double corner_d;   // d(t, jn)
    int cfresh_idx;
    // row chain wave -> wave 0 (only where the row chain of a step that is not a hit runs on another wave)
    int rfresh_idx;
    double rfresh_min;
    // wave 0 -> everyone: the next step.  The pipelined kernel alternates between the two slots, because its hit
    // steps have no barrier between the other waves' read of a plan and wave 0's write of the next one.
    int plan_t[2], plan_j0[2], plan_flags[2];
    int t, j;                         // final position, published at exit for the epilogue
    // feature-major ring of reference frames (index y & (W-1)).  Last member: the pipelined kernel reads the
    // reference -- shared by all streams, L2-resident -- straight from global memory in its helper waves and
    // allocates the struct only up to here (58 KB instead of 82 KB at W = 512 with float32 features).
    E refw[kF][W];
};

// Extra LDS of the pipelined kernel (SPEC).  The row band lives in R or ShR and the column band in C or ShC
// (the plan says which); the other buffer of each pair is the *shadow* the speculating wave fills with the strip the
// next Row-only / Column-only step needs, minus its last cell.  When that step comes, the shadow simply becomes the
// band.  ex[] (indexed by step parity) is what the speculating waves hand to wave 0 along with a shadow.
struct OtwSpecOut {
    double min;  // np.argmin of the shadow strip
    double d;    // cost of the strip's last cell (the one the shadow leaves out)
    double d2;   // column speculation only: cost of the corner (t+1, j+1) a Both step would add
    int idx;
    int flag;    // column speculation: 1 if dropping the strip's first cell leaves every other cell unchanged
};
template <int W>
struct alignas(16) OtwSpecLds {
    static constexpr int SWZ = (W / 64) * 65;
    double ShR[SWZ];
    double ShC[SWZ];
    OtwSpecOut row[2], col[2];
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

// lane i receives lane i-1's value; lane 0 receives `lane0` (DPP wave_shr:1; lanes without a source keep `old`).
__device__ __forceinline__ double wave_shift_up(double v, double lane0) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(lane0), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(lane0), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_bcast(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// A value every lane holds identically (an LDS broadcast read) -> SGPRs, so the control logic that
// consumes it runs on the scalar unit.
__device__ __forceinline__ double rfl(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Minimum over the 64 lanes, returned in every lane.  Six DPP steps (row_shr 1/2/4/8, then
// row_bcast 15 and 31) leave the total in lane 63; lanes without a DPP source keep their own value.
#define RTS_DPP_MIN_STEP(CTRL, ROWMASK)                                                                       \
    do {                                                                                                      \
        const int tlo = __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), CTRL, ROWMASK, 0xf, false); \
        const int thi = __builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), CTRL, ROWMASK, 0xf, false); \
        x = vmin(x, __hiloint2double(thi, tlo));                                                              \
    } while (0)

__device__ __forceinline__ double wave_min(double x) {
    RTS_DPP_MIN_STEP(0x111, 0xf);  // row_shr:1
    RTS_DPP_MIN_STEP(0x112, 0xf);  // row_shr:2
    RTS_DPP_MIN_STEP(0x114, 0xf);  // row_shr:4
    RTS_DPP_MIN_STEP(0x118, 0xf);  // row_shr:8
    RTS_DPP_MIN_STEP(0x142, 0xa);  // row_bcast:15
    RTS_DPP_MIN_STEP(0x143, 0xc);  // row_bcast:31
    return wave_bcast(x, 63);
}

// LDS access with the row offset as an instruction immediate (the pipelined kernel's strips; see strip_chain).  The
// compiler does not see these as memory operations it must wait for: chain_wait() is the s_waitcnt, tied to the loaded
// values so that no use can move above it.
__device__ __forceinline__ unsigned lds_addr(const double *p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const double *)p;
}
template <int OFF>
__device__ __forceinline__ double lds_read_imm(unsigned addr) {
    double r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF>
__device__ __forceinline__ void lds_write_imm(unsigned addr, double v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int L, int R, int M = 0>
__device__ __forceinline__ void chain_load(unsigned bA, unsigned bB, unsigned dA, unsigned dB, double (&prevb)[L + 1],
                                           double (&D)[L]) {
    if constexpr (M <= L) {
        constexpr int off = ((R + M) & (L - 1)) * 65 * 8;
        constexpr bool wrap = (R + M) >= L;
        prevb[M] = lds_read_imm<off>(wrap ? bB : bA);
        if constexpr (M > 0) D[M - 1] = lds_read_imm<off>(wrap ? dB : dA);
        chain_load<L, R, M + 1>(bA, bB, dA, dB, prevb, D);
    }
}
template <int L, int M = 0>
__device__ __forceinline__ void chain_wait(double (&prevb)[L + 1], double (&D)[L]) {
    if constexpr (M == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (M <= L) {
        asm volatile("" : "+v"(prevb[M]));
        if constexpr (M < L) asm volatile("" : "+v"(D[M]));
        chain_wait<L, M + 1>(prevb, D);
    }
}
template <int L, int R, int M = 1>
__device__ __forceinline__ void chain_store(unsigned oA, unsigned oB, const double (&v)[L], double v_top) {
    if constexpr (M <= L) {
        constexpr int off = ((R + M) & (L - 1)) * 65 * 8;
        lds_write_imm<off>((R + M) >= L ? oB : oA, (M == L) ? v_top : v[M - 1]);
        chain_store<L, R, M + 1>(oA, oB, v, v_top);
    }
}

// One strip, one wave.  Cell i (band position k = k1 + i, i in [0, n)) gets
//     a_i   = min(band[k] + d_i, band[k-1] + 2 d_i)      the two predecessors outside the strip
//                                                        (band = previous row for a row strip,
//                                                        previous column for a column strip)
//     acc_i = min(a_i, acc_{i-1} + d_i),  acc_{-1} = x_in the predecessor inside the strip
// and the new strip goes to `band_out`: `band` itself for a regular strip (in place -- the wave reads all old
// values before it writes), a shadow strip for a speculative one.  GUARD (speculative strips, which run while
// wave 0 is writing the corner slots just past the strip's inputs): lanes whose cells lie past the strip end
// read band position k1-1 instead of their own slots, so no slot that is being written is ever read.
// Returns np.argmin (first minimum) of the new strip restricted to band positions >= lo_arg:
// (fmin, fidx), fidx = 0x7fffffff if that range is empty.  The caller reads acc_{n-1} back from
// band[swz(k1+n-1)].
//
// The recurrence is a serial float64 chain whose rounding must not change; it is solved exactly by
// chunked speculative carry propagation (see the file header).
//
// Branch-free by construction.  The wave always covers W consecutive ring positions k1 .. k1+W-1;
// cells past the strip end get d = +inf, hence acc = +inf, and are stored like the others.  That is
// safe because those positions are exactly the ring slots *outside* the live band [k1-1, k1+n-1]
// except k1-1 itself, which the caller rewrites with the sentinel right after, and (Both step) the
// corner slot, which the fix-up rewrites; every band position is written with its real value
// before it is ever read (rows/columns only grow at the top index).  Valid cells are always finite:
// each has a computed predecessor in the previous row (row strip) or column (column strip).
template <int W, bool DENSE, bool GUARD = false, bool ALT = false>
__device__ __forceinline__ void strip_chain(const double *__restrict__ Dv, const double *band, double *band_out, int k1,
                                            int n, double x_in, int lane, int lo_arg, double &fmin_out, int &fidx_out,
                                            double *dense_acc, double *dense_cost, long long dense_stride,
                                            long long *rounds_acc = nullptr, bool prev0_xin = false, double alt_xin = 0.0,
                                            int *alt_ok = nullptr) {
    constexpr int L = W / 64;
    constexpr int LOG_L = (L == 1) ? 0 : (L == 2) ? 1 : (L == 4) ? 2 : (L == 8) ? 3 : (L == 16) ? 4 : 5;
    const double inf = INFINITY;
    // cell (lane, m) sits at band position q + L*lane with q = k1 + m wave-uniform, i.e. at LDS slot
    // (q mod L)*65 + ((q / L + lane) mod 64); slot[m] addresses position k - 1 of cell m
    int slot[L + 1];
    const int nloc = n - L * lane;  // cell m of this lane is inside the strip iff m < nloc
    double prevb[L + 1], D[L], v[L];
    // GUARD strips (the two per step of the pipelined kernel) address their slots through one of L code variants,
    // selected by the strip start's row in the swizzle: inside a variant every row offset is an immediate and only
    // the two column words (this lane's and the next one's) sit in registers.  Cells past the strip end read
    // whatever their slots hold -- possibly a corner slot wave 0 is writing at that moment; their cost is +inf
    // below, so the value never reaches a result (an aligned 8-byte LDS access is one operation, and even a NaN is
    // dropped by v_min_f64).
    const int q0 = k1 - 1 + W;  // + W keeps it >= 0 without changing the slot
    const int r0 = q0 & (L - 1);
    const int colA = ((q0 >> LOG_L) + lane) & 63, colB = (colA + 1) & 63;
    if constexpr (GUARD) {
        const unsigned bA = lds_addr(band + colA), bB = lds_addr(band + colB), dA = lds_addr(Dv + colA), dB = lds_addr(Dv + colB);
        switch (r0) {
// (One wait behind the join serves all cases.  Waiting inside every case instead costs 2 % of the B = 64 step -- 4.40 vs
// 4.32 ms, same-call A/B -- so what the single wait relies on, that no case ends with a copy of a register whose LDS
// data is still in flight, is checked on the built code object: tools/check_lds_waits.py, tests/test_abi.py.)
#define RTS_CHAIN_LOAD(R)                                                        \
    case R:                                                                      \
        if constexpr (R < L) chain_load<L, R>(bA, bB, dA, dB, prevb, D);         \
        break;
            RTS_CHAIN_LOAD(0) RTS_CHAIN_LOAD(1) RTS_CHAIN_LOAD(2) RTS_CHAIN_LOAD(3) RTS_CHAIN_LOAD(4) RTS_CHAIN_LOAD(5)
            RTS_CHAIN_LOAD(6) RTS_CHAIN_LOAD(7) RTS_CHAIN_LOAD(8) RTS_CHAIN_LOAD(9) RTS_CHAIN_LOAD(10) RTS_CHAIN_LOAD(11)
            RTS_CHAIN_LOAD(12) RTS_CHAIN_LOAD(13) RTS_CHAIN_LOAD(14) RTS_CHAIN_LOAD(15)
            RTS_CHAIN_LOAD(16) RTS_CHAIN_LOAD(17) RTS_CHAIN_LOAD(18) RTS_CHAIN_LOAD(19) RTS_CHAIN_LOAD(20) RTS_CHAIN_LOAD(21)
            RTS_CHAIN_LOAD(22) RTS_CHAIN_LOAD(23) RTS_CHAIN_LOAD(24) RTS_CHAIN_LOAD(25) RTS_CHAIN_LOAD(26) RTS_CHAIN_LOAD(27)
            RTS_CHAIN_LOAD(28) RTS_CHAIN_LOAD(29) RTS_CHAIN_LOAD(30) RTS_CHAIN_LOAD(31)
#undef RTS_CHAIN_LOAD
        }
        chain_wait<L>(prevb, D);
    } else {
        // cell (lane, m) sits at band position q + L*lane with q = k1 + m wave-uniform, i.e. at LDS slot
        // (q mod L)*65 + ((q / L + lane) mod 64); slot[m] addresses position k - 1 of cell m
#pragma unroll
        for (int m = 0; m <= L; m++) {
            const int q = k1 + m - 1 + W;
            slot[m] = (q & (L - 1)) * 65 + (((q >> LOG_L) + lane) & 63);
        }
#pragma unroll
        for (int m = 0; m <= L; m++) prevb[m] = band[slot[m]];  // positions k-1 .. k+L-1
#pragma unroll
        for (int m = 0; m < L; m++) D[m] = Dv[slot[m + 1]];
    }
    prevb[0] = (k1 == 0 && lane == 0) ? inf : prevb[0];  // row/column 0 has no diagonal predecessor
    // speculative strips only: position k1-1 is being rewritten with the sentinel by wave 0 in this very step
    if (GUARD && ALT) prevb[0] = (prev0_xin && lane == 0) ? x_in : prevb[0];
#pragma unroll
    for (int m = 0; m < L; m++) D[m] = (m < nloc) ? D[m] : inf;
    // round 0: every lane scans its own cells; only lane 0 knows its true carry-in
    double p = (lane == 0) ? x_in : inf;
    double am1 = inf;  // kept for the alt_ok test below
#pragma unroll
    for (int m = 0; m < L; m++) {
        const double am = vmin(prevb[m + 1] + D[m], prevb[m] + 2 * D[m]);
        if (GUARD && ALT && m == 1) am1 = am;
        p = p + D[m];
        v[m] = vmin(am, p);
        p = v[m];
    }
    if (GUARD && ALT && L >= 2) {
        // Would the strip that starts one cell later -- carry-in alt_xin at cell 1 instead of cell 0's value -- be the
        // same from cell 1 on?  Lane 0's cells are final after round 0, and cell 1's value decides everything after it.
        if (alt_ok) *alt_ok = __builtin_amdgcn_readfirstlane((int)(vmin(am1, alt_xin + D[1]) == v[1]));
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
        if (rounds_acc) *rounds_acc += 1;  // diagnostic builds only
        v[0] = vmin(v[0], q);
#pragma unroll
        for (int m = 1; m < L; m++) {
            q = q + D[m];
            v[m] = vmin(v[m], q);
        }
    }
    if constexpr (GUARD) {
        // a shadow strip is installed as it is: position k1-1 (= k1+W-1 on the ring, lane 63's last cell, always
        // past the strip end) gets the value an in-place strip's caller writes there afterwards
        const double v_top = (lane == 63) ? x_in : v[L - 1];
        const unsigned oA = lds_addr(band_out + colA), oB = lds_addr(band_out + colB);
        switch (r0) {
#define RTS_CHAIN_STORE(R)                                                       \
    case R:                                                                      \
        if constexpr (R < L) chain_store<L, R>(oA, oB, v, v_top);                \
        break;
            RTS_CHAIN_STORE(0) RTS_CHAIN_STORE(1) RTS_CHAIN_STORE(2) RTS_CHAIN_STORE(3) RTS_CHAIN_STORE(4)
            RTS_CHAIN_STORE(5) RTS_CHAIN_STORE(6) RTS_CHAIN_STORE(7) RTS_CHAIN_STORE(8) RTS_CHAIN_STORE(9)
            RTS_CHAIN_STORE(10) RTS_CHAIN_STORE(11) RTS_CHAIN_STORE(12) RTS_CHAIN_STORE(13) RTS_CHAIN_STORE(14)
            RTS_CHAIN_STORE(15) RTS_CHAIN_STORE(16) RTS_CHAIN_STORE(17) RTS_CHAIN_STORE(18) RTS_CHAIN_STORE(19)
            RTS_CHAIN_STORE(20) RTS_CHAIN_STORE(21) RTS_CHAIN_STORE(22) RTS_CHAIN_STORE(23) RTS_CHAIN_STORE(24)
            RTS_CHAIN_STORE(25) RTS_CHAIN_STORE(26) RTS_CHAIN_STORE(27) RTS_CHAIN_STORE(28) RTS_CHAIN_STORE(29)
            RTS_CHAIN_STORE(30) RTS_CHAIN_STORE(31)
#undef RTS_CHAIN_STORE
        }
    } else {
#pragma unroll
        for (int m = 0; m < L; m++) band_out[slot[m + 1]] = v[m];
    }
    if (DENSE) {  // optional: mirror the strip into the reference's dense matrices (cell i at base + i*stride)
#pragma unroll
        for (int m = 0; m < L; m++) {
            if (m < nloc) {
                const long long off = (long long)(L * lane + m) * dense_stride;
                dense_acc[off] = v[m];
                dense_cost[off] = D[m];
            }
        }
    }
    // first minimum over positions >= lo_arg: lo_arg is k1 or k1 + 1, so at most the very first
    // cell is excluded
    const double v0 = (lane == 0 && lo_arg > k1) ? inf : v[0];
    double lm = v0;
#pragma unroll
    for (int m = 1; m < L; m++) lm = vmin(lm, v[m]);
    const double g = wave_min(lm);
    int ml = L;  // first cell of this lane that holds the minimum (L: none)
#pragma unroll
    for (int m = L - 1; m >= 1; m--) ml = (v[m] == g) ? m : ml;
    ml = (v0 == g) ? 0 : ml;
    const unsigned long long mask = __ballot(ml != L);
    const bool some = (g < inf) && (mask != 0);
    const int first = (int)__builtin_ctzll(mask | (1ull << 63));  // lanes hold increasing positions
    fmin_out = g;
    fidx_out = some ? k1 + L * first + __builtin_amdgcn_readlane(ml, first) : 0x7fffffff;
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
// shipped library compiles them away.  -DRTS_OTW_STAMPS=1: per-phase stamps (they inflate what they measure);
// -DRTS_OTW_STAMPS=2: only wave 0's work / end-of-step wait split of the pipelined kernel, two stamps per step.
#if defined(RTS_OTW_STAMPS) && RTS_OTW_STAMPS == 1
#define RTS_STAMP2(slot)      \
    do {                      \
        if (stamp_base)       \
            RTS_STAMP(8 + slot); \
        else                  \
            RTS_STAMP(slot);  \
    } while (0)
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
#define RTS_STAMP2(slot) \
    do {                 \
    } while (0)
#endif

// ---- control state of one stream.  Lives in wave 0's registers (every lane identical) for the whole
// launch; the other waves only ever see the 3-word plan in LDS.
struct OtwCtl {
    int t, j, dir, prev, run_count, status, first, n_path, consumed, rows, cols, truncated, pend_dir;
    int recomputes, pending_col, last_x, last_y, rb_idx, cb_idx;
    int live_hi, ref_hi;  // highest frame index present in the live / reference ring
    int spec_valid;       // pipelined kernel: the shadow strips were computed from the current (t, j)
    int ri, ci;           // pipelined kernel: which buffer of each pair holds the row / column band
    double cA, cU, cL;    // pipelined kernel: band slots acc[t][j], column slot t-1, row slot j-1 (hit steps
                          // finish their last cell from these without touching the old band)
    long long cells;
    double rb_min, cb_min;          // np.argmin state of the two bands at the last decide()
    double pfl0, pfl1, pfr0, pfr1;  // prefetched ring frames (two values per lane of wave 0)
};

struct OtwEnv {  // launch-invariant values every helper needs
    int b, lane, c, N, live_len, live_cap, euclid, variant, mode, max_run_count, path_cap, live_f64, ref_f64;
    long long live_base;
    const void *live, *ref;
    int32_t *path;
    bool deferred_update;
};

constexpr int kPlanRow = 1, kPlanCol = 2, kPlanStop = 4, kPlanExit = 8;
constexpr int kPlanHit = 16;  // pipelined kernel: this Row-only / Column-only step's strip already sits in its shadow
constexpr int kPlanRi = 32, kPlanCi = 64;  // pipelined kernel: buffer holding the row / column band in this step
// pipelined kernel: a Both step with a full band that is a hit if the column speculation says so (its flag is only
// known after the barrier, so every wave resolves this bit at the top of the step: otw_resolve_plan)
constexpr int kPlanHitIf = 128;

__device__ __forceinline__ double otw_load_feat(const void *base, int is_f64, long long idx) {
    return is_f64 ? reinterpret_cast<const double *>(base)[idx] : (double)reinterpret_cast<const float *>(base)[idx];
}

// Asynchronous ring refills (wave 0): the next kFetch frames sit in two registers per lane, loaded a
// refill period (~8 steps) before they are written into the ring.
__device__ __forceinline__ void otw_prefetch_live(OtwCtl &k, const OtwEnv &e) {
    const int i0 = e.lane, i1 = e.lane + 64;
    const int f0 = k.live_hi + 1 + i0 / kF, f1 = k.live_hi + 1 + i1 / kF;
    if (f0 < e.live_len) k.pfl0 = otw_load_feat(e.live, e.live_f64, e.live_base + (long long)f0 * kF + i0 % kF);
    if (i1 < kFetch * kF && f1 < e.live_len)
        k.pfl1 = otw_load_feat(e.live, e.live_f64, e.live_base + (long long)f1 * kF + i1 % kF);
}
__device__ __forceinline__ void otw_prefetch_ref(OtwCtl &k, const OtwEnv &e) {
    const int i0 = e.lane, i1 = e.lane + 64;
    const int f0 = k.ref_hi + 1 + i0 / kF, f1 = k.ref_hi + 1 + i1 / kF;
    if (f0 < e.N) k.pfr0 = otw_load_feat(e.ref, e.ref_f64, (long long)f0 * kF + i0 % kF);
    if (i1 < kFetch * kF && f1 < e.N) k.pfr1 = otw_load_feat(e.ref, e.ref_f64, (long long)f1 * kF + i1 % kF);
}
template <int W, typename RT>
__device__ __forceinline__ void otw_commit_live(OtwLds<W, RT> &S, OtwCtl &k, const OtwEnv &e) {
    const int i0 = e.lane, i1 = e.lane + 64;
    const int f0 = k.live_hi + 1 + i0 / kF, f1 = k.live_hi + 1 + i1 / kF;
    if (f0 < e.live_len) S.livew[i0 % kF][f0 & (W - 1)] = (typename RingElem<RT>::type)k.pfl0;
    if (i1 < kFetch * kF && f1 < e.live_len) S.livew[i1 % kF][f1 & (W - 1)] = (typename RingElem<RT>::type)k.pfl1;
    k.live_hi = (k.live_hi + kFetch < e.live_len - 1) ? k.live_hi + kFetch : e.live_len - 1;
    otw_prefetch_live(k, e);
}
template <int W, typename RT>
__device__ __forceinline__ void otw_commit_ref(OtwLds<W, RT> &S, OtwCtl &k, const OtwEnv &e) {
    const int i0 = e.lane, i1 = e.lane + 64;
    const int f0 = k.ref_hi + 1 + i0 / kF, f1 = k.ref_hi + 1 + i1 / kF;
    if (f0 < e.N) S.refw[i0 % kF][f0 & (W - 1)] = (typename RingElem<RT>::type)k.pfr0;
    if (i1 < kFetch * kF && f1 < e.N) S.refw[i1 % kF][f1 & (W - 1)] = (typename RingElem<RT>::type)k.pfr1;
    k.ref_hi = (k.ref_hi + kFetch < e.N - 1) ? k.ref_hi + kFetch : e.N - 1;
    otw_prefetch_ref(k, e);
}

__device__ __forceinline__ void otw_ref_frame(const OtwEnv &e, int q, double (&rf)[kF]);
__device__ __forceinline__ void otw_live_frame(const OtwEnv &e, int r, double (&lf)[kF]);

// Speculative costs for the step after the one that leaves the state at (t_now, j_now): row t_now+1
// over columns [j_now-c+1, j_now] and column j_now+1 over rows [t_now-c+1, t_now+1] cover every
// possible next step (Row, Both or Column).  hidx / hn: index and count of the threads sharing it.
template <int W, typename RT>
__device__ __forceinline__ void otw_precompute(OtwLds<W, RT> &S, const OtwEnv &e, int t_now, int j_now, double *Drn,
                                               double *Dcn, int hidx, int hn) {
    const int c = e.c;
    const int tn = t_now + 1, jn1 = j_now + 1;
    const bool row_ok = (tn < e.live_len) && (tn < e.live_cap);
    const bool col_ok = jn1 < e.N;
    const int k1 = (j_now - c + 1 > 0) ? j_now - c + 1 : 0;
    const int nrow = row_ok ? j_now - k1 + 1 : 0;
    const int r1 = (t_now - c + 1 > 0) ? t_now - c + 1 : 0;
    const int rtop = row_ok ? tn : t_now;
    const int ncol = col_ok ? rtop - r1 + 1 : 0;
    if constexpr (!kHasLiveRing<RT>) {
        // no rings (the dense mirror of a window of 1024 cells or more: the bands leave no room for them): every frame
        // straight from global memory -- the reference is L2-resident, the live frames of a stream are 48 / 96 bytes each
        if (nrow > 0) {
            double lf[kF];
            otw_live_frame(e, tn, lf);
            for (int i = hidx; i < nrow; i += hn) {
                double rf[kF];
                otw_ref_frame(e, k1 + i, rf);
                Drn[swz<W>(k1 + i)] = cell_cost(lf, rf, e.euclid);
            }
        }
        if (ncol > 0) {
            double rf[kF];
            otw_ref_frame(e, jn1, rf);
            for (int i = hidx; i < ncol; i += hn) {
                double lf[kF];
                otw_live_frame(e, r1 + i, lf);
                Dcn[swz<W>(r1 + i)] = cell_cost(lf, rf, e.euclid);
            }
        }
        return;
    }
    if (nrow > 0) {
        double lf[kF];
#pragma unroll
        for (int f = 0; f < kF; f++) lf[f] = (double)S.livew[f][tn & (W - 1)];
        for (int i = hidx; i < nrow; i += hn) {
            const int k = k1 + i;
            double rf[kF];
#pragma unroll
            for (int f = 0; f < kF; f++) rf[f] = (double)S.refw[f][k & (W - 1)];
            Drn[swz<W>(k)] = cell_cost(lf, rf, e.euclid);
        }
    }
    if (ncol > 0) {
        double rf[kF];
#pragma unroll
        for (int f = 0; f < kF; f++) rf[f] = (double)S.refw[f][jn1 & (W - 1)];
        for (int i = hidx; i < ncol; i += hn) {
            const int r = r1 + i;
            double lf[kF];
#pragma unroll
            for (int f = 0; f < kF; f++) lf[f] = (double)S.livew[f][r & (W - 1)];
            Dcn[swz<W>(r)] = cell_cost(lf, rf, e.euclid);
        }
    }
}

// decide(): best_point + path + direction (otw_eran.py:153-211, livenote_v2.py:193-236).  Wave 0,
// all lanes, register state.  The two band argmins are kept incrementally: a strip computed this
// step brings its own argmin from its chain; a band that merely slid by one cell keeps its minimum
// unless that cell left the window (then a full wave reduction recomputes it); the one cell
// appended at the top index wins only if strictly smaller (np.argmin returns the first minimum).
template <int W, typename RT>
__device__ __forceinline__ void otw_decide(const double *R, const double *C, OtwCtl &k, const OtwEnv &e, int tt, int jj,
                                           bool row_fresh, double rf_min, int rf_idx, bool col_fresh, double cf_min,
                                           int cf_idx, bool row_corner, double rc, bool col_corner, double cc, bool full) {
    const int c = e.c;
    const int j1 = (jj - c + 1 > 0) ? jj - c + 1 : 0;
    const int t1 = (tt - c + 1 > 0) ? tt - c + 1 : 0;
    double rmin = row_fresh ? rf_min : k.rb_min, cmin = col_fresh ? cf_min : k.cb_min;
    int ridx = row_fresh ? rf_idx : k.rb_idx, cidx = col_fresh ? cf_idx : k.cb_idx;
    // full recomputation: always for `full`, otherwise only when a kept minimum slid out of its window
    const bool need_r = full || (!row_fresh && ridx < j1);
    const bool need_c = full || (!col_fresh && cidx < t1);
    if (need_r || need_c) {
        for (int which = 0; which < 2; which++) {  // a single inlined instance of the wave reduction
            if (!(which ? need_c : need_r)) continue;
            const double *band = which ? C : R;
            const int lo = which ? t1 : j1;
            const int hi = which ? ((col_corner && !full) ? tt - 1 : tt) : ((row_corner && !full) ? jj - 1 : jj);
            double vm;
            int im;
            band_argmin<W>(band, lo, hi, e.lane, vm, im);
            if (which) {
                cmin = vm;
                cidx = im;
            } else {
                rmin = vm;
                ridx = im;
            }
            if (!full) k.recomputes += 1;
        }
    }
    if (!full) {
        if (row_corner && rc < rmin) {
            rmin = rc;
            ridx = jj;
        }
        if (col_corner && cc < cmin) {
            cmin = cc;
            cidx = tt;
        }
    }
    k.rb_min = rmin;
    k.rb_idx = ridx;
    k.cb_min = cmin;
    k.cb_idx = cidx;
    int x, y;
    if (rmin < cmin) {
        x = tt;
        y = ridx;
    } else {
        x = cidx;
        y = jj;
    }
    // every lane holds the same control state; saying so lets the direction logic below, and the next plan that
    // follows from it, run on the scalar unit
    x = __builtin_amdgcn_readfirstlane(x);
    y = __builtin_amdgcn_readfirstlane(y);
    k.run_count = __builtin_amdgcn_readfirstlane(k.run_count);
    k.prev = __builtin_amdgcn_readfirstlane(k.prev);
    bool append = true;
    if (e.variant == RTS_VARIANT_LIVENOTE_V2)  // livenote_v2.py:198
        append = (k.n_path == 0) || (x > k.last_x && y >= k.last_y);
    if (append) {
        if (k.n_path < e.path_cap) {
            if (e.lane == 0) {
                int2 *pp = reinterpret_cast<int2 *>(e.path) + ((size_t)e.b * e.path_cap + k.n_path);
                *pp = make_int2(x, y);
            }
        } else {
            k.truncated = 1;
        }
        k.n_path += 1;
        k.last_x = x;
        k.last_y = y;
    }
    int nd;
    if (tt < c)
        nd = RTS_DIR_BOTH;
    else if (k.run_count >= e.max_run_count)
        nd = (k.prev == RTS_DIR_ROW) ? RTS_DIR_COLUMN : RTS_DIR_ROW;
    else if (x < tt)
        nd = RTS_DIR_COLUMN;
    else if (y < jj)
        nd = RTS_DIR_ROW;
    else
        nd = RTS_DIR_BOTH;
    if (e.deferred_update) {
        k.pend_dir = nd;
    } else {
        k.run_count = (nd == k.prev) ? k.run_count + 1 : 1;
        if (nd != RTS_DIR_BOTH) k.prev = nd;
    }
    k.dir = nd;
    k.pending_col = (nd == RTS_DIR_COLUMN);
    k.t = tt;
    k.j = jj;
}

// Plan for the next step from the current register state (wave 0); lane 0 publishes it.
template <int W, typename RT>
__device__ __forceinline__ void otw_refill(OtwLds<W, RT> &S, OtwCtl &k, const OtwEnv &e, int need_live, int need_ref,
                                           bool with_ref = true) {
    const int need_l = (need_live < e.live_len - 1) ? need_live : e.live_len - 1;
    if (need_l > k.live_hi) otw_commit_live<W, RT>(S, k, e);
    if (with_ref) {
        const int need_r = (need_ref < e.N - 1) ? need_ref : e.N - 1;
        if (need_r > k.ref_hi) otw_commit_ref<W, RT>(S, k, e);
    }
}

struct OtwPlan {
    int t, j0, flags;
};

template <int W, typename RT>
__device__ __forceinline__ OtwPlan otw_make_plan(OtwLds<W, RT> &S, OtwCtl &k, const OtwEnv &e, bool spec = false,
                                                 int slot = 0) {
    int flags = 0, pt = k.t;
    if (k.status != RTS_RUNNING) {
        flags = kPlanExit;
    } else if (k.pending_col) {
        flags = kPlanCol;
    } else if (k.t + 1 >= e.live_len) {  // live sequence exhausted
        if (e.mode == RTS_MODE_SET_LIVE) k.t = k.t + 1;  // otw_eran.py:111-115
        flags = kPlanExit;
    } else if (k.t + 1 >= e.live_cap) {  // otw_eran.py:53-55
        k.status = RTS_LIVE_OVERFLOW;
        k.t = e.live_len - 1;
        k.consumed = e.live_len;
        flags = kPlanExit;
    } else {
        pt = k.t + 1;
        flags = kPlanRow | ((k.dir != RTS_DIR_ROW) ? kPlanCol : 0);
    }
    if ((flags & kPlanCol) && k.j + 1 >= e.N) flags |= kPlanStop;  // otw_eran.py:67-71
    if (spec) {
        // the shadow becomes the band: any Row-only / Column-only step, and a Both step while both of its strips
        // still start at 0 like the speculated ones (otw_settle_hit_both)
        const bool both = flags == (kPlanRow | kPlanCol);
        const bool both_ok = both && k.t <= e.c - 2 && k.j <= e.c - 2;
        if (k.spec_valid && (flags == kPlanRow || flags == kPlanCol || both_ok)) {
            flags |= kPlanHit;
            if (flags & kPlanRow) k.ri ^= 1;
            if (flags & kPlanCol) k.ci ^= 1;
        } else if (k.spec_valid && both && W >= 128 && k.t - ((k.t - e.c + 1 > 0) ? k.t - e.c + 1 : 0) >= 2) {
            // a Both step later on: still the two speculated strips, if the column strip survives losing its first
            // cell (otw_resolve_plan); needs two cells per lane in the chain (W >= 128) and two cells in the strip
            flags |= kPlanHitIf;
        }
        flags |= (k.ri ? kPlanRi : 0) | (k.ci ? kPlanCi : 0);
    }
    if (e.lane == 0) {
        S.plan_t[slot] = pt;
        S.plan_j0[slot] = k.j;
        S.plan_flags[slot] = flags;
        if (flags & kPlanExit) {
            S.t = k.t;
            S.j = k.j;
        }
    }
    // keep the rings one frame ahead of what this step's cost pre-computation will read (the pipelined kernel
    // leaves the rings to its first helper wave)
    if constexpr (kHasLiveRing<RT>) {
        if (!spec && !(flags & kPlanExit)) {
            const int jn_p = k.j + ((flags & kPlanCol) ? 1 : 0);
            otw_refill<W, RT>(S, k, e, pt + 1, jn_p + 1);
        }
    }
    OtwPlan p;
    p.t = pt;
    p.j0 = k.j;
    p.flags = flags;
    return p;
}

// This step's row strip (t, [k1r, j0]) on the calling wave; returns its argmin restricted to the row
// band decide() will look at.
template <int W, bool DENSE, typename RT>
__device__ __forceinline__ void otw_row_strip(double *R, const OtwArgs &a, const OtwEnv &e, const double *Dr,
                                              int pt, int j0, int jn, double sentinel, double &rf_min, int &rf_idx) {
    const int c = e.c;
    const int k1r = (j0 - c + 1 > 0) ? j0 - c + 1 : 0, nr = j0 - k1r + 1;
    const double x_in = (k1r > 0) ? sentinel : (double)INFINITY;  // (t, k1r-1) was never evaluated
    const int lo_arg = (jn - c + 1 > 0) ? jn - c + 1 : 0;        // row band's lower end at decide()
    const long long dro = ((long long)e.b * e.live_cap + pt) * e.N + k1r;  // cell (pt, k1r)
    strip_chain<W, DENSE>(Dr, R, R, k1r, nr, x_in, e.lane, lo_arg, rf_min, rf_idx, DENSE ? a.dense_acc + dro : nullptr,
                          DENSE ? a.dense_cost + dro : nullptr, 1);
    if (e.lane == 0 && k1r > 0) R[swz<W>(k1r - 1)] = sentinel;
}

// This step's column strip ([k1c, pt or pt-1], jn) on the calling wave; in a Both step (with_row) the corner
// cell (pt, jn) is left to the fix-up and its diagonal term is stashed before column jn-1 is overwritten.
template <int W, bool DENSE, typename RT>
__device__ __forceinline__ void otw_col_strip(OtwLds<W, RT> &S, double *C, const OtwArgs &a, const OtwEnv &e,
                                              const double *Dc, int pt, int jn, bool with_row, double sentinel) {
    const int c = e.c;
    const int k1c = (pt - c + 1 > 0) ? pt - c + 1 : 0, nc = pt - k1c + 1;
    const double x_in = (k1c > 0) ? sentinel : (double)INFINITY;  // (k1c-1, jn) was never evaluated
    const int ncc = nc - (with_row ? 1 : 0);
    double fm;
    int fi;
    const double dcorner = Dc[swz<W>(pt)];
    const double pa = (with_row && pt > 0) ? C[swz<W>(pt - 1)] + 2 * dcorner : (double)INFINITY;
    const long long dco = ((long long)e.b * e.live_cap + k1c) * e.N + jn;  // cell (k1c, jn)
    strip_chain<W, DENSE>(Dc, C, C, k1c, ncc, x_in, e.lane, k1c, fm, fi, DENSE ? a.dense_acc + dco : nullptr,
                          DENSE ? a.dense_cost + dco : nullptr, e.N);
    if (e.lane == 0) {
        S.corner_pa = pa;
        S.corner_d = dcorner;
        if (k1c > 0) C[swz<W>(k1c - 1)] = sentinel;
        S.cfresh_min = fm;
        S.cfresh_idx = fi;
    }
}

// What decide() needs to know about the step that just ran.
struct OtwSettled {
    bool row_fresh, col_fresh, row_corner, col_corner, stop;
    double rf_min, cf_min, rc, cc;
    int rf_idx, cf_idx;
};

// First half of the control phase of a regular step (wave 0): counters and the corner fix-up.
template <int W, bool DENSE, typename RT>
__device__ __forceinline__ OtwSettled otw_settle(OtwLds<W, RT> &S, double *R, double *C, OtwCtl &k, const OtwArgs &a,
                                                 const OtwEnv &e, int pt, int j0, int pflags, double rf_min, int rf_idx,
                                                 double sentinel) {
    const int c = e.c, lane = e.lane;
    const double inf = INFINITY;
    const bool do_row = (pflags & kPlanRow) != 0, do_col = (pflags & kPlanCol) != 0;
    const bool stop = (pflags & kPlanStop) != 0;
    const bool col_active = do_col && !stop;
    const int jn = j0 + (do_col ? 1 : 0);
    const int k1r = (j0 - c + 1 > 0) ? j0 - c + 1 : 0, nr = j0 - k1r + 1;
    const int k1c = (pt - c + 1 > 0) ? pt - c + 1 : 0, nc = pt - k1c + 1;
    if (!stop && e.deferred_update && k.pend_dir != -2) {  // livenote_v2.py:149-155
        k.run_count = (k.pend_dir == k.prev) ? k.run_count + 1 : 1;
        if (k.pend_dir != RTS_DIR_BOTH) k.prev = k.pend_dir;
        k.pend_dir = -2;
    }
    if (do_row) {
        k.rows += 1;
        k.cells += nr;
        k.consumed = pt + 1;
    }
    double cl = 0.0, cf_min = inf;
    int cf_idx = 0x7fffffff;
    // everything this phase reads from LDS, issued back to back (one round trip instead of six; values a step kind
    // does not use are read anyway and ignored)
    const int ncc = nc - (do_row ? 1 : 0);
    const double ld_row_last = R[swz<W>(j0)];
    const double ld_col_last = C[swz<W>(k1c + (ncc > 0 ? ncc - 1 : 0))];
    const double ld_cf_min = S.cfresh_min, ld_corner_d = S.corner_d, ld_corner_pa = S.corner_pa;
    const int ld_cf_idx = S.cfresh_idx;
    // last cell of the row strip = acc[t][j0]; of the column chain = acc[t-1][jn] (Both) or acc[t][jn]
    const double row_last = do_row ? rfl(ld_row_last) : 0.0;
    if (do_row && !col_active && lane == 0) C[swz<W>(pt)] = row_last;  // column j0 gains row t
    if (col_active) {
        cl = (ncc > 0) ? rfl(ld_col_last) : ((k1c > 0) ? sentinel : inf);
        cf_min = rfl(ld_cf_min);
        cf_idx = __builtin_amdgcn_readfirstlane(ld_cf_idx);
        if (do_row) {
            const double d = rfl(ld_corner_d);
            const double av = vmin(row_last + d, rfl(ld_corner_pa));
            cl = vmin(av, cl + d);  // cl was the value of (t-1, jn), or the sentinel carry
            if (lane == 0) {
                C[swz<W>(pt)] = cl;
                if (DENSE) {
                    const long long o = ((long long)e.b * e.live_cap + pt) * e.N + jn;
                    a.dense_acc[o] = cl;
                    a.dense_cost[o] = d;
                }
            }
        }
        if (lane == 0) R[swz<W>(jn)] = cl;  // row t gains column jn
        k.cols += 1;
        k.cells += nc;
    }
    // row band: fresh from this step's row strip, plus the corner a column strip appended;
    // column band: fresh from this step's column strip (its corner cell is outside the
    // chain), or the old band plus the row strip's last cell
    OtwSettled o;
    o.row_fresh = do_row;
    o.rf_min = rf_min;
    o.rf_idx = rf_idx;
    o.col_fresh = col_active;
    o.cf_min = cf_min;
    o.cf_idx = cf_idx;
    o.row_corner = col_active;
    o.rc = cl;
    o.col_corner = do_row;
    o.cc = col_active ? cl : row_last;
    o.stop = stop;
    return o;
}

// Second half: decide() (or the stop at the reference's end) and the next plan.
template <int W, typename RT>
__device__ __forceinline__ void otw_finish(OtwLds<W, RT> &S, OtwCtl &k, const OtwEnv &e, int pt, int jn,
                                           const OtwSettled &o, bool spec = false) {
    if (o.stop) {
        k.status = RTS_STOP_REF_END;
        k.t = pt;
        k.j = jn;
        k.pending_col = 0;
    } else {
        otw_decide<W, RT>(S.R, S.C, k, e, pt, jn, o.row_fresh, o.rf_min, o.rf_idx, o.col_fresh, o.cf_min, o.cf_idx, o.row_corner,
                          o.rc, o.col_corner, o.cc, false);
    }
    k.spec_valid = spec && !o.stop;  // the shadow strips computed during this control phase belong to (k.t, k.j)
    otw_make_plan<W, RT>(S, k, e, spec);
}

// The 12 features of reference frame q, from global memory (frame-major [N][12]: 48 or 96 contiguous bytes).
__device__ __forceinline__ void otw_ref_frame(const OtwEnv &e, int q, double (&rf)[kF]) {
    if (e.ref_f64) {
        const double2 *p = reinterpret_cast<const double2 *>(reinterpret_cast<const double *>(e.ref) + (size_t)q * kF);
#pragma unroll
        for (int i = 0; i < kF / 2; i++) {
            const double2 v = p[i];
            rf[2 * i] = v.x;
            rf[2 * i + 1] = v.y;
        }
    } else {
        const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(e.ref) + (size_t)q * kF);
#pragma unroll
        for (int i = 0; i < kF / 4; i++) {
            const float4 v = p[i];
            rf[4 * i] = (double)v.x;
            rf[4 * i + 1] = (double)v.y;
            rf[4 * i + 2] = (double)v.z;
            rf[4 * i + 3] = (double)v.w;
        }
    }
}

// The 12 features of this stream's live frame r, from global memory (windows without a live ring).
__device__ __forceinline__ void otw_live_frame(const OtwEnv &e, int r, double (&lf)[kF]) {
    if (e.live_f64) {
        const double2 *p = reinterpret_cast<const double2 *>(reinterpret_cast<const double *>(e.live) + e.live_base +
                                                             (long long)r * kF);
#pragma unroll
        for (int i = 0; i < kF / 2; i++) {
            const double2 v = p[i];
            lf[2 * i] = v.x;
            lf[2 * i + 1] = v.y;
        }
    } else {
        const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(e.live) + e.live_base +
                                                           (long long)r * kF);
#pragma unroll
        for (int i = 0; i < kF / 4; i++) {
            const float4 v = p[i];
            lf[4 * i] = (double)v.x;
            lf[4 * i + 1] = (double)v.y;
            lf[4 * i + 2] = (double)v.z;
            lf[4 * i + 3] = (double)v.w;
        }
    }
}

// ---- pipelined kernel ---------------------------------------------------------------------------------------------
// Cost buffers are keyed by row / column parity instead of by step: Dr[r & 1] holds the costs of live row r and
// Dc[q & 1] those of reference column q, at ring positions by column / row index.  Invariant at the start of the step
// that leaves the state at (t, j), for every move it can make:
//     rows t+1 and t+2 over columns [j-c+1, j+2],  columns j+1 and j+2 over rows [t-c+1, t+2].
// The step planned to end at (pt, jn) therefore finds the costs of its own strips and of the strips speculated during
// it already there, and the helpers only restore the invariant for (pt, jn): one new row (pt+2) if t advanced, one new
// column (jn+2) if j advanced.  The cells by which the *kept* rows / columns must grow are exactly the last cells of
// those new strips, so the thread that computes one stores it twice.  None of it is read before the next step, so all
// of this is off the critical path.
template <int W, typename RT>
__device__ __forceinline__ void otw_cost_row(OtwLds<W, RT> &S, const OtwEnv &e, int r, int k_lo, int k_hi, int dual_lo,
                                             int hidx, int hn) {
    if (r >= e.live_len || r >= e.live_cap) return;  // never consumed
    if (k_lo < 0) k_lo = 0;
    if (k_hi > e.N - 1) k_hi = e.N - 1;
    if (k_lo + hidx > k_hi) return;
    double *Drow = S.Dr[r & 1];
    double lf[kF];
    if constexpr (!kHasLiveRing<RT>) {
        otw_live_frame(e, r, lf);
    } else {
#pragma unroll
        for (int f = 0; f < kF; f++) lf[f] = (double)S.livew[f][r & (W - 1)];
    }
    if constexpr (kThroughput<RT>) {  // one cell in flight: registers for residency
        for (int ka = k_lo + hidx; ka <= k_hi; ka += hn) {
            double ra[kF];
            otw_ref_frame(e, ka, ra);
            const double da = cell_cost(lf, ra, e.euclid);
            Drow[swz<W>(ka)] = da;
            if (ka >= dual_lo) S.Dc[ka & 1][swz<W>(r)] = da;
        }
        return;
    }
    for (int ka = k_lo + hidx; ka <= k_hi; ka += 2 * hn) {  // two cells in flight per thread
        const int kb = ka + hn;
        const int kb_c = (kb <= k_hi) ? kb : ka;
        double ra[kF], rb[kF];
        otw_ref_frame(e, ka, ra);
        otw_ref_frame(e, kb_c, rb);
        const double da = cell_cost(lf, ra, e.euclid), db = cell_cost(lf, rb, e.euclid);
        Drow[swz<W>(ka)] = da;
        if (ka >= dual_lo) S.Dc[ka & 1][swz<W>(r)] = da;  // kept column ka gains row r
        if (kb <= k_hi) {
            Drow[swz<W>(kb)] = db;
            if (kb >= dual_lo) S.Dc[kb & 1][swz<W>(r)] = db;
        }
    }
}
template <int W, typename RT>
__device__ __forceinline__ void otw_cost_col(OtwLds<W, RT> &S, const OtwEnv &e, int q, int r_lo, int r_hi, int dual_lo,
                                             int hidx, int hn) {
    if (q >= e.N) return;
    const int lim = (e.live_len < e.live_cap) ? e.live_len : e.live_cap;
    if (r_lo < 0) r_lo = 0;
    if (r_hi > lim - 1) r_hi = lim - 1;
    if (r_lo + hidx > r_hi) return;
    double *Dcol = S.Dc[q & 1];
    double rf[kF];
    otw_ref_frame(e, q, rf);
    if constexpr (kThroughput<RT>) {
        for (int ra_ = r_lo + hidx; ra_ <= r_hi; ra_ += hn) {
            double la[kF];
            otw_live_frame(e, ra_, la);
            const double da = cell_cost(la, rf, e.euclid);
            Dcol[swz<W>(ra_)] = da;
            if (ra_ >= dual_lo) S.Dr[ra_ & 1][swz<W>(q)] = da;
        }
        return;
    }
    for (int ra_ = r_lo + hidx; ra_ <= r_hi; ra_ += 2 * hn) {
        const int rb_ = ra_ + hn;
        const int rb_c = (rb_ <= r_hi) ? rb_ : ra_;
        double la[kF], lb[kF];
        if constexpr (!kHasLiveRing<RT>) {
            otw_live_frame(e, ra_, la);
            otw_live_frame(e, rb_c, lb);
        } else {
#pragma unroll
            for (int f = 0; f < kF; f++) {
                la[f] = (double)S.livew[f][ra_ & (W - 1)];
                lb[f] = (double)S.livew[f][rb_c & (W - 1)];
            }
        }
        const double da = cell_cost(la, rf, e.euclid), db = cell_cost(lb, rf, e.euclid);
        Dcol[swz<W>(ra_)] = da;
        if (ra_ >= dual_lo) S.Dr[ra_ & 1][swz<W>(q)] = da;  // kept row ra_ gains column q
        if (rb_ <= r_hi) {
            Dcol[swz<W>(rb_)] = db;
            if (rb_ >= dual_lo) S.Dr[rb_ & 1][swz<W>(q)] = db;
        }
    }
}

// Establish the invariant for (t, j) from nothing (launch prologue, all threads).
template <int W, typename RT>
__device__ __forceinline__ void otw_costs_prime(OtwLds<W, RT> &S, const OtwEnv &e, int t, int j, int hidx, int hn) {
    const int c = e.c;
    const int none = 0x7fffffff;
    otw_cost_row<W, RT>(S, e, t + 1, j - c + 1, j + 2, none, hidx, hn);
    otw_cost_row<W, RT>(S, e, t + 2, j - c + 1, j + 2, none, hidx, hn);
    otw_cost_col<W, RT>(S, e, j + 1, t - c + 1, t + 2, none, hidx, hn);
    otw_cost_col<W, RT>(S, e, j + 2, t - c + 1, t + 2, none, hidx, hn);
}

// Restore the invariant for (pt, jn) after the move the plan describes (helper waves).  A Both step computes its new
// row first and its new column second; the one cell they share, (pt+2, jn+2), comes out identical from both.
template <int W, typename RT>
__device__ __forceinline__ void otw_costs_advance(OtwLds<W, RT> &S, const OtwEnv &e, int pt, int jn, bool do_row,
                                                  bool do_col, int hidx, int hn) {
    const int c = e.c;
    // new row pt+2: its cells in columns jn+1, jn+2 are what the two column buffers lack (row pt+2); in a Both step
    // column jn+2 is new as well and covers that cell itself
    if (do_row) otw_cost_row<W, RT>(S, e, pt + 2, jn - c + 1, jn + 2, jn + 1, hidx, hn);
    // new column jn+2: its cells in rows pt+1, pt+2 are what the two row buffers lack (column jn+2)
    if (do_col) otw_cost_col<W, RT>(S, e, jn + 2, pt - c + 1, pt + 2, pt + 1, hidx, hn);
}

// A speculative strip (during the step that leaves the state at (pt, jn)): the strip the next Row-only step (row
// pt+1, pos = jn) or Column-only step (column jn+1, pos = pt) would run, without its last cell `pos`, from the band --
// whose slots below `pos` are final when the speculating wave starts -- into the shadow; plus what wave 0 needs to
// finish that step: the strip's argmin and the cost of the cell left out.
template <int W, bool ALT>  // ALT: the column speculation (exports the flag, may be told to ignore position k1-1)
__device__ __forceinline__ void otw_spec_strip(const double *Dv, const double *band, double *shadow, int pos, int c,
                                               int lane, double sentinel, OtwSpecOut *out, long long *rounds_acc = nullptr,
                                               bool prev0_sentinel = false) {
    const int k1 = (pos - c + 1 > 0) ? pos - c + 1 : 0;
    double fm;
    int fi;
    const double d_last = Dv[swz<W>(pos)];
    const double d_next = Dv[swz<W>(pos + 1)];  // meaningful for the column strip: cost of (pt+1, jn+1)
    int alt_ok = 0;
    strip_chain<W, false, true, ALT>(Dv, band, shadow, k1, pos - k1, (k1 > 0) ? sentinel : (double)INFINITY, lane, k1, fm, fi,
                                nullptr, nullptr, 0, rounds_acc, prev0_sentinel, sentinel, &alt_ok);
    if (lane == 0) {
        out->min = fm;
        out->idx = fi;
        out->d = d_last;
        out->d2 = d_next;
        out->flag = alt_ok;
    }
}

// The last cell of a shadow strip, in the chain's own order: min(min(side + d, diag + 2d), previous cell + d).  `side`
// is the old band's slot at `pos` (up for a row strip, left for a column strip), `diag` its slot at pos-1.
// (`prev_raw`: the band's slot at pos - 1, or at pos when the strip has only this cell -- read by the caller, so that wave 0
// can fetch it in the same LDS round trip as everything else it needs for the step)
__device__ __forceinline__ double otw_last_cell_from(double prev_raw, int pos, int k1, double side_v, double diag_v, double d,
                                                     double sentinel) {
    const double inf = INFINITY;
    const int n = pos - k1;
    const double prev = (n > 0) ? prev_raw : ((k1 > 0) ? sentinel : inf);
    const double diag = (pos > 0) ? diag_v + 2 * d : inf;
    return vmin(vmin(side_v + d, diag), prev + d);
}
template <int W>
__device__ __forceinline__ double otw_last_cell(const double *band, int pos, int k1, double side_v, double diag_v, double d,
                                                double sentinel) {
    const int n = pos - k1;  // the strip is [k1, pos]
    return otw_last_cell_from(band[swz<W>(pos - (n > 0 ? 1 : 0))], pos, k1, side_v, diag_v, d, sentinel);
}

// A Both step as a hit (wave 0).  While the band is still filling (t, j <= c-2: both strips start at 0, exactly like
// the speculated ones) the row strip of a Both step is the row shadow plus its last cell a = (pt, j0), the column strip
// the column shadow plus its last cell b = (pt-1, jn), and the corner follows from a, b and acc[pt-1][j0].  Waves 1 and
// 2 need a / b as input of the next speculation and compute them for themselves (same expression, same bits).
template <int W, typename RT>
__device__ __forceinline__ OtwSettled otw_settle_hit_both(double *R, double *C, const OtwSpecOut *exr, const OtwSpecOut *exc,
                                                          OtwCtl &k, const OtwEnv &e, int pt, int j0, double sentinel) {
    const int c = e.c, lane = e.lane;
    const int jn = j0 + 1, t0 = pt - 1;
    if (e.deferred_update && k.pend_dir != -2) {  // livenote_v2.py:149-155
        k.run_count = (k.pend_dir == k.prev) ? k.run_count + 1 : 1;
        if (k.pend_dir != RTS_DIR_BOTH) k.prev = k.pend_dir;
        k.pend_dir = -2;
    }
    // every LDS word the step needs in one round trip: the two speculation records and the two band slots in front of the
    // strips' last cells (the unconditional readfirstlane keeps the compiler from sinking those two reads into the branch
    // that uses them, i.e. behind the first wait: that was a second and a third round trip per Both step)
    const int k1r_ = (j0 - c + 1 > 0) ? j0 - c + 1 : 0, k1c_ = (pt - c + 1 > 0) ? pt - c + 1 : 0;
    const double pra_raw = R[swz<W>(j0 - (j0 - k1r_ > 0 ? 1 : 0))], prb_raw = C[swz<W>(t0 - (t0 - k1c_ > 0 ? 1 : 0))];
    const double d1_raw = exr->d, d2_raw = exc->d, d3_raw = exc->d2, rmin_raw = exr->min, cmin_raw = exc->min;
    const int ridx_raw = exr->idx, cidx_raw = exc->idx;
    const double d1 = rfl(d1_raw), d2 = rfl(d2_raw), d3 = rfl(d3_raw);
    double rmin = rfl(rmin_raw), cmin = rfl(cmin_raw);
    int ridx = __builtin_amdgcn_readfirstlane(ridx_raw), cidx = __builtin_amdgcn_readfirstlane(cidx_raw);
    const double pra = rfl(pra_raw), prb = rfl(prb_raw);
    // the Both step's strips: row pt over [k1r, j0] -- the speculated row strip plus its last cell; column jn over
    // [k1c, pt-1] -- the speculated column strip [k1c_s, pt-2] without its first cell once the band is full
    // (k1c = k1c_s + 1; the speculation's flag vouches that dropping it changes nothing else) plus its last cell
    const int k1r = (j0 - c + 1 > 0) ? j0 - c + 1 : 0, k1c = (pt - c + 1 > 0) ? pt - c + 1 : 0;
    const int lo_r = (jn - c + 1 > 0) ? jn - c + 1 : 0;  // decide() looks at row pt from here on
    const int k1c_s = (t0 - c + 1 > 0) ? t0 - c + 1 : 0;
    const double a = rfl(otw_last_cell_from(pra, j0, k1r, k.cA, k.cL, d1, sentinel));
    const double b = rfl(otw_last_cell_from(prb, t0, k1c, k.cA, k.cU, d2, sentinel));
    const double pa = k.cA + 2 * d3;  // acc[pt-1][j0] + 2 d(pt, jn)
    const double av = vmin(a + d3, pa);
    const double cl = vmin(av, b + d3);
    if (lane == 0) {
        R[swz<W>(j0)] = a;
        C[swz<W>(t0)] = b;
        R[swz<W>(jn)] = cl;
        C[swz<W>(pt)] = cl;
        if (k1c > k1c_s) C[swz<W>(k1c - 1)] = sentinel;  // the dropped cell's slot: what the in-place strip leaves there
    }
    // the strips' last cells sit at the highest index: they win only if strictly smaller (np.argmin)
    if (a < rmin) {
        rmin = a;
        ridx = j0;
    }
    if (b < cmin) {
        cmin = b;
        cidx = t0;
    }
    // a speculated argmin that sits on a cell outside the Both step's window cannot be used: let decide() reduce the
    // band (which holds a and b by now) instead
    const bool row_ok = ridx >= lo_r, col_ok = cidx >= k1c;
    if (!row_ok) k.rb_idx = -1;
    if (!col_ok) k.cb_idx = -1;
    k.rows += 1;
    k.cols += 1;
    k.consumed = pt + 1;
    k.cells += (j0 - k1r + 1) + (pt - k1c + 1);
    k.cU = b;
    k.cL = a;
    k.cA = cl;
    OtwSettled o;
    o.row_fresh = row_ok;
    o.col_fresh = col_ok;
    o.rf_min = rmin;
    o.rf_idx = ridx;
    o.cf_min = cmin;
    o.cf_idx = cidx;
    o.row_corner = o.col_corner = true;
    o.rc = o.cc = cl;
    o.stop = false;
    return o;
}

// Pipelined kernel, every wave at the top of a step: a Both step planned "hit if the column speculation allows"
// becomes a hit (both shadows become the bands) or stays a regular step.  The flag was exported before the barrier
// that ended the previous step, so all waves read the same value.
__device__ __forceinline__ int otw_resolve_plan(int pflags, int pt, int c, const OtwSpecOut *exc) {
    if (pflags & kPlanHitIf) {
        const bool ok = (pt < c) || __builtin_amdgcn_readfirstlane(exc->flag) != 0;  // pt < c: no cell is dropped
        if (ok) pflags = (pflags | kPlanHit) ^ (kPlanRi | kPlanCi);
    }
    return pflags;
}

// A *hit* step (wave 0): this Row-only / Column-only step's strip, all cells but the last, is the shadow that the plan
// has just made the band.  Finish the last cell -- from register copies of the three old-band slots it depends on,
// so the old band's buffer is free for the next speculation straight away -- and hand decide() the strip's argmin.
template <int W, typename RT>
__device__ __forceinline__ OtwSettled otw_settle_hit(double *R, double *C, const OtwSpecOut *ex, OtwCtl &k, const OtwEnv &e,
                                                     int pt, int j0, bool is_row, double sentinel) {
    const int c = e.c, lane = e.lane;
    const double inf = INFINITY;
    const int jn = j0 + (is_row ? 0 : 1);
    const int pos = is_row ? j0 : pt;                    // band position of the strip's last cell
    const int k1 = (pos - c + 1 > 0) ? pos - c + 1 : 0;  // strip = [k1, pos]; the shadow holds [k1, pos-1]
    const int n = pos - k1;
    double *band = is_row ? R : C;
    double *other = is_row ? C : R;
    if (e.deferred_update && k.pend_dir != -2) {  // livenote_v2.py:149-155
        k.run_count = (k.pend_dir == k.prev) ? k.run_count + 1 : 1;
        if (k.pend_dir != RTS_DIR_BOTH) k.prev = k.pend_dir;
        k.pend_dir = -2;
    }
    // four independent LDS reads (a scheduling barrier behind them, which makes it one wait instead of two, measured neutral)
    const double prev_raw = band[swz<W>(pos - (n > 0 ? 1 : 0))];
    const double d_raw = ex->d, smin_raw = ex->min;
    const int sidx_raw = ex->idx;
    const double d = rfl(d_raw);
    const double smin = rfl(smin_raw);
    const int sidx = __builtin_amdgcn_readfirstlane(sidx_raw);
    const double prev_s = rfl(prev_raw);  // unconditional: keeps the read in the round trip above (see otw_settle_hit_both)
    const double prev = (n > 0) ? prev_s : ((k1 > 0) ? sentinel : inf);
    // row hit: cell (pt, j0), "side" = up; column hit: cell (pt, jn), "side" = left -- in the chain's order
    // min(min(side + d, diag + 2d), previous cell of the strip + d)
    const double side = k.cA + d;
    const double diag = (pos > 0) ? (is_row ? k.cL : k.cU) + 2 * d : inf;
    const double cl = vmin(vmin(side, diag), prev + d);
    if (lane == 0) {
        band[swz<W>(pos)] = cl;
        other[swz<W>(is_row ? pt : jn)] = cl;  // column j0 gains row pt / row pt gains column jn
    }
    if (is_row) {
        k.rows += 1;
        k.consumed = pt + 1;
        k.cU = k.cA;
        k.cL = prev;
    } else {
        k.cols += 1;
        k.cL = k.cA;
        k.cU = prev;
    }
    k.cA = cl;
    k.cells += n + 1;
    OtwSettled o;
    o.row_fresh = is_row;
    o.col_fresh = !is_row;
    o.rf_min = o.cf_min = smin;
    o.rf_idx = o.cf_idx = sidx;
    o.row_corner = o.col_corner = true;
    o.rc = o.cc = cl;
    o.stop = false;
    return o;
}

// The barrier of the pipelined step loops.  The waves of a stream talk to each other through LDS only (bands, cost
// buffers, plan); what they send to global memory -- wave 0's path points, the dense mirror -- is read by nobody before
// the launch ends.  __syncthreads() would also drain those stores (a full release: vmcnt(0)) once per step.
// -DRTS_OTW_FULL_BARRIER restores it for A/B runs.
#ifdef RTS_OTW_FULL_BARRIER
#define RTS_STEP_BARRIER() __syncthreads()
#else
#define RTS_STEP_BARRIER() ::rts::lds_barrier()
#endif

template <int W, int NW, bool DENSE, typename RT, bool SPEC>
// (the throughput flavour exists for its residency -- three workgroups of eight waves per CU, i.e. at most 80 VGPRs: said to
// the compiler here, because the ILP-first scheduling this file is built with otherwise spends 87)
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(kThroughput<RT> ? 6 : 1)))
otw_advance_kernel(OtwArgs a) {
    static_assert(!SPEC || (NW >= 8 && !DENSE), "the pipelined kernel needs 8 waves and no dense mirror");
    constexpr int NT = 64 * NW;
    // Waves >= HW0 pre-compute the next step's cell costs while waves 0/1 run the chains (pipelined kernel:
    // wave 2 runs the speculative column strip).  With fewer than 4 waves there are no spare ones and every
    // wave does its share after its chain.
    constexpr int HW0 = SPEC ? 3 : (NW >= 4) ? 2 : 0;
    constexpr int NHELP = 64 * (NW - HW0);
#if defined(RTS_OTW_STAMPS) && RTS_OTW_STAMPS == 1
    long long stamp_sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int stamp_base = 0;
    long long stamp_last = (long long)__builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
#if defined(RTS_OTW_STAMPS) && RTS_OTW_STAMPS == 2
    long long lw_hit = 0, lb_hit = 0, lw_oth = 0, lb_oth = 0, ln_hit = 0, ln_oth = 0, l_last = 0;
    long long lo_work = 0, lo_t0 = 0;  // waves 1..7: own work in hit steps
    long long lo_rounds = 0;           // waves 1, 2: extra carry rounds of their speculative chains
#define RTS_ROUNDS_ACC (&lo_rounds)
#define RTS_LW_BEGIN() do { lo_t0 = (long long)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); } while (0)
#define RTS_LW_END(hit) do { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); if (hit) lo_work += t_ - lo_t0; } while (0)
#else
#define RTS_ROUNDS_ACC nullptr
#define RTS_LW_BEGIN() do { } while (0)
#define RTS_LW_END(hit) do { } while (0)
#endif
    extern __shared__ __align__(16) unsigned char smem_raw[];
    OtwLds<W, RT> &S = *reinterpret_cast<OtwLds<W, RT> *>(smem_raw);
    using LdsT = OtwLds<W, RT>;
    constexpr bool kNoLiveRing = !kHasLiveRing<RT>;  // live frames read from global memory by the helpers
    // no reference ring either: the pipelined kernel always, the others when they run without a live ring (dense mirror
    // of a window of 1024 cells or more) -- the struct is then allocated only up to `refw`
    constexpr bool kRefRing = !SPEC && !kNoLiveRing;
    constexpr size_t kSpecOff = (offsetof(LdsT, refw) + 15) & ~(size_t)15;  // SPEC: no reference ring
    OtwSpecLds<W> &SP = *reinterpret_cast<OtwSpecLds<W> *>(smem_raw + kSpecOff);  // only touched when SPEC

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably uniform -> scalar branches
    const double sentinel = (a.variant == RTS_VARIANT_OTW) ? 1e10 : (double)INFINITY;
    const double inf = INFINITY;
    OtwEnv e;
    e.b = blockIdx.x;
    e.lane = lane;
    e.c = a.c;
    e.N = a.N;
    e.live_cap = a.live_cap;
    e.euclid = a.cost_kind == RTS_COST_EUCLID;
    e.variant = a.variant;
    e.mode = a.mode;
    e.max_run_count = a.max_run_count;
    e.path_cap = a.path_cap;
    e.live_f64 = a.live_f64;
    e.ref_f64 = a.ref_f64;
    e.live_base = (long long)e.b * a.live_stride * kF;
    e.live = a.live;
    e.ref = a.ref;
    e.path = a.path;
    // LiveNote's set_live applies the run-count update at the bottom of its loop (livenote_v2.py:149-155)
    e.deferred_update = (a.mode == RTS_MODE_SET_LIVE) && (a.variant != RTS_VARIANT_OTW);
    const int c = a.c, N = a.N;
    int32_t *st = a.state + (size_t)e.b * RTS_STATE_LEN;
    int live_len_raw = a.live_len[e.b];
    if (a.clamp_len && live_len_raw > (int)a.live_stride) live_len_raw = (int)a.live_stride;
    if (live_len_raw < 0) live_len_raw = 0;
    e.live_len = live_len_raw;
    const int live_len = e.live_len;

    // every thread loads the state (uniform scalar loads); only wave 0 keeps it current
    OtwCtl k;
    k.t = st[RTS_ST_T];
    k.j = st[RTS_ST_J];
    k.dir = st[RTS_ST_DIRECTION];
    k.prev = st[RTS_ST_PREVIOUS];
    k.run_count = st[RTS_ST_RUN_COUNT];
    k.status = st[RTS_ST_STATUS];
    k.first = st[RTS_ST_FIRST_INSERT];
    k.n_path = st[RTS_ST_N_PATH];
    k.consumed = st[RTS_ST_CONSUMED];
    k.rows = st[RTS_ST_ROW_STRIPS];
    k.cols = st[RTS_ST_COL_STRIPS];
    k.truncated = st[RTS_ST_PATH_TRUNCATED];
    k.pend_dir = st[14];
    k.recomputes = st[RTS_ST_BAND_RECOMPUTES];
    k.cells = ((long long)(uint32_t)st[RTS_ST_CELLS_HI] << 32) | (uint32_t)st[RTS_ST_CELLS_LO];
    k.pending_col = 0;
    k.last_x = -1;
    k.last_y = -1;
    if (k.n_path > 0 && k.n_path <= a.path_cap) {
        const int32_t *pp = a.path + ((size_t)e.b * a.path_cap + (k.n_path - 1)) * 2;
        k.last_x = pp[0];
        k.last_y = pp[1];
    }
    k.rb_min = inf;
    k.cb_min = inf;
    k.rb_idx = 0;
    k.cb_idx = 0;
    k.pfl0 = k.pfl1 = k.pfr0 = k.pfr1 = 0.0;
    k.spec_valid = 0;
    k.ri = k.ci = 0;
    k.cA = k.cU = k.cL = inf;

    if (k.status == RTS_STOP_REF_END) return;  // sticky; the reference's callers stop inserting
    if (k.status == RTS_LIVE_OVERFLOW) {       // otw_eran.py:50-55: t keeps counting inserts
        if (tid == 0 && live_len > k.consumed) {
            st[RTS_ST_T] = live_len - 1;
            st[RTS_ST_CONSUMED] = live_len;
        }
        return;
    }
    if (live_len <= k.consumed) return;  // nothing new

    // ---- prologue: windows (synchronous fill by all threads), persisted bands, first frame
    for (int i = tid; i < OtwLds<W, RT>::SWZ; i += NT) {  // no uninitialised LDS ever reaches the arithmetic
        S.R[i] = 0.0;
        S.C[i] = 0.0;
        S.Dr[0][i] = S.Dr[1][i] = S.Dc[0][i] = S.Dc[1][i] = 0.0;
        if (SPEC) SP.ShR[i] = SP.ShC[i] = 0.0;
    }
    __syncthreads();
    {
        const int lo_l = (k.t - c + 1 > 0) ? k.t - c + 1 : 0;
        const int lo_r = (k.j - c + 1 > 0) ? k.j - c + 1 : 0;
        constexpr int kAhead = SPEC ? 3 : 2;  // frames the first step's cost computations reach past (t, j)
        k.live_hi = (k.t + kAhead < live_len - 1) ? k.t + kAhead : live_len - 1;
        k.ref_hi = (k.j + kAhead < N - 1) ? k.j + kAhead : N - 1;
        if constexpr (!kNoLiveRing) {
            for (int idx = tid; idx < (k.live_hi - lo_l + 1) * kF; idx += NT) {
                const int fr = lo_l + idx / kF, f = idx % kF;
                S.livew[f][fr & (W - 1)] = (typename RingElem<RT>::type)otw_load_feat(a.live, a.live_f64, e.live_base + (long long)fr * kF + f);
            }
        }
        if constexpr (kRefRing) {
            for (int idx = tid; idx < (k.ref_hi - lo_r + 1) * kF; idx += NT) {
                const int fr = lo_r + idx / kF, f = idx % kF;
                S.refw[f][fr & (W - 1)] = (typename RingElem<RT>::type)otw_load_feat(a.ref, a.ref_f64, (long long)fr * kF + f);
            }
        }
        if (!k.first) {
            const double *bb = a.bands + (size_t)e.b * 2 * (c + 1);
            for (int i = tid; i <= c; i += NT) {
                const int y = k.j - c + i, x = k.t - c + i;
                if (y >= 0) S.R[swz<W>(y)] = bb[i];
                if (x >= 0) S.C[swz<W>(x)] = bb[(c + 1) + i];
            }
        }
        __syncthreads();
    }
    if (wave == (SPEC ? HW0 : 0)) {  // the wave that will refill the rings starts its first prefetch
        if constexpr (!kNoLiveRing) otw_prefetch_live(k, e);
        if constexpr (kRefRing) otw_prefetch_ref(k, e);
    }
    if (wave == 0) {
        if (k.first) {
            double lf[kF], rf[kF];
            if constexpr (kNoLiveRing) {
                otw_live_frame(e, 0, lf);
            } else {
#pragma unroll
                for (int f = 0; f < kF; f++) lf[f] = (double)S.livew[f][0];
            }
            otw_ref_frame(e, 0, rf);
            const double d = rfl(cell_cost(lf, rf, e.euclid));
            if (lane == 0) {
                S.R[swz<W>(0)] = d;
                S.C[swz<W>(0)] = d;
                if (DENSE) {
                    a.dense_acc[(long long)e.b * a.live_cap * N] = d;
                    a.dense_cost[(long long)e.b * a.live_cap * N] = d;
                }
            }
            k.first = 0;
            k.consumed = 1;
            k.cells += 1;
            k.t = 0;
            k.j = 0;
            k.pend_dir = -2;
            k.rb_min = d;
            k.cb_min = d;
            k.rb_idx = 0;
            k.cb_idx = 0;
            __builtin_amdgcn_wave_barrier();
            if (a.mode == RTS_MODE_SET_LIVE)
                otw_decide<W, RT>(S.R, S.C, k, e, 0, 0, false, 0.0, 0, false, 0.0, 0, false, 0.0, false, 0.0, true);
        } else {  // band minima are not persisted: rebuild them from the reloaded bands
            for (int which = 0; which < 2; which++) {
                const int hi = which ? k.t : k.j;
                double vm;
                int im;
                band_argmin<W>(which ? S.C : S.R, (hi - c + 1 > 0) ? hi - c + 1 : 0, hi, lane, vm, im);
                if (which) {
                    k.cb_min = vm;
                    k.cb_idx = im;
                } else {
                    k.rb_min = vm;
                    k.rb_idx = im;
                }
            }
        }
        if (lane == 0) {
            S.t = k.t;
            S.j = k.j;
        }
    }
    __syncthreads();
    // prime the cost buffers for the first step (all threads), then publish its plan
    int buf = 0;
    OtwPlan pl;  // pipelined kernel, wave 0: the plan it published, kept in registers
    pl.t = pl.j0 = 0;
    pl.flags = kPlanExit;
    if constexpr (SPEC) {
        otw_costs_prime<W, RT>(S, e, S.t, S.j, tid, NT);
        if (wave == 0) {
            k.cA = rfl(S.R[swz<W>(k.j)]);
            k.cU = (k.t > 0) ? rfl(S.C[swz<W>(k.t - 1)]) : inf;
            k.cL = (k.j > 0) ? rfl(S.R[swz<W>(k.j - 1)]) : inf;
            pl = otw_make_plan<W, RT>(S, k, e, true);
        }
    } else {
        otw_precompute<W, RT>(S, e, S.t, S.j, S.Dr[0], S.Dc[0], tid, NT);
        if (wave == 0) otw_make_plan<W, RT>(S, k, e);
    }
    __syncthreads();

    int sp = 0;  // pipelined kernel: step parity (plan slot read this step; speculation results go to [sp ^ 1])
#ifdef RTS_OTW_PRIO_W0
    // experiment (profiles/experiments/README.md): the waves that pace a step -- control wave 0, chain waves 1 / 2 -- each
    // share a SIMD with a helper wave; a static priority lets them win every issue arbitration
    if constexpr (SPEC) {
        if (wave == 0) __builtin_amdgcn_s_setprio(RTS_OTW_PRIO_W0);
        if (wave == 1 || wave == 2) __builtin_amdgcn_s_setprio(RTS_OTW_PRIO_W12);
    }
#endif
    if constexpr (SPEC) {
        // ---- pipelined step loops.  While wave 0 runs the control work of a step, wave 1 already runs the row strip
        // the next step needs if it turns out Row-only and wave 2 the column strip it needs if it turns out
        // Column-only -- each minus its last cell, which depends on the corner wave 0 is computing -- into the shadow
        // buffers, and the helpers extend the cost buffers one row / column further out.  The next Row-only or
        // Column-only step is then a *hit*: its shadow becomes the band (the plan flips the buffer index), wave 0
        // finishes the last cell from registers and goes straight to decide(), and the other waves start the next
        // speculation at once -- no wave reads anything a hit step's wave 0 writes (the guard in strip_chain), so a hit
        // step has a single barrier, at its end.  Both steps, stop steps and the first step of a launch run their
        // chains first, as in the plain kernel, then speculate during their control phase.  Values are bit-identical
        // either way: a speculative strip is the same chain on the same inputs, the last cell the chain's own expression.
        if (wave >= HW0) {
            for (;;) {
                const int pt = __builtin_amdgcn_readfirstlane(S.plan_t[sp]), j0 = __builtin_amdgcn_readfirstlane(S.plan_j0[sp]);
                const int pflags = otw_resolve_plan(__builtin_amdgcn_readfirstlane(S.plan_flags[sp]), pt, c, &SP.col[sp]);
                if (pflags & kPlanExit) break;
                const bool do_row = (pflags & kPlanRow) != 0, do_col = (pflags & kPlanCol) != 0;
                RTS_LW_BEGIN();
                // rings: frames the *next* step's cost work can reach (rows <= pt+3, columns <= jn+3); what this step
                // reads was made sure of one step ago, and the slots written now are not among it (W >= c + 12)
                if constexpr (!kNoLiveRing)
                    if (wave == HW0) otw_refill<W, RT>(S, k, e, pt + 3, 0, false);  // the reference has no ring here
                if (!(pflags & kPlanHit)) RTS_STEP_BARRIER();  // this step's chains have read their cost buffers
                if (!(pflags & kPlanStop))
                    otw_costs_advance<W, RT>(S, e, pt, j0 + (do_col ? 1 : 0), do_row, do_col, tid - 64 * HW0, NHELP);
                RTS_LW_END(pflags & kPlanHit);
                RTS_STEP_BARRIER();
                sp ^= 1;
            }
        } else if (wave == 1) {
            for (;;) {
                const int pt = __builtin_amdgcn_readfirstlane(S.plan_t[sp]), j0 = __builtin_amdgcn_readfirstlane(S.plan_j0[sp]);
                const int pflags = otw_resolve_plan(__builtin_amdgcn_readfirstlane(S.plan_flags[sp]), pt, c, &SP.col[sp]);
                if (pflags & kPlanExit) break;
                const int jn = j0 + ((pflags & kPlanCol) ? 1 : 0);
                double *R = (pflags & kPlanRi) ? SP.ShR : S.R, *Rsh = (pflags & kPlanRi) ? S.R : SP.ShR;
                RTS_LW_BEGIN();
                if (!(pflags & kPlanHit)) {
                    if ((pflags & (kPlanRow | kPlanCol | kPlanStop)) == (kPlanRow | kPlanCol))  // Both step
                        otw_col_strip<W, DENSE, RT>(S, (pflags & kPlanCi) ? SP.ShC : S.C, a, e, S.Dc[(j0 + 1) & 1], pt,
                                                    j0 + 1, true, sentinel);
                    RTS_STEP_BARRIER();
                }
                if ((pflags & (kPlanHit | kPlanRow | kPlanCol)) == (kPlanHit | kPlanRow | kPlanCol)) {
                    // Both step as a hit: the row's last cell (pt, j0) is input of the next speculation; wave 0 writes
                    // the same value.  Its predecessors sit in the old band, which is this wave's output buffer.
                    const double av = otw_last_cell<W>(R, j0, (j0 - c + 1 > 0) ? j0 - c + 1 : 0, Rsh[swz<W>(j0)],
                                                       Rsh[swz<W>(j0 > 0 ? j0 - 1 : 0)], SP.row[sp].d, sentinel);
                    if (lane == 0) R[swz<W>(j0)] = av;
                }
                if (!(pflags & kPlanStop) && pt + 1 < live_len && pt + 1 < a.live_cap)  // row pt+1 over [.., jn-1]
                    otw_spec_strip<W, false>(S.Dr[(pt + 1) & 1], R, Rsh, jn, c, lane, sentinel, &SP.row[sp ^ 1], RTS_ROUNDS_ACC);
                RTS_LW_END(pflags & kPlanHit);
                RTS_STEP_BARRIER();
                sp ^= 1;
            }
        } else if (wave == 2) {
            for (;;) {
                const int pt = __builtin_amdgcn_readfirstlane(S.plan_t[sp]), j0 = __builtin_amdgcn_readfirstlane(S.plan_j0[sp]);
                const int pflags = otw_resolve_plan(__builtin_amdgcn_readfirstlane(S.plan_flags[sp]), pt, c, &SP.col[sp]);
                if (pflags & kPlanExit) break;
                const int jn = j0 + ((pflags & kPlanCol) ? 1 : 0);
                double *C = (pflags & kPlanCi) ? SP.ShC : S.C, *Csh = (pflags & kPlanCi) ? S.C : SP.ShC;
                RTS_LW_BEGIN();
                if (!(pflags & kPlanHit)) {
                    if constexpr (kThroughput<RT>) {
                        // this step's own strip (the row strip of a Both step), which wave 0 runs in the other flavours
                        const bool stop2 = (pflags & kPlanStop) != 0;
                        if (pflags & kPlanRow) {
                            double rf_min = inf;
                            int rf_idx = 0x7fffffff;
                            otw_row_strip<W, DENSE, RT>((pflags & kPlanRi) ? SP.ShR : S.R, a, e, S.Dr[pt & 1], pt, j0, jn, sentinel,
                                                        rf_min, rf_idx);
                            if (lane == 0) {
                                S.rfresh_min = rf_min;
                                S.rfresh_idx = rf_idx;
                            }
                        } else if ((pflags & kPlanCol) && !stop2) {
                            otw_col_strip<W, DENSE, RT>(S, C, a, e, S.Dc[jn & 1], pt, jn, false, sentinel);
                        }
                    }
                    RTS_STEP_BARRIER();
                }
                bool drop = false;
                if ((pflags & (kPlanHit | kPlanRow | kPlanCol)) == (kPlanHit | kPlanRow | kPlanCol)) {
                    // Both step as a hit: the column's last cell (pt-1, jn), as on wave 1
                    const int t0 = pt - 1;
                    // (the Both step's column strip starts at max(0, pt-c+1): one cell later than the shadow once the band is full)
                    const double bv = otw_last_cell<W>(C, t0, (pt - c + 1 > 0) ? pt - c + 1 : 0, Csh[swz<W>(t0)],
                                                       Csh[swz<W>(t0 > 0 ? t0 - 1 : 0)], SP.col[sp].d, sentinel);
                    if (lane == 0) C[swz<W>(t0)] = bv;
                    drop = pt >= c;  // wave 0 is writing the sentinel into position pt-c, this strip's diagonal input
                }
                if (!(pflags & kPlanStop) && jn + 1 < N)  // column jn+1 over rows [.., pt-1]
                    otw_spec_strip<W, true>(S.Dc[(jn + 1) & 1], C, Csh, pt, c, lane, sentinel, &SP.col[sp ^ 1], RTS_ROUNDS_ACC, drop);
                RTS_LW_END(pflags & kPlanHit);
                RTS_STEP_BARRIER();
                sp ^= 1;
            }
        } else {
            // wave 0.  Diagnostic stamps: slots 0..5 = hit steps, 8..13 = other steps
            // (0 end-of-step barrier wait, 1 chains, 2 mid-step barrier wait, 3 settle, 4 decide, 5 plan); 6 / 14 = counts
#if defined(RTS_OTW_STAMPS) && RTS_OTW_STAMPS == 2
            l_last = (long long)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
            for (;;) {
                RTS_STAMP2(0);
                // wave 0 wrote the plan itself; pinned to SGPRs so that the step's addressing and branches are scalar
                const int pt = __builtin_amdgcn_readfirstlane(pl.t), j0 = __builtin_amdgcn_readfirstlane(pl.j0);
                pl.flags = __builtin_amdgcn_readfirstlane(pl.flags);
                const int pflags = otw_resolve_plan(pl.flags, pt, c, &SP.col[sp]);
                if (pflags & kPlanExit) break;
                if ((pl.flags & kPlanHitIf) && (pflags & kPlanHit)) {  // resolved to a hit: both shadows become bands
                    k.ri ^= 1;
                    k.ci ^= 1;
                }
                const bool do_row = (pflags & kPlanRow) != 0, do_col = (pflags & kPlanCol) != 0;
                const bool stop = (pflags & kPlanStop) != 0;
                const int jn = j0 + (do_col ? 1 : 0);
                double *R = (pflags & kPlanRi) ? SP.ShR : S.R, *C = (pflags & kPlanCi) ? SP.ShC : S.C;
                OtwSettled o;
#if defined(RTS_OTW_STAMPS) && RTS_OTW_STAMPS == 1
                stamp_base = (pflags & kPlanHit) ? 0 : 8;
                if (stamp_base) stamp_sum[14] += 1; else stamp_sum[6] += 1;
#endif
                if (pflags & kPlanHit) RTS_STAMP(1);  // (diagnostic: the step's entry -- plan words, buffer selection -- apart from settle)
                if ((pflags & kPlanHit) && do_row && do_col) {
                    o = otw_settle_hit_both<W, RT>(R, C, &SP.row[sp], &SP.col[sp], k, e, pt, j0, sentinel);
                } else if (pflags & kPlanHit) {
                    o = otw_settle_hit<W, RT>(R, C, do_row ? &SP.row[sp] : &SP.col[sp], k, e, pt, j0, do_row, sentinel);
                } else {
                    double rf_min = inf;
                    int rf_idx = 0x7fffffff;
                    if constexpr (!kThroughput<RT>) {
                        if (do_row)
                            otw_row_strip<W, DENSE, RT>(R, a, e, S.Dr[pt & 1], pt, j0, jn, sentinel, rf_min, rf_idx);
                        else if (do_col && !stop)
                            otw_col_strip<W, DENSE, RT>(S, C, a, e, S.Dc[jn & 1], pt, jn, false, sentinel);
                    }
                    RTS_STAMP(9);
                    RTS_STEP_BARRIER();
                    RTS_STAMP(10);
                    if constexpr (kThroughput<RT>) {  // the row chain ran on wave 2
                        if (do_row) {
                            rf_min = rfl(S.rfresh_min);
                            rf_idx = __builtin_amdgcn_readfirstlane(S.rfresh_idx);
                        }
                    }
                    // the three band slots the next hit step's last cell depends on: acc[pt][jn] is what settle() is about
                    // to write, the other two are not touched by it and are read in the same round trip as its inputs
                    const double ld_u = C[swz<W>(pt > 0 ? pt - 1 : 0)], ld_l = R[swz<W>(jn > 0 ? jn - 1 : 0)];
                    o = otw_settle<W, DENSE, RT>(S, R, C, k, a, e, pt, j0, pflags, rf_min, rf_idx, sentinel);
                    k.cA = o.cc;
                    k.cU = (pt > 0) ? rfl(ld_u) : inf;
                    k.cL = (jn > 0) ? rfl(ld_l) : inf;
                }
                RTS_STAMP2(3);
                if (o.stop) {
                    k.status = RTS_STOP_REF_END;
                    k.t = pt;
                    k.j = jn;
                    k.pending_col = 0;
                } else {
                    otw_decide<W, RT>(R, C, k, e, pt, jn, o.row_fresh, o.rf_min, o.rf_idx, o.col_fresh, o.cf_min, o.cf_idx,
                                      o.row_corner, o.rc, o.col_corner, o.cc, false);
                }
                RTS_STAMP2(4);
                k.spec_valid = !o.stop;  // the shadows being computed during this step belong to (k.t, k.j)
                pl = otw_make_plan<W, RT>(S, k, e, true, sp ^ 1);
                RTS_STAMP2(5);
#if defined(RTS_OTW_STAMPS) && RTS_OTW_STAMPS == 2
                {
                    const long long t0_ = (long long)__builtin_amdgcn_s_memtime();
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    RTS_STEP_BARRIER();
                    const long long t1_ = (long long)__builtin_amdgcn_s_memtime();
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    if (pflags & kPlanHit) {
                        lw_hit += t0_ - l_last;
                        lb_hit += t1_ - t0_;
                        ln_hit += 1;
                    } else {
                        lw_oth += t0_ - l_last;
                        lb_oth += t1_ - t0_;
                        ln_oth += 1;
                    }
                    l_last = t1_;
                }
#else
                RTS_STEP_BARRIER();
#endif
                sp ^= 1;
            }
        }
    } else if constexpr (NW >= 4) {
        // ---- role-specialised step loops.  Every wave runs only its own role's code between the two barriers of
        // a step (a step = one row strip and/or one column strip + one decide()), so each loop keeps only its own
        // values live: helpers pre-compute costs, wave 1 runs the column strip of Both steps, wave 0 runs the
        // (other) strip and then the control phase.  All of them read the plan wave 0 published.
        if (wave >= HW0) {
            for (;;) {
                const int pt = __builtin_amdgcn_readfirstlane(S.plan_t[0]), j0 = __builtin_amdgcn_readfirstlane(S.plan_j0[0]),
                          pflags = __builtin_amdgcn_readfirstlane(S.plan_flags[0]);
                if (pflags & kPlanExit) break;
                const int jn = j0 + ((pflags & kPlanCol) ? 1 : 0);
                if (!(pflags & kPlanStop))
                    otw_precompute<W, RT>(S, e, pt, jn, S.Dr[buf ^ 1], S.Dc[buf ^ 1], tid - 64 * HW0, NHELP);
                __syncthreads();  // strips and next costs complete
                __syncthreads();  // next plan published
                buf ^= 1;
            }
        } else if (wave == 1) {
            for (;;) {
                const int pt = __builtin_amdgcn_readfirstlane(S.plan_t[0]), j0 = __builtin_amdgcn_readfirstlane(S.plan_j0[0]),
                          pflags = __builtin_amdgcn_readfirstlane(S.plan_flags[0]);
                if (pflags & kPlanExit) break;
                if ((pflags & (kPlanRow | kPlanCol | kPlanStop)) == (kPlanRow | kPlanCol))  // Both step
                    otw_col_strip<W, DENSE, RT>(S, S.C, a, e, S.Dc[buf], pt, j0 + 1, true, sentinel);
                __syncthreads();
                __syncthreads();
                buf ^= 1;
            }
        } else {
            for (;;) {
                RTS_STAMP(0);
                const int pt = __builtin_amdgcn_readfirstlane(S.plan_t[0]), j0 = __builtin_amdgcn_readfirstlane(S.plan_j0[0]),
                          pflags = __builtin_amdgcn_readfirstlane(S.plan_flags[0]);
                if (pflags & kPlanExit) break;
                const bool do_row = (pflags & kPlanRow) != 0, do_col = (pflags & kPlanCol) != 0;
                const bool stop = (pflags & kPlanStop) != 0;
                const int jn = j0 + (do_col ? 1 : 0);
                double rf_min = inf;
                int rf_idx = 0x7fffffff;
                if (do_row)
                    otw_row_strip<W, DENSE, RT>(S.R, a, e, S.Dr[buf], pt, j0, jn, sentinel, rf_min, rf_idx);
                else if (do_col && !stop)
                    otw_col_strip<W, DENSE, RT>(S, S.C, a, e, S.Dc[buf], pt, jn, false, sentinel);
                RTS_STAMP(3);
                __syncthreads();
                RTS_STAMP(5);
                const OtwSettled o = otw_settle<W, DENSE, RT>(S, S.R, S.C, k, a, e, pt, j0, pflags, rf_min, rf_idx, sentinel);
                otw_finish<W, RT>(S, k, e, pt, jn, o);
                RTS_STAMP(8);
                __syncthreads();
                buf ^= 1;
            }
        }
    } else {
        // ---- step loop: one iteration = one row strip and/or one column strip + one decide()
        for (;;) {
            RTS_STAMP(0);
            const int pt = __builtin_amdgcn_readfirstlane(S.plan_t[0]), j0 = __builtin_amdgcn_readfirstlane(S.plan_j0[0]),
                      pflags = __builtin_amdgcn_readfirstlane(S.plan_flags[0]);
            if (pflags & kPlanExit) break;
            const bool do_row = (pflags & kPlanRow) != 0, do_col = (pflags & kPlanCol) != 0;
            const bool stop = (pflags & kPlanStop) != 0;
            const int jn = j0 + (do_col ? 1 : 0);
            const int k1r = (j0 - c + 1 > 0) ? j0 - c + 1 : 0, nr = j0 - k1r + 1;  // row strip: columns
            const int k1c = (pt - c + 1 > 0) ? pt - c + 1 : 0, nc = pt - k1c + 1;  // column strip: rows
            const bool col_active = do_col && !stop;
            double *Dr = S.Dr[buf], *Dc = S.Dc[buf];

            // -- chain phase: row strip on wave 0, column strip on wave 1; spare waves pre-compute the
            //    next step's costs meanwhile
            const int col_wave = (NW > 1 && do_row) ? 1 : 0;
            double rf_min = inf;
            int rf_idx = 0x7fffffff;
            if (do_row && wave == 0) {
                const double x_in = (k1r > 0) ? sentinel : inf;  // (t, k1r-1) was never evaluated
                const int lo_arg = (jn - c + 1 > 0) ? jn - c + 1 : 0;  // row band's lower end at decide()
                const long long dro = ((long long)e.b * a.live_cap + pt) * N + k1r;  // cell (pt, k1r)
                strip_chain<W, DENSE>(Dr, S.R, S.R, k1r, nr, x_in, lane, lo_arg, rf_min, rf_idx, DENSE ? a.dense_acc + dro : nullptr,
                                      DENSE ? a.dense_cost + dro : nullptr, 1);
                if (lane == 0 && k1r > 0) S.R[swz<W>(k1r - 1)] = sentinel;
            }
            if (col_active && wave == col_wave) {
                const double x_in = (k1c > 0) ? sentinel : inf;  // (k1c-1, jn) was never evaluated
                const int ncc = nc - (do_row ? 1 : 0);            // corner cell waits for the row strip
                double fm;
                int fi;
                // the corner's diagonal term needs column jn-1 at row t-1, which the chain is about to overwrite
                const double dcorner = Dc[swz<W>(pt)];
                const double pa = (do_row && pt > 0) ? S.C[swz<W>(pt - 1)] + 2 * dcorner : inf;
                const long long dco = ((long long)e.b * a.live_cap + k1c) * N + jn;  // cell (k1c, jn)
                strip_chain<W, DENSE>(Dc, S.C, S.C, k1c, ncc, x_in, lane, k1c, fm, fi, DENSE ? a.dense_acc + dco : nullptr,
                                      DENSE ? a.dense_cost + dco : nullptr, N);
                if (lane == 0) {
                    S.corner_pa = pa;
                    S.corner_d = dcorner;
                    if (k1c > 0) S.C[swz<W>(k1c - 1)] = sentinel;
                    S.cfresh_min = fm;
                    S.cfresh_idx = fi;
                }
            }
            RTS_STAMP(3);
            if (!stop && wave >= HW0)
                otw_precompute<W, RT>(S, e, pt, jn, S.Dr[buf ^ 1], S.Dc[buf ^ 1], tid - 64 * HW0, NHELP);
            RTS_STAMP(4);
            __syncthreads();
            RTS_STAMP(5);

            // -- corner fix-up, decide, next plan (wave 0, register state)
            if (wave == 0) {
                if (!stop && e.deferred_update && k.pend_dir != -2) {  // livenote_v2.py:149-155
                    k.run_count = (k.pend_dir == k.prev) ? k.run_count + 1 : 1;
                    if (k.pend_dir != RTS_DIR_BOTH) k.prev = k.pend_dir;
                    k.pend_dir = -2;
                }
                if (do_row) {
                    k.rows += 1;
                    k.cells += nr;
                    k.consumed = pt + 1;
                }
                double cl = 0.0, cf_min = inf;
                int cf_idx = 0x7fffffff;
                // last cell of the row strip = acc[t][j0]; of the column chain = acc[t-1][jn] (Both) or acc[t][jn]
                const double row_last = do_row ? rfl(S.R[swz<W>(j0)]) : 0.0;
                if (do_row && !col_active && lane == 0) S.C[swz<W>(pt)] = row_last;  // column j0 gains row t
                if (col_active) {
                    const int ncc = nc - (do_row ? 1 : 0);
                    cl = (ncc > 0) ? rfl(S.C[swz<W>(k1c + ncc - 1)]) : ((k1c > 0) ? sentinel : inf);
                    cf_min = rfl(S.cfresh_min);
                    cf_idx = __builtin_amdgcn_readfirstlane(S.cfresh_idx);
                    if (do_row) {
                        const double d = rfl(S.corner_d);
                        const double av = vmin(row_last + d, rfl(S.corner_pa));
                        cl = vmin(av, cl + d);  // cl was the value of (t-1, jn), or the sentinel carry
                        if (lane == 0) {
                            S.C[swz<W>(pt)] = cl;
                            if (DENSE) {
                                const long long o = ((long long)e.b * a.live_cap + pt) * N + jn;
                                a.dense_acc[o] = cl;
                                a.dense_cost[o] = d;
                            }
                        }
                    }
                    if (lane == 0) S.R[swz<W>(jn)] = cl;  // row t gains column jn
                    k.cols += 1;
                    k.cells += nc;
                }
                RTS_STAMP(6);
                if (stop) {
                    k.status = RTS_STOP_REF_END;
                    k.t = pt;
                    k.j = jn;
                    k.pending_col = 0;
                } else {
                    // row band: fresh from this step's row strip, plus the corner a column strip appended;
                    // column band: fresh from this step's column strip (its corner cell is outside the
                    // chain), or the old band plus the row strip's last cell
                    otw_decide<W, RT>(S.R, S.C, k, e, pt, jn, do_row, rf_min, rf_idx, col_active, cf_min, cf_idx, col_active, cl,
                                  do_row, col_active ? cl : row_last, false);
                }
                RTS_STAMP(7);
                otw_make_plan<W, RT>(S, k, e);
            }
            RTS_STAMP(8);
            __syncthreads();
            buf ^= 1;
        }
    }
    __syncthreads();

    // ---- epilogue: persist bands + state
    {
        int te = S.t, je = S.j;
        const int t_state = te;
        if (te > a.live_cap - 1) te = a.live_cap - 1;
        if (je > N - 1) je = N - 1;
        double *bb = a.bands + (size_t)e.b * 2 * (c + 1);
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        const int fl = S.plan_flags[sp];  // pipelined kernel: which buffer of each pair ended up holding the band
        const double *Rf = (SPEC && (fl & kPlanRi)) ? SP.ShR : S.R, *Cf = (SPEC && (fl & kPlanCi)) ? SP.ShC : S.C;
        // set_live that ran out of live frames stops with t one past the last frame (otw_eran.py:113-116): row t was
        // never evaluated and reads as the matrix's initial value
        const bool row_missing = (t_state >= e.live_len) && (t_state <= a.live_cap - 1);
        const double sentinel = (e.variant == RTS_VARIANT_OTW) ? 1e10 : (double)INFINITY;
        for (int i = tid; i <= c; i += NT) {
            const int y = je - c + i, x = te - c + i;
            bb[i] = (y >= 0) ? (row_missing ? sentinel : Rf[swz<W>(y)]) : qnan;
            bb[(c + 1) + i] = (x >= 0 && x <= t_state) ? ((row_missing && x == t_state) ? sentinel : Cf[swz<W>(x)]) : qnan;
        }
    }
#if defined(RTS_OTW_STAMPS) && RTS_OTW_STAMPS == 1
    if (tid == 0 && a.debug)
        for (int i = 0; i < 16; i++) a.debug[(size_t)e.b * 16 + i] = stamp_sum[i];
#endif
#if defined(RTS_OTW_STAMPS) && RTS_OTW_STAMPS == 2
    if (tid == 0 && a.debug) {
        long long *dbg = a.debug + (size_t)e.b * 16;
        dbg[0] = lw_hit, dbg[1] = lb_hit, dbg[2] = ln_hit, dbg[3] = lw_oth, dbg[4] = lb_oth, dbg[5] = ln_oth;
    }
    if (a.debug && lane == 0 && wave >= 1 && wave <= 7) a.debug[(size_t)e.b * 16 + 5 + wave] = lo_work;
    if (a.debug && lane == 0 && wave >= 1 && wave <= 2) a.debug[(size_t)e.b * 16 + 12 + wave] = lo_rounds;
#endif
    if (tid == 0) {
        st[RTS_ST_T] = k.t;
        st[RTS_ST_J] = k.j;
        // LiveNote's set_live keeps the direction in a local; self.direction stays "both"
        st[RTS_ST_DIRECTION] = e.deferred_update ? RTS_DIR_BOTH : k.dir;
        st[RTS_ST_PREVIOUS] = k.prev;
        st[RTS_ST_RUN_COUNT] = k.run_count;
        st[RTS_ST_STATUS] = k.status;
        st[RTS_ST_FIRST_INSERT] = k.first;
        st[RTS_ST_N_PATH] = k.n_path;
        st[RTS_ST_CONSUMED] = k.consumed;
        st[RTS_ST_ROW_STRIPS] = k.rows;
        st[RTS_ST_COL_STRIPS] = k.cols;
        st[RTS_ST_CELLS_LO] = (int32_t)(uint32_t)(k.cells & 0xffffffffLL);
        st[RTS_ST_CELLS_HI] = (int32_t)(uint32_t)((unsigned long long)k.cells >> 32);
        st[RTS_ST_PATH_TRUNCATED] = k.truncated;
        st[14] = k.pend_dir;
        st[RTS_ST_BAND_RECOMPUTES] = k.recomputes;
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

__global__ void otw_fill_kernel(double *p, long long n, double v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long q = i; q < n; q += stride) p[q] = v;
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

// Append n_new[b] (or n_uniform) frames from frames [B][n_max][F] to the history of stream b.
__global__ void otw_append_many_kernel(double *hist, int32_t *hist_len, const void *frames, int frames_f64,
                                       const int32_t *n_new, int n_uniform, int n_max, int B, int cap) {
    const int b = blockIdx.x;
    if (b >= B) return;
    int nn = n_new ? n_new[b] : n_uniform;
    nn = nn < 0 ? 0 : (nn > n_max ? n_max : nn);  // never read a neighbouring stream's frames
    const int base = hist_len[b];
    __syncthreads();
    for (int idx = threadIdx.x; idx < nn * kF; idx += blockDim.x) {
        const int fr = base + idx / kF;
        if (fr < cap) {
            const size_t src = (size_t)b * n_max * kF + idx;
            hist[((size_t)b * cap + fr) * kF + idx % kF] =
                frames_f64 ? reinterpret_cast<const double *>(frames)[src]
                           : (double)reinterpret_cast<const float *>(frames)[src];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) hist_len[b] = base + (nn > 0 ? nn : 0);
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
    double *dense_acc, *dense_cost;  // caller-owned, optional
    int spec;           // 1: pipelined kernel (needs 8 waves and no dense mirror); 0: plain kernel
    int device;         // the HIP device the handle's buffers live on (one handle = one device)
    int cus;            // its compute units
    int tp_from;        // batches of at least tp_from streams per CU take the residency-oriented kernel flavour
    const void *attr_fn[8];  // kernel instantiations whose dynamic-LDS limit is already raised on `device`
    // what the handle has consumed since the last reset, for rts_otw_replay_dense
    int src_kind;       // 0 nothing, 1 the buffers of the last rts_otw_run, 2 the handle-owned history, 3 mixed
    int run_dtype, run_T, run_mode;  // of the last rts_otw_run (the buffers themselves are the caller's and not remembered)
    int32_t *rp_state, *rp_path;     // scratch state of rts_otw_replay_dense, allocated on first use
    double *rp_bands;
};

namespace rts {

// The dynamic-LDS limit is a per-device function attribute: remembered per handle (a handle lives on one device),
// not per process.
static int ensure_lds_attr(rts_otw *h, const void *fn, size_t smem) {
    for (int k = 0; k < 8; k++)
        if (h->attr_fn[k] == fn) return RTS_OK;
    RTS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    for (int k = 0; k < 8; k++)
        if (!h->attr_fn[k]) {
            h->attr_fn[k] = fn;
            break;
        }
    return RTS_OK;
}

template <int W, int NW, bool DENSE, typename RT, bool SPEC>
static int launch_advance_d(rts_otw *h, const OtwArgs &args, int B, hipStream_t s) {
    using LdsT = OtwLds<W, RT>;
    const size_t smem = SPEC ? ((offsetof(LdsT, refw) + 15) & ~(size_t)15) + sizeof(OtwSpecLds<W>)
                             : (kHasLiveRing<RT> ? sizeof(LdsT) : ((offsetof(LdsT, refw) + 15) & ~(size_t)15));
    const int rc = ensure_lds_attr(h, reinterpret_cast<const void *>(&otw_advance_kernel<W, NW, DENSE, RT, SPEC>), smem);
    if (rc != RTS_OK) return rc;
    hipLaunchKernelGGL((otw_advance_kernel<W, NW, DENSE, RT, SPEC>), dim3(B), dim3(64 * NW), smem, s, args);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

// The dense mirror is a separate instantiation so that the default kernel carries none of its code.
template <int W, int NW>
static int launch_advance(rts_otw *h, const OtwArgs &args, int B, hipStream_t s) {
    if constexpr (W >= 1024) {
        // no ring fits beside a 1024-cell window's bands: every frame is read from global memory.  The dense mirror
        // (otw_eran.py:23,27) runs the plain role-specialised kernel that way, in its 8-wave form whatever rts_otw_set_waves says
        if (args.dense_acc) return launch_advance_d<W, 8, true, LiveFromGlobal, false>(h, args, B, s);
        if constexpr (NW >= 8) {
            if (args.spec) return launch_advance_d<W, NW, false, LiveFromGlobal, true>(h, args, B, s);
        }
        return set_error(RTS_ERR_UNSUPPORTED, "band widths above 500 (c=%d) run only on the pipelined 8-wave kernel", h->c);
    } else {  // (an else branch, so that none of the ring kernels below is instantiated for the wide windows)
        if (args.dense_acc) return launch_advance_d<W, NW, true, double, false>(h, args, B, s);
        // float32 rings only when both inputs are float32: every value then widens back exactly
        const bool f32 = !args.ref_f64 && !args.live_f64;
        if constexpr (NW >= 8) {
            if (args.spec) {
                // two streams per CU or more: the 73-register flavour, three workgroups per CU
                if (B >= h->tp_from * h->cus && B > 0) return launch_advance_d<W, NW, false, LiveFromGlobalLean, true>(h, args, B, s);
                if (f32) return launch_advance_d<W, NW, false, float, true>(h, args, B, s);
                // float64 features: the ring (one workgroup per CU at W = 512) while every stream has a CU to itself
                // (B = 64: 4.67 vs 4.93 ms), no ring and up to four workgroups per CU beyond
                return (B <= h->cus) ? launch_advance_d<W, NW, false, double, true>(h, args, B, s)
                                     : launch_advance_d<W, NW, false, LiveFromGlobal, true>(h, args, B, s);
            }
        }
        return f32 ? launch_advance_d<W, NW, false, float, false>(h, args, B, s)
                   : launch_advance_d<W, NW, false, double, false>(h, args, B, s);
    }
}

template <int W>
static int launch_w(rts_otw *h, const OtwArgs &args, int B, int waves, hipStream_t s) {
    switch (waves) {
        case 1: return launch_advance<W, 1>(h, args, B, s);
        case 2: return launch_advance<W, 2>(h, args, B, s);
        case 4: return launch_advance<W, 4>(h, args, B, s);
        case 8: return launch_advance<W, 8>(h, args, B, s);
    }
    return set_error(RTS_ERR_INVALID, "waves must be 1, 2, 4 or 8 (got %d)", waves);
}

// A handle belongs to the device that was current at rts_otw_create; driving it with another device current would
// launch against foreign buffers.
static int check_device(const rts_otw *h) {
    int d = -1;
    RTS_HIP(hipGetDevice(&d));
    if (d != h->device)
        return set_error(RTS_ERR_INVALID, "handle was created on device %d but device %d is current "
                                          "(one process per GPU, or hipSetDevice before the call)", h->device, d);
    return RTS_OK;
}

static int launch(rts_otw *h, const OtwArgs &args, hipStream_t s) {
    if ((uintptr_t)args.live & 15) return set_error(RTS_ERR_INVALID, "live features must be 16-byte aligned");
    switch (h->W) {
        case 64: return launch_w<64>(h, args, h->B, h->waves, s);
        case 128: return launch_w<128>(h, args, h->B, h->waves, s);
        case 256: return launch_w<256>(h, args, h->B, h->waves, s);
        case 512: return launch_w<512>(h, args, h->B, h->waves, s);
        case 1024: return launch_w<1024>(h, args, h->B, h->waves, s);  // c up to 1012: one workgroup per CU
        case 2048: return launch_w<2048>(h, args, h->B, h->waves, s);  // c up to 2036: 32 cells per lane in the chains
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
    a.spec = h->spec;
    a.dense_acc = h->dense_acc;
    a.dense_cost = h->dense_cost;
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
    if (F < 1) return set_error(RTS_ERR_INVALID, "F must be >= 1 (got %d)", F);
    if (F != kF) return set_error(RTS_ERR_UNSUPPORTED, "F must be 12 chroma bins (got %d)", F);
    if (N < 1 || B < 1) return set_error(RTS_ERR_INVALID, "N and B must be >= 1 (got N=%d B=%d)", N, B);
    if (ref_dtype != RTS_F32 && ref_dtype != RTS_F64) return set_error(RTS_ERR_INVALID, "bad ref_dtype %d", ref_dtype);
    if (c < 1) return set_error(RTS_ERR_INVALID, "c must be >= 1 (got %d)", c);
    if (c > 2036)
        return set_error(RTS_ERR_UNSUPPORTED, "band width c=%d exceeds the 2036 cells the LDS-resident kernel holds", c);
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
    h->waves = 8;  // wave 0: control (+ chains of steps that are not hits), waves 1/2: chains, 3..7: cost strips, ring refill
    {
        // The pipelined kernel (58 KB of LDS at c = 500 with float32 features: two workgroups per CU, like the plain
        // kernel) is the faster one at every batch size measured; RTS_OTW_SPEC=0 selects the plain kernel (A/B runs,
        // tests).  Results are identical.
        const char *sp = getenv("RTS_OTW_SPEC");
        h->spec = sp ? (atoi(sp) != 0) : 1;
        const char *tp = getenv("RTS_OTW_TP_FROM");  // tests: 0 selects the residency-oriented flavour at any batch size
        h->tp_from = tp ? atoi(tp) : 2;
    }
    h->live_cap = 2 * N;
    h->path_cap = 3 * N + 8;  // one point per decide(); decides <= row strips + column strips <= 2N + N
    hipError_t e;
    if ((e = hipGetDevice(&h->device)) != hipSuccess ||
        (e = hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, h->device)) != hipSuccess) {
        free(h);
        return set_error(RTS_ERR_HIP, "hipGetDevice failed: %s", hipGetErrorString(e));
    }
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
    if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) {
        rts_otw_destroy(h);
        return set_error(RTS_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    }
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
    if (h->rp_state) (void)hipFree(h->rp_state);
    if (h->rp_path) (void)hipFree(h->rp_path);
    if (h->rp_bands) (void)hipFree(h->rp_bands);
    free(h);
    return RTS_OK;
}

int rts_otw_reset(rts_otw *h, void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (int rc = check_device(h); rc != RTS_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    h->src_kind = 0;
    hipLaunchKernelGGL(otw_reset_kernel, dim3((h->B + 63) / 64), dim3(64), 0, s, h->state, h->B, h->variant);
    RTS_HIP(hipGetLastError());
    RTS_HIP(hipMemsetAsync(h->hist_len, 0, sizeof(int32_t) * (size_t)h->B, s));
    if (h->dense_acc) {  // otw_eran.py:23,27 / livenote_v2.py:21-23
        const long long n = (long long)h->B * h->live_cap * h->N;
        const double sentinel = (h->variant == RTS_VARIANT_OTW) ? 1e10 : (double)INFINITY;
        hipLaunchKernelGGL(otw_fill_kernel, dim3(2048), dim3(256), 0, s, h->dense_acc, n, sentinel);
        hipLaunchKernelGGL(otw_fill_kernel, dim3(2048), dim3(256), 0, s, h->dense_cost, n, -1.0);
        RTS_HIP(hipGetLastError());
    }
    return RTS_OK;
}

int rts_otw_set_dense(rts_otw *h, double *acc_dev, double *cost_dev, void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if ((acc_dev == nullptr) != (cost_dev == nullptr))
        return set_error(RTS_ERR_INVALID, "acc_dev and cost_dev must both be given or both be NULL");
    h->dense_acc = acc_dev;
    h->dense_cost = cost_dev;
    return rts_otw_reset(h, stream);
}

int rts_otw_replay_dense(rts_otw *h, const void *live_dev, int live_dtype, int T_max, const int32_t *live_len_dev,
                         double *acc_dev, double *cost_dev, void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (!acc_dev || !cost_dev) return set_error(RTS_ERR_INVALID, "acc_dev / cost_dev is NULL");
    if (int rc = check_device(h); rc != RTS_OK) return rc;
    if (h->src_kind == 3)
        return set_error(RTS_ERR_UNSUPPORTED, "rts_otw_insert / rts_otw_push after rts_otw_run without a reset: nothing to replay from");
    // The frames of an rts_otw_run are the caller's: the library does not keep a pointer to memory it does not own, the
    // caller hands the buffers in again.  Frames that came through rts_otw_insert / rts_otw_push are in the handle's history.
    if (h->src_kind == 1) {
        if (!live_dev || !live_len_dev)
            return set_error(RTS_ERR_INVALID, "the handle's frames came from rts_otw_run: pass that call's live_dev / live_len_dev again");
        if (live_dtype != RTS_F32 && live_dtype != RTS_F64) return set_error(RTS_ERR_INVALID, "bad live_dtype %d", live_dtype);
        if (live_dtype != h->run_dtype || T_max != h->run_T)
            return set_error(RTS_ERR_INVALID, "live_dtype / T_max differ from the rts_otw_run being replayed (%d / %d then)", h->run_dtype, h->run_T);
    } else if (live_dev) {
        return set_error(RTS_ERR_INVALID, "the handle's frames came through rts_otw_insert / rts_otw_push (or it is fresh): live_dev must be NULL");
    }
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)h->B * h->live_cap * h->N;
    const double sentinel = (h->variant == RTS_VARIANT_OTW) ? 1e10 : (double)INFINITY;
    hipLaunchKernelGGL(otw_fill_kernel, dim3(2048), dim3(256), 0, s, acc_dev, n, sentinel);
    hipLaunchKernelGGL(otw_fill_kernel, dim3(2048), dim3(256), 0, s, cost_dev, n, -1.0);
    RTS_HIP(hipGetLastError());
    if (h->src_kind == 0) return RTS_OK;  // freshly constructed: the matrices are all sentinel (otw_eran.py:23,27)
    // scratch state so that the handle itself is not disturbed; allocated on the first replay, kept with the handle
    if (!h->rp_state) {
        hipError_t e;
        if ((e = hipMalloc((void **)&h->rp_state, sizeof(int32_t) * RTS_STATE_LEN * (size_t)h->B)) != hipSuccess ||
            (e = hipMalloc((void **)&h->rp_path, sizeof(int32_t) * 2 * (size_t)h->path_cap * h->B)) != hipSuccess ||
            (e = hipMalloc((void **)&h->rp_bands, sizeof(double) * 2 * (size_t)(h->c + 1) * h->B)) != hipSuccess) {
            if (h->rp_state) (void)hipFree(h->rp_state);
            if (h->rp_path) (void)hipFree(h->rp_path);
            h->rp_state = h->rp_path = nullptr;
            return set_error(RTS_ERR_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
        }
    }
    hipLaunchKernelGGL(otw_reset_kernel, dim3((h->B + 63) / 64), dim3(64), 0, s, h->rp_state, h->B, h->variant);
    OtwArgs a = base_args(h);
    a.state = h->rp_state;
    a.path = h->rp_path;
    a.bands = h->rp_bands;
    a.dense_acc = acc_dev;
    a.dense_cost = cost_dev;
    if (h->src_kind == 1) {
        a.live = live_dev;
        a.live_len = live_len_dev;
        a.live_stride = T_max;
        a.live_f64 = live_dtype == RTS_F64;
        a.mode = h->run_mode;
        a.clamp_len = 1;
    } else {
        a.live = h->hist;
        a.live_len = h->hist_len;
        a.live_stride = h->live_cap;
        a.live_f64 = 1;
        a.mode = RTS_MODE_INSERT_LOOP;
    }
    int rc = launch(h, a, s);
    if (rc != RTS_OK) return rc;
    RTS_HIP(hipStreamSynchronize(s));
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
    h->src_kind = 1;
    h->run_dtype = live_dtype;
    h->run_T = T_max;
    h->run_mode = mode;
    return launch(h, a, s);
}

int rts_otw_insert(rts_otw *h, const void *frames_dev, int frames_dtype, const uint8_t *active_dev, void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (!frames_dev) return set_error(RTS_ERR_INVALID, "frames_dev is NULL");
    if (frames_dtype != RTS_F32 && frames_dtype != RTS_F64) return set_error(RTS_ERR_INVALID, "bad frames_dtype %d", frames_dtype);
    hipStream_t s = (hipStream_t)stream;
    if (int rc = check_device(h); rc != RTS_OK) return rc;
    if (!h->hist) {
        RTS_HIP(hipMalloc((void **)&h->hist, sizeof(double) * kF * (size_t)h->live_cap * h->B));
    }
    h->src_kind = (h->src_kind == 0 || h->src_kind == 2) ? 2 : 3;
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

int rts_otw_push(rts_otw *h, const void *frames_dev, int frames_dtype, int n_max, const int32_t *n_new_dev,
                 void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (n_max < 0) return set_error(RTS_ERR_INVALID, "n_max < 0");
    if (n_max == 0) return RTS_OK;
    if (!frames_dev) return set_error(RTS_ERR_INVALID, "frames_dev is NULL");
    if (frames_dtype != RTS_F32 && frames_dtype != RTS_F64) return set_error(RTS_ERR_INVALID, "bad frames_dtype %d", frames_dtype);
    hipStream_t s = (hipStream_t)stream;
    if (int rc = check_device(h); rc != RTS_OK) return rc;
    if (!h->hist) {
        RTS_HIP(hipMalloc((void **)&h->hist, sizeof(double) * kF * (size_t)h->live_cap * h->B));
    }
    h->src_kind = (h->src_kind == 0 || h->src_kind == 2) ? 2 : 3;
    hipLaunchKernelGGL(otw_append_many_kernel, dim3(h->B), dim3(128), 0, s, h->hist, h->hist_len, frames_dev,
                       frames_dtype == RTS_F64, n_new_dev, n_max, n_max, h->B, h->live_cap);
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
