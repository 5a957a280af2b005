// Strip DP: the accumulated-cost recurrence of dtw.DTW (/root/reference/dtw.py:32-40) and of
// WTW.run_dtw (/root/reference/wtw.py:173-217) as a barrier-light systolic sweep for gfx950.
//
// Mapping.  A wave owns a STRIP of 64 matrix rows, one row per lane, and sweeps it left to right with
// the lanes skewed in time: at strip-step s lane l evaluates column j = s - l.  Every predecessor of a
// cell is then one DPP move away -- (i, j-1) is the lane's own previous value, (i-1, j) is lane l-1's
// previous value (v_mov_dpp wave_shr:1), (i-1, j-1) is what that move delivered one step earlier -- so
// a step needs no barrier and no LDS round trip for the recurrence.  Each cell performs exactly the
// reference's float64 operations in the reference's order, hence bit-identical results.
//
// Roles.  A strip is served by 1 + H waves.  Its DP wave does nothing but the recurrence: per step one
// ds_read_b64 of the cell cost, two DPP moves, the compares/selects, and the bookkeeping of the outputs.  Its H
// HELPER waves produce the costs one chunk (16 columns) ahead: lane = row again, but they sweep whole columns,
// so the column's feature record is wave-uniform -- helper 0 stages the next chunk's 16 records (1.8 KB) in LDS
// while the current ones are consumed, and every lane reads the same address (a broadcast, seven ds_read_b128
// per column instead of per cell) -- and each cost goes to the slot (step = column + row, lane = row) of a
// 96-step LDS cost ring that the DP wave reads as consecutive 512-byte rows.  Helper 0 also moves the
// accumulated costs of the previous chunk from the DP wave's LDS tile to HBM.
//
// A workgroup owns a ROW GROUP of NS <= 2 consecutive strips; strip s+1 runs kLag chunks (80 steps) behind
// strip s and reads the bottom row of the strip above straight from that strip's output tiles (three tiles
// per strip, so a tile outlives its two readers).  All waves of a workgroup advance in lockstep, one LDS-only
// barrier per chunk of 16 steps.  Consecutive row groups run on different workgroups of the same launch,
// pipelined the same way through HBM: the bottom row of a row group's last strip is published as
// 8-byte write-through (sc1) stores, the next row group's first DP wave polls those words with sc1 loads.
// The data is its own flag: the boundary buffer is pre-filled with a signalling-NaN bit pattern that no
// float64 add/subtract can produce (arithmetic results are always quiet), so a word that differs from
// it is the value (single naturally aligned 8-byte store: untorn; no fences, no separate flag).
//
// Outputs.  Back-pointers: 2 bits per cell, 16 steps of a lane packed into one dword, stored in skewed
// order [strip][chunk][lane] (one coalesced 256-byte store per wave and chunk) -- the backtrack below
// decodes that layout.  The accumulated-cost matrix (dtw.py's acc_cost, wtw.py's D), when wanted, is
// transposed through the strip's LDS tiles so that HBM sees 128-byte row segments instead of 64 scattered
// 8-byte stores per step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "common.h"

namespace rts {
namespace sdp {

constexpr int kF = 12;
constexpr int kChunk = 16;     // steps between two workgroup barriers
constexpr int kLag = 5;        // chunks wave w+1 runs behind wave w: lane 63 finishes column 16m+15 at step 16m+78
constexpr int kYRec = 14;      // doubles per prepared column record: 12 features, norm, pad (112 bytes)
constexpr int kRing = 96;      // steps of the LDS cost ring: a helper runs one chunk ahead and a column spans 64 steps
constexpr int kStageLd = 65;   // leading dimension of the [16][64] output tiles
constexpr int kTiles = 3;      // output tiles per strip: written in chunk m, read in chunks m+1 and m+2
constexpr int kMaxStrips = 2;  // strips per workgroup (LDS: 77 696 bytes per strip)
constexpr unsigned long long kSentinel = 0x7FF4DEAD7FF4DEADull;  // signalling NaN: never an arithmetic result
constexpr uint32_t kSentinel32 = 0x7FF4DEADu;                    // the same as a 32-bit fill pattern
constexpr int kSpinLimit = 1 << 22;

// internal 2-bit step codes
constexpr int kLeft = 0;  // from (i, j-1)
constexpr int kUp = 1;    // from (i-1, j)
constexpr int kDiag = 2;  // from (i-1, j-1)

__host__ __device__ inline int n_strips(int M) { return (M + 63) / 64; }
__host__ __device__ inline int n_chunks(int N) { return (N + 63 + kChunk - 1) / kChunk; }  // per strip
__host__ __device__ inline size_t lds_core_bytes(int NS) {
    return sizeof(double) * (size_t)NS * ((size_t)kRing * 64 + (size_t)kTiles * kChunk * kStageLd + 2 * kChunk * kYRec + 64 + 2 * kChunk);
}
// + 16 bytes behind the strips' buffers: the workgroup's current ticket (for_each_rowgroup).  All of it is dynamic LDS --
// the kernels declare no static LDS, so that the dynamic limit can be raised to the full 160 KB.
__host__ __device__ inline size_t lds_bytes(int NS) { return lds_core_bytes(NS) + 16; }
// Strips per workgroup / helper waves per strip / workgroups of one problem's pipeline.  One strip per workgroup
// (3 helpers; the DP wave, which sets the pace, has a SIMD to itself; 77 KB of LDS, so two workgroups share a CU) when
// every strip of every problem can have a resident workgroup of its own; otherwise two strips per workgroup (2 helpers
// each), which halves the number of passes a workgroup makes over the columns.
//
// Residency.  A problem's row groups wait for one another inside the launch (run_rowgroup polls the bottom row of the
// row group above).  That is deadlock-free WHATEVER the residency, because row groups are not bound to workgroups:
// a workgroup takes the next row group of its problem from a ticket counter when it starts and whenever it finishes
// one (for_each_rowgroup), so row group r - 1 was always taken by a workgroup that is already running -- by induction it
// finishes, and r with it.  `resident1` / `resident2` (workgroups of the one-strip / two-strip kernel the device holds
// at once: compute units x hipOccupancyMaxActiveBlocksPerMultiprocessor, queried by the caller for the instantiation
// it launches) only size the grid: workgroups beyond residency would just queue behind the running ones.
// RTS_SDP_CONFIG=1|2 forces the strips per workgroup, RTS_SDP_GRID=n the workgroups per problem (tuning and tests;
// results do not depend on either).
inline void pick_config(int strips, int B, int resident1, int resident2, int &NS, int &H, int &grid) {
    if (resident1 < 1) resident1 = 1;
    if (resident2 < 1) resident2 = 1;
    NS = ((long long)strips * B <= resident1) ? 1 : 2;
    if (const char *e = getenv("RTS_SDP_CONFIG")) NS = (atoi(e) == 1) ? 1 : 2;
    if (NS > strips) NS = strips;
    H = (NS == 1) ? 3 : 2;
    const int n_rg = (strips + NS - 1) / NS;
    const int resident = (NS == 1) ? resident1 : resident2;
    grid = resident / B;
    if (grid < 1) grid = 1;
    if (grid > n_rg) grid = n_rg;
    if (const char *e = getenv("RTS_SDP_GRID")) grid = atoi(e) > 0 ? atoi(e) : grid;
}
// Test knob: extra bytes of dynamic LDS per workgroup (lowers the residency the occupancy query reports and the
// hardware grants, e.g. 80000 -> one workgroup per CU); 0 in production.
inline size_t lds_pad() {
    const char *e = getenv("RTS_SDP_LDS_PAD");
    const long v = e ? atol(e) : 0;
    return v > 0 && v < 80 * 1024 ? (size_t)v : 0;
}
// Workgroups of `kernel` (block threads, smem bytes of dynamic LDS) resident on the current device at once.
template <typename K>
inline int resident_blocks(K kernel, int block, size_t smem) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, smem) != hipSuccess) {
        (void)hipGetLastError();  // a configuration that does not fit (e.g. padded LDS) is an answer, not a sticky error
        return 0;
    }
    return cus * per_cu;
}
__host__ __device__ inline size_t codes_words(int M, int N) { return (size_t)n_strips(M) * n_chunks(N) * 64; }

struct Problem {
    const void *x;        // row features [M][12]
    int x_f64;
    const double *yrec;   // prepared column records [N][kYRec] (prep_column)
    int M, N;
    double *D;             // optional row-major output, leading dimension ldD
    long long ldD;
    uint32_t *codes;       // [n_strips][n_chunks][64] packed step codes
    unsigned long long *bnd;  // [n_rowgroups][N] bottom rows handed between row groups (sentinel-filled)
    int32_t *entb;         // [n_strips][N]: for every cell of a strip's bottom row, the column at which its best path
                           // entered the strip from the row above (what lets the backtrack hop strip to strip)
    int32_t *err;          // set to 1 if a poll ran into its bound (never expected)
};

__device__ __forceinline__ void load_frame(const void *p, int f64, long long frame, double (&v)[kF]) {
    if (f64) {
        const double2 *q = reinterpret_cast<const double2 *>(reinterpret_cast<const double *>(p) + frame * kF);
#pragma unroll
        for (int k = 0; k < kF / 2; k++) {
            const double2 t = q[k];
            v[2 * k] = t.x;
            v[2 * k + 1] = t.y;
        }
    } else {
        const float4 *q = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + frame * kF);
#pragma unroll
        for (int k = 0; k < kF / 4; k++) {
            const float4 t = q[k];
            v[4 * k] = (double)t.x;
            v[4 * k + 1] = (double)t.y;
            v[4 * k + 2] = (double)t.z;
            v[4 * k + 3] = (double)t.w;
        }
    }
}

// value of lane l-1 (lane 0 receives `lane0`)
__device__ __forceinline__ double shr1(double v, double lane0) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(lane0), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(lane0), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int shr1_i(int v, int lane0) {
    return __builtin_amdgcn_update_dpp(lane0, v, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ double readlane_d(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

using gu64 = __attribute__((address_space(1))) unsigned long long;

__device__ __forceinline__ unsigned long long load_sc1(const unsigned long long *p) {
    return __hip_atomic_load((const gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load_dwordx2 sc1
}
__device__ __forceinline__ void store_sc1(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store((gu64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dwordx2 sc1
}

// v_min_f64 as is (fmin() would wrap it in canonicalising v_max pairs).  IEEE minNum: a NaN operand is ignored.
__device__ __forceinline__ double vmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- policies: the per-cell arithmetic of the two reference recurrences ---------------------------------------

// dtw.py:11 cost = 1 - seq_a.T . seq_b (dgemm k-order: one fma chain), dtw.py:32-40 options (left, up, diag + 2c),
// np.argmin = first minimum.
struct DtwPolicy {
    static constexpr bool kNorm = false;
    static constexpr int kHelper0Cols = 0;  // cost columns per chunk on the helper that also runs the entry-column pass
    static __device__ __forceinline__ double norm(const double (&)[kF]) { return 0.0; }
    static __device__ __forceinline__ double cost(const double (&x)[kF], double, const double (&y)[kF], double) {
        double s = 0.0;
#pragma unroll
        for (int f = 0; f < kF; f++) s = fma(x[f], y[f], s);
        return 1.0 - s;
    }
    // two columns at once, the two dependent fma chains interleaved (a lone chain is latency-bound)
    static __device__ __forceinline__ void cost_pair(const double (&x)[kF], double, const double (&ya)[kF], double,
                                                     const double (&yb)[kF], double, double &ca, double &cb) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int f = 0; f < kF; f++) {
            sa = fma(x[f], ya[f], sa);
            sb = fma(x[f], yb[f], sb);
        }
        ca = 1.0 - sa;
        cb = 1.0 - sb;
    }
    static __device__ __forceinline__ void cell(bool first_row, bool first_col, double up, double left, double diag,
                                                double c, double &dv, int &code) {
        // value and step code separately: the value is what the next step waits for (two v_min_f64 instead of two
        // compare-and-select pairs); the code -- np.argmin's first minimum -- comes from compares off that chain.
        // Same results for finite inputs (chroma columns are finite; NaN inputs are outside dtw.py's domain).
        const double o0 = left + c, o1 = up + c, o2 = diag + 2 * c;
        const double m01 = vmin(o0, o1);
        double best = vmin(m01, o2);
        int s = (o1 < o0) ? kUp : kLeft;
        s = (o2 < m01) ? kDiag : s;
        if (first_col) {  // dtw.py:23-25
            best = o1;
            s = kUp;
        }
        if (first_row) {  // dtw.py:26-28
            best = o0;
            s = kLeft;
        }
        if (first_row && first_col) {  // dtw.py:20-21
            best = c;
            s = kDiag;
        }
        dv = best;
        code = s;
    }
};

// wtw.py:169 cost = 1 - x.y / (|x| |y|) (np.dot on strided columns: OpenBLAS ddot order, two accumulators; norms
// are fma chains), wtw.py:201-215 candidates (i-1,j), (i,j-1), (i-1,j-1) with strict '<' in that order.
struct WtwPolicy {
    static constexpr bool kNorm = true;
    static constexpr int kHelper0Cols = 2;  // a normalised-cosine cost is ~3x the work of a plain dot product
    static __device__ __forceinline__ double norm(const double (&v)[kF]) {
        double s = 0.0;
#pragma unroll
        for (int f = 0; f < kF; f++) s = fma(v[f], v[f], s);
        return sqrt(s);
    }
    static __device__ __forceinline__ double cost(const double (&x)[kF], double nx, const double (&y)[kF], double ny) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int f = 0; f < kF; f += 4) {
            const double m3 = y[f + 2] * x[f + 2];
            const double m4 = y[f + 3] * x[f + 3];
            const double a = fma(y[f], x[f], m3);
            const double b = fma(y[f + 1], x[f + 1], m4);
            t1 = t1 + a;
            t2 = t2 + b;
        }
        return 1.0 - (t1 + t2) / (nx * ny);
    }
    static __device__ __forceinline__ void cost_pair(const double (&x)[kF], double nx, const double (&ya)[kF], double nya,
                                                     const double (&yb)[kF], double nyb, double &ca, double &cb) {
        double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;
#pragma unroll
        for (int f = 0; f < kF; f += 4) {
            const double ma3 = ya[f + 2] * x[f + 2], mb3 = yb[f + 2] * x[f + 2];
            const double ma4 = ya[f + 3] * x[f + 3], mb4 = yb[f + 3] * x[f + 3];
            const double pa = fma(ya[f], x[f], ma3), pb = fma(yb[f], x[f], mb3);
            const double qa = fma(ya[f + 1], x[f + 1], ma4), qb = fma(yb[f + 1], x[f + 1], mb4);
            a1 = a1 + pa;
            b1 = b1 + pb;
            a2 = a2 + qa;
            b2 = b2 + qb;
        }
        ca = 1.0 - (a1 + a2) / (nx * nya);
        cb = 1.0 - (b1 + b2) / (nx * nyb);
    }
    static __device__ __forceinline__ void cell(bool first_row, bool first_col, double up, double left, double diag,
                                                double c, double &dv, int &code) {
        // wtw.py:201-215 replaces min_cost only by a strictly smaller candidate, so a NaN candidate (silent frame) is
        // ignored -- v_min_f64 does exactly that -- while a NaN in min_cost's initial value, (i-1, j), stays
        double mc = vmin(vmin(up, left), diag);
        const bool up_nan = (up != up);
        mc = up_nan ? up : mc;
        int s = (left < up) ? kLeft : kUp;
        s = (diag < vmin(up, left)) ? kDiag : s;
        s = up_nan ? kUp : s;
        if (first_col) {  // wtw.py:187-191
            mc = up;
            s = kUp;
        }
        if (first_row) {  // wtw.py:194-198
            mc = left;
            s = kLeft;
        }
        double v = mc + c;
        if (first_row && first_col) v = c;  // wtw.py:183
        dv = v;
        code = s;
    }
};

template <bool V>
struct BoolC {
    static constexpr bool value = V;
};
template <int I>
struct IntC {
    static constexpr int value = I;
};
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(IntC<I>());
        static_for<I + 1, N>(f);
    }
}

// ---- column records: float64 features (+ norm where the cost needs it), 112 bytes, one thread per column ----------
template <class P>
__device__ __forceinline__ void prep_column(const void *y, int y_f64, long long col, double *yrec) {
    double v[kF];
    load_frame(y, y_f64, col, v);
    double2 *rec = reinterpret_cast<double2 *>(yrec + (size_t)col * kYRec);
#pragma unroll
    for (int k = 0; k < kF / 2; k++) rec[k] = make_double2(v[2 * k], v[2 * k + 1]);
    rec[kF / 2] = make_double2(P::kNorm ? P::norm(v) : 0.0, 0.0);
}

// Division of labour among the H helper waves of a strip.  Helper 0 stages the column records and runs the (serial)
// entry-column recurrence, so it takes few or no cost columns and none of the output traffic.
__host__ __device__ constexpr bool helper_takes_column(int H, int hidx, int kk, int h0cols) {  // kk: column in the chunk
    if (H == 3) return kk < h0cols ? hidx == 0 : (hidx != 0 && ((kk - h0cols) & 1) == (hidx - 1));
    return hidx == 0 ? (kk % 8) < 3 : (kk % 8) >= 3;  // H == 2: 6 + 10
}
__host__ __device__ constexpr int helper_column_ordinal(int H, int hidx, int kk, int h0cols) {  // how many of mine before kk
    int n = 0;
    for (int q = 0; q < kk; q++) n += helper_takes_column(H, hidx, q, h0cols) ? 1 : 0;
    return n;
}
__host__ __device__ constexpr int helper_next_column(int H, int hidx, int kk, int h0cols) {  // my next column after kk, or 16
    for (int q = kk + 1; q < 16; q++)
        if (helper_takes_column(H, hidx, q, h0cols)) return q;
    return 16;
}
__host__ __device__ constexpr bool helper_takes_rows(int H, int hidx, int it) {  // it: group of 8 tile rows
    if (H == 3) return hidx != 0 && (it & 1) == (hidx - 1);
    return hidx == 1;
}

// Row groups are handed out by ticket (see pick_config, "Residency"): `ticket` is a zero-initialised device word per
// problem; f(rg) runs the row group with the whole workgroup.
template <typename F>
__device__ __forceinline__ void for_each_rowgroup(int32_t *ticket, int n_rg, int NS, unsigned char *smem, F f) {
    volatile int *slot = reinterpret_cast<volatile int *>(smem + lds_core_bytes(NS));
    for (;;) {
        __syncthreads();  // every wave is done with the previous row group (LDS tiles, the slot)
        if (threadIdx.x == 0) *slot = atomicAdd(ticket, 1);
        __syncthreads();
        const int rg = __builtin_amdgcn_readfirstlane(*slot);
        if (rg >= n_rg) break;
        f(rg);
    }
}

// ---- one row group (NS strips) of one problem, executed by the whole workgroup ----------------------------------
// Waves: block 0 = helper 0 of strips 0..NS-1, block 1 = the DP waves, blocks 2.. = further helpers (with NS = 2 and
// H = 2 the two DP waves are waves 2 and 3 and have a SIMD to themselves).
template <class P, bool STAGE, int H>
__device__ __forceinline__ void run_rowgroup(const Problem &pb, int rg, int n_rg, int NS, unsigned char *smem) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int blk = wave / NS, sidx = wave - blk * NS;  // my strip within the row group
    const bool is_dp = (blk == 1);
    const int hidx = (blk == 0) ? 0 : blk - 1;  // helper index (helpers only)
    const int M = pb.M, N = pb.N;
    double *cring = reinterpret_cast<double *>(smem) + (size_t)sidx * kRing * 64;         // [kRing][64] cell costs
    double *tiles = reinterpret_cast<double *>(smem) + (size_t)NS * kRing * 64;           // [NS][kTiles][16][65]
    double *tile_mine = tiles + (size_t)sidx * kTiles * kChunk * kStageLd;
    const double *tile_up = tiles + (size_t)(sidx > 0 ? sidx - 1 : 0) * kTiles * kChunk * kStageLd;
    double *yst = tiles + (size_t)NS * kTiles * kChunk * kStageLd + (size_t)sidx * 2 * kChunk * kYRec;  // [2][16][kYRec]
    // the DP wave's packed step codes of its last two chunks, for the helper that derives the entry columns from them
    uint32_t *codes_l = reinterpret_cast<uint32_t *>(tiles + (size_t)NS * kTiles * kChunk * kStageLd +
                                                     (size_t)NS * 2 * kChunk * kYRec) + (size_t)sidx * 128;  // [2][64]
    // row-group boundary: the bottom row of the row group above, fetched from HBM by helper 0 one chunk ahead
    double *upin = tiles + (size_t)NS * kTiles * kChunk * kStageLd + (size_t)NS * 2 * kChunk * kYRec + (size_t)NS * 64;  // [2][16]

    const int strip = rg * NS + sidx;
    const bool strip_ok = strip * 64 < M;
    const int i = strip * 64 + lane;
    const int nch = n_chunks(N);
    const int ncc = (N + kChunk - 1) / kChunk;  // chunks of columns
    const int total = nch + kLag * (NS - 1) + 2;  // + the helpers' head start and the last tile's way out
    const bool from_hbm = (sidx == 0 && rg > 0);
    const bool to_hbm = (sidx == NS - 1 && rg + 1 < n_rg);
    const unsigned long long *bnd_in = pb.bnd + (size_t)(rg > 0 ? rg - 1 : 0) * N;
    unsigned long long *bnd_out = pb.bnd + (size_t)rg * N;

    if (!is_dp) {
        // ================================ helper wave =========================================================
        double x[kF];
        {
            const int ic = i < M ? i : M - 1;
            load_frame(pb.x, pb.x_f64, ic, x);
        }
        const double nx = P::norm(x);
        int ent = 0, ent_upprev = 0;  // helper 0: entry columns (see Problem::entb) of my row's last cell / its upper-left
        // helper 0: a poll of the row-group boundary ran into its bound (here or, recorded in *err, anywhere
        // in the launch: the fault is reported once, the remaining row groups run through without polling)
        bool dead = (__hip_atomic_load(pb.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
#ifdef RTS_SDP_STAMPS
        long long sh_cost = 0, sh_flush = 0, sh_bar = 0;
#endif
        // the records of column chunk `cc` as 112 16-byte pieces: lane l holds pieces l and l + 64 (columns past the
        // last one repeat it: their costs land in ring slots no valid cell reads)
        double2 pc0 = make_double2(0.0, 0.0), pc1 = pc0;
        auto fetch_records = [&](int cc) {
            const double2 *src = reinterpret_cast<const double2 *>(pb.yrec);
            const int p0 = lane, p1 = lane + 64;
            int c_a = kChunk * cc + p0 / 7, c_b = kChunk * cc + p1 / 7;
            c_a = c_a < N ? c_a : N - 1;
            c_b = c_b < N ? c_b : N - 1;
            pc0 = src[(size_t)c_a * 7 + p0 % 7];
            if (p1 < kChunk * 7) pc1 = src[(size_t)c_b * 7 + p1 % 7];
        };
        auto commit_records = [&](int cc) {
            double2 *dst = reinterpret_cast<double2 *>(yst + (size_t)(cc & 1) * kChunk * kYRec);
            dst[lane] = pc0;
            if (lane + 64 < kChunk * 7) dst[lane + 64] = pc1;
        };
        // output address of (tile row lane >> 3, column pair lane & 7) of row-group it = 0, chunk 0; the element of row r
        // and step-in-chunk q0 sits at column 16 mf + q0 - r
        double *drow = STAGE ? pb.D + ((size_t)strip * 64 + (lane >> 3)) * pb.ldD + (2 * (lane & 7) - (lane >> 3)) : nullptr;
        // the helper's role is fixed for the launch: one specialised loop per helper index, no per-column tests
        auto helper_main = [&](auto hic) {
        constexpr int HIDX = decltype(hic)::value;
        for (int k = -1; k < total; k++) {
            if (strip_ok) {
                const int mh = k - kLag * sidx;
                // ---- helper 0: fetch the records of the next column chunk now, park them in LDS at the end of the chunk
                const bool pre = (HIDX == 0) && (mh + 1 >= 0) && (mh + 1 < ncc);
                if (pre) fetch_records(mh + 1);
                // ---- helper 0 of a row group's first strip: the bottom row of the row group above, columns
                //      [16 mh, 16 mh + 16), which the DP wave enters in its next chunk.  Issued now, examined at the end.
                const bool poll = (HIDX == 0) && from_hbm && mh >= 0 && mh < ncc;
                unsigned long long bnd_bits = kSentinel;
                const int bcol = kChunk * mh + lane;
                const bool bwant = poll && lane < kChunk && bcol < N;
                if (bwant) bnd_bits = load_sc1(bnd_in + bcol);
                // ---- costs of the columns the DP wave sweeps into during its next chunk
#ifdef RTS_SDP_STAMPS
                const long long sh_a = (long long)__builtin_amdgcn_s_memtime();
#endif
                if (mh >= 0 && mh < ncc) {
                    const int c0 = kChunk * mh;
                    const int r0 = (c0 + lane) % kRing;  // ring step of (column c0, my row)
                    const double2 *recs = reinterpret_cast<const double2 *>(yst + (size_t)(mh & 1) * kChunk * kYRec);
                    auto load_rec = [&](int kk, double (&y)[kF], double &ny) {
                        const double2 *rec = recs + (size_t)kk * (kYRec / 2);  // same address in every lane
#pragma unroll
                        for (int f = 0; f < kF / 2; f++) {
                            const double2 t = rec[f];
                            y[2 * f] = t.x;
                            y[2 * f + 1] = t.y;
                        }
                        ny = P::kNorm ? rec[kF / 2].x : 0.0;
                    };
                    auto put = [&](int kk, double c) {
                        int r = r0 + kk;
                        r = (r >= kRing) ? r - kRing : r;
                        cring[r * 64 + lane] = c;
                    };
                    // my columns of the chunk, two at a time (two interleaved dependency chains per lane)
                    static_for<0, kChunk>([&](auto kc) {
                        constexpr int kk = decltype(kc)::value;  // column within the chunk
                        if constexpr (helper_takes_column(H, HIDX, kk, P::kHelper0Cols)) {
                            constexpr int ord = helper_column_ordinal(H, HIDX, kk, P::kHelper0Cols);
                            constexpr int nxt = helper_next_column(H, HIDX, kk, P::kHelper0Cols);
                            if constexpr ((ord & 1) == 0) {
                                if constexpr (nxt < kChunk) {
                                    double ya[kF], yb[kF], nya, nyb, ca, cb;
                                    load_rec(kk, ya, nya);
                                    load_rec(nxt, yb, nyb);
                                    P::cost_pair(x, nx, ya, nya, yb, nyb, ca, cb);
                                    put(kk, ca);
                                    put(nxt, cb);
                                } else {
                                    double y[kF], ny;
                                    load_rec(kk, y, ny);
                                    put(kk, P::cost(x, nx, y, ny));
                                }
                            }
                        }
                    });
                }
#ifdef RTS_SDP_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const long long sh_b = (long long)__builtin_amdgcn_s_memtime();
                sh_cost += sh_b - sh_a;
#endif
                // ---- the DP wave's previous chunk leaves LDS: accumulated costs, a third of the tile per helper
                const int mf = k - 2 - kLag * sidx;
                if (STAGE && mf >= 0 && mf < nch) {
                    // lane = (row within a group of 8, pair of columns): two doubles per lane, 16-byte stores (8-byte
                    // aligned: the segment of row r starts at column 16 mf - r)
                    const double *t = tile_mine + (size_t)(mf % kTiles) * kChunk * kStageLd;
                    // all LDS reads first, then the stores: a store does not wait behind an LDS round trip
                    double v0[8], v1[8];
                    static_for<0, 8>([&](auto itc) {
                        constexpr int it = decltype(itc)::value;
                        if constexpr (helper_takes_rows(H, HIDX, it)) {
                            const int r = it * 8 + (lane >> 3), q0 = 2 * (lane & 7);
                            v0[it] = t[q0 * kStageLd + r];
                            v1[it] = t[(q0 + 1) * kStageLd + r];
                        }
                    });
                    typedef double dpair __attribute__((ext_vector_type(2)));
                    // interior chunk of a full strip (wave-uniform test): every lane's pair is inside the matrix
                    const bool interior = (strip * 64 + 63 < M) && (kChunk * mf >= 63) && (kChunk * mf + kChunk <= N);
                    if (interior) {
                        static_for<0, 8>([&](auto itc) {
                            constexpr int it = decltype(itc)::value;
                            if constexpr (helper_takes_rows(H, HIDX, it)) {
                                // one 16-byte store at an 8-byte aligned address (global memory takes it; the compiler
                                // would split it into two 8-byte stores).  The s_nop is the wait state a store of more
                                // than 8 bytes needs before its data registers may be overwritten: the compiler's hazard
                                // recognizer does not look inside inline asm.
                                double *dst = drow + (size_t)(it * 8) * pb.ldD + (kChunk * mf - it * 8);
                                const dpair pv = {v0[it], v1[it]};
                                asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(pv) : "memory");
                            }
                        });
                    } else {
                        static_for<0, 8>([&](auto itc) {
                            constexpr int it = decltype(itc)::value;
                            if constexpr (helper_takes_rows(H, HIDX, it)) {
                                const int r = it * 8 + (lane >> 3), q0 = 2 * (lane & 7);
                                const int row = strip * 64 + r, col = kChunk * mf + q0 - r;
                                double *dst = drow + (size_t)(it * 8) * pb.ldD + (kChunk * mf - it * 8);
                                const bool ok0 = row < M && col >= 0 && col < N, ok1 = row < M && col + 1 >= 0 && col + 1 < N;
                                if (ok0 && ok1) {
                                    const dpair pv = {v0[it], v1[it]};
                                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(pv) : "memory");
                                } else {  // the two ends of a row
                                    if (ok0) dst[0] = v0[it];
                                    if (ok1) dst[1] = v1[it];
                                }
                            }
                        });
                    }
                }
                // ---- helper 0: where did the best path of every cell of that chunk enter the strip?  The recurrence
                //      needs nothing but the step codes: a cell hands on the entry column of the predecessor it chose;
                //      lane 0's upper neighbours are in the row above, so their entry column is their own column
                //      (16 mf + q for "up", one less for "diagonal" = what lane 0 received as ent_up one step earlier)
                if (HIDX == 0 && mf >= 0 && mf < nch) {
                    const uint32_t cw = codes_l[(mf & 1) * 64 + lane];
                    // the strip's exit row: its bottom row, or the matrix's last row in the last strip -- so that the
                    // backtrack can hop over the last strip like over every other one instead of walking it first
                    const int exit_lane = (strip * 64 + 63 < M) ? 63 : (M - 1) & 63;
                    int ecollect = 0;  // lane q: the entry column of the exit row at step q (column 16 mf + q - exit_lane)
                    static_for<0, kChunk>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        // code bits as lane masks (kLeft = 00, kUp = 01, kDiag = 10): bit-selects instead of
                        // compare-and-select, which the compiler would turn into branches here
                        const int m_up = ((int)(cw << (31 - 2 * q))) >> 31;    // -1 where the code is kUp
                        const int m_dg = ((int)(cw << (30 - 2 * q))) >> 31;    // -1 where the code is kDiag
                        const int ent_up = shr1_i(ent, kChunk * mf + q);
                        const int from_above = (ent_up & m_up) | (ent_upprev & ~m_up);
                        const int m_ab = m_up | m_dg;
                        ent = (from_above & m_ab) | (ent & ~m_ab);
                        ent_upprev = ent_up;
                        const int el = __builtin_amdgcn_readlane(ent, exit_lane);
                        int &ec = ecollect;  // (an asm operand alone does not make the generic lambda capture it)
                        asm("v_writelane_b32 %0, %1, %2" : "+v"(ec) : "s"(el), "n"(q));
                    });
                    const int col = kChunk * mf - exit_lane + lane;
                    if (lane < kChunk && col >= 0 && col < N) pb.entb[(size_t)strip * N + col] = ecollect;
                }
                if (pre) commit_records(mh + 1);
                if (poll) {  // wave-uniform
                    int spins = 0;
                    while (!dead && __any(bwant && bnd_bits == kSentinel)) {
                        __builtin_amdgcn_s_sleep(2);
                        if (bwant && bnd_bits == kSentinel) bnd_bits = load_sc1(bnd_in + bcol);
                        if (++spins > kSpinLimit) {
                            dead = true;
                            if (lane == 0) atomicExch(pb.err, 1);
                        }
                    }
                    if (lane < kChunk) upin[(mh & 1) * kChunk + lane] = __longlong_as_double((long long)bnd_bits);
                }
#ifdef RTS_SDP_STAMPS
                sh_flush += (long long)__builtin_amdgcn_s_memtime() - sh_b;
#endif
            }
#ifdef RTS_SDP_STAMPS
            const long long sh_c = (long long)__builtin_amdgcn_s_memtime();
            lds_barrier();
            sh_bar += (long long)__builtin_amdgcn_s_memtime() - sh_c;
#else
            lds_barrier();
#endif
        }
        };
        if (hidx == 0)
            helper_main(IntC<0>());
        else if (hidx == 1)
            helper_main(IntC<1>());
        else
            helper_main(IntC<(H > 2 ? 2 : 1)>());
#ifdef RTS_SDP_STAMPS
        if (rg == RTS_SDP_STAMPS && sidx == 0 && lane == 0) {
            long long *dbg = reinterpret_cast<long long *>(pb.err) + 8 + 4 * hidx;
            dbg[0] = sh_cost;
            dbg[1] = sh_flush;
            dbg[2] = sh_bar;
        }
#endif
        return;
    }

    // ==================================== DP wave ==================================================================
#ifdef RTS_SDP_STAMPS
    long long st_steps = 0, st_bar = 0, st_all0 = (long long)__builtin_amdgcn_s_memtime(), st_n = 0;
#endif
    // The DP wave issues no vector-memory loads at all: its stores (step codes, the row handed to the next row group)
    // are fire-and-forget and it never waits on vmcnt.
    double prev = 0.0, upprev = 0.0;
    for (int k = -1; k < total; k++) {
        const int m = k - 1 - kLag * sidx;  // my strip's chunk
        if (strip_ok && m >= 0 && m < nch) {
            // ---- the row above, columns [16m, 16m+16): lane q holds column 16m+q
            double upbuf = 0.0;
            if (from_hbm) {
                if (lane < kChunk) upbuf = upin[(m & 1) * kChunk + lane];
            } else if (sidx > 0) {
                // the strip above finished column c at its step c + 63: chunk (c + 63) >> 4, row (c + 63) & 15 of
                // its tiles, lane 63
                if (lane < kChunk) {
                    const int t = kChunk * m + lane + 63;
                    upbuf = tile_up[(size_t)((t >> 4) % kTiles) * kChunk * kStageLd + (t & 15) * kStageLd + 63];
                }
            }
            const int jneg = lane - kChunk * m;  // column of step q is q - jneg
            const double *cbase = cring + (size_t)((kChunk * m) % kRing) * 64 + lane;
            double *tile_w = tile_mine + (size_t)(m % kTiles) * kChunk * kStageLd + lane;
            uint32_t codes = 0;

            // 16 branch-free steps.  FIRST: this strip holds matrix row 0 (lane 0); COL0: some lane is at column 0.
            auto steps = [&](auto first_c, auto col0_c) {
                constexpr bool FIRST = decltype(first_c)::value;
                constexpr bool COL0 = decltype(col0_c)::value;
                const bool first_row = FIRST && lane == 0;
                // the chunk's 16 costs were produced during the previous chunk: fetch them all now, so that no step
                // waits for an LDS round trip
                double cst[kChunk];
                static_for<0, kChunk>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    cst[q] = cbase[q * 64];
                });
                static_for<0, kChunk>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    const double c = cst[q];
                    const double up = shr1(prev, readlane_d(upbuf, q));
                    double dv;
                    int code;
                    P::cell(first_row, COL0 && (q == jneg), up, prev, upprev, c, dv, code);
                    upprev = up;
                    prev = dv;
                    codes |= (uint32_t)code << (2 * q);
                    asm("" : "+v"(codes));  // materialise now: do not keep 48 lane masks alive
                    tile_w[q * kStageLd] = dv;
                });
            };
            const bool first_strip = (strip == 0);
            const bool col0 = (kChunk * m < 64);
#ifdef RTS_SDP_STAMPS
            const long long st_a = (long long)__builtin_amdgcn_s_memtime();
#endif
            if (first_strip) {
                if (col0)
                    steps(BoolC<true>(), BoolC<true>());
                else
                    steps(BoolC<true>(), BoolC<false>());
            } else {
                if (col0)
                    steps(BoolC<false>(), BoolC<true>());
                else
                    steps(BoolC<false>(), BoolC<false>());
            }
#ifdef RTS_SDP_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            st_steps += (long long)__builtin_amdgcn_s_memtime() - st_a;
            st_n++;
#endif
            pb.codes[((size_t)strip * nch + m) * 64 + lane] = codes;
            codes_l[(m & 1) * 64 + lane] = codes;
            if (to_hbm) {  // bottom-row columns finished in this chunk: [16m - 63, 16m - 48] (my lane 63, steps 0..15)
                const int col = kChunk * m - 63 + lane;
                if (lane < kChunk && col >= 0 && col < N) {
                    __builtin_amdgcn_wave_barrier();
                    const double v = tile_mine[(size_t)(m % kTiles) * kChunk * kStageLd + lane * kStageLd + 63];
                    store_sc1(bnd_out + col, (unsigned long long)__double_as_longlong(v));
                }
            }
        }
#ifdef RTS_SDP_STAMPS
        const long long st_b = (long long)__builtin_amdgcn_s_memtime();
        lds_barrier();
        st_bar += (long long)__builtin_amdgcn_s_memtime() - st_b;
#else
        lds_barrier();
#endif
    }
#ifdef RTS_SDP_STAMPS
    if (rg == RTS_SDP_STAMPS && sidx == 0 && lane == 0) {
        long long *dbg = reinterpret_cast<long long *>(pb.err) + 2;
        dbg[0] = st_steps;
        dbg[1] = st_bar;
        dbg[2] = (long long)__builtin_amdgcn_s_memtime() - st_all0;
        dbg[3] = st_n;
    }
#endif
}

// ---- backtrack over the packed, skewed step codes ----------------------------------------------------------------
// The path from (M-1, N-1) to (0, 0) is found strip by strip.  The DP left, for every cell of a strip's bottom row,
// the column at which its best path came in from the strip above (Problem::entb), so the columns at which the path
// crosses the strip boundaries follow from one dependent load per strip (hops); then every strip's segment is walked
// independently -- one wave per strip, all strips at once -- first to count its points, then to write them straight
// to their final, forward-ordered positions.
//
// A walk is a uniform (scalar) loop: the position lives in SGPRs, every lane keeps the code word of ITS row for the
// current 16-step chunk in a register and the step code is one v_readlane away, so no step waits for memory.  A
// window of kBtChunks chunks of the strip is staged in LDS (`win`: [2][kBtChunks][64] dwords) and the next window to
// the left is fetched into registers while the current one is being walked.
constexpr int kBtChunks = 8;

// Walks from (i, j) while the position stays inside strip (i >> 6) and has not reached (0, 0).  The k-th visited point
// (the start included, the first point outside the strip excluded) is written to out[2 * (out_base + k)] when `out` is
// not null -- in walk order, i.e. the path backwards.  Returns the number of visited points; (i, j) is left at the first
// position outside the strip (or at (0, 0), which counts as visited).
__device__ __forceinline__ int walk_strip(const uint32_t *codes, int N, int &i, int &j, int32_t *out, int out_base,
                                          uint32_t *win) {
    const int lane = threadIdx.x & 63;
    const int nch = n_chunks(N);
    const int strip = i >> 6;
    int n = 0, buf = 0, wlo = 0, plo = 0;
    bool have_pf = false;
    uint32_t pf[kBtChunks];
#pragma unroll
    for (int k = 0; k < kBtChunks; k++) pf[k] = 0;
    auto fetch = [&](int lo) {  // chunk indices below 0 or beyond nch-1 are never walked
#pragma unroll
        for (int k = 0; k < kBtChunks; k++) {
            const int c = lo + k;
            pf[k] = (c >= 0 && c < nch) ? codes[((size_t)strip * nch + c) * 64 + lane] : 0u;
        }
        plo = lo;
        have_pf = true;
    };
    auto commit = [&]() {  // registers -> the other LDS buffer
        buf ^= 1;
#pragma unroll
        for (int k = 0; k < kBtChunks; k++) win[(buf * kBtChunks + k) * 64 + lane] = pf[k];
        wlo = plo;
        have_pf = false;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    bool inside = true;
    while (inside) {
        const int c0 = (j + (i & 63)) >> 4;
        if (!(have_pf && c0 >= plo && c0 < plo + kBtChunks)) fetch(c0 - kBtChunks + 1);
        commit();
        fetch(wlo - kBtChunks);  // the window further left, in case the walk gets there
        int ccur = c0;
        uint32_t cur = win[(buf * kBtChunks + (c0 - wlo)) * 64 + lane];
        uint32_t nxt = (c0 - 1 >= wlo) ? win[(buf * kBtChunks + (c0 - 1 - wlo)) * 64 + lane] : 0u;
        for (;;) {
            // the current position is inside the strip: visit it
            if (out && lane == 0) *reinterpret_cast<int2 *>(out + 2 * (size_t)(out_base + n)) = make_int2(i, j);
            n++;
            if (i == 0 && j == 0) {
                inside = false;
                break;
            }
            const int l = i & 63, t = j + l;
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)cur, l);
            const int code = (w >> (2 * (t & 15))) & 3;
            i -= (code != kLeft) ? 1 : 0;  // kUp, kDiag
            j -= (code != kUp) ? 1 : 0;    // kLeft, kDiag
            j = j < 0 ? 0 : j;             // a corrupted code cannot leave the matrix
            if (i < 0) {
                i = 0;
                if (j == 0) {  // corrupted code at the origin row: stop instead of spinning
                    inside = false;
                    break;
                }
            }
            if ((i >> 6) != strip) {
                inside = false;
                break;
            }
            const int c = (j + (i & 63)) >> 4;
            if (c < wlo) break;  // restage further left
            if (c != ccur) {     // one chunk down (t shrinks by at most 2 per step)
                cur = nxt;
                ccur = c;
                nxt = (c - 1 >= wlo) ? win[(buf * kBtChunks + (c - 1 - wlo)) * 64 + lane] : 0u;
            }
        }
    }
    return n;
}

// One wave: cross[s] = column at which the path crosses the bottom row of strip s (s < S - 1), cross[S-1] = N - 1.
// entb[s][j] is the column at which the best path of (exit row of strip s, column j) came in from the row above the
// strip (the exit row is the bottom row, or the matrix's last row in the last strip), so the crossings follow from one
// dependent load per strip and no strip has to be walked.
__device__ __forceinline__ void path_hops(const uint32_t *codes, const int32_t *entb, int M, int N, int32_t *cross,
                                          uint32_t *win) {
    const int lane = threadIdx.x & 63;
    const int S = n_strips(M);
    int j = N - 1;
    if (lane == 0) cross[S - 1] = j;
    for (int s = S - 1; s >= 1; s--) {
        j = entb[(size_t)s * N + j];  // uniform load; one dependent round trip per strip
        j = j < 0 ? 0 : (j >= N ? N - 1 : j);
        if (lane == 0) cross[s - 1] = j;
    }
}

// The segment of strip s.  pass 0: one walk, which parks the segment's points -- backwards, as walked -- in `scratch` at
// pair 64 s + (column at which the path enters the strip) and leaves their number in lens[s]; the parking places of
// different strips cannot overlap (a segment has at most 64 + cross[s] - cross[s-1] points), and 64 n_strips + N pairs
// hold them all.  pass 1: with the lens of all strips known, a coalesced copy to the segment's place in path[] in
// forward order; *total (if not null, strip 0 only) receives the path length.
__host__ __device__ inline size_t scratch_pairs(int M, int N) { return (size_t)64 * n_strips(M) + N; }
__device__ __forceinline__ void path_segment(const uint32_t *codes, int M, int N, int s, const int32_t *cross,
                                             int32_t *lens, int pass, int32_t *path, int32_t *total, uint32_t *win,
                                             int32_t *scratch) {
    const int lane = threadIdx.x & 63;
    const int S = n_strips(M);
    const int park = 64 * s + (s > 0 ? cross[s - 1] : 0);
    if (pass == 0) {
        int i = (s == S - 1) ? M - 1 : 64 * s + 63, j = cross[s];
        const int n = walk_strip(codes, N, i, j, scratch, park, win);
        if (lane == 0) lens[s] = n;
        return;
    }
    int off = 0, all = 0;  // points in the strips above mine come first
    for (int q = lane; q < S; q += 64) {
        const int v = lens[q];
        all += v;
        off += (q < s) ? v : 0;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        off += __shfl_xor(off, d);
        all += __shfl_xor(all, d);
    }
    const int n = lens[s];
    const int2 *src = reinterpret_cast<const int2 *>(scratch) + park;
    int2 *dst = reinterpret_cast<int2 *>(path) + off;
    for (int q = lane; q < n; q += 64) dst[n - 1 - q] = src[q];
    if (total && s == 0 && lane == 0) *total = all;
}

// The whole backtrack of one problem by one workgroup of n_strips(M) waves (at most kTailStrips): crossings (wave 0),
// then every wave counts and writes its strip's segment.  For short problems this replaces three launches by one;
// long ones keep a workgroup per strip (path_hops / path_segment from separate kernels).  `win`: the workgroup's
// dynamic LDS, 2 * kBtChunks * 64 dwords per wave.
constexpr int kTailStrips = 12;  // 48 KB of dynamic LDS: below the 64 KB a launch gets without an attribute, static LDS included
__host__ __device__ inline size_t tail_lds_bytes(int S) { return sizeof(uint32_t) * 2 * kBtChunks * 64 * (size_t)S; }
__device__ __forceinline__ void path_tail(const uint32_t *codes, const int32_t *entb, int M, int N, int32_t *cross,
                                          int32_t *lens, int32_t *path, int32_t *total, uint32_t *win, int32_t *scratch) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int S = n_strips(M);
    uint32_t *mywin = win + (size_t)wave * 2 * kBtChunks * 64;
    if (wave == 0) path_hops(codes, entb, M, N, cross, mywin);
    __syncthreads();
    if (wave < S) path_segment(codes, M, N, wave, cross, lens, 0, path, total, mywin, scratch);
    __syncthreads();
    if (wave < S) path_segment(codes, M, N, wave, cross, lens, 1, path, total, mywin, scratch);
}

}  // namespace sdp
}  // namespace rts
