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
// A workgroup of NW waves owns a ROW GROUP of NW consecutive strips; wave w runs kLag chunks (80
// steps) behind wave w-1 and takes the bottom row of the strip above from a small LDS ring.  All waves
// of a workgroup advance in lockstep, one LDS-only barrier per chunk of 16 steps, which also publishes
// the column features that one wave stages into an LDS ring for everyone (14-double records: a 112-byte
// stride is conflict-free for ds_read_b128).  Consecutive row groups run on different workgroups of the
// same launch, pipelined the same way through HBM: the last wave publishes its strip's bottom row as
// 8-byte write-through (sc1) stores, the next row group's first wave polls those words with sc1 loads.
// The data is its own flag: the boundary buffer is pre-filled with a signalling-NaN bit pattern that no
// float64 add/subtract can produce (arithmetic results are always quiet), so a word that differs from
// it is the value (single naturally aligned 8-byte store: untorn; no fences, no separate flag).
//
// Outputs.  Back-pointers: 2 bits per cell, 16 steps of a lane packed into one dword, stored in skewed
// order [strip][chunk][lane] (one coalesced 256-byte store per wave and chunk) -- the backtrack below
// decodes that layout.  The accumulated-cost matrix (dtw.py's acc_cost, wtw.py's D), when wanted, is
// transposed through a per-wave LDS tile so that HBM sees 128-byte row segments instead of 64 scattered
// 8-byte stores per step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace rts {
namespace sdp {

constexpr int kF = 12;
constexpr int kChunk = 16;     // steps between two workgroup barriers
constexpr int kLag = 5;        // chunks wave w+1 runs behind wave w: lane 63 finishes column 16m+15 at step 16m+78
constexpr int kYRec = 14;      // doubles per column record in the LDS ring: 12 features, norm, pad
constexpr int kBRing = 128;    // boundary ring entries between consecutive waves
constexpr int kStageLd = 65;   // leading dimension of the [16][64] output staging tile
constexpr int kMaxWaves = 8;
constexpr unsigned long long kSentinel = 0x7FF4DEAD7FF4DEADull;  // signalling NaN: never an arithmetic result
constexpr uint32_t kSentinel32 = 0x7FF4DEADu;                    // the same as a 32-bit fill pattern
constexpr int kSpinLimit = 1 << 22;

// internal 2-bit step codes
constexpr int kLeft = 0;  // from (i, j-1)
constexpr int kUp = 1;    // from (i-1, j)
constexpr int kDiag = 2;  // from (i-1, j-1)

__host__ __device__ inline int n_strips(int M) { return (M + 63) / 64; }
__host__ __device__ inline int n_chunks(int N) { return (N + 63 + kChunk - 1) / kChunk; }  // per strip
__host__ __device__ inline int yring_slots(int NW) { return kLag * kChunk * (NW - 1) + 112; }
__host__ __device__ inline size_t lds_bytes(int NW, bool stage) {
    return sizeof(double) * ((size_t)(yring_slots(NW) + kChunk) * kYRec + (size_t)NW * kBRing +
                             (stage ? (size_t)NW * kChunk * kStageLd : 0));
}
__host__ __device__ inline size_t codes_words(int M, int N) { return (size_t)n_strips(M) * n_chunks(N) * 64; }

struct Problem {
    const void *x;  // row features [M][12]
    const void *y;  // column features [N][12]
    int x_f64, y_f64;
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

// ---- policies: the per-cell arithmetic of the two reference recurrences ---------------------------------------

// dtw.py:11 cost = 1 - seq_a.T . seq_b (dgemm k-order: one fma chain), dtw.py:32-40 options (left, up, diag + 2c),
// np.argmin = first minimum.
struct DtwPolicy {
    static constexpr bool kNorm = false;
    static __device__ __forceinline__ double norm(const double (&)[kF]) { return 0.0; }
    static __device__ __forceinline__ double cost(const double (&x)[kF], double, const double (&y)[kF], double) {
        double s = 0.0;
#pragma unroll
        for (int f = 0; f < kF; f++) s = fma(x[f], y[f], s);
        return 1.0 - s;
    }
    static __device__ __forceinline__ void cell(bool first_row, bool first_col, double up, double left, double diag,
                                                double c, double &dv, int &code) {
        const double o0 = left + c, o1 = up + c, o2 = diag + 2 * c;
        double best = o0;
        int s = kLeft;
        if (o1 < best) {
            best = o1;
            s = kUp;
        }
        if (o2 < best) {
            best = o2;
            s = kDiag;
        }
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
    static __device__ __forceinline__ void cell(bool first_row, bool first_col, double up, double left, double diag,
                                                double c, double &dv, int &code) {
        double mc = up;
        int s = kUp;
        if (left < mc) {
            mc = left;
            s = kLeft;
        }
        if (diag < mc) {
            mc = diag;
            s = kDiag;
        }
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

// ---- one row group (NW strips) of one problem, executed by the whole workgroup ----------------------------------
template <class P, bool STAGE>
__device__ __forceinline__ void run_rowgroup(const Problem &pb, int rg, int n_rg, int NW, unsigned char *smem) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int M = pb.M, N = pb.N;
    const int YR = yring_slots(NW);
    double *yring = reinterpret_cast<double *>(smem);  // [YR + 16] records; the last 16 mirror the first 16
    double *bring = yring + (size_t)(YR + kChunk) * kYRec;
    double *stage = bring + (size_t)NW * kBRing;
    double *bring_w = bring + wave * kBRing;                              // my strip's bottom row
    const double *bring_up = bring + (wave > 0 ? wave - 1 : 0) * kBRing;  // the strip above
    double *stage_w = stage + (STAGE ? wave * kChunk * kStageLd : 0);

    const int strip = rg * NW + wave;
    const bool strip_ok = strip * 64 < M;
    const int i = strip * 64 + lane;
    const int nch = n_chunks(N);
    const int total = nch + kLag * (NW - 1);
    const bool from_hbm = (wave == 0 && rg > 0);
    const bool to_hbm = (wave == NW - 1 && rg + 1 < n_rg);
    const unsigned long long *bnd_in = pb.bnd + (size_t)(rg > 0 ? rg - 1 : 0) * N;
    unsigned long long *bnd_out = pb.bnd + (size_t)rg * N;

    double x[kF];
    {
        const int ic = i < M ? i : M - 1;
        load_frame(pb.x, pb.x_f64, ic, x);
    }
    const double nx = P::norm(x);

    auto stage_columns = [&](int first_col) {  // 16 lanes: one column record each
        const int col = first_col + lane;
        if (lane < kChunk && col < N) {
            double y[kF];
            load_frame(pb.y, pb.y_f64, col, y);
            const double ny = P::kNorm ? P::norm(y) : 0.0;
            const int slot = col % YR;
            double2 *rec = reinterpret_cast<double2 *>(yring + (size_t)slot * kYRec);
#pragma unroll
            for (int k = 0; k < kF / 2; k++) rec[k] = make_double2(y[2 * k], y[2 * k + 1]);
            if (P::kNorm) rec[kF / 2] = make_double2(ny, 0.0);
            if (slot < kChunk) {
                rec += (size_t)YR * kYRec / 2;
#pragma unroll
                for (int k = 0; k < kF / 2; k++) rec[k] = make_double2(y[2 * k], y[2 * k + 1]);
                if (P::kNorm) rec[kF / 2] = make_double2(ny, 0.0);
            }
        }
    };
    if (wave == 0) stage_columns(0);
    lds_barrier();

    double prev = 0.0, upprev = 0.0;
    int ent = 0, ent_upprev = 0;  // entry column (see Problem::entb) of my previous cell / of the cell diagonally above it
    int yslot = (lane == 0) ? 0 : YR - lane;  // ring slot of column 16m - lane at m = 0
    unsigned long long next_bits = kSentinel;
    bool dead = false;
    if (from_hbm && lane < kChunk && lane < N) next_bits = load_sc1(bnd_in + lane);

    for (int k = 0; k < total; k++) {
        const int m = k - kLag * wave;  // my strip's chunk
        if (strip_ok && m >= 0 && m < nch) {
            // ---- the row above, columns [16m, 16m+16): lane q holds column 16m+q
            double upbuf = 0.0;
            if (from_hbm) {
                const int col = kChunk * m + lane;
                const bool want = lane < kChunk && col < N;
                unsigned long long v = next_bits;
                int spins = 0;
                while (!dead && __any(want && v == kSentinel)) {
                    __builtin_amdgcn_s_sleep(2);
                    if (want && v == kSentinel) v = load_sc1(bnd_in + col);
                    if (++spins > kSpinLimit) {
                        dead = true;
                        if (lane == 0) atomicExch(pb.err, 1);
                    }
                }
                upbuf = __longlong_as_double((long long)v);
                const int ncol = col + kChunk;
                next_bits = kSentinel;
                if (lane < kChunk && ncol < N) next_bits = load_sc1(bnd_in + ncol);  // consumed one chunk later
            } else if (wave > 0) {
                if (lane < kChunk) upbuf = bring_up[(kChunk * m + lane) & (kBRing - 1)];
            }
            const int jneg = lane - kChunk * m;  // column of step q is q - jneg
            const double *ybase = yring + (size_t)yslot * kYRec;
            uint32_t codes = 0;
            double collect = 0.0;  // lane q: my lane 63's value of step q (bottom row, column 16m + q - 63)
            int ecollect = 0;      // the same for its entry column

            // 16 branch-free steps.  FIRST: this strip holds matrix row 0 (lane 0); COL0: some lane is at column 0.
            auto steps = [&](auto first_c, auto col0_c) {
                constexpr bool FIRST = decltype(first_c)::value;
                constexpr bool COL0 = decltype(col0_c)::value;
                const bool first_row = FIRST && lane == 0;
                auto cost_of = [&](int q) {
                    const double2 *rec = reinterpret_cast<const double2 *>(ybase + q * kYRec);
                    double y[kF];
#pragma unroll
                    for (int t = 0; t < kF / 2; t++) {
                        const double2 r = rec[t];
                        y[2 * t] = r.x;
                        y[2 * t + 1] = r.y;
                    }
                    double ny = 0.0;
                    if (P::kNorm) ny = rec[kF / 2].x;
                    return P::cost(x, nx, y, ny);
                };
                double c = cost_of(0);
                static_for<0, kChunk>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    double cn = 0.0;
                    if (q + 1 < kChunk) cn = cost_of(q + 1);  // independent of this step's recurrence
                    const double up = shr1(prev, readlane_d(upbuf, q));
                    double dv;
                    int code;
                    P::cell(first_row, COL0 && (q == jneg), up, prev, upprev, c, dv, code);
                    upprev = up;
                    prev = dv;
                    codes |= (uint32_t)code << (2 * q);
                    asm("" : "+v"(codes));  // materialise now: do not keep 48 lane masks alive
                    // where did this cell's best path enter the strip?  Lane 0's upper neighbours are in the row above:
                    // the entry column is that neighbour's own column (16m + q for "up", one less for "diagonal", which is
                    // exactly what lane 0 received as ent_up one step earlier)
                    const int ent_up = shr1_i(ent, kChunk * m + q);
                    ent = (code == kLeft) ? ent : ((code == kUp) ? ent_up : ent_upprev);
                    ent_upprev = ent_up;
                    {
                        const int lo = __builtin_amdgcn_readlane(__double2loint(dv), 63);
                        const int hi = __builtin_amdgcn_readlane(__double2hiint(dv), 63);
                        const int el = __builtin_amdgcn_readlane(ent, 63);
                        int clo = __double2loint(collect), chi = __double2hiint(collect);
                        asm("v_writelane_b32 %0, %1, %2" : "+v"(clo) : "s"(lo), "n"(q));
                        asm("v_writelane_b32 %0, %1, %2" : "+v"(chi) : "s"(hi), "n"(q));
                        asm("v_writelane_b32 %0, %1, %2" : "+v"(ecollect) : "s"(el), "n"(q));
                        collect = __hiloint2double(chi, clo);
                    }
                    if (STAGE) stage_w[q * kStageLd + lane] = dv;
                    c = cn;
                });
            };
            const bool first_strip = (strip == 0);
            const bool col0 = (kChunk * m < 64);
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
            yslot += kChunk;
            if (yslot >= YR) yslot -= YR;

            pb.codes[((size_t)strip * nch + m) * 64 + lane] = codes;
            // bottom-row columns finished in this chunk: [16m - 63, 16m - 48], held by lanes 0..15 of `collect`
            {
                const int col = kChunk * m - 63 + lane;
                if (lane < kChunk) {
                    bring_w[col & (kBRing - 1)] = collect;
                    if (col >= 0 && col < N) {
                        pb.entb[(size_t)strip * N + col] = ecollect;
                        if (to_hbm) store_sc1(bnd_out + col, (unsigned long long)__double_as_longlong(collect));
                    }
                }
            }
            if (STAGE) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < kChunk; it++) {
                    const int r = it * 4 + (lane >> 4), qq = lane & 15;
                    const double v = stage_w[qq * kStageLd + r];
                    const int row = strip * 64 + r, col = kChunk * m + qq - r;
                    if (row < M && col >= 0 && col < N) pb.D[(size_t)row * pb.ldD + col] = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- column records for the next chunk (needed first by wave 0, lane 0)
        if (wave == k % NW) stage_columns(kChunk * (k + 1));
        lds_barrier();
    }
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

// Walks from (i, j) while the position stays inside strip (i >> 6) and has not reached (0, 0).  Visited points
// (the start included, the first point outside the strip excluded) are written to out[2 * (out_last - k)] for the
// k-th visited point when `out` is not null.  Returns the number of visited points; (i, j) is left at the first
// position outside the strip (or at (0, 0), which counts as visited).
__device__ __forceinline__ int walk_strip(const uint32_t *codes, int N, int &i, int &j, int32_t *out, int out_last,
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
            if (out && lane == 0) *reinterpret_cast<int2 *>(out + 2 * (size_t)(out_last - n)) = make_int2(i, j);
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
__device__ __forceinline__ void path_hops(const uint32_t *codes, const int32_t *entb, int M, int N, int32_t *cross,
                                          uint32_t *win) {
    const int lane = threadIdx.x & 63;
    const int S = n_strips(M);
    int i = M - 1, j = N - 1;
    if (lane == 0) cross[S - 1] = j;
    if (S == 1) return;
    walk_strip(codes, N, i, j, nullptr, 0, win);  // leaves the last strip at (64 (S-1) - 1, j)
    for (int s = S - 2; s >= 0; s--) {
        if (lane == 0) cross[s] = j;
        if (s > 0) j = entb[(size_t)s * N + j];  // uniform load; one dependent round trip per strip
        j = j < 0 ? 0 : (j >= N ? N - 1 : j);
    }
}

// The segment of strip s: start point and walk.  pass 0: lens[s] = number of points.  pass 1: writes the points to
// path[] (forward order) using the lens of all strips; *total (if not null, strip 0 only) receives the path length.
__device__ __forceinline__ void path_segment(const uint32_t *codes, int M, int N, int s, const int32_t *cross,
                                             int32_t *lens, int pass, int32_t *path, int32_t *total, uint32_t *win) {
    const int lane = threadIdx.x & 63;
    const int S = n_strips(M);
    int i = (s == S - 1) ? M - 1 : 64 * s + 63, j = cross[s];
    if (pass == 0) {
        const int n = walk_strip(codes, N, i, j, nullptr, 0, win);
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
    walk_strip(codes, N, i, j, path, off + n - 1, win);
    if (total && s == 0 && lane == 0) *total = all;
}

}  // namespace sdp
}  // namespace rts
