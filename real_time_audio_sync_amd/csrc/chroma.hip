// Chroma front end for gfx950: frame -> Hann -> real FFT -> |.|^2 -> 12-bin filterbank -> L2 normalise.
//
// Reference: /root/reference/chroma.py:35-90 (create_stft, create_chroma, wav_to_chroma_col,
// wav_to_chroma_diff) and the per-hop half of wtw.WTW (wtw.py:37-41, :81-90).
//
//   chroma_frames_kernel   one team of 256 threads per frame, two teams (two frames in flight) per workgroup,
//                          persistent over groups of four frames:
//       1. L samples (zero-padded on the left by `pad_left`, chroma.py:49) x window -> LDS as L/2
//          packed complex float64 (even sample = re, odd = im);
//       2. L/2-point complex Stockham radix-4 (+ one radix-2 stage) FFT in LDS (float64; twiddles exp(-2 pi i n / L),
//          n < L/2, are tabulated once per workgroup in LDS from a host-computed table);
//       3. real-FFT untangling -> the L/2+1 rfft bins, optionally stored (create_stft's output),
//          power spectrum -> LDS;
//       4. projection onto the 12 x (L/2+1) filterbank: each thread owns bins k = tid (mod 256)
//          with 12 running sums, a cross-wave LDS reduction finishes them (a 12-row GEMV per
//          frame: MFMA would run at the fp64 vector rate with 3/4 of a 16-wide tile empty, so a
//          plain reduction it is);
//       5. column L2 normalisation with librosa's tiny-norm rule (chroma.py:74).
//   chroma_project_kernel  steps 4-5 from a power spectrum already in HBM (create_chroma(ft)).
//   chroma_diff_kernel     clip(diff(chroma), 0, inf)  (chroma.py:85-90).
//
// float64 throughout: the reference computes in float64 and the alignment paths downstream must
// not move.  numpy's pocketfft orders its additions differently, so chroma values agree with the
// oracle to ~1e-13 relative, not bitwise (tolerance stated in tests/test_chroma_gpu.py).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace rts {

constexpr int kCh = 12;
constexpr int kChromaNT = 256;

struct ChromaArgs {
    const void *samples;     // [n_samples]
    const double *window;    // [L]
    const double2 *twiddle;  // [L/2]  exp(-2 pi i n / L)
    const double *fb;        // [12][L/2+1]
    const double *fbt;       // [L/2+1][12]: the same, bin-major (a thread's 12 weights of one bin are 96 contiguous bytes)
    void *chroma_out;        // [n_frames][12]
    double2 *stft_out;       // [n_frames][L/2+1] or NULL
    const double *spec_in;   // projection-only entry: [n_frames][L/2+1]
    long long n_samples;
    long long frame_offset;  // sample index of frame 0, element 0 (= -pad_left)
    int L, logL2, hop, n_frames, normalize, samples_f64, out_f64;
    // batched form (blockIdx.y = stream): per-stream sample counts / frame counts, strides in elements
    const int32_t *n_samples_b;  // [B] or NULL
    const int32_t *n_frames_b;   // [B] or NULL
    long long sample_stride;     // samples between consecutive streams
    int out_frames_stride;       // frames between consecutive streams in chroma_out
};

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

constexpr int kChromaFR = 4;  // frames sharing one pass over the filterbank
#ifdef RTS_CHROMA_STAMPS
__device__ long long g_chroma_stamps[8];
#define CH_STAMP(i)                                                        \
    do {                                                                   \
        const long long now_ = (long long)__builtin_amdgcn_s_memtime();    \
        if (threadIdx.x == 0 && blockIdx.x == 0) st_[i] += now_ - last_;   \
        last_ = now_;                                                      \
    } while (0)
#else
#define CH_STAMP(i) \
    do {            \
    } while (0)
#endif

// Sum over each row of 16 consecutive lanes, left in the row's last lane (row_shr 1, 2, 4, 8 with zero fill).
__device__ __forceinline__ double row_sum16(double v) {
#define RTS_ROW_STEP(CTRL)                                                                                     \
    do {                                                                                                       \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);                \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);                \
        v = v + __hiloint2double(hi, lo);                                                                      \
    } while (0)
    RTS_ROW_STEP(0x111);
    RTS_ROW_STEP(0x112);
    RTS_ROW_STEP(0x114);
    RTS_ROW_STEP(0x118);
#undef RTS_ROW_STEP
    return v;
}

// Steps 4-5 for up to kProjMax frames whose power spectra sit in LDS: frames f = f_first + i * f_step (i < kProjMax)
// with f < nf, spectrum at spec + f * spec_stride, output frame frame0 + f.  One team of 256 threads (`tid` is the
// index within the team; `bar` synchronises at least the team): each thread owns bins k = tid (mod 256) and keeps 12
// running sums per frame, so the 197 KB filterbank is read from L2 once per call.  The 12 x 256 partial sums of a
// frame are then reduced in a fixed order, independent of how many frames share the pass (`red`: kProjMax * 192
// doubles of LDS scratch).  Every thread executes every barrier, whatever nf is.
template <int kProjMax, class Barrier>
__device__ __forceinline__ void project_normalize(const ChromaArgs &g, const double *spec, int spec_stride,
                                                  double *red, int frame0, int nf, int f_first, int f_step, int tid,
                                                  Barrier bar) {
    const int nb = g.L / 2 + 1;
#ifdef RTS_CHROMA_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) g_chroma_stamps[7] = (long long)__builtin_amdgcn_s_memtime();
#endif
    double acc[kProjMax][kCh];
#pragma unroll
    for (int i = 0; i < kProjMax; i++)
#pragma unroll
        for (int p = 0; p < kCh; p++) acc[i][p] = 0.0;
    if (f_first < nf) {
        // the 12 weights of the thread's next bin are fetched (L2) while the current bin is accumulated: two register
        // sets that swap roles by name (a copy would have to wait for the load it copies)
        double wa[kCh], wb[kCh];
        auto load_w = [&](double (&w)[kCh], int k) {
            const int kc = k < nb ? k : nb - 1;
            const double2 *src = reinterpret_cast<const double2 *>(g.fbt) + (size_t)kc * (kCh / 2);
#pragma unroll
            for (int p = 0; p < kCh / 2; p++) {
                const double2 t = src[p];
                w[2 * p] = t.x;
                w[2 * p + 1] = t.y;
            }
        };
        auto accumulate = [&](const double (&w)[kCh], int k) {
            if (k >= nb) return;
#pragma unroll
            for (int i = 0; i < kProjMax; i++) {
                const int f = f_first + i * f_step;
                if (f < nf) {
                    const double sv = spec[(size_t)f * spec_stride + k];
#pragma unroll
                    for (int p = 0; p < kCh; p++) acc[i][p] = fma(w[p], sv, acc[i][p]);
                }
            }
        };
        load_w(wa, tid);
        for (int k = tid; k < nb; k += 2 * kChromaNT) {
            load_w(wb, k + kChromaNT);
            accumulate(wa, k);
            load_w(wa, k + 2 * kChromaNT);
            accumulate(wb, k + kChromaNT);
        }
    }
#ifdef RTS_CHROMA_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) g_chroma_stamps[6] += (long long)__builtin_amdgcn_s_memtime() - g_chroma_stamps[7];
#endif
    // Reduction of the 256 partial sums of every (frame, pitch class): 16 consecutive threads by a DPP row reduction
    // (no LDS, no barrier), the 16 row totals through LDS in index order.  All frames of the pass share the two barriers.
    double *part = red;  // [kProjMax][12][16] row totals
#pragma unroll
    for (int i = 0; i < kProjMax; i++) {
#pragma unroll
        for (int p = 0; p < kCh; p++) {
            const double t = row_sum16(acc[i][p]);
            if ((tid & 15) == 15) part[(i * kCh + p) * 16 + (tid >> 4)] = t;
        }
    }
    bar();
    if (tid < 64) {  // one wave: lane i*12 + p finishes (frame i, pitch class p); then the norms
        double cp = 0.0;
        if (tid < kProjMax * kCh) {
#pragma unroll
            for (int q = 0; q < 16; q++) cp = cp + part[tid * 16 + q];
        }
#pragma unroll
        for (int i = 0; i < kProjMax; i++) {
            const int f = f_first + i * f_step;
            double ss = 0.0;
#pragma unroll
            for (int p = 0; p < kCh; p++) {
                const double c = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cp), i * kCh + p),
                                                  __builtin_amdgcn_readlane(__double2loint(cp), i * kCh + p));
                const double sq = c * c;
                ss = ss + sq;
            }
            double len = sqrt(ss);
            if (!g.normalize || len < 2.2250738585072014e-308) len = 1.0;  // librosa.util.normalize, fill=None
            if (f < nf && tid >= i * kCh && tid < (i + 1) * kCh) {
                const double v = cp / len;
                const size_t o = (size_t)(frame0 + f) * kCh + (tid - i * kCh);
                if (g.out_f64)
                    reinterpret_cast<double *>(g.chroma_out)[o] = v;
                else
                    reinterpret_cast<float *>(g.chroma_out)[o] = (float)v;
            }
        }
    }
    bar();
}

// Two teams of 256 threads per workgroup, each transforming its own frame (two waves per SIMD: the FFT is a chain of
// short dependent phases -- LDS round trip, a few dozen flops, barrier -- and a second frame fills the gaps).  A group
// of kChromaFR = 4 consecutive frames is handled in two rounds (team h takes frames h and h + 2) and each team then
// projects its two spectra in one pass over the filterbank.
constexpr int kChromaWG = 2 * kChromaNT;

template <typename ST>  // sample type in HBM: float or double
__global__ void __launch_bounds__(kChromaWG) chroma_frames_kernel(ChromaArgs g) {
    extern __shared__ __align__(16) unsigned char ch_smem[];
    const int L = g.L, N2 = L / 2, NQ = N2 / 2;
    const int sstride = N2 + 2;                               // doubles per power spectrum in LDS
    const int team = threadIdx.x >> 8, tid = threadIdx.x & (kChromaNT - 1);
    double2 *zbase = reinterpret_cast<double2 *>(ch_smem);    // [2][N2] FFT work buffers, one per team
    double2 *z = zbase + (size_t)team * N2;
    double2 *twh = zbase + 2 * (size_t)N2;                    // [N2/2]  exp(-2 pi i n / L), n < L/4
    double *spec = reinterpret_cast<double *>(twh + NQ);      // [kChromaFR][sstride]
    // reduction scratch of a team ([12][256] + [192] doubles): its FFT buffer when that is large enough (it is free by
    // then), a region of its own for short transforms
    double *red = (2 * N2 >= 3264) ? reinterpret_cast<double *>(z)
                                   : spec + (size_t)kChromaFR * sstride + (size_t)team * 3264;

    // quarter-circle twiddle table; the second quarter follows by a rotation: exp(-i (x + pi/2)) = -i exp(-i x)
    for (int n = threadIdx.x; n < NQ; n += kChromaWG) twh[n] = g.twiddle[n];
    __syncthreads();
    auto tw = [&](int t) {  // t < N2
        const double2 w = twh[t & (NQ - 1)];
        return (t & NQ) ? make_double2(w.y, -w.x) : w;
    };

    // batched launch: this workgroup's stream
    const int sb = blockIdx.y;
    if (g.n_frames_b) {
        g.n_frames = g.n_frames_b[sb];
        g.n_samples = g.n_samples_b[sb];
        const long long so = (long long)sb * g.sample_stride;
        g.samples = reinterpret_cast<const ST *>(g.samples) + so;
        const long long oo = (long long)sb * g.out_frames_stride * kCh;
        g.chroma_out = g.out_f64 ? (void *)(reinterpret_cast<double *>(g.chroma_out) + oo)
                                 : (void *)(reinterpret_cast<float *>(g.chroma_out) + oo);
    }

    // (the window coefficients are read from the L2-resident table where they are used: keeping 16 of them per thread in
    // registers, next to the prefetched samples, pushed this kernel into scratch -- 176 B per lane with float64 samples)
    constexpr int kMaxPairs = 8;  // N2/256 <= 8 for L <= 4096

    // Samples of the team's next frame, fetched one frame ahead.  Nothing is done to a loaded value in the frame that
    // fetches it (not even float -> double): a use would make the wave wait for the load right there.  Branch-free:
    // indices are clamped into the buffer; out-of-range positions (the zero padding on the left, the end of the
    // signal, frames past the last one) are zeroed when the values are used.
    ST ps0[kMaxPairs], ps1[kMaxPairs];
#pragma unroll
    for (int r = 0; r < kMaxPairs; r++) ps0[r] = ps1[r] = (ST)0;
    auto clamp_idx = [&](long long idx) {
        const long long hi = g.n_samples > 0 ? g.n_samples - 1 : 0;
        return idx < 0 ? 0 : (idx > hi ? hi : idx);
    };
    auto fetch_frame = [&](long long frame) {
        const long long s0 = g.frame_offset + frame * g.hop;
#pragma unroll
        for (int r = 0; r < kMaxPairs; r++) {
            const int n = tid + r * kChromaNT;
            ps0[r] = reinterpret_cast<const ST *>(g.samples)[clamp_idx(s0 + 2 * n)];
            ps1[r] = reinterpret_cast<const ST *>(g.samples)[clamp_idx(s0 + 2 * n + 1)];
        }
    };
    fetch_frame((long long)blockIdx.x * kChromaFR + team);
#ifdef RTS_CHROMA_STAMPS
    long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long last_ = (long long)__builtin_amdgcn_s_memtime();
#endif
    for (int frame0 = blockIdx.x * kChromaFR; frame0 < g.n_frames; frame0 += gridDim.x * kChromaFR) {
        const int nf = (g.n_frames - frame0 < kChromaFR) ? g.n_frames - frame0 : kChromaFR;
        for (int round = 0; round < kChromaFR / 2; round++) {
            const int f = 2 * round + team;   // my team's frame within the group
            const int frame = frame0 + f;
            const bool live = f < nf;         // team-uniform; a dead team still takes part in every barrier
            CH_STAMP(0);
            // 1. window the samples fetched one frame ago (registers), pack as complex; then fetch the team's next
            //    frame: its HBM latency hides behind this frame's FFT
#pragma unroll
            for (int r = 0; r < kMaxPairs; r++) {
                const int n = tid + r * kChromaNT;
                if (n >= N2) break;
                const long long sidx = g.frame_offset + (long long)frame * g.hop + 2 * n;
                const double x0 = (sidx >= 0 && sidx < g.n_samples) ? (double)ps0[r] : 0.0;
                const double x1 = (sidx + 1 >= 0 && sidx + 1 < g.n_samples) ? (double)ps1[r] : 0.0;
                z[n] = make_double2(x0 * g.window[2 * n], x1 * g.window[2 * n + 1]);
            }
            // (within the group only: a prefetch kept across the projection -- 16 / 32 more live registers next to its 2 x 12
            // accumulators and two weight sets -- is what used to push this kernel into scratch)
            if (round + 1 < kChromaFR / 2) fetch_frame((long long)frame + 2);
            lds_barrier();  // LDS traffic only: the sample prefetch stays in flight
            CH_STAMP(1);
            // 2. Stockham autosort FFT, N2 points, in place via registers (read all, barrier, write all): one radix-2
            //    stage when log2(N2) is odd, then radix-4 stages.  Stage with sub-transform length p: butterfly
            //    i = s*p + k reads i + m*N2/r (m < r), twiddles exp(-2 pi i k m / (r p)), writes s*(r p) + k + m*p.
            int p = 1;
            if ((31 - __clz(N2)) & 1) {  // log2(N2) odd
                const int half = N2 / 2;
                constexpr int kMaxBf = 4;  // N2/2/256 <= 4 for L <= 4096
                double2 o0[kMaxBf], o1[kMaxBf];
#pragma unroll
                for (int r = 0; r < kMaxBf; r++) {
                    const int i = tid + r * kChromaNT;
                    if (i < half) {
                        const double2 u0 = z[i], u1 = z[i + half];  // p = 1: k = 0, twiddle 1
                        o0[r] = make_double2(u0.x + u1.x, u0.y + u1.y);
                        o1[r] = make_double2(u0.x - u1.x, u0.y - u1.y);
                    }
                }
                lds_barrier();
#pragma unroll
                for (int r = 0; r < kMaxBf; r++) {
                    const int i = tid + r * kChromaNT;
                    if (i < half) {
                        z[2 * i] = o0[r];
                        z[2 * i + 1] = o1[r];
                    }
                }
                lds_barrier();
                p = 2;
            }
            const int q = N2 / 4;
            for (; p < N2; p <<= 2) {
                constexpr int kMaxBf4 = 2;  // N2/4/256 <= 2 for L <= 4096
                double2 o[kMaxBf4][4];
                int jj[kMaxBf4];
                const int tstep = N2 / (2 * p);
#pragma unroll
                for (int r = 0; r < kMaxBf4; r++) {
                    const int i = tid + r * kChromaNT;
                    if (i < q) {
                        const int k = i & (p - 1);
                        const int t1 = k * tstep, t2 = 2 * t1, t3 = 3 * t1;  // t1 < N2/2, t2 < N2, t3 < 3 N2/2
                        const double2 w1 = tw(t1), w2 = tw(t2);
                        double2 w3 = tw(t3 & (N2 - 1));
                        if (t3 >= N2) w3 = make_double2(-w3.x, -w3.y);  // exp(-i (pi + x)) = -exp(-i x)
                        const double2 u0 = z[i];
                        const double2 u1 = cmul(w1, z[i + q]);
                        const double2 u2 = cmul(w2, z[i + 2 * q]);
                        const double2 u3 = cmul(w3, z[i + 3 * q]);
                        const double2 a0 = make_double2(u0.x + u2.x, u0.y + u2.y);
                        const double2 a1 = make_double2(u0.x - u2.x, u0.y - u2.y);
                        const double2 a2 = make_double2(u1.x + u3.x, u1.y + u3.y);
                        const double2 a3 = make_double2(u1.y - u3.y, -(u1.x - u3.x));  // -i (u1 - u3)
                        o[r][0] = make_double2(a0.x + a2.x, a0.y + a2.y);
                        o[r][1] = make_double2(a1.x + a3.x, a1.y + a3.y);
                        o[r][2] = make_double2(a0.x - a2.x, a0.y - a2.y);
                        o[r][3] = make_double2(a1.x - a3.x, a1.y - a3.y);
                        jj[r] = ((i - k) << 2) + k;
                    }
                }
                lds_barrier();
#pragma unroll
                for (int r = 0; r < kMaxBf4; r++) {
                    const int i = tid + r * kChromaNT;
                    if (i < q) {
                        z[jj[r]] = o[r][0];
                        z[jj[r] + p] = o[r][1];
                        z[jj[r] + 2 * p] = o[r][2];
                        z[jj[r] + 3 * p] = o[r][3];
                    }
                }
                lds_barrier();
            }
            CH_STAMP(2);
            // 3. untangle: X[k] = E[k] + W_L^k O[k], E = (Z[k] + conj Z[N2-k]) / 2, O = (Z[k] - conj Z[N2-k]) / (2i)
            const int nb = N2 + 1;
            double *sp = spec + (size_t)f * sstride;
            if (live) {
                for (int k = tid; k < nb; k += kChromaNT) {
                    const double2 a = z[k & (N2 - 1)];          // Z[N2] == Z[0]
                    const double2 bq = z[(N2 - k) & (N2 - 1)];
                    const double2 b = make_double2(bq.x, -bq.y);  // conj
                    const double2 e = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y + b.y));
                    const double2 dm = make_double2(a.x - b.x, a.y - b.y);
                    const double2 o = make_double2(0.5 * dm.y, -0.5 * dm.x);  // dm / (2i)
                    const double2 w = (k < N2) ? tw(k) : make_double2(-1.0, 0.0);
                    const double2 wo = cmul(w, o);
                    const double2 x = make_double2(e.x + wo.x, e.y + wo.y);
                    if (g.stft_out) g.stft_out[(size_t)frame * nb + k] = x;
                    sp[k] = x.x * x.x + x.y * x.y;
                }
            }
            lds_barrier();
            CH_STAMP(3);
        }
        // 4-5: each team projects its frames (f = team, team + 2) in one pass over the filterbank; its own FFT buffer
        //      is free now and serves as reduction scratch
        if (g.chroma_out)
            project_normalize<kChromaFR / 2>(g, spec, sstride, red, frame0, nf, team, 2, tid, [] { lds_barrier(); });
        fetch_frame((long long)frame0 + (long long)gridDim.x * kChromaFR + team);  // the next group's first frame
        CH_STAMP(4);
    }
#ifdef RTS_CHROMA_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int i = 0; i < 8; i++) g_chroma_stamps[i] = st_[i];
#endif
}

// ---- fft_len = 8192 (wtw.py:27 takes any length; the reference's own configurations stop at 4096) --------------------
// The 4096 packed complex points of one frame fill 64 KB of LDS, so a workgroup is ONE team of 256 threads with one
// frame in flight; two consecutive frames share a pass over the (394 KB, L2-resident) filterbank.  No sample prefetch
// and the twiddles come from the L2-resident table instead of an LDS copy (the work buffer, two power spectra and the
// reduction scratch leave no room for it): this kernel exists for completeness, not for speed.
constexpr int kBigFR = 2;

template <typename ST>
__global__ void __launch_bounds__(kChromaNT) chroma_frames_big_kernel(ChromaArgs g) {
    extern __shared__ __align__(16) unsigned char ch_smem[];
    const int L = g.L, N2 = L / 2, NQ = N2 / 2;  // 8192, 4096, 2048
    const int sstride = N2 + 2;
    const int tid = threadIdx.x;
    double2 *z = reinterpret_cast<double2 *>(ch_smem);             // [N2] FFT work buffer; reduction scratch afterwards
    double *spec = reinterpret_cast<double *>(z + N2);             // [kBigFR][sstride]
    auto tw = [&](int t) {  // t < N2: quarter-circle symmetry as in the kernels above, from the table in L2
        const double2 w = g.twiddle[t & (NQ - 1)];
        return (t & NQ) ? make_double2(w.y, -w.x) : w;
    };
    const int sb = blockIdx.y;
    if (g.n_frames_b) {
        g.n_frames = g.n_frames_b[sb];
        g.n_samples = g.n_samples_b[sb];
        g.samples = reinterpret_cast<const ST *>(g.samples) + (long long)sb * g.sample_stride;
        const long long oo = (long long)sb * g.out_frames_stride * kCh;
        g.chroma_out = g.out_f64 ? (void *)(reinterpret_cast<double *>(g.chroma_out) + oo)
                                 : (void *)(reinterpret_cast<float *>(g.chroma_out) + oo);
    }
    constexpr int kMaxBf4 = 4;  // N2/4/256 butterflies per thread and stage
    for (int frame0 = blockIdx.x * kBigFR; frame0 < g.n_frames; frame0 += gridDim.x * kBigFR) {
        const int nf = (g.n_frames - frame0 < kBigFR) ? g.n_frames - frame0 : kBigFR;
        for (int f = 0; f < nf; f++) {
            const int frame = frame0 + f;
            // 1. window, pack as complex (out-of-range samples: the zero padding on the left, chroma.py:49)
            for (int n = tid; n < N2; n += kChromaNT) {
                const long long sidx = g.frame_offset + (long long)frame * g.hop + 2 * n;
                const double x0 = (sidx >= 0 && sidx < g.n_samples) ? (double)reinterpret_cast<const ST *>(g.samples)[sidx] : 0.0;
                const double x1 = (sidx + 1 >= 0 && sidx + 1 < g.n_samples) ? (double)reinterpret_cast<const ST *>(g.samples)[sidx + 1] : 0.0;
                z[n] = make_double2(x0 * g.window[2 * n], x1 * g.window[2 * n + 1]);
            }
            __syncthreads();
            // 2. Stockham radix-4 stages, in place via registers (log2(4096) is even: no radix-2 stage)
            const int q = N2 / 4;
            for (int p = 1; p < N2; p <<= 2) {
                double2 o[kMaxBf4][4];
                int jj[kMaxBf4];
                const int tstep = N2 / (2 * p);
#pragma unroll
                for (int r = 0; r < kMaxBf4; r++) {
                    const int i = tid + r * kChromaNT;
                    const int k = i & (p - 1);
                    const int t1 = k * tstep, t2 = 2 * t1, t3 = 3 * t1;
                    const double2 w1 = tw(t1), w2 = tw(t2);
                    double2 w3 = tw(t3 & (N2 - 1));
                    if (t3 >= N2) w3 = make_double2(-w3.x, -w3.y);
                    const double2 u0 = z[i];
                    const double2 u1 = cmul(w1, z[i + q]);
                    const double2 u2 = cmul(w2, z[i + 2 * q]);
                    const double2 u3 = cmul(w3, z[i + 3 * q]);
                    const double2 a0 = make_double2(u0.x + u2.x, u0.y + u2.y);
                    const double2 a1 = make_double2(u0.x - u2.x, u0.y - u2.y);
                    const double2 a2 = make_double2(u1.x + u3.x, u1.y + u3.y);
                    const double2 a3 = make_double2(u1.y - u3.y, -(u1.x - u3.x));
                    o[r][0] = make_double2(a0.x + a2.x, a0.y + a2.y);
                    o[r][1] = make_double2(a1.x + a3.x, a1.y + a3.y);
                    o[r][2] = make_double2(a0.x - a2.x, a0.y - a2.y);
                    o[r][3] = make_double2(a1.x - a3.x, a1.y - a3.y);
                    jj[r] = ((i - k) << 2) + k;
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < kMaxBf4; r++) {
                    z[jj[r]] = o[r][0];
                    z[jj[r] + p] = o[r][1];
                    z[jj[r] + 2 * p] = o[r][2];
                    z[jj[r] + 3 * p] = o[r][3];
                }
                __syncthreads();
            }
            // 3. untangle -> rfft bins, power spectrum
            const int nb = N2 + 1;
            double *sp = spec + (size_t)f * sstride;
            for (int k = tid; k < nb; k += kChromaNT) {
                const double2 a = z[k & (N2 - 1)];
                const double2 bq = z[(N2 - k) & (N2 - 1)];
                const double2 b = make_double2(bq.x, -bq.y);
                const double2 e = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y + b.y));
                const double2 dm = make_double2(a.x - b.x, a.y - b.y);
                const double2 od = make_double2(0.5 * dm.y, -0.5 * dm.x);
                const double2 w = (k < N2) ? tw(k) : make_double2(-1.0, 0.0);
                const double2 wo = cmul(w, od);
                const double2 x = make_double2(e.x + wo.x, e.y + wo.y);
                if (g.stft_out) g.stft_out[(size_t)frame * nb + k] = x;
                sp[k] = x.x * x.x + x.y * x.y;
            }
            __syncthreads();
        }
        if (g.chroma_out)
            project_normalize<kBigFR>(g, spec, sstride, reinterpret_cast<double *>(z), frame0, nf, 0, 1, tid, [] { __syncthreads(); });
        else
            __syncthreads();
    }
}

// ---- fft_len = 4096 (the reference's only setting, chroma.py:20 / wtw.py:27): a specialised kernel -----------------
// Same structure as above (two teams, one frame each; four frames per pass over the filterbank) with the FFT as four
// register passes -- radix 8, 8, 8, 4 over the 2048 packed complex points, one butterfly per thread and pass (two in the
// last) -- instead of eleven radix-2-equivalent levels in six LDS round trips: half the LDS traffic, 8 barriers instead
// of 12.  The work buffer is padded by one 16-byte slot per 8 so that every pass reads and writes without bank
// conflicts, samples are indexed with 32-bit arithmetic, and all compile-time strides fold into immediates.
namespace c4k {
constexpr int L = 4096, N2 = 2048, NQ = 1024;        // real length, packed complex length, quarter-circle table
constexpr int ZSLOTS = N2 + N2 / 8;                  // padded work buffer (double2 slots)
constexpr int SSTRIDE = N2 + 2;
__device__ __forceinline__ constexpr int zi(int n) { return n + (n >> 3); }

__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 mul_mi(double2 a) { return make_double2(a.y, -a.x); }  // * (-i)

// exp(-2 pi i t / 4096), 0 <= t < 4096, from the quarter table (t < 1024): every quarter turn multiplies by -i
__device__ __forceinline__ double2 twf(const double2 *twq, int t) {
    const double2 w = twq[t & (NQ - 1)];
    const int qd = (t >> 10) & 3;
    const double2 r1 = (qd & 1) ? make_double2(w.y, -w.x) : w;
    return (qd & 2) ? make_double2(-r1.x, -r1.y) : r1;
}

// 8-point DFT of u[0..7] in place (decimation in frequency, outputs in natural order in y)
__device__ __forceinline__ void dft8(const double2 (&u)[8], double2 (&y)[8]) {
    const double s = 0.70710678118654752440;
    const double2 a0 = cadd(u[0], u[4]), a1 = cadd(u[1], u[5]), a2 = cadd(u[2], u[6]), a3 = cadd(u[3], u[7]);
    const double2 b0 = csub(u[0], u[4]), d1 = csub(u[1], u[5]), d2 = csub(u[2], u[6]), d3 = csub(u[3], u[7]);
    const double2 b1 = make_double2(s * (d1.x + d1.y), s * (d1.y - d1.x));   // * W8
    const double2 b2 = mul_mi(d2);                                           // * W8^2 = -i
    const double2 b3 = make_double2(s * (d3.y - d3.x), -s * (d3.x + d3.y));  // * W8^3
    {
        const double2 t0 = cadd(a0, a2), t1 = cadd(a1, a3), t2 = csub(a0, a2), t3 = mul_mi(csub(a1, a3));
        y[0] = cadd(t0, t1);
        y[2] = cadd(t2, t3);
        y[4] = csub(t0, t1);
        y[6] = csub(t2, t3);
    }
    {
        const double2 t0 = cadd(b0, b2), t1 = cadd(b1, b3), t2 = csub(b0, b2), t3 = mul_mi(csub(b1, b3));
        y[1] = cadd(t0, t1);
        y[3] = cadd(t2, t3);
        y[5] = csub(t0, t1);
        y[7] = csub(t2, t3);
    }
}

// One radix-8 Stockham pass with sub-transform length P (1, 8 or 64): butterfly i = tid reads z[i + 256 m],
// multiplies input m by exp(-2 pi i k m / (8 P)), k = i mod P, and writes to (i - k) * 8 + k + m * P.
template <int P, class Bar>
__device__ __forceinline__ void pass8(double2 *z, const double2 *twq, int tid, Bar bar) {
    double2 u[8], y[8];
#pragma unroll
    for (int m = 0; m < 8; m++) u[m] = z[zi(tid + 256 * m)];
    const int k = tid & (P - 1);
    if (P > 1) {
        const int t1 = k * (L / (8 * P));  // < 512
#pragma unroll
        for (int m = 1; m < 8; m++) u[m] = cmul(twf(twq, t1 * m), u[m]);
    }
    dft8(u, y);
    bar();
    const int jj = ((tid - k) << 3) + k;
#pragma unroll
    for (int m = 0; m < 8; m++) z[zi(jj + m * P)] = y[m];
    bar();
}
}  // namespace c4k

template <typename ST>
__global__ void __launch_bounds__(kChromaWG) chroma_frames4096_kernel(ChromaArgs g) {
    using namespace c4k;
    extern __shared__ __align__(16) unsigned char ch_smem[];
    const int team = threadIdx.x >> 8, tid = threadIdx.x & (kChromaNT - 1);
    double2 *zbase = reinterpret_cast<double2 *>(ch_smem);       // [2][ZSLOTS]
    double2 *z = zbase + (size_t)team * ZSLOTS;
    double2 *twq = zbase + 2 * (size_t)ZSLOTS;                   // [NQ]
    double *spec = reinterpret_cast<double *>(twq + NQ);         // [kChromaFR][SSTRIDE]
    double *red = reinterpret_cast<double *>(z);                 // the team's reduction scratch (26 KB of its 36 KB)
    auto bar = [] { lds_barrier(); };

    for (int n = threadIdx.x; n < NQ; n += kChromaWG) twq[n] = g.twiddle[n];
    __syncthreads();

    const int sb = blockIdx.y;  // batched launch: this workgroup's stream
    if (g.n_frames_b) {
        g.n_frames = g.n_frames_b[sb];
        g.n_samples = g.n_samples_b[sb];
        g.samples = reinterpret_cast<const ST *>(g.samples) + (long long)sb * g.sample_stride;
        const long long oo = (long long)sb * g.out_frames_stride * kCh;
        g.chroma_out = g.out_f64 ? (void *)(reinterpret_cast<double *>(g.chroma_out) + oo)
                                 : (void *)(reinterpret_cast<float *>(g.chroma_out) + oo);
    }
    const ST *samples = reinterpret_cast<const ST *>(g.samples);
    const int n_samples = (int)g.n_samples;  // the host routes longer signals to the generic kernel
    const int off0 = (int)g.frame_offset, hop = g.hop;

    constexpr int kPairs = N2 / kChromaNT;  // 8 sample pairs per thread: n = tid + 256 r -> x[2n], x[2n+1]
    double win_re[kPairs], win_im[kPairs];
#pragma unroll
    for (int r = 0; r < kPairs; r++) {
        const int n = tid + r * kChromaNT;
        win_re[r] = g.window[2 * n];
        win_im[r] = g.window[2 * n + 1];
    }
    // the team's next frame, fetched one frame ahead and left untouched until it is used (a use would wait for the load);
    // indices clamped into the buffer, out-of-range positions zeroed at use
    ST ps0[kPairs], ps1[kPairs];
#pragma unroll
    for (int r = 0; r < kPairs; r++) ps0[r] = ps1[r] = (ST)0;
    const int hi = n_samples > 0 ? n_samples - 1 : 0;
    auto fetch_frame = [&](int frame) {
        const int s0 = (frame < g.n_frames) ? off0 + frame * hop : 0;
#pragma unroll
        for (int r = 0; r < kPairs; r++) {
            const int i0 = s0 + 2 * (tid + r * kChromaNT), i1 = i0 + 1;
            ps0[r] = samples[i0 < 0 ? 0 : (i0 > hi ? hi : i0)];
            ps1[r] = samples[i1 < 0 ? 0 : (i1 > hi ? hi : i1)];
        }
    };
    fetch_frame(blockIdx.x * kChromaFR + team);
#ifdef RTS_CHROMA_STAMPS
    long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long last_ = (long long)__builtin_amdgcn_s_memtime();
#endif
    for (int frame0 = blockIdx.x * kChromaFR; frame0 < g.n_frames; frame0 += gridDim.x * kChromaFR) {
        const int nf = (g.n_frames - frame0 < kChromaFR) ? g.n_frames - frame0 : kChromaFR;
        for (int round = 0; round < kChromaFR / 2; round++) {
            const int f = 2 * round + team;
            const int frame = frame0 + f;
            const bool live = f < nf;
            CH_STAMP(0);
            // 1. window, pack as complex
            const int s0 = off0 + frame * hop;
#pragma unroll
            for (int r = 0; r < kPairs; r++) {
                const int n = tid + r * kChromaNT;
                const int i0 = s0 + 2 * n;
                const double x0 = (i0 >= 0 && i0 < n_samples) ? (double)ps0[r] : 0.0;
                const double x1 = (i0 + 1 >= 0 && i0 + 1 < n_samples) ? (double)ps1[r] : 0.0;
                z[zi(n)] = make_double2(x0 * win_re[r], x1 * win_im[r]);
            }
            {
                const long long nxt = round + 1 < kChromaFR / 2 ? (long long)frame + 2
                                                                : (long long)frame0 + (long long)gridDim.x * kChromaFR + team;
                fetch_frame(nxt < g.n_frames ? (int)nxt : g.n_frames);
            }
            bar();
            CH_STAMP(1);
            // 2. 2048-point complex FFT: radix 8, 8, 8, then 4
            pass8<1>(z, twq, tid, bar);
            pass8<8>(z, twq, tid, bar);
            pass8<64>(z, twq, tid, bar);
            {
                constexpr int P = 512, Q = N2 / 4;
                double2 o[2][4];
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int i = tid + r * kChromaNT;  // k = i (P = 512 = N2/4)
                    const double2 u0 = z[zi(i)];
                    const double2 u1 = cmul(twf(twq, 2 * i), z[zi(i + Q)]);
                    const double2 u2 = cmul(twf(twq, 4 * i), z[zi(i + 2 * Q)]);
                    const double2 u3 = cmul(twf(twq, 6 * i), z[zi(i + 3 * Q)]);
                    const double2 a0 = cadd(u0, u2), a1 = csub(u0, u2), a2 = cadd(u1, u3), a3 = mul_mi(csub(u1, u3));
                    o[r][0] = cadd(a0, a2);
                    o[r][1] = cadd(a1, a3);
                    o[r][2] = csub(a0, a2);
                    o[r][3] = csub(a1, a3);
                }
                bar();
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int i = tid + r * kChromaNT;
#pragma unroll
                    for (int m = 0; m < 4; m++) z[zi(i + m * P)] = o[r][m];
                }
                bar();
            }
            CH_STAMP(2);
            // 3. untangle: X[k] = E[k] + W_L^k O[k], E = (Z[k] + conj Z[N2-k]) / 2, O = (Z[k] - conj Z[N2-k]) / (2i)
            constexpr int nb = N2 + 1;
            double *sp = spec + (size_t)f * SSTRIDE;
            if (live) {
                // bins k and N2 - k share their two inputs: X[N2 - k] = conj(E[k] - W_L^k O[k])
                for (int k = tid; k <= N2 / 2; k += kChromaNT) {
                    const double2 a = z[zi(k)];
                    const double2 bq = z[zi((N2 - k) & (N2 - 1))];  // Z[N2] == Z[0]
                    const double2 b = make_double2(bq.x, -bq.y);     // conj
                    const double2 e = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y + b.y));
                    const double2 dm = csub(a, b);
                    const double2 o = make_double2(0.5 * dm.y, -0.5 * dm.x);  // dm / (2i)
                    const double2 wo = cmul(twf(twq, k), o);
                    const double2 x = cadd(e, wo);
                    const double2 d = csub(e, wo);
                    const double2 x2 = make_double2(d.x, -d.y);
                    sp[k] = x.x * x.x + x.y * x.y;
                    if (g.stft_out) g.stft_out[(size_t)frame * nb + k] = x;
                    if (k != N2 - k) {
                        sp[N2 - k] = x2.x * x2.x + x2.y * x2.y;
                        if (g.stft_out) g.stft_out[(size_t)frame * nb + (N2 - k)] = x2;
                    }
                }
            }
            bar();
            CH_STAMP(3);
        }
        if (g.chroma_out) project_normalize<kChromaFR / 2>(g, spec, SSTRIDE, red, frame0, nf, team, 2, tid, bar);
        CH_STAMP(4);
    }
#ifdef RTS_CHROMA_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int i = 0; i < 6; i++) g_chroma_stamps[i] = st_[i];
#endif
}

__global__ void __launch_bounds__(kChromaNT) chroma_project_kernel(ChromaArgs g) {
    extern __shared__ __align__(16) unsigned char ch_smem[];
    const int nb = g.L / 2 + 1;
    const int sstride = nb + 1;
    double *spec = reinterpret_cast<double *>(ch_smem);   // [kChromaFR][sstride]
    double *red = spec + (size_t)kChromaFR * sstride;      // [12][256] + [192]
    const int tid = threadIdx.x;
    for (int frame0 = blockIdx.x * kChromaFR; frame0 < g.n_frames; frame0 += gridDim.x * kChromaFR) {
        const int nf = (g.n_frames - frame0 < kChromaFR) ? g.n_frames - frame0 : kChromaFR;
        for (int f = 0; f < nf; f++)
            for (int k = tid; k < nb; k += kChromaNT) spec[(size_t)f * sstride + k] = g.spec_in[(size_t)(frame0 + f) * nb + k];
        __syncthreads();
        project_normalize<kChromaFR>(g, spec, sstride, red, frame0, nf, 0, 1, tid, [] { __syncthreads(); });
    }
}

// out[m][f] = max(in[m+1][f] - in[m][f], 0), m < n_frames - 1   (np.clip(np.diff(chroma), 0, inf))
__global__ void chroma_diff_kernel(const void *in, void *out, long long n_out, int f64) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    if (f64) {
        const double *x = reinterpret_cast<const double *>(in);
        const double d = x[i + kCh] - x[i];
        reinterpret_cast<double *>(out)[i] = d > 0.0 ? d : 0.0;
    } else {
        const float *x = reinterpret_cast<const float *>(in);
        const float d = x[i + kCh] - x[i];
        reinterpret_cast<float *>(out)[i] = d > 0.0f ? d : 0.0f;
    }
}

}  // namespace rts

struct rts_chroma {
    int L, hop, logL2;
    double *window;    // device [L]
    double2 *twiddle;  // device [L/2]
    double *fb;        // device [12][L/2+1]
    double *fbt;       // device [L/2+1][12]
    size_t smem_frames, smem_frames4096, smem_frames_big, smem_project;
    int device;  // the HIP device the plan's tables live on
    int cus;     // its compute units
};

static int chroma_check_device(const rts_chroma *h) {
    int d = -1;
    RTS_HIP(hipGetDevice(&d));
    if (d != h->device)
        return rts::set_error(RTS_ERR_INVALID, "plan was created on device %d but device %d is current "
                                               "(one process per GPU, or hipSetDevice before the call)", h->device, d);
    return RTS_OK;
}

static hipError_t upload_transposed(double *dst_dev, const double *fb_host, int nb) {
    double *t = (double *)malloc(sizeof(double) * rts::kCh * nb);
    if (!t) return hipErrorOutOfMemory;
    for (int k = 0; k < nb; k++)
        for (int p = 0; p < rts::kCh; p++) t[(size_t)k * rts::kCh + p] = fb_host[(size_t)p * nb + k];
    const hipError_t e = hipMemcpy(dst_dev, t, sizeof(double) * rts::kCh * nb, hipMemcpyHostToDevice);
    free(t);
    return e;
}

extern "C" {

long long rts_chroma_num_frames(long long n_samples, int fft_len, int hop, int pad_left) {
    const long long n = n_samples + pad_left;  // chroma.py:49-54
    if (fft_len < 2 || hop < 1 || n < fft_len) return 0;
    return (n - fft_len) / hop + 1;
}

int rts_chroma_create(int fft_len, int hop, const double *window_host, const double *fb_host, rts_chroma **out) {
    using namespace rts;
    if (!out) return set_error(RTS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!fb_host) return set_error(RTS_ERR_INVALID, "fb_host is NULL (12 x (fft_len/2+1) filterbank)");
    if (fft_len < 64 || fft_len > 8192 || (fft_len & (fft_len - 1)))
        return set_error(RTS_ERR_UNSUPPORTED, "fft_len must be a power of two in [64, 8192] (got %d): a frame's FFT runs in LDS", fft_len);
    if (hop < 1) return set_error(RTS_ERR_INVALID, "hop must be >= 1");
    rts_chroma *h = (rts_chroma *)calloc(1, sizeof(rts_chroma));
    if (!h) return set_error(RTS_ERR_INVALID, "out of host memory");
    h->L = fft_len;
    h->hop = hop;
    if (hipGetDevice(&h->device) != hipSuccess ||
        hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || h->cus < 1) {
        free(h);
        return set_error(RTS_ERR_HIP, "hipGetDevice failed");
    }
    const int L = fft_len, N2 = L / 2, nb = N2 + 1;
    double *win = (double *)malloc(sizeof(double) * L);
    double2 *tw = (double2 *)malloc(sizeof(double2) * N2);
    if (window_host) {
        memcpy(win, window_host, sizeof(double) * L);
    } else {  // np.hanning(L): 0.5 + 0.5 cos(pi n / (L-1)), n = 1-L, 3-L, ..., L-1
        for (int i = 0; i < L; i++) win[i] = 0.5 + 0.5 * cos(M_PI * (double)(2 * i + 1 - L) / (double)(L - 1));
    }
    for (int n = 0; n < N2; n++) {
        const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)n / (long double)L;
        tw[n].x = (double)cosl(ang);
        tw[n].y = (double)sinl(ang);
    }
    hipError_t e;
    if ((e = hipMalloc((void **)&h->window, sizeof(double) * L)) != hipSuccess ||
        (e = hipMalloc((void **)&h->twiddle, sizeof(double2) * N2)) != hipSuccess ||
        (e = hipMalloc((void **)&h->fb, sizeof(double) * kCh * nb)) != hipSuccess ||
        (e = hipMemcpy(h->window, win, sizeof(double) * L, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(h->twiddle, tw, sizeof(double2) * N2, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(h->fb, fb_host, sizeof(double) * kCh * nb, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMalloc((void **)&h->fbt, sizeof(double) * kCh * nb)) != hipSuccess ||
        (e = upload_transposed(h->fbt, fb_host, nb)) != hipSuccess) {
        free(win);
        free(tw);
        rts_chroma_destroy(h);
        return set_error(RTS_ERR_HIP, "chroma plan upload failed: %s", hipGetErrorString(e));
    }
    free(win);
    free(tw);
    h->smem_frames = sizeof(double2) * (2 * (size_t)N2 + N2 / 2) + sizeof(double) * (size_t)kChromaFR * (N2 + 2) +
                     ((2 * N2 >= 3264) ? 0 : sizeof(double) * 2 * 3264) + 64;
    h->smem_project = sizeof(double) * ((size_t)kChromaFR * (nb + 1) + 3264) + 64;
    h->smem_frames_big = sizeof(double2) * (size_t)N2 + sizeof(double) * (size_t)kBigFR * (N2 + 2) + 64;
    // per plan, i.e. on the device that is current now (the attribute is per device)
    h->smem_frames4096 = sizeof(double2) * (2 * (size_t)c4k::ZSLOTS + c4k::NQ) + sizeof(double) * (size_t)kChromaFR * c4k::SSTRIDE + 64;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&chroma_frames4096_kernel<float>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&chroma_frames4096_kernel<double>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&chroma_frames_kernel<float>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&chroma_frames_kernel<double>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&chroma_frames_big_kernel<float>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&chroma_frames_big_kernel<double>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&chroma_project_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
        rts_chroma_destroy(h);
        return set_error(RTS_ERR_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return RTS_OK;
}

int rts_chroma_plan_info(const rts_chroma *h, int *fft_len, int *hop) {
    if (!h) return rts::set_error(RTS_ERR_INVALID, "handle is NULL");
    if (fft_len) *fft_len = h->L;
    if (hop) *hop = h->hop;
    return RTS_OK;
}

int rts_chroma_destroy(rts_chroma *h) {
    if (!h) return RTS_OK;
    if (h->window) (void)hipFree(h->window);
    if (h->twiddle) (void)hipFree(h->twiddle);
    if (h->fb) (void)hipFree(h->fb);
    if (h->fbt) (void)hipFree(h->fbt);
    free(h);
    return RTS_OK;
}

int rts_chroma_frames(rts_chroma *h, const void *samples_dev, int sample_dtype, long long n_samples,
                      int pad_left, int n_frames, int normalize, void *chroma_out_dev, int out_dtype,
                      double *stft_out_dev, void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (!samples_dev) return set_error(RTS_ERR_INVALID, "samples_dev is NULL");
    if (!chroma_out_dev && !stft_out_dev) return set_error(RTS_ERR_INVALID, "no output requested");
    if ((sample_dtype != RTS_F32 && sample_dtype != RTS_F64) || (out_dtype != RTS_F32 && out_dtype != RTS_F64))
        return set_error(RTS_ERR_INVALID, "bad dtype");
    if (n_frames < 0 || pad_left < 0 || n_samples < 0) return set_error(RTS_ERR_INVALID, "negative size");
    if ((long long)n_frames > rts_chroma_num_frames(n_samples, h->L, h->hop, pad_left))
        return set_error(RTS_ERR_INVALID, "n_frames=%d exceeds the %lld full frames in %lld samples", n_frames,
                         rts_chroma_num_frames(n_samples, h->L, h->hop, pad_left), n_samples);
    if (n_frames == 0) return RTS_OK;
    if (int rc = chroma_check_device(h); rc != RTS_OK) return rc;
    ChromaArgs g;
    memset(&g, 0, sizeof(g));
    g.samples = samples_dev;
    g.window = h->window;
    g.twiddle = h->twiddle;
    g.fb = h->fb;
    g.fbt = h->fbt;
    g.chroma_out = chroma_out_dev;
    g.stft_out = reinterpret_cast<double2 *>(stft_out_dev);
    g.n_samples = n_samples;
    g.frame_offset = -(long long)pad_left;
    g.L = h->L;
    g.hop = h->hop;
    g.n_frames = n_frames;
    g.normalize = normalize;
    g.samples_f64 = sample_dtype == RTS_F64;
    g.out_f64 = out_dtype == RTS_F64;
    const int groups = (n_frames + kChromaFR - 1) / kChromaFR;
    // one workgroup per CU (155 KB of LDS each) walking its share of the frame groups: 41.7 M frames/s against 40.7 M
    // with four short-lived workgroups per CU (each one loads the twiddle table and its window registers first)
    const int grid = groups < h->cus ? groups : h->cus;
    // fft_len 4096 with 32-bit sample indices: the specialised kernel; anything else: the generic one
    const bool fast = (h->L == 4096) && (n_samples + 2LL * h->L + (long long)n_frames * h->hop < 0x7fffffffLL);
    if (fast) {
        if (g.samples_f64)
            hipLaunchKernelGGL(chroma_frames4096_kernel<double>, dim3(grid), dim3(kChromaWG), h->smem_frames4096, (hipStream_t)stream, g);
        else
            hipLaunchKernelGGL(chroma_frames4096_kernel<float>, dim3(grid), dim3(kChromaWG), h->smem_frames4096, (hipStream_t)stream, g);
    } else if (h->L > 4096) {
        const int groups_big = (n_frames + kBigFR - 1) / kBigFR;
        const int grid_big = groups_big < 4 * h->cus ? groups_big : 4 * h->cus;
        if (g.samples_f64)
            hipLaunchKernelGGL(chroma_frames_big_kernel<double>, dim3(grid_big), dim3(kChromaNT), h->smem_frames_big, (hipStream_t)stream, g);
        else
            hipLaunchKernelGGL(chroma_frames_big_kernel<float>, dim3(grid_big), dim3(kChromaNT), h->smem_frames_big, (hipStream_t)stream, g);
    } else if (g.samples_f64) {
        hipLaunchKernelGGL(chroma_frames_kernel<double>, dim3(grid), dim3(kChromaWG), h->smem_frames, (hipStream_t)stream, g);
    } else {
        hipLaunchKernelGGL(chroma_frames_kernel<float>, dim3(grid), dim3(kChromaWG), h->smem_frames, (hipStream_t)stream, g);
    }
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

int rts_chroma_frames_batch(rts_chroma *h, const void *samples_dev, int sample_dtype, long long sample_stride,
                            const int32_t *n_samples_dev, int pad_left, int B, int n_frames_max,
                            const int32_t *n_frames_dev, int normalize, void *chroma_out_dev, int out_dtype,
                            void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (!samples_dev || !n_samples_dev || !n_frames_dev || !chroma_out_dev)
        return set_error(RTS_ERR_INVALID, "NULL device buffer");
    if ((sample_dtype != RTS_F32 && sample_dtype != RTS_F64) || (out_dtype != RTS_F32 && out_dtype != RTS_F64))
        return set_error(RTS_ERR_INVALID, "bad dtype");
    if (B < 1 || n_frames_max < 0 || pad_left < 0 || sample_stride < 0) return set_error(RTS_ERR_INVALID, "bad size");
    if (n_frames_max == 0) return RTS_OK;
    if (int rc = chroma_check_device(h); rc != RTS_OK) return rc;
    ChromaArgs g;
    memset(&g, 0, sizeof(g));
    g.samples = samples_dev;
    g.window = h->window;
    g.twiddle = h->twiddle;
    g.fb = h->fb;
    g.fbt = h->fbt;
    g.chroma_out = chroma_out_dev;
    g.frame_offset = -(long long)pad_left;
    g.L = h->L;
    g.hop = h->hop;
    g.normalize = normalize;
    g.samples_f64 = sample_dtype == RTS_F64;
    g.out_f64 = out_dtype == RTS_F64;
    g.n_samples_b = n_samples_dev;
    g.n_frames_b = n_frames_dev;
    g.sample_stride = sample_stride;
    g.out_frames_stride = n_frames_max;
    const int groups_b = (n_frames_max + kChromaFR - 1) / kChromaFR;
    const int gx = groups_b < 64 ? groups_b : 64;
    const bool fast = (h->L == 4096) && (sample_stride + 2LL * h->L + (long long)n_frames_max * h->hop < 0x7fffffffLL);
    if (fast) {
        if (g.samples_f64)
            hipLaunchKernelGGL(chroma_frames4096_kernel<double>, dim3(gx, B), dim3(kChromaWG), h->smem_frames4096, (hipStream_t)stream, g);
        else
            hipLaunchKernelGGL(chroma_frames4096_kernel<float>, dim3(gx, B), dim3(kChromaWG), h->smem_frames4096, (hipStream_t)stream, g);
    } else if (h->L > 4096) {
        const int groups_big = (n_frames_max + kBigFR - 1) / kBigFR;
        const int gx_big = groups_big < 64 ? groups_big : 64;
        if (g.samples_f64)
            hipLaunchKernelGGL(chroma_frames_big_kernel<double>, dim3(gx_big, B), dim3(kChromaNT), h->smem_frames_big, (hipStream_t)stream, g);
        else
            hipLaunchKernelGGL(chroma_frames_big_kernel<float>, dim3(gx_big, B), dim3(kChromaNT), h->smem_frames_big, (hipStream_t)stream, g);
    } else if (g.samples_f64) {
        hipLaunchKernelGGL(chroma_frames_kernel<double>, dim3(gx, B), dim3(kChromaWG), h->smem_frames, (hipStream_t)stream, g);
    } else {
        hipLaunchKernelGGL(chroma_frames_kernel<float>, dim3(gx, B), dim3(kChromaWG), h->smem_frames, (hipStream_t)stream, g);
    }
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

int rts_chroma_project(rts_chroma *h, const double *spec_dev, int n_frames, int normalize, void *chroma_out_dev,
                       int out_dtype, void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (!spec_dev || !chroma_out_dev) return set_error(RTS_ERR_INVALID, "NULL device buffer");
    if (out_dtype != RTS_F32 && out_dtype != RTS_F64) return set_error(RTS_ERR_INVALID, "bad dtype");
    if (n_frames < 0) return set_error(RTS_ERR_INVALID, "negative size");
    if (n_frames == 0) return RTS_OK;
    ChromaArgs g;
    memset(&g, 0, sizeof(g));
    g.fb = h->fb;
    g.fbt = h->fbt;
    g.spec_in = spec_dev;
    g.chroma_out = chroma_out_dev;
    g.L = h->L;
    g.n_frames = n_frames;
    g.normalize = normalize;
    g.out_f64 = out_dtype == RTS_F64;
    const int groups_p = (n_frames + kChromaFR - 1) / kChromaFR;
    const int grid = groups_p < 1024 ? groups_p : 1024;
    hipLaunchKernelGGL(chroma_project_kernel, dim3(grid), dim3(kChromaNT), h->smem_project, (hipStream_t)stream, g);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

#ifdef RTS_CHROMA_STAMPS
int rts_chroma_read_stamps(long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(rts::g_chroma_stamps), sizeof(long long) * 8) == hipSuccess ? 0 : -3;
}
#endif

int rts_chroma_diff(const void *chroma_dev, int dtype, int n_frames, void *out_dev, void *stream) {
    using namespace rts;
    if (!chroma_dev || !out_dev) return set_error(RTS_ERR_INVALID, "NULL device buffer");
    if (dtype != RTS_F32 && dtype != RTS_F64) return set_error(RTS_ERR_INVALID, "bad dtype");
    if (n_frames < 2) return RTS_OK;
    const long long n_out = (long long)(n_frames - 1) * kCh;
    hipLaunchKernelGGL(chroma_diff_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       chroma_dev, out_dev, n_out, dtype == RTS_F64);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

}  // extern "C"
