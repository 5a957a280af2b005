/*
 * rtsync.h -- C-ABI of librtsync.so: the MI355X (gfx950) implementation of the chroma + DTW /
 * online-time-warping / windowed-time-warping hot path of smritip/real-time-audio-sync.
 *
 * The reference has no FFI boundary of its own (it is plain in-process Python); each entry point
 * below names the reference interface it replaces (file:line under /root/reference).  The Python
 * classes in real_time_audio_sync_amd/ bind these symbols with ctypes and keep the reference's
 * call surface (INTEGRATION.md shows the binding a maintainer would add).
 *
 * Conventions
 *   - All `*_dev` pointers are device (HIP) pointers on the current device; the library never
 *     frees or reallocates caller memory.  `stream` is a hipStream_t passed as void* (NULL =
 *     the default stream).  Calls taking a stream are asynchronous on it; `*_read_*` getters
 *     synchronise that stream and copy to host memory.
 *   - Feature matrices are FRAME-MAJOR: [frame][feature], feature stride 1 (the reference's
 *     numpy arrays are feature-major (12, N); the Python layer transposes once at upload).
 *   - Return value: 0 = ok, < 0 = error (message via rts_last_error(), thread-local).
 *   - One handle must not be driven from two host threads at once.
 *   - There is no CPU fallback anywhere behind this header.
 */
#ifndef RTSYNC_H
#define RTSYNC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTS_OK 0
#define RTS_ERR_INVALID (-1)     /* bad argument (message says which) */
#define RTS_ERR_UNSUPPORTED (-2) /* valid in the reference, not supported by this build (e.g. c too large) */
#define RTS_ERR_HIP (-3)         /* a HIP runtime call failed */
#define RTS_ERR_NO_DEVICE (-4)   /* no gfx950 device visible */

/* dtype of feature / sample buffers handed to the library */
#define RTS_F32 0
#define RTS_F64 1
#define RTS_I16 2 /* PCM16 samples, rts_live_* only: value / 32768 in float32, exactly what librosa.load returns for a WAV */

/* which reference class an OTW handle follows */
#define RTS_VARIANT_OTW 0         /* otw_eran.py:5   OnlineTimeWarping (sentinel 1e10, run_count starts 1) */
#define RTS_VARIANT_LIVENOTE 1    /* livenote.py:3   LiveNote          (sentinel inf,  run_count starts 0) */
#define RTS_VARIANT_LIVENOTE_V2 2 /* livenote_v2.py:3 LiveNoteV2       (+ forward-only path filter)       */

#define RTS_COST_DOT 0    /* 1 - <live, ref>            otw_eran.py:220, livenote_v2.py:170 */
#define RTS_COST_EUCLID 1 /* ||live - ref||_2           livenote_v2.py:168 (chroma_diff=True) */

/* direction / previous codes in the state vector */
#define RTS_DIR_NONE (-1)
#define RTS_DIR_BOTH 0
#define RTS_DIR_ROW 1
#define RTS_DIR_COLUMN 2

/* per-stream status (replaces insert()'s return value) */
#define RTS_RUNNING 0
#define RTS_STOP_REF_END 1  /* insert() returned "stop": otw_eran.py:69-71, livenote_v2.py:80-82 */
#define RTS_LIVE_OVERFLOW 2 /* "ran out of room in pre-allocated live-sequence": otw_eran.py:53-55 */
#define RTS_DEVICE_FAULT 3  /* an in-launch hand-off between workgroups ran into its bound (never expected) */

/* how rts_otw_run walks the live sequence */
#define RTS_MODE_INSERT_LOOP 0 /* for i: insert(live[:, i])  (tests.py:160-163, test_simple.py:122-125) */
#define RTS_MODE_SET_LIVE 1    /* set_live(live)             (otw_eran.py:91-142, livenote_v2.py:108-155) */

/* layout of the int32 state vector returned by rts_otw_read_state (RTS_STATE_LEN entries) */
#define RTS_STATE_LEN 16
#define RTS_ST_T 0            /* t / live_ptr */
#define RTS_ST_J 1            /* j / ref_ptr */
#define RTS_ST_DIRECTION 2
#define RTS_ST_PREVIOUS 3
#define RTS_ST_RUN_COUNT 4
#define RTS_ST_STATUS 5
#define RTS_ST_FIRST_INSERT 6
#define RTS_ST_N_PATH 7       /* path points recorded */
#define RTS_ST_CONSUMED 8     /* live frames consumed (inserts made, incl. the one that stopped) */
#define RTS_ST_ROW_STRIPS 9
#define RTS_ST_COL_STRIPS 10
#define RTS_ST_CELLS_LO 11    /* cells evaluated, low/high 32 bits */
#define RTS_ST_CELLS_HI 12
#define RTS_ST_PATH_TRUNCATED 13
#define RTS_ST_BAND_RECOMPUTES 15 /* times a band minimum left the window and was recomputed (diagnostic) */

const char *rts_last_error(void);
int rts_version(void);
/* Number of visible HIP devices that are gfx950; < 0 on HIP failure. */
int rts_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Online time warping, batched over B independent live streams against one reference.
 * ------------------------------------------------------------------------------------------ */
typedef struct rts_otw rts_otw;

/* Replaces OnlineTimeWarping.__init__ (otw_eran.py:6-36) / LiveNote.__init__ (livenote.py:5-35) /
 * LiveNoteV2.__init__ (livenote_v2.py:8-40) for B streams at once.  `ref_dev` ([N][F], dtype
 * `ref_dtype`) is held by reference like otw_eran.py:17 and must outlive the handle.  Instead of the
 * reference's dense (2N x N) cost/acc matrices the handle keeps two (c+1)-cell bands per stream.
 * F must be 12.  Supported band widths: 1 <= c <= 2036 (LDS windows of 512 cells up to c = 500, 1024 up to 1012, 2048
 * above).  Above c = 500 no frame ring fits in LDS beside the bands: the default pipelined kernel and the dense mirror
 * (rts_otw_set_dense / rts_otw_replay_dense: the plain 8-wave kernel) read every frame from global memory, live buffers
 * must be 16-byte aligned, and RTS_OTW_SPEC=0 (the plain kernel without the dense mirror: an A/B knob) returns
 * RTS_ERR_UNSUPPORTED with a message. */
int rts_otw_create(const void *ref_dev, int ref_dtype, int F, int N, int B, int c, int max_run_count,
                   int variant, int cost_kind, rts_otw **out);
int rts_otw_destroy(rts_otw *h);
/* Back to the freshly-constructed state (all streams). */
int rts_otw_reset(rts_otw *h, void *stream);

/* Whole live sequences, one launch.  `live_dev`: [B][T_max][F] (dtype `live_dtype`, frame-major);
 * `live_len_dev`: int32[B] valid frames per stream (<= T_max).  Resets the handle, then behaves like
 * the harness loop `for i in range(T): if ln.insert(live[:, i]) == "stop": break`
 * (RTS_MODE_INSERT_LOOP) or like `ln.set_live(live)` (RTS_MODE_SET_LIVE).  The live buffer is read
 * in place and must stay valid until the stream has finished. */
int rts_otw_run(rts_otw *h, const void *live_dev, int live_dtype, int T_max, const int32_t *live_len_dev,
                int mode, void *stream);

/* One new frame per stream: replaces insert(live_sample) (otw_eran.py:38-85, livenote_v2.py:43-104).
 * `frames_dev`: [B][F].  `active_dev`: optional uint8[B]; streams with 0 receive no frame (NULL = all).
 * Frames are appended to a handle-owned history of 2N frames per stream (the reference's
 * pre-allocated self.live, otw_eran.py:14,20). */
int rts_otw_insert(rts_otw *h, const void *frames_dev, int frames_dtype, const uint8_t *active_dev,
                   void *stream);

/* Several new frames per stream in one call: the loop `while len(data) >= 4096: ... ln.insert(col);
 * data = data[2048:]` of livenote_live.py:185-208 when a microphone buffer yields more than one hop.
 * `frames_dev`: [B][n_max][F]; `n_new_dev`: int32[B] frames to take per stream (NULL = n_max for all). */
int rts_otw_push(rts_otw *h, const void *frames_dev, int frames_dtype, int n_max, const int32_t *n_new_dev,
                 void *stream);

/* Getters (synchronise `stream`). */
int rts_otw_read_state(rts_otw *h, int b, int32_t *state /* RTS_STATE_LEN */, void *stream);
int rts_otw_read_states(rts_otw *h, int32_t *states /* [B][RTS_STATE_LEN] */, void *stream);
/* .path: (live_idx, ref_idx) int32 pairs in recording order.  `cap_pairs` = capacity of `pairs`;
 * *n receives the full length (copy is truncated to cap_pairs). */
int rts_otw_read_path(rts_otw *h, int b, int32_t *pairs, int cap_pairs, int *n, void *stream);
/* The live part of .acc_cost: row t over columns [j-c, j] and column j over rows [t-c, t]
 * (c+1 doubles each, index i <-> offset i-c; NaN where the index is negative). */
int rts_otw_read_bands(rts_otw *h, int b, double *row_band, double *col_band, void *stream);
/* Device-side views for zero-copy consumers (torch): path buffer [B][path_cap][2] int32 and state
 * [B][RTS_STATE_LEN] int32. */
int rts_otw_device_views(rts_otw *h, int32_t **path_dev, int *path_cap, int32_t **state_dev);
/* Optional mirror of the reference's dense matrices (otw_eran.py:23,27; plotted by livenote_v2.ipynb):
 * acc_dev / cost_dev are caller-owned double [B][2N][N] buffers that every evaluated cell is also
 * written to (never-evaluated cells hold the sentinel 1e10 / +inf, resp. -1).  NULL, NULL switches it
 * off.  Resets the handle (the matrices are re-initialised on every reset / run). */
int rts_otw_set_dense(rts_otw *h, double *acc_dev, double *cost_dev, void *stream);
/* The same matrices on demand, without slowing the tracker down: recomputes them, from everything the handle has
 * consumed since its last reset, into caller-owned double [B][2N][N] buffers.  Frames that came through rts_otw_insert /
 * rts_otw_push are kept by the handle: pass live_dev = NULL (live_dtype, T_max, live_len_dev ignored).  Frames of an
 * rts_otw_run are the caller's memory and the library keeps no pointer to them: pass that call's live_dev, live_dtype,
 * T_max and live_len_dev again (dtype and T_max are checked against the run being replayed).  The handle's own state is
 * not touched, so the streams keep running on the pipelined kernel.  Every supported band width.  Synchronises `stream`.
 * What the drop-in classes' .acc_cost / .cost
 * (otw_eran.py:23,27) are made of. */
int rts_otw_replay_dense(rts_otw *h, const void *live_dev, int live_dtype, int T_max, const int32_t *live_len_dev,
                         double *acc_dev, double *cost_dev, void *stream);
/* Tuning knob, not semantics: waves per stream workgroup (1, 2, 4 or 8; default 8).  Results are identical.
 * With 8 waves and no dense mirror the library runs its pipelined kernel (the next step's strips are computed
 * beside this step's control work); the environment variable RTS_OTW_SPEC=0, read by rts_otw_create, selects
 * the plain kernel instead (A/B measurements, tests). */
int rts_otw_set_waves(rts_otw *h, int waves);
/* Average device time of the last kernel launches is measured by the caller with HIP events on
 * `stream`; this returns the kernel's name as it appears in rocprofv3 traces. */
const char *rts_otw_kernel_name(const rts_otw *h);

/* ------------------------------------------------------------------------------------------
 * Offline DTW, batched over B independent (a, b) pairs.
 * ------------------------------------------------------------------------------------------ */

/* Bytes of the device workspace rts_dtw needs for (M, N, B): packed step codes (2 bits per cell), the rows
 * handed between the workgroups of one pair's pipeline, and a status word. */
int rts_dtw_workspace_bytes(int M, int N, int B, size_t *bytes);

/* Replaces dtw.DTW(seq_a, seq_b) -> (cost, acc_cost, path) (dtw.py:5-53).
 *   a_dev: [B][M][F] frames of seq_a (rows of the matrices), `a_stride` = frames between
 *          consecutive pairs (0 = every pair shares one a); b_dev / b_stride likewise, [B][N][F].
 *   cost_dev, acc_dev: double [B][M][N] outputs (dtw.py:11, :14); back_dev: optional (may be NULL) int8
 *          [B][M][N], the reference's internal `back` matrix: step codes 0 = (0,-1), 1 = (-1,0),
 *          2 = (-1,-1) (dtw.py:30); path_dev: int32 [B][M+N][2], pairs (i, j) from (0,0) to (M-1,N-1);
 *          path_len_dev: int32 [B] (-1 if the device pipeline reported a fault).
 *   ws_dev / ws_bytes: caller-owned, 16-byte aligned scratch of at least rts_dtw_workspace_bytes(M, N, B).
 * Any M, N >= 1 (a long pair is spread over many workgroups).  B <= 65535.  Asynchronous on `stream`;
 * no allocation, no synchronisation (graph-capturable). */
int rts_dtw(const void *a_dev, int a_dtype, long long a_stride, const void *b_dev, int b_dtype,
            long long b_stride, int F, int M, int N, int B, double *cost_dev, double *acc_dev,
            int8_t *back_dev, int32_t *path_dev, int32_t *path_len_dev, void *ws_dev, size_t ws_bytes,
            void *stream);

/* ------------------------------------------------------------------------------------------
 * Chroma front end: frame -> window -> rFFT -> power -> 12-bin filterbank -> L2 normalise.
 * ------------------------------------------------------------------------------------------ */
typedef struct rts_chroma rts_chroma;

/* num_hops of chroma.create_stft (chroma.py:49-54): floor((n_samples + pad_left - fft_len) / hop) + 1,
 * 0 if the (padded) signal is shorter than one frame. */
long long rts_chroma_num_frames(long long n_samples, int fft_len, int hop, int pad_left);

/* A plan = window + twiddles + filterbank on the device.  `fb_host`: 12 x (fft_len/2+1) doubles, row
 * = pitch class (what librosa.filters.chroma(fs, fft_len) returns at chroma.py:69 / wtw.py:39);
 * `window_host`: fft_len doubles or NULL for np.hanning(fft_len) (chroma.py:39,:62).  fft_len: power
 * of two in [64, 8192] (the reference uses 4096, chroma.py:20). */
int rts_chroma_create(int fft_len, int hop, const double *window_host, const double *fb_host, rts_chroma **out);
int rts_chroma_destroy(rts_chroma *h);

/* Replaces create_stft + create_chroma (chroma.py:44-75), wav_to_chroma_col (chroma.py:35-42) and the
 * per-hop chroma of WTW.insert (wtw.py:81-90).  Frame m covers samples [m*hop - pad_left, +fft_len),
 * indices < 0 read as zero (pad_left = fft_len/2 for create_stft's centred framing, 0 for live
 * buffers).  `chroma_out_dev`: [n_frames][12] (out_dtype) or NULL; `stft_out_dev`: optional
 * [n_frames][fft_len/2+1] complex doubles (re, im) -- create_stft's return value, frame-major.
 * normalize = 0 gives create_chroma(ft, normalize=False). */
int rts_chroma_frames(rts_chroma *h, const void *samples_dev, int sample_dtype, long long n_samples,
                      int pad_left, int n_frames, int normalize, void *chroma_out_dev, int out_dtype,
                      double *stft_out_dev, void *stream);

/* The same for B independent sample buffers in one launch (many live microphones): samples_dev is
 * [B][sample_stride]; stream b holds n_samples_dev[b] valid samples and gets n_frames_dev[b]
 * (<= n_frames_max) frames written to chroma_out_dev [B][n_frames_max][12]. */
int rts_chroma_frames_batch(rts_chroma *h, const void *samples_dev, int sample_dtype, long long sample_stride,
                            const int32_t *n_samples_dev, int pad_left, int B, int n_frames_max,
                            const int32_t *n_frames_dev, int normalize, void *chroma_out_dev, int out_dtype,
                            void *stream);

/* fft_len and hop of a plan (either pointer may be NULL). */
int rts_chroma_plan_info(const rts_chroma *h, int *fft_len, int *hop);

/* create_chroma(ft) for a power spectrum that is already on the device: spec_dev [n_frames][fft_len/2+1]. */
int rts_chroma_project(rts_chroma *h, const double *spec_dev, int n_frames, int normalize, void *chroma_out_dev,
                       int out_dtype, void *stream);

/* wav_to_chroma_diff's last step (chroma.py:85-90): out[m][f] = max(chroma[m+1][f] - chroma[m][f], 0),
 * out has n_frames-1 frames. */
int rts_chroma_diff(const void *chroma_dev, int dtype, int n_frames, void *out_dev, void *stream);

/* ------------------------------------------------------------------------------------------
 * Windowed time warping, batched over B live streams against one reference chroma.
 * ------------------------------------------------------------------------------------------ */
typedef struct rts_wtw rts_wtw;

/* layout of the int32 WTW state vector (8 entries per stream) */
#define RTS_WTW_STATE_LEN 8
#define RTS_WTW_ST_CHROMA_PTR 0
#define RTS_WTW_ST_LIVE_PTR 1
#define RTS_WTW_ST_REF_PTR 2
#define RTS_WTW_ST_STATUS 3
#define RTS_WTW_ST_N_PATH 4
#define RTS_WTW_ST_N_WINDOWS 5
#define RTS_WTW_ST_CELLS_LO 6
#define RTS_WTW_ST_CELLS_HI 7

/* Replaces the chroma-level state of WTW.__init__ (wtw.py:50-68): `chroma_ref_dev` is the reference
 * chroma [M][F] float64 (what wtw.py:37-41 computes; use rts_chroma_frames with pad_left =
 * fft_len/2), held by reference.  win_frames = dtw_win_size / hop_size, hop_frames = dtw_hop_size /
 * hop_size (wtw.py:100,:107), 1 <= win_frames <= 16384 (one workgroup per stream up to 64 frames; a
 * pipeline of workgroups per window above), hop_frames >= 1.  The handle owns a live
 * chroma history of 2M frames per stream (wtw.py:52,:55).  keep_last_d != 0 also keeps the last
 * window's accumulated-cost matrix D for inspection (the reference stores it into self.acc_cost,
 * wtw.py:105). */
int rts_wtw_create(const double *chroma_ref_dev, int F, int M, int B, int win_frames, int hop_frames,
                   int keep_last_d, rts_wtw **out);
int rts_wtw_destroy(rts_wtw *h);
int rts_wtw_reset(rts_wtw *h, void *stream);

/* Replaces the part of WTW.insert below the per-hop chroma (wtw.py:92-128): appends n_new[b] (or
 * n_max when n_new_dev is NULL) already-normalised live chroma columns per stream -- cols_dev is
 * [B][n_max][F] -- then, column by column, applies the boundary check (wtw.py:96-97) and runs every
 * window that has become available (get_cost_matrix, run_dtw, find_path, hand-over).  precheck != 0
 * first applies insert()'s entry check (wtw.py:76-77).  Status STOP_REF_END replaces the "stop"
 * return value and is sticky.  Asynchronous on `stream`. */
int rts_wtw_push(rts_wtw *h, const void *cols_dev, int cols_dtype, int n_max, const int32_t *n_new_dev,
                 int precheck, void *stream);

int rts_wtw_read_states(rts_wtw *h, int32_t *states /* [B][RTS_WTW_STATE_LEN] */, void *stream);
int rts_wtw_read_path(rts_wtw *h, int b, int32_t *pairs, int cap_pairs, int *n, void *stream);
/* The last window's accumulated-cost matrix D, [W][W] doubles (needs keep_last_d). */
int rts_wtw_read_last_d(rts_wtw *h, int b, double *d_host, void *stream);
/* Device views: live chroma history [B][2M][F] float64 and (if kept) the last window's D [B][W][W]. */
int rts_wtw_device_views(rts_wtw *h, double **live_chroma_dev, int *live_capacity, double **last_d_dev);

/* Device view of the state vectors, [B][RTS_WTW_STATE_LEN] int32 (zero-copy consumers; rts_live_* publishes from it). */
int rts_wtw_state_view(rts_wtw *h, int32_t **state_dev);

/* ------------------------------------------------------------------------------------------
 * Live ingestion: raw audio buffers of B microphones -> chroma columns -> alignment state, all on the device.
 * ------------------------------------------------------------------------------------------ */
typedef struct rts_live rts_live;

/* Replaces the audio loops that feed the trackers, for B streams per call:
 *   livenote_live.py:161-209  receive_audio / _process_input: once >= fft_len samples are pending,
 *                             `while len(data) >= 4096: col = wav_to_chroma_col(data[:4096]); ln.insert(col); data = data[2048:]`
 *   wtw.py:71-93              WTW.insert(list): self.buf += list; `while len(self.buf) >= fft_len:` one column per hop
 * Binds a chroma plan (its fft_len / hop; un-padded framing, chroma.py:35-42) and exactly one of `otw` (created with
 * the same B; columns go through rts_otw_push) or `wtw` (rts_wtw_push with precheck, i.e. wtw.py:76-77 once per feed).
 * The plan and the tracker must outlive the handle and live on the current device.  max_pending: capacity in samples of
 * each stream's pending buffer (>= fft_len + hop); a feed that would exceed it is refused with RTS_ERR_INVALID.
 * All calls of one handle must use the same `stream`. */
int rts_live_create(rts_chroma *plan, rts_otw *otw, rts_wtw *wtw, int B, int max_pending, rts_live **out);
int rts_live_destroy(rts_live *h);
/* Drops pending samples and resets the bound tracker.  Synchronises `stream`. */
int rts_live_reset(rts_live *h, void *stream);

/* One feed, zero-copy form.  rts_live_staging hands out the next pinned host staging slot (it waits only if the feed
 * that used the slot four feeds ago has not been consumed by the device yet): the producer writes counts_host[b] = new
 * samples of stream b and the samples of all streams packed back to back in stream order (float32 or int16) to
 * samples_host (capacity_samples = B * max_pending).  rts_live_submit then enqueues, without synchronising anything:
 * one host-to-device copy of the used part of the slot (on an internal copy stream, so that it overlaps the kernels of
 * the previous feed), append to the per-stream pending buffers, chroma of every complete hop, push into the tracker,
 * drop of the consumed samples (hop per column, livenote_live.py:208 / wtw.py:83), publication of the status words. */
int rts_live_staging(rts_live *h, int32_t **counts_host, void **samples_host, long long *capacity_samples);
int rts_live_submit(rts_live *h, int sample_kind /* RTS_F32 | RTS_I16 */, void *stream);
/* The same from caller-owned host arrays (one memcpy into the staging slot): samples_host packed like above. */
int rts_live_feed(rts_live *h, const void *samples_host, int sample_kind, const int32_t *counts_host, void *stream);

/* Non-blocking look at what the device last published (host-mapped memory, written at the end of every feed):
 * status[b] (RTS_RUNNING / RTS_STOP_REF_END = insert() returned "stop" / ...), positions[2b] = live frame index
 * (t / live_ptr), positions[2b+1] = reference frame index (j / ref_ptr); *feeds_done = feeds whose results these words
 * reflect for every stream, *feeds_submitted = feeds enqueued so far.  Any pointer may be NULL. */
int rts_live_poll(rts_live *h, int32_t *status /* [B] */, int32_t *positions /* [B][2] */, int *feeds_done,
                  int *feeds_submitted);
/* Samples pending per stream after everything submitted so far (the host-side mirror; exact). */
int rts_live_pending(rts_live *h, long long *pending_host /* [B] */);

#ifdef __cplusplus
}
#endif
#endif /* RTSYNC_H */
