// Device-side live ingestion for many microphones: the batched form of the reference's audio loops
//   livenote_live.py:161-209  receive_audio -> _process_input: while len(data) >= 4096: wav_to_chroma_col(data[:4096]);
//                              OnlineTimeWarping.insert(col); data = data[2048:]
//   wtw.py:71-93              WTW.insert: self.buf += list; while len(buf) >= fft_len: column of buf[:fft_len]; buf = buf[hop:]
// for B independent streams per call ("feed").
//
// One feed = ONE host-to-device copy and a fixed chain of launches on the caller's stream, no synchronisation:
//   host   counts[B], offsets[B] and the packed new samples of all streams are written into one pinned staging slot
//          (rts_live_staging hands the slot out, so a producer can write there directly) and copied by one
//          hipMemcpyAsync on the handle's copy stream; the compute stream waits for that copy by event, so the copy of
//          feed k+1 overlaps the kernels of feed k (a ring of kSlots pinned + device slots).
//   live_append_kernel    per stream (16 slices each): append the new samples (float32, or PCM16 scaled by 1/32768 like librosa.load)
//                         behind the pending ones in a per-stream device buffer; pending -> n_samples, complete hops ->
//                         n_frames  ((pending - fft_len) / hop + 1 once pending >= fft_len).
//   rts_chroma_frames_batch   un-padded framing of every stream's pending samples -> chroma columns [B][n_max][12]
//   rts_otw_push | rts_wtw_push   the columns into the alignment state (insert semantics, column by column)
//   live_compact_kernel   drop hop * n_frames consumed samples per stream (data = data[2048:], livenote_live.py:208),
//                         and publish {status, live position, ref position, feed number} of every stream into
//                         host-mapped memory -- rts_live_poll reads those words without touching the stream.
// The host keeps an exact mirror of the pending counts (integer arithmetic on the counts it was given), which is how it
// knows n_max for the launch geometry without reading anything back.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace rts {

constexpr int kLiveSlots = 4;
constexpr int kLiveWords = 4;  // per stream in the mapped status block: status, live position, ref position, feed number

struct LiveArgs {
    const unsigned char *stage;  // device copy of the staging slot: int32 counts[B], int32 offs[B], then samples at samples_off
    size_t samples_off;
    int sample_kind;             // RTS_F32 or RTS_I16
    float *buf;                  // [B][cap] pending samples
    int32_t *pending;            // [B]
    int32_t *n_samples, *n_frames;  // [B] inputs of rts_chroma_frames_batch
    int B, cap, L, hop;
    const int32_t *state;        // alignment state of the streams, state_len ints each
    int state_len, st_status, st_live, st_ref;
    int32_t *pub;                // host-mapped [B][kLiveWords]
    int feed_no;
};

__device__ __forceinline__ void live_publish(const LiveArgs &g, int b) {
    const int32_t *st = g.state + (size_t)b * g.state_len;
    volatile int32_t *p = g.pub + (size_t)b * kLiveWords;
    p[0] = st[g.st_status];
    p[1] = st[g.st_live];
    p[2] = st[g.st_ref];
    __threadfence_system();
    p[3] = g.feed_no;  // written last: a reader that sees feed k sees the three words of feed >= k
}

constexpr int kAppendSlices = 16;  // workgroups per stream in the append kernel (a 1-second buffer is 88 KB)

// grid (B, kAppendSlices): slice y of stream b copies its share of the new samples behind the pending ones.  `pending`
// is only read here (every slice needs the same base); the compact kernel, which always follows, writes it.
__global__ void __launch_bounds__(256) live_append_kernel(LiveArgs g) {
    const int b = blockIdx.x;
    const int32_t *counts = reinterpret_cast<const int32_t *>(g.stage);
    const int32_t *offs = counts + g.B;
    const int p = g.pending[b];
    int n = counts[b];
    if (n < 0) n = 0;
    if (p + n > g.cap) n = g.cap - p;  // (the host refuses such a feed before it gets here)
    const long long off = offs[b];
    float *dst = g.buf + (size_t)b * g.cap + p;
    const int per = (n + gridDim.y - 1) / gridDim.y;
    const int lo = blockIdx.y * per, hi = (lo + per < n) ? lo + per : n;
    if (g.sample_kind == RTS_F32) {
        const float *src = reinterpret_cast<const float *>(g.stage + g.samples_off) + off;
        for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) dst[i] = src[i];
    } else {
        const int16_t *src = reinterpret_cast<const int16_t *>(g.stage + g.samples_off) + off;
        for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) dst[i] = (float)src[i] * (1.0f / 32768.0f);  // exact
    }
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        const int q = p + n;
        g.n_samples[b] = q;
        g.n_frames[b] = q >= g.L ? (q - g.L) / g.hop + 1 : 0;
    }
}

// One workgroup per stream, after the tracker has taken the columns (or right after the append when no stream completed
// a hop): drops the consumed samples and publishes.
__global__ void __launch_bounds__(1024) live_compact_kernel(LiveArgs g) {
    const int b = blockIdx.x;
    const int p = g.n_samples[b];
    const int used = g.n_frames[b] * g.hop;
    const int rem = p - used;
    float *buf = g.buf + (size_t)b * g.cap;
    if (used > 0 && rem > 0) {
        // forward move of possibly overlapping ranges: every round reads its elements before any of them is written,
        // and a later round reads only above what earlier rounds wrote
        for (int base = 0; base < rem; base += blockDim.x) {
            const int i = base + threadIdx.x;
            const float v = i < rem ? buf[used + i] : 0.0f;
            __syncthreads();
            if (i < rem) buf[i] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        g.pending[b] = rem;
        live_publish(g, b);
    }
}

}  // namespace rts

struct rts_live {
    rts_chroma *plan;
    rts_otw *otw;
    rts_wtw *wtw;
    int B, L, hop, cap, cols_cap, device;
    size_t samples_off, slot_bytes;
    unsigned char *stage_host[rts::kLiveSlots];  // pinned
    unsigned char *stage_dev[rts::kLiveSlots];
    hipEvent_t copied[rts::kLiveSlots], done[rts::kLiveSlots];
    int used[rts::kLiveSlots];
    int slot;            // the slot rts_live_staging handed out last (-1: none outstanding)
    int next;
    hipStream_t copy_stream;
    float *buf;
    int32_t *pending, *n_samples, *n_frames;
    double *cols;
    long long *pending_host;
    int32_t *pub_host, *pub_dev;
    const int32_t *state_dev;
    int state_len, st_status, st_live, st_ref;
    int feeds;
};

namespace rts {
static int live_check_device(const rts_live *h) {
    int d = -1;
    RTS_HIP(hipGetDevice(&d));
    if (d != h->device)
        return set_error(RTS_ERR_INVALID, "handle was created on device %d but device %d is current", h->device, d);
    return RTS_OK;
}
}  // namespace rts

extern "C" {

int rts_live_destroy(rts_live *h) {
    if (!h) return RTS_OK;
    for (int k = 0; k < rts::kLiveSlots; k++) {
        if (h->stage_host[k]) (void)hipHostFree(h->stage_host[k]);
        if (h->stage_dev[k]) (void)hipFree(h->stage_dev[k]);
        if (h->copied[k]) (void)hipEventDestroy(h->copied[k]);
        if (h->done[k]) (void)hipEventDestroy(h->done[k]);
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->buf) (void)hipFree(h->buf);
    if (h->pending) (void)hipFree(h->pending);
    if (h->n_samples) (void)hipFree(h->n_samples);
    if (h->n_frames) (void)hipFree(h->n_frames);
    if (h->cols) (void)hipFree(h->cols);
    if (h->pub_host) (void)hipHostFree(h->pub_host);
    free(h->pending_host);
    free(h);
    return RTS_OK;
}

int rts_live_create(rts_chroma *plan, rts_otw *otw, rts_wtw *wtw, int B, int max_pending, rts_live **out) {
    using namespace rts;
    if (!out) return set_error(RTS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!plan) return set_error(RTS_ERR_INVALID, "plan is NULL");
    if ((otw == nullptr) == (wtw == nullptr)) return set_error(RTS_ERR_INVALID, "exactly one of otw / wtw must be given");
    if (B < 1) return set_error(RTS_ERR_INVALID, "B must be >= 1");
    int fft_len = 0, hop = 0;
    if (int rc = rts_chroma_plan_info(plan, &fft_len, &hop); rc != RTS_OK) return rc;
    if (max_pending < fft_len + hop || (long long)max_pending * B > 0x7fffffffLL)
        return set_error(RTS_ERR_INVALID, "max_pending must be at least fft_len + hop samples (and B * max_pending < 2^31)");
    rts_live *h = (rts_live *)calloc(1, sizeof(rts_live));
    if (!h) return set_error(RTS_ERR_INVALID, "out of host memory");
    h->plan = plan;
    h->otw = otw;
    h->wtw = wtw;
    h->B = B;
    h->L = fft_len;
    h->hop = hop;
    h->cap = max_pending;
    h->cols_cap = (max_pending - fft_len) / hop + 1;
    h->slot = -1;
    h->samples_off = (2 * sizeof(int32_t) * (size_t)B + 255) & ~(size_t)255;
    h->slot_bytes = h->samples_off + sizeof(float) * (size_t)B * max_pending;
    h->pending_host = (long long *)calloc((size_t)B, sizeof(long long));
    hipError_t e = hipGetDevice(&h->device);
    for (int k = 0; k < kLiveSlots && e == hipSuccess; k++) {
        if ((e = hipHostMalloc((void **)&h->stage_host[k], h->slot_bytes, hipHostMallocDefault)) != hipSuccess) break;
        if ((e = hipMalloc((void **)&h->stage_dev[k], h->slot_bytes)) != hipSuccess) break;
        if ((e = hipEventCreateWithFlags(&h->copied[k], hipEventDisableTiming)) != hipSuccess) break;
        if ((e = hipEventCreateWithFlags(&h->done[k], hipEventDisableTiming)) != hipSuccess) break;
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&h->buf, sizeof(float) * (size_t)B * max_pending);
    if (e == hipSuccess) e = hipMalloc((void **)&h->pending, sizeof(int32_t) * (size_t)B);
    if (e == hipSuccess) e = hipMalloc((void **)&h->n_samples, sizeof(int32_t) * (size_t)B);
    if (e == hipSuccess) e = hipMalloc((void **)&h->n_frames, sizeof(int32_t) * (size_t)B);
    if (e == hipSuccess) e = hipMalloc((void **)&h->cols, sizeof(double) * 12 * (size_t)B * h->cols_cap);
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&h->pub_host, sizeof(int32_t) * kLiveWords * (size_t)B, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&h->pub_dev, h->pub_host, 0);
    if (e == hipSuccess) e = hipMemset(h->pending, 0, sizeof(int32_t) * (size_t)B);
    if (e == hipSuccess) e = hipMemset(h->buf, 0, sizeof(float) * (size_t)B * max_pending);
    if (e != hipSuccess || !h->pending_host) {
        rts_live_destroy(h);
        return set_error(RTS_ERR_HIP, "rts_live_create: %s", hipGetErrorString(e));
    }
    memset(h->pub_host, 0, sizeof(int32_t) * kLiveWords * (size_t)B);
    int rc;
    if (otw) {
        int32_t *st = nullptr;
        rc = rts_otw_device_views(otw, nullptr, nullptr, &st);
        h->state_dev = st;
        h->state_len = RTS_STATE_LEN;
        h->st_status = RTS_ST_STATUS;
        h->st_live = RTS_ST_T;
        h->st_ref = RTS_ST_J;
    } else {
        int32_t *st = nullptr;
        rc = rts_wtw_state_view(wtw, &st);
        h->state_dev = st;
        h->state_len = RTS_WTW_STATE_LEN;
        h->st_status = RTS_WTW_ST_STATUS;
        h->st_live = RTS_WTW_ST_LIVE_PTR;
        h->st_ref = RTS_WTW_ST_REF_PTR;
    }
    if (rc != RTS_OK) {
        rts_live_destroy(h);
        return rc;
    }
    *out = h;
    return RTS_OK;
}

int rts_live_reset(rts_live *h, void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (int rc = live_check_device(h); rc != RTS_OK) return rc;
    RTS_HIP(hipStreamSynchronize((hipStream_t)stream));
    RTS_HIP(hipStreamSynchronize(h->copy_stream));
    RTS_HIP(hipMemsetAsync(h->pending, 0, sizeof(int32_t) * (size_t)h->B, (hipStream_t)stream));
    memset(h->pending_host, 0, sizeof(long long) * (size_t)h->B);
    memset(h->pub_host, 0, sizeof(int32_t) * kLiveWords * (size_t)h->B);
    memset(h->used, 0, sizeof(h->used));
    h->slot = -1;
    h->feeds = 0;
    return h->otw ? rts_otw_reset(h->otw, stream) : rts_wtw_reset(h->wtw, stream);
}

int rts_live_staging(rts_live *h, int32_t **counts_host, void **samples_host, long long *capacity_samples) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (h->slot < 0) {
        const int k = h->next;
        if (h->used[k]) RTS_HIP(hipEventSynchronize(h->done[k]));  // the feed that used this slot kSlots feeds ago
        h->used[k] = 0;
        h->slot = k;
    }
    if (counts_host) *counts_host = reinterpret_cast<int32_t *>(h->stage_host[h->slot]);
    if (samples_host) *samples_host = h->stage_host[h->slot] + h->samples_off;
    if (capacity_samples) *capacity_samples = (long long)h->B * h->cap;
    return RTS_OK;
}

int rts_live_submit(rts_live *h, int sample_kind, void *stream) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    if (h->slot < 0) return set_error(RTS_ERR_INVALID, "rts_live_submit without rts_live_staging");
    if (sample_kind != RTS_F32 && sample_kind != RTS_I16) return set_error(RTS_ERR_INVALID, "samples must be RTS_F32 or RTS_I16");
    if (int rc = live_check_device(h); rc != RTS_OK) return rc;
    const int k = h->slot, B = h->B;
    int32_t *counts = reinterpret_cast<int32_t *>(h->stage_host[k]);
    int32_t *offs = counts + B;
    // validate, prefix offsets, and the mirror of what the device will do with these counts
    long long total = 0;
    for (int b = 0; b < B; b++) {
        if (counts[b] < 0) return set_error(RTS_ERR_INVALID, "stream %d: negative sample count", b);
        if (h->pending_host[b] + counts[b] > h->cap)
            return set_error(RTS_ERR_INVALID, "stream %d: %lld pending + %d new samples exceed max_pending = %d", b,
                             h->pending_host[b], counts[b], h->cap);
        total += counts[b];
    }
    int n_max = 0;
    total = 0;
    for (int b = 0; b < B; b++) {
        offs[b] = (int32_t)total;
        total += counts[b];
        const long long q = h->pending_host[b] + counts[b];
        const int nf = q >= h->L ? (int)((q - h->L) / h->hop + 1) : 0;
        if (nf > n_max) n_max = nf;
        h->pending_host[b] = q - (long long)nf * h->hop;
    }
    h->slot = -1;
    h->next = (k + 1) % kLiveSlots;
    h->used[k] = 1;
    h->feeds += 1;
    hipStream_t s = (hipStream_t)stream;
    const size_t bytes = h->samples_off + (size_t)total * (sample_kind == RTS_F32 ? sizeof(float) : sizeof(int16_t));
    RTS_HIP(hipMemcpyAsync(h->stage_dev[k], h->stage_host[k], bytes, hipMemcpyHostToDevice, h->copy_stream));
    RTS_HIP(hipEventRecord(h->copied[k], h->copy_stream));
    RTS_HIP(hipStreamWaitEvent(s, h->copied[k], 0));
    LiveArgs g;
    memset(&g, 0, sizeof(g));
    g.stage = h->stage_dev[k];
    g.samples_off = h->samples_off;
    g.sample_kind = sample_kind;
    g.buf = h->buf;
    g.pending = h->pending;
    g.n_samples = h->n_samples;
    g.n_frames = h->n_frames;
    g.B = B;
    g.cap = h->cap;
    g.L = h->L;
    g.hop = h->hop;
    g.state = h->state_dev;
    g.state_len = h->state_len;
    g.st_status = h->st_status;
    g.st_live = h->st_live;
    g.st_ref = h->st_ref;
    g.pub = h->pub_dev;
    g.feed_no = h->feeds;
    hipLaunchKernelGGL(live_append_kernel, dim3(B, kAppendSlices), dim3(256), 0, s, g);
    RTS_HIP(hipGetLastError());
    RTS_HIP(hipEventRecord(h->done[k], s));  // the staging slot (host and device side) is free again after this point
    if (n_max == 0) {
        if (h->wtw) {  // wtw.py:76-77 runs on every insert(), new column or not
            if (int rc = rts_wtw_push(h->wtw, nullptr, RTS_F64, 0, nullptr, 1, stream); rc != RTS_OK) return rc;
        }
        hipLaunchKernelGGL(live_compact_kernel, dim3(B), dim3(1024), 0, s, g);  // nothing to drop: pending, publication
        RTS_HIP(hipGetLastError());
        return RTS_OK;
    }
    int rc = rts_chroma_frames_batch(h->plan, h->buf, RTS_F32, h->cap, h->n_samples, 0, B, n_max, h->n_frames, 1, h->cols,
                                     RTS_F64, stream);
    if (rc != RTS_OK) return rc;
    rc = h->otw ? rts_otw_push(h->otw, h->cols, RTS_F64, n_max, h->n_frames, stream)
                : rts_wtw_push(h->wtw, h->cols, RTS_F64, n_max, h->n_frames, 1, stream);
    if (rc != RTS_OK) return rc;
    hipLaunchKernelGGL(live_compact_kernel, dim3(B), dim3(1024), 0, s, g);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

int rts_live_feed(rts_live *h, const void *samples_host, int sample_kind, const int32_t *counts_host, void *stream) {
    using namespace rts;
    if (!h || !counts_host) return set_error(RTS_ERR_INVALID, "NULL argument");
    if (sample_kind != RTS_F32 && sample_kind != RTS_I16) return set_error(RTS_ERR_INVALID, "samples must be RTS_F32 or RTS_I16");
    int32_t *counts = nullptr;
    void *samples = nullptr;
    long long capacity = 0;
    if (int rc = rts_live_staging(h, &counts, &samples, &capacity); rc != RTS_OK) return rc;
    long long total = 0;
    for (int b = 0; b < h->B; b++) {
        if (counts_host[b] < 0) return set_error(RTS_ERR_INVALID, "stream %d: negative sample count", b);
        total += counts_host[b];
    }
    if (total > capacity) return set_error(RTS_ERR_INVALID, "%lld samples in one feed exceed B * max_pending = %lld", total, capacity);
    if (total > 0 && !samples_host) return set_error(RTS_ERR_INVALID, "samples_host is NULL");
    memcpy(counts, counts_host, sizeof(int32_t) * (size_t)h->B);
    if (total > 0) memcpy(samples, samples_host, (size_t)total * (sample_kind == RTS_F32 ? sizeof(float) : sizeof(int16_t)));
    return rts_live_submit(h, sample_kind, stream);
}

int rts_live_poll(rts_live *h, int32_t *status, int32_t *positions, int *feeds_done, int *feeds_submitted) {
    using namespace rts;
    if (!h) return set_error(RTS_ERR_INVALID, "handle is NULL");
    const volatile int32_t *p = h->pub_host;
    int fd = h->feeds;
    for (int b = 0; b < h->B; b++) {
        const int seq = p[(size_t)b * kLiveWords + 3];
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (status) status[b] = p[(size_t)b * kLiveWords + 0];
        if (positions) {
            positions[2 * b] = p[(size_t)b * kLiveWords + 1];
            positions[2 * b + 1] = p[(size_t)b * kLiveWords + 2];
        }
        if (seq < fd) fd = seq;
    }
    if (feeds_done) *feeds_done = fd;
    if (feeds_submitted) *feeds_submitted = h->feeds;
    return RTS_OK;
}

int rts_live_pending(rts_live *h, long long *pending_host /* [B] */) {
    using namespace rts;
    if (!h || !pending_host) return set_error(RTS_ERR_INVALID, "NULL argument");
    memcpy(pending_host, h->pending_host, sizeof(long long) * (size_t)h->B);
    return RTS_OK;
}

}  // extern "C"
