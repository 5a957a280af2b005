/* Sanitizer driver for the HOST half of librtsync.so: every argument-validation and early-error path of the C-ABI
 * (include/rtsync.h) is called without a GPU against a host-only -fsanitize=address,undefined build of csrc/
 * (hipcc --cuda-host-only; the kernels are not even compiled).  The expected return code of every call is asserted; the
 * sanitizers watch the error paths for leaks, overflows and undefined behaviour.  tests/test_sanitize_cpu.py builds and
 * runs it.  No compute entry point can run here: there is no device code in the library under test. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/rtsync.h"

static int fails = 0;
#define EXPECT(call, want)                                                                      \
    do {                                                                                        \
        const int rc_ = (call);                                                                 \
        if (rc_ != (want)) {                                                                    \
            fprintf(stderr, "%s:%d: %s -> %d (%s), expected %d\n", __FILE__, __LINE__, #call, rc_, rts_last_error(), (want)); \
            fails++;                                                                            \
        } else if ((want) != RTS_OK && strlen(rts_last_error()) == 0) {                         \
            fprintf(stderr, "%s:%d: %s failed without a message\n", __FILE__, __LINE__, #call); \
            fails++;                                                                            \
        }                                                                                       \
    } while (0)

int main(void) {
    void *fake = (void *)(uintptr_t)4096; /* never dereferenced on the paths below */
    printf("version %d, gfx950 devices %d\n", rts_version(), rts_device_count());

    /* ---- OTW ---- */
    rts_otw *o = NULL;
    EXPECT(rts_otw_create(fake, RTS_F32, 12, 100, 1, 50, 3, RTS_VARIANT_OTW, RTS_COST_DOT, NULL), RTS_ERR_INVALID);
    EXPECT(rts_otw_create(NULL, RTS_F32, 12, 100, 1, 50, 3, RTS_VARIANT_OTW, RTS_COST_DOT, &o), RTS_ERR_INVALID);
    EXPECT(rts_otw_create(fake, RTS_F32, 0, 100, 1, 50, 3, RTS_VARIANT_OTW, RTS_COST_DOT, &o), RTS_ERR_INVALID);
    EXPECT(rts_otw_create(fake, RTS_F32, 12, 0, 1, 50, 3, RTS_VARIANT_OTW, RTS_COST_DOT, &o), RTS_ERR_INVALID);
    EXPECT(rts_otw_create(fake, RTS_F32, 12, 100, 0, 50, 3, RTS_VARIANT_OTW, RTS_COST_DOT, &o), RTS_ERR_INVALID);
    EXPECT(rts_otw_create(fake, 7, 12, 100, 1, 50, 3, RTS_VARIANT_OTW, RTS_COST_DOT, &o), RTS_ERR_INVALID);
    EXPECT(rts_otw_create(fake, RTS_F32, 12, 100, 1, 0, 3, RTS_VARIANT_OTW, RTS_COST_DOT, &o), RTS_ERR_INVALID);
    EXPECT(rts_otw_create(fake, RTS_F32, 12, 100, 1, 50, 0, RTS_VARIANT_OTW, RTS_COST_DOT, &o), RTS_ERR_INVALID);
    EXPECT(rts_otw_create(fake, RTS_F32, 12, 100, 1, 50, 3, 9, RTS_COST_DOT, &o), RTS_ERR_INVALID);
    EXPECT(rts_otw_create(fake, RTS_F32, 12, 100, 1, 50, 3, RTS_VARIANT_OTW, 5, &o), RTS_ERR_INVALID);
    EXPECT(rts_otw_create(fake, RTS_F32, 12, 0x7fffffff, 1, 50, 3, RTS_VARIANT_OTW, RTS_COST_DOT, &o), RTS_ERR_INVALID);
    /* valid arguments, no device: the handle is built, the first HIP call fails, everything allocated so far is released */
    EXPECT(rts_otw_create(fake, RTS_F64, 12, 100, 4, 50, 3, RTS_VARIANT_LIVENOTE_V2, RTS_COST_EUCLID, &o), RTS_ERR_HIP);
    if (o != NULL) { fprintf(stderr, "rts_otw_create left a handle behind on failure\n"); fails++; }
    EXPECT(rts_otw_destroy(NULL), RTS_OK);
    EXPECT(rts_otw_reset(NULL, NULL), RTS_ERR_INVALID);
    EXPECT(rts_otw_run(NULL, fake, RTS_F32, 10, (const int32_t *)fake, RTS_MODE_INSERT_LOOP, NULL), RTS_ERR_INVALID);
    EXPECT(rts_otw_insert(NULL, fake, RTS_F32, NULL, NULL), RTS_ERR_INVALID);
    EXPECT(rts_otw_push(NULL, fake, RTS_F32, 3, NULL, NULL), RTS_ERR_INVALID);
    int32_t st[RTS_STATE_LEN];
    int n = 0;
    double band[8];
    EXPECT(rts_otw_read_state(NULL, 0, st, NULL), RTS_ERR_INVALID);
    EXPECT(rts_otw_read_states(NULL, st, NULL), RTS_ERR_INVALID);
    EXPECT(rts_otw_read_path(NULL, 0, NULL, 0, &n, NULL), RTS_ERR_INVALID);
    EXPECT(rts_otw_read_bands(NULL, 0, band, band, NULL), RTS_ERR_INVALID);
    EXPECT(rts_otw_device_views(NULL, NULL, NULL, NULL), RTS_ERR_INVALID);
    EXPECT(rts_otw_set_dense(NULL, NULL, NULL, NULL), RTS_ERR_INVALID);
    EXPECT(rts_otw_replay_dense(NULL, NULL, RTS_F64, 0, NULL, band, band, NULL), RTS_ERR_INVALID);
    EXPECT(rts_otw_set_waves(NULL, 8), RTS_ERR_INVALID);

    /* ---- DTW ---- */
    size_t bytes = 0;
    EXPECT(rts_dtw_workspace_bytes(10, 10, 1, NULL), RTS_ERR_INVALID);
    EXPECT(rts_dtw_workspace_bytes(0, 10, 1, &bytes), RTS_ERR_INVALID);
    EXPECT(rts_dtw_workspace_bytes(322, 322, 1, &bytes), RTS_OK);
    if (bytes < 322u * 322u / 4) { fprintf(stderr, "workspace of %zu bytes is implausibly small\n", bytes); fails++; }
    int32_t plen = 0;
    EXPECT(rts_dtw(NULL, RTS_F32, 0, fake, RTS_F32, 0, 12, 10, 10, 1, band, band, NULL, st, &plen, fake, bytes, NULL), RTS_ERR_INVALID);
    EXPECT(rts_dtw(fake, RTS_F32, 0, fake, RTS_F32, 0, 13, 10, 10, 1, band, band, NULL, st, &plen, fake, bytes, NULL), RTS_ERR_UNSUPPORTED);
    EXPECT(rts_dtw(fake, RTS_F32, 0, fake, RTS_F32, 0, 12, 0, 10, 1, band, band, NULL, st, &plen, fake, bytes, NULL), RTS_ERR_INVALID);
    EXPECT(rts_dtw(fake, 3, 0, fake, RTS_F32, 0, 12, 10, 10, 1, band, band, NULL, st, &plen, fake, bytes, NULL), RTS_ERR_INVALID);
    EXPECT(rts_dtw(fake, RTS_F32, 0, fake, RTS_F32, 0, 12, 10, 10, 70000, band, band, NULL, st, &plen, fake, bytes, NULL), RTS_ERR_INVALID);
    EXPECT(rts_dtw(fake, RTS_F32, 0, fake, RTS_F32, 0, 12, 322, 322, 1, band, band, NULL, st, &plen, fake, 16, NULL), RTS_ERR_INVALID);
    EXPECT(rts_dtw(fake, RTS_F32, 0, fake, RTS_F32, 0, 12, 322, 322, 1, band, band, NULL, st, &plen, (void *)(uintptr_t)4100, bytes, NULL), RTS_ERR_INVALID);

    /* ---- chroma ---- */
    if (rts_chroma_num_frames(661500, 4096, 2048, 2048) != 322 || rts_chroma_num_frames(4095, 4096, 2048, 0) != 0 ||
        rts_chroma_num_frames(4096, 4096, 2048, 0) != 1 || rts_chroma_num_frames(100, 1, 2048, 0) != 0) {
        fprintf(stderr, "rts_chroma_num_frames\n");
        fails++;
    }
    rts_chroma *c = NULL;
    static double fb[12 * 33];
    EXPECT(rts_chroma_create(64, 32, NULL, fb, NULL), RTS_ERR_INVALID);
    EXPECT(rts_chroma_create(64, 32, NULL, NULL, &c), RTS_ERR_INVALID);
    EXPECT(rts_chroma_create(100, 32, NULL, fb, &c), RTS_ERR_UNSUPPORTED);
    EXPECT(rts_chroma_create(32, 32, NULL, fb, &c), RTS_ERR_UNSUPPORTED);
    EXPECT(rts_chroma_create(64, 0, NULL, fb, &c), RTS_ERR_INVALID);
    EXPECT(rts_chroma_create(64, 32, NULL, fb, &c), RTS_ERR_HIP); /* valid, no device */
    if (c != NULL) { fprintf(stderr, "rts_chroma_create left a handle behind on failure\n"); fails++; }
    EXPECT(rts_chroma_destroy(NULL), RTS_OK);
    EXPECT(rts_chroma_plan_info(NULL, &n, &n), RTS_ERR_INVALID);
    EXPECT(rts_chroma_frames(NULL, fake, RTS_F32, 10, 0, 1, 1, fake, RTS_F64, NULL, NULL), RTS_ERR_INVALID);
    EXPECT(rts_chroma_frames_batch(NULL, fake, RTS_F32, 10, st, 0, 1, 1, st, 1, fake, RTS_F64, NULL), RTS_ERR_INVALID);
    EXPECT(rts_chroma_project(NULL, band, 1, 1, fake, RTS_F64, NULL), RTS_ERR_INVALID);
    EXPECT(rts_chroma_diff(NULL, RTS_F64, 5, fake, NULL), RTS_ERR_INVALID);
    EXPECT(rts_chroma_diff(fake, 9, 5, fake, NULL), RTS_ERR_INVALID);
    EXPECT(rts_chroma_diff(fake, RTS_F64, 1, fake, NULL), RTS_OK); /* fewer than two frames: nothing to do */

    /* ---- WTW ---- */
    rts_wtw *w = NULL;
    EXPECT(rts_wtw_create(band, 12, 100, 1, 20, 10, 0, NULL), RTS_ERR_INVALID);
    EXPECT(rts_wtw_create(NULL, 12, 100, 1, 20, 10, 0, &w), RTS_ERR_INVALID);
    EXPECT(rts_wtw_create(band, 0, 100, 1, 20, 10, 0, &w), RTS_ERR_INVALID);
    EXPECT(rts_wtw_create(band, 12, 0, 1, 20, 10, 0, &w), RTS_ERR_INVALID);
    EXPECT(rts_wtw_create(band, 12, 100, 1, 0, 10, 0, &w), RTS_ERR_INVALID);
    EXPECT(rts_wtw_create(band, 12, 100, 1, 20, 0, 0, &w), RTS_ERR_INVALID);
    EXPECT(rts_wtw_create(band, 12, 100, 1, 1 << 20, 10, 0, &w), RTS_ERR_UNSUPPORTED);
    EXPECT(rts_wtw_create(band, 12, 100, 2, 20, 10, 1, &w), RTS_ERR_HIP); /* valid, no device */
    EXPECT(rts_wtw_create(band, 12, 3000, 2, 700, 350, 0, &w), RTS_ERR_HIP); /* the strip-DP configuration path */
    if (w != NULL) { fprintf(stderr, "rts_wtw_create left a handle behind on failure\n"); fails++; }
    EXPECT(rts_wtw_destroy(NULL), RTS_OK);
    EXPECT(rts_wtw_reset(NULL, NULL), RTS_ERR_INVALID);
    EXPECT(rts_wtw_push(NULL, fake, RTS_F64, 1, NULL, 1, NULL), RTS_ERR_INVALID);
    EXPECT(rts_wtw_read_states(NULL, st, NULL), RTS_ERR_INVALID);
    EXPECT(rts_wtw_read_path(NULL, 0, NULL, 0, &n, NULL), RTS_ERR_INVALID);
    EXPECT(rts_wtw_read_last_d(NULL, 0, band, NULL), RTS_ERR_INVALID);
    EXPECT(rts_wtw_device_views(NULL, NULL, NULL, NULL), RTS_ERR_INVALID);
    EXPECT(rts_wtw_state_view(NULL, NULL), RTS_ERR_INVALID);

    /* ---- live ingestion ---- */
    rts_live *l = NULL;
    EXPECT(rts_live_create((rts_chroma *)fake, NULL, NULL, 1, 1 << 16, NULL), RTS_ERR_INVALID);
    EXPECT(rts_live_create(NULL, (rts_otw *)fake, NULL, 1, 1 << 16, &l), RTS_ERR_INVALID);
    EXPECT(rts_live_create((rts_chroma *)fake, NULL, NULL, 1, 1 << 16, &l), RTS_ERR_INVALID);
    EXPECT(rts_live_create((rts_chroma *)fake, (rts_otw *)fake, (rts_wtw *)fake, 1, 1 << 16, &l), RTS_ERR_INVALID);
    EXPECT(rts_live_create((rts_chroma *)fake, (rts_otw *)fake, NULL, 0, 1 << 16, &l), RTS_ERR_INVALID);
    EXPECT(rts_live_destroy(NULL), RTS_OK);
    EXPECT(rts_live_reset(NULL, NULL), RTS_ERR_INVALID);
    EXPECT(rts_live_staging(NULL, NULL, NULL, NULL), RTS_ERR_INVALID);
    EXPECT(rts_live_submit(NULL, RTS_F32, NULL), RTS_ERR_INVALID);
    EXPECT(rts_live_feed(NULL, fake, RTS_F32, st, NULL), RTS_ERR_INVALID);
    EXPECT(rts_live_poll(NULL, NULL, NULL, NULL, NULL), RTS_ERR_INVALID);
    EXPECT(rts_live_pending(NULL, NULL), RTS_ERR_INVALID);

    printf("%s (%d mismatches)\n", fails ? "FAILED" : "ok", fails);
    return fails ? 1 : 0;
}
