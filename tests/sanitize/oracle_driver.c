/* Sanitizer driver for the CPU oracle ("race detection / sanitizers" -> -fsanitize=address,undefined on the
 * host code).  Compiled together with oracle/rtsync_oracle.c by tests/test_sanitize_cpu.py; reads one binary case file
 * written by that test (int32 header, float64 arrays), runs the oracle entry points the parity tests use, and prints
 * one checksum line per case -- the test compares them with the regular liboracle.so's results on the same inputs.
 * Test infrastructure only. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_otw orc_otw;
typedef struct orc_wtw orc_wtw;
orc_otw *orc_otw_create(const double *ref, int N, int F, int c, int max_run_count, int variant, int cost, int keep_cost);
void orc_otw_destroy(orc_otw *o);
int orc_otw_insert(orc_otw *o, const double *frame);
int orc_otw_set_live(orc_otw *o, const double *live, int T);
int orc_otw_run(orc_otw *o, const double *live, int T);
int64_t orc_otw_path_len(const orc_otw *o);
void orc_otw_copy_path(const orc_otw *o, int32_t *out);
void orc_otw_state(const orc_otw *o, int32_t *state);
void orc_otw_bands(const orc_otw *o, double *row_band, double *col_band);
int64_t orc_dtw(const double *a, const double *b, int M, int N, int F, double *cost, double *acc, int8_t *back, int32_t *path);
orc_wtw *orc_wtw_create(const double *ref, int M, int F, int W, int hopf);
void orc_wtw_destroy(orc_wtw *w);
int orc_wtw_insert_precheck(orc_wtw *w);
int orc_wtw_push_col(orc_wtw *w, const double *col);
int64_t orc_wtw_path_len(const orc_wtw *w);
void orc_wtw_copy_path(const orc_wtw *w, int32_t *out);
void orc_wtw_state(const orc_wtw *w, int32_t *state);

static uint64_t fnv(const void *p, size_t n, uint64_t h) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[4];
    if (fread(hdr, sizeof(int32_t), 4, f) != 4) return 2;
    const int N = hdr[0], T = hdr[1], F = hdr[2], n_cases = hdr[3];
    double *ref = (double *)malloc(sizeof(double) * (size_t)N * F);
    double *live = (double *)malloc(sizeof(double) * (size_t)T * F);
    if (fread(ref, sizeof(double), (size_t)N * F, f) != (size_t)N * F) return 2;
    if (fread(live, sizeof(double), (size_t)T * F, f) != (size_t)T * F) return 2;
    for (int k = 0; k < n_cases; k++) {
        int32_t cs[6]; /* kind (0 otw insert loop, 1 otw set_live, 2 otw frame-by-frame insert, 3 dtw, 4 wtw), variant, c | W, mrc | hop, cost, T_use */
        if (fread(cs, sizeof(int32_t), 6, f) != 6) return 2;
        const int Tu = cs[5] < T ? cs[5] : T;
        uint64_t h = 1469598103934665603ull;
        if (cs[0] <= 2) {
            orc_otw *o = orc_otw_create(ref, N, F, cs[2], cs[3], cs[1], cs[4], 1);
            if (!o) return 3;
            if (cs[0] == 0) orc_otw_run(o, live, Tu);
            if (cs[0] == 1) orc_otw_set_live(o, live, Tu);
            if (cs[0] == 2)
                for (int i = 0; i < Tu; i++)
                    if (orc_otw_insert(o, live + (size_t)i * F) != 0) break;
            const int64_t n = orc_otw_path_len(o);
            int32_t *p = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(n > 0 ? n : 1));
            orc_otw_copy_path(o, p);
            int32_t st[7];
            orc_otw_state(o, st);
            double *rb = (double *)malloc(sizeof(double) * (size_t)(cs[2] + 1)), *cb = (double *)malloc(sizeof(double) * (size_t)(cs[2] + 1));
            orc_otw_bands(o, rb, cb);
            h = fnv(p, sizeof(int32_t) * 2 * (size_t)n, h);
            h = fnv(st, sizeof(st), h);
            h = fnv(rb, sizeof(double) * (size_t)(cs[2] + 1), h);
            h = fnv(cb, sizeof(double) * (size_t)(cs[2] + 1), h);
            printf("case %d kind %d path %lld hash %016llx\n", k, cs[0], (long long)n, (unsigned long long)h);
            free(p);
            free(rb);
            free(cb);
            orc_otw_destroy(o);
        } else if (cs[0] == 3) {
            const size_t mn = (size_t)Tu * N;
            double *cost = (double *)malloc(sizeof(double) * mn), *acc = (double *)malloc(sizeof(double) * mn);
            int8_t *back = (int8_t *)malloc(mn);
            int32_t *path = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(Tu + N));
            const int64_t n = orc_dtw(live, ref, Tu, N, F, cost, acc, back, path);
            h = fnv(path, sizeof(int32_t) * 2 * (size_t)n, h);
            h = fnv(acc, sizeof(double) * mn, h);
            h = fnv(back, mn, h);
            printf("case %d kind 3 path %lld hash %016llx\n", k, (long long)n, (unsigned long long)h);
            free(cost);
            free(acc);
            free(back);
            free(path);
        } else {
            orc_wtw *w = orc_wtw_create(ref, N, F, cs[2], cs[3]);
            if (!w) return 3;
            orc_wtw_insert_precheck(w);
            for (int i = 0; i < Tu; i++)
                if (orc_wtw_push_col(w, live + (size_t)i * F) != 0) break;
            const int64_t n = orc_wtw_path_len(w);
            int32_t *p = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(n > 0 ? n : 1));
            orc_wtw_copy_path(w, p);
            int32_t st[4];
            orc_wtw_state(w, st);
            h = fnv(p, sizeof(int32_t) * 2 * (size_t)n, h);
            h = fnv(st, sizeof(st), h);
            printf("case %d kind 4 path %lld hash %016llx\n", k, (long long)n, (unsigned long long)h);
            free(p);
            orc_wtw_destroy(w);
        }
    }
    free(ref);
    free(live);
    fclose(f);
    return 0;
}
