/*
 * oracle/rtsync_oracle.c -- CPU restatement of the reference's alignment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker.  The product path (real_time_audio_sync_amd/) never falls back to it.
 *
 * What is restated (citations are /root/reference/<file>:<line>):
 *   - OnlineTimeWarping            otw_eran.py:5-239     (variant ORC_OTW)
 *   - LiveNote                     livenote.py:3-226     (variant ORC_LIVENOTE)
 *   - LiveNoteV2                   livenote_v2.py:3-236  (variant ORC_LIVENOTE_V2, dot or Euclid cost)
 *   - DTW                          dtw.py:5-53
 *   - WTW window DP + hand-over    wtw.py:96-128, 162-240
 * The restatement is *dense* like the reference (a (2N x N) float64 accumulated-cost matrix
 * initialised to the sentinel), so it shares no band-compaction logic with the HIP kernels.
 *
 * Floating point: float64 throughout, built with -ffp-contract=off; every fma() below is explicit.
 * The 12-term dot products follow the summation order numpy's BLAS (OpenBLAS 0.3.29, the build
 * numpy 2.2 ships here) uses at the reference's call sites, so that accumulated costs -- not only
 * path indices -- can be pinned bit-for-bit against outputs of the reference's own code run in
 * the build container (tests/golden/make_golden.py):
 *   - np.dot(col_view, col_view) (strided ddot; otw_eran.py:220, livenote_v2.py:170, wtw.py:169)
 *        -> orc_dot_strided(): 4-way unrolled, two accumulators;
 *   - np.dot(A.T, B) (dgemm, dtw.py:11) and x.dot(x) on a contiguous copy (np.linalg.norm,
 *        wtw.py:169) -> orc_dot_chain(): sequential fma chain;
 *   - np.sum of 12 contiguous doubles (livenote_v2.py:168) -> numpy pairwise-sum order.
 *
 * Layout at this C level is frame-major ([frame][feature], feature stride 1); the Python callers
 * transpose the reference's feature-major (12, N) arrays.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OTW 0
#define ORC_LIVENOTE 1
#define ORC_LIVENOTE_V2 2

#define ORC_COST_DOT 0
#define ORC_COST_EUCLID 1

#define ORC_DIR_NONE (-1)
#define ORC_DIR_BOTH 0
#define ORC_DIR_ROW 1
#define ORC_DIR_COLUMN 2

#define ORC_RUNNING 0
#define ORC_STOP_REF_END 1   /* insert returned "stop" (otw_eran.py:69-71) */
#define ORC_LIVE_OVERFLOW 2  /* ran out of the 2N pre-allocated live rows (otw_eran.py:53-55) */

/* ---------------------------------------------------------------- dot products */

/* np.dot on two strided column views: OpenBLAS ddot, inc != 1 (see header). */
double orc_dot_strided(const double *x, const double *y, int n) {
    double t1 = 0.0, t2 = 0.0;
    int i = 0, n1 = n & -4;
    while (i < n1) {
        double m3 = y[i + 2] * x[i + 2];
        double m4 = y[i + 3] * x[i + 3];
        double a = fma(y[i], x[i], m3);
        double b = fma(y[i + 1], x[i + 1], m4);
        t1 = t1 + a;
        t2 = t2 + b;
        i += 4;
    }
    while (i < n) {
        t1 = fma(y[i], x[i], t1);
        i++;
    }
    return t1 + t2;
}

/* dgemm element / contiguous ddot for n < 32: sequential fma chain. */
double orc_dot_chain(const double *x, const double *y, int n) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s = fma(x[i], y[i], s);
    return s;
}

/* np.sqrt(np.sum((a-b)**2)) with numpy's pairwise summation order (8 partials when n >= 8). */
double orc_euclid(const double *a, const double *b, int n) {
    double sq[64];
    if (n > 64) n = 64; /* chroma has 12 features; guard only */
    for (int i = 0; i < n; i++) {
        double d = a[i] - b[i];
        sq[i] = d * d;
    }
    double res;
    int i;
    if (n < 8) {
        res = 0.0;
        for (i = 0; i < n; i++) res = res + sq[i];
    } else {
        double r[8];
        for (i = 0; i < 8; i++) r[i] = sq[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int q = 0; q < 8; q++) r[q] = r[q] + sq[i + q];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res = res + sq[i];
    }
    return sqrt(res);
}

/* ---------------------------------------------------------------- OTW / LiveNote / LiveNoteV2 */

typedef struct {
    int F, N, M; /* features, ref length, live capacity = 2N (otw_eran.py:12-14) */
    int c, max_run_count, variant, cost_kind;
    double *ref;  /* [N][F] */
    double *live; /* [M][F] */
    double *acc;  /* [M][N], sentinel-initialised (otw_eran.py:27, livenote_v2.py:22-23) */
    double *cost; /* [M][N] or NULL; -1 initialised (otw_eran.py:23) */
    double sentinel;
    int t, j, previous, run_count, direction, first_insert, status;
    int32_t *path; /* pairs (live, ref) */
    int64_t n_path, path_cap;
    /* instrumentation (not in the reference) */
    int64_t cells, n_row_strips, n_col_strips, n_inserts;
    int64_t max_left_run, sum_strip_left_run; /* longest run of cells whose min came from the in-strip neighbour */
    int64_t cur_left_run, strip_left_run;
} orc_otw;

static void otw_reset_state(orc_otw *o) {
    o->t = 0;
    o->j = 0;
    o->previous = ORC_DIR_NONE;
    /* otw_eran.py:33 starts at 1, livenote_v2.py:35 at 0 */
    o->run_count = (o->variant == ORC_OTW) ? 1 : 0;
    o->direction = ORC_DIR_BOTH;
    o->n_path = 0;
    o->first_insert = 1;
    o->status = ORC_RUNNING;
}

orc_otw *orc_otw_create(const double *ref, int N, int F, int c, int max_run_count, int variant,
                        int cost_kind, int keep_cost) {
    orc_otw *o = (orc_otw *)calloc(1, sizeof(orc_otw));
    if (!o) return NULL;
    o->F = F;
    o->N = N;
    o->M = 2 * N;
    o->c = c;
    o->max_run_count = max_run_count;
    o->variant = variant;
    o->cost_kind = cost_kind;
    o->sentinel = (variant == ORC_OTW) ? 1e10 : INFINITY;
    size_t mn = (size_t)o->M * (size_t)N;
    o->ref = (double *)malloc(sizeof(double) * (size_t)N * F);
    o->live = (double *)malloc(sizeof(double) * (size_t)o->M * F);
    o->acc = (double *)malloc(sizeof(double) * mn);
    o->cost = keep_cost ? (double *)malloc(sizeof(double) * mn) : NULL;
    o->path_cap = 4 * (int64_t)o->M + 16;
    o->path = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)o->path_cap);
    if (!o->ref || !o->live || !o->acc || !o->path || (keep_cost && !o->cost)) return NULL;
    memcpy(o->ref, ref, sizeof(double) * (size_t)N * F);
    for (size_t i = 0; i < (size_t)o->M * F; i++) o->live[i] = (variant == ORC_OTW) ? -1.0 : 0.0;
    for (size_t i = 0; i < mn; i++) o->acc[i] = o->sentinel;
    if (o->cost)
        for (size_t i = 0; i < mn; i++) o->cost[i] = -1.0;
    otw_reset_state(o);
    return o;
}

void orc_otw_destroy(orc_otw *o) {
    if (!o) return;
    free(o->ref);
    free(o->live);
    free(o->acc);
    free(o->cost);
    free(o->path);
    free(o);
}

/* otw_eran.py:215-239 / livenote_v2.py:165-189.  in_strip_dir: 0 = cell's in-strip predecessor is
 * (x, y-1) (row strip), 1 = (x-1, y) (column strip); used only for the left-run statistics. */
static void otw_eval(orc_otw *o, int x, int y, int in_strip_dir) {
    const double *lv = o->live + (size_t)x * o->F;
    const double *rf = o->ref + (size_t)y * o->F;
    double d;
    if (o->cost_kind == ORC_COST_EUCLID)
        d = orc_euclid(lv, rf, o->F);
    else
        d = 1.0 - orc_dot_strided(lv, rf, o->F);
    size_t N = (size_t)o->N;
    if (o->cost) o->cost[(size_t)x * N + y] = d;
    o->cells++;
    if (x == 0 && y == 0) {
        o->acc[0] = d;
        return;
    }
    double best = INFINITY, left = INFINITY, up = INFINITY;
    int any = 0;
    if (y > 0) {
        left = o->acc[(size_t)x * N + (y - 1)] + d;
        best = left;
        any = 1;
    }
    if (x > 0) {
        up = o->acc[(size_t)(x - 1) * N + y] + d;
        if (!any || up < best) best = up;
        any = 1;
    }
    if (x > 0 && y > 0) {
        double dg = o->acc[(size_t)(x - 1) * N + (y - 1)] + 2 * d;
        if (dg < best) best = dg;
    }
    o->acc[(size_t)x * N + y] = best;
    /* statistics: did the in-strip neighbour carry the minimum? */
    double carried = in_strip_dir ? up : left;
    if (carried == best) {
        o->cur_left_run++;
        if (o->cur_left_run > o->strip_left_run) o->strip_left_run = o->cur_left_run;
    } else {
        o->cur_left_run = 0;
    }
}

static void otw_strip_begin(orc_otw *o) {
    o->cur_left_run = 0;
    o->strip_left_run = 0;
}
static void otw_strip_end(orc_otw *o) {
    if (o->strip_left_run > o->max_left_run) o->max_left_run = o->strip_left_run;
    o->sum_strip_left_run += o->strip_left_run;
}

static void otw_row_strip(orc_otw *o) { /* otw_eran.py:58-62 */
    int k1 = o->j - o->c + 1;
    if (k1 < 0) k1 = 0;
    otw_strip_begin(o);
    for (int k = k1; k < o->j + 1; k++) otw_eval(o, o->t, k, 0);
    otw_strip_end(o);
    o->n_row_strips++;
}

static void otw_col_strip(orc_otw *o) { /* otw_eran.py:73-77 */
    int k1 = o->t - o->c + 1;
    if (k1 < 0) k1 = 0;
    otw_strip_begin(o);
    for (int k = k1; k < o->t + 1; k++) otw_eval(o, k, o->j, 1);
    otw_strip_end(o);
    o->n_col_strips++;
}

/* otw_eran.py:192-211 / livenote_v2.py:219-236: np.argmin = first minimum; row wins only if
 * strictly smaller. */
static void otw_best_point(const orc_otw *o, int *bx, int *by) {
    size_t N = (size_t)o->N;
    int j1 = o->j - o->c + 1;
    if (j1 < 0) j1 = 0;
    int best_j = j1;
    double cost_j = o->acc[(size_t)o->t * N + j1];
    for (int k = j1 + 1; k < o->j + 1; k++) {
        double v = o->acc[(size_t)o->t * N + k];
        if (v < cost_j) {
            cost_j = v;
            best_j = k;
        }
    }
    int t1 = o->t - o->c + 1;
    if (t1 < 0) t1 = 0;
    int best_t = t1;
    double cost_t = o->acc[(size_t)t1 * N + o->j];
    for (int k = t1 + 1; k < o->t + 1; k++) {
        double v = o->acc[(size_t)k * N + o->j];
        if (v < cost_t) {
            cost_t = v;
            best_t = k;
        }
    }
    if (cost_j < cost_t) {
        *bx = o->t;
        *by = best_j;
    } else {
        *bx = best_t;
        *by = o->j;
    }
}

static void otw_path_append(orc_otw *o, int x, int y) {
    if (o->n_path >= o->path_cap) {
        o->path_cap *= 2;
        o->path = (int32_t *)realloc(o->path, sizeof(int32_t) * 2 * (size_t)o->path_cap);
    }
    o->path[2 * o->n_path] = x;
    o->path[2 * o->n_path + 1] = y;
    o->n_path++;
}

/* otw_eran.py:153-178 (first half of set_direction) == livenote_v2.py:193-217 (get_direction):
 * best point -> path, then the direction rule.  Returns the new direction. */
static int otw_choose(orc_otw *o) {
    int x, y;
    otw_best_point(o, &x, &y);
    if (o->variant == ORC_LIVENOTE_V2) { /* livenote_v2.py:198-199: forward-moving points only */
        if (o->n_path == 0 ||
            (x > o->path[2 * (o->n_path - 1)] && y >= o->path[2 * (o->n_path - 1) + 1]))
            otw_path_append(o, x, y);
    } else {
        otw_path_append(o, x, y);
    }
    int dir;
    if (o->t < o->c)
        dir = ORC_DIR_BOTH;
    else if (o->run_count >= o->max_run_count)
        dir = (o->previous == ORC_DIR_ROW) ? ORC_DIR_COLUMN : ORC_DIR_ROW;
    else if (x < o->t)
        dir = ORC_DIR_COLUMN;
    else if (y < o->j)
        dir = ORC_DIR_ROW;
    else
        dir = ORC_DIR_BOTH;
    return dir;
}

/* otw_eran.py:180-188 (second half of set_direction) == livenote_v2.py:94-100 / :149-155. */
static void otw_update_run(orc_otw *o, int dir) {
    if (dir == o->previous)
        o->run_count += 1;
    else
        o->run_count = 1;
    if (dir != ORC_DIR_BOTH) o->previous = dir;
}

static int otw_decide(orc_otw *o) {
    int dir = otw_choose(o);
    otw_update_run(o, dir);
    return dir;
}

/* otw_eran.py:38-85 / livenote_v2.py:43-104.  Returns the status after this insert. */
int orc_otw_insert(orc_otw *o, const double *frame) {
    /* The reference would raise IndexError on an insert after "stop"; callers break on it
     * (test_simple.py:124-125).  The restatement is sticky instead. */
    if (o->status == ORC_STOP_REF_END) return o->status;
    o->n_inserts++;
    if (o->first_insert) {
        o->first_insert = 0;
        memcpy(o->live + (size_t)o->t * o->F, frame, sizeof(double) * o->F);
        otw_eval(o, o->t, o->j, 0);
        return o->status;
    }
    o->t += 1;
    if (o->t >= o->M) {
        o->status = ORC_LIVE_OVERFLOW;
        return o->status;
    }
    memcpy(o->live + (size_t)o->t * o->F, frame, sizeof(double) * o->F);
    otw_row_strip(o);
    for (;;) {
        if (o->direction != ORC_DIR_ROW) {
            o->j += 1;
            if (o->j >= o->N) {
                o->status = ORC_STOP_REF_END;
                return o->status;
            }
            otw_col_strip(o);
        }
        o->direction = otw_decide(o);
        if (o->direction != ORC_DIR_COLUMN) break;
    }
    return o->status;
}

/* otw_eran.py:91-142 / livenote_v2.py:108-155 (whole live sequence at once; decide() runs at the
 * top of each iteration, so the path starts with the best point of cell (0,0)). */
int orc_otw_set_live(orc_otw *o, const double *live, int T) {
    if (o->variant == ORC_OTW) otw_reset_state(o); /* otw_eran.py:92-97; LiveNote does not reset */
    o->first_insert = 0;
    memcpy(o->live + (size_t)o->t * o->F, live + (size_t)o->t * o->F, sizeof(double) * o->F);
    otw_eval(o, o->t, o->j, 0);
    for (;;) {
        /* OnlineTimeWarping updates run_count/previous inside set_direction; LiveNote does it at
         * the bottom of the loop, i.e. not at all for the iteration that breaks out. */
        int dir = otw_choose(o);
        if (o->variant == ORC_OTW) {
            otw_update_run(o, dir);
            o->direction = dir; /* LiveNote keeps it in a local */
        }
        if (dir != ORC_DIR_COLUMN) {
            o->t += 1;
            if (o->t >= T) break;
            if (o->t >= o->M) {
                o->status = ORC_LIVE_OVERFLOW;
                break;
            }
            memcpy(o->live + (size_t)o->t * o->F, live + (size_t)o->t * o->F, sizeof(double) * o->F);
            otw_row_strip(o);
        }
        if (dir != ORC_DIR_ROW) {
            o->j += 1;
            if (o->j >= o->N) {
                o->status = ORC_STOP_REF_END;
                break;
            }
            otw_col_strip(o);
        }
        if (o->variant != ORC_OTW) otw_update_run(o, dir);
    }
    return o->status;
}

/* Insert-loop driver (tests.py:160-163): insert frames until "stop" or the live sequence ends.
 * Returns the number of frames consumed (inserts made, including the one that returned stop). */
int orc_otw_run(orc_otw *o, const double *live, int T) {
    int n = 0;
    for (int i = 0; i < T; i++) {
        n++;
        if (orc_otw_insert(o, live + (size_t)i * o->F) == ORC_STOP_REF_END) break;
    }
    return n;
}

int64_t orc_otw_path_len(const orc_otw *o) { return o->n_path; }
void orc_otw_copy_path(const orc_otw *o, int32_t *out) {
    memcpy(out, o->path, sizeof(int32_t) * 2 * (size_t)o->n_path);
}
/* state[0..6] = t, j, direction, previous, run_count, status, first_insert */
void orc_otw_state(const orc_otw *o, int32_t *state) {
    state[0] = o->t;
    state[1] = o->j;
    state[2] = o->direction;
    state[3] = o->previous;
    state[4] = o->run_count;
    state[5] = o->status;
    state[6] = o->first_insert;
}
/* counters[0..5] = cells, row strips, col strips, inserts, max in-strip carry run, sum of per-strip max */
void orc_otw_counters(const orc_otw *o, int64_t *counters) {
    counters[0] = o->cells;
    counters[1] = o->n_row_strips;
    counters[2] = o->n_col_strips;
    counters[3] = o->n_inserts;
    counters[4] = o->max_left_run;
    counters[5] = o->sum_strip_left_run;
}
/* The two live bands of the accumulated cost: row t over columns [j-c, j] and column j over rows
 * [t-c, t] (c+1 entries each, out-of-range entries = NaN).  Index i <-> offset i - c. */
void orc_otw_bands(const orc_otw *o, double *row_band, double *col_band) {
    int t = o->t < o->M ? o->t : o->M - 1;
    int j = o->j < o->N ? o->j : o->N - 1;
    for (int i = 0; i <= o->c; i++) {
        int y = j - o->c + i, x = t - o->c + i;
        row_band[i] = (y >= 0) ? o->acc[(size_t)t * o->N + y] : NAN;
        col_band[i] = (x >= 0) ? o->acc[(size_t)x * o->N + j] : NAN;
    }
}
const double *orc_otw_acc(const orc_otw *o) { return o->acc; }
const double *orc_otw_cost(const orc_otw *o) { return o->cost; }

/* ---------------------------------------------------------------- offline DTW (dtw.py:5-53) */

/* a: [M][F], b: [N][F].  cost/acc: [M][N]; back: [M][N] codes 0 left,1 up,2 diag; path pairs (i,j)
 * from (0,0) to (M-1,N-1); returns path length. */
int64_t orc_dtw(const double *a, const double *b, int M, int N, int F, double *cost, double *acc,
                int8_t *back, int32_t *path) {
    size_t n = (size_t)N;
    for (int i = 0; i < M; i++)
        for (int j = 0; j < N; j++)
            cost[(size_t)i * n + j] = 1.0 - orc_dot_chain(a + (size_t)i * F, b + (size_t)j * F, F);
    acc[0] = cost[0];
    back[0] = 2;
    for (int i = 1; i < M; i++) {
        acc[(size_t)i * n] = cost[(size_t)i * n] + acc[(size_t)(i - 1) * n];
        back[(size_t)i * n] = 1;
    }
    for (int j = 1; j < N; j++) {
        acc[j] = cost[j] + acc[j - 1];
        back[j] = 0;
    }
    for (int i = 1; i < M; i++) {
        for (int j = 1; j < N; j++) {
            double cst = cost[(size_t)i * n + j];
            double o0 = acc[(size_t)i * n + j - 1] + cst;
            double o1 = acc[(size_t)(i - 1) * n + j] + cst;
            double o2 = acc[(size_t)(i - 1) * n + j - 1] + 2 * cst;
            int s = 0;
            double best = o0; /* np.argmin: first minimum */
            if (o1 < best) {
                best = o1;
                s = 1;
            }
            if (o2 < best) {
                best = o2;
                s = 2;
            }
            acc[(size_t)i * n + j] = best;
            back[(size_t)i * n + j] = (int8_t)s;
        }
    }
    int i = M - 1, j = N - 1;
    int64_t len = 0;
    path[0] = i;
    path[1] = j;
    len = 1;
    while (i > 0 || j > 0) {
        int s = back[(size_t)i * n + j];
        if (s == 0)
            j -= 1;
        else if (s == 1)
            i -= 1;
        else {
            i -= 1;
            j -= 1;
        }
        path[2 * len] = i;
        path[2 * len + 1] = j;
        len++;
    }
    for (int64_t p = 0; p < len / 2; p++) { /* path.reverse() */
        int32_t x = path[2 * p], y = path[2 * p + 1];
        path[2 * p] = path[2 * (len - 1 - p)];
        path[2 * p + 1] = path[2 * (len - 1 - p) + 1];
        path[2 * (len - 1 - p)] = x;
        path[2 * (len - 1 - p) + 1] = y;
    }
    return len;
}

/* ---------------------------------------------------------------- WTW (wtw.py) */

/* wtw.py:162-171.  x: [n][F] live window, y: [m][F] ref window -> C [n][m]. */
void orc_wtw_cost_matrix(const double *x, const double *y, int n, int m, int F, double *C) {
    for (int i = 0; i < n; i++) {
        double nx = sqrt(orc_dot_chain(x + (size_t)i * F, x + (size_t)i * F, F));
        for (int j = 0; j < m; j++) {
            double ny = sqrt(orc_dot_chain(y + (size_t)j * F, y + (size_t)j * F, F));
            double dot = orc_dot_strided(x + (size_t)i * F, y + (size_t)j * F, F);
            C[(size_t)i * m + j] = 1.0 - dot / (nx * ny);
        }
    }
}

/* wtw.py:173-217: unit step weights; candidates (i-1,j),(i,j-1),(i-1,j-1), strict '<';
 * B codes 0 origin, 1 from (i,j-1), 2 diagonal, 3 from (i-1,j). */
void orc_wtw_run_dtw(const double *C, int n, int m, double *D, int8_t *B) {
    D[0] = C[0];
    B[0] = 0;
    double cost = C[0];
    for (int i = 1; i < n; i++) {
        cost += C[(size_t)i * m];
        D[(size_t)i * m] = cost;
        B[(size_t)i * m] = 3;
    }
    cost = C[0];
    for (int i = 1; i < m; i++) {
        cost += C[i];
        D[i] = cost;
        B[i] = 1;
    }
    for (int i = 1; i < n; i++) {
        for (int j = 1; j < m; j++) {
            double mc = D[(size_t)(i - 1) * m + j];
            int8_t p = 3;
            double v = D[(size_t)i * m + j - 1];
            if (v < mc) {
                mc = v;
                p = 1;
            }
            v = D[(size_t)(i - 1) * m + j - 1];
            if (v < mc) {
                mc = v;
                p = 2;
            }
            D[(size_t)i * m + j] = mc + C[(size_t)i * m + j];
            B[(size_t)i * m + j] = p;
        }
    }
}

/* wtw.py:219-240.  Returns sub-path length; pairs run from (0,0) to (n-1,m-1). */
int orc_wtw_find_path(const int8_t *B, int n, int m, int32_t *sub) {
    int i = n - 1, j = m - 1, len = 0;
    sub[0] = i;
    sub[1] = j;
    len = 1;
    while (!(i == 0 && j == 0)) {
        int8_t p = B[(size_t)i * m + j];
        if (p == 1)
            j -= 1;
        else if (p == 2) {
            i -= 1;
            j -= 1;
        } else if (p == 3)
            i -= 1;
        else
            break; /* cannot happen for a well-formed B */
        sub[2 * len] = i;
        sub[2 * len + 1] = j;
        len++;
    }
    for (int p = 0; p < len / 2; p++) {
        int32_t x = sub[2 * p], y = sub[2 * p + 1];
        sub[2 * p] = sub[2 * (len - 1 - p)];
        sub[2 * p + 1] = sub[2 * (len - 1 - p) + 1];
        sub[2 * (len - 1 - p)] = x;
        sub[2 * (len - 1 - p) + 1] = y;
    }
    return len;
}

typedef struct {
    int F, M, N;  /* features, ref frames, live capacity 2M (wtw.py:52-53 swaps the names) */
    int W, hopf;  /* dtw_win_size/hop_size, dtw_hop_size/hop_size in frames */
    double *ref;  /* [M][F] */
    double *live; /* [N][F] */
    int chroma_ptr, live_ptr, ref_ptr, status;
    int32_t *path;
    int64_t n_path, path_cap;
    int64_t n_windows, cells;
    double *C, *D;
    int8_t *B;
    int32_t *sub;
} orc_wtw;

orc_wtw *orc_wtw_create(const double *ref, int M, int F, int W, int hopf) {
    orc_wtw *w = (orc_wtw *)calloc(1, sizeof(orc_wtw));
    if (!w) return NULL;
    w->F = F;
    w->M = M;
    w->N = 2 * M;
    w->W = W;
    w->hopf = hopf;
    w->ref = (double *)malloc(sizeof(double) * (size_t)M * F);
    w->live = (double *)calloc((size_t)w->N * F, sizeof(double));
    w->path_cap = 4 * (int64_t)w->N + 16;
    w->path = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)w->path_cap);
    w->C = (double *)malloc(sizeof(double) * (size_t)W * W);
    w->D = (double *)malloc(sizeof(double) * (size_t)W * W);
    w->B = (int8_t *)malloc((size_t)W * W);
    w->sub = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(2 * W + 2));
    memcpy(w->ref, ref, sizeof(double) * (size_t)M * F);
    return w;
}

void orc_wtw_destroy(orc_wtw *w) {
    if (!w) return;
    free(w->ref);
    free(w->live);
    free(w->path);
    free(w->C);
    free(w->D);
    free(w->B);
    free(w->sub);
    free(w);
}

/* wtw.py:76-77: the check made once per insert() call, before any column is processed. */
int orc_wtw_insert_precheck(orc_wtw *w) {
    if (w->ref_ptr >= w->M - 1 || w->live_ptr >= w->N - 1) w->status = ORC_STOP_REF_END;
    return w->status;
}

/* wtw.py:92-128: one new live chroma column (already normalised by the caller). */
int orc_wtw_push_col(orc_wtw *w, const double *col) {
    if (w->chroma_ptr >= w->N) { /* the reference would raise IndexError here */
        w->status = ORC_LIVE_OVERFLOW;
        return w->status;
    }
    memcpy(w->live + (size_t)w->chroma_ptr * w->F, col, sizeof(double) * w->F);
    w->chroma_ptr += 1;
    if (w->ref_ptr >= (w->M - 1 - w->W) || w->live_ptr >= (w->N - 1 - w->W)) {
        w->status = ORC_STOP_REF_END;
        return w->status;
    }
    while (w->chroma_ptr - w->live_ptr >= w->W) {
        int n = w->W, m = w->W;
        if (w->ref_ptr + m > w->M) m = w->M - w->ref_ptr; /* numpy slice truncation */
        if (m <= 0) break;
        orc_wtw_cost_matrix(w->live + (size_t)w->live_ptr * w->F, w->ref + (size_t)w->ref_ptr * w->F,
                            n, m, w->F, w->C);
        orc_wtw_run_dtw(w->C, n, m, w->D, w->B);
        int len = orc_wtw_find_path(w->B, n, m, w->sub);
        w->n_windows++;
        w->cells += (int64_t)n * m;
        int change = 0, index = -1;
        for (int i = 0; i < len; i++) {
            int l = w->sub[2 * i], r = w->sub[2 * i + 1];
            if (l <= w->hopf) {
                if (w->n_path >= w->path_cap) {
                    w->path_cap *= 2;
                    w->path = (int32_t *)realloc(w->path, sizeof(int32_t) * 2 * (size_t)w->path_cap);
                }
                w->path[2 * w->n_path] = l + w->live_ptr;
                w->path[2 * w->n_path + 1] = r + w->ref_ptr;
                w->n_path++;
            } else {
                change = 1;
                index = i - 1;
                break;
            }
        }
        if (change) {
            int dl = w->sub[2 * index], dr = w->sub[2 * index + 1];
            w->live_ptr += dl;
            w->ref_ptr += dr;
        } else {
            w->live_ptr += w->hopf;
            w->ref_ptr += w->hopf;
        }
    }
    return w->status;
}

int64_t orc_wtw_path_len(const orc_wtw *w) { return w->n_path; }
void orc_wtw_copy_path(const orc_wtw *w, int32_t *out) {
    memcpy(out, w->path, sizeof(int32_t) * 2 * (size_t)w->n_path);
}
/* state[0..3] = chroma_ptr, live_ptr, ref_ptr, status */
void orc_wtw_state(const orc_wtw *w, int32_t *state) {
    state[0] = w->chroma_ptr;
    state[1] = w->live_ptr;
    state[2] = w->ref_ptr;
    state[3] = w->status;
}
void orc_wtw_counters(const orc_wtw *w, int64_t *counters) {
    counters[0] = w->n_windows;
    counters[1] = w->cells;
}
