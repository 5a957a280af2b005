"""numpy restatement of otw_eran.OnlineTimeWarping's insert loop -- TEST INFRASTRUCTURE ONLY.

This is the "numpy otw_eran.py CPU path" that BASELINE.md 3b / SURVEY 8(d) ask to be timed next to the
GPU result: the same dense (2N x N) float64 matrices, the same per-cell numpy calls as the reference
(``np.dot`` on two strided column views per cell, otw_eran.py:220; Python ``min`` over the candidate
list, :230-237; ``np.argmin`` over the two band slices, :199-206), so its speed is the reference's
(about 4.5 us per cell) and its results are the reference's bit for bit -- pinned against the golden
vectors the reference itself produced (tests/test_oracle_golden.py::test_numpy_restatement_*).
Only tests/ and bench.py's cpu_baseline leg import it; the product never does.

The control flow is written as an explicit state machine (ROW / COLUMN / BOTH codes, one ``_decide``)
instead of the reference's nested loop; per-cell arithmetic and its order are unchanged.
"""
import numpy as np

BOTH, ROW, COLUMN = 0, 1, 2
RUNNING, STOP_REF_END, LIVE_OVERFLOW = 0, 1, 2


class NumpyOTW(object):
    def __init__(self, ref, c, max_run_count):
        self.ref = np.asarray(ref, dtype=np.float64)           # (F, N), feature-major like otw_eran.py:17
        n_feat, n_ref = self.ref.shape
        self.c = int(c)
        self.max_run_count = int(max_run_count)
        self.cap = 2 * n_ref                                   # otw_eran.py:14
        self.live = np.full((n_feat, self.cap), -1.0)          # :20
        self.cost = np.full((self.cap, n_ref), -1.0)           # :23
        self.acc = np.full((self.cap, n_ref), 1e10)            # :27
        self.t = self.j = 0
        self.previous = None
        self.direction = BOTH
        self.run_count = 1
        self.path = []
        self.status = RUNNING
        self._fresh = True
        self.cells = 0

    # ---- one cell (otw_eran.py:214-240)
    def _cell(self, x, y):
        d = 1 - np.dot(self.live[:, x], self.ref[:, y])
        self.cost[x, y] = d
        self.cells += 1
        if x == 0 and y == 0:
            self.acc[0, 0] = d
            return
        cands = []
        if y > 0:
            cands.append(self.acc[x, y - 1] + d)
        if x > 0:
            cands.append(self.acc[x - 1, y] + d)
        if x > 0 and y > 0:
            cands.append(self.acc[x - 1, y - 1] + 2 * d)
        self.acc[x, y] = min(cands)

    def _row_strip(self):
        for k in range(max(0, self.j - self.c + 1), self.j + 1):
            self._cell(self.t, k)

    def _col_strip(self):
        for k in range(max(0, self.t - self.c + 1), self.t + 1):
            self._cell(k, self.j)

    # ---- best point + next direction (otw_eran.py:158-211)
    def _decide(self):
        j_lo = max(0, self.j - self.c + 1)
        t_lo = max(0, self.t - self.c + 1)
        bj = j_lo + int(np.argmin(self.acc[self.t, j_lo:self.j + 1]))
        bt = t_lo + int(np.argmin(self.acc[t_lo:self.t + 1, self.j]))
        if self.acc[self.t, bj] < self.acc[bt, self.j]:
            x, y = self.t, bj
        else:
            x, y = bt, self.j
        self.path.append((x, y))
        if self.t < self.c:
            d = BOTH
        elif self.run_count >= self.max_run_count:
            d = COLUMN if self.previous == ROW else ROW
        elif x < self.t:
            d = COLUMN
        elif y < self.j:
            d = ROW
        else:
            d = BOTH
        self.run_count = self.run_count + 1 if d == self.previous else 1
        if d != BOTH:
            self.previous = d
        self.direction = d

    # ---- insert(live_sample) (otw_eran.py:38-85); returns the status code
    def insert(self, frame):
        if self._fresh:
            self._fresh = False
            self.live[:, 0] = frame
            self._cell(0, 0)
            return RUNNING
        self.t += 1
        if self.t >= self.cap:
            self.status = LIVE_OVERFLOW
            return RUNNING  # the reference prints and returns None
        self.live[:, self.t] = frame
        self._row_strip()
        while True:
            if self.direction != ROW:
                self.j += 1
                if self.j >= self.ref.shape[1]:
                    self.status = STOP_REF_END
                    return STOP_REF_END
                self._col_strip()
            self._decide()
            if self.direction != COLUMN:
                return RUNNING

    def run(self, live):
        """``for i: if insert(live[:, i]) == "stop": break``; returns the frames consumed."""
        n = 0
        for i in range(live.shape[1]):
            n += 1
            if self.insert(live[:, i]) == STOP_REF_END:
                break
        return n

    def bands(self):
        """(acc[t, j-c..j], acc[t-c..t, j]) with NaN at negative indices, like rts_otw_read_bands."""
        c, t, j = self.c, min(self.t, self.cap - 1), min(self.j, self.ref.shape[1] - 1)
        rb = np.full(c + 1, np.nan)
        cb = np.full(c + 1, np.nan)
        for i in range(c + 1):
            y, x = j - c + i, t - c + i
            if y >= 0:
                rb[i] = self.acc[t, y]
            if x >= 0:
                cb[i] = self.acc[x, j]
        return rb, cb
