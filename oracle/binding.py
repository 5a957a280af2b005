"""ctypes binding of oracle/liboracle.so (test infrastructure only -- see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

OTW, LIVENOTE, LIVENOTE_V2 = 0, 1, 2
COST_DOT, COST_EUCLID = 0, 1
DIR_NONE, DIR_BOTH, DIR_ROW, DIR_COLUMN = -1, 0, 1, 2
RUNNING, STOP_REF_END, LIVE_OVERFLOW = 0, 1, 2

_lib = None


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "rtsync_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = ctypes.CDLL(_SO)
    vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
    L.orc_dot_strided.restype = dbl
    L.orc_dot_strided.argtypes = [vp, vp, i32]
    L.orc_dot_chain.restype = dbl
    L.orc_dot_chain.argtypes = [vp, vp, i32]
    L.orc_euclid.restype = dbl
    L.orc_euclid.argtypes = [vp, vp, i32]
    L.orc_otw_create.restype = vp
    L.orc_otw_create.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32]
    L.orc_otw_destroy.argtypes = [vp]
    L.orc_otw_insert.restype = i32
    L.orc_otw_insert.argtypes = [vp, vp]
    L.orc_otw_set_live.restype = i32
    L.orc_otw_set_live.argtypes = [vp, vp, i32]
    L.orc_otw_run.restype = i32
    L.orc_otw_run.argtypes = [vp, vp, i32]
    L.orc_otw_path_len.restype = i64
    L.orc_otw_path_len.argtypes = [vp]
    L.orc_otw_copy_path.argtypes = [vp, vp]
    L.orc_otw_state.argtypes = [vp, vp]
    L.orc_otw_counters.argtypes = [vp, vp]
    L.orc_otw_bands.argtypes = [vp, vp, vp]
    L.orc_otw_acc.restype = vp
    L.orc_otw_acc.argtypes = [vp]
    L.orc_otw_cost.restype = vp
    L.orc_otw_cost.argtypes = [vp]
    L.orc_dtw.restype = i64
    L.orc_dtw.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.orc_wtw_cost_matrix.argtypes = [vp, vp, i32, i32, i32, vp]
    L.orc_wtw_run_dtw.argtypes = [vp, i32, i32, vp, vp]
    L.orc_wtw_find_path.restype = i32
    L.orc_wtw_find_path.argtypes = [vp, i32, i32, vp]
    L.orc_wtw_create.restype = vp
    L.orc_wtw_create.argtypes = [vp, i32, i32, i32, i32]
    L.orc_wtw_destroy.argtypes = [vp]
    L.orc_wtw_insert_precheck.restype = i32
    L.orc_wtw_insert_precheck.argtypes = [vp]
    L.orc_wtw_push_col.restype = i32
    L.orc_wtw_push_col.argtypes = [vp, vp]
    L.orc_wtw_path_len.restype = i64
    L.orc_wtw_path_len.argtypes = [vp]
    L.orc_wtw_copy_path.argtypes = [vp, vp]
    L.orc_wtw_state.argtypes = [vp, vp]
    L.orc_wtw_counters.argtypes = [vp, vp]
    _lib = L
    return L


def _frames(a):
    """Reference layout (F, n) feature-major -> C-level [n][F] float64 contiguous."""
    a = np.asarray(a, dtype=np.float64)
    return np.ascontiguousarray(a.T)


def _vec(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def dot_strided(x, y):
    x, y = _vec(x), _vec(y)
    return lib().orc_dot_strided(x.ctypes.data, y.ctypes.data, x.size)


def dot_chain(x, y):
    x, y = _vec(x), _vec(y)
    return lib().orc_dot_chain(x.ctypes.data, y.ctypes.data, x.size)


def euclid(x, y):
    x, y = _vec(x), _vec(y)
    return lib().orc_euclid(x.ctypes.data, y.ctypes.data, x.size)


class OtwOracle:
    """Restatement of OnlineTimeWarping / LiveNote / LiveNoteV2 (dense, float64).

    ``ref`` is (F, N) feature-major like the reference's constructors (otw_eran.py:6-17)."""

    def __init__(self, ref, c, max_run_count, variant=OTW, cost=COST_DOT, keep_cost=False):
        self._L = lib()
        r = _frames(ref)
        self.N, self.F = r.shape
        self.c = int(c)
        self.variant = variant
        self._h = self._L.orc_otw_create(r.ctypes.data, self.N, self.F, int(c), int(max_run_count),
                                         int(variant), int(cost), int(bool(keep_cost)))
        if not self._h:
            raise MemoryError("orc_otw_create failed")
        self._keep_cost = keep_cost

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.orc_otw_destroy(h)

    def insert(self, frame):
        f = _vec(frame)
        assert f.size == self.F
        return self._L.orc_otw_insert(self._h, f.ctypes.data)

    def run(self, live):
        """Insert loop over live (F, T); returns frames consumed."""
        lv = _frames(live)
        return self._L.orc_otw_run(self._h, lv.ctypes.data, lv.shape[0])

    def set_live(self, live):
        lv = _frames(live)
        return self._L.orc_otw_set_live(self._h, lv.ctypes.data, lv.shape[0])

    @property
    def path(self):
        n = self._L.orc_otw_path_len(self._h)
        out = np.empty((n, 2), dtype=np.int32)
        if n:
            self._L.orc_otw_copy_path(self._h, out.ctypes.data)
        return out

    @property
    def state(self):
        s = np.zeros(7, dtype=np.int32)
        self._L.orc_otw_state(self._h, s.ctypes.data)
        return dict(t=int(s[0]), j=int(s[1]), direction=int(s[2]), previous=int(s[3]),
                    run_count=int(s[4]), status=int(s[5]), first_insert=int(s[6]))

    @property
    def counters(self):
        s = np.zeros(6, dtype=np.int64)
        self._L.orc_otw_counters(self._h, s.ctypes.data)
        return dict(cells=int(s[0]), row_strips=int(s[1]), col_strips=int(s[2]), inserts=int(s[3]),
                    max_carry_run=int(s[4]), sum_strip_carry_run=int(s[5]))

    def bands(self):
        """(row band acc[t, j-c..j], column band acc[t-c..t, j]); NaN where the index is < 0."""
        rb = np.empty(self.c + 1)
        cb = np.empty(self.c + 1)
        self._L.orc_otw_bands(self._h, rb.ctypes.data, cb.ctypes.data)
        return rb, cb

    def acc_cost(self):
        p = self._L.orc_otw_acc(self._h)
        buf = (ctypes.c_double * (2 * self.N * self.N)).from_address(p)
        return np.frombuffer(buf, dtype=np.float64).reshape(2 * self.N, self.N).copy()

    def cost(self):
        p = self._L.orc_otw_cost(self._h)
        if not p:
            raise ValueError("created without keep_cost")
        buf = (ctypes.c_double * (2 * self.N * self.N)).from_address(p)
        return np.frombuffer(buf, dtype=np.float64).reshape(2 * self.N, self.N).copy()


def dtw(seq_a, seq_b):
    """Restatement of dtw.DTW(seq_a, seq_b) -> (cost, acc_cost, path), plus back-pointers."""
    a, b = _frames(seq_a), _frames(seq_b)
    M, F = a.shape
    N = b.shape[0]
    cost = np.empty((M, N))
    acc = np.empty((M, N))
    back = np.empty((M, N), dtype=np.int8)
    path = np.empty((M + N, 2), dtype=np.int32)
    n = lib().orc_dtw(a.ctypes.data, b.ctypes.data, M, N, F, cost.ctypes.data, acc.ctypes.data,
                      back.ctypes.data, path.ctypes.data)
    return cost, acc, path[:n].copy(), back


def wtw_cost_matrix(x, y):
    xf, yf = _frames(x), _frames(y)
    n, F = xf.shape
    m = yf.shape[0]
    C = np.empty((n, m))
    lib().orc_wtw_cost_matrix(xf.ctypes.data, yf.ctypes.data, n, m, F, C.ctypes.data)
    return C


def wtw_run_dtw(C):
    C = np.ascontiguousarray(C, dtype=np.float64)
    n, m = C.shape
    D = np.empty((n, m))
    B = np.empty((n, m), dtype=np.int8)
    lib().orc_wtw_run_dtw(C.ctypes.data, n, m, D.ctypes.data, B.ctypes.data)
    return D, B


def wtw_find_path(B):
    B = np.ascontiguousarray(B, dtype=np.int8)
    n, m = B.shape
    sub = np.empty((n + m + 2, 2), dtype=np.int32)
    k = lib().orc_wtw_find_path(B.ctypes.data, n, m, sub.ctypes.data)
    return sub[:k].copy()


class WtwOracle:
    """Restatement of wtw.WTW from the chroma level down (wtw.py:71-128).

    Raw-sample buffering and per-hop chroma (wtw.py:73, :81-93) are host logic restated in
    ``oracle.chroma_oracle.WtwAudioOracle``; this class takes finished chroma columns."""

    def __init__(self, chroma_ref, win_frames, hop_frames):
        self._L = lib()
        r = _frames(chroma_ref)
        self.M, self.F = r.shape
        if hop_frames < 1:
            raise ValueError("dtw_hop_size/hop_size must be >= 1 (the reference loops forever)")
        self._h = self._L.orc_wtw_create(r.ctypes.data, self.M, self.F, int(win_frames), int(hop_frames))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.orc_wtw_destroy(h)

    def insert_precheck(self):
        return self._L.orc_wtw_insert_precheck(self._h)

    def push_col(self, col):
        c = _vec(col)
        return self._L.orc_wtw_push_col(self._h, c.ctypes.data)

    @property
    def path(self):
        n = self._L.orc_wtw_path_len(self._h)
        out = np.empty((n, 2), dtype=np.int32)
        if n:
            self._L.orc_wtw_copy_path(self._h, out.ctypes.data)
        return out

    @property
    def state(self):
        s = np.zeros(4, dtype=np.int32)
        self._L.orc_wtw_state(self._h, s.ctypes.data)
        return dict(chroma_ptr=int(s[0]), live_ptr=int(s[1]), ref_ptr=int(s[2]), status=int(s[3]))

    @property
    def counters(self):
        s = np.zeros(2, dtype=np.int64)
        self._L.orc_wtw_counters(self._h, s.ctypes.data)
        return dict(windows=int(s[0]), cells=int(s[1]))
