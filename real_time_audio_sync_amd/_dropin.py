"""Shared machinery of the drop-in classes OnlineTimeWarping / LiveNote / LiveNoteV2.

Each instance is a batch-of-one BatchedOTW with float64 features (the reference computes in
float64, otw_eran.py:20-27).  insert() launches the HIP kernel for one frame and reads the
64-byte state word back, because the reference's insert() is synchronous and returns "stop".
The dense acc_cost / cost matrices of the reference are recomputed only when they are read."""
import numpy as np
import torch

from . import _native as nat
from .otw_batch import BatchedOTW

_DIR_NAMES_CAP = {nat.DIR_NONE: None, nat.DIR_BOTH: "Both", nat.DIR_ROW: "Row", nat.DIR_COLUMN: "Column"}
_DIR_NAMES_LOW = {nat.DIR_NONE: None, nat.DIR_BOTH: "both", nat.DIR_ROW: "row", nat.DIR_COLUMN: "column"}


class OtwDropIn(object):
    _variant = "otw"
    _names = _DIR_NAMES_CAP
    _msg_overflow = "Done. Ran out of room in pre-allocated live-sequence"
    _msg_stop = "Done. Ran out of ref-sequence"
    DENSE_LIMIT_BYTES = 2 << 30  # keep acc_cost / cost (2 x 2N x N float64) only below 2 GiB

    def _setup(self, ref, band, max_run_count, euclid=False, device="cuda:0"):
        ref = np.asarray(ref, dtype=np.float64)
        if ref.ndim != 2:
            raise ValueError("ref must be (n_features, n_frames)")
        self._ref_host = ref
        self._eng = BatchedOTW(ref, band, max_run_count, batch=1, variant=self._variant, euclid=euclid,
                               device=device, dtype=torch.float64)
        self._dev = self._eng.device
        # the reference's dense (2N x N) matrices are produced lazily, when .acc_cost / .cost is read (a replay of
        # the kept live history): insert() / set_live() run on the pipelined kernel and never pay for them
        self._dense = 2 * (2 * ref.shape[1]) * ref.shape[1] * 8 <= self.DENSE_LIMIT_BYTES
        self._frame = torch.empty((1, ref.shape[0]), dtype=torch.float64, device=self._dev)
        self._path_is_array = False
        self._reported = nat.RUNNING

    # ---- reference API ------------------------------------------------------------------------
    def insert(self, live_sample):
        """One live chroma column (otw_eran.py:38-85 / livenote_v2.py:43-104).  Returns None, or
        "stop" once the reference sequence is exhausted (and on every later call; the reference
        itself would raise IndexError there)."""
        col = np.ascontiguousarray(np.asarray(live_sample, dtype=np.float64).reshape(1, -1))
        self._frame.copy_(torch.from_numpy(col))
        self._eng.insert(self._frame)
        st = int(self._eng.states()[0, nat.ST_STATUS])
        if st == nat.STOP_REF_END:
            if self._reported != st:
                print(self._msg_stop)
            self._reported = st
            return "stop"
        if st == nat.LIVE_OVERFLOW:
            print(self._msg_overflow)
        self._reported = st
        return None

    def set_live(self, live):
        """Whole live sequence at once (otw_eran.py:91-142 / livenote_v2.py:108-155)."""
        live = np.asarray(live, dtype=np.float64)
        lv, ln = self._eng.pack([live], dtype=torch.float64)
        self._eng.run(lv, ln, mode="set_live")
        self._path_is_array = self._variant == "otw"  # otw_eran.py:142

    # ---- attributes the harnesses read ----------------------------------------------------------
    @property
    def path(self):
        p = self._eng.path(0)
        if self._path_is_array:
            return p.astype(np.int64)
        return [(int(x), int(y)) for x, y in p]

    def _st(self):
        return self._eng.state(0)

    @property
    def run_count(self):
        return self._st()["run_count"]

    @property
    def previous(self):
        return self._names[self._st()["previous"]]

    @property
    def direction(self):
        return self._names[self._st()["direction"]]

    @property
    def first_insert(self):
        return bool(self._st()["first_insert"])

    def bands(self):
        """The live part of acc_cost: (acc_cost[t, j-c:j+1], acc_cost[t-c:t+1, j]), NaN-padded at
        negative indices.  The dense (2N x N) matrices of the reference are never materialised."""
        return self._eng.bands(0)

    def _dense_matrix(self, which):
        if not self._dense:
            raise NotImplementedError(
                "the dense (2N x N) %s of the reference is not kept for a reference this long (it would need "
                "%.1f GiB); use .bands() for the two live bands the algorithm actually reads"
                % (which, 2 * (2 * self._ref_host.shape[1]) * self._ref_host.shape[1] * 8 / 2.0 ** 30))
        if getattr(self, "_dense_cache", (None,))[0] != self._eng._version:
            acc, cost = self._eng.replay_dense()
            self._dense_cache = (self._eng._version, acc[0].cpu().numpy(), cost[0].cpu().numpy())
        return self._dense_cache[1 if which == "acc_cost" else 2]

    @property
    def acc_cost(self):
        """(2N, N) float64, sentinel (1e10 / inf) where never evaluated -- otw_eran.py:27, livenote_v2.py:22-23."""
        return self._dense_matrix("acc_cost")

    @property
    def cost(self):
        """(2N, N) float64, -1 where never evaluated -- otw_eran.py:23."""
        return self._dense_matrix("cost")
