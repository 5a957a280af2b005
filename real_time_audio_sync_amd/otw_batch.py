"""Batched online time warping on one MI355X: B independent live streams against one reference.

This is the host-side object behind the drop-in classes (otw_eran.OnlineTimeWarping,
livenote.LiveNote, livenote_v2.LiveNoteV2) and behind bench.py.  torch supplies device memory and
the stream; all computation happens in librtsync.so's HIP kernels (csrc/otw.hip)."""
import ctypes

import numpy as np
import torch

from . import _native as nat

_VARIANTS = {"otw": nat.VARIANT_OTW, "livenote": nat.VARIANT_LIVENOTE, "livenote_v2": nat.VARIANT_LIVENOTE_V2}


def _np_dtype_code(dt):
    if dt == torch.float32:
        return nat.F32
    if dt == torch.float64:
        return nat.F64
    raise TypeError("feature tensors must be float32 or float64, got %s" % dt)


def frames_tensor(x, device, dtype=None):
    """Reference layout (12, n) feature-major (numpy or torch) -> device tensor [n][12]."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        x = x.to(dtype)
    return x.to(device).t().contiguous()


_on_device = nat.on_device


class BatchedOTW:
    """``ref``: (12, N) feature-major array/tensor, or a device tensor already [N][12] with
    ``frame_major=True``.  ``variant``: 'otw' | 'livenote' | 'livenote_v2'."""

    def __init__(self, ref, c, max_run_count, batch=1, variant="otw", euclid=False, device="cuda:0",
                 dtype=None, frame_major=False, waves=None):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedOTW needs a ROCm GPU (no CPU fallback)")
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        if frame_major:
            self.ref = ref.to(self.device).contiguous()
        else:
            self.ref = frames_tensor(ref, self.device, dtype)
        self.N, F = self.ref.shape
        self.B, self.c = int(batch), int(c)
        self.variant = variant
        h = ctypes.c_void_p()
        nat.check(nat.lib.rts_otw_create(self.ref.data_ptr(), _np_dtype_code(self.ref.dtype), F, self.N, self.B,
                                         self.c, int(max_run_count), _VARIANTS[variant],
                                         nat.COST_EUCLID if euclid else nat.COST_DOT, ctypes.byref(h)))
        self._h = h
        if waves is not None:
            nat.check(nat.lib.rts_otw_set_waves(self._h, int(waves)))
        self._keep = None
        self._version = 0  # bumped by everything that changes what the handle has consumed

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            nat.destroy_on(self.device, nat.lib.rts_otw_destroy, h)

    __del__ = close

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- whole sequences -------------------------------------------------------------------
    def pack(self, lives, dtype=None):
        """List of (12, T_b) arrays -> (device [B][T_max][12], device int32 [B])."""
        assert len(lives) == self.B
        dtype = dtype or self.ref.dtype
        tmax = max(int(l.shape[1]) for l in lives)
        buf = torch.zeros((self.B, tmax, 12), dtype=dtype)
        for b, l in enumerate(lives):
            l = torch.from_numpy(np.ascontiguousarray(l)) if isinstance(l, np.ndarray) else l
            buf[b, : l.shape[1]] = l.t().to(dtype)
        lens = torch.tensor([int(l.shape[1]) for l in lives], dtype=torch.int32)
        return buf.to(self.device), lens.to(self.device)

    @_on_device
    def run(self, live_dev, live_len_dev, mode="insert"):
        """Asynchronous on the current stream.  live_dev: [B][T_max][12]; live_len_dev: int32 [B]."""
        assert live_dev.is_contiguous() and live_dev.shape[0] == self.B and live_dev.shape[2] == 12
        self._keep = (live_dev, live_len_dev)
        self._version += 1
        nat.check(nat.lib.rts_otw_run(self._h, live_dev.data_ptr(), _np_dtype_code(live_dev.dtype),
                                      int(live_dev.shape[1]), live_len_dev.data_ptr(),
                                      nat.MODE_SET_LIVE if mode == "set_live" else nat.MODE_INSERT_LOOP,
                                      self._stream()))

    @_on_device
    def insert(self, frames_dev, active_dev=None):
        """One frame per stream: frames_dev [B][12]; active_dev optional uint8 [B]."""
        assert frames_dev.is_contiguous() and tuple(frames_dev.shape) == (self.B, 12)
        self._version += 1
        nat.check(nat.lib.rts_otw_insert(self._h, frames_dev.data_ptr(), _np_dtype_code(frames_dev.dtype),
                                         active_dev.data_ptr() if active_dev is not None else None,
                                         self._stream()))

    @_on_device
    def push(self, frames_dev, n_new_dev=None):
        """Several frames per stream: frames_dev [B][n_max][12]; n_new_dev optional int32 [B]."""
        assert frames_dev.is_contiguous() and frames_dev.shape[0] == self.B and frames_dev.shape[2] == 12
        self._version += 1
        nat.check(nat.lib.rts_otw_push(self._h, frames_dev.data_ptr(), _np_dtype_code(frames_dev.dtype),
                                       int(frames_dev.shape[1]), n_new_dev.data_ptr() if n_new_dev is not None else None,
                                       self._stream()))

    @_on_device
    def reset(self):
        self._version += 1
        self._keep = None
        nat.check(nat.lib.rts_otw_reset(self._h, self._stream()))

    # ---- results ------------------------------------------------------------------------------
    @_on_device
    def states(self):
        out = np.zeros((self.B, nat.STATE_LEN), dtype=np.int32)
        nat.check(nat.lib.rts_otw_read_states(self._h, out.ctypes.data, self._stream()))
        return out

    def state(self, b=0):
        s = self.states()[b]
        cells = (int(np.uint32(s[nat.ST_CELLS_HI])) << 32) | int(np.uint32(s[nat.ST_CELLS_LO]))
        return dict(t=int(s[nat.ST_T]), j=int(s[nat.ST_J]), direction=int(s[nat.ST_DIRECTION]),
                    previous=int(s[nat.ST_PREVIOUS]), run_count=int(s[nat.ST_RUN_COUNT]),
                    status=int(s[nat.ST_STATUS]), first_insert=int(s[nat.ST_FIRST_INSERT]),
                    n_path=int(s[nat.ST_N_PATH]), consumed=int(s[nat.ST_CONSUMED]),
                    row_strips=int(s[nat.ST_ROW_STRIPS]), col_strips=int(s[nat.ST_COL_STRIPS]), cells=cells,
                    path_truncated=int(s[nat.ST_PATH_TRUNCATED]), band_recomputes=int(s[nat.ST_BAND_RECOMPUTES]))

    @_on_device
    def path(self, b=0):
        n = ctypes.c_int(0)
        nat.check(nat.lib.rts_otw_read_path(self._h, b, None, 0, ctypes.byref(n), self._stream()))
        out = np.empty((n.value, 2), dtype=np.int32)
        if n.value:
            nat.check(nat.lib.rts_otw_read_path(self._h, b, out.ctypes.data, n.value, ctypes.byref(n),
                                                self._stream()))
        return out

    def paths(self):
        return [self.path(b) for b in range(self.B)]

    @_on_device
    def bands(self, b=0):
        rb = np.empty(self.c + 1)
        cb = np.empty(self.c + 1)
        nat.check(nat.lib.rts_otw_read_bands(self._h, b, rb.ctypes.data, cb.ctypes.data, self._stream()))
        return rb, cb

    @_on_device
    def enable_dense(self):
        """Allocate and attach the reference's dense (2N x N) acc_cost / cost matrices per stream
        (float64 device tensors [B][2N][N]); every evaluated cell is mirrored into them."""
        self.dense_acc = torch.empty((self.B, 2 * self.N, self.N), dtype=torch.float64, device=self.device)
        self.dense_cost = torch.empty((self.B, 2 * self.N, self.N), dtype=torch.float64, device=self.device)
        nat.check(nat.lib.rts_otw_set_dense(self._h, self.dense_acc.data_ptr(), self.dense_cost.data_ptr(),
                                            self._stream()))
        return self.dense_acc, self.dense_cost

    @_on_device
    def replay_dense(self):
        """The reference's dense (2N x N) acc_cost / cost matrices (float64 device tensors [B][2N][N]) for
        everything consumed since the last reset, recomputed on demand by a second pass over the kept frames
        (rts_otw_replay_dense); the tracker itself never pays for them."""
        acc = torch.empty((self.B, 2 * self.N, self.N), dtype=torch.float64, device=self.device)
        cost = torch.empty((self.B, 2 * self.N, self.N), dtype=torch.float64, device=self.device)
        if self._keep is not None:   # the frames of the last run(): handed in again, the library keeps no pointer to them
            lv, ln = self._keep
            nat.check(nat.lib.rts_otw_replay_dense(self._h, lv.data_ptr(), _np_dtype_code(lv.dtype), int(lv.shape[1]),
                                                   ln.data_ptr(), acc.data_ptr(), cost.data_ptr(), self._stream()))
        else:                        # frames that came through insert() / push(): the handle's own history
            nat.check(nat.lib.rts_otw_replay_dense(self._h, None, nat.F64, 0, None, acc.data_ptr(), cost.data_ptr(),
                                                   self._stream()))
        return acc, cost

    @_on_device
    def set_waves(self, waves):
        nat.check(nat.lib.rts_otw_set_waves(self._h, int(waves)))

    @property
    def kernel_name(self):
        return nat.lib.rts_otw_kernel_name(self._h).decode()
