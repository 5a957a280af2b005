"""ctypes binding of librtsync.so (include/rtsync.h).  No fallback: if the HIP library is missing
or fails to load, importing this module raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTSYNC_LIB selects another build of the same library (A/B measurements of kernel variants in one process tree)
SO_PATH = os.environ.get("RTSYNC_LIB") or os.path.join(_HERE, "librtsync.so")

# constants mirrored from include/rtsync.h
F32, F64, I16 = 0, 1, 2
VARIANT_OTW, VARIANT_LIVENOTE, VARIANT_LIVENOTE_V2 = 0, 1, 2
COST_DOT, COST_EUCLID = 0, 1
DIR_NONE, DIR_BOTH, DIR_ROW, DIR_COLUMN = -1, 0, 1, 2
RUNNING, STOP_REF_END, LIVE_OVERFLOW, DEVICE_FAULT = 0, 1, 2, 3
MODE_INSERT_LOOP, MODE_SET_LIVE = 0, 1
STATE_LEN = 16
(ST_T, ST_J, ST_DIRECTION, ST_PREVIOUS, ST_RUN_COUNT, ST_STATUS, ST_FIRST_INSERT, ST_N_PATH, ST_CONSUMED,
 ST_ROW_STRIPS, ST_COL_STRIPS, ST_CELLS_LO, ST_CELLS_HI, ST_PATH_TRUNCATED) = range(14)
ST_BAND_RECOMPUTES = 15

# every symbol include/rtsync.h declares (tests/test_abi.py checks the library exports them all)
EXPORTS = {}


class RtsyncError(RuntimeError):
    pass


def _load():
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "librtsync.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python real_time_audio_sync_amd/_build.py`. There is no CPU fallback." % SO_PATH)
    return ctypes.CDLL(SO_PATH)


lib = _load()

_vp, _i32, _pi32 = ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)


def _decl(name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    EXPORTS[name] = fn
    return fn


_decl("rts_last_error", ctypes.c_char_p, [])
_decl("rts_version", _i32, [])
_decl("rts_device_count", _i32, [])
_decl("rts_otw_create", _i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, ctypes.POINTER(_vp)])
_decl("rts_otw_destroy", _i32, [_vp])
_decl("rts_otw_reset", _i32, [_vp, _vp])
_decl("rts_otw_run", _i32, [_vp, _vp, _i32, _i32, _vp, _i32, _vp])
_decl("rts_otw_insert", _i32, [_vp, _vp, _i32, _vp, _vp])
_decl("rts_otw_push", _i32, [_vp, _vp, _i32, _i32, _vp, _vp])
_decl("rts_otw_read_state", _i32, [_vp, _i32, _vp, _vp])
_decl("rts_otw_read_states", _i32, [_vp, _vp, _vp])
_decl("rts_otw_read_path", _i32, [_vp, _i32, _vp, _i32, _pi32, _vp])
_decl("rts_otw_read_bands", _i32, [_vp, _i32, _vp, _vp, _vp])
_decl("rts_otw_device_views", _i32, [_vp, ctypes.POINTER(_vp), _pi32, ctypes.POINTER(_vp)])
_decl("rts_otw_set_waves", _i32, [_vp, _i32])
_decl("rts_otw_set_dense", _i32, [_vp, _vp, _vp, _vp])
_decl("rts_otw_replay_dense", _i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp])
_decl("rts_otw_kernel_name", ctypes.c_char_p, [_vp])
_i64 = ctypes.c_longlong
_decl("rts_dtw_workspace_bytes", _i32, [_i32, _i32, _i32, ctypes.POINTER(ctypes.c_size_t)])
_decl("rts_dtw", _i32, [_vp, _i32, _i64, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                        ctypes.c_size_t, _vp])


_decl("rts_chroma_num_frames", _i64, [_i64, _i32, _i32, _i32])
_decl("rts_chroma_create", _i32, [_i32, _i32, _vp, _vp, ctypes.POINTER(_vp)])
_decl("rts_chroma_destroy", _i32, [_vp])
_decl("rts_chroma_frames", _i32, [_vp, _vp, _i32, _i64, _i32, _i32, _i32, _vp, _i32, _vp, _vp])
_decl("rts_chroma_frames_batch", _i32, [_vp, _vp, _i32, _i64, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp])
_decl("rts_chroma_plan_info", _i32, [_vp, _pi32, _pi32])
_decl("rts_chroma_project", _i32, [_vp, _vp, _i32, _i32, _vp, _i32, _vp])
_decl("rts_chroma_diff", _i32, [_vp, _i32, _i32, _vp, _vp])


_decl("rts_wtw_create", _i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, ctypes.POINTER(_vp)])
_decl("rts_wtw_destroy", _i32, [_vp])
_decl("rts_wtw_reset", _i32, [_vp, _vp])
_decl("rts_wtw_push", _i32, [_vp, _vp, _i32, _i32, _vp, _i32, _vp])
_decl("rts_wtw_read_states", _i32, [_vp, _vp, _vp])
_decl("rts_wtw_read_path", _i32, [_vp, _i32, _vp, _i32, _pi32, _vp])
_decl("rts_wtw_read_last_d", _i32, [_vp, _i32, _vp, _vp])
_decl("rts_wtw_device_views", _i32, [_vp, ctypes.POINTER(_vp), _pi32, ctypes.POINTER(_vp)])
_decl("rts_wtw_state_view", _i32, [_vp, ctypes.POINTER(_vp)])
WTW_STATE_LEN = 8

_decl("rts_live_create", _i32, [_vp, _vp, _vp, _i32, _i32, ctypes.POINTER(_vp)])
_decl("rts_live_destroy", _i32, [_vp])
_decl("rts_live_reset", _i32, [_vp, _vp])
_decl("rts_live_staging", _i32, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_i64)])
_decl("rts_live_submit", _i32, [_vp, _i32, _vp])
_decl("rts_live_feed", _i32, [_vp, _vp, _i32, _vp, _vp])
_decl("rts_live_poll", _i32, [_vp, _vp, _vp, _pi32, _pi32])
_decl("rts_live_pending", _i32, [_vp, _vp])


def check(rc):
    if rc != 0:
        raise RtsyncError("rtsync error %d: %s" % (rc, lib.rts_last_error().decode("utf-8", "replace")))
    return rc


def destroy_on(device, destroy_fn, handle):
    """Free a handle's device buffers with the handle's own device current (hipFree of a foreign device's pointer
    from another current device is an error the destroy functions cannot report)."""
    try:
        import torch
        if device.index is None or torch.cuda.current_device() == device.index:
            destroy_fn(handle)
        else:
            with torch.cuda.device(device):
                destroy_fn(handle)
    except Exception:   # interpreter shutdown: torch may already be gone; the process is ending anyway
        pass


def on_device(fn):
    """Decorator for methods of an object with a ``device`` attribute (a torch.device with an index): run the method
    with that device current.  A handle belongs to the device it was created on; with several devices driven from
    one process (shard.ShardedOTW) another one may be current when the call comes."""
    import functools

    import torch

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        if torch.cuda.current_device() == self.device.index:
            return fn(self, *args, **kwargs)
        with torch.cuda.device(self.device):
            return fn(self, *args, **kwargs)
    return wrapped
