"""Drop-in for the reference's livenote.py: ``LiveNote(ref, params, debug_params)``
(reference: livenote.py:3-226)."""
from ._dropin import OtwDropIn, _DIR_NAMES_LOW


class LiveNote(OtwDropIn):
    _variant = "livenote"
    _names = _DIR_NAMES_LOW
    _msg_overflow = "done - oob live"
    _msg_stop = "done - oob ref"

    def __init__(self, ref, params, debug_params, device="cuda:0"):
        self.search_band_width = params['search_band_width']
        self.max_run_count = params['max_run_count']
        self.seq_ref = ref
        self.N = ref.shape[1] * 2
        self.M = ref.shape[1]
        self.F = ref.shape[0]
        self._setup(ref, self.search_band_width, self.max_run_count, device=device)

    @property
    def live_ptr(self):
        return self._st()["t"]

    @property
    def ref_ptr(self):
        return self._st()["j"]
