"""Drop-in for the reference's livenote_v2.py: ``LiveNoteV2(ref, params, debug_params,
chroma_diff=False)`` -- forward-only path, optional Euclidean chroma-diff cost
(reference: livenote_v2.py:3-236)."""
from ._dropin import OtwDropIn, _DIR_NAMES_LOW


class LiveNoteV2(OtwDropIn):
    _variant = "livenote_v2"
    _names = _DIR_NAMES_LOW
    _msg_overflow = "done - oob live"
    _msg_stop = "done - oob ref"

    def __init__(self, ref, params, debug_params, chroma_diff=False, device="cuda:0"):
        self.search_band_width = params['search_band_width']
        self.max_run_count = params['max_run_count']
        self.seq_ref = ref
        self.N = ref.shape[1] * 2
        self.M = ref.shape[1]
        self.F = ref.shape[0]
        self.chroma_diff = chroma_diff
        self._setup(ref, self.search_band_width, self.max_run_count, euclid=bool(chroma_diff), device=device)

    @property
    def live_ptr(self):
        return self._st()["t"]

    @property
    def ref_ptr(self):
        return self._st()["j"]
