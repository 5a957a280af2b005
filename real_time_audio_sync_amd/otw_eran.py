"""Drop-in for the reference's otw_eran.py: ``OnlineTimeWarping(ref, {'c':..., 'max_run_count':...})``
computed by the gfx950 kernels in csrc/otw.hip (reference: otw_eran.py:5-239)."""
from ._dropin import OtwDropIn


class OnlineTimeWarping(OtwDropIn):
    _variant = "otw"

    def __init__(self, ref, params, device="cuda:0"):
        super(OnlineTimeWarping, self).__init__()
        self.c = params['c']
        self.max_run_count = params['max_run_count']
        self.ref = ref
        self._setup(ref, self.c, self.max_run_count, device=device)

    @property
    def t(self):
        return self._st()["t"]

    @property
    def j(self):
        return self._st()["j"]
