"""CPU oracle for the chroma + DTW/OTW/WTW hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  The product (``real_time_audio_sync_amd``) never imports
it and has no CPU fallback.

Parity pinning (see DESIGN.md "Oracle"): the C restatement (``rtsync_oracle.c``) is checked
bit-for-bit -- path indices *and* accumulated costs -- against outputs of the reference's own
``otw_eran.py`` / ``livenote.py`` / ``livenote_v2.py`` / ``dtw.py`` executed in the build
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``), and the chroma + WTW chain
against the reference's one reproducible known-answer file, ``Songs/chopin/tests/wtw_test_20b.txt``
(committed as ``tests/golden/wtw_test_20b.txt``).
"""
from .binding import (  # noqa: F401
    OTW, LIVENOTE, LIVENOTE_V2, COST_DOT, COST_EUCLID,
    DIR_NONE, DIR_BOTH, DIR_ROW, DIR_COLUMN,
    RUNNING, STOP_REF_END, LIVE_OVERFLOW,
    build, lib, OtwOracle, dtw, wtw_cost_matrix, wtw_run_dtw, wtw_find_path, WtwOracle,
    dot_strided, dot_chain, euclid,
)
