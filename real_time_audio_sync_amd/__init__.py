"""MI355X-native chroma + DTW / online-time-warping / windowed-time-warping engine.

Drop-in for the alignment hot path of smritip/real-time-audio-sync (chroma.py, dtw.py,
otw_eran.py, livenote.py, livenote_v2.py, wtw.py): the same Python call surface, computed by
hand-written gfx950 HIP kernels behind a C-ABI shared library (include/rtsync.h).
There is no CPU fallback: importing the native layer fails loudly if librtsync.so is missing.
"""
__version__ = "0.1.0"
