"""Drop-in for the reference's dtw.py: ``DTW(seq_a, seq_b) -> (cost, acc_cost, path)``, plus a
batched form over many pairs.  Computation: csrc/dtw.hip (reference: dtw.py:5-53)."""
import ctypes

import numpy as np
import torch

from . import _native as nat
from .otw_batch import frames_tensor, _np_dtype_code


def dtw_batch(a_dev, b_dev, want_back=True, check=False):
    """a_dev: [B][M][12] or [M][12] (shared), b_dev: [B][N][12] or [N][12] (shared); device
    tensors, float32/float64.  Returns device tensors (cost [B][M][N] f64, acc [B][M][N] f64,
    back [B][M][N] int8 -- None unless want_back --, path [B][M+N][2] int32, path_len [B] int32).
    Asynchronous.

    Fault contract: ``path_len[k] == -1`` means the device pipeline reported a fault (a bounded in-launch wait between
    workgroups ran out -- never expected); the other outputs of that call are then not to be used.  ``check=True``
    synchronises and raises RtsyncError instead of leaving that to the caller."""
    dev = a_dev.device
    sa = a_dev.dim() == 2
    sb = b_dev.dim() == 2
    B = 1 if (sa and sb) else (b_dev.shape[0] if sa else a_dev.shape[0])
    M, N = a_dev.shape[-2], b_dev.shape[-2]
    a_dev, b_dev = a_dev.contiguous(), b_dev.contiguous()
    cost = torch.empty((B, M, N), dtype=torch.float64, device=dev)
    acc = torch.empty((B, M, N), dtype=torch.float64, device=dev)
    back = torch.empty((B, M, N), dtype=torch.int8, device=dev) if want_back else None
    path = torch.empty((B, M + N, 2), dtype=torch.int32, device=dev)
    plen = torch.zeros((B,), dtype=torch.int32, device=dev)
    nbytes = ctypes.c_size_t(0)
    nat.check(nat.lib.rts_dtw_workspace_bytes(M, N, B, ctypes.byref(nbytes)))
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=dev)
    nat.check(nat.lib.rts_dtw(a_dev.data_ptr(), _np_dtype_code(a_dev.dtype), 0 if sa else M,
                              b_dev.data_ptr(), _np_dtype_code(b_dev.dtype), 0 if sb else N,
                              12, M, N, B, cost.data_ptr(), acc.data_ptr(),
                              back.data_ptr() if back is not None else None, path.data_ptr(),
                              plen.data_ptr(), ws.data_ptr(), nbytes.value,
                              ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    if check and int(plen.min().item()) < 1:
        raise nat.RtsyncError("rts_dtw: the device pipeline reported a fault (path_len = -1)")
    return cost, acc, back, path, plen


def DTW(seq_a, seq_b, device="cuda:0"):
    """seq_a (12, M), seq_b (12, N) feature-major like the reference; rows of the returned matrices
    index seq_a.  Returns (cost (M,N) float64, acc_cost (M,N) float64, path (P,2) int64)."""
    if not torch.cuda.is_available():
        raise RuntimeError("DTW needs a ROCm GPU (no CPU fallback)")
    dev = torch.device(device)
    a = frames_tensor(np.asarray(seq_a, dtype=np.float64), dev)
    b = frames_tensor(np.asarray(seq_b, dtype=np.float64), dev)
    cost, acc, _, path, plen = dtw_batch(a, b, want_back=False)
    n = int(plen[0].item())
    if n < 1:
        raise nat.RtsyncError("rts_dtw: the device pipeline reported a fault")
    return cost[0].cpu().numpy(), acc[0].cpu().numpy(), path[0, :n].cpu().numpy().astype(np.int64)
