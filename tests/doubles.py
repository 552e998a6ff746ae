"""Stand-ins for the GPU calls in the CPU tests of the host mirrors: the same signatures, the oracle behind them."""
import numpy as np


def oracle_piv_pairs(fr, ws, ov, thr=None, pair_offset=0, out=None, scale=None):
    """``pyorc_amd.piv.piv_pairs`` (signature as of round 5, with ``out`` and ``scale``) computed by the C oracle; the px -> m/s scaling
    the way the reference writes it (ffpiv.py:418-419)."""
    from oracle import c_oracle

    res = tuple(a.astype(np.float32) for a in c_oracle.piv_pairs(np.asarray(fr), ws, ov, thr))
    if scale is not None:
        res_x, res_y, dt = scale
        dt = np.asarray(dt, dtype=np.float64)
        res = ((res[0] * res_x / np.expand_dims(dt, (1, 2))).astype(np.float32), (res[1] * res_y / np.expand_dims(dt, (1, 2))).astype(np.float32),
               res[2], res[3])
    if out is None:
        return res
    for dst, src in zip(out, res):
        assert dst.dtype == np.float32 and dst.shape == src.shape and dst.flags.c_contiguous
        dst[...] = src
    return tuple(out)
