"""Mirror of ``ffpiv.window`` for the call sites on pyorc's PIV path.

Reference call sites: pyorc/api/frames.py:85-90 (``get_rect_coordinates``), :167
(``round_to_even``); pyorc/velocimetry/ffpiv.py:120-126 (``required_memory``), :129
(``available_memory``).  The grid functions call the C ABI (host-only code in
liblspiv_hip.so, no GPU needed); the memory functions answer for HBM instead of host RAM,
because that is what bounds a chunk on the ``hip`` engine.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib


def round_to_even(input_tuple: Sequence[float]) -> Tuple[int, ...]:
    """Round window sizes to even integers (pyorc/api/frames.py:167).  The direction for odd sizes is the
    library option ``round_odd``: 0 round-half-even of x / 2 (25 -> 24, 27 -> 28; default), 1 up, 2 down."""
    mode = _lib.get_option("round_odd")
    if mode == 1:
        return tuple(int(np.ceil(float(x) / 2.0) * 2) for x in input_tuple)
    if mode == 2:
        return tuple(int(np.floor(float(x) / 2.0) * 2) for x in input_tuple)
    return tuple(int(np.round(float(x) / 2.0) * 2) for x in input_tuple)


def get_axis_shape(dim_size: int, window_size: int, overlap: int) -> int:
    nr, nc = C.c_int64(), C.c_int64()
    _lib.check(_lib.load().lspiv_grid_shape(dim_size, dim_size, window_size, window_size, overlap, overlap,
                                            C.byref(nr), C.byref(nc)))
    return nr.value


def get_array_shape(dim_size, window_size, overlap) -> Tuple[int, int]:
    nr, nc = C.c_int64(), C.c_int64()
    _lib.check(_lib.load().lspiv_grid_shape(dim_size[0], dim_size[1], window_size[0], window_size[1],
                                            overlap[0], overlap[1], C.byref(nr), C.byref(nc)))
    return nr.value, nc.value


def get_rect_coordinates(dim_size, window_size, overlap, search_area_size=None, center_on_field=False):
    """Window-centre pixel indices ``(x_cols, y_rows)``, int64 (usable for fancy indexing)."""
    if center_on_field:
        raise NotImplementedError("pyorc never centres the grid on the field")
    sa = window_size if search_area_size is None else search_area_size
    n_rows, n_cols = get_array_shape(dim_size, sa, overlap)
    rows = np.empty(max(n_rows, 0), dtype=np.int64)
    cols = np.empty(max(n_cols, 0), dtype=np.int64)
    _lib.check(_lib.load().lspiv_grid_coords(dim_size[0], dim_size[1], sa[0], sa[1], overlap[0], overlap[1],
                                             rows.ctypes.data_as(C.POINTER(C.c_int64)),
                                             cols.ctypes.data_as(C.POINTER(C.c_int64))))
    return cols, rows


def required_memory(n_frames: int, dim_size, window_size, overlap, search_area_size=None,
                    dtype=np.uint8, with_planes: bool = False) -> int:
    """HBM bytes one fused call on ``n_frames`` frames needs (frames + four result planes).

    The reference's figure is the host RAM of the materialised window stack + correlation volume
    (x3.9 .. x14.8 of the frames); the fused kernel materialises neither.
    """
    sa = window_size if search_area_size is None else search_area_size
    code = _lib.DTYPE_CODES[np.dtype(dtype)]
    r = _lib.load().lspiv_required_bytes(n_frames, dim_size[0], dim_size[1], code, sa[0], sa[1],
                                         overlap[0], overlap[1], int(with_planes))
    return _lib.check(r)


def available_memory() -> int:
    """Free HBM on the current device in bytes (library workspaces counted as reusable)."""
    free, total = C.c_int64(), C.c_int64()
    _lib.check(_lib.load().lspiv_available_bytes(C.byref(free), C.byref(total)))
    return free.value


def available_host_memory() -> int:
    """Free host RAM in bytes -- what ``ffpiv.window.available_memory`` answers in the reference (pyorc/velocimetry/ffpiv.py:129), and
    what bounds how much of a LAZY stack ``get_ffpiv`` may materialise at a time (the chunks that ``.load()`` brings in live on the
    host before they cross PCIe).  ``psutil`` when importable, ``/proc/meminfo`` otherwise."""
    try:
        import psutil

        return int(psutil.virtual_memory().available)
    except Exception:
        try:
            with open("/proc/meminfo") as fh:
                for line in fh:
                    if line.startswith("MemAvailable:"):
                        return int(line.split()[1]) * 1024
        except OSError:
            pass
    return 8 << 30   # nothing to ask: a conservative 8 GiB


def chunk_alignment(window_size, dim_size=None, overlap=None) -> int:
    """Frame pairs between two anchors of the time-walking kernels: time chunks that start on a multiple of it reproduce the
    whole-stack result bit for bit.  1 for per-pair kernels.  Host-only.

    The anchor length depends on the window GRID since round 5 (25 pairs; 75 on grids with at least as many windows as the chip has
    lane groups, ``lspiv_chunk_alignment_grid``): pass the frame shape ``dim_size`` and the ``overlap`` whenever chunks of frames of
    that shape are cut.  Without them the alignment that is right on EVERY grid comes back (``lspiv_chunk_alignment``, ABI 5: the
    longest anchor length of the window family, a multiple of every grid's -- 75 where ABI 4 answered 25)."""
    lib = _lib.load()
    if dim_size is None:
        return _lib.check(lib.lspiv_chunk_alignment(int(window_size[0]), int(window_size[1])))
    ov = (int(window_size[0]) // 2, int(window_size[1]) // 2) if overlap is None else overlap
    return _lib.check(lib.lspiv_chunk_alignment_grid(int(dim_size[0]), int(dim_size[1]), int(window_size[0]), int(window_size[1]),
                                                     int(ov[0]), int(ov[1])))


def chunk_alignment_any_grid(window_size) -> int:
    """The alignment that is right for EVERY frame shape (= ``chunk_alignment(window_size)`` since ABI 5).  For callers that cut the time
    axis before they know the frames (``shard.sharded_piv`` without ``frame_shape``)."""
    return chunk_alignment(window_size)
