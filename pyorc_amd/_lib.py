"""ctypes binding of ``liblspiv_hip.so`` (the C ABI declared in ``include/lspiv.h``).

This is the only place the package touches native code.  There is NO CPU fallback: if the
shared object is missing, or no gfx950 device is visible when a compute entry point is
called, a loud exception is raised (``LspivLibraryMissing`` / ``LspivError``).
"""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LSPIV_LIBRARY: another build of the same ABI (A/B measurements of kernel variants, tools/ab_time.py); the default is the in-tree build
LIB_PATH = os.environ.get("LSPIV_LIBRARY") or os.path.join(_HERE, "liblspiv_hip.so")

LSPIV_OK = 0
LSPIV_EINVAL = -1
LSPIV_ESHAPE = -2
LSPIV_ENOMEM = -3
LSPIV_EHIP = -4
LSPIV_ENODEV = -5
LSPIV_EUNSUPPORTED = -6

DTYPE_CODES = {np.dtype(np.uint8): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}


class LspivLibraryMissing(ImportError):
    """liblspiv_hip.so has not been built (run ``python -c 'import __graft_entry__ as g; g.build()'``)."""


class LspivLibraryStale(ImportError):
    """liblspiv_hip.so was built from other sources than the ones in this tree (``make -C pyorc_amd/csrc`` rebuilds it;
    ``LSPIV_ALLOW_STALE=1`` loads it anyway -- every measurement then says so, see ``binary_provenance``)."""


class LspivError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"lspiv error {code}: {message}")
        self.code = code


# every symbol include/lspiv.h declares: name -> (restype, argtypes)
_i64, _i32, _f32, _vp, _sz = C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_size_t
_pf = C.POINTER(C.c_float)
_pi64 = C.POINTER(C.c_int64)
SIGNATURES = {
    "lspiv_abi_version": (_i32, []),
    "lspiv_version": (C.c_char_p, []),
    "lspiv_build_info": (C.c_char_p, [_i32]),
    "lspiv_last_error": (C.c_char_p, []),
    "lspiv_device_count": (_i32, [C.POINTER(_i32)]),
    "lspiv_set_device": (_i32, [_i32]),
    "lspiv_get_device": (_i32, [C.POINTER(_i32)]),
    "lspiv_device_name": (_i32, [_i32, C.c_char_p, _sz]),
    "lspiv_synchronize": (_i32, []),
    "lspiv_set_option": (_i32, [C.c_char_p, _i32]),
    "lspiv_get_option": (_i32, [C.c_char_p, C.POINTER(_i32)]),
    "lspiv_rescue_stats": (_i32, [_vp, _pi64]),
    "lspiv_kernel_kind": (_i32, [_i32, _i32]),
    "lspiv_grid_shape": (_i32, [_i64, _i64, _i32, _i32, _i32, _i32, _pi64, _pi64]),
    "lspiv_grid_coords": (_i32, [_i64, _i64, _i32, _i32, _i32, _i32, _pi64, _pi64]),
    "lspiv_required_bytes": (_i64, [_i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32]),
    "lspiv_available_bytes": (_i32, [_pi64, _pi64]),
    "lspiv_piv_pairs": (_i32, [_vp, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "lspiv_piv_pairs_dev": (_i32, [_vp, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    "lspiv_chunk_alignment": (_i32, [_i32, _i32]),
    "lspiv_chunk_alignment_grid": (_i32, [_i64, _i64, _i32, _i32, _i32, _i32]),
    "lspiv_piv_pairs_at": (_i32, [_vp, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "lspiv_piv_velocity_at": (_i32, [_vp, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _i64, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "lspiv_piv_pairs_dev_at": (_i32, [_vp, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _i64, _vp, _vp, _vp]),
    "lspiv_u_v_displacement": (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "lspiv_ensemble_begin": (_i32, [_i64, _i64, _i32, _i32, _i32, _i32, C.POINTER(_vp)]),
    "lspiv_ensemble_accumulate": (_i32, [_vp, _vp, _i32, _i64, _f32, _f32, _f32, _vp, _vp]),
    "lspiv_ensemble_accumulate_dev": (_i32, [_vp, _vp, _i32, _i64, _f32, _f32, _f32, _vp, _vp]),
    "lspiv_ensemble_finish": (_i32, [_vp, _f32, _f32, _vp, _vp, _vp, _vp]),
    "lspiv_ensemble_set_retain": (_i32, [_vp, _i32]),
    "lspiv_ensemble_stats": (_i32, [_vp, _pi64]),
    "lspiv_ensemble_flag": (_i32, [_vp, _f32, _f32, _pi64]),
    "lspiv_ensemble_partials": (_i32, [_vp, _vp, C.POINTER(_i32)]),
    "lspiv_ensemble_flag_digest": (_i32, [_vp, C.POINTER(C.c_uint64)]),
    "lspiv_ensemble_finish_partials": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "lspiv_ensemble_export": (_i32, [_vp, _vp, _vp]),
    "lspiv_ensemble_import": (_i32, [_vp, _vp, _vp, _i32]),
    "lspiv_ensemble_destroy": (_i32, [_vp]),
    "lspiv_projection_create": (_i32, [_i64, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, C.POINTER(_vp)]),
    "lspiv_project_frames": (_i32, [_vp, _vp, _i32, _i64, _vp]),
    "lspiv_project_frames_dev": (_i32, [_vp, _vp, _i32, _i64, _vp, _vp]),
    "lspiv_project_frames_u8": (_i32, [_vp, _vp, _i64, _vp]),
    "lspiv_project_frames_u8_dev": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "lspiv_projection_destroy": (_i32, [_vp]),
    "lspiv_project_cv_create": (_i32, [_i64, _i64, _i64, _i64, _vp, _vp, _i32, _vp, C.POINTER(_vp)]),
    "lspiv_project_cv_frames": (_i32, [_vp, _vp, _i32, _i64, _vp]),
    "lspiv_project_cv_frames_dev": (_i32, [_vp, _vp, _i32, _i64, _vp, _vp]),
    "lspiv_project_cv_destroy": (_i32, [_vp]),
    "lspiv_time_diff": (_i32, [_vp, _i32, _i64, _i64, _i64, _f32, _i32, _vp]),
    "lspiv_time_diff_dev": (_i32, [_vp, _i32, _i64, _i64, _i64, _f32, _i32, _vp, _vp]),
    "lspiv_time_range": (_i32, [_vp, _i32, _i64, _i64, _i64, _vp]),
    "lspiv_time_range_dev": (_i32, [_vp, _i32, _i64, _i64, _i64, _vp, _vp]),
    "lspiv_minmax": (_i32, [_vp, _i64, _f32, _f32, _vp]),
    "lspiv_minmax_dev": (_i32, [_vp, _i64, _f32, _f32, _vp, _vp]),
    "lspiv_normalize": (_i32, [_vp, _i64, _i64, _i64, _i32, _vp]),
    "lspiv_normalize_dev": (_i32, [_vp, _i64, _i64, _i64, _i32, _vp, _vp]),
    "lspiv_normalize_mean_dev": (_i32, [_vp, _i64, _i64, _i64, _i32, _vp, _vp]),
    "lspiv_normalize_apply_dev": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "lspiv_reduce_rolling": (_i32, [_vp, _i64, _i64, _i64, _i32, _vp]),
    "lspiv_reduce_rolling_dev": (_i32, [_vp, _i64, _i64, _i64, _i32, _vp, _vp]),
    "lspiv_gaussian_blur": (_i32, [_vp, _i32, _i64, _i64, _i64, _i32, _vp]),
    "lspiv_gaussian_blur_dev": (_i32, [_vp, _i32, _i64, _i64, _i64, _i32, _vp, _vp]),
    "lspiv_edge_detect": (_i32, [_vp, _i32, _i64, _i64, _i64, _i32, _i32, _vp]),
    "lspiv_edge_detect_dev": (_i32, [_vp, _i32, _i64, _i64, _i64, _i32, _i32, _vp, _vp]),
    "lspiv_edge_detect_clip_dev": (_i32, [_vp, _i32, _i64, _i64, _i64, _i32, _i32, _f32, _f32, _vp, _vp]),
    "lspiv_mask": (_i32, [_vp, _i64, _i64, _i64, _i32, _vp, _i32, _vp]),
    "lspiv_mask_dev": (_i32, [_vp, _i64, _i64, _i64, _i32, _vp, _i32, _vp, _vp]),
    "lspiv_mask_apply": (_i32, [_vp, _i64, _i64, _i64, _vp, _i32]),
    "lspiv_mask_apply_dev": (_i32, [_vp, _i64, _i64, _i64, _vp, _i32, _vp]),
    "lspiv_time_mean": (_i32, [_vp, _i64, _i64, _i64, _vp]),
    "lspiv_time_mean_dev": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "lspiv_window_replace": (_i32, [_vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32]),
    "lspiv_window_replace_dev": (_i32, [_vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "lspiv_scale_velocity_dev": (_i32, [_vp, _i64, _i64, C.c_double, C.c_double, _vp, _vp]),
    "lspiv_pack_int16": (_i32, [_vp, _i64, _f32, _i32, _vp]),
    "lspiv_pack_int16_dev": (_i32, [_vp, _i64, _f32, _i32, _vp, _vp]),
    "lspiv_dev_malloc": (_i32, [C.POINTER(_vp), _sz]),
    "lspiv_dev_free": (_i32, [_vp]),
    "lspiv_host_alloc": (_i32, [C.POINTER(_vp), _sz]),
    "lspiv_host_free": (_i32, [_vp]),
    "lspiv_memcpy_h2d": (_i32, [_vp, _vp, _sz]),
    "lspiv_memcpy_d2h": (_i32, [_vp, _vp, _sz]),
    "lspiv_memset_dev": (_i32, [_vp, _i32, _sz]),
    "lspiv_event_create": (_i32, [C.POINTER(_vp)]),
    "lspiv_event_record": (_i32, [_vp]),
    "lspiv_event_elapsed_ms": (_i32, [_vp, _vp, C.POINTER(_f32)]),
    "lspiv_event_destroy": (_i32, [_vp]),
    "lspiv_stream_create": (_i32, [C.POINTER(_vp)]),
    "lspiv_stream_create_priority": (_i32, [C.POINTER(_vp), _i32]),
    "lspiv_stream_destroy": (_i32, [_vp]),
    "lspiv_stream_release": (_i32, [_vp]),
    "lspiv_stream_synchronize": (_i32, [_vp]),
    "lspiv_event_record_on": (_i32, [_vp, _vp]),
    "lspiv_stream_wait_event": (_i32, [_vp, _vp]),
    "lspiv_comm_unique_id": (_i32, [_i32, _vp]),
    "lspiv_comm_init": (_i32, [_i32, _i32, _vp, _i32, C.POINTER(_vp)]),
    "lspiv_comm_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "lspiv_comm_allgather_dev": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "lspiv_comm_allreduce_dev": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "lspiv_comm_allgather": (_i32, [_vp, _vp, _vp, _i64, _i32]),
    "lspiv_comm_allreduce": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32]),
    "lspiv_comm_barrier": (_i32, [_vp]),
    "lspiv_comm_destroy": (_i32, [_vp]),
    "lspiv_synth_particles_dev": (_i32, [_vp, _i64, _i64, _i64, C.c_uint64, _f32]),
    "lspiv_debug_fft": (_i32, [_i32, _i32, _vp, _vp, _i64]),
    "lspiv_debug_narrow": (_i32, [_vp, _i64, _i64, _i32, _vp, _vp]),
    "lspiv_debug_segments": (_i32, [_i64, _i64, _i32, _pi64, _pi64]),
    "lspiv_debug_hold_lock": (_i32, [_i32, _i32, _i32]),
    "lspiv_kernel_times": (_i32, [_vp, _i32, C.POINTER(_i32)]),
    "lspiv_upload_frames": (_i32, [_vp, _vp, _i32, _i64, _i64, _i64, _f32]),
    "lspiv_debug_project_division": (_i32, [C.POINTER(_i32)]),
    "lspiv_trace": (_i32, [_i32]),
    "lspiv_trace_read": (_i32, [_i64, _vp, _vp, _vp, C.POINTER(_i64)]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared object once and attach prototypes; raises if it was never built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LspivLibraryMissing(
            f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU fallback; run "
            "`make -C pyorc_amd/csrc` (or __graft_entry__.build())."
        )
    lib = C.CDLL(LIB_PATH)
    # the binary must come from THIS tree (the .so is a build artefact outside the history): csrc/Makefile compiles the hash
    # of every library source into it.  A build loaded on purpose through LSPIV_LIBRARY (A/B measurements) is exempt.  Checked
    # BEFORE the prototypes are attached: a build from older sources lacks the newer entry points, and "rebuild" is the message
    # its user needs, not an AttributeError about a symbol.
    ab_build = bool(os.environ.get("LSPIV_LIBRARY"))
    if not ab_build and not os.environ.get("LSPIV_ALLOW_STALE"):
        prov = binary_provenance(lib)
        if prov["binary_hash_matches"] is False:
            raise LspivLibraryStale(
                f"{LIB_PATH} was built from sources with hash {prov['binary_source_hash']}, the tree has {prov['tree_source_hash']}: "
                "rebuild (`make -C pyorc_amd/csrc`), or set LSPIV_ALLOW_STALE=1 to load it anyway")
    for name, (res, args) in SIGNATURES.items():
        if not hasattr(lib, name):
            if ab_build:
                continue  # an older build loaded for an A/B measurement: it simply lacks the newer entry points
            raise LspivLibraryStale(f"{LIB_PATH} does not export {name} (include/lspiv.h declares it): the binary is older than the "
                                    "header; rebuild (`make -C pyorc_amd/csrc`)")
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


KERNEL_SOURCES = ("piv_fft_impl.h", "fft_regs.h", "common.h", "piv_rescue.hip")


def _hash_files(paths) -> str:
    import hashlib

    h = hashlib.sha256()
    for f in paths:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def source_hash(csrc_dir: Optional[str] = None, header: Optional[str] = None) -> str:
    """sha256 (16 hex digits) over every source of the library: csrc/*.hip, *.h and *.cpp sorted by name, then include/lspiv.h --
    what csrc/Makefile compiles into the binary as LSPIV_BUILD_SOURCE_HASH."""
    d = csrc_dir or os.path.join(_HERE, "csrc")
    names = sorted(n for n in os.listdir(d) if n.endswith((".hip", ".h", ".cpp")))
    return _hash_files([os.path.join(d, n) for n in names] + [header or os.path.join(os.path.dirname(_HERE), "include", "lspiv.h")])


def binary_provenance(lib: Optional[C.CDLL] = None, csrc_dir: Optional[str] = None) -> dict:
    """The hashes the loaded binary carries next to the ones of the tree (bench.py prints this)."""
    lib = lib if lib is not None else load()
    if hasattr(lib, "lspiv_build_info"):
        lib.lspiv_build_info.restype, lib.lspiv_build_info.argtypes = C.c_char_p, [_i32]
        bk, bs = lib.lspiv_build_info(0).decode(), lib.lspiv_build_info(1).decode()
    else:
        bk = bs = "absent"   # a build from before round 4
    try:
        tk, ts = kernel_code_hash(csrc_dir), source_hash(csrc_dir)
    except OSError:      # a deployment without the sources next to the binary: nothing to compare with
        return {"binary_kernel_hash": bk, "tree_kernel_hash": None, "binary_source_hash": bs, "tree_source_hash": None,
                "binary_hash_matches": None}
    return {"binary_kernel_hash": bk, "tree_kernel_hash": tk, "binary_source_hash": bs, "tree_source_hash": ts,
            "binary_hash_matches": bk == tk and bs == ts}


class LspivValueError(LspivError, ValueError):
    """LSPIV_EINVAL / LSPIV_ESHAPE / LSPIV_EUNSUPPORTED: what the reference reports as ``ValueError`` (include/lspiv.h)."""


class LspivMemoryError(LspivError, MemoryError):
    """LSPIV_ENOMEM."""


_ERROR_TYPES = {LSPIV_EINVAL: LspivValueError, LSPIV_ESHAPE: LspivValueError, LSPIV_EUNSUPPORTED: LspivValueError,
                LSPIV_ENOMEM: LspivMemoryError}


def check(rc: int) -> int:
    """Negative status -> the exception type include/lspiv.h maps it to (always an ``LspivError`` as well)."""
    if rc < 0:
        raise _ERROR_TYPES.get(rc, LspivError)(rc, load().lspiv_last_error().decode("utf-8", "replace"))
    return rc


def device_count() -> int:
    n = C.c_int(0)
    check(load().lspiv_device_count(C.byref(n)))
    return n.value


def require_device() -> None:
    if device_count() < 1:
        raise LspivError(LSPIV_ENODEV, "no gfx950 (MI355X) device visible; engine='hip' has no CPU fallback")


def set_option(name: str, value: int) -> None:
    """Run-time options of the library, e.g. ``set_option("walk", 0)``: per-pair kernels, whose results do not depend
    on how the time axis is chunked (bit for bit); 1 = default time-walking kernels; -1 = follow ``LSPIV_WALK``."""
    check(load().lspiv_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(load().lspiv_get_option(name.encode(), C.byref(v)))
    return v.value


class _PinnedOwner:
    def __init__(self, p):
        self.p = p

    def __del__(self):
        try:
            load().lspiv_host_free(self.p)
        except Exception:
            pass


def pinned_empty(shape, dtype=np.uint8) -> np.ndarray:
    """An uninitialised numpy array in pinned (page-locked) host memory.  A uint8 / float32 frame stack that lives in
    one is DMA'd in place by ``piv_pairs`` -- no staging copy, a few GB/s more over PCIe, no host threads."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = C.c_void_p()
    check(load().lspiv_host_alloc(C.byref(p), n))
    buf = (C.c_char * max(n, 1)).from_address(p.value)
    buf._owner = _PinnedOwner(p)          # keeps the allocation alive as long as any view of the buffer
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


def ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def as_frames(imgs) -> np.ndarray:
    """C-contiguous (T,H,W) array of a supported dtype (other dtypes -> float32 / float64)."""
    a = np.asarray(imgs)
    if a.ndim != 3:
        raise ValueError(f"frames must be (T, H, W), got shape {a.shape}")
    if a.dtype not in DTYPE_CODES:
        if a.dtype == np.bool_ or (a.dtype.kind in "ui" and a.dtype.itemsize <= 2):
            a = a.astype(np.float32)  # exact
        else:
            a = a.astype(np.float64)
    return np.ascontiguousarray(a)


def kernel_code_hash(csrc_dir: Optional[str] = None) -> str:
    """sha256 over the sources of the fused PIV kernels (csrc/piv_fft_impl.h, fft_regs.h, common.h, piv_rescue.hip): what a
    committed profile summary is keyed to (tools/summarize_profile.py writes it, bench.py compares it) and what the binary
    carries as LSPIV_BUILD_KERNEL_HASH."""
    d = csrc_dir or os.path.join(_HERE, "csrc")
    return _hash_files([os.path.join(d, n) for n in KERNEL_SOURCES])
