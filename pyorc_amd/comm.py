"""One-process-per-GPU communicator over the C ABI (``lspiv_comm_*``: RCCL over xGMI, no PyTorch).

The reference has no communication (single process).  Here the time axis is sharded over the GPUs of a node and the only
exchange is one all-gather of the packed result block (ensemble mode: one sum all-reduce), see ``pyorc_amd.shard``.

Rendezvous: rank 0 obtains the 128-byte id from the library and publishes it in a file on the node; the other ranks poll
for it.  The file is ``$LSPIV_COMM_ID_FILE`` if set (``bench.py --gpus N`` sets it for the ranks it spawns), otherwise
``/tmp/lspiv_comm_<MASTER_PORT>_<parent pid>`` -- under ``python -m torch.distributed.run`` all ranks of a node share
the parent (the elastic agent) and ``MASTER_PORT``, so the name is unique per job without any torch import.

Transports: ``"rccl"`` (default) and ``"shm"`` (POSIX shared memory; plumbing tests on a 1-GPU box or without a GPU --
RCCL cannot run there; ``LSPIV_COMM=shm`` selects it from the environment).
"""

from __future__ import annotations

import ctypes as C
import os
import time
from typing import Optional

import numpy as np

from . import _lib

RCCL, SHM = 0, 1
SUM, MAX = 0, 1
ID_BYTES = 128
_TRANSPORTS = {"rccl": RCCL, "shm": SHM}
_DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2}


def _rendezvous_dir() -> str:
    """A directory only this user can write: XDG_RUNTIME_DIR, else a 0700 directory of our own under the temp dir."""
    d = os.environ.get("XDG_RUNTIME_DIR")
    if d and os.path.isdir(d) and os.access(d, os.W_OK):
        return d
    import tempfile

    d = os.path.join(tempfile.gettempdir(), f"lspiv-{os.getuid()}")
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.lstat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077) or not os.path.isdir(d) or os.path.islink(d):
        raise PermissionError(f"{d} is not a private directory of uid {os.getuid()}")
    return d


def default_id_file() -> str:
    f = os.environ.get("LSPIV_COMM_ID_FILE")
    if f:
        return f
    # under torch.distributed.run all ranks share the agent as parent; bench.py passes LSPIV_COMM_ID_FILE to its own ranks
    return os.path.join(_rendezvous_dir(), f"lspiv_comm_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")


def _job_nonce() -> bytes:
    """8 bytes that tell this job's id file from a stale one of a crashed run at the same path: the launcher's start time
    (LSPIV_COMM_NONCE when given, else TORCHELASTIC_RUN_ID together with the parent process's start time from /proc)."""
    tag = os.environ.get("LSPIV_COMM_NONCE")
    if not tag:
        # torch.distributed.run's default run id is the constant "none": the parent's start time always goes in
        try:
            with open(f"/proc/{os.getppid()}/stat", "rb") as fh:
                born = fh.read().rsplit(b")", 1)[1].split()[19].decode()   # starttime of the parent, in clock ticks
        except (OSError, IndexError):
            born = str(os.getppid())
        tag = f"{os.environ.get('TORCHELASTIC_RUN_ID', '')}:{born}"
    import hashlib

    return hashlib.sha256(str(tag).encode()).digest()[:8]


def exchange_id(rank: int, world: int, transport: int, path: str, timeout: float = 300.0) -> bytes:
    """Rank 0 creates the communicator id and publishes ``nonce + id`` at ``path``: written to a fresh O_EXCL | O_NOFOLLOW
    file (0600) next to it and renamed over whatever a crashed run may have left there.  The other ranks poll and accept
    only a file that carries this job's nonce, so a stale id is never handed to ncclCommInitRank."""
    lib = _lib.load()
    nonce = _job_nonce()
    if rank == 0:
        buf = C.create_string_buffer(ID_BYTES)
        _lib.check(lib.lspiv_comm_unique_id(transport, buf))
        tmp = f"{path}.{os.getpid()}.tmp"
        try:
            os.unlink(tmp)
        except OSError:
            pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "wb") as fh:
            fh.write(nonce + buf.raw)
        os.replace(tmp, path)
        return buf.raw
    t0 = time.time()
    while True:
        try:
            fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
            with os.fdopen(fd, "rb") as fh:
                raw = fh.read()
            if len(raw) == len(nonce) + ID_BYTES and raw[:len(nonce)] == nonce:
                return raw[len(nonce):]
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no communicator id of this job at {path} after {timeout:.0f} s")
        time.sleep(0.02)


class _rccl_channel_cap:
    """``NCCL_MAX_NCHANNELS = LSPIV_RCCL_MAX_NCHANNELS (default 16)`` for the duration of the block, unless the user set
    ``NCCL_MAX_NCHANNELS`` himself; the previous environment is restored on exit.  ``applied`` records what RCCL saw."""

    applied: Optional[str] = None   # class-wide: the value in force when the last RCCL communicator was created

    def __init__(self, active: bool):
        self.active = active
        self.touched = False

    def __enter__(self):
        if self.active:
            if "NCCL_MAX_NCHANNELS" not in os.environ:
                os.environ["NCCL_MAX_NCHANNELS"] = os.environ.get("LSPIV_RCCL_MAX_NCHANNELS", "16")
                self.touched = True
            _rccl_channel_cap.applied = os.environ["NCCL_MAX_NCHANNELS"]
        return self

    def __exit__(self, *exc):
        if self.touched:
            os.environ.pop("NCCL_MAX_NCHANNELS", None)
        return False


class Comm:
    """A communicator of ``world`` ranks; ``rank`` r must have made device r its current device (RCCL)."""

    def __init__(self, rank: int, world: int, transport: Optional[str] = None, id_file: Optional[str] = None,
                 timeout: float = 300.0):
        transport = transport or os.environ.get("LSPIV_COMM", "rccl")
        if transport not in _TRANSPORTS:
            raise ValueError(f"transport {transport!r} not in {sorted(_TRANSPORTS)}")
        self.rank, self.world, self.transport = int(rank), int(world), transport
        self._lib = _lib.load()
        self._h = C.c_void_p()
        path = id_file or default_id_file()
        uid = exchange_id(self.rank, self.world, _TRANSPORTS[transport], path, timeout)
        # The gather runs on RCCL's own kernels NEXT TO the PIV kernel, which saturates the VALUs of every CU it gets: cap the
        # channels (one workgroup each) RCCL may take unless the user decided otherwise (NCCL_MAX_NCHANNELS already set).  16
        # channels move the 126 MB / rank result block of a 1000-pair step well inside the step's 6 ms on xGMI and leave > 90 % of
        # the CUs to the PIV kernel.  The variable is set only AROUND ncclCommInitRank and put back afterwards, so that code which
        # reads the environment later (a child process, another library) does not inherit a cap it never asked for.  RCCL itself
        # reads such parameters ONCE per process: the first communicator created in a process fixes the value for all later ones
        # (torch.distributed's included, either way round) -- see INTEGRATION.md, "RCCL settings".
        with _rccl_channel_cap(transport == "rccl"):
            _lib.check(self._lib.lspiv_comm_init(self.rank, self.world, uid, _TRANSPORTS[transport], C.byref(self._h)))
        self.barrier()  # every rank has read the id: rank 0 may remove the file
        if self.rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass

    @classmethod
    def from_env(cls, **kw) -> "Comm":
        """RANK / WORLD_SIZE as ``torch.distributed.run`` (or ``bench.py``'s own launcher) exports them."""
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), **kw)

    @property
    def backend_ranks(self) -> int:
        """The rank count the transport itself reports (``ncclCommCount`` for RCCL)."""
        n = C.c_int(0)
        _lib.check(self._lib.lspiv_comm_info(self._h, None, None, None, C.byref(n)))
        return n.value

    # ---- host arrays -------------------------------------------------------------------------
    def allgather(self, arr: np.ndarray) -> np.ndarray:
        """(world,) + arr.shape: every rank's array, in rank order (float32 / float64, same shape on all ranks)."""
        a = np.ascontiguousarray(arr)
        if a.dtype not in _DT:
            raise TypeError(f"collectives take float32 / float64, got {a.dtype}")
        out = np.empty((self.world,) + a.shape, dtype=a.dtype)
        _lib.check(self._lib.lspiv_comm_allgather(self._h, _lib.ptr(a), _lib.ptr(out), a.size, _DT[a.dtype]))
        return out

    def allreduce(self, arr: np.ndarray, op: int = SUM) -> np.ndarray:
        a = np.ascontiguousarray(arr)
        if a.dtype not in _DT:
            raise TypeError(f"collectives take float32 / float64, got {a.dtype}")
        out = np.empty_like(a)
        _lib.check(self._lib.lspiv_comm_allreduce(self._h, _lib.ptr(a), _lib.ptr(out), a.size, _DT[a.dtype], op))
        return out

    # ---- device pointers ---------------------------------------------------------------------
    def allgather_dev(self, d_send: int, d_recv: int, count: int, dtype=np.float32, stream: Optional[int] = None):
        _lib.check(self._lib.lspiv_comm_allgather_dev(self._h, C.c_void_p(d_send), C.c_void_p(d_recv), int(count),
                                                      _DT[np.dtype(dtype)], C.c_void_p(stream) if stream else None))

    def allreduce_dev(self, d_send: int, d_recv: int, count: int, dtype=np.float32, op: int = SUM,
                      stream: Optional[int] = None):
        _lib.check(self._lib.lspiv_comm_allreduce_dev(self._h, C.c_void_p(d_send), C.c_void_p(d_recv), int(count),
                                                      _DT[np.dtype(dtype)], op, C.c_void_p(stream) if stream else None))

    def barrier(self):
        _lib.check(self._lib.lspiv_comm_barrier(self._h))

    def close(self):
        if self._h:
            self._lib.lspiv_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
