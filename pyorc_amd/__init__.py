"""pyorc_amd -- MI355X-native LSPIV cross-correlation engine (drop-in for pyorc's ``Frames.get_piv``).

Only the hot path lives here: hand-written HIP kernels behind a C ABI (``liblspiv_hip.so``,
declared in ``include/lspiv.h``) and the thin Python host code that mirrors the reference
interfaces around it:

  pyorc_amd.window        <-> ffpiv.window            (grid, memory planner)
  pyorc_amd.piv           <-> ffpiv                   (cross_corr, u_v_displacement + fused piv_pairs)
  pyorc_amd.velocimetry   <-> pyorc.velocimetry.ffpiv (get_ffpiv: chunking, halo, ensemble, px -> m/s)
  pyorc_amd.frames        <-> pyorc.api.frames        (get_piv, engine="hip")
  pyorc_amd.shard / comm                               (frame-pair sharding over the GPUs of a node, RCCL through the C ABI)
  pyorc_amd.device                                     (DeviceFrames: HBM-resident stacks, so normalize -> project -> get_piv never leaves the GPU)
  pyorc_amd.executor                                   (chunk executor: lazy chunks are materialised ahead of the launches that consume them)
  pyorc_amd.plugin                                     (install(): engine="hip" inside an installed, unmodified pyorc -- called on import when pyorc is found)
"""

__version__ = "0.1.0"

from . import window  # noqa: F401
from .piv import cross_corr, piv_pairs, u_v_displacement  # noqa: F401
from ._lib import get_option, pinned_empty, set_option  # noqa: F401,E402
from .device import DeviceFrames  # noqa: F401,E402
from .plugin import install, uninstall  # noqa: F401,E402
from . import plugin as _plugin  # noqa: E402

_plugin.auto_install()   # engine="hip" in an installed pyorc (no-op without pyorc; LSPIV_NO_AUTO_INSTALL=1 switches it off)
