import faulthandler, sys, os, time
faulthandler.dump_traceback_later(40, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from pyorc_amd import _lib, piv
from pyorc_amd.device import DeviceFrames
from pyorc_amd.synth import particle_stack
def say(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)
mode = sys.argv[1]
lib = _lib.load(); _lib.require_device()
def stats():
    st = (C.c_int64 * 5)(); _lib.check(lib.lspiv_rescue_stats(None, st)); return list(st)
P = int(sys.argv[2]) if len(sys.argv) > 2 else 25
fr = particle_stack(P, 1080, 1920, seed=7)
say(mode, "start")
if mode == "dev":
    d = DeviceFrames.from_host(fr)
    out = piv.piv_pairs(d, (32, 32), (16, 16))
elif mode == "host":
    out = piv.piv_pairs(fr, (32, 32), (16, 16))
say(mode, "done", stats(), float(np.nanmean(out[0])))
