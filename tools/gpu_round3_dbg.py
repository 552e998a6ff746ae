import faulthandler, sys, os, time
faulthandler.dump_traceback_later(45, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from pyorc_amd import _lib, piv
from pyorc_amd.synth import particle_stack
def say(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)
lib = _lib.load(); _lib.require_device()
fr = particle_stack(4, 128, 160, seed=7)
for resc in (0, 1):
    _lib.set_option("rescue", resc)
    say("rescue", resc, "launch small")
    out = piv.piv_pairs(fr, (32, 32), (16, 16))
    say("done", float(np.nanmean(out[0])))
    st = (C.c_int64 * 5)(); _lib.check(lib.lspiv_rescue_stats(None, st)); say("stats", list(st))
say("planes path")
out = piv.piv_pairs(fr, (32, 32), (16, 16), return_planes=True)
st = (C.c_int64 * 5)(); _lib.check(lib.lspiv_rescue_stats(None, st)); say("stats", list(st))
fr = particle_stack(25, 1080, 1920, seed=7)
say("big host-fed")
out = piv.piv_pairs(fr, (32, 32), (16, 16))
st = (C.c_int64 * 5)(); _lib.check(lib.lspiv_rescue_stats(None, st)); say("stats", list(st))
say("big host-fed + planes")
out = piv.piv_pairs(fr, (32, 32), (16, 16), return_planes=True)
st = (C.c_int64 * 5)(); _lib.check(lib.lspiv_rescue_stats(None, st)); say("stats", list(st))
say("64")
out = piv.piv_pairs(fr[:7], (64, 64), (48, 48))
st = (C.c_int64 * 5)(); _lib.check(lib.lspiv_rescue_stats(None, st)); say("stats", list(st))
say("ok")
