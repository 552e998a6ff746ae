import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pyorc_amd
from oracle import c_oracle
from pyorc_amd.synth import particle_stack
fr = particle_stack(3, 128, 144, seed=17, density=0.06)
ws, ov = (4, 6), (2, 3)
u, v, cm, sn, pl = pyorc_amd.piv_pairs(fr, ws, ov, return_planes=True)
uo, vo, cmo, sno, plo, cond = c_oracle.piv_pairs(fr, ws, ov, return_planes=True, return_cond=True)
ok = c_oracle.well_posed(cond, min_neighbour=0.2)
e = np.abs(u - uo) / np.maximum(np.abs(uo), 0.05); e[~ok] = 0; e = np.nan_to_num(e)
i = np.unravel_index(e.argmax(), e.shape); print(i, e[i], u[i], uo[i], cond[i], ok.mean())
w = i[1] * u.shape[2] + i[2]
print(pl[i[0], w]); print(plo[i[0], w]); print(np.abs(pl[i[0], w] - plo[i[0], w]).max())
