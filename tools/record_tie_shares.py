"""tests/golden/tie_shares.json from a tie-share log of the GPU suite (LSPIV_TIE_LOG=<file> python -m pytest tests -m gpu):
per test id the share of windows that check_against_oracle gated (1 - exact float64 ties OF THE ORACLE'S OWN planes), call by call.

    python tools/record_tie_shares.py gpurun_out/r5b/ties.log [more logs]"""
import collections
import json
import os
import sys

out = collections.OrderedDict()
for path in sys.argv[1:]:
    for line in open(path):
        test, share, n = line.rstrip("\n").split("\t")
        test = test.split(" (")[0].split("/")[-1]      # "test_gpu_parity.py::test_x[param]": independent of the directory pytest was started in
        out.setdefault(test, []).append(round(float(share), 4))
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tie_shares.json")
json.dump(out, open(dst, "w"), indent=0, sort_keys=True)
print(f"{len(out)} tests, {sum(len(v) for v in out.values())} comparisons, smallest share {min(min(v) for v in out.values()):.4f} -> {dst}")
