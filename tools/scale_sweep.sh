#!/bin/bash
# RCCL settings sweep for the multi-GPU line: NCCL_MAX_NCHANNELS in {8, 16, 32} x gather-stream priority in
# {default, high}, each a full `bench.py --gpus N` run (weak `value` + the strong pass in config.comm).  For whoever has an N-GPU node --
# the build container has none and gpurun boxes have one GPU; the defaults (16 channels, high priority) were chosen without ever seeing
# two ranks.  Usage: tools/scale_sweep.sh [N=8] [steps=20]; one JSON line per setting on stdout, prefixed with the setting.
N=${1:-8}; STEPS=${2:-20}
R=$(cd "$(dirname "$0")/.." && pwd)
for ch in 8 16 32; do
  for prio in 0 1; do
    line=$(LSPIV_RCCL_MAX_NCHANNELS=$ch LSPIV_GATHER_STREAM_PRIORITY=$prio python "$R/bench.py" --gpus "$N" --steps "$STEPS" --warmup 5 --no-extras --cpu-pairs 0 --sustained-s 0 2>/dev/null | tail -1)
    echo "{\"nchannels\": $ch, \"gather_priority\": \"$([ $prio = 1 ] && echo high || echo default)\", \"line\": $line}"
  done
done
