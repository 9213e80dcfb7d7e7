#!/bin/bash
# On the GPU box: two builds of the library on the same box, alternating (FLX_LIBRARY picks the build): the default one against $2
T=${1:-ab}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
B=${2:-$R/floxer_amd/libfloxer_amd_base.so}
for round in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export FLX_LIBRARY=$B; else unset FLX_LIBRARY; fi
    python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-host-inputs-leg --no-repeat-rich-leg $3 > $O/${v}_$round.json 2> $O/${v}_$round.err || { tail -3 $O/${v}_$round.err; exit 1; }
    python3 -c "
import json
d=json.load(open('$O/${v}_$round.json')); print('$v build, run $round:', d['value'], 'reads/s', d['ms_per_step'], 'ms/step', {k:round(v['device_ms'],1) for k,v in (d.get('kernels_isolated') or {}).items()})"
  done
done
