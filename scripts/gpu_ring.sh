#!/bin/bash
# On the GPU box: rings that wait (FLX_RING_STRETCH: how much longer than round 3's schedule a job may take, percent; 100 = round 3's shapes)
# against the knobs that share the chip between the kernels. cfg = stretch,K1 launches at a time,lanes
T=${1:-ring_sweep}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
CFGS=${2:-100,6,16 135,6,16 180,6,16 135,3,16 135,8,16 135,6,20 135,6,12}
for cfg in $CFGS; do
  IFS=, read st k1 l <<< "$cfg"
  FLX_RING_STRETCH=$st FLX_K1_CONCURRENT=$k1 FLX_LANES=$l python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-host-inputs-leg --no-repeat-rich-leg $3 > $O/s${st}_k${k1}_l$l.json 2> $O/s${st}_k${k1}_l$l.err || { tail -3 $O/s${st}_k${k1}_l$l.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/s${st}_k${k1}_l$l.json')); print('stretch $st %, K1 launches at a time $k1, lanes $l:', d['value'], 'reads/s', d['ms_per_step'], 'ms/step', {k:round(v['device_ms'],1) for k,v in (d.get('kernels_isolated') or {}).items()})"
done
