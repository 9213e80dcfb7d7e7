#!/bin/bash
# On the GPU box: the grid of an existence launch (FLX_EXISTS_MAX_WAVES; default 8192) now that its waves hold 117-128 registers (4 per SIMD = 4096 on the chip)
T=${1:-exwaves}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
for w in ${2:-8192 4096 2048 8192 4096 2048}; do
  FLX_EXISTS_MAX_WAVES=$w python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-host-inputs-leg --no-repeat-rich-leg --no-isolated-pass > $O/w$w.json 2> $O/w$w.err || { tail -3 $O/w$w.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/w$w.json')); print('existence launches of at most $w waves:', d['value'], 'reads/s', d['ms_per_step'], 'ms/step')"
done
