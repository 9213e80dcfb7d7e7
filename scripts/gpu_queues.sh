#!/bin/bash
# On the GPU box: lanes against hardware queues (the library asks for GPU_MAX_HW_QUEUES=16 unless the variable is set): do 20 or 24 lanes lose
# because they share 16 queues?
T=${1:-queues}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
CFGS=${2:-16,16 20,24 24,32 16,32 24,24}
for cfg in $CFGS; do
  l=${cfg%,*}; q=${cfg#*,}
  FLX_LANES=$l GPU_MAX_HW_QUEUES=$q python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-host-inputs-leg --no-repeat-rich-leg --no-isolated-pass > $O/l${l}_q$q.json 2> $O/l${l}_q$q.err || { tail -3 $O/l${l}_q$q.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/l${l}_q$q.json')); print('lanes $l, hardware queues $q:', d['value'], 'reads/s', d['ms_per_step'], 'ms/step')"
done
