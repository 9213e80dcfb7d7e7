#!/bin/bash
# On the GPU box: fewer lanes with larger chunks against 16 lanes of 2048 reads (larger launches fill the chip on their own; fewer kernels share it).
# cfg = lanes,chunk reads,trace arena MB for the whole context
T=${1:-chunks}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
CFGS=${2:-16,2048,65536 12,2048,49152 8,4096,65536 6,4096,49152 4,8192,65536 8,2048,32768}
for cfg in $CFGS; do
  IFS=, read l c a <<< "$cfg"
  FLX_LANES=$l FLX_CHUNK_READS=$c FLX_CHUNK_BASES=$((c * 12000)) FLX_TRACE_ARENA_MB=$a python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-host-inputs-leg --no-repeat-rich-leg --no-isolated-pass > $O/l${l}_c$c.json 2> $O/l${l}_c$c.err || { tail -3 $O/l${l}_c$c.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/l${l}_c$c.json')); print('lanes $l, chunks of $c reads, trace arena $a MB:', d['value'], 'reads/s', d['ms_per_step'], 'ms/step')"
done
