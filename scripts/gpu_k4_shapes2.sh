#!/bin/bash
# On the GPU box: K4's launch shape in the pipeline (FLX_FORCE_SHAPE=W,R reaches choose_align_shape: the root alignments; the rounds' existence
# tests keep their own choice). "-" = the chooser's.
T=${1:-k4shapes2}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
SHAPES=${2:-- 6,8 2,16 4,8 5,16 -}
i=0
for sh in $SHAPES; do
  i=$((i + 1))
  if [ "$sh" = "-" ]; then unset FLX_FORCE_SHAPE; else export FLX_FORCE_SHAPE=$sh; fi
  FLX_ALIGN_DEBUG=1 python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-host-inputs-leg --no-repeat-rich-leg > $O/s$i.json 2> $O/s$i.err || { tail -3 $O/s$i.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/s$i.json')); print('K4 shape $sh:', d['value'], 'reads/s', d['ms_per_step'], 'ms/step', {k:round(v['device_ms'],1) for k,v in (d.get('kernels_isolated') or {}).items()})"
  grep -m1 "ed_align_trace\] unions" $O/s$i.err | cut -c1-160
done
