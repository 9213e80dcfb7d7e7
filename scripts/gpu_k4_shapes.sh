#!/bin/bash
# On the GPU box: the root alignments (K4, ed_trace_block_kernel<W>) under forced launch shapes "W,R" (words per lane, lanes per job): the
# one-lane pass's ed_align_trace time. (The shape is forced on the existence tests as well where it holds a job: their time is not the point here.)
T=${1:-k4}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
for sh in default 3,32 4,32 2,64 6,16 8,16; do
  e="FLX_DUMMY=1"; [ $sh == default ] || e="FLX_FORCE_SHAPE=$sh"
  env $e timeout -k 10 400 python3 $R/bench.py --isolated-only --no-cpu-baseline > $O/shape_$sh.json 2> $O/shape_$sh.err || { tail -3 $O/shape_$sh.err; continue; }
  python3 -c "
import json
d=json.load(open('$O/shape_$sh.json'))
k=d['kernels_isolated']
print('shape $sh:', {n: (round(v['device_ms'],1), v['launches']) for n,v in k.items() if n in ('ed_align_trace','ed_align_exists','ed_traceback')})"
done
