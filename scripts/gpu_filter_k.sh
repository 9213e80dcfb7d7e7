#!/bin/bash
# On the GPU box: the presence filter's K (table of 4^K bits: 8.6 GB at 18, 2.1 GB at 17, 0.5 GB at 16, 134 MB at 15 - inside the 256-MB
# infinity cache) against the default, on the metric's workload: one-lane fm_search and the pipeline.
T=${1:-filterk}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
for k in ${2:-18 17 16 15}; do
  FLX_FILTER_K=$k FLX_SEARCH_DEBUG=1 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-inputs-leg --no-repeat-rich-leg > $O/k$k.json 2> $O/k$k.err || { tail -3 $O/k$k.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/k$k.json')); print('FLX_FILTER_K=$k:', d['value'], 'reads/s', d['ms_per_step'], 'ms/step', {k:round(v['device_ms'],1) for k,v in d['kernels_isolated'].items()}, 'cursor extensions', d['path']['cursor_extensions'])"
  grep -m1 "fm_search filtered" $O/k$k.err | cut -c1-260
done
