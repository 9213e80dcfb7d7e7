#!/bin/bash
# On the GPU box: the tests that touch the text walk and the text's padding, then the 20 kb @ 2 % shape (long seeds: the text walk's large LDS windows).
T=${1:-hifi}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
timeout -k 10 600 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -k "search or whole_path or dollar or repeat_rich_reference or baseline_read_shapes or hifi or maximum_length or image or many_references" > $O/tests.log 2>&1
echo "pytest exit $?" >> $O/tests.log
tail -3 $O/tests.log
grep -q "pytest exit 0" $O/tests.log || exit 1
for w in 0 1; do
  FLX_FM_NO_WINDOWS=$w FLX_SEARCH_DEBUG=1 python3 $R/bench.py --config hifi --steps 12 --warmup 3 --no-cpu-baseline --no-host-inputs-leg > $O/b$w.json 2> $O/b$w.err || { tail -3 $O/b$w.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/b$w.json')); print('FLX_FM_NO_WINDOWS=$w:', d['value'], 'reads/s', d['ms_per_step'], 'ms/step', {k:round(v['device_ms'],1) for k,v in d['kernels_isolated'].items()})"
done
