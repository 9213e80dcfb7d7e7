#!/bin/bash
# On the GPU box: the existence-kernel parity tests (ring, lane, team forms), then the one-lane bench pass with the ring form (default), the
# lane form with one lane per job, and the lane form with teams; and the pipeline with the best. usage: bash scripts/gpu_k3_team.sh <tag>
T=${1:-k3}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
timeout -k 10 800 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -k "existence or align_batch or whole_path or repeat_rich_reference or baseline_read_shapes or full_size_reads" > $O/tests.log 2>&1
echo "pytest exit $?" >> $O/tests.log
tail -5 $O/tests.log | cut -c1-300
grep -q "pytest exit 0" $O/tests.log || exit 1
run() {
    tag=$1; shift
    env "$@" FLX_BENCH_VERBOSE=1 timeout -k 10 400 python3 $R/bench.py --isolated-only --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err || { tail -5 $O/$tag.err; return 1; }
    python3 -c "
import json
d=json.load(open('$O/$tag.json'))
print('$tag one-lane:', {n: round(v['device_ms'],1) for n,v in d['kernels_isolated'].items()})"
}
run ring FLX_DUMMY=0 && run lanes1 FLX_EXISTS_LANES=1 FLX_EXISTS_TEAM=1 && run teams FLX_EXISTS_LANES=1 && run teams4 FLX_EXISTS_LANES=1 FLX_EXISTS_TEAM=4 && run teams16 FLX_EXISTS_LANES=1 FLX_EXISTS_TEAM=16 || exit 1
for f in ring teams; do
  e=""; [ $f == teams ] && e="FLX_EXISTS_LANES=1"
  env $e timeout -k 10 400 python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-repeat-rich-leg --no-isolated-pass --no-host-inputs-leg > $O/pipe_$f.json 2> $O/pipe_$f.err || { tail -5 $O/pipe_$f.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/pipe_$f.json'))
print('pipeline $f:', d['value'], 'reads/s', d['ms_per_step'], 'ms/step')"
done
