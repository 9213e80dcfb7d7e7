#!/bin/bash
# On the GPU box: the tests that touch the search, then bench.py --repeat-rich with and without the hand-over of subtrees between the
# lanes of a wave. usage: bash scripts/gpu_repeat_ab.sh <tag> [pytest -k expression]
T=${1:-rep}; K=${2:-"search or whole_path or dollar or repeat or seeds_written or config0 or grch38"}
R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
timeout -k 10 800 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -s -k "$K" > $O/tests.log 2>&1
echo "pytest exit $?" >> $O/tests.log
tail -6 $O/tests.log | cut -c1-300
grep -q "pytest exit 0" $O/tests.log || exit 1
run() {
    tag=$1; shift
    env "$@" FLX_SEARCH_DEBUG=1 FLX_BENCH_VERBOSE=1 timeout -k 10 500 python3 $R/bench.py --repeat-rich --steps 4 --warmup 2 --no-cpu-baseline --no-host-inputs-leg > $O/$tag.json 2> $O/$tag.err || { tail -5 $O/$tag.err; return 1; }
    python3 - <<PY
import json
d=json.load(open('$O/$tag.json'))
print('$tag', d['value'], 'reads/s', d['ms_per_step'], 'ms/step; isolated', {n: round(v['device_ms'],1) for n,v in d['kernels_isolated'].items()}, 'reruns', d['path']['search_reruns'], 'host-selected', d['path']['seeds_selected_on_host'])
PY
    grep -h "^\[fm_search\]" $O/$tag.err | tail -1 | cut -c1-330
}
run mailboxes FLX_FM_NO_MAILBOXES=0 || exit 1
exit 0
# the text walk with and without its LDS windows, one-lane pass on the uniform reference
for w in 0; do
  FLX_FM_NO_WINDOWS=$w FLX_SEARCH_DEBUG=1 timeout -k 10 400 python3 $R/bench.py --isolated-only --no-cpu-baseline > $O/iso_nowin$w.json 2> $O/iso_nowin$w.err || { tail -5 $O/iso_nowin$w.err; exit 1; }
  python3 -c "
import json
d=json.load(open('$O/iso_nowin$w.json'))
print('uniform one-lane, FLX_FM_NO_WINDOWS=$w:', {n: round(v['device_ms'],1) for n,v in d['kernels_isolated'].items()})"
done
# the uniform reference (the metric's configuration) with the same build
FLX_BENCH_VERBOSE=1 timeout -k 10 500 python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-repeat-rich-leg > $O/uniform.json 2> $O/uniform.err || { tail -5 $O/uniform.err; exit 1; }
python3 - <<PY
import json
d=json.load(open('$O/uniform.json'))
print('uniform', d['value'], 'reads/s', d['ms_per_step'], 'ms/step; host inputs', d['value_host_inputs'], '; isolated', {n: round(v['device_ms'],1) for n,v in d['kernels_isolated'].items()}, 'roofline', d['roofline']['frac'], d['roofline']['achieved'])
PY
