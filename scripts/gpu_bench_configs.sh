#!/bin/bash
# On the GPU box: one bench line per BASELINE.json configuration, without and with -I, plus the repeat-rich GRCh38-size reference.
# usage: bash scripts/gpu_bench_configs.sh <tag>      -> gpurun_out/<tag>_bench_<config>[_I].json
T=${1:-r04}
R=/root/repo
mkdir -p $R/gpurun_out
run() {
    name=$1; shift
    timeout -k 10 420 python3 $R/bench.py --steps 12 --warmup 3 --no-host-inputs-leg --no-repeat-rich-leg "$@" > $R/gpurun_out/${T}_bench_${name}.json 2> $R/gpurun_out/${T}_bench_${name}.err
    python3 -c "
import json
try:
    d=json.load(open('$R/gpurun_out/${T}_bench_${name}.json'))
    p=d.get('path') or {}
    print('$name:', d['value'], 'reads/s', d['ms_per_step'], 'ms/step; records/read', round(d['records']/max(1,p.get('reads',1)),2), 'excluded seeds', p.get('seeds_excluded_by_hard_cap'), 'of', p.get('seeds'), 'host-selected', p.get('seeds_selected_on_host'))
except Exception as e:
    print('$name: failed', e)
"
}
for cfg in ecoli chr1 hifi; do
    run $cfg --config $cfg --no-cpu-baseline
    run ${cfg}_I --config $cfg --no-cpu-baseline --interval-optimization
done
run grch38_I --config grch38 --no-cpu-baseline --interval-optimization
run repeat_rich --config grch38 --repeat-rich --cpu-sample 128
run repeat_rich_I --config grch38 --repeat-rich --no-cpu-baseline --interval-optimization
