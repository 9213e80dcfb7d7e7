#!/bin/bash
# On the GPU box: the one-lane bench pass (search counters on stderr) under a few settings of the K1 knobs; prints fm_search ms per 16384 reads.
R=/root/repo
mkdir -p $R/gpurun_out
run() {
    tag=$1; shift
    env "$@" FLX_SEARCH_DEBUG=1 timeout -k 10 300 python3 $R/bench.py --isolated-only --no-cpu-baseline > $R/gpurun_out/ab_$tag.json 2> $R/gpurun_out/ab_$tag.err
    python3 - <<PY
import json
d=json.load(open('$R/gpurun_out/ab_$tag.json'))
k=d['kernels_isolated']
print('$tag', {n: round(v['device_ms'],2) for n,v in k.items() if n in ('fm_search','fm_select')})
PY
    grep -h "fm_search filtered" $R/gpurun_out/ab_$tag.err | tail -1 | cut -c1-260
}
run looks2 FLX_FM_LOOKS=2
run looks1 FLX_FM_LOOKS=1
run textwaves4k FLX_FM_TEXT_WAVES=4096
run textwaves16k FLX_FM_TEXT_WAVES=16384
run spw128 FLX_FM_SEEDS_PER_WAVE=128
run textmin4 FLX_FM_TEXT_MIN=4
