#!/bin/bash
# On the GPU box: (1) the team form of the existence kernel with its per-round counters, (2) the repeat-rich bench with wave-to-wave hand-over,
# (3) the CLI end to end with the default BGZF encoder. usage: bash scripts/gpu_mix1.sh <tag>
T=${1:-mix1}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
for team in 0 1; do
  e="FLX_EXISTS_LANES=1"; [ $team == 1 ] || e="FLX_EXISTS_LANES=1 FLX_EXISTS_TEAM=1"
  env $e FLX_ALIGN_DEBUG=1 timeout -k 10 400 python3 $R/bench.py --isolated-only --no-cpu-baseline > $O/k3_team$team.json 2> $O/k3_team$team.err || { tail -5 $O/k3_team$team.err; exit 1; }
  echo "== team form $team (one-lane pass, last chunk's rounds)"; grep "exists round" $O/k3_team$team.err | tail -9 | cut -c1-260
done
FLX_SEARCH_DEBUG=1 FLX_BENCH_VERBOSE=1 timeout -k 10 500 python3 $R/bench.py --repeat-rich --steps 4 --warmup 2 --no-cpu-baseline --no-host-inputs-leg > $O/rr.json 2> $O/rr.err || { tail -5 $O/rr.err; exit 1; }
python3 -c "
import json
d=json.load(open('$O/rr.json'))
print('repeat-rich', d['value'], 'reads/s', d['ms_per_step'], 'ms/step; isolated', {n: round(v['device_ms'],1) for n,v in d['kernels_isolated'].items()}, 'reruns', d['path']['search_reruns'])"
grep -h "^\[fm_search\]" $O/rr.err | tail -1 | cut -c1-330
bash $R/scripts/cli_throughput2.sh $O/cli_throughput.txt 262144 65536 > $O/cli.log 2>&1 || { tail -5 $O/cli.log; exit 1; }
cat $O/cli_throughput.txt | cut -c1-300
