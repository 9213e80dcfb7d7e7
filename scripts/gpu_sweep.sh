#!/bin/bash
# On the GPU box: bench.py's timed region under a list of environment settings, one line each.
# usage: bash scripts/gpu_sweep.sh "<bench flags>" "VAR=a VAR2=b" "VAR=c" ...
R=/root/repo
FLAGS=$1; shift
for setting in "$@"; do
    out=$(env $setting timeout -k 10 240 python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-isolated-pass $FLAGS 2>/dev/null)
    python3 -c "
import json,sys
try:
    d=json.loads('''$out''')
    print('$setting:', d['value'], 'reads/s', d['ms_per_step'], 'ms/step')
except Exception as e:
    print('$setting: failed', e)
"
done
