#!/bin/bash
# On the GPU box: bench.py's timed region under a list of environment settings, one line each.
# usage: bash scripts/gpu_sweep.sh "<bench flags>" "VAR=a VAR2=b" "VAR=c" ...
R=/root/repo
FLAGS=$1; shift
mkdir -p $R/gpurun_out/sweep
i=0
for setting in "$@"; do
    i=$((i + 1))
    # (the line goes through a file: quotes or backslashes in it must not end up in Python source)
    env $setting timeout -k 10 240 python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-isolated-pass --no-host-inputs-leg --no-repeat-rich-leg $FLAGS > $R/gpurun_out/sweep/$i.json 2> $R/gpurun_out/sweep/$i.err
    python3 - "$setting" $R/gpurun_out/sweep/$i.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print(sys.argv[1] + ":", d["value"], "reads/s", d["ms_per_step"], "ms/step")
except Exception as e:
    print(sys.argv[1] + ": failed", e)
PY
done
