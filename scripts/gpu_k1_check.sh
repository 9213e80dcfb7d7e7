#!/bin/bash
# On the GPU box: the K1 parity tests, then the one-lane bench pass under rocprofv3 (kernel trace + stats) with the search counters.
# usage: bash scripts/gpu_k1_check.sh <tag>      outputs under gpurun_out/
T=${1:-k1}
R=/root/repo
mkdir -p $R/gpurun_out/prof
timeout -k 10 500 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -k "search or whole_path or dollar or repeat" > $R/gpurun_out/${T}_tests.log 2>&1
echo "pytest exit $?" >> $R/gpurun_out/${T}_tests.log
tail -4 $R/gpurun_out/${T}_tests.log
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/k2
FLX_SEARCH_DEBUG=1 timeout -k 10 420 rocprofv3 --kernel-trace --stats -d /tmp/k2 -o k2 --output-format csv -- python3 $R/bench.py --isolated-only --no-cpu-baseline "${@:2}" > $R/gpurun_out/${T}_iso.json 2> $R/gpurun_out/${T}_iso.err
F=$(find /tmp/k2 -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && cp "$F" $R/gpurun_out/prof/${T}_kernel_stats.csv || { echo "no kernel stats csv"; tail -5 $R/gpurun_out/${T}_iso.err; exit 1; }
grep -h "fm_search" $R/gpurun_out/${T}_iso.err | tail -3
head -14 $R/gpurun_out/prof/${T}_kernel_stats.csv
