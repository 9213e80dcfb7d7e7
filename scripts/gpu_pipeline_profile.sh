#!/bin/bash
# On the GPU box: where the timed region's time goes. (1) host phases per chunk (FLX_HOST_PROFILE), (2) kernel concurrency of the timed
# region from a rocprofv3 kernel trace. usage: bash scripts/gpu_pipeline_profile.sh <tag> [bench flags]; outputs under gpurun_out/
T=${1:-pp}; shift
R=/root/repo
mkdir -p $R/gpurun_out
FLX_HOST_PROFILE=1 timeout -k 10 300 python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-isolated-pass "$@" > $R/gpurun_out/${T}_hostprof.json 2> $R/gpurun_out/${T}_hostprof.err
python3 $R/scripts/host_profile_summary.py $R/gpurun_out/${T}_hostprof.err 32 > $R/gpurun_out/${T}_host_profile.txt
python3 -c "import json; d=json.load(open('$R/gpurun_out/${T}_hostprof.json')); print('host-profile run:', d['value'], 'reads/s')" >> $R/gpurun_out/${T}_host_profile.txt
head -40 $R/gpurun_out/${T}_host_profile.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/kt -o kt --output-format csv -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-isolated-pass "$@" > $R/gpurun_out/${T}_traced.json 2> /dev/null
F=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
head -1 $F > $R/gpurun_out/${T}_trace_header.txt; python3 $R/scripts/trace_concurrency.py $F 28 > $R/gpurun_out/${T}_concurrency.txt
python3 -c "import json; d=json.load(open('$R/gpurun_out/${T}_traced.json')); print('traced run:', d['value'], 'reads/s')" >> $R/gpurun_out/${T}_concurrency.txt
cat $R/gpurun_out/${T}_concurrency.txt
