#!/bin/bash
# On the GPU box: the two HBM-traffic counter passes (FETCH_SIZE, WRITE_SIZE: separate runs) over the one-lane pass and the per-kernel
# summary bench.py reads (profiles/<tag>_pmc_traffic_<kernel>.json). usage: bash scripts/gpu_pmc_traffic.sh <tag> [bench flags]
P=${1:-r03}; shift
ROOT=/root/repo
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pmc_csv() { find $1 -name "*counter_collection.csv" | head -1; }
timeout -k 10 300 python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > $OUT/${P}_bench_isolated.json 2> /dev/null || exit 1
rm -rf /tmp/k3 && timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/k3 -o k3 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > /tmp/iso_f.json 2> /dev/null || exit 1
rm -rf /tmp/k4 && timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/k4 -o k4 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > /tmp/iso_w.json 2> /dev/null || exit 1
python3 $ROOT/scripts/make_traffic_json.py "$(pmc_csv /tmp/k3)" "$(pmc_csv /tmp/k4)" $OUT/${P}_bench_isolated.json $OUT $P
