#!/bin/bash
# Collects the round's profile set on the GPU box into gpurun_out/prof (copy what is to be judged into profiles/).
# usage: bash scripts/collect_profiles.sh <prefix> [extra bench.py flags]     e.g. r03
# The counter passes profile one command (`python3 bench.py --isolated-only --no-cpu-baseline`: the first timed batch on one lane,
# kernels not overlapped) three ways: kernel trace + stats, FETCH_SIZE, WRITE_SIZE (separate runs, as the MI355X guide prescribes),
# plus one SQ counter pass. rocprofv3 gets the interpreter itself after `--`.
set -o pipefail
P=${1:-r03}; shift
ROOT=/root/repo
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats_csv() { find $1 -name "*kernel_stats.csv" | head -1; }
pmc_csv() { find $1 -name "*counter_collection.csv" | head -1; }

timeout -k 10 600 python3 $ROOT/bench.py --steps 20 --warmup 5 "$@" > $OUT/${P}_bench_default.json 2> $OUT/${P}_bench_default.err || exit 1
python3 - <<PY
import json
d = json.load(open("$OUT/${P}_bench_default.json"))
c = d["config"]; cb = d.get("cpu_baseline") or {}
if "cursor_extensions_per_read" in cb:
    key = f"{c['reference_symbols']}/{c['reference_sequences']}/{c['read_length']}/{c['error_rate']}/{int(bool(c.get('repeat_rich')))}" if 'reference_symbols' in c else None
    print("oracle extensions per read:", cb["cursor_extensions_per_read"], "key", key)
print("default:", d["value"], d["unit"], d["ms_per_step"], "ms/step; roofline", d.get("roofline"))
PY
echo "default done"
timeout -k 10 400 python3 $ROOT/bench.py --steps 20 --warmup 5 --interval-optimization --no-cpu-baseline --no-isolated-pass "$@" > $OUT/${P}_bench_interval_optimization.json 2> /dev/null || exit 1
echo "-I done"
rm -rf /tmp/k2 && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/k2 -o k2 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > $OUT/${P}_bench_isolated.json 2> /dev/null || exit 1
cp "$(stats_csv /tmp/k2)" $OUT/${P}_bench_isolated_kernel_stats.csv
echo "rocprof isolated done"
rm -rf /tmp/k3 && timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/k3 -o k3 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > /tmp/iso_f.json 2> /dev/null || exit 1
echo "FETCH_SIZE pass done"
rm -rf /tmp/k4 && timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/k4 -o k4 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > /tmp/iso_w.json 2> /dev/null || exit 1
echo "WRITE_SIZE pass done"
python3 $ROOT/scripts/make_traffic_json.py "$(pmc_csv /tmp/k3)" "$(pmc_csv /tmp/k4)" $OUT/${P}_bench_isolated.json $OUT $P
rm -rf /tmp/k5 && timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d /tmp/k5 -o k5 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > /dev/null 2>&1 || exit 1
python3 $ROOT/scripts/pmc_summary.py "$(pmc_csv /tmp/k5)" > $OUT/${P}_pmc_sq_pass1.txt
echo "counter passes done"
ls -la $OUT
