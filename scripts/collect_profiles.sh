#!/bin/bash
# Collects the round's profile set on the GPU box into gpurun_out/prof (copy what is to be judged into profiles/).
# usage: bash scripts/collect_profiles.sh <prefix> [extra bench.py flags]     e.g. r04
# The counter passes profile one command (`python3 bench.py --isolated-only --no-cpu-baseline`: the first timed batch on one lane,
# kernels not overlapped) three ways: kernel trace + stats, FETCH_SIZE, WRITE_SIZE (separate runs, as the MI355X guide prescribes),
# plus one SQ counter pass. rocprofv3 gets the interpreter itself after `--`. Every pass keeps its stderr in a file of its own.
set -o pipefail
P=${1:-r04}; shift
ROOT=/root/repo
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats_csv() { find $1 -name "*kernel_stats.csv" | head -1; }
pmc_csv() { find $1 -name "*counter_collection.csv" | head -1; }
fail() { echo "FAILED: $1"; tail -5 "$2"; exit 1; }

# FLX_PROFILE_PHASE=a: the two bench lines only; b: the rocprofv3 passes only (a gpurun call is at most 20 minutes); unset: both
PHASE=${FLX_PROFILE_PHASE:-ab}
if [[ $PHASE == *a* ]]; then
# the driver's command: the metric's configuration with the oracle leg (cpu_baseline + parity_sample), the host-inputs leg and the repeat-rich leg
timeout -k 10 900 python3 $ROOT/bench.py --steps 20 --warmup 5 "$@" > $OUT/${P}_bench_default.json 2> $OUT/${P}_bench_default.err || fail "default bench" $OUT/${P}_bench_default.err
python3 - <<PY
import json
d = json.load(open("$OUT/${P}_bench_default.json"))
print("default:", d["value"], d["unit"], d["ms_per_step"], "ms/step; host inputs", d.get("value_host_inputs"), "; parity", d.get("parity_sample"))
print("  roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "avg_launch_ms")}, "fm_search", {k: d["roofline_fm_search"][k] for k in ("achieved", "frac", "traffic", "avg_launch_ms", "reference_walk_equivalent_GBps") if k in d["roofline_fm_search"]})
print("  cpu_baseline", d.get("cpu_baseline", {}).get("value"), "repeat_rich", (d.get("repeat_rich") or {}).get("value"), (d.get("repeat_rich") or {}).get("error"))
PY
timeout -k 10 400 python3 $ROOT/bench.py --steps 20 --warmup 5 --interval-optimization --no-cpu-baseline --no-isolated-pass --no-repeat-rich-leg --no-host-inputs-leg "$@" > $OUT/${P}_bench_interval_optimization.json 2> $OUT/${P}_bench_interval_optimization.err || fail "-I bench" $OUT/${P}_bench_interval_optimization.err
echo "-I done"
fi
[[ $PHASE == *b* ]] || { ls -la $OUT; exit 0; }
rm -rf /tmp/k2 && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/k2 -o k2 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > $OUT/${P}_bench_isolated.json 2> $OUT/${P}_bench_isolated.err || fail "isolated pass under rocprofv3" $OUT/${P}_bench_isolated.err
F=$(stats_csv /tmp/k2); [ -n "$F" ] && cp "$F" $OUT/${P}_bench_isolated_kernel_stats.csv || fail "no kernel stats csv" $OUT/${P}_bench_isolated.err
echo "rocprof isolated done"
rm -rf /tmp/k3 && timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/k3 -o k3 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > /tmp/iso_f.json 2> $OUT/${P}_pmc_fetch.err || fail "FETCH_SIZE pass" $OUT/${P}_pmc_fetch.err
rm -rf /tmp/k4 && timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/k4 -o k4 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > /tmp/iso_w.json 2> $OUT/${P}_pmc_write.err || fail "WRITE_SIZE pass" $OUT/${P}_pmc_write.err
python3 $ROOT/scripts/make_traffic_json.py "$(pmc_csv /tmp/k3)" "$(pmc_csv /tmp/k4)" $OUT/${P}_bench_isolated.json $OUT $P
python3 $ROOT/scripts/pmc_by_symbol.py "$(pmc_csv /tmp/k3)" FETCH_SIZE fm_search seed_ hit_scatter ed_ vr2 lastrow > $OUT/${P}_pmc_fetch_by_symbol.txt
python3 $ROOT/scripts/pmc_by_symbol.py "$(pmc_csv /tmp/k4)" WRITE_SIZE fm_search seed_ hit_scatter ed_ vr2 lastrow > $OUT/${P}_pmc_write_by_symbol.txt
rm -rf /tmp/k5 && timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d /tmp/k5 -o k5 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > /dev/null 2> $OUT/${P}_pmc_sq.err || fail "SQ pass" $OUT/${P}_pmc_sq.err
python3 $ROOT/scripts/pmc_summary.py "$(pmc_csv /tmp/k5)" > $OUT/${P}_pmc_sq_pass1.txt
rm -f $OUT/${P}_pmc_fetch.err $OUT/${P}_pmc_write.err $OUT/${P}_pmc_sq.err
echo "counter passes done"
ls -la $OUT
