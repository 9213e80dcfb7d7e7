#!/bin/bash
# Collects the round's profile set on the GPU box into gpurun_out/prof (copy what is to be judged into profiles/).
# usage: bash scripts/collect_profiles.sh <prefix> [extra bench.py flags]     e.g. r02
# Every pass is the same command (`python3 bench.py --isolated-only --no-cpu-baseline`: the first timed batch twice on one lane,
# kernels not overlapped), profiled three ways: kernel trace + stats, FETCH_SIZE, WRITE_SIZE (separate runs, as the MI355X guide
# prescribes), plus two SQ counter passes. rocprofv3 gets the interpreter itself after `--`.
set -o pipefail
P=${1:-r02}; shift
ROOT=/root/repo
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats_csv() { find $1 -name "*kernel_stats.csv" | head -1; }
pmc_csv() { find $1 -name "*counter_collection.csv" | head -1; }

python3 $ROOT/bench.py --steps 20 --warmup 5 "$@" > $OUT/${P}_bench_default.json || exit 1
echo "default done"
python3 $ROOT/bench.py --interval-optimization --no-cpu-baseline "$@" > $OUT/${P}_bench_interval_optimization.json || exit 1
echo "-I done"
rm -rf /tmp/k2 && rocprofv3 --kernel-trace --stats -d /tmp/k2 -o k2 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > $OUT/${P}_bench_isolated.json || exit 1
cp "$(stats_csv /tmp/k2)" $OUT/${P}_bench_isolated_kernel_stats.csv
echo "rocprof isolated done"
rm -rf /tmp/k3 && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/k3 -o k3 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > /tmp/iso_f.json || exit 1
echo "FETCH_SIZE pass done"
rm -rf /tmp/k4 && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/k4 -o k4 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > /tmp/iso_w.json || exit 1
echo "WRITE_SIZE pass done"
python3 $ROOT/scripts/make_traffic_json.py "$(pmc_csv /tmp/k3)" "$(pmc_csv /tmp/k4)" $OUT/${P}_bench_isolated.json $OUT $P
rm -rf /tmp/k5 && rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d /tmp/k5 -o k5 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > /dev/null || exit 1
python3 $ROOT/scripts/pmc_summary.py "$(pmc_csv /tmp/k5)" > $OUT/${P}_pmc_sq_pass1.txt
rm -rf /tmp/k6 && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE TA_TA_BUSY_sum --kernel-trace -d /tmp/k6 -o k6 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline "$@" > /dev/null || exit 1
python3 $ROOT/scripts/pmc_summary.py "$(pmc_csv /tmp/k6)" > $OUT/${P}_pmc_mem_pass2.txt
echo "counter passes done"
# calibration of FETCH_SIZE on K1's access shape: random 64-byte blocks, one or two 16-byte loads each, known block count
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value $ROOT/scripts/micro/gather_cost.hip -o /tmp/gather_cost || exit 1
/tmp/gather_cost 4000000000 > $OUT/${P}_gather_cost.txt
rm -rf /tmp/k7 && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/k7 -o k7 --output-format csv -- /tmp/gather_cost 4000000000 calib > $OUT/${P}_gather_calib.txt || exit 1
python3 $ROOT/scripts/pmc_summary.py "$(pmc_csv /tmp/k7)" >> $OUT/${P}_gather_calib.txt
ls -la $OUT
