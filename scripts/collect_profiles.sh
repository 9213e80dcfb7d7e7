#!/bin/bash
# Collects the round's profile set on the GPU box into gpurun_out/prof (copy what is to be judged into profiles/).
# usage: bash scripts/collect_profiles.sh <prefix>      e.g. r01_e
set -o pipefail
P=${1:-r01_x}
ROOT=/root/repo
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats_csv() { find $1 -name "*kernel_stats.csv" | head -1; }
pmc_csv() { find $1 -name "*counter_collection.csv" | head -1; }

python3 $ROOT/bench.py > $OUT/${P}_bench_default.json || exit 1
echo "default done"
python3 $ROOT/bench.py --interval-optimization --no-cpu-baseline > $OUT/${P}_bench_interval_optimization.json || exit 1
echo "-I done"
rm -rf /tmp/k1 && rocprofv3 --kernel-trace --stats -d /tmp/k1 -o k1 --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/${P}_bench_default_under_rocprof.json || exit 1
cp "$(stats_csv /tmp/k1)" $OUT/${P}_bench_default_kernel_stats.csv
echo "rocprof default done"
rm -rf /tmp/k2 && rocprofv3 --kernel-trace --stats -d /tmp/k2 -o k2 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline > $OUT/${P}_bench_isolated.json || exit 1
cp "$(stats_csv /tmp/k2)" $OUT/${P}_bench_isolated_kernel_stats.csv
echo "rocprof isolated done"
rm -rf /tmp/k3 && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/k3 -o k3 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline > /tmp/iso_f.json || exit 1
echo "FETCH_SIZE pass done"
rm -rf /tmp/k4 && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/k4 -o k4 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline > /tmp/iso_w.json || exit 1
echo "WRITE_SIZE pass done"
python3 $ROOT/scripts/make_traffic_json.py "$(pmc_csv /tmp/k3)" "$(pmc_csv /tmp/k4)" $OUT/${P}_bench_isolated.json $OUT $P
rm -rf /tmp/k5 && rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d /tmp/k5 -o k5 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline > /dev/null || exit 1
python3 $ROOT/scripts/pmc_summary.py "$(pmc_csv /tmp/k5)" > $OUT/${P}_pmc_sq_pass1.txt
rm -rf /tmp/k6 && rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE TCC_HIT_sum --kernel-trace -d /tmp/k6 -o k6 --output-format csv -- python3 $ROOT/bench.py --isolated-only --no-cpu-baseline > /dev/null || exit 1
python3 $ROOT/scripts/pmc_summary.py "$(pmc_csv /tmp/k6)" > $OUT/${P}_pmc_sq_pass2.txt
echo "SQ passes done"
ls -la $OUT
