#!/bin/bash
# On the GPU box: K1 parity tests, then the one-lane bench pass three times: kernel trace + stats (search counters on stderr), FETCH_SIZE,
# WRITE_SIZE (separate passes), with the traffic per kernel symbol. usage: bash scripts/gpu_k1_traffic.sh <tag> [ENV=VALUE ...] [-- bench flags]
T=${1:-k1}; shift
R=/root/repo
O=$R/gpurun_out/$T
mkdir -p $O
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for e in "${ENVS[@]}"; do export "$e"; done
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 560 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -k "search or whole_path or dollar or repeat or seeds_written" > $O/tests.log 2>&1
echo "pytest exit $?" >> $O/tests.log
tail -3 $O/tests.log
grep -q "pytest exit 0" $O/tests.log || exit 1
fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k2 && FLX_SEARCH_DEBUG=1 timeout -k 10 420 rocprofv3 --kernel-trace --stats -d /tmp/k2 -o k2 --output-format csv -- python3 $R/bench.py --isolated-only --no-cpu-baseline "$@" > $O/iso.json 2> $O/iso.err || { tail -5 $O/iso.err; exit 1; }
cp "$(find /tmp/k2 -name '*kernel_stats.csv' | head -1)" $O/kernel_stats.csv
grep -h "fm_search" $O/iso.err | tail -2 | cut -c1-400
head -12 $O/kernel_stats.csv | cut -c1-160
rm -rf /tmp/k3 && timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/k3 -o k3 --output-format csv -- python3 $R/bench.py --isolated-only --no-cpu-baseline "$@" > $O/iso_f.json 2> $O/iso_f.err || { tail -5 $O/iso_f.err; exit 1; }
python3 $R/scripts/pmc_by_symbol.py "$(find /tmp/k3 -name '*counter_collection.csv' | head -1)" FETCH_SIZE fm_search seed_ hit_scatter ed_ vr2 > $O/fetch_by_symbol.txt
rm -rf /tmp/k4 && timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/k4 -o k4 --output-format csv -- python3 $R/bench.py --isolated-only --no-cpu-baseline "$@" > $O/iso_w.json 2> $O/iso_w.err || { tail -5 $O/iso_w.err; exit 1; }
python3 $R/scripts/pmc_by_symbol.py "$(find /tmp/k4 -name '*counter_collection.csv' | head -1)" WRITE_SIZE fm_search seed_ hit_scatter ed_ vr2 > $O/write_by_symbol.txt
cat $O/fetch_by_symbol.txt $O/write_by_symbol.txt | cut -c1-200
python3 $R/scripts/make_traffic_json.py "$(find /tmp/k3 -name '*counter_collection.csv' | head -1)" "$(find /tmp/k4 -name '*counter_collection.csv' | head -1)" $O/iso.json $O r04
