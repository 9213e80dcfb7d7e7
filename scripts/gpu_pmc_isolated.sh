#!/bin/bash
# On the GPU box: SQ counters of every kernel of the one-lane pass (16384 reads of the metric's configuration), per kernel family.
# usage: bash scripts/gpu_pmc_isolated.sh <tag> [bench flags]
T=${1:-pmc}; shift
R=/root/repo
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/p1 -o p1 --output-format csv -- python3 $R/bench.py --isolated-only --no-cpu-baseline "$@" > $R/gpurun_out/${T}_iso.json 2> /dev/null
python3 $R/scripts/pmc_summary.py "$(find /tmp/p1 -name '*counter_collection.csv' | head -1)" > $R/gpurun_out/${T}_sq.txt
grep -E "dispatches|SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_ANY" $R/gpurun_out/${T}_sq.txt | grep -A5 -E "fm_search|ed_exists|ed_band|traceback|seed_select|vr2_" | cut -c1-150
