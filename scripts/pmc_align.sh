#!/bin/bash
# SQ counters of the alignment kernels on the default batch (one lane). Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
run() { rm -rf /tmp/$1; rocprofv3 --pmc $2 --kernel-trace -d /tmp/$1 -o $1 --output-format csv -- python3 /root/repo/scripts/search_only.py > /dev/null 2>&1; f=$(find /tmp/$1 -name "*counter_collection.csv" | head -1); python3 /root/repo/scripts/pmc_summary.py $f | grep -A9 "true>\|traceback"; }
run a1 "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
run a2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
