#!/bin/bash
# Stall-side counters of fm_search on the default batch (one lane). Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
run() { rm -rf /tmp/$1; rocprofv3 --pmc $2 --kernel-trace -d /tmp/$1 -o $1 --output-format csv -- python3 /root/repo/scripts/search_only.py > /dev/null 2>&1; f=$(find /tmp/$1 -name "*counter_collection.csv" | head -1); python3 /root/repo/scripts/pmc_summary.py $f | grep -A9 fm_search; }
run q1 "SQ_WAIT_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES"
run q2 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run q3 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_CYCLES"
