#!/bin/bash
# SQ instruction counters of fm_search on the default batch (one lane). Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d /tmp/p1 -o p1 --output-format csv -- python3 /root/repo/scripts/search_only.py > /dev/null 2>&1
f=$(find /tmp/p1 -name "*counter_collection.csv" | head -1)
python3 /root/repo/scripts/pmc_summary.py $f | grep -A8 fm_search
