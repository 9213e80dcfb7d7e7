#!/bin/bash
# Counter passes over fm_search (K1) alone on a reference far larger than the caches. Run on the GPU box from the repo root:
#   bash scripts/pmc_k1.sh <out_dir> <genome_bp> <n_reads> <read_len>
OUT=$(realpath $1); G=${2:-1000000000}; NR=${3:-4096}; L=${4:-5000}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
    rm -rf /tmp/$1
    rocprofv3 --pmc $2 --kernel-trace -d /tmp/$1 -o $1 --output-format csv -- python3 /root/repo/scripts/search_scale.py $G $NR $L > $OUT/$1.log 2>&1 || return 1
    f=$(find /tmp/$1 -name "*counter_collection.csv" | head -1)
    python3 /root/repo/scripts/pmc_summary.py $f | grep -A12 fm_search > $OUT/$1.txt
    cat $OUT/$1.txt
}
run q1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" &&
run q2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" &&
run q3 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" &&
run q4 "FETCH_SIZE GRBM_GUI_ACTIVE" &&
run q5 "WRITE_SIZE TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_TA_BUSY_sum"
