#!/bin/bash
# End-to-end rate of the drop-in CLI at a size where start-up does not dominate, with the stage profile (FLX_CLI_PROFILE).
#   bash scripts/cli_throughput2.sh <out file> [reads for the -I run] [reads for the default-flags run]
OUT=$(realpath $1); M=${2:-262144}; MD=${3:-65536}
W=/tmp/flx_cli_tp2; rm -rf $W; mkdir -p $W
BIN=/root/repo/floxer_amd
set -e
$BIN/simulated_dataset create --genomes $W/g.fasta --reads $W/r.fastq -c 50000000 -n 5 -l 10000 -m $M -e 0.08 -s 7 --revcomp-fraction 0.5
head -n $((4 * MD)) $W/r.fastq > $W/rd.fastq
ls -l $W > $OUT
run() {   # name, reads, queries, output, extra flags
    local t0=$(date +%s.%N)
    FLX_CLI_PROFILE=1 $BIN/floxer --reference $W/g.fasta --queries $3 --output $4 --error-probability 0.08 --index $W/g.index --threads 16 $5 2> $W/$1.err
    local t1=$(date +%s.%N)
    local align=$(grep -o "finished aligning successfully in [0-9.]* seconds" $W/$1.err | grep -o "[0-9.]*" | head -1)
    python3 -c "print('$1: wall %.1f s, aligning phase %.2f s -> %.0f reads/s end to end' % ($t1 - $t0, $align, $2 / $align), '$(grep -o "([0-9]* queries, [0-9]* records)" $W/$1.err)')" | tee -a $OUT
    grep "flx cli profile" $W/$1.err | tee -a $OUT
}
run index_build $MD $W/rd.fastq $W/o0.bam "--interval-optimization"
run fastq_to_bam_I $M $W/r.fastq $W/o1.bam "--interval-optimization"
run fastq_to_sam_I $M $W/r.fastq $W/o1.sam "--interval-optimization"
run fastq_to_bam_default $MD $W/rd.fastq $W/o2.bam ""
ls -l $W/*.bam $W/*.sam >> $OUT
rm -rf $W
