#!/bin/bash
# End-to-end rate of the drop-in CLI (FASTQ in -> BAM out) beside bench.py's resident-reads number. Run on the GPU box from the repo
# root:   bash scripts/cli_throughput.sh <out_dir> [chromosome_length] [chromosomes] [reads] [read_length]
# Makes a synthetic data set with simulated_dataset create, saves the index once, then times
#   floxer --queries reads.fastq     --output out.bam   (plain FASTQ; 16 I/O threads)
#   floxer --queries reads.fastq.gz  --output out.bam   (gzip: one inflate stream)
#   floxer ... --output out.sam, and the same with --interval-optimization
# and checks the BAM with simulated_dataset verify.
OUT=$(realpath $1); C=${2:-50000000}; N=${3:-5}; M=${4:-65536}; L=${5:-10000}
mkdir -p $OUT
W=/tmp/flx_cli_tp; rm -rf $W; mkdir -p $W
BIN=/root/repo/floxer_amd
set -e
$BIN/simulated_dataset create --genomes $W/g.fasta --reads $W/r.fastq -c $C -n $N -l $L -m $M -e 0.08 -s 7 --revcomp-fraction 0.5
gzip -1 -k $W/r.fastq
ls -l $W > $OUT/cli_throughput.txt
run() {   # name, queries, output, extra flags
    local t0=$(date +%s.%N)
    $BIN/floxer --reference $W/g.fasta --queries $2 --output $3 --error-probability 0.08 --index $W/g.index --threads 16 $4 2> $W/$1.err
    local t1=$(date +%s.%N)
    local align=$(grep -o "finished aligning successfully in [0-9.]* seconds" $W/$1.err | grep -o "[0-9.]*" | head -1)
    python3 -c "print('$1: wall %.1f s, aligning phase %.2f s -> %.0f reads/s end to end' % ($t1 - $t0, $align, $M / $align), '$(grep -o "([0-9]* queries, [0-9]* records)" $W/$1.err)')" | tee -a $OUT/cli_throughput.txt
}
run index_build_and_first_run $W/r.fastq $W/o0.bam ""
run fastq_to_bam $W/r.fastq $W/o1.bam ""
run fastq_gz_to_bam $W/r.fastq.gz $W/o2.bam ""
run fastq_to_sam $W/r.fastq $W/o3.sam ""
run fastq_to_bam_interval_optimization $W/r.fastq $W/o4.bam "--interval-optimization"
FLX_BGZF_LEVEL=6 run fastq_to_bam_zlib_level_6 $W/r.fastq $W/o5.bam ""
cmp $W/o1.bam $W/o2.bam && echo "plain and gz input give the same BAM" | tee -a $OUT/cli_throughput.txt
$BIN/simulated_dataset verify --alignments $W/o4.bam -p $((L / 10)) > $W/verify.txt 2> $W/verify.err
echo "accuracy (-I run): $(grep -c FoundOptimal $W/verify.txt) of $M FoundOptimal; $(cat $W/verify.err)" | tee -a $OUT/cli_throughput.txt
ls -l $W/*.bam $W/*.sam >> $OUT/cli_throughput.txt
rm -rf $W
