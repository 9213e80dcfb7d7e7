#!/bin/bash
# On the GPU box: the BAM writer alone on synthetic records, then the CLI end to end at default flags with a look into its output.
T=${1:-cli}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
python3 $R/scripts/bam_ratio_check.py 16 | tee $O/bam_ratio_check.txt
W=/tmp/flx_cli_peek; rm -rf $W; mkdir -p $W
BIN=$R/floxer_amd
$BIN/simulated_dataset create --genomes $W/g.fasta --reads $W/r.fastq -c 50000000 -n 5 -l 10000 -m 16384 -e 0.08 -s 7 --revcomp-fraction 0.5 > /dev/null 2>&1
FLX_CLI_PROFILE=1 $BIN/floxer --reference $W/g.fasta --queries $W/r.fastq --output $W/o.bam --error-probability 0.08 --threads 16 2> $O/peek.err
grep "flx cli profile\|finished aligning" $O/peek.err | cut -c1-300
ls -l $W/o.bam
python3 $R/scripts/bam_peek.py $W/o.bam 100 > $O/bam_peek.txt
head -60 $O/bam_peek.txt
rm -rf $W
