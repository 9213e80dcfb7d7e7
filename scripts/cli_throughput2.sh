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
    FLX_CLI_PROFILE=1 FLX_WRITER_PROFILE=1 $BIN/floxer --reference $W/g.fasta --queries $3 --output $4 --error-probability 0.08 --index $W/g.index --threads 16 $5 2> $W/$1.err
    local t1=$(date +%s.%N)
    local align=$(grep -o "finished aligning successfully in [0-9.]* seconds" $W/$1.err | grep -o "[0-9.]*" | head -1)
    python3 -c "print('$1: wall %.1f s, aligning phase %.2f s -> %.0f reads/s end to end' % ($t1 - $t0, $align, $2 / $align), '$(grep -o "([0-9]* queries, [0-9]* records)" $W/$1.err)')" | tee -a $OUT
    grep "flx cli profile\|flx writer profile" $W/$1.err | tee -a $OUT
}
run index_build $MD $W/rd.fastq $W/o0.bam "--interval-optimization"
run fastq_to_bam_I $M $W/r.fastq $W/o1.bam "--interval-optimization"
run fastq_to_sam_I $M $W/r.fastq $W/o1.sam "--interval-optimization"
run fastq_to_bam_default $MD $W/rd.fastq $W/o2.bam ""
# every BGZF member of the default-flags file through Python's gzip (checks each member's CRC-32 and length), and the BAM records it holds
python3 - $W/o2.bam <<'PY' | tee -a $OUT
import gzip, struct, sys, time
t0 = time.time(); n = 0; recs = 0
with gzip.open(sys.argv[1]) as f:
    head = f.read(8); assert head[:4] == b"BAM\1"
    f.read(struct.unpack("<i", head[4:8])[0])
    n_ref = struct.unpack("<i", f.read(4))[0]
    for _ in range(n_ref):
        l = struct.unpack("<i", f.read(4))[0]; f.read(l + 4)
    while True:
        h = f.read(4)
        if not h: break
        bs = struct.unpack("<i", h)[0]
        body = f.read(bs); assert len(body) == bs
        n += bs + 4; recs += 1
print("default-flags BAM: %d records, %.2f GB of records, every BGZF member's checksum good (%.0f s in Python)" % (recs, n / 1e9, time.time() - t0))
PY
ls -l $W/*.bam $W/*.sam >> $OUT
rm -rf $W
