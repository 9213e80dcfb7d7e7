#!/bin/bash
# On the GPU box: the CLI end to end, then the residency study. usage: bash scripts/gpu_mix2.sh <tag>
T=${1:-mix2}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
bash $R/scripts/cli_throughput2.sh $O/cli_throughput.txt 262144 65536 > $O/cli.log 2>&1 || { tail -5 $O/cli.log; exit 1; }
cat $O/cli_throughput.txt | cut -c1-300
bash $R/scripts/gpu_residency.sh $T
