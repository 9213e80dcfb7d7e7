#!/bin/bash
# On the GPU box: why more than 16 lanes lose throughput. (1) kernel-trace timelines of the timed region at 16 and at 20 lanes with the
# resident demand of every kernel family (scripts/trace_concurrency.py), (2) the resource-allocation counters of the shader processor input
# (SPI_RA_*: waves that could not be placed for want of wave slots / VGPRs / LDS) per kernel of the one-lane pass, where they count (the
# profiler serialises kernels under --pmc). usage: bash scripts/gpu_residency.sh <tag>
T=${1:-res}; R=/root/repo; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in 16 20; do
  I=3; [ $L == 20 ] && I=4
  rm -rf /tmp/kt$L
  timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/kt$L -o kt --output-format csv -- python3 $R/bench.py --steps 8 --warmup 3 --lanes $L --inflight $I --no-cpu-baseline --no-isolated-pass --no-host-inputs-leg --no-repeat-rich-leg > $O/traced_l$L.json 2> $O/traced_l$L.err || { tail -5 $O/traced_l$L.err; exit 1; }
  F=$(find /tmp/kt$L -name '*kernel_trace.csv' | head -1)
  python3 $R/scripts/trace_concurrency.py $F 28 > $O/concurrency_l$L.txt
  python3 -c "import json; d=json.load(open('$O/traced_l$L.json')); print('traced run, $L lanes:', d['value'], 'reads/s', d['ms_per_step'], 'ms/step')" >> $O/concurrency_l$L.txt
  tail -32 $O/concurrency_l$L.txt
done
rocprofv3 --list-avail 2>/dev/null | grep -o "SPI_RA_[A-Z_]*\|SQ_LEVEL_WAVES\|SQ_BUSY_CU_CYCLES" | sort -u > $O/avail.txt
cat $O/avail.txt | tr '\n' ' '; echo
C=$(grep -E "SPI_RA_(REQ_NO_ALLOC|RES_STALL|WAVE_SIMD_FULL|VGPR_SIMD_FULL|LDS_CU_FULL|BAR_CU_FULL|TMP_STALL)_CSN$" $O/avail.txt | head -7 | tr '\n' ' ')
[ -z "$C" ] && C=$(grep -E "SPI_RA" $O/avail.txt | head -6 | tr '\n' ' ')
echo "counters: $C"
if [ -n "$C" ]; then
  rm -rf /tmp/k6 && timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace -d /tmp/k6 -o k6 --output-format csv -- python3 $R/bench.py --isolated-only --no-cpu-baseline > $O/pmc_spi.json 2> $O/pmc_spi.err || { tail -5 $O/pmc_spi.err; exit 1; }
  python3 $R/scripts/pmc_summary.py "$(find /tmp/k6 -name '*counter_collection.csv' | head -1)" > $O/pmc_spi.txt
  grep -A12 -E "fm_search_filter|ed_trace_block|ed_exists_block|ed_traceback_wave" $O/pmc_spi.txt | head -80
fi
