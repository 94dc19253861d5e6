#!/bin/bash
# Sustained MFMA rate + shader-clock trace (rocm-smi sampled every 0.25 s while the kernels run).  Usage: tools/mfma_peak.sh OUTDIR [seconds]
out=${1:-gpurun_out/mfma_peak}; secs=${2:-3}
mkdir -p "$out"
( while true; do date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" ; sleep 0.25; done ) > "$out/clocks.log" 2>&1 &
sampler=$!
tools/_bin/mfma_peak "$secs" | tee "$out/mfma_peak.txt"
kill $sampler
grep -E "sclk" "$out/clocks.log" | sed -E 's/.*\(([0-9]+)Mhz\).*/\1/' | sort -n | uniq -c | sort -k2 -n > "$out/sclk_histogram.txt"
echo "sclk histogram (count MHz):"; cat "$out/sclk_histogram.txt"
