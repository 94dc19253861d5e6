#!/bin/bash
# Runs a command while sampling the shader clock and the power of the GPU every 0.2 s (rocm-smi); writes OUT/clocks.log and a histogram.
# Usage: tools/with_clocks.sh OUTDIR command [args...]
out=$1; shift
mkdir -p "$out"
( while true; do date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" ; sleep 0.2; done ) > "$out/clocks.log" 2>&1 &
sampler=$!
"$@"
rc=$?
kill $sampler 2>/dev/null
grep -E "sclk" "$out/clocks.log" | sed -E 's/.*\(([0-9]+)Mhz\).*/\1/' | sort -n | uniq -c > "$out/sclk_histogram.txt"
grep -E "Power" "$out/clocks.log" | sed -E 's/.*: ([0-9.]+).*/\1/' | sort -n | awk '{a[NR]=$1} END {if (NR) print "power W: min", a[1], "median", a[int((NR+1)/2)], "max", a[NR]}' > "$out/power.txt"
exit $rc
