#!/bin/bash
# Kernel trace of a short run, then the per-stream view of one iteration (tools/stream_view.py):  gpurun -- 'bash tools/profile_streams.sh r03_streams'
set -u
TAG=${1:-streams}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $R/bench.py --steps 12 --warmup 3 --settle 0 --settle-low 8 --no-gemm-events --no-fine --no-cpu-baseline --no-sdf-throughput ${PROFILE_EXTRA:-} > $OUT/run.log 2>&1
python $R/tools/stream_view.py $OUT/trace 16 > $OUT/streams.txt 2>&1
python $R/tools/iteration_timeline.py $OUT/trace 16 > $OUT/timeline.txt 2>&1
find $OUT/trace -name "*kernel_trace.csv" -delete
