#!/bin/bash
# Kernel trace of the clean pass only (per-kernel stats, GPU busy / idle, one iteration's timeline): the cheap half of profile_round.sh.
#   gpurun --timeout 600 -- 'bash tools/profile_light.sh r03'
set -u
TAG=${1:-r03}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROF="python $R/bench.py --steps 30 --warmup 5 --settle 0 --settle-low 10 --noise-observations --no-fine --no-cpu-baseline ${PROFILE_EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $PROF > $OUT/stats.log 2>&1
python $R/tools/gap_analysis.py $OUT/stats 17 44 > $OUT/gaps.txt 2>&1
python $R/tools/iteration_timeline.py $OUT/stats 45 > $OUT/timeline.txt 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/stats/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
# launches per kernel name inside iterations 17..43 of the clean pass is what gap_analysis uses; here: whole-run counts by name
c = collections.Counter(r['Kernel_Name'][:90] for r in rows)
t = collections.Counter()
for r in rows:
    t[r['Kernel_Name'][:90]] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
with open(out + '/kernel_counts.txt', 'w') as fh:
    for k, n in c.most_common(70):
        fh.write(f"{n:8d} {t[k] / 1e6:10.2f} ms  {k}\n")
PY
find $OUT/stats -name "*kernel_trace.csv" -delete
cd $R; ls $OUT
