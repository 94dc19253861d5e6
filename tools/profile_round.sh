#!/bin/bash
# Collects the round's profiles on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/profile_round.sh r01'
# Writes under gpurun_out/<tag>/ (merged back by gpurun); tools/summarize_profile.py turns them into profiles/<tag>_*.
# rocprofv3 rules of this pool: --pmc passes carry --kernel-trace only, one counter family per pass.
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the un-profiled bench line (with the CPU baseline leg)
python $R/bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
# 2. per-kernel statistics of the same command (no CPU leg: it is host-only)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_trace.csv" -delete
# 3. HBM traffic of the layer GEMMs: FETCH_SIZE and WRITE_SIZE in separate passes
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1
  python - "$OUT/pmc_$C" "$C" <<'PY'
import csv, glob, sys, json, collections
d, c = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != c: continue
        n = r['Kernel_Name']
        key = 'gemm_nt_kernel' if 'gemm_nt_kernel' in n else 'gemm_tn_kernel' if 'gemm_tn_kernel' in n else None
        if key: agg[key][0] += 1; agg[key][1] += float(r['Counter_Value'])
json.dump({k: {'launches': v[0], 'avg': v[1] / max(v[0], 1)} for k, v in agg.items()}, open(d + '.json', 'w'), indent=1)
PY
  rm -rf $OUT/pmc_$C
done
# 4. kernel micro-benchmarks against their rooflines
python $R/tools/kernel_bench.py > $OUT/hbm_kernels.md 2> $OUT/hbm_kernels.err
python $R/tools/gemm_bench.py > $OUT/gemm_bench.txt 2>&1
cd $R
ls -la $OUT
