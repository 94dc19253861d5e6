#!/bin/bash
# MFMA instructions of a whole iteration from the hardware counter (north_star: "MFMA utilisation against gfx950 peak"), next to the
# algorithmic count the roofline record uses:   gpurun -- 'bash tools/pmc_mfma_iteration.sh r04'
# One rocprofv3 pass (--kernel-trace + --pmc only) over `bench.py --steps 6 --settle 0 ...` (every launch of the process is an iteration's);
# SQ_INSTS_MFMA counts wave-level MFMA instructions; a v_mfma_f32_32x32x2_f32 is 4096 FLOP and occupies its SIMD's matrix pipe for 64 cycles.
set -u
TAG=${1:-r04}
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $OUT/pmc_mfma -- python $R/bench.py --steps 6 --warmup 0 --settle 0 --settle-low 0 --noise-observations --no-fine --no-cpu-baseline --no-extra-records --no-sdf-throughput --shape-log $OUT/pmc_shapes_mfma.json > $OUT/pmc_mfma.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(out + '/pmc_mfma/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        key = ('gemm_nt_kernel' if 'gemm_nt_kernel' in n else 'gemm_tn_kernel' if 'gemm_tn_kernel' in n else 'mlp_layer_pair_kernel' if 'mlp_layer_pair' in n
               else 'mlp_chain_kernel' if 'mlp_chain' in n else 'other')
        agg[key][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_INSTS_MFMA':
            calls[key] += 1
line = json.loads([x for x in open(out + '/pmc_mfma.log') if x.startswith('{')][-1])
steps = line['steps'] * 2                       # the clean pass and the instrumented pass
shapes = json.load(open(out + '/pmc_shapes_mfma.json'))
alg = sum(r['flop'] for r in shapes) / line['steps']                 # (the shape log covers the instrumented pass)
mfma = sum(v['SQ_INSTS_MFMA'] for v in agg.values()) / steps
res = {"what": "rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU over bench.py --steps 6 --settle 0 (12 iterations: clean + instrumented pass); per iteration",
       "mfma_instructions_per_iteration": mfma, "flop_issued_per_iteration": mfma * 4096.0, "algorithmic_flop_per_iteration": alg,
       "issued_over_algorithmic": mfma * 4096.0 / alg,
       "mfma_pipe_ms_per_iteration_at_2400_MHz": mfma * 64.0 / 1024.0 / 2.4e9 * 1e3,
       "by_kernel": {k: {"launches_per_iteration": calls[k] / steps, "mfma_per_iteration": v['SQ_INSTS_MFMA'] / steps,
                         "valu_per_mfma": (v['SQ_INSTS_VALU'] / v['SQ_INSTS_MFMA']) if v['SQ_INSTS_MFMA'] else None} for k, v in agg.items()},
       "ms_per_step_of_this_run_under_the_counter_pass": line['ms_per_step']}
json.dump(res, open(out + '/pmc_mfma.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/pmc_mfma
