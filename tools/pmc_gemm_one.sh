#!/bin/bash
# Counter passes over one GEMM kernel (each pass its own run; --kernel-trace + --pmc only).  Usage: tools/pmc_gemm_one.sh OUT f32
R=$PWD; out=$R/$1; mode=$2
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pg_$i -- python $R/tools/pmc_gemm_one.py $mode > $out/pass_$i.log 2>&1
  python - /tmp/pg_$i "$out/pass_$i.json" <<'PY'
import csv, glob, sys, json, collections
d, o = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(d + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'gemm_nt' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
dur = []
for f in glob.glob(d + '/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'gemm_nt' in r['Kernel_Name']:
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
res = {k: sorted(v)[len(v) // 2] for k, v in agg.items()}
res['duration_us'] = sorted(dur)[len(dur) // 2] / 1e3 if dur else None
json.dump(res, open(o, 'w'), indent=1); print(json.dumps(res))
PY
  rm -rf /tmp/pg_$i
done
