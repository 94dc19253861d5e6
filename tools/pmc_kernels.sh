#!/bin/bash
# HBM traffic of the non-GEMM kernels: tools/kernel_bench.py under FETCH_SIZE / WRITE_SIZE passes (separate, --kernel-trace only).
#   gpurun --timeout 600 -- 'bash tools/pmc_kernels.sh r01'   ->  gpurun_out/<tag>_pmc_kernels.json
set -u
TAG=${1:-r01}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pk_$C -- python $R/tools/kernel_bench.py > /tmp/pk_$C.log 2>&1
done
python - "$R/gpurun_out/${TAG}_pmc_kernels.json" <<'PY'
import csv, glob, json, sys, collections
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"/tmp/pk_{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if r["Counter_Name"] != c or "at::native" in n or "rocprim" in n or n.startswith("Cijk") or "rocclr" in n:
                continue
            short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[(short, r["Grid_Size"])].append(float(r["Counter_Value"]))
    for (k, grid), v in agg.items():
        v = sorted(v)
        out[f"{k} [grid {grid}]"][c + "_KiB_median"] = v[len(v) // 2]
        out[f"{k} [grid {grid}]"]["launches"] = len(v)
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
