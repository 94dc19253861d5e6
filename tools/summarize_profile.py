"""Turns what tools/profile_round.sh left under gpurun_out/<tag>/ into the tracked files under profiles/:
  <tag>_bench.json            the un-profiled bench line
  <tag>_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the same command
  <tag>_pmc_gemm_nt.json      HBM bytes per launch of the layer GEMM (bench.py reads this for roofline.traffic)
  <tag>_hbm_kernels.md        micro-benchmarks of the HBM-bound kernels against the 8 TB/s roof
  <tag>_gemm_bench.txt        NT / TN micro-benchmarks next to the vendor library on the same shapes
and prints the per-kernel table for <tag>_summary.md.
Units (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE count KiB; FETCH_SIZE is doubled on
gfx950 (128-byte requests counted as 64)."""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
stats = max(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)   # newest (gpurun merges, never deletes)
shutil.copy(stats, os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
for name in ("hbm_kernels.md", "gemm_bench.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
f = json.load(open(os.path.join(src, "pmc_FETCH_SIZE.json")))
w = json.load(open(os.path.join(src, "pmc_WRITE_SIZE.json")))
out = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 4 --warmup 2 "
                 "--no-cpu-baseline` (tools/profile_round.sh); per-launch mean over every launch of the kernel family; counters in KiB; "
                 "FETCH_SIZE doubled (gfx950 correction)"}
for k in ("gemm_nt_kernel", "gemm_tn_kernel"):
    fb = 2.0 * 1024.0 * f[k]["avg"]
    wb = 1024.0 * w[k]["avg"]
    out[k] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb,
              "launches_fetch_pass": f[k]["launches"], "launches_write_pass": w[k]["launches"]}
out["kernel"] = "gemm_nt_kernel (all tile configurations)"
out["traffic_bytes_per_launch"] = out["gemm_nt_kernel"]["traffic_bytes_per_launch"]
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_gemm_nt.json"), "w"), indent=1)

rows = list(csv.DictReader(open(stats)))
total = sum(int(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
iters = bench["steps"] + bench["warmup"]
print(f"GPU kernel time: {total / 1e6 / iters:.1f} ms / iteration, {calls / iters:.0f} launches / iteration; "
      f"un-profiled wall {bench['ms_per_step']:.1f} ms / iteration ({bench['value']:.2f} it/s).\n")
print("| kernel | share | calls | avg us |\n|---|---|---|---|")
for r in rows[:16]:
    print(f"| `{r['Name'][:80]}` | {100 * int(r['TotalDurationNs']) / total:.2f}% | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} |")
print("\nroofline:", json.dumps(bench["roofline"]))
print("cpu_baseline:", json.dumps(bench["cpu_baseline"]))
print("pmc:", json.dumps(out["gemm_nt_kernel"]), json.dumps(out["gemm_tn_kernel"]))
