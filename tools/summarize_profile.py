"""Turns what tools/profile_round.sh left under gpurun_out/<tag>/ into the tracked files under profiles/:
  <tag>_bench.json              the un-profiled bench line (driver flags: --steps 20 --warmup 5)
  <tag>_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the profiled command (+ <tag>_profiled_bench.json, its own line)
  <tag>_gemm_shapes.json        per-(M,N,K,mode,group) table of the layer-GEMM launches of the un-profiled run (HIP events): launches,
                                FLOP, ms, TFLOP/s -- what `roofline.achieved` is the total of
  <tag>_pmc_gemm_nt.json        HBM bytes per launch of the layer GEMMs (calibrated), next to the algorithmic bytes of the same launches
  <tag>_hbm_kernels.md, <tag>_gemm_bench.txt   micro-benchmarks
  <tag>_gaps.txt, <tag>_timeline.txt           GPU busy / idle of the traced run and one iteration as segments of GPU activity
and prints the tables for <tag>_summary.md.
Counter units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE count KiB.  The guide's gfx950 rule (double FETCH_SIZE)
holds for wide coalesced streams; for THIS kernel's operand-tile loads the factor is measured on launches with a known byte count
(tools/pmc_calibrate.py) and used instead; WRITE_SIZE is calibrated the same way."""
import csv, glob, json, os, shutil, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
line = lambda path: json.loads([x for x in open(path) if x.startswith("{")][-1])
bench = line(os.path.join(src, "bench.json"))
json.dump(bench, open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)
prof = line(os.path.join(src, "stats.log"))
json.dump(prof, open(os.path.join(dst, f"{tag}_profiled_bench.json"), "w"), indent=1)
stats = max(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats, os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
for name in ("hbm_kernels.md", "gemm_bench.txt", "gaps.txt", "timeline.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
shapes = json.load(open(os.path.join(src, "gemm_shapes.json")))
json.dump(shapes, open(os.path.join(dst, f"{tag}_gemm_shapes.json"), "w"), indent=0)

# ---- calibration: counter KiB per known byte on an INDEPENDENT streaming kernel (torch's vectorised copy, 16 bytes per lane -- the
# access width of the GEMM's operand loads), tools/pmc_calibrate.py.  The plain NT forward launches of the same pass (known operand
# sizes, far larger than the 256 MB Infinity Cache) are then READ with these factors: their measured / algorithmic ratio is a result.
cal, cal_gemm = {}, {}
exp = {4194304: (524288 * 512 * 4 + 512 * 512 * 4, 524288 * 512 * 4), 1048576: (131072 * 512 * 4 + 512 * 512 * 4, 131072 * 512 * 4)}   # GEMM grid size -> (read, write) bytes
for c, idx in (("FETCH_SIZE", 0), ("WRITE_SIZE", 1)):
    d = json.load(open(os.path.join(src, f"cal_{c}.json")))
    copies = {k: v for k, v in d.items() if k.startswith("stream_copy") and v["launches"] >= 4 and v["avg"] * 1024.0 > 1e8}      # the 1 GiB / 256 MiB copies (fills of the randn calls are smaller or write-only)
    fs = []
    for k, v in copies.items():
        nbytes = min((524288 * 512 * 4, 131072 * 512 * 4), key=lambda b: abs(b - v["avg"] * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)))
        fs.append(nbytes / (v["avg"] * 1024.0))
    cal[c] = sum(fs) / len(fs) if fs else (2.0 if c == "FETCH_SIZE" else 1.0)        # fall-back: the guide's factors
    cal_gemm[c] = {int(k.split("grid ")[1]): v["avg"] * 1024.0 * cal[c] / exp[int(k.split("grid ")[1])][idx]
                   for k, v in d.items() if k.startswith("gemm_nt_kernel") and int(k.split("grid ")[1]) in exp}

# ---- in-situ traffic: wide-output NT launches of the PMC passes against the algorithmic bytes of the SAME launches (their shape log)
def alg_bytes(rows):
    tot, n = 0.0, 0
    for r in rows:
        if r["mode"] in ("chain", "tn"):
            continue
        M, N, K = r["M"], r["N"], r["K"]
        b = 4.0 * (M * K + N * K + M * N) + (4.0 * M * N if r["mode"] == 1 else 0.0)      # the backward epilogue re-reads the stored activation
        tot += b * r["launches"]; n += r["launches"]
    return tot, n
out = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `bench.py --steps 6 --warmup 0 --settle 0 --settle-low 0 "
                 "--noise-observations --no-fine --no-cpu-baseline --no-sdf-throughput --shape-log ...` (every launch of the process is in the timed region) (tools/profile_round.sh); counters in KiB; calibration factors from "
                 "tools/pmc_calibrate.py (bytes that the kernel provably moves / counter bytes)",
       "calibration": {"FETCH_SIZE_bytes_per_counted_byte": round(cal["FETCH_SIZE"], 4), "WRITE_SIZE_bytes_per_counted_byte": round(cal["WRITE_SIZE"], 4),
                       "calibrated_on": "torch's vectorised copy of 1 GiB / 256 MiB (16 bytes per lane, coalesced): an independent streaming kernel, not the GEMM",
                       "gemm_nt_524288x512x512_measured_over_algorithmic": {"fetch": cal_gemm["FETCH_SIZE"].get(4194304), "write": cal_gemm["WRITE_SIZE"].get(4194304)},
                       "gemm_nt_131072x512x512_measured_over_algorithmic": {"fetch": cal_gemm["FETCH_SIZE"].get(1048576), "write": cal_gemm["WRITE_SIZE"].get(1048576)},
                       "note": "MI355X_MICROARCH.md: FETCH_SIZE counts half the bytes of a wide coalesced read on gfx950 (factor 2), WRITE_SIZE is exact (factor 1)"}}
wide_prefix = ("gemm_nt_kernel<2,2,1,1", "gemm_nt_kernel<2,2,1,2", "gemm_nt_kernel<2,2,2,2")      # (+ the K-tail flag since round 4: <2,2,2,2,false>)
class _Wide:
    def __contains__(self, k):
        return any(k.startswith(w) for w in wide_prefix)
wide = _Wide()
meas = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = json.load(open(os.path.join(src, f"pmc_{c}.json")))
    out[c] = d
    tot = sum(v["avg"] * v["launches"] for k, v in d.items() if k in wide)
    n = sum(v["launches"] for k, v in d.items() if k in wide)
    meas[c] = (tot * 1024.0 * cal[c], n)
    a, na = alg_bytes(json.load(open(os.path.join(src, f"pmc_shapes_{c}.json"))))
    out[c + "_algorithmic"] = {"bytes_total": a, "launches": na}
fetch_b, nf = meas["FETCH_SIZE"]; write_b, nw = meas["WRITE_SIZE"]
af, naf = out["FETCH_SIZE_algorithmic"]["bytes_total"], out["FETCH_SIZE_algorithmic"]["launches"]
out["gemm_nt_wide"] = {"fetch_bytes_per_launch": fetch_b / max(nf, 1), "write_bytes_per_launch": write_b / max(nw, 1),
                       "traffic_bytes_per_launch": fetch_b / max(nf, 1) + write_b / max(nw, 1),
                       "algorithmic_bytes_per_launch": af / max(naf, 1), "launches_counted": nf, "launches_in_shape_log": naf}
out["traffic_bytes_per_launch"] = out["gemm_nt_wide"]["traffic_bytes_per_launch"]
g_ = out["gemm_nt_wide"]
out["traffic_note"] = (f"HBM bytes per launch of the wide-output layer GEMMs from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, counter factors fixed on an "
                       f"independent streaming copy (profiles/{tag}_pmc_gemm_nt.json): fetch {g_['fetch_bytes_per_launch'] / 1e6:.1f} MB + write {g_['write_bytes_per_launch'] / 1e6:.1f} MB "
                       f"against {g_['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic = x{g_['traffic_bytes_per_launch'] / max(g_['algorithmic_bytes_per_launch'], 1.0):.2f}")
out["kernel"] = "gemm_nt_kernel, wide-output tile configurations (the launches bench.py's roofline record covers)"
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_gemm_nt.json"), "w"), indent=1)

# ---- tables
rows = list(csv.DictReader(open(stats)))
total = sum(int(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
passes = 2 if prof.get("ms_per_step_instrumented") is not None else 1          # bench.py times the K steps twice: clean, then with the event pairs
iters = prof["steps"] * passes + prof["warmup"] + prof["config"]["optimizer"]["settle_iters_lr_timed"] + prof["config"]["optimizer"]["settle_iters_lr_1e-4"]
print(f"un-profiled bench: {bench['ms_per_step']:.2f} ms / iteration ({bench['value']:.2f} it/s), converged {bench['config']['rays_converged_frac']:.3f}; "
      f"profiled command: {prof['ms_per_step']:.1f} ms / iteration under tracing, converged {prof['config']['rays_converged_frac']:.3f}, {iters} iterations")
print(f"GPU kernel time {total / 1e6 / iters:.1f} ms / iteration, {calls / iters:.0f} launches / iteration\n")
print("| kernel | share | calls / it | avg us |\n|---|---|---|---|")
for r in rows[:22]:
    print(f"| `{r['Name'][:70]}` | {100 * int(r['TotalDurationNs']) / total:.2f}% | {int(r['Calls']) / iters:.1f} | {float(r['AverageNs']) / 1e3:.1f} |")
aten = sum(int(r["TotalDurationNs"]) for r in rows if "at::native" in r["Name"] or "rocprim" in r["Name"] or "rocclr" in r["Name"])
aten_calls = sum(int(r["Calls"]) for r in rows if "at::native" in r["Name"] or "rocprim" in r["Name"] or "rocclr" in r["Name"])
print(f"\ntorch glue (at::native / rocprim / copies): {aten / 1e6 / iters:.2f} ms / iteration in {aten_calls / iters:.0f} launches")
ref = sum(int(r["Calls"]) for r in rows if "refine_" in r["Name"] or "mlp_layer_pair" in r["Name"] or "mlp_chain" in r["Name"])
print(f"refiner launches / iteration: {ref / iters:.0f}")
# GEMM classes from the shape log (events, un-profiled run)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in shapes:
    if r["mode"] == "chain":
        key = ("refiner chain (all layers of both nets)", "-", "-", "-")
    else:
        mb = "M<8k" if r["M"] < 8192 else "M<64k" if r["M"] < 65536 else "M>=64k"
        key = (mb, r["N"], r["K"], "weight-grad (TN)" if r["mode"] == "tn" else "bwd-data" if r["mode"] == 1 else "fwd")
    a = agg[key]; a[0] += r["launches"]; a[1] += r["flop"]; a[2] += r["ms"]
steps = bench["steps"]
print("\n| class | N | K | epilogue | launches / it | ms / it | TFLOP/s |\n|---|---|---|---|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])[:24]:
    print(f"| {k[0]} | {k[1]} | {k[2]} | {k[3]} | {v[0] / steps:.1f} | {v[2] / steps:.2f} | {v[1] / v[2] / 1e9:.1f} |")
nt = {k: v for k, v in agg.items() if k[3] != "weight-grad (TN)"}; tn = {k: v for k, v in agg.items() if k[3] == "weight-grad (TN)"}
tf = sum(v[1] for v in nt.values()); tm = sum(v[2] for v in nt.values())
tf2 = sum(v[1] for v in tn.values()); tm2 = sum(v[2] for v in tn.values())
print(f"\nNT tile code (forward, backward-data, refiner chains), all recorded launches: {tf / steps / 1e12:.3f} TFLOP / iteration in {tm / steps:.2f} ms of event "
      f"time -> {tf / tm / 1e9:.1f} TFLOP/s ({tf / tm / 1e9 / 157.3:.3f} of 157.3)")
if tm2 > 0:
    print(f"weight-gradient kernel (TN + slab reduction): {tf2 / steps / 1e12:.3f} TFLOP / iteration in {tm2 / steps:.2f} ms -> {tf2 / tm2 / 1e9:.1f} TFLOP/s "
          f"({tf2 / tm2 / 1e9 / 157.3:.3f} of 157.3)")
print(f"whole step: {(tf + tf2) / steps / 1e12:.3f} TFLOP / {bench['ms_per_step']:.2f} ms = {(tf + tf2) / steps / bench['ms_per_step'] / 1e9:.1f} TFLOP/s "
      f"({(tf + tf2) / steps / bench['ms_per_step'] / 1e9 / 157.3:.3f} of 157.3)")
print("\nroofline:", json.dumps(bench["roofline"]))
print("cpu_baseline:", json.dumps(bench["cpu_baseline"]))
print("pmc:", json.dumps(out["gemm_nt_wide"]), json.dumps(out["calibration"]))
