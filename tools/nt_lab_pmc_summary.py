"""Counter passes of tools/nt_lab_run.sh (rocprofv3 --kernel-trace --pmc ..., one counter family per pass, the plain 128x128 NT launch at
262144 x 512 x 512) -> one JSON: per-launch medians, and what they say about the matrix pipe:
  mfma_util         = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / XCDs * SIMDs)      (matrix-pipe busy fraction of the launch)
  effective_ghz     = GRBM_GUI_ACTIVE / XCDs / kernel duration                          (the clock the launch ran at)
  tflops_from_count = SQ_INSTS_MFMA * 4096 FLOP / duration                              (must equal the algorithmic rate: no padding at this shape)
usage: python tools/nt_lab_pmc_summary.py gpurun_out/<tag>_lab out.json"""
import csv, glob, json, sys, collections, os


def main(d, out):
    res, durs = {}, []
    for p in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(p):
            continue
        agg = collections.defaultdict(list)
        for f in glob.glob(p + "/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "lab_kernel" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(p + "/*/*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                if "lab_kernel" in r["Kernel_Name"]:
                    durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in agg.items():
            res[k] = sorted(v)[len(v) // 2]
    dur = sorted(durs)[len(durs) // 2] if durs else None
    out_d = {"what": "rocprofv3 --kernel-trace --pmc passes over tools/_bin/nt_lab 262144 512 512 (relu epilogue, product tile code); per-launch medians",
             "kernel_duration_us_under_the_profiler": dur, "counters": res}
    xcds, simds = 8, 1024
    if "GRBM_GUI_ACTIVE" in res and "SQ_VALU_MFMA_BUSY_CYCLES" in res:
        cyc = res["GRBM_GUI_ACTIVE"] / xcds
        out_d["launch_cycles"] = cyc
        out_d["mfma_util"] = round(res["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * simds), 4)
        if dur:
            out_d["effective_ghz"] = round(cyc / dur / 1e3, 3)
    if "SQ_INSTS_MFMA" in res and dur:
        out_d["tflops_from_mfma_count"] = round(res["SQ_INSTS_MFMA"] * 4096 / dur / 1e6, 1)
    if "SQ_WAVE_CYCLES" in res:
        w = res["SQ_WAVE_CYCLES"]
        out_d["wave_cycle_shares"] = {k: round(res[k] / w, 4) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if k in res}
        out_d["note_wave_cycles"] = "SQ_WAVE_CYCLES etc. count quad-cycles; SQ_WAIT_INST_ANY (a wave with an instruction that cannot issue: here the next MFMA waiting for the pipe) dominates, as it must in an MFMA-bound loop"
    if "SQ_INSTS_MFMA" in res:
        out_d["instructions_per_mfma"] = {k: round(res[k] / res["SQ_INSTS_MFMA"], 3) for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD") if k in res}
    json.dump(out_d, open(out, "w"), indent=1)
    print(json.dumps(out_d, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
