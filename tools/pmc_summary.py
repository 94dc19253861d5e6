"""Summarises a tools/pmc_gemm.py counter pass: MFMA-pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), with
kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; wave-time split from SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES."""
import csv, glob, sys, json, collections
d = sys.argv[1]
cc = glob.glob(d + '/*/*counter_collection.csv')[0]
kt = glob.glob(d + '/*/*kernel_trace.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    n = r['Kernel_Name']
    key = 'ours NT' if 'gemm_nt_kernel' in n else 'ours TN' if 'gemm_tn_kernel' in n else 'vendor NT (Cijk_Alik_Bljk)' if n.startswith('Cijk_Alik_Bljk') else \
          'vendor TN (Cijk_Ailk_Bjlk)' if n.startswith('Cijk_Ailk_Bjlk') else None
    if key: agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    n = r['Kernel_Name']
    key = 'ours NT' if 'gemm_nt_kernel' in n else 'ours TN' if 'gemm_tn_kernel' in n else 'vendor NT (Cijk_Alik_Bljk)' if n.startswith('Cijk_Alik_Bljk') else \
          'vendor TN (Cijk_Ailk_Bjlk)' if n.startswith('Cijk_Ailk_Bjlk') else None
    if key: dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = {}
for k, c in agg.items():
    m = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
    cyc = m['GRBM_GUI_ACTIVE'] / 8.0
    us = sorted(dur[k])[len(dur[k]) // 2] / 1e3
    out[k] = {"duration_us": round(us, 1), "shader_clock_ghz": round(cyc / us / 1e3, 3), "mfma_pipe_occupancy": round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024), 3),
              "wave_time_parked_waitcnt_barrier": round(m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], 3),
              "wave_time_issue_stalled": round(m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], 3),
              "wave_time_issuing": round(m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES'], 3)}
print(json.dumps(out, indent=1))
