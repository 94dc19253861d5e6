"""profiles/<tag>_summary.md from gpurun_out/<tag>/bench.json + the tables tools/summarize_profile.py prints:
   python tools/summarize_profile.py r06 > /tmp/tables.txt; python tools/write_round_summary.py r06 /tmp/tables.txt"""
import json, sys, os
tag, tables_path = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tables = open(tables_path).read()
body = tables[:tables.find('roofline: {')].rstrip()
d = json.loads([x for x in open(os.path.join(root, 'gpurun_out', tag, 'bench.json')) if x.startswith('{')][-1])
s, r, g = d['strong_scaling_model'], d['roofline'], d['diagnostics']
head = f'''# Round 6 profile summary (1 x MI355X, exact fp32; `tools/profile_round.sh {tag}` + the host / GPU timelines in ONE `gpurun` call on the final tree)

Files: `{tag}_bench.json` (the bench line as the driver runs it: `--steps 20 --warmup 5`), `{tag}_bench_kernel_stats.csv` + `{tag}_profiled_bench.json`
(rocprofv3 `--kernel-trace --stats` of the profiled command and that command's own JSON line), `{tag}_gemm_shapes.json` (per-(M,N,K,epilogue) table of
every layer-GEMM launch of the instrumented pass), `{tag}_pmc_gemm_nt.json` (FETCH_SIZE / WRITE_SIZE passes, counter factors from an independent
streaming kernel), `{tag}_gaps.txt` / `{tag}_timeline.txt` (GPU busy / idle and one iteration as segments, **under the tracer**), `{tag}_host_vs_gpu.txt` /
`{tag}_host_vs_gpu_one_rank_of_8.txt` / `..._masked.txt` (UNTRACED host clock against device-clock stamps at the marks of the iteration; `..._selection_first.txt`:
the issue order that was tried and reverted), `{tag}_nt_lab.md` + `{tag}_nt_stalls.json` + `{tag}_nt_stalls_round5_kernel.json` + `{tag}_nt_kloop_isa.txt` (where a
128x128 tile's time goes: stamps, counters, ISA), `{tag}_quality_trajectory_full_consistent.json` (32 free-running full-size iterations on the converging
scene against the reference's own run), `{tag}_race_amplifier.txt` (twelve forced-timing configurations, bit-identical), `{tag}_sync_points.txt` (the calls of
an iteration that make the host wait: the two count round trips), `{tag}_hbm_kernels.md`, `{tag}_gemm_bench.txt` (10-launch bursts: see the clock caveat in
`{tag}_nt_lab.md` -- these read 15-20 % below the sustained rates).

Headline (`{tag}_bench.json`): **{d['ms_per_step']:.2f} ms / iteration = {d['value']:.2f} it/s on the coarse stage at lr 1e-4** (round 5: 44.07 on the builder's boxes, 44.79 on the
driver's); the same tree and command on a second box (`{tag}_bench_second_box.json`, 2332 MHz at 1147 W): 42.72 ms, whole step 0.678; earlier trees of this round on other
boxes: 42.51, 42.56, 42.99; A / B runs of this tree: 42.2 - 42.7 -- the boxes differ by ~3 %, every kernel of the trace with them;
**round 5's tree and this one alternating on ONE box (`{tag}_vs_r05_same_box.md`): 43.58 -> 42.63 ms, whole step 0.657 -> 0.683; one rank of 8: 20.03 -> 18.15 ms**.  Late rate {d['late_schedule_lr']['ms_per_step']:.2f} ms, fine stage
{d['fine_stage']['ms_per_step']:.2f} ms, configs[4] (1080 x 1080, config_loose.conf) {d['loose1080']['ms_per_step']:.2f} ms.  Roofline record: NT tile code {r['achieved']:.1f} TFLOP/s over all
recorded launches ({r['frac']:.3f} of 157.3; >= 64k rows {r['achieved_launches_ge_64k_rows']:.1f}), weight-gradient kernel {r['weight_gradient_gemm']['achieved']:.1f} ({r['weight_gradient_gemm']['frac']:.3f}; round 5: 0.646),
**whole step 4.568 TFLOP / {d['ms_per_step']:.2f} ms = {r['whole_step_tflops']:.1f} TFLOP/s ({r['whole_step_frac']:.3f} of the fp32 MFMA peak; round 5: 0.659 builder / 0.648 driver)**.  Clock in the window
{g['sensors_mid_window']['sclk_mhz']} - {g['sensors_end_of_window']['sclk_mhz']} MHz at {g['sensors_mid_window']['power_w']:.0f} - {g['sensors_end_of_window']['power_w']:.0f} W; host enqueue {g['host_issue_ms_per_step']:.1f} ms per step, the final synchronisation waits {g['host_ahead_ms_at_the_end']:.1f} ms (the GPU paces the
step).  HBM traffic of the layer GEMMs: {r['traffic'] / 1e6:.1f} MB per launch against 195.2 MB algorithmic (x 1.11; a plain 524288-row launch fetches 1.42 x its
algorithmic reads, 1.35 in round 5; the kernel is an order of magnitude under the HBM roof either way).

configs[2] (8 frames over 8 GPUs) on ONE GPU (`strong_scaling_model`; the one-frame runs with the masked ray branch, `--masked-ray-branch-below 4096`): 8 frames
per step {s['ms_8_frames_one_gpu']:.1f} ms, one rank of 8 {s['ms_one_rank_of_8']:.2f} ms -> modelled speed-up **{s['modelled_speedup_8_gpus']}**; the same rank's step with EVERY collective live through
RCCL at world size 1: {s['ms_one_rank_of_8_collectives_live_world1']:.2f} ms (all-reduce section of the main stream {s['grad_allreduce_section_ms_world1']} ms) + {s['modelled_wire_ms_8_gpus']} ms of modelled xGMI wire time
-> **{s['modelled_speedup_8_gpus_with_collectives']}** (two separate scenes on a step paced by a dependent chain: +-0.5 ms between runs; the A / B probe on one box: 20.30 ms live
against 20.18 -- before the blocking-copy fix 24.3 against 20.0).

'''
open(os.path.join(root, 'profiles', f'{tag}_summary.md'), 'w').write(head + body + '\n')
print("wrote", f'profiles/{tag}_summary.md')
