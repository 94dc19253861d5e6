#!/usr/bin/env python
"""bench.py -- `train.py`-equivalent iterations/s of the SelfRecon SDF-optimisation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--stage coarse|fine] [--lr LR] [--no-fine] [--no-extra-records] [--scaling weak|strong] [--frames-per-gpu F] [--simulate-world R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
(a plain `python bench.py --gpus N` with N > 1 and no launcher environment re-executes itself through torch.distributed.run, one rank
per GPU -- all ranks on device 0 if the box has fewer than N GPUs, which is a functional run, not a measurement)

Scaling modes (the path shards over FRAMES, no data-path collective besides the gradient all-reduce):
  weak   (default)  every rank optimises `--frames-per-gpu` frames per step (default: the stage's batch size, 3 coarse / 1 fine);
                    `--frames-per-gpu 1` at N = 8 is BASELINE.json configs[2] (8 frames, one per GPU);
  strong            `--global-frames` frames per step (default 8 = configs[2]) are split over the ranks: 8 / N frames each.
`value` counts reference iterations: frames processed per second / the stage's batch size (3 coarse, 1 fine), whole job.

One "step" = one full training iteration of BASELINE.json configs[1] (female-3-casual-like, 540x540): template deformation +
point-silhouette mask loss + template SGD step, mesh rasterisation + ray seeding + Newton refiner, eikonal / offset / deformation-
regulariser / DCT / colour / normal losses, backward, implicit-gradient propagation, Adam step, and the periodic remesh
(Seg3dLossless + marching cubes).  Headline workload: the coarse stage (3 frames x 2048 rays per rank, 225x321x129 grid, remesh
every 30 iterations; config.conf:28-34); `--stage fine` (and, at N=1, the `fine_stage` record of the default run) is the stage
189 of the reference's 201 epochs run in (1 frame x 6144 rays, 321x417x225 grid, remesh every 120; config.conf:39-48,113).

What the timed iterations look like is decided by the state of the scene, so the scene is brought to the state a sequence is in
while that stage runs before anything is timed (all of it outside the timed region):
  1. observations (colour, normal, silhouette images of every frame) are RENDERED from the scene itself
     (OptimNetwork.render_frames = the colour pass of the reference's `infer`), not drawn from noise;
  2. `--settle` iterations at the configured learning rate 1e-4 let Adam's moments and the SDF / template equilibrium form, and the
     observations are rendered again from the settled model;
  3. the record is then timed AT THE LEARNING RATE THE REFERENCE RUNS THAT STAGE AT: the coarse stage (the headline) is epochs 0-5 of
     config.conf, all of them at 1e-4 (config.conf:16-34); the fine stage (epochs 12-200) is timed at 1e-4 * 0.333^3, the MultiStepLR
     value of epochs 80-129 (`--late-lr`; `--settle-low` iterations at that rate first).  Rounds 2-3 timed the COARSE workload at
     the late rate -- a combination the reference never runs; it stays as the secondary record `late_schedule_lr`.
At lr 1e-4 the refiner accepts 7-16 % of the rays between two remeshes: Adam's limit cycle on the L1 template term
60*mean|f(TmpVs)| keeps the seeds ~3e-3 off the zero set, 60x the acceptance threshold |f| < 5e-5 (profiles/r02_convergence.md).
The reference's own run does the same: tests/test_trajectory_long_gpu.py holds the product's acceptance rate to the reference's
(0.09 / 0.25 / 0.15 / 0.26 per 16 iterations of a 64-iteration run at lr 1e-4 with four remeshes).  At the late rate ~75 % converge.
Other records of the default single-GPU run: `fine_stage`, `strong_scaling_model`
(configs[2]: 8 frames on one GPU against the workload of one rank of 8), `seg3d_mc_513` (configs[3]), `loose1080` (configs[4]).
The timed window always contains exactly one remesh when K <= the remesh interval (for K = 20 that over-counts its share:
1/20 instead of 1/30 or 1/120); its duration is reported, with the properly amortised figure beside.  The K timed steps run
twice: the first pass carries no instrumentation and gives `value` (refiner on the side stream, concurrent with the template
branch); the second repeats them with HIP-event pairs around every layer-GEMM launch (roofline leg), the remesh and the refiner,
with the refiner and the weight-gradient GEMMs on the main stream so that an event interval is one kernel's duration.
Camera focal length, principal point and T are learnable as in config.conf:10-15 (quaternion fixed).
Inputs are synthetic and resident in HBM before the timed region.  Prints ONE JSON line (rank 0).
"""
import argparse
import gc
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (multi-process GPU work on this pool: the host driver only supports dmabuf IPC; must be set before HIP initialises)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STAGES = {"coarse": dict(frames=3, rays=2048), "fine": dict(frames=1, rays=6144)}


def gpu_sensors(device_index=0):
    """Shader clock (MHz) and socket power (W) of THIS process's GPU from sysfs -- two small file reads, cheap enough to take INSIDE the
    timed window while the GPU is loaded (rocm-smi is a Python process of its own: 0.3 s of host CPU per sample).  The card is found by
    the PCI bus id torch reports (a box has several GPUs and the visible one is not card0); if that fails, the busiest card (highest
    current clock) is reported and marked.  None where the files do not exist (containers without /sys/class/drm)."""
    import glob

    def read(card):
        out = {}
        for line in open(os.path.join(card, "pp_dpm_sclk")):
            if "*" in line:
                out["sclk_mhz"] = int("".join(c for c in line.split(":")[1] if c.isdigit()))
        for name in ("power1_average", "power1_input"):
            hs = glob.glob(os.path.join(card, "hwmon", "hwmon*", name))
            if hs:
                out["power_w"] = round(int(open(hs[0]).read().strip()) / 1e6, 1)
                break
        return out
    try:
        cards = sorted(d for d in glob.glob("/sys/class/drm/card*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")))
        if not cards:
            return None
        want = None
        try:
            want = int(getattr(torch.cuda.get_device_properties(device_index), "pci_bus_id"))
        except Exception:
            pass
        for d in cards:
            addr = os.path.basename(os.path.realpath(d))            # 0000:bb:dd.f
            if want is not None and len(addr.split(":")) == 3 and int(addr.split(":")[1], 16) == want:
                return dict(read(d), card=os.path.basename(os.path.dirname(d)), matched_by="pci_bus_id") or None
        best = max((read(d) for d in cards), key=lambda r: r.get("sclk_mhz", 0))
        return dict(best, matched_by="highest clock of %d cards" % len(cards)) or None
    except (OSError, ValueError, IndexError):
        return None


def _eager_ray_branch():
    from selfreconcode_amd.model import optim_network
    return bool(optim_network.EAGER_RAY_BRANCH)


def frames_per_rank(stage, args, world):
    """Frames one rank optimises per step (see the scaling modes in the module docstring)."""
    if args.scaling == "strong":
        if args.global_frames % world:
            raise SystemExit(f"--scaling strong: --global-frames {args.global_frames} is not divisible by {world} ranks")
        return args.global_frames // world
    return args.frames_per_gpu or STAGES[stage]["frames"]


def run_stage(stage, args, rank, world, device, steps, warmup, settle, settle_low, gemm_events, timed_lr=None, late_lr=None, frames=None, scene=None):
    """`timed_lr`: Adam learning rate of the timed region (None: the configuration's own 1e-4, no change after the settle phase);
    `late_lr`: additionally time the same workload at this later rate of the schedule (secondary record `late_schedule_lr`);
    `frames`: frames per rank and step (default: frames_per_rank); `scene`: keyword overrides of build_synthetic_scene (image size, config)."""
    from selfreconcode_amd import dist as srdist
    from selfreconcode_amd import mlp_engine
    from selfreconcode_amd.synthetic import build_synthetic_scene
    FR, RAYS = frames or frames_per_rank(stage, args, world), STAGES[stage]["rays"]
    net, ds, conf = build_synthetic_scene(device=device, frame_num=max(64, 2 * FR * world), stage=stage, consistent_masks=False, **(scene or {}))
    params = [p for p in net.parameters() if p.requires_grad]
    net.refiner_stream = args.refiner_stream
    net.masked_ray_branch_below = args.masked_ray_branch_below      # at most this many selected rays (one frame per rank: 2048): no converged-ray round trip
    from selfreconcode_amd.utils import FindSurfacePs as _fsp
    _fsp.DEVICE_DRIVEN = args.refiner_impl == "device"
    mlp_engine.set_deferred_param_grads(True)              # one weight-norm backward + grad add per layer per step
    lr0 = conf.get_float('train.learning_rate')
    from selfreconcode_amd.optim import FusedAdam          # torch.optim.Adam's update (train.py:139) in one launch per step (--torch-adam: torch's own)
    Adam = torch.optim.Adam if args.torch_adam else FusedAdam
    opt = Adam([{'params': ds.learnable_weights()}, {'params': params}], lr=lr0)
    bucket = srdist.GradBucket(list(ds.learnable_weights()) + params, early=list(net.netRender.parameters()) + [ds.conds[1]])
    bucket.sync_initial_state()
    ratio_of = lambda it: {'sdfRatio': 1., 'deformerRatio': min(1.0, it / 2500. + 0.5), 'renderRatio': 1.}

    def frames_of(it):
        base = (it * FR * world) % (ds.frame_num - FR * world + 1)
        glob = torch.arange(base, base + FR * world, device=device)
        return srdist.shard_frames(glob, rank, world)

    conv = []
    state = {"it": 0}
    coll = {"on": False, "events": []}      # (N > 1 or forced collectives) event pairs around the gradient all-reduce of the timed steps

    def step():
        it = state["it"]
        f = frames_of(it)
        opt.zero_grad(set_to_none=True)
        loss = net(ds.batch(f), RAYS, ratio_of(it), f)
        loss.backward()
        net.propagateTmpPsGrad(f, ratio_of(it), overlap=bucket)
        if coll["on"]:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            bucket.all_reduce_mean()
            e1.record()
            coll["events"].append((e0, e1))
        else:
            bucket.all_reduce_mean()
        opt.step()
        conv.append(net.info['rayInfo'])
        state["it"] = it + 1

    def barrier():
        if srdist.is_distributed():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    diag = {}

    def timed(n, sense=False):
        """`sense`: also record what explains a box-to-box difference of the figure -- the shader clock / power half-way through and at
        the end of the window (GPU loaded), and how long the final synchronisation waits after the host has issued the last step
        (~0: the host is the bottleneck; several ms: the GPU is, the host runs ahead)."""
        conv.clear()
        coll["events"].clear()
        coll["on"] = sense and srdist.is_distributed()
        barrier()
        t0 = time.perf_counter()
        for i in range(n):
            step()
            if sense and i == n // 2:
                diag["sensors_mid_window"] = gpu_sensors()
        t_issued = time.perf_counter()
        if sense:
            diag["sensors_end_of_window"] = gpu_sensors()
        barrier()
        el = time.perf_counter() - t0
        if sense:
            diag["host_issue_ms_per_step"] = round((t_issued - t0) / n * 1e3, 3)
            diag["host_ahead_ms_at_the_end"] = round((t0 + el - t_issued) * 1e3, 3)
        coll["on"] = False
        t = torch.tensor([el], device=device, dtype=torch.float64)
        if sense and srdist.is_distributed():
            # a multi-rank record explains itself: every rank's own step time (the headline is their maximum), and the time the main
            # stream spends in the gradient all-reduce section (flat gather + all-reduce + scatter of both buckets; the early bucket
            # has been running under the implicit-gradient pass) per step on every rank
            ar = sum(a.elapsed_time(b) for a, b in coll["events"]) / max(len(coll["events"]), 1)
            mine = torch.tensor([el / n * 1e3, ar], device=device, dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(torch.distributed.get_world_size())]
            torch.distributed.all_gather(allr, mine)
            per = [round(float(x[0]), 3) for x in allr]
            diag["per_rank_ms_per_step"] = per
            diag["straggler_ms"] = round(max(per) - min(per), 3)
            diag["grad_allreduce_section_ms_per_step_by_rank"] = [round(float(x[1]), 3) for x in allr]
        if srdist.is_distributed():
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        tot = sum(int(r[0]) for r in conv); cv = sum(int(r[1]) for r in conv)
        return float(t), tot / max(len(conv), 1), cv / max(tot, 1)

    rec = {}
    if not args.noise_observations:
        ds.attach_rendered_observations(net, ratio_of(0))
    timed_lr = lr0 if timed_lr is None else timed_lr
    if settle > 0:                                                        # phase 2: lr 1e-4
        if timed_lr == lr0:
            for _ in range(settle):
                step()
        else:
            for _ in range(max(settle - 30, 0)):
                step()
            n = min(30, settle)
            el, rays, cf = timed(n)
            rec["regime_lr_config"] = {"lr": lr0, "ms_per_step": round(el / n * 1e3, 3), "rays_converged_frac": round(cf, 4), "steps": n,
                                       "note": "same workload at the learning rate config.conf starts with (epochs 0-9), measured at the end of the settle phase"}
        if not args.noise_observations:
            ds.attach_rendered_observations(net, ratio_of(state["it"]))
    if timed_lr != lr0:
        for g in opt.param_groups:                                        # phase 3: a later rate of the schedule
            g['lr'] = timed_lr
        for _ in range(settle_low):
            step()
    # remesh phase: exactly one remesh inside the timed window when steps <= interval
    Rm = net.remesh_intersect
    net.forward_time = (-(warmup + steps // 2)) % Rm or Rm
    for _ in range(warmup):
        step()
    # Pass 1 -- the headline: no instrumentation at all (an event pair around each of the ~400 layer-GEMM launches of an iteration
    # costs 2-6 ms of it: every hipEventRecord is a marker packet the command processor has to retire between two kernels).
    net.remesh_events = net.refiner_events = None
    mlp_engine.PROFILE.reset(enabled=False)
    el, rays, cf = timed(steps, sense=True)
    # Pass 2 -- the same K steps again with the HIP-event pairs (roofline leg, per-shape table), the remesh and refiner events
    prof, shapes, rem, refiner_ms, el_i = {}, None, [], None, None
    net.refiner_stream = "main"          # instrumented pass: one stream of GEMMs, so that an event interval is a kernel's own duration
    if gemm_events:
        net.forward_time = (-(steps // 2)) % Rm or Rm
        net.remesh_events, net.refiner_events = [], []
        mlp_engine.PROFILE.reset(enabled=True, reserve=1100 * steps)
        el_i, _, _ = timed(steps)
        prof = mlp_engine.PROFILE.summary()
        shapes = mlp_engine.PROFILE.by_shape()
        mlp_engine.PROFILE.reset(enabled=False)
        torch.cuda.synchronize()
        rem = [a.elapsed_time(b) for a, b in net.remesh_events]
        if net.refiner_events:
            refiner_ms = sum(a.elapsed_time(b) for a, b in net.refiner_events) / steps
        net.remesh_events = net.refiner_events = None
    else:                                                                 # (fine-stage record: remesh / refiner durations only, a handful of events)
        net.forward_time = (-(steps // 2)) % Rm or Rm
        net.remesh_events, net.refiner_events = [], []
        el_i, _, _ = timed(steps)
        torch.cuda.synchronize()
        rem = [a.elapsed_time(b) for a, b in net.remesh_events]
        if net.refiner_events:
            refiner_ms = sum(a.elapsed_time(b) for a, b in net.refiner_events) / steps
        net.remesh_events = net.refiner_events = None
    if late_lr is not None:          # the same workload at a later rate of the MultiStepLR schedule (more rays converge: a heavier colour / normal branch)
        net.refiner_stream = args.refiner_stream
        for g in opt.param_groups:
            g['lr'] = late_lr
        if not args.noise_observations:
            ds.attach_rendered_observations(net, ratio_of(state["it"]))
        for _ in range(settle_low):
            step()
        n = min(steps, 20)
        net.forward_time = (-(n // 2)) % Rm or Rm
        el_l, rays_l, cf_l = timed(n)
        rec["late_schedule_lr"] = {"lr": late_lr, "ms_per_step": round(el_l / n * 1e3, 3), "rays_converged_frac": round(cf_l, 4), "steps": n,
                                   "note": "same workload after the learning rate has dropped to the MultiStepLR value of epochs 80-129 (config.conf:18-27); the reference runs "
                                           "its FINE stage at that rate, not this one -- kept because it was the headline of rounds 2-3 (49.8 ms) and because it is the heavier mix"}
    ms = el / steps * 1e3
    rem_each = sum(rem) / len(rem) if rem else None
    rec.update({"ms_per_step": round(ms, 3), "elapsed": el, "ms_per_step_instrumented": None if el_i is None else round(el_i / steps * 1e3, 3), "rays_per_iter": round(rays, 1), "rays_converged_frac": round(cf, 4),
                "template_vertices": int(net.TmpVs.shape[0]), "remesh": {"in_window": len(rem), "ms_each": None if rem_each is None else round(rem_each, 3),
                                                                        "interval": Rm},
                "ms_per_step_remesh_amortised": None if rem_each is None else round((el * 1e3 - rem_each) / steps + rem_each / Rm, 3),
                "refiner_ms_per_step": None if refiner_ms is None else round(refiner_ms, 3),
                "diagnostics": diag, "frames_per_gpu": FR, "rays_per_frame": RAYS, "image": [ds.H, ds.W], "lr_timed": timed_lr, "prof": prof, "shapes": shapes, "net": net})
    return rec


def cpu_baseline_record(args, device):
    """`cpu_baseline`: the CPU oracle (checker-side code -- this leg is the only place bench.py touches oracle/) timed on THIS box's host
    cores on ONE whole iteration at the size of the timed workload (measured, not extrapolated), on a freshly built scene of the same
    configuration; beside it the figure of the REFERENCE'S OWN modules at the same size, measured in the build container by
    oracle/gen_fullsize_golden.py --time (profiles/r03_cpu_reference.json: the reference tree does not exist on the GPU box)."""
    from oracle.cpu_baseline import full_iteration_seconds
    from selfreconcode_amd.synthetic import build_synthetic_scene
    FR, RAYS = frames_per_rank(args.stage, args, 1), STAGES[args.stage]["rays"]
    net, ds, conf = build_synthetic_scene(device=device, frame_num=64, stage=args.stage, consistent_masks=False)
    ratio = {'sdfRatio': 1., 'deformerRatio': 0.5, 'renderRatio': 1.}
    with torch.no_grad():
        net.angThred = net._cameras(FR, device)[0].angThreshold(0.5)
        net.TmpVs, net.Tmpfs = net.discretizeSDF(ratio, None, 0.0)
    fids = torch.arange(FR, device=device)
    sec, sec_raster, threads, info = full_iteration_seconds(net, ds, fids, RAYS, ratio)
    rec = {"value": round(1.0 / (sec - sec_raster), 5), "unit": "iterations/s", "cores": threads, "kind": "port",
           "sample": f"ONE whole iteration of the CPU oracle (restated reference PyTorch path: forward + backward + propagateTmpPsGrad, its own refiner) at the "
                     f"full size of the timed workload ({FR} frame(s) x {RAYS} rays, {info['template_vertices']} template vertices, 540x540, 65x225x129 volume), "
                     f"float32, {threads} torch threads, one sample after a warm-up of the thread pool (two eikonal steps of the oracle's SDF on 16k points); the numpy restatement of the third-party rasterisers ({sec_raster:.1f} s) is excluded; remesh excluded",
           "seconds_per_iteration": round(sec - sec_raster, 3), "seconds_rasteriser_restatement": round(sec_raster, 3), **info}
    ref = os.path.join(ROOT, "profiles", "r03_cpu_reference.json")
    if os.path.isfile(ref):
        with open(ref) as fh:
            r = json.load(fh)
        st = [x for x in r["stages"] if x["stage"] == args.stage]
        if st:
            rec["reference_modules"] = {"kind": "reference", "where": r["where"], "cores": r["cores"], "dtype": r.get("dtype", "f32"), "value": st[0]["iterations_per_s"],
                                        "unit": "iterations/s", "seconds_per_iteration": st[0]["seconds_per_iteration"], "protocol": r["protocol"],
                                        "note": "the reference's OWN modules (model/network.py:451-814) at the same size, timed in the build container (profiles/r03_cpu_reference.json); "
                                                "the reference tree does not travel to the GPU box"}
    del net, ds
    gc.collect(); torch.cuda.empty_cache()
    return rec


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-execute through torch.distributed.run, one rank per GPU, same arguments."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < n:                       # functional run of the N-rank path on a smaller box (not a measurement)
        env["SR_ALL_RANKS_ON_DEVICE0"] = "1"
        env.setdefault("SR_DIST_BACKEND", "gloo")           # RCCL refuses two ranks on one device
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def seg3d_mc_513_record(device):
    """BASELINE.json configs[3]: Seg3dLossless + marching cubes on the 33..513 cubic pyramid (train.py:55-61) over the SDF MLP."""
    from selfreconcode_amd.MCAcc import Seg3dLossless
    from selfreconcode_amd.model.network import getTmpSdf
    from selfreconcode_amd.ext import MCGpu
    from selfreconcode_amd.synthetic import sphere_sdf_params
    net = getTmpSdf(device, 6, 0.6, 256)
    net.load_state_dict(sphere_sdf_params(7), strict=True)

    def q(points):
        with torch.no_grad():
            return net(points.reshape(-1, 3), 1.0, sdf_only=True).reshape(1, 1, -1)
    res = [(33,) * 3, (65,) * 3, (129,) * 3, (257,) * 3, (513,) * 3]
    eng = Seg3dLossless(q, [-0.9] * 3, [0.9] * 3, res, balance_value=0.0, use_cuda_impl=True).to(device)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    vol = eng.forward()                                                     # warm-up (allocator, weight packs)
    sdf = vol[0, 0].permute(2, 1, 0).contiguous()
    MCGpu.mc_gpu(sdf, eng.spacing_x, eng.spacing_y, eng.spacing_z, eng.bx, eng.by, eng.bz, 0.)
    t_seg, t_mc = [], []
    for _ in range(3):
        a, b, c = ev(), ev(), ev()
        a.record()
        vol = eng.forward()
        sdf = vol[0, 0].permute(2, 1, 0).contiguous()
        b.record()
        verts, faces = MCGpu.mc_gpu(sdf, eng.spacing_x, eng.spacing_y, eng.spacing_z, eng.bx, eng.by, eng.bz, 0.)
        c.record()
        torch.cuda.synchronize()
        t_seg.append(a.elapsed_time(b)); t_mc.append(b.elapsed_time(c))
    t_seg.sort(); t_mc.sort()
    nvox, V, F = 513 ** 3, int(verts.shape[0]), int(faces.shape[0])
    mc_bytes = 4 * nvox + 12 * V + 24 * F                                   # 4 B / voxel compulsory read + 12 V + 24 F out
    rec = {"workload": "configs[3]: Seg3dLossless (33^3 .. 513^3, fused 2x upsample + candidate selection) over the SDF MLP + marching cubes at 513^3; median of 3",
           "seg3d_ms": round(t_seg[1], 3), "queries": int(eng.stats["queries"]), "voxels": nvox, "queries_frac": round(eng.stats["queries"] / nvox, 5),
           "mc_ms": round(t_mc[1], 3), "mc_vertices": V, "mc_faces": F, "mc_algorithmic_bytes": mc_bytes,
           "mc_roofline": {"bound": "hbm", "achieved": round(mc_bytes / (t_mc[1] * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                           "frac": round(mc_bytes / (t_mc[1] * 1e-3) / 8e12, 4),
                           "note": "mc_gpu call as the step makes it (classify, scans, emit, host read of the two counts, result tensors); kernels alone: profiles/"}}
    del net, eng, vol, sdf
    gc.collect(); torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--stage", choices=list(STAGES), default="coarse")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="weak: fixed frames per GPU; strong: --global-frames split over the GPUs")
    ap.add_argument("--frames-per-gpu", type=int, default=0, help="weak scaling: frames per rank and step (default: the stage's batch size; 1 at N=8 = configs[2])")
    ap.add_argument("--global-frames", type=int, default=8, help="strong scaling: frames per step over all ranks (8 = configs[2])")
    ap.add_argument("--lr", type=float, default=None, help="Adam learning rate of the timed region (default: the rate config.conf runs the stage at -- 1e-4 for the "
                                                           "coarse stage = epochs 0-5, the late rate for the fine stage)")
    ap.add_argument("--late-lr", type=float, default=1e-4 * 0.333 ** 3, help="the MultiStepLR value of epochs 80-129 of config.conf (fine-stage record, `late_schedule_lr` record)")
    ap.add_argument("--settle", type=int, default=120, help="untimed iterations at lr 1e-4 before the timed region")
    ap.add_argument("--settle-low", type=int, default=40, help="untimed iterations at a later rate before it is timed")
    ap.add_argument("--noise-observations", action="store_true", help="uniform-noise colour/normal targets instead of rendered ones (round-1 workload)")
    ap.add_argument("--no-fine", action="store_true", help="skip the fine-stage record of the default single-GPU run")
    ap.add_argument("--refiner-stream", choices=["main", "side"], default="side", help="headline pass: run the refiner concurrently with (side) or after (main) the template branch; the instrumented pass always uses main")
    ap.add_argument("--refiner-impl", choices=["device", "layerwise"], default="device", help="device-driven compacting refiner or the layer-by-layer host loop")
    ap.add_argument("--no-sdf-throughput", action="store_true", help="skip the SDF-MLP Gsamples/s leg (PMC passes: keeps the launch population = the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-records", action="store_true", help="skip the late-rate, configs[3] (Seg3d + MC at 513^3), configs[4] (1080 x 1080, config_loose.conf) and strong-scaling-model records")
    ap.add_argument("--simulate-world", type=int, default=0, help="N=1 only: time the workload ONE rank of this many would run (the replicated template term evaluated on 1/R of the vertices, no collectives)")
    ap.add_argument("--masked-ray-branch-below", type=int, default=4096, help="OptimNetwork.masked_ray_branch_below: with at most this many selected rays the colour / normal "
                    "terms and the implicit-gradient pass run on all of them, the rejected ones masked -- no host round trip for the converged-ray count (0: always compact the list, as the "
                    "reference does).  The headline workload (3 x 2048 rays) is above the default and unaffected; one frame per rank (configs[2]) is below it")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam (multi-tensor launches) instead of the one-launch FusedAdam")
    ap.add_argument("--no-gemm-events", action="store_true", help="skip the HIP-event pairs around the layer GEMMs (roofline leg) to see their cost")
    ap.add_argument("--shape-log", default=None, help="write the per-(M,N,K) GEMM launch table (events) to this JSON file")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    from selfreconcode_amd import dist as srdist
    rank, world, device = srdist.init_from_env("cuda", bind_cpus=True)     # (host threads on a core group of the GPU's NUMA node; SR_BIND_CPUS=0: off)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus must agree")
    rccl = srdist.describe()                                  # (a collective: every rank calls it)
    if world > 1 and rccl.get("devices") and os.environ.get("SR_ALL_RANKS_ON_DEVICE0") != "1" and str(rccl.get("backend", "")).startswith("rccl"):
        ids = {(d["pci_bus_id"], d["pci_device_id"], d["device_index"]) for d in rccl["devices"]}
        if len(ids) != world:                                 # an N-GPU record must be N ranks on N different devices
            raise SystemExit(f"--gpus {world}: the ranks sit on {len(ids)} distinct device(s): {rccl['devices']}")
    if args.simulate_world > 1:
        if world != 1:
            raise SystemExit("--simulate-world is a single-GPU measurement")
        srdist.simulate_world((0, args.simulate_world))

    # The learning rate of a record is the one config.conf runs that stage at: the coarse stage is epochs 0-5, all at 1e-4
    # (config.conf:16-34); the fine stage starts at epoch 12 and spends epochs 80-129 at 1e-4 * 0.333^3 (the MultiStepLR milestones).
    lr_of = lambda stage: args.lr if args.lr is not None else (None if stage == "coarse" else args.late_lr)
    extras = world == 1 and not args.no_extra_records and not args.simulate_world and args.scaling == "weak" and not args.frames_per_gpu
    main_rec = run_stage(args.stage, args, rank, world, device, args.steps, args.warmup, args.settle, args.settle_low, not args.no_gemm_events,
                         timed_lr=lr_of(args.stage), late_lr=args.late_lr if (extras and args.stage == "coarse" and args.lr is None) else None)
    net = main_rec.pop("net")
    prof, shapes = main_rec.pop("prof"), main_rec.pop("shapes")
    elapsed = main_rec.pop("elapsed")
    lr_timed = main_rec["lr_timed"]

    # secondary headline: SDF MLP forward throughput (no-grad, 393216 samples per call)
    sdf_gs = 0.0
    with torch.no_grad():
        x = (torch.rand(393216 if not args.no_sdf_throughput else 0, 3, device=device) - 0.5) * 1.6
        for _ in range(2):
            net.sdf(x, 1.0)
        torch.cuda.synchronize(); s = time.perf_counter()
        for _ in range(5):
            net.sdf(x, 1.0)
        torch.cuda.synchronize(); sdf_gs = 5 * x.shape[0] / (time.perf_counter() - s) / 1e9 if x.shape[0] else 0.0
    V = main_rec["template_vertices"]
    hbm = {"peak_allocated_gb": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2), "reserved_gb": round(torch.cuda.memory_reserved(device) / 2 ** 30, 2),
           "device_allocations": torch.cuda.memory_stats(device).get("num_device_alloc", None)}
    del net
    gc.collect(); torch.cuda.empty_cache()

    def strip(r):
        for k in ("net", "prof", "shapes", "elapsed"):
            r.pop(k, None)
        gc.collect(); torch.cuda.empty_cache()
        return r

    fine_rec = None
    if world == 1 and args.stage == "coarse" and not args.no_fine and args.scaling == "weak" and not args.frames_per_gpu and not args.simulate_world:
        fine_rec = strip(run_stage("fine", args, rank, world, device, args.steps, args.warmup, max(args.settle // 2, 0), args.settle_low, False, timed_lr=lr_of("fine")))
        fine_rec["workload"] = "fine stage (1 frame x 6144 rays, 321x417x225 grid, remesh every 120: config.conf:39-48,113) at the MultiStepLR rate of epochs 80-129"

    # configs[2] (8 frames over 8 GPUs) cannot be run on one GPU; what CAN be measured here is both ends of its strong-scaling
    # ratio: the whole 8-frame step on one GPU, and the step ONE rank of 8 would run (1 frame, the replicated template term on 1/8 of
    # the vertices, no collectives).  Their quotient is the modelled 8-GPU speed-up (all-reduce of 15 MB over xGMI: ~0.2 ms, not in it).
    strong_rec = seg_rec = loose_rec = None
    if extras and args.stage == "coarse":
        short = dict(steps=min(args.steps, 20), warmup=args.warmup, settle=max(args.settle // 3, 0), settle_low=0)
        r8 = strip(run_stage("coarse", args, rank, world, device, short["steps"], short["warmup"], short["settle"], 0, False, frames=8))
        r1 = strip(run_stage("coarse", args, rank, world, device, short["steps"], short["warmup"], short["settle"], 0, False, frames=1))
        srdist.simulate_world((0, 8))
        try:
            rs = strip(run_stage("coarse", args, rank, world, device, short["steps"], short["warmup"], short["settle"], 0, False, frames=1))
            # the same rank's step with every collective LIVE through RCCL at world size 1 (both gradient buckets, the template-vertex
            # all-reduce, the pooled-mean weights, the count guards): what they cost in launches, copies and host time; the wire time of
            # the 8-GPU group is added analytically below
            rc, rc_error = None, None
            try:
                srdist.force_collectives(True, device)
                rc = strip(run_stage("coarse", args, rank, world, device, short["steps"], short["warmup"], short["settle"], 0, False, frames=1))
            except Exception as e:                     # (a one-rank process group needs a free local port; the record then says why it is missing)
                rc_error = f"{type(e).__name__}: {e}"[:300]
            finally:
                srdist.force_collectives(False)
        finally:
            srdist.simulate_world(None)
        am = lambda r: r["ms_per_step_remesh_amortised"] or r["ms_per_step"]       # one remesh per 30 iterations (the 20-step window holds one: 1/20)
        # wire time of the step's collectives on 8 GPUs, from the guide's xGMI figures (7 links x ~153 GB/s per GPU, point to point): a ring
        # all-reduce moves 2 (N-1)/N of the buffer over each rank's slowest link; the two gradient buckets (3.8 MB early -- under the
        # implicit-gradient pass, so not on the critical path -- and 11.4 MB main) and the template-vertex gradient (V x 3 floats)
        wire = lambda nbytes: 2.0 * 7.0 / 8.0 * nbytes / 153e9 * 1e3 + 0.03            # ms: bandwidth term + ~30 us of latency per collective
        wire_ms = round(wire(11.4e6) + wire(rs["template_vertices"] * 12) + 2 * 0.03, 3)   # main bucket + template all-reduce + two 4-byte weights
        strong_rec = {"workload": "configs[2]: 8 frames x 2048 rays per step (coarse stage, lr 1e-4); measured on ONE GPU; ms with the remesh amortised over its interval of 30",
                      "ms_8_frames_one_gpu": am(r8), "ms_1_frame_replicated_template_term": am(r1),
                      "ms_one_rank_of_8": am(rs),
                      "ms_one_rank_of_8_collectives_live_world1": None if rc is None else am(rc),
                      "collectives_live_error": rc_error,
                      "grad_allreduce_section_ms_world1": None if rc is None else (rc["diagnostics"].get("grad_allreduce_section_ms_per_step_by_rank") or [None])[0],
                      "modelled_wire_ms_8_gpus": wire_ms,
                      "ms_one_rank_of_8_with_collectives": None if rc is None else round(am(rc) + wire_ms, 3),
                      "modelled_speedup_8_gpus_with_collectives": None if rc is None else round(am(r8) / (am(rc) + wire_ms), 2),
                      "modelled_speedup_8_gpus": round(am(r8) / am(rs), 2),
                      "modelled_speedup_8_gpus_without_sharding": round(am(r8) / am(r1), 2),
                      "ms_in_the_20_step_window": {"8_frames": r8["ms_per_step"], "1_frame": r1["ms_per_step"], "one_rank_of_8": rs["ms_per_step"]},
                      "remesh_ms_each": rs["remesh"]["ms_each"],
                      "note": "one_rank_of_8 = 1 frame per step with mean|f(TmpVs)| evaluated on vertices 0::8 (dist.shard_world; the gradient all-reduce restores "
                              "the full mean); the remesh's SDF queries are sharded the same way on a real group (one all-gather per level) but run in full "
                              "here.  `ms_one_rank_of_8_collectives_live_world1` runs the same step with every collective live through RCCL at world size 1 (launches, "
                              "flat copies, host time: everything but the wire); `modelled_wire_ms_8_gpus` = ring all-reduce of the 11.4 MB main bucket and the "
                              "template-vertex gradient at 2 x 7/8 x bytes / 153 GB/s + 30 us each (the 3.8 MB early bucket runs under the implicit-gradient pass); "
                              "`ms_one_rank_of_8_with_collectives` is their sum"}
        seg_rec = seg3d_mc_513_record(device)
        from selfreconcode_amd.config import loose_config
        rl = strip(run_stage("coarse", args, rank, world, device, min(args.steps, 20), args.warmup, max(args.settle // 3, 0), 0, False,
                             scene=dict(H=1080, W=1080, conf=loose_config())))
        loose_rec = {"workload": "configs[4]: config_loose.conf (normal loss off in the coarse stage, focal length the only learnable camera tensor) at 1080 x 1080, "
                                 "coarse stage, 3 frames x 2048 rays, lr 1e-4; one GPU (the 8-GPU form shards frames as configs[2])",
                     "ms_per_step": rl["ms_per_step"], "iterations_per_s": round(1e3 / rl["ms_per_step"], 4), "rays_per_iter": rl["rays_per_iter"],
                     "rays_converged_frac": rl["rays_converged_frac"], "template_vertices": rl["template_vertices"], "remesh": rl["remesh"], "image": rl["image"]}
    if rank != 0:
        return
    FR, RAYS = main_rec["frames_per_gpu"], STAGES[args.stage]["rays"]
    FR_REF = STAGES[args.stage]["frames"]                      # frames of one reference iteration of this stage (config.conf batch_size)
    shared = os.environ.get("SR_ALL_RANKS_ON_DEVICE0") == "1" and world > 1
    flops_step = (prof.get("flops_total", 0.0) + prof.get("flops_total_tn", 0.0)) / max(args.steps, 1)      # every layer GEMM: forward, backward-data, weight-gradient, refiner chains
    epochs = {"coarse": "epochs 0-5 of config.conf (train.coarse, :28-34), all of them at lr 1e-4 (:16-27)", "fine": "epochs 12-200 of config.conf (train.fine, :43-48)"}[args.stage]
    out = {
        "metric": f"train.py-equivalent iterations/sec (540x540, {RAYS} rays/frame x {FR} frame{'s' if FR > 1 else ''} per GPU; {args.stage} stage at Adam lr {lr_timed:.3g})",
        "value": round(args.steps * world * FR / FR_REF / elapsed, 4), "unit": "iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_rec["ms_per_step"],
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "methodology": {"version": 2, "since": "round 4",
                        "note": "version 2 times the coarse stage at the learning rate config.conf runs it at (1e-4: 7-22 % of the rays converge); rounds 2-3 (version 1) "
                                "timed it at the late MultiStepLR rate (~75 % converge: a heavier colour / normal branch) -- compare BENCH_r02 / r03 with "
                                "`late_schedule_lr.ms_per_step` of this line, not with `value`"},
        "config": {"workload": f"configs[1]: female-3-casual-like 540x540, {args.stage} stage = {epochs}, Adam lr {lr_timed:.3g}, {FR} frame(s) x {RAYS} rays per rank, full iteration "
                               "(template deform + K=50 point-silhouette mask loss + template SGD, mesh rasteriser + seeds + Newton refiner, "
                               "eikonal/offset/def-regu/DCT/colour/normal, backward, implicit-grad propagation, Adam, one remesh in the timed window)"
                               + (f"; ONE RANK OF A SIMULATED {args.simulate_world}-RANK GROUP (template term on 1/{args.simulate_world} of the vertices, no collectives)" if args.simulate_world > 1 else ""),
                   "scaling_mode": (f"strong: {args.global_frames} frames per step split over {world} rank(s)" if args.scaling == "strong" else
                                    f"weak: {FR} frame(s) per rank and step, {FR * world} per step over {world} rank(s)")
                                   + ("; ALL RANKS SHARE DEVICE 0 (functional run, not a measurement)" if shared else ""),
                   "frames_per_step_all_ranks": FR * world, "frames_per_s": round(args.steps * world * FR / elapsed, 3),
                   "value_definition": f"frames per second / {FR_REF} (the frames of one reference iteration of the {args.stage} stage), whole job",
                   "stage": args.stage, "frames_per_gpu": FR, "rays_per_frame": RAYS, "image": main_rec["image"], "template_vertices": V,
                   "rays_per_iter": main_rec["rays_per_iter"], "rays_converged_frac": main_rec["rays_converged_frac"],
                   "observations": "uniform noise" if args.noise_observations else "rendered from the scene (render_frames), re-rendered after the settle phase",
                   "refiner": {"impl": args.refiner_impl, "stream_headline_pass": args.refiner_stream, "stream_instrumented_pass": "main"},
                   "ray_branch": ("colour / normal terms back-propagated inside forward() and the implicit-gradient pass, both on the side stream under the sampled terms and "
                                  "their backward (headline pass); main stream in the instrumented pass") if _eager_ray_branch() else "one backward of the total loss (SR_EAGER_RAY_BRANCH=0)",
                   "optimizer": {"impl": "torch.optim.Adam" if args.torch_adam else "FusedAdam (same update rule, one launch)", "lr_timed": lr_timed, "settle_iters_lr_1e-4": args.settle,
                                 "settle_iters_lr_timed": args.settle_low if lr_timed != 1e-4 else 0},
                   "rasterisation": "in-repo HIP kernels with pytorch3d 0.4.0 semantics (nearest-face mesh rasteriser -> FindSurfacePs; K=50 nearest-in-z "
                                    "point compositor); pytorch3d itself is third-party and not in the reference repository",
                   "parallelism": f"frame-parallel dp{world}: one flat grad all-reduce/step (overlapped with the implicit-gradient pass) + template-vertex grad all-reduce"
                                  + ("; replicated template term and remesh queries sharded over the ranks" if world > 1 else "")},
        "rccl": rccl,
        "remesh": main_rec["remesh"], "ms_per_step_remesh_amortised": main_rec["ms_per_step_remesh_amortised"],
        "ms_per_step_instrumented": main_rec["ms_per_step_instrumented"],
        "refiner_ms_per_step": main_rec["refiner_ms_per_step"],
        "diagnostics": dict(main_rec["diagnostics"], host_threads=srdist.PLACEMENT, note="host_threads = the CPUs rank 0's threads are confined to (selfreconcode_amd/affinity.py); the rest is taken inside the headline window: shader clock / socket power from sysfs half-way through and after the last step was "
                                                          "issued; host_issue_ms_per_step = host time to enqueue a step, host_ahead_ms_at_the_end = how long the final synchronisation waited "
                                                          "(~0 means the host paces the step)"),
        "regime_lr_config": main_rec.get("regime_lr_config"),
        "late_schedule_lr": main_rec.get("late_schedule_lr"),
        "fine_stage": fine_rec,
        "strong_scaling_model": strong_rec,
        "seg3d_mc_513": seg_rec,
        "loose1080": loose_rec,
        "sdf_mlp_gsamples_per_s": round(sdf_gs, 5),
        "hbm": hbm,
        "roofline": {"bound": "mfma", "kernel": "gemm_nt_kernel / mlp_layer_pair_kernel (fp32 MFMA 32x32x2 layer GEMM tile code with fused epilogue; the layer-pair kernel runs one layer of the refiner's two networks per launch on the device-side live-ray count), EVERY launch of the timed region with >= 128 rows and > 32 columns and every chain launch; "
                                                "launches issued while two streams feed the GPU are included (their event intervals can contain the other stream's kernels, "
                                                "which only lowers the figure); `achieved_alone` restricts to the launches that had the GPU to themselves",
                     "achieved": prof.get("tflops_all", 0.0), "peak": 157.3, "unit": "TFLOP/s", "frac": round(prof.get("tflops_all", 0.0) / 157.3, 4),
                     "launches": prof.get("launches_all"), "avg_launch_us": prof.get("avg_us_all"), "flop_per_launch": prof.get("avg_flop_all"),
                     "achieved_alone": prof.get("tflops"), "launches_alone": prof.get("launches"),
                     "achieved_launches_ge_64k_rows": prof.get("tflops_large"), "launches_ge_64k_rows": prof.get("launches_large"),
                     "whole_step_tflops": round(flops_step / (main_rec["ms_per_step"] * 1e-3) / 1e12, 3) if flops_step else None,
                     "whole_step_frac": round(flops_step / (main_rec["ms_per_step"] * 1e-3) / 1e12 / 157.3, 4) if flops_step else None,
                     "weight_gradient_gemm": {"kernel": "gemm_tn_kernel + slab_reduce_kernel (dW = Z^T A over the rows, deterministic slab reduction)",
                                              "achieved": prof.get("tflops_tn"), "launches": prof.get("launches_tn"), "avg_launch_us": prof.get("avg_us_tn"),
                                              "frac": round((prof.get("tflops_tn") or 0.0) / 157.3, 4)},
                     "flop_per_step": {"nt_and_chains": round(prof.get("flops_total", 0.0) / max(args.steps, 1)), "weight_gradient": round(prof.get("flops_total_tn", 0.0) / max(args.steps, 1))},
                     "note": "event pairs are recorded in a SECOND pass over the same K steps (ms_per_step_instrumented); the headline pass carries no events",
                     "traffic": None},
    }
    for name in ("r06_pmc_gemm_nt.json", "r05_pmc_gemm_nt.json", "r04_pmc_gemm_nt.json", "r03_pmc_gemm_nt.json", "r02_pmc_gemm_nt.json", "r01_pmc_gemm_nt.json"):       # PMC passes cannot run inside this process; latest committed collection
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.isfile(pmc):
            with open(pmc) as fh:
                t = json.load(fh)
            out["roofline"]["traffic"] = round(t["traffic_bytes_per_launch"])
            out["roofline"]["traffic_note"] = t.get("traffic_note") or (
                f"HBM bytes per launch of the wide-output layer GEMMs, calibrated FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc "
                f"passes (profiles/{name}; algorithmic bytes of the same launches: "
                f"{round(t.get('gemm_nt_wide', {}).get('algorithmic_bytes_per_launch', 0))})")
            break
    if shapes is not None and args.shape_log:
        with open(args.shape_log, "w") as fh:
            json.dump(shapes, fh, indent=1)
    if world == 1 and not args.no_cpu_baseline and not args.simulate_world:
        from selfreconcode_amd import affinity
        affinity.restore()                                    # the CPU leg gets the machine the process was started with
        out["cpu_baseline"] = cpu_baseline_record(args, device)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
