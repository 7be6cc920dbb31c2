#!/usr/bin/env python3
"""Headline benchmark: views/s of the texture-optimisation step (fwd + bwd into the texture gradient + fused
Adam) on synthetic ScanNet-shaped input. One process per GPU.

``python bench.py --gpus N`` started plainly (no WORLD_SIZE in the environment) launches N fresh rank processes itself
(``stylemesh_amd/launch.py``: decided before anything touches the GPU; the parent relays rank 0's JSON line and exits with
the ranks' worst return code); under ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` it is one of
the ranks. Views shard over the ranks and the texture gradient is all-reduced over RCCL before the update (SURVEY.md
section 8 e, BASELINE config 5); ``--replicas`` instead runs N INDEPENDENT scenes, one per GPU, no communicator (BASELINE
config 4) and reports the aggregate and the per-GPU rates. ``STYLEMESH_DIST_BACKEND=gloo`` runs either protocol with the
ranks sharing the visible GPU(s) (a functional check on a 1-GPU box, not a measurement).

Prints ONE JSON line on rank 0 (contract in the task description): whole-job views/s, the roofline of the
dominant kernel (the fp16x2-split MFMA implicit-GEMM conv - fp32 operands split into two fp16 parts, fp32 accumulate -
timed with HIP events on its launch stream inside the timed region; the fp32-MFMA conv in the f32_mode leg) and, at N = 1, the CPU baseline (the oracle, timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from stylemesh_amd.data import synthetic as S  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same table: dense bf16 MFMA
# algorithmic HBM bytes of the 25 grouped conv launches of one step: every input / output / gate plane and the
# layer's weights touched once
WORKLOADS = {
    # SURVEY.md section 8 d. c3 = scripts/train/optimize_texture_scannet_with_angle_and_depth.sh at 4096^2
    "c3": dict(tex=4096, level_hw=S.SCANNET_LEVEL_HW, view_hw=S.SCANNET_VIEW_HW, mode="multi", thr=30.0, angle=True,
               depth=True, index_repeat=20, min_depth=0.25, desc="ScanNet with_angle_and_depth, 4096^2 hier-4 texture, "
               "4 UV levels 256x341..784x1045, multi style pyramid, angle+depth reweighting"),
    # c2 = optimize_texture_scannet_only2D.sh at 2048^2
    "c2": dict(tex=2048, level_hw=[S.SCANNET_VIEW_HW], view_hw=S.SCANNET_VIEW_HW, mode="single", thr=3000.0,
               angle=False, depth=False, index_repeat=20, min_depth=0.25, desc="ScanNet only2D, 2048^2 hier-4 texture, "
               "1 UV level 256x341, single style pyramid"),
    # c5 = optimize_texture_matterport_with_angle_and_depth.sh shapes
    "c5": dict(tex=4096, level_hw=S.MATTERPORT_LEVEL_HW, view_hw=S.MATTERPORT_VIEW_HW, mode="multi", thr=40.0, angle=True,
               depth=True, index_repeat=100, min_depth=0.2, desc="Matterport with_angle_and_depth, 4096^2 hier-4 texture, "
               "4 UV levels 256x320..784x980"),
    # optimize_texture_scannet_with_angle.sh:3-20 - angle weighting on, depth scaling off, ONE UV level
    "with_angle": dict(tex=4096, level_hw=[S.SCANNET_VIEW_HW], view_hw=S.SCANNET_VIEW_HW, mode="multi", thr=30.0, angle=True,
                       depth=False, index_repeat=20, min_depth=0.25, desc="ScanNet with_angle, 4096^2 hier-4 texture, 1 UV "
                       "level 256x341, multi style pyramid, angle reweighting, no depth scaling"),
    # optimize_texture_scannet_dip.sh:3-20 - flat texture, no regulariser, 10-deep Gram history, a NEW view every step
    # (RepeatingSampler with index_repeat 1, data/abstract_dataset.py:498-512)
    "dip": dict(tex=4096, level_hw=[S.SCANNET_VIEW_HW], view_hw=S.SCANNET_VIEW_HW, mode="single", thr=3000.0, angle=False,
                depth=False, index_repeat=1, min_depth=0.25, n_layers=1, gram_mode="average", decay=15,
                loss_weights={"content": 7e1, "style": 1e-3, "tex_reg": 0.0},
                desc="ScanNet dip, 4096^2 1-layer texture, 1 UV level 256x341, single style pyramid, gram_mode average, "
                "index_repeat 1 (a view change every step)"),
}
LOSS_WEIGHTS = {"content": 7e1, "style": 1e-4, "tex_reg": 5e3}
STYLE_WEIGHTS = [1000., 1000., 10., 10., 1000.]
STYLE_HW = (1528, 1200)   # "The Scream" (styles/120styles/17.jpg) is 1200 x 1528 px


def engine_config(wl):
    from stylemesh_amd.runtime.engine import EngineConfig
    return EngineConfig(tex_w=wl["tex"], tex_h=wl["tex"], hierarchical=True, n_layers=wl.get("n_layers", 4),
                        style_weights=STYLE_WEIGHTS, angle_threshold=wl["thr"], style_pyramid_mode=wl["mode"],
                        gram_mode=wl.get("gram_mode", "current"), use_angle_weight=wl["angle"],
                        use_depth_scaling=wl["depth"], loss_weights=dict(wl.get("loss_weights", LOSS_WEIGHTS)),
                        learning_rate=1.0, decay_step_size=wl.get("decay", 3))


def make_views(wl, seeds):
    room = S.BoxRoom((12.0, 9.0, 3.0))   # large enough that all four UV levels are populated
    return [S.make_view(s, view_hw=wl["view_hw"], level_hw=wl["level_hw"], level_heights=[h for h, _ in wl["level_hw"]],
                        min_pyramid_depth=wl["min_depth"], room=room) for s in seeds]


def to_device(batch, dev):
    """Everything to the GPU except the view index (element 8: host-side bookkeeping, read by the host)."""
    return tuple([u.to(dev) for u in x] if isinstance(x, list) else (x.to(dev) if torch.is_tensor(x) and i != 8 else x)
                 for i, x in enumerate(batch))


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(wl, view_cpu, steps):
    """The oracle (CPU restatement of the reference path, parity-pinned against the reference's goldens) on the
    same workload / view / seeds, on this host's cores: the better of two thread counts (torch's CPU kernels stop
    scaling - and with all 256 hardware threads of the GPU node's host oversubscribe badly - far below the core count)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import stylemesh_oracle as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    env = os.environ.get("STYLEMESH_CPU_THREADS")
    from stylemesh_amd.runtime.hostcpu import effective_cpus
    quota = effective_cpus()      # what the container may really use (cgroup cpu.max), e.g. 16 of 256 visible threads
    counts = [int(env)] if env else sorted({max(1, min(avail, c)) for c in (32, int(quota))})
    cfg = O.OracleConfig(hierarchical=True, style_weights=STYLE_WEIGHTS, angle_threshold=wl["thr"], style_pyramid_mode=wl["mode"], gram_mode=wl.get("gram_mode", "current"),
                         use_angle_weight=wl["angle"], use_depth_scaling=wl["depth"],
                         loss_weights=dict(wl.get("loss_weights", LOSS_WEIGHTS)), learning_rate=1.0,
                         decay_step_size=wl.get("decay", 3))
    torch.set_num_threads(counts[0])
    pipe = O.OraclePipeline(S.seeded_vgg_state(0), S.style_image(1, *STYLE_HW), cfg, (wl["tex"], wl["tex"]),
                            n_layers=wl.get("n_layers", 4))
    best, tried = None, []
    for threads in counts:
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        pipe.training_step(view_cpu)   # warm-up (allocations, MKLDNN primitive caches)
        warm = time.perf_counter() - t0
        if warm > 30.0:   # keep the default bench run bounded: report the warm-up step itself
            rate, sample = 1.0 / warm, f"1 step of the same workload and view, no warm-up ({warm:.1f} s)"
        else:
            t0 = time.perf_counter()
            for _ in range(steps):
                pipe.training_step(view_cpu)
            dt = time.perf_counter() - t0
            rate, sample = steps / dt, f"{steps} step(s) of the same workload and view after 1 warm-up step ({dt:.1f} s)"
        tried.append({"threads": threads, "views_per_s": round(rate, 4)})
        if best is None or rate > best["value"]:
            best = {"value": rate, "unit": "views/s", "cores": threads, "kind": "port", "sample": sample}
    #  `cores` is the contract's field: the THREADS of torch's intra-op pool that gave the best rate -
    # under a CPU quota (cgroup cpu.max) of `cpu_quota` of the host's `hardware_threads_visible`: not physical cores
    best["cores_are"] = "threads of the CPU pool (not physical cores)"
    best["cpu_quota"] = float(quota)
    best["hardware_threads_visible"] = int(avail)
    best["thread_counts_tried"] = tried
    best["host_cpu"] = f"{cpu_model_string()} ({avail} hardware threads visible, CPU quota of the container {quota:g})"
    return best


def timed_leg(eng, schedule, args, wl, world, reducer, barrier, timer):
    """Warm-up + the timed region (barrier + synchronize on both sides). -> seconds of the timed region."""
    from stylemesh_amd.runtime import ops
    rep = wl["index_repeat"]
    solo = world == 1 or reducer is None   # (a replica is a single-rank job)

    def ahead(i):   # during a view's second step: the NEXT view's constants are computed on a side stream
        if solo and rep > 1 and i % rep == 1 and i - 1 + rep < len(schedule):
            eng.prepare_view(schedule[i - 1 + rep])

    def upcoming(i):   # index_repeat 1: every step changes the view - the next one is prepared beside THIS step
        return schedule[i + 1] if (solo and rep == 1 and i + 1 < len(schedule)) else None

    def step(i):
        ahead(i)
        eng.training_step(schedule[i], world_size=world if reducer is not None else 1, reducer=reducer,
                          new_view=(i % rep == 0), next_batch=upcoming(i))
    for i in range(args.warmup):
        step(i)
    ops.CONV_TIMER = timer
    if getattr(eng, "phase_timer", None) is not None:
        eng.phase_timer.enabled = True
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        if timer is not None:
            timer.enabled = (i - args.warmup) % args.timer_every == 0
        step(i)
    # (the last step's deferred sums, if the exchange defers any: applied INSIDE the timed region)
    eng.finish_exchange(world if reducer is not None else 1, reducer)
    barrier()
    dt = time.perf_counter() - t0
    ops.CONV_TIMER = None
    if getattr(eng, "phase_timer", None) is not None:
        eng.phase_timer.enabled = False
    return dt


def f32_leg(args, wl, cfg, schedule, dev, barrier):
    """The same workload on a second engine with v_mfma_f32_32x32x2_f32 convolutions and Gram kernels everywhere
    (STYLEMESH_CONV_MODE = STYLEMESH_GRAM_MODE = f32): a short leg of the same run, priced against the fp32-MFMA peak."""
    import copy
    from stylemesh_amd.runtime import ops
    from stylemesh_amd.runtime.engine import StepEngine
    saved = ops.CONV_MODE, ops.GRAM_MODE
    ops.CONV_MODE = ops.GRAM_MODE = "f32"
    try:
        eng = StepEngine(cfg, S.seeded_vgg_state(0), device=dev)
        eng.set_style_image(S.style_image(1, *STYLE_HW))
        a = copy.copy(args)
        a.steps, a.warmup, a.timer_every = args.f32_steps, 3, max(1, args.f32_steps // 2)
        timer = ops.KernelTimer()
        dt = timed_leg(eng, schedule, a, wl, 1, None, barrier, timer)
        n, ms, flops = timer.summary("f32")
        ach = flops / (ms * 1e-3) / 1e12
        return {"value": round(a.steps / dt, 3), "unit": "views/s", "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(1e3 * dt / a.steps, 3), "conv_tflops": round(ach, 2),
                "peak": PEAK_FP32_MFMA_TFLOPS, "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                "kernel": "conv3x3_mfma_kernel", "launches_timed": n}
    finally:
        ops.CONV_MODE, ops.GRAM_MODE = saved


def exchange_report(eng, comm, reducer, args):
    """N > 1: which communicator ran, and the event-timed exchange / update of the timed steps (rank 0's stream)."""
    rep = {"communicator": "own RCCL communicator (sm_comm_init / sm_allreduce_grad)"
           if type(comm).__name__ == "RcclComm" else "torch.distributed",
           "bytes_per_step": getattr(reducer, "last_bytes", None),
           "flagged_fraction_of_arena": None if not hasattr(reducer, "fraction") else round(reducer.fraction, 4)}
    if hasattr(comm, "info"):
        rep["rccl"] = comm.info()          # ranks RCCL saw, its version, link type of this rank's device to the others
    else:
        rep["process_group"] = {"backend": str(comm.get_backend()), "nranks": comm.get_world_size()}
    if getattr(reducer, "owner_aware", False):
        # critical = all-reduced before the update; deferred = in the background of the next step
        rep["deferred_exchange"] = {"enabled": bool(eng.use_deferred_exchange(reducer)),
                                    "critical_bytes_per_step": reducer.last_critical_bytes,
                                    "deferred_bytes_per_step": reducer.last_deferred_bytes,
                                    "shared_chunks": reducer.n_shared, "single_owner_chunks": reducer.n_single}
    rep["pipelined"] = bool(eng.use_pipelined_exchange(reducer))
    rep["pipelined_policy"] = {True: "forced on", False: "forced off"}.get(
        eng.pipeline_exchange, f"auto: from {eng.pipeline_min_bytes >> 20} MB of flagged chunks on")
    # DESIGN.md section 6's cost model beside the measurement, so that a driver-run scaling record checks itself:
    # t(N) = 2 (N - 1) / N * B / (L(N) * 76.8 GB/s * eta) + 2 B / 4 TB/s, L(N) = N - 1 xGMI links per GPU (point-to-point
    # mesh, 153.6 GB/s per link = 76.8 per direction), eta = the fraction RCCL realises on tens-of-MB messages (0.7
    # assumed), B = the bytes exchanged per step; the second term = gather + scatter of the flagged chunks
    nb, N = rep["bytes_per_step"], args.gpus
    if nb and N > 1:
        model = lambda eta: 1e3 * (2.0 * (N - 1) / N * nb / ((N - 1) * 76.8e9 * eta) + 2.0 * nb / 4e12)
        rep["model"] = {"formula": "2(N-1)/N * B / ((N-1) * 76.8 GB/s * eta) + 2 B / 4 TB/s", "B_bytes": nb, "N": N,
                        "predicted_exchange_ms": {"eta_0.6": round(model(0.6), 3), "eta_0.7": round(model(0.7), 3),
                                                  "eta_0.8": round(model(0.8), 3)},
                        "note": "xGMI figures: only meaningful for the own-RCCL communicator on separate GPUs"}
        if "deferred_exchange" in rep:
            cb, db = rep["deferred_exchange"]["critical_bytes_per_step"], rep["deferred_exchange"]["deferred_bytes_per_step"]
            mb = lambda b, eta: round(1e3 * (2.0 * (N - 1) / N * b / ((N - 1) * 76.8e9 * eta) + 2.0 * b / 4e12), 3)
            rep["model"]["predicted_critical_ms"] = {f"eta_{e}": mb(cb, e) for e in (0.6, 0.7, 0.8)}
            rep["model"]["predicted_deferred_ms_in_the_background"] = {f"eta_{e}": mb(db, e) for e in (0.6, 0.7, 0.8)}
    t = getattr(eng, "phase_timer", None)
    if t is not None:
        for tag, key in (("exchange", "exchange_ms"), ("update", "update_ms"), ("exchange+update", "exchange_update_ms")):
            n, ms, _ = t.summary(tag)
            if n:
                rep[key] = round(ms / n, 4)
                rep["steps_timed"] = n
    return rep


def scene_coverage_flags(eng, wl, n_views, seed=0):
    """Ever-touched flags of a whole scene: the union of the chunks the UV maps of ``n_views`` random poses in the bench's
    box room reach, rendered with the product's HIP rasteriser (65 us per frame) - what ``eng.touched`` holds after one
    epoch over such a scene."""
    from stylemesh_amd import render as R
    from stylemesh_amd.runtime import ops
    room = S.BoxRoom((12.0, 9.0, 3.0))
    mesh = R.box_room_mesh(room, device="cuda", subdiv=8)
    rng = np.random.default_rng(seed)
    L = room.size
    flags = torch.zeros_like(eng.touched)
    vh, vw = wl["view_hw"]
    for _ in range(n_views):
        pos = np.array([rng.uniform(0.8, L[0] - 0.8), rng.uniform(0.8, L[1] - 0.8), rng.uniform(1.0, 1.7)])
        K, c2w = S.camera_matrices(pos, rng.uniform(0, 2 * np.pi), rng.uniform(-0.35, 0.25), (vh, vw))
        for h, w in wl["level_hw"]:
            intr = np.array([K[0, 0] * w / vw, K[1, 1] * h / vh, (K[0, 2] + 0.5) * w / vw, (K[1, 2] + 0.5) * h / vh], np.float32)
            uv = R.render_maps(mesh, c2w, intr, (h, w), znear=0.05, zfar=50.0)[0]
            grid = (uv[..., :2] * 2.0 - 1.0).contiguous()
            ops.tex_touch_flags(eng.grads, eng.arena.g, grid, None, flags, eng.touched_log2)
    return flags


def late_epoch_leg(eng, schedule, args, wl, barrier):
    """The regime of a scene's LATER epochs: the ever-touched set of the sparse update is the whole
    scene's coverage instead of the 2-3 views the main leg has seen. Same engine, same schedule, timed the same way."""
    import copy
    flags = scene_coverage_flags(eng, wl, args.late_epoch_views)
    from stylemesh_amd.runtime import ops
    ops.flags_or(eng.touched, flags)
    eng._other_flags = None
    frac = float((eng.touched != 0).float().mean())
    a = copy.copy(args)
    a.warmup = 5
    dt = timed_leg(eng, schedule, a, wl, 1, None, barrier, None)
    return {"value": round(a.steps / dt, 3), "unit": "views/s", "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3), "ever_touched_fraction_of_arena": round(frac, 4),
            "views_seeded": args.late_epoch_views,
            "note": "ever-touched flags pre-seeded with the coverage of a whole scene (UV maps of that many random poses, "
                    "HIP rasteriser): the update's early half (side stream) walks all of it, the closing half the view's own chunks"}


def resident_views_leg(eng, schedule, args, wl, barrier):
    """A schedule that changes the view every step, from its SECOND epoch on: the per-view state of every view of the leg is
    resident in HBM (``viewplan.ResidentView``; first visit during the untimed warm-up) and a view change copies it back
    instead of recomputing it. Same engine, same steps, timed like the main leg."""
    import copy
    n_distinct = len({id(v) for v in schedule})
    eng.view_cache_gb = float(os.environ.get("STYLEMESH_VIEW_CACHE_GB", "0")) or 96.0
    a = copy.copy(args)
    a.warmup, a.steps = max(args.warmup, n_distinct + 2), args.resident_steps
    sched = [schedule[i % len(schedule)] for i in range(a.warmup + a.steps)]
    h0 = eng.view_cache_hits
    dt = timed_leg(eng, sched, a, wl, 1, None, barrier, None)
    out = {"value": round(a.steps / dt, 3), "unit": "views/s", "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(1e3 * dt / a.steps, 3), "distinct_views": n_distinct,
           "resident_mb_per_view": round(eng._resident_bytes / max(len(eng._resident), 1) / 2 ** 20, 2),
           "view_changes_served_from_hbm": eng.view_cache_hits - h0,
           "note": "every view change of the timed steps copies the view's state back from HBM (level maps, layer masks, "
                   "content targets, active lists, sorted scatter plan, touch flags: computed at the view's first visit, in "
                   "the warm-up) - the regime of epochs 2 .. 7 of a scene's schedule; the main leg computes every view change"}
    eng.view_cache_gb = 0.0
    return out


def many_views_leg(eng, wl, args, dev, barrier, good, first_seeds):
    """The same engine and workload over ELEVEN more views (one warm-up view, then ``many_views_steps`` timed steps = ten
    views at index_repeat steps per view): the main leg's 2-3 views are the first of the camera path and happen to carry 20 % more
    listed conv work than the average view - this leg is the rate a scene's schedule sees (compare
    scene_schedule.mean_views_per_s, measured through the CLI)."""
    import copy
    rep = wl["index_repeat"]
    n_views = 1 + (args.many_views_steps + rep - 1) // rep
    seeds = [good[v % len(good)] for v in range(n_views)]
    made = dict(first_seeds)
    todo = sorted(set(seeds) - set(made))
    made.update({s_: to_device(v, dev) for s_, v in zip(todo, make_views(wl, todo))})
    a = copy.copy(args)
    a.warmup, a.steps = rep, args.many_views_steps
    schedule = [made[seeds[(i // rep) % n_views]] for i in range(a.warmup + a.steps)]
    dt = timed_leg(eng, schedule, a, wl, 1, None, barrier, None)
    return {"value": round(a.steps / dt, 3), "unit": "views/s", "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3), "views": n_views - 1,
            "note": "same engine, timed like the main leg, no per-launch events"}


def measured_schedule(workload):
    """The fixed schedule of one scene MEASURED on the product path (tools/run_schedule.py: directory loader, MiniTrainer,
    the CLI's flags) - a committed record of a GPU-box run, the default bench run cannot afford 4 minutes of it."""
    for rnd in sorted((d for d in os.listdir(os.path.join(REPO, "profiles")) if d[:1] == "r" and d[1:].isdigit()), reverse=True):
        f = os.path.join(REPO, "profiles", rnd, f"schedule_{workload}.json")   # (the latest committed record)
        if os.path.exists(f):
            d = json.load(open(f))
            return {"measured_schedule_s": d.get("measured_schedule_s"), "measured_schedule_live": False, "steps": d.get("steps"),
                    "mean_views_per_s": d.get("mean_views_per_s"), "per_epoch_views_per_s": [e["views_per_s"] for e in d.get("per_epoch", [])],
                    "source": f"committed record profiles/{rnd}/{os.path.basename(f)}, not measured in this run (python -m stylemesh_amd.model.optimize on a {d.get('views')}-view "
                              "on-disk scene, texture exports and validation included)"}
    return None


def live_schedule_leg(args, wl):
    """BASELINE's second metric UNDER THIS RUN'S CLOCK: the scene's fixed schedule - ``--schedule-epochs``
    epochs (default: all 7) of index_repeat x 273 views - through ``python -m stylemesh_amd.model.optimize`` as a fresh
    child process on an on-disk synthetic scene the HIP rasteriser writes first (stylemesh_amd/schedule.py). The child is
    ended at ``--schedule-budget-s``; the epochs that finished by then are reported. Never fails the bench line: any error
    is reported as ``live_error`` and the committed record stays the only figure."""
    import shutil
    import tempfile
    from stylemesh_amd import schedule as SCH
    epochs_full = 7
    out = {"measured_schedule_live": False}
    root = tempfile.mkdtemp(prefix="stylemesh_scene_")
    try:
        heights = [256, 432, 608, 784] if args.workload == "c3" else [256]
        t0 = time.time()
        SCH.write_scene(root, "scene0000_00", 276, heights)
        torch.cuda.synchronize()
        t_scene = time.time() - t0
        cmd = SCH.cli_command(root, os.path.join(root, "logs"), args.workload, args.schedule_epochs, wl["index_repeat"], 4)
        stdout, stderr, rc, wall = SCH.run_cli(cmd, deadline_s=args.schedule_budget_s)
        epochs, loops, per_epoch = SCH.parse_epochs(stdout)
        if not epochs:
            out["live_error"] = f"no epoch finished (rc {rc}): {(stderr or stdout)[-400:]}"
            return out
        steps, secs = epochs[-1][1], epochs[-1][2]
        per_epoch_steps = wl["index_repeat"] * 273
        done = len(epochs)
        out.update({
            "measured_schedule_live": True,
            "live_epochs": done, "live_epochs_requested": args.schedule_epochs, "live_steps": steps,
            "live_seconds": round(secs, 1), "live_epoch_s": [e["seconds"] for e in per_epoch],
            "live_epoch_views_per_s": [e["views_per_s"] for e in per_epoch],
            "live_train_loop_views_per_s": [l[1] for l in loops],
            "live_mean_views_per_s": round(steps / secs, 2),
            "live_cli_wall_clock_s_incl_process_start_and_style_setup": round(wall, 1),
            "live_scene_write_s": round(t_scene, 1),
            "live_ended_by_budget": rc is None,
            "live_source": "this run: python -m stylemesh_amd.model.optimize as a fresh child on a 276-view on-disk scene "
                           "(HIP rasteriser), validation + texture exports included"})
        if done >= epochs_full and steps == epochs_full * per_epoch_steps:
            out["live_schedule_s"] = round(secs, 1)          # the WHOLE fixed schedule of one scene, measured
        else:                                                # fewer epochs: the remaining ones at the measured epochs' mean
            out["live_projected_schedule_s"] = round(secs / done * epochs_full, 1)
        if rc not in (0, None):
            out["live_error"] = f"the CLI ended with code {rc}: {stderr[-300:]}"
    except BaseException as e:      # (a leg of a benchmark line: report, do not raise)
        out["live_error"] = f"{type(e).__name__}: {e}"[:500]
    finally:
        shutil.rmtree(root, ignore_errors=True)
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=list(WORKLOADS))
    ap.add_argument("--replicas", action="store_true", help="BASELINE config 4: the N ranks optimise N INDEPENDENT scenes, "
                    "one per GPU - no communicator, no gradient exchange (a CPU-side barrier brackets the timed region); "
                    "reports the aggregate and the per-GPU views/s")
    ap.add_argument("--cpu-steps", type=int, default=4, help="oracle steps timed for cpu_baseline (0 = skip)")
    ap.add_argument("--no-conv-timer", action="store_true", help="skip the per-launch HIP events")
    ap.add_argument("--dense", action="store_true", help="run the VGG convs on every tile instead of only the tiles that "
                    "can influence the loss (stylemesh_amd/runtime/sparsity.py); results are identical")
    ap.add_argument("--graphs", action="store_true", help="replay the captured hipGraph of the step instead of launching "
                    "its ~150 kernels eagerly (measured: no gain, the step is GPU-bound; event-timed steps run eagerly)")
    ap.add_argument("--overlap-style", action="store_true", help="run the style branches (Gram -> loss -> Gram backward) "
                    "on a side stream concurrently with the conv trunk")
    ap.add_argument("--dense-allreduce", action="store_true", help="N > 1: all-reduce the whole gradient arena instead of "
                    "only the chunks the ranks' current views can touch")
    ap.add_argument("--atomic-scatter", action="store_true", help="texture scatter with the tiled atomic kernel (one launch "
                    "per UV level) instead of the sorted gather over the per-view plan")
    ap.add_argument("--pipeline-exchange", action="store_true", help="N > 1: all-reduce the (sparse) gradient in pieces "
                    "with the update of each arena range issued as its sums arrive (default: exchange, then update; "
                    "STYLEMESH_PIPELINE_EXCHANGE=1 selects it for the trainer)")
    ap.add_argument("--deferred-exchange", action="store_true", help="N > 1: only the chunks two or more ranks' views touch are "
                    "all-reduced before the update; single-owner chunks are updated by their owner at once and reach the "
                    "other ranks in a background all-reduce (runtime/distributed.py:OwnerAwareGradReducer; "
                    "STYLEMESH_DEFERRED_EXCHANGE=1 selects it for the trainer)")
    ap.add_argument("--mfma", choices=["split2", "f32"], default=None, help="matrix-core path of the conv and Gram kernels: "
                    "'split2' (default; fp16 MFMA on fp16x2-split operands, 3 partial products, fp32 accumulate) "
                    "or 'f32' (v_mfma_f32_32x32x2_f32 everywhere); same as "
                    "STYLEMESH_CONV_MODE / STYLEMESH_GRAM_MODE")
    ap.add_argument("--timer-every", type=int, default=7, help="HIP-event-time the conv launches of every n-th timed "
                    "step (event pairs around ~50 launches serialise the stream: timing every step costs 10-45 %% "
                    "of the throughput, so the roofline is sampled; 7 -> 3 timed steps of --steps 20, 6 of 40)")
    ap.add_argument("--f32-steps", type=int, default=10, help="N = 1: steps of the second, short leg that runs the same "
                    "workload with v_mfma_f32_32x32x2_f32 everywhere (reported as 'f32_mode'; 0 = skip)")
    ap.add_argument("--dense-adam", action="store_true", help="fused update over every texel instead of the chunks "
                    "some view has touched so far")
    ap.add_argument("--resident-steps", type=int, default=200, help="N = 1, workloads that change the view every step: timed "
                    "steps of the 'resident_views' leg (0 = skip)")
    ap.add_argument("--many-views-steps", type=int, default=200, help="N = 1: timed steps of the 'many_views' leg (the same "
                    "workload over ten more views instead of the main leg's first two or three; 0 = skip the leg)")
    ap.add_argument("--late-epoch-views", type=int, default=276, help="N = 1: views whose coverage seeds the ever-touched "
                    "set of the 'late_epoch' leg (0 = skip the leg)")
    ap.add_argument("--schedule-epochs", type=int, default=7, help="N = 1, workloads c3 / c2: epochs of the scene's fixed "
                    "schedule run LIVE through the CLI as a child process (7 = the whole schedule, ~3 min for c3; 0 = skip)")
    ap.add_argument("--schedule-budget-s", type=float, default=420.0, help="seconds after which the live schedule's child is "
                    "ended (the epochs finished by then are reported)")
    ap.add_argument("--launch-timeout", type=float, default=None, help="self-launched N > 1 runs: seconds after which the "
                    "rank processes are ended (default: none)")
    return ap.parse_args(argv)


def arena_checksum(eng):
    """Exact checksum of the texture values (sum of the fp32 bit patterns as int64, in slices: no 2 GB temporary)."""
    p = eng.arena.p
    tot = torch.zeros((), dtype=torch.int64, device=p.device)
    for lo in range(0, p.numel(), 1 << 26):
        tot += p[lo:lo + (1 << 26)].view(torch.int32).to(torch.int64).sum()
    return int(tot)


def gather_values(dist, backend, dev, values, dtype=torch.float64):
    """[values of rank 0, values of rank 1, ...] on every rank (a small all_gather on the process group's own device)."""
    t = torch.tensor(values, dtype=dtype, device=dev if backend == "nccl" else "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def main():
    args = parse_args()
    from stylemesh_amd import launch
    if launch.needs_launch(args.gpus, os.environ):
        # plain `python bench.py --gpus N`: N fresh rank processes, decided before this process touches the GPU (the
        # devices are counted from the KFD topology in sysfs, not through the runtime); the parent relays rank 0's line
        sys.exit(launch.launch(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:],
                               device_count=launch.kfd_gpu_count, timeout_s=args.launch_timeout))
    run(args)


def run(args):
    """One rank. Standard output carries ONE line - the result -, so everything else a rank's libraries write there (RCCL
    prints a version banner on stdout when a communicator comes up) is sent to standard error for the duration of the run."""
    sys.stdout.flush()
    out_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        line = _run(args)
    finally:
        sys.stdout.flush()
        os.dup2(out_fd, 1)
        os.close(out_fd)
    if line is not None:
        print(line, flush=True)


def _run(args):
    wl = WORKLOADS[args.workload]

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start it plainly (python bench.py --gpus N launches its "
                         "own ranks) or under torch.distributed.run with --nproc-per-node N")
    # STYLEMESH_DIST_BACKEND=gloo: functional test of the N > 1 protocol on a 1-GPU box (all ranks share cuda:0)
    backend = os.environ.get("STYLEMESH_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and n_dev < world:
        raise SystemExit(f"bench.py --gpus {world}: {n_dev} GPU(s) visible, one rank per GPU needs {world}")
    if backend != "nccl":
        local_rank %= max(n_dev, 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.replicas:
            backend = "gloo"                 # config 4: no RCCL anywhere - the process group only carries barriers / timings
            dist.init_process_group("gloo")
        elif backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
    sharded = world > 1 and not args.replicas   # views of ONE scene over the ranks, gradient exchange every step

    from stylemesh_amd.runtime import ops
    from stylemesh_amd.runtime.engine import StepEngine
    if args.mfma is not None:
        ops.CONV_MODE = args.mfma
        ops.GRAM_MODE = args.mfma   # the Gram kernels follow the conv mode

    cfg = engine_config(wl)
    eng = StepEngine(cfg, S.seeded_vgg_state(0), device=dev)
    eng.set_style_image(S.style_image(1, *STYLE_HW))
    eng.use_graphs = args.graphs
    eng.sparse_tiles = not args.dense
    eng.overlap_style = args.overlap_style
    if args.pipeline_exchange:
        eng.pipeline_exchange = True      # (else the engine's default: exchange-then-update; STYLEMESH_PIPELINE_EXCHANGE)
    eng.planned_scatter = not args.atomic_scatter
    eng.sparse_update = not args.dense_adam
    # Every timed leg below COMPUTES each view change: the engine's resident views (a revisited view's state copied back
    # from HBM instead of recomputed - the regime of a scene's later epochs) are switched off here and measured in a leg of
    # their own (`resident_views`), so that no number of this line contains re-used results unless its name says so.
    eng.view_cache_gb = 0.0

    # sharded: views of the scene shard over ranks - rank r takes views r, r + R, ... ; replicas: every rank has a scene
    # (a camera path) of its own; each view is repeated index_repeat times
    total_steps = args.warmup + args.steps
    n_views = (total_steps + wl["index_repeat"] - 1) // wl["index_repeat"]
    # seeds whose view populates every UV level (the 4-level worst case the FLOP figure is quoted for)
    good = (0, 2, 6, 7, 9, 11, 12, 14, 16, 18, 22, 23, 26, 27, 29, 30, 32, 33, 35, 36, 37, 38, 39)
    if args.replicas:
        seeds = [good[(v + 5 * rank) % len(good)] for v in range(max(1, n_views))]
    else:
        seeds = [good[(v * world + rank) % len(good)] for v in range(max(1, n_views))]
    distinct = {s_: v for s_, v in zip(sorted(set(seeds)), make_views(wl, sorted(set(seeds))))}   # (seeds repeat on long runs)
    views_cpu = [distinct[s_] for s_ in seeds]
    on_dev = {s_: to_device(v, dev) for s_, v in distinct.items()}
    views = [on_dev[s_] for s_ in seeds]
    schedule = [views[(i // wl["index_repeat"]) % len(views)] for i in range(total_steps)]

    from stylemesh_amd.runtime.distributed import make_comm, make_grad_reducer, make_sparse_grad_reducer
    # the gradient exchange runs on the product's own RCCL communicator (csrc/comm.hip); torch.distributed carries
    # the unique id, the barriers and the max-over-ranks of the timing
    comm = make_comm(dist, rank, world, dev) if sharded else None
    reducer = None
    if sharded:
        if args.dense_allreduce:
            reducer = make_grad_reducer(comm, world)
        elif args.deferred_exchange:
            # (a second communicator for the background exchange: collectives of one communicator are serialised)
            comm2 = make_comm(dist, rank, world, dev)
            reducer = make_sparse_grad_reducer(comm, world, rank=rank, deferred_dist=comm2)
            eng.deferred_exchange = True
        else:
            reducer = make_sparse_grad_reducer(comm, world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from stylemesh_amd.runtime.engine import trunk_stream
    main_stream = trunk_stream(dev)     # the step's trunk on a high-priority stream (the side streams keep normal priority)
    if main_stream is not None:
        torch.cuda.set_stream(main_stream)
    timer = None if args.no_conv_timer else ops.KernelTimer()
    if sharded:   # event-timed exchange and update of every step (two event pairs per step)
        eng.phase_timer = ops.KernelTimer()
        eng.phase_timer.enabled = False
    dt_own = timed_leg(eng, schedule, args, wl, world, reducer, barrier, timer)
    active_levels = [lv.index for lv in eng.view if lv.active]
    losses = eng.losses()
    touched_fraction = None if eng.touched is None else float(eng.touched.float().mean())

    dt, per_rank_dt, ranks_consistent = dt_own, [dt_own], None
    if world > 1:
        per_rank_dt = [v[0] for v in gather_values(dist, backend, dev, [dt_own])]
        dt = max(per_rank_dt)                         # the job's time = the slowest rank's
        if sharded:
            # every rank applied the same update to the same all-reduced gradient: the textures must be IDENTICAL, bit for bit
            sums = [v[0] for v in gather_values(dist, backend, dev, [arena_checksum(eng)], torch.int64)]
            ranks_consistent = len(set(sums)) == 1

    roofline = None
    if timer is not None:
        n_timed = len([i for i in range(args.steps) if i % args.timer_every == 0])
        n_all = ms_all = flops_all = 0.0      # every conv launch (the HBM-bound kernels carry 'hbm:' tags of their own)
        for t_ in ("f32", "split2"):
            n_, ms_, fl_ = timer.summary(t_)
            n_all, ms_all, flops_all = n_all + n_, ms_all + ms_, flops_all + fl_
        n_all = int(n_all)
        # the dominant kernel: the fp16x2-split conv when the engine runs in split2 mode, else the fp32-MFMA conv
        tag = "split2" if ops.CONV_MODE == "split2" else "f32"
        n, ms, flops = timer.summary(tag)
        ach = flops / (ms * 1e-3) / 1e12
        products = {"split2": 3}.get(tag)
        peak = PEAK_BF16_MFMA_TFLOPS / products if products else PEAK_FP32_MFMA_TFLOPS
        traffic, traffic_src = None, None   # HBM bytes per conv launch from the committed PMC pass of this workload
        rounds = sorted((d for d in os.listdir(os.path.join(REPO, "profiles")) if d[:1] == "r" and d[1:].isdigit()), reverse=True)
        for rnd in rounds:   # the latest committed pass (offline: PMC runs cannot be live inside bench.py)
            tf = os.path.join(REPO, "profiles", rnd, f"conv_traffic_{args.workload}_{tag}.json")
            if os.path.exists(tf):
                traffic, traffic_src = round(json.load(open(tf))["hbm_bytes_per_launch"]), f"profiles/{rnd}/{os.path.basename(tf)}"
                break
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "traffic_live": False,
                    "traffic_unit": f"HBM bytes per launch (committed PMC pass, {traffic_src}; not measured in this run)",
                    "algorithmic_bytes_per_launch": round(timer.bytes.get(tag, 0.0) / max(n, 1)),
                    "kernel": {"split2": "conv3x3_split_kernel (fp16 x 2)"}.get(tag, "conv3x3_mfma_kernel"),
                    "peak_basis": (f"16-bit dense MFMA peak 2500 TFLOP/s / {products} MFMA partial products per fp32 "
                                   "multiply-add (operands split into 2 fp16 parts, fp32 "
                                   "accumulate); achieved = algorithmic fp32 FLOPs / time, i.e. frac = executed MFMA FLOPs "
                                   "/ 2500. A pure-MFMA loop on random operands sustains 0.72 of 2500 on this part "
                                   "(profiles/r02/mfma_rate_operand_sweep.txt)") if products else
                                  "fp32 dense MFMA peak (v_mfma_f32_32x32x2_f32)",
                    "achieved_vs_fp32_mfma_peak": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                    "launches_timed": n, "timed_steps": n_timed, "avg_launch_us": round(1e3 * ms / n, 2),
                    "algorithmic_gflop_per_step": round(flops / n_timed / 1e9, 1),
                    "share_of_step_time": round(ms * 1e-3 / n_timed / (dt / args.steps), 3),
                    "all_conv_launches": {"launches_timed": n_all, "achieved": round(flops_all / (ms_all * 1e-3) / 1e12, 2),
                                          "algorithmic_gflop_per_step": round(flops_all / n_timed / 1e9, 1),
                                          "share_of_step_time": round(ms_all * 1e-3 / n_timed / (dt / args.steps), 3)}}

    roofline_hbm = None
    if timer is not None:
        # the step's HBM-bound kernels, event-timed in the same sampled steps: algorithmic bytes / time against the
        # measured device copy rate (6.29 TB/s: profiles/r02; the nominal HBM3E peak is 8 TB/s)
        shares = {"adam_closing(x flagged share)": None if eng._view_flags is None else float((eng._view_flags != 0).float().mean()),
                  "adam_early(x flagged share)": None if not eng._other_flags else float((eng._other_flags[1] != 0).float().mean())}
        roofline_hbm = {"copy_rate_TBps": 6.29, "peak_TBps": 8.0, "kernels": {}}
        for tag in sorted({r[3] for r in timer.records if str(r[3]).startswith("hbm:")}):
            n_l, ms_l, bytes_l = timer.summary(tag)
            name = tag[4:]
            share = shares.get(name, 1.0)
            if share is None or n_l == 0:
                continue
            extra = {}
            plan = getattr(eng, "_scatter_plan", None)
            if name == "scatter" and plan is not None and plan.level_hw is not None:
                # entries without weight sort to the tail and are skipped: only the live ones move bytes
                live = plan.live_share()
                pack = sum(h * w for h, w in plan.level_hw) * 28.0 * n_l
                bytes_l = pack + (bytes_l - pack) * live
                extra = {"live_entries_last_view": round(live, 4)}
            tbps = bytes_l * share / (ms_l * 1e-3) / 1e12
            roofline_hbm["kernels"][name] = {"launches_timed": n_l, "avg_us": round(1e3 * ms_l / n_l, 1), **extra,
                                            "algorithmic_MB_per_launch": round(bytes_l * share / n_l / 1e6, 1),
                                            "TBps": round(tbps, 2), "frac_of_copy_rate": round(tbps / 6.29, 3),
                                            **({"flagged_share_last_view": round(share, 4)} if name in shares else {})}
    line = None
    if rank == 0:
        value = world * args.steps / dt
        lw = wl.get("loss_weights", LOSS_WEIGHTS)
        if args.replicas and world > 1:
            parallelism = f"{world} independent scenes, one per GPU, no communicator (BASELINE config 4)"
        elif world > 1:
            parallelism = f"views sharded over {world} rank(s)" + (
                ", RCCL all-reduce of the 267 MB texture gradient per step" if args.dense_allreduce else
                f", RCCL all-reduce of the view-touched chunks of the texture gradient per step "
                f"({reducer.last_bytes / 1e6:.1f} of {4 * eng.arena.n / 1e6:.0f} MB on the last step)"
                + (f", in {reducer.n_pieces} pieces overlapped with the update" if eng.use_pipelined_exchange(reducer) else ""))
        else:
            parallelism = "views sharded over 1 rank(s)"
        out = {"metric": "views/sec (fwd+bwd into 4096^2 texture)" if wl["tex"] == 4096 else
               f"views/sec (fwd+bwd into {wl['tex']}^2 texture)", "value": round(value, 3), "unit": "views/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": {"split2": "f32 (fp16x2 split multiply, fp32 accumulate)"}.get(ops.CONV_MODE, "f32"),
               "dtype_note": {"split2": "all tensors, sums and the optimizer are fp32; the VGG convolutions multiply on the fp16 "
                              "matrix cores with every fp32 operand (scaled by a power of two from its tensor's recorded "
                              "max) split into 2 fp16 parts = 22 significand bits (3 partial products, fp32 accumulate) - an "
                              "emulation of the fp32 multiply, NOT native fp32: the strict-fp32 number of this run is f32_mode"}
               .get(ops.CONV_MODE, "fp32 throughout (v_mfma_f32_32x32x2_f32 convolutions)")
               + ("; error vs an fp64 convolution <= the fp32-MFMA kernel's (tests/test_kernels_gpu.py, tools/bench_conv.py); "
                  "STYLEMESH_CONV_MODE=f32 selects the fp32-MFMA kernel (the f32_mode leg)" if ops.CONV_MODE != "f32" else ""),
               "data": "synthetic",
               "config": {"workload": f"{args.workload}: {wl['desc']}",
                          "texture": f"{wl['tex']}x{wl['tex']} x {wl.get('n_layers', 4)} layer(s)",
                          "active_uv_levels": active_levels, "views_per_step": world,
                          "index_repeat": wl["index_repeat"], "style_image": f"synthetic {STYLE_HW[1]}x{STYLE_HW[0]}",
                          "loss_weights": lw, "gram_mode": wl.get("gram_mode", "current"),
                          "vgg_weights": "He-normal, seeded", "parallelism": parallelism},
               # BASELINE.json's second metric: the reference has no convergence criterion, a scene is trained for a
               # fixed schedule (SURVEY.md section 8 d): 7 epochs x index_repeat x 0.99 V views, V = 276 for ScanNet
               # scene0000_00 at every 20th frame. Projected from the measured rate (view changes are inside it).
               "scene_schedule": {"views": 273, "epochs": 7, "index_repeat": wl["index_repeat"],
                                  "steps": 7 * wl["index_repeat"] * 273,
                                  "projected_wall_clock_s": round(7 * wl["index_repeat"] * 273 / (value / world), 1),
                                  "note": "fixed schedule of one scene / measured views per second per scene"},
               "roofline": roofline, "roofline_hbm": roofline_hbm, "losses_last_step": {k: round(v, 3) for k, v in losses.items()},
               "fused_update": {"ever_touched_fraction_of_arena": None if touched_fraction is None else round(touched_fraction, 4),
                                "note": "the update skips 256-byte chunks no view has touched yet (exact for a "
                                        "zero-initialised texture); the fraction grows with the views of the scene"},
               "exchange": exchange_report(eng, comm, reducer, args) if sharded else None}
        if world > 1:
            out["per_rank_views_per_s"] = [round(args.steps / t, 3) for t in per_rank_dt]
            out["ranks_consistent"] = ranks_consistent
            out["launched_by"] = os.environ.get("STYLEMESH_LAUNCHED_BY", "external launcher (torch.distributed.run)")
            out["process_group_backend"] = backend
        sched = measured_schedule(args.workload)
        if sched is not None:
            out["scene_schedule"].update(sched)
        # (not under a profiler: its preloaded library would ride into the child processes)
        profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
        live_sched_wanted = world == 1 and args.schedule_epochs > 0 and args.workload in ("c3", "c2") and not profiled
        if world == 1 and args.f32_steps > 0 and ops.CONV_MODE != "f32" and args.mfma is None:
            out["f32_mode"] = f32_leg(args, wl, cfg, schedule, dev, barrier)
        if world == 1 and args.many_views_steps > 0:
            out["many_views"] = many_views_leg(eng, wl, args, dev, barrier, good, on_dev)
        if world == 1 and args.late_epoch_views > 0 and eng.touched is not None and not args.dense_adam:
            out["late_epoch"] = late_epoch_leg(eng, schedule, args, wl, barrier)
        if world == 1 and args.resident_steps > 0 and wl["index_repeat"] == 1:
            out["resident_views"] = resident_views_leg(eng, schedule, args, wl, barrier)
        if world == 1 and args.cpu_steps > 0:
            out["cpu_baseline"] = cpu_baseline(wl, views_cpu[(total_steps - 1) // wl["index_repeat"] % len(views)], args.cpu_steps)
        else:
            out["cpu_baseline"] = None
        if live_sched_wanted:
            # (last: the engine's legs are done; the child gets the GPU to itself apart from this process's idle buffers)
            del eng
            torch.cuda.empty_cache()
            live = live_schedule_leg(args, wl)
            ss = out["scene_schedule"]
            if live.get("live_schedule_s") is not None:
                # the whole fixed schedule ran in THIS run: it is the measured figure; the committed record of an earlier
                # round keeps a key of its own
                ss["committed_record"] = {k: ss.pop(k) for k in ("measured_schedule_s", "mean_views_per_s", "per_epoch_views_per_s", "source")
                                          if k in ss}
                ss["measured_schedule_s"] = live["live_schedule_s"]
                ss["mean_views_per_s"] = live["live_mean_views_per_s"]
                ss["per_epoch_views_per_s"] = live["live_epoch_views_per_s"]
                ss["source"] = live["live_source"]
            ss.update(live)
        line = json.dumps(out)
    if world > 1:
        if comm is not None and hasattr(comm, "destroy"):
            comm.destroy()
        dist.destroy_process_group()
    return line


if __name__ == "__main__":
    main()
