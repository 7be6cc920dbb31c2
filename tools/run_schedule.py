"""BASELINE.json's second metric - "wall-clock to converge 1 scene" - MEASURED on the product path (run on the GPU box).

The reference has no convergence criterion: a scene is trained for the fixed schedule of its training script
(scripts/train/optimize_texture_scannet_with_angle_and_depth.sh:11-15: 7 epochs, index_repeat 20, train_split 0.99;
data/abstract_dataset.py:498-512 RepeatingSampler). ScanNet scene0000_00 has 5578 frames, every 20th is exported:
V = 276 -> 273 train views, 3 validation views, 7 x 20 x 273 = 38 220 steps (+ 21 validation steps, + 7 texture
exports).

What runs: (1) a synthetic scene in the reference's ON-DISK format (color jpg, 16-bit depth png, pose txt, intrinsics,
``uv_<h>/*.npy`` pyramid, ``uv/*.angle.npy``) is written with the product's HIP rasteriser (``render_trajectory``: the
f3 row) from random poses in the 12 x 9 x 3 m box room; (2) ``python -m stylemesh_amd.model.optimize`` - the CLI a user
runs - trains on it through the directory loader (background view prefetch + pinned upload), MiniTrainer and the
LightningModule mirror with the flags of the script, texture export included. Prints one JSON line with the
measured wall-clock, the per-epoch rates and the loader / trainer settings.

Usage: run_schedule.py [--workload c3|c2] [--views 276] [--epochs 7] [--index-repeat 20] [--out out.json]
"""
import argparse
import json
import os
import re
import shutil
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

from stylemesh_amd import schedule as SCH  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=list(SCH.FLAGS))
    ap.add_argument("--views", type=int, default=276)
    ap.add_argument("--epochs", type=int, default=7)
    ap.add_argument("--index-repeat", type=int, default=20)
    ap.add_argument("--num-workers", type=int, default=4, help="0 = no background prefetch (decode inside the loop)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    root = tempfile.mkdtemp(prefix="stylemesh_scene_")
    heights = [256, 432, 608, 784] if args.workload == "c3" else [256]
    t0 = time.time()
    sp = SCH.write_scene(root, "scene0000_00", args.views, heights)
    torch.cuda.synchronize()
    t_scene = time.time() - t0
    size_gb = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(sp) for f in fs) / 1e9
    log_dir = os.path.join(root, "logs")
    cmd = SCH.cli_command(root, log_dir, args.workload, args.epochs, args.index_repeat, args.num_workers)
    stdout, stderr, rc, wall = SCH.run_cli(cmd)
    print("\n".join(l for l in stdout.splitlines() if l.startswith(("epoch ", "fit:", "loader:", "set_view:", "host ms"))),
          file=sys.stderr)
    if rc != 0:
        print(stdout[-3000:], stderr[-3000:], file=sys.stderr)
        raise SystemExit(rc)
    epochs, loops, per_epoch = SCH.parse_epochs(stdout)
    fit = re.search(r"fit: ([\d.]+) s", stdout)
    ld = re.search(r"loader: decode ([\d.]+) s .* blocked ([\d.]+) s", stdout)
    hm = re.search(r"host ms per step: (.*)", stdout)
    sv = re.search(r"set_view: (.*)", stdout)
    n_train = int(0.99 * args.views)
    tex = [f for f in os.listdir(os.path.join(log_dir, "lightning_logs/version_0")) if f.endswith(".jpg")]
    out = {"workload": args.workload, "views": args.views, "train_views": n_train, "epochs": args.epochs,
           "index_repeat": args.index_repeat, "steps": epochs[-1][1] if epochs else None,
           "measured_schedule_s": round(epochs[-1][2], 1) if epochs else None,
           "fit_seconds_incl_setup": None if fit is None else float(fit.group(1)),
           "cli_wall_clock_s_incl_process_start_and_style_setup": round(wall, 1),
           "mean_views_per_s": round(epochs[-1][1] / epochs[-1][2], 2) if epochs else None,
           "per_epoch": per_epoch,
           "train_loop_views_per_s": [l[1] for l in loops], "train_loop_s": [l[0] for l in loops],
           "validation_and_export_s": [l[2] for l in loops], "scene_write_s": round(t_scene, 1), "scene_size_gb": round(size_gb, 2),
           "scene_source": "HIP rasteriser (render_trajectory), ScanNet directory layout, jpg / png / npy files",
           "loader": f"ScanNetSingleSceneDataModule, prefetch thread {'on' if args.num_workers > 0 else 'off'}, pinned upload one view ahead",
           "loader_decode_s": None if ld is None else float(ld.group(1)),
           "training_loop_blocked_on_loader_s": None if ld is None else float(ld.group(2)),
           "trainer_host_ms_per_step": None if hm is None else hm.group(1),
           "set_view_host": None if sv is None else sv.group(1),
           "texture_exports": len(tex), "command": " ".join(cmd[1:]).replace(root, "<scene-root>")}
    print(json.dumps(out))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)
    if not args.keep:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
