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
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

from stylemesh_amd.data import synthetic as S  # noqa: E402
from stylemesh_amd import render as R  # noqa: E402


def write_scene(root, scene, n_views, heights, seed=0):
    """ScanNet layout under <root>/train/images/<scene>/ (stylemesh_amd/data/scannet.py), maps by the HIP rasteriser."""
    room = S.BoxRoom((12.0, 9.0, 3.0))
    mesh = R.box_room_mesh(room, device="cuda", subdiv=8)
    sp = os.path.join(root, "train/images", scene)
    for d in ("color", "depth", "pose"):
        os.makedirs(os.path.join(sp, d), exist_ok=True)
    native_hw = (480, 640)
    rng = np.random.default_rng(seed)
    L = room.size
    poses, names = [], []
    K = None
    for n in range(n_views):
        pos = np.array([rng.uniform(0.8, L[0] - 0.8), rng.uniform(0.8, L[1] - 0.8), rng.uniform(1.0, 1.7)])
        K, c2w = S.camera_matrices(pos, rng.uniform(0, 2 * np.pi), rng.uniform(-0.35, 0.25), native_hw)
        poses.append(c2w)
        names.append(str(n))
        np.savetxt(os.path.join(sp, "pose", f"{n}.txt"), c2w, fmt="%.6f", delimiter=" ")
        rgb = S.smooth_noise(rng, 3, 120, 160)
        Image.fromarray((np.clip(rgb, 0, 1).transpose(1, 2, 0) * 255 + 0.5).astype(np.uint8)).resize(
            (native_hw[1], native_hw[0]), Image.BILINEAR).save(os.path.join(sp, "color", f"{n}.jpg"), quality=90)
    # OpenGL sample convention of the rasteriser: pixel (i, j) sampled at (i + 0.5, j + 0.5)
    Kgl = np.array(K, dtype=np.float64)
    Kgl[0, 2] += 0.5
    Kgl[1, 2] += 0.5
    R.render_trajectory(mesh, poses, names, Kgl, (native_hw[1], native_hw[0]), sp, heights, full_hw=native_hw)
    for n in range(n_views):   # "sensor" depth: the rendered depth in millimetres
        d = np.load(os.path.join(sp, "uv", f"{n}.rendered_depth.npy"))[:, :, 0]
        Image.fromarray(np.round(d * 1000).astype(np.uint16)).save(os.path.join(sp, "depth", f"{n}.png"))
        os.remove(os.path.join(sp, "uv", f"{n}.rendered_depth.npy"))
    with open(os.path.join(sp, "_info.txt"), "w") as f:
        f.write(f"colorHeight = {native_hw[0]}\ncolorWidth = {native_hw[1]}\nfx_color = {Kgl[0, 0]}\nfy_color = {Kgl[1, 1]}\n"
                f"mx_color = {Kgl[0, 2]}\nmy_color = {Kgl[1, 2]}\n")
    return sp


FLAGS = {   # scripts/train/optimize_texture_scannet_{with_angle_and_depth,only2D}.sh, texture size of BASELINE's configs
    "c3": ["--texture_size", "4096,4096", "--style_pyramid_mode", "multi", "--angle_threshold", "30", "--pyramid_levels", "4"],
    "c2": ["--texture_size", "2048,2048", "--style_pyramid_mode", "single", "--angle_threshold", "3000",
           "--pyramid_levels", "1", "--no_depth_scaling", "--no_angle_weight"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=list(FLAGS))
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
    sp = write_scene(root, "scene0000_00", args.views, heights)
    torch.cuda.synchronize()
    t_scene = time.time() - t0
    size_gb = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(sp) for f in fs) / 1e9
    log_dir = os.path.join(root, "logs")
    cmd = [sys.executable, "-m", "stylemesh_amd.model.optimize", "--gpus", "1", "--root_path", root, "--dataset", "scannet",
           "--resize_size", "256", "--min_images", "1", "--max_images", "1000", "--scene", "scene0000_00",
           "--hierarchical", "--hierarchical_layers", "4", "--loss_weight", "content=7e1", "--loss_weight", "style=1e-4",
           "--style_weights=1000,1000,10,10,1000", "--loss_weight", "tex_reg=5e3", "--vgg_gatys_model_path", "random:0",
           "--learning_rate", "1", "--decay_step_size", "3", "--log_images_nth", "5000", "--batch_size", "1",
           "--max_epochs", str(args.epochs), "--train_split", "0.99", "--val_split", "0.01", "--sampler_mode", "repeat",
           "--index_repeat", str(args.index_repeat), "--save_texture", "--split_mode", "sequential",
           "--num_workers", str(args.num_workers), "--style_image_path", "synthetic:1:1528x1200", "--gram_mode", "current",
           "--min_pyramid_depth", "0.25", "--min_pyramid_height", "256", "--default_root_dir", log_dir] + FLAGS[args.workload]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True)
    wall = time.time() - t0
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith(("epoch ", "fit:", "loader:", "set_view:", "host ms"))),
          file=sys.stderr)
    if r.returncode != 0:
        print(r.stdout[-3000:], r.stderr[-3000:], file=sys.stderr)
        raise SystemExit(r.returncode)
    epochs = [(int(m.group(1)), int(m.group(2)), float(m.group(3)))
              for m in re.finditer(r"epoch (\d+): (\d+) steps, ([\d.]+) s", r.stdout)]
    loops = [(float(m.group(1)), float(m.group(2)), float(m.group(3)))
             for m in re.finditer(r"train loop ([\d.]+) s = ([\d.]+) steps/s, validation \+ epoch-end hooks ([\d.]+) s", r.stdout)]
    fit = re.search(r"fit: ([\d.]+) s", r.stdout)
    ld = re.search(r"loader: decode ([\d.]+) s .* blocked ([\d.]+) s", r.stdout)
    hm = re.search(r"host ms per step: (.*)", r.stdout)
    sv = re.search(r"set_view: (.*)", r.stdout)
    n_train = int(0.99 * args.views)
    per_epoch = []
    prev_steps, prev_t = 0, 0.0
    for e, steps, t in epochs:
        per_epoch.append({"epoch": e, "steps": steps - prev_steps, "seconds": round(t - prev_t, 2),
                          "views_per_s": round((steps - prev_steps) / max(t - prev_t, 1e-9), 2)})
        prev_steps, prev_t = steps, t
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
