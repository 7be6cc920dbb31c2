"""BASELINE.json's second metric - "wall-clock to converge 1 scene" - measured on the product path.

The reference has no convergence criterion: a scene is trained for the fixed schedule of its training script
(scripts/train/optimize_texture_scannet_with_angle_and_depth.sh:11-15: 7 epochs, index_repeat 20, train_split 0.99;
data/abstract_dataset.py:498-512 RepeatingSampler). ScanNet scene0000_00 has 5578 frames, every 20th is exported:
V = 276 -> 273 train views, 3 validation views, 7 x 20 x 273 = 38 220 steps (+ 21 validation steps, + 7 texture exports).

``write_scene`` writes a synthetic scene in the reference's ON-DISK format (color jpg, 16-bit depth png, pose txt,
intrinsics, ``uv_<h>/*.npy`` pyramid, ``uv/*.angle.npy``) with the product's HIP rasteriser (``render_trajectory``: the f3
row) from random poses in the 12 x 9 x 3 m box room; ``run_cli`` runs ``python -m stylemesh_amd.model.optimize`` - the
entry point a user runs (model/optimize.py:28-165) - on it as a FRESH CHILD PROCESS through the directory loader, MiniTrainer
and the LightningModule mirror with the flags of the script, texture export included, and reads the per-epoch lines as
they are printed (a deadline ends the child and keeps the epochs that finished).

Used by ``bench.py`` (the live ``scene_schedule`` leg) and ``tools/run_schedule.py`` (the committed full-schedule records).
"""
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np
import torch
from PIL import Image

from .data import synthetic as S
from . import render as R

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def cli_command(root, log_dir, workload, epochs, index_repeat, num_workers):
    return [sys.executable, "-m", "stylemesh_amd.model.optimize", "--gpus", "1", "--root_path", root, "--dataset", "scannet",
            "--resize_size", "256", "--min_images", "1", "--max_images", "1000", "--scene", "scene0000_00",
            "--hierarchical", "--hierarchical_layers", "4", "--loss_weight", "content=7e1", "--loss_weight", "style=1e-4",
            "--style_weights=1000,1000,10,10,1000", "--loss_weight", "tex_reg=5e3", "--vgg_gatys_model_path", "random:0",
            "--learning_rate", "1", "--decay_step_size", "3", "--log_images_nth", "5000", "--batch_size", "1",
            "--max_epochs", str(epochs), "--train_split", "0.99", "--val_split", "0.01", "--sampler_mode", "repeat",
            "--index_repeat", str(index_repeat), "--save_texture", "--split_mode", "sequential",
            "--num_workers", str(num_workers), "--style_image_path", "synthetic:1:1528x1200", "--gram_mode", "current",
            "--min_pyramid_depth", "0.25", "--min_pyramid_height", "256", "--default_root_dir", log_dir] + FLAGS[workload]


def run_cli(cmd, deadline_s=None):
    """Run the CLI as a fresh child, collecting its standard output line by line. ``deadline_s``: seconds after which the
    child (this exact process) is ended; what it printed until then is kept. Returns (stdout, stderr tail, return code or
    None when the deadline ended it, wall-clock seconds)."""
    t0 = time.time()
    proc = subprocess.Popen(cmd, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, bufsize=1)
    lines, err = [], []
    readers = [threading.Thread(target=lambda: lines.extend(iter(proc.stdout.readline, "")), daemon=True),
               threading.Thread(target=lambda: err.extend(iter(proc.stderr.readline, "")), daemon=True)]
    for t in readers:
        t.start()
    timed_out = False
    try:
        proc.wait(timeout=deadline_s)
    except subprocess.TimeoutExpired:
        timed_out = True
        proc.terminate()
        try:
            proc.wait(timeout=20)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.wait()
    for t in readers:
        t.join(timeout=5)
    return "".join(lines), "".join(err)[-3000:], (None if timed_out else proc.returncode), time.time() - t0


def parse_epochs(stdout):
    """[{epoch, steps, seconds, views_per_s}] and the train-loop figures from the trainer's per-epoch lines."""
    epochs = [(int(m.group(1)), int(m.group(2)), float(m.group(3)))
              for m in re.finditer(r"epoch (\d+): (\d+) steps, ([\d.]+) s", stdout)]
    loops = [(float(m.group(1)), float(m.group(2)), float(m.group(3)))
             for m in re.finditer(r"train loop ([\d.]+) s = ([\d.]+) steps/s, validation \+ epoch-end hooks ([\d.]+) s", stdout)]
    per_epoch, prev_steps, prev_t = [], 0, 0.0
    for e, steps, t in epochs:
        per_epoch.append({"epoch": e, "steps": steps - prev_steps, "seconds": round(t - prev_t, 2),
                          "views_per_s": round((steps - prev_steps) / max(t - prev_t, 1e-9), 2)})
        prev_steps, prev_t = steps, t
    return epochs, loops, per_epoch
