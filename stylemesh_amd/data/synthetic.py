"""Seeded synthetic scenes, style images and VGG weights.

Neither the ScanNet / Matterport datasets nor ``vgg_conv.pth`` are available (SURVEY.md section 8 d),
so every workload is restated as synthetic input of the same shapes:

* a *box room* seen from seeded camera poses: every pixel ray is intersected with the six faces of an
  axis-aligned room; each face owns an axis-aligned chart rectangle of the UV atlas, so the per-view
  UV maps are piecewise projective (spatially coherent - random per-pixel maps are wiped out by the
  reference's 3x3 mask erosion), ``cos(theta)`` comes from the face normal, depth is metric camera-z.
  One face carries a rectangular *window* that is rendered as background (uv = 0) to exercise masks.
  This is the same per-view data the reference's offline OpenGL renderer writes
  (``scripts/scannet/render_uv/shader/{uvmap,angle,depth}.frag``; formats in SURVEY.md section 2.2).
* ``rgb`` = low-pass noise in [0,1]; the style image = low-pass noise of the requested size.
* VGG-19 conv weights = He-normal from ``np.random.default_rng(seed)`` with the reference's
  ``state_dict`` key names (``conv{b}_{i}.{weight,bias}``, ``content_and_style_losses.py:11-26``).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import view_contract as vc

# (name, C_in, C_out) in network order - reference content_and_style_losses.py:11-26
VGG_CONVS = [
    ("conv1_1", 3, 64), ("conv1_2", 64, 64),
    ("conv2_1", 64, 128), ("conv2_2", 128, 128),
    ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("conv3_4", 256, 256),
    ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), ("conv4_4", 512, 512),
    ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512), ("conv5_4", 512, 512),
]

SCANNET_VIEW_HW = (256, 341)
SCANNET_LEVEL_HW = [(256, 341), (432, 576), (608, 811), (784, 1045)]
MATTERPORT_VIEW_HW = (256, 320)
MATTERPORT_LEVEL_HW = [(256, 320), (432, 540), (608, 760), (784, 980)]


def seeded_vgg_state(seed: int = 0, bias_std: float = 0.1) -> dict:
    """Random-init VGG-19 conv stack as a ``state_dict`` of torch fp32 tensors (He-normal weights)."""
    rng = np.random.default_rng(seed)
    state = {}
    for name, cin, cout in VGG_CONVS:
        std = np.sqrt(2.0 / (9 * cin))
        state[f"{name}.weight"] = torch.from_numpy(
            (rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) * np.float32(std)))
        state[f"{name}.bias"] = torch.from_numpy(
            (rng.standard_normal((cout,), dtype=np.float32) * np.float32(bias_std)))
    return state


def smooth_noise(rng: np.random.Generator, c: int, h: int, w: int, cells: int = 8) -> np.ndarray:
    """Low-pass noise in [0,1], shape (c,h,w): bilinear up-sampling of a coarse random lattice."""
    gh = max(2, cells)
    gw = max(2, int(round(cells * w / h)))
    g = torch.from_numpy(rng.random((1, c, gh, gw), dtype=np.float32))
    x = F.interpolate(g, size=(h, w), mode="bilinear", align_corners=True)[0]
    fine = torch.from_numpy(rng.random((c, h, w), dtype=np.float32))
    return (0.85 * x + 0.15 * fine).clamp(0, 1).numpy()


def style_image(seed: int, h: int, w: int) -> torch.Tensor:
    """Synthetic style image ``[3,h,w]`` already passed through ``pre()`` (reference optimize.py:118-126)."""
    rng = np.random.default_rng(seed)
    return vc.pre(torch.from_numpy(smooth_noise(rng, 3, h, w, cells=12)))


class BoxRoom:
    """Axis-aligned room [0,Lx]x[0,Ly]x[0,Lz] whose six faces tile a 3x2 UV atlas."""

    def __init__(self, size=(6.0, 4.5, 2.8), margin: float = 0.02):
        self.size = np.asarray(size, dtype=np.float64)
        self.margin = margin
        # face f: (axis, side); charts laid out on a 3x2 grid of the unit square
        self.faces = [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1)]

    def chart(self, f: int):
        cx, cy = f % 3, f // 3
        m = self.margin
        return (cx / 3 + m, cy / 2 + m, 1 / 3 - 2 * m, 1 / 2 - 2 * m)  # u0, v0, du, dv

    def render(self, cam_pos, yaw, pitch, hw, fov_deg=60.0, window=True):
        """Ray-cast one view at resolution ``hw``. Returns uv(H,W,3) f32, cos(H,W) f32, depth(H,W) f32."""
        H, W = hw
        f = 0.5 * H / np.tan(np.radians(fov_deg) / 2)
        ys, xs = np.meshgrid((np.arange(H) + 0.5 - H / 2) / f, (np.arange(W) + 0.5 - W / 2) / f, indexing="ij")
        d_cam = np.stack([xs, ys, np.ones_like(xs)], -1)  # camera looks along +z, y down
        cyaw, syaw, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        Ry = np.array([[cyaw, 0, syaw], [0, 1, 0], [-syaw, 0, cyaw]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        # world: x, y horizontal, z up; camera y-down maps to world -z
        cam_to_world = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=np.float64) @ Ry @ Rx
        d = d_cam @ cam_to_world.T
        o = np.asarray(cam_pos, dtype=np.float64)
        t_best = np.full((H, W), np.inf)
        face_best = np.full((H, W), -1, dtype=np.int64)
        for fi, (axis, side) in enumerate(self.faces):
            plane = self.size[axis] * side
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (plane - o[axis]) / d[..., axis]
            ok = (t > 1e-6) & (t < t_best)
            p = o + t[..., None] * d
            for a in range(3):
                if a != axis:
                    ok &= (p[..., a] >= 0) & (p[..., a] <= self.size[a])
            t_best = np.where(ok, t, t_best)
            face_best = np.where(ok, fi, face_best)
        hit = face_best >= 0
        t_safe = np.where(hit, t_best, 1.0)
        p = o + t_safe[..., None] * d
        uv = np.zeros((H, W, 3), dtype=np.float32)
        cos = np.zeros((H, W), dtype=np.float32)
        dnorm = np.linalg.norm(d, axis=-1)
        for fi, (axis, side) in enumerate(self.faces):
            sel = face_best == fi
            if not sel.any():
                continue
            a0, a1 = [a for a in range(3) if a != axis]
            u0, v0, du, dv = self.chart(fi)
            s = p[..., a0] / self.size[a0]
            r = p[..., a1] / self.size[a1]
            uv[..., 0] = np.where(sel, u0 + du * s, uv[..., 0])
            uv[..., 1] = np.where(sel, v0 + dv * r, uv[..., 1])
            cos = np.where(sel, np.abs(d[..., axis]) / dnorm, cos)
            if window and fi == 3:  # a window in one wall: rendered as background
                win = sel & (s > 0.35) & (s < 0.6) & (r > 0.35) & (r < 0.75)
                uv[win] = 0
                cos = np.where(win, 0, cos)
                hit = hit & ~win
        depth = np.where(hit, t_safe, 0.0).astype(np.float32)  # d_cam z == 1 -> t is camera-z depth
        uv[~hit] = 0
        return uv, cos.astype(np.float32), depth


def camera_matrices(cam_pos, yaw, pitch, hw, fov_deg=60.0):
    """(K [4,4], cam2world [4,4]) fp32 of the pin-hole camera ``BoxRoom.render`` casts rays from: pixel (x, y)
    looks along ((x - cx) / f, (y - cy) / f, 1) with cx = W/2 - 0.5, cy = H/2 - 0.5 (pixel centres), camera z
    forward / y down - the ScanNet pose / intrinsics convention of the reference's evaluation scripts."""
    H, W = hw
    f = 0.5 * H / np.tan(np.radians(fov_deg) / 2)
    cyaw, syaw, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Ry = np.array([[cyaw, 0, syaw], [0, 1, 0], [-syaw, 0, cyaw]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    R = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=np.float64) @ Ry @ Rx
    c2w = np.eye(4)
    c2w[:3, :3] = R
    c2w[:3, 3] = np.asarray(cam_pos, dtype=np.float64)
    K = np.eye(4)
    K[0, 0] = K[1, 1] = f
    K[0, 2], K[1, 2] = W / 2 - 0.5, H / 2 - 0.5
    return K.astype(np.float32), c2w.astype(np.float32)


def make_view(seed: int, view_hw=SCANNET_VIEW_HW, level_hw=None, level_heights=None,
              min_pyramid_depth: float = 0.25, room: BoxRoom | None = None, use_depth_in_mask=True,
              depth_holes: bool = True):
    """One synthetic posed view as the collated B = 1 batch tuple (see ``view_contract.assemble_batch``).

    ``level_hw``: list of UV-pyramid resolutions (default: just ``view_hw``);
    ``level_heights``: the ``levels`` array of the depth-level computation (default: the heights).
    """
    rng = np.random.default_rng(seed)
    room = room or BoxRoom()
    level_hw = list(level_hw) if level_hw is not None else [tuple(view_hw)]
    if level_heights is None:
        level_heights = [h for h, _ in level_hw]
    L = room.size
    pos = np.array([rng.uniform(0.8, L[0] - 0.8), rng.uniform(0.8, L[1] - 0.8), rng.uniform(1.0, 1.7)])
    yaw = rng.uniform(0, 2 * np.pi)
    pitch = rng.uniform(-0.35, 0.25)
    uvs = [room.render(pos, yaw, pitch, hw)[0] for hw in level_hw]
    _, cos, depth = room.render(pos, yaw, pitch, tuple(view_hw))
    if depth_holes:  # sensor drop-outs: a few rectangular holes of zero depth
        H, W = view_hw
        for _ in range(2):
            hh, ww = int(rng.integers(H // 16, H // 6)), int(rng.integers(W // 16, W // 6))
            y0, x0 = int(rng.integers(0, H - hh)), int(rng.integers(0, W - ww))
            depth[y0:y0 + hh, x0:x0 + ww] = 0
    rgb = torch.from_numpy(smooth_noise(rng, 3, view_hw[0], view_hw[1]))
    K, c2w = camera_matrices(pos, yaw, pitch, tuple(view_hw))
    return vc.assemble_batch(rgb, depth, uvs, cos, level_heights, min_pyramid_depth, idx=seed,
                             use_depth_in_mask=use_depth_in_mask, extrinsics=torch.from_numpy(c2w).double()[None],
                             intrinsics=torch.from_numpy(K).double()[None])


def trajectory_poses(n: int, room: BoxRoom, seed: int = 0):
    """``n`` camera poses (pos, yaw, pitch) along a smooth path through the room: neighbouring frames overlap, as the
    consecutive frames of a scan do (what the reprojection-error pairs of the evaluation rely on)."""
    rng = np.random.default_rng(seed)
    L = room.size
    p0 = np.array([0.3 * L[0], 0.35 * L[1], 1.4]) + rng.uniform(-0.1, 0.1, 3)
    p1 = np.array([0.7 * L[0], 0.6 * L[1], 1.5]) + rng.uniform(-0.1, 0.1, 3)
    yaw0 = rng.uniform(0, 2 * np.pi)
    out = []
    for i in range(n):
        t = i / max(n - 1, 1)
        out.append((p0 + t * (p1 - p0), yaw0 + 1.2 * t, 0.05 * np.sin(3.0 * t)))
    return out
