"""Host side of the HIP rasteriser ``sm_raster_maps``: the replacement of the reference's OpenGL renderer
(scripts/scannet/render_uv, driven by scripts/scannet/render_uvs.py) that produces the per-frame inputs of the
hot path - ``<frame>.npy`` UV maps at the pyramid resolutions, ``<frame>.angle.npy`` and
``<frame>.rendered_depth.npy`` - from a UV-parameterised mesh, the poses and the intrinsics.

GPU + HIP library only. The on-disk layout written by ``render_trajectory`` is the one
``stylemesh_amd.data.scannet.ScanNetSceneDataset`` (and the reference's ``ScanNetDataset``) reads.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from ..runtime import hip
from ..runtime.hip import lib, ptr


class Mesh:
    """Triangle mesh with per-vertex normal and texture coordinate (OBJ corners are unrolled by ``load_obj``)."""

    def __init__(self, verts, normals, uvs, faces, device="cuda"):
        self.verts = torch.as_tensor(verts, dtype=torch.float32).to(device).contiguous()
        self.normals = torch.as_tensor(normals, dtype=torch.float32).to(device).contiguous()
        self.uvs = torch.as_tensor(uvs, dtype=torch.float32).to(device).contiguous()
        self.faces = torch.as_tensor(faces, dtype=torch.int32).to(device).contiguous()
        V = self.verts.shape[0]
        if self.verts.shape != (V, 3) or self.normals.shape != (V, 3) or self.uvs.shape != (V, 2):
            raise ValueError("verts / normals [V,3] and uvs [V,2] expected")
        if self.faces.ndim != 2 or self.faces.shape[1] != 3:
            raise ValueError("faces [F,3] expected")
        if self.faces.numel() and (int(self.faces.min()) < 0 or int(self.faces.max()) >= V):
            raise ValueError("face index out of range")
        self._scratch = {}


def load_obj(path, device="cuda", flip_uvs=True) -> Mesh:
    """Wavefront OBJ with ``v`` / ``vt`` / ``vn`` / ``f a/b/c ...`` records (polygons are fanned); every face corner
    becomes its own vertex, as Assimp hands them to the reference renderer (include/model.h). Missing normals are
    replaced by the face normal. ``flip_uvs``: the reference imports with ``aiProcess_FlipUVs``
    (scripts/scannet/render_uv/include/model.h:57), i.e. the rendered UV maps hold (u, 1 - v) of the file's ``vt``
    records (the file's v = 0 is the BOTTOM row of the texture image, the UV map's v = 0 the top row)."""
    v, vt, vn, corners, faces = [], [], [], {}, []
    out_v, out_t, out_n = [], [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                v.append([float(x) for x in p[1:4]])
            elif p[0] == "vt":
                vt.append([float(x) for x in p[1:3]])
            elif p[0] == "vn":
                vn.append([float(x) for x in p[1:4]])
            elif p[0] == "f":
                idx = []
                for c in p[1:]:
                    a = (c.split("/") + ["", ""])[:3]
                    key = (int(a[0]), int(a[1]) if a[1] else 0, int(a[2]) if a[2] else 0)
                    if key not in corners:
                        corners[key] = len(out_v)
                        out_v.append(key[0]); out_t.append(key[1]); out_n.append(key[2])
                    idx.append(corners[key])
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    v = np.asarray(v, dtype=np.float32)
    res = lambda i, n: i - 1 if i > 0 else n + i          # OBJ indices are 1-based, negative = relative
    verts = v[[res(i, len(v)) for i in out_v]]
    if vt and all(out_t):
        uvs = np.asarray(vt, dtype=np.float64)[[res(i, len(vt)) for i in out_t]]
        if flip_uvs:
            uvs = np.stack([uvs[:, 0], 1.0 - uvs[:, 1]], 1)    # in double: exact inverse of save_obj's flip
        uvs = uvs.astype(np.float32)
    else:
        uvs = np.zeros((len(out_v), 2), np.float32)
    faces = np.asarray(faces, dtype=np.int32).reshape(-1, 3)
    if vn and all(out_n):
        normals = np.asarray(vn, dtype=np.float32)[[res(i, len(vn)) for i in out_n]]
    else:
        normals = np.zeros_like(verts)
        fn = np.cross(verts[faces[:, 1]] - verts[faces[:, 0]], verts[faces[:, 2]] - verts[faces[:, 0]])
        for k in range(3):
            np.add.at(normals, faces[:, k], fn)
        normals /= np.maximum(np.linalg.norm(normals, axis=1, keepdims=True), 1e-20)
    return Mesh(verts, normals, uvs, faces, device)


def save_obj(mesh: Mesh, path, flip_uvs=True):
    """Write ``mesh`` as a Wavefront OBJ that ``load_obj`` (same ``flip_uvs``) reads back unchanged: with ``flip_uvs``
    the file holds (u, 1 - v), the convention of the reference's input meshes."""
    v, n, t, f = (x.detach().cpu().numpy() for x in (mesh.verts, mesh.normals, mesh.uvs, mesh.faces))
    with open(path, "w") as fh:
        for p in v:
            fh.write(f"v {float(p[0])!r} {float(p[1])!r} {float(p[2])!r}\n")
        for p in t:
            fh.write(f"vt {float(p[0])!r} {(1.0 - float(p[1])) if flip_uvs else float(p[1])!r}\n")
        for p in n:
            fh.write(f"vn {float(p[0])!r} {float(p[1])!r} {float(p[2])!r}\n")
        for a, b, c in f + 1:
            fh.write(f"f {a}/{a}/{a} {b}/{b}/{b} {c}/{c}/{c}\n")


def project_points(cam2world, intrinsics4, points_world):
    """The projection ``sm_raster_maps`` implements, on the host (numpy, float64): world points [N,3] -> (x, y, depth)
    with x = fx X / Z + cx, y = fy Y / Z + cy in the camera frame of the ScanNet pose (x right, y down, z forward) and
    depth = Z. Pixel (i, j) of the output maps is the sample at (x, y) = (i + 0.5, j + 0.5); row 0 is the TOP row of
    the image. This equals the reference's OpenGL pipeline - view matrix of scannet_renderer.cpp:19-84, projection of
    include/util.h:11-35 (intrinsics normalised by the native image size), window transform of the render size,
    ``glReadPixels`` rows bottom-up with ``flip = 0``, ``LinearizeDepth`` of depth.frag - term by term
    (tests/test_reference_conventions.py restates that pipeline and compares)."""
    c2w = np.asarray(cam2world.detach().cpu() if torch.is_tensor(cam2world) else cam2world, dtype=np.float64)
    w2c = np.linalg.inv(c2w)
    p = np.asarray(points_world, dtype=np.float64)
    pc = p @ w2c[:3, :3].T + w2c[:3, 3]
    fx, fy, cx, cy = [float(v) for v in intrinsics4]
    return np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy, pc[:, 2]], 1)


def matterport_cam2world(extrinsics):
    """The ``cam2world`` that ``render_maps`` / ``project_points`` need for a Matterport image, from the 4x4 extrinsic
    matrix E of its ``.house`` record (row-major as in the file). The reference's Matterport renderer uses E DIRECTLY as
    the OpenGL view matrix - ``render(glm::mat4(1.0f), extr, projection)``, scripts/matterport/render_uv/src/renderer/
    mp_renderer.cpp:98,126 - with no look-at rebuild (the ScanNet renderer derives its view matrix from the pose's
    columns instead, scannet_renderer.cpp:19-84). An OpenGL eye frame looks down -z with +y up; this build's camera
    frame is x right, y towards larger row indices, z forward, and - as for ScanNet - file row j with ``flip = 0`` is
    GL window row j counted from the BOTTOM: camera = diag(1, 1, -1) eye, i.e. ``world -> camera = D E`` and
    ``cam2world = E^-1 D``. (The reference renders Matterport with ``flip = 1``, render_mipmap_matterport.py:17:
    ``render_trajectory(..., flip=True)`` reverses the rows as ``saveUV`` does.)"""
    E = np.asarray(extrinsics.detach().cpu() if torch.is_tensor(extrinsics) else extrinsics, dtype=np.float64).reshape(4, 4)
    D = np.diag([1.0, 1.0, -1.0, 1.0])
    return np.linalg.inv(E) @ D


def box_room_mesh(room, device="cuda", subdiv: int = 1) -> Mesh:
    """The synthetic box room (``stylemesh_amd.data.synthetic.BoxRoom``) as a mesh: 6 faces x ``subdiv``^2 quads, UVs
    from the room's chart layout, normals pointing into the room."""
    verts, normals, uvs, faces = [], [], [], []
    L = room.size
    for fi, (axis, side) in enumerate(room.faces):
        a0, a1 = [a for a in range(3) if a != axis]
        u0, v0, du, dv = room.chart(fi)
        n = np.zeros(3); n[axis] = 1.0 if side == 0 else -1.0
        base = len(verts)
        for j in range(subdiv + 1):
            for i in range(subdiv + 1):
                s, r = i / subdiv, j / subdiv
                p = np.zeros(3); p[axis] = L[axis] * side; p[a0] = s * L[a0]; p[a1] = r * L[a1]
                verts.append(p); normals.append(n); uvs.append([u0 + du * s, v0 + dv * r])
        for j in range(subdiv):
            for i in range(subdiv):
                q = base + j * (subdiv + 1) + i
                faces += [[q, q + 1, q + subdiv + 2], [q, q + subdiv + 2, q + subdiv + 1]]
    return Mesh(np.asarray(verts), np.asarray(normals), np.asarray(uvs), np.asarray(faces), device)


def scaled_intrinsics(K, native_wh, render_wh):
    """fx, fy, cx, cy of the render resolution: the reference's projection normalises the native intrinsics by the
    native image size (include/util.h:11-35) and the viewport stretches them to the render size."""
    K = np.asarray(K.detach().cpu() if torch.is_tensor(K) else K, dtype=np.float64)
    sx, sy = render_wh[0] / native_wh[0], render_wh[1] / native_wh[1]
    return np.array([K[0, 0] * sx, K[1, 1] * sy, K[0, 2] * sx, K[1, 2] * sy], dtype=np.float32)


def render_maps(mesh: Mesh, cam2world, intrinsics4, hw, znear=0.1, zfar=10.0):
    """One pose. ``cam2world`` [4,4] (ScanNet pose: x right, y down, z forward), ``intrinsics4`` = (fx, fy, cx, cy) at
    the render resolution in the OpenGL sample convention (pixel (i, j) sampled at (i + 0.5, j + 0.5)).
    Returns device tensors uv [H,W,3], angle [H,W], depth [H,W]."""
    H, W = hw
    dev = mesh.verts.device
    if dev.type != "cuda":
        raise RuntimeError("the rasteriser runs on the GPU only")
    c2w = torch.as_tensor(cam2world).detach().double().cpu()
    w2c = torch.linalg.inv(c2w).float().contiguous()
    intr = torch.as_tensor(np.asarray(intrinsics4, dtype=np.float32))
    key = (H, W)
    if key not in mesh._scratch:
        cap = 4096
        mesh._scratch[key] = (torch.empty(H * W, dtype=torch.int64, device=dev),
                              torch.empty(1 + 10 * cap, dtype=torch.float32, device=dev), cap)
    zbuf, big, cap = mesh._scratch[key]
    uv = torch.empty(H, W, 3, device=dev)
    ang = torch.empty(H, W, device=dev)
    dep = torch.empty(H, W, device=dev)
    hip.check(lib.sm_raster_maps(ptr(mesh.verts), ptr(mesh.normals), ptr(mesh.uvs), ptr(mesh.faces), mesh.faces.shape[0],
                                 w2c.data_ptr(), intr.data_ptr(), H, W, float(znear), float(zfar), ptr(zbuf), ptr(big), cap,
                                 ptr(uv), ptr(ang), ptr(dep), hip.stream()), "sm_raster_maps")
    return uv, ang, dep


def render_trajectory(mesh: Mesh, poses, names, K, native_wh, out_dir, heights, aspect=None, full_hw=None, flip=False):
    """Write what scripts/scannet/render_uvs.py produces for one scene: ``uv_<h>/<name>.npy`` (H,W,3 float32) for
    every pyramid height, and in ``uv/`` (at ``full_hw``, default the largest level) ``<name>.npy``,
    ``<name>.angle.npy`` and ``<name>.rendered_depth.npy`` (3 identical channels, as the reference's read-back).
    ``flip``: the renderer's ``<flip>`` argument (src/main.cpp:39-43, ``Renderer::saveUV``): rows written in reverse
    order. With ``flip = 0`` row 0 of the file is the top image row (the reference's projection does not negate y, so
    its bottom-up ``glReadPixels`` rows come out top-down)."""
    aspect = aspect if aspect is not None else native_wh[0] / native_wh[1]
    levels = [(int(h), int(round(h * aspect))) for h in heights]
    full_hw = tuple(full_hw) if full_hw is not None else levels[-1]
    os.makedirs(os.path.join(out_dir, "uv"), exist_ok=True)
    rows = (lambda a: np.ascontiguousarray(a[::-1])) if flip else (lambda a: a)
    for pose, name in zip(poses, names):
        for (h, w), hh in zip(levels, heights):
            d = os.path.join(out_dir, f"uv_{hh}")
            os.makedirs(d, exist_ok=True)
            uv, _, _ = render_maps(mesh, pose, scaled_intrinsics(K, native_wh, (w, h)), (h, w))
            np.save(os.path.join(d, f"{name}.npy"), rows(uv.cpu().numpy()))
        uv, ang, dep = render_maps(mesh, pose, scaled_intrinsics(K, native_wh, (full_hw[1], full_hw[0])), full_hw)
        np.save(os.path.join(out_dir, "uv", f"{name}.npy"), rows(uv.cpu().numpy()))
        np.save(os.path.join(out_dir, "uv", f"{name}.angle.npy"), rows(ang[..., None].expand(-1, -1, 3).contiguous().cpu().numpy()))
        np.save(os.path.join(out_dir, "uv", f"{name}.rendered_depth.npy"),
                rows(dep[..., None].expand(-1, -1, 3).contiguous().cpu().numpy()))


def build_mipmaps(image: torch.Tensor, min_size: int = 1):
    """2x2 box-filter pyramid of an RGB image [3,H,W] (``glGenerateMipmap`` of the reference's texture upload)."""
    lv = [image.detach().float().contiguous()]
    while min(lv[-1].shape[1:]) > min_size:
        _, h, w = lv[-1].shape
        dst = torch.empty(3, max(h // 2, 1), max(w // 2, 1), device=image.device)
        hip.check(lib.sm_mip_downsample(ptr(lv[-1]), ptr(dst), 3, h, w, hip.stream()), "sm_mip_downsample")
        lv.append(dst)
    return lv


def sample_mipmapped(mips, uv_map: torch.Tensor, return_lod=False):
    """Trilinear lookup of the pyramid at a rasterised UV map [H,W,3] -> RGB [3,H,W] (background pixels: 0)."""
    H, W = uv_map.shape[:2]
    out = torch.empty(3, H, W, device=uv_map.device)
    lod = torch.empty(H, W, device=uv_map.device) if return_lod else None
    hip.check(lib.sm_tex_sample_mip(hip.ptr_array(mips), hip.int_array([m.shape[2] for m in mips]),
                                    hip.int_array([m.shape[1] for m in mips]), len(mips), ptr(uv_map.contiguous()), H, W,
                                    ptr(out), ptr(lod), hip.stream()), "sm_tex_sample_mip")
    return (out, lod) if return_lod else out


def render_textured(mesh: Mesh, texture_image: torch.Tensor, cam2world, intrinsics4, hw, znear=0.1, zfar=10.0):
    """The reference renderer's textured re-render of one pose: rasterise, then mip-mapped lookup of ``texture_image``
    [3,Ht,Wt] (row 0 = v = 0)."""
    uv, _, _ = render_maps(mesh, cam2world, intrinsics4, hw, znear, zfar)
    return sample_mipmapped(build_mipmaps(texture_image), uv)
