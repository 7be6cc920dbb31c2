"""the HIP rasteriser against the analytic ray caster of the synthetic box room (the
reference's OpenGL renderer cannot run here - DESIGN.md: parity unpinned for this row)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO
from gpu_util import require_gpu

from stylemesh_amd.data import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    require_gpu()
    from stylemesh_amd import render
    return render


def _gl_intrinsics(K):
    """``S.camera_matrices`` puts pixel centres at integer coordinates (cx = W/2 - 0.5); the rasteriser samples pixel
    (i, j) at (i + 0.5, j + 0.5), the OpenGL convention of the reference: cx_gl = cx + 0.5."""
    return np.array([K[0, 0], K[1, 1], K[0, 2] + 0.5, K[1, 2] + 0.5], dtype=np.float32)


def _interior(uv_ref, depth_ref):
    """Pixels whose 3x3 neighbourhood lies on one chart and has continuous depth (no silhouette / chart seam)."""
    d = torch.from_numpy(depth_ref)[None, None]
    u = torch.from_numpy(uv_ref[..., 0])[None, None]
    pad = lambda t: torch.nn.functional.pad(t, (1, 1, 1, 1), mode="replicate")
    unf = lambda t: torch.nn.functional.unfold(pad(t), 3).reshape(9, *t.shape[2:])
    dd, uu = unf(d), unf(u)
    ok = (dd.min(0).values > 0) & ((dd.max(0).values - dd.min(0).values) < 0.05 * dd.max(0).values) \
        & ((uu.max(0).values - uu.min(0).values) < 0.02)
    return ok


@pytest.mark.parametrize("seed,hw,subdiv", [(0, (96, 128), 1), (3, (240, 320), 4), (7, (61, 83), 16)])
def test_rasteriser_matches_analytic_ray_caster(R, seed, hw, subdiv):
    rng = np.random.default_rng(seed)
    room = S.BoxRoom((6.0, 4.5, 2.8))
    pos = room.size * np.array([0.5, 0.5, 0.5]) + rng.uniform(-0.8, 0.8, 3) * np.array([1, 1, 0.3])
    yaw, pitch = rng.uniform(0, 2 * np.pi), rng.uniform(-0.3, 0.3)
    uv_ref, cos_ref, depth_ref = room.render(pos, yaw, pitch, hw, window=False)
    K, c2w = S.camera_matrices(pos, yaw, pitch, hw)
    mesh = R.box_room_mesh(room, subdiv=subdiv)
    uv, ang, dep = R.render_maps(mesh, c2w, _gl_intrinsics(K), hw, znear=0.05, zfar=50.0)
    ok = _interior(uv_ref, depth_ref)
    assert float(ok.float().mean()) > 0.7
    hit = dep.cpu() > 0
    assert float((hit == torch.from_numpy(depth_ref > 0)).float().mean()) > 0.995      # closed room: everything hits
    np.testing.assert_allclose(dep.cpu()[ok].numpy(), depth_ref[ok.numpy()], rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(uv.cpu()[..., :2][ok].numpy(), uv_ref[..., :2][ok.numpy()], rtol=0, atol=2e-4)
    np.testing.assert_allclose(ang.cpu()[ok].numpy(), cos_ref[ok.numpy()], rtol=0, atol=2e-4)
    assert float(uv[..., 2].abs().max()) == 0.0


def test_near_plane_clipping_far_plane_and_empty(R):
    room = S.BoxRoom((6.0, 4.5, 2.8))
    hw = (80, 100)
    pos, yaw, pitch = np.array([0.12, 2.2, 1.4]), -np.pi / 2 + 0.3, 0.1      # 12 cm from a wall, looking at it obliquely
    uv_ref, cos_ref, depth_ref = room.render(pos, yaw, pitch, hw, window=False)
    K, c2w = S.camera_matrices(pos, yaw, pitch, hw)
    mesh = R.box_room_mesh(room, subdiv=2)
    uv, ang, dep = R.render_maps(mesh, c2w, _gl_intrinsics(K), hw, znear=0.1, zfar=3.0)
    dref = torch.from_numpy(depth_ref)
    visible = (dref >= 0.1 + 1e-3) & (dref <= 3.0 - 1e-3)
    clipped = (dref < 0.1 - 1e-3) | (dref > 3.0 + 1e-3)
    assert bool(visible.any()) and bool(clipped.any())
    ok = _interior(uv_ref, depth_ref) & visible
    np.testing.assert_allclose(dep.cpu()[ok].numpy(), depth_ref[ok.numpy()], rtol=2e-4, atol=1e-4)
    # a pixel whose nearest surface is clipped shows whatever lies behind it inside [near, far], or background
    behind = dep.cpu()[clipped]
    assert bool(((behind == 0) | ((behind >= 0.1) & (behind <= 3.0))).all())
    # no faces -> background everywhere
    empty = R.Mesh(np.zeros((3, 3)), np.zeros((3, 3)), np.zeros((3, 2)), np.zeros((0, 3), np.int32))
    uv0, ang0, dep0 = R.render_maps(empty, c2w, _gl_intrinsics(K), hw)
    assert float(uv0.abs().max()) == 0 and float(dep0.abs().max()) == 0 and float(ang0.abs().max()) == 0


def test_obj_loader_and_scene_writer_feed_the_loader(R, tmp_path):
    """OBJ round trip, and the files written by ``render_trajectory`` are readable by the ScanNet-layout loader."""
    room = S.BoxRoom((6.0, 4.5, 2.8))
    m = R.box_room_mesh(room, subdiv=1)
    v, n, t, f = (x.cpu().numpy() for x in (m.verts, m.normals, m.uvs, m.faces))
    obj = tmp_path / "room.obj"
    with open(obj, "w") as fh:
        for p in v: fh.write(f"v {p[0]} {p[1]} {p[2]}\n")
        for p in t: fh.write(f"vt {p[0]} {p[1]}\n")
        for p in n: fh.write(f"vn {p[0]} {p[1]} {p[2]}\n")
        for a, b, c in f + 1: fh.write(f"f {a}/{a}/{a} {b}/{b}/{b} {c}/{c}/{c}\n")
    m2 = R.load_obj(str(obj))
    hw = (48, 64)
    K, c2w = S.camera_matrices((3.0, 2.0, 1.4), 0.4, 0.0, hw)
    intr = np.array([K[0, 0], K[1, 1], K[0, 2] + 0.5, K[1, 2] + 0.5], dtype=np.float32)
    a, b = R.render_maps(m, c2w, intr, hw), R.render_maps(m2, c2w, intr, hw)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    Kgl = K.copy(); Kgl[0, 2] += 0.5; Kgl[1, 2] += 0.5
    out = tmp_path / "scene"
    R.render_trajectory(m, [c2w], ["0"], Kgl, (hw[1], hw[0]), str(out), heights=[24, 48])
    assert np.load(out / "uv_24" / "0.npy").shape == (24, 32, 3)
    assert np.load(out / "uv_48" / "0.npy").shape == (48, 64, 3)
    ang = np.load(out / "uv" / "0.angle.npy")
    assert ang.shape == (48, 64, 3) and float(ang.max()) <= 1.0 + 1e-6 and float(ang.min()) >= 0.0
    np.testing.assert_allclose(np.load(out / "uv" / "0.npy"), a[0].cpu().numpy())


def test_mipmapped_textured_rerender(R):
    """Trilinear mip-mapped lookup at the rasterised UV map: level of detail follows the screen-space footprint, a
    magnified view reproduces the texture function, a minified one its box-filtered pyramid."""
    room = S.BoxRoom((6.0, 4.5, 2.8))
    mesh = R.box_room_mesh(room, subdiv=2)
    T = 1024
    vv, uu = torch.meshgrid((torch.arange(T) + 0.5) / T, (torch.arange(T) + 0.5) / T, indexing="ij")
    f = lambda u, v: torch.stack([torch.sin(9 * u) * torch.cos(7 * v), u, v * v])         # smooth: bilinear-exact to 1e-3
    tex = f(uu, vv).cuda()
    mips = R.build_mipmaps(tex)
    assert [m.shape[1] for m in mips][:3] == [1024, 512, 256] and mips[-1].shape == (3, 1, 1)
    ref1 = torch.nn.functional.avg_pool2d(tex[None], 2)[0]
    assert float((mips[1] - ref1).abs().max()) < 1e-6
    hw = (120, 160)
    K, c2w = S.camera_matrices((3.0, 2.2, 1.4), 0.3, 0.0, hw)
    intr = np.array([K[0, 0], K[1, 1], K[0, 2] + 0.5, K[1, 2] + 0.5], dtype=np.float32)
    uv, _, dep = R.render_maps(mesh, c2w, intr, hw, znear=0.05, zfar=50.0)
    rgb, lod = R.sample_mipmapped(mips, uv, return_lod=True)
    hit = (dep > 0)
    # footprint: ~3 m away, 160 px over ~3.5 m of wall = 1/3 of a 1024-texel-wide atlas -> a few texels per pixel
    assert 0.5 < float(lod[hit].median()) < 3.5
    want = f(uv[..., 0], uv[..., 1])
    smooth = hit & (lod < float(lod[hit].median()) + 1.0)
    err = (rgb - want).abs()[:, smooth]
    assert float(err.mean()) < 2e-2 and float(err.max()) < 0.3      # box-filtered smooth function ~ the function
    assert float(rgb[:, ~hit].abs().max()) == 0.0 if bool((~hit).any()) else True
    # magnified: a 16x larger render of the same view -> footprint below one texel -> level 0, plain bilinear
    hw2 = (480, 640)
    K2, _ = S.camera_matrices((3.0, 2.2, 1.4), 0.3, 0.0, hw2)
    intr2 = np.array([K2[0, 0] * 4, K2[1, 1] * 4, (K2[0, 2] + 0.5), (K2[1, 2] + 0.5)], dtype=np.float32)  # 4x zoom
    uv2, _, dep2 = R.render_maps(mesh, c2w, intr2, hw2, znear=0.05, zfar=50.0)
    rgb2, lod2 = R.sample_mipmapped(mips, uv2, return_lod=True)
    hit2 = dep2 > 0
    assert float(lod2[hit2].median()) == 0.0
    want2 = f(uv2[..., 0], uv2[..., 1])
    assert float((rgb2 - want2).abs()[:, hit2].mean()) < 2e-3
