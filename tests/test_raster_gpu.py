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
    obj = tmp_path / "room.obj"
    R.save_obj(m, str(obj))                      # (u, 1 - v) in the file: the reference's meshes, model.h:57
    m2 = R.load_obj(str(obj))
    assert torch.equal(m2.uvs[m2.faces.long()], m.uvs[m.faces.long()])
    raw = R.load_obj(str(obj), flip_uvs=False)
    assert torch.allclose(raw.uvs[:, 1], 1.0 - m2.uvs[:, 1], atol=1e-7) and torch.equal(raw.uvs[:, 0], m2.uvs[:, 0])
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


def test_pixel_row_depth_and_uv_conventions_of_the_reference_pipeline(R):
    """f3 pin on the kernel itself: small camera-facing triangles around known camera-space points, a ScanNet-style pose
    and off-centre intrinsics at a render size different from the native one. ``project_points`` - shown equal to the
    reference's OpenGL pipeline (view / projection matrices, window transform, bottom-up read-back with flip = 0,
    LinearizeDepth) in tests/test_reference_conventions.py - names the pixel each triangle must cover, its depth and,
    after ``aiProcess_FlipUVs``, its uv."""
    rng = np.random.default_rng(11)
    native_wh, hw = (1296, 968), (256, 343)
    K = np.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 1170.2, 1170.2, 0.52 * 1296, 0.47 * 968
    intr = R.scaled_intrinsics(K, native_wh, (hw[1], hw[0]))
    a, b = 0.4, -0.3
    Rm = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ \
        np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    c2w = np.eye(4)
    c2w[:3, :3], c2w[:3, 3] = Rm, [0.7, -0.2, 1.1]
    n = 60
    px = np.stack([rng.uniform(8, hw[1] - 8, n), rng.uniform(8, hw[0] - 8, n)], 1)          # target (x, y) in pixels
    Z = rng.uniform(0.5, 6.0, n)
    cam = np.stack([(px[:, 0] - intr[0 + 2]) / intr[0] * Z, (px[:, 1] - intr[3]) / intr[1] * Z, Z], 1)
    d = 2.5 * Z / intr[0]                                                                   # ~2.5 px half-size
    tri = np.stack([cam + np.stack([-d, -d, 0 * d], 1), cam + np.stack([2 * d, -d, 0 * d], 1),
                    cam + np.stack([-d, 2 * d, 0 * d], 1)], 1)                              # [n,3,3], plane z = Z
    world = tri.reshape(-1, 3) @ Rm.T + c2w[:3, 3]
    uv_file = np.repeat(rng.uniform(0.05, 0.95, (n, 2)), 3, 0)
    obj = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"probe_{os.getpid()}.obj")
    with open(obj, "w") as fh:
        for p in world: fh.write(f"v {float(p[0])!r} {float(p[1])!r} {float(p[2])!r}\n")
        for t in uv_file: fh.write(f"vt {float(t[0])!r} {float(t[1])!r}\n")
        for k in range(n): fh.write(f"f {3*k+1}/{3*k+1} {3*k+2}/{3*k+2} {3*k+3}/{3*k+3}\n")
    mesh = R.load_obj(obj)                                                                   # FlipUVs: (u, 1 - v)
    os.remove(obj)
    uv, ang, dep = (t.cpu().numpy() for t in R.render_maps(mesh, c2w, intr, hw))
    proj = R.project_points(c2w, intr, cam @ Rm.T + c2w[:3, 3])
    np.testing.assert_allclose(proj[:, :2], px, atol=1e-3)
    col, row = np.floor(proj[:, 0]).astype(int), np.floor(proj[:, 1]).astype(int)           # row 0 = top image row
    hit = dep[row, col] > 0
    assert hit.all()
    nearest = np.array([Z[(np.abs(px[:, 0] - px[k, 0]) < 9) & (np.abs(px[:, 1] - px[k, 1]) < 9)].min() for k in range(n)])
    front = nearest == Z                                                                    # not hidden by a closer probe
    np.testing.assert_allclose(dep[row, col][front], Z[front], rtol=2e-4)
    want_uv = np.stack([uv_file[::3, 0], 1.0 - uv_file[::3, 1]], 1)
    np.testing.assert_allclose(uv[row, col][front][:, :2], want_uv[front], atol=2e-5)
    assert float((dep > 0).mean()) < 0.05                                                    # background stays 0
