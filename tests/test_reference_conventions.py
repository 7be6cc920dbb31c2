"""Known-answer pins of the conventions on either side of the hot path (VERDICT r1
item 8), hand-derived from the reference's sources - its OpenGL renderer and cv2 cannot run in this image, so the
maths is restated here line by line and compared with what the product implements. CPU only.

f3 (scripts/scannet/render_uv): view matrix of src/renderer/scannet_renderer.cpp:19-84, projection of
include/util.h:11-35, the window transform + bottom-up ``glReadPixels`` + ``flip`` of src/renderer/renderer.cpp
(``saveUV``), ``LinearizeDepth`` of shader/depth.frag, ``aiProcess_FlipUVs`` of include/model.h:57.
f1 (data/abstract_dataset.py:291-311): cv2.INTER_LINEAR for the depth, cv2.INTER_NEAREST for the angle map,
Pillow NEAREST for the mask."""
import numpy as np
import pytest
import torch

from stylemesh_amd.data import view_contract as vc


# ------------------------------------------------------------------ f3: the reference's OpenGL pipeline, restated
def ref_view_matrix(extr):
    """scannet_renderer.cpp:19-84. ``extr`` = the pose file's 4x4 camera-to-world matrix M (rows as in the file). The
    parser hands it to glm so that ``extr[c]`` is COLUMN c of M: eye = column 3, right / up / look = columns 0 / 1 / 2."""
    M = np.asarray(extr, dtype=np.float64)
    eye, right, up, look = M[:3, 3], M[:3, 0], M[:3, 1], M[:3, 2]
    right, up, look = (v / np.linalg.norm(v) for v in (right, up, look))
    V = np.eye(4)
    V[0, :3], V[0, 3] = right, -right @ eye          # glm view[c][r]: row 0 = right
    V[1, :3], V[1, 3] = up, -up @ eye
    V[2, :3], V[2, 3] = -look, look @ eye            # "third row --> need to multiply with -1"
    return V


def ref_projection(K, width, height, n=0.1, f=10.0):
    """util.h:11-35 (written there in row-major and transposed for glm): rows of the clip-space matrix."""
    return np.array([[2 * K[0, 0] / width, 0, -(2 * (K[0, 2] / width) - 1), 0],
                     [0, 2 * K[1, 1] / height, -(2 * (K[1, 2] / height) - 1), 0],
                     [0, 0, -(f + n) / (f - n), -2 * f * n / (f - n)],
                     [0, 0, -1, 0]], dtype=np.float64)


def ref_pipeline(extr, K, native_wh, render_wh, pts, n=0.1, f=10.0):
    """World points -> (window x, window y, saved row index with flip = 0, linearised depth)."""
    P, V = ref_projection(K, native_wh[0], native_wh[1], n, f), ref_view_matrix(extr)
    clip = (P @ V @ np.concatenate([pts, np.ones((len(pts), 1))], 1).T).T
    ndc = clip[:, :3] / clip[:, 3:4]
    xw = (ndc[:, 0] + 1) / 2 * render_wh[0]          # glViewport(0, 0, W, H)
    yw = (ndc[:, 1] + 1) / 2 * render_wh[1]
    zw = (ndc[:, 2] + 1) / 2                          # gl_FragCoord.z, default depth range
    z = zw * 2 - 1
    lin = (2 * n * f) / (f + n - z * (f - n))         # depth.frag LinearizeDepth
    # glReadPixels row j = window rows [j, j + 1) from the BOTTOM; saveUV with flip = 0 writes row j of the file = GL row j
    return xw, yw, np.floor(yw).astype(int), lin


def random_pose(rng):
    a, b, c = rng.uniform(-0.6, 0.6, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    M = np.eye(4)
    M[:3, :3] = Rz @ Ry @ Rx
    M[:3, 3] = rng.uniform(-2, 2, 3)
    return M


@pytest.mark.parametrize("native_wh,render_wh", [((640, 480), (640, 480)), ((1296, 968), (341, 256)), ((640, 480), (1045, 784))])
def test_projection_and_view_conventions_equal_the_reference_pipeline(native_wh, render_wh):
    from stylemesh_amd.render.rasterizer import project_points, scaled_intrinsics
    rng = np.random.default_rng(5)
    K = np.eye(4)
    K[0, 0], K[1, 1] = 0.9 * native_wh[0], 0.92 * native_wh[0]
    K[0, 2], K[1, 2] = 0.52 * native_wh[0], 0.47 * native_wh[1]          # principal point off-centre
    for _ in range(5):
        M = random_pose(rng)
        cam = np.stack([rng.uniform(-1.5, 1.5, 200), rng.uniform(-1.2, 1.2, 200), rng.uniform(0.3, 8.0, 200)], 1)
        pts = cam @ M[:3, :3].T + M[:3, 3]                                  # world points in front of the camera
        xw, yw, row, lin = ref_pipeline(M, K, native_wh, render_wh, pts)
        mine = project_points(M, scaled_intrinsics(K, native_wh, render_wh), pts)
        # (scaled_intrinsics hands the kernel float32 intrinsics: a few 1e-5 px)
        np.testing.assert_allclose(mine[:, 0], xw, rtol=2e-6, atol=1e-4)
        np.testing.assert_allclose(mine[:, 1], yw, rtol=2e-6, atol=1e-4)    # no vertical flip:
        np.testing.assert_allclose(mine[:, 2], lin, rtol=1e-9)              # LinearizeDepth == camera-space z
        np.testing.assert_allclose(mine[:, 2], cam[:, 2], rtol=1e-9)
        inside = (yw > 0) & (yw < render_wh[1]) & (np.abs(yw - np.round(yw)) > 1e-3)
        assert np.array_equal(np.floor(mine[inside, 1]).astype(int), row[inside])   # file row (flip = 0) = floor(y)


def test_projection_matrix_known_answer():
    """util.h:11-35 for K = (fx 577.87, fy 577.87, cx 319.5, cy 239.5), 640 x 480, near 0.1, far 10 (ScanNet's depth
    intrinsics; renderer.h:19-20) - the sixteen numbers worked out by hand."""
    K = np.eye(3)
    K[0, 0] = K[1, 1] = 577.87
    K[0, 2], K[1, 2] = 319.5, 239.5
    P = ref_projection(K, 640, 480)
    want = np.array([[1.80584375, 0, 0.0015625, 0], [0, 2.407791666666667, 0.002083333333333, 0],
                     [0, 0, -1.02020202020202, -0.202020202020202], [0, 0, -1, 0]])
    np.testing.assert_allclose(P, want, rtol=1e-9, atol=1e-12)
    # a point on the optical axis one metre ahead lands on the principal point; window rows count from the top row
    from stylemesh_amd.render.rasterizer import project_points
    x, y, z = project_points(np.eye(4), (577.87, 577.87, 319.5, 239.5), np.array([[0.0, 0.0, 1.0], [0.0, 0.1, 1.0]])).T
    np.testing.assert_allclose([x[0], y[0], z[0]], [319.5, 239.5, 1.0])
    assert y[1] > y[0]      # +y of the ScanNet camera frame (down) = larger row index


def ref_pipeline_matterport(E, K, native_wh, render_wh, pts, flip, n=0.1, f=10.0):
    """scripts/matterport/render_uv/src/renderer/mp_renderer.cpp:85-130 restated: ``extr = transpose(make_mat4(
    image->extrinsics))`` turns the file's row-major floats into the matrix E as written; the intrinsics are scaled to the
    buffer size (:101-107) and handed to the same ``camera_utils::perspective`` WITH THE BUFFER SIZE (:109); the view
    matrix is E itself (:126, model matrix = identity); ``saveUV(filename, flip)`` as in the ScanNet renderer.
    World points -> (window x, window y from the bottom, file row, linearised depth)."""
    sx, sy = render_wh[0] / native_wh[0], render_wh[1] / native_wh[1]
    Kb = np.array(K, dtype=np.float64)
    Kb[0, 0] *= sx; Kb[1, 1] *= sy; Kb[0, 2] *= sx; Kb[1, 2] *= sy
    P = ref_projection(Kb, render_wh[0], render_wh[1], n, f)
    clip = (P @ np.asarray(E, dtype=np.float64) @ np.concatenate([pts, np.ones((len(pts), 1))], 1).T).T
    ndc = clip[:, :3] / clip[:, 3:4]
    xw = (ndc[:, 0] + 1) / 2 * render_wh[0]
    yw = (ndc[:, 1] + 1) / 2 * render_wh[1]
    z = ndc[:, 2]
    lin = (2 * n * f) / (f + n - z * (f - n))
    gl_row = np.floor(yw).astype(int)
    return xw, yw, (render_wh[1] - 1 - gl_row) if flip else gl_row, lin


@pytest.mark.parametrize("native_wh,render_wh", [((1280, 1024), (1280, 1024)), ((1280, 1024), (320, 256)), ((1280, 1024), (980, 784))])
def test_matterport_view_convention_equals_the_reference_pipeline(native_wh, render_wh):
    """The Matterport renderer uses the ``.house`` extrinsics as the view matrix directly (no look-at rebuild): the
    product's ``matterport_cam2world`` + ``project_points`` reproduce window position, file row (flip = 0 and the
    scripts' flip = 1) and linearised depth of that pipeline for random rigid E, off-centre principal points and
    render sizes != the image size."""
    from stylemesh_amd.render.rasterizer import matterport_cam2world, project_points, scaled_intrinsics
    rng = np.random.default_rng(11)
    K = np.eye(4)
    K[0, 0], K[1, 1] = 1.05 * native_wh[0], 1.07 * native_wh[0]
    K[0, 2], K[1, 2] = 0.49 * native_wh[0], 0.53 * native_wh[1]
    for _ in range(5):
        E = random_pose(rng)                                             # a rigid world -> eye transform
        eye = np.stack([rng.uniform(-1.5, 1.5, 200), rng.uniform(-1.2, 1.2, 200), -rng.uniform(0.3, 8.0, 200)], 1)
        Einv = np.linalg.inv(E)
        pts = eye @ Einv[:3, :3].T + Einv[:3, 3]                         # world points in front of the GL camera (-z)
        xw, yw, row0, lin = ref_pipeline_matterport(E, K, native_wh, render_wh, pts, flip=0)
        _, _, row1, _ = ref_pipeline_matterport(E, K, native_wh, render_wh, pts, flip=1)
        mine = project_points(matterport_cam2world(E), scaled_intrinsics(K, native_wh, render_wh), pts)
        np.testing.assert_allclose(mine[:, 0], xw, rtol=2e-6, atol=1e-4)
        np.testing.assert_allclose(mine[:, 1], yw, rtol=2e-6, atol=1e-4)
        np.testing.assert_allclose(mine[:, 2], lin, rtol=1e-9)
        np.testing.assert_allclose(mine[:, 2], -eye[:, 2], rtol=1e-9)   # depth = distance along the viewing direction
        inside = (yw > 0) & (yw < render_wh[1]) & (np.abs(yw - np.round(yw)) > 1e-3)
        assert np.array_equal(np.floor(mine[inside, 1]).astype(int), row0[inside])
        assert np.array_equal(render_wh[1] - 1 - np.floor(mine[inside, 1]).astype(int), row1[inside])


def test_matterport_view_known_answer():
    """Hand-worked: E = identity, K = (fx 1000, fy 1000, cx 640, cy 512) at 1280 x 1024. The eye-space point
    (0.1, 0.2, -2) (up-right of the axis, 2 m ahead) lands at window x = 640 + 1000 * 0.1 / 2 = 690, window y (from the
    BOTTOM) = 512 + 1000 * 0.2 / 2 = 612: file row 612 with flip = 0, row 1023 - 612 = 411 with flip = 1; depth 2."""
    from stylemesh_amd.render.rasterizer import matterport_cam2world, project_points
    K = np.eye(3)
    K[0, 0] = K[1, 1] = 1000.0
    K[0, 2], K[1, 2] = 640.0, 512.0
    pts = np.array([[0.1, 0.2, -2.0]])
    xw, yw, row0, lin = ref_pipeline_matterport(np.eye(4), K, (1280, 1024), (1280, 1024), pts, flip=0)
    np.testing.assert_allclose([xw[0], yw[0], lin[0]], [690.0, 612.0, 2.0], rtol=1e-12)
    assert row0[0] == 612
    x, y, z = project_points(matterport_cam2world(np.eye(4)), (1000.0, 1000.0, 640.0, 512.0), pts)[0]
    np.testing.assert_allclose([x, y, z], [690.0, 612.0, 2.0], rtol=1e-12)


def test_load_obj_applies_flip_uvs(tmp_path):
    """include/model.h:57 ``aiProcess_FlipUVs``: vt (0.25, 0.1) of the file is rendered as (0.25, 0.9)."""
    from stylemesh_amd.render.rasterizer import load_obj
    obj = tmp_path / "tri.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0.25 0.1\nvt 0.75 0.1\nvt 0.25 0.6\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1\n")
    m = load_obj(str(obj), device="cpu")
    np.testing.assert_allclose(m.uvs.numpy(), [[0.25, 0.9], [0.75, 0.9], [0.25, 0.4]], atol=1e-7)
    raw = load_obj(str(obj), device="cpu", flip_uvs=False)
    np.testing.assert_allclose(raw.uvs.numpy(), [[0.25, 0.1], [0.75, 0.1], [0.25, 0.6]], atol=1e-7)


# ------------------------------------------------------------------ f1: the loader's resizes
def cv2_linear_1d(src, n_out):
    """cv2.resize INTER_LINEAR for float data, one axis: fx = (dx + 0.5) * scale - 0.5, sx = floor(fx), the fraction
    is dropped at both borders (imgproc/resize.cpp)."""
    n_in = len(src)
    scale = n_in / n_out
    out = np.empty(n_out, np.float32)
    for dx in range(n_out):
        fx = np.float32((dx + 0.5) * scale - 0.5)
        sx = int(np.floor(fx))
        fx = np.float32(fx - sx)
        if sx < 0:
            sx, fx = 0, np.float32(0)
        if sx >= n_in - 1:
            sx, fx = n_in - 1, np.float32(0)
        out[dx] = src[sx] * (np.float32(1) - fx) + src[min(sx + 1, n_in - 1)] * fx
    return out


def test_depth_resize_is_cv2_inter_linear():
    """abstract_dataset.py:301 / scannet_dataset.py:323. Known answers: 7 -> 3 samples of arange(7) are
    0.6667, 3.0, 5.3333 (SURVEY.md section 8 a9 probe); 4 -> 6 of (0, 10, 20, 30) clamps at both ends."""
    np.testing.assert_allclose(vc.resize_bilinear_np(np.arange(7, dtype=np.float32)[None], (1, 3))[0],
                               [2 / 3, 3.0, 16 / 3], rtol=1e-6)
    np.testing.assert_allclose(vc.resize_bilinear_np(np.array([[0, 10, 20, 30]], np.float32), (1, 6))[0],
                               [0.0, 5.0, 35 / 3, 55 / 3, 25.0, 30.0], rtol=1e-6)
    rng = np.random.default_rng(0)
    for (h, w, oh, ow) in [(480, 640, 256, 341), (9, 7, 4, 5), (5, 6, 13, 17), (968, 1296, 256, 343)]:
        a = rng.uniform(0, 5, (h, w)).astype(np.float32)
        want = np.stack([cv2_linear_1d(r, ow) for r in a])                   # separable: rows, then columns
        want = np.stack([cv2_linear_1d(c, oh) for c in want.T]).T
        np.testing.assert_allclose(vc.resize_bilinear_np(a, (oh, ow)), want, rtol=2e-6, atol=2e-6)


def test_angle_resize_is_cv2_inter_nearest():
    """abstract_dataset.py:308: source index = min(floor(dst * in / out), in - 1). 7 -> 3 picks 0, 2, 4; 7 -> 16 picks
    0,0,0,1,1,2,2,3,3,3,4,4,5,5,6,6 (SURVEY.md section 8 a9 probe)."""
    a = np.arange(7, dtype=np.float32)[None]
    assert vc.resize_nearest_np(a, (1, 3))[0].tolist() == [0, 2, 4]
    assert vc.resize_nearest_np(a, (1, 16))[0].tolist() == [0, 0, 0, 1, 1, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 6]
    b = np.arange(480 * 640, dtype=np.float32).reshape(480, 640)
    r = vc.resize_nearest_np(b, (256, 341))
    ys = np.minimum(np.floor(np.arange(256) * (480 / 256)).astype(int), 479)
    xs = np.minimum(np.floor(np.arange(341) * (640 / 341)).astype(int), 639)
    assert np.array_equal(r, b[ys][:, xs])


def test_mask_resize_is_pillow_nearest_at_pixel_centres():
    """abstract_dataset.py:311: ``mask.resize(size, Image.NEAREST)`` on the mode-"1" image of calculate_mask. Pillow
    samples at pixel centres: 7 -> 3 picks source indices 1, 3, 5 (not cv2's 0, 2, 4)."""
    row = np.zeros((1, 7), bool)
    for src, dst in ((1, 0), (3, 1), (5, 2)):
        m = row.copy()
        m[0, src] = True
        assert vc.resize_mask_pil(m, (1, 3))[0].tolist() == [i == dst for i in range(3)]
    m = row.copy()
    m[0, 0] = m[0, 2] = m[0, 4] = True                  # cv2's picks: none of them is sampled by Pillow
    assert not vc.resize_mask_pil(m, (1, 3)).any()
    # assemble_batch (the synthetic views) and the ScanNet loader both go through it
    rng = np.random.default_rng(1)
    big = rng.random((64, 88)) > 0.5
    uv = np.zeros((64, 88, 3), np.float32)
    uv[..., 0] = big
    depth = np.ones((40, 56), np.float32)
    batch = vc.assemble_batch(torch.zeros(3, 40, 56), depth, [uv[:40, :56], uv], np.ones((40, 56), np.float32),
                              levels=[40, 64], min_pyramid_depth=0.9)
    assert np.array_equal(batch[10][0].numpy(), vc.resize_mask_pil(big, (40, 56)))
