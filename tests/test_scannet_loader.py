"""The ScanNet directory loader (f1) on a synthetic on-disk scene written in the reference's formats."""
import numpy as np
import pytest
import torch

from stylemesh_amd.data import synthetic as S
from stylemesh_amd.data import view_contract as vc
from stylemesh_amd.data.scannet import ScanNetSceneDataset, ScanNetSingleSceneDataModule, write_scene


def render_views(n):
    room = S.BoxRoom((12.0, 9.0, 3.0))
    views = []
    for s in range(n):
        rng = np.random.default_rng(s)
        pos = np.array([rng.uniform(1, 11), rng.uniform(1, 8), 1.5])
        yaw, pitch = rng.uniform(0, 6.28), rng.uniform(-0.3, 0.2)
        full = room.render(pos, yaw, pitch, (120, 160))
        levels = [room.render(pos, yaw, pitch, hw)[0] for hw in [(64, 85), (108, 144)]]
        _, _, depth = room.render(pos, yaw, pitch, (60, 80))
        views.append(dict(rgb01=S.smooth_noise(rng, 3, 60, 80), depth=depth, uv_full=full[0], angle_full=full[1],
                          uv_levels=levels, pose=np.eye(4) + 0.01 * s))
    return views


def test_loader_reads_the_reference_layout(tmp_path):
    views = render_views(5)
    write_scene(str(tmp_path), "scene0000_00", views, [64, 108])
    (tmp_path / "train/images/scene0000_00/uv_64").mkdir()          # duplicate of uv_64.0: must be ignored
    ds = ScanNetSceneDataset(str(tmp_path / "train/images"), "scene0000_00", resize_size=64, pyramid_levels=4,
                             min_pyramid_depth=0.25, min_pyramid_height=32)
    assert len(ds) == 5 and list(ds.levels) == [64.0, 108.0] and not ds.rendered_depth
    b = ds[2]
    assert len(b) == 13
    rgb, extr, K, depth, dl, rl, ol, w, idx, uvs, mask, ag, adeg = b
    assert rgb.shape == (1, 3, 64, 85) and depth.shape == (1, 1, 64, 85) and int(idx) == 2
    assert [tuple(u.shape) for u in uvs] == [(1, 64, 85, 2), (1, 108, 144, 2)] and mask.shape == (1, 64, 85)
    v = views[2]
    # depth: png millimetres -> metres, bilinear 60x80 -> 64x85
    ref_depth = vc.resize_bilinear_np(np.round(v["depth"] * 1000) / 1000.0, (64, 85))
    np.testing.assert_allclose(depth[0, 0].numpy(), ref_depth, atol=1e-6)
    np.testing.assert_array_equal(uvs[1][0].numpy(), vc.uv_to_grid(v["uv_levels"][1]).numpy())
    ref = vc.calculate_depth_level(ref_depth, [64.0, 108.0], 0.25)
    np.testing.assert_array_equal(rl[0, 0].numpy(), ref[1])
    np.testing.assert_allclose(w[0, 0].numpy(), ref[3], atol=1e-6)
    np.testing.assert_allclose(ag[0, 0].numpy(), vc.resize_nearest_np(v["angle_full"], (64, 85)), atol=1e-7)
    np.testing.assert_allclose(extr[0].numpy(), v["pose"], atol=1e-5)
    np.testing.assert_allclose(K[0, 0, 0].item(), 0.9 * 80 * 85 / 80, rtol=1e-6)
    assert 0.05 < float(mask.float().mean()) <= 1.0
    np.testing.assert_allclose(float(adeg.max()), float(torch.rad2deg(torch.acos(ag)).max()))


def test_datamodule_split_and_sharding(tmp_path):
    write_scene(str(tmp_path), "s", render_views(5), [64, 108])
    dm = ScanNetSingleSceneDataModule(str(tmp_path), "s", resize_size=64, split=(0.8, 0.2), index_repeat=3,
                                      rank=1, world_size=2)
    dm.setup()
    assert dm.train_indices == [0, 1, 2, 3] and dm.val_indices == [4]
    assert [int(b[8]) for b in dm.train_dataloader()] == [1, 1, 1, 3, 3, 3]
    assert [int(b[8]) for b in dm.val_dataloader()] == [4]
    with pytest.raises(ValueError):
        ScanNetSceneDataset(str(tmp_path / "train/images"), "missing")


def test_matterport_region_layout(tmp_path):
    """Same views written in the Matterport region layout (names <pano>_i<cam>_<yaw>, uv_<W>_<H>, depth / 4000)."""
    import os
    from PIL import Image
    from stylemesh_amd.data.matterport import MatterportRegionDataset, MatterportSingleRegionDataModule
    views = render_views(4)
    rp = tmp_path / "v1" / "scans" / "house1" / "rendered" / "region_2"
    names = ["aaa_i0_1", "aaa_i1_0", "aaa_i0_0", "bbb_i0_3"]            # sorted order: aaa_i0_0, aaa_i0_1, aaa_i1_0, bbb_i0_3
    order = [2, 0, 1, 3]
    for d in ["color", "depth", "pose", "angle", "uv_85_64", "uv_144_108"]:
        os.makedirs(rp / d)
    for n, v in zip(names, views):
        Image.fromarray((v["rgb01"].transpose(1, 2, 0) * 255).astype(np.uint8)).save(rp / "color" / f"{n}.png")
        Image.fromarray(np.round(v["depth"] * 4000).astype(np.uint16)).save(rp / "depth" / f"{n.replace('_i', '_d')}.png")
        np.savetxt(rp / "pose" / f"{n}.txt", v["pose"], fmt="%.6f", delimiter=" ")
        np.save(rp / "angle" / f"{n}.angle.npy", np.repeat(v["angle_full"][:, :, None], 3, 2).astype(np.float32))
        for folder, u in zip(["uv_85_64", "uv_144_108"], v["uv_levels"]):
            np.save(rp / folder / f"{n}.uvs.npy", u.astype(np.float32))
    with open(rp / "pose" / "aaa_i0_0.intrinsics.txt", "w") as f:
        f.write("70 0 40\n0 70 30\n0 0 1\n80 60\n")
    ds = MatterportRegionDataset(str(tmp_path / "v1" / "scans"), "house1", region_index=2, resize_size=64, pyramid_levels=4,
                                 min_pyramid_depth=0.2, min_pyramid_height=32)
    assert len(ds) == 4 and list(ds.levels) == [64.0, 108.0]
    for i, src in enumerate(order):
        b = ds[i]
        np.testing.assert_allclose(b[1][0].numpy(), views[src]["pose"], atol=1e-5)
    b = ds[0]
    v = views[order[0]]
    ref_depth = vc.resize_bilinear_np(np.round(v["depth"] * 4000) / 4000.0, (64, 85))
    np.testing.assert_allclose(b[3][0, 0].numpy(), ref_depth, atol=1e-6)
    expected_mask = vc.resize_nearest_np(vc.calculate_mask(v["uv_levels"][1]), (64, 85))   # no depth test
    np.testing.assert_array_equal(b[10][0].numpy(), expected_mask)
    np.testing.assert_allclose(b[2][0, 0, 0].item(), 70 * 85 / 80, rtol=1e-6)
    dm = MatterportSingleRegionDataModule(str(tmp_path), "house1", region_index=2, resize_size=64, split=(0.75, 0.25))
    dm.setup()
    assert dm.train_indices == [0, 1, 2] and dm.val_indices == [3]


def test_datamodule_prefetch_process_yields_the_same_schedule(tmp_path):
    """The decode PROCESS behind the directory loader (default ``prefetch=2``): same views, same order, same tensors as
    the in-loop decode; one worker serves consecutive epochs; an abandoned epoch does not poison the next."""
    write_scene(str(tmp_path), "s", render_views(5), [64, 108])
    kw = dict(resize_size=64, split=(0.8, 0.2), index_repeat=2)
    plain = ScanNetSingleSceneDataModule(str(tmp_path), "s", prefetch=0, **kw)
    ahead = ScanNetSingleSceneDataModule(str(tmp_path), "s", prefetch=2, decode_workers=3, **kw)   # 4 views over 3 workers
    plain.setup(), ahead.setup()
    want = list(plain.train_dataloader())
    for epoch in range(2):
        got = list(ahead.train_dataloader())
        assert [int(b[8]) for b in got] == [int(b[8]) for b in want] == [0, 0, 1, 1, 2, 2, 3, 3]
        for a, b in zip(got, want):
            assert torch.equal(a[0], b[0]) and torch.equal(a[9][1], b[9][1]) and torch.equal(a[10], b[10])
            assert a.new_view == b.new_view
    pid = ahead._worker.proc.pid
    it = iter(ahead.train_dataloader())
    next(it)
    it.close()                                   # epoch abandoned after one step
    got = list(ahead.train_dataloader())         # a fresh worker takes over
    assert [int(b[8]) for b in got] == [0, 0, 1, 1, 2, 2, 3, 3] and ahead._worker.proc.pid != pid
    ahead._worker.close()
