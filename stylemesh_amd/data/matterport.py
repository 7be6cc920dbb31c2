"""Matterport3D single-region loader: the file-system half of the reference's
``MatterportDataset`` (data/matterport_dataset.py:55-311). Differences from ScanNet: everything lives under
``<root>/<scene>/rendered/region_<i>/``; file names are ``<panorama>_<d|i><cam>_<yaw>.<ext>`` sorted by
(panorama, cam * 100 + yaw); sensor depth is in 0.25 mm units (/ 4000); the pyramid folders are ``uv_<W>_<H>``
with files containing ``uvs``; angle maps live in ``angle/``; intrinsics are a 3x3 matrix + "W H" line in
``pose/*.intrinsics.txt``; the mask has no depth test (:295-311)."""
from __future__ import annotations

import os
from os.path import join

import numpy as np

from .scannet import ScanNetSceneDataset, ScanNetSingleSceneDataModule


def _sort_key(x):
    stem = x.split(".")[0].split("_")
    return [stem[0], int(stem[1][1]) * 100 + int(stem[2])]


def _listed(folder, pred):
    if not os.path.isdir(folder):
        return []
    return [join(folder, f) for f in sorted([f for f in os.listdir(folder) if pred(f)], key=_sort_key)]


class MatterportRegionDataset(ScanNetSceneDataset):
    depth_scale = 4000.0        # matterport_dataset.py:288
    mask_uses_depth = False     # matterport_dataset.py:295-311

    def __init__(self, root_path, scene, region_index=0, **kw):
        self.region_index = region_index
        super().__init__(root_path, scene, **kw)

    def _scan(self, pyramid_levels, min_pyramid_height):
        rp = join(self.scene_path, "rendered", f"region_{self.region_index}")
        if not os.path.isdir(rp):
            raise ValueError(f"region folder not found: {rp}")
        self.region_path = rp
        self.rgb_images = _listed(join(rp, "color"), lambda f: f.endswith("jpg") or f.endswith("png"))
        sensor = _listed(join(rp, "depth"), lambda f: True)
        self.rendered_depth = len(sensor) == 0
        self.depth_images = sensor if sensor else _listed(join(rp, "rendered_depth"), lambda f: "npy" in f and "depth" in f)
        self.extrinsics = _listed(join(rp, "pose"), lambda f: "intrinsic" not in f)
        self.angle_maps = _listed(join(rp, "angle"), lambda f: "npy" in f and "angle" in f)
        folders = sorted([f for f in os.listdir(rp) if "uv_" in f], key=lambda x: int(x.split("_")[-1]))
        folders = [f for f in folders if int(f.split("_")[-1]) >= min_pyramid_height][:pyramid_levels]
        if not folders:
            raise ValueError(f"no uv_<W>_<H> pyramid folders in {rp}")
        self.levels = np.array([float(f.split("_")[-1]) for f in folders])
        self.uv_maps = [_listed(join(rp, f), lambda f: "npy" in f and "uvs" in f) for f in folders]

    def _read_intrinsics(self):
        K, w, h = np.identity(4, dtype=np.float32), 0, 0
        files = [join(self.region_path, "pose", f) for f in os.listdir(join(self.region_path, "pose")) if ".intrinsics.txt" in f]
        if files:
            self.intrinsics_file = files[0]
            for i, l in enumerate(open(files[0]).read().splitlines()):
                e = l.strip().split(" ")
                if i < 3:
                    K[i][:3] = [float(e[0]), float(e[1]), float(e[2])]
                elif i == 3:
                    w, h = int(e[0]), int(e[1])
                else:
                    raise ValueError("index too large", i)
        return K, (w, h)


class MatterportSingleRegionDataModule(ScanNetSingleSceneDataModule):
    def __init__(self, root_path, scene, region_index=0, **kw):
        super().__init__(root_path, scene, **kw)
        self.args["root_path"] = join(root_path, "v1/scans")   # matterport_single_scene_dataset.py:40
        self.args["region_index"] = region_index

    def setup(self, stage=None):
        self.train_dataset = self.val_dataset = MatterportRegionDataset(**self.args)
        n = len(self.train_dataset)
        n_train = int(self.split[0] * n)
        self.train_indices, self.val_indices = list(range(n_train)), list(range(n_train, n))
