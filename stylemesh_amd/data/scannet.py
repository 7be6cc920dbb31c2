"""ScanNet single-scene directory loader: the file-system half of the reference's
``ScanNetDataset`` / ``ScanNet_Single_Scene_DataModule`` (data/scannet_dataset.py:99-306,
data/scannet_single_scene_dataset.py, data/abstract_dataset.py:100-167,270-345,434-495).

Expected layout (written by the reference's export + render scripts)::

    <root_path>/train/images/<scene>/
        color/<n>.jpg|png            captured RGB
        depth/<n>.png                16-bit sensor depth in millimetres        (-> metres: / 1000)
        pose/<n>.txt                 4x4 camera-to-world, space separated
        <anything>.txt               intrinsics: fx_color / fy_color / mx_color / my_color / colorWidth / colorHeight
        uv/<n>.npy                   (H,W,3) float32 UV map at the full 480x640 render size
        uv/<n>.angle.npy             (H,W,3) float32 cos(theta)
        uv/<n>.depth.npy             (H,W,3) float32 rendered depth (used when depth/ is empty)
        uv_<height>/<n>.npy          the UV pyramid, one folder per height (e.g. uv_256.0 ... uv_960.0)

``__getitem__`` returns the same 13-tuple as the reference (``view_contract.assemble_batch`` order), batch
dimension included (B = 1), ready for ``training_step``. ``write_scene`` writes a synthetic scene in this format
(tests, demos).
"""
from __future__ import annotations

import os
from os.path import join

import numpy as np
import torch
from PIL import Image

from . import view_contract as vc


def _is_float(s):
    try:
        float(s)
        return True
    except ValueError:
        return False


def _numbered(folder, pred):
    if not os.path.isdir(folder):
        return []
    files = sorted(os.listdir(folder), key=lambda x: int(x.split(".")[0]))
    return [join(folder, f) for f in files if pred(f)]


class ScanNetSceneDataset:
    depth_scale = 1000.0        # sensor depth png units per metre (scannet_dataset.py:300)
    mask_uses_depth = True      # calculate_mask ANDs "resized depth > 0" (scannet_dataset.py:322-324)

    def __init__(self, root_path, scene, resize_size=256, pyramid_levels=5, min_pyramid_depth=0.25,
                 min_pyramid_height=32, max_images=-1):
        self.scene_path = join(root_path, scene)
        if not os.path.isdir(self.scene_path):
            raise ValueError(f"scene folder not found: {self.scene_path}")
        self.scene, self.resize_size, self.min_pyramid_depth = scene, resize_size, min_pyramid_depth
        self._scan(pyramid_levels, min_pyramid_height)
        n = len(self.rgb_images)
        sp = self.scene_path
        if not (n > 0 and n == len(self.depth_images) == len(self.extrinsics) == len(self.angle_maps)
                and all(len(u) == n for u in self.uv_maps)):
            raise ValueError(f"Scene {sp} rendered incomplete: colors {n}, depth {len(self.depth_images)}, "
                             f"uvs {[len(u) for u in self.uv_maps]}, angles {len(self.angle_maps)}, "
                             f"extr {len(self.extrinsics)}")
        if max_images > 0:
            n = min(n, max_images)
        self.size = n
        self.intrinsics, self.intrinsic_image_size = self._read_intrinsics()

    def _scan(self, pyramid_levels, min_pyramid_height):
        """Collect the sorted file lists of the scene (scannet_dataset.py:102-251)."""
        sp = self.scene_path
        self.rgb_images = _numbered(join(sp, "color"), lambda f: f.endswith("jpg") or f.endswith("png"))
        sensor = _numbered(join(sp, "depth"), lambda f: True)
        self.rendered_depth = len(sensor) == 0     # fall back to the OpenGL depth (scannet_dataset.py:139-145)
        self.depth_images = sensor if sensor else _numbered(join(sp, "uv"), lambda f: "npy" in f and "depth" in f)
        self.extrinsics = _numbered(join(sp, "pose"), lambda f: True)
        self.angle_maps = _numbered(join(sp, "uv"), lambda f: "npy" in f and "angle" in f)
        # UV pyramid folders uv_<height>, sorted by height, duplicates (256 vs 256.0) dropped, >= min height,
        # first `pyramid_levels` kept (scannet_dataset.py:196-236)
        folders = sorted([f for f in os.listdir(sp) if f.startswith("uv_") and _is_float(f.split("_")[1])],
                         key=lambda x: float(x.split("_")[1]))
        folders = [f for i, f in enumerate(folders) if i == 0 or float(f.split("_")[1]) != float(folders[i - 1].split("_")[1])]
        folders = [f for f in folders if float(f.split("_")[1]) >= min_pyramid_height][:pyramid_levels]
        if not folders:
            raise ValueError(f"no uv_<height> pyramid folders in {sp}")
        self.levels = np.array([float(f.split("_")[1]) for f in folders])
        is_uv = lambda f: "npy" in f and "angle" not in f and "depth" not in f
        self.uv_maps = [_numbered(join(sp, f), is_uv) for f in folders]

    def _read_intrinsics(self):
        K, w, h = np.identity(4, dtype=np.float32), 0, 0
        files = [join(self.scene_path, f) for f in os.listdir(self.scene_path) if ".txt" in f]
        if len(files) == 1:
            self.intrinsics_file = files[0]
            for l in open(files[0]).read().splitlines():
                l = l.strip()
                if " = " not in l:
                    continue
                k, v = l.split(" = ")
                if "fx_color" in k: K[0, 0] = float(v)
                if "fy_color" in k: K[1, 1] = float(v)
                if "mx_color" in k: K[0, 2] = float(v)
                if "my_color" in k: K[1, 2] = float(v)
                if "colorWidth" in k: w = int(v)
                if "colorHeight" in k: h = int(v)
        return K, (w, h)

    def __len__(self):
        return self.size

    def __getitem__(self, i):
        rgb = Image.open(self.rgb_images[i]).convert("RGB")
        if self.rendered_depth:
            depth = np.load(self.depth_images[i])[:, :, 0].astype(np.float32)
        else:
            depth = (np.asarray(Image.open(self.depth_images[i])) / self.depth_scale).astype(np.float32)
        uvs = [np.load(u[i]) for u in self.uv_maps]
        angle = np.load(self.angle_maps[i])[:, :, 0]
        mask_big = vc.calculate_mask(uvs[-1], depth if self.mask_uses_depth else None)   # largest UV map (:288)
        w, h = rgb.size
        h_new = self.resize_size
        w_new = round(w * h_new / h)
        rgb = rgb.resize((w_new, h_new))                                 # PIL default filter, as the reference (:299)
        depth_r = vc.resize_bilinear_np(depth, (h_new, w_new))           # cv2.INTER_LINEAR (:301)
        angle_r = vc.resize_nearest_np(angle, (h_new, w_new))            # cv2.INTER_NEAREST (:308)
        mask_r = vc.resize_mask_pil(mask_big, (h_new, w_new))            # PIL NEAREST on the mode-"1" image (:311)
        K = np.array(self.intrinsics)
        iw, ih = self.intrinsic_image_size
        if (iw, ih) != (w_new, h_new) and iw > 0 and ih > 0:            # modify_intrinsics_matrix (:257-265)
            K[0, 0] *= w_new / iw; K[0, 2] *= w_new / iw
            K[1, 1] *= h_new / ih; K[1, 2] *= h_new / ih
        extr = np.array([[float(x) for x in line.split(" ") if x.strip()] for line in open(self.extrinsics[i]) if line.strip()],
                        dtype=np.float32)
        cont, rounded, other, wgt = vc.calculate_depth_level(depth_r, self.levels, self.min_pyramid_depth)
        rgb_t = torch.from_numpy(np.array(rgb)).permute(2, 0, 1).float() / 255
        angle_t = torch.from_numpy(np.ascontiguousarray(angle_r, dtype=np.float32))[None, None]
        return (vc.pre(rgb_t)[None], torch.from_numpy(extr)[None], torch.from_numpy(K)[None],
                torch.from_numpy(depth_r)[None, None], torch.from_numpy(cont)[None, None],
                torch.from_numpy(rounded)[None, None], torch.from_numpy(other)[None, None],
                torch.from_numpy(wgt)[None, None], torch.tensor([i]), [vc.uv_to_grid(u)[None] for u in uvs],
                torch.from_numpy(np.ascontiguousarray(mask_r))[None], angle_t, torch.rad2deg(torch.acos(angle_t)))


class ScanNetSingleSceneDataModule:
    """Sequential split + RepeatingSampler over one scene (data/abstract_dataset.py:434-495), batch size 1;
    ``rank`` / ``world_size`` shard the train views (SURVEY.md section 8 e)."""
    split_modes = ["sequential"]
    sampler_modes = ["repeat", "sequential"]

    def __init__(self, root_path, scene, resize_size=256, pyramid_levels=5, min_pyramid_depth=0.25,
                 min_pyramid_height=32, max_images=-1, split=(0.8, 0.2), index_repeat=1, sampler_mode="repeat",
                 rank=0, world_size=1, prefetch=2, decode_workers=1):
        self.prefetch = prefetch   # views decoded ahead of the training loop (0 = decode inside the loop)
        self.decode_workers = decode_workers   # decode processes (the reference's DataLoader ``num_workers``)
        self.args = dict(root_path=join(root_path, "train/images"), scene=scene, resize_size=resize_size,
                         pyramid_levels=pyramid_levels, min_pyramid_depth=min_pyramid_depth,
                         min_pyramid_height=min_pyramid_height, max_images=max_images)
        self.split, self.index_repeat, self.sampler_mode = split, index_repeat, sampler_mode
        self.rank, self.world_size = rank, world_size

    def prepare_data(self):
        pass

    def setup(self, stage=None):
        self.train_dataset = self.val_dataset = ScanNetSceneDataset(**self.args)
        n = len(self.train_dataset)
        n_train = int(self.split[0] * n)
        self.train_indices, self.val_indices = list(range(n_train)), list(range(n_train, n))

    def _decode_worker(self):
        """The persistent decode process(es) of all epochs (re-started if an epoch was abandoned)."""
        from ..runtime.distributed import DecodeProcess
        if self.prefetch <= 0:
            return None
        if getattr(self, "_worker", None) is None or not self._worker.alive() or self._worker.busy:
            if getattr(self, "_worker", None) is not None:
                self._worker.close()
            self._worker = DecodeProcess(self.train_dataset.__getitem__, depth=self.prefetch,
                                         n_workers=self.decode_workers)
        return self._worker

    def warm_start(self):
        """Start the decode processes NOW (a spawned interpreter needs ~1 s to import its modules): called by the CLI
        before it builds the model, so that the first view is decoded while the training process still sets itself up
        instead of the first step waiting for it."""
        self._decode_worker()

    def train_dataloader(self):
        from ..runtime.distributed import scheduled_batches   # equal step counts + lock-step view changes
        worker = self._decode_worker()
        return scheduled_batches(self.train_dataset.__getitem__, self.train_indices, self.rank, self.world_size,
                                 self.index_repeat, repeat=self.sampler_mode == "repeat", prefetch=self.prefetch,
                                 worker=worker)

    def val_dataloader(self):
        return (self.val_dataset[i] for i in self.val_indices) if self.val_indices else None


def write_scene(root_path, scene, views, level_heights, full_hw=(480, 640), color_ext="png"):
    """Write synthetic views in the on-disk format above. ``views``: list of dicts with ``rgb01`` (3,h,w) float,
    ``depth`` (h,w) metres, ``uv_full`` (H,W,3), ``angle_full`` (H,W), ``uv_levels`` [(H_i,W_i,3)], ``pose`` 4x4."""
    sp = join(root_path, "train/images", scene)
    for d in ["color", "depth", "pose", "uv"] + [f"uv_{float(h)}" for h in level_heights]:
        os.makedirs(join(sp, d), exist_ok=True)
    for n, v in enumerate(views):
        Image.fromarray((np.clip(v["rgb01"], 0, 1).transpose(1, 2, 0) * 255 + 0.5).astype(np.uint8)).save(
            join(sp, "color", f"{n}.{color_ext}"))
        Image.fromarray(np.round(v["depth"] * 1000).astype(np.uint16)).save(join(sp, "depth", f"{n}.png"))
        np.savetxt(join(sp, "pose", f"{n}.txt"), v["pose"], fmt="%.6f", delimiter=" ")
        np.save(join(sp, "uv", f"{n}.npy"), v["uv_full"].astype(np.float32))
        np.save(join(sp, "uv", f"{n}.angle.npy"), np.repeat(v["angle_full"][:, :, None], 3, 2).astype(np.float32))
        for h, u in zip(level_heights, v["uv_levels"]):
            np.save(join(sp, f"uv_{float(h)}", f"{n}.npy"), u.astype(np.float32))
    h, w = views[0]["depth"].shape
    with open(join(sp, "_info.txt"), "w") as f:
        f.write(f"colorHeight = {h}\ncolorWidth = {w}\nfx_color = {0.9 * w}\nfy_color = {0.9 * w}\n"
                f"mx_color = {w / 2}\nmy_color = {h / 2}\n")
    return sp
