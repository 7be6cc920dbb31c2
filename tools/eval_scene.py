"""End-to-end demo of the rows of SURVEY.md section 8 working together on a synthetic scene (run on the GPU box):
  f3  the HIP rasteriser renders the UV / angle / depth maps of a box-room mesh along a camera trajectory,
  a*  the texture-optimisation hot path (LightningModule mirror + MiniTrainer) stylises the scene,
  f2  the texture is exported, the styled frames are rendered from it,
  f4  the reprojection-error metric is evaluated on the styled frames (and, for comparison, on the un-optimised
      texture rendered the same way).
Prints one JSON line."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from stylemesh_amd.data import synthetic as S
from stylemesh_amd.data import view_contract as vc
from stylemesh_amd import render as R
from stylemesh_amd import eval as E
from stylemesh_amd.model.model import TextureOptimizationStyleTransferPipeline
from stylemesh_amd.trainer import MiniTrainer, JsonlLogger


class TrajectoryDataModule:
    """Views rendered by the rasteriser along ``S.trajectory_poses`` (same 13-tuple contract as the real loaders)."""

    def __init__(self, n_views, view_hw, level_hw, min_pyramid_depth, index_repeat, room, device):
        self.index_repeat = index_repeat
        mesh = R.box_room_mesh(room, device=device, subdiv=8)
        self.frames, self.batches = [], []
        rng = np.random.default_rng(0)
        for i, (pos, yaw, pitch) in enumerate(S.trajectory_poses(n_views, room)):
            K, c2w = S.camera_matrices(pos, yaw, pitch, view_hw)
            gl = lambda hw: np.array([K[0, 0] * hw[1] / view_hw[1], K[1, 1] * hw[0] / view_hw[0],
                                      (K[0, 2] + 0.5) * hw[1] / view_hw[1], (K[1, 2] + 0.5) * hw[0] / view_hw[0]], np.float32)
            uvs = [R.render_maps(mesh, c2w, gl(hw), hw, znear=0.05, zfar=50.0)[0].cpu().numpy() for hw in level_hw]
            _, ang, dep = R.render_maps(mesh, c2w, gl(view_hw), view_hw, znear=0.05, zfar=50.0)
            rgb = torch.from_numpy(S.smooth_noise(rng, 3, *view_hw))
            self.batches.append(vc.assemble_batch(rgb, dep.cpu().numpy(), uvs, ang.cpu().numpy(), [h for h, _ in level_hw],
                                                  min_pyramid_depth, idx=i, extrinsics=torch.from_numpy(c2w).double()[None],
                                                  intrinsics=torch.from_numpy(K).double()[None]))
            self.frames.append(dict(depth=dep, pose=torch.from_numpy(c2w).to(device), K=torch.from_numpy(K).to(device)))
        self.train_indices, self.val_indices = list(range(n_views)), [n_views - 1]   # the reference saves the
        # texture only after a validation epoch (model/model.py:378-385)

    def prepare_data(self): pass
    def setup(self, stage=None): pass
    def train_dataloader(self):
        return (self.batches[i] for i in vc.RepeatingSampler(self.train_indices, self.index_repeat))
    def val_dataloader(self): return (self.batches[i] for i in self.val_indices)


def styled_frames(model, dm, device):
    out = []
    with torch.no_grad():
        for b, f in zip(dm.batches, dm.frames):
            uv0 = b[9][0].to(device)                          # the base-resolution UV grid
            img = model.texture(uv0)[0]                       # [3,H,W] in the pre() colour space
            out.append(dict(styled=img.float(), depth=f["depth"], pose=f["pose"]))
    return out


def main(n_views=12, epochs=3, index_repeat=10, tex=512):
    device = torch.device("cuda")
    view_hw, level_hw = (96, 128), [(96, 128), (160, 214)]
    room = S.BoxRoom((6.0, 4.5, 2.8))
    t0 = time.time()
    dm = TrajectoryDataModule(n_views, view_hw, level_hw, 0.75, index_repeat, room, device)
    t_render = time.time() - t0
    tmp = tempfile.mkdtemp()
    vgg_path = os.path.join(tmp, "vgg.pth")
    torch.save(S.seeded_vgg_state(0), vgg_path)
    model = TextureOptimizationStyleTransferPipeline(
        W=tex, H=tex, hierarchical_texture=True, hierarchical_layers=4, style_image=S.style_image(1, 300, 260),
        vgg_gatys_model_path=vgg_path, style_weights=[1000., 1000., 10., 10., 1000.], angle_threshold=60,
        style_pyramid_mode="multi", learning_rate=1.0, tex_reg_weights=[8, 4, 2, 0],
        loss_weights={"content": 7e1, "style": 1e-4, "tex_reg": 5e3}, save_texture=True, texture_dir=tmp)
    before = None
    trainer = MiniTrainer(max_epochs=epochs, logger=JsonlLogger(save_dir=tmp, version=0), device=device)
    model.to(device)
    K = dm.frames[0]["K"]
    before = E.evaluate_sequence(styled_frames(model, dm, device), K, pair_threshold=2, pair_threshold_short=1,
                                 pair_threshold_long=3)
    t0 = time.time()
    trainer.fit(model, dm)
    torch.cuda.synchronize()
    t_train = time.time() - t0
    frames = styled_frames(model, dm, device)
    after = E.evaluate_sequence(frames, K, pair_threshold=2, pair_threshold_short=1, pair_threshold_long=3)
    saved = sorted(f for f in os.listdir(tmp) if f.endswith(".jpg"))
    var = float(torch.stack([f["styled"] for f in frames]).var())
    print(json.dumps({"views": n_views, "steps": epochs * index_repeat * n_views, "render_s": round(t_render, 3),
                      "train_s": round(t_train, 3), "texture_files": saved[:3], "styled_pixel_variance": round(var, 2),
                      "reprojection_mse_zero_texture": before, "reprojection_mse_styled": after}))
    return before, after, var, saved


if __name__ == "__main__":
    main()
