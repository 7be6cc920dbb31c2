"""The reference-shaped Python surface (stylemesh_amd.model.*) on the GPU: the autograd formulation
(forward_with_loss + backward, the way the reference computes) and the fused Trainer path against the
reference-generated goldens."""
import os
import tempfile

import numpy as np
import pytest
import torch

from conftest import batch_from_golden, load_golden
from golden_cases import FLAGSETS, LOSS_WEIGHTS, STYLE_HW, STYLE_SEED, STYLE_WEIGHTS, TEX, VGG_SEED
from gpu_util import assert_close, require_gpu
from stylemesh_amd.data import synthetic as S
from test_engine_gpu import grad_close, texture_close

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def vgg_file():
    f = tempfile.NamedTemporaryFile(suffix=".pth", delete=False)
    torch.save(S.seeded_vgg_state(VGG_SEED), f.name)
    return f.name


def make_model(cfgd, init=None, **kw):
    require_gpu()
    from stylemesh_amd.model.model import TextureOptimizationStyleTransferPipeline
    m = TextureOptimizationStyleTransferPipeline(
        W=TEX, H=TEX, hierarchical_texture=cfgd["hier"], hierarchical_layers=4, style_image=S.style_image(STYLE_SEED, *STYLE_HW),
        style_weights=STYLE_WEIGHTS, vgg_gatys_model_path=vgg_file(), use_angle_weight=cfgd["angle"],
        use_depth_scaling=cfgd["depth"], style_pyramid_mode=cfgd["mode"], gram_mode=cfgd["gram"],
        angle_threshold=cfgd["thr"], save_texture=False, learning_rate=1, decay_gamma=0.1, decay_step_size=1,
        loss_weights=dict(LOSS_WEIGHTS), **kw)
    if init is not None:
        with torch.no_grad():
            params = [l.data for l in m.texture.layers] if cfgd["hier"] else [m.texture.data]
            for p, t in zip(params, init):
                p.copy_(t)
    return m.cuda()


def to_cuda(batch):
    return tuple([u.cuda() for u in x] if isinstance(x, list) else (x.cuda() if torch.is_tensor(x) else x) for x in batch)


def test_texture_classes_match_golden():
    require_gpu()
    from stylemesh_amd.model.texture.texture import HierarchicalNeuralTexture, NeuralTexture
    d = load_golden("g1_texture")
    grid, up = T(d["grid"]).cuda(), T(d["upstream"]).cuda()
    flat = NeuralTexture.from_tensor(T(d["layer0"]).clone()).cuda()
    y = flat(grid)
    (y * up).sum().backward()
    assert_close(y, d["flat_out"], 1e-5, 2e-4)
    assert_close(flat.data.grad, d["flat_grad"], 1e-5, 1e-5 * float(np.abs(d["flat_grad"]).max()))
    assert_close(flat.data, d["flat_data_after"], 0, 0)     # normalize() clamped the parameter in place
    hier = HierarchicalNeuralTexture.from_tensor([T(d[f"layer{i}"]).clone() for i in range(4)]).cuda()
    yh = hier(grid)
    (yh * up).sum().backward()
    assert_close(yh, d["hier_out"], 1e-5, 2e-4)
    for i in range(4):
        assert_close(hier.layers[i].data.grad, d[f"hier_grad{i}"], 1e-5, 1e-5 * float(np.abs(d[f"hier_grad{i}"]).max()))
    assert_close(hier.regularizer([8, 4, 2, 0]), d["reg"], 1e-5, 0)
    assert_close(hier.get_image(), d["hier_get_image"], 1e-5, 2e-4)
    with pytest.raises(AssertionError):
        HierarchicalNeuralTexture.from_tensor([torch.zeros(3, 8, 8), torch.zeros(3, 5, 4)])


def test_vgg_and_gram_modules_match_golden():
    require_gpu()
    from stylemesh_amd.model.losses.content_and_style_losses import VGG, ContentAndStyleLoss, GramMatrix
    d = load_golden("g2_vgg")
    vgg = VGG(model_path=vgg_file()).cuda()
    assert sorted(vgg.state_dict()) == sorted(S.seeded_vgg_state(0))
    x = T(d["x"]).cuda().requires_grad_(True)
    keys = ['r11', 'r21', 'r31', 'r41', 'r51', 'r42']
    out = vgg(x, keys)
    for k in keys:
        assert_close(out[k], d["out_" + k], 1e-4, 2e-4 * float(np.abs(d["out_" + k]).max()), k)
    sum((out[k] * T(d["up_" + k]).cuda()).sum() for k in keys).backward()
    grad_close(x.grad, d["grad_x"], "VGG input gradient")
    g3 = load_golden("g3_gram")
    f = torch.zeros(1, 64, 5, 7)
    f[:, :8] = T(g3["f"])
    f = f.cuda().requires_grad_(True)
    G = GramMatrix()(f)
    assert_close(G[0, :8, :8], g3["gram_full"][0], 1e-5, 1e-6)
    with pytest.raises(ValueError, match="No model_path provided"):
        ContentAndStyleLoss(None)
    with pytest.raises(ValueError):
        ContentAndStyleLoss(vgg_file(), style_pyramid_mode="bogus")


@pytest.mark.parametrize("name", ["with_angle_and_depth", "only2d", "flat_single"])
def test_reference_formulation_forward_with_loss(name):
    """forward_with_loss + loss.backward(): autograd through the differentiable HIP classes."""
    d = load_golden("g5_" + name)
    cfgd = FLAGSETS[name]
    m = make_model(cfgd, [T(d[f"init{i}"]) for i in range(4)])
    out = m.forward_with_loss(to_cuda(batch_from_golden(d)), 0, "train")
    out["loss"].backward()
    for k in ("content", "style", "tex_reg", "total"):
        np.testing.assert_allclose(float(m.loss_history[k]["train"][-1]), float(d[f"loss_{k}"].reshape(-1)[0]), rtol=2e-4, err_msg=k)
    params = [l.data for l in m.texture.layers] if cfgd["hier"] else [m.texture.data]
    for i, p in enumerate(params):
        grad_close(p.grad, d[f"grad{i}"], f"{name} grad{i}")


def test_fused_trainer_path_matches_adam_golden():
    """MiniTrainer + training_step + FusedTextureAdam + StepLR = the reference's 5-step trajectory (golden G6)."""
    from stylemesh_amd.trainer import MiniTrainer
    import stylemesh_oracle as O
    d = load_golden("g6_adam_seeded")
    g5 = load_golden("g5_with_angle_and_depth")
    m = make_model(FLAGSETS["with_angle_and_depth"], [T(g5[f"init{i}"]) for i in range(4)])
    batch = batch_from_golden(g5)

    class TwoStepsPerEpoch:
        def train_dataloader(self):
            return iter([batch, batch])

    with tempfile.TemporaryDirectory() as tmp:
        from stylemesh_amd.trainer import JsonlLogger
        tr = MiniTrainer(max_epochs=3, logger=JsonlLogger(tmp), progress=False, limit_train_batches=2)
        tr.fit(m, TwoStepsPerEpoch())
        assert os.path.exists(os.path.join(tmp, "lightning_logs/version_0/scalars.jsonl"))
    # 6 steps were taken (lr 1,1,.1,.1,.01,.01); the golden has the state after 5: replay 5 on a fresh model
    m = make_model(FLAGSETS["with_angle_and_depth"], [T(g5[f"init{i}"]) for i in range(4)])
    (opt,), (sched,) = m.configure_optimizers()
    cb = to_cuda(batch)
    for step in range(5):
        opt.zero_grad()
        m.training_step(cb, step)["loss"].backward()
        opt.step()
        if step % 2 == 1:
            sched.step()
        if step in (0, 1, 4):
            for i, l in enumerate(m.texture.layers):
                texture_close(l.data, T(d[f"p{i}_after{step + 1}"]).clamp(O.CLAMP_LO, O.CLAMP_HI), step, f"layer {i} step {step}")
    assert abs(opt.param_groups[0]["lr"] - 0.01) < 1e-12


def test_optimize_cli_runs_and_saves_texture():
    require_gpu()
    from stylemesh_amd.model.optimize import build_parser, main
    with tempfile.TemporaryDirectory() as tmp:
        args = build_parser().parse_args([
            "--dataset", "synthetic", "--max_images", "3", "--texture_size", "128,128", "--hierarchical",
            "--loss_weight", "content=7e1", "--loss_weight", "style=1e-4", "--loss_weight", "tex_reg=5e3",
            "--style_weights", "1000,1000,10,10,1000", "--vgg_gatys_model_path", "random:7",
            "--style_image_path", "synthetic:3:300x270", "--learning_rate", "1", "--decay_step_size", "3",
            "--max_epochs", "1", "--train_split", "0.7", "--val_split", "0.3", "--index_repeat", "2", "--save_texture",
            "--style_pyramid_mode", "multi", "--angle_threshold", "30", "--pyramid_levels", "2",
            "--min_pyramid_depth", "0.25", "--default_root_dir", tmp])
        model = main(args)
        files = os.listdir(os.path.join(tmp, "lightning_logs/version_0"))
        assert "epoch_0_texture.jpg" in files and "epoch_0_layer0_texture.jpg" in files, files
        assert float(model.texture.layers[0].data.abs().max()) > 0
