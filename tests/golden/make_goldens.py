#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING THE REFERENCE (container only).

    python tests/golden/make_goldens.py            # needs /root/reference, CPU only

The reference (lukasHoel/stylemesh, mounted read-only at /root/reference) has no tests or golden
vectors of its own, so the pin for the oracle is: outputs of the reference's
own Python, run here on seeded inputs, committed as small ``.npz`` files. Nothing of the reference
(source, bytecode) is written anywhere: only input / output arrays.

What has to be faked to import it (SURVEY.md section 8 c):
  * ``torchvision`` / ``pytorch_lightning`` / ``cv2`` are not installed -> minimal ``sys.modules`` stubs
    (Compose / Lambda / Normalize real; LightningModule = nn.Module + no-op logger).
  * ``Tensor.type_as`` on CPU returns the same leaf tensor, which makes the reference's
    ``style_loss += l`` (content_and_style_losses.py:298-299,340) raise; on CUDA ``type_as`` copies.
    The shim returns ``self.clone()`` in exactly that case - numerically the CUDA behaviour.
  * ``np.int`` (removed from NumPy) is used at data/scannet_dataset.py:365-366 -> ``np.int = int``.

VGG weights are regenerated from a seed (``stylemesh_amd.data.synthetic.seeded_vgg_state``), never stored.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("STYLEMESH_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

from stylemesh_amd.data import synthetic as S  # noqa: E402
from stylemesh_amd.data import view_contract as VC  # noqa: E402


# --------------------------------------------------------------------------------------------
# stubs + shims
# --------------------------------------------------------------------------------------------
def install_stubs():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvu = types.ModuleType("torchvision.utils")

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Lambda:
        def __init__(self, fn):
            self.fn = fn

        def __call__(self, x):
            return self.fn(x)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, x):
            m = torch.as_tensor(self.mean, dtype=x.dtype).view(-1, 1, 1)
            s = torch.as_tensor(self.std, dtype=x.dtype).view(-1, 1, 1)
            return (x - m) / s

    class ToTensor:
        def __call__(self, x):
            x = np.asarray(x)
            if x.ndim == 2:
                x = x[:, :, None]
            t = torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))
            return t.float() / 255 if t.dtype == torch.uint8 else t

    class _Placeholder:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            raise NotImplementedError("placeholder transform")

    tvt.Compose, tvt.Lambda, tvt.Normalize, tvt.ToTensor = Compose, Lambda, Normalize, ToTensor
    tvt.ToPILImage = tvt.Resize = tvt.RandomCrop = _Placeholder
    tvu.make_grid = lambda *a, **k: None
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvf.InterpolationMode = types.SimpleNamespace(NEAREST=0, BILINEAR=2)
    tvt.functional = tvf
    tvt.__path__ = []  # so that ``import torchvision.transforms.functional`` resolves
    tv.transforms, tv.utils = tvt, tvu
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.utils": tvu,
                        "torchvision.transforms.functional": tvf})

    pl = types.ModuleType("pytorch_lightning")

    class _Exp:
        def add_scalar(self, *a, **k):
            pass

        add_scalars = add_image = add_scalar

    class _Logger:
        experiment = _Exp()

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.current_epoch = 0
            self.logger = _Logger()

        def save_hyperparameters(self, *a, **k):
            pass

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass

    pl.LightningModule, pl.LightningDataModule = LightningModule, LightningDataModule
    sys.modules["pytorch_lightning"] = pl

    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR, cv2.INTER_NEAREST = 1, 0

    def resize(img, size_wh, interpolation=1):
        w, h = size_wh
        if interpolation == 1:
            return VC.resize_bilinear_np(np.asarray(img, dtype=np.float32), (h, w)).astype(np.asarray(img).dtype)
        return VC.resize_nearest_np(np.asarray(img), (h, w))

    # The stub stands in for OpenCV (absent from this image) in the reference's data code, so fixture g7's depth / mask path
    # is pinned only as far as this function IS cv2.resize. Known answers of cv2.resize(..., INTER_LINEAR) stated here, from
    # its documented sampling rule - source coordinate (dst + 0.5) * in / out - 0.5, clamped to the image, two-tap linear
    # weights, no anti-aliasing - worked by hand:
    #   [0 10 20 30] -> 2 samples: source 0.5, 2.5                     -> 5, 25
    #   [0 10]       -> 4 samples: source -0.25, 0.25, 0.75, 1.25       -> 0, 2.5, 7.5, 10
    #   [0 .. 6]     -> 3 samples: source 2/3, 3, 16/3                  -> 2/3, 3, 16/3
    #   [[0 10] [20 30]] -> 1 x 1: source (0.5, 0.5)                    -> 15;   3 x 3 -> 3 x 3: the identity
    def _row(v, n):
        return resize(np.asarray([v], dtype=np.float32), (n, 1))[0]
    np.testing.assert_allclose(_row([0, 10, 20, 30], 2), [5, 25], rtol=0, atol=1e-5)
    np.testing.assert_allclose(_row([0, 10], 4), [0, 2.5, 7.5, 10], rtol=0, atol=1e-5)
    np.testing.assert_allclose(_row(list(range(7)), 3), [2 / 3, 3, 16 / 3], rtol=0, atol=1e-5)
    np.testing.assert_allclose(resize(np.asarray([[0, 10], [20, 30]], dtype=np.float32), (1, 1)), [[15]], rtol=0, atol=1e-5)
    eye = np.arange(9, dtype=np.float32).reshape(3, 3)
    np.testing.assert_array_equal(resize(eye, (3, 3)), eye)
    # INTER_NEAREST: source index floor(dst * in / out)
    np.testing.assert_array_equal(resize(np.asarray([[0, 1, 2, 3, 4]], dtype=np.int32), (2, 1), interpolation=0), [[0, 2]])

    cv2.resize = resize
    sys.modules["cv2"] = cv2

    np.int = int  # noqa: NPY001 (reference uses the removed alias)

    orig_type_as = torch.Tensor.type_as

    def type_as(self, other):
        out = orig_type_as(self, other)
        if out is self and self.requires_grad and self.is_leaf:
            return self.clone()
        return out

    torch.Tensor.type_as = type_as


install_stubs()
sys.path.insert(0, REF)
from model.texture.texture import NeuralTexture, HierarchicalNeuralTexture  # noqa: E402
from model.losses import content_and_style_losses as RL  # noqa: E402
from model.model import TextureOptimizationStyleTransferPipeline  # noqa: E402
from model.losses.rgb_transform import pre as ref_pre  # noqa: E402
import data.scannet_dataset as ref_scannet  # noqa: E402
import data.matterport_dataset as ref_matterport  # noqa: E402
import data.utils as ref_utils  # noqa: E402


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def vgg_path(seed):
    f = tempfile.NamedTemporaryFile(suffix=".pth", delete=False)
    torch.save(S.seeded_vgg_state(seed), f.name)
    return f.name


def seeded_texture_layers(seed, W, H, n_layers, scale=40.0):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_layers):
        w, h = W // 2 ** i, H // 2 ** i
        t = (S.smooth_noise(rng, 3, h, w, cells=4) - 0.5) * 2 * scale / (i + 1)
        out.append(torch.from_numpy(t.astype(np.float32)))
    return out


# --------------------------------------------------------------------------------------------
# G1: texture sampling forward + backward (reference model/texture/texture.py:22-100)
# --------------------------------------------------------------------------------------------
def g1_texture():
    rng = np.random.default_rng(11)
    W, H = 40, 24  # non-square to catch transposes
    h, w = 20, 28
    grid = rng.uniform(-1, 1, (1, h, w, 2)).astype(np.float32)
    # special cases: exact corners, centre, out-of-range (border clamp), background uv == 0 -> grid == -1
    special = [(-1, -1), (1, 1), (0, 0), (1.2, -3.0), (-1, 1), (1, -1), (0.999999, 0.999999), (-1.5, 0.3)]
    for i, (gx, gy) in enumerate(special):
        grid[0, 0, i] = (gx, gy)
    grid[0, 1, :10] = -1.0
    # texel-centre hits: x = 2*i/(W-1)-1
    for i in range(8):
        grid[0, 2, i] = (2 * (3 * i) / (W - 1) - 1, 2 * (2 * i) / (H - 1) - 1)
    grid = torch.from_numpy(grid)
    upstream = torch.from_numpy(rng.standard_normal((1, 3, h, w)).astype(np.float32))

    layers = seeded_texture_layers(5, W, H, 4, scale=200.0)  # beyond the clamp range on purpose
    flat = NeuralTexture.from_tensor(layers[0].clone())
    y = flat(grid)
    (y * upstream).sum().backward()
    hier = HierarchicalNeuralTexture.from_tensor([l.clone() for l in layers])
    yh = hier(grid)
    (yh * upstream).sum().backward()
    reg = hier.regularizer([8, 4, 2, 0])
    arrays = dict(grid=grid, upstream=upstream, flat_out=y, flat_grad=flat.data.grad,
                  flat_data_after=flat.data, hier_out=yh, reg=reg,
                  probe_out=NeuralTexture.from_tensor(torch.arange(12.).view(1, 3, 4))(
                      torch.tensor([[[[-1., -1.], [1., 1.], [0., 0.], [1.2, -3.]]]])))
    for i, l in enumerate(layers):
        arrays[f"layer{i}"] = l
        arrays[f"hier_grad{i}"] = hier.layers[i].data.grad
    ident = hier.get_image()
    arrays["hier_get_image"] = ident
    save("g1_texture", **arrays)


# --------------------------------------------------------------------------------------------
# G2: VGG feature stack forward + input gradient (reference content_and_style_losses.py:7-70)
# --------------------------------------------------------------------------------------------
VGG_SEED = 7
LAYERS = ['r11', 'r21', 'r31', 'r41', 'r51', 'r42']


def g2_vgg():
    rng = np.random.default_rng(21)
    vgg = RL.VGG(model_path=vgg_path(VGG_SEED))
    x = torch.from_numpy(((S.smooth_noise(rng, 3, 36, 52) - 0.45) * 255).astype(np.float32))[None]
    x.requires_grad_(True)
    out = vgg(x, LAYERS + ['r12', 'p1', 'r22', 'r34', 'r44', 'p4'])
    ups = {k: torch.from_numpy(rng.standard_normal(tuple(out[k].shape)).astype(np.float32)) for k in LAYERS}
    sum((out[k] * ups[k]).sum() for k in LAYERS).backward()
    arrays = dict(x=x, grad_x=x.grad, vgg_seed=VGG_SEED)
    for k, v in out.items():
        arrays["out_" + k] = v
    for k, v in ups.items():
        arrays["up_" + k] = v
    save("g2_vgg", **arrays)


# --------------------------------------------------------------------------------------------
# G3: Gram / masked features / MSE (reference content_and_style_losses.py:74-80,136-143,265)
# --------------------------------------------------------------------------------------------
def g3_gram():
    rng = np.random.default_rng(31)
    f = torch.from_numpy(rng.standard_normal((1, 8, 5, 7)).astype(np.float32)).requires_grad_(True)
    mask = torch.from_numpy((rng.random((1, 1, 5, 7)) > 0.4).astype(np.float32))
    target = torch.from_numpy(rng.standard_normal((1, 8, 8)).astype(np.float32))
    g_full = RL.GramMatrix()(f)
    mf = RL.masked_features(f, mask)
    g_masked = RL.GramMatrix()(mf)
    loss = nn.MSELoss()(target, g_masked)
    loss.backward()
    grad_style = f.grad.clone()
    f.grad = None
    tgt_feat = torch.from_numpy(rng.standard_normal((1, 8, 5, 7)).astype(np.float32))
    closs = nn.MSELoss()(RL.masked_features(tgt_feat, mask), RL.masked_features(f, mask))
    closs.backward()
    empty = RL.masked_features(f, torch.zeros_like(mask))
    save("g3_gram", f=f, mask=mask, target=target, gram_full=g_full, masked_n=mf.shape[2], gram_masked=g_masked,
         style_mse=loss, grad_style=grad_style, tgt_feat=tgt_feat, content_mse=closs, grad_content=f.grad,
         empty_shape=np.array(empty.shape), empty_gram=RL.GramMatrix()(empty))


# --------------------------------------------------------------------------------------------
# G4: style image pyramid + style targets (reference content_and_style_losses.py:83-133,273-286)
# --------------------------------------------------------------------------------------------
def g4_style():
    arrays = {}
    # shapes only, for a range of sizes (incl. The Scream 1200x1528 and Starry Night 970x768: h x w)
    for (h, w) in [(600, 520), (1528, 1200), (768, 970), (300, 260), (256, 256), (2048, 1400)]:
        img = torch.zeros(1, 1, h, w)
        pyr = RL.image_pyramid(img, [0, 1, 2, 3, 4], reverse=True)
        arrays[f"shapes_{h}x{w}"] = np.array([p.shape[2:] for p in pyr])
        pyr_fwd = RL.image_pyramid(img, [0, 1, 2, 3, 4], reverse=False)
        arrays[f"shapes_fwd_{h}x{w}"] = np.array([p.shape[2:] for p in pyr_fwd])
    style = S.style_image(41, 600, 520)[None]
    pyr = RL.image_pyramid(style, [0, 1, 2, 3, 4], reverse=True)
    arrays["style_seed"] = 41
    arrays["pyr0_sub"] = pyr[0][0, :, ::5, ::5]
    arrays["pyr1_sub"] = pyr[1][0, :, ::5, ::5]
    loss = RL.ContentAndStyleLoss(vgg_path(VGG_SEED))
    loss.set_style_image(style)
    for li, layer in enumerate(loss.style_layers):
        for lvl in (0, 1, 2):
            g = loss.style_targets[li][lvl][0]
            if g.shape[0] <= 128:
                arrays[f"target_{layer}_{lvl}"] = g
            else:
                arrays[f"target_{layer}_{lvl}_sub"] = g[::5, ::7]
                arrays[f"target_{layer}_{lvl}_sum"] = g.double().sum()
                arrays[f"target_{layer}_{lvl}_sqsum"] = (g.double() ** 2).sum()
    save("g4_style", **arrays)


# --------------------------------------------------------------------------------------------
# G5 / G6 / G8: the pipeline (reference model/model.py:143-401)
# --------------------------------------------------------------------------------------------
SMALL_VIEW_HW = (40, 56)
SMALL_LEVEL_HW = [(40, 56), (64, 88)]
SMALL_ROOM = (6.0, 4.5, 2.8)
STYLE_HW = (300, 270)
STYLE_SEED = 43
FLAGSETS = {
    # name: (hierarchical, style_pyramid_mode, gram_mode, angle_threshold, use_angle, use_depth, n_levels)
    # = scripts/train/optimize_texture_scannet_{only2D,with_angle,with_angle_and_depth,dip}.sh flag sets
    "only2d": dict(hier=True, mode="single", gram="current", thr=3000, angle=False, depth=False, nlev=1),
    "with_angle": dict(hier=True, mode="multi", gram="current", thr=30, angle=True, depth=False, nlev=1),
    "with_angle_and_depth": dict(hier=True, mode="multi", gram="current", thr=30, angle=True, depth=True, nlev=2),
    "dip_average": dict(hier=True, mode="single", gram="average", thr=3000, angle=False, depth=False, nlev=1),
    "flat_single": dict(hier=False, mode="single", gram="current", thr=60, angle=True, depth=True, nlev=2),
}
LOSS_WEIGHTS = {"content": 7e1, "style": 1e-4, "tex_reg": 5e3}
STYLE_WEIGHTS = [1000., 1000., 10., 10., 1000.]
TEX = 64


def small_view(seed, nlev):
    room = S.BoxRoom(SMALL_ROOM)
    level_hw = SMALL_LEVEL_HW[:nlev] if nlev > 1 else [SMALL_VIEW_HW]
    # min depth 0.9 m: uv_h = 32 * d / 0.9 spans both levels (40, 64) inside the small room
    return S.make_view(seed, view_hw=SMALL_VIEW_HW, level_hw=level_hw, level_heights=[h for h, _ in level_hw],
                       min_pyramid_depth=0.9, room=room)


def make_model(cfg, vggp, init_layers=None):
    style = S.style_image(STYLE_SEED, *STYLE_HW)
    m = TextureOptimizationStyleTransferPipeline(
        W=TEX, H=TEX, hierarchical_texture=cfg["hier"], hierarchical_layers=4, style_image=style,
        style_weights=STYLE_WEIGHTS, vgg_gatys_model_path=vggp, use_angle_weight=cfg["angle"],
        use_depth_scaling=cfg["depth"], style_pyramid_mode=cfg["mode"], gram_mode=cfg["gram"],
        angle_threshold=cfg["thr"], save_texture=False, learning_rate=1, decay_gamma=0.1, decay_step_size=1,
        loss_weights=dict(LOSS_WEIGHTS))
    if init_layers is not None:
        with torch.no_grad():
            if cfg["hier"]:
                for l, t in zip(m.texture.layers, init_layers):
                    l.data.copy_(t)
            else:
                m.texture.data.copy_(init_layers[0])
    return m


def tex_params(m):
    return [l.data for l in m.texture.layers] if m.hierarchical_texture else [m.texture.data]


def batch_arrays(batch, prefix=""):
    rgb, e, i, depth, dl, rl, ol, w, idx, uvs, mask, ang, angd = batch
    out = {prefix + "rgb": rgb, prefix + "depth": depth, prefix + "depth_level": dl, prefix + "rounded_level": rl,
           prefix + "other_level": ol, prefix + "interp_weight": w, prefix + "mask": mask,
           prefix + "angle_guidance": ang, prefix + "angle_degrees": angd}
    for k, u in enumerate(uvs):
        out[prefix + f"uv{k}"] = u
    return out


def g5_pipeline():
    vggp = vgg_path(VGG_SEED)
    for name, cfg in FLAGSETS.items():
        batch = small_view(3, cfg["nlev"])
        init = seeded_texture_layers(9, TEX, TEX, 4, scale=30.0)
        m = make_model(cfg, vggp, init)
        arrays = batch_arrays(batch)
        n_steps = 3 if cfg["gram"] == "average" else 1
        for s in range(n_steps):
            for p in tex_params(m):
                p.grad = None
            m.forward_with_loss(batch, s, "train")["loss"].backward()
            tag = f"_s{s}" if n_steps > 1 else ""
            for lt in ("content", "style", "tex_reg", "total"):
                arrays[f"loss_{lt}{tag}"] = m.loss_history[lt]["train"][-1]
            for i, p in enumerate(tex_params(m)):
                arrays[f"grad{i}{tag}"] = p.grad.clone()
        with torch.no_grad():
            for k, p in enumerate(m.forward(batch)):
                arrays[f"pred{k}"] = p
        for i, t in enumerate(init):
            arrays[f"init{i}"] = t
        save("g5_" + name, **arrays)


def g6_adam():
    """Lightning's automatic optimisation order (zero_grad -> training_step -> backward -> step; StepLR per
    epoch) replayed by hand: 2 steps per epoch, decay_step_size = 1 -> lr 1, 1, 0.1, 0.1, 0.01."""
    vggp = vgg_path(VGG_SEED)
    cfg = FLAGSETS["with_angle_and_depth"]
    batch = small_view(3, cfg["nlev"])
    for init_name, init in (("zero", None), ("seeded", seeded_texture_layers(9, TEX, TEX, 4, scale=30.0))):
        m = make_model(cfg, vggp, init)
        (opt,), (sched,) = m.configure_optimizers()
        arrays = {}
        grads_first = None
        for step in range(5):
            opt.zero_grad()
            m.forward_with_loss(batch, step, "train")["loss"].backward()
            if step == 0:
                grads_first = [p.grad.clone() for p in tex_params(m)]
            opt.step()
            if step % 2 == 1:
                sched.step()
            if step in (0, 1, 4):
                for i, p in enumerate(tex_params(m)):
                    arrays[f"p{i}_after{step + 1}"] = p.detach().clone()
                    st = opt.state[p]
                    arrays[f"m{i}_after{step + 1}"] = st["exp_avg"].clone()
                    arrays[f"v{i}_after{step + 1}"] = st["exp_avg_sq"].clone()
            arrays[f"loss_total_step{step}"] = m.loss_history["total"]["train"][-1]
        for i, g in enumerate(grads_first):
            arrays[f"grad{i}_step0"] = g
        save("g6_adam_" + init_name, **arrays)


def g8_multiview():
    """R-GPU step semantics (SURVEY.md section 8 e): R independent B = 1 evaluations on the same texture,
    gradients averaged (DDP mean), one Adam step."""
    vggp = vgg_path(VGG_SEED)
    cfg = FLAGSETS["with_angle_and_depth"]
    init = seeded_texture_layers(9, TEX, TEX, 4, scale=30.0)
    seeds = [3, 4, 6, 8]
    grads = []
    arrays = {"view_seeds": np.array(seeds)}
    for s in seeds:
        m = make_model(cfg, vggp, init)
        m.forward_with_loss(small_view(s, cfg["nlev"]), 0, "train")["loss"].backward()
        grads.append([p.grad.clone() for p in tex_params(m)])
        arrays[f"loss_total_view{s}"] = m.loss_history["total"]["train"][-1]
    m = make_model(cfg, vggp, init)
    m.texture.layers[0](torch.zeros(1, 2, 2, 2))  # the clamp every forward starts with (texture.py:47)
    (opt,), _ = m.configure_optimizers()
    for i, p in enumerate(tex_params(m)):
        p.grad = torch.stack([g[i] for g in grads]).mean(0)
        arrays[f"mean_grad{i}"] = p.grad.clone()
    opt.step()
    for i, p in enumerate(tex_params(m)):
        arrays[f"p{i}_after"] = p.detach().clone()
    save("g8_multiview", **arrays)


# --------------------------------------------------------------------------------------------
# G7: depth levels + mask (reference data/scannet_dataset.py:308-366, data/matterport_dataset.py:295-349)
# --------------------------------------------------------------------------------------------
def g7_contract():
    rng = np.random.default_rng(71)
    me = types.SimpleNamespace(min_pyramid_depth=0.25, levels=np.linspace(256, 960, 5)[:4])
    known = np.array([[0, 0.5, 2.0, 2.5, 2.6875, 3.0], [3.375, 4.0, 5.0, 6.125, 7.0, 9.0]])
    out_known = ref_scannet.ScanNetDataset.calculate_depth_level(me, None, known, None, None)
    depth = (S.smooth_noise(rng, 1, 24, 30)[0] * 9.0).astype(np.float64)
    depth[3:6, 4:9] = 0
    out = ref_scannet.ScanNetDataset.calculate_depth_level(me, None, depth, None, None)
    me2 = types.SimpleNamespace(min_pyramid_depth=0.2, levels=np.array([256., 432., 608., 784.]))
    out_mp = ref_matterport.MatterportDataset.calculate_depth_level(me2, None, depth.astype(np.float32), None, None)
    uv = rng.random((48, 60, 3)).astype(np.float32)
    uv[10:20, 5:25] = 0
    uv[30:34, 40:50, 0] = 0  # only u == 0 -> still valid
    mask_scannet = np.asarray(ref_scannet.ScanNetDataset.calculate_mask(me, uv, depth))
    mask_mp = np.asarray(ref_matterport.MatterportDataset.calculate_mask(me2, uv))
    rgb = torch.from_numpy(S.smooth_noise(rng, 3, 6, 7))
    save("g7_contract", known_depth=known, known_cont=out_known[0], known_rounded=out_known[1],
         known_other=out_known[2], known_weight=out_known[3], depth=depth, cont=out[0], rounded=out[1],
         other=out[2], weight=out[3], mp_cont=out_mp[0], mp_rounded=out_mp[1], mp_other=out_mp[2],
         mp_weight=out_mp[3], uv=uv, mask_scannet=mask_scannet, mask_matterport=mask_mp,
         rgb01=rgb, rgb_pre=ref_pre()(rgb.clone()))


# --------------------------------------------------------------------------------------------
# G9: reprojection warp of the evaluation metric (reference data/utils.py:36-194)
# --------------------------------------------------------------------------------------------
def reproject_case(seed, hw=(48, 64)):
    """Two nearby seeded cameras in the small box room: intrinsics, poses, depths, a smooth target image."""
    rng = np.random.default_rng(seed)
    room = S.BoxRoom(SMALL_ROOM)
    L = np.asarray(SMALL_ROOM)
    pos = L * np.array([0.35, 0.4, 0.5]) + rng.uniform(-0.2, 0.2, 3)
    yaw, pitch = rng.uniform(0, 2 * np.pi), rng.uniform(-0.15, 0.15)
    pos2 = pos + rng.uniform(-0.25, 0.25, 3) * np.array([1, 1, 0.3])
    yaw2, pitch2 = yaw + rng.uniform(-0.25, 0.25), pitch + rng.uniform(-0.08, 0.08)
    K, c2w_src = S.camera_matrices(pos, yaw, pitch, hw)
    _, c2w_tar = S.camera_matrices(pos2, yaw2, pitch2, hw)
    _, _, d_src = room.render(pos, yaw, pitch, hw)
    _, _, d_tar = room.render(pos2, yaw2, pitch2, hw)
    d_src[5:9, 10:20] = 0                                   # a sensor hole in the source depth
    color_tar = S.smooth_noise(rng, 3, hw[0], hw[1]).astype(np.float32) * 100.0
    return K, c2w_src, c2w_tar, d_src.astype(np.float32), d_tar.astype(np.float32), color_tar


def g9_reproject():
    arrays = {}
    for n, seed in enumerate((3, 8, 21)):
        K, c2w_s, c2w_t, d_s, d_t, col = reproject_case(seed)
        H, W = d_s.shape
        t = lambda a: torch.from_numpy(a)[None]
        mask_tar = t(d_t) > 0
        color, mask = ref_utils.reproject(t(c2w_s), t(c2w_t), W, H, t(K), t(d_s)[:, None], t(d_t)[:, None], t(col),
                                          mask_tar)
        pts = ref_utils.unproject(t(c2w_s).transpose(1, 2), t(K), t(d_s)[:, None])
        arrays.update({f"K{n}": K, f"c2w_src{n}": c2w_s, f"c2w_tar{n}": c2w_t, f"depth_src{n}": d_s, f"depth_tar{n}": d_t,
                       f"color_tar{n}": col, f"out_color{n}": color[0], f"out_mask{n}": mask[0],
                       f"unproject{n}": pts[0]})
    save("g9_reproject", **arrays)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9"]
    fns = dict(g1=g1_texture, g2=g2_vgg, g3=g3_gram, g4=g4_style, g5=g5_pipeline, g6=g6_adam, g7=g7_contract,
               g8=g8_multiview, g9=g9_reproject)
    for w in which:
        fns[w]()
