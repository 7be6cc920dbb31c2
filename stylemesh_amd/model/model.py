"""``TextureOptimizationStyleTransferPipeline`` with the reference's LightningModule surface
(model/model.py:16-420) over the fused HIP step.

Constructor arguments, attribute names, hook names, loss names and log keys are the reference's. Two ways to
run a step:
* ``training_step`` (what a Trainer calls) uses the fused engine (``stylemesh_amd.runtime.engine.StepEngine``):
  forward + hand-written backward deposit the data-term gradient straight into ``param.grad`` (which aliases the
  engine's gradient arena); the returned ``loss`` is a detached scalar whose ``backward()`` is a no-op, and
  ``configure_optimizers`` returns the fused regulariser + Adam + clamp kernel wrapped as an optimizer.
* ``forward`` / ``forward_with_loss`` give the same step as an autograd graph over the differentiable classes of
  ``model/texture`` and ``model/losses``, with the level masks / pixel weights taken from the fused engine's per-view
  kernels (used by the parity tests; slower: dense copies per call).
"""
from __future__ import annotations

import os

import torch

from ..runtime import ops
from ..runtime.engine import EngineConfig, StepEngine
from .losses.content_and_style_losses import ContentAndStyleLoss
from .losses.rgb_transform import post
from .texture.texture import HierarchicalNeuralTexture, NeuralTexture, to_image

try:  # subclass the real LightningModule when it is installed (it is not in the build image)
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # pragma: no cover - exercised in this image
    class _NullExperiment:
        def add_scalar(self, *a, **k):
            pass
        add_scalars = add_image = add_scalar

    class _NullLogger:
        experiment = _NullExperiment()

    class _Base(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.current_epoch = 0
            self.logger = _NullLogger()

        def save_hyperparameters(self, *a, **k):
            pass


class FusedTextureAdam:
    """Optimizer facade over ``sm_adam_fused`` (Adam lr/betas/eps of model/model.py:395, weight_decay 0)."""

    def __init__(self, module, lr):
        self.module = module
        self.param_groups = [{"lr": lr, "initial_lr": lr, "params": list(module.texture.parameters())}]
        self.world_size = 1

    def zero_grad(self, set_to_none=False):
        pass   # the fused update leaves a zeroed gradient arena

    def step(self, closure=None):
        if closure is not None:
            closure()
        eng = self.module._engine
        eng.cfg.learning_rate = self.param_groups[0]["lr"]
        eng.cfg.decay_gamma, eng.epoch = 1.0, 0    # the scheduler owns the learning rate
        reducer = getattr(self.module, "grad_reducer", None)
        if eng.use_deferred_exchange(reducer):
            # STYLEMESH_DEFERRED_EXCHANGE=1 with an owner-aware reducer: critical exchange of the shared chunks, the
            # single-owner ones in the background (``StepEngine.exchange_and_update_deferred``)
            eng._world_size = self.world_size
            eng.exchange_and_update_deferred(self.world_size, reducer)
        elif eng.use_pipelined_exchange(reducer):
            # STYLEMESH_PIPELINE_EXCHANGE=1: ``step_compute`` left the gradient un-exchanged - all-reduce it in pieces,
            # the update of each arena range issued as its sums arrive (``StepEngine.exchange_and_update``)
            eng.exchange_and_update(self.world_size, reducer)
        else:
            eng.optimizer_step(self.world_size)

    def _eng(self):
        return self.module._ensure_engine(self.param_groups[0]["params"][0].device)

    def state_dict(self):
        eng = self._eng()
        return {"step": eng.step_count, "m": eng.arena.m.clone(), "v": eng.arena.v.clone(), "lr": self.param_groups[0]["lr"],
                "touched": None if eng.touched is None else eng.touched.clone()}

    def load_state_dict(self, sd):
        """Restores step count, moments and learning rate. Load the texture itself FIRST (``from_tensor`` switches the
        engine to the dense update; the saved ever-touched flags, taken at the same moment as the moments, switch the
        sparse update back on)."""
        eng = self._eng()
        eng.step_count = int(sd["step"])
        eng.arena.m.copy_(sd["m"])
        eng.arena.v.copy_(sd["v"])
        t = sd.get("touched")
        eng.touched = None if t is None else t.to(eng.device, torch.int32).clone()
        self.param_groups[0]["lr"] = sd["lr"]


class StepLR:
    """torch.optim.lr_scheduler.StepLR semantics (model/model.py:397-399) for any object with param_groups."""

    def __init__(self, optimizer, step_size, gamma=0.1):
        self.optimizer, self.step_size, self.gamma, self.last_epoch = optimizer, step_size, gamma, 0

    def step(self):
        self.last_epoch += 1
        for g in self.optimizer.param_groups:
            g["lr"] = g["initial_lr"] * self.gamma ** (self.last_epoch // self.step_size)


class TextureOptimizationStyleTransferPipeline(_Base):
    states = ["train", "val"]
    loss_types = ["tex_reg", "content", "style", 'total']
    default_loss_weights = {l: 0.0 for l in loss_types}

    def __init__(self, W, H, hierarchical_texture=True, hierarchical_layers=4, random_texture_init=False,
                 style_image=None, style_layers=ContentAndStyleLoss.style_layers,
                 content_layers=ContentAndStyleLoss.content_layers, style_weights=ContentAndStyleLoss.style_weights,
                 content_weights=ContentAndStyleLoss.content_weights, vgg_gatys_model_path=None,
                 use_angle_weight=True, use_depth_scaling=True, style_pyramid_mode='single', gram_mode='current',
                 angle_threshold=60, log_images_nth=-1, save_texture=True, texture_dir="", texture_prefix="",
                 learning_rate=1e-3, decay_gamma=0.1, decay_step_size=30, loss_weights=default_loss_weights,
                 tex_reg_weights=None, extra_args={}):
        super().__init__()
        orig_style_image, style_image = style_image, None
        self.save_hyperparameters()
        style_image = orig_style_image

        self.hierarchical_texture, self.hierarchical_layers, self.C = hierarchical_texture, hierarchical_layers, 3
        self.random_texture_init = random_texture_init
        if hierarchical_texture:
            self.texture = HierarchicalNeuralTexture(W, H, self.C, hierarchical_layers, random_texture_init)
        else:
            self.texture = NeuralTexture(W, H, self.C, random_texture_init)
        self.tex_reg_weights = tex_reg_weights
        if hierarchical_texture and not tex_reg_weights:
            self.tex_reg_weights = [pow(2, hierarchical_layers - i - 1) for i in range(hierarchical_layers)]
            self.tex_reg_weights[-1] = 0
            print(f"No tex_reg_weights specified. Setting them to {self.tex_reg_weights}")
        if hierarchical_texture and hierarchical_layers != len(self.tex_reg_weights):
            raise ValueError(
                f"Have {hierarchical_layers} texture layers, but only {len(self.tex_reg_weights)} weights specified")

        self.loss_history = {loss: {k: [] for k in self.states} for loss in self.loss_types}
        self.loss_weights = dict(loss_weights) if loss_weights else {}
        for loss in self.loss_history.keys():
            if loss not in self.loss_weights:
                self.loss_weights[loss] = self.default_loss_weights[loss]
                print(f"No weight specified for the '{loss}' loss. Setting it to {self.loss_weights[loss]}")

        self.vgg_gatys_model_path = vgg_gatys_model_path
        self.vgg_loss = ContentAndStyleLoss(vgg_gatys_model_path, style_layers, content_layers, style_weights,
                                            content_weights, angle_threshold=angle_threshold,
                                            style_pyramid_mode=style_pyramid_mode, gram_mode=gram_mode)
        self.style_image = style_image
        self.orig_style_image = style_image.clone()
        self.angle_threshold, self.style_pyramid_mode, self.gram_mode = angle_threshold, style_pyramid_mode, gram_mode
        self.use_angle_weight, self.use_depth_scaling = use_angle_weight, use_depth_scaling
        self.learning_rate, self.decay_gamma, self.decay_step_size = learning_rate, decay_gamma, decay_step_size
        self.log_images_nth, self.save_texture = log_images_nth, save_texture
        self.texture_prefix, self.texture_dir = texture_prefix, texture_dir
        self.batches_per_epoch = {k: 0 for k in self.states}
        self.train_epoch_end = False
        self.val_epoch_end = False
        self._engine = None
        self.world_size = 1
        self.grad_reducer = None   # callable(flat_gradient_arena) for the multi-GPU all-reduce

    # ------------------------------------------------------------------ fused engine plumbing
    def _texture_params(self):
        return [l.data for l in self.texture.layers] if self.hierarchical_texture else [self.texture.data]

    def _ensure_engine(self, device) -> StepEngine:
        """Create the fused engine on first use and make the texture Parameters (and their ``.grad``) views of its
        arenas, so that the module's parameters ARE the memory the kernels update."""
        if self._engine is None:
            cfg = EngineConfig(tex_w=self.texture.W, tex_h=self.texture.H, hierarchical=self.hierarchical_texture,
                               n_layers=self.hierarchical_layers, style_layers=list(self.vgg_loss.style_layers),
                               content_layers=list(self.vgg_loss.content_layers),
                               style_weights=list(self.vgg_loss.style_weights),
                               content_weights=list(self.vgg_loss.content_weights), angle_threshold=self.angle_threshold,
                               style_pyramid_mode=self.style_pyramid_mode, gram_mode=self.gram_mode,
                               use_angle_weight=self.use_angle_weight, use_depth_scaling=self.use_depth_scaling,
                               loss_weights=dict(self.loss_weights), tex_reg_weights=self.tex_reg_weights,
                               learning_rate=self.learning_rate, decay_gamma=self.decay_gamma,
                               decay_step_size=self.decay_step_size)
            eng = StepEngine(cfg, self.vgg_loss.vgg.state_dict(), device)
            params = [p.detach() for p in self._texture_params()]
            # an untouched zero-initialised texture (texture.py:26-28) keeps the engine's sparse update; any other
            # content (random_init, from_tensor, a loaded checkpoint) makes every texel part of the update
            if any(bool(p.any()) for p in params):
                eng.load_texture(params)
            for p, view, g in zip(self._texture_params(), eng.layers, eng.grads):
                p.data = view
                p.grad = g
            style = self.style_image
            eng.set_style_image(style if style.dim() == 3 else style[0])
            eng.set_scene(getattr(self, "scene_identity", None))
            self._engine = eng
        return self._engine

    # ------------------------------------------------------------------ reference-formulation forward
    def forward(self, x):
        image, _, _, _, _, _, _, _, _, uv_map, _, _, _ = x
        if self.style_image.shape != image.shape and len(self.style_image.shape) != 4:
            self.style_image = self.style_image.repeat(image.shape[0], 1, 1, 1).type_as(image)
            self.vgg_loss.vgg.to(image.device)
            self.vgg_loss.set_style_image(self.style_image)
        return [self.texture(v) for v in uv_map]

    def tex_reg_loss(self):
        if self.hierarchical_texture:
            return self.texture.regularizer(self.tex_reg_weights)
        return torch.zeros(1).type_as(self.texture.data)

    def update_batch_count(self, batch_idx, state):
        self.batches_per_epoch[state] = max(self.batches_per_epoch[state], batch_idx + 1)

    def _level_constants(self, batch, shapes):
        """Per UV level (mask, pixel weight or None) from the SAME kernels the fused engine's ``set_view`` runs
        (``sm_level_masks``: eroded level masks + interpolation weights, model/model.py:204-239; ``sm_level_maps``: their
        nearest up-sampling, the bilinear angle map and the product weight cos(theta) * depth weight, :195-202,245-251)."""
        (_, _, _, _, _, rounded, other, interp_w, _, _, mask, angle_guidance, angle_degrees) = batch
        dev = mask.device
        h, w = mask.shape[-2:]
        n = len(shapes)
        f32 = lambda t: t.to(dev, torch.float32).contiguous()
        mask_u8 = mask[0].to(torch.uint8).contiguous()
        ag, adeg = f32(angle_guidance[0, 0]), f32(angle_degrees[0, 0])
        if self.use_depth_scaling:
            E, Wt = torch.empty(n, h, w, device=dev), torch.empty(n, h, w, device=dev)
            ops.level_masks(rounded[0, 0].to(dev, torch.int64).contiguous(), other[0, 0].to(dev, torch.int64).contiguous(),
                            f32(interp_w[0, 0]), mask_u8, n, E, Wt)
        else:
            maskf = mask_u8.float()
        out = []
        for i, (H, W) in enumerate(shapes):
            if not self.use_depth_scaling and i != n - 1:      # only the last level carries the view (:253-254)
                out.append((torch.zeros(1, 1, H, W, device=dev), None))
                continue
            M = torch.empty(H, W, device=dev)
            want_pw = self.use_angle_weight or self.use_depth_scaling
            pw = torch.empty(H, W, device=dev) if want_pw else None
            passed = torch.empty(H, W, dtype=torch.uint8, device=dev)
            ops.level_maps(E[i] if self.use_depth_scaling else maskf, Wt[i] if self.use_depth_scaling else None,
                           ag if self.use_angle_weight else None, adeg, float(self.angle_threshold), h, w, H, W, M, pw,
                           passed, torch.zeros(1, device=dev))
            out.append((M[None, None], None if pw is None else pw[None, None]))
        return out

    def forward_with_loss(self, batch, batch_idx, state):
        """The autograd formulation of a step (what reference model/model.py:178-327 computes), for callers that want
        ``loss.backward()`` through the differentiable classes: texture sampling (``texture.forward``: K1 / K2 under an
        autograd.Function), ONE gradient hook per predicted image carrying the product's per-view pixel weight, the
        ``ContentAndStyleLoss`` module over the level masks. Masks and weights are the fused engine's per-view constants
        (``_level_constants``), not a restatement of the reference's erode / interpolate chain. The Trainer path
        (``training_step``) does not come through here."""
        log_idx = batch_idx + self.current_epoch * self.batches_per_epoch[state]
        self.update_batch_count(batch_idx, state)
        preds = self.forward(batch)
        consts = self._level_constants(batch, [tuple(p.shape[2:]) for p in preds])
        live = []
        for p, (M, pw) in zip(preds, consts):
            if pw is not None and p.requires_grad:
                p.register_hook(lambda g, pw=pw: g * pw)
            if bool(M.any()):                                   # the empty-level filter (:256-257)
                live.append((p, M))
        style_loss, content_loss, _ = self.vgg_loss([p for p, _ in live], batch[0], [M for _, M in live],
                                                    angle_unnormalized=batch[12])
        losses = {"content": self.loss_weights["content"] * content_loss,
                  "style": self.loss_weights["style"] * style_loss}
        reg_on = self.loss_weights["tex_reg"] > 0
        losses["tex_reg"] = self.loss_weights["tex_reg"] * self.tex_reg_loss() if reg_on else torch.zeros_like(losses["content"])
        losses["total"] = losses["content"] + losses["style"] + losses["tex_reg"]
        self._log_losses(losses, state, log_idx)
        return {"loss": losses["total"]}

    def _log_losses(self, losses, state, log_idx):
        for loss_type, loss in losses.items():
            if loss_type in self.loss_history:
                self.loss_history[loss_type][state].append(loss.detach())   # device tensor: no host sync per step
                self.logger.experiment.add_scalar(f"Batch/Loss/{state}/{loss_type}", loss.detach(), log_idx)

    def prepare_view(self, batch, ready_event=None):
        """Loader hook (``MiniTrainer``): the NEXT view's device batch is resident - let the engine compute its per-view
        constants on a side stream during the current view's steps (``StepEngine.prepare_view``)."""
        if self._engine is not None and self.grad_reducer is None:
            # (deferred to the engine's next ``begin_step``: with index_repeat 1 the loader knows the next view before the
            # engine has made the current one current)
            self._engine.request_prepare(batch, ready_event)

    # ------------------------------------------------------------------ Lightning hooks
    def training_step(self, batch, batch_idx, optimizer_idx=0):
        eng = self._ensure_engine(batch[0].device)
        log_idx = batch_idx + self.current_epoch * self.batches_per_epoch["train"]
        self.update_batch_count(batch_idx, "train")
        for p, g in zip(self._texture_params(), eng.grads):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():   # a foreign optimizer dropped / replaced .grad
                eng.arena.g.zero_()
                p.grad = g
        # the engine's own step up to the optimizer (set_view on a new view key, per-view collective by schedule position,
        # step head, split update's early half, forward + backward, gradient exchange): the same launches ``bench.py``
        # times - ``FusedTextureAdam.step`` (the Trainer's ``optimizer.step()``) closes the step
        opt = getattr(self, "_fused_optimizer", None)
        if opt is not None:   # the scheduler owns the learning rate: the split update's early half needs THIS step's
            eng.cfg.learning_rate = opt.param_groups[0]["lr"]
            eng.cfg.decay_gamma, eng.epoch = 1.0, 0
        losses = dict(eng.step_compute(batch, self.grad_reducer))   # device tensors that stay valid: no copies
        losses["total"] = losses["content"] + losses["style"] + losses["tex_reg"]
        self._log_losses(losses, "train", log_idx)
        # The data-term gradient is already in ``param.grad``: a Trainer's ``loss.backward()`` has nothing left to do. A
        # caller that knows this (``MiniTrainer``: ``backward_done``) skips it - through the autograd engine the no-op
        # costs 0.66 ms of host time per step (a device-thread hand-off + a ones_like launch), more than half of a
        # single-level step; a stock Lightning loop gets a leaf that requires grad and may call backward() on it.
        if getattr(self, "fused_backward_done", False):
            return {"loss": losses["total"].detach(), "backward_done": True}
        return {"loss": losses["total"].detach().requires_grad_()}   # backward() of this scalar is a no-op

    def validation_step(self, batch, batch_idx):
        eng = self._ensure_engine(batch[0].device)
        log_idx = batch_idx + self.current_epoch * self.batches_per_epoch["val"]
        self.update_batch_count(batch_idx, "val")
        eng.set_view(batch)
        lt = eng.loss_tensors()
        eng.forward_backward(accumulate_grad=False)   # losses only: the texture scatter is skipped
        eng.view_key = None
        losses = {k: v.clone() for k, v in lt.items()}
        losses["total"] = losses["content"] + losses["style"] + losses["tex_reg"]
        self._log_losses(losses, "val", log_idx)
        return {"loss": losses["total"]}

    def reset_loss_count(self, state):
        for loss_type in self.loss_history.keys():
            self.loss_history[loss_type][state].clear()

    def compute_mean_loss(self, state):
        for loss_type, loss in self.loss_history.items():
            if isinstance(state, list):
                mean_loss = {s: torch.stack(loss[s]).mean().cpu() for s in state if loss[s]}
                self.logger.experiment.add_scalars(f"Loss/{'-'.join(state)}/{loss_type}", mean_loss, self.current_epoch)
            elif loss[state]:
                self.logger.experiment.add_scalar(f"Loss/{state}/{loss_type}", torch.stack(loss[state]).mean().cpu(),
                                                  self.current_epoch)

    def on_train_epoch_start(self) -> None:
        self.train_epoch_end = False
        self.val_epoch_end = False
        self.reset_loss_count("train")

    def on_validation_epoch_start(self) -> None:
        self.val_epoch_end = False
        self.reset_loss_count("val")

    def on_train_epoch_end(self) -> None:
        self.train_epoch_end = True

    def on_validation_epoch_end(self) -> None:
        self.val_epoch_end = True

    def on_epoch_end(self) -> None:
        if not self.train_epoch_end or not self.val_epoch_end:
            return
        self.compute_mean_loss("train")
        self.compute_mean_loss("val")
        self.compute_mean_loss(["train", "val"])
        if self.save_texture:
            with torch.no_grad():
                self.texture.save_layers(self.texture_dir, f"{self.texture_prefix}epoch_{self.current_epoch}",
                                         normalize_transform=post())
                self.texture.save_image(self.texture_dir, f"{self.texture_prefix}epoch_{self.current_epoch}_",
                                        normalize_transform=post())

    def configure_optimizers(self):
        optimizer = self._fused_optimizer = FusedTextureAdam(self, self.learning_rate)
        scheduler = StepLR(optimizer, gamma=self.decay_gamma, step_size=self.decay_step_size)
        return [optimizer], [scheduler]


def to_tensor_image(t, idx=0):
    if len(t.shape) == 4:
        return torch.stack([to_tensor_image(t[b], idx) for b in range(t.shape[0])], dim=0)
    import numpy as np
    img = to_image(t, idx, normalize_transform=post())
    return torch.from_numpy(np.asarray(img)).permute(2, 0, 1).float() / 255


def find_pyramid_size(pyramid, sample):
    for i, p in enumerate(pyramid):
        if p.shape[2] == sample.shape[2]:
            return i, p
    return 0, p[0]
