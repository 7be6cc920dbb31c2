"""``python -m stylemesh_amd.model.optimize`` - the reference's training harness (model/optimize.py:28-165,237-290)
for the MI355X path: same flag names / defaults / parsing (``--loss_weight k=v``, ``--tex_reg_weight i=v``,
``--texture_size W,H`` ...), same construction order (DataModule -> style image -> pipeline -> ``trainer.fit``).

Differences, all outside the hot path: ``--dataset synthetic`` (seeded box-room views; ``--dataset scannet`` reads
and ``--dataset matterport`` read
the reference's on-disk scene / region layouts), ``--vgg_gatys_model_path random:<seed>`` for seeded He-normal weights
when ``vgg_conv.pth`` is unavailable, ``--style_image_path synthetic:<seed>:<H>x<W>``, and no post-hoc mip-map
render / video / evaluation (reference :167-234, out of scope). ``--num_workers`` > 0 (the reference's DataLoader
worker count, data/abstract_dataset.py:480) switches the loaders' background view prefetch on (one thread is enough
here: a view is decoded during the previous view's ``index_repeat`` steps). Multi-GPU: launch under
``torch.distributed.run``; views shard over ranks and the texture gradient is all-reduced over RCCL.
"""
from __future__ import annotations

import os
import tempfile
from argparse import ArgumentParser

import torch

from ..data import synthetic as S
from ..data.datamodule import SyntheticSceneDataModule
from ..trainer import JsonlLogger, MiniTrainer
from .losses.content_and_style_losses import ContentAndStyleLoss
from .losses.rgb_transform import pre
from .model import TextureOptimizationStyleTransferPipeline


def load_style_image(path: str) -> torch.Tensor:
    if path.startswith("synthetic:"):
        _, seed, hw = path.split(":")
        h, w = (int(v) for v in hw.split("x"))
        return S.style_image(int(seed), h, w)
    import numpy as np
    import PIL
    from PIL import Image
    PIL.Image.MAX_IMAGE_PIXELS = 933120000
    img = Image.open(path).convert("RGB")
    if img.size[0] > 2048 or img.size[1] > 2048:   # Resize(2048): shorter side -> 2048 (reference :122-123)
        w, h = img.size
        s = 2048 / min(w, h)
        img = img.resize((int(round(w * s)), int(round(h * s))), Image.BILINEAR)
    t = torch.from_numpy(np.asarray(img)).permute(2, 0, 1).float() / 255
    return pre()(t)


def resolve_vgg_path(path: str) -> str:
    if path.startswith("random:"):
        f = tempfile.NamedTemporaryFile(suffix=".pth", delete=False)
        torch.save(S.seeded_vgg_state(int(path.split(":")[1])), f.name)
        return f.name
    return path


def main(args):
    import time as _time
    _t0 = _time.time()
    _timing = os.environ.get("STYLEMESH_MAIN_TIMING") == "1"

    def _mark(what):
        if _timing:
            print(f"[main +{_time.time() - _t0:6.2f} s] {what}", flush=True)
    from ..runtime.hostcpu import limit_host_threads
    limit_host_threads()      # (the visible core count is not what the container may use: see runtime/hostcpu.py)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
    logger = JsonlLogger(save_dir=args.default_root_dir, version=args.version, rank=rank)
    log_dir = logger.log_dir

    if args.dataset == "synthetic":
        heights = S.SCANNET_LEVEL_HW[:args.pyramid_levels]
        dm = SyntheticSceneDataModule(n_views=args.max_images if args.max_images > 0 else 8,
                                      view_hw=S.SCANNET_VIEW_HW, level_hw=heights,
                                      min_pyramid_depth=args.min_pyramid_depth, split=(args.train_split, args.val_split),
                                      index_repeat=args.index_repeat, sampler_mode=args.sampler_mode, rank=rank,
                                      world_size=world, prefetch=2 if args.num_workers > 0 else 0)
    elif args.dataset == "scannet":
        from ..data.scannet import ScanNetSingleSceneDataModule
        dm = ScanNetSingleSceneDataModule(args.root_path, args.scene, resize_size=args.resize_size,
                                          pyramid_levels=args.pyramid_levels, min_pyramid_depth=args.min_pyramid_depth,
                                          min_pyramid_height=args.min_pyramid_height, max_images=args.max_images,
                                          split=(args.train_split, args.val_split), index_repeat=args.index_repeat,
                                          sampler_mode=args.sampler_mode, rank=rank, world_size=world,
                                          prefetch=2 if args.num_workers > 0 else 0,
                                          decode_workers=max(1, min(args.num_workers, 4)))
    elif args.dataset == "matterport":
        from ..data.matterport import MatterportSingleRegionDataModule
        dm = MatterportSingleRegionDataModule(args.root_path, args.scene, region_index=args.matterport_region_index,
                                              resize_size=args.resize_size, pyramid_levels=args.pyramid_levels,
                                              min_pyramid_depth=args.min_pyramid_depth,
                                              min_pyramid_height=args.min_pyramid_height, max_images=args.max_images,
                                              split=(args.train_split, args.val_split), index_repeat=args.index_repeat,
                                              sampler_mode=args.sampler_mode, rank=rank, world_size=world,
                                              prefetch=2 if args.num_workers > 0 else 0,
                                              decode_workers=max(1, min(args.num_workers, 4)))
    else:
        raise ValueError(f"Unsupported dataset: {args.dataset}")
    _mark("datamodule built")
    dm.prepare_data()
    dm.setup()
    _mark("datamodule set up")
    if hasattr(dm, "warm_start"):
        dm.warm_start()       # (the decode processes import their modules while the model is being built)
    _mark("decode processes started")

    if args.loss_weights:
        args.loss_weights = {l[0]: float(l[1]) for l in args.loss_weights}
    if args.tex_reg_weights:
        w = {int(w[0]): float(w[1]) for w in args.tex_reg_weights}
        args.tex_reg_weights = [w[i] for i in range(len(w))]

    style_image = load_style_image(args.style_image_path)
    model = TextureOptimizationStyleTransferPipeline(
        W=args.texture_size[0], H=args.texture_size[1], hierarchical_texture=args.hierarchical,
        hierarchical_layers=args.hierarchical_layers, random_texture_init=args.random_texture_init,
        style_image=style_image, style_layers=args.style_layers, content_layers=args.content_layers,
        style_weights=args.style_weights, content_weights=args.content_weights,
        vgg_gatys_model_path=resolve_vgg_path(args.vgg_gatys_model_path), use_angle_weight=not args.no_angle_weight,
        use_depth_scaling=not args.no_depth_scaling, angle_threshold=args.angle_threshold,
        style_pyramid_mode=args.style_pyramid_mode, gram_mode=args.gram_mode, learning_rate=args.learning_rate,
        tex_reg_weights=args.tex_reg_weights, decay_gamma=args.decay_gamma, decay_step_size=args.decay_step_size,
        loss_weights=args.loss_weights, extra_args=vars(args), log_images_nth=args.log_images_nth,
        save_texture=args.save_texture and rank == 0, texture_dir=log_dir)
    from ..runtime.distributed import make_comm, make_sparse_grad_reducer
    comm = make_comm(dist, rank, world, device) if world > 1 else None   # the product's own RCCL communicator
    if os.environ.get("STYLEMESH_DEFERRED_EXCHANGE", "0") == "1" and world > 1:
        # owner-aware reducer + a second communicator for its background exchange (runtime/distributed.py)
        model.grad_reducer = make_sparse_grad_reducer(comm, world, rank=rank, deferred_dist=make_comm(dist, rank, world, device))
    else:
        model.grad_reducer = make_sparse_grad_reducer(comm, world)

    trainer = MiniTrainer(max_epochs=args.max_epochs, logger=logger, device=device, rank=rank, world_size=world)
    _mark("model built")
    trainer.fit(model, dm)
    _mark("fit done")
    if world > 1:
        if hasattr(comm, "destroy"):
            comm.destroy()
        dist.destroy_process_group()
    return model


def build_parser():
    parser = ArgumentParser()
    # the Trainer flags the reference's scripts use (scripts/train/*.sh:1,14)
    parser.add_argument('--gpus', default=1, type=int)
    parser.add_argument('--max_epochs', default=1, type=int)
    parser.add_argument('--default_root_dir', default=".", type=str)
    parser.add_argument('--version', default=0, type=int)
    # custom flags: names, defaults and parsing of model/optimize.py:244-290
    parser.add_argument('--root_path', default="/path/to/datasets/scannet")
    parser.add_argument('--dataset', default="scannet", choices=["icl", "scannet", "vase", "3dfuture", "matterport", "synthetic"])
    parser.add_argument('--matterport_region_index', default=0, type=int)
    parser.add_argument('--train_split', default=0.8, type=float)
    parser.add_argument('--val_split', default=0.2, type=float)
    parser.add_argument('--split_mode', default="sequential", type=str, choices=SyntheticSceneDataModule.split_modes)
    parser.add_argument('--scene', default="")
    parser.add_argument('--max_images', default=-1, type=int)
    parser.add_argument('--min_images', default=1000, type=int)
    parser.add_argument('--resize_size', default=256, type=int)
    parser.add_argument('--texture_size', default="512,512", type=lambda s: [int(f) for f in s.split(",")], dest='texture_size')
    parser.add_argument('--hierarchical', default=False, action="store_true")
    parser.add_argument('--hierarchical_layers', default=4, type=int)
    parser.add_argument('--random_texture_init', default=False, action="store_true")
    parser.add_argument('--batch_size', default=1, type=int)
    parser.add_argument('--learning_rate', default=1, type=float)
    parser.add_argument("--loss_weight", action='append', type=lambda kv: kv.split("="), dest='loss_weights')
    parser.add_argument("--tex_reg_weight", action='append', type=lambda kv: kv.split("="), dest='tex_reg_weights')
    parser.add_argument('--decay_gamma', default=0.1, type=float)
    parser.add_argument('--decay_step_size', default=30, type=int)
    parser.add_argument('--num_workers', default=4, type=int)
    parser.add_argument('--log_images_nth', default=-1, type=int)
    parser.add_argument('--save_texture', default=False, action="store_true")
    parser.add_argument('--shuffle', default=False, action="store_true")
    parser.add_argument('--sampler_mode', default="repeat", type=str, choices=SyntheticSceneDataModule.sampler_modes)
    parser.add_argument('--index_repeat', default=1, type=int)
    parser.add_argument('--vgg_gatys_model_path', default="/path/to/models/vgg_conv.pth", type=str)
    parser.add_argument('--style_image_path', required=True, type=str)
    parser.add_argument('--style_layers', type=lambda s: [f for f in s.split(",")], dest='style_layers', default=ContentAndStyleLoss.style_layers)
    parser.add_argument('--content_layers', type=lambda s: [f for f in s.split(",")], dest='content_layers', default=ContentAndStyleLoss.content_layers)
    parser.add_argument('--style_weights', type=lambda s: [float(f) for f in s.split(",")], dest='style_weights', default=ContentAndStyleLoss.style_weights)
    parser.add_argument('--content_weights', type=lambda s: [float(f) for f in s.split(",")], dest='content_weights', default=ContentAndStyleLoss.content_weights)
    parser.add_argument('--no_angle_weight', default=False, action="store_true")
    parser.add_argument('--no_depth_scaling', default=False, action="store_true")
    parser.add_argument('--angle_threshold', default=60.0, required=False, type=float)
    parser.add_argument('--pyramid_levels', default=8, required=False, type=int)
    parser.add_argument('--min_pyramid_depth', default=0.25, required=False, type=float)
    parser.add_argument('--min_pyramid_height', default=32, required=False, type=int)
    parser.add_argument('--style_pyramid_mode', default='single', required=False, choices=ContentAndStyleLoss.style_pyramid_modes)
    parser.add_argument('--gram_mode', default='current', required=False, choices=ContentAndStyleLoss.gram_modes)
    parser.add_argument('--renderer_mipmap', default=None, required=False, type=str)
    return parser


if __name__ == '__main__':
    main(build_parser().parse_args())
