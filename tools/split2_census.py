"""Dynamic range of the fp16x2 operands LATE in the schedule -> profiles/r06/split2_dynamic_range.json.

Trains the c3 workload (ScanNet with_angle_and_depth, 4096^2 hier-4 texture, 4 UV levels) from the zero texture for the
script's schedule shape - 7 epochs x index_repeat 20 over --views synthetic views, StepLR(3, 0.1): lr 1 / 0.1 / 0.01
(scripts/train/optimize_texture_scannet_with_angle_and_depth.sh:11-15, model/model.py:387-401) - and takes the census of
stylemesh_amd/diagnostics.py (histogram of log2(bound / |x|) per operand tensor, share beyond 2^18) at three points: after
epoch 0 (lr 1), after epoch 3 (behind the first decay) and at the end, each on a trained view and on a view the
training never saw.  Usage (GPU box): python tools/split2_census.py [--views 48] [--out gpurun_out/split2_dynamic_range.json]"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from stylemesh_amd.data import synthetic as S  # noqa: E402
from stylemesh_amd.diagnostics import operand_census, summarize  # noqa: E402
from stylemesh_amd.runtime.engine import EngineConfig, StepEngine  # noqa: E402

GOOD = (0, 2, 6, 7, 9, 11, 12, 14, 16, 18, 22, 23, 26, 27, 29, 30, 32, 33, 35, 36, 37, 38, 39)   # all four levels populated


def view(seed):
    return S.make_view(seed, view_hw=S.SCANNET_VIEW_HW, level_hw=S.SCANNET_LEVEL_HW,
                       level_heights=[h for h, _ in S.SCANNET_LEVEL_HW], min_pyramid_depth=0.25, room=S.BoxRoom((12.0, 9.0, 3.0)))


def census_at(eng, v, tag):
    """census of one dense step on view ``v`` WITHOUT disturbing the training state"""
    state = [t.clone() for t in (eng.arena.g,)]
    sparse = eng.sparse_tiles
    eng.sparse_tiles = False
    eng.set_view(v)
    eng.arena.g.zero_()
    eng.forward_backward()
    c = operand_census(eng)
    eng.arena.g.copy_(state[0])
    eng.sparse_tiles = sparse
    eng.view_key = None          # the next training step prepares its view again (sparse lists)
    return {"at": tag, "summary": summarize(c), "tensors": c}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=48)
    ap.add_argument("--epochs", type=int, default=7)
    ap.add_argument("--repeat", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "split2_dynamic_range.json"))
    args = ap.parse_args()
    cfg = EngineConfig(tex_w=4096, tex_h=4096, hierarchical=True, n_layers=4, style_weights=[1000., 1000., 10., 10., 1000.],
                       angle_threshold=30.0, style_pyramid_mode="multi", use_angle_weight=True, use_depth_scaling=True,
                       loss_weights={"content": 7e1, "style": 1e-4, "tex_reg": 5e3}, learning_rate=1.0, decay_step_size=3)
    eng = StepEngine(cfg, S.seeded_vgg_state(0))
    eng.set_style_image(S.style_image(1, 1528, 1200))
    seeds = [GOOD[i % len(GOOD)] + 40 * (i // len(GOOD)) for i in range(args.views)]
    views = [view(s) for s in seeds]
    unseen = view(1000)
    out = {"workload": "c3", "views": args.views, "epochs": args.epochs, "index_repeat": args.repeat, "points": []}
    t0 = time.time()
    for epoch in range(args.epochs):
        for v in views:
            for _ in range(args.repeat):
                eng.training_step(v)
        eng.end_epoch()
        if epoch in (0, 3, args.epochs - 1):
            tag = f"after epoch {epoch} ({eng.step_count} steps, next lr {eng.lr:g})"
            for name, v in (("trained view", views[len(views) // 2]), ("unseen view", unseen)):
                p = census_at(eng, v, f"{tag}, {name}")
                out["points"].append(p)
                print(p["at"], json.dumps(p["summary"]), flush=True)
    torch.cuda.synchronize()
    out["seconds"] = round(time.time() - t0, 1)
    out["steps"] = eng.step_count
    out["worst_share_beyond_2^18_over_all_points"] = max(p["summary"]["worst_share_beyond_2^18"] for p in out["points"])
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print("written", args.out, "worst share beyond 2^18:", out["worst_share_beyond_2^18_over_all_points"])


if __name__ == "__main__":
    main()
