"""How much of the conv work do the 1-D position tiles (128 / 256 consecutive positions) waste against the exact need
maps, and what would 2-D tiles (8 x 16, 16 x 16) cover? c3 bench views, weighted by each layer's FLOPs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from stylemesh_amd.runtime import ops, hip
from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
from stylemesh_amd.runtime.sparsity import need_maps
from stylemesh_amd.runtime.vgg import NODES, depth_of

wl = B.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
dev = torch.device("cuda")
cfg = EngineConfig(tex_w=wl["tex"], tex_h=wl["tex"], hierarchical=True, n_layers=4, style_weights=B.STYLE_WEIGHTS,
                   angle_threshold=wl["thr"], style_pyramid_mode=wl["mode"], use_angle_weight=wl["angle"],
                   use_depth_scaling=wl["depth"], loss_weights=dict(B.LOSS_WEIGHTS), learning_rate=1.0, decay_step_size=3)
eng = StepEngine(cfg, B.S.seeded_vgg_state(0), device=dev)
eng.set_style_image(B.S.style_image(1, *B.STYLE_HW))
tot = {"dense": 0.0, "need": 0.0, "1d": 0.0, "8x16": 0.0, "16x16": 0.0, "4x32": 0.0}
for seed in (0, 2, 6):
    view = B.to_device(B.make_views(wl, [seed])[0], dev)
    eng.set_view(view)
    for lv in eng.view:
        if not lv.active:
            continue
        need = need_maps(lv.M, lv.H, lv.W, set(eng.injected), eng.deepest)
        for kind, src, dst, cin, cout in NODES[:depth_of(eng.deepest) + 1]:
            if kind == "pool":
                continue
            for direction, layer, ci, co in (("f", dst, cin, cout), ("b", src, cout, cin)):
                if layer == "img":
                    continue
                nd = need[layer]
                h, w = nd.shape
                fl = 2.0 * 9 * ci * co                       # per position
                tot["dense"] += fl * h * w
                tot["need"] += fl * float(nd.sum())
                bn = 256 if co % 128 else 128
                wp = hip.row_stride(w)
                padded = torch.zeros(h, wp, device=dev); padded[:, 1:w + 1] = nd
                flat = padded.flatten()
                n_t = (flat.numel() + bn - 1) // bn
                flat = torch.nn.functional.pad(flat, (0, n_t * bn - flat.numel()))
                tot["1d"] += fl * bn * float((flat.view(n_t, bn).sum(1) > 0).sum())
                for sub in (64, 32, 16):   # dead sub-ranges inside live tiles skipped
                    tot[f"1d/{sub}"] = tot.get(f"1d/{sub}", 0.0) + fl * sub * float((flat.view(-1, sub).sum(1) > 0).sum())
                for name, (th, tw) in (("8x16", (8, 16)), ("16x16", (16, 16)), ("4x32", (4, 32))):
                    hh, ww = -(-h // th) * th, -(-w // tw) * tw
                    p2 = torch.zeros(hh, ww, device=dev); p2[:h, :w] = nd
                    t = p2.view(hh // th, th, ww // tw, tw).sum((1, 3)) > 0
                    tot[name] += fl * th * tw * float(t.sum())
for k, v in tot.items():
    print(f"{k:6s} {v / 3e9:9.1f} GFLOP per view   {v / tot['dense']:.3f} of dense")
