"""Per VGG layer: exact need fraction vs the fraction covered by 1-D segments of 128 / 32 / 16 / 8 positions (c3 bench views,
all four levels together, weighted by positions; forward lists). Where is the remaining dead conv work? (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from stylemesh_amd.runtime import hip
from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
from stylemesh_amd.runtime.sparsity import need_maps
from stylemesh_amd.runtime.vgg import NODES, depth_of

wl = B.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
seeds = [int(a) for a in sys.argv[2:]] or [0, 2, 6, 7, 9]
dev = torch.device("cuda")
cfg = EngineConfig(tex_w=wl["tex"], tex_h=wl["tex"], hierarchical=True, n_layers=4, style_weights=B.STYLE_WEIGHTS,
                   angle_threshold=wl["thr"], style_pyramid_mode=wl["mode"], use_angle_weight=wl["angle"],
                   use_depth_scaling=wl["depth"], loss_weights=dict(B.LOSS_WEIGHTS), learning_rate=1.0, decay_step_size=3)
eng = StepEngine(cfg, B.S.seeded_vgg_state(0), device=dev)
eng.set_style_image(B.S.style_image(1, *B.STYLE_HW))
acc = {}
for seed in seeds:
    eng.set_view(B.to_device(B.make_views(wl, [seed])[0], dev))
    for lv in eng.view:
        if not lv.active:
            continue
        need = need_maps(lv.M, lv.H, lv.W, set(eng.injected), eng.deepest)
        for kind, src, dst, cin, cout in NODES[:depth_of(eng.deepest) + 1]:
            if kind == "pool":
                continue
            nd = need[dst]
            h, w = nd.shape
            wp = hip.row_stride(w)
            padded = torch.zeros(h, wp, device=dev); padded[:, 1:w + 1] = nd
            flat = padded.flatten()
            a = acc.setdefault(kind, {"flops": 2.0 * 9 * cin * cout, "dense": 0.0, "need": 0.0, 128: 0.0, 32: 0.0, 16: 0.0, 8: 0.0})
            a["dense"] += h * w
            a["need"] += float(nd.sum())
            for g in (128, 32, 16, 8):
                f = torch.nn.functional.pad(flat, (0, (-flat.numel()) % g))
                a[g] += g * float((f.view(-1, g).sum(1) > 0).sum())
            # greedy cover by 32-position segments that may START anywhere on the 4-position grid (no overlaps)
            import numpy as np
            nz = np.flatnonzero(flat.cpu().numpy() > 0)
            cnt, cur, k = 0, 0, 0
            while k < len(nz):
                st = max(nz[k] // 4 * 4, cur)
                cur = st + 32
                cnt += 1
                k = np.searchsorted(nz, cur)
            a["free32"] = a.get("free32", 0.0) + 32.0 * cnt
tot = {k: 0.0 for k in ("dense", "need", 128, 32, 16, 8, "free32")}
print(f"{'layer':8s} {'need':>6s} {'seg128':>7s} {'seg32':>6s} {'seg16':>6s} {'seg8':>6s}   (fraction of the dense positions)")
for kind, a in acc.items():
    print(f"{kind:8s} {a['need']/a['dense']:6.3f} {a[128]/a['dense']:7.3f} {a[32]/a['dense']:6.3f} {a[16]/a['dense']:6.3f} {a[8]/a['dense']:6.3f}   free-start 32: {a['free32']/a['dense']:6.3f}")
    for k in tot:
        tot[k] += a["flops"] * a[k]
print(f"{'FLOP-weighted':8s} need {tot['need']/tot['dense']:.3f}  seg128 {tot[128]/tot['dense']:.3f}  seg32 {tot[32]/tot['dense']:.3f}  seg16 {tot[16]/tot['dense']:.3f}  seg8 {tot[8]/tot['dense']:.3f}  free-start 32 {tot['free32']/tot['dense']:.3f}")
