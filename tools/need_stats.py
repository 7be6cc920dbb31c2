"""Exact need fraction vs active-tile fraction per layer for the c3 bench view (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from stylemesh_amd.runtime import ops, sparsity, engine as E

orig = sparsity.build_tile_lists
def patched(needs, last_layer):
    out = orig(needs, last_layer)
    from stylemesh_amd.runtime import hip
    for layer in needs[0]:
        if layer == "img": continue
        tot = act = need = 0
        for nd in needs:
            m = nd[layer]
            h, w = m.shape
            tot += h * w; need += float(m.sum())
        print(f"{layer:4s} need {need/tot:.3f}", end="")
        for key, (lst, frac, n_all) in out.items():
            pass
        print()
    for key, (lst, frac, n_all) in out.items():
        print(key, f"active tile fraction {frac:.3f}")
    sparsity.build_tile_lists = orig
    return out
sparsity.build_tile_lists = patched
E.sparsity = sparsity
sys.argv = [sys.argv[0], "--steps", "2", "--warmup", "1", "--cpu-steps", "0", "--no-conv-timer"]
bench.main()
