"""Texture scatter of a c3 view: tiled atomic kernel (4 launches) vs the planned sorted gather (1 launch) + the
per-view plan cost. Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from stylemesh_amd.runtime import ops
from stylemesh_amd.runtime.fmap import FMap
from stylemesh_amd.runtime.engine import TextureArena

wl = bench.WORKLOADS["c3"]
view = bench.to_device(bench.make_views(wl, [0])[0], "cuda")
uv = view[9]
grids = [u[0].contiguous() for u in uv]
arena = TextureArena(4096, 4096, 4, "cuda")
grads = arena.views(arena.g)
gimgs, pws = [], []
for g in grids:
    h, w = g.shape[:2]
    gimgs.append(FMap(3, h, w).from_dense(torch.randn(3, h, w, device="cuda")))
    yy, xx = torch.meshgrid(torch.arange(h, device="cuda"), torch.arange(w, device="cuda"), indexing="ij")
    pws.append(((yy > 0.15 * h) & (xx < 0.8 * w)).float() * torch.rand(h, w, device="cuda"))

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

def atomic():
    for g, gi, p in zip(grids, gimgs, pws): ops.tex_sample_bwd(grads, g, gi, p)
plan = ops.ScatterPlan(grads, arena.g)
print(f"tiled atomic scatter, 4 levels: {timeit(atomic):8.1f} us")
print(f"plan build (entries + radix sort of {4 * 4 * sum(g.shape[0] * g.shape[1] for g in grids) / 1e6:.1f} M entries): {timeit(lambda: plan.build(grids, pws), 5):8.1f} us")
print(f"planned sorted gather, 1 launch: {timeit(lambda: plan.scatter(gimgs)):8.1f} us (accumulate) "
      f"{timeit(lambda: plan.scatter(gimgs, accumulate=False)):8.1f} us (arena known zero)")
arena.g.zero_(); atomic(); a = arena.g.clone(); arena.g.zero_(); plan.scatter(gimgs, accumulate=False)
print("max |diff| / max |ref|:", float((arena.g - a).abs().max() / a.abs().max()))
k = plan.bufs[plan.sorted_in].view(torch.int32)
ncross = int(plan.bufs[5][:4].view(torch.int32)[0])
valid = k != ((1 << plan.key_bits) - 1)
kv = k[valid]
runs = int((kv[1:] != kv[:-1]).sum()) + 1
print(f"valid entries {int(valid.sum())/1e6:.1f} M, runs (touched texels) {runs/1e6:.2f} M, mean run length {int(valid.sum())/runs:.1f}, runs crossing a 64-entry chunk boundary {ncross/1e6:.2f} M")
