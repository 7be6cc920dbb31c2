import torch, time, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylemesh_amd.runtime.distributed import _pack_view, _unpack_view, _PinnedRing
from stylemesh_amd.data import synthetic as S
v = S.make_view(0, view_hw=S.SCANNET_VIEW_HW, level_hw=[S.SCANNET_VIEW_HW], level_heights=[256], min_pyramid_depth=0.25, room=S.BoxRoom((12.0, 9.0, 3.0)))
flat, meta = _pack_view(v)
ring = _PinnedRing(4)
torch.cuda.init()
dev = torch.device("cuda")
def upload(items):
    mv = lambda t: t.to(dev, non_blocking=True) if torch.is_tensor(t) else t
    return tuple(x if k == 8 else ([mv(u) for u in x] if isinstance(x, (list, tuple)) else mv(x)) for k, x in enumerate(items))
for name, src in (("pageable", v), ("pinned ring views", None), ("pin_memory per tensor", None)):
    for rep in range(4):
        if name == "pinned ring views":
            items = _unpack_view(ring.stage(flat), meta)
        elif name == "pin_memory per tensor":
            items = tuple([u.pin_memory() for u in x] if isinstance(x, list) else (x.pin_memory() if torch.is_tensor(x) else x) for x in v)
        else:
            items = v
        pinned = [t.is_pinned() for t in items if torch.is_tensor(t)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d = upload(items)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: host {1e3*(t1-t0):.2f} ms, until done {1e3*(t2-t0):.2f} ms, is_pinned {all(pinned)} ({sum(pinned)}/{len(pinned)}), bytes {flat.numel()}")
