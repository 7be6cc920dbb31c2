"""Two engines of one process in lock-step from a zero texture (sparse ever-touched update), one rank: how often does a step
of engine a differ from the same step of engine b? usage: stress_lockstep.py [steps=36] [repeat=3]"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from golden_cases import MULTIVIEW_SEEDS
from stepcmp import lock, step_deviation
from test_round3_gpu import _engine, _small_view
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 36
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = _engine(), _engine()
from stylemesh_amd.runtime import ops as _ops
_orig_sample = _ops.tex_sample_fwd_grouped
snap = {}
def _snap_sample(layers, grids, outs):
    eng = a if layers[0].data_ptr() == a.layers[0].data_ptr() else b
    p0 = eng.arena.p.clone()
    snap[id(eng)] = (p0, [g.clone() for g in grids])
    runs = []
    for r in range(4):
        _orig_sample(layers, grids, outs)
        runs.append([o.planes.clone() for o in outs])
    snap[id(eng)] += (runs[0],)
    same = [all(torch.equal(x, y) for x, y in zip(runs[0], runs[r])) for r in range(1, 4)]
    if not all(same):
        ne = [(x != y) for x, y in zip(runs[0], runs[[i for i, e in enumerate(same) if not e][0] + 1])]
        print(f"   [{'a' if eng is a else 'b'}] 4 launches of the sampler on unchanged inputs, equal to the first: {same}; differing elements per level "
              f"{[int(d.sum()) for d in ne]}, per output plane {[[int(d[c].sum()) for c in range(3)] for d in ne]}")
if os.environ.get('DIAG', '0') == '1':
    _ops.tex_sample_fwd_grouped = _snap_sample
views = [_small_view(s) for s in MULTIVIEW_SEEDS]
devs = []
import fcntl
_lockf = open("/tmp/gpu_turn.lock", "w") if os.environ.get("GPU_TURNS", "0") == "1" else None
for k in range(steps):
    v = views[(k // rep) % len(views)]
    if _lockf:
        fcntl.flock(_lockf, fcntl.LOCK_EX)
    m0, v0 = lock(a, b)
    if os.environ.get("SYNC_EACH", "0") == "1":
        torch.cuda.synchronize()
    ls = []
    for e in (b, a):
        ls.append(e.training_step(v))
        if os.environ.get("SYNC_EACH", "0") == "1":
            torch.cuda.synchronize()
    d = step_deviation(a, b, m0, v0)
    devs.append(d)
    if _lockf:
        torch.cuda.synchronize()
        fcntl.flock(_lockf, fcntl.LOCK_UN)
    if os.environ.get("DIAG", "0") == "1" and float(d[0]) > 1e-6 * float(d[1]):          # (a host sync per step: diagnosis mode)
        ga = (a.arena.m - 0.9 * m0) / 0.1
        gb = (b.arena.m - 0.9 * m0) / 0.1
        diff = (ga - gb).abs()
        sel = diff > 1e-4 * gb.abs().clamp_min(1e-12)
        sel &= gb.abs() > 1e-3 * gb.abs().max()
        ratio = (ga[sel] / gb[sel])
        seg = [0] + list(a.arena.seg_end)
        per_layer = [int(sel[seg[i]:seg[i + 1]].sum()) for i in range(len(seg) - 1)]
        la = {k_: float(x) for k_, x in (ls[1] or {}).items()} if isinstance(ls[1], dict) else ls[1]
        lb = {k_: float(x) for k_, x in (ls[0] or {}).items()} if isinstance(ls[0], dict) else ls[0]
        from stylemesh_amd.runtime.vgg import AmaxBook
        W = AmaxBook.W
        names = list(a.amax.idx)
        va = a.amax.buf.view(-1, W).max(1).values.cpu()
        vb = b.amax.buf.view(-1, W).max(1).values.cpu()
        dif = [(names[i], float(va[i]), float(vb[i])) for i in range(len(names)) if float(va[i]) != float(vb[i])]
        print("   amax bounds that differ (name, a, b):", dif[:12], "of", len(names))
        if not snap:
            continue
        pa, ga_, oa = snap[id(a)]; pb, gb_, ob = snap[id(b)]
        dp_ = (pa != pb)
        print(f"   at sampling time: p differs at {int(dp_.sum())} elements" + (f" (first index {int(dp_.nonzero()[0])}, chunk {int(dp_.nonzero()[0]) // 64}, values {float(pa[dp_][0]):.4f} vs {float(pb[dp_][0]):.4f})" if dp_.any() else "")
              + f"; grids equal {all(torch.equal(x, y) for x, y in zip(ga_, gb_))}; sampled images equal right after the launch {all(torch.equal(x, y) for x, y in zip(oa, ob))}")
        for hw, ba in a._bufs.items():
            bb = b._bufs[hw]
            for name, fa in ba.act.items():
                fbm = bb.act[name]
                ne = fa.planes != fbm.planes
                if bool(ne.any()):
                    Wp = fa.Wp
                    pos = ne.any(0).nonzero().flatten()
                    rows = (pos // Wp - 1)
                    print(f"   level {hw} first differing activation {name}: {int(ne.sum())} elements at {pos.numel()} positions, rows {int(rows.min())}..{int(rows.max())} "
                          f"cols {int((pos % Wp - 1).min())}..{int((pos % Wp - 1).max())} max|d| {float((fa.planes - fbm.planes).abs().max()):.4g} max|x| {float(fbm.planes.abs().max()):.4g}")
                    break
        print(f"BAD step {k} view {(k // rep) % len(views)}: texels {int(sel.sum())} per layer {per_layer} ratio ga/gb quantiles "
              f"{[round(float(q), 4) for q in torch.quantile(ratio.float()[:1000000], torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95], device=ratio.device))] if ratio.numel() else None}"
              f" max|ga-gb|/max|gb| {float(diff.max() / gb.abs().max()):.4f}\n   losses a {la}\n   losses b {lb}")
torch.cuda.synchronize()
L = torch.stack(devs).cpu()
bad = [(k, round(float(d[0] / (0.1 * d[1])), 6), round(float(d[4]), 4)) for k, d in enumerate(L) if float(d[0]) > 0.1 * 1e-5 * float(d[1]) or float(d[2]) != 0]
print("steps", steps, "bad", len(bad), bad[:8], "touched" , a.touched is not None)
