"""Cost of the fused update as a function of the flagged (touched) share of the arena (GPU box). 66.8 M texel floats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylemesh_amd.runtime import ops
n = 66846720
p, g, m, v = (torch.zeros(n, device="cuda") for _ in range(4))
seg = [n // 64 * 1, n // 64 * 5, n // 64 * 21, n]
reg = [1e-4] * 4
ssq = torch.zeros(4, device="cuda")
def timed(fn, k=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / k
nch = -(-n // 64)
for name, flags in [("dense (no flags)", None)] + [(f"{f:4.0%} in runs of {run} chunks", None) for f in ()]:
    print(f"{name:34s} {timed(lambda: ops.adam_fused(p, g, m, v, seg, reg, 1.0, 3, sumsq_out=ssq)):7.1f} us")
for frac in (1.0, 0.4, 0.16, 0.05, 0.0):
    for run in (1, 16, 1024):
        blocks = torch.rand(-(-nch // run), device="cuda") < frac
        fl = blocks.repeat_interleave(run)[:nch].to(torch.int32).contiguous()
        t = timed(lambda: ops.adam_fused(p, g, m, v, seg, reg, 1.0, 3, sumsq_out=ssq, touched=fl, touched_log2=6))
        print(f"{frac:4.0%} flagged, runs of {run:4d} chunks: {t:7.1f} us   ({float(fl.float().mean()) * n * 28 / t / 1e6:5.2f} TB/s of flagged traffic)")

# the flags of a REAL view (the bench's c3 views): the closing update walks the view's own chunks, the early half the rest
if os.environ.get("REAL_FLAGS", "1") == "1":
    import bench as B
    from stylemesh_amd.data import synthetic as S
    from stylemesh_amd.runtime.engine import StepEngine
    wl = B.WORKLOADS["c3"]
    eng = StepEngine(B.engine_config(wl), S.seeded_vgg_state(0))
    eng.set_style_image(S.style_image(1, *B.STYLE_HW))
    views = [B.to_device(v, "cuda") for v in B.make_views(wl, [0, 2, 6])]
    for v in views:
        eng.training_step(v)
    a = eng.arena
    for name, fl in (("view's own chunks (closing update)", eng._view_flags), ("ever-touched minus the view (early half)", eng._other_flags[1] if eng._other_flags else None)):
        if fl is None:
            continue
        share = float((fl != 0).float().mean())
        t = timed(lambda: ops.adam_fused(a.p, a.g, a.m, a.v, a.seg_end, eng.reg_coef, 1.0, 3, sumsq_out=eng.sumsq, touched=fl, touched_log2=6))
        print(f"real c3 view, {name}: {share:5.1%} flagged: {t:7.1f} us ({share * a.n * 28 / t / 1e6:5.2f} TB/s of flagged traffic)")
