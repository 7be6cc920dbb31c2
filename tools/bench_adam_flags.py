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
