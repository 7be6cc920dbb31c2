"""conv1_1 forward (3 -> 64) and its data gradient (64 -> 3) over the four c3 UV levels, grouped, dense (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap
LEVELS = [(256, 341), (432, 576), (608, 811), (784, 1045)]
wgt = torch.randn(64, 3, 3, 3, device="cuda") * 0.2
wd, wf, b = ops.pack_conv_dgrad(wgt), ops.pack_conv_fwd(wgt), torch.zeros(64, device="cuda")
dz = [FMap(64, h, w).from_dense(torch.randn(64, h, w, device="cuda")) for h, w in LEVELS]
img = [FMap(4, h, w).from_dense(torch.randn(3, h, w, device="cuda")) for h, w in LEVELS]
out3 = [FMap(3, h, w) for h, w in LEVELS]
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
mb = sum(64 * h * w * 4 for h, w in LEVELS) / 1e6
t = timed(lambda: ops.conv3x3_dgrad_c3_grouped(list(zip(dz, out3)), wd))
print(f"dgrad_c3 grouped, dense: {t:7.1f} us  ({mb / t:.2f} TB/s of the 64-channel gradient)")
t = timed(lambda: ops.conv3x3_grouped([(i, d, None) for i, d in zip(img, dz)], wf, b, hip.EPI_BIAS_RELU))
print(f"conv1_1 forward grouped, dense: {t:7.1f} us  ({mb / t:.2f} TB/s of the 64-channel output)")
