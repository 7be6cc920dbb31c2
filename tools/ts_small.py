"""Cycle stamps of the fp16x2 conv kernel on a SMALL grid (one-level layer shapes), for SM_CONV_KG=1 / 2.
Run with SM_CONV_STAMP=1 [SM_CONV_KG=..] [SM_CONV_FORCE_SPLITS=..]: ts_small.py cin cout H W
Prints, per wave class: prologue, per-stage mean of the loop, exchange (KG = 2), epilogue / slab store, whole kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import torch.nn.functional as F
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap
cin, cout, H, W = [int(a) for a in sys.argv[1:5]]
kg = int(os.environ.get("SM_CONV_KG", "2"))
x = FMap(cin, H, W).from_dense(F.relu(torch.randn(cin, H, W, device="cuda")))
w = ops.pack_conv_fwd(torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5)
w2 = ops.pack_conv_split2(w)
b = torch.zeros(cout, device="cuda")
out = FMap(cout, H, W)
ops.CONV_MODE = "split2"
amax_in = ops.new_amax("cuda", float(x.planes.abs().max()))
ws = ops.splitk_workspace(w.device)
for _ in range(3):
    ws[15 * 1024 * 1024:].zero_()
    ops.conv3x3(x, w, b, out, hip.EPI_BIAS_RELU, wt2=w2, amax_in=amax_in, amax_out=ops.new_amax("cuda"))
torch.cuda.synchronize()
nb = 256
ts = ws[15 * 1024 * 1024:].view(torch.int64)[: nb * kg * 4 * 64].view(nb * kg * 4, 64).cpu().numpy().astype(np.int64)
ok = ts[:, 0] > 0
t = ts[ok]
print(f"{cin}->{cout} {H}x{W} KG={kg}: stamped waves {ok.sum()} (blocks {ok.sum() // (4 * kg)})")
t0 = t[:, 0].min()
pro = t[:, 1] - t[:, 0]
loop = t[:, 30] - t[:, 1]
ch0 = t[:, 10] - t[:, 1]
print(f"start skew {(t[:,0]-t0).mean():.0f} (max {(t[:,0]-t0).max()})  prologue {pro.mean():.0f}  loop {loop.mean():.0f}  chunk0 (9 stages) {ch0.mean():.0f}")
has1 = t[:, 22] > 0
if has1.any():
    ch1 = t[has1][:, 22] - t[has1][:, 10]
    print(f"chunk1 (9 stages) {ch1.mean():.0f} = {ch1.mean()/9:.0f} per stage")
    for tap in range(9):
        tt = t[has1]
        start = tt[:, 13 + tap] if tap else tt[:, 10]
        mf = tt[:, 32 + tap] - start
        end = tt[:, 14 + tap]
        bar = (end - tt[:, 41 + tap]) if tap % 3 == 2 else np.zeros_like(end)
        tail = end - tt[:, 32 + tap] - bar
        print(f"  tap {tap}: mfma-phase {mf.mean():6.0f}  tail {tail.mean():6.0f}  barrier-wait {bar.mean():6.0f}  stage {(end - start).mean():6.0f}")
if kg == 2:
    ex = t[:, 50] - t[:, 30]
    print(f"exchange {ex.mean():.0f}")
    after = t[:, 50]
else:
    after = t[:, 30]
fin = np.where(t[:, 51] > 0, t[:, 51], t[:, 31])
print(f"epilogue / slab store {(fin - after).mean():.0f}   wave total {(fin - t[:,0]).mean():.0f}   kernel span {fin.max() - t0}")
