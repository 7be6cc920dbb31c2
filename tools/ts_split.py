"""Per-stage cycle stamps of the split conv kernel (run with SM_CONV_STAMP=1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap
cin, cout, H, W = [int(a) for a in sys.argv[2:6]] if len(sys.argv) > 5 else (512, 512, 98, 130)
x = FMap(cin, H, W); x.planes.normal_()
w = ops.pack_conv_fwd(torch.randn(cout, cin, 3, 3, device="cuda") * 0.02)
w3, w2 = ops.pack_conv_split(w), ops.pack_conv_split2(w)
b = torch.zeros(cout, device="cuda")
out = FMap(cout, H, W)
ops.CONV_MODE = sys.argv[1] if len(sys.argv) > 1 else "split2"
amax_in = ops.new_amax("cuda", float(x.planes.abs().max()))
ws = ops.splitk_workspace(w.device)
for _ in range(2):
    ws.zero_()
    ops.conv3x3(x, w, b, out, hip.EPI_BIAS_RELU, wt3=w3, wt2=w2, amax_in=amax_in, amax_out=ops.new_amax("cuda"))
torch.cuda.synchronize()
ts = ws[15 * 1024 * 1024:].view(torch.int64)[: 256 * 4 * 64].view(256, 4, 64).cpu().numpy()
for blk in (0, 1, 100, 255):
    t = ts[blk, 0]
    print(f"block {blk} wave0: prologue {t[1]-t[0]}  stages(ch0): {np.diff(t[1:12]).tolist()}  stages(ch1): {np.diff(t[[11]+list(range(14,24))]).tolist()}  loop total {t[30]-t[1]} epilogue {t[31]-t[30]}")
# chunk 1 (steady state), all stamped blocks / waves: per tap  MFMA phase | tail (loads, convert, stores) | barrier wait
t = ts.astype(np.int64)
ok = (t[:, :, 30] > t[:, :, 1]) & (t[:, :, 14] > 0)
for tap in range(9):
    start = t[:, :, 13 + tap] if tap else t[:, :, 10]          # end of the previous stage (tap 8 of chunk 0 = slot 10)
    mf = t[:, :, 32 + tap] - start
    end = t[:, :, 14 + tap]
    bar = (end - t[:, :, 41 + tap]) if tap % 3 == 2 else np.zeros_like(end)
    tail = end - t[:, :, 32 + tap] - bar
    print(f"tap {tap}: mfma-phase {mf[ok].mean():7.0f}  tail {tail[ok].mean():7.0f}  barrier-wait {bar[ok].mean():7.0f}  stage {(end - start)[ok].mean():7.0f}")
tot = ts[:, :, 31] - ts[:, :, 0]
loop = ts[:, :, 30] - ts[:, :, 1]
print("mean total", tot.mean(), "mean loop", loop.mean(), "per stage", loop.mean() / (32 * 9), " span of whole kernel", ts[:, :, 31].max() - ts[:, :, 0].min())
