"""Block timeline of the resident-input kernel (library built with -DSM_RES_TRACE): per block start / staged / loop done /
end (100 MHz realtime counter) and the CU it ran on; prints how many blocks a CU holds at a time and the phase lengths.
conv1_2 forward + pool over one 768 x 512 level (3072 quads).   (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import torch.nn.functional as F
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap
from bench_resident_lib import rows_list

ops.CONV_MODE = "split2"
H, W = 768, 512
wgt = torch.randn(64, 64, 3, 3) * (2.0 / (9 * 64)) ** 0.5
b = (torch.randn(64) * 0.3).cuda()
w = ops.pack_conv_fwd(wgt).cuda()
w2 = ops.pack_conv_split2(w)
x = FMap(64, H, W).from_dense(F.relu(torch.randn(64, H, W, device="cuda")))
amax_in = ops.new_amax("cuda", float(x.planes.abs().max()))
out, pooled = FMap(64, H, W), FMap(64, H // 2, W // 2)
codes = torch.zeros(8 * pooled.plane, dtype=torch.int32, device="cuda")
lst = rows_list([(H, W)], 4, 4)
ws = ops.splitk_workspace(w.device)
n = lst.numel() // 4
for _ in range(3):
    ws[15 * 1024 * 1024:].zero_()
    ops.conv3x3_grouped([(x, out, None, None, pooled, codes)], w, b, hip.EPI_BIAS_RELU | hip.EPI_POOL, lst, 1.0, None, w2,
                        amax_in, ops.new_amax("cuda"), quads=True)
torch.cuda.synchronize()
t = ws[15 * 1024 * 1024:].view(torch.int64)[: n * 8].view(n, 8).cpu().numpy()
t0 = t[:, 0].min()
us = (t[:, :4] - t0) / 100.0
hw, xcc = t[:, 4], t[:, 5] & 0xF
cu = ((xcc << 16) | (hw & 0xFFFF00)).astype(np.int64)     # everything of HW_ID above the wave / SIMD bits + the XCC
print(f"{n} blocks on {len(set(cu.tolist()))} distinct (XCC, SE / SH / CU) ids; kernel span {us[:, 3].max():.1f} us")
iss, arr = (t[:, 6] - t0) / 100.0, (t[:, 7] - t0) / 100.0
print(f"staging, wave 0: address plan + weight prefetch + load issue {np.mean(iss - us[:, 0]):.2f}  loads arrive {np.mean(arr - iss):.2f}  "
      f"convert + store + barrier {np.mean(us[:, 1] - arr):.2f}")
print(f"phases (us, mean): staging {np.mean(us[:, 1] - us[:, 0]):.2f}  loop {np.mean(us[:, 2] - us[:, 1]):.2f}  epilogue {np.mean(us[:, 3] - us[:, 2]):.2f}"
      f"  block life {np.mean(us[:, 3] - us[:, 0]):.2f}")
conc = []
for c in set(cu.tolist()):
    sel = us[cu == c]
    ev = sorted([(s, 1) for s in sel[:, 0]] + [(e, -1) for e in sel[:, 3]])
    cur = peak = 0
    area = 0.0
    last = ev[0][0]
    for tt, d in ev:
        area += cur * (tt - last)
        last = tt
        cur += d
        peak = max(peak, cur)
    conc.append((peak, area / (ev[-1][0] - ev[0][0])))
print(f"blocks resident per CU: peak {np.mean([c[0] for c in conc]):.2f} (max {max(c[0] for c in conc)}), time-average {np.mean([c[1] for c in conc]):.2f}")
