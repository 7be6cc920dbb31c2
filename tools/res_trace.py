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
ops.GRAM_MODE = "split2"
VARIANT = sys.argv[1] if len(sys.argv) > 1 else "fwd"      # fwd: conv1_2 forward + pool; gram: conv1_2's data gradient (un-pool + Gram)
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
if VARIANT == "gram":
    C = 64
    wd = ops.pack_conv_dgrad(wgt).cuda()
    wd2 = ops.pack_conv_split2(wd)
    D0, D1 = (torch.randn(C, C) * 3e-3).cuda(), (torch.randn(C, C) * 1e-3).cuda()
    feat = x
    sel = torch.rand(H, W)
    mk = torch.stack([(sel < 0.5).float(), (sel >= 0.5).float()])
    masks = FMap(2, H, W).from_dense(mk.cuda())
    act = FMap(C, H, W).from_dense(F.relu(torch.randn(C, H, W, device="cuda")))
    pl = FMap(C, H // 2, W // 2)
    code = torch.zeros(C // 8 * pl.plane, dtype=torch.int32, device="cuda")
    ops.maxpool_fwd_grouped([(act, pl)], None, [code])
    dp = FMap(C, H // 2, W // 2).from_dense(torch.randn(C, H // 2, W // 2, device="cuda") * 1e-4)
    af = ops.new_amax("cuda", float(feat.planes.abs().max()))
    ad = ops.new_amax("cuda", max(float(D0.abs().max()), float(D1.abs().max())))
    a_in = ops.new_amax("cuda", float(dp.planes.abs().max()))
    gws = torch.empty(ops.gram_backward_ws_bytes(C), dtype=torch.uint8, device="cuda")
    ops.gram_backward_grouped(ops.struct_array(hip.GramBwdProblem, [
        ops.gram_bwd_problem(feat, masks.channel_ptr(0), masks.channel_ptr(1), D0, D1, None, gws, af, ad, relu_gate=False)]))

    def launch():
        ops.conv3x3_grouped([(dp, out, feat, code, None, None, (gws, masks.channel_ptr(0), masks.channel_ptr(1), af, ad))], wd, None,
                            hip.EPI_RELU_MASK | hip.EPI_GRAM, lst, 1.0, None, wd2, a_in, ops.new_amax("cuda"), quads=True)
else:
    def launch():
        ops.conv3x3_grouped([(x, out, None, None, pooled, codes)], w, b, hip.EPI_BIAS_RELU | hip.EPI_POOL, lst, 1.0, None, w2,
                            amax_in, ops.new_amax("cuda"), quads=True)
for _ in range(3):
    ws[15 * 1024 * 1024:].zero_()
    launch()
torch.cuda.synchronize()
t = ws[15 * 1024 * 1024:].view(torch.int64)[: n * 16].view(n, 16).cpu().numpy()
t0 = t[:, 0].min()
us = (t[:, :4] - t0) / 100.0
hw, xcc = t[:, 4], t[:, 5] & 0xF
cu = ((xcc << 16) | (hw & 0xFFFF00)).astype(np.int64)     # everything of HW_ID above the wave / SIMD bits + the XCC
print(f"{n} blocks on {len(set(cu.tolist()))} distinct (XCC, SE / SH / CU) ids; kernel span {us[:, 3].max():.1f} us")
iss, arr = (t[:, 6] - t0) / 100.0, (t[:, 7] - t0) / 100.0
print(f"staging, wave 0: address plan + weight prefetch + load issue {np.mean(iss - us[:, 0]):.2f}  loads arrive {np.mean(arr - iss):.2f}  "
      f"convert + store + barrier {np.mean(us[:, 1] - arr):.2f}")
cv = (t[:, 10:14] - t0) / 100.0
print("converted + stored, waves 0..3 after wave 0's loads arrived: " + "  ".join(f"{np.mean(cv[:, k] - arr):.2f}" for k in range(4)))
if VARIANT == "gram":
    g8, g9 = (t[:, 8] - t0) / 100.0, (t[:, 9] - t0) / 100.0
    print(f"Gram epilogue: F staged {np.mean(g8 - us[:, 2]):.2f}  Gram MFMAs {np.mean(g9 - g8):.2f}  gate + stores {np.mean(us[:, 3] - g9):.2f}")
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
