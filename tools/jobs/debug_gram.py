import sys, os, types
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch, pytest
import test_resident_gpu as T
import torch.nn.functional as F
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap
ops.CONV_MODE = "split2"; ops.GRAM_MODE = "split2"
hws, two_masks = [(37, 50)], True
C = 64
torch.manual_seed(len(hws) * 7 + two_masks)
wgt = torch.randn(C, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
wd = ops.pack_conv_dgrad(wgt).cuda(); wd2 = ops.pack_conv_split2(wd)
D0 = (torch.randn(C, C) * 3e-3).cuda(); D1 = (torch.randn(C, C) * 1e-3).cuda()
feats, masks, dps, codes = [], [], [], []
for g, (H, W) in enumerate(hws):
    feats.append(FMap(C, H, W).from_dense(F.relu(torch.randn(C, H, W) * 2).cuda()))
    mk = torch.zeros(2, H, W); sel = torch.rand(H, W)
    mk[0] = (sel < 0.3).float(); mk[1] = ((sel >= 0.3) & (sel < 0.45)).float(); mk[:, H // 2:, : W // 3] = 0
    masks.append(FMap(2, H, W).from_dense(mk.cuda()))
    a, pooled = FMap(C, H, W).from_dense(F.relu(torch.randn(C, H, W)).cuda()), FMap(C, H // 2, W // 2)
    code = torch.zeros(C // 8 * pooled.plane, dtype=torch.int32, device="cuda")
    ops.maxpool_fwd_grouped([(a, pooled)], None, [code]); codes.append(code)
    dps.append(FMap(C, H // 2, W // 2).from_dense((torch.randn(C, H // 2, W // 2) * 1e-4).cuda()))
af = ops.new_amax("cuda", max(float(f.planes.abs().max()) for f in feats))
ad = ops.new_amax("cuda", max(float(D0.abs().max()), float(D1.abs().max())))
amax_in = ops.new_amax("cuda", max(float(d.planes.abs().max()) for d in dps))
needs = T._needs(hws, False, 13)
lst = torch.cat([T._quad_cover(ops, hip, nd, g) for g, nd in enumerate(needs)])
big = torch.zeros(256, device="cuda")
ops.splitk_workspace = lambda device: big
def run(pipe, blocks=None):
    os.environ["SM_RES_PIPE_MIN"] = "8" if pipe else "0"
    if blocks: os.environ["SM_RES_PIPE_BLOCKS"] = str(blocks)
    out = [FMap(C, H, W) for (H, W) in hws]
    ws2 = [torch.empty(ops.gram_backward_ws_bytes(C), dtype=torch.uint8, device="cuda") for _ in hws]
    ops.gram_backward_grouped(ops.struct_array(hip.GramBwdProblem, [
        ops.gram_bwd_problem(f, m.channel_ptr(0), m.channel_ptr(1), D0, D1, None, w_, af, ad, relu_gate=False)
        for f, m, w_ in zip(feats, masks, ws2)]))
    ops.conv3x3_grouped([(d, o, f, c, None, None, (w_, m.channel_ptr(0), m.channel_ptr(1), af, ad))
                         for d, o, f, c, w_, m in zip(dps, out, feats, codes, ws2, masks)], wd, None,
                        hip.EPI_RELU_MASK | hip.EPI_GRAM, lst, 1.0, wd2, amax_in, ops.new_amax("cuda"), quads=True)
    torch.cuda.synchronize()
    return out
ref = run(False)
for blocks in (None, 8):
    out = run(True, blocks)
    o, r = out[0].planes, ref[0].planes
    bad = (o != r)
    print("blocks", blocks, "n_list", lst.numel(), "mismatch", int(bad.sum()), "of", int((r != 0).sum()), "nonzero ref")
    if bad.any():
        ch = bad.any(1).nonzero().flatten().tolist(); print("channels with mismatch:", ch[:70])
        pos = bad.any(0).nonzero().flatten(); Wp = out[0].Wp
        print("positions:", len(pos), [(int(p) // Wp - 1, int(p) % Wp - 1) for p in pos[:40]])
        d = (o - r).abs(); print("max diff", float(d.max()), "max ref", float(r.abs().max()))
        # does the output equal the ref WITHOUT the gram term or without gate?
        k = bad.nonzero()[:10]
        for c_, p_ in k.tolist(): print(c_, p_, float(o[c_, p_]), float(r[c_, p_]))
