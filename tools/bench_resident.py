"""The resident-input kernel (SM_LIST_QUADS) against the ring kernel on the three 64-output-channel launches of a c3 step:
conv1_2 forward with the pooling epilogue, conv2_1's data gradient, conv1_2's data gradient over the un-pooled gradient
(ReLU gate; the Gram epilogue is measured in situ). Dense host-built lists over the same rows / columns for both kernels;
results compared bit for bit, launches timed with events. Usage: bench_resident.py [c3|c2]   (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import torch.nn.functional as F
from stylemesh_amd.runtime import hip, ops
from stylemesh_amd.runtime.fmap import FMap

LEVELS = {"c3": [(256, 341), (432, 576), (608, 811), (784, 1045)], "c2": [(256, 341)]}[sys.argv[1] if len(sys.argv) > 1 else "c3"]
from bench_resident_lib import rows_list, timed  # noqa: E402

ops.CONV_MODE = "split2"
torch.manual_seed(1)
# ---- conv1_2 forward, pooling epilogue
hws = LEVELS
wgt = torch.randn(64, 64, 3, 3) * (2.0 / (9 * 64)) ** 0.5
b = (torch.randn(64) * 0.3).cuda()
w = ops.pack_conv_fwd(wgt).cuda()
w2 = ops.pack_conv_split2(w)
xs = [F.relu(torch.randn(64, H, W, device="cuda") * 2) for H, W in hws]
ins = [FMap(64, H, W).from_dense(x) for x, (H, W) in zip(xs, hws)]
amax_in = ops.new_amax("cuda", max(float(x.abs().max()) for x in xs))
res = {}
for name, lst, quads in (("ring <64,256> pairs", rows_list(hws, 2, 8), False), ("resident quads", rows_list(hws, 4, 4), True)):
    outs = [FMap(64, H, W) for H, W in hws]
    pooled = [FMap(64, H // 2, W // 2) for H, W in hws]
    codes = [torch.zeros(8 * p.plane, dtype=torch.int32, device="cuda") for p in pooled]
    am = ops.new_amax("cuda")
    run = lambda: ops.conv3x3_grouped([(i, o, None, None, p, c) for i, o, p, c in zip(ins, outs, pooled, codes)], w, b,
                                      hip.EPI_BIAS_RELU | hip.EPI_POOL, lst, 1.0, None, w2, amax_in, am, quads=quads)
    t = timed(run)
    res[name] = ([p.to_dense() for p in pooled], [c.clone() for c in codes], float(am.max()))
    gf = sum(2 * 9 * 64 * 64 * H * W for H, W in hws) / 1e9
    print(f"conv1_2 forward + pool, {name:22s}: {t:7.1f} us  ({gf / t * 1e3:.1f} TFLOP/s)")
a, r = res["ring <64,256> pairs"], res["resident quads"]
print("  bit-identical pooled maps:", all(torch.equal(x, y) for x, y in zip(a[0], r[0])),
      " codes:", all(torch.equal(x, y) for x, y in zip(a[1], r[1])), " bound:", a[2] == r[2])

# ---- conv2_1 data gradient: 128 -> 64 at half resolution, plain epilogue
hw2 = [(H // 2, W // 2) for H, W in hws]
wgt = torch.randn(128, 64, 3, 3) * (2.0 / (9 * 64)) ** 0.5
wd = ops.pack_conv_dgrad(wgt).cuda()
wd2 = ops.pack_conv_split2(wd)
gs = [torch.randn(128, H, W, device="cuda") for H, W in hw2]
gin = [FMap(128, H, W).from_dense(x) for x, (H, W) in zip(gs, hw2)]
amax_in = ops.new_amax("cuda", max(float(x.abs().max()) for x in gs))
res = {}
for name, lst, quads in (("ring <64,256> rows", rows_list(hw2, 1, 8), False), ("resident quads", rows_list(hw2, 4, 4), True)):
    outs = [FMap(64, H, W) for H, W in hw2]
    am = ops.new_amax("cuda")
    run = lambda: ops.conv3x3_grouped([(i, o, None) for i, o in zip(gin, outs)], wd, None, 0, lst, 1.0, None, wd2, amax_in, am,
                                      quads=quads)
    t = timed(run)
    res[name] = ([o.to_dense() for o in outs], float(am.max()))
    gf = sum(2 * 9 * 128 * 64 * H * W for H, W in hw2) / 1e9
    print(f"conv2_1 data gradient,  {name:22s}: {t:7.1f} us  ({gf / t * 1e3:.1f} TFLOP/s)")
a, r = res["ring <64,256> rows"], res["resident quads"]
print("  bit-identical:", all(torch.equal(x, y) for x, y in zip(a[0], r[0])), " bound:", a[1] == r[1])

# ---- conv1_2 data gradient: un-pooled gradient of p1 (64 channels at half resolution + codes) -> 64 channels, ReLU gate
wgt = torch.randn(64, 64, 3, 3) * (2.0 / (9 * 64)) ** 0.5
wd = ops.pack_conv_dgrad(wgt).cuda()
wd2 = ops.pack_conv_split2(wd)
gp = [FMap(64, H, W).from_dense(torch.randn(64, H, W, device="cuda")) for H, W in hw2]
code = []
rng = np.random.default_rng(3)
for p in gp:   # nibble k of dword [g][position] = code of channel 8 g + k (0..3: window element, 4: closed window)
    c = rng.integers(0, 5, (8, 8, p.plane), dtype=np.uint32)
    dw = np.zeros((8, p.plane), dtype=np.uint32)
    for k in range(8):
        dw |= c[:, k] << np.uint32(4 * k)
    code.append(torch.from_numpy(dw.view(np.int32).reshape(-1)).cuda())
gate = [FMap(64, H, W).from_dense(F.relu(torch.randn(64, H, W, device="cuda"))) for H, W in hws]
amax_in = ops.new_amax("cuda", max(float(p.planes.abs().max()) for p in gp))
res = {}
for name, lst, quads in (("ring <64,256> rows", rows_list(hws, 1, 8), False), ("resident quads", rows_list(hws, 4, 4), True)):
    outs = [FMap(64, H, W) for H, W in hws]
    am = ops.new_amax("cuda")
    run = lambda: ops.conv3x3_grouped([(i, o, g, c) for i, o, g, c in zip(gp, outs, gate, code)], wd, None, hip.EPI_RELU_MASK,
                                      lst, 1.0, None, wd2, amax_in, am, quads=quads)
    t = timed(run)
    res[name] = ([o.to_dense() for o in outs], float(am.max()))
    gf = sum(2 * 9 * 64 * 64 * H * W for H, W in hws) / 1e9
    print(f"conv1_2 data gradient (un-pool, gate), {name:22s}: {t:7.1f} us  ({gf / t * 1e3:.1f} TFLOP/s)")
a, r = res["ring <64,256> rows"], res["resident quads"]
print("  bit-identical:", all(torch.equal(x, y) for x, y in zip(a[0], r[0])), " bound:", a[1] == r[1])
