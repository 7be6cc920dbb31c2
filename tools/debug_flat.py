import sys, os
sys.path[:0] = [os.getcwd(), 'oracle', 'tests']
import numpy as np, torch
import stylemesh_oracle as O
from conftest import batch_from_golden, load_golden
from golden_cases import *
from stylemesh_amd.data import synthetic as S
from test_engine_gpu import make_engine
name = sys.argv[1] if len(sys.argv) > 1 else "flat_single"
cfgd = dict(FLAGSETS[name])
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    if k in LOSS_WEIGHTS: LOSS_WEIGHTS[k] = float(v)
    elif k in ("mode", "gram"): cfgd[k] = v
    else: cfgd[k] = type(cfgd[k])(eval(v))
print(cfgd, LOSS_WEIGHTS)
d = load_golden("g5_" + name)
T = torch.from_numpy
init = [T(d[f"init{i}"]) for i in range(4)]
eng = make_engine(cfgd, init)
batch = batch_from_golden(d)
ocfg = O.OracleConfig(hierarchical=cfgd["hier"], style_weights=STYLE_WEIGHTS, angle_threshold=cfgd["thr"],
                      style_pyramid_mode=cfgd["mode"], gram_mode=cfgd["gram"], use_angle_weight=cfgd["angle"],
                      use_depth_scaling=cfgd["depth"], loss_weights=dict(LOSS_WEIGHTS))
pipe = O.OraclePipeline(S.seeded_vgg_state(VGG_SEED), S.style_image(STYLE_SEED, *STYLE_HW), ocfg, (TEX, TEX), init_layers=init)
rec = {}
losses, grads = pipe.grads(batch, rec)
eng.set_view(batch); eng.forward_backward()
print("active", [lv.index for lv in eng.view if lv.active], rec["active"])
for a, i in enumerate(rec["active"]):
    lv = eng.view[i]; b = eng._level_bufs(lv.H, lv.W)
    for layer in eng.loss_layers:
        ref = rec["enc"][a][layer].detach()[0]; mine = b.act[layer].to_dense().cpu()
        print(i, layer, "feat relerr %.2e" % float((mine-ref).abs().max()/ref.abs().max()), "factor", float(lv.factor[layer]), float(rec["factors"][a][layer]),
              "mask diff", int((lv.masks[layer].to_dense()[0].cpu() != rec["info"][a][layer]["m"][0,0]).sum()))
    for layer in [n for n in b.grad if n != "img"]:
        t = rec["all_acts"][a][layer]
        if t.grad is None: continue
        ref = (t.grad * (t.detach() > 0))[0] if layer.startswith("r") else t.grad[0]
        mine = b.grad[layer].to_dense().cpu()
        err = (mine - ref).abs()
        print(i, layer, "dZ: max err %.3e max ref %.3e n>1e-3max %d / %d" % (float(err.max()), float(ref.abs().max()), int((err > 1e-3*ref.abs().max()).sum()), err.numel()))
    ref = rec["pred_grads_raw"][i][0]; mine = b.grad["img"].to_dense().cpu()
    err = (mine-ref).abs()
    print(i, "raw img grad: max err %.3e max ref %.3e  n>1e-3max %d / %d" % (float(err.max()), float(ref.abs().max()), int((err > 1e-3*ref.abs().max()).sum()), err.numel()))
    ys, xs = np.where((err.max(0).values > 1e-3*float(ref.abs().max())).numpy())
    if len(ys): print("   bbox y %d..%d x %d..%d" % (ys.min(), ys.max(), xs.min(), xs.max()))
for i, (g, c, p) in enumerate(zip(eng.grads, eng.reg_coef, eng.layers)):
    mine = (g + c*p).cpu(); ref = grads[i]
    err = (mine-ref).abs()
    print("tex grad", i, "max err %.3e max ref %.3e n bad %d" % (float(err.max()), float(ref.abs().max()), int((err > 1e-3*ref.abs()+2e-4*ref.abs().max()).sum())))
