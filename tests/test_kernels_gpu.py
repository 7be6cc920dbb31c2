"""Per-kernel parity: each C-ABI entry point on the GPU against the oracle (torch CPU fp32) on seeded inputs.
fp32 tolerances are stated per test; summation order differs (MFMA k-order, atomics) so nothing is bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import stylemesh_oracle as O
from conftest import load_golden
from gpu_util import assert_close, rel_err, require_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    require_gpu()
    from stylemesh_amd.runtime import fmap, hip, ops
    return type("RT", (), dict(FMap=fmap.FMap, hip=hip, ops=ops))


def dev(x):
    return x.contiguous().cuda()


# ------------------------------------------------------------------ K1 / K2
def test_tex_sample_forward_backward_golden(rt):
    d = load_golden("g1_texture")
    grid, up = torch.from_numpy(d["grid"]), torch.from_numpy(d["upstream"])
    layers = [torch.from_numpy(d[f"layer{i}"]).clamp(O.CLAMP_LO, O.CLAMP_HI) for i in range(4)]
    dl = [dev(l) for l in layers]
    h, w = grid.shape[1:3]
    for n_layers, key_out, key_grad in ((1, "flat_out", "flat_grad"), (4, "hier_out", "hier_grad")):
        out = rt.FMap(4, h, w)
        rt.ops.tex_sample_fwd(dl[:n_layers], dev(grid), out)
        assert_close(out.to_dense(3), d[key_out][0], 1e-5, 2e-4, key_out)
        assert out.border_is_zero()
        gimg = rt.FMap(3, h, w).from_dense(up[0])
        grads = [torch.zeros_like(l) for l in dl[:n_layers]]
        rt.ops.tex_sample_bwd(grads, dev(grid), gimg)
        for i, g in enumerate(grads):
            ref = d[key_grad] if n_layers == 1 else d[f"{key_grad}{i}"]
            assert_close(g, ref, 1e-5, 1e-5 * float(np.abs(ref).max()), f"{key_grad}{i}")
    # known answers of SURVEY.md 8 a3
    tex = dev(torch.arange(12.).view(1, 3, 4).repeat(3, 1, 1))
    pg = dev(torch.tensor([[[[-1., -1.], [1., 1.], [0., 0.], [1.2, -3.]]]]))
    out = rt.FMap(3, 1, 4)
    rt.ops.tex_sample_fwd([tex], pg, out)
    assert_close(out.to_dense()[0, 0], [0, 11, 5.5, 3], 1e-6, 1e-6)


def test_tex_sample_grouped_equals_per_level_and_step_begin(rt):
    """sm_tex_sample_fwd_grouped over several images = sm_tex_sample_fwd per image, bit for bit; sm_step_begin computes
    sum_l coef_l * sumsq_l and zeroes both ranges (and only them)."""
    d = load_golden("g1_texture")
    layers = [dev(torch.from_numpy(d[f"layer{i}"]).clamp(O.CLAMP_LO, O.CLAMP_HI)) for i in range(4)]
    torch.manual_seed(5)
    grids = [dev(torch.rand(1, h, w, 2) * 2.4 - 1.2) for h, w in ((17, 23), (40, 31), (9, 64), (33, 33))]
    outs = [rt.FMap(4, g.shape[1], g.shape[2]) for g in grids]
    refs = [rt.FMap(4, g.shape[1], g.shape[2]) for g in grids]
    rt.ops.tex_sample_fwd_grouped(layers, grids, outs)
    for g, r in zip(grids, refs):
        rt.ops.tex_sample_fwd(layers, g, r)
    for o, r in zip(outs, refs):
        assert torch.equal(o.to_dense(3), r.to_dense(3)) and o.border_is_zero()
    sumsq = dev(torch.tensor([1.5, 2.25, 4.0, 0.125]))
    coef = dev(torch.tensor([2.0, 0.5, 0.25, 8.0]))
    reg = torch.full((1,), -1.0).cuda()
    buf = torch.full((4096 + 8 + 1024 + 8,), 3.0).cuda()
    a, b = buf[4:4 + 4096], buf[4 + 4096 + 4:4 + 4096 + 4 + 1024]
    rt.ops.step_begin(sumsq, coef, reg, a, b)
    assert float(reg) == 1.5 * 2.0 + 2.25 * 0.5 + 4.0 * 0.25 + 0.125 * 8.0
    assert float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0
    keep = torch.ones_like(buf, dtype=torch.bool)
    keep[4:4 + 4096] = False
    keep[4 + 4096 + 4:4 + 4096 + 4 + 1024] = False
    assert bool((buf[keep] == 3.0).all())
    rt.ops.step_begin(sumsq, coef, reg, None, None)      # regulariser only (graph mode)
    assert float(reg) == 6.125


def test_tex_sample_backward_pixel_weight_and_accumulate(rt):
    torch.manual_seed(0)
    H, W, h, w = 33, 47, 40, 56
    grid = torch.rand(1, h, w, 2) * 2.2 - 1.1
    up = torch.randn(1, 3, h, w)
    pw = torch.rand(h, w)
    pw[:5] = 0
    ref = O.grid_sample_border_backward_explicit((3, H, W), grid, up * pw)
    g = torch.full((3, H, W), 0.5).cuda()
    rt.ops.tex_sample_bwd([g], dev(grid), rt.FMap(3, h, w).from_dense(up[0]), dev(pw))
    assert_close(g - 0.5, ref, 1e-4, 1e-5 * float(ref.abs().max()))


def test_tex_scatter_planned_matches_oracle_and_atomic_path(rt):
    """The sorted-gather form of the scatter (plan once per view, walk the sorted list every step) over several UV
    levels and a 3-layer hierarchical texture, with pixel weights (some zero), out-of-range UVs (border clamp),
    a coarse layer (long runs of equal texels that cross wave boundaries) and a non-zero arena (it accumulates)."""
    torch.manual_seed(3)
    shapes = [(3, 64, 96), (3, 32, 48), (3, 4, 6)]
    n = sum(c * h * w for c, h, w in shapes)
    arena = torch.full((n,), 0.25).cuda()
    arena2 = arena.clone()
    views, off = [], 0
    for c, h, w in shapes:
        views.append((arena[off:off + c * h * w].view(c, h, w), arena2[off:off + c * h * w].view(c, h, w)))
        off += c * h * w
    g_plan, g_atomic = [v[0] for v in views], [v[1] for v in views]
    levels = [(40, 56), (23, 31), (64, 64), (96, 128)]
    grids = [torch.rand(1, h, w, 2) * 2.3 - 1.15 for h, w in levels]
    grids[3][:] = 1.7      # 12288 pixels clamped onto ONE corner texel per layer: a run of thousands of entries
    ups = [torch.randn(1, 3, h, w) for h, w in levels]
    pws = [torch.rand(h, w) for h, w in levels]
    pws[0][:7] = 0
    pws[2] = None
    pws[3] = None
    gimgs = [rt.FMap(3, h, w).from_dense(u[0]) for (h, w), u in zip(levels, ups)]
    dgrids = [dev(g) for g in grids]
    dpws = [None if p is None else dev(p) for p in pws]
    plan = rt.ops.ScatterPlan(g_plan, arena)
    plan.build([g[0] for g in dgrids], dpws)
    for rep in range(2):     # the same plan serves every step of the view
        plan.scatter(gimgs)
        for g, gi, p in zip(dgrids, gimgs, dpws):
            rt.ops.tex_sample_bwd(g_atomic, g, gi, p)
    for li, (c, h, w) in enumerate(shapes):
        ref = sum(O.grid_sample_border_backward_explicit((3, h, w), g, u if p is None else u * p)
                  for g, u, p in zip(grids, ups, pws))
        scale = float(ref.abs().max())
        assert_close(g_plan[li] - 0.25, 2 * ref, 1e-4, 2e-5 * scale, f"planned vs oracle, layer {li}")
        assert_close(g_plan[li], g_atomic[li], 1e-4, 2e-5 * scale, f"planned vs atomic path, layer {li}")
    # without monster runs every texel has exactly one writer: bit-reproducible, stored or accumulated
    plan3 = rt.ops.ScatterPlan(g_plan, arena)
    plan3.build([g[0] for g in dgrids[:3]], dpws[:3])
    outs = []
    for accumulate in (False, True, False):
        arena.zero_()
        plan3.scatter(gimgs[:3], accumulate=accumulate)
        outs.append(arena.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


# ------------------------------------------------------------------ K3 / K4
CONV_CASES = [
    # (Cin, Cout, H, W)
    (3, 64, 17, 21),      # first layer, KC = 4
    (3, 64, 40, 301),     # first layer, many 256-position tiles, ragged last tile
    (64, 64, 40, 56),     # BM = 64 tile
    (64, 128, 20, 28),    # BM = 128 tile
    (128, 128, 13, 131),  # wide & short, several N tiles per row
    (256, 256, 9, 11),    # tiny plane: one partially filled N tile
    (512, 512, 5, 7),
    (64, 64, 3, 300),
]


@pytest.mark.parametrize("cin,cout,H,W", CONV_CASES)
def test_conv3x3_forward(rt, cin, cout, H, W):
    torch.manual_seed(cin + cout + H)
    x = torch.randn(1, cin, H, W) * 3
    wgt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout) * 0.3
    ref = F.relu(F.conv2d(x, wgt, b, padding=1))[0]
    xin = rt.FMap(max(cin, 4), H, W).from_dense(x[0])
    out = rt.FMap(cout, H, W)
    out.planes.fill_(7.0)  # the kernel must (re)write every position of rows 1..H, zeros on the border columns
    out.planes[:, :out.Wp] = 0
    out.planes[:, (H + 1) * out.Wp:] = 0
    rt.ops.conv3x3(xin, dev(rt.ops.pack_conv_fwd(wgt)), dev(b), out, rt.hip.EPI_BIAS_RELU)
    assert_close(out.to_dense(), ref, 1e-4, 1e-4 * float(ref.abs().max()))
    assert out.border_is_zero()


@pytest.mark.parametrize("cin,cout,H,W", [(64, 64, 17, 21), (64, 128, 20, 28), (256, 512, 6, 9), (128, 128, 13, 131)])
def test_conv3x3_dgrad_with_gate_and_add(rt, cin, cout, H, W):
    torch.manual_seed(cin * 3 + W)
    x = torch.randn(1, cin, H, W)          # forward activation of the conv's input (post-ReLU values incl. zeros)
    x = F.relu(x).requires_grad_(True)
    wgt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    dy = torch.randn(1, cout, H, W)
    addend = torch.randn(cin, H, W)
    F.conv2d(x, wgt, None, padding=1).backward(dy)
    gate = (x.detach()[0] > 0).float()
    wd = dev(rt.ops.pack_conv_dgrad(wgt))
    dyf = rt.FMap(cout, H, W).from_dense(dy[0])
    act = rt.FMap(cin, H, W).from_dense(x.detach()[0])
    scale = float(x.grad.abs().max())
    out = rt.FMap(cin, H, W)
    rt.ops.conv3x3(dyf, wd, None, out, 0)
    assert_close(out.to_dense(), x.grad[0], 1e-4, 1e-4 * scale, "plain dgrad")
    rt.ops.conv3x3(dyf, wd, None, out, rt.hip.EPI_RELU_MASK, gate=act)
    assert_close(out.to_dense(), x.grad[0] * gate, 1e-4, 1e-4 * scale, "gated dgrad")
    out.from_dense(addend)
    rt.ops.conv3x3(dyf, wd, None, out, rt.hip.EPI_RELU_MASK | rt.hip.EPI_ADD, gate=act)
    assert_close(out.to_dense(), (x.grad[0] + addend) * gate, 1e-4, 1e-4 * scale, "gated dgrad + add")
    assert out.border_is_zero()


@pytest.mark.parametrize("cin,cout,H,W", [(64, 128, 20, 28), (256, 512, 6, 9), (128, 128, 13, 131), (512, 256, 40, 77),
                                          (64, 64, 17, 21), (128, 64, 33, 47), (128, 128, 120, 300), (256, 512, 54, 72),
                                          (64, 64, 250, 301)])
@pytest.mark.parametrize("in_scale", [1.0, 3e-7, 4e4])
def test_conv3x3_split2_matches_fp64_as_well_as_f32(rt, cin, cout, H, W, in_scale, monkeypatch):
    """fp16x2-split conv (3 partial products of fp16 pairs on the matrix cores, power-of-two operand scales from the
    recorded max |x|) against an fp64 convolution: error class of the fp32-MFMA kernel (<= 2x its rms error, inside the
    fp32 tolerance) for inputs of very different magnitude (gradients ~1e-7, activations ~1e4); the max |output| the
    launch records is the true maximum (whole tiles and K-split tail units)."""
    torch.manual_seed(cin + cout + W)
    x = F.relu(torch.randn(1, cin, H, W) * 3) * in_scale
    wgt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout) * 0.3 * in_scale
    ref = F.relu(F.conv2d(x.double(), wgt.double(), b.double(), padding=1))[0]
    xin = rt.FMap(cin, H, W).from_dense(x[0])
    w = dev(rt.ops.pack_conv_fwd(wgt))
    w2 = rt.ops.pack_conv_split2(w)
    errs = {}
    for mode in ("f32", "split2"):
        monkeypatch.setattr(rt.ops, "CONV_MODE", mode)
        out = rt.FMap(cout, H, W)
        out.planes.fill_(7.0)
        out.planes[:, :out.Wp] = 0
        out.planes[:, (H + 1) * out.Wp:] = 0
        amax_in = rt.ops.new_amax("cuda", float(x.abs().max()))
        amax_out = rt.ops.new_amax("cuda")
        rt.ops.conv3x3(xin, w, dev(b), out, rt.hip.EPI_BIAS_RELU, wt2=w2, amax_in=amax_in, amax_out=amax_out)
        assert out.border_is_zero()
        got = out.to_dense()
        assert float(amax_out.max()) == float(got.abs().max()), mode
        d = got.double().cpu() - ref
        errs[mode] = float((d ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
        assert float(d.abs().max()) <= 2e-6 * float(ref.abs().max()), mode
    # (<= 2x the fp32-MFMA kernel's rms error, or within 4e-7 relative rms - the class of an fp32 chain of this length: since
    # round 4 small launches of the fp32 kernel are no longer K-split, and an unsplit K = 576 chain of its exact 2-deep MFMA
    # steps comes out at 1.5e-7, below what a 16-bit split reaches)
    assert errs["split2"] <= max(2.0 * errs["f32"], 4e-7) + 1e-9, errs
    # a loose upper bound of the input maximum (a max-pool hands its input's bound through) gives the same class
    monkeypatch.setattr(rt.ops, "CONV_MODE", "split2")
    out = rt.FMap(cout, H, W)
    rt.ops.conv3x3(xin, w, dev(b), out, rt.hip.EPI_BIAS_RELU, wt2=w2, amax_in=rt.ops.new_amax("cuda", float(x.abs().max()) * 7.3),
                   amax_out=rt.ops.new_amax("cuda"))
    d = out.to_dense().double().cpu() - ref
    assert float((d ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()) <= 2.0 * errs["f32"] + 1e-9


@pytest.mark.parametrize("cin,cout,H,W", [(128, 128, 13, 131), (256, 512, 6, 9), (64, 64, 17, 21), (64, 128, 20, 28)])
def test_conv3x3_split2_dgrad_with_gate_and_add(rt, cin, cout, H, W, monkeypatch):
    monkeypatch.setattr(rt.ops, "CONV_MODE", "split2")
    torch.manual_seed(cin * 3 + W)
    x = F.relu(torch.randn(1, cin, H, W)).requires_grad_(True)
    wgt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    dy = torch.randn(1, cout, H, W) * 1e-5
    addend = torch.randn(cin, H, W) * 1e-5
    F.conv2d(x, wgt, None, padding=1).backward(dy)
    gate = (x.detach()[0] > 0).float()
    wd = dev(rt.ops.pack_conv_dgrad(wgt))
    wd2 = rt.ops.pack_conv_split2(wd)
    dyf = rt.FMap(cout, H, W).from_dense(dy[0])
    act = rt.FMap(cin, H, W).from_dense(x.detach()[0])
    scale = float(x.grad.abs().max())
    kw = dict(wt2=wd2, amax_in=rt.ops.new_amax("cuda", float(dy.abs().max())))
    out = rt.FMap(cin, H, W)
    rt.ops.conv3x3(dyf, wd, None, out, 0, amax_out=rt.ops.new_amax("cuda"), **kw)
    assert_close(out.to_dense(), x.grad[0], 1e-4, 1e-4 * scale, "plain dgrad")
    rt.ops.conv3x3(dyf, wd, None, out, rt.hip.EPI_RELU_MASK, gate=act, amax_out=rt.ops.new_amax("cuda"), **kw)
    assert_close(out.to_dense(), x.grad[0] * gate, 1e-4, 1e-4 * scale, "gated dgrad")
    out.from_dense(addend)
    am = rt.ops.new_amax("cuda")
    rt.ops.conv3x3(dyf, wd, None, out, rt.hip.EPI_RELU_MASK | rt.hip.EPI_ADD, gate=act, amax_out=am, **kw)
    assert_close(out.to_dense(), (x.grad[0] + addend) * gate, 1e-4, 1e-4 * scale, "gated dgrad + add")
    assert out.border_is_zero() and float(am.max()) == float(out.to_dense().abs().max())   # the bound covers the addend
    # stale values far above the recorded bound (a skipped tile of an earlier view) stay finite: clamped, not inf
    dyf.planes[:, dyf.Wp + 3] = 1e3
    rt.ops.conv3x3(dyf, wd, None, out, 0, amax_out=rt.ops.new_amax("cuda"), **kw)
    assert torch.isfinite(out.to_dense()).all()
    am = rt.ops.new_amax("cuda")
    rt.ops.fmap_amax(dyf, am)
    assert float(am.max()) == 1e3


@pytest.mark.parametrize("C,cout,H,W", [(64, 64, 37, 50), (128, 128, 40, 53), (256, 256, 21, 30), (64, 64, 150, 201)])
def test_conv3x3_split2_fused_pool_backward(rt, C, cout, H, W, monkeypatch):
    """The data-gradient conv below a max-pool with the pool's backward taken on the fly from the forward's argmax codes
    (sm_conv_problem::unpool_code) against the two-pass form (sm_maxpool2x2_bwd_relu, then the conv): identical results -
    including exact ties (constant regions: the first maximum wins), closed ReLU gates and odd sizes."""
    monkeypatch.setattr(rt.ops, "CONV_MODE", "split2")
    torch.manual_seed(C + W)
    act = F.relu(torch.randn(C, H, W))
    act[:, 4:12, 6:20] = 0.75                      # exact 4-way ties
    act[:, 14:18, :] = 0.0                         # windows whose maximum is 0: no gradient
    below = F.relu(torch.randn(cout, H, W))        # forward activation of the conv's input layer (its ReLU gate)
    dpooled = torch.randn(C, H // 2, W // 2) * 1e-4
    wgt = torch.randn(C, cout, 3, 3) * (2.0 / (9 * C)) ** 0.5     # conv below the pool: cout -> C channels
    wd = dev(rt.ops.pack_conv_dgrad(wgt))
    wd2 = rt.ops.pack_conv_split2(wd)
    a = rt.FMap(C, H, W).from_dense(act)
    gate = rt.FMap(cout, H, W).from_dense(below)
    pooled, dp = rt.FMap(C, H // 2, W // 2), rt.FMap(C, H // 2, W // 2).from_dense(dpooled)
    code = torch.zeros(C // 8 * pooled.plane, dtype=torch.int32, device="cuda")
    rt.ops.maxpool_fwd_grouped([(a, pooled)], None, [code])
    assert_close(pooled.to_dense(), F.max_pool2d(act[None], 2)[0], 0, 0)
    amax_in = rt.ops.new_amax("cuda", float(dpooled.abs().max()))
    # two passes
    da = rt.FMap(C, H, W)
    rt.ops.maxpool_bwd_relu(a, pooled, dp, da)
    ref = rt.FMap(cout, H, W)
    addend = torch.randn(cout, H, W) * 1e-4
    for flags, name in ((rt.hip.EPI_RELU_MASK, "gated"), (rt.hip.EPI_RELU_MASK | rt.hip.EPI_ADD, "gated + add")):
        ref.from_dense(addend)
        out = rt.FMap(cout, H, W).from_dense(addend)
        am_ref, am = rt.ops.new_amax("cuda"), rt.ops.new_amax("cuda")
        rt.ops.conv3x3_grouped([(da, ref, gate)], wd, None, flags, wt2=wd2, amax_in=amax_in, amax_out=am_ref)
        rt.ops.conv3x3_grouped([(dp, out, gate, code)], wd, None, flags, wt2=wd2, amax_in=amax_in, amax_out=am)
        assert torch.equal(out.to_dense(), ref.to_dense()), name
        assert out.border_is_zero() and float(am.max()) == float(am_ref.max())
    # and the two-pass form is the reference's max_pool2d backward through the ReLU
    x = act.clone().requires_grad_(True)
    F.max_pool2d(F.relu(x)[None], 2).backward(dpooled[None])
    assert_close(da.to_dense(), x.grad, 0, 0)


@pytest.mark.parametrize("mode", ["split2"])
@pytest.mark.parametrize("cin,cout,H,W", [(128, 128, 60, 70), (512, 512, 33, 45), (256, 64, 40, 52), (512, 512, 16, 21)])
def test_conv3x3_split_tail_units_deterministic(rt, cin, cout, H, W, mode, monkeypatch):
    """Fewer tiles than CUs: every tile is a K-split tail reduced by the second pass. Repeated launches into the same
    workspace must be bit-identical (fixed summation order), match an fp64 convolution and record the true output
    maximum."""
    monkeypatch.setattr(rt.ops, "CONV_MODE", mode)
    torch.manual_seed(cin + H)
    x = F.relu(torch.randn(1, cin, H, W) * 2)
    wgt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout) * 0.3
    ref = F.relu(F.conv2d(x.double(), wgt.double(), b.double(), padding=1))[0]
    xin = rt.FMap(cin, H, W).from_dense(x[0])
    w = dev(rt.ops.pack_conv_fwd(wgt))
    w2 = rt.ops.pack_conv_split2(w)
    first = None
    for rep in range(25):
        out = rt.FMap(cout, H, W)
        amax_in, amax_out = rt.ops.new_amax("cuda", float(x.abs().max())), rt.ops.new_amax("cuda")
        rt.ops.conv3x3(xin, w, dev(b), out, rt.hip.EPI_BIAS_RELU, wt2=w2, amax_in=amax_in, amax_out=amax_out)
        got = out.to_dense()
        if first is None:
            first = got
            d = got.double().cpu() - ref
            assert float(d.abs().max()) <= 2e-6 * float(ref.abs().max())
            assert float(amax_out.max()) == float(got.abs().max())
        else:
            assert torch.equal(got, first), rep


def test_conv3x3_first_layer_dgrad(rt):
    torch.manual_seed(5)
    H, W = 23, 37
    x = torch.randn(1, 3, H, W, requires_grad=True)
    wgt = torch.randn(64, 3, 3, 3) * 0.3
    dy = torch.randn(1, 64, H, W)
    F.conv2d(x, wgt, None, padding=1).backward(dy)
    out = rt.FMap(3, H, W)
    rt.ops.conv3x3_dgrad_c3(rt.FMap(64, H, W).from_dense(dy[0]), dev(rt.ops.pack_conv_dgrad(wgt)), out)
    assert_close(out.to_dense(), x.grad[0], 1e-4, 1e-5 * float(x.grad.abs().max()))
    assert out.border_is_zero()


@pytest.mark.parametrize("H,W", [(8, 12), (9, 13), (17, 21), (2, 2)])
def test_maxpool_forward_backward(rt, H, W):
    torch.manual_seed(H * W)
    C = 5
    x = F.relu(torch.randn(1, C, H, W))
    x[0, 0, :2, :2] = 1.5   # a tie inside one window: first element (row-major) wins
    x = x.requires_grad_(True)
    y = F.max_pool2d(x, 2, 2)
    dy = torch.randn_like(y)
    y.backward(dy)
    ref_dx = x.grad[0] * (x.detach()[0] > 0)
    act = rt.FMap(C, H, W).from_dense(x.detach()[0])
    pooled = rt.FMap(C, H // 2, W // 2)
    rt.ops.maxpool_fwd(act, pooled)
    assert_close(pooled.to_dense(), y[0], 0, 0)
    assert pooled.border_is_zero()
    dact = rt.FMap(C, H, W)
    rt.ops.maxpool_bwd_relu(act, pooled, rt.FMap(C, H // 2, W // 2).from_dense(dy[0]), dact)
    assert_close(dact.to_dense(), ref_dx, 0, 0)
    assert dact.border_is_zero()


def test_plane_kernels_with_tile_lists(rt):
    """The `_tiles` forms of the pool kernels and of conv1_1's data gradient touch the listed blocks only: inside them
    the result equals the dense launch, outside the output keeps what it held (two levels in one launch)."""
    torch.manual_seed(21)
    C = 6
    sizes = [(40, 70), (24, 52)]
    acts = [rt.FMap(C, h, w).from_dense(F.relu(torch.randn(C, h, w))) for h, w in sizes]
    dys = [rt.FMap(C, h // 2, w // 2).from_dense(torch.randn(C, h // 2, w // 2)) for h, w in sizes]
    bp = rt.ops.plane_tile_positions(1)
    n_blocks = [(-(-(h // 2) * rt.hip.row_stride(w // 2) // bp)) for h, w in sizes]
    assert min(n_blocks) >= 2
    chosen = [(0, 0), (0, n_blocks[0] - 1), (1, 1)]                       # (problem, block of 256 pooled positions)
    tl = torch.tensor([(g << 24) | b for g, b in chosen], dtype=torch.int32).cuda()

    def run(tile_list):
        pooled = [rt.FMap(C, h // 2, w // 2) for h, w in sizes]
        dact = [rt.FMap(C, h, w) for h, w in sizes]
        for f in pooled + dact:
            f.planes.fill_(-7.0)
        rt.ops.maxpool_fwd_grouped(list(zip(acts, pooled)), tile_list)
        dense_pooled = [rt.FMap(C, h // 2, w // 2) for h, w in sizes]
        rt.ops.maxpool_fwd_grouped(list(zip(acts, dense_pooled)))
        rt.ops.maxpool_bwd_relu_grouped(list(zip(acts, dense_pooled, dys, dact)), tile_list)
        return pooled, dact
    (p_t, d_t), (p_d, d_d) = run(tl), run(None)
    for g, (h, w) in enumerate(sizes):
        Wpo = rt.hip.row_stride(w // 2)
        inside = torch.zeros(p_t[g].planes.shape[1], dtype=torch.bool)
        for gg, b in chosen:
            if gg == g:
                inside[Wpo + b * bp: Wpo + (b + 1) * bp] = True
        inside[(h // 2 + 1) * Wpo:] = False
        pt, pd = p_t[g].planes.cpu(), p_d[g].planes.cpu()
        assert torch.equal(pt[:, inside], pd[:, inside]) and bool((pt[:, ~inside] == -7.0).all())
        # backward: the windows of the listed pooled positions
        qs = torch.nonzero(inside).flatten()
        yo, xo = qs // Wpo - 1, qs % Wpo - 1
        ok = (xo >= 0) & (xo < w // 2)
        win = torch.zeros(h, w, dtype=torch.bool)
        for dy_ in (0, 1):
            for dx_ in (0, 1):
                win[2 * yo[ok] + dy_, 2 * xo[ok] + dx_] = True
        dt, dd = d_t[g].to_dense().cpu(), d_d[g].to_dense().cpu()
        assert torch.equal(dt[:, win], dd[:, win]) and bool((dt[:, ~win] == -7.0).all())
    # conv1_1 data gradient
    dz = [rt.FMap(64, h, w).from_dense(torch.randn(64, h, w)) for h, w in sizes]
    wd = dev(rt.ops.pack_conv_dgrad(torch.randn(64, 3, 3, 3) * 0.2))
    b0 = rt.ops.plane_tile_positions(0)
    tl0 = torch.tensor([(0 << 24) | 1, (1 << 24) | 0], dtype=torch.int32).cuda()
    outs = []
    for tile_list in (tl0, None):
        o = [rt.FMap(4, h, w) for h, w in sizes]
        for f in o:
            f.planes.fill_(-7.0)
        rt.ops.conv3x3_dgrad_c3_grouped(list(zip(dz, o)), wd, tile_list)
        outs.append(o)
    for g, (h, w) in enumerate(sizes):
        Wp = rt.hip.row_stride(w)
        inside = torch.zeros(outs[0][g].planes.shape[1], dtype=torch.bool)
        b = 1 if g == 0 else 0
        inside[Wp + b * b0: min(Wp + (b + 1) * b0, (h + 1) * Wp)] = True
        a, d = outs[0][g].planes[:3].cpu(), outs[1][g].planes[:3].cpu()
        assert torch.equal(a[:, inside], d[:, inside]) and bool((a[:, ~inside] == -7.0).all())


def test_touch_flags_cover_the_scatter(rt):
    """Every chunk of the gradient arena that the scatter writes is flagged (the flags are a superset)."""
    torch.manual_seed(11)
    sizes = [(3, 64, 64), (3, 32, 32)]
    arena = torch.zeros(sum(c * h * w for c, h, w in sizes)).cuda()
    layers, o = [], 0
    for c, h, w in sizes:
        layers.append(arena[o:o + c * h * w].view(c, h, w)); o += c * h * w
    H, W = 23, 31
    grid = (torch.rand(1, H, W, 2) * 0.9 - 0.3).cuda()      # a corner of the texture, incl. border hits
    pw = (torch.rand(H, W) > 0.5).float().cuda()
    gimg = rt.FMap(3, H, W).from_dense(torch.randn(3, H, W))
    rt.ops.tex_sample_bwd(layers, grid, gimg, pw)
    chunk_log2 = 4
    flags = torch.zeros((arena.numel() + 15) // 16, dtype=torch.int32).cuda()
    rt.ops.tex_touch_flags(layers, arena, grid, pw, flags, chunk_log2)
    written = (arena.view(-1, 16) != 0).any(1)
    assert bool((flags.bool() | ~written).all())
    assert 0 < int(flags.sum()) < flags.numel()             # and it is not trivially everything


# ------------------------------------------------------------------ K5 / K6
@pytest.mark.parametrize("gram_mode", ["f32", "split2"])
@pytest.mark.parametrize("C,H,W,multi", [(64, 20, 28, True), (128, 10, 14, True), (256, 5, 7, False), (64, 50, 70, False),
                                           (64, 150, 200, True), (512, 12, 17, True), (128, 40, 56, True),
                                           (256, 33, 41, True)])
def test_gram_style_loss_and_backward(rt, C, H, W, multi, gram_mode, monkeypatch):
    monkeypatch.setattr(rt.ops, "GRAM_MODE", gram_mode)
    torch.manual_seed(C + H)
    feat = F.relu(torch.randn(1, C, H, W)).requires_grad_(True)
    m_all = (torch.rand(1, 1, H, W) > 0.3).float()
    passed = torch.rand(1, 1, H, W) > 0.5
    m_pass, m_fail = m_all * passed, m_all * (~passed)
    Y2 = torch.randn(C, C); Y2 = (Y2 + Y2.T) / 2
    Y0 = torch.randn(C, C); Y0 = (Y0 + Y0.T) / 2
    weight, factor = 1e-4 * 1000.0, 0.37
    mse = torch.nn.MSELoss()
    if multi:
        gp = O.gram_matrix(O.masked_features(feat, m_pass))
        gf = O.gram_matrix(O.masked_features(feat, m_fail))
        loss = weight * factor * (mse(Y2[None], gp) + mse(Y2[None], gf) + mse(Y0[None], gp))
        masks, targets, term_mask, skip = (m_pass, m_fail), [Y2, Y2, Y0], [0, 1, 0], [0, 1]
    else:
        gp = O.gram_matrix(O.masked_features(feat, m_all))
        loss = weight * factor * mse(Y0[None], gp)
        masks, targets, term_mask, skip = (m_all, None), [Y0], [0], [0, 0]
    loss.backward()
    f = rt.FMap(C, H, W).from_dense(feat.detach()[0])
    mk = [rt.FMap(1, H, W).from_dense(m[0]) if m is not None else None for m in masks]
    ns, na = rt.ops.gram_num_slabs(C, H, W), rt.ops.gram_workspace_slabs(C, H, W)
    S = [torch.full((na, C, C), 7.0).cuda(), torch.full((na, C, C), 7.0).cuda() if multi else None]   # no pre-zeroing needed
    af = rt.ops.new_amax("cuda", float(feat.detach().abs().max()))     # the bound the producing conv records (fp16x2 mode)
    assert rt.ops.gram_masked(f, mk[0], mk[1], S[0], S[1], amax_feat=af) == ns
    n0 = float(masks[0].sum())
    ref_S0 = gp[0] * n0
    T = C // 64
    tile_upper = torch.ones(T, T).triu().repeat_interleave(64, 0).repeat_interleave(64, 1).bool()
    assert_close(S[0][:ns].sum(0).cpu()[tile_upper], ref_S0.detach()[tile_upper], 1e-4, 1e-4 * float(ref_S0.abs().max()))
    if gram_mode != "f32":
        # accumulate variant: adds into caller-zeroed slabs (the engine zeroes every slab of a step with one fill)
        A = [torch.zeros(na, C, C).cuda(), torch.zeros(na, C, C).cuda() if multi else None]
        for rep in (1, 2):
            rt.ops.gram_masked(f, mk[0], mk[1], A[0], A[1], prezeroed=True, amax_feat=af)
            assert_close(A[0][0].cpu()[tile_upper], rep * S[0][:ns].sum(0).cpu()[tile_upper], 1e-5,
                         1e-5 * float(ref_S0.abs().max()))
    counts = dev(torch.tensor([float(m.sum()) if m is not None else 0.0 for m in masks]))
    D = [torch.empty(C, C).cuda(), torch.empty(C, C).cuda() if multi else None]
    loss_out = torch.zeros(1).cuda()
    ad = rt.ops.new_amax("cuda")
    rt.ops.style_loss(S[0], S[1], counts, dev(torch.tensor([factor])), [dev(t) for t in targets], term_mask, skip,
                      weight, C, D[0], D[1], loss_out, n_slabs=ns, amax_d_out=ad)
    assert_close(loss_out, loss.detach().reshape(1), 1e-4, 0)
    assert float(ad.max()) == max(float(d.abs().max()) for d in D if d is not None)    # bound of the derivative matrices
    df = rt.FMap(C, H, W)
    rt.ops.gram_backward(f, mk[0], mk[1], D[0], D[1], df, relu_gate=False, amax_feat=af, amax_d=ad)
    assert_close(df.to_dense(), feat.grad[0], 1e-4, 2e-5 * float(feat.grad.abs().max()))
    assert df.border_is_zero()
    rt.ops.gram_backward(f, mk[0], mk[1], D[0], D[1], df, relu_gate=True, amax_feat=af, amax_d=ad)
    assert_close(df.to_dense(), feat.grad[0] * (feat.detach()[0] > 0), 1e-4, 2e-5 * float(feat.grad.abs().max()))


def test_gram_grouped_matches_fp64(rt):
    """All (level, layer) Gram problems of a step in one grouped call (two launches: 64- and 128-channel tile classes)
    against fp64 masked Grams; one of the problems has a single mask, one an empty mask."""
    shapes = [(64, 150, 200), (64, 37, 50), (128, 75, 100), (256, 37, 50), (512, 18, 25), (512, 9, 12), (128, 20, 28)]
    probs, keep, refs = [], [], []
    for n, (C, H, W) in enumerate(shapes):
        torch.manual_seed(C + H)
        feat = F.relu(torch.randn(C, H, W)) * (10.0 ** (n - 3))          # very different magnitudes per problem
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        m_all = ((yy > 0.1 * H) & (xx < 0.8 * W) & (torch.rand(H, W) > 0.05)).float()
        passed = (xx < 0.4 * W).float()
        m0, m1 = m_all * passed, m_all * (1 - passed)
        if n == 3: m1 = None
        if n == 4: m1 = torch.zeros(H, W)
        f = rt.FMap(C, H, W).from_dense(feat)
        mk = [rt.FMap(1, H, W).from_dense(m[None]) if m is not None else None for m in (m0, m1)]
        S = [torch.zeros(C, C).cuda(), torch.zeros(C, C).cuda() if m1 is not None else None]
        af = rt.ops.new_amax("cuda", float(feat.abs().max()))
        keep.append((f, mk, S, af))
        probs.append(rt.ops.gram_problem(f, mk[0], mk[1], S[0], S[1], af))
        fd = feat.double().reshape(C, -1)
        refs.append([(fd * m.double().reshape(1, -1)) @ fd.T if m is not None else None for m in (m0, m1)])
    rt.ops.gram_masked_grouped(rt.ops.gram_problem_array(probs))
    for (C, H, W), (f, mk, S, af), ref in zip(shapes, keep, refs):
        T = C // 64
        tile_upper = torch.ones(T, T).triu().repeat_interleave(64, 0).repeat_interleave(64, 1).bool()
        for k in range(2):
            if ref[k] is None: continue
            scale = max(float(ref[0].abs().max()), 1e-30)
            assert_close(S[k].cpu().double()[tile_upper], ref[k][tile_upper], 1e-5, 2e-6 * scale, f"C={C} {H}x{W} mask {k}")


def test_loss_phase_grouped_matches_per_problem_calls(rt, monkeypatch):
    """sm_style_loss_grouped / sm_gram_backward_split2_grouped over several (level, layer) problems against the
    per-problem entry points on the same inputs: loss value, derivative matrices, their recorded bound, dF."""
    monkeypatch.setattr(rt.ops, "GRAM_MODE", "split2")
    shapes = [(64, 60, 80, True), (128, 30, 40, True), (256, 15, 20, False), (512, 9, 12, True), (64, 21, 30, True)]
    keep, fwd, sty, bwd = [], [], [], []
    loss_ref = torch.zeros(1).cuda()
    for n, (C, H, W, multi) in enumerate(shapes):
        torch.manual_seed(C + H)
        feat = F.relu(torch.randn(C, H, W)) * (3.0 ** n)
        m_all = (torch.rand(H, W) > 0.3).float()
        passed = (torch.rand(H, W) > 0.5).float()
        masks = (m_all * passed, m_all * (1 - passed)) if multi else (m_all, None)
        f = rt.FMap(C, H, W).from_dense(feat)
        mk = [rt.FMap(1, H, W).from_dense(m[None]) if m is not None else None for m in masks]
        Y = [dev((lambda t: (t + t.T) / 2)(torch.randn(C, C))) for _ in range(2)]
        targets, term_mask, skip = ([Y[0], Y[0], Y[1]], [0, 1, 0], [0, 1]) if multi else ([Y[0]], [0], [0, 0])
        counts = dev(torch.tensor([float(m.sum()) if m is not None else 0.0 for m in masks]))
        factor = dev(torch.tensor([0.3 + 0.1 * n]))
        weight = 0.1 * (n + 1)
        af = rt.ops.new_amax("cuda", float(feat.abs().max()))
        # per-problem reference
        S = [torch.zeros(1, C, C).cuda(), torch.zeros(1, C, C).cuda() if multi else None]
        rt.ops.gram_masked(f, mk[0], mk[1], S[0], S[1], amax_feat=af)
        Dr = [torch.empty(C, C).cuda(), torch.empty(C, C).cuda() if multi else None]
        adr = rt.ops.new_amax("cuda")
        rt.ops.style_loss(S[0], S[1], counts, factor, targets, term_mask, skip, weight, C, Dr[0], Dr[1], loss_ref, amax_d_out=adr)
        dfr = rt.FMap(C, H, W)
        gate = n == 3
        rt.ops.gram_backward(f, mk[0], mk[1], Dr[0], Dr[1], dfr, relu_gate=gate, amax_feat=af, amax_d=adr)
        # grouped problem entries (own outputs)
        G = [torch.zeros(C, C).cuda(), torch.zeros(C, C).cuda() if multi else None]
        D = [torch.empty(C, C).cuda(), torch.empty(C, C).cuda() if multi else None]
        ad = rt.ops.new_amax("cuda")
        df = rt.FMap(C, H, W)
        ws = torch.empty(rt.ops.gram_backward_ws_bytes(C), dtype=torch.uint8, device="cuda")
        fwd.append(rt.ops.gram_problem(f, mk[0], mk[1], G[0], G[1], af))
        sty.append(rt.ops.style_problem(G[0], G[1], counts, factor, targets, term_mask, skip, weight, C, D[0], D[1], ad))
        bwd.append(rt.ops.gram_bwd_problem(f, mk[0], mk[1], D[0], D[1], df, ws, af, ad, relu_gate=gate))
        keep.append((f, mk, Y, counts, factor, af, S, Dr, adr, dfr, G, D, ad, df, ws))
    loss = torch.zeros(1).cuda()
    rt.ops.gram_masked_grouped(rt.ops.struct_array(rt.hip.GramProblem, fwd))
    rt.ops.style_loss_grouped(rt.ops.struct_array(rt.hip.StyleProblem, sty), loss)
    rt.ops.gram_backward_grouped(rt.ops.struct_array(rt.hip.GramBwdProblem, bwd))
    assert_close(loss, loss_ref, 1e-5, 0)
    for (C, H, W, multi), k in zip(shapes, keep):
        f, mk, Y, counts, factor, af, S, Dr, adr, dfr, G, D, ad, df, ws = k
        for a, b in zip(D, Dr):
            if a is not None:
                assert_close(a, b, 1e-4, 1e-5 * float(b.abs().max()), f"D C={C}")
        assert abs(float(ad.max()) - float(adr.max())) <= 1e-4 * float(adr.max())
        assert_close(df.to_dense(), dfr.to_dense(), 1e-4, 2e-5 * float(dfr.to_dense().abs().max()), f"dF C={C}")
        assert df.border_is_zero()


@pytest.mark.parametrize("gram_mode", ["f32", "split2"])
def test_style_loss_empty_masks(rt, gram_mode, monkeypatch):
    """N_pass == 0 -> Gram of zeros still compared with the target; N_fail == 0 -> term dropped (:332)."""
    monkeypatch.setattr(rt.ops, "GRAM_MODE", gram_mode)
    C, H, W = 64, 6, 6
    feat = torch.rand(1, C, H, W)
    Y = torch.randn(C, C); Y = (Y + Y.T) / 2
    f = rt.FMap(C, H, W).from_dense(feat[0])
    zero = rt.FMap(1, H, W)
    ns, na = rt.ops.gram_num_slabs(C, H, W), rt.ops.gram_workspace_slabs(C, H, W)
    S = [torch.zeros(na, C, C).cuda(), torch.zeros(na, C, C).cuda()]
    rt.ops.gram_masked(f, zero, zero, S[0], S[1], amax_feat=rt.ops.new_amax("cuda", float(feat.max())))
    D = [torch.ones(C, C).cuda(), torch.ones(C, C).cuda()]
    loss_out = torch.zeros(1).cuda()
    rt.ops.style_loss(S[0], S[1], dev(torch.zeros(2)), dev(torch.tensor([0.5])), [dev(Y), dev(Y)], [0, 1], [0, 1],
                      2.0, C, D[0], D[1], loss_out, n_slabs=ns)
    assert_close(loss_out, (2.0 * 0.5 * (Y ** 2).mean()).reshape(1), 1e-5, 0)
    assert float(D[0].abs().max()) == 0 and float(D[1].abs().max()) == 0


def test_style_loss_average_mode(rt):
    torch.manual_seed(3)
    C = 64
    Y = torch.randn(C, C); Y = (Y + Y.T) / 2
    hist = torch.zeros(9, C, C).cuda()
    cache = []
    for step in range(12):
        G = torch.randn(C, C); G = (G + G.T) / 2
        Gt = G.clone().requires_grad_(True)
        c = [g.detach() for g in cache[:9]]
        c.insert(0, Gt)
        cache = c
        loss = 3.0 * 0.25 * F.mse_loss(Y, torch.stack(c).mean(0))
        loss.backward()
        N = 10.0
        D0 = torch.empty(C, C).cuda()
        loss_out = torch.zeros(1).cuda()
        rt.ops.style_loss(dev(G * N), None, dev(torch.tensor([N, 0.0])), dev(torch.tensor([0.25])), [dev(Y)], [0],
                          [0, 0], 3.0, C, D0, None, loss_out, history=hist, hist_len=min(step, 9), hist_slot=step % 9)
        assert_close(loss_out, loss.detach().reshape(1), 1e-5, 0)
        assert_close(D0, Gt.grad * 2 / N, 1e-4, 1e-7)   # D = 2 dL/dS = 2 dL/dG / N


def test_mse_masked(rt):
    torch.manual_seed(4)
    C, H, W = 512, 5, 7
    p = torch.randn(1, C, H, W, requires_grad=True)
    t = torch.randn(1, C, H, W)
    m = (torch.rand(1, 1, H, W) > 0.4).float()
    weight, factor = 70.0, 0.6
    loss = weight * factor * F.mse_loss(O.masked_features(t, m), O.masked_features(p, m))
    loss.backward()
    dp = rt.FMap(C, H, W)
    loss_out = torch.zeros(1).cuda()
    rt.ops.mse_masked(rt.FMap(C, H, W).from_dense(p.detach()[0]), rt.FMap(C, H, W).from_dense(t[0]),
                      rt.FMap(1, H, W).from_dense(m[0]), dev(m.sum().reshape(1)), dev(torch.tensor([factor])), weight,
                      dp, loss_out)
    assert_close(loss_out, loss.detach().reshape(1), 1e-5, 0)
    assert_close(dp.to_dense(), p.grad[0], 1e-5, 1e-6 * float(p.grad.abs().max()))
    loss_out.zero_()
    rt.ops.mse_masked(rt.FMap(C, H, W).from_dense(p.detach()[0]), rt.FMap(C, H, W).from_dense(t[0]),
                      rt.FMap(1, H, W), dev(torch.zeros(1)), dev(torch.tensor([factor])), weight, dp, loss_out)
    assert float(loss_out) == 0 and float(dp.to_dense().abs().max()) == 0


# ------------------------------------------------------------------ K7
def test_adam_fused_matches_oracle(rt):
    torch.manual_seed(7)
    sizes = [3 * 64 * 64, 3 * 32 * 32, 3 * 16 * 16, 3 * 8 * 8 + 5]   # last one not a multiple of 4*256
    seg_end = np.cumsum(sizes).tolist()
    n = seg_end[-1]
    p = (torch.randn(n) * 60).clamp(O.CLAMP_LO, O.CLAMP_HI)
    m, v = torch.zeros(n), torch.zeros(n)
    reg = [0.41, 0.2, 0.05, 0.0]
    regv = torch.cat([torch.full((s,), r) for s, r in zip(sizes, reg)])
    P, M, V = dev(p), dev(m), dev(v)
    for step in range(1, 5):
        g = torch.randn(n) * 10 ** torch.randint(-5, 2, (n,)).float()
        g[::7] = 0
        G = dev(g)
        lr = 1.0 if step < 3 else 0.1
        sumsq = torch.zeros(4).cuda()
        rt.ops.adam_fused(P, G, M, V, seg_end, reg, lr, step, grad_scale=0.5, sumsq_out=sumsq)
        p, m, v = O.adam_step_explicit(p, 0.5 * g + regv * p, m, v, step, lr)
        p = p.clamp(O.CLAMP_LO, O.CLAMP_HI)
        assert_close(P, p, 1e-5, 1e-5)
        assert_close(M, m, 1e-5, 1e-7)
        assert_close(V, v, 1e-5, 1e-9)
        assert float(G.abs().max()) == 0
        ref_sq = [float((p[a:b] ** 2).sum()) for a, b in zip([0] + seg_end[:-1], seg_end)]
        assert_close(sumsq, ref_sq, 1e-4, 0)
    big = dev(torch.tensor([500.0, -500.0, 1.0, 2.0, 3.0]))
    sq = torch.zeros(1).cuda()
    rt.ops.clamp_sumsq(big, [5], sq)
    assert_close(big, [O.CLAMP_HI, O.CLAMP_LO, 1, 2, 3], 1e-7, 0)


def test_adam_fused_by_ranges_equals_one_launch(rt):
    """The pipelined multi-GPU update runs the fused optimizer range by range (runtime/distributed.py): parameters and
    moments must be bit-identical to one launch over the arena, the per-layer sums of squares equal up to summation order."""
    torch.manual_seed(11)
    sizes = [3 * 64 * 64, 3 * 32 * 32, 3 * 16 * 16, 3 * 8 * 8]
    seg_end = np.cumsum(sizes).tolist()
    n = seg_end[-1]
    reg = [0.41, 0.2, 0.05, 0.0]
    p0 = (torch.randn(n) * 60).clamp(O.CLAMP_LO, O.CLAMP_HI)
    m0, v0, g0 = torch.randn(n) * 0.1, torch.rand(n) * 0.01, torch.randn(n)
    res = []
    for bounds in ([0, n], [0, 4096, 4096 + 64, seg_end[0] + 128, seg_end[2] - 64, n]):
        P, M, V, G = dev(p0), dev(m0), dev(v0), dev(g0)
        sumsq = torch.zeros(4).cuda()
        for lo, hi in zip(bounds, bounds[1:]):
            rt.ops.adam_fused(P, G, M, V, seg_end, reg, 0.5, 3, grad_scale=0.5, sumsq_out=sumsq, lo=lo, hi=hi)
        assert float(G.abs().max()) == 0
        res.append((P, M, V, sumsq))
    for a, b in zip(res[0][:3], res[1][:3]):
        assert torch.equal(a, b)
    assert_close(res[1][3], res[0][3], 1e-5, 0)


def test_adam_fused_without_gradient_pointer_equals_zero_gradient(rt):
    """g = NULL (the early half of the split update: chunks whose data-term gradient is known to be zero) gives the bits of
    a launch over an all-zero gradient arena, which it neither reads nor writes; flagged chunks only."""
    torch.manual_seed(4)
    n = 3 * 64 * 64 + 3 * 32 * 32 + 8
    seg_end = [3 * 64 * 64, n]
    reg = [0.3, 0.0]
    p0 = (torch.randn(n) * 60).clamp(O.CLAMP_LO, O.CLAMP_HI)
    m0, v0 = torch.randn(n) * 0.1, torch.rand(n) * 0.01
    flags = (torch.rand(-(-n // 64)) < 0.5).int().cuda()
    res = []
    for with_g in (True, False):
        P, M, V = dev(p0), dev(m0), dev(v0)
        G = torch.zeros(n).cuda() if with_g else None
        sumsq = torch.zeros(2).cuda()
        rt.ops.adam_fused(P, G, M, V, seg_end, reg, 1.0, 5, grad_scale=1.0, sumsq_out=sumsq, touched=flags, touched_log2=6)
        res.append((P, M, V, sumsq))
    for a, b in zip(res[0][:3], res[1][:3]):
        assert torch.equal(a, b)
    assert_close(res[1][3], res[0][3], 1e-5, 0)   # (block sums meet in atomics: order not fixed)
    assert not torch.equal(res[0][0], dev(p0))


# ------------------------------------------------------------------ per-view constants
def test_view_constants_match_oracle(rt):
    from golden_cases import SMALL_LEVEL_HW, SMALL_ROOM, SMALL_VIEW_HW
    from stylemesh_amd.data import synthetic as S
    batch = S.make_view(3, view_hw=SMALL_VIEW_HW, level_hw=SMALL_LEVEL_HW, level_heights=[40, 64],
                        min_pyramid_depth=0.9, room=S.BoxRoom(SMALL_ROOM))
    cfg = O.OracleConfig(use_angle_weight=True, use_depth_scaling=True, angle_threshold=30)
    masks, weights = O.level_masks_and_weights(batch, SMALL_LEVEL_HW, cfg)
    _, _, _, _, _, rounded, other, iw, _, _, mask, ag, adeg = batch
    h, w = SMALL_VIEW_HW
    E = torch.empty(2, h, w).cuda()
    Wt = torch.empty(2, h, w).cuda()
    rt.ops.level_masks(dev(rounded), dev(other), dev(iw), dev(mask.to(torch.uint8)), 2, E, Wt)
    for i, (H, W) in enumerate(SMALL_LEVEL_HW):
        M = torch.empty(H, W).cuda()
        pw = torch.empty(H, W).cuda()
        passed = torch.empty(H, W, dtype=torch.uint8).cuda()
        msum = torch.zeros(1).cuda()
        rt.ops.level_maps(E[i], Wt[i], dev(ag), dev(adeg), 30.0, h, w, H, W, M, pw, passed, msum)
        assert_close(M, masks[i][0, 0], 0, 0)
        assert float(msum) == float(masks[i].sum())
        ref_pw = F.interpolate(ag, (H, W), mode="bilinear")[0, 0] * weights[i][0, 0]
        assert_close(pw, ref_pw, 1e-5, 1e-6)
        ref_passed = F.interpolate(adeg, (H, W), mode="bilinear")[0, 0] < 30
        assert (passed.cpu().bool() != ref_passed).float().mean() < 1e-3
        for (hl, wl) in [(H, W), (H // 2, W // 2), (H // 8, W // 8)]:
            ma, mp, mf = rt.FMap(1, hl, wl), rt.FMap(1, hl, wl), rt.FMap(1, hl, wl)
            counts = torch.zeros(3).cuda()
            rt.ops.layer_masks(M, passed, H, W, hl, wl, ma, mp, mf, counts)
            pb = passed.cpu().bool()[None, None]
            r_all = F.interpolate(masks[i], (hl, wl), mode="nearest")
            r_pass = F.interpolate(masks[i] * pb, (hl, wl), mode="nearest")
            r_fail = F.interpolate(masks[i] * (~pb), (hl, wl), mode="nearest")
            assert_close(ma.to_dense(), r_all[0], 0, 0)
            assert_close(mp.to_dense(), r_pass[0], 0, 0)
            assert_close(mf.to_dense(), r_fail[0], 0, 0)
            assert_close(counts, [float(r_all.sum()), float(r_pass.sum()), float(r_fail.sum())], 0, 0)


def test_resizes_and_factors(rt):
    torch.manual_seed(9)
    x = torch.randn(1, 6, 5, 7)
    src = rt.FMap(6, 5, 7).from_dense(x[0])
    for (H, W) in [(8, 11), (5, 7), (13, 30), (3, 4)]:
        dst = rt.FMap(6, H, W)
        rt.ops.fmap_resize_bilinear(src, dst)
        assert_close(dst.to_dense(), F.interpolate(x, (H, W), mode="bilinear")[0], 1e-5, 1e-5)
        dst2 = rt.FMap(6, H, W)
        rt.ops.image_to_fmap(dev(x[0]), dst2)
        assert_close(dst2.to_dense(), F.interpolate(x, (H, W), mode="bilinear")[0], 1e-5, 1e-5)
        assert dst2.border_is_zero()
    assert_close(rt.ops.fmap_to_image(src), x[0], 0, 0)
    counts = [dev(torch.tensor([30.0])), dev(torch.tensor([10.0])), dev(torch.tensor([0.0]))]
    factors = [torch.zeros(1).cuda() for _ in range(3)]
    rt.ops.level_factors(counts, [60.0, 40.0, 20.0], factors)
    assert_close(torch.cat(factors), [0.5 / 0.75, 0.25 / 0.75, 0.0], 1e-6, 0)


# ------------------------------------------------------------------ dynamic-range stress of the fp16x2-split kernels
# (task 1). The split represents an operand x of a tensor with maximum A = max |x| as
#     x s = h + l,   h = fp16(x s),  l = fp16(x s - h),   s = the power of two with A s in [2^14, 2^15)
# so  |x - (h + l) / s| <= 2^-22 |x|  while l is a normal fp16 number, and <= 2^-25 / s <= 2^-39 A below that (l in the
# subnormal range: elements more than ~2^18 below the tensor's maximum). A product drops ll' (< 2^-22 of it), and the
# partial products are summed in fp32 (16 exact products per MFMA, n_acc = 3 K / 16 sequential accumulations). The bound
# the tests assert ELEMENTWISE against an fp64 result, with a factor 2 of margin on each term:
#     |err| <= (2^-21 + 4 sqrt(n_acc) 2^-24) sum |x||w|  +  2^-38 (A_x sum |w| + A_w sum |x|)
# i.e. fp32-class relative accuracy for the terms within 2^18 of the tensor maximum and an ABSOLUTE floor of 2^-38
# max|x| per unit weight below that - a single 10^6 x outlier lowers the accuracy of the ordinary elements to ~2^-18
# relative, it cannot produce garbage. The fp32-MFMA kernel is run on the same inputs for comparison (recorded).
def _stress_input(kind, shape, gen):
    if kind == "loguniform":    # magnitudes log-uniform over 2^-30 .. 1 of the maximum, random signs
        mag = torch.exp2(-30.0 * torch.rand(shape, generator=gen)) * 50.0
        sign = torch.where(torch.rand(shape, generator=gen) < 0.5, -1.0, 1.0)
        x = mag * sign
        x.view(-1)[int(torch.randint(0, x.numel(), (1,), generator=gen))] = 50.0
        return x
    assert kind == "outlier"    # ReLU'd normal values with ONE element 10^6 times the rest's maximum
    x = F.relu(torch.randn(shape, generator=gen))
    x.view(-1)[int(torch.randint(0, x.numel(), (1,), generator=gen))] = float(x.max()) * 1e6
    return x


def _split2_bound(abs_prod, amax_x, sum_w, amax_w, sum_x, K):
    n_acc = 3.0 * K / 16.0
    return (2.0 ** -21 + 4.0 * n_acc ** 0.5 * 2.0 ** -24) * abs_prod + 2.0 ** -38 * (amax_x * sum_w + amax_w * sum_x)


@pytest.mark.parametrize("kind", ["loguniform", "outlier"])
@pytest.mark.parametrize("cin,cout,H,W", [(64, 64, 40, 52), (128, 256, 33, 47), (512, 512, 12, 17)])
def test_conv3x3_split2_dynamic_range_stress(rt, cin, cout, H, W, kind, monkeypatch):
    gen = torch.Generator().manual_seed(cin + cout + H + len(kind))
    x = _stress_input(kind, (1, cin, H, W), gen)
    wgt = torch.randn(cout, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5
    xd, wd_ = x.double(), wgt.double()
    ref = F.conv2d(xd, wd_, None, padding=1)[0]
    abs_prod = F.conv2d(xd.abs(), wd_.abs(), None, padding=1)[0]
    sum_w = wd_.abs().sum((1, 2, 3)).view(-1, 1, 1)
    sum_x = F.conv2d(xd.abs().sum(1, keepdim=True), torch.ones(1, 1, 3, 3, dtype=torch.float64), None, padding=1)[0]
    amax_x, amax_w = float(x.abs().max()), float(wgt.abs().max())
    bound = _split2_bound(abs_prod, amax_x, sum_w, amax_w, sum_x, 9 * cin)
    xin = rt.FMap(cin, H, W).from_dense(x[0])
    w = dev(rt.ops.pack_conv_fwd(wgt))
    w2 = rt.ops.pack_conv_split2(w)
    worst = {}
    for mode in ("f32", "split2"):
        monkeypatch.setattr(rt.ops, "CONV_MODE", mode)
        out = rt.FMap(cout, H, W)
        rt.ops.conv3x3(xin, w, None, out, 0, wt2=w2, amax_in=rt.ops.new_amax("cuda", amax_x), amax_out=rt.ops.new_amax("cuda"))
        got = out.to_dense().double().cpu()
        assert torch.isfinite(got).all()
        err = (got - ref).abs()
        worst[mode] = float((err / bound).max())
        worst[mode + "_vs_amax"] = float(err.max()) / (amax_x * float(sum_w.max()))
    print(f"\n[conv stress {kind} {cin}->{cout}] max err / bound: {worst}")
    assert worst["split2"] <= 1.0, worst
    # relative to the tensor maximum (the statement of DESIGN.md section 2): |err| <= 2^-21 A sum|w| always
    assert worst["split2_vs_amax"] <= 2.0 ** -21, worst


@pytest.mark.parametrize("kind", ["loguniform", "outlier"])
@pytest.mark.parametrize("C,H,W", [(64, 60, 81), (128, 40, 56), (512, 12, 17)])
def test_gram_split2_dynamic_range_stress(rt, C, H, W, kind, monkeypatch):
    """The masked Gram contraction and its backward GEMM (fp16x2 operands: the feature map for both sides of the
    forward; the derivative matrix and the feature map for the backward) on heavy-tailed feature maps, against fp64."""
    monkeypatch.setattr(rt.ops, "GRAM_MODE", "split2")
    gen = torch.Generator().manual_seed(C + H + len(kind))
    feat = _stress_input(kind, (C, H, W), gen)
    m0 = (torch.rand(H, W, generator=gen) > 0.3).float()
    fd = feat.double().reshape(C, -1)
    md = m0.double().reshape(1, -1)
    A = float(feat.abs().max())
    f = rt.FMap(C, H, W).from_dense(feat)
    mk = rt.FMap(1, H, W).from_dense(m0[None])
    af = rt.ops.new_amax("cuda", A)
    # forward: S = (F m) F^T
    S = torch.zeros(rt.ops.gram_workspace_slabs(C, H, W), C, C).cuda()
    n = rt.ops.gram_masked(f, mk, None, S, None, amax_feat=af)
    got = S[:n].sum(0).double().cpu()
    ref = (fd * md) @ fd.T
    abs_prod = (fd.abs() * md) @ fd.abs().T
    row = (fd.abs() * md).sum(1)
    bound = _split2_bound(abs_prod, A, row.view(-1, 1), A, row.view(1, -1), H * W)
    T = C // 64
    upper = torch.ones(T, T).triu().repeat_interleave(64, 0).repeat_interleave(64, 1).bool()
    assert torch.isfinite(got[upper]).all()
    worst_f = float(((got - ref).abs() / bound)[upper].max())
    # backward: dF = m (D F), D symmetric with its own recorded bound
    D = torch.randn(C, C, generator=gen) * 1e-3
    D = (D + D.T) / 2
    ad = rt.ops.new_amax("cuda", float(D.abs().max()))
    df = rt.FMap(C, H, W)
    rt.ops.gram_backward(f, mk, None, dev(D), None, df, relu_gate=False, amax_feat=af, amax_d=ad)
    gotb = df.to_dense().double().cpu().reshape(C, -1)
    refb = (D.double() @ fd) * md
    abs_b = (D.double().abs() @ fd.abs()) * md
    boundb = _split2_bound(abs_b, A, D.double().abs().sum(1).view(-1, 1) * md, float(D.abs().max()),
                           fd.abs().sum(0).view(1, -1) * md, C)
    assert torch.isfinite(gotb).all()
    live = (md > 0).expand_as(gotb)
    worst_b = float(((gotb - refb).abs()[live] / boundb[live].clamp_min(1e-300)).max())
    assert float(gotb[~live].abs().max()) == 0.0
    print(f"\n[gram stress {kind} C={C}] max err / bound: forward {worst_f:.3f}, backward {worst_b:.3f}")
    assert worst_f <= 1.0 and worst_b <= 1.0, (worst_f, worst_b)


@pytest.mark.parametrize("density", [0.0, 0.02, 0.15, 0.6, 1.0])
@pytest.mark.parametrize("sizes,with_g", [([3 * 256 * 256, 3 * 128 * 128, 3 * 64 * 64, 3 * 32 * 32], True),
                                           ([3 * 64 * 64, 3 * 32 * 32, 3 * 16 * 16, 3 * 8 * 8], True),     # segments start on chunk 3 mod 4
                                           ([3 * 512 * 512 + 64 * 7 + 20], False),                          # one layer, partial last chunk
                                           ([3 * 16 * 16, 3 * 4 * 4, 3 * 2 * 2], True)])                    # unaligned segments: the tile walk
def test_adam_flagged_span_kernel_equals_the_elementwise_update(rt, sizes, with_g, density):
    """sm_adam_fused over FLAGGED 256-byte chunks (round 6: a block compacts the flags of a span of 1024 chunks of one
    layer and streams the listed chunks) against the same update computed element by element in torch on the flagged chunks:
    p, m, v within fp32 rounding of the reference formula (model/model.py:387-395 + the regulariser gradient,
    texture.py:102-108, + the clamp, texture.py:41-44), unflagged chunks untouched bit for bit, the gradient zeroed on the
    flagged chunks only, sum(p^2) per layer over the WHOLE flagged set."""
    torch.manual_seed(len(sizes) * 7 + int(density * 100))
    seg_end = np.cumsum(sizes).tolist()
    n = seg_end[-1]
    reg = [0.41, 0.2, 0.05, 0.0][:len(sizes)]
    p0 = (torch.randn(n) * 60).clamp(O.CLAMP_LO, O.CLAMP_HI)
    m0, v0, g0 = torch.randn(n) * 0.1, torch.rand(n) * 0.01, torch.randn(n)
    n_chunks = -(-n // 64)
    flags = (torch.rand(n_chunks) < density).int()
    if density >= 1.0:
        flags[:] = 1
    P, M, V = dev(p0), dev(m0), dev(v0)
    G = dev(g0) if with_g else None
    sumsq = torch.zeros(len(sizes)).cuda()
    lr, step, gs = 0.5, 3, 0.5
    rt.ops.adam_fused(P, G, M, V, seg_end, reg, lr, step, grad_scale=gs, sumsq_out=sumsq, touched=flags.cuda(), touched_log2=6)
    on = flags.bool().repeat_interleave(64)[:n]
    seg = torch.bucketize(torch.arange(n), torch.tensor(seg_end), right=True)
    gr = (g0 if with_g else torch.zeros(n)) * gs + torch.tensor(reg)[seg] * p0
    m1 = m0 + (gr - m0) * 0.1
    v1 = v0 * 0.999 + gr * gr * (1 - 0.999)
    bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
    p1 = (p0 - (lr / bc1) * (m1 / (v1.sqrt() / bc2 ** 0.5 + 1e-8))).clamp(O.CLAMP_LO, O.CLAMP_HI)
    Pc, Mc, Vc = P.cpu(), M.cpu(), V.cpu()
    assert torch.equal(Pc[~on], p0[~on]) and torch.equal(Mc[~on], m0[~on]) and torch.equal(Vc[~on], v0[~on])
    if bool(on.any()):
        assert_close(Mc[on], m1[on], 1e-6, 1e-7)
        assert_close(Vc[on], v1[on], 1e-6, 1e-9)
        assert_close(Pc[on], p1[on], 1e-5, 1e-4)
    if with_g:
        Gc = G.cpu()
        assert float(Gc[on].abs().max() if bool(on.any()) else 0.0) == 0.0 and torch.equal(Gc[~on], g0[~on])
    want = torch.zeros(len(sizes), dtype=torch.float64).index_add_(0, seg[on], Pc[on].double() ** 2)
    assert_close(sumsq, want.float(), 2e-5, 1e-3)
