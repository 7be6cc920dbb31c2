"""Round 5: pair images - feature maps stored as packed fp16 pairs by their producer (include/stylemesh_hip.h, PAIR
IMAGES; stylemesh_amd/csrc/conv_split_kernel.h). Kernel level: with the table's scales equal to the scales the fp32-plane
path derives from the exact bounds, every consumer of a pair image must reproduce that path BIT FOR BIT (same pairs, same
product order) - convolutions (plain / pooled / un-pooling / gated / with addend / with the Gram epilogue), the Gram
forward and backward; and the overflow protocol (sm_pair_check -> sm_adam_hyper_step guard -> sm_adam_fused) must leave an
invalidated step without a trace. Reference operators: content_and_style_losses.py:47-70 (convs), :74-80 (Gram)."""
import pytest
import torch
import torch.nn.functional as F

from gpu_util import require_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rt(monkeypatch):
    require_gpu()
    from stylemesh_amd.runtime import hip, ops
    from stylemesh_amd.runtime.fmap import FMap
    monkeypatch.setattr(ops, "CONV_MODE", "split2")
    monkeypatch.setattr(ops, "GRAM_MODE", "split2")

    class RT:
        pass
    r = RT()
    r.hip, r.ops, r.FMap = hip, ops, FMap
    return r


def table_for(rt, amax_value: float, headroom: float = 1.0):
    """{scale, 1 / scale} of a tensor whose bound is ``amax_value`` - through the library's own sm_pair_roll."""
    book = rt.ops.new_amax("cuda", float(amax_value))
    tab = torch.zeros(2, device="cuda")
    rt.ops.pair_roll(book, 1, headroom, tab)
    return tab


def encode(x: torch.Tensor, s: float) -> torch.Tensor:
    """The pair words of fp32 values under scale ``s`` (host restatement of pair_encode)."""
    xs = (x.float() * s).clamp(-65000.0, 65000.0)
    h = xs.half()
    l = (xs - h.float()).half()
    return (h.view(torch.int16).to(torch.int32) & 0xFFFF) | (l.view(torch.int16).to(torch.int32) << 16)


def words(fm) -> torch.Tensor:
    return fm.to_dense().view(torch.int32)


def put_words(fm, w: torch.Tensor):
    fm.from_dense(w.view(torch.float32))
    return fm


def decode(w: torch.Tensor, inv: float) -> torch.Tensor:
    h = (w & 0xFFFF).to(torch.int16).view(torch.float16).float()
    l = (w >> 16).to(torch.int16).view(torch.float16).float()
    return (h + l) * inv


def dev(t):
    return t.cuda()


@pytest.mark.parametrize("cin,cmid,cout,H,W", [(64, 128, 128, 30, 41), (128, 256, 64, 17, 23), (256, 512, 512, 9, 13),
                                                (64, 64, 128, 150, 201)])
def test_pair_conv_chain_is_bit_identical_to_fp32_planes(rt, cin, cmid, cout, H, W):
    """conv -> conv with the intermediate stored as pairs == the same with fp32 planes (the consumer of an fp32 plane
    builds exactly the pairs the producer stored, given the same scale); the stored words are the host's encoding."""
    ops, hip = rt.ops, rt.hip
    torch.manual_seed(cin + W)
    x = F.relu(torch.randn(cin, H, W) * 2)
    w1 = torch.randn(cmid, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    w2 = torch.randn(cout, cmid, 3, 3) * (2.0 / (9 * cmid)) ** 0.5
    b1, b2 = dev(torch.randn(cmid) * 0.2), dev(torch.randn(cout) * 0.2)
    p1, p2 = dev(ops.pack_conv_fwd(w1)), dev(ops.pack_conv_fwd(w2))
    q1, q2 = ops.pack_conv_split2(p1), ops.pack_conv_split2(p2)
    xin = rt.FMap(cin, H, W).from_dense(x)
    am_x = ops.new_amax("cuda", float(x.abs().max()))
    # fp32 planes
    y, z_ref = rt.FMap(cmid, H, W), rt.FMap(cout, H, W)
    am_y, am_z = ops.new_amax("cuda"), ops.new_amax("cuda")
    ops.conv3x3_grouped([(xin, y, None)], p1, b1, hip.EPI_BIAS_RELU, wt2=q1, amax_in=am_x, amax_out=am_y)
    ops.conv3x3_grouped([(y, z_ref, None)], p2, b2, hip.EPI_BIAS_RELU, wt2=q2, amax_in=am_y, amax_out=am_z)
    # pairs, stored under the scale the fp32 path derives from the exact bound
    tab = table_for(rt, float(am_y.max()))
    s, inv = float(tab[0]), float(tab[1])
    yp, z = rt.FMap(cmid, H, W), rt.FMap(cout, H, W)
    am_y2, am_z2 = ops.new_amax("cuda"), ops.new_amax("cuda")
    ops.conv3x3_grouped([(xin, yp, None)], p1, b1, hip.EPI_BIAS_RELU, wt2=q1, amax_in=am_x, amax_out=am_y2, pair_out=tab)
    assert torch.equal(words(yp), encode(y.to_dense(), s))
    assert float(am_y2.max()) == float(am_y.max())           # the bound is taken of the fp32 values
    assert float((decode(words(yp), inv) - y.to_dense()).abs().max()) <= 2.0 ** -21 * float(am_y.max())
    ops.conv3x3_grouped([(yp, z, None)], p2, b2, hip.EPI_BIAS_RELU, wt2=q2, amax_out=am_z2, pair_in=tab)
    assert torch.equal(z.to_dense(), z_ref.to_dense())
    assert float(am_z2.max()) == float(am_z.max()) and z.border_is_zero() and yp.border_is_zero()


@pytest.mark.parametrize("C,cout,H,W", [(64, 64, 37, 50), (128, 128, 40, 53), (256, 256, 21, 30)])
def test_pair_dgrad_below_a_pool_gate_addend_and_pair_output(rt, C, cout, H, W):
    """The un-pooling data gradient with EVERY tensor in its pair form - pooled gradient in, ReLU gate planes, output -
    and an fp32 addend plane: the stored words are the encoding of the fp32 path's result."""
    ops, hip = rt.ops, rt.hip
    torch.manual_seed(C + W)
    act = F.relu(torch.randn(C, H, W))
    act[:, 4:12, 6:20] = 0.75
    act[:, 14:18, :] = 0.0
    below = F.relu(torch.randn(cout, H, W))
    below[:, :, 5:9] = 0.0                          # closed gates
    dpooled = torch.randn(C, H // 2, W // 2) * 1e-4
    addend = torch.randn(cout, H, W) * 1e-4
    wgt = torch.randn(C, cout, 3, 3) * (2.0 / (9 * C)) ** 0.5
    wd = dev(ops.pack_conv_dgrad(wgt))
    wd2 = ops.pack_conv_split2(wd)
    a = rt.FMap(C, H, W).from_dense(act)
    pooled = rt.FMap(C, H // 2, W // 2)
    code = torch.zeros(C // 8 * pooled.plane, dtype=torch.int32, device="cuda")
    ops.maxpool_fwd_grouped([(a, pooled)], None, [code])
    gate = rt.FMap(cout, H, W).from_dense(below)
    dp = rt.FMap(C, H // 2, W // 2).from_dense(dpooled)
    amax_in = ops.new_amax("cuda", float(dpooled.abs().max()))
    t_in, t_gate = table_for(rt, float(dpooled.abs().max())), table_for(rt, float(below.abs().max()))
    dp_p = put_words(rt.FMap(C, H // 2, W // 2), encode(dpooled, float(t_in[0])).cuda())
    gate_p = put_words(rt.FMap(cout, H, W), encode(below, float(t_gate[0])).cuda())
    add_f = rt.FMap(cout, H, W).from_dense(addend)
    for flags in (hip.EPI_RELU_MASK, hip.EPI_RELU_MASK | hip.EPI_ADD):
        ref = rt.FMap(cout, H, W).from_dense(addend)
        am_ref, am = ops.new_amax("cuda"), ops.new_amax("cuda")
        ops.conv3x3_grouped([(dp, ref, gate, code)], wd, None, flags, wt2=wd2, amax_in=amax_in, amax_out=am_ref)
        t_out = table_for(rt, float(am_ref.max()))
        out = rt.FMap(cout, H, W)
        ops.conv3x3_grouped([(dp_p, out, gate_p, code)], wd, None, flags, wt2=wd2, amax_out=am, pair_in=t_in,
                            pair_out=t_out, pair_gate=t_gate, addends=[add_f] if flags & hip.EPI_ADD else None)
        assert torch.equal(words(out), encode(ref.to_dense(), float(t_out[0]))), flags
        assert float(am.max()) == float(am_ref.max()) and out.border_is_zero()


def test_pair_pooling_epilogue_and_tail_tiles(rt):
    """The forward conv below a pool stores the POOLED map as pairs (whole tiles and K-split tail tiles through the second
    pass): words = encoding of the fp32 path's pooled map, same argmax codes."""
    ops, hip = rt.ops, rt.hip
    torch.manual_seed(5)
    for cin, cout, H, W in ((64, 64, 60, 83), (128, 128, 24, 37), (256, 256, 12, 17)):
        x = F.relu(torch.randn(cin, H, W) * 2)
        wgt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
        b = dev(torch.randn(cout) * 0.2)
        p = dev(ops.pack_conv_fwd(wgt))
        q = ops.pack_conv_split2(p)
        xin = rt.FMap(cin, H, W).from_dense(x)
        am_x = ops.new_amax("cuda", float(x.abs().max()))
        t_x = table_for(rt, float(x.abs().max()))
        xin_p = put_words(rt.FMap(cin, H, W), encode(x, float(t_x[0])).cuda())
        Ho, Wo = H // 2, W // 2
        # pair lists over the whole pooled plane
        _, group = ops.conv_list_format(cin, cout)
        need = torch.ones(Ho, Wo, device="cuda")
        cap = 2 * Ho * ((Wo + 15) // 16 + 1) + 2
        starts = torch.empty(cap, dtype=torch.int32, device="cuda")
        count = torch.zeros(1, dtype=torch.int32, device="cuda")
        ops.cover_segments([(need, starts, count, 0, W)])
        n = int(count)
        lst = torch.cat([starts[:n], torch.full(((-n) % group,), 0xFFFFFF, dtype=torch.int32, device="cuda")])
        outs = []
        for pair in (False, True):
            pre, pooled = rt.FMap(cout, H, W), rt.FMap(cout, Ho, Wo)
            code = torch.zeros(cout // 8 * pooled.plane, dtype=torch.int32, device="cuda")
            am = ops.new_amax("cuda")
            kw = {}
            if pair:
                kw = dict(pair_in=t_x, pair_out=table_for(rt, outs[0][2]))
            ops.conv3x3_grouped([(xin_p if pair else xin, pre, None, None, pooled, code)], p, b,
                                hip.EPI_BIAS_RELU | hip.EPI_POOL, lst, 1.0, None, q, None if pair else am_x, am, **kw)
            outs.append((pooled, code, float(am.max())))
        (ref, code_ref, am_ref), (got, code_got, am_got) = outs
        s_out = float(table_for(rt, am_ref)[0])
        assert torch.equal(words(got), encode(ref.to_dense(), s_out)), (cin, cout)
        assert torch.equal(code_got, code_ref) and am_got == am_ref


@pytest.mark.parametrize("C,two_masks", [(128, True), (256, False), (512, True)])
def test_pair_gram_forward_and_backward_match_the_fp32_plane_kernels(rt, C, two_masks):
    """Masked Gram forward (atomic accumulation: compared to rounding) and backward (bit-identical) of a style layer whose
    feature map is stored as pairs, against the kernels that convert the fp32 planes themselves."""
    ops, hip = rt.ops, rt.hip
    torch.manual_seed(C)
    H, W = 33, 45
    f = F.relu(torch.randn(C, H, W) * 2)
    mk = torch.zeros(2, H, W)
    sel = torch.rand(H, W)
    mk[0] = (sel < 0.4).float()
    mk[1] = ((sel >= 0.4) & (sel < 0.7)).float()
    feat = rt.FMap(C, H, W).from_dense(f)
    masks = rt.FMap(2, H, W).from_dense(mk)
    af = ops.new_amax("cuda", float(f.abs().max()))
    tab = table_for(rt, float(f.abs().max()))
    feat_p = put_words(rt.FMap(C, H, W), encode(f, float(tab[0])).cuda())
    m0, m1 = masks.channel_ptr(0), masks.channel_ptr(1) if two_masks else None
    res = []
    for fm, pf in ((feat, None), (feat_p, tab)):
        S0, S1 = torch.zeros(C, C, device="cuda"), torch.zeros(C, C, device="cuda") if two_masks else None
        ops.gram_masked_grouped(ops.struct_array(hip.GramProblem, [ops.gram_problem(fm, m0, m1, S0, S1, af, pair_feat=pf)]))
        res.append((S0, S1))
    for a, b in zip(res[0], res[1]):
        if a is not None:
            ta, tb = torch.triu(a), torch.triu(b)
            assert float(ta.abs().max()) > 0
            assert float((ta - tb).abs().max()) <= 2e-6 * float(ta.abs().max())
    D0 = (torch.randn(C, C) * 3e-3).cuda()
    D0 = D0 + D0.t()
    D1 = None
    if two_masks:
        D1 = (torch.randn(C, C) * 1e-3).cuda()
        D1 = D1 + D1.t()
    ad = ops.new_amax("cuda", max(float(D0.abs().max()), float(D1.abs().max()) if two_masks else 0.0))
    for relu_gate in (False, True):
        outs = []
        for fm, pf in ((feat, None), (feat_p, tab)):
            d = rt.FMap(C, H, W)
            ws = torch.empty(ops.gram_backward_ws_bytes(C), dtype=torch.uint8, device="cuda")
            am = ops.new_amax("cuda")
            ops.gram_backward_grouped(ops.struct_array(hip.GramBwdProblem, [
                ops.gram_bwd_problem(fm, m0, m1, D0, D1, d, ws, af, ad, relu_gate=relu_gate, amax_out=am, pair_feat=pf)]))
            outs.append((d.to_dense(), float(am.max())))
        assert float(outs[0][0].abs().max()) > 0
        assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1], relu_gate


def test_pair_overflow_is_detected_and_the_update_left_untouched(rt):
    """A tensor that outgrows its predicted scale: the stored pairs saturate, sm_pair_check reports it, the guarded
    hyper-step marks the update invalid and sm_adam_fused changes nothing but the gradient (zeroed) and sum(p^2); with
    the scales rolled from the bounds the failed attempt recorded the repeat is valid."""
    ops, hip = rt.ops, rt.hip
    torch.manual_seed(1)
    cin, cout, H, W = 64, 64, 20, 27
    x = F.relu(torch.randn(cin, H, W)) * 50.0
    wgt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    p = dev(ops.pack_conv_fwd(wgt))
    q = ops.pack_conv_split2(p)
    b = dev(torch.zeros(cout))
    xin = rt.FMap(cin, H, W).from_dense(x)
    am_x = ops.new_amax("cuda", float(x.abs().max()))
    W_ = ops.AMAX_FLOATS
    book = torch.zeros(2 * W_, device="cuda")          # entry 0: the conv's output, entry 1: unused (stays zero)
    book[0] = 1e-3                                      # "previous step": a bound 5 orders of magnitude too small
    table = torch.zeros(4, device="cuda")
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    entries = torch.tensor([0, 1], dtype=torch.int32, device="cuda")
    hyper = torch.zeros(3, device="cuda")
    state = torch.tensor([1.0, 7.0], dtype=torch.float64, device="cuda")
    n = 4096
    pv, gv = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    mv, vv = torch.randn(n, device="cuda") * 0.1, torch.rand(n, device="cuda") * 0.01
    ref_y = rt.FMap(cout, H, W)
    ops.conv3x3_grouped([(xin, ref_y, None)], p, b, hip.EPI_BIAS_RELU, wt2=q, amax_in=am_x, amax_out=ops.new_amax("cuda"))
    for attempt in range(2):
        ops.pair_roll(book, 2, 4.0, table)
        book.zero_()
        y = rt.FMap(cout, H, W)
        ops.conv3x3_grouped([(xin, y, None)], p, b, hip.EPI_BIAS_RELU, wt2=q, amax_in=am_x, amax_out=book[:W_],
                            pair_out=table[0:2])
        ops.pair_check(book, table, entries, status)
        ops.adam_hyper_step(state, hyper, guard=status)
        p0, g0, m0, v0 = pv.clone(), gv.clone(), mv.clone(), vv.clone()
        sumsq = torch.zeros(1, device="cuda")
        ops.adam_fused(pv, gv, mv, vv, [n], [0.0], 1.0, 8, sumsq_out=sumsq, dev_hyper=hyper)
        st = status.tolist()
        got = decode(words(y), float(table[1]))
        if attempt == 0:
            assert st[0] == 0 and st[1] == 1 and st[2] == 1 and st[3] == 0
            assert float(hyper[2]) == 0.0 and state.tolist() == [1.0, 7.0]
            assert float(got.max()) < 0.5 * float(ref_y.to_dense().max())          # saturated
            assert torch.equal(pv, p0) and torch.equal(mv, m0) and torch.equal(vv, v0) and float(gv.abs().max()) == 0.0
            assert abs(float(sumsq) - float((p0.clamp(ops.CLAMP_LO, ops.CLAMP_HI) ** 2).sum())) <= 1e-3 * float(sumsq)
            gv.copy_(g0)
        else:
            assert st[0] == 1 and st[1] == 1 and st[2] == 2 and st[3] == -1
            assert float(hyper[2]) == 1.0 and state.tolist() == [1.0, 8.0]
            assert float((got - ref_y.to_dense()).abs().max()) <= 2.0 ** -19 * float(ref_y.to_dense().max())
            assert not torch.equal(pv, p0) and float(gv.abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------------------------
# engine level: the pair-image steps (grouped side-stream path) against the reference's goldens and against the fp32-plane
# steps of the same engine
# ---------------------------------------------------------------------------------------------------------------------
def _engine(pair: bool, init=None):
    import test_engine_gpu as E
    from golden_cases import FLAGSETS
    eng = E.make_engine(FLAGSETS["with_angle_and_depth"], init)
    eng.overlap_min_pixels = 0          # the small golden views take the side-stream path of the full-size steps
    eng.pair_images = pair
    return eng


def test_pair_image_steps_match_the_reference_golden_from_a_zero_texture():
    """Five Adam steps from the all-zero texture (golden g6, the reference's own run): the first steps' bounds grow by
    orders of magnitude, so the predicted scales fail and the device-invalidated steps are repeated - the textures must
    still be the reference's, and every invalidated step must have been made up for."""
    require_gpu()
    import numpy as np
    import stylemesh_oracle as O
    import test_engine_gpu as E
    from conftest import batch_from_golden, load_golden
    d = load_golden("g6_adam_zero")
    g5 = load_golden("g5_with_angle_and_depth")
    eng = _engine(True)
    batch = batch_from_golden(g5)
    for step in range(5):
        lt = eng.training_step(batch)
        eng.finish_pending()
        assert eng.step_count == step + 1
        if step % 2 == 1:
            eng.end_epoch()
        if step in (0, 1, 4):
            for i in range(4):
                ref = torch.from_numpy(d[f"p{i}_after{step + 1}"]).clamp(O.CLAMP_LO, O.CLAMP_HI)
                E.texture_close(eng.layers[i], ref, step, f"pair images, layer {i} after {step + 1} steps")
    st = eng.pair_stats
    assert st["steps"] >= 6 and st["invalid"] >= 1 and st["repeated"] == st["invalid"], st   # (the very first step has no bounds)
    assert float(eng.arena.g.abs().max()) == 0.0
    dev_step = float(eng._hyper_state3[1])
    assert dev_step == 5.0, dev_step


def test_pair_image_step_equals_the_fp32_plane_step_and_an_invalid_step_leaves_no_trace():
    """Same view, same seeded texture: (1) a pair-image step without bounds is invalidated - texture, moments and step
    count stay exactly as they were; (2) its repeat produces the gradient of the fp32-plane step to rounding (the pairs
    differ only where head-room moves low bits of l) and the same losses; (3) poisoned bounds in the middle of a run are
    caught and made up for."""
    require_gpu()
    import numpy as np
    from conftest import batch_from_golden, load_golden
    g5 = load_golden("g5_with_angle_and_depth")
    init = [torch.from_numpy(g5[f"init{i}"]) for i in range(4)]
    batch = batch_from_golden(g5)
    ref = _engine(False, init)
    lt_ref = ref.step_compute(batch)
    g_ref = [g.clone() for g in ref.grads]
    loss_ref = {k: float(v) for k, v in lt_ref.items()}
    ref.optimizer_step()
    eng = _engine(True, init)
    p0, m0, v0 = eng.arena.p.clone(), eng.arena.m.clone(), eng.arena.v.clone()
    eng.step_compute(batch)                       # no bounds yet: scales 1
    assert eng._pair_step
    eng.optimizer_step()
    torch.cuda.synchronize()
    assert eng._pair_status.tolist()[0] == 0
    assert torch.equal(eng.arena.p, p0) and torch.equal(eng.arena.m, m0) and torch.equal(eng.arena.v, v0)
    assert float(eng.arena.g.abs().max()) == 0.0
    lt = eng.step_compute(batch)                  # the repeat (driven by hand here)
    for a, b in zip(eng.grads, g_ref):
        mx = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * mx, (float((a - b).abs().max()), mx)
    for k in ("content", "style"):
        np.testing.assert_allclose(float(lt[k]), loss_ref[k], rtol=2e-6)
    eng.optimizer_step()
    torch.cuda.synchronize()
    assert eng._pair_status.tolist()[0] == 1
    eng._pair_poll(block=True)
    assert eng._pair_failed == 1 and eng.step_count == 1      # one verdict of two was "invalid"
    eng._pair_failed = 0                                      # (made up for by hand above)
    assert float((eng.arena.p - ref.arena.p).abs().max()) <= 1e-3
    # poisoned bounds: the next step must be invalidated and repeated by the engine itself
    for e in (eng, ref):
        e.training_step(batch)
    eng.amax.buf.zero_()
    before = dict(eng.pair_stats)
    for e in (eng, ref):
        e.training_step(batch)
    eng.finish_pending()
    assert eng.pair_stats["invalid"] >= before["invalid"] + 1 and eng.pair_stats["repeated"] >= before["repeated"] + 1
    assert eng.step_count == ref.step_count == 3
    err = (eng.arena.p - ref.arena.p).abs()
    assert float((err > 2e-3).float().mean()) <= 0.01 and float(err.max()) <= 0.3, (float(err.max()), float((err > 2e-3).float().mean()))


def test_step_program_recording_is_scoped_to_the_engines_own_calls():
    """the recorder stands in for the module-global ``ops.lib`` only while ONE engine issues its own
    calls. Two engines whose steps interleave (a.compute, b.compute, b.update, a.update) and a caller that launches
    library work between ``step_compute`` and ``optimizer_step`` (a hook rendering the texture) must each get programs
    with their own update segment - every step still takes its Adam update and equals the unrecorded engines' step."""
    require_gpu()
    import numpy as np
    import test_round4_gpu as R4
    from stylemesh_amd.runtime import hip, ops
    c = R4.PROGRAM_CASES["only2D"]
    views = R4._small_views((0, 2))
    a, b = R4._program_engine(c, "1"), R4._program_engine(c, "1")
    ra, rb = R4._program_engine(c, "0"), R4._program_engine(c, "0")
    scratch = torch.zeros(64, device="cuda")
    for i in range(14):
        batch = views[i // 7]
        for e, r in ((a, ra), (b, rb)):
            for dst, src in ((e.arena.p, r.arena.p), (e.arena.m, r.arena.m), (e.arena.v, r.arena.v), (e.sumsq, r.sumsq)):
                dst.copy_(src)             # lock-step with the unrecorded twins
            if e.touched is not None:
                e.touched.copy_(r.touched)
        la = a.step_compute(batch)
        assert ops.lib is hip.lib          # paused between the two halves of a's step
        ops.zero_floats(scratch)           # a caller's own library call: must not enter anybody's program
        lb = b.step_compute(batch)
        b.optimizer_step()
        a.optimizer_step()
        assert ops.lib is hip.lib
        lra = ra.step_compute(batch)
        ra.optimizer_step()
        lrb = rb.step_compute(batch)
        rb.optimizer_step()
        np.testing.assert_allclose(a.losses(la)["total"], ra.losses(lra)["total"], rtol=1e-5)
        np.testing.assert_allclose(b.losses(lb)["total"], rb.losses(lrb)["total"], rtol=1e-5)
        for e, r in ((a, ra), (b, rb)):
            err = (e.arena.p - r.arena.p).abs()
            assert float((err > 2e-3).float().mean()) < 0.01, (i, float(err.max()))
        assert float((a.arena.p - ra.arena.p).abs().max()) < 0.3
    for e in (a, b):
        assert e.program_replays >= 4, e.program_replays
        for prog in e._programs.values():
            assert "sm_zero_floats" not in prog.names[prog.n_compute:] or prog.names.count("sm_adam_fused") >= 1
            assert any(n == "sm_adam_fused" for n in prog.names[prog.n_compute:])


def test_two_rank_pipelined_exchange_equals_exchange_then_update_bit_for_bit():
    """the pipelined exchange + update - round 5's default from 32 MB of flagged chunks on - against
    the all-reduce followed by one fused update, from the same state and the same local gradients on two ranks (one
    device, gloo): p, m, v, the zeroed gradient and sum(p^2) identical to the bit, identical across the ranks; the policy
    answers the same on every rank."""
    require_gpu()
    import os
    import tempfile
    import test_round2_gpu as R2
    from conftest import REPO
    with tempfile.TemporaryDirectory() as tmp:
        r = R2._launch_ranks([os.path.join(REPO, "tests", "two_rank_worker.py"), "pipelined", tmp], 2,
                             {"STYLEMESH_TEST_BACKEND": "gloo"}, timeout=1200)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        r0, r1 = (torch.load(os.path.join(tmp, f"rank{k}.pt")) for k in (0, 1))
    assert len(r0["steps"]) >= 2
    for s0, s1 in zip(r0["steps"], r1["steps"]):
        (plain0, pipe0), (plain1, pipe1) = s0, s1
        for a, b, c in zip(plain0[:4], pipe0[:4], plain1[:4]):
            assert torch.equal(a, b) and torch.equal(a, c)
        assert float(plain0[1].abs().max()) == 0.0 and float(plain0[0].abs().max()) > 0
        # (sum(p^2) is added with atomics, per range in the pipelined form: equal to rounding)
        assert torch.allclose(plain0[4], pipe0[4], rtol=1e-5) and torch.allclose(plain0[4], plain1[4], rtol=1e-5)
    for r in (r0, r1):
        assert r["auto_small"] is False and r["auto_large"] is True


@pytest.mark.parametrize("n,key_bits", [(1, 27), (255, 9), (4097, 18), (350_003, 22), (3_000_017, 27), (1_000_003, 32)])
def test_own_radix_sort_is_the_stable_sort(n, key_bits):
    """sm_radix_sort_pairs (round 5: the scatter plan's own sort instead of rocPRIM's) against torch's stable sort: same
    keys, and equal keys keep their input order - with spatially coherent keys (long runs), random keys, heavy
    duplicates and the all-ones 'invalid' key that must end up in the tail."""
    require_gpu()
    import ctypes as C
    from stylemesh_amd.runtime import hip
    g = torch.Generator(device="cuda").manual_seed(n)
    top = (1 << key_bits) - 1
    coherent = (torch.arange(n, device="cuda") // 7 * 3) % (top + 1)                       # runs of equal keys, slowly rising
    rand = torch.randint(0, top + 1, (n,), device="cuda", generator=g, dtype=torch.int64)
    dup = torch.randint(0, 17, (n,), device="cuda", generator=g, dtype=torch.int64) * (top // 17)
    pick = torch.randint(0, 4, (n,), device="cuda", generator=g)
    keys64 = torch.where(pick == 0, coherent, torch.where(pick == 1, rand, torch.where(pick == 2, dup, torch.full_like(rand, top))))
    k0 = keys64.to(torch.int32) if key_bits < 32 else (keys64 - (keys64 >= (1 << 31)).long() * (1 << 32)).to(torch.int32)
    v0 = torch.arange(n, device="cuda", dtype=torch.int64) * 3 + 1
    k1, v1 = torch.empty_like(k0), torch.empty_like(v0)
    tb = hip.lib.sm_tex_scatter_plan_temp_bytes(n, key_bits)
    temp = torch.empty(max(tb, 16), dtype=torch.uint8, device="cuda")
    which = C.c_int(-1)
    kin, vin = k0.clone(), v0.clone()
    hip.check(hip.lib.sm_radix_sort_pairs(k0.data_ptr(), k1.data_ptr(), v0.data_ptr(), v1.data_ptr(), n, key_bits,
                                          temp.data_ptr(), temp.numel(), C.byref(which), hip.stream()), "sm_radix_sort_pairs")
    torch.cuda.synchronize()
    ks, vs = (k0, v0) if which.value == 0 else (k1, v1)
    order = torch.sort(keys64, stable=True).indices
    assert torch.equal(ks.long() & 0xFFFFFFFF, keys64[order])
    assert torch.equal(vs, vin[order])
