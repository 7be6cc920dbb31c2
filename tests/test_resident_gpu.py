"""Round 5: the resident-input conv kernel (``SM_LIST_QUADS``, csrc/conv_split_kernel.h RES) and the quad modes of
``sm_cover_segments`` that feed it.

* the quad covers against the greedy restated on the host (pooled and flat), and their structure: four entries per run =
  the same 32 columns of four consecutive rows, disjoint runs per row group, every needed position covered;
* every epilogue variant of the resident kernel - forward (+ pooling), plain / gated / adding data gradients, the
  un-pooling input, two 64-channel phases, the Gram epilogue - against the ring kernel on whole tiles: bit-identical
  outputs on the positions the quads cover, nothing written elsewhere, the recorded bound = max |written output|;
* the engine: a step with and without the quad lists gives the same losses (reference model/losses/content_and_style_losses.py:49-69,
  the conv stack).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
PAD = 0xFFFFFF


def require_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _host_quads(need, pooled, hip, tag, full_w=None):
    """The greedy of sm_cover_segments' quad modes, restated: [(entries of a run) ...] flattened."""
    h, w = need.shape
    G = 2 if pooled else 4
    Wp = hip.row_stride(full_w if pooled else w)
    H = 2 * h if pooled else h
    step = 16 if pooled else 32
    out = []
    for Y in range((h + G - 1) // G):
        bits = need[Y * G:(Y + 1) * G].any(axis=0)
        cur = 0
        while cur < w:
            nz = np.nonzero(bits[cur:])[0]
            if len(nz) == 0:
                break
            p = cur + int(nz[0])
            if not pooled:
                p &= ~1                      # flat quads start on even columns
            q = (4 * Y + 1) * Wp + (2 * p if pooled else p) + 1
            for i in range(4):
                out.append((tag << 24) | (q + i * Wp if 4 * Y + i < H else PAD))
            cur = p + step
    return np.array(out, dtype=np.int64)


def _quad_cover(ops, hip, need, tag, full_w=0):
    """Device quad cover of one need map -> int32 tensor of entries (a multiple of four)."""
    h, w = need.shape
    cap = 4 * ((h + 1) // 2 if full_w else (h + 3) // 4) * ((w + (15 if full_w else 31)) // (16 if full_w else 32) + 1) + 4
    starts = torch.full((cap,), -1, dtype=torch.int32, device="cuda")
    count = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.cover_segments([(need, starts, count, tag, full_w, 1)])
    n = int(count)
    assert 0 <= n <= cap and n % 4 == 0
    return starts[:n]


@pytest.mark.parametrize("hw", [(37, 50), (150, 201), (6, 3), (64, 85), (256, 341)])
def test_quad_covers_equal_the_host_greedy(hw):
    require_gpu()
    from stylemesh_amd.runtime import hip, ops
    H, W = hw
    rng = np.random.default_rng(H * 1000 + W)
    for density in (0.02, 0.4, 1.0):
        # flat quads over an H x W need map
        nd = (rng.random((H, W)) < density)
        nd[H // 2:, : W // 3] = False
        got = _quad_cover(ops, hip, torch.from_numpy(nd.astype(np.float32)).cuda(), 3).cpu().numpy().astype(np.int64)
        want = _host_quads(nd, False, hip, 3)
        assert np.array_equal(got, want)
        Wp = hip.row_stride(W)
        covered = np.zeros(((H + 6) * Wp,), dtype=np.int32)
        for k in range(0, len(got), 4):
            q = got[k] & PAD
            assert (got[k] >> 24) == 3 and q != PAD
            y, x = q // Wp - 1, q % Wp - 1
            assert y % 4 == 0 and 0 <= x < W and x % 2 == 0
            for i in range(4):
                e = got[k + i] & PAD
                assert e == (q + i * Wp if y + i < H else PAD)
                if e != PAD:
                    covered[e:e + min(32, Wp - 1 - x)] += 1       # (positions of the run inside its own row)
        assert covered.max() <= 1                                 # disjoint
        ys, xs = np.nonzero(nd)
        assert np.all(covered[(ys + 1) * Wp + xs + 1] == 1)       # every needed position
        # pooled quads: the need map of the pooled plane, full-resolution entries
        if H >= 2 and W >= 2:
            ndp = (rng.random((H // 2, W // 2)) < density)
            got = _quad_cover(ops, hip, torch.from_numpy(ndp.astype(np.float32)).cuda(), 1, W).cpu().numpy().astype(np.int64)
            want = _host_quads(ndp, True, hip, 1, W)
            assert np.array_equal(got, want)
            for k in range(0, len(got), 4):
                q = got[k] & PAD
                y, x = q // Wp - 1, q % Wp - 1
                assert y % 4 == 0 and x % 2 == 0


def _dense_rows(hip, hws, rows, group):
    """Dense list of row blocks (rows x 32 columns), every level padded to whole tiles of `group` entries."""
    parts = []
    for g, (H, W) in enumerate(hws):
        Wp = hip.row_stride(W)
        e = []
        for Y in range(0, H, rows):
            for x0 in range(0, W, 32):
                for i in range(rows):
                    e.append((g << 24) | ((Y + i + 1) * Wp + x0 + 1 if Y + i < H else PAD))
        e += [(g << 24) | PAD] * ((-len(e)) % group)
        parts += e
    return torch.tensor(np.array(parts, dtype=np.int64).astype(np.int32), device="cuda")


def _covered_mask(hip, lst, g, H, W, plane):
    """Boolean [plane]: the positions of level g the list's segments cover."""
    Wp = hip.row_stride(W)
    m = torch.zeros(plane + 64, dtype=torch.bool)
    for e in lst.cpu().numpy().astype(np.int64):
        if (e >> 24) == g and (e & PAD) != PAD:
            q = int(e & PAD)
            m[q:min(q + 32, (q // Wp + 1) * Wp)] = True      # (a lane stays in its segment's row)
    return m[:plane].cuda()


def _needs(hws, pooled, seed):
    gen = torch.Generator().manual_seed(seed)
    out = []
    for (H, W) in hws:
        h, w = (H // 2, W // 2) if pooled else (H, W)
        nd = (torch.rand(h, w, generator=gen) < 0.35).float()
        nd[h // 2:, : w // 3] = 0
        out.append(nd.cuda())
    return out


CASES = [[(37, 50)], [(150, 201), (64, 85)], [(256, 341)], [(6, 5), (40, 53)]]


@pytest.mark.parametrize("hws", CASES)
def test_resident_forward_with_pooling_epilogue(hws, monkeypatch):
    require_gpu()
    import torch.nn.functional as F
    from stylemesh_amd.runtime import hip, ops
    from stylemesh_amd.runtime.fmap import FMap
    monkeypatch.setattr(ops, "CONV_MODE", "split2")
    torch.manual_seed(len(hws))
    wgt = torch.randn(64, 64, 3, 3) * (2.0 / (9 * 64)) ** 0.5
    b = (torch.randn(64) * 0.3).cuda()
    w = ops.pack_conv_fwd(wgt).cuda()
    w2 = ops.pack_conv_split2(w)
    xs = [F.relu(torch.randn(64, H, W) * 2) for H, W in hws]
    for x in xs:
        x[:, : x.shape[1] // 3] = 0              # closed windows (code 4) and ties
    ins = [FMap(64, H, W).from_dense(x.cuda()) for x, (H, W) in zip(xs, hws)]
    amax_in = ops.new_amax("cuda", max(float(x.abs().max()) for x in xs))
    tiny = torch.zeros(4, device="cuda")
    monkeypatch.setattr(ops, "splitk_workspace", lambda device: tiny)      # the ring kernel on whole tiles: same sums

    def run(lst, quads):
        outs = [FMap(64, H, W) for H, W in hws]
        pooled = [FMap(64, H // 2, W // 2) for H, W in hws]
        codes = [torch.zeros(8 * p.plane, dtype=torch.int32, device="cuda") for p in pooled]
        am = ops.new_amax("cuda")
        ops.conv3x3_grouped([(i, o, None, None, p, c) for i, o, p, c in zip(ins, outs, pooled, codes)], w, b,
                            hip.EPI_BIAS_RELU | hip.EPI_POOL, lst, 1.0, w2, amax_in, am, quads=quads)
        return outs, pooled, codes, am
    _, p_ref, c_ref, _ = run(_dense_rows(hip, hws, 2, 8), False)
    needs = _needs(hws, True, 5)
    lst = torch.cat([_quad_cover(ops, hip, nd, g, W) for g, (nd, (H, W)) in enumerate(zip(needs, hws))])
    outs, pooled, codes, am = run(lst, True)
    true_max = 0.0
    for g, ((H, W), nd) in enumerate(zip(hws, needs)):
        assert float(outs[g].planes.abs().max()) == 0.0                 # the full-resolution output is not written
        pm = pooled[g]
        # pooled positions the quads cover: upper rows of the vertical pairs
        cov = torch.zeros(pm.plane, dtype=torch.bool)
        Wp, Wpo = hip.row_stride(W), pm.Wp
        ent = lst.cpu().numpy().astype(np.int64)
        ent = ent[(ent >> 24) == g]
        for k in range(0, len(ent), 2):
            if (ent[k] & PAD) == PAD:
                continue
            q = int(ent[k] & PAD)
            y, x = q // Wp - 1, q % Wp - 1
            x1 = min(x + 32, 2 * (W // 2))
            if y < 2 * (H // 2):
                cov[(y // 2 + 1) * Wpo + x // 2 + 1:(y // 2 + 1) * Wpo + x1 // 2 + 1] = True
        cov = cov.cuda()
        needed = torch.zeros(pm.plane, dtype=torch.bool, device="cuda")
        needed[: (pm.H + 2) * Wpo].view(pm.H + 2, Wpo)[1:pm.H + 1, 1:pm.W + 1] = nd > 0
        assert bool((cov | ~needed).all())                             # every needed window is covered
        assert torch.equal(pm.planes[:, cov], p_ref[g].planes[:, cov])
        # (a run that passes the end of its row continues in the next rows - two rows down it meets windows again: what
        # is stored there is the window's value all the same)
        written = pm.planes != 0
        assert torch.equal(pm.planes[written], p_ref[g].planes[written])
        if W >= 64:
            assert float(pm.planes[:, ~cov].abs().max()) == 0.0
        cg, cr = codes[g].view(8, -1), c_ref[g].view(8, -1)
        assert torch.equal(cg[:, cov], cr[:, cov])
        true_max = max(true_max, float(pm.planes.abs().max()))
    assert float(am.max()) == true_max


@pytest.mark.parametrize("variant", ["fwd", "plain128", "gate_unpool", "gate_add_unpool", "gate_add", "gate"])
@pytest.mark.parametrize("hws", CASES)
def test_resident_kernel_matches_ring_kernel(variant, hws, monkeypatch):
    require_gpu()
    import torch.nn.functional as F
    from stylemesh_amd.runtime import hip, ops
    from stylemesh_amd.runtime.fmap import FMap
    monkeypatch.setattr(ops, "CONV_MODE", "split2")
    torch.manual_seed(len(hws) + len(variant))
    cin = 128 if variant == "plain128" else 64
    wgt = torch.randn(64, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    if variant == "fwd":
        w = ops.pack_conv_fwd(wgt).cuda()
        bias, flags = (torch.randn(64) * 0.3).cuda(), hip.EPI_BIAS_RELU
    else:
        w = ops.pack_conv_dgrad(wgt.transpose(0, 1).contiguous()).cuda()     # forward weights [cin][64]: the gradient has 64 channels
        bias = None
        flags = {"plain128": 0, "gate_unpool": hip.EPI_RELU_MASK, "gate_add_unpool": hip.EPI_RELU_MASK | hip.EPI_ADD,
                 "gate_add": hip.EPI_RELU_MASK | hip.EPI_ADD, "gate": hip.EPI_RELU_MASK}[variant]
    w2 = ops.pack_conv_split2(w)
    unpool = variant.endswith("unpool")
    ins, gates, codes, addends = [], [], [], []
    for (H, W) in hws:
        if unpool:
            act = F.relu(torch.randn(cin, H, W))
            a, pooled = FMap(cin, H, W).from_dense(act.cuda()), FMap(cin, H // 2, W // 2)
            code = torch.zeros(cin // 8 * pooled.plane, dtype=torch.int32, device="cuda")
            ops.maxpool_fwd_grouped([(a, pooled)], None, [code])
            codes.append(code)
            ins.append(FMap(cin, H // 2, W // 2).from_dense(torch.randn(cin, H // 2, W // 2).cuda()))
        else:
            ins.append(FMap(cin, H, W).from_dense((F.relu(torch.randn(cin, H, W)) if variant == "fwd" else torch.randn(cin, H, W)).cuda()))
        gates.append(FMap(64, H, W).from_dense(F.relu(torch.randn(64, H, W)).cuda()) if flags & hip.EPI_RELU_MASK else None)
        addends.append(torch.randn(64, H, W).cuda() if flags & hip.EPI_ADD else None)
    amax_in = ops.new_amax("cuda", max(float(i.planes.abs().max()) for i in ins))
    tiny = torch.zeros(4, device="cuda")
    monkeypatch.setattr(ops, "splitk_workspace", lambda device: tiny)

    def run(lst, quads):
        outs = [FMap(64, H, W) for H, W in hws]
        for o, ad in zip(outs, addends):
            if ad is not None:
                o.from_dense(ad)
        am = ops.new_amax("cuda")
        probs = [(i, o, g) + ((c,) if unpool else ()) for i, o, g, c in zip(ins, outs, gates, codes if unpool else [None] * len(hws))]
        ops.conv3x3_grouped(probs, w, bias, flags, lst, 1.0, w2, amax_in, am, quads=quads)
        return outs, am
    ref, _ = run(None, False)                    # the ring kernel over the whole planes
    needs = _needs(hws, False, 9)
    lst = torch.cat([_quad_cover(ops, hip, nd, g) for g, nd in enumerate(needs)])
    outs, am = run(lst, True)
    true_max = 0.0
    for g, ((H, W), nd) in enumerate(zip(hws, needs)):
        o, r = outs[g], ref[g]
        cov = _covered_mask(hip, lst, g, H, W, o.plane)
        needed = torch.zeros(o.plane, dtype=torch.bool, device="cuda")
        needed[: (H + 2) * o.Wp].view(H + 2, o.Wp)[1:H + 1, 1:W + 1] = nd > 0
        assert bool((cov | ~needed).all()) and bool(needed.any())
        assert torch.equal(o.planes[:, cov], r.planes[:, cov])
        if addends[g] is None:
            assert float(o.planes[:, ~cov].abs().max()) == 0.0
            true_max = max(true_max, float(o.planes.abs().max()))
        else:   # outside the quads the addend stays as it was
            keep = FMap(64, H, W).from_dense(addends[g])
            assert torch.equal(o.planes[:, ~cov], keep.planes[:, ~cov])
            true_max = max(true_max, float(o.planes[:, cov].abs().max()))
        assert o.border_is_zero()
    assert float(am.max()) == true_max


@pytest.mark.parametrize("hws,two_masks", [([(37, 50)], True), ([(150, 201), (64, 85)], True), ([(40, 53)], False)])
def test_resident_kernel_with_gram_epilogue(hws, two_masks, monkeypatch):
    """conv1_2's data gradient as the step launches it: un-pooled input, relu1_1's Gram backward in the epilogue, its ReLU
    gate from the staged operand - against the two-launch form on the ring kernel (Gram backward, then EPI_ADD)."""
    require_gpu()
    import torch.nn.functional as F
    from stylemesh_amd.runtime import hip, ops
    from stylemesh_amd.runtime.fmap import FMap
    monkeypatch.setattr(ops, "CONV_MODE", "split2")
    monkeypatch.setattr(ops, "GRAM_MODE", "split2")
    C = 64
    torch.manual_seed(len(hws) * 7 + two_masks)
    wgt = torch.randn(C, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
    wd = ops.pack_conv_dgrad(wgt).cuda()
    wd2 = ops.pack_conv_split2(wd)
    D0 = (torch.randn(C, C) * 3e-3).cuda()
    D1 = (torch.randn(C, C) * 1e-3).cuda() if two_masks else None
    feats, masks, dps, codes = [], [], [], []
    for g, (H, W) in enumerate(hws):
        feats.append(FMap(C, H, W).from_dense(F.relu(torch.randn(C, H, W) * 2).cuda()))
        mk = torch.zeros(2, H, W)
        sel = torch.rand(H, W)
        mk[0] = (sel < 0.3).float()
        mk[1] = ((sel >= 0.3) & (sel < 0.45)).float()
        mk[:, H // 2:, : W // 3] = 0
        masks.append(FMap(2, H, W).from_dense(mk.cuda()))
        a, pooled = FMap(C, H, W).from_dense(F.relu(torch.randn(C, H, W)).cuda()), FMap(C, H // 2, W // 2)
        code = torch.zeros(C // 8 * pooled.plane, dtype=torch.int32, device="cuda")
        ops.maxpool_fwd_grouped([(a, pooled)], None, [code])
        codes.append(code)
        dps.append(FMap(C, H // 2, W // 2).from_dense((torch.randn(C, H // 2, W // 2) * 1e-4).cuda()))
    af = ops.new_amax("cuda", max(float(f.planes.abs().max()) for f in feats))
    ad = ops.new_amax("cuda", max(float(D0.abs().max()), float(D1.abs().max()) if two_masks else 0.0))
    amax_in = ops.new_amax("cuda", max(float(d.planes.abs().max()) for d in dps))

    def mptr(m, k):
        return m.channel_ptr(k)
    ref = [FMap(C, H, W) for (H, W) in hws]
    ws1 = [torch.empty(ops.gram_backward_ws_bytes(C), dtype=torch.uint8, device="cuda") for _ in hws]
    ops.gram_backward_grouped(ops.struct_array(hip.GramBwdProblem, [
        ops.gram_bwd_problem(f, mptr(m, 0), mptr(m, 1) if two_masks else None, D0, D1, r, w_, af, ad, relu_gate=False)
        for f, m, r, w_ in zip(feats, masks, ref, ws1)]))
    tiny = torch.zeros(4, device="cuda")
    with monkeypatch.context() as mp:
        mp.setattr(ops, "splitk_workspace", lambda device: tiny)
        ops.conv3x3_grouped([(d, r, f, c) for d, r, f, c in zip(dps, ref, feats, codes)], wd, None,
                            hip.EPI_RELU_MASK | hip.EPI_ADD, None, 1.0, wd2, amax_in, ops.new_amax("cuda"))
    needs = _needs(hws, False, 13)
    lst = torch.cat([_quad_cover(ops, hip, nd, g) for g, nd in enumerate(needs)])
    out = [FMap(C, H, W) for (H, W) in hws]
    ws2 = [torch.empty(ops.gram_backward_ws_bytes(C), dtype=torch.uint8, device="cuda") for _ in hws]
    ops.gram_backward_grouped(ops.struct_array(hip.GramBwdProblem, [
        ops.gram_bwd_problem(f, mptr(m, 0), mptr(m, 1) if two_masks else None, D0, D1, None, w_, af, ad, relu_gate=False)
        for f, m, w_ in zip(feats, masks, ws2)]))
    amax_out = ops.new_amax("cuda")
    ops.conv3x3_grouped([(d, o, f, c, None, None, (w_, mptr(m, 0), mptr(m, 1) if two_masks else None, af, ad))
                         for d, o, f, c, w_, m in zip(dps, out, feats, codes, ws2, masks)], wd, None,
                        hip.EPI_RELU_MASK | hip.EPI_GRAM, lst, 1.0, wd2, amax_in, amax_out, quads=True)
    for g, ((H, W), o, r) in enumerate(zip(hws, out, ref)):
        cov = _covered_mask(hip, lst, g, H, W, o.plane)
        assert torch.equal(o.planes[:, cov], r.planes[:, cov]) and int((o.planes != 0).sum()) > 0
        assert float(o.planes[:, ~cov].abs().max()) == 0.0
        assert o.border_is_zero()
    assert float(amax_out.max()) == max(float(o.planes.abs().max()) for o in out)


@pytest.mark.parametrize("env", [{}, {"STYLEMESH_FUSE_POOL_FWD": "0"}, {"STYLEMESH_FUSE_POOL_BWD": "0"},
                                 {"STYLEMESH_FUSE_GRAM_BWD": "0"}, {"STYLEMESH_SIDE_STREAMS": "0"},
                                 {"STYLEMESH_VALIDATE_LISTS": "1"}])   # (the engine's own quad lists pass ops.check_quad_list)
def test_engine_step_with_and_without_quad_lists(env, monkeypatch):
    """A multi-level step with the quad lists (resident-input kernel) and with the ring kernel's lists: the same losses and
    the same GRADIENT after one forward + backward from the same random texture (tests/stepcmp.py) - up to the operand
    scales: the quads list a few more dead positions, whose values may raise a tensor's recorded bound."""
    require_gpu()
    from golden_cases import MULTIVIEW_SEEDS
    from stepcmp import assert_same_pass, one_pass
    from test_round3_gpu import _engine, _small_view
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("STYLEMESH_RESIDENT", on)
        monkeypatch.setenv("STYLEMESH_OVERLAP_MIN_PIXELS", "0")     # the small test view takes the side-stream path
        for k, v in env.items():                                    # every epilogue variant the resident kernel is built for
            monkeypatch.setenv(k, v)
        if env.get("STYLEMESH_FUSE_POOL_BWD") == "0":               # (read once, when the module is imported)
            from stylemesh_amd.runtime import vgg
            monkeypatch.setattr(vgg, "FUSE_POOL_BWD", False)
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        eng = _engine(random_init=True)
        res[on] = one_pass(eng, _small_view(MULTIVIEW_SEEDS[0]))
        quads = getattr(eng.view_tiles, "quads", frozenset())
        assert (len(quads) == 3) == (on == "1"), quads          # conv1_2 forward / data gradient, conv2_1's data gradient
        if on == "1":
            assert (("conv1_2", "fp") in quads) == (env.get("STYLEMESH_FUSE_POOL_FWD") != "0" and env.get("STYLEMESH_FUSE_POOL_BWD") != "0")
    d = assert_same_pass(res["1"], res["0"], what=f"quad lists vs ring lists {env}")
    print(f"\n[quads {env}] max|dg| / max|g| = {d:.2e}")
