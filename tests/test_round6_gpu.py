"""Round 6: every composition switch of the step against the default composition on the one quantity that is stable - the
gradient after ONE forward + backward (tests/stepcmp.py); the flag-compacting update against the tile walk, bit for bit
through whole training steps; the operand census; the owner-aware exchange through the whole ``training_step`` path.
Reference path: model/model.py:178-327 (step), :387-395 (Adam), content_and_style_losses.py:47-70,288-350."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from conftest import REPO
from gpu_util import require_gpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env", [{"STYLEMESH_SEGMENT_LISTS": "0"}, {"STYLEMESH_SEGMENT_STARTS": "grid"}, {"STYLEMESH_FAST_VIEW": "0"},
                                 {"STYLEMESH_SIDE_STYLE": "r11,r21"}, {"STYLEMESH_SIDE_STYLE": ""}, {"STYLEMESH_EARLY_STYLE_AT": "r21"},
                                 {"STYLEMESH_SIDE_STREAMS": "inline"}, {"STYLEMESH_OVERLAP_MIN_PIXELS": "1000000000"},
                                 {"STYLEMESH_CONTENT_GRAPH": "0"}, {"STYLEMESH_PREPARE_AHEAD": "0"}])
def test_every_composition_switch_gives_the_default_gradient(env, monkeypatch):
    """One switch of runtime/config.py at a time (list granularity, grouped view path, which style branches fork where, side
    streams issued inline or not at all, the content pass eager): losses to 2e-6 and the gradient arena to 1e-5 max|g| of
    the default composition, from the same random texture."""
    require_gpu()
    from golden_cases import MULTIVIEW_SEEDS
    from stepcmp import assert_same_pass, one_pass
    from test_round3_gpu import _engine, _small_view
    res = []
    for e in ({}, env):
        with monkeypatch.context() as mp:
            mp.setenv("STYLEMESH_OVERLAP_MIN_PIXELS", "0")
            for k, v in e.items():
                mp.setenv(k, v)
            torch.manual_seed(11)
            torch.cuda.manual_seed(11)
            eng = _engine(random_init=True)
            res.append(one_pass(eng, _small_view(MULTIVIEW_SEEDS[0])))
    d = assert_same_pass(res[1], res[0], what=str(env))
    print(f"\n[{env}] max|dg| / max|g| = {d:.2e}")


_ADAM_WORKER = r'''
import os, sys, torch
sys.path[:0] = [sys.argv[2], os.path.join(sys.argv[2], "tests"), os.path.join(sys.argv[2], "oracle")]
os.environ.setdefault("STYLEMESH_OVERLAP_MIN_PIXELS", "0")
from golden_cases import MULTIVIEW_SEEDS
from test_round3_gpu import _engine, _small_view
eng = _engine()                                   # zero texture: the ever-touched sparse update, split in two halves
views = [_small_view(s) for s in MULTIVIEW_SEEDS[:2]]
grads = torch.load(sys.argv[3]) if os.path.exists(sys.argv[3]) else None
made = []
for k in range(8):
    v = views[k // 4]
    eng.begin_step(v)
    eng._step_begin()
    eng._adam_early()
    if grads is None:
        eng.forward_backward()
        made.append(eng.arena.g.clone().cpu())
    else:
        eng.forward_backward()                    # (keeps the engine's bookkeeping in step; the gradient is replaced)
        eng.arena.g.copy_(grads[k].cuda())
    eng.optimizer_step()
    if k == 3:
        eng.end_epoch()
torch.cuda.synchronize()
if grads is None:
    torch.save(made, sys.argv[3])
torch.save({"p": eng.arena.p.cpu(), "m": eng.arena.m.cpu(), "v": eng.arena.v.cpu(), "g": eng.arena.g.cpu(),
            "sumsq": eng.sumsq.cpu(), "walk": os.environ.get("SM_ADAM_DENSE_WALK", "")}, sys.argv[1])
'''


def test_flag_compacting_update_equals_the_tile_walk_bit_for_bit_over_training_steps():
    """The fused update over flagged chunks (round 6: a block compacts the flags of its span, then streams the listed
    chunks) against the tile walk of rounds 2-5 (SM_ADAM_DENSE_WALK=1, a library-wide switch read once: two child
    processes), both halves of the split update, eight steps over two views with a learning-rate decay, fed the SAME
    gradients: p, m, v and the zeroed gradient identical to the bit, sum p^2 to summation order."""
    require_gpu()
    with tempfile.TemporaryDirectory() as tmp:
        outs = {}
        for name, extra in (("span", {}), ("walk", {"SM_ADAM_DENSE_WALK": "1"})):
            env = {k: v for k, v in os.environ.items() if k != "SM_ADAM_DENSE_WALK"}
            env.update(extra)
            out = os.path.join(tmp, name + ".pt")
            r = subprocess.run([sys.executable, "-c", _ADAM_WORKER, out, REPO, os.path.join(tmp, "grads.pt")], cwd=REPO, env=env,
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
            outs[name] = torch.load(out)
    a, b = outs["span"], outs["walk"]
    assert a["walk"] == "" and b["walk"] == "1"
    for k in ("p", "m", "v", "g"):
        assert torch.equal(a[k], b[k]), k
    assert float(a["g"].abs().max()) == 0.0 and float(a["p"].abs().max()) > 0
    np.testing.assert_allclose(a["sumsq"].numpy(), b["sumsq"].numpy(), rtol=1e-5)


def test_operand_census_of_a_small_step():
    """stylemesh_amd/diagnostics.py on a two-level step: every fp16x2 operand tensor is listed, its recorded bound is an
    upper bound of its true maximum and within a factor 2 of it (the bound is the producer's max |x|), the shares beyond
    2^k decrease with k."""
    require_gpu()
    from golden_cases import MULTIVIEW_SEEDS
    from stylemesh_amd.diagnostics import operand_census, summarize
    from test_round3_gpu import _engine, _small_view
    torch.manual_seed(5)
    eng = _engine(random_init=True)
    eng.sparse_tiles = False
    eng.set_view(_small_view(MULTIVIEW_SEEDS[0]))
    eng.arena.g.zero_()
    eng.forward_backward()
    c = operand_census(eng)
    acts = [k for k in c if k.startswith("a:")]
    grads = [k for k in c if k.startswith("g:")]
    ds = [k for k in c if k.startswith("D:")]
    assert len(acts) >= 12 and len(grads) >= 8 and len(ds) == 2 * 5          # two levels x five style layers
    for k, e in c.items():
        assert e["bound"] >= e["true_max"] > 0, k
        if not k.startswith("D:"):
            assert e["bound"] <= 2.0 * e["true_max"] * 1.0001 or k.startswith("a:p"), (k, e["bound"], e["true_max"])
        sh = [e["share_beyond_2^k"][str(t)] for t in (12, 16, 18, 20, 22, 24)]
        assert all(x >= y for x, y in zip(sh, sh[1:])) and 0.0 <= sh[0] <= 1.0, k
        assert sum(e["histogram_log2_bound_over_x"]) == e["nonzero_elements"]
    s = summarize(c)
    assert s["tensors"] == len(c) and 0.0 <= s["worst_share_beyond_2^18"] < 0.5


def test_two_rank_trainer_with_the_deferred_exchange_ends_with_identical_textures():
    """MiniTrainer + LightningModule mirror + OwnerAwareGradReducer (STYLEMESH_DEFERRED_EXCHANGE=1) over an odd view count on
    two ranks (gloo, one device): the deferred sums are drained at view changes and at the epoch's end - both ranks finish
    with the same texture, bit for bit, as they do with the plain exchange."""
    require_gpu()
    from test_round2_gpu import _launch_ranks
    with tempfile.TemporaryDirectory() as tmp:
        r = _launch_ranks([os.path.join(REPO, "tests", "two_rank_worker.py"), "trainer", tmp], 2,
                          {"STYLEMESH_TEST_BACKEND": "gloo", "STYLEMESH_DEFERRED_EXCHANGE": "1"}, timeout=1200)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        r0, r1 = (torch.load(os.path.join(tmp, f"rank{k}.pt")) for k in (0, 1))
    assert r0["steps"] == r1["steps"] == 6 and r0.get("deferred") is True
    for a, b in zip(r0["layers"], r1["layers"]):
        assert torch.equal(a, b) and float(a.abs().max()) > 0
