"""Round 5: step programs scoped to one engine's calls, the pipelined two-rank exchange (bit for bit) and the
library's own radix sort. (The pair-image tests of this file left with the pair images in round 6: DESIGN.md section 9.)"""
import numpy as np
import pytest
import torch

from gpu_util import require_gpu

pytestmark = pytest.mark.gpu


def test_step_program_recording_is_scoped_to_the_engines_own_calls():
    """the recorder stands in for the module-global ``ops.lib`` only while ONE engine issues its own
    calls. Two engines whose steps interleave (a.compute, b.compute, b.update, a.update) and a caller that launches
    library work between ``step_compute`` and ``optimizer_step`` (a hook rendering the texture) must each get programs
    with their own update segment - every step still takes its Adam update and equals the unrecorded engines' step."""
    require_gpu()
    import test_round4_gpu as R4
    from stepcmp import assert_same_step, lock
    from stylemesh_amd.runtime import hip, ops
    c = R4.PROGRAM_CASES["only2D"]
    views = R4._small_views((0, 2))
    a, b = R4._program_engine(c, "1"), R4._program_engine(c, "1")
    ra, rb = R4._program_engine(c, "0"), R4._program_engine(c, "0")
    scratch = torch.zeros(64, device="cuda")
    for i in range(14):
        batch = views[i // 7]
        start = {}
        for e, r in ((a, ra), (b, rb)):
            start[id(e)] = lock(e, r)      # lock-step with the unrecorded twins (tests/stepcmp.py)
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
            assert_same_step(e, r, *start[id(e)], what=f"step {i}")
    for e in (a, b):
        assert e.program_replays >= 4, e.program_replays
        for prog in e._programs.values():
            assert "sm_zero_floats" not in prog.names[prog.n_compute:] or prog.names.count("sm_adam_fused") >= 1
            assert any(n == "sm_adam_fused" for n in prog.names[prog.n_compute:])


def test_two_rank_pipelined_exchange_equals_exchange_then_update_bit_for_bit():
    """the pipelined exchange + update (opt-in; 'auto' = from 32 MB of flagged chunks on) against
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


def test_two_rank_deferred_exchange_equals_exchange_then_update_bit_for_bit():
    """single-owner chunks off the step's critical path (``OwnerAwareGradReducer`` +
    ``StepEngine.exchange_and_update_deferred``) against the union exchange followed by one update, with the real kernels on
    two ranks (one device, gloo), two views x three steps and a learning-rate decay in between: after the drain p, m and v
    are identical to the bit, on each rank and across the ranks; the gradient arena is zero; the critical exchange carried
    less than the union."""
    require_gpu()
    import os
    import tempfile
    from conftest import REPO
    from test_round2_gpu import _launch_ranks
    with tempfile.TemporaryDirectory() as tmp:
        r = _launch_ranks([os.path.join(REPO, "tests", "two_rank_worker.py"), "deferred", tmp], 2,
                          {"STYLEMESH_TEST_BACKEND": "gloo"}, timeout=1200)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        r0, r1 = (torch.load(os.path.join(tmp, f"rank{k}.pt")) for k in (0, 1))
    for res in (r0, r1):
        assert res["steps"] == 6 and float(res["g"].abs().max()) == 0.0
        for a, b in zip(res["A"][:3], res["B"][:3]):
            assert torch.equal(a, b)
        # (sum p^2 - the regulariser loss VALUE - of a non-owner holds the other rank's single-owner chunks one update late)
        np.testing.assert_allclose(res["B"][3].numpy(), res["A"][3].numpy(), rtol=5e-2)
        assert float(res["A"][0].abs().max()) > 0
    for x, y in zip(r0["B"][:3], r1["B"][:3]):
        assert torch.equal(x, y)
    assert r0["stats"] == r1["stats"]
    assert all(c + d == u for c, d, u in r0["stats"]) and any(d > 0 for _, d, _ in r0["stats"])
