"""Round-2 additions, on the GPU: the ever-touched sparse update, the device-side Adam step state of the graph path,
checkpoint / resume, the engine in fp32-MFMA mode against the reference golden, the product's own RCCL communicator
(one rank always; two ranks when the box has two GPUs) and the two-rank view-sharded trainer path over gloo on one
device with an ODD view count."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

import stylemesh_oracle as O
from conftest import REPO, batch_from_golden, load_golden
from golden_cases import FLAGSETS, SMALL_LEVEL_HW, SMALL_ROOM, SMALL_VIEW_HW
from gpu_util import assert_close, require_gpu
from stylemesh_amd.data import synthetic as S
from test_engine_gpu import full_grads, grad_close, make_engine

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def small_views(seeds):
    return [S.make_view(s, view_hw=SMALL_VIEW_HW, level_hw=SMALL_LEVEL_HW, level_heights=[40, 64], min_pyramid_depth=0.9,
                        room=S.BoxRoom(SMALL_ROOM)) for s in seeds]


# ------------------------------------------------------------------ K7 over ever-touched chunks
def test_adam_kernel_skips_unflagged_chunks_only():
    require_gpu()
    from stylemesh_amd.runtime import ops
    torch.manual_seed(3)
    sizes = [3 * 64 * 64, 3 * 32 * 32, 3 * 16 * 16, 3 * 8 * 8]
    seg_end = np.cumsum(sizes).tolist()
    n = seg_end[-1]
    reg = [0.41, 0.2, 0.05, 0.0]
    flags = (torch.rand(n // 64) < 0.3).to(torch.int32).cuda()
    live = flags.bool().repeat_interleave(64)
    p0 = (torch.randn(n) * 20).cuda() * live          # zero wherever the chunk is unflagged: the exactness condition
    g0 = torch.randn(n).cuda() * live
    res = []
    for touched in (None, flags):
        P, G, M, V = p0.clone(), g0.clone(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
        sumsq = torch.zeros(4).cuda()
        for step in (1, 2, 3):
            G.copy_(g0 * step)
            sumsq.zero_()
            ops.adam_fused(P, G, M, V, seg_end, reg, 0.7, step, sumsq_out=sumsq, touched=touched, touched_log2=6)
        res.append((P, M, V, G, sumsq))
    for a, b in zip(res[0][:4], res[1][:4]):
        assert torch.equal(a, b)                      # bit-identical: skipped elements are exact zeros either way
    assert_close(res[1][4], res[0][4], 1e-6, 0)
    assert float(res[1][0][~live].abs().max()) == 0.0
    # a flagged chunk with non-zero content IS updated, an unflagged one with non-zero content is left alone
    P = torch.ones(n).cuda()
    G, M, V = torch.zeros(n).cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    ops.adam_fused(P, G, M, V, seg_end, reg, 0.7, 1, touched=flags, touched_log2=6)
    moved = P[:sizes[0]] != 1.0
    assert torch.equal(moved, live[:sizes[0]])


def test_sparse_update_equals_dense_and_touched_texels_keep_decaying():
    """Zero-initialised texture, view A then view B. (i) Every step's update over the ever-touched chunks is
    bit-identical to the update over the whole arena (both applied to the SAME state and gradient: the Gram kernels'
    atomics make two forward / backward passes differ in the last bits, so two free-running engines cannot be compared
    bit for bit); (ii) a texel only view A reaches keeps moving under the regulariser (and its Adam moments) during
    view B's steps - it was touched once, so it stays in the update; (iii) texels no view has reached stay exactly
    zero with zero moments."""
    views = small_views((3, 4))
    eng = make_engine(FLAGSETS["with_angle_and_depth"], None)
    assert eng.touched is not None
    a = eng.arena
    snaps = []
    for v in views:
        for _ in range(3):
            eng.begin_step(v)
            eng.forward_backward()
            state = [t.clone() for t in (a.p, a.g, a.m, a.v, eng.sumsq)]
            count = eng.step_count
            eng.sparse_update = True
            eng.optimizer_step()
            sparse = [t.clone() for t in (a.p, a.g, a.m, a.v, eng.sumsq)]
            for dst, src in zip((a.p, a.g, a.m, a.v, eng.sumsq), state):
                dst.copy_(src)
            eng.step_count = count
            eng.sparse_update = False
            eng.optimizer_step()
            for got, want in zip(sparse[:4], (a.p, a.g, a.m, a.v)):
                assert torch.equal(got, want)
            assert_close(sparse[4], eng.sumsq, 1e-6, 0)
        snaps.append((a.p.clone(), a.m.clone(), a.v.clone(), eng.touched.clone()))
    (pA, _, _, tA), (pB, mB, vB, tB) = snaps
    assert int(tA.sum()) < int(tB.sum()) < tB.numel()          # B adds chunks, not everything is touched
    only_a = (pA != 0) & (tA.bool().repeat_interleave(64)[:pA.numel()])
    assert bool(only_a.any())
    changed = (pB != pA) & only_a
    assert float(changed.float().sum()) > 0.5 * float(only_a.float().sum())   # still decaying / moving in view B
    never = ~tB.bool().repeat_interleave(64)[:pB.numel()]
    assert bool(never.any()) and float(pB[never].abs().max()) == 0.0
    assert float(mB[never].abs().max()) == 0.0 and float(vB[never].abs().max()) == 0.0


def test_split_update_equals_dense():
    """The split update (chunks the view cannot reach updated at the HEAD of the step on a side stream, the view's own
    chunks after the scatter) leaves exactly the state of one dense update with the same gradient - over two views, so
    that 'ever touched but not in this view' is a non-empty set."""
    views = small_views((3, 4))
    eng = make_engine(FLAGSETS["with_angle_and_depth"], None)
    assert eng.touched is not None and eng.split_update
    a = eng.arena
    n_other = 0
    for v in views:
        for _ in range(3):
            eng.begin_step(v)
            before = [t.clone() for t in (a.p, a.m, a.v)]
            count = eng.step_count
            eng._adam_early()                      # first half: beside the forward pass
            assert eng._adam_early_done is not None
            n_other += int(eng._other_flags[1].sum())
            eng.forward_backward()
            g = a.g.clone()
            eng.optimizer_step()                   # second half
            split = [t.clone() for t in (a.p, a.g, a.m, a.v)]
            ssq = eng.sumsq.clone()
            for dst, src in zip((a.p, a.m, a.v), before):
                dst.copy_(src)
            a.g.copy_(g)
            eng.step_count = count
            eng.sparse_update = False
            eng.optimizer_step()                   # one dense update of the same state and gradient
            eng.sparse_update = True
            for got, want in zip(split, (a.p, a.g, a.m, a.v)):
                assert torch.equal(got, want)
            assert_close(ssq, eng.sumsq, 1e-6, 0)
    assert n_other > 0                             # the second view left chunks of the first one to the early half


def test_load_texture_switches_to_dense_update():
    g5 = load_golden("g5_with_angle_and_depth")
    eng = make_engine(FLAGSETS["with_angle_and_depth"], [T(g5[f"init{i}"]) for i in range(4)])
    assert eng.touched is None      # arbitrary initial content: every texel takes part (regulariser decay)
    before = eng.arena.p.clone()
    eng.training_step(batch_from_golden(g5))
    assert float((eng.arena.p != before).float().mean()) > 0.9


# ------------------------------------------------------------------ graph replay without per-step host syncs
def test_graph_steps_run_ahead_of_the_gpu_with_correct_bias_corrections():
    """with hipGraph replay the host enqueues many steps before the GPU runs them; the step-dependent
    Adam scalars must not travel through a host buffer a later step overwrites. Run 12 steps WITHOUT reading anything
    back (plus a learning-rate decay in the middle) and compare with eager stepping."""
    from stepcmp import check_deviation, lock, step_deviation
    g5 = load_golden("g5_with_angle_and_depth")
    init = [T(g5[f"init{i}"]) for i in range(4)]
    batch = batch_from_golden(g5)
    engs = {}
    for graphs in (False, True):
        eng = engs[graphs] = make_engine(FLAGSETS["with_angle_and_depth"], init)
        eng.use_graphs = graphs
        eng.cfg.decay_step_size = 1
    # LOCK-STEP without a single read-back (tests/stepcmp.py): the graph engine receives the eager engine's state by
    # stream-ordered device copies, both steps are enqueued, the step's deviation stays on the device until the end
    devs, lrs = [], []
    for step in range(12):
        m0, v0 = lock(engs[True], engs[False])
        for eng in engs.values():
            eng.training_step(batch)
        devs.append(step_deviation(engs[True], engs[False], m0, v0))
        lrs.append(engs[False].lr)
        if step == 6:
            for eng in engs.values():
                eng.end_epoch()
    torch.cuda.synchronize()
    eng = engs[True]
    assert eng._opt_graph is not None
    lr_dev, step_dev = eng._hyper_state.tolist()     # the device-side {lr, step} followed both changes
    assert abs(lr_dev - 0.1) < 1e-12 and step_dev == 12.0
    assert engs[True].step_count == engs[False].step_count == 12
    # a wrong bias correction (a later step's) or learning rate would scale EVERY update of that step: the texture on the
    # well-conditioned texels (dev[3]) would be off by ~lr, the moments are untouched by it
    for step, (dev, lr) in enumerate(zip(devs, lrs)):
        check_deviation(dev, lr, what=f"step {step}")
        assert float(dev[4]) > 0.5, (step, float(dev[4]))          # most texels are well-conditioned: the check has teeth


# ------------------------------------------------------------------ checkpoint / resume
def test_optimizer_state_dict_resume_round_trip():
    from test_model_surface_gpu import make_model, to_cuda
    g5 = load_golden("g5_with_angle_and_depth")
    batch = to_cuda(batch_from_golden(g5))

    def run(model, opt, sched, steps, start=0):
        for step in range(start, start + steps):
            opt.zero_grad()
            model.training_step(batch, step)["loss"].backward()
            opt.step()
            if step % 2 == 1:
                sched.step()

    m1 = make_model(FLAGSETS["with_angle_and_depth"], None)
    (o1,), (s1,) = m1.configure_optimizers()
    run(m1, o1, s1, 3)
    ckpt = {"texture": [l.data.detach().clone() for l in m1.texture.layers], "opt": o1.state_dict(),
            "sched_epoch": s1.last_epoch}
    assert ckpt["opt"]["touched"] is not None
    run(m1, o1, s1, 3, start=3)

    m2 = make_model(FLAGSETS["with_angle_and_depth"], [t.cpu() for t in ckpt["texture"]])   # the saved texture
    (o2,), (s2,) = m2.configure_optimizers()
    o2.load_state_dict(ckpt["opt"])
    s2.last_epoch = ckpt["sched_epoch"]
    assert m2._engine.touched is not None and m2._engine.step_count == 3
    # the restored state is the saved state, bit for bit
    assert torch.equal(m2._engine.arena.m, ckpt["opt"]["m"]) and torch.equal(m2._engine.arena.v, ckpt["opt"]["v"])
    assert torch.equal(m2._engine.touched, ckpt["opt"]["touched"]) and o2.param_groups[0]["lr"] == ckpt["opt"]["lr"]
    for a, b in zip(m2.texture.layers, ckpt["texture"]):
        assert torch.equal(a.data, b)
    run(m2, o2, s2, 3, start=3)
    # ... and the resumed run follows the uninterrupted one (two runs differ in the last bits of the Gram sums -
    # atomics - which Adam at lr 1 amplifies: the tolerance of the 5-step Adam goldens)
    from test_engine_gpu import texture_close
    for i, (a, b) in enumerate(zip(m1.texture.layers, m2.texture.layers)):
        texture_close(b.data, a.data.cpu(), 3, f"resumed layer {i}")
    assert m1._engine.step_count == m2._engine.step_count == 6
    rel = float((m1._engine.arena.m - m2._engine.arena.m).abs().max() / m1._engine.arena.m.abs().max())
    assert rel < 2e-2, rel


# ------------------------------------------------------------------ the engine with fp32-MFMA kernels everywhere
@pytest.mark.parametrize("name", ["with_angle_and_depth", "only2d"])
def test_engine_f32_mode_matches_reference_golden(name, monkeypatch):
    from stylemesh_amd.runtime import ops
    monkeypatch.setattr(ops, "CONV_MODE", "f32")
    monkeypatch.setattr(ops, "GRAM_MODE", "f32")
    d = load_golden("g5_" + name)
    eng = make_engine(FLAGSETS[name], [T(d[f"init{i}"]) for i in range(4)])
    eng.set_view(batch_from_golden(d))
    eng.arena.g.zero_()
    lt = eng.loss_tensors()
    eng.forward_backward()
    losses = eng.losses(lt)
    for k in ("content", "style", "tex_reg", "total"):
        np.testing.assert_allclose(losses[k], float(d[f"loss_{k}"].reshape(-1)[0]), rtol=2e-4, err_msg=k)
    for i, g in enumerate(full_grads(eng)):
        grad_close(g, d[f"grad{i}"], f"f32 mode {name} grad{i}")


# ------------------------------------------------------------------ RCCL
def test_own_rccl_communicator_single_rank():
    """sm_comm_init / sm_allreduce_grad / sm_allreduce_flags_max / sm_comm_destroy with one rank: the library links
    and drives RCCL on the caller's stream (an all-reduce over one rank is the identity)."""
    require_gpu()
    from stylemesh_amd.runtime.distributed import RcclComm, SparseGradReducer
    comm = RcclComm(None, 0, 1, torch.device("cuda", 0))
    g = torch.randn(1 << 16, device="cuda")
    ref = g.clone()
    comm.all_reduce(g, op=comm.ReduceOp.SUM)
    flags = (torch.rand(1024, device="cuda") < 0.3).to(torch.int32)
    fref = flags.clone()
    comm.all_reduce(flags, op=comm.ReduceOp.MAX)
    w = comm.all_reduce(g, op=comm.ReduceOp.SUM, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    assert torch.equal(g, ref) and torch.equal(flags, fref)
    red = SparseGradReducer(comm, 1, chunk_log2=6)
    red.new_view(flags.clone())
    red(g)
    torch.cuda.synchronize()
    assert torch.equal(g, ref) and red.last_bytes == int(fref.sum()) * 64 * 4
    comm.destroy()


def _launch_ranks(script_args, nproc, env_extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args
    return subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_rank_rccl_gradient_step_matches_golden():
    """Two ranks, two GPUs, the product's own RCCL communicator: each rank accumulates the HIP gradients of its two
    views of golden G8, the arenas are all-reduced over RCCL, every rank applies the update with grad_scale 1/4 and
    lands on the reference-generated mean-gradient step."""
    require_gpu()
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU box); the one-device gloo variant below always runs")
    with tempfile.TemporaryDirectory() as tmp:
        r = _launch_ranks([os.path.join(REPO, "tests", "two_rank_worker.py"), "g8", tmp], 2, {"STYLEMESH_TEST_BACKEND": "nccl"})
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        _check_g8(tmp)


def test_two_rank_gloo_one_device_gradient_step_matches_golden():
    """The same protocol with both ranks on cuda:0 and the exchange over gloo (the 1-GPU box cannot run RCCL between
    two ranks): the HIP gradients, the sparse reducer and the fused update are the real ones."""
    require_gpu()
    with tempfile.TemporaryDirectory() as tmp:
        r = _launch_ranks([os.path.join(REPO, "tests", "two_rank_worker.py"), "g8", tmp], 2, {"STYLEMESH_TEST_BACKEND": "gloo"})
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        _check_g8(tmp)


def _check_g8(tmp):
    from test_engine_gpu import texture_close
    d = load_golden("g8_multiview")
    r0, r1 = (torch.load(os.path.join(tmp, f"rank{r}.pt")) for r in (0, 1))
    for i in range(4):
        grad_close(r0["mean_grad"][i], d[f"mean_grad{i}"], f"2-rank mean grad {i}")
        assert torch.equal(r0["layers"][i], r1["layers"][i])          # identical update on every rank
        texture_close(r0["layers"][i], T(d[f"p{i}_after"]).clamp(O.CLAMP_LO, O.CLAMP_HI), 1, f"2-rank layer {i}")
    assert r0["exchange"] == r1["exchange"]


def test_two_rank_trainer_with_odd_view_count_over_gloo_one_device():
    """MiniTrainer + training_step + SparseGradReducer with 5 train views on 2 ranks (index_repeat
    2): rank 1's shard is padded by repeating its last view; the per-view collective follows the schedule position, so
    both ranks finish the epoch with identical textures instead of dead-locking."""
    require_gpu()
    with tempfile.TemporaryDirectory() as tmp:
        r = _launch_ranks([os.path.join(REPO, "tests", "two_rank_worker.py"), "trainer", tmp], 2,
                          {"STYLEMESH_TEST_BACKEND": "gloo"}, timeout=1200)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        r0, r1 = (torch.load(os.path.join(tmp, f"rank{r}.pt")) for r in (0, 1))
        assert r0["steps"] == r1["steps"] == 6
        for a, b in zip(r0["layers"], r1["layers"]):
            assert torch.equal(a, b) and float(a.abs().max()) > 0
        assert os.path.exists(os.path.join(tmp, "lightning_logs/version_0/scalars.jsonl"))
        assert os.path.exists(os.path.join(tmp, "lightning_logs/version_0/scalars.rank1.jsonl"))
