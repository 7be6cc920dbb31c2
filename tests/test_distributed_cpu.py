"""The N > 1 protocol on CPU: 2 gloo ranks shard 4 views, accumulate per-view gradients (computed by the ORACLE
here - there is no GPU), all-reduce the flat gradient arena, apply the update with grad_scale = 1/R, and land on
the reference-generated golden G8 (mean of 4 independent B = 1 gradients + one Adam step)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, load_golden
from golden_cases import FLAGSETS, MULTIVIEW_SEEDS, SMALL_LEVEL_HW, SMALL_ROOM, SMALL_VIEW_HW
from stylemesh_amd.runtime import distributed as D


def test_sharding_and_schedule():
    assert D.shard_views(range(7), 0, 2) == [0, 2, 4, 6]
    assert D.shard_views(range(7), 1, 2) == [1, 3, 5]
    assert D.steps_per_epoch(7, 20, 2) == 80
    s0, s1 = D.rank_schedule(range(7), 0, 2, 3), D.rank_schedule(range(7), 1, 2, 3)
    assert len(s0) == len(s1) == 12 and s1[-3:] == [5, 5, 5] and s0[:6] == [0, 0, 0, 2, 2, 2]
    with pytest.raises(ValueError):
        D.rank_schedule(range(1), 1, 2, 1)
    assert D.make_grad_reducer(dist, 1) is None


def _worker(rank, world, port, out_dir):
    for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import stylemesh_oracle as O
    from stylemesh_amd.data import synthetic as S
    from test_oracle_vs_golden import make_oracle
    g5 = load_golden("g5_with_angle_and_depth")
    init = [torch.from_numpy(g5[f"init{i}"]) for i in range(4)]
    pipe = make_oracle(FLAGSETS["with_angle_and_depth"], init)
    sizes = [l.numel() for l in pipe.layers]
    arena = torch.zeros(sum(sizes))                       # the flat gradient arena of this rank
    reg = [torch.zeros_like(l) for l in pipe.layers]
    for seed in D.shard_views(MULTIVIEW_SEEDS, rank, world):
        batch = S.make_view(seed, view_hw=SMALL_VIEW_HW, level_hw=SMALL_LEVEL_HW, level_heights=[40, 64],
                            min_pyramid_depth=0.9, room=S.BoxRoom(SMALL_ROOM))
        _, grads = pipe.grads(batch)
        # the data term only: the regulariser's gradient is a function of p and is added locally after the reduce
        reg = [2 * 5e3 * w / l.numel() * l.detach() for w, l in zip([8, 4, 2, 0], pipe.layers)]
        arena += torch.cat([(g - r).reshape(-1) for g, r in zip(grads, reg)])
    D.make_grad_reducer(dist, world)(arena)
    R = len(MULTIVIEW_SEEDS)
    full = [a.view_as(l) / R + r for a, l, r in zip(arena.split(sizes), pipe.layers, reg)]
    pipe.apply_adam(full)
    torch.save({"grads": full, "layers": [l.detach().clone() for l in pipe.layers]}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_gloo_mean_gradient_step(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d = load_golden("g8_multiview")
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in (0, 1))
    for i in range(4):
        ref = d[f"mean_grad{i}"]
        np.testing.assert_allclose(r0["grads"][i].numpy(), ref, rtol=1e-4, atol=2e-5 * float(np.abs(ref).max()))
        assert torch.equal(r0["layers"][i], r1["layers"][i])          # every rank applies the identical update
        bad = (r0["layers"][i] - torch.from_numpy(d[f"p{i}_after"])).abs() > 2e-3
        assert bad.sum() <= max(3, 2e-3 * bad.numel())


def _sparse_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    red = D.make_sparse_grad_reducer(dist, world, chunk_log2=4)
    g = torch.Generator().manual_seed(100 + rank)
    n_chunks, chunk = 64, 16
    out = {}
    for view in range(2):                                   # two "views": the dirty set changes between them
        touched = torch.rand(n_chunks, generator=g) < (0.2 if view == 0 else 0.9)   # view 1 -> dense fallback
        flags = touched.to(torch.int32)
        red.new_view(flags)
        for step in range(2):
            arena = torch.zeros(n_chunks, chunk)
            arena[touched] = torch.randn(int(touched.sum()), chunk, generator=g)
            arena = arena.reshape(-1)
            dense = arena.clone()
            dist.all_reduce(dense)
            red(arena)
            out[(view, step)] = (arena.clone(), dense, red.fraction, red.last_bytes)
            # pipelined exchange + update: ranges must tile the arena in order and see the summed gradient
            local = torch.zeros(n_chunks, chunk)
            local[touched] = torch.randn(int(touched.sum()), chunk, generator=g)
            local = local.reshape(-1)
            want = local.clone()
            dist.all_reduce(want)
            param, ranges = torch.ones(local.numel()), []

            def update_range(lo, hi):
                ranges.append((lo, hi))
                param[lo:hi] -= 0.1 * local[lo:hi]
                local[lo:hi] = 0
            red.pipelined(local, update_range)
            out[("pipe", view, step)] = (param, 1.0 - 0.1 * want, ranges, int(local.abs().sum()), red.last_bytes)
    torch.save(out, os.path.join(out_dir, f"sparse{rank}.pt"))
    dist.destroy_process_group()


def test_sparse_grad_reducer_equals_dense_allreduce(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_sparse_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"sparse{r}.pt") for r in (0, 1))
    for key in [k for k in r0 if k[0] == "pipe"]:
        param, want, ranges, left, nbytes = r0.pop(key)
        r1.pop(key)
        assert torch.equal(param, want) and left == 0
        assert ranges[0][0] == 0 and ranges[-1][1] == 64 * 16 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        assert len(ranges) == (4 if key[1] == 0 else 1)      # view 1 is the dense fallback of a 1024-float arena
    for key in r0:
        mine, dense, frac, nbytes = r0[key]
        assert torch.equal(mine, dense) and torch.equal(mine, r1[key][0])
        if key[0] == 0:
            assert frac < 0.6 and nbytes == int(round(frac * 64)) * 16 * 4      # only the dirty chunks travelled
        else:
            assert nbytes == 64 * 16 * 4                                        # mostly dirty: dense fallback


def test_scheduled_batches_mark_view_changes_by_position():
    """Every rank sees ``new_view`` at the same schedule positions, padded ranks included (a padded rank
    keeps its view key while the others change theirs, and must still enter the per-view collective)."""
    scheds = []
    for rank in range(2):
        seq = list(D.scheduled_batches(lambda i: (f"view{i}",), range(5), rank, 2, index_repeat=2))
        scheds.append([(b[0], b.new_view) for b in seq])
    assert [v for v, _ in scheds[0]] == ["view0", "view0", "view2", "view2", "view4", "view4"]
    assert [v for v, _ in scheds[1]] == ["view1", "view1", "view3", "view3", "view3", "view3"]   # padded: view3 again
    assert [f for _, f in scheds[0]] == [f for _, f in scheds[1]] == [True, False] * 3
    seq = list(D.scheduled_batches(lambda i: (i,), range(3), 0, 1, index_repeat=4, repeat=False))
    assert [(b[0], b.new_view) for b in seq] == [(0, True), (1, True), (2, True)]
    # repeats of one view are the same host object after the first (one decode / upload per view)
    seq = list(D.scheduled_batches(lambda i: (object(),), range(2), 0, 1, index_repeat=3))
    assert seq[1] is seq[2] and seq[0][0] is seq[1][0] and seq[3][0] is not seq[0][0]


def _schedule_worker(rank, world, port, out_dir):
    """The real ``StepEngine.begin_step`` + ``SparseGradReducer`` over an ODD view count: 5 views, 2 ranks, repeat 2.
    Only the kernels are stubbed (no GPU here): ``set_view`` records the key, ``touch_flags`` flags a key-dependent
    chunk set, the 'gradient' is the view key in the view's chunks."""
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stylemesh_amd.runtime.engine import StepEngine
    n_chunks, chunk = 32, 16
    eng = object.__new__(StepEngine)
    eng.view, eng.view_key, eng._last_batch, eng.touched = None, None, None, None
    log = []

    def set_view(batch, reducer=None):
        eng.view, eng.view_key, eng._last_batch = [batch], batch[8], batch
        log.append(("set_view", batch[8]))
        if reducer is not None:     # the real set_view enters the per-view collective itself when it is due
            flags = touch_flags(reducer.chunk_log2)
            reducer.new_view_end(int(reducer.new_view_begin(flags)))
            eng._union_flags = flags

    def touch_flags(chunk_log2, levels=None):
        f = torch.zeros(n_chunks, dtype=torch.int32)
        f[(eng.view_key * 5) % n_chunks:(eng.view_key * 5) % n_chunks + 4] = 1
        return f
    eng.set_view, eng.touch_flags = set_view, touch_flags
    red = D.make_sparse_grad_reducer(dist, world, chunk_log2=4)
    sums = []
    get_view = lambda i: tuple([None] * 8 + [i] + [None] * 4)
    for batch in D.scheduled_batches(get_view, range(5), rank, world, index_repeat=2):
        eng.begin_step(batch, red)
        arena = torch.zeros(n_chunks, chunk)
        arena[touch_flags(4).bool()] = float(batch[8] + 1)
        arena = arena.reshape(-1)
        want = arena.clone()
        dist.all_reduce(want)
        red(arena)
        assert torch.equal(arena, want)
        sums.append(float(arena.sum()))
    torch.save({"log": log, "sums": sums}, os.path.join(out_dir, f"sched{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_odd_view_count_keeps_collectives_matched(tmp_path):
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_schedule_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"sched{r}.pt") for r in (0, 1))
    assert [k for _, k in r0["log"]] == [0, 2, 4] and [k for _, k in r1["log"]] == [1, 3]   # rank 1: no 3rd set_view
    assert r0["sums"] == r1["sums"] and len(r0["sums"]) == 6


def _comm_worker(rank, world, port, out_dir):
    """``RcclComm`` / ``make_comm`` with the library's communicator entry points stubbed so that ``sm_comm_init`` fails
    on rank 1 ONLY: both ranks must come out the same way (a per-rank fallback left the ranks in different
    collectives)."""
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stylemesh_amd.runtime import hip
    destroyed = []

    class Lib:
        def __init__(self, fail_init_on, fail_id=False):
            self.fail_init_on, self.fail_id = fail_init_on, fail_id

        def sm_comm_unique_id_bytes(self):
            return 128

        def sm_comm_get_unique_id(self, buf):
            return 1 if self.fail_id else 0

        def sm_comm_init(self, handle_ref, world_size, uid, r):
            if r in self.fail_init_on:
                return 1
            handle_ref._obj.value = 1234
            return 0

        def sm_comm_destroy(self, handle):
            destroyed.append(rank)
            return 0
    real = hip.lib
    out = {}
    try:
        for name, lib in (("one_rank_fails", Lib({1})), ("id_fails", Lib(set(), fail_id=True)), ("all_fine", Lib(set()))):
            hip.lib = lib
            destroyed.clear()
            res = {}
            try:
                os.environ.pop("STYLEMESH_COMM", None)
                c = D.make_comm(dist, rank, world, None, kind="rccl")
                res["explicit"] = type(c).__name__
            except RuntimeError as e:
                res["explicit"] = "raised: " + str(e)
            res["destroyed_after_explicit"] = list(destroyed)
            # the non-explicit path: pretend the backend is nccl so that 'rccl' is only the default
            class FakeNccl:
                ReduceOp = dist.ReduceOp
                broadcast_object_list = staticmethod(dist.broadcast_object_list)
                all_reduce = staticmethod(dist.all_reduce)

                @staticmethod
                def get_backend():
                    return "gloo"   # tensors of the agreement stay on the CPU
            try:
                # default selection: backend string "nccl" -> rccl; emulate by calling RcclComm through make_comm's
                # default branch with a module whose get_backend() says nccl for the SELECTION only
                class Sel(FakeNccl):
                    calls = [0]

                    @staticmethod
                    def get_backend():
                        Sel.calls[0] += 1
                        return "nccl" if Sel.calls[0] == 1 else "gloo"
                c = D.make_comm(Sel, rank, world, None)
                res["default"] = "RcclComm" if type(c).__name__ == "RcclComm" else "torch"
            except RuntimeError as e:
                res["default"] = "raised: " + str(e)
            # whatever was chosen, the next collective must line up on both ranks
            t = torch.tensor([rank + 1.0])
            dist.all_reduce(t)
            res["sum"] = float(t)
            out[name] = res
    finally:
        hip.lib = real
    torch.save(out, os.path.join(out_dir, f"comm{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_comm_creation_failure_is_decided_collectively(tmp_path):
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_comm_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"comm{r}.pt") for r in (0, 1))
    for name in ("one_rank_fails", "id_fails"):
        assert r0[name]["explicit"].startswith("raised") and r1[name]["explicit"].startswith("raised"), (r0, r1)
        assert r0[name]["default"] == r1[name]["default"] == "torch"
        assert r0[name]["sum"] == r1[name]["sum"] == 3.0
    # the rank whose ncclCommInitRank succeeded gives its communicator back
    assert r0["one_rank_fails"]["destroyed_after_explicit"] == [0] and r1["one_rank_fails"]["destroyed_after_explicit"] == []
    assert r0["all_fine"]["explicit"] == r1["all_fine"]["explicit"] == "RcclComm"
    assert r0["all_fine"]["default"] == r1["all_fine"]["default"] == "RcclComm"


def test_view_prefetcher_order_peek_errors_and_early_close():
    """``ViewPrefetcher`` (background decode of the schedule's views): order kept, at most ``depth`` views held, ``peek``
    never blocks and shows the NEXT view once decoded, a decode error surfaces in the consumer, an abandoned iterator
    stops the thread; ``scheduled_batches(prefetch=...)`` yields the same schedule as without."""
    import threading
    import time as _time
    calls = []

    def get(i):
        calls.append(i)
        _time.sleep(0.01)
        return (torch.full((2,), float(i)),) + (None,) * 7 + (i,)
    pf = D.ViewPrefetcher(get, range(6), depth=2, pin=False)
    seen = []
    for i, items in pf:
        seen.append(i)
        assert float(items[0][0]) == i and len(calls) <= len(seen) + 2 + 1    # never more than depth (+1 in flight) ahead
        _time.sleep(0.03)
        nxt = pf.peek()
        assert nxt is None or float(nxt[0][0]) == i + 1
    assert seen == list(range(6))

    def bad(i):
        if i == 2:
            raise OSError("unreadable view")
        return (i,)
    with pytest.raises(OSError, match="unreadable"):
        list(D.ViewPrefetcher(bad, range(4), depth=1, pin=False))
    pf = D.ViewPrefetcher(get, range(100), depth=2, pin=False)
    it = iter(pf)
    next(it)
    it.close()                      # abandoned after one view (limit_train_batches, an exception in the loop)
    pf._thread.join(timeout=5)
    assert not pf._thread.is_alive()
    plain = [(b[8], b.new_view) for b in D.scheduled_batches(get, range(5), 1, 2, index_repeat=2)]
    ahead = list(D.scheduled_batches(get, range(5), 1, 2, index_repeat=2, prefetch=2))
    assert [(b[8], b.new_view) for b in ahead] == plain and all(callable(b.upcoming) for b in ahead)
    assert ahead[0][0] is ahead[1][0]           # repeats of a view share the decoded tensors


def _eight_rank_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    red = D.make_sparse_grad_reducer(dist, world)            # the product's chunk size (64 floats) and 0.75 threshold
    n_chunks, chunk = 4096, red.chunk
    g = torch.Generator().manual_seed(7 + rank)
    out = {}
    # three "views" per rank: per-rank footprints of 3 %, 12 % and 22 % of the arena - the unions over eight ranks land
    # below, near and ABOVE the dense-fallback threshold
    for view, share in enumerate((0.03, 0.12, 0.22)):
        start = int(torch.randint(0, n_chunks, (1,), generator=g))
        touched = torch.zeros(n_chunks, dtype=torch.bool)
        touched[(start + torch.arange(int(share * n_chunks))) % n_chunks] = True      # a blob, as a view's footprint is
        flags = touched.to(torch.int32)
        red.new_view(flags)
        arena = torch.zeros(n_chunks, chunk)
        arena[touched] = torch.randn(int(touched.sum()), chunk, generator=g)
        arena = arena.reshape(-1)
        dense = arena.clone()
        dist.all_reduce(dense)
        red(arena)
        out[view] = (arena.clone(), dense, red.fraction, red.last_bytes, bool(red._sparse(arena.numel())))
    torch.save(out, os.path.join(out_dir, f"eight{rank}.pt"))
    dist.destroy_process_group()


def test_sparse_grad_reducer_at_eight_ranks_crosses_the_dense_fallback(tmp_path):
    """the reducer at the world size of the node - the union of eight footprints grows past the 0.75
    threshold and the exchange switches to the dense all-reduce on EVERY rank at the same view; sums equal the plain
    all-reduce in both regimes, bytes follow the union."""
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_eight_rank_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    res = [torch.load(tmp_path / f"eight{r}.pt") for r in range(8)]
    fr = [res[0][v][2] for v in range(3)]
    assert fr[0] < fr[1] < fr[2] and fr[0] < 0.3 and fr[2] > 0.75, fr
    for v in range(3):
        mine, dense, frac, nbytes, sparse = res[0][v]
        # (eight summands: the ring adds a chunk's contributions in an order that depends on the buffer's length - equal to
        # rounding against the dense all-reduce, and IDENTICAL on every rank, which is what the update relies on)
        assert torch.allclose(mine, dense, rtol=1e-5, atol=1e-6) and torch.equal(mine != 0, dense != 0)
        for r in res[1:]:
            assert torch.equal(r[v][0], mine) and r[v][2] == frac and r[v][3] == nbytes and r[v][4] == sparse
        assert sparse == (frac <= 0.75)
        assert nbytes == (int(round(frac * 4096)) * 64 * 4 if sparse else 4096 * 64 * 4)


def _adam_chunks(p, m, v, g, flags, lr, step, world, chunk):
    """The fused update restricted to flagged chunks, in torch (elementwise: the same result whatever the launch order);
    the data-term gradient is zeroed where it was applied (sm_adam_fused)."""
    on = flags.bool().repeat_interleave(chunk)
    gr = g[on] / world + 0.01 * p[on]
    m[on] = m[on] + (gr - m[on]) * 0.1
    v[on] = v[on] * 0.999 + gr * gr * 0.001
    p[on] = p[on] - (lr / (1 - 0.9 ** step)) * (m[on] / (v[on].sqrt() / (1 - 0.999 ** step) ** 0.5 + 1e-8))
    g[on] = 0


def _owner_aware_worker(rank, world, port, out_dir):
    """``OwnerAwareGradReducer`` (critical exchange of the shared chunks, deferred exchange of the single-owner ones) against
    the plain union exchange followed by one update: the SAME local gradients, 3 views x 3 steps with a learning-rate change,
    per-rank footprints that overlap partly. Compared after the drain."""
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_chunks, chunk = 192, 16
    n = n_chunks * chunk
    plain = D.make_sparse_grad_reducer(dist, world, chunk_log2=4)
    aware = D.make_sparse_grad_reducer(dist, world, rank=rank, chunk_log2=4)
    gen = torch.Generator().manual_seed(7)             # the SAME stream on every rank: footprints are derived per rank below
    A = [torch.randn(n, generator=gen), torch.zeros(n), torch.zeros(n)]        # p, m, v of the plain path
    B = [t.clone() for t in A]                                                  # ... of the owner-aware path
    stats, step = [], 0
    for view in range(3):
        # rank r touches a window of chunks that overlaps its neighbours' windows by a third (+ one chunk everybody touches)
        lo = (view * 17 + rank * 10) % n_chunks
        mine = torch.zeros(n_chunks, dtype=torch.int32)
        mine[torch.arange(lo, lo + 15) % n_chunks] = 1
        mine[(view * 31) % n_chunks] = 1
        fa, fb = mine.clone(), mine.clone()
        plain.new_view(fa)
        upd_b = lambda flags, lr_, st_: _adam_chunks(B[0], B[1], B[2], gB, flags, lr_, st_, world, chunk)
        gB = torch.zeros(n)
        aware.drain(gB, upd_b)                         # (the engine drains before the per-view collective)
        aware.new_view(fb)
        assert torch.equal(fa, fb)                     # both leave the union in place
        assert aware.n_shared + aware.n_single == aware.n_idx == plain.n_idx
        for k in range(3):
            step += 1
            lr = 1.0 if step < 5 else 0.1
            g_local = torch.zeros(n_chunks, chunk)
            g_local[mine.bool()] = torch.randn(int(mine.sum()), chunk, generator=torch.Generator().manual_seed(1000 * step + rank))
            gA, gB = g_local.reshape(-1).clone(), g_local.reshape(-1).clone()
            plain(gA)
            _adam_chunks(A[0], A[1], A[2], gA, fa, lr, step, world, chunk)
            aware.step(gB, upd_b, lr, step)
            assert float(gA.abs().max()) == 0.0
            stats.append((aware.last_critical_bytes, aware.last_deferred_bytes, plain.last_bytes))
    gB = torch.zeros(n)
    aware.drain(gB, upd_b)
    assert aware._pending is None
    torch.save({"A": A, "B": B, "stats": stats}, os.path.join(out_dir, f"aware{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 8])
def test_owner_aware_exchange_equals_exchange_then_update(tmp_path, world):
    """chunks exactly one rank touches skip the critical exchange - their owner updates them at once,
    the others one step later from a background all-reduce. After the drain p, m, v equal the plain path's BIT FOR BIT on
    every rank (2 ranks; 8 ranks: the single-owner chunks bit for bit, the shared ones to the collective's summation
    order), are identical across the ranks, and the critical bytes are what the shared chunks alone weigh."""
    port = 35500 + (os.getpid() % 2000) + world
    mp.spawn(_owner_aware_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"aware{r}.pt") for r in range(world)]
    for r in res:
        for a, b in zip(r["A"], r["B"]):
            if world == 2:
                assert torch.equal(a, b)
            else:
                assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
        for x, y in zip(r["B"], res[0]["B"]):
            assert torch.equal(x, y)                                  # every rank holds the same texture and moments
        assert all(c + d == p and d > 0 for c, d, p in r["stats"])    # critical + deferred = the union; something was deferred
        assert r["stats"] == res[0]["stats"]
    assert any(c > 0 for c, _, _ in res[0]["stats"])                  # and something was shared
