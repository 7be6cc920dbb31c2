"""Worker of the two-rank GPU tests (tests/test_round2_gpu.py), launched by ``torch.distributed.run``.

    two_rank_worker.py g8 <out_dir>        mean-gradient step of golden G8 with the HIP gradients
    two_rank_worker.py trainer <out_dir>   MiniTrainer epoch over 5 train views (odd) with index_repeat 2
    two_rank_worker.py dense_vs_sparse <out_dir>   zero-initialised texture, 6 steps, dense and sparse reducer
    two_rank_worker.py pipelined <out_dir>  pipelined exchange + update against exchange-then-update, same state
    two_rank_worker.py deferred <out_dir>   owner-aware (critical / deferred) exchange against exchange-then-update

STYLEMESH_TEST_BACKEND=nccl: one GPU per rank, exchange over the product's own RCCL communicator;
gloo: both ranks on cuda:0, collectives staged through the host over torch.distributed's gloo backend (1-GPU boxes) - the
ranks TAKE TURNS on the GPU (see ``main``: two processes with kernels in flight on one GPU are not a configuration of the
product, and not a deterministic one)."""
import os
import sys
import tempfile

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)

from conftest import load_golden  # noqa: E402
from golden_cases import (FLAGSETS, LOSS_WEIGHTS, MULTIVIEW_SEEDS, SMALL_LEVEL_HW, SMALL_ROOM, SMALL_VIEW_HW,  # noqa: E402
                          STYLE_HW, STYLE_SEED, STYLE_WEIGHTS, TEX, VGG_SEED)
from stylemesh_amd.data import synthetic as S  # noqa: E402
from stylemesh_amd.runtime import distributed as D  # noqa: E402


def main():
    mode, out_dir = sys.argv[1], sys.argv[2]
    rank, world, local = D.env_rank_world()
    backend = os.environ.get("STYLEMESH_TEST_BACKEND", "gloo")
    dev_index = local if backend == "nccl" else 0
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    comm = D.make_comm(dist, rank, world, dev)
    if backend == "gloo" and comm is dist:
        # TWO PROCESSES ON ONE GPU TAKE TURNS. With kernels of two processes in flight on one MI355X at the same time a kernel
        # now and then reads a stale 64-byte sector of data its own process wrote a kernel earlier: a plain gather kernel
        # launched four times on unchanged inputs - one stream, device syncs in between - gave different results in 7 % of
        # the steps of a stress run; never with one process on the GPU, never when the processes take turns (0 of 288 steps
        # against 21 of 288: profiles/r06/two_processes_one_gpu.txt). The product runs one process per GPU; this rig does
        # not, so a rank holds an inter-process lock whenever it may have GPU work in flight and hands it over - after a
        # device sync - around every blocking collective, which it stages through the host (torch's gloo backend would
        # otherwise run CUDA copy streams of its own behind the lock's back).
        import fcntl
        turn = open(os.path.join(out_dir, "gpu_turn.lock"), "w")

        class HostStaged:
            ReduceOp = dist.ReduceOp

            def __getattr__(self, name):
                return getattr(dist, name)

            @staticmethod
            def all_reduce(t, op=dist.ReduceOp.SUM, async_op=False):
                h = t.detach().cpu() if t.is_cuda else t
                torch.cuda.synchronize()
                fcntl.flock(turn, fcntl.LOCK_UN)
                dist.all_reduce(h, op=op)
                fcntl.flock(turn, fcntl.LOCK_EX)
                if t.is_cuda:
                    t.copy_(h)
                return _Done() if async_op else None

        class _Done:
            def wait(self):
                return True

            def is_completed(self):
                return True
        comm = HostStaged()
        fcntl.flock(turn, fcntl.LOCK_EX)
    exchange = type(comm).__name__ if comm is not dist else "torch.distributed"
    if backend == "nccl":
        assert exchange == "RcclComm", exchange
    cfgd = FLAGSETS["with_angle_and_depth"]
    g5 = load_golden("g5_with_angle_and_depth")
    init = [torch.from_numpy(g5[f"init{i}"]) for i in range(4)]

    if mode == "g8":
        from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
        cfg = EngineConfig(tex_w=TEX, tex_h=TEX, hierarchical=True, n_layers=4, style_weights=STYLE_WEIGHTS,
                           angle_threshold=cfgd["thr"], style_pyramid_mode=cfgd["mode"], gram_mode=cfgd["gram"],
                           use_angle_weight=True, use_depth_scaling=True, loss_weights=dict(LOSS_WEIGHTS),
                           learning_rate=1, decay_gamma=0.1, decay_step_size=1)
        eng = StepEngine(cfg, S.seeded_vgg_state(VGG_SEED), device=dev)
        eng.load_texture(init)
        eng.set_style_image(S.style_image(STYLE_SEED, *STYLE_HW))
        red = D.make_sparse_grad_reducer(comm, world)
        flags = None
        for seed in D.shard_views(MULTIVIEW_SEEDS, rank, world):
            batch = S.make_view(seed, view_hw=SMALL_VIEW_HW, level_hw=SMALL_LEVEL_HW, level_heights=[40, 64],
                                min_pyramid_depth=0.9, room=S.BoxRoom(SMALL_ROOM))
            eng.set_view(batch)
            eng.forward_backward()                      # accumulates into the gradient arena
            f = eng.touch_flags(red.chunk_log2)
            flags = f if flags is None else torch.maximum(flags, f)
        red.new_view(flags)                             # union of both ranks' (two-view) footprints
        red(eng.arena.g)
        R = len(MULTIVIEW_SEEDS)
        mean_grad = [(g / R + c * p).cpu() for g, c, p in zip(eng.grads, eng.reg_coef, eng.layers)]
        eng.optimizer_step(world_size=R)
        torch.cuda.synchronize()
        torch.save({"mean_grad": mean_grad, "layers": [l.cpu().clone() for l in eng.layers],
                    "exchange": f"{exchange}, {red.last_bytes} bytes"}, os.path.join(out_dir, f"rank{rank}.pt"))
    elif mode == "trainer":
        from stylemesh_amd.data.datamodule import SyntheticSceneDataModule
        from stylemesh_amd.model.model import TextureOptimizationStyleTransferPipeline
        from stylemesh_amd.trainer import JsonlLogger, MiniTrainer
        f = tempfile.NamedTemporaryFile(suffix=".pth", delete=False)
        torch.save(S.seeded_vgg_state(VGG_SEED), f.name)
        model = TextureOptimizationStyleTransferPipeline(
            W=TEX, H=TEX, hierarchical_texture=True, hierarchical_layers=4, style_image=S.style_image(STYLE_SEED, *STYLE_HW),
            style_weights=STYLE_WEIGHTS, vgg_gatys_model_path=f.name, use_angle_weight=True, use_depth_scaling=True,
            style_pyramid_mode=cfgd["mode"], gram_mode=cfgd["gram"], angle_threshold=cfgd["thr"], save_texture=False,
            learning_rate=1, decay_gamma=0.1, decay_step_size=1, loss_weights=dict(LOSS_WEIGHTS))
        deferred = os.environ.get("STYLEMESH_DEFERRED_EXCHANGE", "0") == "1"
        # (dense_above 1: the tiny test texture's union footprint is most of the arena - keep the chunk lists anyway)
        model.grad_reducer = (D.make_sparse_grad_reducer(comm, world, rank=rank, dense_above=1.0) if deferred
                              else D.make_sparse_grad_reducer(comm, world))
        dm = SyntheticSceneDataModule(n_views=7, view_hw=SMALL_VIEW_HW, level_hw=SMALL_LEVEL_HW, min_pyramid_depth=0.9,
                                      split=(0.8, 0.2), index_repeat=2, room_size=SMALL_ROOM, rank=rank, world_size=world)
        dm.setup()
        assert len(dm.train_indices) == 5
        tr = MiniTrainer(max_epochs=1, logger=JsonlLogger(out_dir, rank=rank), device=dev, rank=rank, world_size=world,
                         progress=False)
        tr.fit(model, dm)
        torch.cuda.synchronize()
        used = bool(deferred and model._engine is not None and model._engine.use_deferred_exchange(model.grad_reducer))
        torch.save({"layers": [l.data.detach().cpu().clone() for l in model.texture.layers], "steps": tr.global_step,
                    "deferred": used},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    elif mode == "dense_vs_sparse":
        # a ZERO-initialised texture (sparse ever-touched update) with the plain dense reducer. Each
        # rank optimises its own views; the other rank's gradients arrive in chunks this rank's views never flagged.
        # LOCK-STEP (tests/stepcmp.py): before every step the sparse-reducer engine receives the dense-reducer engine's
        # state, both take the step, and the step is compared through Adam's moments - free-running engines at lr 1 drift
        # apart chaotically, which says nothing about either reducer.
        from stepcmp import lock, step_deviation
        from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
        engs, reds = {}, {}
        # (STYLEMESH_TEST_PAIR=dense,dense / sparse,sparse: a reducer against itself - the harness's own noise)
        pair = os.environ.get("STYLEMESH_TEST_PAIR", "dense,sparse").split(",")
        for slot, kind0 in zip(("dense", "sparse"), pair):
            kind = slot
            cfg = EngineConfig(tex_w=TEX, tex_h=TEX, hierarchical=True, n_layers=4, style_weights=STYLE_WEIGHTS,
                               angle_threshold=cfgd["thr"], style_pyramid_mode=cfgd["mode"], gram_mode=cfgd["gram"],
                               use_angle_weight=True, use_depth_scaling=True, loss_weights=dict(LOSS_WEIGHTS),
                               learning_rate=1, decay_gamma=0.1, decay_step_size=1)
            eng = engs[kind] = StepEngine(cfg, S.seeded_vgg_state(VGG_SEED), device=dev)
            eng.set_style_image(S.style_image(STYLE_SEED, *STYLE_HW))
            assert eng.touched is not None
            reds[kind] = D.make_grad_reducer(comm, world) if kind0 == "dense" else D.make_sparse_grad_reducer(comm, world)
        get = lambda i: S.make_view(MULTIVIEW_SEEDS[i], view_hw=SMALL_VIEW_HW, level_hw=SMALL_LEVEL_HW,
                                    level_heights=[40, 64], min_pyramid_depth=0.9, room=S.BoxRoom(SMALL_ROOM))
        devs = []
        for batch in D.scheduled_batches(get, range(len(MULTIVIEW_SEEDS)), rank, world, index_repeat=3):
            m0, v0 = lock(engs["sparse"], engs["dense"])
            for kind in ("dense", "sparse"):
                engs[kind].training_step(batch, world_size=world, reducer=reds[kind])
            devs.append(step_deviation(engs["sparse"], engs["dense"], m0, v0))
        torch.cuda.synchronize()
        out = {kind: {"p": eng.arena.p.cpu().clone(), "g": eng.arena.g.cpu().clone(), "m": eng.arena.m.cpu().clone(),
                      "dense_update": eng.touched is None} for kind, eng in engs.items()}
        out["lockstep"] = torch.stack(devs).cpu()
        out["lr"] = engs["dense"].lr
        torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    elif mode == "pipelined":
        # the pipelined exchange + update (opt-in: STYLEMESH_PIPELINE_EXCHANGE=1 / auto) against exchange-then-update, from the SAME state and the SAME local gradients: bit-identical p, m, v.
        from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
        cfg = EngineConfig(tex_w=TEX, tex_h=TEX, hierarchical=True, n_layers=4, style_weights=STYLE_WEIGHTS,
                           angle_threshold=cfgd["thr"], style_pyramid_mode=cfgd["mode"], gram_mode=cfgd["gram"],
                           use_angle_weight=True, use_depth_scaling=True, loss_weights=dict(LOSS_WEIGHTS),
                           learning_rate=1, decay_gamma=0.1, decay_step_size=1)
        eng = StepEngine(cfg, S.seeded_vgg_state(VGG_SEED), device=dev)
        eng.load_texture(init)
        eng.set_style_image(S.style_image(STYLE_SEED, *STYLE_HW))
        red = D.make_sparse_grad_reducer(comm, world)
        out = {"auto_small": None, "auto_large": None, "steps": []}
        seeds = D.shard_views(MULTIVIEW_SEEDS, rank, world)
        for k, seed in enumerate(seeds):
            batch = S.make_view(seed, view_hw=SMALL_VIEW_HW, level_hw=SMALL_LEVEL_HW, level_heights=[40, 64],
                                min_pyramid_depth=0.9, room=S.BoxRoom(SMALL_ROOM))
            eng.set_view(batch)
            red.new_view(eng.touch_flags(red.chunk_log2))
            eng.forward_backward()
            torch.cuda.synchronize()
            state = [t.clone() for t in (eng.arena.p, eng.arena.g, eng.arena.m, eng.arena.v, eng.sumsq)]
            count = eng.step_count
            res = []
            for pipe in (False, True):
                for dst, src in zip((eng.arena.p, eng.arena.g, eng.arena.m, eng.arena.v, eng.sumsq), state):
                    dst.copy_(src)
                eng.step_count = count
                eng._grad_dirty = True
                if pipe:
                    eng.exchange_and_update(world, red)
                else:
                    red(eng.arena.g)
                    eng.optimizer_step(world)
                torch.cuda.synchronize()
                res.append([t.cpu().clone() for t in (eng.arena.p, eng.arena.g, eng.arena.m, eng.arena.v, eng.sumsq)])
            out["steps"].append(res)
            # the policy: the same answer on every rank, on for large exchanges only
            eng.pipeline_exchange, eng.pipeline_min_bytes = "auto", 1 << 40
            small = eng.use_pipelined_exchange(red)
            eng.pipeline_min_bytes = 1
            large = eng.use_pipelined_exchange(red)
            out["auto_small"], out["auto_large"] = small, large
        torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    elif mode == "deferred":
        # the owner-aware exchange (critical: shared chunks; deferred: single-owner chunks, one step late on
        # the ranks that do not own them) against exchange-then-update - the real kernels, the SAME local gradients, two
        # views x three steps with a learning-rate change in between; the two paths' states (p, m, v, sum p^2) live side by
        # side and are swapped into the engine's arena for their tails. Compared after the drain.
        from stylemesh_amd.runtime import ops
        from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
        cfg = EngineConfig(tex_w=TEX, tex_h=TEX, hierarchical=True, n_layers=4, style_weights=STYLE_WEIGHTS,
                           angle_threshold=cfgd["thr"], style_pyramid_mode=cfgd["mode"], gram_mode=cfgd["gram"],
                           use_angle_weight=True, use_depth_scaling=True, loss_weights=dict(LOSS_WEIGHTS),
                           learning_rate=1, decay_gamma=0.1, decay_step_size=1)
        eng = StepEngine(cfg, S.seeded_vgg_state(VGG_SEED), device=dev)      # zero texture: the ever-touched sparse update
        eng.set_style_image(S.style_image(STYLE_SEED, *STYLE_HW))
        assert eng.touched is not None
        # (dense_above 1: the tiny test texture's union footprint is most of the arena - keep the chunk lists anyway)
        plain = D.make_sparse_grad_reducer(comm, world, dense_above=1.0)
        aware = D.make_sparse_grad_reducer(comm, world, rank=rank, dense_above=1.0)
        a = eng.arena
        fields = lambda: (a.p, a.m, a.v, eng.sumsq)
        A = [t.clone() for t in fields()]
        B = [t.clone() for t in fields()]

        def load(state):
            for dst, src in zip(fields(), state):
                dst.copy_(src)

        def save(state):
            for dst, src in zip(state, fields()):
                dst.copy_(src)
        stats = []
        seeds = D.shard_views(MULTIVIEW_SEEDS, rank, world)[:2]
        for k, seed in enumerate(seeds):
            batch = S.make_view(seed, view_hw=SMALL_VIEW_HW, level_hw=SMALL_LEVEL_HW, level_heights=[40, 64],
                                min_pyramid_depth=0.9, room=S.BoxRoom(SMALL_ROOM))
            load(B)                                        # (as the engine does before the per-view collective)
            a.g.zero_()
            eng.finish_exchange(world, aware)
            save(B)
            load(A)
            eng.set_view(batch)
            fa = eng.touch_flags(plain.chunk_log2)
            fb = fa.clone()
            plain.new_view(fa)
            aware.new_view(fb)
            assert torch.equal(fa, fb) and aware.n_shared + aware.n_single == plain.n_idx
            ops.flags_or(eng.touched, fa)                  # what begin_step does with the union of the ranks' views
            ops.flags_or(eng._view_flags, fa)
            eng._other_flags = None
            if k == 1:
                eng.end_epoch()                            # StepLR: lr 1 -> 0.1
            for rep in range(3):
                load(A)
                a.g.zero_()
                eng.forward_backward()                     # this rank's local gradient of the step (from the texture both paths share on the view's own texels)
                g_local = a.g.clone()
                count = eng.step_count
                red_bytes = None
                # path A: union exchange, then one update over the ever-touched chunks
                plain(a.g)
                eng.optimizer_step(world)
                save(A)
                assert float(a.g.abs().max()) == 0.0
                # path B: the deferred exchange
                load(B)
                a.g.copy_(g_local)
                eng.step_count = count
                eng.exchange_and_update_deferred(world, aware)
                save(B)
                stats.append((aware.last_critical_bytes, aware.last_deferred_bytes, plain.last_bytes))
        load(B)
        a.g.zero_()
        eng.finish_exchange(world, aware)
        save(B)
        torch.cuda.synchronize()
        torch.save({"A": [t.cpu() for t in A], "B": [t.cpu() for t in B], "g": a.g.cpu().clone(), "stats": stats,
                    "steps": eng.step_count}, os.path.join(out_dir, f"rank{rank}.pt"))
    else:
        raise SystemExit(f"unknown mode {mode}")
    if hasattr(comm, "destroy"):
        comm.destroy()
    if backend == "gloo":
        torch.cuda.synchronize()
        try:
            import fcntl
            fcntl.flock(turn, fcntl.LOCK_UN)
        except NameError:
            pass
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
