"""``MiniTrainer``: the slice of ``pytorch_lightning.Trainer`` (1.4.9, reference requirements.txt:76) that
``model/optimize.py:30,165`` relies on - automatic optimisation with Lightning's hook order
(SURVEY.md section 3.2-3.4), one process per GPU. pytorch_lightning is not installed in the target image; a
module written for Lightning (``training_step`` / ``validation_step`` / ``configure_optimizers`` / epoch hooks)
runs unchanged.

Per epoch: on_train_epoch_start -> for each batch [optimizer.zero_grad -> training_step -> loss.backward ->
optimizer.step] -> validation [on_validation_epoch_start -> validation_step* -> on_validation_epoch_end ->
on_epoch_end] -> on_train_epoch_end -> on_epoch_end -> scheduler.step.

Multi-GPU (``world_size > 1``): views shard over ranks by the sampler; the module's ``grad_reducer`` all-reduces
the flat texture-gradient arena over RCCL between backward and the optimizer step (SURVEY.md section 8 e).
"""
from __future__ import annotations

import json
import os
import time

import torch



def _alloc_note(device) -> str:
    """Device allocations so far (caching-allocator misses: each is a hipMalloc and a stall of the launch loop) - a
    first epoch that is slower than the later ones shows here."""
    if device == "cpu" or not torch.cuda.is_available():
        return ""
    st = torch.cuda.memory_stats()
    return (f" [device allocs {st.get('num_device_alloc', 0)}, frees {st.get('num_device_free', 0)}, "
            f"reserved {st.get('reserved_bytes.all.current', 0) / 2 ** 30:.1f} GiB]")

class JsonlLogger:
    """TensorBoard-shaped scalar logger (``logger.experiment.add_scalar(s)``) writing JSON lines; device tensors
    are only converted when the line is written, every ``flush_every`` records (no per-step host sync)."""

    def __init__(self, save_dir="lightning_logs", version=0, flush_every=200, rank=0):
        """``rank``: with several ranks every rank logs ITS OWN views' losses; rank 0 writes ``scalars.jsonl``, rank r
        ``scalars.rank{r}.jsonl`` (one shared file would interleave records with identical tags and steps)."""
        self.save_dir, self.version = save_dir, version
        self.log_dir = os.path.join(save_dir, f"lightning_logs/version_{version}")
        os.makedirs(self.log_dir, exist_ok=True)
        self.file_name = "scalars.jsonl" if rank == 0 else f"scalars.rank{rank}.jsonl"
        self.experiment = self
        self._pending, self._flush_every = [], flush_every

    def add_scalar(self, tag, value, step):
        self._pending.append((tag, value, step))
        if len(self._pending) >= self._flush_every:
            self.flush()

    def add_scalars(self, tag, values, step):
        for k, v in values.items():
            self.add_scalar(f"{tag}/{k}", v, step)

    def add_image(self, *a, **k):
        pass

    def flush(self):
        # device scalars cross to the host in ONE transfer per device (a float() per record is a synchronisation each:
        # 200 of them per flush were 0.2 ms of host time per step on the single-level workload)
        vals = [v for _, v, _ in self._pending]
        dev = [i for i, v in enumerate(vals) if torch.is_tensor(v) and v.is_cuda]
        if dev:
            host = torch.stack([vals[i].detach().reshape(-1)[0].float() for i in dev]).tolist()
            for i, h in zip(dev, host):
                vals[i] = h
        with open(os.path.join(self.log_dir, self.file_name), "a") as f:
            for (tag, _, step), value in zip(self._pending, vals):
                f.write(json.dumps({"tag": tag, "step": int(step), "value": float(value)}) + "\n")
        self._pending.clear()


class MiniTrainer:
    def __init__(self, max_epochs=1, logger=None, device=None, rank=0, world_size=1, limit_train_batches=None,
                 limit_val_batches=None, progress=True):
        self.max_epochs, self.rank, self.world_size = max_epochs, rank, world_size
        self.logger = logger if logger is not None else JsonlLogger()
        self.device = device if device is not None else (torch.device("cuda", 0) if torch.cuda.is_available() else "cpu")
        self.limit_train_batches, self.limit_val_batches, self.progress = limit_train_batches, limit_val_batches, progress
        self.global_step = 0
        self.call_log = []   # hook names in call order (tests)

    def _upload(self, batch):
        mv = lambda t: t.to(self.device, non_blocking=True) if torch.is_tensor(t) else t
        return tuple(x if k == 8 else ([mv(u) for u in x] if isinstance(x, (list, tuple)) else mv(x))
                     for k, x in enumerate(batch))

    def _to_device(self, batch):
        """Host batch -> device batch. The view index (element 8) stays on the host (reading it back would be a
        device-to-host sync per step); consecutive repeats of one view (same host tensors) reuse the device copy; the
        schedule's ``new_view`` flag (``runtime.distributed.ViewBatch``) is carried over. With a prefetching loader
        (``batch.upcoming``) the NEXT view is uploaded one view ahead on a copy stream, during the current view's steps:
        a view change then finds its inputs resident in HBM (pinned host memory, asynchronous copies)."""
        from .runtime.distributed import ViewBatch
        last = getattr(self, "_last_dev", None)
        ahead = getattr(self, "_ahead", None)
        if last is not None and last[0] is batch[0]:
            items = last[1]
        elif ahead is not None and ahead[0] is batch[0]:
            torch.cuda.current_stream().wait_event(ahead[2])
            items = ahead[1]
            self._last_dev, self._ahead = (batch[0], items), None
        else:
            items = self._upload(batch)
            self._last_dev = (batch[0], items)
        upcoming = getattr(batch, "upcoming", None)
        # (also at a view's FIRST step - with index_repeat 1 there is no other: the module's hook leaves a request that the
        # engine serves once this batch's own view has become current, ``StepEngine.request_prepare``)
        if upcoming is not None and torch.cuda.is_available() and getattr(self, "_ahead", None) is None:
            nxt = upcoming()
            if nxt is not None and nxt[0] is not batch[0]:
                if not hasattr(self, "_copy_stream"):
                    self._copy_stream = torch.cuda.Stream(device=self.device)
                with torch.cuda.stream(self._copy_stream):
                    dev_items = self._upload(nxt)
                    done = torch.cuda.Event()
                    done.record()
                main = torch.cuda.current_stream()
                for x in dev_items:      # allocated on the copy stream, consumed on the main one
                    for t in (x if isinstance(x, (list, tuple)) else [x]):
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(main)
                self._ahead = (nxt[0], dev_items, done)
                hook = getattr(getattr(self, "_model", None), "prepare_view", None)
                if hook is not None and self.world_size == 1:   # the next view's per-view constants, one view ahead
                    hook(ViewBatch(dev_items, new_view=True), done)
        return ViewBatch(items, new_view=getattr(batch, "new_view", None))

    def _call(self, model, name, *a):
        self.call_log.append(name)
        fn = getattr(model, name, None)
        return fn(*a) if fn is not None else None

    def fit(self, model, datamodule):
        try:
            model.logger = self.logger
        except AttributeError:   # a real LightningModule exposes ``logger`` as a read-only property
            pass
        model.world_size = self.world_size
        self._model = model
        # which scene / datamodule the batches come from: the engine's resident views are keyed by it (a module fitted on
        # another datamodule must not find the first one's views under the same indices)
        model.scene_identity = (type(datamodule).__name__, id(datamodule), getattr(datamodule, "scene", None))
        if getattr(model, "_engine", None) is not None:
            model._engine.set_scene(model.scene_identity)
        if self.device != "cpu":
            from .runtime.hostcpu import limit_host_threads
            limit_host_threads()
        if hasattr(model, "_ensure_engine"):
            model.fused_backward_done = True   # training_step's gradients are final: no autograd pass over its scalar
        if hasattr(model, "to") and self.device != "cpu":
            model.to(self.device)
        caller_stream = None
        if self.device != "cpu" and torch.cuda.is_available():
            from .runtime.engine import trunk_stream
            st = trunk_stream(self.device)       # high-priority stream for the step's trunk (side work fills the rest)
            if st is not None:
                caller_stream = torch.cuda.current_stream()
                torch.cuda.set_stream(st)
        try:
            return self._fit(model, datamodule)
        finally:
            if caller_stream is not None:        # (hand the caller's stream back, ordered behind the training)
                caller_stream.wait_stream(torch.cuda.current_stream())
                torch.cuda.set_stream(caller_stream)

    def _fit(self, model, datamodule):
        t0 = time.time()                         # (the schedule's clock: engine set-up and loader start-up included)
        if hasattr(datamodule, "warm_start"):
            datamodule.warm_start()              # decode processes up before anything waits for them
        if hasattr(model, "_ensure_engine") and self.device != "cpu" and torch.cuda.is_available():
            # the fused engine (weight packing, style targets: ~0.3 s) is built here, beside the loader's start-up,
            # instead of inside the first training step behind the first view
            model._ensure_engine(torch.device(self.device))
        optimizers, schedulers = model.configure_optimizers()
        opt, sched = optimizers[0], (schedulers[0] if schedulers else None)
        if hasattr(opt, "world_size"):
            opt.world_size = self.world_size
        def now():
            if self.device != "cpu" and torch.cuda.is_available():
                torch.cuda.synchronize()
            return time.time()
        for epoch in range(self.max_epochs):
            model.current_epoch = epoch
            t_epoch = now()
            steps0 = self.global_step
            self._call(model, "on_train_epoch_start")
            # a view uploaded / prepared ahead that the previous epoch never reached (limit_train_batches, an exception):
            # forget it, or upload-ahead and prepare-ahead would stay off for the rest of the run
            self._ahead = None
            eng = getattr(model, "_engine", None)
            if eng is not None and hasattr(eng, "drop_prepared"):
                eng.drop_prepared()
            timing = os.environ.get("STYLEMESH_TRAINER_TIMING") == "1"   # host-side seconds per phase of the loop
            tm = getattr(self, "host_seconds", None) or {"next_batch": 0.0, "to_device": 0.0, "training_step": 0.0,
                                                         "backward": 0.0, "optimizer_step": 0.0}
            self.host_seconds = tm
            loader = iter(datamodule.train_dataloader())
            batch_idx = -1
            while True:
                t_a = time.perf_counter() if timing else 0.0
                try:
                    batch = next(loader)
                except StopIteration:
                    break
                batch_idx += 1
                if self.limit_train_batches is not None and batch_idx >= self.limit_train_batches:
                    loader.close() if hasattr(loader, "close") else None
                    break
                t_b = time.perf_counter() if timing else 0.0
                batch = self._to_device(batch)
                opt.zero_grad()
                t_c = time.perf_counter() if timing else 0.0
                out = self._call(model, "training_step", batch, batch_idx)
                t_d = time.perf_counter() if timing else 0.0
                if not out.get("backward_done", False):    # (the fused module has deposited its gradients already)
                    out["loss"].backward()
                t_e = time.perf_counter() if timing else 0.0
                opt.step()
                self.global_step += 1
                if timing:
                    t_f = time.perf_counter()
                    tm["next_batch"] += t_b - t_a; tm["to_device"] += t_c - t_b; tm["training_step"] += t_d - t_c
                    tm["backward"] += t_e - t_d; tm["optimizer_step"] += t_f - t_e
                    if getattr(batch, "new_view", False):     # the first step of a view (set_view inside), per VIEW
                        tm["first_step_of_view_total"] = tm.get("first_step_of_view_total", 0.0) + (t_f - t_a)
                        tm["first_steps"] = tm.get("first_steps", 0) + 1
                        fs = tm.setdefault("_first", [0.0, 0.0, 0.0])
                        fs[0] += t_b - t_a; fs[1] += t_c - t_b; fs[2] += t_d - t_c
            # N > 1 with the deferred exchange: the last step's background sums are applied before anything (validation,
            # the epoch-end texture export, the next epoch's first collective) reads the texture
            eng = getattr(model, "_engine", None)
            if eng is not None and hasattr(eng, "finish_exchange"):
                eng.finish_exchange(self.world_size, getattr(model, "grad_reducer", None))
            t_train = now()
            val_loader = datamodule.val_dataloader() if hasattr(datamodule, "val_dataloader") else None
            if val_loader is not None:
                self._call(model, "on_validation_epoch_start")
                with torch.no_grad():
                    for batch_idx, batch in enumerate(val_loader):
                        if self.limit_val_batches is not None and batch_idx >= self.limit_val_batches:
                            break
                        self._call(model, "validation_step", self._to_device(batch), batch_idx)
                self._call(model, "on_validation_epoch_end")
                self._call(model, "on_epoch_end")
            self._call(model, "on_train_epoch_end")
            self._call(model, "on_epoch_end")
            if sched is not None:
                sched.step()
            if self.progress and self.rank == 0:
                t_end = now()
                n = self.global_step - steps0
                print(f"epoch {epoch}: {self.global_step} steps, {t_end - t0:.1f} s "
                      f"[train loop {t_train - t_epoch:.2f} s = {n / max(t_train - t_epoch, 1e-9):.1f} steps/s, "
                      f"validation + epoch-end hooks {t_end - t_train:.2f} s]" + _alloc_note(self.device), flush=True)
        if self.device != "cpu" and torch.cuda.is_available():
            torch.cuda.synchronize()
        try:    # texture exports still being encoded by the writer thread
            from .model.texture.texture import IMAGE_WRITER
            IMAGE_WRITER.wait()
        except ImportError:
            pass
        if self.progress and self.rank == 0:
            print(f"fit: {time.time() - t0:.1f} s")
            if getattr(self, "host_seconds", None) and os.environ.get("STYLEMESH_TRAINER_TIMING") == "1":
                n = max(self.global_step, 1)
                hs = dict(self.host_seconds)
                nf = hs.pop("first_steps", 0)
                first = hs.pop("first_step_of_view_total", 0.0)
                fs = hs.pop("_first", None)
                eng = getattr(model, "_engine", None)
                if eng is not None and getattr(eng, "set_view_calls", 0):
                    marks = getattr(eng, "set_view_marks", {})
                    print(f"set_view: {1e3 * eng.set_view_host_s / eng.set_view_calls:.2f} ms of host time per call "
                          f"(x{eng.set_view_calls}, {getattr(eng, 'prepared_swaps', 0)} prepared ahead)" + "".join(f" {k} {1e3 * v / eng.set_view_calls:.2f}" for k, v in marks.items())
                          + "; allocator misses: " + str(getattr(eng, "set_view_alloc_misses", {})))
                print("host ms per step: " + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in hs.items())
                      + (f"; first step of a view {1e3 * first / nf:.2f} ms (x{nf}: next_batch {1e3 * fs[0] / nf:.2f}, "
                         f"to_device {1e3 * fs[1] / nf:.2f}, training_step {1e3 * fs[2] / nf:.2f})" if nf else ""))
            from .runtime.distributed import LOADER_STATS
            if LOADER_STATS:
                LOADER_STATS.settle()
                print(f"loader: decode {sum(p.decode_s for p in LOADER_STATS):.1f} s in the prefetch thread, "
                      f"training loop blocked {sum(p.wait_s for p in LOADER_STATS):.1f} s waiting for views")
        if hasattr(self.logger, "flush"):
            self.logger.flush()
        return model
