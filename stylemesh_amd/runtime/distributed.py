"""Multi-GPU protocol of the view-sharded path (SURVEY.md section 8 e): one process per GPU, views of the scene
shard over ranks, ONE exchange per step - a sum all-reduce of the flat texture-gradient arena, restricted to the
chunks the ranks' current views can touch (``SparseGradReducer``; RCCL over xGMI on the GPU box: ``torch.distributed``
backend "nccl"; "gloo" in the CPU tests) - then the identical fused update on every rank with
``grad_scale = 1 / world_size`` (DDP-mean semantics).

Nothing here touches the kernels; it is the host logic around them, testable without a GPU.
"""
from __future__ import annotations

import os


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def shard_views(indices, rank: int, world_size: int):
    """Rank r takes views r, r + R, r + 2R, ... of the scene's train list (keeps the per-rank consecutive-repeat
    structure of RepeatingSampler, so the per-view constant cache still hits)."""
    return list(indices)[rank::world_size]


def padded_shard(indices, rank: int, world_size: int):
    """``shard_views`` padded (by repeating the last view) to ceil(n / R) entries: every rank then takes the same
    number of steps and changes views at the same steps - every step contains a collective."""
    mine = shard_views(indices, rank, world_size)
    per_rank = -(-len(list(indices)) // world_size)
    if not mine:
        raise ValueError("more ranks than views")
    return mine + [mine[-1]] * (per_rank - len(mine))


def steps_per_epoch(n_views: int, index_repeat: int, world_size: int) -> int:
    """All ranks must take the same number of steps (every step contains a collective): ranks with one view
    fewer repeat their last view."""
    per_rank = -(-n_views // world_size)
    return per_rank * index_repeat


def rank_schedule(indices, rank, world_size, index_repeat):
    """View index per step for this rank, padded so that every rank has ``steps_per_epoch`` entries."""
    return [v for v in padded_shard(indices, rank, world_size) for _ in range(index_repeat)]


class ViewBatch(tuple):
    """The 13-tuple of the input contract (``view_contract.assemble_batch``) plus ``new_view``: True on the first step
    of every ``index_repeat`` block of the rank's schedule. The flag is a function of the schedule POSITION only, so
    it is the same on every rank at every step - a rank whose shard was padded by repeating its last view sees
    ``new_view`` although its view key did not change, and still enters the per-view collective
    (``SparseGradReducer.new_view``) that the other ranks are entering."""
    new_view = None
    upcoming = None   # optional callable: the NEXT view's decoded host items once ready (``ViewPrefetcher.peek``)

    def __new__(cls, items, new_view=None):
        self = super().__new__(cls, items)
        self.new_view = new_view
        return self


class _PinnedRing:
    """Pinned staging buffers for the decoded views, allocated ONCE: a ring of ``slots`` buffers per tensor position of
    the batch structure, filled by ``copy_`` (a host memcpy). ``tensor.pin_memory()`` per view allocates fresh pinned
    memory for ~20 tensors every time (hipHostMalloc: milliseconds each until the allocator's cache warms, and a
    single-level view leaves only 24 ms per view): 23 ms per view measured in the receiver thread."""

    def __init__(self, slots: int):
        self.slots, self.turn, self.bufs = slots, 0, {}

    def _put(self, x, path):
        import torch
        if torch.is_tensor(x):
            if x.is_cuda or not torch.cuda.is_available():
                return x
            key = (path, tuple(x.shape), x.dtype)
            ring = self.bufs.get(key)
            if ring is None:
                ring = self.bufs[key] = [torch.empty(x.shape, dtype=x.dtype).pin_memory() for _ in range(self.slots)]
            dst = ring[self.turn % self.slots]
            dst.copy_(x)
            return dst
        if isinstance(x, (list, tuple)):
            return type(x)(self._put(u, path + (k,)) for k, u in enumerate(x))
        return x

    def stage(self, items):
        out = self._put(items, ())
        self.turn += 1
        return out


def _pin_tree(x):
    """Host tensors of a batch tuple into pinned memory (asynchronous H2D copies need it); lists keep their shape."""
    import torch
    if torch.is_tensor(x):
        return x.pin_memory() if (not x.is_cuda and torch.cuda.is_available()) else x
    if isinstance(x, (list, tuple)):
        return type(x)(_pin_tree(u) for u in x)
    return x


class _LoaderStat:
    """decode / wait seconds of one epoch's ``ViewPrefetcher`` - scalars only: holding the prefetcher itself would keep
    its ring of pinned staging buffers (6 x 14 MB for a 4-level view) alive for every epoch of the run."""

    def __init__(self, source):
        self._source = source
        self.decode_s = self.wait_s = 0.0

    def settle(self):
        if self._source is not None:
            self.decode_s, self.wait_s, self._source = self._source.decode_s, self._source.wait_s, None


class _LoaderStats(list):
    def settle(self):
        for st in self:
            st.settle()


LOADER_STATS = _LoaderStats()   # per epoch: decode / wait seconds of its ViewPrefetcher (diagnostics)


def _pack_view(items):
    """A decoded view as ONE flat uint8 tensor + a small description (worker side). Twenty tensors through a
    torch.multiprocessing queue are twenty shared-memory hand-offs un-pickled under the training process's interpreter
    lock; one buffer is one."""
    import torch
    chunks, off = [], [0]

    def walk(x):
        if torch.is_tensor(x):
            t = x.detach().contiguous()
            start = (off[0] + 15) // 16 * 16                      # 16-byte aligned pieces
            nbytes = t.numel() * t.element_size()
            chunks.append((start, t.reshape(-1).view(torch.uint8) if nbytes else None))
            off[0] = start + nbytes
            return ("t", start, tuple(t.shape), str(t.dtype).replace("torch.", ""))
        if isinstance(x, (list, tuple)):
            return ("l" if isinstance(x, list) else "u", [walk(u) for u in x])
        return ("o", x)
    meta = walk(items)
    flat = torch.empty(max(off[0], 1), dtype=torch.uint8)
    for start, b in chunks:
        if b is not None:
            flat[start:start + b.numel()] = b
    return flat, meta


def _unpack_view(flat, meta):
    """Views into ``flat`` with the structure ``_pack_view`` saw (no copies)."""
    import torch
    kind = meta[0]
    if kind == "t":
        _, start, shape, dtype = meta
        dt = getattr(torch, dtype)
        n = 1
        for d in shape:
            n *= d
        return flat[start:start + n * torch.empty(0, dtype=dt).element_size()].view(dt).view(shape)
    if kind in ("l", "u"):
        seq = [_unpack_view(flat, m) for m in meta[1]]
        return seq if kind == "l" else tuple(seq)
    return meta[1]


def _decode_main(get_view, tasks, results):
    """Body of the decode PROCESS: per request (a list of view indices) the decoded views, in order, then None."""
    import torch
    torch.set_num_threads(1)
    try:
        os.nice(10)      # the training process's launch loop and its wake-ups from device syncs come first
    except OSError:
        pass
    if get_view is None:
        # the decoder arrives as the first message: a process started under 'spawn' receives its arguments through a pipe
        # the parent WRITES SYNCHRONOUSLY - beyond the pipe's 64 KB (a 276-view dataset's file lists pickle to 100 KB) the
        # parent's start() blocks until the child has imported its modules (0.8 s per worker, measured); a queue's
        # feeder thread writes in the background instead
        get_view = tasks.get()
        if get_view is None:
            return
        if isinstance(get_view, (bytes, bytearray)):
            import pickle
            try:
                get_view = pickle.loads(get_view)
            except BaseException as e:   # noqa: BLE001 - a decoder that does not arrive whole: tell the consumer, do not hang it
                err = f"the view decoder could not be un-pickled in the worker: {type(e).__name__}: {e}"
                while True:              # every request gets the error (the consumer raises on the first)
                    if tasks.get() is None:
                        return
                    results.put(("__error__", err))
                    results.put(None)
    while True:
        order = tasks.get()
        if order is None:
            return
        for i in order:
            try:
                results.put((i,) + _pack_view(get_view(i)))
            except BaseException as e:   # noqa: BLE001 - reported to the consumer, which raises it
                results.put(("__error__", f"{type(e).__name__}: {e}"))
                break
        results.put(None)


class DecodeProcess:
    """Persistent worker process(es) that decode views (``get_view(i)``, picklable: a dataset's bound ``__getitem__``).
    Why processes: the decode is a chain of numpy / PIL / torch calls with Python in between - run as a THREAD of the
    training process it competes for the interpreter lock with the loop that enqueues ~200 kernel launches per step,
    and both slow down (measured on the 276-view c3 scene: 235 ms per view decoded beside the loop against 56 ms
    alone, the epoch 76 instead of ~150 views/s). The reference uses DataLoader worker processes for the same reason
    (data/abstract_dataset.py:480). ``n_workers`` > 1: worker k decodes every n-th view of a request and the results are
    taken in order - a 4-level view (56 ms of decode per 130 ms of steps) needs one worker, a single-level view (its 20
    steps last 24 ms) three. Decoded tensors come back through shared memory (torch.multiprocessing); the processes
    never touch the GPU. Started with 'spawn' (the parent has initialised HIP: no fork)."""

    def __init__(self, get_view, depth: int = 2, n_workers: int = 1):
        import torch.multiprocessing as mp
        ctx = mp.get_context("spawn")
        self.n = max(1, n_workers)
        self.tasks = [ctx.Queue() for _ in range(self.n)]
        self.results = [ctx.Queue(maxsize=max(1, depth)) for _ in range(self.n)]
        self.procs = [ctx.Process(target=_decode_main, args=(None, self.tasks[k], self.results[k]), daemon=True,
                                  name=f"stylemesh-view-decode-{k}") for k in range(self.n)]
        # the decoders are single-threaded by construction (numpy indexing, PIL, small torch ops): keep their math
        # libraries from spinning up one thread per VISIBLE core (256 on the GPU boxes, under a 16-CPU quota)
        saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
        os.environ.update({k: "1" for k in saved})
        try:
            # pickled HERE, in the caller: a decoder that cannot be pickled raises now instead of leaving a traceback in
            # the queue's feeder thread and a child that waits for its first message forever; the bytes still
            # travel through the queue (see _decode_main: not through the start-up pipe)
            # ... and ONCE PER WORKER: a ForkingPickler payload may hold single-use handles (a torch CPU tensor
            # under the file_descriptor sharing strategy, a Connection) - the second worker to load the same bytes would die
            from multiprocessing.reduction import ForkingPickler
            payloads = [bytes(ForkingPickler.dumps(get_view)) for _ in self.procs]
            for p in self.procs:
                p.start()
            for q, payload in zip(self.tasks, payloads):
                q.put(payload)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        self.proc = self.procs[0]
        self.ring = None      # pinned staging ring of the consumer side (``ViewPrefetcher``), kept across epochs
        self.busy = False
        self._next, self._open = 0, 0

    def request(self, order):
        assert not self.busy, "one request at a time"
        order = list(order)
        self.busy = True
        self._next, self._open = 0, self.n
        for k in range(self.n):
            self.tasks[k].put(order[k::self.n])

    class NotReady(Exception):
        pass

    def get(self, block: bool = True):
        """The next view of the request, in request order; None once every worker has delivered its share.
        ``block=False``: raises ``DecodeProcess.NotReady`` instead of waiting for a view that is still being decoded."""
        import queue
        while self._open > 0:
            try:
                item = self.results[self._next % self.n].get(block)
            except queue.Empty:
                raise DecodeProcess.NotReady()
            if item is None:              # this worker's share is done: the shares end within one view of each other
                self._open -= 1
                self._next += 1
                continue
            self._next += 1
            return item
        self.busy = False
        return None

    def alive(self):
        return all(p.is_alive() for p in self.procs)

    def close(self):
        for p in self.procs:
            if p.is_alive():
                p.terminate()
        for p in self.procs:
            p.join(timeout=5)


class ViewPrefetcher:
    """Decodes the views of a schedule AHEAD of the training loop.

    The reference hides view loading behind the step with 4 DataLoader worker processes (data/abstract_dataset.py:480;
    ``__getitem__`` :270-345 reads a jpg, a png and 6 ``.npy`` files and runs the numpy / PIL resizes). Here ONE decoder
    is enough: a view is optimised for ``index_repeat`` (20-100) consecutive steps - 0.1-0.6 s of GPU time - and its
    decode (tens of ms) runs during the PREVIOUS view's steps: in a ``DecodeProcess`` (``worker``; default of the
    directory loaders) or, for cheap / unpicklable ``get_view``s, in this object's own thread. A receiver thread pins the
    decoded batches, so that the host-to-device copy the trainer issues is asynchronous (``MiniTrainer`` issues it one
    view ahead, on a copy stream). ``depth`` decoded views are held at most (a 4-level ScanNet view is 14 MB). With
    pinning, a yielded view's HOST tensors live in a ring of staging buffers and stay valid until ``depth + 4`` further
    views have been produced - a consumer that keeps host views longer must copy them."""

    def __init__(self, get_view, order, depth: int = 2, pin: bool = True, worker: "DecodeProcess | None" = None):
        import collections
        import threading
        self._get, self._order, self._pin, self._worker = get_view, list(order), pin, worker
        self._ready = collections.deque()
        self._cv = threading.Condition()
        self._depth, self._stop, self._finished = max(1, depth), False, False
        # a view's pinned copy must outlive: the ready queue (depth), the view being trained on, the one uploaded ahead
        self._ring = None
        if pin and worker is not None:    # ONE ring for the life of the decode process(es): epochs re-use it
            if getattr(worker, "ring", None) is None or worker.ring.slots != self._depth + 4:
                worker.ring = _PinnedRing(self._depth + 4)
            self._ring = worker.ring
        elif pin:
            self._ring = _PinnedRing(self._depth + 4)
        self.decode_s = 0.0     # time spent decoding / receiving (and pinning) views on the consumer's side
        self.wait_s = 0.0       # time the consumer spent blocked on a view that was not ready
        self._thread = None
        if worker is not None:
            # Decode PROCESSES: no thread in the training process at all. The consumer itself takes a finished view off
            # the workers' queue (one un-pickle of a flat buffer + one memcpy into the pinned ring: 0.3-1.5 ms per view)
            # - a receiver THREAD doing that beside the launch loop cost the loop 14 ms per view change through the
            # interpreter lock (set_view: 19.9 ms with the thread, 5.8 ms without; round 3).
            worker.request(self._order)
            self._head, self._taken = None, 0
        else:
            self._thread = threading.Thread(target=self._run, name="stylemesh-view-prefetch", daemon=True)
            self._thread.start()

    def _fetch(self, block: bool):
        """(worker mode) The next decoded view as (index, items), ``None`` at the end of the request; with
        ``block=False`` raises ``DecodeProcess.NotReady`` if it is still being decoded."""
        import time
        t0 = time.perf_counter()
        item = self._worker.get(block)
        if item is None:
            self._finished = True
            return None
        if item[0] == "__error__":
            raise RuntimeError(f"view decode failed in the worker process: {item[1]}")
        i, flat, meta = item     # one flat buffer per view (``_pack_view``)
        if self._pin and self._ring is not None:
            flat = self._ring.stage(flat)          # ONE memcpy into the pinned ring
        out = (i, _unpack_view(flat, meta))
        dt = time.perf_counter() - t0
        if block:
            self.wait_s += dt
        self.decode_s += dt
        return out

    def _run(self):
        import time
        try:
            for i in self._order:
                with self._cv:
                    while len(self._ready) >= self._depth and not self._stop:
                        self._cv.wait()
                    if self._stop:
                        return
                t0 = time.perf_counter()
                items = self._get(i)
                if self._pin:
                    items = self._ring.stage(items)
                self.decode_s += time.perf_counter() - t0
                with self._cv:
                    self._ready.append((i, items))
                    self._cv.notify_all()
            self._finished = True
            with self._cv:
                self._ready.append(None)
                self._cv.notify_all()
        except BaseException as e:   # noqa: BLE001 - handed to the consumer, which re-raises it
            with self._cv:
                self._ready.append(e)
                self._cv.notify_all()

    def peek(self):
        """The next decoded view's items if its decode has finished, else None (never blocks)."""
        if self._worker is not None:
            if self._head is None and not self._finished:
                try:
                    self._head = self._fetch(block=False)
                except DecodeProcess.NotReady:
                    return None
            return self._head[1] if self._head is not None else None
        with self._cv:
            head = self._ready[0] if self._ready else None
        return head[1] if isinstance(head, tuple) else None

    def __iter__(self):
        import time
        try:
            if self._worker is not None:
                while True:
                    head, self._head = self._head, None
                    if head is None:
                        if self._finished:
                            return
                        head = self._fetch(block=True)
                        if head is None:
                            return
                    yield head
            while True:
                t0 = time.perf_counter()
                with self._cv:
                    while not self._ready:
                        self._cv.wait()
                    head = self._ready.popleft()
                    self._cv.notify_all()
                self.wait_s += time.perf_counter() - t0
                if head is None:
                    return
                if isinstance(head, BaseException):
                    raise head
                yield head
        finally:
            self.close()

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        if self._worker is not None and not self._finished:
            # abandoned mid-request (limit_train_batches, an exception in the loop): the worker still holds undelivered
            # views of this request - end it; the owner starts a fresh one for the next epoch
            self._worker.close()


def scheduled_batches(get_view, indices, rank: int, world_size: int, index_repeat=1, repeat=True, prefetch: int = 0,
                      worker=None):
    """The rank's train schedule as ``ViewBatch``es: views ``padded_shard(indices)``, each ``index_repeat``
    consecutive times (``repeat=False``: once, the 'sequential' sampler mode). Consecutive repeats yield the SAME
    object (one decode / upload per view; callers may cache by identity), ``new_view`` marks the block boundaries.
    ``prefetch`` > 0: views are decoded (and pinned) up to that many ahead by a ``ViewPrefetcher`` (in the ``worker``
    ``DecodeProcess`` if one is given, else in a thread); every
    yielded batch then carries ``upcoming`` - a callable returning the NEXT view's decoded items once they are ready -
    so that the consumer can start its upload during the current view's steps."""
    rep = index_repeat if repeat else 1
    count = lambda i: rep[i] if isinstance(rep, list) else rep
    order = [i for i in padded_shard(indices, rank, world_size) if count(i) >= 1]
    if prefetch > 0:
        source = ViewPrefetcher(get_view, order, depth=prefetch, worker=worker)
        views, upcoming = iter(source), source.peek
        stat = _LoaderStat(source)
        LOADER_STATS.append(stat)
    else:
        views, upcoming, stat = ((i, get_view(i)) for i in order), None, None
    try:
        for i, items in views:
            first = ViewBatch(items, new_view=True)
            first.upcoming = upcoming
            yield first
            rest = ViewBatch(items, new_view=False)
            rest.upcoming = upcoming
            for _ in range(count(i) - 1):
                yield rest
    finally:
        if stat is not None:
            stat.settle()      # keep the two numbers, let go of the prefetcher and its pinned ring


class _ReduceOp:
    SUM, MAX = "sum", "max"


class _Work:
    """Handle of an asynchronous collective on the communicator's side stream: ``wait()`` orders the caller's current
    stream behind it (no host block), as torch.distributed's NCCL work objects do."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        import torch
        torch.cuda.current_stream().wait_event(self.event)


class RcclComm:
    """The product's own RCCL communicator (csrc/comm.hip: ``sm_comm_init`` = ncclCommInitRank, one per process /
    GPU), exposing the slice of the ``torch.distributed`` module interface the reducers use (``all_reduce`` with
    ``ReduceOp.SUM`` on fp32 / ``ReduceOp.MAX`` on int32, ``async_op``), so that ``SparseGradReducer`` runs unchanged
    over it. Synchronous collectives are enqueued on the CURRENT HIP stream - ordered with the scatter that wrote the
    gradient and the fused update that reads it, with no stream hop - asynchronous ones on a side stream.
    ``torch.distributed`` (any backend) only carries the 128-byte unique id at start-up."""
    ReduceOp = _ReduceOp

    def __init__(self, dist_module, rank: int, world_size: int, device):
        """COLLECTIVE over ``dist_module``: every rank runs the same sequence of ``torch.distributed`` calls whether or
        not its own part succeeds (the id broadcast carries an empty payload when rank 0 could not make one; the
        outcome of ``ncclCommInitRank`` is agreed on with a MIN all-reduce), so that a failure on SOME ranks ends in
        the same exception on ALL of them instead of mismatched collectives."""
        import ctypes
        import torch
        from . import hip
        self._hip, self._torch = hip, torch
        self.rank, self.world_size, self.device = rank, world_size, device
        self.handle, self._side = None, None
        nbytes = hip.lib.sm_comm_unique_id_bytes()
        buf = ctypes.create_string_buffer(nbytes)
        err = None
        if rank == 0:
            rc = hip.lib.sm_comm_get_unique_id(buf)
            if rc != 0:
                err = f"sm_comm_get_unique_id failed ({rc})"
        box = [bytes(buf.raw) if err is None else b""]
        if world_size > 1:
            dist_module.broadcast_object_list(box, src=0)
        if err is None and len(box[0]) != nbytes:
            err = "rank 0 could not create the RCCL unique id"
        handle = ctypes.c_void_p()
        if err is None:
            import contextlib
            with (torch.cuda.device(device) if device is not None else contextlib.nullcontext()):
                rc = hip.lib.sm_comm_init(ctypes.byref(handle), world_size, box[0], rank)
            if rc != 0:
                err, handle = f"sm_comm_init (ncclCommInitRank) failed ({rc})", ctypes.c_void_p()
        if world_size > 1:   # agree on the outcome
            ok = torch.tensor([0 if err else 1], dtype=torch.int32,
                              device=device if str(dist_module.get_backend()) == "nccl" else "cpu")
            dist_module.all_reduce(ok, op=dist_module.ReduceOp.MIN)
            if int(ok) == 0 and err is None:
                err = "another rank could not join the RCCL communicator"
        if err is not None:
            if handle:
                hip.lib.sm_comm_destroy(handle)
            raise RuntimeError(err)
        self.handle = handle

    def _launch(self, tensor, op):
        hip, torch = self._hip, self._torch
        assert tensor.is_cuda and tensor.is_contiguous()
        if op == _ReduceOp.SUM and tensor.dtype == torch.float32:
            hip.check(hip.lib.sm_allreduce_grad(self.handle, tensor.data_ptr(), tensor.numel(), hip.stream()),
                      "sm_allreduce_grad")
        elif op == _ReduceOp.MAX and tensor.dtype == torch.int32:
            hip.check(hip.lib.sm_allreduce_flags_max(self.handle, tensor.data_ptr(), tensor.numel(), hip.stream()),
                      "sm_allreduce_flags_max")
        else:
            raise ValueError(f"unsupported collective: {op} on {tensor.dtype}")

    def all_reduce(self, tensor, op=_ReduceOp.SUM, async_op=False):
        torch = self._torch
        if not async_op:
            self._launch(tensor, op)
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        ready = torch.cuda.Event()
        ready.record()
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            self._launch(tensor, op)
            done = torch.cuda.Event()
            done.record()
        tensor.record_stream(self._side)
        return _Work(done)

    def info(self):
        """What RCCL says the communicator is (``sm_comm_info``) and how the runtime links this rank's device to the others
        (``sm_device_link``: 4 = xGMI, 1 = PCIe) - the record a multi-GPU run leaves behind."""
        import ctypes
        hip = self._hip
        raw = (ctypes.c_int * 4)()
        hip.check(hip.lib.sm_comm_info(self.handle, raw), "sm_comm_info")
        out = {"nranks": raw[0], "rank": raw[1], "device": raw[2], "rccl_version": raw[3]}
        links = []
        n_dev = self._torch.cuda.device_count()
        for other in range(min(self.world_size, n_dev)):
            if other == raw[2]:
                continue
            t, h = ctypes.c_int(), ctypes.c_int()
            if hip.lib.sm_device_link(raw[2], other, ctypes.byref(t), ctypes.byref(h)) == 0:
                links.append({"to_device": other, "type": {4: "xgmi", 1: "pcie"}.get(t.value, str(t.value)), "hops": h.value})
        out["links"] = links
        return out

    def destroy(self):
        if self.handle:
            self._hip.lib.sm_comm_destroy(self.handle)
            self.handle = None


def make_comm(dist_module, rank: int, world_size: int, device, kind=None):
    """The collective provider of the gradient exchange. ``kind`` (default: env STYLEMESH_COMM, else 'rccl' when the
    process group's backend is nccl = RCCL): 'rccl' = the product's own communicator (``RcclComm``); 'torch' = the
    ``torch.distributed`` module itself (the only choice over gloo: CPU tests, two ranks sharing one GPU).
    The decision is COLLECTIVE: ``RcclComm`` either comes up on every rank or raises on every rank. When 'rccl' was
    asked for explicitly (argument or STYLEMESH_COMM=rccl) that exception propagates - the job fails loudly; when it
    was only the default, every rank prints the reason and uses ``torch.distributed`` (same interface)."""
    if world_size <= 1:
        return None
    explicit = kind or os.environ.get("STYLEMESH_COMM")
    kind = explicit
    if kind is None:
        kind = "rccl" if str(dist_module.get_backend()) == "nccl" else "torch"
    if kind not in ("rccl", "torch"):
        raise ValueError(f"STYLEMESH_COMM / kind must be 'rccl' or 'torch', not {kind!r}")
    if kind == "rccl":
        try:
            return RcclComm(dist_module, rank, world_size, device)
        except RuntimeError as e:
            if explicit == "rccl":
                raise
            import sys
            print(f"[stylemesh_amd] rank {rank}: own RCCL communicator unavailable on some rank ({e}); every rank uses "
                  "torch.distributed", file=sys.stderr)
    return dist_module


def make_grad_reducer(dist_module, world_size: int):
    """In-place SUM all-reduce of the gradient arena (None for a single rank)."""
    if world_size <= 1:
        return None

    def reduce(flat_grad):
        dist_module.all_reduce(flat_grad, op=dist_module.ReduceOp.SUM)
    return reduce


class SparseGradReducer:
    """SUM all-reduce of the gradient arena restricted to the chunks any rank's current view can touch.

    A view writes gradient only where its UV maps land (a few % - tens of % of a 4096^2 hierarchical texture), and
    which chunks those are depends only on the view, not on the step. Per VIEW (a collective, driven by the schedule
    position - the ranks' schedules change views in lock-step, ``rank_schedule``): the per-rank touch flags are
    max-all-reduced and compacted ON THE DEVICE into the ascending list of flagged chunks (``sm_flags_compact``:
    deterministic, so identical flags give identical lists on every rank); only the COUNT crosses to the host - RCCL
    needs the element count of the all-reduce - and it rides the one read-back ``StepEngine.set_view`` does anyway
    (``new_view_begin`` / ``new_view_end``). Per STEP: the flagged chunks are gathered into a compact buffer
    (``sm_chunks_gather``), that buffer is all-reduced, and the sums are copied back (``sm_chunks_scatter``) - three
    launches on the caller's stream, no host synchronisation. Untouched chunks are zero on every rank and stay zero.
    Falls back to the dense all-reduce when most chunks are dirty or the arena is not a whole number of chunks.
    Chunks are 64 floats (256 B): a view's footprint is a 2-D blob of scattered texels in row-major planes, so 4 KB
    chunks flag 49 % of the arena where 256 B chunks flag 14-24 % (exact non-zero fraction 8-11 %; bench views,
    tools/touch_fraction.py). CPU tensors (the gloo tests of the protocol) take the same steps with torch indexing.
    """

    def __init__(self, dist_module, world_size: int, chunk_log2: int = 6, dense_above: float = 0.75, n_pieces: int = 4):
        self.dist, self.world, self.chunk_log2, self.dense_above = dist_module, world_size, chunk_log2, dense_above
        self.n_pieces = n_pieces
        self.idx = None           # flagged chunk indices, ascending (device tensor; only the first n_idx entries count)
        self.n_idx = 0
        self.count_dev = None     # device int32 [1]: n_idx as the compaction left it
        self.cuts = None          # piece k exchanges idx[cuts[k]:cuts[k+1]] ...
        self.bounds = None        # ... and finishes the arena range [bounds[k], bounds[k+1])
        self.fraction = 1.0
        self.last_bytes = 0
        self._buf = None
        self._ws = None

    @property
    def chunk(self):
        return 1 << self.chunk_log2

    def n_chunks(self, numel: int) -> int:
        return -(-numel // self.chunk)

    # ---- per view ------------------------------------------------------------------------------------------
    def new_view_begin(self, flags):
        """``flags``: int32 [n_chunks], non-zero where THIS rank's view can write; on return the union over the ranks
        (in place). Collective. Enqueues the compaction; returns the device tensor that will hold the count (int32 [1])
        - read it back whenever convenient and pass the value to ``new_view_end``."""
        import torch
        self.dist.all_reduce(flags, op=self.dist.ReduceOp.MAX)
        n = flags.numel()
        if flags.is_cuda:
            from . import ops
            if self.idx is None or self.idx.numel() < n or self.idx.device != flags.device:
                self.idx = torch.empty(n, dtype=torch.int32, device=flags.device)
                self._ws = torch.empty(ops.flags_compact_ws_ints(n), dtype=torch.int32, device=flags.device)
                self.count_dev = torch.zeros(1, dtype=torch.int32, device=flags.device)
            ops.flags_compact(flags, self.idx, self.count_dev, self._ws)
        else:
            self.idx = flags.nonzero().flatten().to(torch.int32)
            self.count_dev = torch.tensor([self.idx.numel()], dtype=torch.int32)
        self._n_flags = n
        self.n_idx = None          # unknown until new_view_end
        return self.count_dev

    def new_view_end(self, count: int):
        self.n_idx = int(count)
        self.fraction = self.n_idx / max(self._n_flags, 1)
        self.cuts = self.bounds = None    # the pipelined exchange derives its pieces on first use

    def new_view(self, flags):
        """``new_view_begin`` + one host read-back of the count + ``new_view_end`` (callers without a read-back of
        their own to ride on)."""
        self.new_view_end(int(self.new_view_begin(flags)))

    def _pieces(self):
        if self.cuts is None:
            n_idx = self.n_idx
            k = max(1, min(self.n_pieces, n_idx))
            self.cuts = [n_idx * j // k for j in range(k + 1)]
            # arena range a piece completes: from its first flagged chunk up to the next piece's first flagged chunk
            # (everything between flagged chunks is zero on every rank and needs no exchange); one small read-back
            firsts = self.idx[self.cuts[1:-1]].tolist() if k > 1 else []
            self.bounds = [0] + [int(f) * self.chunk for f in firsts] + [None]
        return self.cuts, self.bounds

    # ---- per step ------------------------------------------------------------------------------------------
    def _sparse(self, n):
        return not (self.idx is None or self.n_idx is None or self.fraction > self.dense_above or n % self.chunk != 0)

    def _gather(self, flat_grad, lo=0, hi=None):
        """compact buffer holding chunks idx[lo:hi] of the arena"""
        import torch
        hi = self.n_idx if hi is None else hi
        if flat_grad.is_cuda:
            from . import ops
            if self._buf is None or self._buf.numel() < self.n_idx * self.chunk or self._buf.device != flat_grad.device:
                self._buf = torch.empty(self.n_idx * self.chunk, dtype=torch.float32, device=flat_grad.device)
            buf = self._buf[lo * self.chunk:hi * self.chunk]
            ops.chunks_gather(flat_grad, self.idx[lo:hi], hi - lo, self.chunk_log2, buf)
            return buf
        return flat_grad.view(-1, self.chunk).index_select(0, self.idx[lo:hi].long()).reshape(-1)

    def _scatter(self, flat_grad, buf, lo=0, hi=None):
        hi = self.n_idx if hi is None else hi
        if flat_grad.is_cuda:
            from . import ops
            ops.chunks_scatter(flat_grad, self.idx[lo:hi], hi - lo, self.chunk_log2, buf)
        else:
            flat_grad.view(-1, self.chunk).index_copy_(0, self.idx[lo:hi].long(), buf.view(-1, self.chunk))

    def __call__(self, flat_grad):
        n = flat_grad.numel()
        if not self._sparse(n):
            self.dist.all_reduce(flat_grad, op=self.dist.ReduceOp.SUM)
            self.last_bytes = 4 * n
            return
        if self.n_idx == 0:      # no rank's view touches the texture: nothing to exchange (same on every rank)
            self.last_bytes = 0
            return
        buf = self._gather(flat_grad)
        self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)
        self._scatter(flat_grad, buf)
        self.last_bytes = 4 * buf.numel()

    def pipelined(self, flat_grad, update_range):
        """Exchange and update overlapped: the flagged chunks are all-reduced in ``n_pieces`` asynchronous pieces
        and ``update_range(lo, hi)`` - the fused optimizer over arena elements [lo, hi) - is issued for a piece as
        soon as its sums have arrived, while the later pieces are still on the links. The ranges tile [0, n) in
        order; same arithmetic as ``__call__`` followed by one update over the whole arena. OPT-IN
        (``StepEngine.pipeline_exchange`` / STYLEMESH_PIPELINE_EXCHANGE=1|auto, ``bench.py --pipeline-exchange``): verified
        functionally and bit for bit (2 and 8 ranks over gloo, CPU and one shared GPU), never timed over RCCL on separate
        GPUs - the 1-GPU boxes cannot - so the plain exchange-then-update, which keeps the measured early half of the split
        update, stays the default."""
        n = flat_grad.numel()
        dist = self.dist
        sparse = self._sparse(n)
        if sparse and self.n_idx == 0:
            self.last_bytes = 0
            update_range(0, n)
            return
        if not sparse:
            k = self.n_pieces
            align = 1024
            bounds = sorted({min(n, (n * j // k) // align * align) for j in range(k)} | {0, n})   # no empty pieces
            works = [dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
                     for lo, hi in zip(bounds, bounds[1:])]
            for w, lo, hi in zip(works, bounds, bounds[1:]):
                w.wait()
                update_range(lo, hi)
            self.last_bytes = 4 * n
            return
        cuts, bounds = self._pieces()
        buf = self._gather(flat_grad)
        k = len(cuts) - 1
        c = self.chunk
        works = [dist.all_reduce(buf[cuts[j] * c:cuts[j + 1] * c], op=dist.ReduceOp.SUM, async_op=True) for j in range(k)]
        for j in range(k):
            works[j].wait()
            self._scatter(flat_grad, buf[cuts[j] * c:cuts[j + 1] * c], cuts[j], cuts[j + 1])
            update_range(bounds[j], n if bounds[j + 1] is None else bounds[j + 1])
        self.last_bytes = 4 * buf.numel()


class OwnerAwareGradReducer(SparseGradReducer):
    """``SparseGradReducer`` that takes the chunks only ONE rank's view touches off the step's critical path (round 6,
    VERDICT r5 item 8).

    At N = 8 the union of the ranks' footprints is 75 % of the arena (profiles/r05/final_bench_n8_gloo.json) and every
    chunk of it crossed the links before the update - although a chunk only one rank's view reaches needs NO reduction: its
    sum over the ranks is that rank's own gradient (the others hold zeros there: x + 0 = x, bit for bit), and no other rank
    samples it before its own view changes. Per VIEW the collective therefore also finds, for every chunk of the union,
    whether exactly one rank touches it (one MAX all-reduce of two int32 words per chunk: the highest and - mirrored - the
    lowest rank that flags it; equal = a single owner). Three compact lists come out of it, identical on every rank:

    * ``idx`` (base class): the union - what the plain exchange and ``pipelined`` use;
    * ``idx_shared``: chunks two or more ranks touch - the CRITICAL exchange of a step, all-reduced before the update;
    * ``idx_single``: chunks with one owner - the DEFERRED exchange.

    Per STEP (``step``; the engine's ``exchange_and_update_deferred``), in this order on the caller's stream:

    1. the deferred sums of the PREVIOUS step, if any, have arrived on the side stream: scatter the other ranks' single
       chunks into the gradient arena and update them with THAT step's learning rate and step count (``update``);
    2. stage this step's deferred buffer (the gradients of all single chunks: this rank's own, zeros elsewhere), then update
       this rank's OWN single chunks at once from its own gradient;
    3. all-reduce the shared chunks (gather / all-reduce / scatter as the base class) and update them;
    4. start the deferred all-reduce asynchronously: it overlaps the next step's forward and backward passes.

    ``drain`` (step 1 alone) runs before every per-view collective and at the end of training. After a drain the textures
    and Adam moments are those of exchange-then-update: every chunk received the same sums and the same sequence of
    updates - the single ones one step later on the ranks that do not own them (tests/test_distributed_cpu.py: bit for
    bit on two ranks; on more ranks the SHARED sums may differ in the last bit where the collective's summation order
    depends on a chunk's offset in the buffer). The regulariser loss VALUE a non-owner reports lags one step on those
    chunks (their p^2 enters ``sumsq`` when their update is applied)."""

    owner_aware = True

    def __init__(self, dist_module, world_size: int, rank: int, deferred_dist=None, **kw):
        """``deferred_dist``: the collective provider of the deferred exchange - a SECOND communicator (``make_comm`` again /
        a second process group), so that a deferred all-reduce still on the links cannot hold up the next step's critical
        one (collectives of one communicator are serialised in issue order); default: the same provider."""
        super().__init__(dist_module, world_size, **kw)
        self.rank = rank
        self.deferred_dist = dist_module if deferred_dist is None else deferred_dist
        self.idx_shared = self.idx_single = None
        self.n_shared = self.n_single = None
        self.f_shared = self.f_single_mine = self.f_single_others = None
        self._counts_dev = None
        self._pending = None          # (work, buffer, lr, step, flags of the others' single chunks, their (buffer row, chunk) lists)
        self._bufs = [None, None]     # deferred send buffers, alternating
        self._flip = 0
        self._buf_shared = None
        self.last_critical_bytes = self.last_deferred_bytes = 0

    @property
    def ready(self):
        """lists of the current view known, and sparse enough for chunk lists to pay (else: the plain dense exchange)"""
        return self.n_shared is not None and self.n_idx is not None and self.fraction <= self.dense_above

    # ---- per view ------------------------------------------------------------------------------------------
    def new_view_begin(self, flags):
        import torch
        assert self._pending is None, "drain() the deferred exchange before the next per-view collective"
        n, R = flags.numel(), self.world
        mine = flags != 0
        enc = torch.zeros(2 * n, dtype=torch.int32, device=flags.device)
        enc[:n] = mine.to(torch.int32) * (self.rank + 1)         # -> highest owner + 1
        enc[n:] = mine.to(torch.int32) * (R - self.rank)         # -> R - lowest owner
        self.dist.all_reduce(enc, op=self.dist.ReduceOp.MAX)
        hi, lo_m = enc[:n], enc[n:]
        union = hi > 0
        single = union & (hi + lo_m == R + 1)                    # highest owner == lowest owner
        self.f_shared = (union & ~single).to(torch.int32)
        self.f_single_mine = (single & mine).to(torch.int32)
        self.f_single_others = (single & ~mine).to(torch.int32)
        f_single = single.to(torch.int32)
        flags.copy_(union.to(torch.int32))                       # (contract of the base class: the union, in place)
        # the union list exactly as the base class builds it (MAX of identical flags changes nothing)
        n_flags = n
        if flags.is_cuda:
            from . import ops
            if self.idx is None or self.idx.numel() < n or self.idx.device != flags.device:
                self.idx = torch.empty(n, dtype=torch.int32, device=flags.device)
                self._ws = torch.empty(ops.flags_compact_ws_ints(n), dtype=torch.int32, device=flags.device)
                self.count_dev = torch.zeros(1, dtype=torch.int32, device=flags.device)
            if self.idx_shared is None or self.idx_shared.numel() < n or self.idx_shared.device != flags.device:
                self.idx_shared = torch.empty(n, dtype=torch.int32, device=flags.device)
                self.idx_single = torch.empty(n, dtype=torch.int32, device=flags.device)
                self._counts_dev = torch.zeros(2, dtype=torch.int32, device=flags.device)
            ops.flags_compact(flags, self.idx, self.count_dev, self._ws)
            ops.flags_compact(self.f_shared, self.idx_shared, self._counts_dev[0:1], self._ws)
            ops.flags_compact(f_single, self.idx_single, self._counts_dev[1:2], self._ws)
        else:
            self.idx = flags.nonzero().flatten().to(torch.int32)
            self.count_dev = torch.tensor([self.idx.numel()], dtype=torch.int32)
            self.idx_shared = self.f_shared.nonzero().flatten().to(torch.int32)
            self.idx_single = f_single.nonzero().flatten().to(torch.int32)
            self._counts_dev = torch.tensor([self.idx_shared.numel(), self.idx_single.numel()], dtype=torch.int32)
        self._n_flags = n_flags
        self.n_idx = self.n_shared = self.n_single = None
        return self.count_dev

    def new_view_end(self, count: int):
        super().new_view_end(count)
        self.n_shared, self.n_single = (int(x) for x in self._counts_dev.tolist())      # (one small read-back per view)
        import torch
        # rows of the deferred buffer (= entries of idx_single) that belong to OTHER ranks, and their chunks
        ids = self.idx_single[:self.n_single].long()
        sel = (self.f_single_mine[ids] == 0).nonzero().flatten()
        self._others_rows, self._others_idx = sel, self.idx_single[:self.n_single][sel].contiguous()

    # ---- per step ------------------------------------------------------------------------------------------
    def _move(self, flat_grad, idx, n, buf, gather):
        if n == 0:
            return
        if flat_grad.is_cuda:
            from . import ops
            (ops.chunks_gather if gather else ops.chunks_scatter)(flat_grad, idx, n, self.chunk_log2, buf)
        elif gather:
            buf.copy_(flat_grad.view(-1, self.chunk).index_select(0, idx[:n].long()).reshape(-1))
        else:
            flat_grad.view(-1, self.chunk).index_copy_(0, idx[:n].long(), buf.view(-1, self.chunk))

    def drain(self, flat_grad, update):
        """Step 1: apply the deferred sums that are outstanding (no-op when there are none)."""
        if self._pending is None:
            return
        work, buf, lr, step, f_others, rows, idx_others = self._pending
        self._pending = None
        if work is not None:
            work.wait()                                           # (stream-ordered on the device path; blocking over gloo)
        if rows.numel() > 0:
            part = buf.view(-1, self.chunk).index_select(0, rows).reshape(-1)
            self._move(flat_grad, idx_others, rows.numel(), part, gather=False)
            update(f_others, lr, step)

    def step(self, flat_grad, update, lr, step):
        """Steps 1 - 4 of the class docstring. ``update(flags, lr, step)``: the fused optimizer over the chunks whose int32
        flag is set, with the given learning rate and step count (it zeroes the gradient there)."""
        import torch
        assert self.ready
        self.drain(flat_grad, update)
        c = self.chunk
        # 2. the deferred buffer, then the own single chunks
        buf = None
        if self.n_single > 0:
            b = self._bufs[self._flip]
            if b is None or b.numel() < self.n_single * c or b.device != flat_grad.device:
                b = self._bufs[self._flip] = torch.empty(self.n_single * c, dtype=torch.float32, device=flat_grad.device)
            buf = b[:self.n_single * c]
            self._flip ^= 1
            self._move(flat_grad, self.idx_single, self.n_single, buf, gather=True)
            update(self.f_single_mine, lr, step)
        # 3. the critical exchange
        if self.n_shared > 0:
            if self._buf_shared is None or self._buf_shared.numel() < self.n_shared * c or self._buf_shared.device != flat_grad.device:
                self._buf_shared = torch.empty(self.n_shared * c, dtype=torch.float32, device=flat_grad.device)
            sb = self._buf_shared[:self.n_shared * c]
            self._move(flat_grad, self.idx_shared, self.n_shared, sb, gather=True)
            self.dist.all_reduce(sb, op=self.dist.ReduceOp.SUM)
            self._move(flat_grad, self.idx_shared, self.n_shared, sb, gather=False)
            update(self.f_shared, lr, step)
        # 4. the deferred exchange, in the background
        if buf is not None:
            work = self.deferred_dist.all_reduce(buf, op=self.deferred_dist.ReduceOp.SUM, async_op=True)
            self._pending = (work, buf, lr, step, self.f_single_others, self._others_rows, self._others_idx)
        self.last_critical_bytes = 4 * self.n_shared * c
        self.last_deferred_bytes = 4 * self.n_single * c
        self.last_bytes = self.last_critical_bytes + self.last_deferred_bytes


def make_sparse_grad_reducer(dist_module, world_size: int, rank: int | None = None, **kw):
    """``rank`` given: the owner-aware reducer (critical / deferred exchange, opt-in through the engine:
    STYLEMESH_DEFERRED_EXCHANGE=1); else the plain union reducer."""
    if world_size <= 1:
        return None
    if rank is not None:
        return OwnerAwareGradReducer(dist_module, world_size, rank, **kw)
    return SparseGradReducer(dist_module, world_size, **kw)
