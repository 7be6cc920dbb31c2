"""Single-node rank launcher of the view-sharded path (SURVEY.md section 8 e): ``python bench.py --gpus N`` with no
``WORLD_SIZE`` in the environment starts N FRESH rank processes (one per GPU: RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT as ``torch.distributed.run`` would set them), relays rank 0's standard output and exits with
the ranks' worst return code. The reference runs ``--gpus 1`` only (scripts/train/*.sh:1, model/optimize.py:241), so
this is the build's own entry point for BASELINE configs 4 (``--replicas``) and 5.

Standard library only, and the decision is taken BEFORE anything in the parent touches the GPU: the parent never
initialises HIP, never re-executes itself, and only ever signals the exact child processes it started.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time

RC_NO_DEVICES = 3      # fewer GPUs than ranks (nccl backend)
RC_TIMEOUT = 124


def needs_launch(n_gpus: int, env) -> bool:
    """True when this process was started plainly (no launcher set WORLD_SIZE) although N > 1 ranks are asked for."""
    return n_gpus > 1 and "WORLD_SIZE" not in env


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def check_devices(n_gpus: int, device_count: int, backend: str):
    """None if N ranks can each have a GPU, else the message to print. ``STYLEMESH_DIST_BACKEND=gloo`` is the functional
    mode in which the ranks share the devices that exist (two ranks on one GPU box)."""
    if backend == "nccl" and device_count < n_gpus:
        return (f"bench.py --gpus {n_gpus}: this node exposes {device_count} GPU(s); one rank per GPU over RCCL needs "
                f"{n_gpus}. (STYLEMESH_DIST_BACKEND=gloo runs the N-rank protocol with the ranks sharing the visible "
                "device(s): a functional check, not a measurement.)")
    if device_count < 1:
        return f"bench.py --gpus {n_gpus}: no GPU visible"
    return None


def rank_plans(n_gpus: int, argv, env, port: int, python=None):
    """[(command, environment)] of the N rank processes. ``argv`` = the script and its arguments as given to the parent."""
    python = python or sys.executable
    plans = []
    for r in range(n_gpus):
        e = dict(env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_gpus), "LOCAL_WORLD_SIZE": str(n_gpus),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "STYLEMESH_LAUNCHED_BY": "bench.py"})
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL across processes needs it on this pool
        e.setdefault("OMP_NUM_THREADS", "4")
        plans.append(([python] + list(argv), e))
    return plans


def _relay(pipe, sink, prefix):
    for line in iter(pipe.readline, ""):
        sink.write(prefix + line)
        sink.flush()
    pipe.close()


def worst_rc(codes) -> int:
    """Exit status of the job: 0 only if every rank returned 0; else the first non-zero code in rank order, a signal's
    negative code mapped to 128 + signal as a shell would."""
    for c in codes:
        if c:
            return 128 - c if c < 0 else c
    return 0


def run_ranks(plans, timeout_s=None, out=None, err=None, poll_s=0.2, grace_s=10.0) -> int:
    """Start the rank processes, relay rank 0's stdout to ``out`` (the ONE JSON line) and everything else to ``err``;
    when a rank fails the others are given ``grace_s`` to notice (a collective that lost a peer raises) and are then
    terminated - by PID, only the processes started here."""
    out, err = out or sys.stdout, err or sys.stderr
    procs, threads = [], []
    for r, (cmd, env) in enumerate(plans):
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, bufsize=1)
        procs.append(p)
        for pipe, sink, prefix in ((p.stdout, out if r == 0 else err, "" if r == 0 else f"[rank {r}] "),
                                   (p.stderr, err, f"[rank {r}] ")):
            t = threading.Thread(target=_relay, args=(pipe, sink, prefix), daemon=True)
            t.start()
            threads.append(t)
    t0, failed_at, timed_out = time.monotonic(), None, False
    while any(p.poll() is None for p in procs):
        now = time.monotonic()
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = now
        if timeout_s is not None and now - t0 > timeout_s and not timed_out:
            timed_out, failed_at = True, now - grace_s
        if failed_at is not None and now - failed_at >= grace_s:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.monotonic() + 5.0
            while any(p.poll() is None for p in procs) and time.monotonic() < deadline:
                time.sleep(poll_s)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(poll_s)
    for t in threads:
        t.join(timeout=2.0)
    if timed_out:
        return RC_TIMEOUT
    return worst_rc([p.returncode for p in procs])


def kfd_gpu_count(topology="/sys/class/kfd/kfd/topology/nodes", env=None, dri="/dev/dri"):
    """GPUs this process can use, counted WITHOUT the HIP / HSA runtime (the launcher's parent must not open /dev/kfd
    before it starts the ranks; ``torch.cuda.device_count()`` falls through to ``hipGetDeviceCount`` on ROCm builds without
    amdsmi): KFD topology nodes with SIMDs (CPU nodes report ``simd_count 0``) whose properties are readable and
    whose render node exists under ``/dev/dri`` (a container is given ONE GPU of an eight-GPU node by cgroup device
    rules: the other nodes' properties answer 'Operation not permitted', their render nodes are absent), narrowed by
    ``HIP_VISIBLE_DEVICES`` / ``ROCR_VISIBLE_DEVICES``. 0 without a KFD; None when nothing can be said (a malformed
    topology) - the ranks then refuse by themselves."""
    env = os.environ if env is None else env
    try:
        nodes = sorted(os.listdir(topology))
    except FileNotFoundError:
        return 0         # no KFD: no AMD GPU on this node
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(topology, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue
            minor = int(props.get("drm_render_minor", "-1"))
        except PermissionError:
            continue     # a device of the node this container was not given
        except (OSError, ValueError):
            return None
        if dri is None or minor < 0 or os.path.exists(os.path.join(dri, f"renderD{minor}")):
            n += 1
    # Visibility lists narrow the count in the runtime's own order: ROCR_VISIBLE_DEVICES first (the ROCr layer),
    # then HIP_VISIBLE_DEVICES - of which CUDA_VISIBLE_DEVICES is an alias HIP only reads when the former is unset - over what
    # ROCr left. Only integer indices BELOW the count before them are counted (an index the container does not have names
    # no device); a list with anything else in it (UUIDs) cannot be judged from sysfs: None, the ranks decide.
    def narrow(count, value):
        if value is None:
            return count
        items = [x.strip() for x in value.split(",") if x.strip() != ""]
        if any(not x.lstrip("-").isdigit() for x in items):
            return None
        seen = []
        for x in items:
            i = int(x)
            if i < 0 or i >= count:
                break            # (the runtime stops at the first invalid index)
            if i not in seen:
                seen.append(i)
        return len(seen)
    n = narrow(n, env.get("ROCR_VISIBLE_DEVICES"))
    if n is None:
        return None
    hv = env.get("HIP_VISIBLE_DEVICES")
    return narrow(n, hv if hv is not None else env.get("CUDA_VISIBLE_DEVICES"))


def launch(n_gpus: int, argv, env=None, device_count=None, backend=None, timeout_s=None, out=None, err=None) -> int:
    """The parent's whole job. ``device_count``: callable or int, only used here, in the parent, to refuse early with ONE
    clear message - ``kfd_gpu_count`` by default in bench.py: it reads sysfs and never touches the runtime (None: no
    check, every rank refuses on its own when it finds fewer devices than ranks)."""
    env = dict(os.environ if env is None else env)
    backend = backend or env.get("STYLEMESH_DIST_BACKEND", "nccl")
    n_dev = device_count() if callable(device_count) else device_count
    if n_dev is not None:
        msg = check_devices(n_gpus, n_dev, backend)
        if msg is not None:
            print(msg, file=err or sys.stderr)
            return RC_NO_DEVICES
    return run_ranks(rank_plans(n_gpus, argv, env, free_port()), timeout_s=timeout_s, out=out, err=err)
