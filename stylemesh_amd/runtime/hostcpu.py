"""How many CPUs this process may really use, and a cap on torch's intra-op threads derived from it.

A GPU node shows hundreds of hardware threads (256 on the MI355X boxes) while the container's CPU controller grants a
fraction (cpu.max: 16 CPUs here). torch sizes its intra-op pool by the VISIBLE count: every small CPU operation of the
training process (the logger's ``torch.stack``, the 14 MB copy into the pinned ring) wakes a pool of spinning threads,
the group burns its quota within a few milliseconds of every 100 ms period and the kernel THROTTLES all of it - the
launch loop included. Measured in round 4 on the CLI path of the one-level workload: the GPU idle 0.9 ms of every 2.1 ms
step, two thirds of it in gaps >= 5 ms, cpu.stat nr_throttled 1455 of 2163 periods. The training process needs a handful
of CPU threads; ``limit_host_threads`` caps the pool (STYLEMESH_HOST_THREADS overrides)."""
from __future__ import annotations

import os


def effective_cpus() -> float:
    """min(CPUs in the affinity mask, CPU-controller quota); cgroup v2 ``cpu.max`` or v1 ``cpu.cfs_quota_us``."""
    n = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, float(quota) / float(period))
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, q / p)
        except (OSError, ValueError):
            pass
    return max(n, 1.0)


def limit_host_threads(cap: int = 4) -> int:
    """Cap torch's intra-op pool of THIS process at min(cap, a quarter of the effective CPUs) (at least 1); returns the
    thread count in force. The decode workers are processes of their own (one thread each)."""
    import torch
    env = os.environ.get("STYLEMESH_HOST_THREADS")
    want = int(env) if env else max(1, min(cap, int(effective_cpus() // 4) or 1))
    if torch.get_num_threads() > want:
        torch.set_num_threads(want)
    return torch.get_num_threads()
