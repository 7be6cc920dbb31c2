"""Per-launch table of the conv kernels of one c3 step (HIP-event timed): layer shape, active-tile fraction,
duration, algorithmic TFLOP/s. Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from stylemesh_amd.runtime import ops

sys.argv = [sys.argv[0], "--steps", "3", "--warmup", "2", "--cpu-steps", "0", "--timer-every", "1"]
# reuse bench.main() but keep the timer: monkeypatch KernelTimer.summary to dump the records of the LAST timed step
orig = ops.KernelTimer.summary
def dump(self, tag=None):
    out = orig(self, tag)
    if tag is None and not getattr(self, "_dumped", False):
        self._dumped = True
        n = len(self.records) // 3
        tot_ms = tot_f = 0.0
        for (a, b, work, tg), info in list(zip(self.records, self.info))[-n:]:
            ms = a.elapsed_time(b)
            tot_ms += ms; tot_f += work
            print(f"{tg:5s} {info:48s} {ms*1e3:8.1f} us {work/ms/1e9:7.1f} TF  {work/1e9:7.1f} GF")
        print(f"step total {tot_ms:.3f} ms, {tot_f/tot_ms/1e9:.1f} TF")
    return out
ops.KernelTimer.summary = dump
bench.main()
