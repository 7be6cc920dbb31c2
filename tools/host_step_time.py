"""Host-side time of each training_step() call during a bench run (no synchronisation added): if the host enqueues a
step faster than the GPU executes it, the GPU never waits for launches. Run on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stylemesh_amd.runtime.engine import StepEngine

sys.argv = [sys.argv[0], "--steps", "40", "--warmup", "5", "--cpu-steps", "0", "--timer-every", "1000"]
orig = StepEngine.training_step
times = []
def timed(self, *a, **k):
    t = time.perf_counter()
    r = orig(self, *a, **k)
    times.append(1e3 * (time.perf_counter() - t))
    return r
StepEngine.training_step = timed
bench.main()
print("host ms per training_step call:", " ".join(f"{t:.1f}" for t in times))
