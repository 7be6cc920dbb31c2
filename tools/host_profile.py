"""Where the HOST time of a bench workload's step loop goes: cProfile over the timed loop (+ wall per step, GPU-synchronised).
Usage: host_profile.py <workload> [steps=200]"""
import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from stylemesh_amd.data import synthetic as S
from stylemesh_amd.runtime.engine import StepEngine, trunk_stream

wl_name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wl = B.WORKLOADS[wl_name]
eng = StepEngine(B.engine_config(wl), S.seeded_vgg_state(0))
eng.set_style_image(S.style_image(1, *B.STYLE_HW))
rep = wl["index_repeat"]
good = (0, 2, 6, 7, 9, 11, 12, 14, 16, 18, 22, 23)
views = [B.to_device(v, "cuda") for v in B.make_views(wl, good)]
sched = [views[(i // rep) % len(views)] for i in range(steps + 40)]
st = trunk_stream(torch.device("cuda", 0))
if st is not None:
    torch.cuda.set_stream(st)

def step(i):
    if rep > 1 and i % rep == 1 and i - 1 + rep < len(sched):
        eng.prepare_view(sched[i - 1 + rep])
    nxt = sched[i + 1] if (rep == 1 and i + 1 < len(sched)) else None
    eng.training_step(sched[i], new_view=(i % rep == 0), next_batch=nxt)

for i in range(40):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40, 40 + steps):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{wl_name}: host enqueue {1e3 * (t1 - t0) / steps:.3f} ms/step, wall {1e3 * (t2 - t0) / steps:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(40, 40 + steps):
    step(i)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
