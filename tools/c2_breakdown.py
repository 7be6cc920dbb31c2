"""c2 (one 256 x 341 level, 2048^2 texture): where does a step's wall time go? Host enqueue time per step, GPU time of a
steady-state step (events around 50 steps of ONE view), set_view cost per view. Run on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from stylemesh_amd.data import synthetic as S
from stylemesh_amd.runtime.engine import EngineConfig, StepEngine
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
wl = B.WORKLOADS[name]
cfg = EngineConfig(tex_w=wl["tex"], tex_h=wl["tex"], hierarchical=True, n_layers=4, style_weights=B.STYLE_WEIGHTS,
                   angle_threshold=wl["thr"], style_pyramid_mode=wl["mode"], use_angle_weight=wl["angle"],
                   use_depth_scaling=wl["depth"], loss_weights=dict(B.LOSS_WEIGHTS), learning_rate=1.0, decay_step_size=3)
eng = StepEngine(cfg, S.seeded_vgg_state(0)); eng.set_style_image(S.style_image(1, *B.STYLE_HW))
eng.use_graphs = "--graphs" in sys.argv
views = [B.to_device(v, "cuda") for v in B.make_views(wl, [0, 2, 6])]
for v in views:
    for _ in range(5): eng.training_step(v)
torch.cuda.synchronize()
v = views[0]
for _ in range(5): eng.training_step(v)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
host = []
for _ in range(100):
    t = time.perf_counter(); eng.training_step(v); host.append(time.perf_counter() - t)
e1.record(); t_enq = time.perf_counter() - t0
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(f"{name}: 100 steps of one view: wall {1e3*wall/100:.3f} ms/step, GPU (events) {e0.elapsed_time(e1)/100:.3f} ms/step, "
      f"host enqueue {1e3*t_enq/100:.3f} ms/step (median call {1e3*sorted(host)[50]:.3f})")
ts = []
for v in views * 3:
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.set_view(v); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("set_view ms:", [round(1e3 * t, 2) for t in ts])
