"""Host time per training step of the LightningModule mirror against the bare engine, one c2-sized view (GPU box)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from stylemesh_amd.data import synthetic as S
from stylemesh_amd.model.model import TextureOptimizationStyleTransferPipeline
from stylemesh_amd.trainer import JsonlLogger

wl = B.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
f = tempfile.NamedTemporaryFile(suffix=".pth", delete=False); torch.save(S.seeded_vgg_state(0), f.name)
m = TextureOptimizationStyleTransferPipeline(
    W=wl["tex"], H=wl["tex"], hierarchical_texture=True, hierarchical_layers=4, style_image=S.style_image(1, *B.STYLE_HW),
    style_weights=B.STYLE_WEIGHTS, vgg_gatys_model_path=f.name, use_angle_weight=wl["angle"], use_depth_scaling=wl["depth"],
    style_pyramid_mode=wl["mode"], angle_threshold=wl["thr"], save_texture=False, learning_rate=1, decay_step_size=3,
    loss_weights=dict(B.LOSS_WEIGHTS)).cuda()
m.logger = JsonlLogger(tempfile.mkdtemp())
m.fused_backward_done = True
(opt,), _ = m.configure_optimizers()
v = B.to_device(B.make_views(wl, [0])[0], "cuda")
eng = m._ensure_engine(v[0].device)

def run(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n

i = [0]
def full():
    i[0] += 1
    m.training_step(v, i[0]); opt.step()
def bare():
    eng.training_step(v)
def compute_only():
    eng.step_compute(v); eng.optimizer_step()
print("module.training_step + opt.step: host %.3f ms, wall %.3f ms" % run(full))
m._log_losses = lambda *a, **k: None
print("same without _log_losses       : host %.3f ms, wall %.3f ms" % run(full))
print("engine.training_step           : host %.3f ms, wall %.3f ms" % run(bare))
print("engine.step_compute + optimizer : host %.3f ms, wall %.3f ms" % run(compute_only))
