"""Where does starting the decode processes spend its time? (GPU box: needs the rasteriser to write a scene)"""
import os, sys, tempfile, time, pickle
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import run_schedule as RS
from stylemesh_amd.data.scannet import ScanNetSingleSceneDataModule
if __name__ == "__main__":
    root = tempfile.mkdtemp(prefix="stylemesh_scene_")
    RS.write_scene(root, "scene0000_00", int(sys.argv[1]) if len(sys.argv) > 1 else 41, [256])
    dm = ScanNetSingleSceneDataModule(root, "scene0000_00", resize_size=256, pyramid_levels=1, min_pyramid_depth=0.25,
                                      min_pyramid_height=256, max_images=1000, split=(0.99, 0.01), index_repeat=20,
                                      sampler_mode="repeat", rank=0, world_size=1, prefetch=2, decode_workers=4)
    dm.prepare_data(); dm.setup()
    t = time.time(); b = pickle.dumps(dm.train_dataset.__getitem__); print("pickle bytes", len(b), round(time.time() - t, 3), "s")
    big = sorted(((len(pickle.dumps(v)), k) for k, v in vars(dm.train_dataset).items()), reverse=True)[:6]
    print("largest attributes", big)
    t = time.time(); dm.warm_start(); print("warm_start", round(time.time() - t, 3), "s")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    t = time.time(); p = ctx.Process(target=print, args=("child up",)); p.start(); print("bare spawn start()", round(time.time() - t, 3), "s"); p.join()
    # step by step, as DecodeProcess.__init__ does it
    from stylemesh_amd.runtime import distributed as D
    t = time.time(); qs = [ctx.Queue() for _ in range(8)]; print("8 queues", round(time.time() - t, 3))
    gv = dm.train_dataset.__getitem__
    for k in range(3):
        t = time.time()
        p = ctx.Process(target=D._decode_main, args=(gv, qs[0], qs[1]), daemon=True)
        p.start()
        print("decode process start()", k, round(time.time() - t, 3), "s")
    t = time.time(); p = ctx.Process(target=print, args=("child up", gv)); p.start(); print("spawn with dataset arg", round(time.time() - t, 3), "s"); p.join()
