"""Host-side profile (cProfile) of the CLI training path on a small on-disk scene: where does the training THREAD spend its
time per step? Run on the GPU box. Usage: profile_cli_host.py [c2|c3] [views=41]"""
import cProfile, io, os, pstats, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import run_schedule as RS
from stylemesh_amd.model import optimize as OPT

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
views = int(sys.argv[2]) if len(sys.argv) > 2 else 41
if __name__ == "__main__":
    root = tempfile.mkdtemp(prefix="stylemesh_scene_")
    RS.write_scene(root, "scene0000_00", views, [256, 432, 608, 784] if wl == "c3" else [256])
    argv = ["--gpus", "1", "--root_path", root, "--dataset", "scannet", "--resize_size", "256", "--min_images", "1",
            "--max_images", "1000", "--scene", "scene0000_00", "--hierarchical", "--hierarchical_layers", "4",
            "--loss_weight", "content=7e1", "--loss_weight", "style=1e-4", "--style_weights=1000,1000,10,10,1000",
            "--loss_weight", "tex_reg=5e3", "--vgg_gatys_model_path", "random:0", "--learning_rate", "1", "--decay_step_size", "3",
            "--max_epochs", "2", "--train_split", "0.99", "--val_split", "0.01", "--sampler_mode", "repeat", "--index_repeat", "20",
            "--num_workers", os.environ.get("NW", "4"), "--style_image_path", "synthetic:1:1528x1200", "--gram_mode", "current",
            "--min_pyramid_depth", "0.25", "--min_pyramid_height", "256", "--default_root_dir", os.path.join(root, "logs")] + RS.FLAGS[wl]
    args = OPT.build_parser().parse_args(argv)
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    OPT.main(args)
    pr.disable()
    print(f"wall {time.time() - t0:.1f} s")
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(os.environ.get("SORT", "tottime")).print_stats(int(os.environ.get("TOP", "28")))
    print(s.getvalue()[:6000])
