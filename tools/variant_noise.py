"""How far apart are two launch compositions of the SAME step, measured where it is meaningful: on the gradient arena and
the losses after ONE forward_backward from identical state (no optimizer in between), next to the distance of a
composition from ITSELF run twice (the fp32 atomics of the Gram sums and of the scatter fallback make even that non-zero).
The numbers size the tolerances of tests/stepcmp.py.  Usage (GPU box): python tools/variant_noise.py > gpurun_out/variant_noise.txt"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "oracle")]
os.environ.setdefault("STYLEMESH_OVERLAP_MIN_PIXELS", "0")

from golden_cases import MULTIVIEW_SEEDS  # noqa: E402
from test_round3_gpu import _engine, _small_view  # noqa: E402


def one(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        eng = _engine(random_init=True)
        view = _small_view(MULTIVIEW_SEEDS[0])
        eng.set_view(view)
        eng.arena.g.zero_()
        lt = eng.loss_tensors()
        eng.forward_backward()
        torch.cuda.synchronize()
        return eng.losses(lt), eng.arena.g.clone()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def dist(a, b):
    (la, ga), (lb, gb) = a, b
    mx = float(gb.abs().max())
    err = (ga - gb).abs()
    return dict(max_rel=float(err.max()) / mx, beyond_1e5=float((err > 1e-5 * mx).float().mean()),
                beyond_1e4=float((err > 1e-4 * mx).float().mean()), n_beyond_1e3=int((err > 1e-3 * mx).sum()),
                loss_rel=max(abs(la[k] - lb[k]) / (abs(lb[k]) + 1e-30) for k in la), equal=bool(torch.equal(ga, gb)))


if __name__ == "__main__":
    base = [one({}) for _ in range(3)]
    print("self", dist(base[0], base[1]), dist(base[2], base[1]))
    for env in ({"STYLEMESH_FUSE_POOL_FWD": "0"}, {"STYLEMESH_FUSE_GRAM_BWD": "0"}, {"STYLEMESH_RESIDENT": "0"},
                {"STYLEMESH_SIDE_STREAMS": "0"}, {"STYLEMESH_RESIDENT": "0", "STYLEMESH_FUSE_POOL_FWD": "0"},
                {"STYLEMESH_RESIDENT": "0", "STYLEMESH_FUSE_GRAM_BWD": "0"}):
        v = [one(env) for _ in range(2)]
        print(env, "self", dist(v[0], v[1]), "cross", dist(v[0], base[1]), dist(v[1], base[0]))
