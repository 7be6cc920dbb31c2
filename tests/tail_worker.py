"""Child process of tests/test_round6_gpu.py: a set of fp16x2 conv launches whose grids are K-split tails (fewer tiles than
the chip has block slots), run under the environment it was started with (SM_CONV_TAIL_PASS=1: the second-pass reduction of
rounds 2-5; unset: the in-kernel reduction of csrc/conv_tail.h), outputs saved for a bit-for-bit comparison.
Reference operators: nn.Conv2d + F.relu + MaxPool2d forward / backward, content_and_style_losses.py:11-32,49-69."""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]

from stylemesh_amd.runtime import hip, ops  # noqa: E402
from stylemesh_amd.runtime.fmap import FMap  # noqa: E402


def main(out_path):
    ops.CONV_MODE = "split2"
    out = {}
    # forward (bias + ReLU) and data gradients (plain / gated / gated + addend), one and two levels, with and without lists
    for n, (cin, cout, hws) in enumerate([(128, 128, [(60, 70)]), (512, 512, [(33, 45)]), (256, 64, [(40, 52)]),
                                          (512, 512, [(16, 21), (9, 12)]), (256, 256, [(64, 85)]), (64, 128, [(128, 170)])]):
        torch.manual_seed(100 + n)
        wgt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
        b = (torch.randn(cout) * 0.3).cuda()
        w = ops.pack_conv_fwd(wgt).cuda()
        w2 = ops.pack_conv_split2(w)
        xs = [F.relu(torch.randn(cin, H, W) * 2) for (H, W) in hws]
        ins = [FMap(cin, H, W).from_dense(x.cuda()) for x, (H, W) in zip(xs, hws)]
        amax_in = ops.new_amax("cuda", max(float(x.abs().max()) for x in xs))
        for flags, name in ((hip.EPI_BIAS_RELU, "fwd"), (0, "plain"), (hip.EPI_RELU_MASK, "gated"),
                            (hip.EPI_RELU_MASK | hip.EPI_ADD, "gated_add")):
            gates = [FMap(cout, H, W).from_dense(torch.randn(cout, H, W).cuda()) for (H, W) in hws]
            for rep in range(3):       # the counters must be back at zero after every launch
                outs = [FMap(cout, H, W) for (H, W) in hws]
                if flags & hip.EPI_ADD:
                    for o, (H, W) in zip(outs, hws):
                        o.from_dense(torch.full((cout, H, W), 0.25).cuda())
                am = ops.new_amax("cuda")
                ops.conv3x3_grouped([(i, o, g if flags & hip.EPI_RELU_MASK else None) for i, o, g in zip(ins, outs, gates)],
                                    w, b if flags & hip.EPI_BIAS_RELU else None, flags, None, 1.0, w2, amax_in, am)
                key = f"{n}:{name}"
                res = [o.planes.clone().cpu() for o in outs] + [am.max().cpu()]
                if rep == 0:
                    out[key] = res
                else:
                    assert all(torch.equal(a, b_) for a, b_ in zip(out[key], res)), (key, rep)
    # the pooling epilogue on tail tiles (segment-pair lists)
    from test_round3_gpu import _pair_list
    for n, (C, hws) in enumerate([(128, [(40, 53), (33, 47)]), (256, [(21, 30)]), (512, [(12, 17), (16, 21)])]):
        torch.manual_seed(200 + n)
        wgt = torch.randn(C, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
        b = (torch.randn(C) * 0.3).cuda()
        w = ops.pack_conv_fwd(wgt).cuda()
        w2 = ops.pack_conv_split2(w)
        _, group = ops.conv_list_format(C, C)
        xs = [F.relu(torch.randn(C, H, W) * 2) for (H, W) in hws]
        needs = [torch.ones(H // 2, W // 2).cuda() for (H, W) in hws]
        lst, _ = _pair_list(ops, hip, needs, hws, group)
        ins = [FMap(C, H, W).from_dense(x.cuda()) for x, (H, W) in zip(xs, hws)]
        amax_in = ops.new_amax("cuda", max(float(x.abs().max()) for x in xs))
        outs = [FMap(C, H, W) for (H, W) in hws]
        pooled = [FMap(C, H // 2, W // 2) for (H, W) in hws]
        codes = [torch.zeros(C // 8 * p.plane, dtype=torch.int32, device="cuda") for p in pooled]
        am = ops.new_amax("cuda")
        ops.conv3x3_grouped([(i, o, None, None, p, c) for i, o, p, c in zip(ins, outs, pooled, codes)], w, b,
                            hip.EPI_BIAS_RELU | hip.EPI_POOL, lst, 1.0, w2, amax_in, am)
        out[f"pool{n}"] = [p.planes.clone().cpu() for p in pooled] + [c.clone().cpu() for c in codes] + [am.max().cpu()]
    torch.cuda.synchronize()
    ws = ops.splitk_workspace(torch.device("cuda"))
    out["counters_zero"] = bool((ws[-1024:].view(torch.int32) == 0).all())
    out["tail_pass_env"] = os.environ.get("SM_CONV_TAIL_PASS", "")
    torch.save(out, out_path)


if __name__ == "__main__":
    main(sys.argv[1])
