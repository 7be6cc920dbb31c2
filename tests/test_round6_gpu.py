"""Round 6: the in-kernel reduction of the K-split tail (csrc/conv_tail.h) - the fp16x2 conv kernel's unit that arrives
last at its tile reduces the tile's partial slabs before it exits, instead of a second launch per layer. Reference
operators: nn.Conv2d + F.relu + MaxPool2d and their backward (content_and_style_losses.py:11-32,49-69)."""
import os
import subprocess
import sys
import tempfile

import pytest
import torch

from conftest import REPO
from gpu_util import require_gpu

pytestmark = pytest.mark.gpu


def _run_tail_worker(tmp, name, env_extra):
    env = {k: v for k, v in os.environ.items() if k != "SM_CONV_TAIL_PASS"}
    env.update(env_extra)
    path = os.path.join(tmp, name + ".pt")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "tail_worker.py"), path], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return torch.load(path)


def test_in_kernel_tail_reduction_equals_the_second_pass_bit_for_bit():
    """Forward / plain / gated / gated + addend convs and the pooling epilogue on launches that are all K-split tail, one
    and two levels, with and without segment lists: the in-kernel reduction (default) and the second-pass launch
    (SM_CONV_TAIL_PASS=1, a library-wide switch read once - hence two child processes) give the same outputs, argmax
    codes and recorded bounds to the bit; repeated launches reproduce themselves and leave the arrival counters zero."""
    require_gpu()
    with tempfile.TemporaryDirectory() as tmp:
        fused = _run_tail_worker(tmp, "fused", {})
        twopass = _run_tail_worker(tmp, "twopass", {"SM_CONV_TAIL_PASS": "1"})
    assert fused["tail_pass_env"] == "" and twopass["tail_pass_env"] == "1"
    assert fused["counters_zero"] and twopass["counters_zero"]
    keys = [k for k in fused if k not in ("counters_zero", "tail_pass_env")]
    assert len(keys) >= 27 and set(keys) == set(k for k in twopass if k not in ("counters_zero", "tail_pass_env"))
    for k in keys:
        assert len(fused[k]) == len(twopass[k])
        for a, b in zip(fused[k], twopass[k]):
            assert torch.equal(a, b), k
        assert float(fused[k][0].abs().max()) > 0, k


def test_tail_counters_are_zero_after_replayed_steps():
    """A one-level step is tail all over (every conv layer has fewer tiles than the chip has block slots): after steps that
    were recorded and REPLAYED as step programs (runtime/program.py: the same launches, the same workspace) every arrival
    counter of every split-K workspace is back at zero - the in-kernel reduction cleans up after itself."""
    require_gpu()
    import test_round4_gpu as R4
    from stylemesh_amd.runtime import ops
    c = R4.PROGRAM_CASES["only2D"]
    eng = R4._program_engine(c, "1")
    view = R4._small_views((0,))[0]
    for _ in range(6):
        eng.training_step(view)
    torch.cuda.synchronize()
    assert eng.program_replays >= 2 and len(ops._SPLITK_WS) >= 1
    for ws in ops._SPLITK_WS.values():
        assert bool((ws[-1024:].view(torch.int32) == 0).all())
