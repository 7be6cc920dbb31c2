"""The single-node rank launcher behind ``python bench.py --gpus N`` (stylemesh_amd/launch.py): argument / environment
plan, device check, exit-code logic and output relay - with stand-in rank programs, no GPU."""
import io
import os
import subprocess
import sys
import textwrap

from conftest import REPO
from stylemesh_amd import launch


def test_launch_is_needed_only_for_a_plain_start_with_several_ranks():
    assert launch.needs_launch(2, {})
    assert not launch.needs_launch(1, {})
    assert not launch.needs_launch(8, {"WORLD_SIZE": "8"})       # torch.distributed.run (the driver) already made the ranks


def test_rank_plans_carry_the_rendezvous_environment():
    plans = launch.rank_plans(4, ["bench.py", "--gpus", "4", "--steps", "20"], {"PATH": "/bin", "FOO": "1"}, 29511, python="py")
    assert len(plans) == 4
    for r, (cmd, env) in enumerate(plans):
        assert cmd == ["py", "bench.py", "--gpus", "4", "--steps", "20"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"]) == (str(r), str(r), "4")
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29511" and env["FOO"] == "1"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    kept = launch.rank_plans(1, ["x"], {"HSA_ENABLE_IPC_MODE_LEGACY": "1"}, 1)[0][1]
    assert kept["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"             # an explicit setting is the caller's


def test_device_check_refuses_too_few_gpus_with_one_message():
    assert launch.check_devices(2, 8, "nccl") is None
    msg = launch.check_devices(8, 1, "nccl")
    assert "8" in msg and "1 GPU" in msg
    assert launch.check_devices(2, 1, "gloo") is None            # functional mode: ranks share the device
    assert launch.check_devices(2, 0, "gloo") is not None
    err = io.StringIO()
    rc = launch.launch(8, ["never-started"], env={}, device_count=lambda: 1, err=err)
    assert rc == launch.RC_NO_DEVICES and err.getvalue().count("\n") == 1


def test_worst_rc():
    assert launch.worst_rc([0, 0, 0]) == 0
    assert launch.worst_rc([0, 3, 1]) == 3
    assert launch.worst_rc([0, -9]) == 137                       # killed by SIGKILL


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_run_ranks_relays_rank0_stdout_and_reports_success(tmp_path):
    prog = _script(tmp_path, """
        import os, sys
        r = os.environ["RANK"]
        print('{"rank": %s, "world": %s}' % (r, os.environ["WORLD_SIZE"]))
        print("noise from rank " + r, file=sys.stderr)
    """)
    out, err = io.StringIO(), io.StringIO()
    rc = launch.run_ranks(launch.rank_plans(3, [prog], dict(os.environ), launch.free_port()), out=out, err=err)
    assert rc == 0
    assert out.getvalue() == '{"rank": 0, "world": 3}\n'         # ONE line on stdout: rank 0's
    assert '[rank 1] {"rank": 1, "world": 3}' in err.getvalue() and "[rank 2] noise from rank 2" in err.getvalue()


def test_run_ranks_ends_the_survivors_when_a_rank_fails(tmp_path):
    prog = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)          # a rank stuck in a collective whose peer has died
    """)
    import time
    t0 = time.monotonic()
    rc = launch.run_ranks(launch.rank_plans(2, [prog], dict(os.environ), launch.free_port()), out=io.StringIO(),
                          err=io.StringIO(), grace_s=0.5)
    assert time.monotonic() - t0 < 30
    assert rc != 0


def test_run_ranks_timeout(tmp_path):
    prog = _script(tmp_path, "import time; time.sleep(600)\n")
    rc = launch.run_ranks(launch.rank_plans(2, [prog], dict(os.environ), launch.free_port()), timeout_s=1.0,
                          out=io.StringIO(), err=io.StringIO(), grace_s=0.5)
    assert rc == launch.RC_TIMEOUT


def test_bench_refuses_more_ranks_than_gpus_before_touching_a_gpu():
    """`python bench.py --gpus 8` on a box without 8 GPUs: one message, exit code RC_NO_DEVICES, no rank started."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "STYLEMESH_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() >= 8:
        return
    assert r.returncode == launch.RC_NO_DEVICES, (r.returncode, r.stderr[-400:])
    assert "one rank per GPU" in r.stderr and r.stdout == ""


def test_gpus_are_counted_from_the_kfd_topology_without_the_runtime(tmp_path):
    """The launcher's parent counts devices from sysfs (``torch.cuda.device_count()`` may open /dev/kfd): KFD
    nodes with SIMDs whose render node this process was given, narrowed by the visibility variables; no KFD = 0 GPUs; a
    malformed topology = no check."""
    topo, dri = tmp_path / "nodes", tmp_path / "dri"
    dri.mkdir()
    for i, (simd, minor) in enumerate(((0, 0), (0, 0), (1024, 128), (1024, 136), (1024, 144))):      # two CPU nodes, three GPUs
        d = topo / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\ndrm_render_minor {minor}\n")
        if simd:
            (dri / f"renderD{minor}").write_text("")
    assert launch.kfd_gpu_count(str(topo), env={}, dri=str(dri)) == 3
    assert launch.kfd_gpu_count(str(topo), env={"HIP_VISIBLE_DEVICES": "0,2"}, dri=str(dri)) == 2
    assert launch.kfd_gpu_count(str(topo), env={"ROCR_VISIBLE_DEVICES": ""}, dri=str(dri)) == 0
    #  an index the container does not have names no device; CUDA_VISIBLE_DEVICES is HIP's alias, read only when
    # HIP_VISIBLE_DEVICES is unset; ROCR narrows first; a UUID list cannot be judged from sysfs
    assert launch.kfd_gpu_count(str(topo), env={"HIP_VISIBLE_DEVICES": "5"}, dri=str(dri)) == 0
    assert launch.kfd_gpu_count(str(topo), env={"HIP_VISIBLE_DEVICES": "0,1", "CUDA_VISIBLE_DEVICES": "0"}, dri=str(dri)) == 2
    assert launch.kfd_gpu_count(str(topo), env={"CUDA_VISIBLE_DEVICES": "1"}, dri=str(dri)) == 1
    assert launch.kfd_gpu_count(str(topo), env={"ROCR_VISIBLE_DEVICES": "0,1", "HIP_VISIBLE_DEVICES": "0,1,2"}, dri=str(dri)) == 2
    assert launch.kfd_gpu_count(str(topo), env={"HIP_VISIBLE_DEVICES": "GPU-abcdef"}, dri=str(dri)) is None
    assert launch.kfd_gpu_count(str(tmp_path / "absent"), env={}, dri=str(dri)) == 0
    (dri / "renderD136").unlink()                     # a container that was given two of the three render nodes
    assert launch.kfd_gpu_count(str(topo), env={}, dri=str(dri)) == 2
    if os.geteuid() != 0:                             # (root reads everything: the permission case cannot be staged)
        os.chmod(topo / "4" / "properties", 0)
        assert launch.kfd_gpu_count(str(topo), env={}, dri=str(dri)) == 1
        os.chmod(topo / "4" / "properties", 0o644)
    (topo / "5").mkdir()
    (topo / "5" / "properties").write_text("simd_count not-a-number\n")
    assert launch.kfd_gpu_count(str(topo), env={}, dri=str(dri)) is None


def test_quad_list_checker_accepts_quads_and_rejects_other_lists():
    """ops.check_quad_list = the documented preconditions of SM_LIST_QUADS, host side."""
    import numpy as np
    import pytest
    import torch
    from stylemesh_amd.runtime import hip, ops
    H, W = 10, 70
    Wp = hip.row_stride(W)
    PAD = 0xFFFFFF

    def quad(g, y, x, live=4):
        q = (y + 1) * Wp + x + 1
        return [(g << 24) | (q + i * Wp if i < live else PAD) for i in range(4)]
    good = torch.tensor(quad(0, 0, 0) + quad(0, 4, 32) + quad(0, 8, 2, live=2) + quad(1, 0, 6), dtype=torch.int32)
    ops.check_quad_list(good, [(H, W), (H, W)])
    ops.check_quad_list(good, [(H, W), (H, W)], unpool=True)
    with pytest.raises(ValueError, match="multiple of four"):
        ops.check_quad_list(good[:-1], [(H, W), (H, W)])
    with pytest.raises(ValueError, match="several problems"):
        bad = good.clone(); bad[1] = (1 << 24) | (int(bad[1]) & PAD); ops.check_quad_list(bad, [(H, W), (H, W)])
    with pytest.raises(ValueError, match="first entry is padding"):
        bad = good.clone(); bad[0] = PAD; ops.check_quad_list(bad, [(H, W), (H, W)])
    with pytest.raises(ValueError, match="vertically adjacent"):
        bad = good.clone(); bad[2] = int(bad[2]) + 4; ops.check_quad_list(bad, [(H, W), (H, W)])
    with pytest.raises(ValueError, match="even columns"):
        ops.check_quad_list(torch.tensor(quad(0, 4, 3), dtype=torch.int32), [(H, W)], unpool=True)
    with pytest.raises(ValueError, match="even columns"):
        ops.check_quad_list(torch.tensor(quad(0, 2, 4), dtype=torch.int32), [(H, W)], unpool=True)
    with pytest.raises(ValueError, match="outside the image rows"):
        ops.check_quad_list(torch.tensor(quad(0, H, 0), dtype=torch.int32), [(H, W)])
