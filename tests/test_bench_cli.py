"""bench.py's command line on a CPU-only box: `--gpus N` outside a launcher must launch itself as one rank per GPU (round-3 review: the
driver's SCALE step may well be `python bench.py --gpus 8 ...`), and the pose-file sweep deals scenes to instances by a STABLE hash."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import importlib
    return importlib.import_module("bench")


def test_self_launch_argv_is_the_drivers_command_shape():
    b = _bench()
    argv = b.self_launch_argv(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29517)
    assert argv[0] == sys.executable and argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29517"
    k = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    free = b.self_launch_argv(2, [])                    # a free port is picked when none is given
    assert 1024 < int(free[free.index("--master-port") + 1]) < 65536


def test_gpus_2_outside_a_launcher_relaunches_itself_one_rank_per_gpu():
    """No GPU here, so every rank stops at bench.py's own "needs a GPU" assertion — which it can only reach as a rank of the relaunched
    job (WORLD_SIZE=2): two such failures, not the old "must be launched with torch.distributed.run" exit."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    out = p.stdout + p.stderr
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU box runs the real thing (tests/test_gpu_sharded.py)")
    assert p.returncode != 0
    assert "must be launched with torch.distributed.run" not in out
    assert out.count("bench.py needs a GPU") >= 2, out[-2000:]


def test_scene_instances_are_stable_and_cover_every_scene_once(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "sage-3d_official_amd"))
    from sage_gs import sweep
    ids = [f"{i:04d}" for i in range(1, 41)] + ["0839_b", "scene-α"]
    for i in ids:
        d = tmp_path / "actions" / i
        d.mkdir(parents=True)
        (d / "action_groundtruth.json").write_text('{"groundtruth_data": []}')
    (tmp_path / "actions" / "not_a_scene").mkdir()          # no action file: ignored
    for total in (1, 2, 8):
        shares = [sweep.scenes_of_instance(str(tmp_path / "actions"), k, total) for k in range(total)]
        assert sorted(i for s in shares for i in s) == sorted(ids)                 # every scene exactly once
        assert all(s == sorted(s) for s in shares)
    # known answers (CRC-32, not Python's salted hash): the same in every process, whatever PYTHONHASHSEED says
    assert [sweep.scene_instance(i, 8) for i in ("0001", "0002", "0839", "scene-α")] == [4, 6, 5, 7]
    code = ("import sys; sys.path.insert(0, %r); from sage_gs import sweep; "
            "print([sweep.scene_instance(i, 8) for i in ('0001', '0002', '0839', 'scene-α')])" % os.path.join(ROOT, "sage-3d_official_amd"))
    for seed in ("0", "12345"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONHASHSEED=seed), capture_output=True, text=True, timeout=120)
        assert out.stdout.strip() == "[4, 6, 5, 7]", out.stdout + out.stderr
    with pytest.raises(ValueError):
        sweep.scenes_of_instance(str(tmp_path / "actions"), 8, 8)
    # scene files: plain, nested, compressed
    root = tmp_path / "scenes"; (root / "0002").mkdir(parents=True)
    (root / "0001.ply").write_bytes(b"x"); (root / "0002" / "3dgs_compressed.ply").write_bytes(b"x")
    assert sweep.find_scene_file(str(root), "0001") == (str(root / "0001.ply"), False)
    assert sweep.find_scene_file(str(root), "0002") == (str(root / "0002" / "3dgs_compressed.ply"), True)
    assert sweep.find_scene_file(str(root), "0003") == (None, False)
