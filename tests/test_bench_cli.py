"""bench.py's launch contract on a box without GPUs: a GPU count that cannot run is an error, never a line."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          timeout=300, env=e)


def test_more_gpus_than_visible_fails_loudly():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return          # an 8-GPU node really runs it; covered by the driver's scaling run
    r = _run("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert r.returncode != 0
    assert "needs 2 visible GPUs" in r.stderr
    assert "n_gpus" not in r.stdout          # no JSON line claiming a GPU count


def test_world_size_must_match_gpus_flag():
    import torch
    if not torch.cuda.is_available():
        # the visible-GPU check fires first on a CPU box; the message still names the flag
        r = _run("--gpus", "1", env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
        assert r.returncode != 0 and "--gpus 1" in r.stderr
        return
    r = _run("--gpus", "1", "--steps", "2", "--warmup", "1", env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0",
                                                                   "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_host_cores_respects_affinity_and_quota():
    sys.path.insert(0, ROOT)
    import bench
    n = bench.host_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))
