"""bench.py's launch logic (no GPU): --gpus N never reports another number of ranks than ran."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "--gpus 8" in r.stderr and "n_gpus" not in r.stdout


def test_plain_multi_gpu_launch_spawns_the_ranks(monkeypatch):
    """`python bench.py --gpus 4` outside torchrun: the parent starts torch.distributed.run with 4 ranks and returns its
    exit code without importing torch itself."""
    b = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(b.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    was_loaded = "torch" in sys.modules
    try:
        b.main()
        code = None
    except SystemExit as e:
        code = e.code
    assert code == 7
    assert "--nproc-per-node=4" in seen["cmd"] and "torch.distributed.run" in seen["cmd"]
    assert seen["cmd"][-4:] == ["--gpus", "4", "--steps", "5"] and "127.0.0.1" in seen["cmd"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert ("torch" in sys.modules) == was_loaded
