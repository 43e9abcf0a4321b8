"""CPU-only: `python bench.py --gpus N` without a launcher starts N ranks itself.

The driver launches N > 1 through torch.distributed.run; a plain `python bench.py --gpus N` used to measure ONE GPU and print
n_gpus: 1.  Now bench.py spawns N child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), relays rank 0's line and returns
the worst child status -- before torch is imported or the GPU touched (`--rank-echo` = the children report their environment and
exit; the GPU leg of the same path is tests/test_dp_gpu.py::test_bench_self_launch_two_ranks)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], env=e, capture_output=True, text=True, timeout=120)


def test_self_launch_spawns_n_ranks_and_relays_rank0():
    r = _run("--gpus", "4", "--rank-echo", "ok")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one JSON line (rank 0's) on stdout"
    d = json.loads(lines[0])
    assert d == {"rank": 0, "local_rank": 0, "world": 4, "master": "127.0.0.1", "port": d["port"], "self_launched": True}
    assert 1024 < d["port"] < 65536


def test_self_launch_returns_a_failing_ranks_status():
    r = _run("--gpus", "3", "--rank-echo", "fail")        # the last rank exits 3
    assert r.returncode == 3


def test_external_launcher_is_left_alone():
    """Under torch.distributed.run (RANK set) bench.py must not spawn anything: it IS a rank."""
    r = _run("--gpus", "2", "--rank-echo", "ok", env={"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1",
                                                        "MASTER_PORT": "29511"})
    assert r.returncode == 0
    d = json.loads(r.stdout)
    assert d["rank"] == 1 and d["world"] == 2 and d["self_launched"] is False


def test_single_gpu_default_does_not_spawn():
    r = _run("--rank-echo", "ok")
    assert r.returncode == 0
    d = json.loads(r.stdout)
    assert d["world"] == 1 and d["self_launched"] is False
