"""bench.py's launcher (CPU): `python bench.py --gpus N` outside torchrun must start its N ranks as child processes,
relay exactly one JSON line and propagate the exit code -- the way the driver calls it for the scaling runs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, env=env, timeout=300)


def test_dry_launch_prints_the_child_command():
    p = _run("--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-launch")
    assert p.returncode == 0, p.stderr
    cmd = json.loads(p.stdout)["dry_launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "8", "--steps", "20", "--warmup", "5"]          # the children get the same arguments


def test_two_ranks_are_started_and_one_line_comes_back():
    p = _run("--gpus", "2", "--launch-selftest")
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["sum_of_ones"] == 2


def test_child_failure_propagates():
    # ranks cannot select a GPU in the CPU container: the launcher must come back non-zero, not hang or print a line
    if os.path.exists("/dev/kfd"):
        import pytest
        pytest.skip("a GPU box would run the bench for real")
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_mismatched_world_size_is_refused():
    p = _run("--gpus", "4", env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


_SHARDED_WORKER = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
import bench
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
out = bench.end_to_end_sharded("f16x3", dist, rank, world, "cpu")
if rank == 0:
    print("SHARDED " + json.dumps(out))
else:
    assert out is None
dist.destroy_process_group()
"""


def test_sharded_leg_reports_a_failing_rank_instead_of_hanging(tmp_path):
    """The N > 1 end-to-end leg must never cost the headline line: here (no GPU) every rank's inference_run raises; both ranks
    still meet at the leg's barriers, rank 0 gets the errors of all ranks, nothing hangs and nothing is left on disk."""
    if os.path.exists("/dev/kfd"):
        import pytest
        pytest.skip("a GPU box would run the leg for real (tests/test_gpu_end_to_end.py does)")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "w.py"
    script.write_text(_SHARDED_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["TMPDIR"] = str(tmp_path)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script), ROOT], capture_output=True, text=True, env=env, timeout=300)
    line = [l for l in p.stdout.splitlines() if l.startswith("SHARDED ")]
    assert p.returncode == 0 and len(line) == 1, p.stdout[-2000:] + p.stderr[-2000:]
    out = json.loads(line[0][8:])
    assert sorted(e.split(":")[0] for e in out["error"]) == ["rank 0", "rank 1"]
