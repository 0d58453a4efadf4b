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
    """The multi-rank plumbing of the line, without a GPU: the ranks meet on a HOST (gloo) group -- the bench needs no RCCL, the path has
    no exchange step --, every rank reports its identity and CPU binding, RCCL is a self-test in throw-away children whose failure (no
    GPU here) is REPORTED and costs nothing, and rank 0 alone runs the cpu_baseline leg while the other waits at the host barrier."""
    p = _run("--gpus", "2", "--launch-selftest")
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["sum_of_ones"] == 2
    assert d["barrier_backend"] == "gloo"
    dev = d["per_rank_device"]
    assert [r["rank"] for r in dev] == [0, 1] and len({r["pid"] for r in dev}) == 2
    if dev[0]["cpus"] is not None:                                   # pinned: the two ranks' CPU sets are disjoint
        from seq2squiggle_amd.placement import parse_cpulist
        assert not set(parse_cpulist(dev[0]["cpus"])) & set(parse_cpulist(dev[1]["cpus"]))
    st = d["rccl_selftest"]
    assert st["n_ranks"] == 2 and st["ok"] is (not st.get("error"))
    if not os.path.exists("/dev/kfd"):
        assert st["ok"] is False and "error" in st                   # no GPU: reported, and the line came out all the same
    cb = d["cpu_baseline"]                                          # N > 1 lines carry the CPU baseline too (rank 0 ran it)
    assert cb["value"] > 0 and cb["unit"] == "samples/s" and cb["kind"] == "port" and cb["cores"] >= 1


def test_a_selftest_child_that_never_answers_is_killed_and_reported():
    """RCCL bring-up with more than one rank has never run on this project's hardware: if it hangs on the first real node, the
    children are killed at the wall limit, the line says so and still carries everything else."""
    import time
    t0 = time.time()
    p = _run("--gpus", "2", "--launch-selftest", "--rccl-selftest-limit", "3", "--no-cpu-baseline", env_extra={"S2S_BENCH_SELFTEST_HANG": "1"})
    assert p.returncode == 0, p.stderr
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_ranks_seen"] == 2 and d["barrier_backend"] == "gloo"
    st = d["rccl_selftest"]
    assert st["ok"] is False and "killed" in st["error"] and 2.5 < st["seconds"] < 30
    assert time.time() - t0 < 120
    # nothing of the self-test is left running
    out = subprocess.run(["ps", "-eo", "pid,args"], capture_output=True, text=True).stdout
    assert not [l for l in out.splitlines() if "--rccl-selftest-child" in l]


def test_eight_ranks_meet_and_report(tmp_path):
    """The driver's largest N through the same plumbing (no GPU): eight ranks on the host group, eight identities, eight disjoint
    CPU sets (when this box has eight CPUs to give), the self-test reported, one line."""
    p = _run("--gpus", "8", "--launch-selftest", "--rccl-selftest-limit", "60", "--no-cpu-baseline")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_ranks_seen"] == 8 and d["sum_of_ones"] == 8 and d["barrier_backend"] == "gloo"
    dev = d["per_rank_device"]
    assert [r["rank"] for r in dev] == list(range(8)) and len({r["pid"] for r in dev}) == 8
    if all(r["cpus"] for r in dev) and len(os.sched_getaffinity(0)) >= 8:
        from seq2squiggle_amd.placement import parse_cpulist
        sets = [set(parse_cpulist(r["cpus"])) for r in dev]
        assert sum(len(x) for x in sets) == len(set().union(*sets))              # pairwise disjoint
    assert d["rccl_selftest"]["n_ranks"] == 8 and "ok" in d["rccl_selftest"]


def test_selftest_limit_zero_skips_it():
    p = _run("--gpus", "2", "--launch-selftest", "--rccl-selftest-limit", "0", "--no-cpu-baseline")
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["rccl_selftest"] is None and d["n_ranks_seen"] == 2


def test_the_deadline_thread_writes_the_line_when_a_leg_does_not_return(tmp_path):
    """bench.Line: armed once the headline fields exist; if a secondary leg hangs, the deadline writes the line and ends the process."""
    script = tmp_path / "d.py"
    script.write_text(
        "import os, sys, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "line = bench.Line(os.dup(1))\n"
        "line.arm({'value': 1.0}, 0.5)\n"
        "time.sleep(60)\n"
        "line.write()\n")
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and d["secondary_legs_timed_out"]["after_seconds"] == 0.5


def test_line_is_written_once(tmp_path, capfd):
    sys.path.insert(0, ROOT)
    import bench
    r, w = os.pipe()
    line = bench.Line(w)
    line.arm({"value": 2.0}, 30)
    line.write()
    line.write({"value": 3.0})
    os.close(w)
    assert json.loads(os.read(r, 1 << 16).decode()) == {"value": 2.0}
    os.close(r)


def test_children_of_a_pinned_rank_start_from_its_original_mask(monkeypatch):
    """A child inherits the affinity of the thread that starts it: the RCCL self-test's children and the `predict --gpus N` leg (which
    places its own ranks) must not be squeezed into rank 0's share of the node."""
    if not hasattr(os, "sched_setaffinity") or len(os.sched_getaffinity(0)) < 2:
        import pytest
        pytest.skip("needs two allowed CPUs")
    sys.path.insert(0, ROOT)
    import bench
    from seq2squiggle_amd.placement import format_cpulist
    before = sorted(os.sched_getaffinity(0))
    try:
        os.sched_setaffinity(0, before[:1])                          # "pinned" to one CPU
        monkeypatch.setattr(bench, "_PIN", {"allowed": format_cpulist(before)})
        with bench.unpinned():
            child = subprocess.run([sys.executable, "-c", "import os; print(sorted(os.sched_getaffinity(0)))"], capture_output=True, text=True).stdout
        assert eval(child) == before
        assert sorted(os.sched_getaffinity(0)) == before[:1]         # ... and this rank is bound again afterwards
        monkeypatch.setattr(bench, "_PIN", None)
        with bench.unpinned():
            assert sorted(os.sched_getaffinity(0)) == before[:1]     # not pinned by the bench: nothing to undo
    finally:
        os.sched_setaffinity(0, before)


def test_rocm_smi_text_is_parsed():
    sys.path.insert(0, ROOT)
    import bench
    text = ("GPU[0]\t\t: mclk clock level: 0: (2000Mhz)\nGPU[0]\t\t: sclk clock level: 1: (2263Mhz)\n"
            "======================================= Power Cap ========================================\n"
            "GPU[0]\t\t: Max Graphics Package Power (W): 1400.0\n"
            "GPU[0]\t\t: Current Socket Graphics Package Power (W): 1348.0\n")
    assert bench.parse_rocm_smi(text) == {"watts": 1348.0, "cap_watts": 1400.0, "sclk_mhz": 2263.0}
    assert bench.parse_rocm_smi("nothing here") == {"watts": None, "cap_watts": None, "sclk_mhz": None}


def test_devices_distinct():
    sys.path.insert(0, ROOT)
    import bench
    a, b = {"pci": "0000:05:00", "uuid": "u0"}, {"pci": "0000:15:00", "uuid": "u1"}
    assert bench.devices_distinct([a, b]) is True
    assert bench.devices_distinct([a, dict(a)]) is False
    assert bench.devices_distinct([a, {}]) is None
    assert bench.devices_distinct([{"pci": None, "uuid": "x"}, {"pci": None, "uuid": "y"}]) is True


def _fake_summary(tmp_path, sha, mops_f16=None):
    c = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 500.0, "GRBM_GUI_ACTIVE": 8.0e6, "SQ_VALU_MFMA_BUSY_CYCLES": 4.0e8, "SQ_ACTIVE_INST_VALU": 1.5e8,
         "SQ_INSTS_MFMA": 1000.0 * 16100, "_launch": {"chunks": 1000}, "_effective_clock_ghz": 2.2}
    if mops_f16 is not None:
        c["SQ_INSTS_VALU_MFMA_MOPS_F16"] = mops_f16
    d = tmp_path / "profiles" / "r99"
    d.mkdir(parents=True, exist_ok=True)
    whole = {"f16x3": {"void s2s_fused_kernel<1, false, false>": c}}
    if sha != "absent":
        whole["_meta"] = {"csrc_sha256": sha, "git_commit": "abc123", "by_mode": {"f16x3": sha}}
    (d / "pmc_summary.json").write_text(json.dumps(whole))


def test_bench_says_whether_its_counters_describe_the_source_it_runs(tmp_path):
    """the profiled counters in the line are constants from a committed summary; the summary carries the hash of the
    kernel sources it was measured on and the line says whether that is the source of the library being benched."""
    sys.path.insert(0, ROOT)
    import bench
    from seq2squiggle_amd import _build
    now = _build.source_hash()
    assert len(now) == 64 and now == _build.source_hash()
    _fake_summary(tmp_path, now)
    x = bench.pmc_counters("f16x3", 2000, root=str(tmp_path))
    assert x["extra"]["pmc_stale"] is False and x["extra"]["pmc_git_commit"] == "abc123" and x["extra"]["csrc_sha256"] == now
    assert x["traffic"] == (2 * 1000 + 500) * 1024 / 1000 * 2000 and x["source"] == os.path.join("profiles", "r99", "pmc_summary.json")
    _fake_summary(tmp_path, "0" * 64)
    assert bench.pmc_counters("f16x3", 2000, root=str(tmp_path))["extra"]["pmc_stale"] is True
    _fake_summary(tmp_path, "absent")                               # a summary from before round 6
    assert bench.pmc_counters("f16x3", 2000, root=str(tmp_path))["extra"]["pmc_stale"] is None
    assert bench.pmc_counters("f16x3", 2000, root=str(tmp_path / "nothing"))["extra"] == {"pmc_stale": None}


def test_source_hash_follows_the_sources(tmp_path, monkeypatch):
    from seq2squiggle_amd import _build
    before = _build.source_hash()
    extra = tmp_path / "x.h"
    extra.write_text("// one more header\n")
    real = _build.deps
    monkeypatch.setattr(_build, "deps", lambda: real() + [str(extra)])
    assert _build.source_hash() != before


def test_issued_mfma_work_per_chunk_from_the_instruction_counters(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    # per chunk: 12,000 wave instructions of 16x16x32 (16,384 FLOP) + 4,100 of 32x32x16 (32,768 FLOP)
    issued = 12000 * 16384 + 4100 * 32768
    _fake_summary(tmp_path, "absent", mops_f16=1000.0 * issued / 512)
    e = bench.pmc_counters("f16x3", 1000, root=str(tmp_path))["extra"]
    assert abs(e["mfma_issued_flop_per_chunk"] - issued) < 1 and abs(e["mfma_useful_frac"] - bench.FLOP_PER_CHUNK / issued) < 1e-12
    assert abs(e["mfma_by_shape_per_chunk"]["16x16x32_f16"] - 12000) < 1e-6 and abs(e["mfma_by_shape_per_chunk"]["32x32x16_f16"] - 4100) < 1e-6
    assert e["mfma_wave_instructions_per_chunk"] == 16100
    _fake_summary(tmp_path, "absent")                               # the counter was not collected: no guess
    e = bench.pmc_counters("f16x3", 1000, root=str(tmp_path))["extra"]
    assert e["mfma_issued_flop_per_chunk"] is None and e["mfma_useful_frac"] is None


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
