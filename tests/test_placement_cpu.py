"""Placement of the ranks of a one-process-per-GPU run (seq2squiggle_amd/placement.py): one visible device per
rank, device index 0 when narrowed, CPU sets that are disjoint, socket-local and inside what the container allows.  No GPU, no
multi-GPU node is needed: the topology is a fake sysfs tree (2 sockets x 4 GPUs, SMT siblings 64 apart)."""
import json
import os
import subprocess
import sys

import pytest

from seq2squiggle_amd import placement as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# bus numbers of an 8-GPU node: four per socket
BUSES = [0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xE5, 0xF5]
SOCKET_CPUS = ["0-31,64-95", "32-63,96-127"]


def fake_node(tmp_path, hidden=(), no_cpulist=False):
    """sysfs + /dev of a 2-socket host: KFD nodes 0, 1 are the CPUs, 2..9 the GPUs (render minors 128..135)."""
    sysfs, dev = tmp_path / "sys", tmp_path / "dev"
    (dev / "dri").mkdir(parents=True)
    nodes = sysfs / "class" / "kfd" / "kfd" / "topology" / "nodes"
    for i in range(2):
        (nodes / str(i)).mkdir(parents=True)
        (nodes / str(i) / "properties").write_text("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\nlocation_id 0\ndomain 0\n")
        nd = sysfs / "devices" / "system" / "node" / f"node{i}"
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(SOCKET_CPUS[i] + "\n")
    for g, bus in enumerate(BUSES):
        d = nodes / str(2 + g)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + g}\n"
                                      f"location_id {bus << 8}\ndomain 0\nunique_id {1000 + g}\n")
        if g not in hidden:
            (dev / "dri" / f"renderD{128 + g}").write_text("")
        pci = sysfs / "bus" / "pci" / "devices" / f"0000:{bus:02x}:00.0"
        pci.mkdir(parents=True)
        (pci / "numa_node").write_text(f"{g // 4}\n")
        if not no_cpulist:
            (pci / "local_cpulist").write_text(SOCKET_CPUS[g // 4] + "\n")
    return str(sysfs), str(dev)


def test_cpulist_round_trip():
    assert P.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert P.format_cpulist([11, 10, 8, 3, 2, 1, 0]) == "0-3,8,10-11"
    assert P.parse_cpulist("") == []


def test_gpu_nodes_come_in_runtime_order_with_their_sockets(tmp_path):
    sysfs, dev = fake_node(tmp_path)
    g = P.gpu_nodes(sysfs, dev)
    assert [x["bdf"] for x in g] == [f"0000:{b:02x}:00.0" for b in BUSES]
    assert [x["numa_node"] for x in g] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert g[0]["cpus"] == P.parse_cpulist(SOCKET_CPUS[0]) and g[7]["cpus"] == P.parse_cpulist(SOCKET_CPUS[1])


def test_hidden_render_nodes_are_skipped_like_the_runtime_does(tmp_path):
    # a container that was given GPUs 4..7 only: the runtime numbers them 0..3
    sysfs, dev = fake_node(tmp_path, hidden=(0, 1, 2, 3))
    g = P.gpu_nodes(sysfs, dev)
    assert [x["numa_node"] for x in g] == [1, 1, 1, 1]


def test_numa_cpulist_is_the_fallback_for_a_missing_local_cpulist(tmp_path):
    sysfs, dev = fake_node(tmp_path, no_cpulist=True)
    g = P.gpu_nodes(sysfs, dev)
    assert g[5]["cpus"] == P.parse_cpulist(SOCKET_CPUS[1])


def _table(tmp_path, n, allowed, env=None, **kw):
    sysfs, dev = fake_node(tmp_path, **kw)
    gpus = P.gpu_nodes(sysfs, dev)
    env = env or {}
    phys = [P.physical_index(r, len(gpus), env) for r in range(n)]
    return P.rank_cpus(n, gpus, phys, allowed), gpus, phys


def test_eight_ranks_get_disjoint_socket_local_shares(tmp_path):
    sets, gpus, phys = _table(tmp_path, 8, range(128))
    assert phys == list(range(8))
    flat = [c for s in sets for c in s]
    assert len(flat) == len(set(flat)) == 128                       # disjoint, nothing left idle
    for r, s in enumerate(sets):
        assert len(s) == 16 and set(s) <= set(gpus[r]["cpus"])      # on the GPU's own socket
        # whole cores: hardware thread c and its sibling c + 64 belong to the same rank
        assert len({c % 64 for c in s}) == 8
    assert P.format_cpulist(sets[0]) == "0-7,64-71" and P.format_cpulist(sets[3]) == "24-31,88-95"
    assert P.format_cpulist(sets[4]) == "32-39,96-103"


def test_a_cpuset_cgroup_is_respected(tmp_path):
    # the container may run on CPUs 8-23 of socket 0 and 40-47 of socket 1 only
    allowed = list(range(8, 24)) + list(range(40, 48))
    sets, gpus, _ = _table(tmp_path, 8, allowed)
    flat = [c for s in sets for c in s]
    assert len(flat) == len(set(flat)) and set(flat) <= set(allowed)
    assert [len(s) for s in sets] == [4, 4, 4, 4, 2, 2, 2, 2]
    for r, s in enumerate(sets):
        assert set(s) <= set(gpus[r]["cpus"])


def test_a_socket_without_allowed_cpus_falls_back_to_the_equal_split(tmp_path):
    allowed = list(range(0, 16))                                     # socket 0 only: ranks 4..7 have no local CPU
    sets, _, _ = _table(tmp_path, 8, allowed)
    assert [len(s) for s in sets] == [2] * 8
    assert sorted(c for s in sets for c in s) == allowed


def test_fewer_cpus_than_ranks_pins_nobody_to_nothing(tmp_path):
    sets, _, _ = _table(tmp_path, 8, [0, 1, 2])
    assert all(s == [0, 1, 2] for s in sets)


def test_unknown_topology_is_the_equal_split():
    sets = P.rank_cpus(4, [], [None] * 4, range(16))
    assert sets == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]


def test_visibility_lists_map_ranks_to_physical_devices(tmp_path):
    env = {"HIP_VISIBLE_DEVICES": "4,5,6,7"}
    sets, gpus, phys = _table(tmp_path, 4, range(128), env=env)
    assert phys == [4, 5, 6, 7]
    assert all(set(s) <= set(P.parse_cpulist(SOCKET_CPUS[1])) and len(s) == 16 for s in sets)
    assert P.physical_index(1, 8, {"ROCR_VISIBLE_DEVICES": "6,7", "HIP_VISIBLE_DEVICES": "1,0"}) == 6
    assert P.physical_index(0, 8, {"HIP_VISIBLE_DEVICES": "GPU-abcdef"}) is None
    assert P.physical_index(3, 2, {}) is None                       # more ranks than devices


def test_rank_visibility_and_local_device():
    assert P.rank_visibility(3, {}) == {"HIP_VISIBLE_DEVICES": "3", "CUDA_VISIBLE_DEVICES": "3"}
    assert P.rank_visibility(1, {"HIP_VISIBLE_DEVICES": "4,5,6,7"})["HIP_VISIBLE_DEVICES"] == "5"
    assert P.rank_visibility(1, {"CUDA_VISIBLE_DEVICES": "2,3"}) == {"HIP_VISIBLE_DEVICES": "3", "CUDA_VISIBLE_DEVICES": "3"}
    assert P.rank_visibility(0, {"HIP_VISIBLE_DEVICES": ""})["HIP_VISIBLE_DEVICES"] == "0"       # empty = unset, as the runtime reads it
    assert P.rank_visibility(2, {"S2S_ONE_GPU": "1"}) == {}
    with pytest.raises(ValueError):
        P.rank_visibility(2, {"HIP_VISIBLE_DEVICES": "0,1"})
    assert P.local_device({"LOCAL_RANK": "5"}) == 5                                               # a user's own torchrun
    assert P.local_device({"LOCAL_RANK": "5", "HIP_VISIBLE_DEVICES": "5"}) == 0                   # a child of the launcher
    assert P.local_device({"LOCAL_RANK": "5", "HIP_VISIBLE_DEVICES": "0,1,2,3,4,5,6,7"}) == 5
    assert P.local_device({"LOCAL_RANK": "2", "S2S_ONE_GPU": "1"}) == 0


def test_pin_rank_binds_a_launcher_child_to_its_gpus_socket(tmp_path, monkeypatch):
    sysfs, dev = fake_node(tmp_path)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)))
    bound = {}
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: bound.setdefault("cpus", sorted(cpus)))
    monkeypatch.delenv("S2S_PINNED_CPUS", raising=False)
    tables = []
    for r in range(8):
        env = {"LOCAL_RANK": str(r), "LOCAL_WORLD_SIZE": "8", "S2S_PARENT_VISIBLE": "", **P.rank_visibility(r, {})}
        info = P.pin_rank(env=env, sysfs=sysfs, dev=dev, apply=False)
        assert info["source"] == "sysfs" and info["numa_node"] == r // 4 and info["bdf"] == f"0000:{BUSES[r]:02x}:00.0"
        tables.append(P.parse_cpulist(info["cpus"]))
    flat = [c for t in tables for c in t]
    assert len(flat) == len(set(flat)) == 128                       # every rank worked out the same table on its own
    # ... and it is applied before anything else runs
    env = {"LOCAL_RANK": "6", "LOCAL_WORLD_SIZE": "8", "S2S_PARENT_VISIBLE": "", **P.rank_visibility(6, {})}
    info = P.pin_rank(env=env, sysfs=sysfs, dev=dev)
    assert bound["cpus"] == tables[6] and os.environ["S2S_PINNED_CPUS"] == info["cpus"]
    monkeypatch.delenv("S2S_PINNED_CPUS")
    # opt-out, single rank: nothing happens
    assert P.pin_rank(env=dict(env, S2S_NO_PIN="1"), sysfs=sysfs, dev=dev) is None
    assert P.pin_rank(env={"LOCAL_RANK": "0", "LOCAL_WORLD_SIZE": "1"}, sysfs=sysfs, dev=dev) is None
    # somebody else's launcher narrowed this rank: the neighbours' devices are unknown -> the equal split, the same on every rank
    info = P.pin_rank(env={"LOCAL_RANK": "1", "LOCAL_WORLD_SIZE": "2", "HIP_VISIBLE_DEVICES": "5"}, sysfs=sysfs, dev=dev, apply=False)
    assert info["source"] == "equal split" and info["n_cpus"] == 64
    # a missing sysfs never raises
    info = P.pin_rank(env={"LOCAL_RANK": "1", "LOCAL_WORLD_SIZE": "2"}, sysfs=str(tmp_path / "nowhere"), dev=dev, apply=False)
    assert info["source"] == "equal split"


def test_cpu_share_counts_a_pinned_mask_once(monkeypatch):
    from seq2squiggle_amd import signal_io
    monkeypatch.delenv("S2S_CPU_SHARE", raising=False)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(128)))
    monkeypatch.delenv("S2S_PINNED_CPUS", raising=False)
    unpinned = signal_io.cpu_share()
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(16)))
    monkeypatch.setenv("S2S_PINNED_CPUS", "0-7,64-71")
    assert signal_io.cpu_share() == unpinned                        # 128 / 8 either way (or the cgroup quota / 8 when that is lower)


def test_launcher_gives_every_rank_one_device(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "HIP_VISIBLE_DEVICES",
                                                            "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "S2S_ONE_GPU")}
    env["S2S_DRY_LAUNCH"] = "1"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")

    def rank_envs(extra):
        p = subprocess.run([sys.executable, "-m", "seq2squiggle_amd", "predict", "x.fa", "-o", str(tmp_path / "o.blow5"), "--gpus", "8"],
                           env=dict(env, **extra), capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr
        return json.loads(p.stdout)["rank_env"]
    envs = rank_envs({})
    assert [e["HIP_VISIBLE_DEVICES"] for e in envs] == [e["CUDA_VISIBLE_DEVICES"] for e in envs] == [str(r) for r in range(8)]
    assert all("ROCR_VISIBLE_DEVICES" not in e and e["LOCAL_WORLD_SIZE"] == "8" and e["S2S_PARENT_VISIBLE"] == "" for e in envs)
    envs = rank_envs({"HIP_VISIBLE_DEVICES": "7,6,5,4,3,2,1,0"})
    assert [e["HIP_VISIBLE_DEVICES"] for e in envs] == [str(7 - r) for r in range(8)]
    assert all(e["S2S_PARENT_VISIBLE"] == "7,6,5,4,3,2,1,0" for e in envs)
    envs = rank_envs({"S2S_ONE_GPU": "1"})                          # the one-GPU rehearsal narrows nothing
    assert all("HIP_VISIBLE_DEVICES" not in e for e in envs)
