"""Where a rank of a one-process-per-GPU run lives ("one process per GPU ... pinned staging"): which device it
may see, which device index it uses, and which CPUs its threads and pinned buffers stay on.

The reference has nothing to mirror here: its predict is unsharded (dataloader.py:448-449) and multi-GPU runs are left to
Lightning's DDP wrapper (inference.py:430-445).  Everything in this module is standard library only and runs BEFORE the process
touches the GPU: the parent of `predict --gpus N` imports it without loading numpy or torch, and a rank binds itself before its
first allocation, so that its host threads, its page faults and the pinned staging buffers land on its GPU's socket.

  * visibility: child r of the launcher gets HIP_VISIBLE_DEVICES = the r-th device the parent may see, so a rank sees exactly ONE
    device and no call site can place anything on a neighbour's GPU (`rank_visibility`);
  * device index: 0 when visibility is narrowed to one device, LOCAL_RANK otherwise -- a user's own torchrun (`local_device`);
  * CPUs: an equal, contiguous, disjoint share of the allowed CPUs of the GPU's NUMA node, read from the KFD topology and the
    PCI device's `local_cpulist` in sysfs (`rank_cpus`, `pin_rank`); falls back to a plain equal split of the allowed CPUs;
    S2S_NO_PIN=1 opts out.
"""
import logging
import os
from typing import Dict, List, Optional, Sequence

logger = logging.getLogger("seq2squiggle")

_VIS = ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")       # (the HIP runtime reads both; HIP_ wins)


def _entries(value: Optional[str]) -> Optional[List[str]]:
    if value is None or value.strip() == "":      # (the HIP runtime reads an empty variable as an unset one)
        return None
    return [x.strip() for x in value.split(",") if x.strip() != ""]


def visible_list(env=None) -> Optional[List[str]]:
    """The HIP-level device list of this environment (entries are indices into what ROCR_VISIBLE_DEVICES leaves), or None."""
    env = os.environ if env is None else env
    for name in _VIS:
        got = _entries(env.get(name))
        if got is not None:
            return got
    return None


def rank_visibility(local_rank: int, env=None) -> Dict[str, str]:
    """Environment of launcher child `local_rank`: exactly one visible device -- the local_rank-th of the parent's list when the
    parent is narrowed itself (HIP_VISIBLE_DEVICES=4,5,6,7 -> rank 1 gets 5), else device `local_rank`.  ROCR_VISIBLE_DEVICES is
    left as it is (HIP indices are relative to it).  S2S_ONE_GPU (rehearsal: every rank on the one device) narrows nothing."""
    env = os.environ if env is None else env
    if env.get("S2S_ONE_GPU"):
        return {}
    parent = visible_list(env)
    if parent is None:
        one = str(local_rank)
    elif local_rank < len(parent):
        one = parent[local_rank]
    else:
        raise ValueError(f"rank {local_rank} has no device: the environment narrows the GPUs to {','.join(parent) or 'none'}")
    return {name: one for name in _VIS}


def local_device(env=None) -> int:
    """Device index of this rank inside ITS process: 0 when the process sees one device only (a child of the launcher, or
    S2S_ONE_GPU), LOCAL_RANK under a user's own torchrun with all devices visible."""
    env = os.environ if env is None else env
    if env.get("S2S_ONE_GPU"):
        return 0
    vis = visible_list(env)
    if vis is not None and len(vis) == 1:
        return 0
    return int(env.get("LOCAL_RANK", "0"))


# ------------------------------------------------------------------------------------------------------------------------------
# topology


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's list format)."""
    out = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            out.extend(range(int(lo), int(hi) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def format_cpulist(cpus: Sequence[int]) -> str:
    cpus = sorted(set(cpus))
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(runs)


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def gpu_nodes(sysfs: str = "/sys", dev: str = "/dev") -> List[dict]:
    """The GPUs in the order the ROCm runtime enumerates them: KFD topology nodes with SIMDs, by node number, kept only when
    their render node can be opened (the runtime skips devices a container's device cgroup hides) -> [{"node", "bdf",
    "numa_node", "cpus", "unique_id"}].  `cpus` is None when sysfs does not say."""
    root = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        ids = sorted(int(x) for x in os.listdir(root) if x.isdigit())
    except OSError:
        return []
    out = []
    for i in ids:
        text = _read(os.path.join(root, str(i), "properties"))
        if text is None:
            continue
        prop = {}
        for line in text.splitlines():
            kv = line.split()
            if len(kv) == 2:
                try:
                    prop[kv[0]] = int(kv[1])
                except ValueError:
                    pass
        if prop.get("simd_count", 0) <= 0:
            continue                                         # a CPU node
        minor = prop.get("drm_render_minor", -1)
        if minor >= 0:
            node = os.path.join(dev, "dri", f"renderD{minor}")
            if os.path.isdir(os.path.join(dev, "dri")) and not os.access(node, os.R_OK | os.W_OK):
                continue
        loc = prop.get("location_id", 0)
        bdf = f"{prop.get('domain', 0):04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
        pci = os.path.join(sysfs, "bus", "pci", "devices", bdf)
        numa = _read(os.path.join(pci, "numa_node"))
        cpus = _read(os.path.join(pci, "local_cpulist"))
        try:
            numa = int(numa) if numa is not None else -1
        except ValueError:
            numa = -1
        try:
            cpus = parse_cpulist(cpus) if cpus is not None and cpus.strip() else None
        except ValueError:
            cpus = None
        if (cpus is None or not cpus) and numa >= 0:          # (no local_cpulist: the NUMA node's own list)
            t = _read(os.path.join(sysfs, "devices", "system", "node", f"node{numa}", "cpulist"))
            try:
                cpus = parse_cpulist(t) if t else None
            except ValueError:
                cpus = None
        out.append({"node": i, "bdf": bdf, "numa_node": numa, "cpus": cpus or None, "unique_id": prop.get("unique_id")})
    return out


def physical_index(local_rank: int, n_gpus: int, env=None) -> Optional[int]:
    """Index of this rank's GPU in gpu_nodes() order, from LOCAL_RANK and the visibility variables; None when the lists are not
    plain indices (UUID entries) or do not reach that far."""
    env = os.environ if env is None else env
    try:
        hip = visible_list(env)
        if hip is None:
            idx = local_rank
        elif len(hip) == 1:
            idx = int(hip[0])
        else:
            idx = int(hip[local_rank])
        rocr = _entries(env.get("ROCR_VISIBLE_DEVICES"))
        if rocr is not None:
            idx = int(rocr[idx])
        return idx if 0 <= idx < n_gpus else None
    except (ValueError, IndexError):
        return None


def _runs(cpus: Sequence[int]) -> List[List[int]]:
    runs = []
    for c in sorted(cpus):
        if runs and c == runs[-1][-1] + 1:
            runs[-1].append(c)
        else:
            runs.append([c])
    return runs


def _split(cpus: Sequence[int], n: int) -> List[List[int]]:
    """n disjoint shares of `cpus`: every maximal run of consecutive CPU numbers is cut into n equal contiguous pieces and share i
    takes the i-th piece of each -- on a socket listed as '0-63,128-191' (SMT siblings 128 apart) a share then holds whole cores,
    not one hardware thread of a neighbour's."""
    shares = [[] for _ in range(n)]
    for run in _runs(cpus):
        for i in range(n):
            shares[i].extend(run[len(run) * i // n: len(run) * (i + 1) // n])
    return shares


def rank_cpus(n_ranks: int, gpus: Sequence[dict], phys: Sequence[Optional[int]], allowed: Sequence[int]) -> List[List[int]]:
    """CPU set of each of the n_ranks local ranks: the ranks whose GPUs hang off one NUMA node divide that node's ALLOWED CPUs
    (sched_getaffinity / the cpuset cgroup) between them; if the topology is unknown for any rank, or any rank would end up
    without a CPU, all ranks fall back to an equal split of the allowed CPUs.  The sets are pairwise disjoint."""
    allowed = sorted(set(allowed))
    plain = _split(allowed, n_ranks)
    if any(not s for s in plain):                            # fewer CPUs than ranks: everybody may run everywhere
        return [list(allowed) for _ in range(n_ranks)]
    groups: Dict[tuple, List[int]] = {}
    for r in range(n_ranks):
        p = phys[r] if r < len(phys) else None
        g = gpus[p] if p is not None and p < len(gpus) else None
        if g is None or not g.get("cpus"):
            return plain
        groups.setdefault(tuple(g["cpus"]), []).append(r)
    out: List[Optional[List[int]]] = [None] * n_ranks
    taken = set()
    for cpus, ranks in groups.items():
        local = [c for c in cpus if c in set(allowed) and c not in taken]
        shares = _split(local, len(ranks))
        if any(not s for s in shares):
            return plain
        for r, s in zip(ranks, shares):
            out[r] = s
            taken.update(s)
    return out                                               # type: ignore[return-value]


def pin_rank(local_rank: Optional[int] = None, local_world: Optional[int] = None, env=None, sysfs: str = "/sys", dev: str = "/dev",
             apply: bool = True) -> Optional[dict]:
    """Binds the calling process (before it has started threads or touched the GPU) to its share of its GPU's socket.
    -> {"cpus": "0-15,128-143", "numa_node", "bdf", "source": "sysfs" | "equal split"} or None (one rank, S2S_NO_PIN, no affinity
    call on this platform).  Never raises: a rank that cannot be pinned runs unpinned."""
    env = os.environ if env is None else env
    try:
        local_rank = int(env.get("LOCAL_RANK", "0")) if local_rank is None else local_rank
        local_world = int(env.get("LOCAL_WORLD_SIZE", env.get("WORLD_SIZE", "1"))) if local_world is None else local_world
        if env.get("S2S_NO_PIN") or local_world <= 1 or not hasattr(os, "sched_getaffinity"):
            return None
        allowed = sorted(os.sched_getaffinity(0))
        gpus = gpu_nodes(sysfs, dev)
        # every local rank computes the whole table from the same inputs, so the shares are disjoint without any exchange; under the
        # launcher (one visible device per child) rank r's device is the r-th of the PARENT's list, which S2S_PARENT_VISIBLE carries
        base = dict(env)
        if env.get("S2S_PARENT_VISIBLE") is not None:
            for name in _VIS:
                base.pop(name, None)
            if env["S2S_PARENT_VISIBLE"] != "":
                base[_VIS[0]] = env["S2S_PARENT_VISIBLE"]
        one_gpu = bool(env.get("S2S_ONE_GPU"))
        vis = visible_list(base)
        if vis is not None and len(vis) == 1 and not one_gpu:
            # narrowed by somebody else's launcher: the neighbours' devices are not known here -> the plain split, which every rank
            # computes alike
            phys = [None] * local_world
        else:
            phys = [physical_index(0 if one_gpu else r, len(gpus), base) for r in range(local_world)]
        sets = rank_cpus(local_world, gpus, phys, allowed)
        mine = sets[local_rank]
        p = phys[local_rank]
        g = gpus[p] if p is not None and p < len(gpus) else {}
        from_sysfs = bool(g.get("cpus")) and set(mine) <= set(g["cpus"])
        info = {"cpus": format_cpulist(mine), "n_cpus": len(mine), "allowed": format_cpulist(allowed), "numa_node": g.get("numa_node") if from_sysfs else None,
                "bdf": g.get("bdf"), "source": "sysfs" if from_sysfs else "equal split"}
        if apply:
            os.sched_setaffinity(0, mine)
            os.environ["S2S_PINNED_CPUS"] = info["cpus"]      # (signal_io.cpu_share: the affinity mask IS this rank's share now)
        logger.debug(f"rank {local_rank}/{local_world}: CPUs {info['cpus']} ({info['source']}"
                     + (f", NUMA node {info['numa_node']} of GPU {info['bdf']}" if from_sysfs else "") + ")")
        return info
    except Exception as e:                                   # (an odd sysfs, a refused affinity call)
        logger.debug(f"rank not pinned: {type(e).__name__}: {e}")
        return None
