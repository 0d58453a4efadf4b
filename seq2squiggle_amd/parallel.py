"""Read sharding for multi-GPU predict: one process per GPU, contiguous ranges of
reads balanced by chunk count, no data-path collective.  The global chunk index of a rank's first chunk
keys its RNG counters, so the union of the per-rank outputs is identical to the single-GPU output."""
import os
from typing import List, Sequence, Tuple

import numpy as np

from .chunker import n_chunks


def rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def local_device() -> int:
    """The GPU of this rank as ITS process numbers it: 0 when the process sees one device only -- a child of `predict --gpus N`,
    whose launcher narrows HIP_VISIBLE_DEVICES per rank (placement.rank_visibility) -- else LOCAL_RANK (a user's own torchrun,
    all devices visible).  S2S_ONE_GPU=1 puts every rank on GPU 0: a rehearsal of the multi-process path on a one-GPU box (the
    ranks then share the device; there is no data-path collective to object)."""
    from .placement import local_device as _ld
    return _ld()


def shard_reads(read_lens: Sequence[int], k: int, world: int) -> List[Tuple[int, int, int]]:
    """-> per rank (first_read, end_read, first_global_chunk): contiguous read ranges whose chunk counts are as
    equal as a prefix split allows."""
    chunks = np.array([n_chunks(int(L), k) for L in read_lens], dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(chunks)])
    total = int(cum[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        i = int(np.searchsorted(cum, target, side="left"))
        # choose the read boundary closest to the target
        if i > 0 and abs(cum[i - 1] - target) <= abs(cum[min(i, len(chunks))] - target):
            i -= 1
        bounds.append(max(bounds[-1], min(i, len(chunks))))
    bounds.append(len(chunks))
    return [(bounds[r], bounds[r + 1], int(cum[bounds[r]])) for r in range(world)]


def rank_output_path(out: str, rank: int, world: int) -> str:
    """Per-rank shard file: out.blow5 -> out.rank3.blow5 (single file when world == 1)."""
    if world == 1:
        return out
    base, ext = os.path.splitext(out)
    return f"{base}.rank{rank}{ext}"


def shared_seed(seed: int) -> int:
    """One seed for every rank of a multi-process run.  An explicit seed is shared already; `--seed 0` means "draw a
    fresh one" (utils.py:722-741), which each process would do on its own -- different read sets, shard bounds and RNG
    keys per rank.  Rank 0 draws it and the others receive it over a short-lived gloo group (host side, no GPU)."""
    rank, _, world = rank_world()
    if world == 1 or seed:
        return seed
    import torch.distributed as dist
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo")
    box = [int.from_bytes(os.urandom(4), "big") or 1 if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    if created:
        dist.destroy_process_group()
    return int(box[0])
