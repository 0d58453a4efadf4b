#!/usr/bin/env python
"""How fast can this host fill files?  (What bounds a sharded run's output and its merge into ONE file.)
  own   : N processes, each writing ITS OWN file with write() of 32 MiB blocks     (the rank shard files)
  append: N processes appending 32 MiB blocks to ONE file opened O_APPEND          (one shared output file)
  mapped: N processes storing 32 MiB blocks into THEIR region of ONE file through a shared mapping (no inode lock)
  pwrite: N processes pwrite()-ing 32 MiB blocks into THEIR region of ONE file     (inode lock per call)
  mapped+ / pwrite+: the same two into a file whose pages were allocated BEFOREHAND (posix_fallocate, timed on its own): what a
          join costs if the destination is pre-faulted while the ranks still compute
  anon  : N threads of numpy copies in anonymous memory                           (the memory system, no file system)
python tools/probes/fs_write_probe.py [dir=/dev/shm] [GB per case=4]"""
import os
import sys
import time
import multiprocessing as mp

import numpy as np

d = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
gb = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
BLK = 32 << 20


def writer(path, flags, nblk, barrier, q):
    buf = np.random.default_rng(os.getpid()).integers(0, 255, BLK, dtype=np.uint8).tobytes()
    fd = os.open(path, flags, 0o644)
    barrier.wait()
    t = time.perf_counter()
    for _ in range(nblk):
        os.write(fd, buf)
    os.close(fd)
    q.put(time.perf_counter() - t)


def region_writer(path, kind, idx, nblk, barrier, q):
    import mmap
    buf = np.random.default_rng(os.getpid()).integers(0, 255, BLK, dtype=np.uint8)
    fd = os.open(path, os.O_RDWR)
    base = idx * nblk * BLK
    mm = mmap.mmap(fd, 0) if kind.startswith("mapped") else None
    view = np.frombuffer(mm, np.uint8) if mm is not None else None
    barrier.wait()
    t = time.perf_counter()
    raw = buf.tobytes()
    for b in range(nblk):
        if view is not None:
            view[base + b * BLK: base + (b + 1) * BLK] = buf
        else:
            os.pwrite(fd, raw, base + b * BLK)
    q.put(time.perf_counter() - t)
    del view
    if mm is not None:
        mm.close()
    os.close(fd)


def run_region(kind, n):
    total_blk = int(gb * 1e9 / BLK)
    per = max(1, total_blk // n)
    path = os.path.join(d, f"probe_{os.getpid()}_r")
    with open(path, "wb") as f:
        f.truncate(per * n * BLK)
        if kind.endswith("+"):
            t = time.perf_counter()
            os.posix_fallocate(f.fileno(), 0, per * n * BLK)
            dt = time.perf_counter() - t
            if n == 1:
                print(f"fallocate: {per * n * BLK / 1e9:.2f} GB in {dt:.2f} s = {per * n * BLK / dt / 1e9:.2f} GB/s", flush=True)
    kind = kind.rstrip("+") if False else kind
    barrier, q = mp.Barrier(n + 1), mp.Queue()
    ps = [mp.Process(target=region_writer, args=(path, kind, i, per, barrier, q)) for i in range(n)]
    for p in ps:
        p.start()
    barrier.wait()
    t = time.perf_counter()
    for p in ps:
        p.join()
    dt = time.perf_counter() - t
    size = os.path.getsize(path)
    os.remove(path)
    print(f"{kind:6s} N={n:2d}: {size / 1e9:.2f} GB in {dt:.2f} s = {size / dt / 1e9:.2f} GB/s", flush=True)


def run(kind, n):
    total_blk = int(gb * 1e9 / BLK)
    per = max(1, total_blk // n)
    paths = [os.path.join(d, f"probe_{os.getpid()}_{i if kind == 'own' else 0}") for i in range(n)]
    for p in set(paths):
        open(p, "wb").close()
    barrier, q = mp.Barrier(n + 1), mp.Queue()
    flags = os.O_WRONLY | (os.O_APPEND if kind == "append" else 0)
    ps = [mp.Process(target=writer, args=(paths[i], flags, per, barrier, q)) for i in range(n)]
    for p in ps:
        p.start()
    barrier.wait()
    t = time.perf_counter()
    for p in ps:
        p.join()
    dt = time.perf_counter() - t
    size = sum(os.path.getsize(p) for p in set(paths))
    t2 = time.perf_counter()
    for p in set(paths):
        os.remove(p)
    print(f"{kind:6s} N={n:2d}: {size / 1e9:.2f} GB in {dt:.2f} s = {size / dt / 1e9:.2f} GB/s   (removing them: {time.perf_counter() - t2:.2f} s)", flush=True)


def anon(n):
    from concurrent.futures import ThreadPoolExecutor
    src = [np.ones(256 << 20, np.uint8) for _ in range(n)]
    dst = [np.empty(256 << 20, np.uint8) for _ in range(n)]
    for x in dst:
        x[:] = 0
    t = time.perf_counter()
    with ThreadPoolExecutor(n) as ex:
        list(ex.map(lambda i: [np.copyto(dst[i], src[i]) for _ in range(4)], range(n)))
    dt = time.perf_counter() - t
    print(f"anon   N={n:2d}: {n * 4 * 256 / 1024:.1f} GiB copied in {dt:.2f} s = {n * 4 * (256 << 20) / dt / 1e9:.2f} GB/s", flush=True)


if __name__ == "__main__":
    print("dir", d, "| cpus visible", os.cpu_count(), "| affinity", len(os.sched_getaffinity(0)))
    try:
        print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
    except OSError:
        pass
    for n in (1, 2, 4, 8):
        anon(n)
    for kind in ("own", "append"):
        for n in (1, 2, 4, 8):
            run(kind, n)
    for kind in ("mapped", "pwrite", "mapped+", "pwrite+"):
        for n in (1, 2, 4, 8):
            run_region(kind, n)
