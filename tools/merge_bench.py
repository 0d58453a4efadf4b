#!/usr/bin/env python
"""Throughput of the rank-shard merge (seq2squiggle_amd/merge.py, pod5_io.merge_pod5) on this host: N synthetic shard files of
`--gb` GB in all (uncompressed records / signal rows, written by the product's writers) on `--dir`, merged through both engines of
s2s_copy_ranges (map: posix_fallocate + memcpy between shared mappings on 1..T threads; fd: copy_file_range, one writer), plain and
with consume (first shard becomes the output, the others are deleted as they are consumed).  No GPU.     python tools/merge_bench.py [--gb 4] [--shards 8] [--dir /dev/shm]"""
import argparse
import logging
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

logging.getLogger("seq2squiggle").setLevel(logging.ERROR)
from seq2squiggle_amd import parallel, signal_io, utils as U

ap = argparse.ArgumentParser()
ap.add_argument("--gb", type=float, default=4.0)
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--dir", default="/dev/shm")
ap.add_argument("--threads", default="1,4,8,16")
a = ap.parse_args()
prof = U.get_profile("dna-r10-prom")
tmp = tempfile.mkdtemp(dir=a.dir)
rng = np.random.default_rng(6)
LEN = 100000
reads = max(1, int(a.gb * 1e9 / a.shards / (2 * LEN)))
flat = np.tile((600 + 40 * rng.standard_normal(reads * LEN // 10)).astype(np.int16), 10)
offs = np.arange(reads + 1, dtype=np.int64) * LEN
try:
    for ext in ("blow5", "pod5"):
        shards = []
        for r in range(a.shards):
            path = parallel.rank_output_path(os.path.join(tmp, f"o.{ext}"), r, a.shards)
            ids = [f"read{r}_{i}" for i in range(reads)]
            if ext == "pod5":
                os.environ["S2S_POD5_SIGNAL"] = "none"
                w = signal_io.POD5Writer(path, prof, True, "dna-r10-prom", False)
            else:
                w = signal_io.BLOW5Writer(path, prof, True, "dna-r10-prom", False, record_compression="none")
            if r:
                w.start_at(r * reads)
            for lo in range(0, reads, 250):
                hi = min(lo + 250, reads)
                recs = w.dac_records(ids[lo:hi], flat[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo])
                w.write_records(recs)
            if hasattr(w, "close"):
                w.close()
            shards.append(path)
        total = sum(os.path.getsize(p) for p in shards)
        print(f"{ext}: {a.shards} shards, {total / 1e9:.2f} GB on {a.dir}", flush=True)
        out = os.path.join(tmp, "m." + ext)
        for how in ("map", "fd"):
            os.environ["S2S_MERGE_ENGINE"] = how
            for th in ([int(x) for x in a.threads.split(",")] if how == "map" else [1]):
                best = None
                for _ in range(2):
                    if os.path.exists(out):
                        os.remove(out)
                    t = time.perf_counter()
                    n = signal_io.merge_shards(shards, out, threads=th)
                    dt = time.perf_counter() - t
                    best = dt if best is None else min(best, dt)
                print(f"  {how:3s} threads {th:2d}: {n} records, {best:.3f} s, {total / best / 1e9:.2f} GB/s", flush=True)
        os.remove(out)
        os.environ.pop("S2S_MERGE_ENGINE", None)
        th = None
        t = time.perf_counter()
        n = signal_io.merge_shards(shards, out, threads=th, consume=True)
        dt = time.perf_counter() - t
        st = signal_io.merge_shards.last
        print(f"  consume, engine {st.get('engine')}, threads {st.get('threads')}: {n} records, {dt:.3f} s in all (removing the shards beside the copy: {st.get('remove_seconds', 0):.3f} s of unlink), "
              f"{total / dt / 1e9:.2f} GB/s of output; {st['bytes_copied'] / 1e9:.2f} GB moved", flush=True)
        os.remove(out)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
