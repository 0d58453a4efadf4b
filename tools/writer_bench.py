#!/usr/bin/env python
"""Host-only timing of the BLOW5 writer on one super-batch of 110 reads x 57,000 samples: record building, compression +
file write, on /tmp and on /dev/shm (tmpfs).  python tools/writer_bench.py [record_compression] [signal_compression]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from seq2squiggle_amd import signal_io, utils as U

prof = U.get_profile("dna-r10-prom")
rng = np.random.default_rng(0)
n = 110
sigs = [rng.integers(300, 700, 57000).astype(np.int16) for _ in range(n)]
dac = np.concatenate(sigs)
offs = np.arange(0, (n + 1) * 57000, 57000)
rc = sys.argv[1] if len(sys.argv) > 1 else None
sc = sys.argv[2] if len(sys.argv) > 2 else None
print("cpus", len(os.sched_getaffinity(0)))
for path in ("/tmp/wb.blow5", "/dev/shm/wb.blow5"):
    if os.path.exists(path):
        os.remove(path)
    w = signal_io.BLOW5Writer(path, prof, False, "dna-r10-prom", False, record_compression=rc, signal_compression=sc)
    for it in range(4):
        t0 = time.perf_counter()
        recs = w.dac_records([f"r{i}" for i in range(n)], dac, offs)
        t1 = time.perf_counter()
        w.write_records(recs)
        t2 = time.perf_counter()
        print(f"{path}: records {(t1 - t0) * 1e3:.1f} ms, compress+write {(t2 - t1) * 1e3:.1f} ms")
    print("file MB", os.path.getsize(path) / 1e6)
    os.remove(path)
r = recs[0]
t0 = time.perf_counter()
for _ in range(20):
    b = w._blow5_record(r)
print(f"one record on one thread: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms, 114000 -> {len(b)} bytes")
