#!/usr/bin/env python
"""Event timeline of one warm end-to-end `predict` run (GPU box): python tools/e2e_timeline.py [n_reads] [ext] [out_dir]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seq2squiggle_amd import inference
from seq2squiggle_amd.cli import set_config
from seq2squiggle_amd.utils import set_seeds

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ext = sys.argv[2] if len(sys.argv) > 2 else "blow5"
out_dir = sys.argv[3] if len(sys.argv) > 3 else None


def run(out):
    set_seeds(42)
    m = inference.inference_run(config=set_config(None), saved_weights=os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"),
                                fasta=os.path.join(ROOT, "tests", "golden", "example_lambda_genome.fasta"), read_input=False, n=n, r=5000,
                                c=-1, out=out, profile="dna-r10-prom", dwell_mean=None, dwell_std=0.0, noise_std=2.0,
                                noise_sampling=True, duration_sampling=True, distr="expon", predict_batch_size=1024,
                                export_every_n_samples=1000000, sample_rate=None, bps=None, digitisation=None, range_val=None,
                                offset_mean=None, offset_std=None, median_before_mean=None, median_before_std=None, min_noise=0.0,
                                min_duration=3, min_read_len=30, preserve_read_ids=False, seed=42)
    m.engine.close()


with tempfile.TemporaryDirectory(dir=out_dir) as td:
    run(os.path.join(td, f"warm.{ext}"))
    run(os.path.join(td, f"warm2.{ext}"))      # (the second call of a process still pays a few one-time costs: pinned sizes)
    inference._TRACE = tr = []
    t0 = time.perf_counter()
    run(os.path.join(td, f"a.{ext}"))
    t1 = time.perf_counter()
    print(f"total {1e3 * (t1 - t0):.1f} ms")
    for ev, t in tr:
        print(f"{1e3 * (t - t0):9.1f} ms  {ev}")
