#!/usr/bin/env python
"""End-to-end wall time of `predict` (GPU box) for one output format, with a cProfile of the calling thread:
python tools/e2e_stages.py <n_reads> <ext: blow5|pod5|slow5> [read_len] -- lambda genome, default samplers, seed 42.
Worker threads (compression, file writes) are not in the profile; their effect shows as time waiting in result()."""
import cProfile, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seq2squiggle_amd.cli import set_config
from seq2squiggle_amd.inference import inference_run
from seq2squiggle_amd.utils import set_seeds

n, ext = int(sys.argv[1]), sys.argv[2]
r = int(sys.argv[3]) if len(sys.argv) > 3 else 5000


def run(out):
    set_seeds(42)
    m = inference_run(config=set_config(None), saved_weights=os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"),
                      fasta=os.path.join(ROOT, "tests", "golden", "example_lambda_genome.fasta"), read_input=False, n=n, r=r, c=-1,
                      out=out, profile="dna-r10-prom", dwell_mean=None, dwell_std=0.0, noise_std=2.0, noise_sampling=True,
                      duration_sampling=True, distr="expon", predict_batch_size=1024, export_every_n_samples=1000000,
                      sample_rate=None, bps=None, digitisation=None, range_val=None, offset_mean=None, offset_std=None,
                      median_before_mean=None, median_before_std=None, min_noise=0.0, min_duration=3, min_read_len=30,
                      preserve_read_ids=False, seed=42)
    c = m.chunks_done
    m.engine.close()
    return c


with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    run(os.path.join(td, f"warm.{ext}"))
    t0 = time.perf_counter()
    chunks = run(os.path.join(td, f"a.{ext}"))
    el = time.perf_counter() - t0
    size = os.path.getsize(os.path.join(td, f"a.{ext}"))
    print(f"{ext}: {n} reads, {chunks} chunks in {el:.3f} s -> {chunks / el / 1e6:.3f} M chunks/s, {n / el:.0f} reads/s, file {size / 1e6:.1f} MB")
    pr = cProfile.Profile()
    pr.enable()
    run(os.path.join(td, f"b.{ext}"))
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
