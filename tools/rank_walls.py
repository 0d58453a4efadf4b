#!/usr/bin/env python
"""BASELINE configs[4] as 8 ranks would run it, one rank at a time on the one GPU of a box: the full synthetic 100 Mb reference
(10 contigs x 10 Mb, rng 1234), `-c 30 -r 10000` -> 300,000 reads, rank R of 8 (RANK / WORLD_SIZE as torchrun sets them) writes
its out.rankR.pod5.  Prints each rank's wall time inside inference_run (reference parse, sampler replay, kernels, export, codec,
file) -- the largest is what an 8-GPU node's wall would be when the ranks do not disturb each other.
python tools/rank_walls.py [ranks, default 0,3,7]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
from seq2squiggle_amd import inference
from seq2squiggle_amd.cli import set_config
from seq2squiggle_amd.utils import set_seeds
ref, out = sys.argv[1], sys.argv[2]
def run(fasta, o, n=-1, c=30, r=10000):
    set_seeds(42)
    t0 = time.perf_counter()
    m = inference.inference_run(config=set_config(None), saved_weights=os.path.join(%r, "tests", "golden", "synthetic_k9.ckpt"), fasta=fasta,
                                read_input=False, n=n, r=r, c=c, out=o, profile="dna-r10-prom", dwell_mean=None, dwell_std=0.0, noise_std=2.0,
                                noise_sampling=True, duration_sampling=True, distr="expon", predict_batch_size=1024, export_every_n_samples=1000000,
                                sample_rate=None, bps=None, digitisation=None, range_val=None, offset_mean=None, offset_std=None,
                                median_before_mean=None, median_before_std=None, min_noise=0.0, min_duration=3, min_read_len=30,
                                preserve_read_ids=False, seed=42)
    el = time.perf_counter() - t0
    chunks = m.chunks_done - getattr(m, "first_global_chunk", 0)
    m.engine.close()
    return el, chunks
lam = os.path.join(%r, "tests", "golden", "example_lambda_genome.fasta")
w = os.environ.pop("WORLD_SIZE"); rk = os.environ.pop("RANK")
run(lam, out + ".warm.pod5", n=500, c=-1, r=5000)          # the process's one-time costs, single-process
os.environ["WORLD_SIZE"], os.environ["RANK"] = w, rk
el, chunks = run(ref, out)
print("RANKWALL", rk, round(el, 3), chunks, flush=True)
''' % (ROOT, ROOT, ROOT)
ranks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,3,7").split(",")]
sys.path.insert(0, ROOT)
from seq2squiggle_amd.utils import write_synthetic_reference
with tempfile.TemporaryDirectory() as td:
    ref = os.path.join(td, "ref100.fasta")
    write_synthetic_reference(ref, [10_000_000] * 10)
    rows = []
    for r in ranks:
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1")
        p = subprocess.run([sys.executable, "-c", CHILD, ref, os.path.join(td, "out.pod5")], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RANKWALL")]
        if not line:
            print(p.stderr[-2000:]); sys.exit(1)
        _, rk, el, chunks = line[0].split()
        size = os.path.getsize(os.path.join(td, f"out.rank{r}.pod5"))
        rows.append((int(rk), float(el), int(chunks), size))
        print(f"rank {rk} of 8: {float(el):.2f} s for {int(chunks)} chunks = {int(chunks) / float(el):.3e} chunks/s, {size / 1e9:.2f} GB", flush=True)
        for f in os.listdir(td):
            if f.endswith(".pod5"):
                os.remove(os.path.join(td, f))
worst = max(r[1] for r in rows)
print(f"slowest rank {worst:.2f} s -> 300,000 ten-kb reads in {worst:.2f} s on 8 GPUs if the ranks do not disturb each other "
      f"({300000 / worst:.0f} reads/s, {sum(r[2] for r in rows) / len(rows) * 8 / worst:.3e} chunks/s)")
