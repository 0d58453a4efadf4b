#!/usr/bin/env python
"""Where the end-to-end time of `predict` goes (GPU box): times the stages of inference.run_streaming."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import seq2squiggle_amd as S
from seq2squiggle_amd import utils as U, chunker, signal_io
from seq2squiggle_amd.model import seq2squiggle

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
T = {}
def tic(name, t0): torch.cuda.synchronize(); T[name] = T.get(name, 0) + time.perf_counter() - t0; return time.perf_counter()
t = time.perf_counter()
prof = U.get_profile("dna-r10-prom")
w = signal_io.BLOW5Writer("/tmp/e2e.blow5", prof, False, "dna-r10-prom", False)
if os.path.exists("/tmp/e2e.blow5"): os.remove("/tmp/e2e.blow5")
m = seq2squiggle.load_from_checkpoint(os.path.join(ROOT, "tests/golden/synthetic_k9.ckpt"), out_writer=w, dwell_mean=12.5, noise_std=2.0,
                                      noise_sampling=True, duration_sampling=True, min_noise=0.0, min_duration=3, seed=42)
t = tic("engine create", t)
random.seed(42)
genome, lens = U.preprocess_genome(os.path.join(ROOT, "tests/golden/example_lambda_genome.fasta"))
t = tic("genome load", t)
reads, _ = U.sample_reads_from_reference(genome, lens, n_reads, 5000, -1, {"max_dna_len": 16}, "x", 42)
reads = list(reads)
t = tic("read sampling", t)
dev = m.device
win = torch.arange(24, device=dev)
for s in range(0, len(reads), 110):
    group = reads[s:s + 110]
    flat, chunk_start, n_valid, read_first = chunker.pack_reads([x for x, _ in group], 9)
    t = tic("pack_reads (host)", t)
    flat_d = torch.from_numpy(flat).to(dev)
    cs_d = torch.from_numpy(chunk_start).to(dev)
    nv = torch.from_numpy(n_valid).to(dev); rf = torch.from_numpy(read_first).to(dev)
    t = tic("H2D (packed reads)", t)
    out = m.engine.predict_packed(flat_d, cs_d, nv, m._params(), first_global_chunk=m.chunks_done)
    m.chunks_done += int(cs_d.shape[0])
    t = tic("predict kernels", t)
    ex = m.engine.export_reads(out["signal"], rf, prof["digitisation"], prof["range"], prof["offset_mean"], want_pa=False, want_dac=True)
    t = tic("export kernels", t)
    offs = ex["offsets"].cpu().numpy(); dac = ex["dac"][: int(offs[-1])].cpu().numpy()
    t = tic("D2H", t)
    w.save_dac([n for _, n in group], dac, offs)
    t = tic("writer (records + zlib + file)", t)
tot = sum(T.values())
for k, v in T.items(): print(f"{k:34s} {v * 1e3:9.1f} ms  {100 * v / tot:5.1f} %")
print(f"total {tot:.3f} s for {len(reads)} reads, {m.chunks_done} chunks -> {len(reads) / tot:.0f} reads/s")
