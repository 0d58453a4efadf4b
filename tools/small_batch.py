#!/usr/bin/env python
"""Throughput at small batch sizes (GPU box): chunks/s of predict_chunks for B = 256 .. 32768 chunks per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import seq2squiggle_amd as S
sd, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"))
rng = np.random.default_rng(0)
reads = ["".join(rng.choice(list("ACGT"), 5000)) for _ in range(110)]
bases, nv, _ = S.encode_reads(reads, 9)
b, n = torch.from_numpy(bases).cuda(), torch.from_numpy(nv).cuda()
eng = S.Engine(sd, cfg, mode="f16x3")
p = S.PredictParams(seed=1)
for B in (256, 512, 1024, 2048, 4096, 8192, 32768):
    sig = torch.empty(B, 250, device="cuda"); dur = torch.empty(B, 16, dtype=torch.int32, device="cuda")
    bb, nn = b[:B].contiguous(), n[:B].contiguous()
    reps = max(3, 200000 // B)
    eng.predict_chunks(bb, nn, p, out_signal=sig, out_dur=dur)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): eng.predict_chunks(bb, nn, p, out_signal=sig, out_dur=dur)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"B={B:6d}: {B * reps / el / 1e6:.3f} M chunks/s  ({el / reps * 1e6:.0f} us per call)")
