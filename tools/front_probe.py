#!/usr/bin/env python
"""Timing probe for the frontend phase (GPU box): run with S2S_HIP_LIB pointing at a -DS2S_ABL=65536 build (decoder loop skipped)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import seq2squiggle_amd as S
sd, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"))
rng = np.random.default_rng(0)
reads = ["".join(rng.choice(list("ACGT"), 5000)) for _ in range(1000)]
bases, nv, _ = S.encode_reads(reads, 9)
b, n = torch.from_numpy(bases).cuda(), torch.from_numpy(nv).cuda()
sig = torch.empty(b.shape[0], 250, device="cuda"); dur = torch.empty(b.shape[0], 16, dtype=torch.int32, device="cuda")
for name, sdv, kw in (("default", sd, {}), ("no duration sampling", sd, dict(duration_sampling=False)),
                      ("noise off too", sd, dict(duration_sampling=False, noise_std=0.0))):
    eng = S.Engine(sdv, cfg, mode="f16x3")
    p = S.PredictParams(seed=1, **kw)
    eng.predict_chunks(b, n, p, out_signal=sig, out_dur=dur)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): eng.predict_chunks(b, n, p, out_signal=sig, out_dur=dur)
    torch.cuda.synchronize(); print(f"{name:28s} {(time.perf_counter() - t0) / 3 * 1e3:8.3f} ms per {b.shape[0]} chunks")
    eng.close()
