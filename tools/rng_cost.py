#!/usr/bin/env python
"""What the built-in samplers cost: bench workload with the noise / duration samplers switched off at run time."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seq2squiggle_amd as S
import bench as B
sd, cfg = S.load_checkpoint(os.path.join(ROOT, "tests/golden/synthetic_k9.ckpt"))
eng = S.Engine(sd, cfg)
bases, nv, _ = S.encode_reads(B.make_reads(1000, 1234), 9)
b, n = torch.from_numpy(bases).cuda(), torch.from_numpy(nv).cuda()
sig = torch.empty(b.shape[0], 250, device="cuda"); dur = torch.empty(b.shape[0], 16, dtype=torch.int32, device="cuda")
cases = {"default": {}, "noise_std=0": dict(noise_std=0.0), "duration_sampling off": dict(duration_sampling=False, dwell_mean=12.5),
         "both off": dict(noise_std=0.0, duration_sampling=False, dwell_mean=12.5)}
for rep in range(2):
    for name, kw in cases.items():
        pp = S.PredictParams(seed=42, **kw)
        eng.predict_chunks(b, n, pp, out_signal=sig, out_dur=dur); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): eng.predict_chunks(b, n, pp, out_signal=sig, out_dur=dur)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t) / 3
        print(f"{name:24s} {el*1e3:8.2f} ms/step  {b.shape[0]/el/1e6:.3f} M chunks/s")
