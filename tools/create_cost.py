#!/usr/bin/env python
"""What s2s_create costs, with and without the attention-path calibration launch (GPU box): median of 20 creations each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import seq2squiggle_amd as S
sd, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"))
torch.cuda.init(); torch.zeros(1, device="cuda")
for env in (None, "fast"):
    if env: os.environ["S2S_ATTENTION_PATH"] = env
    else: os.environ.pop("S2S_ATTENTION_PATH", None)
    ts = []
    for i in range(22):
        t0 = time.perf_counter()
        e = S.Engine(sd, cfg, mode="f16x3")
        ts.append(time.perf_counter() - t0)
        e.close()
    print(f"S2S_ATTENTION_PATH={env}: Engine() median {1e3 * float(np.median(ts[2:])):.2f} ms (min {1e3 * min(ts[2:]):.2f}, first {1e3 * ts[0]:.1f})")
