#!/usr/bin/env python
"""Determinism soak (GPU box): the same 312,000 chunks through the fused predict kernel back to back for SECONDS, every launch's
signal and dwell output compared bit for bit with the first launch's (and, every 16th launch, a 65,520-chunk slice through a
second call with another grouping of chunks onto workgroups).  A race in the kernel's own synchronisation (hand-off slots, wave
progress words, LDS staging) would show as a mismatch.   python tools/soak.py [seconds] [mode]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import seq2squiggle_amd as S

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
mode = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
sd, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"))
eng = S.Engine(sd, cfg, device=0, mode=mode)
bases, nv, _ = S.encode_reads(bench.make_reads(1000, 1234), cfg["seq_kmer"])
b, n = torch.from_numpy(bases).to(eng.device), torch.from_numpy(nv).to(eng.device)
B = b.shape[0]
p = S.PredictParams(seed=42)
ref_sig = torch.empty(B, 250, dtype=torch.float32, device=eng.device); ref_dur = torch.empty(B, 16, dtype=torch.int32, device=eng.device)
eng.predict_chunks(b, n, p, out_signal=ref_sig, out_dur=ref_dur)
sig, dur = torch.empty_like(ref_sig), torch.empty_like(ref_dur)
lo, hi = 100_003, 100_003 + 65_520
s2, d2 = torch.empty(hi - lo, 250, dtype=torch.float32, device=eng.device), torch.empty(hi - lo, 16, dtype=torch.int32, device=eng.device)
bad = launches = slices = 0
t0 = t_log = time.perf_counter()
while time.perf_counter() - t0 < seconds:
    sig.zero_(); dur.zero_()
    eng.predict_chunks(b, n, p, out_signal=sig, out_dur=dur)
    launches += 1
    ok = torch.equal(sig, ref_sig) and torch.equal(dur, ref_dur)
    if launches % 16 == 0:
        eng.predict_chunks(b[lo:hi].contiguous(), n[lo:hi].contiguous(), p, first_global_chunk=lo, out_signal=s2, out_dur=d2)
        slices += 1
        ok = ok and torch.equal(s2, ref_sig[lo:hi]) and torch.equal(d2, ref_dur[lo:hi])
    if not ok:
        bad += 1
        print(f"MISMATCH at launch {launches}: {(sig != ref_sig).sum().item()} samples, {(dur != ref_dur).sum().item()} dwells differ", flush=True)
    if time.perf_counter() - t_log > 60:
        t_log = time.perf_counter()
        print(f"  {launches} launches, {slices} slices, {bad} mismatches after {t_log - t0:.0f} s", flush=True)
el = time.perf_counter() - t0
print(f"soak {mode}: {launches} launches of {B} chunks + {slices} offset slices in {el:.0f} s ({launches * B / el:.3e} chunks/s incl. the compares), "
      f"{bad} mismatches")
sys.exit(1 if bad else 0)
