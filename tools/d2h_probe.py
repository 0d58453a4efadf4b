#!/usr/bin/env python
"""Does a device -> pinned-host copy on a second stream finish while the (register-file-filling, persistent) predict kernel is
running, or only once it has ended?  python tools/d2h_probe.py   (GPU box; tries a few HIP runtime copy settings in child
processes, because the runtime reads them once at start-up)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from seq2squiggle_amd import model as M, engine as S
m = M.seq2squiggle.load_from_checkpoint(checkpoint_path=os.path.join(%r, "tests", "golden", "synthetic_k9.ckpt"), out_writer=None, device=0, mode="f16x3", seed=42)
eng = m.engine
dev = eng.device
B = 131072
rng = np.random.default_rng(0)
bases = torch.from_numpy(rng.choice(np.frombuffer(b"ACGT", np.uint8), (B, 24))).to(dev)
nv = torch.full((B,), 16, dtype=torch.uint8, device=dev)
src = torch.zeros(16 << 20, dtype=torch.uint8, device=dev)
for nbytes in (4096, (9 << 20) + 3, 16 << 20):
    dst = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    side = torch.cuda.Stream(dev)
    for it in range(2):
        eng.predict_chunks(bases[:4096], nv[:4096], S.PredictParams())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.predict_chunks(bases, nv, S.PredictParams())
        k = torch.cuda.Event(); k.record()
        with torch.cuda.stream(side):
            dst.copy_(src[:nbytes], non_blocking=True)
            c = torch.cuda.Event(); c.record(side)
        c.synchronize(); tc = time.perf_counter() - t0
        k.synchronize(); tk = time.perf_counter() - t0
    print(f"  copy of {nbytes:>9} B done after {tc*1e3:7.2f} ms, kernel after {tk*1e3:7.2f} ms", flush=True)
# the same with a destination pinned just now, and with the copy stream told to wait for an (already complete) event first
for what in ("fresh pinned destination", "after wait_event", "fresh + wait_event, odd size"):
    nbytes = (9 << 20) + (3 if "odd" in what else 0)
    side = torch.cuda.Stream(dev)
    eng.predict_chunks(bases[:4096], nv[:4096], S.PredictParams())
    ready = torch.cuda.Event(); ready.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.predict_chunks(bases, nv, S.PredictParams())
    k = torch.cuda.Event(); k.record()
    with torch.cuda.stream(side):
        if "wait" in what:
            side.wait_event(ready)
        d2 = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True) if "fresh" in what else dst
        d2[:nbytes].copy_(src[:nbytes], non_blocking=True) if d2.numel() >= nbytes else None
        c = torch.cuda.Event(); c.record(side)
    c.synchronize(); tc = time.perf_counter() - t0
    k.synchronize(); tk = time.perf_counter() - t0
    print(f"  {what}: copy done after {tc*1e3:7.2f} ms, kernel after {tk*1e3:7.2f} ms", flush=True)
'''
for env in ({}, {"GPU_FORCE_BLIT_COPY_SIZE": "0"}):
    print(env or "default", flush=True)
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, ROOT)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    print(r.stdout + r.stderr[-600:] if r.returncode else r.stdout, flush=True)
