#!/usr/bin/env python
"""Where engine creation time goes: first vs second s2s_create in one process, and the Python-side blob packing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
import seq2squiggle_amd as S
from seq2squiggle_amd import checkpoint, _lib
t = time.perf_counter(); sd, cfg = S.load_checkpoint(os.path.join(ROOT, "tests/golden/synthetic_k9.ckpt")); print(f"load_checkpoint {1e3*(time.perf_counter()-t):.1f} ms")
t = time.perf_counter(); blob = checkpoint.state_dict_to_blob(sd, cfg); print(f"state_dict_to_blob {1e3*(time.perf_counter()-t):.1f} ms")
t = time.perf_counter(); _lib.lib(); print(f"dlopen {1e3*(time.perf_counter()-t):.1f} ms")
for i in range(3):
    t = time.perf_counter(); e = S.Engine(sd, cfg); torch.cuda.synchronize(); print(f"Engine() #{i} {1e3*(time.perf_counter()-t):.1f} ms")
    e.close()
