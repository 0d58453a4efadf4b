#!/usr/bin/env python
"""CPU model of the fast softmax path's redo rate (no GPU; oracle/redo_model.py) for several decoder weights under different
choices of the pass-0 key set.  It reproduces the redo rates the kernel's counters report (x 2 with the first 64 keys: 0.55 model /
0.59 measured; positional 2 I: 0.093 / 0.104) and is what the key order of the K / V^T images (s2s_device_h.h: att32_key_at) was
chosen with."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import redo_model as R
import seq2squiggle_amd as S

torch.set_float32_matmul_precision("highest")
sd0, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"))
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "stages_k9.npz")))
codes, inj = g["codes"][:48], torch.from_numpy(g["g"][:48])
QK = ("w_qs.weight", "w_ks.weight", "w_qs.bias", "w_ks.bias")


def variants():
    for sc in (1.0, 2.0, 4.0, 16.0):
        yield f"committed x{sc:g}", {k: (v * sc if k.startswith("decoders.") and k.endswith(QK) else v.clone()) for k, v in sd0.items()}
    for c in (2.0, 3.0, 5.0):
        sd = {k: v.clone() for k, v in sd0.items()}
        for k in sd:
            if k.startswith("decoders.") and k.endswith(QK):
                sd[k] = c * torch.eye(64) if k.endswith("weight") else torch.zeros(64)
        yield f"positional {c:g} I", sd


for name, sd in variants():
    print(name, {s: f"{R.predicted_redo_rate(sd, cfg, codes, inj, s):.4f}" for s in R.SCHEMES})
