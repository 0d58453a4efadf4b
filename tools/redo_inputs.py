#!/usr/bin/env python
"""Redo share of the fast softmax path per INPUT family and checkpoint (GPU): the kernel's live counter (s2s_stats_read) beside the
checkpoint's calibration launch (512 pseudo-random chunks) and the CPU model (oracle/redo_model.py).  The share is a property of
the weights AND the reads; tests/test_gpu_parity.py::test_redo_share_depends_on_the_input_and_parity_holds holds parity on the same
families.      python tools/redo_inputs.py [reads per family = 8] [read length = 640]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import seq2squiggle_amd as S
from oracle import redo_model as R, s2s_oracle as O

torch.set_float32_matmul_precision("highest")
n_reads, read_len = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), (int(sys.argv[2]) if len(sys.argv) > 2 else 640)
QK = ("w_qs.weight", "w_ks.weight", "w_qs.bias", "w_ks.bias")


def variant(sd, kind):
    out = {k: v.clone() for k, v in sd.items()}
    for k in out:
        if k.startswith("decoders.") and k.endswith(QK):
            if kind == "x2":
                out[k] = out[k] * 2.0
            elif kind.startswith("pos"):
                out[k] = float(kind[3:]) * torch.eye(64) if k.endswith("weight") else torch.zeros(64)
    return out


fam = R.input_families(seed=5, n_reads=n_reads, read_len=read_len)
print(f"{n_reads} reads x {read_len} nt per family; columns: live redo share on the fast path / CPU model   (calibration launch first)")
for tag in ("k9", "k6"):
    sd0, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", f"synthetic_{tag}.ckpt"))
    for kind in ("committed", "x2", "pos2", "pos3"):
        sd = sd0 if kind == "committed" else variant(sd0, kind)
        eng = S.Engine(sd, cfg, mode="f16x3")
        calib, chosen = eng.calibration_redo_rate, eng.attention_path
        eng.attention_path = "fast"
        cells = []
        for name, reads in fam.items():
            bases, nv, _ = S.encode_reads(reads, cfg["seq_kmer"])
            codes = np.concatenate([O.encode_read(r, cfg["seq_kmer"]) for r in reads], 0)
            gi = torch.rand(bases.shape[0], 16, generator=torch.Generator().manual_seed(11)) * 20
            eng.stats()
            eng.predict_chunks(torch.from_numpy(bases).cuda(), torch.from_numpy(nv).cuda(), S.PredictParams(noise_std=0.0), inject_g=gi.cuda())
            live = eng.stats()["redo_rate"]
            cells.append(f"{name} {live:.4f}/{R.predicted_redo_rate(sd, cfg, codes, gi):.4f}")
        print(f"{tag} {kind:9s} calibration {calib:.4f} -> {chosen:5s} | " + " | ".join(cells), flush=True)
        eng.close()
