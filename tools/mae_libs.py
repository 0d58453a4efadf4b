#!/usr/bin/env python
"""Distance to the fp64 oracle for one prebuilt library variant (S2S_HIP_LIB), larger sample than the unit test:
python tools/mae_libs.py [n_reads].  Run once per variant (the library is bound at import)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import seq2squiggle_amd as S
from seq2squiggle_amd import chunker
from oracle import s2s_oracle as O

sd, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"))
eng = S.Engine(sd, cfg, mode=os.environ.get("S2S_MODE", "f16x3"))
rng = np.random.default_rng(3)
reads = ["".join(rng.choice(list("ACGT"), 5000)) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2)]
bases, nv, _ = chunker.encode_reads(reads, 9)
B = bases.shape[0]
codes = np.concatenate([O.encode_read(r, 9) for r in reads])            # [B, 16, 9] k-mer letter codes (oracle form)
g = torch.distributions.Gamma(torch.full((B, 16), 9.0), torch.ones(B, 16)).sample().float()
p = dict(dwell_mean=12.5, dwell_std=0.0, noise_std=0.0, noise_sampling=True, duration_sampling=True, min_noise=0.0, min_duration=3.0)
out = eng.predict_chunks(torch.from_numpy(bases).cuda(), torch.from_numpy(nv).cuda(), S.PredictParams(**p), inject_g=g.cuda())
o64 = O.predict_chunks(sd, cfg, codes, O.PredictParams(**p), inject_g=g, dtype=torch.float64)
y, t = out["signal"].cpu().numpy().astype(np.float64), o64["signal"].numpy()
same = (y == 0) == (t == 0)
d = np.abs(y - t)[same]
print(f"{os.path.basename(os.environ.get('S2S_HIP_LIB', 'default'))}: chunks {B}  MAE {d.mean():.3e}  max {d.max():.3e}  "
      f"zero-pattern {same.mean():.6f}  dwell equal {bool((out['dur'].cpu().numpy() == o64['dur'].numpy()).all())}")
