#!/usr/bin/env python
"""Per-tile score-gap histogram of the decoder attention (round 4, review item 1, step 1).

Builds the -DS2S_TILEHIST library and, for several checkpoints, counts for every 32-key x 32-query tile and every 16-key step
of the fast softmax the largest shifted score max(s - m_row) in log2 units, m_row = the row's maximum over its first 64 keys
(what the kernel knows when it reaches the tile).  A step whose largest score sits T units below that maximum has every
P <= 2^-T of a row sum that is >= 1: below T = 14 its P_lo halves contribute < 2^-25 of the row sum each (hi-only step: no
v_fma_mix, no second P.V MFMA), below T = 30 the whole step does (skip: no exp either).  Prints the share of steps that would
classify at those thresholds, per decoder layer.    tools/tile_hist.py [n_reads]
"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "seq2squiggle_amd", "lib", "libs2s_hip_tilehist.so")
from seq2squiggle_amd import _build
_build.compile_to(lib, ["-DS2S_TILEHIST"])
os.environ["S2S_HIP_LIB"] = lib
import numpy as np, torch
import seq2squiggle_amd as S
from seq2squiggle_amd import _lib

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)
reads = ["".join(rng.choice(list("ACGT"), 5000)) for _ in range(n_reads)]


def variants():
    for tag, k in (("k9", 9), ("k6", 6)):
        sd, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", f"synthetic_{tag}.ckpt"))
        yield f"{tag} as committed (w_q, w_k x 3 over the default init)", sd, cfg, k
        if tag == "k9":
            for scale, name in ((1 / 3.0, "default init (x 1)"), (4.0, "x 12 (test_peaked_attention x 4)"), (16.0, "x 48 (test_peaked_attention x 16)")):
                sd2 = {kk: v.clone() for kk, v in sd.items()}
                for kk in sd2:
                    if kk.startswith("decoders.") and kk.endswith(("w_qs.weight", "w_ks.weight", "w_qs.bias", "w_ks.bias")):
                        sd2[kk] *= scale
                yield f"k9 decoder w_q, w_k {name}", sd2, cfg, k


res = []
for name, sd, cfg, k in variants():
    eng = S.Engine(sd, cfg, mode="f16x3")
    bases, nv, _ = S.encode_reads(reads, k)
    b, n = torch.from_numpy(bases).cuda(), torch.from_numpy(nv).cuda()
    out = (C.c_uint64 * 384)()
    _lib.lib().s2s_diag_read(eng._h, out)
    eng.stats()
    eng.predict_chunks(b, n, S.PredictParams(seed=1))
    _lib.lib().s2s_diag_read(eng._h, out)
    st = eng.stats()
    h = np.array(list(out)[:256], dtype=np.float64).reshape(2, 2, 64)
    print(f"== {name}: {bases.shape[0]} chunks, redo rate {st['redo_rate']:.5f}")
    row = {"checkpoint": name, "chunks": int(bases.shape[0]), "redo_rate": st["redo_rate"]}
    for layer in range(2):
        for gi, gran in enumerate(("tile32", "step16")):
            v = h[layer, gi]
            tot = v.sum()
            expect = bases.shape[0] * 8 * 8 * 8 * (1 if gi == 0 else 2)
            assert tot == expect, (tot, expect)
            ge = lambda T: float(v[T:63].sum() / tot)          # largest score at least T units below the pass-0 row maximum
            print(f"  layer {layer} {gran}: >= 14 below: {100 * ge(14):6.2f} %   >= 30 below: {100 * ge(30):6.2f} %   above the pass-0 max: {100 * v[63] / tot:6.2f} %"
                  f"   [0,1): {100 * v[0] / tot:5.1f} %  [1,4): {100 * v[1:4].sum() / tot:5.1f} %  [4,14): {100 * v[4:14].sum() / tot:5.1f} %")
            row[f"layer{layer}_{gran}"] = {"hi_only_share_T14": ge(14) - ge(30), "skip_share_T30": ge(30), "above_pass0_max": float(v[63] / tot),
                                           "hist": [int(x) for x in v]}
    res.append(row)
    eng.close()
print("TILEHISTJSON " + json.dumps(res))
