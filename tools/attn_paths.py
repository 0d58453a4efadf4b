#!/usr/bin/env python
"""Both decoder attention paths (s2s_set_attention_path: fast / exact) on the committed checkpoints and on sharpened ones:
speed (chunks/s, shader cycles per chunk and CU from the kernel's own counters, redo share) on 1000 x 5 kb reads,
and error against an fp64 evaluation of the oracle on the golden chunks (the fp32 oracle's own error beside it).
    tools/attn_paths.py [n_reads]          (S2S_HIP_LIB selects a library variant)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import seq2squiggle_amd as S
from seq2squiggle_amd import chunker
from oracle import s2s_oracle as O

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
torch.set_float32_matmul_precision("highest")
rng = np.random.default_rng(0)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)


def golden(tag):
    return dict(np.load(os.path.join(ROOT, "tests", "golden", f"stages_{tag}.npz")))


rows = []
for tag, k, scales in (("k9", 9, (1 / 3.0, 1.0, 2.0, 4.0, 16.0)), ("k6", 6, (1.0,))):
    sd0, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", f"synthetic_{tag}.ckpt"))
    g = golden(tag)
    gb, gnv = chunker.codes_to_bases(g["codes"])
    reads = [lut[c].tobytes().decode() for c in rng.integers(0, 4, size=(n_reads, 5000), dtype=np.uint8)]
    bases, nv, _ = S.encode_reads(reads, k)
    for scale in scales:
        sd = {kk: v.clone() for kk, v in sd0.items()}
        for kk in sd:
            if kk.startswith("decoders.") and kk.endswith(("w_qs.weight", "w_ks.weight", "w_qs.bias", "w_ks.bias")):
                sd[kk] *= scale
        p = dict(dwell_mean=12.5, dwell_std=0.0, noise_std=0.0, noise_sampling=True, duration_sampling=True, min_noise=0.0, min_duration=3.0)
        r32 = O.predict_chunks(sd, cfg, g["codes"], O.PredictParams(**p), inject_g=torch.from_numpy(g["g"]))["signal"].numpy()
        r64 = O.predict_chunks(sd, cfg, g["codes"], O.PredictParams(**p), inject_g=torch.from_numpy(g["g"]), dtype=torch.float64)["signal"].numpy()
        ok = (r32 == 0) == (r64 == 0)
        eng = S.Engine(sd, cfg, mode="f16x3")
        auto = eng.attention_path
        b, n = torch.from_numpy(bases).cuda(), torch.from_numpy(nv).cuda()
        sig = torch.empty(b.shape[0], 250, dtype=torch.float32, device="cuda")
        dur = torch.empty(b.shape[0], 16, dtype=torch.int32, device="cuda")
        for path in ("fast", "exact"):
            eng.attention_path = path
            y = eng.predict_chunks(torch.from_numpy(gb).cuda(), torch.from_numpy(gnv).cuda(), S.PredictParams(**p),
                                   inject_g=torch.from_numpy(g["g"]).cuda())["signal"].cpu().numpy()
            same = (y == 0) == (r64 == 0)
            pp = S.PredictParams(seed=42)
            eng.predict_chunks(b, n, pp, out_signal=sig, out_dur=dur)
            torch.cuda.synchronize(); eng.stats()
            t0 = time.perf_counter()
            for _ in range(3):
                eng.predict_chunks(b, n, pp, out_signal=sig, out_dur=dur)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            st = eng.stats()
            row = {"checkpoint": tag, "wq_wk_scale": round(scale, 4), "path": path, "calibrated": auto, "calibration_redo_rate": eng.calibration_redo_rate,
                   "chunks_per_sec": 3 * b.shape[0] / el, "cycles_per_chunk_and_cu": st["cycles_per_chunk_and_cu"], "clock_ghz": st["in_kernel_clock_ghz"],
                   "redo_rate": st["redo_rate"],
                   "mae_vs_fp64": float(np.abs(y - r64)[same].mean()), "max_vs_fp64": float(np.abs(y - r64)[same].max()),
                   "fp32_oracle_mae_vs_fp64": float(np.abs(r32 - r64)[ok].mean()), "fp32_oracle_max_vs_fp64": float(np.abs(r32 - r64)[ok].max()),
                   "zero_pattern_equal": float(same.mean())}
            rows.append(row)
            print(f"{tag} x{scale:<6.3g} {path:5s} (calibrated: {auto}, redo at calibration {eng.calibration_redo_rate:.4f}): {row['chunks_per_sec'] / 1e6:.3f} M chunks/s, "
                  f"{row['cycles_per_chunk_and_cu'] / 1e3:.1f} k cycles at {row['clock_ghz']:.3f} GHz, redo {row['redo_rate']:.4f} | "
                  f"MAE vs fp64 {row['mae_vs_fp64']:.2e} (max {row['max_vs_fp64']:.2e}); fp32 oracle {row['fp32_oracle_mae_vs_fp64']:.2e} (max {row['fp32_oracle_max_vs_fp64']:.2e})", flush=True)
        eng.close()
print("ATTNPATHSJSON " + json.dumps(rows))
