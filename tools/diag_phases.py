#!/usr/bin/env python
"""Build the -DS2S_DIAG library, run the bench workload once and print where decoder waves spend cycles."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "seq2squiggle_amd", "lib", "libs2s_hip_diag.so")
from seq2squiggle_amd import _build
_build.compile_to(lib, ["-DS2S_DIAG", *os.environ.get("S2S_DIAG_FLAGS", "").split()])
os.environ["S2S_HIP_LIB"] = lib
import numpy as np, torch
import seq2squiggle_amd as S
from seq2squiggle_amd import _lib
sd, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"))
eng = S.Engine(sd, cfg, mode=(sys.argv[1] if len(sys.argv) > 1 else "f16x3"))
rng = np.random.default_rng(0)
reads = ["".join(rng.choice(list("ACGT"), 5000)) for _ in range(105)]
bases, nv, _ = S.encode_reads(reads, 9)
b, n = torch.from_numpy(bases).cuda(), torch.from_numpy(nv).cuda()
out = (C.c_uint64 * 384)()
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < (float(os.environ.get("S2S_DIAG_HEAT", "3"))):   # the clock under sustained load, not a cold burst
    eng.predict_chunks(b, n, S.PredictParams(seed=1))
    torch.cuda.synchronize()
_lib.lib().s2s_diag_read(eng._h, out)
for it in range(4):
    eng.predict_chunks(b, n, S.PredictParams(seed=1))
_lib.lib().s2s_diag_read(eng._h, out)
raw = np.array(list(out), dtype=np.float64).reshape(8, 48)
v = raw.sum(0)
LAUNCHES = 4
front = {32: "frontend: embedding gather", 33: "frontend: pre-net", 36: "frontend: three heads", 37: "frontend: dwell sampler, stores, + PE",
         34: "frontend: encoder attention (K/V/Q, softmax, PV, fc)", 35: "frontend: encoder LN1 + FFN + LN2"}
names = {7: "frontend phase + its two barriers (fused kernel)", 0: "entry barrier wait", 1: "K/V GEMM + LDS store", 2: "barrier 2 wait", 3: "attention (Q, S, softmax, PV, fc)",
         4: "LN1 (+ FFN fill issue, operand split)", 5: "FFN", 6: "LN2", 8: "prologue (LR gather)", 9: "epilogue",
         12: "barrier: attention done (FFN_LDS)", 13: "wait: FFN weights landed", 14: "barrier: weights visible", 15: "blocks total"}
tot = sum(v[i] for i in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 14))
nw = bases.shape[0] * 8 * LAUNCHES
for i in sorted(names):
    print(f"{names[i]:40s} {v[i] / nw:12.0f} cycles/wave  {100 * v[i] / tot:5.1f} %")
print(f"total {tot / nw:.0f} cycles per wave per chunk")
if v[17]:
    # slots 16 / 17 / 18: per wave, whole-kernel shader-clock cycles (s_memtime), 100 MHz ticks (s_memrealtime), wave count
    ghz = v[16] / v[17] * 0.1
    print(f"kernel per wave: {v[16] / v[18]:.0f} shader cycles over {v[17] / v[18] * 10:.0f} ns -> the SIMDs ran at {ghz:.3f} GHz "
          f"({v[16] / v[18] / (bases.shape[0] / 256):.0f} cycles per chunk and CU)")
    print("DIAGJSON " + __import__("json").dumps({"mode": sys.argv[1] if len(sys.argv) > 1 else "f16x3", "ghz": ghz,
          "cycles_per_chunk_and_cu": v[16] / v[18] / (bases.shape[0] / 256), "chunks_per_launch": int(bases.shape[0])}))
    print("per wave of the workgroup (waves w and w+4 share SIMD w), cycles per chunk:")
    print("  wave   K/V    attention  att-done barrier  entry barrier   FFN   frontend")
    per = bases.shape[0] * LAUNCHES
    for w in range(8):
        r = raw[w]
        print(f"  {w:4d} {r[1] / per:7.0f} {r[3] / per:10.0f} {r[12] / per:14.0f} {r[0] / per:14.0f} {r[5] / per:7.0f} {r[7] / per:8.0f}")
print("frontend waves (cycles per frontend wave and chunk; the waves run one frontend per group of chunks):")
for i in sorted(front):
    print(f"{front[i]:50s} {v[i] / bases.shape[0] / LAUNCHES:12.0f}")
if v[11]:
    print(f"safe-path redos: {v[10]} of {v[11]} (wave, head) softmax runs = {100 * v[10] / v[11]:.2f} %")
