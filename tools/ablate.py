#!/usr/bin/env python
"""Timing-only ablations of the decoder kernel (-DS2S_ABL=mask builds; outputs are garbage).
f32 block: bit0 no softmax VALU, bit1 no LDS operand reads, bit2 no barriers, bit3 no K/V LDS stores, bit4 no LayerNorm.
f16 block: 1 no exp, 2 no split, 4 no barriers, 32 no max/branch, 64 no row-sum MFMAs, 128 no FFN, 256 one key pass of four; unit loads L1-hot: 2048 K/V phase, 8192 attention phase, 16384 FFN.
S2S_ABL_FLAGS adds extra -D flags (e.g. -DS2S_NO_FALLBACK so that garbage data cannot take the safe softmax path).
The decoder's attention core is softmax_pv32 (32x32x16 MFMA) since round 2: of the f16-block masks it honours 1, 2 (exp / split), 4,
128 and the unit-load masks; 32, 64, 256, 512, 1024 belong to the 16x16x32 core (add -DS2S_ATT32=0 to S2S_ABL_FLAGS to ablate that one)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from seq2squiggle_amd import _build
masks = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 4, 8, 16, 31]
for m in masks:
    lib = os.path.join(ROOT, "seq2squiggle_amd", "lib", f"libs2s_hip_abl{m}.so")
    _build.compile_to(lib, [f"-DS2S_ABL={m}", *os.environ.get("S2S_ABL_FLAGS", "").split()])
    env = dict(os.environ, S2S_HIP_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--reads", "420",
                        "--no-cpu-baseline", "--mode", os.environ.get("S2S_ABL_MODE", "f16x3")], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"ABL={m:3d}  chunks/s {d['chunks_per_sec']:10.0f}  decoder frac-of-peak {d['roofline']['frac']:.3f}", flush=True)
    except Exception as e:
        print("ABL", m, "failed", e, r.stderr[-500:])
