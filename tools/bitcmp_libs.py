"""Are two builds of the library bit-identical on the bench-like workload?  python tools/bitcmp_libs.py libA.so libB.so [mode]
Each library is loaded in a process of its own (S2S_HIP_LIB), runs 6,000 chunks with the built-in samplers (seed 42) and
prints a hash of signal + dur; the parent compares."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import seq2squiggle_amd as S
    sd, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"))
    eng = S.Engine(sd, cfg, mode=sys.argv[2])
    rng = np.random.default_rng(7)
    reads = ["".join(rng.choice(list("ACGT"), int(n))) for n in rng.integers(9, 3000, size=70)]
    bases, nv, _ = S.encode_reads(reads, 9)
    out = eng.predict_chunks(torch.from_numpy(bases).cuda(), torch.from_numpy(nv).cuda(), S.PredictParams(seed=42))
    h = hashlib.sha1(out["signal"].cpu().numpy().tobytes() + out["dur"].cpu().numpy().tobytes()).hexdigest()
    print("HASH", bases.shape[0], h)
    sys.exit(0)
mode = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
res = []
for lib in sys.argv[1:3]:
    r = subprocess.run([sys.executable, __file__, "--child", mode], env=dict(os.environ, S2S_HIP_LIB=os.path.abspath(lib)),
                       capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("HASH")]
    res.append(line[0] if line else "FAILED " + r.stderr[-300:])
    print(lib, res[-1])
print("IDENTICAL" if res[0] == res[1] and res[0].startswith("HASH") else "DIFFERENT")
