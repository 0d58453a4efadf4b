import os, sys, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
import seq2squiggle_amd as S
import bench as B
sd, cfg = S.load_checkpoint("tests/golden/synthetic_k9.ckpt")
eng = S.Engine(sd, cfg)
reads = B.make_reads(1000, 1234)
bases, nv, _ = S.encode_reads(reads, 9)
b, n = torch.from_numpy(bases).cuda(), torch.from_numpy(nv).cuda()
sig = torch.empty(b.shape[0], 250, device="cuda"); dur = torch.empty(b.shape[0], 16, dtype=torch.int32, device="cuda")
pp = S.PredictParams(seed=42)
for prof in (False, True, False, True):
    eng.set_profiling(prof)
    eng.predict_chunks(b, n, pp, out_signal=sig, out_dur=dur); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): eng.predict_chunks(b, n, pp, out_signal=sig, out_dur=dur)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t) / 5
    print("profiling", prof, f"{el*1e3:.2f} ms/step", f"{b.shape[0]/el/1e6:.3f} M chunks/s", eng.kernel_ms() if prof else "")
