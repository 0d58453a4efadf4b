import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import seq2squiggle_amd as S
from seq2squiggle_amd import chunker
from conftest import load_ckpt, load_npz
sd, cfg = load_ckpt("k9"); g = load_npz("stages_k9.npz")
bases, nv = chunker.codes_to_bases(g["codes"])
for mode in ("f32", "f16x3"):
    eng = S.Engine(sd, cfg, mode=mode)
    out = eng.predict_chunks(torch.from_numpy(bases).cuda(), torch.from_numpy(nv).cuda(),
                             S.PredictParams(noise_std=0.0), inject_g=torch.from_numpy(g["g"]).cuda(), debug=True)
    y = out["y_scaled"].cpu().numpy(); ref = g["y_scaled_gamma"]
    d = np.abs(y - ref)
    print(mode, "nan:", np.isnan(y).sum(), "max", np.nanmax(d), "mean", np.nanmean(d), "bad chunks", np.where(np.nanmax(d, 1) > 1e-4)[0][:20], "zero rows", np.where((y == 0).all(1))[0])
    bad = np.where(np.nanmax(d, 1) > 1e-4)[0]
    if len(bad):
        b = bad[0]; print(" chunk", b, "t of max err", np.argmax(d[b]), y[b, :6], ref[b, :6])
