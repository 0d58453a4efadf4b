#!/usr/bin/env python
"""bench.py -- throughput of the predict hot path on MI355X (contract: see the task statement).

A "step" is one pass of the hot path (s2s_predict_chunks: the fused frontend + decoder kernel, ONE launch
per step -- a launch takes up to 2^20 chunks) over one batch of synthetic reads already resident in HBM.  Workload at N=1 (BASELINE.json configs[1]
shape): 1000 reads x 5000 nt = 312,000 chunks, default noise + duration samplers (noise_std 2.0,
min_duration 3), synthetic k=9 checkpoint.  With N>1 every rank runs the same amount of work on
its own read shard (weak scaling, no data-path collective).  The path has no exchange step, so the bench needs no RCCL:
the barrier around the timed region, the max-over-ranks and the per-rank gather run on a HOST (gloo) group, each side of a
torch.cuda.synchronize().  RCCL is a reported self-test (`rccl_selftest`: N throw-away child processes, wall-limited, started
as the last leg that touches a GPU so that nothing it does can cost a measurement), not a dependency.

`python bench.py --gpus N` without a torchrun environment starts its N ranks itself, as CHILD processes
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...`), before
anything in the parent has touched the GPU, relays rank 0's JSON line and exits with the children's code.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import seq2squiggle_amd as S  # noqa: E402

FLOP_PER_CHUNK = 85_083_392            # 2 x 42,541,696 MAC, k = 9
FLOP_PER_CHUNK_DOMINANT = FLOP_PER_CHUNK  # the one kernel of the path does all of it (of which 81,184,000 decoder-side)
DOMINANT_KERNEL = "s2s_fused_kernel"
PEAK_F32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 256 CU x 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense f16/bf16 MFMA
# roofline.frac is always quoted against the guide's dense peak of the MFMA dtype the mode issues.  (f16x3 evaluates every
# algorithmic product as three f16 MFMA products, so its own ceiling for ALGORITHMIC flops is a third of that: reported
# as the secondary field frac_of_f16x3_ceiling.)
PEAK = {"f32": PEAK_F32_MFMA_TFLOPS, "f16x3": PEAK_F16_MFMA_TFLOPS, "f16": PEAK_F16_MFMA_TFLOPS}
READ_LEN, CHUNKS_PER_READ = 5000, 312


def make_reads(n_reads, seed):
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, size=(n_reads, READ_LEN), dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    return [lut[c].tobytes().decode() for c in codes]


def pmc_counters(mode, chunks_per_launch, root=None):
    """From the newest committed PMC passes (tools/pmc_run.sh -> profiles/*/pmc_summary.json), for the decoder kernel:
    HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB -- FETCH_SIZE reports half of a wide coalesced read on
    gfx950 (MI355X_MICROARCH.md, HBM) -- measured on one full dispatch and scaled to this run's launch size; matrix-core
    busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs; vector issue fraction =
    4 x SQ_ACTIVE_INST_VALU (quad-cycles) / 1024 over the same cycles."""
    import glob
    root = root or ROOT
    for path in sorted(glob.glob(os.path.join(root, "profiles", "*", "pmc_summary.json")), reverse=True):
        try:
            whole = json.load(open(path))
            d = whole.get(mode, {})
            k = max((k for k in d if "fused" in k), key=lambda k: d[k].get("_launch", {}).get("chunks", 0))   # (not the 512-chunk calibration launch)
            c = d[k]
            chunks = c.get("_launch", {}).get("chunks", 32768)
            per_chunk = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / chunks
            cyc = c["GRBM_GUI_ACTIVE"] / 8
            clock = None                       # in-kernel clock of the un-profiled diagnostic build (tools/diag_phases.py), same round
            try:
                clock = json.load(open(os.path.join(os.path.dirname(path), "diag_clock.json"))).get(mode)
            except Exception:
                pass
            return {"traffic": per_chunk * chunks_per_launch, "mfma_busy": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc,
                    "valu_issue": 4 * c["SQ_ACTIVE_INST_VALU"] / 1024 / cyc, "source": os.path.relpath(path, root),
                    "effective_clock_ghz": c.get("_effective_clock_ghz"), "in_kernel_clock": clock,
                    "extra": dict(pmc_staleness(whole.get("_meta")), **mfma_issue_fields(mode, c, chunks))}
        except Exception:
            continue
    return {"traffic": None, "mfma_busy": None, "valu_issue": None, "source": None, "effective_clock_ghz": None, "in_kernel_clock": None,
            "extra": {"pmc_stale": None}}


def pmc_staleness(meta):
    """Do the committed counters describe the kernel source that is running?  tools/pmc_run.sh records the sha256 of the library's
    sources (_build.source_hash) beside the passes, tools/pmc_summarize.py carries it into pmc_summary.json with the git commit;
    here it is compared with the sources of THIS tree.  -> pmc_stale true | false | None (a summary from before round 6: no hash)."""
    from seq2squiggle_amd import _build
    now = _build.source_hash()
    then = (meta or {}).get("csrc_sha256")
    return {"pmc_stale": None if not then else then != now, "pmc_csrc_sha256": then, "csrc_sha256": now,
            "pmc_git_commit": (meta or {}).get("git_commit")}


MFMA_FLOP = {"16x16x32_f16": 2 * 16 * 16 * 32, "32x32x16_f16": 2 * 32 * 32 * 16, "16x16x4_f32": 2 * 16 * 16 * 4}


def mfma_issue_fields(mode, c, chunks):
    """What the matrix pipe was ASKED to do per chunk, from the instruction counters of the profiled launch: SQ_INSTS_VALU_MFMA_MOPS_*
    counts issued matrix math in units of 512 operations (a full EXEC mask), SQ_INSTS_MFMA the wave instructions; with two shapes in
    the kernel (f16: 16x16x32 for the GEMMs, 32x32x16 for the attention core) the two counts give the split.  mfma_useful_frac =
    algorithmic FLOP / issued FLOP: the rest is padding (phantom keys, unused A rows of the P.V product, the hi / lo stacking)."""
    mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32" if mode == "f32" else "SQ_INSTS_VALU_MFMA_MOPS_F16")
    n = c.get("SQ_INSTS_MFMA")
    if not mops or not n or not chunks:
        return {"mfma_issued_flop_per_chunk": None, "mfma_useful_frac": None}
    issued = 512.0 * mops / chunks
    out = {"mfma_issued_flop_per_chunk": issued, "mfma_useful_frac": FLOP_PER_CHUNK / issued,
           "mfma_wave_instructions_per_chunk": n / chunks}
    if mode != "f32":
        # n16 + n32 = n;  16384 n16 + 32768 n32 = 512 mops
        n32 = (512.0 * mops - MFMA_FLOP["16x16x32_f16"] * n) / (MFMA_FLOP["32x32x16_f16"] - MFMA_FLOP["16x16x32_f16"])
        out["mfma_by_shape_per_chunk"] = {"16x16x32_f16": (n - n32) / chunks, "32x32x16_f16": n32 / chunks}
        out["mfma_useful_frac_note"] = ("algorithmic FLOP over issued FLOP; of the issued, 1/3 at most is algorithmic in the three-product "
                                        "split, the rest of the gap is padding")
    return out


def _e2e_dirs():
    return "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None


def _e2e_run(mode, n_reads, ext="blow5", where="default", fasta=None, r=5000, c=-1, ckpt="synthetic_k9.ckpt",
             profile="dna-r10-prom", noise_std=2.0):
    """One inference_run (FASTA -> container file in a fresh temporary directory) -> (seconds, chunks, output bytes)."""
    import tempfile
    from seq2squiggle_amd.cli import set_config
    from seq2squiggle_amd.inference import inference_run
    from seq2squiggle_amd.utils import set_seeds
    fasta = fasta or os.path.join(ROOT, "tests", "golden", "example_lambda_genome.fasta")
    with tempfile.TemporaryDirectory(dir=_e2e_dirs() if where == "default" else where) as td:
        out = os.path.join(td, "o." + ext)
        set_seeds(42)
        t0 = time.perf_counter()
        m = inference_run(config=set_config(None), saved_weights=os.path.join(ROOT, "tests", "golden", ckpt),
                          fasta=fasta, read_input=False, n=n_reads, r=r, c=c, out=out, profile=profile,
                          dwell_mean=None, dwell_std=0.0, noise_std=noise_std, noise_sampling=True, duration_sampling=True,
                          distr="expon", predict_batch_size=1024, export_every_n_samples=1000000, sample_rate=None,
                          bps=None, digitisation=None, range_val=None, offset_mean=None, offset_std=None,
                          median_before_mean=None, median_before_std=None, min_noise=0.0, min_duration=3, min_read_len=30,
                          preserve_read_ids=False, seed=42, mode=mode)
        el = time.perf_counter() - t0
        size = os.path.getsize(out)
        chunks = m.chunks_done
        m.engine.close()
    return el, chunks, size


def end_to_end_one(n_reads, mode="f16x3"):
    """A single rank's FASTA -> BLOW5 run of `n_reads` lambda reads (second, warm call timed): the yardstick the sharded legs and
    tests/test_gpu_end_to_end.py compare with."""
    _e2e_run(mode, min(n_reads, 1000))
    el, chunks, size = _e2e_run(mode, n_reads)
    return {"reads": n_reads, "seconds": el, "chunks": chunks, "chunks_per_sec": chunks / el, "output_bytes": size}


def end_to_end(mode):
    """FASTA -> BLOW5 wall time through inference_run (engine creation, read sampling, chunking, the fused kernel, GPU
    zero-strip + DAC, D2H, record compression, file write) for BASELINE.json configs[1] (lambda genome -n 1000 -r 5000,
    default samplers), for one GPU's share of configs[2] (12,500 of the 100,000 reads) and of configs[4] (37,500 10-kb reads
    from a synthetic 12.5 Mb reference -> .pod5).  Reported beside the resident-input kernel throughput, never as `value`.
    Output goes to /dev/shm when it exists, except where stated (`real_file_system`, `config5_share`: the default temp
    directory, an overlay file system on the GPU boxes)."""
    import tempfile
    out_dir = _e2e_dirs()

    def run(n_reads, ext="blow5", where=out_dir, fasta=None, r=5000, c=-1, **kw):
        return _e2e_run(mode, n_reads, ext, where, fasta, r, c, **kw)
    first, _, _ = run(1000)            # the first call also pays the process's one-time costs (pinned buffers, thread pools)
    warm = [run(1000) for _ in range(3)]                # host-side timing moves by several ms from call to call: median of three
    el, chunks, size = sorted(warm)[1]
    el3, chunks3, size3 = run(12500)
    # BASELINE.json configs[3] ("dna_r9_min profile -n 100000 -r 8000 --noise-std 1.5", k = 6: utils.py:257-260): one GPU's share
    run(1000, r=8000, ckpt="synthetic_k6.ckpt", profile="dna-r9-min", noise_std=1.5)
    el4, chunks4, size4 = run(12500, r=8000, ckpt="synthetic_k6.ckpt", profile="dna-r9-min", noise_std=1.5)
    run(1000, "pod5")
    warm_p = [run(1000, "pod5") for _ in range(3)]
    elp, chunksp, sizep = sorted(warm_p)[1]
    # the same configs[1] run onto the default temp directory (the box's real file system, not tmpfs)
    disk_dir = tempfile.gettempdir()
    disk = sorted(run(1000, where=None) for _ in range(3))[1]
    # BASELINE.json configs[4] ("synthetic 100 Mb reference -c 30 -r 10000, pod5 out", 300,000 reads over 8 GPUs): ONE GPU's
    # share -- a 12.5 Mb reference of 5 unequal contigs (rng 1234), -c 30 -r 10000 -> 37,500 reads ~ 23.6 M chunks ~ 5.7 GB of
    # .pod5 -- written to the real file system when it has room (else tmpfs; else a stated fraction of the share)
    import resource
    import shutil
    from seq2squiggle_amd.utils import write_synthetic_reference
    need = 9 << 30
    where5, frac5 = None, 1.0
    if shutil.disk_usage(disk_dir).free < need:
        where5 = out_dir
        if out_dir is None or shutil.disk_usage(out_dir).free < need:
            where5, frac5 = out_dir, 0.125
    with tempfile.TemporaryDirectory(dir=out_dir) as td:
        ref = os.path.join(td, "synthetic_ref.fasta")
        total5 = write_synthetic_reference(ref, [int(L * frac5) for L in (4_000_000, 3_000_000, 2_500_000, 2_000_000, 1_000_000)])
        rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        el5, chunks5, size5 = run(-1, "pod5", where=where5, fasta=ref, r=10000, c=30)
        rss5 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    reads5 = round(30 * total5 / 10000)
    return {"workload": "example lambda genome -n 1000 -r 5000 -> .blow5 (zlib records), seed 42", "seconds": el,
            "warm_calls_seconds": [w[0] for w in warm], "pod5_warm_calls_seconds": [w[0] for w in warm_p],
            "first_call_seconds": first, "reads_per_sec": 1000 / el, "chunks": chunks, "chunks_per_sec": chunks / el,
            "output_bytes": size, "output_dir": out_dir or tempfile.gettempdir(),
            "config3_share": {"workload": "example lambda genome -n 12500 -r 5000 -> .blow5: one GPU's share of configs[2]",
                              "seconds": el3, "reads_per_sec": 12500 / el3, "chunks": chunks3, "chunks_per_sec": chunks3 / el3,
                              "output_bytes": size3},
            "config4_share": {"workload": "example lambda genome -n 12500 -r 8000 --profile dna-r9-min --noise-std 1.5 (k = 6 checkpoint) "
                                          "-> .blow5: one GPU's share of BASELINE configs[3]",
                              "seconds": el4, "reads_per_sec": 12500 / el4, "chunks": chunks4, "chunks_per_sec": chunks4 / el4,
                              "output_bytes": size4},
            "pod5": {"workload": "configs[1]'s reads -> .pod5 (VBZ signal rows: svb16 on the GPU, zstd on host threads), the "
                                 "container BASELINE configs[4] asks for", "seconds": elp, "reads_per_sec": 1000 / elp,
                     "chunks": chunksp, "chunks_per_sec": chunksp / elp, "output_bytes": sizep},
            "real_file_system": {"workload": "configs[1] again, output on the default temp directory instead of tmpfs",
                                 "output_dir": disk_dir, "seconds": disk[0], "chunks_per_sec": disk[1] / disk[0],
                                 "reads_per_sec": 1000 / disk[0]},
            "config5_share": {"workload": f"synthetic {total5 / 1e6:.3f} Mb reference (5 contigs, rng 1234) -c 30 -r 10000 -> .pod5 "
                                          f"(VBZ rows): {frac5:g} of one GPU's share of BASELINE configs[4] (300,000 reads over 8 GPUs)",
                              "fraction_of_one_gpu_share": frac5, "reads": reads5, "seconds": el5, "reads_per_sec": reads5 / el5,
                              "chunks": chunks5, "chunks_per_sec": chunks5 / el5, "output_bytes": size5,
                              "output_dir": where5 or disk_dir,
                              "process_peak_rss_mb_before": rss0 / 1024.0, "process_peak_rss_mb_after": rss5 / 1024.0},
            "includes": "engine creation, read sampling, chunking, kernels, export, D2H, compression, file write"}


def parity_vs_oracle(sd, cfg, eng, n=256):
    """Engine against the CPU oracle on n chunks with injected variates."""
    from oracle import s2s_oracle as O
    from seq2squiggle_amd import chunker
    torch.set_float32_matmul_precision("highest")
    codes = np.concatenate([O.encode_read(r, cfg["seq_kmer"]) for r in make_reads(1, 99)], 0)[:n]
    gen = torch.Generator().manual_seed(0)
    g = torch.rand(n, 16, generator=gen) * 20
    z = torch.randn(n, 250, generator=gen)
    ref = O.predict_chunks(sd, cfg, codes, O.PredictParams(), inject_g=g, inject_z01=z)
    b_, nv_ = chunker.codes_to_bases(codes)
    got = eng.predict_chunks(torch.from_numpy(b_).to(eng.device), torch.from_numpy(nv_).to(eng.device), S.PredictParams(),
                             inject_g=g.to(eng.device), inject_z01=z.to(eng.device))
    y, r = got["signal"].cpu().numpy(), ref["signal"].numpy()
    same = (y == 0) == (r == 0)
    return {"chunks": n, "signal_mae_pa": float(np.abs(y - r)[same].mean()), "signal_max_abs_pa": float(np.abs(y - r)[same].max()),
            "dwell_indices_equal": bool(np.array_equal(got["dur"].cpu().numpy(), ref["dur"].numpy())),
            "zero_pattern_equal_fraction": float(same.mean())}


def reduced_precision_leg(sd, cfg, bases_d, nv_d, sig, dur, params, steps):
    """NOT the headline: the same workload in S2S_MODE_F16 (decoder operands rounded to f16 once, one MFMA product per product;
    frontend unchanged, so the dwell indices stay bit-exact).  Reported with its measured error, beside the error of the reference's
    own GPU precision (fp16 autocast, inference.py:403-404) on the same chunks, so that nobody has to guess what it costs."""
    eng = S.Engine(sd, cfg, device=bases_d.device.index, mode="f16")
    eng.predict_chunks(bases_d, nv_d, params, out_signal=sig, out_dur=dur)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.predict_chunks(bases_d, nv_d, params, out_signal=sig, out_dur=dur)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {"mode": "f16", "chunks_per_sec": bases_d.shape[0] * steps / el, "samples_per_sec": bases_d.shape[0] * steps * 250 / el,
           "parity": parity_vs_oracle(sd, cfg, eng), "tolerance_note": "outside the 1e-4 pA parity bound by design"}
    # beside the reference's OWN GPU arithmetic (inference.py:403-404: "16-mixed" whenever a GPU is present): the imported reference's
    # predict_step under fp16 autocast on the golden chunks, same injected variates (tests/golden/mixed16.npz, tools/make_goldens.py
    # mixed16), against the mode on the same chunks -- both measured against the reference's fp32 result
    try:
        from seq2squiggle_amd import chunker
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", "stages_k9.npz")))
        m16 = dict(np.load(os.path.join(ROOT, "tests", "golden", "mixed16.npz")))
        gb, gnv = chunker.codes_to_bases(g["codes"])
        dev = bases_d.device
        got = eng.predict_chunks(torch.from_numpy(gb).to(dev), torch.from_numpy(gnv).to(dev), S.PredictParams(),
                                 inject_g=torch.from_numpy(g["g"]).to(dev), inject_z01=torch.from_numpy(np.ascontiguousarray(g["z01"])).to(dev))
        y, t = got["signal"].cpu().numpy(), g["y_gamma_nsamp"]
        agree = (m16["dur_gamma_16mixed_k9"] == g["dur_gamma"]).all(1)
        ref_d = np.abs(m16["y_gamma_nsamp_16mixed_k9"] - t)
        out["vs_reference_gpu_precision"] = {
            "chunks": int(len(t)), "mode_mae_pa": float(np.abs(y - t).mean()), "mode_max_abs_pa": float(np.abs(y - t).max()),
            "mode_dwell_indices_equal": bool(np.array_equal(got["dur"].cpu().numpy(), g["dur_gamma"])),
            "reference_16mixed_mae_pa": float(ref_d[agree].mean()), "reference_16mixed_max_abs_pa": float(ref_d[agree].max()),
            "reference_16mixed_mae_pa_all_chunks": float(ref_d.mean()), "reference_16mixed_dwell_indices_differing": int(m16["dwell_indices_differing_k9"]),
            "note": "distance to the reference's fp32 golden on the golden chunks; reference_16mixed = the imported reference under "
                    "torch.autocast(float16) (CPU stand-in for its GPU path), on the chunks whose dwell indices it keeps"}
    except Exception as e:
        out["vs_reference_gpu_precision"] = {"error": f"{type(e).__name__}: {e}"}
    eng.close()
    return out


def parse_rocm_smi(text):
    """`rocm-smi --showpower --showclocks --showmaxpower` (text form) -> {"watts", "cap_watts", "sclk_mhz"} of GPU[0] (None where absent)."""
    import re

    def first(pat):
        m = re.search(pat, text)
        return float(m.group(1)) if m else None
    return {"watts": first(r"GPU\[0\]\s*:\s*Current Socket Graphics Package Power \(W\):\s*([0-9.]+)") or
            first(r"GPU\[0\]\s*:\s*Average Graphics Package Power \(W\):\s*([0-9.]+)"),
            "cap_watts": first(r"GPU\[0\]\s*:\s*Max Graphics Package Power \(W\):\s*([0-9.]+)"),
            "sclk_mhz": first(r"GPU\[0\]\s*:\s*sclk clock level:[^(]*\(([0-9.]+)Mhz\)")}


def power_leg(eng, bases_d, nv_d, sig, dur, params, seconds=7.0):
    """What bounds this kernel, LIVE: the package power and shader clock `rocm-smi` reports while the headline's launches run back to
    back on a helper thread (three samples, the first after 2.5 s), and the energy per chunk that follows.  The f16x3 kernel sits at
    the package's power cap: its speed is energy per chunk, not issue slots (profiles/r06/ab_mfma_split.txt)."""
    import threading
    stop, done = threading.Event(), {"launches": 0}

    def run():
        torch.cuda.set_device(eng.device)
        while not stop.is_set():
            eng.predict_chunks(bases_d, nv_d, params, out_signal=sig, out_dur=dur)
            done["launches"] += 1
            if done["launches"] % 4 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = threading.Thread(target=run, name="s2s-power-load")
    th.start()
    samples = []
    try:
        time.sleep(2.5)
        while time.perf_counter() - t0 < seconds and len(samples) < 3:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=20)
            samples.append(parse_rocm_smi(r.stdout))
            time.sleep(0.7)
    finally:
        stop.set()
        th.join()
    el = time.perf_counter() - t0
    rate = done["launches"] * bases_d.shape[0] / el
    watts = [x["watts"] for x in samples if x.get("watts")]
    if not watts:
        return {"error": "rocm-smi reported no package power", "chunks_per_sec": rate}
    w = sum(watts) / len(watts)
    cap = next((x["cap_watts"] for x in samples if x.get("cap_watts")), None)
    return {"package_watts": w, "package_watts_samples": watts, "cap_watts": cap, "of_cap": (w / cap) if cap else None,
            "sclk_mhz_samples": [x["sclk_mhz"] for x in samples], "chunks_per_sec": rate, "microjoule_per_chunk": w / rate * 1e6,
            "source": "rocm-smi --showpower --showclocks --showmaxpower, sampled while the headline's launches run back to back"}


FLOP_PER_CHUNK_K6 = 85_052_672         # k = 6: the embedding gather takes 6 columns instead of 9, everything else is the same


def _timed_steps(eng, bases_d, nv_d, sig, dur, params, steps):
    """`steps` launches of the resident batch -> (chunks/s by wall, Engine.stats() of exactly those launches, kernel ms per launch)."""
    eng.predict_chunks(bases_d, nv_d, params, out_signal=sig, out_dur=dur)
    torch.cuda.synchronize()
    eng.stats()
    eng.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.predict_chunks(bases_d, nv_d, params, out_signal=sig, out_dur=dur)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    eng.set_profiling(False)
    ms, nl, _ = eng.kernel_ms()
    return bases_d.shape[0] * steps / el, eng.stats(), (ms / nl if nl else None)


def _stats_fields(st):
    return {"softmax_redo_rate": st["redo_rate"], "in_kernel_clock_ghz": st["in_kernel_clock_ghz"],
            "cycles_per_chunk_and_cu": st["cycles_per_chunk_and_cu"]}


def config4_k6_leg(mode, dev, steps):
    """BASELINE.json configs[3]'s shape with inputs resident: the k = 6 chemistry (dna-r9-min, utils.py:257-260), 8 kb reads,
    --noise-std 1.5 -- 625 reads x 500 chunks = 312,500 chunks per launch, synthetic k = 6 checkpoint -- with its own
    roofline fraction and the kernel's own counters for those launches."""
    sd, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", "synthetic_k6.ckpt"))
    eng = S.Engine(sd, cfg, device=dev.index, mode=mode)
    rng = np.random.default_rng(4321)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = [lut[c].tobytes().decode() for c in rng.integers(0, 4, size=(625, 8000), dtype=np.uint8)]
    bases, nv, _ = S.encode_reads(reads, 6)
    b, n = torch.from_numpy(bases).to(dev), torch.from_numpy(nv).to(dev)
    sig = torch.empty(b.shape[0], 250, dtype=torch.float32, device=dev)
    dur = torch.empty(b.shape[0], 16, dtype=torch.int32, device=dev)
    rate, st, ms = _timed_steps(eng, b, n, sig, dur, S.PredictParams(seed=42, noise_std=1.5), steps)
    eng.close()
    tflops = FLOP_PER_CHUNK_K6 * b.shape[0] / (ms * 1e-3) / 1e12
    return {"workload": "625 synthetic reads x 8000 nt (312,500 chunks per launch), k = 6 checkpoint, noise_std 1.5, default samplers",
            "chunks_per_sec": rate, "reads_per_sec": rate / 500, "samples_per_sec": rate * 250, "avg_launch_ms": ms,
            "roofline": {"bound": "mfma", "achieved": tflops, "peak": PEAK[mode], "unit": "TFLOP/s", "frac": tflops / PEAK[mode],
                         "flop_per_chunk": FLOP_PER_CHUNK_K6}, **_stats_fields(st)}


def weight_sensitivity_leg(mode, bases_d, nv_d, sig, dur, params, steps):
    """How the headline moves with the WEIGHTS.  No trained checkpoint exists offline (the reference downloads them,
    inference.py:151-208), and this kernel is data dependent in two ways: the fast softmax path is redone on a safe path for
    heads whose later keys beat the first 64 keys' maximum by more than the f16 range, and the chip's clock under the kernel
    follows the operands.  Same workload as the headline, decoder w_qs / w_ks (weights and biases) of the committed k = 9
    checkpoint scaled: x 1/3 is torch's default init, x 1 the committed checkpoint (the headline), x 4 and x 16 the sharpened
    ones of tests/test_gpu_parity.py::test_peaked_attention_forces_the_rescale_fallback.  `attention_path` is what s2s_create's
    calibration launch chose for those weights (include/s2s_hip.h: s2s_set_attention_path); the k = 6 checkpoint is the
    `config4_k6` object."""
    sd0, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"))
    rows = []
    for scale, name in ((1.0 / 3.0, "default init (committed / 3)"), (1.0, "committed synthetic_k9.ckpt (headline)"),
                        (2.0, "committed x 2"), (4.0, "committed x 4"), (8.0, "committed x 8"), (16.0, "committed x 16"),
                        (None, "positional: decoder w_qs = w_ks = 2 I, biases 0 (scores follow the sinusoid table: local attention)")):
        sd = {k: v.clone() for k, v in sd0.items()}
        for k in sd:
            if k.startswith("decoders.") and k.endswith(("w_qs.weight", "w_ks.weight", "w_qs.bias", "w_ks.bias")):
                if scale is not None:
                    sd[k] *= scale
                else:                      # a structured case beside the scaled random ones: q.k = 4 x.x', dominated by position_enc
                    sd[k] = 2.0 * torch.eye(64) if k.endswith("weight") else torch.zeros(64)
        eng = S.Engine(sd, cfg, device=bases_d.device.index, mode=mode)
        rate, st, ms = _timed_steps(eng, bases_d, nv_d, sig, dur, params, steps)
        rows.append({"decoder_wq_wk_scale": scale, "checkpoint": name, "chunks_per_sec": rate, "avg_launch_ms": ms,
                     "attention_path": eng.attention_path, "calibration_redo_rate": eng.calibration_redo_rate, **_stats_fields(st)})
        eng.close()
    rates = [r["chunks_per_sec"] for r in rows]
    return {"workload": "the headline's resident batch, decoder w_qs / w_ks scaled", "rows": rows,
            "min_chunks_per_sec": min(rates), "max_chunks_per_sec": max(rates)}


def cpu_baseline(sd, cfg, eng, seconds_target=12.0, sample_chunks=1024, whole_node=None):
    """Oracle (CPU port of the reference op sequence, torch fp32 'highest') on the host cores, plus a live parity
    check of the engine against it on the first 256 chunks of the sample (injected variates)."""
    from oracle import s2s_oracle as O
    from seq2squiggle_amd.signal_io import cpu_share
    torch.set_float32_matmul_precision("highest")
    threads = cpu_share()                   # the cores this container may actually keep busy (cgroup quota), not the host's count
    if whole_node:
        # N > 1: rank 0 runs this leg alone while the other ranks wait at the host barrier -- it may use the node's whole CPU share,
        # not its own 1 / N of it: the mask placement.pin_rank narrowed is widened again for the leg's threads
        try:
            from seq2squiggle_amd.placement import parse_cpulist
            os.sched_setaffinity(0, parse_cpulist(whole_node))
        except (OSError, ValueError, AttributeError):
            pass
        threads *= max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    torch.set_num_threads(threads)
    reads = make_reads(4, 99)
    codes = np.concatenate([O.encode_read(r, cfg["seq_kmer"]) for r in reads], 0)[:sample_chunks]
    p = O.PredictParams()
    gen = torch.Generator().manual_seed(0)
    O.predict_chunks(sd, cfg, codes[:min(64, sample_chunks)], p, generator=gen)            # warm-up
    t0, done = time.perf_counter(), 0
    while True:
        O.predict_chunks(sd, cfg, codes, p, generator=gen)
        done += codes.shape[0]
        el = time.perf_counter() - t0
        if el >= seconds_target or done >= 8 * 1024:
            break
    if seconds_target < 1.0:                               # (self-test: no second timing pass)
        return {"value": done * 250 / el, "unit": "samples/s", "cores": torch.get_num_threads(), "parity": None, "kind": "port",
                "sample": f"{done} chunks in {el:.1f} s (launcher self-test)"}
    # the reference ships torch.set_float32_matmul_precision("medium") (model.py:22): time that too (bf16-capable CPUs
    # may take a faster, less exact matmul path); parity is only ever claimed against "highest"
    torch.set_float32_matmul_precision("medium")
    t1, done_m = time.perf_counter(), 0
    while True:
        O.predict_chunks(sd, cfg, codes, p, generator=gen)
        done_m += codes.shape[0]
        el_m = time.perf_counter() - t1
        if el_m >= seconds_target / 3 or done_m >= 3 * 1024:
            break
    torch.set_float32_matmul_precision("highest")
    parity = None
    if eng is not None:                                    # (the launcher's CPU self-test times the oracle without an engine)
        n = 256
        g = torch.rand(n, 16, generator=gen) * 20
        z = torch.randn(n, 250, generator=gen)
        ref = O.predict_chunks(sd, cfg, codes[:n], p, inject_g=g, inject_z01=z)
        from seq2squiggle_amd import chunker
        b_, nv_ = chunker.codes_to_bases(codes[:n])
        got = eng.predict_chunks(torch.from_numpy(b_).to(eng.device), torch.from_numpy(nv_).to(eng.device), S.PredictParams(),
                                 inject_g=g.to(eng.device), inject_z01=z.to(eng.device))
        y, r = got["signal"].cpu().numpy(), ref["signal"].numpy()
        parity = {"chunks": n, "signal_mae_pa": float(np.abs(y - r).mean()), "signal_max_abs_pa": float(np.abs(y - r).max()),
                  "dwell_indices_equal": bool(np.array_equal(got["dur"].cpu().numpy(), ref["dur"].numpy())),
                  "zero_pattern_equal": bool(np.array_equal(y == 0, r == 0)), "tolerance_mae_pa": 1e-4}
    return {"value": done * 250 / el, "unit": "samples/s", "cores": torch.get_num_threads(), "parity": parity,
            "host_logical_cpus": len(os.sched_getaffinity(0)), "cpu_quota": cpu_share(), "kind": "port",
            "reads_per_sec": done / CHUNKS_PER_READ / el, "value_matmul_precision_medium": done_m * 250 / el_m,
            "sample": f"{done} chunks (batches of {sample_chunks}, same 5 kb synthetic reads, default samplers) in {el:.1f} s"}


def end_to_end_sharded(mode, dist, rank, world, dev):
    """N > 1: what the resident-input `value` cannot show.  Every rank runs the real product path on ITS shard of BASELINE
    configs[2] scaled to the rank count (lambda genome -n 12500 x world -r 5000: inference_run shards the read set by RANK /
    WORLD_SIZE exactly as `torchrun ... seq2squiggle_amd predict` does -- shared seed, native sampler skip-ahead, per-rank
    out.rankN.blow5 on /dev/shm), all ranks at once: host threads are cpu_share() = quota / LOCAL_WORLD_SIZE per rank, the
    FASTA is parsed and the sampler replayed by every rank.  Wall = max over ranks between two barriers.

    The leg must never cost the headline line: its barriers and its gather run on a gloo group of their own with a short timeout
    (a rank that died would otherwise hold the others in an RCCL barrier for ten minutes), a rank whose run raises still reaches
    both barriers and reports the error, and the output directory is chosen by free space (1.3 GB per rank and pass).

    Then the part a sharded run owes the reference's ONE output file: rank 0 joins the rank files (`merge_seconds`, `with_merge`),
    and runs the user-facing command `predict --gpus N` itself once (`one_command`: its launch, predict and merge seconds)."""
    import shutil
    import tempfile
    from datetime import timedelta
    from seq2squiggle_amd.cli import set_config
    from seq2squiggle_amd.inference import inference_run
    from seq2squiggle_amd.signal_io import cpu_share
    from seq2squiggle_amd.utils import set_seeds
    fasta = os.path.join(ROOT, "tests", "golden", "example_lambda_genome.fasta")
    grp = dist.new_group(backend="gloo", timeout=timedelta(seconds=300))
    n_total = 12500 * world
    need = 3.0e9 * int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))     # rank files + the merged file, twice over (the leg, then the command)
    choice = [None]
    if rank == 0:
        for d in ("/dev/shm", tempfile.gettempdir()):
            if os.path.isdir(d) and os.access(d, os.W_OK) and shutil.disk_usage(d).free > need:
                choice[0] = d
                break
    dist.broadcast_object_list(choice, src=0, group=grp)
    out_dir = choice[0]
    if out_dir is None:
        return {"skipped": f"no directory with {need / 1e9:.0f} GB free for the rank shard files"} if rank == 0 else None

    def run(keep):
        td = tempfile.mkdtemp(dir=out_dir)
        err, own, chunks, size = None, 0.0, 0, 0
        try:
            set_seeds(42)
            dist.barrier(group=grp)
            t0 = time.perf_counter()
            try:
                m = inference_run(config=set_config(None), saved_weights=os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"),
                                  fasta=fasta, read_input=False, n=n_total, r=5000, c=-1, out=os.path.join(td, "o.blow5"),
                                  profile="dna-r10-prom", dwell_mean=None, dwell_std=0.0, noise_std=2.0, noise_sampling=True,
                                  duration_sampling=True, distr="expon", predict_batch_size=1024, export_every_n_samples=1000000,
                                  sample_rate=None, bps=None, digitisation=None, range_val=None, offset_mean=None, offset_std=None,
                                  median_before_mean=None, median_before_std=None, min_noise=0.0, min_duration=3, min_read_len=30,
                                  preserve_read_ids=False, seed=42, mode=mode)
                own = time.perf_counter() - t0
                first = m.first_global_chunk if hasattr(m, "first_global_chunk") else 0
                chunks = m.chunks_done - first
                size = sum(os.path.getsize(os.path.join(td, f)) for f in os.listdir(td))
                m.engine.close()
            except Exception as e:                      # this rank still meets the others at the barrier below
                err = f"rank {rank}: {type(e).__name__}: {e}"
            dist.barrier(group=grp)
            wall = time.perf_counter() - t0
        finally:
            if not keep or err:
                shutil.rmtree(td, ignore_errors=True)
        return own, wall, chunks, size, err, td
    first_pass = run(False)                # the first call pays the process's one-time costs (pinned buffers, thread pools)
    own, wall, chunks, size, err, td = run(True) if first_pass[4] is None else first_pass
    rows = [None] * world
    try:
        dist.all_gather_object(rows, (own, wall, float(chunks), float(size), float(cpu_share()), err, td), group=grp)
        errs = [r[5] for r in rows if r[5]]
        merged, command = None, None
        if rank == 0 and not errs:
            # the ONE file the reference leaves (inference.py:65-79): rank 0 joins the rank files while the others wait -- as the
            # parent of `predict --gpus N` does once its ranks have exited
            try:
                from seq2squiggle_amd.parallel import rank_output_path
                from seq2squiggle_amd.signal_io import merge_shards
                shards = [rank_output_path(os.path.join(rows[r][6], "o.blow5"), r, world) for r in range(world)]
                # (the other ranks wait at the barrier below: rank 0 may use the node's CPU share for the fill, as the command's parent does)
                threads = max(1, min(8, cpu_share() * int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))))
                t1 = time.perf_counter()
                n_rec = merge_shards(shards, os.path.join(td, "merged.blow5"), threads=threads, consume=True)
                merged = dict(merge_shards.last, seconds=time.perf_counter() - t1, records=n_rec)
            except Exception as e:
                merged = {"error": f"{type(e).__name__}: {e}"}
            shutil.rmtree(td, ignore_errors=True)              # (room for the command below)
            os.makedirs(td, exist_ok=True)
            command = one_command(mode, world, n_total, fasta, td)
            if "seconds" in command:                       # ... and with the join running WHILE the ranks compute (another record order)
                os.remove(os.path.join(td, "cmd.blow5"))
                command["join_live"] = one_command(mode, world, n_total, fasta, td, join="live")
        dist.barrier(group=grp)                                # nobody removes a rank file before rank 0 has joined them
    finally:
        shutil.rmtree(td, ignore_errors=True)
    dist.destroy_process_group(grp)
    if rank != 0:
        return None
    if errs:
        return {"error": errs}
    wall = max(r[1] for r in rows)
    total = sum(r[2] for r in rows)
    out = {"workload": f"example lambda genome -n {n_total} -r 5000 -> out.rankN.blow5 on {out_dir}: "
                       f"BASELINE configs[2]'s per-GPU share on each of {world} ranks, sharded by inference_run",
           "seconds": wall, "chunks": total, "chunks_per_sec": total / wall, "reads_per_sec": n_total / wall,
           "per_rank_seconds": [r[0] for r in rows], "per_rank_chunks": [r[2] for r in rows],
           "per_rank_cpu_share_threads": [int(r[4]) for r in rows], "output_bytes": sum(r[3] for r in rows),
           "local_world_size": int(os.environ.get("LOCAL_WORLD_SIZE", "1")),
           "includes": "per rank: FASTA parse, sampler skip-ahead to its shard, engine creation, chunking, kernels, export, D2H, "
                       "compression on cpu_share threads, file write"}
    if merged and "error" not in merged:
        out["merge_seconds"] = merged["seconds"]
        out["merge"] = {"bytes": merged.get("bytes"), "bytes_copied": merged.get("bytes_copied"), "threads": merged["threads"],
                        "engine": merged.get("engine"),
                        "gb_per_sec": merged.get("bytes", 0) / merged["seconds"] / 1e9, "records": merged["records"],
                        "remove_seconds": merged.get("remove_seconds"),
                        "how": "first rank file becomes the output, the record sections of the others move as byte ranges (tmpfs: the range "
                               "is preallocated, then filled through a shared mapping on several threads; elsewhere copy_file_range with one "
                               "writer: seq2squiggle_amd/merge.py), each rank file deleted as soon as it is in"}
        out["with_merge"] = {"seconds": wall + merged["seconds"], "chunks_per_sec": total / (wall + merged["seconds"]),
                             "reads_per_sec": n_total / (wall + merged["seconds"])}
    elif merged:
        out["merge"] = merged
    if command:
        out["one_command"] = command
        out["launch_seconds"] = command.get("launch_seconds")
        for c_ in (command, command.get("join_live", {})):
            if "seconds" in c_:
                c_["chunks_per_sec"] = total / c_["seconds"]
    return out


def one_command(mode, world, n_total, fasta, td, join="after"):
    """The command people run: `python -m seq2squiggle_amd predict <lambda> -n N -r 5000 -o OUT.blow5 --gpus <world>` as a child of
    rank 0, started while the bench's own ranks wait at a barrier (two processes per GPU for its duration): wall clock of the
    whole command with its own account of launch (spawn -> the slowest rank has the interpreter, torch and the library loaded),
    predict (-> the slowest rank has written its file) and merge seconds.  Skipped on a one-GPU rehearsal with more than two ranks
    (the pool allows six processes on a device)."""
    if os.environ.get("S2S_BENCH_ONE_GPU") and world > 2:
        return {"skipped": "one-GPU rehearsal: ranks of the bench + ranks of the command would exceed the pool's process limit"}
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                        "TORCHELASTIC_RUN_ID", "OMP_NUM_THREADS", "S2S_PINNED_CPUS", "S2S_PARENT_VISIBLE")
           and not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_"))}
    env["S2S_TIMING_JSON"] = os.path.join(td, "timing.json")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    if os.environ.get("S2S_BENCH_ONE_GPU"):
        env["S2S_ONE_GPU"] = "1"
    cmd = [sys.executable, "-m", "seq2squiggle_amd", "predict", fasta, "-n", str(n_total), "-r", "5000", "-o",
           os.path.join(td, "cmd.blow5"), "--gpus", str(world), "--seed", "42", "-m",
           os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"), "--compute-mode", mode, "-v", "warning"] + (["--join", join] if join != "after" else [])
    t0 = time.perf_counter()
    # its own process group: on a timeout the command AND the ranks it started are ended (by that group id), nothing is orphaned on a GPU
    with unpinned():                                              # (the command places its own ranks: it starts from this rank's ORIGINAL mask)
        proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        stdout, stderr = proc.communicate(timeout=110)            # (both joins fit the leg's 300-s barrier)
    except subprocess.TimeoutExpired:
        import signal
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.communicate()
        return {"error": "timed out after 110 s", "cmd": " ".join(cmd[1:])}
    seconds = time.perf_counter() - t0
    if proc.returncode != 0:
        return {"error": f"exit code {proc.returncode}: {(stderr or stdout)[-400:]}", "cmd": " ".join(cmd[1:])}
    out = {"cmd": "python " + " ".join(cmd[1:]).replace(ROOT + os.sep, "").replace(td + os.sep, "OUT/"), "seconds": seconds,
           "reads_per_sec": n_total / seconds, "output_bytes": os.path.getsize(os.path.join(td, "cmd.blow5"))}
    try:
        with open(env["S2S_TIMING_JSON"]) as f:
            out.update({k: v for k, v in json.load(f).items() if k.endswith("_seconds") or k in ("merge_bytes", "merge_bytes_copied", "reads", "join", "live_bytes", "join_order")})
    except (OSError, ValueError):
        pass
    return out


_PIN = None               # placement.pin_rank's record for this rank (None: not pinned)


class unpinned:
    """A child process inherits the affinity mask of the thread that starts it: children of a PINNED bench rank (the RCCL self-test,
    `predict --gpus N` with ranks and a placement of its own) must start from the mask this rank had before it bound itself."""

    def __enter__(self):
        self.back = None
        if _PIN and _PIN.get("allowed") and hasattr(os, "sched_setaffinity"):
            try:
                from seq2squiggle_amd.placement import parse_cpulist
                self.back = os.sched_getaffinity(0)
                os.sched_setaffinity(0, parse_cpulist(_PIN["allowed"]))
            except (OSError, ValueError):
                self.back = None
        return self

    def __exit__(self, *exc):
        if self.back is not None:
            try:
                os.sched_setaffinity(0, self.back)
            except OSError:
                pass
        return False


class HostGroup:
    """The ranks' meeting point: a gloo process group on host tensors (no GPU, no RCCL).  One rank: every call is a no-op."""

    def __init__(self, world, timeout_s=900):
        self.world, self.dist = world, None
        if world > 1 or os.environ.get("S2S_BENCH_FORCE_DIST"):
            from datetime import timedelta
            import torch.distributed as dist
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # one node: never resolve the container's hostname
            dist.init_process_group("gloo", timeout=timedelta(seconds=timeout_s))
            self.dist = dist

    @property
    def backend(self):
        return "gloo" if self.dist else None

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max(self, x):
        if not self.dist:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, obj):
        if not self.dist:
            return [obj]
        rows = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(rows, obj)
        return rows

    def size(self):
        return self.dist.get_world_size() if self.dist else 1

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


def device_identity(index):
    """What tells two GPUs apart: PCI address and uuid of device `index` as this process numbers it (None off the GPU)."""
    if index is None or not torch.cuda.is_available():
        return None
    p = torch.cuda.get_device_properties(index)
    out = {"index": index, "name": p.name}
    try:
        out["pci"] = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
    except AttributeError:
        out["pci"] = None
    try:
        out["uuid"] = str(p.uuid)
    except Exception:
        out["uuid"] = None
    return out


def devices_distinct(rows):
    """True when every rank reported another GPU (by PCI address, else uuid); None when nobody could tell."""
    keys = [(r or {}).get("pci") or (r or {}).get("uuid") for r in rows]
    if any(k is None for k in keys):
        return None
    return len(set(keys)) == len(keys)


def rccl_selftest_child():
    """One rank of the RCCL self-test (a fresh process that does nothing else): init_process_group("nccl"), one all_reduce of ones
    on its device; rank 0 prints {"ok", "sum", "init_seconds", "allreduce_seconds"}."""
    if os.environ.get("S2S_BENCH_SELFTEST_HANG"):            # (tests: a child that never comes back must be killed and reported)
        time.sleep(3600)
    import torch.distributed as dist
    from seq2squiggle_amd.placement import local_device
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", local_device())
    torch.cuda.set_device(dev)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", device_id=dev)
    t = torch.ones(1 << 20, device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ok = bool((t == float(world) ** 6).all().item())
    if rank == 0:
        print("RCCL_SELFTEST " + json.dumps({"ok": ok, "n_ranks": world, "init_and_first_seconds": t1 - t0,
                                             "allreduce_4MiB_ms": (t2 - t1) / 5 * 1e3}), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


def rccl_selftest(n_ranks, limit_s=120.0):
    """RCCL as a REPORTED pre-/post-flight, not a dependency: N throw-away children (fresh processes -- nothing of this process is
    re-used or re-exec'ed) each bind GPU r, bring up an RCCL communicator and all_reduce once.  Wall-limited: on expiry every
    child is killed (own session each, by process group) and the line says so.  -> {"ok", "n_ranks", "seconds", ...| "error"}."""
    import signal
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    from seq2squiggle_amd.placement import rank_visibility
    base = {k: v for k, v in os.environ.items()
            if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                         "OMP_NUM_THREADS", "S2S_PINNED_CPUS") and not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_"))}
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    one_gpu = bool(os.environ.get("S2S_BENCH_ONE_GPU"))
    t0 = time.perf_counter()
    procs = []
    try:
        for r in range(n_ranks):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            if one_gpu:
                env["S2S_ONE_GPU"] = "1"
            # (all devices stay visible to a child: RCCL's peer-to-peer transport wants to see its neighbours; the child picks LOCAL_RANK)
            with unpinned():
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rccl-selftest-child"], env=env, cwd=ROOT,
                                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True))
        deadline = time.perf_counter() + limit_s
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=max(0.1, deadline - time.perf_counter())))
            except subprocess.TimeoutExpired:
                raise TimeoutError
        seconds = time.perf_counter() - t0
        codes = [p.returncode for p in procs]
        line = [l for l in outs[0][0].splitlines() if l.startswith("RCCL_SELFTEST ")]
        if any(codes) or not line:
            bad = next((i for i, c in enumerate(codes) if c), 0)
            return {"ok": False, "n_ranks": n_ranks, "seconds": seconds, "exit_codes": codes,
                    "error": (outs[bad][1] or outs[bad][0]).strip()[-400:]}
        return dict(json.loads(line[0][len("RCCL_SELFTEST "):]), seconds=seconds)
    except TimeoutError:
        return {"ok": False, "n_ranks": n_ranks, "seconds": time.perf_counter() - t0,
                "error": f"no answer within {limit_s:g} s: the {n_ranks} self-test children were killed"}
    except Exception as e:
        return {"ok": False, "n_ranks": n_ranks, "seconds": time.perf_counter() - t0, "error": f"{type(e).__name__}: {e}"}
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
        for p in procs:
            try:
                p.communicate(timeout=30)
            except Exception:
                pass


WORKLOADS = {"config2": 1000, "config3": 12500}     # reads per GPU: BASELINE.json configs[1] / configs[2] (100,000 reads over 8 GPUs)


def launch_ranks(a, argv):
    """`--gpus N` outside torchrun: start the N ranks as children of this (GPU-untouched) process and relay rank 0's line."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + [x for x in argv if x != "--dry-launch"]
    if a.dry_launch:
        print(json.dumps({"dry_launch": cmd}))
        return 0
    import signal
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the launcher and its ranks in a session of their own: SIGTERM / SIGHUP / Ctrl-C to this parent ends all of them (by process
    # group), nothing stays behind on a GPU
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)

    def end(signum, frame):
        try:
            os.killpg(p.pid, signal.SIGTERM)
            try:
                p.wait(timeout=15)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        sys.exit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
        signal.signal(sig, end)
    out, _ = p.communicate()
    for line in out.splitlines():
        if line.startswith("{"):
            print(line)
        elif line.strip():
            print(line, file=sys.stderr)
    return p.returncode


class Line:
    """Rank 0's ONE JSON line: built up as the legs finish and written exactly once -- by the end of main(), or by the deadline
    thread if a secondary leg (anything after the timed region) has not come back in time: the measurement is never lost to a leg."""

    def __init__(self, fd):
        import threading
        self.fd, self.out, self.lock, self.done, self.timer = fd, None, threading.Lock(), False, None

    def arm(self, out, seconds):
        import threading
        self.out = out
        self.timer = threading.Timer(seconds, self._expired, args=(seconds,))
        self.timer.daemon = True
        self.timer.start()

    def _expired(self, seconds):
        with self.lock:
            if self.done:
                return
            self.out["secondary_legs_timed_out"] = {"after_seconds": seconds, "note": "the headline fields were complete; a later leg did not "
                                                    "return -- the line was written by the deadline thread and the process ended"}
            self._write()
        os._exit(0)

    def _write(self):
        self.done = True
        sys.stdout.flush()
        os.write(self.fd, (json.dumps(self.out) + "\n").encode())

    def write(self, out=None):
        with self.lock:
            if self.done:
                return
            if out is not None:
                self.out = out
            if self.timer:
                self.timer.cancel()
            self._write()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS),
                    help="reads per GPU and step: config2 = 1000 x 5 kb (BASELINE configs[1]), config3 = 12500 x 5 kb "
                         "(configs[2]: each GPU's share of 100,000 reads over 8)")
    ap.add_argument("--reads", type=int, default=None, help="override the workload's reads per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-launch", action="store_true", help="print the child command --gpus N would start, run nothing")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="CPU-only pass through the multi-rank plumbing of the line (host group, device identities, RCCL self-test, "
                         "cpu_baseline on rank 0 while the others wait): no GPU work")
    ap.add_argument("--rccl-selftest-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rccl-selftest-only", type=int, default=0, metavar="N", help="run only the RCCL self-test with N children and print its result")
    ap.add_argument("--rccl-selftest-limit", type=float, default=float(os.environ.get("S2S_RCCL_SELFTEST_LIMIT", "120")),
                    help="wall limit of the RCCL self-test in seconds (0 = skip it)")
    ap.add_argument("--leg-deadline", type=float, default=float(os.environ.get("S2S_BENCH_LEG_DEADLINE", "1500")),
                    help="seconds after the timed region by which the line is written, whatever the secondary legs are doing")
    ap.add_argument("--mode", default="f16x3", choices=["f32", "f16x3", "f16"],
                    help="decoder arithmetic: f32-input MFMA, or split-f16 (3 f16 MFMA products, fp32 accumulate)")
    a = ap.parse_args()
    if a.rccl_selftest_child:
        return rccl_selftest_child()
    if a.rccl_selftest_only:                 # the self-test alone (this process stays off the GPU): proves the bring-up code on whatever is there
        print(json.dumps({"rccl_selftest": rccl_selftest(a.rccl_selftest_only, a.rccl_selftest_limit)}))
        return
    n_reads = a.reads if a.reads is not None else WORKLOADS[a.workload]

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.dry_launch):
        sys.exit(launch_ranks(a, sys.argv[1:]))            # nothing above this line has initialised the GPU
    if a.gpus != world:
        sys.exit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # stdout carries ONE JSON line: everything else that writes to file descriptor 1 from here on (RCCL's version banner, library
    # chatter) goes to stderr; the line itself is written to the saved descriptor at the end
    sys.stdout.flush()
    line = Line(os.dup(1))
    os.dup2(2, 1)
    one_gpu = bool(os.environ.get("S2S_BENCH_ONE_GPU"))          # rehearsal on a 1-GPU box: every rank on cuda:0
    if one_gpu:
        os.environ["S2S_ONE_GPU"] = "1"                         # (placement.local_device, inference_run)
    from seq2squiggle_amd import placement
    global _PIN
    pinned = _PIN = placement.pin_rank() if world > 1 else None  # before the first GPU call and the first pinned page
    local = placement.local_device()
    host = HostGroup(world)                                      # gloo on host tensors: the bench needs no RCCL (the path has no exchange)
    selftest_mode = a.launch_selftest
    if not selftest_mode:
        torch.cuda.set_device(local)
    devices = host.gather(dict(device_identity(None if selftest_mode else local) or {}, rank=rank, pid=os.getpid(),
                               cpus=(pinned or {}).get("cpus"), numa_node=(pinned or {}).get("numa_node")))
    distinct = devices_distinct(devices)

    def multi_rank_fields(out):
        out["barrier_backend"] = host.backend
        out["per_rank_device"] = devices
        out["devices_distinct"] = distinct

    def rccl_leg():
        """Rank 0 runs the self-test's children while the others wait at the host barrier."""
        res = None
        if rank == 0 and world > 1 and a.rccl_selftest_limit > 0:
            if one_gpu and world > 2:
                # (the pool allows six processes on a device: N ranks + N self-test children would be 2 N)
                res = {"skipped": "one-GPU rehearsal with more than two ranks: the self-test's children would exceed the pool's process limit"}
            else:
                res = rccl_selftest(world, a.rccl_selftest_limit)
        host.barrier()
        return res

    if selftest_mode:
        # the multi-rank plumbing of the line without a GPU: the same host group, gather, self-test and rank-0-only leg as the real run
        t0 = time.perf_counter()
        host.barrier()
        el = host.max(time.perf_counter() - t0)
        ones = host.gather(1)
        out = {"launch_selftest": True, "n_gpus": a.gpus, "n_ranks_seen": host.size(), "sum_of_ones": int(sum(ones)), "barrier_seconds": el}
        multi_rank_fields(out)
        if rank == 0:
            line.arm(out, a.leg_deadline)
        res = rccl_leg()
        if rank == 0:
            out["rccl_selftest"] = res
            if not a.no_cpu_baseline:
                sd, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"))
                out["cpu_baseline"] = cpu_baseline(sd, cfg, None, seconds_target=0.2, sample_chunks=16, whole_node=(pinned or {}).get("allowed"))
        host.barrier()
        if rank == 0:
            line.write()
        host.close()
        return

    sd, cfg = S.load_checkpoint(os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"))
    eng = S.Engine(sd, cfg, device=local, mode=a.mode)
    dev = eng.device
    reads = make_reads(n_reads, 1234 + rank)               # every rank: its own shard of the read set
    bases, nv, first = S.encode_reads(reads, cfg["seq_kmer"])
    B = bases.shape[0]
    bases_d, nv_d = torch.from_numpy(bases).to(dev), torch.from_numpy(nv).to(dev)
    sig = torch.empty(B, 250, dtype=torch.float32, device=dev)
    dur = torch.empty(B, 16, dtype=torch.int32, device=dev)
    params = S.PredictParams(seed=42)                      # defaults: noise_std 2.0, samplers on, min_duration 3
    first_chunk = rank * B

    def step():
        eng.predict_chunks(bases_d, nv_d, params, first_global_chunk=first_chunk, out_signal=sig, out_dur=dur)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    eng.stats()                                            # reset: the counters below are those of the timed launches
    eng.set_profiling(True)
    host.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0                         # this rank's own time, before waiting for the others
    host.barrier()
    el = time.perf_counter() - t0
    eng.set_profiling(False)
    dec_ms, dec_launches, dec_chunks = eng.kernel_ms()
    live = eng.stats()
    el = host.max(el)
    ranks_seen = host.size()
    per_rank = [float(x) for x in host.gather(B * a.steps / own)]
    emitted = int((sig != 0).sum().item())
    # the reference's default batch (1024 chunks per call), reported for transparency only; skipped with
    # --no-cpu-baseline so that a rocprofv3 --stats of that command averages the bench launches alone
    small_rate = None
    if world == 1 and not a.no_cpu_baseline:
        small, reps = 1024, 200
        b1, n1, s1, d1 = bases_d[:small].contiguous(), nv_d[:small].contiguous(), sig[:small], dur[:small]
        for i in range(reps + 3):
            if i == 3:                                     # (three untimed calls first)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            eng.predict_chunks(b1, n1, params, first_global_chunk=first_chunk, out_signal=s1, out_dur=d1)
        torch.cuda.synchronize()
        small_rate = reps * small / (time.perf_counter() - t1)

    out = None
    if rank == 0:
        chunks_total = B * a.steps * world
        chunks_s = chunks_total / el
        tflops = FLOP_PER_CHUNK_DOMINANT * dec_chunks / (dec_ms * 1e-3) / 1e12 if dec_ms > 0 else None
        cpl = dec_chunks / dec_launches if dec_launches else None
        pmc = pmc_counters(a.mode, cpl) if cpl else pmc_counters(a.mode, 0)
        peak = PEAK[a.mode]
        roof = {"bound": "mfma", "kernel": DOMINANT_KERNEL, "achieved": tflops, "peak": peak, "unit": "TFLOP/s",
                "frac": (tflops / peak) if tflops else None, "traffic": pmc["traffic"],
                "traffic_unit": "bytes per launch (PMC, separate profiled run)", "traffic_source": pmc["source"],
                "mfma_busy": pmc["mfma_busy"], "valu_issue": pmc["valu_issue"],
                # the clock the chip held: GRBM_GUI_ACTIVE / 8 XCDs / kernel wall of the PMC pass (profiled runs clock lower), and the
                # in-kernel s_memtime / s_memrealtime ratio of the un-profiled diagnostic build (cycles per chunk and CU beside it)
                "effective_clock_ghz": pmc["effective_clock_ghz"], "in_kernel_clock": pmc["in_kernel_clock"],
                # LIVE, from the kernel's own counters over exactly the timed launches of this run (s2s_stats_read):
                "live": {"in_kernel_clock_ghz": live["in_kernel_clock_ghz"], "cycles_per_chunk_and_cu": live["cycles_per_chunk_and_cu"],
                         "softmax_redo_rate": live["redo_rate"], "softmax_runs": live["softmax_runs"],
                         "source": "s_memtime / s_memrealtime of one thread per workgroup; redo counter of the fast softmax path"},
                "pmc_note": "traffic, mfma_busy, valu_issue, effective_clock_ghz, in_kernel_clock: constants from the committed profiled passes "
                            "named in traffic_source (another run, another device); `live` is this run",
                "algorithmic_bytes_per_launch": 1088 * cpl if cpl else None,
                "peak_note": {"f32": "f32-input MFMA peak (MI355X_MICROARCH.md)"}.get(
                    a.mode, "dense f16 MFMA peak (MI355X_MICROARCH.md); achieved counts ALGORITHMIC flops"),
                "flop_per_chunk": FLOP_PER_CHUNK_DOMINANT,
                "avg_launch_ms": dec_ms / dec_launches if dec_launches else None,
                "launches": dec_launches, "chunks_per_launch": cpl}
        roof.update(pmc.get("extra", {}))
        if a.mode == "f16x3" and tflops:
            # secondary readings: every algorithmic product costs three f16 MFMA products in this arithmetic, and the
            # native exact-f32 alternative is the f32-input MFMA
            roof["frac_of_f16x3_ceiling"] = tflops / (PEAK_F16_MFMA_TFLOPS / 3.0)
            roof["x_of_f32_mfma_peak"] = tflops / PEAK_F32_MFMA_TFLOPS
        out = {
            "metric": "signal samples/sec at 5 kb reads (padded [chunks x 250] samples the predict path emits)",
            "value": chunks_s * 250, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "f16": "f16"}.get(a.mode, "f16x3"),
            "dtype_note": {"f32": "f32-input MFMA",
                           "f16": "REDUCED PRECISION (outside the 1e-4 pA parity bound): decoder operands rounded to f16 once, one "
                                  "f16 MFMA product per product, fp32 accumulate; frontend f16x3"}.get(
                a.mode, "every operand split into two f16 halves, three f16 MFMA products per product, fp32 accumulate: "
                        "fp32-class accuracy, see cpu_baseline.parity"),
            "data": "synthetic",
            "config": {"mode": a.mode, "workload": f"{n_reads} synthetic reads x {READ_LEN} nt per GPU ({B} chunks/step/GPU), "
                                   "default noise+duration samplers, synthetic k=9 checkpoint",
                       "chunks_per_step_per_gpu": B, "profile": "dna-r10-prom", "seed": 42},
            "n_ranks_seen": ranks_seen, "per_rank_chunks_per_sec": per_rank,
            "reads_per_sec": chunks_s / CHUNKS_PER_READ, "chunks_per_sec": chunks_s,
            "emitted_samples_per_sec": emitted * world / (el / a.steps),
            "chunks_per_sec_at_reference_batch_1024": small_rate,
            "roofline": roof,
        }
        multi_rank_fields(out)
        if one_gpu:
            out["one_gpu_rehearsal"] = "S2S_BENCH_ONE_GPU: all ranks share cuda:0 -- NOT a scaling measurement"
        elif world > 1 and distinct is False:
            out["invalid"] = "two ranks reported the same GPU: this is not an N-GPU measurement"
            print("bench.py: " + out["invalid"], file=sys.stderr)
        # from here on the line exists: whatever the secondary legs do, it is written by the deadline at the latest
        line.arm(out, a.leg_deadline)

    def leg(name, fn, *args):                    # a secondary leg that fails (a full disk, ...) must not cost the line
        try:
            out[name] = fn(*args)
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}
            print(f"bench.py: {name} failed: {type(e).__name__}: {e}", file=sys.stderr)

    sharded_failed = False
    if host.dist and not a.no_cpu_baseline:
        sharded = None
        try:
            sharded = end_to_end_sharded(a.mode, host.dist, rank, world, dev)
        except Exception as e:                                 # (a lost rank, a gloo timeout): the headline line still goes out
            sharded = {"error": [f"rank {rank}: {type(e).__name__}: {e}"]}
            sharded_failed = True
        if rank == 0 and sharded:
            out["end_to_end_sharded"] = sharded
            chunks_s = out["chunks_per_sec"]
            if "chunks_per_sec" in sharded:
                sharded["of_resident_rate"] = sharded["chunks_per_sec"] / chunks_s
                for part in (sharded.get("with_merge", {}), sharded.get("one_command", {}), sharded.get("one_command", {}).get("join_live", {})):
                    if "chunks_per_sec" in part:                    # ... with the merge into ONE file, and as the one command with its launch
                        part["of_resident_rate"] = part["chunks_per_sec"] / chunks_s
    if sharded_failed:                                         # the groups are in an unknown state: no further collective
        if rank == 0:
            line.write()
        os._exit(0)
    if rank == 0 and not a.no_cpu_baseline:
        if world == 1:
            leg("end_to_end", end_to_end, a.mode)
        # (N > 1: rank 0 alone, on the node's whole CPU share, while the others wait at the host barrier below)
        leg("cpu_baseline", lambda: cpu_baseline(sd, cfg, eng, whole_node=(pinned or {}).get("allowed") if world > 1 else None))
        if "value" in out["cpu_baseline"]:
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            out["gpu_over_cpu_note"] = "whole job over ONE host's CPU baseline" if world > 1 else None
        if world == 1:
            leg("power", power_leg, eng, bases_d, nv_d, sig, dur, params)
            leg("config4_k6", config4_k6_leg, a.mode, dev, a.steps)
            if a.mode != "f32":
                leg("weight_sensitivity", weight_sensitivity_leg, a.mode, bases_d, nv_d, sig, dur, params, max(2, min(a.steps, 5)))
            if a.mode == "f16x3":
                leg("reduced_precision", reduced_precision_leg, sd, cfg, bases_d, nv_d, sig, dur, params, a.steps)
    selftest_clean = 1.0
    try:
        host.barrier()                                         # (N > 1: the others waited here while rank 0 ran its legs)
        if world > 1:
            # RCCL, reported: the LAST thing that touches the GPUs, in processes of their own (rank 0 starts them, the others wait at the
            # host barrier inside rccl_leg): whatever it does to a device, every other field of the line has been measured by now
            try:
                res = rccl_leg()
            except Exception as e:
                res = {"ok": False, "error": f"{type(e).__name__}: {e}"}
            if rank == 0:
                out["rccl_selftest"] = res
                selftest_clean = 0.0 if (res and res.get("ok") is False and "killed" in str(res.get("error", ""))) else 1.0
            selftest_clean = -host.max(-selftest_clean)         # (min over ranks: rank 0's verdict reaches everybody)
    except Exception as e:                                     # (rank 0 took longer than the group's timeout: the others just leave)
        print(f"bench.py: rank {rank}: final barrier: {type(e).__name__}: {e}", file=sys.stderr)
        if rank != 0:
            os._exit(0)
    if rank == 0:
        line.write()
    if selftest_clean < 1.0:
        # self-test children had to be killed: a device may be wedged under them -- leave without a teardown that could wait for it
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3 if (rank == 0 and out.get("invalid")) else 0)
    eng.close()
    host.close()
    if rank == 0 and out.get("invalid"):
        sys.exit(3)


if __name__ == "__main__":
    main()
