"""GPU tests of the BASELINE.json configurations that round 1 left unexercised.

configs[4] ("synthetic 100 Mb reference -c 30 -r 10000, pod5 out"): coverage-mode sampling (reference
utils.py:495-582: seq_num = round(c * total_len / r)), 10 kb reads (~625 chunks per read), a multi-contig reference and
a POD5 stream of hundreds of reads -- on a reference scaled to 0.3 Mb so that the test finishes in about a minute; every
quantity checked is size-independent.

configs[3] ("dna_r9_min profile -n 100000 -r 8000 --noise-std 1.5"): the k = 6 chemistry at its full per-GPU size class
(>= 300 k chunks in one call) with the properties test_full_size_properties checks for k = 9, plus the CLI-level flow."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import seq2squiggle_amd as S
from seq2squiggle_amd import pod5_io, signal_io
from seq2squiggle_amd import utils as U
from seq2squiggle_amd.cli import set_config
from seq2squiggle_amd.inference import inference_run
from conftest import GOLDEN, load_ckpt

pytestmark = pytest.mark.gpu
LUT = np.frombuffer(b"ACGT", dtype=np.uint8)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


write_synthetic_reference = U.write_synthetic_reference


def _run(out, fasta, profile, ckpt, *, n=-1, r=1000, c=-1, noise_std=2.0, seed=42, streaming=True, read_input=False):
    U.set_seeds(seed)
    m = inference_run(config=set_config(None), saved_weights=os.path.join(GOLDEN, ckpt), fasta=str(fasta), read_input=read_input,
                      n=n, r=r, c=c, out=str(out), profile=profile, dwell_mean=None, dwell_std=0.0, noise_std=noise_std,
                      noise_sampling=True, duration_sampling=True, distr="expon", predict_batch_size=1024,
                      export_every_n_samples=1000000, sample_rate=None, bps=None, digitisation=None, range_val=None,
                      offset_mean=None, offset_std=None, median_before_mean=None, median_before_std=None, min_noise=0.0,
                      min_duration=3, min_read_len=30, preserve_read_ids=False, seed=seed, streaming=streaming)
    chunks = m.chunks_done
    m.engine.close()
    return chunks


@pytest.fixture(scope="module")
def synthetic_reference(tmp_path_factory):
    """i.i.d. ACGT, rng 1234, two contigs of unequal length, a few N runs (utils.py:402-403 path)."""
    rng = np.random.default_rng(1234)
    path = tmp_path_factory.mktemp("ref") / "synthetic_ref.fasta"
    lens = (180_000, 120_000)
    with open(path, "w") as f:
        for i, L in enumerate(lens):
            seq = bytearray(LUT[rng.integers(0, 4, L)].tobytes())
            seq[1000:1010] = b"N" * 10
            f.write(f">contig{i}\n")
            for lo in range(0, L, 80):
                f.write(seq[lo:lo + 80].decode() + "\n")
    return path, sum(lens)


def test_config5_coverage_mode_10kb_reads_pod5_stream(tmp_path, synthetic_reference):
    fasta, total = synthetic_reference
    c, r = 30, 10000
    n_expected = round(c * total / r)                                  # utils.py:509 (seq_num in coverage mode)
    chunks = _run(tmp_path / "a.pod5", fasta, "dna-r10-prom", "synthetic_k9.ckpt", c=c, r=r)
    a = pod5_io.read_pod5(str(tmp_path / "a.pod5"))
    assert len(a["reads"]) == n_expected == 900
    lens = np.array([len(x["signal"]) for x in a["reads"]])
    assert chunks > 400_000 and chunks / n_expected > 450               # ~10 kb reads: hundreds of chunks per read
    assert lens.min() > 0 and lens.sum() > 200 * chunks                 # dwell ~9-12 samples per k-mer, 16 k-mers per chunk
    assert [x["read_number"] for x in a["reads"]] == list(range(n_expected))
    assert len({x["read_id"] for x in a["reads"]}) == n_expected
    # long reads span several signal-table rows (102,400 samples each): the row lists must tile every read exactly
    assert a["signal_rows"] > n_expected and lens.max() > pod5_io.SIGNAL_CHUNK

    # the same command into BLOW5: same read set, same samples (container-independent), read by read
    chunks_b = _run(tmp_path / "b.blow5", fasta, "dna-r10-prom", "synthetic_k9.ckpt", c=c, r=r)
    _, recs = signal_io.read_blow5(str(tmp_path / "b.blow5"))
    assert chunks_b == chunks and len(recs) == n_expected
    for p, b in zip(a["reads"], recs):
        assert len(p["signal"]) == b["len_raw_signal"]
        assert int(p["signal"].astype(np.int64).sum()) == int(b["signal"].astype(np.int64).sum())
    assert all(np.array_equal(p["signal"], b["signal"]) for p, b in zip(a["reads"][::37], recs[::37]))

    # determinism: a second run of the same command writes the same samples and calibration
    _run(tmp_path / "a2.pod5", fasta, "dna-r10-prom", "synthetic_k9.ckpt", c=c, r=r)
    a2 = pod5_io.read_pod5(str(tmp_path / "a2.pod5"))
    assert len(a2["reads"]) == n_expected
    for x, y in zip(a["reads"], a2["reads"]):
        assert np.array_equal(x["signal"], y["signal"]) and x["calibration_offset"] == y["calibration_offset"]


def test_config4_k6_full_size_properties():
    """>= 300 k chunks of 8 kb reads through the k = 6 checkpoint with the dna-r9-min scalars (dwell_mean 4000/450,
    noise_std 1.5): determinism, batch-split invariance of the counter-based RNG, strip/offset bookkeeping, DAC range."""
    sd, cfg = load_ckpt("k6")
    assert cfg["seq_kmer"] == 6
    eng = S.Engine(sd, cfg, mode="f16x3")
    prof = U.get_profile("dna-r9-min")
    rng = np.random.default_rng(4321)
    reads = [LUT[rng.integers(0, 4, 8000)].tobytes().decode() for _ in range(610)]
    bases, nv, first = S.encode_reads(reads, 6)
    B = bases.shape[0]
    assert B == 610 * 500 and bases.shape[1] == 21                      # ceil((8000-6+1)/16) = 500 chunks per read
    b, n = torch.from_numpy(bases).cuda(), torch.from_numpy(nv).cuda()
    pp = S.PredictParams(dwell_mean=prof["sample_rate"] / prof["bps"], noise_std=1.5, seed=42)
    a = eng.predict_chunks(b, n, pp)
    a_sig, a_dur = a["signal"].clone(), a["dur"].clone()
    again = eng.predict_chunks(b, n, pp)
    assert torch.equal(a_sig, again["signal"]) and torch.equal(a_dur, again["dur"])
    cut = 123457
    lo = eng.predict_chunks(b[:cut].contiguous(), n[:cut].contiguous(), pp)
    hi = eng.predict_chunks(b[cut:].contiguous(), n[cut:].contiguous(), pp, first_global_chunk=cut)
    assert torch.equal(a_sig, torch.cat([lo["signal"], hi["signal"]])) and torch.equal(a_dur, torch.cat([lo["dur"], hi["dur"]]))
    assert torch.isfinite(a_sig).all() and (a_sig >= 0).all() and (a_dur >= 3).all()
    # the last chunk of every read is ragged (7995 k-mers = 499 * 16 + 11): its pad k-mers still get a dwell
    assert int(nv.reshape(610, 500)[:, -1].max()) == 11 and int(nv.reshape(610, 500)[:, :-1].min()) == 16
    # noise path at noise_std 1.5: a noise-free run differs exactly where the clean signal is non-zero
    clean = eng.predict_chunks(b[:4096].contiguous(), n[:4096].contiguous(),
                               S.PredictParams(dwell_mean=pp.dwell_mean, noise_std=0.0, seed=42))
    assert torch.equal(clean["dur"], a_dur[:4096])
    assert torch.equal(clean["signal"] == 0, (clean["signal"] == 0) & (a_sig[:4096] == 0))
    # (only on live rows: past cum[15] the length regulator pads sigma with zeros, so sd = 0 there -- modules.py:386-388, model.py:227-232)
    changed = (clean["signal"] != a_sig[:4096])
    live = torch.arange(250, device="cuda")[None, :] < a_dur[:4096].sum(1, keepdim=True)
    assert 0.3 < float(live.float().mean()) < 0.9                       # dwell ~ 9: a chunk fills ~ 140 of its 250 rows
    assert float(changed[live & (clean["signal"] != 0)].float().mean()) > 0.99
    assert not bool(changed[~live].any())
    nz = (a_sig != 0).sum(1)
    ex = eng.export_reads(a_sig, torch.from_numpy(first).cuda(), prof["digitisation"], prof["range"], prof["offset_mean"],
                          want_pa=True, want_dac=True)
    offs = ex["offsets"].cpu().numpy()
    assert np.array_equal(np.diff(offs), nz.cpu().numpy().reshape(610, 500).sum(1)) and offs[-1] == int(nz.sum())
    pa = ex["pa"][: offs[-1]]
    assert torch.equal(pa, a_sig[a_sig != 0])
    dac = ex["dac"][: offs[-1]].cpu().numpy()
    ref = signal_io.signal_to_dac(pa[:200000].cpu().numpy(), prof["digitisation"], prof["range"], prof["offset_mean"], False)
    assert np.array_equal(dac[:200000], ref)                            # signal_io.py:134-141 on the r9 calibration
    eng.close()


def test_config4_cli_flow_k6(tmp_path):
    """lambda genome -n 60 -r 8000 --profile dna-r9-min --noise-std 1.5: streaming BLOW5 equals the reference-shaped
    predict_step / export_and_clear_results flow, and the profile's kit / calibration reach the file."""
    lam = os.path.join(GOLDEN, "example_lambda_genome.fasta")
    outs = []
    for streaming in (True, False):
        out = tmp_path / f"s{int(streaming)}.blow5"
        _run(out, lam, "dna-r9-min", "synthetic_k6.ckpt", n=60, r=8000, noise_std=1.5, seed=7, streaming=streaming)
        outs.append(signal_io.read_blow5(str(out)))
    (ha, a), (hb, b) = outs
    assert "FLO-MIN110" in ha and "SQK-LSK109" in ha and len(a) == len(b) == 60
    for x, y in zip(a, b):
        assert x["read_id"] == y["read_id"] and np.array_equal(x["signal"], y["signal"])
        assert x["digitisation"] == 8192.0 and x["sampling_rate"] == 4000.0 and abs(x["range"] - 1443.030273) < 1e-6


def _cli_child(args, timeout=1500):
    """`python -m seq2squiggle_amd predict ...` as a child process -> (seconds, its peak RSS in MB)."""
    import resource
    import time
    before = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "seq2squiggle_amd", "predict"] + [str(a) for a in args], cwd=ROOT,
                       capture_output=True, text=True, timeout=timeout)
    el = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    peak = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss          # KB; the largest child so far
    return el, peak / 1024.0, peak > before


def test_config5_quarter_of_one_gpu_share(tmp_path):
    """BASELINE.json configs[4] ("synthetic 100 Mb reference -c 30 -r 10000, pod5 out, 8 GPUs") at ONE QUARTER of one GPU's share:
    the 300,000 reads of the full job are 37,500 per GPU; here a 3.125 Mb reference of 5 unequal contigs (rng 1234) with
    `-c 30 -r 10000` gives round(30 * 3,125,000 / 10,000) = 9,375 reads (reference utils.py:509) ~ 5.9 M chunks ~ 1.1 G
    samples through the CLI, streamed into .pod5 (VBZ rows) by run_streaming.  Checked: the read count, every read tiled
    exactly by its signal-table rows, per-read sample counts AND sums equal to the same command written as .blow5, and the
    host's peak RSS bounded well below the data volume (the reference keeps every read in RAM until one final save(),
    inference.py:72-79 -- at this size 2.3 GB of int16 plus the fp32 tensors it was made from)."""
    fasta = tmp_path / "ref.fasta"
    lens = (1_000_000, 750_000, 625_000, 500_000, 250_000)
    total = write_synthetic_reference(fasta, lens)
    assert total == 3_125_000
    c, r = 30, 10000
    n_expected = round(c * total / r)
    assert n_expected == 9375
    ckpt = os.path.join(GOLDEN, "synthetic_k9.ckpt")
    common = [fasta, "-c", c, "-r", r, "-m", ckpt, "--seed", 42, "-v", "warning"]
    # RSS baseline: the same program on a job 1/100 the size (interpreter, torch, HIP runtime, pinned staging, the reference)
    _, rss_small, _ = _cli_child([fasta, "-n", 94, "-r", r, "-m", ckpt, "--seed", 42, "-v", "warning", "-o", tmp_path / "small.pod5"])
    t_pod5, rss_pod5, fresh = _cli_child(common + ["-o", tmp_path / "a.pod5"])
    t_blow5, _, _ = _cli_child(common + ["-o", tmp_path / "b.blow5"])
    size_pod5, size_blow5 = os.path.getsize(tmp_path / "a.pod5"), os.path.getsize(tmp_path / "b.blow5")
    n = samples = rows = 0
    blow = signal_io.iter_blow5(str(tmp_path / "b.blow5"))
    next(blow)                                                              # header text
    longest = 0
    for (row, raw), b in zip(pod5_io.iter_pod5(str(tmp_path / "a.pod5")), blow):
        assert row["read_number"] == n == b["read_number"] and str(row["read_id"]) == b["read_id"]
        # the rows of a read tile it exactly: full rows of SIGNAL_CHUNK samples, then the remainder
        assert len(raw) == row["num_samples"] == b["len_raw_signal"] > 0
        assert len(row["signal"]) == -(-len(raw) // pod5_io.SIGNAL_CHUNK)
        assert list(row["signal"]) == list(range(rows, rows + len(row["signal"])))
        assert int(raw.astype(np.int64).sum()) == int(b["signal"].astype(np.int64).sum())
        if n % 97 == 0:
            assert np.array_equal(raw, b["signal"])
        rows += len(row["signal"])
        samples += len(raw)
        longest = max(longest, len(raw))
        n += 1
    assert n == n_expected and next(blow, None) is None
    chunks_lo = n * 450
    assert samples > 150 * chunks_lo and longest > 3 * pod5_io.SIGNAL_CHUNK   # ~10 kb reads: several rows each, the longest many
    raw_bytes = 2 * samples
    assert size_pod5 < 0.75 * raw_bytes and size_blow5 < 0.85 * raw_bytes      # VBZ ~ 1.25 B/sample, zlib records ~ 1.55
    # bounded memory: the big job may cost at most 1.2 GB more than the tiny one (its output alone is ~1.4 GB, the int16 samples
    # 2.3 GB) -- reads are sampled lazily, super-batches are recycled, the POD5 writer keeps ~100 B per read until close()
    print(f"config5/4: {n} reads, {samples} samples, pod5 {size_pod5 / 1e9:.2f} GB in {t_pod5:.1f} s, blow5 {size_blow5 / 1e9:.2f} GB in "
          f"{t_blow5:.1f} s, peak RSS {rss_pod5:.0f} MB (small job {rss_small:.0f} MB)")
    if fresh:
        assert rss_pod5 < rss_small + 1200, (rss_pod5, rss_small)
    assert rss_pod5 < 0.5 * raw_bytes / 2 ** 20 + rss_small


def test_config1_exact_command(tmp_path):
    """BASELINE.json configs[1], the configuration the headline metric is quoted on, as the command itself:
    `seq2squiggle predict example/lamda_genome.fasta -n 1000 -r 5000` (default samplers, seed 42).  1000 reads and 286,622
    chunks (the figure SURVEY 8d-M3 measured by importing the reference's sampler), every read present, and two runs write the
    same samples and the same record draws."""
    lam = os.path.join(GOLDEN, "example_lambda_genome.fasta")
    outs = []
    for i in range(2):
        chunks = _run(tmp_path / f"c1_{i}.blow5", lam, "dna-r10-prom", "synthetic_k9.ckpt", n=1000, r=5000, seed=42)
        assert chunks == 286_622
        outs.append(signal_io.read_blow5(str(tmp_path / f"c1_{i}.blow5"))[1])
    a, b = outs
    assert len(a) == len(b) == 1000 and [x["read_number"] for x in a] == list(range(1000))
    assert len({x["read_id"] for x in a}) == 1000
    for x, y in zip(a, b):
        assert np.array_equal(x["signal"], y["signal"]) and x["offset"] == y["offset"] and x["median_before"] == y["median_before"]
    n_samples = sum(len(x["signal"]) for x in a)
    assert 150 * 286_622 < n_samples < 250 * 286_622
