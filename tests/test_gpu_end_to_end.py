"""GPU end-to-end: model object + batching + export + writer + CLI against reference-derived expectations.

Deterministic configuration (BASELINE.json configs[0]: example/test.fasta --read-input, samplers off, ideal
dwell): the reference's own predict_step output for these chunks is the golden `y_ideal`; per read the expected
file content is zero-strip + DAC of those rows (both pinned separately against the reference)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import seq2squiggle_amd as S
from seq2squiggle_amd import signal_io
from seq2squiggle_amd import utils as U
from seq2squiggle_amd.inference import inference_run, iter_batches
from seq2squiggle_amd.model import seq2squiggle
from oracle import s2s_oracle as O
from conftest import GOLDEN, ROOT, load_npz

pytestmark = pytest.mark.gpu
FASTA = os.path.join(GOLDEN, "example_test.fasta")


def expected_records(tag, profile_name):
    g = load_npz(f"stages_{tag}.npz")
    names = [str(n) for n in g["names"]]
    p = U.get_profile(profile_name)
    out = {}
    for rid in dict.fromkeys(names):
        rows = [torch.from_numpy(g["y_ideal"][i]) for i, n in enumerate(names) if n == rid]
        pa = O.strip_zeros(rows).numpy()
        out[rid] = (pa, O.to_dac(pa, p["digitisation"], p["range"], p["offset_mean"]))
    return out


def check_file(path, exp, ids):
    _, recs = (signal_io.read_slow5 if str(path).endswith(".slow5") else signal_io.read_blow5)(str(path))
    assert [r["read_id"] for r in recs] == ids
    for r in recs:
        pa, dac = exp[r["read_id"]]
        assert r["len_raw_signal"] == len(dac)
        # |pA error| < 2e-3 -> at most one DAC step of difference, and only at rounding ties
        d = np.abs(r["signal"].astype(np.int32) - dac.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() < 0.01


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
@pytest.mark.parametrize("tag,profile_name", [("k9", "dna-r10-prom"), ("k6", "dna-r9-min")])
def test_model_object_export_flow(tmp_path, mode, tag, profile_name):
    exp = expected_records(tag, profile_name)
    reads = [(s, n) for s, n in U.read_fasta(FASTA)]
    w = signal_io.BLOW5Writer(str(tmp_path / "o.blow5"), U.get_profile(profile_name), True, profile_name, True)
    m = seq2squiggle.load_from_checkpoint(os.path.join(GOLDEN, f"synthetic_{tag}.ckpt"), out_writer=w, dwell_mean=12.5,
                                          dwell_std=0.0, noise_std=0.0, noise_sampling=False, duration_sampling=False,
                                          export_every_n_samples=16, min_noise=0.0, min_duration=3, mode=mode)
    k = m.config["seq_kmer"]
    saves = 0
    for batch in iter_batches(reads, k, 10, m.device):         # batches straddle reads; export every 16 chunks
        before = len(m.results)
        m.predict_step(batch)
        saves += len(m.results) <= 1 and before >= 1
    m.on_predict_epoch_end()
    assert saves >= 2 and m.results == []
    check_file(tmp_path / "o.blow5", exp, [n for _, n in reads])


@pytest.mark.parametrize("ext", [".blow5", ".slow5", ".pod5", ".blow5+svb-zd"])
def test_streaming_path_equals_predict_step_path(tmp_path, ext, monkeypatch):
    """run_streaming (GPU zero-strip + DAC, no per-chunk Python objects; for .pod5 and svb-zd .blow5 also the StreamVByte
    stage of the signal codec on the GPU and the record / row compression on native threads) writes the same file content as
    the reference-shaped predict_step / export_and_clear_results / writer.save() flow, samplers on, fixed seed."""
    from seq2squiggle_amd.cli import set_config
    if ext.endswith("+svb-zd"):
        ext = ".blow5"
        monkeypatch.setenv("S2S_BLOW5_SIGNAL", "svb-zd")
    outs = []
    rng = np.random.default_rng(1)
    fa = tmp_path / "reads.fa"
    with open(fa, "w") as f:
        for i, n in enumerate([5, 9, 40, 333, 1200, 16 + 8, 2500]):
            f.write(f">r{i}\n{''.join(rng.choice(list('ACGT'), n))}\n")
    for streaming in (True, False):
        out = tmp_path / f"s{int(streaming)}{ext}"
        np.random.seed(0)
        inference_run(config=set_config(None), saved_weights=os.path.join(GOLDEN, "synthetic_k9.ckpt"), fasta=str(fa),
                      read_input=True, n=-1, r=1000, c=-1, out=str(out), profile="dna-r10-prom", dwell_mean=None, dwell_std=0.0,
                      noise_std=2.0, noise_sampling=True, duration_sampling=True, distr="expon", predict_batch_size=64,
                      export_every_n_samples=100, sample_rate=None, bps=None, digitisation=None, range_val=None,
                      offset_mean=None, offset_std=None, median_before_mean=None, median_before_std=None, min_noise=0.0,
                      min_duration=3, min_read_len=30, preserve_read_ids=True, seed=11, streaming=streaming)
        if ext == ".pod5":
            from seq2squiggle_amd import pod5_io
            outs.append(pod5_io.read_pod5(str(out))["reads"])
        else:
            outs.append((signal_io.read_slow5 if ext == ".slow5" else signal_io.read_blow5)(str(out))[1])
    a, b = outs
    if ext == ".pod5":
        import uuid
        assert [r["read_id"] for r in a] == [r["read_id"] for r in b] == [uuid.uuid5(uuid.NAMESPACE_DNS, f"r{i}") for i in range(1, 7)]
        for ra, rb in zip(a, b):
            assert np.array_equal(ra["signal"], rb["signal"]) and ra["num_samples"] == rb["num_samples"] == len(ra["signal"])
            assert ra["calibration_offset"] == rb["calibration_offset"] and ra["read_number"] == rb["read_number"]
        return
    assert [r["read_id"] for r in a] == [r["read_id"] for r in b] == ["r1", "r2", "r3", "r4", "r5", "r6"]
    for ra, rb in zip(a, b):
        assert np.array_equal(ra["signal"], rb["signal"])
        assert ra["start_time"] == rb["start_time"] and ra["len_raw_signal"] == rb["len_raw_signal"]


def test_reference_style_onehot_batch(tmp_path):
    """predict_step also accepts the reference's own batch format (tuple[str], fp16 one-hot [B,16,k,5])."""
    g = load_npz("stages_k9.npz")
    codes = torch.from_numpy(g["codes"].astype(np.int64))
    oh = torch.zeros(*codes.shape, 5, dtype=torch.float16)
    oh.scatter_(-1, codes.clamp(max=4).unsqueeze(-1), (codes < 5).unsqueeze(-1).to(torch.float16))

    class W:
        signals = None

        def save(self):
            self.saved = dict(self.signals)
    w = W()
    m = seq2squiggle.load_from_checkpoint(os.path.join(GOLDEN, "synthetic_k9.ckpt"), out_writer=w, dwell_mean=12.5, noise_std=0.0,
                                          min_duration=3, mode="f32")
    m.predict_step((tuple(str(n) for n in g["names"]), oh))
    m.on_predict_epoch_end()
    sig = load_npz("signals_k9.npz")                         # same chunks, noise on in that golden -> compare lengths to ideal instead
    for rid in dict.fromkeys(str(n) for n in g["names"]):
        rows = [torch.from_numpy(g["y_ideal"][i]) for i, n in enumerate(g["names"]) if str(n) == rid]
        ref = O.strip_zeros(rows).numpy()
        got = w.saved[rid].cpu().numpy()
        assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-3


def test_inference_run_and_cli(tmp_path):
    exp = expected_records("k9", "dna-r10-prom")
    ids = [n for _, n in U.read_fasta(FASTA)]
    from seq2squiggle_amd.cli import set_config
    out = tmp_path / "run.slow5"
    inference_run(config=set_config(None), saved_weights=os.path.join(GOLDEN, "synthetic_k9.ckpt"), fasta=FASTA, read_input=True,
                  n=-1, r=1000, c=-1, out=str(out), profile="dna-r10-prom", dwell_mean=None, dwell_std=0.0, noise_std=0.0,
                  noise_sampling=False, duration_sampling=False, distr="expon", predict_batch_size=1024,
                  export_every_n_samples=1000000, sample_rate=None, bps=None, digitisation=None, range_val=None,
                  offset_mean=None, offset_std=None, median_before_mean=None, median_before_std=None, min_noise=0.0,
                  min_duration=3, min_read_len=30, preserve_read_ids=True, seed=1)
    check_file(out, exp, ids)
    out2 = tmp_path / "cli.blow5"
    r = subprocess.run([sys.executable, "-m", "seq2squiggle_amd", "predict", FASTA, "--read-input", "-o", str(out2), "-m",
                        os.path.join(GOLDEN, "synthetic_k9.ckpt"), "--noise-std", "0", "--noise-sampler", "False",
                        "--duration-sampler", "False", "--preserve-read-ids", "--seed", "1"], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    check_file(out2, exp, ids)
    # default samplers on, reference mode, sampled reads: runs and writes the requested number of reads
    out3 = tmp_path / "ref.blow5"
    r = subprocess.run([sys.executable, "-m", "seq2squiggle_amd", "predict", os.path.join(GOLDEN, "example_lambda_genome.fasta"),
                        "-n", "20", "-r", "2000", "-o", str(out3), "-m", os.path.join(GOLDEN, "synthetic_k9.ckpt"), "--seed", "3"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    _, recs = signal_io.read_blow5(str(out3))
    assert len(recs) == 20 and all(r_["len_raw_signal"] > 100 for r_ in recs)
    assert recs[1]["read_id"] == "00000000-0000-0000-0000-000000000002"
    with pytest.raises(ValueError):                          # seq_kmer mismatch: k=9 checkpoint with an r9 profile
        inference_run(config=set_config(None), saved_weights=os.path.join(GOLDEN, "synthetic_k9.ckpt"), fasta=FASTA,
                      read_input=True, n=-1, r=1000, c=-1, out=str(tmp_path / "x.slow5"), profile="dna-r9-min", dwell_mean=None,
                      dwell_std=0.0, noise_std=0.0, noise_sampling=False, duration_sampling=False, distr="expon",
                      predict_batch_size=1024, export_every_n_samples=1000000, sample_rate=None, bps=None, digitisation=None,
                      range_val=None, offset_mean=None, offset_std=None, median_before_mean=None, median_before_std=None,
                      min_noise=0.0, min_duration=3, min_read_len=30, preserve_read_ids=True, seed=1)


def test_cli_pod5_output_equals_blow5_output(tmp_path):
    """`-o x.pod5` (reference flow: everything kept until the end, one POD5Writer.save) holds the same int16 samples as
    the streaming .blow5 run of the same command; read ids are the uuid5 of the FASTA names (signal_io.py:262)."""
    import uuid
    from seq2squiggle_amd import pod5_io
    outs = {}
    for ext in (".pod5", ".blow5"):
        out = tmp_path / ("o" + ext)
        r = subprocess.run([sys.executable, "-m", "seq2squiggle_amd", "predict", FASTA, "--read-input", "-o", str(out), "-m",
                            os.path.join(GOLDEN, "synthetic_k9.ckpt"), "--preserve-read-ids", "--seed", "5"], cwd=ROOT,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[ext] = out
    d = pod5_io.read_pod5(str(outs[".pod5"]))
    _, recs = signal_io.read_blow5(str(outs[".blow5"]))
    assert len(d["reads"]) == len(recs) > 0
    for p, b in zip(d["reads"], recs):
        assert p["read_id"] == uuid.uuid5(uuid.NAMESPACE_DNS, b["read_id"])
        assert np.array_equal(p["signal"], b["signal"])
        assert abs(p["calibration_scale"] - b["range"] / b["digitisation"]) < 1e-6


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the driver's keys, the roofline object and (at N=1) the CPU baseline."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--reads", "60"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["unit"] == "samples/s" and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["avg_launch_ms"] > 0
    assert rf["peak"] == 2500.0                       # the guide's dense f16 peak, not a derated one
    assert abs(rf["frac_of_f16x3_ceiling"] - 3 * rf["frac"]) < 1e-9 and rf["x_of_f32_mfma_peak"] > 0
    assert d["n_ranks_seen"] == 1 and len(d["per_rank_chunks_per_sec"]) == 1
    # round 6: the profiled constants in the line say which sources they were measured on -- the committed summary must be that of
    # THIS tree's kernel sources (re-run tools/pmc_run.sh + pmc_summarize.py after touching csrc/) -- and how much of the matrix work is padding
    assert rf["pmc_stale"] is False, (rf["pmc_csrc_sha256"], rf["csrc_sha256"], rf["traffic_source"])
    assert 0.15 < rf["mfma_useful_frac"] < 1 / 3 and abs(rf["mfma_useful_frac"] * rf["mfma_issued_flop_per_chunk"] - rf["flop_per_chunk"]) < 1
    assert rf["traffic"] > rf["algorithmic_bytes_per_launch"] > 0
    assert d["barrier_backend"] is None and d["devices_distinct"] is True and len(d["per_rank_device"]) == 1 and d["per_rank_device"][0]["pci"]
    pw = d["power"]                                   # live: what the package drew while the launches ran (here: small launches, not the cap)
    assert "error" in pw or (pw["package_watts"] > 100 and pw["cap_watts"] >= pw["package_watts"] * 0.5 and pw["microjoule_per_chunk"] > 0)
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["parity"]["dwell_indices_equal"] and cb["parity"]["signal_mae_pa"] < cb["parity"]["tolerance_mae_pa"]


def test_rna_profile_streaming_equals_reference_flow(tmp_path):
    """rna-004: dwell 4000/130 = 30.8 samples per base (16 k-mers overflow the 250-sample chunk: crop path), signals
    reversed per read (signal_io.py:140-141) -- on the GPU in the streaming path, on the host in the reference flow."""
    from seq2squiggle_amd.cli import set_config
    rng = np.random.default_rng(2)
    fa = tmp_path / "reads.fa"
    with open(fa, "w") as f:
        for i, n in enumerate([30, 100, 700]):
            f.write(f">t{i}\n{''.join(rng.choice(list('ACGU'), n))}\n")
    outs = []
    for streaming in (True, False):
        out = tmp_path / f"r{int(streaming)}.blow5"
        inference_run(config=set_config(None), saved_weights=os.path.join(GOLDEN, "synthetic_k9.ckpt"), fasta=str(fa),
                      read_input=True, n=-1, r=1000, c=-1, out=str(out), profile="rna-004-prom", dwell_mean=None, dwell_std=0.0,
                      noise_std=0.0, noise_sampling=False, duration_sampling=False, distr="expon", predict_batch_size=16,
                      export_every_n_samples=20, sample_rate=None, bps=None, digitisation=None, range_val=None,
                      offset_mean=None, offset_std=None, median_before_mean=None, median_before_std=None, min_noise=0.0,
                      min_duration=3, min_read_len=30, preserve_read_ids=True, seed=3, streaming=streaming)
        outs.append(signal_io.read_blow5(str(out)))
    (ha, a), (hb, b) = outs
    assert "rna" in ha and [r["read_id"] for r in a] == [r["read_id"] for r in b] == ["t0", "t1", "t2"]
    for ra, rb in zip(a, b):
        assert np.array_equal(ra["signal"], rb["signal"]) and len(ra["signal"]) > 0
    # every chunk is full (16 x 31 > 250): 250 samples per chunk minus ReLU zeros
    assert len(a[2]["signal"]) > 0.9 * 250 * -(-(700 - 8) // 16)


def test_a_narrowed_rank_uses_its_one_device_and_ranks_never_share_one(tmp_path):
    """a child of `predict --gpus N` sees exactly one device (HIP_VISIBLE_DEVICES = its own) and therefore uses
    index 0 whatever its LOCAL_RANK; on a box with fewer GPUs than ranks the command FAILS (the rank whose device does not exist says
    so, the launcher ends the others and returns non-zero) instead of quietly putting two ranks on one device."""
    lam = os.path.join(GOLDEN, "example_lambda_genome.fasta")
    base = [sys.executable, "-m", "seq2squiggle_amd", "predict", lam, "-n", "24", "-r", "2000", "-m",
            os.path.join(GOLDEN, "synthetic_k9.ckpt"), "--seed", "5", "-v", "debug"]
    env0 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "S2S_ONE_GPU")}
    first = (os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES") or "0").split(",")[0]
    # rank 3 of 4 as the launcher would start it on a node where its GPU is this box's one: narrowed to that device
    env = dict(env0, RANK="3", LOCAL_RANK="3", WORLD_SIZE="4", LOCAL_WORLD_SIZE="4", HIP_VISIBLE_DEVICES=first, CUDA_VISIBLE_DEVICES=first,
               S2S_PARENT_VISIBLE="")
    r = subprocess.run(base + ["-o", str(tmp_path / "n.blow5")], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "bound to CPUs" in r.stderr                                # placement.pin_rank ran before the first allocation (debug log)
    _, narrowed = signal_io.read_blow5(str(tmp_path / "n.rank3.blow5"))
    # ... the same shard from the rehearsal switch (every rank on cuda:0, nothing narrowed)
    env = dict(env0, RANK="3", LOCAL_RANK="3", WORLD_SIZE="4", LOCAL_WORLD_SIZE="4", S2S_ONE_GPU="1", S2S_NO_PIN="1")
    r = subprocess.run(base + ["-o", str(tmp_path / "o.blow5")], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "bound to CPUs" not in r.stderr                            # S2S_NO_PIN opts out
    _, plain = signal_io.read_blow5(str(tmp_path / "o.rank3.blow5"))
    assert len(narrowed) == len(plain) > 0
    for a, b in zip(narrowed, plain):
        assert a["read_id"] == b["read_id"] and np.array_equal(a["signal"], b["signal"])
    # two ranks, one GPU, no rehearsal switch: rank 1's device does not exist -> the command fails, no output file, no rank left behind
    if len((os.environ.get("HIP_VISIBLE_DEVICES") or "0").split(",")) == 1:
        if torch.cuda.device_count() == 1:
            r = subprocess.run(base + ["-o", str(tmp_path / "two.blow5"), "--gpus", "2"], cwd=ROOT, capture_output=True, text=True,
                               timeout=600, env=dict(env0, S2S_RANK_GRACE="5"))
            assert r.returncode != 0, r.stdout[-1000:]
            assert not os.path.exists(tmp_path / "two.blow5")
            assert "rank 1 exited" in r.stderr or "rank 1" in r.stderr


def test_three_rank_shards_equal_the_single_gpu_run(tmp_path):
    """SURVEY 8e on the real device path: RANK/WORLD_SIZE = r/3 runs (one after the other, all on GPU 0) write
    out.rank{r}.blow5; their reads, concatenated in rank order, carry exactly the samples of the single-process run
    (same read set from the seed, RNG keyed by the global chunk index)."""
    lam = os.path.join(GOLDEN, "example_lambda_genome.fasta")
    base = [sys.executable, "-m", "seq2squiggle_amd", "predict", lam, "-n", "30", "-r", "2000", "-m",
            os.path.join(GOLDEN, "synthetic_k9.ckpt"), "--seed", "9"]
    env0 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(base + ["-o", str(tmp_path / "one.blow5")], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env0)
    assert r.returncode == 0, r.stderr[-2000:]
    _, one = signal_io.read_blow5(str(tmp_path / "one.blow5"))
    parts = []
    for rank in range(3):
        env = dict(env0, RANK=str(rank), WORLD_SIZE="3", LOCAL_RANK="0")
        r = subprocess.run(base + ["-o", str(tmp_path / "out.blow5")], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        _, recs = signal_io.read_blow5(str(tmp_path / f"out.rank{rank}.blow5"))
        assert len(recs) > 0
        parts += recs
    assert len(parts) == len(one) == 30
    for a, b in zip(parts, one):
        assert np.array_equal(a["signal"], b["signal"])
        # the shard files merge without collisions: ids, read numbers and the per-read offset / median_before draws are
        # those of the single-process run (start_time is per file)
        assert a["read_id"] == b["read_id"] and a["read_number"] == b["read_number"]
        assert a["offset"] == b["offset"] and a["median_before"] == b["median_before"]
    assert len({r["read_id"] for r in parts}) == 30
    # the same job as ONE command: `predict --gpus 3` starts its three ranks itself (children under torch.distributed.run; here
    # all on this box's one GPU), `merge-shards` joins their files
    for rank in range(3):
        os.remove(tmp_path / f"out.rank{rank}.blow5")
    r = subprocess.run(base + ["-o", str(tmp_path / "out.blow5"), "--gpus", "3", "--keep-shards"], cwd=ROOT, capture_output=True,
                       text=True, timeout=900, env=dict(env0, S2S_ONE_GPU="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    shards = [str(tmp_path / f"out.rank{rank}.blow5") for rank in range(3)]
    r = subprocess.run([sys.executable, "-m", "seq2squiggle_amd", "merge-shards", *shards, "-o", str(tmp_path / "merged.blow5")],
                       cwd=ROOT, capture_output=True, text=True, timeout=300, env=env0)
    assert r.returncode == 0 and "30 records" in r.stdout, r.stdout + r.stderr
    _, merged = signal_io.read_blow5(str(tmp_path / "merged.blow5"))
    # without --keep-shards the command itself leaves ONE file, like the reference (here .pod5: re-tabled, VBZ rows copied as stored)
    r = subprocess.run(base + ["-o", str(tmp_path / "all.pod5"), "--gpus", "2"], cwd=ROOT, capture_output=True, text=True,
                       timeout=900, env=dict(env0, S2S_ONE_GPU="1"))
    assert r.returncode == 0 and "30 reads from 2 ranks" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("all.")) == ["all.pod5"]
    from seq2squiggle_amd import pod5_io
    p5 = pod5_io.read_pod5(str(tmp_path / "all.pod5"))["reads"]
    assert len(p5) == 30 and all(np.array_equal(a["signal"], b["signal"]) and a["read_number"] == b["read_number"] for a, b in zip(p5, one))
    for a, b in zip(merged, one):
        assert np.array_equal(a["signal"], b["signal"]) and a["read_id"] == b["read_id"] and a["read_number"] == b["read_number"]
        assert a["offset"] == b["offset"] and a["median_before"] == b["median_before"]


def test_more_ranks_than_reads(tmp_path):
    """`--gpus 3` with two reads: the rank without reads leaves an empty shard, the merge takes it, one file with both reads
    comes out (and equals the single-process file); no partial output appears under the final name when a merge fails."""
    lam = os.path.join(GOLDEN, "example_lambda_genome.fasta")
    base = [sys.executable, "-m", "seq2squiggle_amd", "predict", lam, "-n", "2", "-r", "2000", "-m",
            os.path.join(GOLDEN, "synthetic_k9.ckpt"), "--seed", "9"]
    env0 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(base + ["-o", str(tmp_path / "one.blow5")], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env0)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run(base + ["-o", str(tmp_path / "few.blow5"), "--gpus", "3"], cwd=ROOT, capture_output=True, text=True,
                       timeout=900, env=dict(env0, S2S_ONE_GPU="1"))
    assert r.returncode == 0 and "2 reads from 3 ranks" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("few.")) == ["few.blow5"]
    _, one = signal_io.read_blow5(str(tmp_path / "one.blow5"))
    _, few = signal_io.read_blow5(str(tmp_path / "few.blow5"))
    assert len(few) == len(one) == 2
    for a, b in zip(few, one):
        assert a["read_id"] == b["read_id"] and np.array_equal(a["signal"], b["signal"]) and a["offset"] == b["offset"]
    with pytest.raises(FileNotFoundError):
        signal_io.merge_shards([str(tmp_path / "one.blow5"), str(tmp_path / "absent.blow5")], str(tmp_path / "m.blow5"))
    assert not os.path.exists(tmp_path / "m.blow5") and not os.path.exists(tmp_path / "m.partial.blow5")


def test_rank_shards_in_read_mode_skip_dropped_reads(tmp_path):
    """--read-input with reads too short for one chunk scattered through the file (they produce no record,
    reference dataloader.py:393-398): the shard writers must count RECORDS, not reads, so that ids, read numbers and
    the per-record np.random draws of the 2-rank run still equal the single-process run's."""
    rng = np.random.default_rng(17)
    lines = []
    for i in range(24):
        n = 5 if i in (0, 3, 4, 11, 12, 20) else int(rng.integers(200, 1500))        # 5 < k = 9: dropped
        lines.append(f">read{i}\n" + "".join(rng.choice(list("ACGT"), n)) + "\n")
    fa = tmp_path / "reads.fasta"
    fa.write_text("".join(lines))
    base = [sys.executable, "-m", "seq2squiggle_amd", "predict", str(fa), "--read-input", "-m",
            os.path.join(GOLDEN, "synthetic_k9.ckpt"), "--seed", "5"]
    env0 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(base + ["-o", str(tmp_path / "one.blow5")], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env0)
    assert r.returncode == 0, r.stderr[-2000:]
    _, one = signal_io.read_blow5(str(tmp_path / "one.blow5"))
    parts = []
    for rank in range(2):
        env = dict(env0, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0")
        r = subprocess.run(base + ["-o", str(tmp_path / "out.blow5")], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        parts += signal_io.read_blow5(str(tmp_path / f"out.rank{rank}.blow5"))[1]
    assert len(parts) == len(one) == 18
    for a, b in zip(parts, one):
        assert np.array_equal(a["signal"], b["signal"]) and a["read_id"] == b["read_id"] and a["read_number"] == b["read_number"]
        assert a["offset"] == b["offset"] and a["median_before"] == b["median_before"]


def test_two_ranks_on_one_gpu_bench_rehearsal(tmp_path):
    """`bench.py --gpus 2` exactly as the driver starts it (children of an untouched parent, torchrun rendezvous on
    127.0.0.1), rehearsed on this box's ONE GPU (S2S_BENCH_ONE_GPU: both ranks on cuda:0; the barrier is a host group at any N).  It cannot measure scaling (the ranks share the GPU) but it runs everything the 8-GPU line will run:
    the N > 1 code path of the timed loop, the sharded end-to-end leg (each rank: FASTA parse, native sampler skip-ahead,
    its read shard, cpu_share() = quota / LOCAL_WORLD_SIZE threads, its own shard file) and the max-over-ranks timing.
    Checked: one JSON line, both ranks seen, the shards add up to the whole job, and two ranks sharing one GPU move at
    least 70 % of what one rank moves end to end (a host-side collapse under halved thread counts would show here)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["S2S_BENCH_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "500"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["scaling"] == "weak" and len(d["per_rank_chunks_per_sec"]) == 2
    # the N > 1 line (round 6): the ranks met on a HOST group, every rank says what it ran on, RCCL is a reported self-test (two ranks
    # on ONE device: RCCL refuses -- reported, nothing else lost), and the CPU baseline with its live parity check is there as at N = 1
    assert "end_to_end" not in d and d["barrier_backend"] == "gloo"
    dev = d["per_rank_device"]
    assert len(dev) == 2 and dev[0]["pci"] == dev[1]["pci"] and dev[0]["pid"] != dev[1]["pid"]
    assert d["devices_distinct"] is False and "one_gpu_rehearsal" in d and "invalid" not in d
    assert dev[0]["cpus"] and dev[1]["cpus"] and dev[0]["cpus"] != dev[1]["cpus"]          # each rank on its own share of the GPU's socket
    assert d["rccl_selftest"]["n_ranks"] == 2 and d["rccl_selftest"]["ok"] in (True, False)
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["parity"]["signal_mae_pa"] < 1e-4 and cb["parity"]["dwell_indices_equal"] and cb["parity"]["zero_pattern_equal"]
    assert "secondary_legs_timed_out" not in d
    e = d["end_to_end_sharded"]
    # lambda genome -n 25000 -r 5000, seed 42: the read set of the single-process run, split in two contiguous shards
    assert len(e["per_rank_chunks"]) == 2 and abs(e["per_rank_chunks"][0] - e["per_rank_chunks"][1]) < 0.02 * e["chunks"]
    assert 6_900_000 < e["chunks"] < 7_300_000 and e["output_bytes"] > 2.5e9
    assert e["per_rank_cpu_share_threads"][0] == e["per_rank_cpu_share_threads"][1] >= 1 and e["local_world_size"] == 2
    one = subprocess.run([sys.executable, "-c", (
        "import json,sys,os;sys.path.insert(0,%r);import bench;e=bench.end_to_end_one(12500);print(json.dumps(e))" % ROOT)],
        cwd=ROOT, capture_output=True, text=True, timeout=900, env={k: v for k, v in env.items() if k != "S2S_BENCH_ONE_GPU"})
    assert one.returncode == 0, one.stderr[-3000:]
    single = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    ratio = e["chunks_per_sec"] / single["chunks_per_sec"]
    print(f"two ranks on one GPU: {e['chunks_per_sec']:.3e} chunks/s end to end ({e['per_rank_cpu_share_threads'][0]} host threads per rank) "
          f"vs one rank {single['chunks_per_sec']:.3e}: x{ratio:.2f}")
    assert ratio > 0.7, (e, single)          # measured 0.99-1.04; a host-side collapse under halved thread counts would read <= 0.5


def test_a_run_far_above_the_redo_threshold_on_the_fast_path_says_so(tmp_path, caplog):
    """the calibration launch decides the attention path on 512 pseudo-random chunks; a RUN that ends on the
    fast path with more than the calibration's threshold (5.5 %) of its heads redone (here: the decoder's w_qs / w_ks x 4 forced onto the fast path, 79 %) logs a
    WARNING that names `--attention-path exact` -- at the default verbosity, not at debug -- and the output is the file the exact
    path writes (no run-time switching: a chunk's result never depends on its batch).  The committed checkpoint says nothing."""
    import logging
    from seq2squiggle_amd.cli import set_config
    ck = torch.load(os.path.join(GOLDEN, "synthetic_k9.ckpt"), map_location="cpu", weights_only=True)
    for k in ck["state_dict"]:
        if k.startswith("decoders.") and k.endswith(("w_qs.weight", "w_ks.weight", "w_qs.bias", "w_ks.bias")):
            ck["state_dict"][k] = ck["state_dict"][k] * 4.0
    sharp = str(tmp_path / "sharp.ckpt")
    torch.save(ck, sharp)
    lam = os.path.join(GOLDEN, "example_lambda_genome.fasta")

    def run(weights, out, path):
        caplog.clear()
        U.set_seeds(9)                     # (the CLI does this before inference_run: the read sampler draws from the global generators)
        with caplog.at_level(logging.INFO, logger="seq2squiggle"):
            m = inference_run(config=set_config(None), saved_weights=weights, fasta=lam, read_input=False, n=12, r=1500, c=-1,
                              out=str(out), profile="dna-r10-prom", dwell_mean=None, dwell_std=0.0, noise_std=2.0, noise_sampling=True,
                              duration_sampling=True, distr="expon", predict_batch_size=1024, export_every_n_samples=1000000,
                              sample_rate=None, bps=None, digitisation=None, range_val=None, offset_mean=None, offset_std=None,
                              median_before_mean=None, median_before_std=None, min_noise=0.0, min_duration=3, min_read_len=30,
                              preserve_read_ids=False, seed=9, attention_path=path)
        warned = [r for r in caplog.records if r.levelno == logging.WARNING and "--attention-path exact" in r.getMessage()]
        return m, warned
    m, warned = run(sharp, tmp_path / "fast.blow5", "fast")
    assert len(warned) == 1 and m.run_stats["attention_path"] == "fast" and m.run_stats["redo_rate"] > 0.5
    assert "%" in warned[0].getMessage()
    m, warned = run(sharp, tmp_path / "auto.blow5", "auto")              # the calibration sends these weights to the exact instance: nothing to warn about
    assert not warned and m.run_stats["attention_path"] == "exact" and m.run_stats["softmax_redone"] == 0
    m, warned = run(os.path.join(GOLDEN, "synthetic_k9.ckpt"), tmp_path / "plain.blow5", "auto")
    assert not warned and m.run_stats["attention_path"] == "fast" and m.run_stats["redo_rate"] < 0.01
    a, b = signal_io.read_blow5(str(tmp_path / "fast.blow5"))[1], signal_io.read_blow5(str(tmp_path / "auto.blow5"))[1]
    assert len(a) == len(b) == 12
    equal = 0
    for x, y in zip(a, b):        # two roundings of the same (near-one-hot, rounding-amplifying) softmax: a pre-ReLU value within 1e-4 of
        # zero may land on either side, and an exact 0 is stripped from the read (model.py:284-286) -- lengths agree to a few samples,
        # and reads of equal length hold the same int16 samples but for ties
        assert abs(x["len_raw_signal"] - y["len_raw_signal"]) <= 0.02 * y["len_raw_signal"] + 1
        if x["len_raw_signal"] == y["len_raw_signal"]:
            equal += 1
            d = np.abs(x["signal"].astype(np.int32) - y["signal"].astype(np.int32))
            assert d.max() <= 1 and (d != 0).mean() < 0.01
    assert equal >= 3
    with pytest.raises(ValueError, match="attention_path"):
        run(sharp, tmp_path / "x.blow5", "sideways")


@pytest.mark.parametrize("ext", ["blow5", "pod5"])
def test_live_join_of_a_three_rank_run_holds_the_single_process_reads(tmp_path, ext):
    """`predict --gpus 3 --join live` (all ranks on this box's one GPU): the parent joins the rank files WHILE the ranks write them
    (merge.LiveJoin); the one file that comes out holds exactly the reads of the single-process run -- ids, read numbers, offset
    draws, int16 samples -- in the round-robin order of the live join (BLOW5) / with the reads table in read order (POD5), no rank
    file is left, and a third rank without reads (more ranks than reads in the second case) is no obstacle."""
    from seq2squiggle_amd import pod5_io
    lam = os.path.join(GOLDEN, "example_lambda_genome.fasta")
    env0 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for n_reads, tag in ((900, "many"), (2, "few")):
        base = [sys.executable, "-m", "seq2squiggle_amd", "predict", lam, "-n", str(n_reads), "-r", "3000", "-m",
                os.path.join(GOLDEN, "synthetic_k9.ckpt"), "--seed", "21"]
        one, live = str(tmp_path / f"{tag}_one.{ext}"), str(tmp_path / f"{tag}_live.{ext}")
        r = subprocess.run(base + ["-o", one], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env0)
        assert r.returncode == 0, r.stderr[-2000:]
        r = subprocess.run(base + ["-o", live, "--gpus", "3", "--join", "live"], cwd=ROOT, capture_output=True, text=True, timeout=900,
                           env=dict(env0, S2S_ONE_GPU="1"))
        assert r.returncode == 0 and f"{n_reads} reads from 3 ranks" in r.stdout and "joined meanwhile" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
        assert sorted(f for f in os.listdir(tmp_path) if f.startswith(f"{tag}_live")) == [f"{tag}_live.{ext}"]
        if ext == "blow5":
            a = {x["read_id"]: x for x in signal_io.read_blow5(one)[1]}
            got = signal_io.read_blow5(live)[1]
            assert len(got) == len(a) == n_reads and {x["read_id"] for x in got} == set(a)
            for x in got:
                y = a[x["read_id"]]
                assert np.array_equal(x["signal"], y["signal"]) and x["read_number"] == y["read_number"] and x["offset"] == y["offset"]
        else:
            a, got = pod5_io.read_pod5(one)["reads"], pod5_io.read_pod5(live)["reads"]
            assert len(got) == len(a) == n_reads
            for x, y in zip(got, a):                                     # the reads table keeps the read order
                assert x["read_id"] == y["read_id"] and x["read_number"] == y["read_number"] and np.array_equal(x["signal"], y["signal"])
                assert x["calibration_offset"] == y["calibration_offset"]
