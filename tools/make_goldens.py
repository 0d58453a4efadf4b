#!/usr/bin/env python
"""Generate golden vectors by IMPORTING the reference (build container only).

Runs only where /root/reference exists.  It stubs the third-party modules the
reference imports but this image lacks, forces true-fp32
matmuls (the reference sets "medium" at import, model.py:22), builds synthetic
checkpoints in the Lightning .ckpt layout, and drives the reference's own
modules / predict_step / export / BLOW5Writer.save with *injected* random
variates.  Outputs (data only -- inputs and expected outputs) go to
tests/golden/.  Nothing of the reference's source text is written.

    python tools/make_goldens.py [pod5 | mixed16 | wide]
"""
import os
import sys
import types
import math

import numpy as np
import torch
import yaml

REF = "/root/reference"
OUT = os.environ.get("S2S_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ----------------------------------------------------------------------------- stubs
def _stub(name, **a):
    m = types.ModuleType(name)
    m.__dict__.update(a)
    sys.modules[name] = m
    return m


class _LM(torch.nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass


class _FakeS5:
    """Captures what BLOW5Writer.save hands to pyslow5 (signal_io.py:88-172)."""
    captured = []          # list of dict(mode, header, records, auxs)

    def __init__(self, path, mode):
        self.path, self.mode, self.header = path, mode, None
        open(path, "a").close()            # so the next save() sees an existing file -> append mode

    def get_empty_header(self, aux=False):
        return {}, ["unknown", "partial", "mux_change", "unblock_mux_change", "data_service_unblock_mux_change",
                    "signal_positive", "signal_negative"]

    def write_header(self, header, end_reason_labels=None):
        self.header = dict(header)
        return 0

    def get_empty_record(self, aux=False):
        return {}, {}

    def write_record_batch(self, records, threads=1, batchsize=1, aux=None):
        _FakeS5.captured.append(dict(mode=self.mode, header=self.header,
                                     records={k: dict(v) for k, v in records.items()},
                                     auxs={k: dict(v) for k, v in aux.items()}))
        return 0

    def close(self):
        pass


_stub("numba", jit=lambda *a, **k: (lambda f: f))
_stub("pytorch_lightning", LightningModule=_LM, LightningDataModule=object, Trainer=object, __version__="stub")
_stub("pytorch_lightning.loggers", WandbLogger=object)
_stub("pytorch_lightning.strategies", DDPStrategy=object)
_stub("pytorch_lightning.callbacks", LearningRateMonitor=object, ModelCheckpoint=object)
_stub("pysam", FastxFile=object)
_stub("prettytable", PrettyTable=object)
_stub("pyslow5", Open=_FakeS5)
_stub("pod5")
sys.path.insert(0, os.path.join(REF, "src"))
import seq2squiggle.model as RM          # noqa: E402  (sets matmul precision "medium", model.py:22)
import seq2squiggle.utils as RU          # noqa: E402
import seq2squiggle.signal_io as RS      # noqa: E402

torch.set_float32_matmul_precision("highest")      # true fp32 goldens (SURVEY.md section 0 finding 3)
torch.set_grad_enabled(False)


def read_fasta(path):
    name, seq, out = None, [], []
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            if name is not None:
                out.append(("".join(seq), name))
            name, seq = line[1:].split()[0], []
        elif line:
            seq.append(line)
    if name is not None:
        out.append(("".join(seq), name))
    return out


def base_config(k):
    cfg = yaml.safe_load(open(os.path.join(REF, "src/seq2squiggle/config.yaml")))
    cfg["seq_kmer"] = k
    return cfg


def make_model(seed, k):
    """Synthetic checkpoint: seeded default init, sharpened attention, calibrated head biases."""
    cfg = base_config(k)
    torch.manual_seed(seed)
    m = RM.seq2squiggle(config=cfg)
    sd = m.state_dict()
    for name, t in sd.items():
        if name.endswith(("w_qs.weight", "w_ks.weight")):
            t.mul_(3.0)                               # non-degenerate softmax rows
        if "layer_norm.weight" in name:
            t.add_(0.25 * torch.randn_like(t))       # non-trivial affine
        if "layer_norm.bias" in name:
            t.add_(0.1 * torch.randn_like(t))
    sd["length_regulator.duration_sampler.conc_layer.3.bias"].fill_(9.0)     # Gamma(~9, ~0.8): dwell ~ 11
    sd["length_regulator.duration_sampler.rate_layer.3.bias"].fill_(math.log(math.exp(0.8) - 1.0))
    sd["length_regulator.duration_sampler.conc_layer.3.weight"].mul_(4.0)
    sd["noise_sampler.stdv_layer.3.bias"].fill_(math.log(math.exp(0.01) - 1.0))  # sigma ~ 0.01 scaled
    sd["noise_sampler.stdv_layer.3.weight"].mul_(4.0)
    sd["decoders.out_linear.bias"].fill_(0.5)                                # ~ 80 pA, some ReLU zeros
    m.load_state_dict(sd)
    m.eval()
    return m, cfg


def save_ckpt(m, cfg, path):
    ckpt = {
        "epoch": 0, "global_step": 0, "pytorch-lightning_version": "2.5.1.post0",
        "state_dict": {k: v.clone() for k, v in m.state_dict().items()},
        "hyper_parameters": {"config": dict(cfg), "save_valid_plots": True, "out_writer": None,
                             "dwell_mean": 9.0, "dwell_std": 0.0, "noise_std": -1, "noise_sampling": False,
                             "duration_sampling": False, "export_every_n_samples": 2000000, "min_noise": 0.5,
                             "min_duration": 1},
        "loops": {}, "callbacks": {}, "optimizer_states": [], "lr_schedulers": [],
    }
    torch.save(ckpt, path)


def codes_from_onehot(x):
    """[C,16,k,5] one-hot -> uint8 codes, 255 for an all-zero row."""
    c = x.argmax(-1).astype(np.uint8)
    c[x.sum(-1) == 0] = 255
    return c


class Inject:
    """Replace the reference's RNG draws with recorded/injected variates."""

    def __init__(self, sg=None, z_normal=None):
        self.sg, self.z = sg, list(z_normal or [])
        self.used = []

    def __enter__(self):
        self._g, self._n = torch._standard_gamma, torch.normal
        inj = self

        def std_gamma(conc, generator=None):
            assert inj.sg is not None and inj.sg.shape == conc.shape, (conc.shape,)
            return inj.sg.clone()

        def normal(mean=0.0, std=1.0, size=None, **kw):
            z = inj.z.pop(0)
            inj.used.append(z)
            if isinstance(std, torch.Tensor) and isinstance(mean, torch.Tensor):
                return mean + z * std              # at::normal(Tensor, Tensor)
            if isinstance(std, torch.Tensor):
                return z * std + mean              # at::normal(double, Tensor)
            return z * std + mean                  # at::normal(double, double, size)
        torch._standard_gamma, torch.normal = std_gamma, normal
        return self

    def __exit__(self, *a):
        torch._standard_gamma, torch.normal = self._g, self._n


class FakeWriter:
    def __init__(self):
        self.signals, self.saved = None, []

    def save(self):
        self.saved.append({k: v.clone() for k, v in self.signals.items()})


def run_predict_step(m, names, x16, **over):
    """One reference predict_step on a fresh results list; returns [B,250] prediction rows."""
    for k, v in over.items():
        setattr(m, k, v)
    m.results, m.total_samples, m.out_writer = [], 0, None
    m.predict_step((tuple(names), x16))
    rows = []
    d = m.results[0]
    it = {k: iter(v) for k, v in d.items()}
    for n in names:
        rows.append(next(it[n]))
    return torch.stack(rows)


def stage_goldens(tag, m, cfg, reads, seed):
    k = cfg["seq_kmer"]
    names, chunks = [], []
    for seq, name in reads:
        br = RU.split_sequence(seq, cfg)
        for c in br:
            names.append(name)
            chunks.append(c)
    x = np.stack(chunks)                                  # [B,16,k,5] fp16
    codes = codes_from_onehot(x)
    B = x.shape[0]
    x16 = torch.from_numpy(x)
    g = {"codes": codes, "names": np.array(names)}

    data = x16.reshape(B, 16, -1)
    enc_out, emb_out = m.encoders(data)
    sigma = m.noise_sampler(emb_out)
    ds = m.length_regulator.duration_sampler
    conc = torch.clamp(ds.conc_layer(emb_out), min=1e-8)
    rate = torch.clamp(ds.rate_layer(emb_out), min=1e-8)
    g.update(emb_out=emb_out.numpy(), enc_out=enc_out.numpy(), sigma=sigma.numpy(),
             conc=conc.flatten(1).numpy(), rate=rate.flatten(1).numpy())

    gen = torch.Generator().manual_seed(seed)
    sg = torch._standard_gamma(conc, generator=gen)      # realistic draws, then stress cases
    sg[1] *= 2.5                                          # crop: sum(dur) > 250
    sg[2] *= 0.05                                         # clamp(1.0)/min_duration floor
    sg[3, 5:] *= 3.0
    z250 = torch.randn(B, 250, generator=gen)
    z16 = torch.randn(B, 16, generator=gen)
    g.update(sg=sg.flatten(1).numpy(), z01=z250.numpy(), zdw=z16.numpy())

    # -- gamma mode: duration sampler output (post clamp 1.0) == injected g
    with Inject(sg=sg):
        g_samp, _ = ds(emb_out)
    g["g"] = g_samp.numpy()
    with Inject(sg=sg):
        lr_out, dpo, _, sig_ext, _ = m.length_regulator(emb_out=emb_out, x=enc_out, target=None,
                                                        noise_std_prediction=sigma[:, :, None], max_length=250,
                                                        dwell_mean=12.5, dwell_std=0.0, duration_sampling=True,
                                                        min_length=3)
    dur = torch.round(dpo).int()
    g["dur_gamma"] = dur.numpy()
    g["sigma_ext_gamma"] = sig_ext.squeeze(-1).numpy()
    g["lr_rowsum_gamma"] = lr_out.sum(-1).numpy()         # compact check of the gather
    y_scaled = m.decoders(lr_out, None).squeeze(-1)
    g["y_scaled_gamma"] = y_scaled.numpy()
    common = dict(dwell_mean=12.5, dwell_std=0.0, min_duration=3)
    # full predict_step, default samplers (noise_sampling + duration_sampling), min_noise 0
    with Inject(sg=sg, z_normal=[z250]):
        g["y_gamma_nsamp"] = run_predict_step(m, names, x16, noise_std=2.0, noise_sampling=True,
                                              duration_sampling=True, min_noise=0.0, **common).numpy()
    with Inject(sg=sg, z_normal=[z250]):
        g["y_gamma_nsamp_minnoise"] = run_predict_step(m, names, x16, noise_std=1.5, noise_sampling=True,
                                                       duration_sampling=True, min_noise=0.02, **common).numpy()
    with Inject(sg=sg, z_normal=[z250]):
        g["y_gamma_nconst"] = run_predict_step(m, names, x16, noise_std=2.0, noise_sampling=False,
                                               duration_sampling=True, min_noise=0.0, **common).numpy()
    with Inject(sg=sg):
        g["y_gamma_nonoise"] = run_predict_step(m, names, x16, noise_std=0.0, noise_sampling=True,
                                                duration_sampling=True, min_noise=0.0, **common).numpy()
    # ideal dwell (config 1 of BASELINE.json): everything deterministic
    g["y_ideal"] = run_predict_step(m, names, x16, noise_std=0.0, noise_sampling=False, duration_sampling=False,
                                    min_noise=0.0, **common).numpy()
    with Inject(z_normal=[z250]):
        g["y_ideal_nsamp"] = run_predict_step(m, names, x16, noise_std=2.0, noise_sampling=True,
                                              duration_sampling=False, min_noise=0.0, **common).numpy()
    # normal dwell (dwell_std > 0): torch.normal(mean, std) then clamp(min_duration)
    with Inject(z_normal=[z16, z250]):
        g["y_normal_nsamp"] = run_predict_step(m, names, x16, noise_std=2.0, noise_sampling=True,
                                               duration_sampling=False, min_noise=0.0, dwell_mean=12.5,
                                               dwell_std=4.0, min_duration=3).numpy()
    d_norm = torch.round(torch.clamp(torch.full((B, 16), 12.5) + z16 * torch.full((B, 16), 4.0), min=3)).int()
    g["dur_normal"] = d_norm.numpy()
    # rna-like long dwell: 16*31 = 496 > 250 -> every chunk cropped
    g["y_ideal_dwell31"] = run_predict_step(m, names, x16, noise_std=0.0, noise_sampling=False,
                                            duration_sampling=False, min_noise=0.0, dwell_mean=4000 / 130,
                                            dwell_std=0.0, min_duration=3).numpy()
    np.savez_compressed(os.path.join(OUT, f"stages_{tag}.npz"), **g)
    print(tag, "chunks", B, "zeros in y_ideal:", int((g["y_ideal"] == 0).sum()),
          "dur_gamma sum range", int(dur.sum(1).min()), int(dur.sum(1).max()))
    return names, x16, sg, z250


def export_goldens(tag, m, cfg, names, x16, sg, z250, profile_name):
    """Emulated Lightning predict loop -> export_and_clear_results -> BLOW5Writer.save (reference code)."""
    profile = RU.get_profile(profile_name)
    path = os.path.join("/tmp", f"golden_{tag}.blow5")
    if os.path.exists(path):
        os.remove(path)
    _FakeS5.captured = []
    w = RS.BLOW5Writer(path, profile, ideal_mode=True, profile_name=profile_name, preserve_read_ids=True)
    for k, v in dict(noise_std=2.0, noise_sampling=True, duration_sampling=True, min_noise=0.0, dwell_mean=12.5,
                     dwell_std=0.0, min_duration=3, export_every_n_samples=16).items():
        setattr(m, k, v)
    m.results, m.total_samples, m.out_writer = [], 0, w
    bs = 10
    B = x16.shape[0]
    for s in range(0, B, bs):
        e = min(B, s + bs)
        with Inject(sg=sg[s:e], z_normal=[z250[s:e]]):
            m.predict_step((tuple(names[s:e]), x16[s:e]))
    m.on_predict_epoch_end()
    out = {"profile_name": np.array(profile_name), "batch_size": np.array(bs), "export_every": np.array(16)}
    order = []
    for si, cap in enumerate(_FakeS5.captured):
        for rid, rec in cap["records"].items():
            key = f"s{si}__{rid}"
            order.append(key)
            out[key + "__raw"] = np.asarray(rec["signal"])
            out[key + "__meta"] = np.array([rec["digitisation"], rec["offset"], rec["range"], rec["sampling_rate"],
                                            rec["len_raw_signal"], cap["auxs"][rid]["median_before"],
                                            cap["auxs"][rid]["start_time"], cap["auxs"][rid]["read_number"]],
                                           dtype=np.float64)
    out["order"] = np.array(order)
    hdr = _FakeS5.captured[0]["header"]
    out["header_keys"] = np.array(sorted(k for k in hdr if k != "exp_start_time"))
    out["header_vals"] = np.array([str(hdr[k]) for k in sorted(hdr) if k != "exp_start_time"])
    np.savez_compressed(os.path.join(OUT, f"export_{tag}.npz"), **out)
    os.remove(path)
    print(tag, "export saves:", len(_FakeS5.captured), "records:", len(order))

    # pA float signals per read (FakeWriter) with one final export
    fw = FakeWriter()
    m.results, m.total_samples, m.out_writer = [], 0, fw
    with Inject(sg=sg, z_normal=[z250]):
        m.predict_step((tuple(names), x16))
    m.on_predict_epoch_end()
    sig = {f"sig__{k}": v.numpy() for k, v in fw.saved[0].items()}
    sig["read_order"] = np.array(list(fw.saved[0].keys()))
    np.savez_compressed(os.path.join(OUT, f"signals_{tag}.npz"), **sig)


def chunker_goldens():
    rng = np.random.default_rng(7)
    reads = read_fasta(os.path.join(REF, "example/test.fasta"))
    extra = [("ACGTA", "short_lt_k"), ("ACGTACGTA", "exact_k"),
             ("".join(rng.choice(list("ACGT"), 9 + 15)), "exact_16_kmers"),
             ("".join(rng.choice(list("ACGT"), 9 + 16)), "17_kmers"),
             ("ACGTNNACGTacgtACGTRYACGTACGTACGTACGTAC_GT", "unknown_letters"),
             ("".join(rng.choice(list("ACGT"), 333)), "rand333")]
    out = {}
    for k in (9, 6):
        cfg = base_config(k)
        for seq, name in reads + extra:
            br = RU.split_sequence(seq, cfg)
            c = codes_from_onehot(br) if br.size else np.zeros((0, 16, k), np.uint8)
            out[f"k{k}__{name}"] = c
    out["names"] = np.array([n for _, n in reads + extra])
    out["seqs"] = np.array([s for s, _ in reads + extra])
    np.savez_compressed(os.path.join(OUT, "chunker.npz"), **out)


def profile_goldens():
    names = ["dna-r10-prom", "dna-r10-min", "dna-r9-prom", "dna-r9-min", "rna-004-prom", "rna-004-min"]
    rng = np.random.default_rng(11)
    sig = np.concatenate([rng.uniform(0, 200, 500), [0.0, 1e-3, 164.999, 400.0, 2000.0, 5000.0]]).astype(np.float32)
    out = {"signal": sig, "names": np.array(names)}
    for n in names:
        p = RU.get_profile(n)
        out[n + "__profile"] = np.array([p["digitisation"], p["sample_rate"], p["bps"], p["range"], p["offset_mean"],
                                         p["offset_std"], p["median_before_mean"], p["median_before_std"]], np.float64)
        path = f"/tmp/golden_prof_{n}.blow5"
        if os.path.exists(path):
            os.remove(path)
        _FakeS5.captured = []
        w = RS.BLOW5Writer(path, p, ideal_mode=True, profile_name=n, preserve_read_ids=False)
        w.signals = {"r0": torch.from_numpy(sig)}
        with np.errstate(all="ignore"):
            w.save()
        rec = list(_FakeS5.captured[0]["records"].items())
        out[n + "__read_id"] = np.array(rec[0][0])
        out[n + "__raw"] = np.asarray(rec[0][1]["signal"])
        hdr = _FakeS5.captured[0]["header"]
        out[n + "__header_keys"] = np.array(sorted(k for k in hdr if k != "exp_start_time"))
        out[n + "__header_vals"] = np.array([str(hdr[k]) for k in sorted(hdr) if k != "exp_start_time"])
        os.remove(path)
    np.savez_compressed(os.path.join(OUT, "profiles.npz"), **out)


def sampler_goldens():
    """Read sets drawn by the reference's own sampler for fixed seeds (utils.py:415-582)."""
    import random, hashlib
    genome = read_fasta(os.path.join(REF, "example/lamda_genome.fasta"))
    seqs, lens = zip(*[RU.process_genome(s) for s, _ in genome])
    # a two-contig reference with N runs and lower case, to cover the genome lookup and N handling
    rng = np.random.default_rng(3)
    c2 = "".join(rng.choice(list("ACGTN"), 20000, p=[.24, .24, .24, .24, .04])) + "acgtnnacgt" * 50
    seqs2, lens2 = zip(*[RU.process_genome(s) for s in (genome[0][0][:30000], c2)])
    out = {}
    cases = [("lambda_expon_dna", seqs, lens, 25, 5000, -1, "expon", "dna-r10-prom", 42),
             ("lambda_beta_dna", seqs, lens, 15, 3000, -1, "beta", "dna-r9-min", 7),
             ("lambda_gamma_rna", seqs, lens, 15, 2000, -1, "gamma", "rna-004-prom", 11),
             ("two_contigs_cov", seqs2, lens2, -1, 4000, 2, "expon", "dna-r10-min", 5)]
    for name, gs, gl, n, r, c, distr, prof, seed in cases:
        random.seed(seed)
        reads, total_l = RU.sample_reads_from_reference(list(gs), list(gl), n, r, c, base_config(9), "x.fasta", seed,
                                                        False, distr, prof, 30)
        reads = [rd for rd, _ in reads]
        out[name + "__lens"] = np.array([len(x) for x in reads])
        out[name + "__sha"] = np.array([hashlib.sha1(x.encode()).hexdigest() for x in reads])
        out[name + "__total_l"] = np.array(total_l)
        out[name + "__args"] = np.array([n, r, c, seed])
        out[name + "__distr_profile"] = np.array([distr, prof])
        print(name, len(reads), "reads, mean len", np.mean([len(x) for x in reads]) if reads else 0)
    out["two_contigs__seq1"] = np.array(c2)
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **out)


class _Rec:
    """Stands in for pod5.RunInfo / Pore / Calibration / EndReason / Read: keeps the keyword arguments."""
    def __init__(self, **kw):
        self.kw = kw


class _FakePod5Writer:
    captured = []          # list of (path, [Read kwargs])

    def __init__(self, path):
        self.path, self.reads = str(path), []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        _FakePod5Writer.captured.append((self.path, self.reads))

    def add_read(self, read):
        self.reads.append(read)


def pod5_goldens():
    """What the reference's POD5Writer.save (signal_io.py:201-287) hands to the pod5 library, for the per-read pA
    signals of signals_{k9,k6}.npz: the pod5 module is a recorder (the real library is not in the image)."""
    import enum
    import json
    P5 = sys.modules["pod5"]
    P5.RunInfo = P5.Pore = P5.Calibration = P5.EndReason = P5.Read = _Rec
    P5.EndReasonEnum = enum.Enum("EndReasonEnum", "UNKNOWN MUX_CHANGE UNBLOCK_MUX_CHANGE DATA_SERVICE_UNBLOCK_MUX_CHANGE "
                                                  "SIGNAL_POSITIVE SIGNAL_NEGATIVE", start=0)
    P5.Writer = _FakePod5Writer
    out = {}
    cases = [("k9", "dna-r10-prom", True, True), ("k9", "dna-r10-prom", True, False), ("k6", "rna-004-min", True, True)]
    for ci, (tag, prof, ideal, preserve) in enumerate(cases):
        sig = np.load(os.path.join(OUT, f"signals_{tag}.npz"))
        order = [str(x) for x in sig["read_order"]]
        _FakePod5Writer.captured = []
        w = RS.POD5Writer(f"/tmp/golden_{ci}.pod5", RU.get_profile(prof), ideal, prof, preserve)
        w.signals = {rid: torch.from_numpy(sig["sig__" + rid]) for rid in order}
        w.save()
        (path, reads), = _FakePod5Writer.captured
        key = f"case{ci}"
        out[key + "__args"] = np.array([tag, prof, str(ideal), str(preserve)])
        ri = reads[0].kw["run_info"].kw
        out[key + "__run_info"] = np.array(json.dumps({k: (v if isinstance(v, (str, int, float, dict)) else "<datetime>")
                                                        for k, v in ri.items()}, sort_keys=True))
        out[key + "__read_ids"] = np.array([str(r.kw["read_id"]) for r in reads])
        out[key + "__meta"] = np.array([[r.kw["calibration"].kw["offset"], r.kw["calibration"].kw["scale"],
                                         r.kw["median_before"], r.kw["read_number"], r.kw["start_sample"],
                                         r.kw["pore"].kw["channel"], r.kw["pore"].kw["well"],
                                         int(r.kw["end_reason"].kw["reason"].value), int(r.kw["end_reason"].kw["forced"])]
                                        for r in reads], np.float64)
        out[key + "__pore_type"] = np.array(reads[0].kw["pore"].kw["pore_type"])
        out[key + "__end_reason"] = np.array(reads[0].kw["end_reason"].kw["reason"].name)
        for i, r in enumerate(reads):
            out[f"{key}__raw{i}"] = np.asarray(r.kw["signal"])
            assert r.kw["run_info"] is reads[0].kw["run_info"]
        print(key, tag, prof, len(reads), "pod5 reads captured")
    np.savez_compressed(os.path.join(OUT, "pod5_records.npz"), **out)


def mixed16_goldens():
    """What the reference's OWN GPU arithmetic gives for the golden chunks: inference.py:403-404 selects precision "16-mixed" whenever a
    GPU is present, i.e. Lightning runs predict_step under torch.autocast(device, dtype=torch.float16).  No GPU build of the
    reference can run here, so the same predict_step (same committed checkpoints, same injected variates as y_gamma_nsamp) runs
    under torch.autocast("cpu", dtype=torch.float16): the CPU autocast policy casts the same Linear / matmul / bmm inputs to
    fp16 and keeps softmax / layer_norm in fp32 as the CUDA policy does; accumulation order differs from a GPU's, which is noise
    two orders below the fp16 rounding this measures.  -> tests/golden/mixed16.npz: per checkpoint the 16-mixed signal rows, the
    dwell indices under 16-mixed, and its MAE / max distance to the fp32 golden -- the bar the engine's opt-in reduced-precision
    mode (S2S_MODE_F16) is held to (tests/test_gpu_parity.py: test_reduced_precision_f16_mode)."""
    out = {}
    for tag in ("k9", "k6"):
        ck = torch.load(os.path.join(OUT, f"synthetic_{tag}.ckpt"), map_location="cpu", weights_only=True)
        cfg = ck["hyper_parameters"]["config"]
        m = RM.seq2squiggle(config=cfg)
        m.load_state_dict(ck["state_dict"])
        m.eval()
        g = dict(np.load(os.path.join(OUT, f"stages_{tag}.npz")))
        codes = g["codes"]
        x = np.zeros(codes.shape + (5,), np.float16)
        known = codes < 5
        x[known] = np.eye(5, dtype=np.float16)[codes[known]]
        x16 = torch.from_numpy(x)
        names = [str(n) for n in g["names"]]
        B = len(names)
        sg = torch.from_numpy(g["sg"]).reshape(B, 16, 1)
        z250 = torch.from_numpy(g["z01"])
        common = dict(dwell_mean=12.5, dwell_std=0.0, min_duration=3)
        with Inject(sg=sg, z_normal=[z250]):                 # sanity: the fp32 run reproduces the committed golden bit for bit
            y32 = run_predict_step(m, names, x16, noise_std=2.0, noise_sampling=True, duration_sampling=True, min_noise=0.0, **common)
        assert np.array_equal(y32.numpy(), g["y_gamma_nsamp"]), tag
        with torch.autocast("cpu", dtype=torch.float16):
            with Inject(sg=sg, z_normal=[z250]):
                y16 = run_predict_step(m, names, x16, noise_std=2.0, noise_sampling=True, duration_sampling=True, min_noise=0.0,
                                       **common).float()
            enc_out, emb_out = m.encoders(x16.reshape(B, 16, -1))
            sigma = m.noise_sampler(emb_out)
            with Inject(sg=sg):
                _, dpo, _, _, _ = m.length_regulator(emb_out=emb_out, x=enc_out, target=None, noise_std_prediction=sigma[:, :, None],
                                                     max_length=250, dwell_mean=12.5, dwell_std=0.0, duration_sampling=True, min_length=3)
        d = (y16 - y32).abs()
        same = (y16 == 0) == (y32 == 0)
        out[f"y_gamma_nsamp_16mixed_{tag}"] = y16.numpy()
        out[f"dur_gamma_16mixed_{tag}"] = torch.round(dpo.float()).int().numpy()
        out[f"mae_vs_fp32_{tag}"] = np.float64(d.mean())
        out[f"max_vs_fp32_{tag}"] = np.float64(d.max())
        out[f"zero_pattern_equal_share_{tag}"] = np.float64(same.float().mean())
        # a dwell index that rounds the other way under fp16 shifts every later sample of its chunk: the distance on the chunks
        # whose indices all agree is the arithmetic's own
        agree = torch.from_numpy((out[f"dur_gamma_16mixed_{tag}"] == g["dur_gamma"]).all(1))
        out[f"dwell_indices_differing_{tag}"] = np.int64((out[f"dur_gamma_16mixed_{tag}"] != g["dur_gamma"]).sum())
        out[f"mae_vs_fp32_where_dwell_equal_{tag}"] = np.float64(d[agree].mean())
        out[f"max_vs_fp32_where_dwell_equal_{tag}"] = np.float64(d[agree].max())
        print(tag, "dwell indices differing:", int(out[f"dwell_indices_differing_{tag}"]), "of", g["dur_gamma"].size, "| on the", int(agree.sum()),
              "chunks whose indices agree: MAE", float(d[agree].mean()), "max", float(d[agree].max()))
        print(tag, "16-mixed vs fp32 golden: MAE", float(d.mean()), "max", float(d.max()), "zero pattern equal", float(same.float().mean()),
              "dwell indices equal", bool(np.array_equal(out[f"dur_gamma_16mixed_{tag}"], g["dur_gamma"])))
    np.savez_compressed(os.path.join(OUT, "mixed16.npz"), **out)


def wide_goldens():
    """More chunks through the reference itself (round 6: the stage goldens hold 51 / 56 chunks per chemistry; everything wider was
    checked through the oracle only).  The committed checkpoints, ~256 chunks per chemistry of REAL sequence -- windows of the
    reference's lambda genome example -- plus the edge reads of the test suites (homopolymers, a dinucleotide repeat, an N-rich read,
    reads that end in a short tail), the reference's predict_step with injected variates in two configurations (the default
    samplers: Gamma dwell + sampled noise; ideal dwell + constant noise), and the dwell indices of the Gamma run.
    -> tests/golden/wide_{k9,k6}.npz: codes, sg (injected standard-gamma draws), g (the duration sampler's output for them), z01
    (injected normals), y_gamma_nsamp, dur_gamma, y_ideal_nconst.  The normals are stored as float16-exact values (drawn, then rounded to float16 and used as such) to halve
    the fixture; they are injected variates, any values do."""
    lam = read_fasta(os.path.join(REF, "example", "lamda_genome.fasta"))[0][0].upper()        # (the reference's own spelling)
    for tag, seed in (("k9", 31), ("k6", 32)):
        ck = torch.load(os.path.join(OUT, f"synthetic_{tag}.ckpt"), map_location="cpu", weights_only=True)
        cfg = ck["hyper_parameters"]["config"]
        m = RM.seq2squiggle(config=cfg)
        m.load_state_dict(ck["state_dict"])
        m.eval()
        rng = np.random.default_rng(seed)
        reads = []
        for i, start in enumerate(rng.integers(0, len(lam) - 700, size=6)):
            L = int(rng.integers(380, 640))
            reads.append((lam[int(start):int(start) + L], f"lambda_{i}"))
        reads += [("A" * 90, "homopolymer_A"), ("T" * 57, "homopolymer_T"), ("ACACACACAC" * 11, "dinucleotide"),
                  ("".join(rng.choice(list("ACGTN"), 130, p=[.2, .2, .2, .2, .2])), "n_rich"),
                  (lam[1000:1000 + 16 * 3 + cfg["seq_kmer"]], "ends_on_a_chunk_boundary"),
                  (lam[5000:5000 + 16 * 2 + cfg["seq_kmer"] + 5], "short_tail")]
        names, chunks = [], []
        for seq, name in reads:
            for c in RU.split_sequence(seq, cfg):
                names.append(name)
                chunks.append(c)
        x = np.stack(chunks)
        B = x.shape[0]
        x16 = torch.from_numpy(x)
        gen = torch.Generator().manual_seed(1000 + seed)
        enc_out, emb_out = m.encoders(x16.reshape(B, 16, -1))
        sigma = m.noise_sampler(emb_out)
        ds = m.length_regulator.duration_sampler
        conc = torch.clamp(ds.conc_layer(emb_out), min=1e-8)
        sg = torch._standard_gamma(conc, generator=gen)
        z250 = torch.randn(B, 250, generator=gen).half().float()
        common = dict(dwell_mean=12.5, dwell_std=0.0, min_duration=3)
        with Inject(sg=sg, z_normal=[z250]):
            y = run_predict_step(m, names, x16, noise_std=2.0, noise_sampling=True, duration_sampling=True, min_noise=0.0, **common)
        with Inject(sg=sg):
            _, dpo, _, _, _ = m.length_regulator(emb_out=emb_out, x=enc_out, target=None, noise_std_prediction=sigma[:, :, None],
                                                 max_length=250, dwell_mean=12.5, dwell_std=0.0, duration_sampling=True, min_length=3)
        with Inject(z_normal=[z250]):
            yi = run_predict_step(m, names, x16, noise_std=1.0, noise_sampling=False, duration_sampling=False, min_noise=0.0, **common)
        with Inject(sg=sg):
            g_samp, _ = ds(emb_out)                          # the duration sampler's output for these draws: what the engine takes as inject_g
        out = {"codes": codes_from_onehot(x), "names": np.array(names), "sg": sg.flatten(1).numpy(), "g": g_samp.numpy(), "z01": z250.half().numpy(),
               "y_gamma_nsamp": y.numpy(), "dur_gamma": torch.round(dpo).int().numpy(), "y_ideal_nconst": yi.numpy()}
        np.savez_compressed(os.path.join(OUT, f"wide_{tag}.npz"), **out)
        print(tag, "wide:", B, "chunks from", len(reads), "reads; zeros in y:", int((out["y_gamma_nsamp"] == 0).sum()),
              "bytes", os.path.getsize(os.path.join(OUT, f"wide_{tag}.npz")))


def main():
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["wide"]:          # only the wide goldens (they read the committed checkpoints)
        wide_goldens()
        return
    if sys.argv[1:] == ["pod5"]:          # only the POD5 record goldens (they read the committed signals_*.npz)
        pod5_goldens()
        return
    if sys.argv[1:] == ["mixed16"]:       # only the reference-under-fp16-autocast vectors (they read the committed checkpoints and stages_*.npz)
        mixed16_goldens()
        return
    sampler_goldens()
    chunker_goldens()
    profile_goldens()
    reads = read_fasta(os.path.join(REF, "example/test.fasta"))
    rng = np.random.default_rng(5)
    reads_x = reads + [("".join(rng.choice(list("ACGTN"), 150, p=[.24, .24, .24, .24, .04])), "rand150_with_N")]
    for tag, seed, k, prof in (("k9", 1, 9, "dna-r10-prom"), ("k6", 2, 6, "dna-r9-min")):
        m, cfg = make_model(seed, k)
        save_ckpt(m, cfg, os.path.join(OUT, f"synthetic_{tag}.ckpt"))
        names, x16, sg, z250 = stage_goldens(tag, m, cfg, reads_x, seed=100 + k)
        export_goldens(tag, m, cfg, names, x16, sg, z250, prof)
    pe = {"enc": RM.Encoder(base_config(9)).position_enc.numpy(), "dec": RM.Decoder(base_config(9)).position_enc.numpy()}
    np.savez_compressed(os.path.join(OUT, "position_enc.npz"), **pe)
    pod5_goldens()
    mixed16_goldens()
    wide_goldens()
    print("goldens written to", OUT)


if __name__ == "__main__":
    main()
