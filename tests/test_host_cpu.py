"""CPU-side checks of the host logic and of the C-ABI library surface (no GPU compute)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import seq2squiggle_amd as S
from seq2squiggle_amd import _lib, chunker
from conftest import ROOT, load_ckpt, load_npz


def test_chunker_matches_reference_codes():
    g = load_npz("chunker.npz")
    lut = np.full(256, 255, np.uint8)
    for i, ch in enumerate("_ACGT"):
        lut[ord(ch)] = i
    for k in (9, 6):
        for name, seq in zip(g["names"], g["seqs"]):
            bases, nv = S.encode_read(str(seq), k)
            exp = g[f"k{k}__{name}"]
            assert bases.shape == (exp.shape[0], 16 + k - 1)
            # rebuild the reference's [C,16,k] codes from (bases, n_valid) exactly as the kernel reads them
            got = np.zeros_like(exp)
            for c in range(exp.shape[0]):
                for j in range(16):
                    got[c, j] = lut[bases[c, j:j + k]] if j < nv[c] else 0
            assert np.array_equal(got, exp), (k, name)


def test_codes_to_bases_roundtrip():
    g = load_npz("stages_k9.npz")
    bases, nv = chunker.codes_to_bases(g["codes"])
    lut = np.full(256, 255, np.uint8)
    for i, ch in enumerate("_ACGT"):
        lut[ord(ch)] = i
    for b in range(bases.shape[0]):
        for j in range(16):
            exp = g["codes"][b, j]
            got = lut[bases[b, j:j + 9]] if j < nv[b] else np.zeros(9, np.uint8)
            assert np.array_equal(got, exp)


def test_encode_reads_ranges():
    bases, nv, first = S.encode_reads(["ACGT" * 10, "AC", "ACGTACGTACG"], 9)
    assert first.tolist() == [0, 2, 2, 3]
    assert bases.shape == (3, 24) and nv.tolist() == [16, 16, 3]


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "s2s_hip.h")).read()
    declared = set(re.findall(r"\b(s2s_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_blob_layout_and_size():
    L = _lib.lib()
    for tag in ("k9", "k6"):
        sd, cfg = load_ckpt(tag)
        blob = S.state_dict_to_blob(sd, cfg)
        c = S.config_to_c(cfg)
        assert blob.size == L.s2s_blob_floats(ctypes.byref(c)) == sum(v.numel() for v in sd.values())
    bad = S.config_to_c(cfg)
    bad.dmodel = 128
    assert L.s2s_blob_floats(ctypes.byref(bad)) == 0


def test_create_rejects_bad_arguments_without_gpu():
    L = _lib.lib()
    sd, cfg = load_ckpt("k9")
    c = S.config_to_c(cfg)
    h = ctypes.c_void_p()
    c.dff = 512
    assert L.s2s_create(ctypes.byref(c), None, 0, 0, ctypes.byref(h)) == -1
    assert b"dff" in L.s2s_last_error(None)
    c = S.config_to_c(cfg)
    blob = S.state_dict_to_blob(sd, cfg)
    assert L.s2s_create(ctypes.byref(c), blob.ctypes.data_as(ctypes.c_void_p), blob.nbytes - 4, 0, ctypes.byref(h)) == -3


def test_checkpoint_loader_errors(tmp_path):
    p = tmp_path / "x.ckpt"
    torch.save({"foo": 1}, p)
    with pytest.raises(ValueError):
        S.load_checkpoint(str(p))
    torch.save({"state_dict": {}}, p)
    with pytest.raises(ValueError):
        S.load_checkpoint(str(p))


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_engine_fails_loudly_without_gpu():
    sd, cfg = load_ckpt("k9")
    with pytest.raises(RuntimeError):
        S.Engine(sd, cfg)


def test_plain_c_consumer(tmp_path):
    """gcc-compiled C program dlopens the library, resolves every symbol and exercises the argument checks."""
    import subprocess
    exe = tmp_path / "cabi_check"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "cabi_check.c"), "-o", str(exe), "-ldl"], check=True)
    r = subprocess.run([str(exe), _lib._LIB_DEFAULT], capture_output=True, text=True)
    assert r.returncode == 0 and "CABI_OK 236804" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_pore_data_module_predict_loader():
    """The reference's PoreDataModule(...).predict_dataloader() surface (dataloader.py:27-149): batches of predict_batch_size
    chunks in read order, reads shorter than k dropped (dataloader.py:393-398), len() = total_l, and the reference's own fp16
    one-hot batch format on request -- equal to the one-hot of the oracle's chunk codes (utils.py:56-89, 342-347)."""
    import torch
    from seq2squiggle_amd.dataloader import PoreDataModule
    from seq2squiggle_amd import chunker
    from oracle import s2s_oracle as O
    rng = np.random.default_rng(2)
    reads = [("".join(rng.choice(list("ACGTN"), int(n))), f"r{i}") for i, n in enumerate([5, 9, 40, 41, 300, 8, 77])]
    cfg = {"seq_kmer": 9, "max_dna_len": 16}
    dm = PoreDataModule(cfg, total_l=123, data_dir=reads, batch_size=7)
    dm.setup("predict")
    loader = dm.predict_dataloader()
    assert len(loader) == 123 and len(loader.dataset) == 123
    batches = list(loader)
    want = [(name, chunker.encode_read(seq, 9)) for seq, name in reads]
    ids = [i for b in batches for i in b[0]]
    assert ids == [name for name, (b, nv) in want for _ in range(b.shape[0])] and "r0" not in ids and "r5" not in ids
    assert all(len(b[0]) == 7 for b in batches[:-1]) and 0 < len(batches[-1][0]) <= 7
    assert np.array_equal(torch.cat([b[1] for b in batches]).numpy(), np.concatenate([b for _, (b, nv) in want if b.shape[0]]))
    hot = torch.cat([b[1] for b in PoreDataModule(cfg, 1, reads, batch_size=16, onehot=True).predict_dataloader()])
    codes = np.concatenate([O.encode_read(seq, 9) for seq, _ in reads if len(seq) >= 9])
    assert hot.dtype == torch.float16 and torch.equal(hot.float(), O.one_hot(codes).reshape(hot.shape))
    with pytest.raises(NotImplementedError):
        dm.setup("fit")
