"""Container validation against the real libraries (tools/validate_containers.py): runs when pyslow5 / pod5 are importable,
SKIPS otherwise (they are absent from the build image -- reference signal_io.py:96-101, 167-171, 268-282 writes through them).
The CPU part pins what the tool writes against the in-tree readers, so that a later run with the libraries checks the same
files a user would check."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT

spec = importlib.util.spec_from_file_location("validate_containers", os.path.join(ROOT, "tools", "validate_containers.py"))
V = importlib.util.module_from_spec(spec)
spec.loader.exec_module(V)


@pytest.mark.parametrize("rec,sig", V.BLOW5_VARIANTS)
def test_tool_blow5_files_round_trip_in_tree(tmp_path, rec, sig):
    from seq2squiggle_amd import signal_io
    p = str(tmp_path / "v.blow5")
    V.write_blow5(p, rec, sig)
    ids, dac, offs = V.synthetic_reads()
    _, recs = signal_io.read_blow5(p)
    assert len(recs) == len(ids)
    for i, r in enumerate(recs):
        assert np.array_equal(r["signal"], dac[offs[i]:offs[i + 1]]) and r["read_number"] == i


@pytest.mark.parametrize("sig", V.POD5_VARIANTS)
def test_tool_pod5_files_round_trip_in_tree(tmp_path, sig, monkeypatch):
    from seq2squiggle_amd import pod5_io
    monkeypatch.setenv("S2S_POD5_SIGNAL", sig)
    p = str(tmp_path / "v.pod5")
    V.write_pod5(p, sig)
    ids, dac, offs = V.synthetic_reads()
    reads = pod5_io.read_pod5(p)["reads"]
    assert len(reads) == len(ids)
    for i, r in enumerate(reads):
        assert np.array_equal(r["signal"], dac[offs[i]:offs[i + 1]]) and r["read_number"] == i
    assert sum(1 for _ in pod5_io.iter_pod5(p)) == len(ids)


def test_blow5_against_pyslow5(tmp_path):
    pytest.importorskip("pyslow5")
    for rec, sig in V.BLOW5_VARIANTS:
        p = str(tmp_path / f"v_{rec}_{sig}.blow5")
        V.write_blow5(p, rec, sig)
        assert V.check_blow5(p) == []
    p = str(tmp_path / "v.slow5")
    V.write_blow5(p, "none", "none")
    assert V.check_blow5(p) == []


def test_pod5_against_libpod5(tmp_path):
    pytest.importorskip("pod5")
    for sig in V.POD5_VARIANTS:
        p = str(tmp_path / f"v_{sig}.pod5")
        V.write_pod5(p, sig)
        assert V.check_pod5(p) == []
