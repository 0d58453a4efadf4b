"""`predict` without `-m`: the cache half of the reference's get_saved_weights (inference.py:85-149) -- a reference user whose
weights were downloaded once runs offline; the download half needs the network and is out of scope."""
import logging
import os

import pytest

from seq2squiggle_amd import inference as I


def reference_pick(names, profile_name, version="0.3.4"):
    """The reference's loop (inference.py:108-149) restated on a list of file names in visiting order -> picked name or None."""
    import re
    keyword = "R10" if profile_name.startswith("dna-r10") else "R9" if profile_name.startswith("dna-r9") else None
    pick, score = None, 0
    for filename in names:
        root, ext = os.path.splitext(filename)
        if ext == ".ckpt":
            file_version = tuple(g for g in re.match(r".*@v(\d+).(\d+).(\d+)", root).groups())
            m = [i == j for i, j in zip(version, file_version)]
            match = sum(m) if m[0] else 0
            if match > score and keyword and keyword in root:
                pick, score = filename, match
    return pick


@pytest.fixture
def cache(tmp_path, monkeypatch):
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path))
    d = tmp_path / "seq2squiggle"
    d.mkdir()
    return d


def test_cache_dir_follows_the_xdg_rule(tmp_path, monkeypatch):
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path))
    assert I.user_cache_dir("seq2squiggle") == str(tmp_path / "seq2squiggle")
    monkeypatch.delenv("XDG_CACHE_HOME")
    monkeypatch.setenv("HOME", str(tmp_path / "home"))
    assert I.user_cache_dir("seq2squiggle") == str(tmp_path / "home" / ".cache" / "seq2squiggle")


@pytest.mark.parametrize("order", [0, 1])
def test_picks_what_the_references_loop_picks(cache, monkeypatch, order, caplog):
    names = ["R10@v0.3.4.ckpt", "R10@v0.2.0.ckpt", "R9@v0.3.4.ckpt", "notes.txt"]
    for n in names:
        (cache / n).write_bytes(b"")
    visit = names if order == 0 else names[::-1]
    real = os.listdir
    monkeypatch.setattr(os, "listdir", lambda d: list(visit) if str(d) == str(cache) else real(d))
    with caplog.at_level(logging.INFO, logger="seq2squiggle"):
        got = I.get_saved_weights("dna-r10-prom")
    # both R10 files score 1 in the reference's comparison (its version is a string: '0' == major, '.' never equals the minor, '3' is
    # compared with the PATCH), so the first one visited stays
    assert os.path.basename(got) == reference_pick(visit, "dna-r10-prom") == [n for n in visit if n.startswith("R10")][0]
    assert os.path.dirname(got) == str(cache)
    text = caplog.text
    assert "Weights file path is not provided." in text and "Detected R10.4.1 chemistry profile." in text
    assert "Found matching weights in local cache" in text
    assert os.path.basename(I.get_saved_weights("dna-r9-min")) == reference_pick(visit, "dna-r9-min") == "R9@v0.3.4.ckpt"
    assert os.path.basename(I.get_saved_weights("dna-r9-prom")) == "R9@v0.3.4.ckpt"
    # neither keyword: the reference goes to the network with "latest weights" -- here that is the error naming the directory
    assert reference_pick(visit, "rna-004-prom") is None
    with pytest.raises(FileNotFoundError) as e:
        I.get_saved_weights("rna-004-prom")
    assert str(cache) in str(e.value) and "--model" in str(e.value)


def test_version_scoring_is_the_references(cache, monkeypatch):
    names = ["R10@v0.3.4.ckpt", "R10@v1.3.4.ckpt", "R10-hac@v0.1.3.ckpt", "R10@v0.3.0.ckpt"]
    for n in names:
        (cache / n).write_bytes(b"")
    real = os.listdir
    for visit in (names, names[::-1]):
        monkeypatch.setattr(os, "listdir", lambda d, v=visit: list(v) if str(d) == str(cache) else real(d))
        # patch '3' equals the third CHARACTER of "0.3.4": score 2, the only strictly better candidate; another major never counts
        assert os.path.basename(I.get_saved_weights("dna-r10-min")) == reference_pick(visit, "dna-r10-min") == "R10-hac@v0.1.3.ckpt"


def test_empty_or_missing_cache_is_a_clear_error(cache, tmp_path, monkeypatch):
    (cache / "R10-without-version.ckpt").write_bytes(b"")          # (the reference's loop dies on this name; here it is skipped)
    with pytest.raises(FileNotFoundError) as e:
        I.get_saved_weights("dna-r10-prom")
    assert str(cache) in str(e.value)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "fresh"))
    with pytest.raises(FileNotFoundError):
        I.get_saved_weights("dna-r10-prom")
    assert (tmp_path / "fresh" / "seq2squiggle").is_dir()          # created, as the reference does (inference.py:105)


def test_inference_run_resolves_the_cache_before_anything_touches_the_gpu(cache, tmp_path):
    """saved_weights=None goes through get_saved_weights: with an empty cache the FileNotFoundError names the directory."""
    from seq2squiggle_amd.cli import set_config
    with pytest.raises(FileNotFoundError) as e:
        I.inference_run(config=set_config(None), saved_weights=None, fasta="x.fa", read_input=False, n=1, r=100, c=-1,
                        out=str(tmp_path / "o.blow5"), profile="dna-r10-prom", dwell_mean=None, dwell_std=0.0, noise_std=2.0,
                        noise_sampling=True, duration_sampling=True, distr="expon", predict_batch_size=1024,
                        export_every_n_samples=1000000, sample_rate=None, bps=None, digitisation=None, range_val=None,
                        offset_mean=None, offset_std=None, median_before_mean=None, median_before_std=None, min_noise=0.0,
                        min_duration=3, min_read_len=30, preserve_read_ids=False, seed=1)
    assert str(cache) in str(e.value)
