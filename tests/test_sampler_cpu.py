"""Host front end: FASTA/FASTQ reader and the read sampler against read sets drawn by
the reference's own `sampling` for the same seeds (tools/make_goldens.py: sampler_goldens)."""
import gzip
import hashlib
import os
import random

import numpy as np
import pytest

from seq2squiggle_amd import utils as U
from conftest import GOLDEN, ROOT, load_npz

LAMBDA = os.path.join(GOLDEN, "example_lambda_genome.fasta")


def test_read_fasta_and_fastq(tmp_path):
    reads = list(U.read_fasta(os.path.join(GOLDEN, "example_test.fasta")))
    assert len(reads) == 7 and reads[0][1] == "sequenceID-003-GAC-repeat" and len(reads[0][0]) == 96
    fa = tmp_path / "m.fa"
    fa.write_text(">r1 some description\nACGT\nacgtNN\n\n>r2\nTTTT\n")
    assert list(U.read_fasta(str(fa))) == [("ACGTacgtNN", "r1"), ("TTTT", "r2")]
    fq = tmp_path / "m.fastq.gz"
    with gzip.open(fq, "wt") as f:
        f.write("@q1 desc\nACGTAC\n+\nIIIIII\n@q2\nGG\n+q2\n@@\n")
    assert list(U.read_fasta(str(fq))) == [("ACGTAC", "q1"), ("GG", "q2")]


def test_process_genome():
    g, n = U.process_genome("acgtRYn-ACGT")
    assert g == "ACGTNNNNACGT" and n == 12


CASES = ["lambda_expon_dna", "lambda_beta_dna", "lambda_gamma_rna", "two_contigs_cov"]


@pytest.mark.parametrize("name", CASES)
def test_sampler_matches_reference_read_sets(name):
    g = load_npz("sampler.npz")
    n, r, c, seed = (int(x) for x in g[name + "__args"])
    distr, prof = (str(x) for x in g[name + "__distr_profile"])
    lam = next(U.read_fasta(LAMBDA))[0]
    if name == "two_contigs_cov":
        seqs, lens = zip(*[U.process_genome(s) for s in (lam[:30000], str(g["two_contigs__seq1"]))])
    else:
        seqs, lens = zip(*[U.process_genome(lam)])
    random.seed(seed)
    cfg = {"max_dna_len": 16}
    reads, total_l = U.sample_reads_from_reference(list(seqs), list(lens), n, r, c, cfg, "x.fasta", seed, False, distr, prof, 30)
    reads = [rd for rd, _ in reads]
    assert [len(x) for x in reads] == g[name + "__lens"].tolist()
    assert [hashlib.sha1(x.encode()).hexdigest() for x in reads] == [str(x) for x in g[name + "__sha"]]
    assert total_l == int(g[name + "__total_l"])


def test_vectorised_length_draws_equal_the_per_seed_scipy_call():
    """draw_expon_dis_many re-implements RandomState(seed) + scipy's expon.rvs for a vector of seeds: bit-identical."""
    seeds = np.concatenate([np.arange(0, 300), 42 + 21 * np.arange(2000), np.array([2 ** 31 - 1, 2 ** 32 - 1, 123456789])])
    for mean in (5000, 150, 20000):
        fast = U.draw_expon_dis_many(mean, seeds, 48502)
        ref = np.array([int(U.draw_length("expon", mean, int(sd), 48502)) for sd in seeds[::7]])
        assert np.array_equal(fast[::7], ref)


def test_sampler_argument_errors():
    cfg = {"max_dna_len": 16}
    with pytest.raises(ValueError):
        U.sample_reads_from_reference(["ACGT" * 100], [400], -1, 100, -1, cfg, "x", 1)
    with pytest.raises(ValueError):
        U.sample_reads_from_reference(["ACGT" * 100], [400], 5, 100, 3, cfg, "x", 1)
    with pytest.raises(ValueError):
        U.sample_reads_from_reference(["ACGT" * 100], [400], 5, 0, -1, cfg, "x", 1)


def test_get_reads_read_mode(tmp_path):
    cfg = {"max_dna_len": 16}
    path = os.path.join(GOLDEN, "example_test.fasta")
    reads, total = U.get_reads(path, True, -1, 1000, -1, cfg, "expon", 3, "dna-r10-prom", 30)
    reads = list(reads)
    assert len(reads) == 7 and total == sum(len(s) for s, _ in reads)
    sampled, eff = U.get_reads(path, True, 5, 1000, -1, cfg, "expon", 3, "dna-r10-prom", 30)
    sampled = list(sampled)
    rng = random.Random(3)
    expect = [rng.choice(reads)[0] for _ in range(5)]
    assert [s for s, _ in sampled] == expect and eff == sum(round(len(s) / 16) for s in expect)


def test_profiles_match_reference_table():
    g = load_npz("profiles.npz")
    for n in g["names"]:
        p = U.get_profile(str(n))
        assert [p[k] for k in ("digitisation", "sample_rate", "bps", "range", "offset_mean", "offset_std",
                               "median_before_mean", "median_before_std")] == g[str(n) + "__profile"].tolist()
    assert U.get_profile("nope") is None
    cfg = U.update_config("dna-r9-min", {"seq_kmer": 9})
    assert cfg["seq_kmer"] == 6 and U.update_config("rna-004-prom", {})["seq_kmer"] == 9
    with pytest.raises(ValueError):
        U.update_config("dna-r7", {})
    p = U.update_profile(U.get_profile("dna-r10-prom"), sample_rate=4000, bps=None)
    assert p["sample_rate"] == 4000 and p["bps"] == 400


@pytest.mark.parametrize("block", [8192, 7])
@pytest.mark.parametrize("world", [1, 3, 8])
def test_sharded_sampling_equals_slices_of_the_full_read_set(world, block, monkeypatch):
    """sample_read_shard (lengths-only replay + strings for one range) gives every rank exactly its slice of the read set
    the unsharded sampler draws for the same seed, including reads with N (extra `random` draws) and rejected tries; also when
    the replay runs in several blocks and a rank resumes from the generator state kept in front of the block of its first read."""
    from seq2squiggle_amd.parallel import shard_reads
    monkeypatch.setattr(U, "_REPLAY_BLOCK", block)
    rng = np.random.default_rng(3)
    contig2 = "".join(rng.choice(list("ACGTN"), 20000, p=[.24, .24, .24, .24, .04]))
    seqs, lens = zip(*[U.process_genome(s) for s in (next(iter(U.read_fasta(LAMBDA)))[0][:30000], contig2)])
    args = (list(seqs), list(lens), 60, 3000, -1)
    random.seed(5)
    full, _ = U.sample_reads_from_reference(*args, {"max_dna_len": 16}, "x.fa", 5, False, "expon", "dna-r10-min", 30)
    full = [s for s, _ in full]
    got = []
    for rank in range(world):
        random.seed(5)
        box = {}

        def shard_of(ls, rank=rank):
            lo, hi, box["first"] = shard_reads(ls, 9, world)[rank]
            return lo, hi
        reads, ls = U.sample_read_shard(*args, 5, "expon", "dna-r10-min", 30, shard_of)
        assert ls == [len(s) for s in full]
        assert box["first"] == sum(-(-(max(L - 8, 0)) // 16) for L in ls[: len(got)])
        got += [s for s, _ in reads]
    assert got == full and any("N" in s for s in seqs)


@pytest.mark.parametrize("distr", ["expon", "gamma", "beta"])
@pytest.mark.parametrize("profile,r,n_frac,min_len", [("dna-r10-prom", 3000, 0.04, 30), ("rna-004-prom", 2500, 0.02, 30),
                                                      ("dna-r9-min", 300000, 0.0, 30), ("dna-r10-min", 400, 0.12, 300)])
def test_native_replay_equals_the_interpreter_draw_for_draw(profile, r, n_frac, min_len, distr):
    """s2s_sampler_replay_law (the rank skip-ahead of sharded runs) against sampling_iter, for each of the reference's three
    read-length laws (--distr expon | gamma | beta, utils.py:311-331: scipy on a generator seeded per (read, retry)): same accepted-read lengths, same
    index of the next read, and the SAME `random` generator state afterwards -- with N runs (extra draws), end-of-contig
    rejections, N-rich rejections (> 10 % N), reads whose 20 retries all fail (-r 300000 on 7-30 kb contigs) and an RNA
    profile (no strand draw, short reads allowed); also stopped part-way (stop_after) and resumed."""
    rng = np.random.default_rng(11)
    contigs = []
    for L in (30000, 20000, 7000):
        s = rng.choice(list("ACGT"), L)
        if n_frac:
            for lo in rng.integers(0, L - 400, 12):
                s[lo: lo + int(rng.integers(1, 400))] = "N"
            s[rng.random(L) < n_frac / 4] = "N"
        contigs.append("".join(s))
    seqs, lens = zip(*[U.process_genome(s) for s in contigs])
    seqs, lens = list(seqs), list(lens)
    total, seed, n = sum(lens), 77, (400 if distr == "expon" else 150)
    random.seed(seed)
    want = U.sampling(n, seqs, lens, r, seed, total, distr, profile, min_len, materialise=(0, 0))
    end_state = random.getstate()
    random.seed(seed)
    got = U.replay_sampler(n, seqs, lens, r, seed, total, distr, profile, min_len)
    assert got is not None, "native replay unavailable"
    assert got[0].tolist() == want and got[1] == n and random.getstate() == end_state
    assert len(want) <= n and (r < 300000 or len(want) < n)                         # the 300 kb case loses reads to 20 failed retries
    assert len(want) > 0 or (r == 300000 and distr != "expon")                      # (... all of them under the narrower laws)
    # stop part-way, then let the interpreter continue from there: the tail must be the same reads
    k = len(want) // 3
    random.seed(seed)
    full = U.sampling(n, seqs, lens, r, seed, total, distr, profile, min_len)
    random.seed(seed)
    part = U.replay_sampler(n, seqs, lens, r, seed, total, distr, profile, min_len, stop_after=k)
    assert part[0].tolist() == want[:k]
    rest = list(U.sampling_iter(n, seqs, lens, r, seed, total, distr, profile, min_len, first_read_i=part[1], n_accepted=k))
    assert rest == full[k:] and random.getstate() == end_state
    # cases the native path must decline (-> None): a seed beyond the scipy fast range, a fractional -r
    assert U.replay_sampler(n, seqs, lens, r, 2 ** 32 - 100, total, distr, profile, min_len) is None
    assert U.replay_sampler(n, seqs, lens, r + 0.5, seed, total, distr, profile, min_len) is None


def test_native_length_laws_equal_scipy():
    """s2s_length_law (numpy's legacy standard_gamma / beta on a per-seed MT19937, mirrored in the host library) against
    utils.draw_length (scipy expon / gamma / beta .rvs with random_state = seed), seeds across the 32-bit range."""
    from seq2squiggle_amd._lib import lib
    L = lib()
    rng = np.random.default_rng(5)
    seeds = list(range(300)) + [2 ** 32 - 1, 2 ** 31] + [int(x) for x in rng.integers(0, 2 ** 32, 300)]
    for law, name in enumerate(("expon", "gamma", "beta")):
        for seed in seeds:
            for r in (5000, 137):
                assert L.s2s_length_law(law, seed, float(r), 48502) == int(U.draw_length(name, r, seed, 48502)), (name, seed, r)


def test_native_fasta_parser_equals_the_line_loop(tmp_path, monkeypatch):
    """utils.read_fasta / preprocess_genome take plain FASTA files through the library's host-side parser (s2s_fasta_count /
    s2s_fasta_clean); the records must be those of the interpreter's line loop (pysam.FastxFile semantics, utils.py:290-308, and
    process_genome, 594-597) on well-formed and odd inputs alike; FASTQ and gzip keep the line loop."""
    import gzip
    from seq2squiggle_amd import utils as U

    def both(path):
        res = []
        for native in (True, False):
            if not native:
                monkeypatch.setattr(U, "_read_fasta_native", lambda p, map_acgtn=False, limit=0: None)
            try:
                res.append((list(U.read_fasta(path)), U.preprocess_genome(path)))
            except Exception as e:
                res.append(("raised", type(e).__name__))
            monkeypatch.undo()
        return res
    cases = {
        "plain": ">a desc here\nACGT\nacgtn\n>b\n\nGG\n>c\n", "crlf": ">a x\r\nACGT\r\nAC\r\n>b\r\nTT\r\n", "lead": "\n\n>a\nAC\n",
        "junk_before": "hello\n>a\nAC\n>b\nGT", "spaces": ">a\n ACGT \nAC GT\t\n>b  \nTT\n", "empty": "", "blank": "\n\n",
        "noname": ">\nACGT\n> \nGG\n", "gt_inside": ">a\nAC>GT\n>b\nTT\n", "fastq": "@q1\nACGT\n+\nIIII\n",
        "no_final_newline": ">a\nACGT", "only_header": ">a", "iupac": ">a\nACGTRYKMnnxx-*\n", "tabname": ">a\tdesc\nAC\n",
        "lone_cr": "\r> >C\nAC\rGT\n>b\nTT\n", "lone_cr_seq": ">a\nAC\rGT\n", "indent_header": "  >a\nAC\n", "blank_first": "  \n>a\nAC\n", "cr_only_lines": ">a\r\n\r\nAC\r\r\n",
    }
    for name, text in cases.items():
        p = tmp_path / f"{name}.fa"
        p.write_text(text, encoding="latin-1", newline="")
        native, loop = both(str(p))
        assert native == loop, (name, native, loop)
    for f in ("example_test.fasta", "example_lambda_genome.fasta"):
        native, loop = both(os.path.join(GOLDEN, f))
        assert native == loop and native[0] != "raised"
    rng = np.random.default_rng(0)
    for it in range(40):
        text = ""
        for i in range(int(rng.integers(1, 6))):
            L, w = int(rng.integers(0, 500)), int(rng.integers(1, 90))
            s = "".join(rng.choice(list("ACGTNacgtRY"), L))
            text += f">r{i} d{it}\n" + "".join(s[j:j + w] + ("\r\n" if it % 2 else "\n") for j in range(0, L, w)) + ("\n" if it % 3 == 0 else "")
        p = tmp_path / "rand.fa"
        p.write_text(text, newline="")
        native, loop = both(str(p))
        assert native == loop
    # four-line FASTQ through s2s_fastq_clean: same records as the line loop, odd files handed back to it (same exception or result)
    fq = {
        "fq_plain": "@q1 desc\nACGT\n+\nIIII\n@q2\nacgtn\n+q2\nIIIII\n", "fq_crlf": "@q1\r\nACGT\r\n+\r\nIIII\r\n", "fq_blank_between": "@a\nAC\n+\nII\n\n\n@b\nGT\n+\nII\n",
        "fq_no_final_newline": "@a\nAC\n+\nII", "fq_empty_seq": "@a\n\n+\n\n@b\nAC\n+\nII\n", "fq_truncated": "@a\nAC\n+\n", "fq_no_plus": "@a\nAC\nII\nII\n",
        "fq_bad_header": "@a\nAC\n+\nII\nxx\nAC\n+\nII\n", "fq_at_quality": "@a\nAC\n+\n@I\n@b\nGG\n+\nII\n", "fq_spaces": "@a  x\n AC GT \n+\nIIIIIII\n",
        "fq_lone_cr": "@a\nAC\rGT\n+\nIIIII\n", "fq_noname": "@\nAC\n+\nII\n",
    }
    for name, text in fq.items():
        p = tmp_path / f"{name}.fastq"
        p.write_text(text, encoding="latin-1", newline="")
        native, loop = both(str(p))
        assert native == loop, (name, native, loop)
    # bytes the byte-level parser and the text-mode loop would read differently -- UTF-8 names (the loop decodes them),
    # invalid UTF-8 (the loop raises), the separators 0x1c-0x1f and U+0085 (blanks for str.split), "\r\r\n" (two line ends) -- go to the loop
    raw = {
        "utf8_name.fastq": "@r\u00e9ad x\nAC\n+\nII\n".encode("utf-8"), "latin1_name.fastq": "@r\u00e9ad\nAC\n+\nII\n".encode("latin-1"),
        "fs_sep.fastq": b"@a\x1cb\nAC\n+\nII\n", "nel_sep.fastq": "@a\u0085b\nAC\n+\nII\n".encode("utf-8"), "crcrlf.fastq": b"@a\nAC\r\r\n+\nII\n",
        "utf8_name.fa": ">r\u00e9ad x\nAC\n".encode("utf-8"), "latin1_name.fa": ">r\u00e9ad\nAC\n".encode("latin-1"), "gs_sep.fa": b">a\x1db\nAC\n",
        "us_in_seq.fa": b">a\n\x1fAC\x1f\nGT\n", "crcrlf.fa": b">a\nAC\r\r\nGT\n", "nbsp_seq.fa": ">a\n\u00a0AC\n".encode("utf-8"),
    }
    for name, data in raw.items():
        p = tmp_path / name
        p.write_bytes(data)
        assert U._read_fasta_native(str(p)) is None, name
        native, loop = both(str(p))
        assert native == loop, (name, native, loop)
    assert list(U.read_fasta(str(tmp_path / "utf8_name.fastq"))) == [("AC", "r\u00e9ad")]
    assert list(U.read_fasta(str(tmp_path / "fs_sep.fastq"))) == [("AC", "a")]
    assert U._read_fasta_native(str(tmp_path / "fq_plain.fastq")) == [("ACGT", "q1"), ("acgtn", "q2")]
    assert U._read_fasta_native(str(tmp_path / "fq_no_plus.fastq")) is None
    for it in range(20):
        text = ""
        for i in range(int(rng.integers(1, 8))):
            sq = "".join(rng.choice(list("ACGTNacgt"), int(rng.integers(0, 300))))
            text += f"@r{i} d{it}\n{sq}\n+\n{'I' * len(sq)}\n".replace("\n", "\r\n" if it % 2 else "\n") + ("\n" if it % 3 == 0 else "")
        p = tmp_path / "rand.fq"
        p.write_text(text, newline="")
        native, loop = both(str(p))
        assert native == loop and native[0] != "raised"
    # the native path is really taken for plain FASTA, and not for gzip
    assert U._read_fasta_native(os.path.join(GOLDEN, "example_test.fasta")) is not None
    gz = tmp_path / "x.fa.gz"
    with gzip.open(gz, "wt") as f:
        f.write(">a\nACGT\n")
    assert U._read_fasta_native(str(gz)) is None and list(U.read_fasta(str(gz))) == [("ACGT", "a")]
    assert U._read_fasta_native(str(tmp_path / "fastq.fa")) == [("ACGT", "q1")]
    monkeypatch.setenv("S2S_FASTA_NATIVE_LIMIT", "10")                      # larger files stream through the line loop
    assert U._read_fasta_native(os.path.join(GOLDEN, "example_test.fasta")) is None


def test_native_replay_without_avx2_takes_the_same_draws():
    """The replay's eight-seeds-at-a-time seeding has an AVX2 and a plain build of the same loop; S2S_NO_AVX2 selects the plain
    one (a process-wide choice, hence the child process): both must give the interpreter's lengths and generator state."""
    import subprocess, sys
    code = r"""
import random, sys
sys.path.insert(0, %r)
import numpy as np
from seq2squiggle_amd import utils as U
rng = np.random.default_rng(2)
seqs = ["".join(rng.choice(list("ACGT"), L)) for L in (30000, 9000)]
lens = [len(s) for s in seqs]
random.seed(3)
want = [len(x) if isinstance(x, str) else x for x in U.sampling(300, seqs, lens, 2500, 3, sum(lens), "expon", "dna-r10-prom", 30, materialise=(0, 0))]
state_after = random.getstate()
random.seed(3)
got = U.replay_sampler(300, seqs, lens, 2500, 3, sum(lens), "expon", "dna-r10-prom", 30)
assert got is not None and got[0].tolist() == want and random.getstate() == state_after
print("REPLAY_OK", len(want))
""" % ROOT
    for extra in ({}, {"S2S_NO_AVX2": "1"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **extra), timeout=300)
        assert "REPLAY_OK" in r.stdout, r.stdout + r.stderr[-2000:]
