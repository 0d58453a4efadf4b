"""Property tests (hypothesis) of the host-side chunking and sharding against the oracle's restatement of the
reference (dataloader.py:358-398, utils.py:56-89, 342-356)."""
import numpy as np
from hypothesis import given, settings, strategies as st

from seq2squiggle_amd import chunker
from seq2squiggle_amd.parallel import shard_reads
from oracle import s2s_oracle as O

reads_st = st.lists(st.text(alphabet="ACGTNacgtnRYU_", min_size=0, max_size=200), min_size=0, max_size=12)


@settings(max_examples=150, deadline=None)
@given(reads=reads_st, k=st.sampled_from([6, 9]))
def test_packed_windows_equal_dense_windows(reads, k):
    """pack_reads + (chunk_start, n_valid) addresses exactly the bytes encode_reads materialises, read by read."""
    flat, cs, nv, rf = chunker.pack_reads(reads, k)
    bases, nv2, rf2 = chunker.encode_reads(reads, k)
    assert np.array_equal(nv, nv2) and np.array_equal(rf, rf2)
    assert rf[-1] == sum(chunker.n_chunks(len(r), k) for r in reads)
    nb = 16 + k - 1
    for b in range(len(nv)):
        assert cs[b] + nb <= flat.size
        # only the first n_valid + k - 1 bytes of a window are defined (the rest is never read by the kernel)
        assert np.array_equal(flat[cs[b]: cs[b] + nv[b] + k - 1], bases[b, : nv[b] + k - 1])


@settings(max_examples=100, deadline=None)
@given(read=st.text(alphabet="ACGTN", min_size=0, max_size=400), k=st.sampled_from([6, 9]))
def test_windows_decode_to_the_oracle_kmer_codes(read, k):
    """The raw-byte windows carry the same information as the reference's k-mer code array (via the oracle)."""
    bases, nv = chunker.encode_read(read, k)
    if len(read) < k:
        assert bases.shape[0] == 0
        return
    codes = O.encode_read(read, k)                       # [C, 16, k] letter codes, pad k-mers all "_"
    back, nv_back = chunker.codes_to_bases(codes)
    assert np.array_equal(nv, nv_back)
    for b in range(len(nv)):
        assert np.array_equal(bases[b, : nv[b] + k - 1], back[b, : nv[b] + k - 1])


@settings(max_examples=200, deadline=None)
@given(lens=st.lists(st.integers(min_value=0, max_value=20000), min_size=0, max_size=60), world=st.integers(1, 8),
       k=st.sampled_from([6, 9]))
def test_shards_partition_the_reads_and_number_the_chunks(lens, world, k):
    sh = shard_reads(lens, k, world)
    assert len(sh) == world and sh[0][0] == 0 and sh[-1][1] == len(lens)
    chunks = [chunker.n_chunks(L, k) for L in lens]
    for r in range(world):
        lo, hi, first = sh[r]
        assert lo <= hi and (r == 0 or sh[r - 1][1] == lo)
        assert first == sum(chunks[:lo])                  # global chunk index keys the RNG: independent of the split


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(1, 2 ** 31 - 1), r=st.integers(40, 4000), n=st.integers(1, 120),
       contigs=st.lists(st.integers(200, 6000), min_size=1, max_size=4), n_density=st.sampled_from([0.0, 0.01, 0.08, 0.3]),
       profile=st.sampled_from(["dna-r10-prom", "dna-r9-min", "rna-004-min"]), min_len=st.sampled_from([30, 150]),
       stop=st.integers(0, 40))
def test_native_sampler_replay_property(seed, r, n, contigs, n_density, profile, min_len, stop):
    """s2s_sampler_replay (the rank skip-ahead) walks exactly the draws of the reference sampler (utils.py:415-479) for arbitrary
    genomes, N densities, read lengths and profiles: same accepted lengths, same next read index, same `random` state -- whole,
    and stopped after `stop` accepted reads."""
    import random
    from seq2squiggle_amd import utils as U
    rng = np.random.default_rng(seed % 100003)
    seqs = []
    for L in contigs:
        s = rng.choice(list("ACGT"), L)
        s[rng.random(L) < n_density] = "N"
        seqs.append("".join(s))
    lens = [len(s) for s in seqs]
    total = sum(lens)
    seed = seed % (2 ** 31)                                           # inside the scipy fast range for any n
    random.seed(seed)
    want = U.sampling(n, seqs, lens, r, seed, total, "expon", profile, min_len, materialise=(0, 0))
    end_state = random.getstate()
    random.seed(seed)
    got = U.replay_sampler(n, seqs, lens, r, seed, total, "expon", profile, min_len)
    assert got is not None and got[0].tolist() == want and got[1] == n and random.getstate() == end_state
    k = min(stop, len(want))
    random.seed(seed)
    part = U.replay_sampler(n, seqs, lens, r, seed, total, "expon", profile, min_len, stop_after=k)
    assert part[0].tolist() == want[:k]
    rest = list(U.sampling_iter(n, seqs, lens, r, seed, total, "expon", profile, min_len, first_read_i=part[1], n_accepted=k,
                                materialise=(0, 0)))
    assert rest == want[k:] and random.getstate() == end_state
