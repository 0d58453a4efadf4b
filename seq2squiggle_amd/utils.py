"""Host-side helpers of the predict path: profiles, config patching, FASTA/FASTQ input, read
sampling, seeds, logging.

Mirrors the public names of the reference's utils.py so that `inference_run` reads the same:
get_profile (utils.py:129-215), update_profile (218-243), update_config (245-263), read_fasta
(290-308), draw_{gamma,beta,expon}_dis (311-331), sampling (415-479),
sample_reads_from_reference (495-582), preprocess_genome (608-638), get_reads (641-671),
set_seeds (722-741), setup_logging (687-719).  Training-only helpers (plots, parameter tables)
are out of scope.
"""
import gzip
import bisect
import itertools
import logging
import os
import math
import random
import sys
from typing import Dict, Generator, Iterable, List, Tuple
from uuid import uuid4

import numpy as np

logger = logging.getLogger("seq2squiggle")

_PROFILES = {
    # name: digitisation, sample_rate, bps, range, offset_mean, offset_std, median_before_mean, median_before_std
    "dna-r10-min": (8192, 5000, 400, 1536.598389, 13.380569389019, 16.311471649012, 202.15407438804, 13.406139241768),
    "dna-r10-prom": (2048, 5000, 400, 281.345551, -127.5655735, 19.377283387665, 189.87607393756, 15.788097978713),
    "dna-r9-min": (8192, 4000, 450, 1443.030273, 13.7222605, 10.25279688, 200.815801, 20.48933762),
    "dna-r9-prom": (2048, 4000, 450, 748.5801, -237.4102, 14.1575, 214.2890337, 18.0127916),
    "rna-004-min": (8192, 4000, 130, 1437.976685, 12.47686423863, 10.442126577137, 205.08496731088, 8.6671292866233),
    "rna-004-prom": (2048, 4000, 130, 299.432068, -259.421128, 16.010841823643, 189.87607393756, 15.788097978713),
}
_PROFILE_KEYS = ("digitisation", "sample_rate", "bps", "range", "offset_mean", "offset_std", "median_before_mean",
                 "median_before_std")


def get_profile(profile: str) -> Dict[str, float]:
    """Profile dict for the writers (reference: utils.py:129-215).  Unknown name -> logged error, None."""
    if profile in _PROFILES:
        return dict(zip(_PROFILE_KEYS, _PROFILES[profile]))
    logger.error(f"Incorrect value for profile: {profile}")
    return None


def update_profile(profile_dict: dict, **overrides) -> dict:
    """CLI overrides of profile entries (reference utils.py:218-243): a None leaves the entry alone, an unknown key is
    reported and ignored."""
    for key in overrides.keys() - profile_dict.keys():
        logger.warning(f"Warning: {key} is not a valid key in the profile")
    profile_dict.update({k: v for k, v in overrides.items() if v is not None and k in profile_dict})
    return profile_dict


_KMER_OF_CHEMISTRY = (("dna-r10", 9), ("rna-004", 9), ("dna-r9", 6))


def update_config(profile_name: str, config: dict) -> dict:
    """The k-mer size is a property of the chemistry, not of the YAML (reference utils.py:245-263)."""
    for prefix, k in _KMER_OF_CHEMISTRY:
        if profile_name.startswith(prefix):
            config["seq_kmer"] = k
            return config
    raise ValueError(f"Unsupported profile name: {profile_name}. Expected 'dna-r10' or 'dna-r9' prefix.")


# --------------------------------------------------------------------------------------- FASTA / FASTQ
def _open_text(path):
    path = str(path)
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path, "r")


def _read_fasta_native(path, map_acgtn: bool = False, limit: int = 4 << 30):
    """Plain (not gzipped) FASTA files through the library's host-side parser (s2s_fasta_clean): the same (sequence, name)
    pairs as the line loop of read_fasta -- with map_acgtn also process_genome's upper-casing and non-ACGT -> N -- without the
    interpreter touching a line (a 100 Mb reference: 0.1 s instead of 0.7 s, which every rank of a sharded run pays before its
    first kernel); four-line FASTQ files through s2s_fastq_clean.  None: not applicable (gzip, files over 4 GB, odd FASTQ, library
    not built), the caller falls back to the line loop."""
    path = str(path)
    try:
        if path.endswith(".gz"):
            return None
        size = os.path.getsize(path)
        if size == 0:
            return []
        if size > int(os.environ.get("S2S_FASTA_NATIVE_LIMIT", limit)):
            return None                      # the records of this path are all in memory at once: larger files keep the streaming loop
        from ._lib import lib
        L = lib()
        if not all(hasattr(L, f) for f in ("s2s_fasta_count", "s2s_fasta_clean", "s2s_fastq_clean")):
            return None                      # an older library selected with S2S_HIP_LIB (_lib.py binds leniently then): the line loop
        data = np.memmap(path, dtype=np.uint8, mode="r")
    except Exception:
        return None
    n_rec = int(L.s2s_fasta_count(data.ctypes.data, size))
    clean = L.s2s_fasta_clean
    if n_rec == -2:                          # not FASTA: four-line FASTQ records, or something the line loop has to judge
        n_rec = int(L.s2s_fastq_clean(data.ctypes.data, size, 0, None, None, None, 0))
        clean = L.s2s_fastq_clean
    if n_rec < 0:
        return None
    out = np.empty(size, np.uint8)
    seq_offs = np.zeros(n_rec + 1, np.int64)
    names = np.zeros(2 * n_rec + 2, np.int64)
    got = int(clean(data.ctypes.data, size, 1 if map_acgtn else 0, out.ctypes.data, seq_offs.ctypes.data,
                    names.ctypes.data, n_rec))
    if got != n_rec:
        return None
    recs = []
    for r in range(n_rec):
        recs.append((out[seq_offs[r]:seq_offs[r + 1]].tobytes().decode("latin-1"),
                     data[names[2 * r]:names[2 * r + 1]].tobytes().decode("latin-1")))
    return recs


def read_fasta(path: str, rna: bool = False) -> Generator[Tuple[str, str], None, None]:
    """(sequence, name) for every FASTA or FASTQ record (reference: pysam.FastxFile, utils.py:290-308).
    name = header up to the first whitespace, as pysam's `entry.name`; multi-line FASTA is joined."""
    fast = _read_fasta_native(path, limit=256 << 20)     # read files stream (line loop below) unless they are small; references
    if fast is not None:                                  # (preprocess_genome) are held whole anyway and go native up to 4 GB
        yield from fast
        return
    with _open_text(path) as fh:
        name, seq, mode = None, [], None
        it = iter(fh)
        for line in it:
            line = line.rstrip("\r\n")
            if not line:
                continue
            if mode is None:
                mode = "fq" if line[0] == "@" else "fa"
            if mode == "fa":
                if line[0] == ">":
                    if name is not None:
                        yield "".join(seq), name
                    parts = line[1:].split()
                    name, seq = (parts[0] if parts else ""), []
                elif name is not None:
                    seq.append(line.strip())
            else:
                if line[0] != "@":
                    raise ValueError(f"{path}: malformed FASTQ record header: {line[:40]!r}")
                parts = line[1:].split()
                rname = parts[0] if parts else ""
                s = next(it).rstrip("\r\n")
                plus = next(it)
                next(it)
                if not plus.startswith("+"):
                    raise ValueError(f"{path}: malformed FASTQ record {rname}")
                yield s, rname
        if mode == "fa" and name is not None:
            yield "".join(seq), name


# --------------------------------------------------------------------------------------- read sampling
# Read-length laws (reference utils.py:311-331): a scipy distribution fitted to real runs, rescaled so that its mean is the
# requested -r.  name -> (distribution, shape/loc/scale arguments, mean of the fitted law)
# (scipy.stats is imported when a length is drawn through it -- 0.3 - 0.6 s that a rank whose sampler is replayed natively never pays)
_LENGTH_LAWS = {
    "expon": ("expon", dict(loc=213.98910256668592, scale=6972.5319847131141), 7106.0),
    "beta": ("beta", dict(a=1.778, b=7.892, loc=316.758, scale=34191.257), 6615.0),
    "gamma": ("gamma", dict(a=6.3693711, loc=0.53834893), 4.39),
}


_NATIVE_LAWS = {"expon": 0, "gamma": 1, "beta": 2}     # s2s_sampler_replay_law's `law`: numpy's legacy samplers mirrored draw for draw


def draw_length(distr: str, mean, seed, total_len):
    """One read length: a single variate from a generator seeded with `seed` (scipy builds a fresh legacy RandomState per
    call), truncated to an integer and clipped to [1, total_len]."""
    import scipy.stats as st
    law, args, fitted_mean = _LENGTH_LAWS[distr]
    x = getattr(st, law).rvs(size=1, random_state=seed, **args)[0]
    return np.clip(int(x * mean / fitted_mean), 1, total_len)


def _mt19937_first_double(seeds: np.ndarray) -> np.ndarray:
    """random_sample() of np.random.RandomState(seed) for a whole vector of integer seeds at once: init_genrand, the
    twist of state words 0 and 1 (which only needs words 0-2, 397, 398) and the 53-bit double of the legacy generator.
    uint32 arrays wrap modulo 2^32 like the C code; the recurrence runs in place on one L1-sized vector."""
    x = (np.asarray(seeds, dtype=np.int64) & 0xFFFFFFFF).astype(np.uint32)
    t = np.empty_like(x)
    keep = {0: x.copy()}
    mul = np.uint32(1812433253)
    for i in range(1, 399):
        np.right_shift(x, 30, out=t)
        t ^= x
        t *= mul
        t += np.uint32(i)
        x, t = t, x
        if i in (1, 2, 397, 398):
            keep[i] = x.copy()

    def twist(a, b, c):
        y = (a & np.uint32(0x80000000)) | (b & np.uint32(0x7FFFFFFF))
        return c ^ (y >> 1) ^ np.where(y & np.uint32(1), np.uint32(0x9908B0DF), np.uint32(0))

    def temper(y):
        y = y ^ (y >> 11)
        y = y ^ ((y << 7) & np.uint32(0x9D2C5680))
        y = y ^ ((y << 15) & np.uint32(0xEFC60000))
        return y ^ (y >> 18)

    a = temper(twist(keep[0], keep[1], keep[397])) >> 5
    b = temper(twist(keep[1], keep[2], keep[398])) >> 6
    return (a.astype(np.float64) * 67108864.0 + b.astype(np.float64)) / 9007199254740992.0


def draw_expon_dis_many(mean, seeds: np.ndarray, total_len) -> np.ndarray:
    """draw_length("expon", ...) for a vector of seeds, bit-identical to the per-seed scipy call (tests/test_sampler_cpu.py): scipy
    seeds a fresh legacy RandomState per call and takes loc + scale * standard_exponential(), i.e. -log(1 - double)
    with the C library's log (math.log; np.log may differ in the last bit)."""
    e = np.array([-math.log(1.0 - x) for x in _mt19937_first_double(seeds)])
    _, args, fitted_mean = _LENGTH_LAWS["expon"]
    sample = ((args["loc"] + args["scale"] * e) * mean / fitted_mean).astype(int)
    return np.clip(sample, 1, total_len)


_COMPLEMENT = str.maketrans("ATCG", "TAGC")


def locate(contig_ends, position):
    """Genome-wide coordinate -> (contig index, offset inside it); contig_ends = running sum of the contig lengths
    (reference utils.py:359-372 walks the contigs instead)."""
    i = bisect.bisect_right(contig_ends, position)
    return i, position - (contig_ends[i - 1] if i else 0)


def read_check(read, read_length, read_i, profile, min_read_len=30, may_have_n=True):
    """utils.py:381-400: DNA reads must have the full drawn length (end-of-contig rejection), be at least
    min_read_len long and carry at most 10 % N (may_have_n = False: the contig is known to hold none)."""
    if profile.startswith("dna") and len(read) != read_length:
        return False
    if len(read) < min_read_len:
        return False
    if may_have_n and read.count("N") > 0.1 * read_length:
        return False
    return True


def fill_unknown_bases(read: str) -> str:
    """Every N becomes a uniformly drawn base: one draw from the global `random` stream per N, left to right (the order the
    reference consumes them in, utils.py:402-403)."""
    head, *rest = read.split("N")
    return head + "".join(random.choice("ACGT") + piece for piece in rest)


def reverse_complement(f):
    return f.translate(_COMPLEMENT)[::-1]


def sampling(num_seqs, genome_seqs, genome_lens, r, seed, total_len, distr, profile, min_read_len=30, max_retries=20,
             materialise=None):
    """The whole read set as a list (the reference's return value); see sampling_iter."""
    return list(sampling_iter(num_seqs, genome_seqs, genome_lens, r, seed, total_len, distr, profile, min_read_len, max_retries,
                              materialise))


def replay_sampler(num_seqs, genome_seqs, genome_lens, r, seed, total_len, distr, profile, min_read_len=30, max_retries=20,
                   first_read_i=0, stop_after=-1, want_lengths=True, _cache=None):
    """The draws of sampling_iter() for reads first_read_i.. without building a read, in native code (s2s_sampler_replay in
    libs2s_hip.so, host only): advances the global `random` generator exactly as sampling_iter would and returns
    (lengths of the accepted reads | None, index of the next read to attempt) -- or None when the native path does not
    apply (a non-integer -r, a seed outside the scipy fast range, a genome of 2 Gb or more, library not loadable): the
    caller then replays in Python.  stop_after: stop once that many reads have been accepted."""
    import ctypes as C
    genome_total = sum(genome_lens)
    if distr not in _NATIVE_LAWS or r <= 0 or int(r) != r or not (0 <= seed and seed + num_seqs * (max_retries + 1) < 2 ** 32) or genome_total >= 2 ** 31:
        return None
    try:
        from ._lib import lib
        L = lib()
    except (RuntimeError, OSError):
        return None
    if not hasattr(L, "s2s_sampler_replay_law"):      # an older library selected with S2S_HIP_LIB: replay in the interpreter
        return None
    version, internal, gauss = random.getstate()
    if version != 3 or len(internal) != 625:
        return None
    state = np.array(internal, dtype=np.uint32)
    ends = np.array(list(itertools.accumulate(genome_lens)), dtype=np.int64)
    cache = _cache if _cache is not None else {}
    if "n_pos" not in cache:                       # per contig: the sorted offsets of its N bases (empty for most contigs)
        cache["n_pos"] = [np.flatnonzero(np.frombuffer(g_.encode("latin-1"), np.uint8) == ord("N")).astype(np.int64)
                          if "N" in g_ else None for g_ in genome_seqs]
    n_pos = cache["n_pos"]
    ptrs = (C.c_void_p * len(n_pos))(*[None if p_ is None else p_.ctypes.data for p_ in n_pos])
    counts = np.array([0 if p_ is None else p_.size for p_ in n_pos], dtype=np.int64)
    n_left = max(num_seqs - first_read_i, 0)
    out = np.empty(n_left if stop_after < 0 else min(n_left, stop_after), np.int32) if want_lengths else None
    nxt = C.c_int64(0)
    got = L.s2s_sampler_replay_law(state.ctypes.data, ends.ctypes.data, len(ends), C.cast(ptrs, C.c_void_p), counts.ctypes.data, int(num_seqs),
                                   int(first_read_i), int(r), int(seed), int(total_len), int(profile.startswith("dna")),
                                   int(min_read_len), int(max_retries), int(stop_after), _NATIVE_LAWS[distr],
                                   None if out is None else out.ctypes.data, C.byref(nxt))
    if got < 0:
        return None
    random.setstate((3, tuple(int(x) for x in state), gauss))
    return (out[:got] if out is not None else None), int(nxt.value)


def sampling_iter(num_seqs, genome_seqs, genome_lens, r, seed, total_len, distr, profile, min_read_len=30, max_retries=20,
                  materialise=None, first_read_i=0, n_accepted=0):
    """Sample reads from the reference (utils.py:415-479), one at a time: the predict loop pulls reads as it packs batches,
    so the GPU starts after the first batch's worth of draws instead of after all of them.  The order of draws from the global `random`
    stream (start position, strand, N replacement) and the per-(read, retry) scipy seed
    `seed + read_i * (max_retries + 1) + retries` are those of the reference, so a seed selects the same
    read set.

    materialise = (lo, hi): only the accepted reads lo..hi-1 are built as strings, the others are returned as their
    LENGTH (an int) -- every draw is still made, so the stream and the read set are unchanged; a rank of a sharded run
    pays the string work (slice copy, reverse complement) only for its own reads.

    first_read_i / n_accepted: continue a run whose earlier reads were consumed elsewhere (replay_sampler): the global
    `random` state must be the one in front of read first_read_i, n_accepted the number of reads accepted before it."""
    total_genome_len = sum(genome_lens)
    # the first-try lengths of a block of reads in one vectorised pass (the per-seed scipy call costs ~100 us and is what
    # the reference's sampler spends its time in); retries and the other distributions go through scipy one by one
    fast = distr == "expon" and r > 0 and 0 <= seed and seed + num_seqs * (max_retries + 1) < 2 ** 32
    blocks = {}                        # retry level -> (first read of the block, lengths); level 0 in blocks of 8192 reads

    def fast_length(read_i, level):
        lo, vals = blocks.get(level, (0, ()))
        if not (lo <= read_i < lo + len(vals)):
            lo = read_i
            idx = np.arange(read_i, min(num_seqs, read_i + (8192 if level == 0 else 2048)), dtype=np.int64)
            vals = draw_expon_dis_many(r, seed + idx * (max_retries + 1) + level, total_len)
            blocks[level] = (lo, vals)
        return int(vals[read_i - lo])

    contig_ends = list(itertools.accumulate(genome_lens))
    contig_has_n = ["N" in g_ for g_ in genome_seqs]      # one scan per contig instead of one per read
    is_dna = profile.startswith("dna")

    def attempt(read_i, retry):
        """One try at read `read_i`: the read (N's already replaced), its strand, or None when it is rejected.  Draw order
        on the global `random` stream: start position, strand (DNA only), then the N replacements of an accepted read."""
        where, offset = locate(contig_ends, random.randint(0, total_genome_len - 1))
        genome = genome_seqs[where]
        if r <= 0:
            length = len(genome)
        elif fast:
            length = fast_length(read_i, retry)
        else:
            length = int(draw_length(distr, r, seed + read_i * (max_retries + 1) + retry, total_len))
        read = genome[offset:offset + length]
        strand = random.choice("+-") if is_dna else "+"
        if not read_check(read, length, read_i, profile, min_read_len, contig_has_n[where]):
            return None
        return (fill_unknown_bases(read) if contig_has_n[where] and "N" in read else read), strand

    for read_i in range(first_read_i, num_seqs):
        for retry in range(max_retries):
            got = attempt(read_i, retry)
            if got is None:
                continue
            read, strand = got
            if materialise is not None and not (materialise[0] <= n_accepted < materialise[1]):
                yield len(read)
            else:
                yield reverse_complement(read) if strand == "-" else read
            n_accepted += 1
            break
        else:
            logger.debug(f"Failed to sample a valid read after {max_retries} retries for read {read_i}. Skipping this read.")


class CountedReads:
    """An iterator over (sequence, read_id) pairs that knows how many are still to come (`operator.length_hint`): the predict
    loop shortens its last super-batches with it."""

    def __init__(self, pairs, count: int):
        self._pairs = iter(pairs)
        self._left = int(count)

    def __iter__(self):
        return self

    def __next__(self):
        item = next(self._pairs)
        self._left -= 1
        return item

    def __length_hint__(self):
        return max(self._left, 0)


def yield_reads(reads, count: int = None):
    pairs = ((read, str(uuid4())) for read in reads)
    if count is None and hasattr(reads, "__len__"):
        count = len(reads)
    return pairs if count is None else CountedReads(pairs, count)


def export_fasta(read_l, fasta):
    file_name, _ = os.path.splitext(fasta)
    out_file = f"{file_name}_reads.fasta"
    with open(out_file, "w") as f:
        for read in read_l:
            f.write(f">{uuid4()}\n{read}\n")
    return out_file


def _check_sampling_args(n, r, c):
    """utils.py:528-541."""
    if n <= 0 and c <= 0:
        raise ValueError("You need to specify the coverage c or the number of reads n")
    if n != -1 and c != -1:
        raise ValueError("You can only either specify the coverage c or the number of reads, but not both")
    if r <= 0:
        raise ValueError("You need to specify the read length r")


_REPLAY_BLOCK = 8192


def sample_read_shard(genome_seqs, genome_lens, n, r, c, seed, distr, profile, min_read_len, shard_of):
    """One rank's share of the read set sample_reads_from_reference would draw: pass 1 replays the sampler for the read
    LENGTHS only, `shard_of(lengths)` -> (lo, hi) picks the rank's contiguous range, pass 2 moves the `random` generator in
    front of read lo, then the rank's reads are built lazily (the predict loop pulls them while it packs batches) and the
    iterator stops behind read hi-1.  Both replays run in native code when the reference's default length law is in use
    (replay_sampler: ~1 us per read -- most of it seeding scipy's per-read generator -- instead of 7 in the interpreter; 300,000
    reads of BASELINE configs[4]: 0.3 s, which every rank pays before its first kernel; pass 2 costs at most one block);
    otherwise in the interpreter, draw for draw.
    -> (iterator over reads lo..hi-1 as (seq, uuid) pairs, all read lengths)."""
    _check_sampling_args(n, r, c)
    total_len = sum(len(seq) for seq in genome_seqs)
    seq_num = n if n != -1 else round(c * total_len / r)
    state = random.getstate()
    cache = {}
    # pass 1 in blocks of _REPLAY_BLOCK accepted reads, the generator state kept in front of every block: pass 2 then starts at
    # the block that holds read lo instead of at read 0 (the last rank of eight would replay 7/8 of the run a second time)
    marks, lens, acc, nxt, native = [], [], 0, 0, True
    while nxt < seq_num:
        marks.append((acc, nxt, random.getstate()))
        got = replay_sampler(seq_num, genome_seqs, genome_lens, r, seed, total_len, distr, profile, min_read_len,
                             first_read_i=nxt, stop_after=_REPLAY_BLOCK, _cache=cache)
        if got is None:
            native = False
            break
        lens.extend(got[0].tolist())
        acc += len(got[0])
        nxt = got[1]
    if not native:
        random.setstate(state)
        lens = sampling(seq_num, genome_seqs, genome_lens, r, seed, total_len, distr, profile, min_read_len, materialise=(0, 0))
    lo, hi = shard_of(lens)
    random.setstate(state)
    start_i, start_acc = 0, 0
    if native and lo > 0:
        acc0, nxt0, st0 = max((m for m in marks if m[0] <= lo), key=lambda m: m[0])
        random.setstate(st0)
        start_i = nxt0
        if lo > acc0:
            skipped = replay_sampler(seq_num, genome_seqs, genome_lens, r, seed, total_len, distr, profile, min_read_len,
                                     first_read_i=nxt0, stop_after=lo - acc0, want_lengths=False, _cache=cache)
            start_i = skipped[1]
        start_acc = lo

    def own_reads():
        it = sampling_iter(seq_num, genome_seqs, genome_lens, r, seed, total_len, distr, profile, min_read_len,
                           materialise=(lo, hi), first_read_i=start_i, n_accepted=start_acc)
        for i, rd in enumerate(it, start=start_acc):
            if i >= hi:
                return
            if i >= lo:
                yield rd, str(uuid4())
    return own_reads(), lens


def sample_reads_from_reference(genome_seqs, genome_lens, n, r, c, config, fasta, seed, save=False, distr="expon",
                                profile="dna-r10-min", min_read_len=30, lazy=False):
    """utils.py:495-582 (same argument checks and messages).  lazy: the reads come from a generator that samples while it is
    being consumed, and the chunk count returned beside it is the estimate n * r / max_dna_len instead of the exact sum."""
    _check_sampling_args(n, r, c)
    total_len = sum(len(seq) for seq in genome_seqs)
    avg_genome_len = total_len / len(genome_seqs)
    seq_num = n if n != -1 else round(c * total_len / r)
    if r > avg_genome_len and profile.startswith("dna"):
        logger.warning(f"Average reference sequence length ({avg_genome_len:.2f}) is smaller than the desired average "
                       f"read length ({r}). Reads longer than their reference sequence are skipped; consider a smaller -r.")
    if lazy and not save:
        return (yield_reads(sampling_iter(seq_num, genome_seqs, genome_lens, r, seed, total_len, distr, profile, min_read_len),
                            count=seq_num),
                round(seq_num * r / config["max_dna_len"]))
    read_list = sampling(seq_num, genome_seqs, genome_lens, r, seed, total_len, distr, profile, min_read_len)
    total_l = sum(round(len(read) / config["max_dna_len"]) for read in read_list)
    reads_fasta = export_fasta(read_list, fasta) if save else yield_reads(read_list)
    return reads_fasta, total_l


_NON_ACGT = bytes(ord("N") if chr(b) not in "ACGT" else b for b in range(256))


def process_genome(genome_seq: str):
    """upper-case, everything but ACGT -> N (utils.py:594-597)."""
    g = genome_seq.upper().encode("latin-1").translate(_NON_ACGT).decode("latin-1")
    return g, len(g)


def preprocess_genome(fasta: str):
    fast = _read_fasta_native(fasta, map_acgtn=True)          # parse + process_genome in one native pass
    results = [(seq, len(seq)) for seq, _ in fast] if fast is not None else [process_genome(seq) for seq, _ in read_fasta(fasta)]
    if not results:
        raise ValueError(f"{fasta}: no sequences found")
    seqs, lens = zip(*results)
    return list(seqs), list(lens)


def get_reads(fasta, read_input, n, r, c, config, distr, seed, profile, min_read_len, save=False, lazy=False):
    """-> (iterable of (sequence, read_id), approximate chunk count)  (utils.py:641-671).  lazy: reference mode samples while
    the iterable is consumed."""
    logger.info(f"{'Read' if read_input else 'Reference'} mode.")
    is_rna = profile.startswith("rna")
    if read_input:
        if n <= 0:
            total = sum(len(s) for s, _ in read_fasta(fasta, is_rna))
            return read_fasta(fasta, is_rna), total
        all_reads = list(read_fasta(fasta, is_rna))
        rng = random.Random(seed)
        sampled = [rng.choice(all_reads) for _ in range(n)]
        effective = sum(round(len(seq) / config["max_dna_len"]) for seq, _ in sampled)
        return CountedReads(((seq, str(uuid4())) for seq, _ in sampled), len(sampled)), effective
    genome_seqs, genome_lens = preprocess_genome(fasta)
    reads_fasta, total_l = sample_reads_from_reference(genome_seqs, genome_lens, n, r, c, config, str(fasta), seed, save,
                                                       distr, profile, min_read_len, lazy=lazy)
    return (read_fasta(reads_fasta, is_rna), total_l) if save else (reads_fasta, total_l)


# --------------------------------------------------------------------------------------- seeds / logging
def write_synthetic_reference(path, contig_lens, seed=1234, n_runs=True):
    """The synthetic reference of BASELINE.json configs[4] at any size (tests, bench.py): i.i.d. ACGT from numpy
    default_rng(seed), one contig per entry of contig_lens, 80 columns, optionally a short run of N per contig (the
    N -> random base path of the sampler, reference utils.py:402-403).  Returns the total length."""
    _ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        for i, L in enumerate(contig_lens):
            seq = _ACGT[rng.integers(0, 4, L)]
            if n_runs:
                seq[1000:1010] = ord("N")
            f.write(f">contig{i}\n".encode())
            rows = -(-L // 80)
            body = np.full((rows, 81), ord("\n"), np.uint8)
            padded = np.zeros(rows * 80, np.uint8)
            padded[:L] = seq
            body[:, :80] = padded.reshape(rows, 80)
            lines = body.reshape(-1)
            if L % 80:                                              # drop the padding of the last line, keep its newline
                lines = np.concatenate([lines[: (L // 80) * 81 + L % 80], np.array([ord("\n")], np.uint8)])
            f.write(lines.tobytes())
    return sum(contig_lens)


def set_seeds(seed: int) -> int:
    """utils.py:722-741: seed 0 draws a fresh one; seeds python, numpy and torch.  Returns the seed used
    (it also keys the device-side Philox generator)."""
    import torch
    if not seed:
        seed = int.from_bytes(os.urandom(4), byteorder="big", signed=False)
        logger.info(f"No seed provided. Generated random seed: {seed}")
    logger.info(f"Setting all random seeds to {seed}")
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    np.random.seed(seed)
    return seed


def setup_logging(verbosity: str):
    """Console handler on stderr, `{name} {levelname} {asctime}: {message}` (utils.py:687-719)."""
    levels = {"debug": logging.DEBUG, "info": logging.INFO, "warning": logging.WARNING, "error": logging.ERROR}
    logging.captureWarnings(True)
    root = logging.getLogger()
    root.setLevel(logging.DEBUG)
    handler = logging.StreamHandler(sys.stderr)
    handler.setLevel(levels[verbosity.lower()])
    handler.setFormatter(logging.Formatter("{name} {levelname} {asctime}: {message}", style="{", datefmt="%H:%M:%S"))
    root.addHandler(handler)
    logging.getLogger("py.warnings").addHandler(handler)
    for noisy in ("fsspec", "h5py", "torch", "urllib3"):
        logging.getLogger(noisy).setLevel(logging.WARNING)
