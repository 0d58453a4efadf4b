#!/usr/bin/env python
"""ctypes fuzz driver for a (sanitizer) build of csrc/s2s_host.cpp: tools/sanitize_host.sh runs it under ASan+UBSan and TSan.

    fuzz_host.py LIB [rounds] [tag]

Every buffer handed to the library is malloc()ed at its exact size (so an off-by-one lands in a red zone) and every result is
checked against an independent implementation: s2s_blow5_pack methods 0 / 1 / 3 (zlib.decompress; 2 = zstd when libzstd loads)
and s2s_compress_rows, s2s_fasta_count / s2s_fasta_clean / s2s_fastq_clean against the line loop of utils.read_fasta,
s2s_sampler_replay_law against utils.sampling_iter for the three length laws, s2s_length_law against scipy, s2s_copy_ranges /
s2s_blow5_scan (the shard merge's helpers) against slices of the source files."""
import ctypes as C
import os
import random
import struct
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

lib_path, rounds = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
tag = sys.argv[3] if len(sys.argv) > 3 else "plain"
L = C.CDLL(lib_path)
libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.free.argtypes = [C.c_void_p]
vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
L.s2s_blow5_pack_bound.restype = i64
L.s2s_blow5_pack_bound.argtypes = [i64, i32]
L.s2s_blow5_pack.restype = i64
L.s2s_blow5_pack.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i64]
L.s2s_compress_rows.restype = i64
L.s2s_compress_rows.argtypes = [vp, vp, i32, i32, i32, i32, vp, i64, vp]
L.s2s_fasta_count.restype = i64
L.s2s_fasta_count.argtypes = [vp, i64]
for f in (L.s2s_fasta_clean, L.s2s_fastq_clean):
    f.restype = i64
    f.argtypes = [vp, i64, i32, vp, vp, vp, i64]
L.s2s_sampler_replay_law.restype = i64
L.s2s_sampler_replay_law.argtypes = [vp, vp, i32, vp, vp, i64, i64, i64, C.c_uint64, i64, i32, i32, i32, i64, i32, vp, vp]
L.s2s_length_law.restype = i64
L.s2s_length_law.argtypes = [i32, C.c_uint32, C.c_double, i64]


class Buf:
    """malloc(n) exactly (n = 0: one byte nobody may touch is not representable -- malloc(1), never dereferenced by a correct callee)."""
    def __init__(self, data=None, n=None):
        if data is not None:
            data = bytes(data)
            n = len(data)
        self.n = n
        self.p = libc.malloc(max(n, 1))
        assert self.p
        if data:
            C.memmove(self.p, data, n)

    def bytes(self, n=None):
        return C.string_at(self.p, self.n if n is None else n)

    def free(self):
        libc.free(self.p)


def arr(a, dtype):
    return Buf(np.ascontiguousarray(a, dtype=dtype).tobytes())


rng = np.random.default_rng(int.from_bytes(tag.encode(), "little") % (2 ** 32))
have_zstd = None


def records(n):
    """n record bodies as (prefix, signal, suffix) of assorted sizes: empty ones, tiny ones, > 128 KiB (several deflate pieces),
    skewed byte histograms (one symbol; 40-deep Huffman trees) and signal-like noise on a level."""
    out = []
    for _ in range(n):
        kind = int(rng.integers(0, 7))
        size = [0, 1, int(rng.integers(2, 64)), int(rng.integers(64, 5000)), int(rng.integers(5000, 60000)),
                int(rng.integers(131072, 300000)), int(rng.integers(100, 3000))][kind]
        if kind == 6:
            sig = np.full(size, int(rng.integers(0, 256)), np.uint8)
        elif kind % 2:
            w = np.cumsum(rng.integers(1, 4, 256).astype(np.float64)) ** 6          # a very skewed histogram
            sig = rng.choice(256, size=size, p=w / w.sum()).astype(np.uint8)
        else:
            sig = (400 + 12 * rng.standard_normal(size // 2 + 1)).astype("<i2").view(np.uint8)[:size]
        out.append((bytes(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8)), sig.tobytes(),
                    bytes(rng.integers(0, 256, int(rng.integers(0, 12)), dtype=np.uint8))))
    return out


def offs(parts):
    o = [0]
    for p in parts:
        o.append(o[-1] + len(p))
    return o


def check_pack(threads):
    global have_zstd
    recs = records(int(rng.integers(1, 14)))
    pre, sig, suf = [r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs]
    bp, bs, bf = Buf(b"".join(pre)), Buf(b"".join(sig)), Buf(b"".join(suf))
    op, os_, of = arr(offs(pre), np.int64), arr(offs(sig), np.int64), arr(offs(suf), np.int64)
    total = sum(len(a) + len(b) + len(c) for a, b, c in recs)
    cap = L.s2s_blow5_pack_bound(total, len(recs))
    for method in (0, 1, 3, 2):
        if method == 2 and have_zstd is False:
            continue
        out = Buf(n=cap)
        got = L.s2s_blow5_pack(bp.p, op.p, bf.p, of.p, bs.p, os_.p, len(recs), method, 1, threads, out.p, cap)
        if method == 2 and got < 0:
            have_zstd = False
            out.free()
            continue
        assert 0 <= got <= cap, (method, got, cap)
        raw = out.bytes(got)
        pos = 0
        for a, b, c in recs:
            (sz,) = struct.unpack_from("<Q", raw, pos)
            blob = raw[pos + 8: pos + 8 + sz]
            pos += 8 + sz
            body = a + b + c
            if method == 0:
                assert blob == body
            elif method in (1, 3):
                assert zlib.decompress(blob) == body, (method, len(body))
            else:
                have_zstd = True
                assert len(blob) > 0 or not body
        assert pos == got
        out.free()
    # the same bodies as plain rows
    rows = [a + b + c for a, b, c in recs]
    bi, oi = Buf(b"".join(rows)), arr(offs(rows), np.int64)
    for method in (1, 2):
        if method == 2 and have_zstd is False:
            continue
        out, oo = Buf(n=cap), Buf(n=8 * (len(rows) + 1))
        got = L.s2s_compress_rows(bi.p, oi.p, len(rows), method, 1, threads, out.p, cap, oo.p)
        if method == 2 and got < 0:
            have_zstd = False
        else:
            assert 0 <= got <= cap
            bounds = np.frombuffer(oo.bytes(), np.int64)
            assert bounds[0] == 0 and bounds[-1] == got and (np.diff(bounds) >= 0).all()
            if method == 1:
                raw = out.bytes(got)
                for r, row in enumerate(rows):
                    assert zlib.decompress(raw[bounds[r]: bounds[r + 1]]) == row
        out.free(); oo.free()
    for b in (bp, bs, bf, op, os_, of, bi, oi):
        b.free()


def line_loop(text, fastq):
    """utils.read_fasta's loop on a latin-1 text (universal newlines), -> records or "raised"."""
    import io
    from seq2squiggle_amd import utils as U
    path = "/tmp/s2s_fuzz_%d.%s" % (os.getpid(), "fq" if fastq else "fa")
    with open(path, "w", encoding="latin-1", newline="") as f:
        f.write(text)
    saved = U._read_fasta_native
    U._read_fasta_native = lambda p, map_acgtn=False, limit=0: None
    try:
        return list(U.read_fasta(path))
    except Exception:
        return "raised"
    finally:
        U._read_fasta_native = saved
        os.unlink(path)


def native_parse(data: bytes, fastq, map_acgtn=0):
    d = Buf(data)
    n = len(data)
    if fastq:
        n_rec = L.s2s_fastq_clean(d.p, n, 0, None, None, None, 0)
        clean = L.s2s_fastq_clean
    else:
        n_rec = L.s2s_fasta_count(d.p, n)
        clean = L.s2s_fasta_clean
    if n_rec < 0:
        d.free()
        return None
    out, so, ns = Buf(n=n), Buf(n=8 * (n_rec + 1)), Buf(n=16 * n_rec)
    got = clean(d.p, n, map_acgtn, out.p, so.p, ns.p, n_rec)
    assert got == n_rec, (got, n_rec)
    o = np.frombuffer(so.bytes(), np.int64)
    s = np.frombuffer(ns.bytes(), np.int64) if n_rec else np.zeros(0, np.int64)
    ob = out.bytes()
    recs = [(ob[o[r]: o[r + 1]].decode("latin-1"), data[s[2 * r]: s[2 * r + 1]].decode("latin-1")) for r in range(n_rec)]
    assert n_rec == 0 or (0 <= o).all() and o[-1] <= n
    for b in (d, out, so, ns):
        b.free()
    return recs


ALPHA = list("ACGTNacgtRY>@+ \t\r\n\n\n;I!") + ["\x1c", "\x1f", "\x85", "\xa0", "\xe9"]     # (the last five: the parser must hand the file back)


def check_parsers():
    for fastq in (False, True):
        kind = int(rng.integers(0, 3))
        if kind == 0:                                           # byte soup from the interesting alphabet
            text = "".join(rng.choice(ALPHA, int(rng.integers(0, 200))))
        else:                                                   # well-formed records, sometimes damaged
            text = ""
            for i in range(int(rng.integers(0, 6))):
                sq = "".join(rng.choice(list("ACGTNacgt"), int(rng.integers(0, 120))))
                nl = "\r\n" if rng.integers(0, 2) else "\n"
                if fastq:
                    text += f"@r{i} x{nl}{sq}{nl}+{nl}{'I' * len(sq)}{nl}"
                else:
                    w = int(rng.integers(1, 70))
                    text += f">r{i} x{nl}" + "".join(sq[j:j + w] + nl for j in range(0, len(sq), w))
            if kind == 2 and text:
                cut = int(rng.integers(0, len(text)))
                text = text[:cut] + str(rng.choice(ALPHA)) + text[cut + int(rng.integers(0, 3)):]
        data = text.encode("latin-1")
        got = native_parse(data, fastq)
        want = line_loop(text, fastq)
        first = next((ln for ln in text.replace("\r\n", "\n").replace("\r", "\n").split("\n") if ln), "")
        if got is None:
            continue                                            # handed back to the line loop: nothing to compare
        assert not any(ord(ch) >= 0x80 or 0x1c <= ord(ch) <= 0x1f for ch in text), ("parsed a file with bytes the line loop reads differently", text)
        if not fastq and first.startswith("@"):
            continue
        assert got == want, (fastq, text, got, want)
        if not fastq:                                           # process_genome's mapping
            mapped = native_parse(data, fastq, 1)
            assert [m[0] for m in mapped] == ["".join(c if c in "ACGT" else "N" for c in s.upper()) for s, _ in got]


def check_replay(distr_i):
    from seq2squiggle_amd import utils as U
    distr = ("expon", "gamma", "beta")[distr_i]
    n_ctg = int(rng.integers(1, 4))
    contigs = []
    for _ in range(n_ctg):
        Lc = int(rng.integers(200, 9000))
        s = rng.choice(list("ACGT"), Lc)
        if rng.integers(0, 2):
            for lo in rng.integers(0, max(1, Lc - 50), 4):
                s[lo: lo + int(rng.integers(1, 50))] = "N"
        contigs.append("".join(s))
    lens = [len(c) for c in contigs]
    total, seed, n = sum(lens), int(rng.integers(0, 10 ** 6)), int(rng.integers(1, 60))
    r, profile = int(rng.choice([300, 1200, 5000])), str(rng.choice(["dna-r10-prom", "rna-004-prom"]))
    random.seed(seed)
    want = U.sampling(n, contigs, lens, r, seed, total, distr, profile, 30, materialise=(0, 0))
    end_state = random.getstate()
    random.seed(seed)
    st = Buf(np.array(random.getstate()[1], dtype=np.uint32).tobytes())
    ends = arr(np.cumsum(lens), np.int64)
    npos = [np.flatnonzero(np.frombuffer(c.encode(), np.uint8) == ord("N")).astype(np.int64) for c in contigs]
    nbufs = [arr(p, np.int64) if p.size else None for p in npos]
    ptrs = Buf(np.array([b.p if b else 0 for b in nbufs], dtype=np.uint64).tobytes())
    cnts = arr([p.size for p in npos], np.int64)
    out, nxt = Buf(n=4 * len(want)), Buf(n=8)                   # exactly as many slots as reads will be accepted
    got = L.s2s_sampler_replay_law(st.p, ends.p, n_ctg, ptrs.p, cnts.p, n, 0, r, seed, total, int(profile.startswith("dna")), 30, 20, -1,
                                   distr_i, out.p, nxt.p)
    assert got == len(want), (got, len(want))
    assert np.frombuffer(out.bytes(), np.int32).tolist() == want
    assert np.frombuffer(nxt.bytes(), np.int64)[0] == n
    assert tuple(int(x) for x in np.frombuffer(st.bytes(), np.uint32)) == end_state[1]
    for b in [st, ends, ptrs, cnts, out, nxt] + [b for b in nbufs if b]:
        b.free()


def check_laws():
    from seq2squiggle_amd import utils as U
    for law, name in enumerate(("expon", "gamma", "beta")):
        seed = int(rng.integers(0, 2 ** 32))
        assert L.s2s_length_law(law, seed, 5000.0, 48502) == int(U.draw_length(name, 5000, seed, 48502))


L.s2s_copy_ranges.restype = i64
L.s2s_copy_ranges.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32]
L.s2s_blow5_scan.restype = i64
L.s2s_blow5_scan.argtypes = [i32, i64, i64]
L.s2s_blow5_scan_upto.restype = i64
L.s2s_blow5_scan_upto.argtypes = [i32, i64, i64, i64, vp]


def check_merge_helpers(threads):
    """s2s_copy_ranges: random non-overlapping destination ranges from two source files, on `threads` threads; s2s_blow5_scan: a
    chain of [u64 size][body] records, whole and damaged."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        srcs = []
        for i in range(2):
            data = rng.integers(0, 256, int(rng.integers(1, 300000)), dtype=np.uint8).tobytes()
            with open(os.path.join(td, f"s{i}"), "wb") as f:
                f.write(data)
            srcs.append((os.open(os.path.join(td, f"s{i}"), os.O_RDONLY), data))
        n = int(rng.integers(0, 12))
        jobs, at, want = [], 0, bytearray()
        for _ in range(n):
            k = int(rng.integers(0, 2))
            fd, data = srcs[k]
            so = int(rng.integers(0, len(data)))
            ln = int(rng.integers(0, len(data) - so + 1))
            gap = int(rng.integers(0, 9))
            want += bytes(gap) + data[so:so + ln]
            jobs.append((fd, so, at + gap, ln))
            at += gap + ln
        dst = os.open(os.path.join(td, "d"), os.O_RDWR | os.O_CREAT)
        os.ftruncate(dst, at)
        cols = [Buf(np.array([j[i] for j in jobs], dt).tobytes()) for i, dt in ((0, np.int32), (1, np.int64), (2, np.int64), (3, np.int64))]
        dfd = Buf(np.full(n, dst, np.int32).tobytes())
        for engine in (0, 2):                              # descriptors; preallocate + mapped fill (2: on any file system)
            os.ftruncate(dst, 0)
            os.ftruncate(dst, at)
            got = L.s2s_copy_ranges(n, cols[0].p, cols[1].p, dfd.p, cols[2].p, cols[3].p, threads, engine)
            assert got == sum(j[3] for j in jobs), (engine, got)
            assert os.pread(dst, at, 0) == bytes(want), engine
        for b in cols + [dfd]:
            b.free()
        # record chain
        sizes = [int(rng.integers(0, 500)) for _ in range(int(rng.integers(0, 20)))]
        chain = b"".join(struct.pack("<Q", z) + bytes(z) for z in sizes)
        with open(os.path.join(td, "c"), "wb") as f:
            f.write(b"HEAD" + chain + b"5WOLB")
        cfd = os.open(os.path.join(td, "c"), os.O_RDONLY)
        assert L.s2s_blow5_scan(cfd, 4, 4 + len(chain)) == len(sizes)
        if chain:
            assert L.s2s_blow5_scan(cfd, 4, 4 + len(chain) - 1) == -2          # the last record runs past the end
        assert L.s2s_blow5_scan(cfd, 4, 3) < 0 and L.s2s_blow5_scan(-1, 0, 0) < 0
        # the walk over a file "still being written": any limit inside the chain yields the complete records in front of it
        ends = np.cumsum([0] + [8 + z for z in sizes]) + 4
        for _ in range(6):
            limit = int(rng.integers(0, 4 + len(chain) + 6 + 1))
            cap = int(rng.integers(0, len(sizes) + 3))
            out_end = Buf(n=8)
            got = L.s2s_blow5_scan_upto(cfd, 4, limit, cap, out_end.p)
            want = min(cap, int(np.searchsorted(ends, limit, side="right")) - 1) if limit >= 4 else 0
            want = max(want, 0)
            # (the 5-byte end marker behind the chain never counts: it is shorter than a size prefix, or its "size" runs past the limit)
            assert got == want, (got, want, limit, cap, sizes)
            assert np.frombuffer(out_end.bytes(), np.int64)[0] == ends[got]
            out_end.free()
        for fd in [x[0] for x in srcs] + [dst, cfd]:
            os.close(fd)


for it in range(rounds):
    check_pack(threads=[1, 2, 4, 8][it % 4])
    check_merge_helpers(threads=[1, 3, 8][it % 3])
    for _ in range(8):
        check_parsers()
    if it % 3 == 0:
        check_replay(it // 3 % 3)
        check_laws()
print(f"FUZZ_OK {tag}: {rounds} rounds (zstd {'checked' if have_zstd else 'not loadable'})")
