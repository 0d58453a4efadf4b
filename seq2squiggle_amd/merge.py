"""Joining the per-rank files of a sharded run (`predict --gpus N`, parallel.rank_output_path) into the ONE file the reference
leaves (inference.py:65-79; signal_io.py:167-171, 268-282) without re-reading a record in the interpreter.

A shard's payload -- the BLOW5 record section, the lines of a SLOW5 file, the buffers of a POD5 signal table -- is a handful of
byte ranges whose place in the merged file follows from prefix sums, so the merge is: lay out, copy every range
(s2s_copy_ranges in libs2s_hip.so), then write the few KB that are new (BLOW5: the end marker; POD5: batch metadata, the reads
table, the footers -- pod5_io.merge_pod5).

What bounds it is how the file system lets ONE file be filled (profiles/r05/fs_write_probe_shm.txt, the MI355X box's tmpfs): a
buffered writer (pwrite, copy_file_range) holds the inode lock and allocates the pages as it goes -- 6.5 GB/s, and 2-8 such
writers of the same file are SLOWER (3.2-4.1 GB/s), while separate files scale to 42 GB/s, which is why the ranks write their own
files.  But allocating the pages WITHOUT data (posix_fallocate) runs at 18.6 GB/s, and filling pages that already exist through a
shared mapping takes no lock and scales with threads.  Hence the default engine ("map"): on tmpfs reserve a shard's destination
range, then fill it on merge_threads() threads -- 8.0 GB/s of output against 5.7 GB/s for copy_file_range with its one writer on
the same box (profiles/r05/merge_bench_shm.txt); on any other file system (a disk's page cache: fallocate is a block allocation
there, and the one writer does 7-10 GB/s) s2s_copy_ranges takes copy_file_range by itself.  `S2S_MERGE_ENGINE=fd` forces that.
`take_first=True` turns the first shard INTO the output file (its payload is already where it belongs: 1/N fewer bytes move);
`consume=True` deletes every other shard as soon as its bytes are in the output, on a helper thread beside the copy of the
next one (freeing 6 GB of tmpfs pages takes 0.5 s, and the pages go straight back to the copy: the peak is the output + one shard,
not twice the output)."""
import os
import struct
import time
from concurrent.futures import ThreadPoolExecutor
from typing import List, Sequence, Tuple

import numpy as np

BLOW5_EOF = b"5WOLB"


def merge_engine() -> int:
    """0: copy_file_range on descriptors, one writer; 1 (default): on tmpfs preallocate the destination range, then fill it through a
    shared mapping on merge_threads() threads (0 on other file systems or where refused); 2: that on any file system (A/B).
    S2S_MERGE_ENGINE = fd | map | map-anywhere."""
    want = os.environ.get("S2S_MERGE_ENGINE", "map").lower()
    return 0 if want.startswith("f") else 2 if want.endswith("anywhere") else 1


def merge_threads() -> int:
    """Copy threads of the mapped engine: up to 8 of this process's CPU share (the measured knee); S2S_MERGE_THREADS overrides."""
    from .signal_io import cpu_share
    return max(1, int(os.environ.get("S2S_MERGE_THREADS", "0") or 0) or min(8, cpu_share()))


class _Remover:
    """consume=True: shard files are deleted, in order, by one helper thread while the copy of the next shard runs."""

    def __init__(self, enabled: bool):
        self.ex = ThreadPoolExecutor(max_workers=1, thread_name_prefix="s2s-unlink") if enabled else None
        self.pending = []
        self.seconds = 0.0

    def remove(self, path: str) -> None:
        if self.ex is not None:
            self.pending.append(self.ex.submit(self._one, path))

    def _one(self, path):
        t = time.perf_counter()
        os.remove(path)
        self.seconds += time.perf_counter() - t

    def finish(self) -> float:
        for f in self.pending:
            f.result()
        if self.ex is not None:
            self.ex.shutdown()
        return self.seconds


def copy_ranges(jobs: Sequence[Tuple[int, int, int, int, int]], threads: int = None, engine: int = None) -> int:
    """jobs: (src_fd, src_offset, dst_fd, dst_offset, length).  Ranges of one file must not overlap.  -> bytes copied."""
    jobs = [j for j in jobs if j[4] > 0]
    if not jobs:
        return 0
    threads = threads or merge_threads()
    engine = merge_engine() if engine is None else engine
    a = np.array(jobs, dtype=np.int64)
    try:
        from ._lib import lib
        fn = getattr(lib(), "s2s_copy_ranges", None)
    except (RuntimeError, OSError):
        fn = None                                  # `merge-shards` on a host without the built library: copy_file_range from Python
    if fn is not None:
        src, so, dst, do, ln = (np.ascontiguousarray(a[:, i], dtype=t) for i, t in
                                enumerate((np.int32, np.int64, np.int32, np.int64, np.int64)))
        got = fn(len(jobs), src.ctypes.data, so.ctypes.data, dst.ctypes.data, do.ctypes.data, ln.ctypes.data, int(threads), int(engine))
        if got < 0:
            raise OSError(f"s2s_copy_ranges failed ({got}: a bad argument, or -errno of the failing copy_file_range / pread / pwrite)")
        return int(got)
    for s_, so, d, do, ln in jobs:                 # one writer (see the module text)
        while ln > 0:
            try:
                r = os.copy_file_range(s_, d, min(ln, 1 << 30), so, do)
            except OSError:
                r = os.pwrite(d, os.pread(s_, min(ln, 8 << 20), so), do)
            if r <= 0:
                raise OSError("short copy while merging shard files")
            so, do, ln = so + r, do + r, ln - r
    return int(a[:, 4].sum())


class _Files:
    """Open descriptors of the shard files (+ the output), closed together."""

    def __init__(self):
        self.fds: List[int] = []

    def open(self, path: str, flags: int = os.O_RDONLY, mode: int = 0o644) -> int:
        fd = os.open(path, flags, mode)
        self.fds.append(fd)
        return fd

    def __enter__(self):
        return self

    def __exit__(self, *a):
        for fd in self.fds:
            try:
                os.close(fd)
            except OSError:
                pass


def _blow5_layout(fd: int, path: str):
    """-> (64-byte file header, header text bytes, begin, end of the record section) of one BLOW5 shard."""
    size = os.fstat(fd).st_size
    head = os.pread(fd, 68, 0)
    if len(head) < 68 or head[:6] != b"BLOW5\x01":
        raise ValueError(f"{path}: not a BLOW5 file")
    hlen = struct.unpack_from("<I", head, 64)[0]
    text = os.pread(fd, hlen, 68)
    end = size - len(BLOW5_EOF)
    if end < 68 + hlen or os.pread(fd, len(BLOW5_EOF), end) != BLOW5_EOF:
        raise ValueError(f"{path}: no end-of-file marker (truncated shard?)")
    return head[:64], text, 68 + hlen, end


def _scan_blow5(fd: int, begin: int, end: int, path: str) -> int:
    """Number of records between begin and end (size prefixes only; s2s_blow5_scan, or the same walk with os.pread)."""
    try:
        from ._lib import lib
        fn = getattr(lib(), "s2s_blow5_scan", None)
    except (RuntimeError, OSError):
        fn = None
    if fn is not None:
        n = int(fn(fd, begin, end))
    else:
        n, pos = 0, begin
        while pos < end:
            raw = os.pread(fd, 8, pos)
            if len(raw) < 8 or struct.unpack("<Q", raw)[0] > end - pos - 8:
                n = -2
                break
            pos += 8 + struct.unpack("<Q", raw)[0]
            n += 1
    if n < 0:
        raise ValueError(f"{path}: truncated record (the size prefixes do not end at the end-of-file marker)")
    return n


def merge_blow5(paths: Sequence[str], out: str, threads: int = None, take_first: bool = False, consume: bool = False) -> Tuple[int, dict]:
    """BLOW5 shards with identical headers -> `out`: header of the first, every shard's record section at its prefix-sum offset,
    one end-of-file marker.  Nothing is decompressed or parsed beyond the u64 size prefixes (counted for the return value and
    as the truncation check; all shards are checked before the first byte moves).  -> (records, {"bytes_copied", "seconds", ...})."""
    t0 = time.perf_counter()
    threads = threads or merge_threads()
    same = lambda text: [l for l in text.decode().splitlines() if not l.startswith("@exp_start_time")]   # the wall clock may differ
    remover = _Remover(consume)
    with _Files() as files:
        fds = [files.open(p_) for p_ in paths]
        lay = [_blow5_layout(fd, p_) for fd, p_ in zip(fds, paths)]
        for p_, (head, text, _, _) in zip(paths[1:], lay[1:]):
            if head != lay[0][0] or same(text) != same(lay[0][1]):
                raise ValueError(f"{p_}: header differs from {paths[0]} (another profile, compression or run?)")
        with ThreadPoolExecutor(max_workers=min(8, len(paths))) as ex:       # (reads 8 bytes per record: not the destination's business)
            counts = list(ex.map(lambda i: _scan_blow5(fds[i], lay[i][2], lay[i][3], paths[i]), range(len(paths))))
        t_scan = time.perf_counter() - t0
        sizes = [end - begin for _, _, begin, end in lay]
        at = np.concatenate([[lay[0][2]], lay[0][2] + np.cumsum(sizes)]).astype(np.int64)
        if take_first:
            os.replace(paths[0], out)
            dst = files.open(out, os.O_RDWR)
            first = 1
        else:
            dst = files.open(out, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
            os.pwrite(dst, lay[0][0] + struct.pack("<I", len(lay[0][1])) + lay[0][1], 0)
            first = 0
        os.ftruncate(dst, int(at[-1]) + len(BLOW5_EOF))
        copied = 0
        for i in range(first, len(paths)):
            copied += copy_ranges([(fds[i], lay[i][2], dst, int(at[i]), sizes[i])], threads)
            os.close(fds[i])
            files.fds.remove(fds[i])
            remover.remove(paths[i])
        os.pwrite(dst, BLOW5_EOF, int(at[-1]))
    removing = remover.finish()
    return int(sum(counts)), {"bytes_copied": copied, "seconds": time.perf_counter() - t0, "scan_seconds": t_scan,
                              "remove_seconds": removing, "threads": threads, "engine": "map" if merge_engine() else "fd"}


def _slow5_header_end(fd: int, path: str) -> int:
    """Byte offset of the first record line (the first line that starts with neither '#' nor '@')."""
    size, pos, carry = os.fstat(fd).st_size, 0, b""
    while pos < size:
        blk = os.pread(fd, 1 << 16, pos)
        data = carry + blk
        start = pos - len(carry)
        o = 0
        while True:
            if o >= len(data):
                break
            if data[o:o + 1] not in (b"#", b"@"):
                return start + o
            nl = data.find(b"\n", o)
            if nl < 0:
                break
            o = nl + 1
        carry, pos = data[o:], pos + len(blk)
    return size                                   # a header without records


def merge_slow5(paths: Sequence[str], out: str, threads: int = None) -> Tuple[int, dict]:
    """SLOW5 (ASCII) shards: header lines of the first, the record lines of all -- raw ranges; the records are counted as
    line ends while the ranges are in flight."""
    t0 = time.perf_counter()
    with _Files() as files:
        fds = [files.open(p_) for p_ in paths]
        begins = [_slow5_header_end(fd, p_) for fd, p_ in zip(fds, paths)]
        ends = [os.fstat(fd).st_size for fd in fds]
        for p_, fd, b, e in zip(paths, fds, begins, ends):
            if e > b and os.pread(fd, 1, e - 1) != b"\n":
                raise ValueError(f"{p_}: the last record has no line end (truncated shard?)")
        header = os.pread(fds[0], begins[0], 0)
        at = np.concatenate([[len(header)], len(header) + np.cumsum([e - b for b, e in zip(begins, ends)])]).astype(np.int64)
        dst = files.open(out, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
        os.pwrite(dst, header, 0)
        os.ftruncate(dst, int(at[-1]))

        def lines(i):
            n, pos = 0, begins[i]
            while pos < ends[i]:
                blk = os.pread(fds[i], min(16 << 20, ends[i] - pos), pos)
                n += blk.count(b"\n")
                pos += len(blk)
            return n
        with ThreadPoolExecutor(max_workers=max(1, min(threads or merge_threads(), len(paths)))) as ex:
            counting = ex.map(lines, range(len(paths)))
            copied = copy_ranges([(fds[i], begins[i], dst, int(at[i]), ends[i] - begins[i]) for i in range(len(paths))], threads or merge_threads())
            n = sum(counting)
    return int(n), {"bytes_copied": copied, "seconds": time.perf_counter() - t0}
