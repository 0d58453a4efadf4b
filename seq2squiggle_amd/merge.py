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
import logging
import os
import struct
import time
from concurrent.futures import ThreadPoolExecutor
from typing import List, Sequence, Tuple

import numpy as np

logger = logging.getLogger("seq2squiggle")

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


# ------------------------------------------------------------------------------------------------------------- the live join
class LiveJoin:
    """The join WHILE the ranks run (`predict --gpus N --join live`): the parent tails the rank files -- whose writers know nothing
    of it -- and copies every complete unit (BLOW5: a record; POD5: a full 100-row signal batch, whose Arrow message is position
    independent and moves verbatim) into the output as it appears, so that the ONE file is all but finished when the ranks are:
    the command costs max(compute, output / fill rate) instead of their sum.

    The price is the ORDER of the payload: it cannot be the rank order (where rank 1's bytes go is not known before rank 0 has
    finished), so it is a fixed round robin -- QUANTUM units of rank 0, of rank 1, ..., again -- that depends on the ranks'
    record sequences only, never on timing: a rank that has fewer than a quantum ready is waited for unless it is done (then its
    rest goes).  Same records, same ids, numbers and samples as the join after the fact; BLOW5 record order and POD5 signal-row
    placement differ (the POD5 reads table stays in read order).  That is why `--join after` (rank order, byte-equal to a
    single-process file's layout) stays the default.

    What can be read from a file that is being written: a writer appends with plain write() calls, bytes below the size a reader
    sees are final, and both formats frame their units with a length in front -- a unit is taken only when the file is long
    enough to hold all of it (s2s_blow5_scan_upto; the Arrow message header).  BLOW5Writer drops and rewrites its 5-byte end
    marker around every batch: 5 bytes never parse as the 8-byte size prefix of a record."""

    QUANTUM = {"blow5": 256, "pod5": 4}          # (about 30-40 MB of payload per turn for 5-10 kb reads)

    def __init__(self, paths: Sequence[str], out: str, threads: int = None, punch: bool = True):
        self.paths, self.out, self.n = list(paths), out, len(paths)
        self.kind = "pod5" if out.endswith(".pod5") else "blow5"
        if not out.endswith((".pod5", ".blow5")):
            raise ValueError("the live join handles .blow5 and .pod5 outputs (SLOW5 text: --join after)")
        self.threads = threads or merge_threads()
        self.q = self.QUANTUM[self.kind]
        self.fds = [None] * self.n
        self.pos = [None] * self.n               # next byte of shard r not yet taken (None: its header is not there yet)
        self.ended = [False] * self.n            # POD5: the shard's full batches are all seen (end-of-stream or its partial batch reached)
        self.exhausted = [False] * self.n
        self.headers = [None] * self.n
        self.turn = 0
        self.out_fd, self.out_pos = None, 0
        self.units, self.bytes_live, self.copy_seconds = 0, 0, 0.0
        self.blocks, self.placed = [], [[] for _ in range(self.n)]      # POD5
        self.head_len = None
        self._punch = _Puncher() if punch else None
        self._unpunched = []                     # copied ranges whose pages wait for the last shard's header check
        self.punched = 0                         # bytes of rank files already given back: from then on the partial output is the only copy
        self._pa = None
        if self.kind == "pod5":                  # everything the finish needs is loaded BEFORE the first byte is consumed
            from . import pod5_io as P
            self._pa = P._pa()

    @staticmethod
    def _blow5_header_lines(text: bytes):
        return [l for l in text.decode().splitlines() if not l.startswith("@exp_start_time")]

    def _check_against_first(self, r: int) -> None:
        """A shard that does not belong to shard 0's run (another profile, compression, schema) is refused when its header is first
        seen -- before anything of it is copied or punched -- not when the ranks are done and their files are hollow."""
        if r == 0 or self.headers[0] is None:
            return
        if self.kind == "blow5":
            head, text = self.headers[r]
            if head != self.headers[0][0] or self._blow5_header_lines(text) != self._blow5_header_lines(self.headers[0][1]):
                raise ValueError(f"{self.paths[r]}: header differs from {self.paths[0]} (another profile, compression or run?)")
        else:
            pa = self._pa
            schemas = [pa.ipc.read_schema(pa.py_buffer(self.headers[i][32:])).remove_metadata() for i in (0, r)]
            if schemas[0] != schemas[1]:
                raise ValueError(f"{self.paths[r]}: signal table schema differs from {self.paths[0]} (VBZ and uncompressed shards do not mix)")

    # ---- shard headers
    def _open(self, r: int) -> bool:
        if self.fds[r] is None:
            try:
                self.fds[r] = os.open(self.paths[r], os.O_RDWR)
            except FileNotFoundError:
                return False
        if self.pos[r] is not None:
            return True
        fd = self.fds[r]
        size = os.fstat(fd).st_size
        if self.kind == "blow5":
            if size < 68:
                return False
            head = os.pread(fd, 68, 0)
            if head[:6] != b"BLOW5\x01":
                raise ValueError(f"{self.paths[r]}: not a BLOW5 file")
            hlen = struct.unpack_from("<I", head, 64)[0]
            if size < 68 + hlen:
                return False
            self.headers[r] = (head[:64], os.pread(fd, hlen, 68))
            self.pos[r] = 68 + hlen
        else:
            from . import pod5_io as P
            if size < 24 + 16:
                return False
            head = os.pread(fd, 40, 0)
            if head[:8] != P.SIGNATURE or head[24:30] != b"ARROW1":
                raise ValueError(f"{self.paths[r]}: not a POD5 file of this writer")
            cont, mlen = struct.unpack_from("<Ii", head, 32)
            if cont != 0xFFFFFFFF or mlen <= 0:
                raise ValueError(f"{self.paths[r]}: no schema message where the signal table starts")
            if size < 40 + mlen:
                return False
            self.headers[r] = os.pread(fd, 40 + mlen, 0)               # signature, marker, Arrow magic, schema message (no body)
            self.pos[r] = 40 + mlen
        try:
            self._check_against_first(r)
        except BaseException:
            self.pos[r] = None
            raise
        return True

    def _start_output(self) -> bool:
        if self.out_fd is not None:
            return True
        if not self._open(0):
            return False
        self.out_fd = os.open(self.out, os.O_RDWR | os.O_CREAT | os.O_EXCL, 0o644)
        if self.kind == "blow5":
            head, text = self.headers[0]
            first = head + struct.pack("<I", len(text)) + text
        else:
            first = self.headers[0]
            self.head_len = len(first) - 24
        os.pwrite(self.out_fd, first, 0)
        self.out_pos = len(first)
        return True

    # ---- one turn of the round robin: True = the turn is over (units taken, or nothing will ever come), False = wait
    def _take(self, r: int, done: bool) -> bool:
        if self.exhausted[r]:
            return True
        if not self._open(r):
            if done:                                # (a rank always leaves a file, even without reads: inference_run sees to it)
                raise FileNotFoundError(f"{self.paths[r]}: the rank is done but its file is missing or has no header")
            return False
        fd = self.fds[r]
        size = os.fstat(fd).st_size
        begin = self.pos[r]
        if self.kind == "blow5":
            from ._lib import lib
            import ctypes as C
            end = C.c_int64(0)
            got = int(lib().s2s_blow5_scan_upto(fd, begin, size, self.q, C.byref(end)))
            if got < 0:
                raise OSError(f"s2s_blow5_scan_upto failed ({got})")
            end = int(end.value)
            last = done and got < self.q
            if got < self.q and not done:
                return False
            if last and end != size - len(BLOW5_EOF):
                raise ValueError(f"{self.paths[r]}: the records do not end at the end-of-file marker (truncated shard?)")
        else:
            got, end, blocks = 0, begin, []
            from . import pod5_io as P
            while got < self.q and not self.ended[r]:
                if size - end < 8:
                    break
                cont, mlen = struct.unpack("<Ii", os.pread(fd, 8, end))
                if cont != 0xFFFFFFFF:
                    raise ValueError(f"{self.paths[r]}: Arrow message without continuation marker at {end}")
                if mlen == 0:                        # end of stream: the signal table is closed
                    self.ended[r] = True
                    break
                if size - end < 8 + mlen:
                    break
                meta = os.pread(fd, 8 + mlen, end)
                m = P._batch_message(meta, 0)
                body = struct.unpack_from("<q", meta, m["body_len"])[0]
                rows = struct.unpack_from("<q", meta, m["length"])[0]
                if rows < P.SIGNAL_BATCH_ROWS:       # the shard's last, partial batch: its rows are laid out by the finish
                    self.ended[r] = True
                    break
                if size - end < 8 + mlen + body:
                    break
                blocks.append((8 + mlen, body))
                end += 8 + mlen + body
                got += 1
            last = self.ended[r] and got < self.q
            if got < self.q and not self.ended[r]:
                if done:
                    raise ValueError(f"{self.paths[r]}: the signal table has no end (truncated shard?)")
                return False
        if got:
            t = time.perf_counter()
            copy_ranges([(fd, begin, self.out_fd, self.out_pos, end - begin)], self.threads)
            self.copy_seconds += time.perf_counter() - t
            if self.kind == "pod5":
                at = self.out_pos
                for meta_len, body in blocks:
                    self.placed[r].append(len(self.blocks))
                    self.blocks.append((at - 24, meta_len, 0, body))       # (Arrow block offsets count from the embedded file's start)
                    at += meta_len + body
            if self._punch is not None:
                # nothing is given back before EVERY shard's header has been seen and held against shard 0's: until then the rank
                # files stay whole, and a shard that does not belong here fails the join with nothing lost
                self._unpunched.append((fd, begin, end))
                if all(p_ is not None for p_ in self.pos):
                    for fd_, lo, hi in self._unpunched:
                        self._punch.punch(fd_, lo, hi)
                        self.punched += hi - lo
                    self._unpunched = []
            self.out_pos += end - begin
            self.pos[r] = end
            self.units += got
            self.bytes_live += end - begin
        if last:
            self.exhausted[r] = True
        return True

    def step(self, done: Sequence[bool]) -> bool:
        """Copies what the round robin allows right now.  done[r]: rank r has closed its file.  -> True if anything moved."""
        if not self._start_output():
            return False
        moved = False
        while not all(self.exhausted):
            before = self.units
            if not self._take(self.turn, bool(done[self.turn])):
                break
            moved = moved or self.units > before
            self.turn = (self.turn + 1) % self.n
        return moved

    def finish(self, consume: bool = True) -> Tuple[int, dict]:
        """Every rank is done: the rest of the round robin, then what only the end can write (BLOW5: the end marker; POD5: the rows
        behind every shard's last full batch, the tables, the footers).  -> (records / reads, stats)."""
        t0 = time.perf_counter()
        live_bytes = self.bytes_live
        self.step([True] * self.n)
        if not all(self.exhausted):
            raise RuntimeError("live join: a rank file did not come to its end")
        if self._punch is not None:
            self._punch.finish()
        if self.kind == "blow5":
            for r in range(1, self.n):                 # (checked when each header was first seen; once more costs nothing)
                self._check_against_first(r)
            os.pwrite(self.out_fd, BLOW5_EOF, self.out_pos)
            os.ftruncate(self.out_fd, self.out_pos + len(BLOW5_EOF))
            n = self.units
            self._close_shards()
            self._close_out()
            if consume:
                for p_ in self.paths:
                    os.remove(p_)
        else:
            from . import pod5_io as P
            self._close_shards()                       # (merge_pod5 opens its own descriptors: ours must not be closed again later)
            blocks = np.array(self.blocks, dtype=P._BLOCK) if self.blocks else np.zeros(0, P._BLOCK)
            n = P.merge_pod5(self.paths, self.out, threads=self.threads, consume=consume,
                             _live={"fd": self.out_fd, "blocks": blocks, "placed": self.placed, "msg_base": self.out_pos - 24,
                                    "head_len": self.head_len})
            self._close_out()
        stats = {"live_bytes": live_bytes, "bytes": os.path.getsize(self.out), "finish_seconds": time.perf_counter() - t0,
                 "copy_seconds": self.copy_seconds, "units_live": self.units, "threads": self.threads,
                 "engine": "map" if merge_engine() else "fd", "order": f"round robin, {self.q} {'records' if self.kind == 'blow5' else 'signal batches'} per rank and turn"}
        return n, stats

    def _close_shards(self) -> None:
        for r, fd in enumerate(self.fds):
            if fd is not None:
                self.fds[r] = None                     # (ours no longer: the number may be handed out again at once)
                try:
                    os.close(fd)
                except OSError:
                    pass

    def _close_out(self) -> None:
        fd, self.out_fd = self.out_fd, None
        if fd is not None:
            try:
                os.close(fd)
            except OSError:
                pass

    def abort(self) -> None:
        """Gives up.  While nothing has been punched out of the rank files they still hold everything and the partial output goes;
        once pages have been given back the partial output is the ONLY copy of those records: it stays, and the log says where."""
        self._close_shards()
        self._close_out()
        if self._punch is not None:
            try:
                self._punch.finish()
            except Exception:
                pass
        if not os.path.exists(self.out):
            return
        if self.punched:
            logger.error(f"live join failed after {self.punched} bytes of the rank files had been released (they read as zeros there now): "
                         f"the records copied so far are kept in {self.out} (incomplete: no end marker / tables); the rank files "
                         f"{', '.join(self.paths)} keep what had not been copied yet. Re-run with --join after to be safe.")
        else:
            os.remove(self.out)


class _Puncher:
    """Frees the pages of a shard file behind the live join (fallocate PUNCH_HOLE | KEEP_SIZE on the whole pages of the range, on a
    helper thread): the rank files are staging buffers then, and the peak is the output plus what is in flight, not twice the output."""

    def __init__(self):
        import ctypes as C
        self.libc = C.CDLL(None, use_errno=True)
        self.libc.fallocate.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64]
        self.ex = ThreadPoolExecutor(max_workers=1, thread_name_prefix="s2s-punch")
        self.pending = []

    def punch(self, fd: int, lo: int, hi: int) -> None:
        lo, hi = (lo + 4095) & ~4095, hi & ~4095
        if hi > lo:
            self.pending.append(self.ex.submit(self.libc.fallocate, fd, 3, lo, hi - lo))      # FALLOC_FL_KEEP_SIZE | FALLOC_FL_PUNCH_HOLE
            if len(self.pending) > 256:
                self.pending = [f for f in self.pending if not f.done()]

    def finish(self) -> None:
        for f in self.pending:
            f.result()
        self.ex.shutdown()
