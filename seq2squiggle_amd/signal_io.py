"""SLOW5 / BLOW5 / POD5 writers for the predict path.

Reference: signal_io.py:62-172 (BLOW5Writer on pyslow5) and 175-282 (POD5Writer on pod5).  Neither
library exists in this image, so the SLOW5 ASCII and BLOW5 binary encodings (slow5 specification
v0.2.0) are written natively.  The writer protocol is the reference's: the model sets
`writer.signals = {read_id: 1-D fp32 tensor}` and calls `writer.save()`; every save appends.

Deviations from the reference, on purpose:
  * read ids / read_number count over the whole run, not per save() call (the reference restarts
    `idx` at every export batch and so writes duplicate `indexed_uuid`s, signal_io.py:123,145);
  * BLOW5 records are zlib-compressed (pyslow5's default record method) with the int16 signal stored as is (signal method
    "none" in the file header).  pyslow5's default signal method, svb-zd, and zstd records are available
    (`signal_compression="svb-zd"`, `record_compression="zstd"`, or S2S_BLOW5_SIGNAL / S2S_BLOW5_RECORD in the environment):
    the codec itself is pinned by known-answer vectors (codecs.py), but how slow5lib frames the compressed signal inside a
    record is written from memory of its source (see _blow5_record) and could not be checked against the library, which
    is why it is opt-in.
  * POD5 goes through the native container writer of pod5_io.py: record content pinned against the reference, signal rows
    VBZ-compressed like the pod5 library's (codecs.py), container unvalidated against libpod5.
"""
import logging
import os
import struct
import uuid
import zlib
from datetime import datetime

import numpy as np

logger = logging.getLogger("seq2squiggle")


def indexed_uuid(index: int) -> uuid.UUID:
    """UUID-shaped incrementing id (signal_io.py:19-23)."""
    return uuid.UUID(f"00000000-0000-0000-0000-{index:012d}")


_KITS = {
    "rna-004": {"seq_kit": "sqk-rna004", "prom": "FLO-PRO004RA", "min": "FLO-MIN004RA"},
    "rna-002": {"seq_kit": "sqk-rna002", "prom": "FLO-PRO002", "min": "FLO-MIN106"},
    "dna-r10": {"seq_kit": "SQK-LSK114", "prom": "FLO-PRO114", "min": "FLO-MIN114"},
    "dna-r9": {"seq_kit": "SQK-LSK109", "prom": "FLO-PRO001", "min": "FLO-MIN110"},
}


def get_seq_kit_and_flow_cell(profile_name: str):
    """(sequencing kit, flow cell product code) of a profile (signal_io.py:26-60)."""
    for prefix, data in _KITS.items():
        if profile_name.startswith(prefix):
            key = "prom" if "prom" in profile_name else "min" if "min" in profile_name else None
            if key is None:
                break
            return data["seq_kit"], data[key]
    raise ValueError(f"Unsupported profile name: {profile_name}")


def signal_to_dac(signal: np.ndarray, digitisation: float, signal_range: float, offset: float, rna: bool) -> np.ndarray:
    """pA -> int16 (signal_io.py:134-141): float32 arithmetic, round-half-even, C cast (wraps)."""
    s = np.asarray(signal, dtype=np.float32)
    with np.errstate(all="ignore"):
        raw = np.round(s * float(digitisation) / float(signal_range) - float(offset))
    raw = raw.astype(np.int64).astype(np.int16)
    return np.ascontiguousarray(raw[::-1]) if rna else raw


_COLS = ("read_id", "read_group", "digitisation", "offset", "range", "sampling_rate", "len_raw_signal", "raw_signal",
         "channel_number", "median_before", "read_number", "start_mux", "start_time")
_TYPES = ("char*", "uint32_t", "double", "double", "double", "double", "uint64_t", "int16_t*",
          "char*", "double", "int32_t", "uint8_t", "uint64_t")


def _fmt_double(x: float) -> str:
    return repr(float(x))


def _skip_record_draws(n_reads: int) -> None:
    """Advance np.random past the per-read offset / median_before draws (two normals per read, signal_io.py:129-133) of the
    `n_reads` reads that earlier rank shards own, so that a shard writer continues the single-process stream."""
    for _ in range(n_reads // 65536):
        np.random.normal(size=2 * 65536)
    if n_reads % 65536:
        np.random.normal(size=2 * (n_reads % 65536))


_WARNED = set()


def warn_once(key: str, message: str) -> None:
    """A standing caveat (a container layout nobody has opened with the real library) is said once per process, not once per writer."""
    if key not in _WARNED:
        _WARNED.add(key)
        logger.warning(message)


def cpu_share() -> int:
    """Worker threads this process may keep busy: the cores it is allowed on, capped by the container's CPU quota (cgroup
    cpu.max -- more runnable threads than the quota buys get the whole group throttled for the rest of the scheduler period,
    a stall of tens of milliseconds), divided between the ranks of a multi-process run on this node."""
    if os.environ.get("S2S_CPU_SHARE"):                # explicit override (tools/host_scaling.py: what if a rank only gets n threads?)
        return max(1, int(os.environ["S2S_CPU_SHARE"]))
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    cores = len(os.sched_getaffinity(0))
    if os.environ.get("S2S_PINNED_CPUS"):              # placement.pin_rank has narrowed the mask to this rank's share already
        cores *= ranks
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                          # cgroup v2: "<quota> <period>" | "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                quota, period = int(f.read()), int(g.read())
            if quota > 0:
                cores = min(cores, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, min(128, cores // ranks))


class BLOW5Writer:
    """Writes `.slow5` (ASCII) or `.blow5` (binary) by the file extension."""

    RECORD_METHODS = {"none": 0, "zlib": 1, "zstd": 2}     # on-disk codes (slow5lib slow5_encode_record_press)
    SIGNAL_METHODS = {"none": 0, "svb-zd": 1}              # (slow5_encode_signal_press)

    def __init__(self, filename, profile, ideal_mode, profile_name, preserve_read_ids, record_compression=None,
                 signal_compression=None):
        self.filename = str(filename)
        self.profile = profile
        self.ideal_mode = ideal_mode
        self.profile_name = profile_name
        self.preserve_read_ids = preserve_read_ids
        self.record_compression = record_compression or os.environ.get("S2S_BLOW5_RECORD", "zlib")
        self.signal_compression = signal_compression or os.environ.get("S2S_BLOW5_SIGNAL", "none")
        if self.record_compression not in self.RECORD_METHODS or self.signal_compression not in self.SIGNAL_METHODS:
            raise ValueError(f"BLOW5 compression must be one of {sorted(self.RECORD_METHODS)} x {sorted(self.SIGNAL_METHODS)}")
        self.signals = None
        self.dac = None                       # optional {read_id: int16 array} computed on the GPU
        self.median_before = float(profile["median_before_mean"])
        self.median_before_std = float(profile["median_before_std"])
        self.offset = float(profile["offset_mean"])
        self.offset_std = float(profile["offset_std"])
        self.digitisation = float(profile["digitisation"])
        self.signal_range = float(profile["range"])
        self.sample_rate = float(profile["sample_rate"])
        self.start_time = 0
        self.n_written = 0
        self.binary = self.filename.endswith(".blow5")
        self.compress_level = 1               # zlib / zstd level of BLOW5 records; any level is a valid stream
        # zlib records: "huffman" = the library's own Huffman-only deflate (a noisy signal has no LZ77 matches worth the search:
        # within 3 % of libdeflate level 1's size at a fifth of its CPU time, long records in parallel pieces); "lz" = libdeflate
        self.deflate = "huffman"
        self.threads = cpu_share()            # compression threads (the reference: cpu_count)
        self._out = None                      # packed records of the batch being written
        self._warned_fallback = False
        if self.signal_compression == "svb-zd":
            warn_once("svb-zd", "BLOW5 signal compression svb-zd is EXPERIMENTAL here: the codec is pinned by known-answer vectors, "
                           "but how slow5lib frames the blob inside a record could not be checked against the library; files "
                           "written this way may not open in slow5tools. The default (zlib records, raw int16 signal) is safe.")

    def start_at(self, read_index: int) -> None:
        """Rank shards (parallel.py): `read_index` records precede this writer's first one in the whole job (inference_run counts
        the reads of earlier shards that have at least one chunk), so read ids and read_number continue from there and the
        record draws continue the single-process np.random stream.  One caveat remains: a read whose signal strips to nothing
        (every sample exactly 0) writes no record, which a later shard cannot know -- ids stay unique, but from there on they
        differ from the single-process numbering.  start_time stays per file."""
        if self.n_written:
            raise RuntimeError("start_at() must come before the first record")
        self.n_written = int(read_index)
        if not self.ideal_mode:
            _skip_record_draws(int(read_index))

    # ------------------------------------------------------------------ header
    def header_attributes(self) -> dict:
        seq_kit, flow_cell = get_seq_kit_and_flow_cell(self.profile_name)
        return {
            "asic_id": "asic_id_0",
            "exp_start_time": datetime.now().strftime("%Y-%m-%dT%H:%M:%SZ"),
            "run_id": "run_id_0",
            "flow_cell_id": "FAN00000",
            "flow_cell_product_code": flow_cell,
            "experiment_type": "rna" if self.profile_name.startswith("rna") else "genomic_dna",
            "sample_frequency": int(self.sample_rate),
            "sequencing_kit": seq_kit,
        }

    def _header_text(self) -> str:
        lines = ["#slow5_version\t0.2.0", "#num_read_groups\t1"]
        for k, v in sorted(self.header_attributes().items()):
            lines.append(f"@{k}\t{v}")
        lines.append("#" + "\t".join(_TYPES))
        lines.append("#" + "\t".join(_COLS))
        return "\n".join(lines) + "\n"

    # ------------------------------------------------------------------ records
    def _record(self, read_id, raw, n_samples=None):
        """One record as the reference builds it (signal_io.py:123-161).  n_samples: the read's length when `raw` is not
        carried (the samples arrive compressed, svb_records)."""
        n_samples = len(raw) if n_samples is None else n_samples
        if self.ideal_mode:
            median_before_value, offset_value = self.median_before, self.offset
        else:
            median_before_value = np.random.normal(self.median_before, self.median_before_std)
            offset_value = np.random.normal(self.offset, self.offset_std)
        self.n_written += 1
        rid = read_id if self.preserve_read_ids else indexed_uuid(self.n_written)
        rec = {"read_id": str(rid), "read_group": 0, "digitisation": self.digitisation, "offset": offset_value,
               "range": self.signal_range, "sampling_rate": self.sample_rate, "len_raw_signal": n_samples,
               "signal": raw, "channel_number": "0", "median_before": median_before_value,
               "read_number": self.n_written - 1, "start_mux": 0, "start_time": self.start_time}
        self.start_time += n_samples
        return rec

    def records(self):
        """The records of the current `signals` (pA tensors; converted here like signal_io.py:134-141)."""
        rna = self.profile_name.startswith("rna")
        for read_id, signal in self.signals.items():
            if len(signal) == 0:
                logger.debug("Empty signal, skipping {}".format(read_id))
                continue
            if self.dac is not None and read_id in self.dac:
                raw = np.asarray(self.dac[read_id], dtype=np.int16)
            else:
                sig = signal.detach().cpu().numpy() if hasattr(signal, "detach") else np.asarray(signal)
                raw = signal_to_dac(sig, self.digitisation, self.signal_range, self.offset, rna)
            yield self._record(read_id, raw)

    def save_dac(self, read_ids, dac: np.ndarray, offsets: np.ndarray):
        """Streaming path: samples already converted to int16 on the GPU (s2s_export_reads), packed read after read;
        read r is dac[offsets[r]:offsets[r+1]].  Same records and append semantics as save()."""
        self._write(self.dac_records(read_ids, dac, offsets))

    def dac_records(self, read_ids, dac: np.ndarray, offsets: np.ndarray) -> list:
        """The records save_dac() writes, built now (read numbering and the offset draws happen here, in order)."""
        return [self._record(rid, dac[offsets[i]:offsets[i + 1]]) for i, rid in enumerate(read_ids)
                if offsets[i + 1] > offsets[i]]

    def write_records(self, recs) -> None:
        """Append already-built records; safe to call from one background thread at a time."""
        self._write(recs)

    def gpu_signal_rows(self):
        """(StreamVByte variant, samples per row) when the streaming path should encode the signal on the GPU
        (s2s_svb_encode) instead of handing over int16 samples: svb-zd BLOW5 = one 32-bit-variant blob per read."""
        return (32, 1 << 40) if self.binary and self.signal_compression == "svb-zd" else None

    def svb_records(self, read_ids, offsets, row_read, row_offsets, blob) -> list:
        """dac_records() for reads whose samples arrive as svb-zd blobs: row i (read row_read[i]) is
        blob[row_offsets[i]:row_offsets[i+1]]; a read with no samples has an empty blob and is skipped."""
        recs = []
        for i in range(len(row_read)):
            r = int(row_read[i])
            n = int(offsets[r + 1] - offsets[r])
            if n <= 0:
                continue
            rec = self._record(read_ids[r], np.zeros(0, np.int16), n_samples=n)
            rec["svb"] = blob[row_offsets[i]:row_offsets[i + 1]]
            recs.append(rec)
        return recs

    def save(self):
        if self.signals is None:
            logger.warning("SLOW5 was not exported. No signals were found")
            raise ValueError("SLOW5 was not exported. No signals were found")
        self._write(self.records())

    def _write(self, recs):
        append = os.path.exists(self.filename)
        if self.binary:
            self._save_blow5(append, recs)
        else:
            self._save_slow5(append, recs)

    def _save_slow5(self, append: bool, recs):
        with open(self.filename, "a" if append else "w") as f:
            if not append:
                f.write(self._header_text())
            for r in recs:
                f.write("\t".join([r["read_id"], str(r["read_group"]), _fmt_double(r["digitisation"]),
                                   _fmt_double(r["offset"]), _fmt_double(r["range"]), _fmt_double(r["sampling_rate"]),
                                   str(r["len_raw_signal"]), ",".join(map(str, r["signal"].tolist())),
                                   r["channel_number"], _fmt_double(r["median_before"]), str(r["read_number"]),
                                   str(r["start_mux"]), str(r["start_time"])]) + "\n")

    # BLOW5 v0.2.0: 64-byte file header, u32 size + ASCII header, records (u64 size + zlib stream), "5WOLB"
    _EOF = b"5WOLB"

    def _record_fields(self, r):
        """(bytes before raw_signal, signal bytes or array, bytes after it) of one record body: read_id_len u16, read_id,
        read_group u32, digitisation, offset, range, sampling_rate f64, len_raw_signal u64 | raw_signal | the auxiliary
        fields in header order.  With signal method svb-zd the raw_signal bytes are the codec's blob (u32 sample count +
        StreamVByte stream) and -- as remembered from slow5lib's slow5_rec_to_mem / slow5_rec_parse, NOT verified against the
        library -- the len_raw_signal field then carries the blob's byte length (the parser needs it to find the auxiliary
        fields; the sample count is inside the blob)."""
        rid, ch = r["read_id"].encode(), r["channel_number"].encode()
        if self.signal_compression == "svb-zd":
            from .codecs import svb_zd_compress
            sig = r.get("svb")
            if sig is None:
                sig = svb_zd_compress(np.ascontiguousarray(r["signal"]).astype("<i2"))
            sig = np.frombuffer(sig, dtype=np.uint8)
            n_field = sig.size
        else:
            sig = np.ascontiguousarray(r["signal"]).astype("<i2", copy=False).view(np.uint8)
            n_field = r["len_raw_signal"]
        head = (struct.pack("<H", len(rid)) + rid
                + struct.pack("<IddddQ", r["read_group"], r["digitisation"], r["offset"], r["range"], r["sampling_rate"], n_field))
        tail = (struct.pack("<H", len(ch)) + ch
                + struct.pack("<diBQ", r["median_before"], r["read_number"], r["start_mux"], r["start_time"]))
        return head, sig, tail

    def _blow5_record(self, r) -> bytes:
        """u64 size + the (record-compressed) body of ONE record, in Python (tests pin _pack_native against it)."""
        head, sig, tail = self._record_fields(r)
        body = head + sig.tobytes() + tail
        if self.record_compression == "zlib":
            body = zlib.compress(body, self.compress_level)
        elif self.record_compression == "zstd":
            from .codecs import zstd_compress
            body = zstd_compress(body, 1)
        return struct.pack("<Q", len(body)) + body

    def _pack_native(self, recs) -> memoryview:
        """All records of a batch, framed and compressed by s2s_blow5_pack (libs2s_hip.so, host threads)."""
        import ctypes as C
        from ._lib import lib
        fields = [self._record_fields(r) for r in recs]
        n = len(fields)

        def flat(parts):
            offs = np.zeros(n + 1, np.int64)
            np.cumsum([len(p) for p in parts], out=offs[1:])
            return np.frombuffer(b"".join(parts), dtype=np.uint8) if offs[-1] else np.zeros(1, np.uint8), offs
        head, head_offs = flat([f[0] for f in fields])
        tail, tail_offs = flat([f[2] for f in fields])
        sigs = [f[1] for f in fields]
        sig_offs = np.zeros(n + 1, np.int64)
        np.cumsum([s_.size for s_ in sigs], out=sig_offs[1:])
        # the records of a super-batch are consecutive slices of one packed array (run_streaming): no copy then
        first = sigs[0].__array_interface__["data"][0]
        contiguous = all(s_.__array_interface__["data"][0] == first + int(sig_offs[i]) for i, s_ in enumerate(sigs))
        sig = np.concatenate(sigs) if not contiguous else None
        sig_ptr = sig.ctypes.data if sig is not None else first
        total = int(head_offs[-1] + tail_offs[-1] + sig_offs[-1])
        L = lib()
        cap = int(L.s2s_blow5_pack_bound(total, n))
        if self._out is None or self._out.size < cap:              # kept between batches: no fresh pages per call
            self._out = np.empty(cap + cap // 4, np.uint8)
        out = self._out
        method = self.RECORD_METHODS[self.record_compression]
        if self.record_compression == "zlib" and self.deflate == "huffman":
            method = 3
        got = L.s2s_blow5_pack(head.ctypes.data, head_offs.ctypes.data, tail.ctypes.data, tail_offs.ctypes.data, C.c_void_p(sig_ptr),
                               sig_offs.ctypes.data, n, method, self.compress_level,
                               self.threads, out.ctypes.data, cap)   # (`fields` keeps the sample arrays alive across the call)
        if got < 0:
            raise RuntimeError(f"s2s_blow5_pack failed ({got})")
        return memoryview(out)[:got]

    def _save_blow5(self, append: bool, recs):
        if append:
            with open(self.filename, "r+b") as f:      # drop the end-of-file marker, then append
                f.seek(-len(self._EOF), os.SEEK_END)
                if f.read(len(self._EOF)) == self._EOF:
                    f.seek(-len(self._EOF), os.SEEK_END)
                    f.truncate()
        with open(self.filename, "ab" if append else "wb") as f:
            if not append:
                hdr = self._header_text().encode()
                fh = (b"BLOW5\x01" + bytes([0, 2, 0]) + bytes([self.RECORD_METHODS[self.record_compression]])
                      + struct.pack("<I", 1) + bytes([self.SIGNAL_METHODS[self.signal_compression]]))
                f.write(fh + bytes(64 - len(fh)))
                f.write(struct.pack("<I", len(hdr)) + hdr)
            # framing + compression on native worker threads (s2s_blow5_pack), one write per batch
            # (the reference: write_record_batch(threads=cpu_count), signal_io.py:167-171)
            recs = recs if isinstance(recs, list) else list(recs)
            if recs:
                try:
                    packed = self._pack_native(recs)
                except (RuntimeError, OSError) as e:          # library not loadable / codec missing: same bytes layout, in Python
                    if not self._warned_fallback:
                        logger.warning(f"native BLOW5 record packer unavailable ({e}); compressing records in Python")
                        self._warned_fallback = True
                    packed = b"".join(map(self._blow5_record, recs))
                f.write(packed)
            f.write(self._EOF)


def read_blow5(path):
    """Minimal reader of the files written above (tests / round-trip only): -> (header text, list of records)."""
    it = iter_blow5(path)
    header = next(it)
    return header, list(it)


def iter_blow5(path):
    """read_blow5() with bounded memory (the file is memory-mapped): yields the header text, then one record at a time."""
    import mmap
    from . import codecs
    with open(path, "rb") as f:
        data = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    assert data[:6] == b"BLOW5\x01" and data[-5:] == BLOW5Writer._EOF
    rec_m, sig_m = data[9], data[14]
    hlen = struct.unpack_from("<I", data, 64)[0]
    yield data[68:68 + hlen].decode()
    pos = 68 + hlen
    while pos < len(data) - 5:
        n = struct.unpack_from("<Q", data, pos)[0]
        body = data[pos + 8:pos + 8 + n]
        if rec_m == 1:
            body = zlib.decompress(body)
        elif rec_m == 2:
            body = codecs.zstd_decompress(body, codecs.zstd_frame_content_size(body))
        pos += 8 + n
        o = 0
        (ln,) = struct.unpack_from("<H", body, o); o += 2
        rid = body[o:o + ln].decode(); o += ln
        rg, dig, off, rng, sr, nsig = struct.unpack_from("<IddddQ", body, o); o += struct.calcsize("<IddddQ")
        if sig_m == 1:
            sig, used = codecs.svb_zd_decompress(body[o:o + nsig])
            assert used == nsig
            o += nsig
            nsig = len(sig)
        else:
            sig = np.frombuffer(body, dtype="<i2", count=nsig, offset=o); o += 2 * nsig
        (cl,) = struct.unpack_from("<H", body, o); o += 2
        ch = body[o:o + cl].decode(); o += cl
        mb, rn, mux, st_ = struct.unpack_from("<diBQ", body, o)
        yield {"read_id": rid, "read_group": rg, "digitisation": dig, "offset": off, "range": rng,
               "sampling_rate": sr, "len_raw_signal": nsig, "signal": sig, "channel_number": ch,
               "median_before": mb, "read_number": rn, "start_mux": mux, "start_time": st_}


def merge_shards(paths, out: str, threads: int = None, consume: bool = False) -> int:
    """The out.rankN files of a sharded run (parallel.rank_output_path) -> the ONE file the reference writes (inference.py:65-79):
    BLOW5 / SLOW5 shards with identical headers (header of the first, every shard's records in order, one end-of-file marker),
    POD5 shards by pod5_io.merge_pod5.  Nothing is decompressed and no record passes through the interpreter: the payload moves as
    byte ranges with copy_file_range (merge.py: one writer per destination file, which is what the file system rewards).
    consume=True: the first shard becomes the output and every other one is deleted as soon as its bytes are in (what
    `predict --gpus N` does).  -> number of records; merge_shards.last = {"seconds", "bytes_copied", "bytes"}.
    Header attributes that differ between shards (only the wall-clock exp_start_time may) are taken from the first."""
    from . import merge as M
    paths = list(paths)
    if not paths:
        raise ValueError("no shard files given")
    missing = [p_ for p_ in paths if not os.path.exists(p_)]
    if missing:
        raise FileNotFoundError(f"shard file(s) missing: {', '.join(missing)}")
    pod5 = all(p_.endswith(".pod5") for p_ in paths)
    binary = [p_.endswith(".blow5") for p_ in paths]
    if pod5:
        if not out.endswith(".pod5"):
            raise ValueError("POD5 shards merge into a .pod5 file")
        if os.path.exists(out):
            raise FileExistsError(f"{out} exists (the POD5 writer refuses to overwrite, like pod5.Writer)")
    elif any(b != binary[0] for b in binary) or (not binary[0] and not all(p_.endswith(".slow5") for p_ in paths)):
        raise ValueError("shards must be all .blow5, all .slow5 or all .pod5")
    ext = os.path.splitext(out)[1]
    tmp = out[:len(out) - len(ext)] + ".partial" + ext              # the merged file appears under its name only when it is whole
    if os.path.exists(tmp):
        os.remove(tmp)
    take_first = consume and (pod5 or binary[0])
    try:
        if pod5:
            from .pod5_io import merge_pod5
            n = merge_pod5(paths, tmp, threads=threads, take_first=take_first, consume=consume)
            stats = dict(merge_pod5.last)
        elif binary[0]:
            n, stats = M.merge_blow5(paths, tmp, threads=threads, take_first=take_first, consume=consume)
        else:
            n, stats = M.merge_slow5(paths, tmp, threads=threads)
        stats["bytes"] = os.path.getsize(tmp)
        os.replace(tmp, out)
    except BaseException:
        if take_first and os.path.exists(tmp) and not os.path.exists(paths[0]):
            # the first shard had already become the partial output: it is no shard any more, say where the bytes are
            left = [p_ for p_ in paths[1:] if os.path.exists(p_)]
            logger.error(f"merge failed after {paths[0]} had been taken over as {tmp}: that file holds the records of the shards consumed "
                         f"so far (it has no valid end yet); still on disk, untouched: {', '.join(left) or 'none'}")
        elif os.path.exists(tmp):
            os.remove(tmp)
        raise
    if consume:
        for p_ in paths:                     # (SLOW5 shards; BLOW5 / POD5 shards went while the merge ran)
            if os.path.exists(p_):
                os.remove(p_)
    merge_shards.last = stats
    return n


merge_shards.last = {}


def read_slow5(path):
    """Minimal SLOW5 ASCII reader (tests only)."""
    header, recs = [], []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(("#", "@")):
                header.append(line)
                continue
            v = line.split("\t")
            recs.append({"read_id": v[0], "read_group": int(v[1]), "digitisation": float(v[2]), "offset": float(v[3]),
                         "range": float(v[4]), "sampling_rate": float(v[5]), "len_raw_signal": int(v[6]),
                         "signal": np.array(v[7].split(","), dtype=np.int16) if v[7] else np.zeros(0, np.int16),
                         "channel_number": v[8], "median_before": float(v[9]), "read_number": int(v[10]),
                         "start_mux": int(v[11]), "start_time": int(v[12])})
    return "\n".join(header) + "\n", recs


class POD5Writer:
    """The reference's POD5Writer (signal_io.py:175-287) on the native container writer of pod5_io.py.  The records
    (ids, calibration, int16 samples, run info) are pinned against the reference; the container is UNVALIDATED against
    libpod5 (see pod5_io.py; tools/validate_containers.py checks it where the pod5 package exists); signal rows are VBZ-compressed
    like libpod5's (S2S_POD5_SIGNAL=none stores them uncompressed)."""

    def __init__(self, filename, profile, ideal_mode, profile_name, preserve_read_ids):
        self.filename = str(filename)
        self.profile = profile
        self.ideal_mode = ideal_mode
        self.profile_name = profile_name
        self.preserve_read_ids = preserve_read_ids
        self.signals = None
        self.dac = None
        self.median_before = float(profile["median_before_mean"])
        self.median_before_std = float(profile["median_before_std"])
        self.offset = float(profile["offset_mean"])
        self.offset_std = float(profile["offset_std"])
        self.digitisation = float(profile["digitisation"])
        self.signal_range = float(profile["range"])
        self.sample_rate = float(profile["sample_rate"])
        self.start_time = 0
        self._stream, self._stream_idx, self._stream_run_info = None, 0, None

    def start_at(self, read_index: int) -> None:
        """Rank shards: see BLOW5Writer.start_at (streaming path: read ids / read_number continue from read_index)."""
        if self._stream_idx:
            raise RuntimeError("start_at() must come before the first record")
        self._stream_idx = int(read_index)
        if not self.ideal_mode:
            _skip_record_draws(int(read_index))

    def run_info(self) -> dict:
        """signal_io.py:210-231."""
        seq_kit, flow_cell = get_seq_kit_and_flow_cell(self.profile_name)
        now = datetime.now()
        return dict(acquisition_id="", acquisition_start_time=now, adc_max=4095, adc_min=-4096, context_tags={},
                    experiment_name="", flow_cell_id="", flow_cell_product_code=flow_cell, protocol_name="",
                    protocol_run_id="", protocol_start_time=now, sample_id="test", sample_rate=int(self.sample_rate),
                    sequencing_kit=seq_kit, sequencer_position="", sequencer_position_type="", software="", system_name="",
                    system_type="", tracking_id={})

    def _record(self, idx, read_id, raw, run_info) -> dict:
        """What the reference passes to pod5.Read for one read (signal_io.py:240-281)."""
        if self.ideal_mode:
            median_before_value, offset_value = self.median_before, self.offset
        else:
            median_before_value = np.random.normal(self.median_before, self.median_before_std)
            offset_value = np.random.normal(self.offset, self.offset_std)
        rid = uuid.uuid5(uuid.NAMESPACE_DNS, read_id) if self.preserve_read_ids else indexed_uuid(idx + 1)
        return dict(read_id=rid, signal=raw, read_number=idx, start_sample=0, median_before=median_before_value,
                    channel=123, well=3, pore_type="not_set", calibration_offset=offset_value,
                    calibration_scale=self.signal_range / self.digitisation, end_reason="signal_positive",
                    end_reason_forced=False, run_info=run_info)

    def records(self) -> list:
        """One dict per non-empty read of `signals` (pA tensors; converted here like signal_io.py:246-253)."""
        rna = self.profile_name.startswith("rna")
        run_info = self.run_info()
        recs = []
        for idx, (read_id, signal) in enumerate(self.signals.items(), start=self._stream_idx):   # (start_at: rank shards)
            if len(signal) == 0:
                logger.debug("Empty signal, skipping {}".format(read_id))
                continue
            if self.dac is not None and read_id in self.dac:
                raw = np.asarray(self.dac[read_id], dtype=np.int16)
            else:
                sig = signal.detach().cpu().numpy() if hasattr(signal, "detach") else np.asarray(signal)
                raw = signal_to_dac(sig, self.digitisation, self.signal_range, self.offset, rna)
            recs.append(self._record(idx, read_id, raw, run_info))
        return recs

    def save(self):
        """The reference's one-shot export (everything in `signals`, a new file)."""
        if self.signals is None:
            logger.warning("POD5 was not exported. No signals were found")
            raise ValueError("POD5 was not exported. No signals were found")
        from . import pod5_io
        warn_once("pod5", "POD5 output comes from seq2squiggle_amd's own container writer (unvalidated against ONT's pod5 "
                  "library, which this image lacks).")
        pod5_io.write_pod5(self.filename, self.records())

    # ---- streaming path (inference.run_streaming): samples already int16 on the GPU, reads arrive in super-batches and
    #      the signal table grows on disk, so memory stays bounded (the reference keeps every read in RAM, inference.py:72-79)
    def dac_records(self, read_ids, dac: np.ndarray, offsets: np.ndarray) -> list:
        if self._stream_run_info is None:
            self._stream_run_info = self.run_info()
        recs = []
        for i, rid in enumerate(read_ids):
            idx = self._stream_idx
            self._stream_idx += 1
            if offsets[i + 1] > offsets[i]:
                recs.append(self._record(idx, rid, dac[offsets[i]:offsets[i + 1]], self._stream_run_info))
        return recs

    def gpu_signal_rows(self):
        """(StreamVByte variant, samples per row): the svb16 stage of VBZ runs on the GPU, one stream per signal-table row."""
        from . import pod5_io
        return (16, pod5_io.SIGNAL_CHUNK) if os.environ.get("S2S_POD5_SIGNAL", "vbz") == "vbz" else None

    def svb_records(self, read_ids, offsets, row_read, row_offsets, blob) -> list:
        """dac_records() for reads whose samples arrive as svb16 streams, one per signal-table row (row i belongs to read
        row_read[i]; rows past a read's end are empty).  The zstd stage follows in write_records()."""
        from . import pod5_io
        if self._stream_run_info is None:
            self._stream_run_info = self.run_info()
        rows_of = {}
        for i in range(len(row_read)):
            if row_offsets[i + 1] > row_offsets[i]:
                rows_of.setdefault(int(row_read[i]), []).append(i)
        recs = []
        for r, rid in enumerate(read_ids):
            idx = self._stream_idx
            self._stream_idx += 1
            n = int(offsets[r + 1] - offsets[r])
            if n <= 0:
                continue
            rec = self._record(idx, rid, None, self._stream_run_info)
            rec["num_samples"] = n
            rec["svb_rows"] = [(blob[row_offsets[i]:row_offsets[i + 1]], min(pod5_io.SIGNAL_CHUNK, n - j * pod5_io.SIGNAL_CHUNK))
                               for j, i in enumerate(rows_of.get(r, []))]
            recs.append(rec)
        return recs

    def _zstd_rows(self, recs) -> None:
        """svb_rows -> vbz_rows: every svb16 stream of the batch becomes one zstd frame, on the native worker threads."""
        import ctypes as C
        from ._lib import lib
        rows = [row for r in recs for row in r.get("svb_rows", ())]
        if not rows:
            return
        offs = np.zeros(len(rows) + 1, np.int64)
        np.cumsum([len(b) for b, _ in rows], out=offs[1:])
        # svb_records hands over consecutive slices of the one blob that came off the GPU: no copy then
        first = rows[0][0].__array_interface__["data"][0]
        if all(b.__array_interface__["data"][0] == first + int(offs[i]) for i, (b, _) in enumerate(rows)):
            src, keep = C.c_void_p(first), rows
        else:
            keep = np.concatenate([np.frombuffer(b, np.uint8) for b, _ in rows])
            src = C.c_void_p(keep.ctypes.data)
        L = lib()
        cap = int(L.s2s_blow5_pack_bound(int(offs[-1]), len(rows)))
        out, out_offs = np.empty(cap, np.uint8), np.zeros(len(rows) + 1, np.int64)
        got = L.s2s_compress_rows(src, offs.ctypes.data, len(rows), 2, 1, cpu_share(),
                                  out.ctypes.data, cap, out_offs.ctypes.data)
        del keep
        if got < 0:                                   # S2S_ERR_CODEC: libzstd.so.1 not loadable -> the pyarrow / ctypes codec of codecs.py
            from .codecs import zstd_compress
            logger.warning(f"s2s_compress_rows failed ({got}); compressing signal rows in Python")
            blobs = [zstd_compress(bytes(b), 1) for b, _ in rows]
            np.cumsum([len(b) for b in blobs], out=out_offs[1:])
            out = np.frombuffer(b"".join(blobs), dtype=np.uint8)
        view = memoryview(out)
        i = 0
        for r in recs:
            if "svb_rows" in r:
                k = len(r["svb_rows"])
                r["vbz_rows"] = [(view[out_offs[i + j]:out_offs[i + j + 1]], r["svb_rows"][j][1]) for j in range(k)]
                del r["svb_rows"]
                i += k

    def write_records(self, recs) -> None:
        self._zstd_rows(recs)
        if self._stream is None:
            from . import pod5_io
            warn_once("pod5", "POD5 output comes from seq2squiggle_amd's own container writer: record content and the VBZ codec "
                      "are pinned against the reference / known-answer vectors, but no file has been opened with ONT's "
                      "pod5 library yet (absent from this image). Prefer .blow5 where the consumer allows it.")
            self._stream = pod5_io.Pod5FileWriter(self.filename)
        self._stream.add_reads(recs)

    def close(self) -> None:
        if self._stream is not None:
            self._stream.close()
            self._stream = None
