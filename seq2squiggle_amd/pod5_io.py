"""Native POD5 container writer -- UNVALIDATED against the pod5 library.

The reference writes POD5 through ONT's `pod5` package (signal_io.py:175-287), which this image lacks, so the
container below follows the published format description (pod5-file-format: docs/SPECIFICATION.md, docs/tables/*.toml,
c++/pod5_format/footer.fbs) from the author's knowledge of it, with pyarrow for the embedded Arrow IPC files and a
hand-written flatbuffer for the footer.  What IS pinned: the per-read record content (ids, calibration, int16 samples,
run info) against what the reference's POD5Writer hands to the library (tests/golden/pod5_records.npz), and a round
trip through the reader in this file.  What is NOT: acceptance by libpod5 / dorado.  Signal rows are VBZ-compressed as
libpod5 does it (`large_binary` column with the "minknow.vbz" extension type; codecs.vbz_compress = svb16 zig-zag delta +
zstd level 1, pinned by known-answer vectors); `signal_compression="none"` (or S2S_POD5_SIGNAL=none) writes the format's
uncompressed variant (`large_list<int16>`) instead.

Layout:  signature | marker | signal table | pad8 | marker | run-info table | pad8 | marker | reads table | pad8 |
         marker | "FOOTER\\0\\0" | flatbuffer (size multiple of 8) | int64 footer length | marker | signature
"""
import datetime as _dt
import io
import struct
import uuid
from typing import Dict, List, Sequence

import numpy as np

SIGNATURE = b"\x8bPOD\r\n\x1a\n"
FOOTER_MAGIC = b"FOOTER\x00\x00"
POD5_VERSION = "0.1.0"          # reads-table spec v3 (flattened columns), before open_pore_level was added
SOFTWARE = "seq2squiggle_amd"
SIGNAL_CHUNK = 102400           # samples per signal-table row (libpod5's default)
SIGNAL_BATCH_ROWS, READ_BATCH_ROWS = 100, 1000    # fixed batch sizes: readers index rows as batch = row // size
CT_READS, CT_SIGNAL, CT_READ_ID_INDEX, CT_OTHER_INDEX, CT_RUN_INFO = 0, 1, 2, 3, 4
END_REASONS = ["unknown", "mux_change", "unblock_mux_change", "data_service_unblock_mux_change", "signal_positive",
               "signal_negative"]

RUN_INFO_FIELDS = ("acquisition_id", "acquisition_start_time", "adc_max", "adc_min", "context_tags", "experiment_name",
                   "flow_cell_id", "flow_cell_product_code", "protocol_name", "protocol_run_id", "protocol_start_time",
                   "sample_id", "sample_rate", "sequencing_kit", "sequencer_position", "sequencer_position_type", "software",
                   "system_name", "system_type", "tracking_id")


def _pa():
    import pyarrow as pa
    return pa


# ------------------------------------------------------------------------------------------------ footer flatbuffer
def build_footer(file_identifier: str, software: str, version: str, files: Sequence[tuple]) -> bytes:
    """footer.fbs: table Footer { file_identifier, software, pod5_version: string; contents: [EmbeddedFile]; }
    table EmbeddedFile { offset, length: int64; format: Format(short) = FeatherV2; content_type: ContentType(short); }
    Built front to back: every uoffset points forward, vtables sit in front of their tables."""
    buf = bytearray(8)                                   # [0] root uoffset, [4] padding

    def align(n):
        while len(buf) % n:
            buf.append(0)

    def string(s):
        align(4)
        pos = len(buf)
        b = s.encode()
        buf.extend(struct.pack("<I", len(b)) + b + b"\x00")
        return pos

    # root vtable + table
    align(4)
    vt = len(buf)
    buf.extend(struct.pack("<HHHHHH", 12, 20, 4, 8, 12, 16))
    root = len(buf)
    buf.extend(struct.pack("<i", root - vt) + bytes(16))
    struct.pack_into("<I", buf, 0, root)
    strings = [string(file_identifier), string(software), string(version)]
    for i, pos in enumerate(strings):
        struct.pack_into("<I", buf, root + 4 + 4 * i, pos - (root + 4 + 4 * i))
    # contents vector
    align(4)
    vec = len(buf)
    struct.pack_into("<I", buf, root + 16, vec - (root + 16))
    buf.extend(struct.pack("<I", len(files)) + bytes(4 * len(files)))
    # one vtable shared by all EmbeddedFile tables: soffset at 0, offset at 8, length at 16, format at 24, type at 26
    align(2)
    evt = len(buf)
    buf.extend(struct.pack("<HHHHHH", 12, 28, 8, 16, 24, 26))
    for i, (offset, length, content_type) in enumerate(files):
        align(8)
        t = len(buf)
        buf.extend(struct.pack("<iiqqhh", t - evt, 0, offset, length, 0, content_type))
        struct.pack_into("<I", buf, vec + 4 + 4 * i, t - (vec + 4 + 4 * i))
    align(8)
    return bytes(buf)


def parse_footer(fb: bytes) -> dict:
    def u32(p):
        return struct.unpack_from("<I", fb, p)[0]

    def field(table, idx):
        vt = table - struct.unpack_from("<i", fb, table)[0]
        vsize = struct.unpack_from("<H", fb, vt)[0]
        if 4 + 2 * idx >= vsize:
            return 0
        off = struct.unpack_from("<H", fb, vt + 4 + 2 * idx)[0]
        return table + off if off else 0

    def string(p):
        p += u32(p)
        return fb[p + 4: p + 4 + u32(p)].decode()

    root = u32(0)
    out = {"file_identifier": string(field(root, 0)), "software": string(field(root, 1)),
           "pod5_version": string(field(root, 2)), "contents": []}
    v = field(root, 3)
    v += u32(v)
    for i in range(u32(v)):
        p = v + 4 + 4 * i
        t = p + u32(p)
        get = lambda idx, fmt, default=0: struct.unpack_from(fmt, fb, field(t, idx))[0] if field(t, idx) else default
        out["contents"].append({"offset": get(0, "<q"), "length": get(1, "<q"), "format": get(2, "<h"),
                                "content_type": get(3, "<h")})
    return out


# ------------------------------------------------------------------------------------------------ arrow tables
def _uuid_field(pa):
    return pa.field("read_id", pa.binary(16), nullable=False,
                    metadata={"ARROW:extension:name": "minknow.uuid", "ARROW:extension:metadata": ""})


def _ipc_bytes(pa, schema, batches) -> bytes:
    sink = io.BytesIO()
    with pa.ipc.new_file(sink, schema) as w:
        for b in batches:
            w.write_batch(b)
    return sink.getvalue()


class _SubFile:
    """File-like view for pyarrow's IPC writer: positions are relative to where the embedded file starts."""

    def __init__(self, f):
        self.f, self.base, self.closed = f, f.tell(), False

    def write(self, b):
        return self.f.write(b)

    def tell(self):
        return self.f.tell() - self.base

    def flush(self):
        self.f.flush()

    def close(self):
        self.closed = True

    def writable(self):
        return True

    def readable(self):
        return False

    def seekable(self):
        return False


def _signal_schema(pa, meta, vbz=True):
    if vbz:
        sig = pa.field("signal", pa.large_binary(), nullable=False,
                       metadata={"ARROW:extension:name": "minknow.vbz", "ARROW:extension:metadata": ""})
    else:
        sig = pa.field("signal", pa.large_list(pa.int16()))
    return pa.schema([_uuid_field(pa), sig, pa.field("samples", pa.uint32())], metadata=meta)


def _signal_batch(pa, schema, rows, vbz=True):
    """rows: (read_id bytes, int16 array or None, VBZ bytes or None, sample count) per signal-table row."""
    counts = [r[3] for r in rows]
    ids = pa.array([r[0] for r in rows], pa.binary(16))
    if vbz:
        from .codecs import vbz_compress
        # the column is assembled from ONE data buffer + offsets.  (pa.array(list of memoryviews) must not be used here: pyarrow
        # 25 leaks a memoryview per item, each of which pins the whole compressed batch it is a slice of -- the writer's memory
        # then grows by the size of the output, 7 GB for one GPU's share of BASELINE configs[4].)
        blobs = [np.frombuffer(r[2] if r[2] is not None else vbz_compress(r[1]), dtype=np.uint8) for r in rows]
        offs = np.zeros(len(blobs) + 1, np.int64)
        np.cumsum([b.size for b in blobs], out=offs[1:])
        data = np.concatenate(blobs) if blobs else np.zeros(0, np.uint8)
        col = pa.LargeBinaryArray.from_buffers(pa.large_binary(), len(blobs), [None, pa.py_buffer(offs), pa.py_buffer(data)])
        return pa.record_batch([ids, col, pa.array(counts, pa.uint32())], schema=schema)
    flat = np.concatenate([r[1] for r in rows]) if rows else np.zeros(0, np.int16)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return pa.record_batch([ids, pa.LargeListArray.from_arrays(pa.array(offs, pa.int64()), pa.array(flat, pa.int16())),
                            pa.array(counts, pa.uint32())], schema=schema)


def _ms(t):
    if isinstance(t, _dt.datetime):
        return int(t.replace(tzinfo=t.tzinfo or _dt.timezone.utc).timestamp() * 1000)
    return int(t)


def _run_info_table(pa, meta, run_infos: List[dict]) -> bytes:
    ts, s, m = pa.timestamp("ms", "UTC"), pa.utf8(), pa.map_(pa.utf8(), pa.utf8())
    types = {"acquisition_start_time": ts, "protocol_start_time": ts, "adc_max": pa.int16(), "adc_min": pa.int16(),
             "sample_rate": pa.uint16(), "context_tags": m, "tracking_id": m}
    schema = pa.schema([pa.field(n, types.get(n, s)) for n in RUN_INFO_FIELDS], metadata=meta)
    cols = []
    for n in RUN_INFO_FIELDS:
        t = types.get(n, s)
        vals = [ri[n] for ri in run_infos]
        if t == ts:
            vals = [_ms(v) for v in vals]
        elif t == m:
            vals = [sorted(dict(v).items()) for v in vals]
        cols.append(pa.array(vals, t))
    return _ipc_bytes(pa, schema, [pa.record_batch(cols, schema=schema)])


def _reads_table(pa, meta, reads, rows_of, run_ids: List[str], pore_types: List[str]) -> bytes:
    f32, d = pa.float32(), lambda: pa.dictionary(pa.int16(), pa.utf8())
    fields = [_uuid_field(pa), pa.field("signal", pa.list_(pa.uint64())), pa.field("read_number", pa.uint32()),
              pa.field("start", pa.uint64()), pa.field("median_before", f32), pa.field("num_minknow_events", pa.uint64()),
              pa.field("tracked_scaling_scale", f32), pa.field("tracked_scaling_shift", f32),
              pa.field("predicted_scaling_scale", f32), pa.field("predicted_scaling_shift", f32),
              pa.field("num_reads_since_mux_change", pa.uint32()), pa.field("time_since_mux_change", f32),
              pa.field("num_samples", pa.uint64()), pa.field("channel", pa.uint16()), pa.field("well", pa.uint8()),
              pa.field("pore_type", d()), pa.field("calibration_offset", f32), pa.field("calibration_scale", f32),
              pa.field("end_reason", d()), pa.field("end_reason_forced", pa.bool_()), pa.field("run_info", d())]
    schema = pa.schema(fields, metadata=meta)
    pore_dict, end_dict, run_dict = pa.array(pore_types, pa.utf8()), pa.array(END_REASONS, pa.utf8()), pa.array(run_ids, pa.utf8())
    nan = float("nan")
    batches = []
    for lo in range(0, len(reads), READ_BATCH_ROWS):
        part, rows = reads[lo: lo + READ_BATCH_ROWS], rows_of[lo: lo + READ_BATCH_ROWS]
        n = len(part)
        col = lambda key, t: pa.array([r[key] for r in part], t)
        const = lambda v, t: pa.array([v] * n, t)
        dic = lambda idx, dictionary: pa.DictionaryArray.from_arrays(pa.array(idx, pa.int16()), dictionary)
        batches.append(pa.record_batch([
            pa.array([r["read_id"].bytes for r in part], pa.binary(16)), pa.array(rows, pa.list_(pa.uint64())),
            col("read_number", pa.uint32()), col("start_sample", pa.uint64()), col("median_before", f32),
            const(0, pa.uint64()), const(nan, f32), const(nan, f32), const(nan, f32), const(nan, f32),
            const(0, pa.uint32()), const(0.0, f32), col("num_samples", pa.uint64()),
            col("channel", pa.uint16()), col("well", pa.uint8()), dic([pore_types.index(r["pore_type"]) for r in part], pore_dict),
            col("calibration_offset", f32), col("calibration_scale", f32),
            dic([END_REASONS.index(r["end_reason"]) for r in part], end_dict), col("end_reason_forced", pa.bool_()),
            dic([run_ids.index(r["run_info"]["acquisition_id"]) for r in part], run_dict)], schema=schema))
    return _ipc_bytes(pa, schema, batches)


# ------------------------------------------------------------------------------------------------ file
class Pod5FileWriter:
    """Streaming writer: the signal table is the first embedded file and grows batch by batch while reads arrive; the
    per-read rows (about 100 B each) are kept until close(), which writes the run-info and reads tables and the footer.
    Memory stays bounded by one signal batch, unlike the reference path (all reads in RAM until the end,
    inference.py:72-79).

    reads: dicts with read_id (uuid.UUID), signal (int16 array), read_number, start_sample, median_before, channel, well,
    pore_type, calibration_offset, calibration_scale, end_reason (a name from END_REASONS), end_reason_forced, run_info
    (dict with RUN_INFO_FIELDS; reads may share one)."""

    def __init__(self, path: str, file_identifier: uuid.UUID = None, section_marker: bytes = None,
                 signal_compression: str = None):
        import os
        pa = self.pa = _pa()
        signal_compression = signal_compression or os.environ.get("S2S_POD5_SIGNAL", "vbz")
        if signal_compression not in ("vbz", "none"):
            raise ValueError("POD5 signal_compression must be 'vbz' or 'none'")
        self.vbz = signal_compression == "vbz"
        self.file_identifier = file_identifier or uuid.uuid4()
        self.marker = section_marker or uuid.uuid4().bytes
        self.meta = {"MINKNOW:file_identifier": str(self.file_identifier), "MINKNOW:software": SOFTWARE,
                     "MINKNOW:pod5_version": POD5_VERSION}
        self.f = open(path, "xb")                     # like pod5.Writer: refuses to overwrite
        self.f.write(SIGNATURE + self.marker)
        self.entries = []
        self._sig_start = self.f.tell()
        self._sig_schema = _signal_schema(pa, self.meta, self.vbz)
        self._sig_writer = pa.ipc.new_file(_SubFile(self.f), self._sig_schema)
        self._pending, self._n_rows = [], 0           # signal rows not yet written; rows written + pending
        self._reads, self._rows_of = [], []
        self._run_infos, self._run_ids = [], []
        self.closed = False

    def add_reads(self, reads: Sequence[dict]) -> None:
        """A read carries its samples as `signal` (int16), or -- from the GPU codec path -- as `vbz_rows`: the finished VBZ blob
        and sample count of each of its signal-table rows (SIGNAL_CHUNK samples per row) plus `num_samples`."""
        for r in reads:
            rows = []
            if r.get("vbz_rows") is not None:
                if not self.vbz:
                    raise ValueError("precompressed rows need signal_compression='vbz'")
                n_samples = int(r["num_samples"])
                for blob, count in r["vbz_rows"]:
                    rows.append(self._n_rows)
                    self._pending.append((r["read_id"].bytes, None, blob, int(count)))
                    self._n_rows += 1
            else:
                raw = np.ascontiguousarray(r["signal"], dtype=np.int16)
                n_samples = len(raw)
                for lo in range(0, max(len(raw), 1), SIGNAL_CHUNK):
                    rows.append(self._n_rows)
                    piece = raw[lo: lo + SIGNAL_CHUNK]
                    self._pending.append((r["read_id"].bytes, piece, None, len(piece)))
                    self._n_rows += 1
            self._rows_of.append(rows)
            if r["run_info"]["acquisition_id"] not in self._run_ids:
                self._run_ids.append(r["run_info"]["acquisition_id"])
                self._run_infos.append(r["run_info"])
            self._reads.append({k: v for k, v in r.items() if k not in ("signal", "vbz_rows")} | {"num_samples": n_samples})
        while len(self._pending) >= SIGNAL_BATCH_ROWS:
            self._sig_writer.write_batch(_signal_batch(self.pa, self._sig_schema, self._pending[:SIGNAL_BATCH_ROWS], self.vbz))
            del self._pending[:SIGNAL_BATCH_ROWS]

    def _end_embedded(self, start, content_type):
        n = self.f.tell() - start
        self.entries.append((start, n, content_type))
        self.f.write(bytes(-n % 8) + self.marker)

    def close(self) -> None:
        if self.closed:
            return
        self.closed = True
        pa = self.pa
        if self._pending:
            self._sig_writer.write_batch(_signal_batch(pa, self._sig_schema, self._pending, self.vbz))
            self._pending = []
        self._sig_writer.close()
        self._end_embedded(self._sig_start, CT_SIGNAL)
        pore_types = sorted({r["pore_type"] for r in self._reads})
        for ct, data in ((CT_RUN_INFO, _run_info_table(pa, self.meta, self._run_infos)),
                         (CT_READS, _reads_table(pa, self.meta, self._reads, self._rows_of, self._run_ids, pore_types))):
            start = self.f.tell()
            self.f.write(data)
            self._end_embedded(start, ct)
        fb = build_footer(str(self.file_identifier), SOFTWARE, POD5_VERSION, self.entries)
        self.f.write(FOOTER_MAGIC + fb + struct.pack("<q", len(fb)) + self.marker + SIGNATURE)
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def write_pod5(path: str, reads: List[dict], file_identifier: uuid.UUID = None, section_marker: bytes = None,
               signal_compression: str = None) -> None:
    with Pod5FileWriter(path, file_identifier, section_marker, signal_compression) as w:
        w.add_reads(reads)


def _signal_rows(sig) -> List[np.ndarray]:
    """int16 samples of every signal-table row: VBZ rows decoded, uncompressed rows as numpy views of the Arrow buffers."""
    pa = _pa()
    rows = []
    col, counts = sig.column("signal"), sig.column("samples").to_numpy()
    if pa.types.is_large_binary(col.type):
        from .codecs import vbz_decompress
        i = 0
        for ch in col.chunks:
            for v in ch:
                rows.append(vbz_decompress(v.as_buffer(), int(counts[i])))
                i += 1
        return rows
    for ch in col.chunks:
        offs = ch.offsets.to_numpy()
        vals = ch.values.to_numpy(zero_copy_only=False)
        rows.extend(vals[offs[i]:offs[i + 1]] for i in range(len(ch)))
    return rows


def read_pod5(path: str) -> dict:
    """Reader of the files written above (tests / round trip): -> dict(footer, run_info rows, reads with their signal)."""
    pa = _pa()
    data = open(path, "rb").read()
    if data[:8] != SIGNATURE or data[-8:] != SIGNATURE:
        raise ValueError("not a POD5 file: bad signature")
    marker = data[8:24]
    if data[-24:-8] != marker:
        raise ValueError("section markers differ")
    flen = struct.unpack_from("<q", data, len(data) - 32)[0]
    fstart = len(data) - 32 - flen
    if data[fstart - 8: fstart] != FOOTER_MAGIC:
        raise ValueError("footer magic not found")
    footer = parse_footer(data[fstart: fstart + flen])
    tabs = {}
    for e in footer["contents"]:
        end = e["offset"] + e["length"]
        if data[end + (-e["length"] % 8): end + (-e["length"] % 8) + 16] != marker:
            raise ValueError("embedded file is not followed by the section marker")
        tabs[e["content_type"]] = pa.ipc.open_file(io.BytesIO(data[e["offset"]: end])).read_all()
    sig, reads_t, run_t = tabs[CT_SIGNAL], tabs[CT_READS], tabs[CT_RUN_INFO]
    sig_rows = _signal_rows(sig)
    reads = []
    for row in reads_t.to_pylist():
        raw = np.concatenate([sig_rows[i] for i in row["signal"]]) if row["signal"] else np.zeros(0, np.int16)
        row = dict(row, read_id=uuid.UUID(bytes=row["read_id"]), signal=raw)
        reads.append(row)
    return {"footer": footer, "schema_metadata": {k.decode(): v.decode() for k, v in reads_t.schema.metadata.items()},
            "run_info": run_t.to_pylist(), "reads": reads, "signal_rows": sig.num_rows}


def iter_pod5(path: str, decode: bool = True):
    """read_pod5() for files that do not fit in memory: the file is memory-mapped, the reads table (about 100 B per read) is read
    whole, and the signal table is walked one record batch (SIGNAL_BATCH_ROWS rows) at a time -- yields (row dict without
    `signal`, int16 samples) per read in file order; same container checks as read_pod5.  decode=False (VBZ files only) yields
    the read's signal-table rows as they are stored instead of its samples: [(VBZ blob, sample count), ...]; the row dict then
    carries the run-info record of the read as `run_info_record`."""
    pa = _pa()
    src = pa.memory_map(path, "r")
    buf = src.read_buffer()
    size = buf.size
    head, tail = buf.slice(0, 24).to_pybytes(), buf.slice(size - 32, 32).to_pybytes()
    if head[:8] != SIGNATURE or tail[-8:] != SIGNATURE:
        raise ValueError("not a POD5 file: bad signature")
    marker = head[8:24]
    if tail[8:24] != marker:
        raise ValueError("section markers differ")
    flen = struct.unpack_from("<q", tail, 0)[0]
    fstart = size - 32 - flen
    if buf.slice(fstart - 8, 8).to_pybytes() != FOOTER_MAGIC:
        raise ValueError("footer magic not found")
    footer = parse_footer(buf.slice(fstart, flen).to_pybytes())
    files = {}
    for e in footer["contents"]:
        end = e["offset"] + e["length"]
        if buf.slice(end + (-e["length"] % 8), 16).to_pybytes() != marker:
            raise ValueError("embedded file is not followed by the section marker")
        files[e["content_type"]] = pa.ipc.open_file(pa.BufferReader(buf.slice(e["offset"], e["length"])))
    sig = files[CT_SIGNAL]
    vbz = pa.types.is_large_binary(sig.schema.field("signal").type.storage_type
                                   if isinstance(sig.schema.field("signal").type, pa.ExtensionType) else sig.schema.field("signal").type)
    cache = {"i": -1}

    def row(i):
        b, j = divmod(i, SIGNAL_BATCH_ROWS)
        if cache["i"] != b:
            rb = sig.get_batch(b)
            col = rb.column(rb.schema.get_field_index("signal"))
            if isinstance(col, pa.ExtensionArray):
                col = col.storage
            cache.update(i=b, col=col, counts=rb.column(rb.schema.get_field_index("samples")).to_numpy())
        col, n = cache["col"], int(cache["counts"][j])
        if not decode:
            return (col[j].as_buffer().to_pybytes(), n)
        if vbz:
            from .codecs import vbz_decompress
            return vbz_decompress(col[j].as_buffer(), n)
        return np.asarray(col[j].values.to_numpy(zero_copy_only=False), dtype=np.int16)
    reads_t = files[CT_READS]
    if not decode:
        if not vbz:
            raise ValueError(f"{path}: decode=False needs a VBZ-compressed signal table")
        runs = {ri["acquisition_id"]: ri for ri in files[CT_RUN_INFO].read_all().to_pylist()}
    for b in range(reads_t.num_record_batches):
        for r in reads_t.get_batch(b).to_pylist():
            if not decode:
                yield dict(r, read_id=uuid.UUID(bytes=r["read_id"]), run_info_record=runs[r["run_info"]]), [row(i) for i in r["signal"]]
                continue
            raw = np.concatenate([row(i) for i in r["signal"]]) if r["signal"] else np.zeros(0, np.int16)
            yield dict(r, read_id=uuid.UUID(bytes=r["read_id"])), raw


def merge_pod5(paths: Sequence[str], out: str) -> int:
    """Concatenate POD5 files written by this package (the out.rankN.pod5 shards of a multi-process run) into one file: reads in
    the order given, their VBZ signal rows copied as stored (nothing is decoded), run-info records united by acquisition id, one
    reads table.  Streams shard by shard; memory = one signal batch + 100 B per read.  -> number of reads."""
    n = 0
    with Pod5FileWriter(out, signal_compression="vbz") as w:
        for p_ in paths:
            batch = []
            for r, rows in iter_pod5(p_, decode=False):
                ri = dict(r["run_info_record"])
                ri["context_tags"], ri["tracking_id"] = dict(ri["context_tags"]), dict(ri["tracking_id"])
                batch.append(dict(read_id=r["read_id"], vbz_rows=rows, num_samples=r["num_samples"], read_number=r["read_number"],
                                  start_sample=r["start"], median_before=r["median_before"], channel=r["channel"], well=r["well"],
                                  pore_type=r["pore_type"], calibration_offset=r["calibration_offset"],
                                  calibration_scale=r["calibration_scale"], end_reason=r["end_reason"],
                                  end_reason_forced=r["end_reason_forced"], run_info=ri))
                n += 1
                if len(batch) >= 256:
                    w.add_reads(batch)
                    batch = []
            w.add_reads(batch)
    return n
