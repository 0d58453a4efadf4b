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
from typing import List, Sequence

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


READ_COLUMNS = (("read_number", "uint32"), ("start", "uint64"), ("median_before", "float32"), ("num_samples", "uint64"),
                ("channel", "uint16"), ("well", "uint8"), ("calibration_offset", "float32"), ("calibration_scale", "float32"),
                ("end_reason_forced", "bool"))


def _read_columns(reads, rows_of, run_ids: List[str], pore_types: List[str]) -> dict:
    """The per-read dicts of Pod5FileWriter as the columns _reads_table writes (numpy arrays, n reads)."""
    src = {"start": "start_sample"}
    cols = {name: np.array([r[src.get(name, name)] for r in reads], dtype=dt) for name, dt in READ_COLUMNS}
    cols["read_id"] = np.frombuffer(b"".join(r["read_id"].bytes for r in reads), dtype=np.uint8).reshape(-1, 16)
    cols["signal_offsets"] = np.concatenate([[0], np.cumsum([len(x) for x in rows_of])]).astype(np.int32)
    cols["signal_rows"] = np.fromiter((i for x in rows_of for i in x), dtype=np.uint64, count=int(cols["signal_offsets"][-1]))
    cols["pore_type"] = np.array([pore_types.index(r["pore_type"]) for r in reads], dtype=np.int16)
    cols["end_reason"] = np.array([END_REASONS.index(r["end_reason"]) for r in reads], dtype=np.int16)
    cols["run_info"] = np.array([run_ids.index(r["run_info"]["acquisition_id"]) for r in reads], dtype=np.int16)
    return cols


def _reads_table(pa, meta, cols: dict, run_ids: List[str], pore_types: List[str]) -> bytes:
    """The reads table (spec v3) from columns: read_id [n][16] u8, signal_offsets [n+1] i32 + signal_rows (the signal-table rows of
    every read, back to back), READ_COLUMNS, and the dictionary indices pore_type / end_reason / run_info (int16).  Every batch is
    assembled from slices of these arrays: no per-read Python object (the writer and merge_pod5 both come through here, so their
    files agree byte for byte)."""
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
    n_reads = len(cols["read_number"])
    ids, sig_offs, sig_rows = np.ascontiguousarray(cols["read_id"], dtype=np.uint8), cols["signal_offsets"], cols["signal_rows"]
    batches = []
    for lo in range(0, n_reads, READ_BATCH_ROWS):
        hi = min(lo + READ_BATCH_ROWS, n_reads)
        n = hi - lo
        col = lambda key, t: pa.array(np.ascontiguousarray(cols[key][lo:hi]), t)
        const = lambda v, dt, t: pa.array(np.full(n, v, dtype=dt), t)
        dic = lambda key, dictionary: pa.DictionaryArray.from_arrays(pa.array(np.ascontiguousarray(cols[key][lo:hi]), pa.int16()), dictionary)
        rows = pa.ListArray.from_arrays(pa.array((sig_offs[lo:hi + 1] - sig_offs[lo]).astype(np.int32), pa.int32()),
                                        pa.array(np.ascontiguousarray(sig_rows[sig_offs[lo]:sig_offs[hi]], dtype=np.uint64), pa.uint64()))
        batches.append(pa.record_batch([
            pa.FixedSizeBinaryArray.from_buffers(pa.binary(16), n, [None, pa.py_buffer(ids[lo:hi].tobytes())]), rows,
            col("read_number", pa.uint32()), col("start", pa.uint64()), col("median_before", f32),
            const(0, np.uint64, pa.uint64()), const(np.nan, np.float32, f32), const(np.nan, np.float32, f32),
            const(np.nan, np.float32, f32), const(np.nan, np.float32, f32),
            const(0, np.uint32, pa.uint32()), const(0.0, np.float32, f32), col("num_samples", pa.uint64()),
            col("channel", pa.uint16()), col("well", pa.uint8()), dic("pore_type", pore_dict),
            col("calibration_offset", f32), col("calibration_scale", f32),
            dic("end_reason", end_dict), col("end_reason_forced", pa.bool_()), dic("run_info", run_dict)], schema=schema))
    return _ipc_bytes(pa, schema, batches)


# ------------------------------------------------------------------------------------------------ file
def _write_tail(f, pa, meta, marker: bytes, file_identifier, sig_start: int, run_infos: List[dict], cols: dict,
                run_ids: List[str], pore_types: List[str]) -> None:
    """Everything behind the signal table (f stands at its end): pad + marker, the run-info and reads tables, the footer."""
    entries = []

    def end_embedded(start, content_type):
        n = f.tell() - start
        entries.append((start, n, content_type))
        f.write(bytes(-n % 8) + marker)
    end_embedded(sig_start, CT_SIGNAL)
    for ct, data in ((CT_RUN_INFO, _run_info_table(pa, meta, run_infos)),
                     (CT_READS, _reads_table(pa, meta, cols, run_ids, pore_types))):
        start = f.tell()
        f.write(data)
        end_embedded(start, ct)
    fb = build_footer(str(file_identifier), SOFTWARE, POD5_VERSION, entries)
    f.write(FOOTER_MAGIC + fb + struct.pack("<q", len(fb)) + marker + SIGNATURE)


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

    def close(self) -> None:
        if self.closed:
            return
        self.closed = True
        pa = self.pa
        if self._pending:
            self._sig_writer.write_batch(_signal_batch(pa, self._sig_schema, self._pending, self.vbz))
            self._pending = []
        self._sig_writer.close()
        pore_types = sorted({r["pore_type"] for r in self._reads})
        cols = _read_columns(self._reads, self._rows_of, self._run_ids, pore_types)
        _write_tail(self.f, pa, self.meta, self.marker, self.file_identifier, self._sig_start, self._run_infos, cols,
                    self._run_ids, pore_types)
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


# ------------------------------------------------------------------------------------------------ shard merge
_BLOCK = np.dtype([("offset", "<i8"), ("meta", "<i4"), ("pad", "<i4"), ("body", "<i8")])   # Arrow File.fbs: struct Block


def _fb_fields(fb, table: int) -> List[int]:
    """Absolute positions of a flatbuffer table's fields (0 = absent)."""
    vt = table - struct.unpack_from("<i", fb, table)[0]
    vsize = struct.unpack_from("<H", fb, vt)[0]
    return [table + o if o else 0 for o in struct.unpack_from("<%dH" % ((vsize - 4) // 2), fb, vt + 4)]


def _fb_follow(fb, pos: int) -> int:
    return pos + struct.unpack_from("<I", fb, pos)[0]


def _arrow_blocks(buf, start: int, length: int):
    """The record-batch blocks of the Arrow IPC file at buf[start : start+length] -> (structured array of _BLOCK: offsets from
    `start`; position of the blocks inside the footer flatbuffer; position of the footer flatbuffer from `start`)."""
    end = start + length
    if bytes(buf[start:start + 6]) != b"ARROW1" or bytes(buf[end - 6:end]) != b"ARROW1":
        raise ValueError("embedded table is not an Arrow IPC file")
    flen = struct.unpack_from("<i", buf, end - 10)[0]
    fpos = end - 10 - flen
    fb = bytes(buf[fpos:fpos + flen])
    f = _fb_fields(fb, struct.unpack_from("<I", fb, 0)[0])
    if len(f) < 4 or not f[3]:
        return np.zeros(0, _BLOCK), 0, fpos - start
    vec = _fb_follow(fb, f[3])
    n = struct.unpack_from("<I", fb, vec)[0]
    return np.frombuffer(fb, _BLOCK, n, vec + 4), vec + 4, fpos - start


def _batch_message(buf, pos: int) -> dict:
    """Where the numbers of one encapsulated RecordBatch message sit (Message.fbs): positions are relative to `pos`, the
    message's continuation marker.  nodes / buffers: (position of the first struct, count); every struct is two int64."""
    cont, mlen = struct.unpack_from("<Ii", buf, pos)
    if cont != 0xFFFFFFFF:
        raise ValueError("Arrow message without continuation marker (a pre-0.15 stream?)")
    fb = bytes(buf[pos + 8: pos + 8 + mlen])
    f = _fb_fields(fb, struct.unpack_from("<I", fb, 0)[0])
    if fb[f[1]] != 3:
        raise ValueError("not a RecordBatch message")
    h = _fb_fields(fb, _fb_follow(fb, f[2]))
    nodes, buffers = _fb_follow(fb, h[1]), _fb_follow(fb, h[2])
    return {"meta": 8 + mlen, "body_len": 8 + f[3], "length": 8 + h[0],
            "nodes": (8 + nodes + 4, struct.unpack_from("<I", fb, nodes)[0]),
            "buffers": (8 + buffers + 4, struct.unpack_from("<I", fb, buffers)[0]),
            "compressed": len(h) > 3 and bool(h[3])}


def _pad8(n):
    return (n + 7) & ~7


class _Shard:
    """One shard file, memory-mapped: container checks as in iter_pod5, then the byte position of every signal-table row."""

    def __init__(self, path: str):
        import mmap
        import os
        self.path = path
        self.fd = os.open(path, os.O_RDONLY)
        size = os.fstat(self.fd).st_size
        self.buf = buf = mmap.mmap(self.fd, 0, access=mmap.ACCESS_READ)
        if buf[:8] != SIGNATURE or buf[size - 8:size] != SIGNATURE:
            raise ValueError(f"{path}: not a POD5 file: bad signature")
        self.marker = buf[8:24]
        if buf[size - 24:size - 8] != self.marker:
            raise ValueError(f"{path}: section markers differ")
        flen = struct.unpack_from("<q", buf, size - 32)[0]
        fstart = size - 32 - flen
        if buf[fstart - 8:fstart] != FOOTER_MAGIC:
            raise ValueError(f"{path}: footer magic not found")
        self.footer = parse_footer(buf[fstart:fstart + flen])
        self.at = {}
        for e in self.footer["contents"]:
            end = e["offset"] + e["length"]
            if buf[end + (-e["length"] % 8): end + (-e["length"] % 8) + 16] != self.marker:
                raise ValueError(f"{path}: embedded file is not followed by the section marker")
            self.at[e["content_type"]] = (e["offset"], e["length"])

    def table(self, content_type):
        pa = _pa()
        o, n = self.at[content_type]
        return pa.ipc.open_file(pa.BufferReader(pa.py_buffer(self.buf).slice(o, n)))

    def signal_rows(self, vbz: bool, skip_batches: int = 0):
        """-> (ids [n][16] u8, samples [n] u32, lens [n] i64 in bytes (VBZ) or int16 samples (uncompressed), pos [n] i64: file offset
        of each row's data) of the rows behind the first skip_batches record batches (the live join has copied those and may
        have punched them out of the file)."""
        buf = self.buf
        start, length = self.at[CT_SIGNAL]
        blocks, _, _ = _arrow_blocks(buf, start, length)
        ids, counts, lens, pos = [], [], [], []
        width = 1 if vbz else 2
        for b in blocks[skip_batches:]:
            at = start + int(b["offset"])
            m = _batch_message(buf, at)
            if m["compressed"]:
                raise ValueError(f"{self.path}: Arrow buffer compression in the signal table is not something this writer produces")
            k = struct.unpack_from("<q", buf, at + m["length"])[0]
            bufs = np.frombuffer(buf, "<i8", 2 * m["buffers"][1], at + m["buffers"][0]).reshape(-1, 2)
            if len(bufs) != (7 if vbz else 8):
                raise ValueError(f"{self.path}: unexpected signal-table layout ({len(bufs)} buffers)")
            body = at + int(b["meta"])
            ids.append(np.frombuffer(buf, np.uint8, 16 * k, body + int(bufs[1, 0])).reshape(k, 16))
            offs = np.frombuffer(buf, "<i8", k + 1, body + int(bufs[3, 0]))
            lens.append(np.diff(offs))
            pos.append(body + int(bufs[4 if vbz else 5, 0]) + width * offs[:-1])
            counts.append(np.frombuffer(buf, "<u4", k, body + int(bufs[-1, 0])))
        cat = lambda xs, dt, shape: np.concatenate(xs) if xs else np.zeros(shape, dt)
        return (cat(ids, np.uint8, (0, 16)), cat(counts, np.uint32, 0), cat(lens, np.int64, 0), cat(pos, np.int64, 0))

    def close(self):
        import os
        self.buf = None        # (views handed out keep the map alive until they go)
        if self.fd >= 0:
            os.close(self.fd)
            self.fd = -1


def _signal_template(pa, schema, vbz: bool, k: int):
    """One k-row signal batch as pyarrow writes it, with one byte (one sample) of data per row: -> (file head = magic + schema
    message, message bytes, _batch_message of it, its buffer table, tail = end-of-stream marker + footer + length + magic with ONE
    block)."""
    rows = [(bytes(16), None if vbz else np.zeros(1, np.int16), b"\0" if vbz else None, 1) for _ in range(k)]
    data = _ipc_bytes(pa, schema, [_signal_batch(pa, schema, rows, vbz)])
    blocks, _, fpos = _arrow_blocks(data, 0, len(data))
    at, end = int(blocks[0]["offset"]), int(blocks[0]["offset"] + blocks[0]["meta"] + blocks[0]["body"])
    msg = data[at:end]
    m = _batch_message(msg, 0)
    return data[:at], msg, m, np.frombuffer(msg, "<i8", 2 * m["buffers"][1], m["buffers"][0]).reshape(-1, 2).copy()


def _arrow_tail(pa, schema, blocks: np.ndarray) -> bytes:
    """End-of-stream marker + file footer + footer length + magic of an Arrow IPC file of `schema` whose record batches sit at
    `blocks`: pyarrow writes the footer (for that many empty batches), the Block structs are then overwritten in place."""
    import io
    sink = io.BytesIO()
    empty = pa.RecordBatch.from_arrays([pa.array([], f.type) for f in schema], schema=schema)
    with pa.ipc.new_file(sink, schema) as w:
        for _ in range(len(blocks)):
            w.write_batch(empty)
    data = bytearray(sink.getvalue())
    mine, at, fpos = _arrow_blocks(data, 0, len(data))
    assert len(mine) == len(blocks)
    tail_from = int(mine[-1]["offset"] + mine[-1]["meta"] + mine[-1]["body"]) if len(mine) else fpos - 8
    if len(blocks):
        np.frombuffer(data, _BLOCK, len(blocks), fpos + at)[:] = blocks
    return bytes(data[tail_from:])


def merge_pod5(paths: Sequence[str], out: str, threads: int = None, take_first: bool = False, file_identifier=None,
               section_marker: bytes = None, consume: bool = False, _live: dict = None) -> int:
    """POD5 shards written by this package (the out.rankN.pod5 files of a multi-process run) -> one file with the reads in the order
    given.  The signal table is re-batched WITHOUT touching a sample: every output batch of SIGNAL_BATCH_ROWS rows is a patched
    copy of pyarrow's own message metadata, the 16-byte ids / offsets / sample counts of its rows (3 KB, from the shards' memory
    maps) and the rows' stored bytes, which are consecutive in their shard and move by copy_file_range (merge.copy_ranges: one
    writer, front to back).  Only the reads table (signal-row indices shifted by the rows of the earlier shards, dictionaries united;
    columnar, no per-read object), the run-info table and the two footers are built anew.  The result is byte for byte what
    Pod5FileWriter writes when it is handed all the reads in turn.  take_first: the first shard BECOMES the output (its full signal
    batches stay where they are); consume: every other shard is deleted once its rows are in.  -> number of reads;
    merge_pod5.last holds bytes / seconds.
    _live (merge.LiveJoin): the output is open already (`fd`) and holds shard 0's head and the full batches the live join copied
    while the ranks ran -- `blocks` (their Arrow blocks), `placed[r]` (the output batch of each of shard r's copied batches) --
    so only every shard's rows behind its copied batches are laid out here, behind those batches, and the tables are built with
    the row indices that follow from both."""
    import os
    import time
    from . import merge as M
    t0 = time.perf_counter()
    pa = _pa()
    threads = threads or M.merge_threads()
    shards = [_Shard(p_) for p_ in paths]
    try:
        sig_schemas = [s_.table(CT_SIGNAL).schema for s_ in shards]
        st = sig_schemas[0].field("signal").type
        vbz = pa.types.is_large_binary(st.storage_type if isinstance(st, pa.ExtensionType) else st)
        for p_, sc in zip(paths[1:], sig_schemas[1:]):
            if sc.remove_metadata() != sig_schemas[0].remove_metadata():
                raise ValueError(f"{p_}: signal table schema differs from {paths[0]} (VBZ and uncompressed shards do not mix)")
        if take_first or _live is not None:                # (the live join started the output with shard 0's head)
            file_identifier, section_marker = shards[0].footer["file_identifier"], bytes(shards[0].marker)
        file_identifier = file_identifier or uuid.uuid4()
        marker = section_marker or uuid.uuid4().bytes
        meta = {"MINKNOW:file_identifier": str(file_identifier), "MINKNOW:software": SOFTWARE, "MINKNOW:pod5_version": POD5_VERSION}
        schema = _signal_schema(pa, meta, vbz)
        width = 1 if vbz else 2

        # ---- every signal row of every shard, in output order
        R = SIGNAL_BATCH_ROWS
        live_blocks = np.zeros(0, _BLOCK)
        placed = [np.zeros(0, np.int64)] * len(shards)
        if _live is not None:
            placed = [np.asarray(x, dtype=np.int64) for x in _live["placed"]]
            live_blocks = np.asarray(_live["blocks"], dtype=_BLOCK)
        per = [s_.signal_rows(vbz, skip_batches=len(k_)) for s_, k_ in zip(shards, placed)]
        n_rows = [len(p_[1]) for p_ in per]
        rows_in_shard = [R * len(k_) + n_ for k_, n_ in zip(placed, n_rows)]
        # where every signal row of shard r ends up in the output: behind the rows of the earlier shards -- or, live, in the batch
        # its own batch was copied to, and for the rows behind those batches: behind all copied batches, shard after shard
        row_maps, base = [], R * len(live_blocks)
        for r in range(len(shards)):
            tail_map = np.arange(n_rows[r], dtype=np.int64) + base
            base += n_rows[r]
            if _live is None:
                row_maps.append(tail_map)
            else:
                local = np.arange(R * len(placed[r]), dtype=np.int64)
                row_maps.append(np.concatenate([R * placed[r][local // R] + local % R, tail_map]))
            assert len(row_maps[-1]) == rows_in_shard[r]
        ids, counts = np.concatenate([p_[0] for p_ in per]), np.concatenate([p_[1] for p_ in per])
        lens, pos = np.concatenate([p_[2] for p_ in per]), np.concatenate([p_[3] for p_ in per])
        shard_of = np.repeat(np.arange(len(shards)), n_rows)
        n = len(counts)
        n_batches = -(-n // R)
        first_row = np.arange(n_batches, dtype=np.int64) * R
        cum = np.concatenate([[0], np.cumsum(lens)])                       # in bytes (VBZ) or samples
        data_units = cum[np.minimum(first_row + R, n)] - cum[first_row]
        data_bytes = width * data_units

        # ---- layout: head | batch messages | tail
        head, msg_full, m_full, bufs_full = _signal_template(pa, schema, vbz, R)
        k_last = n - (n_batches - 1) * R if n_batches else 0
        tmpl = {R: (msg_full, m_full, bufs_full)}
        if n_batches and k_last != R:
            tmpl[k_last] = _signal_template(pa, schema, vbz, k_last)[1:]
        di = 4 if vbz else 5                                               # the buffer that holds the rows' data
        sig_start = len(SIGNATURE) + 16
        msg_at = np.zeros(n_batches + 1, np.int64)                         # from the start of the embedded file
        body_len, data_at = np.zeros(n_batches, np.int64), np.zeros(n_batches, np.int64)
        for b in range(n_batches):
            msg, m, bufs = tmpl[R if b < n_batches - 1 or k_last == R else k_last]
            grow = _pad8(int(data_bytes[b])) - _pad8(int(bufs[di, 1]))
            body_len[b] = len(msg) - m["meta"] + grow
            data_at[b] = m["meta"] + bufs[di, 0]
            msg_at[b + 1] = m["meta"] + body_len[b]
        msg_base = len(head) if _live is None else int(_live["msg_base"])          # (live: behind the batches copied while the ranks ran)
        if _live is not None and _live["head_len"] != len(head):
            raise ValueError("the shards' signal tables do not start as this writer starts them")
        msg_at = msg_base + np.concatenate([[0], np.cumsum(msg_at[1:])])
        blocks = np.zeros(n_batches, _BLOCK)
        blocks["offset"], blocks["body"] = msg_at[:-1], body_len
        blocks["meta"] = [tmpl[R if b < n_batches - 1 or k_last == R else k_last][1]["meta"] for b in range(n_batches)]
        tail = _arrow_tail(pa, schema, np.concatenate([live_blocks, blocks]))

        # ---- the data runs: rows that follow each other in a shard batch AND in an output batch move as one range
        batch_of = np.arange(n, dtype=np.int64) // R
        dst = sig_start + msg_at[batch_of] + data_at[batch_of] + width * (cum[:-1] - cum[first_row][batch_of]) if n else np.zeros(0, np.int64)
        nbytes = width * lens
        brk = np.ones(n, bool)
        if n > 1:
            brk[1:] = (batch_of[1:] != batch_of[:-1]) | (shard_of[1:] != shard_of[:-1]) | (pos[1:] != pos[:-1] + nbytes[:-1])
        run = np.flatnonzero(brk)
        run_bytes = np.add.reduceat(nbytes, run) if n else np.zeros(0, np.int64)

        # ---- take_first: full batches of shard 0 stay in place (same head, same messages: the layout above IS shard 0's layout)
        keep = 0
        if take_first:
            own = _arrow_blocks(shards[0].buf, *shards[0].at[CT_SIGNAL])[0]
            keep = min(n_rows[0] // R, n_batches)
            if shards[0].at[CT_SIGNAL][0] != sig_start or (keep and (
                    not np.array_equal(own["offset"][:keep], blocks["offset"][:keep]) or
                    not np.array_equal(own["body"][:keep], blocks["body"][:keep]))):
                raise ValueError(f"{paths[0]}: signal table is not laid out as this writer lays it out; merge without take_first")
        moving = batch_of[run] >= keep
        in_place = [(int(shard_of[i]), int(pos[i]), int(dst[i]), int(nb)) for i, nb in zip(run[moving], run_bytes[moving])
                    if take_first and shard_of[i] == 0]
        stash = [(d, bytes(shards[0].buf[p_:p_ + nb])) for _, p_, d, nb in in_place]   # the one partial batch of shard 0: <= R rows
        # (the tables behind shard 0's signal table are read before take_first lets the copies run over them)
        cols, run_ids, run_infos, pore_types = _merged_read_columns(pa, shards, row_maps)
        del per

        # ---- the small parts of every batch that moves -- message metadata + ids + offsets in front of the data, sample counts behind
        # it -- are laid back to back in a memory file, so that they are byte ranges like the rest and ONE writer fills the output
        # front to back (merge.py: a second writer of the same file slows both down)
        parts, part_jobs = bytearray(), []
        for b in range(keep, n_batches):
            lo, hi = b * R, min(b * R + R, n)
            msg, m, bufs = tmpl[hi - lo]
            part = bytearray(msg)
            grow = _pad8(int(data_bytes[b])) - _pad8(int(bufs[di, 1]))
            nb = bufs.copy()
            nb[di, 1] = data_bytes[b]
            nb[di + 1:, 0] += grow
            part[m["buffers"][0]: m["buffers"][0] + nb.size * 8] = nb.tobytes()
            struct.pack_into("<q", part, m["body_len"], int(body_len[b]))
            if not vbz:                                                  # FieldNode of the list's child: its length is the sample count
                struct.pack_into("<q", part, m["nodes"][0] + 16 * 2, int(data_units[b]))
            o = m["meta"]
            part[o + bufs[1, 0]: o + bufs[1, 0] + 16 * (hi - lo)] = ids[lo:hi].tobytes()
            part[o + bufs[3, 0]: o + bufs[3, 0] + 8 * (hi - lo + 1)] = (cum[lo:hi + 1] - cum[lo]).astype("<i8").tobytes()
            front = o + int(bufs[di, 0])
            back = front + _pad8(int(bufs[di, 1]))                       # the template's data (one unit per row) is cut out
            last = o + int(bufs[-1, 0])
            part[last: last + 4 * (hi - lo)] = counts[lo:hi].astype("<u4").tobytes()
            at = sig_start + int(msg_at[b])
            behind = bytes(_pad8(int(data_bytes[b])) - int(data_bytes[b])) + bytes(part[back:])
            part_jobs.append((len(parts), at, front))
            parts += part[:front]
            part_jobs.append((len(parts), at + front + int(data_bytes[b]), len(behind)))
            parts += behind
        end_of_batches = sig_start + int(msg_at[-1])
        part_jobs.append((len(parts), end_of_batches, len(tail)))
        parts += tail
        for d, blob in stash:                                            # (take_first: shard 0's rows of the first batch that moves)
            part_jobs.append((len(parts), d, len(blob)))
            parts += blob

        remover = M._Remover(consume)
        with M._Files() as files:
            if _live is not None:
                fd = _live["fd"]                                         # (the live join's descriptor; it closes it)
            elif take_first:
                os.replace(paths[0], out)
                fd = files.open(out, os.O_RDWR)
            else:
                fd = files.open(out, os.O_RDWR | os.O_CREAT | os.O_EXCL)
                os.pwrite(fd, SIGNATURE + marker + head, 0)
            os.ftruncate(fd, max(end_of_batches + len(tail), os.fstat(fd).st_size))     # (grown at once; cut to its final size below)
            try:
                pfd = os.memfd_create("s2s-merge-parts")
            except (AttributeError, OSError):
                import tempfile
                tf = tempfile.TemporaryFile()
                pfd = os.dup(tf.fileno())
                tf.close()
            files.fds.append(pfd)
            os.pwrite(pfd, bytes(parts), 0)
            # every job, in the order of the output; a shard is closed and (consume) deleted once its last range is in
            jobs = [(pfd, so, fd, d, ln, -1) for so, d, ln in part_jobs]
            jobs += [(shards[int(shard_of[i])].fd, int(pos[i]), fd, int(dst[i]), int(nb), int(shard_of[i]))
                     for i, nb in zip(run[moving], run_bytes[moving]) if not (take_first and shard_of[i] == 0)]
            jobs.sort(key=lambda j_: j_[3])
            last_job_of = {}
            for k_, j_ in enumerate(jobs):
                if j_[5] >= 0:
                    last_job_of[j_[5]] = k_
            cut_after = sorted((k_, s_i) for s_i, k_ in last_job_of.items()) + [(len(jobs) - 1, None)]
            copied, start = 0, 0
            done_shards = set()
            for k_, s_i in cut_after:
                if k_ + 1 > start:
                    copied += M.copy_ranges([j_[:5] for j_ in jobs[start:k_ + 1]], threads)
                    start = k_ + 1
                if s_i is not None and not (take_first and s_i == 0):
                    shards[s_i].close()
                    done_shards.add(s_i)
                    remover.remove(paths[s_i])
            for s_i in range(len(shards)):                               # shards without a row
                if s_i not in done_shards and not (take_first and s_i == 0):
                    shards[s_i].close()
                    remover.remove(paths[s_i])
            # ---- reads + run-info tables, footer
            with os.fdopen(os.dup(fd), "r+b") as f:
                f.seek(end_of_batches + len(tail))
                f.truncate()
                _write_tail(f, pa, meta, marker, file_identifier, sig_start, run_infos, cols, run_ids, pore_types)
        removing = remover.finish()
        merge_pod5.last = {"bytes_copied": int(copied), "seconds": time.perf_counter() - t0, "signal_rows": int(n) + R * len(live_blocks),
                           "batches_in_place": int(keep), "batches_copied_live": len(live_blocks), "remove_seconds": removing, "threads": threads,
                           "engine": "map" if M.merge_engine() else "fd"}
        return len(cols["read_number"])
    finally:
        for s_ in shards:
            try:
                s_.close()
            except (BufferError, OSError):
                pass


merge_pod5.last = {}


def _merged_read_columns(pa, shards, row_maps):
    """The reads tables of the shards as ONE set of columns (for _reads_table): signal-row indices translated by row_maps[r] (where
    signal row i of shard r sits in the output), pore-type and run-info dictionaries united (run infos in order of first
    appearance, by acquisition id)."""
    names = [n for n, _ in READ_COLUMNS]
    parts = {k: [] for k in names + ["read_id", "signal_rows", "pore_type", "end_reason", "run_info"]}
    list_lens = []
    run_ids, run_infos, pore_seen, pending = [], [], set(), []
    for s_, row_map in zip(shards, row_maps):
        t = s_.table(CT_READS).read_all()
        runs = {ri["acquisition_id"]: ri for ri in s_.table(CT_RUN_INFO).read_all().to_pylist()}
        for name, dt in READ_COLUMNS:
            parts[name].append(t.column(name).to_numpy().astype(dt, copy=False) if t.num_rows else np.zeros(0, dt))
        rid = t.column("read_id").combine_chunks()
        rid = rid.storage if isinstance(rid, pa.ExtensionArray) else rid
        parts["read_id"].append(np.frombuffer(rid.buffers()[1], np.uint8, 16 * len(rid), 16 * rid.offset).reshape(-1, 16)
                                if len(rid) else np.zeros((0, 16), np.uint8))
        sig = t.column("signal").combine_chunks()
        offs = sig.offsets.to_numpy()
        list_lens.append(np.diff(offs))
        parts["signal_rows"].append(np.asarray(row_map, dtype=np.uint64)[sig.values.to_numpy()[offs[0]:offs[-1]].astype(np.int64)]
                                    if len(sig) else np.zeros(0, np.uint64))
        for key in ("pore_type", "end_reason", "run_info"):
            c = t.column(key).combine_chunks()
            idx = c.indices.to_numpy().astype(np.int64) if len(c) else np.zeros(0, np.int64)
            values = c.dictionary.to_pylist()
            used = [values[i] for i in np.unique(idx)]
            if key == "run_info":
                for v in values:                                   # dictionary order = order of first appearance in the shard
                    if v in used and v not in run_ids:
                        run_ids.append(v)
                        ri = dict(runs[v])
                        ri["context_tags"], ri["tracking_id"] = dict(ri["context_tags"]), dict(ri["tracking_id"])
                        run_infos.append(ri)
                parts[key].append(np.array([run_ids.index(v) for v in values], np.int16)[idx] if len(idx) else np.zeros(0, np.int16))
            elif key == "pore_type":
                pore_seen.update(used)
                pending.append((values, idx))
            else:
                remap = np.array([END_REASONS.index(v) for v in values], np.int16)
                parts[key].append(remap[idx] if len(idx) else np.zeros(0, np.int16))
    pore_types = sorted(pore_seen)
    for values, idx in pending:
        remap = np.array([pore_types.index(v) if v in pore_seen else -1 for v in values], np.int16)
        parts["pore_type"].append(remap[idx] if len(idx) else np.zeros(0, np.int16))
    cols = {k: (np.concatenate(v) if v else np.zeros(0)) for k, v in parts.items()}
    cols["signal_offsets"] = np.concatenate([[0], np.cumsum(np.concatenate(list_lens))]).astype(np.int32) if list_lens else np.zeros(1, np.int32)
    return cols, run_ids, run_infos, pore_types
