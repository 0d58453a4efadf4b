"""CPU tests of the host side around the hot path: writers, model-object helpers, batching, sharding, CLI."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest
import torch

from seq2squiggle_amd import chunker, parallel, signal_io
from seq2squiggle_amd import utils as U
from conftest import GOLDEN, ROOT, load_npz


def _writer(path, profile_name="dna-r10-prom", ideal=True, preserve=True):
    return signal_io.BLOW5Writer(str(path), U.get_profile(profile_name), ideal, profile_name, preserve)


@pytest.mark.parametrize("tag,profile_name", [("k9", "dna-r10-prom"), ("k6", "dna-r9-min")])
@pytest.mark.parametrize("ext", [".slow5", ".blow5"])
def test_writer_records_match_reference_writer(tmp_path, tag, profile_name, ext):
    """Feed the reference's per-read pA signals to the native writer: header attributes, int16 samples and record
    fields must equal what the reference's BLOW5Writer.save handed to pyslow5 (export_*.npz)."""
    sig = load_npz(f"signals_{tag}.npz")
    ex = load_npz(f"export_{tag}.npz")
    w = _writer(tmp_path / ("o" + ext), profile_name)
    order = [str(x) for x in sig["read_order"]]
    half = len(order) // 2
    for part in (order[:half], order[half:]):                     # two save() calls: append mode
        w.signals = {rid: torch.from_numpy(sig["sig__" + rid]) for rid in part}
        w.save()
    header, recs = (signal_io.read_slow5 if ext == ".slow5" else signal_io.read_blow5)(str(tmp_path / ("o" + ext)))
    attrs = dict(line[1:].split("\t") for line in header.splitlines() if line.startswith("@"))
    for k, v in zip(ex["header_keys"], ex["header_vals"]):
        assert attrs[str(k)] == str(v), k
    assert "exp_start_time" in attrs
    ref = {}
    for key in ex["order"]:
        rid = str(key).split("__", 1)[1]
        ref[rid] = (ex[str(key) + "__raw"], ex[str(key) + "__meta"])
    assert [r["read_id"] for r in recs] == order
    t = 0
    for i, r in enumerate(recs):
        raw, meta = ref[r["read_id"]]
        assert np.array_equal(r["signal"], raw)
        assert (r["digitisation"], r["offset"], r["range"], r["sampling_rate"], r["len_raw_signal"],
                r["median_before"]) == tuple(meta[:6])
        assert r["start_time"] == t and r["read_number"] == i and r["channel_number"] == "0" and r["start_mux"] == 0
        t += len(raw)


def test_writer_ids_rna_and_errors(tmp_path):
    w = _writer(tmp_path / "a.slow5", "rna-004-prom", ideal=False, preserve=False)
    with pytest.raises(ValueError):
        w.save()
    np.random.seed(0)
    w.signals = {"x": torch.tensor([100.0, 101.0, 0.5]), "empty": torch.zeros(0), "y": torch.tensor([50.0])}
    w.save()
    _, recs = signal_io.read_slow5(str(tmp_path / "a.slow5"))
    assert [r["read_id"] for r in recs] == ["00000000-0000-0000-0000-000000000001", "00000000-0000-0000-0000-000000000002"]
    p = U.get_profile("rna-004-prom")
    exp = signal_io.signal_to_dac(np.array([100.0, 101.0, 0.5], np.float32), p["digitisation"], p["range"], p["offset_mean"], True)
    assert np.array_equal(recs[0]["signal"], exp) and recs[0]["offset"] != p["offset_mean"]    # sampled offset, mean in the DAC
    assert signal_io.get_seq_kit_and_flow_cell("dna-r9-min") == ("SQK-LSK109", "FLO-MIN110")
    with pytest.raises(ValueError):
        signal_io.get_seq_kit_and_flow_cell("dna-r7")


@pytest.mark.parametrize("case", [0, 1, 2])
def test_pod5_records_match_reference_writer(tmp_path, case):
    """POD5Writer on the reference's per-read pA signals: every value the reference's POD5Writer.save hands to the pod5
    library (recorded by tools/make_goldens.py pod5 -> pod5_records.npz) comes back out of the written file."""
    import json
    from seq2squiggle_amd import pod5_io
    g = load_npz("pod5_records.npz")
    tag, prof, ideal, preserve = [str(x) for x in g[f"case{case}__args"]]
    sig = load_npz(f"signals_{tag}.npz")
    order = [str(x) for x in sig["read_order"]]
    path = tmp_path / "o.pod5"
    w = signal_io.POD5Writer(path, U.get_profile(prof), ideal == "True", prof, preserve == "True")
    with pytest.raises(ValueError):
        w.save()
    w.signals = {rid: torch.from_numpy(sig["sig__" + rid]) for rid in order}
    w.save()
    with pytest.raises(FileExistsError):             # like pod5.Writer: no appending (inference.py:71-79 exports once)
        w.save()
    d = pod5_io.read_pod5(str(path))
    assert [str(r["read_id"]) for r in d["reads"]] == [str(x) for x in g[f"case{case}__read_ids"]]
    meta = g[f"case{case}__meta"]
    for i, r in enumerate(d["reads"]):
        assert np.array_equal(r["signal"], g[f"case{case}__raw{i}"]) and r["num_samples"] == len(r["signal"])
        off, scale, mb, num, start, channel, well, reason, forced = meta[i]
        assert (r["calibration_offset"], r["calibration_scale"], r["median_before"]) == \
               (np.float32(off), np.float32(scale), np.float32(mb))
        assert (r["read_number"], r["start"], r["channel"], r["well"]) == (num, start, channel, well)
        assert r["end_reason"] == pod5_io.END_REASONS[int(reason)] == str(g[f"case{case}__end_reason"]).lower()
        assert r["end_reason_forced"] == bool(forced) and r["pore_type"] == str(g[f"case{case}__pore_type"])
    ref_ri = json.loads(str(g[f"case{case}__run_info"]))
    (ri,) = d["run_info"]
    for k, v in ref_ri.items():
        if v == "<datetime>":
            assert ri[k] is not None
        elif isinstance(v, dict):
            assert dict(ri[k]) == v
        else:
            assert ri[k] == v, k
    assert set(ri) == set(pod5_io.RUN_INFO_FIELDS)


@pytest.mark.parametrize("compression", ["vbz", "none"])
def test_pod5_container_layout(tmp_path, compression):
    """Signature, section markers, 8-byte padding, footer magic/length, flatbuffer footer (parsed by hand here, field by
    field, independently of pod5_io.parse_footer) and the embedded Arrow files' schema metadata."""
    import io, struct, uuid, datetime
    import pyarrow as pa
    from seq2squiggle_amd import pod5_io as P
    ri = dict(zip(P.RUN_INFO_FIELDS, [""] * len(P.RUN_INFO_FIELDS)))
    ri.update(acquisition_start_time=datetime.datetime(2024, 1, 2, 3, 4, 5), protocol_start_time=0, adc_max=4095, adc_min=-4096,
              sample_rate=5000, context_tags={"k": "v"}, tracking_id={})
    rng = np.random.default_rng(0)
    lens = [10, 2 * P.SIGNAL_CHUNK + 7, 1] + [50] * 250                # > SIGNAL_BATCH_ROWS rows, one multi-row read
    reads = [dict(read_id=uuid.UUID(int=i + 1), signal=rng.integers(-4096, 4096, n).astype(np.int16), read_number=i,
                  start_sample=0, median_before=200.0, channel=1, well=2, pore_type="not_set", calibration_offset=-3.0,
                  calibration_scale=0.5, end_reason="signal_positive", end_reason_forced=False, run_info=ri)
             for i, n in enumerate(lens)]
    path = str(tmp_path / "c.pod5")
    marker, ident = bytes(range(16)), uuid.UUID(int=7)
    P.write_pod5(path, reads, file_identifier=ident, section_marker=marker, signal_compression=compression)
    data = open(path, "rb").read()
    assert data[:8] == data[-8:] == b"\x8bPOD\r\n\x1a\n" and data[8:24] == data[-24:-8] == marker
    flen = struct.unpack_from("<q", data, len(data) - 32)[0]
    fb = data[len(data) - 32 - flen: len(data) - 32]
    assert flen % 8 == 0 and data[len(data) - 32 - flen - 8: len(data) - 32 - flen] == b"FOOTER\x00\x00"
    # flatbuffer by hand: root table -> vtable -> 4 fields
    root = struct.unpack_from("<I", fb, 0)[0]
    vt = root - struct.unpack_from("<i", fb, root)[0]
    vsize, tsize, *offs = struct.unpack_from("<HHHHHH", fb, vt)
    assert vsize == 12 and all(0 < o < tsize for o in offs)

    def fb_string(pos):
        pos += struct.unpack_from("<I", fb, pos)[0]
        n = struct.unpack_from("<I", fb, pos)[0]
        assert fb[pos + 4 + n] == 0 and pos % 4 == 0
        return fb[pos + 4: pos + 4 + n].decode()
    assert [fb_string(root + o) for o in offs[:3]] == [str(ident), P.SOFTWARE, P.POD5_VERSION]
    vec = root + offs[3] + struct.unpack_from("<I", fb, root + offs[3])[0]
    assert struct.unpack_from("<I", fb, vec)[0] == 3
    seen = {}
    for i in range(3):
        t = vec + 4 + 4 * i + struct.unpack_from("<I", fb, vec + 4 + 4 * i)[0]
        tv = t - struct.unpack_from("<i", fb, t)[0]
        _, _, o_off, o_len, o_fmt, o_ct = struct.unpack_from("<HHHHHH", fb, tv)
        assert (t + o_off) % 8 == 0 and (t + o_len) % 8 == 0
        off, ln = struct.unpack_from("<q", fb, t + o_off)[0], struct.unpack_from("<q", fb, t + o_len)[0]
        assert struct.unpack_from("<h", fb, t + o_fmt)[0] == 0
        seen[struct.unpack_from("<h", fb, t + o_ct)[0]] = (off, ln)
        assert off % 8 == 0 and data[off: off + 6] == b"ARROW1" and data[off + ln - 6: off + ln] == b"ARROW1"
        assert data[off + ln + (-ln % 8): off + ln + (-ln % 8) + 16] == marker
    assert set(seen) == {P.CT_READS, P.CT_SIGNAL, P.CT_RUN_INFO}
    sig = pa.ipc.open_file(io.BytesIO(data[seen[P.CT_SIGNAL][0]: sum(seen[P.CT_SIGNAL])]))
    assert sig.schema.metadata[b"MINKNOW:pod5_version"] == P.POD5_VERSION.encode()
    assert sig.schema.field("read_id").metadata[b"ARROW:extension:name"] == b"minknow.uuid"
    if compression == "vbz":          # libpod5's compressed signal column: large_binary tagged minknow.vbz, one zstd frame per row
        assert sig.schema.field("signal").type == pa.large_binary()
        assert sig.schema.field("signal").metadata[b"ARROW:extension:name"] == b"minknow.vbz"
        first = sig.get_batch(0)
        blob, n = first.column(1)[1].as_py(), first.column(2)[1].as_py()          # row 1 = first row of the long read
        assert blob[:4] == b"\x28\xb5\x2f\xfd" and n == P.SIGNAL_CHUNK
        from seq2squiggle_amd import codecs
        assert np.array_equal(codecs.vbz_decompress(blob, n), reads[1]["signal"][:n])
    else:
        assert sig.schema.field("signal").type == pa.large_list(pa.int16())
    sizes = [sig.get_batch(i).num_rows for i in range(sig.num_record_batches)]
    assert all(n == P.SIGNAL_BATCH_ROWS for n in sizes[:-1]) and sum(sizes) == len(lens) + 2
    d = P.read_pod5(path)
    assert [len(r["signal"]) for r in d["reads"]] == lens and d["reads"][1]["signal"].tolist() == reads[1]["signal"].tolist()
    assert d["reads"][1]["signal"].dtype == np.int16 and len(reads_rows := d["reads"][1]) and d["signal_rows"] == len(lens) + 2
    assert d["run_info"][0]["acquisition_start_time"].year == 2024 and d["run_info"][0]["context_tags"] == [("k", "v")]


@pytest.mark.parametrize("rec", ["none", "zlib", "zstd"])
@pytest.mark.parametrize("sig", ["none", "svb-zd"])
def test_blow5_compression_methods(tmp_path, rec, sig):
    """Header method bytes (slow5lib's on-disk codes: record none/zlib/zstd = 0/1/2, signal none/svb-zd = 0/1), record
    framing, and for svb-zd the blob of codecs.svb_zd_compress in place of the raw samples."""
    import struct, zlib
    from seq2squiggle_amd import codecs
    w = signal_io.BLOW5Writer(str(tmp_path / "c.blow5"), U.get_profile("dna-r10-prom"), True, "dna-r10-prom", True,
                              record_compression=rec, signal_compression=sig)
    sigs = [np.array([100, 102, 99, 400, -5], np.int16), np.cumsum(np.random.default_rng(0).integers(-30, 31, 3000)).astype(np.int16)]
    offs = np.array([0, 5, 3005])
    w.save_dac(["a", "b"], np.concatenate(sigs), offs)
    data = open(w.filename, "rb").read()
    assert data[:9] == b"BLOW5\x01\x00\x02\x00" and data[9] == {"none": 0, "zlib": 1, "zstd": 2}[rec]
    assert struct.unpack_from("<I", data, 10)[0] == 1 and data[14] == {"none": 0, "svb-zd": 1}[sig] and not any(data[15:64])
    pos = 68 + struct.unpack_from("<I", data, 64)[0]
    n = struct.unpack_from("<Q", data, pos)[0]
    body = data[pos + 8: pos + 8 + n]
    body = zlib.decompress(body) if rec == "zlib" else codecs.zstd_decompress(body, codecs.zstd_frame_content_size(body)) \
        if rec == "zstd" else body
    assert body[:3] == b"\x01\x00a"
    field = struct.unpack_from("<Q", body, 3 + 4 + 32)[0]
    blob = body[3 + 4 + 32 + 8:]
    if sig == "svb-zd":      # the hand-computed vector of tests/golden/codec_kat.json
        assert field == 13 and blob[:13].hex() == "050000004001c804055a022903"
    else:
        assert field == 5 and blob[:10] == sigs[0].astype("<i2").tobytes()
    _, recs = signal_io.read_blow5(w.filename)
    assert [r["read_id"] for r in recs] == ["a", "b"] and all(np.array_equal(r["signal"], x) for r, x in zip(recs, sigs))
    with pytest.raises(ValueError):
        signal_io.BLOW5Writer(str(tmp_path / "d.blow5"), U.get_profile("dna-r10-prom"), True, "dna-r10-prom", True,
                              record_compression="lzma")


@pytest.mark.parametrize("rec", ["none", "zlib", "zstd"])
def test_native_record_packer_equals_the_python_framing(tmp_path, rec):
    """s2s_blow5_pack (host threads inside libs2s_hip.so) against _blow5_record, record by record after decompression:
    ragged sizes incl. a one-sample read, more threads than records and fewer, signals that are and are not slices of one
    packed array, a second call on the same (persistent) thread pool, and records long enough (> 160 KiB) to be deflated in
    pieces -- those must still be single ordinary zlib streams (zlib.decompress takes them whole)."""
    import struct, zlib
    from seq2squiggle_amd import codecs
    rng = np.random.default_rng(5)
    prof = U.get_profile("dna-r9-min")

    def unpack(buf):
        out, pos = [], 0
        while pos < len(buf):
            n = struct.unpack_from("<Q", buf, pos)[0]
            body = bytes(buf[pos + 8: pos + 8 + n])
            out.append(zlib.decompress(body) if rec == "zlib" else
                       codecs.zstd_decompress(body, codecs.zstd_frame_content_size(body)) if rec == "zstd" else body)
            pos += 8 + n
        return out
    for n_reads, threads in ((1, 4), (7, 2), (150, 64), (150, 3), (6, 8)):
        w = signal_io.BLOW5Writer(str(tmp_path / "p.blow5"), prof, False, "dna-r9-min", False, record_compression=rec)
        w.threads = threads
        lens = rng.integers(1, 40000, n_reads)
        lens[0] = 1
        if n_reads == 6:
            lens[1:] = (81900, 81920, 200000, 32768 * 7, 400001)        # bodies around and far beyond the split threshold
        offs = np.concatenate([[0], np.cumsum(lens)])
        dac = np.cumsum(rng.integers(-40, 41, int(offs[-1]))).astype(np.int16)
        np.random.seed(3)
        recs = w.dac_records([f"read{i}" for i in range(n_reads)], dac, offs)
        want = [w._blow5_record(r) for r in recs]
        got = unpack(w._pack_native(recs))
        assert got == unpack(b"".join(want))
        scattered = [dict(r, signal=r["signal"].copy()) for r in recs]      # not slices of one array any more
        assert unpack(w._pack_native(scattered)) == got
        if rec == "zlib":
            w.deflate = "lz"                                                # libdeflate / zlib instead of the Huffman-only encoder
            assert unpack(w._pack_native(recs)) == got


def test_huffman_only_deflate_streams(tmp_path):
    """The library's own deflate encoder (s2s_blow5_pack method 3) on inputs that reach every branch: incompressible bytes (stored
    blocks), one repeated value, a geometric histogram (code lengths beyond 15 bits before limiting), tiny and multi-piece
    records.  zlib.decompress must return the body and the stream must be smaller than stored + 1 % whenever it can be."""
    import struct, zlib
    rng = np.random.default_rng(11)
    prof = U.get_profile("dna-r10-prom")
    w = signal_io.BLOW5Writer(str(tmp_path / "h.blow5"), prof, False, "dna-r10-prom", False)
    assert w.deflate == "huffman"
    geo = np.minimum(rng.geometric(0.5, 300000), 40).astype(np.int16)        # p(v) ~ 2^-v: a 40-deep Huffman tree
    # runs of the rarest byte values (15-bit codes, several in a row at every bit phase: the encoder's four-codes-per-store path
    # must fall back when they would not fit its 64-bit word) inside a sea of one common value
    rare = np.zeros(60000, np.int16)
    for at in range(1000, 59000, 997):
        k = int(rng.integers(3, 12))
        rare[at: at + k] = (rng.integers(100, 250, k) | (rng.integers(100, 250, k) << 8)).astype(np.uint16).view(np.int16)
    sigs = [rng.integers(-32768, 32767, 100000).astype(np.int16), np.full(70001, 513, np.int16), geo,
            np.array([7], np.int16), (600 + rng.normal(0, 30, 500000)).astype(np.int16), np.zeros(2, np.int16), rare]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in sigs])])
    np.random.seed(1)
    recs = w.dac_records([f"r{i}" for i in range(len(sigs))], np.concatenate(sigs), offs)
    buf = bytes(w._pack_native(recs))
    pos = 0
    for r, sig in zip(recs, sigs):
        n = struct.unpack_from("<Q", buf, pos)[0]
        body = zlib.decompress(buf[pos + 8: pos + 8 + n])
        head, s_, tail = w._record_fields(r)
        assert body == head + s_.tobytes() + tail
        assert n <= len(body) * 1.01 + 64
        pos += 8 + n
    assert pos == len(buf)
    assert struct.unpack_from("<Q", buf, 0)[0] > 200000                      # random int16: stored
    second = 8 + struct.unpack_from("<Q", buf, 0)[0]
    assert struct.unpack_from("<Q", buf, second)[0] < 70001 * 2 / 5          # two byte values: 1.5 bits per byte


@pytest.mark.parametrize("method", [1, 2])
def test_native_row_compressor(method):
    """s2s_compress_rows: every row a stream of its own (zlib container / zstd frame), empty rows and n = 0 included, output
    bounds in out_offs, too small a buffer refused."""
    import ctypes as C, zlib
    from seq2squiggle_amd import codecs
    from seq2squiggle_amd._lib import lib
    L = lib()
    rng = np.random.default_rng(9)
    rows = [rng.integers(0, 7, n).astype(np.uint8).tobytes() for n in (0, 1, 5000, 0, 123457, 17)]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    flat = np.frombuffer(b"".join(rows) + b"\0", np.uint8)
    cap = int(L.s2s_blow5_pack_bound(int(offs[-1]), len(rows)))
    out, out_offs = np.zeros(cap, np.uint8), np.full(len(rows) + 1, -1, np.int64)
    got = L.s2s_compress_rows(flat.ctypes.data, offs.ctypes.data, len(rows), method, 1, 5, out.ctypes.data, cap, out_offs.ctypes.data)
    assert got == out_offs[-1] > 0 and out_offs[0] == 0 and np.all(np.diff(out_offs) > 0)
    for i, r in enumerate(rows):
        blob = out[out_offs[i]:out_offs[i + 1]].tobytes()
        back = zlib.decompress(blob) if method == 1 else codecs.zstd_decompress(blob, len(r))
        assert back == r
    assert L.s2s_compress_rows(flat.ctypes.data, offs.ctypes.data, len(rows), method, 1, 5, out.ctypes.data, 100, out_offs.ctypes.data) < 0
    assert L.s2s_compress_rows(None, None, 0, method, 1, 5, None, 0, out_offs.ctypes.data) == 0 and out_offs[0] == 0
    assert L.s2s_compress_rows(flat.ctypes.data, offs.ctypes.data, len(rows), 7, 1, 5, out.ctypes.data, cap, out_offs.ctypes.data) < 0


def test_onehot_to_bases_equals_chunker():
    from seq2squiggle_amd.model import onehot_to_bases
    lut = np.full(256, 255, np.uint8)
    for i, ch in enumerate("_ACGT"):
        lut[ord(ch)] = i
    for tag, k in (("k9", 9), ("k6", 6)):
        codes = load_npz(f"stages_{tag}.npz")["codes"]
        oh = torch.zeros(*codes.shape, 5, dtype=torch.float16)
        c = torch.from_numpy(codes.astype(np.int64))
        oh.scatter_(-1, c.clamp(max=4).unsqueeze(-1), (c < 5).unsqueeze(-1).to(torch.float16))
        bases, nv = onehot_to_bases(oh)
        bases, nv = bases.numpy(), nv.numpy()
        for b in range(codes.shape[0]):
            for j in range(16):
                got = lut[bases[b, j:j + k]] if j < nv[b] else np.zeros(k, np.uint8)
                assert np.array_equal(got, codes[b, j]), (tag, b, j)


def test_iter_batches_order_and_sizes():
    from seq2squiggle_amd.inference import iter_batches
    reads = [("ACGT" * 30, "a"), ("AC", "too_short"), ("ACGTT" * 50, "b"), ("A" * 9, "c")]
    batches = list(iter_batches(reads, 9, 5, "cpu"))
    ids = [i for b in batches for i in b[0]]
    assert ids == ["a"] * 7 + ["b"] * 16 + ["c"]
    assert [len(b[0]) for b in batches] == [5, 5, 5, 5, 4]
    allb = torch.cat([b[1] for b in batches]).numpy()
    exp = np.concatenate([chunker.encode_read(s, 9)[0] for s, _ in reads])
    assert np.array_equal(allb, exp)


def test_get_writer_and_check_model(tmp_path):
    from seq2squiggle_amd.inference import check_model, get_writer
    from types import SimpleNamespace
    p = U.get_profile("dna-r10-prom")
    existing = tmp_path / "sub" / "o.blow5"
    os.makedirs(existing.parent)
    existing.write_text("old")
    w, n = get_writer(str(existing), p, True, 123, "dna-r10-prom", False)
    assert isinstance(w, signal_io.BLOW5Writer) and n == 123 and not existing.exists()
    with pytest.raises(ValueError):
        get_writer(str(tmp_path / "o.fast5"), p, True, 1, "dna-r10-prom", False)
    m = SimpleNamespace(hparams=SimpleNamespace(config={"seq_kmer": 9, "dff": 256, "log_name": "x"}))
    check_model(m, {"seq_kmer": 9, "dff": 128, "log_name": "y"})           # warns only
    with pytest.raises(ValueError):
        check_model(m, {"seq_kmer": 6})


def test_shard_reads_properties():
    rng = np.random.default_rng(0)
    lens = rng.integers(5, 9000, size=500).tolist()
    for world in (1, 2, 3, 8):
        sh = parallel.shard_reads(lens, 9, world)
        assert sh[0][0] == 0 and sh[-1][1] == len(lens)
        total = sum(chunker.n_chunks(L, 9) for L in lens)
        for r in range(world):
            lo, hi, first = sh[r]
            assert first == sum(chunker.n_chunks(L, 9) for L in lens[:lo])
            if r:
                assert lo == sh[r - 1][1]
            n = sum(chunker.n_chunks(L, 9) for L in lens[lo:hi])
            assert abs(n - total / world) <= max(chunker.n_chunks(L, 9) for L in lens)
    assert parallel.rank_output_path("a/b.blow5", 3, 8) == "a/b.rank3.blow5"
    assert parallel.rank_output_path("a/b.blow5", 0, 1) == "a/b.blow5"


_DIST_WORKER = r"""
import os, sys, zlib, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from seq2squiggle_amd import parallel, utils as U
from seq2squiggle_amd.chunker import n_chunks
dist.init_process_group("gloo")
rank, local, world = parallel.rank_world()
assert (rank, world) == (dist.get_rank(), dist.get_world_size())
import random
random.seed(5)
genome, lens = U.preprocess_genome(sys.argv[2])
reads, _ = U.sample_reads_from_reference(genome, lens, 40, 3000, -1, {"max_dna_len": 16}, "x", 5)
reads = list(reads)
lo, hi, first = parallel.shard_reads([len(s) for s, _ in reads], 9, world)[rank]
mine = torch.tensor([lo, hi, first, sum(n_chunks(len(s), 9) for s, _ in reads[lo:hi]),
                     zlib.crc32(''.join(s for s, _ in reads).encode())], dtype=torch.int64)
out = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(out, mine)
if rank == 0:
    assert out[0][0] == 0 and out[-1][1] == len(reads)
    for r in range(1, world):
        assert out[r][0] == out[r - 1][1] and out[r][2] == out[r - 1][2] + out[r - 1][3]
        assert out[r][4] == out[0][4]            # every rank derived the same read set from the seed
    print("SHARDS_OK", [o.tolist()[:4] for o in out])
dist.destroy_process_group()
"""


def _free_port() -> str:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def test_two_rank_sharding_over_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_DIST_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), str(script), ROOT,
                        os.path.join(GOLDEN, "example_lambda_genome.fasta")], capture_output=True, text=True, env=env,
                       timeout=300)
    assert "SHARDS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_shard_writers_continue_the_single_process_run(tmp_path):
    """rank shard files must merge without duplicate ids -- a writer started at read index lo numbers its
    reads from lo and continues the np.random stream of the offset / median_before draws."""
    prof = U.get_profile("dna-r10-prom")
    sigs = [np.arange(1, 5 + i, dtype=np.int16) for i in range(7)]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in sigs])])
    flat = np.concatenate(sigs)
    for cls, ext in ((signal_io.BLOW5Writer, "blow5"), (signal_io.POD5Writer, "pod5")):
        np.random.seed(11)
        one = cls(str(tmp_path / f"one.{ext}"), prof, False, "dna-r10-prom", False).dac_records([f"r{i}" for i in range(7)], flat, offs)
        parts = []
        for lo, hi in ((0, 3), (3, 7)):
            np.random.seed(11)                                   # every rank process seeds the same way
            w = cls(str(tmp_path / f"p{lo}.{ext}"), prof, False, "dna-r10-prom", False)
            if lo:
                w.start_at(lo)
            parts += w.dac_records([f"r{i}" for i in range(lo, hi)], flat[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo])
        assert len(parts) == len(one) == 7
        for a, b in zip(parts, one):
            assert str(a["read_id"]) == str(b["read_id"]) and a["read_number"] == b["read_number"]
            assert a.get("offset", a.get("calibration_offset")) == b.get("offset", b.get("calibration_offset"))
            assert a["median_before"] == b["median_before"] and np.array_equal(a["signal"], b["signal"])
        assert len({str(r["read_id"]) for r in parts}) == 7


def test_merge_shards_equals_the_single_process_file(tmp_path):
    """`seq2squiggle_amd merge-shards`: the out.rankN files of a sharded run concatenate into the file one process writes
    (same records, ids and read numbers in order; one header, one end-of-file marker); bad inputs are refused."""
    prof = U.get_profile("dna-r10-prom")
    rng = np.random.default_rng(5)
    lens = rng.integers(3, 4000, 11)
    offs = np.concatenate([[0], np.cumsum(lens)])
    flat = rng.normal(600, 60, offs[-1]).astype(np.int16)
    ids = [f"read{i}" for i in range(11)]
    for ext, rd in (("blow5", signal_io.read_blow5), ("slow5", signal_io.read_slow5)):
        np.random.seed(4)
        w = signal_io.BLOW5Writer(str(tmp_path / f"one.{ext}"), prof, False, "dna-r10-prom", False)
        w.save_dac(ids, flat, offs)
        shards = []
        for r, (lo, hi) in enumerate(((0, 4), (4, 4), (4, 11))):              # the middle rank has no reads
            np.random.seed(4)
            path = parallel.rank_output_path(str(tmp_path / f"out.{ext}"), r, 3)
            w = signal_io.BLOW5Writer(path, prof, False, "dna-r10-prom", False)
            if lo:
                w.start_at(lo)
            w.save_dac(ids[lo:hi], flat[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo])
            shards.append(path)
        merged = str(tmp_path / f"merged.{ext}")
        r = subprocess.run([sys.executable, "-m", "seq2squiggle_amd", "merge-shards", *shards, "-o", merged], cwd=ROOT,
                           capture_output=True, text=True)
        assert r.returncode == 0 and "11 records" in r.stdout, r.stdout + r.stderr
        h1, one = rd(str(tmp_path / f"one.{ext}"))
        h2, got = rd(merged)
        assert len(got) == 11
        for a, b in zip(got, one):
            assert a["read_id"] == b["read_id"] and a["read_number"] == b["read_number"] and a["offset"] == b["offset"]
            assert a["median_before"] == b["median_before"] and np.array_equal(a["signal"], b["signal"])
            assert all(a[k] == b[k] for k in a if k not in ("signal", "start_time"))   # start_time counts per shard file (start_at)
        if ext == "blow5":                                                    # one header, one end-of-file marker, nothing else added
            a, b = open(merged, "rb").read(), open(tmp_path / "one.blow5", "rb").read()
            assert len(a) == len(b) and a[-5:] == b[-5:] and a[:64] == b[:64]
    # refusals: mixed containers, a truncated shard, a shard of another profile
    with pytest.raises(ValueError, match="all .blow5, all .slow5 or all .pod5"):
        signal_io.merge_shards([str(tmp_path / "one.blow5"), str(tmp_path / "one.slow5")], str(tmp_path / "x.blow5"))
    cut = tmp_path / "cut.blow5"
    cut.write_bytes(open(tmp_path / "one.blow5", "rb").read()[:-9])
    with pytest.raises(ValueError, match="end-of-file"):
        signal_io.merge_shards([str(tmp_path / "one.blow5"), str(cut)], str(tmp_path / "x.blow5"))
    other = signal_io.BLOW5Writer(str(tmp_path / "rna.blow5"), U.get_profile("rna-004-prom"), False, "rna-004-prom", False)
    other.save_dac(ids[:1], flat[:offs[1]], offs[:2])
    with pytest.raises(ValueError, match="header differs"):
        signal_io.merge_shards([str(tmp_path / "one.blow5"), str(tmp_path / "rna.blow5")], str(tmp_path / "x.blow5"))


def test_merge_pod5_shards_equals_the_single_process_file(tmp_path):
    """POD5 rank shards merge into one container with the single-process run's reads: ids, numbers, calibration, samples; the VBZ
    rows are copied as stored (also reads longer than one signal row, and a shard without reads)."""
    from seq2squiggle_amd import pod5_io
    prof = U.get_profile("dna-r10-prom")
    rng = np.random.default_rng(6)
    lens = np.array([5, 300, 102400, 102401, 250000, 17, 4000, 1, 64000])
    offs = np.concatenate([[0], np.cumsum(lens)])
    flat = (600 + 40 * rng.standard_normal(offs[-1])).astype(np.int16)
    ids = [f"read{i}" for i in range(len(lens))]
    np.random.seed(8)
    w = signal_io.POD5Writer(str(tmp_path / "one.pod5"), prof, False, "dna-r10-prom", False)
    w.write_records(w.dac_records(ids, flat, offs))
    w.close()
    shards = []
    for r, (lo, hi) in enumerate(((0, 3), (3, 3), (3, 9))):
        np.random.seed(8)
        path = parallel.rank_output_path(str(tmp_path / "out.pod5"), r, 3)
        w = signal_io.POD5Writer(path, prof, False, "dna-r10-prom", False)
        if lo:
            w.start_at(lo)
        w.write_records(w.dac_records(ids[lo:hi], flat[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo]))
        w.close()
        shards.append(path)
    merged = str(tmp_path / "merged.pod5")
    r = subprocess.run([sys.executable, "-m", "seq2squiggle_amd", "merge-shards", *shards, "-o", merged], cwd=ROOT,
                       capture_output=True, text=True)
    assert r.returncode == 0 and "9 records" in r.stdout, r.stdout + r.stderr
    one, got = pod5_io.read_pod5(str(tmp_path / "one.pod5")), pod5_io.read_pod5(merged)
    assert len(got["reads"]) == 9 and len(got["run_info"]) == 1 and got["signal_rows"] == one["signal_rows"]
    for a, b in zip(got["reads"], one["reads"]):
        assert a["read_id"] == b["read_id"] and a["read_number"] == b["read_number"] and a["num_samples"] == b["num_samples"]
        assert a["calibration_offset"] == b["calibration_offset"] and a["median_before"] == b["median_before"]
        assert a["pore_type"] == b["pore_type"] and a["end_reason"] == b["end_reason"] and np.array_equal(a["signal"], b["signal"])
    assert [len(r["signal"]) for r in got["reads"]] == list(lens)
    with pytest.raises(ValueError, match=".pod5 file"):
        signal_io.merge_shards(shards, str(tmp_path / "x.blow5"))


def _write_shards(tmp_path, ext, lens, splits, rng, signal_compression=None, tag="o"):
    prof = U.get_profile("dna-r10-prom")
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = (600 + 40 * rng.standard_normal(offs[-1])).astype(np.int16)
    ids = [f"read{i}" for i in range(len(lens))]
    shards = []
    for r, (lo, hi) in enumerate(splits):
        np.random.seed(8)
        path = parallel.rank_output_path(str(tmp_path / f"{tag}.{ext}"), r, len(splits))
        if ext == "pod5":
            from seq2squiggle_amd import pod5_io
            w = signal_io.POD5Writer(path, prof, False, "dna-r10-prom", False)
            w._stream = pod5_io.Pod5FileWriter(path, signal_compression=signal_compression)
            if lo:
                w.start_at(lo)
            w.write_records(w.dac_records(ids[lo:hi], flat[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo]))
            w.close()
        else:
            w = signal_io.BLOW5Writer(path, prof, False, "dna-r10-prom", False)
            if lo:
                w.start_at(lo)
            w.save_dac(ids[lo:hi], flat[offs[lo]:offs[hi]], offs[lo:hi + 1] - offs[lo])
        shards.append(path)
    return shards


_SHARD_SHAPES = [([5, 300, 102400, 102401, 250000, 17, 4000, 1, 64000], ((0, 3), (3, 3), (3, 9))),     # multi-row reads, an empty shard
                 (None, ((0, 120), (120, 333), (333, 457))),                                             # partial batches at every seam
                 (None, ((0, 100), (100, 200), (200, 300))),                                             # seams on batch boundaries
                 (None, ((0, 0), (0, 30))),                                                              # empty FIRST shard
                 ([7, 9], ((0, 1), (1, 2), (2, 2)))]


@pytest.mark.parametrize("signal_compression", ["vbz", "none"])
@pytest.mark.parametrize("shape", range(len(_SHARD_SHAPES)))
def test_pod5_byte_range_merge_equals_the_read_by_read_merge(tmp_path, monkeypatch, signal_compression, shape):
    """merge_pod5 moves the signal rows as raw byte ranges (copy_file_range on threads) and re-batches the signal
    table by patching pyarrow's own message metadata -- the file must be BYTE FOR BYTE what handing every read of every shard to a
    fresh Pod5FileWriter produces (tests/_merge_serial.py: the merge of rounds 2-4), for VBZ and uncompressed signal tables, seams
    inside and on batch boundaries, empty shards; take_first (shard 0 becomes the output, its full batches never move) as well."""
    import shutil
    import uuid as _uuid
    import _merge_serial as serial
    from seq2squiggle_amd import pod5_io
    rng = np.random.default_rng(60 + shape)
    lens, splits = _SHARD_SHAPES[shape]
    if lens is None:
        lens = list(rng.integers(1, 3000, splits[-1][1]))
    shards = _write_shards(tmp_path, "pod5", lens, splits, rng, signal_compression)
    fid, marker = _uuid.uuid4(), _uuid.uuid4().bytes
    want = str(tmp_path / "want.pod5")
    n = serial.merge_pod5_rebuild(shards, want, fid, marker, signal_compression)
    for threads, engine in ((1, "fd"), (3, "map-anywhere")):             # (pytest's tmp_path is a disk: "map" alone would take copy_file_range there)
        monkeypatch.setenv("S2S_MERGE_ENGINE", engine)
        got = str(tmp_path / f"got{threads}.pod5")
        assert pod5_io.merge_pod5(shards, got, threads=threads, file_identifier=fid, section_marker=marker) == n == len(lens)
        assert open(got, "rb").read() == open(want, "rb").read() and pod5_io.merge_pod5.last["engine"] == engine[:3].rstrip("-")
    assert pod5_io.merge_pod5.last["signal_rows"] == sum(-(-int(x) // pod5_io.SIGNAL_CHUNK) for x in lens)
    # take_first: identity and marker are shard 0's
    own = pod5_io.read_pod5(shards[0])["footer"]["file_identifier"]
    own_marker = open(shards[0], "rb").read()[8:24]
    want2 = str(tmp_path / "want2.pod5")
    serial.merge_pod5_rebuild(shards, want2, own, own_marker, signal_compression)
    copies = [shutil.copy(p_, p_ + ".copy.pod5") for p_ in shards]
    assert signal_io.merge_shards(copies, str(tmp_path / "taken.pod5"), threads=2, consume=True) == n
    assert open(tmp_path / "taken.pod5", "rb").read() == open(want2, "rb").read()
    assert not any(os.path.exists(c) for c in copies)
    rows = lambda xs: sum(-(-int(x) // pod5_io.SIGNAL_CHUNK) for x in xs)
    assert signal_io.merge_shards.last["batches_in_place"] == min(rows(lens[splits[0][0]:splits[0][1]]) // pod5_io.SIGNAL_BATCH_ROWS,
                                                                  -(-rows(lens) // pod5_io.SIGNAL_BATCH_ROWS))
    # and it reads back as the reads of the shards, in order
    back = pod5_io.read_pod5(str(tmp_path / "taken.pod5"))
    assert [len(r["signal"]) for r in back["reads"]] == [int(x) for x in lens]
    assert [r["read_number"] for r in back["reads"]] == list(range(len(lens)))


@pytest.mark.parametrize("ext", ["blow5", "slow5"])
def test_blow5_byte_range_merge_equals_the_record_by_record_merge(tmp_path, monkeypatch, ext):
    """The BLOW5 / SLOW5 merge copies each shard's record section as ONE range to its prefix-sum offset: byte-equal to the
    record-by-record copy of rounds 2-4; consume=True (first shard becomes the output) gives the same bytes and removes the shards;
    the record count comes from the size prefixes (s2s_blow5_scan) and doubles as the truncation check."""
    import shutil
    import _merge_serial as serial
    rng = np.random.default_rng(5)
    lens = list(rng.integers(3, 4000, 23))
    for k, splits in enumerate((((0, 4), (4, 4), (4, 23)), ((0, 0), (0, 23)), ((0, 23),), ((0, 11), (11, 23), (23, 23)))):
        shards = _write_shards(tmp_path, ext, lens, splits, rng, tag=f"s{k}")
        want = str(tmp_path / f"want{k}.{ext}")
        n = (serial.merge_blow5_serial if ext == "blow5" else serial.merge_slow5_serial)(shards, want)
        for threads, engine in ((1, "fd"), (4, "map-anywhere")):
            monkeypatch.setenv("S2S_MERGE_ENGINE", engine)
            got = str(tmp_path / f"got{k}_{threads}.{ext}")
            assert signal_io.merge_shards(shards, got, threads=threads) == n == 23
            assert open(got, "rb").read() == open(want, "rb").read()
            assert signal_io.merge_shards.last["bytes"] == os.path.getsize(want)
        copies = [shutil.copy(p_, p_ + f".copy.{ext}") for p_ in shards]
        assert signal_io.merge_shards(copies, str(tmp_path / f"taken{k}.{ext}"), consume=True) == n
        assert open(tmp_path / f"taken{k}.{ext}", "rb").read() == open(want, "rb").read()
        assert not any(os.path.exists(c) for c in copies)
    if ext == "blow5":                       # a record whose size prefix points past the end-of-file marker
        from seq2squiggle_amd import merge as M
        bad = bytearray(open(shards[0], "rb").read())
        _, _, begin, _ = M._blow5_layout(os.open(shards[0], os.O_RDONLY), shards[0])
        bad[begin:begin + 8] = (1 << 40).to_bytes(8, "little")
        (tmp_path / "bad.blow5").write_bytes(bytes(bad))
        with pytest.raises(ValueError, match="truncated record"):
            signal_io.merge_shards([shards[1], str(tmp_path / "bad.blow5")], str(tmp_path / "x.blow5"))
        assert not os.path.exists(tmp_path / "x.blow5") and not os.path.exists(tmp_path / "x.partial.blow5")


def test_copy_ranges_engines_and_python_agree(tmp_path):
    """merge.copy_ranges: both engines of s2s_copy_ranges (copy_file_range on descriptors; posix_fallocate + memcpy between shared
    mappings on several threads) and the Python fallback used when the library is absent put the same bytes in the same places;
    zero-length jobs are ignored; unaligned offsets and a destination hull that starts inside a page are fine."""
    from seq2squiggle_amd import merge as M
    rng = np.random.default_rng(3)
    src = tmp_path / "src.bin"
    data = rng.integers(0, 256, 3_000_000, dtype=np.uint8).tobytes()
    src.write_bytes(data)
    spans = [(0, 10), (10, 0), (100_000, 1_234_567), (2_999_000, 1000), (5, 70_000)]
    want = bytearray(4_000_000)
    at, jobs_of = 4099, []
    for so, ln in spans:
        want[at:at + ln] = data[so:so + ln]
        jobs_of.append((so, at, ln))
        at += ln + 3
    outs = []
    for name, hide, engine in (("fd", False, 0), ("map", False, 2), ("python", True, None)):      # (2: the mapped engine on any file system; 1 takes it on tmpfs only)
        fs, fd = os.open(src, os.O_RDONLY), os.open(tmp_path / f"{name}.bin", os.O_RDWR | os.O_CREAT)
        os.ftruncate(fd, len(want))
        if hide:
            import seq2squiggle_amd._lib as LB
            real = LB.lib
            LB.lib = lambda: (_ for _ in ()).throw(RuntimeError("hidden"))
        try:
            assert M.copy_ranges([(fs, so, fd, do, ln) for so, do, ln in jobs_of], threads=3, engine=engine) == sum(ln for _, ln in spans)
        finally:
            if hide:
                LB.lib = real
            os.close(fs)
            os.close(fd)
        outs.append(open(tmp_path / f"{name}.bin", "rb").read())
    assert outs[0] == outs[1] == outs[2] == bytes(want)


def test_predict_gpus_option_starts_one_rank_per_gpu():
    """`predict --gpus N` outside torchrun starts the same command line once per GPU as child processes (never exec) with the
    environment torchrun would give them; inside a rank (WORLD_SIZE set) the option is inert."""
    import json
    r = subprocess.run([sys.executable, "-m", "seq2squiggle_amd", "predict", "g.fa", "--gpus", "8", "-o", "o.pod5", "-c", "30",
                        "--gpus=8", "--seed", "7"], cwd=ROOT, capture_output=True, text=True,
                       env={k: v for k, v in dict(os.environ, S2S_DRY_LAUNCH="1").items() if k != "WORLD_SIZE"})
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    cmd, envs = plan["dry_launch"], plan["rank_env"]
    assert cmd[1:] == ["-m", "seq2squiggle_amd", "predict", "g.fa", "-o", "o.pod5", "-c", "30", "--seed", "7"]
    assert len(envs) == 8 and [e["RANK"] for e in envs] == [e["LOCAL_RANK"] for e in envs] == [str(i) for i in range(8)]
    assert all(e["WORLD_SIZE"] == e["LOCAL_WORLD_SIZE"] == "8" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1


_RANK_STUB = r"""
import json, os, sys, time
# stands in for `python -m seq2squiggle_amd predict ...` inside a rank: writes what the parent reads, fails on request
rank = int(os.environ["RANK"])
if os.environ.get("STUB_FAIL_RANK") == str(rank):
    sys.exit(7)
if os.environ.get("STUB_FAIL_RANK") is not None:
    time.sleep(30)                       # the parent must end this rank when its sibling fails
stamp = os.path.join(os.environ["S2S_TIMING_DIR"], "rank%d.json" % rank)
with open(stamp + ".tmp", "w") as f:
    json.dump({"ready": time.time(), "done": time.time() + 0.01}, f)
os.replace(stamp + ".tmp", stamp)
time.sleep(0.3)                          # teardown: the parent does not wait for it before it starts the merge
"""


def test_launch_ranks_collects_timing_and_ends_siblings_on_failure(tmp_path, monkeypatch):
    """cli._launch_ranks: N children with the rank environment; their ready / done stamps become launch_seconds / predict_seconds and
    it returns as soon as every rank has left its stamp (output file complete), before the processes have exited; the first
    failing rank's exit code is returned and the others are ended (by pid) instead of running on."""
    from seq2squiggle_amd import cli
    stub = tmp_path / "stub.py"
    stub.write_text(_RANK_STUB)
    monkeypatch.setattr(sys, "argv", ["seq2squiggle_amd", "predict", "g.fa", "--gpus", "3", "-o", "o.blow5"])
    monkeypatch.setattr(sys, "executable", sys.executable)
    real_popen = subprocess.Popen
    monkeypatch.setattr(subprocess, "Popen", lambda cmd, env=None, **kw: real_popen([cmd[0], str(stub)], env=env, **kw))
    monkeypatch.delenv("S2S_DRY_LAUNCH", raising=False)
    rc, timing, reap = cli._launch_ranks(3)
    assert rc == 0 and 0 <= timing["launch_seconds"] < 30 and 0 <= timing["predict_seconds"] < 1
    assert reap() == 0                                            # (the ranks' teardown is collected after the merge)
    monkeypatch.setenv("STUB_FAIL_RANK", "1")
    t0 = __import__("time").time()
    rc, timing, reap = cli._launch_ranks(3)
    assert rc == 7 and timing == {} and __import__("time").time() - t0 < 20
    reap()


_STUBBORN_RANK = r"""
import os, signal, sys, time
# a rank stuck where SIGTERM does not reach it (a HIP call that never returns): only SIGKILL ends it
rank = int(os.environ["RANK"])
open(os.path.join(os.environ["STUB_DIR"], "pid%d" % rank), "w").write(str(os.getpid()))
if rank == 0 and os.environ.get("STUB_MODE") == "fail":
    time.sleep(0.3)
    sys.exit(5)
signal.signal(signal.SIGTERM, signal.SIG_IGN)
time.sleep(120)
"""


def _alive(pid):
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    try:                                               # (a zombie that nobody has waited for yet still answers signal 0)
        return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def test_launcher_kills_a_rank_that_ignores_sigterm(tmp_path, monkeypatch):
    """after one rank fails the siblings get terminate() -- and, after a grace period, kill(): a rank stuck in a device
    call must not hang the command."""
    import time
    from seq2squiggle_amd import cli
    stub = tmp_path / "stubborn.py"
    stub.write_text(_STUBBORN_RANK)
    monkeypatch.setenv("STUB_DIR", str(tmp_path))
    monkeypatch.setenv("STUB_MODE", "fail")
    monkeypatch.setenv("S2S_RANK_GRACE", "1")
    monkeypatch.delenv("S2S_DRY_LAUNCH", raising=False)
    t0 = time.time()
    rc, timing, reap = cli._launch_ranks(3, cmd=[sys.executable, str(stub)])
    assert rc == 5 and timing == {} and time.time() - t0 < 30
    reap()
    pids = [int(open(tmp_path / f"pid{r}").read()) for r in range(3)]
    assert not any(_alive(p) for p in pids)
    assert not [d for d in os.listdir(__import__("tempfile").gettempdir()) if d.startswith("s2s-ranks-")
                and os.stat(os.path.join(__import__("tempfile").gettempdir(), d)).st_mtime >= t0 - 1]


_LAUNCHER_UNDER_SIGNAL = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
from seq2squiggle_amd import cli
open(os.path.join(os.environ["STUB_DIR"], "parent"), "w").write(str(os.getpid()))
cli._launch_ranks(2, cmd=[sys.executable, sys.argv[2]])
"""


def test_sigterm_to_the_launcher_ends_its_ranks(tmp_path):
    """SIGTERM / SIGHUP to the parent of `predict --gpus N` must not leave N ranks orphaned on their GPUs."""
    import signal
    import time
    stub = tmp_path / "stubborn.py"
    stub.write_text(_STUBBORN_RANK)
    parent = tmp_path / "parent.py"
    parent.write_text(_LAUNCHER_UNDER_SIGNAL)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "S2S_DRY_LAUNCH")}
    env.update(STUB_DIR=str(tmp_path), STUB_MODE="hang", S2S_RANK_GRACE="1")
    p = subprocess.Popen([sys.executable, str(parent), ROOT, str(stub)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    deadline = time.time() + 60
    while time.time() < deadline and not all((tmp_path / f"pid{r}").exists() and (tmp_path / f"pid{r}").read_text() for r in range(2)):
        time.sleep(0.05)
    pids = [int((tmp_path / f"pid{r}").read_text()) for r in range(2)]
    assert all(_alive(x) for x in pids)
    p.send_signal(signal.SIGTERM)
    p.communicate(timeout=60)
    assert p.returncode != 0
    time.sleep(0.2)
    assert not any(_alive(x) for x in pids)


_SEED_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
from seq2squiggle_amd import parallel
s0 = parallel.shared_seed(0)                      # (its own short-lived gloo group: created, used, destroyed inside)
assert s0 != 0 and parallel.shared_seed(77) == 77
# every rank reports through a file: a second process group in the same processes right after the first one's teardown is a
# rendezvous race of the TEST's making (it hung one run in eight), not something the product does
with open(os.path.join(sys.argv[2], "seed.rank%s" % os.environ["RANK"]), "w") as f:
    f.write(str(s0))
"""


def test_seed_zero_is_shared_between_ranks(tmp_path):
    script = tmp_path / "s.py"
    script.write_text(_SEED_WORKER)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), str(script), ROOT, str(tmp_path)],
                       capture_output=True, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    seeds = [int(open(tmp_path / f"seed.rank{i}").read()) for i in range(2)]
    assert seeds[0] == seeds[1] != 0
    assert parallel.shared_seed(0) == 0            # single process: left to set_seeds (utils.py:722-741)


def test_cli_surface():
    r = subprocess.run([sys.executable, "-m", "seq2squiggle_amd", "predict"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 1 and "required" in r.stderr + r.stdout
    r = subprocess.run([sys.executable, "-m", "seq2squiggle_amd", "predict", "--show-advanced-options"], cwd=ROOT,
                       capture_output=True, text=True)
    for opt in ("--noise-sampler", "--duration-sampler", "--dwell-mean", "--dwell-std", "--noise-std", "--distr",
                "--predict-batch-size", "--export-every-n-samples", "--sample-rate", "--bps", "--digitisation", "--range_val",
                "--offset_mean", "--offset_std", "--median_before_mean", "--median_before_std", "--min_noise", "--min_duration",
                "--min_read_len", "--preserve-read-ids", "--read-input", "--num-reads", "--read-length", "--coverage",
                "--profile", "--seed", "--model", "--config", "--verbosity"):
        assert opt in r.stdout, opt
    from seq2squiggle_amd.cli import set_config
    cfg = set_config(None)
    assert cfg["seq_kmer"] == 9 and cfg["dmodel"] == 64 and cfg["max_signal_len"] == 250 and cfg["scaling_max_value"] == 165.0
    with pytest.raises(FileNotFoundError):
        set_config("/nonexistent.yaml")


def test_super_batches_ramp_and_taper():
    """run_streaming's grouping policy: every read exactly once and in order, short first groups, full-size middle, and -- when
    the iterable knows its length -- a tail that shrinks to the floor; a plain generator (no length hint) keeps full groups."""
    from seq2squiggle_amd.inference import super_batches, _n_chunks
    from seq2squiggle_amd.utils import CountedReads
    rng = np.random.default_rng(2)
    reads = [("A" * int(L), f"r{i}") for i, L in enumerate(rng.integers(5, 9000, 800))]
    reads[3] = ("ACG", "too-short")
    live = [r for r in reads if _n_chunks(len(r[0]), 9) > 0]
    for source, knows in ((lambda: list(reads), True), (lambda: CountedReads(iter(reads), len(reads)), True),
                          (lambda: (r for r in reads), False)):
        groups = list(super_batches(source(), 9, 32768))
        assert [r for g in groups for r in g] == live
        sizes = [sum(_n_chunks(len(s), 9) for s, _ in g) for g in groups]
        assert 4096 <= sizes[0] < 4096 + 600 and 8192 <= sizes[1] < 8192 + 600 and 16384 <= sizes[2] < 16384 + 600
        assert max(sizes) < 32768 + 600
        if knows:
            assert sizes[-1] < 8192 and sizes[-2] <= 16384 + 600           # the tail shrinks
        else:
            assert all(sz >= 32768 for sz in sizes[3:-1])
    # a long job whose length is known: the groups go on doubling to 4 x max_chunks while four of that size remain, step down again
    # (never below max_chunks before the tail) and end small; an unknown length never exceeds max_chunks
    long_reads = [("A" * 10008, f"r{i}") for i in range(6000)]                      # 625 chunks each, 3.75 M chunks
    groups = list(super_batches(CountedReads(iter(long_reads), len(long_reads)), 9, 32768))
    assert [r for g in groups for r in g] == long_reads
    sizes = [625 * len(g) for g in groups]
    assert sizes[:4] == [4375, 8750, 16875, 33125] and sizes[4] >= 65536 and max(sizes) >= 131072 and max(sizes) < 131072 + 625
    peak = sizes.index(max(sizes))
    assert all(a >= b - 625 for a, b in zip(sizes[peak:], sizes[peak + 1:]))         # non-increasing from the peak on
    assert sizes[-1] < 8192 and len(sizes) < 45 and sum(sz > 131000 for sz in sizes) >= 22
    assert max(625 * len(g) for g in super_batches((r for r in long_reads), 9, 32768)) < 32768 + 625
    assert list(super_batches([], 9, 1024)) == [] and list(super_batches([("AC", "x")], 9, 1024)) == []


def test_cpu_share_honours_quota_and_ranks(monkeypatch):
    import builtins, io
    real_open = builtins.open

    def fake(quota):
        def opener(path, *a, **k):
            if path == "/sys/fs/cgroup/cpu.max":
                return io.StringIO(quota)
            return real_open(path, *a, **k)
        return opener
    cores = len(os.sched_getaffinity(0))
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.setattr(builtins, "open", fake("200000 100000\n"))
    assert signal_io.cpu_share() == min(2, cores)
    monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
    assert signal_io.cpu_share() == min(128, cores)
    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert signal_io.cpu_share() == max(1, min(16, cores) // 8)


def _live_join_case(tmp_path, ext, lens, splits, delays, tag, batch=40, signal_compression=None):
    """Rank writers on threads (each appending its shard in batches with its own pauses) beside a LiveJoin that steps in a loop, as
    cli._launch_ranks does; -> (output path, shard paths kept as copies for the reference join)."""
    import shutil
    import threading
    import time
    from seq2squiggle_amd import merge as M, pod5_io
    prof = U.get_profile("dna-r10-prom")
    rng = np.random.default_rng(17)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = (600 + 40 * rng.standard_normal(offs[-1])).astype(np.int16)
    ids = [f"read{i}" for i in range(len(lens))]
    shards = [parallel.rank_output_path(str(tmp_path / f"{tag}.{ext}"), r, len(splits)) for r in range(len(splits))]
    done = [False] * len(splits)
    # every rank's writer and records first, one rank after the other (np.random's global stream: a rank PROCESS seeds it and draws
    # alone); the threads below only write them out, each at its own pace
    writers, batches = [], []
    for r, (lo, hi) in enumerate(splits):
        np.random.seed(8)
        if ext == "pod5":
            w = signal_io.POD5Writer(shards[r], prof, False, "dna-r10-prom", False)
        else:
            w = signal_io.BLOW5Writer(shards[r], prof, False, "dna-r10-prom", False)
        if lo:
            w.start_at(lo)
        writers.append(w)
        batches.append([w.dac_records(ids[a:min(a + batch, hi)], flat[offs[a]:offs[min(a + batch, hi)]], offs[a:min(a + batch, hi) + 1] - offs[a])
                        for a in range(lo, hi, batch)] or [[]])

    def rank(r):
        w = writers[r]
        time.sleep(delays[r][0])
        if ext == "pod5":
            w._stream = pod5_io.Pod5FileWriter(shards[r], signal_compression=signal_compression)
        for recs in batches[r]:
            w.write_records(recs)
            time.sleep(delays[r][1])
        if hasattr(w, "close"):
            w.close()
        done[r] = True
    out = str(tmp_path / f"{tag}.live.{ext}")
    live = M.LiveJoin(shards, out, threads=2, punch=True)
    live.q = 7 if ext == "blow5" else 2           # small quanta: many turns on a small case
    ts = [threading.Thread(target=rank, args=(r,)) for r in range(len(splits))]
    for t in ts:
        t.start()
    while not all(done):
        if not live.step(list(done)):
            time.sleep(0.002)
    for t in ts:
        t.join()
    keep = [shutil.copy(p_, p_ + f".keep.{ext}") for p_ in shards]     # (punched where the live join has been: NOT valid inputs any more)
    n, st = live.finish(consume=True)
    assert not any(os.path.exists(p_) for p_ in shards)
    return out, n, st


@pytest.mark.parametrize("ext", ["blow5", "pod5", "pod5-none"])
def test_live_join_holds_the_same_reads_in_a_timing_independent_order(tmp_path, ext):
    """--join live (merge.LiveJoin): the parent copies complete records / full signal batches out of rank files that are still being
    written.  The result must be a valid container with exactly the reads of a single-process run -- ids, numbers, draws, samples --
    and its layout must depend on the ranks' record sequences only, not on who was faster: two runs with opposite pauses give the
    same BLOW5 bytes / the same POD5 row placement."""
    from seq2squiggle_amd import pod5_io
    comp = "none" if ext.endswith("-none") else None
    ext = ext.split("-")[0]
    rng = np.random.default_rng(23)
    lens = [int(x) for x in rng.integers(200, 5000, 523)] + [102400, 250000, 7]
    lens = lens if ext == "pod5" else lens[:523]
    n_reads = len(lens)
    splits = ((0, 170), (170, 170), (170, 400), (400, n_reads))                 # rank 1 has no reads
    prof = U.get_profile("dna-r10-prom")
    # the single-process file
    rng2 = np.random.default_rng(17)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = (600 + 40 * rng2.standard_normal(offs[-1])).astype(np.int16)
    ids = [f"read{i}" for i in range(n_reads)]
    np.random.seed(8)
    if ext == "pod5":
        w = signal_io.POD5Writer(str(tmp_path / "one.pod5"), prof, False, "dna-r10-prom", False)
        w._stream = pod5_io.Pod5FileWriter(str(tmp_path / "one.pod5"), signal_compression=comp)
        w.write_records(w.dac_records(ids, flat, offs))
        w.close()
        one = {str(r["read_id"]): r for r in pod5_io.read_pod5(str(tmp_path / "one.pod5"))["reads"]}
    else:
        w = signal_io.BLOW5Writer(str(tmp_path / "one.blow5"), prof, False, "dna-r10-prom", False)
        w.save_dac(ids, flat, offs)
        one = {r["read_id"]: r for r in signal_io.read_blow5(str(tmp_path / "one.blow5"))[1]}
    outs = []
    for k, delays in enumerate((((0.0, 0.004), (0.01, 0.0), (0.02, 0.001), (0.0, 0.0)),
                                ((0.03, 0.0), (0.0, 0.0), (0.0, 0.003), (0.01, 0.005)))):
        out, n, st = _live_join_case(tmp_path, ext, lens, splits, delays, f"run{k}", signal_compression=comp)
        assert n == n_reads and st["units_live"] > 0 and 0 <= st["live_bytes"] < st["bytes"], st     # (how much went before the "ranks" were done is timing)
        outs.append(out)
        if ext == "pod5":
            got = pod5_io.read_pod5(out)
            assert [r["read_number"] for r in got["reads"]] == list(range(n_reads))     # the reads table stays in read order
            assert len(got["run_info"]) == 1 and got["signal_rows"] == sum(-(-x // pod5_io.SIGNAL_CHUNK) for x in lens)
            for r in got["reads"]:
                ref = one[str(r["read_id"])]
                assert np.array_equal(r["signal"], ref["signal"]) and r["read_number"] == ref["read_number"]
                assert r["calibration_offset"] == ref["calibration_offset"] and r["median_before"] == ref["median_before"]
            assert sum(1 for _ in pod5_io.iter_pod5(out)) == n_reads                    # (the batch-at-a-time reader agrees with the layout)
        else:
            _, got = signal_io.read_blow5(out)
            assert len(got) == n_reads and {r["read_id"] for r in got} == set(one)
            for r in got:
                ref = one[r["read_id"]]
                assert np.array_equal(r["signal"], ref["signal"]) and r["read_number"] == ref["read_number"] and r["offset"] == ref["offset"]
            assert [r["read_number"] for r in got] != list(range(n_reads))              # an interleave, by design
    if ext == "blow5":
        # the same bytes -- but for the header's wall-clock attribute (exp_start_time, to the second: the two runs may straddle one)
        def parts(path):
            raw = open(path, "rb").read()
            hlen = struct.unpack_from("<I", raw, 64)[0]
            text = [l for l in raw[68:68 + hlen].decode().splitlines() if not l.startswith("@exp_start_time")]
            return raw[:64], text, raw[68 + hlen:]
        assert parts(outs[0]) == parts(outs[1])
    else:
        place = []
        for o in outs:
            t = pod5_io._Shard(o)
            tab = t.table(pod5_io.CT_READS).read_all()
            place.append(tab.column("signal").to_pylist())
            t.close()
        assert place[0] == place[1]


def test_live_join_refuses_a_rank_file_that_does_not_end(tmp_path):
    """A rank that is 'done' but whose file stops in the middle of a record (a crash after the stamp cannot happen, a full disk
    can): the live join raises instead of closing a container with a torn record, and abort() leaves no output behind."""
    from seq2squiggle_amd import merge as M
    rng = np.random.default_rng(2)
    shards = _write_shards(tmp_path, "blow5", list(rng.integers(100, 900, 30)), ((0, 12), (12, 30)), rng, tag="t")
    whole = open(shards[1], "rb").read()
    open(shards[1], "wb").write(whole[:-9])                      # the end marker and four bytes of the last record are missing
    live = M.LiveJoin(shards, str(tmp_path / "t.live.blow5"), threads=1, punch=False)
    with pytest.raises(ValueError, match="end-of-file marker"):
        live.finish()
    live.abort()
    assert not os.path.exists(tmp_path / "t.live.blow5") and all(os.path.exists(p_) for p_ in shards)
    with pytest.raises(ValueError, match=".blow5 and .pod5"):
        M.LiveJoin(shards, str(tmp_path / "x.slow5"))


def _allocated(path):
    return os.stat(path).st_blocks


@pytest.mark.parametrize("ext", ["blow5", "pod5"])
def test_live_join_checks_every_header_before_it_gives_a_page_back(tmp_path, ext):
    """the live join frees the pages of what it has copied.  Two shards that do not belong together -- another
    profile in the BLOW5 header, a VBZ and an uncompressed POD5 signal table -- used to be noticed in finish(), when the rank files
    were hollow and abort() had deleted the partial output: everything gone.  Now every shard's header is held against shard 0's
    when it is first seen and nothing is punched before all of them have passed: the join fails with the rank files whole."""
    from seq2squiggle_amd import merge as M
    rng = np.random.default_rng(3)
    lens = list(rng.integers(3000, 9000, 700))
    if ext == "blow5":
        shards = _write_shards(tmp_path, ext, lens, ((0, 350), (350, 700)), rng, tag="h")
        raw = open(shards[1], "rb").read()
        assert b"dna-r10-prom" in raw[:4096] or b"sample_frequency" in raw[:4096]
        other = raw.replace(b"@asic_id", b"@asic_ix", 1) if b"@asic_id" in raw[:4096] else raw.replace(b"sample_frequency", b"sample_frequencx", 1)
        assert other != raw and len(other) == len(raw)
        open(shards[1], "wb").write(other)
        match = "header differs"
    else:
        a = _write_shards(tmp_path, ext, lens[:350], ((0, 350),), rng, signal_compression="vbz", tag="h0")
        b = _write_shards(tmp_path, ext, lens[350:], ((0, 350),), rng, signal_compression="none", tag="h1")
        shards = [str(tmp_path / "h.rank0.pod5"), str(tmp_path / "h.rank1.pod5")]
        os.rename(a[0], shards[0])
        os.rename(b[0], shards[1])
        match = "schema differs"
    before = [open(p_, "rb").read() for p_ in shards]
    blocks = [_allocated(p_) for p_ in shards]
    out = str(tmp_path / f"h.live.{ext}")
    live = M.LiveJoin(shards, out, threads=1, punch=True)
    with pytest.raises(ValueError, match=match):
        live.finish()
    live.abort()
    assert live.punched == 0 and not os.path.exists(out)
    assert [open(p_, "rb").read() for p_ in shards] == before           # nothing was hollowed out
    assert [_allocated(p_) for p_ in shards] == blocks


def test_live_join_keeps_the_partial_output_once_pages_are_gone(tmp_path, caplog):
    """... and if the join fails AFTER pages have been given back (a torn last record, a full disk), the partial output is the only
    copy of those records: abort() keeps it and the log says where, instead of deleting it."""
    import logging
    from seq2squiggle_amd import merge as M
    rng = np.random.default_rng(4)
    lens = list(rng.integers(3000, 9000, 1200))
    shards = _write_shards(tmp_path, "blow5", lens, ((0, 600), (600, 1200)), rng, tag="k")
    whole = open(shards[1], "rb").read()
    open(shards[1], "wb").write(whole[:-9])                          # rank 1's file stops inside its last record
    out = str(tmp_path / "k.live.blow5")
    live = M.LiveJoin(shards, out, threads=1, punch=True)
    with caplog.at_level(logging.ERROR, logger="seq2squiggle"):
        with pytest.raises(ValueError, match="end-of-file marker"):
            live.finish()
        assert live.punched > 0
        live.abort()
    assert os.path.exists(out) and os.path.getsize(out) > 1 << 20
    assert out in caplog.text and "kept" in caplog.text
    assert live.fds == [None, None] and live.out_fd is None
    live.abort()                                                      # (idempotent: closes nothing twice)


def test_cli_rank_file_names_are_those_of_the_ranks():
    """`predict --gpus N --join live` names the rank files itself before it starts the ranks (so that nothing heavy is imported in front
    of their start): the names must be parallel.rank_output_path's, which the ranks use."""
    for out in ("a/b/out.blow5", "x.pod5", "dir.v2/reads.final.blow5"):
        ext = os.path.splitext(out)[1]
        base = out[:len(out) - len(ext)]
        assert [f"{base}.rank{r}{ext}" for r in range(3)] == [parallel.rank_output_path(out, r, 3) for r in range(3)]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_live_join_random_shapes_and_paces(tmp_path, seed):
    """LiveJoin over random rank counts, splits (empty ranks included), read lengths and writer paces, BLOW5 and POD5 by turns: the
    output always parses, holds every read once with the single-stream ids and samples, and leaves no rank file behind."""
    from seq2squiggle_amd import pod5_io
    rng = np.random.default_rng(100 + seed)
    ext = ("blow5", "pod5", "pod5")[seed % 3]
    n_ranks = int(rng.integers(2, 6))
    n_reads = int(rng.integers(150, 700))
    lens = [int(x) for x in rng.integers(50, 3000, n_reads)]
    cuts = sorted(int(x) for x in rng.integers(0, n_reads + 1, n_ranks - 1))
    bounds = [0] + cuts + [n_reads]
    splits = tuple((bounds[i], bounds[i + 1]) for i in range(n_ranks))
    delays = tuple((float(rng.uniform(0, 0.02)), float(rng.uniform(0, 0.004))) for _ in range(n_ranks))
    out, n, st = _live_join_case(tmp_path, ext, lens, splits, delays, f"rnd{seed}", batch=int(rng.integers(5, 90)))
    assert n == n_reads and not [f for f in os.listdir(tmp_path) if ".rank" in f and ".keep." not in f]
    if ext == "pod5":
        got = pod5_io.read_pod5(out)["reads"]
        assert [r["read_number"] for r in got] == list(range(n_reads)) and [len(r["signal"]) for r in got] == lens
    else:
        got = signal_io.read_blow5(out)[1]
        assert sorted(r["read_number"] for r in got) == list(range(n_reads))
        assert {r["read_number"]: r["len_raw_signal"] for r in got} == dict(enumerate(lens))
