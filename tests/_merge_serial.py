"""The record-by-record / read-by-read shard merges of rounds 2-4, kept as the DEFINITION the parallel byte-range merge of
seq2squiggle_amd/merge.py is tested against (test infrastructure: nothing under seq2squiggle_amd/ imports this)."""
import os
import struct

from seq2squiggle_amd import pod5_io
from seq2squiggle_amd.signal_io import BLOW5Writer


def merge_blow5_serial(paths, out: str) -> int:
    """Header of the first shard, every shard's records in order (copied one at a time), one end-of-file marker."""
    n = 0
    with open(out, "wb") as fo:
        for i, p_ in enumerate(paths):
            size = os.path.getsize(p_)
            with open(p_, "rb") as fi:
                head = fi.read(68)
                hlen = struct.unpack_from("<I", head, 64)[0]
                text = fi.read(hlen)
                if i == 0:
                    fo.write(head + text)
                pos, end = 68 + hlen, size - len(BLOW5Writer._EOF)
                while pos < end:
                    (rec,) = struct.unpack("<Q", fi.read(8))
                    fo.write(struct.pack("<Q", rec) + fi.read(rec))
                    pos += 8 + rec
                    n += 1
        fo.write(BLOW5Writer._EOF)
    return n


def merge_slow5_serial(paths, out: str) -> int:
    n = 0
    with open(out, "w") as fo:
        for i, p_ in enumerate(paths):
            with open(p_) as fi:
                for line in fi:
                    if line.startswith(("#", "@")):
                        if i == 0:
                            fo.write(line)
                        continue
                    fo.write(line)
                    n += 1
    return n


def merge_pod5_rebuild(paths, out: str, file_identifier=None, section_marker=None, signal_compression="vbz") -> int:
    """Every read of every shard handed to a fresh Pod5FileWriter (VBZ rows as stored, or decoded samples for uncompressed
    signal tables); run-info records united by acquisition id."""
    n = 0
    vbz = signal_compression == "vbz"
    with pod5_io.Pod5FileWriter(out, file_identifier, section_marker, signal_compression=signal_compression) as w:
        for p_ in paths:
            runs = None
            for r, rows in pod5_io.iter_pod5(p_, decode=not vbz):
                if vbz:
                    ri = dict(r["run_info_record"])
                else:
                    if runs is None:
                        runs = {x["acquisition_id"]: x for x in pod5_io.read_pod5(p_)["run_info"]}
                    ri = dict(runs[r["run_info"]])
                ri["context_tags"], ri["tracking_id"] = dict(ri["context_tags"]), dict(ri["tracking_id"])
                rec = dict(read_id=r["read_id"], num_samples=r["num_samples"], read_number=r["read_number"],
                           start_sample=r["start"], median_before=r["median_before"], channel=r["channel"], well=r["well"],
                           pore_type=r["pore_type"], calibration_offset=r["calibration_offset"],
                           calibration_scale=r["calibration_scale"], end_reason=r["end_reason"],
                           end_reason_forced=r["end_reason_forced"], run_info=ri)
                if vbz:
                    rec["vbz_rows"] = rows
                else:
                    rec["signal"] = rows
                w.add_reads([rec])
                n += 1
    return n
