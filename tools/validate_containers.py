#!/usr/bin/env python
"""Opt-in validator of the container writers against the REAL libraries (pyslow5 / pod5), for whoever has them.

The reference writes its output through pyslow5 1.3.0 and pod5 0.3.27 (reference signal_io.py:96-101, 167-171, 268-282).
Neither library exists in the build image, so seq2squiggle_amd's BLOW5 and POD5 writers follow the published format
descriptions and have only ever been read back by the in-tree readers.  This tool closes that gap on a machine that has the
libraries:

    python tools/validate_containers.py                 # synthetic records through the writers (no GPU needed)
    python tools/validate_containers.py --from-cli      # `seq2squiggle_amd predict tests/golden/example_test.fasta` (needs the GPU)

For every variant -- BLOW5 {zlib, zstd, none} records x {raw, svb-zd} signal; SLOW5 ASCII; POD5 {vbz, none} -- it writes a
file, opens it with the external library and compares, read by read, ids / calibration / auxiliary fields / samples with
what the in-tree reader (signal_io.read_blow5, pod5_io.read_pod5) returns for the same file.  Exit code 0 = every variant
whose library is present validated; 2 = nothing could be checked (no library); 1 = a mismatch (printed).

tests/test_container_validation.py runs the same checks under pytest and skips when the libraries are absent.
"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from seq2squiggle_amd import pod5_io, signal_io  # noqa: E402
from seq2squiggle_amd.utils import get_profile  # noqa: E402

BLOW5_VARIANTS = [("zlib", "none"), ("zstd", "none"), ("none", "none"), ("zlib", "svb-zd"), ("zstd", "svb-zd")]
POD5_VARIANTS = ["vbz", "none"]


def have(name):
    try:
        __import__(name)
        return True
    except Exception:
        return False


def synthetic_reads(n=40, seed=3):
    """(read ids, packed int16 samples, offsets): noisy levels like a nanopore signal, one read longer than a POD5 signal row,
    one of a single sample, deltas that need 1, 2 and 3 svb bytes."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(50, 9000, n)
    lens[1], lens[2] = 1, 3 * pod5_io.SIGNAL_CHUNK + 17
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dac = (rng.normal(500, 40, offs[-1]) + 120 * rng.integers(-1, 2, offs[-1])).astype(np.int16)
    dac[offs[3]: offs[3] + 8] = [-32768, 32767, -32768, 0, 32767, 32767, -1, 1]      # 3-byte zig-zag deltas
    return [f"read_{i}" for i in range(n)], dac, offs


def write_blow5(path, record, signal, profile_name="dna-r10-prom", from_cli=False):
    if from_cli:
        env = dict(os.environ, S2S_BLOW5_RECORD=record, S2S_BLOW5_SIGNAL=signal)
        cli(path, env)
        return
    np.random.seed(11)
    w = signal_io.BLOW5Writer(path, get_profile(profile_name), False, profile_name, False, record_compression=record,
                              signal_compression=signal)
    ids, dac, offs = synthetic_reads()
    half = len(ids) // 2                                           # two batches: the append path (EOF marker dropped, re-written)
    w.save_dac(ids[:half], dac, offs[:half + 1])
    w.save_dac(ids[half:], dac, offs[half:])


def write_pod5(path, signal, profile_name="dna-r10-prom", from_cli=False):
    if from_cli:
        cli(path, dict(os.environ, S2S_POD5_SIGNAL=signal))
        return
    os.environ["S2S_POD5_SIGNAL"] = signal
    np.random.seed(11)
    w = signal_io.POD5Writer(path, get_profile(profile_name), False, profile_name, False)
    ids, dac, offs = synthetic_reads()
    half = len(ids) // 2
    w.write_records(w.dac_records(ids[:half], dac, offs[:half + 1]))
    w.write_records(w.dac_records(ids[half:], dac, offs[half:]))
    w.close()


def cli(out, env):
    r = subprocess.run([sys.executable, "-m", "seq2squiggle_amd", "predict", os.path.join(ROOT, "tests", "golden", "example_test.fasta"),
                        "--read-input", "-o", out, "-m", os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"), "--seed", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("CLI failed: " + r.stderr[-1500:])


def check_blow5(path):
    """The file through pyslow5 against signal_io.read_blow5 / read_slow5 -> list of mismatch strings."""
    import pyslow5
    bad = []
    ours = (signal_io.read_slow5 if path.endswith(".slow5") else signal_io.read_blow5)(path)[1]
    s5 = pyslow5.Open(path, "r")
    theirs = list(s5.seq_reads(aux="all"))
    if len(theirs) != len(ours):
        return [f"{path}: pyslow5 sees {len(theirs)} reads, the in-tree reader {len(ours)}"]
    for a, b in zip(theirs, ours):
        rid = b["read_id"]
        if a["read_id"] != rid:
            bad.append(f"{path}: read id {a['read_id']} != {rid}")
        if not np.array_equal(np.asarray(a["signal"], dtype=np.int16), b["signal"]):
            bad.append(f"{path}: samples of {rid} differ")
        for k in ("digitisation", "offset", "range", "sampling_rate", "len_raw_signal", "median_before", "read_number",
                  "start_mux", "start_time"):
            if k in a and float(a[k]) != float(b[k]):
                bad.append(f"{path}: {rid}.{k}: {a[k]} != {b[k]}")
        if "channel_number" in a and str(a["channel_number"]) != b["channel_number"]:
            bad.append(f"{path}: {rid}.channel_number")
    for attr in ("flow_cell_product_code", "sequencing_kit", "sample_frequency", "experiment_type"):
        try:
            if s5.get_header_value(attr) in (None, ""):
                bad.append(f"{path}: header attribute {attr} missing")
        except Exception as e:                                       # noqa: BLE001
            bad.append(f"{path}: header attribute {attr}: {e}")
    s5.close()
    return bad


def check_pod5(path):
    """The file through pod5.Reader against pod5_io.read_pod5 -> list of mismatch strings."""
    import pod5
    bad = []
    ours = pod5_io.read_pod5(path)["reads"]
    with pod5.Reader(path) as rd:
        theirs = list(rd.reads())
        if len(theirs) != len(ours):
            return [f"{path}: pod5 sees {len(theirs)} reads, the in-tree reader {len(ours)}"]
        for a, b in zip(theirs, ours):
            rid = str(b["read_id"])
            if str(a.read_id) != rid:
                bad.append(f"{path}: read id {a.read_id} != {rid}")
            if not np.array_equal(np.asarray(a.signal, dtype=np.int16), b["signal"]):
                bad.append(f"{path}: samples of {rid} differ")
            if a.num_samples != b["num_samples"] or a.read_number != b["read_number"]:
                bad.append(f"{path}: {rid}: num_samples / read_number")
            if np.float32(a.calibration.offset) != np.float32(b["calibration_offset"]) or \
                    np.float32(a.calibration.scale) != np.float32(b["calibration_scale"]):
                bad.append(f"{path}: {rid}: calibration")
            if a.end_reason.name != b["end_reason"] or a.pore.channel != b["channel"] or a.pore.well != b["well"]:
                bad.append(f"{path}: {rid}: end_reason / pore")
            if a.run_info.sample_rate != 5000 or not a.run_info.flow_cell_product_code:
                bad.append(f"{path}: {rid}: run_info")
    return bad


def validate(from_cli=False, workdir=None, log=print):
    """-> (variants checked, mismatches).  Variants whose library is missing are reported and skipped."""
    checked, bad = 0, []
    with tempfile.TemporaryDirectory(dir=workdir) as td:
        if have("pyslow5"):
            for rec, sig in BLOW5_VARIANTS:
                p = os.path.join(td, f"v_{rec}_{sig}.blow5")
                write_blow5(p, rec, sig, from_cli=from_cli)
                b = check_blow5(p)
                log(f"BLOW5 records={rec:5s} signal={sig:7s}: {'OK' if not b else 'MISMATCH'}")
                bad += b
                checked += 1
            p = os.path.join(td, "v.slow5")
            write_blow5(p, "none", "none", from_cli=from_cli)
            b = check_blow5(p)
            log(f"SLOW5 ASCII                        : {'OK' if not b else 'MISMATCH'}")
            bad += b
            checked += 1
        else:
            log("pyslow5 is not importable: BLOW5 / SLOW5 variants NOT validated")
        if have("pod5"):
            for sig in POD5_VARIANTS:
                p = os.path.join(td, f"v_{sig}.pod5")
                write_pod5(p, sig, from_cli=from_cli)
                b = check_pod5(p)
                log(f"POD5 signal={sig:4s}                  : {'OK' if not b else 'MISMATCH'}")
                bad += b
                checked += 1
        else:
            log("pod5 is not importable: POD5 variants NOT validated")
    return checked, bad


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--from-cli", action="store_true", help="write the files with `seq2squiggle_amd predict` (needs the MI355X)")
    a = ap.parse_args()
    checked, bad = validate(a.from_cli)
    for b in bad:
        print("MISMATCH:", b)
    if not checked:
        print("nothing validated: install pyslow5 and/or pod5 (pip install pyslow5 pod5) and run again")
        sys.exit(2)
    print(f"{checked} variants checked, {len(bad)} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
