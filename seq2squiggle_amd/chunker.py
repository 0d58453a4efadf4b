"""Read -> chunk encoding for the device path.

Reference: split_sequence (utils.py:350-356) = extract_kmers (334-339) + add_remainder (342-347)
+ one_hot_encode (56-89) + regular_break_points (266-287), driven per read by process_read
(dataloader.py:358-398).  The reference materialises fp16 one-hot [C,16,k,5]; the device path
takes the raw bases instead: chunk c is read[16c : 16c+15+k] plus the number of real k-mers in
it (k-mers past the read's end are the all-"_" pad k-mer, which is NOT a window over padded
bases, so it cannot be expressed by padding the bytes alone).
"""
from typing import Iterable, List, Sequence, Tuple

import numpy as np

T_ENC = 16


def n_chunks(read_len: int, k: int) -> int:
    n_kmer = read_len - k + 1
    return 0 if n_kmer <= 0 else -(-n_kmer // T_ENC)


def encode_read(seq: str, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """-> (bases uint8 [C, 16+k-1] ASCII, n_valid uint8 [C]).  A read shorter than k gives C == 0
    (the reference skips it, dataloader.py:393-398)."""
    nb = T_ENC + k - 1
    C = n_chunks(len(seq), k)
    if C == 0:
        return np.zeros((0, nb), np.uint8), np.zeros((0,), np.uint8)
    raw = np.frombuffer(seq.encode("latin-1"), dtype=np.uint8)
    buf = np.full(T_ENC * C + k - 1, ord("_"), dtype=np.uint8)
    buf[: raw.size] = raw
    idx = (T_ENC * np.arange(C))[:, None] + np.arange(nb)[None, :]
    n_kmer = len(seq) - k + 1
    nv = np.minimum(T_ENC, n_kmer - T_ENC * np.arange(C)).astype(np.uint8)
    return buf[idx], nv


def encode_reads(seqs: Sequence[str], k: int):
    """Encode many reads -> (bases [C_total, nb], n_valid [C_total], read_first int32 [R+1]).
    Reads that yield no chunk keep an empty range."""
    parts, nvs, first = [], [], [0]
    for s in seqs:
        b, nv = encode_read(s, k)
        parts.append(b)
        nvs.append(nv)
        first.append(first[-1] + b.shape[0])
    nb = T_ENC + k - 1
    bases = np.concatenate(parts, 0) if parts else np.zeros((0, nb), np.uint8)
    n_valid = np.concatenate(nvs, 0) if nvs else np.zeros((0,), np.uint8)
    return bases, n_valid, np.asarray(first, dtype=np.int32)


def codes_to_bases(codes: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Inverse helper for tests: integer codes [B,16,k] (0..4, 255 = unknown letter) as produced by
    the reference's one-hot -> (bases, n_valid).  Trailing all-"_" k-mers are taken as padding."""
    B, T, k = codes.shape
    lut = np.array([ord(c) for c in "_ACGT"] + [ord("N")] * 251, dtype=np.uint8)
    is_pad = (codes == 0).all(-1)
    n_valid = np.zeros(B, np.uint8)
    bases = np.full((B, T + k - 1), ord("_"), np.uint8)
    for b in range(B):
        nv = T
        while nv > 0 and is_pad[b, nv - 1]:
            nv -= 1
        nv = max(nv, 1) if not is_pad[b].all() else 0
        n_valid[b] = nv
        for j in range(nv):
            bases[b, j] = lut[codes[b, j, 0]]
        if nv:
            bases[b, nv - 1: nv - 1 + k] = lut[codes[b, nv - 1]]
    return bases, n_valid
