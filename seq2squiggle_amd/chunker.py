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


def pack_reads(seqs: Sequence[str], k: int):
    """Flat form for device-side chunking: -> (flat uint8 bytes with every read padded by "_" to 16*C + k - 1,
    chunk_start int64 [C_total] into flat, n_valid uint8 [C_total], read_first int32 [R+1]).  Chunk b is
    flat[chunk_start[b] : chunk_start[b] + 16 + k - 1]: one gather on the GPU replaces the per-read window copies."""
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    n_kmer = np.maximum(lens - k + 1, 0)
    C = -(-n_kmer // T_ENC)
    padded = np.where(C > 0, T_ENC * C + k - 1, 0)
    base = np.concatenate([[0], np.cumsum(padded)])
    # one join instead of a per-read array copy; reads that yield no chunk (len < k) contribute nothing
    blob = b"".join((s_.encode("latin-1") + b"_" * int(padded[i] - lens[i])) if C[i] > 0 else b"" for i, s_ in enumerate(seqs))
    flat = np.frombuffer(bytearray(blob + b"_"), dtype=np.uint8)          # (bytearray: a writable buffer for torch.from_numpy)
    read_first = np.concatenate([[0], np.cumsum(C)]).astype(np.int32)
    within = np.arange(int(read_first[-1])) - np.repeat(read_first[:-1], C)          # chunk index inside its read
    chunk_start = np.repeat(base[:-1], C) + T_ENC * within
    n_valid = np.minimum(T_ENC, np.repeat(n_kmer, C) - T_ENC * within).astype(np.uint8)
    return flat, chunk_start.astype(np.int64), n_valid, read_first
