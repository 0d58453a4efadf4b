"""Signal codecs of the output containers, vectorised numpy + zstd.

The reference never touches these bytes itself: pyslow5 1.3.0 and pod5 0.3.27 (poetry.lock) compress inside
`write_record_batch` / `Writer.add_reads` (reference signal_io.py:96-101,167-171 and 201-282).  Neither library is in this
image, so the published algorithms are restated here:

* svb-zd (slow5lib `slow5_press.c`: ptr_compress_svb_zd): int16 samples -> int32 -> delta against the previous
  sample (the first against 0) -> zig-zag ((v << 1) ^ (v >> 31)) -> StreamVByte "1234" encoding of the uint32 values
  (Lemire & Kurz: a control stream of 2 bits per value = byte length - 1, value i in bits 2(i%4) of control byte
  i/4, followed by the data stream of 1..4 little-endian bytes per value), the whole prefixed by the uint32 number of
  values.
* VBZ (pod5 `signal_compression.cpp`: compress_signal): int16 samples -> delta (uint16 arithmetic, first against 0) ->
  zig-zag ((v + v) ^ (v >> 15)) -> svb16 (Rimmer's 16-bit StreamVByte: a control stream of 1 bit per value, LSB first,
  set = two data bytes, followed by the data stream of 1..2 little-endian bytes per value) -> one zstd frame (level 1).

The encoders are pinned by hand-computed known-answer vectors (tests/test_codecs_cpu.py, tests/golden/codec_kat.json)
and by decode(encode(x)) == x over random and extreme inputs.  s2s_svb_* in the HIP library produce the same bytes on
the GPU (tests/test_gpu_codecs.py)."""
import ctypes
import struct
from typing import Tuple

import numpy as np

# ------------------------------------------------------------------------------------------------ zstd
_ZSTD = None


def _zstd():
    """(compress(bytes, level) -> bytes, decompress(bytes, size) -> bytes): pyarrow's codec when present (releases the
    GIL), else libzstd.so.1 through ctypes.  Both emit ordinary zstd frames."""
    global _ZSTD
    if _ZSTD is not None:
        return _ZSTD
    try:
        import pyarrow as pa
        if not pa.Codec.is_available("zstd"):
            raise ImportError
        codecs = {}

        def comp(data, level=1):
            c = codecs.get(level) or codecs.setdefault(level, pa.Codec("zstd", compression_level=level))
            return c.compress(data, asbytes=True)

        def decomp(data, size):
            return pa.Codec("zstd").decompress(data, decompressed_size=size, asbytes=True)
        _ZSTD = (comp, decomp)
    except ImportError:
        z = ctypes.CDLL("libzstd.so.1")
        z.ZSTD_compressBound.restype = ctypes.c_size_t
        z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        z.ZSTD_compress.restype = ctypes.c_size_t
        z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        z.ZSTD_decompress.restype = ctypes.c_size_t
        z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        z.ZSTD_isError.restype = ctypes.c_uint
        z.ZSTD_isError.argtypes = [ctypes.c_size_t]

        def comp(data, level=1):
            data = bytes(data)
            cap = z.ZSTD_compressBound(len(data))
            out = ctypes.create_string_buffer(cap)
            n = z.ZSTD_compress(out, cap, data, len(data), level)
            if z.ZSTD_isError(n):
                raise RuntimeError("ZSTD_compress failed")
            return out.raw[:n]

        def decomp(data, size):
            data = bytes(data)
            out = ctypes.create_string_buffer(max(size, 1))
            n = z.ZSTD_decompress(out, size, data, len(data))
            if z.ZSTD_isError(n) or n != size:
                raise ValueError("ZSTD_decompress failed")
            return out.raw[:n]
        _ZSTD = (comp, decomp)
    return _ZSTD


def zstd_compress(data, level: int = 1) -> bytes:
    return _zstd()[0](data, level)


def zstd_decompress(data, size: int) -> bytes:
    return _zstd()[1](data, size)


# ------------------------------------------------------------------------------------------------ zig-zag delta
def zigzag_delta16(samples: np.ndarray) -> np.ndarray:
    """int16 [n] -> uint16 [n]: delta in uint16 arithmetic (wraps), then zig-zag (pod5 svb16 encode<Delta, Zigzag>)."""
    x = np.ascontiguousarray(samples, dtype=np.int16).view(np.uint16)
    d = x.copy()
    d[1:] -= x[:-1]                                     # first delta is against 0
    s = d.view(np.int16)
    return ((s.astype(np.int32) << 1) ^ (s.astype(np.int32) >> 15)).astype(np.uint16)


def unzigzag_delta16(u: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(u, dtype=np.uint16)
    d = ((u >> 1) ^ (-(u & 1).astype(np.int16)).view(np.uint16)).astype(np.uint16)
    return np.cumsum(d, dtype=np.uint16).view(np.int16)


def zigzag_delta32(samples: np.ndarray) -> np.ndarray:
    """int16 [n] -> uint32 [n]: widen, delta against the previous sample (first against 0), zig-zag (slow5lib)."""
    x = np.ascontiguousarray(samples, dtype=np.int16).astype(np.int32)
    d = x.copy()
    d[1:] -= x[:-1]
    return ((d << 1) ^ (d >> 31)).astype(np.uint32)


def unzigzag_delta32(u: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(u, dtype=np.uint32)
    d = ((u >> 1).astype(np.int64) ^ -(u & 1).astype(np.int64)).astype(np.int32)
    return np.cumsum(d, dtype=np.int32).astype(np.int16)


# ------------------------------------------------------------------------------------------------ StreamVByte
def svb16_encode(u: np.ndarray) -> bytes:
    """uint16 [n] -> control bytes ((n+7)//8, bit i%8 of byte i//8 set when value i takes two bytes) + data bytes."""
    u = np.ascontiguousarray(u, dtype=np.uint16)
    n = u.size
    two = u > 0xFF
    keys = np.packbits(two, bitorder="little")
    pos = np.arange(n, dtype=np.int64) + np.cumsum(two, dtype=np.int64) - two
    data = np.empty(n + int(two.sum()), dtype=np.uint8)
    data[pos] = (u & 0xFF).astype(np.uint8)
    data[pos[two] + 1] = (u[two] >> 8).astype(np.uint8)
    return keys.tobytes() + data.tobytes()


def svb16_decode(buf, n: int) -> np.ndarray:
    b = np.frombuffer(buf, dtype=np.uint8)
    nk = (n + 7) // 8
    two = np.unpackbits(b[:nk], bitorder="little")[:n].astype(bool)
    data = b[nk:]
    pos = np.arange(n, dtype=np.int64) + np.cumsum(two, dtype=np.int64) - two
    if n and pos[-1] + 1 + two[-1] > data.size:
        raise ValueError("svb16 stream is truncated")
    u = data[pos].astype(np.uint16)
    u[two] |= data[pos[two] + 1].astype(np.uint16) << 8
    return u


def svb32_encode(u: np.ndarray) -> bytes:
    """uint32 [n] -> control bytes ((n+3)//4, code = byte length - 1 of value i in bits 2(i%4) of byte i//4) + data."""
    u = np.ascontiguousarray(u, dtype=np.uint32)
    n = u.size
    code = (u > 0xFF).astype(np.uint8) + (u > 0xFFFF) + (u > 0xFFFFFF)
    pad = np.zeros((-n) % 4, dtype=np.uint8)
    c4 = np.concatenate([code, pad]).reshape(-1, 4)
    keys = (c4[:, 0] | (c4[:, 1] << 2) | (c4[:, 2] << 4) | (c4[:, 3] << 6)).astype(np.uint8)
    lens = code.astype(np.int64) + 1
    pos = np.cumsum(lens) - lens
    data = np.empty(int(lens.sum()), dtype=np.uint8)
    for k in range(4):
        m = code >= k
        data[pos[m] + k] = ((u[m] >> (8 * k)) & 0xFF).astype(np.uint8)
    return keys.tobytes() + data.tobytes()


def svb32_decode(buf, n: int) -> Tuple[np.ndarray, int]:
    """-> (uint32 [n], bytes consumed)."""
    b = np.frombuffer(buf, dtype=np.uint8)
    nk = (n + 3) // 4
    k = b[:nk]
    code = np.stack([k & 3, (k >> 2) & 3, (k >> 4) & 3, (k >> 6) & 3], axis=1).reshape(-1)[:n]
    lens = code.astype(np.int64) + 1
    pos = np.cumsum(lens) - lens
    total = int(lens.sum())
    data = b[nk:nk + total]
    if data.size != total:
        raise ValueError("streamvbyte stream is truncated")
    u = np.zeros(n, dtype=np.uint32)
    for j in range(4):
        m = code >= j
        u[m] |= data[pos[m] + j].astype(np.uint32) << (8 * j)
    return u, nk + total


# ------------------------------------------------------------------------------------------------ the two signal codecs
def svb_zd_compress(samples: np.ndarray) -> bytes:
    """slow5 svb-zd blob of one read: uint32 n, then the StreamVByte stream of the zig-zag deltas."""
    s = np.ascontiguousarray(samples, dtype=np.int16)
    return struct.pack("<I", s.size) + svb32_encode(zigzag_delta32(s))


def svb_zd_decompress(buf) -> Tuple[np.ndarray, int]:
    """-> (int16 samples, bytes consumed)."""
    (n,) = struct.unpack_from("<I", buf, 0)
    u, used = svb32_decode(memoryview(buf)[4:], n)
    return unzigzag_delta32(u), 4 + used


def vbz_compress(samples: np.ndarray, level: int = 1) -> bytes:
    """pod5 VBZ blob of one signal-table row (the sample count travels in the row's `samples` column)."""
    return zstd_compress(svb16_encode(zigzag_delta16(samples)), level)


def vbz_decompress(buf, n: int) -> np.ndarray:
    return unzigzag_delta16(svb16_decode(_vbz_inflate(buf, n), n))


def _vbz_inflate(buf, n: int) -> bytes:
    """The svb16 stream's length is data dependent: read it from the zstd frame header (ZSTD_compress always writes it)."""
    size = zstd_frame_content_size(buf)
    if size is None or size > (n + 7) // 8 + 2 * n:
        raise ValueError("VBZ row: implausible zstd frame")
    return zstd_decompress(buf, size)


def zstd_frame_content_size(buf):
    """Frame_Content_Size of a zstd frame (RFC 8878 section 3.1.1.1), None when the frame does not carry it."""
    b = bytes(buf[:18])
    if len(b) < 6 or b[:4] != b"\x28\xb5\x2f\xfd":
        raise ValueError("not a zstd frame")
    fhd = b[4]
    fcs_flag, single, did_flag = fhd >> 6, (fhd >> 5) & 1, fhd & 3
    pos = 5 + (0 if single else 1) + (0, 1, 2, 4)[did_flag]
    if fcs_flag == 0:
        return b[pos] if single else None
    if fcs_flag == 1:
        return struct.unpack_from("<H", b, pos)[0] + 256
    if fcs_flag == 2:
        return struct.unpack_from("<I", b, pos)[0]
    return struct.unpack_from("<Q", b, pos)[0]
