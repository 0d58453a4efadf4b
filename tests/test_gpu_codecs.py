"""GPU StreamVByte encoders (s2s_svb_encode) against the numpy codecs, which are pinned by hand-computed vectors
(tests/test_codecs_cpu.py): byte-for-byte equal blobs for whole reads (slow5 svb-zd) and for POD5 signal-table rows."""
import numpy as np
import pytest
import torch

import seq2squiggle_amd as S
from seq2squiggle_amd import codecs as C
from conftest import load_ckpt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    sd, cfg = load_ckpt("k9")
    e = S.Engine(sd, cfg)
    yield e
    e.close()


def _reads(seed, lens):
    rng = np.random.default_rng(seed)
    out = []
    for i, n in enumerate(lens):
        if i % 3 == 0:
            out.append(rng.integers(-32768, 32768, n).astype(np.int16))                 # every byte length, wrapping deltas
        else:
            out.append((np.cumsum(rng.integers(-40, 41, n)) + 700).astype(np.int16))     # signal-like
    return out


def test_known_answer_vectors_on_the_gpu(eng):
    import json, os
    from conftest import GOLDEN
    kat = json.load(open(os.path.join(GOLDEN, "codec_kat.json")))
    for variant, key in ((32, "svb_zd"), (16, "svb16_zd")):
        cases = [c for c in kat[key] if c["samples"]]
        flat = np.concatenate([np.array(c["samples"], np.int16) for c in cases])
        offs = np.concatenate([[0], np.cumsum([len(c["samples"]) for c in cases])]).astype(np.int64)
        r = eng.svb_encode(torch.from_numpy(flat).cuda(), torch.from_numpy(offs).cuda(),
                           torch.arange(len(cases), dtype=torch.int32).cuda(), torch.zeros(len(cases), dtype=torch.int32).cuda(),
                           1 << 40, variant, int(offs[-1]))
        o, out = r["offsets"].cpu().numpy(), r["out"].cpu().numpy()
        for i, c in enumerate(cases):
            assert out[o[i]:o[i + 1]].tobytes().hex() == c["hex"], (key, c["samples"])


def test_whole_reads_svb_zd(eng):
    lens = [1, 2, 3, 4, 5, 7, 8, 9, 2047, 2048, 2049, 60000, 131, 250000, 4097]
    reads = _reads(1, lens)
    flat = np.concatenate(reads)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    R = len(lens)
    r = eng.svb_encode(torch.from_numpy(flat).cuda(), torch.from_numpy(offs).cuda(), torch.arange(R, dtype=torch.int32).cuda(),
                       torch.zeros(R, dtype=torch.int32).cuda(), 1 << 40, 32, int(offs[-1]))
    o, out = r["offsets"].cpu().numpy(), r["out"].cpu().numpy()
    assert o[0] == 0 and (np.diff(o) > 0).all()
    for i, x in enumerate(reads):
        assert out[o[i]:o[i + 1]].tobytes() == C.svb_zd_compress(x), i


def test_pod5_rows_svb16(eng):
    """Rows of 102,400 samples; candidate rows are enumerated from an upper bound of the read length, so some are empty."""
    CH = 102400
    lens = [5, CH, CH + 1, 3 * CH - 1, 1234, 2 * CH]
    reads = _reads(2, lens)
    flat = np.concatenate(reads)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rr, ri = [], []
    for i, n in enumerate(lens):
        for k in range(-(-(n + 70000) // CH)):          # bound = real length + slack: trailing candidates do not exist
            rr.append(i); ri.append(k)
    N = len(rr)
    r = eng.svb_encode(torch.from_numpy(flat).cuda(), torch.from_numpy(offs).cuda(), torch.tensor(rr, dtype=torch.int32).cuda(),
                       torch.tensor(ri, dtype=torch.int32).cuda(), CH, 16, int(offs[-1]) + 70000 * len(lens))
    o, out = r["offsets"].cpu().numpy(), r["out"].cpu().numpy()
    seen = 0
    for j in range(N):
        x = reads[rr[j]][ri[j] * CH:(ri[j] + 1) * CH]
        blob = out[o[j]:o[j + 1]].tobytes()
        if len(x) == 0:
            assert blob == b""
            continue
        seen += 1
        assert blob == C.svb16_encode(C.zigzag_delta16(x)), j
        assert np.array_equal(C.unzigzag_delta16(C.svb16_decode(blob, len(x))), x)
    assert seen == sum(-(-n // CH) for n in lens) and seen < N


def test_bad_arguments(eng):
    z = torch.zeros(4, dtype=torch.int16).cuda()
    o = torch.tensor([0, 4], dtype=torch.int64).cuda()
    i0 = torch.zeros(1, dtype=torch.int32).cuda()
    with pytest.raises(ValueError):
        eng.svb_encode(z, o, i0, i0, 100, 8, 4)
    with pytest.raises(ValueError):
        eng.svb_encode(z.float(), o, i0, i0, 100, 16, 4)
    r = eng.svb_encode(z, o, i0[:0], i0[:0], 100, 16, 0)         # no rows: a single zero offset
    assert r["offsets"].cpu().tolist() == [0]


def test_outputs_into_one_caller_buffer(eng):
    """The streaming path's layout: read offsets, row offsets and the coded signal are views of ONE device buffer (it leaves
    the GPU as a single DMA copy).  Same bytes as with library-allocated outputs; undersized or mistyped views are refused."""
    rng = np.random.default_rng(4)
    B = 40
    sig = torch.from_numpy(np.where(rng.random((B, 250)) < 0.1, 0.0, rng.normal(90, 15, (B, 250))).astype(np.float32)).cuda()
    first = torch.tensor([0, 7, 7, 25, 40], dtype=torch.int32).cuda()            # four reads, the second one empty
    R = 4
    ref = eng.export_reads(sig, first, 2048.0, 281.345551, -127.5655735, want_pa=False, want_dac=True)
    rr = torch.tensor([0, 1, 2, 3], dtype=torch.int32).cuda()
    ri = torch.zeros(4, dtype=torch.int32).cuda()
    ref_svb = eng.svb_encode(ref["dac"], ref["offsets"], rr, ri, 1 << 40, 32, B * 250)
    cap = eng.svb_capacity(B * 250, 4, 32)
    head = 8 * (R + 1) + 8 * 5
    head += -head % 16
    buf = torch.zeros(head + cap, dtype=torch.uint8, device="cuda")
    offs_v, rows_v = buf[:8 * (R + 1)].view(torch.int64), buf[8 * (R + 1): 8 * (R + 1) + 40].view(torch.int64)
    ex = eng.export_reads(sig, first, 2048.0, 281.345551, -127.5655735, want_pa=False, want_dac=True, out_offsets=offs_v)
    got = eng.svb_encode(ex["dac"], ex["offsets"], rr, ri, 1 << 40, 32, B * 250, out=buf[head:], out_offsets=rows_v)
    assert got["out"].data_ptr() == buf[head:].data_ptr() and ex["offsets"].data_ptr() == buf.data_ptr()
    host = buf.cpu().numpy()
    n = int(ref_svb["offsets"][-1])
    assert np.array_equal(host[:8 * (R + 1)].view(np.int64), ref["offsets"].cpu().numpy())
    assert np.array_equal(host[8 * (R + 1): 8 * (R + 1) + 40].view(np.int64), ref_svb["offsets"].cpu().numpy())
    assert np.array_equal(host[head: head + n], ref_svb["out"][:n].cpu().numpy())
    # int16 samples straight into the buffer (the BLOW5 path)
    buf2 = torch.zeros(48 + 2 * B * 250, dtype=torch.uint8, device="cuda")
    ex2 = eng.export_reads(sig, first, 2048.0, 281.345551, -127.5655735, want_pa=False, want_dac=True,
                           out_offsets=buf2[:40].view(torch.int64), out_dac=buf2[48:].view(torch.int16))
    m = int(ref["offsets"][-1])
    assert torch.equal(ex2["dac"][:m], ref["dac"][:m]) and ex2["dac"].data_ptr() == buf2[48:].data_ptr()
    with pytest.raises(ValueError):
        eng.export_reads(sig, first, 2048.0, 281.345551, -127.5655735, want_pa=False, out_offsets=buf2[:32].view(torch.int64))
    with pytest.raises(ValueError):
        eng.export_reads(sig, first, 2048.0, 281.345551, -127.5655735, want_pa=False, out_dac=buf2[48:1000].view(torch.int16))
    with pytest.raises(ValueError):
        eng.svb_encode(ex["dac"], ex["offsets"], rr, ri, 1 << 40, 32, B * 250, out=buf[head: head + 100])


def test_worst_case_rows_fill_the_stated_capacity_and_overflow_is_reported(eng):
    """Alternating -32768 / 32767 samples: every zig-zag delta of the 32-bit variant needs 3 bytes, so a row of n samples takes
    4 + ceil(n/4) + 3n bytes -- exactly the bound include/s2s_hip.h states (round 2's bound forgot the control bytes).  The blobs
    must fit Engine.svb_capacity, equal the numpy codec, and a buffer that is too small must come back with a NEGATIVE total
    (minus the bytes needed) instead of silently skipped rows."""
    lens = [1, 4, 5, 4099, 70001]
    reads = []
    for n in lens:
        x = np.where(np.arange(n) % 2 == 0, -32768, 32767).astype(np.int16)
        reads.append(x)
    flat = np.concatenate(reads)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    R, total = len(lens), int(offs[-1])
    args = (torch.from_numpy(flat).cuda(), torch.from_numpy(offs).cuda(), torch.arange(R, dtype=torch.int32).cuda(),
            torch.zeros(R, dtype=torch.int32).cuda(), 1 << 40, 32, total)
    r = eng.svb_encode(*args)
    o, out = r["offsets"].cpu().numpy(), r["out"].cpu().numpy()
    need = sum(4 + -(-n // 4) + 3 * n - 1 for n in lens)          # (only a row's first delta, against 0, takes 2 bytes: -32768 -> 65535)
    assert o[-1] == need <= eng.svb_capacity(total, R, 32)
    assert eng.svb_capacity(total, R, 32) - need < 4 * R + 8                        # the bound is tight
    for i, x in enumerate(reads):
        assert out[o[i]:o[i + 1]].tobytes() == C.svb_zd_compress(x), i
    # 16-bit variant on the same data: 2 bytes per value
    r16 = eng.svb_encode(*args[:4], 1 << 40, 16, total)
    o16 = r16["offsets"].cpu().numpy()
    assert o16[-1] <= eng.svb_capacity(total, R, 16)
    # a too-small buffer: flagged, rows that fit are still exact
    small = torch.zeros(need - 1000, dtype=torch.uint8, device="cuda")
    with pytest.raises(ValueError):
        eng.svb_encode(*args, out=small)                                            # the engine refuses views below its bound ...
    L = S._lib.lib()
    import ctypes as Ct
    offs_d = torch.zeros(R + 1, dtype=torch.int64, device="cuda")
    rc = L.s2s_svb_encode(eng._h, None, Ct.c_void_p(args[0].data_ptr()), Ct.c_void_p(args[1].data_ptr()), Ct.c_void_p(args[2].data_ptr()),
                          Ct.c_void_p(args[3].data_ptr()), R, 1 << 40, 32, Ct.c_void_p(small.data_ptr()), small.numel(),
                          Ct.c_void_p(offs_d.data_ptr()))                           # ... the C ABI reports it
    torch.cuda.synchronize()
    assert rc == 0 and int(offs_d[-1]) == -need
    got = small.cpu().numpy()
    oo = offs_d.cpu().numpy()
    for i in range(R - 1):                                                          # every row but the last fits
        assert got[oo[i]:oo[i + 1]].tobytes() == C.svb_zd_compress(reads[i]), i
