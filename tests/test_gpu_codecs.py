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
