"""TEST INFRASTRUCTURE (like everything under oracle/): a CPU model of the share of attention heads the kernel's FAST softmax path
has to redo, computed from the oracle's decoder scores (reference layers.py:20-40: one unmasked softmax over 250 keys per query).

The fast path (csrc/s2s_device_h.h: softmax_pv32<SAFE = false>) shifts a row's scores by its maximum over the keys of PASS 0 plus
2 log2 units and keeps P = exp2(score - shift) in f16: a row whose largest score beats that pass-0 maximum by more than 18 log2
units overflows, and the (wave = 32 queries, head, layer) run it belongs to is redone.  Which keys pass 0 holds is the key order
of the K / V^T images (att32_key_at): "head" = the blocks of four consecutive keys b = 0 (mod 4) -- what HEAD's fast instance
stores; "first64" = the first 64 keys (rounds 1-3).  tools/redo_model.py prints the table the order was chosen with;
tests/test_gpu_parity.py holds the kernel's redo counter to this model."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import s2s_oracle as O

_KEYS = np.arange(250)
SCHEMES = {
    "first64": _KEYS < 64,
    "head": ((_KEYS >> 2) & 3) == 0,
    "every4th": (_KEYS & 3) == 0,
}
OVERFLOW_LOG2 = 18.0          # f16 max = 2^16 (65504), head-room 2 units


def decoder_scores(sd, cfg, codes, inject_g=None, params=None):
    """Yields per decoder layer the scores [B, head, query, key] in log2 units (what the kernel's score MFMAs produce)."""
    p = params or O.PredictParams(noise_std=0.0)
    out = O.predict_chunks(sd, cfg, codes, p, inject_g=inject_g, stages=True)
    h = out["lr_out"] + sd["decoders.position_enc"][0]
    nh = cfg["n_heads"] if "n_heads" in cfg else 8
    for l in range(cfg["decoder_layers"]):
        pfx = f"decoders.layer_stack_FFT.{l}.slf_attn."
        B, T, D = h.shape
        dk = D // nh
        q = F.linear(h, sd[pfx + "w_qs.weight"], sd[pfx + "w_qs.bias"]).view(B, T, nh, dk).permute(0, 2, 1, 3)
        k = F.linear(h, sd[pfx + "w_ks.weight"], sd[pfx + "w_ks.bias"]).view(B, T, nh, dk).permute(0, 2, 1, 3)
        yield (q @ k.transpose(-1, -2)) / math.sqrt(dk) * math.log2(math.e)
        h = O.fft_block(sd, f"decoders.layer_stack_FFT.{l}.", h, nh)


def predicted_redo_rate(sd, cfg, codes, inject_g=None, scheme="head", params=None) -> float:
    """Share of (wave, head, layer) softmax runs with at least one overflowing row, for pass-0 key set `scheme`."""
    mask = torch.from_numpy(SCHEMES[scheme])
    redone = total = 0
    for s in decoder_scores(sd, cfg, codes, inject_g, params):
        B, H, T, _ = s.shape
        over = (s.max(-1).values - s[..., mask].max(-1).values) > OVERFLOW_LOG2
        runs = F.pad(over, (0, (-T) % 32)).view(B, H, -1, 32).any(-1)          # a wave owns 32 consecutive queries
        redone += int(runs.sum())
        total += runs.numel()
    return redone / total if total else 0.0


def input_families(seed: int = 0, n_reads: int = 4, read_len: int = 320, lambda_fasta: str = None) -> dict:
    """Read sets whose redo share may differ from the calibration launch's pseudo-random chunks: name -> list of sequences.
    random: uniform ACGT; lambda: slices of the example lambda genome (tests/golden/example_lambda_genome.fasta); homopolymer: A.. C..
    G.. T..; dinucleotide: (AC)n (AT)n (CG)n (GA)n; trinucleotide: (CAG)n (GAA)n (CCG)n; n_rich: lambda slices with 30 % of the
    bases replaced by N, and one read of N only."""
    import os
    rng = np.random.default_rng(seed)
    if lambda_fasta is None:
        lambda_fasta = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "example_lambda_genome.fasta")
    genome = "".join(line.strip() for line in open(lambda_fasta) if not line.startswith(">")).upper()
    starts = rng.integers(0, len(genome) - read_len, n_reads)
    lam = [genome[s:s + read_len] for s in starts]
    rep = lambda unit: (unit * (read_len // len(unit) + 1))[:read_len]
    n_rich = []
    for s_ in lam[:max(1, n_reads - 1)]:
        a = np.array(list(s_))
        a[rng.random(len(a)) < 0.3] = "N"
        n_rich.append("".join(a))
    n_rich.append("N" * read_len)
    return {
        "random": ["".join(rng.choice(list("ACGT"), read_len)) for _ in range(n_reads)],
        "lambda": lam,
        "homopolymer": [b * read_len for b in "ACGT"][:max(n_reads, 4)],
        "dinucleotide": [rep(u) for u in ("AC", "AT", "CG", "GA")],
        "trinucleotide": [rep(u) for u in ("CAG", "GAA", "CCG")],
        "n_rich": n_rich,
    }
