"""CPU oracle for the seq2squiggle *predict* hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``seq2squiggle_amd/`` may import this
module: it is the checker for the HIP path (``tests/``, ``__graft_entry__.smoke``)
and the ``cpu_baseline`` leg of ``bench.py`` -- never the product path.

It is a from-the-spec restatement of the reference
algorithm in eager CPU PyTorch ops, the same aten kernels the reference issues,
working on the raw ``state_dict`` tensors instead of ``nn.Module`` objects.
Every function cites the reference file:line it follows (paths relative to the
reference repository, ``src/seq2squiggle/``).

Parity status: PINNED.  ``tools/make_goldens.py`` imports the real reference in
the build container, runs it with injected random variates and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file against
those vectors (the reference itself ships no tests or known-answer vectors,
CONTRIBUTING.md:36-38).

Random variates: the reference draws from torch's CPU mt19937 stream
(Gamma.sample -> torch._standard_gamma, torch.normal).  That stream cannot be
mirrored on another engine, so parity is defined with *injected* variates:
``inject_g`` is the value of ``Gamma(conc, rate).sample()`` before any clamp and
``inject_z01`` are standard normals (``torch.normal(0, std)`` == ``z01 * std``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

ALPHABET = "_ACGT"          # utils.py:74  letter_to_int
CODE_UNKNOWN = 255          # a letter outside the alphabet -> all-zero one-hot row (utils.py:86)
T_ENC = 16                  # config.yaml:18 max_dna_len
T_DEC = 250                 # config.yaml:19 max_signal_len
FP32_TINY = float(np.finfo(np.float32).tiny)


@dataclass
class PredictParams:
    """Scalars held by the reference LightningModule (model.py:55-63)."""
    dwell_mean: float = 12.5
    dwell_std: float = 0.0
    noise_std: float = 2.0
    noise_sampling: bool = True
    duration_sampling: bool = True
    min_noise: float = 0.0
    min_duration: float = 3.0
    scaling_max_value: float = 165.0


# --------------------------------------------------------------------------- chunk encoder
def encode_read(seq: str, k: int, max_dna_len: int = T_ENC) -> np.ndarray:
    """split_sequence (utils.py:350-356) as integer codes.

    extract_kmers (utils.py:334-339): the L-k+1 overlapping windows;
    add_remainder (utils.py:342-347): whole "_"*k k-mers up to a multiple of 16;
    one_hot_encode (utils.py:56-89): letter -> index, unknown letter -> zero row;
    regular_break_points (utils.py:266-287): consecutive groups of 16.
    Returns uint8 [C, 16, k] with values 0..4 or CODE_UNKNOWN.  A read shorter
    than k yields C == 0 (dataloader.py:393-398 skips it).
    """
    n_kmer = len(seq) - k + 1
    if n_kmer <= 0:
        return np.zeros((0, max_dna_len, k), dtype=np.uint8)
    lut = np.full(256, CODE_UNKNOWN, dtype=np.uint8)
    for i, ch in enumerate(ALPHABET):
        lut[ord(ch)] = i
    b = lut[np.frombuffer(seq.encode("latin-1"), dtype=np.uint8)]
    idx = np.arange(n_kmer)[:, None] + np.arange(k)[None, :]
    kmers = b[idx]                                             # [n_kmer, k]
    remain = max_dna_len - (n_kmer % max_dna_len)
    if remain % max_dna_len > 0:
        kmers = np.concatenate([kmers, np.zeros((remain, k), dtype=np.uint8)], 0)
    return kmers.reshape(-1, max_dna_len, k)


def one_hot(codes: np.ndarray, dtype=torch.float32) -> torch.Tensor:
    """codes [B,16,k] -> [B,16,5k] (model.py:197-198 reshape of the fp16 [B,16,k,5] batch)."""
    c = torch.as_tensor(np.ascontiguousarray(codes)).long()
    oh = torch.zeros(*c.shape, 5, dtype=dtype)
    valid = c < 5
    oh.scatter_(-1, c.clamp(max=4).unsqueeze(-1), valid.unsqueeze(-1).to(dtype))
    return oh.reshape(c.shape[0], c.shape[1], -1)


# --------------------------------------------------------------------------- layers.py
def sinusoid_table(n_position: int, d_hid: int) -> torch.Tensor:
    """get_sinusoid_encoding_table (layers.py:145-165): python-float64 angles -> fp32 -> sin/cos in fp32."""
    tab = torch.tensor([[pos / 10000 ** (2 * (j // 2) / d_hid) for j in range(d_hid)]
                        for pos in range(n_position)])
    tab[:, 0::2] = torch.sin(tab[:, 0::2])
    tab[:, 1::2] = torch.cos(tab[:, 1::2])
    return tab.float()


def mha(sd: Dict[str, torch.Tensor], p: str, x: torch.Tensor, n_head: int) -> torch.Tensor:
    """MultiHeadAttention.forward (layers.py:64-88) + ScaledDotProductAttention (layers.py:19-41), eval, mask=None."""
    B, T, D = x.shape
    d_k = D // n_head
    q = F.linear(x, sd[p + "w_qs.weight"], sd[p + "w_qs.bias"]).view(B, T, n_head, d_k)
    k = F.linear(x, sd[p + "w_ks.weight"], sd[p + "w_ks.bias"]).view(B, T, n_head, d_k)
    v = F.linear(x, sd[p + "w_vs.weight"], sd[p + "w_vs.bias"]).view(B, T, n_head, d_k)
    q = q.permute(2, 0, 1, 3).contiguous().view(-1, T, d_k)
    k = k.permute(2, 0, 1, 3).contiguous().view(-1, T, d_k)
    v = v.permute(2, 0, 1, 3).contiguous().view(-1, T, d_k)
    attn = torch.bmm(q, k.transpose(1, 2)) / (d_k ** 0.5)          # layers.py:20-21, temperature layers.py:58
    attn = torch.softmax(attn, dim=2)                             # layers.py:39
    out = torch.bmm(attn, v)                                      # layers.py:40
    out = out.view(n_head, B, T, d_k).permute(1, 2, 0, 3).contiguous().view(B, T, -1)
    out = F.linear(out, sd[p + "fc.weight"], sd[p + "fc.bias"])   # layers.py:85
    return F.layer_norm(out + x, (D,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)


def ffn(sd, p: str, x: torch.Tensor) -> torch.Tensor:
    """PositionwiseFeedForward.forward (layers.py:108-113), eval."""
    h = F.linear(torch.relu(F.linear(x, sd[p + "w_1.weight"], sd[p + "w_1.bias"])),
                 sd[p + "w_2.weight"], sd[p + "w_2.bias"])
    return F.layer_norm(h + x, (x.shape[-1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)


def fft_block(sd, p: str, x: torch.Tensor, n_head: int) -> torch.Tensor:
    """FFTBlock.forward (layers.py:135-142)."""
    return ffn(sd, p + "pos_ffn.", mha(sd, p + "slf_attn.", x, n_head))


# --------------------------------------------------------------------------- modules.py
def encoder(sd, cfg, x: torch.Tensor):
    """Encoder.forward (modules.py:65-89) -> (enc_out, emb_out)."""
    s = torch.relu(F.linear(x, sd["encoders.src_emb.weight"], sd["encoders.src_emb.bias"]))
    for i in range(cfg["pre_layers"]):
        s = torch.relu(F.linear(s, sd[f"encoders.pre_net_stack.{i}.weight"], sd[f"encoders.pre_net_stack.{i}.bias"]))
    e = s + sd["encoders.position_enc"][0]                         # modules.py:80 (slice is a no-op)
    for l in range(cfg["encoder_layers"]):
        e = fft_block(sd, f"encoders.layer_stack.{l}.", e, cfg["encoder_heads"])
    return e, s


def _mlp_softplus(sd, p: str, x: torch.Tensor) -> torch.Tensor:
    h = torch.relu(F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"]))
    return F.softplus(F.linear(h, sd[p + "3.weight"], sd[p + "3.bias"])).flatten(1)


def noise_sampler(sd, emb_out):
    """NoiseSampler.forward (modules.py:275-278) -> sigma [B,16] (scaled units)."""
    return _mlp_softplus(sd, "noise_sampler.stdv_layer.", emb_out)


def duration_params(sd, emb_out):
    """DurationSampler.forward up to the distribution (modules.py:213-219) -> conc, rate [B,16]."""
    conc = _mlp_softplus(sd, "length_regulator.duration_sampler.conc_layer.", emb_out).clamp(min=1e-8)
    rate = _mlp_softplus(sd, "length_regulator.duration_sampler.rate_layer.", emb_out).clamp(min=1e-8)
    return conc, rate


def standard_gamma_to_sample(sg: torch.Tensor, rate: torch.Tensor) -> torch.Tensor:
    """torch.distributions.Gamma.sample: _standard_gamma(conc) / rate, clamp_(min=tiny)."""
    return (sg / rate).clamp(min=FP32_TINY)


def durations(params: PredictParams, B: int, g: Optional[torch.Tensor] = None,
              zdw: Optional[torch.Tensor] = None, dtype=torch.float32) -> torch.Tensor:
    """LengthRegulator.forward duration source (modules.py:396-438) -> int32 [B,16].

    g   : Gamma sample value (pre clamp) when duration_sampling;
    zdw : standard normals for the dwell_std > 0 mode (torch.normal(mean, std) == mean + z*std).
    """
    if params.duration_sampling:
        d = g.to(dtype).clamp(min=1.0)                             # modules.py:223
        d = d.clamp(min=params.min_duration)                       # modules.py:414-416
    elif params.dwell_std <= 0:
        d = torch.full((B, T_ENC), params.dwell_mean, dtype=dtype)  # modules.py:420-423
    else:
        d = torch.full((B, T_ENC), params.dwell_mean, dtype=dtype) + zdw.to(dtype) * params.dwell_std
        d = d.clamp(min=params.min_duration)                       # modules.py:425-432
    return torch.round(d).int()                                    # modules.py:437-438 (half-to-even)


def length_regulate(x: torch.Tensor, sigma: torch.Tensor, dur: torch.Tensor, max_len: int = T_DEC):
    """LengthRegulator.LR (modules.py:344-392) as the gather it is.

    M[b,j,t] = (t < cum[b,j]) - (t < cum[b,j-1]) selects x[b, i(t)] with
    i(t) = #{j : cum[b,j] <= t}; rows t >= cum[b,15] are zero; F.pad with a
    negative amount crops at max_len (modules.py:386).
    """
    B, Te, D = x.shape
    cum = torch.cumsum(dur.long(), dim=1)                          # modules.py:368
    t = torch.arange(max_len)
    idx = (cum.unsqueeze(2) <= t.view(1, 1, -1)).sum(dim=1)        # [B, max_len]
    live = idx < Te
    idxc = idx.clamp(max=Te - 1)
    out = torch.gather(x, 1, idxc.unsqueeze(-1).expand(B, max_len, D)) * live.unsqueeze(-1).to(x.dtype)
    sx = torch.gather(sigma, 1, idxc) * live.to(sigma.dtype)
    return out, sx


def decoder(sd, cfg, h: torch.Tensor) -> torch.Tensor:
    """Decoder.forward (modules.py:133-142) -> [B,250] scaled units."""
    h = h + sd["decoders.position_enc"][0]                         # modules.py:136
    for l in range(cfg["decoder_layers"]):
        h = fft_block(sd, f"decoders.layer_stack_FFT.{l}.", h, cfg["decoder_heads"])
    return torch.relu(F.linear(h, sd["decoders.out_linear.weight"], sd["decoders.out_linear.bias"])).squeeze(-1)


# --------------------------------------------------------------------------- model.py
def finish(y_scaled: torch.Tensor, sigma_ext: torch.Tensor, z01: Optional[torch.Tensor], params: PredictParams):
    """scale + noise + clamp (model.py:221-240)."""
    y = y_scaled * params.scaling_max_value
    if params.noise_std > 0:
        nz = y != 0
        if params.noise_sampling:
            sd_ = sigma_ext.clamp(min=params.min_noise) * params.noise_std * params.scaling_max_value
            gen = z01.to(y.dtype) * sd_                            # torch.normal(mean=0, std=sd_)
        else:
            gen = z01.to(y.dtype) * params.noise_std               # torch.normal(0, noise_std, size)
        y = torch.where(nz, y + gen, y)
    return y.clamp(min=0)


def predict_chunks(sd: Dict[str, torch.Tensor], cfg: dict, codes: np.ndarray, params: PredictParams,
                   inject_g: Optional[torch.Tensor] = None, inject_z01: Optional[torch.Tensor] = None,
                   inject_zdw: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
                   dtype=torch.float32, stages: bool = False):
    """predict_step (model.py:195-250) for a batch of chunks given as codes [B,16,k].

    Returns dict with at least ``signal`` [B,250] (pA, fp) and ``dur`` [B,16] int32.
    With ``inject_*`` None the oracle draws from torch's own generators (statistical use only).
    """
    if dtype != torch.float32:
        sd = {k: v.to(dtype) for k, v in sd.items()}
    B = codes.shape[0]
    x = one_hot(codes, dtype)
    enc_out, emb_out = encoder(sd, cfg, x)
    sigma = noise_sampler(sd, emb_out)
    conc = rate = None
    g = inject_g
    if params.duration_sampling:
        conc, rate = duration_params(sd, emb_out)
        if g is None:
            sg = torch._standard_gamma(conc, generator=generator)
            g = standard_gamma_to_sample(sg, rate)
    zdw = inject_zdw
    if (not params.duration_sampling) and params.dwell_std > 0 and zdw is None:
        zdw = torch.randn(B, T_ENC, generator=generator, dtype=dtype)
    dur = durations(params, B, g, zdw, dtype)
    h, sigma_ext = length_regulate(enc_out, sigma, dur, cfg["max_signal_len"])
    y_scaled = decoder(sd, cfg, h)
    z01 = inject_z01
    if params.noise_std > 0 and z01 is None:
        z01 = torch.randn(B, cfg["max_signal_len"], generator=generator, dtype=dtype)
    y = finish(y_scaled, sigma_ext, z01, params)
    out = {"signal": y, "dur": dur}
    if stages:
        out.update(emb_out=emb_out, enc_out=enc_out, sigma=sigma, conc=conc, rate=rate,
                   sigma_ext=sigma_ext, y_scaled=y_scaled, lr_out=h)
    return out


# --------------------------------------------------------------------------- export path
def strip_zeros(rows: Sequence[torch.Tensor]) -> torch.Tensor:
    """export_and_clear_results (model.py:284-286): cat the chunk rows of a read, drop every element == 0."""
    cat = torch.cat(list(rows))
    return cat[cat != 0]


def to_dac(signal: np.ndarray, digitisation: float, signal_range: float, offset_mean: float,
           rna: bool = False) -> np.ndarray:
    """pA -> int16 (signal_io.py:134-141): round-half-even of float32 expr, C cast to int16 (wraps)."""
    s = np.asarray(signal, dtype=np.float32)
    # the reference multiplies a float32 array by python floats: float32 arithmetic throughout
    raw = np.round(s * float(digitisation) / float(signal_range) - float(offset_mean))
    assert raw.dtype == np.float32
    raw = raw.astype(np.int64).astype(np.int16)  # wrap like the reference's astype(np.int16) on x86
    if rna:
        raw = np.ascontiguousarray(raw[::-1])
    return raw
