"""The reference's sub-module call surface (modules.py: Encoder 65-89, NoiseSampler 275-278, LengthRegulator 396-441,
Decoder 133-142) on top of the fused HIP kernel, for stage-by-stage parity work.

The kernel computes the whole path from the bases in one launch.  `Stages(engine)` hands out four callables with the reference's
signatures that share one context: `encoder(src_seq)` runs the launch with the stage outputs switched on and remembers them;
fed what the previous stage returned -- the way `predict_step` chains them (model.py:197-221) -- `noise_sampler`,
`length_regulator` and `decoder` hand out the tensors of that same launch.  Fed ANY OTHER tensor they are stand-alone
operators: the test instance of the kernel takes a stage's input from memory (s2s_debug.emb_in: the heads, the dwell source
and the encoder blocks on a caller's emb_out; s2s_debug.dec_in: the decoder blocks and output projection on a caller's
[B,250,64]), one launch per call, and the length regulator's expansion is index arithmetic on the caller's tensors.
"""
from typing import Optional

import torch

from .engine import Engine, PredictParams
from .model import onehot_to_bases


class _Context:
    def __init__(self, engine: Engine):
        self.engine = engine
        self.out = None                # stage tensors of the last encoder() call
        self.lr_out = None             # the length-regulated tensor of that batch, once length_regulator() has built it
        self.params = None
        self.inject = {}


def _same(a: torch.Tensor, b: Optional[torch.Tensor]) -> bool:
    """Is `a` the tensor the previous stage of this context returned (then the remembered launch is continued)?"""
    return b is not None and (a is b or (a.shape == b.shape and (a.data_ptr() == b.data_ptr() or torch.equal(a, b))))


def _dummy_bases(eng: Engine, B: int):
    nb = 16 + eng.k - 1
    return (torch.full((B, nb), ord("A"), dtype=torch.uint8, device=eng.device),
            torch.full((B,), 16, dtype=torch.uint8, device=eng.device))


def _f32(x: torch.Tensor, eng: Engine, shape) -> torch.Tensor:
    if tuple(x.shape) != tuple(shape):
        raise ValueError(f"expected a tensor of shape {tuple(shape)}, got {tuple(x.shape)}")
    return x.to(device=eng.device, dtype=torch.float32).contiguous()


class Encoder:
    """modules.py:22-89.  forward(src_seq [B,16,5k] or [B,16,k,5] one-hot) -> (enc_out, emb_out), each [B,16,64]."""

    def __init__(self, ctx: _Context):
        self._ctx = ctx

    def forward(self, src_seq: torch.Tensor, return_attns: bool = False, mask=None):
        if return_attns or mask is not None:
            raise NotImplementedError("attention maps / masks are not part of the predict path (model.py:199)")
        ctx, eng = self._ctx, self._ctx.engine
        onehot = src_seq.reshape(src_seq.shape[0], 16, eng.k, 5)
        bases, n_valid = onehot_to_bases(onehot.to(eng.device))
        ctx.out = eng.predict_chunks(bases.contiguous(), n_valid.contiguous(), ctx.params, debug=True, **ctx.inject)
        return ctx.out["enc_out"], ctx.out["emb_out"]

    __call__ = forward


class NoiseSampler:
    """modules.py:259-278.  forward(emb_out) -> sigma [B,16,1] (scaled units)."""

    def __init__(self, ctx: _Context):
        self._ctx = ctx

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ctx = self._ctx
        if ctx.out is not None and _same(x, ctx.out["emb_out"]):
            return ctx.out["sigma"].unsqueeze(-1)
        eng = ctx.engine                                                         # stand-alone: any emb_out [B,16,64]
        x = _f32(x, eng, (x.shape[0], 16, 64))
        out = eng.predict_chunks(*_dummy_bases(eng, x.shape[0]), ctx.params, debug=True, emb_in=x, **ctx.inject)
        return out["sigma"].unsqueeze(-1)

    __call__ = forward


class LengthRegulator:
    """modules.py:344-441.  forward(emb_out, x, noise_std_prediction, ...) -> (out [B,250,64], dur_float [B,16], dist,
    noise_ext [B,250,1], None).  The dwell source is the one the context's PredictParams select (the keyword arguments of
    the reference call are checked against them); `out` is assembled here from enc_out and the kernel's dwell counts -- the
    kernel itself never materialises it -- and `dist` is the Gamma of the kernel's conc / rate."""

    def __init__(self, ctx: _Context):
        self._ctx = ctx

    def forward(self, emb_out, x, noise_std_prediction, alpha: float = 1.0, target=None, max_length: Optional[int] = None,
                dwell_mean: Optional[float] = None, dwell_std: Optional[float] = None,
                duration_sampling: Optional[bool] = None, min_length: Optional[float] = None):
        ctx = self._ctx
        o, p = ctx.out, ctx.params
        if target is not None or alpha != 1.0:
            raise NotImplementedError("teacher-forced durations / alpha belong to training (modules.py:399-411)")
        for name, given, have in (("dwell_mean", dwell_mean, p.dwell_mean), ("dwell_std", dwell_std, p.dwell_std),
                                  ("duration_sampling", duration_sampling, p.duration_sampling),
                                  ("min_length", min_length, p.min_duration), ("max_length", max_length, 250)):
            if given is not None and float(given) != float(have):
                raise ValueError(f"LengthRegulator: {name}={given} differs from the context's PredictParams ({have})")
        chained = o is not None and _same(emb_out, o["emb_out"])
        if not chained:                              # stand-alone: the dwell source on the caller's emb_out (one launch), then pure indexing
            eng = ctx.engine
            e = _f32(emb_out, eng, (emb_out.shape[0], 16, 64))
            o = eng.predict_chunks(*_dummy_bases(eng, e.shape[0]), p, debug=True, emb_in=e, **ctx.inject)
        x_src = o["enc_out"] if chained and _same(x, o["enc_out"]) else _f32(x, ctx.engine, (o["dur"].shape[0], 16, 64))
        sig = noise_std_prediction
        sig_src = (o["sigma"] if chained and _same(sig, o["sigma"].unsqueeze(-1))
                   else _f32(sig.reshape(sig.shape[0], 16), ctx.engine, (o["dur"].shape[0], 16)))
        dur = o["dur"].long()
        cum = dur.cumsum(1)                                                     # modules.py:368
        t = torch.arange(250, device=dur.device).view(1, 250, 1)
        idx = (cum.unsqueeze(1) <= t).sum(-1)                                   # [B,250]: 16 = past the last dwell
        live = (idx < 16).unsqueeze(-1)
        row = idx.clamp(max=15)
        out = torch.where(live, x_src.gather(1, row.unsqueeze(-1).expand(-1, -1, 64)), torch.zeros((), device=dur.device))
        noise_ext = torch.where(live, sig_src.gather(1, row).unsqueeze(-1), torch.zeros((), device=dur.device))
        dist = torch.distributions.Gamma(o["conc"], o["rate"]) if p.duration_sampling else None
        ctx.lr_out = out if chained and x_src is o["enc_out"] else None
        return out, o["dur"].float(), dist, noise_ext, None

    __call__ = forward


class Decoder:
    """modules.py:97-142.  forward(enc_seq [B,250,64]) -> [B,250,1] (scaled units, after the ReLU)."""

    def __init__(self, ctx: _Context):
        self._ctx = ctx

    def forward(self, enc_seq: torch.Tensor, mask=None) -> torch.Tensor:
        if mask is not None:
            raise NotImplementedError("the predict path runs the decoder unmasked (model.py:217)")
        ctx = self._ctx
        if ctx.out is not None and _same(enc_seq, getattr(ctx, "lr_out", None)):
            return ctx.out["y_scaled"].unsqueeze(-1)
        eng = ctx.engine                                                         # stand-alone: any [B,250,64]; position_enc is added
        x = _f32(enc_seq, eng, (enc_seq.shape[0], 250, 64)) + eng.decoder_position_enc()   # inside the operator (modules.py:136)
        params = PredictParams(**{**ctx.params.__dict__, "noise_std": 0.0})
        out = eng.predict_chunks(*_dummy_bases(eng, x.shape[0]), params, debug=True, dec_in=x.contiguous())
        return out["y_scaled"].unsqueeze(-1)

    __call__ = forward


class Stages:
    """encoder / noise_sampler / length_regulator / decoder of one engine, chained like predict_step (model.py:197-221)."""

    def __init__(self, engine: Engine, params: Optional[PredictParams] = None, inject_g: Optional[torch.Tensor] = None,
                 inject_zdw: Optional[torch.Tensor] = None):
        self._ctx = _Context(engine)
        self._ctx.params = params or PredictParams(noise_std=0.0)
        self._ctx.inject = {k: v for k, v in (("inject_g", inject_g), ("inject_zdw", inject_zdw)) if v is not None}
        self.encoder = Encoder(self._ctx)
        self.noise_sampler = NoiseSampler(self._ctx)
        self.length_regulator = LengthRegulator(self._ctx)
        self.decoder = Decoder(self._ctx)
