"""GPU tests of the built-in Philox samplers (Appendix C).

The reference draws from torch's CPU generator, whose stream no other engine can mirror, so the built-in samplers are
held to the DISTRIBUTIONS of the reference's samplers:
  * Gamma dwell (modules.py:221-222 -> torch._standard_gamma, Marsaglia-Tsang with the alpha < 1 boost): two-sample
    Kolmogorov-Smirnov against torch._standard_gamma at the same (conc, rate), n ~ 1e5, plus a one-sample KS of the
    probability-integral transform against the exact Gamma CDF -- for conc ~ 9 (the committed synthetic checkpoint),
    conc ~ 0.5 and conc ~ 0.13 (checkpoint variants with conc_layer.3.bias lowered, so the alpha < 1 branch runs);
  * Gaussian noise (model.py:224-240): the standard normals the kernel used (debug z01) are N(0,1) by KS, and the noisy
    signal equals clamp(clean + z01 * sd) bit for bit with sd = max(sigma_ext, min_noise) * noise_std * 165, exactly
    `clean` (zero) where the clean signal is zero.
"""
import numpy as np
import pytest
import torch
from scipy import stats

import seq2squiggle_amd as S
from conftest import load_ckpt

pytestmark = pytest.mark.gpu
KS_ALPHA = 1e-3                      # a correct sampler fails one of the KS checks below with probability ~ 1e-3 each


def _chunks(k, n_chunks, seed=0):
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    read = lut[rng.integers(0, 4, 16 * n_chunks + k - 1)].tobytes().decode()
    bases, nv, _ = S.encode_reads([read], k)
    return torch.from_numpy(bases).cuda(), torch.from_numpy(nv).cuda()


@pytest.mark.parametrize("conc_bias,rate_bias", [(None, None), (-0.43, -6.0), (-2.0, -9.0)],
                         ids=["conc9", "conc0.5", "conc0.13"])
def test_gamma_dwell_distribution(conc_bias, rate_bias):
    sd, cfg = load_ckpt("k9")
    sd = dict(sd)
    if conc_bias is not None:        # softplus(-0.43) ~ 0.5, softplus(-2) ~ 0.13: Marsaglia-Tsang's alpha < 1 boost
        sd["length_regulator.duration_sampler.conc_layer.3.bias"] = torch.tensor([conc_bias])
        sd["length_regulator.duration_sampler.rate_layer.3.bias"] = torch.tensor([rate_bias])   # rate ~ 2.5e-3 / 1.2e-4: under half of the draws fall below the clamp at 1
    eng = S.Engine(sd, cfg, mode="f16x3")
    b, n = _chunks(9, 7000)                                             # 112,000 dwell draws
    out = eng.predict_chunks(b, n, S.PredictParams(min_duration=0.0, noise_std=0.0, seed=11), debug=True)
    conc, rate, g = (out[x].cpu().double().flatten() for x in ("conc", "rate", "g"))
    if conc_bias is None:
        assert 5 < conc.mean() < 15
    else:
        assert conc.max() < 1.0 and abs(conc.mean().item() - (0.5 if conc_bias > -1 else 0.127)) < 0.1
    assert (g >= 1.0).all()                                             # modules.py:223 clamp(min=1.0)

    # (1) two-sample KS against the reference's sampler at the same parameters (clamped the same way)
    gen = torch.Generator().manual_seed(5)
    ref = (torch._standard_gamma(conc.float(), generator=gen).double() / rate).clamp(min=1.0)
    ks2 = stats.ks_2samp(g.numpy(), ref.numpy())
    assert ks2.pvalue > KS_ALPHA, ks2

    # (2) one-sample KS of the probability-integral transform against U(0,1): u = F(g; conc, rate) with the censored
    #     draws (g == 1, i.e. Gamma value <= 1) spread uniformly over [0, F(1)]
    F = stats.gamma.cdf(g.numpy(), a=conc.numpy(), scale=1.0 / rate.numpy())
    cens = g.numpy() <= 1.0
    u = np.where(cens, np.random.default_rng(3).random(F.shape) * stats.gamma.cdf(1.0, a=conc.numpy(), scale=1.0 / rate.numpy()), F)
    ks1 = stats.kstest(u, "uniform")
    assert ks1.pvalue > KS_ALPHA, ks1
    assert cens.mean() < 0.5
    # the rounded dwell is round-half-even of g (modules.py:437-438)
    assert np.array_equal(out["dur"].cpu().numpy().flatten(), np.rint(out["g"].cpu().numpy().flatten()).astype(np.int32))
    # different seed / different chunk counter: different draws, same law
    out2 = eng.predict_chunks(b, n, S.PredictParams(min_duration=0.0, noise_std=0.0, seed=12), debug=True)
    assert not torch.equal(out2["g"], out["g"])
    assert stats.ks_2samp(out2["g"].cpu().double().flatten().numpy(), g.numpy()).pvalue > KS_ALPHA
    eng.close()


@pytest.mark.parametrize("tag", ["k9", "k6"])
@pytest.mark.parametrize("noise_sampling,min_noise,noise_std", [(True, 0.0, 2.0), (True, 0.5, 1.5), (False, 0.0, 2.0)])
def test_builtin_noise_is_the_reference_formula(tag, noise_sampling, min_noise, noise_std):
    sd, cfg = load_ckpt(tag)
    eng = S.Engine(sd, cfg, mode="f16x3")
    b, n = _chunks(cfg["seq_kmer"], 800)                                # 200,000 samples
    kw = dict(noise_sampling=noise_sampling, min_noise=min_noise, seed=21)
    noisy = eng.predict_chunks(b, n, S.PredictParams(noise_std=noise_std, **kw), debug=True)
    clean = eng.predict_chunks(b, n, S.PredictParams(noise_std=0.0, **kw), debug=True)
    assert torch.equal(noisy["dur"], clean["dur"])
    y, c, z = noisy["signal"].cpu(), clean["signal"].cpu(), noisy["z01"].cpu()
    # sigma expanded by the length regulator (modules.py:377-388): sample t takes sigma[i(t)], 0 past the last dwell
    dur, sigma = noisy["dur"].cpu().long(), noisy["sigma"].cpu()
    cum = dur.cumsum(1)
    t = torch.arange(250).view(1, 250, 1)
    idx = (cum.view(-1, 1, 16) <= t).sum(-1)                            # [B,250], 16 = past the end
    sig_ext = torch.cat([sigma, torch.zeros(sigma.shape[0], 1)], 1).gather(1, idx)
    scale = torch.tensor(float(cfg["scaling_max_value"]))
    if noise_sampling:                                                  # model.py:227-232, float32 step by step
        sdv = (torch.clamp(sig_ext, min=min_noise) * torch.tensor(noise_std)) * scale
    else:                                                               # model.py:236
        sdv = torch.full_like(c, noise_std)
    expect = torch.where(c != 0, c + z * sdv, c).clamp(min=0.0)         # model.py:234/238, 240
    assert torch.equal(y, expect)                                       # bit for bit, given the normals the kernel drew
    assert torch.equal(y[c == 0], torch.zeros_like(y[c == 0]))          # no noise where the clean signal is zero
    # the normals themselves: N(0,1) by KS on the positions that received noise
    zz = z[c != 0].double().numpy()
    assert zz.size > 100_000
    ks = stats.kstest(zz, "norm")
    assert ks.pvalue > KS_ALPHA, ks
    assert abs(zz.mean()) < 0.01 and abs(zz.std() - 1.0) < 0.01 and abs(stats.kurtosis(zz)) < 0.05
    # independence across positions / chunks (lag-1 autocorrelation along time and across chunks)
    zc = z.double().numpy()
    assert abs(np.corrcoef(zc[:, :-1].ravel(), zc[:, 1:].ravel())[0, 1]) < 0.01
    assert abs(np.corrcoef(zc[:-1].ravel(), zc[1:].ravel())[0, 1]) < 0.01
    eng.close()


def test_normal_dwell_distribution():
    """dwell_std > 0 mode (modules.py:425-432): max(N(dwell_mean, dwell_std), min_duration), rounded half-even."""
    sd, cfg = load_ckpt("k9")
    eng = S.Engine(sd, cfg, mode="f16x3")
    b, n = _chunks(9, 7000)
    out = eng.predict_chunks(b, n, S.PredictParams(duration_sampling=False, dwell_mean=12.5, dwell_std=4.0, min_duration=3.0,
                                                   noise_std=0.0, seed=2), debug=True)
    g = out["g"].cpu().double().flatten().numpy()
    assert g.min() >= 3.0
    un = g[g > 3.0]                                                     # uncensored part: a normal truncated at 3
    a = (3.0 - 12.5) / 4.0
    ks = stats.kstest(un, stats.truncnorm(a, np.inf, loc=12.5, scale=4.0).cdf)
    assert ks.pvalue > KS_ALPHA, ks
    assert abs((g <= 3.0).mean() - stats.norm.cdf(a)) < 0.003
    assert np.array_equal(out["dur"].cpu().numpy().flatten(), np.rint(out["g"].cpu().numpy().flatten()).astype(np.int32))
    eng.close()
