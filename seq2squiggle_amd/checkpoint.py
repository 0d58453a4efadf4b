"""Checkpoint reading: the Lightning .ckpt the reference trains and loads.

Reference: seq2squiggle.load_from_checkpoint (inference.py:386-397) reads
ckpt["hyper_parameters"] and ckpt["state_dict"]; ModelCheckpoint(save_weights_only=True)
(train.py:81-87) writes them.  Here the file is read with plain torch.load and flattened
into the fp32 blob of include/s2s_hip.h (s2s_blob_floats documents the order).
"""
import os
import pickle
from typing import Dict, Tuple

import numpy as np
import torch

from ._lib import S2SConfig

_LAYER = ("slf_attn.w_qs.weight", "slf_attn.w_qs.bias", "slf_attn.w_ks.weight", "slf_attn.w_ks.bias",
          "slf_attn.w_vs.weight", "slf_attn.w_vs.bias", "slf_attn.fc.weight", "slf_attn.fc.bias",
          "slf_attn.layer_norm.weight", "slf_attn.layer_norm.bias", "pos_ffn.w_1.weight", "pos_ffn.w_1.bias",
          "pos_ffn.w_2.weight", "pos_ffn.w_2.bias", "pos_ffn.layer_norm.weight", "pos_ffn.layer_norm.bias")
_MLP = ("0.weight", "0.bias", "3.weight", "3.bias")


_PARSED = {}              # (real path, mtime_ns, size, allow_pickle) -> (state_dict, config) of the last file read


def load_checkpoint(path: str, allow_pickle: bool = None) -> Tuple[Dict[str, torch.Tensor], dict]:
    """-> (state_dict, config).  Only `state_dict` and `hyper_parameters["config"]` are required.

    The file is read with torch's restricted unpickler (tensors, plain dicts, lists and scalars: all a
    ModelCheckpoint(save_weights_only=True) file holds).  A checkpoint that carries other pickled objects is refused
    unless the caller opts into full unpickling (allow_pickle=True or S2S_ALLOW_PICKLE=1): that executes code
    from the file."""
    if allow_pickle is None:
        allow_pickle = os.environ.get("S2S_ALLOW_PICKLE", "") == "1"
    st = os.stat(path)
    key = (os.path.realpath(path), st.st_mtime_ns, st.st_size, bool(allow_pickle))
    hit = _PARSED.get(key)
    if hit is not None:                     # same file as last time (a service predicting again): skip the unpickling
        return dict(hit[0]), dict(hit[1])
    try:
        ck = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if not allow_pickle:
            raise ValueError(f"{path}: holds pickled objects beyond tensors and plain containers ({e}); "
                             "set S2S_ALLOW_PICKLE=1 to unpickle it anyway (this runs code from the file)") from e
        ck = torch.load(path, map_location="cpu", weights_only=False)
    if "state_dict" not in ck:
        raise ValueError(f"{path}: not a seq2squiggle checkpoint (no 'state_dict')")
    cfg = (ck.get("hyper_parameters") or {}).get("config")
    if cfg is None:
        raise ValueError(f"{path}: checkpoint carries no hyper_parameters['config']")
    sd = {k: v.detach().float().cpu() for k, v in ck["state_dict"].items()}
    _PARSED.clear()                         # one entry: the last checkpoint read
    _PARSED[key] = (sd, dict(cfg))
    return dict(sd), dict(cfg)


def blob_names(cfg: dict):
    names = ["encoders.position_enc", "encoders.src_emb.weight", "encoders.src_emb.bias"]
    for i in range(cfg["pre_layers"]):
        names += [f"encoders.pre_net_stack.{i}.weight", f"encoders.pre_net_stack.{i}.bias"]
    for l in range(cfg["encoder_layers"]):
        names += [f"encoders.layer_stack.{l}.{n}" for n in _LAYER]
    for head in ("noise_sampler.stdv_layer", "length_regulator.duration_sampler.conc_layer",
                 "length_regulator.duration_sampler.rate_layer"):
        names += [f"{head}.{n}" for n in _MLP]
    names.append("decoders.position_enc")
    for l in range(cfg["decoder_layers"]):
        names += [f"decoders.layer_stack_FFT.{l}.{n}" for n in _LAYER]
    names += ["decoders.out_linear.weight", "decoders.out_linear.bias"]
    return names


def state_dict_to_blob(sd: Dict[str, torch.Tensor], cfg: dict) -> np.ndarray:
    missing = [n for n in blob_names(cfg) if n not in sd]
    if missing:
        raise ValueError(f"state_dict lacks {missing[:3]}{'...' if len(missing) > 3 else ''}")
    return np.concatenate([sd[n].detach().float().cpu().numpy().ravel() for n in blob_names(cfg)]).astype(np.float32)


MODES = {"f32": 0, "f16x3": 1, "f16": 3}     # f16: reduced precision (see include/s2s_hip.h)


def config_to_c(cfg: dict, mode: str = "f16x3") -> S2SConfig:
    if cfg["encoder_heads"] != cfg["decoder_heads"]:
        raise ValueError("encoder_heads != decoder_heads is not supported")
    if mode not in MODES:
        raise ValueError(f"mode must be one of {sorted(MODES)}")
    if cfg.get("allowed_chars", "_ACGT") != "_ACGT":
        raise ValueError("allowed_chars must be '_ACGT'")
    return S2SConfig(seq_kmer=int(cfg["seq_kmer"]), max_dna_len=int(cfg["max_dna_len"]),
                     max_signal_len=int(cfg["max_signal_len"]), dmodel=int(cfg["dmodel"]), dff=int(cfg["dff"]),
                     n_heads=int(cfg["encoder_heads"]), encoder_layers=int(cfg["encoder_layers"]),
                     decoder_layers=int(cfg["decoder_layers"]), pre_layers=int(cfg["pre_layers"]),
                     scaling_max_value=float(cfg["scaling_max_value"]), compute_mode=MODES[mode])
