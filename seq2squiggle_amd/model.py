"""Host-side model object of the predict path: the reference's `seq2squiggle` LightningModule
(model.py:25-63, 195-307) without Lightning, with `predict_step` running on the HIP engine.

Same constructor keywords, same attributes (`results`, `total_samples`, `out_writer`, `hparams.config`),
same methods (`predict_step`, `export_and_clear_results`, `on_predict_epoch_end`), same writer
hand-off (`writer.signals = {read_id: 1-D fp32 tensor}; writer.save()`).  Training is out of scope.
"""
import logging
from collections import defaultdict
from types import SimpleNamespace
from typing import Optional

import torch

from .checkpoint import load_checkpoint
from .engine import Engine, PredictParams
from .signal_io import BLOW5Writer

logger = logging.getLogger("seq2squiggle")

_LETTERS = torch.tensor([ord(c) for c in "_ACGT"] + [ord("N")], dtype=torch.uint8)


def onehot_to_bases(data: torch.Tensor):
    """The reference's batch tensor [B,16,k,5] (one-hot, any float dtype; all-zero row = unknown letter,
    dataloader.py/utils.py:56-89) -> (bases uint8 [B,16+k-1], n_valid uint8 [B]) on the same device."""
    B, T, k, _ = data.shape
    hot = data > 0
    code = torch.where(hot.any(-1), hot.float().argmax(-1), torch.full((), 5, device=data.device)).long()   # [B,T,k]
    is_pad = (code == 0).all(-1)                                                   # all-"_" k-mer
    # n_valid = index after the last non-pad k-mer
    idx = torch.arange(1, T + 1, device=data.device).view(1, T)
    n_valid = ((~is_pad) * idx).max(dim=1).values                                   # 0 for an all-pad chunk
    letters = _LETTERS.to(data.device)
    bases = torch.full((B, T + k - 1), ord("_"), dtype=torch.uint8, device=data.device)
    bases[:, :T] = letters[code[:, :, 0]]
    last = (n_valid - 1).clamp(min=0)
    tail = letters[code[torch.arange(B, device=data.device), last]]                # [B,k] letters of the last real k-mer
    pos = last.view(B, 1) + torch.arange(k, device=data.device).view(1, k)
    bases.scatter_(1, pos, tail)
    # positions of pad k-mers inside the window keep whatever the overlapping real k-mers wrote; the kernel
    # ignores bytes of k-mers >= n_valid
    return bases.contiguous(), n_valid.to(torch.uint8).contiguous()


class seq2squiggle:
    """Feed-forward-transformer signal predictor, predict path only."""

    def __init__(self, *, config: dict, save_valid_plots: bool = True, out_writer=None, dwell_mean: float = 9.0,
                 dwell_std: float = 0.0, noise_std: float = -1, noise_sampling: bool = False,
                 duration_sampling: bool = False, export_every_n_samples: int = 2000000, min_noise: float = 0.5,
                 min_duration: int = 1, state_dict=None, device: Optional[int] = None, mode: str = "f16x3",
                 seed: int = 0, first_global_chunk: int = 0):
        if state_dict is None:
            raise ValueError("the predict-only model needs trained weights: use load_from_checkpoint()")
        self.config = config
        self.hparams = SimpleNamespace(config=config)
        self.save_valid_plots = save_valid_plots
        self.results = []
        self.out_writer = out_writer
        self.dwell_mean = dwell_mean
        self.dwell_std = dwell_std
        self.noise_std = noise_std
        self.noise_sampling = noise_sampling
        self.duration_sampling = duration_sampling
        self.export_every_n_samples = export_every_n_samples
        self.total_samples = 0
        self.min_noise = min_noise
        self.min_duration = min_duration
        self.seed = seed
        self.chunks_done = int(first_global_chunk)      # global chunk index: keys the device RNG counters
        self.engine = Engine(state_dict, config, device=device, mode=mode)
        self.device = self.engine.device

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, **overrides):
        """Reference: Lightning's load_from_checkpoint (inference.py:386-397): cls(**hyper_parameters, **overrides)
        then load_state_dict."""
        sd, cfg = load_checkpoint(str(checkpoint_path))
        return cls(config=cfg, state_dict=sd, **overrides)

    def eval(self):
        return self

    def _params(self) -> PredictParams:
        return PredictParams(dwell_mean=float(self.dwell_mean), dwell_std=float(self.dwell_std), noise_std=float(self.noise_std),
                             noise_sampling=bool(self.noise_sampling), duration_sampling=bool(self.duration_sampling),
                             min_noise=float(self.min_noise), min_duration=float(self.min_duration), seed=int(self.seed))

    def predict_step(self, batch):
        """model.py:195-250.  batch = (read_ids, data[, n_valid]): data is either the reference's one-hot
        [B,16,k,5] tensor or raw bases uint8 [B,16+k-1] with n_valid uint8 [B]."""
        read_id, data, *args = batch
        if data.dim() == 4:
            bases, n_valid = onehot_to_bases(data.to(self.device))
        else:
            bases, n_valid = data.to(self.device), args[0].to(self.device)
        out = self.engine.predict_chunks(bases.contiguous(), n_valid.contiguous(), self._params(),
                                         first_global_chunk=self.chunks_done)
        self.chunks_done += bases.shape[0]
        prediction = out["signal"]
        self.last_durations = out["dur"]

        d = {}
        for read, pred in zip(read_id, prediction):
            d.setdefault(read, []).append(pred)
        self.results.append(d)

        self.total_samples += bases.shape[0]
        if isinstance(self.out_writer, BLOW5Writer) and self.total_samples >= self.export_every_n_samples:
            self.export_and_clear_results(keep_last=True)
            self.total_samples = 0

    def export_and_clear_results(self, keep_last: bool = True):
        """model.py:253-302: merge the per-batch dicts, hold back the last read while batches are still coming,
        concatenate each read's rows and strip every sample equal to 0, hand the dict to the writer."""
        res = defaultdict(list)
        for d in self.results:
            for k, v in d.items():
                res[k].extend(v)
        last_read = None
        if keep_last and res:
            last_key = next(reversed(res))
            last_read = {last_key: res.pop(last_key)}
        if res:
            # one GPU compaction for all reads instead of a cat + nonzero per read (model.py:284-286)
            keys = list(res.keys())
            rows = torch.stack([r for k in keys for r in res[k]])
            first = [0]
            for k in keys:
                first.append(first[-1] + len(res[k]))
            ex = self.engine.export_reads(rows, torch.tensor(first, dtype=torch.int32, device=self.device), want_pa=True)
            offs = ex["offsets"].tolist()
            for i, k in enumerate(keys):
                res[k] = ex["pa"][offs[i]:offs[i + 1]]
        self.out_writer.signals = res
        self.out_writer.save()
        self.out_writer.signals = []
        self.results.clear()
        self.results = []
        if last_read:
            self.results.append(last_read)
        logger.debug("Results exported and memory cleared.")

    def on_predict_epoch_end(self):
        if self.results:
            self.export_and_clear_results(keep_last=False)
        logger.debug("Epoch end operation completed.")
