"""The predict side of the reference's data module (dataloader.py: PoreDataModule 27-149, load_fasta 401-453,
IterableFastaDataSet 321-355) for code that drives `predict_step` batch by batch the way Lightning does.

The reference cuts every read into chunks with a process pool, one-hot encodes them (fp16 [16,k,5] per chunk, 1,440 B) and lets
a DataLoader collate batches of `predict_batch_size` chunks.  Here a batch is the raw bases of its chunks (uint8 [B,16+k-1] +
n_valid uint8 [B], 25 B per chunk; chunker.encode_read), which `seq2squiggle.predict_step` takes directly; `onehot=True` yields
the reference's own batch format instead (predict_step accepts both).  The streaming path (inference.run_streaming) does not go
through batches at all; this class is the drop-in for callers of the reference's Trainer.predict loop.  Training / validation
loaders are out of scope."""
import logging
from typing import Iterable, Iterator, Optional, Tuple

import torch

from .chunker import n_chunks
from .inference import iter_batches

logger = logging.getLogger("seq2squiggle")


class _PredictLoader:
    """What DataLoader(IterableFastaDataSet(...), batch_size=...) is to the reference: iterable once, len() = the approximate
    chunk count the caller passed as total_l (dataloader.py:354-355)."""

    def __init__(self, reads, k: int, batch_size: int, total_l: int, device, onehot: bool):
        self._reads, self._k, self._bs, self._len, self._dev, self._onehot = reads, k, batch_size, total_l, device, onehot
        self.dataset = self

    def __len__(self) -> int:
        return int(self._len)

    def __iter__(self) -> Iterator[tuple]:
        for ids, bases, n_valid in iter_batches(self._reads, self._k, self._bs, self._dev):
            if not self._onehot:
                yield ids, bases, n_valid
                continue
            # the reference's batch: fp16 one-hot [B,16,k,5] over the alphabet "_ACGT" (utils.py:56-89); unknown letters: zero rows
            B, k = bases.shape[0], self._k
            idx = torch.arange(16, device=bases.device).view(1, 16, 1) + torch.arange(k, device=bases.device).view(1, 1, k)
            win = bases.long()[:, idx.reshape(-1)].reshape(B, 16, k)
            lut = torch.full((256,), -1, dtype=torch.long, device=bases.device)
            for i, ch in enumerate(b"_ACGT"):
                lut[ch] = i
            code = lut[win]
            pad = torch.arange(16, device=bases.device).view(1, 16, 1) >= n_valid.long().view(B, 1, 1)
            code = torch.where(pad, torch.zeros_like(code), code)           # pad k-mers are "_" * k (utils.py:342-347)
            hot = torch.zeros(B, 16, k, 5, dtype=torch.float16, device=bases.device)
            known = code >= 0
            hot[known] = torch.nn.functional.one_hot(code[known], 5).to(torch.float16)
            yield ids, hot


class PoreDataModule:
    """PoreDataModule(config, total_l, data_dir=reads, batch_size=...).predict_dataloader() (dataloader.py:27-149).
    `data_dir` is what the reference passes for prediction: an iterable of (read sequence, read name) pairs
    (utils.get_reads).  rank / world_size are accepted and, as in the reference (dataloader.py:448-449), do not shard the
    dataset: sharded runs shard the READ SET (parallel.shard_reads, inference_run)."""

    def __init__(self, config: dict, total_l: int = 1, data_dir: Optional[Iterable[Tuple[str, str]]] = None,
                 valid_dir: str = "path/to/dir", batch_size: int = 128, n_workers: int = 1, rank: int = 0, world_size: int = 1,
                 device="cpu", onehot: bool = False):
        self.config, self.total_l, self.data_dir, self.valid_dir = config, total_l, data_dir, valid_dir
        self.batch_size, self.n_workers, self.rank, self.world_size = batch_size, n_workers, rank, world_size
        self.device, self.onehot = device, onehot
        self._reads = None

    def setup(self, stage: Optional[str] = None) -> None:
        if stage in ("fit", "validate"):
            raise NotImplementedError("training / validation data loading is out of scope of the predict engine")
        if stage in (None, "predict"):
            logger.debug("Loading fasta started")
            self._reads = self.data_dir
            logger.debug("Loading fasta ended")

    def train_dataloader(self):
        raise NotImplementedError("training is out of scope of the predict engine")

    val_dataloader = train_dataloader

    def predict_dataloader(self) -> _PredictLoader:
        if self._reads is None:
            self.setup("predict")
        loader = _PredictLoader(self._reads, int(self.config["seq_kmer"]), self.batch_size, self.total_l, self.device, self.onehot)
        logger.info(f"True Prediction dataset size {len(loader)}")
        return loader
