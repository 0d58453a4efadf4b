"""Predict orchestration: the reference's inference.py (get_writer 30-82, check_model 224-267,
inference_run 270-427) with the Lightning Trainer replaced by a plain loop over batches of chunks.
Weight download (inference.py:85-221) needs the network and is out of scope: pass --model."""
import logging
import os
from typing import Iterable, Iterator, Tuple

import numpy as np
import torch

from .chunker import encode_read, n_chunks as _n_chunks, pack_reads
from .model import seq2squiggle
from .parallel import local_device, rank_output_path, rank_world, shard_reads
from .signal_io import BLOW5Writer, POD5Writer
from .utils import get_profile, get_reads, update_config, update_profile

logger = logging.getLogger("seq2squiggle")


def redo_threshold() -> float:
    """The redo share above which the exact attention instance is the faster one: the threshold of s2s_create's calibration
    (include/s2s_hip.h: S2S_ATTENTION_REDO_THRESHOLD)."""
    from ._lib import lib
    fn = getattr(lib(), "s2s_attention_redo_threshold", None)
    return float(fn()) if fn is not None else 0.055


def get_writer(out: str, profile: object, ideal_mode: bool, export_every_n_samples: int, profile_name: str,
               preserve_read_ids: bool) -> tuple:
    """Writer by output extension (inference.py:30-82): deletes an existing file, creates the directory."""
    out = str(out)
    out_base = os.path.basename(out)
    out_dir = os.path.dirname(out)
    if out_dir and not os.path.exists(out_dir):
        os.makedirs(out_dir, exist_ok=True)
    if os.path.exists(out):
        logger.warning(f"Output file {out} already exists. File will be deleted.")
        os.remove(out)
    if any(out_base.endswith(ext) for ext in (".blow5", ".slow5")):
        return BLOW5Writer(out, profile, ideal_mode, profile_name, preserve_read_ids), export_every_n_samples
    if out_base.endswith(".pod5"):
        return POD5Writer(out, profile, ideal_mode, profile_name, preserve_read_ids), float("inf")
    logger.error("Output file must have .pod5, .slow5, or .blow5 extension.")
    raise ValueError("Output file must have .pod5, .slow5, or .blow5 extension.")


REFERENCE_VERSION = "0.3.4"        # the release of the reference this engine mirrors (pyproject.toml): the `@vX.Y.Z` its cache is searched for


def user_cache_dir(appname: str = "seq2squiggle") -> str:
    """appdirs.user_cache_dir(appname, False, opinion=False) on Linux, without the dependency: $XDG_CACHE_HOME (default ~/.cache) /
    appname -- the directory the reference keeps its downloaded weights in (inference.py:104)."""
    return os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.expanduser("~/.cache"), appname)


def get_saved_weights(profile_name: str, version: str = REFERENCE_VERSION) -> str:
    """The cache half of the reference's get_saved_weights (inference.py:85-149): when `-m` is omitted, look in the reference's own
    cache directory for `<name>@vX.Y.Z.ckpt` files of this chemistry ("R10" / "R9" in the file name, by --profile) and return the
    best version match -- what a reference user with downloaded weights runs offline today.  The download half (inference.py:151-221)
    needs the network and stays out of scope: with no match this raises FileNotFoundError naming the directory searched.

    The loop is the reference's, quirks included, so that the same cache yields the same file: `version` is the version STRING, so
    `zip(version, file_version)` pairs its characters '0', '.', '3' with the file's (major, minor, patch) -- a file scores 1 for an
    equal major plus 1 when its patch equals the minor's digit, candidates are visited in os.listdir order and only a strictly
    better score replaces the pick; a profile with neither keyword (rna-004-*) never matches a cached file.  One deliberate
    difference: a `.ckpt` without `@vX.Y.Z` in its name is skipped (the reference's loop dies on it with an AttributeError)."""
    import re
    logger.info("Weights file path is not provided.")
    cache_dir = user_cache_dir("seq2squiggle")
    if profile_name.startswith("dna-r10"):
        logger.info("Detected R10.4.1 chemistry profile.")
        logger.info("Profile can be changed with the --profile parameter")
        profile_keyword = "R10"
    elif profile_name.startswith("dna-r9"):
        logger.info("Detected R9.4.1 chemistry profile.")
        logger.info("Profile can be changed with the --profile parameter")
        profile_keyword = "R9"
    else:
        logger.warning("Profile name '%s' does not match known patterns (R10- or R9-). Proceeding with latest weights.", profile_name)
        profile_keyword = None
    best, best_score = None, 0
    try:
        os.makedirs(cache_dir, exist_ok=True)                  # (inference.py:105)
        names = os.listdir(cache_dir)
    except OSError:
        names = []
    for filename in names:
        root, ext = os.path.splitext(filename)
        if ext != ".ckpt":
            continue
        m = re.match(r".*@v(\d+).(\d+).(\d+)", root)
        if m is None:
            logger.debug(f"{filename}: no @vX.Y.Z in the name, skipped")
            continue
        same = [i == j for i, j in zip(version, m.groups())]
        score = sum(same) if same[0] else 0
        if score > best_score and profile_keyword and profile_keyword in root:
            best, best_score = os.path.join(cache_dir, filename), score
    if best_score > 0:
        logger.info("Found matching weights in local cache: %s", best)
        return best
    raise FileNotFoundError(f"no model weights given and none for profile {profile_name} (release v{version}) in the cache directory "
                            f"{cache_dir}: downloading released weights needs network access; pass --model <file.ckpt>")


_EXCLUDE = ("log_name", "wandb_logger_state", "max_chunks_train", "max_chunks_valid", "train_valid_split",
            "train_batch_size", "save_model")


def check_model(model: object, config: dict) -> None:
    """Warn on checkpoint/config mismatches, raise on a seq_kmer mismatch (inference.py:224-267)."""
    model_params = model.hparams.config
    for param, value in config.items():
        if param in _EXCLUDE or model_params.get(param) == value:
            continue
        if param == "seq_kmer":
            raise ValueError(f"Parameter 'seq_kmer' mismatch: Model checkpoint value is {model_params.get(param)}, while "
                             f"config value is {value}. The model was trained on {model_params.get(param)}-mers, while the "
                             f"config file expects {value}-mers. Choose a different model or change the config value or "
                             "the --profile option. ")
        logger.warning(f"Mismatching {param} parameter in model checkpoint ({model_params.get(param)}) and in config "
                       f"file ({value})")


def iter_batches(reads: Iterable[Tuple[str, str]], k: int, batch_size: int, device) -> Iterator[tuple]:
    """(read_ids, bases, n_valid) batches of up to batch_size chunks in read order: the job of load_fasta +
    DataLoader (dataloader.py:401-453, 141-149).  Reads shorter than k yield nothing (dataloader.py:393-398)."""
    ids, parts, nvs, n = [], [], [], 0
    for seq, name in reads:
        b, nv = encode_read(seq, k)
        s = 0
        while s < b.shape[0]:
            take = min(batch_size - n, b.shape[0] - s)
            ids.extend([name] * take)
            parts.append(b[s:s + take])
            nvs.append(nv[s:s + take])
            n += take
            s += take
            if n == batch_size:
                yield (tuple(ids), torch.from_numpy(np.concatenate(parts)).to(device),
                       torch.from_numpy(np.concatenate(nvs)).to(device))
                ids, parts, nvs, n = [], [], [], 0
    if n:
        yield tuple(ids), torch.from_numpy(np.concatenate(parts)).to(device), torch.from_numpy(np.concatenate(nvs)).to(device)


class _Staging:
    """One super-batch's inputs, host -> device: the arrays are laid back to back in a persistent pinned buffer (pinning is
    paid once, not per batch -- a fresh pinned allocation costs tens of milliseconds) and go up as ONE copy on the upload
    stream, so the calling thread never waits for the compute stream the way a pageable copy does."""

    def __init__(self, device):
        self.device = device
        self.host = None

    def upload(self, arrays, stream):
        offs, total = [], 0
        for a in arrays:
            offs.append(total)
            total += -(-a.nbytes // 16) * 16
        if self.host is None or self.host.numel() < total:
            self.host = torch.empty(max(total + total // 4, 1 << 20), dtype=torch.uint8, pin_memory=True)
        hv = self.host.numpy()
        for a, o in zip(arrays, offs):
            hv[o:o + a.nbytes] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(stream):
            d = self.host[:total].to(self.device, non_blocking=True)
            up = torch.cuda.Event()
            up.record(stream)
        main.wait_event(up)
        d.record_stream(main)
        return [d[o:o + a.nbytes].view(torch.from_numpy(a[:0]).dtype) for a, o in zip(arrays, offs)]


_STAGING = {}             # device -> the two pinned staging buffers, kept for the life of the process
_STREAMS = {}             # device -> (download, upload) copy streams, kept for the life of the process
_IO = None
_TRACE = None             # tools/e2e_timeline.py points this at a list to receive run_streaming's event times


def _io_executor():
    """ONE writer thread for the whole process.  (Not one per run: with the pyarrow build of this image, Arrow calls from
    a second thread after a first Arrow-using thread has exited segfault, so the container writers always run on this
    long-lived thread.)"""
    global _IO
    if _IO is None:
        from concurrent.futures import ThreadPoolExecutor
        _IO = ThreadPoolExecutor(max_workers=1, thread_name_prefix="s2s-writer")
    return _IO


def _copy_streams(dev):
    """The two copy streams of run_streaming, made once per device and process.  Every call pushes a few bytes through each
    and waits for them: a HIP stream gets its hardware queue at its first submission (milliseconds), and after some
    milliseconds without work the copy engines take 10-25 ms to come back for the first transfer (measured: the first upload of
    a run that starts ~15 ms after the previous one ended) -- inference_run pays both on its loader thread while the main
    thread parses the FASTA."""
    key = str(dev)
    if key not in _STREAMS:
        _STREAMS[key] = (tuple(torch.cuda.Stream(dev) for _ in range(2)), torch.zeros(4096, dtype=torch.uint8, pin_memory=True),
                         torch.zeros(4096, dtype=torch.uint8, device=dev))
    streams, host, devbuf = _STREAMS[key]
    for st in streams:
        with torch.cuda.stream(st):
            devbuf.copy_(host, non_blocking=True)
            host.copy_(devbuf, non_blocking=True)
    for st in streams:
        st.synchronize()
    return streams


def super_batches(reads: Iterable[Tuple[str, str]], k: int, max_chunks: int):
    """Whole reads grouped into super-batches of about max_chunks chunks (reads too short for one chunk are dropped).  The first
    groups are short (1/8, 1/4, 1/2 of max_chunks): the GPU starts as soon as a few reads exist, and the host, which prepares a
    chunk faster than the GPU predicts one, is ahead from then on.  When the iterable says how many reads are left
    (operator.length_hint: lists, utils.CountedReads) the groups of a LONG job go on doubling, up to GROW x max_chunks while at
    least four groups of that size remain -- the export and codec kernels between two predict launches cost the same 0.1-0.3 ms
    whatever the size (they do not fill the GPU), and a predict launch has its own ramp and tail: 131,072 instead of 32,768 chunks
    per group is + 3.5 % end to end into .pod5 on one GPU's share of BASELINE configs[4] -- and the last ones shrink again: what
    remains after the GPU's last kernel is then the D2H + compression + write of a small super-batch only."""
    import operator
    GROW = 4
    floor = max(max_chunks // 8, 1)
    ramp = want = floor
    seen_reads = seen_chunks = 0
    group, n = [], 0
    reads = iter(reads)
    for seq, name in reads:
        c = _n_chunks(len(seq), k)
        if c == 0:
            logger.debug(f"Skipped read {name}.")
            continue
        group.append((seq, name))
        n += c
        seen_reads += 1
        seen_chunks += c
        if n >= want:
            yield group
            group, n = [], 0
            left = operator.length_hint(reads, 0) * seen_chunks // seen_reads      # chunks still to come; 0: not known
            ramp = want = min(2 * ramp, GROW * max_chunks)
            while want > max_chunks and left < 4 * want:
                want //= 2
            if 0 < left < 2 * want:
                want = max(left // 2, floor)
    if group:
        yield group


def run_streaming(model, reads: Iterable[Tuple[str, str]], writer, profile_dict: dict, profile_name: str,
                  max_chunks: int = 32768, trace: list = None) -> int:
    """The predict loop without per-chunk Python objects: whole reads are grouped into super-batches of about
    `max_chunks` chunks; per super-batch one H2D of the packed read bytes, s2s_predict_packed, s2s_export_reads
    (zero-strip + int16 conversion on the GPU), one D2H of the packed int16 samples on a copy stream, then the writer.  Produces the
    same records as predict_step + export_and_clear_results + writer.save().

    Three stages overlap: while the GPU works on super-batch i the host samples/packs i+1, and a writer thread
    compresses and writes i-1.  Record metadata (the np.random offset draws of signal_io.py:129-132) is built on the
    calling thread in read order, so the output does not depend on thread timing.  `trace`: a list that receives
    (event, seconds) pairs for tools/e2e_timeline.py."""
    import time

    def mark(ev):
        if trace is not None:
            trace.append((ev, time.perf_counter()))
    k = model.config["seq_kmer"]
    dev = model.device
    rna = profile_name.startswith("rna")
    total = 0
    io = _io_executor()
    import collections
    pending = collections.deque()   # writer jobs in flight on the (single) writer thread, oldest first: at most MAX_PENDING super-batches
    MAX_PENDING = 3                 # of records wait there (~17 MB each), so a slow batch on the writer does not stall the next launch
    inflight = None           # (ids, device buffer, layout, kernels-done event) of the super-batch on the GPU
    # copy_stream: D2H of finished super-batches; up_stream: H2D of the next one (its own stream: never queued behind a D2H that
    # waits for kernels)
    copy_stream, up_stream = _copy_streams(dev)

    staging = _STAGING.setdefault(str(dev), (_Staging(dev), _Staging(dev)))   # super-batch i + 2 reuses i's buffer: i has been collected by then
    n_launched = 0

    gpu_rows = writer.gpu_signal_rows() if hasattr(writer, "gpu_signal_rows") else None   # (svb variant, samples per row) | None

    def launch(group):
        nonlocal total, n_launched
        mark("pack")
        flat, chunk_start, n_valid, read_first = pack_reads([s for s, _ in group], k)
        B = int(read_first[-1])
        if B == 0:
            return None
        arrays = [flat, chunk_start, n_valid, read_first]
        if gpu_rows:
            # candidate rows of the signal codec from the chunk counts (the stripped lengths are not known here): read r may
            # need up to ceil(250 * chunks / row_samples) rows; those it turns out not to need come back empty
            n_rows = -(-(np.diff(read_first).astype(np.int64) * 250) // gpu_rows[1])
            row_read = np.repeat(np.arange(len(group), dtype=np.int32), n_rows)
            row_index = (np.arange(int(n_rows.sum())) - np.repeat(np.cumsum(n_rows) - n_rows, n_rows)).astype(np.int32)
            arrays += [row_read, row_index]
        mark("h2d+launch")
        # H2D from pinned staging on the copy stream: a pageable copy on the compute stream would hold this thread until
        # the previous super-batch's kernels have drained
        ins = staging[n_launched % 2].upload(arrays, up_stream)
        mark("uploaded")
        out = model.engine.predict_packed(ins[0], ins[1], ins[2], model._params(), first_global_chunk=model.chunks_done)
        model.chunks_done += B
        total += B
        n_launched += 1
        mark("predict queued")
        names = [n for _, n in group]
        R = len(group)
        main = torch.cuda.current_stream(dev)
        # Everything the host needs from this super-batch sits in ONE device buffer and leaves as ONE copy: [read offsets int64
        # R+1][row offsets int64 N+1 (coded signal only)][payload].  A copy of a few KB would be a shader copy, and no other
        # wave fits on a CU while the predict kernel holds its whole register file: it would wait for the NEXT super-batch's
        # kernel to end; a large copy goes through the DMA engines beside it.  The whole capacity is copied (its size is known
        # without a sync: 16 MB of int16 / 17 MB of coded signal per 32 k chunks) into pinned memory.
        cap = B * 250
        if gpu_rows:
            N = int(row_read.shape[0])
            blob_cap = model.engine.svb_capacity(cap, N, gpu_rows[0])
            head = 8 * (R + 1) + 8 * (N + 1)
            head += -head % 16
            buf = torch.empty(head + max(blob_cap, 1), dtype=torch.uint8, device=dev)
            offs_d = buf[:8 * (R + 1)].view(torch.int64)
            rows_d = buf[8 * (R + 1): 8 * (R + 1) + 8 * (N + 1)].view(torch.int64)
            ex = model.engine.export_reads(out["signal"], ins[3], profile_dict["digitisation"], profile_dict["range"],
                                           profile_dict["offset_mean"], rna=rna, want_pa=False, want_dac=True, out_offsets=offs_d)
            mark("export queued")
            # the signal leaves the GPU StreamVByte-coded (~1.1-1.3 bytes per sample)
            model.engine.svb_encode(ex["dac"], ex["offsets"], ins[4], ins[5], gpu_rows[1], gpu_rows[0], cap, out=buf[head:],
                                    out_offsets=rows_d)
        else:
            head = 8 * (R + 1)
            head += -head % 16
            buf = torch.empty(head + 2 * cap, dtype=torch.uint8, device=dev)
            model.engine.export_reads(out["signal"], ins[3], profile_dict["digitisation"], profile_dict["range"],
                                      profile_dict["offset_mean"], rna=rna, want_pa=False, want_dac=True,
                                      out_offsets=buf[:8 * (R + 1)].view(torch.int64), out_dac=buf[head:].view(torch.int16))
            mark("export queued")
        ready = torch.cuda.Event()
        ready.record(main)
        mark("launched")
        return names, buf, (R, head, row_read if gpu_rows else None), ready

    def collect(job):
        ids, buf, (R, head, row_read), ready = job
        # The D2H is issued only once the super-batch's kernels have FINISHED (the calling thread has nothing else to do at this
        # point: the next super-batch is already queued behind them).  Queued earlier, behind a stream-side wait for `ready`, the
        # runtime carries the copy out as a 256-workgroup shader copy, and no wave of it fits on a CU while the next predict kernel
        # holds the register files: it ended with THAT kernel, one super-batch late, and every second launch then waited 1.5-2.5 ms
        # for this thread (rocprofv3 --kernel-trace: 9 % of the GPU's time on one GPU's share of configs[4]).  A copy with nothing
        # pending in front of it goes through the DMA engines, beside the running kernel (tools/d2h_probe.py: 9 MB in 0.23 ms).
        mark("wait gpu")
        ready.synchronize()
        mark("d2h")
        with torch.cuda.stream(copy_stream):
            buf_h = torch.empty(buf.shape, dtype=torch.uint8, pin_memory=True)
            buf_h.copy_(buf, non_blocking=True)
            buf.record_stream(copy_stream)
            done = torch.cuda.Event()
            done.record(copy_stream)
        mark("wait d2h")
        done.synchronize()
        mark("records")
        host = buf_h.numpy()
        offs = host[:8 * (R + 1)].view(np.int64)
        if gpu_rows:
            N = int(row_read.shape[0])
            row_offs = host[8 * (R + 1): 8 * (R + 1) + 8 * (N + 1)].view(np.int64)
            if row_offs[-1] < 0:               # s2s_svb_encode: the coded rows did not fit (cannot happen with svb_capacity's bound)
                raise RuntimeError(f"s2s_svb_encode needed {-int(row_offs[-1])} bytes, the buffer holds {host.size - head}")
            recs = writer.svb_records(ids, offs, row_read, row_offs, host[head: head + int(row_offs[-1])])
        else:
            recs = writer.dac_records(ids, host[head: head + 2 * int(offs[-1])].view(np.int16), offs)
        mark("wait writer")
        while pending and (pending[0].done() or len(pending) >= MAX_PENDING):
            pending.popleft().result()              # (re-raises what the writer thread raised)
        mark("submit")

        def job_(recs=recs):
            mark("writer start")
            writer.write_records(recs)
            mark("writer end")
        pending.append(io.submit(job_))

    t_start = t_log = time.perf_counter()
    n_reads = 0
    try:
        mark("first read wanted")
        for group in super_batches(reads, k, max_chunks):
            job = launch(group)
            if inflight is not None:
                collect(inflight)
            inflight = job
            n_reads += len(group)
            now = time.perf_counter()
            if now - t_log > 10.0:                 # long jobs: a progress line every ten seconds (the reference shows Lightning's bar)
                t_log = now
                logger.info(f"{n_reads} reads, {total} chunks queued after {now - t_start:.0f} s "
                            f"({total / (now - t_start):.3g} chunks/s)")
        if inflight is not None:
            collect(inflight)
        while pending:
            pending.popleft().result()
    finally:
        if hasattr(writer, "close"):               # POD5: run-info and reads tables, footer (on the writer thread as well)
            io.submit(writer.close).result()
        mark("done")
    return total


def inference_run(config: dict, saved_weights: str, fasta: str, read_input: bool, n: int, r: int, c: int, out: str,
                  profile: dict, dwell_mean: int, dwell_std: float, noise_std: float, noise_sampling: bool,
                  duration_sampling: bool, distr: str, predict_batch_size: int, export_every_n_samples: int,
                  sample_rate: int, bps: int, digitisation: int, range_val: float, offset_mean: float, offset_std: float,
                  median_before_mean: float, median_before_std: float, min_noise: float, min_duration: float,
                  min_read_len: int, preserve_read_ids: bool, seed: int, mode: str = "f16x3", streaming: bool = True,
                  attention_path: str = "auto"):
    """Same 30 parameters as the reference (inference.py:270-301) plus `mode` (decoder arithmetic), `streaming`
    (True: run_streaming; False: the reference's predict_step / export_and_clear_results flow, batch by batch) and
    `attention_path` ("auto": the engine's calibration decides; "fast" / "exact": Engine.attention_path is set to it)."""
    if attention_path not in ("auto", "fast", "exact"):
        raise ValueError("attention_path must be 'auto', 'fast' or 'exact'")
    profile_dict = get_profile(profile)
    profile_dict = update_profile(profile_dict, sample_rate=sample_rate, bps=bps, digitisation=digitisation, range=range_val,
                                  offset_mean=offset_mean, offset_std=offset_std, median_before_mean=median_before_mean,
                                  median_before_std=median_before_std)
    if dwell_mean is None:
        dwell_mean = profile_dict["sample_rate"] / profile_dict["bps"]
    config = update_config(profile, config)
    ideal_mode = not (duration_sampling or dwell_std > 0)

    rank, _, world = rank_world()
    local_rank = local_device()
    writer, export_every_n_samples = get_writer(rank_output_path(str(out), rank, world), profile_dict, ideal_mode,
                                                export_every_n_samples, profile_name=profile,
                                                preserve_read_ids=preserve_read_ids)
    if saved_weights is None:
        saved_weights = get_saved_weights(profile)             # (inference.py:370-372; the cache only: no network here)
    first_chunk, first_read, total_l = 0, 0, 0
    if world > 1 and not seed:
        raise ValueError("multi-process runs need one seed for all ranks: pass an explicit --seed, or let the CLI share a "
                         "fresh one (parallel.shared_seed) before calling inference_run")
    # belt and braces for a user's own torchrun (all devices visible): whatever allocates without naming a device -- a pinned buffer's
    # primary-context search, a library call -- lands on this rank's GPU.  Children of `predict --gpus N` see one device only.
    if torch.cuda.is_available():              # (without a GPU the engine's constructor is what raises, with its own message)
        torch.cuda.set_device(local_rank)
    # the checkpoint is read and the engine created (weight re-packing, device allocations: native code, no interpreter lock)
    # on a helper thread while this one parses the FASTA and samples the reads
    from concurrent.futures import ThreadPoolExecutor
    loader = ThreadPoolExecutor(max_workers=1, thread_name_prefix="s2s-load")

    def load(**kw):
        if streaming:
            _copy_streams(torch.device("cuda", int(kw["device"])))
        return seq2squiggle.load_from_checkpoint(**kw)
    loading = loader.submit(
        load, checkpoint_path=saved_weights, out_writer=writer, dwell_mean=dwell_mean,
        dwell_std=dwell_std, noise_std=noise_std, noise_sampling=noise_sampling, duration_sampling=duration_sampling,
        export_every_n_samples=export_every_n_samples, min_noise=min_noise, min_duration=min_duration, device=local_rank,
        mode=mode, seed=seed)
    try:
        if world > 1 and not read_input:
            # every rank replays the sampler for the read lengths, then builds only its own contiguous share of the reads
            from .utils import preprocess_genome, sample_read_shard
            genome_seqs, genome_lens = preprocess_genome(fasta)
            picked = {}

            def shard_of(lens):
                lo, hi, picked["first"] = shard_reads(lens, config["seq_kmer"], world)[rank]
                picked["lo"] = lo
                picked["records_before"] = sum(1 for L in lens[:lo] if _n_chunks(int(L), config["seq_kmer"]) > 0)
                return lo, hi
            reads, lens = sample_read_shard(genome_seqs, genome_lens, n, r, c, seed, distr, profile, min_read_len, shard_of)
            first_chunk, first_read = picked["first"], picked["records_before"]
            logger.info(f"rank {rank}/{world}: reads {picked['lo']}.. of {len(lens)}, first global chunk {first_chunk}")
        else:
            reads, total_l = get_reads(fasta, read_input, n, r, c, config, distr, seed, profile, min_read_len, lazy=world == 1)
            if world > 1:                  # read mode: every rank parses the same file, then keeps its contiguous share
                reads = list(reads)
                lo, hi, first_chunk = shard_reads([len(s) for s, _ in reads], config["seq_kmer"], world)[rank]
                # records written before this shard: reads too short for one chunk produce none (dataloader.py:393-398), so they
                # must not advance the shard writer's read numbering / record draws either
                first_read = sum(1 for s_, _ in reads[:lo] if _n_chunks(len(s_), config["seq_kmer"]) > 0)
                reads = reads[lo:hi]
                logger.info(f"rank {rank}/{world}: reads {lo}..{hi}, first global chunk {first_chunk}")
    finally:
        if _TRACE is not None:
            import time
            _TRACE.append(("reads ready", time.perf_counter()))
        loader.shutdown(wait=True)

    if first_read:
        writer.start_at(first_read)            # read ids / read_number / record draws continue the single-process run
    load_model = loading.result()
    if attention_path != "auto":
        load_model.engine.attention_path = attention_path
    load_model.chunks_done = int(first_chunk)  # global chunk index of this rank's first chunk: keys the device RNG counters
    load_model.first_global_chunk = int(first_chunk)
    check_model(load_model, config)

    n_chunks = 0
    if _TRACE is not None:
        import time
        _TRACE.append(("model ready", time.perf_counter()))
    if streaming and hasattr(writer, "dac_records"):
        n_chunks = run_streaming(load_model, reads, writer, profile_dict, profile, trace=_TRACE)
    else:
        for batch in iter_batches(reads, config["seq_kmer"], predict_batch_size, load_model.device):
            load_model.predict_step(batch)
            n_chunks += len(batch[0])
        load_model.on_predict_epoch_end()
    torch.cuda.synchronize(load_model.device)
    logger.info(f"Predicted {n_chunks} chunks ({n_chunks * 250} padded samples).")
    # how THESE weights behaved on THIS input (s2s_stats_read; the read resets the counters): the redo share of the fast softmax
    # path depends on the reads as well as on the checkpoint, and the calibration launch only saw 512 pseudo-random chunks
    eng = load_model.engine
    st = eng.stats()
    load_model.run_stats = dict(st, attention_path=eng.attention_path, calibration_redo_rate=eng.calibration_redo_rate)
    logger.debug(f"attention path {eng.attention_path} (calibration launch: {100 * eng.calibration_redo_rate:.2f} % of the heads redone); "
                 f"this run: {100 * st['redo_rate']:.3f} % redone, {st['in_kernel_clock_ghz'] or 0:.2f} GHz in the kernel, "
                 f"{st['cycles_per_chunk_and_cu'] or 0:.0f} cycles per chunk and CU")
    if eng.attention_path == "fast" and st["redo_rate"] > redo_threshold():
        logger.warning(f"{100 * st['redo_rate']:.1f} % of the attention heads of this run overflowed the fast softmax path and were redone "
                       f"(the checkpoint's calibration launch saw {100 * max(eng.calibration_redo_rate, 0):.1f} %; above "
                       f"{100 * redo_threshold():.1f} % the exact path is faster). The output is the same either way; "
                       f"for this kind of input run with --attention-path exact.")
    shard = rank_output_path(str(out), rank, world)
    if world > 1 and not os.path.exists(shard) and hasattr(writer, "write_records"):
        # a rank without reads (more ranks than reads) still leaves its -- empty -- shard, so that the rank files always merge

        def empty():
            writer.write_records([])
            if hasattr(writer, "close"):
                writer.close()
        _io_executor().submit(empty).result()
    return load_model
