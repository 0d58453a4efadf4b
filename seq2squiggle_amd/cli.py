"""`seq2squiggle predict` command line (reference: seq2squiggle.py:43-84, 230-456, 458-599) on plain click.
Same option names and defaults; the other commands of the reference (preprocess, train, sweep) are training
and out of scope."""
import logging
import pathlib
import sys

import click
import yaml

from . import __version__

logger = logging.getLogger("seq2squiggle")


def set_config(config_path) -> dict:
    """YAML config loader (seq2squiggle.py:640-657)."""
    default_config_path = pathlib.Path(__file__).parent / "config.yaml"
    path_to_use = default_config_path if config_path is None else config_path
    try:
        with open(path_to_use, "r") as f_in:
            config = yaml.safe_load(f_in)
    except FileNotFoundError:
        logger.error(f"Configuration file not found: {path_to_use}")
        raise
    except yaml.YAMLError as exc:
        logger.error(f"Error parsing YAML file: {path_to_use} - {exc}")
        raise
    if config_path is None:
        logger.info("Config file was not specified. Default config will be used.")
    return config


@click.group(context_settings=dict(help_option_names=["-h", "--help"]))
def main():
    """seq2squiggle (MI355X engine): predicts nanopore sequencing signals with a feed-forward transformer."""


@main.command()
def version():
    """Print the version."""
    click.echo(__version__)


@main.command(context_settings={"ignore_unknown_options": True})
@click.argument("fasta", required=False, type=click.Path(exists=False, file_okay=True, dir_okay=False, path_type=pathlib.Path))
@click.option("--read-input", default=False, is_flag=True, show_default=True,
              help="Read mode: simulate signals directly from the reads of a FASTA/FASTQ file.")
@click.option("-n", "--num-reads", type=int, default=-1, help="Number of generated reads.")
@click.option("-r", "--read-length", type=int, default=1000, show_default=True, help="Average read length.")
@click.option("-c", "--coverage", type=int, default=-1, help="Genome coverage.")
@click.option("-o", "--out", required=False, type=click.Path(file_okay=True, dir_okay=False, path_type=pathlib.Path),
              help="Output POD5/SLOW5/BLOW5 file.")
@click.option("--profile", default="dna-r10-prom", show_default=True,
              type=click.Choice(["dna-r10-prom", "dna-r10-min", "dna-r9-prom", "dna-r9-min", "rna-004-prom", "rna-004-min"]))
@click.option("--show-advanced-options", is_flag=True, default=False, help="Show advanced options.")
@click.option("--noise-sampler", default=True, type=bool, show_default=True, hidden=True)
@click.option("--duration-sampler", default=True, type=bool, show_default=True, hidden=True)
@click.option("--dwell-mean", default=None, type=float, hidden=True)
@click.option("--dwell-std", default=0.0, type=float, hidden=True)
@click.option("--noise-std", default=2.0, type=float, hidden=True)
@click.option("--distr", default="expon", type=click.Choice(["expon", "beta", "gamma"]), hidden=True)
@click.option("--predict-batch-size", default=1024, type=int, hidden=True)
@click.option("--export-every-n-samples", default=1000000, type=int, hidden=True)
@click.option("--sample-rate", default=None, type=int, hidden=True)
@click.option("--bps", default=None, type=int, hidden=True)
@click.option("--digitisation", default=None, type=int, hidden=True)
@click.option("--range_val", default=None, type=float, hidden=True)
@click.option("--offset_mean", default=None, type=float, hidden=True)
@click.option("--offset_std", default=None, type=float, hidden=True)
@click.option("--median_before_mean", default=None, type=float, hidden=True)
@click.option("--median_before_std", default=None, type=float, hidden=True)
@click.option("--min_noise", default=0.0, type=float, hidden=True)
@click.option("--min_duration", default=3, type=int, hidden=True)
@click.option("--min_read_len", default=30, type=int, hidden=True)
@click.option("--preserve-read-ids", is_flag=True, default=False,
              help="Keep the input read ids instead of generated ones.")
@click.option("-s", "--seed", type=int, default=0, help="Seed for reproducibility (0 = random).")
@click.option("-m", "--model", type=click.Path(exists=False, dir_okay=False), help="Model weights (.ckpt).")
@click.option("-y", "--config", help="YAML configuration file overriding the defaults.")
@click.option("-v", "--verbosity", type=click.Choice(["debug", "info", "warning", "error"], case_sensitive=False), default="info")
@click.option("--compute-mode", default="f16x3", type=click.Choice(["f16x3", "f32", "f16"]), hidden=True,
              help="Decoder arithmetic of the MI355X engine.")
@click.option("--attention-path", default="auto", type=click.Choice(["auto", "fast", "exact"]), hidden=True,
              help="Softmax path of the split-f16 decoder: auto = chosen per checkpoint by the engine's calibration launch "
                   "(include/s2s_hip.h: s2s_set_attention_path); exact = the same time whatever the weights.")
@click.option("--gpus", default=1, type=int, hidden=True,
              help="Run on this many GPUs of the node: one process per GPU, the read set sharded, one OUT.rankN file per process "
                   "(the same as starting the command under torchrun --nproc-per-node N).")
@click.option("--keep-shards", is_flag=True, hidden=True,
              help="With --gpus N: leave the OUT.rankN files as they are instead of merging them into OUT.")
@click.option("--join", "join_mode", default="after", type=click.Choice(["after", "live"]), hidden=True,
              help="With --gpus N: after = the rank files are joined in rank order once the ranks are done (the layout of a single-process "
                   "file); live = the parent copies complete records / signal batches into OUT while the ranks run (round robin over the "
                   "ranks: same reads, another record order; .blow5 and .pod5).")
@click.pass_context
def predict(ctx, fasta, read_input, num_reads, read_length, coverage, out, profile, show_advanced_options, noise_sampler,
            duration_sampler, dwell_mean, dwell_std, noise_std, distr, predict_batch_size, export_every_n_samples,
            sample_rate, bps, digitisation, range_val, offset_mean, offset_std, median_before_mean, median_before_std,
            min_noise, min_duration, min_read_len, preserve_read_ids, seed, model, config, verbosity, compute_mode, attention_path,
            gpus, keep_shards, join_mode):
    """Generate nanopore signals from a reference genome (default) or from reads (--read-input)."""
    import os
    if gpus > 1 and "WORLD_SIZE" not in os.environ:
        # The reference leaves multi-GPU runs to Lightning (devices="auto", DDP: inference.py:430-445); here the command starts its
        # own ranks as CHILD processes -- before anything in this process has touched the GPU -- and returns their exit code.
        if fasta is None or out is None:
            click.echo("FASTA and -o/--out are required")
            ctx.exit(1)
        if str(out).endswith(".pod5") and os.path.exists(out) and not keep_shards:
            raise FileExistsError(f"{out} exists (the POD5 writer refuses to overwrite, like pod5.Writer)")
        import time
        t0 = time.time()
        live, make_live = None, None
        if join_mode == "live" and not keep_shards and not os.environ.get("S2S_DRY_LAUNCH"):
            if not str(out).endswith((".blow5", ".pod5")):
                raise click.UsageError("--join live handles .blow5 and .pod5 outputs")
            ext = os.path.splitext(str(out))[1]
            base_name = str(out)[:len(str(out)) - len(ext)]
            partial = base_name + ".partial" + ext
            shard_paths = [f"{base_name}.rank{r}{ext}" for r in range(gpus)]       # (parallel.rank_output_path, without its imports)
            for stale in [partial] + shard_paths:           # (a rank file left by an earlier run must not be mistaken for this run's)
                if os.path.exists(stale):
                    os.remove(stale)
            holder = {}

            def make_live():                                # (called once the ranks are started: numpy loads beside their start-up, not before it)
                from .merge import LiveJoin
                holder["live"] = LiveJoin(shard_paths, partial)
                return holder["live"]
        try:
            rc, timing, reap = _launch_ranks(gpus, make_live)
        except BaseException:
            if make_live is not None and holder.get("live") is not None:
                holder["live"].abort()          # (a rank file the live join could not make sense of: no partial output stays behind)
            raise
        live = holder.get("live") if make_live is not None else None
        try:
            timing["ranks_seconds"] = time.time() - t0
            if live is not None:
                if rc == 0:
                    t1 = time.time()
                    try:
                        n, st = live.finish(consume=True)
                        os.replace(live.out, str(out))
                    except BaseException:
                        live.abort()
                        raise
                    timing.update(merge_seconds=time.time() - t1, merge_bytes=st["bytes"], merge_bytes_copied=st["bytes"], reads=n, join="live",
                                  live_bytes=st["live_bytes"], live_copy_seconds=st["copy_seconds"], join_order=st["order"])
                    click.echo(f"{n} reads from {gpus} ranks -> {out}  [launch {round(timing.get('launch_seconds', 0), 2)} s, ranks "
                               f"{timing['ranks_seconds']:.2f} s in all with {st['live_bytes'] / 1e9:.2f} of {st['bytes'] / 1e9:.2f} GB joined meanwhile, "
                               f"{timing['merge_seconds']:.2f} s to finish the file; {st['order']}]")
                else:
                    live.abort()
            elif rc == 0 and not keep_shards and not os.environ.get("S2S_DRY_LAUNCH"):
                # one output file, as the reference writes (inference.py:65-79): the first rank's file becomes OUT, the payload of the
                # others moves in as byte ranges on copy threads (merge.py), the rank files are gone afterwards
                from .parallel import rank_output_path
                from .signal_io import merge_shards as _merge
                shards = [rank_output_path(str(out), r, gpus) for r in range(gpus)]
                t1 = time.time()
                n = _merge(shards, str(out), consume=True)
                timing["merge_seconds"] = time.time() - t1
                timing["merge_bytes"] = _merge.last.get("bytes", 0)
                timing["merge_bytes_copied"] = _merge.last.get("bytes_copied", 0)
                timing["reads"] = n
                launch = timing.get("launch_seconds")
                click.echo(f"{n} reads from {gpus} ranks -> {out}  [launch {launch if launch is None else round(launch, 2)} s, "
                           f"ranks {timing['ranks_seconds']:.2f} s in all, merge {timing['merge_seconds']:.2f} s for "
                           f"{timing['merge_bytes'] / 1e9:.2f} GB]")
            late = reap()                                       # (the merge did not wait for the ranks' teardown: see _launch_ranks)
            rc = rc or late
            timing["total_seconds"] = time.time() - t0
            try:
                import resource
                # (the peak counts the tmpfs pages of the output while they are mapped for the fill -- one shard's span, shared memory that
                #  belongs to the file, not to this process; what the process itself holds is the anonymous part)
                timing["parent_peak_rss_mb"] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
                for line in open("/proc/self/status"):
                    if line.startswith("RssAnon:"):
                        timing["parent_anon_rss_mb"] = int(line.split()[1]) / 1024.0
            except (ImportError, OSError):
                pass
            if os.environ.get("S2S_TIMING_JSON") and not os.environ.get("S2S_DRY_LAUNCH"):
                import json
                with open(os.environ["S2S_TIMING_JSON"], "w") as f:
                    json.dump(timing, f)
        except BaseException:
            reap(force=True)                    # (a failed join: no rank process and no s2s-ranks-* directory stays behind)
            raise
        ctx.exit(rc)
    # a rank of a multi-process run binds itself to its share of its GPU's socket FIRST: before torch starts its thread pools,
    # before the first page of a pinned staging buffer is touched (placement.pin_rank; S2S_NO_PIN=1 opts out)
    from .placement import pin_rank
    pinned = pin_rank()
    from .inference import inference_run
    from .utils import set_seeds, setup_logging

    if show_advanced_options:
        for p in ctx.command.params:
            p.hidden = False
        click.echo(ctx.get_help())
        ctx.exit()
    if not fasta or not out:
        logger.error("FASTA file and Output file are required for prediction.")
        ctx.exit(1)
    setup_logging(verbosity)
    logger.info("seq2squiggle (MI355X engine) version %s", str(__version__))
    if pinned:
        logger.debug(f"rank {os.environ.get('RANK', '0')}: bound to CPUs {pinned['cpus']} ({pinned['source']}"
                     + (f": NUMA node {pinned['numa_node']} of GPU {pinned['bdf']}" if pinned["source"] == "sysfs" else "") + ")")
    cfg = set_config(config)
    import time
    t_ready = time.time()                      # interpreter, torch and the library are loaded: what a rank pays before its first read
    from .parallel import shared_seed
    seed = set_seeds(shared_seed(seed))        # --seed 0 under torchrun: rank 0's fresh seed, for every rank
    inference_run(config=cfg, saved_weights=model, fasta=str(fasta), read_input=read_input, n=num_reads, r=read_length,
                  c=coverage, out=str(out), profile=profile, dwell_mean=dwell_mean, dwell_std=dwell_std, noise_std=noise_std,
                  noise_sampling=noise_sampler, duration_sampling=duration_sampler, distr=distr,
                  predict_batch_size=predict_batch_size, export_every_n_samples=export_every_n_samples,
                  sample_rate=sample_rate, bps=bps, digitisation=digitisation, range_val=range_val,
                  offset_mean=offset_mean, offset_std=offset_std, median_before_mean=median_before_mean,
                  median_before_std=median_before_std, min_noise=min_noise, min_duration=min_duration,
                  min_read_len=min_read_len, preserve_read_ids=preserve_read_ids, seed=seed, mode=compute_mode,
                  attention_path=attention_path)
    logger.info("Prediction finished.")
    if os.environ.get("S2S_TIMING_DIR"):       # a rank of `predict --gpus N`: when it was ready and when it was done, for the parent's summary
        import json
        stamp = os.path.join(os.environ["S2S_TIMING_DIR"], f"rank{os.environ.get('RANK', '0')}.json")
        with open(stamp + ".tmp", "w") as f:
            json.dump({"ready": t_ready, "done": time.time()}, f)
        os.replace(stamp + ".tmp", stamp)      # (appears whole: the parent starts the merge when it sees every rank's)
        # The file is closed and the parent has been told: leave without the interpreter's and torch's teardown (0.3 - 0.5 s per
        # rank that the parent would wait for before it may return; the driver reclaims the device memory of a process that ends)
        logging.shutdown()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


def _launch_ranks(gpus: int, make_live=None, cmd=None):
    """`predict --gpus N` outside torchrun: the same command line once per GPU, as N CHILD processes of this one (which never touches
    the GPU) with the environment torchrun would give them (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1,
    MASTER_PORT) -- started directly: the elastic agent of torch.distributed.run costs an import of torch in the parent and a
    rendezvous before the first rank starts, and there is nothing here for it to supervise.  The first rank that fails ends the
    others (by their pids).  S2S_DRY_LAUNCH=1 prints the child command and environment instead of running it.
    -> (exit code, {"launch_seconds": spawn -> the slowest rank is ready to read its input, "predict_seconds": ... -> the slowest is
    done}, reap): returns as soon as every rank's output file is complete and closed; reap() collects the processes afterwards.
    make_live() -> merge.LiveJoin, called once the ranks are started: its step() runs in the waiting loop, so that the rank files are
    joined while they grow."""
    import json
    import os
    import shutil
    import socket
    import subprocess
    import tempfile
    import time
    argv, skip = [], False
    for a in sys.argv[1:]:
        if skip:
            skip = False
            continue
        if a == "--gpus":
            skip = True
            continue
        if a.startswith("--gpus=") or a == "--keep-shards":
            continue
        argv.append(a)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = cmd or [sys.executable, "-m", "seq2squiggle_amd"] + argv       # (cmd: the supervision tests start stand-in ranks)
    from .placement import rank_visibility, visible_list
    parent_visible = visible_list()

    def rank_env(r):
        # one device per rank (HIP_VISIBLE_DEVICES = the r-th device this process may see): a rank cannot place anything on a
        # neighbour's GPU, whatever a call site passes; S2S_PARENT_VISIBLE lets every rank work out the same CPU table (placement.pin_rank)
        return {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(gpus), "LOCAL_WORLD_SIZE": str(gpus),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                "S2S_PARENT_VISIBLE": ",".join(parent_visible) if parent_visible is not None else "", **rank_visibility(r)}
    if os.environ.get("S2S_DRY_LAUNCH"):
        click.echo(json.dumps({"dry_launch": cmd, "rank_env": [rank_env(r) for r in range(gpus)]}))
        return 0, {}, lambda force=False: 0
    timing_dir = tempfile.mkdtemp(prefix="s2s-ranks-")
    base = dict(os.environ, S2S_TIMING_DIR=timing_dir)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.time()
    import signal

    class _Ended(BaseException):
        """SIGTERM / SIGHUP to the launcher: unwinds through the handlers below, which end the ranks."""

    def _on_signal(signum, frame):
        raise _Ended(signum)
    old_handlers = {}
    for sig in (signal.SIGTERM, signal.SIGHUP):
        try:                                     # (only the main thread may install handlers: tests call this from there too)
            old_handlers[sig] = signal.signal(sig, _on_signal)
        except (ValueError, OSError):
            pass
    procs = []
    try:
        for r in range(gpus):
            procs.append(subprocess.Popen(cmd, env=dict(base, **rank_env(r))))
    except BaseException:
        for p_ in procs:
            p_.kill()
        raise
    rc = 0

    def end_ranks(which, grace=float(os.environ.get("S2S_RANK_GRACE", "10"))):
        """terminate(), then kill() for whoever has not left after `grace` seconds (a rank stuck in a HIP call ignores SIGTERM)."""
        alive = [procs[q] for q in which if procs[q].poll() is None]
        for p_ in alive:
            p_.terminate()
        deadline = time.time() + grace
        for p_ in alive:
            try:
                p_.wait(timeout=max(0.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p_.kill()
        for p_ in alive:
            try:
                p_.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass

    def stamps():
        try:
            names = os.listdir(timing_dir)
            return [json.load(open(os.path.join(timing_dir, f))) for f in names if f.endswith(".json")]
        except (OSError, ValueError):
            return []

    def reap(force: bool = False) -> int:
        """Waits for the ranks that are still shutting down (interpreter, torch and HIP teardown: ~0.3 s each, which the merge
        does not have to wait for) -> the first non-zero exit code, or 0.  force: the command is failing -- end them now.
        Idempotent; also gives the signal handlers back and removes the stamp directory."""
        code = 0
        if force:
            end_ranks(range(len(procs)))
        for p_ in procs:
            try:
                code = code or p_.wait(timeout=120)
            except subprocess.TimeoutExpired:
                p_.kill()
                code = code or 1
        for sig, h in old_handlers.items():
            try:
                signal.signal(sig, h)
            except (ValueError, OSError):
                pass
        old_handlers.clear()
        shutil.rmtree(timing_dir, ignore_errors=True)
        return code
    live = None
    try:
        if make_live is not None:
            live = make_live()
        try:                                    # the merge's imports (numpy, pyarrow, the library) load while the ranks work
            from . import merge, pod5_io, signal_io  # noqa: F401
            pod5_io._pa()
            from ._lib import lib
            lib()
        except Exception:                       # (reported by the merge itself, if it comes to that)
            pass
        left = set(range(gpus))
        # a rank writes its stamp when its output file is complete and closed (end of `predict`): once every rank has, the files
        # can be joined -- the processes may still be tearing down
        while left:
            stamped = {f for f in os.listdir(timing_dir) if f.endswith(".json")}
            if len(stamped) >= gpus:
                break
            moved = live is not None and live.step([f"rank{r}.json" in stamped for r in range(gpus)])   # (--join live: copy what is complete)
            for r in sorted(left):
                code = procs[r].poll()
                if code is None:
                    continue
                left.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    logger.error(f"rank {r} exited with code {code}; ending the other ranks")
                    end_ranks(sorted(left))
            if left and not moved:
                time.sleep(0.005)
    except BaseException:
        reap(force=True)
        raise
    timing = {}
    rows = stamps() if rc == 0 else []
    if len(rows) < gpus and rc == 0 and not left:      # every rank exited cleanly; a stamp may have been written a moment ago
        rows = stamps()
    if len(rows) == gpus:
        timing = {"launch_seconds": max(x["ready"] for x in rows) - t0,
                  "predict_seconds": max(x["done"] for x in rows) - max(x["ready"] for x in rows)}
    elif rc == 0:
        rc = reap() or 1                              # a rank ended without its stamp: not a complete set of files
    return rc, timing, reap


@main.command("merge-shards")
@click.argument("shards", nargs=-1, required=True, type=click.Path(exists=True, dir_okay=False))
@click.option("-o", "--out", required=True, type=click.Path(dir_okay=False), help="Merged .blow5 / .slow5 / .pod5 file.")
@click.option("--threads", default=None, type=int, help="Copy threads (default: this process's CPU share).")
@click.option("--consume", is_flag=True, help="Turn the first shard into the output and delete the others (moves 1/N fewer bytes).")
def merge_shards(shards, out, threads, consume):
    """Join the OUT.rankN.blow5 (or .slow5 / .pod5) shard files of a multi-GPU run, in the order given, into one file: raw byte
    ranges on copy threads, nothing is decompressed.  (Read ids and read numbers already continue across the shards of one run.)"""
    from .signal_io import merge_shards as _merge
    n = _merge(list(shards), out, threads=threads, consume=consume)
    st = _merge.last
    click.echo(f"{n} records -> {out}  [{st.get('bytes', 0) / 1e9:.3f} GB in {st.get('seconds', 0):.2f} s]")


if __name__ == "__main__":
    main()
