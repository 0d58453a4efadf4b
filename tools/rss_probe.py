"""Where does the host memory of a long streaming run go?  Runs inference_run on a synthetic reference (-c 30 -r 10000) into
.pod5 / .blow5 and samples the process's RSS beside the allocators' own counters (torch pinned-host cache, pyarrow pool,
glibc arenas) every half second.   python tools/rss_probe.py pod5 0.125 [/dev/shm]"""
import ctypes
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import psutil  # noqa: E402
import torch  # noqa: E402

from seq2squiggle_amd.cli import set_config  # noqa: E402
from seq2squiggle_amd.inference import inference_run  # noqa: E402
from seq2squiggle_amd.utils import set_seeds, write_synthetic_reference  # noqa: E402


def main():
    ext = sys.argv[1] if len(sys.argv) > 1 else "pod5"
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.125
    where = sys.argv[3] if len(sys.argv) > 3 else None
    proc = psutil.Process()
    stop = threading.Event()
    samples = []

    def host_stats():
        try:
            st = torch.cuda.host_memory_stats()
            return st.get("allocated_bytes.current", 0) >> 20, st.get("reserved_bytes.current", st.get("segment.current", 0)) >> 20
        except Exception:
            return -1, -1

    def watch():
        import pyarrow as pa
        t0 = time.perf_counter()
        while not stop.is_set():
            m = proc.memory_info()
            samples.append((time.perf_counter() - t0, m.rss >> 20, getattr(m, "shared", 0) >> 20, pa.total_allocated_bytes() >> 20) + host_stats())
            time.sleep(0.5)
    with tempfile.TemporaryDirectory(dir=where) as td:
        ref = os.path.join(td, "ref.fasta")
        write_synthetic_reference(ref, [int(L * frac) for L in (4_000_000, 3_000_000, 2_500_000, 2_000_000, 1_000_000)])
        set_seeds(42)
        th = threading.Thread(target=watch, daemon=True)
        th.start()
        t0 = time.perf_counter()
        m = inference_run(config=set_config(None), saved_weights=os.path.join(ROOT, "tests", "golden", "synthetic_k9.ckpt"),
                          fasta=ref, read_input=False, n=-1, r=10000, c=30, out=os.path.join(td, "o." + ext), profile="dna-r10-prom",
                          dwell_mean=None, dwell_std=0.0, noise_std=2.0, noise_sampling=True, duration_sampling=True,
                          distr="expon", predict_batch_size=1024, export_every_n_samples=1000000, sample_rate=None,
                          bps=None, digitisation=None, range_val=None, offset_mean=None, offset_std=None,
                          median_before_mean=None, median_before_std=None, min_noise=0.0, min_duration=3, min_read_len=30,
                          preserve_read_ids=False, seed=42)
        el = time.perf_counter() - t0
        stop.set()
        th.join()
        size = os.path.getsize(os.path.join(td, "o." + ext))
    print(f"{ext} frac {frac}: {m.chunks_done} chunks in {el:.2f} s, output {size >> 20} MB")
    print("  t[s]   rss  shared  arrow  pinned_alloc  pinned_reserved   (MB)")
    for s in samples[:: max(1, len(samples) // 24)] + samples[-1:]:
        print("  %5.1f %6d %6d %6d %8d %10d" % s)
    try:
        ctypes.CDLL("libc.so.6").malloc_stats()
    except Exception as e:
        print(e)


if __name__ == "__main__":
    main()
