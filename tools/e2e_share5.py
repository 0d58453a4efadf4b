"""Where does one GPU's share of BASELINE configs[4] (synthetic 12.5 Mb reference -c 30 -r 10000 -> .pod5) spend its wall time
outside the kernel?  Prints the main thread's event gaps summed by kind (inference._TRACE)."""
import collections, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from seq2squiggle_amd import inference
from seq2squiggle_amd.utils import write_synthetic_reference

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
ext = sys.argv[2] if len(sys.argv) > 2 else "pod5"
bench._e2e_run("f16x3", 1000, ext); bench._e2e_run("f16x3", 1000, ext)      # warm: pinned buffers, pyarrow, thread pools
with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    ref = os.path.join(td, "ref.fasta")
    write_synthetic_reference(ref, [int(L * frac) for L in (4_000_000, 3_000_000, 2_500_000, 2_000_000, 1_000_000)])
    inference._TRACE = []
    t0 = time.perf_counter()
    el, chunks, size = bench._e2e_run("f16x3", -1, ext, None, ref, 10000, 30)
    ev = inference._TRACE
    inference._TRACE = None
print(f"{chunks} chunks in {el:.3f} s = {chunks / el:.3e} chunks/s, {size >> 20} MB")
first = {}
last = {}
main, writer = collections.Counter(), collections.Counter()
prev_m = prev_w = t0
WRITER = ("writer start", "writer end")
for name, t in ev:
    first.setdefault(name, t - t0)
    last[name] = t - t0
    if name in WRITER:                 # logged by the writer thread
        writer[name] += t - prev_w
        prev_w = t
    else:                              # logged by the thread that drives the GPU
        main[name] += t - prev_m
        prev_m = t
for k in ("reads ready", "model ready", "first read wanted", "predict queued", "launched", "done"):
    if k in first:
        print(f"  first {k:20s} {first[k]:8.3f} s   last {last[k]:8.3f} s")
print("  driving thread, time ending in each event (s):", {k: round(v, 3) for k, v in main.most_common(14)})
print("  writer thread: busy", round(writer["writer end"], 3), "s, idle", round(writer["writer start"], 3), "s")
big = []
prev = t0
for name, t in ev:
    if name not in WRITER:
        big.append((t - prev, t - t0, name))
        prev = t
print("  largest single gaps on the driving thread (ms, at s, ending in):", [(round(1e3 * g, 1), round(at, 3), n) for g, at, n in sorted(big, reverse=True)[:12] if n != "records"])
if len(sys.argv) > 3:                     # raw events of a window in the middle of the run
    mid = [i for i, (n, t) in enumerate(ev) if n == "pack"]
    lo = mid[len(mid) // 2]
    for name, t in ev[lo: lo + 60]:
        print(f"   {1e3 * (t - ev[lo][1]):8.2f} ms  {name}")
