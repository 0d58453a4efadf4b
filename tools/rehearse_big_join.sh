#!/bin/bash
# The join at BASELINE configs[4] scale on ONE GPU box: `predict --gpus N` on a synthetic N x 12.5 Mb reference, -c 30 -r 10000 -> ONE
# .pod5 of N x 7.1 GB (every rank on cuda:0: the predict phase is N times longer than on N GPUs, the join is the real size).
# Reports the command's own launch / ranks / merge seconds, the parent's peak RSS, and checks the merged file's read count and a
# sample of the reads.      tools/rehearse_big_join.sh [N=5] [out_dir] [dir=/dev/shm] [join=after|live]
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
N=${1:-5}; out=${2:-gpurun_out/big_join}; base=${3:-/dev/shm}; join=${4:-after}; mkdir -p $out
tmp=$(mktemp -d -p $base); trap "rm -rf $tmp" EXIT
python -c "
from seq2squiggle_amd.utils import write_synthetic_reference
print('reference:', write_synthetic_reference('$tmp/ref.fasta', [2_500_000] * (5 * $N)), 'bases')" | tee $out/README.txt
( time env S2S_ONE_GPU=1 S2S_TIMING_JSON=$out/timing.json timeout -k 10 1000 python -m seq2squiggle_amd predict $tmp/ref.fasta -c 30 -r 10000 \
    -m tests/golden/synthetic_k9.ckpt --seed 11 -o $tmp/multi.pod5 --gpus $N --join $join ) > $out/predict.log 2>&1 || { echo "predict --gpus $N failed"; tail -20 $out/predict.log; exit 1; }
grep -E "reads from|^real" $out/predict.log | tee -a $out/README.txt
ls -l $tmp | tee -a $out/README.txt
python - $tmp/multi.pod5 $out/timing.json <<'PY' | tee -a $out/README.txt
import json, os, sys, time
from seq2squiggle_amd import pod5_io
from seq2squiggle_amd.codecs import vbz_decompress
t = time.perf_counter()
n = samples = 0
last = -1
for r, rows in pod5_io.iter_pod5(sys.argv[1], decode=False):      # (stored VBZ rows; every 997th read is decoded)
    n += 1
    if n % 997 == 0:
        got = sum(len(vbz_decompress(blob, cnt)) for blob, cnt in rows)
        assert got == r["num_samples"] and r["read_number"] == n - 1
        samples += got
    assert r["read_number"] > last and sum(c for _, c in rows) == r["num_samples"]
    last = r["read_number"]
tm = json.load(open(sys.argv[2]))
size = os.path.getsize(sys.argv[1])
print(f"merged .pod5: {n} reads, {size / 1e9:.2f} GB, read numbers consecutive, row sample counts add up, {samples} samples of every 997th read decoded ({time.perf_counter() - t:.1f} s)")
print("timing:", {k: round(v, 2) if isinstance(v, float) else v for k, v in tm.items()})
if tm.get("join") == "live":
    print(f"join live: {tm['live_bytes'] / 1e9:.2f} of {tm['merge_bytes'] / 1e9:.2f} GB copied while the ranks ran, {tm['merge_seconds']:.2f} s to finish the file afterwards")
else:
    print(f"join: {tm['merge_bytes'] / tm['merge_seconds'] / 1e9:.2f} GB/s of output ({tm['merge_bytes_copied'] / 1e9:.2f} GB moved)")
PY
