#!/bin/bash
# Rehearsal of the N > 1 paths at the largest rank count the GPU pool allows on ONE device (its process guard kills a run with
# more than 6 GPU processes, the launcher counts as one: 5 ranks, not 8): bench.py --gpus N and `predict --gpus N` with every rank on cuda:0.
# THIS IS NOT A SCALING MEASUREMENT -- the ranks share one GPU; it exercises what an 8-GPU node exercises on the host:
# N child ranks (bench: under torch.distributed.run; predict --gpus N: started directly by the CLI), cpu_share() = quota /
# LOCAL_WORLD_SIZE threads per rank, N shard files at once, the gloo barriers of the sharded leg, the N-way byte-range merge
# (the CLI prints its launch / ranks / merge seconds).       tools/rehearse_ranks.sh [N=5] [out_dir]
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
N=${1:-5}; out=${2:-gpurun_out/rehearsal}; mkdir -p $out
echo "NOT A SCALING MEASUREMENT: $N ranks on ONE GPU (host-side rehearsal of the multi-GPU paths)" | tee $out/README.txt
S2S_BENCH_ONE_GPU=1 timeout -k 10 900 python bench.py --gpus $N --steps 3 --warmup 1 > $out/bench_${N}ranks_one_gpu.json 2> $out/bench_${N}ranks_one_gpu.err || { echo "bench --gpus $N failed"; tail -5 $out/bench_${N}ranks_one_gpu.err; exit 1; }
python - $out/bench_${N}ranks_one_gpu.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
s = d.get("end_to_end_sharded", {})
print("bench: n_ranks_seen", d["n_ranks_seen"], "| sharded leg:", {k: s.get(k) for k in ("seconds", "chunks_per_sec", "per_rank_seconds", "per_rank_cpu_share_threads", "merge_seconds", "merge", "with_merge", "one_command", "error", "skipped")})
PY
# 1/8-scale BASELINE configs[4]: a 12.5 Mb synthetic reference (one GPU's share of the 100 Mb job), -c 30 -r 10000 -> .pod5, N ranks, merged
tmp=$(mktemp -d -p ${S2S_REHEARSE_TMP:-/tmp}); trap "rm -rf $tmp" EXIT
python -c "
from seq2squiggle_amd.utils import write_synthetic_reference
print('reference:', write_synthetic_reference('$tmp/ref.fasta', [2_000_000, 1_500_000, 1_250_000, 1_000_000, 500_000]), 'bases')"
common="$tmp/ref.fasta -c 30 -r 10000 -m tests/golden/synthetic_k9.ckpt --seed 11"
( time S2S_ONE_GPU=1 timeout -k 10 900 python -m seq2squiggle_amd predict $common -o $tmp/multi.pod5 --gpus $N ) > $out/predict_${N}ranks_one_gpu.log 2>&1 || { echo "predict --gpus $N failed"; tail -20 $out/predict_${N}ranks_one_gpu.log; exit 1; }
( time timeout -k 10 900 python -m seq2squiggle_amd predict $common -o $tmp/single.pod5 ) > $out/predict_single.log 2>&1 || { echo "single predict failed"; tail -20 $out/predict_single.log; exit 1; }
python - $tmp/multi.pod5 $tmp/single.pod5 $N <<'PY' | tee -a $out/README.txt
import sys, os, numpy as np
from seq2squiggle_amd import pod5_io
a, b = pod5_io.read_pod5(sys.argv[1])["reads"], pod5_io.read_pod5(sys.argv[2])["reads"]
same = len(a) == len(b) and all(np.array_equal(x["signal"], y["signal"]) and x["read_number"] == y["read_number"] for x, y in zip(a, b))
print(f"predict --gpus {sys.argv[3]} (one GPU) -> merged .pod5: {len(a)} reads, {os.path.getsize(sys.argv[1])} bytes; single process: {len(b)} reads; per-read samples and numbering equal: {same}")
sys.exit(0 if same else 1)
PY
grep -E "real|reads from" $out/predict_${N}ranks_one_gpu.log | tail -5 | tee -a $out/README.txt
( time S2S_ONE_GPU=1 timeout -k 10 900 python -m seq2squiggle_amd predict $common -o $tmp/live.pod5 --gpus $N --join live ) > $out/predict_${N}ranks_one_gpu_join_live.log 2>&1 || { echo "predict --gpus $N --join live failed"; tail -20 $out/predict_${N}ranks_one_gpu_join_live.log; exit 1; }
python - $tmp/live.pod5 $tmp/single.pod5 <<'PY' | tee -a $out/README.txt
import sys, numpy as np
from seq2squiggle_amd import pod5_io
a, b = pod5_io.read_pod5(sys.argv[1])["reads"], pod5_io.read_pod5(sys.argv[2])["reads"]
same = len(a) == len(b) and all(np.array_equal(x["signal"], y["signal"]) and x["read_number"] == y["read_number"] and x["read_id"] == y["read_id"] for x, y in zip(a, b))
print(f"--join live: {len(a)} reads; per-read ids, samples and numbering equal to the single-process file: {same}")
sys.exit(0 if same else 1)
PY
grep -E "real|reads from" $out/predict_${N}ranks_one_gpu_join_live.log | tail -3 | sed 's/^/join live: /' | tee -a $out/README.txt
grep -E "real" $out/predict_single.log | tail -2 | sed 's/^/single process: /' | tee -a $out/README.txt
