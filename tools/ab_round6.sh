#!/bin/bash
# Round 6 A/B of library variants on ONE GPU box, back to back: error against the fp64 oracle on both attention paths, then speed
# (bench.py --no-cpu-baseline: chunks/s, kernel ms, the kernel's own cycle counter) on both paths, two rounds.
#   tools/ab_round6.sh out_dir lib1 lib2 ...
cd $GRAFT_REPO_ROOT
out=$1; shift; mkdir -p $out
for lib in "$@"; do for path in fast exact; do
  S2S_ATTENTION_PATH=$path S2S_HIP_LIB=$PWD/$lib python tools/mae_libs.py 2 2>&1 | tail -1 | sed "s/^/$path /"
done; done | tee $out/mae.txt
for rep in 1 2; do for path in fast exact; do for lib in "$@"; do
  S2S_ATTENTION_PATH=$path S2S_HIP_LIB=$PWD/$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); l=d['roofline']['live']; print('$path $lib', round(d['chunks_per_sec']), round(d['roofline']['avg_launch_ms'],3), round(l['cycles_per_chunk_and_cu']), round(l['in_kernel_clock_ghz'],3), l['softmax_redo_rate'])"
done; done; done | tee $out/speed.txt
