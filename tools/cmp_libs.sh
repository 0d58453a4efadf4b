#!/bin/bash
# bench several prebuilt library variants back to back on one GPU box: tools/cmp_libs.sh mode lib1 lib2 ...
cd $GRAFT_REPO_ROOT
mode=$1; shift
for rep in 1 2; do for lib in "$@"; do
  S2S_HIP_LIB=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --mode $mode 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['chunks_per_sec'], d['roofline']['avg_launch_ms'], d['roofline'].get('live'))"
done; done
