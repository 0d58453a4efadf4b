#!/bin/bash
# distance to the fp64 oracle for several prebuilt library variants: tools/cmp_mae.sh n_reads lib1 lib2 ...
cd $GRAFT_REPO_ROOT
n=$1; shift
for lib in "$@"; do S2S_HIP_LIB=$PWD/$lib python tools/mae_libs.py $n 2>&1 | tail -1; done
