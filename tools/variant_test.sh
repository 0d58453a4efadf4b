#!/bin/bash
# build the library with extra -D flags and run the f16x3 parity tests against it (GPU box)
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  python -c "import sys; from seq2squiggle_amd import _build; _build.compile_to(sys.argv[1], sys.argv[2:])" $PWD/seq2squiggle_amd/lib/libs2s_var.so $flags 2>&1 | grep error
  echo "== flags: $flags"
  S2S_HIP_LIB=$PWD/seq2squiggle_amd/lib/libs2s_var.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "${S2S_TEST_K:-f16x3 and (stage or modes)}" 2>&1 | tail -1
done
