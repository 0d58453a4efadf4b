#!/bin/bash
# CPU sanitizer builds of the host library (csrc/s2s_host.cpp alone: thread pool, record packer, Huffman-only deflate, sampler
# replay, FASTA / FASTQ parsers -- no GPU code) and the ctypes fuzz driver under them.
#   tools/sanitize_host.sh [asan|tsan|all] [rounds]
# asan = -fsanitize=address,undefined; tsan = -fsanitize=thread (the Pool and everything that runs on it).  The driver allocates
# every buffer it hands over with malloc() at its exact size, so an off-by-one read or write lands in a red zone.
# GPU sanitizers are not available on this pool; this is the host half only.
set -e
cd "$(dirname "$0")/.."
what=${1:-all}; rounds=${2:-60}
out=${S2S_SAN_DIR:-/tmp/s2s_sanitize}; mkdir -p "$out"
src=seq2squiggle_amd/csrc/s2s_host.cpp
common="-O1 -g -std=c++17 -fPIC -shared -fno-omit-frame-pointer -ffp-contract=off -Wno-unknown-pragmas"
run() {   # $1 = tag, $2 = sanitizer flags, $3 = runtime library to preload, $4 = extra env
  g++ $common $2 -o "$out/libs2s_host_$1.so" $src -lz -ldl -lpthread
  env LD_PRELOAD="$(g++ -print-file-name=$3)" PYTHONMALLOC=malloc $4 \
      python tools/fuzz_host.py "$out/libs2s_host_$1.so" "$rounds" "$1"
}
if [ "$what" = asan ] || [ "$what" = all ]; then
  run asan "-fsanitize=address,undefined -fno-sanitize-recover=undefined" libasan.so "ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1"
fi
if [ "$what" = tsan ] || [ "$what" = all ]; then
  run tsan "-fsanitize=thread" libtsan.so "TSAN_OPTIONS=halt_on_error=1:exitcode=66:report_signal_unsafe=0"
fi
echo "SANITIZE_OK $what"
