"""The host library (csrc/s2s_host.cpp: thread pool, record packer, Huffman-only deflate, sampler replay, FASTA / FASTQ parsers)
under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer, driven by tools/fuzz_host.py with exact-size
malloc()ed buffers and every result checked against an independent implementation (zlib.decompress, the interpreter's line loop
and sampler, scipy).  CPU only -- GPU sanitizers do not exist on this pool.  Skipped where g++ or its sanitizer runtimes are missing."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    gxx = shutil.which("g++")
    if not gxx:
        return None
    p = subprocess.run([gxx, f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.parametrize("which,runtime", [("asan", "libasan.so"), ("tsan", "libtsan.so")])
def test_host_library_under_sanitizers(which, runtime, tmp_path):
    if _runtime(runtime) is None:
        pytest.skip(f"g++ / {runtime} not available")
    env = dict(os.environ, S2S_SAN_DIR=str(tmp_path))
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize_host.sh"), which, "24"], capture_output=True, text=True,
                       env=env, timeout=900, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-4000:]
    for cannot_start in ("unexpected memory mapping", "Shadow memory range interleaves", "failed to allocate", "ReserveShadowMemoryRange failed"):
        if cannot_start in tail:           # the sanitizer runtime itself cannot run on this kernel / address-space layout
            pytest.skip(f"{runtime} cannot start here: {cannot_start}")
    assert r.returncode == 0 and f"FUZZ_OK {which}" in r.stdout and f"SANITIZE_OK {which}" in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail and "WARNING: ThreadSanitizer" not in tail, tail
