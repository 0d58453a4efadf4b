"""Build the HIP shared library in-tree (hipcc cross-compiles for gfx950 without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "s2s_hip.hip")              # kernels + the C ABI of the GPU path
SRC_HOST = os.path.join(HERE, "csrc", "s2s_host.cpp")       # host-only helpers (record framing / compression threads)


def deps():
    """Every file the library is compiled from: csrc/*.hip, csrc/*.cpp, csrc/*.h, include/*.h."""
    import glob
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")) + glob.glob(os.path.join(HERE, "csrc", "*.cpp")) +
                  glob.glob(os.path.join(HERE, "csrc", "*.h")) +
                  glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h")))

LIB = os.path.join(HERE, "lib", "libs2s_hip.so")


def source_hash() -> str:
    """sha256 over every file the library is compiled from (deps(): path relative to the repository, NUL, content): what ties a
    committed profile (profiles/rNN/pmc_summary.json: `_meta.csrc_sha256`) to the kernel source it was measured on."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(HERE)
    for d in deps():
        h.update(os.path.relpath(d, root).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in deps())


def compile_to(out: str, extra=(), verbose: bool = False, report: bool = False):
    """hipcc -> `out` with extra flags (tools build -D variants of the library this way).  report=True: -> the compiler's
    kernel-resource-usage remarks parsed per kernel (resource_usage) instead of the path."""
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC", *extra, "-o", out,
           SRC, SRC_HOST, "-lz", "-ldl", "-lpthread"]
    if verbose or report:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    if report:
        return resource_usage(subprocess.run(cmd, check=True, capture_output=True, text=True).stderr)
    subprocess.run(cmd, check=True)
    return out


def resource_usage(remarks: str) -> dict:
    """hipcc -Rpass-analysis=kernel-resource-usage output -> {mangled kernel name: {"VGPRs", "ScratchSize [bytes/lane]",
    "VGPRs Spill", "SGPRs Spill", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", ...: int}}."""
    import re
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        text = m.group(1)
        if text.startswith("Function Name:"):
            cur = out.setdefault(text.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in text:
            key, val = text.rsplit(":", 1)
            try:
                cur[key.strip()] = int(val)
            except ValueError:
                cur[key.strip()] = val.strip()
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    return compile_to(LIB, verbose=verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
