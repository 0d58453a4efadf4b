"""Build-time guard: every PRODUCTION instance of the predict kernel (s2s_fused_kernel<MODE, TEST = false, EXACT>)
must compile without scratch memory or spilled vector registers -- round 4's key interleave pushed the exact instance to 32 B of
scratch per lane and 7 spills without anybody noticing until a review ran the compiler with remarks on.  hipcc cross-compiles
for gfx950 without a GPU (about 35 s)."""
import os
import re

from seq2squiggle_amd import _build


def test_production_kernel_instances_have_no_scratch(tmp_path):
    usage = _build.compile_to(str(tmp_path / "libcheck.so"), report=True)
    fused = {k: v for k, v in usage.items() if "s2s_fused_kernel" in k}
    # _Z16s2s_fused_kernelILi<MODE>ELb<TEST>ELb<EXACT>EE...
    inst = {}
    for name, u in fused.items():
        m = re.match(r"_Z16s2s_fused_kernelILi(\d)ELb([01])ELb([01])EE", name)
        assert m, name
        inst[(int(m.group(1)), bool(int(m.group(2))), bool(int(m.group(3))))] = u
    production = {k: u for k, u in inst.items() if not k[1]}
    assert set(production) == {(0, False, False), (1, False, False), (3, False, False), (1, False, True), (3, False, True)}, sorted(inst)
    for k, u in production.items():
        assert u["ScratchSize [bytes/lane]"] == 0 and u["VGPRs Spill"] == 0, (k, u)
        assert u["Occupancy [waves/SIMD]"] == 2 and u["VGPRs"] <= 256, (k, u)            # 512 threads per CU: two waves per SIMD
        # 256 B of static LDS (the production counters) in front of the dynamic region: together they must fit the CU's 160 KB
        assert u["LDS Size [bytes/block]"] == 256, (k, u)
    # the export / codec kernels as well
    for name, u in usage.items():
        if "s2s_fused_kernel" not in name:
            assert u["ScratchSize [bytes/lane]"] == 0 and u["VGPRs Spill"] == 0, (name, u)
