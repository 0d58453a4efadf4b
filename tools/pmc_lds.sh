#!/bin/bash
# SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS / SQ_LDS_IDX_ACTIVE of the fused kernel for several library builds:
#   tools/pmc_lds.sh <out-dir-under-gpurun_out> lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  S2S_HIP_LIB=$R/$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT -o lds_$tag -- python3 $R/bench.py --steps 1 --warmup 1 --reads 210 --no-cpu-baseline > $OUT/lds_$tag.log 2>&1
  python3 - $OUT $tag <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
tot = collections.Counter()
n = 0
for f in glob.glob(f"{out}/**/lds_{tag}_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fused" in r["Kernel_Name"] and ", true," not in r["Kernel_Name"].split("(")[0]:     # (not s2s_create's 512-chunk calibration launch on the TEST instance)
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            n += 1
chunks = 65520 * 2            # warm-up + timed launch of 210 reads x 312 chunks
print(tag, {k: round(v / chunks, 1) for k, v in tot.items()}, "per chunk")
PY
done
