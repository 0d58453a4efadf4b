#!/bin/bash
# PMC passes for the bench (run on the GPU box through gpurun).  Each counter group is its own
# rocprofv3 run (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2); --pmc is combined with
# --kernel-trace only.  Usage: tools/pmc_run.sh <out-dir-under-gpurun_out> [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --reads 210 --no-cpu-baseline $*"
# what was measured: the hash of the kernel sources of THIS snapshot (bench.py compares it with the library it runs: pmc_stale)
(cd $R && python3 -c "from seq2squiggle_amd import _build; print(_build.source_hash())") > $OUT/csrc_sha256.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS" \
  "TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU_MFMA_MOPS_F16" ; do
  i=$((i+1))
  # (eight SQ counters are what one pass can hold: a ninth makes rocprofv3 abort with "exceeds the capabilities of the hardware" and
  #  then sit in its signal handler -- hence the wall limit on every pass)
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT -o pmc$i -- python3 $R/bench.py $ARGS > $OUT/pmc$i.log 2>&1
  echo "pass $i ($grp): rc=$?"
done
ls $OUT
