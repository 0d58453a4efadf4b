#!/bin/bash
# Everything profiles/rNN holds, in one GPU-box call: tools/profile_round.sh <tag>
#   gpurun_out/<tag>/stats_{f16x3,f32}: rocprofv3 --kernel-trace --stats of bench.py --steps 3 (+ the bench line it printed)
#   gpurun_out/<tag>/pmc_{f16x3,f32}:   the PMC passes of tools/pmc_run.sh
#   gpurun_out/<tag>/bench_f16x3.json:  the plain default bench line (CPU baseline + end-to-end included)
#   gpurun_out/<tag>/power_<mode>.txt:   rocm-smi power / clock samples under the kernel (tools/power_probe.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$1
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
for m in f16x3 f32; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/stats_$m -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --mode $m > $R/gpurun_out/$T/bench_${m}_under_rocprof.json 2> $R/gpurun_out/$T/stats_$m.err
  bash $R/tools/pmc_run.sh $T/pmc_$m --mode $m
done
# the exact attention instance (s2s_fused_kernel<1, false, true>) forced on the same workload: kernel stats only
S2S_ATTENTION_PATH=exact timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/stats_f16x3_exact -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --mode f16x3 > $R/gpurun_out/$T/bench_f16x3_exact_under_rocprof.json 2> $R/gpurun_out/$T/stats_f16x3_exact.err
S2S_ATTENTION_PATH=exact bash $R/tools/pmc_run.sh $T/pmc_f16x3_exact --mode f16x3
cd $R && python3 bench.py > gpurun_out/$T/bench_f16x3.json 2> gpurun_out/$T/bench_f16x3.err
python3 tools/pmc_summarize.py f32=gpurun_out/$T/pmc_f32:65520 f16x3=gpurun_out/$T/pmc_f16x3:65520 f16x3_exact=gpurun_out/$T/pmc_f16x3_exact:65520 > gpurun_out/$T/pmc_summary.json
# the un-profiled diagnostic build: per-wave phase stamps + the clock the SIMDs held (s_memtime / s_memrealtime)
S2S_DIAG_HEAT=3 python3 tools/diag_phases.py f16x3 > gpurun_out/$T/diag_phases.txt 2>&1
S2S_DIAG_HEAT=2 python3 tools/diag_phases.py f32 > gpurun_out/$T/diag_phases_f32.txt 2>&1
python3 - gpurun_out/$T <<'PY'
import json, sys
d = {}
for f in ("diag_phases.txt", "diag_phases_f32.txt"):
    for line in open(sys.argv[1] + "/" + f):
        if line.startswith("DIAGJSON "):
            j = json.loads(line[9:])
            d[j["mode"]] = {"ghz": j["ghz"], "cycles_per_chunk_and_cu": j["cycles_per_chunk_and_cu"]}
json.dump(d, open(sys.argv[1] + "/diag_clock.json", "w"))
PY
timeout -k 10 120 tools/probes/pass_probe > gpurun_out/$T/pass_probe.txt 2>&1
# socket power + shader clock under the kernel (rocm-smi), per mode
for m in f16x3 f32 f16; do timeout -k 10 120 bash tools/power_probe.sh $m > /dev/null 2>&1; cp gpurun_out/power_$m.txt gpurun_out/$T/; done
ls gpurun_out/$T gpurun_out/$T/stats_f16x3
