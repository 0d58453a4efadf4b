#!/bin/bash
# cycles per chunk and CU + the clock the SIMDs held, for several -D flag sets (diagnostic builds):
#   tools/diag_cycles.sh "flags1" "flags2" ...      (a flag set may start with "mode=f16 " / "mode=f32 ")
cd ${GRAFT_REPO_ROOT:-/root/repo}
for f in "$@"; do
  mode=f16x3
  flags="$f"
  case "$f" in mode=*) mode=${f%% *}; mode=${mode#mode=}; flags=${f#* }; [ "$flags" = "$f" ] && flags="";; esac
  S2S_DIAG_FLAGS="$flags" S2S_DIAG_HEAT=${S2S_DIAG_HEAT:-1} python tools/diag_phases.py $mode 2>&1 | grep "kernel per wave" | sed "s|^|[$f] |"
done
