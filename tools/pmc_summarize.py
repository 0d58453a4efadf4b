#!/usr/bin/env python
"""Condense the rocprofv3 --pmc passes written by tools/pmc_run.sh into one JSON:
python tools/pmc_summarize.py mode=dir[:chunks] [mode=dir[:chunks] ...] > profiles/rNN/pmc_summary.json
For every kernel the counters of its FIRST dispatch in each pass (tools/pmc_run.sh: bench.py --reads 210 = one launch of
65,520 chunks; pass that count as :chunks so that bench.py can scale the counters per chunk); Grid_Size / Workgroup_Size /
Scratch_Size / VGPR_Count are kept so the launch can be identified."""
import csv, glob, json, os, sys

def _is_test_instance(name):
    """s2s_fused_kernel<MODE, TEST, EXACT>: TEST = true is s2s_create's 512-chunk calibration launch (and the parity tests' instance)."""
    if "s2s_fused_kernel<" not in name:
        return False
    args = [a.strip() for a in name.split("s2s_fused_kernel<", 1)[1].split(">")[0].split(",")]
    return len(args) > 1 and args[1] == "true"


import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
meta = {"csrc_sha256": None, "git_commit": None, "by_mode": {}}
try:
    meta["git_commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    meta["git_dirty_csrc"] = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "seq2squiggle_amd/csrc", "include"],
                                                 capture_output=True, text=True).stdout.strip())
except OSError:
    pass
for arg in sys.argv[1:]:
    mode, d = arg.split("=", 1)
    d, _, chunks = d.partition(":")
    per = {}
    for path in sorted(glob.glob(os.path.join(d, "pmc*_counter_collection.csv"))):
        seen = {}
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0][:110]
            first = seen.setdefault(name, row["Dispatch_Id"])
            if row["Dispatch_Id"] != first:
                continue
            k = per.setdefault(name, {})
            k[row["Counter_Name"]] = float(row["Counter_Value"])
            k.setdefault("_launch", {"grid": int(row["Grid_Size"]), "workgroup": int(row["Workgroup_Size"]),
                                     "scratch_bytes_per_lane": int(row["Scratch_Size"]), "vgprs": int(row["VGPR_Count"]),
                                     "lds_bytes": int(row["LDS_Block_Size"]),
                                     # (the TEST instance of the fused kernel is s2s_create's calibration launch: 512 chunks)
                                     **({"chunks": 512 if _is_test_instance(name) else int(chunks)} if chunks else {})})
    # wall time of that first dispatch in every pass (pmcN_kernel_trace.csv), so that a counter can be turned into a rate: the
    # clock the chip held in the pass that counted GRBM_GUI_ACTIVE is GRBM_GUI_ACTIVE / 8 XCDs / that pass's duration
    for path in sorted(glob.glob(os.path.join(d, "pmc*_kernel_trace.csv"))):
        tag = os.path.basename(path).split("_")[0]
        seen = set()
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0][:110]
            if name in seen or name not in per:
                continue
            seen.add(name)
            per[name].setdefault("_duration_ns", {})[tag] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    for name, k in per.items():
        if "GRBM_GUI_ACTIVE" in k:
            for path in sorted(glob.glob(os.path.join(d, "pmc*_counter_collection.csv"))):
                tag = os.path.basename(path).split("_")[0]
                if any(r["Counter_Name"] == "GRBM_GUI_ACTIVE" for r in csv.DictReader(open(path))) and tag in k.get("_duration_ns", {}):
                    k["_effective_clock_ghz"] = k["GRBM_GUI_ACTIVE"] / 8 / k["_duration_ns"][tag]
                    break
    out[mode] = per
    try:                                  # tools/pmc_run.sh wrote the hash of the sources the passes ran on
        meta["by_mode"][mode] = open(os.path.join(d, "csrc_sha256.txt")).read().strip()
    except OSError:
        meta["by_mode"][mode] = None
hashes = {h for h in meta["by_mode"].values() if h}
meta["csrc_sha256"] = hashes.pop() if len(hashes) == 1 else None          # one hash for all modes, or none
out["_meta"] = meta
json.dump(out, sys.stdout, indent=1)
