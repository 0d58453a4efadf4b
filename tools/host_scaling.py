"""How many host threads does one rank need to keep its GPU busy?  On an 8-GPU node every rank gets cpu_share() = CPU quota /
LOCAL_WORLD_SIZE threads for record compression (signal_io.cpu_share); this runs one rank's share of BASELINE configs[2]
(lambda genome -n 12500 -r 5000) end to end with the thread count forced to 1, 2, 4, 8, 16 (S2S_CPU_SHARE) and prints the
rate next to the resident-input kernel rate -- the predictor of end-to-end scaling that a one-GPU box can measure.
    python tools/host_scaling.py [blow5|pod5 ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    exts = sys.argv[1:] or ["blow5", "pod5"]
    print("container  threads  seconds  chunks/s  output MB")
    for ext in exts:
        for t in (1, 2, 4, 8, 16):
            env = dict(os.environ, S2S_CPU_SHARE=str(t))
            code = ("import json,sys;sys.path.insert(0,%r);import bench;"
                    "bench._e2e_run('f16x3',1000,%r);el,ch,size=bench._e2e_run('f16x3',12500,%r);"
                    "print(json.dumps(dict(el=el,ch=ch,size=size)))" % (ROOT, ext, ext))
            r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                print(ext, t, "FAILED", r.stderr[-400:])
                continue
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            print(f"{ext:9s} {t:7d} {d['el']:8.3f} {d['ch'] / d['el']:9.3e} {d['size'] / 2 ** 20:9.0f}", flush=True)


if __name__ == "__main__":
    main()
