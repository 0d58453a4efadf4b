#!/usr/bin/env python
"""cProfile of bench.py's end_to_end leg (FASTA -> BLOW5 for lambda -n 1000 -r 5000), second run (warm)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
print(bench.end_to_end("f16x3")["seconds"])
pr = cProfile.Profile(); pr.enable(); r = bench.end_to_end("f16x3"); pr.disable()
print(r["seconds"])
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
