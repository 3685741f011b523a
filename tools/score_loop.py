"""C3 scoring pass in a loop (profiling target of tools/pmc_instmix.sh): python3 tools/score_loop.py [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
genes = [synth.simulate_alignment(50, 1000, 1 + i, 0.8) for i in range(128)]
ctx = engine.Context(0)
b = engine.Batch(ctx, [(g[0], g[1]) for g in genes], [g[2] for g in genes], alpha=0.8)
for _ in range(n): l = b.score()
print("lnL[0] %.6f" % l[0])
