"""diagnostic: does the cached-plan replay (score after set_alpha) always use the NEW rates?  cycles through alphas and compares
every replayed lnL with the value a fresh batch gives for that alpha"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pepr_amd import engine, synth
ctx = engine.Context(0)
if os.environ.get("BIG_FIRST"):
    big = synth.simulate_genes(128, 50, 1000)
    b = engine.Batch(ctx, [(g[0], g[1]) for g in big], [g[2] for g in big], alpha=1.0); b.score(); b.optimize(True, 0.1); b.close()
    print("big batch done", flush=True)
genes = [synth.simulate_alignment(9 + i % 4, 160 + 10 * i, 740 + i) for i in range(8)]
alphas = [0.3, 0.5889, 0.8, 1.3, 2.1, 0.4245, 1.698]
for sel in ([7], [2], list(range(8))):
    G = [(genes[i][0], genes[i][1]) for i in sel]; NW = [genes[i][2] for i in sel]
    ref = {}
    for a in alphas:
        bb = engine.Batch(ctx, G, NW, alpha=a); ref[a] = bb.score().copy(); bb.close()
    b = engine.Batch(ctx, G, NW, alpha=1.0)
    b.score()
    bad = 0
    for it in range(4000):
        a = alphas[(it * 3) % len(alphas)]
        b.set_alpha(a)
        l = b.score()
        if not np.array_equal(l, ref[a]):
            bad += 1
            if bad <= 4: print("  genes", sel, "iteration", it, "alpha", a, "got", l - ref[a])
    print("genes", sel, ": stale / wrong replays", bad, "of 4000", flush=True)
    b.close()
