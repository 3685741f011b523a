"""C4-shard search (NJ + NNI) of `ng` genes with the PML_GROUPS of the environment: seconds"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pepr_amd import synth, engine
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 32
genes = [synth.simulate_alignment(200, 5000, 1 + i, 0.8) for i in range(ng)]
G = [(g[0], g[1]) for g in genes]
ctx = engine.Context(0)
for rep in range(2):
    t0 = time.time(); out = ctx.search(G, None, nni=True, spr_radius=0, epsilon=1e-3); dt = time.time() - t0
    print("groups %s: C4-shape %d genes NNI search call %d: %.2f s = %.2f gene-trees/s, fallbacks %s" % (os.environ.get("PML_GROUPS", "default"), ng, rep, dt, ng / dt, ctx.newton_fallbacks()), flush=True)
