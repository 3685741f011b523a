"""C5-shaped rehearsal: 500 taxa x 2000 sites, 20 % of the taxa absent per gene, NNI + SPR search and parsimony."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pepr_amd import synth, engine
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = engine.Context(0)
genes = [synth.simulate_alignment(500, 2000, 9000 + i, missing_frac=0.2) for i in range(ng)]
G = [(g[0], g[1]) for g in genes]
t0 = time.time(); out = ctx.search(G, None, nni=True, spr_radius=5, epsilon=1e-3); dt = time.time() - t0
rf = [engine.rf_distance(genes[i][2], out[i]["newick"]) for i in range(ng)]
print("C5-shaped search (NNI + SPR r=5) %d genes 500x2000: %.1f s -> %.2f gene-trees/s; RF to generating tree mean %.1f max %d; finite %s" %
      (ng, dt, ng / dt, np.mean(rf), max(rf), all(np.isfinite(o["lnl"]) for o in out)), flush=True)
t0 = time.time(); p = ctx.parsimony(G, seed=1, spr_radius=20); dt = time.time() - t0
rfp = [engine.rf_distance(genes[i][2], p[i]["newick"]) for i in range(ng)]
print("parsimony (radius 20) %d genes 500x2000: %.1f s; RF to generating tree mean %.1f" % (ng, dt, np.mean(rfp)))
