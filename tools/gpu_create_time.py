"""Times Batch creation (encode + NJ start + arena) on C4-shaped genes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import synth, engine
ng, nt, ns = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = engine.Context(0)
t0 = time.time(); genes = synth.simulate_genes(ng, nt, ns); print("simulate %.2fs" % (time.time() - t0))
G = [(g[0], g[1]) for g in genes]
for rep in range(2):
    t0 = time.time(); b = engine.Batch(ctx, G, None, alpha=1.0); print("create %.3fs" % (time.time() - t0), flush=True); b.close()
