"""Times the batched NNI search on C3-shaped genes and checks RF to the generating trees."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get('IMPORT_TORCH'):
    import torch; torch.cuda.init(); torch.cuda.synchronize(); print('torch initialised')
from pepr_amd import synth, engine
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
spr = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx = engine.Context(0, profile=not os.environ.get("NO_PROFILE"))
genes = synth.simulate_genes(ng, nt, ns)
G = [(g[0], g[1]) for g in genes]
t0 = time.time(); b = engine.Batch(ctx, G, None, alpha=1.0); print("create (NJ start) %.2fs" % (time.time() - t0), flush=True)
t0 = time.time(); lnl, al = b.search(True, True, spr, 1e-3); dt = time.time() - t0
rf = [engine.rf_distance(genes[g][2], b.newick(g)) for g in range(ng)]
print("spr radius", spr); print("search %d genes %dx%d: %.2f s -> %.2f gene-trees/s; RF to true tree: mean %.2f max %d; alpha mean %.3f" % (ng, nt, ns, dt, ng / dt, np.mean(rf), max(rf), al.mean()))
st = ctx.kernel_stats()
for k, v in st.items():
    if v["launches"]:
        print("  %-10s launches %6d  total %.1f ms" % (k, v["launches"], v["ms"]))
