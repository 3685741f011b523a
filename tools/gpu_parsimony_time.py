"""Times device parsimony (pml_parsimony_batch) on C3-/C4-shaped genes; oracle on one gene for scale."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import synth, engine
from oracle import po
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
rad = int(sys.argv[4]) if len(sys.argv) > 4 else 20
ctx = engine.Context(0)
genes = synth.simulate_genes(ng, nt, ns)
G = [(g[0], g[1]) for g in genes]
ctx.parsimony(G[:1], seed=1, spr_radius=rad)                       # warm-up (module load)
for r in (0, rad):
    t0 = time.time(); out = ctx.parsimony(G, seed=1, spr_radius=r); dt = time.time() - t0
    rf = [engine.rf_distance(genes[i][2], out[i]["newick"]) for i in range(ng)]
    print("parsimony %d genes %dx%d radius %d: %.3f s -> %.1f trees/s; mean length %.0f; RF to true mean %.2f" %
          (ng, nt, ns, r, dt, ng / dt, sum(o["length"] for o in out) / ng, sum(rf) / ng), flush=True)
t0 = time.time(); t, L, mv = po.parsimony_tree(po.Alignment(*G[0]), 1, rad); dt = time.time() - t0
print("oracle 1 gene: %.3f s, length %d (device %d), %d SPR moves" % (dt, L, out[0]["length"], mv))
