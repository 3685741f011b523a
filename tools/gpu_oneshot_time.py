"""PCIe-inclusive rate: one-shot pml_score_batch (host char rows in, lnL out) on C3-shaped genes,
cold (fresh arena) and warm (context keeps the arena)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import synth, engine
ng, nt, ns = 128, 50, 1000
genes = synth.simulate_genes(ng, nt, ns)
G = [(g[0], g[1]) for g in genes]; NW = [g[2] for g in genes]
ctx = engine.Context(0)
for i in range(4):
    t0 = time.time(); out = ctx.score(G, NW, alpha=0.8); dt = time.time() - t0
    npat = sum(o["npatterns"] for o in out)
    print("one-shot score call %d: %.1f ms -> %.2f M site-lnL/s (%d patterns; host rows -> encode -> H2D -> traversal -> lnL)" % (i, dt * 1e3, npat / dt / 1e6, npat), flush=True)
