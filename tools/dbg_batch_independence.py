"""A gene's search result must not depend on what else is in the batch, nor on its position: random subsets in random
order vs each gene alone, bitwise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pepr_amd import synth, engine
ctx = engine.Context(0)
genes = [synth.simulate_alignment(9 + i % 4, 160 + 10 * i, 740 + i) for i in range(8)]
G = [(g[0], g[1]) for g in genes]
alone = [ctx.search([g], None, nni=True, spr_radius=0)[0] for g in G]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    k = int(rng.integers(2, 9)); sel = [int(x) for x in rng.permutation(8)[:k]]
    out = ctx.search([G[i] for i in sel], None, nni=True, spr_radius=0)
    for pos, i in enumerate(sel):
        a, b = alone[i], out[pos]
        if a["newick"] != b["newick"] or a["lnl"] != b["lnl"] or a["alpha"] != b["alpha"]:
            bad += 1
            print("batch %s: gene %d at position %d DIFF lnl %.12f vs %.12f alpha %.12f vs %.12f rf %d" % (sel, i, pos, a["lnl"], b["lnl"], a["alpha"], b["alpha"], engine.rf_distance(a["newick"], b["newick"])), flush=True)
print("mismatches:", bad)
