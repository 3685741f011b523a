"""A gene's search result must not depend on what else is in the batch: batch of 8 vs each alone, bitwise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import synth, engine
ctx = engine.Context(0)
genes = [synth.simulate_alignment(9 + i % 4, 160 + 10 * i, 740 + i) for i in range(8)]
G = [(g[0], g[1]) for g in genes]
for mode in ("search", "optimize"):
    if mode == "search":
        whole = ctx.search(G, None, nni=True, spr_radius=0)
        alone = [ctx.search([g], None, nni=True, spr_radius=0)[0] for g in G]
    else:
        NW = [g[2] for g in genes]
        whole = ctx.optimize(G, NW)
        alone = [ctx.optimize([g], [nw])[0] for g, nw in zip(G, NW)]
    for i, (a, b) in enumerate(zip(whole, alone)):
        same = a["newick"] == b["newick"] and a["lnl"] == b["lnl"] and a["alpha"] == b["alpha"]
        print(mode, i, "same" if same else "DIFF lnl %.12f vs %.12f alpha %.10f vs %.10f rf %d" % (a["lnl"], b["lnl"], a["alpha"], b["alpha"], engine.rf_distance(a["newick"], b["newick"])))
