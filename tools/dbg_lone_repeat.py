"""diagnostic: distinct outcomes of the SAME lone search repeated in one process"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import engine, synth
ctx = engine.Context(0)
genes = [synth.simulate_alignment(9 + i % 4, 160 + 10 * i, 740 + i) for i in range(8)]
for gi in (2, 5):
    g = genes[gi]
    c = collections.Counter()
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
        r = ctx.search([(g[0], g[1])], None, nni=True, spr_radius=0)[0]
        c[(r["alpha"], r["lnl"])] += 1
    print("gene", gi, "distinct outcomes:", dict(c), flush=True)
