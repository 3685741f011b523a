import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import po
from pepr_amd import synth, engine
ctx = engine.Context(0)
for i, (nt, ns) in enumerate([(12, 200), (20, 150), (9, 120)]):
    names, rows, nw = synth.simulate_alignment(nt, ns, 13 + 2 * i)
    rng = np.random.default_rng(13 + 2 * i)
    start = synth.random_tree(nt, rng, [names[j] for j in rng.permutation(nt)])[0]
    a = po.Alignment(names, rows)
    for rad in (0, 5):
        e = po.Engine(a, po.Model(0), 4, 1.0)
        lnl, tree = e.search(po.Tree(start, a), rad, 1e-3)
        r = ctx.search([(names, rows)], [start], alpha=1.0, nni=True, spr_radius=rad, epsilon=1e-3)[0]
        got = po.Tree(r["newick"], a)
        e2 = po.Engine(a, po.Model(0), 4, r["alpha"])
        print(nt, ns, "rad", rad, "oracle %.6f gpu %.6f  rf(gpu,oracle) %d  oracle-rescore-of-gpu-tree %.6f alpha %.4f/%.4f" % (lnl, r["lnl"], got.rf(tree), e2.lnl(got), e.alpha, r["alpha"]), flush=True)
