"""root-branch (lnL, d1, d2) of a few genes in the Newton form selected by the environment, next to the oracle"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import engine, synth
from oracle import po
shapes = [(6, 90), (10, 400), (14, 1500)]
genes = [synth.simulate_alignment(nt, ns, 500 + i, missing_frac=0.1 * (i % 2)) for i, (nt, ns) in enumerate(shapes)]
ctx = engine.Context(0)
b = engine.Batch(ctx, [(g[0], g[1]) for g in genes], [g[2] for g in genes], alpha=0.8)
l, d1, d2 = b.root_derivs()
for i, g in enumerate(genes):
    a = po.Alignment(g[0], g[1]); t = po.Tree(g[2], a); e = po.Engine(a, po.Model(0), 4, 0.8)
    print(os.environ.get("MODE"), i, "gpu", l[i], d1[i], d2[i], "oracle lnl", e.lnl(t), "fallbacks", ctx.newton_fallbacks())
