"""First GPU check: GPU vs oracle on small cases, then a C3-shaped timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import po
from pepr_amd import synth, engine

ctx = engine.Context(0, profile=True)
m = po.Model(po.PI_RAXML3DP)
for (nt, ns, seed, miss) in [(4, 60, 3, 0.0), (8, 300, 5, 0.0), (12, 2000, 7, 0.3), (50, 1000, 1, 0.0)]:
    names, rows, nw = synth.simulate_alignment(nt, ns, seed, missing_frac=miss)
    a = po.Alignment(names, rows); t = po.Tree(nw, a); e = po.Engine(a, m, 4, 0.7)
    ref, refsites = e.site_lnl(t)
    r = ctx.score([(names, rows)], [nw], alpha=0.7, site_lnl=True)[0]
    print("score %dx%d npat %d/%d oracle %.9f gpu %.9f diff %.3e site maxdiff %.3e" % (
        nt, ns, r["npatterns"], a.npat, ref, r["lnl"], r["lnl"] - ref, np.abs(r["site_lnl"] - refsites).max()), flush=True)
    b = engine.Batch(ctx, [(names, rows)], [nw], alpha=0.7)
    l, d1, d2 = b.root_derivs()
    ol, o1, o2 = e.branch_derivs(t, 0, -1) if False else (None, None, None)
    print("   gpu root derivs", l[0], d1[0], d2[0], flush=True)
    b.close()

# optimise parity (small)
names, rows, nw = synth.simulate_alignment(12, 500, 11)
a = po.Alignment(names, rows); t = po.Tree(nw, a); e = po.Engine(a, m, 4, 1.0)
t0 = time.time(); ol = e.optimize(t, True, 1e-4); to = time.time() - t0
t0 = time.time(); r = ctx.optimize([(names, rows)], [nw], alpha=1.0, epsilon=1e-4)[0]; tg = time.time() - t0
print("optimize 12x500: oracle %.6f a=%.6f (%.1fs)  gpu %.6f a=%.6f (%.1fs) diff %.3e" % (ol, e.alpha, to, r["lnl"], r["alpha"], tg, r["lnl"] - ol), flush=True)

# C3-shaped timing: 128 genes x 50 taxa x 1000 sites
t0 = time.time(); genes = synth.simulate_genes(128, 50, 1000); print("simulated", time.time() - t0, flush=True)
G = [(g[0], g[1]) for g in genes]; NW = [g[2] for g in genes]
t0 = time.time(); b = engine.Batch(ctx, G, NW, alpha=0.8); print("batch create", time.time() - t0, flush=True)
npat = b.npatterns(); print("patterns total", sum(npat))
l = b.score(); ctx.kernel_stats(reset=True)
t0 = time.time(); K = 10
for _ in range(K): l = b.score()
dt = (time.time() - t0) / K
print("C3 score: %.3f ms/step  %.2f M site-lnL/s" % (dt * 1e3, sum(npat) / dt / 1e6))
st = ctx.kernel_stats()
for k, v in st.items():
    if v["launches"]:
        print("  %-9s launches %5d  %.3f ms/step  %.1f GB/s algo" % (k, v["launches"], v["ms"] / K, v["algo_bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else 0))
a0 = po.Alignment(*G[0]); t00 = po.Tree(NW[0], a0); e0 = po.Engine(a0, m, 4, 0.8)
print("gene0 oracle %.6f gpu %.6f" % (e0.lnl(t00), l[0]))
