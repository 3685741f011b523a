"""Experiment: does visiting the larger child subtree first (smaller write->read distance, better
Infinity-Cache reuse) change the full-traversal time?  Re-orders children in the Newick strings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import util
from pepr_amd import synth, engine

def reorder(nw, mode):
    t = util.parse_newick(nw)
    def size(nd): return 1 if not nd[0] else sum(size(k) for k in nd[0])
    def fmt(nd):
        kids, name, l = nd
        if not kids: return "%s:%.6f" % (name, l)
        ks = sorted(kids, key=size, reverse=(mode == "large_first")) if mode != "asis" else kids
        return "(" + ",".join(fmt(k) for k in ks) + "):%.6f" % l
    kids = t[0]
    ks = sorted(kids, key=size, reverse=(mode == "large_first")) if mode != "asis" else kids
    return "(" + ",".join(fmt(k) for k in ks) + ");"

ctx = engine.Context(0, profile=True)
genes = synth.simulate_genes(128, 50, 1000)
G = [(g[0], g[1]) for g in genes]
for mode in ("asis", "large_first", "small_first", "asis"):
    NW = [reorder(g[2], mode) for g in genes]
    b = engine.Batch(ctx, G, NW, alpha=0.8)
    for _ in range(3): l = b.score()
    ctx.kernel_stats(reset=True)
    t0 = time.time()
    for _ in range(20): l = b.score()
    dt = (time.time() - t0) / 20
    st = ctx.kernel_stats()
    print("%-12s %.3f ms/step  oplist %.3f ms  lnl0 %.4f" % (mode, dt * 1e3, st["newview"]["ms"] / 20, l[0]), flush=True)
    b.close()
