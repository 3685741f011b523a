"""Repeats the concurrent single-call scenario and reports any result that differs from the lone call."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import synth, engine
ctx = engine.Context(0)
genes = [synth.simulate_alignment(9 + i % 4, 160 + 10 * i, 740 + i) for i in range(8)]
alone = [ctx.search([(g[0], g[1])], None, nni=True, spr_radius=0)[0] for g in genes]
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    out, errs = [None] * 9, [None] * 9
    def work(i):
        try:
            out[i] = ctx.search_one((genes[0][0], genes[0][1]), start="(nope:1,t1:1,t2:1);") if i == 8 else ctx.search_one((genes[i][0], genes[i][1]))
        except Exception as e:
            errs[i] = e
    before = ctx.coalescing_stats()
    th = [threading.Thread(target=work, args=(i,)) for i in range(9)]
    for t in th: t.start()
    for t in th: t.join()
    st = ctx.coalescing_stats()
    for i in range(8):
        a, b = alone[i], out[i]
        if errs[i] is not None or a["newick"] != b["newick"] or a["lnl"] != b["lnl"] or a["alpha"] != b["alpha"]:
            bad += 1
            print("rep %d gene %d DIFF err=%s lnl %.12f vs %.12f alpha %.12f vs %.12f rf %s batches %d" % (rep, i, errs[i], a["lnl"], b["lnl"] if b else 0, a["alpha"], b["alpha"] if b else 0,
                  engine.rf_distance(a["newick"], b["newick"]) if b else "-", st["batches"] - before["batches"]), flush=True)
    if errs[8] is None:
        print("rep %d: bad request did not fail" % rep)
print("mismatches:", bad)
