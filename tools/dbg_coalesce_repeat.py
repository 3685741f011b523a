"""diagnostic: repeat the coalesced-vs-lone comparison and print every bitwise difference (which gene, which field, by how much)"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import engine, synth
ctx = engine.Context(0)
genes = [synth.simulate_alignment(9 + i % 4, 160 + 10 * i, 740 + i) for i in range(8)]
alone = [ctx.search([(g[0], g[1])], None, nni=True, spr_radius=0)[0] for g in genes]
alone2 = [ctx.search([(g[0], g[1])], None, nni=True, spr_radius=0)[0] for g in genes]
print("lone vs lone identical:", all(a["newick"] == b["newick"] and a["lnl"] == b["lnl"] and a["alpha"] == b["alpha"] for a, b in zip(alone, alone2)))
nbad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    out = [None] * 9
    def work(i):
        try:
            if i == 8 and os.environ.get("WITH_BAD"):
                out[i] = ctx.search_one((genes[0][0], genes[0][1]), start="(nope:1,t1:1,t2:1);")
            elif i < 8:
                out[i] = ctx.search_one((genes[i][0], genes[i][1]))
        except Exception:
            pass
    th = [threading.Thread(target=work, args=(i,)) for i in range(9)]
    for t in th: t.start()
    for t in th: t.join()
    for i, (a, b) in enumerate(zip(alone, out[:8])):
        if not (a["newick"] == b["newick"] and a["lnl"] == b["lnl"] and a["alpha"] == b["alpha"]):
            nbad += 1
            print("rep %d gene %d: dlnl %.3e dalpha %.3e newick_equal %s" % (rep, i, b["lnl"] - a["lnl"], b["alpha"] - a["alpha"], a["newick"] == b["newick"]))
            if a["newick"] != b["newick"]:
                import re
                la = [float(x) for x in re.findall(r":([0-9.eE+-]+)", a["newick"])]; lb = [float(x) for x in re.findall(r":([0-9.eE+-]+)", b["newick"])]
                print("   max |dlen| %.3e  rf %d" % (max(abs(x - y) for x, y in zip(la, lb)) if len(la) == len(lb) else -1, engine.rf_distance(a["newick"], b["newick"])))
print("differences:", nbad, "stats", ctx.coalescing_stats())
