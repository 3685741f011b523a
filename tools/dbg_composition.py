"""diagnostic: is a gene's search result independent of the batch it shares?  every subset (size >= 2) of the 8 genes of
test_concurrent_single_calls_are_coalesced as ONE batched call, each member compared bitwise with its lone result"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import engine, synth
ctx = engine.Context(0)
genes = [synth.simulate_alignment(9 + i % 4, 160 + 10 * i, 740 + i) for i in range(8)]
G = [(g[0], g[1]) for g in genes]
if os.environ.get("BIG_FIRST"):
    big = synth.simulate_genes(128, 50, 1000)
    if "search" in os.environ["BIG_FIRST"]:
        b = engine.Batch(ctx, [(g[0], g[1]) for g in big], None, alpha=1.0); b.search(True, True, 0, 1e-3); b.close()
    if "score" in os.environ["BIG_FIRST"]:
        n_, r_, t_ = synth.simulate_alignment(200, 5000, 11)
        ctx.score([(n_, r_)], [t_], alpha=0.8)
    print("big batches done", flush=True)
alone = [ctx.search([g], None, nni=True, spr_radius=0)[0] for g in G]
print("npatterns", [a["npatterns"] for a in alone])
bad = 0; n = 0
for k in range(2, 9):
    for sub in itertools.combinations(range(8), k):
        out = ctx.search([G[i] for i in sub], None, nni=True, spr_radius=0)
        n += 1
        for i, o in zip(sub, out):
            a = alone[i]
            if not (a["newick"] == o["newick"] and a["lnl"] == o["lnl"] and a["alpha"] == o["alpha"]):
                bad += 1
                print("subset %s: gene %d differs: dlnl %.3e dalpha %.3e" % (sub, i, o["lnl"] - a["lnl"], o["alpha"] - a["alpha"]), flush=True)
print("subsets %d, differing members %d" % (n, bad))
