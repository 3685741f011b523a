"""Prints (as float.hex JSON) what a fixed set of requests returns; tests/test_gpu_chaining.py runs it once per PML_CHAIN mode
(the switch is read once per process) and compares the outputs bit for bit."""
import json
import sys

import numpy as np

from pepr_amd import engine, synth


def main():
    ctx = engine.Context(0)
    out = {}
    genes = [synth.simulate_alignment(nt, ns, 900 + i, missing_frac=0.1 * (i % 2)) for i, (nt, ns) in enumerate([(5, 40), (9, 333), (16, 700), (33, 129), (50, 1000), (64, 2100)])]
    A = [(g[0], g[1]) for g in genes]; T = [g[2] for g in genes]
    # whole-tree scoring passes (recorded plan, then a replay with another alpha) and per-site likelihoods (plain evaluation)
    b = engine.Batch(ctx, A, T, alpha=0.8)
    out["score"] = [float(x).hex() for x in b.score()]
    out["score_again"] = [float(x).hex() for x in b.score()]
    out["score_stored"] = [float(x).hex() for x in b.score(stored=True)]        # every CLV written (bench.py's second leg)
    out["score_stored_replay"] = [float(x).hex() for x in b.score(stored=True)]
    lnl, al = b.optimize()             # after a scoring pass that kept its chained results in registers only
    out["batch_optimize"] = [float(x).hex() for x in lnl] + [float(x).hex() for x in al]
    b.close()
    r = ctx.score(A, T, alpha=0.6, site_lnl=True)
    out["site"] = [[float(x).hex() for x in g["site_lnl"][:50]] for g in r]
    # deep caterpillar: the 2^256 rescue inside chains of tip + previous-result operations
    n, L = 300, 64
    rng = np.random.default_rng(5)
    names = ["s%d" % i for i in range(n)]
    rows = ["".join(rng.choice(list(synth.AA), L)) for _ in range(n)]
    nw = names[0]
    for i in range(1, n):
        nw = "(%s:0.9,%s:1.3)" % (nw, names[i])
    r = ctx.score([(names, rows)], [nw + ";"], alpha=0.9, site_lnl=True)[0]
    out["caterpillar"] = [float(r["lnl"]).hex()] + [float(x).hex() for x in r["site_lnl"]]
    # branch-length + alpha optimisation and complete searches (NNI, then lazy SPR from a parsimony start)
    o = ctx.optimize(A[:4], T[:4])
    out["optimize"] = [[float(x["lnl"]).hex(), float(x["alpha"]).hex(), x["newick"]] for x in o]
    s = ctx.search(A[:5], None, nni=True, spr_radius=0)
    out["search"] = [[float(x["lnl"]).hex(), float(x["alpha"]).hex(), x["newick"]] for x in s]
    s = ctx.search(A[1:4], None, nni=True, spr_radius=5, seed=3)
    out["search_spr"] = [[float(x["lnl"]).hex(), float(x["alpha"]).hex(), x["newick"]] for x in s]
    ctx.close()
    json.dump(out, sys.stdout)


if __name__ == "__main__":
    main()
