"""Prints (as float.hex JSON) what a fixed set of branch-length / search requests returns plus the k_newton fallback counters;
tests/test_gpu_newton_fallback.py runs it with and without PML_NEWTON_TIMEOUT_US=0 (every split exchange gives up at its first
unsuccessful poll, so the work goes through the no-exchange form) and compares the outputs bit for bit."""
import json
import sys

from pepr_amd import engine, synth


def main():
    ctx = engine.Context(0)
    out = {}
    shapes = [(6, 90), (10, 400), (14, 1500), (24, 700), (20, 12000)]       # 1, 3, 9, 6 slices; the last one streams (> 8192 patterns)
    genes = [synth.simulate_alignment(nt, ns, 500 + i, missing_frac=0.1 * (i % 2)) for i, (nt, ns) in enumerate(shapes)]
    A = [(g[0], g[1]) for g in genes]; T = [g[2] for g in genes]
    b = engine.Batch(ctx, A, T, alpha=0.8)
    out["npat"] = b.npatterns()
    out["root_derivs"] = [[float(x).hex() for x in v] for v in b.root_derivs()]
    lnl, al = b.optimize()
    out["batch_optimize"] = [float(x).hex() for x in lnl] + [float(x).hex() for x in al]
    out["trees"] = [b.newick(i, 17) for i in range(len(A))]
    b.close()
    s = ctx.search(A[:4], None, nni=True, spr_radius=0)
    out["search"] = [[float(x["lnl"]).hex(), float(x["alpha"]).hex(), x["newick"]] for x in s]
    s = ctx.search(A[1:4], None, nni=True, spr_radius=5, seed=3)
    out["search_spr"] = [[float(x["lnl"]).hex(), float(x["alpha"]).hex(), x["newick"]] for x in s]
    s = ctx.sh_support([A[1]], [T[1]], alpha=0.7, nboot=200, seed=5)
    out["sh"] = [x["newick"] for x in s]
    out["fallbacks"] = ctx.newton_fallbacks()
    ctx.close()
    json.dump(out, sys.stdout)


if __name__ == "__main__":
    main()
