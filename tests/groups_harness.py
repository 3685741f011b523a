"""Prints (float.hex JSON) the results of one pml_search_batch call over 72 small genes under the PML_GROUPS of the environment;
tests/test_gpu_newton_fallback.py compares undivided (1), default and three groups bit for bit."""
import json
import sys

from pepr_amd import engine, synth


def main():
    ctx = engine.Context(0)
    genes = [synth.simulate_alignment(9 + i % 5, 150 + 40 * (i % 7), 6100 + i, missing_frac=0.1 * (i % 3 == 0)) for i in range(72)]
    G = [(g[0], g[1]) for g in genes]
    out = {}
    s = ctx.search(G, None, nni=True, spr_radius=0)
    out["nni"] = [[float(x["lnl"]).hex(), float(x["alpha"]).hex(), x["newick"]] for x in s]
    s = ctx.search(G, None, nni=True, spr_radius=5, seed=11)
    out["spr"] = [[float(x["lnl"]).hex(), float(x["alpha"]).hex(), x["newick"]] for x in s]
    out["stats"] = {k: v["launches"] for k, v in ctx.kernel_stats().items()}
    out["fallbacks"] = ctx.newton_fallbacks()
    ctx.close()
    json.dump(out, sys.stdout)


if __name__ == "__main__":
    main()
