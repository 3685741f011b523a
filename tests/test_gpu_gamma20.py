"""FastTree's `-gamma` step on the device (pml_gamma20: five traversals of four fixed rates + k_g20) against the CPU oracle
(po_gamma20): Gamma20 lnL, alpha, rescale and the rescaled tree.  Reference call site: FastTreeRunner.java:67-70."""
import re

import numpy as np
import pytest

from pepr_amd import engine, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ntax,nsites,seed,alpha,miss", [(8, 300, 5, 0.7, 0.0), (14, 250, 71, 2.5, 0.1), (50, 1000, 1, 0.8, 0.0), (30, 64, 9, 0.3, 0.4)])
def test_gamma20_vs_oracle(gpu_ctx, oracle_lib, ntax, nsites, seed, alpha, miss):
    po = oracle_lib
    names, rows, nw = synth.simulate_alignment(ntax, nsites, seed, alpha, missing_frac=miss)
    a = po.Alignment(names, rows); t = po.Tree(nw, a)
    ref_lnl, ref_alpha, ref_rescale = po.gamma20(a, po.Model(1), t)
    g = gpu_ctx.gamma20([(names, rows)], [nw])[0]
    assert g["npatterns"] == a.npat
    # the table agrees to ~1e-12, so both optimisers walk the same path; the tolerances are those of the fit (1e-3 in
    # log alpha / log mult) and of the likelihood it reports
    assert abs(g["lnl"] - ref_lnl) < 1e-6 * abs(ref_lnl), (g["lnl"], ref_lnl)
    assert abs(g["alpha"] - ref_alpha) < 2e-3 * ref_alpha and abs(g["rescale"] - ref_rescale) < 2e-3 * ref_rescale
    assert abs(g["tree_length"] - t.length() * g["rescale"]) < 1e-9 * max(1.0, t.length())
    assert engine.rf_distance(g["newick"], nw) == 0


def test_gamma20_batch_is_composition_independent(gpu_ctx):
    genes = [synth.simulate_alignment(10 + 3 * i, 150 + 40 * i, 300 + i, 0.5 + 0.3 * i) for i in range(5)]
    G = [(g[0], g[1]) for g in genes]; NW = [g[2] for g in genes]
    whole = gpu_ctx.gamma20(G, NW)
    for i in (0, 3):
        one = gpu_ctx.gamma20([G[i]], [NW[i]])[0]
        assert one["lnl"] == whole[i]["lnl"] and one["alpha"] == whole[i]["alpha"] and one["rescale"] == whole[i]["rescale"]


def test_gamma20_batch_sub_batches_under_an_hbm_budget(gpu_ctx, monkeypatch):
    """a gene list that does not fit in HBM at once is fitted in consecutive sub-batches (as pml_search_batch does): same bits"""
    genes = [synth.simulate_alignment(12, 200, 400 + i, 0.8) for i in range(6)]
    G = [(g[0], g[1]) for g in genes]; NW = [g[2] for g in genes]
    whole = gpu_ctx.gamma20(G, NW)
    monkeypatch.setenv("PML_HBM_BUDGET_MB", "5")          # one 12 x 200 gene needs ~3.6 MB: one or two genes per sub-batch
    parts = gpu_ctx.gamma20(G, NW)
    assert [(p["lnl"], p["alpha"], p["rescale"], p["newick"]) for p in parts] == [(w["lnl"], w["alpha"], w["rescale"], w["newick"]) for w in whole]


def test_gamma20_recovers_a_known_rescale(gpu_ctx):
    """the same data scored on a tree whose lengths were all shrunk by 1.25: the fitted rescale grows by that factor and the
    Gamma20 likelihood barely moves (the continuous model is invariant under lengths / c, mean rate x c; the FIXED 20-rate grid
    samples the shifted distribution at other quantiles, hence a fraction of a log unit)"""
    names, rows, nw = synth.simulate_alignment(16, 600, 17, 0.9)
    small = re.sub(r":([0-9.eE+-]+)", lambda m: ":%.10f" % (float(m.group(1)) / 1.25), nw)
    a = gpu_ctx.gamma20([(names, rows)], [nw])[0]
    b = gpu_ctx.gamma20([(names, rows)], [small])[0]
    assert abs(b["rescale"] / a["rescale"] - 1.25) < 0.01
    assert abs(a["lnl"] - b["lnl"]) < 0.5
