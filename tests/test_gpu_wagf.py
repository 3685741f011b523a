"""PROTGAMMAWAGF on the device (pml_model.pi_mode = PML_PI_EMPIRICAL): WAG exchangeabilities with frequencies counted from
each gene's alignment -- a per-gene eigen-system in k_pmat / k_newton / the eigen-basis fragments.  One of the 23 names
PEPR's -matrix_eval compares (PhylogenomicPipeline2.java:260-284, scored through RAxMLRunner.runRaxmlPerSiteLL :162-213).
Parity against oracle/ built with po.empirical_freqs of the same alignment (RAxML's own counting is parity-unpinned)."""
import os
import subprocess

import numpy as np
import pytest

from pepr_amd import engine, synth
from util import rf_collapsed

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle(po, names, rows, alpha=1.0):
    a = po.Alignment(names, rows)
    return a, po.Engine(a, po.Model(pi=po.empirical_freqs(a)), 4, alpha)


@pytest.mark.parametrize("ntax,nsites,seed,alpha,miss", [(7, 300, 1, 0.6, 0.0), (20, 900, 2, 1.3, 0.15), (50, 1000, 3, 0.8, 0.0)])
def test_wagf_score_and_site_lnl_vs_oracle(gpu_ctx, oracle_lib, ntax, nsites, seed, alpha, miss):
    po = oracle_lib
    names, rows, nw = synth.simulate_alignment(ntax, nsites, 8800 + seed, missing_frac=miss)
    a, e = _oracle(po, names, rows, alpha)
    ref, refs = e.site_lnl(po.Tree(nw, a))
    r = gpu_ctx.score([(names, rows)], [nw], alpha=alpha, pi_mode=engine.PI_EMPIRICAL, site_lnl=True)[0]
    assert abs(r["lnl"] - ref) < 1e-9 * abs(ref) and np.abs(r["site_lnl"] - refs).max() < 1e-9
    # it IS another likelihood function than PROTGAMMAWAG on the same data
    w = gpu_ctx.score([(names, rows)], [nw], alpha=alpha)[0]
    assert abs(w["lnl"] - r["lnl"]) > 1e-3


def test_wagf_batch_of_genes_each_with_its_own_frequencies(gpu_ctx, oracle_lib):
    """a batch mixes genes of different composition: each is scored, optimised and searched under ITS frequencies, and a gene's
    result does not depend on what shares its batch"""
    po = oracle_lib
    genes = [synth.simulate_alignment(8 + 3 * i, 250 + 120 * i, 8900 + i, 0.7 + 0.2 * i) for i in range(5)]
    G = [(g[0], g[1]) for g in genes]; NW = [g[2] for g in genes]
    sc = gpu_ctx.score(G, NW, alpha=0.9, pi_mode=engine.PI_EMPIRICAL)
    for g, r in zip(genes, sc):
        a, e = _oracle(po, g[0], g[1], 0.9)
        ref = e.lnl(po.Tree(g[2], a))
        assert abs(r["lnl"] - ref) < 1e-9 * abs(ref)
    opt = gpu_ctx.optimize(G, NW, pi_mode=engine.PI_EMPIRICAL)
    for i in (0, 3):
        a, e = _oracle(po, *G[i])
        t = po.Tree(NW[i], a)
        ref = e.optimize(t, True, 1e-4)
        assert abs(opt[i]["lnl"] - ref) < 1e-3 and abs(opt[i]["alpha"] - e.alpha) < 1e-3 * e.alpha
        lone = gpu_ctx.optimize([G[i]], [NW[i]], pi_mode=engine.PI_EMPIRICAL)[0]
        assert lone["lnl"] == opt[i]["lnl"] and lone["newick"] == opt[i]["newick"]


def test_wagf_counts_ambiguity_codes_like_the_oracle(gpu_ctx, oracle_lib):
    """an alignment full of B / Z / X / - / ? (they enter the frequency count proportionally) and with amino acids that never
    occur (floored at 0.001): the device model is built from the same frequencies as the oracle's"""
    po = oracle_lib
    rng = np.random.default_rng(3)
    names = ["t%d" % i for i in range(9)]
    letters = list("ARNDCQEGHILK") + list("BZX-?") * 2          # M F P S T W Y V never occur
    rows = ["".join(rng.choice(letters, 260)) for _ in names]
    _, _, nw = synth.simulate_alignment(9, 10, 4, names=names)
    a, e = _oracle(po, names, rows, 0.9)
    f = po.empirical_freqs(a)
    assert (f < 0.0011).sum() == 8 and abs(f.sum() - 1) < 1e-12
    ref, refs = e.site_lnl(po.Tree(nw, a))
    r = gpu_ctx.score([(names, rows)], [nw], alpha=0.9, pi_mode=engine.PI_EMPIRICAL, site_lnl=True)[0]
    assert abs(r["lnl"] - ref) < 1e-9 * abs(ref) and np.abs(r["site_lnl"] - refs).max() < 1e-9


def test_wagf_search_vs_oracle(gpu_ctx, oracle_lib):
    po = oracle_lib
    names, rows, nw = synth.simulate_alignment(14, 500, 8950, 0.8)
    a, e = _oracle(po, names, rows)
    ref, tree = e.search(None, 5, 1e-3)
    r = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5, pi_mode=engine.PI_EMPIRICAL)[0]
    assert abs(r["lnl"] - ref) < 1e-3 and rf_collapsed(r["newick"], tree.newick(12)) == 0


def test_wagf_resident_batch_and_refusals(gpu_ctx):
    genes = [synth.simulate_alignment(10, 300, 8960 + i) for i in range(3)]
    b = engine.Batch(gpu_ctx, [(g[0], g[1]) for g in genes], [g[2] for g in genes], alpha=0.8, pi_mode=engine.PI_EMPIRICAL)
    l1 = b.score(); l2 = b.score(stored=True)
    assert np.array_equal(l1, l2)
    one = gpu_ctx.score([(genes[1][0], genes[1][1])], [genes[1][2]], alpha=0.8, pi_mode=engine.PI_EMPIRICAL)[0]
    assert one["lnl"] == l1[1]
    b.close()
    with pytest.raises(engine.PmlError):            # device-gathered replicates have no per-replicate frequency count
        gpu_ctx.jackknife([(g[0], g[1]) for g in genes], reps=2, pi_mode=engine.PI_EMPIRICAL)


def test_raxml_shim_accepts_wagf_and_refuses_the_other_names(tmp_path):
    """`raxmlHPC -f e -m PROTGAMMAWAGF` (what -matrix_eval issues for that name) runs and reports another likelihood than
    PROTGAMMAWAG; names whose tables the reference does not hold are still refused with rc != 0"""
    RX = os.path.join(ROOT, "bin", "raxmlHPC")
    names, rows, nw = synth.simulate_alignment(8, 200, 8970)
    (tmp_path / "g.phy").write_text("%d %d\n" % (len(names), len(rows[0])) + "".join("%s %s\n" % (n, r) for n, r in zip(names, rows)))
    (tmp_path / "in.nwk").write_text(nw + "\n")
    out = {}
    for m in ("PROTGAMMAWAG", "PROTGAMMAWAGF"):
        p = subprocess.run([RX, "-f", "e", "-m", m, "-s", "g.phy", "-n", m, "-t", "in.nwk"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr
        info = (tmp_path / ("RAxML_info." + m)).read_text()
        out[m] = float([l for l in info.splitlines() if "Final GAMMA" in l][0].split()[-1])
    assert abs(out["PROTGAMMAWAG"] - out["PROTGAMMAWAGF"]) > 1e-3
    for m in ("PROTGAMMALGF", "PROTCATWAG", "PROTGAMMAJTT"):
        p = subprocess.run([RX, "-f", "e", "-m", m, "-s", "g.phy", "-n", "x" + m, "-t", "in.nwk"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert p.returncode != 0 and "PROTGAMMAWAG" in p.stderr
