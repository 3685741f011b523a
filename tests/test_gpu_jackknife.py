"""The reference's data-parallel loop as one call: full tree + gene-subset support trees +
support counts (PhylogenomicPipeline2.java:994-1126)."""
import numpy as np
import pytest

from pepr_amd import engine, synth

pytestmark = pytest.mark.gpu


def _genes(ntax, ngenes, nsites, seed, drop=0):
    """genes simulated on ONE species tree (same names), optionally missing a taxon in some genes"""
    rng = np.random.default_rng(seed)
    names = ["sp%02d" % i for i in range(ntax)]
    newick, kids, blen, root = synth.random_tree(ntax, rng, names)
    out = []
    for g in range(ngenes):
        # re-simulate sequences on the same tree by reseeding the generator part that draws states
        n2, rows, _ = synth.simulate_alignment(ntax, nsites, seed, alpha=0.9, names=names)
        # different columns per gene: permute/perturb by drawing an independent alignment on the same tree
        rg = np.random.default_rng(1000 * seed + g)
        cols = rg.permutation(nsites)
        rows = ["".join(r[c] for c in cols) for r in rows]
        nm = list(names)
        if drop and g % 3 == 1:
            k = 1 + g % (ntax - 1)
            nm = nm[:k] + nm[k + 1:]; rows = rows[:k] + rows[k + 1:]
        out.append((nm, rows))
    return names, newick, out


def test_jackknife_supports(gpu_ctx):
    names, true_nw, genes = _genes(10, 8, 120, 5, drop=1)
    r = gpu_ctx.jackknife(genes, reps=12, subset_size=0, seed=7, spr_radius_full=5)
    assert len(r["support_trees"]) == 12 and r["nsites"] == 8 * 120
    # the decorated tree parses in the reference's dialect (supports = inner labels) and keeps the topology
    main_plain = gpu_ctx.search([engine.concatenate(genes)], None, nni=True, spr_radius=5)[0]
    assert engine.rf_distance(r["newick"], main_plain["newick"]) == 0 and abs(r["lnl"] - main_plain["lnl"]) < 1e-6
    import re
    sup = [int(x) for x in re.findall(r"\)(\d+):", r["newick"])]
    assert len(sup) == 10 - 3 and all(0 <= s <= 12 for s in sup)
    # deterministic under the same seed, different subsets under another
    r2 = gpu_ctx.jackknife(genes, reps=12, subset_size=0, seed=7, spr_radius_full=5)
    assert r2["newick"] == r["newick"] and r2["support_trees"] == r["support_trees"]


def test_jackknife_all_genes_gives_full_support(gpu_ctx):
    """subset = all genes: every replicate is the full concatenation, so every branch that NNI and
    NNI+SPR agree on gets support = reps (README:19-20 '100% support for all branches')."""
    names, true_nw, genes = _genes(8, 5, 150, 9)
    r = gpu_ctx.jackknife(genes, reps=6, subset_size=5, seed=1, spr_radius_full=0)
    import re
    sup = [int(x) for x in re.findall(r"\)(\d+):", r["newick"])]
    assert sup == [6] * (8 - 3)
    assert len(set(r["support_trees"])) == 1


def test_jackknife_sharded_equals_unsharded(gpu_ctx):
    """pml_jackknife_opts.shard_*: the multi-GPU split of the replicates (one rank per GPU) reproduces the
    single-GPU call: same replicate trees in rank-interleaved order, same supports after decoration."""
    import re
    names, true_nw, genes = _genes(9, 8, 100, 11, drop=1)
    whole = gpu_ctx.jackknife(genes, reps=7, seed=3, spr_radius_full=5)
    parts = [gpu_ctx.jackknife(genes, reps=7, seed=3, spr_radius_full=5, shard=(r, 3)) for r in range(3)]
    assert parts[1]["newick"] is None and parts[2]["newick"] is None
    assert [len(p["support_trees"]) for p in parts] == [3, 2, 2]
    merged = [None] * 7
    for r, p in enumerate(parts):
        for i, t in enumerate(p["support_trees"]):
            merged[r + 3 * i] = t
    assert merged == whole["support_trees"]
    plain = re.sub(r"\)\d+:", "):", parts[0]["newick"])
    decorated = engine.support_tree(plain, merged, digits=6)
    assert sorted(re.findall(r"\)(\d+):", decorated)) == sorted(re.findall(r"\)(\d+):", whole["newick"]))
    assert engine.rf_distance(decorated, whole["newick"]) == 0


def test_jackknife_replicate_sub_batching(gpu_ctx, monkeypatch):
    """replicates whose arenas do not fit in HBM together run as consecutive sub-batches: same result"""
    names, true_nw, genes = _genes(9, 6, 90, 21)
    whole = gpu_ctx.jackknife(genes, reps=8, seed=5, spr_radius_full=0)
    monkeypatch.setenv("PML_HBM_BUDGET_MB", "30")
    parts = gpu_ctx.jackknife(genes, reps=8, seed=5, spr_radius_full=0)
    assert parts["support_trees"] == whole["support_trees"] and parts["newick"] == whole["newick"]
