"""The reference's data-parallel loop as one call: full tree + gene-subset support trees +
support counts (PhylogenomicPipeline2.java:994-1126)."""
import numpy as np
import pytest

from pepr_amd import engine, synth
from util import rf_collapsed

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


# ---- SURVEY 8f-3 against the oracle side: the bytes k_gather produces, and replicates scored by oracle/ ----

def _sliced_genes(ntax, ngenes, nsites, seed, drop=True, ambiguous=True):
    """one long alignment simulated on ONE tree, cut into genes; some genes lack a taxon, some residues are B/Z/X/-"""
    names, rows, nw = synth.simulate_alignment(ntax, ngenes * nsites, seed, alpha=0.9)
    rng = np.random.default_rng(seed)
    genes = []
    for g in range(ngenes):
        nm, rw = list(names), [r[g * nsites:(g + 1) * nsites] for r in rows]
        if ambiguous:
            rw = [list(r) for r in rw]
            for _ in range(nsites // 10):
                rw[rng.integers(ntax)][rng.integers(nsites)] = "BZX-"[rng.integers(4)]
            rw = ["".join(r) for r in rw]
        if drop and g % 3 == 1:
            k = 1 + g % (ntax - 1)
            nm, rw = nm[:k] + nm[k + 1:], rw[:k] + rw[k + 1:]
        genes.append((nm, rw))
    return names, nw, genes


def _column_multiset(codes, weights):
    out = {}
    for p in range(codes.shape[1]):
        if weights[p] != 0:
            k = codes[:, p].tobytes()
            out[k] = out.get(k, 0) + int(weights[p])
    return out


def _encode_text(rows):
    """independent restatement of the code table of peprml.h (pml_debug_gather) on raw text columns"""
    table = np.full(256, 22, dtype=np.uint8)
    for i, ch in enumerate(synth.AA):
        table[ord(ch)] = i; table[ord(ch.lower())] = i
    table[ord("B")] = table[ord("b")] = 20
    table[ord("Z")] = table[ord("z")] = 21
    arr = np.frombuffer("".join(rows).encode(), dtype=np.uint8).reshape(len(rows), -1)
    return table[arr]


def test_gather_bytes_equal_concatenated_text(gpu_ctx):
    """k_gather's replicate matrix (read back through pml_debug_gather) holds exactly the columns of the
    MSAConcatenator text (MSAConcatenator.java:78-189: sorted taxon union, '?' rows for absent genes): same taxon
    order, same multiset of (column, weight) -- integer/byte work, bit-exact."""
    from oracle import po
    names, nw, genes = _sliced_genes(9, 7, 64, 31)
    draws = engine.jackknife_draw(len(genes), 5, 0, 99)
    for sel in draws[:3] + [None, [1], [4, 1]]:
        nm_dev, codes, w = gpu_ctx.debug_gather(genes, sel)
        nm_txt, rows = engine.concatenate(genes, sel)
        assert nm_dev == nm_txt == sorted(nm_txt)
        txt = _encode_text(rows)
        assert txt.shape[0] == codes.shape[0] and w.sum() == txt.shape[1]
        assert _column_multiset(codes, w) == _column_multiset(txt, np.ones(txt.shape[1]))
        # the oracle's own encoder compresses the same text to the same distinct columns
        a = po.Alignment(nm_txt, rows)
        assert a.npat == len(_column_multiset(codes, w)) and int(a.weight.sum()) == txt.shape[1]
        # per-gene compression only: within one gene's segment no column repeats
        off = 0
        for g in (sel if sel is not None else range(len(genes))):
            npat_g = len(_column_multiset(_encode_text(genes[g][1]), np.ones(len(genes[g][1][0]))))
            seg = codes[:, off:off + npat_g]
            assert len({seg[:, p].tobytes() for p in range(npat_g)}) == npat_g
            off += npat_g
        assert off == codes.shape[1]


def test_jackknife_vs_oracle_on_concatenated_text(gpu_ctx):
    """pml_jackknife's full tree and one replicate against oracle/ run on the concatenated TEXT of the same genes
    (PhylogenomicPipeline2.java:959-977 builds that text per replicate): RF 0, |dlnL| < 1e-3."""
    from oracle import po
    names, nw, genes = _sliced_genes(8, 6, 110, 17)
    reps, seed = 3, 5
    r = gpu_ctx.jackknife(genes, reps=reps, subset_size=0, seed=seed, spr_radius_full=5)
    draws = engine.jackknife_draw(len(genes), reps, 0, seed)
    m = po.Model(0)
    # full tree: NJ + NNI + SPR radius 5 on the text of all genes
    nm, rows = engine.concatenate(genes)
    a = po.Alignment(nm, rows); e = po.Engine(a, m, 4, 1.0)
    lnl_o, t_o = e.search(None, 5, 1e-3)
    assert abs(r["lnl"] - lnl_o) < 1e-3 and abs(r["alpha"] - e.alpha) < 1e-3 * e.alpha
    assert rf_collapsed(r["newick"], t_o.newick()) == 0
    # replicate 1 (its genes come from pml_jackknife_draw): NJ + NNI on its text
    nm, rows = engine.concatenate(genes, draws[1])
    a = po.Alignment(nm, rows); e = po.Engine(a, m, 4, 1.0)
    lnl_o, t_o = e.search(None, 0, 1e-3)
    sup = r["support_trees"][1]
    assert rf_collapsed(sup, t_o.newick()) == 0
    # the support tree as returned (6-digit lengths) scores within 1e-3 of the oracle's optimum under the oracle's alpha
    assert abs(e.lnl(po.Tree(sup, a)) - lnl_o) < 1e-3
