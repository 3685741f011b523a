"""BASELINE configs C1/C2 as far as they can be run here: the bundled Erysipelotrichales and
Aquificales proteomes reduced to stand-in gene alignments (tests/golden/standin_*.json, built by
tools/make_standin_alignments.py because blastall/muscle/Gblocks/Java are unavailable), pushed
through the engine's jackknife call = PEPR's tree-building step (full tree + support trees +
support counts).  The reference binaries cannot be run, so what is asserted is what the reference
states qualitatively (README:19-20, 32-33: well-supported trees) plus taxonomy every tree of
these genomes must show."""
import json
import os
import re

import pytest

from pepr_amd import engine

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    d = json.load(open(os.path.join(GOLD, "standin_%s.json" % name)))
    return d, [(g["names"], g["rows"]) for g in d["genes"]]


def _clades(newick):
    """dict frozenset(taxa) -> support label for every labelled inner node (either side of the split)"""
    import util
    tree = util.parse_newick(re.sub(r"\)(\d+):", r")\1:", newick))
    out = {}
    def rec(nd):
        if not nd[0]:
            return frozenset([nd[1]])
        s = frozenset().union(*[rec(k) for k in nd[0]])
        if nd[1].isdigit():
            out[s] = int(nd[1])
        return s
    allt = rec(tree)
    return out, allt


def _support(newick, members):
    cl, allt = _clades(newick)
    m = frozenset(members)
    return cl.get(m, cl.get(allt - m, None))


def test_aquificales_standin(gpu_ctx):
    d, genes = _load("Aquificales")
    reps = 20
    r = gpu_ctx.jackknife(genes, reps=reps, seed=11, spr_radius_full=5)
    assert r["nsites"] == sum(len(g[1][0]) for g in genes) and len(r["support_trees"]) == reps
    taxa = d["taxa"]
    hbac = [t for t in taxa if t.startswith("Hydrogenobaculum")]
    hthe = [t for t in taxa if t.startswith("Hydrogenobacter_thermophilus")]
    sulf = [t for t in taxa if t.startswith("Sulfurihydrogenibium")]
    hydrogenothermaceae = sulf + [t for t in taxa if t.startswith("Persephonella")]
    assert len(hbac) == 4 and len(hthe) == 2 and len(sulf) == 2
    for clade in (hbac, hthe, sulf, hydrogenothermaceae):
        assert _support(r["newick"], clade) == reps, (clade, r["newick"])
    sup = [int(x) for x in re.findall(r"\)(\d+):", r["newick"])]
    assert len(sup) == len(taxa) - 3 and sum(s == reps for s in sup) >= len(sup) - 3


def test_erysipelotrichales_standin_missing_genes(gpu_ctx):
    """two genomes carry paralogs of every selected product and drop out of all families (the
    union-of-taxa rule of MSAConcatenator decides the taxon set, not the genome list)"""
    d, genes = _load("Erysipelotrichales")
    assert all(len(g[0]) < len(d["taxa"]) for g in genes)
    reps = 10
    r = gpu_ctx.jackknife(genes, reps=reps, seed=3, spr_radius_full=5)
    taxa = sorted(set().union(*[set(g[0]) for g in genes]))       # genomes present in >= 1 family
    ery = [t for t in taxa if t.startswith("Erysipelothrix")]
    ram = ["Clostridium_ramosum_DSM_1402", "Clostridium_spiroforme_DSM_1552"]
    outg = [t for t in d["outgroup"] if t in taxa]
    assert len(ery) == 2 and len(outg) == 4
    for clade in (ery, ram, outg):
        sup = _support(r["newick"], clade)
        assert sup is not None and sup >= reps - 1, (clade, r["newick"])
    # same concatenation scored through the plain ABI agrees (gaps/? = all-ones tips)
    names, rows = engine.concatenate(genes)
    assert sorted(names) == taxa and len(rows[0]) == r["nsites"]
    plain = re.sub(r"\)\d+:", "):", r["newick"])
    again = gpu_ctx.score([(names, rows)], [plain], alpha=r["alpha"])[0]["lnl"]
    assert abs(again - r["lnl"]) < 1e-2          # 6-decimal branch lengths in the returned string
