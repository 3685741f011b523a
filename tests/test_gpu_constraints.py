"""Topological constraints (FastTree -constraints as PEPR builds them, FastTreeRunner.java:54-64,
243-273).  No reference binary can be run, so these are property tests: the result
displays every constrained split; constraints the ML tree already satisfies change nothing; a
false constraint is honoured at a likelihood cost."""
import os
import subprocess

import numpy as np
import pytest

import util
from pepr_amd import engine, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _displays(newick, clade, taxa):
    sp = util.splits(newick)
    c = frozenset(clade); allt = frozenset(taxa)
    return c in sp or (allt - c) in sp


def test_true_constraints_change_nothing(gpu_ctx):
    names, rows, nw = synth.simulate_alignment(14, 300, 901)
    free = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5)[0]
    cons = engine.constraints_from_tree(free["newick"])
    con = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5, constraints=cons)[0]
    assert engine.rf_distance(con["newick"], free["newick"]) == 0 and abs(con["lnl"] - free["lnl"]) < 1e-3


def test_false_constraint_is_honoured(gpu_ctx):
    names, rows, nw = synth.simulate_alignment(12, 300, 902)
    free = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5)[0]
    sp = util.splits(free["newick"])
    # pick a clade that the ML tree does NOT contain
    rng = np.random.default_rng(1)
    while True:
        clade = frozenset(rng.choice(names, 4, replace=False))
        if clade not in sp and frozenset(names) - clade not in sp:
            break
    cnames = list(names)
    crows = ["1" if t in clade else "0" for t in cnames]
    con = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5, constraints=(cnames, crows))[0]
    assert _displays(con["newick"], clade, names)
    assert con["lnl"] < free["lnl"] - 1e-3
    # a given start tree that violates the constraint is replaced, not trusted
    con2 = gpu_ctx.search([(names, rows)], [free["newick"]], nni=True, spr_radius=0, constraints=(cnames, crows))[0]
    assert _displays(con2["newick"], clade, names)


def test_partial_constraints_and_batch(gpu_ctx):
    """'-' entries and taxa missing from the matrix are free; one matrix applies to every gene of a batch"""
    genes = [synth.simulate_alignment(10, 200, 910 + i) for i in range(3)]
    names = genes[0][0]
    clade = ["t1", "t4", "t7"]
    cnames = [t for t in names if t != "t9"]                      # t9 not named at all
    crows = [("1" if t in clade else ("-" if t == "t0" else "0")) for t in cnames]
    out = gpu_ctx.search([(g[0], g[1]) for g in genes], None, nni=True, spr_radius=5, constraints=(cnames, crows))
    for r in out:
        sp = util.splits(r["newick"])
        ok = False
        for s in sp:                                               # some split separates the clade from the other constrained taxa
            for side in (s, frozenset(names) - s):
                if set(clade) <= side and not (side & (set(cnames) - set(clade) - {"t0"})):
                    ok = True
        assert ok, r["newick"]


def test_fasttree_shim_constraints(tmp_path, gpu_ctx):
    names, rows, nw = synth.simulate_alignment(10, 200, 920)
    with open(tmp_path / "g.faa", "w") as f:
        for n, r in zip(names, rows):
            f.write(">%s\n%s\n" % (n, r))
    clade = ["t2", "t5", "t8"]
    with open(tmp_path / "g.faa.con", "w") as f:                  # FastTreeRunner.java:54-64 writes <file>.con
        for n in names:
            f.write(">%s\n%s\n" % (n, "1" if n in clade else "0"))
    r = subprocess.run([os.path.join(ROOT, "bin", "FastTree_WAG"), "-gamma", "-nosupport", "-constraints", "g.faa.con", "g.faa"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert _displays(r.stdout.splitlines()[0], clade, names)


def _false_clade(names, newick, k, seed):
    sp = util.splits(newick)
    rng = np.random.default_rng(seed)
    while True:
        clade = frozenset(rng.choice(names, k, replace=False))
        if clade not in sp and frozenset(names) - clade not in sp:
            return clade


def test_constrained_search_vs_oracle(gpu_ctx, oracle_lib):
    """The oracle restates the constraint handling (constrained NJ, NNI and SPR candidate filters, replacement of a
    violating start tree: oracle/pml_oracle.c "Topological constraints"); the constrained search on the device and in the
    oracle must end in the same tree (RF 0 over resolved branches) with |dlnL| < 1e-3.  Cases: two conflicting (false)
    clades at once from the NJ start (the constrained NJ tree is used), a false clade with free ('-') and unnamed taxa from
    a GIVEN start tree that violates it, true constraints from a random start (filters only)."""
    po = oracle_lib
    cases = []
    # (a) two false clades, NJ start
    names, rows, nw = synth.simulate_alignment(13, 260, 930)
    c1 = _false_clade(names, nw, 4, 1); c2 = _false_clade([t for t in names if t not in c1], nw, 3, 2)
    crows = [("1" if t in c1 else "0") + ("1" if t in c2 else "0") for t in names]
    cases.append((names, rows, None, (list(names), crows), [c1, c2]))
    # (b) false clade, some taxa free, given (violating) start tree = the generating tree
    names, rows, nw = synth.simulate_alignment(11, 220, 931)
    c1 = _false_clade(names, nw, 4, 3)
    cn = [t for t in names if t != names[-1]]
    free_t = [t for t in cn if t not in c1][0]
    crows = [("-" if t == free_t else ("1" if t in c1 else "0")) for t in cn]
    cases.append((names, rows, nw, (cn, crows), []))
    # (c) true constraints (every split of the generating tree), random start tree
    names, rows, nw = synth.simulate_alignment(12, 240, 932)
    rng = np.random.default_rng(5)
    start = synth.random_tree(12, rng, [names[j] for j in rng.permutation(12)])[0]
    cases.append((names, rows, start, engine.constraints_from_tree(nw), []))
    for names, rows, start, cons, clades in cases:
        a = po.Alignment(names, rows); e = po.Engine(a, po.Model(0), 4, 1.0)
        e.set_constraints(cons[0], cons[1])
        ref_lnl, ref_tree = e.search(po.Tree(start, a) if start else None, 5, 1e-3)
        assert e.displays(ref_tree)
        g = gpu_ctx.search([(names, rows)], [start] if start else None, nni=True, spr_radius=5, epsilon=1e-3, constraints=cons)[0]
        assert util.rf_collapsed(g["newick"], ref_tree.newick(12)) == 0, (g["newick"], ref_tree.newick(6))
        assert abs(g["lnl"] - ref_lnl) < 1e-3, (g["lnl"], ref_lnl)
        for c in clades:
            assert _displays(g["newick"], c, names)
