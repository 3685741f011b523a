"""Oracle parsimony (`raxmlHPC -y` restatement, oracle/pml_oracle.c): Fitch length against an
independent numpy implementation, brute-force minimum length over all 5-/6-taxon topologies,
and the hill-climbing invariants (SPR never lengthens the tree; seed 0 = input order)."""
import itertools
import sys, os

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import util
from oracle import po
from pepr_amd import synth


def _all_unrooted(names):
    """every unrooted binary topology on `names` as Newick (stepwise edge insertion)"""
    def insert_everywhere(tree, x):
        out = []
        def rec(node, rebuild):
            out.append(rebuild(("join", node, x)))
            if isinstance(node, tuple) and node[0] == "join":
                _, l, r = node
                rec(l, lambda s: rebuild(("join", s, r)))
                rec(r, lambda s: rebuild(("join", l, s)))
        a, b, c = tree
        rec(a, lambda s: (s, b, c)); rec(b, lambda s: (a, s, c)); rec(c, lambda s: (a, b, s))
        return out
    def fmt(nd):
        return nd if isinstance(nd, str) else "(%s,%s)" % (fmt(nd[1]), fmt(nd[2]))
    trees = [(names[0], names[1], names[2])]
    for x in names[3:]:
        trees = [t2 for t in trees for t2 in insert_everywhere(t, x)]
    return ["(%s,%s,%s);" % tuple(fmt(p) for p in t) for t in trees]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fitch_length_vs_numpy(seed):
    names, rows, nw = synth.simulate_alignment(9, 120, 300 + seed, missing_frac=0.05)
    rows = [r.replace("A", "B", 1).replace("E", "Z", 1) for r in rows]
    aln = po.Alignment(names, rows)
    t = po.Tree(nw, aln)
    assert po.parsimony_length(aln, t) == util.fitch_length(names, rows, t.newick())


@pytest.mark.parametrize("ntax,count", [(5, 15), (6, 105)])
def test_search_reaches_bruteforce_minimum(ntax, count):
    names, rows, nw = synth.simulate_alignment(ntax, 150, 77 + ntax)
    aln = po.Alignment(names, rows)
    tops = _all_unrooted(names)
    assert len(tops) == count
    best = min(util.fitch_length(names, rows, t) for t in tops)
    for seed in (0, 5, 9):
        tree, length, moves = po.parsimony_tree(aln, seed, radius=20)
        assert length == util.fitch_length(names, rows, tree.newick())
        assert length == best          # tiny trees: SPR neighbourhood covers everything


def test_spr_never_lengthens_and_seed_semantics():
    names, rows, nw = synth.simulate_alignment(24, 200, 41)
    aln = po.Alignment(names, rows)
    t0, l0, m0 = po.parsimony_tree(aln, 0, radius=0)      # stepwise addition only
    t1, l1, m1 = po.parsimony_tree(aln, 0, radius=20)
    assert m0 == 0 and l1 <= l0 and (m1 == 0) == (l1 == l0)
    ta, la, _ = po.parsimony_tree(aln, 7, radius=20)
    tb, lb, _ = po.parsimony_tree(aln, 7, radius=20)
    assert ta.rf(tb) == 0 and la == lb                     # deterministic per seed
    true = po.Tree(nw, aln)
    assert l1 <= po.parsimony_length(aln, true) + 5        # close to (usually below) the generating tree
    assert t1.rf(true) <= 6
