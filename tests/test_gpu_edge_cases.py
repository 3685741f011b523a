"""Edge cases of the rows added late in the round (parsimony, supports, bootstrap, constraints): smallest inputs,
degenerate data, and combinations."""
import re

import numpy as np
import pytest

import util
from pepr_amd import engine, synth

pytestmark = pytest.mark.gpu
LABEL = re.compile(r"\)([0-9.]+):")


def test_four_taxa_everything(gpu_ctx):
    """4 taxa: one internal edge, three topologies"""
    names, rows, nw = synth.simulate_alignment(4, 200, 6001)
    s = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5)[0]
    assert np.isfinite(s["lnl"])
    sh = gpu_ctx.sh_support([(names, rows)], [s["newick"]], alpha=s["alpha"], nboot=200, seed=3)[0]
    lab = LABEL.findall(sh["newick"])
    assert len(lab) == 1 and 0.0 <= float(lab[0]) <= 1.0
    p = gpu_ctx.parsimony([(names, rows)], seed=3, spr_radius=20)[0]
    assert p["length"] == util.fitch_length(names, rows, p["newick"])
    b = gpu_ctx.bootstrap((names, rows), reps=3, seed=1)
    assert len(b["replicates"]) == 3 and len(re.findall(r"\)(\d+):", b["newick"])) == 1
    # 3 taxa: no internal edge, nothing to search; supports / parsimony still answer
    n3, r3 = names[:3], rows[:3]
    assert np.isfinite(gpu_ctx.search([(n3, r3)], None)[0]["lnl"])
    assert gpu_ctx.parsimony([(n3, r3)])[0]["length"] == util.fitch_length(n3, r3, "(%s,%s,%s);" % tuple(n3))
    assert LABEL.findall(gpu_ctx.sh_support([(n3, r3)], ["(%s:0.1,%s:0.1,%s:0.1);" % tuple(n3)])[0]["newick"]) == []


def test_degenerate_columns_and_identical_sequences(gpu_ctx):
    names = ["s%d" % i for i in range(7)]
    base = "ARNDCQEGHILKMFPSTWYV" * 5
    rows = [base, base, base[:50] + "-" * 50, "?" * 100, base.replace("A", "G"), base[::-1], "X" * 100]
    p = gpu_ctx.parsimony([(names, rows)], seed=0, spr_radius=20)[0]
    assert p["length"] == util.fitch_length(names, rows, p["newick"])
    s = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5, seed=7)[0]      # parsimony start on degenerate data
    assert np.isfinite(s["lnl"]) and s["lnl"] < 0
    sh = gpu_ctx.sh_support([(names, rows)], [s["newick"]], alpha=s["alpha"], nboot=100)[0]
    assert all(0.0 <= float(x) <= 1.0 for x in LABEL.findall(sh["newick"]))


def test_constraints_with_spr_and_parsimony_start(gpu_ctx):
    names, rows, nw = synth.simulate_alignment(14, 250, 6003)
    clade = ["t0", "t5", "t9", "t12"]
    cons = (list(names), ["1" if t in clade else "0" for t in names])
    for seed in (0, 11):                       # NJ start and parsimony start (replaced when it violates the constraint)
        r = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5, seed=seed, constraints=cons)[0]
        sp = util.splits(r["newick"])
        assert frozenset(clade) in sp or frozenset(names) - frozenset(clade) in sp
    free = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5)[0]
    assert r["lnl"] <= free["lnl"] + 1e-3


def test_bootstrap_and_sh_support_on_ragged_batch(gpu_ctx):
    genes, trees = [], []
    for i, (n, m) in enumerate([(6, 90), (11, 310), (5, 33), (17, 140)]):
        names, rows, nw = synth.simulate_alignment(n, m, 6100 + i, missing_frac=0.15 * (i % 2))
        genes.append((names, rows)); trees.append(nw)
    opt = gpu_ctx.optimize(genes, trees)
    sh = gpu_ctx.sh_support(genes, [o["newick"] for o in opt], alpha=1.0, nboot=500, seed=9)
    for (names, rows), r in zip(genes, sh):
        lab = [float(x) for x in LABEL.findall(r["newick"])]
        assert len(lab) == len(names) - 3 and all(0 <= v <= 1 for v in lab)
        assert all(abs(v * 500 - round(v * 500)) < 1e-6 for v in lab)
