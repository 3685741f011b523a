"""Device parsimony (`raxmlHPC -f d -y`, RAxMLRunner.java:215-251) against the C oracle:
integer work, so trees and Fitch lengths must be identical, not merely close."""
import os
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import util
from oracle import po
from pepr_amd import engine, synth, tree_builder as tb

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle(names, rows, seed, radius):
    aln = po.Alignment(names, rows)
    t, length, moves = po.parsimony_tree(aln, seed, radius)
    return t.newick(), length, moves


@pytest.mark.parametrize("ntax,nsites,seed,radius", [(5, 60, 0, 20), (12, 300, 0, 0), (12, 300, 3, 20), (30, 500, 11, 20), (50, 1000, 7, 5)])
def test_parsimony_vs_oracle(gpu_ctx, ntax, nsites, seed, radius):
    names, rows, nw = synth.simulate_alignment(ntax, nsites, 1000 + ntax)
    g = gpu_ctx.parsimony([(names, rows)], seed=seed, spr_radius=radius)[0]
    onw, olen, omoves = _oracle(names, rows, seed, radius)
    assert g["length"] == olen
    assert engine.rf_distance(g["newick"], onw) == 0
    assert g["length"] == util.fitch_length(names, rows, g["newick"])      # independent numpy Fitch


def test_parsimony_batch_ragged_and_ambiguity(gpu_ctx):
    """genes of different shapes in one batch; '?' blocks, gaps, B/Z/X; more than 256 and fewer than 32 patterns"""
    genes = []
    for i, (n, m) in enumerate([(7, 20), (16, 700), (9, 257), (24, 90), (4, 50), (3, 10)]):
        names, rows, _ = synth.simulate_alignment(n, m, 2000 + i, missing_frac=0.1 if i % 2 else 0.0)
        rows = [r[:3] + "BZX-?"[j % 5] + r[4:] for j, r in enumerate(rows)]
        genes.append((names, rows))
    out = gpu_ctx.parsimony(genes, seed=5, spr_radius=20)
    for (names, rows), g in zip(genes, out):
        onw, olen, _ = _oracle(names, rows, 5, 20)
        assert g["length"] == olen == util.fitch_length(names, rows, g["newick"])
        assert engine.rf_distance(g["newick"], onw) == 0


def test_parsimony_sub_batching(gpu_ctx, monkeypatch):
    """gene lists larger than the HBM budget run as consecutive sub-batches with identical results"""
    genes = [(lambda t: (t[0], t[1]))(synth.simulate_alignment(10 + i, 150 + 20 * i, 3000 + i)) for i in range(6)]
    whole = gpu_ctx.parsimony(genes, seed=2, spr_radius=20)
    monkeypatch.setenv("PML_HBM_BUDGET_MB", "1")
    parts = gpu_ctx.parsimony(genes, seed=2, spr_radius=20)
    assert [(p["newick"], p["length"]) for p in parts] == [(w["newick"], w["length"]) for w in whole]


def test_parsimony_large_patterns_single_gene(gpu_ctx):
    """one long concatenation (many pattern blocks, prune groups spread over workgroups)"""
    names, rows, nw = synth.simulate_alignment(20, 6000, 77)
    g = gpu_ctx.parsimony([(names, rows)], seed=1, spr_radius=20)[0]
    onw, olen, _ = _oracle(names, rows, 1, 20)
    assert g["length"] == olen and engine.rf_distance(g["newick"], onw) == 0


def test_parsimony_bl_builder_and_shim(gpu_ctx, tmp_path):
    names, rows, nw = synth.simulate_alignment(10, 300, 55)
    b = tb.PhylogeneticTreeBuilder(gpu_ctx)
    b.setAlignment(tb.SequenceAlignment(names, rows)); b.setTreeBuildingMethod(tb.PARSIMONY_BL); b.run()
    t_bl = b.getTreeString()
    b.setTreeBuildingMethod(tb.PARSIMONY); b.run()
    t_mp = b.getTreeString()
    assert ":" not in t_mp and ":" in t_bl and engine.rf_distance(t_mp, t_bl) == 0
    # the two raxml invocations of runRaxmlParsimonyWithBranchLengths (RAxMLRunner.java:241-272)
    with open(tmp_path / "a.phy", "w") as f:
        f.write("%d %d\n" % (len(names), len(rows[0])))
        for n, r in zip(names, rows):
            f.write("%s %s\n" % (n, r))
    exe = os.path.join(ROOT, "bin", "raxmlHPC")
    r = subprocess.run([exe, "-f", "d", "-m", "PROTGAMMAWAG", "-s", "a.phy", "-n", "r1", "-y"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    mp = open(tmp_path / "RAxML_parsimonyTree.r1").read().strip()
    assert not os.path.exists(tmp_path / "RAxML_result.r1")
    r = subprocess.run([exe, "-f", "e", "-m", "PROTGAMMAWAG", "-s", "a.phy", "-n", "r1BL", "-t", "RAxML_parsimonyTree.r1"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    bl = open(tmp_path / "RAxML_result.r1BL").read().strip()
    assert engine.rf_distance(mp, bl) == 0 and engine.rf_distance(mp, t_mp) == 0


def test_search_from_parsimony_start_vs_oracle(gpu_ctx):
    """`raxmlHPC -f d -p seed` starts from a randomised stepwise-addition parsimony tree: pml_search_opts.seed != 0.
    Same start tree on both sides (integer work), then the usual search parity (RF 0, |dlnL| < 1e-3)."""
    names, rows, nw = synth.simulate_alignment(16, 300, 5100)
    g = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5, seed=4242)[0]
    aln = po.Alignment(names, rows)
    start, _, _ = po.parsimony_tree(aln, 4242, 20)
    e = po.Engine(aln, po.Model(0), 4, 1.0)
    lnl, tree = e.search(start, 5, 1e-3)
    assert util.rf_collapsed(g["newick"], tree.newick(12)) == 0 and abs(g["lnl"] - lnl) < 1e-3
    nj = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5)[0]
    assert abs(nj["lnl"] - g["lnl"]) < 5.0          # two starts, one likelihood surface: same neighbourhood
