"""SH-like local supports (FastTree's default output without -nosupport; FastTreeRunner.java:67-70) against the
oracle's restatement of FastTree's SHSupport.  The device resamples with the same counter hash, so the values agree
except where a resample's comparison is decided by the last bits of a sum (tolerance: 1 % of the resamples)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import util
from oracle import po
from pepr_amd import engine, synth, tree_builder as tb

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LABEL = re.compile(r"\)([0-9.]+):")


@pytest.mark.parametrize("ntax,nsites,seed", [(8, 200, 1), (14, 300, 2), (25, 150, 3)])
def test_sh_support_vs_oracle(gpu_ctx, ntax, nsites, seed):
    names, rows, nw = synth.simulate_alignment(ntax, nsites, 4000 + seed, missing_frac=0.1 * (seed == 2))
    start = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=0)[0]
    r = gpu_ctx.sh_support([(names, rows)], [start["newick"]], alpha=start["alpha"], nboot=1000, seed=77)[0]
    got = sorted(float(x) for x in LABEL.findall(r["newick"]))
    assert len(got) == ntax - 3 and all(0.0 <= v <= 1.0 for v in got)
    assert all(abs(v * 1000 - round(v * 1000)) < 1e-6 for v in got)           # multiples of 1/nboot
    assert abs(r["lnl"] - start["lnl"]) < 1e-3 and engine.rf_distance(re.sub(LABEL, "):", r["newick"]), start["newick"]) == 0
    a = po.Alignment(names, rows); e = po.Engine(a, po.Model(0), 4, start["alpha"])
    ref = sorted(e.sh_support(po.Tree(start["newick"], a), 1000, 77))
    assert np.max(np.abs(np.array(got) - np.array(ref))) <= 0.01 + 1e-9
    # seeded: same seed, same labels; the tree's well-supported splits stay well supported under another seed
    again = gpu_ctx.sh_support([(names, rows)], [start["newick"]], alpha=start["alpha"], nboot=1000, seed=77)[0]
    assert again["newick"] == r["newick"]
    other = sorted(float(x) for x in LABEL.findall(gpu_ctx.sh_support([(names, rows)], [start["newick"]], alpha=start["alpha"], nboot=1000, seed=78)[0]["newick"]))
    assert np.max(np.abs(np.array(got) - np.array(other))) < 0.08


def test_sh_support_meaning(gpu_ctx):
    """a split the data strongly support gets ~1, a split that contradicts the data (an NNI away from the ML tree)
    gets 0 (FastTree returns 0 when an alternative arrangement is more likely)"""
    names, rows, nw = synth.simulate_alignment(10, 600, 4100)
    ml = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5)[0]
    lab = [float(x) for x in LABEL.findall(gpu_ctx.sh_support([(names, rows)], [ml["newick"]], alpha=ml["alpha"])[0]["newick"])]
    assert np.mean(lab) > 0.8 and max(lab) > 0.99
    # batch of two different trees of the same gene: the true tree and a deliberately wrong one
    rng = np.random.default_rng(3)
    wrong = synth.random_tree(10, rng, [names[j] for j in rng.permutation(10)])[0]
    wopt = gpu_ctx.optimize([(names, rows)], [wrong])[0]
    out = gpu_ctx.sh_support([(names, rows), (names, rows)], [ml["newick"], wopt["newick"]], alpha=ml["alpha"])
    lw = [float(x) for x in LABEL.findall(out[1]["newick"])]
    assert min(lw) == 0.0 and np.mean(lw) < np.mean(lab)


def test_fasttree_shim_and_mirror_with_supports(tmp_path, gpu_ctx):
    names, rows, nw = synth.simulate_alignment(9, 300, 4200)
    with open(tmp_path / "g.faa", "w") as f:
        for n, r in zip(names, rows):
            f.write(">%s\n%s\n" % (n, r))
    exe = os.path.join(ROOT, "bin", "FastTree_WAG")
    p = subprocess.run([exe, "-gamma", "g.faa"], cwd=tmp_path, capture_output=True, text=True)      # no -nosupport
    assert p.returncode == 0, p.stderr
    tree = p.stdout.splitlines()[0]
    lab = [float(x) for x in LABEL.findall(tree)]
    assert len(lab) == 9 - 3 and all(0 <= v <= 1 for v in lab)
    q = subprocess.run([exe, "-gamma", "-nosupport", "g.faa"], cwd=tmp_path, capture_output=True, text=True)
    assert not LABEL.findall(q.stdout) and engine.rf_distance(re.sub(LABEL, "):", tree), q.stdout.splitlines()[0]) == 0
    # mirror: FastTreeRunner with bootstrapReps > 0 keeps the supports (FastTreeRunner.java:67-70)
    f = tb.FastTreeRunner(gpu_ctx); f.setAlignment(tb.SequenceAlignment(names, rows)); f.setBootstrapReps(100); f.run()
    assert len(LABEL.findall(f.getResult())) == 9 - 3
