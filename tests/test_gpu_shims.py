"""CLI shims on the GPU box: same argv, files and stdout conventions as the tools PEPR spawns
(SURVEY.md Appendix A), results equal to the C-ABI path."""
import os
import subprocess

import numpy as np
import pytest

from pepr_amd import engine, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FT = os.path.join(ROOT, "bin", "FastTree_WAG")
RX = os.path.join(ROOT, "bin", "raxmlHPC")


def _write(tmp, names, rows):
    with open(tmp / "g.faa", "w") as f:          # SequenceAlignment.getAlignmentAsFasta (SequenceAlignment.java:405-416)
        for n, r in zip(names, rows):
            f.write(">%s\n%s\n" % (n, r))
    w = max(len(n) for n in names) + 1            # ...ExtendedPhylipUsingTaxonNames (:489-522)
    with open(tmp / "g.phy", "w") as f:
        f.write("%d %d\n" % (len(names), len(rows[0])))
        for n, r in zip(names, rows):
            f.write(n.ljust(w) + r + "\n")


def test_fasttree_shim(tmp_path, gpu_ctx):
    names, rows, nw = synth.simulate_alignment(14, 250, 71, missing_frac=0.1)
    _write(tmp_path, names, rows)
    r = subprocess.run([FT, "-gamma", "-nosupport", "g.faa"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    tree = r.stdout.splitlines()[0]               # FastTreeRunner.java:95-96: stdout line 0 is the tree
    assert tree.endswith(");") and "LogLk" in r.stderr
    ref = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=0, pi_mode=engine.PI_WAG_FULL)[0]
    assert engine.rf_distance(tree, ref["newick"]) == 0
    # -gamma: FastTree's closing line and a tree whose lengths carry the fitted rescale
    import re
    m = re.search(r"Gamma\(20\) LogLk = (-?[0-9.]+) alpha = ([0-9.]+) rescaling lengths by ([0-9.]+)", r.stderr)
    assert m, r.stderr
    g20 = gpu_ctx.gamma20([(names, rows)], [ref["newick"]])[0]
    assert abs(float(m.group(1)) - g20["lnl"]) < 2e-3 and abs(float(m.group(2)) - g20["alpha"]) < 2e-3 and abs(float(m.group(3)) - g20["rescale"]) < 2e-3
    tl = sum(float(x) for x in re.findall(r":([0-9.]+)", tree))
    assert abs(tl - ref["tree_length"] * g20["rescale"]) < 1e-3 * max(1.0, tl)
    # without -gamma the tree keeps the optimised lengths: 5-decimal lengths as FastTree prints them lose < 0.05 lnL
    r2 = subprocess.run([FT, "-nosupport", "g.faa"], cwd=tmp_path, capture_output=True, text=True)
    assert r2.returncode == 0 and "Gamma(20)" not in r2.stderr
    again = gpu_ctx.score([(names, rows)], [r2.stdout.splitlines()[0]], alpha=ref["alpha"], pi_mode=engine.PI_WAG_FULL)[0]["lnl"]
    assert abs(again - ref["lnl"]) < 0.05


def test_raxml_shim_modes(tmp_path, gpu_ctx):
    names, rows, nw = synth.simulate_alignment(10, 180, 72)
    _write(tmp_path, names, rows)
    (tmp_path / "in.nwk").write_text(nw + "\n")
    run = lambda args: subprocess.run([RX] + args, cwd=tmp_path, capture_output=True, text=True)
    # -f e : branch lengths + alpha on a fixed topology (FastTreeRunner.java:174-184)
    r = run(["-f", "e", "-m", "PROTGAMMAWAG", "-s", "g.phy", "-n", "e1", "-t", "in.nwk", "-T", "4"])
    assert r.returncode == 0, r.stderr
    res = (tmp_path / "RAxML_result.e1").read_text().strip()
    assert res.endswith("):0.0;")
    ref = gpu_ctx.optimize([(names, rows)], [nw])[0]
    info = (tmp_path / "RAxML_info.e1").read_text()
    lnl = float([l for l in info.splitlines() if "likelihood" in l][0].split(":")[1])
    assert abs(lnl - ref["lnl"]) < 1e-5 and engine.rf_distance(res, nw) == 0
    # -f g : per-site lnL file format (RAxMLRunner.java:196-213, 290-299)
    (tmp_path / "two.nwk").write_text(nw + "\n" + res + "\n")
    r = run(["-f", "g", "-m", "PROTGAMMAWAG", "-s", "g.phy", "-n", "g1", "-z", "two.nwk"])
    assert r.returncode == 0, r.stderr
    lines = (tmp_path / "RAxML_perSiteLLs.g1").read_text().splitlines()
    assert lines[0].split() == ["2", "180"] and lines[1].startswith("tr1\t") and lines[2].startswith("tr2\t")
    v1 = np.array([float(x) for x in lines[1].split("\t")[1].split()])
    assert len(v1) == 180 and abs(v1.sum() - ref["lnl"]) < 1e-3
    # -f d : search; RAxML_result + RAxML_bestTree + RAxML_info + RAxML_log
    r = run(["-f", "d", "-m", "PROTGAMMAWAG", "-s", "g.phy", "-n", "d1"])
    assert r.returncode == 0, r.stderr
    for f in ("RAxML_result.d1", "RAxML_bestTree.d1", "RAxML_info.d1", "RAxML_log.d1"):
        assert (tmp_path / f).exists()
    best = (tmp_path / "RAxML_result.d1").read_text().strip()
    sref = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5, seed=12345)[0]      # parsimony start, default -p
    assert engine.rf_distance(best, sref["newick"]) == 0
    # same run name again is refused
    assert run(["-f", "d", "-m", "PROTGAMMAWAG", "-s", "g.phy", "-n", "d1"]).returncode != 0


def test_bootstrap_api_shim_and_mirror(tmp_path, gpu_ctx):
    """`-f a -x seed -N reps` (RAxMLRunner.java:115-132, bootstrapReps > 0): supports are percentages of
    column-resampled replicate trees; files as SURVEY Appendix A lists them (no RAxML_result)."""
    import re
    from pepr_amd import tree_builder as tb
    import util
    names, rows, nw = synth.simulate_alignment(10, 400, 73)
    r = gpu_ctx.bootstrap((names, rows), reps=20, seed=7)
    assert len(r["replicates"]) == 20
    labels = [int(x) for x in re.findall(r"\)(\d+):", r["newick"])]
    assert len(labels) == 10 - 3 and all(0 <= v <= 100 for v in labels)
    # each label is exactly the share of replicates that contain the split
    main_splits = util.splits(re.sub(r"\)\d+:", "):", r["newick"]))
    rep_splits = [util.splits(t) for t in r["replicates"]]
    counts = sorted(int(0.5 + 100.0 * sum(s in rs or (frozenset(names) - s) in rs for rs in rep_splits) / 20) for s in main_splits
                    if 1 < len(s) < len(names) - 1)
    assert counts == sorted(labels)
    assert gpu_ctx.bootstrap((names, rows), reps=20, seed=7)["newick"] == r["newick"]          # seeded
    assert sum(labels) / len(labels) > 60                                                      # 400 sites: mostly well supported
    plain = gpu_ctx.search([(names, rows)], None, nni=True, spr_radius=5)[0]
    assert engine.rf_distance(re.sub(r"\)\d+:", "):", r["newick"]), plain["newick"]) == 0
    # shim
    _write(tmp_path, names, rows)
    p = subprocess.run([RX, "-f", "a", "-m", "PROTGAMMAWAG", "-s", "g.phy", "-n", "bs", "-x", "12345", "-N", "10"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    bip = open(tmp_path / "RAxML_bipartitions.bs").read().strip()
    assert re.search(r"\)\d+:", bip) and bip.endswith(":0.0;")
    assert len(open(tmp_path / "RAxML_bootstrap.bs").read().splitlines()) == 10
    assert os.path.exists(tmp_path / "RAxML_bestTree.bs") and not os.path.exists(tmp_path / "RAxML_result.bs")
    assert ")" in open(tmp_path / "RAxML_bestTree.bs").read() and not re.search(r"\)\d+:", open(tmp_path / "RAxML_bestTree.bs").read())
    # mirror: PhylogeneticTreeBuilder with bootstrapReps > 0 returns the supported tree (PhylogeneticTreeBuilder.java:175-179)
    b = tb.PhylogeneticTreeBuilder(gpu_ctx)
    b.setAlignment(tb.SequenceAlignment(names, rows)); b.setTreeBuildingMethod(tb.ML); b.setBootstrapReps(5); b.run()
    assert re.search(r"\)\d+:", b.getTreeString())
