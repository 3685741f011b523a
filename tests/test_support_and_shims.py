"""Support decoration (TreeSupportDecorator.java:86-163 semantics) and the CLI shims' argument
handling -- no GPU needed.  The shims' numeric paths are covered by tests/test_gpu_shims.py."""
import os
import subprocess

import pytest

from pepr_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FT = os.path.join(ROOT, "bin", "FastTree_WAG")
RX = os.path.join(ROOT, "bin", "raxmlHPC")


def test_support_counts():
    main = "((a:1,b:1):0.5,(c:1,d:1):0.4,(e:1,f:1):0.3);"
    sup = ["((a:1,b:1):1,(c:1,d:1):1,(e:1,f:1):1);", "((a:1,c:1):1,(b:1,d:1):1,(e:1,f:1):1);",
           "(a:1,(b:1,(c:1,(d:1,(e:1,f:1):1):1):1):1);"]
    out = engine.support_tree(main, sup, 2)
    assert out == "(a:1.00,b:1.00,((c:1.00,d:1.00)1:0.40,(e:1.00,f:1.00)3:0.30)2:0.50);"
    assert engine.rf_distance(out, main) == 0                  # labels are parsed back as supports
    assert engine.support_tree(main, [], 2).count(")0:") == 3
    # every support tree identical to the main tree -> all counts = number of trees (README:19-20:
    # "100% support for all branches")
    full = engine.support_tree(main, [main] * 100, 3)
    assert full.count(")100:") == 3
    with pytest.raises(engine.PmlError):
        engine.support_tree(main, ["((a:1,b:1):1,(c:1,x:1):1,(e:1,f:1):1);"])


def test_shims_exist_and_reject_bad_usage(tmp_path):
    for exe in (FT, RX, RX + "-PTHREADS"):
        assert os.access(exe, os.X_OK), exe
    def run(args):
        return subprocess.run(args, cwd=tmp_path, capture_output=True, text=True)
    r = run([FT, "-gamma", "-nosupport", "missing.faa"])
    assert r.returncode != 0 and "cannot open" in r.stderr and r.stdout == ""
    r = run([FT, "-gtr", "-nt", "x.faa"])
    assert r.returncode != 0 and "nucleotide" in r.stderr
    r = run([RX, "-f", "d", "-m", "PROTGAMMAWAG", "-s", "x.phy", "-n", "r1", "-Y", "-N", "10"])
    assert r.returncode != 0 and "parsimony bootstrap" in r.stderr
    r = run([RX, "-f", "d", "-m", "GTRGAMMA", "-s", "x.phy", "-n", "r1"])
    assert r.returncode != 0 and "WAG" in r.stderr
    (tmp_path / "RAxML_info.r2").write_text("old run\n")          # RAxML refuses a used run name (RAxMLRunner.java:518-532)
    r = run([RX, "-f", "d", "-m", "PROTGAMMAWAG", "-s", "x.phy", "-n", "r2"])
    assert r.returncode != 0 and "already exist" in r.stderr
    (tmp_path / "bad.phy").write_text("3 5\na AAAAA\nb AAAA\n")
    r = run([RX, "-f", "d", "-m", "PROTGAMMAWAG", "-s", "bad.phy", "-n", "r3"])
    assert r.returncode != 0


def test_shim_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    (tmp_path / "a.faa").write_text(">a\nARND\n>b\nARNE\n>c\nAQND\n>d\nGRND\n")
    r = subprocess.run([FT, "-gamma", "-nosupport", "a.faa"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 and "HIP device" in r.stderr and r.stdout == ""


def test_concatenate_matches_reference_rules():
    """MSAConcatenator.concatenate (MSAConcatenator.java:78-189): sorted taxon union, '?' padding."""
    g = [(["b", "a", "c"], ["AR", "NN", "DD"]), (["c", "d", "a"], ["KKK", "LLL", "MMM"]), (["a", "b", "c", "d"], ["W", "Y", "V", "F"])]
    names, rows = engine.concatenate(g)
    assert names == ["a", "b", "c", "d"] and rows == ["NNMMMW", "AR???Y", "DDKKKV", "??LLLF"]
    names, rows = engine.concatenate(g, [2, 0])
    assert names == ["a", "b", "c", "d"] and rows == ["WNN", "YAR", "VDD", "F??"]
    names, rows = engine.concatenate(g, [1])
    assert names == ["a", "c", "d"] and rows == ["MMM", "KKK", "LLL"]
    with pytest.raises(engine.PmlError):
        engine.concatenate(g, [7])


def test_refine_next_follows_refiner_rules():
    """PhylogeneticTreeRefiner.getNextIndexToRefine (:298-359) + AdvancedTree.getMeanDescendantSupportValues
    (:1061-1098), worked by hand: supports default to 100 (tips too), mean = floor(sum/count) over ALL
    descendants including tips; the scan starts at the third node in preorder."""
    from pepr_amd import engine
    nw = "((a:1,b:1)100:1,((c:1,d:1)60:1,(e:1,(f:1,g:1)100:1)100:1)100:1,h:1);"
    ingroup, means = engine.refine_next(nw, 100)
    # nodes in order of appearance: root, (a,b), a, b, (cdefg), (c,d), c, d, (e,(f,g)), e, (f,g), f, g, h
    assert len(means) == 14
    assert means[1] == 100 and means[5] == 100 and means[10] == 100
    # (cdefg): descendants (c,d)=60, c, d, (e,(f,g))=100, e, (f,g)=100, f, g -> (60 + 7*100)/8 = 95
    assert means[4] == 95
    assert ingroup == ["c", "d", "e", "f", "g"]              # mean 95 < 100, own support 100, a child below cutoff
    assert engine.refine_next(nw, 100, done=[ingroup])[0] is None
    # cutoff 50: everything is above it
    assert engine.refine_next(nw, 50)[0] is None
    # bracket form and fractional supports (FastTree's 0-1 scale) read the same
    nw2 = "((a:1,b:1):1[100],((c:1,d:1)0.6:1,(e:1,(f:1,g:1)1.0:1):1[100])1.00:1,h:1);"
    assert engine.refine_next(nw2, 100)[0] == ["c", "d", "e", "f", "g"]
    # the second node in preorder (first child of the root) is never offered (loop starts at index 2)
    nw3 = "(((a:1,b:1)10:1,c:1)100:1,d:1,e:1);"
    assert engine.refine_next(nw3, 100)[0] is None
    with pytest.raises(engine.PmlError):
        engine.refine_next("((a,b)x9,c);", 100)


def test_raxml_shim_f_b_draws_bipartitions(tmp_path):
    """`raxmlHPC -f b -z trees -t tree` (RAxMLRunner.getSupportDecoratedTree :453-516): host-only, percent labels"""
    import re
    (tmp_path / "main.nwk").write_text("((a:0.1,b:0.1):0.1,(c:0.1,d:0.1):0.1,e:0.1);\n")
    (tmp_path / "sup.nwk").write_text("((a,b),(c,d),e);\n((a,b),(c,e),d);\n((a,c),(b,d),e);\n((a,b),c,(d,e));\n")
    (tmp_path / "x.phy").write_text("5 2\na AR\nb AR\nc AR\nd AR\ne AR\n")
    r = subprocess.run([RX, "-f", "b", "-z", "sup.nwk", "-t", "main.nwk", "-T", "2", "-m", "PROTGAMMAWAG", "-n", "b1", "-s", "x.phy"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = (tmp_path / "RAxML_bipartitions.b1").read_text().strip()
    assert sorted(int(x) for x in re.findall(r"\)(\d+):", out)) == [25, 75]        # (c,d) in 1 of 4, (a,b) in 3 of 4
    assert out.endswith(":0.0;")


def test_jackknife_draw_host_only():
    """pml_jackknife_draw (RandomSetUtils.java:9-35 restated with a seed): subsets without replacement, ascending,
    deterministic per seed, half the genes by default (PhylogenomicPipeline2.java:1599-1617)."""
    from pepr_amd import engine
    d = engine.jackknife_draw(11, 40, 0, 7)
    assert len(d) == 40 and all(len(s) == 5 and s == sorted(set(s)) and 0 <= s[0] and s[-1] < 11 for s in d)
    assert d == engine.jackknife_draw(11, 40, 0, 7) and d != engine.jackknife_draw(11, 40, 0, 8)
    assert len({tuple(s) for s in d}) > 20                      # genuinely different subsets
    assert engine.jackknife_draw(4, 3, 9, 1) == [[0, 1, 2, 3]] * 3   # subset larger than the gene list = all genes
    counts = [sum(g in s for s in engine.jackknife_draw(10, 2000, 0, 3)) for g in range(10)]
    assert min(counts) > 850 and max(counts) < 1150             # every gene drawn about half the time
