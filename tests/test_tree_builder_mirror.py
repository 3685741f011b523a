"""Host mirror of PhylogeneticTreeBuilder / RAxMLRunner / FastTreeRunner (reference
PhylogeneticTreeBuilder.java:97-129,168-196): dispatch and error behaviour (CPU) and a full run
through the C ABI (GPU)."""
import pytest

from pepr_amd import synth, tree_builder as tb


def test_dispatch_and_errors_cpu():
    b = tb.PhylogeneticTreeBuilder()
    b.setAlignment(tb.SequenceAlignment(["a", "b", "c"], ["AR", "AR", "AQ"]))
    b.setTreeBuildingMethod(tb.NEIGHBOR_JOINING)
    with pytest.raises(ValueError):
        b.run()                                   # outside the GPU path, said loudly
    with pytest.raises(ValueError):
        tb._model_from_matrix("GTRGAMMA")
    assert tb._model_from_matrix("PROTGAMMAWAG")["ncat"] == 4
    # -matrix_eval (PhylogenomicPipeline2.java:260-284) compares lnL across model names: a name that is not built must
    # never be run as plain WAG under its label
    from pepr_amd import engine
    assert tb._model_from_matrix("PROTGAMMAWAGF") == {"ncat": 4, "pi_mode": engine.PI_EMPIRICAL}      # the F variant IS built
    for other in ("PROTCATWAG", "PROTGAMMAIWAG", "PROTGAMMALGF", "PROTGAMMAJTT", "PROTMIXWAG", "PROTGAMMAWAGFX"):
        with pytest.raises(ValueError):
            tb._model_from_matrix(other)
    r = tb.RAxMLRunner()
    r.setAlignment(tb.SequenceAlignment(["a", "b", "c"], ["AR", "AR", "AQ"])); r.setMatrix("PROTGAMMALGF")
    with pytest.raises(ValueError):
        r.run()                                   # refused before anything touches the device
    b.setBootstrapReps(0); assert b.getBootstrapReps() == 0
    b.setRunName("x"); assert b.getRunName() == "x"
    b.setTreeString("(a,b,c);"); assert b.getTreeString() == "(a,b,c);"
    with pytest.raises(ValueError):
        b.setNucleotide(True)


@pytest.mark.gpu
def test_builder_runs_both_methods(gpu_ctx):
    from pepr_amd import engine
    names, rows, nw = synth.simulate_alignment(12, 300, 91)
    aln = tb.SequenceAlignment(names, rows)
    b = tb.PhylogeneticTreeBuilder(gpu_ctx)
    b.setAlignment(aln); b.setTreeBuildingMethod(tb.FAST_TREE); b.setBootstrapReps(0); b.run()
    ft = b.getTreeString()
    b2 = tb.PhylogeneticTreeBuilder(gpu_ctx)
    b2.setAlignment(aln); b2.setTreeBuildingMethod(tb.ML); b2.setMLMatrix("PROTGAMMAWAG"); b2.setProcesses(8); b2.setBootstrapReps(0); b2.run()
    ml = b2.getTreeString()
    assert ft and ml and engine.rf_distance(ft, nw) <= 2 and engine.rf_distance(ml, nw) <= 2
    # failed build -> null tree string, as FastTreeRunner.java:125-131
    bad = tb.FastTreeRunner(gpu_ctx); bad.setAlignment(tb.SequenceAlignment(["a", "b"], ["AR", "AR"])); bad.run()
    assert bad.getResult() is None
    # per-site lnL mode (RAxMLRunner.runRaxmlPerSiteLL :162-213)
    r = tb.RAxMLRunner(1, gpu_ctx); r.setAlignment(aln); r.setPerSiteLogLikelihoods(True); r.setPerSiteLLTrees([nw]); r.run()
    assert len(r.getPerSiteLLs()) == 1 and len(r.getPerSiteLLs()[0]) == 300


def test_constraint_tree_text_cpu():
    """setConstraintTree builds the FASTA text FastTreeRunner.java:243-273 writes to <aln>.con:
    taxa sorted, one 0/1 column per node of the tree, in the order nodes close."""
    f = tb.FastTreeRunner()
    f.setConstraintTree("((b:1,a:1):1,(c:1,(e:1,d:1):1):1);")
    names, rows = f._constraint_matrix()
    assert names == ["a", "b", "c", "d", "e"] and all(len(r) == 9 for r in rows)
    cols = ["".join(r[j] for r in rows) for j in range(9)]
    assert sorted(cols) == sorted(["01000", "10000", "11000", "00100", "00001", "00010", "00011", "00111", "11111"])
    assert f.constraints.startswith(">a\n")
    f2 = tb.FastTreeRunner(); f2.setConstraintTree(None)
    assert f2._constraint_matrix() is None


@pytest.mark.gpu
def test_builder_constraint_tree(gpu_ctx):
    import util
    names, rows, nw = synth.simulate_alignment(10, 200, 93)
    b = tb.PhylogeneticTreeBuilder(gpu_ctx)
    b.setAlignment(tb.SequenceAlignment(names, rows)); b.setTreeBuildingMethod(tb.FAST_TREE)
    b.setConstraintTree("((t1,t5,t8),(t0,t2,t3,t4,t6,t7,t9));")      # one multifurcating clade, rest free
    b.run()
    sp = util.splits(b.getTreeString())
    assert frozenset(["t1", "t5", "t8"]) in sp or frozenset(names) - frozenset(["t1", "t5", "t8"]) in sp
