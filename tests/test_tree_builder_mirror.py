"""Host mirror of PhylogeneticTreeBuilder / RAxMLRunner / FastTreeRunner (reference
PhylogeneticTreeBuilder.java:97-129,168-196): dispatch and error behaviour (CPU) and a full run
through the C ABI (GPU)."""
import pytest

from pepr_amd import synth, tree_builder as tb


def test_dispatch_and_errors_cpu():
    b = tb.PhylogeneticTreeBuilder()
    b.setAlignment(tb.SequenceAlignment(["a", "b", "c"], ["AR", "AR", "AQ"]))
    b.setTreeBuildingMethod(tb.PARSIMONY)
    with pytest.raises(ValueError):
        b.run()                                   # outside the GPU path, said loudly
    with pytest.raises(ValueError):
        tb._model_from_matrix("GTRGAMMA")
    assert tb._model_from_matrix("PROTGAMMAWAG")["ncat"] == 4
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
