"""Pins of the oracle's OPTIMISER and SEARCH that do not depend on how they were written: the optimum of a likelihood surface and the
best of all topologies are properties of the function, not of the control flow.  (The reference's own binaries cannot be run, so
these independent checks are what stands behind `-f e` / `-f d` parity: DESIGN.md section 3.)
  * `-f e` (RAxMLRunner.java:253-272, FastTreeRunner.java:142-199): the oracle's branch-length + alpha optimum against
    scipy.optimize on an independent numpy likelihood (tests/util.numpy_lnl);
  * `-f d` (RAxMLRunner.java:79-152): the oracle's NNI / SPR search against exhaustive enumeration of all 15 (5 taxa) and all 105
    (6 taxa) unrooted topologies, each optimised by the oracle -- the search must end in the best of them."""
import re
import sys, os

import numpy as np
import pytest
from scipy.optimize import minimize

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import util
from oracle import po
from pepr_amd import synth
from test_oracle_parsimony import _all_unrooted


def test_optimize_reaches_the_scipy_optimum_of_an_independent_likelihood():
    names, rows, nw = synth.simulate_alignment(5, 400, 77, 0.7)
    a = po.Alignment(names, rows)
    t = po.Tree(nw, a); e = po.Engine(a, po.Model(0), 4, 1.0)
    best = e.optimize(t, True, 1e-6)
    # the same surface in numpy: parameters = log branch lengths (in the order they appear in the Newick) + log alpha
    lengths = [float(x) for x in re.findall(r":([0-9.eE+-]+)", nw)]
    def with_lengths(ls):
        it = iter(ls)
        return re.sub(r":([0-9.eE+-]+)", lambda m: ":%.17g" % next(it), nw)
    def neg(x):
        ls = np.exp(x[:-1]); al = float(np.exp(x[-1]))
        return -util.numpy_lnl(names, rows, with_lengths(ls), al)[0]
    x0 = np.log(np.array([max(l, 1e-3) for l in lengths] + [1.0]))
    r = minimize(neg, x0, method="L-BFGS-B", options={"maxiter": 500, "ftol": 1e-13, "gtol": 1e-8})
    r = minimize(neg, r.x, method="Nelder-Mead", options={"xatol": 1e-9, "fatol": 1e-10, "maxiter": 4000})
    assert abs(best - (-r.fun)) < 2e-4, (best, -r.fun)            # the optimum is the function's, not the optimiser's
    assert best >= -r.fun - 2e-4 and abs(e.alpha - np.exp(r.x[-1])) < 2e-2 * e.alpha


@pytest.mark.parametrize("ntax,seed", [(5, 11), (6, 12)])
def test_search_ends_in_the_best_of_all_topologies(ntax, seed):
    names, rows, nw = synth.simulate_alignment(ntax, 300, 500 + seed, 0.9)
    a = po.Alignment(names, rows)
    m = po.Model(0)
    scores = []
    for top in _all_unrooted(list(names)):
        nwl = re.sub(r"([A-Za-z0-9_]+)", r"\1:0.1", top).replace(")", "):0.1").replace("):0.1;", ");")
        e = po.Engine(a, m, 4, 1.0)
        t = po.Tree(nwl, a)
        scores.append((e.optimize(t, True, 1e-4), t.newick(6)))
    scores.sort(reverse=True)
    assert len(scores) == {5: 15, 6: 105}[ntax]
    e = po.Engine(a, m, 4, 1.0)
    lnl, tree = e.search(None, 5, 1e-3)
    assert abs(lnl - scores[0][0]) < 2e-3, (lnl, scores[0][0], scores[1][0])
    assert util.rf_collapsed(tree.newick(6), scores[0][1]) == 0 or scores[0][0] - scores[1][0] < 1e-2      # (a tie between the two best would excuse it)
