"""Register chaining (kernels.h OPF_CHAIN_*) must not change a single bit: a child taken from the wave's
registers holds exactly the values that would have been read back, and swapping the two factors of a newview commutes.
The switch PML_CHAIN is read once per process, so each mode runs tests/chain_harness.py in its own process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, variant=None):
    env = dict(os.environ, PML_CHAIN=str(mode), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    if variant is not None:
        env["PML_CHAIN_VARIANT"] = str(variant)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "chain_harness.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout)


def test_chained_results_are_bit_identical_to_unchained():
    off = _run(0)                                  # every child written and read back (k_oplist<1>)
    for mode, variant in ((2, None), (1, None), (2, 9)):       # default; scoring passes only; chained kernel without LDS-DMA staging
        on = _run(mode, variant)
        for key in off:
            assert on[key] == off[key], (mode, variant, key)
    assert off["score_stored"] == off["score"] == off["score_again"] == off["score_stored_replay"]
    assert len(off["caterpillar"]) == 65 and float.fromhex(off["caterpillar"][0]) < -10000      # the rescue path was exercised


def _newview_kinds(nw, first_taxon):
    """inner-inner / tip-inner / tip-tip newviews of one full traversal towards `first_taxon` (where the engine evaluates),
    counted from the Newick alone; a bifurcating top node of a rooted Newick is a degree-2 node of the unrooted tree."""
    import itertools
    import util
    adj, ids = {}, itertools.count()

    def build(nd, parent):
        me = next(ids)
        adj[me] = {"tip": not nd[0], "nbr": [] if parent is None else [parent], "name": nd[1]}
        for k in nd[0]:
            adj[me]["nbr"].append(build(k, me))
        return me
    build(util.parse_newick(nw), None)
    kinds = {"ii": 0, "ti": 0, "tt": 0}

    def below(v, up):            # the real node hanging below (v, up): degree-2 nodes are passed through
        while not adj[v]["tip"] and len(adj[v]["nbr"]) == 2:
            v, up = next(w for w in adj[v]["nbr"] if w != up), v
        return v, up

    def walk(v, up):
        v, up = below(v, up)
        if adj[v]["tip"]:
            return True
        tips = sum(walk(w, v) for w in adj[v]["nbr"] if w != up)
        kinds["tt" if tips == 2 else ("ti" if tips == 1 else "ii")] += 1
        return False
    t0 = next(i for i, a in adj.items() if a["tip"] and a["name"] == first_taxon)
    walk(adj[t0]["nbr"][0], t0)
    return kinds


def test_flop_accounting_follows_survey_8d():
    """pml_kernel_flops counts SURVEY 8d's PER-OPERATION flops (inner-inner 6480, tip-inner 3280, tip-tip 80, evaluate 3360
    per pattern) -- what bench.py's roofline divides by the launch time.  Checked against a count made from the tree alone."""
    from pepr_amd import engine, synth
    ntax = 24
    names, rows, nw = synth.simulate_alignment(ntax, 300, 77)
    ctx = engine.Context(0, profile=True)
    b = engine.Batch(ctx, [(names, rows)], [nw], alpha=0.7)
    npat = b.npatterns()[0]
    kinds = _newview_kinds(nw, names[0])
    assert sum(kinds.values()) == ntax - 2
    expect = npat * (6480 * kinds["ii"] + 3280 * kinds["ti"] + 80 * kinds["tt"] + 3360)
    for stored in (False, True):
        ctx.kernel_stats(reset=True)
        b.score(stored=stored)
        st = ctx.kernel_stats()["newview"]
        assert st["launches"] == 1 and st["algo_flops"] == expect, (stored, st["algo_flops"], expect, kinds)
    assert expect < npat * (6480 * (ntax - 2) + 3360)                      # below 8d's upper bound ("ignoring tip savings")
    b.close(); ctx.close()
