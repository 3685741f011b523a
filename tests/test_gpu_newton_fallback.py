"""k_newton's forward-progress design (kernels.hip "FORWARD PROGRESS"; VERDICT r02 item 2, gpurun_out/r05e.err): slices are
claimed by ticket (no reliance on dispatch order), waits are bounded in wall-clock time, and a wait that gives up re-issues
the work through the no-exchange form with the bits of the split form -- the call succeeds."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from pepr_amd import engine, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "newton_harness.py")], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout)


def test_giveup_falls_back_to_the_no_exchange_form_with_identical_bits():
    normal = _run({})
    forced = _run({"PML_NEWTON_TIMEOUT_US": "0"})        # every split exchange gives up at its first unsuccessful poll
    assert max(normal["npat"]) > 8192 and min(normal["npat"]) <= 128        # streaming form and single-slice requests are covered
    assert normal["fallbacks"] == {"giveups": 0, "reissued": 0, "seq_launches": 0}
    fb = forced["fallbacks"]
    assert fb["giveups"] > 0 and fb["reissued"] > 0 and fb["seq_launches"] > 0
    for key in normal:
        if key != "fallbacks":
            assert forced[key] == normal[key], key


def test_fused_newton_has_the_bits_of_the_unfused_form():
    """kernels.h OPF_FUSED_NEWTON: a smoothing step's branch Newton iterated inside k_oplist<15> on the register-resident
    sumtable tile against PML_NO_FUSE=1 (sumtable stored, k_newton on it): every result bit for bit -- the unfused form is the
    fused one's fallback, so the two must be interchangeable at any step"""
    fused = _run({})
    unfused = _run({"PML_NO_FUSE": "1"})
    for key in fused:
        assert fused[key] == unfused[key], key


def test_search_groups_return_the_bits_of_the_undivided_call():
    """api.cpp deals a search call of >= 64 small genes over groups (worker contexts, own streams and host threads): the genes'
    results must not depend on the division -- undivided, the default (two groups) and three groups agree bit for bit"""
    def run(groups):
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        if groups is not None:
            env["PML_GROUPS"] = str(groups)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "groups_harness.py")], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads(p.stdout)
    one, default, three = run(1), run(None), run(3)
    for key in ("nni", "spr"):
        assert default[key] == one[key] and three[key] == one[key], key
    assert default["stats"]["newview"] > one["stats"]["newview"]        # the default really ran as more than one batch
    # and with every exchange forced to give up inside the groups (each worker context falls back on its own): the same bits
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), PML_NEWTON_TIMEOUT_US="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "groups_harness.py")], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    forced = json.loads(p.stdout)
    for key in ("nni", "spr"):
        assert forced[key] == one[key], key
    assert forced["fallbacks"]["giveups"] > 0
    # an NNI round deals a gene's edges over independent runs of the launch (search.cpp, PendingOp::part): one run per gene, the
    # A-B arm, evaluates the same requests on the same inputs -- the same bits
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), PML_NNI_PARTS="1", PML_GROUPS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "groups_harness.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    unparted = json.loads(p.stdout)
    for key in ("nni", "spr"):
        assert unparted[key] == one[key], key


def test_two_contexts_search_concurrently_in_one_process(gpu_ctx):
    """two contexts (two streams) of ONE process searching at the same time: both succeed and each gene gets the bits of a
    lone search -- the layout whose spinning slices could starve each other under the old in-grid-order assumption"""
    sets = [[synth.simulate_alignment(20 + 3 * (i % 3), 500 + 100 * (i % 4), 7000 + 100 * k + i) for i in range(12)] for k in range(2)]
    lone = [gpu_ctx.search([(g[0], g[1]) for g in S], None, nni=True, spr_radius=5, seed=9) for S in sets]
    ctxs = [engine.Context(0), engine.Context(0)]
    got, errs = [None, None], [None, None]
    barrier = threading.Barrier(2)

    def work(k):
        try:
            barrier.wait()
            got[k] = [ctxs[k].search([(g[0], g[1]) for g in sets[k]], None, nni=True, spr_radius=5, seed=9) for _ in range(2)]
        except Exception as e:      # noqa: BLE001 -- reported below
            errs[k] = e
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert errs == [None, None], errs
    fb = [c.newton_fallbacks() for c in ctxs]
    for k in range(2):
        for rep in got[k]:
            assert [(r["lnl"], r["alpha"], r["newick"]) for r in rep] == [(r["lnl"], r["alpha"], r["newick"]) for r in lone[k]]
    print("fallbacks while sharing the GPU:", fb)
    for c in ctxs:
        c.close()
