"""Generates tests/golden/scoring_cases.json with the CPU oracle (oracle/pml_oracle.c).

The reference holds no golden vectors for this path and its bundled executables may not be run
in this pipeline, so these vectors pin the HIP engine to the ORACLE, not to the reference
("parity unpinned" -- see DESIGN.md).  Every case was additionally cross-checked at generation
time against tests/util.numpy_lnl (an independent numpy restatement).
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import po
from pepr_amd import synth
import util

CASES = [  # ntax, nsites, seed, alpha, missing_frac, pi_mode
    (3, 40, 21, 1.0, 0.0, 0), (4, 60, 3, 0.7, 0.0, 0), (5, 33, 4, 0.3, 0.2, 0), (8, 300, 5, 2.518330, 0.0, 0),
    (12, 200, 7, 0.8, 0.3, 0), (12, 200, 8, 0.5, 0.0, 1), (20, 150, 9, 0.05, 0.1, 0), (7, 1, 10, 1.0, 0.0, 0),
]

def main():
    out = []
    for nt, ns, seed, alpha, miss, pm in CASES:
        names, rows, nw = synth.simulate_alignment(nt, ns, seed, missing_frac=miss)
        if nt == 5:   # ambiguity codes and lower case, all-gap column
            rows = [r[:3] + "BZXbz-"[i % 6] + r[4:] for i, r in enumerate(rows)]
            rows = [r[:10] + "-" + r[11:] for r in rows]
        m = po.Model(pm)
        a = po.Alignment(names, rows); t = po.Tree(nw, a); e = po.Engine(a, m, 4, alpha)
        tot, sites = e.site_lnl(t)
        chk, _ = util.numpy_lnl(names, rows, nw, alpha, "raxml" if pm == 0 else "full")
        assert abs(chk - tot) < 1e-7 * max(1, abs(tot)), (chk, tot)
        t2 = po.Tree(nw, a); e2 = po.Engine(a, m, 4, alpha)
        opt = e2.optimize(t2, True, 1e-4)
        out.append({"names": names, "rows": rows, "newick": nw, "alpha": alpha, "pi_mode": pm, "npat": a.npat,
                    "lnl": tot, "site_lnl": [float("%.12g" % x) for x in sites],
                    "opt_lnl": opt, "opt_alpha": e2.alpha, "opt_tree_length": t2.length()})
        print(nt, ns, "lnl %.6f opt %.6f alpha %.5f" % (tot, opt, e2.alpha))
    with open(os.path.join(ROOT, "tests", "golden", "scoring_cases.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py (CPU oracle; parity unpinned vs reference binaries)", "cases": out}, f)

if __name__ == "__main__":
    main()
