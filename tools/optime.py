"""Shader-clock cycles per operation of the chained C3 scoring pass, by the kinds of its sides (diagnostic builds with
-DPML_OPTIME, see tools/optime.sh).  Per kind: operations per launch, mean cycles per operation (wave 0..3 of a tile each
count), MFMAs the operation issues per wave, and cycles per MFMA (16 = the matrix pipe's own time)."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pepr_amd import engine, synth, _lib
genes = [synth.simulate_alignment(50, 1000, 1 + i, 0.8) for i in range(128)]
ctx = engine.Context(0)
b = engine.Batch(ctx, [(g[0], g[1]) for g in genes], [g[2] for g in genes], alpha=0.8)
lib = ctypes.CDLL(os.environ["PEPRML_LIB"])
buf = (ctypes.c_ulonglong * 512)()
for _ in range(200): b.score()
lib.pml_abl_optime(buf, 1)
N = 20
for _ in range(N): b.score()
lib.pml_abl_optime(buf, 0)
a = np.array(buf[:], dtype=np.float64).reshape(256, 2)
KN = ["clv", "tip", "cherry", "pitch"]; MN = ["newview", "sumtable", "evaluate", "evalcat"]
def mfmas(lk, rk, mode, chained):
    # per wave: 100 per contracted side and category-set (4 x 25 k-steps x ... = 2 patterns x 50), tips looked up = 0, cherry = 0 (+0), pitch = +100 inner
    per = {0: 100, 1: 0, 2: 0, 3: 100}
    n = (per[lk] + (100 if lk in (2, 3) else 0)) + (per[rk] + (100 if rk in (2, 3) else 0))
    return n * 2
life, bar = a[238].copy(), a[243].copy(); a[230:] = 0
tot = a[:, 0].sum()
if life[1]: print("  wave lifetimes: %.0f waves/launch, %.0f cycles each; inside operations %.3f of it, waiting at the per-operation barrier (+ staged fragments) %.3f, rest (start-up, descriptors, staging) %.3f"
                  % (life[1] / N, life[0] / life[1], tot / life[0], bar[0] / life[0], 1 - (tot + bar[0]) / life[0]))
print("%s: total %.3f Gcycles over %d launches" % (os.environ.get("TAG"), tot / 1e9, N))
rows = []
for k in range(256):
    if a[k, 1] == 0: continue
    lk, rk, mode, ch, ns = k & 3, (k >> 2) & 3, (k >> 4) & 3, (k >> 6) & 1, (k >> 7) & 1
    rows.append((a[k, 0], "%-8s L=%-6s R=%-6s %s%s" % (MN[mode], KN[lk], KN[rk], "chained " if ch else "        ", "nostore" if ns else "       "), a[k, 1] / N, a[k, 0] / a[k, 1]))
for r in sorted(rows, reverse=True):
    print("  %s  ops/launch %8.0f  cycles/op %8.0f  share %.3f" % (r[1], r[2], r[3], r[0] / tot))
