"""C2-like run on the GPU box: Aquificales stand-in alignments, full tree + 100 jackknife trees."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import engine
name = sys.argv[1] if len(sys.argv) > 1 else "Aquificales"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
d = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "standin_%s.json" % name)))
genes = [(g["names"], g["rows"]) for g in d["genes"]]
ctx = engine.Context(0)
t0 = time.time(); r = ctx.jackknife(genes, reps=reps, seed=1, spr_radius_full=5); dt = time.time() - t0
print("%s: %d genes, %d taxa, %d columns (%d patterns); full tree lnL %.3f alpha %.4f; %d support trees; %.2f s total" % (
    name, len(genes), len(d["taxa"]), r["nsites"], r["npatterns"], r["lnl"], r["alpha"], reps, dt))
print(r["newick"])
