"""optimise (smoothing passes only) a few gene sets of given shapes; prints fallbacks: finds which shapes make the fused Newton give up"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import synth, engine
ctx = engine.Context(0)
for spec in sys.argv[1:]:
    nt, ns, ng = [int(x) for x in spec.split("x")]
    genes = [synth.simulate_alignment(nt, ns, 1 + i, 0.8) for i in range(ng)]
    b = engine.Batch(ctx, [(g[0], g[1]) for g in genes], [g[2] for g in genes], alpha=0.8)
    f0 = ctx.newton_fallbacks()["giveups"]
    t0 = time.time(); b.optimize(optimize_alpha=False, epsilon=1.0); dt = time.time() - t0
    print("%s: mpad tiles %s  optimize %.2f s  giveups %d" % (spec, sorted(set((p + 127) // 128 for p in b.npatterns())), dt, ctx.newton_fallbacks()["giveups"] - f0), flush=True)
    b.close()
