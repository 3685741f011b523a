"""diagnostic: is the scoring step (k_pmat + k_oplist + k_reduce) bit-reproducible when repeated?"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pepr_amd import engine, synth
ctx = engine.Context(0)
genes = [synth.simulate_alignment(9 + i % 4, 160 + 10 * i, 740 + i) for i in range(8)]
b = engine.Batch(ctx, [(g[0], g[1]) for g in genes], [g[2] for g in genes], alpha=0.8667)
ref = b.score().copy()
stop = False
def noise():
    c2 = engine.Context(0)
    big = [synth.simulate_alignment(50, 1000, 1 + i) for i in range(32)]
    b2 = engine.Batch(c2, [(g[0], g[1]) for g in big], [g[2] for g in big], alpha=0.8)
    while not stop:
        b2.score()
    b2.close(); c2.close()
for phase in ("quiet", "with a second context scoring concurrently"):
    th = None
    if phase != "quiet":
        th = threading.Thread(target=noise); th.start()
    bad = 0
    for it in range(3000):
        if it % 7 == 0: b.set_alpha(0.8667)           # forces fresh matrices like the alpha optimisation does
        l = b.score()
        if not np.array_equal(l, ref):
            bad += 1
            if bad <= 5: print(phase, "iteration", it, "differs:", (l - ref)[l != ref], np.nonzero(l != ref)[0])
    print(phase, ": differing evaluations", bad, "of 3000", flush=True)
    if th: stop = True; th.join()
