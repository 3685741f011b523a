"""Where the wave-time of a C3 NNI search goes in the op-list kernels (diagnostic build -DPML_OPTIME): per variant (fused-Newton
smoothing launches = k_oplist<15>, everything else = k_oplist<11>) the shares of a wave's lifetime: slot claim + descriptor, tip
table + first fragment staging, inside operations, fused Newton, per-operation barrier."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pepr_amd import engine, synth
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 128
genes = synth.simulate_genes(ng, 50, 1000)
G = [(g[0], g[1]) for g in genes]
ctx = engine.Context(0)
lib = ctypes.CDLL(os.environ["PEPRML_LIB"])
buf = (ctypes.c_ulonglong * 512)()
ctx.search(G, None, nni=True, spr_radius=0, epsilon=1e-3)
lib.pml_abl_optime(buf, 1)
ctx.search(G, None, nni=True, spr_radius=0, epsilon=1e-3)
lib.pml_abl_optime(buf, 0)
a = np.array(buf[:], dtype=np.float64).reshape(256, 2)
for name, b in (("k_oplist<15> (fused Newton)", 230), ("k_oplist<11>", 238)):
    n, life = a[b, 1], a[b, 0]
    if n == 0: continue
    print("%s: %d waves, %.0f cycles each: claim+descriptor %.3f, table+first staging %.3f, operations %.3f, Newton %.3f, barrier %.3f, rest %.3f"
          % (name, n, life / n, a[b + 1, 0] / life, a[b + 2, 0] / life, a[b + 3, 0] / life, a[b + 4, 0] / life, a[b + 5, 0] / life,
             1 - (a[b + 1, 0] + a[b + 2, 0] + a[b + 3, 0] + a[b + 4, 0] + a[b + 5, 0]) / life))
if a[246, 1]:
    print("exchange gathers: %d, %.0f cycles each (wave 1 of a tile, from publishing its sums to having everybody's); the first gather of a request %.0f cycles on average over %d requests"
          % (a[246, 1], a[246, 0] / a[246, 1], a[247, 0] / max(a[230, 1] / 4, 1), a[230, 1] / 4))
