"""Wall time of one CLI-shim invocation (process start + HIP init + code-object load + the inference), the cost a stock
pepr.jar pays per tree when it uses the shims instead of the JNI path."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepr_amd import synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names, rows, nw = synth.simulate_alignment(12, 9000, 31)
with tempfile.TemporaryDirectory() as d:
    with open(os.path.join(d, "g.faa"), "w") as f:
        for n, r in zip(names, rows):
            f.write(">%s\n%s\n" % (n, r))
    with open(os.path.join(d, "g.phy"), "w") as f:
        f.write("%d %d\n" % (len(names), len(rows[0])))
        for n, r in zip(names, rows):
            f.write("%s %s\n" % (n, r))
    for i in range(3):
        t0 = time.time(); p = subprocess.run([os.path.join(ROOT, "bin", "FastTree_WAG"), "-gamma", "-nosupport", "g.faa"], cwd=d, capture_output=True, text=True); dt = time.time() - t0
        print("FastTree_WAG shim, 12 taxa x 9000 columns: %.2f s wall (rc %d)" % (dt, p.returncode), flush=True)
    t0 = time.time(); p = subprocess.run([os.path.join(ROOT, "bin", "raxmlHPC"), "-f", "d", "-m", "PROTGAMMAWAG", "-s", "g.phy", "-n", "x1"], cwd=d, capture_output=True, text=True); dt = time.time() - t0
    print("raxmlHPC -f d shim, same alignment: %.2f s wall (rc %d)" % (dt, p.returncode))
