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
    assert len(off["caterpillar"]) == 65 and float.fromhex(off["caterpillar"][0]) < -10000      # the rescue path was exercised
