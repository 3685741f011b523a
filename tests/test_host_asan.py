"""The host-side C++ of libpeprml (Newick dialect, encoder, NJ, RF / supports, refinement queries) compiled with
g++ -fsanitize=address,undefined and fed valid, odd and 20 000 mutated inputs (GPU sanitizers are not available on the
pool; this is the CPU build the task statement asks sanitizers to run on)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_under_asan(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "driver")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "host_asan", "driver.cpp"), os.path.join(ROOT, "pepr_amd", "csrc", "host.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "checks ok" in r.stdout


def test_oracle_under_asan(tmp_path):
    """the CPU oracle (plain C) with the same sanitizers: every public entry point once on small random data"""
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "oracle_driver")
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "host_asan", "oracle_driver.c"), os.path.join(ROOT, "oracle", "pml_oracle.c"), "-lm", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "oracle asan driver ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
