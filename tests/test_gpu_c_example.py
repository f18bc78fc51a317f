"""The C ABI is sufficient on its own: examples/gate_demo.c (plain C99, no Python, no Julia) builds
against include/sgfhe_hip.h, links libsgfhe_hip.so, and runs the reference's README flow --
keys, encrypt, split, gate bootstraps on the GPU, decrypt -- to a correct truth table."""

import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_host_runs_the_readme_flow(S, tmp_path):
    lib_dir = os.path.dirname(S.build())
    exe = str(tmp_path / "gate_demo")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "gate_demo.c"), "-L", lib_dir, "-lsgfhe_hip",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "gate_demo OK: 32 AND / OR / XOR gate bootstraps" in r.stdout
