"""The recorded expectations of the GPU suite (tests/expect.py, tests/golden/gpu_expect.json) are what the oracle
gives TODAY: a sample of them is recorded again here, on the CPU, by the oracle alone, into a scratch table, and
compared with the committed one.  A table that was edited by hand, or left behind by a change of the oracle or of a
test's inputs, fails here in the CPU suite -- before a GPU box falls back to the live oracle for every stale tag."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_expectations_are_what_the_oracle_gives(tmp_path):
    committed = json.load(open(os.path.join(ROOT, "tests", "golden", "gpu_expect.json")))
    assert len(committed) >= 160 and all(len(v["sha256"]) == 64 and v["shape"] and v["dtype"] for v in committed.values())
    scratch = str(tmp_path / "expect.json")
    env = dict(os.environ, SGFHE_EXPECT_RECORD="1", SGFHE_EXPECT_PATH=scratch)
    # the small synthetic rings (every pass structure below the full-size rings), Params(128) and the latency-form
    # cases of Params(64) and the m = 256 ring: about 70 tags in a few seconds of oracle time
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        os.path.join(ROOT, "tests", "test_gpu_round4.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "small_synthetic or (params128_256 and 128) or (latency_form_calls and (params64 or synthetic))"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    again = json.load(open(scratch))
    assert len(again) >= 60
    for tag, d in again.items():
        assert committed.get(tag) == d, "tests/golden/gpu_expect.json is stale for %s: re-record it " \
                                        "(SGFHE_EXPECT_RECORD=1 python -m pytest tests -m gpu)" % tag
