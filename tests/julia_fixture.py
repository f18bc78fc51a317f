"""Reader (and, for exercising the reader, a writer) of the fixtures that
sgfhe.jl_amd/julia/make_fixtures.jl produces when a maintainer runs it under Julia with the
reference installed: tests/golden/julia_p<n>.json + julia_p<n>_key.bin.  They are the only way the
oracle's parity status ("parity unpinned") can ever become "pinned": the
reference holds no numeric fixture and cannot run in the build container."""

import hashlib
import json
import os

import numpy as np


def load(directory, n):
    """(fixture dict, key [n][4][2][m][2] uint64) or None when the files are not there."""
    path = os.path.join(directory, "julia_p%d.json" % n)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    p = d["params"]
    key = np.fromfile(os.path.join(directory, d["key_file"]), dtype="<u8")
    assert key.size == p["n"] * 8 * p["m"] * 2, "key file: n * 4 * 2 * m residues of two words"
    assert hashlib.sha256(key.tobytes()).hexdigest() == d["key_sha256"]
    return d, key.reshape(p["n"], 4, 2, p["m"], 2)


def inputs(d):
    a1 = np.array([c["lwe1"]["a"] for c in d["cases"]], dtype=np.uint64)
    a2 = np.array([c["lwe2"]["a"] for c in d["cases"]], dtype=np.uint64)
    b1 = np.array([c["lwe1"]["b"] for c in d["cases"]], dtype=np.uint64)
    b2 = np.array([c["lwe2"]["b"] for c in d["cases"]], dtype=np.uint64)
    return a1, b1, a2, b2


def expected(d):
    """(ModRed words [cases][3][n + 1] uint64, raw residues [cases][3][n + 1] as Python ints)."""
    out = np.array([c["out"] for c in d["cases"]], dtype=np.uint64)
    raw = [[[int(x) for x in g] for g in c["raw"]] for c in d["cases"]]
    return out, raw


def check(d, out, raw_ints, decrypt):
    """out: [cases][3][n + 1] uint64 of the implementation under test, raw_ints: its un-reduced LWEs
    as nested lists of ints, decrypt(case index, gate) -> bit."""
    want_out, want_raw = expected(d)
    assert np.array_equal(out, want_out), "ModRed words differ from the Julia reference's"
    assert raw_ints == want_raw, "_bootstrap_internal residues differ from the Julia reference's"
    for i, c in enumerate(d["cases"]):
        y1, y2 = c["bits"]
        assert [decrypt(i, g) for g in range(3)] == [y1 & y2, y1 | y2, y1 ^ y2]


def write_like_julia(directory, o, n, sk, bkey, bits, a, b, out, raw_ints):
    """The same two files from the oracle's own results (tests only: exercises load / check)."""
    name = "julia_p%d_key.bin" % n
    np.ascontiguousarray(bkey, dtype="<u8").tofile(os.path.join(directory, name))
    cases = []
    for i in range(len(bits) // 2):
        cases.append({"bits": [int(bits[2 * i]), int(bits[2 * i + 1])],
                      "lwe1": {"a": [int(x) for x in a[2 * i]], "b": int(b[2 * i])},
                      "lwe2": {"a": [int(x) for x in a[2 * i + 1]], "b": int(b[2 * i + 1])},
                      "out": [[int(x) for x in g] for g in out[i]],
                      "raw": [[str(x) for x in g] for g in raw_ints[i]]})
    d = {"generated_by": "tests/julia_fixture.py (oracle, NOT the Julia reference)",
         "params": {"n": o.n, "r": o.r, "m": o.m, "Q": str(o.Q), "B": str(o.B), "DQ_tilde": str(o.DQ_tilde)},
         "sk": [int(x) for x in sk], "key_file": name,
         "key_sha256": hashlib.sha256(np.ascontiguousarray(bkey, dtype="<u8").tobytes()).hexdigest(),
         "cases": cases}
    with open(os.path.join(directory, "julia_p%d.json" % n), "w") as f:
        json.dump(d, f)
