"""The exact-integer CRT at the bound it is sized for.  The engine recovers
D = (x^j - 1) sum_row u_row (*) C_row from residues modulo a few 29-bit primes whose product M only
just covers it (Params(1024): |D| <= 2 m B Q = 0.33 M, five primes, 0.29 bits of head-room over the
5 m B Q the ctx asks for).  Random inputs stay far from that bound, so these cases drive one
k-loop iteration (sgfhe_debug_cmux: k_flatten_canon -> k_extprod -> the k-loop's own CRT kernel,
k_crt_lean wherever the parameter set admits it) with digits at
+-B/2, key residues at +-Q/2 and j = m (x^m - 1 = -2) and compare with the big-integer oracle's
external_product(a, b, (x^j - 1) C .+ G) (src/fhe.jl:519-530,580).
Run on the GPU box with `pytest -m gpu`."""

import numpy as np
import pytest

import bigint_oracle as BO

pytestmark = pytest.mark.gpu


def _u128(vals):
    out = np.zeros((len(vals), 2), dtype=np.uint64)
    out[:, 0] = [v & 0xFFFFFFFFFFFFFFFF for v in vals]
    out[:, 1] = [v >> 64 for v in vals]
    return out


def _ints(arr):
    flat = np.ascontiguousarray(arr).reshape(-1, 2)
    return [int(lo) | (int(hi) << 64) for lo, hi in flat]


def _params(S, name):
    import bench
    if name == "rns2":
        return bench.make_params(S, "rns2")
    if name == "synth64":
        return bench.make_params(S, "synth64")
    return S.Params(int(name))


def _extreme_acc(p, sign):
    """Accumulator value whose deterministic flatten digits are as large (sign > 0) or as small
    (sign < 0) as a residue of Z_Q allows: x' = acc + off at the top / bottom of [0, Q)."""
    s = p.B // 2 - 1 if p.B % 2 == 0 else (p.B - 1) // 2
    off = (1 + p.B) * s % p.Q
    if sign > 0:
        hi_max = (p.Q - 1) // p.B
        xp = (hi_max - 1) * p.B + (p.B - 1)           # digits (B - 1 - s, hi_max - 1 - s)
    else:
        xp = 0                                         # digits (-s, -s)
    return (xp - off) % p.Q


@pytest.mark.parametrize("name", ["1024", "rns2", "512", "64", "2048", "synth64"])
def test_one_iteration_at_the_exactness_bound(S, name):
    p = _params(S, name)
    m, Q, B = p.m, p.Q, p.B
    bp = BO.Params.custom(p.n, Q, B, DQ_tilde=p.DQ_tilde)
    eng = S.Engine(p)
    G = BO.gadget_matrix(bp)
    kpos, kneg = (Q - 1) // 2, (Q + 1) // 2            # centred +(Q-1)/2 and -(Q-1)/2
    top, bot = _extreme_acc(p, +1), _extreme_acc(p, -1)
    d = BO.flatten(top, B, 2, Q)
    assert 0.49 * B < d[0] < Q // 2                                          # low digit at +B/2 ...
    assert name == "synth64" or 0.49 * B < d[1] < Q // 2                     # ... and the high one (B^2 ~ Q)
    cases = [
        ("all +, j = m", [top] * m, [top] * m, [[[kpos] * m] * 2] * 4, m),
        ("all -, j = m", [bot] * m, [bot] * m, [[[kpos] * m] * 2] * 4, m),
        ("mixed signs, j = m", [top] * m, [bot] * m, [[[kpos] * m, [kneg] * m]] * 4, m),
        ("alternating coefficients, j = 1",
         [top if i & 1 else bot for i in range(m)], [bot if i & 1 else top for i in range(m)],
         [[[kpos if i & 1 else kneg for i in range(m)], [kneg if i & 1 else kpos for i in range(m)]]] * 4, 1),
        ("all +, j = 2m - 1", [top] * m, [top] * m, [[[kpos] * m] * 2] * 4, 2 * m - 1),
        ("j = 0 is the identity", [top] * m, [bot] * m, [[[kpos] * m] * 2] * 4, 0),
    ]
    for label, a, b, C, j in cases:
        A = []
        for row in range(4):
            Arow = []
            for col in range(2):
                x = BO.mul_by_xj_minus_one(C[row][col], j, Q)                # src/fhe.jl:554-556
                x[0] = (x[0] + G[row][col]) % Q                              # `.+ G`, fhe.jl:580
                Arow.append(x)
            A.append(Arow)
        want_a, want_b = BO.external_product(a, b, A, B, 2, Q)               # src/fhe.jl:581
        Cw = np.stack([np.stack([_u128(C[row][col]) for col in range(2)]) for row in range(4)])
        ra, rb = eng.debug_cmux(_u128(a), _u128(b), Cw, j)
        assert _ints(ra) == want_a, "%s: %s, column a" % (name, label)
        assert _ints(rb) == want_b, "%s: %s, column b" % (name, label)
        if j == 0:
            assert want_a == a and want_b == b
    with pytest.raises(S.SgfheError):
        eng.debug_cmux(_u128([0] * m), _u128([0] * m), np.zeros((4, 2, m, 2), dtype=np.uint64), 2 * m)
    eng.close()
