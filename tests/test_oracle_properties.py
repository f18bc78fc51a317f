"""The reference's own property tests (test/internals.test.jl, test/api.test.jl, manual doctest),
restated on the oracle: both restatements (oracle/bigint_oracle.py, oracle/sgfhe_oracle.c) must
satisfy them and agree with each other.  No GPU."""

from fractions import Fraction

import numpy as np
import pytest

import bigint_oracle as BO


def _u128(vals):
    import oracle_c
    return oracle_c.ints_to_u128(vals)


# ---- test/internals.test.jl:6-47 -----------------------------------------------------------------

def _rescale_ref(new_max, x, old_max, round_result):
    """rescale_ref of test/internals.test.jl:6-20 in exact rational arithmetic.  Julia's
    round(BigInt, .) breaks ties to even, but x * new_max / old_max is never a tie in the cases
    used here (old_max odd), so round-half-up is the same function."""
    v = Fraction(x * new_max, old_max)
    if round_result:
        assert (2 * v).denominator != 1 or v.denominator == 1
        res = int(v + Fraction(1, 2))
        if res == new_max:
            res = 0
        return res
    return int(v)


@pytest.mark.parametrize("new_max", [16, 17])
@pytest.mark.parametrize("round_result", [False, True])
def test_rescale_exhaustive(oc, new_max, round_result):
    old_max = 2 ** 12 + 1
    for x in range(old_max):
        ref = _rescale_ref(new_max, x, old_max, round_result)
        assert BO.rescale(new_max, x, old_max, round_result) == ref
        assert oc.rescale(new_max, x, old_max, round_result) == ref


def test_rescale_wide_operands(oc):
    """ModRed operands of the hot path: x < Q (87 bits), new_max = r, old_max = Q."""
    p = BO.Params.make(1024)
    g = BO.SplitMix64(5)
    for x in [0, 1, p.Q - 1, p.Q // 2, p.Q // 2 + 1, p.Q // p.r, p.Q - p.Q // (2 * p.r)] + \
            [g.below_wide(p.Q) for _ in range(200)]:
        assert oc.rescale(p.r, x, p.Q, True) == BO.rescale(p.r, x, p.Q, True) == \
            _rescale_ref(p.r, x, p.Q, True)


# ---- test/internals.test.jl:50-112 ---------------------------------------------------------------

def _limits(B, q):
    s = (B - 1) // 2 if B % 2 else B // 2 - 1
    return q - s, B - s - 1


@pytest.mark.parametrize("B", [4, 5])
@pytest.mark.parametrize("ell", [2, 3, 4])
def test_flatten_exhaustive(B, ell):
    q = B ** ell - 1
    lo, hi = _limits(B, q)
    for a in range(q):
        d = BO.flatten(a, B, ell, q)
        assert sum(x * B ** i for i, x in enumerate(d)) % q == a
        assert all(x <= hi or x >= lo for x in d)


@pytest.mark.parametrize("B", [4, 5, 6, 7])
def test_flatten_c_matches_python(oc, B):
    q = B * B - 1
    if q % 2 == 0:
        q -= 1               # the C restatement is Montgomery-based: odd modulus
    o = oc.Oracle(n=8, r=128, m=64, Q=q, B=B, DQ_tilde=q // 8)
    for a in range(q):
        assert o.flatten(a) == BO.flatten(a, B, 2, q)


# ---- test/internals.test.jl:115-141 ---------------------------------------------------------------

def test_flatten_poly_restore_and_range(oc):
    B = 1 << 30
    q = B * B - 1
    o = oc.Oracle(n=8, r=128, m=64, Q=q, B=B, DQ_tilde=q // 8)
    lo, hi = _limits(B, q)
    rng = np.random.default_rng(3)
    for a in rng.integers(0, q, size=64, dtype=np.uint64):
        a = int(a)
        d = o.flatten(a)
        assert d == BO.flatten(a, B, 2, q)
        assert all(x <= hi or x >= lo for x in d)
        assert (d[0] + d[1] * B) % q == a


# ---- test/internals.test.jl:144-166 ---------------------------------------------------------------

def test_external_product_identity(oc):
    m, B = 64, 1 << 30
    q = B * B - 1
    o = oc.Oracle(n=8, r=128, m=m, Q=q, B=B, DQ_tilde=q // 8)
    assert not o.uses_ntt            # composite modulus: the non-NTT exact multiply
    rng = np.random.default_rng(4)
    a = [int(v) for v in rng.integers(0, q, size=m, dtype=np.uint64)]
    b = [int(v) for v in rng.integers(0, q, size=m, dtype=np.uint64)]
    z = [0] * m
    G = [[z[:], z[:]] for _ in range(4)]
    for row, col, g in ((0, 0, 1), (1, 0, B), (2, 1, 1), (3, 1, B)):
        G[row][col][0] = g
    ra, rb = BO.external_product(a, b, G, B, 2, q)
    assert ra == a and rb == b
    Gc = np.stack([np.stack([_u128(G[r][c]) for c in range(2)]) for r in range(4)])
    ca, cb = o.external_product(_u128(a), _u128(b), Gc)
    assert oc.u128_to_ints(ca) == a and oc.u128_to_ints(cb) == b


# ---- polynomial product: NTT vs schoolbook vs Kronecker ---------------------------------------------

def test_poly_mul_three_ways(oc):
    p = BO.Params.make(64)
    o = oc.Oracle.from_params(p)
    assert o.uses_ntt
    g = BO.SplitMix64(8)
    a = [g.below_wide(p.Q) for _ in range(p.m)]
    b = [g.below_wide(p.Q) for _ in range(p.m)]
    ref = BO.poly_mul(a, b, p.Q)
    assert ref[:64] == BO.poly_mul_schoolbook(a, b, p.Q)[:64]
    assert oc.u128_to_ints(o.poly_mul(_u128(a), _u128(b))) == ref
    assert oc.u128_to_ints(o.poly_mul(_u128(a), _u128(b), schoolbook=True)) == ref


def test_mul_by_monomial_negative_powers():
    """docs/src/theory.md:23-32."""
    Q = 97
    a = [1, 2, 3, 4]
    assert BO.mul_by_monomial(a, 1, Q) == [Q - 4, 1, 2, 3]
    assert BO.mul_by_monomial(a, -1, Q) == [2, 3, 4, Q - 1]
    assert BO.mul_by_monomial(a, 4, Q) == [(Q - x) % Q for x in a]
    assert BO.mul_by_monomial(a, 8, Q) == a
    assert BO.mul_by_monomial(BO.mul_by_monomial(a, 3, Q), -3, Q) == a


# ---- test/api.test.jl:45-83 (deterministic branch) ------------------------------------------------

def test_bootstrap_truth_table_params64(oc):
    p = BO.Params.make(64)
    o = oc.Oracle.from_params(p)
    sk = o.private_key(41)
    bkey = o.bootstrap_key(sk, 42)
    bits = np.random.default_rng(43).integers(0, 2, size=32).astype(np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 44)
    assert np.array_equal(o.lwe_decrypt_bits(sk, a, b), bits)
    out = o.bootstrap_batch(bkey, a[0::2], b[0::2], a[1::2], b[1::2])
    y1, y2 = bits[0::2], bits[1::2]
    for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
        assert np.array_equal(o.lwe_decrypt_bits(sk, out[:, g, :p.n], out[:, g, p.n]), fn(y1, y2))
    # docs/src/manual.md:155-172: rng = nothing is deterministic
    assert np.array_equal(out, o.bootstrap_batch(bkey, a[0::2], b[0::2], a[1::2], b[1::2]))


def test_c_equals_bigint_params64(oc):
    p = BO.Params.make(64)
    o = oc.Oracle.from_params(p)
    sk = BO.private_key(p, 1)
    assert [int(x) for x in o.private_key(1)] == sk
    bk = BO.bootstrap_key(p, sk, 2)
    bkc = o.bootstrap_key(o.private_key(1), 2)
    assert oc.u128_to_ints(bkc) == [c for k in bk for row in k for col in row for c in col]
    g = BO.SplitMix64(3)
    l1 = BO.lwe_encrypt_bit(p, sk, 1, g)
    l2 = BO.lwe_encrypt_bit(p, sk, 1, g)
    raw = BO.bootstrap_internal(p, bk, l1, l2)
    rawc = o.bootstrap_batch(bkc, [l1[0]], [l1[1]], [l2[0]], [l2[1]], raw=True)
    for gi in range(3):
        assert oc.u128_to_ints(rawc[0, gi]) == raw[gi][0] + [raw[gi][1]]


# ---- src/rns.jl ---------------------------------------------------------------------------------------

def test_rns2_roundtrip():
    m1 = BO.find_modulus(1 << 14, 1 << 40)
    m2 = BO.find_modulus(1 << 14, m1 + 1)
    g = BO.SplitMix64(9)
    for _ in range(200):
        x = g.below_wide(m1 * m2)
        v1, v2 = BO.rns2_from_int(x, m1, m2)
        assert BO.rns2_to_int(v1, v2, m1, m2) == x


# ---- test/api.test.jl:86-108 (packing, deterministic branch) on a small synthetic ring ------------

def test_pack_small_ring_c_equals_bigint(oc):
    n, m = 8, 64
    Q = BO.find_modulus(2 * m, 1 << 50)
    p = BO.Params.custom(n, Q, 1 << 26)
    o = oc.Oracle.from_params(p)
    sk = o.private_key(7)
    bkey = o.bootstrap_key(sk, 8, noise=2)
    bits = np.array([1, 0, 0, 1, 1, 1, 0, 1], dtype=np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 9)
    w, v = o.pack_encrypted_bits(bkey, a, b)
    skl = [int(x) for x in sk]
    vals = oc.u128_to_ints(bkey)
    bk = [[[vals[((k * 4 + r) * 2 + c) * m:((k * 4 + r) * 2 + c + 1) * m] for c in range(2)]
           for r in range(4)] for k in range(n)]
    lwes = [([int(x) for x in a[i]], int(b[i])) for i in range(n)]
    pw, pv = BO.pack_encrypted_bits(p, bk, lwes)
    assert [int(x) for x in w] == pw and [int(x) for x in v] == pv
    assert BO.decrypt_ciphertext(p, skl, pw, pv) == [int(x) for x in bits]
