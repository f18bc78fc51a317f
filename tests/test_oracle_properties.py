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


@pytest.mark.parametrize("B", [4, 5])
@pytest.mark.parametrize("ell", [3, 4])
def test_flatten_random_exhaustive(B, ell):
    """test/internals.test.jl:69-112, use_rng = true: for every a in Z_q the randomised flatten
    restores a and every digit lies within (-2B, 2B] (decomposition_limits :48-52), with draws from
    a numpy generator standing for the reference's MersenneTwister."""
    q = B ** ell - 1
    lim_lo, lim_hi = q - 2 * B, 2 * B
    xmax = BO.flatten_xmax(B)
    rng = np.random.default_rng(100 * B + ell)
    seen = set()
    for a in range(q):
        d = BO.flatten_random(lambda i: int(rng.integers(-xmax, xmax + 1)), a, B, ell, q)
        assert sum(x * B ** i for i, x in enumerate(d)) % q == a
        assert all(x <= lim_hi or x >= lim_lo for x in d)
        seen.add(tuple(d))
    assert len(seen) > q // 2                   # it is actually randomised
    for v in (-xmax, xmax):                     # and the extreme draws stay inside the limits
        for a in range(q):
            d = BO.flatten_random(lambda i: v, a, B, ell, q)
            assert sum(x * B ** i for i, x in enumerate(d)) % q == a
            assert all(x <= lim_hi or x >= lim_lo for x in d)


def test_draw_stream_known_answers_and_draw_range():
    """The draw stream's block function against published ChaCha vectors: ChaCha8 and ChaCha20 of
    the all-zero key / counter / nonce (draft-strombergson-chacha-test-vectors, TC1, 256-bit key)
    and the RFC 8439 section 2.3.2 block; the addressing of rnd128 (coefficient x = a quarter of
    block x div 4); the engine's draw mapping stays inside [-xmax, xmax] and reaches both ends."""
    z = bytes(32)
    assert BO.chacha_blocks(z, 0, 0, 0, 0, 8)[0].astype("<u4").tobytes().hex() == (
        "3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e"
        "984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42")
    assert BO.chacha_blocks(z, 0, 0, 0, 0, 20)[0].astype("<u4").tobytes().hex() == (
        "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
        "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    rfc = BO.chacha_blocks(bytes(range(32)), 1, 0x09000000, 0x4a000000, 0, 20)[0]
    assert [int(v) for v in rfc[:4]] == [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3]
    assert int(rfc[15]) == 0x4e3c50a2
    assert BO.RND_ROUNDS == 8
    key = bytes(range(100, 132))
    blk = [int(v) for v in BO.chacha_blocks(key, 0x123, 7, 8, 9, 8)[0]]
    for q in range(4):
        assert BO.rnd128(key, (4 * 0x123 + q, 7, 8, 9)) == blk[4 * q:4 * q + 4]
    assert BO.seed_bytes(5) == (5).to_bytes(32, "little")
    p = BO.Params.custom(8, 17, 5)                                 # tiny base: both ends show up
    g = BO.ChaChaFlatten(p, 12345)
    vals = [g.draws(c, y)(j, i) for c in (0, 1) for y in range(8) for j in range(64) for i in (0, 1)]
    assert min(vals) == -g.xmax and max(vals) == g.xmax and g.xmax == 6


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


def test_c_rns2_limbwise_products_equal_bigint(oc):
    """BASELINE.json config 4 ring, small: Q = B * Bp (rule of src/fhe2.jl:57-58).  The C
    restatement in RNS2Number mode (products per limb, src/rns.jl:51-60; CRT of rns.jl:32-40)
    equals its schoolbook product, the big-integer Kronecker product, and a whole bootstrap agrees
    between the three."""
    n, m = 16, 128
    Bp = BO.find_modulus(2 * m, 1 << 24)
    B = BO.find_modulus(2 * m, Bp + 1)
    Q = B * Bp
    p = BO.Params.custom(n, Q, B)
    o_sb = oc.Oracle.from_params(p)
    o_rns = oc.Oracle.from_params(p, rns2=(B, Bp))
    assert o_rns.uses_rns2 and not o_rns.uses_ntt and not o_sb.uses_rns2
    with pytest.raises(ValueError):
        oc.Oracle.from_params(p, rns2=(B, Bp + 2))                 # m1 * m2 != Q
    g = BO.SplitMix64(3)
    for trial in range(3):
        a = [g.below_wide(Q) for _ in range(m)]
        b = [g.below_wide(Q) for _ in range(m)] if trial else [Q - 1] * m
        want = BO.poly_mul(a, b, Q)
        assert oc.u128_to_ints(o_rns.poly_mul(_u128(a), _u128(b))) == want
        assert oc.u128_to_ints(o_sb.poly_mul(_u128(a), _u128(b), schoolbook=True)) == want
    sk = o_sb.private_key(9)
    bkey = o_sb.bootstrap_key(sk, 10, noise=2)
    assert np.array_equal(bkey, o_rns.bootstrap_key(sk, 10, noise=2))
    bits = np.array([1, 1, 0, 1], dtype=np.uint8)
    a, b = o_sb.lwe_encrypt_bits(sk, bits, 11)
    r1 = o_sb.bootstrap_batch(bkey, a[0::2], b[0::2], a[1::2], b[1::2], raw=True)
    r2 = o_rns.bootstrap_batch(bkey, a[0::2], b[0::2], a[1::2], b[1::2], raw=True)
    assert np.array_equal(r1, r2)
    vals = oc.u128_to_ints(bkey)
    bk = [[[vals[((k * 4 + r) * 2 + c) * m:((k * 4 + r) * 2 + c + 1) * m] for c in range(2)]
           for r in range(4)] for k in range(n)]
    big = BO.bootstrap_internal(p, bk, ([int(x) for x in a[0]], int(b[0])), ([int(x) for x in a[1]], int(b[1])))
    for gate in range(3):
        assert oc.u128_to_ints(r2[0, gate]) == big[gate][0] + [big[gate][1]]
    out = o_rns.bootstrap_batch(bkey, a[0::2], b[0::2], a[1::2], b[1::2])
    assert list(o_rns.lwe_decrypt_bits(sk, out[:, 0, :n], out[:, 0, n])) == [1, 0]


def test_cpu_opt_variant_is_bit_identical(oc):
    """BASELINE.md section 3 `cpu_opt` (NTT-domain key, CMux form: 6 NTTs per iteration) against
    the reference-shaped restatement (24 NTTs): accumulators after 1, 2, n iterations, raw and
    ModRed outputs, at Params(64) and a synthetic ring; refused where Q has no NTT."""
    for o, noise in ((oc.Oracle.make(64), None),
                     (oc.Oracle.from_params(BO.Params.custom(16, BO.find_modulus(256, 1 << 52), 1 << 27)), 2)):
        sk = o.private_key(3)
        bkey = o.bootstrap_key(sk, 4, noise=noise)
        khat = o.key_transform(bkey)
        bits = np.array([0, 1, 1, 1, 1, 0], dtype=np.uint8)
        a, b = o.lwe_encrypt_bits(sk, bits, 5)
        lwe = (a[0::2], b[0::2], a[1::2], b[1::2])
        for it in (1, 2, o.n):
            _, acc = o.bootstrap_batch(bkey, *lwe, n_iters=it, want_acc=True)
            _, acc2 = o.bootstrap_batch(khat, *lwe, n_iters=it, want_acc=True, opt=True)
            assert np.array_equal(acc, acc2)
        assert np.array_equal(o.bootstrap_batch(bkey, *lwe, raw=True),
                              o.bootstrap_batch(khat, *lwe, raw=True, opt=True))
        assert np.array_equal(o.bootstrap_batch(bkey, *lwe), o.bootstrap_batch(khat, *lwe, opt=True))
    comp = oc.Oracle(n=8, r=128, m=64, Q=(1 << 60) - 1, B=1 << 30, DQ_tilde=1 << 57)
    with pytest.raises(RuntimeError):
        comp.key_transform(np.zeros((8, 4, 2, 64, 2), dtype=np.uint64))


def test_cpu_opt_variant_on_the_rns2_ring(oc):
    """The NTT-domain loop over the RNS2Number ring (src/rns.jl; key held limb-wise, digit polynomials
    reduced into each limb, limb products put together by the CRT of rns.jl:32-40 per column) against
    the reference-shaped limb-wise loop and against the schoolbook ring: accumulators, raw and ModRed
    outputs, both flatten modes.  It is what lets the config-4 soak on the GPU box check 32 distinct
    bootstraps in a third of the time."""
    n, m = 16, 128
    p1 = BO.find_modulus(2 * m, 1 << 25)
    p2 = BO.find_modulus(2 * m, p1 + 1)
    Q, B = p1 * p2, p2
    o = oc.Oracle(n, 16 * n, m, Q, B, Q // 8, rns2=(p2, p1))
    plain = oc.Oracle(n, 16 * n, m, Q, B, Q // 8)             # composite Q without limbs: schoolbook products
    assert o.uses_rns2 and not o.uses_ntt and not plain.uses_rns2
    sk = o.private_key(1)
    bkey = o.bootstrap_key(sk, 2, noise=2)
    khat = o.key_transform(bkey)
    assert khat.shape == (2,) + bkey.shape
    bits = np.array([1, 0, 1, 1, 0, 0], dtype=np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 3)
    lwe = (a[0::2], b[0::2], a[1::2], b[1::2])
    for rnd in (None, (5, 1, 7)):
        for it in (1, 2, n):
            _, acc = o.bootstrap_batch(bkey, *lwe, n_iters=it, want_acc=True, rnd=rnd)
            _, acc2 = o.bootstrap_batch(khat, *lwe, n_iters=it, want_acc=True, opt=True, rnd=rnd)
            _, acc3 = plain.bootstrap_batch(bkey, *lwe, n_iters=it, want_acc=True, rnd=rnd)
            assert np.array_equal(acc, acc2) and np.array_equal(acc, acc3)
        assert np.array_equal(o.bootstrap_batch(bkey, *lwe, raw=True, rnd=rnd),
                              o.bootstrap_batch(khat, *lwe, raw=True, opt=True, rnd=rnd))
        out = o.bootstrap_batch(khat, *lwe, opt=True, rnd=rnd)
        assert np.array_equal(out, o.bootstrap_batch(bkey, *lwe, rnd=rnd))
        y1, y2 = bits[0::2], bits[1::2]
        for g3, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
            assert np.array_equal(o.lwe_decrypt_bits(sk, out[:, g3, :n], out[:, g3, n]), fn(y1, y2))
    with pytest.raises(RuntimeError):
        plain.key_transform(bkey)                              # no NTT and no limbs: refused


def test_chacha20_key_stream_vectors(oc):
    """The bootstrap-key generator: RFC 8439 section 2.3.2 block test vector, and the C / Python
    generators produce the same key from an int seed and from 32 explicit bytes."""
    key = bytes(range(32))
    blk = BO.chacha20_blocks(key, (0x09000000, 0x4a000000, 0), 2)[1]
    assert [int(v) for v in blk] == [
        0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
        0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    p = BO.Params.custom(8, BO.find_modulus(128, 1 << 50), 1 << 26)
    o = oc.Oracle.from_params(p)
    sk = o.private_key(5)
    k_int = o.bootstrap_key(sk, 6, noise=2)
    k_bytes = o.bootstrap_key(sk, (6).to_bytes(32, "little"), noise=2)
    assert np.array_equal(k_int, k_bytes)
    assert not np.array_equal(k_int, o.bootstrap_key(sk, 7, noise=2))
    flat = [c for k in BO.bootstrap_key(p, [int(x) for x in sk], 6, noise=2) for row in k for col in row for c in col]
    assert oc.u128_to_ints(k_int) == flat
    with pytest.raises(ValueError):
        o.bootstrap_key(sk, b"short")


# ---- test/api.test.jl:45-83 / :86-108 with use_rng = true, on the engine's ChaCha8 stream -----------

def test_randomised_bootstrap_and_pack_decrypt():
    """bootstrap(bkey, rng, ...) and pack_encrypted_bits(bkey, rng, ...) with the randomised
    flatten (src/utils.jl:198-241): truth table and packing round trip at Params(64) noise
    levels on a small ring; different from the deterministic result, reproducible per seed."""
    n, m = 8, 64
    Q = BO.find_modulus(2 * m, 1 << 50)
    p = BO.Params.custom(n, Q, 1 << 26)
    sk = BO.private_key(p, 7)
    bk = BO.bootstrap_key(p, sk, 8, noise=2)
    g = BO.SplitMix64(9)
    for y1 in (0, 1):
        for y2 in (0, 1):
            l1, l2 = BO.lwe_encrypt_bit(p, sk, y1, g), BO.lwe_encrypt_bit(p, sk, y2, g)
            det = BO.bootstrap(p, bk, l1, l2)
            rnd = BO.bootstrap(p, bk, l1, l2, rng=BO.ChaChaFlatten(p, 5, boot=3, call=1))
            assert rnd != det
            assert rnd == BO.bootstrap(p, bk, l1, l2, rng=BO.ChaChaFlatten(p, 5, boot=3, call=1))
            assert rnd != BO.bootstrap(p, bk, l1, l2, rng=BO.ChaChaFlatten(p, 5, boot=4, call=1))
            assert [BO.lwe_decrypt_bit(p, sk, o) for o in rnd] == [y1 & y2, y1 | y2, y1 ^ y2]
    bits = [1, 0, 0, 1, 1, 1, 0, 1]
    lwes = [BO.lwe_encrypt_bit(p, sk, b, g) for b in bits]
    w0, v0 = BO.pack_encrypted_bits(p, bk, lwes)
    w1, v1 = BO.pack_encrypted_bits(p, bk, lwes, seed=77)
    assert (w1, v1) != (w0, v0)
    assert BO.decrypt_ciphertext(p, sk, w1, v1) == bits
    assert [BO.lwe_decrypt_bit(p, sk, l) for l in BO.split_ciphertext(p, w1, v1)] == bits


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


def test_pack_c_randomised_and_ntt_domain_equal_bigint(oc):
    """Round 5: the C restatement's pack_encrypted_bits(bkey, rng, ...) (src/fhe.jl:660-696 with the
    randomised flatten of src/utils.jl:198-241 in the n bootstraps, fhe.jl:673, and in the flatten of every
    as_i, fhe.jl:683-684) on the engine's ChaCha8 stream equals the big-integer restatement for two
    ciphertext positions of a call; and its NTT-domain form (khat) gives the bytes of the reference-shaped
    one in both modes.  This is what makes full-size packing fixtures affordable (tests/golden/pack512.json)."""
    n, m = 8, 64
    Q = BO.find_modulus(2 * m, 1 << 50)
    p = BO.Params.custom(n, Q, 1 << 26)
    o = oc.Oracle.from_params(p)
    sk = o.private_key(7)
    bkey = o.bootstrap_key(sk, 8, noise=2)
    khat = o.key_transform(bkey)
    bits = np.array([1, 0, 0, 1, 1, 1, 0, 1], dtype=np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 9)
    vals = oc.u128_to_ints(bkey)
    bk = [[[vals[((k * 4 + r) * 2 + c) * m:((k * 4 + r) * 2 + c + 1) * m] for c in range(2)]
           for r in range(4)] for k in range(n)]
    lwes = [([int(x) for x in a[i]], int(b[i])) for i in range(n)]
    det = o.pack_encrypted_bits(bkey, a, b)
    det_opt = o.pack_encrypted_bits(bkey, a, b, khat=khat)
    assert np.array_equal(det[0], det_opt[0]) and np.array_equal(det[1], det_opt[1])
    key32 = bytes(range(3, 35))
    for ct, call in ((0, 0), (2, 5)):
        pw, pv = BO.pack_encrypted_bits(p, bk, lwes, seed=key32, ct=ct, call=call)
        for kh in (None, khat):
            w, v = o.pack_encrypted_bits(bkey, a, b, khat=kh, rnd=(key32, ct, call))
            assert [int(x) for x in w] == pw and [int(x) for x in v] == pv
        assert not np.array_equal(w, det[0])
        assert BO.decrypt_ciphertext(p, [int(x) for x in sk], pw, pv) == [int(x) for x in bits]


# ---- the C restatement of the randomised flatten (round 4) against the big-integer one --------------

@pytest.mark.parametrize("B", [4, 5, 6, 7])
def test_flatten_random_c_matches_python_exhaustive(oc, B):
    """flatten(rng, a, Val(B), Val(2)) (src/utils.jl:198-241): the C restatement equals the literal
    big-integer one for every a in Z_q and every pair of draws in [-xmax, xmax]^2, restores a and
    keeps every digit inside (-2B, 2B] (test/internals.test.jl:48-52,69-112 with use_rng = true)."""
    q = B * B - 1
    if q % 2 == 0:
        q -= 1               # Montgomery-based C restatement: odd modulus
    o = oc.Oracle(n=8, r=128, m=64, Q=q, B=B, DQ_tilde=q // 8)
    xmax = BO.flatten_xmax(B)
    for a in range(q):
        for x0 in range(-xmax, xmax + 1):
            for x1 in (-xmax, -1, 0, 2, xmax):
                d = o.flatten_random(a, x0, x1)
                assert d == BO.flatten_random(lambda i: (x0, x1)[i], a, B, 2, q)
                assert (d[0] + d[1] * B) % q == a
                if q > 4 * B:            # the limits are only distinguishable when they do not wrap
                    assert all(x <= 2 * B or x >= q - 2 * B for x in d)


def test_flatten_draws_c_equal_python_stream(oc):
    """The C oracle's ChaCha8 draw stream is the big-integer oracle's (and hence the engine's):
    every draw of several polynomials, both accumulators, several tags / bootstraps / calls, for a
    64-bit and a 87-bit ring."""
    for p in (BO.Params.make(64), BO.Params.custom(16, BO.find_modulus(256, 1 << 86), 35 * (1 << 38))):
        o = oc.Oracle.from_params(p)
        for seed in (5, bytes(range(7, 39))):
            for c, y, boot, call in ((0, 0, 0, 0), (1, 3, 2, 1), (1, (1 << 31) | 5, 4097, 70000)):
                g = BO.ChaChaFlatten(p, seed, boot, call)
                f = g.draws(c, y)
                d = o.flatten_draws(seed, c, y, boot, call)
                assert [[f(j, 0), f(j, 1)] for j in range(p.m)] == d.tolist()
                assert abs(d).max() <= g.xmax


@pytest.mark.parametrize("ring", ["params64", "synthetic", "wide base"])
def test_c_randomised_bootstrap_equals_bigint(oc, ring):
    """bootstrap(bkey, rng, ...) through the C restatement (reference-shaped and in the GPU path's
    algebra) against the literal big-integer restatement on the same stream: accumulators after 1,
    2 and n iterations, raw and ModRed outputs, rows of a batch drawing as bootstraps boot0 + t."""
    if ring == "params64":
        p, noise, its = BO.Params.make(64), None, (1, 2)
    elif ring == "synthetic":
        p, noise, its = BO.Params.custom(16, BO.find_modulus(256, 1 << 52), 1 << 27), 2, (1, 2, 16)
    else:       # B = 3 * 2^45 + 1 (odd, >= 2^46), Q just under B^2: the ring of test_gpu_random's wide case
        p, noise, its = BO.Params.custom(8, BO.find_modulus(128, 1 << 92), 3 * (1 << 45) + 1), 2, (1, 2, 8)
    o = oc.Oracle.from_params(p)
    sk = o.private_key(300)
    bkey = o.bootstrap_key(sk, 301, noise=noise)
    khat = o.key_transform(bkey)
    bits = np.array([1, 1, 0, 1], dtype=np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, 302)
    lwe = (a[0::2], b[0::2], a[1::2], b[1::2])
    vals = oc.u128_to_ints(bkey)
    m, n = p.m, p.n
    bk = [[[vals[((k * 4 + r) * 2 + c) * m:((k * 4 + r) * 2 + c + 1) * m] for c in range(2)]
           for r in range(4)] for k in range(n)]
    seed, call, boot0 = 0x0123456789ABCDEF, 3, 5
    want_acc = {}
    want_raw = []
    for t in range(2):
        acc = {}

        def trace(k, aa, bb, acc=acc):
            if k + 1 in its:
                acc[k + 1] = list(aa) + list(bb)
        raw = BO.bootstrap_internal(p, bk, ([int(x) for x in lwe[0][t]], int(lwe[1][t])),
                                    ([int(x) for x in lwe[2][t]], int(lwe[3][t])), trace=trace,
                                    rng=BO.ChaChaFlatten(p, seed, boot0 + t, call))
        want_acc[t] = acc
        want_raw.append([x for g3 in raw for x in g3[0] + [g3[1]]])
    for opt, key in ((False, bkey), (True, khat)):
        for it in its:
            _, acc = o.bootstrap_batch(key, *lwe, n_iters=it, want_acc=True, opt=opt, rnd=(seed, call, boot0))
            for t in range(2):
                assert oc.u128_to_ints(acc[t]) == want_acc[t][it], (opt, it, t)
        raw = o.bootstrap_batch(key, *lwe, raw=True, opt=opt, rnd=(seed, call, boot0))
        for t in range(2):
            assert oc.u128_to_ints(raw[t]) == want_raw[t]
        out = o.bootstrap_batch(key, *lwe, opt=opt, rnd=(seed, call, boot0))
        for t in range(2):
            assert [int(v) for v in out[t].reshape(-1)] == [BO.reduce_modulus(p.r, x, p.Q) for x in want_raw[t]]
        assert not np.array_equal(out, o.bootstrap_batch(key, *lwe, opt=opt))          # not the deterministic result
        assert not np.array_equal(out, o.bootstrap_batch(key, *lwe, opt=opt, rnd=(seed, call + 1, boot0)))
    y1, y2 = bits[0::2], bits[1::2]
    for g3, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
        assert np.array_equal(o.lwe_decrypt_bits(sk, out[:, g3, :n], out[:, g3, n]), fn(y1, y2))
