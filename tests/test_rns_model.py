"""Host model of the device algorithm (tests/rns_model.py) against the oracle: validates the RNS
restructuring (constants, CRT, digit handling) and the pass / swizzle / twiddle indexing of the
LDS NTT without a GPU."""

import math

import numpy as np
import pytest

import bigint_oracle as BO
import rns_model as RM


def _textbook_fwd(a, tw, p):
    a = [int(v) for v in a]
    m = len(a)
    t, mm = m, 1
    while mm < m:
        t >>= 1
        for i in range(mm):
            W = int(tw[mm + i])
            for j in range(2 * i * t, 2 * i * t + t):
                U, V = a[j], a[j + t] * W % p
                a[j], a[j + t] = (U + V) % p, (U - V) % p
        mm <<= 1
    return a


@pytest.mark.parametrize("loge", [3, 4])
@pytest.mark.parametrize("logm", [6, 7, 8, 9, 11, 13])
def test_ntt_model(logm, loge):
    m = 1 << logm
    C = RM.Consts(m // 8, m, (1 << 50) + 1, 1 << 26, 12345)
    N = RM.NttModel(logm, loge)
    P = C.pk[logm % C.npr]
    p = P["p"]
    poly = np.random.default_rng(logm).integers(0, p, size=m, dtype=np.uint64)
    x = N.forward(N.to_regs(poly), P["twf"], P)               # i32() inside checks the int32 range
    assert int(np.abs(x).max()) < 3.95 * 2 ** 29
    got = [int(v) % p for v in x.reshape(-1)]
    Rinv = pow(1 << 32, p - 2, p)
    plain = np.array([int(w) * Rinv % p for w in P["twf"]], dtype=np.uint64)
    assert got == _textbook_fwd(poly, plain, p)
    if logm <= 7:
        assert got == RM.ntt_reference([int(v) for v in poly], P["psi"], p)
    slots = RM.sred(np.array(got, dtype=np.int64).reshape(N.T, N.E), P)
    back = N.from_regs(N.inverse(slots, P["twi"], P))
    assert int(np.abs(back).max()) < 1.4 * 2 ** 29
    minv = pow(m, p - 2, p)
    assert [int(v) * minv % p for v in back] == [int(v) for v in poly]


@pytest.mark.parametrize("loge", [3, 4])
@pytest.mark.parametrize("logm", [6, 7, 8, 9, 10, 11, 12, 13, 14])
def test_worst_case_ranges(logm, loge):
    """Interval analysis of the signed lazy arithmetic with the exact pass structure: no int32
    overflow and every range reduction inside its precondition, for every RNS prime, from the
    largest inputs the kernels feed (forward: digits / key residues <= 1.01 * 2^29; inverse:
    <= 0.75 * 2^29)."""
    for p in RM.rns_primes():
        R = RM.RangeModel(logm, loge, p)
        assert R.forward(1.01) < 3.95        # pointwise products assume |U| < 3.95 * 2^29: four products
                                             # of column 0 stay below 2^61, column 1 sums to 2.99 * 2^29
        assert R.inverse(0.75) < 1.4         # the rotate-and-subtract epilogue assumes < 1.4 * 2^29
        if R.N.wide_ok():                    # k_extprod's column 0: no input reduction, 16 points per thread
            assert R.inverse(1.5, wide=True) < 1.4
        assert R.peak < 3.99


def test_adversarial_ranges_on_data():
    """Extreme inputs (all +-max) through the data model: the int32 checks inside must hold."""
    for logm, loge in ((13, 4), (13, 3), (12, 4), (9, 3), (14, 4)):
        m = 1 << logm
        C = RM.Consts(m // 8, m, (1 << 50) + 1, 1 << 26, 12345)
        N = RM.NttModel(logm, loge)
        P = C.pk[0]
        p = P["p"]
        big = p + (1 << 16)
        for pattern in (np.full(m, big), np.full(m, -big),
                        np.where(np.arange(m) & 1, big, -big),
                        np.where(np.arange(m) < m // 2, big, -big)):
            x = N.forward(N.to_regs(pattern.astype(np.int64)), P["twf"], P)
            z = RM.sred(RM.sred_floor(x, P), P)
            N.inverse(z, P["twi"], P)
            lim = int(0.75 * 2 ** 29)
            for pat2 in (np.full(m, lim), np.where(np.arange(m) & 1, lim, -lim)):
                N.inverse(pat2.astype(np.int64).reshape(N.T, N.E), P["twi"], P)


def test_swizzle_is_conflict_free():
    """Every b32 LDS access of every pass hits 32 distinct banks per 32-lane group."""
    for logm, loge in ((9, 3), (12, 3), (13, 3), (10, 4), (12, 4), (13, 4)):
        N = RM.NttModel(logm, loge)
        S_list = {N.STOP} | set(range(0, N.STOP + 1, loge))
        for S in S_list:
            for e in range(N.E):
                addr = N.lds_addr(S, e)
                assert len(set(addr.tolist())) == N.T
                for g0 in range(0, N.T, 32):
                    banks = addr[g0:g0 + 32] & 31
                    assert len(set(banks.tolist())) == min(32, N.T)


def test_pipeline_matches_oracle():
    """init -> n x (k_extprod, k_crt_acc) in the model == oracle accumulators after every k."""
    n, m = 8, 64
    Q = BO.find_modulus(2 * m, 1 << 50)
    B = 1 << 26
    p = BO.Params.custom(n, Q, B)
    sk = BO.private_key(p, 5)
    bk = BO.bootstrap_key(p, sk, 6, noise=2)
    E = RM.EngineModel(n, m, Q, B, p.DQ_tilde)
    key = [[[E.key_transform(bk[k][rc // 2][rc % 2], pi) for rc in range(8)]
            for pi in range(E.C.npr)] for k in range(n)]
    g = BO.SplitMix64(7)
    l1 = BO.lwe_encrypt_bit(p, sk, 1, g)
    l2 = BO.lwe_encrypt_bit(p, sk, 0, g)
    trace = []
    BO.bootstrap_internal(p, bk, l1, l2, trace=lambda k, a, b: trace.append((list(a), list(b))))
    ua = [(x + y) % p.r for x, y in zip(l1[0], l2[0])]
    ub = (l1[1] + l2[1]) % p.r
    b0 = [(c * p.DQ_tilde) % Q for c in BO.mul_by_monomial(BO.initial_poly(p), -ub, Q)]
    dig_a = [E.C.digits_of(0)] * m
    dig_b = [E.C.digits_of(v) for v in b0]
    for k in range(n):
        ys = E.extprod(dig_a, dig_b, key[k], ua[k])
        dig_a = E.crt_acc(ys[0], dig_a)
        dig_b = E.crt_acc(ys[1], dig_b)
        assert E.acc_from_digits(dig_a) == trace[k][0]
        assert E.acc_from_digits(dig_b) == trace[k][1]


def test_pipeline_matches_oracle_randomised():
    """The same with the randomised flatten: the model's ChaCha8 digits (device counter layout,
    shifted representation e = u + s + xmax) against the oracle's literal restatement of
    src/utils.jl:198-241 on the same stream -- accumulators after every k, and the stored digits
    equal the oracle's flatten output shifted, inside (-2B, 2B]."""
    n, m = 8, 64
    Q = BO.find_modulus(2 * m, 1 << 50)
    B = 1 << 26
    p = BO.Params.custom(n, Q, B)
    sk = BO.private_key(p, 5)
    bk = BO.bootstrap_key(p, sk, 6, noise=2)
    E = RM.EngineModel(n, m, Q, B, p.DQ_tilde, random_flatten=True)
    C = E.C
    key = [[[E.key_transform(bk[k][rc // 2][rc % 2], pi) for rc in range(8)]
            for pi in range(C.npr)] for k in range(n)]
    g = BO.SplitMix64(7)
    l1 = BO.lwe_encrypt_bit(p, sk, 1, g)
    l2 = BO.lwe_encrypt_bit(p, sk, 0, g)
    seed, boot, call = 0x1234567890ABCDEF, 11, 2
    rng = BO.ChaChaFlatten(p, seed, boot, call)
    trace = []
    BO.bootstrap_internal(p, bk, l1, l2, trace=lambda k, a, b: trace.append((list(a), list(b))), rng=rng)
    ua = [(x + y) % p.r for x, y in zip(l1[0], l2[0])]
    ub = (l1[1] + l2[1]) % p.r
    b0 = [(c * p.DQ_tilde) % Q for c in BO.mul_by_monomial(BO.initial_poly(p), -ub, Q)]
    # k_init in the randomised mode: digits of (acc + off_rnd) with the draws tagged y = 0
    dig_a = [E.random_digits(C.off_rnd % Q, seed, (i, 0, boot, call)) for i in range(m)]
    dig_b = [E.random_digits((v + C.off_rnd) % Q, seed, (m + i, 0, boot, call)) for i, v in enumerate(b0)]
    shift = C.s + C.xmax
    acc_a, acc_b = [0] * m, b0
    for k in range(n):
        # the stored digits are the oracle's randomised flatten of the current accumulators
        for c, (dig, acc) in enumerate(((dig_a, acc_a), (dig_b, acc_b))):
            ref = BO.flatten_poly(acc, B, 2, Q, rng.draws(c, k))
            for i in range(m):
                u = [dig[i][0] - shift, dig[i][1] - shift]
                assert all(-2 * B < x <= 2 * B for x in u)
                assert [x % Q for x in u] == [ref[0][i], ref[1][i]]
        ys = E.extprod(dig_a, dig_b, key[k], ua[k], random=True)
        dig_a = E.crt_acc(ys[0], dig_a, rnd=(seed, 0, k + 1, boot, call))
        dig_b = E.crt_acc(ys[1], dig_b, rnd=(seed, 1, k + 1, boot, call))
        acc_a, acc_b = E.acc_from_digits(dig_a, random=True), E.acc_from_digits(dig_b, random=True)
        assert acc_a == trace[k][0]
        assert acc_b == trace[k][1]


def test_model_at_the_exactness_bound():
    """Digits at +-B/2, key residues at +-Q/2, j = m (x^m - 1 = -2) on a ring whose 5 m B Q sits
    just under the product of three primes: |D| reaches 0.39 M, the edge of what crt_value's float
    estimate of alpha is sized for.  (tests/test_gpu_worstcase.py does the same on the device at
    the real parameter sets.)"""
    import math
    n, m, B = 8, 64, 1 << 27
    primes = RM.rns_primes()[:3]
    have = sum(math.log2(p) for p in primes)
    Q = BO.find_modulus(2 * m, int(2 ** (have - math.log2(5.0) - 6 - 27 - 0.02)))
    while math.log2(5.0) + 6 + 27 + math.log2(Q) + 0.001 >= have:
        Q = BO.find_modulus(2 * m, Q // 2)
    E = RM.EngineModel(n, m, Q, B, Q // 8)
    C = E.C
    assert C.npr == 3 and 2 * m * B * Q / C.Mrns > 0.36
    s_ = C.s
    top = ((((Q - 1) // B) - 1) * B + (B - 1) - C.off) % Q
    bot = (0 - C.off) % Q
    kpos, kneg = (Q - 1) // 2, (Q + 1) // 2
    G = [[1, 0], [B, 0], [0, 1], [0, B]]
    for a, b, Ck, j in (([top] * m, [top] * m, [[[kpos] * m] * 2] * 4, m),
                        ([bot] * m, [top] * m, [[[kpos] * m, [kneg] * m]] * 4, m),
                        ([top if i & 1 else bot for i in range(m)], [bot] * m,
                         [[[kneg if i & 1 else kpos for i in range(m)], [kpos] * m]] * 4, 2 * m - 1)):
        A = []
        for row in range(4):
            Arow = []
            for col in range(2):
                x = BO.mul_by_xj_minus_one(Ck[row][col], j, Q)
                x[0] = (x[0] + G[row][col]) % Q
                Arow.append(x)
            A.append(Arow)
        want = BO.external_product(a, b, A, B, 2, Q)
        key = [[E.key_transform(Ck[rc // 2][rc % 2], pi) for rc in range(8)] for pi in range(C.npr)]
        dig_a, dig_b = [C.digits_of(v) for v in a], [C.digits_of(v) for v in b]
        ys = E.extprod(dig_a, dig_b, key, j)
        got = (E.acc_from_digits(E.crt_acc(ys[0], dig_a)), E.acc_from_digits(E.crt_acc(ys[1], dig_b)))
        assert got[0] == want[0] and got[1] == want[1]


def test_exactness_bound_reference_params():
    """5 m B Q < product of the RNS primes for every reference parameter set (20 m B Q when the
    ctx is created for the randomised flatten): prime counts the engine ends up with."""
    want = {64: (4, 4), 512: (5, 5), 1024: (5, 6), 2048: (6, 6)}
    for n, (det, rnd) in want.items():
        p = BO.Params.make(n)
        assert RM.Consts(p.n, p.m, p.Q, p.B, p.DQ_tilde).npr == det
        assert RM.select_npr(p.m.bit_length() - 1, p.B, p.Q, random_flatten=True) == rnd


def _lean_cases():
    import bench
    import sgfhe_jl_amd as S
    for name in ("params1024", "params512", "params64", "synth64", "rns2", "params2048"):
        p = bench.make_params(S, name)
        yield name, (p.n, p.m, p.Q, p.B, p.DQ_tilde)
    Q = BO.find_modulus(2 * 64, 1 << 50)
    yield "synthetic m = 64", (8, 64, Q, 1 << 26, Q // 8)
    yield "external-product ring", (8, 64, (1 << 60) - 1, 1 << 30, ((1 << 60) - 1) // 8)


@pytest.mark.parametrize("name,args", list(_lean_cases()), ids=[c[0] for c in _lean_cases()])
def test_crt_lean_model(name, args):
    """k_crt_lean's integer algorithm (29-bit limb sums, reciprocal quotient estimates, one
    conditional correction each) on every parameter set of BASELINE.json and of the tests: equal to
    the exact (x_old + D) mod Q and its base-B digits, with every intermediate inside its register
    width and both estimates within one of the exact quotients (asserted inside RM.CrtLean).  The
    residues are built as k_extprod hands them over: non-negative representatives below 5.7 p,
    plus (p - 1) / 2 on the last prime; D runs to the edge of the exactness bound, x_old and
    x_new to the edges of [0, Q) and of the digit boundaries."""
    import random
    rnd = random.Random(hash(name) & 0xFFFF)
    C = RM.Consts(*args)
    L = RM.CrtLean(C)
    assert L.ok
    E = RM.EngineModel.__new__(RM.EngineModel)
    E.C = C
    bound = min(int(0.4 * C.Mrns), 2 * C.M * C.B * C.Q)
    inv = [pow(C.Mrns // p, -1, p) for p in C.primes]
    for it in range(4000):
        D = rnd.randint(-bound, bound)
        if it % 11 == 0:
            D = rnd.choice([bound, -bound, 0, 1, -1])
        y = []
        for i, p in enumerate(C.primes):
            r = D * inv[i] % p
            r += rnd.randint(0, 5 if r < 0.7 * p else 4) * p
            y.append(r + C.pk[i]["hoff"])
            assert r < 5.7 * p
        xo = rnd.randrange(C.Q)
        if it % 7 == 0:
            xo = rnd.choice([0, C.Q - 1, C.B - 1, C.B % C.Q, (C.Q - C.B) % C.Q])
        if it % 13 == 0:        # x_new at 0, Q - 1 and at digit boundaries
            xo = (-D + rnd.choice([0, 1, -1, C.B, C.B - 1])) % C.Q
        lo, hq, alpha = L.digits(y, xo % C.B, xo // C.B)
        xn = E.crt_value(y, xo)                       # the float-alpha / table form of k_crt_acc
        assert alpha == E.last_alpha
        assert xn == (xo + D) % C.Q
        assert (lo, hq) == (xn % C.B, xn // C.B)


@pytest.mark.parametrize("name,args", list(_lean_cases()), ids=[c[0] for c in _lean_cases()])
def test_crt_lean_random_model(name, args):
    """The randomised flatten through k_crt_lean's limb sums (k_crt_lean<.., RND>): the draws enter
    as 2 xmax - r_i added to the old stored digits, a constant c Q - 2 xmax (1 + B) puts the shift
    back, and one quotient step gives x2 = (x_old + D - r_0 - r_1 B) mod Q -- equal to
    random_digits of k_crt_acc (EngineModel.random_digits) for D up to the 4-times-wider exactness
    bound of that mode, old stored digits up to their maxima (B - 1 + 2 xmax, Q / B + 2 xmax) and
    draws at 0, 2 xmax and in between; every intermediate inside its register (RM.CrtLean)."""
    import random
    rnd = random.Random((hash(name) & 0xFFFF) + 1)
    if 2 * math.log2(args[3]) + 2 - math.log2(args[2]) >= 50:
        pytest.skip("the engine refuses the randomised flatten for this base")
    C = RM.Consts(*args, random_flatten=True)
    L = RM.CrtLean(C)
    assert L.ok
    E = RM.EngineModel.__new__(RM.EngineModel)
    E.C = C
    bound = min(int(0.4 * C.Mrns), 8 * C.M * C.B * C.Q)
    inv = [pow(C.Mrns // p, -1, p) for p in C.primes]
    xm2 = 2 * C.xmax
    for it in range(4000):
        D = rnd.randint(-bound, bound)
        if it % 11 == 0:
            D = rnd.choice([bound, -bound, 0, 1, -1])
        y = []
        for i, p in enumerate(C.primes):
            r = D * inv[i] % p
            r += rnd.randint(0, 5 if r < 0.7 * p else 4) * p
            y.append(r + C.pk[i]["hoff"])
        # old stored digits: (lo, hi) of some x2 < Q plus the draws of the previous iteration
        x2o = rnd.randrange(C.Q)
        if it % 7 == 0:
            x2o = rnd.choice([0, C.Q - 1, C.B - 1, C.B % C.Q, (C.Q - C.B) % C.Q])
        ro = [rnd.choice([0, xm2, rnd.randint(0, xm2)]) for _ in range(2)]
        e_lo, e_hi = x2o % C.B + ro[0], x2o // C.B + ro[1]
        r0, r1 = (rnd.choice([0, xm2, rnd.randint(0, xm2)]) for _ in range(2))
        xo = e_hi * C.B + e_lo                       # what k_crt_acc adds (congruent to xn_old)
        xn = E.crt_value(y, xo)
        assert xn == (xo + D) % C.Q
        x2 = (xn - r0 - r1 * C.B) % C.Q
        if it % 13 == 0:                             # drive x2 to 0, Q - 1 and the digit boundaries
            want = rnd.choice([0, C.Q - 1, C.B, C.B - 1])
            shift = (want - x2) % C.Q
            x2o2 = (x2o + shift) % C.Q
            e_lo, e_hi = x2o2 % C.B + ro[0], x2o2 // C.B + ro[1]
            xo = e_hi * C.B + e_lo
            xn = E.crt_value(y, xo)
            x2 = (xn - r0 - r1 * C.B) % C.Q
            assert x2 == want
        lo, hq, alpha = L.digits_random(y, e_lo, e_hi, r0, r1)
        assert (lo, hq) == (x2 % C.B + r0, x2 // C.B + r1)


# ---- the quarter form of the latency kernels (round 4: k_fwd_quarter / k_inv_quarter / k_crt_lean1q) ----

def _quarter_tables(P, logm):
    """engine.hip build_basis: entry mm' + i' of quarter q's tables = entry 4 mm' + q mm' + i' of the big ones."""
    m, ms = 1 << logm, 1 << (logm - 2)
    out = []
    for q in range(4):
        t = {k: np.zeros(ms, dtype=np.int64) for k in ("twf", "twi", "twfp", "twip")}
        for jj in range(1, ms):
            mmq = 1 << (jj.bit_length() - 1)
            J = 4 * mmq + q * mmq + (jj - mmq)
            t["twf"][jj], t["twi"][jj] = P["twf"][J], P["twi"][J]
            if jj >= 2:
                t["twfp"][jj], t["twip"][jj] = P["twfp"][J], P["twip"][J]
        out.append(dict(P, **t))
    return out


@pytest.mark.parametrize("logm", [8, 12, 13])
def test_quarter_form_equals_the_whole_transforms(logm):
    """kernels.h k_fwd_quarter / k_inv_quarter / k_crt_lean1q in the numpy model, every operation checked
    against int32: (a) the radix-4 combination of the four input quarters followed by the quarter-size
    transform on the re-indexed twiddle sub-tree gives the slots of the whole forward transform;
    (b) multiplying slot s by psi^(j (2 brv(s) + 1)) - 1 and running the quarter-size inverse
    transforms, then the last two Gentleman-Sande stages on the four partial values, gives
    x^j P - P of the whole inverse transform's P, for j across the wrap (0, 1, m - 1, m, 2m - 1)."""
    m, ms = 1 << logm, 1 << (logm - 2)
    C = RM.Consts(m // 8, m, (1 << 50) + 1, 1 << 26, 12345)
    P = C.pk[1]
    p = P["p"]
    big = RM.NttModel(logm, 3)
    sub = RM.NttModel(logm - 2, 3)
    QT = _quarter_tables(P, logm)
    rng = np.random.default_rng(logm)
    # (a) forward: inputs as digit_reduce leaves them, |x| <= p + 2^16
    poly = rng.integers(-(p + 65536), p + 65536 + 1, size=m, dtype=np.int64)
    want = big.forward(big.to_regs(RM.sred_floor(poly, P)), P["twf"], P).reshape(-1) % p
    X = poly.reshape(4, ms)                                        # X[t][i] = coefficient i + t m / 4
    f1, f2, f3, fp2, fp3 = (int(P["twf"][1]), int(P["twf"][2]), int(P["twf"][3]), int(P["twfp"][2]), int(P["twfp"][3]))
    u = RM.smont(X[2], f1, P)
    for q in range(4):
        a = RM.i32(X[0] - u) if q & 2 else RM.i32(X[0] + u)
        wB, wP = (f3, fp3) if q & 2 else (f2, fp2)
        w = RM.sredc(RM.i32(X[1]) * wB + RM.i32(X[3]) * wP, P)
        xq = RM.sred_floor(RM.i32(a - w) if q & 1 else RM.i32(a + w), P)
        got = sub.forward(sub.to_regs(xq), QT[q]["twf"], QT[q]).reshape(-1) % p
        assert np.array_equal(got, want[q * ms:(q + 1) * ms]), "forward quarter %d" % q
    # (b) inverse: slot values as the summed partial products leave them, |z| < 2.99 * 2^29
    z = rng.integers(-int(2.9 * 2 ** 29), int(2.9 * 2 ** 29), size=m, dtype=np.int64)
    R1 = (1 << 32) % p
    # (inverse transforms take slots E tid + e per thread: a plain reshape, not to_regs)
    whole = big.from_regs(big.inverse(RM.sred(z, P).reshape(big.T, big.E), P["twi"], P)) % p       # P, unscaled
    v1, v2, v3 = int(P["twi"][1]), int(P["twi"][2]), int(P["twi"][3])
    for j in (0, 1, 5, m - 1, m, m + 3, 2 * m - 1):
        rot = np.zeros(m, dtype=object)
        for i in range(m):
            k = (i + j) % (2 * m)
            rot[k % m] = (-int(whole[i]) if k >= m else int(whole[i])) % p
        want = np.array([(int(a) - int(b)) % p for a, b in zip(rot, whole)], dtype=np.int64)   # x^j P - P
        Y = []
        for q in range(4):
            s = q * ms + np.arange(ms)
            br = np.array([RM.bitrev(int(v), logm) for v in s], dtype=np.int64)
            e = (j * (2 * br + 1)) % (2 * m)
            pw = np.array([RM.centre(pow(P["psi"], int(v), p) * R1, p) for v in e], dtype=np.int64)   # PrimeK::pw
            d = RM.scentre(RM.i32(pw - P["r1"]), P)
            zq = RM.smont(z[q * ms:(q + 1) * ms], d, P)
            assert int(np.abs(zq).max()) < 0.75 * 2 ** 29
            Y.append(sub.from_regs(sub.inverse(zq.reshape(sub.T, sub.E), QT[q]["twi"], QT[q])))
            assert int(np.abs(Y[-1]).max()) < 1.4 * 2 ** 29
        c0, c2 = RM.sred(RM.i32(Y[0] + Y[1]), P), RM.sred(RM.i32(Y[2] + Y[3]), P)
        c1, c3 = RM.smont(RM.i32(Y[0] - Y[1]), v2, P), RM.smont(RM.i32(Y[2] - Y[3]), v3, P)
        got = np.concatenate([RM.i32(c0 + c2), RM.i32(c1 + c3), RM.smont(RM.i32(c0 - c2), v1, P),
                              RM.smont(RM.i32(c1 - c3), v1, P)])
        assert int(np.abs(got).max()) < 1.5 * 2 ** 29                # + 3 p stays a residue below 4.6 p
        assert np.array_equal(got % p, want), "inverse, j = %d" % j


def test_fused_quarter_products_in_the_model():
    """kernels.h k_ext_quarter between its two transforms: the pair products as one Montgomery step of a
    64-bit sum (worst-case slot values 3.95 * 2^29 against key residues of magnitude p / 2), the two groups'
    shares added unreduced, the rotation factor as a Montgomery product -- every step inside int32 / int64,
    the inverse transform's entry bound kept, and the value equal modulo p to the two-launch form's
    (four separate Montgomery products summed, k_fwd_quarter / k_inv_quarter)."""
    C = RM.Consts(64, 512, (1 << 50) + 1, 1 << 26, 12345)
    for P in C.pk:
        p = P["p"]
        rng = np.random.default_rng(p % 1000)
        n = 4096
        umax = int(3.95 * 2 ** 29)
        u = rng.integers(-umax, umax + 1, size=(4, n), dtype=np.int64)
        K = rng.integers(-(p // 2), p // 2 + 1, size=(4, n), dtype=np.int64)
        u[:, :8] = umax * np.array([1, 1, -1, -1, 1, -1, 1, -1])            # corners
        K[:, :8] = (p // 2) * np.array([1, -1, 1, -1, 1, 1, -1, -1])
        d = RM.scentre(RM.i32(rng.integers(-(p // 2), p // 2 + 1, size=n, dtype=np.int64) - P["r1"]), P)
        w = []
        for g in range(2):
            s = u[2 * g] * K[2 * g] + u[2 * g + 1] * K[2 * g + 1]
            assert int(np.abs(s).max()) < 2 ** 62
            w.append(RM.sredc(s, P))
            assert int(np.abs(w[-1]).max()) < 0.99 * 2 ** 29
        z = RM.smont(RM.i32(w[0] + w[1]), d, P)
        assert int(np.abs(z).max()) < 0.75 * 2 ** 29
        two = sum(RM.smont(u[r], K[r], P) for r in range(4))
        assert int(np.abs(two).max()) < 2.99 * 2 ** 29
        assert np.array_equal(z % p, RM.smont(two, d, P) % p)
