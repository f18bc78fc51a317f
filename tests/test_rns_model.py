"""Host model of the device algorithm (tests/rns_model.py) against the oracle: validates the RNS
restructuring (constants, CRT, digit handling) and the pass / swizzle / twiddle indexing of the
LDS NTT without a GPU."""

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
    x = N.forward(N.to_regs(poly), P["twf"], p, P["ninv"])
    assert int(x.max()) < 4 * p
    got = [int(v) % p for v in x.reshape(-1)]
    Rinv = pow(1 << 32, p - 2, p)
    plain = np.array([int(w) * Rinv % p for w in P["twf"]], dtype=np.uint64)
    assert got == _textbook_fwd(poly, plain, p)
    if logm <= 7:
        assert got == RM.ntt_reference([int(v) for v in poly], P["psi"], p)
    back = N.from_regs(N.inverse(np.array(got, dtype=np.uint64).reshape(N.T, N.E), P["twi"], p, P["ninv"]))
    minv = pow(m, p - 2, p)
    assert [int(v) * minv % p for v in back] == [int(v) for v in poly]


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


def test_exactness_bound_reference_params():
    """8 m B Q < product of the RNS primes for every reference parameter set and the synthetic
    configurations of BASELINE.json."""
    for n in (64, 512, 1024):
        p = BO.Params.make(n)
        RM.Consts(p.n, p.m, p.Q, p.B, p.DQ_tilde)
