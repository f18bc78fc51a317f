import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np
import sgfhe_jl_amd as S, rns_model as RM, bigint_oracle as BO
for logm in (10, 11, 12, 13):
    m = 1 << logm; n = m // 8
    Q = BO.find_modulus(2 * m, 1 << 50); B = 1 << 26
    eng = S.Engine(S.Params.custom(n, Q, B))
    C = RM.Consts(n, m, Q, B, Q // 8); N = RM.NttModel(logm)
    rng = np.random.default_rng(logm)
    for pi in (0, 4):
        p = C.primes[pi]
        poly = rng.integers(0, p, size=m, dtype=np.uint64)
        for rep in range(3):
            fwd = eng.debug_ntt(pi, poly.astype(np.uint32)).astype(np.uint64)
            model = N.forward(N.to_regs(poly), C.pk[pi]["twf"], p).reshape(-1) % p
            bad = np.nonzero(fwd != model)[0]
            print("logm", logm, "pi", pi, "rep", rep, "mismatches", len(bad), bad[:16], bad[-4:] if len(bad) else "")
    eng.close()
print("---- roundtrip + linearity at logm 11")
logm = 11; m = 1 << logm; n = m // 8
Q = BO.find_modulus(2 * m, 1 << 50); B = 1 << 26
eng = S.Engine(S.Params.custom(n, Q, B))
C = RM.Consts(n, m, Q, B, Q // 8); N = RM.NttModel(logm)
p = C.primes[0]
rng = np.random.default_rng(5)
poly = rng.integers(0, p, size=m, dtype=np.uint64)
fwd = eng.debug_ntt(0, poly.astype(np.uint32))
back = eng.debug_ntt(0, fwd, inverse=True).astype(np.uint64)
print("roundtrip ok:", np.array_equal(back, poly))
# delta at position 0 -> all slots should be 1 ; delta at 1 -> slot k = psi^(2 bitrev(k)+1)
d0 = np.zeros(m, dtype=np.uint32); d0[0] = 1
f0 = eng.debug_ntt(0, d0); print("delta0 all ones:", bool((f0 == 1).all()), f0[:8])
d1 = np.zeros(m, dtype=np.uint32); d1[1] = 1
f1 = eng.debug_ntt(0, d1).astype(np.uint64)
psi = C.pk[0]["psi"]
exp = np.array([pow(psi, 2 * RM.bitrev(k, logm) + 1, p) for k in range(m)], dtype=np.uint64)
print("delta1 matches psi powers:", np.array_equal(f1, exp), "mismatch count", int((f1 != exp).sum()))
# is f1[0] a primitive 2m-th root?
g = int(f1[0]); print("f1[0]", g, "model psi", psi, "g^m == -1:", pow(g, m, p) == p - 1)
mdl = N.forward(N.to_regs(d1.astype(np.uint64)), C.pk[0]["twf"], p).reshape(-1) % p
print("model delta1 == exp:", np.array_equal(mdl, exp))
