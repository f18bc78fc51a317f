import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np
import sgfhe_jl_amd as S, rns_model as RM, bigint_oracle as BO
logm = 11; m = 1 << logm; n = m // 8
Q = BO.find_modulus(2 * m, 1 << 50); B = 1 << 26
eng = S.Engine(S.Params.custom(n, Q, B))
C = RM.Consts(n, m, Q, B, Q // 8); N = RM.NttModel(logm)
p = C.primes[0]
def ranges(idx):
    out = []; s = None; prev = None
    for i in idx:
        if s is None: s = prev = i
        elif i == prev + 1: prev = i
        else: out.append((s, prev)); s = prev = i
    if s is not None: out.append((s, prev))
    return out
for pos in (0, 1, 256, 300, 2047):
    d = np.zeros(m, dtype=np.uint32); d[pos] = 1
    f = eng.debug_ntt(0, d).astype(np.uint64)
    mdl = N.forward(N.to_regs(d.astype(np.uint64)), C.pk[0]["twf"], p).reshape(-1) % p
    bad = np.nonzero(f != mdl)[0]
    print("delta at", pos, "bad count", len(bad), "ranges", ranges(bad.tolist())[:12])
    if pos == 0:
        print("  values at bad[:8]", f[bad[:8]], "nonzero count", int((f != 0).sum()))
