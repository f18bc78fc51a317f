"""
Host model of the device algorithm (sgfhe.jl_amd/csrc/{rns_arith,ntt,kernels}.h, engine.hip):
same RNS primes, constants, pass structure, LDS swizzle and index formulas, written with numpy so
that the restructured algorithm and its indexing can be checked against the oracle without a GPU.
Test infrastructure only (used by tests/test_rns_model.py).
"""

import numpy as np

NPR_MAX = 6
MASK32 = 0xFFFFFFFF


# ---- engine.hip: build_constants -------------------------------------------------------------

def is_prime32(x):
    if x < 2:
        return False
    for q in (2, 3, 5, 7, 11, 13):
        if x % q == 0:
            return x == q
    d, s = x - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7):
        y = pow(a, d, x)
        if y in (1, x - 1):
            continue
        for _ in range(s - 1):
            y = y * y % x
            if y == x - 1:
                break
        else:
            return False
    return True


def rns_primes(count=NPR_MAX):
    """The `count` largest primes below 2^30 that are 1 mod 2^15 (candidates of build_constants)."""
    out = []
    kk = ((1 << 30) - 1) >> 15
    while len(out) < count:
        cand = (kk << 15) + 1
        if cand < (1 << 30) and is_prime32(cand):
            out.append(cand)
        kk -= 1
    return out


def select_npr(logm, B, Q):
    """build_constants: the fewest primes covering 8 m B Q (times 4 for the randomised flatten when
    NPR_MAX primes allow it), at least 2; same floating-point rule as the engine."""
    import math
    cand = rns_primes()
    need = 3.0 + logm + math.log2(float(B)) + math.log2(float(Q)) + 0.01

    def count(target):
        have, k = 0.0, 0
        while k < NPR_MAX and (k < 2 or have < target):
            have += math.log2(float(cand[k]))
            k += 1
        return k, have

    k, have = count(need + 2.0)
    if have < need + 2.0:
        k, have = count(need)
    assert have >= need, "exactness bound"
    return k


def bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


class Consts:
    def __init__(self, n, m, Q, B, DQ_tilde):
        self.n, self.M, self.Q, self.B = n, m, Q, B
        self.logm = m.bit_length() - 1
        self.npr = NPR = select_npr(self.logm, B, Q)
        self.primes = rns_primes()[:NPR]
        prod = 1
        for p in self.primes:
            prod *= p
        self.Mrns = prod
        assert 8 * m * B * Q < prod, "exactness bound"
        self.s = (B - 1) // 2 if B % 2 else B // 2 - 1
        self.off = (1 + B) * self.s % Q
        self.DQ = DQ_tilde % Q
        self.c = [prod // p % Q for p in self.primes]
        plast = self.primes[-1]
        cM = prod % Q
        cH = self.c[-1] * ((plast - 1) // 2) % Q
        self.T = [(Q - (a * cM + cH) % Q) % Q for a in range(NPR + 1)]
        self.pk = []
        for i, p in enumerate(self.primes):
            psi = None
            for x in range(2, 2000):
                g = pow(x, (p - 1) // (2 * m), p)
                if pow(g, m, p) == p - 1:
                    psi = g
                    break
            ipsi = pow(psi, p - 2, p)
            R1 = (1 << 32) % p
            twf = np.zeros(m, dtype=np.uint64)
            twi = np.zeros(m, dtype=np.uint64)
            pw = ipw = 1
            for t in range(m):
                br = bitrev(t, self.logm)
                twf[br] = pw * R1 % p          # Montgomery form
                twi[br] = ipw * R1 % p
                pw = pw * psi % p
                ipw = ipw * ipsi % p
            Rinv = pow(R1, p - 2, p)
            Mi = prod // p % p
            ei = pow(Mi, p - 2, p)
            minv = pow(m, p - 2, p)
            kappa = R1 * R1 % p * minv % p * ei % p
            self.pk.append(dict(
                p=p, ninv=(-pow(p, -1, 1 << 32)) & MASK32, sR=p - self.s % p * Rinv % p,
                hoff=(p - 1) // 2 if i == NPR - 1 else 0, r1=R1, r2=R1 * R1 % p,
                r3=R1 * R1 * R1 % p, qmodp=Q % p, kappaR=kappa * R1 % p, minvR=minv * R1 % p,
                twf=twf, twi=twi, psi=psi, kappa=kappa, ei=ei))

    def digits_of(self, acc):
        x = (acc + self.off) % self.Q
        return x % self.B, x // self.B


# ---- rns_arith.h ---------------------------------------------------------------------------------

def u32(a):
    return np.asarray(a, dtype=np.uint64) & MASK32


def csub(x, p):
    x = u32(x)
    return np.minimum(x, u32(x + np.uint64((1 << 32) - p)))


def mulhi(a, b):
    return (u32(a) * u32(b)) >> 32


def redc64(T, p, ninv):
    T = np.asarray(T, dtype=np.uint64)
    tlo, thi = T & MASK32, T >> 32
    mq = u32(tlo * ninv)
    h = mulhi(mq, p)
    return u32(thi + h + (tlo != 0))


def redc_mad(T, p, ninv):
    """hi32(T + (T_lo * ninv mod 2^32) * p): the v_mad_u64_u32 form of REDC."""
    T = np.asarray(T, dtype=np.uint64)
    mq = u32((T & MASK32) * np.uint64(ninv))
    return (mq * np.uint64(p) + T) >> 32


def mont_mul(a, b, p, ninv):
    return csub(redc64(u32(a) * u32(b), p, ninv), p)


def mont_lazy(y, wM, p, ninv):
    T = u32(wM) * u32(y)
    mq = u32((T & MASK32) * ninv)
    U = mq * np.uint64(p) + T
    return U >> 32


def bfly_fwd(X, Y, wM, p, ninv):
    p2 = 2 * p
    x = np.minimum(u32(X), u32(u32(X) + np.uint64((1 << 32) - p2)))
    t = mont_lazy(Y, wM, p, ninv)
    return u32(x + t), u32(x + p2 + np.uint64(1 << 32) - t)


def bfly_inv(X, Y, wM, p, ninv):
    p2 = 2 * p
    s = u32(X + Y)
    t = u32(X + p2 + np.uint64(1 << 32) - Y)
    return np.minimum(s, u32(s + np.uint64((1 << 32) - p2))), mont_lazy(t, wM, p, ninv)


# ---- ntt.h -----------------------------------------------------------------------------------------

LOGE = 4


def swz(idx, loge=LOGE):
    idx = np.asarray(idx, dtype=np.int64)
    if loge == 3:
        return idx ^ (((idx >> 6) & 1) * 0x09) ^ (((idx >> 7) & 1) * 0x12) ^ (((idx >> 5) & 1) * 0x04)
    return (idx ^ (((idx >> 5) & 1) * 0x01) ^ (((idx >> 6) & 1) * 0x02) ^ (((idx >> 7) & 1) * 0x04)
            ^ (((idx >> 8) & 1) * 0x18))


class NttModel:
    """x has shape [T, E] (one polynomial); lds is a flat array of M words."""

    def __init__(self, logm, loge=LOGE):
        self.LOGM, self.LOGE = logm, loge
        self.M = 1 << logm
        self.E = 1 << loge
        self.T = self.M // self.E
        self.RHO = logm % loge
        self.STOP = logm - loge
        self.SFIRST = logm - self.RHO - loge if self.RHO else logm - 2 * loge
        self.SLAST_INV = logm - self.RHO - loge if self.RHO else self.STOP
        self.tid = np.arange(self.T, dtype=np.int64)

    def lds_addr(self, S, e):
        lo = self.tid & ((1 << S) - 1)
        hi = self.tid >> S
        return swz((hi << (S + self.LOGE)) | lo, self.LOGE) ^ int(swz(e << S, self.LOGE))

    def store(self, x, lds, S):
        for e in range(self.E):
            lds[self.lds_addr(S, e)] = x[:, e]

    def load(self, lds, S):
        x = np.zeros((self.T, self.E), dtype=np.uint64)
        for e in range(self.E):
            x[:, e] = lds[self.lds_addr(S, e)]
        return x

    def stage(self, x, tw, p, ninv, B, S, fwd):
        """butterflies on local bit B of the pass over [S, S + LOGE)."""
        hi = self.tid >> S
        base = (1 << (self.LOGM - 1 - S - B)) + (hi << (self.LOGE - 1 - B))
        f = bfly_fwd if fwd else bfly_inv
        for g in range(1 << (self.LOGE - 1 - B)):
            w = tw[base + g]
            for l in range(1 << B):
                e0 = (g << (B + 1)) | l
                e1 = e0 | (1 << B)
                x[:, e0], x[:, e1] = f(x[:, e0], x[:, e1], w, p, ninv)

    def forward(self, x, tw, p, ninv):
        """x[tid, e] = coefficient tid + T e -> slot E tid + e."""
        x = x.copy()
        lds = np.zeros(self.M, dtype=np.uint64)
        blo = 0 if self.RHO == 0 else self.LOGE - self.RHO
        for B in range(self.LOGE - 1, blo - 1, -1):
            self.stage(x, tw, p, ninv, B, self.STOP, True)
        sprev, S = self.STOP, self.SFIRST
        while S >= 0:
            self.store(x, lds, sprev)
            x = self.load(lds, S)
            for B in range(self.LOGE - 1, -1, -1):
                self.stage(x, tw, p, ninv, B, S, True)
            sprev, S = S, S - self.LOGE
        return x

    def inverse(self, x, tw, p, ninv):
        """slots E tid + e -> coefficient tid + T e (unscaled)."""
        x = x.copy()
        lds = np.zeros(self.M, dtype=np.uint64)
        if self.SLAST_INV >= 0:
            S = 0
            while True:
                for B in range(self.LOGE):
                    self.stage(x, tw, p, ninv, B, S, False)
                if S >= self.SLAST_INV:
                    break
                self.store(x, lds, S)
                x = self.load(lds, S + self.LOGE)
                S += self.LOGE
        if self.RHO:
            if self.SLAST_INV >= 0:
                self.store(x, lds, self.SLAST_INV)
                x = self.load(lds, self.STOP)
            for B in range(self.LOGE - self.RHO, self.LOGE):
                self.stage(x, tw, p, ninv, B, self.STOP, False)
        return x

    def to_regs(self, poly):
        """natural-order polynomial -> [T, E] register layout (coefficient tid + T e)."""
        return np.asarray(poly, dtype=np.uint64).reshape(self.E, self.T).T.copy()

    def from_regs(self, x):
        return x.T.reshape(-1).copy()


def ntt_reference(poly, psi, p):
    """Evaluations of poly at psi^(2 bitrev(k) + 1), k = 0..m-1: the slot order of the merged
    Cooley-Tukey transform."""
    m = len(poly)
    logm = m.bit_length() - 1
    out = []
    for k in range(m):
        x = pow(psi, 2 * bitrev(k, logm) + 1, p)
        acc = 0
        for c in reversed(poly):
            acc = (acc * x + int(c)) % p
        out.append(acc)
    return out


# ---- kernels.h: one k-loop iteration ---------------------------------------------------------------

class EngineModel:
    def __init__(self, n, m, Q, B, DQ_tilde):
        self.C = Consts(n, m, Q, B, DQ_tilde)
        self.ntt = NttModel(self.C.logm)

    def key_transform(self, canon_poly, pi):
        """k_key_transform for one polynomial (list of ints in [0, Q)) and prime index."""
        C, P = self.C, self.C.pk[pi]
        p, ninv = P["p"], P["ninv"]
        vals = np.zeros(C.M, dtype=np.uint64)
        for i, v in enumerate(canon_poly):
            c0, c1, c2 = v & MASK32, (v >> 32) & MASK32, (v >> 64) & MASK32
            r = int(csub(int(mont_mul(c0, P["r1"], p, ninv)) + int(mont_mul(c1, P["r2"], p, ninv)), p))
            r = int(csub(r + int(mont_mul(c2, P["r3"], p, ninv)), p))
            if v > C.Q // 2:
                r = (r - P["qmodp"]) % p
            vals[i] = int(mont_mul(r, P["kappaR"], p, ninv))
        x = self.ntt.forward(self.ntt.to_regs(vals), P["twf"], p, ninv)
        x = np.minimum(x, u32(x - 2 * p + (1 << 32)))
        x = csub(x, p)
        return x.reshape(-1)          # slot E tid + e

    def extprod(self, dig_a, dig_b, keyslice, j, plain=False):
        """k_extprod for one bootstrap: dig_* lists of (lo, hi); keyslice[pi][row*2+col] slot
        arrays; returns y[c][pi] arrays (natural order)."""
        C = self.C
        NPR = C.npr
        M, T = C.M, self.ntt.T
        ys = [[None] * NPR for _ in range(2)]
        for pi in range(NPR):
            P = C.pk[pi]
            p, ninv = P["p"], P["ninv"]
            digs = [np.array([d[0] for d in dig_a], dtype=np.uint64),
                    np.array([d[1] for d in dig_a], dtype=np.uint64),
                    np.array([d[0] for d in dig_b], dtype=np.uint64),
                    np.array([d[1] for d in dig_b], dtype=np.uint64)]
            U = []
            for d in digs:
                v = u32(redc_mad(d, p, ninv) + P["sR"])          # digit_reduce: lazy, in [0, 4p)
                x = self.ntt.forward(self.ntt.to_regs(v), P["twf"], p, ninv)
                U.append(np.minimum(x, u32(x + np.uint64((1 << 32) - 2 * p))).reshape(-1))   # [0, 2p)
            for c in range(2):
                if c == 0:
                    # column 0: 64-bit accumulation over the four phases, one reduction
                    acc = sum(U[row] * keyslice[pi][row * 2] for row in range(4))
                    r = redc_mad(acc, p, ninv)                                   # [0, 3p)
                    z = np.minimum(r, u32(r + np.uint64((1 << 32) - 2 * p)))
                else:
                    # column 1: reduced per phase, summed lazily mod 2p (LDS accumulator)
                    z = np.zeros(M, dtype=np.uint64)
                    for row in range(4):
                        zs = u32(z + redc_mad(U[row] * keyslice[pi][row * 2 + 1], p, ninv))
                        z = np.minimum(zs, u32(zs + np.uint64((1 << 32) - 2 * p)))
                z = z.reshape(T, self.ntt.E)
                z = self.ntt.inverse(z, P["twi"], p, ninv)
                Pn = self.ntt.from_regs(csub(z, p))            # natural order
                if plain:
                    ys[c][pi] = csub(Pn + P["hoff"], p)
                    continue
                i = np.arange(M, dtype=np.int64)
                s = (i - j) & (2 * M - 1)
                v = Pn[s & (M - 1)]
                v = np.where((s & M) != 0, csub(p - v, p), v)
                y = u32(v - Pn + (1 << 32))
                y = np.minimum(y, u32(y + p))
                ys[c][pi] = csub(y + P["hoff"], p)
        return ys

    def crt_acc(self, ys_c, dig_old, noacc=False, canon=False):
        """k_crt_acc for one polynomial: ys_c[pi] arrays -> new digits (or canonical values)."""
        C = self.C
        NPR = C.npr
        out = []
        for i in range(C.M):
            y = [int(ys_c[pi][i]) for pi in range(NPR)]
            f = np.float32(0)
            for pi in range(NPR):
                f = np.float32(f + np.float32(y[pi]) * np.float32(np.float32(1.0) / np.float32(C.primes[pi])))
            alpha = int(f)
            S = C.T[alpha] + sum(y[pi] * C.c[pi] for pi in range(NPR))
            if not noacc:
                S += dig_old[i][1] * C.B + dig_old[i][0]
            xn = S % C.Q
            out.append(xn if canon else (xn % C.B, xn // C.B))
        return out

    def acc_from_digits(self, dig):
        C = self.C
        return [(d[1] * C.B + d[0] - C.off) % C.Q for d in dig]
