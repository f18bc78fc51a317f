"""
Host model of the device algorithm (sgfhe.jl_amd/csrc/{rns_arith,ntt,kernels}.h, engine.hip):
same RNS primes, constants, pass structure, LDS swizzle and index formulas, written with numpy so
that the restructured algorithm and its indexing can be checked against the oracle without a GPU.
Test infrastructure only (used by tests/test_rns_model.py).
"""

import numpy as np

NPR_MAX = 7
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
    """The `count` largest primes below 2^29 that are 1 mod 2^15 (candidates of build_constants)."""
    out = []
    kk = ((1 << 29) - 1) >> 15
    while len(out) < count:
        cand = (kk << 15) + 1
        if cand < (1 << 29) and is_prime32(cand):
            out.append(cand)
        kk -= 1
    return out


def select_npr(logm, B, Q, random_flatten=False):
    """build_constants: the fewest primes covering 5 m B Q (20 m B Q when the ctx is created for
    the randomised flatten), at least 2; same floating-point rule as the engine."""
    import math
    cand = rns_primes()
    need = math.log2(5.0) + logm + math.log2(float(B)) + math.log2(float(Q)) + 0.001
    target = need + (2.0 if random_flatten else 0.0)
    have, k = 0.0, 0
    while k < NPR_MAX and (k < 2 or have < target):
        have += math.log2(float(cand[k]))
        k += 1
    assert have >= target, "exactness bound"
    return k


def bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def centre(x, p):
    x %= p
    return x - p if x > (p - 1) // 2 else x


class Consts:
    def __init__(self, n, m, Q, B, DQ_tilde, random_flatten=False):
        self.n, self.M, self.Q, self.B = n, m, Q, B
        self.logm = m.bit_length() - 1
        self.npr = NPR = select_npr(self.logm, B, Q, random_flatten)
        self.primes = rns_primes()[:NPR]
        prod = 1
        for p in self.primes:
            prod *= p
        self.Mrns = prod
        assert 5 * m * B * Q < prod, "exactness bound"
        self.s = (B - 1) // 2 if B % 2 else B // 2 - 1
        self.off = (1 + B) * self.s % Q
        self.xmax = (B - 1) // 2 * 3 if B % 2 else B // 2 * 3
        self.off_rnd = (1 + B) * (self.s + self.xmax) % Q
        self.DQ = DQ_tilde % Q
        self.c = [prod // p % Q for p in self.primes]
        plast = self.primes[-1]
        cM = prod % Q
        cH = self.c[-1] * ((plast - 1) // 2) % Q
        self.T = [(Q - (a * cM + cH) % Q) % Q for a in range(6 * NPR + 2)]
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
            twf = np.zeros(m, dtype=np.int64)
            twi = np.zeros(m, dtype=np.int64)
            pf, pv = [0] * m, [0] * m
            pw = ipw = 1
            for t in range(m):
                br = bitrev(t, self.logm)
                twf[br] = centre(pw * R1, p)          # Montgomery form, centred
                twi[br] = centre(ipw * R1, p)
                pf[br], pv[br] = pw, ipw
                pw = pw * psi % p
                ipw = ipw * ipsi % p
            # product twiddles of the forward radix-4 steps (table tw + 2 m on the device):
            # twfp[j] = +-tw[j >> 1] tw[j], minus for odd j
            twfp = np.zeros(m, dtype=np.int64)
            twip = np.zeros(m, dtype=np.int64)
            for j in range(2, m):
                v = pf[j >> 1] * pf[j] % p * R1 % p
                twfp[j] = centre((p - v) % p if j & 1 else v, p)
                v = pv[j >> 1] * pv[j] % p * R1 % p
                twip[j] = centre((p - v) % p if j & 1 else v, p)
            Rinv = pow(R1, p - 2, p)
            Mi = prod // p % p
            ei = pow(Mi, p - 2, p)
            minv = pow(m, p - 2, p)
            kappa = R1 * R1 % p * minv % p * ei % p
            self.pk.append(dict(
                p=p, pinv=pow(p, -1, 1 << 32), sR=centre(-(self.s % p) * Rinv, p),
                sRr=centre(-((self.s + self.xmax) % p) * Rinv, p),
                hoff=(p - 1) // 2 if i == NPR - 1 else 0, r1=centre(R1, p), r2=centre(R1 * R1, p),
                r3=centre(R1 * R1 * R1, p), qmodp=centre(Q, p), kappaR=centre(kappa * R1, p),
                minvR=centre(minv * R1, p), twf=twf, twi=twi, twfp=twfp, twip=twip, psi=psi, kappa=kappa, ei=ei))

    def digits_of(self, acc):
        x = (acc + self.off) % self.Q
        return x % self.B, x // self.B


# ---- rns_arith.h ---------------------------------------------------------------------------------
# int32 device values are modelled as numpy int64 arrays; `i32` checks that nothing left the
# int32 range (a wrap-around on the device would be a silent error) and records the largest
# magnitude seen, in units of 2^29.

PEAK = {"v": 0.0}


def i32(x, limit=1 << 31):
    x = np.asarray(x, dtype=np.int64)
    if x.size:
        mx = int(np.max(np.abs(x)))
        assert mx < limit, "int32 range exceeded: %d" % mx
        PEAK["v"] = max(PEAK["v"], mx / float(1 << 29))
    return x


def sredc(T, P):
    """Signed Montgomery reduction: T R^-1 mod p, |T| < 2^62."""
    T = np.asarray(T, dtype=np.int64)
    assert int(np.max(np.abs(T))) < (1 << 62) if T.size else True
    lo = T & MASK32
    m = (lo * np.int64(P["pinv"])) & MASK32
    m = np.where(m >= (1 << 31), m - (1 << 32), m)
    U = T - m * np.int64(P["p"])
    assert not np.any(U & MASK32)
    return i32(U >> 32)


def smont(a, w, P):
    return sredc(i32(a) * np.asarray(w, dtype=np.int64), P)


def sred(x, P):
    x = i32(x)
    i32(x + (1 << 28))                       # the rounding add itself must not overflow
    q = (x + (1 << 28)) >> 29
    return i32(x - q * np.int64(P["p"]))


def sred_floor(x, P):
    """rns_arith.h sred_floor: x - floor(x / 2^29) p for any int32 x; result in [-4 delta, 2^29 + 3 delta)."""
    x = i32(x)
    r = i32(x - (x >> 29) * np.int64(P["p"]))
    d = (1 << 29) - P["p"]
    assert int(r.min()) >= -4 * d and int(r.max()) < (1 << 29) + 3 * d
    return r


def scanon(x, P):
    x = i32(x)
    assert int(np.max(np.abs(x))) < P["p"]
    return np.where(x < 0, x + P["p"], x)


def sfull(x, P):
    return scanon(sred(x, P), P)


def scentre(x, P):
    x = i32(x).copy()
    h = (P["p"] - 1) >> 1
    x = np.where(x > h, x - P["p"], x)
    x = np.where(x < -h, x + P["p"], x)
    return x


def bfly_fwd(X, Y, wM, P):
    t = smont(Y, wM, P)
    return i32(X + t), i32(X - t)


def bfly_inv(X, Y, wM, P, red):
    s, d = i32(X + Y), i32(X - Y)
    return (sred(s, P) if red else s), smont(d, wM, P)


# ---- ntt.h -----------------------------------------------------------------------------------------

LOGE = 4


def swz(idx, loge=LOGE):
    idx = np.asarray(idx, dtype=np.int64)
    if loge == 3:
        return idx ^ (((idx >> 6) & 1) * 0x09) ^ (((idx >> 7) & 1) * 0x12) ^ (((idx >> 5) & 1) * 0x04)
    return (idx ^ (((idx >> 5) & 1) * 0x01) ^ (((idx >> 6) & 1) * 0x02) ^ (((idx >> 7) & 1) * 0x04)
            ^ (((idx >> 8) & 1) * 0x18))


def inv_red_mask(loge, B, bfirst, bhi, lastred):
    """ntt.h inv_red_mask: the set of X registers e0 whose sum is range-reduced in inverse stage B."""
    E = 1 << loge
    if loge == 4 and bfirst == 0 and bhi == 3 and lastred:
        out = set()
        for e0 in range(E):
            if e0 & (1 << B):
                continue
            if lastred == 3:        # first pass, inputs up to 1.5 * 2^29
                red = B == 0 or B == 3 or (B == 2 and (e0 & 2) == 0)
            else:
                red = ((e0 & 1) == 0) if B == 1 else ((e0 & 3) == 1) if B == 2 else \
                    ((e0 in (0, 2, 3)) if lastred == 2 else True) if B == 3 else False
            if red:
                out.add(e0)
        return out
    allred = bool((B - bfirst) & 1) or (lastred and B == bhi)
    return {e0 for e0 in range(E) if not e0 & (1 << B)} if allred else set()


def fwd_vec4(logm):
    """ntt.h SGFHE_FWD_VEC4: forward passes with per-lane twiddles take the radix-4 form except where
    k_extprod would spill registers (m = 4096, 16384)."""
    return logm not in (12, 14)


def inv_r4(logm):
    """ntt.h SGFHE_INV_R4: the inverse keeps its radix-2 form at m = 4096."""
    return logm != 12


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
        x = np.zeros((self.T, self.E), dtype=np.int64)
        for e in range(self.E):
            x[:, e] = lds[self.lds_addr(S, e)]
        return x

    def stage(self, x, tw, P, B, S, fwd, red=()):
        """butterflies on local bit B of the pass over [S, S + LOGE)."""
        hi = self.tid >> S
        base = (1 << (self.LOGM - 1 - S - B)) + (hi << (self.LOGE - 1 - B))
        for g in range(1 << (self.LOGE - 1 - B)):
            w = tw[base + g]
            for l in range(1 << B):
                e0 = (g << (B + 1)) | l
                e1 = e0 | (1 << B)
                if fwd:
                    x[:, e0], x[:, e1] = bfly_fwd(x[:, e0], x[:, e1], w, P)
                else:
                    x[:, e0], x[:, e1] = bfly_inv(x[:, e0], x[:, e1], w, P, e0 in red)

    def step4(self, x, tw, twp, P, BH, S):
        """ntt.h fwd_step4: the forward stages on local bits BH and BH - 1 as one radix-4 step, the
        products of the second stage summed in 64 bits before one Montgomery reduction."""
        hi = self.tid >> S
        nga = 1 << (self.LOGE - 1 - BH)
        base_a = (1 << (self.LOGM - 1 - S - BH)) + (hi << (self.LOGE - 1 - BH))
        base_b = (1 << (self.LOGM - S - BH)) + (hi << (self.LOGE - BH))
        lo, hb = 1 << (BH - 1), 1 << BH
        for g in range(nga):
            wA = tw[base_a + g]
            wB0, wB1 = tw[base_b + 2 * g], tw[base_b + 2 * g + 1]
            P0, P1 = twp[base_b + 2 * g], twp[base_b + 2 * g + 1]
            for l in range(lo):
                e0 = (g << (BH + 1)) | l
                X0, X1, X2, X3 = (i32(x[:, e]) for e in (e0, e0 | lo, e0 | hb, e0 | hb | lo))
                u = smont(X2, wA, P)
                a, b = i32(X0 + u), i32(X0 - u)
                s_ = sredc(X1 * np.asarray(wB0, dtype=np.int64) + X3 * np.asarray(P0, dtype=np.int64), P)
                r_ = sredc(X1 * np.asarray(wB1, dtype=np.int64) + X3 * np.asarray(P1, dtype=np.int64), P)
                x[:, e0], x[:, e0 | lo] = i32(a + s_), i32(a - s_)
                x[:, e0 | hb], x[:, e0 | hb | lo] = i32(b + r_), i32(b - r_)

    def forward(self, x, tw, P):
        """x[tid, e] = coefficient tid + T e -> slot E tid + e."""
        x = np.asarray(x, dtype=np.int64).copy()
        twp = P["twfp"]
        lds = np.zeros(self.M, dtype=np.int64)
        blo = 0 if self.RHO == 0 else self.LOGE - self.RHO
        for B in range(self.LOGE - 1, blo - 1, -1):
            self.stage(x, tw, P, B, self.STOP, True)
        sprev, S = self.STOP, self.SFIRST
        E = self.E
        while S >= 0:
            self.store(x, lds, sprev)
            x = self.load(lds, S)
            if self.LOGE == 4 and (1 << S) >= 64:          # wave-uniform twiddles: two radix-4 steps
                for BH in (3, 1):
                    x0 = [e for e in range(E) if (e & (3 << (BH - 1))) == 0]   # fwd_reduce_x0
                    x[:, x0] = sred_floor(x[:, x0], P)
                    self.step4(x, tw, twp, P, BH, S)
            elif (1 << S) >= 64 or fwd_vec4(self.LOGM):    # one radix-4 step, then radix 2
                x[:, :E // 2] = sred_floor(x[:, :E // 2], P)                   # fwd_reduce_x
                self.step4(x, tw, twp, P, self.LOGE - 1, S)
                for B in range(self.LOGE - 3, -1, -1):
                    self.stage(x, tw, P, B, S, True)
            else:                                          # per-lane twiddles at m = 4096 / 16384: radix 2
                x[:, :E // 2] = sred_floor(x[:, :E // 2], P)
                for B in range(self.LOGE - 1, -1, -1):
                    self.stage(x, tw, P, B, S, True)
            sprev, S = S, S - self.LOGE
        return x

    def inv_stages(self, x, tw, P, S, blo, bhi, lastred):
        for B in range(blo, bhi + 1):
            self.stage(x, tw, P, B, S, False, inv_red_mask(self.LOGE, B, blo, bhi, lastred))

    def step4_inv(self, x, tw, twp, P, B, S):
        """ntt.h inv_step4: the Gentleman-Sande stages on local bits B and B + 1 as one radix-4 step:
        y0 = x0 + x1 + x2 + x3, y2 = wA ((x0 + x1) - (x2 + x3)), and the two outputs that go through
        both twiddles from 64-bit sums of two products, one Montgomery reduction each."""
        hi = self.tid >> S
        nga = 1 << (self.LOGE - 2 - B)
        base_a = (1 << (self.LOGM - 2 - S - B)) + (hi << (self.LOGE - 2 - B))
        base_b = (1 << (self.LOGM - 1 - S - B)) + (hi << (self.LOGE - 1 - B))
        lo, hb = 1 << B, 1 << (B + 1)
        for g in range(nga):
            wA = np.asarray(tw[base_a + g], dtype=np.int64)
            wB0, wB1 = (np.asarray(tw[base_b + 2 * g + k], dtype=np.int64) for k in (0, 1))
            P0, P1 = (np.asarray(twp[base_b + 2 * g + k], dtype=np.int64) for k in (0, 1))
            for l in range(lo):
                e0 = (g << (B + 2)) | l
                x0, x1, x2, x3 = (i32(x[:, e]) for e in (e0, e0 | lo, e0 | hb, e0 | hb | lo))
                s0, s1, d0, d1 = i32(x0 + x1), i32(x2 + x3), i32(x0 - x1), i32(x2 - x3)
                x[:, e0] = i32(s0 + s1)
                x[:, e0 | hb] = smont(i32(s0 - s1), wA, P)
                x[:, e0 | lo] = sredc(d0 * wB0 + d1 * wB1, P)
                x[:, e0 | hb | lo] = sredc(d0 * P0 + d1 * P1, P)

    def wide_ok(self):
        """k_extprod's WIDE0 (column 0 enters the inverse un-reduced, |.| < 1.5 * 2^29): only where the
        inverse is the radix-2 one; the radix-4 inverse takes every polynomial below 0.75 * 2^29."""
        return (not inv_r4(self.LOGM)) and self.LOGE == 4 and self.SLAST_INV >= 0 and \
            not (self.RHO == 0 and self.STOP == 0)

    def inv_pass(self, x, tw, P, S, final):
        """One full inverse pass over [S, S + LOGE) as ntt.h InvPasses runs it."""
        E = self.E
        twp = P["twip"]
        sums4 = lambda B: [e for e in range(E) if (e & (3 << B)) == 0]      # the y0 registers of a step on (B, B + 1)
        self.step4_inv(x, tw, twp, P, 0, S)
        x[:, sums4(0)] = sred(x[:, sums4(0)], P)
        if self.LOGE == 4 and (1 << S) >= 64:          # wave-uniform twiddles: second radix-4 step
            self.step4_inv(x, tw, twp, P, 2, S)
            x[:, sums4(2)] = sred(x[:, sums4(2)], P)
            return
        for B in range(2, self.LOGE):                  # radix-2 stages above the step
            last = B == self.LOGE - 1
            if not last:
                red = ()
            elif final:                                # outputs only have to stay below 1.4 * 2^29:
                red = [e for e in range(E) if (e & (3 << (B - 1))) == 0] if self.LOGE == 4 else ()
            else:
                red = [e for e in range(E) if not e & (1 << B)]
            self.stage(x, tw, P, B, S, False, red)

    def inverse(self, x, tw, P, wide=False):
        """slots E tid + e -> coefficient tid + T e (unscaled).  wide: ntt_inverse<..., WIDE0> for
        this polynomial (radix-2 inverse only)."""
        assert not wide or self.wide_ok()
        x = np.asarray(x, dtype=np.int64).copy()
        lds = np.zeros(self.M, dtype=np.int64)
        if self.SLAST_INV >= 0:
            S = 0
            while True:
                final = (not self.RHO) and S >= self.SLAST_INV     # InvPasses<..., FINAL>
                if inv_r4(self.LOGM):
                    self.inv_pass(x, tw, P, S, final)
                else:
                    mode = 3 if (wide and S == 0) else (2 if final else 1)
                    self.inv_stages(x, tw, P, S, 0, self.LOGE - 1, mode)
                if S >= self.SLAST_INV:
                    break
                self.store(x, lds, S)
                x = self.load(lds, S + self.LOGE)
                S += self.LOGE
        if self.RHO:
            if self.SLAST_INV >= 0:
                self.store(x, lds, self.SLAST_INV)
                x = self.load(lds, self.STOP)
            self.inv_stages(x, tw, P, self.STOP, self.LOGE - self.RHO, self.LOGE - 1, 0)
        return x

    def to_regs(self, poly):
        """natural-order polynomial -> [T, E] register layout (coefficient tid + T e)."""
        return np.asarray(poly, dtype=np.int64).reshape(self.E, self.T).T.copy()

    def from_regs(self, x):
        return x.T.reshape(-1).copy()


# ---- worst-case range analysis of the same pass structure -------------------------------------------
# Magnitude bounds in units of 2^29 (p < 2^29).  smont: |t| <= |a| / 16 + 0.5; sred: |r| <= 0.5 +
# 4 delta with delta = (2^29 - p) / 2^29; sred needs x + 2^28 < 2^31, i.e. a bound below 3.5.

class RangeModel:
    def __init__(self, logm, loge, p):
        self.N = NttModel(logm, loge)
        self.delta = ((1 << 29) - p) / float(1 << 29)
        self.peak = 0.0

    def _chk(self, b, limit=4.0):
        assert b < limit, "bound %.3f * 2^29 exceeds %.1f" % (b, limit)
        self.peak = max(self.peak, b)
        return b

    def sred(self, b):
        self._chk(b, 3.5)
        return 0.5 + 4 * self.delta

    def forward(self, b_in):
        N = self.N
        b = b_in
        first = N.RHO if N.RHO else N.LOGE
        for _ in range(first):                               # register pass: no reduction
            b = self._chk(b + b / 16 + 0.5)
        red = 1.0 + 3 * self.delta                           # sred_floor output: [-4 delta, 1 + 3 delta)
        S = N.SFIRST
        while S >= 0:
            self._chk(b)                                     # sred_floor takes any int32
            if N.LOGE == 4 and (1 << S) >= 64:
                # two radix-4 steps; before each, its X0 inputs are reduced; X1..X3 are whatever
                # the previous step (or pass) left: u <= X2 / 16 + 0.5, s <= (X1 + X3) / 16 + 0.5
                for _step in range(2):
                    assert 2 * b / 16 * 2 ** 29 * 2 ** 28 < 2 ** 62     # the 64-bit sum of two products
                    b = self._chk(red + (b / 16 + 0.5) + (2 * b / 16 + 0.5))
            elif (1 << S) >= 64 or fwd_vec4(N.LOGM):
                # X0, X1 reduced; radix-4 step on the top two stages, radix-2 stages below
                x = self._chk(red + (b / 16 + 0.5) + ((red + b) / 16 + 0.5))
                for _s in range(N.LOGE - 2):
                    x = self._chk(x + x / 16 + 0.5)
                b = x
            else:                                            # radix-2 pass
                x, y = red, b
                for _s in range(N.LOGE):
                    x = self._chk(x + y / 16 + 0.5)
                    y = x
                b = x
            S -= N.LOGE
        return b

    def inverse_radix2(self, b_in, wide=False):
        """Bounds of the radix-2 inverse with the searched reduction pattern (inv_red_mask)."""
        N = self.N
        E = N.E
        assert not wide or N.wide_ok()

        def run(bv, blo, bhi, lastred):
            for B in range(blo, bhi + 1):
                mask = inv_red_mask(N.LOGE, B, blo, bhi, lastred)
                nb = list(bv)
                for e0 in range(E):
                    if e0 & (1 << B):
                        continue
                    e1 = e0 | (1 << B)
                    red = e0 in mask
                    s = self._chk(bv[e0] + bv[e1], 3.5 if red else 4.0)
                    nb[e0] = (0.5 + 4 * self.delta) if red else s
                    nb[e1] = s / 16 + 0.5
                bv = nb
            return bv

        bv = [b_in] * E
        full = (N.SLAST_INV // N.LOGE + 1) if N.SLAST_INV >= 0 else 0
        for i in range(full):
            final = (not N.RHO) and i == full - 1
            bv = [max(run(bv, 0, N.LOGE - 1, 3 if (wide and i == 0) else (2 if final else 1)))] * E
        if N.RHO:
            bv = run(bv, N.LOGE - N.RHO, N.LOGE - 1, False)
        return max(bv)

    def inverse(self, b_in, wide=False):
        """Bounds of the radix-4 inverse (NttModel.inv_pass) from inputs below b_in * 2^29, per
        register: step sums y0 = x0 + x1 + x2 + x3, y2 = smont(s0 - s1), y1 / y3 = one reduction of
        a sum of two products of differences."""
        N = self.N
        E, LOGE = N.E, N.LOGE
        if not inv_r4(N.LOGM):
            return self.inverse_radix2(b_in, wide)
        assert not wide
        red_out = 0.5 + 4 * self.delta

        def step4(bv, B):
            nb = list(bv)
            lo, hb = 1 << B, 1 << (B + 1)
            for e0 in range(E):
                if e0 & (lo | hb):
                    continue
                x0, x1, x2, x3 = bv[e0], bv[e0 | lo], bv[e0 | hb], bv[e0 | hb | lo]
                s0, s1 = self._chk(x0 + x1), self._chk(x2 + x3)
                d0, d1 = s0, s1                                   # |x0 - x1| <= |x0| + |x1|
                nb[e0] = self._chk(s0 + s1)
                nb[e0 | hb] = self._chk(s0 + s1) / 16 + 0.5       # smont of s0 - s1
                assert (d0 + d1) * 2 ** 57 < 2 ** 62              # 64-bit sum of two products
                nb[e0 | lo] = nb[e0 | hb | lo] = (d0 + d1) / 16 + 0.5
            return nb

        def reduce(bv, regs):
            for e in regs:
                bv[e] = self.sred(bv[e])                          # checks the 3.5 precondition
            return bv

        def radix2(bv, B, red):
            nb = list(bv)
            for e0 in range(E):
                if e0 & (1 << B):
                    continue
                e1 = e0 | (1 << B)
                ssum = self._chk(bv[e0] + bv[e1], 3.5 if e0 in red else 4.0)
                nb[e0] = red_out if e0 in red else ssum
                nb[e1] = ssum / 16 + 0.5
            return nb

        bv = [b_in] * E
        if N.SLAST_INV >= 0:
            S = 0
            while True:
                final = (not N.RHO) and S >= N.SLAST_INV
                bv = step4(bv, 0)
                bv = reduce(bv, [e for e in range(E) if (e & 3) == 0])
                if LOGE == 4 and (1 << S) >= 64:
                    bv = step4(bv, 2)
                    bv = reduce(bv, [e for e in range(E) if (e & 12) == 0])
                else:
                    for B in range(2, LOGE):
                        last = B == LOGE - 1
                        if not last:
                            red = ()
                        elif final:
                            red = [e for e in range(E) if (e & (3 << (B - 1))) == 0] if LOGE == 4 else ()
                        else:
                            red = [e for e in range(E) if not e & (1 << B)]
                        bv = radix2(bv, B, red)
                if S >= N.SLAST_INV:
                    break
                bv = [max(bv)] * E                                # the exchange mixes the registers
                S += LOGE
        if N.RHO:
            bv = [max(bv)] * E
            blo, bhi = LOGE - N.RHO, LOGE - 1
            for B in range(blo, bhi + 1):
                mask = inv_red_mask(LOGE, B, blo, bhi, False)
                bv = radix2(bv, B, [e for e in range(E) if e in mask])
        return max(bv)


# ---- k_crt_lean (kernels.h): the k-loop's CRT + accumulate + flatten in integer arithmetic -------
# Model of the device algorithm with every intermediate checked against the width of the register
# it lives in.  Constants as build_constants derives them.

M64 = (1 << 64) - 1


def u64(x):
    assert 0 <= x <= M64, "64-bit overflow on the device"
    return x


class CrtLean:
    """S = sum_i y'_i c_i + alpha (-M mod Q) + hi_o B + lo_o as 29-bit-limb sums L_k
    (64-bit multiply-accumulate chains), quotient by Q and by B through 44 / 52-bit reciprocals
    (estimate = true or true - 1), remainders modulo 2^96 / 2^64, one conditional correction each."""

    def __init__(self, C):
        self.C = C
        Q, B, npr = C.Q, C.B, C.npr
        self.nq, self.nb = Q.bit_length(), B.bit_length()
        self.NL = NL = max(2, -(-self.nq // 29))
        self.ok = self.nb >= 13 and NL <= 4 and 29 * (NL - 1) - (self.nq - 29) <= 28
        if not self.ok:
            return
        lim = lambda v: [(v >> (29 * k)) & ((1 << 29) - 1) for k in range(NL)]
        self.c = [lim(ci) for ci in C.c]
        cM = C.Mrns % Q
        plast = C.primes[-1]
        self.hoff = (plast - 1) // 2
        self.cMn = lim((Q - cM) % Q)
        self.w = [(1 << 58) // p for p in C.primes]
        self.B0, self.B1 = B & ((1 << 29) - 1), B >> 29
        # quotient by Q: X = S >> t, t = nq - 29; q_est = (X mq) >> 72, mq = floor(2^(t + 72) / Q) < 2^44
        self.t = self.nq - 29
        self.a = 29 * (NL - 1) - self.t            # top limb << a, next >> 29 - a, next >> 58 - a
        assert 0 <= self.a <= 28
        self.mq = (1 << (self.t + 72)) // Q
        assert self.mq < (1 << 44)
        # quotient by B: X2 = x >> t2, t2 = max(0, nq - 63); hq_est = (X2 mb) >> (nb + 51 - t2)
        self.t2 = max(0, self.nq - 63)
        self.mb = (1 << (self.nb + 51)) // B
        assert self.mb <= (1 << 52)
        self.sB = self.nb + 51 - self.t2 - 64
        assert self.sB >= 0 and self.t2 <= 31
        # randomised flatten (kernels.h k_crt_lean<.., RND>): the draws enter as r'_i = 2 xmax - r_i
        # >= 0 added to the old digits, and cR = c Q - 2 xmax (1 + B) >= 0 puts the constant back:
        #   S' = D + (hi_o + r'_1) B + (lo_o + r'_0) + cR  ==  x_old + D - r_0 - r_1 B   (mod Q)
        self.xmax = C.xmax
        two = 2 * self.xmax * (1 + B)
        self.cRv = (-two) % Q
        self.cR = lim(self.cRv)

    @staticmethod
    def mulhi64(x, m):
        """(x m) >> 64 for x < 2^64 and m = m1 2^32 + m0 as the device forms it."""
        x0, x1, m0, m1 = x & 0xFFFFFFFF, x >> 32, m & 0xFFFFFFFF, m >> 32
        t0 = (x0 * m0) >> 32
        t1 = u64(x0 * m1 + t0)
        t2 = u64(x1 * m0 + t1)
        return u64(x1 * m1 + (t2 >> 32))

    def digits_random(self, y, e_lo, e_hi, r0, r1):
        """The randomised flatten through the same limb sums: old stored digits (e_lo, e_hi), draws
        r_i in [0, 2 xmax]; returns the new stored digits."""
        lo, hq, alpha = self.digits(y, e_lo + 2 * self.xmax - r0, e_hi + 2 * self.xmax - r1, rnd=True)
        return lo + r0, hq + r1, alpha

    def digits(self, y, lo_o, hi_o, rnd=False):
        C, NL = self.C, self.NL
        assert lo_o < (1 << 64) and hi_o < (1 << 64)
        acc = 0
        for i in range(C.npr):
            acc = u64(acc + y[i] * self.w[i])
        alpha = acc >> 58
        L = [0] * NL
        L[0] = lo_o
        yl = list(y)
        yl[-1] -= self.hoff                  # takes H' out again
        assert yl[-1] >= 0
        for k in range(NL):
            for i in range(C.npr):
                L[k] = u64(L[k] + yl[i] * self.c[i][k])
            L[k] = u64(L[k] + alpha * self.cMn[k])
            if rnd:
                L[k] = u64(L[k] + self.cR[k])
        h0, h1 = hi_o & 0xFFFFFFFF, hi_o >> 32
        L[0] = u64(L[0] + h0 * self.B0)
        L[1] = u64(L[1] + h0 * self.B1)
        L[1] = u64(L[1] + (u64(h1 * self.B0) << 3))
        if NL >= 3:
            L[2] = u64(L[2] + (u64(h1 * self.B1) << 3))
        else:
            assert h1 * self.B1 == 0
        S = sum(L[k] << (29 * k) for k in range(NL))
        # quotient estimate from the top limbs
        X = u64(L[NL - 1] << self.a) + (L[NL - 2] >> (29 - self.a))
        if NL >= 3:
            X += L[NL - 3] >> (58 - self.a)
        X = u64(X)
        q = self.mulhi64(X, self.mq) >> 8
        assert S // C.Q - 1 <= q <= S // C.Q, "quotient estimate"
        # x = S - q Q modulo 2^96, in [0, 2 Q)
        x = (S - q * C.Q) & ((1 << 96) - 1)
        assert x == S - q * C.Q and x < 2 * C.Q
        if x >= C.Q:
            x -= C.Q
        X2 = x >> self.t2
        assert X2 < (1 << 63)
        hq = self.mulhi64(X2, self.mb) >> self.sB
        assert x // C.B - 1 <= hq <= x // C.B, "digit estimate"
        lo = (x - hq * C.B) & M64
        assert lo == x - hq * C.B and lo < 2 * C.B
        if lo >= C.B:
            lo -= C.B
            hq += 1
        return lo, hq, alpha


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

def chacha_block(key_words, c12, c13, c14, c15, rounds=8):
    """chacha_block<ROUNDS> of kernels.h (RFC 8439 block function): 16 output words."""
    s = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + [int(k) & MASK32 for k in key_words] + \
        [int(c) & MASK32 for c in (c12, c13, c14, c15)]
    x = list(s)

    def rotl(v, n):
        return ((v << n) | (v >> (32 - n))) & MASK32

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & MASK32; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & MASK32; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & MASK32; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & MASK32; x[b] = rotl(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(x[i] + s[i]) & MASK32 for i in range(16)]


_rnd_cache = {}


def rnd128(key, ctr):
    """rnd128 of kernels.h: the four words of the draw addressed by ctr = (x, y, z, w); key = the
    seed of sgfhe_set_random_flatten (an int: 32 little-endian bytes).  Coefficients 4 q .. 4 q + 3
    share block q."""
    x, y, z, w = (int(v) & MASK32 for v in ctr)
    tag = (key, x >> 2, y, z, w)
    if tag not in _rnd_cache:
        _rnd_cache.clear()
        _rnd_cache[tag] = chacha_block([(key >> (32 * i)) & MASK32 for i in range(8)], x >> 2, y, z, w)
    blk = _rnd_cache[tag]
    return blk[4 * (x & 3):4 * (x & 3) + 4]


class EngineModel:
    def __init__(self, n, m, Q, B, DQ_tilde, random_flatten=False):
        self.C = Consts(n, m, Q, B, DQ_tilde, random_flatten)
        self.ntt = NttModel(self.C.logm)

    def limbs_mod_p(self, vals, P):
        """limbs_mod_p: canonical residues (Python ints) -> centred-lifted residues mod p."""
        C = self.C
        v = [int(x) for x in vals]
        c0 = np.array([x & MASK32 for x in v], dtype=np.int64)
        c1 = np.array([(x >> 32) & MASK32 for x in v], dtype=np.int64)
        c2 = np.array([(x >> 64) & MASK32 for x in v], dtype=np.int64)
        r = i32(sredc(c0 * P["r1"], P) + sredc(c1 * P["r2"], P) + sredc(c2 * P["r3"], P))
        lift = np.array([x > C.Q // 2 for x in v])
        return i32(np.where(lift, r - P["qmodp"], r))

    def key_transform(self, canon_poly, pi):
        """k_key_transform for one polynomial (list of ints in [0, Q)) and prime index."""
        P = self.C.pk[pi]
        vals = smont(self.limbs_mod_p(canon_poly, P), P["kappaR"], P)
        x = self.ntt.forward(self.ntt.to_regs(vals), P["twf"], P)
        return scentre(sred(sred_floor(x, P), P), P).reshape(-1)   # slot E tid + e, centred

    def extprod(self, dig_a, dig_b, keyslice, j, plain=False, random=False):
        """k_extprod for one bootstrap: dig_* lists of (lo, hi) stored digits; keyslice[pi][row*2+col]
        slot arrays; returns y[c][pi] arrays (natural order, in [0, p))."""
        C = self.C
        NPR = C.npr
        M, T = C.M, self.ntt.T
        ys = [[None] * NPR for _ in range(2)]
        for pi in range(NPR):
            P = C.pk[pi]
            p = P["p"]
            sR = P["sRr"] if random else P["sR"]
            digs = [np.array([d[0] for d in dig_a], dtype=np.int64),
                    np.array([d[1] for d in dig_a], dtype=np.int64),
                    np.array([d[0] for d in dig_b], dtype=np.int64),
                    np.array([d[1] for d in dig_b], dtype=np.int64)]
            U = []
            for d in digs:
                v = i32(sredc(d, P) + sR)                        # digit_reduce
                assert int(np.max(np.abs(v))) <= p + (1 << 16)
                U.append(self.ntt.forward(self.ntt.to_regs(v), P["twf"], P).reshape(-1))
            for c in range(2):
                if c == 0:
                    # column 0: 64-bit accumulation over the four phases, one reduction
                    acc = sum(U[row] * keyslice[pi][row * 2] for row in range(4))
                    z = sredc(acc, P)
                    assert int(np.max(np.abs(z))) < 1.5 * 2 ** 29
                    if not self.ntt.wide_ok():
                        z = sred(z, P)
                else:
                    # column 1: Montgomery product per phase (|.| < 0.72 * 2^29), summed in the LDS
                    # accumulator over the four phases, one sred
                    z = np.zeros(M, dtype=np.int64)
                    for row in range(4):
                        z = i32(z + smont(U[row], keyslice[pi][row * 2 + 1], P))
                    z = sred(z, P)
                z = self.ntt.inverse(z.reshape(T, self.ntt.E), P["twi"], P, wide=(c == 0 and self.ntt.wide_ok()))
                Pn = self.ntt.from_regs(z)                         # natural order, |.| < 1.4 * 2^29
                if plain:
                    ys[c][pi] = (sfull(Pn, P) + P["hoff"]) % p
                    continue
                i = np.arange(M, dtype=np.int64)
                s = (i - j) & (2 * M - 1)
                v = Pn[s & (M - 1)]
                v = np.where((s & M) != 0, -v, v)
                y = i32(v - Pn) + 3 * p + P["hoff"]               # non-canonical, non-negative
                assert int(y.min()) > 0 and int(y.max()) < (1 << 32)
                ys[c][pi] = y
        return ys

    def crt_value(self, y, x_old=0):
        """crt_reduce for one coefficient: residues y[pi] -> (sum y c + T[alpha] + x_old) mod Q."""
        C = self.C
        f = np.float32(0)
        for pi in range(C.npr):
            f = np.float32(f + np.float32(y[pi]) * np.float32(np.float32(1.0) / np.float32(C.primes[pi])))
        alpha = int(f)
        self.last_alpha = alpha
        return (C.T[alpha] + sum(y[pi] * C.c[pi] for pi in range(C.npr)) + x_old) % C.Q

    def random_digits(self, xn, key, ctr):
        """random_digits of kernels.h: stored digits e_i = u_i + s + xmax of the randomised flatten of
        the accumulator whose shifted value is xn = (acc + (s + xmax)(1 + B)) mod Q."""
        C = self.C
        rv = rnd128(key, ctr)
        span = 2 * C.xmax + 1
        r0 = (((rv[1] << 32) | rv[0]) * span) >> 64
        r1 = (((rv[3] << 32) | rv[2]) * span) >> 64
        x2 = (xn - r0 - r1 * C.B) % C.Q
        return x2 % C.B + r0, x2 // C.B + r1

    def crt_acc(self, ys_c, dig_old, noacc=False, canon=False, rnd=None):
        """k_crt_acc for one polynomial: ys_c[pi] arrays -> new digits (or canonical values).
        rnd = (seed, c, iter, bootstrap, call) selects the randomised flatten."""
        C = self.C
        out = []
        for i in range(C.M):
            y = [int(ys_c[pi][i]) for pi in range(C.npr)]
            xn = self.crt_value(y, 0 if noacc else dig_old[i][1] * C.B + dig_old[i][0])
            if canon:
                out.append(xn)
            elif rnd is not None:
                seed, c, it, boot, call = rnd
                out.append(self.random_digits(xn, seed, ((c << C.logm) + i, it, boot, call)))
            else:
                out.append((xn % C.B, xn // C.B))
        return out

    def acc_from_digits(self, dig, random=False):
        C = self.C
        off = C.off_rnd if random else C.off
        return [(d[1] * C.B + d[0] - off) % C.Q for d in dig]
