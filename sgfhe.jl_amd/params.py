"""Scheme parameters: host mirror of `Params` of SGFHE.jl (/root/reference/src/fhe.jl:27-99)."""

_SMALL_PRIMES = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
_MR_BASES = _SMALL_PRIMES + (41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107,
                             109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167, 173)


def isprime(x):
    """Strong-probable-prime test to 40 fixed bases (deterministic far beyond 2^64); the role
    of Primes.isprime at src/utils.jl:19."""
    if x < 2:
        return False
    for p in _SMALL_PRIMES:
        if x % p == 0:
            return x == p
    d, s = x - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in _MR_BASES:
        if a % x == 0:
            continue
        y = pow(a, d, x)
        if y == 1 or y == x - 1:
            continue
        for _ in range(s - 1):
            y = y * y % x
            if y == x - 1:
                break
        else:
            return False
    return True


def find_modulus(n, qmin, qmax=None):
    """src/utils.jl:7-28: the smallest prime q >= qmin (q <= qmax) with n | q - 1."""
    j = -((-(qmin - 1)) // n)
    while True:
        q = j * n + 1
        if qmax is not None and q > qmax:
            raise ValueError("Could not find a modulus between %d and %d" % (qmin, qmax))
        if isprime(q):
            return q
        j += 1


class Params:
    """`Params(n)` (src/fhe.jl:43-97).  Field names follow the reference struct
    (src/fhe.jl:27-41); `ell` is the decomposition length fixed at src/fhe.jl:576."""

    __slots__ = ("n", "r", "q", "Q", "t", "m", "B", "Dr", "Dq", "DQ_tilde", "ell", "rlwe_type", "mod_repr")

    # widths of the unsigned types the reference accepts for `rlwe_type` (src/fhe.jl:21-22: `UInt` or
    # `MLUInt` of a large enough size; test/performance.test.jl:32,59,86 use MLUInt{2, UInt64} etc.)
    _MOD_REPRS = (None, "ModUInt", "MgModUInt")

    @staticmethod
    def _type_bits(rlwe_type):
        """Bits of an unsigned type given as a width (64, 128, ...) or by its Julia name: "UInt64",
        "UInt128", "MLUInt{N, UIntK}" (N limbs of K bits)."""
        import re
        if isinstance(rlwe_type, int):
            return rlwe_type
        m = re.fullmatch(r"UInt(\d+)", rlwe_type)
        if m:
            return int(m.group(1))
        m = re.fullmatch(r"MLUInt\{\s*(\d+)\s*,\s*UInt(\d+)\s*\}", rlwe_type)
        if m:
            return int(m.group(1)) * int(m.group(2))
        raise AssertionError("rlwe_type: an unsigned integer type (width, \"UInt128\", \"MLUInt{2, UInt64}\", ...)")

    def __init__(self, n, rlwe_type=None, mod_repr=None):
        """`rlwe_type` / `mod_repr` are the reference's keyword arguments (src/fhe.jl:43-47,71-85): the
        integer type and the residue representation (ModUInt or MgModUInt) of the bootstrap ring.  They
        change how the REFERENCE stores residues, not their values: this engine takes and returns
        canonical residues as 16-byte words whatever they are, so they are validated exactly as the
        reference validates them and recorded, nothing else."""
        if n < 64 or n & (n - 1):
            raise AssertionError("n must be a power of 2, >= 64 (src/fhe.jl:45-46)")
        if mod_repr not in self._MOD_REPRS:
            raise AssertionError("mod_repr must be None, \"ModUInt\" or \"MgModUInt\" (src/fhe.jl:47)")
        r = 16 * n                                                    # src/fhe.jl:53
        q = find_modulus(2 * n, r * n)                                # src/fhe.jl:57
        m = r // 2                                                    # src/fhe.jl:62
        Q = find_modulus(2 * m, r ** 4 * n ** 2 * 1220, r ** 4 * n ** 2 * 1225)   # :64-69
        if rlwe_type is None:                                         # src/fhe.jl:71-78
            if Q.bit_length() <= 64:
                rlwe_type = "UInt64"
            elif Q.bit_length() <= 128:
                rlwe_type = "UInt128"
            else:
                raise ValueError("n=%d is too large" % n)             # src/fhe.jl:77
        elif not self._type_bits(rlwe_type) > Q.bit_length() - 1:     # sizeof(rlwe_type) * 8 > log2(Q), :80
            raise AssertionError("rlwe_type is too narrow for Q (src/fhe.jl:80)")
        self._set(n, r, q, Q, m, r * r * n * 35, q // 4, Q // 8)      # src/fhe.jl:87-90
        self.rlwe_type = rlwe_type
        self.mod_repr = mod_repr or "MgModUInt"                       # src/fhe.jl:83-85

    def _set(self, n, r, q, Q, m, B, Dq, DQ_tilde):
        self.n, self.r, self.q, self.Q, self.m, self.B = n, r, q, Q, m, B
        self.t = r.bit_length() - 2                                   # log2(r) - 1, src/fhe.jl:61
        self.Dr, self.Dq, self.DQ_tilde, self.ell = r // 4, Dq, DQ_tilde, 2
        self.rlwe_type, self.mod_repr = ("UInt64" if Q.bit_length() <= 64 else "UInt128"), "MgModUInt"

    @classmethod
    def custom(cls, n, Q, B, DQ_tilde=None):
        """Synthetic parameter set with the structure of Params(n) (r = 16 n, m = r / 2, ell = 2)
        but a caller-chosen bootstrap modulus Q and gadget base B (BASELINE.json configs 3, 4)."""
        if n < 8 or n & (n - 1):
            raise AssertionError("n must be a power of 2, >= 8")
        if B * B < Q:
            raise AssertionError("B^2 >= Q required (src/utils.jl:145)")
        self = object.__new__(cls)
        self._set(n, 16 * n, 0, Q, 8 * n, B, 0, Q // 8 if DQ_tilde is None else DQ_tilde)
        return self

    def __repr__(self):
        return "Params(" + ", ".join("%s=%s" % (k, getattr(self, k)) for k in self.__slots__) + ")"

    def __eq__(self, other):
        return isinstance(other, Params) and all(
            getattr(self, k) == getattr(other, k) for k in self.__slots__[:11])   # values, not representation

    def __hash__(self):
        return hash(tuple(getattr(self, k) for k in self.__slots__[:11]))
