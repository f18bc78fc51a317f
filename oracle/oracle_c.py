"""
TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

ctypes binding of oracle/libsgfhe_oracle.so (the C restatement of the reference path, see
sgfhe_oracle.h).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  128-bit residues are numpy uint64 arrays with a trailing axis of 2
({lo, hi}).
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsgfhe_oracle.so")

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "sgfhe_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsgfhe_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.sgo_ctx_create.restype = ctypes.c_void_p
        L.sgo_ctx_create.argtypes = [_u64p]
        L.sgo_ctx_destroy.argtypes = [ctypes.c_void_p]
        L.sgo_ctx_uses_ntt.argtypes = [ctypes.c_void_p]
        L.sgo_ctx_set_rns2.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
        L.sgo_ctx_uses_rns2.argtypes = [ctypes.c_void_p]
        L.sgo_find_modulus.argtypes = [ctypes.c_uint64, _u64p, _u64p, _u64p]
        L.sgo_params_make.argtypes = [ctypes.c_uint64, _u64p]
        L.sgo_rescale.argtypes = [_u64p, _u64p, _u64p, ctypes.c_int, _u64p]
        L.sgo_flatten.argtypes = [ctypes.c_void_p, _u64p, _u64p]
        L.sgo_poly_mul.argtypes = [ctypes.c_void_p, _u64p, _u64p, _u64p]
        L.sgo_poly_mul_schoolbook.argtypes = [ctypes.c_void_p, _u64p, _u64p, _u64p]
        L.sgo_external_product.argtypes = [ctypes.c_void_p, _u64p, _u64p, _u64p, _u64p, _u64p]
        L.sgo_private_key.argtypes = [ctypes.c_void_p, ctypes.c_uint64, _u64p]
        L.sgo_bootstrap_key.argtypes = [ctypes.c_void_p, _u64p, ctypes.c_char_p, ctypes.c_uint64,
                                        _u64p, ctypes.c_int]
        L.sgo_lwe_encrypt_bits.argtypes = [ctypes.c_void_p, _u64p, _u8p, ctypes.c_size_t,
                                           ctypes.c_uint64, _u64p, _u64p]
        L.sgo_lwe_decrypt_bit.argtypes = [ctypes.c_void_p, _u64p, _u64p, ctypes.c_uint64]
        L.sgo_bootstrap_batch.argtypes = [ctypes.c_void_p, _u64p, _u64p, _u64p, _u64p, _u64p,
                                          ctypes.c_size_t, _u64p, ctypes.c_int, ctypes.c_uint64,
                                          _u64p, ctypes.c_int]
        L.sgo_key_transform.argtypes = [ctypes.c_void_p, _u64p, _u64p, ctypes.c_int]
        L.sgo_bootstrap_batch_opt.argtypes = L.sgo_bootstrap_batch.argtypes
        L.sgo_pack_encrypted_bits.argtypes = [ctypes.c_void_p, _u64p, _u64p, _u64p, _u64p, _u64p,
                                              ctypes.c_int]
        L.sgo_pack_encrypted_bits_ex.argtypes = [ctypes.c_void_p, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p,
                                                 ctypes.c_int, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32]
        L.sgo_flatten_random.argtypes = [ctypes.c_void_p, _u64p, ctypes.c_int64, ctypes.c_int64, _u64p]
        L.sgo_flatten_draws.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint, ctypes.c_uint32,
                                        ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int64)]
        L.sgo_bootstrap_batch_rnd.argtypes = [ctypes.c_void_p, ctypes.c_int, _u64p, _u64p, _u64p, _u64p,
                                              _u64p, ctypes.c_size_t, _u64p, ctypes.c_int,
                                              ctypes.c_uint64, _u64p, ctypes.c_int, ctypes.c_char_p,
                                              ctypes.c_uint32, ctypes.c_uint32,
                                              ctypes.POINTER(ctypes.c_uint32)]
        _lib = L
    return _lib


def _p(arr):
    return arr.ctypes.data_as(_u64p)


def seed_bytes(seed):
    """Key-generation seed: 32 bytes, or an int taken as 32 little-endian bytes."""
    if isinstance(seed, (bytes, bytearray)):
        if len(seed) != 32:
            raise ValueError("key seed must be 32 bytes")
        return bytes(seed)
    return int(seed).to_bytes(32, "little")


def to_words(x):
    """Python int -> [lo, hi]."""
    return [x & 0xFFFFFFFFFFFFFFFF, (x >> 64) & 0xFFFFFFFFFFFFFFFF]


def ints_to_u128(vals):
    """Sequence of Python ints -> uint64 array [..., 2]."""
    out = np.empty((len(vals), 2), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i, 0] = v & 0xFFFFFFFFFFFFFFFF
        out[i, 1] = v >> 64
    return out


def u128_to_ints(arr):
    flat = np.ascontiguousarray(arr).reshape(-1, 2)
    return [int(lo) | (int(hi) << 64) for lo, hi in flat]


def params_words(n, r, m, ell, Q, B, DQ_tilde):
    return np.array([n, r, m, ell] + to_words(Q) + to_words(B) + to_words(DQ_tilde),
                    dtype=np.uint64)


def params_make(n):
    """fhe.jl:43-97 through the C restatement; returns dict."""
    w = np.zeros(10, dtype=np.uint64)
    rc = lib().sgo_params_make(n, _p(w))
    if rc:
        raise ValueError("sgo_params_make(%d) failed: %d" % (n, rc))
    return dict(n=int(w[0]), r=int(w[1]), m=int(w[2]), ell=int(w[3]),
                Q=int(w[4]) | (int(w[5]) << 64), B=int(w[6]) | (int(w[7]) << 64),
                DQ_tilde=int(w[8]) | (int(w[9]) << 64))


class Oracle:
    """One parameter set of the C restatement."""

    def __init__(self, n, r, m, Q, B, DQ_tilde, ell=2, rns2=None):
        """rns2 = (m1, m2): Q = m1 m2 held as RNS2Number limbs (src/rns.jl): products per limb."""
        self.n, self.r, self.m, self.Q, self.B, self.DQ_tilde, self.ell = n, r, m, Q, B, DQ_tilde, ell
        self._words = params_words(n, r, m, ell, Q, B, DQ_tilde)
        self._ctx = lib().sgo_ctx_create(_p(self._words))
        if not self._ctx:
            raise ValueError("sgo_ctx_create rejected the parameters")
        if rns2 is not None:
            rc = lib().sgo_ctx_set_rns2(self._ctx, int(rns2[0]), int(rns2[1]))
            if rc:
                raise ValueError("sgo_ctx_set_rns2 failed: %d" % rc)

    @classmethod
    def from_params(cls, p, rns2=None):
        """p: anything with n, r, m, Q, B, DQ_tilde attributes."""
        return cls(p.n, p.r, p.m, p.Q, p.B, p.DQ_tilde, rns2=rns2)

    @classmethod
    def make(cls, n):
        d = params_make(n)
        return cls(d["n"], d["r"], d["m"], d["Q"], d["B"], d["DQ_tilde"])

    def __del__(self):
        try:
            if self._ctx:
                lib().sgo_ctx_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    @property
    def uses_ntt(self):
        return bool(lib().sgo_ctx_uses_ntt(self._ctx))

    @property
    def uses_rns2(self):
        return bool(lib().sgo_ctx_uses_rns2(self._ctx))

    def flatten(self, a):
        ain = np.array(to_words(a), dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        lib().sgo_flatten(self._ctx, _p(ain), _p(out))
        return [int(out[0]) | (int(out[1]) << 64), int(out[2]) | (int(out[3]) << 64)]

    def flatten_random(self, a, x0, x1):
        """flatten(rng, a, Val(B), Val(2)) (utils.jl:198-241) with the two draws given."""
        ain = np.array(to_words(a), dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        lib().sgo_flatten_random(self._ctx, _p(ain), int(x0), int(x1), _p(out))
        return [int(out[0]) | (int(out[1]) << 64), int(out[2]) | (int(out[3]) << 64)]

    def flatten_draws(self, seed, c, y, boot=0, call=0):
        """[m][2] int64: the draws of accumulator c in the flatten tagged y (engine's ChaCha8 stream)."""
        d = np.zeros((self.m, 2), dtype=np.int64)
        lib().sgo_flatten_draws(self._ctx, seed_bytes(seed), c, y, boot, call,
                                d.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        return d

    def poly_mul(self, a, b, schoolbook=False):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.zeros((self.m, 2), dtype=np.uint64)
        fn = lib().sgo_poly_mul_schoolbook if schoolbook else lib().sgo_poly_mul
        fn(self._ctx, _p(a), _p(b), _p(out))
        return out

    def external_product(self, a, b, A):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        A = np.ascontiguousarray(A, dtype=np.uint64)
        assert A.shape == (4, 2, self.m, 2)
        ra = np.zeros((self.m, 2), dtype=np.uint64)
        rb = np.zeros((self.m, 2), dtype=np.uint64)
        lib().sgo_external_product(self._ctx, _p(a), _p(b), _p(A), _p(ra), _p(rb))
        return ra, rb

    def private_key(self, seed):
        sk = np.zeros(self.n, dtype=np.uint64)
        lib().sgo_private_key(self._ctx, seed, _p(sk))
        return sk

    def bootstrap_key(self, sk, seed, noise=None, threads=None):
        """[n][4][2][m][2] uint64 canonical residues (fhe.jl:181-201)."""
        bkey = np.zeros((self.n, 4, 2, self.m, 2), dtype=np.uint64)
        sk = np.ascontiguousarray(sk, dtype=np.uint64)
        noise = self.n if noise is None else int(noise)
        if not 0 <= noise < (1 << 30) or 2 * noise >= self.Q:   # the bound sgfhe_bkey_generate enforces
            raise ValueError("noise must be below 2^30 and below Q / 2")
        lib().sgo_bootstrap_key(self._ctx, _p(sk), seed_bytes(seed), noise,
                                _p(bkey), threads or os.cpu_count() or 1)
        return bkey

    def lwe_encrypt_bits(self, sk, bits, seed):
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        sk = np.ascontiguousarray(sk, dtype=np.uint64)
        a = np.zeros((len(bits), self.n), dtype=np.uint64)
        b = np.zeros(len(bits), dtype=np.uint64)
        lib().sgo_lwe_encrypt_bits(self._ctx, _p(sk), bits.ctypes.data_as(_u8p), len(bits), seed,
                                   _p(a), _p(b))
        return a, b

    def lwe_decrypt_bits(self, sk, a, b):
        sk = np.ascontiguousarray(sk, dtype=np.uint64)
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, self.n)
        b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1)
        return np.array([lib().sgo_lwe_decrypt_bit(self._ctx, _p(sk), _p(a[i]), int(b[i]))
                         for i in range(len(b))], dtype=np.uint8)

    def key_transform(self, bkey, threads=None):
        """NTT-domain key for bootstrap_batch(..., opt=True) (sgo_key_transform)."""
        bkey = np.ascontiguousarray(bkey, dtype=np.uint64)
        # (RNS2Number ring: the key limb-wise, [2][n][4][2][m] residues mod m_limb)
        khat = np.zeros(((2,) if self.uses_rns2 and not self.uses_ntt else ()) + bkey.shape, dtype=np.uint64)
        rc = lib().sgo_key_transform(self._ctx, _p(bkey), _p(khat), threads or os.cpu_count() or 1)
        if rc:
            raise RuntimeError("sgo_key_transform failed: %d" % rc)
        return khat

    def bootstrap_batch(self, bkey, a1, b1, a2, b2, raw=False, n_iters=None, want_acc=False,
                        threads=None, opt=False, rnd=None):
        """fhe.jl:559-621 over a batch.  Returns out ([batch][3][n+1] uint64, or [..][2] if raw)
        and, if want_acc, the accumulators [batch][2][m][2] after `n_iters` iterations.
        opt=True: `bkey` is the NTT-domain key of key_transform and the k-loop runs in the GPU
        path's algebra (4 + 2 NTTs per iteration, BASELINE.md `cpu_opt`); same outputs.
        rnd=(seed, call[, boot0]): bootstrap(bkey, rng, ...) -- the randomised flatten
        (utils.jl:198-241) on the engine's ChaCha8 stream keyed with `seed` (32 bytes or an int);
        row t of the batch draws as bootstrap boot0 + t of call `call`; boot0 may be an array of
        one index per row (rows picked out of a larger call)."""
        bkey = np.ascontiguousarray(bkey, dtype=np.uint64)
        a1 = np.ascontiguousarray(a1, dtype=np.uint64).reshape(-1, self.n)
        a2 = np.ascontiguousarray(a2, dtype=np.uint64).reshape(-1, self.n)
        b1 = np.ascontiguousarray(b1, dtype=np.uint64).reshape(-1)
        b2 = np.ascontiguousarray(b2, dtype=np.uint64).reshape(-1)
        batch = a1.shape[0]
        shape = (batch, 3, self.n + 1, 2) if raw else (batch, 3, self.n + 1)
        out = np.zeros(shape, dtype=np.uint64)
        acc = np.zeros((batch, 2, self.m, 2), dtype=np.uint64) if want_acc else None
        nthreads = threads or min(batch, os.cpu_count() or 1)
        niter = self.n if n_iters is None else n_iters
        if rnd is not None:
            seed, call = rnd[0], rnd[1]
            boot0 = rnd[2] if len(rnd) > 2 else 0
            boots = None
            if not np.isscalar(boot0):
                boots = np.ascontiguousarray(boot0, dtype=np.uint32)
                assert boots.shape == (batch,)
                boot0 = 0
            rc = lib().sgo_bootstrap_batch_rnd(self._ctx, 1 if opt else 0, _p(bkey), _p(a1), _p(b1), _p(a2),
                                               _p(b2), batch, _p(out), 1 if raw else 0, niter,
                                               _p(acc) if want_acc else None, nthreads, seed_bytes(seed),
                                               call, int(boot0),
                                               boots.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
                                               if boots is not None else None)
        else:
            fn = lib().sgo_bootstrap_batch_opt if opt else lib().sgo_bootstrap_batch
            rc = fn(self._ctx, _p(bkey), _p(a1), _p(b1), _p(a2), _p(b2), batch, _p(out), 1 if raw else 0,
                    niter, _p(acc) if want_acc else None, nthreads)
        if rc:
            raise RuntimeError("sgo_bootstrap_batch failed: %d" % rc)
        return (out, acc) if want_acc else out


def _pack(self, bkey, a, b, threads=None, khat=None, rnd=None):
    """fhe.jl:660-696: n LWEs (a [n][n], b [n]) -> RLWE (w, v) over Z_r, [m] each.
    khat: the NTT-domain key (key_transform) for the n bootstraps (same bytes, faster).
    rnd = (seed, ct, call): pack_encrypted_bits(bkey, rng, ...) on the engine's ChaCha8 stream, this
    ciphertext being number `ct` of call `call`."""
    bkey = np.ascontiguousarray(bkey, dtype=np.uint64)
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(self.n, self.n)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(self.n)
    w = np.zeros(self.m, dtype=np.uint64)
    v = np.zeros(self.m, dtype=np.uint64)
    if khat is not None:
        khat = np.ascontiguousarray(khat, dtype=np.uint64)
    rc = lib().sgo_pack_encrypted_bits_ex(self._ctx, _p(bkey), _p(khat) if khat is not None else None, _p(a),
                                          _p(b), _p(w), _p(v), threads or os.cpu_count() or 1,
                                          seed_bytes(rnd[0]) if rnd is not None else None,
                                          int(rnd[1]) if rnd is not None else 0,
                                          int(rnd[2]) if rnd is not None else 0)
    if rc:
        raise RuntimeError("sgo_pack_encrypted_bits failed: %d" % rc)
    return w, v


Oracle.pack_encrypted_bits = _pack


def rescale(new_max, x, old_max, round_result):
    o = np.zeros(2, dtype=np.uint64)
    lib().sgo_rescale(_p(np.array(to_words(new_max), dtype=np.uint64)),
                      _p(np.array(to_words(x), dtype=np.uint64)),
                      _p(np.array(to_words(old_max), dtype=np.uint64)), int(round_result), _p(o))
    return int(o[0]) | (int(o[1]) << 64)
