"""ctypes face of the host-side ciphertext plumbing in libsgfhe_hip.so (`sgfhe_host_*`,
include/sgfhe_hip.h; csrc/host_plumbing.h): the steps either side of the bootstrap path
 as C++ behind the C ABI -- extract / split_ciphertext, private
encryption given its random draws, the space-optimal form, decryption.  Pure functions, no device.
`scheme.py` holds the same steps in numpy; tests/test_host_plumbing.py compares the two bit for bit.
Paths in docstrings are relative to /root/reference."""

import ctypes

import numpy as np

from . import _lib


def _params(p):
    w = lambda x: (ctypes.c_uint64 * 2)(x & 0xFFFFFFFFFFFFFFFF, x >> 64)
    return _lib.SgfheParams(p.n, p.r, p.m, p.ell, w(p.Q), w(p.B), w(p.DQ_tilde))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _chk(rc):
    if rc:
        raise ValueError("sgfhe_host call failed with status %d (bad argument)" % rc)


def _arr(x, dtype):
    return np.ascontiguousarray(x, dtype=dtype)


def deterministic_expand(p, u):
    """deterministic_expand(params, u) (src/fhe.jl:304-307), SHAKE-256 based."""
    u = _arr(u, np.uint8)
    a = np.zeros(p.n, dtype=np.uint64)
    _chk(_lib.lib().sgfhe_host_deterministic_expand(ctypes.byref(_params(p)), _ptr(u), _ptr(a)))
    return a


def encrypt_private(p, sk, u, w, message):
    """_encrypt_private (src/fhe.jl:310-328) given the draws u (n bits) and w (n values in
    [-Dr/8, Dr/8]); returns the RLWE (a, b)."""
    sk, u, w, message = _arr(sk, np.uint64), _arr(u, np.uint8), _arr(w, np.int64), _arr(message, np.uint8)
    a, b = np.zeros(p.n, dtype=np.uint64), np.zeros(p.n, dtype=np.uint64)
    _chk(_lib.lib().sgfhe_host_encrypt_private(ctypes.byref(_params(p)), _ptr(sk), _ptr(u), _ptr(w),
                                               _ptr(message), _ptr(a), _ptr(b)))
    return a, b


def pack_private(p, b):
    """v of encrypt_optimal(key::PrivateKey, ...) (src/fhe.jl:339-345): (5, n) bits."""
    b = _arr(b, np.uint64)
    v = np.zeros((5, p.n), dtype=np.uint8)
    _chk(_lib.lib().sgfhe_host_pack_private(ctypes.byref(_params(p)), _ptr(b), _ptr(v)))
    return v


def normalize_private(p, u, v):
    """normalize_ciphertext(::PrivateEncryptedCiphertext) (src/fhe.jl:354-359)."""
    u, v = _arr(u, np.uint8), _arr(v, np.uint8)
    a, b = np.zeros(p.n, dtype=np.uint64), np.zeros(p.n, dtype=np.uint64)
    _chk(_lib.lib().sgfhe_host_normalize_private(ctypes.byref(_params(p)), _ptr(u), _ptr(v), _ptr(a), _ptr(b)))
    return a, b


def split_ciphertext(p, a, b):
    """split_ciphertext (src/fhe.jl:287-290): RLWE of length n or m -> (lwe_a [n][n], lwe_b [n])."""
    a, b = _arr(a, np.uint64), _arr(b, np.uint64)
    la, lb = np.zeros((p.n, p.n), dtype=np.uint64), np.zeros(p.n, dtype=np.uint64)
    _chk(_lib.lib().sgfhe_host_split_ciphertext(ctypes.byref(_params(p)), _ptr(a), _ptr(b), len(a),
                                                _ptr(la), _ptr(lb)))
    return la, lb


def decrypt_lwe(p, sk, lwe_a, lwe_b):
    """decrypt(key, ::EncryptedBit) (src/fhe.jl:504-507) over a batch."""
    sk, la, lb = _arr(sk, np.uint64), _arr(lwe_a, np.uint64).reshape(-1, p.n), _arr(lwe_b, np.uint64).reshape(-1)
    bits = np.zeros(len(lb), dtype=np.uint8)
    _chk(_lib.lib().sgfhe_host_decrypt_lwe(ctypes.byref(_params(p)), _ptr(sk), _ptr(la), _ptr(lb), len(lb),
                                           _ptr(bits)))
    return bits.astype(bool)


def decrypt_rlwe(p, sk, a, b):
    """decrypt(key, ::Union{Ciphertext, PackedCiphertext}) (src/fhe.jl:471-494)."""
    sk, a, b = _arr(sk, np.uint64), _arr(a, np.uint64), _arr(b, np.uint64)
    bits = np.zeros(p.n, dtype=np.uint8)
    _chk(_lib.lib().sgfhe_host_decrypt_rlwe(ctypes.byref(_params(p)), _ptr(sk), _ptr(a), _ptr(b), len(a),
                                            _ptr(bits)))
    return bits.astype(bool)


# ---- the public-key side ------------------------------------------

def public_key(p, sk, k0, e):
    """PublicKey(rng, sk) (src/fhe.jl:146-168) given its draws k0 in [0, q) and the centred noise e:
    returns k1 = k0 s + e over Z_q."""
    sk, k0, e = _arr(sk, np.uint64), _arr(k0, np.uint64), _arr(e, np.int64)
    k1 = np.zeros(p.n, dtype=np.uint64)
    _chk(_lib.lib().sgfhe_host_public_key(ctypes.byref(_params(p)), p.q, _ptr(sk), _ptr(k0), _ptr(e), _ptr(k1)))
    return k1


def encrypt_public(p, k0, k1, u, w1, w2, message):
    """_encrypt_public (src/fhe.jl:386-409) given its draws u in {-1, 0, 1}, w1, w2 -> RLWE (a, b)."""
    k0, k1 = _arr(k0, np.uint64), _arr(k1, np.uint64)
    u, w1, w2, msg = _arr(u, np.int8), _arr(w1, np.int64), _arr(w2, np.int64), _arr(message, np.uint8)
    a, b = np.zeros(p.n, dtype=np.uint64), np.zeros(p.n, dtype=np.uint64)
    _chk(_lib.lib().sgfhe_host_encrypt_public(ctypes.byref(_params(p)), p.q, _ptr(k0), _ptr(k1), _ptr(u),
                                              _ptr(w1), _ptr(w2), _ptr(msg), _ptr(a), _ptr(b)))
    return a, b


def pack_public(p, a, b):
    """The bit matrices of encrypt_optimal(key::PublicKey, ...) (src/fhe.jl:420-436):
    a_bits [t + 1][n], b_bits [6][n]."""
    a, b = _arr(a, np.uint64), _arr(b, np.uint64)
    a_bits, b_bits = np.zeros((p.t + 1, p.n), dtype=np.uint8), np.zeros((6, p.n), dtype=np.uint8)
    _chk(_lib.lib().sgfhe_host_pack_public(ctypes.byref(_params(p)), _ptr(a), _ptr(b), _ptr(a_bits), _ptr(b_bits)))
    return a_bits, b_bits


def normalize_public(p, a_bits, b_bits):
    """normalize_ciphertext(::PublicEncryptedCiphertext) (src/fhe.jl:444-449)."""
    a_bits, b_bits = _arr(a_bits, np.uint8), _arr(b_bits, np.uint8)
    a, b = np.zeros(p.n, dtype=np.uint64), np.zeros(p.n, dtype=np.uint64)
    _chk(_lib.lib().sgfhe_host_normalize_public(ctypes.byref(_params(p)), _ptr(a_bits), _ptr(b_bits), _ptr(a), _ptr(b)))
    return a, b
