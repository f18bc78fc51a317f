"""Thin Python handle on one `sgfhe_ctx` of libsgfhe_hip.so (include/sgfhe_hip.h).

Host arrays are numpy; 128-bit residues are uint64 arrays with a trailing axis of 2 ({lo, hi}).
Device-resident entry points take raw device pointers (e.g. `torch.Tensor.data_ptr()`); PyTorch
is only plumbing for device memory and torch.distributed (RCCL), never the compute path.
"""

import ctypes
import threading

import numpy as np

from . import _lib

FLAG_RAW_MODQ = 1
FLAG_RAW_RNS2 = 2
CTX_RANDOM_FLATTEN = 1          # accepted, without effect since ABI revision 6
CTX_DETERMINISTIC_ONLY = 2


class SgfheError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sgfhe_hip error %d: %s" % (code, msg))
        self.code = code


def _words(x):
    return (ctypes.c_uint64 * 2)(x & 0xFFFFFFFFFFFFFFFF, (x >> 64) & 0xFFFFFFFFFFFFFFFF)


def _c(arr, dtype=np.uint64):
    a = np.ascontiguousarray(arr, dtype=dtype)
    return a, a.ctypes.data_as(ctypes.c_void_p)


class Engine:
    """One bootstrap engine (ctx) for one parameter set on one HIP device."""

    def __init__(self, params, device=0, random_flatten=False, deterministic_only=False):
        """Both flatten modes (the `rng` argument of bootstrap / pack_encrypted_bits) are available on
        every engine: where the randomised one needs a prime more (Params(1024): six against five)
        the ctx keeps a basis and a key form per mode and `set_random_flatten` switches between them.
        deterministic_only=True keeps the smaller basis only (SGFHE_CTX_DETERMINISTIC_ONLY: no second
        key form; the randomised mode is then refused where it needs the extra prime).
        random_flatten is accepted for compatibility and has no effect."""
        self.params = params
        self.device = device
        # Held around every C call together with the read of its error string, and by the scheme
        # layer from the choice of the flatten mode to the end of the call that uses it, so that
        # threads sharing an Engine cannot interleave between the two (the C library serialises the
        # individual calls; the pairs are this binding's to keep together).
        self.lock = threading.RLock()
        self._L = _lib.lib()
        sp = _lib.SgfheParams(params.n, params.r, params.m, params.ell, _words(params.Q),
                              _words(params.B), _words(params.DQ_tilde))
        h = ctypes.c_void_p()
        rc = self._L.sgfhe_ctx_create_ex(ctypes.byref(sp), device,
                                         (CTX_RANDOM_FLATTEN if random_flatten else 0) |
                                         (CTX_DETERMINISTIC_ONLY if deterministic_only else 0), ctypes.byref(h))
        self._h = h
        if rc != 0:
            msg = self._L.sgfhe_last_error_string(h).decode() if h else "ctx_create failed"
            if h:
                self._L.sgfhe_ctx_destroy(h)
                self._h = None
            raise SgfheError(rc, msg)

    def clone(self):
        """A second engine on the same device, parameter set and key for an independent caller
        (sgfhe_ctx_clone): shares the device key and constants, owns its work buffers and streams, so
        calls on the clone overlap on the device with calls on this engine (one clone per thread).
        While clones exist the shared key is read-only."""
        h = ctypes.c_void_p()
        with self.lock:
            rc = self._L.sgfhe_ctx_clone(self._h, ctypes.byref(h))
            if rc != 0:
                msg = self._L.sgfhe_last_error_string(h if h else self._h).decode()
                if h:
                    self._L.sgfhe_ctx_destroy(h)
                raise SgfheError(rc, msg)
        other = Engine.__new__(Engine)
        other.params = self.params
        other.device = self.device
        other.lock = threading.RLock()
        other._L = self._L
        other._h = h
        return other

    def set_coalesce(self, enable=True, req_max=32, gates_max=256, window_us=300):
        """Gathering of small bootstrap_batch calls across the engines that share this key (sgfhe_set_coalesce):
        on by default once clones exist; the setting belongs to the shared key."""
        self._call("sgfhe_set_coalesce", int(bool(enable)), req_max, gates_max, window_us)

    def coalesce_stats(self, reset=False):
        st = (ctypes.c_uint64 * 4)()
        self._call("sgfhe_coalesce_stats", st, int(reset))
        return dict(calls=int(st[0]), requests=int(st[1]), gates=int(st[2]), max_requests=int(st[3]))

    def close(self):
        if getattr(self, "_h", None):
            self._L.sgfhe_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        with self.lock:
            rc = getattr(self._L, name)(self._h, *args)
            if rc != 0:
                raise SgfheError(rc, self._L.sgfhe_last_error_string(self._h).decode())

    # ---- key ------------------------------------------------------------------------------
    def upload_key(self, canonical):
        """canonical: [n][4][2][m][2] uint64, value.(coeffs) of BootstrapKey.key (fhe.jl:176-201)."""
        p = self.params
        a, ptr = _c(canonical)
        if a.size != p.n * 8 * p.m * 2:
            raise ValueError("bootstrap key must hold n*4*2*m residues of 2 words")
        self._call("sgfhe_bkey_upload", ptr, a.size)

    def upload_key_rns2(self, pairs, m1, m2):
        """pairs: [n][4][2][m][2] uint64 (v1, v2) RNS2Number limbs (src/rns.jl:8-24)."""
        a, ptr = _c(pairs)
        self._call("sgfhe_bkey_upload_rns2", ptr, a.size, m1, m2)

    def rns2_convert(self, values, m1, m2, to_pairs):
        """src/rns.jl on the device: [..., 2] uint64 canonical {lo, hi} -> (x mod m1, x mod m2)
        (to_pairs=True, rns.jl:16-18) or limb pairs -> canonical (to_pairs=False, rns.jl:32-40)."""
        a, ptr = _c(values)
        out = np.zeros_like(a)
        self._call("sgfhe_rns2_convert", int(bool(to_pairs)), ptr, a.size // 2, m1, m2,
                                             out.ctypes.data_as(ctypes.c_void_p))
        return out

    def generate_key(self, sk_bits, seed, noise=None):
        """BootstrapKey(rng, sk) (fhe.jl:181-201) generated on the device from a 32-byte seed
        (ChaCha20 streams); an int is taken as 32 little-endian bytes (tests, benchmarks)."""
        a, ptr = _c(sk_bits)
        if not isinstance(seed, (bytes, bytearray)):
            seed = int(seed).to_bytes(32, "little")
        if len(seed) != 32:
            raise ValueError("key seed must be 32 bytes")
        self._call("sgfhe_bkey_generate", ptr, a.size, bytes(seed),
                                              self.params.n if noise is None else noise)

    def key_device_form_bytes(self):
        n = ctypes.c_size_t()
        self._call("sgfhe_bkey_device_form_bytes", ctypes.byref(n))
        return n.value

    def export_key_device_form(self, dst_device_ptr):
        self._call("sgfhe_bkey_export_device_form", ctypes.c_void_p(dst_device_ptr))

    def import_key_device_form(self, src_device_ptr):
        self._call("sgfhe_bkey_import_device_form", ctypes.c_void_p(src_device_ptr))

    # ---- bootstrap ----------------------------------------------------------------------------
    def set_chunk(self, chunk):
        self._call("sgfhe_set_chunk", chunk)

    def set_lanes(self, lanes):
        self._call("sgfhe_set_lanes", lanes)

    def set_small_batch_max(self, max_bootstraps):
        """Chunks of at most this many bootstraps use the small-batch form of the k-loop (0: never)."""
        self._call("sgfhe_set_small_batch_max", max_bootstraps)

    def set_random_flatten(self, enable, seed=0):
        """rng != nothing branch of flatten (utils.jl:198-241): the draws come from a ChaCha8 counter
        stream keyed with `seed` -- 32 bytes (sgfhe_set_random_flatten_key), or an int taken as 32
        little-endian bytes (tests, benchmarks; the convention of generate_key)."""
        if isinstance(seed, (bytes, bytearray)):
            if len(seed) != 32:
                raise ValueError("random-flatten key: 32 bytes")
            key = bytes(seed)
        else:
            key = (int(seed) % (1 << 256)).to_bytes(32, "little")
        self._call("sgfhe_set_random_flatten_key", int(bool(enable)), key)

    def _lwe_args(self, a1, b1, a2, b2):
        n = self.params.n
        a1, p1 = _c(a1)
        a2, p2 = _c(a2)
        b1, q1 = _c(b1)
        b2, q2 = _c(b2)
        a1 = a1.reshape(-1, n)
        a2 = a2.reshape(-1, n)
        batch = a1.shape[0]
        if a2.shape[0] != batch or b1.size != batch or b2.size != batch:
            raise ValueError("ragged LWE batch")
        keep = (a1, a2, b1, b2)
        return batch, (p1, q1, p2, q2), keep

    def bootstrap_batch(self, a1, b1, a2, b2, raw=False, rns2=False, out=None):
        """bootstrap(bkey, nothing, ., .) (fhe.jl:608-621) over a batch of LWE pairs.
        Returns [batch][3][n+1] uint64 (AND, OR, XOR; a then b), or [batch][3][n+1][2] residues
        mod Q with raw=True (_bootstrap_internal, fhe.jl:559-595): {lo, hi} of the canonical
        value, or with rns2=True the RNS2Number limb pair (v1, v2) (src/rns.jl:16-18).
        `out`: a C-contiguous uint64 array of that shape to write into (a caller in a loop keeps
        its result buffer and its already-touched pages)."""
        batch, (p1, q1, p2, q2), _keep = self._lwe_args(a1, b1, a2, b2)
        n = self.params.n
        raw = raw or rns2
        shape = (batch, 3, n + 1, 2) if raw else (batch, 3, n + 1)
        if out is None:
            out = np.zeros(shape, dtype=np.uint64)
        elif out.shape != shape or out.dtype != np.uint64 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("bootstrap_batch: out must be a C-contiguous uint64 array of shape %r" % (shape,))
        if batch:
            self._call("sgfhe_bootstrap_batch", p1, q1, p2, q2, batch, out.ctypes.data_as(ctypes.c_void_p),
                (FLAG_RAW_MODQ if raw else 0) | (FLAG_RAW_RNS2 if rns2 else 0))
        return out

    def bootstrap_batch_device(self, a1_ptr, b1_ptr, a2_ptr, b2_ptr, batch, out_ptr, raw=False,
                               stream=None):
        """Asynchronous, all buffers device-resident (raw pointers)."""
        self._call("sgfhe_bootstrap_batch_device", ctypes.c_void_p(a1_ptr), ctypes.c_void_p(b1_ptr), ctypes.c_void_p(a2_ptr),
            ctypes.c_void_p(b2_ptr), batch, ctypes.c_void_p(out_ptr),
            FLAG_RAW_MODQ if raw else 0, ctypes.c_void_p(stream or 0))

    def sync(self):
        self._call("sgfhe_sync")

    def pack_encrypted_bits(self, a, b):
        """pack_encrypted_bits (fhe.jl:660-696) for `count` groups of n LWEs: a [count][n][n],
        b [count][n] -> (w, v), each [count][m] uint64 over Z_r."""
        p = self.params
        a, pa = _c(a)
        b, pb = _c(b)
        if a.size % (p.n * p.n) or b.size * p.n != a.size:
            raise ValueError("pack_encrypted_bits: a is [count][n][n], b is [count][n]")
        count = a.size // (p.n * p.n)
        w = np.zeros((count, p.m), dtype=np.uint64)
        v = np.zeros((count, p.m), dtype=np.uint64)
        self._call("sgfhe_pack_encrypted_bits", pa, pb, count,
                                                    w.ctypes.data_as(ctypes.c_void_p),
                                                    v.ctypes.data_as(ctypes.c_void_p))
        return w, v

    # ---- parity / debug hooks ---------------------------------------------------------------
    def external_product(self, a, b, A):
        """external_product(nothing, a, b, A, Val(B), Val(2)) (fhe.jl:519-530)."""
        m = self.params.m
        a, pa = _c(a)
        b, pb = _c(b)
        A, pA = _c(A)
        if a.size != 2 * m or b.size != 2 * m or A.size != 16 * m:
            raise ValueError("external_product: a, b are [m][2]; A is [4][2][m][2]")
        ra = np.zeros((m, 2), dtype=np.uint64)
        rb = np.zeros((m, 2), dtype=np.uint64)
        self._call("sgfhe_external_product", pa, pb, pA,
                                                 ra.ctypes.data_as(ctypes.c_void_p),
                                                 rb.ctypes.data_as(ctypes.c_void_p))
        return ra, rb

    def debug_cmux(self, a, b, C, j):
        """One k-loop iteration on chosen operands (sgfhe_debug_cmux): a, b [m][2], C [4][2][m][2]
        -> (a, b) + (x^j - 1) sum_row flatten(a, b)_row (*) C[row]."""
        m = self.params.m
        a, pa = _c(a)
        b, pb = _c(b)
        C, pC = _c(C)
        if a.size != 2 * m or b.size != 2 * m or C.size != 16 * m:
            raise ValueError("debug_cmux: a, b are [m][2]; C is [4][2][m][2]")
        ra = np.zeros((m, 2), dtype=np.uint64)
        rb = np.zeros((m, 2), dtype=np.uint64)
        self._call("sgfhe_debug_cmux", pa, pb, pC, int(j), ra.ctypes.data_as(ctypes.c_void_p),
                                           rb.ctypes.data_as(ctypes.c_void_p))
        return ra, rb

    def debug_accumulators(self, a1, b1, a2, b2, n_iters):
        batch, (p1, q1, p2, q2), _keep = self._lwe_args(a1, b1, a2, b2)
        acc = np.zeros((batch, 2, self.params.m, 2), dtype=np.uint64)
        if batch:
            self._call("sgfhe_debug_accumulators", p1, q1, p2, q2, batch, n_iters,
                                                       acc.ctypes.data_as(ctypes.c_void_p))
        return acc

    def debug_digits(self, a1, b1, a2, b2, n_iters):
        """Stored digit planes after n_iters iterations: [batch][2][2][m] uint64 (sgfhe_debug_digits)."""
        batch, (p1, q1, p2, q2), _keep = self._lwe_args(a1, b1, a2, b2)
        dig = np.zeros((batch, 2, 2, self.params.m), dtype=np.uint64)
        if batch:
            self._call("sgfhe_debug_digits", p1, q1, p2, q2, batch, n_iters,
                                                 dig.ctypes.data_as(ctypes.c_void_p))
        return dig

    def debug_flatten(self, values):
        """Deterministic flatten_poly of two polynomials: values [2][m][2] uint64 canonical residues
        -> stored digits [2][2][m] (sgfhe_debug_flatten)."""
        m = self.params.m
        a, ptr = _c(values)
        if a.size != 4 * m:
            raise ValueError("debug_flatten: values is [2][m][2]")
        dig = np.zeros((2, 2, m), dtype=np.uint64)
        self._call("sgfhe_debug_flatten", ptr, dig.ctypes.data_as(ctypes.c_void_p))
        return dig

    def debug_ntt(self, prime_index, poly, inverse=False):
        x, px = _c(poly, np.uint32)
        out = np.zeros(self.params.m, dtype=np.uint32)
        self._call("sgfhe_debug_ntt", prime_index, int(inverse), px,
                                          out.ctypes.data_as(ctypes.c_void_p))
        return out

    def primes(self):
        cnt = ctypes.c_uint32()
        arr = (ctypes.c_uint32 * 8)()
        self._call("sgfhe_debug_primes", ctypes.byref(cnt), arr)
        return [int(arr[i]) for i in range(cnt.value)]

    # ---- measurement ----------------------------------------------------------------------------
    def timing_enable(self, on=True):
        self._call("sgfhe_timing_enable", int(on))

    def build_id(self):
        """sgfhe_build_id() of the loaded library: hash of the kernel sources it was compiled from
        (+ ablation flags)."""
        return self._L.sgfhe_build_id().decode()

    def kernel_names(self):
        """(external-product kernel, CRT kernel) of the k-loop in the present flatten mode, as a
        rocprofv3 kernel trace names them (sgfhe_kernel_names)."""
        a, b = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
        self._call("sgfhe_kernel_names", a, 64, b, 64)
        return a.value.decode(), b.value.decode()

    def release_host_staging(self):
        """Free the device and page-locked staging buffers bootstrap_batch keeps on the ctx."""
        self._call("sgfhe_release_host_staging")

    def timing_read(self, reset=True):
        st = (ctypes.c_double * 8)()
        self._call("sgfhe_timing_read", st, int(reset))
        return dict(extprod_ms=st[0], extprod_samples=int(st[1]), crt_ms=st[2],
                    crt_samples=int(st[3]), chunk=int(st[4]), call_ms=st[5], calls=int(st[6]),
                    call_batch=st[7])
