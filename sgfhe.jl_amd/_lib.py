"""ctypes loader of libsgfhe_hip.so (C ABI: include/sgfhe_hip.h).

The library is hand-written HIP for gfx950 and is the only compute path of this package: there
is no CPU fallback.  A missing library is built in-tree with hipcc; a failed build or load
raises.
"""

import ctypes
import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("SGFHE_HIP_LIB") or os.path.join(CSRC, "libsgfhe_hip.so")

ABI_VERSION = 7   # SGFHE_ABI_VERSION of include/sgfhe_hip.h this binding was written for

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u32p = ctypes.POINTER(ctypes.c_uint32)


class SgfheParams(ctypes.Structure):
    """struct sgfhe_params of include/sgfhe_hip.h."""
    _fields_ = [("n", ctypes.c_uint64), ("r", ctypes.c_uint64), ("m", ctypes.c_uint64),
                ("ell", ctypes.c_uint64), ("Q", ctypes.c_uint64 * 2), ("B", ctypes.c_uint64 * 2),
                ("DQ_tilde", ctypes.c_uint64 * 2)]


def source_hash():
    """Identity of the sources the library is compiled from: SHA-256 over csrc/{*.h, *.hip} in
    file-name order, then include/sgfhe_hip.h (what the Makefile embeds in the library as
    sgfhe_build_id()): a change of the ABI header alone makes the in-tree library stale too."""
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".hip")))]
    files.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "sgfhe_hip.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def embedded_build_id(path):
    """sgfhe_build_id() of a library file, read without loading it (None if absent)."""
    try:
        with open(path, "rb") as fh:
            data = fh.read()
    except OSError:
        return None
    i = data.find(b"SGFHE_BUILD_ID=")
    if i < 0:
        return None
    j = data.find(b"\0", i)
    return data[i + 15:j].decode("ascii", "replace")


def _stale():
    """The in-tree library is missing or was not built from the sources next to it."""
    return embedded_build_id(LIB_PATH) != source_hash()


def build(force=False):
    """Compile the HIP engine for gfx950 (hipcc cross-compiles without a GPU).  A library whose
    embedded build id differs from the sources in csrc/ is rebuilt."""
    if force or _stale():
        subprocess.check_call(["make", "-C", CSRC, "-B", "libsgfhe_hip.so"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get("SGFHE_HIP_LIB"):
        # an explicitly chosen library (ablation builds of tools/exp_*.sh): loaded as it is; its
        # sgfhe_build_id() tells bench.py that the committed profile counters are not its own
        if not os.path.exists(LIB_PATH):
            raise OSError("SGFHE_HIP_LIB=%s does not exist" % LIB_PATH)
    elif _stale():
        build()
    try:
        # PyTorch-ROCm ships its own copy of the HIP runtime.  Loading it first makes this
        # library bind to the same libamdhip64 instead of bringing a second runtime into the
        # process (with two runtimes, whichever initialises second sees no GPU).
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, u32, u64, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t
    sig = {
        "sgfhe_version": (ctypes.c_char_p, []),
        "sgfhe_abi_version": (u32, []),
        "sgfhe_build_id": (ctypes.c_char_p, []),
        "sgfhe_ctx_create": (i32, [ctypes.POINTER(SgfheParams), ctypes.c_int, ctypes.POINTER(vp)]),
        "sgfhe_ctx_create_ex": (i32, [ctypes.POINTER(SgfheParams), ctypes.c_int, u32, ctypes.POINTER(vp)]),
        "sgfhe_ctx_clone": (i32, [vp, ctypes.POINTER(vp)]),
        "sgfhe_set_coalesce": (i32, [vp, ctypes.c_int, u32, u32, u32]),
        "sgfhe_coalesce_stats": (i32, [vp, _u64p, ctypes.c_int]),
        "sgfhe_ctx_destroy": (i32, [vp]),
        "sgfhe_last_error_string": (ctypes.c_char_p, [vp]),
        "sgfhe_set_chunk": (i32, [vp, u32]),
        "sgfhe_set_lanes": (i32, [vp, u32]),
        "sgfhe_set_small_batch_max": (i32, [vp, u32]),
        "sgfhe_set_random_flatten": (i32, [vp, ctypes.c_int, u64]),
        "sgfhe_set_random_flatten_key": (i32, [vp, ctypes.c_int, ctypes.c_char_p]),
        "sgfhe_bkey_upload": (i32, [vp, vp, sz]),
        "sgfhe_bkey_upload_rns2": (i32, [vp, vp, sz, u64, u64]),
        "sgfhe_rns2_convert": (i32, [vp, ctypes.c_int, vp, sz, u64, u64, vp]),
        "sgfhe_bkey_generate": (i32, [vp, vp, sz, ctypes.c_char_p, u32]),
        "sgfhe_bkey_device_form_bytes": (i32, [vp, ctypes.POINTER(sz)]),
        "sgfhe_bkey_export_device_form": (i32, [vp, vp]),
        "sgfhe_bkey_import_device_form": (i32, [vp, vp]),
        "sgfhe_bootstrap_batch": (i32, [vp, vp, vp, vp, vp, sz, vp, u32]),
        "sgfhe_bootstrap_batch_device": (i32, [vp, vp, vp, vp, vp, sz, vp, u32, vp]),
        "sgfhe_sync": (i32, [vp]),
        "sgfhe_external_product": (i32, [vp, vp, vp, vp, vp, vp]),
        "sgfhe_pack_encrypted_bits": (i32, [vp, vp, vp, sz, vp, vp]),
        "sgfhe_debug_cmux": (i32, [vp, vp, vp, vp, u64, vp, vp]),
        "sgfhe_debug_accumulators": (i32, [vp, vp, vp, vp, vp, sz, u64, vp]),
        "sgfhe_debug_digits": (i32, [vp, vp, vp, vp, vp, sz, u64, vp]),
        "sgfhe_debug_flatten": (i32, [vp, vp, vp]),
        "sgfhe_debug_ntt": (i32, [vp, u32, ctypes.c_int, vp, vp]),
        "sgfhe_debug_primes": (i32, [vp, _u32p, _u32p]),
        "sgfhe_host_deterministic_expand": (i32, [ctypes.POINTER(SgfheParams), vp, vp]),
        "sgfhe_host_encrypt_private": (i32, [ctypes.POINTER(SgfheParams), vp, vp, vp, vp, vp, vp]),
        "sgfhe_host_pack_private": (i32, [ctypes.POINTER(SgfheParams), vp, vp]),
        "sgfhe_host_normalize_private": (i32, [ctypes.POINTER(SgfheParams), vp, vp, vp, vp]),
        "sgfhe_host_split_ciphertext": (i32, [ctypes.POINTER(SgfheParams), vp, vp, sz, vp, vp]),
        "sgfhe_host_decrypt_lwe": (i32, [ctypes.POINTER(SgfheParams), vp, vp, vp, sz, vp]),
        "sgfhe_host_decrypt_rlwe": (i32, [ctypes.POINTER(SgfheParams), vp, vp, vp, sz, vp]),
        "sgfhe_host_public_key": (i32, [ctypes.POINTER(SgfheParams), u64, vp, vp, vp, vp]),
        "sgfhe_host_encrypt_public": (i32, [ctypes.POINTER(SgfheParams), u64, vp, vp, vp, vp, vp, vp, vp, vp]),
        "sgfhe_host_pack_public": (i32, [ctypes.POINTER(SgfheParams), vp, vp, vp, vp]),
        "sgfhe_host_normalize_public": (i32, [ctypes.POINTER(SgfheParams), vp, vp, vp, vp]),
        "sgfhe_timing_enable": (i32, [vp, ctypes.c_int]),
        "sgfhe_timing_read": (i32, [vp, ctypes.POINTER(ctypes.c_double), ctypes.c_int]),
        "sgfhe_kernel_names": (i32, [vp, ctypes.c_char_p, sz, ctypes.c_char_p, sz]),
        "sgfhe_release_host_staging": (i32, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if L.sgfhe_abi_version() != ABI_VERSION:
        raise OSError("%s implements ABI revision %d, this binding needs %d (stale library?)"
                      % (LIB_PATH, L.sgfhe_abi_version(), ABI_VERSION))
    _lib = L
    return L


EXPORTED_SYMBOLS = (
    "sgfhe_version", "sgfhe_abi_version", "sgfhe_build_id", "sgfhe_ctx_create", "sgfhe_ctx_create_ex", "sgfhe_ctx_clone", "sgfhe_set_coalesce", "sgfhe_coalesce_stats", "sgfhe_ctx_destroy", "sgfhe_last_error_string",
    "sgfhe_set_chunk", "sgfhe_set_lanes", "sgfhe_set_small_batch_max", "sgfhe_set_random_flatten", "sgfhe_set_random_flatten_key", "sgfhe_bkey_upload", "sgfhe_bkey_upload_rns2", "sgfhe_rns2_convert", "sgfhe_bkey_generate",
    "sgfhe_bkey_device_form_bytes", "sgfhe_bkey_export_device_form",
    "sgfhe_bkey_import_device_form", "sgfhe_bootstrap_batch", "sgfhe_bootstrap_batch_device",
    "sgfhe_sync", "sgfhe_external_product", "sgfhe_pack_encrypted_bits", "sgfhe_debug_cmux", "sgfhe_debug_accumulators", "sgfhe_debug_digits", "sgfhe_debug_flatten", "sgfhe_debug_ntt",
    "sgfhe_debug_primes", "sgfhe_host_deterministic_expand", "sgfhe_host_encrypt_private",
    "sgfhe_host_pack_private", "sgfhe_host_normalize_private", "sgfhe_host_split_ciphertext",
    "sgfhe_host_decrypt_lwe", "sgfhe_host_decrypt_rlwe", "sgfhe_host_public_key",
    "sgfhe_host_encrypt_public", "sgfhe_host_pack_public", "sgfhe_host_normalize_public", "sgfhe_timing_enable", "sgfhe_timing_read",
    "sgfhe_kernel_names", "sgfhe_release_host_staging")
