"""sgfhe.jl_amd -- MI355X-native gate-bootstrap engine for Gao's FHE scheme.

Host-side mirror of the SGFHE.jl API for the bootstrap path (`Params`, `PrivateKey`,
`BootstrapKey`, `encrypt`, `split_ciphertext`, `decrypt`, `bootstrap`) over the C ABI of
libsgfhe_hip.so (hand-written HIP for gfx950; include/sgfhe_hip.h).  Import as
`import sgfhe_jl_amd` (repo-root shim; the directory name contains a dot).
"""

from ._lib import build, lib, LIB_PATH, EXPORTED_SYMBOLS, ABI_VERSION, source_hash, embedded_build_id
from .engine import Engine, SgfheError, FLAG_RAW_MODQ, CTX_RANDOM_FLATTEN, CTX_DETERMINISTIC_ONLY
from .params import Params, find_modulus, isprime
from . import distributed
from . import host
from .scheme import (PrivateKey, PublicKey, PublicEncryptedCiphertext, BootstrapKey, LWE, RLWE, EncryptedBit, PackedCiphertext,
                     Ciphertext, encrypt, extract, split_ciphertext, decrypt, bootstrap,
                     bootstrap_batch, pack_encrypted_bits, encrypt_optimal, normalize_ciphertext,
                     PrivateEncryptedCiphertext, packbits, unpackbits, prng_expand)

__all__ = ["distributed", "host", "build", "lib", "LIB_PATH", "EXPORTED_SYMBOLS", "ABI_VERSION", "source_hash", "embedded_build_id", "Engine", "SgfheError",
           "FLAG_RAW_MODQ", "CTX_RANDOM_FLATTEN", "CTX_DETERMINISTIC_ONLY", "Params", "find_modulus", "isprime", "PrivateKey", "BootstrapKey",
           "PublicKey", "PublicEncryptedCiphertext",
           "LWE", "RLWE", "EncryptedBit", "PackedCiphertext", "encrypt", "extract",
           "split_ciphertext", "decrypt", "bootstrap", "bootstrap_batch", "Ciphertext",
           "pack_encrypted_bits", "encrypt_optimal", "normalize_ciphertext",
           "PrivateEncryptedCiphertext", "packbits", "unpackbits", "prng_expand"]
