"""One-off check at the largest reference parameter set, Params(2048): device key generation,
pack_encrypted_bits (2048 bootstraps + shortened external products) -> decrypt, one gate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, sgfhe_jl_amd as S
rng = np.random.default_rng(1)
params = S.Params(2048)
key = S.PrivateKey(params, rng)
t = time.time(); bkey = S.BootstrapKey(rng, key); print("keygen", time.time() - t)
msg = rng.integers(0, 2, size=params.n).astype(bool)
bits = S.split_ciphertext(S.encrypt(key, rng, msg))
t = time.time(); ct = S.pack_encrypted_bits(bkey, None, bits); print("pack", time.time() - t)
print("ok", np.array_equal(S.decrypt(key, ct), msg))
r = S.bootstrap(bkey, None, bits[0], bits[1])
print([S.decrypt(key, x) for x in r], msg[0], msg[1])
