"""The reference's README example (README.md:10-29 of nucypher/SGFHE.jl) on the MI355X engine:
keys, encryption of a block of bits, one gate bootstrap of two of them, decryption.
Run on a GPU box:  python examples/readme_example.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgfhe_jl_amd as S

rng = np.random.default_rng()
params = S.Params(64)
key = S.PrivateKey(params, rng)
bkey = S.BootstrapKey(rng, key)                     # generated on the GPU

bits = rng.integers(0, 2, size=params.n).astype(bool)   # a ciphertext holds n bits at once
encrypted_bits = S.split_ciphertext(S.encrypt(key, rng, bits))

i1, i2 = 9, 19                                       # the 10th and the 20th bit, as in the README
y1, y2 = bits[i1], bits[i2]
enc_and, enc_or, enc_xor = S.bootstrap(bkey, None, encrypted_bits[i1], encrypted_bits[i2])

res_and, res_or, res_xor = (S.decrypt(key, e) for e in (enc_and, enc_or, enc_xor))
assert res_and == (y1 & y2) and res_or == (y1 | y2) and res_xor == (y1 ^ y2)
print("y1 = %d, y2 = %d: AND = %d, OR = %d, XOR = %d" % (y1, y2, res_and, res_or, res_xor))
