"""examples/depth.jl of the reference: feed the (AND, XOR) outputs of a gate back into the next
gate for 100 levels and watch the LWE error, which a bootstrap refreshes at every level.
Params(512), 16 independent chains in one batch; odd levels use the randomised flatten.
Run on a GPU box:  python examples/depth.py [levels]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgfhe_jl_amd as S


def lwe_error(key, enc_bit, ref):
    p = key.params
    e = (int(enc_bit.lwe.b) - int(np.sum(enc_bit.lwe.a * key.key, dtype=np.uint64))
         - int(ref) * p.Dr) % p.r
    return e - p.r if e > p.r // 2 else e


def main(levels=100, chains=16):
    rng = np.random.default_rng()
    params = S.Params(512)
    key = S.PrivateKey(params, rng)
    bkey = S.BootstrapKey(rng, key, random_flatten=True)   # both flatten modes are used below
    bits = rng.integers(0, 2, size=params.n).astype(bool)
    enc = S.split_ciphertext(S.encrypt(key, rng, bits))
    e1, e2 = enc[0:2 * chains:2], enc[1:2 * chains:2]
    y1, y2 = bits[0:2 * chains:2].copy(), bits[1:2 * chains:2].copy()
    print("input errors:", [lwe_error(key, e, y) for e, y in zip(e1, y1)][:8], "Dr/4 =", params.Dr // 4)
    for level in range(levels):
        res = S.bootstrap_batch(bkey, rng if level % 2 else None, e1, e2)
        for i, (r_and, r_or, r_xor) in enumerate(res):
            assert S.decrypt(key, r_and) == (y1[i] & y2[i])
            assert S.decrypt(key, r_or) == (y1[i] | y2[i])
            assert S.decrypt(key, r_xor) == (y1[i] ^ y2[i])
        e1, e2 = [r[0] for r in res], [r[2] for r in res]
        y1, y2 = y1 & y2, y1 ^ y2
        if level % 10 == 9:
            errs = [abs(lwe_error(key, e, y)) for e, y in zip(e1 + e2, list(y1) + list(y2))]
            print("level %3d: max |error| = %d" % (level + 1, max(errs)))
    print("%d levels x %d chains decrypted correctly" % (levels, chains))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100)
