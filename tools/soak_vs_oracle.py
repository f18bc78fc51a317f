"""Soak of the production schedule against the oracle on DISTINCT inputs: `count` bootstraps with
uniformly random LWE words (every rotation amount, not only valid encryptions) plus encryptions of
all four bit pairs, tiled to a full batch so that the engine runs its default two lanes of
full-size chunks; every output word of every copy is compared with the C restatement (run in the
GPU path's algebra on the host cores, bit-identical to its reference-shaped form by
tests/test_oracle_properties.py).  The parity tests pin 2-4 distinct inputs per parameter set at
full batch; this widens the sample.
A case `name:count:rnd` runs the randomised flatten (bootstrap(bkey, rng, ...)): there a row's draws depend on its
position in the call, so the oracle computes `count` rows at random positions of the full batch, each on the
draw stream of its position (rnd = (key, call, positions)), and those rows of the engine's output are compared.
usage (GPU box): python tools/soak_vs_oracle.py [params1024:128 params512:512 params1024:64:rnd ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import bench
import sgfhe_jl_amd as S
import oracle_c


def threads():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(q) // int(per))
    except OSError:
        pass
    return os.cpu_count() or 1


def main():
    cases = sys.argv[1:] or ["params1024:128", "params512:512", "synth64:64"]   # prime moduli (NTT oracle)
    for case in cases:
        name, count, *mode = case.split(":")
        count = int(count)
        random_mode = mode == ["rnd"]
        p = bench.make_params(S, name)
        o = oracle_c.Oracle.from_params(p)
        sk = o.private_key(31)
        t0 = time.perf_counter()
        bkey = o.bootstrap_key(sk, 32)
        eng = S.Engine(p)
        eng.generate_key(sk, 32)
        rng = np.random.default_rng(33)
        a1 = rng.integers(0, p.r, size=(count, p.n), dtype=np.uint64)
        a2 = rng.integers(0, p.r, size=(count, p.n), dtype=np.uint64)
        b1 = rng.integers(0, p.r, size=count, dtype=np.uint64)
        b2 = rng.integers(0, p.r, size=count, dtype=np.uint64)
        bits = np.array([0, 0, 0, 1, 1, 0, 1, 1] * 2, dtype=np.uint8)       # valid encryptions up front
        ea, eb = o.lwe_encrypt_bits(sk, bits, 34)
        k = len(bits) // 2
        a1[:k], b1[:k], a2[:k], b2[:k] = ea[0::2], eb[0::2], ea[1::2], eb[1::2]
        t1 = time.perf_counter()
        if not o.uses_ntt:
            raise SystemExit("%s: composite modulus -- tests/test_gpu_rns2.py covers that ring" % name)
        khat = o.key_transform(bkey, threads=threads())
        full = 4096 if p.m >= 4096 else 16384
        idx = np.random.default_rng(35).permutation(np.resize(np.arange(count), full))
        if random_mode:
            fkey = bytes(range(100, 132))
            pos = np.sort(np.random.default_rng(36).choice(full, size=count, replace=False))
            pos[:k] = np.arange(k)                       # the valid pairs are compared (and decrypted) too
            idx[:k] = np.arange(k)
            rows = idx[pos]
            ref = o.bootstrap_batch(khat, a1[rows], b1[rows], a2[rows], b2[rows], threads=threads(), opt=True,
                                    rnd=(fkey, 0, pos.astype(np.uint32)))
            t2 = time.perf_counter()
            eng.set_random_flatten(True, fkey)
            out = eng.bootstrap_batch(a1[idx], b1[idx], a2[idx], b2[idx])
            t3 = time.perf_counter()
            ok = np.array_equal(out[pos], ref)
            name += " randomised flatten"
            compared = ref.size
        else:
            ref = o.bootstrap_batch(khat, a1, b1, a2, b2, threads=threads(), opt=True)
            t2 = time.perf_counter()
            out = eng.bootstrap_batch(a1[idx], b1[idx], a2[idx], b2[idx])
            t3 = time.perf_counter()
            ok = np.array_equal(out, ref[idx])
            compared = out.size
        y1, y2 = bits[0::2], bits[1::2]
        dec_ok = all(np.array_equal(o.lwe_decrypt_bits(sk, ref[:k, g, :p.n], ref[:k, g, p.n]), fn(y1, y2))
                     for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)))
        print("%s: %d distinct bootstraps tiled to a batch of %d: %s (%d output words compared); truth table of "
              "the %d valid pairs: %s; oracle %.1f s on %d threads, GPU call %.2f s"
              % (name, count, full, "bit-exact" if ok else "MISMATCH", compared, k, "ok" if dec_ok else "WRONG",
                 t2 - t1, threads(), t3 - t2), flush=True)
        eng.close()
        if not (ok and dec_ok):
            raise SystemExit(1)


if __name__ == "__main__":
    main()
