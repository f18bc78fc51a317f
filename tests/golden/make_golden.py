#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ with the big-integer oracle
(oracle/bigint_oracle.py: literal restatement of the reference path, Kronecker products).
The reference itself holds no fixtures and cannot run here, so these vectors are
the build's own; they pin the C oracle and the HIP engine to the big-integer restatement.

cfg4 (BASELINE.json config 4: n = 1024 over the composite Q = B * Bp of the src/fhe2.jl:57-58
prime rule) is generated with the C restatement in its RNS2Number mode (limb-wise NTT products,
src/rns.jl:51-60) -- the big-integer oracle would need hours for the 1 GiB key -- and pinned to
the big-integer oracle where that is affordable: every key polynomial product of slices 0 and 1
and the accumulators after the first two iterations are recomputed with Kronecker products.

Usage: python tests/golden/make_golden.py [p64] [p64rnd] [extprod] [tables] [p512] [p1024] [p1024rnd] [pack64] [cfg4]
"""

import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bigint_oracle as O  # noqa: E402


def h_ints(vals, nbytes=16):
    h = hashlib.sha256()
    for v in vals:
        h.update(int(v).to_bytes(nbytes, "little"))
    return h.hexdigest()


def key_hash(bk):
    h = hashlib.sha256()
    for k in bk:
        for row in k:
            for col in row:
                for c in col:
                    h.update(int(c).to_bytes(16, "little"))
    return h.hexdigest()


def bootstrap_case(n, sk_seed, key_seed, in_seed, pairs, checkpoints, full_outputs=True):
    p = O.Params.make(n)
    sk = O.private_key(p, sk_seed)
    t0 = time.time()
    bk = O.bootstrap_key(p, sk, key_seed)
    print("  key", time.time() - t0, "s", flush=True)
    g = O.SplitMix64(in_seed)
    cases = []
    for (y1, y2) in pairs:
        l1 = O.lwe_encrypt_bit(p, sk, y1, g)
        l2 = O.lwe_encrypt_bit(p, sk, y2, g)
        cps = {}

        def trace(k, a, b):
            if (k + 1) in checkpoints:
                cps[str(k + 1)] = [h_ints(a), h_ints(b)]
        t0 = time.time()
        raw = O.bootstrap_internal(p, bk, l1, l2, trace=trace)
        out = [([O.reduce_modulus(p.r, x, p.Q) for x in a], O.reduce_modulus(p.r, b, p.Q))
               for a, b in raw]
        print("  bootstrap", (y1, y2), time.time() - t0, "s", flush=True)
        dec = [O.lwe_decrypt_bit(p, sk, o) for o in out]
        assert dec == [y1 & y2, y1 | y2, y1 ^ y2], dec
        case = {"bits": [y1, y2], "lwe1": {"a": l1[0], "b": l1[1]}, "lwe2": {"a": l2[0], "b": l2[1]},
                "acc_sha256_after": cps,
                "raw_sha256": [h_ints(a + [b]) for a, b in raw]}
        if full_outputs:
            case["out"] = [a + [b] for a, b in out]
        else:
            case["out_sha256"] = [h_ints(a + [b], 8) for a, b in out]
            case["out_head"] = [(a + [b])[:8] + [b] for a, b in out]
        cases.append(case)
    return {"params": {"n": p.n, "r": p.r, "m": p.m, "Q": str(p.Q), "B": str(p.B),
                       "DQ_tilde": str(p.DQ_tilde)},
            "prng": "SplitMix64 (oracle/bigint_oracle.py)", "sk_seed": sk_seed,
            "key_seed": key_seed, "in_seed": in_seed, "sk": sk, "key_sha256": key_hash(bk),
            "cases": cases}


def random_flatten_case():
    """Params(64) with the randomised flatten (src/utils.jl:198-241) on the engine's ChaCha8 draw
    stream: the key and inputs of p64's cases (1, 0) and (1, 1) as ONE call of two bootstraps
    (bootstrap j draws with index j, call 0).  Pins the stream's definition (key bytes, counter
    layout, draw mapping) as well as the algorithm."""
    p = O.Params.make(64)
    sk = O.private_key(p, 1)
    bk = O.bootstrap_key(p, sk, 2)
    g = O.SplitMix64(3)
    lwes = {}
    for pair in [(0, 0), (0, 1), (1, 0), (1, 1)]:                  # the draws of bootstrap_case
        lwes[pair] = (O.lwe_encrypt_bit(p, sk, pair[0], g), O.lwe_encrypt_bit(p, sk, pair[1], g))
    fkey = bytes(range(7, 39))
    cases = []
    for j, pair in enumerate([(1, 0), (1, 1)]):
        l1, l2 = lwes[pair]
        cps = {}

        def trace(k, a, b):
            if (k + 1) in (1, 2, 64):
                cps[str(k + 1)] = [h_ints(a), h_ints(b)]
        raw = O.bootstrap_internal(p, bk, l1, l2, trace=trace, rng=O.ChaChaFlatten(p, fkey, boot=j, call=0))
        out = [([O.reduce_modulus(p.r, x, p.Q) for x in a], O.reduce_modulus(p.r, b, p.Q)) for a, b in raw]
        y1, y2 = pair
        assert [O.lwe_decrypt_bit(p, sk, o) for o in out] == [y1 & y2, y1 | y2, y1 ^ y2]
        cases.append({"bits": list(pair), "lwe1": {"a": l1[0], "b": l1[1]}, "lwe2": {"a": l2[0], "b": l2[1]},
                      "acc_sha256_after": cps, "raw_sha256": [h_ints(a + [b]) for a, b in raw],
                      "out": [a + [b] for a, b in out]})
    return {"params": {"n": p.n, "r": p.r, "m": p.m, "Q": str(p.Q), "B": str(p.B)},
            "sk_seed": 1, "key_seed": 2, "in_seed": 3, "key_sha256": key_hash(bk),
            "flatten_key_hex": fkey.hex(), "stream": "ChaCha8, oracle/bigint_oracle.py ChaChaFlatten",
            "call": 0, "cases": cases}


def random_flatten_case_1024():
    """Params(1024) with the randomised flatten (src/utils.jl:198-241) -- the reference's documented
    call bootstrap(bkey, rng, ...) (README.md:24, docs/src/manual.md:144) at its own parameter set --
    on the engine's ChaCha8 draw stream: key and input pair of p1024.json's case, as bootstrap 0 and
    (the same inputs again) bootstrap 5 of call 2 of the stream, so that the counter words for the
    bootstrap index and the call are pinned at this ring too.  Literal big-integer restatement."""
    p = O.Params.make(1024)
    sk = O.private_key(p, 21)
    t0 = time.time()
    bk = O.bootstrap_key(p, sk, 22)
    print("  key", time.time() - t0, "s", flush=True)
    g = O.SplitMix64(23)
    l1 = O.lwe_encrypt_bit(p, sk, 1, g)
    l2 = O.lwe_encrypt_bit(p, sk, 1, g)
    fkey = bytes(range(11, 43))
    cases = []
    for boot, call in ((0, 0), (5, 2)):
        cps = {}

        def trace(k, a, b):
            if (k + 1) in (1, 2, 512, 1024):
                cps[str(k + 1)] = [h_ints(a), h_ints(b)]
        t0 = time.time()
        raw = O.bootstrap_internal(p, bk, l1, l2, trace=trace, rng=O.ChaChaFlatten(p, fkey, boot=boot, call=call))
        out = [([O.reduce_modulus(p.r, x, p.Q) for x in a], O.reduce_modulus(p.r, b, p.Q)) for a, b in raw]
        print("  bootstrap", boot, call, time.time() - t0, "s", flush=True)
        assert [O.lwe_decrypt_bit(p, sk, o) for o in out] == [1, 1, 0]
        cases.append({"bits": [1, 1], "boot": boot, "call": call,
                      "lwe1": {"a": l1[0], "b": l1[1]}, "lwe2": {"a": l2[0], "b": l2[1]},
                      "acc_sha256_after": cps, "raw_sha256": [h_ints(a + [b]) for a, b in raw],
                      "out_sha256": [h_ints(a + [b], 8) for a, b in out],
                      "out_head": [(a + [b])[:8] + [b] for a, b in out]})
    return {"params": {"n": p.n, "r": p.r, "m": p.m, "Q": str(p.Q), "B": str(p.B)},
            "sk_seed": 21, "key_seed": 22, "in_seed": 23, "key_sha256": key_hash(bk),
            "flatten_key_hex": fkey.hex(), "stream": "ChaCha8, oracle/bigint_oracle.py ChaChaFlatten",
            "cases": cases}


def extprod_case():
    """test/internals.test.jl:144-166 shape: q = 2^60 - 1, B = 2^30, length 64; random matrix."""
    m, n = 64, 8
    B = 1 << 30
    Q = B * B - 1
    p = O.Params.custom(n, Q, B)
    g = O.SplitMix64(99)
    a = [g.below(Q) for _ in range(m)]
    b = [g.below(Q) for _ in range(m)]
    A = [[[g.below(Q) for _ in range(m)] for _ in range(2)] for _ in range(4)]
    ra, rb = O.external_product(a, b, A, B, 2, Q)
    return {"Q": str(Q), "B": str(B), "m": m, "n": n, "a": a, "b": b, "A": A, "a_res": ra, "b_res": rb}


def tables():
    """test/internals.test.jl:26-112 shapes: rescale table and exhaustive flatten tables."""
    out = {"rescale": [], "flatten": []}
    old_max = 2 ** 12 + 1
    for new_max in (16, 17):
        for rnd in (False, True):
            out["rescale"].append({"new_max": new_max, "old_max": old_max, "round": rnd,
                                   "values": [O.rescale(new_max, x, old_max, rnd) for x in range(old_max)]})
    for B in (4, 5):
        for ell in (2, 3, 4):
            q = B ** ell - 1
            out["flatten"].append({"B": B, "ell": ell, "q": q,
                                   "values": [O.flatten(a, B, ell, q) for a in range(q)]})
    return out


def pack_case():
    """test/api.test.jl:86-108 at Params(64): n LWEs -> pack_encrypted_bits -> RLWE (w, v)."""
    p = O.Params.make(64)
    sk = O.private_key(p, 1)
    bk = O.bootstrap_key(p, sk, 2)
    g = O.SplitMix64(5)
    bits = [g.next() & 1 for _ in range(p.n)]
    lwes = [O.lwe_encrypt_bit(p, sk, b, g) for b in bits]
    w, v = O.pack_encrypted_bits(p, bk, lwes)
    assert O.decrypt_ciphertext(p, sk, w, v) == bits
    assert [O.lwe_decrypt_bit(p, sk, l) for l in O.split_ciphertext(p, w, v)] == bits
    return {"n": p.n, "sk_seed": 1, "key_seed": 2, "in_seed": 5, "bits": bits,
            "a": [l[0] for l in lwes], "b": [l[1] for l in lwes], "w": w, "v": v,
            "key_sha256": key_hash(bk)}


def cfg4_case():
    import math
    import numpy as np
    import oracle_c as OC
    n = 1024
    r = 16 * n
    bound = math.isqrt(1220 * r ** 4 * n ** 2) + 1
    Bp = O.find_modulus(r, bound)                        # src/fhe2.jl:57
    B = O.find_modulus(r, Bp + 1)                        # src/fhe2.jl:58
    Q = B * Bp
    p = O.Params.custom(n, Q, B)
    o = OC.Oracle.from_params(p, rns2=(B, Bp))
    sk_seed, key_seed, in_seed = 41, 42, 45
    sk = o.private_key(sk_seed)
    t0 = time.time()
    bkey = o.bootstrap_key(sk, key_seed)
    print("  key (C, RNS2 limbs)", time.time() - t0, "s", flush=True)
    skl = [int(x) for x in sk]
    # big-integer check of the first two key slices (a, a * s + e + s_k G)
    ext = O.resize(skl, p.m)
    G = O.gadget_matrix(p)
    for k in range(2):
        for row in range(4):
            a = OC.u128_to_ints(bkey[k, row, 0])
            bb = OC.u128_to_ints(bkey[k, row, 1])
            a0 = list(a)
            a0[0] = (a0[0] - ext[k] * G[row][0]) % Q
            prod = O.poly_mul(a0, ext, Q)
            e = [(x - y) % Q for x, y in zip(bb, prod)]
            e[0] = (e[0] - ext[k] * G[row][1]) % Q
            assert all(v <= n or v >= Q - n for v in e), "noise out of range: key product wrong"
    print("  key slices 0, 1 verified with Kronecker products", flush=True)
    bits = np.array([1, 1], dtype=np.uint8)
    a, b = o.lwe_encrypt_bits(sk, bits, in_seed)
    l1 = ([int(x) for x in a[0]], int(b[0]))
    l2 = ([int(x) for x in a[1]], int(b[1]))
    cps = {}
    for k in (1, 2, 512, 1024):
        t0 = time.time()
        _, acc = o.bootstrap_batch(bkey, a[0:1], b[0:1], a[1:2], b[1:2], n_iters=k, want_acc=True)
        cps[str(k)] = [h_ints(OC.u128_to_ints(acc[0, 0])), h_ints(OC.u128_to_ints(acc[0, 1]))]
        print("  acc after", k, time.time() - t0, "s", flush=True)
    # big-integer oracle for the first two iterations (needs key slices 0 and 1 only)
    bk2 = [[[OC.u128_to_ints(bkey[k, row, c]) for c in range(2)] for row in range(4)] for k in range(2)]
    ua = [(x + y) % p.r for x, y in zip(l1[0], l2[0])]
    ub = (l1[1] + l2[1]) % p.r
    ba = [0] * p.m
    bbp = [(c * p.DQ_tilde) % Q for c in O.mul_by_monomial(O.initial_poly(p), -ub, Q)]
    for k in range(2):
        A = []
        for row in range(4):
            Arow = []
            for col in range(2):
                x = O.mul_by_xj_minus_one(bk2[k][row][col], ua[k], Q)
                x[0] = (x[0] + G[row][col]) % Q
                Arow.append(x)
            A.append(Arow)
        ba, bbp = O.external_product(ba, bbp, A, p.B, p.ell, Q)
        assert [h_ints(ba), h_ints(bbp)] == cps[str(k + 1)], "C RNS2 path differs from the big-integer oracle"
    print("  iterations 1, 2 verified with the big-integer oracle", flush=True)
    raw = o.bootstrap_batch(bkey, a[0:1], b[0:1], a[1:2], b[1:2], raw=True)
    out = o.bootstrap_batch(bkey, a[0:1], b[0:1], a[1:2], b[1:2])
    dec = o.lwe_decrypt_bits(sk, out[0, :, :n], out[0, :, n])
    assert list(dec) == [1, 1, 0], dec
    case = {"bits": [1, 1], "lwe1": {"a": l1[0], "b": l1[1]}, "lwe2": {"a": l2[0], "b": l2[1]},
            "acc_sha256_after": cps,
            "raw_sha256": [h_ints(OC.u128_to_ints(raw[0, g])) for g in range(3)],
            "out_sha256": [h_ints([int(v) for v in out[0, g]], 8) for g in range(3)],
            "out_head": [[int(v) for v in out[0, g, :8]] + [int(out[0, g, n])] for g in range(3)]}
    import hashlib as _h
    return {"params": {"n": n, "r": r, "m": p.m, "Q": str(Q), "B": str(B), "Bp": str(Bp),
                       "DQ_tilde": str(p.DQ_tilde)},
            "prng": "ChaCha20 key streams, SplitMix64 test plumbing (oracle/)", "sk_seed": sk_seed,
            "key_seed": key_seed, "in_seed": in_seed,
            "key_sha256": _h.sha256(np.ascontiguousarray(bkey).tobytes()).hexdigest(),
            "generated_by": "oracle/sgfhe_oracle.c in RNS2Number mode; key slices 0-1 and iterations "
                            "1-2 re-derived with oracle/bigint_oracle.py",
            "cases": [case]}


def main():
    what = sys.argv[1:] or ["p64", "p64rnd", "extprod", "tables", "p512", "p1024", "p1024rnd", "pack64", "cfg4"]
    pairs4 = [(0, 0), (0, 1), (1, 0), (1, 1)]
    for w in what:
        print(w, flush=True)
        if w == "p64":
            d = bootstrap_case(64, 1, 2, 3, pairs4, {1, 2, 32, 64})
        elif w == "p512":
            d = bootstrap_case(512, 11, 12, 13, [(1, 0)], {1, 2, 256, 512})
        elif w == "p1024":
            d = bootstrap_case(1024, 21, 22, 23, [(1, 1)], {1, 2, 512, 1024})
        elif w == "extprod":
            d = extprod_case()
        elif w == "p64rnd":
            d = random_flatten_case()
        elif w == "p1024rnd":
            d = random_flatten_case_1024()
        elif w == "tables":
            d = tables()
        elif w == "pack64":
            d = pack_case()
        elif w == "cfg4":
            d = cfg4_case()
        else:
            raise SystemExit("unknown " + w)
        with open(os.path.join(HERE, w + ".json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
    print("done")


if __name__ == "__main__":
    main()
