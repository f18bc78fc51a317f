#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ with the big-integer oracle
(oracle/bigint_oracle.py: literal restatement of the reference path, Kronecker products).
The reference itself holds no fixtures and cannot run here, so these vectors are
the build's own; they pin the C oracle and the HIP engine to the big-integer restatement.

Usage: python tests/golden/make_golden.py [p64] [extprod] [tables] [p512] [p1024]
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


def main():
    what = sys.argv[1:] or ["p64", "extprod", "tables", "p512", "p1024", "pack64"]
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
        elif w == "tables":
            d = tables()
        elif w == "pack64":
            d = pack_case()
        else:
            raise SystemExit("unknown " + w)
        with open(os.path.join(HERE, w + ".json"), "w") as f:
            json.dump(d, f, separators=(",", ":"))
    print("done")


if __name__ == "__main__":
    main()
