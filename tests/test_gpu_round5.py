"""Round-5 parity evidence on the GPU: the full-size paths whose only GPU check was
decryption, pinned to bytes of the C restatement committed as fixtures (tests/golden/make_golden_c.py,
run in the build container):

 * pack_encrypted_bits (src/fhe.jl:632-641,660-696) at Params(512) and Params(1024), `rng = nothing` and
   `rng::AbstractRNG` (two ciphertexts of one call: the draws of the n bootstraps, fhe.jl:673, and of the
   flatten of every as_i, fhe.jl:683-684, at their stream positions);
 * complete bootstraps at Params(2048) (src/fhe.jl:71-78: the largest ring the reference can build; six
   primes, B > 2^46: the randomised mode runs k_extprod<14, 4, true> with the third digit plane) in both
   flatten modes, both kernel forms: accumulators after 1 and 2 iterations, raw residues mod Q, ModRed words;
 * Params(128) and Params(256) in the randomised mode (src/utils.jl:198-241), likewise.

A fixture holds seeds and digests: the secret key and the inputs are regenerated here with the oracle's cheap
plumbing, the bootstrap key on the device from the key seed (byte-identical to the oracle's -- which these
digests pin at these sizes too).  Everything goes through the C ABI."""

import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, G)


def _load(name):
    path = os.path.join(G, name + ".json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/%s.json not generated (tests/golden/make_golden_c.py %s)" % (name, name))
    with open(path) as f:
        return json.load(f)


@pytest.mark.parametrize("n", [512, 1024])
def test_pack_encrypted_bits_full_size_vs_committed_oracle_bytes(S, oc, n):
    """pack_encrypted_bits at Params(512) / Params(1024): every word of (w, v) through its SHA-256, the first
    words in clear, and decryption (test/api.test.jl:86-108), deterministic and randomised; the exact-accumulation
    group size of the packing kernels (engine.hip pack_G / pack_G_rnd) and their m = 4096 / 8192 instantiations
    are what these sizes add to the small rings of test_gpu_golden.py."""
    import make_golden_c as MG
    d = _load("pack%d" % n)
    params = S.Params(n)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(d["sk_seed"])
    eng = S.Engine(params)
    eng.generate_key(sk, d["key_seed"])
    bits = np.random.default_rng(d["in_seed"]).integers(0, 2, size=n).astype(np.uint8)
    a0, b0 = o.lwe_encrypt_bits(sk, bits, d["in_seed"] + 1)
    a1, b1 = o.lwe_encrypt_bits(sk, 1 - bits, d["in_seed"] + 2)

    def same(got_w, got_v, rec, want_bits, what):
        assert MG.head(got_w) == rec["w_head"] and MG.head(got_v) == rec["v_head"], what
        assert MG.sha_words(got_w) == rec["w_sha256"] and MG.sha_words(got_v) == rec["v_sha256"], what
        assert np.array_equal(S.host.decrypt_rlwe(params, sk, got_w, got_v), want_bits.astype(bool)), what

    w, v = eng.pack_encrypted_bits(a0[None], b0[None])
    same(w[0], v[0], d["det"], bits, "deterministic")
    eng.set_random_flatten(True, bytes.fromhex(d["flatten_key_hex"]))          # call counter 0 = d["call"]
    w, v = eng.pack_encrypted_bits(np.stack([a0, a1]), np.stack([b0, b1]))     # ciphertexts 0 and 1 of call 0
    same(w[0], v[0], d["rnd"][0], bits, "randomised, ciphertext 0")
    same(w[1], v[1], d["rnd"][1], 1 - bits, "randomised, ciphertext 1")
    assert MG.sha_words(w[0]) != d["det"]["w_sha256"]
    eng.set_random_flatten(False)
    w, v = eng.pack_encrypted_bits(a0[None], b0[None])                         # and back to the bit-exact path
    same(w[0], v[0], d["det"], bits, "deterministic again")
    eng.close()


@pytest.mark.parametrize("name", ["p128rnd", "p256rnd", "p2048"])
def test_complete_bootstraps_vs_committed_oracle_bytes(S, oc, name):
    """Complete bootstraps in both flatten modes and both kernel forms against the C restatement's committed
    bytes: accumulators after k = 1, 2, raw residues mod Q and ModRed words of six rows (the four bit pairs, which
    must decrypt, and two rows of uniformly random words), the rows of ONE call (row t draws as bootstrap t)."""
    import make_golden_c as MG
    d = _load(name)
    n = d["n"]
    params = S.Params(n)
    o = oc.Oracle.from_params(params)
    sk = o.private_key(d["sk_seed"])
    eng = S.Engine(params)
    eng.generate_key(sk, d["key_seed"])
    a1, b1, a2, b2, bits = MG.mixed_inputs(o, sk, n, d["rows"], d["in_seed"])
    assert [int(x) for x in bits] == d["bits"]
    fkey = bytes.fromhex(d["flatten_key_hex"])
    rows = d["rows"]
    for form in (None, 0):                       # the latency form these few rows take by default, then k_extprod
        if form is not None:
            eng.set_small_batch_max(form)
        for mode in ("det", "rnd"):
            rec = d[mode]

            def fresh():                         # every call of the randomised mode as call 0 of the stream
                eng.set_random_flatten(mode == "rnd", fkey)
            if mode == "rnd" and n == 2048:
                fresh()
                assert eng.kernel_names()[0] == "k_extprod<14, 4, true>"
            for it, want in rec["acc_sha256_after"].items():
                fresh()
                acc = eng.debug_accumulators(a1, b1, a2, b2, int(it))
                assert [MG.sha_words(acc[t]) for t in range(rows)] == want, (name, mode, form, "accumulators", it)
            fresh()
            raw = eng.bootstrap_batch(a1, b1, a2, b2, raw=True)
            assert [MG.sha_words(raw[t]) for t in range(rows)] == rec["raw_sha256"], (name, mode, form, "raw")
            fresh()
            out = eng.bootstrap_batch(a1, b1, a2, b2)
            assert [MG.head(out[t]) for t in range(rows)] == rec["out_head"], (name, mode, form)
            assert [MG.sha_words(out[t]) for t in range(rows)] == rec["out_sha256"], (name, mode, form, "ModRed")
            k = len(bits) // 2
            y1, y2 = bits[0::2], bits[1::2]
            for g, fn in enumerate((np.bitwise_and, np.bitwise_or, np.bitwise_xor)):
                assert np.array_equal(o.lwe_decrypt_bits(sk, out[:k, g, :n], out[:k, g, n]), fn(y1, y2))
    assert d["det"]["out_sha256"] != d["rnd"]["out_sha256"]
    eng.close()


def test_batches_above_the_page_locked_limit_take_direct_copies(S, monkeypatch):
    """a host-pointer batch whose arrays exceed the page-locked staging limit (1 GiB per buffer:
    65536 gates at Params(1024)) is copied directly before and after the k-loop instead of chunk by chunk
    beside it.  SGFHE_PIN_MAX_MB lowers the limit so that a small batch crosses it: inputs under and results
    over the limit, both over, and the limit back at its default -- the same bytes every time."""
    params = S.Params(64)
    rng = np.random.default_rng(17)
    eng = S.Engine(params)
    eng.generate_key(rng.integers(0, 2, size=params.n, dtype=np.uint64), 18)
    batch = 3000                                   # inputs 3.1 MB, results 4.7 MB
    a1 = rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64)
    a2 = rng.integers(0, params.r, size=(batch, params.n), dtype=np.uint64)
    b1 = rng.integers(0, params.r, size=batch, dtype=np.uint64)
    b2 = rng.integers(0, params.r, size=batch, dtype=np.uint64)
    ref = eng.bootstrap_batch(a1, b1, a2, b2)                      # pipelined through the page-locked mirrors
    raw = eng.bootstrap_batch(a1, b1, a2, b2, raw=True)
    for mb in ("4", "1", "0"):
        monkeypatch.setenv("SGFHE_PIN_MAX_MB", mb)
        assert eng.bootstrap_batch(a1, b1, a2, b2).tobytes() == ref.tobytes(), mb
        assert eng.bootstrap_batch(a1, b1, a2, b2, raw=True).tobytes() == raw.tobytes(), mb
        assert eng.bootstrap_batch(a1[:5], b1[:5], a2[:5], b2[:5]).tobytes() == ref[:5].tobytes(), mb
    monkeypatch.delenv("SGFHE_PIN_MAX_MB")
    assert eng.bootstrap_batch(a1, b1, a2, b2).tobytes() == ref.tobytes()
    eng.close()
