# make_fixtures.jl -- run ONCE by a maintainer who has Julia, SGFHE.jl and DarkIntegers.jl:
#
#     julia --project=/path/to/SGFHE.jl make_fixtures.jl [outdir] [64 512 1024 ...]      (default: 64 512 1024)
#
# It cannot run in the build container (no julia binary there; SURVEY.md section 8c), which is why the
# oracle of this repository is "parity unpinned": the reference holds no numeric fixture of its own.
# This script makes them.  For every requested Params(n) it writes, in the layout of tests/golden/:
#
#   julia_p<n>.json      parameters, the private key bits, a few LWE input pairs, and for each pair the
#                        reference's own  bootstrap(bkey, nothing, bit1, bit2)  (three LWEs over Z_r,
#                        src/fhe.jl:608-621) and  _bootstrap_internal  (three LWEs over Z_Q,
#                        src/fhe.jl:559-595, residues as decimal strings); SHA-256 of the key file
#   julia_p<n>_key.bin   value.(coeffs) of BootstrapKey.key (src/fhe.jl:176-201), order
#                        [k][row][col][coef], every residue as two little-endian UInt64 (lo, hi):
#                        exactly the array sgfhe_bkey_upload takes (2 MiB at n = 64, 256 MiB at n = 512, 1 GiB
#                        at n = 1024)
#
# Drop both files into tests/golden/: tests/test_golden.py::test_julia_reference_fixture then pins
# the C restatement (and through it the big-integer one) to the Julia build's bytes, and
# tests/test_gpu_golden.py::test_engine_matches_julia_reference_fixture pins the HIP engine, with the
# SAME key (uploaded as it is, not regenerated).  Only `rng = nothing` outputs are recorded: the
# randomised flatten draws from Julia's rng, which nothing outside Julia reproduces (SURVEY.md F6).

using Random
using SHA
using SGFHE
using SGFHE: _bootstrap_internal
using DarkIntegers
using DarkIntegers: value

words(x) = (UInt64(UInt128(x) & typemax(UInt64)), UInt64(UInt128(x) >> 64))

function key_words(bkey)
    p = bkey.params
    canon = Vector{UInt64}(undef, p.n * 8 * p.m * 2)
    pos = 1
    for k in 1:p.n, row in 1:4, col in 1:2                 # BootstrapKey.key[k][row, col], src/fhe.jl:196
        for c in bkey.key[k][row, col].coeffs
            lo, hi = words(value(c))
            canon[pos] = lo; canon[pos + 1] = hi
            pos += 2
        end
    end
    canon
end

# (unsigned integers print in hexadecimal in Julia: everything goes through BigInt before it is written)
ints(v) = "[" * join(string.(BigInt.(v)), ",") * "]"
strs(v) = "[" * join(["\"" * string(BigInt(x)) * "\"" for x in v], ",") * "]"
lwe_r(l) = (BigInt.(value.(l.a)), BigInt(value(l.b)))
lwe_Q(l) = (BigInt.(value.(l.a)), BigInt(value(l.b)))

function make(n::Int, outdir::String)
    rng = MersenneTwister(1000 + n)
    params = Params(n)
    key = PrivateKey(params, rng)
    bkey = BootstrapKey(rng, key)
    canon = key_words(bkey)
    keyfile = "julia_p$(n)_key.bin"
    open(joinpath(outdir, keyfile), "w") do io
        write(io, canon)                                   # little-endian host assumed
    end
    key_sha = bytes2hex(sha256(reinterpret(UInt8, canon)))

    message = rand(rng, Bool, params.n)
    enc_bits = split_ciphertext(encrypt(key, rng, message))
    cases = String[]
    for i in 1:4                                           # bit pairs (1,2), (3,4), ... as test/api.test.jl:61-68
        b1, b2 = enc_bits[2i - 1], enc_bits[2i]
        e_and, e_or, e_xor = bootstrap(bkey, nothing, b1, b2)
        @assert [decrypt(key, e) for e in (e_and, e_or, e_xor)] ==
                [message[2i - 1] & message[2i], message[2i - 1] | message[2i], xor(message[2i - 1], message[2i])]
        raw = _bootstrap_internal(bkey, nothing, b1, b2)
        a1, bb1 = lwe_r(b1.lwe)
        a2, bb2 = lwe_r(b2.lwe)
        out = [vcat(lwe_r(e.lwe)[1], [lwe_r(e.lwe)[2]]) for e in (e_and, e_or, e_xor)]
        rawv = [vcat(lwe_Q(l)[1], [lwe_Q(l)[2]]) for l in raw]
        push!(cases, "{\"bits\":[$(Int(message[2i - 1])),$(Int(message[2i]))]," *
                     "\"lwe1\":{\"a\":$(ints(a1)),\"b\":$(bb1)},\"lwe2\":{\"a\":$(ints(a2)),\"b\":$(bb2)}," *
                     "\"out\":[" * join(ints.(out), ",") * "]," *
                     "\"raw\":[" * join(strs.(rawv), ",") * "]}")
    end
    sk = Int.(value.(key.key.coeffs))
    json = "{\"generated_by\":\"julia/make_fixtures.jl: SGFHE.jl reference, rng = nothing\"," *
           "\"julia_version\":\"$(VERSION)\"," *
           "\"params\":{\"n\":$(Int(params.n)),\"r\":$(Int(params.r)),\"m\":$(Int(params.m))," *
           "\"Q\":\"$(BigInt(params.Q))\",\"B\":\"$(BigInt(params.B))\",\"DQ_tilde\":\"$(BigInt(params.DQ_tilde))\"}," *
           "\"sk\":$(ints(sk)),\"key_file\":\"$(keyfile)\",\"key_sha256\":\"$(key_sha)\"," *
           "\"cases\":[" * join(cases, ",") * "]}"
    open(joinpath(outdir, "julia_p$(n).json"), "w") do io
        write(io, json)
    end
    println("Params($n): wrote julia_p$(n).json and $(keyfile) ($(length(canon) * 8) bytes)")
end

outdir = length(ARGS) >= 1 ? ARGS[1] : "."
# Params(1024) -- the parameter set of the headline benchmark (src/fhe.jl:43-97: Q 86.25 bits) -- is in the default
# list since round 5: BootstrapKey(rng, key) takes a while there and the key file is 1 GiB, but it is the ring every
# throughput number of this repository is quoted on.
ns = length(ARGS) >= 2 ? parse.(Int, ARGS[2:end]) : [64, 512, 1024]
for n in ns
    make(n, outdir)
end
