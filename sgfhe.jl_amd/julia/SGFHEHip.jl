# SGFHEHip.jl -- Julia-side binding of libsgfhe_hip.so (include/sgfhe_hip.h).
#
# Drop-in for the bootstrap path of nucypher/SGFHE.jl: `HipBootstrapKey(bkey)` uploads an
# existing `SGFHE.BootstrapKey` to the GPU once; `SGFHE.bootstrap(hkey, nothing, bit1, bit2)`
# then runs on the MI355X and returns the same three `EncryptedBit`s, bit for bit, as
# `SGFHE.bootstrap(bkey, nothing, bit1, bit2)` (src/fhe.jl:608-621).  A vector method batches
# independent gates into one call.
#
# NOT EXECUTED in the build container (no julia binary, no DarkIntegers.jl there); the Python
# ctypes binding in ../engine.py exercises the identical C entry points in the tests.

module SGFHEHip

using SGFHE
using SGFHE: Params, BootstrapKey, EncryptedBit, LWE
using DarkIntegers
using DarkIntegers: ModUInt, value, _verbatim
using Random: AbstractRNG

const libsgfhe_hip = get(ENV, "SGFHE_HIP_LIB", "libsgfhe_hip.so")

# Revision of include/sgfhe_hip.h these ccalls were written for (SGFHE_ABI_VERSION): a library
# built from another revision is refused when the module loads.
const ABI_VERSION = UInt32(7)

function __init__()
    got = ccall((:sgfhe_abi_version, libsgfhe_hip), UInt32, ())
    got == ABI_VERSION ||
        error("$libsgfhe_hip implements ABI revision $got, SGFHEHip.jl needs $ABI_VERSION")
end

# struct sgfhe_params (include/sgfhe_hip.h)
struct CParams
    n::UInt64
    r::UInt64
    m::UInt64
    ell::UInt64
    Q::NTuple{2,UInt64}
    B::NTuple{2,UInt64}
    DQ_tilde::NTuple{2,UInt64}
end

# little-endian 64-bit words of a residue / modulus below 2^128, whatever unsigned type holds it
# (UInt64, UInt128, or an MLUInt chosen with Params(n; rlwe_type = ...), src/fhe.jl:43,79-81)
function words(x)
    b = BigInt(x)
    (UInt64(b & typemax(UInt64)), UInt64(b >> 64))
end

function check(ctx::Ptr{Cvoid}, rc::Int32)
    rc == 0 && return
    msg = unsafe_string(ccall((:sgfhe_last_error_string, libsgfhe_hip), Cstring, (Ptr{Cvoid},), ctx))
    error("sgfhe_hip error $rc: $msg")
end

const CTX_RANDOM_FLATTEN = UInt32(1)
const CTX_DETERMINISTIC_ONLY = UInt32(2)

function create_ctx(p::Params, device::Integer, random_flatten::Bool, deterministic_only::Bool=false)
    cp = Ref(CParams(p.n, p.r, p.m, 2, words(p.Q), words(p.B), words(p.DQ_tilde)))
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:sgfhe_ctx_create_ex, libsgfhe_hip), Int32,
               (Ref{CParams}, Cint, UInt32, Ref{Ptr{Cvoid}}), cp, device,
               (random_flatten ? CTX_RANDOM_FLATTEN : UInt32(0)) |
               (deterministic_only ? CTX_DETERMINISTIC_ONLY : UInt32(0)), ctx)
    check(ctx[], rc)
    ctx
end

# One caller's share of a key: a ctx of its own (the key's first ctx, or a clone of it: same device key and
# constants, own work buffers and streams -- sgfhe_ctx_clone, ABI revision 7) and its result buffer.
mutable struct Slot
    ctx::Ptr{Cvoid}
    # result words of the last batched call, kept and regrown on demand: releasing a
    # multi-megabyte array between calls can stall the next call's kernels (include/sgfhe_hip.h)
    scratch::Vector{UInt64}
end

# The reference call is pure (src/fhe.jl:608-621): tasks may run bootstrap(bkey, ...) on one key side by
# side.  Calls on one ctx are serialised, so the key hands every call a slot of its own: the first slot
# is the ctx that holds the key, further ones are clones made on demand, up to `max_slots` (tasks beyond
# that wait for a slot).  A call keeps its slot from the choice of the flatten mode to the last read of
# its result words.  Tasks on different Julia threads (Threads.@spawn) then call the library at the same
# time, and the library runs the small calls that arrive together as ONE launch chain (sgfhe_set_coalesce:
# eight tasks of one gate get 5.8 x the rate of one at Params(1024), each the bytes of its call made alone,
# with an rng too -- every slot draws from its own stream); a ccall blocks its thread, so tasks of ONE
# thread still take turns.
mutable struct HipBootstrapKey
    params::Params
    ctx::Ptr{Cvoid}              # the ctx the key was uploaded to (slot 1)
    slots::Vector{Slot}          # every slot made so far
    free::Channel{Slot}          # the ones not in use
    max_slots::Int
    lock::ReentrantLock          # guards `slots` (slot creation)

    # Both `rng = nothing` and `rng::AbstractRNG` calls work with every key (ABI revision 6: at
    # Params(1024) the engine keeps a basis per flatten mode).  `random_flatten` is kept for callers
    # written against revision 5 and has no effect; `deterministic_only = true` asks for the smaller
    # basis only (SGFHE_CTX_DETERMINISTIC_ONLY: rng calls are then refused at Params(1024)).
    function HipBootstrapKey(bkey::BootstrapKey; device::Integer=0, random_flatten::Bool=false,
                             deterministic_only::Bool=false, max_slots::Integer=max(1, Threads.nthreads()))
        p = bkey.params
        ctx = create_ctx(p, device, random_flatten, deterministic_only)
        # value.(p.coeffs) of every polynomial, order [k][row][col][coef], 2 x UInt64 each
        # (BootstrapKey.key is a Vector of 4x2 Matrix{Polynomial}, src/fhe.jl:176-201).
        canon = Vector{UInt64}(undef, p.n * 8 * p.m * 2)
        pos = 1
        for k in 1:p.n, row in 1:4, col in 1:2
            for c in bkey.key[k][row, col].coeffs
                lo, hi = words(value(c))
                canon[pos] = lo; canon[pos+1] = hi
                pos += 2
            end
        end
        rc = ccall((:sgfhe_bkey_upload, libsgfhe_hip), Int32,
                   (Ptr{Cvoid}, Ptr{UInt64}, Csize_t), ctx[], canon, length(canon))
        check(ctx[], rc)
        with_slots(new(p, ctx[], Slot[], Channel{Slot}(Int(max_slots)), Int(max_slots), ReentrantLock()))
    end

    # raw constructor used by the generate-on-device method below
    HipBootstrapKey(params::Params, ctx::Ptr{Cvoid}, max_slots::Integer) =
        with_slots(new(params, ctx, Slot[], Channel{Slot}(Int(max_slots)), Int(max_slots), ReentrantLock()))
end

# first slot = the key's own ctx; the finalizer destroys every ctx (clones and owner in any order:
# the shared device memory goes with the last of them)
function with_slots(key::HipBootstrapKey)
    s = Slot(key.ctx, UInt64[])
    push!(key.slots, s)
    put!(key.free, s)
    finalizer(key) do k
        for sl in k.slots
            ccall((:sgfhe_ctx_destroy, libsgfhe_hip), Int32, (Ptr{Cvoid},), sl.ctx)
        end
    end
    key
end

# A slot for the duration of `f`: a free one, else a new clone while fewer than max_slots exist, else wait.
function with_slot(f, key::HipBootstrapKey)
    lock(key.lock) do
        if !isready(key.free) && length(key.slots) < key.max_slots
            c = Ref{Ptr{Cvoid}}(C_NULL)
            rc = ccall((:sgfhe_ctx_clone, libsgfhe_hip), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), key.ctx, c)
            if rc != 0
                msg = unsafe_string(ccall((:sgfhe_last_error_string, libsgfhe_hip), Cstring, (Ptr{Cvoid},),
                                          c[] == C_NULL ? key.ctx : c[]))
                c[] == C_NULL || ccall((:sgfhe_ctx_destroy, libsgfhe_hip), Int32, (Ptr{Cvoid},), c[])
                error("sgfhe_hip error $rc: $msg")
            end
            fresh = Slot(c[], UInt64[])
            push!(key.slots, fresh)
            put!(key.free, fresh)
        end
    end
    s = take!(key.free)
    try
        f(s)
    finally
        put!(key.free, s)
    end
end

# Vector{ModUInt{UInt64, r}} is an isbits array: reinterpret is a zero-copy n x UInt64 view
# (src/fhe.jl:206-209,272-274).
lwe_words(bits::AbstractVector{EncryptedBit}, n) =
    (reduce(vcat, [reinterpret(UInt64, b.lwe.a) for b in bits]),
     UInt64[reinterpret(UInt64, [b.lwe.b])[1] for b in bits])

# rng = nothing: deterministic flatten (src/utils.jl:155-189), bit-exact with the CPU path.
# rng::AbstractRNG: randomised flatten (src/utils.jl:198-241) from a device ChaCha8 counter stream
# keyed with 32 bytes of `rng` -- same distribution, not the same stream as the CPU path.
# The 32 bytes are drawn from the caller's rng BEFORE a slot is taken (flatten_key), so the rng is touched by
# the calling task only and in the order of its calls.
flatten_key(rng) = rng === nothing ? nothing : rand(rng, UInt8, 32)
function set_flatten_mode(ctx::Ptr{Cvoid}, key::Union{Vector{UInt8},Nothing})
    check(ctx, ccall((:sgfhe_set_random_flatten_key, libsgfhe_hip), Int32,
                     (Ptr{Cvoid}, Cint, Ptr{UInt8}), ctx, key === nothing ? 0 : 1,
                     key === nothing ? zeros(UInt8, 32) : key))
end

"""
    bootstrap(hkey, rng, enc_bits1, enc_bits2)

Batched gate bootstrap on the GPU; returns a vector of (AND, OR, XOR) triples.
"""
function SGFHE.bootstrap(hkey::HipBootstrapKey, rng::Union{AbstractRNG,Nothing},
                         bits1::AbstractVector{EncryptedBit}, bits2::AbstractVector{EncryptedBit})
    p = hkey.params
    n = p.n
    batch = length(bits1)
    @assert length(bits2) == batch
    a1, b1 = lwe_words(bits1, n)
    a2, b2 = lwe_words(bits2, n)
    # the reference's own constructor form: reduce_modulus ends in
    # `new_repr(convert.(new_base_type, x_rs), new_modulus, DarkIntegers._verbatim)` with
    # new_repr = ModUInt, new_base_type = UInt64, new_modulus = params.r (src/utils.jl:116, src/fhe.jl:616-618)
    mk(x::UInt64) = ModUInt(x, p.r, _verbatim)
    res = Vector{NTuple{3,EncryptedBit}}(undef, batch)
    fkey = flatten_key(rng)
    with_slot(hkey) do slot
        set_flatten_mode(slot.ctx, fkey)          # mode and call stay together on the slot's own ctx
        length(slot.scratch) < batch * 3 * (n + 1) && resize!(slot.scratch, batch * 3 * (n + 1))
        out = slot.scratch
        rc = ccall((:sgfhe_bootstrap_batch, libsgfhe_hip), Int32,
                   (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt64}, Csize_t,
                    Ptr{UInt64}, UInt32),
                   slot.ctx, a1, b1, a2, b2, batch, out, 0)
        check(slot.ctx, rc)
        for t in 1:batch
            base = (t - 1) * 3 * (n + 1)
            res[t] = ntuple(3) do g
                o = base + (g - 1) * (n + 1)
                EncryptedBit(LWE(mk.(out[o+1:o+n]), mk(out[o+n+1])))
            end
        end
    end
    res
end

"""
    HipBootstrapKey(params, sk::PrivateKey, seed::Vector{UInt8}; device=0, random_flatten=false)

Generates the bootstrap key on the GPU (BootstrapKey(rng, sk), src/fhe.jl:181-201) from a
32-byte seed (ChaCha20 streams on the device): `rand(RandomDevice(), UInt8, 32)` for real keys.
"""
function HipBootstrapKey(params::Params, sk::SGFHE.PrivateKey, seed::Vector{UInt8};
                         device::Integer=0, random_flatten::Bool=false,
                         max_slots::Integer=max(1, Threads.nthreads()))
    @assert length(seed) == 32
    ctx = create_ctx(params, device, random_flatten)
    bits = UInt64[UInt64(value(c)) for c in sk.key.coeffs]
    check(ctx[], ccall((:sgfhe_bkey_generate, libsgfhe_hip), Int32,
                       (Ptr{Cvoid}, Ptr{UInt64}, Csize_t, Ptr{UInt8}, UInt32),
                       ctx[], bits, length(bits), seed, UInt32(params.n)))
    HipBootstrapKey(params, ctx[], max_slots)
end

"""
    pack_encrypted_bits(hkey, rng, enc_bits)

src/fhe.jl:660-696 on the GPU: n EncryptedBits -> one RLWE Ciphertext; `rng` selects the flatten
mode exactly as for `bootstrap`.
"""
function SGFHE.pack_encrypted_bits(hkey::HipBootstrapKey, rng::Union{AbstractRNG,Nothing},
                                   enc_bits::AbstractVector{EncryptedBit})
    p = hkey.params
    @assert length(enc_bits) == p.n
    a, b = lwe_words(enc_bits, p.n)
    w = Vector{UInt64}(undef, p.m)
    v = Vector{UInt64}(undef, p.m)
    fkey = flatten_key(rng)
    with_slot(hkey) do slot
        set_flatten_mode(slot.ctx, fkey)
        rc = ccall((:sgfhe_pack_encrypted_bits, libsgfhe_hip), Int32,
                   (Ptr{Cvoid}, Ptr{UInt64}, Ptr{UInt64}, Csize_t, Ptr{UInt64}, Ptr{UInt64}),
                   slot.ctx, a, b, 1, w, v)
        check(slot.ctx, rc)
    end
    mk(x::UInt64) = ModUInt(x, p.r, _verbatim)     # as at src/utils.jl:116 (reduce_modulus of src/fhe.jl:692-693)
    SGFHE.Ciphertext(p, SGFHE.RLWE(Polynomial(mk.(w), negacyclic_modulus),
                                   Polynomial(mk.(v), negacyclic_modulus)))
end

# The drop-in: same signature as src/fhe.jl:608-610, batch of one.
SGFHE.bootstrap(hkey::HipBootstrapKey, rng::Union{AbstractRNG,Nothing},
                bit1::EncryptedBit, bit2::EncryptedBit) =
    SGFHE.bootstrap(hkey, rng, [bit1], [bit2])[1]

end # module
