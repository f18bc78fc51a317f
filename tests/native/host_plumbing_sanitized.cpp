// Sanitizer build of csrc/host_plumbing.h (the host side of rows N3 / N4): compiled by
// tests/test_host_sanitizer.py with g++ -fsanitize=address,undefined and run on the CPU.  Exercises
// every function on sizes around its internal boundaries (the SHAKE rate, both branches of extract,
// empty rows of the convolutions); AddressSanitizer / UBSan abort on any out-of-bounds access, use
// of uninitialised stack, signed overflow or misaligned access.  Prints a digest of all outputs so
// that the Python side can compare it with its own computation.
#include <stdio.h>
#include <stdlib.h>

#include "host_plumbing.h"

static uint64_t lcg(uint64_t &s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return s >> 11;
}

int main() {
    using namespace sgfhe_host;
    uint64_t seed = 12345, digest = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { digest = (digest ^ v) * 1099511628211ull; };

    // SHAKE-256: input lengths around the 136-byte rate, output lengths around it as well
    for (size_t inlen : {0u, 1u, 135u, 136u, 137u, 271u, 272u, 273u, 1000u})
        for (size_t outlen : {1u, 32u, 135u, 136u, 137u, 500u}) {
            std::vector<uint8_t> in(inlen ? inlen : 1), out(outlen);
            for (size_t i = 0; i < inlen; i++) in[i] = (uint8_t)lcg(seed);
            shake256(in.data(), inlen, out.data(), outlen);
            for (uint8_t b : out) mix(b);
        }

    for (size_t n : {8u, 64u, 512u}) {
        const uint64_t r = 16 * n, rmask = r - 1;
        std::vector<uint64_t> a(n), s(n), out(n), big(8 * n), ex(n);
        for (auto &x : a) x = lcg(seed) & rmask;
        for (auto &x : s) x = lcg(seed) & 1;
        negacyclic_mul(a.data(), s.data(), n, rmask, out.data());
        for (uint64_t v : out) mix(v);
        std::fill(s.begin(), s.end(), 0);                       // all-zero key: every row skipped
        negacyclic_mul(a.data(), s.data(), n, rmask, out.data());
        for (uint64_t v : out) mix(v);
        for (auto &x : big) x = lcg(seed) & rmask;
        for (size_t i : {(size_t)1, (size_t)2, n - 1, n, n + 1, 8 * n}) {   // both branches (1-based i)
            extract(big.data(), 8 * n, i, n, rmask, ex.data());
            for (uint64_t v : ex) mix(v);
        }
        // bits
        const size_t t = 5;
        std::vector<uint8_t> bits(t * n), back(t * n);
        for (auto &b : bits) b = (uint8_t)(lcg(seed) & 1);
        packbits(bits.data(), t, n, out.data());
        unpackbits(out.data(), n, t, back.data());
        if (bits != back) { fprintf(stderr, "packbits / unpackbits round trip\n"); return 1; }
        std::vector<uint8_t> seq(n);
        for (auto &b : seq) b = (uint8_t)(lcg(seed) & 1);
        for (size_t factor : {(size_t)1, (size_t)9, (size_t)14}) {
            prng_expand(seq.data(), n, factor, out.data());
            for (uint64_t v : out) {
                if (v >> factor) { fprintf(stderr, "prng_expand out of range\n"); return 1; }
                mix(v);
            }
        }
        // the public-key side: short convolution modulo q and rescale at its extremes
        const uint64_t q = n == 8 ? 257 : n == 64 ? 65537 : 4205569;     // 2 n | q - 1
        std::vector<int8_t> u(n);
        std::vector<int64_t> ks(n);
        for (auto &x : a) x = lcg(seed) % q;
        a[0] = q - 1;
        for (auto &x : u) x = (int8_t)((int)(lcg(seed) % 3) - 1);
        negacyclic_mul_short(a.data(), u.data(), n, q, ks.data());
        for (int64_t v : ks) {
            if (v < 0 || (uint64_t)v >= q) { fprintf(stderr, "negacyclic_mul_short out of range\n"); return 1; }
            mix((uint64_t)v);
        }
        std::fill(u.begin(), u.end(), (int8_t)-1);               // the most negative sums
        std::fill(a.begin(), a.end(), q - 1);
        negacyclic_mul_short(a.data(), u.data(), n, q, ks.data());
        for (int64_t v : ks) mix((uint64_t)v);
        for (uint64_t x : {(uint64_t)0, (uint64_t)1, q / 2, q / 2 + 1, q - 1}) {
            mix(rescale(r, x, q, true));
            mix(rescale(r, x, q, false));
            mix(rescale(r >> 3, x, q, false));
        }
        if (rescale(r, q - 1, q, true) != 0) { fprintf(stderr, "rescale: the rounded value r must wrap to 0\n"); return 1; }
    }
    printf("%016llx\n", (unsigned long long)digest);
    return 0;
}
