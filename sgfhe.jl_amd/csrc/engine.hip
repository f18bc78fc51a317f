// Host side of libsgfhe_hip.so: context set-up (RNS primes, twiddle tables, CRT / flatten
// constants), bootstrap-key upload, lock-step batch scheduling, and the C ABI of
// include/sgfhe_hip.h.  Reference citations are relative to /root/reference.
#include "../../include/sgfhe_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "kernels.h"
#include "host_plumbing.h"
#include "coalescer.h"

using namespace sgfhe;

namespace {

// ---------------------------------------------------------------- host number theory (word size)

uint32_t mulmod32(uint32_t a, uint32_t b, uint32_t p) { return (uint32_t)((uint64_t)a * b % p); }
uint32_t powmod32(uint32_t a, uint64_t e, uint32_t p) {
    uint32_t r = 1 % p;
    a %= p;
    while (e) {
        if (e & 1) r = mulmod32(r, a, p);
        a = mulmod32(a, a, p);
        e >>= 1;
    }
    return r;
}
bool is_prime32(uint32_t x) {
    if (x < 2) return false;
    for (uint32_t q : {2u, 3u, 5u, 7u, 11u, 13u}) {
        if (x % q == 0) return x == q;
    }
    uint32_t d = x - 1;
    int s = 0;
    while (!(d & 1)) { d >>= 1; s++; }
    for (uint32_t a : {2u, 3u, 5u, 7u}) {  // deterministic below 3,215,031,751
        uint32_t y = powmod32(a, d, x);
        if (y == 1 || y == x - 1) continue;
        bool comp = true;
        for (int i = 0; i < s - 1; i++) {
            y = mulmod32(y, y, x);
            if (y == x - 1) { comp = false; break; }
        }
        if (comp) return false;
    }
    return true;
}
uint32_t bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
inline u128 ld128(const uint64_t *w) { return ((u128)w[1] << 64) | w[0]; }
inline int32_t centre32(uint32_t x, uint32_t p) { return x > (p - 1) / 2 ? (int32_t)x - (int32_t)p : (int32_t)x; }
double u128_log2(u128 x) { return log2((double)(uint64_t)(x >> 64) * 18446744073709551616.0 + (double)(uint64_t)x); }
double u128_dbl(u128 x) { return (double)(uint64_t)(x >> 64) * 18446744073709551616.0 + (double)(uint64_t)x; }

}  // namespace

// ---------------------------------------------------------------- ctx

// Device memory that a ctx and its clones (sgfhe_ctx_clone) have in common: the per-prime records and
// CRT / flatten constants of every basis, the twiddle tables, and the key in every form -- everything a
// bootstrap only reads.  Each allocation is registered here when it is made and freed when the last ctx
// that holds the block goes, so a clone may outlive the ctx it was taken from.
// (struct Coalescer -- the gathering of small calls across the ctxs that share a key -- is csrc/coalescer.h)
struct SharedDev {
    int device = 0;
    std::mutex mu;
    std::vector<void *> mem;
    Coalescer co;
    std::atomic<unsigned> clones{0};     // clones made of this block so far (their streams are shifted by it)
    void own(void *p) {
        std::lock_guard<std::mutex> g(mu);
        mem.push_back(p);
    }
    ~SharedDev() {
        (void)hipSetDevice(device);
        for (void *p : mem) (void)hipFree(p);
    }
};

struct sgfhe_ctx {
    sgfhe_params par;
    int device = 0;
    int logm = 0;
    uint32_t M = 0, n = 0;
    u128 Q = 0, B = 0;
    // ---- RNS bases ------------------------------------------------------------------------------
    // A ctx has one basis of word-size primes, or two: the deterministic flatten needs 5 m B Q below
    // the product of the primes, the randomised one 20 m B Q, and where that takes one prime more
    // (Params(1024): five against six) the ctx keeps a basis for each mode -- the key in both forms
    // (the form of the smaller basis is the larger one's times a constant per prime, derived on the
    // device), constants for both -- so that `bootstrap(bkey, rng, ...)` works on every ctx and the
    // deterministic mode never pays for the extra prime.  The fields below (npr ... pack_G_rnd,
    // d_key) are the ACTIVE basis: a copy of basis[active], switched by sgfhe_set_random_flatten
    // (activate()); everything that launches kernels reads them.
    struct Basis {
        uint32_t npr = 0;
        uint32_t primes[NPR_MAX] = {};
        PrimeK *d_primes = nullptr;
        CrtConst *d_crt = nullptr;
        CrtConst h_crt;
        CrtLean *d_lean = nullptr;
        CrtLean h_lean;
        bool lean_rnd_ok = false;
        uint32_t pack_G = 1, pack_G_rnd = 0;
        int32_t *d_key = nullptr;
        size_t key_bytes = 0;
    } basis[2];
    int nb = 1, active = 0;       // bases in use (basis[0]: deterministic mode, basis[nb - 1]: randomised), active one
    uint32_t npr_max = 0;         // primes of the larger basis: work buffers are sized for it
    uint32_t tw_done = 0;         // primes whose twiddle tables are on the device
    uint32_t primes[NPR_MAX];
    uint32_t npr = 0;  // RNS primes in use: the fewest whose product covers the exactness bound
    std::string err;
    // the read-only device memory of this ctx, shared with its clones (use_count() > 1: the key may not
    // be replaced, sgfhe_ctx_clone)
    std::shared_ptr<SharedDev> shared;
    // Every C-ABI entry point that takes a ctx holds this lock for the whole call: a ctx may be
    // shared by host threads (their calls are serialised), different ctxs run concurrently.
    mutable std::recursive_mutex mu;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // second lane: chunk i+1 overlaps its memory-bound k_crt_acc
                                    // with the VALU-bound k_extprod of chunk i
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // Calls on one ctx are ordered on the device whatever stream each names: every call that queues
    // work first makes its stream wait for ev_done, the end of the previous call's work, and records
    // it again when its own work is queued (the lanes' work buffers, the key and the constants are
    // shared by all calls).  sgfhe_sync and everything that frees or replaces a buffer wait for it on
    // the host.
    hipEvent_t ev_done = nullptr;
    bool pending = false;           // ev_done has been recorded and may not have completed
    // copies of the host-pointer entry point, beside the lanes' kernels: uploads and downloads on
    // streams of their own (on one stream the upload of the next chunks would queue behind the
    // download of the present ones, which waits for their last kernel)
    hipStream_t stream_io = nullptr, stream_io2 = nullptr;
    std::vector<hipEvent_t> ev_pool;   // events of the pipelined host-pointer path (created once, reused)
    // device constants
    PrimeK *d_primes = nullptr;
    CrtConst *d_crt = nullptr;
    uint32_t *d_bad = nullptr;  // set by k_key_transform when a key residue is >= Q
    int32_t *d_tw = nullptr;  // npr * 4 * M entries
    int32_t *d_twq = nullptr; // quarter form of the latency kernels: per prime 4 blocks [f_q | v_q | fp_q | vp_q] of m / 4 words
    int32_t *d_pow = nullptr; //   and psi^e * R mod p, e in [0, 2 m)
    int32_t tw_head[NPR_MAX][8] = {};   // f[1..3], fp[2], fp[3], v[1..3] of every prime (PrimeK::f1 ... v3)
    // the quarter form's two transform kernels as one launch (k_ext_quarter) from this many gates per chain, up to
    // what fills the device once (fused_cap below); SGFHE_SMALL_FUSED=0: never, =n: from n gates
    uint32_t fused_min = 7;
    bool iter_all = false;     // -DSGFHE_WITH_ITER_ALL builds, SGFHE_ITER_ALL=1: one launch per iteration (prototype)
    uint32_t split_max = 7;   // calls of at most this many gates take the quarter form (SGFHE_SMALL_SPLIT, 0 = never):
                              // 1 / 2 / 4 / 6 / 8 gates 15.1 / 15.9 / 17.8 / 19.7 / 23.5 ms against 17.9 / 18.6 / 20.3 /
                              // 22.0 / 23.3 with one workgroup per transform (profiles/r04_exp_quarter.txt)
    CrtConst h_crt;
    CrtLean *d_lean = nullptr;  // constants of k_crt_lean (nl = 0: this parameter set keeps k_crt_acc)
    CrtLean h_lean;
    bool lean_rnd_ok = false;   // k_crt_lean_rnd's width bounds hold for this parameter set (build_constants)
    uint32_t pack_G = 1, pack_G_rnd = 0;  // key slices per exact-accumulation group of the packing path
                                          // (deterministic / randomised flatten; 0 = not available)
    // key
    int32_t *d_key = nullptr;
    size_t key_bytes = 0;
    bool have_key = false;
    // work buffers: two lanes, each sized for `cap` bootstraps
    uint32_t chunk = 0, cap = 0, lanes = 2;
    // randomised flatten (rng != nothing, utils.jl:198-241)
    bool rnd = false, rnd_ok = false;
    ChaChaKey rnd_key = {};   // 32-byte key of the draw stream (sgfhe_set_random_flatten[_key])
    uint32_t rnd_call = 0, last_call = 0;
    int64_t call_fixed = -1;              // the number of the next call, when the coalescer has already assigned it
    const RndRow *gather_rows = nullptr;  // set around a gathered randomised call this ctx leads: d_rows
    RndRow *d_rows = nullptr;             // per-row draw streams of such a call (device), for up to rows_cap rows
    size_t rows_cap = 0;
    std::vector<RndRow> h_rows;
    uint32_t create_flags = 0;
    // RNS2Number form of Z_Q (src/rns.jl): set by sgfhe_bkey_upload_rns2 / sgfhe_rns2_convert
    bool have_rns2 = false;
    Rns2Const rns2;
    struct Lane {
        uint64_t *dig = nullptr;
        uint32_t *yres = nullptr;
        uint32_t *ua = nullptr;
        int32_t *zpart = nullptr;  // small-batch form only: [cap_small][npr][4][2][m]
    } lane[2];
    bool use_lean = true;     // SGFHE_CRT_LEAN=0 in the environment: keep k_crt_acc2 (A/B measurements)
    bool small_padded = false;  // SGFHE_SMALL_PADDED=1: small-batch grids padded to 8 bootstraps as up to round 3 (A/B)
    bool small_lanes = true;    // SGFHE_SMALL_LANES=0: a call of a few gates as one chunk on one stream (A/B)
    uint32_t small_lanes_max = 24;   // ... up to this many gates (SGFHE_SMALL_LANES=<n>)
    uint32_t crt1_max = 0;      // k_crt_lean launches of at most this many coefficients take one per thread
                                // (SGFHE_CRT1_GATES gates' worth: default 8; 0 = never.  Same call, 1 / 2 / 4 / 8
                                // gates: 18.6 / 19.4 / 20.8 / 23.9 ms with four per thread, 17.9 / 18.6 / 20.2 / 23.4
                                // with one, profiles/r04_exp_crt1.txt)
    uint32_t small_max = 24;  // chunks of at most this many bootstraps take the small-batch form
                              // (measured crossover at Params(1024): 24 -> 32.3 vs 40.9 ms, 32 -> 44.0 vs 42.3 ms)
    // staging buffers of the host-pointer entry point (sgfhe_bootstrap_batch: the drop-in signature of
    // fhe.jl:608-610, usually called with one gate): grown on demand and kept, so that a call does
    // not pay a hipMalloc / hipFree pair (hipFree synchronises the device)
    uint64_t *io_in = nullptr, *io_out = nullptr;
    size_t io_in_words = 0, io_out_words = 0;
    // ... and their page-locked host mirrors.  hipMemcpyAsync on pageable memory above about 1 MB
    // pins the caller's pages for the copy, a fixed cost of several milliseconds per copy
    // (profiles/r03_latency.txt: 256 gates 155.6 ms through host pointers against 127.6 ms from
    // device buffers, 64 gates 43.6 against 43.4); below PIN_MAX_BYTES per buffer the engine
    // copies through its own pinned buffers instead (one CPU memcpy + one true DMA).
    uint64_t *pin_in = nullptr, *pin_out = nullptr;
    size_t pin_in_words = 0, pin_out_words = 0;
    std::vector<uint64_t> co_buf;   // gathered inputs and results of a combined call this ctx leads (Coalescer)
    bool use_pin = true;   // SGFHE_HOST_PIN=0 in the environment: direct copies (A/B measurements)
    // timing
    bool timing = false;
    struct EvTriple { hipEvent_t e0, e1, e2; };  // ext = e0 -> e1, crt = e1 -> e2
    std::vector<EvTriple> ev;
    double t_ext = 0, t_crt = 0;
    uint64_t n_ext = 0, n_crt = 0;
    uint32_t last_chunk = 0;
    // whole bootstrap_device calls (fork ... join): the k-loop's wall time on the device when the
    // kernels of two lanes overlap and per-kernel durations no longer add up
    struct EvPair { hipEvent_t e0, e1; uint64_t batch; };
    std::vector<EvPair> ev_call;
    double t_call = 0;
    uint64_t n_call = 0, boots_call = 0;
    // kernels whose dynamic-LDS limit has been raised on this ctx's device (hipFuncSetAttribute is
    // per device; a ctx is bound to one device and used by one host thread)
    uint32_t attr_done = 0;
};
enum : uint32_t { ATTR_EXTPROD = 1u, ATTR_SMALL = 2u, ATTR_SHORTPROD = 4u, ATTR_FUSED = 8u, ATTR_ITER = 16u };
// Page-locked mirrors of the host-pointer entry point's staging buffers: per buffer at most this much
// (a batch of 16384 at Params(1024) needs 268 + 403 MB).  Round 3 staged whole buffers and stopped at
// 48 MB, where one CPU memcpy cost what pinning the caller's pages did; the copies are now pipelined
// chunk by chunk beside the kernels (HostPipe), so their size no longer matters.
static constexpr size_t PIN_MAX_BYTES = (size_t)1 << 30;
// SGFHE_PIN_MAX_MB=<n> in the environment lowers the limit (read at every host-pointer call): lets a test
// send a small batch down the direct-copy path that batches above 1 GiB take (tests/test_gpu_round5.py).
static size_t pin_max_bytes() {
    const char *env = getenv("SGFHE_PIN_MAX_MB");
    if (env && *env) {
        const long mb = atol(env);
        if (mb >= 0 && (size_t)mb < (PIN_MAX_BYTES >> 20)) return (size_t)mb << 20;
    }
    return PIN_MAX_BYTES;
}

namespace {

#define SGFHE_LOCK(ctx) std::lock_guard<std::recursive_mutex> lock_((ctx)->mu)
// A synchronous entry point that touches the key, the constants or the lanes' buffers on the ctx
// stream first waits (on the host) for the work of earlier asynchronous calls.
#define SGFHE_QUIESCE(ctx)                                                                        \
    do {                                                                                          \
        int32_t rq_ = drain(ctx);                                                                 \
        if (rq_) return rq_;                                                                      \
    } while (0)

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            char buf_[512];                                                                       \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),    \
                     __FILE__, __LINE__);                                                         \
            (ctx)->err = buf_;                                                                    \
            return SGFHE_ERR_HIP;                                                                 \
        }                                                                                         \
    } while (0)

int32_t fail(sgfhe_ctx *ctx, int32_t code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    return code;
}

// Start of a call that queues work on `st`: that work begins after everything earlier calls queued.
int32_t fence_begin(sgfhe_ctx *c, hipStream_t st) {
    if (c->pending) HIPCHK(c, hipStreamWaitEvent(st, c->ev_done, 0));
    return SGFHE_OK;
}
// End of such a call: its last work on `st` (the lanes have been joined into it).
int32_t fence_end(sgfhe_ctx *c, hipStream_t st) {
    HIPCHK(c, hipEventRecord(c->ev_done, st));
    c->pending = true;
    return SGFHE_OK;
}
// Host waits for all work queued on the ctx (before buffers are freed or replaced, and in sgfhe_sync).
int32_t drain(sgfhe_ctx *c) {
    if (c->pending) {
        HIPCHK(c, hipEventSynchronize(c->ev_done));
        c->pending = false;
    }
    return SGFHE_OK;
}

// basis[b] -> the active fields
void activate(sgfhe_ctx *c, int b) {
    const sgfhe_ctx::Basis &S = c->basis[b];
    c->active = b;
    c->npr = S.npr;
    memcpy(c->primes, S.primes, sizeof c->primes);
    c->d_primes = S.d_primes;
    c->d_crt = S.d_crt;
    c->h_crt = S.h_crt;
    c->d_lean = S.d_lean;
    c->h_lean = S.h_lean;
    c->lean_rnd_ok = S.lean_rnd_ok;
    c->pack_G = S.pack_G;
    c->pack_G_rnd = S.pack_G_rnd;
    c->d_key = S.d_key;
    c->key_bytes = S.key_bytes;
}
// the active fields (as build_basis / key_alloc leave them) -> basis[b]
void save_basis(sgfhe_ctx *c, int b) {
    sgfhe_ctx::Basis &S = c->basis[b];
    S.npr = c->npr;
    memcpy(S.primes, c->primes, sizeof S.primes);
    S.d_primes = c->d_primes;
    S.d_crt = c->d_crt;
    S.h_crt = c->h_crt;
    S.d_lean = c->d_lean;
    S.h_lean = c->h_lean;
    S.lean_rnd_ok = c->lean_rnd_ok;
    S.pack_G = c->pack_G;
    S.pack_G_rnd = c->pack_G_rnd;
    S.d_key = c->d_key;
    S.key_bytes = c->key_bytes;
}
// the basis of the present flatten mode
int mode_basis(const sgfhe_ctx *c) { return c->rnd ? c->nb - 1 : 0; }
// The quarter form of the latency kernels exists for m >= 4096 where the lean CRT kernel of the flatten mode does
// (both modes; not the three-plane digit records of B >= 2^46, MODE_WIDE).
bool quarter_ok(const sgfhe_ctx *c, uint32_t mode) {
    return c->logm >= 12 && c->split_max && c->use_lean &&
           (mode == 0u ? c->h_lean.nl != 0 : (mode == MODE_RANDOM && c->lean_rnd_ok));
}
// Largest chain the fused quarter kernel takes: one workgroup per (gate, prime, quarter), one round of the
// device's 256 compute units (12 gates on five primes, 10 on six); 0 where it does not exist (m < 4096,
// m = 16384: kernels.h) or is switched off.
uint32_t fused_cap(const sgfhe_ctx *c, uint32_t mode) {
    if (!c->fused_min || c->small_padded || (c->logm != 12 && c->logm != 13) || !quarter_ok(c, mode)) return 0;
    return 256u / (c->npr * 4u);
}
bool fused_takes(const sgfhe_ctx *c, uint32_t cnt, uint32_t mode) { return cnt >= c->fused_min && cnt <= fused_cap(c, mode); }

size_t lds_bytes(int logm, int npoly) { return (size_t)npoly * ((size_t)4 << logm); }
// points per thread of k_extprod: 16, or 8 where 16 would leave half a wavefront idle (m <= 512)
#ifndef SGFHE_EXT_LE3_MAX
#define SGFHE_EXT_LE3_MAX 9
#endif
template <int LOGM> constexpr int ext_loge() { return LOGM <= SGFHE_EXT_LE3_MAX ? 3 : LOGE; }
template <int LOGM> constexpr int threads_of() { return NttGeom<LOGM, LOGE>::T; }

// ---- per-LOGM dispatch ------------------------------------------------------------------------

#define SGFHE_FOR_LOGM(X) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14)

template <int LOGM>
int32_t launch_extprod_t(sgfhe_ctx *c, const sgfhe_ctx::Lane &L, const int32_t *keyk,
                         uint32_t cpad, uint32_t k, uint32_t mode, hipStream_t st) {
    const size_t lds = lds_bytes(LOGM, 2);  // exchange buffer + z1 accumulator
    if (!(c->attr_done & ATTR_EXTPROD)) {
        HIPCHK(c, hipFuncSetAttribute((const void *)k_extprod<LOGM, ext_loge<LOGM>()>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(c, hipFuncSetAttribute((const void *)k_extprod<LOGM, ext_loge<LOGM>(), true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->attr_done |= ATTR_EXTPROD;
    }
    if (mode & MODE_WIDE)   // digit planes with a third plane (randomised flatten, B >= 2^46)
        hipLaunchKernelGGL((k_extprod<LOGM, ext_loge<LOGM>(), true>), dim3(cpad * c->npr),
                           dim3(NttGeom<LOGM, ext_loge<LOGM>()>::T), lds, st, L.dig,
                           keyk, L.yres, L.ua, c->d_primes, k, c->n, mode);
    else
        hipLaunchKernelGGL((k_extprod<LOGM, ext_loge<LOGM>()>), dim3(cpad * c->npr),
                           dim3(NttGeom<LOGM, ext_loge<LOGM>()>::T), lds, st, L.dig,
                           keyk, L.yres, L.ua, c->d_primes, k, c->n, mode);
    HIPCHK(c, hipGetLastError());
    return SGFHE_OK;
}
int32_t launch_extprod(sgfhe_ctx *c, const sgfhe_ctx::Lane &L, const int32_t *keyk, uint32_t cpad,
                       uint32_t k, uint32_t mode, hipStream_t st) {
    switch (c->logm) {
#define X(LM) case LM: return launch_extprod_t<LM>(c, L, keyk, cpad, k, mode, st);
        SGFHE_FOR_LOGM(X)
#undef X
    }
    return fail(c, SGFHE_ERR_UNSUPPORTED, "unsupported log2(m)");
}

#ifdef SGFHE_WITH_ITER_ALL   // prototype, not in the default build (kernels.h)
// One launch per iteration (kernels.h k_iter_all): deterministic flatten, m = 8192, five primes, lean CRT constants.
int32_t launch_iter_all(sgfhe_ctx *c, const sgfhe_ctx::Lane &L, const int32_t *keyk, uint32_t cnt, uint32_t k,
                        hipStream_t st) {
    constexpr int LM = 13;
    const size_t lds = lds_bytes(LM, 2 + SGFHE_IA_LP);   // exchange buffer of the inverse pair + the private accumulator planes
    if (!(c->attr_done & ATTR_ITER)) {
        HIPCHK(c, hipFuncSetAttribute((const void *)k_iter_all<LM, 5, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->attr_done |= ATTR_ITER;
    }
    hipLaunchKernelGGL((k_iter_all<LM, 5, 3>), dim3(cnt), dim3(NttGeom<LM, 3>::T), lds, st, L.dig, keyk, L.ua,
                       c->d_primes, c->d_lean, k, c->n);
    HIPCHK(c, hipGetLastError());
    return SGFHE_OK;
}
#endif

template <int LOGM>
int32_t launch_small_t(sgfhe_ctx *c, const sgfhe_ctx::Lane &L, const int32_t *keyk, uint32_t cpad,
                       uint32_t k, uint32_t mode, hipStream_t st) {
    // 8 points per thread while m / 8 threads fit a workgroup, else the engine's 16
    constexpr int LE = (LOGM - 3 <= 10 && LOGM >= 9) ? 3 : LOGE;
    constexpr int TH = NttGeom<LOGM, LE>::T;
    const size_t lds = lds_bytes(LOGM, 1);
    if (!(c->attr_done & ATTR_SMALL)) {
        HIPCHK(c, hipFuncSetAttribute((const void *)k_fwd_phase<LOGM, LE>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(c, hipFuncSetAttribute((const void *)k_fwd_phase<LOGM, LE, true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(c, hipFuncSetAttribute((const void *)k_inv_column<LOGM, LE>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->attr_done |= ATTR_SMALL;
    }
    // (Round 4 measured the per-prime records passed by value in the kernel arguments instead of
    // through this pointer -- one dependent load less at the head of each launch: 24.75 against
    // 24.74 ms per call, no change, profiles/r04_exp_small_args.txt.)
    PrimeSet ps = c->d_primes;
    if (mode & MODE_WIDE)
        hipLaunchKernelGGL((k_fwd_phase<LOGM, LE, true>), dim3(cpad * c->npr * 4), dim3(TH), lds, st, L.dig,
                           keyk, L.zpart, ps, mode);
    else
        hipLaunchKernelGGL((k_fwd_phase<LOGM, LE>), dim3(cpad * c->npr * 4), dim3(TH), lds, st, L.dig,
                           keyk, L.zpart, ps, mode);
    hipLaunchKernelGGL((k_inv_column<LOGM, LE>), dim3(cpad * c->npr * 2), dim3(TH), lds, st, L.zpart,
                       L.yres, L.ua, ps, k, c->n);
    HIPCHK(c, hipGetLastError());
    return SGFHE_OK;
}
int32_t launch_small(sgfhe_ctx *c, const sgfhe_ctx::Lane &L, const int32_t *keyk, uint32_t cpad,
                     uint32_t k, uint32_t mode, hipStream_t st) {
    switch (c->logm) {
#define X(LM) case LM: return launch_small_t<LM>(c, L, keyk, cpad, k, mode, st);
        SGFHE_FOR_LOGM(X)
#undef X
    }
    return fail(c, SGFHE_ERR_UNSUPPORTED, "unsupported log2(m)");
}

// The quarter form of the two transform kernels (kernels.h k_fwd_quarter / k_inv_quarter), m >= 4096.
template <int LOGM>
int32_t launch_quarter_t(sgfhe_ctx *c, const sgfhe_ctx::Lane &L, const int32_t *keyk, uint32_t cnt,
                         uint32_t k, uint32_t mode, hipStream_t st) {
    if constexpr (LOGM >= 12) {
        constexpr int LE = 3;
        constexpr int TH = NttGeom<LOGM - 2, LE>::T;
        // both transform kernels in one launch (kernels.h k_ext_quarter); at m = 16384 its workgroup of 1024
        // threads leaves 128 registers per thread and spills: that ring keeps the two launches
        if constexpr (LOGM <= 13) if (fused_takes(c, cnt, mode)) {
            const size_t ldsf = (size_t)6 * (sizeof(uint32_t) << (LOGM - 2));
            if (!(c->attr_done & ATTR_FUSED)) {
                HIPCHK(c, hipFuncSetAttribute((const void *)k_ext_quarter<LOGM, LE>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf));
                c->attr_done |= ATTR_FUSED;
            }
            hipLaunchKernelGGL((k_ext_quarter<LOGM, LE>), dim3(cnt * c->npr * 4), dim3(2 * TH), ldsf, st, L.dig, keyk,
                               reinterpret_cast<int32_t *>(L.yres), L.ua, c->d_primes, mode, k, c->n);
            HIPCHK(c, hipGetLastError());
            return SGFHE_OK;
        }
        const size_t lds = lds_bytes(LOGM - 2, 1);
        hipLaunchKernelGGL((k_fwd_quarter<LOGM, LE>), dim3(cnt * c->npr * 16), dim3(TH), lds, st, L.dig, keyk,
                           L.zpart, c->d_primes, mode);
        hipLaunchKernelGGL((k_inv_quarter<LOGM, LE>), dim3(cnt * c->npr * 8), dim3(TH), lds, st, L.zpart,
                           reinterpret_cast<int32_t *>(L.yres), L.ua, c->d_primes, k, c->n);
        HIPCHK(c, hipGetLastError());
        return SGFHE_OK;
    } else {
        return fail(c, SGFHE_ERR_UNSUPPORTED, "quarter form needs m >= 4096");
    }
}
int32_t launch_quarter(sgfhe_ctx *c, const sgfhe_ctx::Lane &L, const int32_t *keyk, uint32_t cnt, uint32_t k,
                       uint32_t mode, hipStream_t st) {
    switch (c->logm) {
#define X(LM) case LM: return launch_quarter_t<LM>(c, L, keyk, cnt, k, mode, st);
        SGFHE_FOR_LOGM(X)
#undef X
    }
    return fail(c, SGFHE_ERR_UNSUPPORTED, "unsupported log2(m)");
}

template <int LOGM>
int32_t launch_shortprod_t(sgfhe_ctx *c, const uint64_t *pdig, uint32_t *yg, uint32_t count,
                           uint32_t G, uint32_t groups, uint32_t mode, hipStream_t st) {
    const size_t lds = lds_bytes(LOGM, 2);
    if (!(c->attr_done & ATTR_SHORTPROD)) {
        HIPCHK(c, hipFuncSetAttribute((const void *)k_shortprod<LOGM>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->attr_done |= ATTR_SHORTPROD;
    }
    hipLaunchKernelGGL(k_shortprod<LOGM>, dim3(count * groups * c->npr), dim3(threads_of<LOGM>()), lds,
                       st, pdig, c->d_key, yg, c->d_primes, c->d_crt, c->n, G, groups, mode);
    HIPCHK(c, hipGetLastError());
    return SGFHE_OK;
}
int32_t launch_shortprod(sgfhe_ctx *c, const uint64_t *pdig, uint32_t *yg, uint32_t count,
                         uint32_t G, uint32_t groups, uint32_t mode, hipStream_t st) {
    switch (c->logm) {
#define X(LM) case LM: return launch_shortprod_t<LM>(c, pdig, yg, count, G, groups, mode, st);
        SGFHE_FOR_LOGM(X)
#undef X
    }
    return fail(c, SGFHE_ERR_UNSUPPORTED, "unsupported log2(m)");
}

template <int LOGM>
int32_t launch_keygen_ntt_t(sgfhe_ctx *c, const uint64_t *d_sk, int32_t *d_shat,
                            const ulonglong2 *d_acan, uint32_t *d_y, uint32_t R, bool shat_pass,
                            hipStream_t st) {
    if (shat_pass)
        hipLaunchKernelGGL(k_shat<LOGM>, dim3(c->npr), dim3(threads_of<LOGM>()), lds_bytes(LOGM, 1), st,
                           d_sk, d_shat, c->d_primes, c->n);
    else
        hipLaunchKernelGGL(k_polymul_s<LOGM>, dim3(R * c->npr), dim3(threads_of<LOGM>()),
                           lds_bytes(LOGM, 1), st, d_acan, d_shat, d_y, c->d_primes, c->d_crt);
    HIPCHK(c, hipGetLastError());
    return SGFHE_OK;
}
int32_t launch_keygen_ntt(sgfhe_ctx *c, const uint64_t *d_sk, int32_t *d_shat,
                          const ulonglong2 *d_acan, uint32_t *d_y, uint32_t R, bool shat_pass,
                          hipStream_t st) {
    switch (c->logm) {
#define X(LM) case LM: return launch_keygen_ntt_t<LM>(c, d_sk, d_shat, d_acan, d_y, R, shat_pass, st);
        SGFHE_FOR_LOGM(X)
#undef X
    }
    return fail(c, SGFHE_ERR_UNSUPPORTED, "unsupported log2(m)");
}

template <int LOGM>
int32_t launch_keytr_t(sgfhe_ctx *c, const ulonglong2 *canon, int32_t *keyhat, uint32_t poly0,
                       uint32_t npolys, hipStream_t st) {
    hipLaunchKernelGGL(k_key_transform<LOGM>, dim3(npolys * c->npr), dim3(threads_of<LOGM>()),
                       lds_bytes(LOGM, 1), st, canon, keyhat, c->d_primes, c->d_crt, poly0,
                       c->d_bad);
    HIPCHK(c, hipGetLastError());
    return SGFHE_OK;
}
int32_t launch_keytr(sgfhe_ctx *c, const ulonglong2 *canon, int32_t *keyhat, uint32_t poly0,
                     uint32_t npolys, hipStream_t st) {
    switch (c->logm) {
#define X(L) case L: return launch_keytr_t<L>(c, canon, keyhat, poly0, npolys, st);
        SGFHE_FOR_LOGM(X)
#undef X
    }
    return fail(c, SGFHE_ERR_UNSUPPORTED, "unsupported log2(m)");
}

template <int LOGM>
int32_t launch_dbgntt_t(sgfhe_ctx *c, const uint32_t *in, uint32_t *out, uint32_t pi, int inverse,
                        hipStream_t st) {
    hipLaunchKernelGGL(k_debug_ntt<LOGM>, dim3(1), dim3(threads_of<LOGM>()), lds_bytes(LOGM, 1), st,
                       in, out, c->d_primes, pi, (uint32_t)inverse);
    HIPCHK(c, hipGetLastError());
    return SGFHE_OK;
}
int32_t launch_dbgntt(sgfhe_ctx *c, const uint32_t *in, uint32_t *out, uint32_t pi, int inverse,
                      hipStream_t st) {
    switch (c->logm) {
#define X(L) case L: return launch_dbgntt_t<L>(c, in, out, pi, inverse, st);
        SGFHE_FOR_LOGM(X)
#undef X
    }
    return fail(c, SGFHE_ERR_UNSUPPORTED, "unsupported log2(m)");
}

// k_crt_acc is compiled once per prime count (its residue loops are unrolled)
#define SGFHE_FOR_NPR(X) X(2) X(3) X(4) X(5) X(6) X(7)
template <int NP>
void launch_crt_lean_t(sgfhe_ctx *c, const uint32_t *yres, uint64_t *dig, uint32_t total, hipStream_t st) {
    // a handful of gates (the latency form): one coefficient per thread, four times the threads
    if (total <= c->crt1_max) {
        const dim3 grid1((total + 255) / 256), block1(256);
        switch (c->h_lean.nl) {
        case 2: hipLaunchKernelGGL((k_crt_lean1<NP, 2>), grid1, block1, 0, st, yres, dig, c->d_lean, total, (uint32_t)c->logm); break;
        case 3: hipLaunchKernelGGL((k_crt_lean1<NP, 3>), grid1, block1, 0, st, yres, dig, c->d_lean, total, (uint32_t)c->logm); break;
        default: hipLaunchKernelGGL((k_crt_lean1<NP, 4>), grid1, block1, 0, st, yres, dig, c->d_lean, total, (uint32_t)c->logm); break;
        }
        return;
    }
    const dim3 grid((total / 4 + 255) / 256), block(256);
    switch (c->h_lean.nl) {
    case 2: hipLaunchKernelGGL((k_crt_lean<NP, 2>), grid, block, 0, st, yres, dig, c->d_lean, total / 4, (uint32_t)c->logm); break;
    case 3: hipLaunchKernelGGL((k_crt_lean<NP, 3>), grid, block, 0, st, yres, dig, c->d_lean, total / 4, (uint32_t)c->logm); break;
    default: hipLaunchKernelGGL((k_crt_lean<NP, 4>), grid, block, 0, st, yres, dig, c->d_lean, total / 4, (uint32_t)c->logm); break;
    }
}
template <int NP, bool WIDE>
void launch_crt_lean_rnd_t(sgfhe_ctx *c, const uint32_t *yres, uint64_t *dig, uint32_t total, hipStream_t st,
                           RndArgs ra, uint32_t iter) {
    // ROWS: a gathered call, every row on the draw stream of the ctx it came in on (kernels.h RndRow)
#define SGFHE_RND1(NL, ROWS) hipLaunchKernelGGL((k_crt_lean_rnd1<NP, NL, false, ROWS>), grid1, block1, 0, st, yres, dig, c->d_lean, total, (uint32_t)c->logm, ra, iter, (const PrimeK *)nullptr)
#define SGFHE_RND4(NL, ROWS) hipLaunchKernelGGL((k_crt_lean_rnd<NP, NL, WIDE, ROWS>), grid, block, 0, st, yres, dig, c->d_lean, total / 4, (uint32_t)c->logm, ra, iter)
    if (!WIDE && total <= c->crt1_max) {   // the latency form: one coefficient per thread
        const dim3 grid1((total + 255) / 256), block1(256);
        switch (c->h_lean.nl) {
        case 2: if (ra.rows) SGFHE_RND1(2, true); else SGFHE_RND1(2, false); break;
        case 3: if (ra.rows) SGFHE_RND1(3, true); else SGFHE_RND1(3, false); break;
        default: if (ra.rows) SGFHE_RND1(4, true); else SGFHE_RND1(4, false); break;
        }
        return;
    }
    const dim3 grid((total / 4 + 255) / 256), block(256);
    switch (c->h_lean.nl) {
    case 2: if (ra.rows) SGFHE_RND4(2, true); else SGFHE_RND4(2, false); break;
    case 3: if (ra.rows) SGFHE_RND4(3, true); else SGFHE_RND4(3, false); break;
    default: if (ra.rows) SGFHE_RND4(4, true); else SGFHE_RND4(4, false); break;
    }
#undef SGFHE_RND1
#undef SGFHE_RND4
}
int32_t launch_crt_raw(sgfhe_ctx *c, const uint32_t *yres, uint64_t *dig, uint32_t total,
                       uint32_t mode, hipStream_t st, RndArgs ra, uint32_t iter) {
    // the k-loop's own case (deterministic flatten, accumulator present): the integer-only kernel,
    // four coefficients per thread; every other mode, and parameter sets outside its bounds
    // (B < 2^12, Q < 2^30), the general one
    const bool lean = mode == 0u && c->h_lean.nl != 0 && c->use_lean;
    // the same for the randomised flatten (the k-loop's modes MODE_RANDOM and MODE_RANDOM | MODE_WIDE)
    const bool lean_rnd = (mode & ~MODE_WIDE) == MODE_RANDOM && c->lean_rnd_ok && c->use_lean;
    switch (c->npr) {
#define X(NP)                                                                                     \
    case NP:                                                                                      \
        if (lean)                                                                                 \
            launch_crt_lean_t<NP>(c, yres, dig, total, st);                                       \
        else if (lean_rnd && (mode & MODE_WIDE))                                                  \
            launch_crt_lean_rnd_t<NP, true>(c, yres, dig, total, st, ra, iter);                   \
        else if (lean_rnd)                                                                        \
            launch_crt_lean_rnd_t<NP, false>(c, yres, dig, total, st, ra, iter);                  \
        else if (mode == 0u) /* two coefficients per thread */                                   \
            hipLaunchKernelGGL(k_crt_acc2<NP>, dim3((total / 2 + 255) / 256), dim3(256), 0, st,   \
                               yres, dig, c->d_crt, total / 2, (uint32_t)c->logm);                \
        else if (ra.rows)                                                                         \
            hipLaunchKernelGGL((k_crt_acc<NP, true>), dim3((total + 255) / 256), dim3(256), 0, st, yres, \
                               dig, c->d_crt, total, (uint32_t)c->logm, mode, ra, iter);          \
        else                                                                                      \
            hipLaunchKernelGGL((k_crt_acc<NP, false>), dim3((total + 255) / 256), dim3(256), 0, st, yres, \
                               dig, c->d_crt, total, (uint32_t)c->logm, mode, ra, iter);          \
        break;
        SGFHE_FOR_NPR(X)
#undef X
    default: return fail(c, SGFHE_ERR_UNSUPPORTED, "unsupported number of RNS primes");
    }
    HIPCHK(c, hipGetLastError());
    return SGFHE_OK;
}
int32_t launch_crt(sgfhe_ctx *c, const sgfhe_ctx::Lane &L, uint32_t cpad, uint32_t mode,
                   hipStream_t st, RndArgs ra = RndArgs{}, uint32_t iter = 0) {
    return launch_crt_raw(c, L.yres, L.dig, cpad * 2 * c->M, mode, st, ra, iter);
}

// CRT kernel of the quarter form: partial residues in, the last two inverse stages inside
int32_t launch_crt_quarter(sgfhe_ctx *c, const sgfhe_ctx::Lane &L, uint32_t cnt, uint32_t mode, hipStream_t st,
                           RndArgs ra, uint32_t iter) {
    const uint32_t total = cnt * 2 * c->M;
    const dim3 grid((total + 255) / 256), block(256);
    const int32_t *yp = reinterpret_cast<const int32_t *>(L.yres);
    if (mode & MODE_RANDOM) {   // the randomised flatten: one coefficient per thread here too (k_crt_lean_rnd1)
#define SGFHE_RNDQ(NPQ, NL, ROWS) hipLaunchKernelGGL((k_crt_lean_rnd1<NPQ, NL, true, ROWS>), grid, block, 0, st, L.yres, L.dig, c->d_lean, total, (uint32_t)c->logm, ra, iter, c->d_primes)
        switch (c->npr) {
#define X(NP)                                                                                                     \
        case NP:                                                                                                  \
            switch (c->h_lean.nl) {                                                                               \
            case 2: if (ra.rows) SGFHE_RNDQ(NP, 2, true); else SGFHE_RNDQ(NP, 2, false); break;                          \
            case 3: if (ra.rows) SGFHE_RNDQ(NP, 3, true); else SGFHE_RNDQ(NP, 3, false); break;                          \
            default: if (ra.rows) SGFHE_RNDQ(NP, 4, true); else SGFHE_RNDQ(NP, 4, false); break;                         \
            }                                                                                                     \
            break;
            SGFHE_FOR_NPR(X)
#undef X
        default: return fail(c, SGFHE_ERR_UNSUPPORTED, "unsupported number of RNS primes");
        }
#undef SGFHE_RNDQ
        HIPCHK(c, hipGetLastError());
        return SGFHE_OK;
    }
    switch (c->npr) {
#define X(NP)                                                                                                     \
    case NP:                                                                                                      \
        switch (c->h_lean.nl) {                                                                                   \
        case 2: hipLaunchKernelGGL((k_crt_lean1q<NP, 2>), grid, block, 0, st, yp, L.dig, c->d_primes, c->d_lean, total, (uint32_t)c->logm); break; \
        case 3: hipLaunchKernelGGL((k_crt_lean1q<NP, 3>), grid, block, 0, st, yp, L.dig, c->d_primes, c->d_lean, total, (uint32_t)c->logm); break; \
        default: hipLaunchKernelGGL((k_crt_lean1q<NP, 4>), grid, block, 0, st, yp, L.dig, c->d_primes, c->d_lean, total, (uint32_t)c->logm); break; \
        }                                                                                                         \
        break;
        SGFHE_FOR_NPR(X)
#undef X
    default: return fail(c, SGFHE_ERR_UNSUPPORTED, "unsupported number of RNS primes");
    }
    HIPCHK(c, hipGetLastError());
    return SGFHE_OK;
}

// ---- buffers ------------------------------------------------------------------------------------

uint32_t round_up8(uint32_t x) { return (x + 7u) & ~7u; }

size_t per_bootstrap_bytes(const sgfhe_ctx *c) {
    return (size_t)2 * c->M * sizeof(ulonglong2) + (size_t)2 * c->npr * c->M * 4 + (size_t)c->n * 4;
}

// Default chunk: the per-iteration working set (digits + residues) of a chunk stays near the size
// of the 256 MiB Infinity Cache while a launch is long enough to amortise its ramp-up and drain
// (about 9 us per k_extprod launch).  Measured at Params(1024) (tools/chunk_sweep.py, us of
// k_extprod per bootstrap and iteration): chunk 256 0.568, 408 0.553, 512 0.547, 608 0.572,
// 816 0.575.
uint32_t default_chunk(const sgfhe_ctx *c) {
    // Two lanes (the default): half the budget per lane, so that the working sets of both chunks
    // stay near the Infinity Cache together.  Measured at Params(1024) with k_crt_lean
    // (profiles/r03_exp_lanes_sweep.txt): two lanes of 128 / 192 / 256 / 320 / 384 give
    // 1892 / 1916 / 1907 / 1825 / 1814 bootstraps/s against 1864 for one lane of 512.
    size_t budget = (size_t)(c->lanes == 2 ? 160 : 320) << 20;
    size_t k = budget / per_bootstrap_bytes(c);
    if (k >= 256) k = (k / 256) * 256;
    else k = (k / 8) * 8;
    if (k < 8) k = 8;
    if (k > 4096) k = 4096;
    return (uint32_t)k;
}

void free_lanes(sgfhe_ctx *c) {
    for (auto &L : c->lane) {
        if (L.dig) (void)hipFree(L.dig);
        if (L.yres) (void)hipFree(L.yres);
        if (L.ua) (void)hipFree(L.ua);
        if (L.zpart) (void)hipFree(L.zpart);
        L = sgfhe_ctx::Lane();
    }
    c->cap = 0;
}

int32_t ensure_work(sgfhe_ctx *c, uint32_t cpad) {
    if (cpad <= c->cap) return SGFHE_OK;
    int32_t rcd = drain(c);   // an earlier asynchronous call may still be using the buffers
    if (rcd) return rcd;
    free_lanes(c);
    for (auto &L : c->lane) {
        HIPCHK(c, hipMalloc(&L.dig, (size_t)cpad * 4 * c->M * sizeof(uint64_t)));
        HIPCHK(c, hipMalloc(&L.yres, (size_t)cpad * 2 * c->npr_max * c->M * 4));
        HIPCHK(c, hipMalloc(&L.ua, (size_t)cpad * c->n * 4));
        const uint32_t cs = cpad < c->small_max ? cpad : c->small_max;
        if (cs) HIPCHK(c, hipMalloc(&L.zpart, (size_t)cs * c->npr_max * 8 * c->M * 4));
    }
    c->cap = cpad;
    return SGFHE_OK;
}

// ---- timing events ------------------------------------------------------------------------------

void timing_flush(sgfhe_ctx *c) {
    // (the samples may sit on a caller's stream: wait for the events, not for the ctx stream)
    for (auto &t : c->ev) {
        float ms = 0;
        (void)hipEventSynchronize(t.e2);
        if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) { c->t_ext += ms; c->n_ext++; }
        if (hipEventElapsedTime(&ms, t.e1, t.e2) == hipSuccess) { c->t_crt += ms; c->n_crt++; }
        (void)hipEventDestroy(t.e0);
        (void)hipEventDestroy(t.e1);
        (void)hipEventDestroy(t.e2);
    }
    c->ev.clear();
    for (auto &t : c->ev_call) {
        float ms = 0;
        (void)hipEventSynchronize(t.e1);
        if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) { c->t_call += ms; c->n_call++; c->boots_call += t.batch; }
        (void)hipEventDestroy(t.e0);
        (void)hipEventDestroy(t.e1);
    }
    c->ev_call.clear();
}

// ---- the k-loop (fhe.jl:579-582) over one chunk, or over two chunks in a pipeline --------------------

// One chunk of the batch on its lane (work buffers + stream).
struct ChunkJob {
    const sgfhe_ctx::Lane *L;
    hipStream_t st;
    uint32_t cb, cpad;  // bootstraps in the chunk, padded to a multiple of 8
    size_t c0;          // index of its first bootstrap in the call
    RndArgs ra;
    bool sampled;       // HIP-event timing samples are taken on this chunk
};

// Iteration k of every job: k_extprod (or its small-batch form) then k_crt_acc, each job on its
// own stream.  (Measured and rejected in round 2: chaining the two streams so that one chunk's
// CRT kernel runs beside the other chunk's external product -- also with the CRT as a
// one-workgroup-per-CU grid-stride "rider" kernel -- costs the external product as much time
// as the CRT takes alone)
int32_t run_iterations(sgfhe_ctx *c, ChunkJob *jobs, int njobs, uint64_t n_iters, uint32_t mode) {
    const size_t slice = (size_t)c->npr * 8 * c->M;
    for (uint64_t k = 0; k < n_iters; k++) {
        for (int j = 0; j < njobs; j++) {
            const ChunkJob &J = jobs[j];
            // few bootstraps: 6 workgroups per (bootstrap, prime) instead of 1 (k_fwd_phase / k_inv_column)
            const bool small = J.cpad <= c->small_max && J.L->zpart != nullptr;
            const bool sample = c->timing && J.sampled && (k % 64 == 1) && c->ev.size() < 2048;
            hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
            if (sample) {
                HIPCHK(c, hipEventCreate(&e0));
                HIPCHK(c, hipEventCreate(&e1));
                HIPCHK(c, hipEventCreate(&e2));
                HIPCHK(c, hipEventRecord(e0, J.st));
            }
            // (Round 4 also tried warming the next iteration's key slice from a second stream, one iteration
            // ahead, with k_fwd_phase's own workgroup -> XCD mapping: 18.06 / 19.01 / 20.85 ms against 17.94 /
            // 18.80 / 20.24 for 1 / 2 / 4 gates, no gain, profiles/r04_exp_prefetch.txt.)
            // (the small-batch kernels index bootstraps directly: no padding to a multiple of 8, which is
            // k_extprod's XCD mapping's; a one-gate call then runs one gate's workgroups, not eight's)
            const uint32_t cnt = small && !c->small_padded ? J.cb : J.cpad;
            // a few gates, deterministic flatten, m >= 4096: each transform cut across four workgroups
            // (both flatten modes; not the three-plane digit records of B >= 2^46, MODE_WIDE)
            const bool quarter = small && quarter_ok(c, mode) && (cnt <= c->split_max || fused_takes(c, cnt, mode));
#ifdef SGFHE_WITH_ITER_ALL
            if (!small && c->iter_all && mode == 0u && c->logm == 13 && c->npr == 5 && c->use_lean && c->h_lean.nl == 3) {
                const int32_t rci = launch_iter_all(c, *J.L, c->d_key + k * slice, J.cpad, (uint32_t)k, J.st);
                if (rci) return rci;
                continue;
            }
#endif
            int32_t rc = quarter ? launch_quarter(c, *J.L, c->d_key + k * slice, cnt, (uint32_t)k, mode, J.st)
                         : small ? launch_small(c, *J.L, c->d_key + k * slice, cnt, (uint32_t)k, mode, J.st)
                                 : launch_extprod(c, *J.L, c->d_key + k * slice, J.cpad, (uint32_t)k, mode, J.st);
            if (rc) return rc;
            if (sample) HIPCHK(c, hipEventRecord(e1, J.st));
            rc = quarter ? launch_crt_quarter(c, *J.L, cnt, mode, J.st, J.ra, (uint32_t)k + 1)
                         : launch_crt(c, *J.L, cnt, mode, J.st, J.ra, (uint32_t)k + 1);
            if (rc) return rc;
            if (sample) {
                HIPCHK(c, hipEventRecord(e2, J.st));
                c->ev.push_back({e0, e1, e2});
            }
        }
    }
    return SGFHE_OK;
}

// memcpy between the caller's pageable arrays and the page-locked mirrors.  The copies of the first
// chunk's inputs and of the last chunks' results are the part of a host-pointer call that no kernel
// hides: above 2 MB they are cut across up to four threads (one thread moves 5-8 GB/s).
// Called inside extern "C" entry points: nothing may throw out of it.  std::thread's constructor throws
// std::system_error when the process may not start another thread (EAGAIN: ulimit -u, a container's pid
// limit); the part that thread would have copied, and everything after it, is then copied here.
void host_copy(void *dst, const void *src, size_t bytes) {
    const size_t parts = bytes >= ((size_t)2 << 20) ? (bytes >= ((size_t)8 << 20) ? 4 : 2) : 1;
    if (parts == 1) { memcpy(dst, src, bytes); return; }
    const size_t step = ((bytes / parts) + 4095) & ~(size_t)4095;   // parts 1 .. of `step` bytes; the last takes the rest
    std::thread th[3];
    size_t nth = 0, off = step;
    for (; nth < parts - 1 && off < bytes; off += step) {
        const size_t len = (nth == parts - 2 || bytes - off < step) ? bytes - off : step;
        char *d = static_cast<char *>(dst) + off;
        const char *sr = static_cast<const char *>(src) + off;
        try {
            th[nth] = std::thread([d, sr, len] { memcpy(d, sr, len); });
            nth++;
        } catch (...) {            // no thread to be had: this part and the rest, on the caller's thread
            memcpy(d, sr, bytes - off);
            break;
        }
        if (len == bytes - off) break;
    }
    memcpy(dst, src, step < bytes ? step : bytes);
    for (size_t i = 0; i < nth; i++) th[i].join();
}

// Host buffers of sgfhe_bootstrap_batch, moved chunk by chunk beside the lanes' kernels (stream_io):
// a chunk's inputs go up while the chunks before it compute, its outputs come down while the chunks
// after it compute, and the CPU copies between the caller's (pageable) arrays and the page-locked
// mirrors happen while the device is busy.  What stays exposed is the first chunk's input (a few MB)
// and the last chunks' output.
struct HostPipe {
    const uint64_t *a1, *b1, *a2, *b2;   // the caller's arrays
    uint64_t *out;                        // the caller's result array
    uint64_t *d_in, *d_out;               // device staging: chunk at row c0 = words [c0 (2 n + 2) ...) laid out
                                          // [a1 rows | a2 rows | b1 | b2]; result rows as in `out`
    uint64_t *p_in, *p_out;               // page-locked mirrors of both, same layouts
    size_t out_row_words;                 // 3 (n + 1), twice that with SGFHE_FLAG_RAW_MODQ
};

int32_t bootstrap_device(sgfhe_ctx *c, const uint64_t *a1, const uint64_t *b1, const uint64_t *a2,
                         const uint64_t *b2, size_t batch, uint64_t *out, uint32_t flags,
                         uint64_t n_iters, ulonglong2 *acc_out, hipStream_t st,
                         uint64_t *dig_out = nullptr, const HostPipe *hp = nullptr) {
    if (!c->have_key) return fail(c, SGFHE_ERR_NO_KEY, "no bootstrap key uploaded");
    uint32_t chunk = c->chunk ? c->chunk : default_chunk(c);
    const uint32_t mode = c->rnd ? (MODE_RANDOM | ((c->B >> 46) ? MODE_WIDE : 0u)) : 0u;
    if (!c->chunk && c->lanes == 2 && batch > 2 * (size_t)c->small_max) {
        // Automatic chunk size with two lanes: cut the batch into an even number of equal chunks no
        // larger than the default, so that both lanes are busy from the first bootstrap to the last
        // (a batch of 256 runs as 128 + 128 instead of one chunk on one lane, 384 as 192 + 192
        // instead of 256 + 128).  Measured at Params(1024) (profiles/r03_exp_mid_batches.txt):
        // 64 / 128 / 256 / 384 / 640 bootstraps 1158 -> 1401, 1555 -> 1858, 1875 -> 1981, 1847 -> 2071,
        // 1850 -> 2030 per second.  Halves that would fall to the small-batch form (batch <= 48) are not
        // split: 48 bootstraps run at 1277 per second as one chunk and at 1024 as 24 + 24.
        const size_t pairs = (batch + 2 * (size_t)chunk - 1) / (2 * (size_t)chunk);
        chunk = round_up8((uint32_t)((batch + 2 * pairs - 1) / (2 * pairs)));
    } else if (!c->chunk && c->lanes == 2 && c->small_lanes && c->logm >= 12 &&
               batch >= (fused_cap(c, mode) ? fused_cap(c, mode) + 1 : 8u) &&
               batch <= c->small_lanes_max && (batch + 1) / 2 <= c->small_max) {
        // A call of 8 to 24 gates in the latency form: two halves on the two lanes.  Each half is a chain of
        // dependent launches on a mostly idle device, and two chains overlap; the halves need no rounding
        // to 8 (the latency kernels index their gates directly), so 8 gates run as 4 + 4 and 12 as 6 + 6
        // in the quarter form.  Same call (profiles/r04_exp_small_lanes.txt): 8 / 10 / 12 / 16 / 20 / 24 gates
        // 20.8 / 21.6 / 22.8 / 27.4 / 27.8 / 29.7 ms against 23.4 / 25.2 / 27.0 / 27.8 / 30.5 / 31.4 as one chunk;
        // below 8 gates two chains cost more than they overlap (2 / 4 / 6 gates 18.3 / 18.3 / 19.0 against
        // 15.8 / 17.8 / 19.8), so those stay on one stream.
        // With the fused quarter kernel (m = 4096, 8192) one chain takes up to 12 gates at the cost of two
        // chains of half the size or less (8 / 10 / 12 gates 20.9 / 21.9 / 22.7 ms against 21.0 / 22.0 / 22.9;
        // Params(512) 7.7 / 8.0 / 8.3 against 9.1 / 9.0 / 9.2), and the halves of 13 to 24 gates take it
        // (profiles/r04_exp_fused.txt): two chains from 13 gates there.
        // Rings below m = 4096 stay on one chain: their launches are too short for two chains to overlap, and
        // the halves then cost their sum (Params(64) / (128) / (256), 8 to 24 gates: 1.4 / 2.75 / 5.4 ms as two
        // halves against 0.7 / 1.4 / 3.3 as one chunk, profiles/r04_exp_small_rings_lanes.txt).
        chunk = (uint32_t)((batch + 1) / 2);
    }
    const uint32_t n = c->n, M = c->M;
    const bool raw = flags & SGFHE_FLAG_RAW_MODQ;
    if (flags & SGFHE_FLAG_RAW_RNS2) {
        if (!raw) return fail(c, SGFHE_ERR_INVALID_ARG, "SGFHE_FLAG_RAW_RNS2 needs SGFHE_FLAG_RAW_MODQ");
        if (!c->have_rns2)
            return fail(c, SGFHE_ERR_INVALID_ARG, "SGFHE_FLAG_RAW_RNS2: no RNS2 moduli (upload the key with sgfhe_bkey_upload_rns2)");
    }
    const bool two_lanes = c->lanes == 2 && batch > chunk;
    // (the call number of the draw stream: this ctx's counter -- unless the call was numbered when it was handed
    //  to the coalescer, or is a gathered call, whose rows carry their own numbers: kernels.h RndRow)
    uint32_t call = 0u;
    if (c->rnd) {
        if (c->gather_rows) call = 0u;
        else if (c->call_fixed >= 0) { call = (uint32_t)c->call_fixed; c->call_fixed = -1; }
        else call = c->rnd_call++;
    }
    c->last_call = call;
    {   // work buffers for the largest chunk of this call, before anything of it is queued
        const uint32_t first = (uint32_t)(batch < chunk ? batch : chunk);
        int32_t rc = ensure_work(c, round_up8(first));
        if (rc) return rc;
        c->last_chunk = round_up8(first);
    }
    {   // after everything earlier calls queued on this ctx, whatever their streams
        int32_t rc = fence_begin(c, st);
        if (rc) return rc;
    }
    // From here on work is queued: should a later step fail, whatever has been queued still ends in
    // ev_done, so that the next call (or the destructor) waits for it before touching the buffers.
    struct QueuedWork {
        sgfhe_ctx *c;
        hipStream_t st;
        bool closed = false;
        ~QueuedWork() {
            if (closed) return;
            if (c->lanes == 2) {   // the second lane may hold part of it
                if (hipEventRecord(c->ev_join, c->stream2) == hipSuccess) (void)hipStreamWaitEvent(st, c->ev_join, 0);
            }
            if (hipEventRecord(c->ev_done, st) == hipSuccess) c->pending = true;
        }
    } queued{c, st};
    hipEvent_t ecall0 = nullptr, ecall1 = nullptr;
    if (c->timing && c->ev_call.size() < 256) {
        HIPCHK(c, hipEventCreate(&ecall0));
        HIPCHK(c, hipEventCreate(&ecall1));
        HIPCHK(c, hipEventRecord(ecall0, st));
    }
    // A host-pointer call that is one group of chunks (a handful of gates up to two chunks' worth) has
    // nothing to overlap its copies with: they go on `st` itself, in order with the kernels -- one upload
    // before the first kernel, one download behind the last -- without the copy streams' events.
    const size_t stride = (size_t)chunk * (two_lanes ? 2 : 1);
    const bool hp_single = hp && batch <= stride;
    if (hp_single) {
        const size_t nn = n;
        host_copy(hp->p_in, hp->a1, batch * nn * 8);
        host_copy(hp->p_in + batch * nn, hp->a2, batch * nn * 8);
        memcpy(hp->p_in + 2 * batch * nn, hp->b1, batch * 8);
        memcpy(hp->p_in + 2 * batch * nn + batch, hp->b2, batch * 8);
        HIPCHK(c, hipMemcpyAsync(hp->d_in, hp->p_in, batch * (2 * nn + 2) * 8, hipMemcpyHostToDevice, st));
    }
    if (two_lanes) {  // fork: the second lane starts after everything already queued on st
        HIPCHK(c, hipEventRecord(c->ev_fork, st));
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    }
    // events of the host pipeline, from the ctx's pool
    size_t ev_next = 0;
    auto next_event = [&](hipEvent_t *e) -> hipError_t {
        if (ev_next == c->ev_pool.size()) {
            hipEvent_t ne = nullptr;
            const hipError_t er = hipEventCreateWithFlags(&ne, hipEventDisableTiming);
            if (er != hipSuccess) return er;
            c->ev_pool.push_back(ne);
        }
        *e = c->ev_pool[ev_next++];
        return hipSuccess;
    };
    struct OutJob { hipEvent_t done; size_t c0; uint32_t cb; };
    std::vector<OutJob> outq;   // result rows on their way to the page-locked mirror
    size_t out_drained = 0;
    auto drain_out = [&](size_t upto) -> hipError_t {   // ... and from there into the caller's array
        for (; out_drained < upto; out_drained++) {
            const OutJob &o = outq[out_drained];
            const hipError_t er = hipEventSynchronize(o.done);
            if (er != hipSuccess) return er;
            host_copy(hp->out + o.c0 * hp->out_row_words, hp->p_out + o.c0 * hp->out_row_words,
                      (size_t)o.cb * hp->out_row_words * 8);
        }
        return hipSuccess;
    };
    // SGFHE_DEBUG_IO=1: wall-clock phases of a pipelined host-pointer call
    const bool dbg_io = hp && getenv("SGFHE_DEBUG_IO") != nullptr;
    // SGFHE_IO_EXP: experiments of round 4 (tools/io_variants.py, profiles/r04_exp_io_variants.txt):
    // 1 = results collected only at the end, 2 = all results after the last kernel, 4 = all inputs
    // before the first kernel.  The default (0) measured best: 1.003 x the device-resident call.
    const int io_exp = hp && getenv("SGFHE_IO_EXP") ? atoi(getenv("SGFHE_IO_EXP")) : 0;
    auto wall = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tw0 = dbg_io ? wall() : 0.0;
    double tw_first = 0.0, tw_wait = 0.0;
    hipEvent_t e_allin = nullptr;
    if ((io_exp & 4) && !hp_single) {   // experiment: every chunk's inputs up front, one event
        for (size_t c0 = 0; c0 < batch; c0 += chunk) {
            const size_t cb = batch - c0 < chunk ? batch - c0 : chunk, w0 = c0 * (2 * (size_t)n + 2);
            uint64_t *pi = hp->p_in + w0;
            host_copy(pi, hp->a1 + c0 * n, cb * n * 8);
            host_copy(pi + cb * n, hp->a2 + c0 * n, cb * n * 8);
            memcpy(pi + 2 * cb * n, hp->b1 + c0, cb * 8);
            memcpy(pi + 2 * cb * n + cb, hp->b2 + c0, cb * 8);
        }
        HIPCHK(c, hipMemcpyAsync(hp->d_in, hp->p_in, batch * (2 * (size_t)n + 2) * 8, hipMemcpyHostToDevice, c->stream_io));
        HIPCHK(c, next_event(&e_allin));
        HIPCHK(c, hipEventRecord(e_allin, c->stream_io));
    }
    for (size_t g0 = 0; g0 < batch; g0 += stride) {
        ChunkJob jobs[2];
        int njobs = 0;
        const size_t out_before = outq.size();
        for (int li = 0; li < (two_lanes ? 2 : 1); li++) {
            const size_t c0 = g0 + (size_t)li * chunk;
            if (c0 >= batch) break;
            ChunkJob &J = jobs[njobs++];
            J.L = &c->lane[li];
            J.st = li ? c->stream2 : st;
            J.cb = (uint32_t)((batch - c0 < chunk) ? batch - c0 : chunk);
            J.cpad = round_up8(J.cb);
            J.c0 = c0;
            J.ra = RndArgs{c->rnd_key, call, (uint32_t)c0, c->gather_rows};
            J.sampled = li == 0 && J.cpad == c->last_chunk;
            const uint64_t *ja1 = a1 + c0 * n, *jb1 = b1 + c0, *ja2 = a2 + c0 * n, *jb2 = b2 + c0;
            if (hp_single) {   // whole-call layout [a1 | a2 | b1 | b2], uploaded above
                ja1 = hp->d_in + c0 * n; ja2 = hp->d_in + (batch + c0) * n;
                jb1 = hp->d_in + 2 * batch * n + c0; jb2 = jb1 + batch;
            } else if (hp && e_allin) {
                const size_t w0 = c0 * (2 * (size_t)n + 2), cb = J.cb;
                uint64_t *di = hp->d_in + w0;
                if (g0 == 0) HIPCHK(c, hipStreamWaitEvent(J.st, e_allin, 0));
                ja1 = di; ja2 = di + cb * n; jb1 = di + 2 * cb * n; jb2 = jb1 + cb;
            } else if (hp) {   // this chunk's inputs: caller's arrays -> page-locked mirror -> device, on stream_io
                const size_t w0 = c0 * (2 * (size_t)n + 2), cb = J.cb;
                uint64_t *pi = hp->p_in + w0, *di = hp->d_in + w0;
                host_copy(pi, hp->a1 + c0 * n, cb * n * 8);
                host_copy(pi + cb * n, hp->a2 + c0 * n, cb * n * 8);
                memcpy(pi + 2 * cb * n, hp->b1 + c0, cb * 8);
                memcpy(pi + 2 * cb * n + cb, hp->b2 + c0, cb * 8);
                HIPCHK(c, hipMemcpyAsync(di, pi, cb * (2 * (size_t)n + 2) * 8, hipMemcpyHostToDevice, c->stream_io));
                hipEvent_t ein;
                HIPCHK(c, next_event(&ein));
                HIPCHK(c, hipEventRecord(ein, c->stream_io));
                HIPCHK(c, hipStreamWaitEvent(J.st, ein, 0));
                ja1 = di; ja2 = di + cb * n; jb1 = di + 2 * cb * n; jb2 = jb1 + cb;
            }
            const uint32_t tot = J.cpad * M;
            if (J.ra.rows)
                hipLaunchKernelGGL(k_init<true>, dim3((tot + 255) / 256), dim3(256), 0, J.st, ja1, jb1, ja2, jb2,
                                   J.L->dig, J.L->ua, c->d_crt, J.cb, J.cpad, n, (uint32_t)c->logm, mode, J.ra);
            else
                hipLaunchKernelGGL(k_init<false>, dim3((tot + 255) / 256), dim3(256), 0, J.st, ja1, jb1, ja2, jb2,
                                   J.L->dig, J.L->ua, c->d_crt, J.cb, J.cpad, n, (uint32_t)c->logm, mode, J.ra);
            HIPCHK(c, hipGetLastError());
        }
        if (dbg_io && g0 == 0) tw_first = wall();
        int32_t rc = run_iterations(c, jobs, njobs, n_iters, mode);
        if (rc) return rc;
        for (int j = 0; j < njobs; j++) {
            const ChunkJob &J = jobs[j];
            if (acc_out) {
                const uint32_t t2 = J.cb * 2 * M;
                hipLaunchKernelGGL(k_dump_acc, dim3((t2 + 255) / 256), dim3(256), 0, J.st, J.L->dig,
                                   acc_out + J.c0 * 2 * M, c->d_crt, t2, (uint32_t)c->logm, mode);
                HIPCHK(c, hipGetLastError());
            }
            if (dig_out) {
                const uint32_t t2 = J.cb * 2 * M;
                hipLaunchKernelGGL(k_dump_digits, dim3((t2 + 255) / 256), dim3(256), 0, J.st, J.L->dig,
                                   dig_out + J.c0 * 4 * M, t2, (uint32_t)c->logm, mode);
                HIPCHK(c, hipGetLastError());
            }
            if (out) {
                const uint32_t t3 = J.cb * (n + 1);
                hipLaunchKernelGGL(k_final, dim3((t3 + 255) / 256), dim3(256), 0, J.st, J.L->dig,
                                   out + J.c0 * 3 * (n + 1) * (raw ? 2 : 1), c->d_crt, J.cb, n,
                                   (uint32_t)c->logm, raw ? 1u : 0u, mode);
                if (raw && (flags & SGFHE_FLAG_RAW_RNS2))  // residues leave as (v1, v2) pairs (rns.jl:16-18)
                    hipLaunchKernelGGL(k_canon_to_rns2, dim3((3 * t3 + 255) / 256), dim3(256), 0, J.st,
                                       reinterpret_cast<ulonglong2 *>(out) + J.c0 * 3 * (n + 1),
                                       (size_t)3 * t3, c->rns2);
                HIPCHK(c, hipGetLastError());
            }
            if (hp_single) {
                // one download behind the join, below
            } else if (hp && (io_exp & 2)) {
                outq.push_back({nullptr, J.c0, J.cb});
            } else if (hp) {   // this chunk's results: device -> page-locked mirror on stream_io2, behind its last kernel
                hipEvent_t ek, eo;
                HIPCHK(c, next_event(&ek));
                HIPCHK(c, next_event(&eo));
                HIPCHK(c, hipEventRecord(ek, J.st));
                HIPCHK(c, hipStreamWaitEvent(c->stream_io2, ek, 0));
                const size_t w0 = J.c0 * hp->out_row_words;
                HIPCHK(c, hipMemcpyAsync(hp->p_out + w0, hp->d_out + w0, (size_t)J.cb * hp->out_row_words * 8,
                                         hipMemcpyDeviceToHost, c->stream_io2));
                HIPCHK(c, hipEventRecord(eo, c->stream_io2));
                outq.push_back({eo, J.c0, J.cb});
            }
        }
        // with this group queued, collect the results of the group before it
        if (hp && !(io_exp & 3)) {
            const double t = dbg_io ? wall() : 0.0;
            HIPCHK(c, drain_out(out_before));
            if (dbg_io) tw_wait += wall() - t;
        }
    }
    if (two_lanes) {  // join
        HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));
    }
    if (ecall0) {
        HIPCHK(c, hipEventRecord(ecall1, st));
        c->ev_call.push_back({ecall0, ecall1, (uint64_t)batch});
    }
    {
        int32_t rc = fence_end(c, st);
        if (rc) return rc;
        queued.closed = true;
    }
    if (hp_single) {
        HIPCHK(c, hipMemcpyAsync(hp->p_out, hp->d_out, batch * hp->out_row_words * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        c->pending = false;
        host_copy(hp->out, hp->p_out, batch * hp->out_row_words * 8);
        return SGFHE_OK;
    }
    if (hp && (io_exp & 2)) {   // experiment: all results after the last kernel
        HIPCHK(c, hipMemcpyAsync(hp->p_out, hp->d_out, batch * hp->out_row_words * 8, hipMemcpyDeviceToHost, st));
        hipEvent_t eo;
        HIPCHK(c, next_event(&eo));
        HIPCHK(c, hipEventRecord(eo, st));
        for (auto &o : outq) o.done = eo;
    }
    if (hp) {
        const double t1 = dbg_io ? wall() : 0.0;
        double t2 = 0.0;
        if (dbg_io && !outq.empty()) {   // the last chunks' kernels are done when their download may start
            (void)hipEventSynchronize(c->ev_done);
            t2 = wall();
        }
        HIPCHK(c, drain_out(outq.size()));
        if (dbg_io)
            fprintf(stderr, "[sgfhe io] pipelined call of %zu: first chunks staged and queued after %.2f ms, all queued after "
                    "%.2f ms (of which %.2f waiting for earlier results), last kernel done at %.2f, results in the "
                    "caller's array at %.2f\n", batch, tw_first - tw0, t1 - tw0, tw_wait, t2 - tw0, wall() - tw0);
    }
    return SGFHE_OK;
}

// ---- constants -------------------------------------------------------------------------------------

// One basis: everything that depends on which primes are in use -- CRT / flatten constants, the
// per-prime records, the twiddle tables of primes not yet on the device.  Leaves the result in the
// ctx's active fields (the caller saves them into basis[b]).
int32_t build_basis(sgfhe_ctx *c, uint32_t npr, const uint32_t *cand_primes, double log_have, double log_mbq) {
    const uint32_t M = c->M;
    const int logm = c->logm;
    const u128 Q = c->Q, B = c->B;
    c->npr = npr;
    for (uint32_t i = 0; i < npr; i++) c->primes[i] = cand_primes[i];
    const int NPR = (int)npr;  // the loops below run over the primes in use

    // Packing (fhe.jl:683-687): G slices of two digit polynomials each are summed exactly before a
    // CRT: |sum| <= G m B Q / 2 (four times that with the randomised flatten) must stay below
    // 0.4 M_rns.
    {
        auto group = [&](double lg) {
            uint32_t G = 0;
            if (lg >= 0) { G = 1; while (G < c->n && (double)(31 - __builtin_clz(2 * G)) <= lg) G *= 2; }
            return G;
        };
        const double lg = log_have + log2(0.8) - log_mbq - 0.001;
        c->pack_G = group(lg);
        c->pack_G_rnd = group(lg - 2.0);
    }

    // flatten constants (utils.jl:162-169)
    const u128 s = (B & 1) ? (B - 1) / 2 : B / 2 - 1;
    const u128 off = (((1 + B) % Q) * (s % Q)) % Q;  // (1+B) < 2^63, s < 2^62
    auto digits_of = [&](u128 acc) {
        u128 x = (acc + off) % Q;
        return make_ulonglong2((uint64_t)(x % B), (uint64_t)(x / B));
    };

    CrtConst &cc = c->h_crt;
    memset(&cc, 0, sizeof cc);
    cc.Q = Q;
    cc.B = B;
    cc.offneg = (Q - off) % Q;
    {   // randomised flatten: v_i in [-xmax, xmax], digits shifted by s + xmax (utils.jl:204-216)
        const u128 xmax = (B & 1) ? (B - 1) / 2 * 3 : B / 2 * 3;
        const u128 stot = s + xmax;
        cc.xmax = (uint64_t)xmax;
        cc.offneg_rnd = (Q - (((1 + B) % Q) * (stot % Q)) % Q) % Q;
    }
    cc.DQ = ld128(c->par.DQ_tilde) % Q;
    cc.halfQ = Q / 2;
    cc.roundthr = Q / 2 + (Q & 1);
    cc.invQ = 1.0 / u128_dbl(Q);
    cc.invB = 1.0 / u128_dbl(B);
    cc.logr = logm + 1;
    cc.dig0 = digits_of(0);
    cc.digP = digits_of(cc.DQ);
    cc.digN = digits_of((Q - cc.DQ) % Q);
    u128 cM = 1 % Q;
    for (int i = 0; i < NPR; i++) cM = (cM * c->primes[i]) % Q;  // < 2^94 * 2^29
    for (int i = 0; i < NPR; i++) {
        u128 ci = 1 % Q;
        for (int j = 0; j < NPR; j++)
            if (j != i) ci = (ci * c->primes[j]) % Q;
        cc.c[i] = ci;
        cc.invp[i] = 1.0f / (float)c->primes[i];
    }
    const uint32_t plast = c->primes[NPR - 1];
    const u128 cH = (cc.c[NPR - 1] * ((plast - 1) / 2)) % Q;
    for (int a = 0; a < 6 * NPR + 2; a++) cc.T[a] = (Q - (((u128)a * cM) % Q + cH) % Q) % Q;
    auto limbs = [](u128 v, uint32_t *w, int n) {
        for (int i = 0; i < n; i++) w[i] = (uint32_t)(v >> (32 * i));
    };
    for (int i = 0; i < NPR; i++) { limbs(cc.c[i], cc.c32[i], 3); cc.cd[i] = u128_dbl(cc.c[i]); }
    for (int a = 0; a < 6 * NPR + 2; a++) { limbs(cc.T[a], cc.T32[a], 4); cc.Td[a] = u128_dbl(cc.T[a]); }
    limbs(Q, cc.Q32, 3);
    cc.Bd = u128_dbl(B);

    {   // k_crt_lean (kernels.h; tests/rns_model.py::CrtLean derives the same values)
        CrtLean &K = c->h_lean;
        memset(&K, 0, sizeof K);
        int nq = 0, nb = 0;
        while (nq < 128 && (Q >> nq)) nq++;
        while (nb < 64 && (B >> nb)) nb++;
        int NL = (nq + 28) / 29;
        if (NL < 2) NL = 2;
        const int t = nq - 29, a = 29 * (NL - 1) - t;
        if (nb >= 13 && NL <= 4 && nq >= 30 && a >= 0 && a <= 28) {
            auto lim = [&](u128 v, uint32_t *w) {
                for (int k = 0; k < 4; k++) w[k] = (uint32_t)((v >> (29 * k)) & 0x1FFFFFFFu);
            };
            for (int i = 0; i < NPR; i++) {
                lim(cc.c[i], K.c[i]);
                K.w[i] = (uint32_t)((1ull << 58) / c->primes[i]);
            }
            lim((Q - cM % Q) % Q, K.cMn);
            K.hoff = (plast - 1) / 2;
            limbs(Q, K.Qw, 3);
            K.B0 = (uint32_t)(B & 0x1FFFFFFFu);
            K.B1 = (uint32_t)(B >> 29);
            K.Bw0 = (uint32_t)B;
            K.Bw1 = (uint32_t)(B >> 32);
            // floor(2^(t + 72) / Q) < 2^44: long division, 2^(t + 72) has up to 138 bits
            {
                u128 rem = 0, quo = 0;
                for (int bit = t + 72; bit >= 0; bit--) {
                    rem = (rem << 1) | (bit == t + 72 ? 1 : 0);
                    quo <<= 1;
                    if (rem >= Q) { rem -= Q; quo |= 1; }
                }
                K.mq0 = (uint32_t)quo;
                K.mq1 = (uint32_t)(quo >> 32);
            }
            {   // floor(2^(nb + 51) / B) <= 2^52
                const u128 quo = ((u128)1 << (nb + 51)) / B;
                K.mb0 = (uint32_t)quo;
                K.mb1 = (uint32_t)(quo >> 32);
            }
            K.a = (uint32_t)a;
            K.t2 = (uint32_t)(nq > 63 ? nq - 63 : 0);
            K.sB = (uint32_t)(nb + 51 - (int)K.t2 - 64);
            K.nl = (uint32_t)NL;
            // randomised flatten: 2 xmax and (-2 xmax (1 + B)) mod Q  (2 xmax (1 + B) < 3.1 B^2 < 2^96)
            const u128 xm2 = (u128)2 * cc.xmax;
            lim((Q - (xm2 * (1 + B)) % Q) % Q, K.cR);
            K.xm2lo = (uint32_t)xm2;
            K.xm2hi = (uint32_t)(xm2 >> 32);
            // k_crt_lean_rnd adds the draws to the old stored digits: its hi operand reaches
            // Q / B + 6 B <= 7 B.  With two limbs (Q < 2^58) the term h1 B1 2^61 of hi B has no limb
            // sum to go to (crt_lean_one drops it), so the kernel is taken only where that term is
            // zero; and 6 B^2 has to stay far below the 2^34 Q of the residue sum for the quotient
            // estimate's width (tests/rns_model.py CrtLean.digits asserts both).  Reference
            // parameter sets have three limbs and B ~ sqrt(Q); others fall back to k_crt_acc.
            c->lean_rnd_ok = (NL >= 3 || ((7 * B) >> 32) == 0 || K.B1 == 0) && B * B <= (Q << 27);
        }
    }

    // twiddle tables and per-prime constants (all residues centred: |.| <= (p - 1) / 2)
    // per prime four tables of m entries: forward twiddles, inverse twiddles, and at + 2 m from each the
    // product twiddles of the radix-4 steps (ntt.h fwd_step4): twp[j] = +-tw[j >> 1] tw[j], minus for odd j
    std::vector<int32_t> tw((size_t)NPR * 4 * M), twq, pwt;
    std::vector<PrimeK> pk(NPR);
    cc.npr = npr;
    for (int i = 0; i < NPR; i++) {
        const uint32_t p = c->primes[i];
        if ((uint32_t)i >= c->tw_done) {   // (a second, smaller basis shares the tables of the first)
        uint32_t psi = 0;
        for (uint32_t x = 2; x < 2000 && !psi; x++) {
            uint32_t g = powmod32(x, (p - 1) / (2 * M), p);
            if (powmod32(g, M, p) == p - 1) psi = g;
        }
        if (!psi) return fail(c, SGFHE_ERR_UNSUPPORTED, "no primitive 2m-th root of unity");
        const uint32_t ipsi = powmod32(psi, p - 2, p);
        int32_t *f = tw.data() + (size_t)(4 * i) * M, *v = f + M, *fp = f + 2 * M, *vp = f + 3 * M;
        const uint32_t R1m = (uint32_t)((1ull << 32) % p);
        std::vector<uint32_t> pf(M), pv(M);   // plain values in table order
        uint32_t pw = 1, ipw = 1;
        for (uint32_t t = 0; t < M; t++) {
            const uint32_t br = bitrev(t, logm);
            pf[br] = pw;
            pv[br] = ipw;
            f[br] = centre32(mulmod32(pw, R1m, p), p);   // Montgomery form
            v[br] = centre32(mulmod32(ipw, R1m, p), p);
            pw = mulmod32(pw, psi, p);
            ipw = mulmod32(ipw, ipsi, p);
        }
        fp[0] = fp[1] = vp[0] = vp[1] = 0;
        for (uint32_t j = 2; j < M; j++) {
            const uint32_t a = mulmod32(mulmod32(pf[j >> 1], pf[j], p), R1m, p);
            const uint32_t b = mulmod32(mulmod32(pv[j >> 1], pv[j], p), R1m, p);
            fp[j] = centre32((j & 1) ? (p - a) % p : a, p);
            vp[j] = centre32((j & 1) ? (p - b) % p : b, p);
        }
        // quarter form of the latency kernels: after the first two Cooley-Tukey stages quarter q of the
        // array runs on the sub-tree of the twiddle table rooted at entry 4 + q, so its tables are a
        // re-indexing of the big ones: entry mm' + i' (mm' a power of two, i' < mm') = big entry
        // 4 mm' + q mm' + i'; the same for the inverse and for both product tables
        if (M >= 16) {
            const uint32_t MS = M / 4;
            twq.assign((size_t)4 * M, 0);
            for (uint32_t q = 0; q < 4; q++)
                for (uint32_t jj = 1; jj < MS; jj++) {
                    const uint32_t mmq = 1u << (31 - __builtin_clz(jj)), J = 4 * mmq + q * mmq + (jj - mmq);
                    int32_t *blk = twq.data() + (size_t)q * M;
                    blk[jj] = f[J];
                    blk[MS + jj] = v[J];
                    blk[2 * MS + jj] = jj >= 2 ? fp[J] : 0;
                    blk[3 * MS + jj] = jj >= 2 ? vp[J] : 0;
                }
            HIPCHK(c, hipMemcpy(c->d_twq + (size_t)(4 * i) * M, twq.data(), (size_t)4 * M * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        {   // psi^e R mod p, centred
            pwt.resize((size_t)2 * M);
            uint32_t pe = 1;
            for (uint32_t e = 0; e < 2 * M; e++) { pwt[e] = centre32(mulmod32(pe, R1m, p), p); pe = mulmod32(pe, psi, p); }
            HIPCHK(c, hipMemcpy(c->d_pow + (size_t)(2 * i) * M, pwt.data(), (size_t)2 * M * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        const int32_t head[8] = {f[1], M > 2 ? f[2] : 0, M > 2 ? f[3] : 0, M > 2 ? fp[2] : 0, M > 2 ? fp[3] : 0,
                                 v[1], M > 2 ? v[2] : 0, M > 2 ? v[3] : 0};
        memcpy(c->tw_head[i], head, sizeof head);
        }
        PrimeK &P = pk[i];
        memset(&P, 0, sizeof P);
        P.p = (int32_t)p;
        uint32_t inv = p;  // Newton: p^-1 mod 2^32
        for (int it = 0; it < 5; it++) inv *= 2u - p * inv;
        P.pinv = inv;
        const uint32_t R1 = (uint32_t)((1ull << 32) % p);
        const uint32_t Rinv = powmod32(R1, p - 2, p);
        const uint32_t R2 = mulmod32(R1, R1, p);
        P.r1 = centre32(R1, p);
        P.r2 = centre32(R2, p);
        P.r3 = centre32(mulmod32(R2, R1, p), p);
        P.sR = centre32((p - mulmod32((uint32_t)(s % p), Rinv, p)) % p, p);
        {
            const u128 xmax = (B & 1) ? (B - 1) / 2 * 3 : B / 2 * 3;
            P.sRr = centre32((p - mulmod32((uint32_t)((s + xmax) % p), Rinv, p)) % p, p);
        }
        P.hoff = (i == NPR - 1) ? (p - 1) / 2 : 0;
        P.qmodp = centre32((uint32_t)(Q % p), p);
        uint32_t Mi = 1;  // (M_rns / p_i) mod p_i
        for (int j = 0; j < NPR; j++)
            if (j != i) Mi = mulmod32(Mi, c->primes[j] % p, p);
        const uint32_t ei = powmod32(Mi, p - 2, p);
        const uint32_t minv = powmod32(M % p, p - 2, p);
        const uint32_t kappa = mulmod32(mulmod32(R2, minv, p), ei, p);
        P.kappaR = centre32(mulmod32(kappa, R1, p), p);
        P.minvR = centre32(mulmod32(minv, R1, p), p);
        P.invp = 1.0f / (float)p;
        P.twf = c->d_tw + (size_t)(4 * i) * M;
        P.twi = c->d_tw + (size_t)(4 * i + 1) * M;
        P.npr = npr;
        P.f1 = c->tw_head[i][0]; P.f2 = c->tw_head[i][1]; P.f3 = c->tw_head[i][2];
        P.fp2 = c->tw_head[i][3]; P.fp3 = c->tw_head[i][4];
        P.v1 = c->tw_head[i][5]; P.v2 = c->tw_head[i][6]; P.v3 = c->tw_head[i][7];
        P.twq = c->d_twq + (size_t)(4 * i) * M;
        P.pw = c->d_pow + (size_t)(2 * i) * M;
    }
    if (c->tw_done < npr) {
        const size_t off = (size_t)c->tw_done * 4 * M;
        HIPCHK(c, hipMemcpy(c->d_tw + off, tw.data() + off, ((size_t)npr * 4 * M - off) * sizeof(int32_t),
                            hipMemcpyHostToDevice));
        c->tw_done = npr;
    }
    HIPCHK(c, hipMalloc(&c->d_primes, NPR * sizeof(PrimeK)));
    c->shared->own(c->d_primes);
    HIPCHK(c, hipMemcpy(c->d_primes, pk.data(), NPR * sizeof(PrimeK), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&c->d_crt, sizeof(CrtConst)));
    c->shared->own(c->d_crt);
    HIPCHK(c, hipMemcpy(c->d_crt, &cc, sizeof(CrtConst), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&c->d_lean, sizeof(CrtLean)));
    c->shared->own(c->d_lean);
    HIPCHK(c, hipMemcpy(c->d_lean, &c->h_lean, sizeof(CrtLean), hipMemcpyHostToDevice));
    return SGFHE_OK;
}

int32_t build_constants(sgfhe_ctx *c) {
    const int logm = c->logm;
    const u128 Q = c->Q, B = c->B;

    // RNS primes: the largest primes below 2^29 with p = 1 mod 2^15 (2 m | p - 1 for every
    // supported m), as many as the exactness bound needs.  The exact integer the CRT has to
    // recover is D = (x^j - 1) sum_row u_row (*) C_row with |u| <= B / 2 (deterministic flatten,
    // utils.jl:155-189) and the key lifted to |C| <= Q / 2, so |D| <= 2 m B Q; k_crt_acc needs
    // |D| <= 0.4 M_rns, i.e. 5 m B Q <= M_rns.  The randomised flatten (|u| <= 2 B,
    // utils.jl:198-241) needs a factor 4 more.  Where that takes one more prime (Params(1024): six
    // against five) the ctx gets a basis per mode (sgfhe_ctx::Basis), unless it was created with
    // SGFHE_CTX_DETERMINISTIC_ONLY.
    uint32_t cand_primes[NPR_MAX];
    {
        int found = 0;
        for (uint64_t kk = ((1ull << 29) - 1) >> 15; kk > 0 && found < NPR_MAX; kk--) {
            uint32_t cand = (uint32_t)((kk << 15) + 1);
            if (cand < (1u << 29) && is_prime32(cand)) cand_primes[found++] = cand;
        }
        if (found < NPR_MAX) return fail(c, SGFHE_ERR_UNSUPPORTED, "could not find RNS primes");
    }
    const double log_mbq = logm + u128_log2(B) + u128_log2(Q);
    const double log_need = log2(5.0) + log_mbq + 0.001;
    auto fewest = [&](double target, double *have) {
        uint32_t npr = 0;
        double lh = 0;
        while (npr < NPR_MAX && (npr < 2 || lh < target)) lh += log2((double)cand_primes[npr++]);
        *have = lh;
        return lh >= target ? npr : 0u;
    };
    double have_det = 0, have_rnd = 0;
    const uint32_t npr_det = fewest(log_need, &have_det);
    if (!npr_det)
        return fail(c, SGFHE_ERR_UNSUPPORTED, "5 m B Q exceeds the product of the RNS primes");
    // |u| <= 2 B in the randomised mode: the exactness bound needs 4 x more head-room.  Its reductions
    // (random_digits, acc_from_digits, crt_reduce) divide values up to about 4 B^2 by Q with a
    // double-precision quotient estimate, exact while the quotient stays below 2^50.
    // (stored digits reach 4 B: above 2^48 they take the third plane of the digit record,
    // MODE_WIDE; B < 2^47 is checked at ctx creation)
    const bool rnd_width_ok = 2.0 * u128_log2(B) + 2.0 - u128_log2(Q) < 50.0;
    const uint32_t npr_rnd = fewest(log_need + 2.0, &have_rnd);
    const bool det_only = (c->create_flags & SGFHE_CTX_DETERMINISTIC_ONLY) != 0;
    c->nb = (!det_only && rnd_width_ok && npr_rnd > npr_det) ? 2 : 1;
    c->rnd_ok = rnd_width_ok && npr_rnd != 0 && (c->nb == 2 || npr_rnd == npr_det);
    c->npr_max = c->nb == 2 ? npr_rnd : npr_det;
    {
        const char *env = getenv("SGFHE_CRT_LEAN");
        c->use_lean = !(env && env[0] == '0');
        env = getenv("SGFHE_HOST_PIN");
        c->use_pin = !(env && env[0] == '0');
        env = getenv("SGFHE_SMALL_PADDED");
        c->small_padded = env && env[0] == '1';
        env = getenv("SGFHE_SMALL_LANES");
        if (env) { c->small_lanes = atoi(env) != 0; if (atoi(env) > 1) c->small_lanes_max = (uint32_t)atoi(env); }
        env = getenv("SGFHE_CRT1_GATES");
        c->crt1_max = (uint32_t)(env ? atoi(env) : 8) * 2u * c->M;
        // m = 16384: the latency form ends at 16 gates (20 / 24 gates take 114 / 144 ms in it, 112 / 112 in the
        // throughput form; 12 / 14 gates 86 / 93 against 107; profiles/r04_exp_small_rings_lanes.txt)
        if (c->logm >= 14) { c->small_max = 16; if (c->small_lanes_max > 16) c->small_lanes_max = 16; }
    }
    HIPCHK(c, hipMalloc(&c->d_tw, (size_t)c->npr_max * 4 * c->M * sizeof(int32_t)));
    c->shared->own(c->d_tw);
    HIPCHK(c, hipMalloc(&c->d_twq, (size_t)c->npr_max * 4 * c->M * sizeof(int32_t)));
    c->shared->own(c->d_twq);
    HIPCHK(c, hipMalloc(&c->d_pow, (size_t)c->npr_max * 2 * c->M * sizeof(int32_t)));
    c->shared->own(c->d_pow);
    {
        const char *env = getenv("SGFHE_SMALL_SPLIT");
        if (env) c->split_max = (uint32_t)atoi(env);
        env = getenv("SGFHE_ITER_ALL");
        c->iter_all = env && env[0] == '1';
        env = getenv("SGFHE_SMALL_FUSED");
        if (env) c->fused_min = (uint32_t)atoi(env);
    }
    HIPCHK(c, hipMalloc(&c->d_bad, sizeof(uint32_t)));
    HIPCHK(c, hipMemset(c->d_bad, 0, sizeof(uint32_t)));
    // the larger basis first: its pass puts every prime's twiddle tables on the device
    for (int b = c->nb - 1; b >= 0; b--) {
        const bool big = b == c->nb - 1 && c->nb == 2;
        int32_t rc = build_basis(c, big ? npr_rnd : npr_det, cand_primes, big ? have_rnd : have_det, log_mbq);
        if (rc) return rc;
        c->d_key = nullptr;
        c->key_bytes = 0;
        save_basis(c, b);
    }
    activate(c, 0);
    return SGFHE_OK;
}

// Key uploads, generation and imports work on the LARGER basis (its key determines the other one);
// key_finish derives the smaller basis's key from it and goes back to the basis of the present mode.
// A key that clones share (sgfhe_ctx_clone) is read-only: their calls may be reading it on the device.
int32_t key_begin(sgfhe_ctx *c) {
    if (c->shared.use_count() > 1)
        return fail(c, SGFHE_ERR_INVALID_ARG,
                    "the bootstrap key of this ctx is shared with clones (sgfhe_ctx_clone): destroy them before replacing it");
    activate(c, c->nb - 1);
    return SGFHE_OK;
}
// A key writer calls this once its arguments are accepted, before the first byte of the key changes:
// from here to the end of key_finish the ctx has no key, so a writer that fails part-way (or whose second
// key form cannot be derived) leaves a ctx that refuses to bootstrap instead of one whose two bases hold
// different keys.
void key_dirty(sgfhe_ctx *c) { c->have_key = false; }
int32_t key_finish(sgfhe_ctx *c, int32_t rc) {
    if (rc == SGFHE_OK && c->nb == 2) {
        const sgfhe_ctx::Basis &Bg = c->basis[1];
        sgfhe_ctx::Basis &Sm = c->basis[0];
        hipError_t e = hipSuccess;
        if (!Sm.d_key) {
            Sm.key_bytes = (size_t)c->n * Sm.npr * 8 * c->M * 4;
            e = hipMalloc(&Sm.d_key, Sm.key_bytes);
            if (e == hipSuccess) c->shared->own(Sm.d_key);
        }
        if (e == hipSuccess) {
            KeyFactors fac = {};
            for (uint32_t i = 0; i < Sm.npr; i++) {
                const uint32_t p = Sm.primes[i];
                uint32_t f = (uint32_t)((1ull << 32) % p);                   // R: Montgomery form
                for (uint32_t j = Sm.npr; j < Bg.npr; j++) f = mulmod32(f, Bg.primes[j] % p, p);
                fac.f[i] = centre32(f, p);
            }
            const size_t total = Sm.key_bytes / 4;
            hipLaunchKernelGGL(k_key_derive, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                               Bg.d_key, Sm.d_key, Sm.d_primes, fac, Bg.npr, Sm.npr, (uint32_t)c->logm, total);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
        if (e != hipSuccess) { rc = fail(c, SGFHE_ERR_HIP, hipGetErrorString(e)); c->have_key = false; }
    }
    activate(c, mode_basis(c));
    return rc;
}

// Device-form key blob = 64-byte header + payload keyhat[k][prime][row * 2 + col][slot].  The
// header pins everything the payload's meaning depends on, so a blob made for another parameter
// set (or by another arithmetic revision of this library) is refused by the import.
struct KeyBlobHeader {
    char magic[8];         // "SGFHEKEY"
    uint32_t version;      // SGFHE_KEY_BLOB_VERSION: layout + residue form of the payload
    uint32_t n, m, npr;
    uint64_t Q[2];
    uint64_t B;
    uint64_t prime_hash;   // FNV-1a over the RNS primes in use
    uint64_t payload_bytes;
};
static_assert(sizeof(KeyBlobHeader) == 64, "key blob header is 64 bytes");
constexpr uint32_t SGFHE_KEY_BLOB_VERSION = 2;

KeyBlobHeader blob_header(const sgfhe_ctx *c) {
    KeyBlobHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "SGFHEKEY", 8);
    h.version = SGFHE_KEY_BLOB_VERSION;
    h.n = c->n;
    h.m = c->M;
    h.npr = c->npr;
    h.Q[0] = (uint64_t)c->Q;
    h.Q[1] = (uint64_t)(c->Q >> 64);
    h.B = (uint64_t)c->B;
    uint64_t f = 0xcbf29ce484222325ull;
    for (uint32_t i = 0; i < c->npr; i++)
        for (int b = 0; b < 4; b++) { f ^= (c->primes[i] >> (8 * b)) & 0xffu; f *= 0x100000001b3ull; }
    h.prime_hash = f;
    h.payload_bytes = (uint64_t)c->n * c->npr * 8 * c->M * 4;
    return h;
}

int32_t key_alloc(sgfhe_ctx *c) {
    if (c->d_key) return SGFHE_OK;
    c->key_bytes = (size_t)c->n * c->npr * 8 * c->M * 4;
    HIPCHK(c, hipMalloc(&c->d_key, c->key_bytes));
    c->shared->own(c->d_key);
    c->basis[c->active].d_key = c->d_key;
    c->basis[c->active].key_bytes = c->key_bytes;
    return SGFHE_OK;
}

// canonical [npolys][m][2 words] in host memory -> NTT-domain key polys poly0.. in keyhat
int32_t key_transform_host(sgfhe_ctx *c, const uint64_t *canon, uint32_t npolys, int32_t *keyhat,
                           const Rns2Const *rns2 = nullptr) {
    const size_t poly_bytes = (size_t)c->M * 16;
    uint32_t stage_polys = (uint32_t)(((size_t)64 << 20) / poly_bytes);
    if (stage_polys < 8) stage_polys = 8;
    if (stage_polys > npolys) stage_polys = npolys;
    ulonglong2 *d_stage = nullptr;
    HIPCHK(c, hipMalloc(&d_stage, (size_t)stage_polys * poly_bytes));
    int32_t rc = SGFHE_OK;
    for (uint32_t p0 = 0; p0 < npolys && rc == SGFHE_OK; p0 += stage_polys) {
        const uint32_t np = (npolys - p0 < stage_polys) ? npolys - p0 : stage_polys;
        hipError_t e = hipMemcpyAsync(d_stage, canon + (size_t)p0 * c->M * 2, (size_t)np * poly_bytes,
                                      hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { rc = fail(c, SGFHE_ERR_HIP, hipGetErrorString(e)); break; }
        if (rns2) {  // (v1, v2) limb pairs -> canonical residues, in place (rns.jl:32-40)
            const size_t cnt = (size_t)np * c->M;
            hipLaunchKernelGGL(k_rns2_to_canon, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0,
                               c->stream, d_stage, cnt, *rns2, c->d_crt, c->d_bad);
        }
        rc = launch_keytr(c, d_stage, keyhat, p0, np, c->stream);
        if (rc) break;
        e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, SGFHE_ERR_HIP, hipGetErrorString(e));
    }
    (void)hipFree(d_stage);
    return rc;
}

int32_t create_streams(sgfhe_ctx *c) {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream_io, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream_io2, hipStreamNonBlocking) != hipSuccess)
        return fail(c, SGFHE_ERR_HIP, "hipStreamCreate / hipEventCreate failed");
    return SGFHE_OK;
}

}  // namespace

// ==================================================================================================
// C ABI
// ==================================================================================================

extern "C" {

const char *sgfhe_version(void) { return "sgfhe_hip 0.5.0 gfx950"; }

uint32_t sgfhe_abi_version(void) { return SGFHE_ABI_VERSION; }

// The Makefile passes the hash of the kernel sources (and any extra -D flags of an ablation
// build); the marker prefix lets tools find the id in the file without loading it.
#ifndef SGFHE_BUILD_ID
#define SGFHE_BUILD_ID "unknown"
#endif
const char *sgfhe_build_id(void) {
    static const char id[] = "SGFHE_BUILD_ID=" SGFHE_BUILD_ID;
    return id + 15;
}

// The message of the last failed call on this ctx, copied for the calling thread (the ctx may be
// shared): valid until the same thread asks again.
const char *sgfhe_last_error_string(const sgfhe_ctx *ctx) {
    if (!ctx) return "null context";
    static thread_local std::string copy;
    SGFHE_LOCK(ctx);
    copy = ctx->err;
    return copy.c_str();
}

int32_t sgfhe_ctx_create(const sgfhe_params *p, int device, sgfhe_ctx **out) {
    return sgfhe_ctx_create_ex(p, device, 0u, out);
}

int32_t sgfhe_ctx_create_ex(const sgfhe_params *p, int device, uint32_t flags, sgfhe_ctx **out) {
    if (!p || !out) return SGFHE_ERR_INVALID_ARG;
    *out = nullptr;
    if (flags & ~(uint32_t)(SGFHE_CTX_RANDOM_FLATTEN | SGFHE_CTX_DETERMINISTIC_ONLY)) return SGFHE_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return SGFHE_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return SGFHE_ERR_NO_DEVICE;
    sgfhe_ctx *c = new (std::nothrow) sgfhe_ctx();
    if (!c) return SGFHE_ERR_OOM;
    *out = c;  // returned even on failure so that the caller can read the error string
    c->par = *p;
    c->device = device;
    c->create_flags = flags;
    try {
        c->shared = std::make_shared<SharedDev>();
    } catch (...) {
        return fail(c, SGFHE_ERR_OOM, "out of host memory");
    }
    c->shared->device = device;
    {   // SGFHE_COALESCE=0 in the environment: ctxs sharing a key never gather their calls (A/B measurements)
        const char *env = getenv("SGFHE_COALESCE");
        if (env && env[0] == '0') c->shared->co.enabled = false;
    }
    const uint64_t m = p->m;
    if (p->ell != 2) return fail(c, SGFHE_ERR_UNSUPPORTED, "ell must be 2 (fhe.jl:576)");
    if (m < 64 || m > 16384 || (m & (m - 1)))
        return fail(c, SGFHE_ERR_UNSUPPORTED, "m must be a power of two in [2^6, 2^14]");
    if (p->r != 2 * m) return fail(c, SGFHE_ERR_INVALID_ARG, "r must equal 2 m (fhe.jl:62)");
    if (p->n == 0 || p->n > m / 4 || (p->n & (p->n - 1)))
        return fail(c, SGFHE_ERR_INVALID_ARG,
                    "n must be a power of two in [1, m / 4] (fhe.jl:45-46; extract, fhe.jl:585-590)");
    c->M = (uint32_t)m;
    c->n = (uint32_t)p->n;
    while ((1u << c->logm) < c->M) c->logm++;
    c->Q = ld128(p->Q);
    c->B = ld128(p->B);
    if (c->Q < 3 || (c->Q >> 94)) return fail(c, SGFHE_ERR_UNSUPPORTED, "Q must be in [3, 2^94)");
    if (c->B < 2 || (c->B >> 47)) return fail(c, SGFHE_ERR_UNSUPPORTED, "B must be in [2, 2^47)");
    {
        // B^2 >= Q (utils.jl:145); B < 2^46 so B*B fits 128 bits
        if (c->B * c->B < c->Q) return fail(c, SGFHE_ERR_INVALID_ARG, "B^2 must be >= Q");
    }
    if (ld128(p->DQ_tilde) >= c->Q) return fail(c, SGFHE_ERR_INVALID_ARG, "DQ_tilde must be < Q");
    int32_t rc = create_streams(c);
    if (rc) return rc;
    return build_constants(c);
}

int32_t sgfhe_ctx_clone(sgfhe_ctx *src, sgfhe_ctx **out) {
    if (!src || !out) return SGFHE_ERR_INVALID_ARG;
    *out = nullptr;
    SGFHE_LOCK(src);
    if (!src->have_key) return fail(src, SGFHE_ERR_NO_KEY, "ctx_clone: the ctx has no bootstrap key to share");
    if (hipSetDevice(src->device) != hipSuccess) return fail(src, SGFHE_ERR_NO_DEVICE, "ctx_clone: hipSetDevice failed");
    sgfhe_ctx *c = new (std::nothrow) sgfhe_ctx();
    if (!c) return fail(src, SGFHE_ERR_OOM, "out of host memory");
    *out = c;  // returned even on failure so that the caller can read the error string (and destroys it)
    // what a bootstrap only reads: parameter set, bases with their device constants and key forms, tables
    c->par = src->par;
    c->device = src->device;
    c->logm = src->logm;
    c->M = src->M;
    c->n = src->n;
    c->Q = src->Q;
    c->B = src->B;
    c->create_flags = src->create_flags;
    c->shared = src->shared;
    for (int b = 0; b < 2; b++) c->basis[b] = src->basis[b];
    c->nb = src->nb;
    c->npr_max = src->npr_max;
    c->tw_done = src->tw_done;
    c->d_tw = src->d_tw;
    c->d_twq = src->d_twq;
    c->d_pow = src->d_pow;
    memcpy(c->tw_head, src->tw_head, sizeof c->tw_head);
    c->rnd_ok = src->rnd_ok;
    c->have_rns2 = src->have_rns2;
    c->rns2 = src->rns2;
    c->have_key = true;
    // the scheduling knobs as they stand (sgfhe_set_chunk / _lanes / _small_batch_max and the environment)
    c->chunk = src->chunk;
    c->lanes = src->lanes;
    c->small_max = src->small_max;
    c->small_lanes = src->small_lanes;
    c->small_lanes_max = src->small_lanes_max;
    c->small_padded = src->small_padded;
    c->crt1_max = src->crt1_max;
    c->split_max = src->split_max;
    c->fused_min = src->fused_min;
    c->iter_all = src->iter_all;
    c->use_lean = src->use_lean;
    c->use_pin = src->use_pin;
    // its own: flatten mode (deterministic, call counter 0), lanes' work buffers, streams, events, staging,
    // timing, the error string and the lock
    activate(c, 0);
    // The runtime deals streams onto its hardware queues (GPU_MAX_HW_QUEUES = 4) in the order they are created, and
    // packets of one hardware queue run in order: with four streams per ctx, the FIRST stream of every ctx -- the
    // one its calls run on -- would sit on the same queue and calls on different clones could never overlap on the
    // device (tools/ubench_streams.hip with UB_EXTRA_STREAMS=3: x1.00 at any number of threads,
    // profiles/r05_concurrent.txt).  A clone therefore shifts its streams by its number among the sharers: a few
    // throw-away streams first, destroyed once its own exist.  (What the gathering of small calls does not cover --
    // calls above its size limit, the asynchronous entry point -- then overlaps as far as the device lets two
    // chains overlap.)
    hipStream_t skip[3] = {nullptr, nullptr, nullptr};
    const unsigned shift = c->shared->clones.fetch_add(1) % 4u + 1u;       // 1, 2, 3, 0, 1, ... (the creator ctx has 0)
    for (unsigned i = 0; i < shift % 4u; i++) (void)hipStreamCreateWithFlags(&skip[i], hipStreamNonBlocking);
    int32_t rc = create_streams(c);
    for (hipStream_t x : skip)
        if (x) (void)hipStreamDestroy(x);
    if (rc) return rc;
    HIPCHK(c, hipMalloc(&c->d_bad, sizeof(uint32_t)));
    HIPCHK(c, hipMemset(c->d_bad, 0, sizeof(uint32_t)));
    return SGFHE_OK;
}

int32_t sgfhe_ctx_destroy(sgfhe_ctx *c) {
    if (!c) return SGFHE_OK;
    (void)hipSetDevice(c->device);
    (void)drain(c);   // work of an asynchronous call on a caller's stream
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->stream_io) (void)hipStreamSynchronize(c->stream_io);
    if (c->stream_io2) (void)hipStreamSynchronize(c->stream_io2);
    timing_flush(c);
    free_lanes(c);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->stream_io) (void)hipStreamDestroy(c->stream_io);
    if (c->stream_io2) (void)hipStreamDestroy(c->stream_io2);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->shared) c->shared->co.forget(c);
    c->shared.reset();   // constants, tables and key: freed with the last ctx that shares them
    if (c->io_in) (void)hipFree(c->io_in);
    if (c->io_out) (void)hipFree(c->io_out);
    if (c->pin_in) (void)hipHostFree(c->pin_in);
    if (c->pin_out) (void)hipHostFree(c->pin_out);
    if (c->d_bad) (void)hipFree(c->d_bad);
    if (c->d_rows) (void)hipFree(c->d_rows);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SGFHE_OK;
}

int32_t sgfhe_set_chunk(sgfhe_ctx *c, uint32_t chunk) {
    if (!c) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    // the kernels index a chunk's planes with 32-bit byte offsets: residues [chunk][2][npr][m] x 4 B
    // and digit records [chunk][2] x 16 m B must stay below 4 GiB
    const uint64_t cap_y = 0xFFFFFFFFull / ((uint64_t)2 * c->npr_max * c->M * 4);
    const uint64_t cap_d = 0xFFFFFFFFull / ((uint64_t)2 * 16 * c->M);
    const uint64_t cap = (cap_y < cap_d ? cap_y : cap_d) & ~7ull;
    if (chunk > cap) return fail(c, SGFHE_ERR_INVALID_ARG, "chunk too large for this ring size");
    c->chunk = chunk ? round_up8(chunk) : 0;
    return SGFHE_OK;
}

static int32_t set_random_flatten(sgfhe_ctx *c, int enable, const ChaChaKey &key) {
    if (enable && !c->rnd_ok)
        return fail(c, SGFHE_ERR_UNSUPPORTED,
                    "randomised flatten: not available on this ctx (created with SGFHE_CTX_DETERMINISTIC_ONLY "
                    "where it needs a prime more, 20 m B Q beyond seven primes, or B^2 > 2^48 Q)");
    c->rnd = enable != 0;
    c->rnd_key = key;
    c->rnd_call = 0;
    activate(c, mode_basis(c));   // constants and key form of the mode's basis (queued work keeps its own)
    return SGFHE_OK;
}

int32_t sgfhe_set_random_flatten(sgfhe_ctx *c, int enable, uint64_t seed) {
    if (!c) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    ChaChaKey key = {};   // the seed as 32 little-endian bytes
    key.k[0] = (uint32_t)seed;
    key.k[1] = (uint32_t)(seed >> 32);
    return set_random_flatten(c, enable, key);
}

int32_t sgfhe_set_random_flatten_key(sgfhe_ctx *c, int enable, const uint8_t *key32) {
    if (!c || (enable && !key32)) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    ChaChaKey key = {};
    if (key32)
        for (int i = 0; i < 8; i++)
            key.k[i] = (uint32_t)key32[4 * i] | ((uint32_t)key32[4 * i + 1] << 8) |
                       ((uint32_t)key32[4 * i + 2] << 16) | ((uint32_t)key32[4 * i + 3] << 24);
    return set_random_flatten(c, enable, key);
}

int32_t sgfhe_set_small_batch_max(sgfhe_ctx *c, uint32_t max_bootstraps) {
    if (!c) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    if (max_bootstraps > 256)
        return fail(c, SGFHE_ERR_INVALID_ARG, "small-batch form: at most 256 bootstraps");
    (void)hipSetDevice(c->device);
    (void)drain(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    free_lanes(c);  // the staging buffer is sized from this value
    c->small_max = max_bootstraps;
    return SGFHE_OK;
}

int32_t sgfhe_set_lanes(sgfhe_ctx *c, uint32_t lanes) {
    if (!c || lanes < 1 || lanes > 2) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    c->lanes = lanes;
    return SGFHE_OK;
}

static int32_t sgfhe_bkey_upload_impl(sgfhe_ctx *c, const uint64_t *canonical, size_t n_words) {
    if (!c || !canonical) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    (void)hipSetDevice(c->device);
    SGFHE_QUIESCE(c);
    const size_t expect = (size_t)c->n * 8 * c->M * 2;
    if (n_words != expect) return fail(c, SGFHE_ERR_INVALID_ARG, "bkey_upload: n_words != n*8*m*2");
    int32_t rc = key_alloc(c);
    if (rc) return rc;
    HIPCHK(c, hipMemset(c->d_bad, 0, sizeof(uint32_t)));
    key_dirty(c);
    rc = key_transform_host(c, canonical, c->n * 8, c->d_key);
    if (rc) return rc;
    uint32_t bad = 0;
    HIPCHK(c, hipMemcpy(&bad, c->d_bad, sizeof bad, hipMemcpyDeviceToHost));
    if (bad) {
        c->have_key = false;
        return fail(c, SGFHE_ERR_INVALID_ARG, "bkey_upload: a residue is not in [0, Q)");
    }
    c->have_key = true;
    return SGFHE_OK;
}

int32_t sgfhe_bkey_upload(sgfhe_ctx *c, const uint64_t *canonical, size_t n_words) {
    if (!c) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    const int32_t rb = key_begin(c);   // on the larger basis; the other basis's key is derived from it
    if (rb) return rb;
    return key_finish(c, sgfhe_bkey_upload_impl(c, canonical, n_words));
}

static int32_t sgfhe_bkey_generate_impl(sgfhe_ctx *c, const uint64_t *sk, size_t n_sk, const uint8_t *seed32,
                            uint32_t noise) {
    if (!c || !sk || !seed32) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    if (c->M < 8) return fail(c, SGFHE_ERR_UNSUPPORTED, "bkey_generate needs m >= 8");
    ChaChaKey ck;
    for (int i = 0; i < 8; i++)
        ck.k[i] = (uint32_t)seed32[4 * i] | ((uint32_t)seed32[4 * i + 1] << 8) |
                  ((uint32_t)seed32[4 * i + 2] << 16) | ((uint32_t)seed32[4 * i + 3] << 24);
    if (n_sk != c->n) return fail(c, SGFHE_ERR_INVALID_ARG, "bkey_generate: secret key must hold n bits");
    if (c->Q < ((u128)1 << 16)) return fail(c, SGFHE_ERR_UNSUPPORTED, "bkey_generate needs Q >= 2^16");
    // e in [-noise, noise] is formed in int32 and lifted as Q - |e| (k_keygen_draw / k_keygen_finish)
    if (noise >= (1u << 30) || (u128)noise * 2 >= c->Q)
        return fail(c, SGFHE_ERR_INVALID_ARG, "bkey_generate: noise must be below 2^30 and below Q / 2");
    (void)hipSetDevice(c->device);
    SGFHE_QUIESCE(c);
    int32_t rc = key_alloc(c);
    if (rc) return rc;
    key_dirty(c);
    const uint32_t M = c->M, rows = c->n * 4;
    uint32_t R = 64;  // rows per batch (16 key slices)
    if (R > rows) R = rows;
    uint64_t *d_sk = nullptr;
    int32_t *d_shat = nullptr;
    uint32_t *d_y = nullptr;
    ulonglong2 *d_acan = nullptr, *d_prod = nullptr, *d_canon = nullptr;
    int32_t *d_e = nullptr;
    hipError_t e = hipSuccess;
    do {
        if ((e = hipMalloc(&d_sk, (size_t)c->n * 8))) break;
        if ((e = hipMalloc(&d_shat, (size_t)c->npr * M * 4))) break;
        if ((e = hipMalloc(&d_y, (size_t)R * c->npr * M * 4))) break;
        if ((e = hipMalloc(&d_acan, (size_t)R * M * 16))) break;
        if ((e = hipMalloc(&d_prod, (size_t)R * M * 16))) break;
        if ((e = hipMalloc(&d_canon, (size_t)R * 2 * M * 16))) break;
        if ((e = hipMalloc(&d_e, (size_t)R * M * 4))) break;
        if ((e = hipMemcpyAsync(d_sk, sk, (size_t)c->n * 8, hipMemcpyHostToDevice, c->stream))) break;
        rc = launch_keygen_ntt(c, d_sk, d_shat, nullptr, nullptr, 0, true, c->stream);
        if (rc) break;
        for (uint32_t row0 = 0; row0 < rows && rc == SGFHE_OK; row0 += R) {
            const uint32_t tot = R * M;
            hipLaunchKernelGGL(k_keygen_draw, dim3((tot / 4 + 255) / 256), dim3(256), 0, c->stream,
                               d_acan, d_e, c->d_crt, ck, noise, row0, R, (uint32_t)c->logm);
            rc = launch_keygen_ntt(c, d_sk, d_shat, d_acan, d_y, R, false, c->stream);
            if (rc) break;
            // CRT of the exact product, canonical residues into d_prod ([row][m] 16-byte values)
            if ((rc = launch_crt_raw(c, d_y, reinterpret_cast<uint64_t *>(d_prod), tot,
                                     MODE_NOACC | MODE_CANON, c->stream, RndArgs{}, 0u)))
                break;
            hipLaunchKernelGGL(k_keygen_finish, dim3((tot + 255) / 256), dim3(256), 0, c->stream, d_acan,
                               d_prod, d_e, d_sk, d_canon, c->d_crt, row0, R, (uint32_t)c->logm);
            if ((e = hipGetLastError())) break;
            rc = launch_keytr(c, d_canon, c->d_key, row0 * 2, R * 2, c->stream);
        }
        if (rc || e) break;
        e = hipStreamSynchronize(c->stream);
    } while (0);
    if (e != hipSuccess && rc == SGFHE_OK) rc = fail(c, SGFHE_ERR_HIP, hipGetErrorString(e));
    for (void *ptr : {(void *)d_sk, (void *)d_shat, (void *)d_y, (void *)d_acan, (void *)d_prod,
                      (void *)d_canon, (void *)d_e})
        if (ptr) (void)hipFree(ptr);
    if (rc == SGFHE_OK) c->have_key = true;
    return rc;
}

int32_t sgfhe_bkey_generate(sgfhe_ctx *c, const uint64_t *sk, size_t n_sk, const uint8_t *seed32,
                            uint32_t noise) {
    if (!c) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    const int32_t rb = key_begin(c);   // on the larger basis; the other basis's key is derived from it
    if (rb) return rb;
    return key_finish(c, sgfhe_bkey_generate_impl(c, sk, n_sk, seed32, noise));
}

static int32_t rns2_configure(sgfhe_ctx *c, uint64_t m1, uint64_t m2) {
    if ((u128)m1 * m2 != c->Q) return fail(c, SGFHE_ERR_INVALID_ARG, "m1 * m2 != Q");
    if (m1 >= (1ull << 47) || m2 >= (1ull << 47) || m1 < 2 || m2 < 2)
        return fail(c, SGFHE_ERR_UNSUPPORTED, "RNS2 limb moduli must be in [2, 2^47)");
    // CRT of rns.jl:32-40: c1 = m2^(m1-1) mod Q = m2 (m2^-1 mod m1), c2 = m1 (m1^-1 mod m2)
    // (Fermat idempotents, m1 and m2 prime)
    auto powmod64 = [](uint64_t a, uint64_t e, uint64_t md) {
        u128 r = 1, b = a % md;
        while (e) { if (e & 1) r = r * b % md; b = b * b % md; e >>= 1; }
        return (uint64_t)r;
    };
    Rns2Const rc;
    rc.m1 = m1;
    rc.m2 = m2;
    rc.i21 = powmod64(m2 % m1, m1 - 2, m1);
    rc.i12 = powmod64(m1 % m2, m2 - 2, m2);
    if ((u128)(m2 % m1) * rc.i21 % m1 != 1 || (u128)(m1 % m2) * rc.i12 % m2 != 1)
        return fail(c, SGFHE_ERR_INVALID_ARG, "RNS2 limb moduli must be distinct primes");
    rc.inv1 = 1.0 / (double)m1;
    rc.inv2 = 1.0 / (double)m2;
    c->rns2 = rc;
    c->have_rns2 = true;
    return SGFHE_OK;
}

static int32_t sgfhe_bkey_upload_rns2_impl(sgfhe_ctx *c, const uint64_t *pairs, size_t n_words, uint64_t m1,
                               uint64_t m2) {
    if (!c || !pairs) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    (void)hipSetDevice(c->device);
    SGFHE_QUIESCE(c);
    const size_t expect = (size_t)c->n * 8 * c->M * 2;
    if (n_words != expect) return fail(c, SGFHE_ERR_INVALID_ARG, "bkey_upload_rns2: bad n_words");
    int32_t rc = rns2_configure(c, m1, m2);
    if (rc) return rc;
    rc = key_alloc(c);
    if (rc) return rc;
    HIPCHK(c, hipMemset(c->d_bad, 0, sizeof(uint32_t)));
    key_dirty(c);
    rc = key_transform_host(c, pairs, c->n * 8, c->d_key, &c->rns2);
    if (rc) return rc;
    uint32_t bad = 0;
    HIPCHK(c, hipMemcpy(&bad, c->d_bad, sizeof bad, hipMemcpyDeviceToHost));
    if (bad) {
        c->have_key = false;
        return fail(c, SGFHE_ERR_INVALID_ARG, "bkey_upload_rns2: a limb is not in [0, m_i)");
    }
    c->have_key = true;
    return SGFHE_OK;
}

int32_t sgfhe_bkey_upload_rns2(sgfhe_ctx *c, const uint64_t *pairs, size_t n_words, uint64_t m1,
                               uint64_t m2) {
    if (!c) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    const int32_t rb = key_begin(c);   // on the larger basis; the other basis's key is derived from it
    if (rb) return rb;
    return key_finish(c, sgfhe_bkey_upload_rns2_impl(c, pairs, n_words, m1, m2));
}

int32_t sgfhe_rns2_convert(sgfhe_ctx *c, int to_pairs, const uint64_t *in, size_t count, uint64_t m1,
                           uint64_t m2, uint64_t *out) {
    if (!c || !in || !out) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    if (count == 0) return SGFHE_OK;
    (void)hipSetDevice(c->device);
    SGFHE_QUIESCE(c);
    int32_t rc = rns2_configure(c, m1, m2);
    if (rc) return rc;
    ulonglong2 *d = nullptr;
    HIPCHK(c, hipMalloc(&d, count * 16));
    hipError_t e = hipSuccess;
    uint32_t bad = 0;
    do {
        if ((e = hipMemsetAsync(c->d_bad, 0, sizeof(uint32_t), c->stream))) break;
        if ((e = hipMemcpyAsync(d, in, count * 16, hipMemcpyHostToDevice, c->stream))) break;
        const dim3 grid((unsigned)((count + 255) / 256));
        if (to_pairs)
            hipLaunchKernelGGL(k_canon_to_rns2, grid, dim3(256), 0, c->stream, d, count, c->rns2);
        else
            hipLaunchKernelGGL(k_rns2_to_canon, grid, dim3(256), 0, c->stream, d, count, c->rns2,
                               c->d_crt, c->d_bad);
        if ((e = hipGetLastError())) break;
        if ((e = hipMemcpyAsync(out, d, count * 16, hipMemcpyDeviceToHost, c->stream))) break;
        if ((e = hipMemcpyAsync(&bad, c->d_bad, sizeof bad, hipMemcpyDeviceToHost, c->stream))) break;
        e = hipStreamSynchronize(c->stream);
    } while (0);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, SGFHE_ERR_HIP, hipGetErrorString(e));
    if (to_pairs) {  // canonical inputs must be below Q (checked on the host copy: debug hook)
        for (size_t i = 0; i < count; i++)
            if (ld128(in + 2 * i) >= c->Q) return fail(c, SGFHE_ERR_INVALID_ARG, "rns2_convert: value >= Q");
    } else if (bad) {
        return fail(c, SGFHE_ERR_INVALID_ARG, "rns2_convert: a limb is not in [0, m_i)");
    }
    return SGFHE_OK;
}

int32_t sgfhe_bkey_device_form_bytes(const sgfhe_ctx *c, size_t *bytes) {
    if (!c || !bytes) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    *bytes = sizeof(KeyBlobHeader) + (size_t)c->n * c->npr_max * 8 * c->M * 4;   // the larger basis's key
    return SGFHE_OK;
}

static int32_t sgfhe_bkey_export_device_form_impl(sgfhe_ctx *c, void *dst) {
    if (!c || !dst) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    if (!c->have_key) return fail(c, SGFHE_ERR_NO_KEY, "no bootstrap key uploaded");
    (void)hipSetDevice(c->device);
    SGFHE_QUIESCE(c);
    const KeyBlobHeader h = blob_header(c);
    HIPCHK(c, hipMemcpy(dst, &h, sizeof h, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpyAsync(static_cast<char *>(dst) + sizeof h, c->d_key, c->key_bytes,
                             hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SGFHE_OK;
}

int32_t sgfhe_bkey_export_device_form(sgfhe_ctx *c, void *dst) {
    if (!c) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    activate(c, c->nb - 1);   // the blob is the key of the larger basis
    const int32_t rc = sgfhe_bkey_export_device_form_impl(c, dst);
    activate(c, mode_basis(c));
    return rc;
}

static int32_t sgfhe_bkey_import_device_form_impl(sgfhe_ctx *c, const void *src) {
    if (!c || !src) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    (void)hipSetDevice(c->device);
    SGFHE_QUIESCE(c);
    KeyBlobHeader got;
    HIPCHK(c, hipMemcpy(&got, src, sizeof got, hipMemcpyDeviceToHost));
    const KeyBlobHeader want = blob_header(c);
    if (memcmp(got.magic, want.magic, sizeof want.magic) != 0)
        return fail(c, SGFHE_ERR_INVALID_ARG, "bkey_import: not a device-form key blob (bad magic)");
    if (got.version != want.version)
        return fail(c, SGFHE_ERR_INVALID_ARG, "bkey_import: blob format version differs from this library");
    if (memcmp(&got, &want, sizeof want) != 0)
        return fail(c, SGFHE_ERR_INVALID_ARG,
                    "bkey_import: the blob belongs to another parameter set (n, m, Q, B or RNS primes differ)");
    int32_t rc = key_alloc(c);
    if (rc) return rc;
    key_dirty(c);
    HIPCHK(c, hipMemcpyAsync(c->d_key, static_cast<const char *>(src) + sizeof got, c->key_bytes,
                             hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_key = true;
    return SGFHE_OK;
}

int32_t sgfhe_bkey_import_device_form(sgfhe_ctx *c, const void *src) {
    if (!c) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    const int32_t rb = key_begin(c);   // on the larger basis; the other basis's key is derived from it
    if (rb) return rb;
    return key_finish(c, sgfhe_bkey_import_device_form_impl(c, src));
}

int32_t sgfhe_bootstrap_batch_device(sgfhe_ctx *c, const uint64_t *a1, const uint64_t *b1,
                                     const uint64_t *a2, const uint64_t *b2, size_t batch,
                                     uint64_t *out, uint32_t flags, void *stream) {
    if (!c || !a1 || !b1 || !a2 || !b2 || !out) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    if (batch == 0) return SGFHE_OK;
    (void)hipSetDevice(c->device);
    // The call's work is queued on the caller's stream (or the ctx's own), behind everything earlier
    // calls queued on this ctx on any stream (fence_begin): calls on one ctx never overlap on the
    // device, so two threads or two streams may share it.
    return bootstrap_device(c, a1, b1, a2, b2, batch, out, flags, c->n, nullptr,
                            stream ? (hipStream_t)stream : c->stream);
}

int32_t sgfhe_sync(sgfhe_ctx *c) {
    if (!c) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    (void)hipSetDevice(c->device);
    // all work queued on the ctx, whichever streams the calls named (every call ends in ev_done)
    int32_t rc = drain(c);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SGFHE_OK;
}

static int32_t bootstrap_host(sgfhe_ctx *c, const uint64_t *a1, const uint64_t *b1,
                              const uint64_t *a2, const uint64_t *b2, size_t batch, uint64_t *out,
                              uint32_t flags, uint64_t n_iters, uint64_t *acc, uint64_t *digs = nullptr) {
    (void)hipSetDevice(c->device);
    const size_t n = c->n;
    const size_t out_row = 3 * (n + 1) * ((flags & SGFHE_FLAG_RAW_MODQ) ? 2 : 1);
    const size_t out_words = batch * out_row;
    const size_t acc_words = batch * 2 * (size_t)c->M * 2;
    uint64_t *d_in = nullptr, *d_out = nullptr, *d_dig = nullptr;
    ulonglong2 *d_acc = nullptr;
    const size_t dig_words = batch * 4 * (size_t)c->M;
    const size_t in_words = 2 * batch * (n + 1);
    int32_t rc = drain(c);   // the staging buffers may be regrown: nothing of an earlier call may be in flight
    if (rc) return rc;
    if (in_words > c->io_in_words) {
        if (c->io_in) (void)hipFree(c->io_in);
        c->io_in = nullptr;
        c->io_in_words = 0;
        HIPCHK(c, hipMalloc(&c->io_in, in_words * 8));
        c->io_in_words = in_words;
    }
    if (out && out_words > c->io_out_words) {
        if (c->io_out) (void)hipFree(c->io_out);
        c->io_out = nullptr;
        c->io_out_words = 0;
        HIPCHK(c, hipMalloc(&c->io_out, out_words * 8));
        c->io_out_words = out_words;
    }
    d_in = c->io_in;
    if (out) d_out = c->io_out;
    const bool dbg = getenv("SGFHE_DEBUG_IO") != nullptr;
    // The production call (results only): page-locked mirrors of both staging buffers, grown on demand
    // and kept, and the copies pipelined chunk by chunk beside the kernels (HostPipe).  A failed
    // allocation, a buffer above PIN_MAX_BYTES, SGFHE_HOST_PIN=0 and the debug hooks take direct copies
    // of the caller's arrays.
    const size_t pin_max = pin_max_bytes();
    bool pipe = c->use_pin && out && !acc && !digs && in_words * 8 <= pin_max && out_words * 8 <= pin_max;
    if (pipe && in_words > c->pin_in_words) {
        if (c->pin_in) (void)hipHostFree(c->pin_in);
        c->pin_in = nullptr;
        c->pin_in_words = 0;
        const hipError_t pe = hipHostMalloc(&c->pin_in, in_words * 8, hipHostMallocDefault);
        if (pe == hipSuccess) c->pin_in_words = in_words;
        else { (void)hipGetLastError(); c->pin_in = nullptr; pipe = false; }
        if (dbg) fprintf(stderr, "[sgfhe io] pin_in %zu bytes: %s\n", in_words * 8, hipGetErrorString(pe));
    }
    if (pipe && out_words > c->pin_out_words) {
        if (c->pin_out) (void)hipHostFree(c->pin_out);
        c->pin_out = nullptr;
        c->pin_out_words = 0;
        const hipError_t pe = hipHostMalloc(&c->pin_out, out_words * 8, hipHostMallocDefault);
        if (pe == hipSuccess) c->pin_out_words = out_words;
        else { (void)hipGetLastError(); c->pin_out = nullptr; pipe = false; }
        if (dbg) fprintf(stderr, "[sgfhe io] pin_out %zu bytes: %s\n", out_words * 8, hipGetErrorString(pe));
    }
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if (pipe) {
        const HostPipe hp = {a1, b1, a2, b2, out, d_in, d_out, c->pin_in, c->pin_out, out_row};
        rc = bootstrap_device(c, nullptr, nullptr, nullptr, nullptr, batch, d_out, flags, n_iters, nullptr,
                              c->stream, nullptr, &hp);
        const hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess && rc == SGFHE_OK) rc = fail(c, SGFHE_ERR_HIP, hipGetErrorString(e));
        if (rc == SGFHE_OK) c->pending = false;
        if (dbg) fprintf(stderr, "[sgfhe io] batch %zu pipelined: %.2f ms\n", batch, now() - t0);
        return rc;
    }
    hipError_t e = hipSuccess;
    uint64_t *d_a1 = d_in, *d_a2 = d_in + batch * n, *d_b1 = d_in + 2 * batch * n,
             *d_b2 = d_b1 + batch;
    const size_t a_bytes = batch * n * 8;
    double t1 = 0, t2 = 0, t3 = 0;
    do {
        if (acc && (e = hipMalloc(&d_acc, acc_words * 8)) != hipSuccess) break;
        if (digs && (e = hipMalloc(&d_dig, dig_words * 8)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_a1, a1, a_bytes, hipMemcpyHostToDevice, c->stream))) break;
        if ((e = hipMemcpyAsync(d_a2, a2, a_bytes, hipMemcpyHostToDevice, c->stream))) break;
        if ((e = hipMemcpyAsync(d_b1, b1, batch * 8, hipMemcpyHostToDevice, c->stream))) break;
        if ((e = hipMemcpyAsync(d_b2, b2, batch * 8, hipMemcpyHostToDevice, c->stream))) break;
        t1 = now();
        rc = bootstrap_device(c, d_a1, d_b1, d_a2, d_b2, batch, d_out, flags, n_iters, d_acc,
                              c->stream, d_dig);
        if (rc) break;
        t2 = now();
        if (digs && (e = hipMemcpyAsync(digs, d_dig, dig_words * 8, hipMemcpyDeviceToHost, c->stream)))
            break;
        if (out && (e = hipMemcpyAsync(out, d_out, out_words * 8, hipMemcpyDeviceToHost, c->stream)))
            break;
        if (acc && (e = hipMemcpyAsync(acc, d_acc, acc_words * 8, hipMemcpyDeviceToHost, c->stream)))
            break;
        e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) c->pending = false;
        t3 = now();
        if (dbg)
            fprintf(stderr, "[sgfhe io] batch %zu direct copies: copy-in %.2f ms, enqueue %.2f, wait + copy-out %.2f\n",
                    batch, t1 - t0, t2 - t1, t3 - t2);
    } while (0);
    if (e != hipSuccess && rc == SGFHE_OK) rc = fail(c, SGFHE_ERR_HIP, hipGetErrorString(e));
    if (d_acc) (void)hipFree(d_acc);   // debug hooks only
    if (d_dig) (void)hipFree(d_dig);
    return rc;
}

// One request through the coalescer of the ctxs that share this key (csrc/coalescer.h).
static int32_t coalesced_call(sgfhe_ctx *c, const uint64_t *a1, const uint64_t *b1, const uint64_t *a2,
                              const uint64_t *b2, size_t batch, uint64_t *out, uint32_t flags) {
    Coalescer &co = c->shared->co;
    Coalescer::Req me;
    me.owner = c;
    me.a1 = a1; me.b1 = b1; me.a2 = a2; me.b2 = b2;
    me.batch = batch; me.out = out; me.flags = flags;
    if (c->rnd) {               // its draw stream: this ctx's key, and the number this call has on this ctx
        me.rnd = true;
        memcpy(me.key, c->rnd_key.k, sizeof me.key);
        me.call = c->rnd_call++;
        c->last_call = me.call;
    }
    std::vector<Coalescer::Req *> take;
    size_t gates = 0;
    const int lead = co.arrive(me, take, gates);
    if (lead < 0) return fail(c, SGFHE_ERR_OOM, "out of host memory");
    if (lead == 0) {                                // a leader ran it
        if (me.rc) c->err = me.err;
        return me.rc;
    }
    // the combined call, on this ctx (the caller holds its lock)
    int32_t rc;
    const size_t n = c->n;
    const size_t row = 3 * (n + 1) * ((flags & SGFHE_FLAG_RAW_MODQ) ? 2 : 1);
    if (take.size() == 1) {
        if (me.rnd) c->call_fixed = me.call;            // (numbered above)
        rc = bootstrap_host(c, a1, b1, a2, b2, batch, out, flags, c->n, nullptr);
        c->call_fixed = -1;
    } else {
        // (the staging vectors may throw: no exception leaves an entry point of the C ABI)
        auto gathered = [&]() -> int32_t {
            int32_t r = SGFHE_OK;
            if (me.rnd) {   // the draw stream of every row of the combined call, on the device before its first kernel
                (void)hipSetDevice(c->device);
                if ((r = drain(c))) return r;
                if (gates > c->rows_cap) {
                    if (c->d_rows) (void)hipFree(c->d_rows);
                    c->d_rows = nullptr;
                    c->rows_cap = 0;
                    const size_t cap = gates > co.gates_max ? gates : co.gates_max;
                    if (hipMalloc(&c->d_rows, cap * sizeof(RndRow)) != hipSuccess)
                        return fail(c, SGFHE_ERR_OOM, "hipMalloc of the gathered call's draw-stream table failed");
                    c->rows_cap = cap;
                }
                c->h_rows.clear();
                for (const Coalescer::Req *q : take)
                    for (size_t t = 0; t < q->batch; t++) {
                        RndRow row_of;
                        memcpy(row_of.key.k, q->key, sizeof row_of.key.k);
                        row_of.call = q->call;
                        row_of.boot = (uint32_t)t;
                        c->h_rows.push_back(row_of);
                    }
                if (hipMemcpy(c->d_rows, c->h_rows.data(), gates * sizeof(RndRow), hipMemcpyHostToDevice) != hipSuccess)
                    return fail(c, SGFHE_ERR_HIP, "copy of the gathered call's draw-stream table failed");
                c->gather_rows = c->d_rows;
            }
            std::vector<uint64_t> &g = c->co_buf;                          // [a1 | a2 | b1 | b2 | out] of all requests
            g.resize(gates * (2 * n + 2 + row));
            uint64_t *ga1 = g.data(), *ga2 = ga1 + gates * n, *gb1 = ga2 + gates * n, *gb2 = gb1 + gates, *gout = gb2 + gates;
            size_t r0 = 0;
            for (const Coalescer::Req *q : take) {
                memcpy(ga1 + r0 * n, q->a1, q->batch * n * 8);
                memcpy(ga2 + r0 * n, q->a2, q->batch * n * 8);
                memcpy(gb1 + r0, q->b1, q->batch * 8);
                memcpy(gb2 + r0, q->b2, q->batch * 8);
                r0 += q->batch;
            }
            if ((r = bootstrap_host(c, ga1, gb1, ga2, gb2, gates, gout, flags, c->n, nullptr))) return r;
            r0 = 0;
            for (const Coalescer::Req *q : take) {
                memcpy(q->out, gout + r0 * row, q->batch * row * 8);
                r0 += q->batch;
            }
            return SGFHE_OK;
        };
        try {
            rc = gathered();
        } catch (...) {
            rc = fail(c, SGFHE_ERR_OOM, "out of host memory");
        }
        c->gather_rows = nullptr;
    }
    co.finish(take, gates, rc, c->err);
    return rc;
}

int32_t sgfhe_bootstrap_batch(sgfhe_ctx *c, const uint64_t *a1, const uint64_t *b1,
                              const uint64_t *a2, const uint64_t *b2, size_t batch, uint64_t *out,
                              uint32_t flags) {
    if (!c || !a1 || !b1 || !a2 || !b2 || !out) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    if (batch == 0) return SGFHE_OK;
    // small calls on a key that other ctxs share: gathered with whatever the other callers bring (Coalescer)
    if (c->shared.use_count() > 1 && c->have_key && c->shared->co.wants(batch))
        return coalesced_call(c, a1, b1, a2, b2, batch, out, flags);
    return bootstrap_host(c, a1, b1, a2, b2, batch, out, flags, c->n, nullptr);
}

int32_t sgfhe_set_coalesce(sgfhe_ctx *c, int enable, uint32_t req_max, uint32_t gates_max, uint32_t window_us) {
    if (!c) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    if (enable && (req_max == 0 || gates_max < req_max || gates_max > 4096 || window_us > 100000))
        return fail(c, SGFHE_ERR_INVALID_ARG, "set_coalesce: 1 <= req_max <= gates_max <= 4096, window at most 100 ms");
    Coalescer &co = c->shared->co;
    std::lock_guard<std::mutex> lk(co.mu);
    co.enabled = enable != 0;
    if (enable) { co.req_max = req_max; co.gates_max = gates_max; co.window_us = window_us; }
    return SGFHE_OK;
}

int32_t sgfhe_coalesce_stats(sgfhe_ctx *c, uint64_t *stats, int reset) {
    if (!c || !stats) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    Coalescer &co = c->shared->co;
    std::lock_guard<std::mutex> lk(co.mu);
    stats[0] = co.n_calls; stats[1] = co.n_reqs; stats[2] = co.n_gates; stats[3] = co.max_reqs;
    if (reset) co.n_calls = co.n_reqs = co.n_gates = co.max_reqs = 0;
    return SGFHE_OK;
}

int32_t sgfhe_debug_accumulators(sgfhe_ctx *c, const uint64_t *a1, const uint64_t *b1,
                                 const uint64_t *a2, const uint64_t *b2, size_t batch,
                                 uint64_t n_iters, uint64_t *acc) {
    if (!c || !a1 || !b1 || !a2 || !b2 || !acc) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    if (n_iters > c->n) return fail(c, SGFHE_ERR_INVALID_ARG, "n_iters > n");
    if (batch == 0) return SGFHE_OK;
    return bootstrap_host(c, a1, b1, a2, b2, batch, nullptr, 0, n_iters, acc);
}

int32_t sgfhe_debug_digits(sgfhe_ctx *c, const uint64_t *a1, const uint64_t *b1, const uint64_t *a2,
                           const uint64_t *b2, size_t batch, uint64_t n_iters, uint64_t *digits) {
    if (!c || !a1 || !b1 || !a2 || !b2 || !digits) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    if (n_iters > c->n) return fail(c, SGFHE_ERR_INVALID_ARG, "n_iters > n");
    if (batch == 0) return SGFHE_OK;
    return bootstrap_host(c, a1, b1, a2, b2, batch, nullptr, 0, n_iters, nullptr, digits);
}

int32_t sgfhe_debug_flatten(sgfhe_ctx *c, const uint64_t *values, uint64_t *digits) {
    if (!c || !values || !digits) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    (void)hipSetDevice(c->device);
    SGFHE_QUIESCE(c);
    const uint32_t M = c->M;
    int32_t rc = ensure_work(c, 8);
    if (rc) return rc;
    const sgfhe_ctx::Lane &L = c->lane[0];
    for (uint32_t i = 0; i < 2 * M; i++)
        if (ld128(values + 2 * (size_t)i) >= c->Q)
            return fail(c, SGFHE_ERR_INVALID_ARG, "debug_flatten: a value is not in [0, Q)");
    ulonglong2 *d_in = nullptr;
    uint64_t *d_out = nullptr;
    HIPCHK(c, hipMalloc(&d_in, (size_t)2 * M * 16));
    hipError_t e = hipMalloc(&d_out, (size_t)4 * M * 8);
    if (e != hipSuccess) { (void)hipFree(d_in); return fail(c, SGFHE_ERR_HIP, hipGetErrorString(e)); }
    do {
        if ((e = hipMemcpyAsync(d_in, values, (size_t)2 * M * 16, hipMemcpyHostToDevice, c->stream))) break;
        hipLaunchKernelGGL(k_flatten_canon, dim3((2 * M + 255) / 256), dim3(256), 0, c->stream, d_in,
                           L.dig, c->d_crt, 2 * M, (uint32_t)c->logm);
        hipLaunchKernelGGL(k_dump_digits, dim3((2 * M + 255) / 256), dim3(256), 0, c->stream, L.dig,
                           d_out, 2 * M, (uint32_t)c->logm, 0u);
        if ((e = hipGetLastError())) break;
        if ((e = hipMemcpyAsync(digits, d_out, (size_t)4 * M * 8, hipMemcpyDeviceToHost, c->stream))) break;
        e = hipStreamSynchronize(c->stream);
    } while (0);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, SGFHE_ERR_HIP, hipGetErrorString(e));
    return SGFHE_OK;
}

int32_t sgfhe_external_product(sgfhe_ctx *c, const uint64_t *a, const uint64_t *b,
                               const uint64_t *A, uint64_t *a_res, uint64_t *b_res) {
    if (!c || !a || !b || !A || !a_res || !b_res) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    (void)hipSetDevice(c->device);
    SGFHE_QUIESCE(c);
    const uint32_t M = c->M;
    const uint32_t cpad = 8;
    int32_t rc = ensure_work(c, cpad);
    if (rc) return rc;
    const sgfhe_ctx::Lane &L = c->lane[0];
    int32_t *d_A = nullptr;
    ulonglong2 *d_ab = nullptr;
    HIPCHK(c, hipMalloc(&d_A, (size_t)c->npr * 8 * M * 4));
    hipError_t e = hipMalloc(&d_ab, (size_t)2 * M * 16);
    if (e != hipSuccess) { (void)hipFree(d_A); return fail(c, SGFHE_ERR_HIP, hipGetErrorString(e)); }
    do {
        // A[4][2][m] is exactly one key slice (k = 0)
        rc = key_transform_host(c, A, 8, d_A);
        if (rc) break;
        if ((e = hipMemsetAsync(L.dig, 0, (size_t)cpad * 2 * M * 16, c->stream))) break;
        if ((e = hipMemcpyAsync(d_ab, a, (size_t)M * 16, hipMemcpyHostToDevice, c->stream))) break;
        if ((e = hipMemcpyAsync(d_ab + M, b, (size_t)M * 16, hipMemcpyHostToDevice, c->stream))) break;
        hipLaunchKernelGGL(k_flatten_canon, dim3((2 * M + 255) / 256), dim3(256), 0, c->stream, d_ab,
                           L.dig, c->d_crt, 2 * M, (uint32_t)c->logm);
        rc = launch_extprod(c, L, d_A, cpad, 0, MODE_PLAIN, c->stream);
        if (rc) break;
        rc = launch_crt(c, L, cpad, MODE_NOACC | MODE_CANON, c->stream);
        if (rc) break;
        if ((e = hipMemcpyAsync(a_res, L.dig, (size_t)M * 16, hipMemcpyDeviceToHost, c->stream))) break;
        if ((e = hipMemcpyAsync(b_res, L.dig + 2 * (size_t)M, (size_t)M * 16, hipMemcpyDeviceToHost, c->stream))) break;
        e = hipStreamSynchronize(c->stream);
    } while (0);
    if (e != hipSuccess && rc == SGFHE_OK) rc = fail(c, SGFHE_ERR_HIP, hipGetErrorString(e));
    (void)hipFree(d_A);
    (void)hipFree(d_ab);
    return rc;
}

int32_t sgfhe_debug_cmux(sgfhe_ctx *c, const uint64_t *a, const uint64_t *b, const uint64_t *C,
                         uint64_t j, uint64_t *a_res, uint64_t *b_res) {
    if (!c || !a || !b || !C || !a_res || !b_res) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    if (j >= 2 * (uint64_t)c->M) return fail(c, SGFHE_ERR_INVALID_ARG, "debug_cmux: j must be in [0, 2 m)");
    (void)hipSetDevice(c->device);
    SGFHE_QUIESCE(c);
    const uint32_t M = c->M;
    const uint32_t cpad = 8;
    int32_t rc = ensure_work(c, cpad);
    if (rc) return rc;
    const sgfhe_ctx::Lane &L = c->lane[0];
    int32_t *d_C = nullptr;
    ulonglong2 *d_ab = nullptr;
    HIPCHK(c, hipMalloc(&d_C, (size_t)c->npr * 8 * M * 4));
    hipError_t e = hipMalloc(&d_ab, (size_t)2 * M * 16);
    if (e != hipSuccess) { (void)hipFree(d_C); return fail(c, SGFHE_ERR_HIP, hipGetErrorString(e)); }
    const uint32_t jw = (uint32_t)j;
    do {
        rc = key_transform_host(c, C, 8, d_C);  // C[4][2][m] is one key slice
        if (rc) break;
        if ((e = hipMemsetAsync(L.dig, 0, (size_t)cpad * 2 * M * 16, c->stream))) break;
        if ((e = hipMemsetAsync(L.ua, 0, (size_t)cpad * c->n * 4, c->stream))) break;
        if ((e = hipMemcpyAsync(L.ua, &jw, 4, hipMemcpyHostToDevice, c->stream))) break;  // u.a[0] of bootstrap 0
        if ((e = hipMemcpyAsync(d_ab, a, (size_t)M * 16, hipMemcpyHostToDevice, c->stream))) break;
        if ((e = hipMemcpyAsync(d_ab + M, b, (size_t)M * 16, hipMemcpyHostToDevice, c->stream))) break;
        hipLaunchKernelGGL(k_flatten_canon, dim3((2 * M + 255) / 256), dim3(256), 0, c->stream, d_ab,
                           L.dig, c->d_crt, 2 * M, (uint32_t)c->logm);
        rc = launch_extprod(c, L, d_C, cpad, 0, 0u, c->stream);     // iteration k = 0, with the rotation
        if (rc) break;
        rc = launch_crt(c, L, cpad, 0u, c->stream);                  // acc <- acc + D, flattened again
        if (rc) break;
        hipLaunchKernelGGL(k_dump_acc, dim3((2 * M + 255) / 256), dim3(256), 0, c->stream, L.dig, d_ab,
                           c->d_crt, 2 * M, (uint32_t)c->logm, 0u);
        if ((e = hipGetLastError())) break;
        if ((e = hipMemcpyAsync(a_res, d_ab, (size_t)M * 16, hipMemcpyDeviceToHost, c->stream))) break;
        if ((e = hipMemcpyAsync(b_res, d_ab + M, (size_t)M * 16, hipMemcpyDeviceToHost, c->stream))) break;
        e = hipStreamSynchronize(c->stream);
    } while (0);
    if (e != hipSuccess && rc == SGFHE_OK) rc = fail(c, SGFHE_ERR_HIP, hipGetErrorString(e));
    (void)hipFree(d_C);
    (void)hipFree(d_ab);
    return rc;
}

int32_t sgfhe_pack_encrypted_bits(sgfhe_ctx *c, const uint64_t *a, const uint64_t *b, size_t count,
                                  uint64_t *out_w, uint64_t *out_v) {
    if (!c || !a || !b || !out_w || !out_v) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    if (count == 0) return SGFHE_OK;
    if (!c->have_key) return fail(c, SGFHE_ERR_NO_KEY, "no bootstrap key uploaded");
    (void)hipSetDevice(c->device);
    SGFHE_QUIESCE(c);
    const size_t n = c->n, M = c->M;
    const size_t nb = count * n;  // bootstraps
    // rng != nothing: the n bootstraps and the flatten of every as_i (all m coefficients of the
    // resized polynomial, utils.jl:253-264) draw from the ctx's ChaCha stream (fhe.jl:673,683-684)
    const uint32_t mode = c->rnd ? MODE_RANDOM : 0u;
    const uint32_t G = c->rnd ? c->pack_G_rnd : c->pack_G;
    if (!G) return fail(c, SGFHE_ERR_UNSUPPORTED, "pack_encrypted_bits: exactness bound of the RNS primes");
    const uint32_t groups = (uint32_t)(n / G);
    const size_t len = c->rnd ? M : n;  // stored coefficients per digit polynomial
    uint64_t *d_lwe = nullptr, *d_pdig = nullptr, *d_wv = nullptr;
    ulonglong2 *d_raw = nullptr;
    uint32_t *d_yg = nullptr;
    hipError_t e = hipSuccess;
    int32_t rc = SGFHE_OK;
    do {
        // [a1 = 0 | a2 | b1 = Dr | b2]: trivial encryption of 1 paired with every bit (fhe.jl:669-673)
        if ((e = hipMalloc(&d_lwe, (2 * nb * n + 2 * nb) * 8))) break;
        uint64_t *d_a1 = d_lwe, *d_a2 = d_lwe + nb * n, *d_b1 = d_a2 + nb * n, *d_b2 = d_b1 + nb;
        if ((e = hipMalloc(&d_raw, nb * 3 * (n + 1) * 16))) break;
        if ((e = hipMalloc(&d_pdig, count * n * 2 * len * 8))) break;
        if ((e = hipMalloc(&d_yg, count * groups * 2 * c->npr * M * 4))) break;
        if ((e = hipMalloc(&d_wv, 2 * count * M * 8))) break;
        if ((e = hipMemsetAsync(d_a1, 0, nb * n * 8, c->stream))) break;
        std::vector<uint64_t> ones(nb, c->par.r / 4);
        if ((e = hipMemcpyAsync(d_b1, ones.data(), nb * 8, hipMemcpyHostToDevice, c->stream))) break;
        if ((e = hipMemcpyAsync(d_a2, a, nb * n * 8, hipMemcpyHostToDevice, c->stream))) break;
        if ((e = hipMemcpyAsync(d_b2, b, nb * 8, hipMemcpyHostToDevice, c->stream))) break;
        rc = bootstrap_device(c, d_a1, d_b1, d_a2, d_b2, nb, (uint64_t *)d_raw, SGFHE_FLAG_RAW_MODQ,
                              c->n, nullptr, c->stream);
        if (rc) break;
        const RndArgs ra = {c->rnd_key, c->last_call, 0u, nullptr};
        const size_t tf = count * n * len;
        hipLaunchKernelGGL(k_pack_flatten, dim3((unsigned)((tf + 255) / 256)), dim3(256), 0, c->stream,
                           d_raw, d_pdig, c->d_crt, (uint32_t)count, (uint32_t)n, (uint32_t)c->logm,
                           mode, ra);
        rc = launch_shortprod(c, d_pdig, d_yg, (uint32_t)count, G, groups, mode, c->stream);
        if (rc) break;
        const size_t tw = count * M;
        switch (c->npr) {
#define X(NP)                                                                                     \
    case NP:                                                                                      \
        hipLaunchKernelGGL(k_pack_finish<NP>, dim3((unsigned)((tw + 255) / 256)), dim3(256), 0,   \
                           c->stream, d_yg, d_raw, d_wv, d_wv + count * M, c->d_crt,              \
                           (uint32_t)count, (uint32_t)n, (uint32_t)c->logm, groups);              \
        break;
            SGFHE_FOR_NPR(X)
#undef X
        }
        if ((e = hipGetLastError())) break;
        if ((e = hipMemcpyAsync(out_w, d_wv, count * M * 8, hipMemcpyDeviceToHost, c->stream))) break;
        if ((e = hipMemcpyAsync(out_v, d_wv + count * M, count * M * 8, hipMemcpyDeviceToHost, c->stream))) break;
        e = hipStreamSynchronize(c->stream);
    } while (0);
    if (e != hipSuccess && rc == SGFHE_OK) rc = fail(c, SGFHE_ERR_HIP, hipGetErrorString(e));
    if (d_lwe) (void)hipFree(d_lwe);
    if (d_raw) (void)hipFree(d_raw);
    if (d_pdig) (void)hipFree(d_pdig);
    if (d_yg) (void)hipFree(d_yg);
    if (d_wv) (void)hipFree(d_wv);
    return rc;
}

int32_t sgfhe_debug_ntt(sgfhe_ctx *c, uint32_t prime_index, int inverse, const uint32_t *in,
                        uint32_t *out) {
    if (!c || !in || !out || prime_index >= c->npr) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    (void)hipSetDevice(c->device);
    SGFHE_QUIESCE(c);
    uint32_t *d = nullptr;
    HIPCHK(c, hipMalloc(&d, (size_t)2 * c->M * 4));
    int32_t rc = SGFHE_OK;
    hipError_t e;
    do {
        if ((e = hipMemcpyAsync(d, in, (size_t)c->M * 4, hipMemcpyHostToDevice, c->stream))) break;
        rc = launch_dbgntt(c, d, d + c->M, prime_index, inverse, c->stream);
        if (rc) break;
        if ((e = hipMemcpyAsync(out, d + c->M, (size_t)c->M * 4, hipMemcpyDeviceToHost, c->stream))) break;
        e = hipStreamSynchronize(c->stream);
    } while (0);
    if (e != hipSuccess && rc == SGFHE_OK) rc = fail(c, SGFHE_ERR_HIP, hipGetErrorString(e));
    (void)hipFree(d);
    return rc;
}

int32_t sgfhe_debug_primes(const sgfhe_ctx *c, uint32_t *count, uint32_t *primes) {
    if (!c || !count || !primes) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    *count = c->npr;
    for (uint32_t i = 0; i < c->npr; i++) primes[i] = c->primes[i];
    return SGFHE_OK;
}

// ---- N3: the ciphertext plumbing either side of the path, on the host (no device, no ctx) ----------

namespace {
// r = 2 m a power of two, n <= m, t = log2(r) - 1, Dr = r / 4
bool host_params_ok(const sgfhe_params *p) {
    return p && p->n >= 1 && p->m >= p->n && p->r == 2 * p->m && (p->r & (p->r - 1)) == 0 && p->r >= 32 &&
           p->r <= (1ull << 32);
}
unsigned host_t(const sgfhe_params *p) {
    unsigned lg = 0;
    while ((1ull << lg) < p->r) lg++;
    return lg - 1;
}
}  // namespace

int32_t sgfhe_host_deterministic_expand(const sgfhe_params *p, const uint8_t *u, uint64_t *a) {
    if (!host_params_ok(p) || !u || !a) return SGFHE_ERR_INVALID_ARG;
    sgfhe_host::prng_expand(u, p->n, host_t(p) + 1, a);
    for (uint64_t i = 0; i < p->n; i++) a[i] &= p->r - 1;
    return SGFHE_OK;
}

int32_t sgfhe_host_encrypt_private(const sgfhe_params *p, const uint64_t *sk, const uint8_t *u,
                                   const int64_t *w, const uint8_t *message, uint64_t *a, uint64_t *b) {
    if (!host_params_ok(p) || !sk || !u || !w || !message || !a || !b) return SGFHE_ERR_INVALID_ARG;
    const uint64_t n = p->n, rmask = p->r - 1, Dr = p->r / 4;
    const int64_t wr = (int64_t)(Dr / 8);
    for (uint64_t i = 0; i < n; i++)
        if (w[i] < -wr || w[i] > wr) return SGFHE_ERR_INVALID_ARG;      // w in -Dr/8 .. Dr/8 (fhe.jl:318-319)
    int32_t rc = sgfhe_host_deterministic_expand(p, u, a);             // fhe.jl:316
    if (rc) return rc;
    std::vector<uint64_t> key(n), as(n);
    for (uint64_t i = 0; i < n; i++) key[i] = sk[i] & 1;
    sgfhe_host::negacyclic_mul(a, key.data(), n, rmask, as.data());
    const unsigned sh = host_t(p) - 4;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t v = (as[i] + (uint64_t)w[i] + (uint64_t)(message[i] & 1) * Dr) & rmask;   // fhe.jl:322
        b[i] = (v >> sh) << sh;                                          // the highmost 5 bits (fhe.jl:325)
    }
    return SGFHE_OK;
}

int32_t sgfhe_host_pack_private(const sgfhe_params *p, const uint64_t *b, uint8_t *v) {
    if (!host_params_ok(p) || !b || !v) return SGFHE_ERR_INVALID_ARG;
    std::vector<uint64_t> packed(p->n);
    const unsigned sh = host_t(p) - 4;
    for (uint64_t i = 0; i < p->n; i++) packed[i] = b[i] >> sh;        // fhe.jl:342
    sgfhe_host::unpackbits(packed.data(), p->n, 5, v);
    return SGFHE_OK;
}

int32_t sgfhe_host_normalize_private(const sgfhe_params *p, const uint8_t *u, const uint8_t *v,
                                     uint64_t *a, uint64_t *b) {
    if (!host_params_ok(p) || !u || !v || !a || !b) return SGFHE_ERR_INVALID_ARG;
    int32_t rc = sgfhe_host_deterministic_expand(p, u, a);             // fhe.jl:356
    if (rc) return rc;
    sgfhe_host::packbits(v, 5, p->n, b);                               // fhe.jl:357
    const unsigned sh = host_t(p) - 4;
    for (uint64_t i = 0; i < p->n; i++) b[i] = (b[i] << sh) & (p->r - 1);
    return SGFHE_OK;
}

int32_t sgfhe_host_split_ciphertext(const sgfhe_params *p, const uint64_t *a, const uint64_t *b,
                                    size_t N, uint64_t *lwe_a, uint64_t *lwe_b) {
    if (!host_params_ok(p) || !a || !b || !lwe_a || !lwe_b || (N != p->n && N != p->m))
        return SGFHE_ERR_INVALID_ARG;
    const uint64_t n = p->n, rmask = p->r - 1;
    for (uint64_t i = 1; i <= n; i++) {                                 // fhe.jl:288-289
        sgfhe_host::extract(a, N, i, n, rmask, lwe_a + (i - 1) * n);
        lwe_b[i - 1] = b[i - 1] & rmask;
    }
    return SGFHE_OK;
}

int32_t sgfhe_host_decrypt_lwe(const sgfhe_params *p, const uint64_t *sk, const uint64_t *lwe_a,
                               const uint64_t *lwe_b, size_t count, uint8_t *bits) {
    if (!host_params_ok(p) || !sk || !lwe_a || !lwe_b || !bits) return SGFHE_ERR_INVALID_ARG;
    const uint64_t n = p->n, rmask = p->r - 1, Dr = p->r / 4;
    for (size_t t = 0; t < count; t++) {
        uint64_t dot = 0;
        for (uint64_t i = 0; i < n; i++) dot += lwe_a[t * n + i] * (sk[i] & 1);
        const uint64_t b1 = (lwe_b[t] - dot) & rmask;                    // fhe.jl:505
        bits[t] = (uint8_t)((((b1 + Dr / 2) & rmask) / Dr) & 1);         // fhe.jl:506
    }
    return SGFHE_OK;
}

int32_t sgfhe_host_decrypt_rlwe(const sgfhe_params *p, const uint64_t *sk, const uint64_t *a,
                                const uint64_t *b, size_t N, uint8_t *bits) {
    if (!host_params_ok(p) || !sk || !a || !b || !bits || (N != p->n && N != p->m))
        return SGFHE_ERR_INVALID_ARG;
    const uint64_t n = p->n, rmask = p->r - 1, Dr = p->r / 4;
    std::vector<uint64_t> key(N, 0), as(N);                              // resize(key, m) for a Ciphertext (fhe.jl:474-478)
    for (uint64_t i = 0; i < n; i++) key[i] = sk[i] & 1;
    sgfhe_host::negacyclic_mul(a, key.data(), N, rmask, as.data());
    for (uint64_t i = 0; i < n; i++) {                                   // the first n coefficients (fhe.jl:481-482)
        const uint64_t b1 = (b[i] - as[i]) & rmask;
        bits[i] = (uint8_t)((((b1 + Dr / 2) & rmask) / Dr) & 1);         // fhe.jl:491-493
    }
    return SGFHE_OK;
}

// ---- the public-key side (row N4) ------------------------------------------------------------------
static bool host_q_ok(const sgfhe_params *p, uint64_t q) {
    // q = find_modulus(2 n, r n) (fhe.jl:57): odd, 2 n | q - 1, above r n, and small enough for the
    // 64-bit products of rescale (q r < 2^64) and of the short convolution (n q < 2^62)
    return host_params_ok(p) && q > p->r * p->n && q < (1ull << 31) && (q - 1) % (2 * p->n) == 0;
}
static int64_t host_e_max(const sgfhe_params *p, uint64_t q) {
    const uint64_t Dq = q / 4, d = 41 * p->n;                            // fhe.jl:159-160
    return (int64_t)(Dq / d) - (Dq % d == 0 ? 1 : 0);
}

int32_t sgfhe_host_public_key(const sgfhe_params *p, uint64_t q, const uint64_t *sk, const uint64_t *k0,
                              const int64_t *e, uint64_t *k1) {
    if (!sk || !k0 || !e || !k1 || !p || !host_q_ok(p, q)) return SGFHE_ERR_INVALID_ARG;
    const size_t n = p->n;
    const int64_t e_max = host_e_max(p, q);
    std::vector<int8_t> s(n);
    for (size_t i = 0; i < n; i++) {
        if (k0[i] >= q || e[i] < -e_max || e[i] > e_max) return SGFHE_ERR_INVALID_ARG;
        s[i] = (int8_t)(sk[i] & 1);
    }
    std::vector<int64_t> ks(n);
    sgfhe_host::negacyclic_mul_short(k0, s.data(), n, q, ks.data());     // fhe.jl:163-164
    for (size_t i = 0; i < n; i++) {
        const int64_t v = (ks[i] + e[i]) % (int64_t)q;
        k1[i] = (uint64_t)(v < 0 ? v + (int64_t)q : v);
    }
    return SGFHE_OK;
}

int32_t sgfhe_host_encrypt_public(const sgfhe_params *p, uint64_t q, const uint64_t *k0, const uint64_t *k1,
                                  const int8_t *u, const int64_t *w1, const int64_t *w2,
                                  const uint8_t *message, uint64_t *a, uint64_t *b) {
    if (!k0 || !k1 || !u || !w1 || !w2 || !message || !a || !b || !p || !host_q_ok(p, q))
        return SGFHE_ERR_INVALID_ARG;
    const size_t n = p->n;
    const uint64_t r = p->r, Dq = q / 4;
    const int t = (int)host_t(p);                                        // r = 2^(t + 1)
    if (t < 5) return SGFHE_ERR_INVALID_ARG;                             // fhe.jl:404
    const int shift = t - 5;
    const int64_t w1_max = (int64_t)(Dq / (41 * n)), w2_max = (int64_t)(Dq / 82);   // fhe.jl:392,395
    for (size_t i = 0; i < n; i++)
        if (k0[i] >= q || k1[i] >= q || u[i] < -1 || u[i] > 1 || w1[i] < -w1_max || w1[i] > w1_max ||
            w2[i] < -w2_max || w2[i] > w2_max)
            return SGFHE_ERR_INVALID_ARG;
    std::vector<int64_t> k0u(n), k1u(n);
    sgfhe_host::negacyclic_mul_short(k0, u, n, q, k0u.data());           // fhe.jl:399
    sgfhe_host::negacyclic_mul_short(k1, u, n, q, k1u.data());           // fhe.jl:400
    for (size_t i = 0; i < n; i++) {
        int64_t a1 = (k0u[i] + w1[i]) % (int64_t)q;
        if (a1 < 0) a1 += (int64_t)q;
        int64_t a2 = (k1u[i] + w2[i] + (int64_t)((message[i] & 1) * Dq)) % (int64_t)q;
        if (a2 < 0) a2 += (int64_t)q;
        a[i] = sgfhe_host::rescale(r, (uint64_t)a1, q, true);            // fhe.jl:402
        b[i] = (sgfhe_host::rescale(r >> shift, (uint64_t)a2, q, false) << shift) & (r - 1);   // :405-406
    }
    return SGFHE_OK;
}

int32_t sgfhe_host_pack_public(const sgfhe_params *p, const uint64_t *a, const uint64_t *b,
                               uint8_t *a_bits, uint8_t *b_bits) {
    if (!host_params_ok(p) || !a || !b || !a_bits || !b_bits) return SGFHE_ERR_INVALID_ARG;
    const size_t n = p->n;
    const int t = (int)host_t(p);
    if (t < 5) return SGFHE_ERR_INVALID_ARG;
    sgfhe_host::unpackbits(a, n, (size_t)t + 1, a_bits);                 // fhe.jl:429
    std::vector<uint64_t> bp(n);
    for (size_t i = 0; i < n; i++) bp[i] = (b[i] & (p->r - 1)) >> (t - 5);   // fhe.jl:431
    sgfhe_host::unpackbits(bp.data(), n, 6, b_bits);                     // fhe.jl:432
    return SGFHE_OK;
}

int32_t sgfhe_host_normalize_public(const sgfhe_params *p, const uint8_t *a_bits, const uint8_t *b_bits,
                                    uint64_t *a, uint64_t *b) {
    if (!host_params_ok(p) || !a_bits || !b_bits || !a || !b) return SGFHE_ERR_INVALID_ARG;
    const size_t n = p->n;
    const int t = (int)host_t(p);
    if (t < 5) return SGFHE_ERR_INVALID_ARG;
    sgfhe_host::packbits(a_bits, (size_t)t + 1, n, a);                   // fhe.jl:446
    sgfhe_host::packbits(b_bits, 6, n, b);                               // fhe.jl:447
    for (size_t i = 0; i < n; i++) {
        a[i] &= p->r - 1;
        b[i] = (b[i] << (t - 5)) & (p->r - 1);
    }
    return SGFHE_OK;
}

int32_t sgfhe_kernel_names(const sgfhe_ctx *c, char *extprod, size_t extprod_cap, char *crt, size_t crt_cap) {
    if (!c || !extprod || !crt || !extprod_cap || !crt_cap) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    const bool wide = c->rnd && (c->B >> 46);
    const int le = c->logm <= SGFHE_EXT_LE3_MAX ? 3 : LOGE;
    snprintf(extprod, extprod_cap, "k_extprod<%d, %d, %s>", c->logm, le, wide ? "true" : "false");
    // the selection of launch_crt_raw for the k-loop's mode
    if (!c->rnd && c->h_lean.nl && c->use_lean)
        snprintf(crt, crt_cap, "k_crt_lean<%u, %u>", c->npr, c->h_lean.nl);
    else if (c->rnd && c->lean_rnd_ok && c->use_lean)
        snprintf(crt, crt_cap, "k_crt_lean_rnd<%u, %u, %s>", c->npr, c->h_lean.nl, wide ? "true" : "false");
    else if (!c->rnd)
        snprintf(crt, crt_cap, "k_crt_acc2<%u>", c->npr);
    else
        snprintf(crt, crt_cap, "k_crt_acc<%u>", c->npr);
    return SGFHE_OK;
}

int32_t sgfhe_release_host_staging(sgfhe_ctx *c) {
    if (!c) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    (void)hipSetDevice(c->device);
    SGFHE_QUIESCE(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->io_in) (void)hipFree(c->io_in);
    if (c->io_out) (void)hipFree(c->io_out);
    if (c->pin_in) (void)hipHostFree(c->pin_in);
    if (c->pin_out) (void)hipHostFree(c->pin_out);
    c->io_in = c->io_out = c->pin_in = c->pin_out = nullptr;
    c->io_in_words = c->io_out_words = c->pin_in_words = c->pin_out_words = 0;
    return SGFHE_OK;
}

int32_t sgfhe_timing_enable(sgfhe_ctx *c, int enable) {
    if (!c) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    c->timing = enable != 0;
    return SGFHE_OK;
}

int32_t sgfhe_timing_read(sgfhe_ctx *c, double *stats, int reset) {
    if (!c || !stats) return SGFHE_ERR_INVALID_ARG;
    SGFHE_LOCK(c);
    (void)hipSetDevice(c->device);
    int32_t rcd = drain(c);
    if (rcd) return rcd;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    timing_flush(c);
    stats[0] = c->n_ext ? c->t_ext / (double)c->n_ext : 0.0;
    stats[1] = (double)c->n_ext;
    stats[2] = c->n_crt ? c->t_crt / (double)c->n_crt : 0.0;
    stats[3] = (double)c->n_crt;
    stats[4] = (double)c->last_chunk;
    stats[5] = c->n_call ? c->t_call / (double)c->n_call : 0.0;
    stats[6] = (double)c->n_call;
    stats[7] = c->n_call ? (double)c->boots_call / (double)c->n_call : 0.0;
    if (reset) {
        c->t_ext = c->t_crt = c->t_call = 0;
        c->n_ext = c->n_crt = c->n_call = c->boots_call = 0;
    }
    return SGFHE_OK;
}

}  // extern "C"
