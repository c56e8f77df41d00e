// host_keys.cpp -- parameters, PRNG, key generation, encrypt/decrypt on the host.
//
// Follows tfhe's new_default_gate_bootstrapping_parameters /
// new_random_gate_bootstrapping_secret_keyset / bootsSymEncrypt /
// bootsSymDecrypt as described in SURVEY.md Appendix A.1-A.2 (the library
// itself is not in /root/reference; the reference calls these at
// src/main.cpp:21-22,63-69,78-83).  Randomness comes from this repo's own
// seeded generator; the draw order is specified in DESIGN.md and is what the
// test oracle regenerates independently.
#include "host_keys.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <pthread.h>
#include <sys/random.h>
#include <unistd.h>

#include <atomic>

namespace tfhe_hip {

bool os_random(void *out, size_t bytes) {
    unsigned char *p = static_cast<unsigned char *>(out);
    size_t got = 0;
    while (got < bytes) {
        const ssize_t r = getrandom(p + got, bytes - got, 0);
        if (r <= 0) break;
        got += (size_t)r;
    }
    if (got == bytes) return true;
    if (FILE *f = std::fopen("/dev/urandom", "rb")) {
        got = std::fread(p, 1, bytes, f);
        std::fclose(f);
    }
    return got == bytes;
}

uint32_t Params::decomp_offset() const {
    uint32_t off = 0;
    const uint32_t half = 1u << (Bgbit - 1);
    for (int j = 1; j <= l; ++j) off += half << (32 - j * Bgbit);
    return off;
}

bool default_params(int32_t lambda, Params &o) {
    if (lambda <= 0 || lambda > 128) return false;
    o.N = 1024; o.k = 1; o.ks_t = 8; o.ks_basebit = 2; o.max_stdev = 0.012467;
    if (lambda > 80) {
        o.n = 630; o.l = 3; o.Bgbit = 7;
        o.ks_stdev = std::ldexp(1.0, -15);
        o.bk_stdev = std::ldexp(1.0, -25);
    } else {
        o.n = 500; o.l = 2; o.Bgbit = 10;
        o.ks_stdev = 2.44e-5; o.bk_stdev = 7.18e-9;
    }
    return true;
}

Params p2048_params() {
    Params o;
    o.N = 2048; o.k = 1; o.n = 1024; o.l = 3; o.Bgbit = 6; o.ks_t = 8; o.ks_basebit = 2;
    o.ks_stdev = std::ldexp(1.0, -15);
    o.bk_stdev = std::ldexp(1.0, -25);
    o.max_stdev = 0.012467;
    return o;
}

ParamBundle *make_param_bundle(const Params &p) {
    ParamBundle *b = new ParamBundle();
    b->p = p;
    b->lwe = LweParams{p.n, p.ks_stdev, p.max_stdev};
    b->tlwe = TLweParams{p.N, p.k, p.bk_stdev, p.max_stdev};
    b->tgsw.l = p.l; b->tgsw.Bgbit = p.Bgbit; b->tgsw.Bg = 1 << p.Bgbit; b->tgsw.halfBg = 1 << (p.Bgbit - 1);
    b->tgsw.maskMod = (1u << p.Bgbit) - 1u;
    b->tgsw.tlwe_params = &b->tlwe;
    b->tgsw.kpl = p.kpl();
    b->tgsw.offset = p.decomp_offset();
    b->set.ks_t = p.ks_t; b->set.ks_basebit = p.ks_basebit;
    b->set.in_out_params = &b->lwe;
    b->set.tgsw_params = &b->tgsw;
    return b;
}

const Params &params_of(const TFheGateBootstrappingParameterSet *set) {
    // `set` is the first member of the bundle it was allocated in
    return reinterpret_cast<const ParamBundle *>(set)->p;
}

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void Rng::reseed(uint64_t seed) {
    chacha_ = false;
    uint64_t z = seed;
    for (auto &word : s_) {
        z += 0x9E3779B97F4A7C15ULL;
        uint64_t t = z;
        t = (t ^ (t >> 30)) * 0xBF58476D1CE4E5B9ULL;
        t = (t ^ (t >> 27)) * 0x94D049BB133111EBULL;
        word = t ^ (t >> 31);
    }
}

// Fork safety of the secure streams: a child of fork() inherits the ChaCha20 key, nonce, counter and the
// unread words of the current block -- parent and child would then draw identical noise and masks, and two encryptions
// under one mask and noise leak the difference of their messages.  A pthread_atfork child handler bumps a generation
// counter; a secure generator that sees another generation than the one it was keyed in re-keys itself from the OS
// before it yields another word.
static std::atomic<uint32_t> g_fork_generation{0};
static void note_fork_in_child() { g_fork_generation.fetch_add(1, std::memory_order_relaxed); }
static void register_fork_handler() {
    static const bool once = [] { return pthread_atfork(nullptr, nullptr, note_fork_in_child) == 0; }();
    (void)once;
}

void Rng::rekey_secure() {
    uint32_t seed[10];
    if (!os_random(seed, sizeof seed)) {
        // a keyset or an encryption without entropy would be worthless: stop, like upstream's fatal paths
        std::fprintf(stderr, "libtfhe-hip: fatal: the operating system offers no entropy (getrandom, /dev/urandom)\n");
        std::abort();
    }
    std::memcpy(key_, seed, sizeof key_);
    nonce_[0] = seed[8]; nonce_[1] = seed[9];
    counter_ = 0;
    pos_ = 16;
    chacha_ = true;
    fork_gen_ = g_fork_generation.load(std::memory_order_relaxed);
}

Rng Rng::secure() {
    register_fork_handler();
    Rng r(0);
    r.rekey_secure();
    return r;
}

static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

// one ChaCha20 block (RFC 8439 section 2.3, with the original 64-bit counter / 64-bit nonce split)
void Rng::refill() {
    uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                       key_[0], key_[1], key_[2], key_[3], key_[4], key_[5], key_[6], key_[7],
                       (uint32_t)counter_, (uint32_t)(counter_ >> 32), nonce_[0], nonce_[1]};
    uint32_t x[16];
    std::memcpy(x, in, sizeof x);
#define CHACHA_QR(a, b, c, d)                      \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);  \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);  \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);   \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7)
    for (int round = 0; round < 10; ++round) {
        CHACHA_QR(0, 4, 8, 12); CHACHA_QR(1, 5, 9, 13); CHACHA_QR(2, 6, 10, 14); CHACHA_QR(3, 7, 11, 15);
        CHACHA_QR(0, 5, 10, 15); CHACHA_QR(1, 6, 11, 12); CHACHA_QR(2, 7, 8, 13); CHACHA_QR(3, 4, 9, 14);
    }
#undef CHACHA_QR
    for (int i = 0; i < 16; ++i) block_[i] = x[i] + in[i];
    ++counter_;
    pos_ = 0;
}

void Rng::chacha_block(const uint32_t key[8], uint64_t counter, const uint32_t nonce[2], uint32_t out[16]) {
    Rng r(0);
    std::memcpy(r.key_, key, sizeof r.key_);
    r.nonce_[0] = nonce[0]; r.nonce_[1] = nonce[1];
    r.counter_ = counter;
    r.chacha_ = true;
    r.refill();
    std::memcpy(out, r.block_, sizeof r.block_);
}

uint64_t Rng::next() {
    if (chacha_) {
        if (fork_gen_ != g_fork_generation.load(std::memory_order_relaxed)) rekey_secure();    // this process is a fork() child
        if (pos_ >= 16) refill();
        const uint64_t v = (uint64_t)block_[pos_] | ((uint64_t)block_[pos_ + 1] << 32);
        pos_ += 2;
        return v;
    }
    const uint64_t result = rotl(s_[1] * 5, 7) * 9;
    const uint64_t t = s_[1] << 17;
    s_[2] ^= s_[0]; s_[3] ^= s_[1]; s_[1] ^= s_[2]; s_[0] ^= s_[3];
    s_[2] ^= t;
    s_[3] = rotl(s_[3], 45);
    return result;
}

double Rng::gauss(double sigma) {
    const double u1 = ((double)(next() >> 11) + 1.0) * (1.0 / 9007199254740992.0);
    const double u2 = (double)(next() >> 11) * (1.0 / 9007199254740992.0);
    return sigma * std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925286766559 * u2);
}

Torus32 dtot32(double d) { return (Torus32)(int64_t)((d - (double)(int64_t)d) * 4294967296.0); }

// body += mask * key  in Z[X]/(X^N+1), key binary
static void add_mul_by_bits(uint32_t *body, const uint32_t *mask, const int32_t *bits, int N) {
    for (int i = 0; i < N; ++i) {
        if (!bits[i]) continue;
        for (int j = 0; j < N - i; ++j) body[i + j] += mask[j];
        for (int j = N - i; j < N; ++j) body[i + j - N] -= mask[j];
    }
}

void generate_keys(const Params &p, Rng &secret, Rng &mask, TfheHipSecretKey &sk, TfheHipCloudKey &ck) {
    const int n = p.n, N = p.N, k = p.k, l = p.l, kpl = p.kpl(), t = p.ks_t, base = 1 << p.ks_basebit;
    sk.p = p; ck.p = p;
    sk.lwe_key.resize(n);
    for (auto &b : sk.lwe_key) b = secret.bit();
    sk.tlwe_key.resize((size_t)k * N);
    for (auto &b : sk.tlwe_key) b = secret.bit();

    // bootstrapping key: BK_i = TGSW_{tlwe_key}(lwe_key[i])
    ck.bk.assign(p.bk_words(), 0);
    for (int i = 0; i < n; ++i)
        for (int row = 0; row < kpl; ++row) {
            uint32_t *smp = reinterpret_cast<uint32_t *>(ck.bk.data()) + ((size_t)i * kpl + row) * (size_t)(k + 1) * N;
            uint32_t *body = smp + (size_t)k * N;
            for (int u = 0; u < k; ++u)
                for (int j = 0; j < N; ++j) smp[(size_t)u * N + j] = (uint32_t)mask.torus();
            for (int j = 0; j < N; ++j) body[j] = (uint32_t)dtot32(secret.gauss(p.bk_stdev));
            for (int u = 0; u < k; ++u) add_mul_by_bits(body, smp + (size_t)u * N, sk.tlwe_key.data() + (size_t)u * N, N);
            const int bloc = row / l, jj = row % l;
            smp[(size_t)bloc * N] += (uint32_t)sk.lwe_key[i] << (32 - (jj + 1) * p.Bgbit);
        }

    // key-switching key: extracted TLWE key (kN bits) -> LWE key
    ck.ksk.assign(p.ksk_words(), 0);
    for (int i = 0; i < k * N; ++i)
        for (int j = 0; j < t; ++j)
            for (int v = 1; v < base; ++v) {
                uint32_t *row = reinterpret_cast<uint32_t *>(ck.ksk.data()) + (((size_t)i * t + j) * base + v) * (size_t)(n + 1);
                uint32_t b = 0;
                for (int q = 0; q < n; ++q) {
                    row[q] = (uint32_t)mask.torus();
                    b += row[q] * (uint32_t)sk.lwe_key[q];
                }
                b += (uint32_t)(sk.tlwe_key[i] * v) << (32 - (j + 1) * p.ks_basebit);
                b += (uint32_t)dtot32(secret.gauss(p.ks_stdev));
                row[n] = b;
            }
}

void encrypt_bit(const TfheHipSecretKey &sk, Rng &secret, Rng &mask, int32_t message, Torus32 *a, Torus32 *b) {
    const uint32_t mu = message ? (1u << 29) : 0u - (1u << 29);
    uint32_t body = mu + (uint32_t)dtot32(secret.gauss(sk.p.ks_stdev));
    for (int i = 0; i < sk.p.n; ++i) {
        a[i] = mask.torus();
        body += (uint32_t)a[i] * (uint32_t)sk.lwe_key[i];
    }
    *b = (Torus32)body;
}

Torus32 phase_of(const TfheHipSecretKey &sk, const Torus32 *a, Torus32 b) {
    uint32_t ph = (uint32_t)b;
    for (int i = 0; i < sk.p.n; ++i) ph -= (uint32_t)a[i] * (uint32_t)sk.lwe_key[i];
    return (Torus32)ph;
}

}  // namespace tfhe_hip
