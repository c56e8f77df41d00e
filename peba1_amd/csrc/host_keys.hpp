// host_keys.hpp -- host-side key material, parameters and the deterministic
// PRNG of libtfhe-hip.  Key generation, encryption and decryption run on the
// CPU (they are off the timed path); only the
// evaluation keys are uploaded to the device.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/tfhe/tfhe_core.h"

namespace tfhe_hip {

struct Params {
    int32_t n, N, k, l, Bgbit, ks_t, ks_basebit;
    double ks_stdev, bk_stdev, max_stdev;
    int32_t kpl() const { return (k + 1) * l; }
    int32_t ct_words() const { return n + 1; }
    int32_t ct_stride() const { return (n + 1 + 3) & ~3; }
    int32_t u_stride() const { return (k * N + 1 + 3) & ~3; }
    uint32_t decomp_offset() const;
    uint32_t ks_prec_offset() const { return 1u << (32 - (1 + ks_basebit * ks_t)); }
    size_t bk_words() const { return (size_t)n * kpl() * (k + 1) * N; }
    size_t ksk_words() const { return (size_t)k * N * ks_t * (size_t)(1 << ks_basebit) * (n + 1); }
};

// The C structs behind a TFheGateBootstrappingParameterSet allocated by this library.
struct ParamBundle {
    TFheGateBootstrappingParameterSet set;
    LweParams lwe;
    TLweParams tlwe;
    TGswParams tgsw;
    Params p;
};
ParamBundle *make_param_bundle(const Params &p);
const Params &params_of(const TFheGateBootstrappingParameterSet *set);
bool default_params(int32_t minimum_lambda, Params &out);
Params p2048_params();

// Two generators behind one interface.
//  * Seeded: xoshiro256** seeded through splitmix64 -- the generator of the key-derivation specification that the
//    product and the test oracle share (DESIGN.md "key derivation"); reproducible, NOT cryptographic; reached only
//    through tfhe_hip_new_secret_keyset_seeded / tfhe_hip_set_encrypt_seed (tests, golden fixtures).
//  * Secure (Rng::secure()): the ChaCha20 key stream (RFC 8439 block function, 64-bit block counter) under a 256-bit
//    key and a 64-bit nonce read from the operating system (getrandom): what new_random_gate_bootstrapping_secret_keyset
//    and bootsSymEncrypt draw from by default.  Secrets (key bits, noise) and the public masks come from two
//    independently keyed streams, so the published masks say nothing about the stream the secrets came from.
class Rng {
public:
    explicit Rng(uint64_t seed = 0) { reseed(seed); }
    static Rng secure();             // aborts with a message if the OS offers no entropy
    void reseed(uint64_t seed);      // (re)start the seeded generator
    uint64_t next();
    Torus32 torus() { return (Torus32)(uint32_t)(next() >> 32); }
    int32_t bit() { return (int32_t)(next() >> 63); }
    double gauss(double sigma);
    bool is_secure() const { return chacha_; }
    // known-answer hook (tests): the key-stream block of (key, counter, nonce)
    static void chacha_block(const uint32_t key[8], uint64_t counter, const uint32_t nonce[2], uint32_t out[16]);
private:
    void refill();
    void rekey_secure();             // key, nonce from the OS; counter 0 (also after fork(): host_keys.cpp)
    uint32_t fork_gen_ = 0;          // the fork generation the secure stream was keyed in
    bool chacha_ = false;
    uint64_t s_[4];                  // xoshiro state
    uint32_t key_[8], nonce_[2];     // ChaCha20 key and nonce
    uint64_t counter_ = 0;
    uint32_t block_[16];
    int pos_ = 16;                   // next unread word of block_
};
Torus32 dtot32(double d);
// fills `bytes` bytes from the operating system (getrandom, /dev/urandom as a fallback); false if neither works
bool os_random(void *out, size_t bytes);

}  // namespace tfhe_hip

struct DeviceKeyImage;   // engine.hpp

// Secret and cloud key material (names match the opaque pointers in tfhe_core.h)
struct TfheHipSecretKey {
    tfhe_hip::Params p;
    std::vector<int32_t> lwe_key;    // [n]
    std::vector<int32_t> tlwe_key;   // [k][N]
};
struct TfheHipCloudKey {
    tfhe_hip::Params p;
    std::vector<Torus32> bk;         // [n][(k+1)l][k+1][N]
    std::vector<Torus32> ksk;        // [kN][t][base][n+1]
    DeviceKeyImage *dev = nullptr;   // null for host-only keysets
};

namespace tfhe_hip {
// `secret` yields the key bits and every noise sample, `mask` the public uniform masks.  The seeded path passes ONE
// generator for both (the draw order of DESIGN.md, which the oracle regenerates); the default path two secure ones.
void generate_keys(const Params &p, Rng &secret, Rng &mask, TfheHipSecretKey &sk, TfheHipCloudKey &ck);
void encrypt_bit(const TfheHipSecretKey &sk, Rng &secret, Rng &mask, int32_t message, Torus32 *a, Torus32 *b);
Torus32 phase_of(const TfheHipSecretKey &sk, const Torus32 *a, Torus32 b);
}  // namespace tfhe_hip
