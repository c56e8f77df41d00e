// dist_host.cpp -- TEST: a C++ host (no Python anywhere in the process) runs the slot-sharded match through
// libpeba1-dist's C ABI (include/peba1_dist.h) the way a PEBA1 server would around
// /root/reference/src/main.cpp:533-542.  Here: the plaintext provider (tests/mock/plain_tfhe.cpp), the host
// transport, and WORLD ranks simulated one after the other in this process (rank 0 last, so that its gather
// finds every other rank's contribution); with libtfhe-hip and RCCL the calls are the same, one process per GPU
// (INTEGRATION.md).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "peba1_dist.h"
#include "tfhe_hip.h"

static std::vector<std::vector<char>> g_mail;     // what each rank has sent

static int gather_cb(void *ctx, const void *send, void *recv, size_t bytes, int root) {
    const int rank = *static_cast<int *>(ctx);
    g_mail[(size_t)rank].assign(static_cast<const char *>(send), static_cast<const char *>(send) + bytes);
    if (rank != root) return 0;
    for (size_t r = 0; r < g_mail.size(); ++r) {
        if (g_mail[r].size() != bytes) return -1;             // a rank has not called yet
        std::memcpy(static_cast<char *>(recv) + r * bytes, g_mail[r].data(), bytes);
    }
    return 0;
}

static std::vector<char> g_bmail;                 // what the broadcast root has sent
static int bcast_cb(void *ctx, void *buffer, size_t bytes, int root) {
    const int rank = *static_cast<int *>(ctx);
    if (rank == root) { g_bmail.assign(static_cast<char *>(buffer), static_cast<char *>(buffer) + bytes); return 0; }
    if (g_bmail.size() != bytes) return -1;                   // the root has not called yet
    std::memcpy(buffer, g_bmail.data(), bytes);
    return 0;
}

static LweSample *encrypt_number(unsigned v, int bits, const TFheGateBootstrappingParameterSet *pp,
                                 const TFheGateBootstrappingSecretKeySet *key) {
    LweSample *a = new_gate_bootstrapping_ciphertext_array(bits, pp);
    for (int i = 0; i < bits; ++i) bootsSymEncrypt(a + i, (v >> i) & 1, key);
    return a;
}

int main(int argc, char **argv) {
    const int world = argc > 1 ? std::atoi(argv[1]) : 3, nslots = 11, bitsize = 8;
    TFheGateBootstrappingParameterSet *pp = new_default_gate_bootstrapping_parameters(128);
    TFheGateBootstrappingSecretKeySet *key = new_random_gate_bootstrapping_secret_keyset(pp);
    const TFheGateBootstrappingCloudKeySet *ck = &key->cloud;
    std::vector<unsigned> tmpl(nslots), probe(nslots);
    long d = 0;
    for (int i = 0; i < nslots; ++i) {
        tmpl[i] = (37u * i + 11) % 255;
        probe[i] = (91u * i + 5) % 256;
        d += ((long)probe[i] - (long)tmpl[i]) * ((long)probe[i] - (long)tmpl[i]);
    }
    for (int pass = 0; pass < 6; ++pass) {
        const long bound = pass & 1 ? d : d - 1;               // both sides of the threshold
        // reference-order phases; fast combine; fast per-rank phase + fast combine (the latency form)
        const int flags = pass < 2 ? 0 : pass < 4 ? PEBA1_DIST_FAST_COMBINE : PEBA1_DIST_FAST_COMBINE | PEBA1_DIST_FAST_PARTIAL;
        g_mail.assign((size_t)world, {});
        LweSample *result_b = new_gate_bootstrapping_ciphertext_array(24, pp);
        LweSample *enc_bound = encrypt_number((unsigned)bound, 24, pp, key);
        std::vector<int> ranks;
        for (int r = 1; r < world; ++r) ranks.push_back(r);
        ranks.push_back(0);
        for (int rank : ranks) {
            int lo, hi;
            peba1_dist_shard_slots(nslots, world, rank, &lo, &hi);
            std::vector<LweSample *> S, T;
            for (int i = lo; i < hi; ++i) { S.push_back(encrypt_number(probe[i], bitsize, pp, key)); T.push_back(encrypt_number(tmpl[i], bitsize, pp, key)); }
            Peba1Comm *comm = peba1_dist_init_host(gather_cb, &rank, world, rank);
            if (!comm) { std::printf("init: %s\n", peba1_dist_last_error()); return 1; }
            if (peba1_sharded_function_f(comm, rank == 0 ? result_b : nullptr, S.data(), T.data(), hi - lo,
                                         rank == 0 ? enc_bound : nullptr, bitsize, ck, flags) != 0) {
                std::printf("rank %d: %s\n", rank, peba1_dist_last_error());
                return 1;
            }
            peba1_dist_destroy(comm);
            for (LweSample *p : S) delete_gate_bootstrapping_ciphertext_array(bitsize, p);
            for (LweSample *p : T) delete_gate_bootstrapping_ciphertext_array(bitsize, p);
        }
        const int bit = bootsSymDecrypt(result_b, key);
        if (bit != (d > bound ? 1 : 0)) { std::printf("pass %d: bit %d, distance %ld, bound %ld\n", pass, bit, d, bound); return 1; }
        delete_gate_bootstrapping_ciphertext_array(24, result_b);
        delete_gate_bootstrapping_ciphertext_array(24, enc_bound);
    }
    // ---- failure containment: rank 1 fails locally -- it must still enter the exchange, so that
    // nobody is left inside it; it returns -1 with its own message and rank 0 returns -1 naming it
    if (world >= 2) {
        g_mail.assign((size_t)world, {});
        LweSample *result_b = new_gate_bootstrapping_ciphertext_array(24, pp);
        LweSample *enc_bound = encrypt_number(1u, 24, pp, key);
        std::vector<int> ranks;
        for (int r = 1; r < world; ++r) ranks.push_back(r);
        ranks.push_back(0);
        for (int rank : ranks) {
            int lo, hi;
            peba1_dist_shard_slots(nslots, world, rank, &lo, &hi);
            std::vector<LweSample *> S, T;
            for (int i = lo; i < hi; ++i) { S.push_back(encrypt_number(probe[i], bitsize, pp, key)); T.push_back(encrypt_number(tmpl[i], bitsize, pp, key)); }
            Peba1Comm *comm = peba1_dist_init_host(gather_cb, &rank, world, rank);
            if (rank == 1) peba1_dist_inject_failure(comm, 1);
            const int rc = peba1_sharded_function_f(comm, rank == 0 ? result_b : nullptr, S.data(), T.data(), hi - lo,
                                                    rank == 0 ? enc_bound : nullptr, bitsize, ck, 0);
            const std::string msg = peba1_dist_last_error();
            const bool expect_fail = rank == 0 || rank == 1;
            if ((rc != 0) != expect_fail) { std::printf("containment: rank %d returned %d (%s)\n", rank, rc, msg.c_str()); return 1; }
            if (rank == 1 && msg.find("injected failure") == std::string::npos) { std::printf("containment: rank 1 says '%s'\n", msg.c_str()); return 1; }
            if (rank == 0 && msg.find("rank 1 reported a failure") == std::string::npos) { std::printf("containment: rank 0 says '%s'\n", msg.c_str()); return 1; }
            if (g_mail[(size_t)rank].empty()) { std::printf("containment: rank %d never entered the exchange\n", rank); return 1; }
            peba1_dist_destroy(comm);
            for (LweSample *p : S) delete_gate_bootstrapping_ciphertext_array(bitsize, p);
            for (LweSample *p : T) delete_gate_bootstrapping_ciphertext_array(bitsize, p);
        }
        delete_gate_bootstrapping_ciphertext_array(24, result_b);
        delete_gate_bootstrapping_ciphertext_array(24, enc_bound);
    }
    // ---- 1-to-N identification from the C++ host (BASELINE configs[3]): every rank matches the probe
    // against its 3 templates, 2 recorded per flush; the match bits of all ranks arrive on rank 0
    {
        const int m_local = 3, ns = 4;
        std::vector<unsigned> pr(ns);
        for (int i = 0; i < ns; ++i) pr[i] = (91u * i + 5) % 256;
        const unsigned bound = 300;
        g_mail.assign((size_t)world, {});
        std::vector<int> expect((size_t)world * m_local);
        LweSample *all = new_gate_bootstrapping_ciphertext_array(world * m_local, pp);
        std::vector<int> ranks;
        for (int r = 1; r < world; ++r) ranks.push_back(r);
        ranks.push_back(0);
        // the ONE encrypted probe: rank 0 encrypts it, peba1_dist_broadcast_samples brings it to every rank (root first here:
        // the ranks are simulated one after the other)
        std::vector<LweSample *> probe_of((size_t)world);
        g_bmail.clear();
        for (int rank = 0; rank < world; ++rank) {
            LweSample *pa = new_gate_bootstrapping_ciphertext_array(ns * bitsize, pp);
            if (rank == 0)
                for (int i = 0; i < ns; ++i)
                    for (int k = 0; k < bitsize; ++k) bootsSymEncrypt(pa + i * bitsize + k, (pr[i] >> k) & 1, key);
            Peba1Comm *comm = peba1_dist_init_host(gather_cb, &rank, world, rank);
            if (peba1_dist_broadcast_samples(comm, pa, ns * bitsize, pp, 0) == 0) { std::printf("broadcast without a callback succeeded\n"); return 1; }
            peba1_dist_set_host_bcast(comm, bcast_cb);
            if (peba1_dist_broadcast_samples(comm, pa, ns * bitsize, pp, 0) != 0) { std::printf("broadcast rank %d: %s\n", rank, peba1_dist_last_error()); return 1; }
            peba1_dist_destroy(comm);
            probe_of[(size_t)rank] = pa;
        }
        for (int rank : ranks) {
            std::vector<LweSample *> P, T;
            for (int i = 0; i < ns; ++i) P.push_back(probe_of[(size_t)rank] + i * bitsize);      // slot i of the broadcast probe
            for (int m = 0; m < m_local; ++m) {
                long dd = 0;
                for (int i = 0; i < ns; ++i) {
                    // template (rank, m): the probe itself for (0, 1) -- the genuine match -- else shifted values
                    const unsigned v = (rank == 0 && m == 1) ? pr[i] : (pr[i] + 7u * (unsigned)(rank * m_local + m + 1) + 3u * (unsigned)i) % 256;
                    T.push_back(encrypt_number(v, bitsize, pp, key));
                    dd += ((long)pr[i] - (long)v) * ((long)pr[i] - (long)v);
                }
                expect[(size_t)rank * m_local + m] = dd > (long)bound ? 1 : 0;
            }
            LweSample *enc_bound = encrypt_number(bound, 24, pp, key);
            LweSample *mine = new_gate_bootstrapping_ciphertext_array(m_local, pp);
            Peba1Comm *comm = peba1_dist_init_host(gather_cb, &rank, world, rank);
            if (peba1_identify(comm, rank == 0 ? all : nullptr, mine, P.data(), T.data(), m_local, ns, enc_bound, bitsize, ck,
                               2, rank & 1 ? PEBA1_IDENTIFY_FAST : 0) != 0) {
                std::printf("identify rank %d: %s\n", rank, peba1_dist_last_error());
                return 1;
            }
            for (int m = 0; m < m_local; ++m)
                if (bootsSymDecrypt(mine + m, key) != expect[(size_t)rank * m_local + m]) { std::printf("identify: rank %d match %d\n", rank, m); return 1; }
            peba1_dist_destroy(comm);
            delete_gate_bootstrapping_ciphertext_array(m_local, mine);
            delete_gate_bootstrapping_ciphertext_array(24, enc_bound);
            delete_gate_bootstrapping_ciphertext_array(ns * bitsize, probe_of[(size_t)rank]);
            for (LweSample *p : T) delete_gate_bootstrapping_ciphertext_array(bitsize, p);
        }
        int zeros = 0;
        for (int k = 0; k < world * m_local; ++k) {
            const int bit = bootsSymDecrypt(all + k, key);
            if (bit != expect[(size_t)k]) { std::printf("identify: gathered bit %d is %d\n", k, bit); return 1; }
            zeros += bit == 0;
        }
        if (expect[1] != 0 || zeros < 1) { std::printf("identify: the genuine template did not match\n"); return 1; }
        delete_gate_bootstrapping_ciphertext_array(world * m_local, all);
    }
    // bad arguments are reported, not fatal
    if (peba1_dist_init_host(nullptr, nullptr, 2, 0) != nullptr || !*peba1_dist_last_error()) return 1;
    std::printf("DIST-HOST-OK world %d distance %ld\n", world, d);
    return 0;
}
