/*
 * peba1_dist.h -- C ABI of libpeba1-dist: the multi-GPU forms of the PEBA1 match for a C++ (or any C-ABI) host,
 * one process per GPU.  north_star keeps the host in C++: a PEBA1 server that calls Function_f per enrolled
 * client (/root/reference/src/main.cpp:533-542) shards exactly there, with no Python in the process.
 *
 *   slot-sharded match (BASELINE configs[2]): every rank evaluates the reference's slot loop
 *     (Math.cpp:351-360) over ITS slots, ONE gather moves each rank's 24-ciphertext partial sum to rank 0,
 *     rank 0 adds the partial sums and runs the comparator (Math.cpp:384).
 *   1-to-N identification (configs[3]): matches are independent, no data-path collective; only the match-bit
 *     ciphertexts are gathered to rank 0 (peba1_dist_gather_samples).
 *
 * Transport.  Either RCCL over xGMI -- device buffers, the gather enqueued on libtfhe-hip's own stream between
 * the stream-ordered export and import (tfhe_hip.h), so nothing waits on the host -- or any host-memory
 * exchange the caller supplies as a callback (MPI, sockets, torch.distributed/gloo): how several processes
 * rehearse the N > 1 path on one GPU, and how a deployment without RCCL plugs in.
 *
 * The library calls only the public gate API (tfhe/tfhe.h, tfhe_hip.h) and libpeba1-circuits; its symbols
 * resolve at load time against whichever provider is loaded (libtfhe-hip.so; the tests' plaintext provider for
 * the host transport).  RCCL is opened on first use (dlopen librccl.so), not linked.
 */
#ifndef PEBA1_DIST_H
#define PEBA1_DIST_H

#include <stddef.h>

#include "peba1_circuits.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct Peba1Comm Peba1Comm;

/* contiguous slot range [*lo, *hi) of `rank`; earlier ranks take the remainder */
void peba1_dist_shard_slots(int nslots, int world, int rank, int *lo, int *hi);

/* ---- RCCL transport ---- */
#define PEBA1_DIST_ID_BYTES 128
/* rank 0: a fresh RCCL unique id (ncclGetUniqueId); ship the 128 bytes to the other ranks by any means */
int peba1_dist_unique_id(void *id128);
/* every rank (collective, ncclCommInitRank); selects nothing: call tfhe_hip_set_device first */
Peba1Comm *peba1_dist_init_rccl(const void *id128, int world, int rank);
/* or adopt a communicator the host already has (an ncclComm_t); not destroyed by peba1_dist_destroy */
Peba1Comm *peba1_dist_adopt_rccl(void *nccl_comm, int world, int rank);

/* ---- host-memory transport ----
 * gather(ctx, send, recv, bytes, root): every rank contributes `bytes` bytes at `send`; on `root`, recv holds
 * world * bytes, rank-major; recv is NULL elsewhere.  Returns 0 on success. */
typedef int (*peba1_gather_fn)(void *ctx, const void *send, void *recv, size_t bytes, int root);
Peba1Comm *peba1_dist_init_host(peba1_gather_fn gather, void *ctx, int world, int rank);

void peba1_dist_destroy(Peba1Comm *comm);
int peba1_dist_rank(const Peba1Comm *comm);
int peba1_dist_world(const Peba1Comm *comm);
/* message of the last failed call of this library on this thread's communicator ("" if none) */
const char *peba1_dist_last_error(void);

/* ---- evidence of what ran (bench.py `dist`) ----
 * version of the RCCL library this process opened (ncclGetVersion: 2xxyy), 0 if RCCL cannot be opened */
int peba1_dist_rccl_version(void);
/* 1 = RCCL transport, 0 = host-memory transport */
int peba1_dist_transport(const Peba1Comm *comm);
/* out4 = {status-word exchanges, gathers, broadcasts, payload bytes this rank sent} since the communicator was made */
void peba1_dist_counters(const Peba1Comm *comm, uint64_t out4[4]);
/* how the status words of the RCCL transport travel: 2 = on a communicator of their own (ncclCommSplit of the data
 * communicator, made with this one) and a stream of their own -- they wait neither for the gates in flight nor for the
 * previous data collective; 1 = on the data communicator and the provider's stream (the loaded RCCL has no ncclCommSplit,
 * the split failed, or PEBA1_DIST_NO_STATUS_COMM is set): ordered by the stream, behind the gates; 0 = host transport (the
 * word rides in front of the payload) */
int peba1_dist_status_channel(const Peba1Comm *comm);
/* out2 = {collectives this communicator has issued, rolling hash of their (kind, words per rank, root) in issue order}:
 * equal on every rank of a job iff every rank issued the same collectives in the same order */
void peba1_dist_sequence(const Peba1Comm *comm, uint64_t out2[2]);

/* flags */
#define PEBA1_DIST_FAST_COMBINE 1   /* rank 0: carry-save compressor + prefix adder + prefix comparator
                                       (peba1_combine_and_compare_fast, ~20 levels) instead of the pairwise tree of the
                                       reference's ripple adders + its comparator (~290 levels at 8 ranks) */

#define PEBA1_DIST_FAST_PARTIAL 2   /* every rank: its slots' squared distance from peba1_euclidean_distance_fast (borrow
                                       chain, squarer, ONE carry-save column compressor, prefix adder: ~55 levels for 32
                                       slots) instead of the reference's slot loop (peba1_partial_distance, ~183 levels).
                                       NOT the reference's gate sequence: same decrypted partial sum.  Both flags
                                       together are the latency form of the sharded match */

/* Slot-sharded Function_f (Math.cpp:379-387 split as SURVEY.md 8e): a[i], b[i] (i < nslots_local) are THIS rank's
 * slots of the sample and the template, `bitsize` samples each; bound_match and result_b (3 * bitsize = 24 samples,
 * caller-allocated) are read / written on rank 0 only (may be NULL elsewhere).  result_b[0] = (distance > bound).
 * Collective: every rank of the communicator calls it.  Returns 0, or -1 with peba1_dist_last_error(). */
int peba1_sharded_function_f(Peba1Comm *comm, LweSample *result_b, LweSample *const *a, LweSample *const *b,
                             int nslots_local, LweSample *bound_match, int bitsize,
                             const TFheGateBootstrappingCloudKeySet *ck, int flags);

/* The two phases on their own (tests, logical ranks on one device): phase 1 leaves this rank's packed partial sum
 * (24 * sample_words int32) in `packed` (host memory); phase 3 combines `nparts` packed partial sums (rank-major)
 * on the calling rank. */
int peba1_sharded_partial_packed(LweSample *const *a, LweSample *const *b, int nslots_local, int bitsize,
                                 const TFheGateBootstrappingCloudKeySet *ck, int32_t *packed, int flags);
int peba1_sharded_combine_packed(LweSample *result_b, const int32_t *packed, int nparts, LweSample *bound_match,
                                 const TFheGateBootstrappingCloudKeySet *ck, int flags);

/* Gather `count` ciphertexts of every rank to rank 0 (identification: the match bits).  `all` (rank 0: world * count
 * samples, rank-major; NULL elsewhere) and `mine` are sample arrays of parameter set `params`. */
int peba1_dist_gather_samples(Peba1Comm *comm, LweSample *all, const LweSample *mine, int count,
                              const TFheGateBootstrappingParameterSet *params);

/* Broadcast `count` ciphertexts from `root` to every rank (identification: the ONE encrypted probe reaches every rank's
 * share of the gallery).  `samples`: sample array of parameter set `params` on every rank -- read on root, written
 * elsewhere.  RCCL: export -> ncclBroadcast -> import on the provider's stream (2.5 KB per sample).  Host transport: through
 * the callback given to peba1_dist_set_host_bcast (bcast(ctx, buffer, bytes, root): in place; returns 0), without one the
 * call fails with a message.  Collective; failure containment as below.  Returns 0 or -1. */
typedef int (*peba1_bcast_fn)(void *ctx, void *buffer, size_t bytes, int root);
void peba1_dist_set_host_bcast(Peba1Comm *comm, peba1_bcast_fn bcast);
int peba1_dist_broadcast_samples(Peba1Comm *comm, LweSample *samples, int count,
                                 const TFheGateBootstrappingParameterSet *params, int root);

/* 1-to-N identification (BASELINE configs[3]): the loop a PEBA1 server puts around Function_f, one call per enrolled
 * client (/root/reference/src/main.cpp:533-542), for THIS rank's share of the gallery -- with no Python in the process.
 *   probe: nslots slot arrays of `bitsize` samples; templates: m_local * nslots slot arrays, template m at
 *   templates[m * nslots .. (m + 1) * nslots); bound_match: 3 * bitsize samples.
 *   mine (m_local samples, caller-allocated): mine[m] = Enc(distance(probe, template m) > bound), the reference's polarity.
 *   all (rank 0: world * m_local samples, rank-major; NULL elsewhere): the match bits of every rank, brought over by
 *   ONE gather at the end -- the matches themselves are independent, there is no data-path collective.
 *   comm: NULL = single process (no gather).  Every rank passes the same m_local.
 * `group` matches are recorded per flush (>= 1; 4 is a good value: the narrow tail levels of one match are filled by the
 * others, and device memory is bounded by the group, not by m_local); with libtfhe-hip the flushes are pipelined
 * (tfhe_hip_flush_async): the next group is recorded while the device runs the one before.
 * flags: PEBA1_IDENTIFY_FAST = the depth-optimised circuit (peba1_function_f_fast; NOT the reference's gate sequence).
 * Collective when comm has more than one rank.  Returns 0, or -1 with peba1_dist_last_error(). */
#define PEBA1_IDENTIFY_FAST 1
int peba1_identify(Peba1Comm *comm, LweSample *all, LweSample *mine, LweSample *const *probe,
                   LweSample *const *templates, int m_local, int nslots, LweSample *bound_match, int bitsize,
                   const TFheGateBootstrappingCloudKeySet *ck, int group, int flags);

/* ---- failure containment (every collective of this library) ----
 * A rank that fails locally (an allocation, the export that enqueues its pending gates) still ENTERS the exchange, carrying
 * a status word, so that no rank is left inside a collective:
 *   RCCL: the status words are all-gathered first (one int per rank, on a stream of the communicator's own: the host
 *     waits for the peers' words, NOT for the gates in flight on the provider's stream -- the recording of the next
 *     circuit overlaps them as in a single-process run); if any rank reports a failure EVERY rank skips the data gather
 *     and returns -1, its message naming the failed rank;
 *   host transport: the status word rides in front of each rank's payload; rank 0 returns -1 naming the failed rank, the
 *     failed rank returns -1 with its own message (a gather cannot tell the other ranks).
 * Every host wait of a communicator of more than one rank is bounded: PEBA1_DIST_TIMEOUT_S (default 600; 0 = unbounded)
 * or peba1_dist_set_timeout().  When a wait expires -- a peer never arrived -- the process prints which wait, on which
 * rank, and exits with TFHE_HIP_EXIT_DEADLINE (tfhe_hip.h): non-zero, no retry, no re-exec. */
int peba1_dist_set_timeout(Peba1Comm *comm, double seconds);
/* test hook: the next `count` collectives of this rank report a local failure instead of contributing */
void peba1_dist_inject_failure(Peba1Comm *comm, int count);

#ifdef __cplusplus
}
#endif
#endif
