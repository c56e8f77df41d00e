/*
 * tfhe_hip.h -- extensions of libtfhe-hip beyond the upstream tfhe C API.
 *
 * Everything here is plain C ABI (pointers and sizes, no C++/torch types).  The
 * upstream-compatible surface is in tfhe/tfhe_gate_bootstrapping_functions.h;
 * this header adds what a one-gate-per-call API cannot express: seeded keys,
 * deferred (batched, levelised) execution, array-wide gates, raw ciphertext
 * words, statistics, and kernel-level entry points used by the parity tests.
 */
#ifndef TFHE_HIP_H
#define TFHE_HIP_H

#include "tfhe/tfhe_core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* gate codes for tfhe_hip_gate_batch (2-input gates share one kernel and differ
 * only in the linear prelude) */
enum TfheHipGate {
    TFHE_HIP_NAND = 0, TFHE_HIP_OR, TFHE_HIP_AND, TFHE_HIP_NOR, TFHE_HIP_XOR, TFHE_HIP_XNOR,
    TFHE_HIP_ANDNY, TFHE_HIP_ANDYN, TFHE_HIP_ORNY, TFHE_HIP_ORYN
};

/* ---- error channel: the upstream API returns void everywhere, so failures
 * are reported here.  Conditions the caller can recover from -- ciphertext slot pool exhausted,
 * a sample this library did not allocate, a sample used with a key of another LWE dimension,
 * a null key, a malformed / truncated / foreign file -- leave the call WITHOUT EFFECT (result
 * untouched; loaders return NULL; int entry points return -1) and set the message.  Only HIP
 * runtime failures and "no GPU present" abort the process, like upstream's fatal paths. ---- */
const char *tfhe_hip_last_error(void);
void tfhe_hip_clear_error(void);

/* ---- device selection (call before the first keyset is created) ---- */
int tfhe_hip_set_device(int device);
int tfhe_hip_get_device(void);
/* PCI bus id ("0000:c1:00.0", NUL-terminated; len >= 16) of the device the library runs on: what a multi-GPU job prints
 * per rank to show that N ranks drive N distinct GPUs (bench.py `dist.devices`).  Initialises the engine.  Returns 0 / -1.
 * The library binds the calling thread to its device only for the duration of the calls that reach the HIP runtime and
 * gives the caller's current device back on return. */
int tfhe_hip_device_pci_bus_id(char *out, int len);

/* ---- parameters ---- */
TFheGateBootstrappingParameterSet *tfhe_hip_new_parameters(
    int32_t n, int32_t N, int32_t k, int32_t l, int32_t Bgbit, int32_t ks_t, int32_t ks_basebit,
    double ks_stdev, double bk_stdev, double max_stdev);
/* BASELINE.json configs[4]: N=2048, Bg=2^6, l=3 (n=1024, ks 8x2 bit fixed by this repo) */
TFheGateBootstrappingParameterSet *tfhe_hip_new_p2048_parameters(void);

/* ---- randomness ----
 * Default (new_random_gate_bootstrapping_secret_keyset, bootsSymEncrypt): ChaCha20 key streams (RFC 8439 block
 * function), each under its own 256-bit key + 64-bit nonce from getrandom(); the secrets (key bits, every noise
 * sample) and the public masks (bk / ksk masks, the `a` words of ciphertexts) come from two independently keyed
 * streams.  No OS entropy = abort with a message.
 * The entry points below replace that with the SEEDED generator of the key-derivation specification this library
 * shares with the test oracle (xoshiro256** through splitmix64, one stream for secrets and masks):
 * reproducible and NOT cryptographic -- for tests and golden fixtures only, never for keys that protect data. ---- */
TFheGateBootstrappingSecretKeySet *tfhe_hip_new_secret_keyset_seeded(
    const TFheGateBootstrappingParameterSet *params, uint64_t seed);
/* host-only keyset (no device upload): lets CPU-only tests check key derivation */
TFheGateBootstrappingSecretKeySet *tfhe_hip_new_secret_keyset_seeded_host(
    const TFheGateBootstrappingParameterSet *params, uint64_t seed);
void tfhe_hip_set_encrypt_seed(uint64_t seed);
/* 1 after tfhe_hip_set_encrypt_seed (bootsSymEncrypt is reproducible, not secure), else 0 */
int tfhe_hip_randomness_is_seeded(void);
/* known-answer hook for the default generator: ChaCha20 key-stream block of (key[8], 64-bit counter, nonce[2]);
 * RFC 8439 2.3.2 = counter 1 | 0x09000000 << 32, nonce {0x4a000000, 0} */
void tfhe_hip_test_chacha20_block(const uint32_t *key8, uint64_t counter, const uint32_t *nonce2, uint32_t *out16);

/* read-only views of the key material (parity tests hash these) */
const int32_t *tfhe_hip_key_lwe(const TFheGateBootstrappingSecretKeySet *key, int64_t *count);
const int32_t *tfhe_hip_key_tlwe(const TFheGateBootstrappingSecretKeySet *key, int64_t *count);
const Torus32 *tfhe_hip_key_bk(const TFheGateBootstrappingCloudKeySet *cloud, int64_t *count);
const Torus32 *tfhe_hip_key_ksk(const TFheGateBootstrappingCloudKeySet *cloud, int64_t *count);

/* ---- raw ciphertext words: n mask words then the body ---- */
int32_t tfhe_hip_sample_words(const TFheGateBootstrappingParameterSet *params);
/* samples[0..count) are consecutive elements of one array; words are packed
 * [count][n+1] */
int tfhe_hip_export_samples(const LweSample *samples, int32_t count,
                            const TFheGateBootstrappingParameterSet *params, Torus32 *out_words);
int tfhe_hip_import_samples(LweSample *samples, int32_t count,
                            const TFheGateBootstrappingParameterSet *params, const Torus32 *in_words);
/* same, to/from DEVICE memory (e.g. a torch tensor's data_ptr) for collectives */
int tfhe_hip_export_samples_device(const LweSample *samples, int32_t count,
                                   const TFheGateBootstrappingParameterSet *params, void *device_words);
int tfhe_hip_import_samples_device(LweSample *samples, int32_t count,
                                   const TFheGateBootstrappingParameterSet *params, const void *device_words);
/* Stream-ordered forms for collectives driven from the host language (libpeba1-dist): the transfer is ENQUEUED on
 * the library's own HIP stream, tfhe_hip_stream(), and the call returns without waiting.  Anything the caller then
 * enqueues on that same stream -- an RCCL collective on the exported buffer, or the next flush after an import -- is
 * ordered behind it by the stream itself, with no host synchronisation.  The buffer must stay allocated until the
 * stream has passed the transfer (hipStreamSynchronize(tfhe_hip_stream()), or an event). */
int tfhe_hip_export_samples_device_async(const LweSample *samples, int32_t count,
                                         const TFheGateBootstrappingParameterSet *params, void *device_words);
int tfhe_hip_import_samples_device_async(LweSample *samples, int32_t count,
                                         const TFheGateBootstrappingParameterSet *params, const void *device_words);
/* the hipStream_t every kernel and transfer of this library runs on (created non-blocking, highest priority) */
void *tfhe_hip_stream(void);
/* Every read of a sample by the host (bootsSymDecrypt, tfhe_hip_sync_samples, the exports) is ordered behind whatever
 * was enqueued on that stream before it -- a flush in flight, a stream-ordered import behind a collective -- and
 * tfhe_hip_wait() returns only when all of it has completed. */
/* refresh the host mirror (a, b) of samples whose value lives on the device */
int tfhe_hip_sync_samples(const LweSample *samples, int32_t count);

/* ---- execution mode ----
 * deferred (default): boots* calls are recorded (SSA-renamed, so overwritten and freed
 * temporaries are safe), levelised by data dependence, and executed level by level as batched
 * kernels at the next bootsSymDecrypt / export / tfhe_hip_flush() (and on their own before the
 * slot pool runs dry).  Everything observable through the API is identical to per-call
 * execution; only the public struct fields sample->a / sample->b are stale until the sample
 * is decrypted, exported or passed to tfhe_hip_sync_samples().
 * immediate (TFHE_HIP_DEFERRED=0 in the environment, or tfhe_hip_set_deferred(0)): every
 * boots* call is complete on return with the host mirror refreshed, as upstream -- one gate
 * per kernel launch, 3.4 ms per gate. */
void tfhe_hip_set_deferred(int on);
int tfhe_hip_get_deferred(void);
int tfhe_hip_flush(void);   /* returns the number of levels executed, <0 on error */
/* Pipelined form for a stream of circuits (a server matching one probe after another): the pending gates are levelised and
 * their launches ENQUEUED, and the call returns while the device works.  The caller goes on recording the next circuit --
 * its recording, the elimination of dead gates, the levelling and the plan of its own flush all overlap the execution of
 * this one (at most one flush is in flight: the next flush waits for it just before it launches).  Everything that
 * observes a result (bootsSymDecrypt, exports, tfhe_hip_sync_samples, tfhe_hip_get_stats) waits first;
 * tfhe_hip_wait() waits explicitly.  tfhe_hip_flush() is the synchronous form and also completes a flush in flight. */
int tfhe_hip_flush_async(void);
int tfhe_hip_wait(void);

/* ---- bounded host waits (multi-process runs) ----
 * "sync_deadline_ms" (tfhe_hip_set_tuning; env TFHE_HIP_SYNC_DEADLINE_MS; default 0 = wait for ever): when > 0, no
 * host wait on the library's stream lasts longer than this.  A wait that does -- a collective whose peer never
 * arrived, a kernel that never ends -- prints what was waited for and the label below on stderr and ends the process
 * with exit code TFHE_HIP_EXIT_DEADLINE (_exit: no retry, no re-exec -- the state of the device is unknown).
 * libpeba1-dist sets it for every communicator of more than one rank (PEBA1_DIST_TIMEOUT_S, default 600 s). */
#define TFHE_HIP_EXIT_DEADLINE 86
/* waits (bounded as above) until everything enqueued on tfhe_hip_stream() so far -- by this library or by the caller
 * (a collective) -- has completed; also completes a flush in flight.  Returns 0. */
int tfhe_hip_stream_sync(void);
/* waits (bounded as above) for a HIP event of the caller's own (a hipEvent_t recorded on any stream of the library's
 * device) -- libpeba1-dist reads the status words of a collective back this way, from a stream of its own, without waiting
 * for the gates in flight on tfhe_hip_stream().  `what` names the wait in the deadline message.  Returns 0. */
int tfhe_hip_wait_event(void *hip_event, const char *what);
/* who is waiting, for that message ("rank 3 of 8: gather of the partial sums"); copied, at most 127 characters */
void tfhe_hip_set_diag_label(const char *label);

/* result[i] = gate(a[i], b[i]) for i < count, one batched launch */
int tfhe_hip_gate_batch(int gate, LweSample *result, const LweSample *a, const LweSample *b,
                        int32_t count, const TFheGateBootstrappingCloudKeySet *bk);

/* ---- tuning (ten names that results never depend on, and the opt-in "fold_constants") ----
 * "br_variant": which form of the blind-rotate kernel runs wide launches (env TFHE_HIP_BR_VARIANT): -1 (default) =
 * the fastest measured for the ring size (N = 1024: 4 waves per rotation; N = 2048: split), 0 = 4 waves (N = 1024),
 * 2 = split (8 waves, every transform as two half-size ones), 4 = 2 waves (N = 1024; the form with the widest admissible
 * gadget range).  A form whose bounds do not admit the key's gadget is replaced by one that does (br_forms.hpp).
 * "br8_max_rotations": launches of at most min(this, CU count) rotations use the 8-wave form at N = 1024 (default 2^30,
 * env TFHE_HIP_BR8_MAX; 0 = never).
 * "br_tail8": 1 (default, env TFHE_HIP_BR_TAIL8) = the last, at most half-filled round of a wide 4-wave launch runs on the
 * 8-wave form as a second launch.
 * "br_digit_table": 1 (default, env TFHE_HIP_BR_TABLE) = products of gadget digits with the first twiddles come from
 * LDS tables where the digits are at most 7 bits wide (split form: 2 = the stage-0 table only); 0 = multiplies.
 * "ks_tile": 16 (default), 24 or 32 (index form only; elsewhere read as 16) = launches of at least 2*tile key switches use
 * a tiled kernel (a workgroup streams the KSK rows of one range once for `tile` gates); 0 = always one workgroup per
 * (gate, range); env TFHE_HIP_KS_TILE.
 * "ks_index": 1 (default, env TFHE_HIP_KS_INDEX) = the tiled kernel keeps a thread's 16-byte column of the three rows of a
 * digit position in pinned registers addressed through the VGPR index mode by the wave-uniform digit (56 ms of key switch
 * per match); 0 = the rows in thread-private LDS strips (105 ms; plain HIP source).
 * "reuse_gates": 1 (default) = in deferred mode a gate recorded again with the same operand
 * samples before the flush shares the pending gate's result instead of being evaluated again
 * (same function of the same ciphertexts, so the same words); 0 = evaluate every call.
 * "eliminate_dead": 1 (default) = at a flush, a recorded gate whose result no sample handle holds any more and no
 * live gate reads is not evaluated (the reference's adders compute carries they then drop: 3 % of a match); nothing
 * observable changes; 0 = evaluate every recorded gate.
 * "balance_levels": 1 (default) = slack-aware level filling at flush, 0 = plain ASAP
 * levels.
 * "fold_constants": 0 (default) / 1 (env TFHE_HIP_FOLD_CONSTANTS) = OPT-IN constant folding at record time: a gate one of
 * whose operands is a trivial sample (bootsCONSTANT, a fresh sample, a copy of either: a PUBLIC constant) is not
 * bootstrapped -- its result is the constant, the other operand or its negation; a MUX with a constant data operand becomes
 * a two-input gate.  The reference's match circuit loses 62 % of its bootstraps that way.  Decrypted results are the same;
 * the ciphertext WORDS are not TFHE's (which bootstraps every gate), which is why it is off unless asked for -- the only
 * tuning results depend on.  The oracle folds by the same rule (orc_boots_set_fold): folded circuits have digests too.
 * "sync_deadline_ms": see "bounded host waits" above.
 * (Environment only: TFHE_HIP_KS_BLOCKS / TFHE_HIP_KS_MAX_SPLITS / TFHE_HIP_KS_SPLIT_TIES, how key switches are cut into
 * coefficient ranges -- engine.hpp.)
 * Returns 0, or -1 for an unknown name. */
int tfhe_hip_set_tuning(const char *name, int64_t value);

/* ---- statistics ---- */
typedef struct TfheHipStats {
    uint64_t blind_rotates;     /* K2 instances */
    uint64_t keyswitches;       /* K3 instances */
    uint64_t linear_ops;        /* NOT */
    uint64_t levels;            /* batched levels executed */
    uint64_t flushes;
    uint64_t br_launches;       /* blind-rotate kernel launches */
    double   ms_blind_rotate;   /* device time, HIP events on the engine stream */
    double   ms_keyswitch;
    double   ms_flush_wall;     /* host wall time inside flush */
    double   ms_blind_rotate_busy; /* time during which at least one blind-rotate launch was running
                                      (== ms_blind_rotate: a flush is one level sequence on one stream) */
    uint64_t reused_gates;      /* recorded gates served by an identical pending gate ("reuse_gates") */
    /* of the blind-rotate totals above, the part run by the 8-wave form (launches of at most one
     * workgroup per CU); the rest is the 4-wave kernel */
    uint64_t br8_launches;
    uint64_t br8_rotations;
    double   ms_blind_rotate8;
    /* with kernel timing on: shader cycles (s_memtime) and 100 MHz reference ticks (s_memrealtime) that every 61st workgroup
     * of every blind-rotate launch of the flushes lived for; 0.1 * cycles / ticks = the shader clock in GHz the
     * timed launches ran at (a cold chip runs them at ~2.0 GHz, a warm one at ~2.37) */
    uint64_t clk_shader_cycles;
    uint64_t clk_ref_ticks;
    /* recorded gates (and NOTs) dropped at a flush because nothing could ever observe their result: no sample
     * handle held it and no live gate read it ("eliminate_dead") */
    uint64_t dead_gates;
    /* gates answered WITHOUT a bootstrap because an operand was a public constant (a trivial sample): "fold_constants",
     * opt-in; a MUX turned into a two-input gate counts once */
    uint64_t folded_gates;
} TfheHipStats;
void tfhe_hip_get_stats(TfheHipStats *out);
void tfhe_hip_reset_stats(void);
/* when on, every kernel launch is bracketed by HIP events, read back after the flush */
void tfhe_hip_set_kernel_timing(int on);

/* ---- host-logic test entry: does blind-rotate kernel form `form` (0 = 4 waves, 1 = split, 2 = 8 waves, 3 = 2 waves)
 * keep its magnitude bounds for gadget (l, Bgbit) at ring size N with digit-table mode
 * `tables` (0, 1, 2 as "br_digit_table")?  A key is refused at upload when no form does; a launch falls back to an
 * admissible form (peba1_amd/csrc/br_forms.hpp).  Returns 1 or 0. ---- */
int tfhe_hip_test_form_admissible(int form, int32_t N, int32_t l, int32_t Bgbit, int tables);

/* ---- test entry: device allocations of key images, the ciphertext slot pool and the per-flush scratch that would bring their total
 * above `bytes` fail as if the card were full (0 = no cap).  Running out of device memory there is RECOVERABLE: the call
 * that needed the memory has no effect (recorded gates stay recorded, tfhe_hip_flush returns -1), tfhe_hip_last_error()
 * says what could not be allocated, and the caller may free ciphertext arrays and carry on. ---- */
void tfhe_hip_test_set_alloc_cap(int64_t bytes);

/* ---- host-logic test entry: levelise a DAG given as count x {kind, dst, a, b, c} slot
 * records (kind: gate code 0..9, 16 = MUX, 17 = NOT; absent operands -1) without
 * touching the device; writes the level of each op, returns the depth ---- */
int tfhe_hip_test_schedule(const int32_t *ops5, int32_t count, int32_t unit, int32_t balance, int32_t *levels_out);
/* Diagnostic (tools/wg_times.py): a 4-wave blind-rotate launch of `width` random gates (the second of two back to back);
 * times4[4i .. 4i+3] = s_memtime (shader cycles; the start stamp carries the XCC / CU id in its top 16
 * bits) at the start and end of workgroup i, then s_memrealtime (constant 100 MHz) at its start and
 * end; *launch_ms = the launch's duration between two stream events. */
int tfhe_hip_test_wg_times(const TFheGateBootstrappingCloudKeySet *bk, int32_t width, uint64_t *times4, double *launch_ms);

/* ---- kernel-level entry points (K2/K3 parity tests against the oracle) ---- */
/* exact negacyclic products res[c] = ip[c] * tp[c] mod (X^N+1) mod 2^32 through
 * the device NTT (two 27-bit primes + CRT); |ip| must be < 2^12 */
int tfhe_hip_kernel_negacyclic(const TFheGateBootstrappingCloudKeySet *bk, const int32_t *ip,
                               const Torus32 *tp, Torus32 *res, int32_t count);
/* modswitch + blind rotate + extract of `count` linear combinations lin[c]
 * (n+1 words each): u_out[c] (kN+1 words) and, if acc_out != NULL, the raw
 * accumulator ((k+1)N words) */
int tfhe_hip_kernel_bootstrap_woks(const TFheGateBootstrappingCloudKeySet *bk, const Torus32 *lin,
                                   int32_t count, Torus32 *u_out, Torus32 *acc_out);
/* key switch of `count` extracted samples u[c] (kN+1 words) -> out[c] (n+1 words) */
int tfhe_hip_kernel_keyswitch(const TFheGateBootstrappingCloudKeySet *bk, const Torus32 *u,
                              int32_t count, Torus32 *out);

#ifdef __cplusplus
}
#endif
#endif
