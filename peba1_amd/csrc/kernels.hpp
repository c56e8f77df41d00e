// kernels.hpp -- host-visible declarations of the HIP kernels' launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfhe_hip {

// Parameters of one key as the kernels see them.
struct DevParams {
    int32_t n, N, k, l, Bgbit, ks_t, ks_basebit;
    int32_t kpl;            // (k+1) l
    int32_t ct_stride;      // words per ciphertext slot: n+1 rounded up to 4
    int32_t u_stride;       // words per extracted sample: kN+1 rounded up to 4
    uint32_t decomp_offset; // sum_j (Bg/2) 2^{32-j Bgbit}
    uint32_t ks_prec_offset;// 2^{32-(1+basebit t)}
    int32_t mu;             // test-vector amplitude, 1/8
    int32_t digit_table;    // 1 = digit products of the first NTT step from an LDS table where Bgbit allows (kernels.hip)
    int32_t br_variant;     // 2 = split transforms: only read by the negacyclic test launcher
    unsigned long long *clock_acc;  // kernel timing: [2] running sums of shader cycles (s_memtime) and of 100 MHz ticks
                                    // (s_memrealtime) over every 61st workgroup of every blind-rotate launch -> the shader
                                    // clock the launches ran at; or null
    unsigned long long *wg_times;   // diagnostic: [4 * grid] s_memtime (shader cycles) at workgroup start and end, then
                                    // s_memrealtime (100 MHz) at start and end; or null
};

// Device-resident evaluation key.
struct DevKey {
    const uint32_t *bk_img;  // [n][kpl][prime 2][w 2][N] NTT image, Montgomery form, x N^-1
    const int32_t *ksk;      // [kN][t][base-1][ct_stride]
    const int32_t *ksk_zero; // one more row of ct_stride zeros (digit 0 of the key switch)
    const uint32_t *tw;      // [prime 2][fwd, inv][N] twiddles, Montgomery form, then the radix-4 quads
                             // [prime 2][fwd, inv][N/2][4] = {w2, w3, w1 w2, P - w1 w3} (ntt_wave.hpp): 12N words;
                             // then the same for the two half transforms of the split kernels, 6N words each
};

// One blind rotation: t = (0, c0) + sa * slot_a + sb * slot_b, then
// modswitch, blind rotate, sample extract into u_buf[u_index].
struct RotDesc {
    int32_t slot_a, slot_b;
    int32_t sa, sb;
    int32_t c0;
    int32_t u_index;
};

// One key switch: (u_buf[u0] (+ u_buf[u1]) + (0, add_b)) -> pool[dst_slot].
struct KsDesc {
    int32_t u0, u1;   // u1 = -1 when absent
    int32_t add_b;
    int32_t dst_slot;
};

struct NotDesc { int32_t src_slot, dst_slot; };


void launch_bk_transform(hipStream_t s, const DevParams &p, const int32_t *raw_polys, uint32_t *img,
                         const uint32_t *tw, int npoly_per_w, int nw, const uint32_t scale[2]);
// 2-wave form (N = 1024): the admissibility fallback (br_forms.hpp BR_FORM_WAVE2)
void launch_blind_rotate2(hipStream_t s, const DevParams &p, const DevKey &key, const int32_t *pool,
                          const RotDesc *rots, int count, int32_t *u_buf, int32_t *acc_dbg);
// 4-wave form (N = 1024): the streaming form, two workgroups per CU
void launch_blind_rotate4(hipStream_t s, const DevParams &p, const DevKey &key, const int32_t *pool,
                          const RotDesc *rots, int count, int32_t *u_buf, int32_t *acc_dbg);
// 8-wave form (N = 1024, l >= 2) for launches of at most one workgroup per CU: a second wave per SIMD
void launch_blind_rotate8(hipStream_t s, const DevParams &p, const DevKey &key, const int32_t *pool,
                          const RotDesc *rots, int count, int32_t *u_buf, int32_t *acc_dbg);
// split form (8 waves per rotation, every transform as two half-size ones; N = 1024 or 2048)
void launch_blind_rotate_split(hipStream_t s, const DevParams &p, const DevKey &key, const int32_t *pool,
                               const RotDesc *rots, int count, int32_t *u_buf, int32_t *acc_dbg);
// splits > 1: each gate's key switch is cut into `splits` ranges of input coefficients
// (partial sums in `partial[count][splits][ct_stride]`, then a reduce launch).  tile = 16, 24 or
// 32: launches of at least 2*tile gates use a tiled kernel (one pass over the KSK rows of
// a range serves `tile` gates); 0 = always one workgroup per (gate, range).  index: the tiled launch is the
// index form (rows in pinned registers picked through the VGPR index mode: keyswitch_index_kernel), else the
// LDS-strip form (tile 16 only: keyswitch_strip_kernel)
void launch_keyswitch(hipStream_t s, const DevParams &p, const DevKey &key, const int32_t *u_buf,
                      const KsDesc *descs, int count, int32_t *pool, int splits, int32_t *partial, int tile,
                      bool index = true);
void launch_not(hipStream_t s, const DevParams &p, const NotDesc *descs, int count, int32_t *pool);
// res[c] = ip[c] * (poly whose image is img[c]) through the device NTT
void launch_negacyclic(hipStream_t s, const DevParams &p, const uint32_t *tw, const int32_t *ip,
                       const uint32_t *img, int32_t *res, int count);


void launch_gather_slots(hipStream_t s, const int32_t *pool, int stride, int words, const int32_t *slots, int count,
                         int32_t *packed);
void launch_scatter_slots(hipStream_t s, int32_t *pool, int stride, int words, const int32_t *slots, int count,
                          const int32_t *packed);

}  // namespace tfhe_hip
