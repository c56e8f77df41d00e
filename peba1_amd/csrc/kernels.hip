// kernels.hip -- hand-written gfx950 kernels of the bootstrapped-gate path.
//
//   K6    bk_transform_kernel     Torus32 bootstrapping key -> NTT image (once per key)
//   K1+K2 blind_rotate4_kernel    gate prelude, modulus switch, blind rotate (n external
//                                 products), sample extract; four wave64 per rotation (N = 1024: the streaming form)
//         blind_rotate8_kernel    eight wave64 per rotation (two per prime and input polynomial) for launches of
//                                 at most one workgroup per CU: narrow levels, single gates (N = 1024)
//         blind_rotate_split_kernel  eight wave64 per rotation, every transform as two half-size ones
//                                 (N = 2048; at N = 1024 the form with the widest admissible gadget range but one)
//         blind_rotate2_kernel    two wave64 per rotation, no tables, one reduction per step: the form whose lazy
//                                 bounds admit the widest gadgets (br_forms.hpp) -- the admissibility fallback (N = 1024)
//   K3/K4 keyswitch_index_kernel  (u0 [+ u1] + const) -> LWE sample under the gate key (default for wide launches): one
//                                 pass over the KSK rows of a coefficient range serves a tile of 16 / 24 / 32 gates; a
//                                 thread's column of the staged rows sits in pinned VGPRs picked through the VGPR index
//                                 mode by the wave-uniform digit (statements generated: ks_index_asm.inc)
//         keyswitch_strip_kernel  the same tile with the staged rows in thread-private LDS strips: plain HIP source, no
//                                 pinned registers (tuning "ks_index" 0)
//         keyswitch_kernel        per-gate form for narrow launches; ks_reduce_kernel adds the
//                                 partial sums of the ranges
//   K5    not_kernel              negation
//         gather/scatter_slots    packed words <-> ciphertext pool (import, export, collectives)
//
// Restates (does not translate) tfhe's tfhe_bootstrap_woKS_FFT / tfhe_blindRotate_FFT
// / tGswFFTExternMulToTLwe / lweKeySwitch as described in SURVEY.md Appendix A.3;
// reference call sites: /root/reference/src/Math.cpp:34-43 (every bootsXOR/bootsAND).
//
// All ring kernels are templates on LOGN (N = 1024: TFHE's parameter sets, N = 2048: BASELINE
// configs[4]); 16 or 32 coefficients per lane (8 or 16 in the split form).  The accumulator (2 x N Torus32) lives in LDS
// for the whole n-step loop.
// 4-wave form: wave (q, u) works modulo prime q on input polynomial u -- per step the rotated
// accumulator, gadget digits, l forward NTTs (ntt_wave.hpp), 64-bit multiply-accumulate
// against the streamed key image (v_mad_i64_i32; 16-byte-per-lane coalesced loads, 1 KiB per
// wave instruction), partial sum of the other output polynomial to wave (q, 1-u) through LDS,
// one inverse NTT, half of the residues swapped with wave (1-q, u), CRT.  Three workgroup
// barriers per step.
// 2-wave form: wave q does all arithmetic modulo prime q (2l forward and 2 inverse NTTs per
// step), swaps one residue polynomial with its partner and CRT-recombines the one it owns.
// Forms measured slower everywhere and removed in round 6 (HISTORY.md): the lean 4-wave forms (three workgroups per
// CU at N = 1024; the only 4-wave form that fitted N = 2048), the time-sliced issue priority, the W = 2 / atomic /
// unpipelined LDS-strip key switches and the scalar-branch register key switch.
#include <type_traits>

#include "kernels.hpp"
#include "ntt_wave.hpp"

namespace tfhe_hip {

namespace {

__device__ __forceinline__ PrimeCtx make_ctx(int q, const uint32_t *tw, int n_ring) {
    PrimeCtx c;
    c.P = q ? NTT_P1 : NTT_P0;
    c.pinv = q ? NTT_PINV1 : NTT_PINV0;
    c.rmod = q ? NTT_R[1] : NTT_R[0];
    c.wf = tw + (size_t)(q * 2 + 0) * n_ring;
    c.wi = tw + (size_t)(q * 2 + 1) * n_ring;
    // radix-4 quads follow the four radix-2 tables: [prime][fwd, inv][N/2] x 16 bytes
    const uint4 *quads = reinterpret_cast<const uint4 *>(tw + (size_t)4 * n_ring);
    c.qf = quads + (size_t)(q * 2 + 0) * (n_ring / 2);
    c.qi = quads + (size_t)(q * 2 + 1) * (n_ring / 2);
    c.dtab = nullptr;
    c.fw1 = nullptr;
    c.fw2 = nullptr;
    return c;
}

// context of half h of a split transform (engine.cpp make_twiddles: the sub-transform tables follow
// the 12N words of the full-size ones, 6N words per half, laid out like an N/2-point table)
__device__ __forceinline__ PrimeCtx make_sub_ctx(int q, int h, const uint32_t *tw, int n_ring) {
    return make_ctx(q, tw + (size_t)12 * n_ring + (size_t)h * 6 * n_ring, n_ring / 2);
}

// acc64 (sum of <= 6 products x*bk, |x| < 11.1P, 0 <= bk < P, so |acc| < 2^61) ->
// Montgomery reduction (|.| < 2.6P), inverse NTT: signed residue, |y| < P (the CRT takes it as it is)
template <int LOGN>
__device__ __forceinline__ void finish_inverse(const int64_t (&acc)[WaveNtt<LOGN>::REGS],
                                               int32_t (&y)[WaveNtt<LOGN>::REGS],
                                               const PrimeCtx &c, uint32_t *scr, int lane) {
    constexpr int REGS = WaveNtt<LOGN>::REGS;
#pragma unroll
    for (int r = 0; r < REGS; ++r) y[r] = mont_redc(acc[r], c.P, c.pinv);
    WaveNtt<LOGN>::inverse(y, c, scr, lane);
}

// ---------------------------------------------------------------------------
// K6: Torus32 polynomials -> NTT image.  grid (npoly, 2 primes), one wave each.
// raw layout [X][w][N]; image layout [X][q][w][N] with word (g*256 + lane*4 + e)
// holding register 4g+e of `lane` in layout L2, i.e. exactly what a lane of the
// blind-rotate kernel fetches with one 16-byte load.
// ---------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(64) void bk_transform_kernel(const int32_t *__restrict__ raw, uint32_t *__restrict__ img,
                                                          const uint32_t *__restrict__ tw, int nw,
                                                          uint32_t scale0, uint32_t scale1) {
    using NTT = WaveNtt<LOGN>;
    constexpr int N = NTT::N, REGS = NTT::REGS;
    __shared__ __align__(16) uint32_t scr[NTT::SCRATCH_WORDS];
    const int lane = threadIdx.x;
    const int q = blockIdx.y;
    const int poly = blockIdx.x;          // X*nw + w
    const int X = poly / nw, w = poly % nw;
    const PrimeCtx c = make_ctx(q, tw, N);
    const uint32_t scale = q ? scale1 : scale0;
    const int32_t *src = raw + (size_t)poly * N;
    int32_t x[REGS];
#pragma unroll
    for (int r = 0; r < REGS; ++r) x[r] = src[r * 64 + lane] % (int32_t)c.P;   // |x| < P
    NTT::forward(x, c, scr, lane);                                             // |x| < 12P
    uint4 *dst = reinterpret_cast<uint4 *>(img + ((size_t)(X * 2 + q) * nw + w) * N) + lane;
#pragma unroll
    for (int g = 0; g < REGS / 4; ++g) {
        uint32_t v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int32_t m = x[4 * g + e] % (int32_t)c.P;
            if (m < 0) m += (int32_t)c.P;
            v[e] = (uint32_t)((uint64_t)(uint32_t)m * scale % c.P);
        }
        dst[g * 64] = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

// ---------------------------------------------------------------------------
// test kernel: res = ip * tp (negacyclic, mod 2^32), tp given as image
// ---------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(128) void negacyclic_kernel(const int32_t *__restrict__ ip, const uint32_t *__restrict__ img,
                                                         const uint32_t *__restrict__ tw, int32_t *__restrict__ res) {
    using NTT = WaveNtt<LOGN>;
    constexpr int N = NTT::N, REGS = NTT::REGS;
    __shared__ __align__(16) uint32_t lds_scr[2][NTT::SCRATCH_WORDS];
    const int tid = threadIdx.x;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const PrimeCtx c = make_ctx(q, tw, N);
    uint32_t *scr = lds_scr[q];
    const int32_t *src = ip + (size_t)blockIdx.x * N;
    int32_t x[REGS];
#pragma unroll
    for (int r = 0; r < REGS; ++r) x[r] = src[r * 64 + lane];
    NTT::forward(x, c, scr, lane);
    const uint4 *bp = reinterpret_cast<const uint4 *>(img + (size_t)(blockIdx.x * 2 + q) * N) + lane;
    int64_t acc[REGS];
#pragma unroll
    for (int g = 0; g < REGS / 4; ++g) {
        const uint4 b = bp[g * 64];
        acc[4 * g + 0] = (int64_t)x[4 * g + 0] * (int32_t)b.x;
        acc[4 * g + 1] = (int64_t)x[4 * g + 1] * (int32_t)b.y;
        acc[4 * g + 2] = (int64_t)x[4 * g + 2] * (int32_t)b.z;
        acc[4 * g + 3] = (int64_t)x[4 * g + 3] * (int32_t)b.w;
    }
    int32_t y[REGS];
    finish_inverse<LOGN>(acc, y, c, scr, lane);
#pragma unroll
    for (int r = 0; r < REGS; ++r) scr[r * 64 + lane] = (uint32_t)y[r];
    __syncthreads();
    if (q == 0) {
        const uint32_t *oscr = lds_scr[1];
#pragma unroll
        for (int r = 0; r < REGS; ++r)
            res[(size_t)blockIdx.x * N + r * 64 + lane] = (int32_t)crt_signed_to_torus(y[r], (int32_t)oscr[r * 64 + lane]);
    }
}

// ---------------------------------------------------------------------------
// shared pieces of the blind-rotate kernels
// ---------------------------------------------------------------------------
// K1: t = (0,c0) + sa*A + sb*B, modulus switch to Z_{2N} (tfhe modSwitchFromTorus32)
template <int LOGN, int THREADS>
__device__ __forceinline__ void prelude_modswitch(const DevParams &p, const RotDesc &rd, const int32_t *__restrict__ pool,
                                                  uint16_t *lds_bar, int tid) {
    const int32_t *A = pool + (size_t)rd.slot_a * p.ct_stride;
    const int32_t *B = pool + (size_t)rd.slot_b * p.ct_stride;
    for (int i = tid; i <= p.n; i += THREADS) {
        uint32_t t = (uint32_t)rd.sa * (uint32_t)A[i] + (uint32_t)rd.sb * (uint32_t)B[i];
        if (i == p.n) t += (uint32_t)rd.c0;
        lds_bar[i] = (uint16_t)((t + (1u << (30 - LOGN))) >> (31 - LOGN));     // round(t * 2N / 2^32) mod 2N
    }
}

// Shader clock of a launch (bench.py reports it beside the times: the same launch takes 15 % longer at the 2.0 GHz
// a cold chip runs than at the 2.37 GHz of a warm one): lane 0 of every 61st workgroup adds the
// shader cycles and the 100 MHz reference ticks it lived for to two running sums.
struct ClockProbe {
    unsigned long long c0 = 0, r0 = 0;
    bool on = false;
    __device__ __forceinline__ void begin(const DevParams &p) {
        on = p.clock_acc != nullptr && blockIdx.x % 61 == 0 && threadIdx.x == 0;    // a sample of workgroups across the launch
        if (on) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    }
    __device__ __forceinline__ void end(const DevParams &p) {
        if (on) {
            atomicAdd(&p.clock_acc[0], __builtin_amdgcn_s_memtime() - c0);
            atomicAdd(&p.clock_acc[1], __builtin_amdgcn_s_memrealtime() - r0);
        }
    }
};

// body polynomial of ACC = (0, X^{-barb} * (mu + mu X + ... + mu X^{N-1})), coefficient j
template <int LOGN>
__device__ __forceinline__ uint32_t testvector_coef(int j, int barb, int32_t mu) {
    constexpr int N = 1 << LOGN;
    const int idx = (j + barb) & (2 * N - 1);
    return (idx & N) ? (uint32_t)(-mu) : (uint32_t)mu;
}

// The accumulator in LDS.  Each polynomial is kept as three runs of N words, (acc, -acc, acc):
// coefficient (j - abar) mod 2N of the negacyclic rotation X^abar * ACC is then word
// ((lane - abar) mod 2N) + 64 r of that array -- one address per lane and step, the 64 r being the
// immediate offset of the LDS read: no wrap and no sign selection per coefficient.
template <int LOGN>
struct AccLds {
    static constexpr int N = 1 << LOGN;
    uint32_t w[2][3 * N];
    __device__ __forceinline__ void set(int u, int j, uint32_t v) {
        w[u][j] = v; w[u][N + j] = 0u - v; w[u][2 * N + j] = v;
    }
    __device__ __forceinline__ uint32_t get(int u, int j) const { return w[u][j]; }
    // D[r] = coefficient 64 r + lane of (X^abar - 1) ACC_u, offset added and digit tops flipped
    template <int REGS>
    __device__ __forceinline__ void rotated_difference(uint32_t (&D)[REGS], int u, int lane, int abar, uint32_t offset) const {
        const uint32_t base = (uint32_t)(lane - abar) & (uint32_t)(2 * N - 1);
        // -ACC comes from the negated run, so that rot - acc + offset is ONE three-operand add (v_add3_u32) instead of a
        // subtraction and an addition (round 4: 16 VALU instructions less per wave and step)
        const uint32_t *own_neg = w[u] + N + lane;
        const uint32_t *rot = w[u] + base;
#pragma unroll
        for (int r = 0; r < REGS; ++r) D[r] = (rot[r * 64] + own_neg[r * 64] + offset) ^ offset;
    }
};

// One input polynomial u of one blind-rotate step, modulo the wave's prime:
// D = (X^abar - 1) * ACC_u, its l gadget digits, forward NTT of each, and the
// multiply-accumulate against key rows u*l+jj for both output polynomials
// (acc0 <- output poly 0, acc1 <- output poly 1; exchanged when swap_outputs).  D and both 64-bit row sums stay in
// registers across the gadget rows; a pass's twiddles are loaded one transpose ahead.
// FRESH: acc0 / acc1 are written, not accumulated into -- the first row multiplies, the others
// multiply-add (no zeroing of 4*REGS registers per step); otherwise every row accumulates.
// Digits (tfhe tGswTorus32PolynomialDecompH): digit_jj = ((D + offset) >> s_jj) & (Bg-1)) - Bg/2.  The
// offset holds Bg/2 at every digit position, so XOR-ing it back flips the top bit of every digit
// field, and a digit is then the SIGNED bit field of (D + offset) ^ offset: one v_bfe_i32.
// t0: the twiddles of the transforms' first pass -- lane-uniform and the same for every row and step, so the caller
// loads them ONCE per kernel (they then live in scalar registers; round 3: -1.4 % of a wide launch against loading
// them per row, and 12 fewer vector registers)
// PROGRESS (4-wave form): the wave's issue priority follows its progress through the step -- 0 for the first gadget row,
// 1 for the second, 2 from the third (blind_rotate4_body raises it to 3 for the step's tail).
template <int LOGN, bool FRESH, bool TABLE, bool LDSTW = false, bool PROGRESS = false>
__device__ __forceinline__ void forward_poly(const DevParams &p, const DevKey &key, const PrimeCtx &c,
                                             const AccLds<LOGN> &lds_acc, uint32_t *scr,
                                             int lane, int q, int i, int u, int abar, bool swap_outputs,
                                             int64_t (&acc0)[WaveNtt<LOGN>::REGS], int64_t (&acc1)[WaveNtt<LOGN>::REGS],
                                             const typename WaveNtt<LOGN>::FwdTw0 &t0, int jbegin = 0, int jend = -1) {
    using NTT = WaveNtt<LOGN>;
    constexpr int N = NTT::N, REGS = NTT::REGS, G4 = REGS / 4;
    uint32_t Dk[REGS];
    lds_acc.template rotated_difference<REGS>(Dk, u, lane, abar, p.decomp_offset);
    const int o0 = swap_outputs ? N / 4 : 0, o1 = N / 4 - o0;          // uint4 offset of output poly 0 / 1
    const int width = p.Bgbit;
    auto row = [&](int jj, auto first) {
        const int prow = u * p.l + jj;
        const uint4 *bp = reinterpret_cast<const uint4 *>(
                              key.bk_img + ((size_t)((size_t)i * p.kpl + prow) * 2 + q) * 2 * N) + lane;
        // key rows: both output polynomials are requested before the transform so that their latency hides under it
        uint4 b0[G4], b1[G4];
#pragma unroll
        for (int g = 0; g < G4; ++g) b0[g] = bp[o0 + g * 64];
#pragma unroll
        for (int g = 0; g < G4; ++g) b1[g] = bp[o1 + g * 64];
        const int shift = 32 - (jj + 1) * width;
        int32_t x[REGS];
        NTT::template forward_digits<TABLE, LDSTW>(x, Dk, shift, width, c, scr, lane, t0);
#pragma unroll
        for (int g = 0; g < G4; ++g) {
            const int32_t bb0[4] = {(int32_t)b0[g].x, (int32_t)b0[g].y, (int32_t)b0[g].z, (int32_t)b0[g].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                if constexpr (decltype(first)::value) acc0[r] = (int64_t)x[r] * bb0[e];
                else acc0[r] += (int64_t)x[r] * bb0[e];
            }
        }
#pragma unroll
        for (int g = 0; g < G4; ++g) {
            const int32_t bb1[4] = {(int32_t)b1[g].x, (int32_t)b1[g].y, (int32_t)b1[g].z, (int32_t)b1[g].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                if constexpr (decltype(first)::value) acc1[r] = (int64_t)x[r] * bb1[e];
                else acc1[r] += (int64_t)x[r] * bb1[e];
            }
        }
    };
    // gadget rows [jbegin, jend) of this input polynomial (default: all l of them)
    if (jend < 0) jend = p.l;
    row(jbegin, std::integral_constant<bool, FRESH>{});
#pragma unroll 1
    for (int jj = jbegin + 1; jj < jend; ++jj) {
        if constexpr (PROGRESS) {
            if (jj == jbegin + 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(2);
        }
        row(jj, std::false_type{});
    }
}

// sample extract at index 0 (tfhe tLweExtractLweSampleIndex) + optional raw accumulator dump
template <int LOGN, int THREADS, typename AccT>
__device__ __forceinline__ void extract_sample(const DevParams &p, const RotDesc &rd,
                                               const AccT &acc,
                                               int32_t *__restrict__ u_buf, int32_t *__restrict__ acc_dbg, int tid) {
    constexpr int N = 1 << LOGN;
    int32_t *u = u_buf + (size_t)rd.u_index * p.u_stride;
    for (int j = tid; j < N; j += THREADS)
        u[j] = (int32_t)(j == 0 ? acc.get(0, 0) : 0u - acc.get(0, N - j));
    if (tid == 0) u[N] = (int32_t)acc.get(1, 0);
    if (acc_dbg) {
        int32_t *d = acc_dbg + (size_t)blockIdx.x * 2 * N;
        for (int j = tid; j < 2 * N; j += THREADS) d[j] = (int32_t)acc.get(j >> LOGN, j & (N - 1));
    }
}

// ---------------------------------------------------------------------------
// K1+K2: blind rotate, 2-wave form.  grid = rotations, 128 threads: wave q works
// modulo prime q and handles both input polynomials (2l forward + 2 inverse NTTs
// per step).  Slower than the 4-wave form at every launch width; kept because all 2l rows of an output polynomial
// are summed in 64 bits and reduced ONCE, without tables: its bounds admit gadgets no other form does (br_forms.hpp
// BR_FORM_WAVE2, e.g. N = 1024, l = 9, Bg = 2^3) -- the last entry of the engine's fallback order, selectable with
// "br_variant" 4 (N = 1024 only).
// ---------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(128) void blind_rotate2_kernel(DevParams p, DevKey key, const int32_t *__restrict__ pool,
                                                           const RotDesc *__restrict__ rots,
                                                           int32_t *__restrict__ u_buf, int32_t *__restrict__ acc_dbg) {
    using NTT = WaveNtt<LOGN>;
    constexpr int N = NTT::N, REGS = NTT::REGS;
    __shared__ __align__(16) AccLds<LOGN> lds_acc;
    __shared__ __align__(16) uint32_t lds_scr[2][NTT::SCRATCH_WORDS];
    __shared__ uint16_t lds_bar[1024 + 8];

    const int tid = threadIdx.x;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const PrimeCtx c = make_ctx(q, key.tw, N);
    uint32_t *scr = lds_scr[q];
    const RotDesc rd = rots[blockIdx.x];
    const int n = p.n;

    prelude_modswitch<LOGN, 128>(p, rd, pool, lds_bar, tid);
    __syncthreads();
    {
        const int barb = lds_bar[n];
#pragma unroll
        for (int r = 0; r < REGS; ++r) {
            const int j = r * 64 + lane;
            lds_acc.set(q, j, q == 0 ? 0u : testvector_coef<LOGN>(j, barb, p.mu));
        }
    }
    __syncthreads();

    typename NTT::FwdTw0 t0;
    t0.load(c, lane);
    for (int i = 0; i < n; ++i) {
        const int abar = __builtin_amdgcn_readfirstlane((int)lds_bar[i]);
        if (abar == 0) continue;                            // tfhe_blindRotate_FFT skips these too

        int64_t acc0[REGS], acc1[REGS];
        forward_poly<LOGN, true, false>(p, key, c, lds_acc, scr, lane, q, i, 0, abar, false, acc0, acc1, t0);
        forward_poly<LOGN, false, false>(p, key, c, lds_acc, scr, lane, q, i, 1, abar, false, acc0, acc1, t0);

        int32_t y0[REGS], y1[REGS];
        finish_inverse<LOGN>(acc0, y0, c, scr, lane);
        finish_inverse<LOGN>(acc1, y1, c, scr, lane);

        // wave q owns output polynomial q: send the other one's residues across
        const uint32_t *oscr = lds_scr[1 - q];
        if (q == 0) {
#pragma unroll
            for (int r = 0; r < REGS; ++r) scr[r * 64 + lane] = (uint32_t)y1[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < REGS; ++r)
                lds_acc.set(0, r * 64 + lane, lds_acc.get(0, r * 64 + lane) + crt_signed_to_torus(y0[r], (int32_t)oscr[r * 64 + lane]));
        } else {
#pragma unroll
            for (int r = 0; r < REGS; ++r) scr[r * 64 + lane] = (uint32_t)y0[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < REGS; ++r)
                lds_acc.set(1, r * 64 + lane, lds_acc.get(1, r * 64 + lane) + crt_signed_to_torus((int32_t)oscr[r * 64 + lane], y1[r]));
        }
        __syncthreads();
    }
    extract_sample<LOGN, 128>(p, rd, lds_acc, u_buf, acc_dbg, tid);
}

#ifdef TFHE_HIP_STAMPS
// Diagnostic build only (tools/diag/build_stamps.sh): per-phase shader-cycle sums of
// the 4-wave kernel, lane 0 of each wave, accumulated into a buffer nothing else reads.
__device__ unsigned long long g_stamps[8][8];      // [wave][phase]; the 4-wave kernel uses waves 0-3
#define STAMP_DECL unsigned long long st_prev = __builtin_amdgcn_s_memtime(), st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(k)                                                             \
    do {                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                   \
        const unsigned long long st_now = __builtin_amdgcn_s_memtime();      \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                  \
        st_sum[k] += st_now - st_prev;                                       \
        st_prev = st_now;                                                    \
        __builtin_amdgcn_sched_barrier(0);                                   \
    } while (0)
#define STAMP_FLUSH                                                          \
    do {                                                                     \
        if (lane == 0)                                                       \
            for (int k = 0; k < 8; ++k) atomicAdd(&g_stamps[wv][k], st_sum[k]); \
    } while (0)
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH
#endif

// ---------------------------------------------------------------------------
// K1+K2: blind rotate, 4-wave form (the default).  grid = rotations, 256 threads:
// wave (q,u) works modulo prime q on input polynomial u (l forward NTTs), hands
// the partial sum of the other output polynomial to wave (q,1-u), runs ONE
// inverse NTT for output polynomial u, and shares the CRT with wave (1-q,u).
// Three workgroup barriers per step.  N = 1024 only (241 VGPRs, two workgroups per CU): D and both 64-bit row sums in
// registers, a pass's twiddles loaded one transpose ahead, three exchange buffers per wave.  (N = 2048 runs the split form:
// 32 coefficients per lane do not fit the register file without reducing both sums per row, 196 against 142 ms per 4,096.)
// ---------------------------------------------------------------------------
// LDS of one 4-wave workgroup (81,428 B: two per CU).  Each wave has two exchange areas: the scratch of its NTT
// transposes (private while a transform runs; from the end of its inverse transform to the step's last barrier it carries
// the residues of the half its CRT partner recombines -- nobody else touches it in between) and the partial sums it sends
// to the wave of the other input polynomial.
template <int LOGN>
struct Br4Lds {
    using NTT = WaveNtt<LOGN>;
    AccLds<LOGN> acc;                              // the accumulator (signed runs), resident for all n steps
    uint32_t buf0[4][NTT::SCRATCH_WORDS];          // per wave: transpose scratch (layouts R and H of ntt_wave.hpp), then residues
    uint32_t buf1[4][NTT::ROW_WORDS];              // per wave: the partial sums sent (rows only: layout R)
    uint16_t bar[1024 + 8];                        // modulus-switched mask and body
    alignas(8) uint32_t dtab[2][5 * DIGIT_TAB];    // per prime: first-step products of the gadget digits (ntt_wave.hpp)
    // per prime: the forward transforms' second-pass twiddles (one image per group of 2^LC lanes) and third-pass
    // twiddles (one per lane)
    uint4 ft1[2][64 >> NTT::LC][NTT::FwdTw1::IMAGE16];
    uint4 ft2[2][64][NTT::FwdTw2::IMAGE16];
    __device__ __forceinline__ uint32_t *scr(int wv) { return buf0[wv]; }
    __device__ __forceinline__ uint32_t *x1(int wv) { return buf1[wv]; }
    __device__ __forceinline__ uint32_t *x2(int wv) { return buf0[wv]; }
};
static_assert(sizeof(Br4Lds<10>) <= 80 * 1024, "two workgroups of the 4-wave form must fit the 160 KB of a CU");

// prelude + modulus switch + the n-step blind rotation of one descriptor; the result is left
// in sh.acc (complete for every thread on return)
// TAB: the first radix-4 step of every forward transform reads digit products from an LDS table
// (gadget digits of at most DIGIT_TAB_BITS bits: every built-in set but the legacy Bg = 2^10 one)
template <int LOGN, bool TAB>
__device__ __forceinline__ void blind_rotate4_body(const DevParams &p, const DevKey &key,
                                                   const int32_t *__restrict__ pool, const RotDesc &rd,
                                                   Br4Lds<LOGN> &sh, int tid) {
    using NTT = WaveNtt<LOGN>;
    constexpr int N = NTT::N, REGS = NTT::REGS, HALF = REGS / 2;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wv & 1, u = wv >> 1;
    const int lane = tid & 63;
    PrimeCtx c = make_ctx(q, key.tw, N);
    uint32_t *scr = sh.scr(wv);
    const int n = p.n;

    prelude_modswitch<LOGN, 256>(p, rd, pool, sh.bar, tid);
    if constexpr (TAB) {
        // the two waves of prime q (tid bits 6 and 7 = q and u) fill that prime's table
        NTT::build_digit_table(sh.dtab[q], c, p.Bgbit, (u << 6) | lane, 128);
        c.dtab = sh.dtab[q];
    }
    // wave (q, 0) copies prime q's pass-2 / pass-3 twiddles of the forward transforms into LDS, once
    if (u == 0) {
        typename NTT::FwdTw1 a;
        a.load(c, lane);
        if ((lane & ((1 << NTT::LC) - 1)) == 0) a.to_image(sh.ft1[q][lane >> NTT::LC]);
        typename NTT::FwdTw2 b;
        b.load(c, lane);
        b.to_image(sh.ft2[q][lane]);
    }
    c.fw1 = sh.ft1[q][lane >> NTT::LC];
    c.fw2 = sh.ft2[q][lane];
    __syncthreads();
    if (q == 0) {
        const int barb = sh.bar[n];
#pragma unroll
        for (int r = 0; r < REGS; ++r) {
            const int j = r * 64 + lane;
            sh.acc.set(u, j, u == 0 ? 0u : testvector_coef<LOGN>(j, barb, p.mu));
        }
    }
    __syncthreads();

    STAMP_DECL;
    typename NTT::FwdTw0 t0;                 // first-pass twiddles of every forward transform: loaded once (forward_poly)
    t0.load(c, lane);

    // Issue priority by PROGRESS (round 5).  The workgroups that share a CU put one wave each on every SIMD, and the SIMD
    // arbitrates their issue by priority, then age: left alone, the older workgroup runs at the pace of a lone wave and the
    // younger gets the leftover slots, then finishes alone.  A wave's priority therefore follows its progress through the
    // step: 0 in the first gadget row, 1 in the second, 2 from the third, 3 from the first barrier to the last (the inverse
    // transform, the CRT, the accumulator update -- the part of a step in which four waves wait for each other three
    // times).  Whichever workgroup is further along wins, gets through its barriers at full pace, drops to 0 for its next
    // step and becomes the filler of the other's bubbles: they leapfrog step by step.  Measured, variants interleaved on one
    // box (profiles/r05_ab_kernel_variants.txt): 4,096 rotations 36.97 -> 35.41 ms (-4.2 %) against the time slices of
    // rounds 1-4 (removed in round 6).
    for (int i = 0; i < n; ++i) {
        const int abar = __builtin_amdgcn_readfirstlane((int)sh.bar[i]);
        if (abar == 0) continue;
        STAMP(0);

        // acc0 accumulates output poly u (kept), acc1 output poly 1-u (sent to wave (q,1-u)); both in 64 bits, reduced once
        int64_t acc0[REGS], acc1[REGS];
        forward_poly<LOGN, true, TAB, true, true>(p, key, c, sh.acc, scr, lane, q, i, u, abar, u != 0, acc0, acc1, t0);
        STAMP(1);

        int32_t t[REGS];
        {
            int32_t send[REGS];
#pragma unroll
            for (int r = 0; r < REGS; ++r) {
                t[r] = mont_redc(acc0[r], c.P, c.pinv);                 // l rows: |.| < 1.2P
                send[r] = mont_redc(acc1[r], c.P, c.pinv);
            }
            NTT::write_row(send, sh.x1(wv), lane);
        }
        // the first twiddles of the inverse transform are requested before the barrier and arrive while the wave waits
        typename NTT::InvTw2 t2;
        t2.load(c, lane);
        STAMP(2);
        lds_barrier();
        __builtin_amdgcn_s_setprio(3);                    // the step's tail (above: progress priority)
        STAMP(3);
        {
            int32_t other[REGS];
            NTT::read_row(other, sh.x1(wv ^ 2), lane);
#pragma unroll
            for (int r = 0; r < REGS; ++r) t[r] += other[r];       // |.| < 3.4P (the inverse takes < 4P)
        }
        NTT::inverse(t, c, scr, lane, t2);                         // signed residues, |t| < P: recombined as they are
        STAMP(4);

        // CRT of output poly u is split with wave (1-q,u): wave q recombines registers [q*HALF, (q+1)*HALF)
        // (the barrier in front of it orders only that pair)
        const uint32_t *ox = sh.x2(wv ^ 1);
        uint32_t *mx = sh.x2(wv);
        if (q == 0) {                                    // (two copies: register indices must be compile-time constants)
#pragma unroll
            for (int r = 0; r < HALF; ++r) mx[r * 64 + lane] = (uint32_t)t[HALF + r];
            LDS_BARRIER_ROLE("q=0");
#pragma unroll
            for (int r = 0; r < HALF; ++r)
                sh.acc.set(u, r * 64 + lane, sh.acc.get(u, r * 64 + lane) + crt_signed_to_torus(t[r], (int32_t)ox[r * 64 + lane]));
        } else {
#pragma unroll
            for (int r = 0; r < HALF; ++r) mx[r * 64 + lane] = (uint32_t)t[r];
            LDS_BARRIER_ROLE("q=1");
#pragma unroll
            for (int r = 0; r < HALF; ++r)
                sh.acc.set(u, (HALF + r) * 64 + lane,
                           sh.acc.get(u, (HALF + r) * 64 + lane) + crt_signed_to_torus((int32_t)ox[r * 64 + lane], t[HALF + r]));
        }
        STAMP(5);
        __builtin_amdgcn_s_setprio(0);
        lds_barrier();
        STAMP(6);
    }
    STAMP_FLUSH;
}

template <int LOGN, bool TAB>
__global__ __launch_bounds__(256, 2) void blind_rotate4_kernel(
    DevParams p, DevKey key, const int32_t *__restrict__ pool, const RotDesc *__restrict__ rots,
    int32_t *__restrict__ u_buf, int32_t *__restrict__ acc_dbg) {
    __shared__ __align__(16) Br4Lds<LOGN> sh;
    if (p.wg_times && threadIdx.x == 0) {
        // diagnostic (tools/wg_times.py): low 48 bits: shader clock; high 16 bits: XCC id and the CU / SH / SE fields of HW_ID
        const uint64_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        const uint64_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        p.wg_times[4 * blockIdx.x] = (__builtin_amdgcn_s_memtime() & 0xFFFFFFFFFFFFull) |
                                     ((((xcc & 0xF) << 8) | ((hw >> 8) & 0xFF)) << 48);
        p.wg_times[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();       // constant 100 MHz
    }
    ClockProbe clk;
    clk.begin(p);
    const RotDesc rd = rots[blockIdx.x];
    blind_rotate4_body<LOGN, TAB>(p, key, pool, rd, sh, threadIdx.x);
    extract_sample<LOGN, 256>(p, rd, sh.acc, u_buf, acc_dbg, threadIdx.x);
    clk.end(p);
    if (p.wg_times && threadIdx.x == 0) {
        p.wg_times[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime();
        p.wg_times[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------
// K1+K2: blind rotate, 8-wave form for launches that cannot fill the chip (at most one
// workgroup per CU).  With four waves a CU runs one wave per SIMD, and a lone wave issues its
// multiplier-class instructions at half the rate two waves share (v_mad_i64_i32: 10 cycles
// against 5; profiles/archive/r01_valu_rates.txt).  Here a second wave sits on each SIMD: for prime q and
// input polynomial u, wave A = (q,u,0) transforms and multiplies the gadget rows 0..l-2, wave
// B = (q,u,1) the last row; all four partial sums of an output polynomial (A's and B's own, and the
// two the waves of the other input polynomial send) go through LDS.  The inverse transform is then SPLIT
// over the pair (round 3): a Cooley-Tukey spectrum is two independent half-size spectra, so A takes the
// lower half of the summed spectrum and B the upper half, each runs an N/2-point inverse transform
// (WaveNtt<LOGN-1> with the sub-tree twiddles of the split kernel), and the stage-0 butterflies, the CRT
// and the accumulator update are shared by the four waves of the output polynomial exactly as in the
// split kernel (split_finish).  In round 2 only B ran a (full-size) inverse while A idled: a lone wave
// issues multiplier-class instructions at half the rate two waves share, and the inverse phase was 40 %
// of a step.  This is the form for narrow levels, single gates (immediate mode) and the short circuits
// between two decryptions that the reference's own test program is made of.  Same integers as the
// other forms.  N = 1024, l >= 2; three workgroup barriers per step.
// ---------------------------------------------------------------------------
// last inverse stage on half-transform outputs (a0, a1 modulo P0; b0, b1 modulo P1, each below P in magnitude) of one
// coefficient pair, then the CRT on the signed residues (|.| < 2P): the Torus32 increment of coefficient j (h = 0)
// or j + N/2 (h = 1)
__device__ __forceinline__ uint32_t split_finish(int h, int32_t a0, int32_t a1, int32_t b0, int32_t b1, uint32_t iw1_0,
                                                 uint32_t iw1_1) {
    if (h == 0) return crt_signed_to_torus(a0 + a1, b0 + b1);
    return crt_signed_to_torus(mont_mul(a0 - a1, iw1_0, NTT_P0, NTT_PINV0), mont_mul(b0 - b1, iw1_1, NTT_P1, NTT_PINV1));
}

// (round 5: wave B raising its issue priority -- in the step's tail before the half inverse, behind it, inside it, for the last
// phase only, or for the reductions in front of the first barrier, where the phase stamps show it 700 cycles behind wave A --
// measured slower in all eight combinations, 2.80-2.84 against 2.76 ms per rotation; profiles/r05_ab_kernel_variants.txt,
// profiles/r05_stamps_8wave.txt)
template <int LOGN>
struct Br8Lds {
    using NTT = WaveNtt<LOGN>;
    AccLds<LOGN> acc;
    // wave-private NTT transposes: full-size forward transforms (layout R), half-size inverse (layout H: smaller)
    static_assert(WaveNtt<LOGN - 1>::SCRATCH_WORDS <= NTT::ROW_WORDS, "the half transform's scratch fits the full-size rows");
    uint32_t scr[8][NTT::ROW_WORDS];
    uint32_t pa0[4][NTT::ROW_WORDS];               // A's sum for its own output polynomial (read by its B)
    uint32_t pa1[4][NTT::ROW_WORDS];               // A's sum for the other output polynomial (read by the other B)
    uint32_t pb0[4][NTT::ROW_WORDS];               // B's sum for its own output polynomial
    uint32_t pb[4][NTT::ROW_WORDS];                // B's sum for the other output polynomial
    uint16_t bar[1024 + 8];
    alignas(8) uint32_t dtab[2][5 * DIGIT_TAB];
    // per prime: LDS copies of the forward transforms' second- and third-pass twiddles (as in Br4Lds)
    uint4 ft1[2][64 >> NTT::LC][NTT::FwdTw1::IMAGE16];
    uint4 ft2[2][64][NTT::FwdTw2::IMAGE16];
};

template <int LOGN, bool TAB>
__global__ __launch_bounds__(512, 2) void blind_rotate8_kernel(
    DevParams p, DevKey key, const int32_t *__restrict__ pool, const RotDesc *__restrict__ rots,
    int32_t *__restrict__ u_buf, int32_t *__restrict__ acc_dbg) {
    using NTT = WaveNtt<LOGN>;
    using SUB = WaveNtt<LOGN - 1>;
    constexpr int N = NTT::N, M = N / 2, REGS = NTT::REGS, RS = SUB::REGS, QUARTER = RS / 4;
    __shared__ __align__(16) Br8Lds<LOGN> sh;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int base = wv & 3, q = base & 1, u = base >> 1;
    const bool role_b = wv >= 4;
    const int h = wv >> 2;                              // the half of every inverse transform this wave runs
    const int lane = tid & 63;
    PrimeCtx c = make_ctx(q, key.tw, N);
    const PrimeCtx ch = make_sub_ctx(q, h, key.tw, N);  // twiddles of half h (engine.cpp make_twiddles)
    const uint32_t iw1_0 = key.tw[(size_t)1 * N + 1], iw1_1 = key.tw[(size_t)3 * N + 1];   // inverse stage 0, both primes
    uint32_t *scr = sh.scr[wv];
    const int n = p.n;
    const RotDesc rd = rots[blockIdx.x];
    ClockProbe clk;
    clk.begin(p);

    prelude_modswitch<LOGN, 512>(p, rd, pool, sh.bar, tid);
    if constexpr (TAB) {
        // the four waves of prime q fill that prime's table
        NTT::build_digit_table(sh.dtab[q], c, p.Bgbit, ((wv >> 1) << 6) | lane, 256);
        c.dtab = sh.dtab[q];
    }
    if (wv < 2) {                                    // waves 0 and 1 = (q, u = 0, A): prime q's twiddles into LDS, once
        typename NTT::FwdTw1 a;
        a.load(c, lane);
        if ((lane & ((1 << NTT::LC) - 1)) == 0) a.to_image(sh.ft1[q][lane >> NTT::LC]);
        typename NTT::FwdTw2 b;
        b.load(c, lane);
        b.to_image(sh.ft2[q][lane]);
    }
    c.fw1 = sh.ft1[q][lane >> NTT::LC];
    c.fw2 = sh.ft2[q][lane];
    __syncthreads();
    if (q == 0 && !role_b) {
        const int barb = sh.bar[n];
#pragma unroll
        for (int r = 0; r < REGS; ++r) {
            const int j = r * 64 + lane;
            sh.acc.set(u, j, u == 0 ? 0u : testvector_coef<LOGN>(j, barb, p.mu));
        }
    }
    __syncthreads();

    // (Round 5 built the balanced assignment -- rows [0, l/2) to A, [(l+1)/2, l) to B, the middle row of an odd l shared: A its
    // transform's first pass, B the rest, 940 / 1,010 vector instructions per step instead of 1,144 / 764 -- and measured it
    // SLOWER, 3.21 against 2.82 ms per rotation: the two waves of a SIMD are arbitrated by age, the older wave A runs at the
    // pace of a lone wave and B fills its bubbles, which 764 instructions just about do; a balanced pair leaves B to finish
    // alone.  profiles/r05_ab_kernel_variants.txt)
    const int last = p.l - 1;
    typename NTT::FwdTw0 t0;
    t0.load(c, lane);
    STAMP_DECL;
    // wave A -- the older wave of its SIMD and the one with more work -- also holds the higher issue priority for the whole
    // rotation: explicit priority instead of age alone measured 2.74 against 2.77 ms per rotation, 3.18 against 3.24 ms per
    // 256 (profiles/r05_ab_kernel_variants.txt); dropping it for the step's tail: 2.88
    if (!role_b) __builtin_amdgcn_s_setprio(1);
    for (int i = 0; i < n; ++i) {
        const int abar = __builtin_amdgcn_readfirstlane((int)sh.bar[i]);
        if (abar == 0) continue;
        STAMP(0);
        {
            int64_t acc0[REGS], acc1[REGS];             // output poly u, output poly 1-u
            if (!role_b)
                forward_poly<LOGN, true, TAB, true>(p, key, c, sh.acc, scr, lane, q, i, u, abar, u != 0,
                                                                                acc0, acc1, t0, 0, last);
            else
                forward_poly<LOGN, true, TAB, true>(p, key, c, sh.acc, scr, lane, q, i, u, abar, u != 0,
                                                                                acc0, acc1, t0, last, last + 1);
            STAMP(1);
            int32_t s0[REGS], s1[REGS];
#pragma unroll
            for (int r = 0; r < REGS; ++r) {
                s0[r] = mont_redc(acc0[r], c.P, c.pinv);                // at most l-1 rows: |.| < 0.93P
                s1[r] = mont_redc(acc1[r], c.P, c.pinv);
            }
            NTT::write_row(s0, role_b ? sh.pb0[base] : sh.pa0[base], lane);
            NTT::write_row(s1, role_b ? sh.pb[base] : sh.pa1[base], lane);
        }
        typename SUB::InvTw2 t2;                            // requested before the barrier, in flight across it
        t2.load(ch, lane);
        STAMP(2);
        lds_barrier();
        STAMP(3);
        {
            // Half h of the summed spectrum of output polynomial u, in the half transform's layout: slot 8 lane + reg of
            // the half is slot 16 (32 h + lane / 2) + 8 (lane & 1) + reg of the full-size rows the forward phase wrote
            const int off = NTT::row_base(32 * h + (lane >> 1)) + RS * (lane & 1);
            const uint32_t *rows[4] = {sh.pa0[base] + off, sh.pb0[base] + off, sh.pa1[base ^ 2] + off, sh.pb[base ^ 2] + off};
            int32_t t[RS];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int g = 0; g < RS / 4; ++g) {
                    const uint4 v = reinterpret_cast<const uint4 *>(rows[k])[g];
                    if (k == 0) { t[4 * g] = (int32_t)v.x; t[4 * g + 1] = (int32_t)v.y; t[4 * g + 2] = (int32_t)v.z; t[4 * g + 3] = (int32_t)v.w; }
                    else { t[4 * g] += (int32_t)v.x; t[4 * g + 1] += (int32_t)v.y; t[4 * g + 2] += (int32_t)v.z; t[4 * g + 3] += (int32_t)v.w; }
                }
            }                                                           // |.| < 3.3P (the inverse takes < 4P)
            // (layout R for this inverse: measured, round 4 -- with layout H the compiler's schedule of this kernel came out
            // 0.13 ms per rotation slower, 2.87 against 2.74 ms, although H saves LDS cycles here too; profiles/archive/r04_ab_*.txt)
            SUB::template inverse<false>(t, ch, scr, lane, t2);      // half-transform outputs, natural order, |t| < P
#pragma unroll
            for (int r = 0; r < RS; ++r) scr[r * 64 + lane] = (uint32_t)t[r];
        }
        STAMP(4);
        lds_barrier();
        STAMP(5);
        {
            // the four waves of output polynomial u take a quarter of the register rows each and finish coefficients
            // j and j + N/2 of it, for both primes (split_finish: last inverse stage, CRT)
            const uint32_t *a0 = sh.scr[(u << 1)], *a1 = sh.scr[(u << 1) | 4];
            const uint32_t *b0 = sh.scr[(u << 1) | 1], *b1 = sh.scr[(u << 1) | 5];
            const int part = q | (h << 1);
#pragma unroll
            for (int r = 0; r < QUARTER; ++r) {
                const int jl = (part * QUARTER + r) * 64 + lane;
                const int32_t va0 = (int32_t)a0[jl], va1 = (int32_t)a1[jl], vb0 = (int32_t)b0[jl], vb1 = (int32_t)b1[jl];
                sh.acc.set(u, jl, sh.acc.get(u, jl) + split_finish(0, va0, va1, vb0, vb1, iw1_0, iw1_1));
                sh.acc.set(u, M + jl, sh.acc.get(u, M + jl) + split_finish(1, va0, va1, vb0, vb1, iw1_0, iw1_1));
            }
        }
        STAMP(6);
        lds_barrier();
        STAMP(7);
    }
    STAMP_FLUSH;
    extract_sample<LOGN, 512>(p, rd, sh.acc, u_buf, acc_dbg, tid);
    clk.end(p);
}

// ---------------------------------------------------------------------------
// K1+K2: blind rotate, split form.  grid = rotations, 512 threads: wave (q, u, h) works modulo
// prime q on input polynomial u and on HALF h of every transform, N/128 coefficients per lane.
// A Cooley-Tukey transform splits after its first stage: y_h[j] = x[j] +/- W[1] x[j + N/2] (j < N/2)
// are the inputs of two independent N/2-point transforms whose twiddles are those of the subtree
// below block h (engine.cpp make_twiddles lays them out like an N/2-point table, so WaveNtt<LOGN-1>
// runs them unchanged), and their outputs are halves [h N/2, (h+1) N/2) of the full spectrum in the
// same order -- the key image is read as it is, each lane fetching its 16-byte groups from two rows.
// The inverse runs the two half transforms and ends with the stage-0 butterflies
// (a0 + a1, (a0 - a1) W^-1[1]) on outputs of BOTH halves: every wave leaves its half-transform
// outputs in LDS, and each of the four waves of output polynomial u finishes coefficients 64 r + lane
// and N/2 + 64 r + lane for a quarter of the register rows r -- for both primes, so that last stage,
// the recombination and the accumulator update need one exchange, three workgroup barriers per step
// as in the 4-wave form.
// What it buys: at N = 2048 a wave holds 16 coefficients per lane instead of 32, both row sums stay in
// 64 bits and two waves per SIMD fit (a 4-wave form fitted N = 2048 only with both sums reduced per row and was 38 %
// slower: removed in round 6); at N = 1024, 8 coefficients per lane and four waves per SIMD.  What it costs: stage 0 is outside the radix-4 pairing (digits:
// one table read and one addition per coefficient and gadget row), D is computed by both halves,
// 8 waves meet at each barrier.  Same integers as the other forms.
// ---------------------------------------------------------------------------
// Digit tables of the split form (TM = table mode).  Stage 0 and the first radix-4 step of a half transform
// multiply gadget digits by constants only (stage 0 has one twiddle, the step's stages one block each):
//   TM = 1: stage 0 from a table, [prime][h][2^Bgbit] words (+/- W[1] d);
//   TM = 2: stage 0 AND the first radix-4 step from eleven tables: with x_i = lo_i +/- W[1] hi_i the step's
//           A = w1 x2, S = w2 x1 + w1w2 x3, S' = w3 x1 - w1w3 x3 are sums of two or four products digit x
//           constant -- 11 reads and 14 additions per group of four outputs instead of 4 reads, 11
//           multiplier-class instructions and 10 additions.  Entries are centred (|.| <= P/2) so that the
//           sums of up to seven of them stay small: below 3.5P + 2^11 after the step, 9.7P after the transform.
//           Digits of at most SPLIT_TAB2_BITS bits (the N = 2048 set: 11 KB of tables).
constexpr int SPLIT_TAB2_BITS = 6;
// Issue priority inside a SIMD (round 5).  The two waves of a SIMD, (q, u, h = 0) and (q, u, h = 1), run the same program
// between the same barriers, and the hardware arbitrates their issue by AGE: the older half runs each phase at the pace of a
// lone wave, the younger fills its bubbles and finishes the phase alone while the older waits at the barrier.  The younger
// half therefore raises its priority when it starts its SECOND gadget row and keeps it to the end of the step: the older
// half, ahead by then, fills ITS bubbles, and the two reach the barriers together.  Measured on one box, variants
// interleaved (profiles/r05_ab_kernel_variants.txt): N = 2048, 4,096 rotations 145.3 -> 142.2 ms, one rotation 9.01 -> 8.73
// ms; raising it at the third row, dropping it at the first barrier, for the middle row only, or by time slices of 2^11 ..
// 2^15 cycles all measured between the two.
template <int LOGN, int TM>
struct BrSplitLds {
    using SUB = WaveNtt<LOGN - 1>;
    AccLds<LOGN> acc;
    uint32_t scr[8][SUB::SCRATCH_WORDS];           // wave-private NTT transposes; from the end of the inverse to the
                                                   // step's last barrier: its outputs (natural order), read by 3 waves
    uint32_t x1[8][SUB::ROW_WORDS];                // partial sums sent to the wave of the other input polynomial
    uint16_t bar[1024 + 8];
    alignas(8) uint32_t tab[2][2][TM == 2 ? 11 << SPLIT_TAB2_BITS : DIGIT_TAB];     // [prime][h][table][digit field]
    // N = 2048 (one workgroup per CU, 30 KB of LDS to spare): per prime and half, LDS copies of the half transforms'
    // second- and third-pass forward twiddles (ntt_wave.hpp forward_digits LDSTW)
    static constexpr bool LTW = LOGN == 11;
    uint4 ft1[LTW ? 2 : 1][LTW ? 2 : 1][LTW ? (64 >> SUB::LC) : 1][SUB::FwdTw1::IMAGE16];
    uint4 ft2[LTW ? 2 : 1][LTW ? 2 : 1][LTW ? 64 : 1][SUB::FwdTw2::IMAGE16];
};
static_assert(sizeof(BrSplitLds<11, 2>) <= 160 * 1024, "the N = 2048 split form must fit a CU's LDS");

template <int LOGN, int TM>
__global__ __launch_bounds__(512, (LOGN == 10 ? 4 : 2)) void blind_rotate_split_kernel(
    DevParams p, DevKey key, const int32_t *__restrict__ pool, const RotDesc *__restrict__ rots,
    int32_t *__restrict__ u_buf, int32_t *__restrict__ acc_dbg) {
    using SUB = WaveNtt<LOGN - 1>;
    constexpr int N = 1 << LOGN, M = N / 2, RS = SUB::REGS, RF = 2 * RS, G4 = RS / 4, QUARTER = RS / 4;
    __shared__ __align__(16) BrSplitLds<LOGN, TM> sh;
    constexpr bool TAB = TM == 1;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wv & 1, u = (wv >> 1) & 1, h = wv >> 2;
    const int lane = tid & 63;
    PrimeCtx c = make_sub_ctx(q, h, key.tw, N);
    constexpr bool LTW = BrSplitLds<LOGN, TM>::LTW;
    uint32_t *scr = sh.scr[wv];
    const int n = p.n;
    const RotDesc rd = rots[blockIdx.x];
    // stage-0 twiddles (entry 1 of the full-size tables): forward of this wave's prime, inverse of both
    const uint32_t w1 = key.tw[(size_t)(q * 2 + 0) * N + 1];
    const uint32_t iw1_0 = key.tw[(size_t)1 * N + 1], iw1_1 = key.tw[(size_t)3 * N + 1];
    ClockProbe clk;
    clk.begin(p);

    prelude_modswitch<LOGN, 512>(p, rd, pool, sh.bar, tid);
    if constexpr (TAB) {
        // the two waves (q, 0, h) and (q, 1, h) fill the table of (q, h)
        const int fields = 1 << p.Bgbit;
        for (int f = (u << 6) | lane; f < fields; f += 128) {
            const int32_t d = f < fields / 2 ? f : f - fields;
            const int32_t v = mont_mul(d, w1, c.P, c.pinv);
            sh.tab[q][h][f] = (uint32_t)(h ? -v : v);
        }
    }
    if constexpr (LTW) {
        if (u == 0) {                                // wave (q, 0, h) copies the twiddles of (q, h) into LDS, once
            typename SUB::FwdTw1 a;
            a.load(c, lane);
            if ((lane & ((1 << SUB::LC) - 1)) == 0) a.to_image(sh.ft1[q][h][lane >> SUB::LC]);
            typename SUB::FwdTw2 b;
            b.load(c, lane);
            b.to_image(sh.ft2[q][h][lane]);
        }
        c.fw1 = sh.ft1[q][h][lane >> SUB::LC];
        c.fw2 = sh.ft2[q][h][lane];
    }
    if constexpr (TM == 2) {
        // table k of (q, h), entry f: digit(f) * [W[1] * (+1 | -1 for h = 1)] * constant_k, centred
        const uint4 qd = c.qf[1];                            // {w2, w3, w1 w2, P - w1 w3} of the half transform's first step
        const uint32_t cst[6] = {0u, c.wf[1], qd.x, qd.z, qd.y, qd.w};          // k = 0 | 1,2 | 3,4 | 5,6 | 7,8 | 9,10
        const int fields = 1 << p.Bgbit;
        for (int e = (u << 6) | lane; e < 11 * fields; e += 128) {
            const int k = e / fields, f = e - k * fields;
            const int32_t d = f < fields / 2 ? f : f - fields;
            const bool hi = k == 0 || (k & 1) == 0;          // tables of the upper-half digits carry +/- W[1]
            int32_t v = d;
            if (hi) { v = mont_mul(v, w1, c.P, c.pinv); if (h) v = -v; }
            if (k > 0) v = mont_mul(v, cst[(k + 1) >> 1], c.P, c.pinv);
            const int32_t half = (int32_t)(c.P >> 1);
            if (v > half) v -= (int32_t)c.P;
            if (v < -half) v += (int32_t)c.P;
            // tables read by one digit sit side by side (8-byte entries, as in ntt_wave.hpp): (3,7) lo1, (4,8) hi1, (5,9) lo3, (6,10) hi3
            const int pos = k < 3 ? (k << SPLIT_TAB2_BITS) + f
                                  : ((3 + 2 * ((k - 3) & 3)) << SPLIT_TAB2_BITS) + 2 * f + (k >= 7 ? 1 : 0);
            sh.tab[q][h][pos] = (uint32_t)v;
        }
    }
    __syncthreads();
    if (q == 0) {
        const int barb = sh.bar[n];
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int j = h * M + r * 64 + lane;
            sh.acc.set(u, j, u == 0 ? 0u : testvector_coef<LOGN>(j, barb, p.mu));
        }
    }
    __syncthreads();

    const int width = p.Bgbit;
    // where the 16-byte groups of this lane's slice of spectrum half h sit in a row of the key image
    // (full-size layout L2: slot j = 2 RS * lane' + reg; this lane holds slots h N/2 + RS * lane + reg)
    const int lane_off = G4 * (lane & 1) * 64 + h * 32 + (lane >> 1);
    const int o0 = u ? N / 4 : 0, o1 = N / 4 - o0;          // acc0 <- output polynomial u (kept), acc1 <- 1-u (sent)
    typename SUB::FwdTw0 t0;                              // first-pass twiddles of the half transforms: loaded once
    t0.load(c, lane);
    for (int i = 0; i < n; ++i) {
        const int abar = __builtin_amdgcn_readfirstlane((int)sh.bar[i]);
        if (abar == 0) continue;
        uint32_t D[RF];                                      // coefficients 64 r + lane, r < 2 RS: both halves of the input
        sh.acc.template rotated_difference<RF>(D, u, lane, abar, p.decomp_offset);
        int64_t acc0[RS], acc1[RS];
        auto row = [&](int jj, auto first) {
            const int prow = u * p.l + jj;
            const uint4 *bp = reinterpret_cast<const uint4 *>(
                                  key.bk_img + ((size_t)((size_t)i * p.kpl + prow) * 2 + q) * 2 * N) + lane_off;
            uint4 b0[G4], b1[G4];
#pragma unroll
            for (int g = 0; g < G4; ++g) b0[g] = bp[o0 + g * 64];
#pragma unroll
            for (int g = 0; g < G4; ++g) b1[g] = bp[o1 + g * 64];
            const int shift = 32 - (jj + 1) * width;
            int32_t x[RS];
            if constexpr (TM == 2) {
                constexpr int RBIT = SUB::rbit_of(0), hb = 1 << RBIT, lb = hb >> 1;
                const uint32_t mask4 = ((1u << width) - 1u) << 2;
                const int sh2 = shift - 2;
                const char *tab = reinterpret_cast<const char *>(sh.tab[q][h]);
                auto entry = [&](int k, uint32_t off) {
                    return (int32_t)*reinterpret_cast<const uint32_t *>(tab + (k << (SPLIT_TAB2_BITS + 2)) + off);
                };
                auto field = [&](int i) { return (D[i] >> sh2) & mask4; };
                const uint32_t mask8 = mask4 << 1;
                const int sh3 = shift - 3;
                auto pair = [&](int k, uint32_t off) {       // tables (k, k + 4) of one digit: an 8-byte entry
                    return *reinterpret_cast<const uint2 *>(tab + ((3 + 2 * (k - 3)) << (SPLIT_TAB2_BITS + 2)) + off);
                };
                auto field8 = [&](int i) { return (D[i] >> sh3) & mask8; };
#pragma unroll
                for (int r = 0; r < RS; ++r)
                    if (!(r & (hb | lb))) {
                        const uint32_t l2 = field(r | hb), h0 = field(RS + r), h2 = field(RS + (r | hb));
                        const uint2 pl1 = pair(3, field8(r | lb)), ph1 = pair(4, field8(RS + (r | lb)));
                        const uint2 pl3 = pair(5, field8(r | hb | lb)), ph3 = pair(6, field8(RS + (r | hb | lb)));
                        const int32_t x0 = __builtin_amdgcn_sbfe((int32_t)D[r], shift, width) + entry(0, h0);
                        const int32_t A = entry(1, l2) + entry(2, h2);
                        const int32_t S = ((int32_t)pl1.x + (int32_t)ph1.x) + ((int32_t)pl3.x + (int32_t)ph3.x);
                        const int32_t T = ((int32_t)pl1.y + (int32_t)ph1.y) + ((int32_t)pl3.y + (int32_t)ph3.y);
                        const int32_t uu = x0 + A, vv = x0 - A;
                        x[r] = uu + S; x[r | lb] = uu - S; x[r | hb] = vv + T; x[r | hb | lb] = vv - T;
                    }
            } else if constexpr (TAB) {
                const uint32_t mask4 = ((1u << width) - 1u) << 2;
                const int sh2 = shift - 2;
                const char *tab = reinterpret_cast<const char *>(sh.tab[q][h]);
#pragma unroll
                for (int r = 0; r < RS; ++r)
                    x[r] = __builtin_amdgcn_sbfe((int32_t)D[r], shift, width) +
                           (int32_t)*reinterpret_cast<const uint32_t *>(tab + ((D[r + RS] >> sh2) & mask4));
            } else {
#pragma unroll
                for (int r = 0; r < RS; ++r) {
                    const int32_t v = mont_mul(__builtin_amdgcn_sbfe((int32_t)D[r + RS], shift, width), w1, c.P, c.pinv);
                    x[r] = __builtin_amdgcn_sbfe((int32_t)D[r], shift, width) + (h ? -v : v);
                }
            }
            if constexpr (TM == 2) SUB::template forward_rest<LTW>(x, c, scr, lane, t0);     // < 3.5P + 2^11 in, < 9.7P out
            else SUB::template forward<LTW>(x, c, scr, lane, t0);    // |x| < P + 2^11 in, < 8.3P out
#pragma unroll
            for (int g = 0; g < G4; ++g) {
                const int32_t bb0[4] = {(int32_t)b0[g].x, (int32_t)b0[g].y, (int32_t)b0[g].z, (int32_t)b0[g].w};
                const int32_t bb1[4] = {(int32_t)b1[g].x, (int32_t)b1[g].y, (int32_t)b1[g].z, (int32_t)b1[g].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    if constexpr (decltype(first)::value) {
                        acc0[r] = (int64_t)x[r] * bb0[e];
                        acc1[r] = (int64_t)x[r] * bb1[e];
                    } else {
                        acc0[r] += (int64_t)x[r] * bb0[e];
                        acc1[r] += (int64_t)x[r] * bb1[e];
                    }
                }
            }
        };
        row(0, std::true_type{});
        // (a loop on purpose: with the rows in line the compiler overlaps them and runs out of registers --
        // measured 4 % slower at N = 2048 and 20 % at N = 1024)
#pragma unroll 1
        for (int jj = 1; jj < p.l; ++jj) {
            if (h == 1 && jj == 1) __builtin_amdgcn_s_setprio(1);     // the younger half, from its second row (above)
            row(jj, std::false_type{});
        }

        int32_t t[RS];
        {
            int32_t send[RS];
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                t[r] = mont_redc(acc0[r], c.P, c.pinv);              // l <= 4 rows of |x| < 9.7P: |.| < 1.8P
                send[r] = mont_redc(acc1[r], c.P, c.pinv);
            }
            SUB::write_row(send, sh.x1[wv], lane);
        }
        typename SUB::InvTw2 t2;                            // requested before the barrier, in flight across it
        t2.load(c, lane);
        lds_barrier();
        {
            int32_t other[RS];
            SUB::read_row(other, sh.x1[wv ^ 2], lane);
#pragma unroll
            for (int r = 0; r < RS; ++r) t[r] += other[r];          // |.| < 3.6P (the inverse takes < 4P)
        }
        SUB::inverse(t, c, scr, lane, t2);
#pragma unroll
        for (int r = 0; r < RS; ++r) scr[r * 64 + lane] = (uint32_t)t[r];
        lds_barrier();
        {
            // the four waves of output polynomial u take a quarter of the register rows each and finish
            // coefficients j and j + N/2 of it (equal work for every wave; the four words read serve both)
            const uint32_t *a0 = sh.scr[(u << 1)], *a1 = sh.scr[(u << 1) | 4];
            const uint32_t *b0 = sh.scr[(u << 1) | 1], *b1 = sh.scr[(u << 1) | 5];
            const int part = q | (h << 1);
#pragma unroll
            for (int r = 0; r < QUARTER; ++r) {
                const int jl = (part * QUARTER + r) * 64 + lane;
                const int32_t va0 = (int32_t)a0[jl], va1 = (int32_t)a1[jl], vb0 = (int32_t)b0[jl], vb1 = (int32_t)b1[jl];
                sh.acc.set(u, jl, sh.acc.get(u, jl) + split_finish(0, va0, va1, vb0, vb1, iw1_0, iw1_1));
                sh.acc.set(u, M + jl, sh.acc.get(u, M + jl) + split_finish(1, va0, va1, vb0, vb1, iw1_0, iw1_1));
            }
        }
        if (h == 1) __builtin_amdgcn_s_setprio(0);          // the younger half kept the priority through the step's tail
        lds_barrier();
    }
    extract_sample<LOGN, 512>(p, rd, sh.acc, u_buf, acc_dbg, tid);
    clk.end(p);
}

// test kernel: the negacyclic product of negacyclic_kernel through split transforms (4 waves: prime, half)
template <int LOGN>
__global__ __launch_bounds__(256) void negacyclic_split_kernel(const int32_t *__restrict__ ip, const uint32_t *__restrict__ img,
                                                               const uint32_t *__restrict__ tw, int32_t *__restrict__ res) {
    using SUB = WaveNtt<LOGN - 1>;
    constexpr int N = 1 << LOGN, M = N / 2, RS = SUB::REGS, G4 = RS / 4, HALF = RS / 2;
    __shared__ __align__(16) uint32_t lds_scr[4][SUB::SCRATCH_WORDS];
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wv & 1, h = wv >> 1;
    const int lane = tid & 63;
    const PrimeCtx c = make_sub_ctx(q, h, tw, N);
    uint32_t *scr = lds_scr[wv];
    const uint32_t w1 = tw[(size_t)(q * 2 + 0) * N + 1];
    const uint32_t iw1_0 = tw[(size_t)1 * N + 1], iw1_1 = tw[(size_t)3 * N + 1];
    const int32_t *src = ip + (size_t)blockIdx.x * N;
    int32_t x[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const int32_t lo = src[r * 64 + lane] % (int32_t)c.P, hi = src[M + r * 64 + lane] % (int32_t)c.P;
        const int32_t v = mont_mul(hi, w1, c.P, c.pinv);
        x[r] = lo + (h ? -v : v);                                       // |x| < 2P
    }
    SUB::forward(x, c, scr, lane);
    const int lane_off = G4 * (lane & 1) * 64 + h * 32 + (lane >> 1);
    const uint4 *bp = reinterpret_cast<const uint4 *>(img + (size_t)(blockIdx.x * 2 + q) * N) + lane_off;
    int32_t t[RS];
#pragma unroll
    for (int g = 0; g < G4; ++g) {
        const uint4 b = bp[g * 64];
        t[4 * g + 0] = mont_redc((int64_t)x[4 * g + 0] * (int32_t)b.x, c.P, c.pinv);
        t[4 * g + 1] = mont_redc((int64_t)x[4 * g + 1] * (int32_t)b.y, c.P, c.pinv);
        t[4 * g + 2] = mont_redc((int64_t)x[4 * g + 2] * (int32_t)b.z, c.P, c.pinv);
        t[4 * g + 3] = mont_redc((int64_t)x[4 * g + 3] * (int32_t)b.w, c.P, c.pinv);
    }
    SUB::inverse(t, c, scr, lane);
#pragma unroll
    for (int r = 0; r < RS; ++r) scr[r * 64 + lane] = (uint32_t)t[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < HALF; ++r) {
        const int jl = (q * HALF + r) * 64 + lane;
        res[(size_t)blockIdx.x * N + h * M + jl] =
            (int32_t)split_finish(h, (int32_t)lds_scr[0][jl], (int32_t)lds_scr[2][jl], (int32_t)lds_scr[1][jl],
                                  (int32_t)lds_scr[3][jl], iw1_0, iw1_1);
    }
}


// ---------------------------------------------------------------------------
// K3/K4: key switch (tfhe lweKeySwitchTranslate_fromArray).  grid = gates.
// One workgroup per gate; thread t owns 4 consecutive output words and
// subtracts the selected KSK rows with 16-byte loads.
// ---------------------------------------------------------------------------
constexpr int KS_MAX_THREADS = 320;   // launched with ct_stride/4 rounded up to a wave: 192 (n=630), 320 (n=1024)

// grid (gates, splits): block (g, s) handles input coefficients [s*nin/splits, (s+1)*nin/splits).
// splits == 1: the result goes straight to the destination slot.  splits > 1 (narrow
// levels, where one workgroup per gate would leave the chip idle and serialise 6,144
// dependent row loads): partial sums go to `partial[g][s][ct_stride]` and
// ks_reduce_kernel adds them (integer adds: any order is bit-exact).
__global__ __launch_bounds__(KS_MAX_THREADS) void keyswitch_kernel(DevParams p, DevKey key, const int32_t *__restrict__ u_buf,
                                                               const KsDesc *__restrict__ descs,
                                                               int32_t *__restrict__ pool, int32_t *__restrict__ partial) {
    __shared__ uint32_t su[2048 + 8];
    const int tid = threadIdx.x;
    const KsDesc d = descs[blockIdx.x];
    const int nin = p.k * p.N;
    const int splits = gridDim.y, split = blockIdx.y;
    const int i0 = (int)((long long)nin * split / splits), i1 = (int)((long long)nin * (split + 1) / splits);
    {
        const int32_t *u0 = u_buf + (size_t)d.u0 * p.u_stride;
        const int32_t *u1 = d.u1 >= 0 ? u_buf + (size_t)d.u1 * p.u_stride : nullptr;
        for (int j = i0 + tid; j < i1; j += (int)blockDim.x) su[j - i0] = (uint32_t)u0[j] + (u1 ? (uint32_t)u1[j] : 0u);
        if (tid == 0) su[i1 - i0] = (uint32_t)u0[nin] + (u1 ? (uint32_t)u1[nin] : 0u) + (uint32_t)d.add_b;   // body
    }
    __syncthreads();
    const int nvec = p.ct_stride >> 2;
    if (tid >= nvec) return;
    const int t = p.ks_t, bb = p.ks_basebit;
    const uint32_t mask = (1u << bb) - 1u;
    const size_t row_vecs = (size_t)nvec;
    const uint4 *ksk = reinterpret_cast<const uint4 *>(key.ksk) + tid;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = i0; i < i1; ++i) {
        const uint32_t aibar = su[i - i0] + p.ks_prec_offset;
        for (int j = 0; j < t; ++j) {
            const uint32_t aij = (aibar >> (32 - (j + 1) * bb)) & mask;
            if (aij == 0) continue;
            const uint4 row = ksk[((size_t)(i * t + j) * mask + (aij - 1)) * row_vecs];
            acc.x -= row.x; acc.y -= row.y; acc.z -= row.z; acc.w -= row.w;
        }
    }
    uint32_t o[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int wi = 4 * tid + e;
        if (wi == p.n && split == 0) o[e] += su[i1 - i0];
        if (wi > p.n) o[e] = 0;
    }
    int32_t *dst = splits == 1 ? pool + (size_t)d.dst_slot * p.ct_stride
                               : partial + ((size_t)blockIdx.x * splits + split) * p.ct_stride;
    reinterpret_cast<uint4 *>(dst)[tid] = make_uint4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(KS_MAX_THREADS) void ks_reduce_kernel(DevParams p, const KsDesc *__restrict__ descs, int splits,
                                                               const int32_t *__restrict__ partial,
                                                               int32_t *__restrict__ pool) {
    const int tid = threadIdx.x;
    if (tid >= (p.ct_stride >> 2)) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(partial + (size_t)blockIdx.x * splits * p.ct_stride) + tid;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int s = 0; s < splits; ++s) {
        const uint4 v = src[(size_t)s * (p.ct_stride >> 2)];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<uint4 *>(pool + (size_t)descs[blockIdx.x].dst_slot * p.ct_stride)[tid] = acc;
}

// Tiled forms for wide launches.  grid (ceil(gates / G), splits), THREADS = ct_stride/4 rounded
// up to a wave.  A workgroup key-switches G gates over one range of input coefficients:
// it streams the range's KSK rows ONCE and applies each row to every gate of the tile
// whose digit selects it.  The per-gate form above fetches 15.5 MB of rows per gate (from
// L2, which is what bounds it); here the fetch is shared by G gates.  Both need ks_t = 8,
// ks_basebit = 2 (every built-in parameter set) and at most 64 coefficients per range;
// partial sums go through ks_reduce_kernel as above.
//
// LDS-strip form (tuning "ks_index" 0; G = 16).  A thread only ever needs its own 16-byte column of a row,
// so the staged rows (12 rows = four digit positions of one coefficient per stage, prefetched a stage
// ahead) sit in a thread-private LDS strip -- LDS because the row is picked by a run-time digit -- and the stage
// loop has no barrier.  Strips are laid out [digit position][digit 0..3]; digit 0 is a strip of zeros,
// so the inner loop is branch-free (address = digit bits, one 16-byte LDS read, four subtractions).
// Digits come from one LDS word per (gate, coefficient), the same for all lanes.  The strip reads of the
// next pair of gates are issued BEFORE the subtractions of the current pair, and the wait is for "all but
// the last eight reads" (a wave's LDS operations complete in order, so s_waitcnt lgkmcnt(8) means the older
// eight have arrived).  Plain HIP source apart from the explicit ds_read / s_waitcnt: what runs if the pinned
// registers of the index form below ever stop compiling.  Bound by LDS instruction issue: 105 ms of key switch
// per match where the index form takes 56 (profiles/archive/r04_ks_tile.txt, r04_ks_register_forms.txt).
typedef uint32_t ks_u4 __attribute__((ext_vector_type(4)));

template <int THREADS, int G>
__global__ __launch_bounds__(THREADS) void keyswitch_strip_kernel(DevParams p, DevKey key, const int32_t *__restrict__ u_buf,
                                                               const KsDesc *__restrict__ descs, int count,
                                                               int32_t *__restrict__ partial) {
    typedef ks_u4 vw;
    constexpr int JB = 4, ROWS = JB * 3, MAXR = 64, EB = 16;         // EB: bytes of a strip entry
    __shared__ __align__(16) vw rows[JB * 4 * THREADS];
    __shared__ __align__(16) uint32_t su[MAXR][G];       // [coefficient][gate]: four gates' digit words per 16-byte read
    __shared__ uint32_t sbody[G];
    const int tid = threadIdx.x;
    const int nin = p.k * p.N;
    const int splits = gridDim.y, split = blockIdx.y;
    const int i0 = (int)((long long)nin * split / splits), i1 = (int)((long long)nin * (split + 1) / splits);
    const int range = i1 - i0;
    const int g0 = blockIdx.x * G;
    for (int e = tid; e < G * range; e += THREADS) {
        const int g = e / range, ii = e - g * range;
        uint32_t v = 0;                                  // gates past the end: every digit 0
        if (g0 + g < count) {
            const KsDesc d = descs[g0 + g];
            v = (uint32_t)u_buf[(size_t)d.u0 * p.u_stride + i0 + ii] + p.ks_prec_offset;
            if (d.u1 >= 0) v += (uint32_t)u_buf[(size_t)d.u1 * p.u_stride + i0 + ii];
        }
        su[ii][g] = v;
    }
    if (tid < G && g0 + tid < count) {
        const KsDesc d = descs[g0 + tid];
        uint32_t b = (uint32_t)u_buf[(size_t)d.u0 * p.u_stride + nin] + (uint32_t)d.add_b;
        if (d.u1 >= 0) b += (uint32_t)u_buf[(size_t)d.u1 * p.u_stride + nin];
        sbody[tid] = b;
    }
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) rows[(jj * 4) * THREADS + tid] = (vw)(0u);
    __syncthreads();
    const int nvec = p.ct_stride / 4;
    if (tid >= nvec) return;                             // no barrier below
    const vw *ksk = reinterpret_cast<const vw *>(key.ksk) + tid;
    vw acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = (vw)(0u);
    const int nst = range * 2;                           // stage = (coefficient, half of its 8 digits)
    const vw *src = ksk + (size_t)(i0 * 8) * 3 * nvec;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) rows[(r / 3 * 4 + r % 3 + 1) * THREADS + tid] = src[(size_t)r * nvec];
    // byte address of this thread's strip in LDS (the low half of a generic LDS pointer)
    const uint32_t strip_addr = (uint32_t)reinterpret_cast<uintptr_t>(rows) + (uint32_t)tid * (uint32_t)EB;
    for (int st = 0; st < nst; ++st) {
        // next stage's rows: loads issued before this stage's arithmetic, stored after it
        // (the strip is private to the thread and a wave's LDS operations stay in order).
        // Unconditional: the last stage re-reads itself.
        // (twelve named values, not an array: with the explicit LDS reads below the compiler
        // would keep an array in scratch memory)
        if (st + 1 < nst) src += (size_t)ROWS * nvec;
#define KS_ROWS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)
#define KS_LOAD(r) const vw pre##r = src[(size_t)(r) * nvec];
        KS_ROWS(KS_LOAD)
#undef KS_LOAD
        const int ii = st >> 1;
        const int sh0 = 24 - 8 * (st & 1);               // digit j sits at bits [31-2j, 30-2j]
        auto issue = [&](vw (&buf)[8], uint32_t x0, uint32_t x1) {      // the eight strip reads of two gates
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t x = (e ? x1 : x0) >> sh0;         // this stage's four digits in the low byte
#pragma unroll
                for (int jj = 0; jj < JB; ++jj) {
                    const uint32_t d = (x >> (6 - 2 * jj)) & 3u;
                    const uint32_t addr = strip_addr + d * (uint32_t)(THREADS * EB);
                    asm volatile("ds_read_b128 %0, %1 offset:%2"
                                 : "=v"(buf[e * 4 + jj]) : "v"(addr), "n"(jj * 4 * THREADS * EB) : "memory");
                }
            }
        };
        // all sixteen gates' digit words first (one wait), then pairs of gates through two buffers of eight rows
        uint32_t xg[G];
#pragma unroll
        for (int g = 0; g < G; g += 4) {
            const uint4 xs = *reinterpret_cast<const uint4 *>(&su[ii][g]);
            xg[g] = xs.x; xg[g + 1] = xs.y; xg[g + 2] = xs.z; xg[g + 3] = xs.w;
        }
        vw bufA[8], bufB[8];
        issue(bufA, xg[0], xg[1]);
#pragma unroll
        for (int g = 0; g < G; g += 2) {
            vw (&cur)[8] = (g & 2) ? bufB : bufA;
            vw (&nxt)[8] = (g & 2) ? bufA : bufB;
            if (g + 2 < G) {
                issue(nxt, xg[g + 2], xg[g + 3]);
                asm volatile("s_waitcnt lgkmcnt(8)"
                             : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int jj = 0; jj < JB; ++jj) acc[g + e] -= cur[e * 4 + jj];
        }
#define KS_STORE(r) rows[((r) / 3 * 4 + (r) % 3 + 1) * THREADS + tid] = pre##r;
        KS_ROWS(KS_STORE)
#undef KS_STORE
#undef KS_ROWS
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g0 + g >= count) break;
        vw out = acc[g];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int wi = 4 * tid + e;
            if (wi == p.n && split == 0) out[e] += sbody[g];
            if (wi > p.n) out[e] = 0;
        }
        reinterpret_cast<vw *>(partial + ((size_t)(g0 + g) * splits + split) * p.ct_stride)[tid] = out;
    }
}

// Index form (round 4; the default, tuning "ks_index" 1).  The strip form pays one 16-byte LDS read per four subtractions,
// which is what bounds it.  But the digit is the same for every lane of a wave: it can live in a SCALAR register and pick
// the row.  A thread holds its 16-byte column of the three non-zero rows of a digit position in registers (loaded
// straight from global memory, a position ahead).  Picking by scalar branches was built first and is bound by the latency
// of its taken branches (about 130 cycles per wave, gate and digit whatever the occupancy; removed in round 6).  gfx9's
// VGPR index mode removes the branches: with SRC1_REL set, the row operand of a v_sub_u32 is VGPR[encoded + M0[7:0]], so
// ONE instruction sequence subtracts whichever row the scalar index picks.  The compiler cannot express that (section 5 of DESIGN.md: a dynamically indexed
// register array goes to scratch memory), and inline asm cannot name a sub-register of a tuple operand -- but an asm operand
// can be PINNED to physical registers ("{v[96:99]}"), and then the asm text may name them.  Register map (pinned only at
// the asm statements; the compiler keeps the values there in between because every statement of the loop wants them there):
// (tile 16; tile 32 has its accumulators in v32..v159 and the rows from v160 -- the statements are generated,
// tools/gen_ks_index_asm.py -> ks_index_asm.inc)
//     v32..v95    accumulators, four per gate
//     v96..v107   rows 1, 2, 3 of an even digit position (set 0)   v108..v119  rows 1, 2, 3 of an odd position (set 1)
// The prologue stores every coefficient word as eight nibbles 4 x digit; a (gate, position) is s_bfe_u32 of its nibble,
// a scalar branch over the gate for digit 0 (a quarter of all digits; s_bfe_u32 leaves SCC = result != 0, and a branch
// that is mostly not taken costs the wave little), s_set_gpr_idx_idx and four v_sub_u32 whose row operand is encoded as
// the register four below row 1.  M0 is saved and restored around each statement (the compiler treats it as reserved).
#include "ks_index_asm.inc"      // KsIndexSub<G, OCT, SETB>::run<NIB>: the asm statements (tools/gen_ks_index_asm.py)

// the coefficient word of a gate (digit j at bits [31-2j, 30-2j]) as eight index nibbles (comment above)
__device__ __forceinline__ uint32_t ks_index_nibbles(uint32_t v) {
    uint32_t y = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t d = (v >> (30 - 2 * j)) & 3u;
        y |= (d << 2) << (4 * j);
    }
    return y;
}

template <int THREADS, int G>
__global__ __launch_bounds__(THREADS) void keyswitch_index_kernel(DevParams p, DevKey key, const int32_t *__restrict__ u_buf,
                                                               const KsDesc *__restrict__ descs, int count,
                                                               int32_t *__restrict__ partial) {
    constexpr int MAXR = 64;
    static_assert(G == 16 || G == 24 || G == 32, "tile sizes the statements are generated for");
    __shared__ __align__(16) uint32_t su[MAXR][G];       // [coefficient][gate]: index nibbles
    __shared__ uint32_t sbody[G];
    const int tid = threadIdx.x;
    const int nin = p.k * p.N;
    const int splits = gridDim.y, split = blockIdx.y;
    const int i0 = (int)((long long)nin * split / splits), i1 = (int)((long long)nin * (split + 1) / splits);
    const int range = i1 - i0;
    const int g0 = blockIdx.x * G;
    for (int e = tid; e < G * range; e += THREADS) {
        const int g = e / range, ii = e - g * range;
        uint32_t v = 0;                                  // gates past the end: every digit 0
        if (g0 + g < count) {
            const KsDesc d = descs[g0 + g];
            v = (uint32_t)u_buf[(size_t)d.u0 * p.u_stride + i0 + ii] + p.ks_prec_offset;
            if (d.u1 >= 0) v += (uint32_t)u_buf[(size_t)d.u1 * p.u_stride + i0 + ii];
        }
        su[ii][g] = ks_index_nibbles(v);
    }
    if (tid < G && g0 + tid < count) {
        const KsDesc d = descs[g0 + tid];
        uint32_t b = (uint32_t)u_buf[(size_t)d.u0 * p.u_stride + nin] + (uint32_t)d.add_b;
        if (d.u1 >= 0) b += (uint32_t)u_buf[(size_t)d.u1 * p.u_stride + nin];
        sbody[tid] = b;
    }
    __syncthreads();
    const int nvec = p.ct_stride >> 2;
    if (tid >= nvec) return;                             // no barrier below
    const ks_u4 *src = reinterpret_cast<const ks_u4 *>(key.ksk) + tid + (size_t)(i0 * 8) * 3 * nvec;
    const size_t step = (size_t)3 * nvec;
    ks_u4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = (ks_u4)(0u);
    ks_u4 a1 = src[0], a2 = src[nvec], a3 = src[2 * nvec];
    for (int ii = 0; ii < range; ++ii) {
        uint32_t y[G];
#pragma unroll
        for (int g = 0; g < G; g += 4) {
            const uint4 ys = *reinterpret_cast<const uint4 *>(&su[ii][g]);
            y[g] = __builtin_amdgcn_readfirstlane(ys.x); y[g + 1] = __builtin_amdgcn_readfirstlane(ys.y);
            y[g + 2] = __builtin_amdgcn_readfirstlane(ys.z); y[g + 3] = __builtin_amdgcn_readfirstlane(ys.w);
        }
        const bool last_coeff = ii + 1 == range;
        // The rows of the next digit position are requested before this position's subtractions (the two register sets in
        // turn); the very last request of the range re-reads its own rows instead of running past the table.  (Four sets,
        // three positions ahead: slower, 67.6 against 60.8 ms per match -- the waves' waits are not load latency.)
#define KS_IDX_OCT(O, SET, NIB, R1, R2, R3)                                                                                     \
        KsIndexSub<G, O, SET>::template run<NIB>(acc[8 * (O)], acc[8 * (O) + 1], acc[8 * (O) + 2], acc[8 * (O) + 3],            \
                                                 acc[8 * (O) + 4], acc[8 * (O) + 5], acc[8 * (O) + 6], acc[8 * (O) + 7],        \
                                                 R1, R2, R3, y[8 * (O)], y[8 * (O) + 1], y[8 * (O) + 2], y[8 * (O) + 3],        \
                                                 y[8 * (O) + 4], y[8 * (O) + 5], y[8 * (O) + 6], y[8 * (O) + 7]);
#define KS_IDX_APPLY(SET, NIB, R1, R2, R3)                                                                                      \
        KS_IDX_OCT(0, SET, NIB, R1, R2, R3) KS_IDX_OCT(1, SET, NIB, R1, R2, R3)                                                 \
        if constexpr (G >= 24) { KS_IDX_OCT(2, SET, NIB, R1, R2, R3) }                                                          \
        if constexpr (G == 32) { KS_IDX_OCT(3, SET, NIB, R1, R2, R3) }
#define KS_IDX_PAIR(JP, LAST)                                                                         \
        {                                                                                             \
            src += step;                                                                              \
            const ks_u4 b1 = src[0], b2 = src[nvec], b3 = src[2 * nvec];                              \
            KS_IDX_APPLY(0, 2 * (JP), a1, a2, a3)                                                     \
            if (!(LAST)) src += step;                                                                 \
            a1 = src[0]; a2 = src[nvec]; a3 = src[2 * nvec];                                          \
            KS_IDX_APPLY(1, 2 * (JP) + 1, b1, b2, b3)                                                 \
        }
        KS_IDX_PAIR(0, false) KS_IDX_PAIR(1, false) KS_IDX_PAIR(2, false) KS_IDX_PAIR(3, last_coeff)
#undef KS_IDX_PAIR
#undef KS_IDX_APPLY
#undef KS_IDX_OCT
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g0 + g >= count) break;
        uint32_t o[4] = {acc[g].x, acc[g].y, acc[g].z, acc[g].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int wi = 4 * tid + e;
            if (wi == p.n && split == 0) o[e] += sbody[g];
            if (wi > p.n) o[e] = 0;
        }
        reinterpret_cast<uint4 *>(partial + ((size_t)(g0 + g) * splits + split) * p.ct_stride)[tid] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// K5: bootsNOT
__global__ __launch_bounds__(256) void not_kernel(DevParams p, const NotDesc *__restrict__ descs, int32_t *__restrict__ pool) {
    const NotDesc d = descs[blockIdx.x];
    const int32_t *src = pool + (size_t)d.src_slot * p.ct_stride;
    int32_t *dst = pool + (size_t)d.dst_slot * p.ct_stride;
    for (int i = threadIdx.x; i < p.ct_stride; i += 256) dst[i] = (int32_t)(0u - (uint32_t)src[i]);
}

// packed words <-> pool slots (import/export of ciphertexts, collectives)
__global__ __launch_bounds__(256) void gather_slots_kernel(const int32_t *__restrict__ pool, int stride, int words,
                                                           const int32_t *__restrict__ slots, int32_t *__restrict__ packed) {
    const int32_t *src = pool + (size_t)slots[blockIdx.x] * stride;
    int32_t *dst = packed + (size_t)blockIdx.x * words;
    for (int i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void scatter_slots_kernel(int32_t *__restrict__ pool, int stride, int words,
                                                            const int32_t *__restrict__ slots, const int32_t *__restrict__ packed) {
    int32_t *dst = pool + (size_t)slots[blockIdx.x] * stride;
    const int32_t *src = packed + (size_t)blockIdx.x * words;
    for (int i = threadIdx.x; i < stride; i += 256) dst[i] = i < words ? src[i] : 0;
}

}  // namespace

void launch_gather_slots(hipStream_t s, const int32_t *pool, int stride, int words, const int32_t *slots, int count,
                         int32_t *packed) {
    if (count <= 0) return;
    hipLaunchKernelGGL(gather_slots_kernel, dim3(count), dim3(256), 0, s, pool, stride, words, slots, packed);
}
void launch_scatter_slots(hipStream_t s, int32_t *pool, int stride, int words, const int32_t *slots, int count,
                          const int32_t *packed) {
    if (count <= 0) return;
    hipLaunchKernelGGL(scatter_slots_kernel, dim3(count), dim3(256), 0, s, pool, stride, words, slots, packed);
}

void launch_bk_transform(hipStream_t s, const DevParams &p, const int32_t *raw_polys, uint32_t *img,
                         const uint32_t *tw, int npoly_per_w, int nw, const uint32_t scale[2]) {
    if (npoly_per_w * nw <= 0) return;
    if (p.N == 2048)
        hipLaunchKernelGGL(bk_transform_kernel<11>, dim3(npoly_per_w * nw, 2), dim3(64), 0, s, raw_polys, img, tw, nw,
                           scale[0], scale[1]);
    else
        hipLaunchKernelGGL(bk_transform_kernel<10>, dim3(npoly_per_w * nw, 2), dim3(64), 0, s, raw_polys, img, tw, nw,
                           scale[0], scale[1]);
}

// 2-wave form: N = 1024 only (the engine's admissibility order never picks it for another ring)
void launch_blind_rotate2(hipStream_t s, const DevParams &p, const DevKey &key, const int32_t *pool,
                          const RotDesc *rots, int count, int32_t *u_buf, int32_t *acc_dbg) {
    if (count <= 0 || p.N != 1024) return;
    hipLaunchKernelGGL(blind_rotate2_kernel<10>, dim3(count), dim3(128), 0, s, p, key, pool, rots, u_buf, acc_dbg);
}

// the LDS digit tables index by (D >> (shift - 3)) & mask (8-byte entries): digits of at most DIGIT_TAB_BITS bits whose
// lowest field starts at bit DIGIT_TAB_MIN_SHIFT or higher
static bool digit_table_usable(const DevParams &p) {
    return p.Bgbit <= DIGIT_TAB_BITS && p.digit_table != 0 && 32 - p.l * p.Bgbit >= DIGIT_TAB_MIN_SHIFT;
}

void launch_blind_rotate8(hipStream_t s, const DevParams &p, const DevKey &key, const int32_t *pool,
                          const RotDesc *rots, int count, int32_t *u_buf, int32_t *acc_dbg) {
    if (count <= 0 || p.N != 1024) return;
    if (digit_table_usable(p))
        hipLaunchKernelGGL((blind_rotate8_kernel<10, true>), dim3(count), dim3(512), 0, s, p, key, pool, rots, u_buf, acc_dbg);
    else
        hipLaunchKernelGGL((blind_rotate8_kernel<10, false>), dim3(count), dim3(512), 0, s, p, key, pool, rots, u_buf, acc_dbg);
}

void launch_blind_rotate_split(hipStream_t s, const DevParams &p, const DevKey &key, const int32_t *pool,
                               const RotDesc *rots, int count, int32_t *u_buf, int32_t *acc_dbg) {
    if (count <= 0) return;
#define BRS(LN, TM) hipLaunchKernelGGL((blind_rotate_split_kernel<LN, TM>), dim3(count), dim3(512), 0, s, p, key, pool, rots, u_buf, acc_dbg)
    // table mode: digit_table 1 = the widest the digits and the LDS budget allow (N = 2048: stage 0 and the first
    // radix-4 step for digits of at most 6 bits), 2 = stage 0 only, 0 = none
    const bool tab = digit_table_usable(p);
    if (p.N == 2048) {
        if (tab && p.digit_table == 1 && p.Bgbit <= SPLIT_TAB2_BITS) BRS(11, 2);
        else if (tab) BRS(11, 1);
        else BRS(11, 0);
    } else {
        if (tab) BRS(10, 1); else BRS(10, 0);
    }
#undef BRS
}

#ifdef TFHE_HIP_STAMPS
void read_stamps(unsigned long long *out, bool reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64);
    if (reset) { unsigned long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z); }
}
#endif

void launch_blind_rotate4(hipStream_t s, const DevParams &p, const DevKey &key, const int32_t *pool,
                          const RotDesc *rots, int count, int32_t *u_buf, int32_t *acc_dbg) {
    if (count <= 0 || p.N != 1024) return;
    if (digit_table_usable(p))
        hipLaunchKernelGGL((blind_rotate4_kernel<10, true>), dim3(count), dim3(256), 0, s, p, key, pool, rots, u_buf, acc_dbg);
    else
        hipLaunchKernelGGL((blind_rotate4_kernel<10, false>), dim3(count), dim3(256), 0, s, p, key, pool, rots, u_buf, acc_dbg);
}

// tile = 0 or splits <= 1: one workgroup per (gate, range) [keyswitch_kernel].  tile = 16 / 24 / 32 with the key switch of
// the built-in sets (t = 8, base 4) and ranges of at most 64 coefficients: the tiled kernels -- the index form, or the
// LDS-strip form (tile 16) when `index` is false.  Every form with splits > 1 leaves partial sums for ks_reduce_kernel.
void launch_keyswitch(hipStream_t s, const DevParams &p, const DevKey &key, const int32_t *u_buf,
                      const KsDesc *descs, int count, int32_t *pool, int splits, int32_t *partial, int tile, bool index) {
    if (count <= 0) return;
    const int threads = ((p.ct_stride / 4 + 63) / 64) * 64;      // one 16-byte lane per 4 output words: 128, 192 or 320
    if (splits <= 1 || !partial) {
        hipLaunchKernelGGL(keyswitch_kernel, dim3(count, 1), dim3(threads), 0, s, p, key, u_buf, descs, pool, nullptr);
        return;
    }
    const int range = (p.k * p.N + splits - 1) / splits;
    const bool tiled = (tile == 16 || ((tile == 24 || tile == 32) && index)) && count >= 2 * tile && p.ks_t == 8 &&
                       p.ks_basebit == 2 && range <= 64 && (threads == 128 || threads == 192 || threads == 320);
    const dim3 grid((count + (tiled ? tile : 1) - 1) / (tiled ? tile : 1), splits);
#define KS_FORM(K, GT)                                                                                                      \
    do {                                                                                                                    \
        if (threads == 128) hipLaunchKernelGGL((K<128, GT>), grid, dim3(128), 0, s, p, key, u_buf, descs, count, partial);  \
        else if (threads == 192) hipLaunchKernelGGL((K<192, GT>), grid, dim3(192), 0, s, p, key, u_buf, descs, count, partial); \
        else hipLaunchKernelGGL((K<320, GT>), grid, dim3(320), 0, s, p, key, u_buf, descs, count, partial);                 \
    } while (0)
    if (!tiled) hipLaunchKernelGGL(keyswitch_kernel, grid, dim3(threads), 0, s, p, key, u_buf, descs, pool, partial);
    else if (!index) KS_FORM(keyswitch_strip_kernel, 16);
    else if (tile == 16) KS_FORM(keyswitch_index_kernel, 16);
    else if (tile == 24) KS_FORM(keyswitch_index_kernel, 24);
    else KS_FORM(keyswitch_index_kernel, 32);
#undef KS_FORM
    hipLaunchKernelGGL(ks_reduce_kernel, dim3(count), dim3(threads), 0, s, p, descs, splits, partial, pool);
}

void launch_not(hipStream_t s, const DevParams &p, const NotDesc *descs, int count, int32_t *pool) {
    if (count <= 0) return;
    hipLaunchKernelGGL(not_kernel, dim3(count), dim3(256), 0, s, p, descs, pool);
}

void launch_negacyclic(hipStream_t s, const DevParams &p, const uint32_t *tw, const int32_t *ip,
                       const uint32_t *img, int32_t *res, int count) {
    if (count <= 0) return;
    if (p.br_variant == 2) {                 // through the split transforms
        if (p.N == 2048) hipLaunchKernelGGL(negacyclic_split_kernel<11>, dim3(count), dim3(256), 0, s, ip, img, tw, res);
        else hipLaunchKernelGGL(negacyclic_split_kernel<10>, dim3(count), dim3(256), 0, s, ip, img, tw, res);
        return;
    }
    if (p.N == 2048) hipLaunchKernelGGL(negacyclic_kernel<11>, dim3(count), dim3(128), 0, s, ip, img, tw, res);
    else hipLaunchKernelGGL(negacyclic_kernel<10>, dim3(count), dim3(128), 0, s, ip, img, tw, res);
}

}  // namespace tfhe_hip
