// SIZING CANDIDATE, not a product kernel: configs[4] (N = 2048, l = 3, Bg = 2^6) in POLYPHASE form.
//
//   a(X) = a_e(X^2) + X a_o(X^2),  Y = X^2,  S = Z[Y] / (Y^1024 + 1):
//   c_e = a_e b_e + Y a_o b_o,     c_o = a_e b_o + a_o b_e
//
// so a 2048-point negacyclic product is four 1024-point ones and the transforms are the N = 1024 ones of the 4-wave kernel
// (WaveNtt<10>, first radix-4 step by table: no stage 0 outside the radix-4 pairing -- what the split form pays for).
// Eight waves per rotation: wave (q, u, par) works modulo prime q on parity par of input polynomial u -- l forward
// transforms of 16 coefficients per lane -- and multiplies every transformed digit row against FOUR key rows, because its
// spectrum feeds both parities of both output polynomials:
//     par = e:  D_e . B_e  -> C_e(w),   D_e . B_o -> C_o(w)          par = o:  D_o . (Y B_o) -> C_e(w),   D_o . B_e -> C_o(w)
// (key image: B_e, B_o, Y B_o per row and output polynomial = 1.5 x today's 201 MB).  The four 64-bit sums are reduced
// once and ADDED into four LDS sum buffers per prime (ds_add_u32: integer adds commute, the result is exact whatever the
// order; three send buffers per wave would need 186 KB of LDS); wave (q, w, par) then runs ONE 1024-point inverse transform
// of C_par(w), shares the CRT with wave (1 - q, w, par) and updates coefficients 2 j + par of accumulator polynomial w.
//
// The dataflow is complete (every load, table read, product, reduction, exchange and accumulator update is there and
// feeds the output), so the compiler's register allocation and the static instruction mix are those a working kernel
// would have; the key-image addressing assumes the 1.5 x layout.  It is NOT run and NOT parity-tested: it exists to be
// compiled by tools/sizing/p2048_polyphase.sh, which reads VGPRs / spills / LDS / the per-step VALU mix off the listing.
#include "../../peba1_amd/csrc/kernels.hpp"
#include "../../peba1_amd/csrc/ntt_wave.hpp"

namespace tfhe_hip {

struct PolyLds {
    using NTT = WaveNtt<10>;
    uint32_t acc[2][3 * 2048];                     // the accumulator, three signed runs per polynomial (as AccLds<11>)
    uint32_t scr[8][NTT::SCRATCH_WORDS];           // per wave: transpose scratch, then the residues its CRT partner reads
    uint32_t sum[2][2][2][1024];                   // [prime][output polynomial][parity]: the reduced row sums, added by 4 waves
    uint16_t bar[1024 + 8];
    alignas(8) uint32_t dtab[2][5 * DIGIT_TAB];
    uint4 ft1[2][64 >> NTT::LC][NTT::FwdTw1::IMAGE16];
    uint4 ft2[2][64][NTT::FwdTw2::IMAGE16];
};

__device__ __forceinline__ PrimeCtx ctx_of(int q, const uint32_t *tw) {
    constexpr int n_ring = 1024;
    PrimeCtx c;
    c.P = q ? NTT_P1 : NTT_P0;
    c.pinv = q ? NTT_PINV1 : NTT_PINV0;
    c.rmod = q ? NTT_R[1] : NTT_R[0];
    c.wf = tw + (size_t)(q * 2 + 0) * n_ring;
    c.wi = tw + (size_t)(q * 2 + 1) * n_ring;
    const uint4 *quads = reinterpret_cast<const uint4 *>(tw + (size_t)4 * n_ring);
    c.qf = quads + (size_t)(q * 2 + 0) * (n_ring / 2);
    c.qi = quads + (size_t)(q * 2 + 1) * (n_ring / 2);
    c.dtab = nullptr; c.fw1 = nullptr; c.fw2 = nullptr;
    return c;
}

__global__ __launch_bounds__(512, 2) void blind_rotate_polyphase_candidate(DevParams p, DevKey key, const int32_t *__restrict__ pool,
                                                                           const RotDesc *__restrict__ rots,
                                                                           int32_t *__restrict__ u_buf) {
    using NTT = WaveNtt<10>;
    constexpr int M = 1024, REGS = NTT::REGS, G4 = REGS / 4, HALF = REGS / 2;
    __shared__ __align__(16) PolyLds sh;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wv & 1, u = (wv >> 1) & 1, par = wv >> 2;
    const int lane = tid & 63;
    PrimeCtx c = ctx_of(q, key.tw);
    uint32_t *scr = sh.scr[wv];
    const int n = p.n;
    const RotDesc rd = rots[blockIdx.x];
    {   // prelude + modulus switch to Z_4096
        const int32_t *A = pool + (size_t)rd.slot_a * p.ct_stride, *B = pool + (size_t)rd.slot_b * p.ct_stride;
        for (int i = tid; i <= n; i += 512) {
            uint32_t t = (uint32_t)rd.sa * (uint32_t)A[i] + (uint32_t)rd.sb * (uint32_t)B[i];
            if (i == n) t += (uint32_t)rd.c0;
            sh.bar[i] = (uint16_t)((t + (1u << 19)) >> 20);
        }
    }
    NTT::build_digit_table(sh.dtab[q], c, p.Bgbit, ((wv >> 1) << 6) | lane, 256);
    c.dtab = sh.dtab[q];
    if (wv < 2) {
        typename NTT::FwdTw1 a; a.load(c, lane);
        if ((lane & ((1 << NTT::LC) - 1)) == 0) a.to_image(sh.ft1[q][lane >> NTT::LC]);
        typename NTT::FwdTw2 b; b.load(c, lane);
        b.to_image(sh.ft2[q][lane]);
    }
    c.fw1 = sh.ft1[q][lane >> NTT::LC];
    c.fw2 = sh.ft2[q][lane];
#pragma unroll
    for (int k = 0; k < 2 * 2 * 2 * M / 512; ++k) (&sh.sum[0][0][0][0])[k * 512 + tid] = 0u;
    __syncthreads();
    if (q == 0) {
        const int barb = sh.bar[n];
#pragma unroll
        for (int r = 0; r < REGS; ++r) {
            const int j = 2 * (r * 64 + lane) + par, idx = (j + barb) & 4095;
            const uint32_t v = u == 0 ? 0u : ((idx & 2048) ? (uint32_t)(-p.mu) : (uint32_t)p.mu);
            sh.acc[u][j] = v; sh.acc[u][2048 + j] = 0u - v; sh.acc[u][4096 + j] = v;
        }
    }
    __syncthreads();
    typename NTT::FwdTw0 t0;
    t0.load(c, lane);
    const int width = p.Bgbit;
    for (int i = 0; i < n; ++i) {
        const int abar = __builtin_amdgcn_readfirstlane((int)sh.bar[i]);
        if (abar == 0) continue;
        // D = coefficients 2 (64 r + lane) + par of (X^abar - 1) ACC_u: this wave's parity only (the split form computes both halves)
        uint32_t D[REGS];
        {
            const uint32_t base = (uint32_t)(2 * lane + par - abar) & 4095u;
            const uint32_t *rot = sh.acc[u] + base, *own_neg = sh.acc[u] + 2048 + 2 * lane + par;
#pragma unroll
            for (int r = 0; r < REGS; ++r) D[r] = (rot[r * 128] + own_neg[r * 128] + p.decomp_offset) ^ p.decomp_offset;
        }
        // four 64-bit sums: [output polynomial w][parity of the output this spectrum feeds]
        int64_t acc[2][2][REGS];
        auto row = [&](int jj, auto first) {
            const int prow = u * p.l + jj;
            // key image, 1.5 x layout: [step][row][prime][output polynomial][B_e, B_o, Y B_o][1024]
            const uint4 *bp = reinterpret_cast<const uint4 *>(key.bk_img + ((size_t)((size_t)i * p.kpl + prow) * 2 + q) * 2 * 3 * M) + lane;
            const int shift = 32 - (jj + 1) * width;
            int32_t x[REGS];
            NTT::template forward_digits<true, true>(x, D, shift, width, c, scr, lane, t0);
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                // par = e: B_e -> C_e, B_o -> C_o;  par = o: Y B_o -> C_e, B_e -> C_o   (image index 0, 1, 2 = B_e, B_o, Y B_o)
                const int img_e = par ? 2 : 0, img_o = par ? 0 : 1;
                uint4 be[G4], bo[G4];
#pragma unroll
                for (int g = 0; g < G4; ++g) be[g] = bp[(w * 3 + img_e) * (M / 4) + g * 64];
#pragma unroll
                for (int g = 0; g < G4; ++g) bo[g] = bp[(w * 3 + img_o) * (M / 4) + g * 64];
#pragma unroll
                for (int g = 0; g < G4; ++g) {
                    const int32_t e4[4] = {(int32_t)be[g].x, (int32_t)be[g].y, (int32_t)be[g].z, (int32_t)be[g].w};
                    const int32_t o4[4] = {(int32_t)bo[g].x, (int32_t)bo[g].y, (int32_t)bo[g].z, (int32_t)bo[g].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        if constexpr (decltype(first)::value) { acc[w][0][r] = (int64_t)x[r] * e4[e]; acc[w][1][r] = (int64_t)x[r] * o4[e]; }
                        else { acc[w][0][r] += (int64_t)x[r] * e4[e]; acc[w][1][r] += (int64_t)x[r] * o4[e]; }
                    }
                }
            }
        };
        row(0, std::true_type{});
#pragma unroll 1
        for (int jj = 1; jj < p.l; ++jj) row(jj, std::false_type{});
        // reduce once, add into the sum buffers of (q, w, parity): four waves add into each (u = 0, 1 x par = e, o)
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int po = 0; po < 2; ++po) {
                uint32_t *dst = sh.sum[q][w][po] + NTT::REGS * lane;       // spectrum slot REGS * lane + reg (layout L2)
#pragma unroll
                for (int r = 0; r < REGS; ++r) atomicAdd(dst + r, (uint32_t)mont_redc(acc[w][po][r], c.P, c.pinv));
            }
        typename NTT::InvTw2 t2;
        t2.load(c, lane);
        lds_barrier();
        // wave (q, w = u, par): inverse transform of C_par(w) modulo q; clears the sum buffer behind itself
        int32_t t[REGS];
        {
            uint32_t *src = sh.sum[q][u][par] + NTT::REGS * lane;
#pragma unroll
            for (int g = 0; g < G4; ++g) {
                const uint4 v = reinterpret_cast<const uint4 *>(src)[g];
                t[4 * g] = (int32_t)v.x; t[4 * g + 1] = (int32_t)v.y; t[4 * g + 2] = (int32_t)v.z; t[4 * g + 3] = (int32_t)v.w;
                reinterpret_cast<uint4 *>(src)[g] = make_uint4(0, 0, 0, 0);
            }
        }
        NTT::inverse(t, c, scr, lane, t2);
        // CRT shared with wave (1 - q, u, par): wave q recombines registers [q HALF, (q + 1) HALF)
        const uint32_t *ox = sh.scr[wv ^ 1];
#pragma unroll
        for (int r = 0; r < HALF; ++r) scr[r * 64 + lane] = (uint32_t)t[q ? r : HALF + r];
        lds_barrier();
#pragma unroll
        for (int r = 0; r < HALF; ++r) {
            const int rr = q ? HALF + r : r;
            const int j = 2 * (rr * 64 + lane) + par;
            const uint32_t inc = q ? crt_signed_to_torus((int32_t)ox[r * 64 + lane], t[rr]) : crt_signed_to_torus(t[rr], (int32_t)ox[r * 64 + lane]);
            const uint32_t v = sh.acc[u][j] + inc;
            sh.acc[u][j] = v; sh.acc[u][2048 + j] = 0u - v; sh.acc[u][4096 + j] = v;
        }
        lds_barrier();
    }
    int32_t *uo = u_buf + (size_t)rd.u_index * p.u_stride;
    for (int j = tid; j < 2048; j += 512) uo[j] = (int32_t)(j == 0 ? sh.acc[0][0] : 0u - sh.acc[0][2048 - j]);
    if (tid == 0) uo[2048] = (int32_t)sh.acc[1][0];
}

}  // namespace tfhe_hip
