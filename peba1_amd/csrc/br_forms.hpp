// br_forms.hpp -- which blind-rotate kernel forms may run a parameter set (host side, no device code).
//
// Every form of the kernel (kernels.hip) works on signed lazy representatives and relies on magnitude bounds
// that depend on the gadget (l, Bgbit): the forward transforms' outputs, the 64-bit row sums, and -- the one
// that binds -- the sum of partial results fed to the inverse transform, which accepts |x| < 4P
// (ntt_wave.hpp make_inv_plan: every sum of four inputs must stay below 2^31).  The kernels are exact only
// inside those bounds, so a form is admissible for a parameter set only if its own worst-case bound holds;
// the built-in sets (l = 2, 3) sit inside all of them, a custom set need not.  The recurrences
// are the interval arithmetic of tools/ntt_model_r4.py (fwd_bound, inv_schedule), in units of the larger prime:
//   Montgomery reduction of T:       |r| <= |T| / 2^32 + P/2
//   forward radix-4 step:            b <- b (1 + 3q) + 1,   radix-2 stage: b <- b (1 + q) + 1/2,   q = P / 2^32
//   a digit-table entry d w mod P:   |e| <= 1/2 + |d| q     (split form, mode 2: centred, |e| <= 1/2)
#pragma once
#include <cstdint>

namespace tfhe_hip {

enum BrForm : int {
    BR_FORM_WIDE4 = 0,    // 4 waves, 64-bit sums kept and sent (N = 1024)
    BR_FORM_SPLIT = 1,    // 8 waves, half transforms, 64-bit sums (N = 1024 or 2048)
    BR_FORM_WAVE8 = 2,    // 8 waves for narrow launches (N = 1024): rows split over two waves
    BR_FORM_WAVE2 = 3,    // 2 waves (N = 1024): one reduction of all 2 l rows, no tables -- the widest gadget range
    BR_FORM_COUNT = 4
};

namespace br_forms_detail {
constexpr double kP = 134176769.0, kQ = kP / 4294967296.0;

// worst-case |x| / P after the forward transform of `logn` stages laid out as the wave NTT lays them out
// (passes of RB, RB, LC stages; radix-4 steps from the start of each pass, a radix-2 stage where one is left),
// starting from magnitude b; skip_first = the first radix-4 step was done by table (b is its output bound)
inline double forward_bound(int logn, double b, bool skip_first) {
    const int rb = logn - 6;
    const int len[3] = {rb, rb, logn - 2 * rb};
    bool first = true;
    for (int p = 0; p < 3; ++p) {
        int cnt = len[p];
        while (cnt >= 2) {
            if (!(first && skip_first)) b = b * (1.0 + 3.0 * kQ) + 1.0;
            first = false;
            cnt -= 2;
        }
        if (cnt) { b = b * (1.0 + kQ) + 0.5; first = false; }
    }
    return b;
}
}  // namespace br_forms_detail

// tables: 0 = the forward transforms multiply; 1 = the widest table mode the form has for these digits
// (4-wave / 8-wave: first radix-4 step, Bgbit <= 7; split: stage 0 and, at Bgbit <= 6 and N = 2048, the first
// radix-4 step too); 2 = split form: stage-0 table only.  Mirrors kernels.hip digit_table_usable / launchers.
inline bool br_form_admissible(int form, int N, int l, int Bgbit, int tables) {
    using namespace br_forms_detail;
    if (l < 1 || Bgbit < 1 || Bgbit > 12 || l * Bgbit > 32) return false;
    const int logn = N == 1024 ? 10 : N == 2048 ? 11 : 0;
    if (!logn) return false;
    const double digit = (double)(1 << (Bgbit - 1)) / kP;              // |d| / P
    const double entry = 0.5 + (double)(1 << (Bgbit - 1)) * kQ / kP;   // |d w mod P| / P, Montgomery output
    // (the lowest digit field must start at bit 3 or higher: ntt_wave.hpp DIGIT_TAB_MIN_SHIFT, 8-byte table entries)
    const bool tab_ok = tables != 0 && Bgbit <= 7 && 32 - l * Bgbit >= 3;
    double F;                                                          // forward outputs / P
    if (form == BR_FORM_SPLIT) {
        // stage 0 on digits (x = d_lo +- W d_hi), then an (N/2)-point transform
        const bool tab2 = tab_ok && tables == 1 && N == 2048 && Bgbit <= 6;
        if (tab2) F = forward_bound(logn - 1, 7 * 0.5 + digit, true);             // seven centred entries + a digit
        else F = forward_bound(logn - 1, digit + entry, false);
    } else {
        if (form == BR_FORM_WAVE2 || !tab_ok) F = forward_bound(logn, digit, false);
        else F = forward_bound(logn, digit + 3 * entry, true);                     // x0 + A + S: three entries
    }
    // 64-bit sums: up to 2 l rows of |x| < F P times a key word < P
    if (2.0 * l * F * kP * kP >= 9.2e18) return false;
    const double wide = l * F * kQ + 0.5;               // l rows summed in 64 bits, reduced once
    double into_inverse;
    switch (form) {
    case BR_FORM_WIDE4: if (N != 1024) return false; into_inverse = 2 * wide; break;
    case BR_FORM_SPLIT: into_inverse = 2 * wide; break;
    case BR_FORM_WAVE8:
        if (N != 1024 || l < 2) return false;
        into_inverse = 2 * ((l - 1) * F * kQ + 0.5) + 2 * (F * kQ + 0.5);          // A's two sums + B's two sums
        break;
    case BR_FORM_WAVE2: if (N != 1024) return false; into_inverse = 2 * l * F * kQ + 0.5; break;
    default: return false;
    }
    return into_inverse < 4.0;
}

}  // namespace tfhe_hip
