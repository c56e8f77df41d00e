// circuits.cpp -- the encrypted integer circuits of lab-incert/peba1, written
// against the public tfhe gate API only (include/tfhe/*.h).
//
// Each function issues exactly the gate sequence of the reference function it
// names (same gates, same operands, same order, including the COPY / CONSTANT
// calls and the temporaries' allocation order), because ciphertext-level parity
// of a whole match depends on that sequence; tests/test_circuits_*.py compare
// the recorded sequences.  The code itself is this repo's: scoped temporaries
// instead of paired new/delete calls, one adder core shared by every caller.
#include <cstddef>
#include <vector>

#include "../../include/peba1_circuits.h"

namespace {

using CK = const TFheGateBootstrappingCloudKeySet;

// scoped array of samples (new_/delete_gate_bootstrapping_ciphertext_array)
class Tmp {
public:
    Tmp(int count, CK *ck) : n_(count), p_(new_gate_bootstrapping_ciphertext_array(count, ck->params)) {}
    ~Tmp() { delete_gate_bootstrapping_ciphertext_array(n_, p_); }
    Tmp(const Tmp &) = delete;
    Tmp &operator=(const Tmp &) = delete;
    LweSample *operator+(int i) const { return p_ + i; }
    operator LweSample *() const { return p_; }
private:
    int n_;
    LweSample *p_;
};

// reference release order differs from reverse-construction order in places;
// gate parity does not depend on frees, so RAII order is used throughout.

void set_zero(LweSample *x, int from, int to, CK *ck) {
    for (int i = from; i < to; ++i) bootsCONSTANT(x + i, 0, ck);
}
void copy_bits(LweSample *dst, const LweSample *src, int count, CK *ck) {
    for (int i = 0; i < count; ++i) bootsCOPY(dst + i, src + i, ck);
}

}  // namespace

extern "C" {

// Math.cpp:27-50
void peba1_add_1bit(LweSample *result, LweSample *a, LweSample *b, LweSample *carry, CK *ck) {
    Tmp t(1, ck), saved(1, ck), cin(1, ck);
    bootsCOPY(cin, carry, ck);
    bootsXOR(t, a, b, ck);              // a ^ b
    bootsXOR(result, t, carry, ck);     // sum
    bootsAND(t, a, b, ck);
    bootsCOPY(saved, t, ck);            // a & b
    bootsAND(t, a, cin, ck);
    bootsXOR(carry, saved, t, ck);      // (a&b) ^ (a&cin)
    bootsAND(t, cin, b, ck);
    bootsXOR(saved, carry, t, ck);      // ... ^ (cin&b)
    bootsCOPY(carry, saved, ck);
}

// Math.cpp:54-67
void peba1_add_nbit(LweSample *result, LweSample *a, LweSample *b, LweSample *carry, int bitsize, CK *ck) {
    Tmp running(1, ck);
    bootsCONSTANT(running, 0, ck);
    for (int i = 0; i < bitsize; ++i) peba1_add_1bit(result + i, a + i, b + i, running, ck);
    bootsCOPY(carry, running, ck);
}

// Math.cpp:71-93
void peba1_twos_complement(LweSample *result, LweSample *a, int bitsize, CK *ck) {
    Tmp one(bitsize + 1, ck), dropped_carry(1, ck);
    bootsCONSTANT(one, 1, ck);
    set_zero(one, 1, bitsize + 1, ck);
    Tmp flipped(bitsize + 1, ck);
    for (int i = 0; i < bitsize; ++i) bootsXOR(flipped + i, a + i, one, ck);   // one's complement
    peba1_add_nbit(result, flipped, one, dropped_carry, bitsize, ck);         // + 1
}

// Math.cpp:97-119
void peba1_abs(LweSample *result, LweSample *a, int bitsize, CK *ck) {
    Tmp sign(bitsize, ck), sum(bitsize + 1, ck), dropped_carry(1, ck);
    for (int i = 0; i < bitsize; ++i) bootsCOPY(sign + i, a + (bitsize - 1), ck);
    peba1_add_nbit(sum, a, sign, dropped_carry, bitsize, ck);
    for (int i = 0; i < bitsize; ++i) bootsXOR(result + i, sum + i, sign + i, ck);
}

// Math.cpp:123-180
void peba1_sub_nbit(LweSample *result, LweSample *a, LweSample *b, int bitsize, CK *ck) {
    const int w = bitsize + 1;
    Tmp minus_b(w, ck), diff(w + 1, ck), wide(w, ck), masked(w, ck), negated(w, ck), borrow(1, ck);

    copy_bits(wide, a, bitsize, ck);
    bootsCONSTANT(wide + bitsize, 0, ck);                     // a >= 0: sign bit 0
    peba1_twos_complement(minus_b, b, bitsize, ck);
    bootsCONSTANT(minus_b + bitsize, 1, ck);                  // -b < 0: sign bit 1
    peba1_add_nbit(diff, wide, minus_b, borrow, w, ck);       // a - b, carry says a >= b

    // branch-free |a-b|: keep diff when the carry is 1, negate it when it is 0
    for (int i = 0; i < w; ++i) bootsCOPY(wide + i, borrow, ck);
    for (int i = 0; i < w; ++i) bootsNOT(minus_b + i, borrow, ck);
    for (int i = 0; i < w; ++i) bootsAND(masked + i, diff + i, minus_b + i, ck);
    peba1_twos_complement(negated, masked, w, ck);
    for (int i = 0; i < w; ++i) bootsAND(masked + i, diff + i, wide + i, ck);
    for (int i = 0; i < w; ++i) bootsOR(result + i, negated + i, masked + i, ck);
}

// Math.cpp:183-190
void peba1_shift_left(LweSample *result, LweSample *a, int bitsize, int n, CK *ck) {
    set_zero(result, 0, n, ck);
    for (int i = n; i < bitsize; ++i) bootsCOPY(result + i, a + (i - n), ck);
}
// Math.cpp:193-200
void peba1_shift_right(LweSample *result, LweSample *a, int bitsize, int n, CK *ck) {
    for (int i = 0; i < bitsize - n; ++i) bootsCOPY(result + i, a + (i + n), ck);
    set_zero(result, bitsize - n, bitsize, ck);
}
// Math.cpp:202-211
void peba1_shift_left_inplace(LweSample *a, int bitsize, int n, CK *ck) {
    Tmp old(bitsize, ck);
    copy_bits(old, a, bitsize, ck);
    peba1_shift_left(a, old, bitsize, n, ck);
}

// Math.cpp:214-250
void peba1_multiply(LweSample *result, LweSample *a, LweSample *b, int bitsize, CK *ck) {
    const int length = 23;                                    // fixed by the reference
    Tmp addend(length, ck), partial(length, ck), total(length, ck), dropped_carry(1, ck);
    for (int i = 0; i < length; ++i) {
        bootsCONSTANT(total + i, 0, ck);
        bootsCONSTANT(addend + i, 0, ck);
        bootsCONSTANT(partial + i, 0, ck);
    }
    for (int i = 0; i < bitsize; ++i) {
        set_zero(partial, 0, i, ck);
        for (int j = 0; j < bitsize; ++j) {
            bootsAND(partial + (j + i), a + j, b + i, ck);    // partial product row i, shifted by i
            bootsCOPY(addend + j, total + j, ck);
        }
        for (int j = bitsize; j < length; ++j) bootsCOPY(addend + j, total + j, ck);
        peba1_add_nbit(total, partial, addend, dropped_carry, length - 1, ck);
    }
    copy_bits(result, total, length, ck);
}

// Math.cpp:259-262
void peba1_compare_bit(LweSample *result, const LweSample *a, const LweSample *b, const LweSample *lsb_carry,
                       LweSample *tmp, CK *ck) {
    bootsXNOR(tmp, a, b, ck);
    bootsMUX(result, tmp, lsb_carry, a, ck);
}

// Math.cpp:265-286
void peba1_minimum(LweSample *result, LweSample *bit, const LweSample *a, const LweSample *b, int nb_bits, CK *ck) {
    Tmp state(2, ck);
    bootsCONSTANT(state, 0, ck);
    for (int i = 0; i < nb_bits; ++i) peba1_compare_bit(state, a + i, b + i, state, state + 1, ck);
    for (int i = 0; i < nb_bits; ++i) bootsMUX(result + i, state, b + i, a + i, ck);
    bootsCOPY(bit, state, ck);
    set_zero(bit, 1, nb_bits, ck);
}

// Math.cpp:333-369
void peba1_euclidean_distance(LweSample *result, LweSample *const *a, LweSample *const *b, int nslots, int bitsize,
                              CK *ck) {
    const int max_bitsize = 24;
    Tmp diff(bitsize + 1, ck), diff2(bitsize + 1, ck), square(max_bitsize, ck), sum(max_bitsize, ck),
        dropped_carry(1, ck);
    for (int i = 0; i < nslots; ++i) {
        peba1_sub_nbit(diff, b[i], a[i], bitsize, ck);
        copy_bits(diff2, diff, bitsize + 1, ck);
        peba1_multiply(square, diff, diff2, bitsize, ck);
        copy_bits(sum, result, max_bitsize - 1, ck);
        peba1_add_nbit(result, square, sum, dropped_carry, max_bitsize - 1, ck);
    }
}

// Math.cpp:379-387.  Like the reference, the distance accumulator is NOT cleared
// first: fresh samples from this library are trivial encryptions of phase 0
// (SURVEY D1), which is what makes this well defined.
void peba1_function_f(LweSample *result_b, LweSample *const *a, LweSample *const *b, int nslots,
                      LweSample *bound_match, int bitsize, CK *ck) {
    Tmp distance(bitsize * 3, ck), smaller(bitsize * 3, ck);
    peba1_euclidean_distance(distance, a, b, nslots, bitsize, ck);
    peba1_minimum(smaller, result_b, distance, bound_match, bitsize * 3, ck);
}

// Math.cpp:390-417, with tmp_r0 sized bitsize+1 (the reference writes bitsize+1
// samples into a bitsize-sample array)
void peba1_function_g(LweSample *result, LweSample *result_b, LweSample *r0, LweSample *r1, int bitsize, CK *ck) {
    Tmp one(bitsize, ck), dropped_carry(1, ck);
    bootsCONSTANT(one, 1, ck);
    set_zero(one, 1, bitsize, ck);
    Tmp not_b(bitsize + 1, ck);
    peba1_sub_nbit(not_b, one, result_b, bitsize, ck);        // 1 - b
    Tmp product(bitsize * 3, ck);
    peba1_multiply(product, not_b, r0, bitsize, ck);          // (1-b) * r0
    copy_bits(not_b, product, bitsize, ck);
    peba1_multiply(product, result_b, r1, bitsize, ck);       // b * r1
    peba1_add_nbit(result, not_b, product, dropped_carry, bitsize, ck);
}

// ---- slot-sharded match -------------------------------------
void peba1_partial_distance(LweSample *partial, LweSample *const *a, LweSample *const *b, int nslots, int bitsize,
                            CK *ck) {
    set_zero(partial, 0, 24, ck);
    peba1_euclidean_distance(partial, a, b, nslots, bitsize, ck);
}

void peba1_combine_and_compare(LweSample *result_b, LweSample *const *partials, int nparts, LweSample *bound_match,
                               CK *ck) {
    const int width = 24;
    // pairwise tree of 23-bit adders over the partial sums (depth log2(nparts))
    std::vector<Tmp *> owned;
    std::vector<LweSample *> cur(partials, partials + nparts);
    Tmp dropped_carry(1, ck);
    while (cur.size() > 1) {
        std::vector<LweSample *> next;
        for (size_t i = 0; i + 1 < cur.size(); i += 2) {
            Tmp *s = new Tmp(width, ck);
            owned.push_back(s);
            set_zero(*s, width - 1, width, ck);
            peba1_add_nbit(*s, cur[i], cur[i + 1], dropped_carry, width - 1, ck);
            next.push_back(*s);
        }
        if (cur.size() & 1) next.push_back(cur.back());
        cur.swap(next);
    }
    Tmp smaller(width, ck);
    peba1_minimum(smaller, result_b, cur[0], bound_match, width, ck);
    for (Tmp *t : owned) delete t;
}

// ---- Hamming distance + threshold (not in the reference) ----------
int peba1_hamming_count_bits(int nbits) {
    int w = 1;
    while ((1 << w) <= nbits) ++w;
    return w;
}

void peba1_hamming_distance(LweSample *count, LweSample *a, LweSample *b, int nbits, CK *ck) {
    struct Term { Tmp *bits; int width; };
    std::vector<Term> cur;
    cur.reserve((size_t)nbits);
    for (int i = 0; i < nbits; ++i) {                       // 1-bit terms: a_i XOR b_i
        Tmp *t = new Tmp(1, ck);
        bootsXOR(*t, a + i, b + i, ck);
        cur.push_back(Term{t, 1});
    }
    Tmp carry(1, ck);
    while (cur.size() > 1) {                                // pairwise sums, widths grow by one per level
        std::vector<Term> next;
        for (size_t i = 0; i + 1 < cur.size(); i += 2) {
            Term &x = cur[i], &y = cur[i + 1];              // x.width >= y.width by construction
            const int w = x.width;
            Tmp padded(w, ck);
            copy_bits(padded, *y.bits, y.width, ck);
            set_zero(padded, y.width, w, ck);
            Tmp *sum = new Tmp(w + 1, ck);
            peba1_add_nbit(*sum, *x.bits, padded, carry, w, ck);
            bootsCOPY(*sum + w, carry, ck);
            delete x.bits;
            delete y.bits;
            next.push_back(Term{sum, w + 1});
        }
        if (cur.size() & 1) next.push_back(cur.back());
        cur.swap(next);
    }
    const int w = peba1_hamming_count_bits(nbits);          // drop structurally-zero top bits
    copy_bits(count, *cur[0].bits, cur[0].width < w ? cur[0].width : w, ck);
    set_zero(count, cur[0].width < w ? cur[0].width : w, w, ck);
    delete cur[0].bits;
}

void peba1_hamming_match(LweSample *result_b, LweSample *a, LweSample *b, int nbits, LweSample *bound_match, CK *ck) {
    const int w = peba1_hamming_count_bits(nbits);
    Tmp count(w, ck), smaller(w, ck);
    peba1_hamming_distance(count, a, b, nbits, ck);
    peba1_minimum(smaller, result_b, count, bound_match, w, ck);
}

}  // extern "C"
