// circuits_fast.cpp -- a depth- and gate-optimised Function_f.
//
// NOT the reference's gate sequence: same inputs, same decrypted result_b, different DAG.
// The reference evaluates sum_i (a_i - b_i)^2 > bound with a 24-bit-wide generic multiplier
// per slot and a slot-by-slot ripple accumulation (Math.cpp:333-387): 215,544 bootstraps,
// dependency depth 377 for 128 slots x 8 bit.  Here:
//   * |a_i - b_i|: one borrow chain + conditional two's complement (5 + 3 gates per bit),
//   * its square as a squarer: the 28 cross products d_i d_j (i < j) at weight i+j+1 and the
//     bits d_i themselves at weight 2i -- no per-slot adder at all,
//   * every product bit of every slot goes into ONE carry-save column compressor (full adder
//     = 2 XOR + 1 MUX), bits combined earliest-first,
//   * one parallel-prefix (Sklansky) adder for the last two rows and a prefix comparator.
// For 128 x 8 bit: about 27,000 bootstraps and depth about 75.  Arithmetic is modulo
// 2^(3*bitsize), as in the reference.
//
// Written against the public gate API only (AND, ANDNY, ANDYN, OR, XOR, XNOR, MUX, NOT, COPY,
// CONSTANT), so it runs on upstream TFHE, on libtfhe-hip and on the test providers alike.
#include <algorithm>
#include <cstddef>
#include <vector>

#include "../../include/peba1_circuits.h"

namespace {

using CK = const TFheGateBootstrappingCloudKeySet;

// a wire: a sample, or the constant 0 (s == nullptr); depth = gate levels below it
struct Net {
    LweSample *s = nullptr;
    int depth = 0;
    bool zero() const { return s == nullptr; }
};

class Builder {
public:
    explicit Builder(CK *ck) : ck_(ck) {}
    ~Builder() {
        for (auto &c : chunks_) delete_gate_bootstrapping_ciphertext_array(CHUNK, c);
    }
    Net input(LweSample *s) const { return Net{s, 0}; }

    Net g_xor(Net a, Net b) { return a.zero() ? b : b.zero() ? a : gate(bootsXOR, a, b); }
    Net g_or(Net a, Net b) { return a.zero() ? b : b.zero() ? a : gate(bootsOR, a, b); }
    Net g_and(Net a, Net b) { return (a.zero() || b.zero()) ? Net{} : gate(bootsAND, a, b); }
    Net g_andny(Net a, Net b) { return b.zero() ? Net{} : a.zero() ? b : gate(bootsANDNY, a, b); }   // ~a & b
    Net g_andyn(Net a, Net b) { return a.zero() ? Net{} : b.zero() ? a : gate(bootsANDYN, a, b); }   // a & ~b
    Net g_xnor(Net a, Net b) {                                                                      // ~(a ^ b)
        if (a.zero() && b.zero()) return one();
        if (a.zero()) return g_not(b);
        if (b.zero()) return g_not(a);
        return gate(bootsXNOR, a, b);
    }
    Net g_not(Net a) {
        if (a.zero()) return one();
        Net r{fresh(), a.depth};
        bootsNOT(r.s, a.s, ck_);
        return r;
    }
    // sel ? t : f, all three real wires
    Net g_mux(Net sel, Net t, Net f) {
        Net r{fresh(), std::max(sel.depth, std::max(t.depth, f.depth)) + 1};
        bootsMUX(r.s, sel.s, t.s, f.s, ck_);
        return r;
    }
    Net one() {
        Net r{fresh(), 0};
        bootsCONSTANT(r.s, 1, ck_);
        return r;
    }
    CK *ck() const { return ck_; }

private:
    using Gate2 = void (*)(LweSample *, const LweSample *, const LweSample *, CK *);
    Net gate(Gate2 f, Net a, Net b) {
        Net r{fresh(), std::max(a.depth, b.depth) + 1};
        f(r.s, a.s, b.s, ck_);
        return r;
    }
    LweSample *fresh() {
        if (used_ == CHUNK) {
            chunks_.push_back(new_gate_bootstrapping_ciphertext_array(CHUNK, ck_->params));
            used_ = 0;
        }
        return chunks_.back() + used_++;
    }
    static constexpr int CHUNK = 4096;
    CK *ck_;
    std::vector<LweSample *> chunks_;
    int used_ = CHUNK;
};

// d = |a - b| for two unsigned `bits`-bit numbers
std::vector<Net> abs_difference(Builder &B, LweSample *a, LweSample *b, int bits) {
    std::vector<Net> diff(bits);
    Net borrow;                                            // constant 0
    for (int i = 0; i < bits; ++i) {
        const Net ai = B.input(a + i), bi = B.input(b + i);
        const Net x = B.g_xor(ai, bi);
        diff[i] = B.g_xor(x, borrow);
        // borrow out = (~a & b) | (~(a ^ b) & borrow in)
        borrow = B.g_or(B.g_andny(ai, bi), B.g_andny(x, borrow));
    }
    // a < b: two's complement of diff.  The lowest bit is unchanged; above it
    // d_i = (diff_i ^ lt) ^ c_i with c_1 = ~diff_0 & lt, c_{i+1} = (diff_i ^ lt) & c_i.
    const Net lt = borrow;
    std::vector<Net> d(bits);
    d[0] = diff[0];
    Net carry = B.g_andny(diff[0], lt);
    for (int i = 1; i < bits; ++i) {
        const Net y = B.g_xor(diff[i], lt);
        d[i] = B.g_xor(y, carry);
        if (i + 1 < bits) carry = B.g_and(y, carry);
    }
    return d;
}

// reduce every column to at most two wires with full adders; carries out of the top column are
// dropped (arithmetic modulo 2^width)
void compress_columns(Builder &B, std::vector<std::vector<Net>> &cols) {
    const int width = (int)cols.size();
    for (;;) {
        bool any = false;
        std::vector<std::vector<Net>> next(width);
        for (int w = 0; w < width; ++w) {
            std::vector<Net> &c = cols[w];
            std::stable_sort(c.begin(), c.end(), [](const Net &x, const Net &y) { return x.depth < y.depth; });
            size_t i = 0;
            for (; i + 3 <= c.size() && c.size() > 2; i += 3) {
                any = true;
                const Net x = B.g_xor(c[i], c[i + 1]);
                next[w].push_back(B.g_xor(x, c[i + 2]));
                if (w + 1 < width) next[w + 1].push_back(B.g_mux(x, c[i + 2], c[i]));   // majority
            }
            for (; i < c.size(); ++i) next[w].push_back(c[i]);
        }
        cols.swap(next);
        if (!any) break;
    }
}

// sum of two rows (each column holds 0, 1 or 2 wires) with a Sklansky parallel-prefix adder
std::vector<Net> prefix_add(Builder &B, const std::vector<std::vector<Net>> &cols) {
    const int width = (int)cols.size();
    std::vector<Net> p(width), G(width), P(width);
    for (int i = 0; i < width; ++i) {
        const Net x = cols[i].size() > 0 ? cols[i][0] : Net{}, y = cols[i].size() > 1 ? cols[i][1] : Net{};
        p[i] = B.g_xor(x, y);
        G[i] = B.g_and(x, y);
        P[i] = p[i];
    }
    for (int l = 0; (1 << l) < width; ++l) {
        for (int i = 0; i < width; ++i) {
            if (!((i >> l) & 1)) continue;
            const int j = ((i >> l) << l) - 1;             // top of the block below
            // (G, P)_i o (G, P)_j; generate and propagate-and-generate exclude each other
            G[i] = B.g_xor(G[i], B.g_and(P[i], G[j]));
            if ((i >> (l + 1)) != 0) P[i] = B.g_and(P[i], P[j]);   // span not yet down to bit 0
        }
    }
    std::vector<Net> sum(width);
    sum[0] = p[0];
    for (int i = 1; i < width; ++i) sum[i] = B.g_xor(p[i], G[i - 1]);
    return sum;
}

// a > b over `width` bits, parallel prefix on (greater, equal) pairs
Net greater_than(Builder &B, const std::vector<Net> &a, LweSample *b, int width) {
    struct GE { Net gt, eq; };
    std::vector<GE> cur(width);
    for (int i = 0; i < width; ++i) {
        const Net bi = B.input(b + i);
        cur[i] = GE{B.g_andyn(a[i], bi), B.g_xnor(a[i], bi)};
    }
    while (cur.size() > 1) {                               // cur[0] is the least significant group
        std::vector<GE> next;
        for (size_t i = 0; i + 1 < cur.size(); i += 2) {
            const GE &lo = cur[i], &hi = cur[i + 1];
            GE m;
            m.gt = B.g_or(hi.gt, B.g_and(hi.eq, lo.gt));
            if (cur.size() > 2) m.eq = B.g_and(hi.eq, lo.eq);
            next.push_back(m);
        }
        if (cur.size() & 1) next.push_back(cur.back());
        cur.swap(next);
    }
    return cur[0].gt;
}

std::vector<Net> squared_distance(Builder &B, LweSample *const *a, LweSample *const *b, int nslots, int bitsize,
                                  int width) {
    std::vector<std::vector<Net>> cols(width);
    for (int s = 0; s < nslots; ++s) {
        const std::vector<Net> d = abs_difference(B, a[s], b[s], bitsize);
        for (int i = 0; i < bitsize; ++i) {
            if (2 * i < width) cols[2 * i].push_back(d[i]);                       // d_i d_i = d_i
            for (int j = i + 1; j < bitsize; ++j)
                if (i + j + 1 < width) cols[i + j + 1].push_back(B.g_and(d[i], d[j]));   // 2 d_i d_j
        }
    }
    compress_columns(B, cols);
    return prefix_add(B, cols);
}

void store(Builder &B, LweSample *dst, Net v) {
    if (v.zero()) bootsCONSTANT(dst, 0, B.ck());
    else bootsCOPY(dst, v.s, B.ck());
}

}  // namespace

extern "C" {

void peba1_euclidean_distance_fast(LweSample *result, LweSample *const *a, LweSample *const *b, int nslots,
                                   int bitsize, CK *ck) {
    const int width = 3 * bitsize;
    Builder B(ck);
    const std::vector<Net> dist = squared_distance(B, a, b, nslots, bitsize, width);
    for (int i = 0; i < width; ++i) store(B, result + i, dist[i]);
}

// Rank-0 tail of the slot-sharded match with the blocks above: the partial sums of all ranks go
// into one carry-save column compressor, then one prefix adder and the prefix comparator --
// depth ~20 for 8 partial sums where the pairwise tree of the reference's ripple adders plus its
// bit-serial comparator (peba1_combine_and_compare) takes ~290 levels.  Same decrypted bit.
void peba1_combine_and_compare_fast(LweSample *result_b, LweSample *const *partials, int nparts, LweSample *bound_match,
                                    CK *ck) {
    const int width = 24;
    Builder B(ck);
    std::vector<std::vector<Net>> cols(width);
    for (int r = 0; r < nparts; ++r)
        for (int i = 0; i < width - 1; ++i) cols[i].push_back(B.input(partials[r] + i));   // 23-bit sums
    compress_columns(B, cols);
    std::vector<Net> dist = prefix_add(B, cols);
    dist[width - 1] = Net{};                               // the reference's accumulator is 23 bits wide; bit 23 stays 0
    store(B, result_b, greater_than(B, dist, bound_match, width));
    for (int i = 1; i < width; ++i) bootsCONSTANT(result_b + i, 0, ck);
}

void peba1_function_f_fast(LweSample *result_b, LweSample *const *a, LweSample *const *b, int nslots,
                           LweSample *bound_match, int bitsize, CK *ck) {
    const int width = 3 * bitsize;
    Builder B(ck);
    const std::vector<Net> dist = squared_distance(B, a, b, nslots, bitsize, width);
    store(B, result_b, greater_than(B, dist, bound_match, width));
    for (int i = 1; i < width; ++i) bootsCONSTANT(result_b + i, 0, ck);    // as minimum() leaves them, Math.cpp:282-285
}

}  // extern "C"
