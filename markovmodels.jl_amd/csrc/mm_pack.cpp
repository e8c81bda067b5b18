// mm_pack.cpp -- see mm_pack.h.
#include "mm_pack.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <utility>

namespace mm {

namespace {

// (g, R) class of a row with d stored entries: the smallest power-of-two lane
// group that leaves at most 4 arcs per lane; beyond 256 arcs, a full wave and
// R rounded up to a multiple of 4 (the kernels walk long rows in chunks of 4).
void row_class(int64_t d, int &log2g, int &R) {
    if (d <= 0) {
        log2g = 0;
        R = 1;
        return;
    }
    log2g = 0;
    while (log2g < 6 && (d + (int64_t(1) << log2g) - 1) / (int64_t(1) << log2g) > 4) ++log2g;
    int64_t g = int64_t(1) << log2g;
    R = int((d + g - 1) / g);
    if (R > 4) R = (R + 3) / 4 * 4;
}

int item_cost(int log2g, int R) { return 8 * R + 8 * log2g + 24; }

}  // namespace

Packed pack_rows(int64_t nrows, const std::vector<int64_t> &rowptr, const std::vector<int32_t> &col,
                 const std::vector<float> &val, const std::vector<int32_t> &row2pdf, float zero_w) {
    Packed out;
    out.nnz = rowptr[nrows];
    // bucket the rows by class, ascending row id inside a class
    std::map<std::pair<int, int>, std::vector<int64_t>> classes;
    for (int64_t r = 0; r < nrows; ++r) {
        int lg, R;
        row_class(rowptr[r + 1] - rowptr[r], lg, R);
        classes[{lg, R}].push_back(r);
    }
    struct Tmp {
        int log2g, R;
        std::vector<int64_t> rows;  // 64 >> log2g entries, -1 = padding
    };
    std::vector<Tmp> tmp;
    for (auto &kv : classes) {
        int lg = kv.first.first, R = kv.first.second;
        int per_item = 64 >> lg;
        auto &rows = kv.second;
        for (size_t i = 0; i < rows.size(); i += per_item) {
            Tmp t{lg, R, {}};
            for (int q = 0; q < per_item; ++q) t.rows.push_back(i + q < rows.size() ? rows[i + q] : -1);
            tmp.push_back(std::move(t));
        }
    }
    // register-eligible items (R <= 4) first, by decreasing cost; long rows last
    std::stable_sort(tmp.begin(), tmp.end(), [](const Tmp &a, const Tmp &b) {
        const bool la = a.R > 4, lb = b.R > 4;
        if (la != lb) return lb;
        return item_cost(a.log2g, a.R) > item_cost(b.log2g, b.R);
    });
    int64_t slot_row = 0;
    for (auto &t : tmp) {
        ItemMeta m;
        m.slot_row = uint32_t(slot_row);
        m.R = uint16_t(t.R);
        m.log2g = uint16_t(t.log2g);
        out.items.push_back(m);
        int g = 1 << t.log2g;
        size_t rbase = out.rowinfo.size();
        out.rowinfo.resize(rbase + 64);
        size_t sbase = out.slots.size();
        out.slots.resize(sbase + size_t(64) * t.R, Slot{0u, zero_w});
        for (int lane = 0; lane < 64; ++lane) {
            int64_t row = t.rows[lane >> t.log2g];
            RowInfo ri;
            ri.row = int32_t(row);
            ri.pdf = row >= 0 ? row2pdf[row] : 0;
            out.rowinfo[rbase + lane] = ri;
            if (row < 0) continue;
            int sub = lane & (g - 1);
            int64_t b = rowptr[row], e = rowptr[row + 1];
            // lane `sub` of the group takes arcs sub, sub+g, sub+2g, ...
            int k = 0;
            for (int64_t a = b + sub; a < e; a += g, ++k) {
                Slot s;
                s.col = uint32_t(col[a]);
                s.w = val[a];
                out.slots[sbase + size_t(k) * 64 + lane] = s;
            }
        }
        slot_row += t.R;
    }
    out.n_slot_rows = slot_row;
    return out;
}

QuadGeometry pick_quad_geometry(int64_t nquads) {
    // KQ multiples of 4 are avoided: lane tid stores its quad sums at tid * KQ + j, and
    // a stride that is a multiple of 4 dwords makes those stores 4-way bank conflicted.
    // Up to 13 quads per lane fit the 128-VGPR budget of 16 waves per CU; larger graphs use
    // 8 waves with twice the registers (measured on the 3032-state WSJ graph: 7.9 ms against
    // 10.7 ms for 16 waves with a streamed overflow).
    static const int kq16[] = {1, 2, 3, 5, 6, 7, 9, 10, 11, 13};
    static const int kq8[] = {15, 17, 19, 21, 23, 25, 27, 29};
    QuadGeometry g{29, 8};
    bool found = false;
    for (int k : kq16)
        if (int64_t(64) * 16 * k >= nquads) {
            g = QuadGeometry{k, 16};
            found = true;
            break;
        }
    if (!found)
        for (int k : kq8)
            if (int64_t(64) * 8 * k >= nquads) {
                g = QuadGeometry{k, 8};
                break;
            }
    g.NW = int(std::min<int64_t>(g.NW, std::max<int64_t>(1, (nquads + 64 * g.KQ - 1) / (64 * g.KQ))));
    return g;
}

namespace {

// LDS cycles of one ds_read_b32 wave instruction under the bank model of the CDNA4 LDS
// (two groups of 32 lanes, 32 banks of 4 bytes, identical addresses broadcast)
struct BankTable {
    // per bank: the distinct addresses present (small)
    std::vector<uint16_t> addr[32];
    int cost_of(uint16_t a) const {  // extra cycles this address would add
        const auto &v = addr[(a >> 2) & 31];
        for (uint16_t x : v)
            if (x == a) return 0;
        return int(v.size());
    }
    void add(uint16_t a) {
        auto &v = addr[(a >> 2) & 31];
        for (uint16_t x : v)
            if (x == a) return;
        v.push_back(a);
    }
    int cycles() const {
        size_t m = 1;
        for (auto &v : addr) m = std::max(m, v.size());
        return int(m);
    }
    int least_loaded_bank() const {
        int b = 0;
        for (int i = 1; i < 32; ++i)
            if (addr[i].size() < addr[b].size()) b = i;
        return b;
    }
};

// the same with multiplicities, so that addresses can be removed again (local search)
struct CountedBanks {
    struct E {
        uint16_t addr;
        uint16_t n;
    };
    std::vector<E> bank[32];
    void add(uint16_t a) {
        auto &v = bank[(a >> 2) & 31];
        for (auto &e : v)
            if (e.addr == a) {
                ++e.n;
                return;
            }
        v.push_back(E{a, 1});
    }
    void remove(uint16_t a) {
        auto &v = bank[(a >> 2) & 31];
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].addr == a) {
                if (--v[i].n == 0) {
                    v[i] = v.back();
                    v.pop_back();
                }
                return;
            }
    }
    int cycles() const {
        size_t m = 1;
        for (auto &v : bank) m = std::max(m, v.size());
        return int(m);
    }
    bool conflicted(uint16_t a) const { return bank[(a >> 2) & 31].size() > 1; }
};

}  // namespace

int64_t count_quads(int64_t nrows, const std::vector<int64_t> &rowptr) {
    int64_t n = 0;
    for (int64_t r = 0; r < nrows; ++r) n += (rowptr[r + 1] - rowptr[r] + 3) / 4;
    return n;
}

std::vector<uint16_t> reach_distance(int64_t n, const std::vector<int64_t> &rowptr, const std::vector<int32_t> &col,
                                     const std::vector<int32_t> &seeds) {
    std::vector<int64_t> d;
    d.assign(size_t(n), -1);
    std::vector<int32_t> frontier;
    for (int32_t s : seeds)
        if (s >= 0 && s < n && d[s] < 0) {
            d[s] = 0;
            frontier.push_back(s);
        }
    for (int64_t level = 1; !frontier.empty(); ++level) {
        std::vector<int32_t> next;
        for (int32_t s : frontier)
            for (int64_t a = rowptr[s]; a < rowptr[s + 1]; ++a)
                if (d[col[a]] < 0) {
                    d[col[a]] = level;
                    next.push_back(col[a]);
                }
        frontier.swap(next);
    }
    std::vector<uint16_t> out;
    out.resize(size_t(n));
    for (int64_t s = 0; s < n; ++s) out[s] = d[s] < 0 ? uint16_t(0xffff) : d[s] >= 0xffff ? uint16_t(0) : uint16_t(d[s]);
    return out;
}

bool quad_range_ok(int64_t nrows, const std::vector<int64_t> &rowptr, const std::vector<float> &val, int32_t P1) {
    if (nrows * 4 > 65535 || P1 > 65535 || count_quads(nrows, rowptr) + MM_QS_PAD > 65535) return false;
    // the linear path needs 2^w and its products with values in [2^-126, 2^127] to stay normal
    for (float v : val)
        if (!(v > -100.f && v < 20.f)) return false;
    return true;
}

void quad_order(int64_t nrows, const std::vector<int64_t> &rowptr, const std::vector<int32_t> &row2pdf, int32_t P1,
                bool pdf_major, std::vector<int32_t> &order, std::vector<int32_t> &pos) {
    auto nq_of = [&](int64_t r) { return (rowptr[r + 1] - rowptr[r] + 3) / 4; };
    order.resize(nrows);
    for (int64_t r = 0; r < nrows; ++r) order[r] = int32_t(r);
    if (pdf_major) {
        std::vector<double> sum(P1, 0.0), cnt(P1, 0.0);
        for (int64_t r = 0; r < nrows; ++r) {
            sum[row2pdf[r]] += double(nq_of(r));
            cnt[row2pdf[r]] += 1.0;
        }
        std::vector<int32_t> pdf_order(P1), pdf_rank(P1);
        for (int32_t p = 0; p < P1; ++p) pdf_order[p] = p;
        std::stable_sort(pdf_order.begin(), pdf_order.end(), [&](int32_t a, int32_t b) {
            return sum[a] / std::max(cnt[a], 1.0) > sum[b] / std::max(cnt[b], 1.0);
        });
        for (int32_t k = 0; k < P1; ++k) pdf_rank[pdf_order[k]] = k;
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            if (row2pdf[a] != row2pdf[b]) return pdf_rank[row2pdf[a]] < pdf_rank[row2pdf[b]];
            return nq_of(a) > nq_of(b);
        });
    } else {
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return nq_of(a) > nq_of(b); });
    }
    pos.resize(nrows);
    for (int64_t i = 0; i < nrows; ++i) pos[order[i]] = int32_t(i);
}

QuadGraph make_quads(int64_t nrows, const std::vector<int64_t> &rowptr, const std::vector<int32_t> &col,
                     const std::vector<float> &val, const std::vector<int32_t> &row2pdf, int32_t P1, bool pdf_major,
                     int KQ) {
    QuadGraph g;
    g.KQ = KQ;
    const int S1p = int((nrows + 3) / 4 * 4);
    g.ncopy = quad_ncopy(S1p);
    const int pstride = quad_pstride(S1p, g.ncopy);
    auto nq_of = [&](int64_t r) { return (rowptr[r + 1] - rowptr[r] + 3) / 4; };
    // ---- internal numbering
    quad_order(nrows, rowptr, row2pdf, P1, pdf_major, g.order, g.pos);
    if (pdf_major) {
        g.pdfstart.assign(P1 + 1, 0);
        std::vector<int64_t> count(P1 + 1, 0);
        // pdfstart is indexed by pdf id; positions of one pdf are contiguous
        std::vector<int64_t> first(P1, -1), last(P1, -1);
        for (int64_t i = 0; i < nrows; ++i) {
            int32_t p = row2pdf[g.order[i]];
            if (first[p] < 0) first[p] = i;
            last[p] = i;
        }
        // store (start, end) pairs compactly: start[p], and end = start + count; keep two arrays in one:
        // pdfstart[p] = first position, and the count is recovered from pdfcount below
        g.pdfstart.assign(2 * size_t(P1), 0);
        for (int32_t p = 0; p < P1; ++p) {
            g.pdfstart[2 * p] = uint16_t(first[p] < 0 ? 0 : first[p]);
            g.pdfstart[2 * p + 1] = uint16_t(first[p] < 0 ? 0 : last[p] + 1);
        }
    }
    // ---- CSR in internal numbering
    g.rowptr.assign(nrows + 1, 0);
    g.col.resize(col.size());
    g.w.resize(val.size());
    g.recs.resize(nrows);
    g.q0.resize(nrows);
    g.nq.resize(nrows);
    int64_t nq_total = 0, a_out = 0;
    for (int64_t i = 0; i < nrows; ++i) {
        const int64_t r = g.order[i];
        g.rowptr[i] = int32_t(a_out);
        for (int64_t a = rowptr[r]; a < rowptr[r + 1]; ++a, ++a_out) {
            g.col[a_out] = g.pos[col[a]];
            g.w[a_out] = val[a];
        }
        RowRec rec;
        const int64_t nqr = nq_of(r), q0 = nq_total, qe = q0 + nqr - 1;
        const int64_t first = q0 + (KQ - 1 - q0 % KQ);
        const int64_t nextra = nqr && first < qe ? (qe - first + KQ - 1) / KQ : 0;
        rec.qe = uint16_t(nqr ? qe + MM_QS_PAD : 0);
        rec.i1 = uint16_t(nextra >= 1 ? first + MM_QS_PAD : 0);
        rec.pdf = uint16_t(row2pdf[r]);
        rec.i2 = uint16_t(nextra >= 3 ? MM_ROW_LONG : nextra == 2 ? first + KQ + MM_QS_PAD : 0);
        g.q0[i] = int32_t(q0);
        g.nq[i] = int32_t(nqr);
        g.recs[i] = rec;
        nq_total += nqr;
    }
    g.rowptr[nrows] = int32_t(a_out);
    g.quads.assign(size_t(nq_total), Quad{});

    // ---- arc placement.  Lane t of the workgroup owns quads [t*KQ, (t+1)*KQ); the gather
    // instruction (j, k) of a half-wave reads slot k of quad j of its 32 lanes.
    const int64_t lanes = (nq_total + KQ - 1) / KQ;
    // row of every quad
    std::vector<int32_t> qrow(static_cast<size_t>(nq_total), 0);
    for (int64_t i = 0; i < nrows; ++i)
        for (int q = 0; q < g.nq[i]; ++q) qrow[g.q0[i] + q] = int32_t(i);
    // Two candidate placements per half-wave: arcs in CSR order, and a greedy bank-aware one
    // (slot by slot, the remaining arc of the row segment that is cheapest there); the cheaper
    // one under the bank model is kept.
    std::vector<char> used(g.col.size(), 0);
    double cyc_naive = 0, cyc_sched = 0;
    int64_t n_instr = 0;
    auto place = [&](int64_t l0, int64_t l1, bool greedy, std::vector<Quad> &out) -> int64_t {
        std::vector<BankTable> tab(size_t(KQ) * 4);
        out.assign(size_t((l1 - l0) * KQ), Quad{});
        for (int64_t l = l0; l < l1; ++l) {
            int j = 0;
            while (j < KQ && l * KQ + j < nq_total) {
                // segment of consecutive quads of this lane that belong to one row
                const int32_t row = qrow[l * KQ + j];
                int j1 = j;
                while (j1 < KQ && l * KQ + j1 < nq_total && qrow[l * KQ + j1] == row) ++j1;
                const int64_t qrel = l * KQ + j - g.q0[row];
                const int64_t a0 = g.rowptr[row] + 4 * qrel;
                const int64_t a1 = std::min<int64_t>(g.rowptr[row + 1], a0 + 4 * int64_t(j1 - j));
                for (int64_t a = a0; a < a1; ++a) used[a] = 0;
                int64_t next = a0;
                for (int jj = j; jj < j1; ++jj)
                    for (int kk = 0; kk < 4; ++kk) {
                        BankTable &bt = tab[size_t(jj) * 4 + kk];
                        int64_t best = -1;
                        int best_copy = 0;
                        if (!greedy) {
                            if (next < a1) best = next++;
                        } else {
                            int best_cost = 1 << 30;
                            for (int64_t a = a0; a < a1 && best_cost > 0; ++a) {
                                if (used[a]) continue;
                                for (int cp = 0; cp < g.ncopy; ++cp) {
                                    const int c = bt.cost_of(uint16_t(4 * (g.col[a] + cp * pstride)));
                                    if (c < best_cost) {
                                        best_cost = c;
                                        best = a;
                                        best_copy = cp;
                                        if (c == 0) break;
                                    }
                                }
                            }
                        }
                        Quad &Q = out[size_t((l - l0) * KQ + jj)];
                        if (best >= 0) {
                            used[best] = 1;
                            Q.wl[kk] = std::exp2(g.w[best]);
                            Q.off[kk] = uint16_t(4 * (g.col[best] + best_copy * pstride));
                        } else {  // padding: weight 0, an address that costs nothing
                            Q.wl[kk] = 0.f;
                            const int bnk = bt.least_loaded_bank();
                            Q.off[kk] = uint16_t(4 * (bnk < nrows ? bnk : 0));
                        }
                        bt.add(Q.off[kk]);
                    }
                j = j1;
            }
        }
        int64_t cyc = 0;
        for (auto &t : tab) cyc += t.cycles();
        return cyc;
    };
    std::vector<Quad> qa, qb;
    for (int64_t l0 = 0; l0 < lanes; l0 += 32) {  // one half-wave
        const int64_t l1 = std::min(lanes, l0 + 32);
        const int64_t ca = place(l0, l1, false, qa), cb = place(l0, l1, true, qb);
        const std::vector<Quad> &best = cb < ca ? qb : qa;
        for (int64_t l = l0; l < l1; ++l)
            for (int j = 0; j < KQ && l * KQ + j < nq_total; ++j) g.quads[size_t(l * KQ + j)] = best[size_t((l - l0) * KQ + j)];
        // local search on the kept placement: flip the copy of a conflicting slot, or swap it with
        // another slot of the same row segment of its lane, whenever that lowers the modelled cycles
        std::vector<CountedBanks> cb2(size_t(KQ) * 4);
        auto Qat = [&](int64_t l, int j) -> Quad & { return g.quads[size_t(l * KQ + j)]; };
        for (int64_t l = l0; l < l1; ++l)
            for (int j = 0; j < KQ && l * KQ + j < nq_total; ++j)
                for (int k = 0; k < 4; ++k) cb2[size_t(j) * 4 + k].add(Qat(l, j).off[k]);
        for (int pass = 0; pass < 4; ++pass) {
            bool improved = false;
            for (int64_t l = l0; l < l1; ++l)
                for (int j = 0; j < KQ && l * KQ + j < nq_total; ++j) {
                    const int32_t row = qrow[l * KQ + j];
                    for (int k = 0; k < 4; ++k) {
                        CountedBanks &A = cb2[size_t(j) * 4 + k];
                        uint16_t a = Qat(l, j).off[k];
                        if (!A.conflicted(a)) continue;
                        // (1) the other copy of the same source
                        if (g.ncopy > 1 && Qat(l, j).wl[k] != 0.f) {
                            const int cpy = int(a / 4) / pstride;
                            const uint16_t alt = uint16_t(a + (cpy ? -4 * pstride : 4 * pstride));
                            const int before = A.cycles();
                            A.remove(a);
                            A.add(alt);
                            if (A.cycles() < before || (A.cycles() == before && !A.conflicted(alt))) {
                                Qat(l, j).off[k] = alt;
                                improved = true;
                                a = alt;
                                if (!A.conflicted(a)) continue;
                            } else {
                                A.remove(alt);
                                A.add(a);
                            }
                        }
                        // (2) swap with another slot of the same row segment in this lane
                        bool done = false;
                        for (int j2 = 0; j2 < KQ && l * KQ + j2 < nq_total && !done; ++j2) {
                            if (qrow[l * KQ + j2] != row) continue;
                            for (int k2 = 0; k2 < 4 && !done; ++k2) {
                                if (j2 == j && k2 == k) continue;
                                CountedBanks &B = cb2[size_t(j2) * 4 + k2];
                                const uint16_t b = Qat(l, j2).off[k2];
                                const int before = A.cycles() + (&A == &B ? 0 : B.cycles());
                                A.remove(a);
                                B.remove(b);
                                A.add(b);
                                B.add(a);
                                const int after = A.cycles() + (&A == &B ? 0 : B.cycles());
                                if (after < before) {
                                    std::swap(Qat(l, j).off[k], Qat(l, j2).off[k2]);
                                    std::swap(Qat(l, j).wl[k], Qat(l, j2).wl[k2]);
                                    improved = done = true;
                                } else {
                                    A.remove(b);
                                    B.remove(a);
                                    A.add(a);
                                    B.add(b);
                                }
                            }
                        }
                    }
                }
            if (!improved) break;
        }
        int64_t cfinal = 0;
        for (auto &t : cb2) cfinal += t.cycles();
        cyc_naive += double(ca);
        cyc_sched += double(cfinal);
        n_instr += int64_t(KQ) * 4;
    }
    // lane masks: which quads continue the row of their predecessor inside the lane
    for (int64_t l = 0; l < lanes; ++l) {
        uint32_t mask = 0;
        for (int j = 1; j < KQ && l * KQ + j < nq_total; ++j)
            if (qrow[l * KQ + j] == qrow[l * KQ + j - 1]) mask |= 1u << j;
        g.quads[size_t(l * KQ)].mask = mask;
        // the kernel reads the same flag from the sign bit of the quad's first weight (weights are
        // linear, hence non-negative: the bit is free), which costs no register
        for (int j = 1; j < KQ && l * KQ + j < nq_total; ++j)
            if ((mask >> j) & 1u) {
                float &w0 = g.quads[size_t(l * KQ + j)].wl[0];
                w0 = std::copysign(w0, -1.0f);
            }
    }
    g.conflict_before = n_instr ? cyc_naive / double(n_instr) : 0;
    g.conflict_after = n_instr ? cyc_sched / double(n_instr) : 0;
    return g;
}

void eval_packed(const Packed &p, int semiring, const float *in, float *out, int32_t *argmax, int64_t nrows) {
    const float NINF = -std::numeric_limits<float>::infinity();
    for (int64_t r = 0; r < nrows; ++r) {
        out[r] = std::numeric_limits<float>::quiet_NaN();  // every row must be produced by exactly one group
        if (argmax) argmax[r] = -2;
    }
    for (size_t it = 0; it < p.items.size(); ++it) {
        const ItemMeta &m = p.items[it];
        int g = 1 << m.log2g;
        for (int grp = 0; grp < (64 >> m.log2g); ++grp) {
            int lane0 = grp * g;
            int32_t row = p.rowinfo[it * 64 + lane0].row;
            if (row < 0) continue;
            // pass 1: max (and, tropical, the lowest-index argmax)
            float mx = NINF;
            int64_t arg = -1;
            for (int l = lane0; l < lane0 + g; ++l)
                for (int k = 0; k < m.R; ++k) {
                    const Slot &s = p.slots[(size_t(m.slot_row) + k) * 64 + l];
                    float x = s.w + in[s.col];
                    if (x > mx || (x == mx && x > NINF && int64_t(s.col) < arg)) {
                        mx = x;
                        arg = s.col;
                    }
                }
            if (semiring == 1) {
                out[row] = mx;
                if (argmax) argmax[row] = int32_t(arg);
                continue;
            }
            double acc = 0;
            if (mx > NINF)
                for (int l = lane0; l < lane0 + g; ++l)
                    for (int k = 0; k < m.R; ++k) {
                        const Slot &s = p.slots[(size_t(m.slot_row) + k) * 64 + l];
                        acc += std::exp2(double(s.w + in[s.col]) - double(mx));  // packed log weights are in the log2 domain
                    }
            out[row] = mx > NINF ? float(double(mx) + std::log2(acc)) : NINF;
        }
    }
}

}  // namespace mm
