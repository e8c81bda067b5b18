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

QuadGraph make_quads(int64_t nrows, const std::vector<int64_t> &rowptr, const std::vector<int32_t> &col,
                     const std::vector<float> &val) {
    QuadGraph g;
    g.rowptr.resize(nrows + 1);
    g.col.assign(col.begin(), col.end());
    g.w.assign(val.begin(), val.end());
    bool ok = nrows * 4 <= 65535;
    for (int64_t r = 0; r <= nrows; ++r) g.rowptr[r] = int32_t(rowptr[r]);
    for (int64_t r = 0; r < nrows; ++r) {
        for (int64_t b = rowptr[r]; b < rowptr[r + 1]; b += 4) {
            Quad q;
            std::memset(&q, 0, sizeof(q));
            q.rowoff = uint16_t(4 * r);
            for (int k = 0; k < 4; ++k) {
                int64_t a = b + k;
                if (a < rowptr[r + 1]) {
                    // the linear path needs 2^w and its products with values in [2^-126, 2^127] to stay normal
                    if (!(val[a] > -100.f && val[a] < 20.f)) ok = false;
                    q.wl[k] = std::exp2(val[a]);
                    q.off[k] = uint16_t(4 * col[a]);
                }
            }
            g.quads.push_back(q);
        }
    }
    if (g.quads.size() > 65535) ok = false;
    g.qstart.assign(nrows + 1, 0);
    std::vector<int32_t> order(nrows);
    int64_t acc = 0;
    for (int64_t r = 0; r < nrows; ++r) {
        g.qstart[r] = uint16_t(acc);
        acc += (rowptr[r + 1] - rowptr[r] + 3) / 4;
        order[r] = int32_t(r);
    }
    g.qstart[nrows] = uint16_t(acc);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        return (rowptr[a + 1] - rowptr[a] + 3) / 4 > (rowptr[b + 1] - rowptr[b] + 3) / 4;
    });
    g.rord.resize(nrows);
    for (int64_t r = 0; r < nrows; ++r) g.rord[r] = uint16_t(order[r]);
    g.fast_ok = ok;
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
