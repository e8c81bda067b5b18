// mm_rows.cpp -- see mm_rows.h
#include "mm_rows.h"
#include <cstdio>
#include <atomic>
#include <chrono>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>

namespace mm {
namespace {

struct Unit {   // one row as the kernel sees it
    int32_t row;
    int g;      // lanes per row (power of two)
    int A;      // arc slots per lane (even, >= 2)
    int deg;
};
struct Segment {  // 64 / g units of one g class
    int g, A;
    std::vector<int32_t> rows;  // <= 64 / g
    int wave = -1;
};

int log2i(int g) {
    int l = 0;
    while ((1 << l) < g) ++l;
    return l;
}

// distinct addresses per bank of one gather instruction of a half-wave, with multiplicities.  Allocation free (a pack of a
// small graph builds thousands of these: with a vector per bank the allocator was most of the time of packing a numerator
// graph): at most 64 addresses live at a time (32 lanes, and the moves of the local search remove before they add).
struct Banks {
    struct E {
        uint32_t addr;
        uint16_t n, bank;
    };
    E ent[64];
    int nent = 0, sqsum = 0;
    uint8_t cnt[32] = {0};  // distinct addresses per bank
    static int of(uint32_t a) { return int(a >> 2) & 31; }
    int find(uint32_t a) const {
        for (int i = 0; i < nent; ++i)
            if (ent[i].addr == a) return i;
        return -1;
    }
    void add(uint32_t a) {
        const int i = find(a);
        if (i >= 0) {
            ++ent[i].n;
            return;
        }
        if (nent >= 64) std::abort();  // (cannot happen: see above)
        const int bk = of(a);
        ent[nent++] = E{a, 1, uint16_t(bk)};
        sqsum += 2 * cnt[bk] + 1;
        ++cnt[bk];
    }
    void remove(uint32_t a) {
        const int i = find(a);
        if (i < 0) return;
        if (--ent[i].n == 0) {
            const int bk = ent[i].bank;
            --cnt[bk];
            sqsum -= 2 * cnt[bk] + 1;
            ent[i] = ent[--nent];
        }
    }
    int cost_of(uint32_t a) const {  // extra cycles this address would add (0: free bank or a broadcast)
        return find(a) >= 0 ? 0 : int(cnt[of(a)]);
    }
    int cycles() const {
        int m = 1;
        for (int c : cnt) m = std::max(m, c);
        return m;
    }
    // smooth objective of the local search: every resolved conflict lowers it, also where the maximum (cycles())
    // is held by several banks at once and no single move lowers that
    int sq() const { return sqsum; }
    bool conflicted(uint32_t a) const { return cnt[of(a)] > 1; }
    int least_loaded() const {
        int b = 0;
        for (int i = 1; i < 32; ++i)
            if (cnt[i] < cnt[b]) b = i;
        return b;
    }
};

struct Plan {
    std::vector<Segment> segs;
    std::vector<std::vector<int>> wave_segs;  // per wave: indices into segs, in execution order
    int KA = 0, maxcost = 0, mincost = 0;
    bool ok = false;
    std::vector<int> n4;  // mixed layout: segments of a wave that sit in 4-slot positions (the first ones)
};

Plan plan_for(const std::vector<int32_t> &rows, const std::vector<int64_t> &rowptr, int acap, const RowPackOpts &opt,
              const std::vector<int32_t> *row2pdf = nullptr) {
    Plan p;
    std::vector<Unit> units;
    units.reserve(rows.size());
    for (int32_t r : rows) {
        const int d = int(rowptr[r + 1] - rowptr[r]);
        int g = 1;
        while ((std::max(d, 1) + g - 1) / g > acap / opt.a_round * opt.a_round && g < 64) g *= 2;
        int A = (std::max(d, 1) + g - 1) / g;
        A = (A + opt.a_round - 1) / opt.a_round * opt.a_round;
        units.push_back(Unit{int32_t(r), g, A, d});
    }
    for (int g = 1; g <= 64; g *= 2) {
        std::vector<Unit> us;
        for (auto &u : units)
            if (u.g == g) us.push_back(u);
        if (us.empty()) continue;
        std::stable_sort(us.begin(), us.end(), [](const Unit &a, const Unit &b) { return a.A > b.A; });
        const size_t per = size_t(64 / g);
        if (opt.spread_pdf && row2pdf) {
            // rows of one pdf go to DIFFERENT segments where the class has several of the same length: the wave kernel adds
            // the posteriors of a segment's rows to their pdfs with one LDS float add, and lanes that hit the same pdf are
            // served one after the other
            for (size_t i0 = 0; i0 < us.size();) {
                size_t i1 = i0;
                while (i1 < us.size() && us[i1].A == us[i0].A) ++i1;
                const size_t n = i1 - i0, nseg = (n + per - 1) / per;
                if (nseg > 1) {
                    std::vector<Unit> run(us.begin() + i0, us.begin() + i1), out(n);
                    std::stable_sort(run.begin(), run.end(), [&](const Unit &a, const Unit &b) { return (*row2pdf)[a.row] < (*row2pdf)[b.row]; });
                    // deal the pdf-sorted rows round robin over the segments of the run (the last segment may be shorter)
                    std::vector<size_t> fill(nseg, 0), cap(nseg, per);
                    cap[nseg - 1] = n - per * (nseg - 1);
                    size_t sgm = 0;
                    for (size_t j = 0; j < n; ++j) {
                        while (fill[sgm] >= cap[sgm]) sgm = (sgm + 1) % nseg;
                        out[sgm * per + fill[sgm]++] = run[j];
                        sgm = (sgm + 1) % nseg;
                    }
                    std::copy(out.begin(), out.end(), us.begin() + i0);
                }
                i0 = i1;
            }
        }
        for (size_t i = 0; i < us.size(); i += per) {
            Segment s;
            s.g = g;
            s.A = us[i].A;
            for (size_t j = i; j < std::min(us.size(), i + per); ++j) s.rows.push_back(us[j].row);
            p.segs.push_back(std::move(s));
        }
    }
    // Pair forms: a finish reads the emission factor of its row, 8 bytes at 8 * pdf -- bank pair pdf mod 32, served per half-wave; two
    // rows of one half with different pdfs in one bank pair cost an LDS cycle more.  The rows of a segment are interchangeable
    // (same lane group, same slots): the pdfs of a bank pair are dealt to the two halves in turn (rows of ONE pdf stay together:
    // a broadcast), as far as the halves' capacities go.
    if (opt.pdf_halves && row2pdf) {
        long long before = 0, after = 0;
        auto half_cost = [&](const Segment &sg) {
            long long c = 0;
            const int per_half = 32 / std::min(sg.g, 32);
            for (int h = 0; h < 2 && sg.g <= 32; ++h) {
                int distinct[32] = {0};
                std::vector<int32_t> seen;
                for (int j = h * per_half; j < std::min<int>((h + 1) * per_half, int(sg.rows.size())); ++j) {
                    const int32_t pd = (*row2pdf)[size_t(sg.rows[size_t(j)])];
                    if (std::find(seen.begin(), seen.end(), pd) == seen.end()) {
                        seen.push_back(pd);
                        ++distinct[pd & 31];
                    }
                }
                int m = 1;
                for (int d : distinct) m = std::max(m, d);
                c += m - 1;
            }
            return c;
        };
        for (auto &sg : p.segs) {
            if (sg.g > 16 || sg.rows.size() < 3) continue;
            before += half_cost(sg);
            const int per_half = 32 / sg.g, n = int(sg.rows.size());
            const int cap0 = std::min(per_half, n), cap1 = n - cap0;
            if (cap1 <= 0) continue;
            // groups of rows by pdf, the pdfs ordered by bank pair
            std::vector<int32_t> rows = sg.rows;
            std::stable_sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) {
                const int32_t pa = (*row2pdf)[size_t(a)], pb = (*row2pdf)[size_t(b)];
                return (pa & 31) != (pb & 31) ? (pa & 31) < (pb & 31) : pa < pb;
            });
            std::vector<int32_t> half[2];
            int nb[2][32] = {{0}, {0}};
            for (size_t i0 = 0; i0 < rows.size();) {
                size_t i1 = i0;
                const int32_t pd = (*row2pdf)[size_t(rows[i0])];
                while (i1 < rows.size() && (*row2pdf)[size_t(rows[i1])] == pd) ++i1;
                const int bk = pd & 31, free0 = cap0 - int(half[0].size()), free1 = cap1 - int(half[1].size());
                int h = nb[0][bk] < nb[1][bk] ? 0 : (nb[1][bk] < nb[0][bk] ? 1 : (free0 >= free1 ? 0 : 1));
                if ((h ? free1 : free0) <= 0) h = 1 - h;
                for (size_t i = i0; i < i1; ++i) {
                    if (int(half[h].size()) >= (h ? cap1 : cap0)) h = 1 - h;  // (the group is cut: both halves read the pdf)
                    if (half[h].empty() || (*row2pdf)[size_t(half[h].back())] != pd) ++nb[h][bk];
                    half[h].push_back(rows[i]);
                }
                i0 = i1;
            }
            std::vector<int32_t> out = half[0];
            out.insert(out.end(), half[1].begin(), half[1].end());
            Segment trial = sg;
            trial.rows = out;
            if (half_cost(trial) <= half_cost(sg)) sg.rows = std::move(out);
            after += half_cost(sg);
        }
        if (getenv("MM_VERBOSE_PLAN")) fprintf(stderr, "[mm] pdf_halves: extra LDS cycles of the emission reads %lld -> %lld (%zu segments)\n", before, after, p.segs.size());
    }
    const int nwc = std::max(1, std::min<int>(opt.nwc_max, int(p.segs.size())));
    auto cost = [&](const Segment &s) { return s.A + opt.finish_cost + opt.group_cost * log2i(s.g); };
    std::vector<int> idx(p.segs.size());
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cost(p.segs[a]) > cost(p.segs[b]); });
    std::vector<int> load(nwc, 0), arcs(nwc, 0), used4(nwc, 0), used2(nwc, 0);
    p.wave_segs.assign(nwc, {});
    p.n4.assign(nwc, 0);
    auto level = [&](int w, int extra) { return float(load[w] + extra) / opt.group_speed[std::min(w >> 2, 3)]; };
    for (int i : idx) {
        int best = -1;
        const int c = cost(p.segs[i]);
        for (int w = 0; w < nwc; ++w) {
            if (int(p.wave_segs[w].size()) >= MM_ROW_MAX_SLOTS) continue;
            if (opt.mix_n4 >= 0) {  // N4 positions of 4 slots, N2 of 2 per wave: a segment of <= 2 arcs takes a narrow one first
                const bool narrow = p.segs[i].A <= 2 && used2[w] < opt.mix_n2;
                if (!narrow && used4[w] >= opt.mix_n4) continue;
            } else if (arcs[w] + std::max(p.segs[i].A, opt.seg_stride) > opt.ka_max) {
                continue;  // (a segment owns seg_stride slots)
            }
            if (best < 0 || level(w, c) < level(best, c) || (level(w, c) == level(best, c) && arcs[w] < arcs[best])) best = w;
        }
        if (best < 0) return p;  // more than MM_ROW_MAX_SLOTS segments per wave
        p.segs[i].wave = best;
        p.wave_segs[best].push_back(i);
        load[best] += cost(p.segs[i]);
        arcs[best] += std::max(p.segs[i].A, opt.seg_stride);
        if (opt.mix_n4 >= 0) {
            if (p.segs[i].A <= 2 && used2[best] < opt.mix_n2) ++used2[best];
            else ++used4[best];
        }
    }
    for (auto &ws : p.wave_segs)
        std::stable_sort(ws.begin(), ws.end(), [&](int a, int b) { return p.segs[a].A > p.segs[b].A; });
    if (opt.mix_n4 >= 0)
        for (int w = 0; w < nwc; ++w) p.n4[w] = used4[w];  // (sorted by A: the wide positions' occupants come first)
    p.KA = *std::max_element(arcs.begin(), arcs.end());
    p.maxcost = 0;
    p.mincost = 1 << 30;
    for (int w = 0; w < nwc; ++w) {  // (levelled cost: what the frame time follows)
        p.maxcost = std::max(p.maxcost, int(level(w, 0) + 0.5f));
        p.mincost = std::min(p.mincost, int(level(w, 0) + 0.5f));
    }
    p.ok = p.KA <= opt.ka_max && p.KA <= 128;
    return p;
}

}  // namespace

}  // namespace mm
extern "C" {
std::atomic<long long> mm_rows_prof_ns[8];
}
namespace mm {
#define PROF_LAP(i)                                                                                                              \
    do {                                                                                                                         \
        const auto now_ = std::chrono::steady_clock::now();                                                                      \
        mm_rows_prof_ns[i] += std::chrono::duration_cast<std::chrono::nanoseconds>(now_ - prof_t_).count();                          \
        prof_t_ = now_;                                                                                                          \
    } while (0)

// ---- bank_opt (RowPackOpts): the banks of the rows of a whole-graph pair form.
// A half-wave gather of segment h (32 lanes, A_h arc slots each) is free of bank conflicts iff the lanes' arcs can be dealt to the
// slots with no bank twice in a slot; by Koenig's edge-colouring theorem that is possible iff no bank holds more than A_h of the
// half-segment's sources (the lanes hold at most A_h arcs by construction).  The bank of a row is its position mod 32, and the
// position is free inside the row's segment (a finish writes wherever the slot table says): rows of one segment trade positions
// while that lowers the excess sum_h sum_b max(0, D[h][b] - A_h) (+ a small term that levels D, + a small price on finishes whose
// 16-lane store groups hit a bank pair twice).
static void optimise_banks(const Plan &plan, const std::vector<int64_t> &rowptr, const std::vector<int32_t> &col, std::vector<int32_t> &pos,
                           std::vector<int32_t> &order) {
    const int nseg = int(plan.segs.size());
    const int NH = 2 * nseg;
    std::vector<int> cap(size_t(NH), 0);
    std::vector<std::vector<std::pair<int, int>>> mem(pos.size());  // row (as a source) -> (half-segment, arcs that read it)
    std::vector<int> D(size_t(NH) * 32, 0);
    {
        std::vector<int> cnt(pos.size(), 0);
        std::vector<int32_t> touched;
        for (int si = 0; si < nseg; ++si) {
            const Segment &s = plan.segs[size_t(si)];
            for (int half = 0; half < 2; ++half) {
                const int h = 2 * si + half;
                cap[size_t(h)] = s.A;
                touched.clear();
                for (int l = 0; l < 32; ++l) {
                    const int lane = half * 32 + l, grp = lane / s.g, sub = lane % s.g;
                    if (grp >= int(s.rows.size())) continue;
                    const int32_t r = s.rows[size_t(grp)];
                    int n = 0;
                    for (int64_t a = rowptr[r] + sub; a < rowptr[r + 1] && n < s.A; a += s.g, ++n) {
                        if (cnt[size_t(col[a])]++ == 0) touched.push_back(col[a]);
                    }
                }
                for (int32_t c : touched) {
                    mem[size_t(c)].push_back({h, cnt[size_t(c)]});
                    D[size_t(h) * 32 + size_t(pos[size_t(c)] & 31)] += cnt[size_t(c)];
                    cnt[size_t(c)] = 0;
                }
            }
        }
    }
    // finishing lane of a row inside its segment, and the store groups' (position mod 16) counts
    std::vector<int> seg_of(pos.size(), -1), lane_of(pos.size(), 0);
    std::vector<int> W(size_t(nseg) * 4 * 16, 0);
    for (int si = 0; si < nseg; ++si) {
        const Segment &s = plan.segs[size_t(si)];
        for (size_t j = 0; j < s.rows.size(); ++j) {
            const int32_t r = s.rows[j];
            seg_of[size_t(r)] = si;
            lane_of[size_t(r)] = int(j) * s.g + s.g - 1;
            ++W[(size_t(si) * 4 + size_t(lane_of[size_t(r)] / 16)) * 16 + size_t(pos[size_t(r)] & 15)];
        }
    }
    constexpr long long WX = 4096, WW = 256;
    auto phi = [&](int h, int d) -> long long {
        const int x = d - cap[size_t(h)];
        return (x > 0 ? WX * x : 0) + (long long)d * d;
    };
    auto wcost = [&](int c) -> long long { return c > 1 ? WW * (c - 1) : 0; };
    long long excess = 0;
    for (int h = 0; h < NH; ++h)
        for (int b = 0; b < 32; ++b) excess += std::max(0, D[size_t(h) * 32 + b] - cap[size_t(h)]);
    if (getenv("MM_VERBOSE_PLAN")) fprintf(stderr, "[mm] bank_opt: excess before %lld\n", excess);
    std::vector<int> dd(size_t(NH), 0);
    std::vector<int> th;
    // all rows in a fixed order; a pass tries every pair (first the pairs of one segment -- the cheap moves --, then, while an
    // excess remains, every pair of rows)
    std::vector<int32_t> all;
    for (int si = 0; si < nseg; ++si)
        for (int32_t r : plan.segs[size_t(si)].rows) all.push_back(r);
    const int nall = int(all.size());
    for (int pass = 0; pass < 40 && excess > 0; ++pass) {
        bool improved = false;
        const bool wide = pass >= 6 && getenv("MM_BANKOPT_WIDE") != nullptr;  // (rows that leave their segment's block scatter the finishes' global stores)
        if (pass >= 6 && !wide) break;
        {
            for (int i = 0; i < nall; ++i)
                for (int j = i + 1; j < nall; ++j) {
                    const int32_t r = all[size_t(i)], t = all[size_t(j)];
                    if (!wide && seg_of[size_t(r)] != seg_of[size_t(t)]) break;  // (rows of a segment are adjacent in `all`)
                    const int b = pos[size_t(r)] & 31, b2 = pos[size_t(t)] & 31;
                    if (b == b2) continue;
                    if (wide) {  // only rows that sit in an overloaded (half-segment, bank) are worth the full search
                        bool hot = false;
                        for (auto &m : mem[size_t(r)]) hot = hot || D[size_t(m.first) * 32 + b] > cap[size_t(m.first)];
                        if (!hot) break;
                    }
                    // net change of bank b in every touched half-segment: + the arcs that read t, - those that read r
                    th.clear();
                    for (auto &m : mem[size_t(r)]) {
                        if (dd[size_t(m.first)] == 0) th.push_back(m.first);
                        dd[size_t(m.first)] -= m.second;
                    }
                    for (auto &m : mem[size_t(t)]) {
                        if (dd[size_t(m.first)] == 0) th.push_back(m.first);
                        dd[size_t(m.first)] += m.second;
                    }
                    long long delta = 0;
                    for (int h : th) {
                        const int d = dd[size_t(h)];
                        if (d == 0) continue;
                        const int o1 = D[size_t(h) * 32 + b], o2 = D[size_t(h) * 32 + b2];
                        delta += phi(h, o1 + d) - phi(h, o1) + phi(h, o2 - d) - phi(h, o2);
                    }
                    const int g1 = lane_of[size_t(r)] / 16, g2 = lane_of[size_t(t)] / 16;
                    const int c1 = pos[size_t(r)] & 15, c2 = pos[size_t(t)] & 15;
                    const int sr = seg_of[size_t(r)], st = seg_of[size_t(t)];
                    if ((g1 != g2 || sr != st) && c1 != c2) {
                        int *w1 = &W[(size_t(sr) * 4 + size_t(g1)) * 16], *w2 = &W[(size_t(st) * 4 + size_t(g2)) * 16];
                        delta += wcost(w1[c1] - 1) - wcost(w1[c1]) + wcost(w1[c2] + 1) - wcost(w1[c2]);
                        delta += wcost(w2[c2] - 1) - wcost(w2[c2]) + wcost(w2[c1] + 1) - wcost(w2[c1]);
                    }
                    if (delta < 0) {
                        for (int h : th) {
                            const int d = dd[size_t(h)];
                            if (d == 0) continue;
                            int &o1 = D[size_t(h) * 32 + b], &o2 = D[size_t(h) * 32 + b2];
                            excess -= std::max(0, o1 - cap[size_t(h)]) + std::max(0, o2 - cap[size_t(h)]);
                            o1 += d;
                            o2 -= d;
                            excess += std::max(0, o1 - cap[size_t(h)]) + std::max(0, o2 - cap[size_t(h)]);
                        }
                        if ((g1 != g2 || sr != st) && c1 != c2) {
                            int *w1 = &W[(size_t(sr) * 4 + size_t(g1)) * 16], *w2 = &W[(size_t(st) * 4 + size_t(g2)) * 16];
                            --w1[c1], ++w1[c2], --w2[c2], ++w2[c1];
                        }
                        std::swap(pos[size_t(r)], pos[size_t(t)]);
                        improved = true;
                    }
                    for (int h : th) dd[size_t(h)] = 0;
                }
        }
        if (!improved) break;
    }
    if (getenv("MM_VERBOSE_PLAN")) {
        long long tot = 0, slots = 0, mx = 0;
        for (int h = 0; h < NH; ++h) {
            int m = 0;
            for (int b = 0; b < 32; ++b) {
                tot += D[size_t(h) * 32 + b];
                m = std::max(m, D[size_t(h) * 32 + b] - cap[size_t(h)]);
            }
            mx += std::max(0, m);
            slots += 32ll * cap[size_t(h)];
        }
        fprintf(stderr, "[mm] bank_opt: %d half-segments, %lld arcs in %lld slots, excess %lld (sum over banks), %lld (sum of the half-segments' worst bank)\n", NH, tot, slots, excess, mx);
    }
    for (size_t r = 0; r < pos.size(); ++r)
        if (pos[r] >= 0) order[size_t(pos[r])] = int32_t(r);
}

// Exact dealing of a half-wave segment's arcs to its A slots: an edge colouring of the bipartite multigraph (lane, bank) with A
// colours.  A bank that holds more than A of the arcs is split into virtual banks of at most A (the arcs beyond A meet another arc
// of their bank in some slot: the conflicts that cannot be avoided).  lane_arcs[l] = {bank, id} of the lane's arcs (at most A);
// slot_of[l][i] receives the slot of the lane's i-th arc.
static void colour_half_segment(int A, const std::vector<std::pair<int, int>> (&lane_arcs)[32], std::vector<int> (&slot_of)[32]) {
    struct Edge {
        int l, v, colour;
    };
    std::vector<Edge> ed;
    int bank_n[32] = {0};
    std::vector<std::pair<int, int>> where;  // edge -> (lane, index in the lane)
    for (int l = 0; l < 32; ++l) {
        slot_of[l].assign(lane_arcs[l].size(), -1);
        for (size_t i = 0; i < lane_arcs[l].size(); ++i) {
            const int b = lane_arcs[l][i].first;
            ed.push_back(Edge{l, b + 32 * (bank_n[b]++ / A), -1});
            where.push_back({l, int(i)});
        }
    }
    int nv = 32;
    for (auto &e : ed) nv = std::max(nv, e.v + 1);
    std::vector<int> lc(size_t(32) * A, -1), vc(size_t(nv) * A, -1);
    for (int e = 0; e < int(ed.size()); ++e) {
        const int l = ed[size_t(e)].l, v = ed[size_t(e)].v;
        int a = 0, b = 0;
        while (lc[size_t(l) * A + a] >= 0) ++a;
        while (vc[size_t(v) * A + b] >= 0) ++b;
        if (vc[size_t(v) * A + a] >= 0) {
            // the a / b alternating path from v: flip it (it cannot reach lane l: it enters lanes by edges of colour a, and l has none)
            std::vector<int> path;
            int cur = vc[size_t(v) * A + a];
            bool want_b = true;  // the next edge leaves a lane, by colour b
            while (cur >= 0) {
                path.push_back(cur);
                cur = want_b ? lc[size_t(ed[size_t(cur)].l) * A + b] : vc[size_t(ed[size_t(cur)].v) * A + a];
                want_b = !want_b;
            }
            for (int pe : path) {
                const Edge &x = ed[size_t(pe)];
                lc[size_t(x.l) * A + x.colour] = -1;
                vc[size_t(x.v) * A + x.colour] = -1;
            }
            for (int pe : path) {
                Edge &x = ed[size_t(pe)];
                x.colour = x.colour == a ? b : a;
                lc[size_t(x.l) * A + x.colour] = pe;
                vc[size_t(x.v) * A + x.colour] = pe;
            }
        }
        ed[size_t(e)].colour = a;
        lc[size_t(l) * A + a] = e;
        vc[size_t(v) * A + a] = e;
    }
    for (int e = 0; e < int(ed.size()); ++e) slot_of[where[size_t(e)].first][size_t(where[size_t(e)].second)] = ed[size_t(e)].colour;
}

bool make_rows(int64_t nrows, const std::vector<int64_t> &rowptr, const std::vector<int32_t> &col,
               const std::vector<float> &val, const std::vector<int32_t> &row2pdf, int32_t P1, bool backward,
               const std::vector<int32_t> &fwd_pos, const RowPackOpts &opt, RowGraph &g) {
    auto prof_t_ = std::chrono::steady_clock::now();
    g = RowGraph();
    // the rows this form computes (all, or a subset: split forms) and the positions of the vector its arcs read
    std::vector<int32_t> myrows;
    if (opt.subset) {
        myrows = *opt.subset;
    } else {
        myrows.resize(size_t(nrows));
        std::iota(myrows.begin(), myrows.end(), 0);
    }
    const int64_t nsub = int64_t(myrows.size());
    const int64_t ntot = opt.gtrash >= 0 ? opt.gtrash : nrows;  // positions of the vector (the trash position follows them)
    if (nrows < 1 || nsub < 1 || (ntot + 1) * (opt.pair ? 8 : 4) > (opt.pair ? 2 : 1) * opt.rs || (ntot + 1) * (opt.pair ? 8 : 4) > 65528 || P1 > 8000)
        return false;
    if (opt.copy_perm && ((ntot + 1 + 31) & ~int64_t(31)) * 4 > opt.rs) return false;  // (copy 1 scrambles inside blocks of 32)
    // ---- schedule: the cap on arcs per lane of one row decides how many rows are split over lane groups; take
    // the cap whose most loaded wave is cheapest
    Plan best;
    for (int acap : {4, 12, 16, 24, 32, 48, 64}) {
        if (opt.acap_force ? acap != opt.acap_force : (acap == 4 || acap > opt.ka_max)) continue;
        Plan p = plan_for(myrows, rowptr, acap, opt, &row2pdf);
        if (!p.ok && getenv("MM_VERBOSE_PLAN"))
            fprintf(stderr, "[mm] plan with at most %d arcs per lane and row: %zu segments on %zu waves, KA %d (limit %d)\n", acap, p.segs.size(),
                    p.wave_segs.size(), p.KA, opt.ka_max);
        if (!p.ok) continue;
        if (!best.ok || p.maxcost < best.maxcost || (p.maxcost == best.maxcost && p.KA < best.KA)) best = std::move(p);
    }
    if (!best.ok) return false;
    PROF_LAP(0);
    const int NWC = int(best.wave_segs.size());
    int KA = std::max(2, (best.KA + 1) & ~1);
    if (opt.mix_n4 >= 0) {
        KA = 4 * opt.mix_n4 + 2 * opt.mix_n2;
        for (auto &sg : best.segs)
            if (sg.A > 4) return false;
        if (opt.keep_order)
            for (int32_t r : myrows)
                if (rowptr[r + 1] - rowptr[r] > 255) return false;
    } else if (opt.seg_stride) {
        size_t most = 0;
        for (auto &ws : best.wave_segs) most = std::max(most, ws.size());
        KA = opt.seg_stride * int(most);
        if (KA > opt.ka_max) return false;
        for (auto &sg : best.segs)
            if (sg.A > opt.seg_stride) return false;  // (a row of more than 64 x seg_stride arcs)
        if (opt.keep_order)  // (a back-pointer is the number of the arc in its row, one byte, 255 = none)
            for (int32_t r : myrows)
                if (rowptr[r + 1] - rowptr[r] > 255) return false;
    }
    for (int c : opt.ka_choices)
        if (c >= KA) {
            KA = c;
            break;
        }
    const int NT = 64 * NWC;
    g.KA = KA;
    g.NWC = NWC;
    g.rs = opt.rs;
    g.trash = int(ntot);
    g.pos_base = opt.pos_base;
    g.nrows = int(nsub);
    g.qtrash = int(nsub);
    g.slot_words = (backward || opt.pair || opt.want_partner) ? 2 : 1;
    g.scale = opt.pair ? 8 : 4;
    g.ncopy = opt.copies ? opt.copies : (opt.pair ? 1 : 2);
    const uint32_t SC = uint32_t(g.scale);
    const bool want_q = (backward || opt.pair || opt.want_partner) && opt.q_positions;  // pdf-major positions
    g.maxcost = best.maxcost;
    g.mincost = best.mincost;
    // ---- numbering: the order in which the rows are finished
    g.order.assign(size_t(nsub), -1);
    g.pos.assign(size_t(nrows), -1);
    {
        int32_t next = 0;
        for (int w = 0; w < NWC; ++w)
            for (int si : best.wave_segs[w])
                for (int32_t r : best.segs[si].rows) {
                    g.pos[r] = next;
                    g.order[next] = r;
                    ++next;
                }
        if (next != nsub) return false;
    }
    const bool bank_opt = opt.bank_opt && opt.pair && !opt.subset && !opt.gpos && !opt.keep_order && opt.place >= 2 &&
                          (opt.copies ? opt.copies : 1) == 1 && !opt.log_weights && opt.mix_n4 < 0 && !opt.seg_stride;
    if (bank_opt) optimise_banks(best, rowptr, col, g.pos, g.order);
    if (opt.plan_only) return true;
    // position of a source state in the vector the arcs read
    auto gp = [&](int32_t r) { return opt.gpos ? (*opt.gpos)[size_t(r)] : g.pos[size_t(r)]; };
    g.rowpdf.resize(size_t(nsub));
    for (int64_t i = 0; i < nsub; ++i) g.rowpdf[i] = uint16_t(row2pdf[g.order[i]]);
    // pdf-major order of the rows (backward: the posterior of a pdf is a sum over contiguous entries)
    std::vector<int32_t> qpos(size_t(nrows), 0);
    if (want_q) {
        std::vector<int32_t> byp(myrows);
        std::sort(byp.begin(), byp.end());
        std::stable_sort(byp.begin(), byp.end(), [&](int32_t a, int32_t b) { return row2pdf[a] < row2pdf[b]; });
        g.pdfse.assign(2 * size_t(P1), 0);
        std::vector<int64_t> first(P1, -1), last(P1, -1);
        for (int64_t i = 0; i < nsub; ++i) {
            qpos[byp[i]] = int32_t(i);
            const int32_t p = row2pdf[byp[i]];
            if (first[p] < 0) first[p] = i;
            last[p] = i;
        }
        for (int32_t p = 0; p < P1; ++p) {
            g.pdfse[2 * p] = uint16_t(first[p] < 0 ? 0 : first[p]);
            g.pdfse[2 * p + 1] = uint16_t(first[p] < 0 ? 0 : last[p] + 1);
        }
        // Inside a pdf's range the order is free: a finish stores the posteriors of its 64 / g rows to their q
        // positions in one LDS store (lane groups of 32 for 4-byte entries, of 16 for the pair form's 8-byte entries;
        // bank = position mod 32), so the rows that share such a group get positions in distinct banks where their
        // ranges allow it (measured on config 3 with conflict-free stores: -3 % in the phase that combines).
        std::vector<char> taken(size_t(nsub), 0);
        std::fill(qpos.begin(), qpos.end(), -1);
        const int G = opt.pair ? 16 : 32;
        const int trash_bank = int(nsub % 32);  // (the lanes that finish nothing store to the position behind the last)
        for (int w = 0; w < NWC; ++w)
            for (int si : best.wave_segs[w]) {
                const Segment &sg = best.segs[si];
                for (int l0 = 0; l0 < 64; l0 += G) {
                    unsigned used = (sg.g > 1 || int(sg.rows.size()) * sg.g < 64) ? 1u << trash_bank : 0u;
                    for (int l = l0; l < l0 + G; ++l) {
                        if (l % sg.g != sg.g - 1 || l / sg.g >= int(sg.rows.size())) continue;
                        const int32_t r = sg.rows[l / sg.g], pd = row2pdf[r];
                        int64_t pick = -1;
                        for (int64_t q = first[pd]; q <= last[pd]; ++q)
                            if (!taken[q]) {
                                if (pick < 0) pick = q;
                                if (!((used >> (q % 32)) & 1u)) {
                                    pick = q;
                                    break;
                                }
                            }
                        taken[pick] = 1;
                        used |= 1u << (pick % 32);
                        qpos[r] = int32_t(pick);
                    }
                }
            }
    }
    PROF_LAP(1);
    // ---- CSR in internal numbering (exact fallback)
    g.rowptr.assign(size_t(nsub) + 1, 0);
    g.col.clear();
    g.cw.clear();
    {
        int64_t a_out = 0;
        for (int64_t i = 0; i < nsub; ++i) {
            const int64_t r = g.order[i];
            g.rowptr[i] = int32_t(a_out);
            for (int64_t a = rowptr[r]; a < rowptr[r + 1]; ++a, ++a_out) {
                g.col.push_back(gp(col[a]));
                g.cw.push_back(val[a]);
            }
        }
        g.rowptr[nsub] = int32_t(a_out);
    }
    PROF_LAP(2);
    // ---- schedules, slot table
    const int zero_pdf = (P1 + 3) & ~3;  // emission slot that always holds zero(K)
    g.sched.assign(NWC, RowSched{0, 0, 0, 0});
    int nslots = 0;
    for (int w = 0; w < NWC; ++w) nslots += int(best.wave_segs[w].size());
    g.nslotrows = nslots + 2;  // + two padding rows: the prefetch after a wave's last finishes reads up to two rows ahead
    g.slots.assign(size_t(g.nslotrows) * 64 * g.slot_words, 0);
    g.w.assign(size_t(KA) * NT, opt.log_weights ? -std::numeric_limits<float>::infinity() : 0.f);
    g.addr.assign(size_t(KA) * NT, 0u);
    // model addresses (4 bytes per position): copy cp of position c, and back
    const uint32_t copy1 = uint32_t(opt.copy_perm ? opt.rs : opt.rs + 64);
    const uint32_t ncopy = uint32_t(g.ncopy);
    const bool perm = opt.copy_perm;
    g.copy1 = (opt.pair ? 2u : 1u) * copy1;
    g.perm = perm;
    auto enc = [&](uint32_t c, uint32_t cp) { return cp ? copy1 + 4u * (perm ? (c ^ ((c >> 5) & 31u)) : c) : 4u * c; };
    auto other = [&](uint32_t a) {
        if (a >= copy1) {
            const uint32_t s = (a - copy1) / 4u;
            return 4u * (perm ? (s ^ ((s >> 5) & 31u)) : s);
        }
        return enc(a / 4u, 1);
    };
    // (bank model: a 4-byte read occupies bank (a / 4) % 32; an 8-byte read of the pair form the bank pair
    // (a / 8) % 32 -- the same structure, so the pair addresses are modelled as a / 2)
    double cyc_naive = 0, cyc_sched = 0;
    int64_t n_instr = 0, real_arcs = 0;
    int slotrow = 0;
    for (int w = 0; w < NWC; ++w) {
        RowSched &sc = g.sched[w];
        sc.slot0 = uint32_t(slotrow);
        int arcs_w = 0;
        for (int si : best.wave_segs[w]) arcs_w += best.segs[si].A;
        (void)arcs_w;
        int k0 = 0, sidx = 0;  // left-aligned: the wave leaves the pair sequence after its last segment
        sc.nslots = uint32_t(best.wave_segs[w].size()) | (uint32_t(k0 / 2) << 16);
        if (opt.mix_n4 >= 0) sc.nslots = uint32_t(best.wave_segs[w].size()) | (uint32_t(best.n4[w]) << 16);
        for (int si : best.wave_segs[w]) {
            const Segment &s = best.segs[si];
            const int lg = log2i(s.g);
            if (opt.seg_stride) k0 = opt.seg_stride * sidx;
            if (opt.mix_n4 >= 0) k0 = sidx < best.n4[w] ? 4 * sidx : 4 * opt.mix_n4 + 2 * (sidx - best.n4[w]);
            sc.lg |= uint64_t(lg) << (4 * sidx);
            sc.endmask |= uint64_t(1) << ((k0 + s.A) / 2 - 1);
            // slot table row
            for (int l = 0; l < 64; ++l) {
                const size_t e = (size_t(slotrow) * 64 + l) * g.slot_words;
                const int grp = l / s.g;
                // the LAST lane of a row's group holds the group sum and finishes the row; the others finish nothing
                if (grp < int(s.rows.size()) && l % s.g == s.g - 1) {
                    const int32_t r = s.rows[grp];
                    g.slots[e] = uint32_t(SC * (g.pos_base + g.pos[r])) | (uint32_t(SC * row2pdf[r]) << 16);
                    if (g.slot_words == 2)
                        g.slots[e + 1] = uint32_t((opt.pair ? 8 : 4) * (fwd_pos.empty() ? 0 : fwd_pos[r])) | (uint32_t(SC * qpos[r]) << 16);
                } else {
                    g.slots[e] = uint32_t(SC * g.trash) | (uint32_t(SC * zero_pdf) << 16);
                    if (g.slot_words == 2) g.slots[e + 1] = 0u | (uint32_t(SC * g.qtrash) << 16);
                }
            }
            // arcs of the segment: every lane of a row's group takes every g-th arc; inside its A slots the
            // lane's arcs are ordered (and their copy chosen) by the bank model, one half-wave at a time
            for (int half = 0; half < 2; ++half) {
                // (scratch of the thread, reused: a pack of a small graph runs this block dozens of times, and its allocations
                // were half the time of packing a numerator graph)
                static thread_local std::vector<Banks> tabn_s, tab_s;
                static thread_local std::vector<int64_t> la_s;
                static thread_local std::vector<uint32_t> ad_s;
                static thread_local std::vector<float> wt_s;
                static thread_local std::vector<char> used_s;
                if (tab_s.size() < size_t(s.A)) tab_s.resize(size_t(s.A));
                if (opt.naive_stats && tabn_s.size() < size_t(s.A)) tabn_s.resize(size_t(s.A));
                for (int k = 0; k < s.A; ++k) {
                    tab_s[size_t(k)] = Banks();
                    if (opt.naive_stats) tabn_s[size_t(k)] = Banks();
                }
                Banks *const tab = tab_s.data(), *const tabn = tabn_s.data();
                // the lane's arcs (indices into the internal CSR): at most A each
                struct LaneArcs {
                    const int64_t *p;
                    size_t n;
                    size_t size() const { return n; }
                    int64_t operator[](size_t i) const { return p[i]; }
                };
                struct {
                    LaneArcs arcs;
                } la[32];
                la_s.resize(size_t(32) * size_t(s.A));
                for (int l = 0; l < 32; ++l) {
                    int64_t *dst = la_s.data() + size_t(l) * size_t(s.A);
                    size_t cnt = 0;
                    const int lane = half * 32 + l, grp = lane / s.g, sub = lane % s.g;
                    if (grp < int(s.rows.size())) {
                        const int64_t i = g.pos[s.rows[grp]];
                        for (int64_t a = g.rowptr[i] + sub; a < g.rowptr[i + 1] && cnt < size_t(s.A); a += s.g) dst[cnt++] = a;
                    }
                    la[l].arcs = LaneArcs{dst, cnt};
                }
                // naive placement (CSR order, copy 0) for the statistics
                for (int l = 0; l < 32 && opt.naive_stats; ++l)
                    for (int k = 0; k < s.A; ++k)
                        tabn[k].add(k < int(la[l].arcs.size()) ? uint32_t(4 * g.col[la[l].arcs[k]]) : uint32_t(4 * (l % int(ntot))));  // (model units)
                // greedy: lane after lane, slot after slot, the remaining arc / copy that is cheapest there
                ad_s.assign(size_t(32) * s.A, 0u);
                wt_s.assign(size_t(32) * s.A, 0.f);
                auto ad = [&, A = s.A](int l) { return ad_s.data() + size_t(l) * A; };
                auto wt = [&, A = s.A](int l) { return wt_s.data() + size_t(l) * A; };
                if (bank_opt) {  // exact: an edge colouring of (lane, bank) with A colours; the padding slots last
                    std::vector<std::pair<int, int>> lane_arcs[32];
                    std::vector<int> slot_of[32];
                    for (int l = 0; l < 32; ++l)
                        for (size_t i = 0; i < la[l].arcs.size(); ++i)
                            lane_arcs[l].push_back({Banks::of(enc(uint32_t(g.col[la[l].arcs[i]]), 0)), int(i)});
                    colour_half_segment(s.A, lane_arcs, slot_of);
                    std::vector<char> filled(size_t(32) * s.A, 0);
                    for (int l = 0; l < 32; ++l)
                        for (size_t i = 0; i < la[l].arcs.size(); ++i) {
                            const int k = slot_of[l][i];
                            ad(l)[k] = enc(uint32_t(g.col[la[l].arcs[i]]), 0);
                            wt(l)[k] = std::exp2(g.cw[la[l].arcs[i]]);
                            filled[size_t(l) * s.A + k] = 1;
                            tab[k].add(ad(l)[k]);
                            ++real_arcs;
                        }
                    for (int l = 0; l < 32; ++l)
                        for (int k = 0; k < s.A; ++k)
                            if (!filled[size_t(l) * s.A + k]) {  // padding: weight 0, an address that costs nothing
                                const int bnk = tab[k].least_loaded();
                                ad(l)[k] = uint32_t(4 * (bnk < ntot ? bnk : 0));
                                wt(l)[k] = 0.f;
                                tab[k].add(ad(l)[k]);
                            }
                }
                for (int l = 0; l < 32 && !bank_opt; ++l) {
                    used_s.assign(la[l].arcs.size(), 0);
                    char *const used = used_s.data();
                    for (int k = 0; k < s.A; ++k) {
                        int bi = -1, bcost = 1 << 30;
                        uint32_t baddr = 0;
                        for (size_t i = 0; i < la[l].arcs.size() && bcost > 0; ++i) {
                            if (used[i]) continue;
                            if ((opt.keep_order || opt.place == 0) && i != size_t(k)) continue;  // (slot k holds the lane's k-th arc)
                            for (uint32_t cp = 0; cp < (opt.place == 0 ? 1u : ncopy); ++cp) {
                                const uint32_t a = enc(uint32_t(g.col[la[l].arcs[i]]), cp);
                                const int c = tab[k].cost_of(a);
                                if (c < bcost) {
                                    bcost = c;
                                    bi = int(i);
                                    baddr = a;
                                    if (c == 0) break;
                                }
                            }
                        }
                        if (bi >= 0) {
                            used[bi] = 1;
                            ad(l)[k] = baddr;
                            wt(l)[k] = opt.log_weights ? g.cw[la[l].arcs[bi]] : std::exp2(g.cw[la[l].arcs[bi]]);
                            ++real_arcs;
                        } else {  // padding: weight 0, an address that costs nothing
                            const int bnk = tab[k].least_loaded();
                            ad(l)[k] = uint32_t(4 * (bnk < ntot ? bnk : 0));
                            wt(l)[k] = opt.log_weights ? -std::numeric_limits<float>::infinity() : 0.f;
                        }
                        tab[k].add(ad(l)[k]);
                    }
                }
                auto real = [&](float w) { return opt.log_weights ? w > -std::numeric_limits<float>::infinity() : w != 0.f; };
                // local search on the sum of squared bank loads: flip the copy of a conflicting slot, or swap it with another
                // slot of the lane (in either copy)
                for (int pass = 0; pass < (opt.keep_order || opt.place < 2 || bank_opt ? 0 : 12); ++pass) {
                    bool improved = false;
                    for (int l = 0; l < 32; ++l)
                        for (int k = 0; k < s.A; ++k) {
                            uint32_t a = ad(l)[k];
                            if (!tab[k].conflicted(a)) continue;
                            if (real(wt(l)[k]) && ncopy > 1) {
                                const uint32_t alt = other(a);
                                const int before = tab[k].sq();
                                tab[k].remove(a);
                                tab[k].add(alt);
                                if (tab[k].sq() < before) {
                                    ad(l)[k] = a = alt;
                                    improved = true;
                                    if (!tab[k].conflicted(a)) continue;
                                } else {
                                    tab[k].remove(alt);
                                    tab[k].add(a);
                                }
                            }
                            // (the lanes of one row are interchangeable: an arc may trade places with any arc of its row in
                            // this half-wave)
                            const int gl = std::min(s.g, 32), l_lo = l / gl * gl;
                            bool moved = false;
                            for (int l2 = l_lo; l2 < l_lo + gl && !moved; ++l2)
                            for (int k2 = 0; k2 < s.A; ++k2) {
                                if (k2 == k) continue;
                                const uint32_t b = ad(l2)[k2];
                                const int before = tab[k].sq() + tab[k2].sq();
                                tab[k].remove(a);
                                tab[k2].remove(b);
                                // the best of the copies of each arc in its new slot
                                uint32_t na = a, nb = b;
                                int best = 1 << 30;
                                for (uint32_t ca = 0; ca < (real(wt(l)[k]) ? ncopy : 1u); ++ca)
                                    for (uint32_t cb = 0; cb < (real(wt(l2)[k2]) ? ncopy : 1u); ++cb) {
                                        const uint32_t xa = ca ? other(a) : a, xb = cb ? other(b) : b;
                                        tab[k].add(xb);
                                        tab[k2].add(xa);
                                        const int q = tab[k].sq() + tab[k2].sq();
                                        tab[k].remove(xb);
                                        tab[k2].remove(xa);
                                        if (q < best) {
                                            best = q;
                                            na = xa;
                                            nb = xb;
                                        }
                                    }
                                if (best < before) {
                                    tab[k].add(nb);
                                    tab[k2].add(na);
                                    ad(l)[k] = nb;
                                    ad(l2)[k2] = na;
                                    std::swap(wt(l)[k], wt(l2)[k2]);
                                    improved = moved = true;
                                    break;
                                }
                                tab[k].add(a);
                                tab[k2].add(b);
                            }
                        }
                    if (!improved) break;
                }
                for (int k = 0; k < s.A; ++k) {
                    cyc_naive += opt.naive_stats ? tabn[k].cycles() : 1;
                    cyc_sched += tab[k].cycles();
                    ++n_instr;
                    for (int l = 0; l < 32; ++l) {
                        const size_t e = size_t(k0 + k) * NT + size_t(w) * 64 + half * 32 + l;
                        g.w[e] = wt(l)[k];
                        g.addr[e] = opt.pair ? 2 * ad(l)[k] : ad(l)[k];  // (placement worked in 4-byte model units)
                    }
                }
            }
            k0 += s.A;
            ++sidx;
            ++slotrow;
        }
    }
    PROF_LAP(3);
    // the padding rows of the slot table
    for (int pr = 0; pr < 2; ++pr)
        for (int l = 0; l < 64; ++l) {
            const size_t e = (size_t(slotrow + pr) * 64 + l) * g.slot_words;
            g.slots[e] = uint32_t(SC * g.trash) | (uint32_t(SC * zero_pdf) << 16);
            if (g.slot_words == 2) g.slots[e + 1] = 0u | (uint32_t(SC * g.qtrash) << 16);
        }
    g.conflict_before = n_instr ? cyc_naive / double(n_instr) : 0;
    g.conflict_after = n_instr ? cyc_sched / double(n_instr) : 0;
    g.pad_eff = n_instr ? double(real_arcs) / (32.0 * double(n_instr)) : 0;
    g.wmin_log2 = 0.f;
    for (float v : val)
        if (v > -std::numeric_limits<float>::infinity()) g.wmin_log2 = std::min(g.wmin_log2, v);
    return true;
}

bool make_rows_split(int H, int64_t nrows, const std::vector<int64_t> &fwd_ptr, const std::vector<int32_t> &fwd_col,
                     const std::vector<float> &fwd_val, const std::vector<int64_t> &bwd_ptr, const std::vector<int32_t> &bwd_col,
                     const std::vector<float> &bwd_val, const std::vector<int32_t> &row2pdf, int32_t P1, const RowPackOpts &opt_f,
                     const RowPackOpts &opt_b, std::vector<RowGraph> &out, SplitInfo &info) {
    if (H < 2 || H > 8 || nrows < 2 * H) return false;
    info = SplitInfo();
    info.H = H;
    // ---- the sets: the same in both directions (what workgroup h of the other direction stored for a frame is then exactly
    // what workgroup h of this direction combines with).  Longest-processing-time first on the pair (forward cost, backward
    // cost) of every state, cost = arcs + a finish; a state goes to the set whose larger relative load stays smallest.
    const std::vector<int64_t> *ptr[2] = {&fwd_ptr, &bwd_ptr};
    const RowPackOpts *opts[2] = {&opt_f, &opt_b};
    std::vector<int32_t> idx(static_cast<size_t>(nrows));
    std::iota(idx.begin(), idx.end(), 0);
    auto cost = [&](int d, int32_t r) { return double((*ptr[d])[r + 1] - (*ptr[d])[r]) + double(opts[d]->finish_cost); };
    double tot[2] = {0, 0};
    for (int32_t r : idx)
        for (int d = 0; d < 2; ++d) tot[d] += cost(d, r);
    std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) {
        return cost(0, a) / tot[0] + cost(1, a) / tot[1] > cost(0, b) / tot[0] + cost(1, b) / tot[1];
    });
    std::vector<double> load[2] = {std::vector<double>(H, 0.0), std::vector<double>(H, 0.0)};
    std::vector<int> cnt(H, 0);
    // (the sets' regions are sized alike: keep the counts close -- within an eighth for teams of up to 4, within a 64th for the
    // larger teams, whose workgroups have no LDS to spare for rows they do not have)
    const int cap = int((nrows + H - 1) / H) + int(nrows / ((H > 4 ? 64 : 8) * H)) + 1;
    info.part.assign(size_t(nrows), 0);
    for (int32_t r : idx) {
        int best = -1;
        double bl = 0;
        for (int h = 0; h < H; ++h) {
            if (cnt[h] >= cap) continue;
            const double l = std::max((load[0][h] + cost(0, r)) / tot[0], (load[1][h] + cost(1, r)) / tot[1]);
            if (best < 0 || l < bl || (l == bl && cnt[h] < cnt[best])) {
                best = h;
                bl = l;
            }
        }
        if (best < 0) return false;
        info.part[size_t(r)] = best;
        for (int d = 0; d < 2; ++d) load[d][best] += cost(d, r);
        ++cnt[best];
    }
    std::vector<std::vector<int32_t>> sets(H);
    for (int64_t r = 0; r < nrows; ++r) sets[info.part[size_t(r)]].push_back(int32_t(r));
    int next = 0;
    for (int h = 0; h < H; ++h) {
        if (sets[h].empty()) return false;
        info.base[h] = next;
        info.count[h] = int(sets[h].size());
        next = (next + info.count[h] + 1) & ~1;  // (regions start at even positions: 16-byte rows of pairs)
    }
    info.total = next;
    // ---- pass 1: the numbering of every (direction, set); pass 2: the forms, their arcs reading the team's vector
    out.assign(size_t(2 * H), RowGraph());
    const std::vector<int32_t> none;
    std::vector<int32_t> lpos[2];  // original row -> position inside its set's region
    for (int d = 0; d < 2; ++d) {
        info.gpos[d].assign(size_t(nrows), -1);
        lpos[d].assign(size_t(nrows), -1);
        for (int h = 0; h < H; ++h) {
            RowPackOpts o = *opts[d];
            o.subset = &sets[h];
            o.gtrash = info.total;
            o.plan_only = true;
            RowGraph tmp;
            if (!make_rows(nrows, *ptr[d], d ? bwd_col : fwd_col, d ? bwd_val : fwd_val, row2pdf, P1, d == 1, none, o, tmp)) {
                if (getenv("MM_VERBOSE")) fprintf(stderr, "[mm] make_rows_split(%d): plan of direction %d, set %d (%zu rows) failed\n", H, d, h, sets[h].size());
                return false;
            }
            for (int32_t r : sets[h]) {
                lpos[d][size_t(r)] = tmp.pos[size_t(r)];
                info.gpos[d][size_t(r)] = info.base[h] + tmp.pos[size_t(r)];
            }
        }
    }
    for (int d = 0; d < 2; ++d)
        for (int h = 0; h < H; ++h) {
            RowPackOpts o = *opts[d];
            o.subset = &sets[h];
            o.gpos = &info.gpos[d];
            o.pos_base = info.base[h];
            o.gtrash = info.total;
            if (!make_rows(nrows, *ptr[d], d ? bwd_col : fwd_col, d ? bwd_val : fwd_val, row2pdf, P1, d == 1, lpos[1 - d], o,
                           out[size_t(d * H + h)])) {
                if (getenv("MM_VERBOSE")) fprintf(stderr, "[mm] make_rows_split(%d): form of direction %d, set %d failed\n", H, d, h);
                return false;
            }
        }
    return true;
}

void set_partner(RowGraph &g, const std::vector<int32_t> &partner_pos) {
    if (g.slot_words != 2) return;
    const uint32_t SC = uint32_t(g.scale);
    for (size_t e = 0; e < g.slots.size(); e += 2) {
        const uint32_t p = (g.slots[e] & 0xffffu) / SC;
        if (int(p) == g.trash) continue;
        g.slots[e + 1] = (g.slots[e + 1] & 0xffff0000u) | uint32_t(SC * partner_pos[g.order[p - uint32_t(g.pos_base)]]);
    }
}

void eval_rows_log(const RowGraph &g, int seg_stride, const float *in_log2, float *out_log2) {
    const int NT = 64 * g.NWC;
    for (int w = 0; w < g.NWC; ++w) {
        const RowSched &sc = g.sched[w];
        const int nseg = int(sc.nslots & 0xffffu);
        for (int i = 0; i < nseg; ++i) {
            const int A = ((sc.endmask >> (seg_stride * i / 2 + 1)) & 1) ? 4 : 2;
            const int lg = int((sc.lg >> (4 * i)) & 15), gsz = 1 << lg;
            double m[64], sum[64];
            for (int l = 0; l < 64; ++l) {  // the lane's (max, scaled sum) over its <= 4 arcs
                double x[4], mx = -std::numeric_limits<double>::infinity();
                for (int k = 0; k < A; ++k) {
                    const size_t e = size_t(seg_stride * i + k) * NT + size_t(w) * 64 + l;
                    x[k] = double(g.w[e]) + double(in_log2[g.addr[e] / uint32_t(g.scale)]);
                    mx = std::max(mx, x[k]);
                }
                m[l] = mx;
                sum[l] = 0;
                for (int k = 0; k < A; ++k) sum[l] += mx > -std::numeric_limits<double>::infinity() ? std::exp2(x[k] - mx) : 0.0;
            }
            for (int l0 = 0; l0 < 64; l0 += gsz) {
                double M = -std::numeric_limits<double>::infinity(), S = 0;
                for (int l = l0; l < l0 + gsz; ++l) M = std::max(M, m[l]);
                for (int l = l0; l < l0 + gsz; ++l) S += M > -std::numeric_limits<double>::infinity() ? sum[l] * std::exp2(m[l] - M) : 0.0;
                const uint32_t info = g.slots[(size_t(sc.slot0 + i) * 64 + l0 + gsz - 1) * g.slot_words];
                const int p = int(info & 0xffffu) / g.scale;
                if (p != g.trash) out_log2[p] = float(M + std::log2(S));
            }
        }
    }
}

void eval_rows(const RowGraph &g, const float *in_lin, float *out_lin) {
    const int NT = 64 * g.NWC;
    for (int w = 0; w < g.NWC; ++w) {
        const RowSched &sc = g.sched[w];
        float acc[64];
        for (int l = 0; l < 64; ++l) acc[l] = 0.f;
        int slot = 0;
        for (int k2 = int(sc.nslots >> 16); k2 < g.KA / 2 && slot < int(sc.nslots & 0xffffu); ++k2) {
            for (int k = 2 * k2; k < 2 * k2 + 2; ++k)
                for (int l = 0; l < 64; ++l) {
                    const size_t e = size_t(k) * NT + size_t(w) * 64 + l;
                    uint32_t a = g.addr[e], c = a / uint32_t(g.scale);
                    if (g.ncopy > 1 && a >= g.copy1) {
                        c = (a - g.copy1) / uint32_t(g.scale);
                        if (g.perm) c ^= (c >> 5) & 31u;
                    }
                    acc[l] = std::fmaf(g.w[e], in_lin[c], acc[l]);
                }
            if (!((sc.endmask >> k2) & 1)) continue;
            const int lg = int((sc.lg >> (4 * slot)) & 15), gsz = 1 << lg;
            for (int l0 = 0; l0 < 64; l0 += gsz) {
                float s = 0.f;
                for (int l = l0; l < l0 + gsz; ++l) s += acc[l];
                const uint32_t info = g.slots[(size_t(sc.slot0 + slot) * 64 + l0 + gsz - 1) * g.slot_words];
                const int p = int(info & 0xffffu) / g.scale;
                if (p != g.trash) out_lin[p] = s;
            }
            for (int l = 0; l < 64; ++l) acc[l] = 0.f;
            ++slot;
        }
    }
}

}  // namespace mm
