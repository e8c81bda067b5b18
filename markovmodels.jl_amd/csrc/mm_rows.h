// mm_rows.h -- host-side "compile" step for the row kernels (mm_kernel_rows.hip): CSR rows -> the
// register-resident "row-lane" form.
//
// Like mm_pack.h this replaces the reference's per-call container plumbing (CSC<->CSR conversion
// src/linalg.jl:12-49, transpose materialisation :55-67) with a prepare-once layout in the spirit of
// CompiledFSM (src/inference.jl:3-12).  Where the reference's SpMV gives every CSR row a 32-lane warp
// (src/linalg.jl:213-233), here a row belongs to ONE lane (or to an aligned group of g = 2..64 lanes when it
// is long), which keeps its arcs in registers for the whole time loop and owns the row's result:
//
//   * a workgroup has NWC compute waves (+ one service wave that moves emissions, alpha rows and posteriors
//     between HBM and LDS); every lane of a compute wave holds KA arc slots (linear weight 2^w and the LDS
//     byte address of the source state's value);
//   * the KA slots of a wave are cut into SEGMENTS, the same cut for all 64 lanes of the wave: in segment s
//     every lane group of g_s lanes sums the arcs of one row (A_s arcs per lane; rows are sorted by size so
//     that the rows of one segment are about equally long; shorter ones are padded with weight 0).  At the
//     end of a segment the wave "finishes" its 64 / g_s rows: (group sum,) log2, emission, normaliser, 2^x,
//     stores -- one finish per row per frame, no partial sums through LDS and no barrier between the
//     products and the finishing step (the quad kernels need both);
//   * the segments are dealt to the waves longest-processing-time first under a cost model (arcs + a constant
//     per finish), so that the waves reach the single barrier of a frame together;
//   * internal numbering ("position") of the states of one direction = the order in which they are finished
//     (wave, segment, lane group): the stores of a finish go to 64 / g consecutive positions (conflict free
//     in LDS, coalesced in the alpha store);
//   * inside a row the arcs are placed on the (slot, lane) grid so that the 32 lanes of a half-wave that
//     execute the same gather hit distinct LDS banks where possible; the linear vector is kept in two copies
//     rotated by 16 banks, and every arc reads the copy whose bank is free in its instruction.
#pragma once
#include <cstdint>
#include <vector>

namespace mm {

enum { MM_ROW_MAX_SLOTS = 16 };  // segments per wave (4 bits of RowSched::lg each)

struct RowSched {       // one compute wave
    uint64_t endmask;   // bit k: a segment ends after arc pair k (arcs 2k, 2k+1)
    uint64_t lg;        // log2(lanes per row) of the wave's i-th segment in bits [4i, 4i+4)
    uint32_t slot0;     // first row of the wave in the slot table
    uint32_t nslots;    // segments of the wave (>= 1) | (first arc pair of the wave, currently always 0) << 16
};
static_assert(sizeof(RowSched) == 24, "RowSched must be 24 bytes");

struct RowGraph {
    int KA = 0;         // arc slots per lane (even)
    int NWC = 0;        // compute waves
    int rs = 0;         // byte stride between the LDS regions the addresses refer to (copy 1 of the linear vector
                        // lives rs + 64 bytes above copy 0: rotated by 16 banks)
    int nslotrows = 0;  // rows of the slot table (sum of the waves' segments) + 1 padding row
    int trash = 0;      // position written by lanes that finish no row (= number of rows; split forms: beyond all regions)
    int pos_base = 0;   // split forms: position of the form's first row in the team's vector
    int nrows = 0;      // rows this form computes
    int qtrash = 0;     // pdf-major position written by lanes that finish no row (= nrows)
    std::vector<int32_t> order;   // position -> original row
    std::vector<int32_t> pos;     // original row -> position
    std::vector<float> w;         // [KA][64 * NWC] linear weights (2^log2 weight; 0 = padding)
    std::vector<uint32_t> addr;   // [KA][64 * NWC] LDS byte address of the source value, relative to copy 0 of the
                                  // buffer being read
    std::vector<uint32_t> slots;  // [nslotrows][64] x words: word 0 = 4 * position | (4 * pdf) << 16 of the row the
                                  // lane finishes in that segment -- the LAST lane of a row's group; every other lane
                                  // gets position `trash` and the pdf slot that always holds zero(K); backward only,
                                  // word 1 = 4 * (position in the forward numbering) | (4 * position in pdf-major order) << 16
    int slot_words = 1;
    int scale = 4;      // bytes per position in the addresses and slot-table fields (8: pair form)
    int ncopy = 2;      // copies of the linear vector the addresses refer to
    uint32_t copy1 = 0; // address of copy 1 relative to copy 0 (in the units of `addr`)
    bool perm = false;  // copy 1 holds position c at slot c ^ ((c >> 5) & 31) (see RowPackOpts::copy_perm)
    std::vector<RowSched> sched;  // [NWC]
    std::vector<uint16_t> rowpdf; // [rows] pdf of the row at each position
    // CSR in internal numbering with log2-domain weights: the exact fallback walks these
    std::vector<int32_t> rowptr, col;
    std::vector<float> cw;
    std::vector<uint16_t> pdfse;  // backward: [2 * P1] (first, end) of each pdf in pdf-major order
    double conflict_before = 0, conflict_after = 0;  // modelled LDS cycles per gather instruction
    double pad_eff = 0;           // real arcs / arc slots
    float wmin_log2 = 0;          // smallest log2 weight of an arc
    int maxcost = 0, mincost = 0; // cost model: most / least loaded wave
};

struct RowPackOpts {
    int nwc_max = 15;     // compute waves available
    int ka_max = 48;      // register budget (arc slots per lane)
    int rs = 8192;        // LDS region stride in bytes
    int finish_cost = 8;  // cost of one finish in units of one arc (gather + FMA)
    int group_cost = 2;   // extra cost per butterfly level of a grouped (g > 1) segment
    // Relative speed of waves 4k..4k+3 of the workgroup (load / speed is levelled).  The four waves of a SIMD (one of
    // each such group) are arbitrated by priority, then oldest first.  With plain oldest-first arbitration the youngest
    // group was 30 % slower (cycle stamps); since the waves lower their own priority as they advance through a step
    // they progress together: the pair forms use speeds of 1 (mm_engine.hip), the row kernels -- whose service wave
    // is the longest of a step -- still measure best with the old weights (config 3 forced: 4.5 against 4.6 ms).
    float group_speed[4] = {1.0f, 0.92f, 0.80f, 0.70f};
    // Pair form (mm_kernel_pairs.hip: two utterances per workgroup share the graph registers; the linear vector holds
    // their values side by side, 8 bytes per state, fetched by one ds_read_b64): addresses and slot-table fields are
    // 8 * position / 8 * pdf, ONE copy of the vector (the pair kernels are not bound by the LDS), and the slot table
    // has two words in both directions: word 1 = 8 * (position in the OTHER direction's numbering) |
    // (8 * position in pdf-major order) << 16 (set_partner() fills the forward form's once the backward form exists).
    bool pair = false;
    int a_round = 2;      // arc slots per lane of a segment are a multiple of this (2, 4 or 8: the kernels test for the end
                          // of a segment only after every a_round / 2 pairs)
    int copies = 0;
    // Second copy: false = the same order, rotated by half the banks (copy 1 at rs + 64 bytes: two states that share a
    // bank in copy 0 share one in copy 1 as well); true = position c at slot c ^ ((c >> 5) & 31) of copy 1 (at rs): the
    // low five bits -- the bank -- are scrambled with the next five, so states that collide in copy 0 mostly do not in
    // copy 1 (two independent choices per arc)
    bool copy_perm = false;       // copies of the linear vector the addresses may use (0: 2 for the row form, 1 for the pair form)
    // register windows the kernels are instantiated for: KA is rounded up to one of them (0-terminated; empty: any even KA)
    int ka_choices[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // Split forms (make_rows_split): this form computes only the rows of `subset` (original ids), while its arcs read a
    // linear vector that holds ALL rows -- the rows of the other workgroups of the team included -- at the positions
    // `gpos` (original row -> position).  The form's own rows occupy positions pos_base .. pos_base + |subset| - 1 of that
    // vector, in finishing order; lanes that finish no row write position `gtrash`.
    const std::vector<int32_t> *subset = nullptr;
    const std::vector<int32_t> *gpos = nullptr;
    int pos_base = 0;
    int gtrash = -1;
    bool plan_only = false;  // stop after the numbering (order / pos): the first pass of make_rows_split
    // Wave form (mm_kernel_wave.hip: ONE wave computes a whole direction of an utterance in the log domain): at most
    // `acap_force` arcs per lane and row (longer rows get lane groups), every segment owns `seg_stride` arc slots of the
    // lane (segment i at slots seg_stride * i ..: statically unrolled code), the weights are kept as log2 values
    // (padding: -inf) and word 1 of the slot table (other direction's position | pdf-major position) exists in both
    // directions.
    int acap_force = 0;
    int seg_stride = 0;
    bool log_weights = false;
    bool want_partner = false;
    bool spread_pdf = false;  // rows of one pdf go to different segments where possible (fewer LDS add conflicts)
    // Viterbi form (mm_kernel_vit.hip): the arcs of a lane stay in the order of the row (ascending source state, every g-th
    // arc of a row split over g lanes): the first maximum a lane meets is then the one with the lowest source state, the
    // tie rule of the back-pointers, and "arc number in the row" is what a back-pointer stores.  No bank-aware placement.
    bool keep_order = false;
    // Mixed layout (Viterbi form): a wave has mix_n4 positions of 4 arc slots followed by mix_n2 positions of 2; segments of
    // up to 2 arcs per lane take the narrow positions first.  RowSched::nslots = segments | (those in wide positions) << 16;
    // the wave's segments are ordered wide positions first.  mix_n4 < 0: off.
    int mix_n4 = -1, mix_n2 = 0;
    // Effort of the bank-aware placement: 2 = greedy + local search (the forms of the LDS-bound kernels), 1 = greedy only,
    // 0 = the arcs stay in CSR order in copy 0 (forms of kernels that are bound by their latency chains, not by LDS cycles:
    // the wave form -- the placement is most of the host time of packing a small graph).
    int place = 2;
    // Pair form of a whole graph (no subset): 1 = the BANKS of the rows are chosen too -- rows of a segment trade their positions
    // (the segment keeps its block of positions) until no bank holds more sources of a half-wave segment than the segment has arc
    // slots, as far as that goes; the arcs of a half-wave segment then go to their slots by an exact edge colouring of the
    // (lane, bank) multigraph (Koenig: as many colours as the largest degree) instead of the greedy pass + local search.
    int bank_opt = 0;
    // Pair forms: the rows of a segment are dealt to its two half-waves so that the emission factors a finish reads (8 bytes at
    // 8 * pdf) meet in as few bank pairs as possible (plan_for)
    bool pdf_halves = false;
    bool naive_stats = true;  // also model the arcs in CSR order (RowGraph::conflict_before: informational)
    bool q_positions = true;  // pdf-major positions of the rows (the kernels that sum the posteriors per pdf over contiguous ranges);
                              // false: the slot table's q field stays 0 (the wave kernel has its own pdf tables)
};

// rowptr/col/val_log2: 0-based CSR of M (out[r] = (+)_k val[k] (*) in[col[k]]), square, nrows rows.
// fwd_pos (backward form only, else empty): position of every original row in the forward numbering.
// Returns false if the graph does not fit (KA > ka_max, too many segments per wave, positions beyond the region).
bool make_rows(int64_t nrows, const std::vector<int64_t> &rowptr, const std::vector<int32_t> &col,
               const std::vector<float> &val_log2, const std::vector<int32_t> &row2pdf, int32_t P1, bool backward,
               const std::vector<int32_t> &fwd_pos, const RowPackOpts &opt, RowGraph &out);

// Split forms: the rows of an FSM cut into H sets, the same sets in both directions, one form per (direction, set).
// A team of H workgroups computes one direction of an utterance pair: workgroup h finishes the rows of set h and
// receives the others' rows once per frame (mm_kernel_pairs.hip, template parameter H).  For graphs whose arcs do not
// fit the registers of one compute unit (the reference's WSJ denominator graph, misc/benchmark/den_fsm_wsj.txt: 52 k
// arcs against ~42 k arc slots), and for the reference's unlimited matrix sizes (src/linalg.jl:170-181) in general.
struct SplitInfo {
    int H = 0;
    int total = 0;                  // positions of the team's vector (regions + alignment padding)
    int base[8] = {0};              // first position of set h (even), the same in both directions
    int count[8] = {0};             // rows of set h
    std::vector<int32_t> part;      // original row -> set
    std::vector<int32_t> gpos[2];   // per direction: original row -> position in the team's vector
};
// fwd_* / bwd_*: CSR of the forward (T_hat') and the backward (T_hat) product.  out[d * H + h]: direction d, set h.
// The slot tables' word 1 refers to the OTHER direction's numbering relative to the set's base (what workgroup h of the
// other direction stored for these rows is a contiguous range).
bool make_rows_split(int H, int64_t nrows, const std::vector<int64_t> &fwd_ptr, const std::vector<int32_t> &fwd_col,
                     const std::vector<float> &fwd_val, const std::vector<int64_t> &bwd_ptr, const std::vector<int32_t> &bwd_col,
                     const std::vector<float> &bwd_val, const std::vector<int32_t> &row2pdf, int32_t P1, const RowPackOpts &opt_f,
                     const RowPackOpts &opt_b, std::vector<RowGraph> &out, SplitInfo &info);

// Pair forms: write the other direction's numbering into word 1 of the slot table (partner_pos: original row ->
// position in the other direction).
void set_partner(RowGraph &g, const std::vector<int32_t> &partner_pos);

// Host evaluation of one product through the row form exactly as a workgroup walks it (lane by lane, segment
// by segment, group sums), in the linear domain: in_lin[position] -> out_lin[position].  Test aid.
void eval_rows(const RowGraph &g, const float *in_lin, float *out_lin);

// The same for the wave form (log2 weights, segments of seg_stride slots, <= 4 arcs per lane and segment, lane-group
// log-sum-exp): in_log2[position] -> out_log2[position].
void eval_rows_log(const RowGraph &g, int seg_stride, const float *in_log2, float *out_log2);

}  // namespace mm
