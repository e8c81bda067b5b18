// mm_pack.h -- host-side "compile" step: CSR rows -> the packed wave-item form
// the HIP kernels consume.
//
// Replaces, for the new engine, the reference's per-call container plumbing
// (CSC<->CSR conversion src/linalg.jl:12-49, transpose materialisation :55-67)
// with a prepare-once layout, in the spirit of CompiledFSM (src/inference.jl:3-12).
//
// Layout ("items"): the reference's GPU SpMV gives every CSR row a whole warp
// (src/linalg.jl:213-233).  LF-MMI graphs have in-degree median 2 / max ~80, so
// here each row gets a lane *group* of g = 1,2,4,...,64 lanes of a 64-wide
// wavefront, sized so that a lane holds at most 4 arcs (R = ceil(nnz/g) <= 4;
// rows beyond 256 arcs keep g = 64 and a longer R).  An item is one
// wavefront's worth of such groups of one (g, R) class:
//   slots   [slot_row + k][lane] = {col, weight}   k < R   (8 B per lane: one
//           coalesced 512 B load per k), padded with {0, zero(K)}
//   rowinfo [item][lane]         = {row or -1, pdf of that row}
// Items are sorted by estimated cost so that wave w of a workgroup taking
// items w, w + NW, ... is balanced.
#pragma once
#include <cstdint>
#include <vector>

namespace mm {

struct ItemMeta {
    uint32_t slot_row;  // first row (of 64 lanes) in the slot array
    uint16_t R;         // arc slots per lane
    uint16_t log2g;     // lanes per row group = 1 << log2g
};

struct Slot {
    uint32_t col;
    float w;
};

struct RowInfo {
    int32_t row;  // -1: padding lane group
    int32_t pdf;
};

struct Packed {
    std::vector<ItemMeta> items;
    std::vector<RowInfo> rowinfo;  // items.size() * 64
    std::vector<Slot> slots;       // n_slot_rows * 64
    int64_t n_slot_rows = 0;
    int64_t nnz = 0;
};

// rowptr/col/val: 0-based CSR of the matrix M of out[r] = (+)_k val[k] (*) in[col[k]].
// zero_w: the semiring zero used for padding slots (-inf).
Packed pack_rows(int64_t nrows, const std::vector<int64_t> &rowptr, const std::vector<int32_t> &col,
                 const std::vector<float> &val, const std::vector<int32_t> &row2pdf, float zero_w);

// ---------------------------------------------------------------------------
// "Quad" form for the fast forward-backward kernel (mm_kernel_quad.hip).
//
// Internal numbering: the rows (states) of one direction are renumbered by
// decreasing number of quads (forward), or grouped by pdf and then by decreasing
// size (backward, so that the posterior of a pdf is a sum over CONTIGUOUS
// positions), so that the per-row loops of a wavefront have similar lengths and
// every per-row LDS access of the finishing phase is contiguous across lanes.
// Every row is cut into quads of 4 arcs (the last one padded with weight 0); quad
// q belongs to lane q / KQ of the workgroup, which keeps its KQ quads in
// registers for the whole time loop.  Weights are LINEAR (2^w), columns are LDS
// byte offsets (4 * internal position of the source).  Inside a row the arcs are
// placed on the (quad, slot) grid so that the 32 lanes of a half-wave that
// execute the same gather instruction hit distinct LDS banks where possible
// (identical addresses broadcast for free).
// ---------------------------------------------------------------------------
struct Quad {
    float wl[4];      // 2^(log2 weight); 0 = padding.  Sign bit of wl[0] set <=> the quad continues the row of
                      // its predecessor in the lane (its sum is added to the lane's running sum)
    uint16_t off[4];  // 4 * internal position of the source state
    uint32_t mask;    // in the FIRST quad of a lane (q % KQ == 0): bit j set <=> quad q + j continues the
                      // row of quad q + j - 1 (the same flags as the sign bits, for host-side checks)
    uint32_t pad;
};
static_assert(sizeof(Quad) == 32, "Quad must be 32 bytes");

// RowRec: one per internal position: what the thread that finishes the row needs, as indices into the array of
// per-lane running quad sums.  That array starts with two slots that always hold 0 (index = quad + 2):
// absent terms point at slot 0, so the common case is three unconditional loads and two adds.
struct RowRec {
    uint16_t qe;   // 2 + last quad of the row (it holds the running sum of the row's quads in its lane);
                   // 0: the row has no arcs (its value is zero(K))
    uint16_t i1;   // 2 + first lane-end quad (q % KQ == KQ - 1) of the row before its last quad, or 0
    uint16_t pdf;
    uint16_t i2;   // 2 + second lane end, or 0; 1 (MM_ROW_LONG): more than two -- walk i1, i1 + KQ, ... < qe
};
enum { MM_ROW_LONG = 1, MM_QS_PAD = 2 };
static_assert(sizeof(RowRec) == 8, "RowRec must be 8 bytes");

// The linear vector p is kept in `ncopy` LDS copies, copy c at float offset c * quad_pstride(): the
// copies are rotated against each other by 32 / ncopy banks, so every arc can be read from ncopy
// different banks and the placement picks the one that is free in its gather instruction.
#ifdef __HIPCC__
#define MM_HD __host__ __device__
#else
#define MM_HD
#endif
MM_HD inline int quad_pstride(int S1p, int ncopy) { return (S1p + 31) / 32 * 32 + 32 / ncopy; }
inline int quad_ncopy(int64_t S1p) { return 4 * (int64_t(quad_pstride(int(S1p), 2)) + S1p) <= 65535 ? 2 : 1; }

struct QuadGeometry {
    int KQ;  // quads per lane held in registers
    int NW;  // wavefronts per workgroup
};
QuadGeometry pick_quad_geometry(int64_t nquads);

struct QuadGraph {
    std::vector<int32_t> order;  // internal position -> original row
    std::vector<int32_t> pos;    // original row -> internal position
    std::vector<Quad> quads;     // in internal row order
    std::vector<RowRec> recs;    // [nrows]
    std::vector<int32_t> q0, nq; // [nrows] first quad and number of quads of every row (host side)
    // CSR in internal numbering with log2-domain weights: the exact fallback walks these
    std::vector<int32_t> rowptr;
    std::vector<int32_t> col;
    std::vector<float> w;
    std::vector<uint16_t> pdfstart;  // [P1 + 1] first internal position of each pdf (pdf-major order only)
    int KQ = 1;                      // quads per lane this form was laid out for
    int ncopy = 1;                   // LDS copies of p the offsets refer to
    double conflict_before = 0, conflict_after = 0;  // mean LDS cycles per gather instruction (model)
};

// rowptr/col/val_log2: 0-based CSR (out[r] = (+)_k val[k] (*) in[col[k]]); col_pos: the internal
// position of every column state in the vector the arcs gather from (= the numbering of the
// SAME direction: forward gathers alpha_{n-1}, stored in forward numbering).
QuadGraph make_quads(int64_t nrows, const std::vector<int64_t> &rowptr, const std::vector<int32_t> &col,
                     const std::vector<float> &val_log2, const std::vector<int32_t> &row2pdf, int32_t P1,
                     bool pdf_major, int KQ);

// Number of quads of a CSR matrix, and whether its weights fit the linear path of the quad kernel.
int64_t count_quads(int64_t nrows, const std::vector<int64_t> &rowptr);
// The internal numbering make_quads() gives the rows of one direction (it does not depend on KQ): forward =
// by decreasing number of quads, backward (pdf_major) = states of one pdf adjacent.  order: position -> row,
// pos: row -> position.
void quad_order(int64_t nrows, const std::vector<int64_t> &rowptr, const std::vector<int32_t> &row2pdf, int32_t P1,
                bool pdf_major, std::vector<int32_t> &order, std::vector<int32_t> &pos);

// Fewest arcs from a seed state to every state, following adjacency rowptr/col (row -> its successors):
// 0xffff = unreachable, 0 = seed or unknown (too far to store).  A state can carry weight at a recursion
// step only if that step is at least its distance, so a row whose quad sum is exactly 0 before that is
// known to be zero(K) without walking its arcs (mm_kernel_quad.hip).
std::vector<uint16_t> reach_distance(int64_t n, const std::vector<int64_t> &rowptr, const std::vector<int32_t> &col,
                                     const std::vector<int32_t> &seeds);

bool quad_range_ok(int64_t nrows, const std::vector<int64_t> &rowptr, const std::vector<float> &val_log2, int32_t P1);

// Host evaluation of one product through the packed form, lane by lane, with the
// same group structure as the kernels (test aid).  semiring 0 = log, 1 = tropical.
void eval_packed(const Packed &p, int semiring, const float *in, float *out, int32_t *argmax, int64_t nrows);

}  // namespace mm
