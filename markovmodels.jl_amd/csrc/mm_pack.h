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
// "Quad" form for the fast forward-backward kernel (mm_kernel_quad.hip): every
// row is cut into quads of 4 arcs (the last one padded with weight 0); quad q
// belongs to lane q / KQ of the workgroup, which keeps its KQ quads in registers
// for the whole time loop.  Weights are LINEAR (2^w), columns are LDS byte
// offsets (4 * col).  Quads are stored in row order.
// ---------------------------------------------------------------------------
struct Quad {
    float wl[4];       // 2^(log2 weight); 0 = padding
    uint16_t off[4];   // 4 * source column (byte offset into the LDS vector)
    uint16_t rowoff;   // 4 * row
    uint16_t pad;
    uint32_t pad2;
};
static_assert(sizeof(Quad) == 32, "Quad must be 32 bytes");

struct QuadGraph {
    std::vector<Quad> quads;
    // row-ordered CSR with log2-domain weights: the exact fallback walks these
    std::vector<int32_t> rowptr;
    std::vector<int32_t> col;
    std::vector<float> w;
    std::vector<uint16_t> qstart;  // [nrows + 1] first quad of every row
    std::vector<uint16_t> rord;    // [nrows] rows by decreasing number of quads (balanced row-finishing loops)
    bool fast_ok = false;  // all weights inside the range the linear path is valid for
};

QuadGraph make_quads(int64_t nrows, const std::vector<int64_t> &rowptr, const std::vector<int32_t> &col,
                     const std::vector<float> &val_log2);

// Host evaluation of one product through the packed form, lane by lane, with the
// same group structure as the kernels (test aid).  semiring 0 = log, 1 = tropical.
void eval_packed(const Packed &p, int semiring, const float *in, float *out, int32_t *argmax, int64_t nrows);

}  // namespace mm
