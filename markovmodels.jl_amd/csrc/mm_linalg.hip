// mm_linalg.hip -- the reference's semiring sparse linear algebra at the C boundary: the `mul!` methods and the sparse-vector
// broadcast that src/linalg.jl defines on CuSparse containers, as stand-alone device operations on caller-owned CSR arrays.
//   mm_spmv   LinearAlgebra.mul!(c, A::CuSparseMatrixCSR{K}, b)            src/linalg.jl:163-184, kernel :213-233, warp_reduce :204-211
//   mm_spmm   LinearAlgebra.mul!(C, A::CuSparseMatrixCSR{K}, B, alpha, beta)   :240-262, kernel :268-280
//   mm_svdv   _copyto!(f, dest, x::CuSparseVector{K}, y), f in {*, /}          :294-315, kernel :320-328
// Generic in K like the reference: LogSemiring / TropicalSemiring / ProbSemiring x Float32 / Float64 (test/test_linalg.jl:88-108
// runs exactly these six).  The forward-backward kernels never call these -- they keep the graph in registers and fuse the
// products into their time loops --; this file is the seam for a caller who uses the package's linear algebra directly.
//
// SpMV: where the reference gives every row a 32-lane warp whatever its length (17 of 32 lanes busy on an LF-MMI denominator,
// 2 of 32 on a numerator), a row gets an aligned GROUP of g = 1 ... 64 lanes of a 64-lane wave, g the largest power of two up to
// the mean row length: the lanes of a group read consecutive entries of the row (coalesced nzVal / colVal), keep a running
// (maximum, scaled sum) pair -- one exp per entry, no second pass over the row --, and the group combines with DPP butterflies
// (quad_perm, row_half_mirror, row_mirror; the LDS crossbar only above 16 lanes).
// SpMM: a thread per row and JT columns, lanes along the rows (column-major C and B: consecutive lanes write consecutive
// addresses); the row's indices are read once per JT columns instead of once per column.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

#include "mm_internal.h"

namespace mm {
namespace {

template <typename T>
struct Lim;
template <>
struct Lim<float> {
    static __device__ __forceinline__ float ninf() { return -__builtin_huge_valf(); }
    static __device__ __forceinline__ float nan() { return __builtin_nanf(""); }
};
template <>
struct Lim<double> {
    static __device__ __forceinline__ double ninf() { return -__builtin_huge_val(); }
    static __device__ __forceinline__ double nan() { return __builtin_nan(""); }
};

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_x(float v) {
    return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v)));
}
template <int CTRL>
__device__ __forceinline__ double dpp_x(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = dpp_i<CTRL>(int(b)), hi = dpp_i<CTRL>(int(b >> 32));
    return __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
// the value of the lane whose number differs in bit LEVEL (butterfly partner); levels 0..3 stay inside a 16-lane DPP row
template <int LEVEL, typename T>
__device__ __forceinline__ T partner(T v) {
    if constexpr (LEVEL == 0) return dpp_x<0xB1>(v);        // quad_perm [1,0,3,2]
    else if constexpr (LEVEL == 1) return dpp_x<0x4E>(v);   // quad_perm [2,3,0,1]
    else if constexpr (LEVEL == 2) return dpp_x<0x141>(v);  // row_half_mirror (the lower levels are uniform already)
    else if constexpr (LEVEL == 3) return dpp_x<0x140>(v);  // row_mirror
    else return __shfl_xor(v, 1 << LEVEL);
}

// one semiring's running reduction of a lane: Log keeps (m, s) with the sum as s * e^m; Tropical m; Prob s
template <typename T, int SR>
struct Acc {
    T m, s;
    __device__ __forceinline__ void init() {
        m = Lim<T>::ninf();
        s = T(0);
    }
    __device__ __forceinline__ void init_from(T c) {  // start from a stored element (beta != 0)
        if constexpr (SR == MM_PROB) {
            m = T(0);
            s = c;
        } else {
            m = c;
            s = c > Lim<T>::ninf() ? T(1) : T(0);
        }
    }
    // (a NaN term must reach the result like in the reference's logaddexp / max -- a diverged network's outputs are not to be
    // hidden: `x > m` is false for a NaN on either side, so NaNs are taken explicitly; once m is a NaN it stays one)
    __device__ __forceinline__ void add(T x) {  // (+)= x
        if constexpr (SR == MM_PROB) s += x;
        else if constexpr (SR == MM_TROPICAL) m = (x > m || x != x) ? x : m;
        else {
            if (x > m) {
                s = s * std::exp(m - x) + T(1);  // (m = -inf: s = 0 * 0 + 1)
                m = x;
            } else if (x > Lim<T>::ninf()) {
                s += std::exp(x - m);  // (m a NaN: s becomes one, value() returns m)
            } else if (x != x) {
                m = x;
                s = T(1);
            }
        }
    }
    template <int LEVEL>
    __device__ __forceinline__ void merge() {  // with the butterfly partner's
        if constexpr (SR == MM_PROB) s += partner<LEVEL>(s);
        else if constexpr (SR == MM_TROPICAL) {
            const T o = partner<LEVEL>(m);
            m = (o > m || o != o) ? o : m;
        } else {
            const T om = partner<LEVEL>(m), os = partner<LEVEL>(s);
            const T M = (om > m || om != om) ? om : m;  // (a NaN on either side wins)
            const T a = m > Lim<T>::ninf() ? s * std::exp(m - M) : T(0), b = om > Lim<T>::ninf() ? os * std::exp(om - M) : T(0);
            m = M;
            s = a + b;
        }
    }
    __device__ __forceinline__ T value() const {
        if constexpr (SR == MM_PROB) return s;
        else if constexpr (SR == MM_TROPICAL) return m;
        else return m > Lim<T>::ninf() ? m + std::log(s) : m;  // (m a NaN: returned as it is)
    }
};

template <typename T, int SR>
__device__ __forceinline__ T sr_mul(T a, T b) {
    if constexpr (SR == MM_PROB) return a * b;
    else return a + b;  // (-inf + x = -inf: zero(K) annihilates)
}
template <typename T, int SR>
__device__ __forceinline__ T sr_div(T a, T b) {
    if constexpr (SR == MM_PROB) return a / b;
    else return a - b;
}
template <typename T, int SR>
__device__ __forceinline__ T sr_zero() {
    if constexpr (SR == MM_PROB) return T(0);
    else return Lim<T>::ninf();
}

// c[r] = (+)_k nzVal[k] (*) b[colVal[k]] over CSR row r (src/linalg.jl:213-233): a group of 1 << LOG2G lanes per row
template <typename T, int SR, int LOG2G>
__global__ __launch_bounds__(256) void mm_spmv_kernel(long long rows, long long cols, const int *__restrict__ rowptr, const int *__restrict__ colval,
                                                       int base, const T *__restrict__ nzval, const T *__restrict__ b, T *__restrict__ c) {
    constexpr int G = 1 << LOG2G;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long r = tid >> LOG2G;
    const int sub = int(tid) & (G - 1);
    const bool live = r < rows;
    const long long beg = live ? (long long)rowptr[r] - base : 0, end = live ? (long long)rowptr[r + 1] - base : 0;
    Acc<T, SR> acc;
    acc.init();
    // (a column index outside [0, cols) -- a corrupt matrix -- contributes a NaN instead of a read out of bounds)
    for (long long k = beg + sub; k < end; k += G) {
        const long long j = (long long)colval[k] - base;
        acc.add((unsigned long long)j < (unsigned long long)cols ? sr_mul<T, SR>(nzval[k], b[j]) : Lim<T>::nan());
    }
    if constexpr (LOG2G >= 1) acc.template merge<0>();
    if constexpr (LOG2G >= 2) acc.template merge<1>();
    if constexpr (LOG2G >= 3) acc.template merge<2>();
    if constexpr (LOG2G >= 4) acc.template merge<3>();
    if constexpr (LOG2G >= 5) acc.template merge<4>();
    if constexpr (LOG2G >= 6) acc.template merge<5>();
    if (live && sub == 0) c[r] = acc.value();
}

// C[i, j] = (beta (*) C[i, j]) (+) (+)_k nzVal[k] (*) B[colVal[k], j] (src/linalg.jl:240-280), column-major B and C
template <typename T, int SR, int JT>
__global__ __launch_bounds__(256) void mm_spmm_kernel(long long rows, long long acols, long long ncols, const int *__restrict__ rowptr,
                                                       const int *__restrict__ colval, int base, const T *__restrict__ nzval,
                                                       const T *__restrict__ B, long long ldb, T *__restrict__ C, long long ldc, int beta_mode,
                                                       T beta) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const long long beg = (long long)rowptr[i] - base, end = (long long)rowptr[i + 1] - base;
    // (the column tiles of a row block are a loop of ONE workgroup, not a grid dimension: the row's indices and weights come from HBM
    // once and from this XCD's L2 for the other tiles -- as grid.y = tiles, every tile's workgroups re-read the CSR arrays and the
    // rows of B through whichever XCD they landed on: 143 MB fetched for 12 MB of operands, profiles/r06_traffic_linalg.json)
    for (long long j0 = (long long)blockIdx.y * JT; j0 < ncols; j0 += (long long)gridDim.y * JT) {
        Acc<T, SR> acc[JT];
#pragma unroll
        for (int jj = 0; jj < JT; ++jj) {
            // beta_mode 0: fill!(C, zero(K)) (:247); 1: C as it is; 2: rmul!(C, beta) first
            if (beta_mode == 0 || j0 + jj >= ncols) acc[jj].init();
            else {
                const T c0 = C[i + (j0 + jj) * ldc];
                acc[jj].init_from(beta_mode == 1 ? c0 : sr_mul<T, SR>(c0, beta));
            }
        }
        for (long long k = beg; k < end; ++k) {
            const T w = nzval[k];
            const long long cj = (long long)colval[k] - base;
            const bool inside = (unsigned long long)cj < (unsigned long long)acols;  // (outside: a NaN, not a read out of bounds)
            const T *brow = B + (inside ? cj : 0);
#pragma unroll
            for (int jj = 0; jj < JT; ++jj)
                if (j0 + jj < ncols) acc[jj].add(inside ? sr_mul<T, SR>(w, brow[(j0 + jj) * ldb]) : Lim<T>::nan());
        }
#pragma unroll
        for (int jj = 0; jj < JT; ++jj)
            if (j0 + jj < ncols) C[i + (j0 + jj) * ldc] = acc[jj].value();
    }
}

template <typename T, int SR>
__global__ void mm_fill_zero_kernel(long long n, T *dest) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dest[i] = sr_zero<T, SR>();
}
// dest[nzInd[i]] = f(nzVal[i], y[nzInd[i]]) (src/linalg.jl:320-328)
template <typename T, int SR>
__global__ void mm_svdv_kernel(long long n, long long nnz, const int *__restrict__ nzind, int base, const T *__restrict__ nzval, const T *__restrict__ y,
                               T *__restrict__ dest, int op) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    const long long at = (long long)nzind[i] - base;
    if ((unsigned long long)at >= (unsigned long long)n) return;  // (an index outside the vector: nothing to write to)
    dest[at] = op == 0 ? sr_mul<T, SR>(nzval[i], y[at]) : sr_div<T, SR>(nzval[i], y[at]);
}

template <typename T, int SR>
int spmv_launch(int log2g, long long rows, long long cols, const int *rowptr, const int *colval, int base, const T *nzval, const T *b, T *c, hipStream_t s) {
    const long long threads = rows << log2g;
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
#define MM_SPMV_CASE(L)                                                                                                  \
    case L:                                                                                                              \
        hipLaunchKernelGGL((mm_spmv_kernel<T, SR, L>), grid, block, 0, s, rows, cols, rowptr, colval, base, nzval, b, c); \
        break;
    switch (log2g) {
        MM_SPMV_CASE(0)
        MM_SPMV_CASE(1)
        MM_SPMV_CASE(2)
        MM_SPMV_CASE(3)
        MM_SPMV_CASE(4)
        MM_SPMV_CASE(5)
        default: MM_SPMV_CASE(6)
    }
#undef MM_SPMV_CASE
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

int check_common(const char *who, int semiring, int val_bytes, int index_base, const void *stream) {
    (void)stream;
    if (semiring != MM_LOG && semiring != MM_TROPICAL && semiring != MM_PROB) return mm_fail(MM_ERR_INVALID, std::string(who) + ": unknown semiring");
    if (val_bytes != 4 && val_bytes != 8) return mm_fail(MM_ERR_INVALID, std::string(who) + ": val_bytes must be 4 or 8");
    if (index_base != 0 && index_base != 1) return mm_fail(MM_ERR_INVALID, std::string(who) + ": index_base must be 0 or 1");
    return MM_OK;
}

// the group of lanes a row gets: the largest power of two up to the mean row length, at most a wave -- measured on the reference's WSJ
// denominator (blockdiag x 128, mean 17.1 entries per row; tools/bench_linalg.py): groups of 4 / 8 / 16 / 32 lanes 0.130 / 0.095 /
// 0.084 / 0.109 ms: with the power of two ABOVE the mean (round 5) half the lanes of a group had no entry and every row paid a fifth
// butterfly level.  (MM_SPMV_L2G under MM_DEBUG, read once per process like the engine's other diagnostic switches: a fixed one)
int debug_log2g() {
    static const int v = [] {
        const char *on = getenv("MM_DEBUG"), *e = getenv("MM_SPMV_L2G");
        return (on && *on && strcmp(on, "0") && e) ? std::max(0, std::min(6, atoi(e))) : -1;
    }();
    return v;
}
int pick_log2g(long long rows, long long nnz) {
    if (debug_log2g() >= 0) return debug_log2g();
    const double mean = rows > 0 ? double(nnz) / double(rows) : 0.0;
    int l = 0;
    while (l < 6 && double(2 << l) <= mean) ++l;
    return l;
}

#define MM_DISPATCH(SR_, VB_, CALL)                                                                  \
    do {                                                                                             \
        if (VB_ == 4) {                                                                              \
            using T = float;                                                                         \
            if (SR_ == MM_LOG) { constexpr int SR = MM_LOG; CALL; }                                  \
            else if (SR_ == MM_TROPICAL) { constexpr int SR = MM_TROPICAL; CALL; }                   \
            else { constexpr int SR = MM_PROB; CALL; }                                               \
        } else {                                                                                     \
            using T = double;                                                                        \
            if (SR_ == MM_LOG) { constexpr int SR = MM_LOG; CALL; }                                  \
            else if (SR_ == MM_TROPICAL) { constexpr int SR = MM_TROPICAL; CALL; }                   \
            else { constexpr int SR = MM_PROB; CALL; }                                               \
        }                                                                                            \
    } while (0)

}  // namespace
}  // namespace mm

using namespace mm;

extern "C" {

int mm_spmv(int semiring, int val_bytes, int64_t rows, int64_t cols, int64_t nnz, const int32_t *rowptr, const int32_t *colval, int index_base,
            const void *nzval, const void *b, int64_t b_len, void *c, int64_t c_len, void *stream) {
    int rc = check_common("mm_spmv", semiring, val_bytes, index_base, stream);
    if (rc) return rc;
    if (rows < 0 || cols < 0 || nnz < 0) return mm_fail(MM_ERR_INVALID, "mm_spmv: negative size");
    // @boundscheck size(A, 2) == size(b, 1), size(A, 1) == size(c, 1) (src/linalg.jl:166-167)
    if (cols != b_len || rows != c_len) return mm_fail(MM_ERR_DIM, "mm_spmv: DimensionMismatch");
    if (nnz == 0 || rows == 0) return MM_OK;  // `if length(A.nzVal) > 0` (:169): c is left as it is
    if (!rowptr || !colval || !nzval || !b || !c) return mm_fail(MM_ERR_INVALID, "mm_spmv: NULL pointer");
    if ((rows << 6) >= (int64_t(1) << 39)) return mm_fail(MM_ERR_UNSUPPORTED, "mm_spmv: too many rows for one launch");
    const int l2g = pick_log2g(rows, nnz);
    hipStream_t s = static_cast<hipStream_t>(stream);
    MM_DISPATCH(semiring, val_bytes,
                rc = (spmv_launch<T, SR>(l2g, rows, cols, rowptr, colval, index_base, static_cast<const T *>(nzval), static_cast<const T *>(b), static_cast<T *>(c), s)));
    return rc;
}

int mm_spmm(int semiring, int val_bytes, int64_t rows, int64_t cols, int64_t nnz, const int32_t *rowptr, const int32_t *colval, int index_base,
            const void *nzval, const void *B, int64_t b_rows, int64_t b_cols, int64_t ldb, void *C, int64_t c_rows, int64_t c_cols, int64_t ldc,
            double beta, void *stream) {
    int rc = check_common("mm_spmm", semiring, val_bytes, index_base, stream);
    if (rc) return rc;
    if (rows < 0 || cols < 0 || nnz < 0 || b_cols < 0) return mm_fail(MM_ERR_INVALID, "mm_spmm: negative size");
    // src/linalg.jl:242-244
    if (cols != b_rows || rows != c_rows || b_cols != c_cols) return mm_fail(MM_ERR_DIM, "mm_spmm: DimensionMismatch");
    if (ldb < b_rows || ldc < c_rows) return mm_fail(MM_ERR_DIM, "mm_spmm: leading dimension smaller than the rows");
    if (rows == 0 || c_cols == 0) return MM_OK;
    if (!C || (nnz > 0 && (!rowptr || !colval || !nzval || !B))) return mm_fail(MM_ERR_INVALID, "mm_spmm: NULL pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int beta_mode = beta == 0.0 ? 0 : (beta == 1.0 ? 1 : 2);
    if (nnz == 0) {
        // `if length(A.nzVal) > 0` (:249): only the beta step (:246-248) happens
        if (beta_mode == 1) return MM_OK;
        if (!rowptr) return mm_fail(MM_ERR_INVALID, "mm_spmm: NULL rowptr");
    }
    constexpr int JT = 4;
    // (row blocks x as many column-tile lanes as it takes to fill the chip ~16 times over; a workgroup loops over the tiles of its lane)
    const long long xb = (rows + 255) / 256, tiles = (c_cols + JT - 1) / JT;
    const dim3 grid((unsigned)xb, (unsigned)std::min<long long>(tiles, std::max<long long>(1, (4096 + xb - 1) / xb))), block(256);
    MM_DISPATCH(semiring, val_bytes,
                hipLaunchKernelGGL((mm_spmm_kernel<T, SR, JT>), grid, block, 0, s, (long long)rows, (long long)cols, (long long)c_cols, rowptr, colval, index_base,
                                   static_cast<const T *>(nzval), static_cast<const T *>(B), (long long)ldb, static_cast<T *>(C), (long long)ldc,
                                   beta_mode, T(beta)));
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

int mm_svdv(int semiring, int val_bytes, int op, int64_t n, int64_t nnz, const int32_t *nzind, int index_base, const void *nzval, const void *y,
            int64_t y_len, void *dest, int64_t dest_len, void *stream) {
    int rc = check_common("mm_svdv", semiring, val_bytes, index_base, stream);
    if (rc) return rc;
    if (op != 0 && op != 1) return mm_fail(MM_ERR_INVALID, "mm_svdv: op must be 0 (*) or 1 (/)");
    if (n < 0 || nnz < 0) return mm_fail(MM_ERR_INVALID, "mm_svdv: negative size");
    if (y_len != n || dest_len != n) return mm_fail(MM_ERR_DIM, "mm_svdv: DimensionMismatch");
    if (n == 0) return MM_OK;
    if (!dest || (nnz > 0 && (!nzind || !nzval || !y))) return mm_fail(MM_ERR_INVALID, "mm_svdv: NULL pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 block(256);
    MM_DISPATCH(semiring, val_bytes, {
        hipLaunchKernelGGL((mm_fill_zero_kernel<T, SR>), dim3((unsigned)((n + 255) / 256)), block, 0, s, (long long)n, static_cast<T *>(dest));  // fill!(dest, zero(K)) (:297)
        if (nnz > 0)
            hipLaunchKernelGGL((mm_svdv_kernel<T, SR>), dim3((unsigned)((nnz + 255) / 256)), block, 0, s, (long long)n, (long long)nnz, nzind, index_base,
                               static_cast<const T *>(nzval), static_cast<const T *>(y), static_cast<T *>(dest), op);
    });
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

}  // extern "C"
