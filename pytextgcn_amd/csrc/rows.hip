// Row movement of the distributed SpMM's exchange (pytextgcn_amd/sharded.py; the reference is single-device,
// flat_amazon.py:84-86, so nothing there corresponds to these).  Three HBM-bound kernels over row-major fp32 rows:
//   tgcn_rows_gather          out[i, :]      = x[idx[i], :]          pack the rows a peer asked for into a send buffer
//   tgcn_rows_scatter         y[idx[i], :]   = x[i, :]               place received rows in the gathered hub block
//   tgcn_rows_reduce_ranked   y[row0 + j * step, :] += sum_q recv[inv[q * n + j], :]   (q = 0 .. W-1 IN ORDER, inv < 0: skip)
//                             add the W ranks' partial rows of the reduce-scatter to this rank's hub rows: one pass
//                             instead of W index_add_ launches, and a summation order (rank order, starting from
//                             zero, then one add into y) that every form of the exchange shares bit for bit.
// One wavefront per row (64 float4 lanes cover 256 columns per pass); plain loads and stores: the rows are about to
// be read by the SpMM (gather / scatter) or were just written by it (reduce).
// Every entry point carries the row count of the indexed buffer: an index outside it is SKIPPED by the kernel (the
// calls only enqueue, so they cannot report it) -- a wrong index list costs a wrong result, never a stray access.
// pytextgcn_amd/sharded.py validates its lists once, when it builds them.
#include <algorithm>

#include "common.h"

namespace tgcn {
namespace {

template <int VEC>
struct RowVec;
template <>
struct RowVec<4> {
    using type = float4;
    static __device__ __forceinline__ type zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ type add(const type &a, const type &b) {
        return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
};
template <>
struct RowVec<1> {
    using type = float;
    static __device__ __forceinline__ type zero() { return 0.f; }
    static __device__ __forceinline__ type add(const type &a, const type &b) { return a + b; }
};

constexpr int kRowWaves = 4;   // rows per 256-thread workgroup

// SCATTER = false: dst row i <- src row idx[i];  SCATTER = true: dst row idx[i] <- src row i
template <int VEC, bool SCATTER>
__global__ __launch_bounds__(64 * kRowWaves) void k_rows_move(const float *__restrict__ src, int64_t lds_,
                                                             const int64_t *__restrict__ idx, int64_t n, int F,
                                                             float *__restrict__ dst, int64_t ldd, int64_t n_indexed) {
    using vec_t = typename RowVec<VEC>::type;
    const int lane = threadIdx.x & 63;
    const int64_t stride = int64_t(gridDim.x) * kRowWaves;
    for (int64_t i = int64_t(blockIdx.x) * kRowWaves + (threadIdx.x >> 6); i < n; i += stride) {
        const int64_t j = idx[i];
        if (j < 0 || j >= n_indexed) continue;                 // wave-uniform
        const float *s = src + (SCATTER ? i : j) * lds_;
        float *d = dst + (SCATTER ? j : i) * ldd;
        for (int c = lane * VEC; c < F; c += 64 * VEC)
            *reinterpret_cast<vec_t *>(d + c) = *reinterpret_cast<const vec_t *>(s + c);
    }
}

template <int VEC>
__global__ __launch_bounds__(64 * kRowWaves) void k_rows_reduce_ranked(const float *__restrict__ recv, int64_t ldr,
                                                                      int64_t n_recv, const int32_t *__restrict__ inv,
                                                                      int W, int64_t n, int F, float *__restrict__ y,
                                                                      int64_t ldy, int64_t row0, int64_t step) {
    using V = RowVec<VEC>;
    using vec_t = typename V::type;
    const int lane = threadIdx.x & 63;
    const int64_t stride = int64_t(gridDim.x) * kRowWaves;
    for (int64_t j = int64_t(blockIdx.x) * kRowWaves + (threadIdx.x >> 6); j < n; j += stride) {
        float *yr = y + (row0 + j * step) * ldy;
        for (int c = lane * VEC; c < F; c += 64 * VEC) {
            vec_t acc = V::zero();
            for (int q = 0; q < W; ++q) {                      // rank order: the summation order of every exchange form
                const int32_t i = inv[int64_t(q) * n + j];     // wave-uniform
                if (i >= 0 && i < n_recv) acc = V::add(acc, *reinterpret_cast<const vec_t *>(recv + int64_t(i) * ldr + c));
            }
            *reinterpret_cast<vec_t *>(yr + c) = V::add(*reinterpret_cast<const vec_t *>(yr + c), acc);
        }
    }
}

int rows_grid(int64_t n) {
    return static_cast<int>(std::max<int64_t>(1, std::min<int64_t>((n + kRowWaves - 1) / kRowWaves, 256 * 32)));
}

bool vec4_ok(int F, int64_t lda, int64_t ldb, const void *a, const void *b) {
    return F % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) % 16) == 0;
}

int check_rows(const char *fn, const void *a, const void *b, const void *idx, int64_t n, int F, int64_t lda,
               int64_t ldb) {
    if (n < 0 || F <= 0 || lda < F || ldb < F || (n > 0 && (!a || !b || !idx))) {
        set_error("%s: bad argument (n=%lld F=%d strides %lld / %lld; pointers must be non-NULL when n > 0)", fn,
                  (long long)n, F, (long long)lda, (long long)ldb);
        return TGCN_E_INVALID;
    }
    return TGCN_OK;
}

}  // namespace
}  // namespace tgcn

extern "C" {

int tgcn_rows_gather(const float *x, int64_t ldx, int64_t n_x_rows, const int64_t *idx, int64_t n, int F, float *out,
                     int64_t ldo, tgcn_stream stream) {
    using namespace tgcn;
    TGCN_CHECK(check_rows("tgcn_rows_gather", x, out, idx, n, F, ldx, ldo));
    if (n == 0) return TGCN_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec4_ok(F, ldx, ldo, x, out))
        k_rows_move<4, false><<<rows_grid(n), 64 * kRowWaves, 0, s>>>(x, ldx, idx, n, F, out, ldo, n_x_rows);
    else
        k_rows_move<1, false><<<rows_grid(n), 64 * kRowWaves, 0, s>>>(x, ldx, idx, n, F, out, ldo, n_x_rows);
    TGCN_HIP_CHECK(hipGetLastError());
    return TGCN_OK;
}

int tgcn_rows_scatter(const float *x, int64_t ldx, const int64_t *idx, int64_t n, int F, float *y, int64_t ldy,
                      int64_t n_y_rows, tgcn_stream stream) {
    using namespace tgcn;
    TGCN_CHECK(check_rows("tgcn_rows_scatter", x, y, idx, n, F, ldx, ldy));
    if (n == 0) return TGCN_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec4_ok(F, ldx, ldy, x, y))
        k_rows_move<4, true><<<rows_grid(n), 64 * kRowWaves, 0, s>>>(x, ldx, idx, n, F, y, ldy, n_y_rows);
    else
        k_rows_move<1, true><<<rows_grid(n), 64 * kRowWaves, 0, s>>>(x, ldx, idx, n, F, y, ldy, n_y_rows);
    TGCN_HIP_CHECK(hipGetLastError());
    return TGCN_OK;
}

int tgcn_rows_reduce_ranked(const float *recv, int64_t ldr, int64_t n_recv_rows, const int32_t *inv, int n_ranks,
                            int64_t n, int F, float *y, int64_t ldy, int64_t n_y_rows, int64_t row0, int64_t row_step,
                            tgcn_stream stream) {
    using namespace tgcn;
    if (n < 0 || F <= 0 || n_ranks <= 0 || ldr < F || ldy < F || row0 < 0 || row_step <= 0 || n_recv_rows < 0 ||
        (n > 0 && (!inv || !y)) || (n_recv_rows > 0 && !recv)) {
        set_error("tgcn_rows_reduce_ranked: bad argument (n=%lld F=%d ranks=%d ldr=%lld ldy=%lld row0=%lld step=%lld)",
                  (long long)n, F, n_ranks, (long long)ldr, (long long)ldy, (long long)row0, (long long)row_step);
        return TGCN_E_INVALID;
    }
    if (n == 0) return TGCN_OK;
    if (row0 + (n - 1) * row_step >= n_y_rows) {
        set_error("tgcn_rows_reduce_ranked: rows %lld + j * %lld, j < %lld, leave the %lld rows of y", (long long)row0,
                  (long long)row_step, (long long)n, (long long)n_y_rows);
        return TGCN_E_RANGE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // recv may be NULL when no rank sent anything (n_recv_rows = 0: every table entry is then skipped)
    if (vec4_ok(F, ldr, ldy, recv, y))
        k_rows_reduce_ranked<4><<<rows_grid(n), 64 * kRowWaves, 0, s>>>(recv, ldr, n_recv_rows, inv, n_ranks, n, F, y, ldy, row0, row_step);
    else
        k_rows_reduce_ranked<1><<<rows_grid(n), 64 * kRowWaves, 0, s>>>(recv, ldr, n_recv_rows, inv, n_ranks, n, F, y, ldy, row0, row_step);
    TGCN_HIP_CHECK(hipGetLastError());
    return TGCN_OK;
}

}  // extern "C"
