// Plan construction: COO edge list -> normalised operator M (and M^T) in CSR + work-item partition.
//
// Replaces PyG-1.6.3 gcn_norm / add_remaining_self_loops, which the reference re-executes inside
// every GCNConv call (textgcn/lib/models.py:11-15 construct the layers with cached=False and
// models.py:20 calls them): here it runs once per graph, on the device, and is kept.
// One-off work, so the sort is rocPRIM's device radix sort; everything else is small hand kernels.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace tgcn {

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t n) {
        bytes = n;
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu bytes): %s", n, hipGetErrorString(e));
            return TGCN_E_NOMEM;
        }
        return TGCN_OK;
    }
    template <class T>
    T *as() const {
        return static_cast<T *>(p);
    }
    void *release() {
        void *q = p;
        p = nullptr;
        return q;
    }
};

constexpr int kThreads = 256;

// `cap` is for grid-stride kernels only; kernels that map one thread (or wave) to one element take the
// default, the launch limit itself (2^31 - 1 workgroups in x), so that no element is ever left out
inline int grid_for(int64_t n, int per_block = kThreads, int cap = INT32_MAX) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return static_cast<int>(g);
}

// flags[0] = index out of range seen; flags[1] = number of self-loop edges in the input
__global__ void k_scan_edges(int64_t E, const int64_t *__restrict__ src, int64_t ss,
                             const int64_t *__restrict__ dst, int64_t ds, int64_t n_rows,
                             int64_t n_cols, int add_loops,
                             unsigned long long *__restrict__ loop_eid, unsigned int *flags) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; e < E; e += stride) {
        const int64_t s = src[e * ss], d = dst[e * ds];
        if (s < 0 || s >= n_cols || d < 0 || d >= n_rows) {
            flags[0] = 1u;
            continue;
        }
        if (add_loops && s == d) {
            // last duplicate in edge order wins (PyG index assignment on CPU)
            atomicMax(&loop_eid[s], static_cast<unsigned long long>(e + 1));
            atomicAdd(&flags[1], 1u);
        }
    }
}

// key = dst << 32 | src; dropped entries (input self-loops when loops are re-appended) get the
// key N << 32 so they sort behind every real row.
__global__ void k_make_keys(int64_t E, const int64_t *__restrict__ src, int64_t ss,
                            const int64_t *__restrict__ dst, int64_t ds,
                            const float *__restrict__ w, int64_t N, int64_t n_cols, int add_loops,
                            float loop_fill, const unsigned long long *__restrict__ loop_eid,
                            uint64_t *__restrict__ keys, float *__restrict__ vals) {
    const int64_t total = E + (add_loops ? N : 0);
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    const uint64_t drop = static_cast<uint64_t>(N) << 32;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
        if (i < E) {
            const int64_t s = src[i * ss], d = dst[i * ds];
            const bool bad = s < 0 || s >= n_cols || d < 0 || d >= N;
            keys[i] = (bad || (add_loops && s == d))
                          ? drop
                          : (static_cast<uint64_t>(d) << 32) | static_cast<uint64_t>(s);
            vals[i] = w ? w[i] : 1.0f;
        } else {
            const int64_t n = i - E;
            const unsigned long long le = loop_eid[n];
            keys[i] = (static_cast<uint64_t>(n) << 32) | static_cast<uint64_t>(n);
            vals[i] = le ? (w ? w[le - 1] : 1.0f) : loop_fill;
        }
    }
}

// ---- degree factors of gcn_norm: ONE routine (degree_factors, below) behind tgcn_plan_create and tgcn_gcn_norm -------
// The edge list is walked in bounded chunks; a chunk is sorted by TARGET only (32-bit keys, stable: the entries of a
// row keep their edge order) and every row's entries are summed by one wavefront, in one of two ways:
//   accurate   float64 partial sums per lane, folded in a fixed order, accumulated in float64 over the chunks and
//              rounded to fp32 ONCE together with the loop weight: the correctly rounded degree for all practical purposes;
//   reference  what PyG-1.6.3 gcn_norm does on the CPU (the reference's path, textgcn/lib/models.py:11-20 ->
//              scatter_add(edge_weight, col)): ONE fp32 accumulator per node, the weights added sequentially in edge
//              order, the node's self loop last (add_remaining_self_loops moves the loops to the tail).  A hub node's
//              ~10^6 terms then carry the rounding of a sequential fp32 sum -- this mode reproduces those bits.
// keys of the chunk of edges [lo, lo + n): the target; dropped entries (input self-loops when loops are re-appended,
// out-of-range indices) get the key N and sort behind every real row
__global__ void k_deg_keys(int64_t lo, int64_t n, const int64_t *__restrict__ src, int64_t ss,
                           const int64_t *__restrict__ dst, int64_t ds, const float *__restrict__ w, int64_t n_rows,
                           int64_t n_cols, int add_loops, uint32_t *__restrict__ keys, float *__restrict__ vals) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t s = src[(lo + i) * ss], d = dst[(lo + i) * ds];
        const bool bad = s < 0 || s >= n_cols || d < 0 || d >= n_rows;
        keys[i] = (bad || (add_loops && s == d)) ? static_cast<uint32_t>(n_rows) : static_cast<uint32_t>(d);
        vals[i] = w ? w[lo + i] : 1.0f;
    }
}

// rowptr[r] = first position of the sorted chunk whose key is >= r
__global__ void k_deg_rowptr(const uint32_t *__restrict__ keys, int64_t n, int64_t N, int32_t *__restrict__ rowptr) {
    const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r > N) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < static_cast<uint32_t>(r))
            lo = mid + 1;
        else
            hi = mid;
    }
    rowptr[r] = static_cast<int32_t>(lo);
}

// accurate: one wavefront per row of a sorted chunk, float64 sums in a fixed lane order, added to the row's float64
__global__ void k_deg_sum_f64(const int32_t *__restrict__ rowptr, const float *__restrict__ vals, int64_t N,
                              double *__restrict__ deg) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = int64_t(gridDim.x) * (blockDim.x >> 6);
    // grid-stride over the rows: the launch grid is capped, N is not -- config c5 has 8 M rows
    for (int64_t r = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6); r < N; r += waves) {
        const int32_t b = rowptr[r], e = rowptr[r + 1];
        if (b == e) continue;
        // four independent chains per lane: the hottest word row of c4 has ~10^6 entries
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        int32_t j = b + lane;
        for (; j + 3 * 64 < e; j += 4 * 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += static_cast<double>(vals[j + u * 64]);
        }
        for (; j < e; j += 64) acc[0] += static_cast<double>(vals[j]);
        double s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) deg[r] += s;
    }
}

// reference: one wavefront per row, but ONE fp32 accumulator: the lanes fetch 64 weights at a time (coalesced) and the
// wave adds them one after the other in edge order, continuing the sum the earlier chunks left in deg[r]
__global__ void k_deg_sum_seq(const int32_t *__restrict__ rowptr, const float *__restrict__ vals, int64_t N,
                              float *__restrict__ deg) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int64_t waves = int64_t(gridDim.x) * (blockDim.x >> 6);
    for (int64_t r = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6); r < N; r += waves) {
        const int32_t b = rowptr[r], e = rowptr[r + 1];
        if (b == e) continue;
        float s = deg[r];
        float next = b + lane < e ? vals[b + lane] : 0.f;
        for (int32_t j = b; j < e; j += 64) {
            const float v = next;
            if (j + 64 < e) next = j + 64 + lane < e ? vals[j + 64 + lane] : 0.f;      // prefetch under the chain
            const int cnt = min(64, e - j);
            if (cnt == 64) {
#pragma unroll
                for (int i = 0; i < 64; ++i) s += __shfl(v, i, 64);
            } else {
                for (int i = 0; i < cnt; ++i) s += __shfl(v, i, 64);
            }
        }
        if (lane == 0) deg[r] = s;
    }
}

// loop weight of every node (the last input loop's weight, else `fill`), added to the degree LAST; dis = deg^-1/2
__global__ void k_deg_finish(int64_t N, int reference, const double *__restrict__ deg64, const float *__restrict__ deg32,
                             int add_loops, float fill, const unsigned long long *__restrict__ loop_eid,
                             const float *__restrict__ w, float *__restrict__ dis, float *__restrict__ loop_w) {
#pragma clang fp contract(off)
    const int64_t n = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float lw = 0.f;
    if (add_loops) {
        const unsigned long long le = loop_eid[n];
        lw = le ? (w ? w[le - 1] : 1.0f) : fill;
    }
    if (loop_w) loop_w[n] = lw;
    const float deg = reference ? deg32[n] + lw : static_cast<float>(deg64[n] + static_cast<double>(lw));
    float d = 1.0f / sqrtf(deg);          // deg.pow(-0.5): torch evaluates it as 1 / sqrt, both correctly rounded
    if (isinf(d)) d = 0.0f;               // masked_fill_(== inf, 0)
    dis[n] = d;
}

// rowptr[r] = first position whose key has row >= r
__global__ void k_rowptr(const uint64_t *__restrict__ keys, int64_t nnz, int64_t N,
                         int32_t *__restrict__ rowptr) {
    const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r > N) return;
    const uint64_t target = static_cast<uint64_t>(r) << 32;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < target)
            lo = mid + 1;
        else
            hi = mid;
    }
    rowptr[r] = static_cast<int32_t>(lo);
}

// w_hat = w * (dis[src] * dis[dst]).  PyG evaluates dis[src] * w * dis[dst] left to right, which
// rounds (i,j) and (j,i) differently; pairing the two degree factors first keeps a symmetric graph
// bitwise symmetric (so M^T need not be stored) at <= 1 ulp from PyG's value.  normalize == 2 (the
// reference-order mode) keeps PyG's association, (dis[src] * w) * dis[dst]: the reference's bits, and M^T is
// then stored next to M.  Emits the CSR pair array of M and the (src << 32 | dst, w_hat) pairs that are
// sorted next to give M^T.
__global__ void k_finalize(const uint64_t *__restrict__ keys, const float *__restrict__ vals,
                           int64_t nnz, const float *__restrict__ dis, int normalize,
                           int2 *__restrict__ cv, uint64_t *__restrict__ keys_t,
                           float *__restrict__ vals_t) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < nnz; j += stride) {
        const uint64_t k = keys[j];
        const uint32_t d = static_cast<uint32_t>(k >> 32), s = static_cast<uint32_t>(k);
        float v = vals[j];
        if (normalize == 2)
            v = (dis[s] * v) * dis[d];
        else if (normalize)
            v = v * (dis[s] * dis[d]);
        cv[j] = make_int2(static_cast<int>(s), __float_as_int(v));
        keys_t[j] = (static_cast<uint64_t>(s) << 32) | d;
        vals_t[j] = v;
    }
}

__global__ void k_unpack(const uint64_t *__restrict__ keys, const float *__restrict__ vals,
                         int64_t nnz, int2 *__restrict__ cv) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < nnz; j += stride)
        cv[j] = make_int2(static_cast<int>(static_cast<uint32_t>(keys[j])), __float_as_int(vals[j]));
}

__global__ void k_compare(const int32_t *__restrict__ rp_a, const int32_t *__restrict__ rp_b,
                          int64_t n_rp, const int2 *__restrict__ cv_a,
                          const int2 *__restrict__ cv_b, int64_t nnz, unsigned int *differ) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    bool diff = false;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_rp; i += stride)
        diff |= rp_a[i] != rp_b[i];
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < nnz; j += stride) {
        const int2 a = cv_a[j], b = cv_b[j];
        diff |= (a.x != b.x) | (a.y != b.y);
    }
    if (diff) *differ = 1u;
}

__global__ void k_slice_rowptr(const int32_t *__restrict__ rowptr, int64_t row_begin,
                               int64_t n_rows, int32_t *__restrict__ out) {
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i <= n_rows) out[i] = rowptr[row_begin + i] - rowptr[row_begin];
}

template <class Key>
int sort_pairs(Key *keys_in, Key *keys_out, float *vals_in, float *vals_out, int64_t n,
               unsigned end_bit, hipStream_t stream) {
    if (n == 0) return TGCN_OK;
    size_t tmp_bytes = 0;
    TGCN_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in,
                                             vals_out, static_cast<size_t>(n), 0u, end_bit, stream));
    DevBuf tmp;
    TGCN_CHECK(tmp.alloc(tmp_bytes));
    TGCN_HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys_in, keys_out, vals_in, vals_out,
                                             static_cast<size_t>(n), 0u, end_bit, stream));
    TGCN_HIP_CHECK(hipStreamSynchronize(stream));  // tmp is freed on return
    return TGCN_OK;
}

// The degree factors of gcn_norm over the WHOLE edge list: dis[n] = deg[n]^-1/2 (inf -> 0), loop_w[n] = the weight
// of node n's self loop.  The one routine behind tgcn_plan_create and tgcn_gcn_norm, so that the single-device plan
// and the per-rank operators of the 1-D partition are normalised by the SAME bits.  `loop_eid` is k_scan_edges'
// result (1 + the edge id of the last input loop per node, 0 = none).  Scratch is O(chunk) + O(N); the stream is
// synchronised (scratch is freed on return).
int degree_factors(int64_t N, int64_t n_cols, int64_t E, const int64_t *src, int64_t ss, const int64_t *dst,
                   int64_t ds, const float *w, int add_loops, int reference,
                   const unsigned long long *loop_eid, float *dis, float *loop_w, hipStream_t stream) {
    const char *cs = std::getenv("TGCN_NORM_CHUNK");
    int64_t chunk = cs ? std::atoll(cs) : (int64_t(1) << 25);           // 32 M edges: 512 MB of sort scratch
    chunk = std::max<int64_t>(1024, std::min<int64_t>(chunk, (int64_t(1) << 31) - 2));
    const int64_t cap = std::min(chunk, std::max<int64_t>(E, 1));
    DevBuf keys_a, keys_b, vals_a, vals_b, rowptr, deg;
    TGCN_CHECK(keys_a.alloc(sizeof(uint32_t) * cap));
    TGCN_CHECK(keys_b.alloc(sizeof(uint32_t) * cap));
    TGCN_CHECK(vals_a.alloc(sizeof(float) * cap));
    TGCN_CHECK(vals_b.alloc(sizeof(float) * cap));
    TGCN_CHECK(rowptr.alloc(sizeof(int32_t) * (N + 1)));
    TGCN_CHECK(deg.alloc((reference ? sizeof(float) : sizeof(double)) * N));
    TGCN_HIP_CHECK(hipMemsetAsync(deg.p, 0, deg.bytes, stream));
    unsigned node_bits = 1;
    while ((int64_t(1) << node_bits) <= N) ++node_bits;                // the drop key N must sort too
    for (int64_t lo = 0; lo < E; lo += chunk) {
        const int64_t n = std::min(chunk, E - lo);
        k_deg_keys<<<grid_for(n, kThreads, 8192), kThreads, 0, stream>>>(lo, n, src, ss, dst, ds, w, N, n_cols, add_loops,
                                                                        keys_a.as<uint32_t>(), vals_a.as<float>());
        TGCN_HIP_CHECK(hipGetLastError());
        // stable: the entries of a row keep their edge order (what the reference-order sum needs)
        TGCN_CHECK(sort_pairs(keys_a.as<uint32_t>(), keys_b.as<uint32_t>(), vals_a.as<float>(), vals_b.as<float>(), n,
                              node_bits, stream));
        k_deg_rowptr<<<grid_for(N + 1), kThreads, 0, stream>>>(keys_b.as<uint32_t>(), n, N, rowptr.as<int32_t>());
        TGCN_HIP_CHECK(hipGetLastError());
        if (reference)
            k_deg_sum_seq<<<grid_for(N, kThreads / 64, 1 << 20), kThreads, 0, stream>>>(rowptr.as<int32_t>(),
                                                                                       vals_b.as<float>(), N, deg.as<float>());
        else
            k_deg_sum_f64<<<grid_for(N, kThreads / 64, 1 << 20), kThreads, 0, stream>>>(rowptr.as<int32_t>(),
                                                                                       vals_b.as<float>(), N, deg.as<double>());
        TGCN_HIP_CHECK(hipGetLastError());
    }
    k_deg_finish<<<grid_for(N), kThreads, 0, stream>>>(N, reference, deg.as<double>(), deg.as<float>(), add_loops,
                                                       static_cast<float>(add_loops), loop_eid, w, dis, loop_w);
    TGCN_HIP_CHECK(hipGetLastError());
    TGCN_HIP_CHECK(hipStreamSynchronize(stream));   // scratch is freed on return
    return TGCN_OK;
}

// Keep rows [row_begin, row_end) of a full CSR as an owned block.
int take_block(const int32_t *rowptr_full, const int2 *cv_full, int64_t n_nodes, int64_t row_begin,
               int64_t row_end, CsrBlock &out, hipStream_t stream) {
    int32_t ends[2];
    TGCN_HIP_CHECK(hipMemcpyAsync(&ends[0], rowptr_full + row_begin, sizeof(int32_t),
                                  hipMemcpyDeviceToHost, stream));
    TGCN_HIP_CHECK(hipMemcpyAsync(&ends[1], rowptr_full + row_end, sizeof(int32_t),
                                  hipMemcpyDeviceToHost, stream));
    TGCN_HIP_CHECK(hipStreamSynchronize(stream));
    out.n_rows = row_end - row_begin;
    out.n_cols = n_nodes;
    out.nnz = int64_t(ends[1]) - ends[0];
    DevBuf rp, cv;
    TGCN_CHECK(rp.alloc(sizeof(int32_t) * (out.n_rows + 1)));
    TGCN_CHECK(cv.alloc(sizeof(int2) * out.nnz));
    k_slice_rowptr<<<grid_for(out.n_rows + 1), kThreads, 0, stream>>>(rowptr_full, row_begin,
                                                                     out.n_rows, rp.as<int32_t>());
    TGCN_HIP_CHECK(hipGetLastError());
    if (out.nnz)
        TGCN_HIP_CHECK(hipMemcpyAsync(cv.p, cv_full + ends[0], sizeof(int2) * out.nnz,
                                      hipMemcpyDeviceToDevice, stream));
    TGCN_HIP_CHECK(hipStreamSynchronize(stream));
    out.bytes = rp.bytes + cv.bytes;
    out.rowptr = static_cast<int32_t *>(rp.release());
    out.cv = static_cast<int2 *>(cv.release());
    return TGCN_OK;
}

// pos[i * (n_blocks + 1) + j] = first entry of long row i whose column is >= j * col_block
// (columns are ascending inside a row: the plan is sorted by (row, col)).
__global__ void k_block_cuts(const int32_t *__restrict__ long_rows, int n_long,
                             const int32_t *__restrict__ rowptr, const int2 *__restrict__ cv,
                             int col_block, int n_blocks, int32_t *__restrict__ pos) {
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= int64_t(n_long) * (n_blocks + 1)) return;
    const int i = static_cast<int>(idx / (n_blocks + 1)), j = static_cast<int>(idx % (n_blocks + 1));
    const int r = long_rows[i];
    int lo = rowptr[r], hi = rowptr[r + 1];
    const int64_t target = int64_t(j) * col_block;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cv[mid].x < target)
            lo = mid + 1;
        else
            hi = mid;
    }
    pos[idx] = lo;
}

// Dense value block of the hot rows: vd[c * 32 + k] = sum of the entries (hot_rows[k], c).  One thread
// per entry; duplicates of a column are adjacent (columns are sorted within a row) and are summed, in
// storage order, by the thread that owns the first of them.
__global__ void k_hot_fill(const int32_t *__restrict__ hot_rows, const int32_t *__restrict__ rowptr,
                           const int2 *__restrict__ cv, float *__restrict__ vd) {
    const int k = blockIdx.y;
    const int32_t r = hot_rows[k];
    const int32_t b = rowptr[r], e = rowptr[r + 1];
    for (int32_t j = b + blockIdx.x * blockDim.x + threadIdx.x; j < e; j += gridDim.x * blockDim.x) {
        const int c = cv[j].x;
        if (j > b && cv[j - 1].x == c) continue;
        float s = __int_as_float(cv[j].y);
        for (int32_t q = j + 1; q < e && cv[q].x == c; ++q) s += __int_as_float(cv[q].y);
        vd[int64_t(c) * kHotRows + k] = s;
    }
}

// Tuning knobs of the work partition (environment, read at plan creation); the defaults are the measured best on
// config c4.  Launch order is fixed: long-row segments in column-block order with the row blocks spread evenly
// between them (row order, segments-first, blocks-first and XCD-affine queues were measured and dropped).
struct Knobs {
    int col_block;  // TGCN_COL_BLOCK: columns per block for long-row cuts; 0 = no column cuts
    int min_piece;  // TGCN_MIN_PIECE: merge adjacent column blocks of a row until a piece has this many entries
    int hot_rows;   // TGCN_HOT_ROWS: 0 = never build the dense hot block (pure gather semantics for non-finite operands)
    double hot_ratio;  // TGCN_HOT_RATIO: build it when the 32 longest rows hold >= hot_ratio * n_cols entries
};

Knobs knobs_from_env() {
    auto geti = [](const char *name, int dflt) {
        const char *s = std::getenv(name);
        return s ? std::atoi(s) : dflt;
    };
    Knobs k;
    k.col_block = geti("TGCN_COL_BLOCK", 8192);
    k.min_piece = std::max(1, geti("TGCN_MIN_PIECE", 32));
    k.hot_rows = geti("TGCN_HOT_ROWS", 1);
    const char *hr = std::getenv("TGCN_HOT_RATIO");
    k.hot_ratio = hr ? std::atof(hr) : 2.0;
    return k;
}

// Non-zeros + rows per wavefront task.  TGCN_ITEM_WEIGHT pins it; otherwise it follows the size of the operator:
//   * 384 for whole-graph operators (round-2 re-sweep on c4, profiles/r02r_resweep_items.log: 384 / 32 is 3 % faster at
//     F = 200 and 2 % slower at F = 64 than round 1's 512 / 64);
//   * 128 for operators below 24 M entries + rows -- the per-rank operators A_r / B_r of the 1-D partition (c4 over 8
//     ranks: 2.2 M and 4.4 M entries).  At 384 such an operator is ~10 k tasks for 8 192 resident wavefronts: one round
//     with a ragged tail.  Measured per rank (tools/sim_shard_compute.py --item-weights, profiles/r04_item_weight_sweep.log):
//     8 ranks 0.644 -> 0.597 ms per distributed SpMM at F = 200 (0.246 -> 0.229 at F = 64), 4 ranks 1.151 -> 1.105,
//     2 ranks 2.142 -> 2.073; 96 is level with 128, 64 slower again.
int item_weight_for(const CsrBlock &b) {
    const char *s = std::getenv("TGCN_ITEM_WEIGHT");
    const int v = s ? std::atoi(s) : 0;
    if (v >= 64 && v <= (1 << 20)) return v;
    return b.nnz + b.n_rows < (int64_t(24) << 20) ? 128 : 384;
}

}  // namespace

void free_block(CsrBlock &b) {
    if (b.rowptr) (void)hipFree(b.rowptr);
    if (b.cv) (void)hipFree(b.cv);
    if (b.items) (void)hipFree(b.items);
    if (b.fix) (void)hipFree(b.fix);
    if (b.row_info) (void)hipFree(b.row_info);
    if (b.items_all) (void)hipFree(b.items_all);
    if (b.fix_all) (void)hipFree(b.fix_all);
    if (b.hot_vals) (void)hipFree(b.hot_vals);
    b = CsrBlock{};
}

// Host-side work partition (one pass over rowptr, a few ms at 2 M rows).
//   * rows of degree <= T are packed, whole, into ROW BLOCKS of weight ~T (weight of a row =
//     degree + 1: its gathered rows plus its output row);
//   * a row of degree > T is cut into SEGMENTS whose partial sums a second pass adds up in a fixed
//     order.  Cuts fall on column-block boundaries (blocks of `col_block` columns, adjacent blocks
//     merged until a piece has >= min_piece entries, pieces > T split evenly), and segments are
//     LAUNCHED in column-block order: the waves resident at any moment then gather from the same
//     slice of X, which is fetched from HBM once and re-used from L2 / Infinity Cache by the other
//     long rows (a hub word row shares its document columns with every other hub row);
//   * row blocks are spread evenly between the segments in launch order, so cache-friendly and
//     streaming work overlap in time.
int build_items(CsrBlock &b, int T, hipStream_t stream) {
    const Knobs kn = knobs_from_env();
    std::vector<int32_t> rp(static_cast<size_t>(b.n_rows) + 1);
    TGCN_HIP_CHECK(hipMemcpyAsync(rp.data(), b.rowptr, sizeof(int32_t) * rp.size(),
                                  hipMemcpyDeviceToHost, stream));
    TGCN_HIP_CHECK(hipStreamSynchronize(stream));
    const int32_t n_rows = static_cast<int32_t>(b.n_rows);

    // pass 1: row blocks and the list of long rows
    std::vector<WorkItem> blocks;
    std::vector<int32_t> long_rows;
    blocks.reserve(static_cast<size_t>((b.nnz + b.n_rows) / T + 16));
    {
        int32_t r0 = 0;
        int64_t wsum = 0;
        for (int32_t r = 0; r < n_rows; ++r) {
            const int32_t d = rp[r + 1] - rp[r];
            if (d > T) {
                if (r > r0) blocks.push_back({r0, r, rp[r0], rp[r]});
                long_rows.push_back(r);
                r0 = r + 1;
                wsum = 0;
            } else {
                wsum += d + 1;
                if (wsum >= T) {
                    blocks.push_back({r0, r + 1, rp[r0], rp[r + 1]});
                    r0 = r + 1;
                    wsum = 0;
                }
            }
        }
        if (r0 < n_rows) blocks.push_back({r0, n_rows, rp[r0], rp[n_rows]});
    }
    // rows of each block by falling degree (stable: equal degrees keep row order), for the sub-group kernels
    if (b.row_info) {
        (void)hipFree(b.row_info);
        b.row_info = nullptr;
    }
    if (n_rows > 0) {
        std::vector<int32_t> perm(static_cast<size_t>(n_rows));
        std::vector<int4> info(static_cast<size_t>(n_rows));
        // blocks are independent: a few host threads share them (2 M rows: ~10 ms instead of ~40)
        const int n_thr = static_cast<int>(std::max<size_t>(1, std::min<size_t>({8, std::thread::hardware_concurrency(), blocks.size() / 4096 + 1})));
        auto work = [&](int t) {
            const size_t b0 = blocks.size() * t / n_thr, b1 = blocks.size() * (t + 1) / n_thr;
            for (size_t k = b0; k < b1; ++k) {
                const WorkItem &it = blocks[k];
                for (int32_t r = it.row_begin; r < it.row_end; ++r) perm[r] = r;
                if (it.row_end - it.row_begin > 1)
                    std::stable_sort(perm.begin() + it.row_begin, perm.begin() + it.row_end,
                                     [&](int32_t x, int32_t y) { return rp[x + 1] - rp[x] > rp[y + 1] - rp[y]; });
                for (int32_t i = it.row_begin; i < it.row_end; ++i)
                    info[i] = make_int4(rp[perm[i]], rp[perm[i] + 1], perm[i], 0);
            }
        };
        // rows outside every block (the long rows) keep their own record: the sub-group kernels never read it
        for (int32_t r : long_rows) info[r] = make_int4(rp[r], rp[r + 1], r, 0);
        std::vector<std::thread> pool;
        for (int t = 1; t < n_thr; ++t) pool.emplace_back(work, t);
        work(0);
        for (std::thread &th : pool) th.join();
        DevBuf d_info;
        TGCN_CHECK(d_info.alloc(sizeof(int4) * info.size()));
        TGCN_HIP_CHECK(hipMemcpyAsync(d_info.p, info.data(), sizeof(int4) * info.size(), hipMemcpyHostToDevice, stream));
        TGCN_HIP_CHECK(hipStreamSynchronize(stream));      // `info` leaves scope
        b.bytes += d_info.bytes;
        b.row_info = static_cast<int4 *>(d_info.release());
    }

    // column-block cut positions of the long rows (binary searches on the device)
    const int32_t n_long = static_cast<int32_t>(long_rows.size());
    int32_t n_cb = 0;
    std::vector<int32_t> cuts;
    if (n_long > 0 && kn.col_block > 0) {
        const int64_t nb = (b.n_cols + kn.col_block - 1) / kn.col_block;
        if (nb > 1 && nb <= 8192 && nb * int64_t(n_long) < (int64_t(1) << 28)) n_cb = static_cast<int32_t>(nb);
    }
    if (n_cb > 0) {
        DevBuf d_rows, d_cuts;
        const size_t n_cut = static_cast<size_t>(n_long) * (n_cb + 1);
        TGCN_CHECK(d_rows.alloc(sizeof(int32_t) * n_long));
        TGCN_CHECK(d_cuts.alloc(sizeof(int32_t) * n_cut));
        TGCN_HIP_CHECK(hipMemcpyAsync(d_rows.p, long_rows.data(), sizeof(int32_t) * n_long,
                                      hipMemcpyHostToDevice, stream));
        k_block_cuts<<<grid_for(static_cast<int64_t>(n_cut)), kThreads, 0, stream>>>(
            d_rows.as<int32_t>(), n_long, b.rowptr, b.cv, kn.col_block, n_cb, d_cuts.as<int32_t>());
        TGCN_HIP_CHECK(hipGetLastError());
        cuts.resize(n_cut);
        TGCN_HIP_CHECK(hipMemcpyAsync(cuts.data(), d_cuts.p, sizeof(int32_t) * n_cut,
                                      hipMemcpyDeviceToHost, stream));
        TGCN_HIP_CHECK(hipStreamSynchronize(stream));
    }

    // The dense hot block (spmm.hip: k_spmm_hot).  The K longest rows of a word-document operator are
    // nearly dense (the most frequent words occur in most documents): gathering X[col] once per
    // entry re-reads every document row once per hot word.  When the K = 32 longest rows together
    // hold >= hot_ratio * n_cols entries they are taken out of the gather partition and computed as
    // a dense [32 x n_cols] x [n_cols x F] product that streams X exactly once (fp32 MFMA).
    std::vector<int32_t> hot_of_long(static_cast<size_t>(n_long), -1);   // long-row index -> hot index
    std::vector<int32_t> hot_rows;
    if (kn.hot_rows > 0 && n_long > 0 && b.n_cols >= 4096) {
        std::vector<int32_t> idx(static_cast<size_t>(n_long));
        for (int32_t i = 0; i < n_long; ++i) idx[i] = i;
        const size_t k = std::min<size_t>(kHotRows, idx.size());
        auto deg = [&](int32_t i) { return rp[long_rows[i] + 1] - rp[long_rows[i]]; };
        std::partial_sort(idx.begin(), idx.begin() + k, idx.end(), [&](int32_t x, int32_t y) {
            return deg(x) != deg(y) ? deg(x) > deg(y) : x < y;
        });
        int64_t sum = 0;
        for (size_t q = 0; q < k; ++q) sum += deg(idx[q]);
        if (static_cast<double>(sum) >= kn.hot_ratio * static_cast<double>(b.n_cols)) {
            for (size_t q = 0; q < k; ++q) {
                hot_of_long[idx[q]] = static_cast<int32_t>(q);
                hot_rows.push_back(long_rows[idx[q]]);
            }
        }
    }
    int32_t n_hot = static_cast<int32_t>(hot_rows.size());
    if (n_hot > 0) {
        // the value block is an optimisation: when its memory cannot be had (n_cols x 128 bytes), the
        // plan is built without it rather than failing
        const int64_t cpw0 = std::max<int64_t>(64, ((b.n_cols + 256 * 8 - 1) / (256 * 8) + 7) & ~int64_t(7));
        const size_t want = sizeof(float) * static_cast<size_t>((b.n_cols + cpw0 * 8 - 1) / (cpw0 * 8)) *
                            static_cast<size_t>(cpw0 * 8) * kHotRows;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || want + (size_t(1) << 30) > free_b) {
            (void)hipGetLastError();
            n_hot = 0;
            hot_rows.clear();
            std::fill(hot_of_long.begin(), hot_of_long.end(), -1);
        }
    }
    if (n_hot > 0) {
        // every wave of the hot kernel streams `cpw` consecutive columns, 8 waves per
        // workgroup, at most 256 workgroups; the value block is padded to the full grid with zeros
        // cpw is a multiple of 8 = one load stage of the kernel (4 column pairs), so a stage never
        // reaches into the next wave's columns
        int64_t cpw = (b.n_cols + 256 * 8 - 1) / (256 * 8);
        cpw = std::max<int64_t>(64, (cpw + 7) & ~int64_t(7));
        const int64_t per_wg = cpw * 8;
        b.hot_parts = static_cast<int32_t>((b.n_cols + per_wg - 1) / per_wg);
        b.hot_cpw = static_cast<int32_t>(cpw);
        b.n_hot = n_hot;
        DevBuf d_vals, d_hot;
        const size_t vbytes = sizeof(float) * static_cast<size_t>(b.hot_parts) * per_wg * kHotRows;
        TGCN_CHECK(d_vals.alloc(vbytes));
        TGCN_CHECK(d_hot.alloc(sizeof(int32_t) * kHotRows));
        TGCN_HIP_CHECK(hipMemsetAsync(d_vals.p, 0, vbytes, stream));
        TGCN_HIP_CHECK(hipMemcpyAsync(d_hot.p, hot_rows.data(), sizeof(int32_t) * n_hot,
                                      hipMemcpyHostToDevice, stream));
        int32_t max_deg = 0;
        for (int32_t r : hot_rows) max_deg = std::max(max_deg, rp[r + 1] - rp[r]);
        dim3 grid(static_cast<unsigned>(std::min<int64_t>((max_deg + kThreads - 1) / kThreads, 4096)), n_hot);
        k_hot_fill<<<grid, kThreads, 0, stream>>>(d_hot.as<int32_t>(), b.rowptr, b.cv, d_vals.as<float>());
        TGCN_HIP_CHECK(hipGetLastError());
        TGCN_HIP_CHECK(hipStreamSynchronize(stream));   // hot_rows (host) is read by the copy above
        b.bytes += d_vals.bytes;
        b.hot_vals = static_cast<float *>(d_vals.release());
    }

    // pass 2: segments of the long rows, launch order, fix list.  Built once for the complete operator
    // and, when there is a hot block, once more without the hot rows (the list the float4 kernels use
    // next to k_spmm_hot; the complete list then only serves the scalar fallback kernel).
    struct Seg {
        WorkItem it;
        int32_t key;
    };
    struct Lists {
        std::vector<WorkItem> items;
        std::vector<FixEntry> fix;
        int64_t slots = 0;
        int32_t hot_slot_base = 0;
    };
    bool overflow = false;
    auto make_lists = [&](bool skip_hot, Lists &out) {
        std::vector<Seg> segs;
        std::vector<FixEntry> &fix = out.fix;
        fix.reserve(n_long);
        int64_t slots = 0;
        auto emit_piece = [&](int32_t r, int32_t nb, int32_t ne, int32_t key) {
            const int32_t d = ne - nb;
            const int32_t nseg = (d + T - 1) / T;
            const int32_t seglen = (d + nseg - 1) / nseg;
            for (int32_t s = 0; s < nseg; ++s) {
                const int32_t sb = std::min(nb + s * seglen, ne);
                const int32_t se = std::min(sb + seglen, ne);
                segs.push_back({{r, -static_cast<int32_t>(slots) - 1, sb, se}, key});
                ++slots;
            }
        };
        for (int32_t i = 0; i < n_long; ++i) {
            if (skip_hot && hot_of_long[i] >= 0) continue;
            const int32_t r = long_rows[i];
            const int64_t slot_begin = slots;
            if (n_cb > 0) {
                const int32_t *c = cuts.data() + static_cast<size_t>(i) * (n_cb + 1);
                int32_t j0 = 0;
                for (int32_t j = 1; j <= n_cb; ++j) {
                    const bool last = j == n_cb;
                    if (c[j] - c[j0] >= kn.min_piece || (last && c[j] > c[j0])) {
                        // a short tail joins the previous piece instead of becoming a tiny one
                        if (last && c[j] - c[j0] < kn.min_piece && !segs.empty() && segs.back().it.row_begin == r &&
                            slots > slot_begin && (segs.back().it.nnz_end - segs.back().it.nnz_begin) + (c[j] - c[j0]) <= T) {
                            segs.back().it.nnz_end = c[j];
                        } else {
                            emit_piece(r, c[j0], c[j], j0);
                        }
                        j0 = j;
                    }
                }
            } else {
                emit_piece(r, rp[r], rp[r + 1], 0);
            }
            fix.push_back({r, static_cast<int32_t>(slot_begin), static_cast<int32_t>(slots - slot_begin), 0});
        }
        if (skip_hot) {
            // hot row k: one partial per workgroup of k_spmm_hot, carry rows [base + k P, base + (k+1) P)
            out.hot_slot_base = static_cast<int32_t>(slots);
            for (int32_t k = 0; k < n_hot; ++k)
                fix.push_back({hot_rows[k], static_cast<int32_t>(slots + int64_t(k) * b.hot_parts), b.hot_parts, 0});
            slots += int64_t(n_hot) * b.hot_parts;
        }
        if (slots > INT32_MAX || blocks.size() + segs.size() > size_t(INT32_MAX)) overflow = true;
        out.slots = slots;
        std::stable_sort(segs.begin(), segs.end(), [](const Seg &x, const Seg &y) { return x.key < y.key; });

        // launch order
        std::vector<WorkItem> &items = out.items;
        items.reserve(blocks.size() + segs.size() + 64);
        {
            // segments in column-block order, the row blocks spread evenly between them
            size_t si = 0, bi = 0;
            const size_t ns = segs.size(), nb = blocks.size();
            while (si < ns || bi < nb) {
                if (bi >= nb || (si < ns && si * nb <= bi * ns))
                    items.push_back(segs[si++].it);
                else
                    items.push_back(blocks[bi++]);
            }
        }
    };

    auto upload = [&](const Lists &l, WorkItem *&d_items_out, int32_t &n_items, FixEntry *&d_fix_out,
                      int32_t &n_fix, int32_t &n_slots) -> int {
        n_items = static_cast<int32_t>(l.items.size());
        n_fix = static_cast<int32_t>(l.fix.size());
        n_slots = static_cast<int32_t>(l.slots);
        DevBuf d_items, d_fix;
        TGCN_CHECK(d_items.alloc(sizeof(WorkItem) * l.items.size()));
        TGCN_CHECK(d_fix.alloc(sizeof(FixEntry) * l.fix.size()));
        if (!l.items.empty())
            TGCN_HIP_CHECK(hipMemcpyAsync(d_items.p, l.items.data(), sizeof(WorkItem) * l.items.size(),
                                          hipMemcpyHostToDevice, stream));
        if (!l.fix.empty())
            TGCN_HIP_CHECK(hipMemcpyAsync(d_fix.p, l.fix.data(), sizeof(FixEntry) * l.fix.size(),
                                          hipMemcpyHostToDevice, stream));
        TGCN_HIP_CHECK(hipStreamSynchronize(stream));  // host vectors die on return
        b.bytes += d_items.bytes + d_fix.bytes;
        d_items_out = static_cast<WorkItem *>(d_items.release());
        d_fix_out = static_cast<FixEntry *>(d_fix.release());
        return TGCN_OK;
    };

    Lists main_lists;
    make_lists(n_hot > 0, main_lists);
    if (!overflow) {
        b.hot_slot_base = main_lists.hot_slot_base;
        TGCN_CHECK(upload(main_lists, b.items, b.n_items, b.fix, b.n_fix, b.n_segments));
    }
    if (n_hot > 0 && !overflow) {
        Lists all;
        make_lists(false, all);
        if (!overflow) TGCN_CHECK(upload(all, b.items_all, b.n_items_all, b.fix_all, b.n_fix_all, b.n_segments_all));
    }
    if (overflow) {
        set_error("work partition exceeds int32 limits");
        return TGCN_E_RANGE;
    }
    return TGCN_OK;
}

namespace {

// Builds the operator with entries M[dst, src] (n_rows x n_cols; square with N = n_rows = n_cols
// whenever add_loops / normalize are set).  `with_transpose` = 0 skips M^T entirely.
int plan_create_impl(int64_t n_rows, int64_t n_cols, int64_t E, const int64_t *src, int64_t ss,
                     const int64_t *dst, int64_t ds, const float *w, int add_loops, int normalize,
                     int with_transpose, int64_t row_begin, int64_t row_end, hipStream_t stream,
                     tgcn_plan &plan) {
    const int64_t N = n_rows;  // the square case below reads naturally with N
    const int64_t total = E + (add_loops ? N : 0);
    if (n_rows >= (int64_t(1) << 31) - 1 || n_cols >= (int64_t(1) << 31) - 1 ||
        total >= (int64_t(1) << 31) - 1) {
        set_error("n_rows=%lld / n_cols=%lld / n_edges=%lld exceed the int32 index range of this build",
                  (long long)n_rows, (long long)n_cols, (long long)E);
        return TGCN_E_RANGE;
    }

    DevBuf loop_eid, flags, keys_a, keys_b, vals_a, vals_b, rowptr, rowptr_t, dis, cv, cv_t;
    TGCN_CHECK(loop_eid.alloc(sizeof(unsigned long long) * (add_loops ? N : 1)));
    TGCN_CHECK(flags.alloc(sizeof(unsigned int) * 4));
    TGCN_CHECK(dis.alloc(sizeof(float) * N));
    TGCN_HIP_CHECK(hipMemsetAsync(loop_eid.p, 0, loop_eid.bytes ? loop_eid.bytes : 16, stream));
    TGCN_HIP_CHECK(hipMemsetAsync(flags.p, 0, sizeof(unsigned int) * 4, stream));

    if (E > 0) {
        k_scan_edges<<<grid_for(E, kThreads, 4096), kThreads, 0, stream>>>(
            E, src, ss, dst, ds, n_rows, n_cols, add_loops, loop_eid.as<unsigned long long>(),
            flags.as<unsigned int>());
        TGCN_HIP_CHECK(hipGetLastError());
    }
    unsigned int h_flags[4] = {0, 0, 0, 0};
    TGCN_HIP_CHECK(hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, stream));
    TGCN_HIP_CHECK(hipStreamSynchronize(stream));
    if (h_flags[0]) {
        set_error("an index lies outside the operator: rows [0, %lld), columns [0, %lld)",
                  (long long)n_rows, (long long)n_cols);
        return TGCN_E_RANGE;
    }
    const int64_t nnz = total - (add_loops ? int64_t(h_flags[1]) : 0);

    // the degree factors come from the routine tgcn_gcn_norm uses too (its scratch is gone before the sort's is taken)
    if (normalize && N > 0)
        TGCN_CHECK(degree_factors(N, n_cols, E, src, ss, dst, ds, w, add_loops, normalize == 2,
                                  loop_eid.as<unsigned long long>(), dis.as<float>(), nullptr, stream));
    TGCN_CHECK(keys_a.alloc(sizeof(uint64_t) * total));
    TGCN_CHECK(keys_b.alloc(sizeof(uint64_t) * total));
    TGCN_CHECK(vals_a.alloc(sizeof(float) * total));
    TGCN_CHECK(vals_b.alloc(sizeof(float) * total));
    TGCN_CHECK(rowptr.alloc(sizeof(int32_t) * (N + 1)));
    TGCN_CHECK(rowptr_t.alloc(sizeof(int32_t) * (n_cols + 1)));

    if (total > 0) {
        k_make_keys<<<grid_for(total, kThreads, 8192), kThreads, 0, stream>>>(
            E, src, ss, dst, ds, w, n_rows, n_cols, add_loops, static_cast<float>(add_loops),
            loop_eid.as<unsigned long long>(), keys_a.as<uint64_t>(), vals_a.as<float>());
        TGCN_HIP_CHECK(hipGetLastError());
    }
    unsigned node_bits = 1;
    while ((int64_t(1) << node_bits) <= N) ++node_bits;  // the drop key N << 32 must sort too
    const unsigned end_bit = 32 + node_bits;
    unsigned col_bits = 1;
    while ((int64_t(1) << col_bits) <= n_cols) ++col_bits;
    const unsigned end_bit_t = 32 + col_bits;
    // stable: duplicate (dst, src) pairs keep their edge order, so sums are deterministic
    TGCN_CHECK(sort_pairs(keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals_a.as<float>(),
                          vals_b.as<float>(), total, end_bit, stream));

    k_rowptr<<<grid_for(N + 1), kThreads, 0, stream>>>(keys_b.as<uint64_t>(), nnz, N,
                                                       rowptr.as<int32_t>());
    TGCN_HIP_CHECK(hipGetLastError());
    TGCN_CHECK(cv.alloc(sizeof(int2) * nnz));
    TGCN_CHECK(cv_t.alloc(sizeof(int2) * nnz));
    if (nnz > 0) {
        k_finalize<<<grid_for(nnz, kThreads, 8192), kThreads, 0, stream>>>(
            keys_b.as<uint64_t>(), vals_b.as<float>(), nnz, dis.as<float>(), normalize,
            cv.as<int2>(), keys_a.as<uint64_t>(), vals_a.as<float>());
        TGCN_HIP_CHECK(hipGetLastError());
    }
    plan.symmetric = false;
    plan.has_transpose = with_transpose != 0;
    if (with_transpose) {
        // transposed operator: sort by (src, dst)
        TGCN_CHECK(sort_pairs(keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), vals_a.as<float>(),
                              vals_b.as<float>(), nnz, end_bit_t, stream));
        k_rowptr<<<grid_for(n_cols + 1), kThreads, 0, stream>>>(keys_b.as<uint64_t>(), nnz, n_cols,
                                                                rowptr_t.as<int32_t>());
        TGCN_HIP_CHECK(hipGetLastError());
        if (nnz > 0) {
            k_unpack<<<grid_for(nnz, kThreads, 8192), kThreads, 0, stream>>>(
                keys_b.as<uint64_t>(), vals_b.as<float>(), nnz, cv_t.as<int2>());
            TGCN_HIP_CHECK(hipGetLastError());
        }
        if (n_rows == n_cols) {
            TGCN_HIP_CHECK(hipMemsetAsync(flags.p, 0, sizeof(unsigned int) * 4, stream));
            k_compare<<<grid_for(std::max<int64_t>(nnz, N + 1), kThreads, 4096), kThreads, 0,
                        stream>>>(rowptr.as<int32_t>(), rowptr_t.as<int32_t>(), N + 1, cv.as<int2>(),
                                  cv_t.as<int2>(), nnz, flags.as<unsigned int>());
            TGCN_HIP_CHECK(hipGetLastError());
            TGCN_HIP_CHECK(
                hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, stream));
            TGCN_HIP_CHECK(hipStreamSynchronize(stream));
            plan.symmetric = h_flags[0] == 0;
        }
    }

    // sort scratch is no longer needed: release it before the blocks are copied out
    (void)hipFree(keys_a.release());
    (void)hipFree(keys_b.release());
    (void)hipFree(vals_a.release());
    (void)hipFree(vals_b.release());

    const bool whole = row_begin == 0 && row_end == N;
    if (whole) {
        plan.fwd.n_rows = n_rows;
        plan.fwd.n_cols = n_cols;
        plan.fwd.nnz = nnz;
        plan.fwd.bytes = rowptr.bytes + cv.bytes;
        plan.fwd.rowptr = static_cast<int32_t *>(rowptr.release());
        plan.fwd.cv = static_cast<int2 *>(cv.release());
    } else {
        TGCN_CHECK(take_block(rowptr.as<int32_t>(), cv.as<int2>(), N, row_begin, row_end, plan.fwd,
                              stream));
    }
    TGCN_CHECK(build_items(plan.fwd, item_weight_for(plan.fwd), stream));
    if (with_transpose && !plan.symmetric) {
        if (whole) {
            plan.bwd.n_rows = n_cols;
            plan.bwd.n_cols = n_rows;
            plan.bwd.nnz = nnz;
            plan.bwd.bytes = rowptr_t.bytes + cv_t.bytes;
            plan.bwd.rowptr = static_cast<int32_t *>(rowptr_t.release());
            plan.bwd.cv = static_cast<int2 *>(cv_t.release());
        } else {
            TGCN_CHECK(take_block(rowptr_t.as<int32_t>(), cv_t.as<int2>(), N, row_begin, row_end,
                                  plan.bwd, stream));
        }
        TGCN_CHECK(build_items(plan.bwd, item_weight_for(plan.bwd), stream));
    }
    return TGCN_OK;
}

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    int enter(int device) {
        TGCN_HIP_CHECK(hipGetDevice(&prev));
        if (prev != device) {
            TGCN_HIP_CHECK(hipSetDevice(device));
            switched = true;
        }
        return TGCN_OK;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

}  // namespace
}  // namespace tgcn

extern "C" {

int tgcn_plan_create(int64_t n_nodes, int64_t n_edges, const int64_t *src, int64_t src_stride,
                     const int64_t *dst, int64_t dst_stride, const float *w, int add_self_loops,
                     int normalize, int64_t row_begin, int64_t row_end, int device,
                     tgcn_stream stream, tgcn_plan **out) {
    using namespace tgcn;
    if (!out) {
        set_error("tgcn_plan_create: out is NULL");
        return TGCN_E_INVALID;
    }
    *out = nullptr;
    if (n_nodes <= 0 || n_edges < 0) {
        set_error("tgcn_plan_create: n_nodes=%lld must be > 0 and n_edges=%lld >= 0",
                  (long long)n_nodes, (long long)n_edges);
        return TGCN_E_INVALID;
    }
    if (n_edges > 0 && (!src || !dst || src_stride <= 0 || dst_stride <= 0)) {
        set_error("tgcn_plan_create: src/dst must be non-NULL with positive strides");
        return TGCN_E_INVALID;
    }
    if (normalize < 0 || normalize > 2) {
        set_error("tgcn_plan_create: normalize=%d must be TGCN_NORM_OFF, TGCN_NORM_ACCURATE or TGCN_NORM_REFERENCE", normalize);
        return TGCN_E_INVALID;
    }
    if (row_begin < 0 || row_end < row_begin || row_end > n_nodes) {
        set_error("tgcn_plan_create: row range [%lld, %lld) outside [0, %lld]", (long long)row_begin,
                  (long long)row_end, (long long)n_nodes);
        return TGCN_E_INVALID;
    }
    DeviceGuard guard;
    TGCN_CHECK(guard.enter(device));
    tgcn_plan *plan = new (std::nothrow) tgcn_plan();
    if (!plan) {
        set_error("tgcn_plan_create: host allocation failed");
        return TGCN_E_NOMEM;
    }
    plan->device = device;
    plan->n_nodes = n_nodes;
    plan->row_begin = row_begin;
    plan->row_end = row_end;
    // PyG adds the loops inside gcn_norm, so GCNConv(normalize=False) never sees them
    // add_self_loops doubles as the fill weight: 1 -> 1.0, 2 -> 2.0 (GCNConv(improved=True))
    const int st = plan_create_impl(n_nodes, n_nodes, n_edges, src, src_stride, dst, dst_stride, w,
                                    normalize != 0 ? std::max(0, std::min(add_self_loops, 2)) : 0, normalize, 1,
                                    row_begin, row_end, static_cast<hipStream_t>(stream), *plan);
    if (st != TGCN_OK) {
        free_block(plan->fwd);
        free_block(plan->bwd);
        delete plan;
        return st;
    }
    *out = plan;
    return TGCN_OK;
}

int tgcn_plan_create_coo(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t *row,
                         const int64_t *col, const float *val, int with_transpose, int device,
                         tgcn_stream stream, tgcn_plan **out) {
    using namespace tgcn;
    if (!out) {
        set_error("tgcn_plan_create_coo: out is NULL");
        return TGCN_E_INVALID;
    }
    *out = nullptr;
    if (n_rows <= 0 || n_cols <= 0 || nnz < 0 || (nnz > 0 && (!row || !col))) {
        set_error("tgcn_plan_create_coo: need n_rows, n_cols > 0, nnz >= 0 and non-NULL row/col "
                  "(n_rows=%lld n_cols=%lld nnz=%lld)", (long long)n_rows, (long long)n_cols,
                  (long long)nnz);
        return TGCN_E_INVALID;
    }
    DeviceGuard guard;
    TGCN_CHECK(guard.enter(device));
    tgcn_plan *plan = new (std::nothrow) tgcn_plan();
    if (!plan) {
        set_error("tgcn_plan_create_coo: host allocation failed");
        return TGCN_E_NOMEM;
    }
    plan->device = device;
    plan->n_nodes = n_cols;
    plan->row_begin = 0;
    plan->row_end = n_rows;
    const int st = plan_create_impl(n_rows, n_cols, nnz, col, 1, row, 1, val, 0, 0, with_transpose,
                                    0, n_rows, static_cast<hipStream_t>(stream), *plan);
    if (st != TGCN_OK) {
        free_block(plan->fwd);
        free_block(plan->bwd);
        delete plan;
        return st;
    }
    *out = plan;
    return TGCN_OK;
}

int tgcn_gcn_norm(int64_t n_nodes, int64_t n_edges, const int64_t *src, int64_t src_stride,
                  const int64_t *dst, int64_t dst_stride, const float *w, int add_self_loops, int degree_sum,
                  float *dis, float *loop_w, int device, tgcn_stream stream_) {
    using namespace tgcn;
    if (n_nodes <= 0 || n_edges < 0 || !dis) {
        set_error("tgcn_gcn_norm: need n_nodes > 0, n_edges >= 0 and a non-NULL dis");
        return TGCN_E_INVALID;
    }
    if (n_edges > 0 && (!src || !dst || src_stride <= 0 || dst_stride <= 0)) {
        set_error("tgcn_gcn_norm: src/dst must be non-NULL with positive strides");
        return TGCN_E_INVALID;
    }
    if (degree_sum != TGCN_DEGREE_ACCURATE && degree_sum != TGCN_DEGREE_REFERENCE) {
        set_error("tgcn_gcn_norm: degree_sum=%d must be TGCN_DEGREE_ACCURATE or TGCN_DEGREE_REFERENCE", degree_sum);
        return TGCN_E_INVALID;
    }
    if (n_nodes >= (int64_t(1) << 31) - 1) {
        set_error("tgcn_gcn_norm: n_nodes=%lld exceeds the int32 index range of this build", (long long)n_nodes);
        return TGCN_E_RANGE;
    }
    DeviceGuard guard;
    TGCN_CHECK(guard.enter(device));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int64_t N = n_nodes, E = n_edges;
    const int add_loops = std::max(0, std::min(add_self_loops, 2));
    DevBuf loop_eid, flags;
    TGCN_CHECK(loop_eid.alloc(sizeof(unsigned long long) * (add_loops ? N : 1)));
    TGCN_CHECK(flags.alloc(sizeof(unsigned int) * 4));
    TGCN_HIP_CHECK(hipMemsetAsync(loop_eid.p, 0, loop_eid.bytes ? loop_eid.bytes : 16, stream));
    TGCN_HIP_CHECK(hipMemsetAsync(flags.p, 0, sizeof(unsigned int) * 4, stream));
    if (E > 0) {
        k_scan_edges<<<grid_for(E, kThreads, 4096), kThreads, 0, stream>>>(
            E, src, src_stride, dst, dst_stride, N, N, add_loops, loop_eid.as<unsigned long long>(),
            flags.as<unsigned int>());
        TGCN_HIP_CHECK(hipGetLastError());
    }
    unsigned int h_flags[4] = {0, 0, 0, 0};
    TGCN_HIP_CHECK(hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, stream));
    TGCN_HIP_CHECK(hipStreamSynchronize(stream));
    if (h_flags[0]) {
        set_error("an index lies outside [0, %lld)", (long long)N);
        return TGCN_E_RANGE;
    }
    return degree_factors(N, N, E, src, src_stride, dst, dst_stride, w, add_loops, degree_sum == TGCN_DEGREE_REFERENCE,
                          loop_eid.as<unsigned long long>(), dis, loop_w, stream);
}

int tgcn_plan_destroy(tgcn_plan *plan) {
    using namespace tgcn;
    if (!plan) return TGCN_OK;
    DeviceGuard guard;
    TGCN_CHECK(guard.enter(plan->device));
    free_block(plan->fwd);
    free_block(plan->bwd);
    delete plan;
    return TGCN_OK;
}

int tgcn_plan_query(const tgcn_plan *plan, int what, int64_t *out) {
    using namespace tgcn;
    if (!plan || !out) {
        set_error("tgcn_plan_query: NULL argument");
        return TGCN_E_INVALID;
    }
    const CsrBlock &f = plan->fwd;
    const CsrBlock &t = plan->symmetric ? plan->fwd : plan->bwd;
    switch (what) {
        case TGCN_Q_N_NODES: *out = plan->n_nodes; break;
        case TGCN_Q_N_ROWS: *out = plan->row_end - plan->row_begin; break;
        case TGCN_Q_NNZ: *out = f.nnz; break;
        case TGCN_Q_NNZ_T: *out = t.nnz; break;
        case TGCN_Q_SYMMETRIC: *out = plan->symmetric ? 1 : 0; break;
        case TGCN_Q_ITEMS: *out = f.n_items; break;
        case TGCN_Q_ITEMS_T: *out = t.n_items; break;
        case TGCN_Q_LONG_ROWS: *out = f.n_fix; break;
        case TGCN_Q_LONG_ROWS_T: *out = t.n_fix; break;
        case TGCN_Q_SEGMENTS: *out = f.n_segments; break;
        case TGCN_Q_SEGMENTS_T: *out = t.n_segments; break;
        case TGCN_Q_DEVICE_BYTES:
            *out = static_cast<int64_t>(f.bytes + (plan->symmetric ? 0 : plan->bwd.bytes));
            break;
        case TGCN_Q_ROW_BEGIN: *out = plan->row_begin; break;
        case TGCN_Q_HAS_TRANSPOSE: *out = (plan->has_transpose || plan->symmetric) ? 1 : 0; break;
        case TGCN_Q_N_ROWS_T: *out = t.n_rows; break;
        case TGCN_Q_HOT_ROWS: *out = f.n_hot; break;
        case TGCN_Q_HOT_ROWS_T: *out = t.n_hot; break;
        default:
            set_error("tgcn_plan_query: unknown selector %d", what);
            return TGCN_E_INVALID;
    }
    return TGCN_OK;
}

namespace {
__global__ void k_export(const int2 *__restrict__ cv, int64_t nnz, int32_t *col, float *val) {
    const int64_t stride = int64_t(gridDim.x) * blockDim.x;
    for (int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; j < nnz; j += stride) {
        const int2 p = cv[j];
        if (col) col[j] = p.x;
        if (val) val[j] = __int_as_float(p.y);
    }
}
}  // namespace

int tgcn_plan_export(const tgcn_plan *plan, int transpose, int32_t *rowptr, int32_t *col, float *val,
                     tgcn_stream stream) {
    using namespace tgcn;
    if (!plan) {
        set_error("tgcn_plan_export: plan is NULL");
        return TGCN_E_INVALID;
    }
    DeviceGuard guard;
    TGCN_CHECK(guard.enter(plan->device));
    if (transpose && !plan->symmetric && !plan->has_transpose) {
        set_error("tgcn_plan_export: this plan was built without its transpose");
        return TGCN_E_INVALID;
    }
    const CsrBlock &b = (transpose && !plan->symmetric) ? plan->bwd : plan->fwd;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (rowptr)
        TGCN_HIP_CHECK(hipMemcpyAsync(rowptr, b.rowptr, sizeof(int32_t) * (b.n_rows + 1),
                                      hipMemcpyDeviceToDevice, s));
    if ((col || val) && b.nnz > 0) {
        k_export<<<grid_for(b.nnz, kThreads, 8192), kThreads, 0, s>>>(b.cv, b.nnz, col, val);
        TGCN_HIP_CHECK(hipGetLastError());
    }
    return TGCN_OK;
}

}  // extern "C"
