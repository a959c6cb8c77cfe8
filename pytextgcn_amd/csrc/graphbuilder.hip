// Word-word PMI edges on the GPU: replaces the reference's only native component, the Cython module
// textgcn/lib/clib/graphbuilder.pyx (`compute_word_word_edges`, :23-68; `sliding_window`, :71-115;
// `edges_from_counts`, :118-211), called from Text2GraphTransformer at text2graph.py:156-160.
//
// Integer / HBM-atomic bound work, written for 288 GB of HBM rather than for MFMA:
//   * the co-occurrence counts c_ij stay a DENSE packed upper triangle (uint32, V(V+1)/2 entries,
//     20 GB at V = 100 000) in HBM, updated with no-return integer atomics;
//   * the reference enumerates every window and every pair inside it, O(D L w^2).  A pair of
//     positions (k, l), k <= l < k + w, lies in the windows starting at max(0, l-w+1) .. min(k, J_d)
//     (J_d = last window start of document d), so one thread per position k adds that COUNT once
//     per l: O(D L w) atomics, identical integers;
//   * PMI uses the reference's exact float sequence (float32 p_i, p_ij and ratio, libm double log
//     rounded to float32, threshold 1e-10), so the edge set and the weights are bit-identical;
//   * edges come out in the reference's order ((i,j),(j,i) interleaved, upper triangle row-major)
//     through a per-row count, a scan over rows and an ordered ballot compaction.
// Extensions beyond the reference's defined behaviour: 64-bit packed indices (its uint32 index math
// wraps for V > 65 535, graphbuilder.pyx:224-259), window > seq_len = one window per document (its
// unsigned `seq_len - window_size + 1` wraps, :92), owned outputs with a destroy call (it leaks its
// malloc'd arrays, :65-66).
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <new>
#include <vector>

#include <rocprim/device/device_scan.hpp>

#include "common.h"

struct tgcn_wwedges {
    int device = 0;
    int64_t n_vocab = 0;
    int64_t n_windows = 0;
    int64_t n_edges = 0;
    uint32_t *cij = nullptr;   // packed upper triangle incl. diagonal
    int32_t *coo = nullptr;    // [n_edges][2]
    float *weights = nullptr;  // [n_edges]
};

namespace tgcn {
namespace {

__host__ __device__ inline int64_t sym_diag_idx(int64_t row, int64_t col, int64_t n) {
    if (row < col) {
        const int64_t t = row;
        row = col;
        col = t;
    }
    return col * n + row - (col + 1) * col / 2;
}

// One wavefront per document: J_d = last window start.  Window j (j >= 1) exists while its last
// element X[j + w - 1] is not padding (graphbuilder.pyx:96-98); window 0 always exists.
__global__ void k_last_window(const int32_t *__restrict__ X, int64_t n_docs, int64_t L, int64_t w,
                              int32_t *__restrict__ last_start, unsigned long long *n_windows) {
    const int lane = threadIdx.x & 63;
    const int64_t d = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (d >= n_docs) return;
    const int64_t n_starts = L >= w ? L - w + 1 : 1;
    const int32_t *x = X + d * L;
    int64_t first_bad = n_starts;  // first j >= 1 whose window reaches the padding
    for (int64_t j0 = 1; j0 < n_starts; j0 += 64) {
        const int64_t j = j0 + lane;
        const bool bad = j < n_starts && x[j + w - 1] == -1;
        const unsigned long long m = __ballot(bad);
        if (m) {
            first_bad = j0 + __ffsll(static_cast<long long>(m)) - 1;
            break;
        }
    }
    if (lane == 0) {
        last_start[d] = static_cast<int32_t>(first_bad - 1);
        atomicAdd(n_windows, static_cast<unsigned long long>(first_bad));
    }
}

// One thread per token position (d, k): for every l in [k, k + w) up to the first padding value,
// add the number of windows that contain both positions (see header).
__global__ void k_pair_counts(const int32_t *__restrict__ X, int64_t n_docs, int64_t L, int64_t w,
                              int64_t V, const int32_t *__restrict__ last_start,
                              uint32_t *__restrict__ cij) {
    const int64_t idx = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= n_docs * L) return;
    const int64_t d = idx / L, k = idx % L;
    const int32_t *x = X + d * L;
    const int32_t a = x[k];
    if (a == -1) return;
    const int64_t jd = last_start[d];
    const int64_t l_end = std::min(L, k + w);
    const int64_t hi = std::min(k, jd);
    for (int64_t l = k; l < l_end; ++l) {
        const int32_t b = x[l];
        if (b == -1) break;                          // graphbuilder.pyx:106-111
        const int64_t lo = std::max<int64_t>(0, l - w + 1);
        if (hi >= lo)
            atomicAdd(&cij[sym_diag_idx(a, b, V)], static_cast<uint32_t>(hi - lo + 1));
    }
}

// ---- hot-pair privatisation ------------------------------------------------------------------------------
// Word frequencies follow a Zipf law: the pairs among the most frequent words -- above all the diagonal
// entries (a, a), one update per token -- hit a handful of addresses, and same-address atomics from all
// XCDs serialise in one L2 channel (k_pair_counts alone: 240 M atomics in 64 ms on a 100 k-document corpus).
// So the H most frequent words (token histogram, top-H chosen on the host) get a workgroup-private packed
// triangle of pair counts in LDS; a workgroup walks a contiguous slice of token positions and flushes its
// non-zero counters once.  Integer sums: the result is the same whatever the order.
constexpr int kHotWords = 128;
constexpr int kHotTri = kHotWords * (kHotWords + 1) / 2;
constexpr int kHistChunk = 32768;        // vocabulary ids per pass of the token histogram (128 KB of LDS)

__global__ __launch_bounds__(1024) void k_token_hist(const int32_t *__restrict__ X, int64_t n, int64_t v0, int64_t v1,
                                                      uint32_t *__restrict__ hist) {
    extern __shared__ uint32_t h[];
    const int span = static_cast<int>(v1 - v0);
    for (int j = threadIdx.x; j < span; j += blockDim.x) h[j] = 0;
    __syncthreads();
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b = int64_t(blockIdx.x) * per, e = std::min(n, b + per);
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const int32_t a = X[i];
        if (a >= v0 && a < v1) atomicAdd(&h[a - v0], 1u);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < span; j += blockDim.x)
        if (h[j] != 0) atomicAdd(&hist[v0 + j], h[j]);
}

// the kernel above with the pairs of hot words counted in LDS; hot_of[word] = hot index or -1
__global__ __launch_bounds__(256) void k_pair_counts_hot(const int32_t *__restrict__ X, int64_t n_docs, int64_t L, int64_t w,
                                                         int64_t V, const int32_t *__restrict__ last_start,
                                                         const int16_t *__restrict__ hot_of,
                                                         const int32_t *__restrict__ hot_word, uint32_t *__restrict__ cij) {
    __shared__ uint32_t tri[kHotTri];
    for (int j = threadIdx.x; j < kHotTri; j += blockDim.x) tri[j] = 0;
    __syncthreads();
    const int64_t n = n_docs * L;
    const int64_t per = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int64_t b = int64_t(blockIdx.x) * per, e = std::min(n, b + per);
    for (int64_t idx = b + threadIdx.x; idx < e; idx += 256) {
        const int64_t d = idx / L, k = idx % L;
        const int32_t *x = X + d * L;
        const int32_t a = x[k];
        if (a == -1) continue;
        const int ha = hot_of[a];
        const int64_t jd = last_start[d];
        const int64_t l_end = std::min(L, k + w);
        const int64_t hi = std::min(k, jd);
        for (int64_t l = k; l < l_end; ++l) {
            const int32_t bb = x[l];
            if (bb == -1) break;                          // graphbuilder.pyx:106-111
            const int64_t lo = std::max<int64_t>(0, l - w + 1);
            if (hi < lo) continue;
            const uint32_t c = static_cast<uint32_t>(hi - lo + 1);
            const int hb = ha >= 0 ? hot_of[bb] : -1;
            if (hb >= 0)
                atomicAdd(&tri[sym_diag_idx(ha, hb, kHotWords)], c);
            else
                atomicAdd(&cij[sym_diag_idx(a, bb, V)], c);
        }
    }
    __syncthreads();
    // flush: packed index -> (row >= col) of the hot triangle -> the two words
    for (int j = threadIdx.x; j < kHotTri; j += blockDim.x) {
        const uint32_t c = tri[j];
        if (c == 0) continue;
        int col = 0, rem = j;
        while (rem >= kHotWords - col) {                  // column `col` of the packed triangle holds kHotWords - col entries
            rem -= kHotWords - col;
            ++col;
        }
        const int row = col + rem;
        atomicAdd(&cij[sym_diag_idx(hot_word[row], hot_word[col], V)], c);
    }
}

__device__ __forceinline__ bool pmi_edge(const uint32_t *__restrict__ cij, int64_t i, int64_t j,
                                         int64_t V, float nw, float pi, float *pmi_out) {
    const float pj = static_cast<float>(cij[sym_diag_idx(j, j, V)]) / nw;
    const float pij = static_cast<float>(cij[sym_diag_idx(i, j, V)]) / nw;
    if (pij == 0.f || pi == 0.f || pj == 0.f) return false;        // graphbuilder.pyx:157-160
    const float pmi = static_cast<float>(log(static_cast<double>(pij / (pi * pj))));  // :161
    *pmi_out = pmi;
    return pmi > 1e-10f;                                           // :20,162
}

// One wavefront per row i of the upper triangle: number of pairs (i, j > i) that become edges.
__global__ void k_row_edge_count(const uint32_t *__restrict__ cij, int64_t V, float nw,
                                 int64_t *__restrict__ row_count) {
    const int lane = threadIdx.x & 63;
    const int64_t i = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= V) return;
    const float pi = static_cast<float>(cij[sym_diag_idx(i, i, V)]) / nw;
    int64_t n = 0;
    for (int64_t j = i + 1 + lane; j < V; j += 64) {
        float pmi;
        n += pmi_edge(cij, i, j, V, nw, pi, &pmi) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    if (lane == 0) row_count[i] = n;
}

// Same sweep, now writing (i,j),(j,i) at 2 * (row_offset[i] + rank of j among the row's edges).
__global__ void k_emit_edges(const uint32_t *__restrict__ cij, int64_t V, float nw,
                             const int64_t *__restrict__ row_offset, int32_t *__restrict__ coo,
                             float *__restrict__ weights) {
    const int lane = threadIdx.x & 63;
    const int64_t i = int64_t(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= V) return;
    const float pi = static_cast<float>(cij[sym_diag_idx(i, i, V)]) / nw;
    int64_t base = row_offset[i];
    for (int64_t j0 = i + 1; j0 < V; j0 += 64) {
        const int64_t j = j0 + lane;
        float pmi = 0.f;
        const bool e = j < V && pmi_edge(cij, i, j, V, nw, pi, &pmi);
        const unsigned long long m = __ballot(e);
        if (e) {
            const int64_t k = 2 * (base + __popcll(m & ((1ull << lane) - 1ull)));
            coo[2 * k] = static_cast<int32_t>(i);
            coo[2 * k + 1] = static_cast<int32_t>(j);
            weights[k] = pmi;
            coo[2 * k + 2] = static_cast<int32_t>(j);
            coo[2 * k + 3] = static_cast<int32_t>(i);
            weights[k + 1] = pmi;
        }
        base += __popcll(m);
    }
}

struct Guard {
    int prev = -1;
    bool sw = false;
    int enter(int dev) {
        TGCN_HIP_CHECK(hipGetDevice(&prev));
        if (prev != dev) {
            TGCN_HIP_CHECK(hipSetDevice(dev));
            sw = true;
        }
        return TGCN_OK;
    }
    ~Guard() {
        if (sw) (void)hipSetDevice(prev);
    }
};

struct Tmp {
    void *p = nullptr;
    ~Tmp() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t n) {
        hipError_t e = hipMalloc(&p, n ? n : 16);
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu bytes): %s", n, hipGetErrorString(e));
            return TGCN_E_NOMEM;
        }
        return TGCN_OK;
    }
};

int build(tgcn_wwedges &we, const int32_t *X, int64_t D, int64_t L, int64_t V, int64_t w,
          hipStream_t s) {
    const int64_t tri = V * (V + 1) / 2;
    {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, sizeof(uint32_t) * static_cast<size_t>(tri));
        if (e != hipSuccess) {
            set_error("hipMalloc of the %lld-entry count triangle: %s", (long long)tri, hipGetErrorString(e));
            return TGCN_E_NOMEM;
        }
        we.cij = static_cast<uint32_t *>(p);
    }
    TGCN_HIP_CHECK(hipMemsetAsync(we.cij, 0, sizeof(uint32_t) * static_cast<size_t>(tri), s));
    Tmp last, nwin, rows, offs;
    TGCN_CHECK(last.alloc(sizeof(int32_t) * D));
    TGCN_CHECK(nwin.alloc(sizeof(unsigned long long)));
    TGCN_CHECK(rows.alloc(sizeof(int64_t) * (V + 1)));
    TGCN_CHECK(offs.alloc(sizeof(int64_t) * (V + 1)));
    TGCN_HIP_CHECK(hipMemsetAsync(nwin.p, 0, sizeof(unsigned long long), s));
    if (D > 0) {
        k_last_window<<<static_cast<unsigned>((D + 3) / 4), 256, 0, s>>>(
            X, D, L, w, static_cast<int32_t *>(last.p), static_cast<unsigned long long *>(nwin.p));
        TGCN_HIP_CHECK(hipGetLastError());
        const int64_t n = D * L;
        if (V > kHotWords && n >= (int64_t(1) << 16)) {
            // token histogram (vocabulary ids in chunks that fit the LDS), top-H words on the host
            Tmp hist, hot_of_d, hot_word_d;
            TGCN_CHECK(hist.alloc(sizeof(uint32_t) * V));
            TGCN_CHECK(hot_of_d.alloc(sizeof(int16_t) * V));
            TGCN_CHECK(hot_word_d.alloc(sizeof(int32_t) * kHotWords));
            TGCN_HIP_CHECK(hipMemsetAsync(hist.p, 0, sizeof(uint32_t) * V, s));
            for (int64_t v0 = 0; v0 < V; v0 += kHistChunk) {
                const int64_t v1 = std::min<int64_t>(V, v0 + kHistChunk);
                const size_t lds = sizeof(uint32_t) * static_cast<size_t>(v1 - v0);
                TGCN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_token_hist),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
                k_token_hist<<<256, 1024, lds, s>>>(X, n, v0, v1, static_cast<uint32_t *>(hist.p));
                TGCN_HIP_CHECK(hipGetLastError());
            }
            std::vector<uint32_t> h_hist(static_cast<size_t>(V));
            TGCN_HIP_CHECK(hipMemcpyAsync(h_hist.data(), hist.p, sizeof(uint32_t) * V, hipMemcpyDeviceToHost, s));
            TGCN_HIP_CHECK(hipStreamSynchronize(s));
            std::vector<int32_t> order(static_cast<size_t>(V));
            for (int64_t v = 0; v < V; ++v) order[v] = static_cast<int32_t>(v);
            std::partial_sort(order.begin(), order.begin() + kHotWords, order.end(), [&](int32_t x, int32_t y) {
                return h_hist[x] != h_hist[y] ? h_hist[x] > h_hist[y] : x < y;
            });
            std::vector<int16_t> h_hot_of(static_cast<size_t>(V), int16_t(-1));
            std::vector<int32_t> h_hot_word(kHotWords);
            for (int k = 0; k < kHotWords; ++k) {
                h_hot_word[k] = order[k];
                h_hot_of[order[k]] = static_cast<int16_t>(k);
            }
            TGCN_HIP_CHECK(hipMemcpyAsync(hot_of_d.p, h_hot_of.data(), sizeof(int16_t) * V, hipMemcpyHostToDevice, s));
            TGCN_HIP_CHECK(hipMemcpyAsync(hot_word_d.p, h_hot_word.data(), sizeof(int32_t) * kHotWords, hipMemcpyHostToDevice, s));
            const unsigned blocks = static_cast<unsigned>(std::min<int64_t>(2048, (n + 4095) / 4096));
            k_pair_counts_hot<<<blocks, 256, 0, s>>>(X, D, L, w, V, static_cast<int32_t *>(last.p),
                                                     static_cast<int16_t *>(hot_of_d.p),
                                                     static_cast<int32_t *>(hot_word_d.p), we.cij);
            TGCN_HIP_CHECK(hipGetLastError());
            TGCN_HIP_CHECK(hipStreamSynchronize(s));      // the host vectors and Tmp buffers leave scope
        } else {
            k_pair_counts<<<static_cast<unsigned>((n + 255) / 256), 256, 0, s>>>(
                X, D, L, w, V, static_cast<int32_t *>(last.p), we.cij);
            TGCN_HIP_CHECK(hipGetLastError());
        }
    }
    unsigned long long h_nw = 0;
    TGCN_HIP_CHECK(hipMemcpyAsync(&h_nw, nwin.p, sizeof(h_nw), hipMemcpyDeviceToHost, s));
    TGCN_HIP_CHECK(hipStreamSynchronize(s));
    we.n_windows = static_cast<int64_t>(h_nw);
    const float nw = static_cast<float>(we.n_windows);          // graphbuilder.pyx:147 <float>n_windows

    int64_t *row_count = static_cast<int64_t *>(rows.p), *row_off = static_cast<int64_t *>(offs.p);
    TGCN_HIP_CHECK(hipMemsetAsync(row_count, 0, sizeof(int64_t) * (V + 1), s));
    k_row_edge_count<<<static_cast<unsigned>((V + 3) / 4), 256, 0, s>>>(we.cij, V, nw, row_count);
    TGCN_HIP_CHECK(hipGetLastError());
    size_t tmp_bytes = 0;
    TGCN_HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, row_count, row_off, int64_t(0),
                                           static_cast<size_t>(V + 1), rocprim::plus<int64_t>(), s));
    Tmp scan_tmp;
    TGCN_CHECK(scan_tmp.alloc(tmp_bytes));
    TGCN_HIP_CHECK(rocprim::exclusive_scan(scan_tmp.p, tmp_bytes, row_count, row_off, int64_t(0),
                                           static_cast<size_t>(V + 1), rocprim::plus<int64_t>(), s));
    int64_t pairs = 0;
    TGCN_HIP_CHECK(hipMemcpyAsync(&pairs, row_off + V, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    TGCN_HIP_CHECK(hipStreamSynchronize(s));
    we.n_edges = 2 * pairs;
    if (we.n_edges >= (int64_t(1) << 31)) {
        set_error("%lld word-word edges exceed the int32 edge count of the reference layout",
                  (long long)we.n_edges);
        return TGCN_E_RANGE;
    }
    {
        void *p = nullptr, *q = nullptr;
        if (hipMalloc(&p, sizeof(int32_t) * 2 * std::max<int64_t>(we.n_edges, 1)) != hipSuccess ||
            hipMalloc(&q, sizeof(float) * std::max<int64_t>(we.n_edges, 1)) != hipSuccess) {
            if (p) (void)hipFree(p);
            set_error("hipMalloc of %lld edges failed", (long long)we.n_edges);
            return TGCN_E_NOMEM;
        }
        we.coo = static_cast<int32_t *>(p);
        we.weights = static_cast<float *>(q);
    }
    if (pairs > 0) {
        k_emit_edges<<<static_cast<unsigned>((V + 3) / 4), 256, 0, s>>>(we.cij, V, nw, row_off, we.coo,
                                                                      we.weights);
        TGCN_HIP_CHECK(hipGetLastError());
    }
    TGCN_HIP_CHECK(hipStreamSynchronize(s));
    return TGCN_OK;
}

}  // namespace
}  // namespace tgcn

extern "C" {

int tgcn_wwedges_destroy(tgcn_wwedges *we) {
    using namespace tgcn;
    if (!we) return TGCN_OK;
    Guard g;
    TGCN_CHECK(g.enter(we->device));
    if (we->cij) (void)hipFree(we->cij);
    if (we->coo) (void)hipFree(we->coo);
    if (we->weights) (void)hipFree(we->weights);
    delete we;
    return TGCN_OK;
}

int tgcn_wwedges_create(const int32_t *X, int64_t n_docs, int64_t seq_len, int64_t n_vocab,
                        int64_t window, int device, tgcn_stream stream, tgcn_wwedges **out) {
    using namespace tgcn;
    if (!out) {
        set_error("tgcn_wwedges_create: out is NULL");
        return TGCN_E_INVALID;
    }
    *out = nullptr;
    if (n_docs < 0 || seq_len <= 0 || n_vocab <= 0 || window <= 0 || (n_docs > 0 && !X)) {
        set_error("tgcn_wwedges_create: need n_docs >= 0, seq_len, n_vocab, window > 0 and X "
                  "(n_docs=%lld seq_len=%lld n_vocab=%lld window=%lld)", (long long)n_docs,
                  (long long)seq_len, (long long)n_vocab, (long long)window);
        return TGCN_E_INVALID;
    }
    if (n_vocab >= (int64_t(1) << 31)) {
        set_error("tgcn_wwedges_create: n_vocab exceeds int32");
        return TGCN_E_RANGE;
    }
    Guard g;
    TGCN_CHECK(g.enter(device));
    tgcn_wwedges *we = new (std::nothrow) tgcn_wwedges();
    if (!we) {
        set_error("tgcn_wwedges_create: host allocation failed");
        return TGCN_E_NOMEM;
    }
    we->device = device;
    we->n_vocab = n_vocab;
    const int st = build(*we, X, n_docs, seq_len, n_vocab, window, static_cast<hipStream_t>(stream));
    if (st != TGCN_OK) {
        if (we->cij) (void)hipFree(we->cij);
        if (we->coo) (void)hipFree(we->coo);
        if (we->weights) (void)hipFree(we->weights);
        delete we;
        return st;
    }
    *out = we;
    return TGCN_OK;
}

int tgcn_wwedges_query(const tgcn_wwedges *we, int what, int64_t *out) {
    using namespace tgcn;
    if (!we || !out) {
        set_error("tgcn_wwedges_query: NULL argument");
        return TGCN_E_INVALID;
    }
    switch (what) {
        case TGCN_WW_N_EDGES: *out = we->n_edges; break;
        case TGCN_WW_N_WINDOWS: *out = we->n_windows; break;
        case TGCN_WW_N_COUNTS: *out = we->n_vocab * (we->n_vocab + 1) / 2; break;
        default:
            set_error("tgcn_wwedges_query: unknown selector %d", what);
            return TGCN_E_INVALID;
    }
    return TGCN_OK;
}

int tgcn_wwedges_export(const tgcn_wwedges *we, int32_t *coo, float *weights, uint32_t *cij,
                        tgcn_stream stream) {
    using namespace tgcn;
    if (!we) {
        set_error("tgcn_wwedges_export: NULL handle");
        return TGCN_E_INVALID;
    }
    Guard g;
    TGCN_CHECK(g.enter(we->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (coo && we->n_edges)
        TGCN_HIP_CHECK(hipMemcpyAsync(coo, we->coo, sizeof(int32_t) * 2 * we->n_edges, hipMemcpyDefault, s));
    if (weights && we->n_edges)
        TGCN_HIP_CHECK(hipMemcpyAsync(weights, we->weights, sizeof(float) * we->n_edges, hipMemcpyDefault, s));
    if (cij)
        TGCN_HIP_CHECK(hipMemcpyAsync(cij, we->cij, sizeof(uint32_t) * (we->n_vocab * (we->n_vocab + 1) / 2),
                                      hipMemcpyDefault, s));
    TGCN_HIP_CHECK(hipStreamSynchronize(s));
    return TGCN_OK;
}

}  // extern "C"
