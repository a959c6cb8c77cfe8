// Word-word PMI edges on the GPU: replaces the reference's only native component, the Cython module
// textgcn/lib/clib/graphbuilder.pyx (`compute_word_word_edges`, :23-68; `sliding_window`, :71-115;
// `edges_from_counts`, :118-211), called from Text2GraphTransformer at text2graph.py:156-160.
//
// Integer / HBM-atomic bound work, written for 288 GB of HBM rather than for MFMA.  Two counters:
//   * DENSE (small vocabularies: the packed triangle takes <= 16 GiB, V <= ~92 000; the faster one): the co-occurrence
//     counts c_ij are a packed upper triangle (uint32, V(V+1)/2 entries) in HBM, updated with no-return integer atomics;
//   * SPARSE (everything larger -- the triangle is 80 GB at V = 200 000 and 320 GB at V = 400 000, the reference's own
//     O(V^2) array at graphbuilder.pyx:44,134): the pairs of a bounded chunk of documents are written out as
//     (min << bits(V) | max, count) records, radix-sorted on exactly those 2 bits(V) key bits, summed by key and merged into ONE running sorted list of distinct
//     pairs, whose order IS the reference's emission order (upper triangle, row-major); memory is O(chunk + distinct
//     pairs), PMI and the ordered emission walk that list;
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

#include <rocprim/device/device_merge.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_reduce_by_key.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "common.h"

struct tgcn_wwedges {
    int device = 0;
    int64_t n_vocab = 0;
    int64_t n_windows = 0;
    int64_t n_edges = 0;
    uint32_t *cij = nullptr;   // packed upper triangle incl. diagonal (dense counter only)
    int32_t *coo = nullptr;    // [n_edges][2]
    float *weights = nullptr;  // [n_edges]
    bool sparse = false;       // the sparse counter ran: the counts are the sorted list below
    int64_t n_pairs = 0;       // distinct (i <= j) pairs with a count
    int key_shift = 32;              // bits(n_vocab)
    uint64_t *pair_keys = nullptr;   // [n_pairs] i << key_shift | j, ascending = upper triangle row-major
    uint32_t *pair_cnt = nullptr;    // [n_pairs]
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
// (The number of windows is the sum of last_start + 1, taken by a device reduction afterwards: one atomicAdd per document
// on a single counter serialised -- 1.2 ms per 100 k documents.)
__global__ void k_last_window(const int32_t *__restrict__ X, int64_t n_docs, int64_t L, int64_t w,
                              int32_t *__restrict__ last_start) {
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
    if (lane == 0) last_start[d] = static_cast<int32_t>(first_bad - 1);
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

// Where a counted pair goes: the dense triangle (atomicAdd) or, for the sparse counter, a (key, count) record appended to
// the chunk's buffer -- one cursor reservation per wavefront instruction (ballot), so the order of the records is
// arbitrary; they are sorted and their integer counts summed afterwards, which does not depend on it.  The cursor is a
// WORKGROUP-private LDS counter over the workgroup's own region of the buffer (a first, store-free pass counts every
// workgroup's records; an exclusive scan gives the regions): one global cursor for the whole chunk serialised 3.7 M
// same-address atomics and was 79 % of the builder's time (60 of 77 ms on a 100 k-document corpus).
struct PairSink {
    uint64_t *keys;
    uint32_t *cnt;
    unsigned long long *cursor;
    unsigned long long capacity;
    int shift;                   // key = i << shift | j with shift = bits(V): no dead bits for the radix sort to walk
};

// `region` != nullptr: the LDS cursor of this workgroup, whose records start at record `region_base` of the buffer;
// nullptr: the chunk-wide cursor in global memory (the tiny hot-pair chunk)
__device__ __forceinline__ void append_pair(const PairSink &s, bool emit, int64_t a, int64_t b, uint32_t c,
                                            unsigned int *region = nullptr, unsigned long long region_base = 0) {
    const unsigned long long m = __ballot(emit);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll(static_cast<long long>(m)) - 1;
    unsigned long long base = 0;
    if (lane == leader) {
        if (region != nullptr)
            base = region_base + atomicAdd(region, static_cast<unsigned int>(__popcll(m)));
        else
            base = atomicAdd(s.cursor, static_cast<unsigned long long>(__popcll(m)));
    }
    const unsigned lo32 = __shfl(static_cast<unsigned>(base), leader, 64);
    const unsigned hi32 = __shfl(static_cast<unsigned>(base >> 32), leader, 64);
    base = (static_cast<unsigned long long>(hi32) << 32) | lo32;
    if (emit) {
        const unsigned long long at = base + __popcll(m & ((1ull << lane) - 1ull));
        if (at < s.capacity) {                        // (the host sizes a chunk by its worst case and checks the cursor)
            const uint64_t i = static_cast<uint64_t>(a < b ? a : b), j = static_cast<uint64_t>(a < b ? b : a);
            s.keys[at] = (i << s.shift) | j;
            s.cnt[at] = c;
        }
    }
}

// One thread per token position of the documents [d0, d1); the pairs of hot words are counted in LDS (hot_of[word] = hot
// index or -1; hot_of == nullptr: no hot words) and flushed once per workgroup into `hot_out` -- the dense triangle
// itself (MODE 0: hot_out == cij, addressed through hot_word) or a small global triangle of kHotTri counters (MODE 1).
// MODE 2 is the store-free first pass of the sparse counter: it only COUNTS the records MODE 1 will write for the same
// grid (block_records[blockIdx.x]); an exclusive scan of those counts gives every workgroup its region (`block_base`).
// Every loop is wave-uniform (append_pair reserves per wavefront).
template <int MODE>
__global__ __launch_bounds__(256) void k_pair_counts_hot(const int32_t *__restrict__ X, int64_t d0, int64_t d1, int64_t L,
                                                         int64_t w, int64_t V, const int32_t *__restrict__ last_start,
                                                         const int16_t *__restrict__ hot_of,
                                                         const int32_t *__restrict__ hot_word, uint32_t *__restrict__ cij,
                                                         uint32_t *__restrict__ hot_tri, PairSink sink,
                                                         unsigned long long *__restrict__ block_records,
                                                         const unsigned long long *__restrict__ block_base) {
    constexpr bool SPARSE = MODE != 0, COUNT = MODE == 2;
    __shared__ uint32_t tri[kHotTri];
    __shared__ unsigned int region;                    // MODE 1: records written so far; MODE 2: records counted
    if constexpr (!COUNT)
        for (int j = threadIdx.x; j < kHotTri; j += blockDim.x) tri[j] = 0;
    if (threadIdx.x == 0) region = 0;
    __syncthreads();
    const unsigned long long region_base = (MODE == 1) ? block_base[blockIdx.x] : 0ull;
    unsigned int mine = 0;                             // MODE 2
    const int64_t n = (d1 - d0) * L;
    const int64_t per = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int64_t b = int64_t(blockIdx.x) * per, e = std::min(n, b + per);
    const int64_t w_eff = std::min(w, L);
    for (int64_t base = b; base < e; base += 256) {
        const int64_t idx = base + threadIdx.x;
        const bool valid = idx < e;
        const int64_t d = d0 + (valid ? idx / L : 0), k = valid ? idx % L : 0;
        const int32_t *x = X + d * L;
        const int32_t a = valid ? x[k] : -1;
        bool alive = a != -1;
        const int ha = (alive && hot_of != nullptr) ? hot_of[a] : -1;
        const int64_t jd = last_start[d];
        const int64_t l_end = std::min(L, k + w);
        const int64_t hi = std::min(k, jd);
        for (int64_t t = 0; t < w_eff; ++t) {
            const int64_t l = k + t;
            alive = alive && l < l_end;
            const int32_t bb = alive ? x[l] : -1;
            alive = alive && bb != -1;                    // graphbuilder.pyx:106-111: the first padding value ends the window
            const int64_t lo = std::max<int64_t>(0, l - w + 1);
            const bool counts = alive && hi >= lo;
            const uint32_t c = static_cast<uint32_t>(hi - lo + 1);
            const int hb = (counts && ha >= 0) ? hot_of[bb] : -1;
            if constexpr (COUNT) {
                mine += (counts && hb < 0) ? 1u : 0u;
                continue;
            }
            if (hb >= 0) {
                atomicAdd(&tri[sym_diag_idx(ha, hb, kHotWords)], c);
            } else if constexpr (!SPARSE) {
                if (counts) atomicAdd(&cij[sym_diag_idx(a, bb, V)], c);
            }
            if constexpr (MODE == 1) append_pair(sink, counts && hb < 0, a, bb, c, &region, region_base);
        }
    }
    if constexpr (COUNT) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&region, mine);
        __syncthreads();
        if (threadIdx.x == 0) block_records[blockIdx.x] = region;
        return;
    }
    __syncthreads();
    // flush: packed index -> (row >= col) of the hot triangle -> the two words
    for (int j = threadIdx.x; j < kHotTri; j += blockDim.x) {
        const uint32_t c = tri[j];
        if (c == 0) continue;
        if constexpr (SPARSE) {
            atomicAdd(&hot_tri[j], c);
        } else {
            int col = 0, rem = j;
            while (rem >= kHotWords - col) {              // column `col` of the packed triangle holds kHotWords - col entries
                rem -= kHotWords - col;
                ++col;
            }
            const int row = col + rem;
            atomicAdd(&cij[sym_diag_idx(hot_word[row], hot_word[col], V)], c);
        }
    }
}

// sparse counter: the global hot triangle as (key, count) records (one more, tiny, chunk)
__global__ void k_hot_tri_to_pairs(const uint32_t *__restrict__ hot_tri, const int32_t *__restrict__ hot_word, PairSink sink) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;            // grid covers kHotTri rounded up to whole waves
    uint32_t c = 0;
    int row = 0, col = 0;
    if (j < kHotTri) {
        c = hot_tri[j];
        int rem = j;
        while (rem >= kHotWords - col) {
            rem -= kHotWords - col;
            ++col;
        }
        row = col + rem;
    }
    append_pair(sink, c != 0, c ? hot_word[row] : 0, c ? hot_word[col] : 0, c);
}

// sparse counter: diag[i] = c_ii from the sorted list (zero elsewhere)
__global__ void k_pairs_diag(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ cnt, int64_t n, int shift,
                             uint32_t *__restrict__ diag) {
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint64_t k = keys[t];
    const uint64_t i = k >> shift, j = k & ((uint64_t(1) << shift) - 1);
    if (i == j) diag[i] = cnt[t];
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

// The same decision from raw counts (sparse counter): graphbuilder.pyx:147-162, float for float
__device__ __forceinline__ bool pmi_from_counts(uint32_t cii, uint32_t cjj, uint32_t cij, float nw, float *pmi_out) {
    const float pi = static_cast<float>(cii) / nw;
    const float pj = static_cast<float>(cjj) / nw;
    const float pij = static_cast<float>(cij) / nw;
    if (pij == 0.f || pi == 0.f || pj == 0.f) return false;
    const float pmi = static_cast<float>(log(static_cast<double>(pij / (pi * pj))));
    *pmi_out = pmi;
    return pmi > 1e-10f;
}

// flag[t] = 1 when the t-th distinct pair (i < j) becomes an edge
__global__ void k_pair_flags(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ cnt, int64_t n, int shift,
                             const uint32_t *__restrict__ diag, float nw, uint8_t *__restrict__ flag) {
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint64_t k = keys[t];
    const uint32_t i = static_cast<uint32_t>(k >> shift), j = static_cast<uint32_t>(k & ((uint64_t(1) << shift) - 1));
    float pmi;
    flag[t] = (i != j && pmi_from_counts(diag[i], diag[j], cnt[t], nw, &pmi)) ? 1 : 0;
}

// (i,j),(j,i) at 2 * rank of the pair among the edges: the list is sorted by (i, j), the reference's emission order
__global__ void k_emit_pairs(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ cnt, int64_t n, int shift,
                             const uint32_t *__restrict__ diag, float nw, const uint8_t *__restrict__ flag,
                             const int64_t *__restrict__ rank, int32_t *__restrict__ coo, float *__restrict__ weights) {
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= n || !flag[t]) return;
    const uint64_t key = keys[t];
    const int32_t i = static_cast<int32_t>(key >> shift), j = static_cast<int32_t>(key & ((uint64_t(1) << shift) - 1));
    float pmi = 0.f;
    (void)pmi_from_counts(diag[i], diag[j], cnt[t], nw, &pmi);
    const int64_t k = 2 * rank[t];
    coo[2 * k] = i;
    coo[2 * k + 1] = j;
    weights[k] = pmi;
    coo[2 * k + 2] = j;
    coo[2 * k + 3] = i;
    weights[k + 1] = pmi;
}

// dense export of the sparse counter's list (tests, sliding_window_tester): tri[sym_diag_idx(i, j)] = count
__global__ void k_pairs_to_triangle(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ cnt, int64_t n, int shift,
                                    int64_t V, uint32_t *__restrict__ tri) {
    const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint64_t k = keys[t];
    tri[sym_diag_idx(static_cast<int64_t>(k >> shift), static_cast<int64_t>(k & ((uint64_t(1) << shift) - 1)), V)] = cnt[t];
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

// the 128 most frequent words (token histogram in LDS-sized vocabulary chunks, top-H chosen on the host)
int find_hot_words(const int32_t *X, int64_t n, int64_t V, Tmp &hot_of_d, Tmp &hot_word_d, hipStream_t s) {
    Tmp hist;
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
    TGCN_HIP_CHECK(hipStreamSynchronize(s));      // the host vectors leave scope
    return TGCN_OK;
}

int alloc_edges(tgcn_wwedges &we, int64_t pairs) {
    we.n_edges = 2 * pairs;
    if (we.n_edges >= (int64_t(1) << 31)) {
        set_error("%lld word-word edges exceed the int32 edge count of the reference layout", (long long)we.n_edges);
        return TGCN_E_RANGE;
    }
    void *p = nullptr, *q = nullptr;
    if (hipMalloc(&p, sizeof(int32_t) * 2 * std::max<int64_t>(we.n_edges, 1)) != hipSuccess ||
        hipMalloc(&q, sizeof(float) * std::max<int64_t>(we.n_edges, 1)) != hipSuccess) {
        if (p) (void)hipFree(p);
        set_error("hipMalloc of %lld edges failed", (long long)we.n_edges);
        return TGCN_E_NOMEM;
    }
    we.coo = static_cast<int32_t *>(p);
    we.weights = static_cast<float *>(q);
    return TGCN_OK;
}

// ---- dense counter: the packed triangle in HBM ----------------------------------------------------------------
int build_dense(tgcn_wwedges &we, const int32_t *X, int64_t D, int64_t L, int64_t V, int64_t w, const int32_t *last,
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
    Tmp rows, offs;
    TGCN_CHECK(rows.alloc(sizeof(int64_t) * (V + 1)));
    TGCN_CHECK(offs.alloc(sizeof(int64_t) * (V + 1)));
    if (D > 0) {
        const int64_t n = D * L;
        if (V > kHotWords && n >= (int64_t(1) << 16)) {
            Tmp hot_of_d, hot_word_d;
            TGCN_CHECK(find_hot_words(X, n, V, hot_of_d, hot_word_d, s));
            const unsigned blocks = static_cast<unsigned>(std::min<int64_t>(2048, (n + 4095) / 4096));
            k_pair_counts_hot<0><<<blocks, 256, 0, s>>>(X, 0, D, L, w, V, last, static_cast<int16_t *>(hot_of_d.p),
                                                        static_cast<int32_t *>(hot_word_d.p), we.cij, nullptr,
                                                        PairSink{nullptr, nullptr, nullptr, 0, 32}, nullptr, nullptr);
            TGCN_HIP_CHECK(hipGetLastError());
            TGCN_HIP_CHECK(hipStreamSynchronize(s));      // the Tmp buffers leave scope
        } else {
            k_pair_counts<<<static_cast<unsigned>((n + 255) / 256), 256, 0, s>>>(X, D, L, w, V, last, we.cij);
            TGCN_HIP_CHECK(hipGetLastError());
        }
    }
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
    TGCN_CHECK(alloc_edges(we, pairs));
    if (pairs > 0) {
        k_emit_edges<<<static_cast<unsigned>((V + 3) / 4), 256, 0, s>>>(we.cij, V, nw, row_off, we.coo, we.weights);
        TGCN_HIP_CHECK(hipGetLastError());
    }
    TGCN_HIP_CHECK(hipStreamSynchronize(s));
    return TGCN_OK;
}

// ---- sparse counter: sorted list of distinct pairs, O(chunk + distinct pairs) memory -----------------------------
struct PairList {                 // a sorted run of (key, count) records on the device
    Tmp keys, cnt;
    int64_t n = 0;
};

struct ToInt64 {
    __host__ __device__ int64_t operator()(uint8_t f) const { return static_cast<int64_t>(f); }
};

struct WindowsOfDoc {             // last window start -> number of windows of the document
    __host__ __device__ unsigned long long operator()(int32_t last) const { return static_cast<unsigned long long>(last + 1); }
};

// records [0, n_c) of (in_k, in_c) -- any order, duplicates -- are sorted, summed by key and merged into `run`
int absorb_chunk(PairList &run, uint64_t *in_k, uint32_t *in_c, uint64_t *tmp_k, uint32_t *tmp_c, int64_t n_c, int key_bits,
                 hipStream_t s) {
    if (n_c == 0) return TGCN_OK;
    Tmp n_unique_d;
    TGCN_CHECK(n_unique_d.alloc(sizeof(unsigned long long)));
    unsigned long long h_unique = 0;
    {
        size_t bytes = 0;
        TGCN_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, in_k, tmp_k, in_c, tmp_c, static_cast<size_t>(n_c), 0u,
                                                 static_cast<unsigned>(key_bits), s));
        Tmp t;
        TGCN_CHECK(t.alloc(bytes));
        TGCN_HIP_CHECK(rocprim::radix_sort_pairs(t.p, bytes, in_k, tmp_k, in_c, tmp_c, static_cast<size_t>(n_c), 0u,
                                                 static_cast<unsigned>(key_bits), s));
        TGCN_HIP_CHECK(hipStreamSynchronize(s));
    }
    {   // sums of equal keys (uint32 arithmetic wraps as the reference's counters do): back into (in_k, in_c)
        size_t bytes = 0;
        TGCN_HIP_CHECK(rocprim::reduce_by_key(nullptr, bytes, tmp_k, tmp_c, static_cast<size_t>(n_c), in_k, in_c,
                                              static_cast<unsigned long long *>(n_unique_d.p), rocprim::plus<uint32_t>(),
                                              rocprim::equal_to<uint64_t>(), s));
        Tmp t;
        TGCN_CHECK(t.alloc(bytes));
        TGCN_HIP_CHECK(rocprim::reduce_by_key(t.p, bytes, tmp_k, tmp_c, static_cast<size_t>(n_c), in_k, in_c,
                                              static_cast<unsigned long long *>(n_unique_d.p), rocprim::plus<uint32_t>(),
                                              rocprim::equal_to<uint64_t>(), s));
        TGCN_HIP_CHECK(hipMemcpyAsync(&h_unique, n_unique_d.p, sizeof(h_unique), hipMemcpyDeviceToHost, s));
        TGCN_HIP_CHECK(hipStreamSynchronize(s));
    }
    const int64_t u = static_cast<int64_t>(h_unique);
    if (run.n == 0) {
        TGCN_CHECK(run.keys.alloc(sizeof(uint64_t) * u));
        TGCN_CHECK(run.cnt.alloc(sizeof(uint32_t) * u));
        TGCN_HIP_CHECK(hipMemcpyAsync(run.keys.p, in_k, sizeof(uint64_t) * u, hipMemcpyDeviceToDevice, s));
        TGCN_HIP_CHECK(hipMemcpyAsync(run.cnt.p, in_c, sizeof(uint32_t) * u, hipMemcpyDeviceToDevice, s));
        TGCN_HIP_CHECK(hipStreamSynchronize(s));
        run.n = u;
        return TGCN_OK;
    }
    // merge the two sorted runs, then sum the keys both held
    const int64_t m = run.n + u;
    Tmp mk, mc, nk, nc;
    TGCN_CHECK(mk.alloc(sizeof(uint64_t) * m));
    TGCN_CHECK(mc.alloc(sizeof(uint32_t) * m));
    {
        size_t bytes = 0;
        TGCN_HIP_CHECK(rocprim::merge(nullptr, bytes, static_cast<uint64_t *>(run.keys.p), in_k, static_cast<uint64_t *>(mk.p),
                                      static_cast<uint32_t *>(run.cnt.p), in_c, static_cast<uint32_t *>(mc.p),
                                      static_cast<size_t>(run.n), static_cast<size_t>(u), rocprim::less<uint64_t>(), s));
        Tmp t;
        TGCN_CHECK(t.alloc(bytes));
        TGCN_HIP_CHECK(rocprim::merge(t.p, bytes, static_cast<uint64_t *>(run.keys.p), in_k, static_cast<uint64_t *>(mk.p),
                                      static_cast<uint32_t *>(run.cnt.p), in_c, static_cast<uint32_t *>(mc.p),
                                      static_cast<size_t>(run.n), static_cast<size_t>(u), rocprim::less<uint64_t>(), s));
        TGCN_HIP_CHECK(hipStreamSynchronize(s));
    }
    if (run.keys.p) (void)hipFree(run.keys.p);
    if (run.cnt.p) (void)hipFree(run.cnt.p);
    run.keys.p = run.cnt.p = nullptr;
    TGCN_CHECK(nk.alloc(sizeof(uint64_t) * m));
    TGCN_CHECK(nc.alloc(sizeof(uint32_t) * m));
    {
        size_t bytes = 0;
        TGCN_HIP_CHECK(rocprim::reduce_by_key(nullptr, bytes, static_cast<uint64_t *>(mk.p), static_cast<uint32_t *>(mc.p),
                                              static_cast<size_t>(m), static_cast<uint64_t *>(nk.p),
                                              static_cast<uint32_t *>(nc.p), static_cast<unsigned long long *>(n_unique_d.p),
                                              rocprim::plus<uint32_t>(), rocprim::equal_to<uint64_t>(), s));
        Tmp t;
        TGCN_CHECK(t.alloc(bytes));
        TGCN_HIP_CHECK(rocprim::reduce_by_key(t.p, bytes, static_cast<uint64_t *>(mk.p), static_cast<uint32_t *>(mc.p),
                                              static_cast<size_t>(m), static_cast<uint64_t *>(nk.p),
                                              static_cast<uint32_t *>(nc.p), static_cast<unsigned long long *>(n_unique_d.p),
                                              rocprim::plus<uint32_t>(), rocprim::equal_to<uint64_t>(), s));
        TGCN_HIP_CHECK(hipMemcpyAsync(&h_unique, n_unique_d.p, sizeof(h_unique), hipMemcpyDeviceToHost, s));
        TGCN_HIP_CHECK(hipStreamSynchronize(s));
    }
    run.keys.p = nk.p;
    run.cnt.p = nc.p;
    nk.p = nc.p = nullptr;
    run.n = static_cast<int64_t>(h_unique);
    return TGCN_OK;
}

int build_sparse(tgcn_wwedges &we, const int32_t *X, int64_t D, int64_t L, int64_t V, int64_t w, const int32_t *last,
                 hipStream_t s) {
    we.sparse = true;
    int shift = 1;
    while ((int64_t(1) << shift) < V) ++shift;
    we.key_shift = shift;
    const int key_bits = 2 * shift;                             // key = i << shift | j with i, j < V <= 2^shift (<= 62 bits)
    // records per chunk: the worst case of a chunk of documents (every position pairs with min(w, L) followers) must fit
    int64_t budget = int64_t(1) << 27;
    if (const char *e = std::getenv("TGCN_WW_CHUNK_PAIRS")) budget = std::max<int64_t>(1, std::atoll(e));
    const int64_t per_doc = L * std::min(w, L);
    const int64_t docs_per_chunk = std::max<int64_t>(1, budget / std::max<int64_t>(1, per_doc));
    const int64_t cap = std::max<int64_t>(std::max<int64_t>(docs_per_chunk * per_doc, kHotTri), 64);
    Tmp in_k, in_c, tmp_k, tmp_c, cursor, hot_tri, hot_of_d, hot_word_d;
    TGCN_CHECK(in_k.alloc(sizeof(uint64_t) * cap));
    TGCN_CHECK(in_c.alloc(sizeof(uint32_t) * cap));
    TGCN_CHECK(tmp_k.alloc(sizeof(uint64_t) * cap));
    TGCN_CHECK(tmp_c.alloc(sizeof(uint32_t) * cap));
    TGCN_CHECK(cursor.alloc(sizeof(unsigned long long)));
    TGCN_CHECK(hot_tri.alloc(sizeof(uint32_t) * kHotTri));
    TGCN_HIP_CHECK(hipMemsetAsync(hot_tri.p, 0, sizeof(uint32_t) * kHotTri, s));
    const int64_t n_all = D * L;
    const bool hot = D > 0 && V > kHotWords && n_all >= (int64_t(1) << 16);
    if (hot) TGCN_CHECK(find_hot_words(X, n_all, V, hot_of_d, hot_word_d, s));
    const PairSink sink{static_cast<uint64_t *>(in_k.p), static_cast<uint32_t *>(in_c.p),
                        static_cast<unsigned long long *>(cursor.p), static_cast<unsigned long long>(cap), shift};
    PairList run;
    auto take = [&]() -> int {                                  // sort / sum / merge what the last launch appended
        unsigned long long n_c = 0;
        TGCN_HIP_CHECK(hipMemcpyAsync(&n_c, cursor.p, sizeof(n_c), hipMemcpyDeviceToHost, s));
        TGCN_HIP_CHECK(hipStreamSynchronize(s));
        if (n_c > static_cast<unsigned long long>(cap)) {
            set_error("graph builder: a chunk produced %llu pair records for a buffer of %lld", n_c, (long long)cap);
            return TGCN_E_INVALID;
        }
        return absorb_chunk(run, sink.keys, sink.cnt, static_cast<uint64_t *>(tmp_k.p), static_cast<uint32_t *>(tmp_c.p),
                            static_cast<int64_t>(n_c), key_bits, s);
    };
    constexpr int kMaxBlocks = 2048;
    Tmp blk_n, blk_off, blk_tmp;
    TGCN_CHECK(blk_n.alloc(sizeof(unsigned long long) * (kMaxBlocks + 1)));
    TGCN_CHECK(blk_off.alloc(sizeof(unsigned long long) * (kMaxBlocks + 1)));
    size_t blk_tmp_bytes = 0;
    TGCN_HIP_CHECK(rocprim::exclusive_scan(nullptr, blk_tmp_bytes, static_cast<unsigned long long *>(blk_n.p),
                                           static_cast<unsigned long long *>(blk_off.p), 0ull, size_t(kMaxBlocks + 1),
                                           rocprim::plus<unsigned long long>(), s));
    TGCN_CHECK(blk_tmp.alloc(blk_tmp_bytes));
    for (int64_t d0 = 0; d0 < D; d0 += docs_per_chunk) {
        const int64_t d1 = std::min(D, d0 + docs_per_chunk);
        const int64_t n = (d1 - d0) * L;
        const unsigned blocks = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(kMaxBlocks, (n + 4095) / 4096)));
        const int16_t *hof = hot ? static_cast<int16_t *>(hot_of_d.p) : nullptr;
        const int32_t *hwd = hot ? static_cast<int32_t *>(hot_word_d.p) : nullptr;
        // pass 1 (no stores): records per workgroup -> exclusive scan -> every workgroup's region of the buffer
        TGCN_HIP_CHECK(hipMemsetAsync(blk_n.p, 0, sizeof(unsigned long long) * (blocks + 1), s));
        k_pair_counts_hot<2><<<blocks, 256, 0, s>>>(X, d0, d1, L, w, V, last, hof, hwd, nullptr, nullptr, sink,
                                                    static_cast<unsigned long long *>(blk_n.p), nullptr);
        TGCN_HIP_CHECK(hipGetLastError());
        TGCN_HIP_CHECK(rocprim::exclusive_scan(blk_tmp.p, blk_tmp_bytes, static_cast<unsigned long long *>(blk_n.p),
                                               static_cast<unsigned long long *>(blk_off.p), 0ull, size_t(blocks + 1),
                                               rocprim::plus<unsigned long long>(), s));
        unsigned long long n_c = 0;
        TGCN_HIP_CHECK(hipMemcpyAsync(&n_c, static_cast<unsigned long long *>(blk_off.p) + blocks, sizeof(n_c),
                                      hipMemcpyDeviceToHost, s));
        TGCN_HIP_CHECK(hipStreamSynchronize(s));
        if (n_c > static_cast<unsigned long long>(cap)) {
            set_error("graph builder: a chunk counts %llu pair records for a buffer of %lld", n_c, (long long)cap);
            return TGCN_E_INVALID;
        }
        // pass 2: the records, each workgroup into its own region through an LDS cursor
        k_pair_counts_hot<1><<<blocks, 256, 0, s>>>(X, d0, d1, L, w, V, last, hof, hwd, nullptr,
                                                    static_cast<uint32_t *>(hot_tri.p), sink, nullptr,
                                                    static_cast<unsigned long long *>(blk_off.p));
        TGCN_HIP_CHECK(hipGetLastError());
        TGCN_CHECK(absorb_chunk(run, sink.keys, sink.cnt, static_cast<uint64_t *>(tmp_k.p), static_cast<uint32_t *>(tmp_c.p),
                                static_cast<int64_t>(n_c), key_bits, s));
    }
    if (hot) {                                                  // the pairs among the hot words: one more, tiny, chunk
        TGCN_HIP_CHECK(hipMemsetAsync(cursor.p, 0, sizeof(unsigned long long), s));
        k_hot_tri_to_pairs<<<(kHotTri + 255) / 256, 256, 0, s>>>(static_cast<uint32_t *>(hot_tri.p),
                                                                 static_cast<int32_t *>(hot_word_d.p), sink);
        TGCN_HIP_CHECK(hipGetLastError());
        TGCN_CHECK(take());
    }
    // the chunk buffers are done with: release them before the emission's own arrays are made
    for (Tmp *t : {&in_k, &in_c, &tmp_k, &tmp_c}) {
        (void)hipFree(t->p);
        t->p = nullptr;
    }
    we.n_pairs = run.n;
    we.pair_keys = static_cast<uint64_t *>(run.keys.p);
    we.pair_cnt = static_cast<uint32_t *>(run.cnt.p);
    run.keys.p = run.cnt.p = nullptr;                           // owned by the handle from here on
    const int64_t U = we.n_pairs;
    const float nw = static_cast<float>(we.n_windows);          // graphbuilder.pyx:147 <float>n_windows
    Tmp diag, flag, rank;
    TGCN_CHECK(diag.alloc(sizeof(uint32_t) * V));
    TGCN_CHECK(flag.alloc(sizeof(uint8_t) * (U + 1)));
    TGCN_CHECK(rank.alloc(sizeof(int64_t) * (U + 1)));
    TGCN_HIP_CHECK(hipMemsetAsync(diag.p, 0, sizeof(uint32_t) * V, s));
    TGCN_HIP_CHECK(hipMemsetAsync(flag.p, 0, sizeof(uint8_t) * (U + 1), s));
    const unsigned grid = static_cast<unsigned>((U + 255) / 256);
    if (U > 0) {
        k_pairs_diag<<<grid, 256, 0, s>>>(we.pair_keys, we.pair_cnt, U, shift, static_cast<uint32_t *>(diag.p));
        k_pair_flags<<<grid, 256, 0, s>>>(we.pair_keys, we.pair_cnt, U, shift, static_cast<uint32_t *>(diag.p), nw,
                                          static_cast<uint8_t *>(flag.p));
        TGCN_HIP_CHECK(hipGetLastError());
    }
    auto flags64 = rocprim::make_transform_iterator(static_cast<const uint8_t *>(flag.p), ToInt64());
    size_t tmp_bytes = 0;
    TGCN_HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, flags64, static_cast<int64_t *>(rank.p), int64_t(0),
                                           static_cast<size_t>(U + 1), rocprim::plus<int64_t>(), s));
    Tmp scan_tmp;
    TGCN_CHECK(scan_tmp.alloc(tmp_bytes));
    TGCN_HIP_CHECK(rocprim::exclusive_scan(scan_tmp.p, tmp_bytes, flags64, static_cast<int64_t *>(rank.p), int64_t(0),
                                           static_cast<size_t>(U + 1), rocprim::plus<int64_t>(), s));
    int64_t pairs = 0;
    TGCN_HIP_CHECK(hipMemcpyAsync(&pairs, static_cast<int64_t *>(rank.p) + U, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    TGCN_HIP_CHECK(hipStreamSynchronize(s));
    TGCN_CHECK(alloc_edges(we, pairs));
    if (pairs > 0) {
        k_emit_pairs<<<grid, 256, 0, s>>>(we.pair_keys, we.pair_cnt, U, shift, static_cast<uint32_t *>(diag.p), nw,
                                          static_cast<uint8_t *>(flag.p), static_cast<int64_t *>(rank.p), we.coo, we.weights);
        TGCN_HIP_CHECK(hipGetLastError());
    }
    TGCN_HIP_CHECK(hipStreamSynchronize(s));
    return TGCN_OK;
}

constexpr int64_t kDenseTriangleMaxBytes = int64_t(16) << 30;   // the dense counter up to here (V <= ~92 000)

int build(tgcn_wwedges &we, const int32_t *X, int64_t D, int64_t L, int64_t V, int64_t w, hipStream_t s) {
    Tmp last, nwin;
    TGCN_CHECK(last.alloc(sizeof(int32_t) * std::max<int64_t>(D, 1)));
    TGCN_CHECK(nwin.alloc(sizeof(unsigned long long)));
    TGCN_HIP_CHECK(hipMemsetAsync(nwin.p, 0, sizeof(unsigned long long), s));
    if (D > 0) {
        k_last_window<<<static_cast<unsigned>((D + 3) / 4), 256, 0, s>>>(X, D, L, w, static_cast<int32_t *>(last.p));
        TGCN_HIP_CHECK(hipGetLastError());
        auto per_doc = rocprim::make_transform_iterator(static_cast<const int32_t *>(last.p), WindowsOfDoc());
        size_t bytes = 0;
        TGCN_HIP_CHECK(rocprim::reduce(nullptr, bytes, per_doc, static_cast<unsigned long long *>(nwin.p), 0ull,
                                       static_cast<size_t>(D), rocprim::plus<unsigned long long>(), s));
        Tmp t;
        TGCN_CHECK(t.alloc(bytes));
        TGCN_HIP_CHECK(rocprim::reduce(t.p, bytes, per_doc, static_cast<unsigned long long *>(nwin.p), 0ull,
                                       static_cast<size_t>(D), rocprim::plus<unsigned long long>(), s));
        TGCN_HIP_CHECK(hipStreamSynchronize(s));                  // (the scratch buffer leaves scope)
    }
    unsigned long long h_nw = 0;
    TGCN_HIP_CHECK(hipMemcpyAsync(&h_nw, nwin.p, sizeof(h_nw), hipMemcpyDeviceToHost, s));
    TGCN_HIP_CHECK(hipStreamSynchronize(s));
    we.n_windows = static_cast<int64_t>(h_nw);
    // which counter: the dense triangle while it is small (faster: one atomic per pair, no sort), the sorted pair list
    // beyond -- TGCN_WW_COUNTER=dense / sparse pins it (tests run both on the same inputs)
    bool sparse = V * (V + 1) / 2 * int64_t(sizeof(uint32_t)) > kDenseTriangleMaxBytes;
    if (const char *e = std::getenv("TGCN_WW_COUNTER")) {
        if (std::strcmp(e, "sparse") == 0) sparse = true;
        if (std::strcmp(e, "dense") == 0) sparse = false;
    }
    return sparse ? build_sparse(we, X, D, L, V, w, static_cast<int32_t *>(last.p), s)
                  : build_dense(we, X, D, L, V, w, static_cast<int32_t *>(last.p), s);
}

void free_outputs(tgcn_wwedges &we) {
    for (void *p : {static_cast<void *>(we.cij), static_cast<void *>(we.coo), static_cast<void *>(we.weights),
                    static_cast<void *>(we.pair_keys), static_cast<void *>(we.pair_cnt)})
        if (p) (void)hipFree(p);
    we.cij = nullptr;
    we.coo = nullptr;
    we.weights = nullptr;
    we.pair_keys = nullptr;
    we.pair_cnt = nullptr;
}

}  // namespace
}  // namespace tgcn

extern "C" {

int tgcn_wwedges_destroy(tgcn_wwedges *we) {
    using namespace tgcn;
    if (!we) return TGCN_OK;
    Guard g;
    TGCN_CHECK(g.enter(we->device));
    free_outputs(*we);
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
        free_outputs(*we);
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
        case TGCN_WW_SPARSE: *out = we->sparse ? 1 : 0; break;
        case TGCN_WW_N_PAIRS: *out = we->sparse ? we->n_pairs : -1; break;
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
    const int64_t tri = we->n_vocab * (we->n_vocab + 1) / 2;
    if (cij && !we->sparse)
        TGCN_HIP_CHECK(hipMemcpyAsync(cij, we->cij, sizeof(uint32_t) * tri, hipMemcpyDefault, s));
    if (cij && we->sparse) {
        // the sorted list laid out as the packed triangle the reference's test hook returns (a small-vocabulary affair:
        // the triangle is built in device memory first)
        Tmp t;
        TGCN_CHECK(t.alloc(sizeof(uint32_t) * static_cast<size_t>(tri)));
        TGCN_HIP_CHECK(hipMemsetAsync(t.p, 0, sizeof(uint32_t) * static_cast<size_t>(tri), s));
        if (we->n_pairs > 0) {
            k_pairs_to_triangle<<<static_cast<unsigned>((we->n_pairs + 255) / 256), 256, 0, s>>>(
                we->pair_keys, we->pair_cnt, we->n_pairs, we->key_shift, we->n_vocab, static_cast<uint32_t *>(t.p));
            TGCN_HIP_CHECK(hipGetLastError());
        }
        TGCN_HIP_CHECK(hipMemcpyAsync(cij, t.p, sizeof(uint32_t) * static_cast<size_t>(tri), hipMemcpyDefault, s));
        TGCN_HIP_CHECK(hipStreamSynchronize(s));
    }
    TGCN_HIP_CHECK(hipStreamSynchronize(s));
    return TGCN_OK;
}

}  // extern "C"
