/* CPU oracle, part 3: restatement of the reference's only native component, the Cython graph
 * builder textgcn/lib/clib/graphbuilder.pyx -- TEST INFRASTRUCTURE (only tests/, smoke() and
 * bench.py's cpu_baseline leg may load it; the product's builder is pytextgcn_amd/csrc/graphbuilder.hip).
 *
 * PINNED by the reference's own golden vector: textgcn/test/test_cfunc.py:83-99 (expected packed
 * c_ij for a 2x8 token matrix, V = 6, window 3) and by the outputs of the reference module itself,
 * compiled here as oracle/_ref/graphbuilder*.so (oracle/Makefile, target `ref`) and compared on
 * random inputs in tests/test_graphbuilder.py.
 *
 * Differences from the reference, all outside its defined behaviour: 64-bit packed indices (the
 * reference's `unsigned int` N*(N+1) wraps for V > 65535, graphbuilder.pyx:224-259); window_size >
 * seq_len is one window over the whole row (the reference's unsigned `seq_len - window_size + 1`
 * wraps, graphbuilder.pyx:92); outputs are caller-allocated (the reference leaks its malloc'd
 * arrays, graphbuilder.pyx:65-66).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* graphbuilder.pyx:214-227: packed upper triangle INCLUDING the diagonal, row-major */
static inline int64_t sym_diag_idx(int64_t row, int64_t col, int64_t n)
{
    if (row >= col) return col * n + row - (col + 1) * col / 2;
    return row * n + col - (row + 1) * row / 2;
}

int64_t oracle_sym_size_diag(int64_t n) { return n * (n + 1) / 2; }

/* graphbuilder.pyx:71-115.  c_ij must hold oracle_sym_size_diag(n_vocab) zeros.  Returns n_windows. */
int64_t oracle_sliding_window(const int32_t *X, uint32_t *c_ij, int64_t window, int64_t n_vocab,
                              int64_t n_docs, int64_t seq_len)
{
    int64_t n_windows = 0;
    const int64_t n_starts = seq_len >= window ? seq_len - window + 1 : 1;
    for (int64_t i = 0; i < n_docs; ++i) {
        const int32_t *x = X + i * seq_len;
        for (int64_t j = 0; j < n_starts; ++j) {
            const int64_t end = j + window < seq_len ? j + window : seq_len;   /* exclusive */
            if (j != 0 && x[j + window - 1] == -1) break;      /* :96-98 window reaches the padding */
            ++n_windows;                                        /* :99 */
            for (int64_t k = j; k < end; ++k) {                 /* :101 */
                for (int64_t l = k; l < end; ++l) {             /* :103 pairs k <= l, diagonal included */
                    if (x[k] != -1 && x[l] != -1)               /* :106 */
                        c_ij[sym_diag_idx(x[k], x[l], n_vocab)] += 1;
                    else
                        break;                                  /* :110-111 */
                }
            }
        }
    }
    return n_windows;
}

/* graphbuilder.pyx:118-211, first sweep: number of (directed) edges that will be emitted. */
int64_t oracle_count_edges(const uint32_t *c_ij, int64_t n_vocab, int64_t n_windows)
{
    int64_t n = 0;
    const float nw = (float)n_windows;
    for (int64_t i = 0; i + 1 < n_vocab; ++i) {
        const float pi = (float)c_ij[sym_diag_idx(i, i, n_vocab)] / nw;
        for (int64_t j = i + 1; j < n_vocab; ++j) {
            const float pj = (float)c_ij[sym_diag_idx(j, j, n_vocab)] / nw;
            const float pij = (float)c_ij[sym_diag_idx(i, j, n_vocab)] / nw;
            if (pij == 0 || pi == 0 || pj == 0) continue;
            const float pmi = (float)log((double)(pij / (pi * pj)));   /* :161 libc double log */
            if (pmi > 1e-10f) n += 2;                                   /* :20,162 EPSILON */
        }
    }
    return n;
}

/* Second sweep (:181-192): (i,j),(j,i) interleaved, upper triangle in row-major order.
 * coo is [n_edges][2] int32, weights [n_edges] float32. */
void oracle_emit_edges(const uint32_t *c_ij, int64_t n_vocab, int64_t n_windows, int32_t *coo,
                       float *weights)
{
    int64_t k = 0;
    const float nw = (float)n_windows;
    for (int64_t i = 0; i + 1 < n_vocab; ++i) {
        const float pi = (float)c_ij[sym_diag_idx(i, i, n_vocab)] / nw;
        for (int64_t j = i + 1; j < n_vocab; ++j) {
            const float pj = (float)c_ij[sym_diag_idx(j, j, n_vocab)] / nw;
            const float pij = (float)c_ij[sym_diag_idx(i, j, n_vocab)] / nw;
            if (pij == 0 || pi == 0 || pj == 0) continue;
            const float pmi = (float)log((double)(pij / (pi * pj)));
            if (pmi > 1e-10f) {
                coo[2 * k] = (int32_t)i, coo[2 * k + 1] = (int32_t)j, weights[k] = pmi, ++k;
                coo[2 * k] = (int32_t)j, coo[2 * k + 1] = (int32_t)i, weights[k] = pmi, ++k;
            }
        }
    }
}
