/* CPU oracle, part 2: CSR restatement of the normalised propagate step -- TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library
 * (through oracle/csr_oracle.py).  It is never linked into, or called from, the product.
 *
 * What it restates: step (3) of PyG-1.6.3 GCNConv.forward as the reference invokes it at
 * textgcn/lib/models.py:20 -- out[col_e,:] += w_hat_e * XW[row_e,:] -- regrouped by target row
 * (the "CPU-csr" row of BASELINE.md section 3).  It exists so that full-size graphs (SURVEY.md
 * 8(a) config c4: 52 M non-zeros x 200 columns), for which the reference formulation would
 * materialise two 41.6 GB temporaries, can still be checked on the host in seconds.
 * PARITY UNPINNED by the reference (see oracle/gcn_oracle.py header); pinned inside this repo
 * against gcn_oracle.propagate on every size that formulation can run.
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC -o oracle/_build/liboracle_csr.so oracle/csr_spmm.c
 */
#include <stdint.h>
#include <stddef.h>

/* y[r, 0:F] = bias + sum_{j in [rowptr[r], rowptr[r+1])} val[j] * x[col[j], 0:F]; fp32 accumulate,
 * in CSR order (the order the oracle's stable sort by target leaves the edges in). */
void oracle_csr_spmm_f32(int64_t n_rows, const int64_t *rowptr, const int32_t *col, const float *val,
                         const float *x, int64_t ldx, int32_t F, const float *bias, float *y,
                         int64_t ldy)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rows; ++r) {
        float *yr = y + r * ldy;
        for (int32_t f = 0; f < F; ++f) yr[f] = 0.0f;
        for (int64_t j = rowptr[r]; j < rowptr[r + 1]; ++j) {
            const float v = val[j];
            const float *xr = x + (int64_t)col[j] * ldx;
            for (int32_t f = 0; f < F; ++f) yr[f] += v * xr[f];
        }
        if (bias)
            for (int32_t f = 0; f < F; ++f) yr[f] += bias[f];
    }
}

/* Same sum with float64 accumulators, rounded once: the tight reference for tolerance checks. */
void oracle_csr_spmm_f64acc(int64_t n_rows, const int64_t *rowptr, const int32_t *col,
                            const float *val, const float *x, int64_t ldx, int32_t F,
                            const float *bias, float *y, int64_t ldy)
{
#pragma omp parallel
    {
        double acc[1024];
#pragma omp for schedule(dynamic, 64)
        for (int64_t r = 0; r < n_rows; ++r) {
            for (int32_t f0 = 0; f0 < F; f0 += 1024) {
                const int32_t fn = (F - f0) < 1024 ? (F - f0) : 1024;
                for (int32_t f = 0; f < fn; ++f) acc[f] = 0.0;
                for (int64_t j = rowptr[r]; j < rowptr[r + 1]; ++j) {
                    const double v = (double)val[j];
                    const float *xr = x + (int64_t)col[j] * ldx + f0;
                    for (int32_t f = 0; f < fn; ++f) acc[f] += v * (double)xr[f];
                }
                float *yr = y + r * ldy + f0;
                for (int32_t f = 0; f < fn; ++f)
                    yr[f] = (float)(acc[f] + (bias ? (double)bias[f0 + f] : 0.0));
            }
        }
    }
}

/* Column sums (the bias gradient db = sum over rows of G), float64 accumulate. */
void oracle_colsum_f64acc(int64_t n_rows, int32_t F, const float *g, int64_t ldg, float *out)
{
    for (int32_t f = 0; f < F; ++f) {
        double s = 0.0;
        for (int64_t r = 0; r < n_rows; ++r) s += (double)g[r * ldg + f];
        out[f] = (float)s;
    }
}
