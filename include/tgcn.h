/* tgcn.h -- C ABI of the MI355X-native TextGCN hot path (libtgcn.so, HIP, gfx950 only).
 *
 * The reference has no FFI on this path: the arithmetic is PyG-1.6.3 `GCNConv`, reached from
 * textgcn/lib/models.py:20 `layer(x, g.edge_index, g.edge_attr)` and differentiated by autograd at
 * flat_amazon.py:105 `loss.backward()`.  This header is the boundary a maintainer binds instead
 * (ctypes stub in INTEGRATION.md); each entry point names the reference step it replaces.
 *
 * Conventions
 *   - plain C, `extern "C"`, no torch types; every data pointer is a DEVICE pointer on the plan's
 *     device unless the comment says "host"; the caller owns every buffer it passes, the library
 *     owns only the opaque plan (which holds its own copies of the graph).
 *   - every function returns a status: 0 ok, <0 error (TGCN_E_*); tgcn_last_error() returns a
 *     thread-local, human readable message for the last failing call on this thread.  Nothing
 *     throws, nothing aborts the process.
 *   - compute entry points (tgcn_spmm, tgcn_colsum) only ENQUEUE on the caller's stream,
 *     never allocate and never synchronise (hipGraph-capturable).  tgcn_plan_create allocates and
 *     synchronises `stream` (it is a one-off per graph).
 *   - a plan is immutable after creation: safe to share between threads and streams as long as
 *     concurrent calls use distinct workspaces.
 *   - all arithmetic is IEEE fp32; results are bitwise reproducible run to run (no floating-point atomics;
 *     the word-word graph builder uses integer atomics, whose sums do not depend on the order).  The one
 *     exception is opt-in: tgcn_set_gemm_split(1) forms the products of three dense kernels from exact bf16
 *     splits of their fp32 operands (fp32-accurate and still reproducible, but not the fp32 FMA chain).
 */
#ifndef TGCN_H_
#define TGCN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGCN_ABI_VERSION 6

enum {
    TGCN_OK = 0,
    TGCN_E_INVALID = -1,   /* bad argument (null pointer, negative size, misaligned, F <= 0 ...) */
    TGCN_E_RANGE = -2,     /* edge_index entry outside [0, n_nodes) or size beyond int32 limits   */
    TGCN_E_HIP = -3,       /* a HIP runtime call failed (message carries hipGetErrorString)       */
    TGCN_E_NOMEM = -4,     /* device or host allocation failed                                    */
    TGCN_E_WORKSPACE = -5  /* workspace smaller than tgcn_*_workspace_bytes()                     */
};

typedef struct tgcn_plan tgcn_plan;   /* opaque */
typedef void *tgcn_stream;            /* a hipStream_t; NULL = the null stream */

int tgcn_abi_version(void);
const char *tgcn_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * tgcn_plan_create -- replaces PyG-1.6.3 gcn_norm (k1-k4 of SURVEY.md 2a), which the reference
 * re-runs inside every GCNConv call (models.py:11-15 never pass cached=True): drop existing
 * self-loops and re-append one per node (kept weight, else 1.0 -- add_remaining_self_loops, last
 * duplicate wins), in-degree at the TARGET `dst = edge_index[1]` incl. the loop, d^-1/2 with inf->0,
 * w_hat = d^-1/2[src] * w * d^-1/2[dst].  The result is kept as M (M[dst,src] = w_hat) in CSR sorted
 * by (dst, src) -- and M^T likewise unless M == M^T bitwise -- with (col,val) interleaved as 8-byte
 * pairs, plus the work-item partition the SpMM kernel walks.
 *
 *   n_nodes, n_edges      N and E = edge_index.shape[1]
 *   src, src_stride       edge_index[0] (int64) and its element stride: the reference passes the
 *   dst, dst_stride       non-contiguous view `coo.T` (text2graph.py:192), so strides are honoured
 *   w                     edge_attr (fp32, contiguous) or NULL for all-ones
 *   add_self_loops        GCNConv(add_self_loops=...) -- the reference always passes True; the value
 *                         is also the weight of an added loop: 1 -> 1.0, 2 -> 2.0 (improved=True)
 *   normalize             GCNConv(normalize=...)      -- the reference keeps the default True; as in
 *                         PyG, loops are added inside gcn_norm, i.e. only when normalize != 0.  The value also
 *                         selects how a node's degree is summed (enum below):
 *                           TGCN_NORM_ACCURATE (1)   float64 sum, rounded to fp32 once (the default of the Python layer);
 *                                                    w_hat = w * (d^-1/2[src] * d^-1/2[dst]): a symmetric graph stays
 *                                                    BITWISE symmetric and M^T is not stored;
 *                           TGCN_NORM_REFERENCE (2)  the bits of the reference's CPU path: ONE fp32 accumulator per node,
 *                                                    weights added sequentially in edge order, the self loop last (PyG
 *                                                    scatter_add on the CPU after add_remaining_self_loops), and PyG's
 *                                                    association (d^-1/2[src] * w) * d^-1/2[dst] -- on hub nodes with
 *                                                    ~10^6 edges this differs from the accurate sum by a few 1e-5 relative.
 *   row_begin, row_end    rows [row_begin,row_end) of M and of M^T that this plan will produce
 *                         (1-D row partition across GPUs); 0, n_nodes for the whole graph.  The
 *                         normalisation is always computed over the WHOLE edge list.
 *   device                HIP device ordinal that owns every pointer
 */
enum { TGCN_NORM_OFF = 0, TGCN_NORM_ACCURATE = 1, TGCN_NORM_REFERENCE = 2 };   /* `normalize` of tgcn_plan_create */
enum { TGCN_DEGREE_ACCURATE = 0, TGCN_DEGREE_REFERENCE = 1 };                  /* `degree_sum` of tgcn_gcn_norm    */
int tgcn_plan_create(int64_t n_nodes, int64_t n_edges,
                     const int64_t *src, int64_t src_stride,
                     const int64_t *dst, int64_t dst_stride,
                     const float *w, int add_self_loops, int normalize,
                     int64_t row_begin, int64_t row_end,
                     int device, tgcn_stream stream, tgcn_plan **out);

/* tgcn_plan_create_coo -- a plan from explicit triplets M[row[i], col[i]] = val[i] (n_rows x n_cols,
 * no loops added, no normalisation, duplicates add up).  It has no counterpart in the reference,
 * which is single-device (flat_amazon.py:84-86): this is what the 1-D partition across GPUs builds
 * its per-rank local operators from (pytextgcn_amd/sharded.py).  `val` NULL = ones.
 * with_transpose = 0 keeps only M (tgcn_spmm(transpose=1) then fails with TGCN_E_INVALID). */
int tgcn_plan_create_coo(int64_t n_rows, int64_t n_cols, int64_t nnz,
                         const int64_t *row, const int64_t *col, const float *val,
                         int with_transpose, int device, tgcn_stream stream, tgcn_plan **out);

/* tgcn_gcn_norm -- the normalisation half of tgcn_plan_create on its own (PyG-1.6.3 gcn_norm as the
 * reference runs it, models.py:11-20): for every node dis[n] = deg[n]^-1/2 (inf -> 0) with deg the weighted
 * in-degree at the target incl. the self loop, and loop_w[n] = the weight of that loop (the last input loop's,
 * else add_self_loops as 1.0 / 2.0; all zeros when add_self_loops = 0).  The normalised weight of edge e is
 * then w[e] * (dis[src] * dis[dst]) -- (dis[src] * w[e]) * dis[dst] in the reference-order mode -- the association
 * tgcn_plan_create uses.  It IS the routine tgcn_plan_create runs (degree_sum = TGCN_DEGREE_ACCURATE / _REFERENCE as
 * its normalize = TGCN_NORM_ACCURATE / _REFERENCE): same chunking, same sums, same bits.  No counterpart in the single-
 * device reference: the 1-D partition (pytextgcn_amd/sharded.py) calls it so that every rank can cut its own
 * local operators out of the edge list WITHOUT building the whole-graph plan -- the edge list is walked in
 * chunks (TGCN_NORM_CHUNK edges, default 2^25), scratch is O(chunk) + O(n_nodes); deterministic.
 *   dis, loop_w   fp32 [n_nodes] device buffers (loop_w may be NULL)
 * Allocates scratch and synchronises `stream` (one-off per graph, like tgcn_plan_create). */
int tgcn_gcn_norm(int64_t n_nodes, int64_t n_edges,
                  const int64_t *src, int64_t src_stride,
                  const int64_t *dst, int64_t dst_stride,
                  const float *w, int add_self_loops, int degree_sum, float *dis, float *loop_w,
                  int device, tgcn_stream stream);

int tgcn_plan_destroy(tgcn_plan *plan);

/* tgcn_plan_query -- integers describing a plan (host pointer `out`). */
enum {
    TGCN_Q_N_NODES = 0,      /* N (= number of columns of M and M^T)                              */
    TGCN_Q_N_ROWS = 1,       /* row_end - row_begin                                               */
    TGCN_Q_NNZ = 2,          /* stored non-zeros of the forward block (E - loops_in_input + N ...) */
    TGCN_Q_NNZ_T = 3,        /* ... of the transposed block                                       */
    TGCN_Q_SYMMETRIC = 4,    /* 1 when M == M^T bitwise (the transposed copy is then not kept)    */
    TGCN_Q_ITEMS = 5,        /* work items (wavefront tasks) of the forward block                 */
    TGCN_Q_ITEMS_T = 6,
    TGCN_Q_LONG_ROWS = 7,    /* rows split into segments (second-pass reduce), forward block      */
    TGCN_Q_LONG_ROWS_T = 8,
    TGCN_Q_SEGMENTS = 9,     /* carry rows written per SpMM, forward block                        */
    TGCN_Q_SEGMENTS_T = 10,
    TGCN_Q_DEVICE_BYTES = 11,/* device memory held by the plan                                    */
    TGCN_Q_ROW_BEGIN = 12,
    TGCN_Q_HAS_TRANSPOSE = 13,/* 0 for a tgcn_plan_create_coo(with_transpose = 0) plan                */
    TGCN_Q_N_ROWS_T = 14,    /* rows of the transposed block (= columns of the forward block)      */
    TGCN_Q_HOT_ROWS = 15,    /* rows computed by the dense hot block (0 or up to 32), forward block  */
    TGCN_Q_HOT_ROWS_T = 16
};
int tgcn_plan_query(const tgcn_plan *plan, int what, int64_t *out);

/* tgcn_plan_export -- copy the stored CSR out (device buffers sized from tgcn_plan_query); any
 * pointer may be NULL.  For tests and for the INTEGRATION.md binding; not on the hot path. */
int tgcn_plan_export(const tgcn_plan *plan, int transpose, int32_t *rowptr /* n_rows+1 */,
                     int32_t *col /* nnz */, float *val /* nnz */, tgcn_stream stream);

/* ---------------------------------------------------------------------------------------------
 * tgcn_spmm -- replaces MessagePassing.propagate + bias of GCNConv.forward (k6-k9: index_select,
 * scale, scatter_add, `out += bias`) when transpose = 0, and its autograd (b1-b4: dXW = M^T dOut)
 * when transpose = 1:
 *       Y[r, 0:F] = sum_j  M(^T)[row_begin + r, j] * X[j, 0:F]  (+ bias[0:F])
 *   X    [n_nodes, F] fp32, row stride ldx (elements)        -- the gathered operand
 *   Y    [n_rows,  F] fp32, row stride ldy                    -- fully overwritten
 *   bias [F] or NULL
 * Fast path: F, ldx, ldy multiples of 4 and X, Y, bias 16-byte aligned; anything else takes the
 * scalar-lane kernel (same results).  Nothing of size nnz x F is ever written.
 * Plans of graphs whose 32 longest rows are nearly dense (TGCN_Q_HOT_ROWS > 0) compute those rows as a
 * dense product over all columns: with a non-finite value in X[c] they become non-finite even where
 * M[row, c] == 0 (0 * inf).  Finite operands give the reference's result; TGCN_HOT_ROWS=0 in the
 * environment at plan creation disables the dense block.
 */
size_t tgcn_spmm_workspace_bytes(const tgcn_plan *plan, int transpose, int F);
int tgcn_spmm(const tgcn_plan *plan, int transpose, const float *X, int64_t ldx, int F,
              const float *bias, float *Y, int64_t ldy, void *workspace, size_t workspace_bytes,
              tgcn_stream stream);

/* tgcn_spmm_split -- tgcn_spmm whose gathered operand lives in two buffers: rows (= columns of the
 * operator) [0, split) in X, rows [split, n_cols) in X2 (row r of X2 is column split + r).  No
 * counterpart in the reference (single device); the 1-D partition keeps the all-gathered hub block
 * and the rank's own rows apart and so avoids one copy per SpMM.  X2 = NULL is tgcn_spmm. */
int tgcn_spmm_split(const tgcn_plan *plan, int transpose, const float *X, int64_t ldx, const float *X2,
                    int64_t ldx2, int64_t split, int F, const float *bias, float *Y, int64_t ldy,
                    void *workspace, size_t workspace_bytes, tgcn_stream stream);

/* tgcn_spmm_acc -- the ACCUMULATE form of tgcn_spmm_split:  Y[r, 0:F] += sum_j M(^T)[r, j] * X[j, 0:F]  for every row r
 * that holds at least one stored entry; a row without entries is neither read nor written, and there is no bias.  No
 * counterpart in the single-device reference: the 1-D partition of a graph WITHOUT hub structure (BASELINE config 5) cuts a
 * rank's operator by the origin of its columns into blocks and runs block k -- accumulating into the same Y -- as soon as
 * the operand rows of exchange stage k have landed, so that stage k + 1 travels under the compute of stage k
 * (pytextgcn_amd/sharded.py, `exchange="pipeline"`; SURVEY.md 8(e) "run local part while halo is in flight").  A row is owned
 * by one wavefront and launches on one stream are ordered, so the sums are reproducible run to run (their association
 * differs from the one-launch product's: block partial sums are added in launch order).  X2 = NULL: single operand.
 * Workspace: tgcn_spmm_workspace_bytes(plan, transpose, F). */
int tgcn_spmm_acc(const tgcn_plan *plan, int transpose, const float *X, int64_t ldx, const float *X2, int64_t ldx2,
                  int64_t split, int F, float *Y, int64_t ldy, void *workspace, size_t workspace_bytes,
                  tgcn_stream stream);

/* tgcn_spmm_adam -- tgcn_spmm whose result rows are never stored: row r of M(^T) @ G is the GRADIENT of row r of
 * `param` and is spent at once on torch.optim.Adam's update of that row (the arithmetic of tgcn_adam_step, op for
 * op; max_exp_avg_sq = NULL means amsgrad=False).  For TextGCN's one-hot features the first layer's weight
 * gradient IS such a product, dW1 = M^T dH1 (the reference computes it at
 * flat_amazon.py:105 and applies it at :106): fusing the two removes the N x h gradient and the optimizer's own pass
 * over W1 and its state.  Same bits as tgcn_spmm followed by tgcn_adam_step.
 *   G [n_cols, F] fp32 stride ldg;  param / exp_avg / exp_avg_sq / max_exp_avg_sq [n_rows, F] stride ldp
 *   F %% 4 == 0, strides multiples of 4, 16-byte aligned buffers (no scalar fallback: TGCN_E_INVALID otherwise)
 *   step  1-based step count AFTER increment; with scalars_dev != NULL the two step-dependent factors are read
 *         from that device buffer instead ({lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)}, as tgcn_adam_step_capturable
 *         leaves them in its `scalars_dev`), so that a captured HIP graph can be replayed per step.
 * Workspace: tgcn_spmm_workspace_bytes(plan, transpose, F). */
int tgcn_spmm_adam(const tgcn_plan *plan, int transpose, const float *G, int64_t ldg, int F, float *param,
                   float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq, int64_t ldp, double lr,
                   double beta1, double beta2, double eps, double weight_decay, int64_t step,
                   const float *scalars_dev, void *workspace, size_t workspace_bytes, tgcn_stream stream);

/* tgcn_spmm_adam_split -- tgcn_spmm_adam whose gathered operand lives in two buffers, as tgcn_spmm_split's does: rows
 * [0, split) in G, rows [split, n_cols) in G2.  For the 1-D partition's backward SpMM: the regular rows of a rank's W1
 * shard are updated inside the launch that computes their gradient from the gathered hub block and the rank's own rows.
 * G2 = NULL is tgcn_spmm_adam. */
int tgcn_spmm_adam_split(const tgcn_plan *plan, int transpose, const float *G, int64_t ldg, const float *G2, int64_t ldg2,
                         int64_t split, int F, float *param, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq,
                         int64_t ldp, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                         const float *scalars_dev, void *workspace, size_t workspace_bytes, tgcn_stream stream);

/* Row movement of the multi-GPU exchange (pytextgcn_amd/sharded.py; nothing in the reference corresponds: it is
 * single-device, flat_amazon.py:84-86).  Row-major fp32 rows, int64 row indices on the device, enqueue only.
 *   tgcn_rows_gather          out[i, :]    = x[idx[i], :]     (pack the hub rows a peer references)
 *   tgcn_rows_scatter         y[idx[i], :] = x[i, :]          (place received rows in the gathered block; idx distinct)
 *   tgcn_rows_reduce_ranked   y[row0 + j * row_step, :] += sum over q = 0 .. n_ranks-1, IN THAT ORDER, of
 *                             recv[inv[q * n + j], :]  (inv < 0: rank q sent nothing for row j), j = 0 .. n-1:
 *                             the reduce-scatter's local sum -- starting from zero, ranks in order, one add into y:
 *                             the summation order all exchange forms share, so that they agree bit for bit.
 * n_x_rows / n_y_rows / n_recv_rows are the row counts of the INDEXED buffers: an index outside them is skipped on the
 * device (the calls only enqueue and cannot report it), so a bad list can never read or write outside a buffer;
 * tgcn_rows_reduce_ranked returns TGCN_E_RANGE when row0 + (n - 1) * row_step >= n_y_rows. */
int tgcn_rows_gather(const float *x, int64_t ldx, int64_t n_x_rows, const int64_t *idx, int64_t n, int F, float *out,
                     int64_t ldo, tgcn_stream stream);
int tgcn_rows_scatter(const float *x, int64_t ldx, const int64_t *idx, int64_t n, int F, float *y, int64_t ldy,
                      int64_t n_y_rows, tgcn_stream stream);
int tgcn_rows_reduce_ranked(const float *recv, int64_t ldr, int64_t n_recv_rows, const int32_t *inv, int n_ranks,
                            int64_t n, int F, float *y, int64_t ldy, int64_t n_y_rows, int64_t row0, int64_t row_step,
                            tgcn_stream stream);

/* tgcn_colsum -- replaces the autograd of `out += bias` (db = sum over rows of dOut).
 *   G [n_rows, F] fp32 stride ldg -> out [F]. */
size_t tgcn_colsum_workspace_bytes(int64_t n_rows, int F);
int tgcn_colsum(const float *G, int64_t ldg, int64_t n_rows, int F, float *out, void *workspace,
                size_t workspace_bytes, tgcn_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Training-step helpers (the caller of the path: flat_amazon.py:82,89,99-106).
 *
 * tgcn_masked_ce -- `CrossEntropyLoss(reduction='mean')(logits[mask], target[mask])`
 * (flat_amazon.py:82,101-102) and, when dlogits != NULL, its gradient w.r.t. the FULL logits matrix
 * (rows outside the mask get zeros), in one pass.
 *   logits [n_rows, n_classes] fp32 stride ld;  target int64 [n_rows];  mask uint8/bool [n_rows]
 *   inv_count = 1 / (number of rows in the mask) -- a host value: masks are static, count them once
 *   loss      device scalar (fp32);  dlogits [n_rows, n_classes] stride ldd or NULL
 */
size_t tgcn_masked_ce_workspace_bytes(void);
int tgcn_masked_ce(const float *logits, int64_t ld, int64_t n_rows, int n_classes,
                   const int64_t *target, const uint8_t *mask, float inv_count, float *loss,
                   float *dlogits, int64_t ldd, void *workspace, size_t workspace_bytes,
                   tgcn_stream stream);
/* tgcn_masked_ce_pred -- the same pass also writes pred[r] = argmax_c logits[r, c] (first index on
 * ties) for EVERY row when pred != NULL: the predictions the reference takes on the host with
 * `np.argmax(logits[mask].cpu().numpy(), axis=1)` (flat_amazon.py:111-114), so that `pred[mask]` (8
 * bytes per row) is what crosses PCIe instead of the masked logits. */
int tgcn_masked_ce_pred(const float *logits, int64_t ld, int64_t n_rows, int n_classes,
                        const int64_t *target, const uint8_t *mask, float inv_count, float *loss,
                        float *dlogits, int64_t ldd, int64_t *pred, void *workspace,
                        size_t workspace_bytes, tgcn_stream stream);

/* tgcn_masked_ce_grad -- tgcn_masked_ce_pred that also returns dbias[c] = sum_r dlogits[r, c]: the gradient of
 * the bias of the layer that produced the logits (the autograd of `out += bias` in the last GCNConv, triggered
 * at flat_amazon.py:105), summed from the registers that hold each gradient row instead of by a second pass
 * over dlogits (tgcn_colsum).  Fixed summation order, no atomics.  dlogits and dbias [n_classes] are required;
 * pred may be NULL. */
size_t tgcn_masked_ce_grad_workspace_bytes(int64_t n_rows, int n_classes);
int tgcn_masked_ce_grad(const float *logits, int64_t ld, int64_t n_rows, int n_classes,
                        const int64_t *target, const uint8_t *mask, float inv_count, float *loss,
                        float *dlogits, int64_t ldd, float *dbias, int64_t *pred, void *workspace,
                        size_t workspace_bytes, tgcn_stream stream);
/* tgcn_scale_by_device_scalar -- x[0..n) *= *scale_dev, skipped on the device when the scalar is exactly 1
 * (multiplying by 1.0f is the identity): the chain rule through the loss node, whose incoming gradient is a
 * device scalar the host cannot read without a synchronisation (`loss.backward()` seeds it with 1). */
int tgcn_scale_by_device_scalar(float *x, int64_t n, const float *scale_dev, tgcn_stream stream);

/* tgcn_adam_step -- one `torch.optim.Adam(..., amsgrad=...)` update of a flat fp32 tensor
 * (flat_amazon.py:89,106), torch's single-tensor formula op for op, fused into one pass.
 * max_exp_avg_sq = NULL means amsgrad=False.  `step` is the 1-based step count AFTER increment. */
int tgcn_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                   float *max_exp_avg_sq, int64_t n, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int64_t step, tgcn_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Tall-skinny fp32 GEMMs on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32): the dense
 * `torch.matmul(x, self.weight)` of PyG-1.6.3 GCNConv.forward (reference call site models.py:20,
 * layers built at models.py:13,15) and its autograd.  N = number of nodes; k, n = layer widths, any value >= 1.
 * The small operand lives in LDS: one launch when k_pad * n_pad * 4 bytes <= 160 KB (k_pad = k rounded up to 8,
 * n_pad = n rounded up to 32; e.g. 200 x 64, 64 x 200), otherwise the product runs as column groups of <= 128 result
 * columns and k chunks of <= 256 that accumulate into C (e.g. DBpedia's 219 classes at hidden width 200, flat_dbpedia.py:80:
 * two groups) -- the tall operand is then read once per group.
 *   tgcn_gemm_nn   C[N,n] = A[N,k] @ B[k,n]        (XW  = H @ W)        A: lda % 4 == 0, 16-B aligned
 *   tgcn_gemm_nt   C[N,n] = A[N,k] @ B[n,k]^T      (dH  = dXW @ W^T)    A: lda % 4 == 0, 16-B aligned
 *   tgcn_gemm_tn   C[k,n] = A[N,k]^T @ G[N,n]      (dW  = H^T @ dXW)    deterministic; one launch covers <= 256 columns
 *                                                                        of A x <= 128 of G, wider ones run in chunks
 */
int tgcn_gemm_nn(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                 int64_t N, int k, int n, tgcn_stream stream);
int tgcn_gemm_nt(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                 int64_t N, int k, int n, tgcn_stream stream);
size_t tgcn_gemm_tn_workspace_bytes(int64_t N, int k, int n);
int tgcn_gemm_tn(const float *A, int64_t lda, const float *G, int64_t ldg, float *C, int64_t ldc,
                 int64_t N, int k, int n, void *workspace, size_t workspace_bytes, tgcn_stream stream);

/* The same products with the inverted dropout of the reference's GCN.forward (textgcn/lib/models.py:23,
 * `x = self.dropout(x)` between the layers) fused in, so that the dropped activation and its mask are
 * never stored: element (r, c) of the masked [N x w] activation is kept iff hash(seed, r, c) >=
 * p * 2^32 and scaled by 1 / (1 - p); all three regenerate the same mask from `seed` (8 bytes in DEVICE
 * memory, read by the kernels: safe under HIP-graph capture).
 *   tgcn_gemm_nn_dropout   C = dropout(A) @ B          mask over A [N x k]   (XW = dropout(H) @ W)
 *   tgcn_gemm_tn_dropout   C = dropout(A)^T @ G        mask over A [N x k]   (dW = dropout(H)^T @ dXW)
 *   tgcn_gemm_nt_dropout   C = dropout'(A @ B^T)       mask over C [N x n]   (dH = mask * (dXW @ W^T))
 * 0 <= p <= 1 (p = 1 yields zeros).  The random stream is this library's own, not torch's. */
int tgcn_gemm_nn_dropout(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                         int64_t N, int k, int n, double p, const uint64_t *seed, tgcn_stream stream);
int tgcn_gemm_nt_dropout(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                         int64_t N, int k, int n, double p, const uint64_t *seed, tgcn_stream stream);
int tgcn_gemm_tn_dropout(const float *A, int64_t lda, const float *G, int64_t ldg, float *C, int64_t ldc,
                         int64_t N, int k, int n, double p, const uint64_t *seed, void *workspace,
                         size_t workspace_bytes, tgcn_stream stream);

/* The same two products with the keep decisions RECORDED instead of hashed twice: the nn product (forward) writes the mask
 * of its A operand, 4 bits per lane step, `tgcn_dropout_mask_words(k, n)` 32-bit words per row (the caller allocates
 * [N x mask_stride] words; 0 = a product of this shape cannot record it -- reductions longer than 256, the split-bf16 mode -- use the
 * entry points above); the tn product (weight gradient) reads it back: one load and four bit tests per 16 bytes of A in
 * place of four hashes.  Bit for bit the results of tgcn_gemm_nn_dropout / tgcn_gemm_tn_dropout with the same seed (the
 * tn side falls back on the hash wherever its kernel does not take the record).  Layout: column c of a row is bit 4 ((c / 8) % 8) + (c & 3) of word
 * ((c / 4) & 1) * (words / 2) + c / 64. */
size_t tgcn_dropout_mask_words(int k, int n);
int tgcn_gemm_nn_dropout_mask(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                              int64_t N, int k, int n, double p, const uint64_t *seed, uint32_t *mask,
                              int64_t mask_stride, tgcn_stream stream);
int tgcn_gemm_tn_dropout_mask(const float *A, int64_t lda, const float *G, int64_t ldg, float *C, int64_t ldc,
                              int64_t N, int k, int n, double p, const uint64_t *seed, const uint32_t *mask,
                              int64_t mask_stride, void *workspace, size_t workspace_bytes, tgcn_stream stream);
/* ... and the input-gradient product with the column sums, its mask (over the [N x n] RESULT: the same matrix the forward
 * product masked) read from that record instead of hashed: bit for bit tgcn_gemm_nt_colsum with the same seed (the kernel
 * falls back on the hash for shapes whose record it does not take). */
int tgcn_gemm_nt_colsum_mask(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                             int64_t N, int k, int n, double p, const uint64_t *seed, const uint32_t *mask,
                             int64_t mask_stride, float *colsum, void *workspace, size_t workspace_bytes,
                             tgcn_stream stream);
/* tgcn_set_dropout_row_keys -- which ROW of the dropout mask a row of the masked matrix is, for every tgcn_gemm_*_dropout*
 * / tgcn_gemm_nt_colsum* call this THREAD makes from now on: row i of the call's [N x ld] matrix takes the keep decisions
 * of mask row i + key0 when i < split, of mask row i + key1 otherwise (the mask is a stateless hash of (seed, mask row,
 * column)).  (0, 0, 0) -- the state of a new thread -- is the row index itself.  No reference counterpart (F.dropout at
 * textgcn/lib/models.py:23 draws an unkeyed mask over the whole activation); it exists for the 1-D partition: a hub
 * (word) row's mask must be the same function of its position in the gathered hub block on the rank that owns the row
 * (its local rows [0, hp) are mask rows rank * hp + i) and on every rank that holds a partial sum of it, so that
 * dropout(sum_q P_q) @ W2 = sum_q dropout(P_q) @ W2 can be exchanged at the class width (pytextgcn_amd/sharded.py,
 * `narrow_exchange`).  The recorded-mask entry points record / read decisions per matrix row, whatever the keys. */
int tgcn_set_dropout_row_keys(int64_t split, int64_t key0, int64_t key1);

/* tgcn_set_gemm_split -- library-wide numerical mode of tgcn_gemm_nn / _nt / _tn for the
 * shapes of the GCN layers (nn: k = 200, 33 <= n <= 64; nt: k = 64, 193 <= n <= 224; tn: k = 200, 33 <= n <= 64,
 * contiguous operands; + the _dropout forms and, for nt, the _colsum form).  Every other shape keeps the fp32 kernels.  on != 0: every fp32 product is
 * formed from an exact three-way bf16 split of both operands on the bf16 matrix cores (six partial products,
 * fp32 accumulation, dropped terms <= 2^-23 relative): fp32-accurate, NOT the fp32 FMA chain bit for bit, +-inf
 * operands give nan.  Default off (TGCN_GEMM_SPLIT=1 in the environment turns it on at load).  Returns the
 * previous setting. */
int tgcn_set_gemm_split(int on);

/* tgcn_gemm_nt_colsum -- tgcn_gemm_nt (seed == NULL) or tgcn_gemm_nt_dropout (seed != NULL) that also returns
 * colsum[j] = sum_r C[r, j], summed from the accumulator tiles as they are stored: with C = dH1 (the gradient of
 * the first layer's output) this is that layer's bias gradient (the autograd of `out += bias`, triggered at
 * flat_amazon.py:105), which otherwise costs a second pass over the N x h matrix (tgcn_colsum).  Fixed summation
 * order, no atomics. */
size_t tgcn_gemm_nt_colsum_workspace_bytes(int n);
int tgcn_gemm_nt_colsum(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                        int64_t N, int k, int n, double p, const uint64_t *seed, float *colsum, void *workspace,
                        size_t workspace_bytes, tgcn_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Word-word PMI edges (graph construction).  Replaces the reference's Cython
 * entry point `compute_word_word_edges(X, n_vocab, n_documents, seq_len, window_size, n_jobs, verbose)`
 * (textgcn/lib/clib/graphbuilder.pyx:23-25, called at text2graph.py:156-160) and its test hook
 * `sliding_window_tester` (:263-275).  Results are bit-identical to the reference: same uint32
 * counts, same edge order ((i,j),(j,i) interleaved, upper triangle row-major), same float32 PMI.
 *   X   int32 [n_docs, seq_len] row-major DEVICE pointer, tokens in [0, n_vocab), -1 = padding
 * Memory: the reference keeps the counts in a dense V(V+1)/2 array (graphbuilder.pyx:44,134) and wraps its uint32 index
 * beyond V = 65 535 (:250).  Here the dense triangle serves while it takes <= 16 GiB (V <= ~92 000); larger vocabularies
 * take the SPARSE counter -- (i << 32 | j, count) records of bounded chunks of documents, sorted, summed by key and merged
 * into one sorted list of distinct pairs: O(chunk + distinct pairs) memory, the same counts, edges, order and weights.
 * Environment (read at create): TGCN_WW_COUNTER=dense|sparse pins the counter, TGCN_WW_CHUNK_PAIRS the records per chunk
 * (default 2^27).  With the sparse counter the `cij` export builds the triangle on demand (small vocabularies only).
 * The handle owns its outputs (the reference leaks its malloc'd arrays, :65-66); export copies
 * them to host or device buffers (hipMemcpyDefault) and synchronises the stream.
 */
typedef struct tgcn_wwedges tgcn_wwedges;
enum { TGCN_WW_N_EDGES = 0, TGCN_WW_N_WINDOWS = 1, TGCN_WW_N_COUNTS = 2 /* n_vocab*(n_vocab+1)/2 */,
       TGCN_WW_SPARSE = 3 /* 1: the sorted-pair-list counter ran (large vocabularies), 0: the dense triangle */,
       TGCN_WW_N_PAIRS = 4 /* distinct pairs i <= j with a count (sparse counter; -1 for the dense one) */ };
int tgcn_wwedges_create(const int32_t *X, int64_t n_docs, int64_t seq_len, int64_t n_vocab,
                        int64_t window, int device, tgcn_stream stream, tgcn_wwedges **out);
int tgcn_wwedges_query(const tgcn_wwedges *we, int what, int64_t *out);
int tgcn_wwedges_export(const tgcn_wwedges *we, int32_t *coo /* [n_edges][2] */,
                        float *weights /* [n_edges] */, uint32_t *cij /* packed triangle or NULL */,
                        tgcn_stream stream);
int tgcn_wwedges_destroy(tgcn_wwedges *we);

/* tgcn_adam_step_capturable -- the same update with the step count on the DEVICE (torch's
 * `capturable=True`): `step_dev` (int64) is incremented by the call and `scalars_dev` (2 floats of
 * scratch) receives the step-dependent factors, so a captured HIP graph can be replayed per step. */
int tgcn_adam_step_capturable(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                              float *max_exp_avg_sq, int64_t n, double lr, double beta1, double beta2,
                              double eps, double weight_decay, int64_t *step_dev, float *scalars_dev,
                              tgcn_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* TGCN_H_ */
