// Internal declarations shared by the translation units of libtgcn.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "tgcn.h"

namespace tgcn {

// thread-local message behind tgcn_last_error()
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define TGCN_HIP_CHECK(expr)                                                                    \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            ::tgcn::set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,         \
                              __LINE__);                                                        \
            return e_ == hipErrorOutOfMemory ? TGCN_E_NOMEM : TGCN_E_HIP;                       \
        }                                                                                       \
    } while (0)

#define TGCN_CHECK(call)                                                                        \
    do {                                                                                        \
        int s_ = (call);                                                                        \
        if (s_ != TGCN_OK) return s_;                                                           \
    } while (0)

// One wavefront's task in the SpMM kernel.
//   row block : rows [row_begin, row_end) complete, their non-zeros [nnz_begin, nnz_end) contiguous
//   segment   : row_end < 0; part [nnz_begin, nnz_end) of the single long row `row_begin`, whose
//               partial sum goes to carry row (-row_end - 1)
struct WorkItem {
    int32_t row_begin;
    int32_t row_end;
    int32_t nnz_begin;
    int32_t nnz_end;
};

// A long row and where its partial sums sit: carry rows [slot_begin, slot_begin + count).
struct FixEntry {
    int32_t row;
    int32_t slot_begin;
    int32_t count;
    int32_t pad;
};

// One stored operator block (rows [row_begin,row_end) of M or of M^T) in device memory.
struct CsrBlock {
    int64_t n_rows = 0;
    int64_t n_cols = 0;
    int64_t nnz = 0;
    int32_t *rowptr = nullptr;  // [n_rows + 1]
    int2 *cv = nullptr;         // [nnz] {col, bits(val)}
    WorkItem *items = nullptr;  // [n_items]
    int32_t n_items = 0;
    FixEntry *fix = nullptr;    // [n_fix]
    int32_t n_fix = 0;
    int32_t n_segments = 0;
    size_t bytes = 0;
    // Dense hot block (plan.hip: build_items; spmm.hip: k_spmm_hot).  When n_hot > 0 the lists above
    // leave the hot rows out (their fix entries point at the partial sums of k_spmm_hot, carry rows
    // [hot_slot_base + k * hot_parts, ...)) and the *_all lists hold the complete partition for the
    // scalar fallback kernel.
    int32_t n_hot = 0;
    int32_t hot_parts = 0;      // workgroups of k_spmm_hot = partial sums per hot row
    int32_t hot_cpw = 0;        // columns per wave of k_spmm_hot (even)
    int32_t hot_slot_base = 0;
    float *hot_vals = nullptr;  // [hot_parts * 8 * hot_cpw][kHotRows], zero padded
    // The rows of every row block in order of falling degree (plan.hip: build_items), as {first entry, end, row, 0}:
    // the sub-group kernels hand the rows of a block to their 4 (2) sub-groups in THIS order, so that the rows a
    // wave sums side by side have about the same length (a wave runs as many gather rounds as its longest row
    // needs); one 16-byte load replaces the two row-pointer loads, the dependent chain stays three loads long.
    int4 *row_info = nullptr;   // [n_rows]
    WorkItem *items_all = nullptr;
    int32_t n_items_all = 0;
    FixEntry *fix_all = nullptr;
    int32_t n_fix_all = 0;
    int32_t n_segments_all = 0;
};

constexpr int kHotRows = 32;    // one MFMA M-tile

void free_block(CsrBlock &b);

// Host-side partition of a block into work items (plan.hip); item_weight = target nnz+rows per item.
int build_items(CsrBlock &b, int item_weight, hipStream_t stream);

// Kernel launchers (spmm.hip / colsum.hip); arguments validated by the caller.
int launch_spmm(const CsrBlock &b, const float *X, int64_t ldx, const float *X2, int64_t ldx2, int split,
                int F, const float *bias, float *Y, int64_t ldy, float *carry, hipStream_t stream, bool acc = false);
int launch_colsum(const float *G, int64_t ldg, int64_t n_rows, int F, float *out, float *partial,
                  int n_blocks, hipStream_t stream);
int colsum_blocks(int64_t n_rows);
int launch_colsum_final(const float *partial, int n_partial, int F, float *out, hipStream_t stream);

inline int64_t round_up4(int64_t v) { return (v + 3) & ~int64_t(3); }

// One element of torch.optim.Adam's update (torch/optim/adam.py, _single_tensor_adam), shared by the
// stand-alone optimizer pass (train.hip: k_adam) and the optimizer fused into the SpMM epilogue (spmm.hip):
//   g += wd * p;  m.lerp_(g, 1 - b1);  v = b2 * v + (1 - b2) * g * g;  vmax = max(vmax, v)   [amsgrad]
//   p -= step_size * m / (sqrt(vmax or v) * inv_bc2_sqrt + eps)
// Floating-point contraction is switched OFF inside: both callers must produce the same bits for the same
// inputs (tgcn_spmm_adam == tgcn_spmm + tgcn_adam_step), whatever the surrounding code lets the compiler fuse.
__device__ __forceinline__ void adam_element(float &p, const float g, float &m, float &v, float &vmax,
                                             const bool amsgrad, const float w1, const float b2, const float w2,
                                             const float eps, const float wd, const float step_size,
                                             const float inv_bc2_sqrt) {
#pragma clang fp contract(off)
    const float gr = g + wd * p;
    const float diff = gr - m;
    m = w1 < 0.5f ? m + w1 * diff : gr - diff * (1.f - w1);   // at::lerp: monotone in the weight
    v = v * b2 + (w2 * gr) * gr;
    float d = v;
    if (amsgrad) {
        vmax = fmaxf(vmax, v);
        d = vmax;
    }
    const float denom = sqrtf(d) * inv_bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

}  // namespace tgcn

struct tgcn_plan {
    int device = 0;
    int64_t n_nodes = 0;
    int64_t row_begin = 0;
    int64_t row_end = 0;
    bool symmetric = false;
    bool has_transpose = true;
    tgcn::CsrBlock fwd;
    tgcn::CsrBlock bwd;  // unused when symmetric
};
