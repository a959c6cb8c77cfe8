// CSR SpMM  Y = M X (+ bias)  for gfx950: the message-passing step of GCNConv.
//
// Replaces, fused into one pass, what PyG-1.6.3 does at textgcn/lib/models.py:20 as index_select
// (nnz x F gather), broadcast multiply (another nnz x F) and scatter_add (k6-k8 of SURVEY.md 2a)
// plus `out += bias` (k9); called with the transposed block it is that step's autograd.
//
// Shape of the kernel (HBM-bound: ~0.5 flop/byte at F = 200):
//   * one wavefront (64 lanes) per work item; an item is ~T non-zeros: either a block of complete
//     short rows or one segment of a long row (common.h).  All waves carry equal work, so degree
//     skew costs nothing and no output row is ever shared between waves (no atomics).
//   * the wave streams its (col,val) pairs 64 at a time with one coalesced 8-byte load per lane and
//     hands them out through v_readlane: column id and weight live in SGPRs, so a gathered row is
//     one `global_load_dwordx4 v, v_off, s[base]` per lane -- lane l owns columns 4l..4l+3, 50 of
//     64 lanes busy at F = 200, the rest are clamped onto the last vector (same cache line, no
//     extra traffic) so the wave never diverges.
//   * U gathered rows are in flight per wave before the first FMA (U x 800 B at F = 200); with
//     >= 4 waves per SIMD that is > 100 KB of loads in flight per CU, enough to cover HBM latency.
//   * row sums stay in registers; a finished row is written once, with the bias added.  Long rows
//     write one partial per segment into the carry workspace and k_spmm_fix adds them in slot
//     order (LDS across the 4 waves), so the result is bitwise reproducible.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace tgcn {
namespace {

typedef float native_f4 __attribute__((ext_vector_type(4)));

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
    using type = float4;
    static __device__ __forceinline__ type load_nt(const float *p) {
        const native_f4 v = __builtin_nontemporal_load(reinterpret_cast<const native_f4 *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    static __device__ __forceinline__ void store_nt(float *p, const type &a) {
        native_f4 v = {a.x, a.y, a.z, a.w};
        __builtin_nontemporal_store(v, reinterpret_cast<native_f4 *>(p));
    }
    static __device__ __forceinline__ type zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ void fma(type &a, float v, const type &x) {
        a.x = fmaf(v, x.x, a.x);
        a.y = fmaf(v, x.y, a.y);
        a.z = fmaf(v, x.z, a.z);
        a.w = fmaf(v, x.w, a.w);
    }
    static __device__ __forceinline__ type add(const type &a, const type &b) {
        return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
};
typedef float native_f2 __attribute__((ext_vector_type(2)));
template <>
struct Vec<2> {
    using type = float2;
    static __device__ __forceinline__ type load_nt(const float *p) {
        const native_f2 v = __builtin_nontemporal_load(reinterpret_cast<const native_f2 *>(p));
        return make_float2(v.x, v.y);
    }
    static __device__ __forceinline__ void store_nt(float *p, const type &a) {
        native_f2 v = {a.x, a.y};
        __builtin_nontemporal_store(v, reinterpret_cast<native_f2 *>(p));
    }
    static __device__ __forceinline__ type zero() { return make_float2(0.f, 0.f); }
    static __device__ __forceinline__ void fma(type &a, float v, const type &x) {
        a.x = fmaf(v, x.x, a.x);
        a.y = fmaf(v, x.y, a.y);
    }
    static __device__ __forceinline__ type add(const type &a, const type &b) { return make_float2(a.x + b.x, a.y + b.y); }
};
template <>
struct Vec<1> {
    using type = float;
    static __device__ __forceinline__ type load_nt(const float *p) { return __builtin_nontemporal_load(p); }
    static __device__ __forceinline__ void store_nt(float *p, const type &a) { __builtin_nontemporal_store(a, p); }
    static __device__ __forceinline__ type zero() { return 0.f; }
    static __device__ __forceinline__ void fma(type &a, float v, const type &x) { a = fmaf(v, x, a); }
    static __device__ __forceinline__ type add(const type &a, const type &b) { return a + b; }
};

__device__ __forceinline__ int readlane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float readlane_f(int bits, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(bits, lane));
}

constexpr int kWavesPerBlock = 4;

// Optimizer fused into the epilogue (tgcn_spmm_adam): the finished row is the GRADIENT of one parameter row
// -- for one-hot features dW1 = M^T dH1 -- and instead of being written it is
// consumed on the spot by torch.optim.Adam's update of that row (the arithmetic of k_adam in train.hip, op
// for op), so that the N x h gradient is never stored and the optimizer's own pass over W1 disappears.
struct AdamRow {
    float *p, *m, *v, *vmax;     // parameter, exp_avg, exp_avg_sq, max_exp_avg_sq (nullptr: amsgrad off)
    int64_t ld;                  // common row stride (elements)
    float w1, b2, w2, eps, wd;   // 1 - beta1, beta2, 1 - beta2, eps, weight decay
    float step_size, inv_bc2_sqrt;
    const float *dev_scalars;    // {step_size, inv_bc2_sqrt} on the device (capturable form) or nullptr
};

struct AdamState4 {              // the optimizer state of one lane's four elements, in flight while the row is summed
    float4 p, m, v, x;
};

__device__ __forceinline__ void adam_prefetch(const AdamRow &ad, int64_t off, AdamState4 &st) {
    st.p = *reinterpret_cast<const float4 *>(ad.p + off);
    st.m = Vec<4>::load_nt(ad.m + off);
    st.v = Vec<4>::load_nt(ad.v + off);
    st.x = ad.vmax != nullptr ? Vec<4>::load_nt(ad.vmax + off) : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ void adam_update(const AdamRow &ad, int64_t off, const float4 &g4, AdamState4 &st,
                                            float step_size, float inv_bc2_sqrt) {
    float *P = &st.p.x, *M = &st.m.x, *V = &st.v.x, *X = &st.x.x;
    const float *G = &g4.x;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        adam_element(P[k], G[k], M[k], V[k], X[k], ad.vmax != nullptr, ad.w1, ad.b2, ad.w2, ad.eps, ad.wd, step_size,
                     inv_bc2_sqrt);
    *reinterpret_cast<float4 *>(ad.p + off) = st.p;
    Vec<4>::store_nt(ad.m + off, st.m);
    Vec<4>::store_nt(ad.v + off, st.v);
    if (ad.vmax != nullptr) Vec<4>::store_nt(ad.vmax + off, st.x);
}

// One work item on one wavefront.  POLICY bits (cache hints, chosen by measurement): 1 = non-temporal
// (col,val) stream, 2 = non-temporal result stores.  The gathered rows themselves always use plain
// loads: a non-temporal hint on them destroys the L2 / Infinity-Cache re-use (7.5 ms instead of 4.4),
// and even a per-entry "cold column" hint behind a branch cost more than it saved.
// ACC (tgcn_spmm_acc): the finished sum is ADDED to the row already in Y, and a row without stored entries is neither
// read nor written -- the column-block launches of the pipelined exchange (pytextgcn_amd/sharded.py) each touch only the
// rows they have entries for.  Rows are wave-owned and launches stream-ordered, so the result stays deterministic.
template <int VEC, int U, int POLICY, bool ADAM = false, bool ACC = false>
__device__ __forceinline__ void spmm_item(
    const WorkItem it, const int lane, const int col0, const int F,
    const int32_t *__restrict__ rowptr, const int2 *__restrict__ cv, const float *__restrict__ X,
    const int64_t ldx, const float *__restrict__ X2, const int64_t ldx2, const int split,
    const float *__restrict__ bias, float *__restrict__ Y, const int64_t ldy,
    float *__restrict__ carry, const int64_t ldc, const AdamRow ad = AdamRow{}) {
    static_assert(!ADAM || VEC == 4, "the fused optimizer epilogue is written for float4 lanes");
    static_assert(!(ADAM && ACC), "the optimizer epilogue consumes the row: nothing to accumulate into");
    using V = Vec<VEC>;
    using vec_t = typename V::type;
    const int nvec = (min(F - col0, 64 * VEC) + VEC - 1) / VEC;  // vectors in this column tile
    const bool active = lane < nvec;
    const int lc = col0 + (active ? lane : nvec - 1) * VEC;      // idle lanes shadow the last one
    const int nnz_end = it.nnz_end;
    const bool segment = it.row_end < 0;

    vec_t bvec = V::zero();
    if (bias != nullptr) bvec = *reinterpret_cast<const vec_t *>(bias + lc);

    // end offsets of the next 64 rows, one per lane (row blocks only)
    int r = it.row_begin;
    int rp_base = r;
    int rp_lane = nnz_end;
    if (!segment && r + lane < it.row_end) rp_lane = rowptr[r + 1 + lane];
    int row_end = segment ? nnz_end : readlane_i(rp_lane, 0);

    vec_t acc = V::zero();
    // split operand: columns [0, split) live in X, columns [split, ...) in X2 (the sharded path keeps
    // the gathered hub block and the rank's own rows in two buffers; split = INT_MAX otherwise)
    const float *xl = X + lc;
    const float *xl2 = X2 + lc - int64_t(split) * ldx2;

    auto store_row = [&](float *dst, const vec_t &v) {
        if (!active) return;
        if constexpr ((POLICY & 2) != 0)
            V::store_nt(dst, v);
        else
            *reinterpret_cast<vec_t *>(dst) = v;
    };
    auto load_cv = [&](int idx) -> int2 {
        if constexpr ((POLICY & 1) != 0) {
            const long long raw = __builtin_nontemporal_load(reinterpret_cast<const long long *>(cv + idx));
            return make_int2(static_cast<int>(raw), static_cast<int>(raw >> 32));
        } else {
            return cv[idx];
        }
    };
    // ADAM: the optimizer state of the row being summed is loaded when the row starts and is in flight under its
    // gathers; the finished sum (+ bias) is the gradient of that row
    AdamState4 ast;
    float a_step = ad.step_size, a_ibc2 = ad.inv_bc2_sqrt;
    if constexpr (ADAM) {
        if (ad.dev_scalars != nullptr) {
            a_step = ad.dev_scalars[0];
            a_ibc2 = ad.dev_scalars[1];
        }
        if (!segment && r < it.row_end && active) adam_prefetch(ad, int64_t(r) * ad.ld + lc, ast);
    }
    bool touched = false;        // ACC: row r has had an entry (wave-uniform: idx and row_end are)
    auto finish_row = [&]() {
        if constexpr (ADAM) {
            if (active) {
                const vec_t g = V::add(acc, bvec);
                adam_update(ad, int64_t(r) * ad.ld + lc, g, ast, a_step, a_ibc2);
                if (r + 1 < it.row_end) adam_prefetch(ad, int64_t(r + 1) * ad.ld + lc, ast);
            }
        } else if constexpr (ACC) {
            if (touched) {
                float *dst = Y + int64_t(r) * ldy + lc;
                if (active) *reinterpret_cast<vec_t *>(dst) = V::add(*reinterpret_cast<const vec_t *>(dst), acc);
            }
            touched = false;
        } else {
            store_row(Y + int64_t(r) * ldy + lc, V::add(acc, bvec));
        }
    };
    auto flush_row = [&]() {
        // row r is complete: write it (or spend it on the optimizer), move to the next one
        finish_row();
        acc = V::zero();
        ++r;
        if (r - rp_base == 64) {
            rp_base = r;
            rp_lane = nnz_end;
            if (r + lane < it.row_end) rp_lane = rowptr[r + 1 + lane];
        }
        row_end = readlane_i(rp_lane, r - rp_base);
    };

    int2 cur = make_int2(0, 0);
    if (it.nnz_begin + lane < nnz_end) cur = load_cv(it.nnz_begin + lane);
    for (int base = it.nnz_begin; base < nnz_end; base += 64) {
        const int2 mine = cur;
        cur = make_int2(0, 0);  // col 0 / weight 0: a harmless row for the padded tail
        if (base + 64 + lane < nnz_end) cur = load_cv(base + 64 + lane);
        const int n = min(64, nnz_end - base);
        for (int j0 = 0; j0 < n; j0 += U) {
            vec_t x[U];
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = readlane_i(mine.x, j0 + u);
                v[u] = readlane_f(mine.y, j0 + u);
                const float *src = c < split ? xl + int64_t(c) * ldx : xl2 + int64_t(c) * ldx2;
                x[u] = *reinterpret_cast<const vec_t *>(src);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + j0 + u;
                if (idx < nnz_end) {
                    if (!segment) {
                        while (idx == row_end) flush_row();
                    }
                    if constexpr (ACC) touched = true;
                    V::fma(acc, v[u], x[u]);
                }
            }
        }
    }
    if (segment) {
        const int slot = -it.row_end - 1;
        if (active) *reinterpret_cast<vec_t *>(carry + int64_t(slot) * ldc + lc) = acc;
    } else {
        // last row with entries, then any trailing empty rows of the block
        while (r < it.row_end) {
            finish_row();
            acc = V::zero();
            ++r;
        }
    }
}

// grid.x = ceil(n_items / 4), grid.y = column tiles of 64*VEC floats; one item per wavefront
template <int VEC, int U, int POLICY, bool ACC = false>
__global__ __launch_bounds__(256) void k_spmm_gather(
    const WorkItem *__restrict__ items, int n_items, const int32_t *__restrict__ rowptr,
    const int2 *__restrict__ cv, const float *__restrict__ X, int64_t ldx,
    const float *__restrict__ X2, int64_t ldx2, int split, int F, const float *__restrict__ bias,
    float *__restrict__ Y, int64_t ldy, float *__restrict__ carry, int64_t ldc) {
    const int lane = threadIdx.x & 63;
    const int item_id =
        __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (item_id >= n_items) return;
    spmm_item<VEC, U, POLICY, false, ACC>(items[item_id], lane, blockIdx.y * (64 * VEC), F, rowptr, cv, X, ldx, X2, ldx2,
                                          split, bias, Y, ldy, carry, ldc);
}

// The same kernel with the optimizer in its epilogue.  U = 4: the 16 registers of optimizer state in flight take
// the place of four gathered rows; 76 VGPRs = 6 waves per SIMD (forcing 64 spills) -- the launch moves twice the
// HBM bytes of the plain one and is bound by them, not by the number of gathers in flight.
template <int U>
__global__ __launch_bounds__(256) void k_spmm_gather_adam(
    const WorkItem *__restrict__ items, int n_items, const int32_t *__restrict__ rowptr,
    const int2 *__restrict__ cv, const float *__restrict__ X, int64_t ldx,
    const float *__restrict__ X2, int64_t ldx2, int split, int F, const float *__restrict__ bias,
    float *__restrict__ carry, int64_t ldc, const AdamRow ad) {
    const int lane = threadIdx.x & 63;
    const int item_id =
        __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (item_id >= n_items) return;
    spmm_item<4, U, 1, true>(items[item_id], lane, blockIdx.y * 256, F, rowptr, cv, X, ldx, X2, ldx2, split, bias,
                             nullptr, 0, carry, ldc, ad);
}

// Narrow rows (F <= 4 G floats, G = 16 or 32 lanes per gathered row; one operand buffer below 4 GB): the wave
// splits into S = 64 / G sub-groups.  In a row block every sub-group walks its OWN rows, so S rows are gathered per
// wave instruction and no cross-lane reduction is needed; in a long-row segment the sub-groups take entries sub,
// sub + S, ... and their partial sums are combined with wavefront shuffles (xor G, 2G) before the carry row is
// written.  A sub-group loads G (col,val) pairs with one coalesced instruction and hands them round with
// ds_bpermute.  The first form of this kernel (round 1, global-pointer gathers) spent 19.5 vector-ALU instructions
// per gather instruction (profiles/r02a_pmc_c4_f64_before_sweep.md: 64-bit address arithmetic, operand select,
// lane-index arithmetic); here
//   * the lane that loads a (col, val) pair turns the column into a 32-bit BYTE OFFSET once (one multiply per
//     16 entries instead of one 64-bit multiply-add per gather); padding lanes get an offset beyond the buffer;
//   * a gather is `buffer_load_dwordx4 v, v_off, s[rsrc], 0 offen`: the hardware adds the base, and an offset
//     outside the buffer returns zeros without touching memory -- padding needs no mask (0 * 0, never 0 * inf);
//   * the ds_bpermute source lane is a constant of the unrolled loop (folded into the instruction's offset);
//   * the four FMAs of an entry are two packed ones (v_pk_fma_f32).
typedef float pk_f2 __attribute__((ext_vector_type(2)));
// beyond any buffer this kernel is launched on (<= 0xFFFF0000 bytes) AND still beyond it after the largest lane
// offset is added (31 * 16 = 496 at G = 32): 32-bit offset arithmetic must not wrap back into the buffer
constexpr unsigned kOobOffset = 0xFFFFF000u;
static_assert(kOobOffset > 0xFFFF0000u && kOobOffset + 63u * 16u > kOobOffset, "padding offset must stay out of bounds");

template <int G, int U, bool ACC = false>
__global__ __launch_bounds__(256) void k_spmm_subb(
    const WorkItem *__restrict__ items, int n_items,
    const int2 *__restrict__ cv, const float *__restrict__ X, unsigned ldx4 /* row stride in bytes */,
    unsigned x_bytes, int F, const float *__restrict__ bias, float *__restrict__ Y, int64_t ldy,
    float *__restrict__ carry, int64_t ldc, const int4 *__restrict__ row_info) {
    constexpr int S = 64 / G;
    static_assert(G % U == 0, "a batch of G entries is gathered in whole groups of U");
    const int lane = threadIdx.x & 63;
    const int item_id =
        __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (item_id >= n_items) return;
    const int sub = lane / G, sl = lane % G;
    const int nvec = F / 4;                                    // <= G (checked by the launcher)
    const bool active = sl < nvec;
    const unsigned lane_off = static_cast<unsigned>((active ? sl : nvec - 1) * 16);
    const WorkItem it = items[item_id];
    const bool segment = it.row_end < 0;
    float4 bvec = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias != nullptr) bvec = *reinterpret_cast<const float4 *>(bias + lane_off / 4);
    // the descriptor is built from kernel arguments only: wave-uniform by construction (no waterfall loop)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(X), /*stride*/ static_cast<short>(0), static_cast<int>(x_bytes), 0x00020000);
    const int bp_base = (sub * G) * 4;                         // ds_bpermute address of this sub-group's lane 0

    // sum over entries start, start + stride, ... < end (per sub-group values)
    auto run = [&](int start, int stride, int end) -> float4 {
        pk_f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
        for (int t0 = start; t0 < end; t0 += G * stride) {
            const int my = t0 + sl * stride;
            int2 e = cv[min(my, end - 1)];                     // unconditional load, clamped index
            unsigned off = static_cast<unsigned>(e.x) * ldx4;  // byte offset of the operand row
            if (my >= end) {                                   // padding: no memory access, zero operand, zero weight
                off = kOobOffset;
                e.y = 0;
            }
            const int nb = min(G, (end - t0 + stride - 1) / stride);
#pragma unroll
            for (int u0 = 0; u0 < G; u0 += U) {
                if (u0 >= nb) break;
                float4 x[U];
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int src = bp_base + (u0 + u) * 4;
                    const unsigned o = static_cast<unsigned>(__builtin_amdgcn_ds_bpermute(src, static_cast<int>(off)));
                    v[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, e.y));
                    // + this lane's 16 bytes of the row (a padding offset stays beyond the buffer: no wrap)
                    const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(o + lane_off), 0, 0);
                    x[u] = make_float4(__int_as_float(raw[0]), __int_as_float(raw[1]), __int_as_float(raw[2]),
                                       __int_as_float(raw[3]));
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const pk_f2 vv = {v[u], v[u]};
                    const pk_f2 x01 = {x[u].x, x[u].y}, x23 = {x[u].z, x[u].w};
                    a01 = __builtin_elementwise_fma(vv, x01, a01);
                    a23 = __builtin_elementwise_fma(vv, x23, a23);
                }
            }
        }
        return make_float4(a01[0], a01[1], a23[0], a23[1]);
    };

    if (segment) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (it.nnz_begin + sub < it.nnz_end) acc = run(it.nnz_begin + sub, S, it.nnz_end);
#pragma unroll
        for (int off = G; off < 64; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off, 64);
            acc.y += __shfl_xor(acc.y, off, 64);
            acc.z += __shfl_xor(acc.z, off, 64);
            acc.w += __shfl_xor(acc.w, off, 64);
        }
        const int slot = -it.row_end - 1;
        if (sub == 0 && active) *reinterpret_cast<float4 *>(carry + int64_t(slot) * ldc + lane_off / 4) = acc;
    } else {
        // rows in order of falling degree (CsrBlock::row_info): the S rows of one pass have about the same length
        for (int i = it.row_begin + sub; i < it.row_end; i += S) {
            const int4 ri = row_info[i];
            const int b = ri.x, e = ri.y, r = ri.z;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < e) acc = run(b, 1, e);
            if constexpr (ACC) {                               // rows without entries are not touched
                if (b < e && active) {
                    float4 *dst = reinterpret_cast<float4 *>(Y + int64_t(r) * ldy + lane_off / 4);
                    const float4 y0 = *dst;
                    *dst = make_float4(y0.x + acc.x, y0.y + acc.y, y0.z + acc.z, y0.w + acc.w);
                }
                continue;
            }
            if (active)
                *reinterpret_cast<float4 *>(Y + int64_t(r) * ldy + lane_off / 4) =
                    make_float4(acc.x + bvec.x, acc.y + bvec.y, acc.z + bvec.z, acc.w + bvec.w);
        }
    }
}

// Dense hot block.  part[k][:] = sum over the columns c of this workgroup of vd[c][k] * X[c][:] for the
// 32 hot rows k (plan.hip: build_items), as v_mfma_f32_32x32x2_f32 products: per step a wave takes two
// consecutive columns c, c+1 -- the A fragment is 64 consecutive floats of vd (hot index = lane % 32,
// column = lane / 32), the B fragments are 32-float pieces of the two operand rows -- and accumulates
// NT tiles of 32 output columns.  X is read exactly once, in order (8 waves x `cpw` consecutive
// columns per workgroup).  The 8 partial tiles of a workgroup are added in wave order through LDS and
// written as ONE carry row per hot row; k_spmm_fix then adds the workgroups' rows in order.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kHotWaves = 8;

template <int NT>
__global__ __launch_bounds__(64 * kHotWaves) void k_spmm_hot(
    const float *__restrict__ vd, int64_t n_cols, int cpw, const float *__restrict__ X, int64_t ldx,
    const float *__restrict__ X2, int64_t ldx2, int split, int F, float *__restrict__ carry, int64_t ldc,
    int slot_base, int n_parts, int n_hot) {
    __shared__ float red[kHotWaves][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, half = lane >> 5;
    const int col0 = blockIdx.y * (32 * NT);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // feature columns past F are clamped to a valid address and NOT zeroed: they only feed result
    // columns >= F of the padded tile, which are never written
    int xoff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) xoff[t] = min(col0 + 32 * t + c, F - 1);
    const int64_t c_begin = (int64_t(blockIdx.x) * kHotWaves + wave) * cpw;   // vd is padded to the grid
    const float *pv = vd + (c_begin + half) * kHotRows + c;
    const int64_t last = n_cols - 1;
    constexpr int UR = 4;                      // column pairs per stage, two stages in flight
    float a[2][UR], x[2][UR][NT];
    auto load_stage = [&](int buf, int64_t cc) {
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int64_t col = cc + 2 * u + half;
            const int64_t cl = min(col, last);                 // padded tail: a valid row, weight 0 in vd
            a[buf][u] = pv[(cc - c_begin + 2 * u) * kHotRows];
            const float *row = cl < split ? X + cl * ldx : X2 + (cl - split) * ldx2;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float v = row[xoff[t]];
                x[buf][u][t] = col <= last ? v : 0.f;          // 0 * garbage must stay 0
            }
        }
    };
    auto mfma_stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < UR; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][u], x[buf][u][t], acc[t], 0, 0, 0);
    };
    // columns of this wave that exist: [c_begin, min(c_begin + cpw, n_cols)), walked in stages of 2 UR
    const int64_t c_stop = min(c_begin + cpw, n_cols);
    const int64_t n_stage = c_stop > c_begin ? (c_stop - c_begin + 2 * UR - 1) / (2 * UR) : 0;
    // cpw is a multiple of 2 UR, so a stage stays inside this wave's columns; past n_cols it reads the
    // zero padding of vd (allocated up to the grid size) and clamped operand rows
    if (n_stage > 0) load_stage(0, c_begin);
    for (int64_t st = 0; st < n_stage; st += 2) {
        if (st + 1 < n_stage) load_stage(1, c_begin + (st + 1) * 2 * UR);
        mfma_stage(0);
        if (st + 1 < n_stage) {
            if (st + 2 < n_stage) load_stage(0, c_begin + (st + 2) * 2 * UR);
            mfma_stage(1);
        }
    }
    // workgroup reduction, tile by tile, in wave order
    const int part = blockIdx.x;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[t][i];
        __syncthreads();
        const int col = col0 + 32 * t + c;
#pragma unroll
        for (int q = 0; q < 16 / kHotWaves; ++q) {
            const int i = wave * (16 / kHotWaves) + q;
            float sum = red[0][i][lane];
#pragma unroll
            for (int w = 1; w < kHotWaves; ++w) sum += red[w][i][lane];
            const int k = (i & 3) + 8 * (i >> 2) + 4 * half;   // hot row of accumulator register i
            if (k < n_hot && col < F)
                carry[(int64_t(slot_base) + int64_t(k) * n_parts + part) * ldc + col] = sum;
        }
    }
}

// One workgroup per long row: Y[row] = bias + carry[slot_begin] + ... + carry[slot_begin+count-1].
// Wave w adds slots w, w+4, ...; the four partials are combined through LDS in wave order.
template <int VEC, bool ADAM = false, bool ACC = false>
__global__ __launch_bounds__(256) void k_spmm_fix(const FixEntry *__restrict__ fix,
                                                  const float *__restrict__ carry, int64_t ldc,
                                                  const float *__restrict__ bias,
                                                  float *__restrict__ Y, int64_t ldy, int F,
                                                  const AdamRow ad = AdamRow{}) {
    using V = Vec<VEC>;
    using vec_t = typename V::type;
    __shared__ vec_t red[kWavesPerBlock][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int col0 = blockIdx.y * (64 * VEC);
    const int nvec = (min(F - col0, 64 * VEC) + VEC - 1) / VEC;
    const bool active = lane < nvec;
    const int lc = col0 + (active ? lane : nvec - 1) * VEC;
    const FixEntry fe = fix[blockIdx.x];
    const float *base = carry + int64_t(fe.slot_begin) * ldc + lc;
    vec_t acc = V::zero();
    int s = wave;
    // (the 32 hot rows hold one partial per workgroup of k_spmm_hot, 256 of them: eight loads in flight per wave;
    // the sum runs in slot order either way)
    for (; s + 7 * kWavesPerBlock < fe.count; s += 8 * kWavesPerBlock) {
        vec_t a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const vec_t *>(base + int64_t(s + u * kWavesPerBlock) * ldc);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = V::add(acc, a[u]);
    }
    for (; s + 3 * kWavesPerBlock < fe.count; s += 4 * kWavesPerBlock) {
        const vec_t a0 = *reinterpret_cast<const vec_t *>(base + int64_t(s) * ldc);
        const vec_t a1 = *reinterpret_cast<const vec_t *>(base + int64_t(s + kWavesPerBlock) * ldc);
        const vec_t a2 = *reinterpret_cast<const vec_t *>(base + int64_t(s + 2 * kWavesPerBlock) * ldc);
        const vec_t a3 = *reinterpret_cast<const vec_t *>(base + int64_t(s + 3 * kWavesPerBlock) * ldc);
        acc = V::add(V::add(V::add(V::add(acc, a0), a1), a2), a3);
    }
    for (; s < fe.count; s += kWavesPerBlock)
        acc = V::add(acc, *reinterpret_cast<const vec_t *>(base + int64_t(s) * ldc));
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && active) {
        vec_t t = V::add(V::add(red[0][lane], red[1][lane]), V::add(red[2][lane], red[3][lane]));
        if (bias != nullptr) t = V::add(t, *reinterpret_cast<const vec_t *>(bias + lc));
        if constexpr (ADAM) {
            AdamState4 st;
            const int64_t off = int64_t(fe.row) * ad.ld + lc;
            adam_prefetch(ad, off, st);
            adam_update(ad, off, t, st, ad.dev_scalars ? ad.dev_scalars[0] : ad.step_size,
                        ad.dev_scalars ? ad.dev_scalars[1] : ad.inv_bc2_sqrt);
        } else {
            vec_t *dst = reinterpret_cast<vec_t *>(Y + int64_t(fe.row) * ldy + lc);
            if constexpr (ACC) t = V::add(*dst, t);            // (a long row always holds entries)
            *dst = t;
        }
    }
}

template <int NT>
void launch_hot(const CsrBlock &b, const float *X, int64_t ldx, const float *X2, int64_t ldx2, int split, int F,
                float *carry, int64_t ldc, hipStream_t stream) {
    dim3 grid(b.hot_parts, (F + 32 * NT - 1) / (32 * NT));
    k_spmm_hot<NT><<<grid, 64 * kHotWaves, 0, stream>>>(b.hot_vals, b.n_cols, b.hot_cpw, X, ldx, X2, ldx2, split, F,
                                                       carry, ldc, b.hot_slot_base, b.hot_parts, b.n_hot);
}

// `ad` != nullptr (VEC == 4 only): the finished rows are spent on the optimizer instead of being stored in Y;
// `acc`: the accumulate form (Y += M X on the rows that hold entries; no bias)
template <int VEC>
int launch_vec(const CsrBlock &blk, const float *X, int64_t ldx, const float *X2, int64_t ldx2, int split,
               int F, const float *bias, float *Y, int64_t ldy, float *carry, hipStream_t stream,
               const AdamRow *ad = nullptr, const bool acc = false) {
    const int tiles = (F + 64 * VEC - 1) / (64 * VEC);
    const int64_t ldc = round_up4(F);
    // With a dense hot block the float4 kernels run on the partition without the hot rows, next to
    // k_spmm_hot; the scalar kernel (unaligned operands, F % 4 != 0) uses the complete partition.
    struct View {
        const WorkItem *items;
        int32_t n_items;
        const FixEntry *fix;
        int32_t n_fix;
    };
    const bool all = VEC == 1 && blk.n_hot > 0;
    const View b = all ? View{blk.items_all, blk.n_items_all, blk.fix_all, blk.n_fix_all}
                       : View{blk.items, blk.n_items, blk.fix, blk.n_fix};
    const int32_t *rowptr = blk.rowptr;
    const int2 *cv = blk.cv;
    if (VEC == 4 && blk.n_hot > 0) {
        const int nt = std::min(8, (F + 31) / 32);   // 32-column MFMA tiles per wave (8 = 256 columns)
        switch (nt) {
            case 1: launch_hot<1>(blk, X, ldx, X2, ldx2, split, F, carry, ldc, stream); break;
            case 2: launch_hot<2>(blk, X, ldx, X2, ldx2, split, F, carry, ldc, stream); break;
            case 3: launch_hot<3>(blk, X, ldx, X2, ldx2, split, F, carry, ldc, stream); break;
            case 4: launch_hot<4>(blk, X, ldx, X2, ldx2, split, F, carry, ldc, stream); break;
            case 5: launch_hot<5>(blk, X, ldx, X2, ldx2, split, F, carry, ldc, stream); break;
            case 6: launch_hot<6>(blk, X, ldx, X2, ldx2, split, F, carry, ldc, stream); break;
            case 7: launch_hot<7>(blk, X, ldx, X2, ldx2, split, F, carry, ldc, stream); break;
            default: launch_hot<8>(blk, X, ldx, X2, ldx2, split, F, carry, ldc, stream); break;
        }
        TGCN_HIP_CHECK(hipGetLastError());
    }
    if (b.n_items > 0) {
        dim3 grid((b.n_items + kWavesPerBlock - 1) / kWavesPerBlock, tiles);
        // operand as one buffer below 4 GB (and rows addressable with 32-bit byte offsets): buffer-addressed form
        const uint64_t x_extent = (static_cast<uint64_t>(blk.n_cols - 1) * static_cast<uint64_t>(ldx) + F) * 4u;
        const bool buf_ok = split == INT32_MAX && x_extent <= 0xFFFF0000ull;
        if (ad != nullptr) {
            if constexpr (VEC == 4) {
                // U = 2 / 4 / 8 gathered rows in flight measured the same (5.62-5.66 ms at c4): the launch is bound by
                // the 37 GB it moves over the fabric, not by gather latency
                k_spmm_gather_adam<4><<<grid, 256, 0, stream>>>(b.items, b.n_items, rowptr, cv, X, ldx, X2, ldx2, split, F,
                                                                bias, carry, ldc, *ad);
            }
        } else if (VEC == 4 && F <= 128 && buf_ok) {
            // narrow feature rows (the layer-2 width C): sub-group kernel.  U = 4 gathers in flight per sub-group:
            // 8 / 16 measured 8 % / 6 % slower at F = 64 (c4), document rows hold ~10 entries and every started
            // group of U is gathered in full (with the rows of a block sorted by degree 2 / 8 measured 10 % / 8 % slower)
            const unsigned ldx4 = static_cast<unsigned>(ldx * 4), xb = static_cast<unsigned>(x_extent);
            if (F <= 64 && !acc)
                k_spmm_subb<16, 4><<<grid, 256, 0, stream>>>(b.items, b.n_items, cv, X, ldx4, xb, F, bias, Y, ldy, carry, ldc,
                                                            blk.row_info);
            else if (F <= 64)
                k_spmm_subb<16, 4, true><<<grid, 256, 0, stream>>>(b.items, b.n_items, cv, X, ldx4, xb, F, bias, Y, ldy, carry,
                                                                  ldc, blk.row_info);
            else if (!acc)
                k_spmm_subb<32, 4><<<grid, 256, 0, stream>>>(b.items, b.n_items, cv, X, ldx4, xb, F, bias, Y, ldy, carry, ldc,
                                                            blk.row_info);
            else
                k_spmm_subb<32, 4, true><<<grid, 256, 0, stream>>>(b.items, b.n_items, cv, X, ldx4, xb, F, bias, Y, ldy, carry,
                                                                  ldc, blk.row_info);
        } else {
            // full-wave kernel: wide rows, split operands, operands beyond 4 GB, unaligned / odd widths (VEC = 1).
            // Streams ((col,val) pairs, result rows) are marked non-temporal on the float4 path (measured best on c4)
            // (the accumulate form reads the rows it adds to and the next column block reads them again: plain stores)
            if (acc) {
                constexpr int P = VEC == 4 ? 1 : 0;
                k_spmm_gather<VEC, 8, P, true><<<grid, 256, 0, stream>>>(b.items, b.n_items, rowptr, cv, X, ldx, X2, ldx2, split,
                                                                         F, bias, Y, ldy, carry, ldc);
            } else if constexpr (VEC == 4)
                k_spmm_gather<4, 8, 3><<<grid, 256, 0, stream>>>(b.items, b.n_items, rowptr, cv, X, ldx, X2, ldx2, split, F,
                                                                 bias, Y, ldy, carry, ldc);
            else
                k_spmm_gather<VEC, 8, 0><<<grid, 256, 0, stream>>>(b.items, b.n_items, rowptr, cv, X, ldx, X2, ldx2, split,
                                                                   F, bias, Y, ldy, carry, ldc);
        }
        TGCN_HIP_CHECK(hipGetLastError());
    }
    if (b.n_fix > 0) {
        dim3 grid(b.n_fix, tiles);
        if (ad != nullptr) {
            if constexpr (VEC == 4) k_spmm_fix<4, true><<<grid, 256, 0, stream>>>(b.fix, carry, ldc, bias, nullptr, 0, F, *ad);
        } else if (acc) {
            k_spmm_fix<VEC, false, true><<<grid, 256, 0, stream>>>(b.fix, carry, ldc, bias, Y, ldy, F);
        } else {
            k_spmm_fix<VEC><<<grid, 256, 0, stream>>>(b.fix, carry, ldc, bias, Y, ldy, F);
        }
        TGCN_HIP_CHECK(hipGetLastError());
    }
    return TGCN_OK;
}

}  // namespace

int launch_spmm(const CsrBlock &b, const float *X, int64_t ldx, const float *X2, int64_t ldx2, int split,
                int F, const float *bias, float *Y, int64_t ldy, float *carry, hipStream_t stream, bool acc) {
    if (X2 == nullptr) {  // single operand
        X2 = X;
        ldx2 = ldx;
        split = INT32_MAX;
    }
    const uintptr_t align = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(X2) |
                            reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(bias) |
                            reinterpret_cast<uintptr_t>(carry);
    const bool vec4 = (F % 4 == 0) && (ldx % 4 == 0) && (ldx2 % 4 == 0) && (ldy % 4 == 0) && (align % 16 == 0);
    return vec4 ? launch_vec<4>(b, X, ldx, X2, ldx2, split, F, bias, Y, ldy, carry, stream, nullptr, acc)
                : launch_vec<1>(b, X, ldx, X2, ldx2, split, F, bias, Y, ldy, carry, stream, nullptr, acc);
}

// the SpMM whose rows feed the optimizer (tgcn_spmm_adam); float4 path only, checked by the caller
int launch_spmm_adam(const CsrBlock &b, const float *X, int64_t ldx, const float *X2, int64_t ldx2, int split, int F,
                     float *carry, hipStream_t stream, const AdamRow &ad) {
    if (X2 == nullptr) {  // single operand
        X2 = X;
        ldx2 = ldx;
        split = INT32_MAX;
    }
    return launch_vec<4>(b, X, ldx, X2, ldx2, split, F, nullptr, nullptr, 0, carry, stream, &ad);
}

// tgcn_spmm_split / tgcn_spmm_acc behind one set of argument checks (below)
int spmm_entry(const tgcn_plan *plan, int transpose, const float *X, int64_t ldx, const float *X2, int64_t ldx2,
               int64_t split, int F, const float *bias, float *Y, int64_t ldy, void *workspace, size_t workspace_bytes,
               tgcn_stream stream, bool acc);

}  // namespace tgcn

extern "C" {

int tgcn_spmm_adam(const tgcn_plan *plan, int transpose, const float *G, int64_t ldg, int F, float *param,
                   float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq, int64_t ldp, double lr, double beta1,
                   double beta2, double eps, double weight_decay, int64_t step, const float *scalars_dev,
                   void *workspace, size_t workspace_bytes, tgcn_stream stream) {
    return tgcn_spmm_adam_split(plan, transpose, G, ldg, nullptr, 0, 0, F, param, exp_avg, exp_avg_sq, max_exp_avg_sq, ldp,
                                lr, beta1, beta2, eps, weight_decay, step, scalars_dev, workspace, workspace_bytes, stream);
}

int tgcn_spmm_adam_split(const tgcn_plan *plan, int transpose, const float *G, int64_t ldg, const float *G2, int64_t ldg2,
                         int64_t split, int F, float *param, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq,
                         int64_t ldp, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                         const float *scalars_dev, void *workspace, size_t workspace_bytes, tgcn_stream stream) {
    using namespace tgcn;
    if (!plan || !G || !param || !exp_avg || !exp_avg_sq) {
        set_error("tgcn_spmm_adam: NULL plan / G / param / state");
        return TGCN_E_INVALID;
    }
    if (F <= 0 || F % 4 != 0 || ldg < F || ldp < F || ldg % 4 != 0 || ldp % 4 != 0 || (step < 1 && !scalars_dev)) {
        set_error("tgcn_spmm_adam: need F %% 4 == 0, ldg, ldp >= F and multiples of 4, step >= 1 (F=%d ldg=%lld ldp=%lld)",
                  F, (long long)ldg, (long long)ldp);
        return TGCN_E_INVALID;
    }
    if (G2 && (ldg2 < F || ldg2 % 4 != 0 || split < 0 || split >= INT32_MAX || reinterpret_cast<uintptr_t>(G2) % 16 != 0)) {
        set_error("tgcn_spmm_adam_split: need ldg2 >= F and a multiple of 4, 0 <= split < 2^31, G2 16-byte aligned");
        return TGCN_E_INVALID;
    }
    const uintptr_t a = reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(param) |
                        reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq) |
                        reinterpret_cast<uintptr_t>(max_exp_avg_sq) | reinterpret_cast<uintptr_t>(workspace);
    if (a % 16 != 0) {
        set_error("tgcn_spmm_adam: buffers must be 16-byte aligned");
        return TGCN_E_INVALID;
    }
    if (transpose && !plan->symmetric && !plan->has_transpose) {
        set_error("tgcn_spmm_adam: transpose requested but the plan was built without it");
        return TGCN_E_INVALID;
    }
    const size_t need = tgcn_spmm_workspace_bytes(plan, transpose, F);
    if (need > 0 && (!workspace || workspace_bytes < need)) {
        set_error("tgcn_spmm_adam: workspace of %zu bytes given, %zu needed", workspace_bytes, need);
        return TGCN_E_WORKSPACE;
    }
    const CsrBlock &b = (transpose && !plan->symmetric) ? plan->bwd : plan->fwd;
    if (b.n_rows == 0) return TGCN_OK;
    AdamRow ad;
    ad.p = param;
    ad.m = exp_avg;
    ad.v = exp_avg_sq;
    ad.vmax = max_exp_avg_sq;
    ad.ld = ldp;
    // every constant derived in double and rounded once, as tgcn_adam_step (and torch) do
    const double st = static_cast<double>(step < 1 ? 1 : step);
    const double bc1 = 1.0 - std::pow(beta1, st), bc2 = 1.0 - std::pow(beta2, st);
    ad.w1 = static_cast<float>(1.0 - beta1);
    ad.b2 = static_cast<float>(beta2);
    ad.w2 = static_cast<float>(1.0 - beta2);
    ad.eps = static_cast<float>(eps);
    ad.wd = static_cast<float>(weight_decay);
    ad.step_size = static_cast<float>(lr / bc1);
    ad.inv_bc2_sqrt = static_cast<float>(1.0 / std::sqrt(bc2));
    ad.dev_scalars = scalars_dev;
    int cur = -1;
    TGCN_HIP_CHECK(hipGetDevice(&cur));
    if (cur != plan->device) TGCN_HIP_CHECK(hipSetDevice(plan->device));
    const int rc = launch_spmm_adam(b, G, ldg, G2, ldg2, static_cast<int>(split), F,
                                    need ? static_cast<float *>(workspace) : nullptr, static_cast<hipStream_t>(stream), ad);
    if (cur != plan->device) (void)hipSetDevice(cur);
    return rc;
}

size_t tgcn_spmm_workspace_bytes(const tgcn_plan *plan, int transpose, int F) {
    if (!plan || F <= 0) return 0;
    const tgcn::CsrBlock &b = (transpose && !plan->symmetric) ? plan->bwd : plan->fwd;
    const size_t slots = static_cast<size_t>(std::max(b.n_segments, b.n_segments_all));
    return sizeof(float) * slots * static_cast<size_t>(tgcn::round_up4(F));
}

int tgcn_spmm(const tgcn_plan *plan, int transpose, const float *X, int64_t ldx, int F,
              const float *bias, float *Y, int64_t ldy, void *workspace, size_t workspace_bytes,
              tgcn_stream stream) {
    return tgcn_spmm_split(plan, transpose, X, ldx, nullptr, 0, 0, F, bias, Y, ldy, workspace,
                           workspace_bytes, stream);
}

int tgcn_spmm_split(const tgcn_plan *plan, int transpose, const float *X, int64_t ldx, const float *X2,
                    int64_t ldx2, int64_t split, int F, const float *bias, float *Y, int64_t ldy,
                    void *workspace, size_t workspace_bytes, tgcn_stream stream) {
    return tgcn::spmm_entry(plan, transpose, X, ldx, X2, ldx2, split, F, bias, Y, ldy, workspace, workspace_bytes, stream,
                            false);
}

int tgcn_spmm_acc(const tgcn_plan *plan, int transpose, const float *X, int64_t ldx, const float *X2, int64_t ldx2,
                  int64_t split, int F, float *Y, int64_t ldy, void *workspace, size_t workspace_bytes,
                  tgcn_stream stream) {
    return tgcn::spmm_entry(plan, transpose, X, ldx, X2, ldx2, split, F, nullptr, Y, ldy, workspace, workspace_bytes,
                            stream, true);
}

}  // extern "C"

namespace tgcn {

int spmm_entry(const tgcn_plan *plan, int transpose, const float *X, int64_t ldx, const float *X2,
               int64_t ldx2, int64_t split, int F, const float *bias, float *Y, int64_t ldy,
               void *workspace, size_t workspace_bytes, tgcn_stream stream, bool acc) {
    if (!plan || !X || !Y) {
        set_error("tgcn_spmm: NULL plan/X/Y");
        return TGCN_E_INVALID;
    }
    if (X2 != nullptr && (ldx2 < F || split < 0 || split > INT32_MAX)) {
        set_error("tgcn_spmm_split: need ldx2 >= F and 0 <= split < 2^31");
        return TGCN_E_INVALID;
    }
    if (F <= 0 || ldx < F || ldy < F) {
        set_error("tgcn_spmm: need F > 0 and ldx, ldy >= F (F=%d ldx=%lld ldy=%lld)", F,
                  (long long)ldx, (long long)ldy);
        return TGCN_E_INVALID;
    }
    if (transpose && !plan->symmetric && !plan->has_transpose) {
        set_error("tgcn_spmm: transpose requested but the plan was built without it");
        return TGCN_E_INVALID;
    }
    const size_t need = tgcn_spmm_workspace_bytes(plan, transpose, F);
    if (need > 0 && (!workspace || workspace_bytes < need)) {
        set_error("tgcn_spmm: workspace of %zu bytes given, %zu needed", workspace_bytes, need);
        return TGCN_E_WORKSPACE;
    }
    if (need > 0 && reinterpret_cast<uintptr_t>(workspace) % 16 != 0) {
        set_error("tgcn_spmm: workspace must be 16-byte aligned");
        return TGCN_E_INVALID;
    }
    const CsrBlock &b = (transpose && !plan->symmetric) ? plan->bwd : plan->fwd;
    if (b.n_rows == 0) return TGCN_OK;
    int cur = -1;
    TGCN_HIP_CHECK(hipGetDevice(&cur));
    if (cur != plan->device) TGCN_HIP_CHECK(hipSetDevice(plan->device));
    const int st = launch_spmm(b, X, ldx, X2, ldx2, static_cast<int>(split), F, bias, Y, ldy,
                               need ? static_cast<float *>(workspace) : nullptr,
                               static_cast<hipStream_t>(stream), acc);
    if (cur != plan->device) (void)hipSetDevice(cur);
    return st;
}

}  // namespace tgcn
