// Training-step kernels around the GCN (row A6 of SURVEY.md 8(a): flat_amazon.py:82,89,99-106).
//
//   k_masked_ce    criterion(outputs[g.train_mask], g.y[g.train_mask]) with CrossEntropyLoss('mean')
//                  (flat_amazon.py:82,101-102) and its gradient w.r.t. the full logits matrix, in one
//                  pass over [N, C]: replaces boolean-mask indexing, log_softmax, nll_loss and their
//                  three backward kernels.  HBM-bound: reads N*C*4 (+ 9 B/row), writes N*C*4.
//   k_adam         torch.optim.Adam(..., amsgrad=True).step() (flat_amazon.py:89,106), one fused
//                  elementwise pass: reads p, g, m, v, vmax and writes p, m, v, vmax (36 B/element)
//                  instead of ~10 separate multi-tensor kernels.  HBM-bound.
// Both are deterministic (fixed-order two-stage reductions, no atomics).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "common.h"

namespace tgcn {
namespace {

constexpr int kCeBlocks = 2048;

// LPR lanes cooperate on one row; a wave handles 64/LPR rows at a time, two such groups per loop
// iteration (independent dependency chains: the kernel is instruction- and latency-bound, not bandwidth-bound).
// For C <= KPL * LPR (the launcher's choice for C <= 256) a lane keeps its <= KPL logits in registers: one read
// of the row, one expf per element, shared by the loss and the gradient.
// V4 (C, ld, ldd multiples of 4, 16-byte aligned rows): the lane's KPL logits are CONSECUTIVE columns -- 16-byte
// loads and stores; otherwise they are LPR columns apart (4-byte accesses).
// colpart != nullptr (register path only): the workgroup also leaves the column sums of the gradient rows it
// wrote -- the bias gradient of the layer that produced the logits -- in colpart[blockIdx.x][0..C).
template <int LPR, bool V4, int KPL = 4>
__global__ __launch_bounds__(256) void k_masked_ce(const float *__restrict__ logits, int64_t ld, int C,
                                                   const int64_t *__restrict__ target,
                                                   const uint8_t *__restrict__ mask, int64_t n_rows,
                                                   float inv_count, float *__restrict__ dlogits,
                                                   int64_t ldd, float *__restrict__ partial,
                                                   int64_t *__restrict__ pred, float *__restrict__ colpart) {
    constexpr int RPW = 64 / LPR;
    static_assert(KPL == 4 || KPL == 8, "logits per lane on the register path");
    constexpr int UN = 2;        // row groups in flight per wave
    __shared__ float red[4];
    __shared__ float cred[4][LPR * KPL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, sl = lane % LPR;
    const int64_t rows_per_iter = int64_t(gridDim.x) * 4 * RPW * UN;
    const bool in_regs = C <= KPL * LPR;
    auto col = [&](int k) { return V4 ? sl * KPL + k : sl + k * LPR; };
    float loss = 0.f;
    float cs[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) cs[k] = 0.f;
    for (int64_t r0 = (int64_t(blockIdx.x) * 4 + wave) * RPW * UN; r0 < n_rows; r0 += rows_per_iter) {
        if (in_regs) {
            float x[UN][KPL];
            bool on[UN], valid[UN];
            int64_t r[UN], t[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                r[u] = r0 + u * RPW + sub;
                valid[u] = r[u] < n_rows;
                on[u] = valid[u] && mask[r[u]] != 0;
                const float *row = logits + (valid[u] ? r[u] : 0) * ld;
                t[u] = on[u] ? target[r[u]] : 0;
                const bool need = on[u] || (pred != nullptr && valid[u]);
                if constexpr (V4) {
#pragma unroll
                    for (int q = 0; q < KPL / 4; ++q) {
                        float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                        if (need && sl * KPL + 4 * q < C) v = *reinterpret_cast<const float4 *>(row + sl * KPL + 4 * q);
                        x[u][4 * q] = v.x, x[u][4 * q + 1] = v.y, x[u][4 * q + 2] = v.z, x[u][4 * q + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < KPL; ++k) {
                        const int c = col(k);
                        x[u][k] = (need && c < C) ? row[c] : -INFINITY;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                float m = fmaxf(fmaxf(x[u][0], x[u][1]), fmaxf(x[u][2], x[u][3]));
#pragma unroll
                for (int q = 1; q < KPL / 4; ++q)
                    m = fmaxf(m, fmaxf(fmaxf(x[u][4 * q], x[u][4 * q + 1]), fmaxf(x[u][4 * q + 2], x[u][4 * q + 3])));
#pragma unroll
                for (int off = LPR / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
                if (pred != nullptr) {
                    // arg-max of the row (first index on ties), for every row, masked or not
                    int bi = INT32_MAX;
#pragma unroll
                    for (int k = KPL - 1; k >= 0; --k)
                        if (x[u][k] == m) bi = col(k);
#pragma unroll
                    for (int off = LPR / 2; off > 0; off >>= 1) bi = min(bi, __shfl_xor(bi, off, 64));
                    if (valid[u] && sl == 0) pred[r[u]] = bi == INT32_MAX ? 0 : bi;
                }
                float e[KPL], sum = 0.f;
#pragma unroll
                for (int k = 0; k < KPL; ++k) {
                    e[k] = on[u] ? expf(x[u][k] - m) : 0.f;   // exp(-inf) = 0 for the padding
                    sum += e[k];
                }
#pragma unroll
                for (int off = LPR / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
                if (on[u]) {
                    const float lse = m + logf(sum);
                    // the lane that holds the target logit adds the row's loss term
#pragma unroll
                    for (int k = 0; k < KPL; ++k)
                        if (col(k) == t[u]) loss += lse - x[u][k];
                    // a class index outside [0, C) poisons the loss instead of being dropped silently
                    // (torch raises; the Python wrapper checks the labels once per tensor)
                    if (sl == 0 && (t[u] < 0 || t[u] >= C)) loss = NAN;
                }
                if (dlogits != nullptr && valid[u]) {
                    float *drow = dlogits + r[u] * ldd;
                    const float inv = on[u] ? 1.f / sum : 0.f;
                    float d[KPL];
#pragma unroll
                    for (int k = 0; k < KPL; ++k) {
                        const int c = col(k);
                        d[k] = (on[u] && c < C) ? (e[k] * inv - (c == t[u] ? 1.f : 0.f)) * inv_count : 0.f;
                        cs[k] += d[k];
                    }
                    if constexpr (V4) {
#pragma unroll
                        for (int q = 0; q < KPL / 4; ++q)
                            if (sl * KPL + 4 * q < C)
                                *reinterpret_cast<float4 *>(drow + sl * KPL + 4 * q) =
                                    make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
                    } else {
#pragma unroll
                        for (int k = 0; k < KPL; ++k)
                            if (col(k) < C) drow[col(k)] = d[k];
                    }
                }
            }
        } else {
            for (int u = 0; u < UN; ++u) {
                const int64_t r = r0 + u * RPW + sub;
                const bool valid = r < n_rows;
                const bool on = valid && mask[r] != 0;
                const float *row = logits + (valid ? r : 0) * ld;
                float m = -INFINITY;
                if (on || (pred != nullptr && valid))
                    for (int c = sl; c < C; c += LPR) m = fmaxf(m, row[c]);
#pragma unroll
                for (int off = LPR / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
                if (pred != nullptr) {
                    int bi = INT32_MAX;
                    if (valid)
                        for (int c = sl; c < C; c += LPR)
                            if (row[c] == m) {
                                bi = c;
                                break;
                            }
#pragma unroll
                    for (int off = LPR / 2; off > 0; off >>= 1) bi = min(bi, __shfl_xor(bi, off, 64));
                    if (valid && sl == 0) pred[r] = bi == INT32_MAX ? 0 : bi;
                }
                float s = 0.f;
                if (on)
                    for (int c = sl; c < C; c += LPR) s += expf(row[c] - m);
#pragma unroll
                for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
                const float lse = m + logf(s);
                int64_t t = 0;
                if (on) {
                    t = target[r];
                    // no read outside the row: an invalid class index poisons the loss (see above)
                    if (sl == 0) loss += (t >= 0 && t < C) ? lse - row[t] : NAN;
                }
                if (dlogits != nullptr && valid) {
                    float *drow = dlogits + r * ldd;
                    for (int c = sl; c < C; c += LPR)
                        drow[c] = on ? (expf(row[c] - lse) - (c == t ? 1.f : 0.f)) * inv_count : 0.f;
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) loss += __shfl_xor(loss, off, 64);
    if (lane == 0) red[wave] = loss;
    if (colpart != nullptr) {
        // column sums of the gradient rows: the sub-groups of a wave, then the four waves, in a fixed order
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1) cs[k] += __shfl_xor(cs[k], off, 64);
            if (sub == 0) cred[wave][sl * KPL + k] = cs[k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    if (colpart != nullptr && wave == 0 && sub == 0) {
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
            const int c = col(k), i = sl * KPL + k;
            if (c < C) colpart[int64_t(blockIdx.x) * C + c] = (cred[0][i] + cred[1][i]) + (cred[2][i] + cred[3][i]);
        }
    }
}

// x *= *scale unless the device scalar is exactly 1 (the seed `loss.backward()` hands to the loss node): the
// multiplication by 1.0f is the identity bit for bit, so skipping the pass changes nothing.
__global__ __launch_bounds__(256) void k_scale_unless_one(float *__restrict__ x, int64_t n, const float *__restrict__ scale) {
    const float s = *scale;
    if (s == 1.0f) return;
    const int64_t stride = int64_t(gridDim.x) * 256;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) x[i] *= s;
}

__global__ void k_ce_final(const float *__restrict__ partial, int n, float inv_count,
                           float *__restrict__ loss) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (static_cast<int>(threadIdx.x) < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = red[0] * inv_count;
}


// torch/optim/adam.py (_single_tensor_adam), op for op:
//   g += wd * p;  m.lerp_(g, 1-b1);  v = b2*v + (1-b2)*g*g;  vmax = max(vmax, v)   [amsgrad]
//   denom = sqrt(vmax or v) / sqrt(1 - b2^t) + eps;  p -= (lr / (1 - b1^t)) * m / denom
// Capturable form: the step count lives on the device (torch's `capturable=True`), so that one
// captured HIP graph can be replayed for every training step.  One thread advances the counter and
// derives the two step-dependent scalars in double, exactly as the host path does.
__global__ void k_adam_scalars(int64_t *step, double lr, double b1, double b2, float *scalars) {
    const int64_t s = *step + 1;
    *step = s;
    const double bc1 = 1.0 - pow(b1, static_cast<double>(s));
    const double bc2 = 1.0 - pow(b2, static_cast<double>(s));
    scalars[0] = static_cast<float>(lr / bc1);
    scalars[1] = static_cast<float>(1.0 / sqrt(bc2));
}

// NT: the optimizer state (m, v, vmax) and the gradient are touched once per step -- stream them past
// the caches with non-temporal loads / stores; the parameter itself is the next SpMM's operand and
// keeps plain accesses.
typedef float adam_f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld4(const float *q) {
    if constexpr (NT) {
        const adam_f4 t = __builtin_nontemporal_load(reinterpret_cast<const adam_f4 *>(q));
        return make_float4(t.x, t.y, t.z, t.w);
    } else {
        return *reinterpret_cast<const float4 *>(q);
    }
}
template <bool NT>
__device__ __forceinline__ void st4(float *q, const float4 &a) {
    if constexpr (NT) {
        adam_f4 t = {a.x, a.y, a.z, a.w};
        __builtin_nontemporal_store(t, reinterpret_cast<adam_f4 *>(q));
    } else {
        *reinterpret_cast<float4 *>(q) = a;
    }
}

template <bool AMSGRAD, bool NT>
__global__ __launch_bounds__(256) void k_adam(float *__restrict__ p, const float *__restrict__ g,
                                              float *__restrict__ m, float *__restrict__ v,
                                              float *__restrict__ vmax, int64_t n, float w1 /* 1-b1 */,
                                              float b2, float w2 /* 1-b2 */, float eps, float wd,
                                              float step_size, float inv_bc2_sqrt,
                                              const float *__restrict__ dev_scalars) {
    if (dev_scalars != nullptr) {
        step_size = dev_scalars[0];
        inv_bc2_sqrt = dev_scalars[1];
    }
    const int64_t stride = int64_t(gridDim.x) * blockDim.x * 4;
    for (int64_t i = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            float4 pp = *reinterpret_cast<float4 *>(p + i);
            const float4 gg = ld4<NT>(g + i);
            float4 mm = ld4<NT>(m + i);
            float4 vv = ld4<NT>(v + i);
            float4 xx = make_float4(0.f, 0.f, 0.f, 0.f);
            if (AMSGRAD) xx = ld4<NT>(vmax + i);
            float *P = &pp.x, *M = &mm.x, *V = &vv.x, *X = &xx.x;
            const float *G = &gg.x;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                adam_element(P[k], G[k], M[k], V[k], X[k], AMSGRAD, w1, b2, w2, eps, wd, step_size, inv_bc2_sqrt);
            *reinterpret_cast<float4 *>(p + i) = pp;
            st4<NT>(m + i, mm);
            st4<NT>(v + i, vv);
            if (AMSGRAD) st4<NT>(vmax + i, xx);
        } else {
            for (int64_t j = i; j < n; ++j) {
                float pj = p[j], mj = m[j], vj = v[j], xj = AMSGRAD ? vmax[j] : 0.f;
                adam_element(pj, g[j], mj, vj, xj, AMSGRAD, w1, b2, w2, eps, wd, step_size, inv_bc2_sqrt);
                p[j] = pj;
                m[j] = mj;
                v[j] = vj;
                if (AMSGRAD) vmax[j] = xj;
            }
        }
    }
}

}  // namespace
}  // namespace tgcn

extern "C" {

size_t tgcn_masked_ce_workspace_bytes(void) { return sizeof(float) * tgcn::kCeBlocks; }

int tgcn_masked_ce(const float *logits, int64_t ld, int64_t n_rows, int n_classes,
                   const int64_t *target, const uint8_t *mask, float inv_count, float *loss,
                   float *dlogits, int64_t ldd, void *workspace, size_t workspace_bytes,
                   tgcn_stream stream) {
    return tgcn_masked_ce_pred(logits, ld, n_rows, n_classes, target, mask, inv_count, loss, dlogits, ldd,
                               nullptr, workspace, workspace_bytes, stream);
}

static int masked_ce_impl(const char *who, const float *logits, int64_t ld, int64_t n_rows, int n_classes,
                          const int64_t *target, const uint8_t *mask, float inv_count, float *loss,
                          float *dlogits, int64_t ldd, float *dbias, int64_t *pred, void *workspace,
                          size_t workspace_bytes, size_t workspace_need, tgcn_stream stream) {
    using namespace tgcn;
    if (!logits || !target || !mask || !loss || n_rows < 0 || n_classes <= 0 || ld < n_classes ||
        (dlogits && ldd < n_classes) || (dbias && !dlogits)) {
        set_error("%s: bad argument (n_rows=%lld C=%d ld=%lld)", who, (long long)n_rows, n_classes, (long long)ld);
        return TGCN_E_INVALID;
    }
    if (!workspace || workspace_bytes < workspace_need) {
        set_error("%s: workspace of %zu bytes given, %zu needed", who, workspace_bytes, workspace_need);
        return TGCN_E_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *partial = static_cast<float *>(workspace);
    const int C = n_classes;
    int64_t rows_per_block = 2 * 4 * (C <= 16 ? 16 : C <= 32 ? 8 : C <= 64 ? 4 : C <= 128 ? 2 : 1);
    int grid = static_cast<int>(std::min<int64_t>(kCeBlocks, std::max<int64_t>(1, (n_rows + rows_per_block - 1) / rows_per_block)));
    const bool v4 = C % 4 == 0 && ld % 4 == 0 && reinterpret_cast<uintptr_t>(logits) % 16 == 0 &&
                    (!dlogits || (ldd % 4 == 0 && reinterpret_cast<uintptr_t>(dlogits) % 16 == 0));
    // the column sums ride along on the register path (C <= 256); wider rows take a separate pass below
    float *colpart = (dbias && C <= 256) ? partial + kCeBlocks : nullptr;
#define TGCN_CE(LPR, KPL)                                                                                              \
    do {                                                                                                               \
        if (v4)                                                                                                        \
            k_masked_ce<LPR, true, KPL><<<grid, 256, 0, s>>>(logits, ld, C, target, mask, n_rows, inv_count, dlogits,   \
                                                             ldd, partial, pred, colpart);                             \
        else                                                                                                           \
            k_masked_ce<LPR, false, KPL><<<grid, 256, 0, s>>>(logits, ld, C, target, mask, n_rows, inv_count, dlogits,  \
                                                              ldd, partial, pred, colpart);                            \
    } while (0)
    // Logits per lane on the register path: 8 (half the lanes per row, one shuffle level less in each of the three
    // row reductions) measured 0.247 -> 0.149 ms without and 0.308 -> 0.232 ms with the gradient at c4 (2 M x 64);
    // 16 was slower again (0.17 / 0.28 ms).  Four per lane remains for C <= 16.
    constexpr int kpl = 8;
    if (kpl == 8 && C > 16 && C <= 256) {
        // eight logits per lane: half the lanes per row, one reduction level less
        rows_per_block = 2 * 4 * (C <= 32 ? 16 : C <= 64 ? 8 : C <= 128 ? 4 : 2);
        grid = static_cast<int>(std::min<int64_t>(kCeBlocks, std::max<int64_t>(1, (n_rows + rows_per_block - 1) / rows_per_block)));
        if (C <= 32)
            TGCN_CE(4, 8);
        else if (C <= 64)
            TGCN_CE(8, 8);
        else if (C <= 128)
            TGCN_CE(16, 8);
        else
            TGCN_CE(32, 8);
    } else if (C <= 16)
        TGCN_CE(4, 4);
    else if (C <= 32)
        TGCN_CE(8, 4);
    else if (C <= 64)
        TGCN_CE(16, 4);
    else if (C <= 128)
        TGCN_CE(32, 4);
    else
        TGCN_CE(64, 4);
#undef TGCN_CE
    TGCN_HIP_CHECK(hipGetLastError());
    k_ce_final<<<1, 256, 0, s>>>(partial, grid, inv_count, loss);
    TGCN_HIP_CHECK(hipGetLastError());
    if (dbias) {
        if (colpart) return launch_colsum_final(colpart, grid, C, dbias, s);
        return launch_colsum(dlogits, ldd, n_rows, C, dbias, partial + kCeBlocks, colsum_blocks(n_rows), s);
    }
    return TGCN_OK;
}

int tgcn_masked_ce_pred(const float *logits, int64_t ld, int64_t n_rows, int n_classes,
                        const int64_t *target, const uint8_t *mask, float inv_count, float *loss,
                        float *dlogits, int64_t ldd, int64_t *pred, void *workspace,
                        size_t workspace_bytes, tgcn_stream stream) {
    return masked_ce_impl("tgcn_masked_ce", logits, ld, n_rows, n_classes, target, mask, inv_count, loss, dlogits, ldd,
                          nullptr, pred, workspace, workspace_bytes, tgcn_masked_ce_workspace_bytes(), stream);
}

size_t tgcn_masked_ce_grad_workspace_bytes(int64_t n_rows, int n_classes) {
    if (n_rows < 0 || n_classes <= 0) return 0;
    const size_t rows = std::max<size_t>(tgcn::kCeBlocks, static_cast<size_t>(tgcn::colsum_blocks(n_rows)));
    return sizeof(float) * (tgcn::kCeBlocks + rows * static_cast<size_t>(n_classes));
}

int tgcn_masked_ce_grad(const float *logits, int64_t ld, int64_t n_rows, int n_classes,
                        const int64_t *target, const uint8_t *mask, float inv_count, float *loss,
                        float *dlogits, int64_t ldd, float *dbias, int64_t *pred, void *workspace,
                        size_t workspace_bytes, tgcn_stream stream) {
    if (!dlogits || !dbias) {
        tgcn::set_error("tgcn_masked_ce_grad: dlogits and dbias are required (use tgcn_masked_ce for the loss alone)");
        return TGCN_E_INVALID;
    }
    return masked_ce_impl("tgcn_masked_ce_grad", logits, ld, n_rows, n_classes, target, mask, inv_count, loss, dlogits,
                          ldd, dbias, pred, workspace, workspace_bytes,
                          tgcn_masked_ce_grad_workspace_bytes(n_rows, n_classes), stream);
}

int tgcn_scale_by_device_scalar(float *x, int64_t n, const float *scale_dev, tgcn_stream stream) {
    using namespace tgcn;
    if (n < 0 || (n > 0 && !x) || !scale_dev) {
        set_error("tgcn_scale_by_device_scalar: bad argument (n=%lld)", (long long)n);
        return TGCN_E_INVALID;
    }
    if (n == 0) return TGCN_OK;
    const int grid = static_cast<int>(std::min<int64_t>(4096, (n + 255) / 256));
    k_scale_unless_one<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(x, n, scale_dev);
    TGCN_HIP_CHECK(hipGetLastError());
    return TGCN_OK;
}

static int adam_impl(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                     float *max_exp_avg_sq, int64_t n, double lr, double beta1, double beta2, double eps,
                     double weight_decay, int64_t step, int64_t *step_dev, float *scalars_dev,
                     tgcn_stream stream);

int tgcn_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                   float *max_exp_avg_sq, int64_t n, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int64_t step, tgcn_stream stream) {
    return adam_impl(param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                     step, nullptr, nullptr, stream);
}

int tgcn_adam_step_capturable(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                              float *max_exp_avg_sq, int64_t n, double lr, double beta1, double beta2,
                              double eps, double weight_decay, int64_t *step_dev, float *scalars_dev,
                              tgcn_stream stream) {
    if (!step_dev || !scalars_dev) {
        tgcn::set_error("tgcn_adam_step_capturable: step_dev / scalars_dev must be device pointers");
        return TGCN_E_INVALID;
    }
    return adam_impl(param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                     1, step_dev, scalars_dev, stream);
}

static int adam_impl(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                     float *max_exp_avg_sq, int64_t n, double lr, double beta1, double beta2, double eps,
                     double weight_decay, int64_t step, int64_t *step_dev, float *scalars_dev,
                     tgcn_stream stream) {
    using namespace tgcn;
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) {
        set_error("tgcn_adam_step: bad argument (n=%lld step=%lld)", (long long)n, (long long)step);
        return TGCN_E_INVALID;
    }
    const uintptr_t a = reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
                        reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq) |
                        reinterpret_cast<uintptr_t>(max_exp_avg_sq);
    if (a % 16 != 0) {
        set_error("tgcn_adam_step: buffers must be 16-byte aligned");
        return TGCN_E_INVALID;
    }
    if (n == 0 && step_dev == nullptr) return TGCN_OK;
    // hyper-parameters are Python floats (doubles) in torch: derive every constant in double and
    // round once, as torch does when it hands `1 - beta2`, `lr / bias_correction1`, ... to its kernels
    const double bc1 = 1.0 - std::pow(beta1, static_cast<double>(step));
    const double bc2 = 1.0 - std::pow(beta2, static_cast<double>(step));
    const float step_size = static_cast<float>(lr / bc1);
    const float w1 = static_cast<float>(1.0 - beta1), w2 = static_cast<float>(1.0 - beta2);
    const float b2f = static_cast<float>(beta2), epsf = static_cast<float>(eps), wdf = static_cast<float>(weight_decay);
    const float inv_bc2_sqrt = static_cast<float>(1.0 / std::sqrt(bc2));
    const int grid = static_cast<int>(std::min<int64_t>(8192, (n / 4 + 255) / 256 + 1));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (step_dev != nullptr) {
        k_adam_scalars<<<1, 1, 0, s>>>(step_dev, lr, beta1, beta2, scalars_dev);
        TGCN_HIP_CHECK(hipGetLastError());
        if (n == 0) return TGCN_OK;   // n = 0: only advance the device step (tgcn_spmm_adam applies the update)
    }
    // large tensors (W1: N x h) stream their state past the caches (measured: 2.9 -> 2.5 ms on 2 M x 200)
    const bool nt = n >= (int64_t(1) << 24);
#define TGCN_ADAM(AMS, NTV)                                                                                  \
    k_adam<AMS, NTV><<<grid, 256, 0, s>>>(param, grad, exp_avg, exp_avg_sq, AMS ? max_exp_avg_sq : nullptr, n, w1, \
                                          b2f, w2, epsf, wdf, step_size, inv_bc2_sqrt, scalars_dev)
    if (max_exp_avg_sq) {
        if (nt) TGCN_ADAM(true, true); else TGCN_ADAM(true, false);
    } else {
        if (nt) TGCN_ADAM(false, true); else TGCN_ADAM(false, false);
    }
#undef TGCN_ADAM
    TGCN_HIP_CHECK(hipGetLastError());
    return TGCN_OK;
}

}  // extern "C"
