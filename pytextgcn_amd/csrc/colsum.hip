// Column sums of a row-major [n_rows, F] fp32 matrix: db = sum over rows of dOut, the autograd of
// GCNConv's `out += bias` (PyG-1.6.3 gcn_conv.py forward, invoked from textgcn/lib/models.py:20 and
// differentiated at flat_amazon.py:105).  HBM-bound: reads n_rows*F*4 bytes once.
// Two passes, fixed order, no atomics: per-workgroup partials (LDS across the 4 waves), then 16 waves per
// 64 columns add the partial rows.
#include "common.h"

namespace tgcn {
namespace {

// VEC = 4: LPR lanes own one row (float4 each), a wave reads 64 / LPR rows per pass (F = 64: four rows of
// 256 B; F = 200: one row), four passes in flight per wave.  VEC = 1: one lane per column, one row per pass.
// Partial sums are combined in a fixed order: a lane over its own rows in row order, then the sub-groups of a
// wave, then the four waves.
template <int VEC, int LPR>
__global__ __launch_bounds__(256) void k_colsum_partial(const float *__restrict__ G, int64_t ldg,
                                                        int64_t n_rows, int F,
                                                        float *__restrict__ partial) {
    constexpr int RPW = 64 / LPR;      // rows per wave and pass
    constexpr int UN = 4;              // passes in flight
    __shared__ float red[4][64 * VEC];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane / LPR, sl = lane % LPR;
    const int col0 = blockIdx.y * (LPR * VEC);
    const int nb = gridDim.x;
    const int64_t rows_per = (n_rows + nb - 1) / nb;
    const int64_t r_begin = int64_t(blockIdx.x) * rows_per;
    const int64_t r_end = min(n_rows, r_begin + rows_per);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    const int c = col0 + sl * VEC;
    if (c < F) {
        int64_t r = r_begin + wave * RPW + sub;
        for (; r + int64_t(3) * 4 * RPW < r_end; r += int64_t(UN) * 4 * RPW) {
            if constexpr (VEC == 4) {
                float4 v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u)
                    v[u] = *reinterpret_cast<const float4 *>(G + (r + int64_t(u) * 4 * RPW) * ldg + c);
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    acc[0] += v[u].x;
                    acc[1] += v[u].y;
                    acc[2] += v[u].z;
                    acc[3] += v[u].w;
                }
            } else {
                float v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) v[u] = G[(r + int64_t(u) * 4 * RPW) * ldg + c];
#pragma unroll
                for (int u = 0; u < UN; ++u) acc[0] += v[u];
            }
        }
        for (; r < r_end; r += 4 * RPW) {
            const float *p = G + r * ldg + c;
            if constexpr (VEC == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(p);
                acc[0] += v.x;
                acc[1] += v.y;
                acc[2] += v.z;
                acc[3] += v.w;
            } else {
                acc[0] += *p;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[wave][lane * VEC + i] = acc[i];
    __syncthreads();
    if (wave == 0 && sub == 0 && c < F) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int sb = 0; sb < RPW; ++sb) s += red[w][(sb * LPR + sl) * VEC + i];
            partial[int64_t(blockIdx.x) * F + c + i] = s;
        }
    }
}

// 16 waves x 64 columns per workgroup: wave w adds partial rows w, w+16, ... (eight loads in flight);
// the 16 sums are combined through LDS in wave order
__global__ __launch_bounds__(1024) void k_colsum_final(const float *__restrict__ partial, int nb, int F,
                                                       float *__restrict__ out) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (f < F) {
        int b = wave;
        for (; b + 7 * 16 < nb; b += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[int64_t(b + u * 16) * F + f];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nb; b += 16) s += partial[int64_t(b) * F + f];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && f < F) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        out[f] = t;
    }
}

}  // namespace

int colsum_blocks(int64_t n_rows) {
    int64_t nb = (n_rows + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;      // one resident round of 256-thread workgroups on 256 CUs
    return static_cast<int>(nb);
}

int launch_colsum(const float *G, int64_t ldg, int64_t n_rows, int F, float *out, float *partial,
                  int n_blocks, hipStream_t stream) {
    const bool vec4 = (F % 4 == 0) && (ldg % 4 == 0) && (reinterpret_cast<uintptr_t>(G) % 16 == 0);
    if (vec4 && F <= 64) {
        dim3 grid(n_blocks, 1);
        k_colsum_partial<4, 16><<<grid, 256, 0, stream>>>(G, ldg, n_rows, F, partial);
    } else if (vec4 && F <= 128) {
        dim3 grid(n_blocks, 1);
        k_colsum_partial<4, 32><<<grid, 256, 0, stream>>>(G, ldg, n_rows, F, partial);
    } else if (vec4) {
        dim3 grid(n_blocks, (F + 255) / 256);
        k_colsum_partial<4, 64><<<grid, 256, 0, stream>>>(G, ldg, n_rows, F, partial);
    } else {
        dim3 grid(n_blocks, (F + 63) / 64);
        k_colsum_partial<1, 64><<<grid, 256, 0, stream>>>(G, ldg, n_rows, F, partial);
    }
    TGCN_HIP_CHECK(hipGetLastError());
    k_colsum_final<<<(F + 63) / 64, 1024, 0, stream>>>(partial, n_blocks, F, out);
    TGCN_HIP_CHECK(hipGetLastError());
    return TGCN_OK;
}

// second pass alone, for producers that leave their own per-workgroup partial rows (k_masked_ce)
int launch_colsum_final(const float *partial, int n_partial, int F, float *out, hipStream_t stream) {
    k_colsum_final<<<(F + 63) / 64, 1024, 0, stream>>>(partial, n_partial, F, out);
    TGCN_HIP_CHECK(hipGetLastError());
    return TGCN_OK;
}

}  // namespace tgcn

extern "C" {

size_t tgcn_colsum_workspace_bytes(int64_t n_rows, int F) {
    if (n_rows < 0 || F <= 0) return 0;
    return sizeof(float) * static_cast<size_t>(tgcn::colsum_blocks(n_rows)) * static_cast<size_t>(F);
}

int tgcn_colsum(const float *G, int64_t ldg, int64_t n_rows, int F, float *out, void *workspace,
                size_t workspace_bytes, tgcn_stream stream) {
    using namespace tgcn;
    if (!out || F <= 0 || n_rows < 0 || (n_rows > 0 && (!G || ldg < F))) {
        set_error("tgcn_colsum: bad argument (n_rows=%lld F=%d ldg=%lld)", (long long)n_rows, F,
                  (long long)ldg);
        return TGCN_E_INVALID;
    }
    const size_t need = tgcn_colsum_workspace_bytes(n_rows, F);
    if (!workspace || workspace_bytes < need) {
        set_error("tgcn_colsum: workspace of %zu bytes given, %zu needed", workspace_bytes, need);
        return TGCN_E_WORKSPACE;
    }
    return launch_colsum(G, ldg, n_rows, F, out, static_cast<float *>(workspace),
                         colsum_blocks(n_rows), static_cast<hipStream_t>(stream));
}

}  // extern "C"
