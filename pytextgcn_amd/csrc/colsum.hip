// Column sums of a row-major [n_rows, F] fp32 matrix: db = sum over rows of dOut, the autograd of
// GCNConv's `out += bias` (PyG-1.6.3 gcn_conv.py forward, invoked from textgcn/lib/models.py:20 and
// differentiated at flat_amazon.py:105).  HBM-bound: reads n_rows*F*4 bytes once.
// Two passes, fixed order, no atomics: per-workgroup partials (LDS across the 4 waves), then one
// thread per column adds the partials in block order.
#include "common.h"

namespace tgcn {
namespace {

template <int VEC>
__global__ __launch_bounds__(256) void k_colsum_partial(const float *__restrict__ G, int64_t ldg,
                                                        int64_t n_rows, int F,
                                                        float *__restrict__ partial) {
    __shared__ float red[4][64 * VEC];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int col0 = blockIdx.y * (64 * VEC);
    const int nb = gridDim.x;
    const int64_t rows_per = (n_rows + nb - 1) / nb;
    const int64_t r_begin = int64_t(blockIdx.x) * rows_per;
    const int64_t r_end = min(n_rows, r_begin + rows_per);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    const int c = col0 + lane * VEC;
    if (c < F) {
        for (int64_t r = r_begin + wave; r < r_end; r += 4) {
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
    if (wave == 0 && c < F) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int k = lane * VEC + i;
            partial[int64_t(blockIdx.x) * F + c + i] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        }
    }
}

// 4 waves x 64 columns per workgroup: wave w adds partial rows w, w+4, ...; LDS combine in wave order
__global__ __launch_bounds__(256) void k_colsum_final(const float *__restrict__ partial, int nb, int F,
                                                      float *__restrict__ out) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (f < F)
        for (int b = wave; b < nb; b += 4) s += partial[int64_t(b) * F + f];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && f < F) out[f] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

}  // namespace

int colsum_blocks(int64_t n_rows) {
    int64_t nb = (n_rows + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    return static_cast<int>(nb);
}

int launch_colsum(const float *G, int64_t ldg, int64_t n_rows, int F, float *out, float *partial,
                  int n_blocks, hipStream_t stream) {
    const bool vec4 = (F % 4 == 0) && (ldg % 4 == 0) && (reinterpret_cast<uintptr_t>(G) % 16 == 0);
    if (vec4) {
        dim3 grid(n_blocks, (F + 255) / 256);
        k_colsum_partial<4><<<grid, 256, 0, stream>>>(G, ldg, n_rows, F, partial);
    } else {
        dim3 grid(n_blocks, (F + 63) / 64);
        k_colsum_partial<1><<<grid, 256, 0, stream>>>(G, ldg, n_rows, F, partial);
    }
    TGCN_HIP_CHECK(hipGetLastError());
    k_colsum_final<<<(F + 63) / 64, 256, 0, stream>>>(partial, n_blocks, F, out);
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
