// EXPERIMENT (not part of libtgcn.so): BASELINE.json's north_star names "X-tiles staged in LDS" for the CSR SpMM.
// Round 1 built that into the product kernel, measured it slower and removed it without keeping a log;
// this file re-creates the experiment in isolation so that the claim has one: the SAME simplified gather kernel --
// one wavefront per work item (rows packed to ~384 non-zeros, long rows cut into pieces whose partial sums go to a
// carry buffer), 8 gathered rows in flight, float4 lanes -- with the rows of the R most-gathered columns of X staged
// in LDS by every workgroup (R = 0: plain kernel).  Two workgroup shapes: persistent 16-wave workgroups (one per CU,
// up to 160 rows = 128 KB of LDS) and 4-wave workgroups (8 per CU, up to 24 rows).
// A column that lives in LDS is marked by the sign bit of its stored id (slot in the low bits).
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Item { int32_t row_begin, row_end, nnz_begin, nnz_end; };   // row_end < 0: piece of row row_begin -> carry[-row_end-1]

template <int WAVES, bool PERSIST>
__global__ __launch_bounds__(64 * WAVES) void k_gather(const Item *__restrict__ items, int n_items,
                                                       const int32_t *__restrict__ rowptr, const int2 *__restrict__ cv,
                                                       const float *__restrict__ X, int64_t ldx, int F,
                                                       const int32_t *__restrict__ lds_cols, int R,
                                                       float *__restrict__ Y, int64_t ldy, float *__restrict__ carry) {
    extern __shared__ __align__(16) float tile[];                  // [R][F]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = F / 4;
    for (int i = threadIdx.x; i < R * nvec; i += 64 * WAVES) {
        const int r = i / nvec, j = i % nvec;
        reinterpret_cast<float4 *>(tile)[r * nvec + j] =
            *reinterpret_cast<const float4 *>(X + int64_t(lds_cols[r]) * ldx + 4 * j);
    }
    if (R > 0) __syncthreads();
    const int lc = 4 * min(lane, nvec - 1);
    const int stride = PERSIST ? gridDim.x * WAVES : 0;
    int id = blockIdx.x * WAVES + wave;
    do {
        if (id >= n_items) break;
        const Item it = items[id];
        const bool seg = it.row_end < 0;
        const int r_end = seg ? it.row_begin + 1 : it.row_end;
        for (int r = it.row_begin; r < r_end; ++r) {
            const int jb = seg ? it.nnz_begin : rowptr[r], je = seg ? it.nnz_end : rowptr[r + 1];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j0 = jb; j0 < je; j0 += 64) {
                const int n = min(64, je - j0);
                int2 mine = make_int2(0, 0);
                if (lane < n) mine = cv[j0 + lane];
                for (int u0 = 0; u0 < n; u0 += 8) {
                    float4 x[8];
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int q = min(u0 + u, n - 1);
                        const int c = __builtin_amdgcn_readlane(mine.x, q);
                        v[u] = (u0 + u < n) ? __int_as_float(__builtin_amdgcn_readlane(mine.y, q)) : 0.f;
                        if (c < 0)                                   // wave-uniform: the row sits in LDS
                            x[u] = *reinterpret_cast<const float4 *>(tile + (c & 0x7fffffff) * F + lc);
                        else
                            x[u] = *reinterpret_cast<const float4 *>(X + int64_t(c) * ldx + lc);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        acc.x += v[u] * x[u].x;
                        acc.y += v[u] * x[u].y;
                        acc.z += v[u] * x[u].z;
                        acc.w += v[u] * x[u].w;
                    }
                }
            }
            if (lane < nvec) {
                float *dst = seg ? carry + int64_t(-it.row_end - 1) * F : Y + int64_t(r) * ldy;
                *reinterpret_cast<float4 *>(dst + 4 * lane) = acc;
            }
        }
        id += stride;
    } while (PERSIST);
}

extern "C" int exp_gather(int waves, int persist, int grid, const void *items, int n_items, const void *rowptr,
                          const void *cv, const float *X, int64_t ldx, int F, const void *lds_cols, int R, float *Y,
                          int64_t ldy, float *carry, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = sizeof(float) * size_t(R) * F;
#define GO(W, P)                                                                                                    \
    do {                                                                                                            \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gather<W, P>),                                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)   \
            return -2;                                                                                              \
        k_gather<W, P><<<grid, 64 * W, lds, s>>>(static_cast<const Item *>(items), n_items,                         \
                                                 static_cast<const int32_t *>(rowptr), static_cast<const int2 *>(cv), \
                                                 X, ldx, F, static_cast<const int32_t *>(lds_cols), R, Y, ldy, carry); \
    } while (0)
    if (waves == 16 && persist) GO(16, true);
    else if (waves == 4 && !persist) GO(4, false);
    else if (waves == 4 && persist) GO(4, true);
    else return -1;
#undef GO
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
