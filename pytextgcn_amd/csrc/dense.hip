// Tall-skinny fp32 GEMMs of the GCN layers on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// bit-for-bit an fma chain; gfx950 has no xf32/TF32 path, and the 1e-5 parity bar forbids bf16).
//
// They replace `torch.matmul(x, self.weight)` of PyG-1.6.3 GCNConv.forward (k5 of SURVEY.md 2a;
// reference call site textgcn/lib/models.py:20) for the DENSE layers and its autograd (b1-b4):
//     tgcn_gemm_nn   C[N, n]  = A[N, k] @ B[k, n]          XW2  = H1d @ W2      (k = h, n = C)
//     tgcn_gemm_nt   C[N, n]  = A[N, k] @ B[n, k]^T        dH1d = dXW2 @ W2^T   (k = C, n = h)
//     tgcn_gemm_tn   C[k, n]  = A[N, k]^T @ G[N, n]        dW2  = H1d^T @ dXW2  (reduction over N)
// N is the node count (millions), k and n are layer widths (<= 256): every operand row is read
// exactly once, so the kernels are co-bound by HBM (N*(k+n)*4 bytes) and by the fp32 MFMA rate
// (2*N*k*n flop at 157 TF/s peak); at N = 2 M, k = 200, n = 64 both limits are ~0.35 ms.
//
// Fragment maps (cdna_hip_programming.md section 3): for 32x32x2, lane l feeds A[i = l&31][k = l>>5]
// and B[k = l>>5][j = l&31]; the accumulator register r of lane l is C[(r&3) + 8*(r>>2) + 4*(l>>5)]
// [l&31].  The sum over k is order-independent, which the nn/nt kernels use to load 4 consecutive k
// of one row with ONE float4 per lane (lanes 0-31 take k = 8q..8q+3, lanes 32-63 k = 8q+4..8q+7) and
// spend it over 4 MFMA steps; the matching rows of the small operand sit in LDS.
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <utility>

#include "common.h"

namespace tgcn {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGroupCols = 128;  // widest column group of one launch (4 tiles of 32): k x 128 floats of LDS
constexpr int kGroupK = 256;     // longest reduction of one launch: 256 x 128 x 4 B = 128 KB of the 160 KB LDS

// Where one launch sits inside a product whose small operand exceeds the LDS (or the accumulator registers): widths
// above 128 columns run as column groups, reductions above 256 as k chunks that accumulate into C.  Only the
// dropout hash needs to know (it is a function of the GLOBAL element position).
struct Place {
    int col0;    // first result column of this launch
    int k0;      // first index of the reduction
    int accum;   // C += instead of C =
};

// Two workgroups per CU (a 256-register budget per lane) for the nt kernels with a masked result and / or column
// sums: hipcc then keeps the accumulator tiles in the same register file as everything else instead of splitting
// them off into AGPRs (186 + 112 registers = one wave per SIMD otherwise).  Measured at the c4 shapes
// (tools/ab_dense.py, interleaved): nt + mask 0.88 -> 0.85 ms, nt + column sums 0.77 -> 0.72, nt + mask + column
// sums 0.89 -> 0.81; the plain nt product is faster with the split (0.755 against 0.82 ms) and keeps it, the nn
// product does not care (0.55 ms either way).
#ifndef TGCN_TALL_MIN_BLOCKS
#define TGCN_TALL_MIN_BLOCKS 2
#endif

// Inverted dropout fused into the GEMMs around it (the dropout between the GCN layers,
// textgcn/lib/models.py:23): element (row, col) of the [N x ld] activation is kept with probability
// 1 - p and scaled by 1 / (1 - p).  The keep decision is a stateless hash of (seed, row, col)
// (murmur3-style mixing steps), so the forward GEMM (mask on its A operand), the weight-gradient GEMM (mask on
// its A operand) and the input-gradient GEMM (mask on its result) regenerate the SAME mask from the
// 8-byte seed instead of storing N x ld bytes.  The seed is read from device memory (graph capture).
struct Drop {
    const uint64_t *seed;  // device pointer
    uint32_t thresh;       // keep iff hash >= thresh;  thresh = p * 2^32
    float scale;           // 1 / (1 - p)
    int ld;                // logical width of the masked matrix
    uint32_t *bits;        // optional record of the keep decisions (layout: drop_bits_word / _shift below); the nn
    int64_t bits_stride;   // product writes it, the tn product reads it instead of hashing every element again
    // Which ROW of the mask a row of this call's matrix is (tgcn_set_dropout_row_keys): row i hashes as i + key0 when
    // i < key_split, else as i + key1.  All zero: the row index itself.  The 1-D partition uses it so that a hub row's
    // mask is a function of its position in the GATHERED hub block -- the same on the rank that owns the row (its rows
    // [0, hp) hash as rank * hp + i) and on every rank that holds a partial sum of it (pytextgcn_amd/sharded.py).
    int64_t key_split, key0, key1;
};

#ifndef TGCN_NT_PIPE
#define TGCN_NT_PIPE 1                // 0: A/B builds with the class-width nt product on k_gemm_tall (tools/build_variant.py)
#endif
#ifndef TGCN_DROP_ROW_KEYS
#define TGCN_DROP_ROW_KEYS 1          // 0: A/B builds without the row-key mapping (tools/build_variant.py)
#endif
__device__ __forceinline__ int64_t drop_row_of(const Drop &d, int64_t row) {
#if TGCN_DROP_ROW_KEYS
    return row + (row < d.key_split ? d.key0 : d.key1);
#else
    return row;
#endif
}

// The recorded keep mask of a [N x k] operand: 4 bits per (row, 8-column step q, lane half h) -- the 4 consecutive
// columns 8 q + 4 h .. + 3 a lane of the nn kernel multiplies in step q -- packed 8 steps to a word; a row holds the
// words of half 0, then those of half 1 (wph each).  Column c: q = c / 8, h = (c / 4) & 1, bit 4 (q % 8) + (c & 3) of
// word h * wph + q / 8.
__host__ __device__ __forceinline__ int drop_bits_wph(int k) { return ((k + 7) / 8 + 7) / 8; }

// The hash is split so that the expensive part is paid once per ROW and lane, not once per element: a row key
// (murmur3 mixing of the 64-bit row index with the seed) and, per element, key + col * golden-ratio constant
// through a two-multiply finaliser.  (The first version hashed row * ld + col per element: 4 quarter-rate
// 32-bit multiplies and a 64-bit multiply-add each; profiles/r02_pmc_gemm_c4.md: 4x the vector-ALU instructions
// of the plain kernels.)  All three GEMMs call the same two functions, so they regenerate the same mask.
__device__ __forceinline__ uint32_t drop_row_key(uint32_t s_lo, uint32_t s_hi, int64_t row) {
    uint32_t h = uint32_t(row) ^ s_lo;
    h *= 0xcc9e2d51u;
    h = (h << 15) | (h >> 17);
    h *= 0x1b873593u;
    h ^= uint32_t(uint64_t(row) >> 32) + s_hi;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    return h;
}

// the same key for a row index that is the same in every lane of the wave (the LDS-staged tn kernels: a wave copies whole
// rows): forced into scalar registers, so that the three multiplies run on the scalar unit instead of as quarter-rate
// vector instructions in all 64 lanes
__device__ __forceinline__ uint32_t drop_row_key_uniform(uint32_t s_lo, uint32_t s_hi, int64_t row) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(row)));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(uint64_t(row) >> 32)));
    return drop_row_key(s_lo, s_hi, int64_t((uint64_t(hi) << 32) | lo));
}

__device__ __forceinline__ uint32_t drop_col_term(int col) { return uint32_t(col) * 0x9E3779B1u; }

__device__ __forceinline__ bool drop_keep(uint32_t row_key, uint32_t col_term, const Drop &d) {
    uint32_t h = row_key + col_term;
    h ^= h >> 15;
    h *= 0x2c1b3c6du;
    h ^= h >> 12;
    h *= 0x297a2d39u;
    h ^= h >> 15;
    return h >= d.thresh;
}

__device__ __forceinline__ float drop_elem(float v, uint32_t row_key, uint32_t col_term, const Drop &d) {
    return drop_keep(row_key, col_term, d) ? v * d.scale : 0.f;
}

// ---------------------------------------------------------------------------------------------
// C[N, n] = A[N, k] @ Bs, where Bs[kk][j] is the small operand staged in LDS as [kpad][npad].
// One wave per 32 rows, NT tiles of 32 columns each (npad = 32*NT).  TRANS_B selects how the small
// operand is read from memory: B[k][n] (nn) or B[n][k] (nt).
// ---------------------------------------------------------------------------------------------
// DROP: nn (TRANS_B = false) masks the A operand, nt (TRANS_B = true) masks the result.
// COLSUM: the workgroup also leaves the column sums of the rows of C it wrote (after the mask) in
// colpart[blockIdx.x][0..n) -- for the nt product that is the bias gradient of the layer whose output the
// product differentiates (db1 = column sums of dH1), taken from the accumulators instead of a second pass over C.
// BITS (nt with DROP): the mask on the result comes from the record the forward product left (Drop::bits) instead of 16 NT
// hashes per lane and block -- see the epilogue.
template <int NT, bool TRANS_B, bool K8, int NQ, bool DROP, bool COLSUM = false, bool BITS = false>
__global__ __launch_bounds__(256, (TRANS_B && NT <= 7 && NQ != 28 && (DROP || COLSUM)) ? TGCN_TALL_MIN_BLOCKS : 1) void k_gemm_tall(const float *__restrict__ A, int64_t lda,
                                                   const float *__restrict__ B, int64_t ldb,
                                                   float *__restrict__ C, int64_t ldc, int64_t N,
                                                   int k, int n, const Drop drop, float *__restrict__ colpart,
                                                   const Place place) {
    extern __shared__ float lds[];  // [kpad][npad] (+ [4][npad] with COLSUM)
    // the fully unrolled kernels (NQ > 0) only ever run whole products: their placement folds to constants, which
    // keeps their register budget where it was (they sit at the limit: an accumulator set + the asm load ring)
    // (the column groups of a wide nt product, NT <= 4, have room for their first column)
    const int p_col0 = (NQ > 0 && NT > 4) ? 0 : place.col0, p_k0 = NQ > 0 ? 0 : place.k0;
    const bool p_accum = NQ > 0 ? false : place.accum != 0;
    float csum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) csum[t] = 0.f;
    uint32_t s_lo = 0, s_hi = 0;
    if constexpr (DROP) {
        const uint64_t sd = *drop.seed;
        s_lo = uint32_t(sd);
        s_hi = uint32_t(sd >> 32);
    }
    constexpr int npad = 32 * NT;
    const int kpad = (k + 7) & ~7;
    // stage the small operand, zero padded
    for (int idx = threadIdx.x; idx < kpad * npad; idx += blockDim.x) {
        const int kk = idx / npad, j = idx % npad;
        float v = 0.f;
        if (kk < k && j < n) v = TRANS_B ? B[int64_t(j) * ldb + kk] : B[int64_t(kk) * ldb + j];
        lds[idx] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, half = lane >> 5;
    const int64_t n_blocks = (N + 31) / 32;
    const int nq = kpad / 8;
    for (int64_t blk = int64_t(blockIdx.x) * 4 + wave; blk < n_blocks; blk += int64_t(gridDim.x) * 4) {
        const int64_t row = blk * 32 + r;
        uint32_t a_key = 0;                    // dropout on the A operand: one row per lane and block
        uint32_t mword = 0;                    // the keep decisions of up to 8 steps, on their way to drop.bits
        uint32_t *mrow = nullptr;
        if constexpr (DROP && !TRANS_B) {
            a_key = drop_row_key(s_lo, s_hi, drop_row_of(drop, row));
            if (drop.bits) mrow = drop.bits + row * drop.bits_stride + half * drop_bits_wph(k);
        }
        // BITS: this block's slice of the recorded mask, 32 rows x 8 words = 1 KB: lane l fetches the 4 words of half l & 1
        // of row l / 2 now (one 16-byte load, in flight under the MFMAs) and parks them in the wave's LDS slice in the epilogue
        uint4 mq = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (BITS) {
            const int64_t mrow_ = blk * 32 + (lane >> 1);
            if (mrow_ < N) mq = *reinterpret_cast<const uint4 *>(drop.bits + mrow_ * drop.bits_stride + 4 * (lane & 1));
        }
        // rows past the end shadow the last row: loads stay in bounds, their results are not stored
        const float *arow = A + std::min(row, N - 1) * lda + 4 * half;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        if (p_accum) {                         // a later k chunk: continue the sums the earlier launches left in C
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = 32 * t + r;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int64_t orow = blk * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                    if (col < n && orow < N) acc[t][i] = C[orow * ldc + col];
                }
            }
        }
        auto load_a = [&](int q) -> float4 {
            if (q >= nq) return make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (K8) {
                return *reinterpret_cast<const float4 *>(arow + 8 * q);
            } else {
                const int k0 = 8 * q + 4 * half;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + 3 < k) {
                    v = *reinterpret_cast<const float4 *>(arow + 8 * q);
                } else {
                    if (k0 < k) v.x = arow[8 * q];
                    if (k0 + 1 < k) v.y = arow[8 * q + 1];
                    if (k0 + 2 < k) v.z = arow[8 * q + 2];
                }
                return v;
            }
        };
        const float *bbase = lds + (4 * half) * npad + r;
        if constexpr (NQ > 0) {
            // k = 8 NQ known at compile time: the k-loop is fully unrolled, so the A pieces can sit in
            // an 8-deep ring of INLINE-ASM loads with counted waits and no loop back edge between an
            // issue and its use (hipcc copies loop-carried asm outputs, which would read registers
            // whose load is still in flight).  hipcc itself waits vmcnt(0) on any prefetch it can
            // see, leaving 8 MFMAs = 512 cycles of cover against ~3000 cycles of HBM latency; here
            // seven pieces (56 MFMAs) stay in flight.  The wait takes the ring slot as an in/out
            // operand so that no consumer can be scheduled above it.
            f32x4 ring[8];
            // !K8 (k is not a multiple of 8; the caller guarantees a row stride >= round_up(k, 4)): the LAST piece of a
            // lane's row may start at or past column k -- it is then loaded from the row's first columns instead (an
            // address that exists) and zeroed below; a piece that starts before k ends inside the row's stride
            const float *alast = (K8 || 8 * (NQ - 1) + 4 * half < k) ? arow + 8 * (NQ - 1) : arow;
#define TGCN_LDA(slot, q) \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ring[slot]) : "v"((q) == NQ - 1 ? alast : arow + 8 * (q)) : "memory")
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < NQ) TGCN_LDA(j, j);
            float bfr[2][4][NT];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int t = 0; t < NT; ++t) bfr[0][s4][t] = bbase[s4 * npad + 32 * t];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + 1 < NQ) {
                    const float *bq = bbase + (8 * (q + 1)) * npad;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int t = 0; t < NT; ++t) bfr[(q + 1) & 1][s4][t] = bq[s4 * npad + 32 * t];
                }
                if (q + 7 < NQ)
                    asm volatile("s_waitcnt vmcnt(7)" : "+v"(ring[q & 7])::"memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[q & 7])::"memory");
                const f32x4 a = ring[q & 7];
                float av[4] = {a[0], a[1], a[2], a[3]};
                if constexpr (!K8) {
                    if (q == NQ - 1) {                   // columns >= k of the last step: whatever was read there counts as 0
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
                            if (8 * q + 4 * half + s4 >= k) av[s4] = 0.f;
                    }
                }
                if constexpr (DROP && !TRANS_B) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const bool keep = drop_keep(a_key, drop_col_term(p_k0 + 8 * q + 4 * half + s4), drop);
                        av[s4] = keep ? av[s4] * drop.scale : 0.f;
                        mword |= keep ? (1u << (4 * (q & 7) + s4)) : 0u;
                    }
                    if ((q & 7) == 7 || q + 1 == NQ) {         // a word of the recorded mask is complete
                        if (drop.bits && row < N) mrow[q >> 3] = mword;
                        mword = 0;
                    }
                }
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s4], bfr[q & 1][s4][t], acc[t], 0, 0, 0);
                if (q + 8 < NQ) TGCN_LDA(q & 7, q + 8);
            }
#undef TGCN_LDA
        } else {
        // B fragments of step group q: rows 8q + 4 half + s, columns r + 32 t; software pipeline:
        // three A pieces in flight, next group's B fragments read from LDS under the current MFMAs.
        // (A static 4-deep ring with the loop unrolled by 4 measured slower: 0.81 vs 0.73 ms.)
        float bcur[4][NT], bnxt[4][NT];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int t = 0; t < NT; ++t) bcur[s4][t] = bbase[s4 * npad + 32 * t];
        float4 a0 = load_a(0), a1 = load_a(1), a2 = load_a(2);
        for (int q = 0; q < nq; ++q) {
            const float4 a3 = load_a(q + 3);
            if (q + 1 < nq) {
                const float *bq = bbase + (8 * (q + 1)) * npad;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int t = 0; t < NT; ++t) bnxt[s4][t] = bq[s4 * npad + 32 * t];
            }
            float av[4] = {a0.x, a0.y, a0.z, a0.w};
            if constexpr (DROP && !TRANS_B) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const bool keep = drop_keep(a_key, drop_col_term(p_k0 + 8 * q + 4 * half + s4), drop);
                    av[s4] = keep ? av[s4] * drop.scale : 0.f;
                    mword |= keep ? (1u << (4 * (q & 7) + s4)) : 0u;
                }
                if ((q & 7) == 7 || q + 1 == nq) {
                    if (drop.bits && row < N) mrow[q >> 3] = mword;
                    mword = 0;
                }
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s4], bcur[s4][t], acc[t], 0, 0, 0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int t = 0; t < NT; ++t) bcur[s4][t] = bnxt[s4][t];
            a0 = a1;
            a1 = a2;
            a2 = a3;
        }
        }
        if constexpr (BITS) {
            // The mask of the result from the record: column c = 32 t + r of a row is bit 16 (t & 1) + 4 (r / 8) + (r & 3) of
            // word t / 2 of half (r / 4) & 1 of that row (drop_bits_wph).  An element costs a 4-byte LDS read, a multiply, a
            // 1-bit signed field extract and an AND -- where the hash costs ~14 vector instructions.
            uint32_t *mw = reinterpret_cast<uint32_t *>(lds + ((k + 7) & ~7) * npad + (COLSUM ? 4 * npad : 0)) + wave * 256;
            *reinterpret_cast<uint4 *>(mw + 4 * lane) = mq;
            __builtin_amdgcn_wave_barrier();             // the slice belongs to this wave: its LDS operations run in order
            const int lane_bit = 4 * (r >> 3) + (r & 3);
            const uint32_t *mlane = mw + 4 * ((r >> 2) & 1);
            const int64_t row0b = blk * 32 + 4 * half;
            float *crowb = C + row0b * ldc + r;
            const bool whole_b = blk * 32 + 32 <= N;
            // Tiles outer, rows inner -- the order of the unmasked epilogue: consecutive stores of a wave go to DIFFERENT rows.
            // (Rows outer, with one 16-byte LDS read per row for all tiles, measured 0.09 ms slower before any masking
            // arithmetic: seven consecutive 128-byte pieces of one row queue up on the same channel.)  The word of (row,
            // t / 2) is a 4-byte LDS read at a constant offset from the lane's base.
            const uint32_t *mrow = mlane + 32 * half;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = 32 * t + r;
                if (col < n) {
                    const int bit = lane_bit + 16 * (t & 1);
                    if (whole_b) {                       // wave-uniform: only the last block of the matrix is ragged
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int ro = (i & 3) + 8 * (i >> 2);
                            const uint32_t keep = uint32_t(__builtin_amdgcn_sbfe(int(mrow[8 * ro + (t >> 1)]), bit, 1));
                            const float out = __uint_as_float(__float_as_uint(acc[t][i] * drop.scale) & keep);
                            crowb[int64_t(ro) * ldc + 32 * t] = out;
                            if constexpr (COLSUM) csum[t] += out;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int ro = (i & 3) + 8 * (i >> 2);
                            const uint32_t keep = uint32_t(__builtin_amdgcn_sbfe(int(mrow[8 * ro + (t >> 1)]), bit, 1));
                            const float out = __uint_as_float(__float_as_uint(acc[t][i] * drop.scale) & keep);
                            if (row0b + ro < N) {
                                crowb[int64_t(ro) * ldc + 32 * t] = out;
                                if constexpr (COLSUM) csum[t] += out;
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);       // one tile at a time
            }
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        // C[(i&3) + 8*(i>>2) + 4*half][32 t + r]
        uint32_t c_key[16];                    // dropout on the result: the lane's 16 rows, shared by all tiles
        if constexpr (DROP && TRANS_B) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                c_key[i] = drop_row_key(s_lo, s_hi, drop_row_of(drop, blk * 32 + (i & 3) + 8 * (i >> 2) + 4 * half));
        }
        // The lane's first row and its address are formed ONCE per block; element (i, t) then sits at the wave-uniform
        // offset ((i&3) + 8*(i>>2)) * ldc + 32 t from it (scalar arithmetic, one 64-bit add per store).  Round 4: written
        // as `C[orow * ldc + col]` with a 64-bit `orow < N` test per element, hipcc emitted two quarter-rate 32-bit
        // multiplies, a 64-bit multiply-add and an exec-mask branch in front of EVERY one of the 16 NT stores of a block --
        // 5.8 vector-ALU instructions per MFMA in the plain nt kernel (profiles/r04_pmc_dense.md), and on this part vector-ALU
        // issue time adds to the matrix pipe's instead of hiding under it.  (Variants tried on top and dropped because the
        // asm-ring kernels then spill, which the build refuses: row keys handed round with wave shuffles instead of 16
        // hashes per lane; scalar row bases + one 32-bit lane offset, which hipcc does not turn into saddr-form stores.)
        const int64_t row0 = blk * 32 + 4 * half;
        float *crow = C + row0 * ldc + r;
        const bool whole_blk = blk * 32 + 32 <= N;          // wave-uniform: only the last block of the matrix is ragged
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = 32 * t + r;
            if (col < n) {
                const uint32_t cterm = drop_col_term(p_col0 + col);
                if (whole_blk) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float out = acc[t][i];
                        if constexpr (DROP && TRANS_B) out = drop_elem(out, c_key[i], cterm, drop);
                        crow[int64_t((i & 3) + 8 * (i >> 2)) * ldc + 32 * t] = out;
                        if constexpr (COLSUM) csum[t] += out;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int ro = (i & 3) + 8 * (i >> 2);
                        float out = acc[t][i];
                        if constexpr (DROP && TRANS_B) out = drop_elem(out, c_key[i], cterm, drop);
                        if (row0 + ro < N) {
                            crow[int64_t(ro) * ldc + 32 * t] = out;
                            if constexpr (COLSUM) csum[t] += out;
                        }
                    }
                }
            }
            // one tile at a time: the fence keeps hipcc from reading all NT accumulator tiles out at once
            if constexpr (TRANS_B && (DROP || COLSUM)) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (COLSUM) {
        // a lane's rows in block order, then the two halves of the wave, then the four waves
        float *cl = lds + ((k + 7) & ~7) * npad;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float v = csum[t] + __shfl_xor(csum[t], 32, 64);
            if (half == 0) cl[wave * npad + 32 * t + r] = v;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += blockDim.x)
            colpart[int64_t(blockIdx.x) * n + j] = (cl[j] + cl[npad + j]) + (cl[2 * npad + j] + cl[3 * npad + j]);
    }
}

// ---------------------------------------------------------------------------------------------
// The input-gradient product of the class-width layer, C[N, n] = A[N, 64] @ B[n, 64]^T (n <= 32 NT), with the operand of
// block b + 1 in flight under the MFMAs of block b (round 5).
//
// k = 64 is ONE ring of eight 16-byte pieces per lane, so k_gemm_tall<.., NQ = 8, ..> issues all of a block's loads at the top
// of the block and its first MFMA waits out a whole memory latency -- with one wave per SIMD (the placement the unmasked
// kernels prefer) nothing covers it: ~6 k of the ~22 k cycles a block takes.  Here a wave owns TWO rings and its loop body
// two blocks: ring B is issued before block b0's k-loop and waited for (free: 14 k cycles later) before b0's epilogue, ring
// A for the block after next is issued before block b1's k-loop and waited for before b1's epilogue.  Every inline-asm
// load is therefore complete at the loop's back edge -- hipcc may copy loop-carried asm outputs there, which must not happen
// to a register whose load is still in flight (HISTORY 4.5) -- and the only exposed latency is the prologue's.  The waits
// sit BEFORE the epilogues because loads and stores share one counter (vmcnt) on this part: behind 112 stores no counted
// wait could single out the loads.  B fragments are fetched one s4-step ahead (2 x NT registers instead of 2 x 4 x NT), which
// pays for the second ring; accumulators, rings and fragments stay in one 256-register file (__launch_bounds__(256, 2):
// 224-250 registers, no spills -- the build checks).
// Epilogues: plain, COLSUM (column sums of the stored rows), and DROP with the mask read from the forward product's record
// (BITS; Drop::bits) -- bit for bit k_gemm_tall's results: same products, same summation orders.
// Measured at c4 (tools/ab_dense.py, profiles/r05_ab_dense.log): nt 0.585 -> 0.52 ms, nt + column sums 0.58 -> 0.535, nt +
// mask from the record + column sums 0.665 -> 0.59.  The same scheme for the hidden-width nn product (two rings of 25 pieces
// = 200 registers, plain loads so that the compiler may use the second half of the register file) was built and measured
// SLOWER (nn 0.525 -> 0.61 ms: 96 accumulator-file registers of copies) and removed.
// ---------------------------------------------------------------------------------------------
template <int NT, int NQ, bool TRANS_B, bool COLSUM, bool BITS>
__global__ __launch_bounds__(256, 2) void k_gemm_pipe(const float *__restrict__ A, int64_t lda,
                                                        const float *__restrict__ B, int64_t ldb,
                                                        float *__restrict__ C, int64_t ldc, int64_t N, int n,
                                                        const Drop drop, float *__restrict__ colpart) {
    extern __shared__ float lds[];          // [8 NQ][npad] (+ [4][npad] with COLSUM) (+ 4 x 256 mask words with BITS)
    constexpr int npad = 32 * NT, K = 8 * NQ;
    float csum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) csum[t] = 0.f;
    for (int idx = threadIdx.x; idx < K * npad; idx += blockDim.x) {
        const int kk = idx / npad, j = idx % npad;
        lds[idx] = j < n ? (TRANS_B ? B[int64_t(j) * ldb + kk] : B[int64_t(kk) * ldb + j]) : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, half = lane >> 5;
    const int64_t n_blocks = (N + 31) / 32;
    const int64_t stride = int64_t(gridDim.x) * 4;
    const float *bbase = lds + (4 * half) * npad + r;
    uint32_t *mw = nullptr;
    if constexpr (BITS) mw = reinterpret_cast<uint32_t *>(lds + K * npad + (COLSUM ? 4 * npad : 0)) + wave * 256;

    // rows past the end shadow the last row: loads stay in bounds, their results are not stored
    auto arow_of = [&](int64_t blk) __attribute__((always_inline)) { return A + std::min(std::min(blk, n_blocks - 1) * 32 + r, N - 1) * lda + 4 * half; };
#define TGCN_ISSUE(ring, ptr)                                                                                         \
    do {                                                                                                              \
        const float *p_ = (ptr);                                                                                      \
        _Pragma("unroll") for (int j_ = 0; j_ < NQ; ++j_)                                                             \
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ring[j_]) : "v"(p_ + 8 * j_) : "memory");           \
    } while (0)
    // the wait takes every slot of the ring as an in/out operand, so that no consumer can be scheduled above it
#define TGCN_R8(ring, o) "+v"(ring[o]), "+v"(ring[o + 1]), "+v"(ring[o + 2]), "+v"(ring[o + 3]), "+v"(ring[o + 4]), \
                         "+v"(ring[o + 5]), "+v"(ring[o + 6]), "+v"(ring[o + 7])
#define TGCN_ARRIVED(ring)                                                                                            \
    do {                                                                                                              \
        static_assert(NQ == 8, "a ring is the whole operand row block: eight 16-byte pieces per lane (k = 64)");       \
        asm volatile("s_waitcnt vmcnt(0)" : TGCN_R8(ring, 0)::"memory");                                              \
    } while (0)
    auto mask_of = [&](int64_t blk) __attribute__((always_inline)) {
        uint4 mq = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (BITS) {
            const int64_t mrow_ = blk * 32 + (lane >> 1);
            if (mrow_ < N) mq = *reinterpret_cast<const uint4 *>(drop.bits + mrow_ * drop.bits_stride + 4 * (lane & 1));
        }
        return mq;
    };
    auto kloop = [&](const f32x4 (&ring)[NQ], f32x16 (&acc)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        float bfr[2][NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bfr[0][t] = bbase[32 * t];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x4 a = ring[q];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int step = 4 * q + s4;
                if (step + 1 < 4 * NQ) {                  // the fragments of the next step, under this step's MFMAs
                    const float *bq = bbase + (8 * ((step + 1) >> 2) + ((step + 1) & 3)) * npad;
#pragma unroll
                    for (int t = 0; t < NT; ++t) bfr[(step + 1) & 1][t] = bq[32 * t];
                }
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s4], bfr[step & 1][t], acc[t], 0, 0, 0);
            }
        }
    };
    // RAGGED (only the last block of the matrix): rows past N are neither stored nor summed
    auto store_tiles = [&](auto ragged, int64_t blk, const f32x16 (&acc)[NT], const uint4 mq) __attribute__((always_inline)) {
        constexpr bool RAGGED = decltype(ragged)::value;
        const int64_t row0 = blk * 32 + 4 * half;
        float *crow = C + row0 * ldc + r;
        if constexpr (BITS) {
            // column c = 32 t + r of a row is bit 16 (t & 1) + 4 (r / 8) + (r & 3) of word t / 2 of half (r / 4) & 1
            *reinterpret_cast<uint4 *>(mw + 4 * lane) = mq;
            __builtin_amdgcn_wave_barrier();
            const int lane_bit = 4 * (r >> 3) + (r & 3);
            const uint32_t *mrow = mw + 4 * ((r >> 2) & 1) + 32 * half;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (32 * t + r < n) {
                    const int bit = lane_bit + 16 * (t & 1);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int ro = (i & 3) + 8 * (i >> 2);
                        const uint32_t keep = uint32_t(__builtin_amdgcn_sbfe(int(mrow[8 * ro + (t >> 1)]), bit, 1));
                        const float out = __uint_as_float(__float_as_uint(acc[t][i] * drop.scale) & keep);
                        if (!RAGGED || row0 + ro < N) {
                            crow[int64_t(ro) * ldc + 32 * t] = out;
                            if constexpr (COLSUM) csum[t] += out;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);       // one tile at a time
            }
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (32 * t + r < n) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int ro = (i & 3) + 8 * (i >> 2);
                        const float out = acc[t][i];
                        if (!RAGGED || row0 + ro < N) {
                            crow[int64_t(ro) * ldc + 32 * t] = out;
                            if constexpr (COLSUM) csum[t] += out;
                        }
                    }
                }
                if constexpr (COLSUM) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto epilogue = [&](int64_t blk, const f32x16 (&acc)[NT], const uint4 mq) __attribute__((always_inline)) {
        if (blk * 32 + 32 <= N)                            // wave-uniform
            store_tiles(std::false_type{}, blk, acc, mq);
        else
            store_tiles(std::true_type{}, blk, acc, mq);
    };

    f32x4 ringA[NQ], ringB[NQ];
    f32x16 acc[NT];
    int64_t blk = int64_t(blockIdx.x) * 4 + wave;
    if (blk < n_blocks) {
        TGCN_ISSUE(ringA, arow_of(blk));
        TGCN_ARRIVED(ringA);                               // the one exposed latency of this wave
    }
    while (blk < n_blocks) {
        const int64_t blk1 = blk + stride, blk2 = blk1 + stride;
        TGCN_ISSUE(ringB, arow_of(blk1));                  // (past the end: the last block's rows again, never used)
        const uint4 mq0 = mask_of(blk);
        kloop(ringA, acc);
        TGCN_ARRIVED(ringB);                               // issued a whole k-loop ago: no wait in the steady state
        epilogue(blk, acc, mq0);
        if (blk1 >= n_blocks) break;
        TGCN_ISSUE(ringA, arow_of(blk2));
        const uint4 mq1 = mask_of(blk1);
        kloop(ringB, acc);
        TGCN_ARRIVED(ringA);                               // every asm load is complete at the back edge
        epilogue(blk1, acc, mq1);
        blk = blk2;
    }
#undef TGCN_ISSUE
#undef TGCN_ARRIVED
#undef TGCN_R8
    if constexpr (COLSUM) {
        // a lane's rows in block order, then the two halves of the wave, then the four waves (k_gemm_tall's order)
        float *cl = lds + K * npad;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float v = csum[t] + __shfl_xor(csum[t], 32, 64);
            if (half == 0) cl[wave * npad + 32 * t + r] = v;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += blockDim.x)
            colpart[int64_t(blockIdx.x) * n + j] = (cl[j] + cl[npad + j]) + (cl[2 * npad + j] + cl[3 * npad + j]);
    }
}

// ---------------------------------------------------------------------------------------------
// Opt-in (tgcn_set_gemm_split): the same tall products with every fp32 product formed from an EXACT three-way
// bf16 split of both operands, a = a1 + a2 + a3 (a1 = bf16(a), a2 = bf16(a - a1), a3 = a - a1 - a2: 8 + 8 + 8
// significand bits, no bit lost), on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Of the nine partial
// products the six down to 2^-16 are taken (a1b1; a1b2, a2b1; a1b3, a2b2, a3b1); the dropped ones are <= 2^-23
// of |a||b|, the size of one fp32 rounding.  12 matrix-pipe cycles per unit of k instead of 32 (the fp32
// MFMA), which takes the product off the matrix-pipe roof and onto the HBM one.  NOT the fp32 FMA chain bit for
// bit; +-inf operands give nan (inf - inf in the split).  Built for the two shapes of the GCN layers only: the
// 16-wide k groups are fully unrolled (NQ of them; k = 16 NQ or 16 NQ - 8) around a counted load ring.
// Fragment maps (cdna_hip_programming.md section 3): lane l = (r = l & 31, h = l >> 5) holds A[row r][8h + j] and
// B[8h + j][col r], j = 0..7; the accumulator layout is that of the fp32 32x32 tile, so the epilogue is shared.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

struct Split3 {
    bf16x8 p1, p2, p3;
};

__device__ __forceinline__ Split3 split3(const f32x8 a) {
    Split3 s;
    s.p1 = __builtin_convertvector(a, bf16x8);                         // round to nearest even
    const f32x8 r1 = a - __builtin_convertvector(s.p1, f32x8);         // exact
    s.p2 = __builtin_convertvector(r1, bf16x8);
    const f32x8 r2 = r1 - __builtin_convertvector(s.p2, f32x8);        // exact, <= 8 significant bits
    s.p3 = __builtin_convertvector(r2, bf16x8);
    return s;
}

template <int V>
__device__ __forceinline__ void wait_vmcnt(f32x4 &lo, f32x4 &hi) {
    // the two ring slots about to be read are in/out operands, so that no consumer is scheduled above the wait
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(lo), "+v"(hi) : "n"(V) : "memory");
}

constexpr int kSplitWaves = 8;

template <class F, int... Q>
__device__ __forceinline__ void unroll_steps(F &f, std::integer_sequence<int, Q...>) {
    (f(std::integral_constant<int, Q>{}), ...);
}

// library-wide switch (tgcn_set_gemm_split; initial value from TGCN_GEMM_SPLIT)
std::atomic<int> g_gemm_split{[] {
    const char *e = std::getenv("TGCN_GEMM_SPLIT");
    return e ? std::atoi(e) : 0;
}()};

template <int NT, bool TRANS_B, int NQ, bool DROP, bool COLSUM>
__global__ __launch_bounds__(64 * kSplitWaves) void k_gemm_tall_split(const float *__restrict__ A, int64_t lda,
                                                                     const float *__restrict__ B, int64_t ldb,
                                                                     float *__restrict__ C, int64_t ldc, int64_t N, int k,
                                                                     int n, const Drop drop, float *__restrict__ colpart,
                                                                     const bool row_stores) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int npad = 32 * NT;
    constexpr int per_part = NQ * 2 * npad;                 // fragments (8 x bf16) per split part: [q][h][col]
    bf16x8 *img = reinterpret_cast<bf16x8 *>(lds_raw);
    // row_stores (contiguous C, n % 4 == 0): the result leaves through a per-wave LDS slice of 8 rows, as 16-byte
    // stores over whole rows -- the 32 rows of a block are one contiguous run of C -- instead of 4-byte stores of
    // 128-byte row pieces that straddle cache lines (rows of 800 bytes)
    float *stage = reinterpret_cast<float *>(lds_raw + size_t(3) * per_part * 16) + (threadIdx.x >> 6) * (8 * npad);
    float csum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) csum[t] = 0.f;
    uint32_t s_lo = 0, s_hi = 0;
    if constexpr (DROP) {
        const uint64_t sd = *drop.seed;
        s_lo = uint32_t(sd);
        s_hi = uint32_t(sd >> 32);
    }
    // stage the small operand: three bf16 images, zero padded, in fragment order
    for (int f = threadIdx.x; f < per_part; f += blockDim.x) {
        const int col = f % npad, hh = (f / npad) & 1, q = f / (2 * npad);
        f32x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 16 * q + 8 * hh + j;
            v[j] = (kk < k && col < n) ? (TRANS_B ? B[int64_t(col) * ldb + kk] : B[int64_t(kk) * ldb + col]) : 0.f;
        }
        const Split3 sp = split3(v);
        img[f] = sp.p1;
        img[per_part + f] = sp.p2;
        img[2 * per_part + f] = sp.p3;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, half = lane >> 5;
    const int64_t n_blocks = (N + 31) / 32;
    // the last k group may hold 8 columns only (k = 16 NQ - 8): its upper half re-reads the lower one (a valid
    // address of the same row; the matching rows of the B images are zero)
    const bool last_half_valid = 16 * (NQ - 1) + 8 * half < k;
    for (int64_t blk = int64_t(blockIdx.x) * kSplitWaves + wave; blk < n_blocks; blk += int64_t(gridDim.x) * kSplitWaves) {
        const int64_t row = blk * 32 + r;
        uint32_t a_key = 0;
        if constexpr (DROP && !TRANS_B) a_key = drop_row_key(s_lo, s_hi, drop_row_of(drop, row));
        const float *arow = A + std::min(row, N - 1) * lda;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        constexpr int RING = NQ < 8 ? NQ : 8;               // k groups in flight
        f32x4 ring[2 * RING];
        auto issue = [&](int slot, int q) {
            const float *p = arow + 16 * q + ((q == NQ - 1 && !last_half_valid) ? 0 : 8 * half);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ring[2 * slot]) : "v"(p) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(ring[2 * slot + 1]) : "v"(p) : "memory");
        };
#pragma unroll
        for (int q = 0; q < RING; ++q) issue(q, q);
        auto step = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int slot = q % RING;
            constexpr int later = (NQ - 1 - q) < (RING - 1) ? (NQ - 1 - q) : (RING - 1);   // k groups issued after this one
            wait_vmcnt<2 * later>(ring[2 * slot], ring[2 * slot + 1]);
            f32x8 a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = ring[2 * slot][j];
                a[4 + j] = ring[2 * slot + 1][j];
            }
            if constexpr (q + RING < NQ) issue(slot, q + RING);
            if constexpr (DROP && !TRANS_B) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = drop_elem(a[j], a_key, drop_col_term(16 * q + 8 * half + j), drop);
            }
            const Split3 sa = split3(a);
            const bf16x8 *bq = img + (q * 2 + half) * npad + r;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 b1 = bq[32 * t], b2 = bq[per_part + 32 * t], b3 = bq[2 * per_part + 32 * t];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.p3, b1, acc[t], 0, 0, 0);   // smallest terms first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.p2, b2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.p1, b3, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.p2, b1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.p1, b2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa.p1, b1, acc[t], 0, 0, 0);
            }
        };
        unroll_steps(step, std::make_integer_sequence<int, NQ>{});
        // epilogue: C[(i&3) + 8*(i>>2) + 4*half][32 t + r], as in k_gemm_tall
        uint32_t c_key[16];
        if constexpr (DROP && TRANS_B) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                c_key[i] = drop_row_key(s_lo, s_hi, drop_row_of(drop, blk * 32 + (i & 3) + 8 * (i >> 2) + 4 * half));
        }
        if (row_stores) {
            const int64_t rows_here = std::min<int64_t>(32, N - blk * 32);
#pragma unroll
            for (int g8 = 0; g8 < 4; ++g8) {                 // rows 8 g8 .. 8 g8 + 7 of the block: registers 4 g8 .. 4 g8 + 3
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int col = 32 * t + r;
                    const uint32_t cterm = drop_col_term(col);
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        const int i = 4 * g8 + i4;
                        float out = acc[t][i];
                        if constexpr (DROP && TRANS_B) out = drop_elem(out, c_key[i], cterm, drop);
                        if (col < n) {
                            stage[(i4 + 4 * half) * n + col] = out;
                            if constexpr (COLSUM)
                                if (8 * g8 + i4 + 4 * half < rows_here) csum[t] += out;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();             // the slice belongs to this wave: its LDS operations run in order
                const int64_t live = std::min<int64_t>(8, rows_here - 8 * g8) * n / 4;      // float4s of existing rows
                float4 *dst = reinterpret_cast<float4 *>(C + (blk * 32 + 8 * g8) * ldc);
                for (int f = lane; f < live; f += 64) dst[f] = reinterpret_cast<const float4 *>(stage)[f];
                __builtin_amdgcn_wave_barrier();
            }
        } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = 32 * t + r;
            if (col < n) {
                const uint32_t cterm = drop_col_term(col);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int64_t orow = blk * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                    float out = acc[t][i];
                    if constexpr (DROP && TRANS_B) out = drop_elem(out, c_key[i], cterm, drop);
                    if (orow < N) {
                        C[orow * ldc + col] = out;      // (non-temporal stores here: 0.62 -> 0.90 ms, the L2 no longer merges the row pieces)
                        if constexpr (COLSUM) csum[t] += out;
                    }
                }
            }
            if constexpr (TRANS_B && (DROP || COLSUM)) __builtin_amdgcn_sched_barrier(0);
        }
        }
    }
    if constexpr (COLSUM) {
        __syncthreads();                                    // every wave is done with the B images
        float *cl = reinterpret_cast<float *>(lds_raw);     // [kSplitWaves][npad]
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float v = csum[t] + __shfl_xor(csum[t], 32, 64);
            if (half == 0) cl[wave * npad + 32 * t + r] = v;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < kSplitWaves; ++w) sum += cl[w * npad + j];
            colpart[int64_t(blockIdx.x) * n + j] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// partial[b][kpad][npad] = sum over this workgroup's rows of A[r, :]^T G[r, :]
// Both operands are read straight from global memory in fragment order: for one MFMA step (2 rows)
// lanes 0-31 / 32-63 read 32 consecutive floats of row 2s / 2s+1 (two coalesced 128-byte segments).
// 4 waves per workgroup; wave w owns the M-tiles {w, w+4} (of kpad/32 <= 8) x all NT column tiles.
// ---------------------------------------------------------------------------------------------
template <int NT, bool DROP>
__global__ __launch_bounds__(256, 2) void k_gemm_tn_partial(const float *__restrict__ A, int64_t lda,
                                                            const float *__restrict__ G, int64_t ldg,
                                                            int64_t N, int k, int n, int64_t rows_per_wg,
                                                            float *__restrict__ partial, const Drop drop, const int k0) {
    // k0: global column of A's first column here (a product with more than 256 columns of A runs in chunks; the
    // dropout hash is a function of the global position)
    uint32_t s_lo = 0, s_hi = 0;
    if constexpr (DROP) {
        const uint64_t sd = *drop.seed;
        s_lo = uint32_t(sd);
        s_hi = uint32_t(sd >> 32);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, half = lane >> 5;
    const int mt = (k + 31) / 32;  // M tiles
    constexpr int npad = 32 * NT;
    const int mpad = 32 * mt;
    const int64_t r_begin = int64_t(blockIdx.x) * rows_per_wg;
    const int64_t r_end = std::min(N, r_begin + rows_per_wg);
    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
    // Columns past k (of A) or n (of G) are clamped to a valid address and NOT zeroed: they only
    // feed rows >= k / columns >= n of the padded result tile, which nobody reads.  That keeps the
    // main loop free of per-lane conditions (a conditional load makes hipcc wait vmcnt(0) per load).
    const int m0 = wave, m1 = wave + 4;
    const float *pa0 = A + r_begin * lda + half * lda + std::min(32 * m0 + c, k - 1);
    const float *pa1 = A + r_begin * lda + half * lda + std::min(32 * m1 + c, k - 1);
    const float *pg[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) pg[t] = G + r_begin * ldg + half * ldg + std::min(32 * t + c, n - 1);
    constexpr int UR = 4;  // row pairs per stage; two stages in flight (static double buffer)
    const int64_t n_rows = r_end > r_begin ? r_end - r_begin : 0;
    const int64_t n_full = n_rows / (2 * UR);
    float a0[2][UR], a1[2][UR], g[2][UR][NT];
    const int ca0 = std::min(32 * m0 + c, k - 1), ca1 = std::min(32 * m1 + c, k - 1);   // columns of A this lane reads
    int64_t next_row = r_begin + half, stage_row[2] = {0, 0};
    auto load_stage = [&](int buf) {
        stage_row[buf] = next_row;
        next_row += 2 * UR;
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            a0[buf][u] = pa0[int64_t(2 * u) * lda];
            a1[buf][u] = pa1[int64_t(2 * u) * lda];
#pragma unroll
            for (int t = 0; t < NT; ++t) g[buf][u][t] = pg[t][int64_t(2 * u) * ldg];
        }
        pa0 += 2 * UR * lda;
        pa1 += 2 * UR * lda;
#pragma unroll
        for (int t = 0; t < NT; ++t) pg[t] += 2 * UR * ldg;
    };
    auto mfma_stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            float x0 = a0[buf][u], x1 = a1[buf][u];
            if constexpr (DROP) {
                const uint32_t key = drop_row_key(s_lo, s_hi, drop_row_of(drop, stage_row[buf] + 2 * u));
                x0 = drop_elem(x0, key, drop_col_term(k0 + ca0), drop);
                x1 = drop_elem(x1, key, drop_col_term(k0 + ca1), drop);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, g[buf][u][t], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, g[buf][u][t], acc[1][t], 0, 0, 0);
            }
        }
    };
    if (n_full > 0) load_stage(0);
    for (int64_t it = 0; it < n_full; it += 2) {
        if (it + 1 < n_full) load_stage(1);
        mfma_stage(0);
        if (it + 1 < n_full) {
            if (it + 2 < n_full) load_stage(0);
            mfma_stage(1);
        }
    }
    // ragged tail of the row range: rows past the end contribute zeros
    for (int64_t r = n_full * 2 * UR; r < n_rows; r += 2) {
        const bool ok = r + half < n_rows;
        const int64_t back = ok ? 0 : 1;      // the odd last row: re-read the previous one, scaled by 0
        const float s = ok ? 1.f : 0.f;
        float x0 = pa0[-back * lda] * s, x1 = pa1[-back * lda] * s;
        if constexpr (DROP) {
            const uint32_t key = drop_row_key(s_lo, s_hi, drop_row_of(drop, r_begin + r + half - back));
            x0 = drop_elem(x0, key, drop_col_term(k0 + ca0), drop);
            x1 = drop_elem(x1, key, drop_col_term(k0 + ca1), drop);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float y = pg[t][-back * ldg] * s;
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y, acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y, acc[1][t], 0, 0, 0);
        }
        pa0 += 2 * lda;
        pa1 += 2 * lda;
#pragma unroll
        for (int t = 0; t < NT; ++t) pg[t] += 2 * ldg;
    }
    float *out = partial + int64_t(blockIdx.x) * mpad * npad;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int mtile = m == 0 ? m0 : m1;
        if (mtile >= mt) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int orow = 32 * mtile + (i & 3) + 8 * (i >> 2) + 4 * half;
                out[orow * npad + 32 * t + c] = acc[m][t][i];
            }
    }
}

// ---------------------------------------------------------------------------------------------
// The same product through LDS stages with the ROWS split over the waves (round 4; the default fp32 path whenever
// the operands are 16-byte friendly).  k_gemm_tn_partial above gives wave w the M tiles {w, w + 4}: at k = 200 (7 M
// tiles) the four waves issue 16 tile products per row pair for the 14 that exist, every fragment is a 4-byte global
// load of a 128-byte row piece, and the matrix pipe is 53 % busy (profiles/r02_pmc_gemm_c4.md).  Here a stage of 32 rows
// of A and G is copied by the whole workgroup with 16-byte loads into a double-buffered LDS image (the loads of stage
// s + 1 are in flight under the MFMAs of stage s), and wave w = (t, par) owns column tile t of G and the row pairs
// p = par (mod PAR) of every stage for ALL M tiles: mt MFMAs per row pair and wave, none wasted, every wave the same
// load.  The PAR row classes are added in a fixed order through LDS at the end (deterministic).
// NT = number of 32-column tiles of G (1, 2 or 4; PAR = 4 / NT).  k, n, lda, ldg multiples of 4, 16-byte aligned bases.
// Fragments past k / n read whatever follows in the LDS image: they only feed rows >= k / columns >= n of the padded
// result tile, which nobody reads.
// ---------------------------------------------------------------------------------------------
constexpr int kTnStageRows = 32;

inline size_t tn_staged_lds_bytes(int k, int n) {
    const size_t stage = size_t(2) * kTnStageRows * ((k + 4) + (n + 4)) + 320;       // + over-read pad (<= 8 tiles of a row)
    const size_t reduce = size_t(2) * ((k + 31) / 32) * 1024;                        // two waves' accumulator tiles
    return sizeof(float) * std::max(stage, reduce);
}

// BITS (with DROP): the keep decisions come from the record the nn product left (Drop::bits) -- one 4-byte load and a few
// bit tests per float4 instead of four hashes: the hash of an element is vector-ALU work that ADDS to the MFMA time on this
// part (profiles/r04_pmc_dense.md), 0.18 ms of the 0.76 ms of the masked product at the c4 shapes.
// Occupancy target: two workgroups per CU up to 7 accumulator tiles (k <= 224: 71 KB of LDS each).  With 8 tiles (k = 256,
// c5's hidden width) the LDS image is 85 KB -- ONE workgroup per CU whatever the registers allow -- and the 128 accumulator
// registers plus the staging registers do not fit the 256-register budget of the two-workgroup target (the masked
// instantiation spilled 11 VGPRs, 48 B / lane of scratch): it is compiled for one workgroup per CU and takes what it needs.
#ifndef TGCN_TN_STAGED8_WGS
#define TGCN_TN_STAGED8_WGS 1         // 2: A/B builds with round 5's occupancy target (spills; tools/build_variant.py)
#endif
template <int NT, int MT, bool DROP, bool BITS = false>
__global__ __launch_bounds__(256, (MT >= 8 ? TGCN_TN_STAGED8_WGS : 2)) void k_gemm_tn_staged(const float *__restrict__ A, int64_t lda,
                                                           const float *__restrict__ G, int64_t ldg, int64_t N, int k,
                                                           int n, int64_t rows_per_wg, float *__restrict__ partial,
                                                           const Drop drop, const int k0) {
    extern __shared__ __align__(16) float tn_lds[];
    constexpr int PAR = 4 / NT, SR = kTnStageRows;       // MT: accumulator tiles held (>= the M tiles of k: 2, 4, 7 or 8)
    uint32_t s_lo = 0, s_hi = 0;
    if constexpr (DROP) {
        const uint64_t sd = *drop.seed;
        s_lo = uint32_t(sd);
        s_hi = uint32_t(sd >> 32);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, half = lane >> 5;
    const int t = wave % NT, par = wave / NT;
    const int mt = (k + 31) / 32;
    constexpr int npad = 32 * NT;
    const int mpad = 32 * mt;
    const int kp = k + 4, np_ = n + 4;                       // LDS row pitches (multiples of 4: 16-byte stores)
    const int stage_floats = SR * (kp + np_);
    const int64_t r_begin = int64_t(blockIdx.x) * rows_per_wg;
    const int64_t r_end = std::min(N, r_begin + rows_per_wg);
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
    // global -> registers -> LDS.  A: wave w copies rows w, w + 4, ... of the stage, lane j the j-th float4 of the row
    // (k / 4 <= 64 of them); G: 16 lanes per row... generally n / 4 <= 32 float4s per row, two rows per wave pass.
    // G rows per load instruction of a wave: n <= 64 (NT <= 2) needs 16 lanes per row, so four rows ride in one
    // instruction instead of two (half the lanes idle) -- the count of memory instructions is what a stage pays for
    constexpr int GL = NT <= 2 ? 16 : 32, GR = 64 / GL;
    constexpr int AV = SR / 4, GV = SR / (4 * GR);
    const int k4 = k >> 2, n4 = n >> 2;
    const bool a_on = lane < k4;
    const int g_sub = lane / GL, g_j = lane % GL;            // G: GL lanes per row, GR rows per instruction
    const bool g_on = g_j < n4;
    float4 ra[AV], rg[GV];
    uint32_t rbw = 0;                                        // BITS: one word of the stage's recorded mask per lane
    // the lane's first row of the slice, formed once: a stage's rows sit at the wave-uniform offsets (row0 - r_begin + 4 u)
    // * lda from it (scalar arithmetic instead of a 64-bit vector multiply per load)
    const float *pa0 = A + (r_begin + wave) * lda + 4 * lane;
    const float *pg0 = G + (r_begin + GR * wave + g_sub) * ldg + 4 * g_j;
    // Recorded mask.  The AV = 8 rows a wave copies per stage hold 8 words each (k <= 256): ONE 4-byte load per lane and
    // stage fetches them all -- lane i word i & 7 of the wave's row i >> 3 -- and the lane that masks the float4 at
    // columns 4 j .. of row u (step q = j / 2, half j & 1 of that row: word (j & 1) wph + j / 16) takes its word from
    // lane 8 u + that index (ds_bpermute).  A load per row and lane instead cost 0.07 ms at the c4 shapes: the
    // instructions, not the bytes.
    const uint32_t *pb0 = nullptr;
    const int b_shift = 4 * ((lane >> 1) & 7);
    int b_src = 0;
    if constexpr (BITS) {
        static_assert(!BITS || AV == 8, "one mask word per lane and stage assumes 8 rows per wave");
        pb0 = drop.bits + (r_begin + wave + 4 * (lane >> 3)) * drop.bits_stride + (lane & 7);
        b_src = (lane & 1) * drop_bits_wph(k) + (lane >> 4);
    }
    auto fetch = [&](int64_t row0) {                          // zeros past the slice
        const int64_t d = row0 - r_begin;                     // wave-uniform
#pragma unroll
        for (int u = 0; u < AV; ++u) {
            const int64_t row = row0 + wave + 4 * u;
            ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_on && row < r_end) ra[u] = *reinterpret_cast<const float4 *>(pa0 + (d + 4 * u) * lda);
        }
        if constexpr (BITS) {
            rbw = 0u;
            if ((lane & 7) < 2 * drop_bits_wph(k) && row0 + wave + 4 * (lane >> 3) < r_end) rbw = pb0[d * drop.bits_stride];
        }
#pragma unroll
        for (int u = 0; u < GV; ++u) {
            const int64_t row = row0 + GR * wave + g_sub + 4 * GR * u;
            rg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g_on && row < r_end) rg[u] = *reinterpret_cast<const float4 *>(pg0 + (d + 4 * GR * u) * ldg);
        }
    };
    // the fused dropout masks A on its way into LDS: every element is hashed ONCE per workgroup, by the thread that
    // copies it (masking the fragments as they are read would hash every element once per column-tile wave)
    auto deposit = [&](int buf, int64_t row0) {
        float *la = tn_lds + buf * stage_floats, *lg = la + SR * kp;
#pragma unroll
        for (int u = 0; u < AV; ++u) {
            int w = 0;                                        // every lane takes part: a lane switched off reads as 0
            if constexpr (DROP && BITS) w = __shfl(static_cast<int>(rbw), 8 * u + b_src, 64);
            if (a_on) {
                float4 v = ra[u];
                // (the factor 1 / (1 - p) of the kept elements is applied ONCE, to the k x n result, by k_gemm_tn_reduce:
                // here an element is kept or zeroed, nothing else)
                if constexpr (DROP && BITS) {
                    // a 1-bit signed field extract gives 0 or ~0: two vector instructions per element
                    v.x = __uint_as_float(__float_as_uint(v.x) & uint32_t(__builtin_amdgcn_sbfe(w, b_shift, 1)));
                    v.y = __uint_as_float(__float_as_uint(v.y) & uint32_t(__builtin_amdgcn_sbfe(w, b_shift + 1, 1)));
                    v.z = __uint_as_float(__float_as_uint(v.z) & uint32_t(__builtin_amdgcn_sbfe(w, b_shift + 2, 1)));
                    v.w = __uint_as_float(__float_as_uint(v.w) & uint32_t(__builtin_amdgcn_sbfe(w, b_shift + 3, 1)));
                } else if constexpr (DROP) {
                    const uint32_t key = drop_row_key_uniform(s_lo, s_hi, drop_row_of(drop, row0 + wave + 4 * u));
                    v.x = drop_keep(key, drop_col_term(k0 + 4 * lane), drop) ? v.x : 0.f;
                    v.y = drop_keep(key, drop_col_term(k0 + 4 * lane + 1), drop) ? v.y : 0.f;
                    v.z = drop_keep(key, drop_col_term(k0 + 4 * lane + 2), drop) ? v.z : 0.f;
                    v.w = drop_keep(key, drop_col_term(k0 + 4 * lane + 3), drop) ? v.w : 0.f;
                }
                *reinterpret_cast<float4 *>(la + (wave + 4 * u) * kp + 4 * lane) = v;
            }
        }
#pragma unroll
        for (int u = 0; u < GV; ++u)
            if (g_on) *reinterpret_cast<float4 *>(lg + (GR * wave + g_sub + 4 * GR * u) * np_ + 4 * g_j) = rg[u];
    };
    auto compute = [&](int buf) {
        const float *la = tn_lds + buf * stage_floats + half * kp + c;
        const float *lg = tn_lds + buf * stage_floats + SR * kp + half * np_ + 32 * t + c;
        // All MT tiles, unconditionally (a run-time `m < mt` test puts a branch in front of every MFMA; tiles past mt
        // multiply whatever follows in the LDS image into accumulators nobody reads: MT is the smallest instantiated count
        // >= mt), and the fragments of row pair q + 1 are read while pair q is multiplied: left to itself hipcc issues each
        // fragment read right in front of the MFMA that needs it, and the wave then waits out the LDS latency once per tile
        // (40 % of the wave cycles waiting, profiles/r04_pmc_dense.md).
        constexpr int n_pairs = SR / 2 / PAR;
        float af[2][MT], gf[2];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[0][m] = la[2 * par * kp + 32 * m];
        gf[0] = lg[2 * par * np_];
#pragma unroll
        for (int q = 0; q < n_pairs; ++q) {
            if (q + 1 < n_pairs) {
                const int p = par + PAR * (q + 1);            // row pair of the stage: rows 2 p, 2 p + 1
#pragma unroll
                for (int m = 0; m < MT; ++m) af[(q + 1) & 1][m] = la[2 * p * kp + 32 * m];
                gf[(q + 1) & 1] = lg[2 * p * np_];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][m], gf[q & 1], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (r_end > r_begin) {
        const int64_t n_st = (r_end - r_begin + SR - 1) / SR;
        fetch(r_begin);
        deposit(0, r_begin);
        __syncthreads();
        for (int64_t it = 0; it < n_st; ++it) {
            if (it + 1 < n_st) fetch(r_begin + SR * (it + 1));      // in flight under the MFMAs of this stage
            compute(int(it & 1));
            if (it + 1 < n_st) deposit(int((it + 1) & 1), r_begin + SR * (it + 1));
            __syncthreads();
        }
    }
    // add the row classes: par in [s, 2 s) hands its tiles to par - s through LDS, s = PAR / 2, PAR / 4, ... (fixed order)
#pragma unroll
    for (int s = PAR / 2; s >= 1; s >>= 1) {
        float *red = tn_lds + ((par - s) * NT + t) * (mt * 1024) + lane;
        if (par >= s && par < 2 * s) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (m < mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) red[(m * 16 + i) * 64] = acc[m][i];
        }
        __syncthreads();
        if (par < s) {
            const float *src = tn_lds + (par * NT + t) * (mt * 1024) + lane;
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (m < mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[m][i] += src[(m * 16 + i) * 64];
        }
        __syncthreads();
    }
    if (par == 0) {
        float *out = partial + int64_t(blockIdx.x) * mpad * npad;
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (m < mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int orow = 32 * m + (i & 3) + 8 * (i >> 2) + 4 * half;
                    out[orow * npad + 32 * t + c] = acc[m][i];
                }
    }
}

// The tn product with split-bf16 operands (see k_gemm_tall_split): the reduction index is the ROW, so a K step of
// v_mfma_f32_32x32x16_bf16 is 16 rows.  A stage = 32 rows (two K steps) of A (contiguous: 32 k floats) and of G,
// copied by the whole workgroup with 16-byte loads into a double-buffered LDS image (row pitch k + 4 / n + 4: the
// two lane halves, 8 rows apart, then sit 32 banks apart); lane (c, h) reads rows 8h .. 8h+7 of its column of A
// (M operand, two 32-wide tiles per wave) and of G (N operand) from there, masks (fused dropout), splits and issues
// 6 MFMAs per tile pair.  (The first version read the fragments straight from global memory with 4-byte loads of
// 128-byte row pieces: 0.67 ms at c4 where the 2.1 GB stream allows 0.43.)  Rows past the end of the workgroup's
// slice are zero-filled in the copy.  Same partial-tile layout as k_gemm_tn_partial; contiguous operands
// (lda == k, ldg == n, both multiples of 4), rows_per_wg a multiple of 16.
template <int NT, bool DROP>
__global__ __launch_bounds__(256, 2) void k_gemm_tn_split(const float *__restrict__ A, const float *__restrict__ G,
                                                          int64_t N, int k, int n, int64_t rows_per_wg,
                                                          float *__restrict__ partial, const Drop drop) {
    extern __shared__ __align__(16) float tn_lds[];
    uint32_t s_lo = 0, s_hi = 0;
    if constexpr (DROP) {
        const uint64_t sd = *drop.seed;
        s_lo = uint32_t(sd);
        s_hi = uint32_t(sd >> 32);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, half = lane >> 5;
    const int mt = (k + 31) / 32;
    constexpr int npad = 32 * NT;
    const int mpad = 32 * mt;
    const int kp = k + 4, np_ = n + 4;                       // LDS row pitches
    constexpr int SR = 32;                                   // rows per stage: two K steps of 16 between workgroup barriers
    const int stage_floats = SR * kp + SR * np_;
    const int64_t r_begin = int64_t(blockIdx.x) * rows_per_wg;
    const int64_t r_end = std::min(N, r_begin + rows_per_wg);
    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
    const int m0 = wave, m1 = wave + 4;
    const bool two = m1 < mt;                                // the last wave may own one tile only
    // columns past k / n are clamped (they only feed rows / columns of the padded tile nobody reads)
    const int ca0 = std::min(32 * m0 + c, k - 1), ca1 = std::min(32 * m1 + c, k - 1);
    int cg[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) cg[t] = std::min(32 * t + c, n - 1);
    const int a_vec = SR * k / 4, g_vec = SR * n / 4;        // float4s of one stage
    constexpr int AV = 8, GV = 4;                            // float4s per thread and stage (k <= 256, n <= 128)
    float4 ra[AV], rg[GV];
    auto fetch = [&](int64_t row0) {                          // global -> registers (zeros past the slice)
#pragma unroll
        for (int u = 0; u < AV; ++u) {
            const int f = threadIdx.x + 256 * u;
            ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < a_vec && row0 + f / (k / 4) < r_end) ra[u] = reinterpret_cast<const float4 *>(A + row0 * k)[f];
        }
#pragma unroll
        for (int u = 0; u < GV; ++u) {
            const int f = threadIdx.x + 256 * u;
            rg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < g_vec && row0 + f / (n / 4) < r_end) rg[u] = reinterpret_cast<const float4 *>(G + row0 * n)[f];
        }
    };
    auto deposit = [&](int buf) {                             // registers -> LDS image `buf`
        float *la = tn_lds + buf * stage_floats, *lg = la + SR * kp;
#pragma unroll
        for (int u = 0; u < AV; ++u) {
            const int f = threadIdx.x + 256 * u;
            if (f < a_vec) *reinterpret_cast<float4 *>(la + (f / (k / 4)) * kp + 4 * (f % (k / 4))) = ra[u];
        }
#pragma unroll
        for (int u = 0; u < GV; ++u) {
            const int f = threadIdx.x + 256 * u;
            if (f < g_vec) *reinterpret_cast<float4 *>(lg + (f / (n / 4)) * np_ + 4 * (f % (n / 4))) = rg[u];
        }
    };
    auto compute = [&](int buf, int64_t row0, int ks) {      // K step ks (0 / 1) of the stage: its rows 16 ks .. 16 ks + 15
        const float *la = tn_lds + buf * stage_floats + (16 * ks + 8 * half) * kp;
        const float *lg = tn_lds + buf * stage_floats + SR * kp + (16 * ks + 8 * half) * np_;
        row0 += 16 * ks;
        f32x8 x0, x1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v0 = la[j * kp + ca0], v1 = la[j * kp + ca1];
            if constexpr (DROP) {
                const uint32_t key = drop_row_key(s_lo, s_hi, drop_row_of(drop, row0 + 8 * half + j));
                v0 = drop_elem(v0, key, drop_col_term(ca0), drop);
                v1 = drop_elem(v1, key, drop_col_term(ca1), drop);
            }
            x0[j] = v0;
            x1[j] = v1;
        }
        const Split3 s0 = split3(x0), s1 = split3(x1);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x8 y;
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = lg[j * np_ + cg[t]];
            const Split3 sg = split3(y);
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s0.p3, sg.p1, acc[0][t], 0, 0, 0);
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s0.p2, sg.p2, acc[0][t], 0, 0, 0);
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s0.p1, sg.p3, acc[0][t], 0, 0, 0);
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s0.p2, sg.p1, acc[0][t], 0, 0, 0);
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s0.p1, sg.p2, acc[0][t], 0, 0, 0);
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s0.p1, sg.p1, acc[0][t], 0, 0, 0);
            if (two) {
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s1.p3, sg.p1, acc[1][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s1.p2, sg.p2, acc[1][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s1.p1, sg.p3, acc[1][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s1.p2, sg.p1, acc[1][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s1.p1, sg.p2, acc[1][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s1.p1, sg.p1, acc[1][t], 0, 0, 0);
            }
        }
    };
    if (r_end > r_begin) {
        const int64_t n_st = (r_end - r_begin + SR - 1) / SR;
        fetch(r_begin);
        deposit(0);
        __syncthreads();
        for (int64_t it = 0; it < n_st; ++it) {
            if (it + 1 < n_st) fetch(r_begin + SR * (it + 1));      // in flight under the MFMAs of this stage
            compute(int(it & 1), r_begin + SR * it, 0);
            if (r_begin + SR * it + 16 < r_end) compute(int(it & 1), r_begin + SR * it, 1);
            if (it + 1 < n_st) deposit(int((it + 1) & 1));
            __syncthreads();
        }
    }
    float *out = partial + int64_t(blockIdx.x) * mpad * npad;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int mtile = m == 0 ? m0 : m1;
        if (mtile >= mt) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int orow = 32 * mtile + (i & 3) + 8 * (i >> 2) + 4 * half;
                out[orow * npad + 32 * t + c] = acc[m][t][i];
            }
    }
}

// C[kk][j] = sum_b partial[b][kk][j], in block order (deterministic); one thread per element,
// 4-way split over b combined through LDS to shorten the serial chain.
// `scale`: the dropout factor 1 / (1 - p) of the LDS-staged masked kernel, which only keeps or zeroes (1 otherwise).
__global__ __launch_bounds__(256) void k_gemm_tn_reduce(const float *__restrict__ partial, int nb,
                                                        int mpad, int npad, int k, int n,
                                                        float *__restrict__ C, int64_t ldc, float scale = 1.0f) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;  // element of the [k][n] result
    const int kk = e / n, j = e % n;
    float s = 0.f;
    if (e < k * n) {
        const float *p = partial + int64_t(kk) * npad + j;
        const int64_t stride = int64_t(mpad) * npad;
        int b = wave;
        for (; b + 28 < nb; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(b + 4 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nb; b += 4) s += p[b * stride];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && e < k * n) {
        const float sum = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        C[int64_t(kk) * ldc + j] = scale == 1.0f ? sum : sum * scale;
    }
}

int tn_blocks(int64_t N) {
    // 2 workgroups per CU (launch bounds), one resident round: 512 on MI355X
    int64_t nb = (N + 1023) / 1024;
    return static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(nb, 512)));
}

constexpr int kTallMaxGrid = 256 * 4 * 2;   // upper bound of the persistent grid (CUs x resident workgroups), sizes colpart

template <bool TRANS_B, bool DROP, bool COLSUM = false>
int launch_tall_one(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                    int64_t N, int k, int n, const Drop drop, hipStream_t s, float *colpart, float *colsum,
                    const Place place) {
    const int nt = (n + 31) / 32;
    const int kpad = (k + 7) & ~7;
    size_t lds_bytes = sizeof(float) * (static_cast<size_t>(kpad) * (32 * nt) + (COLSUM ? 4 * 32 * nt : 0));
    int grid_used = 0;
    if (lds_bytes > 160 * 1024 || nt > 8) {
        set_error("tgcn_gemm: the small operand (%d x %d) does not fit the 160 KB LDS", k, n);
        return TGCN_E_INVALID;
    }
    const int64_t n_blocks = (N + 31) / 32;
    // persistent workgroups: exactly one resident set (a second, partial round of workgroups would
    // leave most CUs idle at the end)
    int n_cu = 256;
    {
        // hipGetDeviceProperties fills a large struct (tens of microseconds): asked once per device
        static std::atomic<int> cached[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            int v = cached[dev].load(std::memory_order_relaxed);
            if (v == 0) {
                hipDeviceProp_t prop;
                if (hipGetDeviceProperties(&prop, dev) == hipSuccess) v = prop.multiProcessorCount;
                if (v > 0) cached[dev].store(v, std::memory_order_relaxed);
            }
            if (v > 0) n_cu = v;
        }
    }
#define TGCN_TALL_K(NT, K8, NQ_) TGCN_TALL_KB(NT, K8, NQ_, false)
#define TGCN_TALL_KB(NT, K8, NQ_, BI)                                                                \
    do {                                                                                          \
        const void *fn = reinterpret_cast<const void *>(&k_gemm_tall<NT, TRANS_B, K8, NQ_, DROP, COLSUM, BI>); \
        TGCN_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                           static_cast<int>(lds_bytes)));                         \
        int per_cu = 1;                                                                           \
        TGCN_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds_bytes)); \
        per_cu = std::max(1, std::min(per_cu, 4));                                                \
        const int grid = static_cast<int>(std::max<int64_t>(                                      \
            1, std::min<int64_t>({(n_blocks + 3) / 4, int64_t(n_cu) * per_cu, int64_t(kTallMaxGrid)}))); \
        grid_used = grid;                                                                         \
        k_gemm_tall<NT, TRANS_B, K8, NQ_, DROP, COLSUM, BI><<<grid, 256, lds_bytes, s>>>(A, lda, B, ldb, C, ldc, N, k, n, \
                                                                                         drop, colpart, place); \
    } while (0)
#define TGCN_TALL(NT)                                                                             \
    do {                                                                                          \
        if (k % 8 == 0)                                                                           \
            TGCN_TALL_K(NT, true, 0);                                                             \
        else                                                                                      \
            TGCN_TALL_K(NT, false, 0);                                                            \
    } while (0)
    // the two shapes of the GCN layers get fully unrolled k-loops (hidden width 200, class width 64)
    auto finish = [&]() -> int {
        TGCN_HIP_CHECK(hipGetLastError());
        if constexpr (COLSUM) return launch_colsum_final(colpart, grid_used, n, colsum, s);
        return TGCN_OK;
    };
    const bool whole = place.col0 == 0 && place.k0 == 0 && !place.accum;      // not a piece of a larger product
    if (whole && g_gemm_split.load(std::memory_order_relaxed) != 0 && lda % 4 == 0) {
        // split-bf16 products for the two shapes of the GCN layers (everything else keeps the fp32 MFMA kernels)
#define TGCN_SPLIT(NT, NQ_)                                                                                         \
    do {                                                                                                            \
        const bool rs = TRANS_B && ldc == n && n % 4 == 0 && reinterpret_cast<uintptr_t>(C) % 16 == 0;              \
        const size_t lb = size_t(3) * NQ_ * 2 * (32 * NT) * 16 + (rs ? size_t(kSplitWaves) * 8 * (32 * NT) * 4 : 0); \
        const void *fn = reinterpret_cast<const void *>(&k_gemm_tall_split<NT, TRANS_B, NQ_, DROP, COLSUM>);        \
        TGCN_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lb))); \
        const int grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>({(n_blocks + kSplitWaves - 1) / kSplitWaves, int64_t(n_cu), int64_t(kTallMaxGrid)}))); \
        grid_used = grid;                                                                                           \
        k_gemm_tall_split<NT, TRANS_B, NQ_, DROP, COLSUM><<<grid, 64 * kSplitWaves, lb, s>>>(A, lda, B, ldb, C, ldc, N, k, n, \
                                                                                            drop, colpart, rs);     \
    } while (0)
        if (!TRANS_B && !COLSUM && k == 200 && nt == 2) {
            TGCN_SPLIT(2, 13);
            return finish();
        }
        if (TRANS_B && k == 64 && nt == 7) {
            TGCN_SPLIT(7, 4);
            return finish();
        }
#undef TGCN_SPLIT
    }
    // ONE resident workgroup per CU for the two GCN shapes when nothing is hashed: measured at c4 (tools/ab_dense.py), with
    // a second workgroup kept off the CU by reserving more than half of its LDS -- nt 0.67 -> 0.59 ms, nt + column sums
    // 0.72 -> 0.58, nn 0.55 -> 0.52 (2 per CU: 0.525; its default is 3); the kernels that hash a mask want their second /
    // third workgroup (nt + mask 0.74 -> 0.75, nn + mask 0.60 -> 0.69) and keep it -- and so does the nt product that reads
    // its mask from the record: its epilogue still works on every element (an LDS read, a multiply, two bit operations:
    // 0.12 ms when nothing overlaps it), which a second workgroup hides.
    // TGCN_TALL_ONE_PER_CU=0 (diagnostic, read once): let as many workgroups onto a CU as fit, as before round 4
    static const bool one_per_cu = [] {
        const char *e = std::getenv("TGCN_TALL_ONE_PER_CU");
        return !(e && e[0] == '0');
    }();
    const size_t kOnePerCu = one_per_cu ? 82 * 1024 : 0;
    if constexpr (!TRANS_B) {
        if (whole && k == 200 && nt == 2) {
            if constexpr (!DROP) lds_bytes = std::max(lds_bytes, kOnePerCu);
            TGCN_TALL_K(2, true, 25);
            return finish();
        }
    }
    if constexpr (!TRANS_B) {
        // a column group of a wider layer at the hidden width 200 (DBpedia's 219 classes = 128 + 91 columns): the fully
        // unrolled k-loop with the counted load ring, as the GCN shape has it (the generic loop leaves 8 MFMAs of cover)
        if (k == 200 && place.k0 == 0 && !place.accum && (nt == 4 || nt == 3)) {
            if (nt == 4) TGCN_TALL_K(4, true, 25); else TGCN_TALL_K(3, true, 25);
            return finish();
        }
    }
    if constexpr (TRANS_B) {
        // a column group of the input-gradient product of a 217 .. 224-class layer (DBpedia l3: 219; hidden 200 = 128 + 72
        // result columns): 28 unrolled steps with the load ring; needs the operand's rows to reach round_up(k, 4) (the
        // zero-padded gradient buffer) so that the last piece of a row is read inside it
        if (k > 216 && k <= 224 && place.k0 == 0 && !place.accum && (nt == 4 || nt == 3) && lda >= ((k + 3) & ~3)) {
            if (k % 8 == 0) {
                if (nt == 4) TGCN_TALL_K(4, true, 28); else TGCN_TALL_K(3, true, 28);
            } else {
                if (nt == 4) TGCN_TALL_K(4, false, 28); else TGCN_TALL_K(3, false, 28);
            }
            return finish();
        }
#if TGCN_NT_PIPE
    if (whole && k == 64 && nt == 7 && lda % 4 == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0) {
        // the block-pipelined kernel (k_gemm_pipe): plain / column sums / mask from the record
        const bool from_record = DROP && drop.bits && n > 192 && drop.bits_stride % 4 == 0 &&
                                 reinterpret_cast<uintptr_t>(drop.bits) % 16 == 0;
        if (!DROP || from_record) {
            size_t lb = sizeof(float) * (size_t(64) * 224 + (COLSUM ? 4 * 224 : 0)) + (from_record ? 4 * 1024 : 0);
            // Workgroups per CU, measured at c4 (tools/ab_dense.py, profiles/r05_ab_dense.log): the unmasked kernels want the CU
            // to themselves (nt 0.515 against 0.565 ms with two, nt + column sums 0.53 / 0.59); the one that masks its result
            // from the record has ~450 vector-ALU instructions of epilogue per block, which a second workgroup's MFMAs hide
            // (0.59 against 0.625 ms).  k_gemm_tall, the kernel this replaces: 0.585 / 0.58 / 0.665 ms.
            const int pipe_per_cu = from_record ? 2 : 1;
            if (pipe_per_cu == 1) lb = std::max(lb, kOnePerCu);
            const int grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>({(n_blocks + 3) / 4, int64_t(n_cu) * pipe_per_cu,
                                                                                      int64_t(kTallMaxGrid)})));
            grid_used = grid;
#define TGCN_PIPE(BI)                                                                                              \
    do {                                                                                                           \
        const void *fn = reinterpret_cast<const void *>(&k_gemm_pipe<7, 8, true, COLSUM, BI>);                     \
        TGCN_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lb))); \
        k_gemm_pipe<7, 8, true, COLSUM, BI><<<grid, 256, lb, s>>>(A, lda, B, ldb, C, ldc, N, n, drop, colpart);   \
    } while (0)
            if (from_record) TGCN_PIPE(true); else TGCN_PIPE(false);
#undef TGCN_PIPE
            return finish();
        }
    }
#endif
    if (whole && k == 64 && nt == 7) {
        if constexpr (DROP) {
            // the mask from the forward product's record: n in (192, 256] has four words per row half (16-byte reads)
            if (drop.bits && n > 192 && drop.bits_stride % 4 == 0 && reinterpret_cast<uintptr_t>(drop.bits) % 16 == 0) {
                lds_bytes += 4 * 1024;                  // the waves' mask slices; two workgroups per CU (one: 0.71 ms, two: 0.68)
                TGCN_TALL_KB(7, true, 8, true);
                return finish();
            }
        } else {
            lds_bytes = std::max(lds_bytes, kOnePerCu);
        }
        TGCN_TALL_K(7, true, 8);
        return finish();
    }
    }
    switch (nt) {
        case 1: TGCN_TALL(1); break;
        case 2: TGCN_TALL(2); break;
        case 3: TGCN_TALL(3); break;
        case 4: TGCN_TALL(4); break;
        case 5: TGCN_TALL(5); break;
        case 6: TGCN_TALL(6); break;
        case 7: TGCN_TALL(7); break;
        default: TGCN_TALL(8); break;
    }
#undef TGCN_TALL
#undef TGCN_TALL_K
#undef TGCN_TALL_KB
    return finish();
}

// The whole product: one launch when the small operand (k x n, padded) fits the LDS, otherwise column groups of
// <= 128 result columns (the tall operand is read once per group: these shapes sit on the fp32 MFMA roof, not on
// the HBM one) and, for reductions longer than 256, k chunks that accumulate into C.  A mask on the RESULT (nt
// with dropout) and the column sums are the business of the last chunk alone.
template <bool TRANS_B, bool DROP, bool COLSUM = false>
int launch_tall(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                int64_t N, int k, int n, const Drop drop, hipStream_t s, float *colpart = nullptr,
                float *colsum = nullptr) {
    const int kpad = (k + 7) & ~7, npad = 32 * ((n + 31) / 32);
    if (npad <= 256 && sizeof(float) * (size_t(kpad) * npad + (COLSUM ? 4 * npad : 0)) <= 160 * 1024)
        return launch_tall_one<TRANS_B, DROP, COLSUM>(A, lda, B, ldb, C, ldc, N, k, n, drop, s, colpart, colsum,
                                                      Place{0, 0, 0});
    for (int j0 = 0; j0 < n; j0 += kGroupCols) {
        const int ng = std::min(kGroupCols, n - j0);
        const float *Bg = TRANS_B ? B + int64_t(j0) * ldb : B + j0;
        for (int k0 = 0; k0 < k; k0 += kGroupK) {
            const int kg = std::min(kGroupK, k - k0);
            const bool last = k0 + kg >= k;
            const float *Bc = TRANS_B ? Bg + k0 : Bg + int64_t(k0) * ldb;
            const Place place{j0, k0, k0 > 0 ? 1 : 0};
            if (TRANS_B && !last) {
                TGCN_CHECK((launch_tall_one<TRANS_B, false, false>(A + k0, lda, Bc, ldb, C + j0, ldc, N, kg, ng, Drop{}, s,
                                                                  nullptr, nullptr, place)));
            } else {
                TGCN_CHECK((launch_tall_one<TRANS_B, DROP, COLSUM>(
                    A + k0, lda, Bc, ldb, C + j0, ldc, N, kg, ng, drop, s,
                    colpart ? colpart + size_t(kTallMaxGrid) * j0 : nullptr, colsum ? colsum + j0 : nullptr, place)));
            }
        }
    }
    return TGCN_OK;
}

int check_common(const char *fn, const void *a, const void *b, const void *c, int64_t N, int k, int n) {
    if (!a || !b || !c || N < 0 || k <= 0 || n <= 0) {
        set_error("%s: bad argument (N=%lld k=%d n=%d)", fn, (long long)N, k, n);
        return TGCN_E_INVALID;
    }
    return TGCN_OK;
}

}  // namespace
}  // namespace tgcn

extern "C" {

static int gemm_nn_impl(const char *fn, const float *A, int64_t lda, const float *B, int64_t ldb, float *C,
                        int64_t ldc, int64_t N, int k, int n, const tgcn::Drop *drop, tgcn_stream stream) {
    using namespace tgcn;
    TGCN_CHECK(check_common(fn, A, B, C, N, k, n));
    if (lda < k || ldb < n || ldc < n || lda % 4 != 0 || reinterpret_cast<uintptr_t>(A) % 16 != 0) {
        set_error("%s: need lda >= k, ldb, ldc >= n, lda %% 4 == 0 and A 16-byte aligned", fn);
        return TGCN_E_INVALID;
    }
    if (N == 0) return TGCN_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return drop ? launch_tall<false, true>(A, lda, B, ldb, C, ldc, N, k, n, *drop, s)
                : launch_tall<false, false>(A, lda, B, ldb, C, ldc, N, k, n, Drop{}, s);
}

static int gemm_nt_impl(const char *fn, const float *A, int64_t lda, const float *B, int64_t ldb, float *C,
                        int64_t ldc, int64_t N, int k, int n, const tgcn::Drop *drop, tgcn_stream stream,
                        float *colsum = nullptr, float *colpart = nullptr) {
    using namespace tgcn;
    TGCN_CHECK(check_common(fn, A, B, C, N, k, n));
    if (lda < k || ldb < k || ldc < n || lda % 4 != 0 || reinterpret_cast<uintptr_t>(A) % 16 != 0) {
        set_error("%s: need lda, ldb >= k, ldc >= n, lda %% 4 == 0 and A 16-byte aligned", fn);
        return TGCN_E_INVALID;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (N == 0) {
        if (colsum) TGCN_HIP_CHECK(hipMemsetAsync(colsum, 0, sizeof(float) * n, s));
        return TGCN_OK;
    }
    if (colsum)
        return drop ? launch_tall<true, true, true>(A, lda, B, ldb, C, ldc, N, k, n, *drop, s, colpart, colsum)
                    : launch_tall<true, false, true>(A, lda, B, ldb, C, ldc, N, k, n, Drop{}, s, colpart, colsum);
    return drop ? launch_tall<true, true>(A, lda, B, ldb, C, ldc, N, k, n, *drop, s)
                : launch_tall<true, false>(A, lda, B, ldb, C, ldc, N, k, n, Drop{}, s);
}

// tgcn_set_dropout_row_keys: which mask row a matrix row is, for the dropout products this THREAD launches next
static thread_local int64_t t_key_split = 0, t_key0 = 0, t_key1 = 0;

int tgcn_set_dropout_row_keys(int64_t split, int64_t key0, int64_t key1) {
    if (split < 0 || key0 < 0 || key1 < 0) {
        tgcn::set_error("tgcn_set_dropout_row_keys: split, key0 and key1 must be >= 0 (%lld, %lld, %lld)", (long long)split,
                        (long long)key0, (long long)key1);
        return TGCN_E_INVALID;
    }
    t_key_split = split;
    t_key0 = key0;
    t_key1 = key1;
    return TGCN_OK;
}

// p in [0, 1]; p = 1 drops everything (scale 0, as torch does)
static int make_drop(const char *fn, double p, const uint64_t *seed, int ld, tgcn::Drop &d) {
    if (!(p >= 0.0 && p <= 1.0) || !seed) {
        tgcn::set_error("%s: need 0 <= p <= 1 and a non-NULL device seed (p=%g)", fn, p);
        return TGCN_E_INVALID;
    }
    d.seed = seed;
    d.ld = ld;
    d.key_split = t_key_split;
    d.key0 = t_key0;
    d.key1 = t_key1;
    if (p >= 1.0) {
        d.thresh = 0xffffffffu;
        d.scale = 0.f;
    } else {
        d.thresh = static_cast<uint32_t>(std::min(p * 4294967296.0, 4294967295.0));
        d.scale = static_cast<float>(1.0 / (1.0 - p));
    }
    return TGCN_OK;
}

int tgcn_gemm_nn(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                 int64_t N, int k, int n, tgcn_stream stream) {
    return gemm_nn_impl("tgcn_gemm_nn", A, lda, B, ldb, C, ldc, N, k, n, nullptr, stream);
}

int tgcn_gemm_nt(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                 int64_t N, int k, int n, tgcn_stream stream) {
    return gemm_nt_impl("tgcn_gemm_nt", A, lda, B, ldb, C, ldc, N, k, n, nullptr, stream);
}

int tgcn_gemm_nn_dropout(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                         int64_t N, int k, int n, double p, const uint64_t *seed, tgcn_stream stream) {
    tgcn::Drop d{};
    TGCN_CHECK(make_drop("tgcn_gemm_nn_dropout", p, seed, k, d));
    return gemm_nn_impl("tgcn_gemm_nn_dropout", A, lda, B, ldb, C, ldc, N, k, n, &d, stream);
}

size_t tgcn_dropout_mask_words(int k, int n) {
    // recorded by the fp32 nn product whenever its reduction runs in ONE piece (k <= 256: a launch then sees whole rows of
    // the masked operand; a wide result runs as column groups, each of which writes the same bits) and the split-bf16
    // mode is off (its kernels do not write the record)
    if (k <= 0 || n <= 0 || k > tgcn::kGroupK || tgcn::g_gemm_split.load(std::memory_order_relaxed) != 0) return 0;
    return 2 * static_cast<size_t>(tgcn::drop_bits_wph(k));
}

int tgcn_gemm_nn_dropout_mask(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                              int64_t N, int k, int n, double p, const uint64_t *seed, uint32_t *mask,
                              int64_t mask_stride, tgcn_stream stream) {
    const char *fn = "tgcn_gemm_nn_dropout_mask";
    const size_t words = tgcn_dropout_mask_words(k, n);
    if (!mask || words == 0 || mask_stride < static_cast<int64_t>(words)) {
        tgcn::set_error("%s: the mask of a [N x %d] operand takes %zu words per row (0: this product cannot record it); "
                        "mask=%p, stride %lld", fn, k, words, static_cast<void *>(mask), (long long)mask_stride);
        return TGCN_E_INVALID;
    }
    tgcn::Drop d{};
    TGCN_CHECK(make_drop(fn, p, seed, k, d));
    d.bits = mask;
    d.bits_stride = mask_stride;
    return gemm_nn_impl(fn, A, lda, B, ldb, C, ldc, N, k, n, &d, stream);
}

int tgcn_gemm_nt_dropout(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                         int64_t N, int k, int n, double p, const uint64_t *seed, tgcn_stream stream) {
    tgcn::Drop d{};
    TGCN_CHECK(make_drop("tgcn_gemm_nt_dropout", p, seed, n, d));
    return gemm_nt_impl("tgcn_gemm_nt_dropout", A, lda, B, ldb, C, ldc, N, k, n, &d, stream);
}

int tgcn_set_gemm_split(int on) { return tgcn::g_gemm_split.exchange(on != 0 ? 1 : 0); }

size_t tgcn_gemm_nt_colsum_workspace_bytes(int n) {
    return n > 0 ? sizeof(float) * static_cast<size_t>(tgcn::kTallMaxGrid) * static_cast<size_t>(n) : 0;
}

int tgcn_gemm_nt_colsum(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                        int64_t N, int k, int n, double p, const uint64_t *seed, float *colsum, void *workspace,
                        size_t workspace_bytes, tgcn_stream stream) {
    const char *fn = "tgcn_gemm_nt_colsum";
    if (!colsum || n <= 0 || !workspace || workspace_bytes < tgcn_gemm_nt_colsum_workspace_bytes(n)) {
        tgcn::set_error("%s: colsum and a workspace of %zu bytes are required (%zu given)", fn,
                        tgcn_gemm_nt_colsum_workspace_bytes(n), workspace_bytes);
        return colsum && n > 0 ? TGCN_E_WORKSPACE : TGCN_E_INVALID;
    }
    float *colpart = static_cast<float *>(workspace);
    if (!seed) return gemm_nt_impl(fn, A, lda, B, ldb, C, ldc, N, k, n, nullptr, stream, colsum, colpart);
    tgcn::Drop d{};
    TGCN_CHECK(make_drop(fn, p, seed, n, d));
    return gemm_nt_impl(fn, A, lda, B, ldb, C, ldc, N, k, n, &d, stream, colsum, colpart);
}

int tgcn_gemm_nt_colsum_mask(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                             int64_t N, int k, int n, double p, const uint64_t *seed, const uint32_t *mask,
                             int64_t mask_stride, float *colsum, void *workspace, size_t workspace_bytes,
                             tgcn_stream stream) {
    const char *fn = "tgcn_gemm_nt_colsum_mask";
    if (!colsum || n <= 0 || !workspace || workspace_bytes < tgcn_gemm_nt_colsum_workspace_bytes(n)) {
        tgcn::set_error("%s: colsum and a workspace of %zu bytes are required (%zu given)", fn,
                        tgcn_gemm_nt_colsum_workspace_bytes(n), workspace_bytes);
        return colsum && n > 0 ? TGCN_E_WORKSPACE : TGCN_E_INVALID;
    }
    if (!mask || !seed || mask_stride < 2 * static_cast<int64_t>(tgcn::drop_bits_wph(n))) {
        tgcn::set_error("%s: need the seed and the mask tgcn_gemm_nn_dropout_mask recorded for the [N x %d] operand (rows of "
                        ">= %d words; stride %lld)", fn, n, 2 * tgcn::drop_bits_wph(n), (long long)mask_stride);
        return TGCN_E_INVALID;
    }
    tgcn::Drop d{};
    TGCN_CHECK(make_drop(fn, p, seed, n, d));
    d.bits = const_cast<uint32_t *>(mask);       // read only on this side
    d.bits_stride = mask_stride;
    return gemm_nt_impl(fn, A, lda, B, ldb, C, ldc, N, k, n, &d, stream, colsum, static_cast<float *>(workspace));
}

size_t tgcn_gemm_tn_workspace_bytes(int64_t N, int k, int n) {
    if (N < 0 || k <= 0 || n <= 0) return 0;
    // one launch's partial tiles (wider products reuse the region launch after launch)
    size_t nt = (std::min(n, tgcn::kGroupCols) + 31) / 32;
    if (nt == 3) nt = 4;                                  // three column tiles run as four in the LDS-staged kernel
    const size_t mpad = 32 * ((std::min(k, tgcn::kGroupK) + 31) / 32), npad = 32 * nt;
    return sizeof(float) * static_cast<size_t>(tgcn::tn_blocks(N)) * mpad * npad;
}

static int gemm_tn_impl(const char *fn, const float *A, int64_t lda, const float *G, int64_t ldg, float *C,
                        int64_t ldc, int64_t N, int k, int n, void *workspace, size_t workspace_bytes,
                        const tgcn::Drop *drop, tgcn_stream stream) {
    using namespace tgcn;
    TGCN_CHECK(check_common(fn, A, G, C, N, k, n));
    if (lda < k || ldg < n || ldc < n) {
        set_error("%s: need lda >= k and ldg, ldc >= n", fn);
        return TGCN_E_INVALID;
    }
    const size_t need = tgcn_gemm_tn_workspace_bytes(N, k, n);
    if (!workspace || workspace_bytes < need) {
        set_error("%s: workspace of %zu bytes given, %zu needed", fn, workspace_bytes, need);
        return TGCN_E_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = tn_blocks(N);
    const int64_t rows_per_wg = ((N + nb - 1) / nb + 1) & ~int64_t(1);  // even: steps are row pairs
    const int nt = (n + 31) / 32, mt = (k + 31) / 32;
    float *partial = static_cast<float *>(workspace);
    if (g_gemm_split.load(std::memory_order_relaxed) != 0 && k == 200 && nt == 2 && lda == k && ldg == n && n % 4 == 0 &&
        reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(G) % 16 == 0) {
        // split-bf16 products for the weight gradient of the GCN's second layer (16-row K steps through LDS)
        const int64_t rpw = ((N + nb - 1) / nb + 31) & ~int64_t(31);
        const size_t lb = sizeof(float) * 2 * (32 * (k + 4) + 32 * (n + 4));
        if (drop) {
            TGCN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_tn_split<2, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lb)));
            k_gemm_tn_split<2, true><<<nb, 256, lb, s>>>(A, G, N, k, n, rpw, partial, *drop);
        } else {
            TGCN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_tn_split<2, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lb)));
            k_gemm_tn_split<2, false><<<nb, 256, lb, s>>>(A, G, N, k, n, rpw, partial, Drop{});
        }
        TGCN_HIP_CHECK(hipGetLastError());
        k_gemm_tn_reduce<<<(k * n + 63) / 64, 256, 0, s>>>(partial, nb, 32 * mt, 32 * nt, k, n, C, ldc);
        TGCN_HIP_CHECK(hipGetLastError());
        return TGCN_OK;
    }
    // One launch covers <= 256 columns of A (8 M-tiles, two per wave) x <= 128 columns of G (4 accumulator tiles per
    // M-tile); wider products run as several launches over column chunks of A and / or G, each followed by its
    // reduction -- they share the workspace, in stream order.
    for (int i0 = 0; i0 < k; i0 += kGroupK) {
        const int kg = std::min(kGroupK, k - i0);
        for (int j0 = 0; j0 < n; j0 += kGroupCols) {
            const int ng = std::min(kGroupCols, n - j0);
            const int ntg = (ng + 31) / 32, mtg = (kg + 31) / 32;
            const float *Ag = A + i0, *Gg = G + j0;
            // 16-byte friendly operands with 1, 2 or 4 column tiles of G: the LDS-staged kernel (rows split over the waves)
            // A width that is not a multiple of 4 (DBpedia's 219 classes: the second group holds 91 columns) is served when
            // the row reaches to the next multiple of 4 -- a zero-padded buffer (plan.alloc_padded), or columns of a wider
            // matrix: what is read there lands in result columns nobody stores.  Three tiles run as four.
            const int ng4 = (ng + 3) & ~3;
            const int nts = ntg == 3 ? 4 : ntg;
            const bool staged = (nts == 1 || nts == 2 || nts == 4) && kg % 4 == 0 && lda % 4 == 0 && ldg % 4 == 0 &&
                                int64_t(j0) + ng4 <= ldg && reinterpret_cast<uintptr_t>(Ag) % 16 == 0 &&
                                reinterpret_cast<uintptr_t>(Gg) % 16 == 0 && tn_staged_lds_bytes(kg, ng4) <= 160 * 1024 &&
                                std::getenv("TGCN_TN_STAGED_OFF") == nullptr;
            if (staged) {
                const int64_t rpw = ((N + nb - 1) / nb + kTnStageRows - 1) / kTnStageRows * kTnStageRows;
                const size_t lb = tn_staged_lds_bytes(kg, ng4);
#define TGCN_TNS2(NT, MT_, DR, BI)                                                                                    \
    do {                                                                                                              \
        TGCN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_tn_staged<NT, MT_, DR, BI>),       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lb)));       \
        k_gemm_tn_staged<NT, MT_, DR, BI><<<nb, 256, lb, s>>>(Ag, lda, Gg, ldg, N, kg, ng4, rpw, partial,            \
                                                              drop ? *drop : Drop{}, i0);                           \
    } while (0)
                // the recorded mask describes whole rows of A: usable when this launch covers all k columns
                const bool from_bits = drop && drop->bits && i0 == 0 && kg == k;
#define TGCN_TNS(NT, MT_, DR)                                                                                         \
    do {                                                                                                              \
        if (DR && from_bits) TGCN_TNS2(NT, MT_, DR, DR); else TGCN_TNS2(NT, MT_, DR, false);                         \
    } while (0)
#define TGCN_TNS_M(NT, DR)                                                                                            \
    do {                                                                                                              \
        if (mtg <= 2) TGCN_TNS(NT, 2, DR); else if (mtg <= 4) TGCN_TNS(NT, 4, DR);                                   \
        else if (mtg <= 7) TGCN_TNS(NT, 7, DR); else TGCN_TNS(NT, 8, DR);                                           \
    } while (0)
                if (drop) {
                    if (nts == 1) TGCN_TNS_M(1, true); else if (nts == 2) TGCN_TNS_M(2, true); else TGCN_TNS_M(4, true);
                } else {
                    if (nts == 1) TGCN_TNS_M(1, false); else if (nts == 2) TGCN_TNS_M(2, false); else TGCN_TNS_M(4, false);
                }
#undef TGCN_TNS_M
#undef TGCN_TNS
#undef TGCN_TNS2
                TGCN_HIP_CHECK(hipGetLastError());
                k_gemm_tn_reduce<<<(kg * ng + 63) / 64, 256, 0, s>>>(partial, nb, 32 * mtg, 32 * nts, kg, ng,
                                                                     C + int64_t(i0) * ldc + j0, ldc,
                                                                     drop ? drop->scale : 1.0f);
                TGCN_HIP_CHECK(hipGetLastError());
                continue;
            }
#define TGCN_TN(NT)                                                                                                   \
    do {                                                                                                              \
        if (drop)                                                                                                     \
            k_gemm_tn_partial<NT, true><<<nb, 256, 0, s>>>(Ag, lda, Gg, ldg, N, kg, ng, rows_per_wg, partial, *drop, i0); \
        else                                                                                                          \
            k_gemm_tn_partial<NT, false><<<nb, 256, 0, s>>>(Ag, lda, Gg, ldg, N, kg, ng, rows_per_wg, partial, Drop{}, i0); \
    } while (0)
            switch (ntg) {
                case 1: TGCN_TN(1); break;
                case 2: TGCN_TN(2); break;
                case 3: TGCN_TN(3); break;
                default: TGCN_TN(4); break;
            }
#undef TGCN_TN
            TGCN_HIP_CHECK(hipGetLastError());
            k_gemm_tn_reduce<<<(kg * ng + 63) / 64, 256, 0, s>>>(partial, nb, 32 * mtg, 32 * ntg, kg, ng,
                                                                 C + int64_t(i0) * ldc + j0, ldc);
            TGCN_HIP_CHECK(hipGetLastError());
        }
    }
    return TGCN_OK;
}

int tgcn_gemm_tn(const float *A, int64_t lda, const float *G, int64_t ldg, float *C, int64_t ldc,
                 int64_t N, int k, int n, void *workspace, size_t workspace_bytes, tgcn_stream stream) {
    return gemm_tn_impl("tgcn_gemm_tn", A, lda, G, ldg, C, ldc, N, k, n, workspace, workspace_bytes, nullptr, stream);
}

int tgcn_gemm_tn_dropout(const float *A, int64_t lda, const float *G, int64_t ldg, float *C, int64_t ldc,
                         int64_t N, int k, int n, double p, const uint64_t *seed, void *workspace,
                         size_t workspace_bytes, tgcn_stream stream) {
    tgcn::Drop d{};
    TGCN_CHECK(make_drop("tgcn_gemm_tn_dropout", p, seed, k, d));
    return gemm_tn_impl("tgcn_gemm_tn_dropout", A, lda, G, ldg, C, ldc, N, k, n, workspace, workspace_bytes, &d,
                        stream);
}

int tgcn_gemm_tn_dropout_mask(const float *A, int64_t lda, const float *G, int64_t ldg, float *C, int64_t ldc,
                              int64_t N, int k, int n, double p, const uint64_t *seed, const uint32_t *mask,
                              int64_t mask_stride, void *workspace, size_t workspace_bytes, tgcn_stream stream) {
    const char *fn = "tgcn_gemm_tn_dropout_mask";
    if (!mask || k <= 0 || mask_stride < 2 * static_cast<int64_t>(tgcn::drop_bits_wph(k))) {
        tgcn::set_error("%s: need the mask tgcn_gemm_nn_dropout_mask recorded for this operand (rows of >= %d words; "
                        "stride %lld)", fn, k > 0 ? 2 * tgcn::drop_bits_wph(k) : 0, (long long)mask_stride);
        return TGCN_E_INVALID;
    }
    tgcn::Drop d{};
    TGCN_CHECK(make_drop(fn, p, seed, k, d));
    d.bits = const_cast<uint32_t *>(mask);       // read only on this side
    d.bits_stride = mask_stride;
    return gemm_tn_impl(fn, A, lda, G, ldg, C, ldc, N, k, n, workspace, workspace_bytes, &d, stream);
}

}  // extern "C"
