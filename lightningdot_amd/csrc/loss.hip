// In-batch contrastive loss kernels (fp32-exact):
//   scores = (1-w) q.ctx^T + w q.cap^T ; log-softmax rows ; NLL at the positive ; arg-max == positive count
// Reference: dvl/models/bi_encoder.py:54-68 (dot_product_scores), :615-656 (BiEncoderNllLoss.calc).
// The contraction uses v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate, bit-for-bit an fmaf chain), so the
// loss and its gradients match an fp32 reference to rounding; the problem (<= 512 x 1536 x 768) is latency
// bound, not MFMA bound, so the 1/16-rate fp32 matrix path costs nothing that matters.
#include <math.h>

#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

#include "kernels.h"

namespace ldot {

constexpr int kSgThreads = 256;
constexpr int kSgSlab = 2048;   // operand elements per K slab: (TILE 64, BK 32) or (TILE 128, BK 16) = 2 float4 per thread

// One operand slab: global -> registers (float4 along the operand's contiguous dimension, bounds-checked) -> LDS [k][r].
// KFAST: element (r, k) at r*ld + k (rows of the operand are K-contiguous), else at k*ld + r.
template <int TILE, bool KFAST>
struct SgOperand {
    static constexpr int BK = kSgSlab / TILE;
    static constexpr int LD = TILE + 4;   // float4-aligned rows; the two k rows of an MFMA operand land 4 banks apart
    float4 v[2];
    __device__ __forceinline__ void fetch(const float* __restrict__ P, int64_t ld, int64_t r0, int64_t R, int64_t k0,
                                          int64_t K, bool vec) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = threadIdx.x + i * kSgThreads;
            int64_t r, k;
            if (KFAST) {
                r = r0 + id / (BK / 4);
                k = k0 + (id % (BK / 4)) * 4;
            } else {
                k = k0 + id / (TILE / 4);
                r = r0 + (id % (TILE / 4)) * 4;
            }
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KFAST) {
                if (r < R) {
                    const float* p = P + r * ld + k;
                    if (vec && k + 3 < K) {
                        x = *(const float4*)p;
                    } else {
                        if (k + 0 < K) x.x = p[0];
                        if (k + 1 < K) x.y = p[1];
                        if (k + 2 < K) x.z = p[2];
                        if (k + 3 < K) x.w = p[3];
                    }
                }
            } else {
                if (k < K) {
                    const float* p = P + k * ld + r;
                    if (vec && r + 3 < R) {
                        x = *(const float4*)p;
                    } else {
                        if (r + 0 < R) x.x = p[0];
                        if (r + 1 < R) x.y = p[1];
                        if (r + 2 < R) x.z = p[2];
                        if (r + 3 < R) x.w = p[3];
                    }
                }
            }
            v[i] = x;
        }
    }
    __device__ __forceinline__ void stash(float (*S)[LD]) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = threadIdx.x + i * kSgThreads;
            if (KFAST) {
                const int r = id / (BK / 4), k = (id % (BK / 4)) * 4;
                S[k + 0][r] = v[i].x;
                S[k + 1][r] = v[i].y;
                S[k + 2][r] = v[i].z;
                S[k + 3][r] = v[i].w;
            } else {
                const int k = id / (TILE / 4), r = (id % (TILE / 4)) * 4;
                *(float4*)&S[k][r] = v[i];
            }
        }
    }
};

// C[m][n] (+)= alpha * sum_k A(m,k) * B(n,k)   (fp32 MFMA 32x32x2: exact fp32 products, fp32 accumulation in K order)
// TILE x TILE output per workgroup, 4 waves as 2 x 2, (TILE/64)^2 MFMA tiles per wave; the next K slab is fetched into
// registers while the current one is multiplied.
template <int TILE, bool AK, bool BKF>
__global__ __launch_bounds__(kSgThreads) void sgemm_kernel(const float* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ B, int64_t ldb,
                                                           float* __restrict__ C, int64_t ldc, int64_t M, int64_t N,
                                                           int64_t K, float alpha, int accumulate, int vec) {
    constexpr int BK = kSgSlab / TILE, LD = TILE + 4, WT = TILE / 64;
    // two LDS stages: slab k + 1 is written (from the registers its global loads landed in) while slab k is being multiplied by waves
    // that are behind — ONE barrier per slab (the single-stage version needed two; 4096 x 4096 x 768: profiles/r05_loss_trace_4096.txt)
    __shared__ __attribute__((aligned(16))) float As[2][BK][LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t m0 = (int64_t)blockIdx.y * TILE, n0 = (int64_t)blockIdx.x * TILE;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    SgOperand<TILE, AK> oa;
    SgOperand<TILE, BKF> ob;
    oa.fetch(A, lda, m0, M, 0, K, vec & 1);
    ob.fetch(B, ldb, n0, N, 0, K, (vec >> 1) & 1);
    oa.stash(As[0]);
    ob.stash(Bs[0]);
    __syncthreads();
    int cur = 0;
    for (int64_t k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) {
            oa.fetch(A, lda, m0, M, k0 + BK, K, vec & 1);
            ob.fetch(B, ldb, n0, N, k0 + BK, K, (vec >> 1) & 1);
        }
#pragma unroll
        for (int ks = 0; ks < BK; ks += 2) {
            float a[WT], b[WT];
#pragma unroll
            for (int i = 0; i < WT; ++i) a[i] = As[cur][ks + (lane >> 5)][(wm * WT + i) * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < WT; ++j) b[j] = Bs[cur][ks + (lane >> 5)][(wn * WT + j) * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) {     // (stage cur ^ 1 was last read two slabs ago: every wave has passed the barrier after that)
            oa.stash(As[cur ^ 1]);
            ob.stash(Bs[cur ^ 1]);
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const int64_t n = n0 + (wn * WT + j) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + (wm * WT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < M && n < N) {
                    const float v = __fmul_rn(alpha, acc[i][j][r]);
                    float* c = C + m * ldc + n;
                    *c = accumulate ? __fadd_rn(*c, v) : v;
                }
            }
        }
}

template <int TILE>
static void sgemm_dispatch(bool ak, bool bk, dim3 grid, hipStream_t st, const float* A, int64_t lda, const float* B,
                           int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha,
                           int accumulate, int vec) {
    if (ak && bk)
        hipLaunchKernelGGL((sgemm_kernel<TILE, true, true>), grid, dim3(kSgThreads), 0, st, A, lda, B, ldb, C, ldc, M, N, K,
                           alpha, accumulate, vec);
    else if (ak && !bk)
        hipLaunchKernelGGL((sgemm_kernel<TILE, true, false>), grid, dim3(kSgThreads), 0, st, A, lda, B, ldb, C, ldc, M, N,
                           K, alpha, accumulate, vec);
    else if (!ak && bk)
        hipLaunchKernelGGL((sgemm_kernel<TILE, false, true>), grid, dim3(kSgThreads), 0, st, A, lda, B, ldb, C, ldc, M, N,
                           K, alpha, accumulate, vec);
    else
        hipLaunchKernelGGL((sgemm_kernel<TILE, false, false>), grid, dim3(kSgThreads), 0, st, A, lda, B, ldb, C, ldc, M, N,
                           K, alpha, accumulate, vec);
}


// ---- small problems (the in-batch loss itself: <= 512 x 1536 x 768): direct-operand GEMM ---------------------------------------------
// The shapes of the loss are latency class (0.4 - 1.2 GFLOP): the 64 x 64 LDS-staged kernel above runs 64 - 192 workgroups through 24
// barrier-separated slabs and takes 30 - 46 us where the flops are worth 3 - 8.  Here a workgroup owns ONE 32 x 32 output tile (256 - 768
// workgroups for the forward shapes) and its KS (4 or 8) waves split K between them; every wave feeds v_mfma_f32_32x32x2_f32 straight from
// global memory — no LDS staging, no barrier inside the K loop:
//   * lane (r = l & 31, h = l >> 5) of the MFMA holds A[m0 + r][k] and B[n0 + r][k] for k = k0 + h of a K pair; a chunk of 32 k is
//     consumed as 16 MFMAs in the order k = kc + 16 h + t, t = 0..15 (both operands use the same permutation of K inside a chunk, so
//     every product pairs the right elements), which lets a lane fetch its 16 values of a K-contiguous operand as four float4
//     (a full 128-B line per row and lane pair), or 16 coalesced dwords of an operand stored K-major;
//   * chunks are dealt round-robin to the waves (wave w: chunks w, w + KS, ...), NB chunk buffers per wave = NB - 1 chunks in flight
//     behind the one being multiplied;
//   * the KS partial tiles meet in LDS once, are summed in a fixed order (((w0 + w1) + (w2 + w3)) [+ ((w4 + w5) + (w6 + w7))]) and
//     leave coalesced.
// The sum over K is a different (still fixed) order than the fmaf chain of the big kernel; both are exact-product fp32 sums.
template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}

constexpr int kTrLd = 36;                 // floats per row of a wave's transposition area (conflict-free 16-byte writes and reads)
constexpr int kTrFloats = 32 * kTrLd;     // one 32 x 32 chunk

template <bool KFAST>
struct DirectChunk {
    float v[16];
    // K-major operand (element (row, k) at k * ld + row): lane (r = lane & 31, h = lane >> 5) loads its MFMA values directly,
    // k = k0 + 16 h + t for t = 0..15 — 16 dword loads, each wave load = two full 128-byte lines.
    // K-contiguous operand (element (row, k) at row * ld + k): the same distribution would make every lane of a load touch a different
    // 128-byte line (32 lines per instruction, each visited by four instructions: the L1 tag rate, not the data, bounded the first
    // version of this kernel at ~25 us for 256 workgroups).  Instead the chunk is fetched COALESCED — instruction i covers rows 8 i .. 8 i + 7,
    // eight lanes per row, 16 bytes each: 8 full lines per instruction — and redistributed to the MFMA layout through a wave-private
    // LDS area right before its MFMAs (to_mfma: 4 x 16-byte writes, 4 x 16-byte reads, no barrier: one wave's LDS operations execute
    // in order).  Rows beyond R are clamped to R - 1: they are computed and never stored.
    // (whole chunks and 16-byte aligned rows only: the launcher sends ragged K and unaligned operands to the staged kernel — a run-time
    // "fast" flag made the compiler duplicate every load behind a branch)
    __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int64_t r0, int64_t R, int64_t k0, int lane) {
        if (KFAST) {
            const int kq = 4 * (lane & 7);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = min(r0 + 8 * i + (lane >> 3), R - 1);
                const float4 x = *(const float4*)(P + row * ld + k0 + kq);
                v[4 * i + 0] = x.x;
                v[4 * i + 1] = x.y;
                v[4 * i + 2] = x.z;
                v[4 * i + 3] = x.w;
            }
        } else {
            const int64_t row = min(r0 + (lane & 31), R - 1);
            // the 16 addresses of a lane differ by multiples of ld — a UNIFORM base per k (scalar registers) plus one 32-bit lane offset
            // (row inside the panel, + 16 rows of K for the upper half wave) instead of 16 address pairs per chunk buffer
            // (the launcher routes operands with 16 ld + 32 >= 2^31 elements to the staged kernel)
            const uint32_t lo = (uint32_t)(row - r0) + (uint32_t)(16 * (lane >> 5)) * (uint32_t)ld;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float* pu = P + (k0 + t) * ld + r0;   // uniform
                v[t] = pu[lo];
            }
        }
    }
    // K-contiguous operands only: coalesced layout -> MFMA layout (lane (r, h): k = 16 h + t) through the wave's LDS area `tr`
    __device__ __forceinline__ void to_mfma(float* tr, int lane) {
        if (!KFAST) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(float4*)(tr + (8 * i + (lane >> 3)) * kTrLd + 4 * (lane & 7)) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        const float* rd = tr + (lane & 31) * kTrLd + 16 * (lane >> 5);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 x = *(const float4*)(rd + 4 * c);
            v[4 * c + 0] = x.x;
            v[4 * c + 1] = x.y;
            v[4 * c + 2] = x.z;
            v[4 * c + 3] = x.w;
        }
    }
};

struct NllFused {   // forward epilogue of the loss (EPI >= 1): per (row, column tile) statistics for the finish kernels
    const int32_t* pos;
    uint4* stat;         // {max, sum exp(s - max), first arg-max column, -} of a row over one column tile: stat[m * ldstat + stat_col0 + tile]
    float* spos;         // the positive's score per row
    int ldstat, stat_col0;
    int64_t col_off;     // global column of this GEMM's column 0 (arg-max and positive indices are global)
    // EPI 2 (the bidirectional train step, train_itm.py:195-222): the OTHER direction's score matrix is the transpose of this one over
    // the first `nshared` columns — S_img[n][m] = S_txt[m][n] — so the tile also leaves its transpose and the COLUMN statistics
    float* Ct;           // transposed scores, Ct[n * ldct + m]
    int64_t ldct, nshared;
    uint4* stat_t;       // stat_t[n * ldstat_t + row tile]
    int ldstat_t;
    float* spos_t;       // Ct[n][pos[n]]
};

// max / first arg-max / sum of exponentials of 32 values held by the 32 lanes of a half wave (idx = the value's global index)
__device__ __forceinline__ void half_wave_softmax_stats(float v, bool valid, int64_t idx, float& mx, int64_t& am, float& z) {
    mx = valid ? v : -INFINITY;
    am = valid ? idx : 0x7fffffffffffffffll;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(mx, o);
        const int64_t a2 = __shfl_xor(am, o);
        if (m2 > mx || (m2 == mx && a2 < am)) {
            mx = m2;
            am = a2;
        }
    }
    z = valid ? expf(v - mx) : 0.f;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) z += __shfl_xor(z, o);
}

// EPI 0: C = alpha * acc (+ C).  EPI 1 (forward of the loss): C = mix, + the softmax statistics of the tile's rows over its 32 columns.
// (body of sgemm_direct_kernel; smem: KS * (MIX ? 3 : 2) * kTrFloats floats; (bx, by) = the workgroup's tile)
template <bool AKF, bool BKF, bool MIX, int EPI, int KS>
__device__ __forceinline__ void sgemm_direct_body(float* __restrict__ smem, const int bx, const int by, const float* __restrict__ A,
                                                  int64_t lda, const float* __restrict__ B, int64_t ldb, const float* __restrict__ B2,
                                                  float* __restrict__ C, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha,
                                                  float w2, int accumulate, const NllFused& nll) {
    // LDS: the waves' transposition areas (A, B, B2) during the K loop, the four partial tiles afterwards
    constexpr int kTrOps = MIX ? 3 : 2;
    static_assert(KS * 32 * 33 <= KS * 2 * kTrFloats, "partial tiles alias the transposition areas");
    float (*red)[32][33] = (float (*)[32][33])smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: scalar chunk addresses)
    const int64_t m0 = (int64_t)by * 32, n0 = (int64_t)bx * 32;
    const int64_t nch = (K + 31) / 32;
    f32x16 acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = acc2[r] = 0.f;
    // chunk buffers per wave: all six chunks of a wave's K quarter in flight at K = 768 (one memory round trip in front of the MFMAs
    // instead of three: the problem is latency class); the mix carries a second B operand and keeps four
    // (KS = 8: eight waves split K — three chunks each at K = 768 — and run two per SIMD: three buffers)
    constexpr int NB = KS == 8 ? 3 : MIX ? 4 : 6;
    DirectChunk<AKF> a[NB];
    DirectChunk<BKF> b[NB], b2[MIX ? NB : 1];
    auto fetch = [&](auto bi_tag, int64_t c) {
        constexpr int bi = decltype(bi_tag)::value;
        a[bi].load(A, lda, m0, M, c * 32, lane);
        b[bi].load(B, ldb, n0, N, c * 32, lane);
        if (MIX) b2[MIX ? bi : 0].load(B2, ldb, n0, N, c * 32, lane);
    };
    float* tr = smem + wave * (kTrOps * kTrFloats);
    auto mult = [&](auto bi_tag) {
        constexpr int bi = decltype(bi_tag)::value;
        a[bi].to_mfma(tr, lane);
        b[bi].to_mfma(tr + kTrFloats, lane);
        if (MIX) b2[MIX ? bi : 0].to_mfma(tr + 2 * kTrFloats, lane);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[bi].v[t], b[bi].v[t], acc1, 0, 0, 0);
            if (MIX) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[bi].v[t], b2[MIX ? bi : 0].v[t], acc2, 0, 0, 0);
        }
    };
    // wave w multiplies the chunks w, w + KS, ...; chunk number j of a wave lives in buffer j % NB and is fetched NB - 1 chunks ahead
    static_for<NB - 1>([&](auto j) {
        if (wave + KS * (int64_t)decltype(j)::value < nch) fetch(j, wave + KS * (int64_t)decltype(j)::value);
    });
    for (int64_t c = wave; c < nch; c += KS * NB) {
        static_for<NB>([&](auto j) {
            constexpr int jj = decltype(j)::value;
            const int64_t cc = c + KS * jj;
            if (cc < nch) {
                if (cc + KS * (NB - 1) < nch) fetch(std::integral_constant<int, (jj + NB - 1) % NB>{}, cc + KS * (NB - 1));
                mult(j);
            }
        });
    }
    // partial tiles -> LDS (MFMA layout: lane holds column lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
    constexpr int ITER = 1024 / (64 * KS);      // outputs per thread
    float fin[ITER];
    for (int pass = 0; pass < (MIX ? 2 : 1); ++pass) {
        __syncthreads();   // (every wave is done with its transposition area / with the previous pass)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = pass ? acc2[r] : acc1[r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int m = (threadIdx.x >> 5) + 2 * KS * i, n = threadIdx.x & 31;
            float sum = (red[0][m][n] + red[1][m][n]) + (red[2][m][n] + red[3][m][n]);
            if (KS == 8) sum += (red[4][m][n] + red[5][m][n]) + (red[6][m][n] + red[7][m][n]);
            // (1 - w) q.ctx + w q.cap exactly like the reference: two rounded products, one rounded add
            fin[i] = pass ? __fadd_rn(fin[i], __fmul_rn(w2, sum)) : __fmul_rn(alpha, sum);
        }
    }
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int64_t m = m0 + (threadIdx.x >> 5) + 2 * KS * i, n = n0 + (threadIdx.x & 31);
        if (m < M && n < N) {
            float* cp = C + m * ldc + n;
            if (accumulate) fin[i] = __fadd_rn(*cp, fin[i]);
            *cp = fin[i];
        }
    }
    if (EPI == 0) return;

    // ---- fused NLL forward: statistics of this tile's 32 columns for each of its rows (a row = the 32 lanes t & 31 of a half wave)
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int64_t m = m0 + (threadIdx.x >> 5) + 2 * KS * i, n = n0 + (threadIdx.x & 31);
        float mx, z;
        int64_t am;
        half_wave_softmax_stats(fin[i], n < N, n + nll.col_off, mx, am, z);
        if (m < M) {
            if ((threadIdx.x & 31) == 0)
                nll.stat[m * nll.ldstat + nll.stat_col0 + bx] = make_uint4(__float_as_uint(mx), __float_as_uint(z), (uint32_t)am, 0u);
            if (n < N && (int64_t)nll.pos[m] == n + nll.col_off) nll.spos[m] = fin[i];
        }
    }
    if (EPI != 2 || n0 >= nll.nshared) return;   // (uniform)
    // ---- the other direction: the tile transposed through LDS — stored as rows of Ct, and its COLUMN statistics
    __syncthreads();   // (every thread is done with the partial tiles)
#pragma unroll
    for (int i = 0; i < ITER; ++i) red[0][(threadIdx.x >> 5) + 2 * KS * i][threadIdx.x & 31] = fin[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int c = (threadIdx.x >> 5) + 2 * KS * i, r = threadIdx.x & 31;
        const int64_t n = n0 + c, m = m0 + r;
        const float v = red[0][r][c];
        const bool col_ok = n < N && n < nll.nshared;
        float mx, z;
        int64_t am;
        half_wave_softmax_stats(v, m < M, m, mx, am, z);
        if (col_ok) {
            if (m < M) {
                nll.Ct[n * nll.ldct + m] = v;
                if ((int64_t)nll.pos[n] == m) nll.spos_t[n] = v;
            }
            if (r == 0) nll.stat_t[n * nll.ldstat_t + by] = make_uint4(__float_as_uint(mx), __float_as_uint(z), (uint32_t)am, 0u);
        }
    }
}

template <bool AKF, bool BKF, bool MIX, int EPI, int KS>
__global__ __launch_bounds__(64 * KS) void sgemm_direct_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                           int64_t ldb, const float* __restrict__ B2, float* __restrict__ C, int64_t ldc,
                                                           int64_t M, int64_t N, int64_t K, float alpha, float w2, int accumulate,
                                                           NllFused nll) {
    __shared__ __attribute__((aligned(16))) float smem[KS * (MIX ? 3 : 2) * kTrFloats];
    sgemm_direct_body<AKF, BKF, MIX, EPI, KS>(smem, (int)blockIdx.x, (int)blockIdx.y, A, lda, B, ldb, B2, C, ldc, M, N, K, alpha, w2,
                                              accumulate, nll);
}

// The two GEMMs of the bidirectional backward at num_hard_negatives = 0 in ONE launch (the step's cost is launches, not flops):
//   workgroups [0, g0):       dimg = dS . txt      (A = dS K-contiguous, B = txt K-major)
//   workgroups [g0, g0 + g1): dtxt = dS^T . img    (A = dS K-major,      B = img K-major)
struct SgPairSide {
    const float *A, *B;
    float* C;
    int64_t lda, ldb, ldc, M, N, K;
    int gx, gy;
};
__global__ __launch_bounds__(256) void sgemm_direct_pair_kernel(SgPairSide p0, SgPairSide p1) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 2 * kTrFloats];
    const int g0 = p0.gx * p0.gy;
    if ((int)blockIdx.x < g0) {
        const int b = blockIdx.x;
        sgemm_direct_body<true, false, false, 0, 4>(smem, b % p0.gx, b / p0.gx, p0.A, p0.lda, p0.B, p0.ldb, nullptr, p0.C, p0.ldc, p0.M,
                                                    p0.N, p0.K, 1.f, 0.f, 0, NllFused{});
    } else {
        const int b = blockIdx.x - g0;
        sgemm_direct_body<false, false, false, 0, 4>(smem, b % p1.gx, b / p1.gx, p1.A, p1.lda, p1.B, p1.ldb, nullptr, p1.C, p1.ldc, p1.M,
                                                     p1.N, p1.K, 1.f, 0.f, 0, NllFused{});
    }
}

// The rest of the forward: per row, the column tiles' statistics merge into max / first arg-max / logsumexp (online-softmax merge), the
// NLL at the positive, the correct count and the deterministic loss sum.  A workgroup takes 64 rows (eight threads per row: a single
// workgroup for all rows was bound by ONE CU's memory bandwidth, 9 - 15 us for 130 - 390 KB of statistics); the workgroups' partial sums
// meet in the last one to arrive (a handful of workgroups: write-through partials + an agent-scope counter that the last one resets, so
// the workspace needs no memset), which adds them in workgroup order — the loss sum is deterministic.
// (Finishing INSIDE the GEMM kernel by "last workgroup to arrive" per row tile was built and measured: with hundreds of workgroups the
// hand-off chain cost 12 - 18 us on top of a 9 - 24 us GEMM, profiles/r04_loss_kernel_trace_*.txt.)
__global__ __launch_bounds__(512) void nll_finish_kernel(const uint4* __restrict__ stat, const float* __restrict__ spos,
                                                         const int32_t* __restrict__ pos, int64_t n1, int64_t n2, int ntn,
                                                         float* __restrict__ row_loss, float* __restrict__ lse_out,
                                                         int32_t* __restrict__ correct, float* __restrict__ loss_sum,
                                                         double* __restrict__ part_sum, int32_t* __restrict__ part_ok,
                                                         int32_t* __restrict__ counter) {
    __shared__ double sh[512];
    __shared__ int ok_sh[8];
    __shared__ int last_flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = threadIdx.x & 7;
    const int64_t m = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 3);
    float mx = -INFINITY, z = 0.f;
    int am = 0x7fffffff;
    if (m < n1)
        for (int t = sub; t < ntn; t += 8) {
            const uint4 st = stat[m * ntn + t];
            const float sx = __uint_as_float(st.x), sz = __uint_as_float(st.y);
            const int a2 = (int)st.z;
            if (sx > mx || (sx == mx && a2 < am)) am = a2;
            const float nm = fmaxf(mx, sx);
            z = z * expf(mx - nm) + sz * expf(sx - nm);   // (exp(-inf - nm) = 0 for the empty start)
            mx = nm;
        }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(mx, o), z2 = __shfl_xor(z, o);
        const int a2 = __shfl_xor(am, o);
        if (m2 > mx || (m2 == mx && a2 < am)) am = a2;
        const float nm = fmaxf(mx, m2);
        z = (nm == -INFINITY) ? 0.f : z * expf(mx - nm) + z2 * expf(m2 - nm);
        mx = nm;
    }
    double acc = 0.0;
    int nok = 0;
    if (m < n1 && sub == 0) {
        const float l = mx + logf(z);
        const int32_t p = pos[m];
        // (a positive outside [0, n2) was matched by no tile: spos[m] is stale workspace — the row's loss is NaN, not garbage)
        const float rl = (p >= 0 && (int64_t)p < n2) ? l - spos[m] : NAN;
        lse_out[m] = l;
        row_loss[m] = rl;
        acc = (double)rl;
        nok = am == p;
    }
    sh[threadIdx.x] = acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nok += __shfl_xor(nok, o);
    if (lane == 0) ok_sh[wave] = nok;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 8; ++w) tot += ok_sh[w];
        // write-through partials, drained, then the arrival counter (agent scope)
        __hip_atomic_store(&part_sum[blockIdx.x], sh[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&part_ok[blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = old == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last_flag || threadIdx.x != 0) return;
    double total = 0.0;
    int tot = 0;
    for (unsigned b = 0; b < gridDim.x; ++b) {   // (workgroup order: deterministic)
        total += __hip_atomic_load(&part_sum[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot += __hip_atomic_load(&part_ok[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    loss_sum[0] = (float)total;
    correct[0] = tot;
    __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call on this stream
}

template <int EPI>
static int launch_sgemm_direct(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk, const float* B2,
                               float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha, float w2, int accumulate,
                               const NllFused& nll, hipStream_t st) {
    const bool ak = (sak == 1), bk = (sbk == 1);
    const int64_t lda = ak ? sam : sak, ldb = bk ? sbn : sbk;
    // both operands K-contiguous (the score GEMMs, K = 768): eight waves split K (measured 10-15 % shorter than four:
    // profiles/r04_loss_kernel_trace_ks8.txt); the others (K = the batch: 512 / 1536) are up to 2.5x SLOWER that way and keep four
#define LDOT_SGD(AK_, BK_, MIX_)                                                                                                 \
    do {                                                                                                                         \
        constexpr int KS_ = (AK_ && BK_) ? 8 : 4;                                                                                \
        hipLaunchKernelGGL((sgemm_direct_kernel<AK_, BK_, MIX_, EPI, KS_>), dim3((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32)), \
                           dim3(64 * KS_), 0, st, A, lda, B, ldb, B2, C, ldc, M, N, K, alpha, w2, accumulate, nll);              \
    } while (0)
    if (B2) {
        if (ak && bk) LDOT_SGD(true, true, true);
        else if (ak && !bk) LDOT_SGD(true, false, true);
        else if (!ak && bk) LDOT_SGD(false, true, true);
        else LDOT_SGD(false, false, true);
    } else {
        if (ak && bk) LDOT_SGD(true, true, false);
        else if (ak && !bk) LDOT_SGD(true, false, false);
        else if (!ak && bk) LDOT_SGD(false, true, false);
        else LDOT_SGD(false, false, false);
    }
#undef LDOT_SGD
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// small problems take the direct kernel (see there); 128^2 LDS tiles once they alone fill the chip
// (and while the contraction is short: a 32 x 32 tile re-reads its operands once per tile — 4096 x 768 x 4096 took 440 us on the direct kernel
// against 250 on the staged one, profiles/r05_loss_trace_4096.txt)
static bool sgemm_is_small(int64_t M, int64_t N, int64_t K = 0) { return (M + 127) / 128 * ((N + 127) / 128) < 256 && K <= 2048; }
// ... if their operands allow its unguarded loads: whole 32-deep chunks, 16-byte aligned rows of a K-contiguous operand, a K-major
// operand's 16 ld inside 32 bits
static bool sgemm_direct_ok(const float* P, int64_t s_row, int64_t s_k, int64_t K) {
    if (K <= 0 || K % 32) return false;
    if (s_k == 1) return s_row % 4 == 0 && ((uintptr_t)P & 15) == 0;
    return s_k < ((int64_t)1 << 26) && ((uintptr_t)P & 3) == 0;
}

// A(m,k) = A[m*sam + k*sak], B(n,k) = B[n*sbn + k*sbk]; exactly one stride of each operand is 1
static int launch_sgemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk, float* C,
                        int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha, int accumulate, hipStream_t st) {
    if (M <= 0 || N <= 0) return LDOT_OK;
    const bool ak = (sak == 1), bk = (sbk == 1);
    const int64_t lda = ak ? sam : sak, ldb = bk ? sbn : sbk;
    const int vec = ((lda % 4 == 0 && ((uintptr_t)A & 15) == 0) ? 1 : 0) | ((ldb % 4 == 0 && ((uintptr_t)B & 15) == 0) ? 2 : 0);
    if (sgemm_is_small(M, N, K) && sgemm_direct_ok(A, sam, sak, K) && sgemm_direct_ok(B, sbn, sbk, K))
        return launch_sgemm_direct<0>(A, sam, sak, B, sbn, sbk, nullptr, C, ldc, M, N, K, alpha, 0.f, accumulate, NllFused{}, st);
    // 128^2 tiles once they alone fill the chip, else 64^2 (more workgroups: these problems are latency bound)
    if ((M + 127) / 128 * ((N + 127) / 128) >= 256) {
        dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128));
        sgemm_dispatch<128>(ak, bk, grid, st, A, lda, B, ldb, C, ldc, M, N, K, alpha, accumulate, vec);
    } else {
        dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
        sgemm_dispatch<64>(ak, bk, grid, st, A, lda, B, ldb, C, ldc, M, N, K, alpha, accumulate, vec);
    }
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// C = (1-w) A.B^T + w A.B2^T  (B2 may be NULL / w == 0)
int launch_sgemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* B2, float w, float* C,
                    int64_t ldc, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    const bool mix = (B2 != nullptr && w != 0.f);
    const float a1 = mix ? (float)(1.0 - (double)w) : 1.f;
    if (M > 0 && N > 0 && sgemm_is_small(M, N, K) && sgemm_direct_ok(A, lda, 1, K) && sgemm_direct_ok(B, ldb, 1, K) &&
        (!mix || sgemm_direct_ok(B2, ldb, 1, K)))   // both products in one launch
        return launch_sgemm_direct<0>(A, lda, 1, B, ldb, 1, mix ? B2 : nullptr, C, ldc, M, N, K, a1, w, 0, NllFused{}, st);
    int rc = launch_sgemm(A, lda, 1, B, ldb, 1, C, ldc, M, N, K, a1, 0, st);
    if (rc || !mix) return rc;
    return launch_sgemm(A, lda, 1, B2, ldb, 1, C, ldc, M, N, K, w, 1, st);
}
// C[M x N] (+)= alpha * A[M x K] . B[K x N]
int launch_sgemm_nn(const float* A, int64_t lda, const float* B, int64_t ldb, float alpha, float* C, int64_t ldc,
                    int64_t M, int64_t N, int64_t K, int accumulate, hipStream_t st) {
    return launch_sgemm(A, lda, 1, B, 1, ldb, C, ldc, M, N, K, alpha, accumulate, st);
}
// C[M x N] (+)= alpha * A[K x M]^T . B[K x N]
int launch_sgemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float alpha, float* C, int64_t ldc,
                    int64_t M, int64_t N, int64_t K, int accumulate, hipStream_t st) {
    return launch_sgemm(A, 1, lda, B, 1, ldb, C, ldc, M, N, K, alpha, accumulate, st);
}

// one wave per row: max / first arg-max, logsumexp, NLL at the positive
__global__ __launch_bounds__(256) void nll_rows_kernel(const float* __restrict__ S, int64_t n1, int64_t n2,
                                                       const int32_t* __restrict__ pos, float* __restrict__ row_loss,
                                                       float* __restrict__ lse_out, int32_t* __restrict__ correct) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n1) return;
    const float* s = S + row * n2;
    float m = -INFINITY;
    int64_t am = 0x7fffffffffffffffll;
    for (int64_t j = lane; j < n2; j += 64) {
        const float v = s[j];
        if (v > m) {   // strict: keeps the first maximal column of this lane's stride
            m = v;
            am = j;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o);
        const int64_t a2 = __shfl_xor(am, o);
        if (m2 > m || (m2 == m && a2 < am)) {
            m = m2;
            am = a2;
        }
    }
    float z = 0.f;
    for (int64_t j = lane; j < n2; j += 64) z += expf(s[j] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o);
    const float lse = m + logf(z);
    if (lane == 0) {
        const int32_t p = pos[row];
        lse_out[row] = lse;
        row_loss[row] = lse - s[p];
        if (am == (int64_t)p) atomicAdd(correct, 1);
    }
}

// deterministic sum of row_loss (fp64 tree) -> loss_sum[0]
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ v, int64_t n, float* __restrict__ out) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) a += (double)v[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sh[0];
}

int launch_nll_rows(const float* scores, int64_t n1, int64_t n2, const int32_t* pos, float* row_loss, float* lse,
                    int32_t* correct, float* loss_sum, hipStream_t st) {
    LDOT_HIP_CHECK(hipMemsetAsync(correct, 0, sizeof(int32_t), st));
    if (n1 > 0) {
        hipLaunchKernelGGL(nll_rows_kernel, dim3((unsigned)((n1 + 3) / 4)), dim3(256), 0, st, scores, n1, n2, pos,
                           row_loss, lse, correct);
        LDOT_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, st, row_loss, n1, loss_sum);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

__global__ __launch_bounds__(256) void nll_dscores_kernel(const float* __restrict__ S, const float* __restrict__ lse,
                                                          const int32_t* __restrict__ pos,
                                                          const float* __restrict__ g_row,
                                                          const float* __restrict__ g_scores, int64_t n1, int64_t n2,
                                                          float* __restrict__ dS) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n1 * n2) return;
    const int64_t r = i / n2, c = i % n2;
    float v = g_row[r] * (expf(S[i] - lse[r]) - ((int64_t)pos[r] == c ? 1.f : 0.f));
    if (g_scores) v += g_scores[i];
    dS[i] = v;
}

int launch_nll_dscores(const float* scores, const float* lse, const int32_t* pos, const float* g_row,
                       const float* g_scores, int64_t n1, int64_t n2, float* ds, hipStream_t st) {
    const int64_t n = n1 * n2;
    if (n <= 0) return LDOT_OK;
    hipLaunchKernelGGL(nll_dscores_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, scores, lse, pos,
                       g_row, g_scores, n1, n2, ds);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// ---- the bidirectional train step (train_itm.py:195-222) in one pass --------------------------------------------------------------
// Both directions' rows are finished by one launch: blockIdx.y = direction (0: img -> txt, 1: txt -> img), the same merge as
// nll_finish_kernel per row; the last workgroup to arrive adds the partials of both directions in workgroup order and writes
//   out = {loss_txt (mean), loss_img (mean), loss_nce = 0.5 loss_txt + 0.5 loss_img, is_correct = (c_txt + c_img) / 2, c_txt, c_img}.
// blockIdx.y == 2 (optional): scores_avg = 0.5 S_txt + 0.5 S_img elementwise (train_itm.py:222).
struct NllDir {
    const uint4* stat;
    const float* spos;
    float *row_loss, *lse;
    int ntn;
};
struct NllFinish2 {
    NllDir dir[2];
    const int32_t* pos;
    int64_t n1, n2;
    float* out;          // [6]
    double* part_sum;    // [2][gridDim.x]
    int32_t* part_ok;    // [2][gridDim.x]
    int32_t* counter;
    const float *s_txt, *s_img;
    float* s_avg;        // (NULL: not wanted)
};

__global__ __launch_bounds__(512) void nll_finish_bidir_kernel(NllFinish2 a) {
    __shared__ double sh[512];
    __shared__ int ok_sh[8];
    __shared__ int last_flag;
    if (blockIdx.y == 2) {
        const int64_t tot = a.n1 * a.n2;
        for (int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 512)
            a.s_avg[i] = __fadd_rn(__fmul_rn(a.s_txt[i], 0.5f), __fmul_rn(a.s_img[i], 0.5f));
        return;
    }
    const NllDir d = a.dir[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = threadIdx.x & 7;
    const int64_t m = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 3);
    float mx = -INFINITY, z = 0.f;
    int am = 0x7fffffff;
    if (m < a.n1)
        for (int t = sub; t < d.ntn; t += 8) {
            const uint4 st = d.stat[m * d.ntn + t];
            const float sx = __uint_as_float(st.x), sz = __uint_as_float(st.y);
            const int a2 = (int)st.z;
            if (sx > mx || (sx == mx && a2 < am)) am = a2;
            const float nm = fmaxf(mx, sx);
            z = z * expf(mx - nm) + sz * expf(sx - nm);
            mx = nm;
        }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(mx, o), z2 = __shfl_xor(z, o);
        const int a2 = __shfl_xor(am, o);
        if (m2 > mx || (m2 == mx && a2 < am)) am = a2;
        const float nm = fmaxf(mx, m2);
        z = (nm == -INFINITY) ? 0.f : z * expf(mx - nm) + z2 * expf(m2 - nm);
        mx = nm;
    }
    double acc = 0.0;
    int nok = 0;
    if (m < a.n1 && sub == 0) {
        const float l = mx + logf(z);
        const int32_t p = a.pos[m];
        const float rl = (p >= 0 && (int64_t)p < a.n2) ? l - d.spos[m] : NAN;
        d.lse[m] = l;
        d.row_loss[m] = rl;
        acc = (double)rl;
        nok = am == p;
    }
    sh[threadIdx.x] = acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nok += __shfl_xor(nok, o);
    if (lane == 0) ok_sh[wave] = nok;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 8; ++w) tot += ok_sh[w];
        const unsigned slot = blockIdx.y * gridDim.x + blockIdx.x;
        __hip_atomic_store(&a.part_sum[slot], sh[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.part_ok[slot], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int old = __hip_atomic_fetch_add(a.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = old == 2 * (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last_flag || threadIdx.x != 0) return;
    double total[2] = {0.0, 0.0};
    int tot[2] = {0, 0};
    for (unsigned dd = 0; dd < 2; ++dd)
        for (unsigned b = 0; b < gridDim.x; ++b) {   // (workgroup order: deterministic)
            total[dd] += __hip_atomic_load(&a.part_sum[dd * gridDim.x + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tot[dd] += __hip_atomic_load(&a.part_ok[dd * gridDim.x + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    // mean like F.nll_loss(reduction='mean'): the fp32 sum divided by the row count; then 0.5 a + 0.5 b as train_itm.py:212 writes it
    const float l0 = (float)total[0] / (float)a.n1, l1 = (float)total[1] / (float)a.n1;
    a.out[0] = l0;
    a.out[1] = l1;
    a.out[2] = __fadd_rn(__fmul_rn(0.5f, l0), __fmul_rn(0.5f, l1));
    a.out[3] = (float)(tot[0] + tot[1]) * 0.5f;
    a.out[4] = (float)tot[0];
    a.out[5] = (float)tot[1];
    __hip_atomic_store(a.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// dS of the bidirectional step from S_txt alone (S_img[n][m] = S_txt[m][n] over the shared block):
//   dS1[m][n]  = c1 (softmax_row(S_txt)[m][n] - [pos[m] == n]) + 0.5 G[m][n]
//              + (n < bs:  c2 (exp(S_txt[m][n] - lse_img[n]) - [pos[n] == m]) + 0.5 G[n][m])        (the transposed direction folded in)
//   dS2b[j][c] = c2 (exp(S_img[j][bs + c] - lse_img[j]) - [pos[j] == bs + c]) + 0.5 G[j][bs + c]      (hard-negative columns of S_img)
// c1 = (0.5 g_nce + g_txt) / bs, c2 = (0.5 g_nce + g_img) / bs from DEVICE scalars (no host round trip); G = gradient w.r.t. scores_avg.
struct NllBwd2 {
    const float *s_txt, *s_img, *lse_txt, *lse_img, *g_nce, *g_txt, *g_img, *g_avg;
    const int32_t* pos;
    int64_t bs, n;
    float *ds1, *ds2b;
};
__global__ __launch_bounds__(256) void nll_dscores_bidir_kernel(NllBwd2 a) {
    const float gn = a.g_nce ? a.g_nce[0] : 0.f;
    const float inv = 1.f / (float)a.bs;
    const float c1 = (0.5f * gn + (a.g_txt ? a.g_txt[0] : 0.f)) * inv, c2 = (0.5f * gn + (a.g_img ? a.g_img[0] : 0.f)) * inv;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tot1 = a.bs * a.n, nb = a.n - a.bs;
    if (i < tot1) {
        const int64_t m = i / a.n, n = i % a.n;
        const float s = a.s_txt[i];
        float v = c1 * (expf(s - a.lse_txt[m]) - ((int64_t)a.pos[m] == n ? 1.f : 0.f));
        if (a.g_avg) v += 0.5f * a.g_avg[i];
        if (n < a.bs) {
            v += c2 * (expf(s - a.lse_img[n]) - ((int64_t)a.pos[n] == m ? 1.f : 0.f));
            if (a.g_avg) v += 0.5f * a.g_avg[n * a.n + m];
        }
        a.ds1[i] = v;
    } else if (i < tot1 + a.bs * nb) {
        const int64_t r = i - tot1, j = r / nb, c = r % nb + a.bs;
        float v = c2 * (expf(a.s_img[j * a.n + c] - a.lse_img[j]) - ((int64_t)a.pos[j] == c ? 1.f : 0.f));
        if (a.g_avg) v += 0.5f * a.g_avg[j * a.n + c];
        a.ds2b[r] = v;
    }
}

__global__ void nll_combine_kernel(const float* loss_sum, const int32_t* correct, int64_t n1, float* out) {
    const float l0 = loss_sum[0] / (float)n1, l1 = loss_sum[1] / (float)n1;
    out[0] = l0;
    out[1] = l1;
    out[2] = __fadd_rn(__fmul_rn(0.5f, l0), __fmul_rn(0.5f, l1));
    out[3] = (float)(correct[0] + correct[1]) * 0.5f;
    out[4] = (float)correct[0];
    out[5] = (float)correct[1];
}
__global__ __launch_bounds__(256) void scores_avg_kernel(const float* a, const float* b, int64_t n, float* o) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        o[i] = __fadd_rn(__fmul_rn(a[i], 0.5f), __fmul_rn(b[i], 0.5f));
}

}  // namespace ldot

using namespace ldot;

extern "C" {

int ldot_dot_product_scores(const float* q, const float* ctx, int64_t n1, int64_t n2, int64_t d, float* out,
                            void* stream) {
    LDOT_REQUIRE(q && ctx && out, LDOT_EINVAL, "NULL buffer");
    LDOT_REQUIRE(n1 >= 0 && n2 >= 0 && d > 0, LDOT_EINVAL, "bad shape");
    return launch_sgemm_nt(q, d, ctx, d, nullptr, 0.f, out, n2, n1, n2, d, (hipStream_t)stream);
}

// Workspace of the fused forward (per-tile statistics), one per (device, stream) that ever ran it: calls on
// different streams may overlap on the GPU, calls on one stream cannot.  Grow-only; a few hundred KB at the config-5 shapes.
namespace {
struct LossWs {
    int device;
    hipStream_t stream;
    void* p;
    size_t bytes;
    int users;   // launchers between loss_workspace() and the end of their enqueue (LossLease): never evicted while > 0
};
std::mutex g_loss_mu;
std::vector<LossWs> g_loss_ws;

// held by a launcher while it enqueues the kernels that use the workspace: another host thread that needs a slot must not free this one
// before they are on the stream (afterwards the eviction's hipDeviceSynchronize covers them)
struct LossLease {
    int device = -1;
    hipStream_t stream = nullptr;
    bool held = false;
    ~LossLease() {
        if (!held) return;
        std::lock_guard<std::mutex> lock(g_loss_mu);
        for (LossWs& w : g_loss_ws)
            if (w.device == device && w.stream == stream && w.users > 0) --w.users;
    }
};

int loss_workspace(hipStream_t st, size_t need, char** out, LossLease* lease) {
    int dev = 0;
    LDOT_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_loss_mu);
    LossWs* ws = nullptr;
    for (LossWs& w : g_loss_ws)
        if (w.device == dev && w.stream == st) ws = &w;
    if (!ws) {
        // at most kLossWsPerDevice cached workspaces per device: a caller that creates streams as it goes evicts the oldest one (the
        // stream it belonged to may be gone: the device is drained before its buffer is freed)
        constexpr size_t kLossWsPerDevice = 8;
        size_t mine = 0, oldest = g_loss_ws.size();
        for (size_t i = 0; i < g_loss_ws.size(); ++i)
            if (g_loss_ws[i].device == dev) {
                if (oldest == g_loss_ws.size() && g_loss_ws[i].users == 0) oldest = i;   // (a workspace another thread is launching on stays)
                ++mine;
            }
        if (mine >= kLossWsPerDevice && oldest < g_loss_ws.size()) {
            LDOT_HIP_CHECK(hipDeviceSynchronize());
            if (g_loss_ws[oldest].p) (void)hipFree(g_loss_ws[oldest].p);
            g_loss_ws.erase(g_loss_ws.begin() + (long)oldest);
        }
        g_loss_ws.push_back(LossWs{dev, st, nullptr, 0, 0});
        ws = &g_loss_ws.back();
    }
    if (ws->bytes < need) {
        if (ws->p) {
            LDOT_HIP_CHECK(hipStreamSynchronize(st));   // (the previous call on this stream may still be reading it)
            (void)hipFree(ws->p);
            ws->p = nullptr;
            ws->bytes = 0;
        }
        LDOT_HIP_CHECK(hipMalloc(&ws->p, need));
        ws->bytes = need;
        LDOT_HIP_CHECK(hipMemsetAsync(ws->p, 0, 256, st));   // the finish kernel's arrival counter: zero now, reset by its last workgroup
    }
    *out = (char*)ws->p;
    ++ws->users;
    lease->device = dev;
    lease->stream = st;
    lease->held = true;
    return LDOT_OK;
}
}  // namespace

// forward of the loss in two launches: the score tiles with the softmax statistics of their rows in the epilogue (max, first arg-max,
// sum of exponentials, the positive's score: sgemm_direct_kernel, EPI = 1 — the scores are never re-read), and one workgroup that merges
// the tiles' statistics into logsumexp / NLL / correct count / loss sum.  SURVEY 2b K4's "fused score + logsumexp + NLL + argmax"
// (dvl/models/bi_encoder.py:624,649-655)
static int launch_nll_fwd_fused(const float* q, const float* ctx, const float* cap, float w, const int32_t* pos, int64_t n1, int64_t n2,
                                int64_t d, float* scores, float* row_loss, float* lse, int32_t* correct, float* loss_sum, hipStream_t st) {
    const bool mix = (cap != nullptr && w != 0.f);
    const float a1 = mix ? (float)(1.0 - (double)w) : 1.f;
    const int64_t ntn = (n2 + 31) / 32;
    const int64_t nfin = (n1 + 63) / 64;   // workgroups of the finish kernel
    // workspace: [arrival counter (kept zero between calls) | per-tile statistics | positives' scores | finish partials]
    const size_t b_cnt = 256, b_st = (size_t)round_up(n1 * ntn * 16, 256), b_spos = (size_t)round_up(n1 * 4, 256),
                 b_ps = (size_t)round_up(nfin * 8, 256), b_po = (size_t)round_up(nfin * 4, 256);
    char* ws = nullptr;
    LossLease lease;
    int rc = loss_workspace(st, b_cnt + b_st + b_spos + b_ps + b_po, &ws, &lease);
    if (rc) return rc;
    NllFused nll{};
    nll.pos = pos;
    nll.stat = (uint4*)(ws + b_cnt);
    nll.spos = (float*)(ws + b_cnt + b_st);
    nll.ldstat = (int)ntn;
    if ((rc = launch_sgemm_direct<1>(q, d, 1, ctx, d, 1, mix ? cap : nullptr, scores, n2, n1, n2, d, a1, w, 0, nll, st))) return rc;
    hipLaunchKernelGGL(nll_finish_kernel, dim3((unsigned)nfin), dim3(512), 0, st, nll.stat, nll.spos, pos, n1, n2, (int)ntn, row_loss, lse, correct,
                       loss_sum, (double*)(ws + b_cnt + b_st + b_spos), (int32_t*)(ws + b_cnt + b_st + b_spos + b_ps), (int32_t*)ws);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int ldot_inbatch_nll_fwd(const float* q, const float* ctx, const float* cap, float w, const int32_t* pos, int64_t n1,
                         int64_t n2, int64_t d, float* scores, float* row_loss, float* lse, int32_t* correct,
                         float* loss_sum, void* stream) {
    LDOT_REQUIRE(q && ctx && pos && scores && row_loss && lse && correct && loss_sum, LDOT_EINVAL, "NULL buffer");
    LDOT_REQUIRE(n1 > 0 && n2 > 0 && d > 0, LDOT_EINVAL, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (sgemm_is_small(n1, n2) && sgemm_direct_ok(q, d, 1, d) && sgemm_direct_ok(ctx, d, 1, d) && (!cap || sgemm_direct_ok(cap, d, 1, d)))
        return launch_nll_fwd_fused(q, ctx, cap, w, pos, n1, n2, d, scores, row_loss, lse, correct, loss_sum, st);
    int rc = launch_sgemm_nt(q, d, ctx, d, cap, w, scores, n2, n1, n2, d, st);
    if (rc) return rc;
    return launch_nll_rows(scores, n1, n2, pos, row_loss, lse, correct, loss_sum, st);
}

// ---- the bidirectional train step: ONE score GEMM for the shared block, both directions finished together ---------------------------
int ldot_inbatch_nll_bidir_fwd(const float* img, const float* txt, const int32_t* pos, int64_t bs, int64_t n, int64_t d, float* s_txt,
                               float* s_img, float* s_avg, float* lse, float* row_loss, float* out, void* stream) {
    LDOT_REQUIRE(img && txt && pos && s_txt && s_img && lse && row_loss && out, LDOT_EINVAL, "NULL buffer");
    LDOT_REQUIRE(bs > 0 && n >= bs && d > 0, LDOT_EINVAL, "bad shape (need 0 < bs <= n)");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = n - bs;
    if (!(sgemm_is_small(bs, n) && sgemm_direct_ok(img, d, 1, d) && sgemm_direct_ok(txt, d, 1, d))) {
        // shapes beyond the direct kernel (or ragged K): the two directions one after the other on the staged kernels
        const size_t b_small = 256;
        char* ws = nullptr;
        LossLease lease;
        int rc = loss_workspace(st, 256 + b_small, &ws, &lease);
        if (rc) return rc;
        float* lsum = (float*)(ws + 256);
        int32_t* corr = (int32_t*)(ws + 256 + 64);
        if ((rc = launch_sgemm_nt(img, d, txt, d, nullptr, 0.f, s_txt, n, bs, n, d, st))) return rc;
        if ((rc = launch_nll_rows(s_txt, bs, n, pos, row_loss, lse, corr, lsum, st))) return rc;
        if ((rc = launch_sgemm_nt(txt, d, img, d, nullptr, 0.f, s_img, n, bs, n, d, st))) return rc;
        if ((rc = launch_nll_rows(s_img, bs, n, pos, row_loss + bs, lse + bs, corr + 1, lsum + 1, st))) return rc;
        hipLaunchKernelGGL(nll_combine_kernel, dim3(1), dim3(1), 0, st, lsum, corr, bs, out);
        if (s_avg) hipLaunchKernelGGL(scores_avg_kernel, dim3((unsigned)std::min<int64_t>((bs * n + 255) / 256, 2048)), dim3(256), 0, st, s_txt, s_img, bs * n, s_avg);
        LDOT_HIP_CHECK(hipGetLastError());
        return LDOT_OK;
    }
    const int64_t ntn1 = (n + 31) / 32, ntm = (bs + 31) / 32, ntn2b = (nb + 31) / 32, nt2 = ntm + ntn2b;
    const int64_t nfin = (bs + 63) / 64;
    const size_t b_cnt = 256, b_st1 = (size_t)round_up(bs * ntn1 * 16, 256), b_st2 = (size_t)round_up(bs * nt2 * 16, 256),
                 b_spos = (size_t)round_up(bs * 4, 256), b_ps = (size_t)round_up(2 * nfin * 8, 256), b_po = (size_t)round_up(2 * nfin * 4, 256);
    char* ws = nullptr;
    LossLease lease;
    int rc = loss_workspace(st, b_cnt + b_st1 + b_st2 + 2 * b_spos + b_ps + b_po, &ws, &lease);
    if (rc) return rc;
    uint4* stat1 = (uint4*)(ws + b_cnt);
    uint4* stat2 = (uint4*)(ws + b_cnt + b_st1);
    float* spos1 = (float*)(ws + b_cnt + b_st1 + b_st2);
    float* spos2 = (float*)(ws + b_cnt + b_st1 + b_st2 + b_spos);
    NllFused f{};
    f.pos = pos;
    f.stat = stat1;
    f.ldstat = (int)ntn1;
    f.spos = spos1;
    f.Ct = s_img;
    f.ldct = n;
    f.nshared = bs;
    f.stat_t = stat2;
    f.ldstat_t = (int)nt2;
    f.spos_t = spos2;
    // S_txt = img[:bs] . txt^T with the row statistics, its transpose over the first bs columns (= S_img's shared block) and the column statistics
    hipLaunchKernelGGL((sgemm_direct_kernel<true, true, false, 2, 8>), dim3((unsigned)ntn1, (unsigned)ntm), dim3(512), 0, st, img, d, txt, d,
                       (const float*)nullptr, s_txt, n, bs, n, d, 1.f, 0.f, 0, f);
    if (nb > 0) {   // hard-negative columns of S_img: txt[:bs] . img[bs:]^T
        NllFused g{};
        g.pos = pos;
        g.stat = stat2;
        g.ldstat = (int)nt2;
        g.stat_col0 = (int)ntm;
        g.col_off = bs;
        g.spos = spos2;
        hipLaunchKernelGGL((sgemm_direct_kernel<true, true, false, 1, 8>), dim3((unsigned)ntn2b, (unsigned)ntm), dim3(512), 0, st, txt, d,
                           img + bs * d, d, (const float*)nullptr, s_img + bs, n, bs, nb, d, 1.f, 0.f, 0, g);
    }
    NllFinish2 a{};
    a.dir[0] = NllDir{stat1, spos1, row_loss, lse, (int)ntn1};
    a.dir[1] = NllDir{stat2, spos2, row_loss + bs, lse + bs, (int)nt2};
    a.pos = pos;
    a.n1 = bs;
    a.n2 = n;
    a.out = out;
    a.part_sum = (double*)(ws + b_cnt + b_st1 + b_st2 + 2 * b_spos);
    a.part_ok = (int32_t*)(ws + b_cnt + b_st1 + b_st2 + 2 * b_spos + b_ps);
    a.counter = (int32_t*)ws;
    a.s_txt = s_txt;
    a.s_img = s_img;
    a.s_avg = s_avg;
    hipLaunchKernelGGL(nll_finish_bidir_kernel, dim3((unsigned)nfin, s_avg ? 3 : 2), dim3(512), 0, st, a);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int ldot_inbatch_nll_bidir_bwd(const float* img, const float* txt, const int32_t* pos, int64_t bs, int64_t n, int64_t d,
                               const float* s_txt, const float* s_img, const float* lse, const float* g_nce, const float* g_txt,
                               const float* g_img, const float* g_avg, float* ds_work, float* dimg, float* dtxt, void* stream) {
    LDOT_REQUIRE(img && txt && pos && s_txt && s_img && lse && ds_work, LDOT_EINVAL, "NULL buffer");
    LDOT_REQUIRE(bs > 0 && n >= bs && d > 0, LDOT_EINVAL, "bad shape (need 0 < bs <= n)");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = n - bs;
    NllBwd2 a{};
    a.s_txt = s_txt;
    a.s_img = s_img;
    a.lse_txt = lse;
    a.lse_img = lse + bs;
    a.g_nce = g_nce;
    a.g_txt = g_txt;
    a.g_img = g_img;
    a.g_avg = g_avg;
    a.pos = pos;
    a.bs = bs;
    a.n = n;
    a.ds1 = ds_work;
    a.ds2b = ds_work + bs * n;
    const int64_t tot = bs * n + bs * nb;
    hipLaunchKernelGGL(nll_dscores_bidir_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, a);
    LDOT_HIP_CHECK(hipGetLastError());
    int rc;
    // S_txt = img[:bs] . txt^T  ->  dimg[:bs] = dS1 . txt,  dtxt = dS1^T . img[:bs]
    if (dimg && dtxt && sgemm_is_small(bs, d, n) && sgemm_is_small(n, d, bs) && sgemm_direct_ok(a.ds1, n, 1, n) && sgemm_direct_ok(txt, 1, d, n) &&
        sgemm_direct_ok(a.ds1, 1, n, bs) && sgemm_direct_ok(img, 1, d, bs)) {
        SgPairSide p0{a.ds1, txt, dimg, n, d, d, bs, d, n, (int)((d + 31) / 32), (int)((bs + 31) / 32)};
        SgPairSide p1{a.ds1, img, dtxt, n, d, d, n, d, bs, (int)((d + 31) / 32), (int)((n + 31) / 32)};
        hipLaunchKernelGGL(sgemm_direct_pair_kernel, dim3((unsigned)(p0.gx * p0.gy + p1.gx * p1.gy)), dim3(256), 0, st, p0, p1);
        LDOT_HIP_CHECK(hipGetLastError());
    } else {
        if (dimg && (rc = launch_sgemm_nn(a.ds1, n, txt, d, 1.f, dimg, d, bs, d, n, 0, st))) return rc;
        if (dtxt && (rc = launch_sgemm_tn(a.ds1, n, img, d, 1.f, dtxt, d, n, d, bs, 0, st))) return rc;
    }
    if (nb > 0) {   // S_img[:, bs:] = txt[:bs] . img[bs:]^T  ->  dtxt[:bs] += dS2b . img[bs:],  dimg[bs:] = dS2b^T . txt[:bs]
        if (dtxt && (rc = launch_sgemm_nn(a.ds2b, nb, img + bs * d, d, 1.f, dtxt, d, bs, d, nb, 1, st))) return rc;
        if (dimg && (rc = launch_sgemm_tn(a.ds2b, nb, txt, d, 1.f, dimg + bs * d, d, nb, d, bs, 0, st))) return rc;
    }
    return LDOT_OK;
}

int ldot_inbatch_nll_bwd(const float* q, const float* ctx, const float* cap, float w, const int32_t* pos, int64_t n1,
                         int64_t n2, int64_t d, const float* scores, const float* lse, const float* g_row,
                         const float* g_scores, float* ds_work, float* dq, float* dctx, float* dcap, void* stream) {
    LDOT_REQUIRE(q && ctx && pos && scores && lse && g_row && ds_work, LDOT_EINVAL, "NULL buffer");
    LDOT_REQUIRE(n1 > 0 && n2 > 0 && d > 0, LDOT_EINVAL, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    const bool mix = (cap != nullptr && w != 0.f);
    const float a1 = mix ? (float)(1.0 - (double)w) : 1.f;
    int rc = launch_nll_dscores(scores, lse, pos, g_row, g_scores, n1, n2, ds_work, st);
    if (rc) return rc;
    if (dq) {
        if ((rc = launch_sgemm_nn(ds_work, n2, ctx, d, a1, dq, d, n1, d, n2, 0, st))) return rc;
        if (mix && (rc = launch_sgemm_nn(ds_work, n2, cap, d, w, dq, d, n1, d, n2, 1, st))) return rc;
    }
    if (dctx && (rc = launch_sgemm_tn(ds_work, n2, q, d, a1, dctx, d, n2, d, n1, 0, st))) return rc;
    if (dcap && mix && (rc = launch_sgemm_tn(ds_work, n2, q, d, w, dcap, d, n2, d, n1, 0, st))) return rc;
    return LDOT_OK;
}

}  // extern "C"
