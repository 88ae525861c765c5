// In-batch contrastive loss kernels (fp32-exact):
//   scores = (1-w) q.ctx^T + w q.cap^T ; log-softmax rows ; NLL at the positive ; arg-max == positive count
// Reference: dvl/models/bi_encoder.py:54-68 (dot_product_scores), :615-656 (BiEncoderNllLoss.calc).
// The contraction uses v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate, bit-for-bit an fmaf chain), so the
// loss and its gradients match an fp32 reference to rounding; the problem (<= 512 x 1536 x 768) is latency
// bound, not MFMA bound, so the 1/16-rate fp32 matrix path costs nothing that matters.
#include <math.h>

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
    __shared__ __attribute__((aligned(16))) float As[BK][LD];
    __shared__ __attribute__((aligned(16))) float Bs[BK][LD];
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
    for (int64_t k0 = 0; k0 < K; k0 += BK) {
        oa.stash(As);
        ob.stash(Bs);
        __syncthreads();
        if (k0 + BK < K) {
            oa.fetch(A, lda, m0, M, k0 + BK, K, vec & 1);
            ob.fetch(B, ldb, n0, N, k0 + BK, K, (vec >> 1) & 1);
        }
#pragma unroll
        for (int ks = 0; ks < BK; ks += 2) {
            float a[WT], b[WT];
#pragma unroll
            for (int i = 0; i < WT; ++i) a[i] = As[ks + (lane >> 5)][(wm * WT + i) * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < WT; ++j) b[j] = Bs[ks + (lane >> 5)][(wn * WT + j) * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
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

// A(m,k) = A[m*sam + k*sak], B(n,k) = B[n*sbn + k*sbk]; exactly one stride of each operand is 1
static int launch_sgemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk, float* C,
                        int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha, int accumulate, hipStream_t st) {
    if (M <= 0 || N <= 0) return LDOT_OK;
    const bool ak = (sak == 1), bk = (sbk == 1);
    const int64_t lda = ak ? sam : sak, ldb = bk ? sbn : sbk;
    const int vec = ((lda % 4 == 0 && ((uintptr_t)A & 15) == 0) ? 1 : 0) | ((ldb % 4 == 0 && ((uintptr_t)B & 15) == 0) ? 2 : 0);
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

}  // namespace ldot

using namespace ldot;

extern "C" {

int ldot_dot_product_scores(const float* q, const float* ctx, int64_t n1, int64_t n2, int64_t d, float* out,
                            void* stream) {
    LDOT_REQUIRE(q && ctx && out, LDOT_EINVAL, "NULL buffer");
    LDOT_REQUIRE(n1 >= 0 && n2 >= 0 && d > 0, LDOT_EINVAL, "bad shape");
    return launch_sgemm_nt(q, d, ctx, d, nullptr, 0.f, out, n2, n1, n2, d, (hipStream_t)stream);
}

int ldot_inbatch_nll_fwd(const float* q, const float* ctx, const float* cap, float w, const int32_t* pos, int64_t n1,
                         int64_t n2, int64_t d, float* scores, float* row_loss, float* lse, int32_t* correct,
                         float* loss_sum, void* stream) {
    LDOT_REQUIRE(q && ctx && pos && scores && row_loss && lse && correct && loss_sum, LDOT_EINVAL, "NULL buffer");
    LDOT_REQUIRE(n1 > 0 && n2 > 0 && d > 0, LDOT_EINVAL, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_sgemm_nt(q, d, ctx, d, cap, w, scores, n2, n1, n2, d, st);
    if (rc) return rc;
    return launch_nll_rows(scores, n1, n2, pos, row_loss, lse, correct, loss_sum, st);
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
