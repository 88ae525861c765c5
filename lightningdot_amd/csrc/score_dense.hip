// Dense score chunk:  S[q][n] = Q16[q,:] . X16[row0 + n,:]   (fp32 out, bf16 MFMA in)
// Used for small indexes (Flickr/COCO sized), for the warm-up chunk of the fused search and as the
// always-correct fallback.  Reference call site replaced: faiss IndexFlatIP.search's sgemm
// (dvl/indexer/faiss_indexers.py:83).
#include "gemm_tile.h"
#include "kernels.h"

namespace ldot {

// A = queries (M side), B = index rows (N side): a lane's 32 "col" lanes are consecutive index rows, so every
// accumulator register stores a coalesced 128-byte run of S[q][n..n+32).
__global__ __launch_bounds__(kGemmThreads, 2) void score_dense_kernel(
    const char* __restrict__ Q16, int64_t ldq_b, int tiles_m, const char* __restrict__ X16, int64_t ldx_b,
    int64_t xrow0, int tiles_n, int nk, float* __restrict__ S, int64_t lds_elems, int64_t m_valid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // supertile order: 8 query tiles x 4 row tiles per group of 32 consecutive logical ids
    const int sm = (tiles_m + 7) / 8;
    const int nwg = gridDim.x;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int sup = wg >> 5, in = wg & 31;
    const int tm = (sup % sm) * 8 + (in & 7);
    const int tn = (sup / sm) * 4 + (in >> 3);
    if (tm >= tiles_m || tn >= tiles_n) return;

    TileCtx c;
    tile_ctx_init(c);
    f32x16 acc[4][2];
    gemm_tile(c, Q16, ldq_b, (int64_t)tm * kBM, X16, ldx_b, xrow0 + (int64_t)tn * kBN, nk, smem, acc);

    const int64_t m_base = (int64_t)tm * kBM + c.wm * 128 + 4 * (c.lane >> 5);
    const int64_t n_base = (int64_t)tn * kBN + c.wn * 64 + (c.lane & 31);
#pragma unroll
    for (int mr = 0; mr < 4; ++mr)
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m_base + mr * 32 + (r & 3) + 8 * (r >> 2);
                if (m < m_valid) S[m * lds_elems + n_base + nr * 32] = acc[mr][nr][r];   // pad queries are not stored
            }
}

int launch_score_dense(const void* q16, int64_t ldq_elems, int64_t nq_pad, const void* x16, int64_t ldx_elems,
                       int64_t xrow0, int64_t nrows_pad, int dpad, float* S, int64_t lds_elems, int64_t nq_valid,
                       hipStream_t st) {
    const int tiles_m = (int)(nq_pad / kBM), tiles_n = (int)(nrows_pad / kBN);
    const int sm = (tiles_m + 7) / 8, sn = (tiles_n + 3) / 4;
    const int nwg = sm * sn * 32;
    static bool attr_set = false;
    if (!attr_set) {
        LDOT_HIP_CHECK(hipFuncSetAttribute((const void*)score_dense_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLdsBytes));
        attr_set = true;
    }
    hipLaunchKernelGGL(score_dense_kernel, dim3(nwg), dim3(kGemmThreads), kGemmLdsBytes, st, (const char*)q16,
                       ldq_elems * 2, tiles_m, (const char*)x16, ldx_elems * 2, xrow0, tiles_n, dpad / kBK, S,
                       lds_elems, nq_valid);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

}  // namespace ldot
