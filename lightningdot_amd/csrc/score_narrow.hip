// Narrow score scan:  S[q][n] = Q16[q,:] . X16[row0 + n,:]  for at most 64 queries — the single-query serving shape
// (dvl/utils.py:204-211 retrieve_query: one text query against the whole image index; faiss IndexFlatIP.search with nq = 1,
// dvl/indexer/faiss_indexers.py:83).
//
// With a handful of queries the search is one pass over the index at HBM speed: 2 flop per byte streamed, the matrix pipe is ~5 % busy.
// The ring engine of score_dense / score_filter (384 x 256 tiles through LDS) is the wrong tool — a 256-query tile is 94 % padding
// and its workgroup count is bounded by the row tiles.  Here every WAVE owns 16-row groups of the blocked bf16 shadow: a group's
// nslab 1-KiB blocks are CONTIGUOUS in memory, a block is exactly one MFMA 16x16x32 A operand in register order (lane l: row l & 15,
// k (l >> 4) * 8 .. + 8 = byte (l & 15) * 64 + (l >> 4) * 16 of the block), so the index goes global -> VGPR -> MFMA with one
// coalesced 1-KiB global_load_dwordx4 per block and no LDS staging.  The query block (B operand, same layout) is read from LDS, where
// the workgroup put it once.  UNR blocks are in flight per wave while the previous UNR are multiplied (register double buffer);
// with ~5 waves per SIMD that is ~300 KiB in flight per CU.
//
// C/D layout of 16x16x32: lane l holds rows (l >> 4) * 4 .. + 4 of column l & 15  ->  query l & 15 stores a float4 of four
// consecutive index rows.
//
// Besides the scores the kernel leaves RUN MAXIMA behind (one per query and run of 16 << run_shift rows, as order-preserving keys
// raised with atomicMax) — what the selection of select_narrow.hip takes its threshold from.
#include "kernels.h"
#include "ldot_common.h"

namespace ldot {

// UNR blocks per chunk, PF chunk buffers per wave (PF - 1 chunks in flight while one is multiplied), QG groups of 16 queries (1, 2 or
// 4), THREADS per workgroup
template <int UNR, int QG, int THREADS, int PF>
__global__ __launch_bounds__(THREADS) void score_narrow_kernel(const char* __restrict__ Q16b, const char* __restrict__ X16b,
                                                               int nslab, int64_t g0, int64_t ngroups, int per, int64_t nrows,
                                                               float* __restrict__ S, int64_t lds_elems, int nq,
                                                               uint32_t* __restrict__ M, int64_t ldm, int run_shift, int tiled,
                                                               const float* __restrict__ Qf32, int64_t ldqf, int d) {
    // QG * nslab KiB: query blocks 0 .. QG-1 of the blocked query shadow (block (g, s) at (g * nslab + s) KiB, like the source)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    if (QG == 1 && Qf32 != nullptr) {
        // (uniform) the caller's fp32 query rows, not yet staged: every workgroup builds the one blocked bf16 query block itself
        // (convert.hip's layout and rounding: element (r, c) at block c / 32, row r, column c % 32; rows >= nq and columns >= d are zero)
        // — a search of <= 16 queries then needs no conversion kernel in front of the scan
        for (int i = tid; i < nslab * 64; i += THREADS) ((uint4*)smem)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        for (int r = 0; r < nq; ++r)
            for (int c = tid * 4; c < d; c += THREADS * 4) {   // (d is a multiple of 32, the rows are 16-byte aligned: caller's contract)
                const float4 v = *(const float4*)(Qf32 + (int64_t)r * ldqf + c);
                uint16_t* o = (uint16_t*)smem + (c >> 5) * 512 + r * 32 + (c & 31);
                *(uint2*)o = make_uint2((uint32_t)f32_to_bf16_bits(v.x) | ((uint32_t)f32_to_bf16_bits(v.y) << 16),
                                        (uint32_t)f32_to_bf16_bits(v.z) | ((uint32_t)f32_to_bf16_bits(v.w) << 16));
            }
    } else {
        for (int i = tid; i < QG * nslab * 64; i += THREADS) ((uint4*)smem)[i] = ((const uint4*)Q16b)[i];
    }
    __syncthreads();
    const int lane = tid & 63;
    // a wave owns `per` CONSECUTIVE groups: one contiguous stream of per * nslab KiB, and a run maximum is raised once per run the
    // stream crosses instead of once per group
    const int64_t gbeg = ((int64_t)blockIdx.x * (THREADS / 64) + (tid >> 6)) * per;
    const int64_t gend = gbeg + per < ngroups ? gbeg + per : ngroups;
    if (gbeg >= gend) return;
    const int lo = (lane & 15) * 64 + (lane >> 4) * 16;
    const int nc = nslab / UNR;                               // chunks of UNR blocks per group
    const char* xi = X16b + (g0 + gbeg) * (int64_t)nslab * 1024 + lo;   // next chunk to ISSUE
    const char* qb = smem + lo;
    const int qstride = nslab * 1024;                         // LDS bytes between two query groups
    int64_t left = (gend - gbeg) * nc;                        // chunks left to multiply
    int64_t to_issue = left;                                  // chunks left to issue

    bf16x8_t a[PF][UNR];
    // (unconditional: a load under a branch makes the compiler's vmcnt bookkeeping assume it did not happen and wait for everything;
    // past the end of the stream the wave re-reads its last chunk and drops it)
    auto issue = [&](bf16x8_t (&dst)[UNR]) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) dst[u] = __builtin_nontemporal_load((const bf16x8_t*)(xi + u * 1024));
        if (to_issue > 1) xi += UNR * 1024;
        --to_issue;
    };
#pragma unroll
    for (int b = 0; b < PF - 1; ++b) issue(a[b]);
    int64_t g = gbeg;
    int c = 0;
    float m[QG];
    f32x4 acc[QG][2];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        m[qg] = -INFINITY;
        acc[qg][0] = acc[qg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (;;) {
        // (PF chunks per trip so that the register buffers are indexed statically)
#pragma unroll
        for (int b = 0; b < PF; ++b) {
            issue(a[(b + PF - 1) % PF]);
            const char* qc = qb + c * UNR * 1024;
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
#pragma unroll
                for (int qg = 0; qg < QG; ++qg)
                    acc[qg][u & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        a[b][u], *(const bf16x8_t*)(qc + qg * qstride + u * 1024), acc[qg][u & 1], 0, 0, 0);
            }
            if (++c == nc) {   // the group is complete
                c = 0;
                const int64_t col = g * 16 + (lane >> 4) * 4;
                // (the zero rows that pad the last group are not index rows: they must not raise a run maximum)
                const int64_t valid = nrows - col;
                const bool flush = ((g + 1) >> run_shift) != (g >> run_shift) || g + 1 == gend;   // last group of a run / the stream
#pragma unroll
                for (int qg = 0; qg < QG; ++qg) {
                    const f32x4 v = acc[qg][0] + acc[qg][1];
                    acc[qg][0] = acc[qg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const int q = qg * 16 + (lane & 15);
                    if (tiled) {
                        // tile (group, query group) = 16 queries x 16 rows = 1 KiB, query-major: the wave's store instruction covers
                        // the tile exactly once — eight full 128-B lines instead of sixteen 64-B pieces of sixteen different rows
                        if (q < nq) *(f32x4*)(S + ((g * QG + qg) * 256 + (lane & 15) * 16 + (lane >> 4) * 4)) = v;
                    } else if (q < nq) {
                        *(f32x4*)(S + (int64_t)q * lds_elems + col) = v;
                    }
                    if (M != nullptr) {
                        m[qg] = fmaxf(m[qg], fmaxf(fmaxf(valid > 0 ? v[0] : -INFINITY, valid > 1 ? v[1] : -INFINITY),
                                                   fmaxf(valid > 2 ? v[2] : -INFINITY, valid > 3 ? v[3] : -INFINITY)));
                        if (flush) {
                            float t = fmaxf(m[qg], __shfl_xor(m[qg], 16));
                            t = fmaxf(t, __shfl_xor(t, 32));
                            if (lane < 16 && qg * 16 + lane < nq)
                                atomicMax(M + (int64_t)(qg * 16 + lane) * ldm + (g >> run_shift), ~desc_key(t));
                            m[qg] = -INFINITY;
                        }
                    }
                }
                ++g;
            }
            if (--left == 0) return;
        }
    }
}

static bool g_narrow_attr[9][64];   // hipFuncSetAttribute once per kernel variant and device (it costs microseconds per call)

// q16b / x16b: the BLOCKED shadows; xrow0 a multiple of 16; scores of rows [xrow0, xrow0 + 16 * ceil(nrows / 16)) are written to
// S[q][row - xrow0] for q < nq <= 64 (the caller's row stride lds_elems covers the rounded-up count).
// tiled != 0: S is written as 1-KiB tiles instead, tile (group g, query group qg) at ((g * QG + qg) * 256) floats holding
// [16 queries][16 rows] with QG = 1, 2, 4 for nq <= 16, 32, 64 (what the run-maxima selection reads; coalesced stores).
// M (optional, all zero on entry): M[q][r] = ascending key (~desc_key) of the max score of query q over the VALID rows of
// run r = rows [r * (16 << run_shift), (r + 1) * (16 << run_shift)) of the launch.
// qf32 (optional, nq <= 16 only): the queries as fp32 rows [nq][d] with row stride ldqf, converted by the kernel itself (q16b unused)
int launch_score_narrow(const void* q16b, const void* x16b, int64_t ld_elems, int64_t xrow0, int64_t nrows, float* S,
                        int64_t lds_elems, int nq, uint32_t* M, int64_t ldm, int run_shift, int tiled, hipStream_t st,
                        const float* qf32, int64_t ldqf, int d) {
    if (nrows <= 0 || nq <= 0) return LDOT_OK;
    LDOT_REQUIRE(qf32 == nullptr || (nq <= 16 && d <= ld_elems && d % 32 == 0 && ldqf % 4 == 0 && ((uintptr_t)qf32 & 15) == 0), LDOT_EINVAL,
                 "fp32 query rows: at most 16 queries, d a multiple of 32, 16-byte aligned rows");
    const int nslab = (int)(ld_elems / 32);
    const int qg = nq <= 16 ? 1 : nq <= 32 ? 2 : 4;
    LDOT_REQUIRE(nq <= kNarrowMaxQueries && xrow0 % 16 == 0 && ld_elems % 64 == 0 && nslab * qg <= kNarrowMaxLdsKiB && run_shift >= 0,
                 LDOT_EINVAL, "bad narrow scan shape");
    const int64_t ngroups = (nrows + 15) / 16;
    LDOT_REQUIRE(lds_elems >= ngroups * 16 && (M == nullptr || ldm > ((ngroups - 1) >> run_shift)), LDOT_EINVAL,
                 "score row stride too short");
    // waves per CU by the query operand's LDS footprint: 20 (5 workgroups of 4 waves) for one query group, 12 for two, 8 (one
    // 512-thread workgroup) for four; every wave the same number of consecutive groups (the last one possibly fewer): with a fixed wave
    // count a 7706-group index would leave half of them a second group to do while the others idle
    const int wg_waves = qg == 4 ? 8 : 4;
    const int64_t max_waves = 256 * (qg == 1 ? 20 : qg == 2 ? 12 : 8);
    const int64_t per = (ngroups + max_waves - 1) / max_waves, waves = (ngroups + per - 1) / per;
    const int grid = (int)((waves + wg_waves - 1) / wg_waves);
    LDOT_REQUIRE(per < ((int64_t)1 << 30), LDOT_EINVAL, "too many rows for one narrow scan");
    const size_t lds = (size_t)nslab * qg * 1024;
    const char* q = (const char*)q16b;
    const char* x = (const char*)x16b;
    int dev = 0;
    LDOT_HIP_CHECK(hipGetDevice(&dev));
#define LDOT_NARROW(U, QG, T, PFV, SLOT)                                                                                          \
    do {                                                                                                                       \
        if (dev >= 64 || !g_narrow_attr[SLOT][dev]) {                                                                          \
            LDOT_HIP_CHECK(hipFuncSetAttribute((const void*)score_narrow_kernel<U, QG, T, PFV>,                                     \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kNarrowMaxLdsKiB * 1024));          \
            if (dev < 64) g_narrow_attr[SLOT][dev] = true;                                                                     \
        }                                                                                                                      \
        hipLaunchKernelGGL((score_narrow_kernel<U, QG, T, PFV>), dim3(grid), dim3(T), lds, st, q, x, nslab, xrow0 / 16, ngroups,    \
                           (int)per, nrows, S, lds_elems, nq, M, ldm, run_shift, tiled, qf32, ldqf, d);                                              \
    } while (0)
#define LDOT_NARROW_U(U, SLOT)               \
    do {                                     \
        if (qg == 1)                         \
            LDOT_NARROW(U, 1, 256, 2, SLOT);    \
        else if (qg == 2)                    \
            LDOT_NARROW(U, 2, 256, 2, SLOT + 1); \
        else                                 \
            LDOT_NARROW(U, 4, 512, 2, SLOT + 2); \
    } while (0)
    if (nslab % 8 == 0)
        LDOT_NARROW_U(8, 0);
    else if (nslab % 4 == 0)
        LDOT_NARROW_U(4, 3);
    else
        LDOT_NARROW_U(2, 6);
#undef LDOT_NARROW_U
#undef LDOT_NARROW
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

}  // namespace ldot
