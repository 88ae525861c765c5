// Dense score chunk:  S[q][n] = Q16[q,:] . X16[row0 + n,:]   (fp32 out, bf16 MFMA in)
// Used for small indexes (Flickr/COCO sized), for the warm-up chunk of the fused search and as the
// always-correct fallback.  Reference call site replaced: faiss IndexFlatIP.search's sgemm
// (dvl/indexer/faiss_indexers.py:83).
//
// Same MFMA engine as the fused kernel (gemm_ring.h: 384 x 256 tile, 4-stage LDS ring, blocked operands), with the roles
// swapped: A = queries (M side, 384 per tile), B = index rows (N side, 256 per tile), so that in the MFMA C/D layout a
// lane's 32 "col" lanes are consecutive index rows and every accumulator register stores a coalesced 128-byte run of
// S[q][n..n+32).  Persistent workgroups walk the (query tile, row tile) units with a fixed stride; the slab stream runs
// across units, so the next unit's loads are in flight during the store epilogue.
#include <stdlib.h>

#include <type_traits>

#include "gemm_ring.h"
#include "kernels.h"

namespace ldot {

template <int N>
__device__ __forceinline__ void dense_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__global__ __launch_bounds__(kRingThreads, 2) void score_dense_kernel(
    const char* __restrict__ Q16, int64_t ld_b, int tiles_m, int64_t q_rows, const char* __restrict__ X16, int64_t xrow0,
    int tiles_n, int nk, float* __restrict__ S, int64_t lds_elems, int64_t m_valid, int64_t tile_stride) {
    constexpr int MR = 6;
    using Geo = RingGeom<MR>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RingCtx c;
    ring_ctx_init(c);
    const int nunits = tiles_m * tiles_n;
    const int stride = gridDim.x;
    const int nmine = (nunits > (int)blockIdx.x) ? (nunits - (int)blockIdx.x + stride - 1) / stride : 0;
    if (nmine == 0) return;
    const int64_t S_total = (int64_t)nmine * nk;

    // ---- load cursor: unit u -> (tm = u % tiles_m, tn = u / tiles_m): consecutive workgroups share the row tile ------------
    const int vo = c.wave * 16 * (int)ld_b + (c.lane >> 2) * 64 + c.st_col;
    const int jstep = 8 * 16 * (int)ld_b;
    int l_u = blockIdx.x, l_k = 0;
    RingSrc sa, sb;
    auto set_src = [&](int u) {
        const int tm = u % tiles_m, tn = u / tiles_m;
        // the last query tile may be short: the buffer bound makes its missing rows read as zero
        const int64_t qrows = q_rows - (int64_t)tm * Geo::kBM;
        sa.rsrc = ring_make_rsrc_n(Q16 + (int64_t)tm * Geo::kBM * ld_b, (qrows < Geo::kBM ? qrows : Geo::kBM) * ld_b);
        sb.rsrc = ring_make_rsrc_n(X16 + (xrow0 + (int64_t)tn * tile_stride) * ld_b, kRBN * ld_b);   // (tile_stride > 256: a spread sample)
    };
    set_src(l_u);
    int64_t issued = 0;
    auto issue = [&]() {
        char* st = smem + (int)(issued & 3) * Geo::kStage;
        const int k0b = l_k * 1024;
#pragma unroll
        for (int j = 0; j < Geo::kALoads; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sa.rsrc, (rg_lptr_t)(st + (j * 8 + c.wave) * 1024), 16, vo, k0b + j * jstep, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sb.rsrc, (rg_lptr_t)(st + Geo::kAOpBytes + (j * 8 + c.wave) * 1024), 16,
                                                     vo, k0b + j * jstep, 0, 0);
        ++issued;
        if (issued < S_total) {
            if (++l_k == nk) {
                l_k = 0;
                l_u += stride;
                set_src(l_u);
            }
        }
    };
    issue();
    issue();
    issue();
    issue();
    dense_wait_vmcnt<3 * Geo::kLoads>();
    __builtin_amdgcn_s_barrier();

    f32x16 acc[MR][2];
    FragsR<MR> f;
    {
        const char* a_w = smem + c.wm * (32 * MR * 64) + c.frag_off0;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) f.a[mr] = *(const bf16x8_t*)(a_w + mr * 2048);
        ringr_read_b<MR>(c, smem, 0, f.b[0]);
        f.b[1][0] = f.b[0][0];
        f.b[1][1] = f.b[0][1];
    }
    int64_t s = 0;
    auto slab = [&](auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        char* st0 = smem + (int)(s & 3) * Geo::kStage;
        if (FIRST) {   // C = 0: no accumulator clearing pass
            ringr_read_b<MR>(c, st0, 1, f.b[1]);
            const char* a_w = st0 + c.wm * (32 * MR * 64) + (c.frag_off0 ^ 32);
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                acc[mr][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mr], f.b[0][0], z, 0, 0, 0);
                acc[mr][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mr], f.b[0][1], z, 0, 0, 0);
                f.a[mr] = *(const bf16x8_t*)(a_w + mr * 2048);
            }
        } else {
            ringr_step<MR>(c, f, 0, st0, 1, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dense_wait_vmcnt<2 * Geo::kLoads>();
        __builtin_amdgcn_s_barrier();
        ++s;
        ringr_step<MR>(c, f, 1, smem + (int)(s & 3) * Geo::kStage, 0, acc);
        __builtin_amdgcn_sched_barrier(0);
        issue();
    };

    int u = blockIdx.x;
#pragma unroll 1
    for (int j = 0; j < nmine; ++j, u += stride) {
        slab(std::true_type{});
#pragma unroll 1
        for (int kk = 1; kk < nk; ++kk) slab(std::false_type{});
        // store epilogue: (the stores join the vmcnt queue behind the ring loads; the counted waits only get more conservative)
        const int tm = u % tiles_m, tn = u / tiles_m;
        const int64_t m_base = (int64_t)tm * Geo::kBM + c.wm * (32 * MR) + 4 * (c.lane >> 5);
        const int64_t n_base = (int64_t)tn * kRBN + c.wn * 64 + (c.lane & 31);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = m_base + mr * 32 + (r & 3) + 8 * (r >> 2);
                    if (m < m_valid) S[m * lds_elems + n_base + nr * 32] = acc[mr][nr][r];   // pad queries are not stored
                }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing dummy loads must land before the LDS is released
}

// q16 / x16 are the BLOCKED shadows; q_rows = rows of the query shadow (multiple of 256, zero padded)
int launch_score_dense(const void* q16, int64_t ldq_elems, int64_t nq_pad, const void* x16, int64_t ldx_elems,
                       int64_t xrow0, int64_t nrows_pad, int dpad, float* S, int64_t lds_elems, int64_t nq_valid,
                       hipStream_t st, int64_t tile_stride) {
    if (nq_pad <= 0 || nrows_pad <= 0) return LDOT_OK;
    LDOT_REQUIRE(ldq_elems == ldx_elems, LDOT_EINVAL, "index and query shadows must have the same row stride");
    LDOT_REQUIRE(xrow0 % 16 == 0 && nrows_pad % kRBN == 0 && nq_pad % 256 == 0 && tile_stride % 16 == 0 && tile_stride >= kRBN, LDOT_EINVAL,
                 "unaligned dense chunk");
    using Geo = RingGeom<6>;
    const int tiles_m = (int)((nq_pad + Geo::kBM - 1) / Geo::kBM), tiles_n = (int)(nrows_pad / kRBN);
    const int nunits = tiles_m * tiles_n;
    static bool attr_set[kAttrDevices];
    LDOT_HIP_CHECK(set_max_dynamic_lds_once((const void*)score_dense_kernel, Geo::kLds, attr_set));
    hipLaunchKernelGGL(score_dense_kernel, dim3(nunits < 256 ? nunits : 256), dim3(kRingThreads), Geo::kLds, st,
                       (const char*)q16, ldq_elems * 2, tiles_m, nq_pad, (const char*)x16, xrow0, tiles_n, dpad / kRBK, S,
                       lds_elems, nq_valid, tile_stride);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

}  // namespace ldot
