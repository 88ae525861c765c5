// Dense score chunk:  S[q][n] = Q16[q,:] . X16[row0 + n,:]   (fp32 out, bf16 MFMA in)
// Used for small indexes (Flickr/COCO sized), for the warm-up chunk of the fused search and as the
// always-correct fallback.  Reference call site replaced: faiss IndexFlatIP.search's sgemm
// (dvl/indexer/faiss_indexers.py:83).
//
// Round 5: the same engine as the fused filter kernel (score_filter.hip) — v_mfma_f32_16x16x32_bf16, 1-KiB fragment blocks of the
// blocked shadows, direct-to-LDS slab ring, waves arranged 4 (rows) x 2 (queries), "swapped" orientation (A = index rows, B = queries)
// — with a store epilogue instead of the filter, on a tile sized for SMALL problems:
//   * workgroup tile 256 index rows x (32 * CB) queries, wave tile 64 x (16 * CB) = 4 x CB MFMA tiles;
//   * CB = 4 (256 x 128): 64 accumulator VGPRs, <= 128 VGPRs in all, 3-stage ring of 24 KiB slabs = 72 KiB -> TWO workgroups per CU:
//     one workgroup's store drain (the epilogue's stores sit in the same vmcnt queue as its ring loads) runs under the other's MFMAs,
//     and 5000 x 1000 (Flickr) is 160 units instead of 56 of the previous 384 x 256 tile;
//   * CB = 8 (256 x 256): 128 accumulator VGPRs, 4-stage ring of 32 KiB slabs, one workgroup per CU — fewer LDS fragment reads per
//     MFMA (12 per 32 instead of 8 per 16) for chunks with many units per CU;
//   * in the C/D layout of 16x16x32 a lane holds FOUR CONSECUTIVE index rows of one query column: every accumulator quad is one
//     16-byte store into S[q][n .. n + 4) (buffer store: tile corner in the scalar resource, ONE address register per column block;
//     the resource's bound drops the pad queries' stores), 64 contiguous bytes per query and store instruction;
//   * the K sum runs in the order of the fused kernel (one MFMA per 32-deep slab, slabs ascending, C = 0 in the first slab), so a
//     score computed here equals the fused kernel's bit for bit.
// Persistent workgroups walk the (query tile, row tile) units with a fixed stride; the slab stream runs across units, so the next
// unit's loads are in flight during the store epilogue.
// (The round-1..4 kernel — v_mfma_f32_32x32x16_bf16, 384-query x 256-row tiles, 19 spilled VGPRs — is in the history: commit d122ec9.)
#include <stdlib.h>

#include <type_traits>

#include "gemm_ring.h"
#include "kernels.h"

namespace ldot {

template <int N>
__device__ __forceinline__ void dense_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int CB, int STAGES>
struct DenseGeom {
    static constexpr int kRowBlocks = 4;                           // 16-row blocks of a wave tile (64 rows)
    static constexpr int kRows = 4 * kRowBlocks * 16;              // index rows per tile (4 wave rows)
    static constexpr int kQ = 2 * CB * 16;                         // queries per tile (2 wave columns)
    static constexpr int kABytes = kRows * kRBK * 2;               // 16 KiB: the row slab
    static constexpr int kBBytes = kQ * kRBK * 2;                  // 8 / 16 KiB: the query slab
    static constexpr int kStage = kABytes + kBBytes;
    static constexpr int kLds = STAGES * kStage;
    static constexpr int kALoads = kABytes / 1024 / 8;             // direct-to-LDS pieces per wave and slab (rows)
    static constexpr int kBLoads = kBBytes / 1024 / 8;             // ... (queries)
    static constexpr int kLoads = kALoads + kBLoads;
};

template <int CB, int STAGES>
__global__ __launch_bounds__(kRingThreads, (CB <= 4 ? 4 : 2)) void score_dense_t16_kernel(
    const char* __restrict__ X16, int64_t ld_b, int64_t xrow0, int tiles_n, int64_t tile_stride, const char* __restrict__ Q16,
    int tiles_q, int nk, float* __restrict__ S, int64_t lds_elems, int64_t m_valid) {
    using Geo = DenseGeom<CB, STAGES>;
    constexpr int RB = Geo::kRowBlocks;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;           // 4 wave rows x 2 wave columns
    const int nunits = tiles_q * tiles_n;
    const int stride = gridDim.x;
    const int nmine = (nunits > (int)blockIdx.x) ? (nunits - (int)blockIdx.x + stride - 1) / stride : 0;
    if (nmine == 0) return;
    const int64_t S_total = (int64_t)nmine * nk;

    // ---- load cursor: unit u -> (tq = u % tiles_q, tn = u / tiles_q): consecutive workgroups share the row tile --------------------
    // staging piece p = j * 8 + wave fills LDS bytes [p * 1024, + 1024) of its operand = the 1-KiB fragment block of 16-row group p;
    // lane i lands on row (i >> 2), physical 16-byte chunk (i & 3) and fetches logical chunk (i & 3) ^ ((row >> 1) & 3)
    const int st_col = ((lane & 3) ^ ((lane >> 3) & 3)) << 4;
    const int vo = wave * 16 * (int)ld_b + (lane >> 2) * 64 + st_col;
    const int jstep = 8 * 16 * (int)ld_b;
    int l_u = blockIdx.x, l_k = 0, l_st = 0;
    RingSrc sa, sb;
    auto set_src = [&](int u) {
        const int tq = u % tiles_q, tn = u / tiles_q;
        sa.rsrc = ring_make_rsrc_n(X16 + (xrow0 + (int64_t)tn * tile_stride) * ld_b, Geo::kRows * ld_b);   // (tile_stride > 256: a spread sample)
        sb.rsrc = ring_make_rsrc_n(Q16 + (int64_t)tq * Geo::kQ * ld_b, Geo::kQ * ld_b);
    };
    set_src(l_u);
    int64_t issued = 0;
    auto issue = [&]() {
        char* st = smem + l_st * Geo::kStage;
        const int k0b = l_k * 1024;   // slab k of a 16-row group = its k-th KiB block
#pragma unroll
        for (int j = 0; j < Geo::kALoads; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sa.rsrc, (rg_lptr_t)(st + (j * 8 + wave) * 1024), 16, vo, k0b + j * jstep, 0, 0);
#pragma unroll
        for (int j = 0; j < Geo::kBLoads; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sb.rsrc, (rg_lptr_t)(st + Geo::kABytes + (j * 8 + wave) * 1024), 16, vo,
                                                     k0b + j * jstep, 0, 0);
        ++issued;
        if (++l_st == STAGES) l_st = 0;
        if (issued < S_total) {
            if (++l_k == nk) {
                l_k = 0;
                l_u += stride;
                set_src(l_u);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < STAGES; ++i) issue();
    dense_wait_vmcnt<(STAGES - 1) * Geo::kLoads>();
    __builtin_amdgcn_s_barrier();

    // ---- compute side ----------------------------------------------------------------------------------------------------------
    f32x4 acc[RB][CB];
    constexpr int kARing = 2;      // A fragment ring (row block i uses a[i % 2], refilled with block i + 2 right after its CB MFMAs)
    bf16x8_t a[kARing], b[CB];     // the B fragments are single-buffered and refilled in place during the last row block of a slab
    const int frow = lane & 15;
    const int foff = frow * 64 + (((lane >> 4) ^ ((frow >> 1) & 3)) << 4);   // lane's 16 bytes inside a 1-KiB fragment block
    const int a_off = wm * (RB * 1024) + foff, b_off = Geo::kABytes + wn * (CB * 1024) + foff;
#pragma unroll
    for (int i = 0; i < kARing; ++i) a[i] = *(const bf16x8_t*)(smem + a_off + i * 1024);
#pragma unroll
    for (int j = 0; j < CB; ++j) b[j] = *(const bf16x8_t*)(smem + b_off + j * 1024);
    int c_st = 0;   // stage of the slab being multiplied
    auto slab = [&](auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;   // first slab of a unit: C = 0, no accumulator clearing pass
        const int n_st = c_st + 1 == STAGES ? 0 : c_st + 1;
        const char* a_cur = smem + c_st * Geo::kStage + a_off;
        const char* a_nxt = smem + n_st * Geo::kStage + a_off;
        const char* b_nxt = smem + n_st * Geo::kStage + b_off;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (i == RB - kARing) {
                // every fragment of the current stage is in registers (the stage may be refilled), the next slab must have landed: its
                // first fragment is read right below
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dense_wait_vmcnt<(STAGES - 2) * Geo::kLoads>();
                __builtin_amdgcn_s_barrier();
            }
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i % kARing], b[j], FIRST ? z : acc[i][j], 0, 0, 0);
                if (i == RB - 1) b[j] = *(const bf16x8_t*)(b_nxt + j * 1024);
            }
            a[i % kARing] = *(const bf16x8_t*)(i + kARing < RB ? a_cur + (i + kARing) * 1024 : a_nxt + (i + kARing - RB) * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
        c_st = n_st;
        issue();   // the stage this slab vacated at its barrier takes the slab STAGES ahead
        __builtin_amdgcn_sched_barrier(0);
    };

    // lane's byte offset inside a 16-query column block of a tile of S: query (lane & 15), rows wm * 64 + 4 * (lane >> 4)
    const uint32_t vq = (uint32_t)(lane & 15) * (uint32_t)(lds_elems * 4) + (uint32_t)(wm * 64 + 4 * (lane >> 4)) * 4u;
    int u = blockIdx.x;
#pragma unroll 1
    for (int t = 0; t < nmine; ++t, u += stride) {
        slab(std::true_type{});
#pragma unroll 1
        for (int kk = 1; kk < nk; ++kk) slab(std::false_type{});
        // store epilogue (the stores join the vmcnt queue behind the ring loads: the counted waits only get more conservative).
        // One buffer resource per column block: base = the block's corner in S, bound = its VALID queries — the pad queries' stores fall
        // outside the bound and are dropped by the hardware (S may hold the valid queries only: dense_scan_wide)
        const int tq = u % tiles_q, tn = u / tiles_q;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const int64_t qj = (int64_t)tq * Geo::kQ + wn * (CB * 16) + j * 16;
            int64_t nv = m_valid - qj;
            nv = nv < 0 ? 0 : nv > 16 ? 16 : nv;
            const __amdgpu_buffer_rsrc_t sr =
                __builtin_amdgcn_make_buffer_rsrc((void*)(S + qj * lds_elems + (int64_t)tn * Geo::kRows), 0, (int)(nv * lds_elems * 4), 0x00020000);
#pragma unroll
            for (int i = 0; i < RB; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), sr, vq + (uint32_t)i * 64u, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing dummy loads must land before the LDS is released
}

template <int CB, int STAGES>
static int launch_dense_variant(const void* q16, int64_t ld_elems, int64_t nq_pad, const void* x16, int64_t xrow0, int64_t nrows_pad,
                                int dpad, float* S, int64_t lds_elems, int64_t nq_valid, hipStream_t st, int64_t tile_stride, bool* attr_set, hipEvent_t ev_a,
                                hipEvent_t ev_b) {
    using Geo = DenseGeom<CB, STAGES>;
    constexpr int kWgPerCu = CB <= 4 ? 2 : 1;
    const int tiles_q = (int)(nq_pad / Geo::kQ), tiles_n = (int)(nrows_pad / Geo::kRows);
    const int64_t nunits = (int64_t)tiles_q * tiles_n;
    auto kern = score_dense_t16_kernel<CB, STAGES>;
    LDOT_HIP_CHECK(set_max_dynamic_lds_once((const void*)kern, Geo::kLds, attr_set));
    const int grid = (int)std::min<int64_t>(nunits, 256 * kWgPerCu);
    if (ev_a) (void)hipEventRecord(ev_a, st);   // (LDOT_OPT_PROFILE: events around the launch; see launch_score_filter)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kRingThreads), Geo::kLds, st, (const char*)x16, ld_elems * 2, xrow0, tiles_n,
                       tile_stride, (const char*)q16, tiles_q, dpad / kRBK, S, lds_elems, nq_valid);
    if (ev_b) (void)hipEventRecord(ev_b, st);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// time of a chunk in units of "one CU multiplying a 256 x 256 x K tile": the variant with the smaller estimate runs.  256 x 128 tiles:
// two workgroups share a CU (each at half its rate), a CU's load is ceil(units / 256) half tiles; f4 = what a half-tile workgroup pays
// for its 8 LDS fragment reads per 16 MFMAs against 12 per 32 (calibrated on the COCO shapes and the headline warm-up chunk:
// profiles/r05_dense_variants.txt)
static double dense_cost(int64_t nq_pad, int64_t nrows_pad, int cb, double f4) {
    const int64_t units = (nq_pad / (32 * cb)) * (nrows_pad / 256);
    const double per_cu = (double)((units + 255) / 256);
    return cb <= 4 ? per_cu * 0.5 * f4 : per_cu;
}

// q16 / x16 are the BLOCKED shadows; nq_pad = rows of the query shadow (multiple of 256, zero padded)
int launch_score_dense(const void* q16, int64_t ldq_elems, int64_t nq_pad, const void* x16, int64_t ldx_elems,
                       int64_t xrow0, int64_t nrows_pad, int dpad, float* S, int64_t lds_elems, int64_t nq_valid,
                       hipStream_t st, int64_t tile_stride, hipEvent_t ev_a, hipEvent_t ev_b) {
    if (nq_pad <= 0 || nrows_pad <= 0) return LDOT_OK;
    LDOT_REQUIRE(ldq_elems == ldx_elems, LDOT_EINVAL, "index and query shadows must have the same row stride");
    LDOT_REQUIRE(xrow0 % 16 == 0 && nrows_pad % kRBN == 0 && nq_pad % 256 == 0 && tile_stride % 16 == 0 && tile_stride >= kRBN, LDOT_EINVAL,
                 "unaligned dense chunk");
    LDOT_REQUIRE(lds_elems >= 256 && lds_elems * 4 * 16 < ((int64_t)1 << 31), LDOT_EINVAL, "dense chunk row stride out of range");
    static bool attr_set4[kAttrDevices], attr_set8[kAttrDevices];
    double f4 = 1.08;
#ifdef LDOT_ABLATION
    if (const char* e = getenv("LDOT_DEBUG_DENSE_F4")) f4 = atof(e);
#endif
    int cb = dense_cost(nq_pad, nrows_pad, 4, f4) <= dense_cost(nq_pad, nrows_pad, 8, f4) ? 4 : 8;
#ifdef LDOT_ABLATION
    static int force = -1;   // LDOT_DEBUG_DENSE_CB: experiment override (4 or 8) — ablation builds only
    if (force < 0) {
        const char* e = getenv("LDOT_DEBUG_DENSE_CB");
        force = e ? atoi(e) : 0;
    }
    if (force == 4 || force == 8) cb = force;
#endif
    if (cb == 4)
        return launch_dense_variant<4, 3>(q16, ldq_elems, nq_pad, x16, xrow0, nrows_pad, dpad, S, lds_elems, nq_valid, st, tile_stride, attr_set4, ev_a, ev_b);
    return launch_dense_variant<8, 4>(q16, ldq_elems, nq_pad, x16, xrow0, nrows_pad, dpad, S, lds_elems, nq_valid, st, tile_stride, attr_set8, ev_a, ev_b);
}

}  // namespace ldot
