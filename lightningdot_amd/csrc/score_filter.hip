// Fused bf16 MFMA score + threshold filter: the Q x N score matrix is never materialised.
//
// Replaces the sgemm + heap inside faiss IndexFlatIP.search (dvl/indexer/faiss_indexers.py:83) for large
// indexes.  Orientation is "swapped": A = index rows (M side), B = queries (N side), so in the MFMA C/D layout
// a lane owns ONE query column (col = lane & 31) and its accumulator registers hold that query's scores
// against 96 different index rows.  The per-query admission threshold tau (the current k'-th best score, a
// valid lower bound of the final one) therefore lives in a lane register and the filter is a v_max3 tree + one
// v_cmp per 4 scores with an exec-masked, almost never taken, append.
//
// Appends go to lane-private sub-pools in HBM (cursor in a VGPR, no atomics, no LDS): for a query q each of the
// 4 lanes x (row slices) that can produce candidates owns kPoolCap RECORDS.  A record is what a lane holds when its
// max-of-8 test fires: the 8 scores of accumulator registers 8h..8h+7 (rows rb + {0,1,2,3,8,9,10,11}) and rb — three 16-byte
// planes, stored entry-major and plane-major, pool[((q * kPoolCap + e) * 3 + plane) * nsubs + sub] in 16-byte units, so that the
// select kernel, which folds the pools into the running top-k' list between launches and raises tau, reads one entry level of
// all sub-pools as contiguous 16-byte words instead of one cache line per record.
//
// Work decomposition (256 persistent workgroups, block b observed on XCD b % 8; qg = query blocks per XCD):
//   XCD x, slot s in [0,32): qsub = s % qg, nsub = s / qg;  row stream = x*(32/qg) + nsub (nstreams = 256 / qg).
//   Work units u = g * ntiles + t (g = group of qg query blocks, t = row tile) are dealt round-robin to the streams
//   (u = stream (mod nstreams)), so streams differ by at most ONE tile per launch, not one per query group; the qg
//   workgroups of a stream multiply the same row tile with their own query block g*qg + qsub.
// With qg = 8 the 32 workgroups of an XCD work on 8 query panels x 4 adjacent row tiles, so the private L2 holds
// 8 Q panels (3 MiB) and streams each row panel once per query group.
//
// Operand layout: both bf16 shadows are read in the BLOCKED layout written by convert_rows_kernel (dst16b): 1 KiB blocks of
// 16 rows x 32 k, block (g, s) of a 16-row group g and K slab s at ((g * nslab + s) KiB.  One wave-level direct-to-LDS load
// instruction (64 lanes x 16 B = 16 rows x 64 B of one slab) then reads one contiguous KiB = eight full 128-B lines; with
// row-major operands the same instruction touched 16 half lines, and the L2 request rate (one 64-B request per half line,
// ~0.8 per clock and channel) bounded the L2->LDS stream: loads-only time 8.7 -> 5.9 ms, kernel -5.5 %.
//
// Earlier generations of this kernel (256x256 two-stage, commit da2a5a4; 256x256 on the LDS ring, commit 62cbd68) are in
// the history; their measurements are in DESIGN.md section 5.2.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "gemm_ring.h"
#include "kernels.h"

namespace ldot {

// ---- third generation: (64*MR) x 256 tile on the ring, A fragments recycled in place (gemm_ring.h) -------------
// max of four accumulator registers in two instructions.  fmaxf() would add a canonicalising v_max per MFMA
// output (hipcc cannot prove MFMA results are quiet); scores are never signalling NaNs, so v_max3 is applied
// directly.  The operands are MFMA results: the caller guarantees >= 18 wait states since the last MFMA issue
// (the s_nop block at the top of the epilogue) because hipcc does not pad hazards for inline asm.
__device__ __forceinline__ float max4_raw(float a0, float a1, float a2, float a3) {
    float m;
    asm volatile("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(m) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    return m;
}

__device__ __forceinline__ float max8_raw(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    float m;
    asm volatile(
        "v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max_f32 %0, %0, %8"
        : "=&v"(m)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
    return m;
}

// MFMA -> VALU read hazard cover for the inline-asm accumulator reads of the epilogue (32x32x16 bf16: 8 passes, <= 18 wait
// states since the last MFMA issue; hipcc does not pad hazards for inline asm)
__device__ __forceinline__ void filter_hazard_cover() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// threshold filter of one 32-row block (mr) of the wave tile: both query columns of the lane, eight scores per test
// (three v_max3 + v_max + compare + branch on the fast path).  A firing test does NOT localise the hit: the lane appends one
// record = the 8 scores (two 16-byte stores straight from the accumulator registers) + their base row to its sub-pool (cursor
// `cur`, clamped at kPoolCap; the true count is kept so that overflow is detectable) and the pool select thresholds them.
// pbase = (q * kPoolCap * 3) * nsubs + sub in 16-byte units; entry e, plane p at pbase + (e * 3 + p) * nsubs.
template <bool NOSTORE>
__device__ __forceinline__ void filter_epilogue_mr(const f32x16& acc0, const f32x16& acc1, int mr, const float (&tau)[2],
                                                   int (&cur)[2], uint32_t pbase0, uint32_t pstep, uint32_t nsubs,
                                                   uint4* __restrict__ pool, int32_t row_wave0) {
#pragma unroll
    for (int nr = 0; nr < 2; ++nr) {
        const f32x16& a = nr ? acc1 : acc0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float a0 = a[8 * h + 0], a1 = a[8 * h + 1], a2 = a[8 * h + 2], a3 = a[8 * h + 3];
            const float a4 = a[8 * h + 4], a5 = a[8 * h + 5], a6 = a[8 * h + 6], a7 = a[8 * h + 7];
            const float m = max8_raw(a0, a1, a2, a3, a4, a5, a6, a7);
            if (m >= tau[nr]) {   // rare (a few per tile and wave)
                // (rare path: the lane's row offset and the sub-pool base are derived here instead of living in registers)
                // registers 8h..8h+3: rows +0..3, 8h+4..8h+7: rows +8..11
                const int32_t rb = row_wave0 + 4 * (int32_t)((threadIdx.x & 63) >> 5) + mr * 32 + 16 * h;
                const int p = cur[nr];
                cur[nr] = p + 1;
                if (p < kPoolCap && !NOSTORE) {
                    uint4* rec = pool + (pbase0 + (uint32_t)nr * pstep + (uint32_t)p * (kPoolPlanes * nsubs));
                    rec[0] = make_uint4(__float_as_uint(a0), __float_as_uint(a1), __float_as_uint(a2), __float_as_uint(a3));
                    rec[nsubs] = make_uint4(__float_as_uint(a4), __float_as_uint(a5), __float_as_uint(a6), __float_as_uint(a7));
                    rec[2 * nsubs] = make_uint4((uint32_t)rb, 0u, 0u, 0u);
                }
            }
        }
    }
}

template <int MR, bool NOSTORE>
__device__ __forceinline__ void filter_epilogue_r(const f32x16 (&acc)[MR][2], const float (&tau)[2], int (&cur)[2],
                                                  uint32_t pbase0, uint32_t pstep, uint32_t nsubs, uint4* __restrict__ pool,
                                                  int32_t row_wave0) {
    filter_hazard_cover();
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
        filter_epilogue_mr<NOSTORE>(acc[mr][0], acc[mr][1], mr, tau, cur, pbase0, pstep, nsubs, pool, row_wave0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// qg_log2: log2 of the number of query blocks an XCD works on concurrently (8 for large batches; 1/2/4 for few
// query blocks, so that all 256 workgroups stream index rows even for a single query block — the serving shape).
template <int VAR>
__global__ __launch_bounds__(kRingThreads, 2) void score_filter_r6_kernel(
    const char* __restrict__ X16, int64_t ldx_b, int64_t row0, int64_t nrows, const char* __restrict__ Q16,
    int64_t ldq_b, int nqb, int nk, const float* __restrict__ tau_g, uint4* __restrict__ pool,
    int32_t* __restrict__ pool_cnt, int qg_log2) {
    constexpr int MR = 6;
    using Geo = RingGeom<MR>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RingCtx c;
    ring_ctx_init(c);
#ifdef LDOT_ABLATION
    if ((VAR & 16384) && c.wave >= 4) __builtin_amdgcn_s_setprio(1);   // static priority for the later-dispatched half (guide: +0..1 %)
    if ((VAR & 32768) && c.wave < 4) __builtin_amdgcn_s_setprio(1);
#endif
    const int qg = 1 << qg_log2;                       // query blocks in flight per XCD
    const int nstream = 32 >> qg_log2;                 // row streams per XCD
    const int nslices = 8 * nstream;                   // row slices of the launch
    const int nsubs = nslices * 4;                     // lane-private sub-pools per query
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qsub = slot & (qg - 1), nsub = slot >> qg_log2;
    const int slice = xcd * nstream + nsub;
    const int ntiles = (int)((nrows + Geo::kBM - 1) / Geo::kBM);
    const int sub = (slice * 2 + c.wm) * 2 + (c.lane >> 5);
    const int nq_iter = (nqb > qsub) ? (nqb - qsub + qg - 1) >> qg_log2 : 0;   // query groups in which this qsub is valid
    const int64_t nunits = (int64_t)nq_iter * ntiles;                            // (group, tile) units of this qsub
    const int ntile_total = (nunits > slice) ? (int)((nunits - slice + nslices - 1) / nslices) : 0;
    if (ntile_total == 0) return;
    const int64_t S = (int64_t)ntile_total * nk;
    const int g0 = slice / ntiles, t0 = slice % ntiles;                          // first unit of this stream

    // ---- load cursor (see the second-generation kernel) ---------------------------------------------------------
    // one per-lane offset serves every staging instruction of both operands (their row strides are equal): the row-group
    // step of instruction j (j * 8 groups of 16 rows) and the slab (KiB block) ride in the scalar offset
    const int vo = c.wave * 16 * (int)ldx_b + (c.lane >> 2) * 64 + c.st_col;
    const int jstep = 8 * 16 * (int)ldx_b;
    int l_q = g0, l_t = t0, l_k = 0;
    RingSrc sa, sb;
    sa.rsrc = ring_make_rsrc_n(X16 + (row0 + (int64_t)l_t * Geo::kBM) * ldx_b, Geo::kBM * ldx_b);
    sb.rsrc = ring_make_rsrc_n(Q16 + (int64_t)(qsub + l_q * qg) * kRBN * ldq_b, kRBN * ldq_b);
    int64_t issued = 0;
    // one slab = kLoads pieces per wave (3 of the row panel, 2 of the query panel).  piece(j) issues one of them into the stage
    // of slab `issued`; advance() moves the cursor to the next slab once all pieces are out.
    auto piece = [&](const int j) {
        char* st = smem + (int)(issued & 3) * Geo::kStage;
        const int k0b = l_k * 1024;   // slab l_k of a 16-row group = its l_k-th KiB block
        if ((VAR & 2) && issued >= 4) return;
        // cache policy of the two streams (ablation): bit 1 of aux = nt (non-temporal: the line is the first to leave the L2), bit 0 =
        // sc0.  VAR & 65536: row-panel loads nt (streamed once per query group: they should not evict the query panels, which every
        // unit re-reads); VAR & 131072: query-panel loads nt (the opposite, as a control)
        constexpr int kAuxA = (VAR & 65536) ? 2 : 0, kAuxB = (VAR & 131072) ? 2 : 0;
        if (j < Geo::kALoads)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sa.rsrc, (rg_lptr_t)(st + (j * 8 + c.wave) * 1024), 16, vo, k0b + j * jstep, 0, kAuxA);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sb.rsrc, (rg_lptr_t)(st + Geo::kAOpBytes + ((j - Geo::kALoads) * 8 + c.wave) * 1024),
                                                     16, vo, k0b + (j - Geo::kALoads) * jstep, 0, kAuxB);
    };
    auto advance = [&]() {
        ++issued;
        if (issued < S) {
            if (++l_k == nk) {   // next unit of this stream
                l_k = 0;
                l_t += nslices;
                if (l_t >= ntiles) {
                    do {
                        l_t -= ntiles;
                        ++l_q;
                    } while (l_t >= ntiles);
                    sb.rsrc = ring_make_rsrc_n(Q16 + (int64_t)(qsub + l_q * qg) * kRBN * ldq_b, kRBN * ldq_b);
                }
                sa.rsrc = ring_make_rsrc_n(X16 + (row0 + (int64_t)((VAR & 512) ? (l_t & 28) : (VAR & 64) ? (l_t & 31) : l_t) * Geo::kBM) * ldx_b, Geo::kBM * ldx_b);
            }
        }
    };
    auto issue = [&]() {
#pragma unroll
        for (int j = 0; j < Geo::kLoads; ++j) piece(j);
        advance();
    };
    // the pieces of a slab ride between the MFMAs of two k-steps: pieces 0..2 after row blocks 1, 3, 5 of the k-step that follows
    // the barrier (the stage they overwrite was vacated there), pieces 3, 4 after row blocks 1, 3 of the next k-step (before the
    // next barrier's counted vmcnt, which therefore still sees whole slabs)
    bool defer_burst = false, defer_burst_b = false;
    auto hook_post = [&](int mr) {
        if (VAR & 8192) return;   // ablation: the burst BEFORE the barrier (end of the k-step that precedes it), see hook_pre
        if (VAR & 2048) {     // ablation: the burst at the START of the k-step that follows the barrier
            if (mr == 0 && !defer_burst) issue();
            return;
        }
        if (VAR & 4096) {     // ablation: row-panel pieces after this k-step, query-panel pieces after the next one
            if (mr == MR - 1 && !defer_burst) {
#pragma unroll
                for (int j = 0; j < Geo::kALoads; ++j) piece(j);
            }
            return;
        }
        if (!(VAR & 128)) {   // default: the whole slab as a burst after the k-step (VAR & 128: piece by piece, see below)
            if (mr == MR - 1 && !defer_burst) issue();
            return;
        }
        // (a deferred slab is issued whole by the MODE 2 slab, after its filter blocks — round 2 issued these three pieces as well
        // AND the whole slab there: one slab too many per tile, which is what made this variant slower than the burst)
        if (defer_burst) return;
        if (mr & 1) piece(mr >> 1);
    };
    auto hook_pre = [&](int mr) {
        if (VAR & 8192) {
            if (mr == MR - 1) issue();
            return;
        }
        if (VAR & 4096) {
            if (mr == MR - 1 && !defer_burst_b) {
                piece(Geo::kALoads);
                piece(Geo::kALoads + 1);
                advance();
            }
            return;
        }
        if (!(VAR & 128)) return;
        if (defer_burst_b && !(VAR & 256)) return;   // MODE 2 with deferral: the whole slab follows the filter blocks
        if (mr == 1) piece(3);
        if (mr == 3) {
            piece(4);
            advance();
        }
    };
    issue();
    issue();
    issue();
    issue();
    wait_vmcnt<3 * Geo::kLoads>();
    __builtin_amdgcn_s_barrier();

    // ---- compute side --------------------------------------------------------------------------------------
    float tau[2];
    int cur[2] = {0, 0};
    uint32_t pbase0 = 0;                                             // sub-pool base of the lane's first query column
    const uint32_t pstep = 32u * kPoolCap * kPoolPlanes * (uint32_t)nsubs;   // ... the second one is 32 queries further
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
    // MODE 0: a slab inside a tile.  MODE 1: the first slab of the first tile (its first MFMAs take C = 0 instead of a
    // cleared accumulator).  MODE 2: the first slab of a later tile, FUSED with the threshold filter of the tile just
    // finished: block by block (mr), the filter reads the finished scores and the C = 0 MFMAs of the new tile overwrite
    // them, so the matrix pipe works on block mr while the VALU filters block mr + 1 (the filter alone leaves the matrix
    // pipe idle: all waves run it at the same time).  The LAST slab of a tile that a MODE 2 slab follows (defer_burst): its trailing
    // burst of slab loads is deferred until after the filter, so that the filter's pool stores do not queue behind 40 LDS-DMA
    // pieces in the CU's texture-address FIFO while the matrix pipe waits for the wave (VAR & 256: no deferral, ablation).
    int32_t epi_row_wave0 = 0;   // (uniform) first row of the wave's block of the tile whose filter is pending
    auto slab = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        char* st0 = smem + (int)(s & 3) * Geo::kStage;
        // k-step 0 of slab s (operands: a, b[0]); a <- k-step 1 of slab s, b[1] <- k-step 1 of slab s
        if (!(VAR & 4)) {
            if (MODE == 0) {
                ringr_step<MR, (VAR & 32) != 0>(c, f, 0, st0, 1, acc, hook_pre);
            } else {
                ringr_read_b<MR>(c, st0, 1, f.b[1]);
                const char* a_w = st0 + c.wm * (32 * MR * 64) + (c.frag_off0 ^ 32);
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (MODE == 2 && !(VAR & 1)) filter_hazard_cover();
                if (MODE == 2) defer_burst_b = true;    // (VAR & 4096 only: the explicit issue() below carries the whole slab)
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    if (MODE == 2 && !(VAR & 1)) {
                        filter_epilogue_mr<(VAR & 8) != 0>(acc[mr][0], acc[mr][1], mr, tau, cur, pbase0, pstep, (uint32_t)nsubs,
                                                           pool, epi_row_wave0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[mr][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mr], f.b[0][0], z, 0, 0, 0);
                    acc[mr][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mr], f.b[0][1], z, 0, 0, 0);
                    f.a[mr] = *(const bf16x8_t*)(a_w + mr * 2048);
                    if (MODE == 2) hook_pre(mr);   // (MODE 1 = the very first slab: no slab is half issued yet)
                    if (MODE == 2 && !(VAR & 1)) __builtin_amdgcn_sched_barrier(0);
                }
                if (MODE == 2) defer_burst_b = false;
                if (MODE == 2 && !(VAR & 256) && !(VAR & 8192)) issue();   // the burst the previous slab deferred
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of slab s are complete (WAR on its stage)
        if (VAR & 2)
            wait_vmcnt<0>();
        else
            wait_vmcnt<2 * Geo::kLoads>();                   // slab s+1 has landed (this thread's part) ...
        __builtin_amdgcn_s_barrier();                        // ... and everybody else's
        ++s;
        // k-step 1 of the old slab (operands: a, b[1]); a, b[0] <- k-step 0 of slab s (just opened)
        // ... and the first three pieces of slab s+3 (or a dummy) -> the stage slab s-1 vacated at the barrier
        if (!(VAR & 4)) ringr_step<MR, (VAR & 32) != 0>(c, f, 1, smem + (int)(s & 3) * Geo::kStage, 0, acc, hook_post);
        __builtin_amdgcn_sched_barrier(0);
    };
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;

    // the counters' query index is recomputed at store time (two VGPR pairs less live across the tile loop)
    auto store_counts = [&](int g) {
#pragma unroll
        for (int nr = 0; nr < 2; ++nr) {
            const int64_t qi = (int64_t)(qsub + g * qg) * kRBN + c.wn * 64 + nr * 32 + (c.lane & 31);
            pool_cnt[qi * nsubs + sub] = cur[nr];
        }
    };
    auto setup_group = [&](int g) {
#pragma unroll
        for (int nr = 0; nr < 2; ++nr) {
            const int64_t qi = (int64_t)(qsub + g * qg) * kRBN + c.wn * 64 + nr * 32 + (c.lane & 31);
            tau[nr] = (VAR & 16) ? INFINITY : ring_launder(tau_g[qi]);
            if (nr == 0) pbase0 = (uint32_t)(qi * kPoolCap * kPoolPlanes * nsubs + sub);
            cur[nr] = 0;
        }
    };
    int c_q = g0, c_t = t0, cur_q = g0;
    setup_group(c_q);
    if (VAR & 4) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.f;
    }
    slab(M1{});
#pragma unroll 1
    for (int j = 0; j < ntile_total; ++j) {
#pragma unroll 1
        for (int kk = 1; kk < nk; ++kk) {
            defer_burst = !(VAR & 256) && !(VAR & 8192) && kk == nk - 1 && j + 1 < ntile_total;   // (uniform) see MODE 2
            slab(M0{});
        }
        defer_burst = false;
        // tile j is complete in acc
        const int64_t trow = row0 + (int64_t)c_t * Geo::kBM;
        epi_row_wave0 = (int32_t)trow + c.wm * (32 * MR);
        c_t += nslices;
        while (c_t >= ntiles) {
            c_t -= ntiles;
            ++c_q;
        }
        if (j + 1 < ntile_total) {
            slab(M2{});                 // filter of tile j fused with the first slab of tile j + 1
            if (c_q != cur_q) {         // (uniform) the stream moves on to the next query group
                store_counts(cur_q);
                cur_q = c_q;
                setup_group(c_q);
            }
        } else if (!(VAR & 1)) {
            filter_epilogue_r<MR, (VAR & 8) != 0>(acc, tau, cur, pbase0, pstep, (uint32_t)nsubs, pool, epi_row_wave0);
        } else {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < 2; ++nr) asm volatile("" ::"v"(acc[mr][nr]));
        }
    }
    store_counts(cur_q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing dummy loads must land before the LDS is released
}

// index rows per fused tile (the host sizes launches and the row padding of the index with it)
int fused_tile_rows() { return RingGeom<6>::kBM; }

// query blocks an XCD works on concurrently for a batch of nqb query blocks (power of two <= 8); the fused launch then
// has 256 / qg row slices and 1024 / qg sub-pools per query
int fused_query_group(int64_t nq_pad) {
    const int64_t nqb = nq_pad / kRBN;
#ifdef LDOT_ABLATION
    static int force = -1;   // LDOT_DEBUG_QG: experiment override (1, 2, 4 or 8) — ablation builds only
    if (force < 0) {
        const char* e = getenv("LDOT_DEBUG_QG");
        force = e ? atoi(e) : 0;
    }
    if (force == 1 || force == 2 || force == 4 || force == 8) return force;
#endif
    return nqb >= 8 ? 8 : nqb >= 4 ? 4 : nqb >= 2 ? 2 : 1;
}

int launch_score_filter(const void* x16, int64_t ldx_elems, int64_t row0, int64_t nrows, const void* q16,
                        int64_t ldq_elems, int64_t nq_pad, int dpad, const float* tau, uint4* pool, int32_t* pool_cnt,
                        hipStream_t st) {
    if (nrows <= 0 || nq_pad <= 0) return LDOT_OK;
    LDOT_REQUIRE(ldx_elems == ldq_elems, LDOT_EINVAL, "index and query shadows must have the same row stride");
    auto rk = score_filter_r6_kernel<0>;
#ifdef LDOT_ABLATION
    // Ablation builds only (python -m lightningdot_amd.build --ablation; tools/ablate.sh): LDOT_DEBUG_VARIANT selects a profiling
    // variant of the kernel.  Results are meaningless under most of them, so the product library does not contain this hook.
    //   16 tau = +inf (epilogue fast path only), 18 = 16 + no global loads after the prologue, 17 no epilogue at all,
    //   64 every row tile aliased onto the first 32 (A panel always L2-resident), 128 slab loads piece by piece between the MFMAs,
    //   256 no deferral of the slab-load burst around the filter
    static int variant = -1;
    if (variant < 0) {
        const char* e = getenv("LDOT_DEBUG_VARIANT");
        variant = e ? atoi(e) : 0;
    }
    if (variant == 16) rk = score_filter_r6_kernel<16>;
    if (variant == 18) rk = score_filter_r6_kernel<18>;
    if (variant == 17) rk = score_filter_r6_kernel<17>;
    if (variant == 48) rk = score_filter_r6_kernel<48>;
    if (variant == 64) rk = score_filter_r6_kernel<64>;
    if (variant == 80) rk = score_filter_r6_kernel<80>;
    if (variant == 128) rk = score_filter_r6_kernel<128>;
    if (variant == 144) rk = score_filter_r6_kernel<144>;
    if (variant == 256) rk = score_filter_r6_kernel<256>;
    if (variant == 65536) rk = score_filter_r6_kernel<65536>;     // row-panel loads non-temporal
    if (variant == 65552) rk = score_filter_r6_kernel<65552>;     // ... with tau = +inf
    if (variant == 131072) rk = score_filter_r6_kernel<131072>;   // query-panel loads non-temporal (control)
    if (variant == 131088) rk = score_filter_r6_kernel<131088>;
    if (variant == 384) rk = score_filter_r6_kernel<384>;   // spread loads, no deferral
    if (variant == 400) rk = score_filter_r6_kernel<400>;
    if (variant == 528) rk = score_filter_r6_kernel<528>;
    if (variant == 4096) rk = score_filter_r6_kernel<4096>;
    if (variant == 4112) rk = score_filter_r6_kernel<4112>;
    if (variant == 16384) rk = score_filter_r6_kernel<16384>;
    if (variant == 32768) rk = score_filter_r6_kernel<32768>;
    if (variant == 8192) rk = score_filter_r6_kernel<8192>;
    if (variant == 8208) rk = score_filter_r6_kernel<8208>;
    if (variant == 2048) rk = score_filter_r6_kernel<2048>;
    if (variant == 2064) rk = score_filter_r6_kernel<2064>;   // 16 + 512: the 4 row streams of an XCD share ONE tile (everything L2-resident)
#endif
#ifdef LDOT_ABLATION
    LDOT_HIP_CHECK(hipFuncSetAttribute((const void*)rk, hipFuncAttributeMaxDynamicSharedMemorySize, RingGeom<6>::kLds));
#else
    static bool attr_set[kAttrDevices];
    LDOT_HIP_CHECK(set_max_dynamic_lds_once((const void*)rk, RingGeom<6>::kLds, attr_set));
#endif
    const int qg = fused_query_group(nq_pad);
    const int qg_log2 = qg == 8 ? 3 : qg == 4 ? 2 : qg == 2 ? 1 : 0;
    hipLaunchKernelGGL(rk, dim3(256), dim3(kRingThreads), RingGeom<6>::kLds, st, (const char*)x16, ldx_elems * 2, row0,
                       nrows, (const char*)q16, ldq_elems * 2, (int)(nq_pad / kRBN), dpad / kRBK, tau, pool, pool_cnt,
                       qg_log2);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

}  // namespace ldot
