// Fused bf16 MFMA score + threshold filter: the Q x N score matrix is never materialised.
//
// Replaces the sgemm + heap inside faiss IndexFlatIP.search (dvl/indexer/faiss_indexers.py:83) for large
// indexes.  Orientation is "swapped": A = index rows (M side), B = queries (N side), so in the MFMA C/D layout
// a lane owns query COLUMNS and its accumulator registers hold those queries' scores against 48 different index rows each.
// The per-query admission threshold tau (the current k'-th best score, a valid lower bound of the final one) therefore lives in
// lane registers and the filter is a v_max3 tree + one v_cmp per 8 scores with an exec-masked, almost never taken, append.
//
// Round 3 — the tile engine is built on v_mfma_f32_16x16x32_bf16 (16 x 16 tiles, K = 32 per instruction) instead of
// v_mfma_f32_32x32x16_bf16.  Same flops per clock on paper; under the chip's power limit, with random operands, the 16x16x32 shape
// sustains 2020 TFLOP/s MFMA-only against 1725 (it accumulates twice the K per accumulator read-modify-write), and this kernel's slab
// loop (LDS fragment stream + direct-to-LDS slab loads) 1600-1615 against 1405-1455 (tools/mfma_ceiling.hip, lines "T16" / "W2").
//   * workgroup = 512 threads = 8 waves; workgroup tile 384 rows x 256 queries; 4-stage LDS ring of 32-deep K slabs (4 x 40 KiB = all
//     160 KiB), ONE k-step per slab;
//   * Round 4: the waves are arranged 4 (rows) x 2 (queries) — wave tile 96 x 128 = 6 x 8 MFMA tiles, the same 192 accumulator VGPRs
//     — instead of 2 x 4 (192 x 64): 6 + 8 = 14 fragment reads per slab and wave instead of 12 + 4 = 16, and in tools/mfma_ceiling.hip
//     ("T16B") the slab loop carries 67.3-67.5 % of the peak against 64.8-65.2 % on the same box (profiles/r04_mfma_ceiling_t16b.txt);
//   * a fragment (16 rows x 32 k) is one 1-KiB block of the ring image: lane l reads row l & 15, 16-byte chunk (l >> 4) ^ ((row >> 1) & 3)
//     (the swizzle is applied on the per-lane SOURCE address of the direct-to-LDS loads, conflict-free ds_read_b128);
//   * A fragments live in a ring of TWO register quads (kARing: row block i uses a[i % 2], refilled with block i + 2 right after its eight
//     MFMAs; three quads — round 4 — cost four registers the filter's rare path needed); the EIGHT B fragments are single-buffered and refilled in place during the last row block of a slab (b[j] right after its
//     last MFMA: seven MFMAs = 112 cycles before its first use in the next slab);
//   * slabs are issued 3 ahead and retired by a counted s_waitcnt vmcnt + one raw s_barrier per slab, placed before row block 3: by
//     then every fragment of the current stage is in registers (the stage is free for the slab 4 ahead) and the next stage is about
//     to be read.
//
// C/D layout of 16x16x32: lane l holds rows (l >> 4) * 4 .. + 4 of column l & 15.  A lane therefore owns EIGHT query columns (column
// block j = 0..7: query wn * 128 + j * 16 + (l & 15)) and, per column, 4 rows of each of the wave's 6 row blocks.
//
// Appends go to sub-pools in HBM (cursor in a VGPR byte, no atomics, no LDS).  For a query q each of the 4 (wave rows) x (row slices) that
// can produce candidates owns a sub-pool of kPoolCap RECORDS, in which the four lanes of a query column (lanes c, c + 16, c + 32, c + 48:
// the four row quads of a 16-row block) own eight entries and one counter byte each — 128 sub-pools and counter words per query at 32 row
// slices, with lane-private cursors (kernels.h).  A record is what a lane holds when its
// max-of-8 test fires: the 4 + 4 scores of two vertically adjacent tiles (rows rb + {0,1,2,3} and rb + 16 + {0,1,2,3}) and rb — three
// 16-byte planes, stored entry-major and plane-major, pool[((q * kPoolCap + e) * 3 + plane) * nsubs + sub] in 16-byte units, so that the
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
// 16 rows x 32 k, block (g, s) of a 16-row group g and K slab s at ((g * nslab + s) KiB — exactly one MFMA operand.  One wave-level
// direct-to-LDS load instruction (64 lanes x 16 B = 16 rows x 64 B of one slab) reads one contiguous KiB = eight full 128-B lines.
//
// Earlier generations of this kernel (256x256 two-stage, commit da2a5a4; 256x256 on the LDS ring, commit 62cbd68; 384x256 on
// v_mfma_f32_32x32x16_bf16 with the filter fused into the next tile's first slab, rounds 1-3 up to commit 34b001c) are in the history; their measurements are in
// DESIGN.md section 5.2.
#include <algorithm>
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "gemm_ring.h"
#include "kernels.h"

namespace ldot {

// max of eight accumulator registers in four instructions (max8x2_raw: two such maxima at a time).  fmaxf() would add a canonicalising v_max per MFMA
// output (hipcc cannot prove MFMA results are quiet); scores are never signalling NaNs, so v_max3 is applied
// directly.  The operands are MFMA results: the caller guarantees the MFMA -> VALU read wait states (hipcc does not pad hazards for
// inline asm).
// Two INTERLEAVED dependency chains per statement, so that no instruction waits for the result of the one before it (18 asm operands — the
// limit is 30, so four chains do not fit one statement).
__device__ __forceinline__ void max8x2_raw(const f32x4& a0, const f32x4& a1, const f32x4& b0, const f32x4& b1, float& ma, float& mb) {
    asm volatile(
        "v_max3_f32 %0, %2, %3, %4\n\tv_max3_f32 %1, %10, %11, %12\n\t"
        "v_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %1, %1, %13, %14\n\t"
        "v_max3_f32 %0, %0, %7, %8\n\tv_max3_f32 %1, %1, %15, %16\n\t"
        "v_max_f32 %0, %0, %9\n\tv_max_f32 %1, %1, %17"
        : "=&v"(ma), "=&v"(mb)
        : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]),
          "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3]));
}

// MFMA -> VALU read hazard cover for the inline-asm accumulator reads of the stand-alone epilogue (the accumulators were written by
// the MFMAs just issued; hipcc does not pad hazards for inline asm)
__device__ __forceinline__ void filter_hazard_cover() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the lane id, produced by an instruction the compiler can neither hoist nor keep: the rare paths of the tile loop (an admitted record,
// a change of query group) derive their per-lane addresses from it on the spot.  Left to itself the compiler hoists those loop-invariant
// per-lane values out of the tile loop (nine registers at the last count), runs out of registers and spills inside the rare paths —
// and a scratch RELOAD is an s_waitcnt vmcnt(0), which drains the LDS-DMA ring.
__device__ __forceinline__ uint32_t lane_now() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

constexpr int kT16RowBlocks = 6;    // 16-row blocks of a wave tile (96 rows)
constexpr int kT16ColBlocks = 8;    // 16-query blocks of a wave tile (128 queries)

constexpr int kFiltCols = 4;        // column blocks per filter call (half a wave tile: the thresholds of a half are unpacked together)

// threshold filter of one PAIR of row blocks (2p, 2p + 1) over FOUR of the lane's eight query columns (column blocks jbase .. jbase + 3),
// eight scores per test (three v_max3 + v_max + compare on the fast path, one wave-uniform branch per call).  A firing test does NOT
// localise the hit: the lane appends one record = the 8 scores (two 16-byte stores straight from the accumulator registers) + their
// base row to ITS entries of the column's sub-pool and the pool select thresholds them.  Cursor byte jj of `curp` is the lane's own count
// for column block jbase + jj (saturating at 255, so that overflow is detectable; records beyond kPoolGroupCap are dropped).  Record
// address in 16-byte units: (q * kPoolCap * 3) * nsubs + sub + (e * 3 + plane) * nsubs with q = q_u + 16 j + (lane & 15) and
// e = (lane >> 4) * kPoolGroupCap + count: pbase_u carries the wave-uniform part incl. the sub-pool (a scalar), the lane's part is derived in
// the rare path.
template <int SMODE>   // 0 product; ablation builds: 1 no record stores, 2 every record store issued twice (same address, same data)
__device__ __forceinline__ void filter_admit(const f32x4* lo, const f32x4* hi, int p, int jbase, const float (&m)[kFiltCols],
                                             const float (&tau)[kFiltCols], uint32_t& curp, uint32_t pbase_u, uint32_t pstep,
                                             uint32_t nsubs, uint4* __restrict__ pool, int32_t row_w) {
    const bool any = (m[0] >= tau[0]) | (m[1] >= tau[1]) | (m[2] >= tau[2]) | (m[3] >= tau[3]);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(any) == 0, 1)) return;   // (laid out of line)
    // The rare path's cost follows its INSTRUCTION count (one wave issuing dependent instructions), not its store count.  Everything per
    // lane that does not depend on the hit — byte offset of the first of the lane's entries (vo0), first of the lane's rows in the pair
    // (rb) — is derived here from lane_now(): values that live across the slab loop would be spilled (see there).
    const uint32_t ln = lane_now();
    const uint32_t g4 = ln >> 4;
    const uint32_t vo0 = (__umul24(ln & 15u, pstep >> 4) << 4) + __umul24(g4, (kPoolGroupCap * kPoolPlanes * 16u) * nsubs);   // (24-bit multiplies: full rate)
    const int32_t rb = row_w + 4 * (int32_t)g4 + p * 32;
#pragma unroll
    for (int jj = 0; jj < kFiltCols; ++jj) {
        if (m[jj] >= tau[jj]) {
            const uint32_t e = (curp >> (8 * jj)) & 255u;
            if (e < 255u) curp += 1u << (8 * jj);
            if (e < (uint32_t)kPoolGroupCap && SMODE != 1) {
                // buffer stores: the wave-uniform base lives in the (scalar) resource, the plane and the column block in the scalar
                // offset, so the lane's address is ONE register
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void*)(pool + pbase_u), 0, (int)(pstep * 16u * kT16ColBlocks), 0x00020000);
                const uint32_t vo = vo0 + __umul24(e, kPoolPlanes * nsubs * 16u);
                // (computed HERE, opaquely: left to itself the compiler hoists the 24 store offsets of a tile's columns and planes out of the tile
                // loop into scalar registers, spills them, and this path reads them back lane by lane)
                uint32_t so;
                asm volatile("s_mul_i32 %0, %1, %2" : "=s"(so) : "s"(pstep), "s"((uint32_t)(jbase + jj) * 16u));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo[jj]), pr, vo, so, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi[jj]), pr, vo, so + nsubs * 16u, 0);
                __builtin_amdgcn_raw_buffer_store_b32((uint32_t)rb, pr, vo, so + nsubs * 32u, 0);
                if (SMODE == 2) {   // (measurement: what do the stores themselves cost?  twice the store instructions, the same records)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo[jj]), pr, vo, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi[jj]), pr, vo, so + nsubs * 16u, 0);
                    __builtin_amdgcn_raw_buffer_store_b32((uint32_t)rb, pr, vo, so + nsubs * 32u, 0);
                }
            }
        }
    }
}

// maxima (two statements of two interleaved chains) + admission of a pair over four column blocks
template <int SMODE>
__device__ __forceinline__ void filter_pair(const f32x4* lo, const f32x4* hi, int p, int jbase, const float (&tau)[kFiltCols],
                                            uint32_t& curp, uint32_t pbase_u, uint32_t pstep, uint32_t nsubs, uint4* __restrict__ pool,
                                            int32_t row_w) {
    float m[kFiltCols];
    max8x2_raw(lo[0], hi[0], lo[1], hi[1], m[0], m[1]);
    max8x2_raw(lo[2], hi[2], lo[3], hi[3], m[2], m[3]);
    filter_admit<SMODE>(lo, hi, p, jbase, m, tau, curp, pbase_u, pstep, nsubs, pool, row_w);
}

// VAR (ablation builds): 1 no filter at all, 8 no record stores, 16 tau = +inf (fast path only), 128 program order not pinned.
// qg_log2: log2 of the number of query blocks an XCD works on concurrently (8 for large batches; 1/2/4 for few
// query blocks, so that all 256 workgroups stream index rows even for a single query block).
template <int VAR>
__global__ __launch_bounds__(kRingThreads, 2) void score_filter_t16_kernel(
    const char* __restrict__ X16, int64_t ldx_b, int64_t row0, int64_t nrows, const char* __restrict__ Q16,
    int64_t ldq_b, int nqb, int nk, const float* __restrict__ tau_g, uint4* __restrict__ pool,
    int32_t* __restrict__ pool_cnt, int qg_log2, ScanOrder so, uint2* __restrict__ cur_save, int later_chunk) {
    using Geo = RingGeom<6>;   // 384-row A slab + 256-row B slab per stage, 5 direct-to-LDS pieces per wave and slab
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;           // 4 wave rows x 2 wave columns: wave tile 96 rows x 128 queries
    const int qg = 1 << qg_log2;                       // query blocks in flight per XCD
    const int nstream = 32 >> qg_log2;                 // row streams per XCD
    const int nslices = 8 * nstream;                   // row slices of the launch
    const int nsubs = nslices * kPoolSubsPerSlice;     // sub-pools per query (one per row slice and wave row)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qsub = slot & (qg - 1), nsub = slot >> qg_log2;
    const int slice = xcd * nstream + nsub;
    const int ntiles = (int)((nrows + Geo::kBM - 1) / Geo::kBM);
    const int nq_iter = (nqb > qsub) ? (nqb - qsub + qg - 1) >> qg_log2 : 0;   // query groups in which this qsub is valid
    const int64_t nunits = (int64_t)nq_iter * ntiles;                            // (group, tile) units of this qsub
    // (measurement, ablation builds, no-filter variants only: VAR & 1024 = TILE-MAJOR unit order — stream s takes the row tiles s, s + nslices,
    // ... and multiplies each with ALL its query groups back to back, so that a row tile comes from HBM once and from the L2 for the other
    // groups; thresholds / cursors are not handled: the bound of what that order could give)
    constexpr bool TM = (VAR & 1024) != 0;
    const int tm_tiles = (ntiles > slice) ? (ntiles - slice + nslices - 1) / nslices : 0;
    const int ntile_total = TM ? tm_tiles * nq_iter : (nunits > slice) ? (int)((nunits - slice + nslices - 1) / nslices) : 0;
    if (ntile_total == 0) return;
    // (measurement, ablation builds: static wave priority — waves w and w + 4 share a SIMD and the second-dispatched half loses every VALU
    // arbitration; VAR & 2 raises waves 4..7, VAR & 4 waves 0..3)
    if ((VAR & 2) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if ((VAR & 4) && wave < 4) __builtin_amdgcn_s_setprio(1);
    const int64_t S = (int64_t)ntile_total * nk;
    const int g0 = TM ? 0 : slice / ntiles, t0 = TM ? slice : slice % ntiles;    // first unit of this stream

    // ---- load cursor -------------------------------------------------------------------------------------------------------
    // staging instruction j of wave w fills LDS bytes [(j*8+w)*1024, +1024) = slab rows (j*8+w)*16 .. +16; lane i lands on row
    // (i >> 2), physical chunk (i & 3) and therefore fetches logical chunk (i & 3) ^ ((row >> 1) & 3).  One per-lane offset serves
    // every staging instruction of both operands (their row strides are equal): the row-group step of instruction j (j * 8 groups
    // of 16 rows) and the slab (KiB block) ride in the scalar offset
    const int st_col = ((lane & 3) ^ ((lane >> 3) & 3)) << 4;
    const int vo = wave * 16 * (int)ldx_b + (lane >> 2) * 64 + st_col;
    const int jstep = 8 * 16 * (int)ldx_b;
    // Row tile t of the launch sits at physical tile p(t) = ((so.base + t) * so.mul) mod so.mod of the panel that starts at row0
    // (kernels.h: ScanOrder).  Sequential scan: mul = 1, mod = 2^30, p(t) = t.  Scrambled scan (api.hip): the launch covers tiles
    // [base, base + ntiles) of a pseudo-random order of ALL row tiles of the index, so that the rows seen so far are a fair sample of
    // the index whatever order it is stored in.  Both cursors carry p along with t: t advances by nslices (p by da = nslices * mul mod
    // m) and wraps at ntiles (p by dn = ntiles * mul mod m) — two scalar additions per unit.
    const int p0 = (int)(((int64_t)(so.base + t0) * so.mul) % so.mod);
    int l_q = g0, l_t = t0, l_p = p0, l_k = 0;
    RingSrc sa, sb;
    // (measurement, ablation builds: VAR & 64 aliases the row tiles onto 32 tiles — the row stream comes from the Infinity Cache instead of
    // HBM —, VAR & 512 onto ONE tile per XCD — rows and query panels L2-resident: the bound of any locality work)
    auto phys = [&](int p) { return (VAR & 512) ? xcd : (VAR & 64) ? (p & 31) : p; };
    sa.rsrc = ring_make_rsrc_n(X16 + (row0 + (int64_t)phys(l_p) * Geo::kBM) * ldx_b, Geo::kBM * ldx_b);
    sb.rsrc = ring_make_rsrc_n(Q16 + (int64_t)(qsub + l_q * qg) * kRBN * ldq_b, kRBN * ldq_b);
    int64_t issued = 0;
    const int krot = (VAR & 32) ? (nsub * nk) / nstream : 0;
    // one slab = kLoads pieces per wave (3 of the row panel, 2 of the query panel); issue() sends them into the stage of slab
    // `issued` as a burst and moves the cursor on
    auto issue = [&]() {
        char* st = smem + (int)(issued & 3) * Geo::kStage;
        // (VAR & 32, measurement: the row streams of an XCD walk K from different starting slabs — stream nsub starts at slab
        // nsub * nk / nstream —, so that a query panel's lines are touched by one of its streams every quarter unit instead of by all of
        // them at once: LRU distance 1.4 MB instead of 5.5 MB, the panels stay in the 4 MiB L2.  The sum over K is rotated, not changed.)
        int kslab = l_k;
        if (VAR & 32) {
            kslab += krot;
            if (kslab >= nk) kslab -= nk;
        }
        const int k0b = kslab * 1024;   // slab kslab of a 16-row group = its kslab-th KiB block
#pragma unroll
        for (int j = 0; j < Geo::kLoads; ++j) {
            if (j < Geo::kALoads)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(sa.rsrc, (rg_lptr_t)(st + (j * 8 + wave) * 1024), 16, vo, k0b + j * jstep, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(sb.rsrc, (rg_lptr_t)(st + Geo::kAOpBytes + ((j - Geo::kALoads) * 8 + wave) * 1024),
                                                         16, vo, k0b + (j - Geo::kALoads) * jstep, 0, 0);
        }
        ++issued;
        if (issued < S) {
            if (TM) {
                if (++l_k == nk) {   // next unit: the next query group on the same row tile, then the stream's next tile
                    l_k = 0;
                    if (++l_q == nq_iter) {
                        l_q = 0;
                        l_t += nslices;
                        l_p = l_t;
                        sa.rsrc = ring_make_rsrc_n(X16 + (row0 + (int64_t)phys(l_p) * Geo::kBM) * ldx_b, Geo::kBM * ldx_b);
                    }
                    sb.rsrc = ring_make_rsrc_n(Q16 + (int64_t)(qsub + l_q * qg) * kRBN * ldq_b, kRBN * ldq_b);
                }
            } else if (++l_k == nk) {   // next unit of this stream
                l_k = 0;
                l_t += nslices;
                l_p += so.da;
                if (l_p >= so.mod) l_p -= so.mod;
                if (l_t >= ntiles) {
                    do {
                        l_t -= ntiles;
                        l_p -= so.dn;
                        if (l_p < 0) l_p += so.mod;
                        ++l_q;
                    } while (l_t >= ntiles);
                    sb.rsrc = ring_make_rsrc_n(Q16 + (int64_t)(qsub + l_q * qg) * kRBN * ldq_b, kRBN * ldq_b);
                }
                sa.rsrc = ring_make_rsrc_n(X16 + (row0 + (int64_t)phys(l_p) * Geo::kBM) * ldx_b, Geo::kBM * ldx_b);
            }
        }
    };
    issue();
    issue();
    issue();
    issue();
    wait_vmcnt<3 * Geo::kLoads>();
    __builtin_amdgcn_s_barrier();

    // ---- compute side --------------------------------------------------------------------------------------------------------
    // The thresholds of the wave's 128 query columns live PACKED in two registers (a lane owns eight columns, but its column's four
    // lanes would hold four copies of each): lane (g4 = lane >> 4, c = lane & 15) keeps tau of column block g4 in tq[0] and of column
    // block 4 + g4 in tq[1]; the filter unpacks the four thresholds of a half with ds_bpermute once per tile.  (Eight threshold
    // registers per lane were spilled around the slab loop, and a scratch reload is an s_waitcnt vmcnt(0) that drains the LDS-DMA ring.)
    float tq[2];
    uint32_t curp[2] = {0u, 0u};                                      // eight 8-bit sub-pool cursors (one per column block)
    uint32_t pbase_u = 0;                                             // (uniform) record base of query column 0 of the wave, its sub-pool included
    const uint32_t pstep = 16u * kPoolCap * kPoolPlanes * (uint32_t)nsubs;   // ... the next column block is 16 queries further
    f32x4 acc[kT16RowBlocks][kT16ColBlocks];
    constexpr int kARing = 2;      // A fragment ring (row block i uses a[i % kARing], refilled with block i + kARing after its eight MFMAs)
    bf16x8_t a[kARing], b[kT16ColBlocks];
    const int frow = lane & 15;
    const int foff = frow * 64 + (((lane >> 4) ^ ((frow >> 1) & 3)) << 4);   // lane's 16 bytes inside a 1-KiB fragment block
    {
        const char* a_w = smem + wm * (96 * 64) + foff;
        const char* b_w = smem + Geo::kAOpBytes + wn * (128 * 64) + foff;
#pragma unroll
        for (int i = 0; i < kARing; ++i) a[i] = *(const bf16x8_t*)(a_w + i * 1024);
        (void)b_w;   // (the first slab of every tile fetches its B fragments itself, see slab())
    }
    int64_t s = 0;
    // MODE 0: a slab inside a tile.  MODE 1: the first slab of a tile (its MFMAs take C = 0 instead of a cleared accumulator; it fetches its
    // own eight B fragments first).  MODE 2: the last slab of a tile: it does NOT prefetch the next slab's B fragments, so that their 32
    // registers are free during the tile's filter — with them live the filter's rare path spilled, and a scratch reload is an
    // s_waitcnt vmcnt(0) that drains the LDS-DMA ring (the price: one exposed LDS round trip per tile, ~150 cycles of 40 000).
    // (Earlier forms — the filter fused into the next tile's first slab, 12 x 4 wave tiles with double-buffered B fragments — are in
    // the history: commits 7a1f30d, 10d3493.)
    auto slab = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const char* st0 = smem + (int)(s & 3) * Geo::kStage;
        const char* st1 = smem + (int)((s + 1) & 3) * Geo::kStage;
        const char* a_cur = st0 + wm * (96 * 64) + foff;
        const char* a_nxt = st1 + wm * (96 * 64) + foff;
        const char* b_nxt = st1 + Geo::kAOpBytes + wn * (128 * 64) + foff;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 1) {
            const char* b_cur = st0 + Geo::kAOpBytes + wn * (128 * 64) + foff;
#pragma unroll
            for (int j = 0; j < kT16ColBlocks; ++j) b[j] = *(const bf16x8_t*)(b_cur + j * 1024);
        }
#pragma unroll
        for (int i = 0; i < kT16RowBlocks; ++i) {
            if (i == kT16RowBlocks - kARing) {
                // every fragment of stage s is in registers or consumed (its stage may be refilled), slab s + 1 must have landed: its
                // first fragment is read right below
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of slab s are complete (WAR on its stage)
                wait_vmcnt<2 * Geo::kLoads>();                       // slab s+1 has landed (this thread's part) ...
                __builtin_amdgcn_s_barrier();                        // ... and everybody else's
            }
#pragma unroll
            for (int j = 0; j < kT16ColBlocks; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i % kARing], b[j], MODE == 1 ? z : acc[i][j], 0, 0, 0);
                if (i == kT16RowBlocks - 1 && MODE != 2) b[j] = *(const bf16x8_t*)(b_nxt + j * 1024);   // in place: next slab's fragment j
            }
            a[i % kARing] = *(const bf16x8_t*)(i + kARing < kT16RowBlocks ? a_cur + (i + kARing) * 1024 : a_nxt + (i + kARing - kT16RowBlocks) * 1024);
            if (!(VAR & 128)) __builtin_amdgcn_sched_barrier(0);   // program order pinned after every row block (VAR & 128: left to the scheduler)
        }
        __builtin_amdgcn_sched_barrier(0);
        ++s;
        issue();   // the stage this slab vacated at its barrier takes the slab four ahead
        __builtin_amdgcn_sched_barrier(0);
    };
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;

    // the counters' query index is recomputed at store time (registers are scarce across the tile loop); a lane stores ITS byte of the
    // sub-pool's counter word
    auto store_counts = [&](int g) {
#pragma unroll
        for (int j = 0; j < kT16ColBlocks; ++j) {
            const uint32_t ln = lane_now();
            const int64_t qi = (int64_t)(qsub + g * qg) * kRBN + wn * 128 + j * 16 + (int)(ln & 15u);
            ((unsigned char*)pool_cnt)[(qi * nsubs + slice * kPoolSubsPerSlice + wm) * 4 + (ln >> 4)] =
                (unsigned char)((curp[j >> 2] >> (8 * (j & 3))) & 255u);
        }
    };
    // Row CHUNKS (launch_score_filter): a long scan is issued as several launches of this kernel over consecutive row chunks, all on the same
    // thresholds and with ONE pool select after the last — every chunk is multiplied with all query groups while its rows are still in the
    // Infinity Cache.  The sub-pool cursors then run on from launch to launch: every lane keeps its eight cursor bytes in a slot of its own,
    // cur_save[group][workgroup][thread], written when a group is left and read back (later_chunk) when it is entered.
    auto setup_group = [&](int g) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t ln = lane_now();
            const int64_t qi = (int64_t)(qsub + g * qg) * kRBN + wn * 128 + (4 * h + (int)(ln >> 4)) * 16 + (int)(ln & 15u);
            tq[h] = (VAR & 16) ? INFINITY : ring_launder(tau_g[qi]);
        }
        pbase_u = (uint32_t)(((int64_t)(qsub + g * qg) * kRBN + wn * 128) * kPoolCap * kPoolPlanes * nsubs + slice * kPoolSubsPerSlice + wm);
        curp[0] = curp[1] = 0;
        if (later_chunk) {   // (uniform)
            const uint32_t* cs = (const uint32_t*)(cur_save + ((size_t)g * 256 + blockIdx.x) * kRingThreads) + 2 * (lane_now() + 64u * (uint32_t)wave);
            curp[0] = __builtin_bit_cast(uint32_t, ring_launder(__builtin_bit_cast(float, cs[0])));
            curp[1] = __builtin_bit_cast(uint32_t, ring_launder(__builtin_bit_cast(float, cs[1])));
        }
    };
    auto save_cursors = [&](int g) {
        if (cur_save) cur_save[((size_t)g * 256 + blockIdx.x) * kRingThreads + lane_now() + 64u * (uint32_t)wave] = make_uint2(curp[0], curp[1]);
    };
    int c_q = g0, c_t = t0, c_p = p0, cur_q = g0;
    setup_group(c_q);
    slab(M1{});
#pragma unroll 1
    for (int jt = 0; jt < ntile_total; ++jt) {
#pragma unroll 1
        for (int kk = 1; kk + 1 < nk; ++kk) slab(M0{});
        slab(M2{});    // (nk >= 2: the row stride is a multiple of 64 elements)
        // tile jt is complete in acc
        const int64_t trow = row0 + (int64_t)c_p * Geo::kBM;
        const int32_t row_w = (int32_t)trow + wm * 96;    // (uniform) first row of the wave's block of the tile
        c_t += nslices;
        c_p += so.da;
        if (c_p >= so.mod) c_p -= so.mod;
        while (c_t >= ntiles) {
            c_t -= ntiles;
            c_p -= so.dn;
            if (c_p < 0) c_p += so.mod;
            ++c_q;
        }
        const bool more = jt + 1 < ntile_total;
        if (!(VAR & 1)) {
            // the filter of the finished tile, on its own: all eight waves run it at the same time and the matrix pipe idles meanwhile
            // (~0.6 us per tile without admissions).  Hiding it in the next tile's first slab was not faster (profiles/r03_t16_standalone.txt).
            filter_hazard_cover();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float tau[kFiltCols];
                const int bp = (int)((lane_now() & 15u) << 2);     // byte address of lane c (lane group 0) for ds_bpermute
#pragma unroll
                for (int jj = 0; jj < kFiltCols; ++jj)
                    tau[jj] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bp + jj * 64, __builtin_bit_cast(int, tq[h])));
#pragma unroll
                for (int p = 0; p < kT16RowBlocks / 2; ++p)
                    filter_pair<(VAR & 8) ? 1 : (VAR & 256) ? 2 : 0>(&acc[2 * p][4 * h], &acc[2 * p + 1][4 * h], p, 4 * h, tau, curp[h], pbase_u, pstep,
                                                (uint32_t)nsubs, pool, row_w);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if (!more) {
#pragma unroll
            for (int i = 0; i < kT16RowBlocks; ++i)
#pragma unroll
                for (int j = 0; j < kT16ColBlocks; ++j) asm volatile("" ::"v"(acc[i][j]));
        }
        if (more) {
            slab(M1{});
            if (c_q != cur_q) {             // (uniform) the stream moves on to the next query group
                store_counts(cur_q);
                save_cursors(cur_q);
                cur_q = c_q;
                setup_group(c_q);
            }
        }
    }
    store_counts(cur_q);
    save_cursors(cur_q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing dummy loads must land before the LDS is released
}

// index rows per fused tile (the host sizes launches and the row padding of the index with it)
int fused_tile_rows() { return RingGeom<6>::kBM; }

// row tiles per chunk of a long sequential scan: 256 tiles = 98 304 rows = 151 MB of bf16 rows at D = 768 — what the 256 MB Infinity Cache keeps
// next to the query panels (tools/mall_probe.hip: a re-read window holds 7.7 TB/s up to 256 MB, 6.1 beyond; tools/maxlen_sweep.sh: launches
// capped at 98 304 rows run 2-3 % faster per row).  A multiple of every nslices (32 .. 256): the chunks deal their units evenly.
constexpr int64_t kChunkTiles = 256;

// query blocks an XCD works on concurrently for a batch of nqb query blocks (power of two <= 8); the fused launch then
// has 256 / qg row slices and kPoolSubsPerSlice * 256 / qg sub-pools per query
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
    // Every XCD works on qg query blocks at a time, so a batch occupies ceil(nqb / qg) * qg block slots: 12 blocks (3000 queries) in
    // groups of 8 leave a quarter of the workgroups idle in the second group, in groups of 4 none.  Smaller groups cost a little
    // themselves (each row tile is shared by fewer workgroups of the XCD: more HBM traffic; more row slices: more sub-pools for the
    // pool select to walk): 3 / 8.5 / 22 % of a search for 4 / 2 / 1 (tools/qg_penalty.py at 8 and 16 blocks, where every width divides the
    // batch: profiles/r04_qg_penalty.txt).
    int best = 1;
    double best_cost = 1e30;
    for (int qg = 8; qg >= 1; qg >>= 1) {
        if (qg > nqb && qg > 1) continue;
        const double slots = (double)((nqb + qg - 1) / qg * qg);
        const double cost = slots * (qg == 8 ? 1.0 : qg == 4 ? 1.03 : qg == 2 ? 1.085 : 1.22);
        if (cost < best_cost) {
            best_cost = cost;
            best = qg;
        }
    }
    return best;
}

// bytes of the cursor save area of a batch of nq_pad queries: one uint2 per thread, workgroup and query group (score_filter_t16_kernel)
size_t fused_cursor_save_bytes(int64_t nq_pad) {
    const int64_t qg = fused_query_group(nq_pad);
    const int64_t ngroups = (nq_pad / kRBN + qg - 1) / qg;
    return (size_t)ngroups * 256 * kRingThreads * sizeof(uint2);
}

// mul coprime to mod, close to mod / golden ratio: consecutive tiles of the scan order land far apart
int scan_order_multiplier(int64_t mod) {
    if (mod <= 2) return 1;
    auto gcd = [](int64_t a, int64_t b) {
        while (b) {
            const int64_t t = a % b;
            a = b;
            b = t;
        }
        return a;
    };
    int64_t a = (int64_t)((double)mod * 0.6180339887498949);
    if (a < 1) a = 1;
    while (gcd(a, mod) != 1) ++a;
    return (int)(a % mod);
}

int launch_score_filter(const void* x16, int64_t ldx_elems, int64_t row0, int64_t nrows, const void* q16,
                        int64_t ldq_elems, int64_t nq_pad, int dpad, const float* tau, uint4* pool, int32_t* pool_cnt,
                        hipStream_t st, int64_t scramble_tiles, int64_t scramble_base, hipEvent_t ev_a, hipEvent_t ev_b, void* cur_save) {
    if (nrows <= 0 || nq_pad <= 0) return LDOT_OK;
    LDOT_REQUIRE(ldx_elems == ldq_elems, LDOT_EINVAL, "index and query shadows must have the same row stride");
    auto rk = score_filter_t16_kernel<0>;
#ifdef LDOT_ABLATION
    // Ablation builds only (python -m lightningdot_amd.build --ablation -> libldot_ablation.so; tools/ab.sh): LDOT_DEBUG_VARIANT selects
    // a profiling variant of the kernel.  Results are meaningless under most of them, so the product library does not contain this hook.
    //   16 tau = +inf (filter fast path only), 17 no filter at all, 8 no record stores, 128 program order not pinned
    static int variant = -1;
    if (variant < 0) {
        const char* e = getenv("LDOT_DEBUG_VARIANT");
        variant = e ? atoi(e) : 0;
    }
    if (variant == 16) rk = score_filter_t16_kernel<16>;
    if (variant == 17) rk = score_filter_t16_kernel<17>;
    if (variant == 8) rk = score_filter_t16_kernel<8>;
    if (variant == 256) rk = score_filter_t16_kernel<256>;    // every record store issued twice (what the stores themselves cost)
    if (variant == 32) rk = score_filter_t16_kernel<32>;      // K walk rotated per row stream (query panels L2-resident?)
    if (variant == 48) rk = score_filter_t16_kernel<48>;      // ... with tau = +inf
    if (variant == 2) rk = score_filter_t16_kernel<2>;        // s_setprio 1 for waves 4..7
    if (variant == 4) rk = score_filter_t16_kernel<4>;        // s_setprio 1 for waves 0..3
    if (variant == 18) rk = score_filter_t16_kernel<18>;      // ... with tau = +inf
    if (variant == 20) rk = score_filter_t16_kernel<20>;
    if (variant == 1041) rk = score_filter_t16_kernel<1041>;  // no filter, tile-major unit order (sequential scan only)
    if (variant == 81) rk = score_filter_t16_kernel<81>;      // no filter, row tiles aliased onto 32 tiles (rows from the Infinity Cache)
    if (variant == 529) rk = score_filter_t16_kernel<529>;    // no filter, one row tile per XCD (everything L2-resident)
    if (variant == 128) rk = score_filter_t16_kernel<128>;    // row blocks NOT pinned in program order (the scheduler sinks the fragment loads)
    LDOT_HIP_CHECK(hipFuncSetAttribute((const void*)rk, hipFuncAttributeMaxDynamicSharedMemorySize, RingGeom<6>::kLds));
#else
    static bool attr_set[kAttrDevices];
    LDOT_HIP_CHECK(set_max_dynamic_lds_once((const void*)rk, RingGeom<6>::kLds, attr_set));
#endif
    const int qg = fused_query_group(nq_pad);
    const int qg_log2 = qg == 8 ? 3 : qg == 4 ? 2 : qg == 2 ? 1 : 0;
    // sequential: tile t of the launch = tile t of [row0, row0 + nrows); scrambled (scramble_tiles = the row tiles of the whole
    // index, row0 = 0): tile t of the launch = tile ((scramble_base + t) * mul) mod scramble_tiles of the index
    const int64_t ntiles = (nrows + RingGeom<6>::kBM - 1) / RingGeom<6>::kBM, nslices = 256 / qg;
    ScanOrder so;
    if (scramble_tiles > 0) {
        LDOT_REQUIRE(row0 == 0 && scramble_tiles < ((int64_t)1 << 30) && scramble_base + ntiles <= scramble_tiles, LDOT_EINVAL,
                     "bad scrambled launch");
        so.mul = scan_order_multiplier(scramble_tiles);
        so.mod = (int)scramble_tiles;
        so.base = (int)scramble_base;
        so.da = (int)((nslices * (int64_t)so.mul) % scramble_tiles);
        so.dn = (int)((ntiles * (int64_t)so.mul) % scramble_tiles);
    } else {
        LDOT_REQUIRE(ntiles + 256 < ((int64_t)1 << 30), LDOT_EINVAL, "too many rows for one launch");
        so.mul = 1;
        so.mod = 1 << 30;
        so.base = 0;
        so.da = (int)nslices;
        so.dn = (int)ntiles;
    }
    // (ev_a / ev_b, LDOT_OPT_PROFILE: events recorded around the launch.  Attaching them to the dispatch itself — hipExtLaunchKernel — was
    // measured in round 5: the dispatch gaps stay, and the queue stays in its profiling mode afterwards, which slowed LATER searches of
    // other indexes in the same process up to 3x: profiles/r05_secondary_probe.txt)
    // Row chunks: a sequential scan of more than kChunkTiles tiles for more than one query group goes out as one launch per chunk (same
    // thresholds, cursors carried in cur_save, ONE pool select after the last): the chunk's rows are read from HBM for the first query group
    // and from the Infinity Cache for the others.  (One query group reads every row once anyway.)
    // (kChunkTiles is the chunk at D = 768; other row lengths — and the split-bf16 shadow, three times as long — get the same ~151 MB: a multiple of
    // the row slices, at least one tile per slice)
    int64_t chunk = kChunkTiles * 768 / std::max<int64_t>(ldx_elems, 1) / nslices * nslices;
    if (chunk < nslices) chunk = nslices;
#ifdef LDOT_ABLATION
    if (const char* e = getenv("LDOT_DEBUG_CHUNK_TILES")) chunk = atoll(e);
#endif
    const int64_t ngroups = (nq_pad / kRBN + qg - 1) / qg;
    const bool chunked = cur_save && chunk > 0 && chunk % nslices == 0 && chunk < ntiles && ngroups > 1;
    if (ev_a) (void)hipEventRecord(ev_a, st);
    if (!chunked) {
        hipLaunchKernelGGL(rk, dim3(256), dim3(kRingThreads), RingGeom<6>::kLds, st, (const char*)x16, ldx_elems * 2, row0,
                           nrows, (const char*)q16, ldq_elems * 2, (int)(nq_pad / kRBN), dpad / kRBK, tau, pool, pool_cnt,
                           qg_log2, so, (uint2*)nullptr, 0);
    } else {
        const int64_t crows = chunk * RingGeom<6>::kBM;
        for (int64_t r = 0, c = 0; r < nrows; r += crows, ++c) {
            const int64_t len = std::min(crows, nrows - r);
            ScanOrder sc = so;
            const int64_t nt = (len + RingGeom<6>::kBM - 1) / RingGeom<6>::kBM;
            if (scramble_tiles > 0) {   // (scrambled order: the chunk = the next `chunk` tiles of the pseudo-random order, wherever they lie)
                sc.base = so.base + (int)(c * chunk);
                sc.dn = (int)((nt * (int64_t)so.mul) % scramble_tiles);
            } else {
                sc.dn = (int)nt;
            }
            hipLaunchKernelGGL(rk, dim3(256), dim3(kRingThreads), RingGeom<6>::kLds, st, (const char*)x16, ldx_elems * 2,
                               scramble_tiles > 0 ? row0 : row0 + r, len, (const char*)q16, ldq_elems * 2, (int)(nq_pad / kRBN), dpad / kRBK, tau,
                               pool, pool_cnt, qg_log2, sc, (uint2*)cur_save, c > 0 ? 1 : 0);
        }
    }
    if (ev_b) (void)hipEventRecord(ev_b, st);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

}  // namespace ldot
