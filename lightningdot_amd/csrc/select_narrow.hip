// Top-k' of a few (<= 16) very long score rows — the selection behind the narrow scan (score_narrow.hip), replacing faiss' per-query
// heap over the whole index (dvl/indexer/faiss_indexers.py:83 -> IndexFlatIP.search with nq = 1, the demo of dvl/utils.py:204-211).
//
// The streaming selector of select.hip needs ~70 us to learn its threshold on a fresh row however many workgroups share the row.
// Here the threshold comes from the RUN MAXIMA the scan kernel leaves behind (one maximum per run of 16 << run_shift rows, <= 16384 per query):
//
//   tau = the k'-th largest run maximum.   Each run maximum IS the score of a row, so at least k' rows score >= tau, hence the k'-th
//   best row scores >= tau and every row of the exact top-k' satisfies score >= tau — and lies in a run whose maximum is >= tau.
//
// 1. narrow_tau_kernel      one workgroup per query: exact k'-th largest of the run maxima (bit search on the order-preserving keys,
//                           values in registers, one barrier per bit).  A COARSER run gives a lower threshold, yet hardly more
//                           candidates (1M unordered rows, k' = 128: ~135 candidates from 512-row runs), so
//                           runs are sized for <= 2048 maxima per query (8 per thread of a 256-thread workgroup) unless k' is large
// 2. narrow_collect_kernel  visits only the runs with maximum >= tau (~k' of them) and appends their rows with score >= tau to the
//                           query's candidate buffer (for unordered rows ~1.01 k' candidates; rows stored in cluster order give more —
//                           a full buffer is reported and the caller redoes the search with the streaming selector)
// 3. narrow_final_kernel    sorts the candidates (score descending, row ascending) into the query's list, sets the list threshold.
#include <math.h>
#include <stdlib.h>

#include "bitonic.h"
#include "kernels.h"

namespace ldot {

// THREADS x NV >= nruns run maxima in registers
template <int THREADS, int NV>
__global__ __launch_bounds__(THREADS) void narrow_tau_kernel(const uint32_t* __restrict__ M, int64_t ldm, int nruns, int kp,
                                                            uint32_t* __restrict__ tau_key) {
    __shared__ int part[2][THREADS / 64];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (nruns < kp) {   // fewer runs than list slots: every row is a candidate
        if (tid == 0) tau_key[q] = 0xffffffffu;
        return;
    }
    uint32_t key[NV];   // descending keys: smaller key = larger score
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int r = v * THREADS + tid;
        key[v] = r < nruns ? ~M[(int64_t)q * ldm + r] : 0xffffffffu;
    }
    // smallest key value t with #(keys <= t) >= kp, most significant bit first.  The search stops after kTauBits bits and leaves
    // the remaining low bits set: a slightly larger key = a slightly lower threshold (2^-9 relative), still a lower bound of the
    // k'-th best score — a few more candidates for 12 fewer barrier rounds
    constexpr int kTauBits = 20;
    uint32_t res = (1u << (32 - kTauBits)) - 1u;
    for (int bit = 31; bit >= 32 - kTauBits; --bit) {
        const uint32_t test = res | ((1u << bit) - 1u);
        // wave count through ballots: scalar popcounts, no cross-lane shuffles (a shuffle reduction is a chain of 6 LDS-pipe
        // round trips per bit and was 3/4 of this kernel)
        int cnt = 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) cnt += __popcll(__ballot(key[v] <= test));
        int* p = part[bit & 1];
        if (lane == 0) p[wave] = cnt;
        __syncthreads();
        int tot = 0;
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) tot += p[w];
        // (pad keys 0xffffffff only count when test is all ones, i.e. never before the real keys have been exhausted)
        if (tot < kp) res |= 1u << bit;
    }
    if (tid == 0) tau_key[q] = res;
}

// grid (ceil(nruns / 64), nq): a WAVE takes 16 consecutive runs — one load fetches their maxima (and leaves them zero for the next
// scan), then it scans the qualifying ones (about one in ten), 4 loads in flight per lane.  (One run per wave left every wave a
// dependent maximum-then-rows chain and thousands of workgroups to dispatch: 28 us for 64 queries.)
__global__ __launch_bounds__(256) void narrow_collect_kernel(const float* __restrict__ S, int64_t lds_elems, uint32_t* __restrict__ M,
                                                             int64_t ldm, int nruns, int run_rows, int64_t nrows, int64_t row0,
                                                             const uint32_t* __restrict__ tau_key, uint64_t* __restrict__ cand,
                                                             int cap, int32_t* __restrict__ cnt,
                                                             const int32_t* __restrict__ nrows_q, int64_t nrows_q_stride,
                                                             int tiled_qg) {
    const int q = blockIdx.y, lane = threadIdx.x & 63;
    const uint32_t tk = tau_key[q];
    if (nrows_q) nrows = nrows_q[q * nrows_q_stride];   // (per-query column counts: the inverted-file scan)
    // score of column c: row-major S[q][c], or the tiled layout of the narrow scan (1 KiB per 16 queries x 16 rows)
    const float* s_row = tiled_qg ? S + ((q >> 4) * 256 + (q & 15) * 16) : S + (int64_t)q * lds_elems;
    const int64_t tile_stride = (int64_t)tiled_qg * 256;
    uint64_t* c_row = cand + (int64_t)q * cap;
    const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    bool pass = false;
    if (lane < 16 && r0 + lane < nruns) {
        pass = ~M[(int64_t)q * ldm + r0 + lane] <= tk;
        M[(int64_t)q * ldm + r0 + lane] = 0;
    }
    unsigned long long runs = __ballot(pass);
    while (runs) {
        const int r = r0 + __ffsll((long long)runs) - 1;
        runs &= runs - 1;
        const int64_t c0 = (int64_t)r * run_rows;
        const int64_t c1 = c0 + run_rows < nrows ? c0 + run_rows : nrows;
        for (int64_t cb = c0; cb < c1; cb += 256) {   // (wave-uniform trip count: the ballots below are wave-wide)
            float sv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t c = cb + u * 64 + lane;
                sv[u] = c < c1 ? (tiled_qg ? s_row[(c >> 4) * tile_stride + (c & 15)] : s_row[c]) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t c = cb + u * 64 + lane;
                const bool hit = c < c1 && desc_key(sv[u]) <= tk;
                const unsigned long long hm = __ballot(hit);
                if (hm) {
                    const int leader = __ffsll((long long)hm) - 1;
                    int base = 0;
                    if (lane == leader) base = atomicAdd(cnt + q * kNarrowCntStride, __popcll(hm));
                    base = __shfl(base, leader);
                    const int pos = base + __popcll(hm & ((1ull << lane) - 1ull));
                    if (hit && pos < cap) c_row[pos] = ((uint64_t)desc_key(sv[u]) << 32) | (uint32_t)(row0 + c);
                }
            }
        }
    }
}

// one workgroup per query: candidates -> sorted list [kp] (score desc, row asc; empty slots: pad score, row -1), list threshold
__global__ __launch_bounds__(256) void narrow_final_kernel(const uint64_t* __restrict__ cand, int cap, int32_t* __restrict__ cnt,
                                                           float* __restrict__ list_s, int32_t* __restrict__ list_i, int kp,
                                                           float* __restrict__ tau, int32_t* __restrict__ over) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    const int q = blockIdx.x;
    int n = cnt[q * kNarrowCntStride];
    __syncthreads();
    if (threadIdx.x == 0) cnt[q * kNarrowCntStride] = 0;   // ready for the next search
    if (threadIdx.x == 0) over[q] = n > cap ? 1 : 0;   // (a full buffer makes the list unusable: the caller redoes the search)
    if (n > cap) n = cap;
    int P = 2;
    while (P < n) P <<= 1;
    for (int i = threadIdx.x; i < P; i += 256) keys[i] = i < n ? cand[(int64_t)q * cap + i] : ~0ull;
    __syncthreads();
    bitonic_sort_lds(keys, P);
    for (int i = threadIdx.x; i < kp; i += 256) {
        const bool have = i < n;
        const uint64_t k = have ? keys[i] : 0;
        list_s[(int64_t)q * kp + i] = have ? desc_key_to_float((uint32_t)(k >> 32)) : LDOT_PAD_SCORE;
        list_i[(int64_t)q * kp + i] = have ? (int32_t)(uint32_t)k : -1;
    }
    if (threadIdx.x == 0) tau[q] = n >= kp ? desc_key_to_float((uint32_t)(keys[kp - 1] >> 32)) : -INFINITY;
}


// ---- <= 16 queries, one scan chunk: everything after the scan in ONE launch ---------------------------------------------------------
// threshold + collect + top-k' + exact fp32 re-score + final order + output (+ list and threshold for callers that want them), one
// 1024-thread workgroup per query.  The four-kernel chain above costs 38 us on the GPU for one query (8 + 5 + 9 + 16) and three launch
// gaps — more than the scan of a 123 287-row index (37 us); here:
//   * the threshold search runs in ONE wave (<= 2048 run maxima = 32 keys per lane): ballots + scalar popcounts, no barrier per bit;
//   * candidates are collected into LDS by all 16 waves (the qualifying runs are few: ~k' of <= 2048);
//   * the k' best candidates and the final order come from RANK COUNTING (every thread counts the keys that beat its own — n is a
//     few hundred) instead of bitonic stages with barriers;
//   * the row gather of the re-score has 16 waves x 4 rows in flight; its arithmetic is the re-score kernel's (4 fmaf chains over the
//     columns lane*4 + 256*i, pairwise sum, xor-shuffle tree), so the scores are bit-identical to the other search paths'.
// over[q] = 1: more candidates than kFinishCap (rows stored in cluster order) — the caller redoes the search (as for narrow_final_kernel).
constexpr int kFinishCap = 4096;      // candidate keys in LDS (32 KiB)
constexpr int kFinishThreads = 1024;

// wave-wide integer sum in 6 DPP adds + one readlane (quad swaps, row mirrors, row broadcasts: the total lands in lane 63): pure VALU,
// ~50 cycles — six ds_bpermute round trips (__shfl_xor) are ~400
__device__ __forceinline__ int finish_wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);   // row_mirror  -> every lane: its row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// the same network for AND / OR (identity element `id` for the lanes a row broadcast does not reach)
template <bool IS_AND>
__device__ __forceinline__ uint32_t finish_wave_bits(uint32_t v) {
    const int id = IS_AND ? -1 : 0;
    auto op = [](uint32_t a, uint32_t b) { return IS_AND ? (a & b) : (a | b); };
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0xB1, 0xF, 0xF, false));
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x4E, 0xF, 0xF, false));
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x141, 0xF, 0xF, false));
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x140, 0xF, 0xF, false));
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x142, 0xA, 0xF, false));
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x143, 0xC, 0xF, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// largest j with cs[j] <= col (cs ascending, cs[0] = 0, n + 1 entries) — the probed list a column of the inverted-file scan belongs to
__device__ __forceinline__ int finish_find_probe(const int32_t* cs, int n, int col) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cs[mid] <= col)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kFinishThreads) void narrow_finish_kernel(
    const float* __restrict__ S, int tiled_qg, int64_t lds_elems, uint32_t* __restrict__ M, int64_t ldm, int nruns, int run_rows,
    int64_t nrows, const float* __restrict__ q32, int64_t ldq, const float* __restrict__ x32, int64_t ldx, int dpad, int kp, int k,
    int do_rescore, float* __restrict__ list_s, int32_t* __restrict__ list_i, float* __restrict__ tau_out, float* __restrict__ out_s,
    int64_t* __restrict__ out_l, int32_t* __restrict__ over, const int32_t* __restrict__ nrows_q, int64_t nrows_q_stride,
    const int64_t* __restrict__ rowbase, const int32_t* __restrict__ cstart, int nprobe, int dbg_phase) {
    __shared__ __attribute__((aligned(16))) uint64_t cand[kFinishCap];
    __shared__ __attribute__((aligned(16))) uint64_t best[1024];       // the k' best candidates, then their exact keys (kp <= 512)
    __shared__ uint32_t tk_sh;
    __shared__ int n_sh, nrun_sh;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* Mq = M + (int64_t)q * ldm;
    if (nrows_q) nrows = nrows_q[q * nrows_q_stride];   // (per-query column counts: the inverted-file scan)
#ifdef LDOT_ABLATION
    if (dbg_phase == 9) return;                         // (profiling: an empty launch)
#endif
    // ---- 1. threshold key: k'-th largest run maximum (wave 0; the other waves wait at the barrier) ------------------------------
    if (tid == 0) n_sh = nrun_sh = 0;
    int32_t* runs = (int32_t*)best;                     // (the run list lives in the best[] buffer until step 3 needs it: 2048 ints)
    if (wave == 0) {
        uint32_t res = 0xffffffffu;                     // fewer runs than list slots: every row is a candidate
        uint32_t key[32];
#pragma unroll
        for (int v = 0; v < 32; ++v) {
            const int r = v * 64 + lane;
            key[v] = r < nruns ? ~Mq[r] : 0xffffffffu;
        }
#ifdef LDOT_ABLATION
        if (dbg_phase == 8) {                           // (profiling: the key loads only)
            uint32_t x = 0;
#pragma unroll
            for (int v = 0; v < 32; ++v) x ^= key[v];
            if (x == 0x12345u) tk_sh = x;
            nruns = 0;
        }
#endif
        if (nruns >= kp) {
            // per-lane counts on the VALU + ONE DPP wave reduction per bit (a ballot + scalar popcount per key costs ~40 cycles of
            // VALU -> SGPR -> SALU latency: 640 of them were 13 us; six ds_bpermute round trips per bit are no better).  The bits every
            // run maximum shares (sign, exponent, ... : typically the top 9-10) are not searched at all.
            constexpr int kTauBits = 20;                // (as narrow_tau_kernel: the low 12 bits only lower the threshold by 2^-9 relative)
            constexpr uint32_t kLow = (1u << (32 - kTauBits)) - 1u;
            uint32_t all_and = 0xffffffffu, all_or = 0u;
#pragma unroll
            for (int v = 0; v < 32; ++v) {
                all_and &= key[v];                      // (pad keys are all ones: neutral here, masked out of the OR)
                all_or |= (v * 64 + lane < nruns) ? key[v] : 0u;
            }
            all_and = finish_wave_bits<true>(all_and);
            all_or = finish_wave_bits<false>(all_or);
            const uint32_t diff = all_and ^ all_or;
            const int hb = diff ? 31 - __clz((int)diff) : -1;     // highest bit in which two run maxima differ
            res = (hb >= 31 ? 0u : (all_or & ~((1u << (hb + 1)) - 1u))) | kLow;   // the shared prefix; low bits stay set
            for (int bit = hb; bit >= 32 - kTauBits; --bit) {
                const uint32_t test = res | ((1u << bit) - 1u);
                int cnt = 0;
#pragma unroll
                for (int v = 0; v < 32; ++v) cnt += key[v] <= test ? 1 : 0;
                if (finish_wave_sum(cnt) < kp) res |= 1u << bit;
            }
        }
        // the qualifying runs (maximum >= threshold, ~k' of them), straight from the registers
        int base = 0;
#pragma unroll
        for (int v = 0; v < 32; ++v) {
            const int r = v * 64 + lane;
            const bool pass = r < nruns && key[v] <= res;
            const unsigned long long pm = __ballot(pass);
            if (pass) runs[base + __popcll(pm & ((1ull << lane) - 1ull))] = r;
            base += __popcll(pm);
        }
        if (lane == 0) {
            tk_sh = res;
            nrun_sh = base;
        }
    }
    __syncthreads();
    const uint32_t tk = tk_sh;
    for (int r = tid; r < nruns; r += kFinishThreads) Mq[r] = 0;   // (left zero for the next scan; wave 0 has read them all)
#ifdef LDOT_ABLATION
    if (dbg_phase == 1) return;   // (profiling: the kernel cut short after its n-th phase, LDOT_DEBUG_FINISH_PHASE)
#endif
    // ---- 2. collect (one run after the other would leave every wave a chain of dependent load round trips: 8 x ~1 us) ---------------
    const float* s_row = tiled_qg ? S + ((q >> 4) * 256 + (q & 15) * 16) : S + (int64_t)q * lds_elems;
    const int64_t tile_stride = (int64_t)tiled_qg * 256;
    const int nqual = nrun_sh;
#ifdef LDOT_ABLATION
    if (dbg_phase == 2) return;
#endif
    // sub-item = 64 consecutive rows of a qualifying run = 16 lanes x f32x4 (rows 4l .. 4l+3 of a 16-row tile are contiguous in both score
    // layouts); one wave-load covers FOUR sub-items, CU wave-loads are in flight per lane
    const int per_shift = run_rows >= 64 ? __ffs(run_rows >> 6) - 1 : 0;     // (run_rows is 16 << s: sub-items per run = 2^per_shift)
    const int nsub = nqual << per_shift;
    constexpr int CU = 8;
    const int grp = lane >> 4, l16 = lane & 15;
    for (int i0 = wave * (4 * CU); i0 < nsub; i0 += (kFinishThreads / 64) * (4 * CU)) {
        f32x4 sv[CU];
        int64_t cc[CU];
        int lim[CU];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int si = i0 + u * 4 + grp;
            int64_t c = -1;
            int valid = 0;
            if (si < nsub) {
                const int ri = si >> per_shift, sb = si & ((1 << per_shift) - 1);
                const int64_t c0 = (int64_t)runs[ri] * run_rows;
                const int64_t c1 = c0 + run_rows < nrows ? c0 + run_rows : nrows;
                c = c0 + sb * 64 + l16 * 4;
                valid = c < c1 ? (int)(c1 - c < 4 ? c1 - c : 4) : 0;
            }
            cc[u] = c;
            lim[u] = valid;
            // (a partly valid quad still lies inside the padded score row: nrows_pad is a multiple of 16)
            sv[u] = valid ? *(const f32x4*)(tiled_qg ? s_row + (c >> 4) * tile_stride + (c & 15) : s_row + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            bool h[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = j < lim[u] && desc_key(sv[u][j]) <= tk;
            if (__ballot(h[0] | h[1] | h[2] | h[3])) {   // rare: about one hit per qualifying run
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned long long hm = __ballot(h[j]);
                    if (hm) {
                        const int leader = __ffsll((long long)hm) - 1;
                        int base = 0;
                        if (lane == leader) base = atomicAdd(&n_sh, __popcll(hm));
                        base = __shfl(base, leader);
                        const int pos = base + __popcll(hm & ((1ull << lane) - 1ull));
                        if (h[j] && pos < kFinishCap) cand[pos] = ((uint64_t)desc_key(sv[u][j]) << 32) | (uint32_t)(cc[u] + j);
                    }
                }
            }
        }
    }
    __syncthreads();
    int n = n_sh;
#ifdef LDOT_ABLATION
    if (dbg_phase == 3) return;
#endif
    if (tid == 0) over[q] = n > kFinishCap ? 1 : 0;
    // (a full buffer makes the result unusable — the caller redoes the search —, but the list written below must still hold valid rows:
    // the re-score that is already enqueued behind this kernel dereferences them)
    if (n > kFinishCap) n = kFinishCap;
    // ---- 3. the k' best candidates by (bf16-input score desc, row asc): rank counting, eight keys per LDS round trip ---------------
    const int npad = (n + 7) & ~7;
    for (int i = n + tid; i < npad; i += kFinishThreads) cand[i] = ~0ull;
    __syncthreads();                                    // (also: everybody is done with the run list in best[])
    for (int i = tid; i < 1024; i += kFinishThreads) best[i] = ~0ull;
    __syncthreads();
    auto rank_of = [&](const uint64_t mine, const int cnt_pad) {
        int rank = 0;
        for (int j = 0; j < cnt_pad; j += 8) {
            const ulonglong2 a = *(const ulonglong2*)(cand + j), b = *(const ulonglong2*)(cand + j + 2);
            const ulonglong2 c = *(const ulonglong2*)(cand + j + 4), d = *(const ulonglong2*)(cand + j + 6);
            rank += (a.x < mine) + (a.y < mine) + (b.x < mine) + (b.y < mine) + (c.x < mine) + (c.y < mine) + (d.x < mine) + (d.y < mine);
        }
        return rank;                                    // (keys are distinct: the row is part of the key; pad keys ~0 never count)
    };
    for (int i = tid; i < n; i += kFinishThreads) {
        const uint64_t mine = cand[i];
        const int rank = rank_of(mine, npad);
        if (rank < kp) best[rank] = mine;
    }
    __syncthreads();
    const int m = n < kp ? n : kp;
    if (list_s) {                                       // the running-list view of the result (threshold exchange, LDOT_OPT_VERIFY)
        for (int i = tid; i < kp; i += kFinishThreads) {
            const uint64_t kv = best[i];
            list_s[(int64_t)q * kp + i] = i < m ? desc_key_to_float((uint32_t)(kv >> 32)) : LDOT_PAD_SCORE;
            list_i[(int64_t)q * kp + i] = i < m ? (int32_t)(uint32_t)kv : -1;
        }
        if (tid == 0) tau_out[q] = n >= kp ? desc_key_to_float((uint32_t)(best[kp - 1] >> 32)) : -INFINITY;
    }
    if (!out_s) return;
#ifdef LDOT_ABLATION
    if (dbg_phase == 4) return;
#endif
    if (cstart) {                                       // inverted-file scan: column of the query's compact space -> index row
        __syncthreads();
        const int32_t* cq = cstart + (int64_t)q * (nprobe + 1);
        for (int i = tid; i < m; i += kFinishThreads) {
            const uint64_t kv = best[i];
            const int col = (int)(uint32_t)kv;
            const int j = finish_find_probe(cq, nprobe, col);
            best[i] = (kv & 0xffffffff00000000ull) | (uint32_t)(rowbase[(int64_t)q * nprobe + j] + (col - cq[j]));
        }
        __syncthreads();
    }
    // ---- 4. exact fp32 scores of the m candidates (the re-score kernel's arithmetic) ---------------------------------------------
    const float* qrow = q32 + (int64_t)q * ldq;
    constexpr int U = 4;
    for (int e0 = wave * U; e0 < m; e0 += (kFinishThreads / 64) * U) {
        int32_t r[U];
        float sc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t kv = (e0 + u < m) ? best[e0 + u] : ~0ull;
            r[u] = (e0 + u < m) ? (int32_t)(uint32_t)kv : -1;
            sc[u] = desc_key_to_float((uint32_t)(kv >> 32));
        }
        if (do_rescore) {
            float acc[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u][0] = acc[u][1] = acc[u][2] = acc[u][3] = 0.f;
            // blocks of 1024 columns: the (up to) four 256-column chunks of all U rows are loaded before the first fmaf, so a block is
            // ONE memory round trip instead of four (the chains still run over the chunks in column order: same bits as rescore_kernel)
            for (int cb = 0; cb < dpad; cb += 1024) {
                f32x4 qv[4], xv[U][4];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    const int c = cb + ci * 256 + lane * 4;
                    const bool in = c < dpad;
                    qv[ci] = in ? *(const f32x4*)(qrow + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        xv[u][ci] = (in && r[u] >= 0) ? *(const f32x4*)(x32 + (int64_t)r[u] * ldx + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    if (cb + ci * 256 + lane * 4 < dpad) {
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            acc[u][0] = fmaf(xv[u][ci][0], qv[ci][0], acc[u][0]);
                            acc[u][1] = fmaf(xv[u][ci][1], qv[ci][1], acc[u][1]);
                            acc[u][2] = fmaf(xv[u][ci][2], qv[ci][2], acc[u][2]);
                            acc[u][3] = fmaf(xv[u][ci][3], qv[ci][3], acc[u][3]);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                sc[u] = (acc[u][0] + acc[u][1]) + (acc[u][2] + acc[u][3]);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sc[u] += __shfl_xor(sc[u], o);
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (e0 + u < m) cand[e0 + u] = ((uint64_t)desc_key(sc[u]) << 32) | (uint32_t)r[u];
        }
    }
    __syncthreads();
    // ---- 5. final order (exact score desc, row asc) by rank counting, top-k out ------------------------------------------------------
#ifdef LDOT_ABLATION
    if (dbg_phase == 5) return;
#endif
    const int mpad = (m + 7) & ~7;
    for (int i = m + tid; i < mpad; i += kFinishThreads) cand[i] = ~0ull;
    __syncthreads();
    for (int i = tid; i < m; i += kFinishThreads) {
        const uint64_t mine = cand[i];
        const int rank = rank_of(mine, mpad);
        if (rank < k) {
            out_s[(int64_t)q * k + rank] = desc_key_to_float((uint32_t)(mine >> 32));
            out_l[(int64_t)q * k + rank] = (int64_t)(uint32_t)mine;
        }
    }
    for (int e = m + tid; e < k; e += kFinishThreads) {
        out_s[(int64_t)q * k + e] = LDOT_PAD_SCORE;
        out_l[(int64_t)q * k + e] = LDOT_PAD_LABEL;
    }
}

int launch_narrow_tau(const uint32_t* M, int64_t ldm, int nruns, int nq, int kp, uint32_t* tau_key, hipStream_t st) {
    LDOT_REQUIRE(nruns >= 1 && nruns <= kNarrowMaxRuns && nq >= 1, LDOT_EINVAL, "bad run count");
    if (nruns <= 2048)
        hipLaunchKernelGGL((narrow_tau_kernel<256, 8>), dim3(nq), dim3(256), 0, st, M, ldm, nruns, kp, tau_key);
    else if (nruns <= 4096)
        hipLaunchKernelGGL((narrow_tau_kernel<512, 8>), dim3(nq), dim3(512), 0, st, M, ldm, nruns, kp, tau_key);
    else
        hipLaunchKernelGGL((narrow_tau_kernel<1024, kNarrowMaxRuns / 1024>), dim3(nq), dim3(1024), 0, st, M, ldm, nruns, kp, tau_key);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_narrow_collect(const float* S, int64_t lds_elems, uint32_t* M, int64_t ldm, int nruns, int run_rows, int64_t nrows,
                          int64_t row0, int nq, const uint32_t* tau_key, uint64_t* cand, int cap, int32_t* cnt, const int32_t* nrows_q,
                          int64_t nrows_q_stride, int tiled_qg, hipStream_t st) {
    hipLaunchKernelGGL(narrow_collect_kernel, dim3((nruns + 63) / 64, nq), dim3(256), 0, st, S, lds_elems, M, ldm, nruns, run_rows,
                       nrows, row0, tau_key, cand, cap, cnt, nrows_q, nrows_q_stride, tiled_qg);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

static bool g_final_attr[64];

// over [nq]: 1 where the candidate buffer was full (may be device-mapped host memory)
int launch_narrow_final(const uint64_t* cand, int cap, int32_t* cnt, int nq, float* list_s, int32_t* list_i, int kp, float* tau,
                        int32_t* over, hipStream_t st) {
    LDOT_REQUIRE(cap >= 2 && (cap & (cap - 1)) == 0 && cap <= kNarrowCandCap && kp <= cap, LDOT_EINVAL, "bad candidate capacity");
    int dev = 0;
    LDOT_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 64 || !g_final_attr[dev]) {
        LDOT_HIP_CHECK(hipFuncSetAttribute((const void*)narrow_final_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kNarrowCandCap * 8));
        if (dev < 64) g_final_attr[dev] = true;
    }
    hipLaunchKernelGGL(narrow_final_kernel, dim3(nq), dim3(256), (size_t)cap * 8, st, cand, cap, cnt, list_s, list_i, kp, tau, over);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// nruns <= 2048, kp <= 512 (one wave holds the run maxima, the best[] buffer the k' keys).  S / M / nrows / nrows_q as for
// launch_narrow_collect (M is left zero).  list_s / list_i / tau_out (optional): the list view; out_s / out_l (optional,
// device-visible): the final top-k.  rowbase / cstart / nprobe (optional): the inverted-file scan's column -> row translation.
int launch_narrow_finish(const float* S, int tiled_qg, int64_t lds_elems, uint32_t* M, int64_t ldm, int nruns, int run_rows,
                         int64_t nrows, int nq, const float* q32, int64_t ldq, const float* x32, int64_t ldx, int dpad, int kp, int k,
                         int do_rescore, float* list_s, int32_t* list_i, float* tau_out, float* out_s, int64_t* out_l, int32_t* over,
                         const int32_t* nrows_q, int64_t nrows_q_stride, const int64_t* rowbase, const int32_t* cstart, int nprobe,
                         hipStream_t st) {
    LDOT_REQUIRE(nruns >= 1 && nruns <= 2048 && kp <= 512 && k <= kp && nq >= 1, LDOT_EINVAL, "narrow finish: bad sizes");
    int dbg_phase = 0;
#ifdef LDOT_ABLATION
    if (const char* e = getenv("LDOT_DEBUG_FINISH_PHASE")) dbg_phase = atoi(e);
#endif
    hipLaunchKernelGGL(narrow_finish_kernel, dim3(nq), dim3(kFinishThreads), 0, st, S, tiled_qg, lds_elems, M, ldm, nruns, run_rows,
                       nrows, q32, ldq, x32, ldx, dpad, kp, k, do_rescore, list_s, list_i, tau_out, out_s, out_l, over, nrows_q,
                       nrows_q_stride, rowbase, cstart, nprobe, dbg_phase);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

}  // namespace ldot
