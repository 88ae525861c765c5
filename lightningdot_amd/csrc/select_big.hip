// Top-k' selection for LONG candidate lists (k' > 512): the mining searches of dvl/hn.py:53-55 at num_tops up to 1000 (k' = 1280 of the
// 29 000 / 145 000 rows of Flickr30k-train, 4.4 % / 0.9 % of the index per query) and k up to 2048 of faiss' flat search
// (dvl/indexer/faiss_indexers.py:83).  Both sources of the short-list selects: a row of a materialised score chunk (dense path) and the
// candidate sub-pools of the fused filter.
//
// Round 6.  What was there: the dense source held a whole <= 32 768-column row in the registers of ONE 1024-thread workgroup per CU and bit-searched
// it (37 us per query: load, search and write-out serialised, 0.8 TB/s), the pool source fell back to a one-wave LDS bitonic sort (93 us per
// query).  Here a 256-thread workgroup per query (four or two resident per CU)
//   1. keeps the running list (<= k' keys) in registers,
//   2. fixes a PIVOT — pools: the list's own threshold (k'-th best so far) or the threshold the caller passes; dense: additionally the
//      r-th best of a 2048-column strided sample of the row, r = r0 + 5 sqrt(r0) + 2 with r0 = k' x 2048 / columns, so that
//      ~1.5 k' columns pass and fewer than k' with probability ~3e-7 —,
//   3. streams the source ONCE and keeps what passes the pivot in LDS (wave-aggregated appends),
//   4. bit-searches the k'-th best (score desc, row asc) over list + survivors in registers (one compare-and-count pass and one
//      workgroup sum per bit: 4 waves, one barrier) and writes the winners out as a set.
// It is exact whatever the data: if the survivors do not fit (thousands of equal scores) or the sample pivot passed fewer than k'
// elements, the workgroup takes the SLOW path — a bit search over the 64-bit keys that re-streams the source once per bit, no buffer at all.
#include <math.h>

#include <type_traits>

#include "kernels.h"
#include "pool_walk.h"

namespace ldot {

constexpr int kBigT = 256;
constexpr int kBigWaves = kBigT / 64;
constexpr int kBigLPT = kMaxKp / kBigT;      // list entries per thread
constexpr int kBigSamplePT = 8;              // sample columns per thread (dense source)
constexpr int kBigSample = kBigSamplePT * kBigT;
static_assert(kMaxKp % kBigT == 0, "list entries per thread");

// sum / maximum over the workgroup: a wave reduction, one LDS word per wave, ONE barrier (`slot` alternates between consecutive calls)
__device__ __forceinline__ int big_sum(int v, int* red, int& slot) {
    const int w = wave_sum_dpp(v);
    if ((threadIdx.x & 63) == 0) red[slot * kBigWaves + (threadIdx.x >> 6)] = w;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int i = 0; i < kBigWaves; ++i) t += red[slot * kBigWaves + i];
    slot ^= 1;
    return t;
}
__device__ __forceinline__ void big_and_or_max(uint32_t& a, uint32_t& o, uint32_t& mx, uint32_t* red3) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a &= __shfl_xor(a, off);
        o |= __shfl_xor(o, off);
        const uint32_t m2 = __shfl_xor(mx, off);
        mx = m2 > mx ? m2 : mx;
    }
    __syncthreads();   // (red3 may still be read from an earlier call)
    if ((threadIdx.x & 63) == 0) {
        red3[threadIdx.x >> 6] = a;
        red3[kBigWaves + (threadIdx.x >> 6)] = o;
        red3[2 * kBigWaves + (threadIdx.x >> 6)] = mx;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kBigWaves; ++w) {
        a &= red3[w];
        o |= red3[kBigWaves + w];
        mx = red3[2 * kBigWaves + w] > mx ? red3[2 * kBigWaves + w] : mx;
    }
}

// inclusive prefix sum over the wave, pure VALU: row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes (a lane without a source adds 0), then the
// totals of the rows before (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3 — the last two steps of wave_sum_dpp)
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}

// the elements of one row of a score chunk, handed to f in per-thread batches (descending score keys, row ids, valid bits); f is called by ALL threads the same number of times
struct BigDenseSrc {
    const float* row;
    int64_t ncols, lds_elems;
    uint32_t idx_base;
    // f(dk[16], rid[16], valid bits): the thread's 16 elements of a block of 4096 columns
    template <class F>
    __device__ __forceinline__ void stream(F&& f) const {
        constexpr int U = 4;   // vectors in flight per thread
        for (int64_t c0 = 0; c0 < ncols; c0 += (int64_t)U * kBigT * 4) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t c = c0 + ((int64_t)u * kBigT + threadIdx.x) * 4;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (c + 3 < lds_elems) {   // (score rows are lds_elems >= round_up(ncols, 4) long)
                    v[u] = *(const f32x4*)(row + c);
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (c + e < ncols) v[u][e] = row[c + e];
                }
            }
            uint32_t dk[4 * U], rid[4 * U], valid = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t c = c0 + ((int64_t)u * kBigT + threadIdx.x) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dk[u * 4 + e] = desc_key(v[u][e]);
                    rid[u * 4 + e] = idx_base + (uint32_t)(c + e);
                    valid |= c + e < ncols ? 1u << (u * 4 + e) : 0u;
                }
            }
            f(std::integral_constant<int, 4 * U>{}, dk, rid, valid);
        }
    }
};

// the 8 scores of every record of a query's sub-pools: the records are numbered densely through the exclusive prefix sums `pre` of the
// sub-pools' record counts (LDS, [nsubs + 1]); thread t takes the records t, t + 256, ...
struct BigPoolSrc {
    const uint4* base;
    int nsubs, total;
    int32_t row_end;
    const uint32_t* cw;   // LDS: clamped counter words
    const int* pre;       // LDS
    // f(dk[8], rid[8], valid bits): the 8 scores of the thread's record
    template <class F>
    __device__ __forceinline__ void stream(F&& f) const {
        for (int j0 = 0; j0 < total; j0 += kBigT) {
            const int j = j0 + threadIdx.x;
            const bool have = j < total;
            const int jj = have ? j : total - 1;
            int lo = 0, hi = nsubs;   // pre[lo] <= jj < pre[hi]: the LAST sub-pool whose first record is <= jj holds it
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (pre[mid] <= jj) lo = mid; else hi = mid;
            }
            const int e = pool_entry_of(jj - pre[lo], cw[lo]);
            const uint4* rec = base + (int64_t)e * kPoolPlanes * nsubs + lo;
            const uint4 p0 = rec[0], p1 = rec[nsubs];
            const int32_t r = (int32_t)rec[2 * nsubs].x;
            uint32_t dk[8] = {desc_key(__uint_as_float(p0.x)), desc_key(__uint_as_float(p0.y)), desc_key(__uint_as_float(p0.z)), desc_key(__uint_as_float(p0.w)),
                              desc_key(__uint_as_float(p1.x)), desc_key(__uint_as_float(p1.y)), desc_key(__uint_as_float(p1.z)), desc_key(__uint_as_float(p1.w))};
            uint32_t rid[8], valid = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int32_t ri = r + (i < 4 ? i : i - 4 + kPoolRecHiRow);
                rid[i] = (uint32_t)ri;
                valid |= (have && ri < row_end) ? 1u << i : 0u;
            }
            f(std::integral_constant<int, 8>{}, dk, rid, valid);
        }
    }
};

struct BigShared {
    int red[2 * kBigWaves];
    uint32_t red3[3 * kBigWaves];
    int scount, wcount;
};

// What both kernels share once the source is set up.  Returns through `out`: the threshold score (the k'-th best, -inf while the list is not
// full) and, for m > 0, the m-th best score of the new list.
struct BigResult {
    float kth_score, mth_score;
    int n_sel;
};

template <int APT, class SRC>
__device__ __forceinline__ BigResult big_select(const SRC& src, uint32_t pivot_in, bool sample_pivot, float* __restrict__ ls,
                                                int32_t* __restrict__ li, int kp, int m_want, uint32_t* skey, uint32_t* srow, BigShared& sh) {
    constexpr int CAP = APT * kBigT;
    const int tid = threadIdx.x, lane = tid & 63;
    int slot = 0;
    // ---- the running list -> registers (descending keys; 0xffffffff = no entry) ----------------------------------------------------
    uint32_t lkey[kBigLPT];
    int32_t lrow[kBigLPT];
    int mine = 0;
    uint32_t l_and = 0xffffffffu, l_or = 0u, l_max = 0u;
#pragma unroll
    for (int r = 0; r < kBigLPT; ++r) {
        const int e = r * kBigT + tid;
        const int el = e < kp ? e : kp - 1;          // (unconditional loads from clamped addresses)
        const int32_t lr = li[el];
        const float lsv = ls[el];
        lrow[r] = e < kp ? lr : -1;
        lkey[r] = lrow[r] >= 0 ? desc_key(lsv) : 0xffffffffu;
        const bool have = lrow[r] >= 0;
        mine += have ? 1 : 0;
        l_and &= lkey[r];
        l_or |= have ? lkey[r] : 0u;
        l_max = (have && lkey[r] > l_max) ? lkey[r] : l_max;
    }
    if (tid == 0) sh.scount = sh.wcount = 0;
    const int n_list = big_sum(mine, sh.red, slot);      // (its barrier publishes the counters)
    big_and_or_max(l_and, l_or, l_max, sh.red3);
    // ---- the pivot: nothing with a key above it is kept ---------------------------------------------------------------------------
    uint32_t pivot = pivot_in;
    if (n_list >= kp && l_max < pivot) pivot = l_max;    // a full list: its worst key (ties with it may still win on the row)
    const uint32_t pivot_sure = pivot;                   // (what holds without the sample)
    if (sample_pivot) pivot = pivot < src.sample_pivot ? pivot : src.sample_pivot;
    // ---- one pass over the source: what passes the pivot -> LDS -------------------------------------------------------------------
    // (a thread counts the hits among its elements, ONE wave scan and ONE LDS atomic per block place them: an atomic per element —
    // 116 dependent LDS round trips per wave and row — was most of this pass)
    src.stream([&](auto n_tag, const uint32_t* dk, const uint32_t* rid, uint32_t valid) {
        constexpr int N = decltype(n_tag)::value;
        uint32_t hit = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) hit |= (((valid >> i) & 1u) && dk[i] <= pivot) ? 1u << i : 0u;
        const int cnt = __popc(hit);
        const int incl = wave_incl_scan_dpp(cnt);
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        if (tot) {   // (uniform)
            int base = 0;
            if (lane == 63) base = atomicAdd(&sh.scount, tot);
            base = __builtin_amdgcn_readlane(base, 63);
            int pos = base + incl - cnt;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if ((hit >> i) & 1u) {
                    if (pos < CAP) {
                        skey[pos] = dk[i];
                        srow[pos] = rid[i];
                    }
                    ++pos;
                }
            }
        }
    });
    __syncthreads();
    const int C = sh.scount;
    int c_le = 0;
#pragma unroll
    for (int r = 0; r < kBigLPT; ++r) c_le += lkey[r] <= pivot ? 1 : 0;   // (0xffffffff never passes: pivot <= 0xfffffffe)
    const int n_le = big_sum(c_le, sh.red, slot) + C;
    const bool fast = C <= CAP && (pivot == pivot_sure || n_le >= kp);
    BigResult out;
    uint32_t kth = 0xffffffffu, row_cut = 0xffffffffu, mth = 0xffffffffu;
    int n_all;
    if (fast) {
        uint32_t ak[APT], ar[APT];
#pragma unroll
        for (int j = 0; j < APT; ++j) {
            const int e = j * kBigT + tid;
            ak[j] = e < C ? skey[e] : 0xffffffffu;
            ar[j] = e < C ? srow[e] : 0xffffffffu;
        }
        n_all = n_list + C;
        // the t-th smallest score word over list + survivors (the bits all of them share are skipped)
        uint32_t a_and = l_and, a_or = l_or, a_mx = 0u;
#pragma unroll
        for (int j = 0; j < APT; ++j) {
            a_and &= ak[j];
            a_or |= ak[j] != 0xffffffffu ? ak[j] : 0u;
        }
        big_and_or_max(a_and, a_or, a_mx, sh.red3);
        const uint32_t diff = a_and ^ a_or;
        const int hb = diff ? 31 - __clz((int)diff) : -1;
        const uint32_t prefix = hb >= 31 ? 0u : hb < 0 ? a_or : (a_or & ~((2u << hb) - 1u));
        auto tth = [&](int t) {
            uint32_t x = prefix;
            for (int bit = hb; bit >= 0; --bit) {
                const uint32_t test = x | ((1u << bit) - 1u);
                int c = 0;
#pragma unroll
                for (int r = 0; r < kBigLPT; ++r) c += lkey[r] <= test ? 1 : 0;
#pragma unroll
                for (int j = 0; j < APT; ++j) c += ak[j] <= test ? 1 : 0;
                if (big_sum(c, sh.red, slot) < t) x |= 1u << bit;
            }
            return x;
        };
        if (m_want > 0 && n_all >= m_want) mth = tth(m_want);
        if (n_all > kp) {
            kth = tth(kp);
            int lt = 0, eq = 0;
#pragma unroll
            for (int r = 0; r < kBigLPT; ++r) {
                lt += lkey[r] < kth ? 1 : 0;
                eq += lkey[r] == kth ? 1 : 0;
            }
#pragma unroll
            for (int j = 0; j < APT; ++j) {
                lt += ak[j] < kth ? 1 : 0;
                eq += ak[j] == kth ? 1 : 0;
            }
            lt = big_sum(lt, sh.red, slot);
            eq = big_sum(eq, sh.red, slot);
            const int need_ties = kp - lt;
            if (eq > need_ties) {            // (uniform) equal scores straddle the k'-th place: the need_ties lowest rows among them
                uint32_t cut = 0;
                for (int b = 31; b >= 0; --b) {
                    const uint32_t test = cut | ((1u << b) - 1u);
                    int c = 0;
#pragma unroll
                    for (int r = 0; r < kBigLPT; ++r) c += (lkey[r] == kth && (uint32_t)lrow[r] <= test) ? 1 : 0;
#pragma unroll
                    for (int j = 0; j < APT; ++j) c += (ak[j] == kth && ar[j] <= test) ? 1 : 0;
                    if (big_sum(c, sh.red, slot) < need_ties) cut |= 1u << b;
                }
                row_cut = cut;
            }
        } else if (n_all == kp) {
            kth = a_mx > l_max ? a_mx : l_max;   // everything is kept; the threshold is the worst key
#pragma unroll
            for (int j = 0; j < APT; ++j) kth = (ak[j] != 0xffffffffu && ak[j] > kth) ? ak[j] : kth;
            uint32_t d0 = 0xffffffffu, d1 = 0u;
            big_and_or_max(d0, d1, kth, sh.red3);
        }
        // ---- the winners, as a set: a thread's list entries, then its survivors; then the empty slots --------------------------------
        uint32_t lselm = 0, selm = 0;
#pragma unroll
        for (int r = 0; r < kBigLPT; ++r)
            lselm |= (lkey[r] != 0xffffffffu && (lkey[r] < kth || (lkey[r] == kth && (uint32_t)lrow[r] <= row_cut))) ? (1u << r) : 0u;
#pragma unroll
        for (int j = 0; j < APT; ++j)
            selm |= (ak[j] != 0xffffffffu && (ak[j] < kth || (ak[j] == kth && ar[j] <= row_cut))) ? (1u << j) : 0u;
        static_assert(APT <= 32 && kBigLPT <= 32, "verdict masks");
        const int sel_cnt = __popc(selm) + __popc(lselm);
        int incl = sel_cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        __syncthreads();                     // (`red` is free: every thread has left the last sum; all list entries are in registers)
        if (lane == 63) sh.red[tid >> 6] = incl;
        __syncthreads();
        int pos = incl - sel_cnt, m = 0;
#pragma unroll
        for (int w = 0; w < kBigWaves; ++w) {
            pos += w < (tid >> 6) ? sh.red[w] : 0;
            m += sh.red[w];
        }
#pragma unroll
        for (int r = 0; r < kBigLPT; ++r) {
            if ((lselm >> r) & 1u) {
                ls[pos] = desc_key_to_float(lkey[r]);
                li[pos] = lrow[r];
                ++pos;
            }
        }
#pragma unroll
        for (int j = 0; j < APT; ++j) {
            if ((selm >> j) & 1u) {
                ls[pos] = desc_key_to_float(ak[j]);
                li[pos] = (int32_t)ar[j];
                ++pos;
            }
        }
        for (int e = m + tid; e < kp; e += kBigT) {
            ls[e] = LDOT_PAD_SCORE;
            li[e] = -1;
        }
        out.n_sel = m;
    } else {
        // ---- SLOW path: bit search over the 64-bit keys {score word, row}, the source streamed once per bit ---------------------------
        int vc = 0;
        src.stream([&](auto, const uint32_t*, const uint32_t*, uint32_t valid) { vc += __popc(valid); });
        n_all = n_list + big_sum(vc, sh.red, slot);
        auto tth64 = [&](int t) {
            uint64_t x = 0;
            for (int bit = 63; bit >= 0; --bit) {
                const uint64_t test = x | ((1ull << bit) - 1ull);
                int c = 0;
#pragma unroll
                for (int r = 0; r < kBigLPT; ++r) c += (lrow[r] >= 0 && (((uint64_t)lkey[r] << 32) | (uint32_t)lrow[r]) <= test) ? 1 : 0;
                src.stream([&](auto n_tag, const uint32_t* dk, const uint32_t* rid, uint32_t valid) {
#pragma unroll
                    for (int i = 0; i < decltype(n_tag)::value; ++i) c += (((valid >> i) & 1u) && (((uint64_t)dk[i] << 32) | rid[i]) <= test) ? 1 : 0;
                });
                if (big_sum(c, sh.red, slot) < t) x |= 1ull << bit;
            }
            return x;
        };
        if (m_want > 0 && n_all >= m_want) mth = (uint32_t)(tth64(m_want) >> 32);
        uint64_t k64 = ~0ull;
        if (n_all >= kp) k64 = tth64(kp);
        kth = n_all >= kp ? (uint32_t)(k64 >> 32) : 0xffffffffu;
        // the winners: list entries first (positions by a workgroup scan), then the source's (positions from a counter)
        uint32_t lselm = 0;
#pragma unroll
        for (int r = 0; r < kBigLPT; ++r) lselm |= (lrow[r] >= 0 && (((uint64_t)lkey[r] << 32) | (uint32_t)lrow[r]) <= k64) ? (1u << r) : 0u;
        const int sel_cnt = __popc(lselm);
        int incl = sel_cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        __syncthreads();
        if (lane == 63) sh.red[tid >> 6] = incl;
        __syncthreads();
        int pos = incl - sel_cnt, m_list = 0;
#pragma unroll
        for (int w = 0; w < kBigWaves; ++w) {
            pos += w < (tid >> 6) ? sh.red[w] : 0;
            m_list += sh.red[w];
        }
#pragma unroll
        for (int r = 0; r < kBigLPT; ++r) {
            if ((lselm >> r) & 1u) {
                ls[pos] = desc_key_to_float(lkey[r]);
                li[pos] = lrow[r];
                ++pos;
            }
        }
        src.stream([&](auto n_tag, const uint32_t* dk, const uint32_t* rid, uint32_t valid) {
            constexpr int N = decltype(n_tag)::value;
            uint32_t hit = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) hit |= (((valid >> i) & 1u) && (((uint64_t)dk[i] << 32) | rid[i]) <= k64) ? 1u << i : 0u;
            const int cnt = __popc(hit);
            const int incl = wave_incl_scan_dpp(cnt);
            const int tot = __builtin_amdgcn_readlane(incl, 63);
            if (tot) {
                int base = 0;
                if (lane == 63) base = atomicAdd(&sh.wcount, tot);
                base = __builtin_amdgcn_readlane(base, 63);
                int p = m_list + base + incl - cnt;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    if ((hit >> i) & 1u) {
                        if (p < kp) {
                            ls[p] = desc_key_to_float(dk[i]);
                            li[p] = (int32_t)rid[i];
                        }
                        ++p;
                    }
                }
            }
        });
        __syncthreads();
        const int m = min(kp, m_list + sh.wcount);
        for (int e = m + tid; e < kp; e += kBigT) {
            ls[e] = LDOT_PAD_SCORE;
            li[e] = -1;
        }
        out.n_sel = m;
    }
    out.kth_score = n_all >= kp ? desc_key_to_float(kth) : -INFINITY;
    out.mth_score = (m_want > 0 && n_all >= m_want) ? desc_key_to_float(mth) : -INFINITY;
    return out;
}

struct BigDenseSrcS : BigDenseSrc {
    uint32_t sample_pivot;
};
struct BigPoolSrcS : BigPoolSrc {
    uint32_t sample_pivot;
};

// dense source: one row of a materialised score chunk + the running list -> the new running list (a set) and its threshold
template <int APT>
__global__ __launch_bounds__(kBigT, APT <= 16 ? 4 : 2) void select_big_dense_kernel(const float* __restrict__ S, int64_t lds_elems, int64_t ncols, int64_t idx_base,
                                                                float* __restrict__ list_s, int32_t* __restrict__ list_i, int kp,
                                                                float* __restrict__ tau) {
    extern __shared__ __attribute__((aligned(16))) uint32_t big_lds[];
    __shared__ BigShared sh;
    constexpr int CAP = APT * kBigT;
    uint32_t* skey = big_lds;
    uint32_t* srow = big_lds + CAP;
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x;
    BigDenseSrcS src;
    src.row = S + q * lds_elems;
    src.ncols = ncols;
    src.lds_elems = lds_elems;
    src.idx_base = (uint32_t)idx_base;
    src.sample_pivot = 0xfffffffeu;
    // the sample pivot: the r-th best of 2048 columns at equal distances (rows longer than the survivor buffer only)
    const bool sample = ncols > CAP;
    if (sample) {
        const int64_t stride = ncols / kBigSample;   // (>= 2: CAP >= kBigSample)
        uint32_t sk[kBigSamplePT];
        uint32_t s_and = 0xffffffffu, s_or = 0u, s_mx = 0u;
#pragma unroll
        for (int i = 0; i < kBigSamplePT; ++i) {
            sk[i] = desc_key(src.row[((int64_t)i * kBigT + tid) * stride]);
            s_and &= sk[i];
            s_or |= sk[i];
        }
        big_and_or_max(s_and, s_or, s_mx, sh.red3);
        const double r0 = (double)kp * kBigSample / (double)ncols;
        const int r = (int)(r0 + 5.0 * sqrt(r0) + 2.0);
        if (r < kBigSample) {
            const uint32_t diff = s_and ^ s_or;
            const int hb = diff ? 31 - __clz((int)diff) : -1;
            uint32_t x = hb >= 31 ? 0u : hb < 0 ? s_or : (s_or & ~((2u << hb) - 1u));
            int slot = 0;
            for (int bit = hb; bit >= 0; --bit) {
                const uint32_t test = x | ((1u << bit) - 1u);
                int c = 0;
#pragma unroll
                for (int i = 0; i < kBigSamplePT; ++i) c += sk[i] <= test ? 1 : 0;
                if (big_sum(c, sh.red, slot) < r) x |= 1u << bit;
            }
            src.sample_pivot = x < 0xfffffffeu ? x : 0xfffffffeu;
            __syncthreads();   // (big_select starts its sums in slot 0 again)
        }
    }
    const BigResult res = big_select<APT>(src, 0xfffffffeu, sample, list_s + q * kp, list_i + q * kp, kp, 0, skey, srow, sh);
    if (tau && tid == 0) tau[q] = res.kth_score;
}

// pool source: the per-query sub-pools filled by the fused filter (score_filter.hip) + the running list; resets the counters.  The
// bookkeeping of select_pools_kernel (select.hip): thresholds never go down, the optimistic scan's next threshold / final check, the
// per-query record count, pool overflows.
template <int APT>
__global__ __launch_bounds__(kBigT, 4) void select_big_pools_kernel(const uint4* __restrict__ pool, int32_t* __restrict__ pool_cnt, int nsubs,
                                                                int64_t nq, int32_t row_end, float* __restrict__ list_s,
                                                                int32_t* __restrict__ list_i, int kp, float* __restrict__ tau,
                                                                int32_t* __restrict__ overflow, int32_t* __restrict__ over_sum,
                                                                int32_t* __restrict__ qcnt, float* __restrict__ tau_opt, int opt_m) {
    extern __shared__ __attribute__((aligned(16))) uint32_t big_lds[];
    __shared__ BigShared sh;
    __shared__ int over_sh;
    constexpr int CAP = APT * kBigT;
    uint32_t* skey = big_lds;
    uint32_t* srow = big_lds + CAP;
    uint32_t* cw = big_lds + 2 * CAP;            // [nsubs]
    int* pre = (int*)(cw + nsubs);               // [nsubs + 1]
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* cnt = pool_cnt + q * (int64_t)nsubs;
    if (tid == 0) over_sh = 0;
    // counter words -> clamped words + exclusive prefix sums of the record counts (thread t: sub-pools [t * per, (t + 1) * per))
    const int per = (nsubs + kBigT - 1) / kBigT;
    bool over = false;
    int tot = 0;
    for (int i = 0; i < per; ++i) {
        const int s = tid * per + i;
        if (s < nsubs) {
            int c;
            cw[s] = pool_counts((uint32_t)cnt[s], c, over);
            pre[s] = tot;                        // (the thread's own running sum; its base is added below)
            tot += c;
            cnt[s] = 0;
        }
    }
    int incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) sh.red[wave] = incl;
    __syncthreads();
    int base = incl - tot, total = 0;
#pragma unroll
    for (int w = 0; w < kBigWaves; ++w) {
        base += w < wave ? sh.red[w] : 0;
        total += sh.red[w];
    }
    for (int i = 0; i < per; ++i) {
        const int s = tid * per + i;
        if (s < nsubs) pre[s] += base;
    }
    if (tid == 0) pre[nsubs] = total;
    if (__any(over) && lane == 0) over_sh = 1;
    __syncthreads();
    BigPoolSrcS src;
    src.base = pool + q * (int64_t)kPoolCap * kPoolPlanes * nsubs;
    src.nsubs = nsubs;
    src.total = total;
    src.row_end = row_end;
    src.cw = cw;
    src.pre = pre;
    src.sample_pivot = 0xfffffffeu;
    // a threshold above the list's own worst key (sharded search: agreed between the ranks) filters the source
    const float t_prev = tau ? tau[q] : -INFINITY;
    uint32_t pivot = 0xfffffffeu;
    if (t_prev > -INFINITY) pivot = desc_key(t_prev);
    const BigResult res = big_select<APT>(src, pivot, false, list_s + q * kp, list_i + q * kp, kp, (tau_opt && opt_m > 0 && opt_m < kp) ? opt_m : 0,
                                          skey, srow, sh);
    if (tid == 0) {
        const float t_guar = fmaxf(t_prev, res.kth_score);
        if (tau) tau[q] = t_guar;                // (never down: see WaveSelector::finish)
        bool unproven = false;
        if (tau_opt) {
            if (opt_m > 0)
                tau_opt[q] = fmaxf(tau_opt[q], opt_m >= kp ? t_guar : res.mth_score);
            else if (opt_m == 0)
                unproven = !(t_guar >= tau_opt[q]);
        }
        if (qcnt) qcnt[q] += total;
        if ((over_sh || unproven) && atomicExch(&overflow[q], 1) == 0) atomicAdd(over_sum, 1);
    }
}

static size_t big_lds_bytes(int apt, int nsubs) { return (size_t)apt * kBigT * 8 + (nsubs > 0 ? (size_t)(2 * nsubs + 1) * 4 : 0); }

// (kp > 512; rows of up to 65 536 columns: a 2048-column sample of a longer row is too coarse a pivot)
bool select_big_dense_ok(int kp, int64_t ncols, int64_t idx_base) {
    return kp > 512 && ncols <= 65536 && idx_base + ncols < ((int64_t)1 << 31);
}

int launch_select_big_dense(const float* S, int64_t lds_elems, int64_t nq, int64_t ncols, int64_t idx_base, float* list_s, int32_t* list_i,
                            int kp, float* tau, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    static bool attr16[kAttrDevices], attr32[kAttrDevices];
    if (kp <= 1536) {   // ~1.5 k' + the sample's spread survivors: 4096 slots
        LDOT_HIP_CHECK(set_max_dynamic_lds_once((const void*)select_big_dense_kernel<16>, (int)big_lds_bytes(16, 0), attr16));
        hipLaunchKernelGGL(select_big_dense_kernel<16>, dim3((unsigned)nq), dim3(kBigT), big_lds_bytes(16, 0), st, S, lds_elems, ncols, idx_base,
                           list_s, list_i, kp, tau);
    } else {
        LDOT_HIP_CHECK(set_max_dynamic_lds_once((const void*)select_big_dense_kernel<32>, (int)big_lds_bytes(32, 0), attr32));
        hipLaunchKernelGGL(select_big_dense_kernel<32>, dim3((unsigned)nq), dim3(kBigT), big_lds_bytes(32, 0), st, S, lds_elems, ncols, idx_base,
                           list_s, list_i, kp, tau);
    }
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_select_big_pools(const uint4* pool, int32_t* pool_cnt, int nsubs, int64_t nq, int32_t row_end, float* list_s, int32_t* list_i,
                            int kp, float* tau, int32_t* overflow_flags, int32_t* over_sum, int32_t* qcnt, hipStream_t st, float* tau_opt,
                            int opt_m) {
    if (nq <= 0) return LDOT_OK;
    LDOT_REQUIRE(nsubs <= kPoolSubsMax, LDOT_EINVAL, "select_big_pools: too many sub-pools");
    static bool attr16[kAttrDevices];
    LDOT_HIP_CHECK(set_max_dynamic_lds_once((const void*)select_big_pools_kernel<16>, (int)big_lds_bytes(16, kPoolSubsMax), attr16));
    hipLaunchKernelGGL(select_big_pools_kernel<16>, dim3((unsigned)nq), dim3(kBigT), big_lds_bytes(16, nsubs), st, pool, pool_cnt, nsubs, nq, row_end,
                       list_s, list_i, kp, tau, overflow_flags, over_sum, qcnt, tau_opt, opt_m);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

}  // namespace ldot
