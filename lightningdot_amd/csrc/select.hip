// Streaming top-k' selection in LDS (one 256-thread workgroup per query).
//
// Replaces the per-query max-heap of faiss' flat search (dvl/indexer/faiss_indexers.py:83 -> IndexFlatIP.search)
// with a CDNA-friendly scheme: candidates are packed into unique 64-bit keys
//     key = (descending-order image of the fp32 score) << 32 | row
// so that "smaller key" == "better" == (higher score, then lower row).  A workgroup streams candidate segments,
// appends those that beat the current k'-th key to an LDS buffer and, only when the buffer could overflow,
// compacts it with a bitonic sort (LDS, 64-bit compare-exchange) and tightens the threshold.
#include <float.h>
#include <stdlib.h>

#include <algorithm>

#include "bitonic.h"
#include "kernels.h"
#include "pool_walk.h"

namespace ldot {

constexpr uint64_t kEmptyKey = ~0ull;

__device__ inline uint64_t make_key(float s, uint32_t row) { return ((uint64_t)desc_key(s) << 32) | row; }

// Workgroup-wide sync for the select kernels.  A single-wave workgroup (the pool select) needs no hardware barrier:
// the LDS executes one wave's operations in issue order, so draining the LDS counter is enough — and an s_barrier per
// bitonic stage is what dominated that kernel.
__device__ __forceinline__ void sel_sync() {
    if (blockDim.x <= 64)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else
        __syncthreads();
}

struct Selector {
    uint64_t* keys;   // LDS [cap]
    int* count;       // LDS
    uint64_t tau;     // uniform
    int kp;
    int cap;          // LDS buffer length (power of two, >= 2 * kp)

    __device__ inline void init(uint64_t* k, int* c, int kp_, int cap_) {
        keys = k;
        count = c;
        kp = kp_;
        cap = cap_;
        tau = kEmptyKey;
        if (threadIdx.x == 0) *count = 0;
        sel_sync();
    }
    // Wave-aggregated append: must be called by ALL lanes of a wave together (valid = false for lanes without a
    // candidate).  One LDS atomic per wave per call instead of one per hit.
    __device__ inline void push(uint64_t key, bool valid) {
        const bool hit = valid && key < tau;
        const unsigned long long mask = __ballot(hit);
        if (mask) {
            const int lane = threadIdx.x & 63;
            const int leader = __ffsll((long long)mask) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(count, __popcll(mask));
            base = __shfl(base, leader);
            if (hit) keys[base + __popcll(mask & ((1ull << lane) - 1ull))] = key;
        }
    }
    // sort, truncate to kp, refresh tau.  Must be called by all threads.
    __device__ inline void compact() {
        sel_sync();
        const int n = *count;
        int P = 2;
        while (P < n) P <<= 1;
        for (int i = n + threadIdx.x; i < P; i += blockDim.x) keys[i] = kEmptyKey;
        sel_sync();
        bitonic_sort_lds(keys, P);
        const int m = n < kp ? n : kp;
        const uint64_t t = (m >= kp) ? keys[kp - 1] : kEmptyKey;
        sel_sync();
        if (threadIdx.x == 0) *count = m;
        tau = t;
        sel_sync();
    }
    // call (all threads) before streaming up to `upcoming` more candidates (upcoming <= cap - kp)
    __device__ inline void reserve(int upcoming) {
        sel_sync();
        const int n = *count;
        sel_sync();                              // everyone has read count before anyone pushes again
        // also compact as soon as a first threshold can be had (list not full yet, >= kp keys buffered): sorting 1024
        // keys now and filtering the rest is cheaper than sorting a full buffer of unfiltered keys later
        if (n + upcoming > cap || (tau == kEmptyKey && n >= kp)) compact();            // uniform branch
    }
    // Adopt the running list (valid entries first, empty slots last; sorted or not — the wave-level selectors keep
    // sets) and take the threshold from it once it is full.
    __device__ inline void load_list(const float* ls, const int32_t* li) {
        for (int e0 = 0; e0 < kp; e0 += blockDim.x) {
            const int e = e0 + threadIdx.x;
            const bool valid = e < kp && li[e] >= 0;
            if (valid) keys[e] = make_key(ls[e], (uint32_t)li[e]);
            const unsigned long long mask = __ballot(valid);
            if ((threadIdx.x & 63) == 0 && mask) atomicAdd(count, __popcll(mask));
        }
        sel_sync();
        const int n = *count;
        sel_sync();
        if (n >= kp) compact();   // sorts the kp keys (cheap) and sets tau = the kp-th best
    }
    __device__ inline void finish(float* ls, int32_t* li, float* tau_out, bool keep_max = false) {
        compact();
        const int n = *count;
        for (int e = threadIdx.x; e < kp; e += blockDim.x) {
            if (e < n) {
                const uint64_t k = keys[e];
                ls[e] = desc_key_to_float((uint32_t)(k >> 32));
                li[e] = (int32_t)(uint32_t)(k & 0xffffffffu);
            } else {
                ls[e] = LDOT_PAD_SCORE;
                li[e] = -1;
            }
        }
        if (tau_out && threadIdx.x == 0) {
            const float t = (n >= kp) ? desc_key_to_float((uint32_t)(keys[kp - 1] >> 32)) : -INFINITY;
            *tau_out = keep_max ? fmaxf(*tau_out, t) : t;
        }
    }
};

// Single-wave variant of the selector (pool select: one wave per query).  The append counter is a wave-uniform
// register, appends are ballot/popcount compactions — no LDS atomics, no barriers.
struct WaveSelector {
    static __device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

    uint64_t* keys;   // LDS [cap]
    uint64_t tau;
    int n, kp, cap;

    __device__ inline void init(uint64_t* k, int kp_, int cap_) {
        keys = k;
        kp = kp_;
        cap = cap_;
        n = 0;
        tau = kEmptyKey;
    }
    __device__ inline void push(uint64_t key, bool valid) {
        const bool hit = valid && key < tau;
        const unsigned long long mask = __ballot(hit);
        const int lane = threadIdx.x & 63;
        if (hit) keys[n + __popcll(mask & ((1ull << lane) - 1ull))] = key;
        n += __popcll(mask);
    }
    // Keep the kp best (smallest) of keys[0..n): the list is a SET here — nothing downstream of the pool select needs it
    // ordered (the next pool select takes its threshold from tau[], the re-score kernel sorts by exact score) — so
    // instead of an LDS bitonic sort (LDS-bandwidth bound: ~370 KB of LDS traffic per query) the kp-th smallest key is
    // found by a bitwise binary search on register-resident keys (ballot + popcount, no LDS traffic) and the survivors
    // are compacted in place.  Buffers larger than 16 keys per lane (kp > 512) take the sort path.
    static constexpr int kRegKeys = 16;
    __device__ inline void compact() {
        if (n <= kp) {   // nothing to drop; the threshold is known only once the list is full
            if (n < kp) {
                tau = kEmptyKey;
                return;
            }
        }
        if (cap > kRegKeys * 64) {
            int P = 2;
            while (P < n) P <<= 1;
            for (int i = n + (threadIdx.x & 63); i < P; i += 64) keys[i] = kEmptyKey;
            wave_sync();
            bitonic_sort_lds(keys, P);   // (sort path: launched with one wave per workgroup)
            n = n < kp ? n : kp;
            tau = (n >= kp) ? keys[kp - 1] : kEmptyKey;
            return;
        }
        // the scalar unit is shared by the CU's waves: keep the per-iteration scalar work minimal (register count
        // specialised, 32-bit search on the score word; the row word is searched only to split exact score ties)
        if (n <= 128)
            select_regs<2>();
        else if (n <= 256)
            select_regs<4>();
        else if (n <= 512)
            select_regs<8>();
        else
            select_regs<16>();
    }
    template <int R>
    __device__ __forceinline__ void select_regs() {
        const int lane = threadIdx.x & 63;
        wave_sync();
        uint32_t hi[R], lo[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t k = (r * 64 + lane < n) ? keys[r * 64 + lane] : kEmptyKey;
            hi[r] = (uint32_t)(k >> 32);
            lo[r] = (uint32_t)k;
        }
        wave_sync();
        // Th = kp-th smallest score word (with multiplicity).  Empty slots are all ones in BOTH words and are never
        // counted (a valid key's row word is < 2^31).
        uint32_t Th = 0;
        for (int b = 31; b >= 0; --b) {
            const uint32_t trial = Th | ((1u << b) - 1u);
            int c = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) c += __popcll(__ballot(hi[r] <= trial && lo[r] != 0xffffffffu));
            if (c < kp) Th |= (1u << b);
        }
        int c_lt = 0, c_eq = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            c_lt += __popcll(__ballot(hi[r] < Th && lo[r] != 0xffffffffu));
            c_eq += __popcll(__ballot(hi[r] == Th && lo[r] != 0xffffffffu));
        }
        uint32_t Tl = 0xfffffffeu;   // every tie fits
        if (c_lt + c_eq > kp) {      // exact score ties straddle the cut: keep the kp - c_lt lowest rows among them
            const int need = kp - c_lt;
            Tl = 0;
            for (int b = 31; b >= 0; --b) {
                const uint32_t trial = Tl | ((1u << b) - 1u);
                int c = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) c += __popcll(__ballot(hi[r] == Th && lo[r] <= trial));
                if (c < need) Tl |= (1u << b);
            }
        }
        int base = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool hit = lo[r] != 0xffffffffu && (hi[r] < Th || (hi[r] == Th && lo[r] <= Tl));
            const unsigned long long mask = __ballot(hit);
            const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (hit && pos < kp) keys[pos] = ((uint64_t)hi[r] << 32) | lo[r];
            base += __popcll(mask);
        }
        wave_sync();
        n = kp;
        tau = ((uint64_t)Th << 32) | Tl;
    }
    __device__ inline void reserve(int upcoming) {
        // also compact as soon as a first threshold can be had (no threshold yet, >= kp keys buffered)
        if (n + upcoming > cap || (tau == kEmptyKey && n >= kp)) compact();
    }
    // the running list is a set (valid entries first, empty slots last); once full, its threshold is its worst key
    __device__ inline void load_list(const float* ls, const int32_t* li) {
        const int lane = threadIdx.x & 63;
        uint64_t worst = 0;
        for (int e0 = 0; e0 < kp; e0 += 64) {
            const int e = e0 + lane;
            const bool valid = e < kp && li[e] >= 0;
            const unsigned long long mask = __ballot(valid);
            if (valid) {
                const uint64_t key = make_key(ls[e], (uint32_t)li[e]);
                keys[n + __popcll(mask & ((1ull << lane) - 1ull))] = key;
                worst = key > worst ? key : worst;
            }
            n += __popcll(mask);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t hi = __shfl_xor((uint32_t)(worst >> 32), o), lo = __shfl_xor((uint32_t)worst, o);
            const uint64_t other = ((uint64_t)hi << 32) | lo;
            worst = other > worst ? other : worst;
        }
        wave_sync();
        tau = (n >= kp) ? worst : kEmptyKey;
    }
    // keep_max: the query's threshold never goes DOWN (the pool selects of a sharded search start from a threshold the ranks agreed on,
    // which may exceed this shard's own k'-th best score: ldot_index_search_scan)
    __device__ inline void finish(float* ls, int32_t* li, float* tau_out, bool keep_max = false) {
        compact();
        for (int e = (threadIdx.x & 63); e < kp; e += 64) {
            if (e < n) {
                const uint64_t k = keys[e];
                ls[e] = desc_key_to_float((uint32_t)(k >> 32));
                li[e] = (int32_t)(uint32_t)(k & 0xffffffffu);
            } else {
                ls[e] = LDOT_PAD_SCORE;
                li[e] = -1;
            }
        }
        if (tau_out && (threadIdx.x & 63) == 0) {
            const float t = (n >= kp) ? desc_key_to_float((uint32_t)(tau >> 32)) : -INFINITY;
            *tau_out = keep_max ? fmaxf(*tau_out, t) : t;
        }
    }
    // after finish(): the m-th best SCORE of the list (keys[0 .. n) are the kept set, sorted on the LDS-sort path); -inf if n < m
    template <int R>
    __device__ __forceinline__ float mth_best_regs(int m) const {
        const int lane = threadIdx.x & 63;
        uint32_t hi[R];
#pragma unroll
        for (int r = 0; r < R; ++r) hi[r] = (r * 64 + lane < n) ? (uint32_t)(keys[r * 64 + lane] >> 32) : 0xffffffffu;
        uint32_t T = 0;   // smallest score word with #(words <= T) >= m
        for (int b = 31; b >= 0; --b) {
            const uint32_t trial = T | ((1u << b) - 1u);
            int c = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) c += __popcll(__ballot(hi[r] <= trial && r * 64 + lane < n));
            if (c < m) T |= 1u << b;
        }
        return desc_key_to_float(T);
    }
    __device__ inline float mth_best(int m) const {
        if (n < m) return -INFINITY;
        wave_sync();
        if (cap > kRegKeys * 64) return desc_key_to_float((uint32_t)(keys[m - 1] >> 32));   // (sorted)
        if (n <= 128) return mth_best_regs<2>(m);
        if (n <= 256) return mth_best_regs<4>(m);
        if (n <= 512) return mth_best_regs<8>(m);
        return mth_best_regs<16>(m);
    }
};

__global__ __launch_bounds__(kSelThreads) void init_lists_kernel(float* ls, int32_t* li, int64_t n, float* tau,
                                                                 int64_t nq, int64_t nq_pad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        ls[i] = LDOT_PAD_SCORE;
        li[i] = -1;
    }
    if (tau && i < nq_pad) tau[i] = i < nq ? -INFINITY : INFINITY;
}

// LDS budget of a select launch: cap 64-bit keys, a power of two >= `want` that holds a full list (kp) plus the
// `upcoming` keys a step may append before the next compaction check
static inline int select_cap(int kp, int want, int upcoming) {
    int c = want;
    while (c < kp + upcoming) c <<= 1;
    return c;
}

// dense source: one row of a materialised score chunk (256 threads, 4096-key buffer)
__global__ __launch_bounds__(kSelThreads) void select_dense_kernel(const float* __restrict__ S, int64_t lds_elems,
                                                                   int64_t ncols, int64_t idx_base,
                                                                   float* __restrict__ list_s,
                                                                   int32_t* __restrict__ list_i, int kp, int cap,
                                                                   float* __restrict__ tau) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    __shared__ int count;
    const int64_t q = blockIdx.x;
    Selector sel;
    sel.init(keys, &count, kp, cap);
    float* ls = list_s + q * kp;
    int32_t* li = list_i + q * kp;
    sel.load_list(ls, li);
    const float* row = S + q * lds_elems;
    // a segment = 1024 columns (4 per thread) <= cap - kp.  Four segments are fetched per round trip: with one WG per
    // query the load latency is exposed once per fetch (every barrier in the selector drains vmcnt), so fewer, wider
    // fetches are what matters for small batches.
    constexpr int SEG = 4 * kSelThreads;
    for (int64_t c0 = 0; c0 < ncols; c0 += 4 * SEG) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t c = c0 + j * SEG + threadIdx.x * 4;
            v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c + 3 < ncols) {
                v[j] = *(const f32x4*)(row + c);
            } else {
                for (int e = 0; e < 4; ++e)
                    if (c + e < ncols) v[j][e] = row[c + e];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (c0 + j * SEG >= ncols) break;   // uniform
            sel.reserve(SEG);
            const int64_t c = c0 + j * SEG + threadIdx.x * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) sel.push(make_key(v[j][e], (uint32_t)(idx_base + c + e)), c + e < ncols);
        }
    }
    sel.finish(ls, li, tau ? tau + q : nullptr);
}

// dense source, wave-per-query variant (k' <= 512): the same register selection as the pool select instead of LDS bitonic
// compactions (which are LDS-bandwidth bound at ~1.3 MB of LDS traffic per query).  QPW independent query-waves per WG.
template <int QPW>
__global__ __launch_bounds__(64 * QPW) void select_dense_wave_kernel(const float* __restrict__ S, int64_t lds_elems,
                                                                     int64_t nq, int64_t ncols, int64_t idx_base,
                                                                     float* __restrict__ list_s,
                                                                     int32_t* __restrict__ list_i, int kp, int cap,
                                                                     float* __restrict__ tau) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * QPW + wq;
    if (q >= nq) return;
    WaveSelector sel;
    sel.init(keys + (size_t)wq * cap, kp, cap);
    float* ls = list_s + q * kp;
    int32_t* li = list_i + q * kp;
    const float* row = S + q * lds_elems;
    // a step = 256 columns (4 per lane); four steps are fetched per round trip
    constexpr int STEP = 256;
    auto fetch = [&](int64_t c) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c + 3 < ncols) {
            v = *(const f32x4*)(row + c);
        } else {
            for (int e = 0; e < 4; ++e)
                if (c + e < ncols) v[e] = row[c + e];
        }
        return v;
    };
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fetch(j * STEP + lane * 4);
    sel.load_list(ls, li);
    for (int64_t c0 = 0; c0 < ncols; c0 += 4 * STEP) {
        f32x4 w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = v[j];
        if (c0 + 4 * STEP < ncols) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fetch(c0 + 4 * STEP + j * STEP + lane * 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (c0 + j * STEP >= ncols) break;   // uniform
            sel.reserve(STEP);
            const int64_t c = c0 + j * STEP + lane * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) sel.push(make_key(w[j][e], (uint32_t)(idx_base + c + e)), c + e < ncols);
        }
    }
    sel.finish(ls, li, tau ? tau + q : nullptr);
}

// dense source, rows of at most 64 * 8 * NRUN columns (4096 / 5120: the warm-up chunk of the fused scan, Flickr / COCO sized indexes):
// the whole row sits in the wave's registers as NRUN runs of 8 CONSECUTIVE columns per lane, so every lane knows the maxima of its
// runs without any cross-lane work.  The k'-th largest RUN MAXIMUM is a lower bound of the k'-th best score (each maximum is a score),
// found by a bit search over NRUN keys per lane; only the ~1.1 k' scores at or above it are pushed to the selector.  The streaming
// variant below pushes and compacts its way through every column while it learns the threshold: 164 us for 10 000 x 4096 (the fused
// scan's warm-up), 104 us for 5000 x 5000; this kernel 68 / 50 us.
template <int NRUN, int QPW>
__global__ __launch_bounds__(64 * QPW) void select_dense_runs_kernel(const float* __restrict__ S, int64_t lds_elems, int64_t nq,
                                                                     int ncols, int64_t idx_base, float* __restrict__ list_s,
                                                                     int32_t* __restrict__ list_i, int kp, int cap,
                                                                     float* __restrict__ tau) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * QPW + wq;
    if (q >= nq) return;
    const float* row = S + q * lds_elems;
    float v[NRUN][8];
    uint32_t mk[NRUN];   // descending keys of the run maxima (all ones: no valid column in the run)
#pragma unroll
    for (int j = 0; j < NRUN; ++j) {
        const int c = (j * 64 + lane) * 8;
        // (rows are 16-B aligned and padded to a multiple of 256 columns by the callers: whole runs can be read; columns past ncols
        // are masked below)
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (c < ncols) {
            a = *(const f32x4*)(row + c);
            b = *(const f32x4*)(row + c + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[j][e] = a[e];
            v[j][4 + e] = b[e];
        }
    }
    WaveSelector sel;
    sel.init(keys + (size_t)wq * cap, kp, cap);
    float* ls = list_s + q * kp;
    int32_t* li = list_i + q * kp;
    sel.load_list(ls, li);
    int nvalid = 0;
#pragma unroll
    for (int j = 0; j < NRUN; ++j) {
        const int c = (j * 64 + lane) * 8;
        // (float maxima, ONE key per run; columns past ncols do not count; a NaN score never becomes a candidate on this path)
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, c + e < ncols ? v[j][e] : -INFINITY);
        mk[j] = c < ncols ? desc_key(m) : 0xffffffffu;
        nvalid += c < ncols ? 1 : 0;
    }
    nvalid = wave_sum_dpp(nvalid);
    // smallest key t with #(run maxima <= t) >= kp; 20 bits, the rest left set (a slightly lower threshold).  Fewer than kp runs:
    // everything is a candidate.
    uint32_t th = 0xffffffffu;
    if (nvalid >= kp) {
        th = 0;
        for (int b = 31; b >= 12; --b) {
            const uint32_t trial = th | ((1u << b) - 1u);
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < NRUN; ++j) cnt += mk[j] <= trial ? 1 : 0;
            if (wave_sum_dpp(cnt) < kp) th |= 1u << b;
        }
        th |= (1u << 12) - 1u;
    }
    // the threshold as a score: `s >= th_f` admits exactly the scores whose key is <= th, plus the other zero of a signed-zero threshold
    // (a superset is fine: the selector orders by key)
    float th_f = th == 0xffffffffu ? -INFINITY : desc_key_to_float(th);
    if (th_f != th_f) th_f = -INFINITY;   // (the 12 set low bits turn a -inf threshold into a NaN pattern)
    // how many scores are at or above the threshold?  (~1.1 k' on ordinary rows; thousands when scores tie at the threshold)
    int total = 0;
#pragma unroll
    for (int j = 0; j < NRUN; ++j) {
        const int c = (j * 64 + lane) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) total += c + e < ncols && v[j][e] >= th_f ? 1 : 0;
    }
    total = wave_sum_dpp(total);
    if (sel.n + total <= sel.cap) {
        // the ordinary case: everything fits the key buffer, no compaction inside the (fully unrolled) loop
#pragma unroll
        for (int j = 0; j < NRUN; ++j) {
            if (__ballot(mk[j] <= th)) {   // (wave-uniform)
                const int c = (j * 64 + lane) * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    sel.push(make_key(v[j][e], (uint32_t)(idx_base + c + e)), c + e < ncols && v[j][e] >= th_f);
            }
        }
    } else {
        // ties by the thousand: stream the row again through the compacting selector (one compaction site, rolled loop)
        for (int c0 = 0; c0 < ncols; c0 += 64) {
            const int c = c0 + lane;
            const float sc = c < ncols ? row[c] : 0.f;
            sel.reserve(64);
            sel.push(make_key(sc, (uint32_t)(idx_base + c)), c < ncols && sc >= th_f);
        }
    }
    sel.finish(ls, li, tau ? tau + q : nullptr);
}

// dense source, SHORT rows (ncols <= 64 * NV, k' <= 256): the whole row of scores sits in the wave's registers (NV per lane)
// together with the running list (4 per lane) and the k'-th best key is found by a bitwise binary search whose counts are pure VALU work
// (per-lane compare-and-count over the registers, one DPP wave reduction per bit) — no ballots, no LDS, no scalar-unit traffic.  Used for
// rows of at most 1024 columns (see launch_select_dense for the measurements that set this limit).
// Output list = a SET (valid entries first), tau = its k'-th best score.
template <int NV, int QPW>
__global__ __launch_bounds__(64 * QPW) void select_dense_regs_kernel(const float* __restrict__ S, int64_t lds_elems, int64_t nq,
                                                                     int ncols, int64_t idx_base, float* __restrict__ list_s,
                                                                     int32_t* __restrict__ list_i, int kp, float* __restrict__ tau) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * QPW + (threadIdx.x >> 6);
    if (q >= nq) return;
    float* ls = list_s + q * kp;
    int32_t* li = list_i + q * kp;
    const float* row = S + q * lds_elems;
    // chunk values: register r holds column (r / 4) * 256 + lane * 4 + r % 4 (coalesced float4 loads); invalid -> all-ones key
    uint32_t hi[NV + 4];
    uint32_t llo[4];   // rows of the running-list entries (chunk rows are implied by the column)
#pragma unroll
    for (int v = 0; v < NV / 4; ++v) {
        const int c0 = v * 256 + lane * 4;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (c0 + 3 < ncols) {
            x = *(const f32x4*)(row + c0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + e < ncols) x[e] = row[c0 + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) hi[v * 4 + e] = (c0 + e < ncols) ? desc_key(x[e]) : 0xffffffffu;
    }
    int nvalid_l = 0;
#pragma unroll
    for (int r = 0; r < NV; ++r) nvalid_l += hi[r] != 0xffffffffu;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int p = e * 64 + lane;
        const int32_t r = p < kp ? li[p] : -1;
        hi[NV + e] = r >= 0 ? desc_key(ls[p]) : 0xffffffffu;
        llo[e] = (uint32_t)r;
        nvalid_l += r >= 0;
    }
    // wave-wide integer sum in 7 DPP adds + one readlane (quad swaps, row mirrors, row broadcasts: the total lands in lane 63) —
    // the bpermute-based __shfl_xor reduction costs ~600 cycles per call, and the search calls it once per bit
    auto wave_sum = [&](int v) {
        v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
        v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);   // row_mirror  -> every lane: its row's sum
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast15 into rows 1 and 3
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast31 into rows 2 and 3
        return __builtin_amdgcn_readlane(v, 63);
    };
    const int nvalid = wave_sum(nvalid_l);
    uint32_t Th = 0xfffffffeu, Tl = 0xfffffffeu;   // "everything valid"
    if (nvalid > kp) {
        // Th = kp-th smallest score word (with multiplicity)
        Th = 0;
        for (int b = 31; b >= 0; --b) {
            const uint32_t trial = Th | ((1u << b) - 1u);
            int c = 0;
#pragma unroll
            for (int r = 0; r < NV + 4; ++r) c += hi[r] <= trial;
            if (wave_sum(c) < kp) Th |= (1u << b);
        }
        int c_lt = 0, c_eq = 0;
#pragma unroll
        for (int r = 0; r < NV + 4; ++r) {
            c_lt += hi[r] < Th;
            c_eq += hi[r] == Th;
        }
        c_lt = wave_sum(c_lt);
        c_eq = wave_sum(c_eq);
        if (c_lt + c_eq > kp) {   // exact score ties straddle the cut: keep the kp - c_lt lowest rows among them
            const int need = kp - c_lt;
            Tl = 0;
            for (int b = 31; b >= 0; --b) {
                const uint32_t trial = Tl | ((1u << b) - 1u);
                int c = 0;
                uint32_t lo0 = (uint32_t)idx_base + (uint32_t)lane * 4u;
                asm volatile("" : "+v"(lo0));   // (rare path: keep the NV row numbers out of registers — no hoisting out of the bit loop)
#pragma unroll
                for (int r = 0; r < NV; ++r)
                    c += hi[r] == Th && lo0 + (uint32_t)((r / 4) * 256 + r % 4) <= trial;
#pragma unroll
                for (int e = 0; e < 4; ++e) c += hi[NV + e] == Th && llo[e] <= trial;
                if (wave_sum(c) < need) Tl |= (1u << b);
            }
        }
    }
    // compaction: lane-exclusive prefix of the per-lane survivor counts, then every lane writes its survivors
    int mine = 0;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        const uint32_t lo = (uint32_t)(idx_base + (r / 4) * 256 + lane * 4 + r % 4);
        mine += hi[r] != 0xffffffffu && (hi[r] < Th || (hi[r] == Th && lo <= Tl));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) mine += hi[NV + e] != 0xffffffffu && (hi[NV + e] < Th || (hi[NV + e] == Th && llo[e] <= Tl));
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    int pos = incl - mine;
    const int total = __shfl(incl, 63);
    // the list entries are read before anything is written (registers), so the in-place update is safe
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (hi[NV + e] != 0xffffffffu && (hi[NV + e] < Th || (hi[NV + e] == Th && llo[e] <= Tl))) {
            if (pos < kp) {
                ls[pos] = desc_key_to_float(hi[NV + e]);
                li[pos] = (int32_t)llo[e];
            }
            ++pos;
        }
    }
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        const uint32_t lo = (uint32_t)(idx_base + (r / 4) * 256 + lane * 4 + r % 4);
        if (hi[r] != 0xffffffffu && (hi[r] < Th || (hi[r] == Th && lo <= Tl))) {
            if (pos < kp) {
                ls[pos] = desc_key_to_float(hi[r]);
                li[pos] = (int32_t)lo;
            }
            ++pos;
        }
    }
    for (int e = total + lane; e < kp; e += 64) {
        ls[e] = LDOT_PAD_SCORE;
        li[e] = -1;
    }
    if (tau && lane == 0) tau[q] = (total >= kp) ? desc_key_to_float(Th) : -INFINITY;
}

// segmented dense source (few queries x many rows): blockIdx.y = segment of `seg_cols` columns; no running list, the
// partial top-kp of the segment goes to part_[sl][seg][q][kp] with int64 labels (merged by select_lists_kernel)
__global__ __launch_bounds__(kSelThreads) void select_dense_parts_kernel(const float* __restrict__ S, int64_t lds_elems,
                                                                         int64_t nq, int64_t ncols, int64_t seg_cols,
                                                                         int64_t idx_base, int kp, int cap,
                                                                         float* __restrict__ part_s,
                                                                         int64_t* __restrict__ part_l) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    __shared__ int count;
    const int64_t q = blockIdx.x, seg = blockIdx.y;
    Selector sel;
    sel.init(keys, &count, kp, cap);
    const float* row = S + q * lds_elems;
    const int64_t cbeg = seg * seg_cols, cend = (cbeg + seg_cols < ncols) ? cbeg + seg_cols : ncols;
    for (int64_t c0 = cbeg; c0 < cend; c0 += 4 * kSelThreads) {
        sel.reserve(4 * kSelThreads);
        const int64_t c = c0 + threadIdx.x * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c + 3 < cend) {
            v = *(const f32x4*)(row + c);
        } else {
            for (int e = 0; e < 4; ++e)
                if (c + e < cend) v[e] = row[c + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) sel.push(make_key(v[e], (uint32_t)(idx_base + c + e)), c + e < cend);
    }
    sel.compact();
    const int n = count;
    float* ps = part_s + (seg * nq + q) * kp;
    int64_t* pl = part_l + (seg * nq + q) * kp;
    for (int e = threadIdx.x; e < kp; e += kSelThreads) {
        if (e < n) {
            const uint64_t k = keys[e];
            ps[e] = desc_key_to_float((uint32_t)(k >> 32));
            pl[e] = (int64_t)(uint32_t)(k & 0xffffffffu);
        } else {
            ps[e] = LDOT_PAD_SCORE;
            pl[e] = LDOT_PAD_LABEL;
        }
    }
}

__global__ __launch_bounds__(256) void lists_to_parts_kernel(const float* __restrict__ ls, const int32_t* __restrict__ li,
                                                             int64_t n, float* __restrict__ ps, int64_t* __restrict__ pl) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        ps[i] = ls[i];
        pl[i] = (int64_t)li[i];
    }
}
// (the merged lists are sorted: the threshold of a full list is its last entry)
__global__ __launch_bounds__(256) void parts_to_lists_kernel(const float* __restrict__ os, const int64_t* __restrict__ ol,
                                                             int64_t n, float* __restrict__ ls, int32_t* __restrict__ li,
                                                             int kp, float* __restrict__ tau) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        ls[i] = os[i];
        li[i] = (int32_t)ol[i];
        if (tau && (i % kp) == kp - 1) tau[i / kp] = ol[i] >= 0 ? os[i] : -INFINITY;
    }
}

// pool source: the per-query sub-pools filled by the fused filter kernel; resets the counters afterwards.
// One wave per query (no cross-wave barriers), small LDS footprint -> many queries resident per CU.
constexpr int kPoolSelThreads = 64;
// A step (one entry level of 64 sub-pools = 64 records of 8 scores) appends at most 512 candidates on top of a full list,
// which sizes the LDS key buffer (cap >= kp + 512) and with it the workgroups per CU.  (LV: unused, kept for the launch sites.)
// rows of the 8 scores of a record relative to its base row (the lane's 4 rows of a 16x16 MFMA tile and of the tile below it)
__device__ __forceinline__ int pool_rec_row(int j) { return j < 4 ? j : j - 4 + kPoolRecHiRow; }

// Walk the records of 64 sub-pools (lane l holds the clamped count c of sub-pool sidx = s0 + l).  The counts are small and uneven
// (1.5 on average, 6-10 at the fullest sub-pool), so walking LEVEL by level (entry e of every sub-pool per round) costs as many
// dependent round trips as the fullest sub-pool has records while most lanes idle.  Instead the records are dealt densely: an
// exclusive wave scan of the counts numbers them 0..T-1, every lane publishes its sub-pool id into slot[] for each of its records, and
// round r lets lane l fetch record r*64 + l = (sub-pool slot[.], entry = number - that sub-pool's first number): ceil(T / 64) rounds
// (~2 instead of ~8), every one with 64 useful records and 8 pushes.  slot[] (one window of 256 records) occupies the LAST 256 bytes
// of the wave's key buffer: the walk reserves 32 keys more than a round can push, so the keys never grow into it (an extra KiB of
// LDS per workgroup would cost a workgroup per CU: 160 KiB / 33 KiB = 4 instead of 5).
constexpr int kSlotWin = 256;
constexpr int kSlotKeys = kSlotWin / 8;

// (c = the sub-pool's record total, cw = its clamped counter word)
__device__ __forceinline__ void walk_subpools(WaveSelector& sel, const uint4* __restrict__ base, int nsubs, int s0, int c, uint32_t cw,
                                              int32_t row_end, unsigned char* slot) {
    const int lane = threadIdx.x & 63;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const int excl = incl - c;
    const int T = __shfl(incl, 63);
    for (int w0 = 0; w0 < T; w0 += kSlotWin) {
        WaveSelector::wave_sync();                       // the previous window's reads are done
        for (int i = 0; i < c; ++i) {
            const int j = excl + i - w0;
            if (j >= 0 && j < kSlotWin) slot[j] = (unsigned char)lane;
        }
        WaveSelector::wave_sync();
        const int wend = min(T, w0 + kSlotWin);
        uint4 n0, n1;
        int32_t nr;
        auto fetch = [&](int r0) {
            const int j = r0 + lane;
            const bool have = j < wend;
            const int sub = have ? (int)slot[j - w0] : 0;
            const int e = pool_entry_of(j - __shfl(excl, sub), (uint32_t)__shfl((int)cw, sub));
            const uint4* rec = base + (int64_t)e * kPoolPlanes * nsubs + s0 + sub;
            n0 = have ? rec[0] : make_uint4(0u, 0u, 0u, 0u);
            n1 = have ? rec[nsubs] : make_uint4(0u, 0u, 0u, 0u);
            nr = have ? (int32_t)rec[2 * nsubs].x : row_end;
        };
        fetch(w0);
        for (int r0 = w0; r0 < wend; r0 += 64) {
            const uint4 p0 = n0, p1 = n1;
            const int32_t r = nr;
            if (r0 + 64 < wend) fetch(r0 + 64);
            sel.reserve(8 * kPoolSelThreads + kSlotKeys);
            sel.push(make_key(__uint_as_float(p0.x), (uint32_t)(r + 0)), r + 0 < row_end);
            sel.push(make_key(__uint_as_float(p0.y), (uint32_t)(r + 1)), r + 1 < row_end);
            sel.push(make_key(__uint_as_float(p0.z), (uint32_t)(r + 2)), r + 2 < row_end);
            sel.push(make_key(__uint_as_float(p0.w), (uint32_t)(r + 3)), r + 3 < row_end);
            sel.push(make_key(__uint_as_float(p1.x), (uint32_t)(r + kPoolRecHiRow + 0)), r + kPoolRecHiRow + 0 < row_end);
            sel.push(make_key(__uint_as_float(p1.y), (uint32_t)(r + kPoolRecHiRow + 1)), r + kPoolRecHiRow + 1 < row_end);
            sel.push(make_key(__uint_as_float(p1.z), (uint32_t)(r + kPoolRecHiRow + 2)), r + kPoolRecHiRow + 2 < row_end);
            sel.push(make_key(__uint_as_float(p1.w), (uint32_t)(r + kPoolRecHiRow + 3)), r + kPoolRecHiRow + 3 < row_end);
        }
    }
}

template <int LV, int QPW>
__global__ __launch_bounds__(kPoolSelThreads * QPW) void select_pools_kernel(const uint4* __restrict__ pool,
                                                                       int32_t* __restrict__ pool_cnt, int nsubs, int64_t nq,
                                                                       int32_t row_end, float* __restrict__ list_s,
                                                                       int32_t* __restrict__ list_i, int kp, int cap,
                                                                       float* __restrict__ tau,
                                                                       int32_t* __restrict__ overflow,
                                                                       int32_t* __restrict__ over_sum,
                                                                       int32_t* __restrict__ qcnt, int dbg,
                                                                       float* __restrict__ tau_opt, int opt_m) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    // QPW independent waves per workgroup, one query each (no workgroup-level synchronisation anywhere): the grid of
    // one-wave workgroups was bound by the workgroup dispatch rate, not by the work
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * QPW + wq;
    if (q >= nq) return;
    WaveSelector sel;
    sel.init(keys + (size_t)wq * cap, kp, cap);
    // the tail of this wave's key buffer — or, on the LDS-sort path (cap > 1024, one wave per workgroup, whose bitonic padding writes up
    // to cap), a window of its own behind it
    unsigned char* slot = cap > WaveSelector::kRegKeys * 64 ? (unsigned char*)(keys + (size_t)QPW * cap) + wq * kSlotWin
                                                            : (unsigned char*)(keys + (size_t)(wq + 1) * cap) - kSlotWin;
    float* ls = list_s + q * kp;
    int32_t* li = list_i + q * kp;
    int32_t* cnt = pool_cnt + q * (int64_t)nsubs;
    // counters of the first 64 sub-pools are requested before the list so that both round trips overlap
    int cn = lane < nsubs ? cnt[lane] : 0;
    if (!(dbg & 4)) sel.load_list(ls, li);
    if (tau) {   // a threshold above the list's own worst key (sharded search: agreed between the ranks) filters the pushes
        const float t0 = tau[q];
        if (t0 > -INFINITY) {
            const uint64_t fk = make_key(t0, 0xfffffffeu);
            if (fk < sel.tau) sel.tau = fk;
        }
    }
    bool over = false;
    int nrec = 0;   // records of this query in this launch (statistics)
    // entry-major, plane-major pools: plane p of level e of all sub-pools is one contiguous run of 16-byte words -> coalesced
    // reads of the few levels in use.  One group of 64 sub-pools x LV entry levels per step.
    const uint4* base = pool + q * (int64_t)kPoolCap * kPoolPlanes * nsubs;
    // One group of 64 sub-pools per step, one entry level per iteration with the next level's record in flight (a single
    // feed site: the selector's compaction is inlined there once).
    for (int s0 = 0; s0 < nsubs; s0 += kPoolSelThreads) {
        const int sidx = s0 + lane;
        int c;
        const uint32_t cw = pool_counts((uint32_t)cn, c, over);
        nrec += c;
        if (sidx < nsubs) cnt[sidx] = 0;
        cn = sidx + kPoolSelThreads < nsubs ? cnt[sidx + kPoolSelThreads] : 0;   // next step's counters
        walk_subpools(sel, base, nsubs, s0, (dbg & 1) ? 0 : c, cw, row_end, slot);
    }
    const bool any_over = __any(over);
    if (dbg & 2) return;
    const float t_prev = (tau_opt && tau) ? tau[q] : -INFINITY;
    sel.finish(ls, li, tau ? tau + q : nullptr, true);
    bool unproven = false;
    if (tau_opt) {
        // optimistic scan (api.hip: fused_rest_chunk_optimistic): the threshold of the NEXT launch = max(so far, m-th best of the new
        // list); after the last launch (opt_m = 0) the check that makes the scheme exact: k' admitted rows at or above every
        // threshold the rows were filtered with, i.e. the list's own threshold >= tau_opt
        const float t_guar = fmaxf(t_prev, sel.n >= kp ? desc_key_to_float((uint32_t)(sel.tau >> 32)) : -INFINITY);
        if (opt_m > 0) {
            const float tm = opt_m >= kp ? t_guar : sel.mth_best(opt_m);
            if (lane == 0) tau_opt[q] = fmaxf(tau_opt[q], tm);
        } else if (opt_m == 0) {
            unproven = !(t_guar >= tau_opt[q]);
        }   // (opt_m < 0: a shard on pooled statistics — the ranks check together, ldot_shard_floor)
    }
    if (qcnt) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nrec += __shfl_xor(nrec, o);
        if (lane == 0) qcnt[q] += nrec;   // (one wave per query: no atomics)
    }
    if (lane == 0 && (any_over || unproven) && atomicExch(&overflow[q], 1) == 0) atomicAdd(over_sum, 1);   // queries counted once
}

// Few queries (the serving shape), many sub-pools: G waves per query, wave g folds sub-pools [g * nsubs / G, (g + 1) * nsubs / G)
// into a partial top-kp list part_[sl][g][q][kp] (int64 labels, -1 = empty); candidates that cannot enter the running list (below
// the query's current threshold) are dropped at the push.  select_lists_kernel then merges the G partial lists with the running
// list.  (A single 256-thread workgroup per query walked 1024 sub-pools with a chain of dependent round trips and LDS bitonic
// sorts: 83 us per launch at one query; this split takes the walk to one step per wave.)
__global__ __launch_bounds__(kPoolSelThreads) void select_pools_parts_kernel(const uint4* __restrict__ pool,
                                                                             int32_t* __restrict__ pool_cnt, int nsubs, int64_t nq,
                                                                             int G, int32_t row_end, int kp, int cap,
                                                                             const float* __restrict__ tau,
                                                                             float* __restrict__ part_s, int64_t* __restrict__ part_l,
                                                                             int32_t* __restrict__ overflow,
                                                                             int32_t* __restrict__ over_sum,
                                                                             int32_t* __restrict__ qcnt) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    const int lane = threadIdx.x & 63;
    const int64_t q = blockIdx.x;
    const int g = blockIdx.y;
    WaveSelector sel;
    sel.init(keys, kp, cap);
    unsigned char* slot = (unsigned char*)(keys + cap) - kSlotWin;
    // nothing below the running list's threshold can enter it: start from that threshold (ties at the threshold score are kept,
    // the merge orders them by row)
    const float t0 = tau[q];
    if (t0 > -INFINITY) sel.tau = make_key(t0, 0xfffffffeu);
    const int per = nsubs / G, s_beg = g * per, s_end = s_beg + per;
    int32_t* cnt = pool_cnt + q * (int64_t)nsubs;
    const uint4* base = pool + q * (int64_t)kPoolCap * kPoolPlanes * nsubs;
    bool over = false;
    int nrec = 0;
    for (int s0 = s_beg; s0 < s_end; s0 += kPoolSelThreads) {
        const int sidx = s0 + lane;
        const int cn = sidx < s_end ? cnt[sidx] : 0;
        if (sidx < s_end) cnt[sidx] = 0;
        int c;
        const uint32_t cw = pool_counts((uint32_t)cn, c, over);
        nrec += c;
        walk_subpools(sel, base, nsubs, s0, sidx < s_end ? c : 0, cw, row_end, slot);
    }
    sel.compact();
    WaveSelector::wave_sync();
    float* ps = part_s + ((int64_t)g * nq + q) * kp;
    int64_t* pl = part_l + ((int64_t)g * nq + q) * kp;
    for (int e = lane; e < kp; e += 64) {
        if (e < sel.n) {
            const uint64_t k = sel.keys[e];
            ps[e] = desc_key_to_float((uint32_t)(k >> 32));
            pl[e] = (int64_t)(uint32_t)(k & 0xffffffffu);
        } else {
            ps[e] = LDOT_PAD_SCORE;
            pl[e] = LDOT_PAD_LABEL;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nrec += __shfl_xor(nrec, o);
    if (lane == 0 && qcnt) atomicAdd(&qcnt[q], nrec);
    if (__any(over) && lane == 0 && atomicExch(&overflow[q], 1) == 0) atomicAdd(over_sum, 1);
}

// Few queries (<= one query block, the serving shape): one 256-thread workgroup per query instead of one wave, so that
// the counters and entry levels of 512 sub-pools are in flight per step (the one-wave walk is a chain of dependent
// global round trips when there is nothing else on the chip to hide them).
__global__ __launch_bounds__(kSelThreads) void select_pools_block_kernel(const uint4* __restrict__ pool,
                                                                         int32_t* __restrict__ pool_cnt, int nsubs,
                                                                         int32_t row_end, float* __restrict__ list_s,
                                                                         int32_t* __restrict__ list_i, int kp, int cap,
                                                                         float* __restrict__ tau,
                                                                         int32_t* __restrict__ overflow,
                                                                         int32_t* __restrict__ over_sum,
                                                                         int32_t* __restrict__ qcnt) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    __shared__ int count;
    const int64_t q = blockIdx.x;
    Selector sel;
    sel.init(keys, &count, kp, cap);
    float* ls = list_s + q * kp;
    int32_t* li = list_i + q * kp;
    const uint4* base = pool + q * (int64_t)kPoolCap * kPoolPlanes * nsubs;
    int32_t* cnt = pool_cnt + q * (int64_t)nsubs;
    bool over = false;
    int nrec = 0;
    // a lone workgroup pays every dependent global round trip in full: the counters of SPT x 256 sub-pools are fetched in one
    // batch (together with the running list), then two entry levels of all of them in another
    constexpr int SPT = 4, LV = 2;
    for (int s0 = 0; s0 < nsubs; s0 += SPT * kSelThreads) {
        int c[SPT];
#pragma unroll
        for (int g = 0; g < SPT; ++g) {
            const int sidx = s0 + g * kSelThreads + threadIdx.x;
            c[g] = sidx < nsubs ? cnt[sidx] : 0;
        }
        if (s0 == 0) sel.load_list(ls, li);
        int cm = 0;
        uint32_t cw[SPT];   // clamped counter words (four lane groups per sub-pool)
#pragma unroll
        for (int g = 0; g < SPT; ++g) {
            const int sidx = s0 + g * kSelThreads + threadIdx.x;
            if (sidx < nsubs) cnt[sidx] = 0;
            int tot;
            cw[g] = pool_counts((uint32_t)c[g], tot, over);
            c[g] = tot;
            nrec += c[g];
            cm = max(cm, c[g]);
        }
        for (int e0 = 0; __syncthreads_or(cm > e0); e0 += LV) {
            uint4 v[SPT][LV][2];
            int32_t rb[SPT][LV];
#pragma unroll
            for (int g = 0; g < SPT; ++g)
#pragma unroll
                for (int u = 0; u < LV; ++u) {
                    const bool have = e0 + u < c[g];
                    const uint4* rec = base + (int64_t)(have ? pool_entry_of(e0 + u, cw[g]) : 0) * kPoolPlanes * nsubs + s0 + g * kSelThreads + threadIdx.x;
                    v[g][u][0] = have ? rec[0] : make_uint4(0u, 0u, 0u, 0u);
                    v[g][u][1] = have ? rec[nsubs] : make_uint4(0u, 0u, 0u, 0u);
                    rb[g][u] = have ? (int32_t)rec[2 * nsubs].x : row_end;
                }
#pragma unroll
            for (int u = 0; u < LV; ++u)
#pragma unroll
                for (int g = 0; g < SPT; ++g) {
                    const uint32_t sc[8] = {v[g][u][0].x, v[g][u][0].y, v[g][u][0].z, v[g][u][0].w,
                                            v[g][u][1].x, v[g][u][1].y, v[g][u][1].z, v[g][u][1].w};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {   // 4 x 256 keys per reservation (the LDS buffer holds a full list + 1024)
                        sel.reserve(4 * kSelThreads);
#pragma unroll
                        for (int j = 4 * h; j < 4 * h + 4; ++j) {
                            const int32_t row = rb[g][u] + pool_rec_row(j);
                            sel.push(make_key(__uint_as_float(sc[j]), (uint32_t)row), row < row_end);
                        }
                    }
                }
        }
    }
    sel.finish(ls, li, tau ? tau + q : nullptr, true);
    if (qcnt) {
        __shared__ int nrec_sh;
        if (threadIdx.x == 0) nrec_sh = 0;
        __syncthreads();
        atomicAdd(&nrec_sh, nrec);
        __syncthreads();
        if (threadIdx.x == 0) qcnt[q] += nrec_sh;
    }
    if (__syncthreads_or(over) && threadIdx.x == 0 && atomicExch(&overflow[q], 1) == 0) atomicAdd(over_sum, 1);
}

// explicit lists source (sharded merge): parts laid out [nparts][nq][k_in], int64 labels (< 2^32-1), -1 = empty
__global__ __launch_bounds__(kSelThreads) void select_lists_kernel(const float* __restrict__ cand_s,
                                                                   const int64_t* __restrict__ cand_l,
                                                                   int64_t part_stride, int64_t part_stride_l, int nparts, int k_in,
                                                                   int k_out, int cap, float* __restrict__ out_s,
                                                                   int64_t* __restrict__ out_l) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    __shared__ int count;
    const int64_t q = blockIdx.x;
    Selector sel;
    sel.init(keys, &count, k_out, cap);
    const int total = nparts * k_in;
    for (int s0 = 0; s0 < total; s0 += kSelThreads) {
        sel.reserve(kSelThreads);
        const int sl = s0 + threadIdx.x;
        bool valid = sl < total;
        uint64_t key = kEmptyKey;
        if (valid) {
            const int p = sl / k_in, e = sl % k_in;
            const int64_t l = cand_l[(int64_t)p * part_stride_l + q * k_in + e];
            valid = l >= 0;
            if (valid) key = make_key(cand_s[(int64_t)p * part_stride + q * k_in + e], (uint32_t)l);
        }
        sel.push(key, valid);
    }
    sel.compact();
    const int n = count;
    for (int e = threadIdx.x; e < k_out; e += kSelThreads) {
        if (e < n) {
            const uint64_t k = keys[e];
            out_s[q * k_out + e] = desc_key_to_float((uint32_t)(k >> 32));
            out_l[q * k_out + e] = (int64_t)(uint32_t)(k & 0xffffffffu);
        } else {
            out_s[q * k_out + e] = LDOT_PAD_SCORE;
            out_l[q * k_out + e] = LDOT_PAD_LABEL;
        }
    }
}

// parts [nparts][nq][kp] (int64 labels, -1 = empty) + the running list (int32 rows, -1 = empty) -> the new running list and its
// threshold, in place: lists_to_parts + select_lists + parts_to_lists of the few-query paths in one launch
__global__ __launch_bounds__(kSelThreads) void merge_parts_into_lists_kernel(const float* __restrict__ part_s,
                                                                             const int64_t* __restrict__ part_l, int nparts,
                                                                             int64_t nq, int kp, int cap, float* __restrict__ list_s,
                                                                             int32_t* __restrict__ list_i, float* __restrict__ tau) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    __shared__ int count;
    const int64_t q = blockIdx.x;
    Selector sel;
    sel.init(keys, &count, kp, cap);
    float* ls = list_s + q * kp;
    int32_t* li = list_i + q * kp;
    sel.load_list(ls, li);
    const int total = nparts * kp;
    for (int s0 = 0; s0 < total; s0 += kSelThreads) {
        sel.reserve(kSelThreads);
        const int sl = s0 + threadIdx.x;
        bool valid = sl < total;
        uint64_t key = kEmptyKey;
        if (valid) {
            const int64_t off = ((int64_t)(sl / kp) * nq + q) * kp + sl % kp;
            const int64_t l = part_l[off];
            valid = l >= 0;
            if (valid) key = make_key(part_s[off], (uint32_t)l);
        }
        sel.push(key, valid);
    }
    sel.finish(ls, li, tau ? tau + q : nullptr, true);   // (thresholds are initialised by init_lists and never go down)
}

// The same merge for at most 4096 keys in all ((nparts + 1) * kp): every key lives in a register (16 per thread), the k'-th best
// score key is found by a bit search (ballots + scalar popcounts, one barrier per bit), the survivors are compacted to LDS and written out
// as a SET (nothing downstream needs the list ordered: the next select takes its threshold from tau[], the re-score sorts by exact
// score).  Only ties on the k'-th score key beyond k' take a sort.  ~13 us instead of ~45 us for 16 parts of k' = 128.
constexpr int kMergeRegKeys = 16;
__global__ __launch_bounds__(kSelThreads) void merge_parts_regs_kernel(const float* __restrict__ part_s, const int64_t* __restrict__ part_l,
                                                                       int nparts, int64_t nq, int kp, float* __restrict__ list_s,
                                                                       int32_t* __restrict__ list_i, float* __restrict__ tau) {
    __shared__ __attribute__((aligned(16))) uint64_t keys[kMergeRegKeys * kSelThreads];
    __shared__ int part[2][kSelThreads / 64];
    __shared__ int count;
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* ls = list_s + q * kp;
    int32_t* li = list_i + q * kp;
    const int total = (nparts + 1) * kp;
    uint32_t hi[kMergeRegKeys], lo[kMergeRegKeys];
#pragma unroll
    for (int v = 0; v < kMergeRegKeys; ++v) {
        const int i = v * kSelThreads + tid;
        hi[v] = lo[v] = 0xffffffffu;   // empty
        if (i < kp) {
            const int32_t r = li[i];
            if (r >= 0) {
                hi[v] = desc_key(ls[i]);
                lo[v] = (uint32_t)r;
            }
        } else if (i < total) {
            const int sl = i - kp;
            const int64_t off = ((int64_t)(sl / kp) * nq + q) * kp + sl % kp;
            const int64_t l = part_l[off];
            if (l >= 0) {
                hi[v] = desc_key(part_s[off]);
                lo[v] = (uint32_t)l;
            }
        }
    }
    if (tid == 0) count = 0;
    // smallest score key t with #(keys <= t) >= kp (all ones when fewer than kp keys exist: everything is kept)
    uint32_t t = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t test = t | ((1u << bit) - 1u);
        int cnt = 0;
#pragma unroll
        for (int v = 0; v < kMergeRegKeys; ++v) cnt += __popcll(__ballot(hi[v] <= test && lo[v] != 0xffffffffu));
        int* p = part[bit & 1];
        if (lane == 0) p[wave] = cnt;
        __syncthreads();
        int tot = 0;
#pragma unroll
        for (int w = 0; w < kSelThreads / 64; ++w) tot += p[w];
        if (tot < kp) t |= 1u << bit;
    }
    // survivors -> LDS
#pragma unroll
    for (int v = 0; v < kMergeRegKeys; ++v) {
        const bool keep = lo[v] != 0xffffffffu && hi[v] <= t;
        const unsigned long long m = __ballot(keep);
        if (m) {
            int base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) base = atomicAdd(&count, __popcll(m));
            base = __shfl(base, leader);
            if (keep) keys[base + __popcll(m & ((1ull << lane) - 1ull))] = ((uint64_t)hi[v] << 32) | lo[v];
        }
    }
    __syncthreads();
    const int n = count;
    if (n > kp) {   // ties on the k'-th score key: order them (score, then row) and keep the first k'
        int P = 2;
        while (P < n) P <<= 1;
        for (int i = n + tid; i < P; i += kSelThreads) keys[i] = kEmptyKey;
        __syncthreads();
        bitonic_sort_lds(keys, P);
    }
    for (int e = tid; e < kp; e += kSelThreads) {
        if (e < n) {
            const uint64_t k = keys[e];
            ls[e] = desc_key_to_float((uint32_t)(k >> 32));
            li[e] = (int32_t)(uint32_t)k;
        } else {
            ls[e] = LDOT_PAD_SCORE;
            li[e] = -1;
        }
    }
    if (tau && tid == 0) tau[q] = fmaxf(tau[q], n >= kp ? desc_key_to_float(t) : -INFINITY);   // (never down: see WaveSelector::finish)
}

int launch_merge_parts_into_lists(const float* part_s, const int64_t* part_l, int nparts, int64_t nq, int kp, float* list_s,
                                  int32_t* list_i, float* tau, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    if ((int64_t)(nparts + 1) * kp <= kMergeRegKeys * kSelThreads) {
        hipLaunchKernelGGL(merge_parts_regs_kernel, dim3((unsigned)nq), dim3(kSelThreads), 0, st, part_s, part_l, nparts, nq, kp, list_s,
                           list_i, tau);
        LDOT_HIP_CHECK(hipGetLastError());
        return LDOT_OK;
    }
    const int cap = select_cap(kp, 1024, kSelThreads);
    hipLaunchKernelGGL(merge_parts_into_lists_kernel, dim3((unsigned)nq), dim3(kSelThreads), (size_t)cap * 8, st, part_s, part_l,
                       nparts, nq, kp, cap, list_s, list_i, tau);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_init_lists(float* list_s, int32_t* list_i, int64_t n, float* tau, int64_t nq, int64_t nq_pad, hipStream_t st) {
    const int64_t m = std::max(n, tau ? nq_pad : (int64_t)0);
    if (m <= 0) return LDOT_OK;
    hipLaunchKernelGGL(init_lists_kernel, dim3((unsigned)((m + kSelThreads - 1) / kSelThreads)), dim3(kSelThreads), 0, st,
                       list_s, list_i, n, tau, nq, nq_pad);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_select_dense(const float* S, int64_t lds_elems, int64_t nq, int64_t ncols, int64_t idx_base,
                        float* list_s, int32_t* list_i, int kp, float* tau, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    if (kp <= 256 && ncols <= 1024 && nq > 64) {   // very short rows (Flickr-1k sized index): the whole row in registers
        // (measured: 53 -> 35 us at 5000 x 1000; with 64 / 128 values per lane — 4096 / 8192 columns — the VALU compare-and-count of
        // the bit search and the register footprint (2 waves per SIMD) make it SLOWER than the streaming selector: 198 vs 170 us at
        // 10240 x 4096, so longer rows keep the streaming path)
        constexpr int QPW = 4;
        hipLaunchKernelGGL((select_dense_regs_kernel<16, QPW>), dim3((unsigned)((nq + QPW - 1) / QPW)), dim3(64 * QPW), 0, st, S,
                           lds_elems, nq, (int)ncols, idx_base, list_s, list_i, kp, tau);
        LDOT_HIP_CHECK(hipGetLastError());
        return LDOT_OK;
    }
    if (kp + 256 <= WaveSelector::kRegKeys * 64 && nq > 64 && ncols <= 64 * 8 * 10) {   // whole row in registers, run-maxima threshold
        constexpr int QPW = 4;
        const int wcap = WaveSelector::kRegKeys * 64;
        if (ncols <= 64 * 8 * 8)
            hipLaunchKernelGGL((select_dense_runs_kernel<8, QPW>), dim3((unsigned)((nq + QPW - 1) / QPW)), dim3(64 * QPW),
                               (size_t)wcap * 8 * QPW, st, S, lds_elems, nq, (int)ncols, idx_base, list_s, list_i, kp, wcap, tau);
        else
            hipLaunchKernelGGL((select_dense_runs_kernel<10, QPW>), dim3((unsigned)((nq + QPW - 1) / QPW)), dim3(64 * QPW),
                               (size_t)wcap * 8 * QPW, st, S, lds_elems, nq, (int)ncols, idx_base, list_s, list_i, kp, wcap, tau);
        LDOT_HIP_CHECK(hipGetLastError());
        return LDOT_OK;
    }
    if (kp + 256 <= WaveSelector::kRegKeys * 64 && nq > 64) {   // wave-per-query register selection (k' <= 512)
        constexpr int QPW = 4;
        const int wcap = WaveSelector::kRegKeys * 64;
        hipLaunchKernelGGL((select_dense_wave_kernel<QPW>), dim3((unsigned)((nq + QPW - 1) / QPW)), dim3(64 * QPW),
                           (size_t)wcap * 8 * QPW, st, S, lds_elems, nq, ncols, idx_base, list_s, list_i, kp, wcap, tau);
        LDOT_HIP_CHECK(hipGetLastError());
        return LDOT_OK;
    }
    if (select_big_dense_ok(kp, ncols, idx_base))   // long lists (round 6): pivot + one streaming pass + bit search in registers
        return launch_select_big_dense(S, lds_elems, nq, ncols, idx_base, list_s, list_i, kp, tau, st);
    const int cap = select_cap(kp, 2048, 4 * kSelThreads);   // 16 KiB of keys for kp <= 1024
    hipLaunchKernelGGL(select_dense_kernel, dim3((unsigned)nq), dim3(kSelThreads), (size_t)cap * 8, st, S, lds_elems,
                       ncols, idx_base, list_s, list_i, kp, cap, tau);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_select_dense_parts(const float* S, int64_t lds_elems, int64_t nq, int64_t ncols, int64_t seg_cols,
                              int64_t idx_base, int kp, float* part_s, int64_t* part_l, hipStream_t st) {
    if (nq <= 0 || ncols <= 0) return LDOT_OK;
    const int cap = select_cap(kp, 2048, 4 * kSelThreads);
    const unsigned nseg = (unsigned)((ncols + seg_cols - 1) / seg_cols);
    hipLaunchKernelGGL(select_dense_parts_kernel, dim3((unsigned)nq, nseg), dim3(kSelThreads), (size_t)cap * 8, st, S,
                       lds_elems, nq, ncols, seg_cols, idx_base, kp, cap, part_s, part_l);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_lists_to_parts(const float* list_s, const int32_t* list_i, int64_t n, float* part_s, int64_t* part_l,
                          hipStream_t st) {
    if (n <= 0) return LDOT_OK;
    hipLaunchKernelGGL(lists_to_parts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, list_s, list_i, n,
                       part_s, part_l);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_parts_to_lists(const float* out_s, const int64_t* out_l, int64_t n, float* list_s, int32_t* list_i, int kp,
                          float* tau, hipStream_t st) {
    if (n <= 0) return LDOT_OK;
    hipLaunchKernelGGL(parts_to_lists_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out_s, out_l, n,
                       list_s, list_i, kp, tau);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_select_pools(const uint4* pool, const int32_t* pool_cnt, int nsubs, int64_t nq, int32_t row_end, float* list_s,
                        int32_t* list_i, int kp, float* tau, int32_t* overflow_flags, int32_t* over_sum, int32_t* qcnt,
                        hipStream_t st, float* tau_opt, int opt_m) {
    if (nq <= 0) return LDOT_OK;
    LDOT_REQUIRE(tau_opt == nullptr || nq > 256, LDOT_EINVAL, "optimistic thresholds are a large-batch schedule");
    if (nq <= 256) {   // few queries: block-per-query walk
        const int bcap = select_cap(kp, 2048, 4 * kSelThreads);   // a step appends up to 4 x 256 candidates on top of a full list
        hipLaunchKernelGGL(select_pools_block_kernel, dim3((unsigned)nq), dim3(kSelThreads), (size_t)bcap * 8, st, pool,
                           (int32_t*)pool_cnt, nsubs, row_end, list_s, list_i, kp, bcap, tau, overflow_flags, over_sum, qcnt);
        LDOT_HIP_CHECK(hipGetLastError());
        return LDOT_OK;
    }
    const int cap = select_cap(kp, 1024, 8 * kPoolSelThreads + kSlotKeys);   // a round appends up to 8 x 64 candidates on top of a full list (+ the slot window)
    int dbg = 0;
#ifdef LDOT_ABLATION
    static int dbg_env = -1;   // LDOT_DEBUG_SEL: ablation bits (ablation builds only; results are then meaningless)
    if (dbg_env < 0) {
        const char* e = getenv("LDOT_DEBUG_SEL");
        dbg_env = e ? atoi(e) : 0;
    }
    dbg = dbg_env;
#endif
    if (cap <= WaveSelector::kRegKeys * 64) {   // register selection path: 4 independent query-waves per workgroup
        constexpr int QPW = 4;
        hipLaunchKernelGGL((select_pools_kernel<4, QPW>), dim3((unsigned)((nq + QPW - 1) / QPW)),
                           dim3(kPoolSelThreads * QPW), (size_t)cap * 8 * QPW, st, pool, (int32_t*)pool_cnt, nsubs, nq,
                           row_end, list_s, list_i, kp, cap, tau, overflow_flags, over_sum, qcnt, dbg, tau_opt, opt_m);
    } else {   // long lists (round 6): a 256-thread workgroup per query, bit search in registers (select_big.hip)
        return launch_select_big_pools(pool, (int32_t*)pool_cnt, nsubs, nq, row_end, list_s, list_i, kp, tau, overflow_flags, over_sum, qcnt, st,
                                       tau_opt, opt_m);
    }
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_select_pools_parts(const uint4* pool, const int32_t* pool_cnt, int nsubs, int64_t nq, int G, int32_t row_end, int kp,
                              const float* tau, float* part_s, int64_t* part_l, int32_t* overflow_flags, int32_t* over_sum,
                              int32_t* qcnt, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    const int cap = select_cap(kp, 1024, 8 * kPoolSelThreads + kSlotKeys);
    LDOT_REQUIRE(cap <= WaveSelector::kRegKeys * 64 && nsubs % G == 0, LDOT_EINVAL, "select_pools_parts: unsupported shape");
    hipLaunchKernelGGL(select_pools_parts_kernel, dim3((unsigned)nq, (unsigned)G), dim3(kPoolSelThreads), (size_t)cap * 8, st, pool,
                       (int32_t*)pool_cnt, nsubs, nq, G, row_end, kp, cap, tau, part_s, part_l, overflow_flags, over_sum, qcnt);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// ---- sharded search: what the ranks exchange after the local warm-up (ldot_index_search_warmup / _scan) ---------------------------
// stat[q] = the query's threshold (its k'-th best warm-up score), stat[nq + q] = MINUS its m-th best warm-up score, m = ceil(k'/parts)
// (+inf when the list holds fewer than m rows).  After an all-reduce(MAX) over the ranks, max(stat[q], -stat[nq + q]) is a lower bound
// of the GLOBAL k'-th best score: some rank has k' rows at or above the first term, and EVERY rank has m rows at or above the
// second (parts * m >= k').  One wave per query; the list is a set of <= 64 * R keys held in registers, the m-th best by bit search.
template <int R>
__global__ __launch_bounds__(256) void list_stats_kernel(const float* __restrict__ ls, const int32_t* __restrict__ li, int kp, int64_t nq,
                                                         int m, const float* __restrict__ tau, float* __restrict__ stat, int slots,
                                                         const float* __restrict__ level, const int32_t* __restrict__ redone) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    uint32_t key[R];
    int nvalid = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = r * 64 + lane;
        const bool valid = e < kp && li[q * kp + e] >= 0;
        key[r] = valid ? desc_key(ls[q * kp + e]) : 0xffffffffu;
        nvalid += __popcll(__ballot(valid));
    }
    float tm = m > 0 ? -INFINITY : INFINITY;   // (m = 0: this list vouches for no row — the neutral element of the reduction)
    if (m > 0 && nvalid >= m) {
        uint32_t T = 0;   // smallest key with #(keys <= T) >= m  (descending keys: the m-th best score)
        for (int b = 31; b >= 0; --b) {
            const uint32_t trial = T | ((1u << b) - 1u);
            int c = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) c += __popcll(__ballot(key[r] <= trial && key[r] != 0xffffffffu));
            if (c < m) T |= 1u << b;
        }
        tm = desc_key_to_float(T);
    }
    if (lane == 0) {
        stat[q] = tau[q];
        stat[nq + q] = -tm;
        // third slot (end of a shard's scan): the list holds every row of the shard at or above this score.  A complete top-k' list
        // (no level, or a query the recovery searched again) reports -inf: the floor is never below its k'-th best anyway
        if (slots > 2) stat[2 * nq + q] = (level && !(redone && redone[q])) ? level[q] : -INFINITY;
    }
}

int launch_list_stats(const float* list_s, const int32_t* list_i, int kp, int64_t nq, int m, const float* tau, float* stat, hipStream_t st,
                      int slots, const float* level, const int32_t* redone) {
    if (nq <= 0) return LDOT_OK;
    const dim3 grid((unsigned)((nq + 3) / 4)), block(256);
    if (kp <= 128)
        hipLaunchKernelGGL(list_stats_kernel<2>, grid, block, 0, st, list_s, list_i, kp, nq, m, tau, stat, slots, level, redone);
    else if (kp <= 256)
        hipLaunchKernelGGL(list_stats_kernel<4>, grid, block, 0, st, list_s, list_i, kp, nq, m, tau, stat, slots, level, redone);
    else if (kp <= 1024)
        hipLaunchKernelGGL(list_stats_kernel<16>, grid, block, 0, st, list_s, list_i, kp, nq, m, tau, stat, slots, level, redone);
    else
        hipLaunchKernelGGL(list_stats_kernel<48>, grid, block, 0, st, list_s, list_i, kp, nq, m, tau, stat, slots, level, redone);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// the floor of a sharded search from the all-reduced (MAX) end-of-scan statistics of its ranks (3 x nq floats, ldot.h):
// floor = max(largest k'-th best of a rank, smallest vouched-for order statistic of the ranks) <= the global k'-th best, and — for the
// check of the pooled statistics — the number of THIS shard's list entries at or above the largest level of any shard (all of its rows
// up there are in the list): the ranks add these up, k' or more of them prove that no shard lost a row of the global top k'.
// One wave per query.  (No pooled level anywhere: nothing to prove, the count reports k'.)
__global__ __launch_bounds__(256) void shard_floor_kernel(const float* __restrict__ stat, int64_t nq, const float* __restrict__ ls,
                                                          const int32_t* __restrict__ li, int kp, float* __restrict__ floor_out,
                                                          int32_t* __restrict__ count_out) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float level = stat[2 * nq + q];
    int c = 0;
    if (level > -INFINITY)
        for (int e = lane; e < kp; e += 64) c += (li[q * kp + e] >= 0 && ls[q * kp + e] >= level) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) {
        floor_out[q] = fmaxf(stat[q], -stat[nq + q]);
        count_out[q] = level > -INFINITY ? c : kp;
    }
}

int launch_shard_floor(const float* stat, int64_t nq, const float* list_s, const int32_t* list_i, int kp, float* floor_out,
                       int32_t* count_out, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    hipLaunchKernelGGL(shard_floor_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, stat, nq, list_s, list_i, kp, floor_out, count_out);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// ---- optimistic thresholds of the fused scan (api.hip: fused_rest_chunk) ---------------------------------------------------------------
// tau_opt[q] = max(tau_opt[q], m-th best score of q's list); m = kp takes the list's own threshold tau[q] (its k'-th best, -inf while
// the list is not full).  Pad queries (q >= nq) stay at +inf.
template <int R>
__global__ __launch_bounds__(256) void tau_opt_kernel(const float* __restrict__ ls, const int32_t* __restrict__ li, int kp, int64_t nq, int m,
                                                      const float* __restrict__ tau, float* __restrict__ tau_opt) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    float tm = tau[q];
    if (m < kp) {
        uint32_t key[R];
        int nvalid = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int e = r * 64 + lane;
            const bool valid = e < kp && li[q * kp + e] >= 0;
            key[r] = valid ? desc_key(ls[q * kp + e]) : 0xffffffffu;
            nvalid += __popcll(__ballot(valid));
        }
        tm = -INFINITY;
        if (nvalid >= m) {
            uint32_t T = 0;   // smallest key with #(keys <= T) >= m  (descending keys: the m-th best score)
            for (int b = 31; b >= 0; --b) {
                const uint32_t trial = T | ((1u << b) - 1u);
                int c = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) c += __popcll(__ballot(key[r] <= trial && key[r] != 0xffffffffu));
                if (c < m) T |= 1u << b;
            }
            tm = desc_key_to_float(T);
        }
    }
    if (lane == 0) tau_opt[q] = fmaxf(tau_opt[q], tm);
}

int launch_tau_opt(const float* list_s, const int32_t* list_i, int kp, int64_t nq, int m, const float* tau, float* tau_opt, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    const dim3 grid((unsigned)((nq + 3) / 4)), block(256);
    if (kp <= 128)
        hipLaunchKernelGGL(tau_opt_kernel<2>, grid, block, 0, st, list_s, list_i, kp, nq, m, tau, tau_opt);
    else if (kp <= 256)
        hipLaunchKernelGGL(tau_opt_kernel<4>, grid, block, 0, st, list_s, list_i, kp, nq, m, tau, tau_opt);
    else if (kp <= 1024)
        hipLaunchKernelGGL(tau_opt_kernel<16>, grid, block, 0, st, list_s, list_i, kp, nq, m, tau, tau_opt);
    else
        hipLaunchKernelGGL(tau_opt_kernel<48>, grid, block, 0, st, list_s, list_i, kp, nq, m, tau, tau_opt);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// per-search state of a fused scan in one launch (instead of two memset nodes and a kernel): the per-query record counters and the
// overflow summary zeroed, tau_opt = -inf for the queries and +inf for the pad rows of the last query block (they never produce candidates)
__global__ __launch_bounds__(256) void init_fused_scan_kernel(float* __restrict__ tau_opt, int32_t* __restrict__ qcnt,
                                                              int32_t* __restrict__ over_sum, int64_t nq, int64_t nq_pad) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q < nq_pad) {
        tau_opt[q] = q < nq ? -INFINITY : INFINITY;
        qcnt[q] = 0;
    }
    if (q < 4) over_sum[q] = 0;
}

int launch_init_fused_scan(float* tau_opt, int32_t* qcnt, int32_t* over_sum, int64_t nq, int64_t nq_pad, hipStream_t st) {
    if (nq_pad <= 0) return LDOT_OK;
    hipLaunchKernelGGL(init_fused_scan_kernel, dim3((unsigned)((nq_pad + 255) / 256)), dim3(256), 0, st, tau_opt, qcnt, over_sum, nq, nq_pad);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// End of an optimistic scan: a query's list is its exact top-k' iff at least k' admitted rows score at or above every threshold its rows
// were filtered with, i.e. its final k'-th best (tau) >= tau_opt.  The others are flagged like pool overflows (redo_flagged searches
// them again on guaranteed thresholds).
__global__ __launch_bounds__(256) void verify_tau_opt_kernel(const float* __restrict__ tau, const float* __restrict__ tau_opt, int64_t nq,
                                                             int32_t* __restrict__ overflow, int32_t* __restrict__ over_sum) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    if (!(tau[q] >= tau_opt[q]) && atomicExch(&overflow[q], 1) == 0) atomicAdd(over_sum, 1);
}

int launch_verify_tau_opt(const float* tau, const float* tau_opt, int64_t nq, int32_t* overflow, int32_t* over_sum, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    hipLaunchKernelGGL(verify_tau_opt_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, tau, tau_opt, nq, overflow, over_sum);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// "no warm-up on this rank": neutral elements of the exchange (a rank without a statistic must not raise anybody's threshold)
__global__ __launch_bounds__(256) void neutral_stats_kernel(int64_t nq, float* __restrict__ stat) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    stat[q] = -INFINITY;
    stat[nq + q] = INFINITY;
}

int launch_neutral_stats(int64_t nq, float* stat, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    hipLaunchKernelGGL(neutral_stats_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, nq, stat);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// tau[q] = max(tau[q], stat[q], -stat[nq + q]): the agreed threshold becomes the floor of this shard's candidate pass
__global__ __launch_bounds__(256) void apply_stats_kernel(int64_t nq, const float* __restrict__ stat, float* __restrict__ tau) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    tau[q] = fmaxf(tau[q], fmaxf(stat[q], -stat[nq + q]));
}

int launch_apply_stats(int64_t nq, const float* stat, float* tau, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    hipLaunchKernelGGL(apply_stats_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, nq, stat, tau);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_select_lists(const float* cand_s, const int64_t* cand_l, int64_t part_stride, int64_t part_stride_l, int nparts, int k_in,
                        int64_t nq, int k_out, float* out_s, int64_t* out_l, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    const int cap = select_cap(k_out, 1024, kSelThreads);
    hipLaunchKernelGGL(select_lists_kernel, dim3((unsigned)nq), dim3(kSelThreads), (size_t)cap * 8, st, cand_s, cand_l,
                       part_stride, part_stride_l, nparts, k_in, k_out, cap, out_s, out_l);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

}  // namespace ldot
