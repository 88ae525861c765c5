// Exact fp32 re-score of the bf16-generated candidates + final ordering (score desc, row asc) + top-k output.
// The reference's scorer is fp32 end to end (faiss IndexFlatIP, dvl/indexer/faiss_indexers.py:83); candidates
// come from the bf16 MFMA pass with a safety margin (k' > k) and every reported score is recomputed here from
// the fp32 master copy of the rows, which is what makes rank order and scores match the fp32 reference.
#include "bitonic.h"
#include "kernels.h"

namespace ldot {


// THREADS = 256 for batches (many queries in flight per CU), 1024 for a handful of queries (16 waves gather the rows of one query)
template <int kRsThreads>
__global__ __launch_bounds__(kRsThreads) void rescore_kernel(const float* __restrict__ q32, int64_t ldq,
                                                             const float* __restrict__ x32, int64_t ldx, int dpad,
                                                             const float* __restrict__ list_s,
                                                             const int32_t* __restrict__ list_i, int kp, int k,
                                                             int do_rescore, const float* __restrict__ floor,
                                                             float* __restrict__ out_s,
                                                             int64_t* __restrict__ out_l, RescoreOut lay, int64_t nq) {
    // [next power of two >= k']: sized per launch — a fixed 32-KiB buffer (kMaxKp keys) held a CU to five workgroups whatever k' was, and the
    // gather of a cache-resident index (Flickr / COCO sized, the mining searches) lives on loads in flight
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    static_assert(kMaxKp <= 4096, "re-score key buffer");
    // Block b is observed on XCD b % 8 (speed only).  Neighbouring queries of a retrieval evaluation are captions of the same image
    // (dvl/trainer.py:130-154 walks the dataset in order) and share most of their candidate rows: sixteen consecutive queries go to ONE XCD,
    // so that the rows they share are gathered from that XCD's L2 instead of eight times from the Infinity Cache / HBM.  Within a group of
    // 128 blocks, block r takes query (r % 8) * 16 + r / 8.
    int64_t q = blockIdx.x;
    if (gridDim.x >= 128) {
        const int64_t r = q & 127;
        q = (q - r) + (r & 7) * 16 + (r >> 3);
    }
    if (q >= nq) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* qrow = q32 + q * ldq;
    // 1. compact the live candidates (valid row, and at or above the floor of a sharded search: candidates below the best
    //    k'-th candidate score of any shard cannot be among the global k' best) to keys[0..m): {candidate score bits, row}.
    //    The order of the compaction does not matter (the final sort orders by exact score, then row).
    __shared__ int m_sh;
    if (threadIdx.x == 0) m_sh = 0;
    __syncthreads();
    const float fl = floor ? floor[q] : -INFINITY;
    for (int e0 = 0; e0 < kp; e0 += kRsThreads) {
        const int e = e0 + threadIdx.x;
        int32_t r = -1;
        float cs = 0.f;
        if (e < kp) {
            r = list_i[q * kp + e];
            cs = list_s[q * kp + e];
        }
        const bool live = r >= 0 && !(cs < fl);
        const unsigned long long mask = __ballot(live);
        int base = 0;
        if (lane == 0 && mask) base = atomicAdd(&m_sh, __popcll(mask));
        base = __shfl(base, 0);
        if (live) keys[base + __popcll(mask & ((1ull << lane) - 1ull))] = ((uint64_t)__float_as_uint(cs) << 32) | (uint32_t)r;
    }
    __syncthreads();
    const int m = m_sh;
    int P = 2;
    while (P < m) P <<= 1;
    // 2. exact scores.  (The gather runs at the memory roofline — 3.9 GB in 0.65 ms = 6.0 TB/s for 10k queries over a 1M-row index,
    //    ~13 TB/s when the rows sit in L2 / Infinity Cache: a variant with the query row in registers and 8 x 3 KiB of row
    //    segments in flight per wave (177 VGPRs, 2 waves per SIMD) was SLOWER, 1.17 ms — more bytes in flight do not help.)
    //    Four candidates per wave iteration: their row gathers are independent, so 4x the loads are in
    //    flight; the arithmetic of one candidate (4 fmaf chains over the columns lane*4 + 256*i, pairwise sum, xor-shuffle
    //    tree) does not depend on the grouping.
    constexpr int U = 4;
    for (int e0 = wave * U; e0 < m; e0 += (kRsThreads / 64) * U) {
        int32_t r[U];
        float s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t kv = (e0 + u < m) ? keys[e0 + u] : ~0ull;
            r[u] = (e0 + u < m) ? (int32_t)(uint32_t)kv : -1;
            s[u] = __uint_as_float((uint32_t)(kv >> 32));
        }
        if (do_rescore) {
            float acc[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u][0] = acc[u][1] = acc[u][2] = acc[u][3] = 0.f;
            for (int c = lane * 4; c < dpad; c += 256) {
                const f32x4 qv = *(const f32x4*)(qrow + c);
                f32x4 xv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {   // r[u] is wave-uniform
                    xv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (r[u] >= 0) xv[u] = *(const f32x4*)(x32 + (int64_t)r[u] * ldx + c);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    acc[u][0] = fmaf(xv[u][0], qv[0], acc[u][0]);
                    acc[u][1] = fmaf(xv[u][1], qv[1], acc[u][1]);
                    acc[u][2] = fmaf(xv[u][2], qv[2], acc[u][2]);
                    acc[u][3] = fmaf(xv[u][3], qv[3], acc[u][3]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s[u] = (acc[u][0] + acc[u][1]) + (acc[u][2] + acc[u][3]);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s[u] += __shfl_xor(s[u], o);
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (e0 + u < m)   // (a shuffled index reports — and breaks ties by — the row's label, not its place in the store)
                    keys[e0 + u] = ((uint64_t)desc_key(s[u]) << 32) | (uint32_t)(lay.label_map ? lay.label_map[r[u]] : r[u]);
        }
    }
    __syncthreads();
    for (int e = m + threadIdx.x; e < P; e += kRsThreads) keys[e] = ~0ull;
    __syncthreads();
    bitonic_sort_lds(keys, P);
    // dense outputs [nq][k], or blocks of block_rows queries (the send buffer of a sharded search: one block per destination rank)
    const int64_t blk = lay.block_rows > 0 ? q / lay.block_rows : 0, in_blk = lay.block_rows > 0 ? q - blk * lay.block_rows : q;
    float* os = out_s + blk * lay.stride_s + in_blk * k;
    int64_t* ol = out_l + blk * lay.stride_l + in_blk * k;
    for (int e = threadIdx.x; e < k; e += kRsThreads) {
        const uint64_t key = (e < m) ? keys[e] : ~0ull;
        if (key != ~0ull) {
            os[e] = desc_key_to_float((uint32_t)(key >> 32));
            ol[e] = (int64_t)(uint32_t)(key & 0xffffffffu) + lay.label_base;
        } else {
            os[e] = LDOT_PAD_SCORE;
            ol[e] = LDOT_PAD_LABEL;
        }
    }
}

// LDOT_OPT_RESULT_SET: the caller consumes the top-k SET (hard-negative mining keeps the ids and samples from them, dvl/hn.py:54-63), not the
// scores and not the order.  Which candidates are among the k best by exact score is then already decided by the bf16 candidate scores for all
// but a band around the k-th: with |exact - bf16| <= E for every candidate (E = the bound of LDOT_OPT_VERIFY: c * 2^-8 * |q| * max|x| / sqrt(d)) and
// c_(k) the k-th largest candidate score,
//     c_i > c_(k) + 2E  =>  every row that beats i exactly has a candidate score above c_(k): fewer than k of them  =>  i is IN,
//     c_i < c_(k) - 2E  =>  the k rows at or above c_(k) all beat i exactly                                        =>  i is OUT,
// and only the candidates in between are gathered from the fp32 master copy (3 KiB per row — the cost of a mining search at top-1000) and
// ordered exactly; the k - #IN best of them complete the set.  Under that bound the labels are the set the full re-score reports (ties at
// the boundary broken by the lower label, as there).  The ORDER of the k outputs is unspecified: the IN candidates (reported score = their
// bf16-input candidate score) in list order, then the band's winners by exact score.  stats[2 s] += candidates gathered, stats[2 s + 1] += live candidates (kSetStatSlots slots s).
//
// No sort anywhere: a workgroup (ONE wave for k' <= 256: no barriers, 32 queries in flight per CU; four waves beyond) holds the k' candidates in
// registers, finds c_(k) by a bit search over their descending keys (count-and-reduce per bit), classifies, gathers the band's rows and
// ranks the band by counting.  (The first version sorted the list twice in a 32-KiB key buffer: 4.2 ms for 145 000 queries at k' = 96,
// as long as the full re-score — LDS occupancy, not the 12 % of the rows it gathered: profiles/r06_mining_kernels_first.txt.)
template <int T>
__device__ __forceinline__ int set_block_sum(int v, int* red, int slot) {
    // sum over the workgroup; `slot` alternates between calls so that ONE barrier per call is enough.  (DPP wave sum: ~50 cycles of VALU; six
    // ds_bpermute round trips per bit of the search were most of a one-wave workgroup's time)
    v = wave_sum_dpp(v);
    if (T == 64) return v;
    if ((threadIdx.x & 63) == 0) red[slot * (T / 64) + (threadIdx.x >> 6)] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) s += red[slot * (T / 64) + w];
    return s;
}

template <int T, int VP>   // VP = candidates per thread: k' <= T * VP
__global__ __launch_bounds__(T) void rescore_set_kernel(const float* __restrict__ q32, int64_t ldq, const float* __restrict__ x32,
                                                        int64_t ldx, int dpad, int d, const float* __restrict__ list_s,
                                                        const int32_t* __restrict__ list_i, int kp, int k,
                                                        const float* __restrict__ max_norm, float band_c, float* __restrict__ out_s,
                                                        int64_t* __restrict__ out_l, const int32_t* __restrict__ label_map,
                                                        unsigned long long* __restrict__ stats, int64_t nq) {
    extern __shared__ __attribute__((aligned(16))) uint64_t band[];   // [next power of two >= k']: {stored row} -> {exact key, label}
    __shared__ int red[2 * (T / 64) + 1];
    __shared__ int cnt_in, cnt_band;
    __shared__ float qq_sh[T / 64];
    int64_t q = blockIdx.x;
    if (T > 64 && gridDim.x >= 128) {   // (the block -> query map of rescore_kernel: sixteen consecutive queries on one XCD)
        const int64_t r = q & 127;
        q = (q - r) + (r & 7) * 16 + (r >> 3);
    }
    if (q >= nq) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* qrow = q32 + q * ldq;
    if (threadIdx.x == 0) cnt_in = cnt_band = 0;
    // candidates -> registers (descending key of the candidate score; 0xffffffff = not a candidate)
    uint32_t key[VP];
    int32_t row[VP];
    int mine = 0;
#pragma unroll
    for (int v = 0; v < VP; ++v) {
        const int e = v * T + threadIdx.x;
        const int el = e < kp ? e : kp - 1;          // (both loads unconditional: a load that depends on the other one's result is a second round trip)
        const int32_t lr = list_i[q * kp + el];
        const float lsv = list_s[q * kp + el];
        row[v] = e < kp ? lr : -1;
        key[v] = row[v] >= 0 ? desc_key(lsv) : 0xffffffffu;
        mine += row[v] >= 0;
    }
    float qq = 0.f;   // |q|^2 (the band is relative to the query's norm; rows are zero-padded to dpad, a multiple of 4)
    for (int c = threadIdx.x * 4; c < dpad; c += T * 4) {
        const f32x4 v4 = *(const f32x4*)(qrow + c);
        qq += (v4[0] * v4[0] + v4[1] * v4[1]) + (v4[2] * v4[2] + v4[3] * v4[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o);
    if (lane == 0) qq_sh[wave] = qq;
    const int m = set_block_sum<T>(mine, red, 0);   // (its barrier also publishes cnt_* and qq_sh)
    float* os = out_s + q * k;
    int64_t* ol = out_l + q * k;
    if (m <= k) {   // everything there is belongs to the set
#pragma unroll
        for (int v = 0; v < VP; ++v) {
            const bool have = row[v] >= 0;
            const unsigned long long hm = __ballot(have);
            int base = 0;
            if (lane == 0 && hm) base = atomicAdd(&cnt_in, __popcll(hm));
            base = __shfl(base, 0);
            if (have) {
                const int pos = base + __popcll(hm & ((1ull << lane) - 1ull));
                os[pos] = desc_key_to_float(key[v]);
                ol[pos] = label_map ? label_map[row[v]] : row[v];
            }
        }
        for (int e = m + threadIdx.x; e < k; e += T) {
            os[e] = LDOT_PAD_SCORE;
            ol[e] = LDOT_PAD_LABEL;
        }
        if (threadIdx.x == 0 && stats) atomicAdd(stats + 2 * (blockIdx.x & (kSetStatSlots - 1)) + 1, (unsigned long long)m);
        return;
    }
    // ---- c_(k): the k-th smallest descending key, by bit search (the bits all candidates share are skipped) ---------------------
    uint32_t kth = 0;
    {
        uint32_t all_and = 0xffffffffu, all_or = 0u;
#pragma unroll
        for (int v = 0; v < VP; ++v) {
            all_and &= key[v];                        // (a non-candidate is all ones: neutral)
            all_or |= row[v] >= 0 ? key[v] : 0u;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            all_and &= __shfl_xor(all_and, o);
            all_or |= __shfl_xor(all_or, o);
        }
        if (T > 64) {
            __shared__ uint32_t ru[2 * (T / 64)];   // (not `red`: a slow wave may still be reading the sum above)
            if (lane == 0) {
                ru[wave] = all_and;
                ru[T / 64 + wave] = all_or;
            }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < T / 64; ++w) {
                all_and &= ru[w];
                all_or |= ru[T / 64 + w];
            }
        }
        const uint32_t diff = all_and ^ all_or;
        const int hb = diff ? 31 - __clz((int)diff) : -1;      // highest bit in which two candidates differ
        kth = hb >= 31 ? 0u : hb < 0 ? all_or : (all_or & ~((2u << hb) - 1u));    // the shared prefix (hb = -1: all keys equal)
        int slot = 1;
        for (int bit = hb; bit >= 0; --bit) {
            const uint32_t test = kth | ((1u << bit) - 1u);
            int c = 0;
#pragma unroll
            for (int v = 0; v < VP; ++v) c += key[v] <= test ? 1 : 0;   // (non-candidates never count: test < 0xffffffff here)
            if (set_block_sum<T>(c, red, slot) < k) kth |= 1u << bit;
            slot ^= 1;
        }
    }
    qq = 0.f;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) qq += qq_sh[w];
    const float ck = desc_key_to_float(kth);
    const float bw = 2.f * band_c * 0.00390625f * sqrtf(qq) * max_norm[0] * rsqrtf((float)d);
    const float hi = ck + bw, lo = ck - bw;
    // ---- classify: IN -> output now; band -> LDS ------------------------------------------------------------------------------------
#pragma unroll
    for (int v = 0; v < VP; ++v) {
        const float c = desc_key_to_float(key[v]);
        const bool have = row[v] >= 0;
        const bool is_in = have && c > hi, is_band = have && !is_in && !(c < lo);
        const unsigned long long im = __ballot(is_in), bm = __ballot(is_band);
        int bi = 0, bb = 0;
        if (lane == 0) {
            if (im) bi = atomicAdd(&cnt_in, __popcll(im));
            if (bm) bb = atomicAdd(&cnt_band, __popcll(bm));
        }
        bi = __shfl(bi, 0);
        bb = __shfl(bb, 0);
        if (is_in) {
            const int pos = bi + __popcll(im & ((1ull << lane) - 1ull));
            os[pos] = c;
            ol[pos] = label_map ? label_map[row[v]] : row[v];
        }
        if (is_band) band[bb + __popcll(bm & ((1ull << lane) - 1ull))] = (uint32_t)row[v];
    }
    __syncthreads();
    const int a = cnt_in, nb = cnt_band;   // a <= k - 1, a + nb >= k
    // ---- exact scores of the band (the arithmetic of rescore_kernel: same exact scores bit for bit) ---------------------------------
    constexpr int U = 4;
    for (int e0 = wave * U; e0 < nb; e0 += (T / 64) * U) {
        int32_t r[U];
        float s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = (e0 + u < nb) ? (int32_t)(uint32_t)band[e0 + u] : -1;
        float acc[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u][0] = acc[u][1] = acc[u][2] = acc[u][3] = 0.f;
        for (int c = lane * 4; c < dpad; c += 256) {
            const f32x4 qv = *(const f32x4*)(qrow + c);
            f32x4 xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r[u] >= 0) xv[u] = *(const f32x4*)(x32 + (int64_t)r[u] * ldx + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc[u][0] = fmaf(xv[u][0], qv[0], acc[u][0]);
                acc[u][1] = fmaf(xv[u][1], qv[1], acc[u][1]);
                acc[u][2] = fmaf(xv[u][2], qv[2], acc[u][2]);
                acc[u][3] = fmaf(xv[u][3], qv[3], acc[u][3]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s[u] = (acc[u][0] + acc[u][1]) + (acc[u][2] + acc[u][3]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s[u] += __shfl_xor(s[u], o);
        }
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (e0 + u < nb) band[e0 + u] = ((uint64_t)desc_key(s[u]) << 32) | (uint32_t)(label_map ? label_map[r[u]] : r[u]);
        }
    }
    __syncthreads();
    // ---- the k - a best of the band by (exact score desc, label asc): rank counting (keys are distinct: the label is part of the key) ----
    const int need = k - a;
    for (int i = threadIdx.x; i < nb; i += T) {
        const uint64_t me = band[i];
        int rank = 0;
        for (int j = 0; j < nb; ++j) rank += band[j] < me ? 1 : 0;
        if (rank < need) {
            os[a + rank] = desc_key_to_float((uint32_t)(me >> 32));
            ol[a + rank] = (int64_t)(uint32_t)me;
        }
    }
    if (threadIdx.x == 0 && stats) {   // (kSetStatSlots counter pairs: 290 000 atomics on ONE pair of addresses took longer than the kernel's work)
        atomicAdd(stats + 2 * (blockIdx.x & (kSetStatSlots - 1)), (unsigned long long)nb);
        atomicAdd(stats + 2 * (blockIdx.x & (kSetStatSlots - 1)) + 1, (unsigned long long)m);
    }
}

// ---- approximate (IVF) search: exact fp32 scores of a query against the rows of its probed lists ---------------------------------
// The reference's approximate alternative is faiss.IndexHNSWFlat (dvl/indexer/faiss_indexers.py:90-154); a graph walk is a poor fit
// for a wide machine, an inverted-file scan is not: the rows are stored sorted by list, a query reads nprobe contiguous row ranges of
// the fp32 master copy (HBM-bound, ~3 KB per row) and scores them exactly.  Workgroup = 4 waves, 64 rows of one (query, probe) pair;
// a wave scores 16 rows, four at a time (the arithmetic of one row is the re-score kernel's: 4 fmaf chains over the columns
// lane*4 + 256*i, pairwise sum, xor-shuffle tree).  S[q][probe * lpad + offset] receives the score, rows past the list's end the
// padding score (their columns are translated to label -1 afterwards).
__global__ __launch_bounds__(256) void scan_lists_kernel(const float* __restrict__ q32, int64_t ldq, const float* __restrict__ x32,
                                                         int64_t ldx, int dpad, const int64_t* __restrict__ list_offsets,
                                                         const int32_t* __restrict__ probes, int nprobe, int nlist, int lpad,
                                                         float* __restrict__ S, int64_t lds_elems) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = blockIdx.y;
    const int probe = blockIdx.x / (lpad / 64), blk = blockIdx.x % (lpad / 64);
    const int32_t list = probes[q * nprobe + probe];
    const bool ok = list >= 0 && list < nlist;      // (-1 = skip; an out-of-range list id is treated the same way)
    const int64_t r_beg = ok ? list_offsets[list] : 0, r_end = ok ? list_offsets[list + 1] : 0;
    const int o0 = blk * 64 + wave * 16;
    float* out = S + q * lds_elems + (int64_t)probe * lpad + o0;
    const float* qrow = q32 + q * ldq;
    constexpr int U = 4;
    for (int u0 = 0; u0 < 16; u0 += U) {
        float acc[U][4];
        int64_t r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u] = r_beg + o0 + u0 + u;
            acc[u][0] = acc[u][1] = acc[u][2] = acc[u][3] = 0.f;
        }
        if (r[0] < r_end) {   // (wave-uniform)
            for (int c = lane * 4; c < dpad; c += 256) {
                const f32x4 qv = *(const f32x4*)(qrow + c);
                f32x4 xv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    xv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (r[u] < r_end) xv[u] = *(const f32x4*)(x32 + r[u] * ldx + c);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    acc[u][0] = fmaf(xv[u][0], qv[0], acc[u][0]);
                    acc[u][1] = fmaf(xv[u][1], qv[1], acc[u][1]);
                    acc[u][2] = fmaf(xv[u][2], qv[2], acc[u][2]);
                    acc[u][3] = fmaf(xv[u][3], qv[3], acc[u][3]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float sc = (acc[u][0] + acc[u][1]) + (acc[u][2] + acc[u][3]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sc += __shfl_xor(sc, o);
            if (lane == 0) out[u0 + u] = r[u] < r_end ? sc : LDOT_PAD_SCORE;
        }
    }
}

// labels of the list search are COLUMNS of S (probe * lpad + offset): -> index rows, or -1 past a list's end / for padding
__global__ __launch_bounds__(256) void translate_cols_kernel(int64_t* __restrict__ labels, int64_t n, int k,
                                                             const int64_t* __restrict__ list_offsets,
                                                             const int32_t* __restrict__ probes, int nprobe, int nlist, int lpad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t col = labels[i];
    int64_t row = -1;
    if (col >= 0) {
        const int probe = (int)(col / lpad), off = (int)(col % lpad);
        const int32_t list = probes[(i / k) * nprobe + probe];
        if (list >= 0 && list < nlist && list_offsets[list] + off < list_offsets[list + 1]) row = list_offsets[list] + off;
    }
    labels[i] = row;
}

int launch_scan_lists(const float* q32, int64_t ldq, const float* x32, int64_t ldx, int dpad, int64_t nq,
                      const int64_t* list_offsets, const int32_t* probes, int nprobe, int nlist, int lpad, float* S,
                      int64_t lds_elems, hipStream_t st) {
    if (nq <= 0 || nprobe <= 0) return LDOT_OK;
    hipLaunchKernelGGL(scan_lists_kernel, dim3((unsigned)(nprobe * (lpad / 64)), (unsigned)nq), dim3(256), 0, st, q32, ldq, x32, ldx,
                       dpad, list_offsets, probes, nprobe, nlist, lpad, S, lds_elems);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_translate_cols(int64_t* labels, int64_t nq, int k, const int64_t* list_offsets, const int32_t* probes, int nprobe,
                          int nlist, int lpad, hipStream_t st) {
    const int64_t n = nq * k;
    if (n <= 0) return LDOT_OK;
    hipLaunchKernelGGL(translate_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, labels, n, k, list_offsets, probes,
                       nprobe, nlist, lpad);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// LDOT_OPT_VERIFY: is the reported top-k PROVEN to be the exact fp32 top-k?  A row that is not among the k' candidates has a bf16
// candidate score <= tau (the k'-th best candidate score), hence an exact score <= tau + E where E bounds the bf16 input-rounding
// error of one score.  If the k-th reported exact score is >= tau + E no such row can displace it.  E is the statistical bound
// c * 2^-8 * |q| * max|x| / sqrt(d) with c = 4 (input roundings treated as independent, relative error <= 2^-9 each, 8 standard
// deviations for vectors whose mass is spread over the d coordinates); a query that fails it is FLAGGED, not wrong — the caller
// re-searches the flagged queries with a larger margin (FlatIPIndex.search(..., verify=True)).
__global__ __launch_bounds__(256) void verify_exact_kernel(const float* __restrict__ q32, int64_t ldq, int d, int64_t nq,
                                                           const float* __restrict__ out_s, const int64_t* __restrict__ out_l,
                                                           int k, const float* __restrict__ tau,
                                                           const float* __restrict__ max_norm, float c,
                                                           int32_t* __restrict__ flags, int32_t* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float* r = q32 + q * ldq;
    float acc = 0.f;
    for (int i = lane; i < d; i += 64) acc = fmaf(r[i], r[i], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        const float t = tau[q];
        const bool full = out_l[q * k + k - 1] >= 0;     // fewer than k rows: everything there is was reported
        const float E = c * 0.00390625f * sqrtf(acc) * max_norm[0] * rsqrtf((float)d);
        const bool proven = !full || t == -INFINITY || out_s[q * k + k - 1] >= t + E;
        flags[q] = proven ? 0 : 1;
        if (!proven) atomicAdd(count, 1);
    }
}

int launch_verify_exact(const float* q32, int64_t ldq, int d, int64_t nq, const float* out_s, const int64_t* out_l, int k,
                        const float* tau, const float* max_norm, int32_t* flags, int32_t* count, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    hipLaunchKernelGGL(verify_exact_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, q32, ldq, d, nq, out_s, out_l, k,
                       tau, max_norm, kVerifyC, flags, count);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_rescore_set(const float* q32, int64_t ldq, const float* x32, int64_t ldx, int dpad, int d, int64_t nq, const float* list_s,
                       const int32_t* list_i, int kp, int k, const float* max_norm, float band_c, float* out_s, int64_t* out_l,
                       const int32_t* label_map, unsigned long long* stats, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    int P = 2;
    while (P < kp) P <<= 1;
    const size_t lds = (size_t)P * 8;
#define LDOT_SET_LAUNCH(T, VP, GRID)                                                                                                  \
    hipLaunchKernelGGL((rescore_set_kernel<T, VP>), dim3((unsigned)(GRID)), dim3(T), lds, st, q32, ldq, x32, ldx, dpad, d, list_s, list_i, \
                       kp, k, max_norm, band_c, out_s, out_l, label_map, stats, nq)
    if (kp <= 128)
        LDOT_SET_LAUNCH(64, 2, nq);
    else if (kp <= 256)
        LDOT_SET_LAUNCH(64, 4, nq);
    else if (kp <= 1024)
        LDOT_SET_LAUNCH(256, 4, round_up(nq, 128));
    else
        LDOT_SET_LAUNCH(256, 12, round_up(nq, 128));
#undef LDOT_SET_LAUNCH
    static_assert(kMaxKp <= 256 * 12, "rescore_set_kernel: candidates per thread");
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_rescore(const float* q32, int64_t ldq, const float* x32, int64_t ldx, int dpad, int64_t nq,
                   const float* list_s, const int32_t* list_i, int kp, int k, int do_rescore, const float* floor,
                   float* out_s, int64_t* out_l, hipStream_t st, const RescoreOut* layout, const int32_t* label_map) {
    if (nq <= 0) return LDOT_OK;
    RescoreOut lay = layout ? *layout : RescoreOut{0, 0, 0, 0, nullptr};
    if (label_map) lay.label_map = label_map;
    int P = 2;
    while (P < kp) P <<= 1;
    const size_t lds = (size_t)P * 8;
    if (nq <= 128)
        hipLaunchKernelGGL(rescore_kernel<1024>, dim3((unsigned)nq), dim3(1024), lds, st, q32, ldq, x32, ldx, dpad, list_s,
                           list_i, kp, k, do_rescore, floor, out_s, out_l, lay, nq);
    else   // (whole groups of 128 blocks: the kernel deals sixteen consecutive queries to one XCD)
        hipLaunchKernelGGL(rescore_kernel<256>, dim3((unsigned)round_up(nq, 128)), dim3(256), lds, st, q32, ldq, x32, ldx, dpad, list_s,
                           list_i, kp, k, do_rescore, floor, out_s, out_l, lay, nq);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

}  // namespace ldot
