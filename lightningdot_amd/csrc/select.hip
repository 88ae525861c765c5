// Streaming top-k' selection in LDS (one 256-thread workgroup per query).
//
// Replaces the per-query max-heap of faiss' flat search (dvl/indexer/faiss_indexers.py:83 -> IndexFlatIP.search)
// with a CDNA-friendly scheme: candidates are packed into unique 64-bit keys
//     key = (descending-order image of the fp32 score) << 32 | row
// so that "smaller key" == "better" == (higher score, then lower row).  A workgroup streams candidate segments,
// appends those that beat the current k'-th key to an LDS buffer and, only when the buffer could overflow,
// compacts it with a bitonic sort (LDS, 64-bit compare-exchange) and tightens the threshold.
#include <float.h>

#include "kernels.h"

namespace ldot {

constexpr uint64_t kEmptyKey = ~0ull;

__device__ inline uint64_t make_key(float s, uint32_t row) { return ((uint64_t)desc_key(s) << 32) | row; }

__device__ inline void bitonic_sort_lds(uint64_t* keys, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += kSelThreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const bool asc = ((i & k) == 0);
                const uint64_t a = keys[i], b = keys[p];
                if ((a > b) == asc) {
                    keys[i] = b;
                    keys[p] = a;
                }
            }
            __syncthreads();
        }
    }
}

struct Selector {
    uint64_t* keys;   // LDS [kSelCap]
    int* count;       // LDS
    uint64_t tau;     // uniform
    int kp;

    __device__ inline void init(uint64_t* k, int* c, int kp_) {
        keys = k;
        count = c;
        kp = kp_;
        tau = kEmptyKey;
        if (threadIdx.x == 0) *count = 0;
        __syncthreads();
    }
    __device__ inline void push(uint64_t key) {
        if (key < tau) {
            const int pos = atomicAdd(count, 1);
            keys[pos] = key;
        }
    }
    // sort, truncate to kp, refresh tau.  Must be called by all threads.
    __device__ inline void compact() {
        __syncthreads();
        const int n = *count;
        int P = 2;
        while (P < n) P <<= 1;
        for (int i = n + threadIdx.x; i < P; i += kSelThreads) keys[i] = kEmptyKey;
        __syncthreads();
        bitonic_sort_lds(keys, P);
        const int m = n < kp ? n : kp;
        const uint64_t t = (m >= kp) ? keys[kp - 1] : kEmptyKey;
        __syncthreads();
        if (threadIdx.x == 0) *count = m;
        tau = t;
        __syncthreads();
    }
    // call before streaming up to kSelSeg more candidates
    __device__ inline void reserve_segment() {
        __syncthreads();
        const int n = *count;
        __syncthreads();                              // everyone has read count before anyone pushes again
        if (n + kSelSeg > kSelCap) compact();         // uniform branch
    }
    __device__ inline void load_list(const float* ls, const int32_t* li) {
        for (int e = threadIdx.x; e < kp; e += kSelThreads) {
            const int32_t r = li[e];
            if (r >= 0) push(make_key(ls[e], (uint32_t)r));
        }
    }
    __device__ inline void finish(float* ls, int32_t* li, float* tau_out) {
        compact();
        const int n = *count;
        for (int e = threadIdx.x; e < kp; e += kSelThreads) {
            if (e < n) {
                const uint64_t k = keys[e];
                ls[e] = desc_key_to_float((uint32_t)(k >> 32));
                li[e] = (int32_t)(uint32_t)(k & 0xffffffffu);
            } else {
                ls[e] = LDOT_PAD_SCORE;
                li[e] = -1;
            }
        }
        if (tau_out && threadIdx.x == 0)
            *tau_out = (n >= kp) ? desc_key_to_float((uint32_t)(keys[kp - 1] >> 32)) : -INFINITY;
    }
};

__global__ __launch_bounds__(kSelThreads) void init_lists_kernel(float* ls, int32_t* li, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        ls[i] = LDOT_PAD_SCORE;
        li[i] = -1;
    }
}

// dense source: one row of a materialised score chunk
__global__ __launch_bounds__(kSelThreads) void select_dense_kernel(const float* __restrict__ S, int64_t lds_elems,
                                                                   int64_t ncols, int64_t idx_base,
                                                                   float* __restrict__ list_s,
                                                                   int32_t* __restrict__ list_i, int kp,
                                                                   float* __restrict__ tau) {
    __shared__ __attribute__((aligned(16))) uint64_t keys[kSelCap];
    __shared__ int count;
    const int64_t q = blockIdx.x;
    Selector sel;
    sel.init(keys, &count, kp);
    float* ls = list_s + q * kp;
    int32_t* li = list_i + q * kp;
    sel.load_list(ls, li);
    sel.compact();
    const float* row = S + q * lds_elems;
    for (int64_t c0 = 0; c0 < ncols; c0 += kSelSeg) {
        sel.reserve_segment();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t c = c0 + h * (kSelSeg / 2) + threadIdx.x * 4;
            if (c + 3 < ncols) {
                const f32x4 v = *(const f32x4*)(row + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) sel.push(make_key(v[e], (uint32_t)(idx_base + c + e)));
            } else {
                for (int e = 0; e < 4; ++e)
                    if (c + e < ncols) sel.push(make_key(row[c + e], (uint32_t)(idx_base + c + e)));
            }
        }
    }
    sel.finish(ls, li, tau ? tau + q : nullptr);
}

// pool source: the per-query sub-pools filled by the fused filter kernel; resets the counters afterwards
__global__ __launch_bounds__(kSelThreads) void select_pools_kernel(const float* __restrict__ pool_s,
                                                                   const int32_t* __restrict__ pool_i,
                                                                   int32_t* __restrict__ pool_cnt,
                                                                   float* __restrict__ list_s,
                                                                   int32_t* __restrict__ list_i, int kp,
                                                                   float* __restrict__ tau,
                                                                   int32_t* __restrict__ overflow) {
    __shared__ __attribute__((aligned(16))) uint64_t keys[kSelCap];
    __shared__ int count;
    __shared__ int cnts[kPoolSubs];
    __shared__ int any_over;
    const int64_t q = blockIdx.x;
    Selector sel;
    sel.init(keys, &count, kp);
    if (threadIdx.x == 0) any_over = 0;
    float* ls = list_s + q * kp;
    int32_t* li = list_i + q * kp;
    for (int s = threadIdx.x; s < kPoolSubs; s += kSelThreads) {
        const int c = pool_cnt[q * kPoolSubs + s];
        cnts[s] = c;
        pool_cnt[q * kPoolSubs + s] = 0;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < kPoolSubs; s += kSelThreads)
        if (cnts[s] > kPoolCap) any_over = 1;
    sel.load_list(ls, li);
    sel.compact();
    const int64_t base = q * (int64_t)(kPoolSubs * kPoolCap);
    for (int s0 = 0; s0 < kPoolSubs * kPoolCap; s0 += kSelSeg) {
        sel.reserve_segment();
        for (int sl = s0 + threadIdx.x; sl < s0 + kSelSeg; sl += kSelThreads) {
            const int sub = sl / kPoolCap, e = sl % kPoolCap;
            if (e < cnts[sub]) sel.push(make_key(pool_s[base + sl], (uint32_t)pool_i[base + sl]));
        }
    }
    sel.finish(ls, li, tau ? tau + q : nullptr);
    if (threadIdx.x == 0 && any_over) overflow[q] = 1;
}

// explicit lists source (sharded merge): parts laid out [nparts][nq][k_in], int64 labels (< 2^32-1), -1 = empty
__global__ __launch_bounds__(kSelThreads) void select_lists_kernel(const float* __restrict__ cand_s,
                                                                   const int64_t* __restrict__ cand_l,
                                                                   int64_t part_stride, int nparts, int k_in,
                                                                   int k_out, float* __restrict__ out_s,
                                                                   int64_t* __restrict__ out_l) {
    __shared__ __attribute__((aligned(16))) uint64_t keys[kSelCap];
    __shared__ int count;
    const int64_t q = blockIdx.x;
    Selector sel;
    sel.init(keys, &count, k_out);
    const int total = nparts * k_in;
    for (int s0 = 0; s0 < total; s0 += kSelSeg) {
        sel.reserve_segment();
        for (int sl = s0 + threadIdx.x; sl < s0 + kSelSeg && sl < total; sl += kSelThreads) {
            const int p = sl / k_in, e = sl % k_in;
            const int64_t off = (int64_t)p * part_stride + q * k_in + e;
            const int64_t l = cand_l[off];
            if (l >= 0) sel.push(make_key(cand_s[off], (uint32_t)l));
        }
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

int launch_init_lists(float* list_s, int32_t* list_i, int64_t n, hipStream_t st) {
    if (n <= 0) return LDOT_OK;
    hipLaunchKernelGGL(init_lists_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, list_s, list_i, n);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_select_dense(const float* S, int64_t lds_elems, int64_t nq, int64_t ncols, int64_t idx_base,
                        float* list_s, int32_t* list_i, int kp, float* tau, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    hipLaunchKernelGGL(select_dense_kernel, dim3((unsigned)nq), dim3(kSelThreads), 0, st, S, lds_elems, ncols,
                       idx_base, list_s, list_i, kp, tau);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_select_pools(const float* pool_s, const int32_t* pool_i, const int32_t* pool_cnt, int64_t nq,
                        float* list_s, int32_t* list_i, int kp, float* tau, int32_t* overflow_flags,
                        hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    hipLaunchKernelGGL(select_pools_kernel, dim3((unsigned)nq), dim3(kSelThreads), 0, st, pool_s, pool_i,
                       (int32_t*)pool_cnt, list_s, list_i, kp, tau, overflow_flags);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_select_lists(const float* cand_s, const int64_t* cand_l, int64_t part_stride, int nparts, int k_in,
                        int64_t nq, int k_out, float* out_s, int64_t* out_l, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    hipLaunchKernelGGL(select_lists_kernel, dim3((unsigned)nq), dim3(kSelThreads), 0, st, cand_s, cand_l,
                       part_stride, nparts, k_in, k_out, out_s, out_l);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

}  // namespace ldot
