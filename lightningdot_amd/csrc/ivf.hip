// Inverted-file scan: exact fp32 scores of a query against the rows of its probed lists, top-k of those — the query path of the
// approximate index that stands in for the reference's faiss.IndexHNSWFlat (dvl/indexer/faiss_indexers.py:90-154, search :145-154).
//
// A query's probed rows form ONE compact column space: column c = (rows of probe 0) ++ (rows of probe 1) ++ ...  (prefix sums of the
// list lengths, computed per query on the device), so a query with 32 lists of ~250 rows is 8000 columns however long the longest list
// of the index is.  (The first version padded every probed list to the longest one: 32 x 6464 columns at 1M rows / 4000 lists,
// 0.24 ms per query — no faster than the exact search.)
//
//   ivf_prefix_kernel   per query: validated list ids, first rows of the lists, exclusive prefix sums of their lengths
//   ivf_scan_kernel     workgroups stride over blocks of 16 columns; a wave scores its columns four rows at a time (the arithmetic
//                       of the re-score kernel: 4 fmaf chains over the columns lane*4 + 256*i, pairwise sum, xor-shuffle tree) and
//                       the block raises the maximum of its run
//   (threshold + collect: the run-maxima selection of select_narrow.hip)
//   ivf_final_kernel    sorts a query's candidates and writes the top k: exact score, column translated back to the index row
#include <math.h>

#include <algorithm>

#include "bitonic.h"
#include "kernels.h"

namespace ldot {

// largest j with cs[j] <= col (cs ascending, cs[0] = 0, n + 1 entries; empty lists are skipped by construction)
__device__ __forceinline__ int ivf_find_probe(const int32_t* cs, int n, int col) {
    int lo = 0, hi = n;   // invariant: cs[lo] <= col < cs[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cs[mid] <= col)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

template <typename ProbeT>
__global__ __launch_bounds__(256) void ivf_prefix_kernel(const ProbeT* __restrict__ probes, int nprobe, int nlist,
                                                         const int64_t* __restrict__ list_offsets, int32_t* __restrict__ plist,
                                                         int64_t* __restrict__ rowbase, int32_t* __restrict__ cstart) {
    __shared__ int wsum[4];
    __shared__ int carry_sh;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_sh = 0;
    __syncthreads();
    for (int j0 = 0; j0 < nprobe; j0 += 256) {
        const int j = j0 + tid;
        int len = 0;
        int32_t l = -1;
        if (j < nprobe) {
            const int64_t p = (int64_t)probes[(int64_t)q * nprobe + j];
            int64_t first = 0;
            if (p >= 0 && p < nlist) {   // (-1 = skip; an out-of-range list id is treated the same way)
                l = (int32_t)p;
                first = list_offsets[p];
                len = (int)(list_offsets[p + 1] - first);
            }
            plist[(int64_t)q * nprobe + j] = l;
            rowbase[(int64_t)q * nprobe + j] = first;
        }
        int incl = len;   // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = carry_sh;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        if (j < nprobe) cstart[(int64_t)q * (nprobe + 1) + j] = base + incl - len;
        __syncthreads();
        if (tid == 255) carry_sh = base + incl;
        __syncthreads();
    }
    if (tid == 0) cstart[(int64_t)q * (nprobe + 1) + nprobe] = carry_sh;
}

// a workgroup scores blocks of 16 columns (4 per wave); 1 << run_shift blocks form a run, whose maximum is raised with atomicMax
// (M is all zero on entry)
__global__ __launch_bounds__(256) void ivf_scan_kernel(const float* __restrict__ q32, int64_t ldq, const float* __restrict__ x32,
                                                       int64_t ldx, int dpad, const int64_t* __restrict__ rowbase,
                                                       const int32_t* __restrict__ cstart, int nprobe, float* __restrict__ S,
                                                       int64_t lds_elems, uint32_t* __restrict__ M, int64_t ldm, int run_shift) {
    constexpr int RUN = 16;
    // the query's first rows [nprobe] and prefix sums [nprobe + 1] in LDS: column -> row needs no further global round trip
    extern __shared__ __attribute__((aligned(8))) int64_t rb[];
    int32_t* cs = (int32_t*)(rb + nprobe);
    __shared__ float wmax[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = blockIdx.y;
    const int32_t* cq = cstart + q * (nprobe + 1);
    for (int i = threadIdx.x; i <= nprobe; i += 256) cs[i] = cq[i];
    for (int i = threadIdx.x; i < nprobe; i += 256) rb[i] = rowbase[q * nprobe + i];
    __syncthreads();
    const int total = cs[nprobe];
    const float* qrow = q32 + q * ldq;
    constexpr int CPW = RUN / 4, U = 4;
    for (int blk = blockIdx.x; blk * RUN < total; blk += gridDim.x) {
        const int c0 = blk * RUN + wave * CPW;
        float m = -INFINITY;
        // probe of the wave's first column; later columns advance linearly
        int j = c0 < total ? ivf_find_probe(cs, nprobe, c0) : nprobe - 1;
        for (int u0 = 0; u0 < CPW; u0 += U) {
            int64_t r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u0 + u;
                r[u] = -1;
                if (c < total) {
                    while (c >= cs[j + 1]) ++j;
                    r[u] = rb[j] + (c - cs[j]);
                }
            }
            float acc[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u][0] = acc[u][1] = acc[u][2] = acc[u][3] = 0.f;
            if (r[0] >= 0) {   // (wave-uniform)
                for (int c = lane * 4; c < dpad; c += 256) {
                    const f32x4 qv = *(const f32x4*)(qrow + c);
                    f32x4 xv[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        xv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (r[u] >= 0) xv[u] = *(const f32x4*)(x32 + r[u] * ldx + c);
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
                if (r[u] >= 0) {
                    if (lane == 0) S[q * lds_elems + c0 + u0 + u] = sc;
                    m = fmaxf(m, sc);
                }
            }
        }
        if (lane == 0) wmax[wave] = m;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicMax(M + q * ldm + (blk >> run_shift), ~desc_key(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
        __syncthreads();
    }
}

// ---- bf16 list scan (round 3): the probed lists are scored from the bf16 SHADOW (half the bytes of the fp32 scan above), candidates are
// re-scored exactly afterwards (narrow_finish_kernel).  A wave takes 16 consecutive COLUMNS of the query's compact space (a run of the
// run-maxima selection) = 16 index rows that are consecutive within a list; lane l gathers, for every 32-k slab, the 16 bytes of row
// (l & 15), k chunk (l >> 4) from the blocked shadow (block (row >> 4, slab) + (row & 15) * 64 + (l >> 4) * 16: the four lanes of a row
// read one 64-byte segment, consecutive rows adjacent segments) — which IS the A operand of v_mfma_f32_16x16x32_bf16 in register order.
// The B operand is the query's 16-query block of the blocked query shadow, staged in LDS once per workgroup; column (q & 15) of the
// result holds the query's 16 scores (lanes with (l & 15) == (q & 15): rows (l >> 4) * 4 .. + 4).
template <int UNR>
__global__ __launch_bounds__(256) void ivf_scan_bf16_kernel(const char* __restrict__ Q16b, const char* __restrict__ X16b, int nslab,
                                                            const int64_t* __restrict__ rowbase, const int32_t* __restrict__ cstart,
                                                            int nprobe, float* __restrict__ S, int64_t lds_elems,
                                                            uint32_t* __restrict__ M, int64_t ldm, int run_shift) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = blockIdx.y;
    int64_t* rb = (int64_t*)(smem + (size_t)nslab * 1024);
    int32_t* cs = (int32_t*)(rb + nprobe);
    const char* qsrc = Q16b + (q >> 4) * (int64_t)nslab * 1024;
    for (int i = threadIdx.x; i < nslab * 64; i += 256) ((uint4*)smem)[i] = ((const uint4*)qsrc)[i];
    const int32_t* cq = cstart + q * (nprobe + 1);
    for (int i = threadIdx.x; i <= nprobe; i += 256) cs[i] = cq[i];
    for (int i = threadIdx.x; i < nprobe; i += 256) rb[i] = rowbase[q * nprobe + i];
    __syncthreads();
    const int total = cs[nprobe];
    const int lo = (lane & 15) * 64 + (lane >> 4) * 16;     // lane's byte offset inside a 1-KiB block (A and B operand alike)
    const int qc = (int)(q & 15);
    for (int blk = blockIdx.x * 4 + wave; blk * 16 < total; blk += gridDim.x * 4) {
        const int c = blk * 16 + (lane & 15);               // the column whose row this lane gathers
        int64_t r = -1;
        if (c < total) {
            const int j = ivf_find_probe(cs, nprobe, c);
            r = rb[j] + (c - cs[j]);
        }
        // (columns past the query's last one read row 0: their scores are never stored)
        const char* xrow = X16b + ((r < 0 ? 0 : r) >> 4) * (int64_t)nslab * 1024 + ((r < 0 ? 0 : r) & 15) * 64 + (lane >> 4) * 16;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        for (int s0 = 0; s0 < nslab; s0 += UNR) {
            bf16x8_t a[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) a[u] = __builtin_nontemporal_load((const bf16x8_t*)(xrow + (int64_t)(s0 + u) * 1024));
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                acc[u & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], *(const bf16x8_t*)(smem + (s0 + u) * 1024 + lo), acc[u & 1], 0, 0, 0);
        }
        const f32x4 v = acc[0] + acc[1];                    // rows (lane >> 4) * 4 .. + 4 of column lane & 15
        const int cbase = blk * 16 + (lane >> 4) * 4;
        float m = -INFINITY;
        if ((lane & 15) == qc) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (cbase + i < total) {
                    S[q * lds_elems + cbase + i] = v[i];
                    m = fmaxf(m, v[i]);
                }
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (lane == qc && m > -INFINITY) atomicMax(M + q * ldm + (blk >> run_shift), ~desc_key(m));
    }
}

// one workgroup per query: candidate keys {descending score key, column} -> top k: exact scores + index rows (-1 / pad score beyond
// the candidates); over[q] = 1 when the candidate buffer was full
__global__ __launch_bounds__(256) void ivf_final_kernel(const uint64_t* __restrict__ cand, int cap, int32_t* __restrict__ cnt,
                                                        const int64_t* __restrict__ rowbase,
                                                        const int32_t* __restrict__ cstart, int nprobe, int k,
                                                        float* __restrict__ out_s, int64_t* __restrict__ out_l,
                                                        int32_t* __restrict__ over) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    const int64_t q = blockIdx.x;
    int n = cnt[q * kNarrowCntStride];
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt[q * kNarrowCntStride] = 0;   // ready for the next search
        over[q] = n > cap ? 1 : 0;
    }
    if (n > cap) n = cap;
    int P = 2;
    while (P < n) P <<= 1;
    for (int i = threadIdx.x; i < P; i += 256) keys[i] = i < n ? cand[q * cap + i] : ~0ull;
    __syncthreads();
    bitonic_sort_lds(keys, P);
    const int32_t* cq = cstart + q * (nprobe + 1);
    for (int i = threadIdx.x; i < k; i += 256) {
        float s = LDOT_PAD_SCORE;
        int64_t row = -1;
        if (i < n) {
            const uint64_t kv = keys[i];
            const int col = (int)(uint32_t)kv;
            const int j = ivf_find_probe(cq, nprobe, col);
            s = desc_key_to_float((uint32_t)(kv >> 32));
            row = rowbase[q * nprobe + j] + (col - cq[j]);
        }
        out_s[q * k + i] = s;
        out_l[q * k + i] = row;
    }
}

int launch_ivf_prefix(const void* probes, int probes_are_int64, int64_t nq, int nprobe, int nlist, const int64_t* list_offsets,
                      int32_t* plist, int64_t* rowbase, int32_t* cstart, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    if (probes_are_int64)
        hipLaunchKernelGGL(ivf_prefix_kernel<int64_t>, dim3((unsigned)nq), dim3(256), 0, st, (const int64_t*)probes, nprobe, nlist,
                           list_offsets, plist, rowbase, cstart);
    else
        hipLaunchKernelGGL(ivf_prefix_kernel<int32_t>, dim3((unsigned)nq), dim3(256), 0, st, (const int32_t*)probes, nprobe, nlist,
                           list_offsets, plist, rowbase, cstart);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// max_cols: upper bound of a query's column count (nprobe * longest list); a run maximum covers 16 << run_shift columns
int launch_ivf_scan(const float* q32, int64_t ldq, const float* x32, int64_t ldx, int dpad, int64_t nq, const int64_t* rowbase,
                    const int32_t* cstart, int nprobe, int64_t max_cols, int run_shift, float* S, int64_t lds_elems, uint32_t* M,
                    int64_t ldm, hipStream_t st) {
    if (nq <= 0 || nprobe <= 0 || max_cols <= 0) return LDOT_OK;
    LDOT_REQUIRE(run_shift >= 0 && run_shift <= 16, LDOT_EINVAL, "bad run length");
    LDOT_REQUIRE((size_t)(nprobe + 1) * 12 <= 64 * 1024, LDOT_EINVAL, "too many probes");
    const int64_t blocks = (max_cols + 15) / 16;
    // workgroups stride over a query's blocks: enough of them to fill the machine with one query, fewer per query in a batch
    const int64_t per_q = std::max<int64_t>(8, 2048 / nq);
    const unsigned gx = (unsigned)std::min<int64_t>(blocks, per_q);
    hipLaunchKernelGGL(ivf_scan_kernel, dim3(gx, (unsigned)nq), dim3(256), (size_t)(nprobe + 1) * 12, st, q32, ldq, x32, ldx, dpad,
                       rowbase, cstart, nprobe, S, lds_elems, M, ldm, run_shift);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

static bool g_ivf_bf16_attr[3][64];

// the same scan from the bf16 shadow (q16b / x16b: the BLOCKED shadows, ld_elems = dpad elements per row); S receives bf16-input
// scores — the caller re-scores its candidates exactly
int launch_ivf_scan_bf16(const void* q16b, const void* x16b, int64_t ld_elems, int64_t nq, const int64_t* rowbase, const int32_t* cstart,
                         int nprobe, int64_t max_cols, int run_shift, float* S, int64_t lds_elems, uint32_t* M, int64_t ldm,
                         hipStream_t st) {
    if (nq <= 0 || nprobe <= 0 || max_cols <= 0) return LDOT_OK;
    const int nslab = (int)(ld_elems / 32);
    const size_t lds = (size_t)nslab * 1024 + (size_t)(nprobe + 1) * 12 + 8;
    LDOT_REQUIRE(run_shift >= 0 && run_shift <= 16 && ld_elems % 64 == 0 && lds <= 64 * 1024, LDOT_EINVAL, "bad bf16 list scan shape");
    const int64_t blocks = (max_cols + 15) / 16;
    const int64_t per_q = std::max<int64_t>(8, 2048 / nq);
    const unsigned gx = (unsigned)std::min<int64_t>((blocks + 3) / 4, per_q);
    int dev = 0;
    LDOT_HIP_CHECK(hipGetDevice(&dev));
#define LDOT_IVF16(U, SLOT)                                                                                                     \
    do {                                                                                                                        \
        if (dev >= 64 || !g_ivf_bf16_attr[SLOT][dev]) {                                                                          \
            LDOT_HIP_CHECK(hipFuncSetAttribute((const void*)ivf_scan_bf16_kernel<U>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               64 * 1024));                                                                      \
            if (dev < 64) g_ivf_bf16_attr[SLOT][dev] = true;                                                                     \
        }                                                                                                                        \
        hipLaunchKernelGGL((ivf_scan_bf16_kernel<U>), dim3(gx, (unsigned)nq), dim3(256), lds, st, (const char*)q16b,              \
                           (const char*)x16b, nslab, rowbase, cstart, nprobe, S, lds_elems, M, ldm, run_shift);                   \
    } while (0)
    if (nslab % 8 == 0)
        LDOT_IVF16(8, 0);
    else if (nslab % 4 == 0)
        LDOT_IVF16(4, 1);
    else
        LDOT_IVF16(2, 2);
#undef LDOT_IVF16
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

static bool g_ivf_final_attr[64];

int launch_ivf_final(const uint64_t* cand, int cap, int32_t* cnt, int64_t nq, const int64_t* rowbase, const int32_t* cstart,
                     int nprobe, int k, float* out_s, int64_t* out_l, int32_t* over, hipStream_t st) {
    if (nq <= 0) return LDOT_OK;
    LDOT_REQUIRE(cap >= 2 && (cap & (cap - 1)) == 0 && cap <= kNarrowCandCap, LDOT_EINVAL, "bad candidate capacity");
    int dev = 0;
    LDOT_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 64 || !g_ivf_final_attr[dev]) {
        LDOT_HIP_CHECK(hipFuncSetAttribute((const void*)ivf_final_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kNarrowCandCap * 8));
        if (dev < 64) g_ivf_final_attr[dev] = true;
    }
    hipLaunchKernelGGL(ivf_final_kernel, dim3((unsigned)nq), dim3(256), (size_t)cap * 8, st, cand, cap, cnt, rowbase, cstart, nprobe, k,
                       out_s, out_l, over);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

}  // namespace ldot
