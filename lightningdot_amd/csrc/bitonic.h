// In-LDS bitonic sort of P (power of two) 64-bit keys, ascending, by one workgroup.
//
// Multi-wave workgroups: wave w owns the contiguous chunk of P / nwaves elements [w * P/nw, (w+1) * P/nw).  Every
// compare-exchange stage whose partner distance j fits inside a chunk touches only the wave's own elements, and the
// LDS executes one wave's operations in issue order — such stages need NO workgroup barrier.  Only the log2(nw) * ...
// stages with j >= chunk/2 (3 of 55 for P = 1024 and 4 waves) are bracketed by barriers.  A single-wave workgroup
// never needs a hardware barrier.
#pragma once
#include "ldot_common.h"

namespace ldot {

__device__ __forceinline__ void bitonic_cmpx(uint64_t* keys, int t, int j, int k) {
    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    const int p = i | j;
    const bool asc = ((i & k) == 0);
    const uint64_t a = keys[i], b = keys[p];
    if ((a > b) == asc) {
        keys[i] = b;
        keys[p] = a;
    }
}

// Callers make the keys visible to the whole workgroup before the call (barrier); on return all keys are visible to all.
__device__ inline void bitonic_sort_lds(uint64_t* keys, int P) {
    const int nthreads = blockDim.x;
    if (nthreads <= 64) {
        for (int k = 2; k <= P; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < (P >> 1); t += nthreads) bitonic_cmpx(keys, t, j, k);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        return;
    }
    const int nw = nthreads >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ppw = (P >> 1) / nw;   // compare-exchange pairs per wave (0 for tiny P: every stage is cross-wave)
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j <= ppw) {   // partner inside the wave's chunk: wave-local stage
                for (int t = wave * ppw + lane; t < (wave + 1) * ppw; t += 64) bitonic_cmpx(keys, t, j, k);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                __syncthreads();
                for (int t = threadIdx.x; t < (P >> 1); t += nthreads) bitonic_cmpx(keys, t, j, k);
                __syncthreads();
            }
        }
    }
    __syncthreads();
}

}  // namespace ldot
