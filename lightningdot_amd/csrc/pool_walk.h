// Counter words and entry numbering of the fused filter's candidate sub-pools (kernels.h), shared by the pool selects (select.hip,
// select_big.hip).
#pragma once
#include "kernels.h"

namespace ldot {

// A sub-pool's counter word packs the counts of its four lane groups (kernels.h).  pool_counts: the word with every byte clamped to the
// group capacity, the total, and whether a group overflowed.  pool_entry_of: the entry of the sub-pool's e-th record (records numbered
// group by group): group g owns the entries [g * kPoolGroupCap, ...).
__device__ __forceinline__ uint32_t pool_counts(uint32_t word, int& total, bool& over) {
    uint32_t clamped = 0;
    total = 0;
#pragma unroll
    for (int g = 0; g < kPoolGroups; ++g) {
        const uint32_t b = (word >> (8 * g)) & 255u;
        over |= b > (uint32_t)kPoolGroupCap;
        const uint32_t cb = b < (uint32_t)kPoolGroupCap ? b : (uint32_t)kPoolGroupCap;
        clamped |= cb << (8 * g);
        total += (int)cb;
    }
    return clamped;
}
__device__ __forceinline__ int pool_entry_of(int e, uint32_t clamped) {
    const int p0 = (int)(clamped & 255u), p1 = p0 + (int)((clamped >> 8) & 255u), p2 = p1 + (int)((clamped >> 16) & 255u);
    const int g = (e >= p0) + (e >= p1) + (e >= p2);
    const int before = g == 0 ? 0 : g == 1 ? p0 : g == 2 ? p1 : p2;
    return g * kPoolGroupCap + (e - before);
}

}  // namespace ldot
