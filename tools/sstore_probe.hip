// Probe (measurement tool, not product): do scalar stores (s_store_dwordx4 / s_store_dword through the scalar data cache) work on gfx950,
// and what do they cost next to vector stores?  Each wave writes `per_wave` records of 9 dwords (two x4 + one dword) to its own slots.
// usage: tools/bin/sstore_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>   // 0 vector stores from lane 0, 1 scalar stores
__global__ __launch_bounds__(256) void probe(unsigned int* out, int per_wave) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const int lane = threadIdx.x & 63;
    for (int r = 0; r < per_wave; ++r) {
        const unsigned int rec = (unsigned int)(wave * per_wave + r);
        unsigned int* dst = out + (size_t)rec * 12;   // 48-byte slots, 36 bytes written
        const u32x4 a = {rec * 16 + 0, rec * 16 + 1, rec * 16 + 2, rec * 16 + 3};
        const u32x4 b = {rec * 16 + 4, rec * 16 + 5, rec * 16 + 6, rec * 16 + 7};
        const unsigned int c = rec * 16 + 8;
        if (MODE == 0) {
            if (lane == 0) {
                *(u32x4*)dst = a;
                *(u32x4*)(dst + 4) = b;
                dst[8] = c;
            }
        } else {
            asm volatile(
                "s_store_dwordx4 %1, %0, 0x0\n\t"
                "s_store_dwordx4 %2, %0, 0x10\n\t"
                "s_store_dword %3, %0, 0x20\n\t"
                "s_waitcnt lgkmcnt(0)"
                :: "s"(dst), "s"(a), "s"(b), "s"(c) : "memory");
        }
    }
    if (MODE == 1) asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

int main() {
    const int blocks = 1024, waves = blocks * 4, per_wave = 64;
    const size_t n = (size_t)waves * per_wave * 12;
    unsigned int* d;
    CHECK(hipMalloc(&d, n * 4));
    std::vector<unsigned int> h(n);
    for (int mode = 0; mode < 2; ++mode) {
        CHECK(hipMemset(d, 0xff, n * 4));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) probe<0><<<blocks, 256>>>(d, per_wave); else probe<1><<<blocks, 256>>>(d, per_wave);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipGetLastError());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        CHECK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t rec = 0; rec < (size_t)waves * per_wave; ++rec)
            for (int j = 0; j < 9; ++j) bad += h[rec * 12 + j] != (unsigned int)(rec * 16 + j);
        printf("%s stores: %d records of 36 B in %.3f ms (%.1f ns per record per wave-serial chain), wrong dwords: %zu\n",
               mode ? "scalar" : "vector", waves * per_wave, best, best * 1e6 / per_wave, bad);
    }
    return 0;
}
