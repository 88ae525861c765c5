// Probe (measurement tool, not product): does the stride between the 1-KiB operand blocks of a K slab matter to the L2?
// The fused kernel reads a slab as 24 + 16 blocks of 1 KiB that lie `nslab` KiB apart (24 KiB at D = 768): if the L2 channel of an
// address were a plain function of its low bits, all blocks of a slab would sit on the same 4 of 16 channels.  Every workgroup (512 threads,
// one per CU, as in the kernel) streams the slabs of one "row tile" (24 groups) and one "query panel" (16 groups) out of an L2-resident
// window of its XCD, block (g, s) at (g * stride_kib + s) KiB, 5 wave-level 1-KiB loads per wave and slab.
// usage: tools/bin/l2_stride_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, long window, int stride_kib, int nk, int iters, unsigned int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const char* w = src + (long)xcd * window;
    // 4 row tiles and 8 panels per XCD, shared like in the kernel (slot = nsub * 8 + qsub)
    const long tile_bytes = 24l * stride_kib * 1024, panel_bytes = 16l * stride_kib * 1024;
    const char* rows = w + (slot >> 3) * tile_bytes;
    const char* panel = w + 4 * tile_bytes + (slot & 7) * panel_bytes;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        for (int s = 0; s < nk; ++s) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int g = j * 8 + wave;   // block index within the slab: 0..23 row groups, 24..39 panel groups
                const char* p = (g < 24 ? rows + (long)g * stride_kib * 1024 : panel + (long)(g - 24) * stride_kib * 1024) + s * 1024 + lane * 16;
                const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
                acc ^= v;
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main() {
    const long window = 16l << 20;
    char* d;
    unsigned int* sink;
    CHECK(hipMalloc(&d, 8 * window));
    CHECK(hipMemset(d, 1, 8 * window));
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int strides[] = {24, 25, 26, 28, 32, 24, 25};
    for (int nk : {24}) {
        for (int st : strides) {
            const int iters = 200;
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                probe<<<256, 512>>>(d, window, st, nk, iters, sink);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double bytes = 256.0 * iters * nk * 40 * 1024;
            printf("block stride %2d KiB: %.3f ms  %.2f TB/s L2 -> CU (window per XCD %.2f MB)\n", st, best, bytes / best / 1e9,
                   (4 * 24 + 8 * 16) * st / 1024.0);
        }
    }
    return 0;
}
