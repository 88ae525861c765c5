// Probe (measurement tool): read bandwidth of a window streamed repeatedly by all CUs, by window size — where does the Infinity Cache stop
// retaining a re-read range?  256 x 4 workgroups x 256 threads, 16-byte loads, each pass reads the whole window once (coalesced, every
// workgroup its own interleaved 4-KiB pieces).  usage: tools/bin/mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void stream(const char* __restrict__ src, long bytes, int passes, unsigned int* sink) {
    const long piece = 4096, npieces = bytes / piece;
    u32x4 acc = {0, 0, 0, 0};
    for (int p = 0; p < passes; ++p)
        for (long i = blockIdx.x; i < npieces; i += gridDim.x) {
            const u32x4 v = NT ? __builtin_nontemporal_load((const u32x4*)(src + i * piece + threadIdx.x * 16)) : *(const u32x4*)(src + i * piece + threadIdx.x * 16);
            acc ^= v;
        }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main() {
    const long maxb = 2048l << 20;
    char* d;
    unsigned int* sink;
    CHECK(hipMalloc(&d, maxb));
    CHECK(hipMemset(d, 1, maxb));
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int mbs[] = {8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 2048};
    for (int nt = 0; nt < 2; ++nt)
    for (int mb : mbs) {
        const long bytes = (long)mb << 20;
        const int passes = (int)((16l << 30) / bytes) < 4 ? 4 : (int)((16l << 30) / bytes);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            if (nt) stream<true><<<1024, 256>>>(d, bytes, passes, sink); else stream<false><<<1024, 256>>>(d, bytes, passes, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%s loads, window %5d MB re-read %4d times: %.2f TB/s\n", nt ? "non-temporal" : "plain", mb, passes, (double)bytes * passes / best / 1e9);
    }
    return 0;
}
