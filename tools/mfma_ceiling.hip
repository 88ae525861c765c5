// Power-limited ceilings of the bf16 matrix pipe on MI355X with RANDOM operands (measurement tool, not product code).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lightningdot_amd/csrc tools/mfma_ceiling.hip -o tools/bin/mfma_ceiling
//   tools/bin/mfma_ceiling [ms_per_variant]
//
// One workgroup per CU (256 persistent workgroups), v_mfma_f32_32x32x16_bf16 only.  Every variant runs the SAME number of
// MFMAs per CU and reports executed TFLOP/s, the effective shader clock (s_memtime ticks / s_memrealtime ticks x 100 MHz)
// and the matrix-pipe issue efficiency (32 cycles per MFMA and SIMD = 100 %).
//
//   geometry  W2 : 512 threads = 8 waves (2 x 4), wave tile 192 x 64  (12 accumulator tiles, 2 waves per SIMD)  [score_filter_r6]
//             W1 : 256 threads = 4 waves (2 x 2), wave tile 192 x 128 (24 accumulator tiles, 1 wave per SIMD, 512 registers)
//   feed      0  : register-resident fragments (MFMA only: the power ceiling of the pipe itself)
//             1  : + ds_read_b128 fragment stream from an LDS ring image (no global loads)
//             2  : + direct-to-LDS slab loads (buffer_load ... lds) through the 4-stage ring, counted vmcnt + one barrier per slab
//             3/4: 2 without the loads / without the barrier;  5: the loads spread between the MFMAs;  6: the two waves of a SIMD
//                  take turns issuing the pair's 10 pieces (measured: 78.6 % pipe issue vs 80.2 % for the plain burst — the idle
//                  partner still meets the issuing one at the slab barrier)
//   data      R  : N(0,1) random bf16 operands;  Z: all-zero operands (the DVFS give-back the guide describes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <random>
#include <vector>
#include <type_traits>

#include "gemm_ring.h"

using namespace ldot;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

constexpr int kStage = 40 * 1024;   // A slab 384 rows x 64 B + B slab 256 rows x 64 B
constexpr int kAOp = 24 * 1024;

template <int NW>   // waves per workgroup
struct Geo {
    static constexpr int MR = 6;
    static constexpr int NR = NW == 8 ? 2 : 4;
    static constexpr int WN = NW == 8 ? 4 : 2;     // waves along the query side
    static constexpr int kLoads = 40 / NW;         // 1 KiB direct-to-LDS loads per wave and slab
};

// One wave per SIMD: hipcc selects the AGPR form for EVERY MFMA of a kernel that may use more than 256 registers, so 24 accumulator
// tiles (384 registers) spill.  The tiles are therefore split by hand: row blocks 0..3 (16 tiles) accumulate in AGPRs, row blocks
// 4..5 (8 tiles) in VGPRs, through inline asm (same-accumulator MFMAs are 24 issues apart; hardware interlocks the rest).
template <bool AGPR>
__device__ __forceinline__ void mfma_asm(f32x16& acc, const bf16x8_t& a, const bf16x8_t& b) {
    if (AGPR)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <int NW, int FEED>
__global__ __launch_bounds__(NW * 64, NW / 4) void ceiling_kernel(const char* __restrict__ src, int64_t src_region, int nslab,
                                                                  float* __restrict__ sink, uint64_t* __restrict__ ticks) {
    using G = Geo<NW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RingCtx c;
    ring_ctx_init(c);
    const int wm = c.wave / G::WN, wn = c.wave % G::WN;
    const int xcd = blockIdx.x & 7;
    // every workgroup of an XCD streams the same window (one L2 miss per XCD and slab, like 8 query panels sharing a row tile)
    const char* base = src + (int64_t)xcd * src_region;
    __amdgpu_buffer_rsrc_t rs = ring_make_rsrc_n(base, src_region);
    const int vo = c.lane * 16;
    int issued = 0, so = 0;
    const int so_end = (int)src_region - kStage;
    auto issue = [&]() {
        char* st = smem + (issued & 3) * kStage;
#pragma unroll
        for (int j = 0; j < G::kLoads; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rg_lptr_t)(st + (j * NW + c.wave) * 1024), 16, vo,
                                                     so + (j * NW + c.wave) * 1024, 0, 0);
        ++issued;
        so += kStage;
        if (so >= so_end) so = 0;
    };
    // FEED 6: the two waves that share a SIMD (w and w ^ 4) take turns: on even slabs the waves 0..3 issue ALL 10 pieces of the pair, on
    // odd slabs the waves 4..7 — one wave of every SIMD keeps the matrix pipe busy while its partner is in the VMEM issue
    auto issue_alt = [&]() {
        char* st = smem + (issued & 3) * kStage;
        if ((c.wave >> 2) == (issued & 1)) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < G::kLoads; ++j) {
                    const int wt = (c.wave & 3) + 4 * h;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rg_lptr_t)(st + (j * NW + wt) * 1024), 16, vo, so + (j * NW + wt) * 1024, 0, 0);
                }
        }
        ++issued;
        so += kStage;
        if (so >= so_end) so = 0;
    };
    // the same slab issued piece by piece (FEED 5: one piece every few MFMAs instead of a burst after the k-step)
    auto issue_piece = [&](const int j) {
        char* st = smem + (issued & 3) * kStage;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rg_lptr_t)(st + (j * NW + c.wave) * 1024), 16, vo,
                                                 so + (j * NW + c.wave) * 1024, 0, 0);
        if (j == G::kLoads - 1) {
            ++issued;
            so += kStage;
            if (so >= so_end) so = 0;
        }
    };
    issue();
    issue();
    issue();
    issue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x16 acc[6][G::NR];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < G::NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // fragments as in score_filter_r6: the B set is double-buffered, every A fragment register is reloaded in place with the
    // next k-step's data right after the MFMAs that consume it
    bf16x8_t a[6], b[2][G::NR];
    {
        const char* a_w = smem + wm * (32 * 6 * 64) + c.frag_off0;
        const char* b_w = smem + kAOp + wn * (32 * G::NR * 64) + c.frag_off0;
#pragma unroll
        for (int i = 0; i < 6; ++i) a[i] = *(const bf16x8_t*)(a_w + i * 2048);
#pragma unroll
        for (int j = 0; j < G::NR; ++j) b[0][j] = b[1][j] = *(const bf16x8_t*)(b_w + j * 2048);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" ::: "memory");
    const uint64_t t0 = __builtin_readcyclecounter(), r0 = wall_clock64();

    // one k-step: MFMAs on (a, b[cur]) while a is reloaded from (nstage, noff) and b[cur ^ 1] is fetched
    auto step = [&](const int cur, const char* nstage, const int noff) {
        constexpr bool kDS = FEED != 0;
        const char* a_w = nstage + wm * (32 * 6 * 64) + noff;
        const char* b_w = nstage + kAOp + wn * (32 * G::NR * 64) + noff;
        if (kDS) {
#pragma unroll
            for (int j = 0; j < G::NR; ++j) b[cur ^ 1][j] = *(const bf16x8_t*)(b_w + j * 2048);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j < G::NR; ++j) {
                if (NW == 8)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[cur][j], acc[i][j], 0, 0, 0);
                else if (i < 4)
                    mfma_asm<true>(acc[i][j], a[i], b[cur][j]);
                else
                    mfma_asm<false>(acc[i][j], a[i], b[cur][j]);
            }
            if (kDS) a[i] = *(const bf16x8_t*)(a_w + i * 2048);
            if (FEED == 5) {
                // W2: 5 pieces per wave and slab: k-step 1 issues pieces 0,1,2 after row blocks 1,3,5; k-step 0 pieces 3,4 after 1,3
                // W1: 10 pieces: k-step 1 issues 0..5 (one per row block), k-step 0 issues 6..9 after row blocks 1..4
                if (NW == 8) {
                    if (cur == 1 && (i & 1)) issue_piece(i >> 1);
                    if (cur == 0 && (i == 1 || i == 3)) issue_piece(3 + (i >> 1));
                } else {
                    if (cur == 1) issue_piece(i);
                    if (cur == 0 && i >= 1 && i <= 4) issue_piece(5 + i);
                }
            }
        }
    };

#pragma unroll 1
    for (int s = 0; s < nslab; ++s) {
        const char* st0 = smem + (s & 3) * kStage;
        const char* st1 = smem + ((s + 1) & 3) * kStage;
        step(0, st0, c.frag_off0 ^ 32);
        __builtin_amdgcn_sched_barrier(0);
        if (FEED == 2 || FEED == 3 || FEED == 5 || FEED == 6) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (FEED != 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G::kLoads) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        step(1, st1, c.frag_off0);
        __builtin_amdgcn_sched_barrier(0);
        if (FEED == 2 || FEED == 4) issue();
        if (FEED == 6) issue_alt();
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    const uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < G::NR; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    if (sum == 12345.678f) sink[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) {
        ticks[blockIdx.x * 2] = t1 - t0;
        ticks[blockIdx.x * 2 + 1] = r1 - r0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NW, int FEED>
static void run(const char* name, const char* src, int64_t region, float* sink, uint64_t* ticks, double target_ms) {
    auto k = ceiling_kernel<NW, FEED>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kStage));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // MFMAs per slab and CU: both geometries cover a 384 x 256 x 32 slab = 96 tile-MFMAs x 2 k-steps
    const double flop_per_slab = 2.0 * 384 * 256 * 32 * 256;   // all 256 CUs
    int nslab = 2000;
    float ms = 0.f;
    for (int it = 0; it < 4; ++it) {   // calibrate, then 3 timed launches (the last two reported)
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(256), dim3(NW * 64), 4 * kStage, 0, src, region, nslab, sink, ticks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it == 0) nslab = (int)(nslab * target_ms / ms);
        if (it >= 2) {
            std::vector<uint64_t> h(512);
            CK(hipMemcpy(h.data(), ticks, 512 * 8, hipMemcpyDeviceToHost));
            double cyc = 0, real = 0;
            for (int i = 0; i < 256; ++i) {
                cyc += (double)h[2 * i];
                real += (double)h[2 * i + 1];
            }
            const double ghz = cyc / real * 0.1;
            const double tf = flop_per_slab * nslab / (ms * 1e-3) / 1e12;
            // issue efficiency: MFMAs per SIMD x 32 cycles / shader cycles of the loop
            const double mf_per_simd = 96.0 * 2 * nslab / 4;
            const double eff = mf_per_simd * 32.0 / (cyc / 256);
            printf("%-34s %8.3f ms  %8.1f TFLOP/s  %5.1f %% of 2500  clock %.3f GHz  pipe-issue %.1f %%\n", name, ms, tf,
                   tf / 25.0, ghz, eff * 100);
        }
    }
    fflush(stdout);
}


// MFMA-only comparison of the two dense bf16 shapes under the power limit: the same 192 accumulator registers and the same flops per
// wave, operands in registers — 12 tiles of v_mfma_f32_32x32x16_bf16 (16 accumulator registers each, K = 16 per instruction) vs 48 tiles
// of v_mfma_f32_16x16x32_bf16 (4 registers each, K = 32 per instruction: half the accumulator traffic per MAC, twice the operand reads)
template <int SHAPE>
__global__ __launch_bounds__(512, 2) void shape_kernel(const char* __restrict__ src, int iters, float* __restrict__ sink,
                                                       uint64_t* __restrict__ ticks) {
    const int lane = threadIdx.x & 63;
    bf16x8_t a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8_t*)(src + ((blockIdx.x & 7) * 65536 + i * 1024 + lane * 16));
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = *(const bf16x8_t*)(src + ((blockIdx.x & 7) * 65536 + 32768 + i * 1024 + lane * 16));
    const uint64_t t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    float sum = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)          // 2 x (12 tiles x 32*32*16) = one 192 x 64 x 32 slab
#pragma unroll
                for (int i = 0; i < 12; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + ks) & 7], b[(i + ks) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[i][r];
    } else {
        f32x4 acc[48];
#pragma unroll
        for (int i = 0; i < 48; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 48; ++i)            // 48 tiles x 16*16*32 = the same slab
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 7], b[(i >> 3) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 48; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    const uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (sum == 12345.678f) sink[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) {
        ticks[blockIdx.x * 2] = t1 - t0;
        ticks[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

template <int SHAPE>
static void run_shape(const char* name, const char* src, float* sink, uint64_t* ticks, double target_ms) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double flop_per_it = 2.0 * 192 * 64 * 32 * 8 * 256;   // 8 waves x 256 CUs
    int iters = 20000;
    float ms = 0.f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(shape_kernel<SHAPE>, dim3(256), dim3(512), 0, 0, src, iters, sink, ticks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it == 0) iters = (int)(iters * target_ms / ms);
        if (it >= 2) {
            std::vector<uint64_t> h(512);
            CK(hipMemcpy(h.data(), ticks, 512 * 8, hipMemcpyDeviceToHost));
            double cyc = 0, real = 0;
            for (int i = 0; i < 256; ++i) {
                cyc += (double)h[2 * i];
                real += (double)h[2 * i + 1];
            }
            printf("%-34s %8.3f ms  %8.1f TFLOP/s  %5.1f %% of 2500  clock %.3f GHz\n", name, ms, flop_per_it * iters / (ms * 1e-3) / 1e12,
                   flop_per_it * iters / (ms * 1e-3) / 1e12 / 25.0, cyc / real * 0.1);
        }
    }
    fflush(stdout);
}

// The slab loop of score_filter_r6 re-tiled for v_mfma_f32_16x16x32_bf16: wave tile 192 x 64 = 12 x 4 tiles of 16 x 16, ONE k-step per
// 32-deep slab (48 MFMAs), A fragments in a ring of 6 registers-quads (reloaded in place 24 MFMAs ahead), the 4 B fragments of the next
// slab fetched during the current one.  Fragment (16 rows x 32 k): lane l reads row l & 15, 16-byte chunk (l >> 4) ^ ((row >> 1) & 3).
// FEED as above: 1 ds_read stream only, 2 + direct-to-LDS burst, 5 + the loads spread between the MFMAs.  Upper bounds of taking the query
// operand out of the LDS: 7 = 2 without the B fragment reads (B stays in registers), 8 = 7 without the B pieces of the slab loads.
template <int FEED>
__global__ __launch_bounds__(512, 2) void ceiling16_kernel(const char* __restrict__ src, int64_t src_region, int nslab,
                                                           float* __restrict__ sink, uint64_t* __restrict__ ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int xcd = blockIdx.x & 7;
    const char* base = src + (int64_t)xcd * src_region;
    __amdgpu_buffer_rsrc_t rs = ring_make_rsrc_n(base, src_region);
    const int vo = lane * 16;
    int issued = 0, so = 0;
    const int so_end = (int)src_region - kStage;
    auto issue_piece = [&](const int j) {
        char* st = smem + (issued & 3) * kStage;
        if (FEED != 8 || j < 3)   // (8: the row-panel pieces only)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rg_lptr_t)(st + (j * 8 + wave) * 1024), 16, vo, so + (j * 8 + wave) * 1024, 0, 0);
        if (j == 4) {
            ++issued;
            so += kStage;
            if (so >= so_end) so = 0;
        }
    };
    auto issue = [&]() {
#pragma unroll
        for (int j = 0; j < 5; ++j)
            issue_piece(j);
    };
    issue();
    issue();
    issue();
    issue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int row = lane & 15;
    const int foff = row * 64 + (((lane >> 4) ^ ((row >> 1) & 3)) << 4);
    f32x4 acc[12][4];
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A fragments in a ring of THREE (row block i uses a[i % 3], which is refilled with block i + 3 right after its four MFMAs: 12 MFMAs
    // = 192 cycles ahead); the 4 B fragments of the next slab are fetched during the last three row blocks of the current one
    bf16x8_t a[3], b[2][4];
    const char* a_w0 = smem + wm * (192 * 64) + foff;
    const char* b_w0 = smem + kAOp + wn * (64 * 64) + foff;
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = *(const bf16x8_t*)(a_w0 + i * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[0][j] = b[1][j] = *(const bf16x8_t*)(b_w0 + j * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" ::: "memory");
    const uint64_t t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    auto slab = [&](auto cur_tag, const char* st0, const char* st1) {
        constexpr int cur = decltype(cur_tag)::value;
        const char* a_cur = st0 + wm * (192 * 64) + foff;
        const char* a_nxt = st1 + wm * (192 * 64) + foff;
        const char* b_nxt = st1 + kAOp + wn * (64 * 64) + foff;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (i == 9) {   // blocks 0..11 of this slab are in registers or consumed: the stage is free, the next one must have landed
                __builtin_amdgcn_sched_barrier(0);
                if (FEED == 2 || FEED == 5 || FEED == 7 || FEED == 8) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FEED == 8 ? 6 : 10) : "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i % 3], b[cur][j], acc[i][j], 0, 0, 0);
            a[i % 3] = *(const bf16x8_t*)((i + 3 < 12 ? a_cur + (i + 3) * 1024 : a_nxt + (i + 3 - 12) * 1024));
            if (i >= 9 && FEED != 7 && FEED != 8) {
                b[cur ^ 1][i - 9] = *(const bf16x8_t*)(b_nxt + (i - 9) * 1024);
                if (i == 11) b[cur ^ 1][3] = *(const bf16x8_t*)(b_nxt + 3 * 1024);
            }
            if (FEED == 5) {   // 5 pieces of the slab that goes into the stage just vacated: after row blocks 9, 10, 11 and 1, 3 of the next slab
                if (i >= 9) issue_piece(i - 9);
                if (i == 1 || i == 3) issue_piece(3 + (i >> 1));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (FEED == 2 || FEED == 7 || FEED == 8) issue();
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
#pragma unroll 1
    for (int s = 0; s + 1 < nslab; s += 2) {
        slab(C0{}, smem + (s & 3) * kStage, smem + ((s + 1) & 3) * kStage);
        slab(C1{}, smem + ((s + 1) & 3) * kStage, smem + ((s + 2) & 3) * kStage);
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    const uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 12345.678f) sink[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) {
        ticks[blockIdx.x * 2] = t1 - t0;
        ticks[blockIdx.x * 2 + 1] = r1 - r0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int FEED>
static void run16(const char* name, const char* src, int64_t region, float* sink, uint64_t* ticks, double target_ms) {
    auto k = ceiling16_kernel<FEED>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kStage));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double flop_per_slab = 2.0 * 384 * 256 * 32 * 256;
    int nslab = 2000;
    float ms = 0.f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 4 * kStage, 0, src, region, nslab, sink, ticks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it == 0) nslab = (int)(nslab * target_ms / ms);
        if (it >= 2) {
            std::vector<uint64_t> h(512);
            CK(hipMemcpy(h.data(), ticks, 512 * 8, hipMemcpyDeviceToHost));
            double cyc = 0, real = 0;
            for (int i = 0; i < 256; ++i) {
                cyc += (double)h[2 * i];
                real += (double)h[2 * i + 1];
            }
            const double tf = flop_per_slab * nslab / (ms * 1e-3) / 1e12;
            // 48 MFMAs of 16 cycles per slab and wave, 2 waves per SIMD
            printf("%-34s %8.3f ms  %8.1f TFLOP/s  %5.1f %% of 2500  clock %.3f GHz  pipe-issue %.1f %%\n", name, ms, tf, tf / 25.0,
                   cyc / real * 0.1, 48.0 * 2 * nslab * 16.0 / (cyc / 256) * 100);
        }
    }
    fflush(stdout);
}


// ---- round 4: one wave per SIMD on v_mfma_f32_16x16x32_bf16 -------------------------------------------------------------------------
// 96 accumulator tiles of 16 x 16 per wave (384 registers): tiles 0..63 accumulate in AGPRs, 64..95 in VGPRs (inline asm, as in W1 above).
template <bool AGPR>
__device__ __forceinline__ void mfma16_asm(f32x4& acc, const bf16x8_t& a, const bf16x8_t& b) {
    if (AGPR)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// T16W1: 256 threads = 4 waves (2 x 2), wave tile 192 x 128 = 12 x 8 tiles, both operands through the LDS ring (the 40 KiB stage of the
// product kernel).  LDS fragment bytes per MFMA: (12 + 8) KiB / 96 = 213 B against 341 B for the 192 x 64 wave tile at two waves per SIMD.
// FEED 1 ds_read stream only, 2 + direct-to-LDS burst (10 pieces per wave and slab), 5 the pieces spread between the row blocks.
template <int FEED>
__global__ __launch_bounds__(256, 1) void ceiling16w1_kernel(const char* __restrict__ src, int64_t src_region, int nslab,
                                                             float* __restrict__ sink, uint64_t* __restrict__ ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7;
    const char* base = src + (int64_t)xcd * src_region;
    __amdgpu_buffer_rsrc_t rs = ring_make_rsrc_n(base, src_region);
    const int vo = lane * 16;
    int issued = 0, so = 0;
    const int so_end = (int)src_region - kStage;
    auto issue_piece = [&](const int j) {
        char* st = smem + (issued & 3) * kStage;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rg_lptr_t)(st + (j * 4 + wave) * 1024), 16, vo, so + (j * 4 + wave) * 1024, 0, 0);
        if (j == 9) {
            ++issued;
            so += kStage;
            if (so >= so_end) so = 0;
        }
    };
    auto issue = [&]() {
#pragma unroll
        for (int j = 0; j < 10; ++j) issue_piece(j);
    };
    issue();
    issue();
    issue();
    issue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int row = lane & 15;
    const int foff = row * 64 + (((lane >> 4) ^ ((row >> 1) & 3)) << 4);
    f32x4 acc[12][8];
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8_t a[3], b[2][8];
    const char* a_w0 = smem + wm * (192 * 64) + foff;
    const char* b_w0 = smem + kAOp + wn * (128 * 64) + foff;
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = *(const bf16x8_t*)(a_w0 + i * 1024);
#pragma unroll
    for (int j = 0; j < 8; ++j) b[0][j] = b[1][j] = *(const bf16x8_t*)(b_w0 + j * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" ::: "memory");
    const uint64_t t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    auto slab = [&](auto cur_tag, const char* st0, const char* st1) {
        constexpr int cur = decltype(cur_tag)::value;
        const char* a_cur = st0 + wm * (192 * 64) + foff;
        const char* a_nxt = st1 + wm * (192 * 64) + foff;
        const char* b_nxt = st1 + kAOp + wn * (128 * 64) + foff;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (i == 9) {
                __builtin_amdgcn_sched_barrier(0);
                if (FEED == 2 || FEED == 5) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (i * 8 + j < 64)
                    mfma16_asm<true>(acc[i][j], a[i % 3], b[cur][j]);
                else
                    mfma16_asm<false>(acc[i][j], a[i % 3], b[cur][j]);
            }
            a[i % 3] = *(const bf16x8_t*)((i + 3 < 12 ? a_cur + (i + 3) * 1024 : a_nxt + (i + 3 - 12) * 1024));
            if (i >= 9) {   // the 8 B fragments of the next slab during the last three row blocks
                b[cur ^ 1][(i - 9) * 3] = *(const bf16x8_t*)(b_nxt + ((i - 9) * 3) * 1024);
                b[cur ^ 1][(i - 9) * 3 + 1] = *(const bf16x8_t*)(b_nxt + ((i - 9) * 3 + 1) * 1024);
                if (i < 11) b[cur ^ 1][(i - 9) * 3 + 2] = *(const bf16x8_t*)(b_nxt + ((i - 9) * 3 + 2) * 1024);
            }
            if (FEED == 5) {   // 10 pieces: after row blocks 9, 10, 11 and 0..6 of the next slab
                if (i >= 9) issue_piece(i - 9);
                if (i <= 6) issue_piece(3 + i);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (FEED == 2) issue();
        __builtin_amdgcn_sched_barrier(0);
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
#pragma unroll 1
    for (int s = 0; s + 1 < nslab; s += 2) {
        slab(C0{}, smem + (s & 3) * kStage, smem + ((s + 1) & 3) * kStage);
        slab(C1{}, smem + ((s + 1) & 3) * kStage, smem + ((s + 2) & 3) * kStage);
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    const uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 v = acc[i][j];
            if (i * 8 + j < 64) asm volatile("" : "+a"(v));
            sum += v[0] + v[1] + v[2] + v[3];
        }
    if (sum == 12345.678f) sink[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) {
        ticks[blockIdx.x * 2] = t1 - t0;
        ticks[blockIdx.x * 2 + 1] = r1 - r0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// AG: 256 threads = 4 waves stacked along the index rows, wave tile 96 rows x 256 queries = 6 x 16 tiles.  Only the QUERY slab (16 KiB)
// goes through the LDS (ring of 4 x 16 KiB, 4 direct-to-LDS pieces per wave and slab) and every wave reads all 16 of its fragments; the
// wave's own 6 row fragments come from L2 straight into VGPRs (a 1-KiB block of the blocked layout IS one MFMA operand), PD slabs ahead.
// Per slab and CU: 64 KiB of LDS fragment reads (128 in the product kernel), 16 KiB of LDS-DMA (40), 24 KiB of global -> VGPR loads (0).
// FEED 1: query fragment stream only (row fragments stay in registers), 2: + query LDS-DMA, 3: + row fragments from global memory (all).
template <int FEED, int PD>
__global__ __launch_bounds__(256, 1) void ceilingAG_kernel(const char* __restrict__ src, int64_t src_region, int nslab,
                                                           float* __restrict__ sink, uint64_t* __restrict__ ticks) {
    constexpr int kBStage = 16 * 1024;
    constexpr int NB = PD + 1;          // row-fragment buffers: the current slab + PD in flight
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7;
    const char* base = src + (int64_t)xcd * src_region;
    __amdgpu_buffer_rsrc_t rs = ring_make_rsrc_n(base, src_region);
    const int vo = lane * 16;
    const int so_end = (int)src_region - kStage;
    int b_issued = 0, so_b = 0, so_a = 0;
    auto issue_b = [&]() {   // the 16 KiB query slab: bytes [24 KiB, 40 KiB) of the source slab
        char* st = smem + (b_issued & 3) * kBStage;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rg_lptr_t)(st + (p * 4 + wave) * 1024), 16, vo, so_b + kAOp + (p * 4 + wave) * 1024, 0, 0);
        ++b_issued;
        so_b += kStage;
        if (so_b >= so_end) so_b = 0;
    };
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto load_a = [&](bf16x8_t (&dst)[6]) {   // the wave's 6 row fragments: bytes [wave * 6 KiB, + 6 KiB) of the source slab
#pragma unroll
        for (int i = 0; i < 6; ++i)
            dst[i] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so_a + (wave * 6 + i) * 1024, 0));
        so_a += kStage;
        if (so_a >= so_end) so_a = 0;
    };
    // FEED 4: the same loads spread over the slab, one VMEM instruction at a time: row fragment i after query block i (i < 6), the
    // fourth piece of the query slab in flight after block 6, the first three pieces of the next one after blocks 13, 14, 15
    auto issue_b_piece = [&](const int p) {
        char* st = smem + (b_issued & 3) * kBStage;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rg_lptr_t)(st + (p * 4 + wave) * 1024), 16, vo, so_b + kAOp + (p * 4 + wave) * 1024, 0, 0);
        if (p == 3) {
            ++b_issued;
            so_b += kStage;
            if (so_b >= so_end) so_b = 0;
        }
    };
    auto load_a_one = [&](bf16x8_t (&dst)[6], const int i) {
        dst[i] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so_a + (wave * 6 + i) * 1024, 0));
        if (i == 5) {
            so_a += kStage;
            if (so_a >= so_end) so_a = 0;
        }
    };
    bf16x8_t a[NB][6], bq[4];
    if (FEED >= 2) {
        issue_b();
        issue_b();
        issue_b();
        if (FEED == 4) {
            issue_b_piece(0);
            issue_b_piece(1);
            issue_b_piece(2);
        } else {
            issue_b();
        }
    } else {
        for (int i = threadIdx.x; i < 4 * kBStage / 16; i += 256) ((uint4*)smem)[i] = ((const uint4*)base)[i];
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        if (FEED >= 3 ? n < PD : true) load_a(a[n]);   // (FEED 3: slabs 0 .. PD-1 are in flight when the loop starts; slab s + PD is issued in slab s)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int row = lane & 15;
    const int foff = row * 64 + (((lane >> 4) ^ ((row >> 1) & 3)) << 4);
    f32x4 acc[16][6];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) bq[j] = *(const bf16x8_t*)(smem + foff + j * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" ::: "memory");
    const uint64_t t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    // one slab: query fragment j (ring of four register quads, refilled four fragments = 24 MFMAs ahead) against the wave's six row fragments
    auto slab = [&](auto n_tag, const char* st0, const char* st1) {
        constexpr int n = decltype(n_tag)::value;
        if (FEED == 3) load_a(a[(n + PD) % NB]);     // = the buffer of the slab that just finished
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j == 13) {   // fragments 0..15 of this stage are in registers or consumed; fragment reads of the next stage start below
                __builtin_amdgcn_sched_barrier(0);
                if (FEED >= 2) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    // query slab s + 1 must have landed: issued after it were (slab s - 1:) 6 row loads + 4 pieces, (slab s:) 6 row loads
                    // (+ slab s - 2: 6 row loads + 4 pieces) = 26 younger loads may stay in flight
                    if (FEED == 3)
                        asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
                    else if (FEED == 4)   // younger than the last piece of query slab s + 1: 3 + (6 + 1 + 3) + (6 + 1) loads
                        asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (FEED != 4) issue_b();   // the stage everybody has left takes the slab four ahead
                }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (j * 6 + i < 64)
                    mfma16_asm<true>(acc[j][i], a[n][i], bq[j & 3]);
                else
                    mfma16_asm<false>(acc[j][i], a[n][i], bq[j & 3]);
            }
            bq[j & 3] = *(const bf16x8_t*)((j + 4 < 16 ? st0 + (j + 4) * 1024 : st1 + (j + 4 - 16) * 1024) + foff);
            if (FEED == 4) {
                if (j < 6) load_a_one(a[(n + PD) % NB], j);
                if (j == 6) issue_b_piece(3);
                if (j >= 13) issue_b_piece(j - 13);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_assert(NB == 3, "PD = 2: three slabs ahead needs 96 + 128 + 16 VGPRs and spills");
#pragma unroll 1
    for (int s = 0; s + 12 <= nslab; s += 12) {   // 12 slabs per trip: a multiple of both NB and the 4 ring stages
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const char* st0 = smem + (u & 3) * kBStage;
            const char* st1 = smem + ((u + 1) & 3) * kBStage;
            if (u % NB == 0) slab(std::integral_constant<int, 0>{}, st0, st1);
            if (u % NB == 1) slab(std::integral_constant<int, 1>{}, st0, st1);
            if (u % NB == 2) slab(std::integral_constant<int, 2>{}, st0, st1);
            if (NB == 4 && u % NB == 3) slab(std::integral_constant<int, 3 % NB>{}, st0, st1);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    const uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            f32x4 v = acc[j][i];
            if (j * 6 + i < 64) asm volatile("" : "+a"(v));
            sum += v[0] + v[1] + v[2] + v[3];
        }
    if (sum == 12345.678f) sink[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) {
        ticks[blockIdx.x * 2] = t1 - t0;
        ticks[blockIdx.x * 2 + 1] = r1 - r0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// common driver of the one-wave-per-SIMD 16x16x32 kernels (256 threads, the same 384 x 256 x 32 slab per CU)
template <typename K>
static void run_w1(K k, int lds_bytes, int slab_multiple, const char* name, const char* src, int64_t region, float* sink, uint64_t* ticks,
                   double target_ms) {
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double flop_per_slab = 2.0 * 384 * 256 * 32 * 256;
    int nslab = 1200;
    float ms = 0.f;
    for (int it = 0; it < 4; ++it) {
        nslab = nslab / slab_multiple * slab_multiple;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(256), dim3(256), lds_bytes, 0, src, region, nslab, sink, ticks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it == 0) nslab = (int)(nslab * target_ms / ms);
        if (it >= 2) {
            std::vector<uint64_t> h(512);
            CK(hipMemcpy(h.data(), ticks, 512 * 8, hipMemcpyDeviceToHost));
            double cyc = 0, real = 0;
            for (int i = 0; i < 256; ++i) {
                cyc += (double)h[2 * i];
                real += (double)h[2 * i + 1];
            }
            const double tf = flop_per_slab * nslab / (ms * 1e-3) / 1e12;
            // 96 MFMAs of 16 cycles per slab and wave, 1 wave per SIMD
            printf("%-34s %8.3f ms  %8.1f TFLOP/s  %5.1f %% of 2500  clock %.3f GHz  pipe-issue %.1f %%\n", name, ms, tf, tf / 25.0,
                   cyc / real * 0.1, 96.0 * nslab * 16.0 / (cyc / 256) * 100);
        }
    }
    fflush(stdout);
}


// T16B (round 4): the product kernel's eight waves arranged 4 (rows) x 2 (queries): wave tile 96 x 128 = 6 x 8 tiles, the same 192 accumulator
// registers, 6 + 8 = 14 fragment reads per slab and wave instead of 12 + 4 = 16 (-12.5 % LDS read bytes).  The eight query fragments are
// single-buffered and refilled IN PLACE during the last row block (b[j] right after its last MFMA: seven MFMAs = 112 cycles before its first
// use in the next slab); A fragments in a ring of three as before.  The slab barrier sits before row block 3, where the first fragment of
// the next stage is read.  FEED 1: fragment stream only, 2: + slab-load burst, 5: loads spread.
template <int FEED, int ARING = 3>
__global__ __launch_bounds__(512, 2) void ceiling16b_kernel(const char* __restrict__ src, int64_t src_region, int nslab,
                                                            float* __restrict__ sink, uint64_t* __restrict__ ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7;
    const char* base = src + (int64_t)xcd * src_region;
    __amdgpu_buffer_rsrc_t rs = ring_make_rsrc_n(base, src_region);
    const int vo = lane * 16;
    int issued = 0, so = 0;
    const int so_end = (int)src_region - kStage;
    auto issue_piece = [&](const int j) {
        char* st = smem + (issued & 3) * kStage;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (rg_lptr_t)(st + (j * 8 + wave) * 1024), 16, vo, so + (j * 8 + wave) * 1024, 0, 0);
        if (j == 4) {
            ++issued;
            so += kStage;
            if (so >= so_end) so = 0;
        }
    };
    auto issue = [&]() {
#pragma unroll
        for (int j = 0; j < 5; ++j) issue_piece(j);
    };
    issue();
    issue();
    issue();
    issue();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int row = lane & 15;
    const int foff = row * 64 + (((lane >> 4) ^ ((row >> 1) & 3)) << 4);
    f32x4 acc[6][8];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8_t a[ARING], b[8];
    const char* a_w0 = smem + wm * (96 * 64) + foff;
    const char* b_w0 = smem + kAOp + wn * (128 * 64) + foff;
#pragma unroll
    for (int i = 0; i < ARING; ++i) a[i] = *(const bf16x8_t*)(a_w0 + i * 1024);
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = *(const bf16x8_t*)(b_w0 + j * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" ::: "memory");
    const uint64_t t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    auto slab = [&](const char* st0, const char* st1) {
        const char* a_cur = st0 + wm * (96 * 64) + foff;
        const char* a_nxt = st1 + wm * (96 * 64) + foff;
        const char* b_nxt = st1 + kAOp + wn * (128 * 64) + foff;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i == 6 - ARING) {   // every fragment of this stage is in registers; the next stage's first fragment is read below
                __builtin_amdgcn_sched_barrier(0);
                if (FEED == 2 || FEED == 5) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i % ARING], b[j], acc[i][j], 0, 0, 0);
                if (i == 5) b[j] = *(const bf16x8_t*)(b_nxt + j * 1024);
            }
            a[i % ARING] = *(const bf16x8_t*)((i + ARING < 6 ? a_cur + (i + ARING) * 1024 : a_nxt + (i + ARING - 6) * 1024));
            if (FEED == 5) {   // the slab that goes into the stage vacated at the barrier: pieces after row blocks 3, 4, 5 and 0, 1 of the next slab
                if (i >= 3) issue_piece(i - 3);
                if (i <= 1) issue_piece(3 + i);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (FEED == 2) issue();
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int s = 0; s + 1 < nslab; s += 2) {
        slab(smem + (s & 3) * kStage, smem + ((s + 1) & 3) * kStage);
        slab(smem + ((s + 1) & 3) * kStage, smem + ((s + 2) & 3) * kStage);
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    const uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 12345.678f) sink[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) {
        ticks[blockIdx.x * 2] = t1 - t0;
        ticks[blockIdx.x * 2 + 1] = r1 - r0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int FEED, int ARING = 3>
static void run16b(const char* name, const char* src, int64_t region, float* sink, uint64_t* ticks, double target_ms) {
    auto k = ceiling16b_kernel<FEED, ARING>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kStage));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double flop_per_slab = 2.0 * 384 * 256 * 32 * 256;
    int nslab = 2000;
    float ms = 0.f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 4 * kStage, 0, src, region, nslab, sink, ticks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it == 0) nslab = (int)(nslab * target_ms / ms);
        if (it >= 2) {
            std::vector<uint64_t> h(512);
            CK(hipMemcpy(h.data(), ticks, 512 * 8, hipMemcpyDeviceToHost));
            double cyc = 0, real = 0;
            for (int i = 0; i < 256; ++i) {
                cyc += (double)h[2 * i];
                real += (double)h[2 * i + 1];
            }
            const double tf = flop_per_slab * nslab / (ms * 1e-3) / 1e12;
            printf("%-34s %8.3f ms  %8.1f TFLOP/s  %5.1f %% of 2500  clock %.3f GHz  pipe-issue %.1f %%\n", name, ms, tf, tf / 25.0,
                   cyc / real * 0.1, 48.0 * 2 * nslab * 16.0 / (cyc / 256) * 100);
        }
    }
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double target_ms = argc > 1 ? atof(argv[1]) : 12.0;
    const int64_t region = 16ll << 20;   // per XCD window of the slab source (8 x 16 MiB: Infinity-Cache resident)
    std::vector<uint16_t> h((size_t)region * 8 / 2);
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : h) v = f32_to_bf16_bits(nd(rng));
    char *src, *zsrc;
    float* sink;
    uint64_t* ticks;
    CK(hipMalloc((void**)&src, region * 8));
    CK(hipMalloc((void**)&zsrc, region * 8));
    CK(hipMalloc((void**)&sink, 256 * 512 * 4));
    CK(hipMalloc((void**)&ticks, 512 * 8));
    CK(hipMemcpy(src, h.data(), region * 8, hipMemcpyHostToDevice));
    CK(hipMemset(zsrc, 0, region * 8));
    if (argc > 2 && !strcmp(argv[2], "t16b")) {   // only the product kernel's slab loop (PMC runs: tools/pmc_waits.sh)
        run16b<2, 2>("T16B burst, A ring of TWO     random", src, region, sink, ticks, target_ms);
        return 0;
    }
    for (int rep = 0; rep < 1; ++rep) {
        printf("---- repetition %d (target %.1f ms per launch) ----\n", rep, target_ms);
        run16b<1>("T16B (4x2 waves) mfma+ds_read  random", src, region, sink, ticks, target_ms);
        run16b<2>("T16B mfma+ds_read+lds-dma     random", src, region, sink, ticks, target_ms);
        run16b<5>("T16B full, dma interleaved    random", src, region, sink, ticks, target_ms);
        run16b<2, 2>("T16B burst, A ring of TWO     random", src, region, sink, ticks, target_ms);
        run16b<1, 2>("T16B ds_read only, A ring 2   random", src, region, sink, ticks, target_ms);
        run_w1(ceiling16w1_kernel<1>, 4 * kStage, 2, "T16W1 mfma+ds_read         random", src, region, sink, ticks, target_ms);
        run_w1(ceiling16w1_kernel<2>, 4 * kStage, 2, "T16W1 mfma+ds_read+lds-dma random", src, region, sink, ticks, target_ms);
        run_w1(ceiling16w1_kernel<5>, 4 * kStage, 2, "T16W1 full, dma interleaved random", src, region, sink, ticks, target_ms);
        run_w1(ceilingAG_kernel<1, 2>, 65536, 12, "AG q ds_read only          random", src, region, sink, ticks, target_ms);
        run_w1(ceilingAG_kernel<2, 2>, 65536, 12, "AG q ds_read + q lds-dma   random", src, region, sink, ticks, target_ms);
        run_w1(ceilingAG_kernel<3, 2>, 65536, 12, "AG full, rows 2 slabs ahead random", src, region, sink, ticks, target_ms);
        run_w1(ceilingAG_kernel<4, 2>, 65536, 12, "AG full, loads spread      random", src, region, sink, ticks, target_ms);
        run_w1(ceilingAG_kernel<4, 2>, 65536, 12, "AG full, loads spread      zeros", zsrc, region, sink, ticks, target_ms);
        run_w1(ceilingAG_kernel<3, 2>, 65536, 12, "AG full, rows 2 slabs ahead zeros", zsrc, region, sink, ticks, target_ms);
        run16<1>("T16 mfma+ds_read           random", src, region, sink, ticks, target_ms);
        run16<2>("T16 mfma+ds_read+lds-dma   random", src, region, sink, ticks, target_ms);
        run16<5>("T16 full, dma interleaved  random", src, region, sink, ticks, target_ms);
        run16<7>("T16 no B fragment reads    random", src, region, sink, ticks, target_ms);
        run16<8>("T16 no B reads, no B dma   random", src, region, sink, ticks, target_ms);
        run16<2>("T16 mfma+ds_read+lds-dma   zeros", zsrc, region, sink, ticks, target_ms);
        run_shape<32>("shape 32x32x16 mfma-only   random", src, sink, ticks, target_ms);
        run_shape<16>("shape 16x16x32 mfma-only   random", src, sink, ticks, target_ms);
        run_shape<32>("shape 32x32x16 mfma-only   zeros", zsrc, sink, ticks, target_ms);
        run_shape<16>("shape 16x16x32 mfma-only   zeros", zsrc, sink, ticks, target_ms);
        run<8, 0>("W2 mfma-only            random", src, region, sink, ticks, target_ms);
        run<8, 0>("W2 mfma-only            zeros", zsrc, region, sink, ticks, target_ms);
        run<4, 0>("W1 mfma-only            random", src, region, sink, ticks, target_ms);
        run<4, 0>("W1 mfma-only            zeros", zsrc, region, sink, ticks, target_ms);
        run<8, 1>("W2 mfma+ds_read         random", src, region, sink, ticks, target_ms);
        run<4, 1>("W1 mfma+ds_read         random", src, region, sink, ticks, target_ms);
        run<8, 2>("W2 mfma+ds_read+lds-dma random", src, region, sink, ticks, target_ms);
        run<4, 2>("W1 mfma+ds_read+lds-dma random", src, region, sink, ticks, target_ms);
        run<8, 2>("W2 mfma+ds_read+lds-dma zeros", zsrc, region, sink, ticks, target_ms);
        run<4, 2>("W1 mfma+ds_read+lds-dma zeros", zsrc, region, sink, ticks, target_ms);
        run<8, 3>("W2 ds_read+barrier, no dma random", src, region, sink, ticks, target_ms);
        run<4, 3>("W1 ds_read+barrier, no dma random", src, region, sink, ticks, target_ms);
        run<8, 4>("W2 ds_read+dma, no barrier random", src, region, sink, ticks, target_ms);
        run<4, 4>("W1 ds_read+dma, no barrier random", src, region, sink, ticks, target_ms);
        run<8, 5>("W2 full, dma interleaved   random", src, region, sink, ticks, target_ms);
        run<4, 5>("W1 full, dma interleaved   random", src, region, sink, ticks, target_ms);
        run<8, 6>("W2 full, SIMD partners alternate random", src, region, sink, ticks, target_ms);
        run<8, 6>("W2 full, SIMD partners alternate zeros", zsrc, region, sink, ticks, target_ms);
        run<8, 5>("W2 full, dma interleaved   zeros", zsrc, region, sink, ticks, target_ms);
        run<4, 5>("W1 full, dma interleaved   zeros", zsrc, region, sink, ticks, target_ms);
    }
    return 0;
}
