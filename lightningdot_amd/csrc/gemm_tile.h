// bf16 MFMA tile engine for C[M x N] = A[M x K] . B[N x K]^T  (both operands K-contiguous, fp32 accumulate).
//
// CDNA4 / gfx950 design:
//   * 256 x 256 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N); each wave owns 128 x 64 =
//     4 x 2 MFMA 32x32 tiles = 128 accumulator registers), K stepped in slabs of 64.
//   * operands go HBM/L2 -> LDS with direct-to-LDS loads (global_load_lds_dwordx4, 16 B per lane); the LDS
//     image is lane-linear, so the bank-conflict swizzle is applied to the per-lane *source* address and to
//     the ds_read address (16-byte chunk index XOR ((row >> 1) & 7)); with it every ds_read_b128 lane group
//     hits 16 distinct 16-B slots.
//   * two LDS stages (2 x 64 KiB): the loads of slab t+1 are issued before the MFMAs of slab t and drained
//     at the single barrier that ends the slab.
//   * fragments: v_mfma_f32_32x32x16_bf16; lane l supplies row (l & 31), k-chunk (l >> 5) of each 16-wide
//     k-step for A and for B alike (any k permutation common to A and B leaves the dot product unchanged);
//     C/D: col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).
#pragma once
#include "ldot_common.h"

namespace ldot {

constexpr int kBM = 256, kBN = 256, kBK = 64;
constexpr int kGemmThreads = 512;
constexpr int kTileBytes = kBM * kBK * 2;         // 32 KiB per operand slab
constexpr int kStageBytes = 2 * kTileBytes;       // A + B
constexpr int kGemmLdsBytes = 2 * kStageBytes;    // two stages = 128 KiB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct TileCtx {
    int lane, wave, wm, wn;
    int frag_off[4];   // per-lane byte offset of the k-step fragment inside a 32-row block
    int st_row[4];     // per-lane source row (within the 256-row slab) of staging instruction j
    int st_col;        // per-lane source byte offset inside the 128-byte slab row
};

__device__ __forceinline__ void tile_ctx_init(TileCtx& c) {
    c.lane = threadIdx.x & 63;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.wm = c.wave >> 2;   // 0..1
    c.wn = c.wave & 3;    // 0..3
    const int r = c.lane & 31;
    const int x = (r >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) c.frag_off[ks] = r * 128 + ((((ks << 1) | (c.lane >> 5)) ^ x) << 4);
    // staging: instruction j of wave w fills LDS bytes [(j*8+w)*1024, +1024) = rows (j*8+w)*8 .. +8;
    // lane i lands on row (i >> 3), physical chunk (i & 7); it must therefore FETCH logical chunk p ^ swz(row)
#pragma unroll
    for (int j = 0; j < 4; ++j) c.st_row[j] = (j * 8 + c.wave) * 8 + (c.lane >> 3);
    const int row0 = c.wave * 8 + (c.lane >> 3);          // row parity/swizzle is the same for every j (j*64 rows)
    c.st_col = (((c.lane & 7) ^ ((row0 >> 1) & 7)) << 4);
}

// Issue the direct-to-LDS loads of one operand slab: rows [row0, row0+256) x k-bytes [k0b, k0b+128).
__device__ __forceinline__ void stage_slab(const TileCtx& c, const char* __restrict__ base, int64_t ld_bytes, int64_t row0,
                                  int k0b, char* lds_slab) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const char* src = base + (row0 + c.st_row[j]) * ld_bytes + k0b + c.st_col;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_slab + (j * 8 + c.wave) * 1024), 16, 0, 0);
    }
}

// One 64-deep slab of MFMAs for this wave: acc[4][2] += A(128 x 64) . B(64 x 64)^T
__device__ __forceinline__ void compute_slab(const TileCtx& c, const char* a_slab, const char* b_slab, f32x16 (&acc)[4][2]) {
    const char* a_w = a_slab + c.wm * (128 * 128);
    const char* b_w = b_slab + c.wn * (64 * 128);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bf16x8_t a[4], b[2];
#pragma unroll
        for (int mr = 0; mr < 4; ++mr) a[mr] = *(const bf16x8_t*)(a_w + mr * 4096 + c.frag_off[ks]);
#pragma unroll
        for (int nr = 0; nr < 2; ++nr) b[nr] = *(const bf16x8_t*)(b_w + nr * 4096 + c.frag_off[ks]);
#pragma unroll
        for (int mr = 0; mr < 4; ++mr)
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
                acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mr], b[nr], acc[mr][nr], 0, 0, 0);
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int mr = 0; mr < 4; ++mr)
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.f;
}

// Ablation switches (debug builds of the filter kernel only; VAR = 0 is the product path):
//   bit 1: no staging after the first slab      bit 2: no MFMA/ds_read work      bit 3: (reserved)
// Full K loop for one 256x256 tile (non-persistent form).  A rows [m0,m0+256), B rows [n0,n0+256); K = nk*64.
template <int VAR = 0>
__device__ __forceinline__ void gemm_tile(const TileCtx& c, const char* __restrict__ A, int64_t lda_b, int64_t m0,
                                 const char* __restrict__ B, int64_t ldb_b, int64_t n0, int nk, char* lds,
                                 f32x16 (&acc)[4][2]) {
    zero_acc(acc);
    stage_slab(c, A, lda_b, m0, 0, lds);
    stage_slab(c, B, ldb_b, n0, 0, lds + kTileBytes);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        char* st = lds + cur * kStageBytes;
        if (kt + 1 < nk && !(VAR & 2)) {
            char* nx = lds + (cur ^ 1) * kStageBytes;
            stage_slab(c, A, lda_b, m0, (kt + 1) * 128, nx);
            stage_slab(c, B, ldb_b, n0, (kt + 1) * 128, nx + kTileBytes);
        }
        if (!(VAR & 4)) compute_slab(c, st, st + kTileBytes, acc);
        __syncthreads();
        cur ^= 1;
    }
}

// Bijective XCD-aware remap of a linear workgroup id (block b runs on XCD b % 8): gives each XCD a contiguous
// range of logical ids so that neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    const int q = nwg / nx, r = nwg % nx;
    const int xcd = bid % nx, loc = bid / nx;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + loc;
}

}  // namespace ldot
