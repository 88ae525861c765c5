// bf16 MFMA engine, second generation: a continuous stream of 32-deep K slabs through a 4-stage LDS ring.
//
//   * workgroup = 512 threads = 8 waves (2 along M x 4 along N), wave tile 128 x 64 (4 x 2 MFMA 32x32 tiles)
//   * one slab = A rows [256] x 32 k  +  B rows [256] x 32 k  = 2 x 16 KiB; ring of 4 slabs = 128 KiB LDS
//   * direct-to-LDS loads (global_load_lds_dwordx4); 64-byte LDS rows, 16-byte chunk index XOR ((row >> 2) & 3)
//     applied on the per-lane SOURCE address and on the ds_read address -> conflict-free ds_read_b128
//   * slabs are issued 3 ahead of the one being multiplied (96 KiB in flight per CU) and retired with a COUNTED
//     s_waitcnt vmcnt(8): the loads of the next slabs stay in flight across the single raw s_barrier per slab
//   * fragments are double-buffered in registers (F: k-step 0, G: k-step 1 of a slab) and the MFMAs of the last
//     k-step of slab s are issued AFTER the barrier that opens slab s+1, so the matrix pipe has 8 MFMAs per wave
//     to chew on while the first ds_reads of the new slab are in flight (no restart bubble)
//   * the slab stream runs across output tiles: the loads of the next tile's first slabs are in flight while the
//     current tile's epilogue (the threshold filter) runs.
#pragma once
#include "ldot_common.h"

namespace ldot {

constexpr int kRBM = 256, kRBN = 256, kRBK = 32;
constexpr int kRingThreads = 512;
constexpr int kRingStages = 4;
constexpr int kROpBytes = kRBM * kRBK * 2;        // 16 KiB per operand slab
constexpr int kRStageBytes = 2 * kROpBytes;       // 32 KiB
constexpr int kRingLdsBytes = kRingStages * kRStageBytes;   // 128 KiB
constexpr int kRLoadsPerSlab = 4;                 // global_load_lds per thread per slab (2 A + 2 B)

typedef const __attribute__((address_space(1))) void* rg_gptr_t;
typedef __attribute__((address_space(3))) void* rg_lptr_t;

struct Frags {
    bf16x8_t a[4], b[2];
};

struct RingCtx {
    int lane, wave, wm, wn;
    int frag_off[2];   // per-lane byte offset of k-step ks inside a 32-row block (64-byte rows)
    int st_row[2];     // source row (within the 256-row slab) of staging instruction j
    int st_col;        // source byte offset inside the 64-byte slab row
};

// Buffer-addressed staging: the operand panel of the current tile is described by a buffer resource (SGPRs, base
// = first row of the panel), each lane keeps two constant 32-bit offsets (its rows of the two staging
// instructions) and the K position rides in the scalar offset -> ZERO vector instructions per load.
struct RingSrc {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[2];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ring_make_rsrc(const char* panel_base, int64_t ld_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)panel_base, 0, (int)(kRBM * ld_bytes), 0x00020000);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ring_make_rsrc_n(const char* panel_base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)panel_base, 0, (int)bytes, 0x00020000);
}

template <int AUX = 0>
__device__ __forceinline__ void ring_stage_operand_buf(const RingCtx& c, const RingSrc& src, int k0b, char* lds_slab) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(src.rsrc, (rg_lptr_t)(lds_slab + (j * 8 + c.wave) * 1024), 16,
                                                 src.voff[j], k0b, 0, AUX);
}

// MFMA with a zero C operand (first k-step of an output tile): no accumulator clearing pass needed
__device__ __forceinline__ void ring_mfma_first(const Frags& f, f32x16 (&acc)[4][2]) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mr = 0; mr < 4; ++mr)
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
            acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mr], f.b[nr], z, 0, 0, 0);
}

// hand a loaded value to the compiler as a plain register (its own s_waitcnt for the load is placed HERE, not at a
// later first use inside the pipelined loop where it would drain the in-flight LDS-DMA slabs)
__device__ __forceinline__ float ring_launder(float v) {
    float o;
    asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(v));
    return o;
}

__device__ __forceinline__ void ring_ctx_init(RingCtx& c) {
    c.lane = threadIdx.x & 63;
    c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.wm = c.wave >> 2;
    c.wn = c.wave & 3;
    const int r = c.lane & 31;
    const int x = (r >> 2) & 3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) c.frag_off[ks] = r * 64 + ((((ks << 1) | (c.lane >> 5)) ^ x) << 4);
    // staging instruction j of wave w fills LDS bytes [(j*8+w)*1024, +1024) = slab rows (j*8+w)*16 .. +16;
    // lane i lands on row (i >> 2), physical chunk (i & 3) and therefore fetches logical chunk (i&3) ^ swz(row)
#pragma unroll
    for (int j = 0; j < 2; ++j) c.st_row[j] = (j * 8 + c.wave) * 16 + (c.lane >> 2);
    c.st_col = (((c.lane & 3) ^ ((c.lane >> 4) & 3)) << 4);
}

__device__ __forceinline__ void ring_stage_operand(const RingCtx& c, const char* __restrict__ base, int64_t ld_bytes,
                                          int64_t row0, int k0b, char* lds_slab) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const char* src = base + (row0 + c.st_row[j]) * ld_bytes + k0b + c.st_col;
        __builtin_amdgcn_global_load_lds((rg_gptr_t)src, (rg_lptr_t)(lds_slab + (j * 8 + c.wave) * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ void ring_read_frags(const RingCtx& c, const char* stage, int ks, Frags& f) {
    const char* a_w = stage + c.wm * (128 * 64) + c.frag_off[ks];
    const char* b_w = stage + kROpBytes + c.wn * (64 * 64) + c.frag_off[ks];
#pragma unroll
    for (int mr = 0; mr < 4; ++mr) f.a[mr] = *(const bf16x8_t*)(a_w + mr * 2048);
#pragma unroll
    for (int nr = 0; nr < 2; ++nr) f.b[nr] = *(const bf16x8_t*)(b_w + nr * 2048);
}

__device__ __forceinline__ void ring_mfma(const Frags& f, f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int mr = 0; mr < 4; ++mr)
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
            acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mr], f.b[nr], acc[mr][nr], 0, 0, 0);
}

__device__ __forceinline__ void ring_zero(f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int mr = 0; mr < 4; ++mr)
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.f;
}

// wait until at most `slabs_in_flight` slabs (4 loads each) of this thread's direct-to-LDS loads are outstanding
__device__ __forceinline__ void ring_wait_loads(int slabs_in_flight) {
    if (slabs_in_flight >= 3)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (slabs_in_flight == 2)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (slabs_in_flight == 1)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---- third generation: (64*MR) x 256 tile, A fragments recycled in place -------------------------------------------
// Wave tile (32*MR) x 64.  Per k-step the wave needs MR A fragments + 2 B fragments; only the B pair is double-buffered,
// every A fragment register is reloaded with the NEXT k-step's data right after the two MFMAs that consume it were
// issued (it then has (MR-1)*2 MFMA times to land).  MR = 6: 192 accumulator + 24 + 16 fragment VGPRs.
template <int MR>
struct RingGeom {
    static constexpr int kBM = 64 * MR;                        // index rows per tile
    static constexpr int kAOpBytes = kBM * kRBK * 2;           // A slab bytes
    static constexpr int kStage = kAOpBytes + kROpBytes;       // + B slab (256 queries)
    static constexpr int kLds = kRingStages * kStage;
    static constexpr int kALoads = kAOpBytes / 1024 / 8;       // direct-to-LDS loads per thread per slab (A)
    static constexpr int kLoads = kALoads + 2;
};

template <int MR>
struct FragsR {
    bf16x8_t a[MR];
    bf16x8_t b[2][2];
};

template <int MR>
__device__ __forceinline__ void ringr_read_b(const RingCtx& c, const char* stage, int ks, bf16x8_t (&b)[2]) {
    const char* b_w = stage + RingGeom<MR>::kAOpBytes + c.wn * (64 * 64) + c.frag_off[ks];
#pragma unroll
    for (int nr = 0; nr < 2; ++nr) b[nr] = *(const bf16x8_t*)(b_w + nr * 2048);
}

// one k-step: MFMAs on (a[*], bcur) while a[*] is reloaded from (nstage, nks) and bnext is fetched
template <int MR, bool SKIP_B = false>
__device__ __forceinline__ void ringr_step(const RingCtx& c, FragsR<MR>& f, const int cur, const char* nstage,
                                           const int nks, f32x16 (&acc)[MR][2]) {
    if (!SKIP_B) ringr_read_b<MR>(c, nstage, nks, f.b[cur ^ 1]);   // (SKIP_B: profiling ablation only)
    const char* a_w = nstage + c.wm * (32 * MR * 64) + c.frag_off[nks];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        acc[mr][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mr], f.b[cur][0], acc[mr][0], 0, 0, 0);
        acc[mr][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mr], f.b[cur][1], acc[mr][1], 0, 0, 0);
        f.a[mr] = *(const bf16x8_t*)(a_w + mr * 2048);
    }
}

template <int MR>
struct RingSrcR {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[RingGeom<MR>::kALoads > 2 ? RingGeom<MR>::kALoads : 2];
};

}  // namespace ldot
