// bf16 MFMA engine on an LDS ring: a continuous stream of 32-deep K slabs through a 4-stage ring, (64*MR) x 256 tile.
//
//   * workgroup = 512 threads = 8 waves (2 along M x 4 along N), wave tile (32*MR) x 64 (MR x 2 MFMA 32x32 tiles)
//   * one slab = A rows [64*MR] x 32 k  +  B rows [256] x 32 k; MR = 6: 24 + 16 KiB, ring of 4 slabs = 160 KiB = all LDS
//   * direct-to-LDS loads (buffer_load ... lds, 16 B/lane); 64-byte LDS rows, 16-byte chunk index XOR ((row >> 2) & 3)
//     applied on the per-lane SOURCE address and on the ds_read address -> conflict-free ds_read_b128
//   * slabs are issued 3 ahead of the one being multiplied and retired with a COUNTED s_waitcnt vmcnt: the loads of the
//     next slabs stay in flight across the single raw s_barrier per slab
//   * only the two B fragments are double-buffered in registers; every A fragment register is reloaded in place with
//     the next k-step's data right after the two MFMAs that consume it (it then has (MR-1)*2 MFMA times to land)
//   * the slab stream runs across output tiles: the loads of the next tile's first slabs are in flight while the
//     current tile's epilogue (the threshold filter) runs.
// (Earlier generations — 256x256 with a two-stage BK=64 pipeline and with double-buffered fragments on this ring — are in
// the history: commits da2a5a4, 62cbd68.)
#pragma once
#include "ldot_common.h"

namespace ldot {

constexpr int kRBN = 256, kRBK = 32;              // queries per tile, K depth of a slab
constexpr int kRingThreads = 512;
constexpr int kRingStages = 4;
constexpr int kROpBytes = kRBN * kRBK * 2;        // 16 KiB: the B (query) slab

typedef __attribute__((address_space(3))) void* rg_lptr_t;

struct RingCtx {
    int lane, wave, wm, wn;
    int frag_off0;     // per-lane byte offset of k-step 0 inside a 32-row block (64-byte rows); k-step 1 = ^ 32
    int st_col;        // source byte offset inside the 64-byte slab row
};

// Buffer-addressed staging: the operand panel of the current tile is described by a buffer resource (SGPRs, base
// = first row of the panel), each lane keeps constant 32-bit offsets (its rows of the staging instructions) and the
// K position rides in the scalar offset -> ZERO vector instructions per load.
struct RingSrc {
    __amdgpu_buffer_rsrc_t rsrc;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ring_make_rsrc_n(const char* panel_base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)panel_base, 0, (int)bytes, 0x00020000);
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
    c.frag_off0 = r * 64 + (((c.lane >> 5) ^ x) << 4);   // chunk index ((ks << 1) | half) ^ x: ks toggles bit 1 = 32 bytes
    // staging instruction j of wave w fills LDS bytes [(j*8+w)*1024, +1024) = slab rows (j*8+w)*16 .. +16;
    // lane i lands on row (i >> 2), physical chunk (i & 3) and therefore fetches logical chunk (i&3) ^ swz(row)
    c.st_col = (((c.lane & 3) ^ ((c.lane >> 4) & 3)) << 4);
}

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
    const char* b_w = stage + RingGeom<MR>::kAOpBytes + c.wn * (64 * 64) + (c.frag_off0 ^ (ks << 5));
#pragma unroll
    for (int nr = 0; nr < 2; ++nr) b[nr] = *(const bf16x8_t*)(b_w + nr * 2048);
}

struct RingNoHook {
    __device__ __forceinline__ void operator()(int) const {}
};

// one k-step: MFMAs on (a[*], bcur) while a[*] is reloaded from (nstage, nks) and bnext is fetched.  hook(mr) runs after
// row block mr: the fused kernel issues its direct-to-LDS slab loads there ONE PIECE AT A TIME instead of as a burst after the
// k-step (a burst of 8 waves x 5 pieces fills the texture-address FIFO and every wave blocks on VMEM issue while the matrix pipe
// drains: tools/mfma_ceiling.hip measures 80 % -> 92 % matrix-pipe issue for this change alone).
template <int MR, bool SKIP_B = false, typename Hook = RingNoHook>
__device__ __forceinline__ void ringr_step(const RingCtx& c, FragsR<MR>& f, const int cur, const char* nstage,
                                           const int nks, f32x16 (&acc)[MR][2], const Hook& hook = Hook()) {
    if (!SKIP_B) ringr_read_b<MR>(c, nstage, nks, f.b[cur ^ 1]);   // (SKIP_B: profiling ablation only)
    const char* a_w = nstage + c.wm * (32 * MR * 64) + (c.frag_off0 ^ (nks << 5));
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        acc[mr][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mr], f.b[cur][0], acc[mr][0], 0, 0, 0);
        acc[mr][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mr], f.b[cur][1], acc[mr][1], 0, 0, 0);
        f.a[mr] = *(const bf16x8_t*)(a_w + mr * 2048);
        hook(mr);
    }
}

}  // namespace ldot
