// Shared device/host helpers for the lightningdot_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ldot.h"

namespace ldot {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// round-to-nearest-even fp32 -> bf16 bits (NaN kept quiet)
__host__ __device__ inline uint16_t f32_to_bf16_bits(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__host__ __device__ inline float bf16_bits_to_f32(uint16_t b) {
    union { float f; uint32_t u; } v;
    v.u = ((uint32_t)b) << 16;
    return v.f;
}
__host__ __device__ inline float f16_bits_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    union { float f; uint32_t u; } v;
    if (exp == 0) {
        if (man == 0) { v.u = sign; return v.f; }
        // subnormal
        float m = (float)man * (1.0f / 1024.0f) * (1.0f / 16384.0f);
        return (h & 0x8000u) ? -m : m;
    }
    if (exp == 31) { v.u = sign | 0x7f800000u | (man << 13); return v.f; }
    v.u = sign | ((exp + 112u) << 23) | (man << 13);
    return v.f;
}

// Order-preserving map float -> uint32 such that LARGER float  <=>  SMALLER key ("descending key").
// -0.0 and +0.0 map to different keys (+0 better than -0); NaNs sort by their bit pattern.
__host__ __device__ inline uint32_t desc_key(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    uint32_t u = v.u;
    uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~asc;
}
__host__ __device__ inline float desc_key_to_float(uint32_t k) {
    uint32_t asc = ~k;
    uint32_t u = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
    union { float f; uint32_t u; } v;
    v.u = u;
    return v.f;
}

// wave-wide integer sum in 7 DPP adds + one readlane (quad swaps, row mirrors, row broadcasts: the total lands in lane 63): pure VALU —
// counting with ballots costs scalar-unit issue slots, which the waves of a CU share, and a bpermute-based __shfl_xor reduction ~600 cycles
__device__ __forceinline__ int wave_sum_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);   // row_mirror  -> every lane: its row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

__host__ __device__ inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

void set_error(const char* fmt, ...);

#define LDOT_HIP_CHECK(expr)                                                                         \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            ::ldot::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (_e == hipErrorOutOfMemory) ? LDOT_ENOMEM : LDOT_EDEVICE;                         \
        }                                                                                            \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and device instead of before every launch: the call costs
// microseconds of host time, which is on the critical path of the short searches (small indexes, few queries).  `flags` is a
// function-local static array of the launcher; a race between two threads sets the attribute twice, which is harmless.
constexpr int kAttrDevices = 64;
inline hipError_t set_max_dynamic_lds_once(const void* kernel, int bytes, bool* flags) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < kAttrDevices && flags[dev]) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < kAttrDevices) flags[dev] = true;
    return e;
}

#define LDOT_REQUIRE(cond, code, ...)       \
    do {                                    \
        if (!(cond)) {                      \
            ::ldot::set_error(__VA_ARGS__); \
            return (code);                  \
        }                                   \
    } while (0)

}  // namespace ldot
