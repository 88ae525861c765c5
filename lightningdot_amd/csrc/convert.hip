// Row ingest: caller rows (fp32 / bf16 / fp16, arbitrary row stride) -> padded fp32 master copy + bf16 shadow,
// optional L2 normalisation.  One wave per row.  Serves index add (dvl/indexer/faiss_indexers.py:77
// IndexFlatIP.add), query ingest for search (:83) and [CLS] pooling (dvl/models/bi_encoder.py:120,188:
// pooled = sequence_output[:, 0, :] is a strided row gather, ld_src = L*D).
#include <algorithm>

#include "kernels.h"

namespace ldot {

template <int DT>
__device__ inline float load_elem(const void* p, int64_t i) {
    if (DT == LDOT_F32) return ((const float*)p)[i];
    if (DT == LDOT_BF16) return bf16_bits_to_f32(((const uint16_t*)p)[i]);
    return f16_bits_to_f32(((const uint16_t*)p)[i]);
}

template <int DT>
__global__ __launch_bounds__(256) void convert_rows_kernel(const void* __restrict__ src, int64_t ld_src, int64_t n,
                                                           int64_t n_pad, int d, int dpad, int normalize,
                                                           float* __restrict__ dst32,
                                                           uint16_t* __restrict__ dst16, int split,
                                                           uint16_t* __restrict__ dst16b, int64_t row0b, RowPerm perm) {
    // dst16b (optional): the same shadow row in the BLOCKED layout the fused kernel streams — 16 rows x 32 k (1 KiB) blocks,
    // block (g, s) at ((g * nslab + s) * 512 elements, row r % 16 at +32 * (r % 16): one wave-level direct-to-LDS load
    // instruction of the ring engine then reads ONE contiguous KiB (full 128-B lines) instead of 16 half lines.
    // dst16b is the ARRAY base and row0b the array row of this launch's row 0 (blocks straddle launches).
    // split = 0: dst16 row = [bf16(v)] (dpad);  split-bf16 operands (3 * dpad per row, v ~ hi + lo to 16 mantissa bits):
    // split = 1 (index side): [hi | hi | lo],  split = 2 (query side): [hi | lo | hi]  so that one K = 3*dpad contraction
    // yields hi.hi + hi.lo + lo.hi
    const int64_t ld16 = split ? 3 * (int64_t)dpad : dpad;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_pad) return;
    if (row >= n) {   // pad rows of the last query tile: zeros
        for (int c = lane; c < dpad; c += 64)
            if (dst32) dst32[row * dpad + c] = 0.f;
        if (dst16)
            for (int64_t c = lane; c < ld16; c += 64) dst16[row * ld16 + c] = 0;
        if (dst16b) {
            const int64_t rb = row0b + row, nslab = ld16 / 32;
            for (int64_t c = lane; c < ld16; c += 64) dst16b[((rb >> 4) * nslab + (c >> 5)) * 512 + (rb & 15) * 32 + (c & 31)] = 0;
        }
        return;
    }
    // (LDOT_OPT_ROW_SHUFFLE: destination row j of an add takes source row (mul * j + add) mod n)
    const int64_t so = (perm.n > 0 ? (row * perm.mul + perm.add) % perm.n : row) * ld_src;
    float scale = 1.f;
    if (normalize) {
        // fp64 accumulation of the squared norm: matches x / max(||x||, eps) of the oracle to fp32 rounding
        double ss = 0.0;
        for (int c = lane; c < d; c += 64) {
            const double v = (double)load_elem<DT>(src, so + c);
            ss += v * v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const double nrm = sqrt(ss);
        scale = (float)(1.0 / (nrm > 1e-12 ? nrm : 1e-12));
    }
    if (DT == LDOT_F32 && !split && (d & 3) == 0 && (ld_src & 3) == 0 && ((uintptr_t)src & 15) == 0) {
        // fp32 rows with 16-byte aligned quads (the common case: query ingest, index add from fp32): four columns per lane and trip —
        // the same values and the same rounding as the scalar loop below, a quarter of the memory instructions
        const int64_t rb = row0b + row, nslab = ld16 / 32;
        for (int c = lane * 4; c < dpad; c += 256) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < d) {
                v = *(const float4*)((const float*)src + so + c);
                if (normalize) v = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
            }
            if (dst32) *(float4*)(dst32 + row * dpad + c) = v;
            const uint2 h = make_uint2((uint32_t)f32_to_bf16_bits(v.x) | ((uint32_t)f32_to_bf16_bits(v.y) << 16),
                                       (uint32_t)f32_to_bf16_bits(v.z) | ((uint32_t)f32_to_bf16_bits(v.w) << 16));
            if (dst16) *(uint2*)(dst16 + row * ld16 + c) = h;
            if (dst16b) *(uint2*)(dst16b + ((rb >> 4) * nslab + (c >> 5)) * 512 + (rb & 15) * 32 + (c & 31)) = h;
        }
        return;
    }
    for (int c = lane; c < dpad; c += 64) {
        float v = 0.f;
        if (c < d) {
            v = load_elem<DT>(src, so + c);
            if (normalize) v = v * scale;
        }
        if (dst32) dst32[row * dpad + c] = v;
        if (dst16 || dst16b) {
            const uint16_t hi = f32_to_bf16_bits(v);
            const uint16_t lo = split ? f32_to_bf16_bits(v - bf16_bits_to_f32(hi)) : 0;   // exact difference, then rounded
            const int nseg = split ? 3 : 1;
            const int64_t rb = row0b + row, nslab = ld16 / 32;
            for (int sgm = 0; sgm < nseg; ++sgm) {
                const uint16_t val = (sgm == 0) ? hi : ((sgm == 1) == (split == 1) ? hi : lo);
                const int64_t cc = (int64_t)sgm * dpad + c;
                if (dst16) dst16[row * ld16 + cc] = val;
                if (dst16b) dst16b[((rb >> 4) * nslab + (cc >> 5)) * 512 + (rb & 15) * 32 + (cc & 31)] = val;
            }
        }
    }
}

// largest L2 norm among rows [0, n) of the padded fp32 master copy: a wave walks a strided set of rows and publishes ONE
// atomicMax (on the non-negative float's bits) — one atomic per row serialised a 1M-row add on a single address
__global__ __launch_bounds__(256) void row_norm_max_kernel(const float* __restrict__ x32, int64_t ld, int64_t n, int d,
                                                           float* __restrict__ out_max) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    float best = 0.f;
    for (int64_t row = wave; row < n; row += nwaves) {
        const float* r = x32 + row * ld;
        float acc = 0.f;
        for (int c = lane; c < d; c += 64) acc = fmaf(r[c], r[c], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        best = fmaxf(best, acc);
    }
    if (lane == 0 && best > 0.f) atomicMax((int*)out_max, __float_as_int(sqrtf(best)));
}

int launch_row_norm_max(const float* x32, int64_t ld, int64_t n, int d, float* out_max, hipStream_t st) {
    if (n <= 0) return LDOT_OK;
    const int64_t blocks = std::min<int64_t>((n + 3) / 4, 2048);
    hipLaunchKernelGGL(row_norm_max_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x32, ld, n, d, out_max);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// coarse query of the inverted-file search: [q (optionally L2-normalised like the row search does), 0, 1] in fp32 — the inner product
// with a centroid row [c~, -|c~|^2 / 2] (c~ = [c, sqrt(phi - |c|^2)], dvl/indexer/faiss_indexers.py:123-131) orders the lists by L2
// distance in the augmented space.  One wave per query.
template <int DT>
__global__ __launch_bounds__(256) void augment_queries_kernel(const void* __restrict__ src, int d, int64_t n, int normalize,
                                                              float* __restrict__ dst, int ld_dst) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float scale = 1.f;
    if (normalize) {
        double ss = 0.0;
        for (int c = lane; c < d; c += 64) {
            const double v = (double)load_elem<DT>(src, row * d + c);
            ss += v * v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const double nrm = sqrt(ss);
        scale = (float)(1.0 / (nrm > 1e-12 ? nrm : 1e-12));
    }
    float* out = dst + row * (int64_t)ld_dst;
    for (int c = lane; c < d; c += 64) {
        const float v = load_elem<DT>(src, row * d + c);
        out[c] = normalize ? v * scale : v;
    }
    if (lane == 0) {
        out[d] = 0.f;
        out[d + 1] = 1.f;
    }
    for (int c = d + 2 + lane; c < ld_dst; c += 64) out[c] = 0.f;   // (rows padded to the coarse index's row stride: see ldot_ivf_search)
}

// dst rows: [q, 0, 1, 0 ...] with row stride ld_dst >= d + 2
int launch_augment_queries(const void* src, int dtype, int d, int64_t n, int normalize, float* dst, int ld_dst, hipStream_t st) {
    if (n <= 0) return LDOT_OK;
    LDOT_REQUIRE(ld_dst >= d + 2, LDOT_EINVAL, "augmented row stride too short");
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    switch (dtype) {
        case LDOT_F32:
            hipLaunchKernelGGL(augment_queries_kernel<LDOT_F32>, grid, block, 0, st, src, d, n, normalize, dst, ld_dst);
            break;
        case LDOT_BF16:
            hipLaunchKernelGGL(augment_queries_kernel<LDOT_BF16>, grid, block, 0, st, src, d, n, normalize, dst, ld_dst);
            break;
        case LDOT_F16:
            hipLaunchKernelGGL(augment_queries_kernel<LDOT_F16>, grid, block, 0, st, src, d, n, normalize, dst, ld_dst);
            break;
        default:
            set_error("unsupported dtype %d", dtype);
            return LDOT_EINVAL;
    }
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// ---- LDOT_OPT_ROW_SHUFFLE: label tables of an index whose rows are stored in a pseudo-random order -------------------------------------
// label[p] = external label (insertion number) of stored row p, pos[l] = stored row of label l
__global__ __launch_bounds__(256) void perm_labels_kernel(int32_t* __restrict__ label, int32_t* __restrict__ pos, int64_t base, int64_t n,
                                                          RowPerm perm) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int64_t sgm = perm.n > 0 ? (j * perm.mul + perm.add) % perm.n : j;   // stored row base + j holds the add's row sgm
    label[base + j] = (int32_t)(base + sgm);
    pos[base + sgm] = (int32_t)(base + j);
}

// re-shuffle of the rows [0, n) already stored: new row j = old row idx[j] = (mul * j + add) mod n, labels follow
__global__ __launch_bounds__(256) void reshuffle_tables_kernel(const int32_t* __restrict__ old_label, int32_t* __restrict__ new_label,
                                                               int32_t* __restrict__ new_pos, int32_t* __restrict__ idx, int64_t n, RowPerm perm) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int64_t o = (j * perm.mul + perm.add) % perm.n;
    const int32_t l = old_label ? old_label[o] : (int32_t)o;
    idx[j] = (int32_t)o;
    new_label[j] = l;
    new_pos[l] = (int32_t)j;
}

int launch_perm_labels(int32_t* label, int32_t* pos, int64_t base, int64_t n, RowPerm perm, hipStream_t st) {
    if (n <= 0) return LDOT_OK;
    hipLaunchKernelGGL(perm_labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, label, pos, base, n, perm);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_reshuffle_tables(const int32_t* old_label, int32_t* new_label, int32_t* new_pos, int32_t* idx, int64_t n, RowPerm perm, hipStream_t st) {
    if (n <= 0) return LDOT_OK;
    hipLaunchKernelGGL(reshuffle_tables_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, old_label, new_label, new_pos, idx, n, perm);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

// ---- recovery of overflowed queries (api.hip: redo_flagged): compact copies of the flagged queries' rows / thresholds, and the
// way back for their finished lists ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const float* __restrict__ src, int64_t ld, const int32_t* __restrict__ idx,
                                                              int64_t n, int64_t n_pad, float* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_pad) return;
    const float* s = row < n ? src + (int64_t)idx[row] * ld : nullptr;
    for (int64_t c = lane; c < ld; c += 64) dst[row * ld + c] = s ? s[c] : 0.f;
}

// thresholds of the flagged queries, lowered by a hair: the threshold is the k'-th best score the FIRST pass saw, partly computed by
// another kernel (the dense warm-up) whose fp32 summation order may differ in the last bits from the filter kernel's.  That difference
// scales with sum |q_i x_i| <= |q| max|x|, not with the threshold itself (which may be near 0 for centred data), so the slack is
// 1e-5 |q| max|x| (max|x| = the index's running row-norm maximum), and never less than 1e-5 |t|.
__global__ __launch_bounds__(256) void gather_tau_kernel(const float* __restrict__ tau, const int32_t* __restrict__ idx, int64_t n,
                                                         const float* __restrict__ q32, int64_t ldq, int d,
                                                         const float* __restrict__ xnorm_max, float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float t = tau[idx[i]];
    float slack = fabsf(t) * 1e-5f;
    if (q32 && xnorm_max) {
        const float* q = q32 + (int64_t)idx[i] * ldq;
        float s = 0.f;
        for (int c = 0; c < d; ++c) s = fmaf(q[c], q[c], s);
        slack = fmaxf(slack, 1e-5f * sqrtf(s) * xnorm_max[0]);
    }
    dst[i] = t - slack - 1e-30f;
}

__global__ __launch_bounds__(256) void scatter_lists_kernel(const float* __restrict__ cs, const int32_t* __restrict__ ci,
                                                            const float* __restrict__ ctau, const int32_t* __restrict__ idx,
                                                            int64_t n, int kp, float* __restrict__ ls, int32_t* __restrict__ li,
                                                            float* __restrict__ tau) {
    const int64_t q = blockIdx.x;
    if (q >= n) return;
    const int64_t dst = idx[q];
    for (int c = threadIdx.x; c < kp; c += 256) {
        ls[dst * kp + c] = cs[q * kp + c];
        li[dst * kp + c] = ci[q * kp + c];
    }
    if (threadIdx.x == 0) tau[dst] = ctau[q];
}

int launch_gather_rows_f32(const float* src, int64_t ld, const int32_t* idx, int64_t n, int64_t n_pad, float* dst, hipStream_t st) {
    if (n_pad <= 0) return LDOT_OK;
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3((unsigned)((n_pad + 3) / 4)), dim3(256), 0, st, src, ld, idx, n, n_pad, dst);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_gather_tau(const float* tau, const int32_t* idx, int64_t n, const float* q32, int64_t ldq, int d, const float* xnorm_max,
                      float* dst, hipStream_t st) {
    if (n <= 0) return LDOT_OK;
    hipLaunchKernelGGL(gather_tau_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tau, idx, n, q32, ldq, d, xnorm_max, dst);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_scatter_lists(const float* cs, const int32_t* ci, const float* ctau, const int32_t* idx, int64_t n, int kp, float* ls,
                         int32_t* li, float* tau, hipStream_t st) {
    if (n <= 0) return LDOT_OK;
    hipLaunchKernelGGL(scatter_lists_kernel, dim3((unsigned)n), dim3(256), 0, st, cs, ci, ctau, idx, n, kp, ls, li, tau);
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

int launch_convert_rows(const void* src, int dtype, int64_t ld_src, int64_t n, int64_t n_pad, int d, int dpad,
                        int normalize, float* dst32, uint16_t* dst16, int split, uint16_t* dst16b, int64_t row0b,
                        hipStream_t st, RowPerm perm) {
    if (n_pad < n) n_pad = n;
    if (n_pad <= 0) return LDOT_OK;
    const dim3 grid((unsigned)((n_pad + 3) / 4)), block(256);
    switch (dtype) {
        case LDOT_F32:
            hipLaunchKernelGGL(convert_rows_kernel<LDOT_F32>, grid, block, 0, st, src, ld_src, n, n_pad, d, dpad, normalize,
                               dst32, dst16, split, dst16b, row0b, perm);
            break;
        case LDOT_BF16:
            hipLaunchKernelGGL(convert_rows_kernel<LDOT_BF16>, grid, block, 0, st, src, ld_src, n, n_pad, d, dpad, normalize,
                               dst32, dst16, split, dst16b, row0b, perm);
            break;
        case LDOT_F16:
            hipLaunchKernelGGL(convert_rows_kernel<LDOT_F16>, grid, block, 0, st, src, ld_src, n, n_pad, d, dpad, normalize,
                               dst32, dst16, split, dst16b, row0b, perm);
            break;
        default:
            set_error("unsupported dtype %d", dtype);
            return LDOT_EINVAL;
    }
    LDOT_HIP_CHECK(hipGetLastError());
    return LDOT_OK;
}

}  // namespace ldot
