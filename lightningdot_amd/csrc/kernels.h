// Host-side launchers of the HIP kernels (internal to libldot.so).
#pragma once
#include "ldot_common.h"

namespace ldot {

constexpr int kBM = 256;           // queries are padded to a multiple of this (the fused kernel's query tile)
constexpr int kBN = 256;           // dense chunks cover a multiple of this many index rows
constexpr int kBK = 64;            // the feature dimension is padded to a multiple of this
constexpr int kMaxK = 2048;        // largest k of a search
constexpr int kMaxKp = 3072;       // largest candidate-list length k' = k + margin: k' + 1024 keys fit a 32 KiB LDS buffer
constexpr float kVerifyC = 4.0f;   // c of the statistical bf16 score-error bound E = c * 2^-8 * |q| * max|x| / sqrt(d) (LDOT_OPT_VERIFY, LDOT_OPT_RESULT_SET)
constexpr int kSelThreads = 256;

// fused-filter candidate pools: per query, nsubs = kPoolSubsPerSlice * (row slices) sub-pools of kPoolCap RECORDS.  A record is what
// one lane holds when its 8-score test fires: the 4 + 4 scores of two vertically adjacent 16x16 MFMA tiles (rows rb + {0,1,2,3} and
// rb + kPoolRecHiRow + {0,1,2,3}) and rb — three 16-byte planes {s0..s3}, {s4..s7}, {rb,-,-,-}, laid out entry-major and plane-major:
// pool[((q * kPoolCap + e) * 3 + plane) * nsubs + sub].  The filter does not localise the hit (no per-score compares, no nested
// branches on the matrix pipe's critical path: three stores straight from the accumulator registers); the pool select applies the
// threshold to the 8 scores of every record.
// A sub-pool (id = slice * 4 + wm: one per row slice and wave row) is shared by the four lanes of a query column — the four row
// quads (lane >> 4) of a 16-row MFMA block — WITHOUT any coordination between them: lane group g owns the entries
// [g * kPoolGroupCap, (g + 1) * kPoolGroupCap) and byte g of the sub-pool's 32-bit counter (its own count, saturating at 255).  The
// select reads ONE counter word per sub-pool (128 per query at 32 row slices; round 3's 256 lane-private sub-pools cost it 109 us per
// launch against 86 us with 128) and maps the records of a sub-pool to their entries through its four byte counts (pool_entry_of).
// (Cursors shared by the four lanes — consecutive slots from a ballot — were built first: the rare path grew from ~12 to ~33
// instructions and the admissions from 0.21 to 0.45 ms per pass, profiles/r04_ab_t16b.txt.)
constexpr int kPoolGroups = 4;
constexpr int kPoolGroupCap = 8;
constexpr int kPoolCap = kPoolGroups * kPoolGroupCap;   // entries of a sub-pool
constexpr int kPoolFill = 4;           // launches are sized for <= this many expected records per sub-pool (1 per lane group of capacity 8:
                                       // P(Poisson(1) > 8) ~ 1e-6 per cell and launch where the bound is the active one)
constexpr int kPoolPlanes = 3;
constexpr int kPoolRecBytes = 16 * kPoolPlanes;
constexpr int kPoolSubsPerSlice = 4;   // 4 wave rows (round 3: 2 wave rows x 4 lane-private groups)
constexpr int kPoolRecHiRow = 16;      // row offset of a record's second quad (the 16-row MFMA tile below)
constexpr int kPoolSubsMax = 1024;     // query-group width 1: 256 row slices x 4

// rows: convert n rows of `dtype` (row stride ld_src elements, d valid columns) into the padded fp32 master copy
// and/or its bf16 shadow (row stride dpad, zero padded); optional L2 normalisation.  Rows [n, n_pad) are zero-filled.
// split != 0: split-bf16 shadow of 3 * dpad per row (1: index side [hi|hi|lo], 2: query side [hi|lo|hi]).
// dst16b (optional, ARRAY base; row0b = array row of this launch's first row): the same shadow in the blocked layout of
// the fused kernel (1 KiB blocks of 16 rows x 32 k).
// perm (LDOT_OPT_ROW_SHUFFLE): n > 0 -> destination row j takes source row (mul * j + add) mod n (mul coprime to n)
struct RowPerm {
    int64_t mul = 0, add = 0, n = 0;
};
int launch_convert_rows(const void* src, int dtype, int64_t ld_src, int64_t n, int64_t n_pad, int d, int dpad,
                        int normalize, float* dst32, uint16_t* dst16, int split, uint16_t* dst16b, int64_t row0b,
                        hipStream_t st, RowPerm perm = RowPerm());
// label tables of a shuffled index: label[p] = external label of stored row p, pos[l] = stored row of label l.  _perm_labels: the rows
// [base, base + n) of one add; _reshuffle_tables: all stored rows re-ordered (new row j = old row idx[j]; old_label NULL = identity)
int launch_perm_labels(int32_t* label, int32_t* pos, int64_t base, int64_t n, RowPerm perm, hipStream_t st);
int launch_reshuffle_tables(const int32_t* old_label, int32_t* new_label, int32_t* new_pos, int32_t* idx, int64_t n, RowPerm perm, hipStream_t st);

// recovery of overflowed queries: dst row i = src row idx[i] (rows [n, n_pad) zero); their thresholds (lowered by a hair); and the
// way back for finished lists: list / threshold of compact query i -> query idx[i]
int launch_gather_rows_f32(const float* src, int64_t ld, const int32_t* idx, int64_t n, int64_t n_pad, float* dst, hipStream_t st);
int launch_gather_tau(const float* tau, const int32_t* idx, int64_t n, const float* q32, int64_t ldq, int d, const float* xnorm_max,
                      float* dst, hipStream_t st);
int launch_scatter_lists(const float* cs, const int32_t* ci, const float* ctau, const int32_t* idx, int64_t n, int kp, float* ls,
                         int32_t* li, float* tau, hipStream_t st);

// q16 / x16 are the BLOCKED shadows (launch_convert_rows dst16b), nq_pad a multiple of 256, xrow0 a multiple of 16
int launch_score_dense(const void* q16, int64_t ldq_elems, int64_t nq_pad, const void* x16, int64_t ldx_elems,
                       int64_t xrow0, int64_t nrows_pad, int dpad, float* S, int64_t lds_elems, int64_t nq_valid,
                       hipStream_t st, int64_t tile_stride = 256,   // (tile_stride: rows between the starts of consecutive 256-row tiles)
                       hipEvent_t ev_a = nullptr, hipEvent_t ev_b = nullptr);   // (start / stop events recorded around the launch: LDOT_OPT_PROFILE)

// <= 64 queries (serving): wave-per-16-row-group scan straight from global memory (score_narrow.hip)
constexpr int kNarrowMaxQueries = 64;    // 1, 2 or 4 groups of 16 queries per scan
constexpr int kNarrowMaxLdsKiB = 128;    // query operand in LDS: slabs x query groups KiB
constexpr int kNarrowMaxRuns = 16384;   // run maxima per query and launch (select_narrow.hip keeps them in one workgroup's registers)
constexpr int kNarrowCandCap = 16384;   // candidate keys per query (128 KiB of LDS in the final sort, sized by the actual count)
constexpr int kNarrowCntStride = 64;    // ints between the candidate counters of two queries (atomics on one cache line serialise)
// M (optional, zero on entry) [nq][ldm]: ascending keys (~desc_key) of the per-run maxima over the valid rows, run = 16 << run_shift rows
int launch_score_narrow(const void* q16b, const void* x16b, int64_t ld_elems, int64_t xrow0, int64_t nrows, float* S,
                        int64_t lds_elems, int nq, uint32_t* M, int64_t ldm, int run_shift, int tiled, hipStream_t st,
                        const float* qf32 = nullptr, int64_t ldqf = 0, int d = 0);
// selection from the run maxima (select_narrow.hip): threshold key per query, candidate collection (leaves M zero again), final
// sorted lists.  cnt [nq * kNarrowCntStride] must be zero before the first collect of a search (the final kernel leaves it zero); over[q] = 1 marks a
// query whose candidate buffer was full (its list is unusable).
int launch_narrow_tau(const uint32_t* M, int64_t ldm, int nruns, int nq, int kp, uint32_t* tau_key, hipStream_t st);
// nrows_q (optional): per-query row counts nrows_q[q * nrows_q_stride] instead of nrows; tiled_qg: 0 = S is row-major [q][lds_elems],
// else the number of query groups of the tiled layout launch_score_narrow(tiled = 1) writes
int launch_narrow_collect(const float* S, int64_t lds_elems, uint32_t* M, int64_t ldm, int nruns, int run_rows, int64_t nrows,
                          int64_t row0, int nq, const uint32_t* tau_key, uint64_t* cand, int cap, int32_t* cnt, const int32_t* nrows_q,
                          int64_t nrows_q_stride, int tiled_qg, hipStream_t st);
int launch_narrow_final(const uint64_t* cand, int cap, int32_t* cnt, int nq, float* list_s, int32_t* list_i, int kp, float* tau,
                        int32_t* over, hipStream_t st);

// <= 16 queries, ONE scan chunk, nruns <= 2048, kp <= 512: threshold + collect + top-k' + exact re-score + final order + output in one
// launch (one 1024-thread workgroup per query).  list_* / tau_out and out_* are optional; over[q] = 1: too many candidates (redo).
// nrows_q: per-query column counts; rowbase / cstart / nprobe: column -> row translation of the inverted-file scan (all optional).
int launch_narrow_finish(const float* S, int tiled_qg, int64_t lds_elems, uint32_t* M, int64_t ldm, int nruns, int run_rows,
                         int64_t nrows, int nq, const float* q32, int64_t ldq, const float* x32, int64_t ldx, int dpad, int kp, int k,
                         int do_rescore, float* list_s, int32_t* list_i, float* tau_out, float* out_s, int64_t* out_l, int32_t* over,
                         const int32_t* nrows_q, int64_t nrows_q_stride, const int64_t* rowbase, const int32_t* cstart, int nprobe,
                         hipStream_t st);

// dst [n][d + 2] fp32 = [q (L2-normalised if asked), 0, 1]: the coarse query of the inverted-file search (convert.hip)
int launch_augment_queries(const void* src, int dtype, int d, int64_t n, int normalize, float* dst, int ld_dst, hipStream_t st);
// inverted-file scan (ivf.hip): per-query compact column space over the probed lists
// plist [nq][nprobe] validated list ids, rowbase [nq][nprobe] first row of each list, cstart [nq][nprobe + 1] exclusive prefix sums of
// the list lengths (last = column count)
int launch_ivf_prefix(const void* probes, int probes_are_int64, int64_t nq, int nprobe, int nlist, const int64_t* list_offsets,
                      int32_t* plist, int64_t* rowbase, int32_t* cstart, hipStream_t st);
// S[q][col] exact fp32 scores, M[q][col / (16 << run_shift)] run maxima (ascending keys, atomicMax into zeros)
int launch_ivf_scan(const float* q32, int64_t ldq, const float* x32, int64_t ldx, int dpad, int64_t nq, const int64_t* rowbase,
                    const int32_t* cstart, int nprobe, int64_t max_cols, int run_shift, float* S, int64_t lds_elems, uint32_t* M,
                    int64_t ldm, hipStream_t st);
// the same scan from the blocked bf16 shadows (half the bytes; S = bf16-input scores, the candidates are re-scored exactly afterwards)
int launch_ivf_scan_bf16(const void* q16b, const void* x16b, int64_t ld_elems, int64_t nq, const int64_t* rowbase, const int32_t* cstart,
                         int nprobe, int64_t max_cols, int run_shift, float* S, int64_t lds_elems, uint32_t* M, int64_t ldm,
                         hipStream_t st);
int launch_ivf_final(const uint64_t* cand, int cap, int32_t* cnt, int64_t nq, const int64_t* rowbase, const int32_t* cstart,
                     int nprobe, int k, float* out_s, int64_t* out_l, int32_t* over, hipStream_t st);

// lists: [nq][kp] fp32 scores + int32 rows, kept sorted (score desc, row asc); empty slots have row -1.
// also initialises the admission thresholds when tau != nullptr: -inf for queries < nq, +inf for the pad queries
int launch_init_lists(float* list_s, int32_t* list_i, int64_t n, float* tau, int64_t nq, int64_t nq_pad, hipStream_t st);
int launch_select_dense(const float* S, int64_t lds_elems, int64_t nq, int64_t ncols, int64_t idx_base,
                        float* list_s, int32_t* list_i, int kp, float* tau, hipStream_t st);
// long lists (k' > 512; select_big.hip): a 256-thread workgroup per query, sample / list pivot + one streaming pass + bit search in registers
bool select_big_dense_ok(int kp, int64_t ncols, int64_t idx_base);
int launch_select_big_dense(const float* S, int64_t lds_elems, int64_t nq, int64_t ncols, int64_t idx_base, float* list_s, int32_t* list_i,
                            int kp, float* tau, hipStream_t st);
int launch_select_big_pools(const uint4* pool, int32_t* pool_cnt, int nsubs, int64_t nq, int32_t row_end, float* list_s, int32_t* list_i,
                            int kp, float* tau, int32_t* overflow_flags, int32_t* over_sum, int32_t* qcnt, hipStream_t st, float* tau_opt,
                            int opt_m);
// segmented variant for few queries x many rows: grid (nq, nseg); every (query, segment) writes an independent partial
// top-kp list part_[sl][seg][q][kp] (int64 labels) that launch_select_lists then merges
int launch_select_dense_parts(const float* S, int64_t lds_elems, int64_t nq, int64_t ncols, int64_t seg_cols,
                              int64_t idx_base, int kp, float* part_s, int64_t* part_l, hipStream_t st);
// list bookkeeping for the segmented path
int launch_lists_to_parts(const float* list_s, const int32_t* list_i, int64_t n, float* part_s, int64_t* part_l,
                          hipStream_t st);
int launch_parts_to_lists(const float* out_s, const int64_t* out_l, int64_t n, float* list_s, int32_t* list_i, int kp,
                          float* tau, hipStream_t st);
// row_end: records may carry rows of the zero padding behind the last index row (>= row_end): dropped here.
// qcnt[q] (optional) += the number of records read for q.  overflow_flags[q] is set (and *over_sum incremented once per query) when one of q's sub-pools held more than kPoolCap records.
// tau_opt / opt_m (optimistic scan, large batches only): tau_opt[q] = max(tau_opt[q], opt_m-th best of the new list) for the next launch;
// opt_m = 0: the end-of-scan check instead (the list's own threshold must reach tau_opt[q], else the query is flagged like an overflow)
int launch_select_pools(const uint4* pool, const int32_t* pool_cnt, int nsubs, int64_t nq, int32_t row_end, float* list_s,
                        int32_t* list_i, int kp, float* tau, int32_t* overflow_flags, int32_t* over_sum, int32_t* qcnt,
                        hipStream_t st, float* tau_opt = nullptr, int opt_m = 0);
// few queries x many sub-pools: G waves per query write partial top-kp lists part_[sl][g][q][kp] (merged by launch_select_lists
// together with the running list); kp + 512 <= 1024
int launch_select_pools_parts(const uint4* pool, const int32_t* pool_cnt, int nsubs, int64_t nq, int G, int32_t row_end, int kp,
                              const float* tau, float* part_s, int64_t* part_l, int32_t* overflow_flags, int32_t* over_sum,
                              int32_t* qcnt, hipStream_t st);
// parts [nparts][nq][kp] + the running list -> the new running list (+ tau), in place
// sharded search: warm-up statistics (stat[q] = threshold, stat[nq + q] = -(m-th best)), their neutral element, and the agreed floor
int launch_list_stats(const float* list_s, const int32_t* list_i, int kp, int64_t nq, int m, const float* tau, float* stat, hipStream_t st,
                      int slots = 2, const float* level = nullptr, const int32_t* redone = nullptr);
// end-of-scan statistics of all ranks (3 x nq, all-reduced with MAX) -> floor[q] + this shard's list entries at or above the largest level
int launch_shard_floor(const float* stat, int64_t nq, const float* list_s, const int32_t* list_i, int kp, float* floor_out,
                       int32_t* count_out, hipStream_t st);
int launch_neutral_stats(int64_t nq, float* stat, hipStream_t st);
int launch_apply_stats(int64_t nq, const float* stat, float* tau, hipStream_t st);
// optimistic thresholds of the fused scan: tau_opt = max(tau_opt, m-th best of the list); initial values; end-of-scan verification
int launch_tau_opt(const float* list_s, const int32_t* list_i, int kp, int64_t nq, int m, const float* tau, float* tau_opt, hipStream_t st);
int launch_init_fused_scan(float* tau_opt, int32_t* qcnt, int32_t* over_sum, int64_t nq, int64_t nq_pad, hipStream_t st);
int launch_verify_tau_opt(const float* tau, const float* tau_opt, int64_t nq, int32_t* overflow, int32_t* over_sum, hipStream_t st);
int launch_merge_parts_into_lists(const float* part_s, const int64_t* part_l, int nparts, int64_t nq, int kp, float* list_s,
                                  int32_t* list_i, float* tau, hipStream_t st);
// generic merge of explicit candidate lists: cand_[sl] is [nq][ncand] (labels int64, -1 = empty) -> [nq][k_out]
int launch_select_lists(const float* cand_s, const int64_t* cand_l, int64_t part_stride, int64_t part_stride_l, int nparts, int k_in,
                        int64_t nq, int k_out, float* out_s, int64_t* out_l, hipStream_t st);

// exact fp32 re-score of the kp candidates of every query, final ordering, top-k output.
// where the re-score kernel puts a query's final list: dense [nq][k] arrays (block_rows = 0), or blocks of block_rows queries
// stride_s floats / stride_l int64 apart (the per-destination blocks of a sharded search's send buffer); label_base is added to every valid label
struct RescoreOut {
    int64_t block_rows, stride_s, stride_l, label_base;
    const int32_t* label_map;   // (LDOT_OPT_ROW_SHUFFLE) label of stored row r = label_map[r]; NULL: the row number itself
};
int launch_rescore(const float* q32, int64_t ldq, const float* x32, int64_t ldx, int dpad, int64_t nq,
                   const float* list_s, const int32_t* list_i, int kp, int k, int do_rescore, const float* floor,
                   float* out_s, int64_t* out_l, hipStream_t st,
                   const RescoreOut* layout = nullptr, const int32_t* label_map = nullptr);
// LDOT_OPT_RESULT_SET (rescore.hip): the top-k SET — only the candidates within 2E of the k-th candidate score are re-scored exactly
// (E = band_c * 2^-8 * |q| * max_norm / sqrt(d)); stats (optional, device, 2 * kSetStatSlots counters): [2 s] += candidates gathered,
// [2 s + 1] += live candidates, slot s = workgroup mod kSetStatSlots
constexpr int kSetStatSlots = 64;
int launch_rescore_set(const float* q32, int64_t ldq, const float* x32, int64_t ldx, int dpad, int d, int64_t nq, const float* list_s,
                       const int32_t* list_i, int kp, int k, const float* max_norm, float band_c, float* out_s, int64_t* out_l,
                       const int32_t* label_map, unsigned long long* stats, hipStream_t st);

// running maximum of the L2 norms of the rows of a padded fp32 matrix (atomicMax into *out_max, a non-negative float)
int launch_row_norm_max(const float* x32, int64_t ld, int64_t n, int d, float* out_max, hipStream_t st);
// LDOT_OPT_VERIFY: flags[q] = 1 when the top-k of query q is not proven exact (see rescore.hip); *count += number flagged
int launch_verify_exact(const float* q32, int64_t ldq, int d, int64_t nq, const float* out_s, const int64_t* out_l, int k,
                        const float* tau, const float* max_norm, int32_t* flags, int32_t* count, hipStream_t st);

// IVF list search: S[q][probe * lpad + o] = exact fp32 score of query q against row list_offsets[probes[q][probe]] + o (padding score
// past the list's end); lpad a multiple of 64.  translate: column labels of a selection over S -> index rows (-1 for padding)
int launch_scan_lists(const float* q32, int64_t ldq, const float* x32, int64_t ldx, int dpad, int64_t nq,
                      const int64_t* list_offsets, const int32_t* probes, int nprobe, int nlist, int lpad, float* S,
                      int64_t lds_elems, hipStream_t st);
int launch_translate_cols(int64_t* labels, int64_t nq, int k, const int64_t* list_offsets, const int32_t* probes, int nprobe,
                          int nlist, int lpad, hipStream_t st);

int fused_tile_rows();
int fused_query_group(int64_t nq_pad);   // 8 / 4 / 2 / 1 -> 1024 / qg sub-pools per query, 256 / qg row slices

// the order in which a fused launch visits its row tiles: tile t sits at physical tile ((base + t) * mul) mod `mod` of the panel
// (da = nslices * mul mod `mod`, dn = ntiles * mul mod `mod`: the cursors' increments); sequential = {1, 0, 2^30, nslices, ntiles}
struct ScanOrder {
    int mul, base, mod, da, dn;
};
int scan_order_multiplier(int64_t mod);

// fused MFMA score + threshold filter over index rows [row0, row0 + nrows) (row0 multiple of 16); x16 / q16 are the BLOCKED
// shadows (launch_convert_rows dst16b).  scramble_tiles > 0 (= the 384-row tiles of the WHOLE index, row0 = 0): the launch covers the
// tiles [scramble_base, scramble_base + ceil(nrows / 384)) of a fixed pseudo-random order of all tiles instead of a contiguous range
int launch_score_filter(const void* x16, int64_t ldx_elems, int64_t row0, int64_t nrows, const void* q16,
                        int64_t ldq_elems, int64_t nq_pad, int dpad, const float* tau, uint4* pool, int32_t* pool_cnt,
                        hipStream_t st, int64_t scramble_tiles = 0, int64_t scramble_base = 0, hipEvent_t ev_a = nullptr,
                        hipEvent_t ev_b = nullptr,   // (ev_a / ev_b: start / stop events recorded around the launch: LDOT_OPT_PROFILE)
                        void* cur_save = nullptr);   // fused_cursor_save_bytes(nq_pad) bytes: lets a long sequential scan run as row CHUNKS (score_filter.hip)
size_t fused_cursor_save_bytes(int64_t nq_pad);

// loss path (fp32-input MFMA)
int launch_sgemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* B2, float w, float* C,
                    int64_t ldc, int64_t M, int64_t N, int64_t K, hipStream_t st);
int launch_sgemm_nn(const float* A, int64_t lda, const float* B, int64_t ldb, float alpha, float* C, int64_t ldc,
                    int64_t M, int64_t N, int64_t K, int accumulate, hipStream_t st);
int launch_sgemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float alpha, float* C, int64_t ldc,
                    int64_t M, int64_t N, int64_t K, int accumulate, hipStream_t st);
int launch_nll_rows(const float* scores, int64_t n1, int64_t n2, const int32_t* pos, float* row_loss, float* lse,
                    int32_t* correct, float* loss_sum, hipStream_t st);
int launch_nll_dscores(const float* scores, const float* lse, const int32_t* pos, const float* g_row,
                       const float* g_scores, int64_t n1, int64_t n2, float* ds, hipStream_t st);

}  // namespace ldot
