/*
 * ldot.h — C ABI of the MI355X-native retrieval hot path (lightningdot_amd).
 *
 * This is the drop-in boundary: the entry points below are what the reference's Python layer would bind
 * (through ctypes) in place of the third-party native code it reaches today.  Every entry point cites the
 * reference interface it replaces (paths relative to the reference checkout, intersun/LightningDOT).
 *
 * Conventions
 *   - plain C, no C++/torch types; all pointers are raw addresses; `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream).
 *   - every function returns an int status: LDOT_OK (0) or a negative LDOT_E*; `ldot_last_error()` returns a
 *     thread-local human readable message for the last failure on the calling thread.  No exceptions cross
 *     the ABI.
 *   - `mem` arguments say where a caller buffer lives: LDOT_HOST (pageable or pinned host memory) or
 *     LDOT_DEVICE (HIP device memory of the current device).  Caller owns all in/out buffers; the library
 *     owns the index storage and its scratch workspaces.
 *   - the library never falls back to a CPU implementation: without a usable HIP device every compute entry
 *     point fails with LDOT_EDEVICE.
 */
#ifndef LDOT_H_
#define LDOT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDOT_ABI_VERSION 7

/* status codes */
#define LDOT_OK 0
#define LDOT_EINVAL (-1)   /* bad argument (dimension / dtype / k / NULL) */
#define LDOT_EDEVICE (-2)  /* HIP runtime error, no device, launch failure */
#define LDOT_ENOMEM (-3)   /* host or device allocation failed */
#define LDOT_EIO (-4)      /* serialize / deserialize failure */
#define LDOT_ESTATE (-5)   /* object in the wrong state (e.g. size mismatch on load) */

/* element types of caller buffers */
#define LDOT_F32 0
#define LDOT_BF16 1
#define LDOT_F16 2

/* memory spaces of caller buffers */
#define LDOT_HOST 0
#define LDOT_DEVICE 1

/* label / score used to pad result rows when fewer than k rows exist (faiss IndexFlatIP convention) */
#define LDOT_PAD_LABEL (-1)
#define LDOT_PAD_SCORE (-3.402823466e+38f)

/* search-mode flags (ldot_index_set_option(LDOT_OPT_MODE, ...)) */
#define LDOT_MODE_AUTO 0   /* chosen per search from the batch and index size: <= 64 queries the narrow search (one pass over the
                            * bf16 index at HBM speed + run-maxima selection; <= 16 queries always, 17..64 when k' fits its candidate
                            * buffer); otherwise the fused filter from 32768 rows (from 20480 rows for >= 4096 queries, from 8192 rows for >= 16384), with one query
                            * block for 65..256 queries; the dense path below that */
#define LDOT_MODE_DENSE 1  /* materialise score chunks + streaming top-k' select */
#define LDOT_MODE_FUSED 2  /* fused MFMA score + threshold filter (never materialises Q x N) */

#define LDOT_OPT_MODE 1
#define LDOT_OPT_RESCORE 2      /* 1 (default): exact fp32 re-score of the bf16 candidates; 0: report bf16-input scores */
#define LDOT_OPT_CHUNK_ROWS 3   /* index rows per scoring chunk (multiple of 256) */
#define LDOT_OPT_MARGIN 4       /* extra candidates kept beyond k before the fp32 re-score (default max(28,k/4)) */
#define LDOT_OPT_PROFILE 5      /* 1: bracket every score-kernel launch with HIP events (see ldot_index_last_profile) */
#define LDOT_OPT_WARM_ROWS 6    /* rows scored densely before the fused filter starts (default 4096, multiple of 256) */
#define LDOT_OPT_GROWTH_PCT 7   /* fused launch i covers growth% of the rows already scanned (default 150) */
#define LDOT_OPT_RESERVE_ROWS 9  /* allocate capacity for this many rows now (no growth copies while filling a large index) */
#define LDOT_OPT_PRECISION 8    /* candidate generation: 0 (default) bf16 operands; 1 split-bf16 operands (x ~ hi + lo, three
                                 * MFMA products per element: ~16 mantissa bits, 3x the MFMA work and 3x the shadow) for
                                 * data whose scores crowd closer than bf16 resolves; reported scores are fp32-exact in both */

#define LDOT_OPT_OPTIMISTIC 11  /* 1 (default): large batches of the fused scan filter with OPTIMISTIC thresholds — order statistics of the rows
                                 * seen so far that lie below the final threshold unless the row order front-loads a query's best rows — and
                                 * verify every query's list afterwards (queries that fail are searched again on guaranteed thresholds):
                                 * a third of the admitted candidates, four launches instead of six per 1M rows; results are identical.
                                 * 0: guaranteed thresholds only (the k'-th best score seen so far) */
#define LDOT_OPT_SCAN_ORDER 12  /* the order in which large batches scan the index rows.  1: storage order.  2: scrambled — the 384-row tiles of the
                                 * whole index in a fixed pseudo-random order and a warm-up on a spread sample, so that the rows seen so far
                                 * are a fair sample of the index when it is stored in long runs of similar rows (sorted by class, source,
                                 * coarse cluster: runs of thousands of rows; runs about as long as a tile stay together and keep failing
                                 * the check — shuffle such rows before adding them): what the optimistic thresholds and the candidate
                                 * pools assume; +1 % on rows stored in random order.  0 (default):
                                 * storage order until one query in a thousand of a search fails the optimistic check, scrambled from then
                                 * on.  Results are identical in every order */
#define LDOT_OPT_ROW_SHUFFLE 13 /* rows stored in a pseudo-random order behind a label table kept inside the library (labels, tie order and
                                 * ldot_index_get_rows / _save keep referring to insertion order; what DenseFlatIndexer(shuffle_seed=) does in
                                 * Python, for callers that bind ldot_index_add / _search directly).  For rows that arrive sorted in SHORT runs
                                 * of similar rows (by class, by fine cluster: runs about as long as a 384-row tile), which fail the optimistic
                                 * check in the scrambled tile order too.  1: the rows of every ldot_index_add call are stored in a
                                 * pseudo-random order ((mul j + add) mod n, mul ~ n / golden ratio).  0 (default): rows are stored as added;
                                 * when a large-batch search fails the check in the scrambled tile order for more than 1 / 64 of its queries the
                                 * library re-orders the stored rows ONCE before the next search (a transient second fp32 copy; skipped when it
                                 * does not fit) and shuffles later adds.  2: never (the inverted-file row store: lists are row ranges).
                                 * Results are identical in every order */
#define LDOT_OPT_DEFER_SYNC 14  /* 1: ldot_index_search with PINNED host outputs does not wait for its results: it returns once the last kernel is
                                 * enqueued, and the results are in the buffers when the caller has synchronised the stream — or when a later
                                 * search on the same stream has returned with the option off.  A retrieval evaluation (the two searches of
                                 * dvl/trainer.py:160-170) issues its text->image search with the option on and its image->text search with it
                                 * off: ONE wait instead of two, no idle gap between the searches.  The index, the queries and the output buffers
                                 * must stay untouched until then.  Pageable outputs and searches under LDOT_OPT_PROFILE / LDOT_OPT_VERIFY wait
                                 * as before (the host reads their events / flags).  0 (default): a search returns with its results in place */
#define LDOT_OPT_RESULT_SET 15  /* 1: searches report the top-k SET for consumers that keep the ids and drop scores and order — hard-negative mining
                                 * (dvl/hn.py:54-63 keeps r[0], strips the positives and samples at random), the largest searches of the reference
                                 * (every train caption x every train image and back, top 50 .. 1000, every epoch).  A candidate whose bf16 score
                                 * lies more than 2E above the k-th candidate score is in the exact top-k whatever its exact score is, one more than
                                 * 2E below it is not (E = the error bound of LDOT_OPT_VERIFY): only the candidates in between are gathered from the
                                 * fp32 master copy (3 KiB per row) and ordered exactly.  The k labels are those of the default search (same set,
                                 * ties at the boundary to the lower label) whenever the bound holds for the candidates; their ORDER is: the
                                 * certain ones by candidate score — their reported score is the bf16-input score —, then the boundary's winners
                                 * by exact score.  Sharded searches (a floor), LDOT_OPT_RESCORE 0 and LDOT_OPT_VERIFY ignore / are ignored by
                                 * the option.  0 (default): exact scores, descending.  See ldot_index_last_set_stats */
#define LDOT_OPT_VERIFY 10      /* 1: after every search flag the queries whose top-k cannot be vouched for: the k-th exact score is not above
                                 * the candidate threshold by E = 4 * 2^-8 * |q| * max|x| / sqrt(d), a STATISTICAL bound of the bf16
                                 * rounding error of a d-term inner product (4 standard deviations for independent rounding errors;
                                 * the worst case, 2^-8 * |q| * max|x|, is sqrt(d)/4 times larger and can be reached by sparse or
                                 * concentrated vectors, for which an unflagged query is not a proof), see ldot_index_last_unproven.
                                 * The default bf16
                                 * candidate pass is exact whenever the true top-k lies inside the bf16 top-k' (k' = k + margin);
                                 * scores that crowd closer than bf16 resolves need a larger LDOT_OPT_MARGIN or LDOT_OPT_PRECISION 1 */

typedef struct ldot_index ldot_index_t;

const char* ldot_last_error(void);
int ldot_abi_version(void);
/* number of visible HIP devices (>=0) or a negative status */
int ldot_device_count(void);

/* ---------------------------------------------------------------------------------------------------------
 * Flat inner-product index — replaces faiss.IndexFlatIP as used by DenseFlatIndexer
 *   dvl/indexer/faiss_indexers.py:67   faiss.IndexFlatIP(vector_sz)        -> ldot_index_create
 *   dvl/indexer/faiss_indexers.py:77   self.index.add(vectors)             -> ldot_index_add
 *   dvl/indexer/faiss_indexers.py:83   self.index.search(query_vectors, k) -> ldot_index_search
 *   dvl/indexer/faiss_indexers.py:52   self.index.ntotal                   -> ldot_index_ntotal
 *   dvl/indexer/faiss_indexers.py:41   faiss.write_index(index, file)      -> ldot_index_save
 *   dvl/indexer/faiss_indexers.py:51   faiss.read_index(file)              -> ldot_index_load
 * Semantics: exact (un-normalised) inner product; the k best rows per query in descending score
 * order, ties broken by the lower row label; rows beyond ntotal are padded with LDOT_PAD_LABEL /
 * LDOT_PAD_SCORE.  Candidates are generated with bf16 MFMA (fp32 accumulate) and re-scored exactly in fp32
 * from the fp32 master copy of the rows, so reported scores are fp32 inner products.
 * --------------------------------------------------------------------------------------------------------- */
int ldot_index_create(int d, ldot_index_t** out);
int ldot_index_destroy(ldot_index_t* ix);
/* Append n rows of dimension d.  `normalize` != 0 scales every row to unit L2 norm first (opt-in; the
 * reference never normalises). */
int ldot_index_add(ldot_index_t* ix, const void* rows, int64_t n, int dtype, int mem, int normalize, void* stream);
int64_t ldot_index_ntotal(const ldot_index_t* ix);
int ldot_index_dim(const ldot_index_t* ix);
/* reset and the storage-changing options (PRECISION, RESERVE_ROWS) take no stream: they synchronise the device before and after */
int ldot_index_reset(ldot_index_t* ix);
int ldot_index_set_option(ldot_index_t* ix, int option, int64_t value);
/* out_scores: [nq*k] float, out_labels: [nq*k] int64, both in `out_mem` space. */
int ldot_index_search(ldot_index_t* ix, const void* queries, int64_t nq, int dtype, int mem, int normalize,
                      int k, float* out_scores, int64_t* out_labels, int out_mem, void* stream);
/* The same search in two halves, for the row-sharded index (no reference counterpart): _begin generates the k'
 * candidates of every query on this shard and copies their thresholds (the k'-th best candidate score per query, -inf
 * while a query has fewer than k' candidates) to tau_out [nq] (device, may be NULL); the caller exchanges them
 * (all-reduce MAX over the shards: at least k' candidates score >= that maximum globally, so nothing below it can be
 * among the global k' best) and passes the result as `floor` [nq] (device, or NULL) to _finish, which re-scores only
 * the candidates at or above the floor and writes the shard's partial top-k.  ldot_index_search = _begin + _finish(NULL). */
int ldot_index_search_begin(ldot_index_t* ix, const void* queries, int64_t nq, int dtype, int mem, int normalize, int k,
                            float* tau_out, void* stream);
int ldot_index_search_finish(ldot_index_t* ix, const float* floor, float* out_scores, int64_t* out_labels, int out_mem,
                             void* stream);
/* _begin in two steps, so that the shards of a row-sharded index agree on thresholds BEFORE their candidate pass instead of after it
 * (each shard then admits ~1/parts of the candidates, and its pass is one or two kernel launches instead of four; the single
 * index.search of dvl/indexer/faiss_indexers.py:82-87 / its callers dvl/trainer.py:167,170 are what is being sharded):
 *   _warmup  ingests the queries, scores them against the first few thousand rows of THIS shard and writes two numbers per query to
 *            stat_out [2*nq] (device): stat_out[q] = the k'-th best warm-up score (-inf if there was no warm-up: small shards and
 *            small batches take one-pass paths), stat_out[nq+q] = MINUS the ceil(k'/parts)-th best (+inf if not available).
 *   the caller all-reduces stat_out with MAX over the `parts` shards.  max(stat[q], -stat[nq+q]) is then a lower bound of the GLOBAL
 *            k'-th best score: some shard has k' rows at or above the first term, every shard has ceil(k'/parts) rows at or above
 *            the second.
 *   _scan    (stat_in = the reduced array, or NULL = no exchange) runs the candidate pass from that threshold and fills tau_out like
 *            _begin does; _finish follows as before.  _begin = _warmup(parts = 1) + _scan(NULL). */
int ldot_index_search_warmup(ldot_index_t* ix, const void* queries, int64_t nq, int dtype, int mem, int normalize, int k, int parts,
                             float* stat_out, void* stream);
int ldot_index_search_scan(ldot_index_t* ix, const float* stat_in, float* tau_out, void* stream);
/* _begin for ONE SHARD of a row-sharded index of `parts` shards, with the statistics its ranks exchange afterwards (the default
 * exchange of lightningdot_amd/sharded.py; same call sites being sharded as above).  stat_out [3*nq] (device):
 *   stat_out[q]        the k'-th best candidate score of this shard (as tau_out of _begin),
 *   stat_out[nq+q]     MINUS its j-th best, j = ceil(k' x share) (+inf when the shard has fewer candidates; -inf for share = 0),
 *   stat_out[2*nq+q]   the LEVEL above which the shard's list is complete (-inf: a complete top-k' list).
 * `share` in [0, 1] is the part of the k' rows this shard vouches for; the shares of all shards must add up to at least 1 (equal
 * shards: 1/parts each; unequal ones: in proportion to their rows, 0 for shards too small to matter — a tiny shard's j-th best
 * would otherwise drag the floor far below the global k'-th best).
 * total_rows = 0: the candidate pass of _begin, level = -inf.  total_rows > 0 (the rows of ALL shards together): large batches
 * scan on order statistics taken against the whole index — the m-th best of the shard's first r rows with
 * P(Poisson(k' r / total_rows) >= m) <= 1e-7 / parts lies below the GLOBAL k'-th best w.h.p. when rows are spread over the shards
 * and stored in no order that correlates with the queries —, which admits ~1/parts of the records of the shard's own thresholds
 * in ONE launch after the warm-up; the level is then that threshold, and the shard alone cannot tell whether it was safe.
 * The caller all-reduces stat_out with MAX over the shards and hands the result to ldot_index_shard_floor (between _begin_shard and
 * _finish; stat [3*nq], floor_out [nq] float, count_out [nq] int32: DEVICE memory; *k_prime_out, host, receives k'):
 *   floor_out[q] = max(stat[q], -stat[nq+q]) <= the global k'-th best (some shard has k' rows at or above the first term, every shard
 *   its ceil(k' x share) at or above the second, k' or more together) — the second term is what makes the floor tight: the largest k'-th
 *   best of a shard still lets ~0.9 k' rows PER SHARD through, the smallest ceil(k'/parts)-th best of equal shards ~1.4 k'/parts.
 *   _finish(floor_out) on every shard + the merge of the partial lists follow as before.
 *   count_out[q] = the rows of THIS shard at or above the LARGEST level of any shard (all of them are in its list; k' when no shard
 *   reported a level).  The caller all-reduces count_out with SUM (off the critical path: only the verdict needs it): a query whose sum
 *   reaches k' has k' rows at or above every level, i.e. no shard dropped one of its global k' best; the merged lists are then exact.
 *   Queries below k' are UNPROVEN (never, with total_rows = 0): EVERY rank repeats the search with total_rows = 0 (all ranks see the
 *   same sums).  Comparing the floor with the levels instead would fail far more often than the thresholds do: the floor is a noisy
 *   lower bound of the global k'-th best, the count is exact.
 * All calls enqueue on `stream`; _begin_shard synchronises it once like _begin (candidate-pool overflows are handled by the shard
 * itself, as in _begin: the queries concerned are searched again and report a complete list). */
int ldot_index_search_begin_shard(ldot_index_t* ix, const void* queries, int64_t nq, int dtype, int mem, int normalize, int k, int parts,
                                  double share, int64_t total_rows, float* stat_out, void* stream);
int ldot_index_shard_floor(ldot_index_t* ix, const float* stat, float* floor_out, int32_t* count_out, int* k_prime_out, void* stream);
/* _finish writing the shard's partial lists straight into the send buffer of the all-to-all that follows (device memory, 16-byte
 * aligned): block b — one per destination rank, block_bytes apart — holds the lists of the queries [b*block_rows, (b+1)*block_rows):
 * scores float [block_rows][k] at byte 0, labels int64 [block_rows][k] at byte LDOT_BLOCK_LABELS_OFFSET(block_rows, k); labels are
 * local rows + label_base (the shard's first global row), LDOT_PAD_LABEL stays.  ldot_merge_topk_blocked merges nparts such blocks
 * (what the all-to-all delivers: one block per source rank) for the first nq <= block_rows queries; outputs are device pointers
 * (device memory, or pinned host memory through its device mapping).  No torch op touches the lists between the two kernels. */
#define LDOT_BLOCK_LABELS_OFFSET(block_rows, k) ((((int64_t)(block_rows) * (k) * 4) + 15) / 16 * 16)
int ldot_index_search_finish_blocked(ldot_index_t* ix, const float* floor, void* out_blocks, int64_t block_rows, int64_t block_bytes,
                                     int64_t label_base, void* stream);
int ldot_merge_topk_blocked(const void* blocks, int nparts, int64_t block_rows, int64_t block_bytes, int64_t nq, int k_in, int k_out,
                            float* out_scores, int64_t* out_labels, void* stream);
/* Approximate search — stands where the reference selects faiss.IndexHNSWFlat (dvl/indexer/faiss_indexers.py:90-154, `--hnsw_index`).
 * The index rows are stored sorted by inverted list (the caller clusters them and adds them in list order): list l = rows
 * [list_offsets[l], list_offsets[l+1]).  Every query is scored EXACTLY (fp32) against the rows of its nprobe lists probes[q][0..nprobe)
 * (-1 = skip) and the k best of those are returned in descending score order with their row labels (LDOT_PAD_LABEL / LDOT_PAD_SCORE
 * when the lists hold fewer than k rows).  list_offsets [nlist+1] int64 and probes [nq*nprobe] int32 are DEVICE memory, the queries
 * are DEVICE memory of `dtype`; max_list_len = the longest list.  (The coarse step — which lists to probe — is an ordinary
 * ldot_index_search over the list centroids.)  The call synchronises `stream` once per chunk of 256 queries, also with DEVICE outputs:
 * a query whose candidate buffer filled up (thousands of equal scores) is noticed there and redone (stats[1] counts them). */
int ldot_index_search_lists(ldot_index_t* ix, const void* queries, int64_t nq, int dtype, int normalize, const int64_t* list_offsets,
                            int nlist, int64_t max_list_len, const int32_t* probes, int nprobe, int k, float* out_scores,
                            int64_t* out_labels, int out_mem, void* stream);
/* The whole approximate query in one call: coarse search (which lists to probe) + list scan.  `coarse` is an ordinary index over the
 * nlist = ntotal(coarse) list centroids in the augmented space of the reference's HNSW indexer (faiss_indexers.py:114-131) plus one
 * coordinate: row l = [c_l, sqrt(phi - |c_l|^2), -|c~_l|^2 / 2] (dimension d + 2), so that its inner product with [q, 0, 1] ranks the
 * lists by L2 distance to the query.  Queries are DEVICE memory of `dtype`, list_offsets [nlist+1] int64 DEVICE memory; the result is
 * that of ldot_index_search_lists with probes = the nprobe nearest lists (nprobe <= 2048, the largest k of a search). */
int ldot_ivf_search(ldot_index_t* ix, ldot_index_t* coarse, const void* queries, int64_t nq, int dtype, int normalize,
                    const int64_t* list_offsets, int64_t max_list_len, int nprobe, int k, float* out_scores, int64_t* out_labels,
                    int out_mem, void* stream);
/* own on-disk format ("LDOTIDX1": header + fp32 rows); bf16 shadow is rebuilt on load */
int ldot_index_save(ldot_index_t* ix, const char* path);
int ldot_index_load(const char* path, ldot_index_t** out);
/* copy the rows with labels [row0,row0+n) (insertion order, also when the store is shuffled) of the fp32 master copy to a caller buffer
 * (inspection / resharding) */
int ldot_index_get_rows(ldot_index_t* ix, int64_t row0, int64_t n, float* out, int out_mem, void* stream);
/* statistics of the last search on this index: [0]=candidate records appended by the fused filter (8 scores each), [1]=queries
 * that were searched again — a candidate pool overflowed, or the end-of-scan check of the optimistic thresholds failed
 * (LDOT_OPT_OPTIMISTIC) —, [2]=(query,row) pairs scored densely, [3]=pairs scored fused */
int ldot_index_last_stats(const ldot_index_t* ix, int64_t out[4]);
/* which regime the last search on this index ran in, and the adaptive state the handle carries (the same search costs more on an index
 * that backed off; nothing else reports it):
 *   out[0] path: 0 none, 1 narrow search (<= 64 queries), 2 dense chunks, 3 fused scan with one query block (65..256 queries), 4 fused scan
 *   out[1] thresholds of a fused scan: 1 guaranteed (k'-th best so far), 2 optimistic (verified per query), 3 pooled over a sharded index
 *   out[2] scan order: 1 storage order, 2 scrambled tile order
 *   out[3] queries searched again (pool overflow / failed optimistic check)
 *   out[4] large-batch searches this index will still run on guaranteed thresholds (back-off after failed checks; 0 = none)
 *   out[5] few-query searches that will still skip the narrow search (back-off after a full candidate buffer)
 *   out[6] 1: the index switched itself to the scrambled tile order (LDOT_OPT_SCAN_ORDER auto)
 *   out[7] rows: 0 stored as added, 1 shuffled at add time, 2 re-ordered by the library (LDOT_OPT_ROW_SHUFFLE auto) */
int ldot_index_last_regime(const ldot_index_t* ix, int64_t out[8]);
/* LDOT_OPT_RESULT_SET = 1: out[0] = candidates of the last search that were re-scored exactly, out[1] = its live candidates (what the default
 * search gathers).  Drains the device (a measurement aid). */
int ldot_index_last_set_stats(ldot_index_t* ix, int64_t out[2]);
/* LDOT_OPT_VERIFY = 1: flags_out [nq of the last search] (host, may be NULL) receives 1 for every query whose result is not proven
 * exact, *count_out their number.  No reference counterpart (faiss IndexFlatIP is fp32 end to end); this is how the bf16 candidate
 * pass reports that its margin may have been too small for a query. */
int ldot_index_last_unproven(ldot_index_t* ix, int32_t* flags_out, int64_t* count_out);

/* ---------------------------------------------------------------------------------------------------------
 * Partial top-k merge (sharded retrieval; no reference counterpart — the reference is single
 * process).  Inputs: `nparts` lists per query laid out [nparts][nq][k_in] (scores fp32, labels int64 already
 * global, LDOT_PAD_LABEL for padding).  Output [nq][k_out], score desc / label asc.
 * --------------------------------------------------------------------------------------------------------- */
int ldot_merge_topk(const float* scores, const int64_t* labels, int nparts, int64_t nq, int k_in, int k_out,
                    float* out_scores, int64_t* out_labels, int mem, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * [CLS] pooling — dvl/models/bi_encoder.py:120 (BertEncoder) and :188 (UniterEncoder):
 *     pooled_output = sequence_output[:, 0, :]
 * seq: [B, L, D] (row strides given in elements) device memory of `dtype`; out_f32 [B, D] and/or out_bf16
 * [B, D] (either may be NULL).  `normalize` is the opt-in L2 variant of north_star.
 * --------------------------------------------------------------------------------------------------------- */
int ldot_cls_pool(const void* seq, int dtype, int64_t B, int64_t stride_b, int64_t D, int normalize,
                  float* out_f32, void* out_bf16, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * In-batch contrastive loss — dvl/models/bi_encoder.py:615-656 (BiEncoderNllLoss.calc) incl. :54-68
 * (dot_product_scores), reached through dvl/utils.py:114-169 (_calc_loss) from train_itm.py:198-210 and
 * dvl/trainer.py:143.
 *   scores = (1-w) * q.ctx^T + w * q.cap^T        (cap may be NULL or w == 0 -> scores = q.ctx^T)
 *   lse_i  = logsumexp_j scores[i][j];  row_loss_i = lse_i - scores[i][pos_i];  argmax_i = first max column
 * All buffers are device memory, fp32, row-major: q [n1,d], ctx [n2,d], cap [n2,d] or NULL, pos int32 [n1].
 * Outputs: scores [n1,n2], row_loss [n1], lse [n1], correct int32 [1] (= #{argmax_i == pos_i}),
 * loss_mean [1] (= mean of row_loss).  fp32-input MFMA (exact fp32 products) is used, so results match an
 * fp32 reference to rounding.
 * --------------------------------------------------------------------------------------------------------- */
int ldot_inbatch_nll_fwd(const float* q, const float* ctx, const float* cap, float w, const int32_t* pos,
                         int64_t n1, int64_t n2, int64_t d, float* scores, float* row_loss, float* lse,
                         int32_t* correct, float* loss_sum, void* stream);
/* Backward: dS_ij = g_row_i * (exp(scores_ij - lse_i) - [j == pos_i]) + g_scores_ij   (g_scores may be NULL)
 *           dq = (1-w) dS.ctx + w dS.cap ;  dctx = (1-w) dS^T.q ;  dcap = w dS^T.q   (NULL outputs skipped)
 * `ds_work` is an [n1,n2] fp32 scratch buffer. */
int ldot_inbatch_nll_bwd(const float* q, const float* ctx, const float* cap, float w, const int32_t* pos,
                         int64_t n1, int64_t n2, int64_t d, const float* scores, const float* lse,
                         const float* g_row, const float* g_scores, float* ds_work,
                         float* dq, float* dctx, float* dcap, void* stream);

/* The bidirectional loss of ONE fine-tuning step — train_itm.py:195-222: two _calc_loss calls (dvl/utils.py:114-169) on the same
 * embeddings, img[:bs] against txt and txt[:bs] against img, positives pos[i] for both, averaged:
 *     S_txt = img[:bs].txt^T   [bs,n]     loss_txt = mean_i NLL(log_softmax(S_txt)_i, pos_i)       (train_itm.py:198-199 / :205-206)
 *     S_img = txt[:bs].img^T   [bs,n]     loss_img = mean_i NLL(log_softmax(S_img)_i, pos_i)       (:200-202 / :208-210)
 *     loss_nce = 0.5 loss_txt + 0.5 loss_img (:212),  is_correct = (#correct_txt + #correct_img) / 2 (:211),
 *     scores_avg = 0.5 S_txt + 0.5 S_img (:222; s_avg may be NULL)
 * img, txt: [n,d] fp32 device rows (n = bs + bs * num_hard_negatives, the first bs rows are the in-batch items).  S_img[:, :bs] is
 * the transpose of S_txt[:, :bs]: ONE GEMM tile pass produces both, with row and column softmax statistics in its epilogue (no caption
 * mixing on this path: callers with caption_score_weight != 0 use ldot_inbatch_nll_fwd twice).
 * Outputs: s_txt, s_img [bs,n]; lse, row_loss [2][bs] (direction txt, then img); out[6] = {loss_txt, loss_img, loss_nce, is_correct,
 * #correct_txt, #correct_img}.  Nothing is read back by the host. */
int ldot_inbatch_nll_bidir_fwd(const float* img, const float* txt, const int32_t* pos, int64_t bs, int64_t n, int64_t d,
                               float* s_txt, float* s_img, float* s_avg, float* lse, float* row_loss, float* out, void* stream);
/* Backward of the above.  g_nce, g_txt, g_img: DEVICE scalars (gradients w.r.t. loss_nce / loss_txt / loss_img; NULL = 0), g_avg [bs,n]
 * (gradient w.r.t. scores_avg — the KD branch of train_itm.py:224-241; NULL = 0).  ds_work: [bs*n + bs*(n-bs)] fp32 scratch.
 * dimg, dtxt: [n,d] (either may be NULL). */
int ldot_inbatch_nll_bidir_bwd(const float* img, const float* txt, const int32_t* pos, int64_t bs, int64_t n, int64_t d,
                               const float* s_txt, const float* s_img, const float* lse, const float* g_nce, const float* g_txt,
                               const float* g_img, const float* g_avg, float* ds_work, float* dimg, float* dtxt, void* stream);

/* plain score matrix (dot_product_scores, bi_encoder.py:54-68): out[n1,n2] = q.ctx^T, fp32 exact products */
int ldot_dot_product_scores(const float* q, const float* ctx, int64_t n1, int64_t n2, int64_t d, float* out,
                            void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Profiling hook (bench.py): with LDOT_OPT_PROFILE = 1 every score-kernel launch of a search (dense GEMM or
 * fused filter) is bracketed by HIP events on the search stream.  After the search:
 *   out[0] = number of score-kernel launches      out[1] = their total duration in milliseconds
 *   out[2] = total algorithmic flops (2*Q*N*D)    out[3] = total algorithmic bytes (index + queries + results)
 * --------------------------------------------------------------------------------------------------------- */
int ldot_index_last_profile(const ldot_index_t* ix, double out[4]);

#ifdef __cplusplus
}
#endif
#endif /* LDOT_H_ */
