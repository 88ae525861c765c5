// Internal state of an index handle + the functions the library's translation units share (not part of the ABI: include/ldot.h is).
//   api.hip      index management behind the C ABI: create / destroy / options / add / get_rows / statistics / save / load / merge / pooling
//   search.hip   the search entry points: ingest -> candidate pass -> (exchange) -> re-score, in one, two or three calls
//   scan.hip     candidate passes and their adaptive state: dense chunks, the narrow search, the fused scan (warm-up, growth schedule,
//                optimistic / pooled thresholds, scan order), overflow check and recovery
//   ivf_api.hip  the approximate (inverted-file) search
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "kernels.h"

namespace ldot {

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return LDOT_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        LDOT_HIP_CHECK(hipMalloc(&p, need));
        bytes = need;
        return LDOT_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

}  // namespace ldot
using namespace ldot;

// every entry point that takes an index runs on the device the index was created on, whatever the caller's current device is
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

struct ldot_index {
    int device = -1;
    int d = 0, dpad = 0;
    int64_t ntotal = 0, cap_rows = 0;
    float* x32 = nullptr;      // [cap_rows][dpad] fp32 master copy (zero padded)
    uint16_t* x16b = nullptr;  // bf16 shadow (dpad per row, or 3*dpad split-bf16 [hi|hi|lo] with precision 1) in the blocked
                               // layout both MFMA kernels stream: 1 KiB blocks of 16 rows x 32 k
    int precision = 0;
    int64_t ld16() const { return precision ? 3 * (int64_t)dpad : dpad; }
    // options
    int mode = LDOT_MODE_AUTO;
    int rescore = 1;
    int64_t chunk_rows = 32768;
    int margin = -1;
    int profile = 0;
    int64_t warm_rows = 4096;
    bool warm_rows_set = false;   // LDOT_OPT_WARM_ROWS was set by the caller (a shard on pooled statistics otherwise warms up on fewer rows)
    int growth_pct = 150;
    int defer_sync = 0;           // LDOT_OPT_DEFER_SYNC
    int result_set = 0;           // LDOT_OPT_RESULT_SET: searches report the top-k SET (exact re-score of the boundary candidates only)
    DevBuf w_set_stats;           // {candidates gathered, live candidates} of the last search in that mode (two uint64 on the device)
    bool set_stats_valid = false;
    struct ProfEv {
        hipEvent_t a, b;
        double flops, bytes;
    };
    std::vector<ProfEv> prof_events;
    std::vector<hipEvent_t> prof_pool;   // events of finished searches, reused (creating and destroying ten per search is host time inside the step)
    double prof[4] = {0, 0, 0, 0};
    // workspaces
    DevBuf w_q16b;
    DevBuf w_stage, w_q32, w_ls, w_li, w_S, w_outs, w_outl, w_tau, w_pool, w_pool_cnt, w_over, w_cur_save;
    DevBuf w_part_s, w_part_l, w_mrg_s, w_mrg_l;
    DevBuf w_redone;   // flags of the queries the recovery searched again (kept for a shard's end-of-scan statistics)
    int64_t stats[4] = {0, 0, 0, 0};
    // the sub-pool counters and overflow flags are all-zero between searches (the pool select resets the counters it
    // reads); they are cleared only after a (re)allocation or an aborted / overflowed search
    bool pools_clean = false, flags_clean = false;
    // overflow summary: w_over_sum = {number of overflowed queries} on the device, mirrored into pinned host memory by one
    // 4-byte copy per search
    DevBuf w_over_sum;
    int32_t* h_over_sum = nullptr;
    // completion stamp of stream_wait (scan.hip): one pinned, device-mapped word the stream writes when everything before it is done
    uint32_t* h_stamp = nullptr;
    void* d_stamp = nullptr;
    uint32_t stamp_seq = 0;
    bool overflow_pending = false;   // a fused scan ran and its overflow summary has not been looked at yet
    // stats[0] (records appended by the fused filter): per-query counts accumulated on the device by the pool selects, summed on
    // the host only when ldot_index_last_stats is called
    DevBuf w_qcnt;
    int64_t qcnt_n = 0;
    // LDOT_OPT_VERIFY: per-query "not proven exact" flags of the last search (see ldot_index_last_unproven)
    int verify = 0;
    DevBuf w_unproven;
    int64_t unproven_n = 0;
    DevBuf w_norm;                   // device scalar: largest L2 norm of an indexed row
    // narrow search (<= 64 queries): run maxima, threshold keys, candidate keys + counters (zero between searches)
    DevBuf w_nmax, w_ntau, w_ncand, w_ncnt;
    DevBuf w_lplist, w_lrowbase, w_lcstart, w_laug, w_lprobe_s, w_lprobe_l;   // list search: validated probes, prefix sums, coarse query / result
    bool narrow_clean = false;
    int32_t *h_nover = nullptr, *d_nover = nullptr;   // per-query "candidate buffer full" flags (pinned, device-mapped)
    int64_t overflow_narrow = 0;                      // > 0: the pending overflow summary is h_nover[0 .. overflow_narrow)
    // rows stored in cluster order can fill the candidate buffer on EVERY search of an index: after an overflow the narrow search is
    // skipped for `narrow_backoff` searches, twice as many after every further overflow (reset by a search that fits)
    int narrow_backoff = 0, narrow_penalty = 16;
    // recovery of overflowed queries (redo_flagged): indices of the flagged queries + compact copies of their operands and lists
    struct Compact {
        DevBuf fidx, q32, q16b, ls, li, tau;
    } compact[2];   // (level 0: the fused re-scan, level 1: the dense last resort for what overflows even then)
    bool overflow_was_narrow = false;   // the overflow the last check reported came from the narrow search's candidate buffers
    int64_t redone = 0;                 // queries searched again by the last search (ldot_index_last_stats: dense_pairs stays the dense work)
    bool pend_done = false;             // the narrow search's finish kernel has already written the caller's outputs
    const void* unstaged_q = nullptr;   // the last search read the caller's fp32 queries directly (DirectOut::qf32): w_q32 / w_q16b are NOT filled
    int64_t unstaged_ld = 0;            // ... their row stride
    // set by ldot_ivf_search around its coarse search (an internal chain, not part of the ABI): the queries are fp32 rows padded with zeros
    // to dpad columns (row stride dpad), and a search whose finish kernel wrote the outputs returns WITHOUT the synchronisation + buffer-full
    // check — the chain checks at its own synchronisation point (overflow_pending stays set)
    bool q_prepadded = false, chain_defer_sync = false;
    // a search in two halves (ldot_index_search_begin / _finish): what _finish needs to know
    int64_t pend_nq = 0;
    int pend_k = 0, pend_kp = 0;
    // ... or in three (ldot_index_search_warmup / _scan / _finish, the sharded search): what _scan needs to know.  split_path: 0 none
    // pending, 1 narrow search, 2 dense scan, 3 fused scan whose warm-up has run
    int split_path = 0, split_parts = 1;
    // optimistic thresholds (LDOT_OPT_OPTIMISTIC, fused_rest_chunk): what the filter compares with while the guaranteed threshold
    // (w_tau: the list's own k'-th best) is still far below the final one
    int optimistic = 1;
    DevBuf w_tau_opt;
    // rows stored in an order that correlates with the queries (cluster-sorted rows: what the inverted-file index keeps) fail the
    // end-of-scan check for a large share of the queries on EVERY search, and a failed query costs a second scan: after a search that
    // flagged more than 1 / 64 of its queries the optimistic schedule is skipped for `opt_backoff` searches, twice as many after every
    // further failure (reset by a search that passes)
    int opt_backoff = 0, opt_penalty = 16;
    // LDOT_OPT_SCAN_ORDER: 0 auto (sequential until the optimistic check fails for more than 1 / 64 of a search's queries, then scrambled
    // for the rest of the index's life), 1 sequential, 2 scrambled.  scrambled_now: the optimistic scan in progress visits the row tiles
    // in the pseudo-random order (fused_rest_chunk_optimistic)
    int scan_order = 0;
    bool scrambled_auto = false, scrambled_now = false;
    bool opt_used = false;           // the scan in progress filtered with optimistic thresholds
    int64_t opt_nq = 0;
    int cur_parts = 1;   // shards of the search in progress (1 = plain search): sizes the warm-up of a fused scan, fused_warm_rows
    // a shard scanning on POOLED statistics (ldot_index_search_begin_shard): rows of the whole sharded index (0 = off) and its number of
    // shards; pooled_used = the scan in progress filtered with thresholds only the ranks together can verify (w_tau_opt = their level)
    int64_t pool_total = 0;
    int pool_parts = 1;
    bool pooled_used = false;
    // LDOT_OPT_ROW_SHUFFLE: rows stored in a pseudo-random order behind a label table.  row_shuffle: 0 auto (rows are stored as added; the
    // store is re-shuffled ONCE when a large-batch search fails the optimistic check in the scrambled tile order too — rows sorted in runs
    // about as long as a tile —, adds are shuffled from then on), 1 every add is shuffled, 2 never.  shuffled: the tables exist — stored row p
    // carries label w_label[p], label l sits at row w_pos[l] (int32 [cap_rows] each)
    int row_shuffle = 0;
    bool shuffled = false, reshuffled = false, want_reshuffle = false;
    DevBuf w_label, w_pos;
    uint64_t shuffle_calls = 0;
    // what the last search did (ldot_index_last_regime)
    int last_path = 0, last_thresholds = 0, last_order = 0;
};


// device-visible destination of a search's final top-k (device memory, or pinned host memory mapped into the device's address space)
struct DirectOut {
    float* scores;
    int64_t* labels;
    int k;
    // the caller's queries when they have NOT been staged (fp32 rows in device memory, row stride = d = dpad): the one-launch narrow
    // search converts them inside the scan kernel and re-scores from them, which saves the conversion kernel of a few-query search
    const float* qf32 = nullptr;
    int64_t ldqf = 0;   // ... their row stride = the number of columns the kernels read (d, or dpad for zero-padded rows)
};

constexpr int64_t kListsQueryChunk = 256;       // queries per pass of the run-maxima selection (bounds its buffers and flag array)
constexpr int64_t kFewSelectMaxQueries = 256;   // one query block: sub-pools folded by 16 waves per query + one merge

inline size_t dtype_size(int dtype) { return dtype == LDOT_F32 ? 4 : 2; }

// api.hip
int reshuffle_rows(ldot_index* ix, hipStream_t st);
// scan.hip
int candidate_len(const ldot_index* ix, int k);
void prof_collect(ldot_index* ix, hipStream_t st);
int narrow_buffers(ldot_index* ix, int64_t nq, int64_t ldm, hipStream_t st);
bool narrow_select_ok(const ldot_index* ix, int64_t nq, int kp);
bool narrow_one_launch(const ldot_index* ix, int64_t nq, int kp);
int narrow_search(ldot_index* ix, int64_t nq, int kp, hipStream_t st, const DirectOut* direct = nullptr);
int dense_scan_all(ldot_index* ix, int64_t nq, int64_t r0, int64_t r1, int kp, float* tau, bool allow_wide,
                          hipStream_t st, int64_t q_base = 0);
int fused_scan(ldot_index* ix, int64_t nq, int64_t nq_pad, int kp, hipStream_t st, int phase = 0, int parts = 1);
bool fused_overflow_check(ldot_index* ix);
int redo_flagged(ldot_index* ix, int64_t nq, int64_t nq_pad, int kp, hipStream_t st, int level = 0);
int stage_unstaged_queries(ldot_index* ix, int64_t nq, hipStream_t st);
bool auto_fused(const ldot_index* ix, int64_t nq);
// Wait until everything enqueued on `st` so far is done — without the runtime's wait.  hipStreamSynchronize spins ~100 us and then sleeps
// until the runtime's event thread has handled the completion interrupt: with other work on the host's cores that thread — and with it the
// waiter — gets the CPU tens of milliseconds late (profiles/r06_stall_trace.txt, r06_host_stall_probe.txt: 30-90 ms stalls of a 0.36-ms
// evaluation with the device idle after 2 ms, the waiting thread runnable-but-not-running; a torch-only loop shows them too).  Here the
// stream writes a sequence number into a pinned word (hipStreamWriteValue32) and the calling thread polls it: no second thread, no sleep.
// LDOT_HOST_WAIT=runtime in the environment (or a runtime without the stream operation) falls back to hipStreamSynchronize.
int stream_wait(ldot_index* ix, hipStream_t st);
// search.hip
int search_finish_impl(ldot_index_t* ix, const float* floor, float* out_scores, int64_t* out_labels, int out_mem,
                              bool keep_pending, hipStream_t st);
