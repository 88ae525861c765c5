// C ABI of libldot.so (see include/ldot.h): index object, search orchestration, merge, pooling.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "kernels.h"

namespace ldot {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

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

// label tables of a shuffled index for `cap` rows (contents of the first ntotal entries are kept)
static int tables_reserve(ldot_index* ix, int64_t cap, hipStream_t st) {
    const size_t need = (size_t)cap * 4;
    for (DevBuf* b : {&ix->w_label, &ix->w_pos}) {
        if (b->bytes >= need) continue;
        void* np = nullptr;
        LDOT_HIP_CHECK(hipMalloc(&np, need));
        hipError_t e = hipSuccess;
        if (b->p && ix->ntotal > 0) e = hipMemcpyAsync(np, b->p, (size_t)ix->ntotal * 4, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            (void)hipFree(np);
            set_error("label table copy failed: %s", hipGetErrorString(e));
            return LDOT_EDEVICE;
        }
        if (b->p) (void)hipFree(b->p);
        b->p = np;
        b->bytes = need;
    }
    return LDOT_OK;
}

static int index_reserve(ldot_index* ix, int64_t rows, hipStream_t st) {
    // capacity is a multiple of 256 rows (the MFMA tile) and rows beyond ntotal are kept zero
    // (+512 zero rows of slack: the fused kernel's 384-row tiles may read past the last 256-row boundary)
    int64_t need = round_up(rows > 0 ? rows : 1, 256) + 512;
    if (need <= ix->cap_rows) return LDOT_OK;
    int64_t cap = std::max<int64_t>(need, ix->cap_rows + ix->cap_rows / 2);
    cap = round_up(cap, 256);
    float* n32 = nullptr;
    uint16_t* n16b = nullptr;
    const size_t b32 = (size_t)cap * ix->dpad * sizeof(float), b16 = (size_t)cap * ix->ld16() * sizeof(uint16_t);
    LDOT_HIP_CHECK(hipMalloc((void**)&n32, b32));
    hipError_t e = hipMalloc((void**)&n16b, b16);
    if (e != hipSuccess) {
        (void)hipFree(n32);
        set_error("hipMalloc(%zu) failed: %s", b16, hipGetErrorString(e));
        return LDOT_ENOMEM;
    }
    const size_t u32 = (size_t)ix->ntotal * ix->dpad * sizeof(float);
    if (ix->ntotal > 0) {
        LDOT_HIP_CHECK(hipMemcpyAsync(n32, ix->x32, u32, hipMemcpyDeviceToDevice, st));
    }
    // blocked shadow: whole 16-row blocks (the last one may be partly filled; its unused rows are zero)
    const size_t u16b = (size_t)round_up(ix->ntotal, 16) * ix->ld16() * 2;
    if (ix->ntotal > 0) LDOT_HIP_CHECK(hipMemcpyAsync(n16b, ix->x16b, u16b, hipMemcpyDeviceToDevice, st));
    LDOT_HIP_CHECK(hipMemsetAsync((char*)n16b + u16b, 0, b16 - u16b, st));
    LDOT_HIP_CHECK(hipMemsetAsync((char*)n32 + u32, 0, b32 - u32, st));
    LDOT_HIP_CHECK(hipStreamSynchronize(st));
    if (ix->x32) (void)hipFree(ix->x32);
    if (ix->x16b) (void)hipFree(ix->x16b);
    ix->x32 = n32;
    ix->x16b = n16b;
    ix->cap_rows = cap;
    if (ix->shuffled) return tables_reserve(ix, cap, st);
    return LDOT_OK;
}

// first shuffled add (or the re-shuffle) of an index: label tables, identity for the rows already stored
static int shuffle_engage(ldot_index* ix, hipStream_t st) {
    if (ix->shuffled) return LDOT_OK;
    int rc = tables_reserve(ix, ix->cap_rows, st);
    if (rc) return rc;
    if ((rc = launch_perm_labels((int32_t*)ix->w_label.p, (int32_t*)ix->w_pos.p, 0, ix->ntotal, RowPerm(), st))) return rc;
    ix->shuffled = true;
    return LDOT_OK;
}

static RowPerm row_perm(ldot_index* ix, int64_t n) {
    RowPerm p;
    if (n < 2) return p;
    p.mul = scan_order_multiplier(n);   // ~ n / golden ratio, coprime to n: consecutive stored rows come from far apart
    p.add = (int64_t)((0x9E3779B97F4A7C15ull * ++ix->shuffle_calls) >> 20) % n;
    p.n = n;
    return p;
}

// LDOT_OPT_ROW_SHUFFLE auto: the rows already stored are re-ordered pseudo-randomly (new row j = old row (mul j + add) mod n), once per index.
// Built into NEW buffers (fp32 master, bf16 shadow, label tables) that replace the old ones only when everything has succeeded: a failure
// leaves the index exactly as it was.  Needs room for a second copy of the store while it runs; without it the index stays as it is.
// `st` is synchronised.
static int reshuffle_rows(ldot_index* ix, hipStream_t st) {
    const int64_t n = ix->ntotal;
    ix->reshuffled = true;
    if (n < 2) return LDOT_OK;
    LDOT_HIP_CHECK(hipStreamSynchronize(st));
    const size_t b32 = (size_t)ix->cap_rows * ix->dpad * 4, b16 = (size_t)ix->cap_rows * ix->ld16() * 2, bt = (size_t)ix->cap_rows * 4;
    float* n32 = nullptr;
    uint16_t* n16b = nullptr;
    void *nl = nullptr, *np = nullptr, *idx = nullptr;
    auto drop = [&]() {
        for (void* p : {(void*)n32, (void*)n16b, nl, np, idx})
            if (p) (void)hipFree(p);
    };
    if (hipMalloc((void**)&n32, b32) != hipSuccess || hipMalloc((void**)&n16b, b16) != hipSuccess || hipMalloc(&nl, bt) != hipSuccess ||
        hipMalloc(&np, bt) != hipSuccess || hipMalloc(&idx, bt) != hipSuccess) {
        (void)hipGetLastError();
        drop();
        return LDOT_OK;
    }
    const RowPerm perm = row_perm(ix, n);
    int rc = launch_reshuffle_tables(ix->shuffled ? (const int32_t*)ix->w_label.p : nullptr, (int32_t*)nl, (int32_t*)np, (int32_t*)idx, n, perm, st);
    if (!rc) rc = launch_gather_rows_f32(ix->x32, ix->dpad, (const int32_t*)idx, n, n, n32, st);
    hipError_t e = hipMemsetAsync(n32 + n * ix->dpad, 0, b32 - (size_t)n * ix->dpad * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(n16b, 0, b16, st);
    if (!rc && e == hipSuccess)
        rc = launch_convert_rows(n32, LDOT_F32, ix->dpad, n, n, ix->d, ix->dpad, 0, nullptr, nullptr, ix->precision ? 1 : 0, n16b, 0, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (rc || e != hipSuccess) {
        drop();
        if (!rc) {
            set_error("re-shuffle failed: %s", hipGetErrorString(e));
            rc = LDOT_EDEVICE;
        }
        return rc;
    }
    (void)hipFree(idx);
    (void)hipFree(ix->x32);
    (void)hipFree(ix->x16b);
    ix->x32 = n32;
    ix->x16b = n16b;
    ix->w_label.release();
    ix->w_pos.release();
    ix->w_label.p = nl;
    ix->w_pos.p = np;
    ix->w_label.bytes = ix->w_pos.bytes = bt;
    ix->shuffled = true;
    return LDOT_OK;
}

static size_t dtype_size(int dtype) { return dtype == LDOT_F32 ? 4 : 2; }

extern "C" {

const char* ldot_last_error(void) { return g_err; }
int ldot_abi_version(void) { return LDOT_ABI_VERSION; }

int ldot_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return LDOT_EDEVICE;
    }
    return n;
}

int ldot_index_create(int d, ldot_index_t** out) {
    LDOT_REQUIRE(out != nullptr, LDOT_EINVAL, "out is NULL");
    LDOT_REQUIRE(d > 0 && d <= 65536, LDOT_EINVAL, "bad dimension %d", d);
    int n = ldot_device_count();
    if (n < 0) return n;
    LDOT_REQUIRE(n > 0, LDOT_EDEVICE, "no HIP device visible: the MI355X path has no CPU fallback");
    ldot_index* ix = new (std::nothrow) ldot_index();
    LDOT_REQUIRE(ix != nullptr, LDOT_ENOMEM, "out of host memory");
    ix->d = d;
    ix->dpad = (int)round_up(d, kBK);
    (void)hipGetDevice(&ix->device);
    *out = ix;
    return LDOT_OK;
}

int ldot_index_destroy(ldot_index_t* ix) {
    if (!ix) return LDOT_OK;
    if (ix->x32) (void)hipFree(ix->x32);
    if (ix->x16b) (void)hipFree(ix->x16b);
    DevBuf* bufs[] = {&ix->w_q16b, &ix->w_stage, &ix->w_q32, &ix->w_ls, &ix->w_li, &ix->w_S, &ix->w_outs,
                      &ix->w_outl, &ix->w_tau, &ix->w_pool, &ix->w_pool_cnt, &ix->w_over, &ix->w_cur_save,
                      &ix->w_part_s, &ix->w_part_l, &ix->w_mrg_s, &ix->w_mrg_l};
    DeviceGuard guard(ix->device);
    for (DevBuf* b : bufs) b->release();
    ix->w_over_sum.release();
    ix->w_qcnt.release();
    ix->w_tau_opt.release();
    ix->w_redone.release();
    ix->w_label.release();
    ix->w_pos.release();
    ix->w_unproven.release();
    ix->w_set_stats.release();
    ix->w_norm.release();
    ix->w_nmax.release();
    ix->w_ntau.release();
    ix->w_ncand.release();
    ix->w_ncnt.release();
    ix->w_lplist.release();
    ix->w_lcstart.release();
    ix->w_lrowbase.release();
    ix->w_laug.release();
    ix->w_lprobe_s.release();
    ix->w_lprobe_l.release();
    for (auto& c : ix->compact)
        for (DevBuf* b : {&c.fidx, &c.q32, &c.q16b, &c.ls, &c.li, &c.tau}) b->release();
    for (auto& ev : ix->prof_events) {
        (void)hipEventDestroy(ev.a);
        (void)hipEventDestroy(ev.b);
    }
    for (hipEvent_t e : ix->prof_pool) (void)hipEventDestroy(e);
    if (ix->h_over_sum) (void)hipHostFree(ix->h_over_sum);
    if (ix->h_nover) (void)hipHostFree(ix->h_nover);
    delete ix;
    return LDOT_OK;
}

int64_t ldot_index_ntotal(const ldot_index_t* ix) { return ix ? ix->ntotal : LDOT_EINVAL; }
int ldot_index_dim(const ldot_index_t* ix) { return ix ? ix->d : LDOT_EINVAL; }

// (reset and the storage-changing options take no stream argument: they drain the DEVICE first and when done, so they are ordered
// against work on any stream, blocking or not)
int ldot_index_reset(ldot_index_t* ix) {
    LDOT_REQUIRE(ix != nullptr, LDOT_EINVAL, "index is NULL");
    DeviceGuard guard(ix->device);
    if (ix->cap_rows > 0) {
        LDOT_HIP_CHECK(hipDeviceSynchronize());
        LDOT_HIP_CHECK(hipMemset(ix->x32, 0, (size_t)ix->cap_rows * ix->dpad * 4));
        LDOT_HIP_CHECK(hipMemset(ix->x16b, 0, (size_t)ix->cap_rows * ix->ld16() * 2));
        LDOT_HIP_CHECK(hipDeviceSynchronize());
    }
    ix->ntotal = 0;
    ix->shuffled = ix->reshuffled = ix->want_reshuffle = false;   // (an empty index is in storage order again; the option stays)
    ix->w_label.release();
    ix->w_pos.release();
    if (ix->w_norm.p) LDOT_HIP_CHECK(hipMemset(ix->w_norm.p, 0, 16));
    return LDOT_OK;
}

int ldot_index_set_option(ldot_index_t* ix, int option, int64_t value) {
    LDOT_REQUIRE(ix != nullptr, LDOT_EINVAL, "index is NULL");
    DeviceGuard guard(ix->device);
    switch (option) {
        case LDOT_OPT_VERIFY:
            LDOT_REQUIRE(value == 0 || value == 1, LDOT_EINVAL, "verify must be 0 or 1");
            ix->verify = (int)value;
            return LDOT_OK;
        case LDOT_OPT_MODE:
            LDOT_REQUIRE(value >= 0 && value <= 2, LDOT_EINVAL, "bad mode %lld", (long long)value);
            ix->mode = (int)value;
            return LDOT_OK;
        case LDOT_OPT_RESCORE:
            ix->rescore = value ? 1 : 0;
            return LDOT_OK;
        case LDOT_OPT_CHUNK_ROWS:
            LDOT_REQUIRE(value >= 256 && value % 256 == 0 && value <= (1 << 22), LDOT_EINVAL,
                         "chunk_rows must be a multiple of 256 in [256, 4194304]");
            ix->chunk_rows = value;
            return LDOT_OK;
        case LDOT_OPT_MARGIN:
            LDOT_REQUIRE(value >= -1 && value <= kMaxKp - kMaxK, LDOT_EINVAL, "bad margin (max %d)", kMaxKp - kMaxK);
            ix->margin = (int)value;
            return LDOT_OK;
        case LDOT_OPT_PROFILE:
            ix->profile = value ? 1 : 0;
            return LDOT_OK;
        case LDOT_OPT_WARM_ROWS:
            LDOT_REQUIRE(value >= 2048 && value % 256 == 0, LDOT_EINVAL, "warm_rows must be a multiple of 256 >= 2048");
            ix->warm_rows = value;
            ix->warm_rows_set = true;
            return LDOT_OK;
        case LDOT_OPT_PRECISION: {
            LDOT_REQUIRE(value == 0 || value == 1, LDOT_EINVAL, "precision must be 0 (bf16) or 1 (split bf16)");
            if ((int)value == ix->precision) return LDOT_OK;
            ix->precision = (int)value;
            if (ix->cap_rows == 0) return LDOT_OK;
            LDOT_HIP_CHECK(hipDeviceSynchronize());
            // rebuild the shadow from the fp32 master copy in the new layout
            uint16_t* n16b = nullptr;
            const size_t b16 = (size_t)ix->cap_rows * ix->ld16() * sizeof(uint16_t);
            hipError_t e = hipMalloc((void**)&n16b, b16);
            if (e != hipSuccess) {
                ix->precision = 1 - ix->precision;
                set_error("hipMalloc(%zu) failed: %s", b16, hipGetErrorString(e));
                return LDOT_ENOMEM;
            }
            LDOT_HIP_CHECK(hipMemsetAsync(n16b, 0, b16, nullptr));
            int rc = launch_convert_rows(ix->x32, LDOT_F32, ix->dpad, ix->ntotal, ix->ntotal, ix->d, ix->dpad, 0, nullptr,
                                         nullptr, ix->precision ? 1 : 0, n16b, 0, nullptr);
            LDOT_HIP_CHECK(hipDeviceSynchronize());
            if (rc) {
                (void)hipFree(n16b);
                return rc;
            }
            (void)hipFree(ix->x16b);
            ix->x16b = n16b;
            return LDOT_OK;
        }
        case LDOT_OPT_RESERVE_ROWS: {
            // allocate capacity up front: a large index then never pays the x1.5 growth copies (which transiently need old +
            // new buffers in HBM)
            LDOT_REQUIRE(value >= 0 && value < 0x7ffffff0ll, LDOT_EINVAL, "bad reserve size");
            LDOT_HIP_CHECK(hipDeviceSynchronize());
            return index_reserve(ix, std::max<int64_t>(value, ix->ntotal), nullptr);
        }
        case LDOT_OPT_OPTIMISTIC:
            LDOT_REQUIRE(value == 0 || value == 1, LDOT_EINVAL, "LDOT_OPT_OPTIMISTIC is 0 or 1");
            ix->optimistic = (int)value;
            return LDOT_OK;
        case LDOT_OPT_SCAN_ORDER:
            LDOT_REQUIRE(value >= 0 && value <= 2, LDOT_EINVAL, "LDOT_OPT_SCAN_ORDER is 0 (auto), 1 (sequential) or 2 (scrambled)");
            ix->scan_order = (int)value;
            ix->scrambled_auto = false;
            return LDOT_OK;
        case LDOT_OPT_ROW_SHUFFLE:
            LDOT_REQUIRE(value >= 0 && value <= 2, LDOT_EINVAL, "LDOT_OPT_ROW_SHUFFLE is 0 (auto), 1 (shuffle every add) or 2 (never)");
            LDOT_REQUIRE(!(value == 2 && ix->shuffled), LDOT_ESTATE, "the rows of this index are shuffled already (reset it first)");
            ix->row_shuffle = (int)value;
            if (value == 2) ix->want_reshuffle = false;
            return LDOT_OK;
        case LDOT_OPT_DEFER_SYNC:
            LDOT_REQUIRE(value == 0 || value == 1, LDOT_EINVAL, "LDOT_OPT_DEFER_SYNC is 0 or 1");
            ix->defer_sync = (int)value;
            return LDOT_OK;
        case LDOT_OPT_RESULT_SET:
            LDOT_REQUIRE(value == 0 || value == 1, LDOT_EINVAL, "LDOT_OPT_RESULT_SET is 0 or 1");
            ix->result_set = (int)value;
            return LDOT_OK;
        case LDOT_OPT_GROWTH_PCT:
            LDOT_REQUIRE(value >= 5 && value <= 10000, LDOT_EINVAL, "growth_pct must be in [5, 10000]");
            ix->growth_pct = (int)value;
            return LDOT_OK;
        default:
            set_error("unknown option %d", option);
            return LDOT_EINVAL;
    }
}

int ldot_index_add(ldot_index_t* ix, const void* rows, int64_t n, int dtype, int mem, int normalize, void* stream) {
    LDOT_REQUIRE(ix != nullptr, LDOT_EINVAL, "index is NULL");
    LDOT_REQUIRE(n >= 0, LDOT_EINVAL, "negative row count");
    LDOT_REQUIRE(dtype >= 0 && dtype <= 2, LDOT_EINVAL, "bad dtype %d", dtype);
    LDOT_REQUIRE(mem == LDOT_HOST || mem == LDOT_DEVICE, LDOT_EINVAL, "bad mem %d", mem);
    if (n == 0) return LDOT_OK;
    LDOT_REQUIRE(rows != nullptr, LDOT_EINVAL, "rows is NULL");
    LDOT_REQUIRE(ix->ntotal + n < 0x7ffffff0ll, LDOT_EINVAL, "index too large for 31-bit row labels");
    DeviceGuard guard(ix->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = index_reserve(ix, ix->ntotal + n, st);
    if (rc) return rc;
    const void* src = rows;
    if (mem == LDOT_HOST) {
        const size_t bytes = (size_t)n * ix->d * dtype_size(dtype);
        rc = ix->w_stage.ensure(bytes);
        if (rc) return rc;
        LDOT_HIP_CHECK(hipMemcpyAsync(ix->w_stage.p, rows, bytes, hipMemcpyHostToDevice, st));
        src = ix->w_stage.p;
    }
    // LDOT_OPT_ROW_SHUFFLE: the rows of this call go into the store in a pseudo-random order; the label tables remember which is which
    RowPerm perm;
    if (ix->row_shuffle == 1 || (ix->row_shuffle == 0 && ix->shuffled)) {
        if ((rc = shuffle_engage(ix, st))) return rc;
        perm = row_perm(ix, n);
    }
    rc = launch_convert_rows(src, dtype, ix->d, n, n, ix->d, ix->dpad, normalize, ix->x32 + ix->ntotal * ix->dpad,
                             nullptr, ix->precision ? 1 : 0, ix->x16b, ix->ntotal, st, perm);
    if (rc) return rc;
    if (ix->shuffled && (rc = launch_perm_labels((int32_t*)ix->w_label.p, (int32_t*)ix->w_pos.p, ix->ntotal, n, perm, st))) return rc;
    // largest row norm so far (device scalar; the LDOT_OPT_VERIFY bound reads it)
    if (ix->w_norm.p == nullptr) {
        if ((rc = ix->w_norm.ensure(16))) return rc;
        LDOT_HIP_CHECK(hipMemsetAsync(ix->w_norm.p, 0, 16, st));
    }
    if ((rc = launch_row_norm_max(ix->x32 + ix->ntotal * ix->dpad, ix->dpad, n, ix->d, (float*)ix->w_norm.p, st))) return rc;
    if (mem == LDOT_HOST) LDOT_HIP_CHECK(hipStreamSynchronize(st));   // staging buffer is reused by the next call
    ix->ntotal += n;
    return LDOT_OK;
}

int ldot_index_get_rows(ldot_index_t* ix, int64_t row0, int64_t n, float* out, int out_mem, void* stream) {
    LDOT_REQUIRE(ix != nullptr && out != nullptr, LDOT_EINVAL, "NULL argument");
    LDOT_REQUIRE(row0 >= 0 && n >= 0 && row0 + n <= ix->ntotal, LDOT_EINVAL, "row range out of bounds");
    if (n == 0) return LDOT_OK;
    DeviceGuard guard(ix->device);
    hipStream_t st = (hipStream_t)stream;
    if (ix->shuffled) {   // rows are addressed by LABEL: gathered through the position table, in bounded chunks (ldot_index_save asks for
                          // the whole index at once: a full-size staging copy would double the fp32 footprint for the duration)
        constexpr int64_t kGatherChunk = 65536;
        int rc = ix->w_stage.ensure((size_t)std::min(n, kGatherChunk) * ix->dpad * 4);
        if (rc) return rc;
        for (int64_t r = 0; r < n; r += kGatherChunk) {
            const int64_t len = std::min(kGatherChunk, n - r);
            if ((rc = launch_gather_rows_f32(ix->x32, ix->dpad, (const int32_t*)ix->w_pos.p + row0 + r, len, len, (float*)ix->w_stage.p, st))) return rc;
            LDOT_HIP_CHECK(hipMemcpy2DAsync(out + r * ix->d, (size_t)ix->d * 4, ix->w_stage.p, (size_t)ix->dpad * 4, (size_t)ix->d * 4, (size_t)len,
                                            out_mem == LDOT_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
            LDOT_HIP_CHECK(hipStreamSynchronize(st));   // (the staging buffer is reused)
        }
        return LDOT_OK;
    }
    LDOT_HIP_CHECK(hipMemcpy2DAsync(out, (size_t)ix->d * 4, ix->x32 + row0 * ix->dpad, (size_t)ix->dpad * 4,
                                    (size_t)ix->d * 4, (size_t)n,
                                    out_mem == LDOT_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
    if (out_mem == LDOT_HOST) LDOT_HIP_CHECK(hipStreamSynchronize(st));
    return LDOT_OK;
}

int ldot_index_last_stats(const ldot_index_t* cix, int64_t out[4]) {
    LDOT_REQUIRE(cix != nullptr && out != nullptr, LDOT_EINVAL, "NULL argument");
    ldot_index* ix = const_cast<ldot_index*>(cix);
    if (ix->qcnt_n > 0) {   // per-query record counts of the last fused scan: summed here, not on the search path
        DeviceGuard guard(ix->device);
        std::vector<int32_t> h((size_t)ix->qcnt_n);
        LDOT_HIP_CHECK(hipDeviceSynchronize());
        LDOT_HIP_CHECK(hipMemcpy(h.data(), ix->w_qcnt.p, h.size() * 4, hipMemcpyDeviceToHost));
        int64_t tot = 0;
        for (int32_t v : h) tot += v;
        ix->stats[0] = tot;
        ix->qcnt_n = 0;
    }
    for (int i = 0; i < 4; ++i) out[i] = ix->stats[i];
    return LDOT_OK;
}

int ldot_index_last_regime(const ldot_index_t* ix, int64_t out[8]) {
    LDOT_REQUIRE(ix != nullptr && out != nullptr, LDOT_EINVAL, "NULL argument");
    out[0] = ix->last_path;
    out[1] = ix->last_thresholds;
    out[2] = ix->last_order;
    out[3] = ix->redone;
    out[4] = ix->opt_backoff;
    out[5] = ix->narrow_backoff;
    out[6] = ix->scrambled_auto ? 1 : 0;
    out[7] = ix->reshuffled && ix->shuffled ? 2 : ix->shuffled ? 1 : 0;
    return LDOT_OK;
}

int ldot_index_last_unproven(ldot_index_t* ix, int32_t* flags_out, int64_t* count_out) {
    LDOT_REQUIRE(ix != nullptr, LDOT_EINVAL, "index is NULL");
    LDOT_REQUIRE(ix->verify, LDOT_ESTATE, "LDOT_OPT_VERIFY is off");
    DeviceGuard guard(ix->device);
    const int64_t n = ix->unproven_n;
    int32_t cnt = 0;
    if (n > 0) {
        LDOT_HIP_CHECK(hipDeviceSynchronize());
        LDOT_HIP_CHECK(hipMemcpy(&cnt, (int32_t*)ix->w_unproven.p + n, 4, hipMemcpyDeviceToHost));
        if (flags_out) LDOT_HIP_CHECK(hipMemcpy(flags_out, ix->w_unproven.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    if (count_out) *count_out = cnt;
    return LDOT_OK;
}

int ldot_index_last_set_stats(ldot_index_t* ix, int64_t out[2]) {
    LDOT_REQUIRE(ix != nullptr && out != nullptr, LDOT_EINVAL, "NULL argument");
    out[0] = out[1] = 0;
    if (!ix->set_stats_valid || ix->w_set_stats.p == nullptr) return LDOT_OK;
    DeviceGuard guard(ix->device);
    unsigned long long h[2] = {0, 0};
    LDOT_HIP_CHECK(hipMemcpy(h, ix->w_set_stats.p, 16, hipMemcpyDeviceToHost));   // (drains the device: a measurement aid)
    out[0] = (int64_t)h[0];
    out[1] = (int64_t)h[1];
    return LDOT_OK;
}

static int candidate_len(const ldot_index* ix, int k) {
    int margin = ix->margin >= 0 ? ix->margin : std::max(28, k / 4);
    if (!ix->rescore) margin = 0;
    int kp = (int)round_up(k + margin, 32);
    return std::min(kp, kMaxKp);
}

static bool prof_event(ldot_index* ix, hipEvent_t* e) {
    if (!ix->prof_pool.empty()) {
        *e = ix->prof_pool.back();
        ix->prof_pool.pop_back();
        return true;
    }
    return hipEventCreate(e) == hipSuccess;
}
static void prof_begin(ldot_index* ix, hipStream_t st, double flops, double bytes) {
    if (!ix->profile) return;
    ldot_index::ProfEv ev;
    if (!prof_event(ix, &ev.a)) return;
    if (!prof_event(ix, &ev.b)) {
        ix->prof_pool.push_back(ev.a);
        return;
    }
    ev.flops = flops;
    ev.bytes = bytes;
    (void)hipEventRecord(ev.a, st);
    ix->prof_events.push_back(ev);
}
// the same for a kernel whose launcher records the two events itself, right around its launch; *a / *b stay NULL when profiling is off
static void prof_attach(ldot_index* ix, double flops, double bytes, hipEvent_t* a, hipEvent_t* b) {
    *a = *b = nullptr;
    if (!ix->profile) return;
    ldot_index::ProfEv ev;
    if (!prof_event(ix, &ev.a)) return;
    if (!prof_event(ix, &ev.b)) {
        ix->prof_pool.push_back(ev.a);
        return;
    }
    ev.flops = flops;
    ev.bytes = bytes;
    ix->prof_events.push_back(ev);
    *a = ev.a;
    *b = ev.b;
}
static void prof_end(ldot_index* ix, hipStream_t st) {
    if (!ix->profile || ix->prof_events.empty()) return;
    (void)hipEventRecord(ix->prof_events.back().b, st);
}
static void prof_collect(ldot_index* ix, hipStream_t st) {
    for (int i = 0; i < 4; ++i) ix->prof[i] = 0;
    if (!ix->profile) return;
    (void)hipStreamSynchronize(st);
    for (auto& ev : ix->prof_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
            ix->prof[0] += 1;
            ix->prof[1] += ms;
            ix->prof[2] += ev.flops;
            ix->prof[3] += ev.bytes;
        }
        ix->prof_pool.push_back(ev.a);
        ix->prof_pool.push_back(ev.b);
    }
    ix->prof_events.clear();
}

// dense scan of rows [r0, r1) for query block [q0, q0+nqb): materialise score chunks + streaming select
static int dense_scan(ldot_index* ix, int64_t q0, int64_t nqb, int64_t nqb_pad, int64_t r0, int64_t r1, int kp,
                      float* tau, hipStream_t st) {
    const uint16_t* q16 = (const uint16_t*)ix->w_q16b.p + q0 * ix->ld16();   // (q0 is a multiple of 256: whole 16-row blocks)
    float* ls = (float*)ix->w_ls.p + q0 * kp;
    int32_t* li = (int32_t*)ix->w_li.p + q0 * kp;
    // (a scan shorter than a chunk — every Flickr / COCO sized index — gets score rows of its own length: contiguous 4-20 KB rows instead of
    // 128-KB strides, and dense_scan_all can then take all queries in one block)
    const int64_t chunk = std::min<int64_t>(ix->chunk_rows, round_up(r1 - r0, kBN));
    int rc = ix->w_S.ensure((size_t)nqb_pad * chunk * sizeof(float));
    if (rc) return rc;
    for (int64_t r = r0; r < r1; r += chunk) {
        const int64_t nrows = std::min(chunk, r1 - r);
        const int64_t nrows_pad = round_up(nrows, kBN);
        // algorithmic work: the VALID queries x rows x d (tile padding is overhead, not work)
        hipEvent_t ea, eb;
        prof_attach(ix, 2.0 * nqb * nrows * ix->d, (double)nrows * ix->d * 2 + (double)nqb * ix->d * 2 + (double)nqb * nrows * 4, &ea, &eb);
        rc = launch_score_dense(q16, ix->ld16(), nqb_pad, ix->x16b, ix->ld16(), r, nrows_pad, (int)ix->ld16(), (float*)ix->w_S.p,
                                chunk, nqb, st, 256, ea, eb);
        if (rc) return rc;
        rc = launch_select_dense((const float*)ix->w_S.p, chunk, nqb, nrows, r, ls, li, kp, tau ? tau + q0 : nullptr,
                                 st);
        if (rc) return rc;
        ix->stats[2] += nrows * nqb;
    }
    return LDOT_OK;
}

static bool narrow_ok(const ldot_index* ix, int64_t nq) {
    const int64_t qg = nq <= 16 ? 1 : nq <= 32 ? 2 : 4;   // groups of 16 queries whose operand blocks sit in LDS
    return nq <= kNarrowMaxQueries && ix->ld16() / 32 * qg <= kNarrowMaxLdsKiB;
}

// Few queries (one query tile) x many rows — the single-query serving shape (dvl/utils.py:204-211): one wide score
// launch over up to 4M rows (only the valid query rows are stored), a segmented select with (query, segment)
// parallelism and one merge.  HBM-bound: the index is streamed once.
static int dense_scan_wide(ldot_index* ix, int64_t nq, int64_t r0, int64_t r1, int kp, float* tau, hipStream_t st) {
    const int64_t wide = (int64_t)1 << 22;
    // segments of 16384 columns; a short scan (the warm-up of a few-query fused search) still gets ~16 segments in flight
    const int64_t seg_cols = std::max<int64_t>(1024, std::min<int64_t>(16384, round_up((r1 - r0 + 15) / 16, 256)));
    float* ls = (float*)ix->w_ls.p;
    int32_t* li = (int32_t*)ix->w_li.p;
    for (int64_t r = r0; r < r1; r += wide) {
        const int64_t nrows = std::min(wide, r1 - r), nrows_pad = round_up(nrows, kBN);
        const int64_t nseg = (nrows + seg_cols - 1) / seg_cols;
        int rc;
        if ((rc = ix->w_S.ensure((size_t)nq * nrows_pad * sizeof(float)))) return rc;
        if ((rc = ix->w_part_s.ensure((size_t)nseg * nq * kp * 4))) return rc;
        if ((rc = ix->w_part_l.ensure((size_t)nseg * nq * kp * 8))) return rc;
        prof_begin(ix, st, 2.0 * nq * nrows * ix->d,
                   (double)nrows * ix->d * 2 + (double)nq * ix->d * 2 + (double)nq * nrows * 4);
        if (narrow_ok(ix, nq))   // <= 64 queries: HBM-speed wave-per-group scan, no query-tile padding
            rc = launch_score_narrow(ix->w_q16b.p, ix->x16b, ix->ld16(), r, nrows, (float*)ix->w_S.p, nrows_pad, (int)nq, nullptr, 0,
                                     0, 0, st);
        else
            rc = launch_score_dense(ix->w_q16b.p, ix->ld16(), kBM, ix->x16b, ix->ld16(), r, nrows_pad, (int)ix->ld16(),
                                    (float*)ix->w_S.p, nrows_pad, nq, st);
        prof_end(ix, st);
        if (rc) return rc;
        float* ps = (float*)ix->w_part_s.p;
        int64_t* pl = (int64_t*)ix->w_part_l.p;
        if ((rc = launch_select_dense_parts((const float*)ix->w_S.p, nrows_pad, nq, nrows, seg_cols, r, kp, ps, pl, st)))
            return rc;
        // the segments' partial lists join the running list (earlier wide chunks) in one merge
        if ((rc = launch_merge_parts_into_lists(ps, pl, (int)nseg, nq, kp, ls, li, tau, st))) return rc;
        ix->stats[2] += nrows * nq;
    }
    return LDOT_OK;
}

constexpr int64_t kListsQueryChunk = 256;   // queries per pass of the run-maxima selection (bounds its buffers and flag array)

// buffers of the run-maxima selection (select_narrow.hip) for up to nq queries x ldm runs; M and the counters are kept all-zero
// between searches by the kernels themselves and cleared here only after a (re)allocation or an aborted search
static int narrow_buffers(ldot_index* ix, int64_t nq, int64_t ldm, hipStream_t st) {
    int rc;
    if (!ix->h_nover) {   // per-query "buffer full" flags: pinned host memory the final kernel writes directly
        LDOT_HIP_CHECK(hipHostMalloc((void**)&ix->h_nover, kListsQueryChunk * 4));
        LDOT_HIP_CHECK(hipHostGetDevicePointer((void**)&ix->d_nover, ix->h_nover, 0));
    }
    const size_t b_max = ix->w_nmax.bytes, b_cnt = ix->w_ncnt.bytes;
    if ((rc = ix->w_nmax.ensure((size_t)nq * ldm * 4))) return rc;
    if ((rc = ix->w_ntau.ensure((size_t)nq * 4))) return rc;
    if ((rc = ix->w_ncand.ensure((size_t)nq * kNarrowCandCap * 8))) return rc;
    if ((rc = ix->w_ncnt.ensure((size_t)nq * kNarrowCntStride * 4))) return rc;
    if (ix->w_nmax.bytes != b_max || ix->w_ncnt.bytes != b_cnt) ix->narrow_clean = false;
    if (!ix->narrow_clean) {
        LDOT_HIP_CHECK(hipMemsetAsync(ix->w_ncnt.p, 0, ix->w_ncnt.bytes, st));
        LDOT_HIP_CHECK(hipMemsetAsync(ix->w_nmax.p, 0, ix->w_nmax.bytes, st));
    }
    ix->narrow_clean = false;   // until the final kernel of this search has run
    return LDOT_OK;
}

// run size of a scan over nrows rows: <= 2048 run maxima per query (a coarser run lowers the threshold but adds hardly any candidates),
// <= 16384 when k' is large
static void narrow_plan(int64_t nrows, int kp, int* run_shift, int* nruns) {
    const int64_t ngroups = (nrows + 15) / 16, max_runs = kp <= 512 ? 2048 : kNarrowMaxRuns;
    int sh = 0;
    while (((ngroups - 1) >> sh) + 1 > max_runs) ++sh;
    *run_shift = sh;
    *nruns = (int)(((ngroups - 1) >> sh) + 1);
}

static bool narrow_select_ok(const ldot_index* ix, int64_t nq, int kp) {
    if (!narrow_ok(ix, nq) || kp > kNarrowCandCap / 4) return false;
    if (ix->ntotal <= kNarrowCandCap) return true;   // every row fits the candidate buffer
    // enough runs for a useful threshold: the k'-th largest of m run maxima admits ~ln(m / (m - k')) * m rows of an unordered index
    int sh, nruns;
    narrow_plan(std::min<int64_t>(ix->ntotal, (int64_t)1 << 22), kp, &sh, &nruns);
    return nruns >= (kp <= 512 ? 2 : 4) * (int64_t)kp;
}

// <= 64 queries against any number of rows (the serving shape): the index is streamed once at HBM speed (score_narrow.hip) and the
// lists are selected from the run maxima the scan leaves behind (select_narrow.hip) — convert + 4 kernels + re-score, no threshold
// to learn.  Speculative like the fused scan: a query whose candidate buffer filled up (thousands of equal scores in a run of rows) is
// flagged in device-mapped host memory (h_nover); the caller sees it at its synchronisation point and redoes the search with the
// streaming selector.
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

// <= 64 queries over one scan chunk: everything after the scan is ONE launch (narrow_finish_kernel, one workgroup per query)
static bool narrow_one_launch(const ldot_index* ix, int64_t nq, int kp) {
    int sh, nruns;
    narrow_plan(ix->ntotal, kp, &sh, &nruns);
    return nq <= kNarrowMaxQueries && ix->ntotal <= ((int64_t)1 << 22) && nruns <= 2048 && kp <= 512;
}

static int narrow_search(ldot_index* ix, int64_t nq, int kp, hipStream_t st, const DirectOut* direct = nullptr) {
    const int64_t wide = (int64_t)1 << 22;
    const int cap = kNarrowCandCap;
    int rc;
    if ((rc = narrow_buffers(ix, kNarrowMaxQueries, kNarrowMaxRuns, st))) return rc;
    uint32_t* M = (uint32_t*)ix->w_nmax.p;
    uint32_t* tk = (uint32_t*)ix->w_ntau.p;
    // <= 64 queries over one scan chunk: everything after the scan is ONE launch (narrow_finish_kernel, a workgroup per query: threshold,
    // collect, top-k', exact re-score, final order, output) instead of threshold + collect + final + re-score kernels — 38 -> ~12 us on the
    // GPU for one query
    {
        int sh, nruns;
        narrow_plan(ix->ntotal, kp, &sh, &nruns);
        if (narrow_one_launch(ix, nq, kp)) {
            const int64_t nrows = ix->ntotal, nrows_pad = round_up(nrows, 16);
            const float* qf = direct ? direct->qf32 : nullptr;   // (not staged: see DirectOut)
            const int qgroups = nq <= 16 ? 1 : nq <= 32 ? 2 : 4;   // S in 1-KiB tiles of 16 queries x 16 rows
            if ((rc = ix->w_S.ensure((size_t)qgroups * 16 * nrows_pad * sizeof(float)))) return rc;
            prof_begin(ix, st, 2.0 * nq * nrows * ix->d, (double)nrows * ix->d * 2 + (double)nq * ix->d * 2 + (double)nq * nrows * 4);
            rc = launch_score_narrow(ix->w_q16b.p, ix->x16b, ix->ld16(), 0, nrows, (float*)ix->w_S.p, nrows_pad, (int)nq, M,
                                     kNarrowMaxRuns, sh, 1, st, qf, qf ? direct->ldqf : 0, qf ? (int)direct->ldqf : 0);
            prof_end(ix, st);
            if (rc) return rc;
            if ((rc = launch_narrow_finish((const float*)ix->w_S.p, qgroups, nrows_pad, M, kNarrowMaxRuns, nruns, 16 << sh, nrows, (int)nq,
                                           qf ? qf : (const float*)ix->w_q32.p, qf ? direct->ldqf : ix->dpad, ix->x32, ix->dpad, ix->dpad, kp,
                                           direct ? direct->k : std::min(kp, 1), ix->rescore, (float*)ix->w_ls.p, (int32_t*)ix->w_li.p,
                                           (float*)ix->w_tau.p, direct ? direct->scores : nullptr, direct ? direct->labels : nullptr,
                                           ix->d_nover, nullptr, 0, nullptr, nullptr, 0, st)))
                return rc;
            ix->stats[2] += nrows * nq;
            ix->narrow_clean = true;
            ix->pend_done = direct != nullptr;
            if (ix->ntotal > 4096) {   // (kFinishCap candidates: an index that fits the buffer cannot fill it)
                ix->overflow_pending = true;
                ix->overflow_narrow = nq;
            }
            return LDOT_OK;
        }
    }
    for (int64_t r = 0; r < ix->ntotal; r += wide) {
        const int64_t nrows = std::min(wide, ix->ntotal - r), nrows_pad = round_up(nrows, 16);
        int sh, nruns;
        narrow_plan(nrows, kp, &sh, &nruns);
        const int qgroups = nq <= 16 ? 1 : nq <= 32 ? 2 : 4;   // S in 1-KiB tiles of 16 queries x 16 rows (coalesced stores)
        if ((rc = ix->w_S.ensure((size_t)qgroups * 16 * nrows_pad * sizeof(float)))) return rc;
        prof_begin(ix, st, 2.0 * nq * nrows * ix->d, (double)nrows * ix->d * 2 + (double)nq * ix->d * 2 + (double)nq * nrows * 4);
        rc = launch_score_narrow(ix->w_q16b.p, ix->x16b, ix->ld16(), r, nrows, (float*)ix->w_S.p, nrows_pad, (int)nq, M,
                                 kNarrowMaxRuns, sh, 1, st);
        prof_end(ix, st);
        if (rc) return rc;
        if ((rc = launch_narrow_tau(M, kNarrowMaxRuns, nruns, (int)nq, kp, tk, st))) return rc;
        if ((rc = launch_narrow_collect((const float*)ix->w_S.p, nrows_pad, M, kNarrowMaxRuns, nruns, 16 << sh, nrows, r, (int)nq, tk,
                                        (uint64_t*)ix->w_ncand.p, cap, (int32_t*)ix->w_ncnt.p, nullptr, 0, qgroups, st)))
            return rc;
        ix->stats[2] += nrows * nq;
    }
    if ((rc = launch_narrow_final((const uint64_t*)ix->w_ncand.p, cap, (int32_t*)ix->w_ncnt.p, (int)nq, (float*)ix->w_ls.p,
                                  (int32_t*)ix->w_li.p, kp, (float*)ix->w_tau.p, ix->d_nover, st)))
        return rc;
    ix->narrow_clean = true;
    if (ix->ntotal > cap) {   // (an index that fits the candidate buffer cannot fill it)
        ix->overflow_pending = true;
        ix->overflow_narrow = nq;
    }
    return LDOT_OK;
}

// (tau is always maintained: the fused scan continues from it, a sharded search exchanges it)
static int dense_scan_all(ldot_index* ix, int64_t nq, int64_t r0, int64_t r1, int kp, float* tau, bool allow_wide,
                          hipStream_t st, int64_t q_base = 0) {
    if (allow_wide && q_base == 0 && nq <= kBM && (r1 - r0 > 2 * ix->chunk_rows || (nq <= 64 && r1 - r0 >= 2048)))
        return dense_scan_wide(ix, nq, r0, r1, kp, tau, st);
    // query blocks bound the dense score workspace (<= ~2 GiB)
    const int64_t chunk = std::min<int64_t>(ix->chunk_rows, round_up(r1 - r0, kBN));
    const int64_t qb_max = std::max<int64_t>(kBM, ((int64_t)1 << 29) / chunk / kBM * kBM);
    for (int64_t q0 = 0; q0 < nq; q0 += qb_max) {
        const int64_t nqb = std::min(qb_max, nq - q0);
        int rc = dense_scan(ix, q_base + q0, nqb, round_up(nqb, kBM), r0, r1, kp, tau, st);
        if (rc) return rc;
    }
    return LDOT_OK;
}

// fused scan: dense warm-up of the first rows (gives every query a full list and a threshold), then
// geometrically growing fused-filter launches, each followed by the pool select that raises the thresholds.
// Queries are processed in chunks of kFusedQueryChunk (bounds the candidate pools: 196 KiB per query at 256 sub-pools of 16 records).
constexpr int64_t kFusedQueryChunk = 16384;
constexpr int64_t kFewSelectMaxQueries = 256;   // one query block: sub-pools folded by 16 waves per query + one merge
constexpr int64_t kFewBlockGrowthPct = 1600;   // launch growth with ONE query block (65 .. 256 queries): 2-3 % faster than growing
                                               // straight to the pool bound (tools/fewgrowth_sweep.py: 0.657 vs 0.675 ms at 100 queries)

// one fused-filter launch over index rows [r, r + len) for the queries [q0, q0 + nq) + the pool select that folds its records into
// the running lists and raises the thresholds
static int fused_launch_and_select(ldot_index* ix, int64_t q0, int64_t nq, int64_t nq_pad, int kp, int64_t r, int64_t len,
                                   hipStream_t st, float* tau_opt = nullptr, int opt_m_next = 0, int64_t scramble_tiles = 0) {
    float* tau = (float*)ix->w_tau.p + q0;
    // (the optimistic scan filters with w_tau_opt; the selects keep the guaranteed w_tau and refresh w_tau_opt for the next launch)
    const float* filter_tau = tau_opt ? tau_opt : tau;
    float* ls = (float*)ix->w_ls.p + q0 * kp;
    int32_t* li = (int32_t*)ix->w_li.p + q0 * kp;
    const uint16_t* q16 = (const uint16_t*)ix->w_q16b.p + q0 * ix->ld16();
    int32_t* over = (int32_t*)ix->w_over.p + q0;
    const int qg = fused_query_group(nq_pad);
    const int64_t nslices = 256 / qg, nsubs = kPoolSubsPerSlice * nslices;
    int rc;
    hipEvent_t ea, eb;
    prof_attach(ix, 2.0 * nq * len * ix->d, (double)len * ix->d * 2 + (double)nq * ix->d * 2 + (double)nq * kp * 8, &ea, &eb);
    // (scrambled scan: rows [r, r + len) of the pseudo-random tile order of the whole index)
    if ((rc = ix->w_cur_save.ensure(fused_cursor_save_bytes(nq_pad)))) return rc;   // (cursors of the chunk-major unit order: score_filter.hip)
    rc = launch_score_filter(ix->x16b, ix->ld16(), scramble_tiles ? 0 : r, len, q16, ix->ld16(), nq_pad, (int)ix->ld16(), filter_tau,
                             (uint4*)ix->w_pool.p, (int32_t*)ix->w_pool_cnt.p, st, scramble_tiles, scramble_tiles ? r / fused_tile_rows() : 0,
                             ea, eb, ix->w_cur_save.p);
    if (rc) return rc;
    if (nq <= kFewSelectMaxQueries && nsubs >= 128 * kPoolSubsPerSlice && kp + 512 + 32 <= 1024) {
        // few queries: G waves per query fold the sub-pools into partial lists, one merge joins them with the running list
        const int G = 16;
        if ((rc = ix->w_part_s.ensure((size_t)G * nq * kp * 4))) return rc;
        if ((rc = ix->w_part_l.ensure((size_t)G * nq * kp * 8))) return rc;
        float* ps = (float*)ix->w_part_s.p;
        int64_t* pl = (int64_t*)ix->w_part_l.p;
        if ((rc = launch_select_pools_parts((const uint4*)ix->w_pool.p, (const int32_t*)ix->w_pool_cnt.p, (int)nsubs, nq, G,
                                            (int32_t)ix->ntotal, kp, tau, ps, pl, over, (int32_t*)ix->w_over_sum.p,
                                            (int32_t*)ix->w_qcnt.p + q0, st)))
            return rc;
        return launch_merge_parts_into_lists(ps, pl, G, nq, kp, ls, li, tau, st);
    }
    return launch_select_pools((const uint4*)ix->w_pool.p, (const int32_t*)ix->w_pool_cnt.p, (int)nsubs, nq, (int32_t)ix->ntotal, ls,
                               li, kp, tau, over, (int32_t*)ix->w_over_sum.p, (int32_t*)ix->w_qcnt.p + q0, st, tau_opt, opt_m_next);
}

// candidate pools + counters for nq_pad queries (the counters are all-zero between searches)
static int fused_pools(ldot_index* ix, int64_t nq_pad, hipStream_t st) {
    const int qg = fused_query_group(nq_pad);
    const int64_t nsubs = kPoolSubsPerSlice * (256 / qg);
    int rc;
    if ((rc = ix->w_pool.ensure((size_t)nq_pad * nsubs * kPoolCap * kPoolRecBytes))) return rc;
    const size_t cnt_bytes = (size_t)nq_pad * nsubs * 4;
    if (cnt_bytes > ix->w_pool_cnt.bytes) ix->pools_clean = false;
    if ((rc = ix->w_pool_cnt.ensure(cnt_bytes))) return rc;
    if (!ix->pools_clean) LDOT_HIP_CHECK(hipMemsetAsync(ix->w_pool_cnt.p, 0, ix->w_pool_cnt.bytes, st));
    ix->pools_clean = false;   // until the scan that uses them has completed
    return LDOT_OK;
}

// Thresholds the shards of a sharded search agree on after their warm-ups (max over the shards of the k'-th best, min of the
// ceil(k'/parts)-th best: ldot.h) are worth fewer scanned rows than parts x warm: the minimum over `parts` noisy order statistics sits
// ~1.4 sigma low (measured: 640 admitted records per query where k' x 120904 / 32768 = 472 were expected, tools/shard_floor.py).  The
// pool bound of a launch on agreed thresholds counts them at 70 % and allows an expectation of 3 records per sub-pool instead of 4 (the
// bound IS the active limit there: P(Poisson(3) > 16) ~ 1e-8 per sub-pool against 4e-7 at 4, times 2.6 M sub-pools per search; an
// overflow costs the flagged queries one more launch).  Measured at 8 x 125 000 rows: 2.5 records per sub-pool, one launch per shard.
constexpr int64_t kAgreedWorthPct = 70, kAgreedFill = 3;
constexpr int64_t kWarmSelectFastCols = 5120;   // the warm-up's select keeps a row in registers up to here (select_dense_runs_kernel)

// rows of the dense warm-up of a fused scan
static int64_t fused_warm_rows(const ldot_index* ix, int64_t nq, int64_t nq_pad, int kp) {
    // Pool sizing rule: a launch over `len` rows after `r` scanned rows admits ~kp*len/r candidates per query,
    // spread over nsubs sub-pools of kPoolCap records (four lane groups of kPoolGroupCap each).  Keeping the expectation <= kPoolFill per sub-pool
    // (16 records, expectation 4: overflow probability ~1e-6 per sub-pool and launch WHERE THIS BOUND IS THE ACTIVE ONE, i.e. for k' in the
    // thousands; at the default growth of 150 % and k' = 128 the expectation is 0.75 and the probability ~1e-17; an overflow costs the
    // flagged queries one more fused launch, redo_flagged) bounds len <= r * kFill * nsubs / kp (1024 r / kp at 256 sub-pools; 8x that for the
    // 2048 sub-pools of a single query block, whose search is then ONE fused launch); the smallest launch is one tile
    // per row slice, hence the warm-up covers at least bm * nslices * kp / (kFill * nsubs) rows.
    constexpr int64_t kFill = kPoolFill;
    const int64_t bm = fused_tile_rows();
    const int qg = fused_query_group(nq_pad);
    const int64_t nslices = 256 / qg, nsubs = kPoolSubsPerSlice * nslices;
    // (a shard scanning on statistics pooled over the whole index takes its first threshold against the GLOBAL row count: a longer warm-up
    // buys it little, and its dense rows cost ~10x fused ones — the pool bound's minimum, 3072 rows, instead of 4096.  While such a shard ran
    // to its pool bound in one launch this overflowed a candidate pool for a few queries on half of the shards (a local redo, 2.14 vs 1.94 ms
    // for the slowest rank); with the scan split at 12x the rows seen (fused_rest_chunk_optimistic) no shard overflows and every rank gains:
    // 1.78 -> 1.75 ms at 8 x 125 000 rows, profiles/r05_shard_warm_probe_growth12.txt)
    const int64_t want = (ix->pool_total > 0 && !ix->warm_rows_set) ? 2048 : ix->warm_rows;
    int64_t warm = std::max<int64_t>(want, round_up(bm * nslices * (int64_t)kp / (kFill * nsubs), 256));
    // few queries (serving): launches and selects cost more than dense rows -> warm up over just enough rows for ONE fused launch
    // to cover the rest within the pool bound (len <= r * kFill * nsubs / kp)
    if (nq <= 64) warm = std::max(warm, round_up(ix->ntotal * kp / (kp + kFill * nsubs) + 1, 256));
    // sharded search: thresholds agreed after the warm-ups of `parts` shards are worth ~kAgreedWorthPct of parts x warm scanned rows
    // (fused_rest_chunk); a warm-up long enough for ONE launch to cover the rest of the shard within the pool bound saves a launch and
    // a pool select — as long as the warm-up's select stays on its fast path (dense rows cost ~10x fused ones, and the streaming select
    // of a longer warm-up 3x the register one: 211 vs 71 us at 10 000 queries, profiles/r04_shard_timeline_*.txt)
    if (ix->cur_parts > 1 && nq > 64) {
        const int64_t per_row = kAgreedWorthPct * ix->cur_parts * kAgreedFill * nsubs / (100 * kp);   // rows one launch may cover per warm-up row
        warm = std::max(warm, std::min<int64_t>(std::max(warm, kWarmSelectFastCols), round_up(ix->ntotal / (per_row + 1) + 1, 256)));
    }
    return std::min(ix->ntotal, warm);
}

// large batches of a plain search (and shards on pooled statistics) filter with optimistic thresholds; few-query searches have their
// own launch schedule, shards of the agreed-threshold exchange their agreed thresholds
static bool optimistic_scan(const ldot_index* ix, int64_t nq, int parts) {
    bool opt_on = ix->optimistic && ix->opt_backoff == 0;
#ifdef LDOT_ABLATION
    if (getenv("LDOT_DEBUG_NOOPT")) opt_on = false;   // (the kernel ablation variants produce no candidates: the end-of-scan check would redo every query)
#endif
    return opt_on && parts == 1 && ix->cur_parts == 1 && nq > kFewSelectMaxQueries;
}

// Scrambled scan order (LDOT_OPT_SCAN_ORDER).  The optimistic thresholds assume that the rows scanned so far are a fair sample of the
// index.  Rows stored in an order that correlates with the queries (sorted by cluster, by class, by source) break that — and the pool
// bound of the guaranteed thresholds with it: a query's best rows arrive together.  The remedy is to scan in an order that does not
// follow the storage order: the fused launches visit the 384-row tiles of the WHOLE index in a fixed pseudo-random order (tile j of the
// order = tile (j x mul) mod T, score_filter.hip) and the warm-up scores a SPREAD sample (every T/16-th 256-row tile) that only yields
// the first thresholds: its rows are scanned again with everybody else, the lists start empty.
static bool scrambled_scan(const ldot_index* ix, int64_t nq, int parts, int kp) {
    return optimistic_scan(ix, nq, parts) && (ix->scan_order == 2 || (ix->scan_order == 0 && ix->scrambled_auto)) &&
           ix->ntotal >= 8 * fused_warm_rows(ix, nq, round_up(nq, kBM), kp);
}

static int fused_warm_chunk(ldot_index* ix, int64_t q0, int64_t nq, int64_t nq_pad, int kp, int parts, hipStream_t st) {
    // (q0 is a multiple of 256: whole 16-row blocks of the query shadow)
    const int64_t warm = fused_warm_rows(ix, nq, nq_pad, kp);
    if (!scrambled_scan(ix, nq, parts, kp)) return dense_scan_all(ix, nq, 0, warm, kp, (float*)ix->w_tau.p, nq <= 64, st, q0);
    // spread sample: `warm` rows in 256-row tiles at equal distances over the index, scored and selected like a contiguous chunk (the
    // labels the select writes are column numbers: the list is only read by the first tau_opt and then cleared)
    const int64_t wpad = round_up(warm, kBN), tiles = wpad / kBN;
    const int64_t stride = tiles > 1 ? (ix->ntotal - kBN) / (tiles - 1) / 16 * 16 : kBN;
    const uint16_t* q16 = (const uint16_t*)ix->w_q16b.p + q0 * ix->ld16();
    int rc = ix->w_S.ensure((size_t)nq_pad * wpad * sizeof(float));
    if (rc) return rc;
    hipEvent_t ea, eb;
    prof_attach(ix, 2.0 * nq * wpad * ix->d, (double)wpad * ix->d * 2 + (double)nq * ix->d * 2 + (double)nq * wpad * 4, &ea, &eb);
    rc = launch_score_dense(q16, ix->ld16(), nq_pad, ix->x16b, ix->ld16(), 0, wpad, (int)ix->ld16(), (float*)ix->w_S.p, wpad, nq, st, stride, ea, eb);
    if (rc) return rc;
    ix->stats[2] += wpad * nq;
    return launch_select_dense((const float*)ix->w_S.p, wpad, nq, wpad, 0, (float*)ix->w_ls.p + q0 * kp, (int32_t*)ix->w_li.p + q0 * kp, kp,
                               (float*)ix->w_tau.p + q0, st);
}

// ---- optimistic thresholds (round 4) ---------------------------------------------------------------------------------------------------
// The guaranteed threshold of a query — the k'-th best score among the r rows scanned so far — admits k' / r of the following rows:
// k' ln(N / warm) ~ 700 records per query over a 1M-row scan in the limit of continuous refresh, ~1150 with six launches, a third of
// them in the first launch.  But the FINAL threshold is known in distribution long before: if the rows are exchangeable (no order in the
// index that correlates with the query), the number of the index's k' best rows among the first r is Poisson(k' r / N), so the m-th best
// score seen so far is BELOW the final k'-th best with probability 1 - P(Poisson(k' r / N) >= m).  The scan therefore filters with
// tau_opt = the m(r)-th best so far, m(r) = the smallest m with P(Poisson(k' r / N) >= m) <= kOptEps (8 at r = 4096 of 1M rows, 18 at
// 28 672, 60 at 225 280, k' from ~620 000 on): ~350 records per query in FOUR launches (each as long as the pool bound and the
// launch-length knee of DESIGN 5.2b allow) instead of ~1150 in six.
// It stays exact without the assumption: every row was admitted iff it scored >= the tau_opt in force, so a query whose final list holds
// k' rows at or above its last (largest) tau_opt has lost nothing that belongs to its top k' — verify_tau_opt_kernel checks exactly that
// and flags the others (rows stored in an order that front-loads a query's best rows, e.g. its own cluster first), which redo_flagged
// searches again on guaranteed thresholds like pool overflows.  The guaranteed threshold w_tau (k'-th best of the admitted rows, a lower
// bound of the k'-th best of all rows seen) keeps being maintained by the selects: the recovery, LDOT_OPT_VERIFY and the sharded
// exchange use it.
constexpr double kOptEps = 1e-7;                 // per query and launch; 10 000 queries x 4 launches: one redo in ~250 searches
constexpr int64_t kOptMaxLaunchRows = 393216;    // launches beyond ~0.6 GB of rows run slower per row
constexpr int64_t kOptGrowthX = 7;               // a launch covers up to 7x the rows already scanned
constexpr int64_t kPooledGrowthX = 12;           // ... a shard on pooled statistics up to 12x (sweep: 10 .. 16 level, 3 .. 8 and one launch slower)

static int optimistic_m(int kp, int64_t r, int64_t n, double eps) {
    const double x = (double)kp * (double)r / (double)n;
    double term = exp(-x), cdf = 0.0;            // P(Poisson(x) < m), accumulated term by term
    for (int m = 1; m < kp; ++m) {
        cdf += term;                             // now cdf = P(Poisson < m)
        if (1.0 - cdf <= eps) return m;
        term *= x / m;
    }
    return kp;
}

static int fused_rest_chunk_optimistic(ldot_index* ix, int64_t q0, int64_t nq, int64_t nq_pad, int kp, hipStream_t st) {
    int rc;
    constexpr int64_t kFill = kPoolFill;
    const int64_t bm = fused_tile_rows();
    const int qg = fused_query_group(nq_pad);
    const int64_t nslices = 256 / qg, nsubs = kPoolSubsPerSlice * nslices, unit = bm * nslices;
    const int64_t warm = fused_warm_rows(ix, nq, nq_pad, kp), N = ix->ntotal;
    if (warm >= N) return LDOT_OK;
    // A shard of a sharded search (pool_total > 0) takes its order statistics against the WHOLE index: of the global k' best rows
    // Poisson(k' r / N_global) lie among this shard's first r rows, so its m(r)-th best is below the GLOBAL k'-th best w.h.p. — a
    // threshold 1 / parts as selective as the shard's own k'-th best, ~k' / parts + a margin admitted rows per query instead of k'
    // ln(..), and ONE launch after the warm-up.  This shard alone cannot check it (its list need not hold k' rows above the threshold):
    // the last select does not verify, the threshold is published as the level above which the list is complete and the ranks decide
    // together (ldot_shard_floor).
    const bool pooled = ix->pool_total > 0;
    const int64_t Ng = pooled ? std::max(ix->pool_total, N) : N;
    ix->pooled_used = pooled;
    const bool scr = scrambled_scan(ix, nq, 1, kp);
    ix->scrambled_now = scr;
    const int64_t T = (N + bm - 1) / bm, Nscan = scr ? T * bm : N;   // (scrambled: every tile of the index, the warm-up's rows included)
    if ((rc = fused_pools(ix, nq_pad, st))) return rc;
    float* tau = (float*)ix->w_tau.p + q0;
    float* tau_opt = (float*)ix->w_tau_opt.p + q0;
    const float* ls = (const float*)ix->w_ls.p + q0 * kp;
    const int32_t* li = (const int32_t*)ix->w_li.p + q0 * kp;
    double eps = pooled ? kOptEps / ix->pool_parts : kOptEps;   // (the floor check fails if ANY shard aimed too high)
    // (pooled statistics: a launch covers up to 12x the rows the thresholds were drawn from.  Up to round 5 a shard ran to its pool bound in ONE
    // launch after the warm-up; at 8 x 125 000 rows that launch admits 274 records per query on the 4096-row threshold and one select folds
    // them all: a 49 152-row launch first, its select, then the rest takes rank 0 from 2.00 to 1.78 ms (144 records per query).  Shards of
    // 250 000 / 500 000 rows already split at the pool bound and are unchanged: profiles/r05_shard_growth_sweep.txt)
    int64_t growth_x = pooled ? kPooledGrowthX : kOptGrowthX, max_rows = kOptMaxLaunchRows;
#ifdef LDOT_ABLATION
    if (const char* e = getenv("LDOT_DEBUG_OPT_EPS")) eps = atof(e);
    if (const char* e = getenv("LDOT_DEBUG_OPT_GROWTHX")) growth_x = atoll(e);
    if (const char* e = getenv("LDOT_DEBUG_OPT_MAXROWS")) max_rows = atoll(e);
#endif
    int64_t r = scr ? 0 : warm;   // rows scanned by the fused launches so far (scrambled: in the pseudo-random tile order, from its start)
    // the first thresholds come from the warm-up's list; every pool select then leaves the next launch's behind (and the last one checks)
    if ((rc = launch_tau_opt(ls, li, kp, nq, optimistic_m(kp, warm, Ng, eps), tau, tau_opt, st))) return rc;
    if (scr && (rc = launch_init_lists((float*)ix->w_ls.p + q0 * kp, (int32_t*)ix->w_li.p + q0 * kp, nq_pad * kp, tau, nq, nq_pad, st)))
        return rc;   // (the spread sample's rows come again with the scan: the lists start empty)
    while (r < Nscan) {
        const int64_t seen = std::max(r, warm);   // the rows the thresholds in force were drawn from
        const int m = optimistic_m(kp, seen, Ng, eps);
        // expected records per query of a launch over len rows: len m / r, kept <= kFill per sub-pool like the guaranteed schedule's bound
        // (pooled statistics run AT this bound, and the m-th best of a few thousand rows is a noisy quantile — some queries admit
        // 1.5x the expectation —: half the fill there)
        int64_t len = std::min<int64_t>(std::min<int64_t>(seen * growth_x, max_rows), seen * (pooled ? kFill / 2 : kFill) * nsubs / m);
        len = std::max<int64_t>(len / unit * unit, unit);
        len = std::min(len, Nscan - r);
        if (Nscan - r - len < len / 4 && Nscan - r <= max_rows + 2 * unit) len = Nscan - r;   // no short tail launch
        const int m_next = r + len < Nscan ? optimistic_m(kp, r + len, Ng, eps) : pooled ? -1 : 0;   // (0: the last select verifies)
        if ((rc = fused_launch_and_select(ix, q0, nq, nq_pad, kp, r, len, st, tau_opt, m_next, scr ? T : 0))) return rc;
        ix->stats[3] += std::min(len, N - std::min(r, N)) * nq;
        r += len;
    }
    ix->pools_clean = true;
    return LDOT_OK;
}

// the fused launches after the warm-up.  parts > 1 (sharded search): the thresholds were raised to a bound the `parts` ranks agreed on
// after their warm-ups (ldot_index_search_scan) — it is worth about parts x warm scanned rows, so the pool bound allows that much longer
// launches, and one long launch on it beats two that each pay a pool select.
static int fused_rest_chunk(ldot_index* ix, int64_t q0, int64_t nq, int64_t nq_pad, int kp, int parts, hipStream_t st) {
    // large batches of a plain search: optimistic thresholds (few-query searches have their own launch schedule, sharded searches
    // their agreed thresholds)
    if (optimistic_scan(ix, nq, parts)) {
        ix->opt_used = true;
        return fused_rest_chunk_optimistic(ix, q0, nq, nq_pad, kp, st);
    }
    int rc;
    constexpr int64_t kFill = kPoolFill;
    const int64_t bm = fused_tile_rows();
    const int qg = fused_query_group(nq_pad);
    const int64_t nslices = 256 / qg, nsubs = kPoolSubsPerSlice * nslices;
    const int64_t warm = fused_warm_rows(ix, nq, nq_pad, kp);
    if (warm >= ix->ntotal) return LDOT_OK;
    if ((rc = fused_pools(ix, nq_pad, st))) return rc;
    // (the pad queries' thresholds are +inf since init_lists: they never produce candidates)
    // few query blocks: admissions are cheap, launches are not -> let the launch length grow up to the pool bound
    int64_t few_growth = kFewBlockGrowthPct;
#ifdef LDOT_ABLATION
    if (const char* e = getenv("LDOT_DEBUG_FEWGROWTH")) few_growth = atoll(e);
#endif
    int64_t growth = nq_pad <= kBM ? std::max<int64_t>(ix->growth_pct, few_growth) : ix->growth_pct;
    if (parts > 1) growth = std::max<int64_t>(growth, kFewBlockGrowthPct);
    const int64_t r_agreed = parts > 1 ? warm * parts * kAgreedWorthPct / 100 : 0;   // what the agreed thresholds are worth, in scanned rows
    int64_t r = warm;
    while (r < ix->ntotal) {
        const bool agreed = r_agreed > r;
        const int64_t r_eff = agreed ? r_agreed : r;
        int64_t len = std::min<int64_t>(r_eff * growth / 100, r_eff * (agreed ? kAgreedFill : kFill) * nsubs / kp);
        len = std::max<int64_t>(len, bm * nslices);
        // whole tiles for every row slice (a launch is as slow as its busiest slice); rounding DOWN keeps the pool bound
        len = len / (bm * nslices) * (bm * nslices);
        len = std::min(len, ix->ntotal - r);
        if (ix->ntotal - r - len < len / 4) len = ix->ntotal - r;   // no short tail launch (the pool bound has that slack)
#ifdef LDOT_ABLATION
        // LDOT_DEBUG_MAXLEN: cap on the rows of one launch (experiment: launches whose row range fits the 256 MB Infinity Cache)
        if (const char* e = getenv("LDOT_DEBUG_MAXLEN")) {
            const int64_t cap = atoll(e) / (bm * nslices) * (bm * nslices);
            if (cap > 0 && len > cap) len = cap;
        }
#endif
        if ((rc = fused_launch_and_select(ix, q0, nq, nq_pad, kp, r, len, st))) return rc;
        ix->stats[3] += len * nq;
        r += len;
    }
    ix->pools_clean = true;   // the pool selects reset every counter they read
    return LDOT_OK;
}

// Enqueues the whole fused scan WITHOUT synchronising: whether a lane-private pool overflowed (adversarial row orders) is
// summarised in w_over_sum; fused_overflow_check() fetches it (4 bytes into pinned memory) when the caller has to wait anyway.
// phase 0: the whole scan; 1: set-up + the dense warm-ups only; 2: the fused launches of a scan whose phase 1 has run (sharded search:
// the ranks exchange thresholds in between, ldot_index_search_warmup / _scan)
static int fused_scan(ldot_index* ix, int64_t nq, int64_t nq_pad, int kp, hipStream_t st, int phase = 0, int parts = 1) {
    int rc;
    if (phase != 2) {
        const size_t over_bytes = (size_t)nq_pad * 4;
        const bool fresh_flags = over_bytes > ix->w_over.bytes;
        if ((rc = ix->w_over.ensure(over_bytes))) return rc;
        if ((rc = ix->w_over_sum.ensure(16))) return rc;
        if ((rc = ix->w_qcnt.ensure((size_t)nq_pad * 4))) return rc;
        if ((rc = ix->w_tau_opt.ensure((size_t)nq_pad * 4))) return rc;
        ix->opt_used = false;
        ix->pooled_used = false;
        ix->scrambled_now = false;
        ix->opt_nq = nq;
        if (ix->opt_backoff > 0 && nq > kFewSelectMaxQueries) --ix->opt_backoff;   // (counted in large-batch searches, the ones it applies to)
        if ((rc = launch_init_fused_scan((float*)ix->w_tau_opt.p, (int32_t*)ix->w_qcnt.p, (int32_t*)ix->w_over_sum.p, nq, nq_pad, st))) return rc;
        ix->qcnt_n = nq;
        if (!ix->h_over_sum) LDOT_HIP_CHECK(hipHostMalloc((void**)&ix->h_over_sum, 16));
        if (fresh_flags || !ix->flags_clean) LDOT_HIP_CHECK(hipMemsetAsync(ix->w_over.p, 0, ix->w_over.bytes, st));
        ix->flags_clean = false;
    }
    for (int64_t q0 = 0; q0 < nq; q0 += kFusedQueryChunk) {
        const int64_t nqc = std::min(kFusedQueryChunk, nq - q0);
        if (phase != 2 && (rc = fused_warm_chunk(ix, q0, nqc, round_up(nqc, kBM), kp, phase == 0 ? 1 : 0, st))) return rc;
        if (phase == 0 && (rc = fused_rest_chunk(ix, q0, nqc, round_up(nqc, kBM), kp, 1, st))) return rc;
    }
    if (phase == 1) return LDOT_OK;
    if (phase == 2)
        for (int64_t q0 = 0; q0 < nq; q0 += kFusedQueryChunk) {
            const int64_t nqc = std::min(kFusedQueryChunk, nq - q0);
            if ((rc = fused_rest_chunk(ix, q0, nqc, round_up(nqc, kBM), kp, parts, st))) return rc;
        }
    LDOT_HIP_CHECK(hipMemcpyAsync(ix->h_over_sum, ix->w_over_sum.p, 4, hipMemcpyDeviceToHost, st));
    ix->overflow_pending = true;
    return LDOT_OK;
}

// after a synchronisation point of `st`: did the last fused scan overflow?  (adversarial row order -> the caller redoes the
// search with the always-correct dense path)
static bool fused_overflow_check(ldot_index* ix) {
    if (!ix->overflow_pending) return false;
    ix->overflow_pending = false;
    ix->overflow_was_narrow = ix->overflow_narrow > 0;
    if (ix->overflow_narrow > 0) {   // narrow search: flags written by its final kernel
        int64_t n = 0;
        for (int64_t q = 0; q < ix->overflow_narrow; ++q) n += ix->h_nover[q];
        ix->overflow_narrow = 0;
        ix->stats[1] = n;
        if (n > 0) {
            ix->narrow_backoff = ix->narrow_penalty;
            ix->narrow_penalty = std::min(2 * ix->narrow_penalty, 1024);
        } else {
            ix->narrow_penalty = 16;
        }
        return n > 0;
    }
    const int64_t n_over = ix->h_over_sum[0];
    ix->stats[1] = n_over;
    ix->flags_clean = n_over == 0;
    if (ix->opt_used) {
        ix->opt_used = false;
        // With rows in a fair order a query fails the check once in ~1e7 launches: a search in which one query in a thousand fails says
        // that the storage order is not a fair sample order — the index scans in the scrambled order from then on.  Failures that
        // persist (or come with the scrambled order: scores that bf16 cannot tell apart, thousands of equal rows) at more than 1 / 64
        // of the queries cost more than the optimistic thresholds save: back off to the guaranteed ones for a while.
        if (ix->scan_order == 0 && !ix->scrambled_auto && !ix->scrambled_now && n_over >= 4 && n_over * 1024 > ix->opt_nq) {
            ix->scrambled_auto = true;
        } else if (n_over * 64 > ix->opt_nq && ix->scrambled_now && ix->row_shuffle == 0 && !ix->shuffled && !ix->reshuffled) {
            // failing in the scrambled TILE order too: similar rows sit in runs about as long as a tile.  The store is re-shuffled row by
            // row before the next search (once per index; LDOT_OPT_ROW_SHUFFLE)
            ix->want_reshuffle = true;
        } else if (n_over * 64 > ix->opt_nq) {
            ix->opt_backoff = ix->opt_penalty;
            ix->opt_penalty = std::min(2 * ix->opt_penalty, 1024);
        } else if (n_over == 0) {
            ix->opt_penalty = 16;
        }
    }
    return n_over > 0;
}

static int dense_redo(ldot_index* ix, int64_t nq, int64_t nq_pad, int kp, hipStream_t st) {
    float* tau = (float*)ix->w_tau.p;
    int rc;
    if ((rc = launch_init_lists((float*)ix->w_ls.p, (int32_t*)ix->w_li.p, nq_pad * kp, tau, nq, nq_pad, st))) return rc;
    ix->redone += nq;
    return dense_scan_all(ix, nq, 0, ix->ntotal, kp, tau, true, st);
}

// Recovery after a fused scan in which some queries' lane-private pools overflowed (row orders that concentrate a query's best rows
// in few tiles: cluster-sorted rows, the adversarial ramp).  ONLY the flagged queries are searched again, and cheaply: a dropped record
// can only have LOWERED a query's threshold, so the threshold the first pass ended with is still a valid lower bound of its final k'-th
// score — and usually a close one.  Level 0: the flagged queries are compacted into a batch of their own and scanned once more over ALL
// rows in ONE fused launch with those thresholds: hardly more than their true top-k' rows are admitted, so the pools hold.  Level 1:
// queries that overflow even then (rows in ascending score order: the dropped records were the BEST ones and the threshold is far too
// low) are compacted again and take the always-correct dense path.  `st` is synchronised.
static int redo_flagged(ldot_index* ix, int64_t nq, int64_t nq_pad, int kp, hipStream_t st, int level = 0) {
    if (level == 0 && ix->overflow_was_narrow) return dense_redo(ix, nq, nq_pad, kp, st);   // (<= 64 queries: the streaming selector is cheap)
    int rc;
    std::vector<int32_t> flags((size_t)nq), fidx;
    LDOT_HIP_CHECK(hipMemcpyAsync(flags.data(), ix->w_over.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    LDOT_HIP_CHECK(hipStreamSynchronize(st));
    for (int64_t q = 0; q < nq; ++q)
        if (flags[(size_t)q]) fidx.push_back((int32_t)q);
    const int64_t nf = (int64_t)fidx.size(), nf_pad = round_up(std::max<int64_t>(nf, 1), kBM);
    if (nf == 0) return LDOT_OK;
    if (level == 0) ix->redone += nf;
    if (level == 0 && ix->pooled_used) {   // (their lists will be complete: the shard statistics must not report the pooled level for them)
        if ((rc = ix->w_redone.ensure((size_t)nq * 4))) return rc;
        LDOT_HIP_CHECK(hipMemcpyAsync(ix->w_redone.p, ix->w_over.p, (size_t)nq * 4, hipMemcpyDeviceToDevice, st));
    }
    ldot_index::Compact& c = ix->compact[level];
    if ((rc = c.fidx.ensure((size_t)nf * 4))) return rc;
    if ((rc = c.q32.ensure((size_t)nf_pad * ix->dpad * 4))) return rc;
    if ((rc = c.q16b.ensure((size_t)nf_pad * ix->ld16() * 2))) return rc;
    if ((rc = c.ls.ensure((size_t)nf_pad * kp * 4))) return rc;
    if ((rc = c.li.ensure((size_t)nf_pad * kp * 4))) return rc;
    if ((rc = c.tau.ensure((size_t)nf_pad * 4))) return rc;
    const int32_t* didx = (const int32_t*)c.fidx.p;
    LDOT_HIP_CHECK(hipMemcpyAsync(c.fidx.p, fidx.data(), (size_t)nf * 4, hipMemcpyHostToDevice, st));
    if ((rc = launch_gather_rows_f32((const float*)ix->w_q32.p, ix->dpad, didx, nf, nf_pad, (float*)c.q32.p, st))) return rc;
    if ((rc = launch_convert_rows(c.q32.p, LDOT_F32, ix->dpad, nf, nf_pad, ix->d, ix->dpad, 0, nullptr, nullptr, ix->precision ? 2 : 0,
                                  (uint16_t*)c.q16b.p, 0, st)))
        return rc;
    if ((rc = launch_init_lists((float*)c.ls.p, (int32_t*)c.li.p, nf_pad * kp, (float*)c.tau.p, nf, nf_pad, st))) return rc;
    if (level == 0 && (rc = launch_gather_tau((const float*)ix->w_tau.p, didx, nf, (const float*)ix->w_q32.p, ix->dpad, ix->d,
                                              (const float*)ix->w_norm.p, (float*)c.tau.p, st)))
        return rc;
    // the compact batch stands where the search's operands and lists are, for the duration of its own scan
    auto swap_in = [&]() {
        std::swap(ix->w_q32, c.q32);
        std::swap(ix->w_q16b, c.q16b);
        std::swap(ix->w_ls, c.ls);
        std::swap(ix->w_li, c.li);
        std::swap(ix->w_tau, c.tau);
    };
    swap_in();
    rc = [&]() -> int {
        int r2;
        LDOT_HIP_CHECK(hipMemsetAsync(ix->w_over.p, 0, ix->w_over.bytes, st));
        LDOT_HIP_CHECK(hipMemsetAsync(ix->w_over_sum.p, 0, 16, st));
        if (level == 1) return dense_scan_all(ix, nf, 0, ix->ntotal, kp, (float*)ix->w_tau.p, true, st);
        if ((r2 = fused_pools(ix, nf_pad, st))) return r2;
        if ((r2 = fused_launch_and_select(ix, 0, nf, nf_pad, kp, 0, ix->ntotal, st))) return r2;
        ix->pools_clean = true;
        ix->stats[3] += ix->ntotal * nf;
        LDOT_HIP_CHECK(hipMemcpyAsync(ix->h_over_sum, ix->w_over_sum.p, 4, hipMemcpyDeviceToHost, st));
        LDOT_HIP_CHECK(hipStreamSynchronize(st));
        if (ix->h_over_sum[0] > 0) return redo_flagged(ix, nf, nf_pad, kp, st, 1);
        return LDOT_OK;
    }();
    swap_in();   // (back)
    ix->flags_clean = true;
    if (rc) return rc;
    return launch_scatter_lists((const float*)c.ls.p, (const int32_t*)c.li.p, (const float*)c.tau.p, didx, nf, kp, (float*)ix->w_ls.p,
                                (int32_t*)ix->w_li.p, (float*)ix->w_tau.p, st);
}

// the queries of a search that read them in place (ldot_index::unstaged_q) -> fp32 + bf16 staging copies, for the recovery paths
static int stage_unstaged_queries(ldot_index* ix, int64_t nq, hipStream_t st) {
    if (!ix->unstaged_q) return LDOT_OK;
    const void* src = ix->unstaged_q;
    ix->unstaged_q = nullptr;
    return launch_convert_rows(src, LDOT_F32, ix->unstaged_ld, nq, round_up(nq, kBM), ix->d, ix->dpad, 0, (float*)ix->w_q32.p, nullptr, 0,
                               (uint16_t*)ix->w_q16b.p, 0, st);
}

// LDOT_MODE_AUTO: fused scan or dense chunks?  The dense path writes and re-reads 8 bytes per (query, row) pair, the fused scan pays a
// warm-up, a pool select per launch and its admissions: it wins from 32 768 rows for any batch, from ~20 000 rows for >= 4096 queries and
// from ~8 000 rows for >= 16 384 (tools/auto_threshold.py with the round-5 dense kernel, profiles/r05_auto_threshold.txt: 5 000 x 24 576
// 0.759 -> 0.715 ms, 25 000 x 8 192 2.16 -> 2.06, 25 000 x 16 384 2.94 -> 2.56).  <= 16 queries whose narrow search is not available take
// the wide dense scan at every size (tools/serving_latency.py).
static bool auto_fused(const ldot_index* ix, int64_t nq) {
    if (nq <= 16 && narrow_ok(ix, nq)) return false;
    const int64_t n = ix->ntotal;
    return n >= 32768 || (n >= 20480 && nq >= 4096) || (n >= 8192 && nq >= 16384);
}

// defer_check: enqueue a fused scan speculatively and leave the overflow check to the caller's own synchronisation point
// warm_only (ldot_index_search_warmup): stop after the local warm-up of a fused scan, leave the statistics the ranks exchange in
// stat_out (2 * nq floats) and remember the path in split_path; ldot_index_search_scan continues from there
static int search_begin_impl(ldot_index_t* ix, const void* queries, int64_t nq, int dtype, int mem, int normalize, int k,
                             float* tau_out, bool defer_check, hipStream_t st, const DirectOut* direct = nullptr,
                             bool warm_only = false, int parts = 1, float* stat_out = nullptr, int64_t shard_total = -1,
                             double shard_share = 0.0) {
    LDOT_REQUIRE(ix != nullptr, LDOT_EINVAL, "index is NULL");
    LDOT_REQUIRE(nq >= 0, LDOT_EINVAL, "negative query count");
    LDOT_REQUIRE(k >= 1 && k <= kMaxK, LDOT_EINVAL, "k must be in [1, %d] (got %d)", kMaxK, k);
    LDOT_REQUIRE(dtype >= 0 && dtype <= 2, LDOT_EINVAL, "bad dtype %d", dtype);
    LDOT_REQUIRE(mem == LDOT_HOST || mem == LDOT_DEVICE, LDOT_EINVAL, "bad memory space");
    ix->pend_nq = 0;
    ix->pend_done = false;
    ix->overflow_pending = false;
    ix->overflow_narrow = 0;
    ix->qcnt_n = 0;
    ix->unproven_n = 0;
    ix->split_path = 0;
    ix->cur_parts = warm_only ? parts : 1;
    // shard_total >= 0: ldot_index_search_begin_shard (one shard of `parts`; > 0: scan on pooled statistics, the whole index has that many rows)
    ix->pool_total = (!warm_only && shard_total > 0 && parts > 1) ? shard_total : 0;
    ix->pool_parts = parts;
    ix->pooled_used = false;
    if (nq == 0) return LDOT_OK;
    LDOT_REQUIRE(queries != nullptr, LDOT_EINVAL, "NULL buffer");
    DeviceGuard guard(ix->device);
    for (int i = 0; i < 4; ++i) ix->stats[i] = 0;
    ix->redone = 0;
    ix->last_path = ix->last_thresholds = 0;
    ix->last_order = 1;
    if (ix->want_reshuffle) {
        ix->want_reshuffle = false;
        int rrc = reshuffle_rows(ix, st);
        if (rrc) return rrc;
    }
    const int kp = candidate_len(ix, k);
    const int64_t nq_pad = round_up(nq, kBM);
    int rc;
    if ((rc = ix->w_q32.ensure((size_t)nq_pad * ix->dpad * 4))) return rc;
    if ((rc = ix->w_q16b.ensure((size_t)nq_pad * ix->ld16() * 2))) return rc;
    if ((rc = ix->w_ls.ensure((size_t)nq_pad * kp * 4))) return rc;
    if ((rc = ix->w_li.ensure((size_t)nq_pad * kp * 4))) return rc;
    if ((rc = ix->w_tau.ensure((size_t)nq_pad * 4))) return rc;

    // ingest queries -> fp32 (exact re-score operand) + bf16 (MFMA operand); pad rows of the last tile are zero
    const void* src = queries;
    if (mem == LDOT_HOST) {
        const size_t bytes = (size_t)nq * ix->d * dtype_size(dtype);
        if ((rc = ix->w_stage.ensure(bytes))) return rc;
        LDOT_HIP_CHECK(hipMemcpyAsync(ix->w_stage.p, queries, bytes, hipMemcpyHostToDevice, st));
        src = ix->w_stage.p;
    }
    float* tau = (float*)ix->w_tau.p;
    // (the narrow search writes complete lists and thresholds itself)
    bool narrow = ix->ntotal > 0 && ix->mode == LDOT_MODE_AUTO && narrow_select_ok(ix, nq, kp);
    if (narrow && ix->narrow_backoff > 0) {   // (this index recently filled the candidate buffer: streaming selector for a while)
        --ix->narrow_backoff;
        narrow = false;
    }
    // A few fp32 device queries answered by the one-launch narrow search with kernel-written outputs are not staged at all: the scan
    // converts them itself (the recovery of an overflowed search stages them then, stage_unstaged_queries).
    DirectOut direct_q;
    ix->unstaged_q = nullptr;
    if (narrow && direct && nq <= 16 && narrow_one_launch(ix, nq, kp) && dtype == LDOT_F32 && mem == LDOT_DEVICE && !normalize && !ix->precision &&
        (ix->d == ix->dpad || ix->q_prepadded) && ((uintptr_t)queries & 15) == 0) {
        direct_q = *direct;
        direct_q.qf32 = (const float*)queries;
        direct_q.ldqf = ix->q_prepadded ? ix->dpad : ix->d;
        direct = &direct_q;
        ix->unstaged_q = queries;
        ix->unstaged_ld = direct_q.ldqf;
    } else if ((rc = launch_convert_rows(src, dtype, ix->q_prepadded ? ix->dpad : ix->d, nq, nq_pad, ix->d, ix->dpad, normalize,
                                         (float*)ix->w_q32.p, nullptr, ix->precision ? 2 : 0, (uint16_t*)ix->w_q16b.p, 0, st))) {
        return rc;
    }
    if (!narrow && (rc = launch_init_lists((float*)ix->w_ls.p, (int32_t*)ix->w_li.p, nq_pad * kp, tau, nq, nq_pad, st))) return rc;

    if (warm_only) {
        const bool fused = !narrow && ix->ntotal > 0 && (ix->mode == LDOT_MODE_FUSED || (ix->mode == LDOT_MODE_AUTO && auto_fused(ix, nq)));
        if (fused) {
            if ((rc = fused_scan(ix, nq, nq_pad, kp, st, 1))) return rc;
            // m = ceil(k' / parts): every rank has m rows at or above its own m-th best warm-up score
            if ((rc = launch_list_stats((const float*)ix->w_ls.p, (const int32_t*)ix->w_li.p, kp, nq, (kp + parts - 1) / parts, tau, stat_out, st)))
                return rc;
        } else if ((rc = launch_neutral_stats(nq, stat_out, st))) {
            return rc;
        }
        if (mem == LDOT_HOST) LDOT_HIP_CHECK(hipStreamSynchronize(st));   // the staging buffer is reused by the next call
        ix->split_path = narrow ? 1 : fused ? 3 : 2;
        ix->split_parts = parts;
        ix->pend_nq = nq;
        ix->pend_k = k;
        ix->pend_kp = kp;
        return LDOT_OK;
    }
    if (narrow) {
        ix->last_path = 1;
        if ((rc = narrow_search(ix, nq, kp, st, direct))) return rc;
        if (!defer_check) {
            LDOT_HIP_CHECK(hipStreamSynchronize(st));
            if (fused_overflow_check(ix)) {
                if ((rc = stage_unstaged_queries(ix, nq, st))) return rc;
                if ((rc = redo_flagged(ix, nq, nq_pad, kp, st))) return rc;
            }
        }
    } else if (ix->ntotal > 0) {
        // AUTO: the fused scan pays off from ~32k rows (tools/auto_threshold.py); very large batches (COCO-5k sized image->text
        // with the reference's un-deduplicated queries) already from 16k rows, where the dense score matrix is the cost
        // <= 16 queries whose narrow search is not available (large k', or the index recently filled its candidate buffer): one pass
        // over the index at HBM speed (score_narrow.hip) + segmented streaming select still beats the fused scan's warm-up / filter /
        // pool-select chain at every index size (tools/serving_latency.py)
        const bool fused = ix->mode == LDOT_MODE_FUSED || (ix->mode == LDOT_MODE_AUTO && auto_fused(ix, nq));
        ix->last_path = !fused ? 2 : nq <= kFewSelectMaxQueries ? 3 : 4;
        if (fused) {
            if ((rc = fused_scan(ix, nq, nq_pad, kp, st))) return rc;
            ix->last_thresholds = ix->pooled_used ? 3 : ix->opt_used ? 2 : 1;
            ix->last_order = ix->scrambled_now ? 2 : 1;
            if (!defer_check) {
                LDOT_HIP_CHECK(hipStreamSynchronize(st));
                if (fused_overflow_check(ix) && (rc = redo_flagged(ix, nq, nq_pad, kp, st))) return rc;
            }
        } else if ((rc = dense_scan_all(ix, nq, 0, ix->ntotal, kp, tau, true, st))) {
            return rc;
        }
    }
    if (tau_out) LDOT_HIP_CHECK(hipMemcpyAsync(tau_out, tau, (size_t)nq * 4, hipMemcpyDeviceToDevice, st));
    if (!warm_only && shard_total >= 0 && stat_out) {
        // what the ranks exchange: the k'-th best, -(the ceil(k'/parts)-th best) and the level above which this list is complete
        // (pooled statistics: the last threshold the rows were filtered with; a query whose candidate pools overflowed was searched
        // again by redo_flagged and has a complete list)
        const bool pooled = ix->pooled_used;
        // the rank of the second statistic: this shard vouches for ceil(k' x share) of the k' rows (0: for none)
        const int j = shard_share > 0.0 ? std::min(kp, std::max(1, (int)ceil((double)kp * shard_share - 1e-9))) : 0;
        if ((rc = launch_list_stats((const float*)ix->w_ls.p, (const int32_t*)ix->w_li.p, kp, nq, j, tau, stat_out, st, 3,
                                    pooled ? (const float*)ix->w_tau_opt.p : nullptr,
                                    pooled && ix->redone > 0 ? (const int32_t*)ix->w_redone.p : nullptr)))
            return rc;
    }
    ix->pool_total = 0;
    if (mem == LDOT_HOST) LDOT_HIP_CHECK(hipStreamSynchronize(st));   // the staging buffer is reused by the next call
    ix->pend_nq = nq;
    ix->pend_k = k;
    ix->pend_kp = kp;
    return LDOT_OK;
}

int ldot_index_search_begin(ldot_index_t* ix, const void* queries, int64_t nq, int dtype, int mem, int normalize, int k,
                            float* tau_out, void* stream) {
    return search_begin_impl(ix, queries, nq, dtype, mem, normalize, k, tau_out, false, (hipStream_t)stream);
}

// One shard's candidate pass of a sharded search + the three numbers per query its ranks all-reduce (MAX) afterwards (ldot.h).
// total_rows > 0: large batches scan on statistics pooled over the whole index (fused_rest_chunk_optimistic).
int ldot_index_search_begin_shard(ldot_index_t* ix, const void* queries, int64_t nq, int dtype, int mem, int normalize, int k, int parts,
                                  double share, int64_t total_rows, float* stat_out, void* stream) {
    LDOT_REQUIRE(parts >= 1 && parts <= 65536, LDOT_EINVAL, "bad number of parts %d", parts);
    LDOT_REQUIRE(share >= 0.0 && share <= 1.0, LDOT_EINVAL, "share must be in [0, 1]");
    LDOT_REQUIRE(total_rows >= 0, LDOT_EINVAL, "negative row count");
    if (nq > 0) LDOT_REQUIRE(stat_out != nullptr, LDOT_EINVAL, "NULL buffer");
    return search_begin_impl(ix, queries, nq, dtype, mem, normalize, k, nullptr, false, (hipStream_t)stream, nullptr, false, parts, stat_out,
                             total_rows, share);
}

int ldot_index_shard_floor(ldot_index_t* ix, const float* stat, float* floor_out, int32_t* count_out, int* k_prime_out, void* stream) {
    LDOT_REQUIRE(ix != nullptr, LDOT_EINVAL, "index is NULL");
    if (k_prime_out) *k_prime_out = ix->pend_kp;
    const int64_t nq = ix->pend_nq;
    if (nq == 0) return LDOT_OK;
    LDOT_REQUIRE(stat != nullptr && floor_out != nullptr && count_out != nullptr, LDOT_EINVAL, "NULL buffer");
    DeviceGuard guard(ix->device);
    return launch_shard_floor(stat, nq, (const float*)ix->w_ls.p, (const int32_t*)ix->w_li.p, ix->pend_kp, floor_out, count_out,
                              (hipStream_t)stream);
}

// A sharded search in three steps (lightningdot_amd/sharded.py): every rank warms up on its own shard and publishes two numbers per
// query; one all-reduce(MAX) later every rank continues with the threshold all of them can vouch for.
int ldot_index_search_warmup(ldot_index_t* ix, const void* queries, int64_t nq, int dtype, int mem, int normalize, int k, int parts,
                             float* stat_out, void* stream) {
    LDOT_REQUIRE(parts >= 1 && parts <= 65536, LDOT_EINVAL, "bad number of parts %d", parts);
    if (nq > 0) LDOT_REQUIRE(stat_out != nullptr, LDOT_EINVAL, "NULL buffer");
    return search_begin_impl(ix, queries, nq, dtype, mem, normalize, k, nullptr, false, (hipStream_t)stream, nullptr, true, parts, stat_out);
}

int ldot_index_search_scan(ldot_index_t* ix, const float* stat_in, float* tau_out, void* stream) {
    LDOT_REQUIRE(ix != nullptr, LDOT_EINVAL, "index is NULL");
    const int64_t nq = ix->pend_nq;
    if (nq == 0) return LDOT_OK;
    LDOT_REQUIRE(ix->split_path != 0, LDOT_EINVAL, "ldot_index_search_scan without a pending ldot_index_search_warmup");
    const int path = ix->split_path, kp = ix->pend_kp;
    ix->split_path = 0;
    DeviceGuard guard(ix->device);
    hipStream_t st = (hipStream_t)stream;
    const int64_t nq_pad = round_up(nq, kBM);
    float* tau = (float*)ix->w_tau.p;
    int rc;
    if (path == 1) {   // (the small-batch and small-index paths do not use the agreed thresholds: their scan is one pass anyway)
        if ((rc = narrow_search(ix, nq, kp, st, nullptr))) return rc;
        LDOT_HIP_CHECK(hipStreamSynchronize(st));
        if (fused_overflow_check(ix) && (rc = redo_flagged(ix, nq, nq_pad, kp, st))) return rc;
    } else if (path == 2) {
        if (ix->ntotal > 0 && (rc = dense_scan_all(ix, nq, 0, ix->ntotal, kp, tau, true, st))) return rc;
    } else {
        if (stat_in && (rc = launch_apply_stats(nq, stat_in, tau, st))) return rc;
        if ((rc = fused_scan(ix, nq, nq_pad, kp, st, 2, stat_in ? ix->split_parts : 1))) return rc;
        LDOT_HIP_CHECK(hipStreamSynchronize(st));
        if (fused_overflow_check(ix) && (rc = redo_flagged(ix, nq, nq_pad, kp, st))) return rc;
    }
    if (tau_out) LDOT_HIP_CHECK(hipMemcpyAsync(tau_out, tau, (size_t)nq * 4, hipMemcpyDeviceToDevice, st));
    return LDOT_OK;
}

// re-score + output
static int search_finish_impl(ldot_index_t* ix, const float* floor, float* out_scores, int64_t* out_labels, int out_mem,
                              bool keep_pending, hipStream_t st) {
    LDOT_REQUIRE(ix != nullptr, LDOT_EINVAL, "index is NULL");
    LDOT_REQUIRE(out_mem == LDOT_HOST || out_mem == LDOT_DEVICE, LDOT_EINVAL, "bad memory space");
    const int64_t nq = ix->pend_nq;
    if (nq == 0) return LDOT_OK;
    LDOT_REQUIRE(out_scores && out_labels, LDOT_EINVAL, "NULL buffer");
    const int k = ix->pend_k, kp = ix->pend_kp;
    if (!keep_pending) ix->pend_nq = 0;
    DeviceGuard guard(ix->device);
    int rc;
    const int32_t* lmap = ix->shuffled ? (const int32_t*)ix->w_label.p : nullptr;   // (LDOT_OPT_ROW_SHUFFLE: stored row -> label)
    // LDOT_OPT_VERIFY (plain searches only: a sharded search compares against the GLOBAL threshold, which this shard cannot judge)
    auto verify = [&](const float* dev_s, const int64_t* dev_l) -> int {
        if (!ix->verify || floor != nullptr || !ix->rescore || ix->w_norm.p == nullptr || ix->result_set) return LDOT_OK;
        int vrc = ix->w_unproven.ensure((size_t)(nq + 1) * 4);
        if (vrc) return vrc;
        LDOT_HIP_CHECK(hipMemsetAsync((int32_t*)ix->w_unproven.p + nq, 0, 4, st));
        ix->unproven_n = nq;
        return launch_verify_exact((const float*)ix->w_q32.p, ix->dpad, ix->d, nq, dev_s, dev_l, k, (const float*)ix->w_tau.p,
                                   (const float*)ix->w_norm.p, (int32_t*)ix->w_unproven.p, (int32_t*)ix->w_unproven.p + nq, st);
    };
    // LDOT_OPT_RESULT_SET (plain searches with the exact re-score on): the top-k set, boundary candidates re-scored only
    const bool as_set = ix->result_set && floor == nullptr && ix->rescore && ix->w_norm.p != nullptr;
    ix->set_stats_valid = false;
    if (as_set) {
        if ((rc = ix->w_set_stats.ensure(16))) return rc;
        LDOT_HIP_CHECK(hipMemsetAsync(ix->w_set_stats.p, 0, 16, st));
        ix->set_stats_valid = true;
    }
    auto rescore_to = [&](float* os, int64_t* ol) -> int {
        if (as_set)
            return launch_rescore_set((const float*)ix->w_q32.p, ix->dpad, ix->x32, ix->dpad, ix->dpad, ix->d, nq, (const float*)ix->w_ls.p,
                                      (const int32_t*)ix->w_li.p, kp, k, (const float*)ix->w_norm.p, kVerifyC, os, ol, lmap,
                                      (unsigned long long*)ix->w_set_stats.p, st);
        return launch_rescore((const float*)ix->w_q32.p, ix->dpad, ix->x32, ix->dpad, ix->dpad, nq, (const float*)ix->w_ls.p,
                              (const int32_t*)ix->w_li.p, kp, k, ix->rescore, floor, os, ol, st, nullptr, lmap);
    };
    if (out_mem == LDOT_DEVICE) {   // device outputs are written by the re-score kernel directly
        if ((rc = rescore_to(out_scores, out_labels))) return rc;
        if ((rc = verify(out_scores, out_labels))) return rc;
        prof_collect(ix, st);
        return LDOT_OK;
    }
    // Pinned (device-mapped) host buffers: the re-score kernel stores its results straight into host memory — the 12 MB of a
    // 10k x top-100 result set leave over PCIe while the kernel is still gathering rows, no staging buffer, no copy kernels.
    void *ms = nullptr, *ml = nullptr;
    const bool mapped = hipHostGetDevicePointer(&ms, out_scores, 0) == hipSuccess && ms != nullptr &&
                        hipHostGetDevicePointer(&ml, out_labels, 0) == hipSuccess && ml != nullptr;
    (void)hipGetLastError();   // (a pageable buffer makes the query fail: not an error of this call)
    if (mapped) {
        if ((rc = rescore_to((float*)ms, (int64_t*)ml))) return rc;
        if ((rc = verify((const float*)ms, (const int64_t*)ml))) return rc;
        // LDOT_OPT_DEFER_SYNC: the caller synchronises (everything this search used stays alive until the handle's next call on this
        // stream).  Profiling events are read on the host and the verify flags are the caller's to read: both keep the synchronisation.
        if (ix->defer_sync && !ix->profile && !ix->verify && floor == nullptr && !keep_pending) return LDOT_OK;
        LDOT_HIP_CHECK(hipStreamSynchronize(st));
        prof_collect(ix, st);
        return LDOT_OK;
    }
    // pageable host buffers: device workspace + two copies (hipMemcpyAsync stages them through the runtime's pinned buffers)
    if ((rc = ix->w_outs.ensure((size_t)nq * k * 4))) return rc;
    if ((rc = ix->w_outl.ensure((size_t)nq * k * 8))) return rc;
    if ((rc = rescore_to((float*)ix->w_outs.p, (int64_t*)ix->w_outl.p))) return rc;
    if ((rc = verify((const float*)ix->w_outs.p, (const int64_t*)ix->w_outl.p))) return rc;
    LDOT_HIP_CHECK(hipMemcpyAsync(out_scores, ix->w_outs.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
    LDOT_HIP_CHECK(hipMemcpyAsync(out_labels, ix->w_outl.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
    LDOT_HIP_CHECK(hipStreamSynchronize(st));
    prof_collect(ix, st);
    return LDOT_OK;
}

int ldot_index_search_finish(ldot_index_t* ix, const float* floor, float* out_scores, int64_t* out_labels, int out_mem,
                             void* stream) {
    return search_finish_impl(ix, floor, out_scores, out_labels, out_mem, false, (hipStream_t)stream);
}

// _finish of a sharded search, straight into the send buffer of the all-to-all that follows: block b (one per destination rank)
// receives the partial lists of the queries [b * block_rows, (b + 1) * block_rows), labels already global (+ label_base)
int ldot_index_search_finish_blocked(ldot_index_t* ix, const float* floor, void* out_blocks, int64_t block_rows, int64_t block_bytes,
                                     int64_t label_base, void* stream) {
    LDOT_REQUIRE(ix != nullptr, LDOT_EINVAL, "index is NULL");
    const int64_t nq = ix->pend_nq;
    if (nq == 0) return LDOT_OK;
    const int k = ix->pend_k, kp = ix->pend_kp;
    LDOT_REQUIRE(out_blocks != nullptr, LDOT_EINVAL, "NULL buffer");
    const int64_t lab_off = LDOT_BLOCK_LABELS_OFFSET(block_rows, k);
    LDOT_REQUIRE(block_rows >= 1 && block_bytes % 16 == 0 && block_bytes >= lab_off + block_rows * k * 8 && ((uintptr_t)out_blocks & 15) == 0,
                 LDOT_EINVAL, "bad block geometry (rows %lld, bytes %lld, k %d)", (long long)block_rows, (long long)block_bytes, k);
    ix->pend_nq = 0;
    DeviceGuard guard(ix->device);
    hipStream_t st = (hipStream_t)stream;
    const RescoreOut lay{block_rows, block_bytes / 4, block_bytes / 8, label_base, ix->shuffled ? (const int32_t*)ix->w_label.p : nullptr};
    int rc = launch_rescore((const float*)ix->w_q32.p, ix->dpad, ix->x32, ix->dpad, ix->dpad, nq, (const float*)ix->w_ls.p,
                            (const int32_t*)ix->w_li.p, kp, k, ix->rescore, floor, (float*)out_blocks,
                            (int64_t*)((char*)out_blocks + lab_off), st, &lay);
    if (rc) return rc;
    prof_collect(ix, st);
    return LDOT_OK;
}

int ldot_index_search(ldot_index_t* ix, const void* queries, int64_t nq, int dtype, int mem, int normalize, int k,
                      float* out_scores, int64_t* out_labels, int out_mem, void* stream) {
    LDOT_REQUIRE(out_mem == LDOT_HOST || out_mem == LDOT_DEVICE, LDOT_EINVAL, "bad memory space");
    if (nq > 0) LDOT_REQUIRE(out_scores && out_labels, LDOT_EINVAL, "NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    // the fused scan is enqueued speculatively and the re-score behind it: ONE synchronisation per search (host outputs need it
    // anyway; device outputs pay a 4-byte round trip) instead of one in the middle that drains the stream before the re-score
    // where the final top-k may be written by a kernel directly (device memory, or pinned host memory through its device mapping):
    // a few-query search then ends in ONE kernel after the scan (narrow_finish_kernel)
    DirectOut direct{nullptr, nullptr, k};
    if (nq > 0 && ix && !ix->verify && !ix->shuffled && !ix->result_set) {   // (a shuffled index translates rows to labels in the re-score kernel;
                                                                            // the top-k set is decided there)
        if (out_mem == LDOT_DEVICE) {
            direct.scores = out_scores;
            direct.labels = out_labels;
        } else {
            void *ms = nullptr, *ml = nullptr;
            if (hipHostGetDevicePointer(&ms, out_scores, 0) == hipSuccess && ms && hipHostGetDevicePointer(&ml, out_labels, 0) == hipSuccess && ml) {
                direct.scores = (float*)ms;
                direct.labels = (int64_t*)ml;
            }
            (void)hipGetLastError();   // (a pageable buffer makes the query fail: not an error of this call)
        }
    }
    int rc = search_begin_impl(ix, queries, nq, dtype, mem, normalize, k, nullptr, true, st, direct.scores ? &direct : nullptr);
    if (rc) return rc;
    if (ix->pend_nq == 0) return LDOT_OK;
    if (ix->pend_done) {   // the results are on their way already; the one synchronisation of the search + the buffer-full check
        ix->pend_done = false;
        if (ix->chain_defer_sync) return LDOT_OK;   // (internal chain: the caller synchronises and checks, see ldot_ivf_search)
        LDOT_HIP_CHECK(hipStreamSynchronize(st));
        prof_collect(ix, st);
        if (fused_overflow_check(ix)) {
            if ((rc = stage_unstaged_queries(ix, nq, st))) return rc;
            if ((rc = redo_flagged(ix, nq, round_up(nq, kBM), ix->pend_kp, st))) return rc;
            return search_finish_impl(ix, nullptr, out_scores, out_labels, out_mem, false, st);
        }
        ix->pend_nq = 0;
        return LDOT_OK;
    }
    const bool check = ix->overflow_pending;
    if ((rc = search_finish_impl(ix, nullptr, out_scores, out_labels, out_mem, check, st))) return rc;
    if (!check) return LDOT_OK;
    if (out_mem == LDOT_DEVICE) LDOT_HIP_CHECK(hipStreamSynchronize(st));
    if (fused_overflow_check(ix)) {   // unfriendly row order: the flagged queries are searched again (redo_flagged), the rest re-scored as is
        if ((rc = redo_flagged(ix, nq, round_up(nq, kBM), ix->pend_kp, st))) return rc;
        return search_finish_impl(ix, nullptr, out_scores, out_labels, out_mem, false, st);
    }
    ix->pend_nq = 0;
    return LDOT_OK;
}

// ---- approximate (inverted-file) search: see ldot.h -------------------------------------------------------------------------------
// first version of the list scan, kept as the always-correct path (large k, or a query whose candidate buffer filled up): every probed
// list padded to the longest one, streaming segmented select.  probes: int32 [n][nprobe] (device)
static int lists_chunk_padded(ldot_index* ix, int64_t n, const int64_t* list_offsets, int nlist, int lpad, const int32_t* probes,
                              int nprobe, int k, int kp, float* ds, int64_t* dl, hipStream_t st) {
    const int64_t ncols = (int64_t)nprobe * lpad;
    int rc;
    if ((rc = ix->w_S.ensure((size_t)n * ncols * 4))) return rc;
    if ((rc = ix->w_ls.ensure((size_t)n * kp * 4))) return rc;
    if ((rc = ix->w_li.ensure((size_t)n * kp * 4))) return rc;
    if ((rc = ix->w_tau.ensure((size_t)n * 4))) return rc;
    const int64_t seg_cols = std::max<int64_t>(1024, std::min<int64_t>(16384, round_up((ncols + 15) / 16, 256)));
    const int64_t nseg = (ncols + seg_cols - 1) / seg_cols;
    if ((rc = ix->w_part_s.ensure((size_t)nseg * n * kp * 4))) return rc;
    if ((rc = ix->w_part_l.ensure((size_t)nseg * n * kp * 8))) return rc;
    if ((rc = launch_scan_lists((const float*)ix->w_q32.p, ix->dpad, ix->x32, ix->dpad, ix->dpad, n, list_offsets, probes, nprobe, nlist,
                                lpad, (float*)ix->w_S.p, ncols, st)))
        return rc;
    float* ls = (float*)ix->w_ls.p;
    int32_t* li = (int32_t*)ix->w_li.p;
    float* tau = (float*)ix->w_tau.p;
    if ((rc = launch_init_lists(ls, li, n * kp, tau, n, n, st))) return rc;
    if ((rc = launch_select_dense_parts((const float*)ix->w_S.p, ncols, n, ncols, seg_cols, 0, kp, (float*)ix->w_part_s.p,
                                        (int64_t*)ix->w_part_l.p, st)))
        return rc;
    if ((rc = launch_merge_parts_into_lists((const float*)ix->w_part_s.p, (const int64_t*)ix->w_part_l.p, (int)nseg, n, kp, ls, li, tau,
                                            st)))
        return rc;
    // final ordering (score desc, column asc) with the sort of the re-score kernel; no re-scoring: the scores are exact already
    if ((rc = launch_rescore((const float*)ix->w_q32.p, ix->dpad, ix->x32, ix->dpad, ix->dpad, n, ls, li, kp, k, 0, nullptr, ds, dl, st)))
        return rc;
    return launch_translate_cols(dl, n, k, list_offsets, probes, nprobe, nlist, lpad, st);
}

// queries: device memory of `dtype`; probes: device [nq][nprobe], int32 or int64 (the labels of a coarse search)
static int lists_search_impl(ldot_index* ix, const void* queries, int64_t nq, int dtype, int normalize, const int64_t* list_offsets,
                             int nlist, int64_t max_list_len, const void* probes, bool probes_int64, int nprobe, int k, float* out_scores,
                             int64_t* out_labels, int out_mem, hipStream_t st) {
    ix->pend_nq = 0;
    ix->overflow_pending = false;
    ix->overflow_narrow = 0;
    ix->qcnt_n = 0;
    ix->unproven_n = 0;
    for (int i = 0; i < 4; ++i) ix->stats[i] = 0;
    const int kp = (int)round_up(k, 32);           // the candidates carry exact scores: no margin
    const int lpad = (int)round_up(std::max<int64_t>(max_list_len, 1), 64);
    const int64_t max_cols = (int64_t)nprobe * lpad;   // upper bound of a query's column count
    LDOT_REQUIRE(max_cols < ((int64_t)1 << 31), LDOT_EINVAL, "nprobe * list length too large");
    // run-maxima selection: runs of 16 << run_shift columns, as long as it takes for <= 2048 runs per query (a coarser run hardly adds
    // candidates and makes the threshold search a 256-thread job) — but a query with fewer than k' runs makes every row a candidate,
    // which must fit the candidate buffer: run * k' <= capacity
    int run_shift = 0;
    while ((max_cols + (16 << run_shift) - 1) / (16 << run_shift) > 2048 && (int64_t)(32 << run_shift) * kp <= kNarrowCandCap) ++run_shift;
    const int run = 16 << run_shift;
    const int64_t nruns = (max_cols + run - 1) / run;
    const bool compact = nruns <= kNarrowMaxRuns && (int64_t)run * kp <= kNarrowCandCap && (size_t)(nprobe + 1) * 12 <= 64 * 1024;
    // queries are processed in chunks that bound the score workspace (<= 1 GiB)
    int64_t qchunk = std::max<int64_t>(1, std::min<int64_t>(nq, ((int64_t)1 << 28) / max_cols));
    if (compact) qchunk = std::min(qchunk, kListsQueryChunk);
    int rc;
    if ((rc = ix->w_q32.ensure((size_t)round_up(qchunk, kBM) * ix->dpad * 4))) return rc;
    if ((rc = ix->w_S.ensure((size_t)qchunk * max_cols * 4))) return rc;
    if ((rc = ix->w_lplist.ensure((size_t)qchunk * nprobe * 4))) return rc;
    if ((rc = ix->w_lcstart.ensure((size_t)qchunk * (nprobe + 1) * 4))) return rc;
    if ((rc = ix->w_lrowbase.ensure((size_t)qchunk * nprobe * 8))) return rc;
    if (out_mem == LDOT_HOST) {
        if ((rc = ix->w_outs.ensure((size_t)qchunk * k * 4))) return rc;
        if ((rc = ix->w_outl.ensure((size_t)qchunk * k * 8))) return rc;
    }
    const size_t esz = dtype_size(dtype), psz = probes_int64 ? 8 : 4;
    int32_t* plist = (int32_t*)ix->w_lplist.p;
    int32_t* cstart = (int32_t*)ix->w_lcstart.p;
    int64_t* rowbase = (int64_t*)ix->w_lrowbase.p;
    for (int64_t q0 = 0; q0 < nq; q0 += qchunk) {
        const int64_t n = std::min(qchunk, nq - q0);
        const char* src = (const char*)queries + (size_t)q0 * ix->d * esz;
        const char* pr = (const char*)probes + (size_t)q0 * nprobe * psz;
        float* ds = out_mem == LDOT_DEVICE ? out_scores + q0 * k : (float*)ix->w_outs.p;
        int64_t* dl = out_mem == LDOT_DEVICE ? out_labels + q0 * k : (int64_t*)ix->w_outl.p;
        // fp32 rows whose stride is the padded one are read where they are (no conversion kernel); the padded-list fallback stages them
        const bool inplace = dtype == LDOT_F32 && !normalize && ix->d == ix->dpad && ((uintptr_t)src & 15) == 0;
        const float* q32p = inplace ? (const float*)src : (const float*)ix->w_q32.p;
        bool staged = !inplace;
        auto stage = [&]() -> int {
            if (staged) return LDOT_OK;
            staged = true;
            return launch_convert_rows(src, dtype, ix->d, n, n, ix->d, ix->dpad, normalize, (float*)ix->w_q32.p, nullptr, 0, nullptr, 0, st);
        };
        if (!inplace && (rc = launch_convert_rows(src, dtype, ix->d, n, n, ix->d, ix->dpad, normalize, (float*)ix->w_q32.p, nullptr, 0,
                                                  nullptr, 0, st)))
            return rc;
        // validated list ids (int32) + the per-query prefix sums of the list lengths
        if ((rc = launch_ivf_prefix(pr, probes_int64 ? 1 : 0, n, nprobe, nlist, list_offsets, plist, rowbase, cstart, st)))
            return rc;
        bool redo = !compact;
        if (compact) {
            if ((rc = narrow_buffers(ix, n, nruns, st))) return rc;
            uint32_t* M = (uint32_t*)ix->w_nmax.p;
            uint32_t* tk = (uint32_t*)ix->w_ntau.p;
            // a few queries: threshold, collect, column -> row translation, (re-score) and final order are ONE launch (narrow_finish_kernel).
            // 8 .. 16 queries scan their lists from the bf16 shadow (half the bytes; k + margin candidates, re-scored exactly): measured
            // against the exact fp32 scan under the same finish kernel (tools/ivf_ab.py, 32 of 4000 lists over 1M rows) 0.149 vs 0.170 ms
            // for 16 queries, but 0.115 vs 0.106 ms for ONE query — its scan is a handful of microseconds either way and the bf16 route
            // pays a query conversion and a row gather on top
            const int kpb = candidate_len(ix, k);
            const bool few = n <= kNarrowMaxQueries && nruns <= 2048 && (int64_t)run * kpb <= 4096 && kpb <= 512;
            bool scan16 = few && n >= 8 && n <= 16 && ix->precision == 0 && ix->rescore &&
                          (size_t)ix->dpad / 32 * 1024 + (size_t)(nprobe + 1) * 12 + 8 <= 64 * 1024;
#ifdef LDOT_ABLATION
            if (getenv("LDOT_DEBUG_IVF_FP32")) scan16 = false;   // (A/B: the exact fp32 list scan under the same finish kernel)
            if (getenv("LDOT_DEBUG_IVF_BF16")) scan16 = few && n <= 16 && ix->precision == 0 && ix->rescore;
#endif
            if (scan16) {
                if ((rc = ix->w_q16b.ensure((size_t)round_up(n, 16) * ix->ld16() * 2))) return rc;
                if ((rc = launch_convert_rows(q32p, LDOT_F32, ix->dpad, n, round_up(n, 16), ix->d, ix->dpad, 0, nullptr, nullptr, 0,
                                              (uint16_t*)ix->w_q16b.p, 0, st)))
                    return rc;
                if ((rc = launch_ivf_scan_bf16(ix->w_q16b.p, ix->x16b, ix->ld16(), n, rowbase, cstart, nprobe, max_cols, run_shift,
                                               (float*)ix->w_S.p, max_cols, M, nruns, st)))
                    return rc;
                if ((rc = launch_narrow_finish((const float*)ix->w_S.p, 0, max_cols, M, nruns, (int)nruns, run, max_cols, (int)n,
                                               q32p, ix->dpad, ix->x32, ix->dpad, ix->dpad, kpb, k, 1, nullptr,
                                               nullptr, nullptr, ds, dl, ix->d_nover, cstart + nprobe, nprobe + 1, rowbase, cstart, nprobe,
                                               st)))
                    return rc;
            } else {
            if ((rc = launch_ivf_scan(q32p, ix->dpad, ix->x32, ix->dpad, ix->dpad, n, rowbase, cstart, nprobe,
                                      max_cols, run_shift, (float*)ix->w_S.p, max_cols, M, nruns, st)))
                return rc;
            if (few && kp <= kpb) {
                // (split-bf16 shadow / re-score switched off: exact fp32 scan, same single finish launch without a re-score)
                if ((rc = launch_narrow_finish((const float*)ix->w_S.p, 0, max_cols, M, nruns, (int)nruns, run, max_cols, (int)n, nullptr, 0,
                                               nullptr, 0, 0, kp, k, 0, nullptr, nullptr, nullptr, ds, dl, ix->d_nover, cstart + nprobe,
                                               nprobe + 1, rowbase, cstart, nprobe, st)))
                    return rc;
            } else {
            if ((rc = launch_narrow_tau(M, nruns, (int)nruns, (int)n, kp, tk, st))) return rc;
            if ((rc = launch_narrow_collect((const float*)ix->w_S.p, max_cols, M, nruns, (int)nruns, run, max_cols, 0, (int)n, tk,
                                            (uint64_t*)ix->w_ncand.p, kNarrowCandCap, (int32_t*)ix->w_ncnt.p, cstart + nprobe, nprobe + 1,
                                            0, st)))
                return rc;
            if ((rc = launch_ivf_final((const uint64_t*)ix->w_ncand.p, kNarrowCandCap, (int32_t*)ix->w_ncnt.p, n, rowbase, cstart, nprobe, k,
                                       ds, dl, ix->d_nover, st)))
                return rc;
            }
            }
            ix->narrow_clean = true;
            // a full candidate buffer (thousands of equal scores) is rare but must not go unnoticed: one synchronisation per chunk
            LDOT_HIP_CHECK(hipStreamSynchronize(st));
            for (int64_t q = 0; q < n; ++q) {
                if (ix->h_nover[q]) {
                    redo = true;
                    ix->stats[1] += 1;
                }
            }
        }
        if (redo && ((rc = stage()) || (rc = lists_chunk_padded(ix, n, list_offsets, nlist, lpad, plist, nprobe, k, kp, ds, dl, st)))) return rc;
        ix->stats[2] += n * max_cols;
        if (out_mem == LDOT_HOST) {
            LDOT_HIP_CHECK(hipMemcpyAsync(out_scores + q0 * k, ds, (size_t)n * k * 4, hipMemcpyDeviceToHost, st));
            LDOT_HIP_CHECK(hipMemcpyAsync(out_labels + q0 * k, dl, (size_t)n * k * 8, hipMemcpyDeviceToHost, st));
            LDOT_HIP_CHECK(hipStreamSynchronize(st));
        }
    }
    return LDOT_OK;
}

int ldot_index_search_lists(ldot_index_t* ix, const void* queries, int64_t nq, int dtype, int normalize, const int64_t* list_offsets,
                            int nlist, int64_t max_list_len, const int32_t* probes, int nprobe, int k, float* out_scores,
                            int64_t* out_labels, int out_mem, void* stream) {
    LDOT_REQUIRE(ix != nullptr, LDOT_EINVAL, "index is NULL");
    LDOT_REQUIRE(nq >= 0 && k >= 1 && k <= kMaxK, LDOT_EINVAL, "bad nq / k");
    LDOT_REQUIRE(dtype >= 0 && dtype <= 2, LDOT_EINVAL, "bad dtype %d", dtype);
    LDOT_REQUIRE(out_mem == LDOT_HOST || out_mem == LDOT_DEVICE, LDOT_EINVAL, "bad memory space");
    LDOT_REQUIRE(nlist >= 1 && nprobe >= 1 && nprobe <= nlist && max_list_len >= 0, LDOT_EINVAL, "bad list geometry");
    LDOT_REQUIRE(!ix->shuffled, LDOT_ESTATE, "list search over an index whose rows are shuffled (LDOT_OPT_ROW_SHUFFLE): lists are row ranges");
    if (nq == 0) return LDOT_OK;
    LDOT_REQUIRE(queries && list_offsets && probes && out_scores && out_labels, LDOT_EINVAL, "NULL buffer");
    DeviceGuard guard(ix->device);
    return lists_search_impl(ix, queries, nq, dtype, normalize, list_offsets, nlist, max_list_len, probes, false, nprobe, k, out_scores,
                             out_labels, out_mem, (hipStream_t)stream);
}

// The whole approximate query in one call: coarse search over the list centroids + list scan.  `coarse` indexes the nlist centroids
// in the augmented space of the reference's HNSW indexer (faiss_indexers.py:114-131) with one more coordinate:
// row l = [c~_l (d + 1), -|c~_l|^2 / 2], so that the inner product with [q, 0, 1] orders the lists by L2 distance to [q, 0].
int ldot_ivf_search(ldot_index_t* ix, ldot_index_t* coarse, const void* queries, int64_t nq, int dtype, int normalize,
                    const int64_t* list_offsets, int64_t max_list_len, int nprobe, int k, float* out_scores, int64_t* out_labels,
                    int out_mem, void* stream) {
    LDOT_REQUIRE(ix != nullptr && coarse != nullptr, LDOT_EINVAL, "index is NULL");
    LDOT_REQUIRE(nq >= 0 && k >= 1 && k <= kMaxK, LDOT_EINVAL, "bad nq / k");
    LDOT_REQUIRE(dtype >= 0 && dtype <= 2, LDOT_EINVAL, "bad dtype %d", dtype);
    LDOT_REQUIRE(out_mem == LDOT_HOST || out_mem == LDOT_DEVICE, LDOT_EINVAL, "bad memory space");
    LDOT_REQUIRE(!ix->shuffled && !coarse->shuffled, LDOT_ESTATE, "inverted-file search over an index whose rows are shuffled (LDOT_OPT_ROW_SHUFFLE)");
    LDOT_REQUIRE(coarse->d == ix->d + 2 && coarse->device == ix->device, LDOT_EINVAL,
                 "the coarse index must hold (d + 2)-dimensional augmented centroids on the same device");
    const int64_t nlist = coarse->ntotal;
    LDOT_REQUIRE(nlist >= 1 && nlist < ((int64_t)1 << 31) && nprobe >= 1 && nprobe <= nlist && nprobe <= kMaxK && max_list_len >= 0,
                 LDOT_EINVAL, "bad list geometry");
    if (nq == 0) return LDOT_OK;
    LDOT_REQUIRE(queries && list_offsets && out_scores && out_labels, LDOT_EINVAL, "NULL buffer");
    DeviceGuard guard(ix->device);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    // augmented queries [q, 0, 1] (fp32, rows padded with zeros to the coarse index's row stride: a few queries are then read in place
    // by its scan, no conversion kernel) and the coarse result (probe labels) live in workspaces of the ROW index
    const int da_ld = coarse->dpad;
    if ((rc = ix->w_laug.ensure((size_t)nq * da_ld * 4))) return rc;
    if ((rc = ix->w_lprobe_s.ensure((size_t)nq * nprobe * 4))) return rc;
    if ((rc = ix->w_lprobe_l.ensure((size_t)nq * nprobe * 8))) return rc;
    if ((rc = launch_augment_queries(queries, dtype, ix->d, nq, normalize, (float*)ix->w_laug.p, da_ld, st))) return rc;
    // The coarse search of a few queries ends in a kernel that writes the probes itself; its synchronisation + buffer-full check is
    // DEFERRED to the synchronisation of the list stage (one host round trip per search instead of two).  A full coarse buffer
    // (thousands of centroids with equal scores) is then found after the fact and the search repeated the plain way.
    // (the chain state lives in the coarse index for the duration of this call; the guard clears it on EVERY way out, so that an error
    // return can not leave a later plain search of the coarse handle reading its queries with the padded stride)
    struct ChainGuard {
        ldot_index_t* c;
        ~ChainGuard() {
            c->q_prepadded = false;
            c->chain_defer_sync = false;
            c->pend_nq = 0;
        }
    } chain_guard{coarse};
    coarse->q_prepadded = true;
    coarse->chain_defer_sync = true;
    rc = ldot_index_search(coarse, ix->w_laug.p, nq, LDOT_F32, LDOT_DEVICE, 0, nprobe, (float*)ix->w_lprobe_s.p,
                           (int64_t*)ix->w_lprobe_l.p, LDOT_DEVICE, stream);
    coarse->chain_defer_sync = false;
    const bool unchecked = rc == LDOT_OK && coarse->overflow_pending;
    if (rc == LDOT_OK)
        rc = lists_search_impl(ix, queries, nq, dtype, normalize, list_offsets, (int)nlist, max_list_len, ix->w_lprobe_l.p, true, nprobe, k,
                               out_scores, out_labels, out_mem, st);
    if (rc == LDOT_OK && unchecked) {
        LDOT_HIP_CHECK(hipStreamSynchronize(st));
        if (fused_overflow_check(coarse)) {   // (what ldot_index_search does at its own synchronisation point)
            if ((rc = stage_unstaged_queries(coarse, nq, st)) == LDOT_OK &&
                (rc = redo_flagged(coarse, nq, round_up(nq, kBM), coarse->pend_kp, st)) == LDOT_OK)
                rc = search_finish_impl(coarse, nullptr, (float*)ix->w_lprobe_s.p, (int64_t*)ix->w_lprobe_l.p, LDOT_DEVICE, false, st);
            if (rc == LDOT_OK)
                rc = lists_search_impl(ix, queries, nq, dtype, normalize, list_offsets, (int)nlist, max_list_len, ix->w_lprobe_l.p, true,
                                       nprobe, k, out_scores, out_labels, out_mem, st);
        }
    }
    return rc;
}

int ldot_index_last_profile(const ldot_index_t* ix, double out[4]) {
    LDOT_REQUIRE(ix != nullptr && out != nullptr, LDOT_EINVAL, "NULL argument");
    for (int i = 0; i < 4; ++i) out[i] = ix->prof[i];
    return LDOT_OK;
}

// ---- serialisation: "LDOTIDX1" | int32 d | int64 ntotal | ntotal*d fp32 (row-major, unpadded) -------------
int ldot_index_save(ldot_index_t* ix, const char* path) {
    LDOT_REQUIRE(ix != nullptr && path != nullptr, LDOT_EINVAL, "NULL argument");
    DeviceGuard guard(ix->device);
    std::vector<float> host;
    try {
        host.resize((size_t)ix->ntotal * ix->d);
    } catch (const std::bad_alloc&) {
        set_error("out of host memory staging %lld rows", (long long)ix->ntotal);
        return LDOT_ENOMEM;
    }
    if (ix->ntotal > 0) {
        int rc = ldot_index_get_rows(ix, 0, ix->ntotal, host.data(), LDOT_HOST, nullptr);
        if (rc) return rc;
    }
    FILE* f = fopen(path, "wb");
    LDOT_REQUIRE(f != nullptr, LDOT_EIO, "cannot open %s for writing", path);
    const char magic[8] = {'L', 'D', 'O', 'T', 'I', 'D', 'X', '1'};
    int32_t d = ix->d;
    int64_t n = ix->ntotal;
    bool ok = fwrite(magic, 1, 8, f) == 8 && fwrite(&d, 4, 1, f) == 1 && fwrite(&n, 8, 1, f) == 1 &&
              (host.empty() || fwrite(host.data(), sizeof(float), host.size(), f) == host.size());
    ok = (fclose(f) == 0) && ok;
    LDOT_REQUIRE(ok, LDOT_EIO, "short write to %s", path);
    return LDOT_OK;
}

int ldot_index_load(const char* path, ldot_index_t** out) {
    LDOT_REQUIRE(path != nullptr && out != nullptr, LDOT_EINVAL, "NULL argument");
    FILE* f = fopen(path, "rb");
    LDOT_REQUIRE(f != nullptr, LDOT_EIO, "cannot open %s", path);
    char magic[8];
    int32_t d = 0;
    int64_t n = 0;
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "LDOTIDX1", 8) == 0 && fread(&d, 4, 1, f) == 1 &&
              fread(&n, 8, 1, f) == 1 && d > 0 && n >= 0;
    if (!ok) {
        fclose(f);
        set_error("%s is not an LDOTIDX1 file", path);
        return LDOT_EIO;
    }
    // the header is untrusted: the payload must be exactly n * d floats (a corrupt count must not size an allocation)
    bool size_ok = d <= 65536 && n < 0x7ffffff0ll;
    if (size_ok) {
        const long here = ftell(f);
        size_ok = here >= 0 && fseek(f, 0, SEEK_END) == 0;
        const long end = size_ok ? ftell(f) : -1;
        size_ok = size_ok && end >= here && (uint64_t)(end - here) == (uint64_t)n * (uint64_t)d * 4u && fseek(f, here, SEEK_SET) == 0;
    }
    if (!size_ok) {
        fclose(f);
        set_error("%s: header (d = %d, n = %lld) does not match the file size", path, d, (long long)n);
        return LDOT_EIO;
    }
    std::vector<float> host;
    try {
        host.resize((size_t)n * d);
    } catch (const std::bad_alloc&) {
        fclose(f);
        set_error("out of host memory loading %s", path);
        return LDOT_ENOMEM;
    }
    ok = host.empty() || fread(host.data(), sizeof(float), host.size(), f) == host.size();
    fclose(f);
    LDOT_REQUIRE(ok, LDOT_EIO, "%s is truncated", path);
    ldot_index* ix = nullptr;
    int rc = ldot_index_create(d, &ix);
    if (rc) return rc;
    rc = ldot_index_add(ix, host.data(), n, LDOT_F32, LDOT_HOST, 0, nullptr);
    if (rc) {
        ldot_index_destroy(ix);
        return rc;
    }
    *out = ix;
    return LDOT_OK;
}

// the merge of a sharded search, reading the all-to-all's receive buffer in place: nparts blocks in the layout of
// ldot_index_search_finish_blocked (one per source rank), of which the first nq (<= block_rows) queries are merged
int ldot_merge_topk_blocked(const void* blocks, int nparts, int64_t block_rows, int64_t block_bytes, int64_t nq, int k_in, int k_out,
                            float* out_scores, int64_t* out_labels, void* stream) {
    LDOT_REQUIRE(blocks && out_scores && out_labels, LDOT_EINVAL, "NULL buffer");
    LDOT_REQUIRE(nparts >= 1 && k_in >= 1 && k_out >= 1 && k_out <= kMaxKp && nq >= 0 && nq <= block_rows, LDOT_EINVAL, "bad sizes");
    const int64_t lab_off = LDOT_BLOCK_LABELS_OFFSET(block_rows, k_in);
    LDOT_REQUIRE(block_bytes % 16 == 0 && block_bytes >= lab_off + block_rows * k_in * 8 && ((uintptr_t)blocks & 15) == 0, LDOT_EINVAL,
                 "bad block geometry");
    if (nq == 0) return LDOT_OK;
    return launch_select_lists((const float*)blocks, (const int64_t*)((const char*)blocks + lab_off), block_bytes / 4, block_bytes / 8, nparts,
                               k_in, nq, k_out, out_scores, out_labels, (hipStream_t)stream);
}

int ldot_merge_topk(const float* scores, const int64_t* labels, int nparts, int64_t nq, int k_in, int k_out,
                    float* out_scores, int64_t* out_labels, int mem, void* stream) {
    LDOT_REQUIRE(scores && labels && out_scores && out_labels, LDOT_EINVAL, "NULL buffer");
    LDOT_REQUIRE(nparts >= 1 && k_in >= 1 && k_out >= 1 && k_out <= kMaxKp && nq >= 0, LDOT_EINVAL, "bad sizes");
    if (nq == 0) return LDOT_OK;
    hipStream_t st = (hipStream_t)stream;
    if (mem == LDOT_DEVICE)
        return launch_select_lists(scores, labels, nq * k_in, nq * k_in, nparts, k_in, nq, k_out, out_scores, out_labels, st);
    LDOT_REQUIRE(mem == LDOT_HOST, LDOT_EINVAL, "bad mem");
    const size_t n_in = (size_t)nparts * nq * k_in, n_out = (size_t)nq * k_out;
    // host-side callers: a grow-only per-thread device workspace (released when the thread exits), no hipMalloc/hipFree per call.
    // The buffers belong to the device they were allocated on: a thread that merges on another device afterwards gets fresh ones
    // (device memory of GPU 0 handed to a kernel on GPU 1's stream would fault, or crawl through peer access).
    struct Ws {
        DevBuf b[4];
        int device = -1;
        void release_all() {
            if (device < 0) return;
            DeviceGuard g(device);
            for (DevBuf& x : b) x.release();
            device = -1;
        }
        ~Ws() { release_all(); }
    };
    static thread_local Ws ws;
    int cur_dev = 0;
    LDOT_HIP_CHECK(hipGetDevice(&cur_dev));
    if (ws.device != cur_dev) {
        ws.release_all();
        ws.device = cur_dev;
    }
    int rc = LDOT_OK;
    if ((rc = ws.b[0].ensure(n_in * 4)) || (rc = ws.b[1].ensure(n_in * 8)) || (rc = ws.b[2].ensure(n_out * 4)) ||
        (rc = ws.b[3].ensure(n_out * 8)))
        return rc;
    void *ds = ws.b[0].p, *dl = ws.b[1].p, *os = ws.b[2].p, *ol = ws.b[3].p;
    if (!rc && (hipMemcpyAsync(ds, scores, n_in * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemcpyAsync(dl, labels, n_in * 8, hipMemcpyHostToDevice, st) != hipSuccess)) {
        set_error("H2D copy failed in ldot_merge_topk");
        rc = LDOT_EDEVICE;
    }
    if (!rc)
        rc = launch_select_lists((const float*)ds, (const int64_t*)dl, nq * k_in, nq * k_in, nparts, k_in, nq, k_out, (float*)os,
                                 (int64_t*)ol, st);
    if (!rc && (hipMemcpyAsync(out_scores, os, n_out * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(out_labels, ol, n_out * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)) {
        set_error("D2H copy failed in ldot_merge_topk");
        rc = LDOT_EDEVICE;
    }
    return rc;
}

int ldot_cls_pool(const void* seq, int dtype, int64_t B, int64_t stride_b, int64_t D, int normalize, float* out_f32,
                  void* out_bf16, void* stream) {
    LDOT_REQUIRE(seq != nullptr && (out_f32 || out_bf16), LDOT_EINVAL, "NULL buffer");
    LDOT_REQUIRE(B >= 0 && D > 0 && D <= 65536 && stride_b >= D, LDOT_EINVAL, "bad shape");
    return launch_convert_rows(seq, dtype, stride_b, B, B, (int)D, (int)D, normalize, out_f32, (uint16_t*)out_bf16, 0,
                               nullptr, 0, (hipStream_t)stream);
}

}  // extern "C"
