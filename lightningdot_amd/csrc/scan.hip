// Candidate passes of a search and their adaptive state (see index_state.h): dense chunks, the narrow search, the fused scan with its
// warm-up, growth schedule, optimistic / pooled thresholds and scan order, the overflow check and the recovery of flagged queries.
#include "index_state.h"

#include <chrono>

int stream_wait(ldot_index* ix, hipStream_t st) {
    static int mode = -1;   // 0: the runtime's wait, 1: poll the stamp
    if (mode < 0) {
        const char* e = getenv("LDOT_HOST_WAIT");
        mode = (e && strcmp(e, "runtime") == 0) ? 0 : 1;
    }
    if (mode == 1 && !ix->h_stamp) {
        if (hipHostMalloc((void**)&ix->h_stamp, 64, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(&ix->d_stamp, ix->h_stamp, 0) != hipSuccess || ix->d_stamp == nullptr) {
            (void)hipGetLastError();
            if (ix->h_stamp) (void)hipHostFree(ix->h_stamp);
            ix->h_stamp = nullptr;
            mode = 0;
        } else {
            *(volatile uint32_t*)ix->h_stamp = 0;
            ix->stamp_seq = 0;
        }
    }
    if (mode == 1) {
        const uint32_t seq = ++ix->stamp_seq;
        if (hipStreamWriteValue32(st, ix->d_stamp, seq, 0) == hipSuccess) {
            volatile uint32_t* w = (volatile uint32_t*)ix->h_stamp;
            const auto t0 = std::chrono::steady_clock::now();
            for (uint64_t spins = 0; *w != seq; ++spins) {
                __builtin_ia32_pause();
                if ((spins & 0xfffff) == 0xfffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) break;   // (a lost write: let the runtime decide)
            }
            if (*w == seq) {
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                return LDOT_OK;
            }
        } else {
            (void)hipGetLastError();
            mode = 0;   // this runtime does not do stream writes: its own wait from now on
        }
    }
    LDOT_HIP_CHECK(hipStreamSynchronize(st));
    return LDOT_OK;
}

int candidate_len(const ldot_index* ix, int k) {
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
void prof_collect(ldot_index* ix, hipStream_t st) {
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


// buffers of the run-maxima selection (select_narrow.hip) for up to nq queries x ldm runs; M and the counters are kept all-zero
// between searches by the kernels themselves and cleared here only after a (re)allocation or an aborted search
int narrow_buffers(ldot_index* ix, int64_t nq, int64_t ldm, hipStream_t st) {
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

bool narrow_select_ok(const ldot_index* ix, int64_t nq, int kp) {
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

// <= 64 queries over one scan chunk: everything after the scan is ONE launch (narrow_finish_kernel, one workgroup per query)
bool narrow_one_launch(const ldot_index* ix, int64_t nq, int kp) {
    int sh, nruns;
    narrow_plan(ix->ntotal, kp, &sh, &nruns);
    return nq <= kNarrowMaxQueries && ix->ntotal <= ((int64_t)1 << 22) && nruns <= 2048 && kp <= 512;
}

int narrow_search(ldot_index* ix, int64_t nq, int kp, hipStream_t st, const DirectOut* direct) {
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
int dense_scan_all(ldot_index* ix, int64_t nq, int64_t r0, int64_t r1, int kp, float* tau, bool allow_wide,
                          hipStream_t st, int64_t q_base) {
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
int fused_scan(ldot_index* ix, int64_t nq, int64_t nq_pad, int kp, hipStream_t st, int phase, int parts) {
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
bool fused_overflow_check(ldot_index* ix) {
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
int redo_flagged(ldot_index* ix, int64_t nq, int64_t nq_pad, int kp, hipStream_t st, int level) {
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
int stage_unstaged_queries(ldot_index* ix, int64_t nq, hipStream_t st) {
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
bool auto_fused(const ldot_index* ix, int64_t nq) {
    if (nq <= 16 && narrow_ok(ix, nq)) return false;
    const int64_t n = ix->ntotal;
    return n >= 32768 || (n >= 20480 && nq >= 4096) || (n >= 8192 && nq >= 16384);
}

// defer_check: enqueue a fused scan speculatively and leave the overflow check to the caller's own synchronisation point
// warm_only (ldot_index_search_warmup): stop after the local warm-up of a fused scan, leave the statistics the ranks exchange in
