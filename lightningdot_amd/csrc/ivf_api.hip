// The approximate (inverted-file) search behind ldot_index_search_lists / ldot_ivf_search (include/ldot.h).
#include "index_state.h"

extern "C" {

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

}  // extern "C"
