// The search entry points of the C ABI (include/ldot.h): ingest -> candidate pass (scan.hip) -> exchange hooks of the sharded search ->
// exact re-score + output.
#include "index_state.h"

extern "C" {

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
            if ((rc = stream_wait(ix, st))) return rc;
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
                if ((rc = stream_wait(ix, st))) return rc;
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
        if ((rc = stream_wait(ix, st))) return rc;
        if (fused_overflow_check(ix) && (rc = redo_flagged(ix, nq, nq_pad, kp, st))) return rc;
    } else if (path == 2) {
        if (ix->ntotal > 0 && (rc = dense_scan_all(ix, nq, 0, ix->ntotal, kp, tau, true, st))) return rc;
    } else {
        if (stat_in && (rc = launch_apply_stats(nq, stat_in, tau, st))) return rc;
        if ((rc = fused_scan(ix, nq, nq_pad, kp, st, 2, stat_in ? ix->split_parts : 1))) return rc;
        if ((rc = stream_wait(ix, st))) return rc;
        if (fused_overflow_check(ix) && (rc = redo_flagged(ix, nq, nq_pad, kp, st))) return rc;
    }
    if (tau_out) LDOT_HIP_CHECK(hipMemcpyAsync(tau_out, tau, (size_t)nq * 4, hipMemcpyDeviceToDevice, st));
    return LDOT_OK;
}

}  // extern "C"

// re-score + output
int search_finish_impl(ldot_index_t* ix, const float* floor, float* out_scores, int64_t* out_labels, int out_mem,
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
        if ((rc = ix->w_set_stats.ensure(16 * kSetStatSlots))) return rc;
        LDOT_HIP_CHECK(hipMemsetAsync(ix->w_set_stats.p, 0, 16 * kSetStatSlots, st));
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
        if ((rc = stream_wait(ix, st))) return rc;
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

extern "C" {

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
        if ((rc = stream_wait(ix, st))) return rc;
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
    if (out_mem == LDOT_DEVICE && (rc = stream_wait(ix, st))) return rc;
    if (fused_overflow_check(ix)) {   // unfriendly row order: the flagged queries are searched again (redo_flagged), the rest re-scored as is
        if ((rc = redo_flagged(ix, nq, round_up(nq, kBM), ix->pend_kp, st))) return rc;
        return search_finish_impl(ix, nullptr, out_scores, out_labels, out_mem, false, st);
    }
    ix->pend_nq = 0;
    return LDOT_OK;
}

}  // extern "C"
