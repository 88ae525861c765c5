// C ABI of libldot.so (see include/ldot.h): index object, search orchestration, merge, pooling.
#include "index_state.h"

namespace ldot {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace ldot

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
int reshuffle_rows(ldot_index* ix, hipStream_t st) {
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
    if (ix->h_stamp) (void)hipHostFree(ix->h_stamp);
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
    unsigned long long h[2 * kSetStatSlots];
    LDOT_HIP_CHECK(hipMemcpy(h, ix->w_set_stats.p, sizeof(h), hipMemcpyDeviceToHost));   // (drains the device: a measurement aid)
    for (int s = 0; s < kSetStatSlots; ++s) {
        out[0] += (int64_t)h[2 * s];
        out[1] += (int64_t)h[2 * s + 1];
    }
    return LDOT_OK;
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
