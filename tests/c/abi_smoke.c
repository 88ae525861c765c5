/* Pure-C client of the drop-in boundary (include/ldot.h): no Python, no torch — what a cgo / JNI / FFI binding would call.
 * Builds a small index from host buffers, searches, and checks the result against a naive CPU scan.
 * Build:  gcc -O2 -I include tests/c/abi_smoke.c -o abi_smoke -L lightningdot_amd -lldot -Wl,-rpath,$PWD/lightningdot_amd -lm
 * Exit code 0 = pass. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "ldot.h"

static float frand(uint64_t* s) {
    *s = *s * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((double)((*s >> 33) & 0xffffff) / 8388608.0 - 1.0);
}

int main(void) {
    const int n = 5000, d = 96, nq = 9, k = 7;
    float* x = (float*)malloc(sizeof(float) * n * d);
    float* q = (float*)malloc(sizeof(float) * nq * d);
    float* s = (float*)malloc(sizeof(float) * nq * k);
    int64_t* l = (int64_t*)malloc(sizeof(int64_t) * nq * k);
    uint64_t seed = 42;
    for (int i = 0; i < n * d; ++i) x[i] = frand(&seed);
    for (int i = 0; i < nq * d; ++i) q[i] = frand(&seed);

    if (ldot_device_count() <= 0) {
        fprintf(stderr, "no HIP device: %s\n", ldot_last_error());
        return 2;
    }
    ldot_index_t* ix = NULL;
    if (ldot_index_create(d, &ix) != LDOT_OK) { fprintf(stderr, "create: %s\n", ldot_last_error()); return 1; }
    /* two adds, like the reference's buffered index_data (faiss_indexers.py:72-77) */
    if (ldot_index_add(ix, x, 3000, LDOT_F32, LDOT_HOST, 0, NULL) != LDOT_OK ||
        ldot_index_add(ix, x + 3000 * d, n - 3000, LDOT_F32, LDOT_HOST, 0, NULL) != LDOT_OK) {
        fprintf(stderr, "add: %s\n", ldot_last_error());
        return 1;
    }
    if (ldot_index_ntotal(ix) != n) { fprintf(stderr, "ntotal\n"); return 1; }
    if (ldot_index_search(ix, q, nq, LDOT_F32, LDOT_HOST, 0, k, s, l, LDOT_HOST, NULL) != LDOT_OK) {
        fprintf(stderr, "search: %s\n", ldot_last_error());
        return 1;
    }
    /* error convention: status code + message, no exceptions across the ABI */
    if (ldot_index_search(ix, q, nq, LDOT_F32, LDOT_HOST, 0, 0, s, l, LDOT_HOST, NULL) == LDOT_OK) { fprintf(stderr, "k=0 accepted\n"); return 1; }

    int bad = 0;
    for (int i = 0; i < nq; ++i) {
        /* naive scan: best row and its score in double */
        double best = -1e300; int besti = -1;
        for (int r = 0; r < n; ++r) {
            double acc = 0;
            for (int c = 0; c < d; ++c) acc += (double)q[i * d + c] * (double)x[r * d + c];
            if (acc > best) { best = acc; besti = r; }
        }
        if (l[i * k] != besti || fabs(s[i * k] - best) > 1e-3) { fprintf(stderr, "query %d: got %lld %.6f want %d %.6f\n", i, (long long)l[i * k], s[i * k], besti, best); ++bad; }
        for (int j = 1; j < k; ++j) if (s[i * k + j] > s[i * k + j - 1]) { fprintf(stderr, "query %d not sorted\n", i); ++bad; }
    }
    /* which regime did that search run in?  (9 queries: the narrow search; nothing redone; rows stored as added) */
    int64_t reg[8];
    if (ldot_index_last_regime(ix, reg) != LDOT_OK || reg[0] != 1 || reg[3] != 0 || reg[7] != 0) {
        fprintf(stderr, "regime: path %lld redone %lld rows %lld\n", (long long)reg[0], (long long)reg[3], (long long)reg[7]);
        ++bad;
    }
    ldot_index_destroy(ix);

    /* LDOT_OPT_ROW_SHUFFLE = 1: a C caller's rows are stored in a pseudo-random order inside the library; labels, scores and
     * ldot_index_get_rows keep referring to insertion order */
    ldot_index_t* sx = NULL;
    float* s2 = (float*)malloc(sizeof(float) * nq * k);
    int64_t* l2 = (int64_t*)malloc(sizeof(int64_t) * nq * k);
    float* back = (float*)malloc(sizeof(float) * 10 * d);
    if (ldot_index_create(d, &sx) != LDOT_OK || ldot_index_set_option(sx, LDOT_OPT_ROW_SHUFFLE, 1) != LDOT_OK ||
        ldot_index_add(sx, x, 3000, LDOT_F32, LDOT_HOST, 0, NULL) != LDOT_OK ||
        ldot_index_add(sx, x + 3000 * d, n - 3000, LDOT_F32, LDOT_HOST, 0, NULL) != LDOT_OK ||
        ldot_index_search(sx, q, nq, LDOT_F32, LDOT_HOST, 0, k, s2, l2, LDOT_HOST, NULL) != LDOT_OK ||
        ldot_index_get_rows(sx, 2995, 10, back, LDOT_HOST, NULL) != LDOT_OK || ldot_index_last_regime(sx, reg) != LDOT_OK) {
        fprintf(stderr, "shuffled index: %s\n", ldot_last_error());
        return 1;
    }
    for (int i = 0; i < nq * k; ++i)
        if (l2[i] != l[i] || s2[i] != s[i]) { fprintf(stderr, "shuffled index: result %d differs (%lld %.6f vs %lld %.6f)\n", i, (long long)l2[i], s2[i], (long long)l[i], s[i]); ++bad; break; }
    for (int i = 0; i < 10 * d; ++i)
        if (back[i] != x[2995 * d + i]) { fprintf(stderr, "shuffled index: get_rows does not return rows by label\n"); ++bad; break; }
    if (reg[7] != 1) { fprintf(stderr, "shuffled index: regime says rows = %lld\n", (long long)reg[7]); ++bad; }
    if (ldot_index_set_option(sx, LDOT_OPT_ROW_SHUFFLE, 2) == LDOT_OK) { fprintf(stderr, "un-shuffling a shuffled index accepted\n"); ++bad; }
    /* LDOT_OPT_DEFER_SYNC: accepted as 0 / 1 only; with pageable outputs (malloc) a search still returns with its results in place */
    if (ldot_index_set_option(sx, LDOT_OPT_DEFER_SYNC, 2) == LDOT_OK) { fprintf(stderr, "LDOT_OPT_DEFER_SYNC = 2 accepted\n"); ++bad; }
    memset(l2, 0xff, sizeof(int64_t) * nq * k);
    if (ldot_index_set_option(sx, LDOT_OPT_DEFER_SYNC, 1) != LDOT_OK ||
        ldot_index_search(sx, q, nq, LDOT_F32, LDOT_HOST, 0, k, s2, l2, LDOT_HOST, NULL) != LDOT_OK ||
        ldot_index_set_option(sx, LDOT_OPT_DEFER_SYNC, 0) != LDOT_OK) {
        fprintf(stderr, "deferred search: %s\n", ldot_last_error());
        return 1;
    }
    for (int i = 0; i < nq * k; ++i)
        if (l2[i] != l[i] || s2[i] != s[i]) { fprintf(stderr, "LDOT_OPT_DEFER_SYNC with pageable outputs: result %d not in place\n", i); ++bad; break; }
    ldot_index_destroy(sx);
    printf(bad ? "FAIL\n" : "abi_smoke ok\n");
    return bad ? 1 : 0;
}
