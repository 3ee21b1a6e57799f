/* abi_smoke.c -- a pure C99 caller of include/zvdb_hip.h: what Zig's @cImport parses and what bindings/hnsw_hip.zig calls
 * (the reference builds its library and tests from build.zig:8-19,:53-84; there is no Zig toolchain in this image, so this
 * translation unit is the stand-in that proves the header is C, not C++, and that the library links and runs from C).
 *
 * Flow (the drop-in's life cycle, src/hnsw.zig:52-71,:73,:194): create -> add / insert_batch (the reference insert, every
 * layer) -> search_graph in reference-compat mode -> brute force on a certified screen -> options -> save / load -> destroy.
 * Checks are self-contained (no oracle): dense ids, a point is its own nearest neighbour at distance 0, results ascending,
 * the two exact brute-force screens agree, the loaded index answers like the saved one.
 *
 * Build: gcc -std=c99 -pedantic -Wall -Wextra -Werror -I include tests/c/abi_smoke.c -L zvdb_amd -lzvdb_hip -lm
 * Exit code 0 = ok, 77 = no GPU (skipped), anything else = failure (message on stderr). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zvdb_hip.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc_ = (call);                                                                             \
        if (rc_ != ZVDB_OK) {                                                                         \
            fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, zvdb_hip_last_error()); \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)
#define EXPECT(cond)                                                              \
    do {                                                                          \
        if (!(cond)) {                                                            \
            fprintf(stderr, "%s:%d: expectation failed: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

static uint64_t mix(uint64_t *s) { /* xorshift: the caller's own data, no library involved */
    uint64_t x = *s;
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    *s = x;
    return x;
}

int main(int argc, char **argv) {
    const uint32_t dim = 48, n = 3000, k = 10, nq = 7, m = 8;
    const char *path = argc > 1 ? argv[1] : "/tmp/abi_smoke.zvdb";
    int ndev = 0;
    uint32_t i, j;
    if (zvdb_hip_abi_version() != ZVDB_HIP_ABI_VERSION) return 2;
    if (zvdb_hip_device_count(&ndev) != ZVDB_OK || ndev == 0) {
        fprintf(stderr, "no GPU: skipped\n");
        return 77;
    }
    {
        zvdb_hip_config cfg;
        zvdb_hip_index *h = NULL, *g = NULL;
        float *rows = (float *)malloc(sizeof(float) * n * dim);
        uint32_t *levels = (uint32_t *)calloc(n, sizeof(uint32_t));
        uint32_t *nrec = (uint32_t *)malloc(sizeof(uint32_t) * n);
        uint32_t *recs = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n * 2 * (4 + m));
        uint64_t *ids = (uint64_t *)malloc(sizeof(uint64_t) * nq * k), *ids2 = (uint64_t *)malloc(sizeof(uint64_t) * nq * k);
        float *dist = (float *)malloc(sizeof(float) * nq * k), *dist2 = (float *)malloc(sizeof(float) * nq * k);
        uint32_t cnt[7], cnt2[7];
        uint64_t first = 99, count = 0, seed = 88172645463325252ull, row_bytes = 0, resident = 0;
        uint32_t got_dim = 0;
        zvdb_hip_search_stats st;
        zvdb_hip_cert_info ci;
        EXPECT(rows && levels && nrec && recs && ids && ids2 && dist && dist2);
        for (i = 0; i < n * dim; ++i) rows[i] = (float)(mix(&seed) >> 40) * (1.0f / 16777216.0f);
        for (i = 0; i < n; ++i) levels[i] = (mix(&seed) & 3u) == 0 ? 1u : 0u; /* the binding draws the levels (hnsw.zig:172-180) */
        memset(&cfg, 0, sizeof(cfg));
        cfg.abi_version = ZVDB_HIP_ABI_VERSION;
        cfg.metric = ZVDB_METRIC_L2SQ;
        cfg.m = m;
        cfg.ef_construction = 200;
        cfg.device = -1;
        cfg.flags = 0;
        CHECK(zvdb_hip_create(&cfg, &h)); /* init(m, ef_construction): no dimension yet (hnsw.zig:52) */
        CHECK(zvdb_hip_count(h, &count));
        EXPECT(count == 0);
        /* the reference insert, every layer, n points in one call (ids dense from 0: hnsw.zig:77) */
        CHECK(zvdb_hip_insert_batch(h, rows, dim, n, levels, &first, nrec, recs, (uint64_t)n * 2));
        EXPECT(first == 0);
        CHECK(zvdb_hip_count(h, &count));
        CHECK(zvdb_hip_dim(h, &got_dim));
        EXPECT(count == n && got_dim == dim);
        EXPECT(nrec[0] == 0); /* the first node links to nothing (hnsw.zig:110-112) */
        for (i = 1; i < n; ++i) EXPECT(nrec[i] >= 1 && nrec[i] <= levels[i] + 1);
        /* search (hnsw.zig:194-236), reference-compat mode, queries = rows of the index: each finds itself first at distance 0 */
        CHECK(zvdb_hip_search_graph(h, rows, nq, dim, k, 0, 0, ids, dist, cnt, &st));
        EXPECT(st.dist_evals > 0 && st.pops > 0);
        for (i = 0; i < nq; ++i) {
            EXPECT(cnt[i] >= 1 && cnt[i] <= k);
            for (j = 1; j < cnt[i]; ++j) EXPECT(dist[i * k + j - 1] <= dist[i * k + j]);
        }
        EXPECT(ids[0] == 0 && dist[0] == 0.0f); /* the entry point is node 0 (hnsw.zig:84): query 0 pops it first */
        /* a dimension mismatch is the reference's @panic (hnsw.zig:183-185), reported as a status */
        EXPECT(zvdb_hip_search_graph(h, rows, 1, dim - 1, k, 0, 0, ids, dist, cnt, NULL) == ZVDB_ERR_DIM_MISMATCH);
        /* brute force on two certified screens: same ids, same exact distances, self first */
        CHECK(zvdb_hip_search_bruteforce(h, rows, nq, dim, k, ZVDB_BF_F32, ids, dist, cnt));
        CHECK(zvdb_hip_search_bruteforce(h, rows, nq, dim, k, ZVDB_BF_I8, ids2, dist2, cnt2));
        for (i = 0; i < nq; ++i) {
            EXPECT(cnt[i] == k && cnt2[i] == k && ids[i * k] == i && dist[i * k] == 0.0f);
            for (j = 0; j < k; ++j) EXPECT(ids[i * k + j] == ids2[i * k + j] && dist[i * k + j] == dist2[i * k + j]);
        }
        CHECK(zvdb_hip_bruteforce_cert_info(h, &ci));
        EXPECT(ci.enabled == 1 && ci.queries == 2 * nq);
        CHECK(zvdb_hip_index_set_option(h, ZVDB_OPT_CERTIFY, 0));
        EXPECT(zvdb_hip_index_set_option(h, 12345u, 0) == ZVDB_ERR_INVALID);
        CHECK(zvdb_hip_index_set_option(h, ZVDB_OPT_CERTIFY, 1));
        CHECK(zvdb_hip_storage_info(h, &row_bytes, &resident));
        EXPECT(row_bytes == 64 * 4 && resident > 0); /* 48 floats padded to 64 */
        /* on-disk format: the loaded index answers like the saved one */
        CHECK(zvdb_hip_save(h, path));
        CHECK(zvdb_hip_load(path, -1, &g));
        CHECK(zvdb_hip_search_graph(g, rows, nq, dim, k, 0, 0, ids2, dist2, cnt2, NULL));
        CHECK(zvdb_hip_search_graph(h, rows, nq, dim, k, 0, 0, ids, dist, cnt, NULL));
        for (i = 0; i < nq; ++i) {
            EXPECT(cnt[i] == cnt2[i]);
            for (j = 0; j < cnt[i]; ++j) EXPECT(ids[i * k + j] == ids2[i * k + j] && dist[i * k + j] == dist2[i * k + j]);
        }
        zvdb_hip_destroy(g);
        zvdb_hip_destroy(h); /* deinit (hnsw.zig:64-71) */
        remove(path);
        free(rows); free(levels); free(nrec); free(recs); free(ids); free(ids2); free(dist); free(dist2);
    }
    printf("abi_smoke ok\n");
    return 0;
}
