/*
 * zvdb_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See zvdb_oracle.h for the usage contract and the parity-pin status.
 *
 * Build: gcc -O3 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  The
 * reference (Zig) emits strict, un-contracted, un-reassociated fp32, so this
 * file must never be compiled with -ffast-math or FMA contraction.
 *
 * Every function cites the reference lines (relative to /root/reference) it
 * restates.  No reference source text is copied: the reference is Zig, this
 * is C with arrays instead of hash maps.
 */
#include "zvdb_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* a1: distance -- src/hnsw.zig:182-192                                      */
/* ------------------------------------------------------------------------ */
float orc_distance_f32(const float *a, const float *b, size_t d) {
    float sum = 0.0f;                      /* hnsw.zig:186 */
    for (size_t i = 0; i < d; ++i) {       /* hnsw.zig:187 */
        const float diff = a[i] - b[i];    /* hnsw.zig:188 */
        sum += diff * diff;                /* hnsw.zig:189 (mul rounded, then add rounded) */
    }
    return sum;                            /* hnsw.zig:191: squared distance, no sqrt */
}

double orc_distance_f64(const double *a, const double *b, size_t d) {
    double sum = 0.0;
    for (size_t i = 0; i < d; ++i) {
        const double diff = a[i] - b[i];
        sum += diff * diff;
    }
    return sum;
}

int32_t orc_distance_i32(const int32_t *a, const int32_t *b, size_t d) {
    /* HNSW(i32) (test_hnsw.zig:244): overflow is a Zig panic/UB; inputs small. */
    int32_t sum = 0;
    for (size_t i = 0; i < d; ++i) {
        const int32_t diff = a[i] - b[i];
        sum += diff * diff;
    }
    return sum;
}

float orc_distance_blocked16_f32(const float *a, const float *b, size_t d) {
    float p[16];
    for (int l = 0; l < 16; ++l) p[l] = 0.0f;
    const size_t nchunks = (d + 3) / 4;
    for (size_t c = 0; c < nchunks; ++c) {
        const int l = (int)(c & 15);
        for (size_t j = 0; j < 4; ++j) {
            const size_t i = 4 * c + j;
            if (i < d) {
                const float diff = a[i] - b[i];
                p[l] += diff * diff;
            }
        }
    }
    for (int off = 8; off > 0; off >>= 1) {
        float t[16];
        for (int l = 0; l < 16; ++l) t[l] = p[l] + p[l ^ off];
        for (int l = 0; l < 16; ++l) p[l] = t[l];
    }
    return p[0];
}

/* the same 16-way blocked order for the inner-product metrics: -(sum a*b) (extension; see orc_hnsw_search_ef_ex) */
float orc_neg_dot_blocked16_f32(const float *a, const float *b, size_t d) {
    float p[16];
    for (int l = 0; l < 16; ++l) p[l] = 0.0f;
    const size_t nchunks = (d + 3) / 4;
    for (size_t c = 0; c < nchunks; ++c) {
        const int l = (int)(c & 15);
        for (size_t j = 0; j < 4; ++j) {
            const size_t i = 4 * c + j;
            if (i < d) p[l] += a[i] * b[i];
        }
    }
    for (int off = 8; off > 0; off >>= 1) {
        float t[16];
        for (int l = 0; l < 16; ++l) t[l] = p[l] + p[l ^ off];
        for (int l = 0; l < 16; ++l) p[l] = t[l];
    }
    return -p[0];
}

float orc_neg_dot_f32(const float *a, const float *b, size_t d) {
    float sum = 0.0f;
    for (size_t i = 0; i < d; ++i) sum += a[i] * b[i];
    return -sum;
}

/* ------------------------------------------------------------------------ */
/* synthetic data -- distribution of shared_benchmarks.zig:53-59             */
/* ------------------------------------------------------------------------ */
uint64_t orc_mix64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_fill_uniform(uint64_t seed, uint64_t first_elem, uint64_t count, float *out) {
    for (uint64_t i = 0; i < count; ++i) {
        const uint64_t z = orc_mix64(seed, first_elem + i);
        out[i] = (float)(z >> 40) * (1.0f / 16777216.0f); /* 24 bits -> [0,1) exactly */
    }
}

float orc_bf16_round(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) { /* inf / nan: truncate, keep nan a nan */
        u = (u & 0xFFFF0000u) | ((u & 0x007FFFFFu) ? 0x00400000u : 0u);
    } else {
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    }
    float r;
    memcpy(&r, &u, 4);
    return r;
}

/* ------------------------------------------------------------------------ */
/* index storage -- Node hnsw.zig:12-42, fields :44-50                       */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint64_t *items;
    size_t len, cap;
} idlist;

typedef struct {
    float *point;   /* owned copy, hnsw.zig:24-26 */
    idlist *conn;   /* level+1 lists, hnsw.zig:19-23 */
    size_t nconn;
} onode;

struct orc_hnsw {
    onode *nodes; /* ids are dense insertion order (hnsw.zig:77) -> array, not a hash map */
    size_t count, cap;
    int has_entry;        /* entry_point: ?usize, hnsw.zig:46 */
    size_t entry_point;
    size_t max_level;     /* hnsw.zig:47 */
    size_t m;             /* hnsw.zig:48 */
    size_t ef_construction; /* stored, never read: hnsw.zig:49,:59 */
    size_t dim;           /* implied by first insert (no dim at init, hnsw.zig:52) */
    uint64_t level_seed, level_ctr;
    orc_counters ctr;
};

static int idlist_push(idlist *l, uint64_t v) {
    if (l->len == l->cap) {
        size_t nc = l->cap ? l->cap * 2 : 8;
        uint64_t *p = (uint64_t *)realloc(l->items, nc * sizeof(uint64_t));
        if (!p) return -1;
        l->items = p;
        l->cap = nc;
    }
    l->items[l->len++] = v;
    return 0;
}

orc_hnsw *orc_hnsw_create(size_t m, size_t ef_construction, uint64_t level_seed) {
    orc_hnsw *h = (orc_hnsw *)calloc(1, sizeof(orc_hnsw));
    if (!h) return NULL;
    h->m = m;
    h->ef_construction = ef_construction;
    h->level_seed = level_seed;
    return h;
}

void orc_hnsw_destroy(orc_hnsw *h) {
    if (!h) return;
    for (size_t i = 0; i < h->count; ++i) {
        for (size_t l = 0; l < h->nodes[i].nconn; ++l) free(h->nodes[i].conn[l].items);
        free(h->nodes[i].conn);
        free(h->nodes[i].point);
    }
    free(h->nodes);
    free(h);
}

/* randomLevel -- hnsw.zig:172-180: geometric(p = 0.5), capped at 31. */
static size_t orc_random_level(orc_hnsw *h) {
    size_t level = 0;
    while (level < 31) {
        const uint64_t z = orc_mix64(h->level_seed ^ 0x1EE7C0DEull, h->level_ctr++);
        const float f = (float)(z >> 40) * (1.0f / 16777216.0f);
        if (!(f < 0.5f)) break;
        level += 1;
    }
    return level;
}

/* shrinkConnections -- hnsw.zig:143-170 */
static int orc_shrink(orc_hnsw *h, size_t node_id, size_t level) {
    onode *node = &h->nodes[node_id];
    idlist *c = &node->conn[level];
    if (c->len <= h->m) return 0;                       /* :146 */
    const size_t n = c->len;
    uint64_t *cand = (uint64_t *)malloc(n * sizeof(uint64_t)); /* :148-150 */
    if (!cand) return -1;
    memcpy(cand, c->items, n * sizeof(uint64_t));
    /* std.sort.insertion (stable; Zig 0.13 std/sort.zig insertionContext:
     * for i in 1..n: j = i; while j > 0 and less(j, j-1): swap(j, j-1); j -= 1).
     * The comparator recomputes both distances on every call (:160-161). */
    for (size_t i = 1; i < n; ++i) {
        size_t j = i;
        while (j > 0) {
            const float da = orc_distance_f32(node->point, h->nodes[cand[j]].point, h->dim);
            const float db = orc_distance_f32(node->point, h->nodes[cand[j - 1]].point, h->dim);
            h->ctr.shrink_dist_evals += 2;
            if (!(da < db)) break;                      /* :162 strict */
            const uint64_t t = cand[j];
            cand[j] = cand[j - 1];
            cand[j - 1] = t;
            --j;
        }
    }
    c->len = h->m;                                      /* :168 */
    memcpy(c->items, cand, h->m * sizeof(uint64_t));    /* :169 */
    free(cand);
    return 0;
}

/* connect -- hnsw.zig:119-141 */
static int orc_connect(orc_hnsw *h, size_t source, size_t target, size_t level) {
    if (source >= h->count || target >= h->count) return -4; /* error.NodeNotFound :120-121 */
    onode *s = &h->nodes[source];
    onode *t = &h->nodes[target];
    if (level < s->nconn) { if (idlist_push(&s->conn[level], target)) return -1; } /* :128-130 */
    if (level < t->nconn) { if (idlist_push(&t->conn[level], source)) return -1; } /* :131-133 */
    if (level < s->nconn) { if (orc_shrink(h, source, level)) return -1; }         /* :135-137 */
    if (level < t->nconn) { if (orc_shrink(h, target, level)) return -1; }         /* :138-140 */
    return 0;
}

/* insert -- hnsw.zig:73-117 */
int orc_hnsw_insert(orc_hnsw *h, const float *point, size_t d, int level_in) {
    if (h->count == 0 && h->dim == 0) h->dim = d;
    if (d != h->dim) return -2; /* hnsw.zig:183-185 would @panic on first distance */

    const size_t id = h->count;                                      /* :77 */
    const size_t level = level_in < 0 ? orc_random_level(h) : (size_t)level_in; /* :78 */

    if (h->count == h->cap) {
        size_t nc = h->cap ? h->cap * 2 : 1024;
        onode *p = (onode *)realloc(h->nodes, nc * sizeof(onode));
        if (!p) return -1;
        h->nodes = p;
        h->cap = nc;
    }
    onode nd;                                                         /* Node.init :18-33 */
    nd.nconn = level + 1;
    nd.conn = (idlist *)calloc(nd.nconn, sizeof(idlist));
    nd.point = (float *)malloc((d ? d : 1) * sizeof(float));
    if (!nd.conn || !nd.point) { free(nd.conn); free(nd.point); return -1; }
    memcpy(nd.point, point, d * sizeof(float));
    h->nodes[id] = nd;                                                /* :82 */
    h->count += 1;

    if (h->has_entry) {                                               /* :84 */
        size_t ep = h->entry_point;                                   /* :85 */
        float curr_dist = orc_distance_f32(nd.point, h->nodes[ep].point, d); /* :86 */
        h->ctr.insert_dist_evals += 1;
        for (size_t layer = 0; layer <= h->max_level; ++layer) {      /* :88 ascending, pre-insert max_level */
            int changed = 1;
            while (changed) {                                         /* :90 */
                changed = 0;
                const onode *curr = &h->nodes[ep];                    /* :92 node current at top of iteration */
                if (layer < curr->nconn) {                            /* :93 */
                    const idlist *lst = &curr->conn[layer];
                    for (size_t i = 0; i < lst->len; ++i) {           /* :94 whole list */
                        const size_t nb = (size_t)lst->items[i];
                        const float dist = orc_distance_f32(nd.point, h->nodes[nb].point, d); /* :96 */
                        h->ctr.insert_dist_evals += 1;
                        if (dist < curr_dist) {                       /* :97 strict */
                            ep = nb;
                            curr_dist = dist;
                            changed = 1;
                        }
                    }
                }
            }
            if (layer <= level) {                                     /* :106 */
                const int rc = orc_connect(h, id, ep, layer);         /* :107 */
                if (rc) return rc;
            }
        }
    } else {
        h->has_entry = 1;                                             /* :110-112 */
        h->entry_point = id;
    }
    if (level > h->max_level) h->max_level = level;                   /* :114-116 */
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Zig std.PriorityQueue restated (un-vendored dependency: Zig std >= 0.13,  */
/* std/priority_queue.zig).  Published algorithm: array binary min-heap;     */
/*   add    = append, then siftUp: move the hole up while child < parent;    */
/*   remove = take items[0], move the last item to the root, siftDown: pick  */
/*            the right child only if right < left, stop if elem < child.    */
/* Only matters for EXACTLY equal distances (parity unpinned, see header).   */
/* ------------------------------------------------------------------------ */
typedef struct { uint64_t id; float distance; } cand_t; /* CandidateNode hnsw.zig:238-245 */

typedef struct { cand_t *items; size_t len, cap; } pqueue;

static int pq_add(pqueue *q, cand_t e) {
    if (q->len == q->cap) {
        size_t nc = q->cap ? q->cap * 2 : 64;
        cand_t *p = (cand_t *)realloc(q->items, nc * sizeof(cand_t));
        if (!p) return -1;
        q->items = p;
        q->cap = nc;
    }
    size_t child = q->len++;
    while (child > 0) {
        const size_t parent = (child - 1) >> 1;
        if (!(e.distance < q->items[parent].distance)) break; /* order(child,parent) != .lt */
        q->items[child] = q->items[parent];
        child = parent;
    }
    q->items[child] = e;
    return 0;
}

static cand_t pq_remove(pqueue *q) {
    const cand_t top = q->items[0];
    const cand_t last = q->items[q->len - 1];
    q->len -= 1;
    if (q->len == 0) return top;
    size_t idx = 0;
    for (;;) {
        size_t lesser = (idx * 2) | 1;
        if (!(lesser < q->len)) break;
        const size_t next = lesser + 1;
        if (next < q->len && q->items[next].distance < q->items[lesser].distance) lesser = next;
        if (last.distance < q->items[lesser].distance) break;
        q->items[idx] = q->items[lesser];
        idx = lesser;
    }
    q->items[idx] = last;
    return top;
}

/* search -- hnsw.zig:194-236.  ctr: the index's counters, or NULL (orc_hnsw_search_many: several threads search one index) */
static inline size_t hnsw_search_body(const orc_hnsw *h, orc_counters *ctr, const float *query, size_t d, size_t k,
                                      uint64_t *out_ids, float *out_dist, unsigned char *scratch) {
    if (!h->has_entry || k == 0) return 0;         /* :201; k = 0 -> loop never runs */
    if (d != h->dim) return 0;
    size_t nres = 0;
    pqueue q = {0, 0, 0};
    /* :205 (set of ids).  scratch: a caller-owned all-zero array of h->count bytes, handed back all-zero (orc_hnsw_search_many: a
     * calloc / free of it per query serialises threads in the allocator and the kernel's page handling) */
    unsigned char *visited = scratch ? scratch : (unsigned char *)calloc(h->count, 1);
    if (!visited) return 0;

    cand_t seed;
    seed.id = h->entry_point;
    seed.distance = orc_distance_f32(query, h->nodes[h->entry_point].point, d); /* :208 */
    if (ctr) ctr->search_dist_evals += 1;
    pq_add(&q, seed);
    visited[h->entry_point] = 1;                                                 /* :209 */

    while (q.len > 0 && nres < k) {                                              /* :211 */
        const cand_t cur = pq_remove(&q);                                        /* :212 */
        if (ctr) ctr->search_pops += 1;
        out_ids[nres] = cur.id;                                                  /* :213-214 */
        out_dist[nres] = cur.distance;
        nres += 1;
        const idlist *lst = &h->nodes[cur.id].conn[0];                           /* :216 layer 0 only */
        for (size_t i = 0; i < lst->len; ++i) {
            const size_t nb = (size_t)lst->items[i];
            if (!visited[nb]) {                                                  /* :217 */
                cand_t c;
                c.id = nb;
                c.distance = orc_distance_f32(query, h->nodes[nb].point, d);     /* :219 */
                if (ctr) ctr->search_dist_evals += 1;
                pq_add(&q, c);                                                   /* :220 */
                visited[nb] = 1;                                                 /* :221 */
            }
        }
    }
    if (scratch) {  /* every id marked is one that was pushed: popped (out_ids) or still queued */
        for (size_t i = 0; i < nres; ++i) scratch[out_ids[i]] = 0;
        for (size_t i = 0; i < q.len; ++i) scratch[q.items[i].id] = 0;
    } else {
        free(visited);
    }
    free(q.items);

    /* final stable insertion sort by distance(query, point) -- :227-233.  The
     * comparator recomputes the same two values on every call (:230); the
     * recomputed value is bit-identical to the stored one, so sort on it. */
    for (size_t i = 1; i < nres; ++i) {
        size_t j = i;
        while (j > 0) {
            if (ctr) ctr->sort_dist_evals += 2;
            if (!(out_dist[j] < out_dist[j - 1])) break;
            const float td = out_dist[j]; out_dist[j] = out_dist[j - 1]; out_dist[j - 1] = td;
            const uint64_t ti = out_ids[j]; out_ids[j] = out_ids[j - 1]; out_ids[j - 1] = ti;
            --j;
        }
    }
    return nres;                                                                 /* :235 */
}

size_t orc_hnsw_search(orc_hnsw *h, const float *query, size_t d, size_t k,
                       uint64_t *out_ids, float *out_dist) {
    return hnsw_search_body(h, &h->ctr, query, d, k, out_ids, out_dist, NULL);
}

/* nq searches in one call, read-only (no counters): what one thread of a query-parallel CPU baseline runs (bench.py: "beyond
 * the reference", whose search takes one global mutex, hnsw.zig:195-196).  out arrays: nq * k entries, out_n: nq. */
void orc_hnsw_search_many(const orc_hnsw *h, const float *queries, size_t nq, size_t d, size_t k,
                          uint64_t *out_ids, float *out_dist, uint64_t *out_n) {
    unsigned char *scratch = (unsigned char *)calloc(h->count ? h->count : 1, 1);
    for (size_t i = 0; i < nq; ++i)
        out_n[i] = scratch ? hnsw_search_body(h, NULL, queries + i * d, d, k, out_ids + i * k, out_dist + i * k, scratch) : 0;
    free(scratch);
}

/* ef-mode beam search (EXTENSION; specification in zvdb_oracle.h). */
static int lex_less(float da, uint64_t ia, float db, uint64_t ib) {
    return da < db || (da == db && ia < ib);
}

size_t orc_hnsw_search_ef_ex(orc_hnsw *h, const float *query, size_t d, size_t k, size_t ef, int metric, size_t entry,
                             uint64_t *out_ids, float *out_dist) {
    return orc_hnsw_search_ef_w(h, query, d, k, ef, metric, entry, 1, out_ids, out_dist);
}

/* The same beam search with a SEARCH WIDTH (extension of the extension; zvdb_oracle.h): every iteration takes the first `width`
 * unexpanded entries of W (in W's order) at once -- all marked expanded, each counted as a pop -- and only then walks their lists,
 * one after the other in that order.  width = 1 is orc_hnsw_search_ef_ex. */
size_t orc_hnsw_search_ef_w(orc_hnsw *h, const float *query, size_t d, size_t k, size_t ef, int metric, size_t entry, size_t width,
                            uint64_t *out_ids, float *out_dist) {
    if (width == 0) width = 1;
    if (width > 8) width = 8;
    if (!h->has_entry || k == 0 || ef == 0 || d != h->dim || entry >= h->count) return 0;
    if (ef < k) ef = k;
    float *wd = (float *)malloc(ef * sizeof(float));
    uint64_t *wi = (uint64_t *)malloc(ef * sizeof(uint64_t));
    unsigned char *wx = (unsigned char *)malloc(ef);
    unsigned char *visited = (unsigned char *)calloc(h->count, 1);
    if (!wd || !wi || !wx || !visited) { free(wd); free(wi); free(wx); free(visited); return 0; }
    size_t wn = 0;

    wd[0] = metric ? orc_neg_dot_blocked16_f32(query, h->nodes[entry].point, d) : orc_distance_blocked16_f32(query, h->nodes[entry].point, d);
    h->ctr.search_dist_evals += 1;
    wi[0] = entry; wx[0] = 0; wn = 1;
    visited[entry] = 1;

    for (;;) {
        size_t sel[8], ns = 0;
        for (size_t c = 0; c < wn && ns < width; ++c)
            if (!wx[c]) { wx[c] = 1; sel[ns++] = (size_t)wi[c]; }
        if (ns == 0) break;
        h->ctr.search_pops += ns;
        for (size_t s = 0; s < ns; ++s) {
        const idlist *lst = &h->nodes[sel[s]].conn[0];
        for (size_t i = 0; i < lst->len; ++i) {
            const size_t nb = (size_t)lst->items[i];
            if (visited[nb]) continue;
            visited[nb] = 1;
            const float dn = metric ? orc_neg_dot_blocked16_f32(query, h->nodes[nb].point, d) : orc_distance_blocked16_f32(query, h->nodes[nb].point, d);
            h->ctr.search_dist_evals += 1;
            if (wn == ef && !lex_less(dn, nb, wd[wn - 1], wi[wn - 1])) continue;
            size_t pos = wn < ef ? wn : ef - 1; /* slot that gets freed / appended */
            if (wn < ef) wn += 1;
            while (pos > 0 && lex_less(dn, nb, wd[pos - 1], wi[pos - 1])) {
                wd[pos] = wd[pos - 1]; wi[pos] = wi[pos - 1]; wx[pos] = wx[pos - 1];
                --pos;
            }
            wd[pos] = dn; wi[pos] = nb; wx[pos] = 0;
        }
        }
    }
    const size_t nres = wn < k ? wn : k;
    /* first k of W, re-evaluated exactly (a1) and ordered by (exact distance, id) */
    for (size_t i = 0; i < nres; ++i) {
        const uint64_t id = wi[i];
        const float de = metric ? orc_neg_dot_f32(query, h->nodes[id].point, d) : orc_distance_f32(query, h->nodes[id].point, d);
        size_t pos = i;
        while (pos > 0 && lex_less(de, id, out_dist[pos - 1], out_ids[pos - 1])) {
            out_dist[pos] = out_dist[pos - 1]; out_ids[pos] = out_ids[pos - 1];
            --pos;
        }
        out_dist[pos] = de; out_ids[pos] = id;
    }
    free(wd); free(wi); free(wx); free(visited);
    return nres;
}

size_t orc_hnsw_search_ef(orc_hnsw *h, const float *query, size_t d, size_t k,
                          size_t ef, uint64_t *out_ids, float *out_dist) {
    return orc_hnsw_search_ef_ex(h, query, d, k, ef, 0, h->entry_point, out_ids, out_dist);
}


/* ------------------------------------------------------------------------ */
/* bulk graph construction + hierarchical entry descent                      */
/* EXTENSION: no reference symbol -- the reference   */
/* insert links one edge per layer (hnsw.zig:106-108) and never reads         */
/* ef_construction (hnsw.zig:59).  These functions ARE the specification the  */
/* HIP builder (graph_build.hip) and k_descend must match bit for bit.        */
/* ------------------------------------------------------------------------ */
uint32_t orc_bulk_level(uint64_t seed, uint64_t id) {
    /* hnsw.zig:172-180's distribution (each further level with probability 1/2, cap 31) from a counter hash:
     * level = number of leading coin flips that came up "continue" = trailing one-bits of the hash */
    const uint64_t x = orc_mix64(seed, id);
    uint32_t level = 0;
    while (level < 31 && ((x >> level) & 1ull)) ++level;
    return level;
}

static float pair_dist(const float *a, const float *b, size_t d, int metric) {
    return metric ? orc_neg_dot_f32(a, b, d) : orc_distance_f32(a, b, d);
}

/* The HNSW neighbour-selection heuristic with keep-pruned fill.  Candidates of `base` arrive sorted ascending by
 * (distance to base, id).  Walk them in that order while fewer than cap are kept: a candidate c is kept iff
 * d(base, c) < d(c, s) for EVERY s kept so far (strict), else it is set aside; afterwards the set-aside ones fill the
 * list up to cap in their order.  Output order = kept order, then fills. */
size_t orc_graph_select(const float *V, size_t d, size_t ld, int metric, const uint64_t *cand_ids, const float *cand_dist,
                        size_t ncand, size_t cap, uint64_t *out_ids, float *out_dist) {
    size_t nk = 0, np = 0;
    uint64_t *pr_i = (uint64_t *)malloc((ncand + 1) * sizeof(uint64_t));
    float *pr_d = (float *)malloc((ncand + 1) * sizeof(float));
    for (size_t i = 0; i < ncand && nk < cap; ++i) {
        const float *c = V + cand_ids[i] * ld;
        int good = 1;
        for (size_t j = 0; j < nk; ++j) {
            const float dcs = pair_dist(c, V + out_ids[j] * ld, d, metric);
            if (!(cand_dist[i] < dcs)) { good = 0; break; }
        }
        if (good) { out_ids[nk] = cand_ids[i]; out_dist[nk] = cand_dist[i]; ++nk; }
        else { pr_i[np] = cand_ids[i]; pr_d[np] = cand_dist[i]; ++np; }
    }
    for (size_t i = 0; i < np && nk < cap; ++i) { out_ids[nk] = pr_i[i]; out_dist[nk] = pr_d[i]; ++nk; }
    free(pr_i); free(pr_d);
    return nk;
}

/* Layer-0 lists of the bulk graph.  cand_ids / cand_dist [N][ncand]: every node's ncand exact nearest neighbours
 * (self excluded) ascending by (distance, id).
 *   S[u]  = select(u, cand[u], cap M)                                  -- forward links
 *   C[v]  = S[v] u { (u, d(u,v)) : v in S[u] }, duplicates removed,     -- plus reverse links
 *           sorted by (distance, id), truncated to the first cmax
 *   L0[v] = select(v, C[v], cap M0)                                     -- degree cap with the same heuristic
 * adj [N][M0] (u32), deg [N]. */
void orc_graph_build_layer0(const float *V, size_t N, size_t d, size_t ld, int metric, const uint64_t *cand_ids,
                            const float *cand_dist, size_t ncand, size_t M, size_t M0, size_t cmax, uint32_t *adj, uint32_t *deg) {
    uint64_t *s_ids = (uint64_t *)malloc(N * M * sizeof(uint64_t));
    float *s_dist = (float *)malloc(N * M * sizeof(float));
    size_t *s_n = (size_t *)malloc(N * sizeof(size_t));
    size_t *r_n = (size_t *)calloc(N + 1, sizeof(size_t));
    for (size_t u = 0; u < N; ++u) {
        s_n[u] = orc_graph_select(V, d, ld, metric, cand_ids + u * ncand, cand_dist + u * ncand, ncand, M, s_ids + u * M, s_dist + u * M);
        for (size_t j = 0; j < s_n[u]; ++j) r_n[s_ids[u * M + j] + 1] += 1;
    }
    for (size_t v = 0; v < N; ++v) r_n[v + 1] += r_n[v];  /* CSR offsets */
    const size_t total = r_n[N];
    uint64_t *r_ids = (uint64_t *)malloc((total + 1) * sizeof(uint64_t));
    float *r_dist = (float *)malloc((total + 1) * sizeof(float));
    size_t *cur = (size_t *)malloc(N * sizeof(size_t));
    for (size_t v = 0; v < N; ++v) cur[v] = r_n[v];
    for (size_t u = 0; u < N; ++u)
        for (size_t j = 0; j < s_n[u]; ++j) {
            const size_t v = (size_t)s_ids[u * M + j];
            r_ids[cur[v]] = u; r_dist[cur[v]] = s_dist[u * M + j]; cur[v] += 1;
        }
    for (size_t v = 0; v < N; ++v) {
        const size_t nr = r_n[v + 1] - r_n[v], nc = s_n[v] + nr;
        uint64_t *ci = (uint64_t *)malloc((nc + 1) * sizeof(uint64_t));
        float *cd = (float *)malloc((nc + 1) * sizeof(float));
        size_t n = 0;
        for (size_t j = 0; j < s_n[v]; ++j) { ci[n] = s_ids[v * M + j]; cd[n] = s_dist[v * M + j]; ++n; }
        for (size_t j = 0; j < nr; ++j) {
            const uint64_t u = r_ids[r_n[v] + j];
            int dup = 0;
            for (size_t t = 0; t < s_n[v]; ++t) if (s_ids[v * M + t] == u) { dup = 1; break; }
            if (!dup) { ci[n] = u; cd[n] = r_dist[r_n[v] + j]; ++n; }
        }
        for (size_t i = 1; i < n; ++i) {  /* insertion sort by (distance, id) */
            const uint64_t ti = ci[i]; const float td = cd[i];
            size_t pos = i;
            while (pos > 0 && lex_less(td, ti, cd[pos - 1], ci[pos - 1])) { ci[pos] = ci[pos - 1]; cd[pos] = cd[pos - 1]; --pos; }
            ci[pos] = ti; cd[pos] = td;
        }
        if (n > cmax) n = cmax;
        uint64_t *oi = (uint64_t *)malloc((M0 + 1) * sizeof(uint64_t));
        float *od = (float *)malloc((M0 + 1) * sizeof(float));
        const size_t k = orc_graph_select(V, d, ld, metric, ci, cd, n, M0, oi, od);
        for (size_t j = 0; j < k; ++j) adj[v * M0 + j] = (uint32_t)oi[j];
        deg[v] = (uint32_t)k;
        free(ci); free(cd); free(oi); free(od);
    }
    free(s_ids); free(s_dist); free(s_n); free(r_n); free(r_ids); free(r_dist); free(cur);
}

/* Hierarchical entry descent: from `top` (the bulk graph's entry: a node of the highest level) go down the layers
 * max_level .. 1; on each layer repeat { scan the WHOLE list of the current node in list order with the traversal
 * distance (16-way blocked, as the beam search), keep the strict arg-min ('<': the earliest entry wins ties), move there }
 * until no neighbour is closer (hnsw.zig:90-104's greedy step, run top-down).  Returns the layer-0 entry of the beam. */
size_t orc_hnsw_descend(orc_hnsw *h, const float *query, size_t d, int metric, size_t top) {
    if (!h->has_entry || d != h->dim || top >= h->count) return 0;
    size_t cur = top;
    float dcur = metric ? orc_neg_dot_blocked16_f32(query, h->nodes[cur].point, d) : orc_distance_blocked16_f32(query, h->nodes[cur].point, d);
    h->ctr.search_dist_evals += 1;
    for (size_t layer = h->nodes[top].nconn - 1; layer >= 1; --layer) {
        for (;;) {
            if (layer >= h->nodes[cur].nconn) break;
            const idlist *lst = &h->nodes[cur].conn[layer];
            size_t best = cur;
            float dbest = dcur;
            for (size_t i = 0; i < lst->len; ++i) {
                const size_t nb = (size_t)lst->items[i];
                const float dn = metric ? orc_neg_dot_blocked16_f32(query, h->nodes[nb].point, d) : orc_distance_blocked16_f32(query, h->nodes[nb].point, d);
                h->ctr.search_dist_evals += 1;
                if (dn < dbest) { dbest = dn; best = nb; }
            }
            if (best == cur) break;
            cur = best; dcur = dbest;
        }
    }
    return cur;
}

/* ------------------------------------------------------------------------ */
/* accessors                                                                 */
/* ------------------------------------------------------------------------ */
size_t orc_hnsw_count(const orc_hnsw *h) { return h->count; } /* nodes.count(), test_hnsw.zig:198 */
size_t orc_hnsw_dim(const orc_hnsw *h) { return h->dim; }
size_t orc_hnsw_max_level(const orc_hnsw *h) { return h->max_level; }
int orc_hnsw_node_level(const orc_hnsw *h, size_t id) {
    return id < h->count ? (int)h->nodes[id].nconn - 1 : -1;
}
size_t orc_hnsw_neighbors(const orc_hnsw *h, size_t id, size_t layer, uint64_t *out, size_t cap) {
    if (id >= h->count || layer >= h->nodes[id].nconn) return (size_t)-1;
    const idlist *l = &h->nodes[id].conn[layer];
    for (size_t i = 0; i < l->len && i < cap; ++i) out[i] = l->items[i];
    return l->len;
}
const float *orc_hnsw_point(const orc_hnsw *h, size_t id) {
    return id < h->count ? h->nodes[id].point : NULL;
}
int orc_hnsw_set_neighbors(orc_hnsw *h, size_t id, size_t layer, const uint64_t *ids, size_t n) {
    if (id >= h->count || layer >= h->nodes[id].nconn) return -4;
    idlist *l = &h->nodes[id].conn[layer];
    l->len = 0;
    for (size_t i = 0; i < n; ++i) if (idlist_push(l, ids[i])) return -1;
    return 0;
}
void orc_hnsw_counters(const orc_hnsw *h, orc_counters *out) { *out = h->ctr; }
void orc_hnsw_reset_counters(orc_hnsw *h) { memset(&h->ctr, 0, sizeof(h->ctr)); }

/* ------------------------------------------------------------------------ */
/* a10: exact brute-force top-k (EXTENSION; parity unpinned by the reference) */
/* ------------------------------------------------------------------------ */
void orc_bruteforce_topk(const float *V, size_t N, size_t d, size_t ld,
                         const float *Q, size_t nq, size_t k, int metric,
                         uint64_t *out_ids, float *out_dist, uint32_t *out_count) {
    for (size_t qi = 0; qi < nq; ++qi) {
        const float *q = Q + qi * d;
        uint64_t *oi = out_ids + qi * k;
        float *od = out_dist + qi * k;
        size_t n = 0;
        for (size_t r = 0; r < N; ++r) {
            const float *v = V + r * ld;
            const float dist = metric == 0 ? orc_distance_f32(q, v, d) : orc_neg_dot_f32(q, v, d);
            if (n == k && !lex_less(dist, r, od[n - 1], oi[n - 1])) continue;
            size_t pos = n < k ? n : k - 1;
            if (n < k) n += 1;
            while (pos > 0 && lex_less(dist, r, od[pos - 1], oi[pos - 1])) {
                od[pos] = od[pos - 1]; oi[pos] = oi[pos - 1];
                --pos;
            }
            od[pos] = dist; oi[pos] = r;
        }
        if (out_count) out_count[qi] = (uint32_t)n;
    }
}

void orc_merge_topk(const uint64_t *ids, const float *dist, const uint32_t *count,
                    size_t G, size_t nq, size_t k,
                    uint64_t *out_ids, float *out_dist, uint32_t *out_count) {
    for (size_t qi = 0; qi < nq; ++qi) {
        uint64_t *oi = out_ids + qi * k;
        float *od = out_dist + qi * k;
        size_t n = 0;
        for (size_t g = 0; g < G; ++g) {
            const size_t base = (g * nq + qi) * k;
            const size_t c = count[g * nq + qi];
            for (size_t j = 0; j < c; ++j) {
                const float dj = dist[base + j];
                const uint64_t ij = ids[base + j];
                if (n == k && !lex_less(dj, ij, od[n - 1], oi[n - 1])) continue;
                size_t pos = n < k ? n : k - 1;
                if (n < k) n += 1;
                while (pos > 0 && lex_less(dj, ij, od[pos - 1], oi[pos - 1])) {
                    od[pos] = od[pos - 1]; oi[pos] = oi[pos - 1];
                    --pos;
                }
                od[pos] = dj; oi[pos] = ij;
            }
        }
        if (out_count) out_count[qi] = (uint32_t)n;
    }
}
