// lds_state.h -- per-query state a wave keeps in LDS, shared by the translation units of the graph kernels (exact_kernels.hip:
// reference-compat search and inserts; ef_search.hip: ef-mode beam search and entry descent; rerank.hip): the visited sets (plain
// 32-bit open addressing, any-size tables, the 21-bit packed entries of the large beams) and the launch helpers for dynamic LDS.
#pragma once
#include "common.h"

namespace zvdb {

inline int set_dyn_lds(const void *fn, size_t bytes) {
    if (bytes > 160 * 1024) {
        set_error("per-query search state needs %zu B of LDS (> 160 KiB): k*m too large", bytes);
        return ZVDB_ERR_UNSUPPORTED;
    }
    if (bytes > 48 * 1024) {
        ZV_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    }
    return ZVDB_OK;
}

inline uint32_t bits_for(uint64_t entries) {  // table with >= 2x entries, at least 64 slots
    uint32_t b = 6;
    while ((1ull << b) < entries * 2) ++b;
    return b;
}

// ---- LDS-resident per-query state helpers (one wave owns them) ------------------
// Visited set: open addressing, key = id + 1 (0 = empty), linear probing.
__device__ __forceinline__ uint32_t hash_slot(uint32_t id, uint32_t bits) {
    return (id * 2654435761u) >> (32 - bits);
}
__device__ __forceinline__ bool hash_contains(const uint32_t *tab, uint32_t bits, uint32_t id) {
    const uint32_t mask = (1u << bits) - 1;
    uint32_t s = hash_slot(id, bits);
    for (;;) {
        const uint32_t v = tab[s];
        if (v == 0) return false;
        if (v == id + 1) return true;
        s = (s + 1) & mask;
    }
}
// wave-uniform insert (every lane executes it with the same id)
__device__ __forceinline__ void hash_insert_uniform(uint32_t *tab, uint32_t bits, uint32_t id) {
    const uint32_t mask = (1u << bits) - 1;
    uint32_t s = hash_slot(id, bits);
    for (;;) {
        const uint32_t v = tab[s];
        if (v == 0) { tab[s] = id + 1; return; }
        if (v == id + 1) return;
        s = (s + 1) & mask;
    }
}
// per-lane concurrent insert; returns true if THIS lane inserted the id (first owner)
__device__ __forceinline__ bool hash_insert_atomic(uint32_t *tab, uint32_t bits, uint32_t id) {
    const uint32_t mask = (1u << bits) - 1;
    uint32_t s = hash_slot(id, bits);
    for (;;) {
        const uint32_t old = atomicCAS(&tab[s], 0u, id + 1);
        if (old == 0) return true;
        if (old == id + 1) return false;
        s = (s + 1) & mask;
    }
}

// The same table with ANY slot count (k_search_ef_reg: the LDS a query's table takes sets the queries per CU, and a power of two
// wastes up to half of it): home slot = high half of hash x size, probing wraps at size.
__device__ __forceinline__ uint32_t hashn_slot(uint32_t id, uint32_t size) { return __umulhi(id * 2654435761u, size); }
__device__ __forceinline__ void hashn_insert_uniform(uint32_t *tab, uint32_t size, uint32_t id) {
    uint32_t s = hashn_slot(id, size);
    for (;;) {
        const uint32_t v = tab[s];
        if (v == 0) { tab[s] = id + 1; return; }
        if (v == id + 1) return;
        s = s + 1 == size ? 0 : s + 1;
    }
}
__device__ __forceinline__ bool hashn_insert_atomic(uint32_t *tab, uint32_t size, uint32_t id) {
    uint32_t s = hashn_slot(id, size);
    for (;;) {
        const uint32_t old = atomicCAS(&tab[s], 0u, id + 1);
        if (old == 0) return true;
        if (old == id + 1) return false;
        s = s + 1 == size ? 0 : s + 1;
    }
}

// Visited set of the LARGE beams (k_search_ef_reg with 8 / 16 beam blocks: ef > 256), r4: 21-bit entries, three to a 64-bit word.
// The table's LDS decides how many queries a CU holds (a third of HBM at ef = 512 with two, an eighth at ef = 1,024 with one), and a
// 32-bit slot spends a third of its bits on what the slot's position already says.  A node id is first sent through a BIJECTION
// of [0, 2^b) (b = bits of the row count: multiply by an odd constant, xor with the own right shift, twice); the low r = max(0, b -
// 20) bits of the image pick one of 2^r regions of the table, the remaining <= 20 bits (+ 1; 0 = empty) are what is stored, and an
// entry never leaves its region (bucketised linear probing over the region's words, wrapping inside it).  (region, stored bits)
// give the image and the image gives the id, so membership is EXACT -- no fingerprint collisions -- at 21.3 bits per entry.  A region
// that fills up (cw words, all three fields taken) reports 2: the caller hands the query to the next pass like any full table.
// Fields of a word fill in the order 0, 1, 2 and are never cleared; lanes of the wave meet in one 64-bit compare-and-swap.
__device__ __forceinline__ uint32_t hp_mix(uint32_t id, uint32_t bmask, uint32_t half) {
    uint32_t h = (id * 0x9E3779B1u) & bmask;
    h ^= h >> half;
    h = (h * 0x85EBCA6Bu) & bmask;
    h ^= h >> half;
    return h;
}
// 0: the id was there; 1: THIS lane put it in; 2: its region is full
__device__ __forceinline__ uint32_t hp_insert_atomic(unsigned long long *tab, uint32_t cw, uint32_t r, uint32_t bmask, uint32_t half, uint32_t id) {
    const uint32_t h = hp_mix(id, bmask, half);
    const uint32_t quo = h >> r;  // < 2^20
    const unsigned long long tag = (unsigned long long)quo + 1ull;
    unsigned long long *reg = tab + (size_t)(h & ((1u << r) - 1u)) * cw;
    uint32_t w = __umulhi(quo << 12, cw);  // home word: quo cw / 2^20
    for (uint32_t n = 0; n < cw; ++n) {
        unsigned long long v = reg[w];
        for (;;) {
            const uint32_t f0 = (uint32_t)v & 0x1FFFFFu, f1 = (uint32_t)(v >> 21) & 0x1FFFFFu, f2 = (uint32_t)(v >> 42) & 0x1FFFFFu;
            if (f0 == (uint32_t)tag || f1 == (uint32_t)tag || f2 == (uint32_t)tag) return 0;
            const int sh = f0 == 0 ? 0 : f1 == 0 ? 21 : f2 == 0 ? 42 : -1;
            if (sh < 0) break;  // the word is full: on to the next
            const unsigned long long old = atomicCAS(&reg[w], v, v | (tag << sh));
            if (old == v) return 1;
            v = old;  // another lane's entry went into this word meanwhile: look again
        }
        w = w + 1 == cw ? 0 : w + 1;
    }
    return 2;
}

}  // namespace zvdb
