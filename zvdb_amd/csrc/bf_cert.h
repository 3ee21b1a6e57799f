// bf_cert.h -- the exactness certificate of the brute-force screens (a10; no reference symbol: the semantics certified are
// "the k smallest hnsw.zig:182-192 distances, ties by lower id").
//
// A screen (fp32 / bf16 / int8 MFMA) hands the exact re-rank the k' rows with the smallest SCREEN keys.  The re-rank
// evaluates them in the reference's arithmetic and returns the best k -- which are the true top-k only if no row the screen
// left out is closer than the k-th of them.  The screen's keys carry rounding (bf16 operands, int8 codes, MFMA accumulation),
// so "k'-th screen key > k-th exact distance" proves nothing by itself.  This header turns it into a proof.
//
// L2, float keys (key = |v'|^2 / 2 - q'.v' over the rounded operands q', v'):
//   suppose a row u that is NOT among the k' survivors had a computed reference distance fl(d_u) <= d_k.  Then
//     d_u <= D := d_k / (1 - g),  g = (dim + 4) 2^-24 1.01      (sequential fp32 sum of hnsw.zig:187-190: |fl - d| <= g d)
//     |u| <= |q| + sqrt(D);  |u' - u| <= r := min(rho_max, 2^-8 (1 + 2^-7) |u|)   (rho_max: largest residual of the mirror,
//                                                                 measured when the mirror is built; RNE to bf16: 2^-8 relative)
//     |q' - u'| <= sqrt(D) + |q' - q| + r                         (triangle inequality)
//     screen key in real arithmetic  <= ((sqrt(D) + |q' - q| + r)^2 - |q'|^2) / 2
//     computed screen key            <= that + eta,  eta = (dim + 64) 2^-23 (U^2 / 2 + |q'| U) + (dim + 8) 2^-24 U^2 / 2,  U = |u| + r
//                                                     (one ulp per accumulation step in ANY order + the stored norm's own sum)
//   =: key_safe.  Every row left out has key >= the k'-th survivor's key T.  So key_safe < T contradicts the supposition:
//   no such row exists and the k returned rows are the exact answer.  The same bound prunes the re-rank itself: a survivor
//   whose key exceeds key_safe cannot enter the top-k and is not evaluated.
// dot / cosine (key = -q'.v'): no norm bound follows from the supposition, so U = the mirror's largest row norm:
//     key_safe = kappa_k + g |q| (U + rho) + |q' - q| U + |q| rho + eta,  eta = (dim + 64) 2^-23 |q'| U.
// int8 (integer keys (|c_v|^2 >> 1) - c_q.c_v over the codes; reconstruction x^ = lo + (c + 128) s, or c s for dot / cosine):
//     L2: key_safe = ((sqrt(D) + |q^ - q| + rho)^2 / s^2 - |c_q|^2) / 2 (the integer arithmetic is exact; rho: largest |u^ - u| of the mirror);
//     dot: the screen ranks by the RESCALED query q~ = f q (k_rescale_queries), kappa~ = f kappa:
//          key_safe = (f kappa_k + f g |q| (U + rho) + |q^ - q~| U + |q~| rho) / s^2 + 1.
// A query that fails the test is appended to a list; the exact scan (exact_scan.hip) recomputes its top-k over ALL rows in the
// reference's arithmetic -- no screen involved.  All certificate arithmetic is fp64 with the rounding directions noted.
#pragma once
#include "common.h"

namespace zvdb {

struct CertArgs {
    const uint32_t *kth = nullptr;   // [nq] k'-th screen key per query (0xFFFFFFFF: every row of the shard is a survivor); null: no certificate
    const uint32_t *keys = nullptr;  // [nq][kc] screen keys of the candidates (sorted ascending with the ids)
    const float *stats = nullptr;    // device: {rho_max, U_max} of the mirror the screen read
    uint32_t *nflag = nullptr;       // device counter + list of the queries that failed the test
    uint32_t *flist = nullptr;
    uint32_t *host_flag = nullptr;   // mapped host word: raised when a query fails (the next calls keep more survivors)
    int mode = 0;                    // enum zvdb_bf_mode of the screen (F32 / BF16 / I8)
    float q_lo = 0.0f, q_inv = 1.0f; // int8 quantiser
    int affine = 0;
    float rescale_target = 0.0f;     // int8, dot / cosine: queries are scaled to this largest magnitude before they are quantised
    int force = 0;                   // ZVDB_OPT_CERT_FORCE_REPAIR (tests): every query is listed for the exact scan
    uint32_t *dbg_rows = nullptr;    // tuning ZVDB_CERT_DEBUG: mapped host counters -- 1: candidate rows the re-rank evaluated; 2: cycle stamps of its phases
    int dbg_mode = 0;
};

struct CertQ {  // per-query constants (wave-uniform)
    double nq;    // |q|
    double nqs;   // |q'| (rounded / reconstructed query; int8 dot: of the rescaled query)
    double dq;    // |q' - q| (int8 dot: |q^ - q~|)
    double cq2;   // int8: |c_q|^2
    double f;     // int8 dot: the rescale factor
    double nqt;   // int8 dot: |q~|
};

__device__ __forceinline__ double cert_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float cert_bf16_round(float v) {
    uint32_t u = __float_as_uint(v);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xFFFF0000u);
}

// q: the caller's query (LDS or global), dim elements.  Wave-collective.
__device__ __forceinline__ CertQ cert_query_consts(const CertArgs &a, const float *q, uint32_t dim, int metric, int lane) {
    CertQ c{};
    double s_q = 0.0, s_qs = 0.0, s_dq = 0.0, s_c = 0.0;
    double f = 1.0;
    if (a.mode == ZVDB_BF_I8 && metric != ZVDB_METRIC_L2SQ) {  // k_rescale_queries: q~ = q * (target / max |q_i|), in fp32
        float mx = 0.0f;
        for (uint32_t i = lane; i < dim; i += kWave) mx = fmaxf(mx, fabsf(q[i]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        const float ff = (mx > 0.0f && mx < __builtin_inff()) ? a.rescale_target / mx : 1.0f;
        f = (double)ff;
        double s_t = 0.0;
        for (uint32_t i = lane; i < dim; i += kWave) {
            const float qt = q[i] * ff + 0.0f;  // (the kernel's expression, `add` = 0 for int8)
            const float r = __builtin_rintf(qt * a.q_inv);
            const double code = (double)fminf(fmaxf(r, -128.0f), 127.0f);
            const double rec = code / (double)a.q_inv;
            s_q += (double)q[i] * (double)q[i];
            s_t += (double)qt * (double)qt;
            s_qs += rec * rec;
            s_dq += (rec - (double)qt) * (rec - (double)qt);
            s_c += code * code;
        }
        c.nqt = sqrt(cert_wave_sum(s_t));
    } else {
        for (uint32_t i = lane; i < dim; i += kWave) {
            const float x = q[i];
            double rec = (double)x;
            if (a.mode == ZVDB_BF_BF16) rec = (double)cert_bf16_round(x);
            else if (a.mode == ZVDB_BF_I8) {
                const float r = a.affine ? __builtin_rintf((x - a.q_lo) * a.q_inv) - 128.0f : __builtin_rintf(x * a.q_inv);
                const double code = (double)fminf(fmaxf(r, -128.0f), 127.0f);
                rec = a.affine ? (double)a.q_lo + (code + 128.0) / (double)a.q_inv : code / (double)a.q_inv;
                s_c += code * code;
            }
            s_q += (double)x * (double)x;
            s_qs += rec * rec;
            s_dq += (rec - (double)x) * (rec - (double)x);
        }
    }
    c.nq = sqrt(cert_wave_sum(s_q));
    c.nqs = sqrt(cert_wave_sum(s_qs));
    c.dq = sqrt(cert_wave_sum(s_dq)) * (1.0 + 1e-12);
    c.cq2 = cert_wave_sum(s_c);
    c.f = f;
    return c;
}

// The largest screen key (in the u32 key domain of the screen) a row can carry if its computed reference distance is <= d_k:
// a candidate whose key is ABOVE the returned value is provably outside the top-k; the query is certified when the returned
// value is BELOW the k'-th survivor's key.  0xFFFFFFFE: no bound (d_k not finite).
__device__ __forceinline__ uint32_t cert_safe_key(const CertArgs &a, const CertQ &c, float d_k, uint32_t dim, int metric) {
    if (!(d_k < __builtin_inff()) || d_k != d_k) return 0xFFFFFFFEu;
    const double rho_max = (double)a.stats[0] * (1.0 + 1e-6), u_max = (double)a.stats[1] * (1.0 + 1e-6);
    const double g = ((double)dim + 4.0) * 5.9604644775390625e-08 * 1.01;  // 2^-24
    const double u23 = 1.1920928955078125e-07, u24 = 5.9604644775390625e-08;
    double ks;  // key_safe in the screen's real key units
    if (a.mode == ZVDB_BF_I8) {
        const double s = 1.0 / (double)a.q_inv;
        if (metric == ZVDB_METRIC_L2SQ) {
            const double D = fmax((double)d_k, 0.0) / (1.0 - g) + 1e-30;
            const double e = c.dq + rho_max;
            const double r = sqrt(D) * (1.0 + 1e-12) + e;
            ks = (r * r / (s * s) * (1.0 + 1e-12) - c.cq2) * 0.5 + 1.0;  // (+1: the >> 1 of the stored norm, and slack)
        } else {
            const double U = u_max, rho = rho_max;
            const double kk = c.f * (double)d_k + c.f * g * c.nq * (U + rho) + c.dq * U + c.nqt * rho;
            ks = kk / (s * s) + 1.0 + fabs(kk / (s * s)) * 1e-12;
        }
        if (!(ks < 2147483000.0)) return 0xFFFFFFFEu;
        if (ks < -2147483000.0) ks = -2147483000.0;
        const int32_t ki = (int32_t)floor(ks);
        return (uint32_t)ki ^ 0x80000000u;
    }
    if (metric == ZVDB_METRIC_L2SQ) {
        const double D = fmax((double)d_k, 0.0) / (1.0 - g) + 1e-30;
        const double sD = sqrt(D) * (1.0 + 1e-12);
        const double unorm = c.nq + sD;                                                     // |u| under the supposition
        const double rho = a.mode == ZVDB_BF_BF16 ? fmin(rho_max, 0.00390625 * 1.0078125 * unorm + 1e-30) : 0.0;
        const double U = fmin(u_max, unorm + rho);
        const double r = sD + c.dq + rho;
        const double eta = ((double)dim + 64.0) * u23 * (0.5 * U * U + c.nqs * U) + ((double)dim + 8.0) * u24 * 0.5 * U * U;
        ks = (r * r - c.nqs * c.nqs) * 0.5;
        ks += eta + fabs(ks) * 1e-12;
    } else {
        const double U = u_max, rho = a.mode == ZVDB_BF_BF16 ? rho_max : 0.0;
        const double eta = ((double)dim + 64.0) * u23 * c.nqs * U;
        ks = (double)d_k + g * c.nq * (U + rho) + c.dq * U + c.nq * rho + eta;
        ks += fabs(ks) * 1e-12;
    }
    if (!(fabs(ks) < 3.0e38)) return 0xFFFFFFFEu;
    float kf = (float)ks;
    if ((double)kf < ks) kf = __uint_as_float(kf >= 0.0f ? __float_as_uint(kf) + 1u : (kf == 0.0f ? 1u : __float_as_uint(kf) - 1u));  // round up
    return f32_key(kf);
}

}  // namespace zvdb
