/* oracle/cpu_ref.c -- TEST INFRASTRUCTURE ONLY: the CPU oracle and the "port" CPU baseline.
 *
 * A plain-C restatement of the algorithms the reference delegates to ark-poly / ark-ec 0.4
 * (in-order radix-2 NTT; signed-digit Pippenger MSM with the arkworks window rule), for
 * BLS12-381 and BN254.  Those crates are crates.io dependencies (plonk/Cargo.toml:13-41,
 * primitives/Cargo.toml:14-56) and are NOT present under /root/reference; there is no Rust
 * toolchain in this image, so the reference itself is unbuildable here.
 * PARITY UNPINNED: the reference holds no golden vectors for this path; this file is pinned
 * against oracle/pyref.py (definition-level big-int arithmetic) by tests/test_oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * curve_id: 0 = BLS12-381, 1 = BN254.  All field elements little-endian u64 limbs.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

#include "constants.h"

/* ---- field instantiations ------------------------------------------------------------ */
#define NL 4
#define FP(n) blsfr_##n
#define FP_MOD BLS_FR_MOD
#define FP_INV BLS_FR_INV
#define FP_R BLS_FR_R
#define FP_R2 BLS_FR_R2
#include "fp_impl.inc"
#undef FP
#undef FP_MOD
#undef FP_INV
#undef FP_R
#undef FP_R2

#define FP(n) bnfr_##n
#define FP_MOD BN_FR_MOD
#define FP_INV BN_FR_INV
#define FP_R BN_FR_R
#define FP_R2 BN_FR_R2
#include "fp_impl.inc"
#undef FP
#undef FP_MOD
#undef FP_INV
#undef FP_R
#undef FP_R2

#define FP(n) bnfq_##n
#define FP_MOD BN_FQ_MOD
#define FP_INV BN_FQ_INV
#define FP_R BN_FQ_R
#define FP_R2 BN_FQ_R2
#include "fp_impl.inc"
#undef FP
#undef FP_MOD
#undef FP_INV
#undef FP_R
#undef FP_R2
#undef NL

#define NL 6
#define FP(n) blsfq_##n
#define FP_MOD BLS_FQ_MOD
#define FP_INV BLS_FQ_INV
#define FP_R BLS_FQ_R
#define FP_R2 BLS_FQ_R2
#include "fp_impl.inc"
#undef FP
#undef FP_MOD
#undef FP_INV
#undef FP_R
#undef FP_R2
#undef NL

/* ---- NTT instantiations --------------------------------------------------------------- */
#define NTT(n) blsntt_##n
#define FR(n) blsfr_##n
#define NTT_ROOT BLS_FR_ROOT
#define NTT_TWO_ADICITY BLS_TWO_ADICITY
#include "ntt_impl.inc"
#undef NTT
#undef FR
#undef NTT_ROOT
#undef NTT_TWO_ADICITY

#define NTT(n) bnntt_##n
#define FR(n) bnfr_##n
#define NTT_ROOT BN_FR_ROOT
#define NTT_TWO_ADICITY BN_TWO_ADICITY
#include "ntt_impl.inc"
#undef NTT
#undef FR
#undef NTT_ROOT
#undef NTT_TWO_ADICITY

/* ---- TurboPlonk quotient instantiations -------------------------------------------------------- */
#define PLK(n) blsplk_##n
#define FR(n) blsfr_##n
#define NTT(n) blsntt_##n
#define PLK_GEN BLS_FR_GENERATOR
#include "plonk_impl.inc"
#undef PLK
#undef FR
#undef NTT
#undef PLK_GEN

#define PLK(n) bnplk_##n
#define FR(n) bnfr_##n
#define NTT(n) bnntt_##n
#define PLK_GEN BN_FR_GENERATOR
#include "plonk_impl.inc"
#undef PLK
#undef FR
#undef NTT
#undef PLK_GEN

/* ---- G1 instantiations ---------------------------------------------------------------- */
#define G1(n) blsg1_##n
#define FQ(n) blsfq_##n
#include "g1_impl.inc"
#undef G1
#undef FQ

#define G1(n) bng1_##n
#define FQ(n) bnfq_##n
#include "g1_impl.inc"
#undef G1
#undef FQ

/* ---- exported C entry points (ctypes) -------------------------------------------------- */
#define EXPORT __attribute__((visibility("default")))

EXPORT int orc_fq_limbs(int curve) { return curve == 0 ? 6 : curve == 1 ? 4 : -1; }

/* in-place (i)NTT, natural order; coset_mont NULL => plain domain */
EXPORT int orc_ntt(int curve, u64 *data, int log_n, int inverse, const u64 *coset_mont, int threads) {
    if (threads < 1) threads = 1;
    if (curve == 0) return blsntt_run(data, log_n, inverse, coset_mont, threads);
    if (curve == 1) return bnntt_run(data, log_n, inverse, coset_mont, threads);
    return -2;
}

/* Fr Montgomery <-> canonical, n elements, in/out may alias */
EXPORT int orc_fr_convert(int curve, const u64 *in, u64 *out, size_t n, int to_mont) {
    for (size_t i = 0; i < n; i++) {
        if (curve == 0) {
            blsfr_t a; memcpy(&a, in + 4 * i, 32);
            if (to_mont) blsfr_to_mont(&a, &a); else blsfr_from_mont(&a, &a);
            memcpy(out + 4 * i, &a, 32);
        } else if (curve == 1) {
            bnfr_t a; memcpy(&a, in + 4 * i, 32);
            if (to_mont) bnfr_to_mont(&a, &a); else bnfr_from_mont(&a, &a);
            memcpy(out + 4 * i, &a, 32);
        } else return -2;
    }
    return 0;
}

/* Fq Montgomery <-> canonical */
EXPORT int orc_fq_convert(int curve, const u64 *in, u64 *out, size_t n, int to_mont) {
    for (size_t i = 0; i < n; i++) {
        if (curve == 0) {
            blsfq_t a; memcpy(&a, in + 6 * i, 48);
            if (to_mont) blsfq_to_mont(&a, &a); else blsfq_from_mont(&a, &a);
            memcpy(out + 6 * i, &a, 48);
        } else if (curve == 1) {
            bnfq_t a; memcpy(&a, in + 4 * i, 32);
            if (to_mont) bnfq_to_mont(&a, &a); else bnfq_from_mont(&a, &a);
            memcpy(out + 4 * i, &a, 32);
        } else return -2;
    }
    return 0;
}

/* Fr element-wise product (Montgomery), used by tests for linearity / pointwise checks */
EXPORT int orc_fr_mul(int curve, const u64 *a, const u64 *b, u64 *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (curve == 0) blsfr_mul((blsfr_t *)(out + 4 * i), (const blsfr_t *)(a + 4 * i), (const blsfr_t *)(b + 4 * i));
        else if (curve == 1) bnfr_mul((bnfr_t *)(out + 4 * i), (const bnfr_t *)(a + 4 * i), (const bnfr_t *)(b + 4 * i));
        else return -2;
    }
    return 0;
}

/* Horner evaluation of a Montgomery coefficient vector at a Montgomery point */
EXPORT int orc_poly_eval(int curve, const u64 *coeffs, size_t n, const u64 *x_mont, u64 *out_mont) {
    if (curve == 0) {
        blsfr_t acc, x; memset(&acc, 0, sizeof acc); memcpy(&x, x_mont, 32);
        for (size_t j = n; j-- > 0;) { blsfr_mul(&acc, &acc, &x); blsfr_add(&acc, &acc, (const blsfr_t *)(coeffs + 4 * j)); }
        memcpy(out_mont, &acc, 32);
    } else if (curve == 1) {
        bnfr_t acc, x; memset(&acc, 0, sizeof acc); memcpy(&x, x_mont, 32);
        for (size_t j = n; j-- > 0;) { bnfr_mul(&acc, &acc, &x); bnfr_add(&acc, &acc, (const bnfr_t *)(coeffs + 4 * j)); }
        memcpy(out_mont, &acc, 32);
    } else return -2;
    return 0;
}

/* quotient of p(X) / (X - z) by synthetic division, len-1 coefficients (ark-poly's `/`, used at prover.rs:504-506) */
EXPORT int orc_poly_div_linear(int curve, const u64 *coeffs, size_t len, const u64 *z_mont, u64 *out) {
    if (len < 2) return 0;
    if (curve == 0) {
        blsfr_t z, q; memcpy(&z, z_mont, 32); memcpy(&q, coeffs + 4 * (len - 1), 32);
        for (size_t k = len - 1; k-- > 0;) { memcpy(out + 4 * k, &q, 32); blsfr_mul(&q, &q, &z); blsfr_add(&q, &q, (const blsfr_t *)(coeffs + 4 * k)); }
    } else if (curve == 1) {
        bnfr_t z, q; memcpy(&z, z_mont, 32); memcpy(&q, coeffs + 4 * (len - 1), 32);
        for (size_t k = len - 1; k-- > 0;) { memcpy(out + 4 * k, &q, 32); bnfr_mul(&q, &q, &z); bnfr_add(&q, &q, (const bnfr_t *)(coeffs + 4 * k)); }
    } else return -2;
    return 0;
}

/* out[j] = sum_k scalars[k] * polys[k][j]; polys are n_terms rows of `stride` elements, row k valid for lens[k] */
EXPORT int orc_poly_lincomb(int curve, int n_terms, const u64 *polys, size_t stride, const size_t *lens, const u64 *scalars, u64 *out, size_t out_len) {
    for (size_t j = 0; j < out_len; j++) {
        if (curve == 0) {
            blsfr_t acc, t; memset(&acc, 0, sizeof acc);
            for (int k = 0; k < n_terms; k++) if (j < lens[k]) { blsfr_mul(&t, (const blsfr_t *)(scalars + 4 * k), (const blsfr_t *)(polys + ((size_t)k * stride + j) * 4)); blsfr_add(&acc, &acc, &t); }
            memcpy(out + 4 * j, &acc, 32);
        } else if (curve == 1) {
            bnfr_t acc, t; memset(&acc, 0, sizeof acc);
            for (int k = 0; k < n_terms; k++) if (j < lens[k]) { bnfr_mul(&t, (const bnfr_t *)(scalars + 4 * k), (const bnfr_t *)(polys + ((size_t)k * stride + j) * 4)); bnfr_add(&acc, &acc, &t); }
            memcpy(out + 4 * j, &acc, 32);
        } else return -2;
    }
    return 0;
}

/* omega_N^k * offset as a Montgomery element (domain.element(k)) */
EXPORT int orc_domain_element(int curve, int log_n, u64 k, const u64 *coset_mont, u64 *out_mont) {
    u64 e[1] = {k};
    if (curve == 0) {
        blsfr_t w; blsntt_root(&w, log_n); blsfr_pow(&w, &w, e, 1);
        if (coset_mont) blsfr_mul(&w, &w, (const blsfr_t *)coset_mont);
        memcpy(out_mont, &w, 32);
    } else if (curve == 1) {
        bnfr_t w; bnntt_root(&w, log_n); bnfr_pow(&w, &w, e, 1);
        if (coset_mont) bnfr_mul(&w, &w, (const bnfr_t *)coset_mont);
        memcpy(out_mont, &w, 32);
    } else return -2;
    return 0;
}

/* TurboPlonk quotient polynomial (one instance); see plonk_impl.inc for the layout */
EXPORT int orc_plonk_quotient(int curve, int log_n, int num_wire_types, const u64 *polys, size_t poly_len, const u64 *k_mont,
                              const u64 *alpha, const u64 *beta, const u64 *gamma, u64 *out, int threads) {
    if (threads < 1) threads = 1;
    if (curve == 0) return blsplk_quotient(log_n, num_wire_types, polys, poly_len, k_mont, alpha, beta, gamma, out, threads);
    if (curve == 1) return bnplk_quotient(log_n, num_wire_types, polys, poly_len, k_mont, alpha, beta, gamma, out, threads);
    return -2;
}

/* permutation grand-product polynomial (constraint_system.rs:1197-1223) */
EXPORT int orc_plonk_perm_product(int curve, int log_n, int num_wire_types, const u64 *wires, const u64 *sigma_vals, const u64 *k_mont,
                                  const u64 *beta, const u64 *gamma, u64 *out, int threads) {
    if (threads < 1) threads = 1;
    if (curve == 0) return blsplk_perm_product(log_n, num_wire_types, wires, sigma_vals, k_mont, beta, gamma, out, threads);
    if (curve == 1) return bnplk_perm_product(log_n, num_wire_types, wires, sigma_vals, k_mont, beta, gamma, out, threads);
    return -2;
}

/* MSM: bases packed x||y Montgomery ((0,0) = infinity), scalars 4 limbs each (canonical, or
 * Montgomery when scalars_are_mont), out = Jacobian X,Y,Z Montgomery (Z = 0 => infinity).
 * window_bits 0 => arkworks rule. */
EXPORT int orc_msm(int curve, const u64 *bases_xy, const u64 *scalars, size_t n, int scalars_are_mont,
                   u64 *out_xyz, int threads, int window_bits) {
    if (threads < 1) threads = 1;
    u64 *canon = NULL;
    if (scalars_are_mont) {
        canon = (u64 *)malloc(32 * (n ? n : 1));
        if (orc_fr_convert(curve, scalars, canon, n, 0)) { free(canon); return -2; }
        scalars = canon;
    }
    int rc = 0;
    if (curve == 0) {
        blsg1_jac r;
        blsg1_msm(&r, (const blsg1_aff *)bases_xy, scalars, n, BLS_FR_BITS, threads, window_bits);
        memcpy(out_xyz, &r, sizeof r);
    } else if (curve == 1) {
        bng1_jac r;
        bng1_msm(&r, (const bng1_aff *)bases_xy, scalars, n, BN_FR_BITS, threads, window_bits);
        memcpy(out_xyz, &r, sizeof r);
    } else rc = -2;
    free(canon);
    return rc;
}

/* Jacobian (Montgomery) -> affine x||y (Montgomery); infinity -> (0,0) */
EXPORT int orc_jac_to_affine(int curve, const u64 *xyz, u64 *out_xy, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if (curve == 0) blsg1_to_aff((blsg1_aff *)(out_xy + 12 * i), (const blsg1_jac *)(xyz + 18 * i));
        else if (curve == 1) bng1_to_aff((bng1_aff *)(out_xy + 8 * i), (const bng1_jac *)(xyz + 12 * i));
        else return -2;
    }
    return 0;
}

/* out = k*G (affine, Montgomery) for one canonical scalar k */
EXPORT int orc_g1_mul_gen(int curve, const u64 k[4], u64 *out_xy) {
    if (curve == 0) {
        blsg1_aff g; blsg1_jac r;
        memcpy(&g.x, BLS_GEN_X, 48); memcpy(&g.y, BLS_GEN_Y, 48);
        blsg1_mul_bigint(&r, &g, k); blsg1_to_aff((blsg1_aff *)out_xy, &r);
    } else if (curve == 1) {
        bng1_aff g; bng1_jac r;
        memcpy(&g.x, BN_GEN_X, 32); memcpy(&g.y, BN_GEN_Y, 32);
        bng1_mul_bigint(&r, &g, k); bng1_to_aff((bng1_aff *)out_xy, &r);
    } else return -2;
    return 0;
}

/* out[i] = k*P for an arbitrary affine base P */
EXPORT int orc_g1_mul(int curve, const u64 *p_xy, const u64 k[4], u64 *out_xy) {
    if (curve == 0) { blsg1_jac r; blsg1_mul_bigint(&r, (const blsg1_aff *)p_xy, k); blsg1_to_aff((blsg1_aff *)out_xy, &r); }
    else if (curve == 1) { bng1_jac r; bng1_mul_bigint(&r, (const bng1_aff *)p_xy, k); bng1_to_aff((bng1_aff *)out_xy, &r); }
    else return -2;
    return 0;
}

/* test bases: P_0 = s*G, P_{i+1} = P_i + t*G, batch-normalised (distinct, on curve, in subgroup)
 * -- the generator recipe of SURVEY.md 8(d) C2. */
EXPORT int orc_g1_arith_bases(int curve, const u64 s[4], const u64 t[4], size_t n, u64 *out_xy) {
    if (curve == 0) {
        blsg1_aff g, d; blsg1_jac cur, dj;
        memcpy(&g.x, BLS_GEN_X, 48); memcpy(&g.y, BLS_GEN_Y, 48);
        blsg1_mul_bigint(&cur, &g, s); blsg1_mul_bigint(&dj, &g, t); blsg1_to_aff(&d, &dj);
        blsg1_jac *tmp = (blsg1_jac *)malloc(sizeof(blsg1_jac) * (n ? n : 1));
        for (size_t i = 0; i < n; i++) { tmp[i] = cur; blsg1_madd(&cur, &cur, &d); }
        blsg1_batch_to_aff((blsg1_aff *)out_xy, tmp, n);
        free(tmp);
    } else if (curve == 1) {
        bng1_aff g, d; bng1_jac cur, dj;
        memcpy(&g.x, BN_GEN_X, 32); memcpy(&g.y, BN_GEN_Y, 32);
        bng1_mul_bigint(&cur, &g, s); bng1_mul_bigint(&dj, &g, t); bng1_to_aff(&d, &dj);
        bng1_jac *tmp = (bng1_jac *)malloc(sizeof(bng1_jac) * (n ? n : 1));
        for (size_t i = 0; i < n; i++) { tmp[i] = cur; bng1_madd(&cur, &cur, &d); }
        bng1_batch_to_aff((bng1_aff *)out_xy, tmp, n);
        free(tmp);
    } else return -2;
    return 0;
}

/* testing SRS: out[i] = beta^i * G, i < n (gen_srs_for_testing, srs.rs:118-153, g = generator).
 * beta canonical.  Uses a 256-entry table of 2^k * G, then per-power mixed additions. */
EXPORT int orc_srs_powers(int curve, const u64 beta[4], size_t n, u64 *out_xy, int threads) {
    if (threads < 1) threads = 1;
    if (curve == 0) {
        blsg1_aff tab[256]; blsg1_jac cur; blsg1_aff g;
        memcpy(&g.x, BLS_GEN_X, 48); memcpy(&g.y, BLS_GEN_Y, 48);
        blsg1_jac_from_aff(&cur, &g);
        blsg1_jac tj[256];
        for (int k = 0; k < 256; k++) { tj[k] = cur; blsg1_dbl(&cur, &cur); }
        blsg1_batch_to_aff(tab, tj, 256);
        blsfr_t b, p; memcpy(&b, beta, 32); blsfr_to_mont(&b, &b); blsfr_set_one(&p);
        u64 *pw = (u64 *)malloc(32 * (n ? n : 1));
        for (size_t i = 0; i < n; i++) { blsfr_t c; blsfr_from_mont(&c, &p); memcpy(pw + 4 * i, &c, 32); blsfr_mul(&p, &p, &b); }
        blsg1_jac *tmp = (blsg1_jac *)malloc(sizeof(blsg1_jac) * (n ? n : 1));
#pragma omp parallel for num_threads(threads) schedule(static)
        for (size_t i = 0; i < n; i++) {
            blsg1_jac acc; blsg1_jac_set_inf(&acc);
            for (int k = 0; k < 256; k++) if ((pw[4 * i + (k >> 6)] >> (k & 63)) & 1) blsg1_madd(&acc, &acc, &tab[k]);
            tmp[i] = acc;
        }
        blsg1_batch_to_aff((blsg1_aff *)out_xy, tmp, n);
        free(tmp); free(pw);
    } else if (curve == 1) {
        bng1_aff tab[256]; bng1_jac cur; bng1_aff g;
        memcpy(&g.x, BN_GEN_X, 32); memcpy(&g.y, BN_GEN_Y, 32);
        bng1_jac_from_aff(&cur, &g);
        bng1_jac tj[256];
        for (int k = 0; k < 256; k++) { tj[k] = cur; bng1_dbl(&cur, &cur); }
        bng1_batch_to_aff(tab, tj, 256);
        bnfr_t b, p; memcpy(&b, beta, 32); bnfr_to_mont(&b, &b); bnfr_set_one(&p);
        u64 *pw = (u64 *)malloc(32 * (n ? n : 1));
        for (size_t i = 0; i < n; i++) { bnfr_t c; bnfr_from_mont(&c, &p); memcpy(pw + 4 * i, &c, 32); bnfr_mul(&p, &p, &b); }
        bng1_jac *tmp = (bng1_jac *)malloc(sizeof(bng1_jac) * (n ? n : 1));
#pragma omp parallel for num_threads(threads) schedule(static)
        for (size_t i = 0; i < n; i++) {
            bng1_jac acc; bng1_jac_set_inf(&acc);
            for (int k = 0; k < 256; k++) if ((pw[4 * i + (k >> 6)] >> (k & 63)) & 1) bng1_madd(&acc, &acc, &tab[k]);
            tmp[i] = acc;
        }
        bng1_batch_to_aff((bng1_aff *)out_xy, tmp, n);
        free(tmp); free(pw);
    } else return -2;
    return 0;
}

/* UltraPlonk (Plookup) restatements; see plonk_impl.inc */
EXPORT int orc_plookup_merge(int curve, size_t n, const u64 *wires, const u64 *tabs, const u64 *q_lookup, const u64 *tau_m, u64 *out_table, u64 *out_lookup) {
    if (curve == 0) blsplk_lookup_merge(n, wires, tabs, q_lookup, tau_m, out_table, out_lookup);
    else if (curve == 1) bnplk_lookup_merge(n, wires, tabs, q_lookup, tau_m, out_table, out_lookup);
    else return -2;
    return 0;
}
EXPORT long orc_plookup_sorted(int curve, size_t n, const u64 *table, const u64 *lookup, u64 *out) {
    if (curve == 0) return blsplk_lookup_sorted(n, table, lookup, out);
    if (curve == 1) return bnplk_lookup_sorted(n, table, lookup, out);
    return -2;
}
EXPORT int orc_plookup_product(int curve, int log_n, const u64 *table, const u64 *lookup, const u64 *sorted, const u64 *beta_m, const u64 *gamma_m, u64 *out, int threads) {
    if (curve == 0) return blsplk_lookup_product(log_n, table, lookup, sorted, beta_m, gamma_m, out, threads);
    if (curve == 1) return bnplk_lookup_product(log_n, table, lookup, sorted, beta_m, gamma_m, out, threads);
    return -2;
}
EXPORT int orc_plonk_quotient_ultra(int curve, int log_n, const u64 *polys, size_t poly_len, const u64 *k_mont, const u64 *tau_m, const u64 *alpha_m,
                                    const u64 *beta_m, const u64 *gamma_m, u64 *out, int threads) {
    if (curve == 0) return blsplk_quotient_ultra(log_n, polys, poly_len, k_mont, tau_m, alpha_m, beta_m, gamma_m, out, threads);
    if (curve == 1) return bnplk_quotient_ultra(log_n, polys, poly_len, k_mont, tau_m, alpha_m, beta_m, gamma_m, out, threads);
    return -2;
}

/* on-curve check of packed affine points; returns number of points NOT on the curve */
EXPORT long orc_g1_count_off_curve(int curve, const u64 *xy, size_t n) {
    long bad = 0;
    for (size_t i = 0; i < n; i++) {
        if (curve == 0) {
            const blsg1_aff *p = (const blsg1_aff *)(xy + 12 * i);
            if (blsg1_aff_is_inf(p)) continue;
            blsfq_t l, r, b; blsfq_sqr(&l, &p->y); blsfq_sqr(&r, &p->x); blsfq_mul(&r, &r, &p->x);
            blsfq_set_u64(&b, 4); blsfq_add(&r, &r, &b);
            bad += !blsfq_eq(&l, &r);
        } else {
            const bng1_aff *p = (const bng1_aff *)(xy + 8 * i);
            if (bng1_aff_is_inf(p)) continue;
            bnfq_t l, r, b; bnfq_sqr(&l, &p->y); bnfq_sqr(&r, &p->x); bnfq_mul(&r, &r, &p->x);
            bnfq_set_u64(&b, 3); bnfq_add(&r, &r, &b);
            bad += !bnfq_eq(&l, &r);
        }
    }
    return bad;
}
