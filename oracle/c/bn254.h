/*
 * ORACLE (test infrastructure, NOT product code): plain-C restatement of BN254 field / G1 arithmetic,
 * independent of the HIP engine (64-bit limbs + unsigned __int128 CIOS here; 32-bit limbs and a
 * product-scanning multiplier there).  "parity unpinned" by reference outputs (see oracle/pyref.py
 * header); pinned by the constants the reference holds (TWO_INV snarks-core/src/field.rs:5-7; R mod r
 * co-noir-spartan/noir-r1cs/noir_proof_scheme.json:7), EIP-196 vectors, and cross-checks against the
 * exact big-int oracle oracle/pyref.py (tests/test_oracle.py).
 * Layout = arkworks Fp256<MontBackend<_,4>>: 4 x u64 LE limbs, Montgomery form.
 */
#ifndef ORACLE_BN254_H
#define ORACLE_BN254_H
#include <stdint.h>
#include <string.h>

typedef struct { uint64_t l[4]; } fp;
typedef struct { const uint64_t mod[4]; uint64_t inv; const uint64_t one[4]; const uint64_t r2[4]; } fp_params;

static const fp_params FR = {
    {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
    0xc2e1f593efffffffull,
    {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full},
    {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull}};
static const fp_params FQ = {
    {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
    0x87d20782e4866389ull,
    {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full},
    {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full}};

typedef unsigned __int128 u128;

static inline int fp_is_zero(const fp* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fp_eq(const fp* a, const fp* b) {
    return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int fp_geq(const uint64_t a[4], const uint64_t m[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > m[i]) return 1;
        if (a[i] < m[i]) return 0;
    }
    return 1;
}
static inline void fp_sub_mod_raw(uint64_t a[4], const uint64_t m[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a[i] - m[i] - (uint64_t)br;
        a[i] = (uint64_t)t;
        br = (t >> 64) & 1;
    }
}
static inline void fp_add(const fp_params* P, fp* r, const fp* a, const fp* b) {
    u128 c = 0;
    uint64_t t[4];
    for (int i = 0; i < 4; i++) {
        c += (u128)a->l[i] + b->l[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    if (fp_geq(t, P->mod)) fp_sub_mod_raw(t, P->mod);
    memcpy(r->l, t, 32);
}
static inline void fp_sub(const fp_params* P, fp* r, const fp* a, const fp* b) {
    uint64_t t[4];
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)br;
        t[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t[i] + P->mod[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(r->l, t, 32);
}
static inline void fp_neg(const fp_params* P, fp* r, const fp* a) {
    fp z = {{0, 0, 0, 0}};
    if (fp_is_zero(a)) *r = *a; else fp_sub(P, r, &z, a);
}
static inline void fp_dbl(const fp_params* P, fp* r, const fp* a) { fp_add(P, r, a, a); }
/* Montgomery product (textbook CIOS, 5-limb accumulator) */
static inline void fp_mul(const fp_params* P, fp* r, const fp* a, const fp* b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * P->inv;
        c = (u128)m * P->mod[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * P->mod[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || fp_geq(t, P->mod)) fp_sub_mod_raw(t, P->mod);
    memcpy(r->l, t, 32);
}
static inline void fp_sqr(const fp_params* P, fp* r, const fp* a) { fp_mul(P, r, a, a); }
static inline void fp_one(const fp_params* P, fp* r) { memcpy(r->l, P->one, 32); }
static inline void fp_zero(fp* r) { memset(r->l, 0, 32); }
static inline void fp_to_mont(const fp_params* P, fp* r, const fp* canonical) {
    fp r2;
    memcpy(r2.l, P->r2, 32);
    fp_mul(P, r, canonical, &r2);
}
static inline void fp_from_mont(const fp_params* P, fp* r, const fp* a) {
    fp one = {{1, 0, 0, 0}};
    fp_mul(P, r, a, &one);
}
static inline void fp_from_u64(const fp_params* P, fp* r, uint64_t v) {
    fp c = {{v, 0, 0, 0}};
    fp_to_mont(P, r, &c);
}
static inline void fp_pow(const fp_params* P, fp* r, const fp* a, const uint64_t e[4]) {
    fp acc;
    fp_one(P, &acc);
    for (int i = 3; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            fp_sqr(P, &acc, &acc);
            if ((e[i] >> b) & 1) fp_mul(P, &acc, &acc, a);
        }
    *r = acc;
}
static inline void fp_inv(const fp_params* P, fp* r, const fp* a) {
    uint64_t e[4];
    memcpy(e, P->mod, 32);
    e[0] -= 2;
    fp_pow(P, r, a, e);
}

/* ---------------------------------------------------------------- G1: y^2 = x^3 + 3, Jacobian */
typedef struct { fp x, y; int inf; } g1a;
typedef struct { fp x, y, z; } g1j; /* z == 0 <=> identity */

static inline void g1j_identity(g1j* p) {
    fp_one(&FQ, &p->x);
    fp_one(&FQ, &p->y);
    fp_zero(&p->z);
}
static inline void g1j_double(g1j* r, const g1j* p) {
    if (fp_is_zero(&p->z)) { *r = *p; return; }
    fp A, B, C, D, E, F, t;
    fp_sqr(&FQ, &A, &p->x);
    fp_sqr(&FQ, &B, &p->y);
    fp_sqr(&FQ, &C, &B);
    fp_add(&FQ, &t, &p->x, &B);
    fp_sqr(&FQ, &t, &t);
    fp_sub(&FQ, &t, &t, &A);
    fp_sub(&FQ, &t, &t, &C);
    fp_dbl(&FQ, &D, &t);
    fp_dbl(&FQ, &E, &A);
    fp_add(&FQ, &E, &E, &A);
    fp_sqr(&FQ, &F, &E);
    fp z3;
    fp_mul(&FQ, &z3, &p->y, &p->z);
    fp_dbl(&FQ, &z3, &z3);
    fp x3;
    fp_dbl(&FQ, &t, &D);
    fp_sub(&FQ, &x3, &F, &t);
    fp y3;
    fp_sub(&FQ, &t, &D, &x3);
    fp_mul(&FQ, &y3, &E, &t);
    fp c8;
    fp_dbl(&FQ, &c8, &C);
    fp_dbl(&FQ, &c8, &c8);
    fp_dbl(&FQ, &c8, &c8);
    fp_sub(&FQ, &y3, &y3, &c8);
    r->x = x3; r->y = y3; r->z = z3;
}
static inline void g1j_add_affine(g1j* r, const g1j* p, const g1a* q) {
    if (q->inf) { *r = *p; return; }
    if (fp_is_zero(&p->z)) { r->x = q->x; r->y = q->y; fp_one(&FQ, &r->z); return; }
    fp z1z1, u2, s2, h, hh, i, j, rr, v, t;
    fp_sqr(&FQ, &z1z1, &p->z);
    fp_mul(&FQ, &u2, &q->x, &z1z1);
    fp_mul(&FQ, &s2, &q->y, &p->z);
    fp_mul(&FQ, &s2, &s2, &z1z1);
    if (fp_eq(&u2, &p->x)) {
        if (fp_eq(&s2, &p->y)) { g1j_double(r, p); return; }
        g1j_identity(r);
        return;
    }
    fp_sub(&FQ, &h, &u2, &p->x);
    fp_sqr(&FQ, &hh, &h);
    fp_dbl(&FQ, &i, &hh);
    fp_dbl(&FQ, &i, &i);
    fp_mul(&FQ, &j, &h, &i);
    fp_sub(&FQ, &rr, &s2, &p->y);
    fp_dbl(&FQ, &rr, &rr);
    fp_mul(&FQ, &v, &p->x, &i);
    fp x3, y3, z3;
    fp_sqr(&FQ, &x3, &rr);
    fp_sub(&FQ, &x3, &x3, &j);
    fp_dbl(&FQ, &t, &v);
    fp_sub(&FQ, &x3, &x3, &t);
    fp_sub(&FQ, &t, &v, &x3);
    fp_mul(&FQ, &y3, &rr, &t);
    fp_mul(&FQ, &t, &p->y, &j);
    fp_dbl(&FQ, &t, &t);
    fp_sub(&FQ, &y3, &y3, &t);
    fp_add(&FQ, &z3, &p->z, &h);
    fp_sqr(&FQ, &z3, &z3);
    fp_sub(&FQ, &z3, &z3, &z1z1);
    fp_sub(&FQ, &z3, &z3, &hh);
    r->x = x3; r->y = y3; r->z = z3;
}
static inline void g1j_add(g1j* r, const g1j* a, const g1j* b) {
    if (fp_is_zero(&a->z)) { *r = *b; return; }
    if (fp_is_zero(&b->z)) { *r = *a; return; }
    fp z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t;
    fp_sqr(&FQ, &z1z1, &a->z);
    fp_sqr(&FQ, &z2z2, &b->z);
    fp_mul(&FQ, &u1, &a->x, &z2z2);
    fp_mul(&FQ, &u2, &b->x, &z1z1);
    fp_mul(&FQ, &s1, &a->y, &b->z);
    fp_mul(&FQ, &s1, &s1, &z2z2);
    fp_mul(&FQ, &s2, &b->y, &a->z);
    fp_mul(&FQ, &s2, &s2, &z1z1);
    if (fp_eq(&u1, &u2)) {
        if (fp_eq(&s1, &s2)) { g1j_double(r, a); return; }
        g1j_identity(r);
        return;
    }
    fp_sub(&FQ, &h, &u2, &u1);
    fp_dbl(&FQ, &i, &h);
    fp_sqr(&FQ, &i, &i);
    fp_mul(&FQ, &j, &h, &i);
    fp_sub(&FQ, &rr, &s2, &s1);
    fp_dbl(&FQ, &rr, &rr);
    fp_mul(&FQ, &v, &u1, &i);
    fp x3, y3, z3;
    fp_sqr(&FQ, &x3, &rr);
    fp_sub(&FQ, &x3, &x3, &j);
    fp_dbl(&FQ, &t, &v);
    fp_sub(&FQ, &x3, &x3, &t);
    fp_sub(&FQ, &t, &v, &x3);
    fp_mul(&FQ, &y3, &rr, &t);
    fp_mul(&FQ, &t, &s1, &j);
    fp_dbl(&FQ, &t, &t);
    fp_sub(&FQ, &y3, &y3, &t);
    fp_add(&FQ, &z3, &a->z, &b->z);
    fp_sqr(&FQ, &z3, &z3);
    fp_sub(&FQ, &z3, &z3, &z1z1);
    fp_sub(&FQ, &z3, &z3, &z2z2);
    fp_mul(&FQ, &z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3;
}
static inline void g1j_to_affine(g1a* r, const g1j* p) {
    if (fp_is_zero(&p->z)) { fp_zero(&r->x); fp_zero(&r->y); r->inf = 1; return; }
    fp zi, zi2, zi3;
    fp_inv(&FQ, &zi, &p->z);
    fp_sqr(&FQ, &zi2, &zi);
    fp_mul(&FQ, &zi3, &zi2, &zi);
    fp_mul(&FQ, &r->x, &p->x, &zi2);
    fp_mul(&FQ, &r->y, &p->y, &zi3);
    r->inf = 0;
}
static inline void g1a_neg(g1a* r, const g1a* p) {
    *r = *p;
    if (!p->inf) fp_neg(&FQ, &r->y, &p->y);
}
/* s (canonical, 4 limbs) * p */
static inline void g1_scalar_mul(g1j* r, const g1a* p, const uint64_t s[4]) {
    g1j acc;
    g1j_identity(&acc);
    for (int i = 3; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            g1j_double(&acc, &acc);
            if ((s[i] >> b) & 1) g1j_add_affine(&acc, &acc, p);
        }
    *r = acc;
}

/* ---------------------------------------------------------------- round 3: the CPU baseline's fast path (bench.py cpu_baseline).
 * Fq product with the modulus as compile-time constants, fully unrolled "no-carry" CIOS (the optimisation arkworks applies to
 * moduli whose top bit is clear, ark-ff montgomery_backend): per outer step one multiply-add row for a * b[i] and one for m * p,
 * carries in 128-bit temporaries.  Same function as fp_mul(&FQ, ..); tests compare the two. */
static inline void fq_mul_fast(fp* r, const fp* a, const fp* b) {
    const uint64_t N0 = 0x3c208c16d87cfd47ull, N1 = 0x97816a916871ca8dull, N2 = 0xb85045b68181585dull, N3 = 0x30644e72e131a029ull;
    const uint64_t INV = 0x87d20782e4866389ull;
    const uint64_t a0 = a->l[0], a1 = a->l[1], a2 = a->l[2], a3 = a->l[3];
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#define FQ_ROW(bi)                                                     \
    do {                                                               \
        u128 c = (u128)a0 * (bi) + t0;                                 \
        uint64_t m = (uint64_t)c * INV;                                \
        u128 d = (u128)m * N0 + (uint64_t)c;                           \
        c = (u128)a1 * (bi) + t1 + (uint64_t)(c >> 64);                \
        d = (u128)m * N1 + (uint64_t)c + (uint64_t)(d >> 64);          \
        t0 = (uint64_t)d;                                              \
        c = (u128)a2 * (bi) + t2 + (uint64_t)(c >> 64);                \
        d = (u128)m * N2 + (uint64_t)c + (uint64_t)(d >> 64);          \
        t1 = (uint64_t)d;                                              \
        c = (u128)a3 * (bi) + t3 + (uint64_t)(c >> 64);                \
        d = (u128)m * N3 + (uint64_t)c + (uint64_t)(d >> 64);          \
        t2 = (uint64_t)d;                                              \
        t3 = (uint64_t)(c >> 64) + (uint64_t)(d >> 64);                \
    } while (0)
    FQ_ROW(b->l[0]);
    FQ_ROW(b->l[1]);
    FQ_ROW(b->l[2]);
    FQ_ROW(b->l[3]);
#undef FQ_ROW
    uint64_t t[4] = {t0, t1, t2, t3};
    if (fp_geq(t, FQ.mod)) fp_sub_mod_raw(t, FQ.mod);
    memcpy(r->l, t, 32);
}
static inline void fq_sqr_fast(fp* r, const fp* a) { fq_mul_fast(r, a, a); }

/* XYZZ coordinates (x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; EFD "xyzz"): the mixed addition is 8M + 2S instead of the Jacobian
 * 7M + 4S, the general one 12M + 2S; zz == 0 <=> identity */
typedef struct { fp x, y, zz, zzz; } g1x;
static inline void g1x_identity(g1x* p) { fp_zero(&p->x); fp_zero(&p->y); fp_zero(&p->zz); fp_zero(&p->zzz); }
static inline void g1x_double_affine(g1x* r, const g1a* q) { /* dbl-2008-s-1 with ZZ1 = ZZZ1 = 1 */
    fp u, v, w, s, m, t;
    fp_dbl(&FQ, &u, &q->y);
    fq_sqr_fast(&v, &u);
    fq_mul_fast(&w, &u, &v);
    fq_mul_fast(&s, &q->x, &v);
    fq_sqr_fast(&m, &q->x);
    fp_dbl(&FQ, &t, &m);
    fp_add(&FQ, &m, &m, &t);
    fq_sqr_fast(&r->x, &m);
    fp_dbl(&FQ, &t, &s);
    fp_sub(&FQ, &r->x, &r->x, &t);
    fp_sub(&FQ, &t, &s, &r->x);
    fq_mul_fast(&r->y, &m, &t);
    fq_mul_fast(&t, &w, &q->y);
    fp_sub(&FQ, &r->y, &r->y, &t);
    r->zz = v;
    r->zzz = w;
}
static inline void g1x_double(g1x* r, const g1x* p) { /* dbl-2008-s-1 */
    if (fp_is_zero(&p->zz)) { *r = *p; return; }
    fp u, v, w, s, m, t, x3, y3;
    fp_dbl(&FQ, &u, &p->y);
    fq_sqr_fast(&v, &u);
    fq_mul_fast(&w, &u, &v);
    fq_mul_fast(&s, &p->x, &v);
    fq_sqr_fast(&m, &p->x);
    fp_dbl(&FQ, &t, &m);
    fp_add(&FQ, &m, &m, &t);
    fq_sqr_fast(&x3, &m);
    fp_dbl(&FQ, &t, &s);
    fp_sub(&FQ, &x3, &x3, &t);
    fp_sub(&FQ, &t, &s, &x3);
    fq_mul_fast(&y3, &m, &t);
    fq_mul_fast(&t, &w, &p->y);
    fp_sub(&FQ, &y3, &y3, &t);
    fq_mul_fast(&r->zz, &v, &p->zz);
    fq_mul_fast(&r->zzz, &w, &p->zzz);
    r->x = x3;
    r->y = y3;
}
/* r = p + (qx, qy) (madd-2008-s); neg != 0 adds the negated point */
static inline void g1x_add_affine(g1x* r, const g1x* p, const g1a* q, int neg) {
    if (q->inf) { *r = *p; return; }
    fp qy = q->y;
    if (neg) fp_neg(&FQ, &qy, &q->y);
    if (fp_is_zero(&p->zz)) { r->x = q->x; r->y = qy; fp_one(&FQ, &r->zz); fp_one(&FQ, &r->zzz); return; }
    fp u2, s2, pp_, rr, PP, PPP, Q, t, x3, y3;
    fq_mul_fast(&u2, &q->x, &p->zz);
    fq_mul_fast(&s2, &qy, &p->zzz);
    fp_sub(&FQ, &pp_, &u2, &p->x);
    fp_sub(&FQ, &rr, &s2, &p->y);
    if (fp_is_zero(&pp_)) {
        if (fp_is_zero(&rr)) { g1a qq = *q; qq.y = qy; g1x_double_affine(r, &qq); return; }
        g1x_identity(r);
        return;
    }
    fq_sqr_fast(&PP, &pp_);
    fq_mul_fast(&PPP, &pp_, &PP);
    fq_mul_fast(&Q, &p->x, &PP);
    fq_sqr_fast(&x3, &rr);
    fp_sub(&FQ, &x3, &x3, &PPP);
    fp_dbl(&FQ, &t, &Q);
    fp_sub(&FQ, &x3, &x3, &t);
    fp_sub(&FQ, &t, &Q, &x3);
    fq_mul_fast(&y3, &rr, &t);
    fq_mul_fast(&t, &p->y, &PPP);
    fp_sub(&FQ, &y3, &y3, &t);
    fq_mul_fast(&r->zz, &p->zz, &PP);
    fq_mul_fast(&r->zzz, &p->zzz, &PPP);
    r->x = x3;
    r->y = y3;
}
static inline void g1x_add(g1x* r, const g1x* a, const g1x* b) { /* add-2008-s */
    if (fp_is_zero(&a->zz)) { *r = *b; return; }
    if (fp_is_zero(&b->zz)) { *r = *a; return; }
    fp u1, u2, s1, s2, pp_, rr, PP, PPP, Q, t, x3, y3;
    fq_mul_fast(&u1, &a->x, &b->zz);
    fq_mul_fast(&u2, &b->x, &a->zz);
    fq_mul_fast(&s1, &a->y, &b->zzz);
    fq_mul_fast(&s2, &b->y, &a->zzz);
    fp_sub(&FQ, &pp_, &u2, &u1);
    fp_sub(&FQ, &rr, &s2, &s1);
    if (fp_is_zero(&pp_)) {
        if (fp_is_zero(&rr)) { g1x_double(r, a); return; }
        g1x_identity(r);
        return;
    }
    fq_sqr_fast(&PP, &pp_);
    fq_mul_fast(&PPP, &pp_, &PP);
    fq_mul_fast(&Q, &u1, &PP);
    fq_sqr_fast(&x3, &rr);
    fp_sub(&FQ, &x3, &x3, &PPP);
    fp_dbl(&FQ, &t, &Q);
    fp_sub(&FQ, &x3, &x3, &t);
    fp_sub(&FQ, &t, &Q, &x3);
    fq_mul_fast(&y3, &rr, &t);
    fq_mul_fast(&t, &s1, &PPP);
    fp_sub(&FQ, &y3, &y3, &t);
    fq_mul_fast(&t, &a->zz, &b->zz);
    fq_mul_fast(&r->zz, &t, &PP);
    fq_mul_fast(&t, &a->zzz, &b->zzz);
    fq_mul_fast(&r->zzz, &t, &PPP);
    r->x = x3;
    r->y = y3;
}
static inline void g1x_to_affine(g1a* r, const g1x* p) {
    if (fp_is_zero(&p->zz)) { fp_zero(&r->x); fp_zero(&r->y); r->inf = 1; return; }
    fp zi, zzi, t;
    fp_inv(&FQ, &zi, &p->zzz);     /* 1 / ZZZ */
    fq_mul_fast(&t, &zi, &p->zz);  /* ZZ / ZZZ = 1 / Z */
    fq_sqr_fast(&zzi, &t);         /* 1 / ZZ */
    fq_mul_fast(&r->x, &p->x, &zzi);
    fq_mul_fast(&r->y, &p->y, &zi);
    r->inf = 0;
}
#endif
