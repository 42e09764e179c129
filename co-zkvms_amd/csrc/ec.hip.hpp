// BN254 G1 (y^2 = x^3 + 3 over Fq) group arithmetic for the MSM / PST13 kernels.
//
// Bucket accumulators use extended-Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// mixed addition of an affine SRS point costs 8M + 2S with no inversion, full addition 12M + 2S.
// XYZZ coordinates live in the lazy range [0, 2p) on the device (ff.hip.hpp l-operations: products skip their
// final conditional subtraction); affine points are always canonical, and to_affine canonicalises.  The
// P == Q / P == -Q special cases are detected by a zero test that accepts 0 and p; every formula handles
// identity, doubling and cancellation, because bucket contents arrive in arbitrary (atomic-scatter) order
// and SRS points may repeat.
// Result convention at the C ABI: affine, Montgomery, LE limbs -- what `into_affine()` yields at
// co-jolt/src/poly/commitment/pst13.rs:294,328.
#pragma once
#include "ff.hip.hpp"

struct alignas(16) g1_affine {  // 64 B; infinity is encoded as (0, 0), which is not on the curve
    fe x, y;
};
struct alignas(16) g1_xyzz {  // 128 B; identity <=> zz == 0
    fe x, y, zz, zzz;
};

struct G1 {
    static FF_HD bool is_inf(const g1_affine& p) { return Fq::is_zero(p.x) && Fq::is_zero(p.y); }
    static FF_HD bool is_identity(const g1_xyzz& p) { return Fq::is_zero(p.zz); }
    static FF_HD g1_xyzz identity() {
        g1_xyzz r;
        r.x = Fq::zero();
        r.y = Fq::zero();
        r.zz = Fq::zero();
        r.zzz = Fq::zero();
        return r;
    }
    static FF_HD g1_xyzz from_affine(const g1_affine& p) {
        if (is_inf(p)) return identity();
        g1_xyzz r;
        r.x = p.x;
        r.y = p.y;
        r.zz = Fq::one();
        r.zzz = Fq::one();
        return r;
    }
    static FF_HD g1_affine neg(const g1_affine& p) {
        g1_affine r;
        r.x = p.x;
        r.y = Fq::neg(p.y);
        return r;
    }
    static FF_HD g1_xyzz neg(const g1_xyzz& p) {
        g1_xyzz r = p;
        r.y = Fq::lneg(p.y);
        return r;
    }
    // 2*(x, y) for an affine point (EFD mdbl-2008-s-1, a = 0)
    static FF_HD g1_xyzz dbl_affine(const g1_affine& p) {
        if (is_inf(p) || Fq::is_zero(p.y)) return identity();
        fe U = Fq::dbl(p.y);
        fe V = Fq::sqr(U);
        fe W = Fq::mul(U, V);
        fe S = Fq::mul(p.x, V);
        fe XX = Fq::sqr(p.x);
        fe M = Fq::add(Fq::dbl(XX), XX);
        g1_xyzz r;
        r.x = Fq::sub(Fq::sqr(M), Fq::dbl(S));
        r.y = Fq::sub(Fq::mul(M, Fq::sub(S, r.x)), Fq::mul(W, p.y));
        r.zz = V;
        r.zzz = W;
        return r;
    }
    // 2*P (EFD dbl-2008-s-1, a = 0), lazy range in and out
    static FF_HD g1_xyzz dbl(const g1_xyzz& p) {
        if (is_identity(p) || Fq::lis_zero(p.y)) return identity();
        fe U = Fq::ldbl(p.y);
        fe V = Fq::lmul(U, U);
        fe W, S;
        Fq::lmul2(U, V, p.x, V, W, S);
        fe XX = Fq::lmul(p.x, p.x);
        fe M = Fq::ladd(Fq::ldbl(XX), XX);
        g1_xyzz r;
        r.x = Fq::lsub(Fq::lmul(M, M), Fq::ldbl(S));
        r.y = Fq::lmul_sub2(M, Fq::lsub(S, r.x), W, p.y);
        Fq::lmul2(V, p.zz, W, p.zzz, r.zz, r.zzz);
        return r;
    }
    // P + Q, Q affine and canonical (EFD madd-2008-s): ten products as four independent pairs (lmul2) plus Y3's
    // two under one reduction; no product pays its final conditional subtraction (lazy range)
    static FF_HD g1_xyzz add_mixed(const g1_xyzz& p, const g1_affine& q) {
        if (is_inf(q)) return p;
        if (is_identity(p)) return from_affine(q);
        fe U2, S2;
        Fq::lmul2(q.x, p.zz, q.y, p.zzz, U2, S2);
        fe Pd = Fq::lsub(U2, p.x);
        fe Rd = Fq::lsub(S2, p.y);
        if (Fq::lis_zero(Pd)) {
            if (Fq::lis_zero(Rd)) return dbl_affine(q);
            return identity();
        }
        fe PP, RR;
        Fq::lmul2(Pd, Pd, Rd, Rd, PP, RR);
        fe PPP, Q;
        Fq::lmul2(Pd, PP, p.x, PP, PPP, Q);
        g1_xyzz r;
        r.x = Fq::lsub(Fq::lsub(RR, PPP), Fq::ldbl(Q));
        r.y = Fq::lmul_sub2(Rd, Fq::lsub(Q, r.x), p.y, PPP);
        Fq::lmul2(p.zz, PP, p.zzz, PPP, r.zz, r.zzz);
        return r;
    }
    // P + Q (EFD add-2008-s): 14 products, lazy range in and out
    static FF_HD g1_xyzz add(const g1_xyzz& p, const g1_xyzz& q) {
        if (is_identity(q)) return p;
        if (is_identity(p)) return q;
        fe U1, U2, S1, S2;
        Fq::lmul2(p.x, q.zz, q.x, p.zz, U1, U2);
        Fq::lmul2(p.y, q.zzz, q.y, p.zzz, S1, S2);
        fe Pd = Fq::lsub(U2, U1);
        fe Rd = Fq::lsub(S2, S1);
        if (Fq::lis_zero(Pd)) {
            if (Fq::lis_zero(Rd)) return dbl(p);
            return identity();
        }
        fe PP, RR;
        Fq::lmul2(Pd, Pd, Rd, Rd, PP, RR);
        fe PPP, Q;
        Fq::lmul2(Pd, PP, U1, PP, PPP, Q);
        g1_xyzz r;
        r.x = Fq::lsub(Fq::lsub(RR, PPP), Fq::ldbl(Q));
        r.y = Fq::lmul_sub2(Rd, Fq::lsub(Q, r.x), S1, PPP);
        fe zz12, zzz12;
        Fq::lmul2(p.zz, q.zz, p.zzz, q.zzz, zz12, zzz12);
        Fq::lmul2(zz12, PP, zzz12, PPP, r.zz, r.zzz);
        return r;
    }
    // affine = (X/ZZ, Y/ZZZ); one field inversion: 1/ZZZ, then 1/ZZ = ZZ^2 / ZZZ^2 ... computed as
    // inv(zz*zzz) to share a single inversion.
    static FF_HD g1_affine to_affine(const g1_xyzz& p) {
        g1_affine r;
        if (is_identity(p)) {
            r.x = Fq::zero();
            r.y = Fq::zero();
            return r;
        }
        fe zz = Fq::lcanon(p.zz), zzz = Fq::lcanon(p.zzz);
        fe t = Fq::inv(Fq::mul(zz, zzz));
        fe izz = Fq::mul(t, zzz);
        fe izzz = Fq::mul(t, zz);
        r.x = Fq::mul(Fq::lcanon(p.x), izz);
        r.y = Fq::mul(Fq::lcanon(p.y), izzz);
        return r;
    }
    static FF_HD bool on_curve(const g1_affine& p) {
        if (is_inf(p)) return true;
        fe three = Fq::from_u64(3);
        fe lhs = Fq::sqr(p.y);
        fe rhs = Fq::add(Fq::mul(Fq::sqr(p.x), p.x), three);
        return Fq::eq(lhs, rhs);
    }
};

static __device__ __forceinline__ g1_affine affine_load(const g1_affine* p) {
    g1_affine r;
    r.x = fe_load(&p->x);
    r.y = fe_load(&p->y);
    return r;
}
static __device__ __forceinline__ void affine_store(g1_affine* p, const g1_affine& v) {
    fe_store(&p->x, v.x);
    fe_store(&p->y, v.y);
}
static __device__ __forceinline__ g1_xyzz xyzz_load(const g1_xyzz* p) {
    g1_xyzz r;
    r.x = fe_load(&p->x);
    r.y = fe_load(&p->y);
    r.zz = fe_load(&p->zz);
    r.zzz = fe_load(&p->zzz);
    return r;
}
static __device__ __forceinline__ void xyzz_store(g1_xyzz* p, const g1_xyzz& v) {
    fe_store(&p->x, v.x);
    fe_store(&p->y, v.y);
    fe_store(&p->zz, v.zz);
    fe_store(&p->zzz, v.zzz);
}
