// Fq in 9 x 29-bit unsaturated limbs for the MSM gather kernel (k_msm_accum0_f9, msm.hip).
//
// Why: on gfx950 v_mad_u64_u32 issues at full rate, so a 254-bit Montgomery product is bound by its instruction
// COUNT.  With saturated 8 x 32-bit limbs every multiply-add needs a second instruction to catch the carry out of
// the 64-bit accumulator (136 mads + 136 addc + moves ~ 330 instructions, 129 G products/s measured).  With 29-bit
// limbs the 18 products of a column (9 of a*b, 9 of m*p, each < 2^58) fit a 64-bit accumulator with room to
// spare, so the carry instructions disappear: 162 mads + ~45 shifts/masks.  Written as plain C++ the compiler
// splits each column into two accumulator chains and merges them with a 64-bit add (170 G products/s); the
// fused single-accumulator chains of fq9_mac.inc avoid that (177 G products/s, gather kernel -3.8 %).
//
// Radix R' = 2^261 leaves 7 spare bits above p (p / R' < 1/169), which the mixed addition uses twice:
//   * VALUES are never reduced: a product of inputs < A p and < B p is < p (1 + A B / 169).  The invariant kept
//     across additions is X < 6p, Y < 2p, ZZ, ZZZ < 1.05p (bounds worked out at madd9 below); subtraction is
//     a + C - b with C a multiple of p whose limbs dominate b's (fq9_consts.inc), no borrow, no comparison.
//   * LIMBS are re-normalised (carry propagation, 25 instructions) only where the next product needs it.
// The SRS table stays in the ordinary Montgomery form (radix R = 2^256).  Multiplying R-form values with the
// R' multiplier scales every product by lambda = R / R' = 2^-5; the XYZZ formulas are homogeneous, and with a fresh
// accumulator embedded as (x lambda, y lambda^2, 1, lambda) the powers of lambda stay consistent through every
// addition: with X, Y, ZZ, ZZZ carrying lambda^x, lambda^y, lambda^z2, lambda^z3, every step keeps x - z2 = 1,
// y - z3 = 1 and 3 z2 - 2 z3 = -2.  On the way out X / lambda, Y / lambda^2, ZZ, ZZZ / lambda (three products with
// constants) is an ordinary R-form XYZZ point again: the right affine point AND ZZ^3 = ZZZ^2, which the XYZZ
// formulas of the later fold levels rely on.  Exceptional inputs (P == +-Q, i.e. PP == 0 mod p) are not
// handled here: the kernel hands the segment to the saturated-limb path.
#pragma once
#include "ec.hip.hpp"
#include "fq9_consts.inc"
#include "fq9_mac.inc"
#include "fq9_mul.inc"

struct f9 {
    uint32_t l[9];
};
static constexpr uint32_t F9_MASK = (1u << 29) - 1u;

// 8 x 32 (value < 2^256) -> 9 x 29 normalised
static __device__ __forceinline__ f9 f9_from_fe(const fe& a) {
    f9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, s = bit & 31;
        uint64_t v = a.l[w];
        if (w + 1 < 8) v |= (uint64_t)a.l[w + 1] << 32;
        r.l[i] = (uint32_t)(v >> s) & F9_MASK;
    }
    return r;
}
// 9 x 29 normalised, value < 2^256 -> 8 x 32
static __device__ __forceinline__ fe f9_to_fe(const f9& a) {
    fe r;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, i = bit / 29, s = bit - 29 * i;  // word j starts s bits into limb i
        uint64_t v = (uint64_t)a.l[i] >> s;
        int have = 29 - s;
        if (i + 1 < 9) v |= (uint64_t)a.l[i + 1] << have;
        have += 29;
        if (have < 32 && i + 2 < 9) v |= (uint64_t)a.l[i + 2] << have;
        r.l[j] = (uint32_t)v;
    }
    return r;
}
static __device__ __forceinline__ f9 f9_const(const uint32_t (&c)[9]) {
    f9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = c[i];
    return r;
}

// a * b / R' mod p, output limbs normalised.  Column sums: 9 a*b + 9 m*p products; callers keep
// 9 * max(a_i) * max(b_i) + 9 * 2^58 below 2^64 (limb bounds stated at each call site).
static __device__ __forceinline__ f9 f9_mul(const f9& a, const f9& b) {
    // fused single-accumulator chains (fq9_mac.inc / fq9_mul.inc): 162 mads + 9 m computations + 17 shifts + 17 masks
    uint64_t acc = 0;
    uint32_t m[9];
    f9 r;
    F9_MUL_BODY
    return r;
}
// two independent products with their accumulator chains interleaved instruction by instruction: a single chain is one
// long run of dependent v_mad_u64_u32, which needs other waves of the SIMD to cover the multiplier's latency
static __device__ __forceinline__ void f9_mul_x2(const f9& a, const f9& b, const f9& a2, const f9& b2, f9& r, f9& r2) {
    uint64_t acc = 0, acc2 = 0;
    uint32_t m[9], m2[9];
    F9_MUL_X2_BODY
}
static __device__ __forceinline__ void f9_sqr_x2(const f9& a, const f9& a2, f9& r, f9& r2) {
    uint64_t acc = 0, acc2 = 0;
    uint32_t m[9], m2[9], dd[9], dd2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        dd[i] = a.l[i] << 1;
        dd2[i] = a2.l[i] << 1;
    }
    F9_SQR_X2_BODY
}
// a * a / R' mod p for a normalised a: the 36 off-diagonal products are taken once against the doubled operand
// (limbs < 2^30, products < 2^59, at most 4 of them + one square + 9 m*p terms per column: < 2^62)
static __device__ __forceinline__ f9 f9_sqr(const f9& a) {
    uint64_t acc = 0;
    uint32_t m[9], dd[9];
    f9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) dd[i] = a.l[i] << 1;
    F9_SQR_BODY
    return r;
}
// (a * b + c * d) / R' mod p under one reduction
static __device__ __forceinline__ f9 f9_mul_add2(const f9& a, const f9& b, const f9& c, const f9& d) {
    uint64_t acc = 0;
    uint32_t m[9];
    f9 r;
    F9_MUL_ADD2_BODY
    return r;
}
// carry propagation: limbs < 2^32 - 8 in, limbs 0..7 < 2^29 out (the top limb keeps the rest)
static __device__ __forceinline__ f9 f9_norm(const f9& a) {
    f9 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t t = a.l[i] + c;
        r.l[i] = t & F9_MASK;
        c = t >> 29;
    }
    r.l[8] = a.l[8] + c;
    return r;
}
// a + C - b limb-wise; C = k p spread so that C_i >= b_i (fq9_consts.inc)
static __device__ __forceinline__ f9 f9_sub(const f9& a, const uint32_t (&C)[9], const f9& b) {
    f9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + C[i] - b.l[i];
    return r;
}
// value == 0 mod p for a product output (normalised, < 2p): all limbs 0, or equal to p
static __device__ __forceinline__ bool f9_is_zero_mod_p(const f9& a) {
    // almost always decided by the lowest limb (it is 0 or p's lowest limb for one value in 2^28)
    if (a.l[0] != 0u && a.l[0] != F9_P[0]) return false;
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        z |= a.l[i];
        e |= a.l[i] ^ F9_P[i];
    }
    return z == 0 || e == 0;
}

struct xyzz9 {
    f9 x, y, zz, zzz;
};

// fresh accumulator from an affine R-form point (not infinity): (x lambda, y lambda^2, 1, lambda)
static __device__ __forceinline__ xyzz9 xyzz9_from_affine(const f9& qx, const f9& qy) {
    xyzz9 r;
    r.x = f9_mul(qx, f9_const(F9_ONE));
    r.y = f9_mul(qy, f9_const(F9_LAM));
    r.zz = f9_const(F9_ONE);
    r.zzz = f9_const(F9_LAM);
    return r;
}

// acc += (qx, qy), EFD madd-2008-s.  Returns false (acc untouched) when PP == 0 mod p, i.e. P == +-Q.
// Value bounds in units of p (product of < A and < B gives < 1 + A B / 169), limb bounds in brackets:
//   in : X < 6, Y < 2, ZZ, ZZZ < 1.05, all normalised [2^29]; qx < 1 canonical; qy < 1 canonical, or 2p - y for a
//        negative digit (value < 2, limbs [2^30]: S2 < 1.02, column sums of qy * ZZZ <= 9 * 2^59 + 9 * 2^58)
//   U2, S2 < 1.01                               Pd = U2 + 7p - X < 8.01, Rd = S2 + 3p - Y < 4.01   -> normalised
//   PP < 1.38, RR < 1.10, PPP < 1.07, Q < 1.05   X3 = RR + 4p - PPP - 2Q < 5.1 [2^29 + 2^31]       -> normalised
//   T = Q + 7p - X3 < 8.05 [3 * 2^29];           NY = 3p - Y [2^30]
//   Y3 = Rd T + NY PPP < 1 + (4.01 * 8.05 + 3 * 1.07) / 169 = 1.21;  column sums <= 9 * 2^58 * (3 + 2 + 1) < 2^63.8
//   ZZ3 = ZZ PP, ZZZ3 = ZZZ PPP < 1.01
static __device__ __forceinline__ bool madd9(xyzz9& acc, const f9& qx, const f9& qy) {
    f9 U2 = f9_mul(qx, acc.zz);
    f9 S2 = f9_mul(qy, acc.zzz);
    f9 Pd = f9_norm(f9_sub(U2, F9_C7, acc.x));
    f9 Rd = f9_norm(f9_sub(S2, F9_C3, acc.y));
    f9 PP = f9_sqr(Pd);
    if (f9_is_zero_mod_p(PP)) return false;
    f9 RR = f9_sqr(Rd);
    f9 PPP = f9_mul(Pd, PP);
    f9 Q = f9_mul(acc.x, PP);
    f9 X3;
#pragma unroll
    for (int i = 0; i < 9; i++) X3.l[i] = RR.l[i] + F9_C4X3[i] - PPP.l[i] - 2u * Q.l[i];
    X3 = f9_norm(X3);
    f9 T = f9_sub(Q, F9_C7, X3);
    f9 NY;
#pragma unroll
    for (int i = 0; i < 9; i++) NY.l[i] = F9_C3[i] - acc.y.l[i];
    acc.y = f9_mul_add2(Rd, T, NY, PPP);
    acc.x = X3;
    acc.zz = f9_mul(acc.zz, PP);
    acc.zzz = f9_mul(acc.zzz, PPP);
    return true;
}

// back to an ordinary R-form XYZZ point in the lazy range [0, 2p) of ec.hip.hpp
static __device__ __forceinline__ g1_xyzz xyzz9_to_xyzz(const xyzz9& a) {
    g1_xyzz r;
    r.x = f9_to_fe(f9_mul(a.x, f9_const(F9_OUT)));
    r.y = f9_to_fe(f9_mul(a.y, f9_const(F9_OUT2)));
    r.zz = f9_to_fe(a.zz);
    r.zzz = f9_to_fe(f9_mul(a.zzz, f9_const(F9_OUT)));
    return r;
}
