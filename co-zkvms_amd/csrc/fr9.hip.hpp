// Fr in 9 x 29-bit unsaturated limbs for the GKR layer kernels (round 3) -- the multiplier of fq9.hip.hpp (fused single-accumulator
// v_mad_u64_u32 chains, no carry instructions, radix R' = 2^261) instantiated for the SCALAR field.
//
// Why: the ISA of k_layer_bind_cubic showed ~6800 vector instructions per output chunk of which only 1280 are the multiply-adds of its 10
// products: the saturated 8 x 32 product-scanning multiplier costs ~550-600 instructions per product in that context (128 mads, ~140
// add-with-carry, ~180 register moves of the 96-bit column accumulators), and the kernels ran at 0.5-0.6 of the vector issue rate with
// neither HBM nor the multiplier busy.  The 9 x 29 product is 162 mads + ~45 shifts / masks, and additions need no modular reduction:
//   * VALUES are lazy: a product of inputs < A r and < B r is < r (1 + A B / 169) (r / R' < 1/169); sums simply grow (7 spare bits);
//     subtraction is a + C - b with C a multiple of r whose limbs dominate b's (fr9_consts.inc);
//   * LIMBS are re-normalised (f9_norm: 25 instructions) where the next product needs it.  Column sums stay below 2^64 when
//     9 max(a_i) max(b_i) + 9 * 2^58 < 2^64: every call site keeps its SECOND operand normalised (< 2^29) and its first below 2^30.6
//     (below 2^30 for the two-product form).
// Memory stays in the ordinary R = 2^256 Montgomery form (byte-identical to arkworks): values are re-limbed on load, and a stored value
// is made canonical again (normalise, re-limb, two conditional subtractions).  A product of two R-form values under the R' reduction
// carries lambda = R / R' = 2^-5: the bind multiplies by the challenge pre-scaled by 2^5 (exact R-form result), the cubic terms carry
// lambda^2 (lambda^3 with the E2 factor), removed by ONE product with a constant per lane at the end.  Results are the same field
// elements as the saturated kernels' (k_layer_bind_cubic / k_layer_cubic remain for small layers; the parity tests run both).
#pragma once
#include "fq9.hip.hpp"
#include "fr9_consts.inc"

#if defined(__HIP_DEVICE_COMPILE__)
// a * b / R' mod r; b normalised, a's limbs < 2^30.6; output normalised, value < r (1 + A B / 169)
static __device__ __forceinline__ f9 fr9_mul(const f9& a, const f9& b) {
    const uint32_t(&F9_P)[9] = FR9_P;  // the generated bodies name the modulus F9_P / F9_INV: shadow Fq's with Fr's
    constexpr uint32_t F9_INV = FR9_INV;
    uint64_t acc = 0;
    uint32_t m[9];
    f9 r;
    F9_MUL_BODY
    return r;
}
// (a * b + c * d) / R' mod r under one reduction; b, d normalised, a's and c's limbs < 2^30
static __device__ __forceinline__ f9 fr9_mul_add2(const f9& a, const f9& b, const f9& c, const f9& d) {
    const uint32_t(&F9_P)[9] = FR9_P;
    constexpr uint32_t F9_INV = FR9_INV;
    uint64_t acc = 0;
    uint32_t m[9];
    f9 r;
    F9_MUL_ADD2_BODY
    return r;
}
#else
__device__ f9 fr9_mul(const f9& a, const f9& b);
__device__ f9 fr9_mul_add2(const f9& a, const f9& b, const f9& c, const f9& d);
#endif
static __device__ __forceinline__ f9 fr9_add(const f9& a, const f9& b) {
    f9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
static __device__ __forceinline__ f9 fr9_zero() {
    f9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
}
// value < 2.1 r (any limbs < 2^31) -> the canonical 8 x 32 element
static __device__ __forceinline__ fe fr9_to_canonical(const f9& a) {
    return Fr::reduce_once(Fr::reduce_once(f9_to_fe(f9_norm(a))));
}

// NC components of a share in the lazy 9 x 29 form
template <int NC>
struct Sh9 {
    f9 c[NC];
};
template <int NC>
static __device__ __forceinline__ Sh9<NC> sh9_load_or_zero(const fe* a, const fe* b, size_t i, size_t len) {
    Sh9<NC> s;
    if (i < len) {
        s.c[0] = f9_from_fe(fe_load(a + i));
        if (NC == 2) s.c[NC - 1] = f9_from_fe(fe_load(b + i));
    } else {
        for (int k = 0; k < NC; k++) s.c[k] = fr9_zero();
    }
    return s;
}
// lo + r (hi - lo) with r5 = the challenge times 2^5 (R-form, normalised): value < 2.02 r, limbs < 2^30
template <int NC>
static __device__ __forceinline__ Sh9<NC> sh9_lerp(const Sh9<NC>& lo, const Sh9<NC>& hi, const f9& r5) {
    Sh9<NC> o;
    for (int k = 0; k < NC; k++) o.c[k] = fr9_add(lo.c[k], fr9_mul(f9_sub(hi.c[k], FR9_C2, lo.c[k]), r5));
    return o;
}
template <int NC>
static __device__ __forceinline__ void sh9_store(fe* a, fe* b, size_t i, const Sh9<NC>& s) {
    fe_store(a + i, fr9_to_canonical(s.c[0]));
    if (NC == 2) fe_store(b + i, fr9_to_canonical(s.c[NC - 1]));
}
// hi - lo for bound values (< 2.1 r, limbs < 2^30), normalised: value < 5.1 r
template <int NC>
static __device__ __forceinline__ Sh9<NC> sh9_diff(const Sh9<NC>& hi, const Sh9<NC>& lo) {
    Sh9<NC> o;
    for (int k = 0; k < NC; k++) o.c[k] = f9_norm(f9_sub(hi.c[k], FR9_C3, lo.c[k]));
    return o;
}
template <int NC>
static __device__ __forceinline__ Sh9<NC> sh9_add_norm(const Sh9<NC>& x, const Sh9<NC>& y) {
    Sh9<NC> o;
    for (int k = 0; k < NC; k++) o.c[k] = f9_norm(fr9_add(x.c[k], y.c[k]));
    return o;
}
// Share x Share -> additive (mpc-types/src/protocols/rep3/arithmetic/ops.rs:71-78) as x.a (y.a + y.b) + x.b y.a under ONE reduction;
// x's limbs < 2^30, y's any < 2^31 (normalised here); output normalised, carries lambda
template <int NC>
static __device__ __forceinline__ f9 sh9_local_mul(const Sh9<NC>& x, const Sh9<NC>& y) {
    if (NC == 1) return fr9_mul(x.c[0], f9_norm(y.c[0]));
    return fr9_mul_add2(x.c[0], f9_norm(fr9_add(y.c[0], y.c[NC - 1])), x.c[NC - 1], f9_norm(y.c[0]));
}
