// Share arithmetic + reduction helpers shared by the polynomial-seam kernels.
//
// A polynomial entry is NC field elements: NC = 1 "plain" (public polynomial / plain prover) or
// NC = 2 a Rep3PrimeFieldShare {a, b} (mpc-types/src/protocols/rep3/arithmetic/types.rs:22-29).
// HBM layout is SoA at the share-component level: all `a` limbs, then all `b` limbs, each a
// dense array of 32-byte Montgomery elements -- `copy_share_a()` (dense_mlpoly.rs:103-110) is then
// a zero-copy view that the MSM consumes directly, and every load is a 16-byte/lane vector load.
#pragma once
#include "common.hpp"

template <int NC>
struct Sh {
    fe c[NC];
};

// TWO_INV = (r+1)/2 in Montgomery form (snarks-core/src/field.rs:5-7)
static FF_HD fe fr_two_inv() {
    fe r;
    const uint32_t v[8] = {0x1ffffffeu, 0x783c14d8u, 0x0c8d1eddu, 0xaf982f6fu,
                           0xfcfd4f45u, 0x8f5f7492u, 0x3d9cbfacu, 0x1f37631au};
    for (int i = 0; i < 8; i++) r.l[i] = v[i];
    return r;
}

template <int NC>
static __device__ __forceinline__ Sh<NC> sh_load(const fe* a, const fe* b, size_t i) {
    Sh<NC> s;
    s.c[0] = fe_load(a + i);
    if (NC == 2) s.c[NC - 1] = fe_load(b + i);
    return s;
}
template <int NC>
static __device__ __forceinline__ Sh<NC> sh_load_or_zero(const fe* a, const fe* b, size_t i, size_t len) {
    if (i < len) return sh_load<NC>(a, b, i);
    Sh<NC> s;
    for (int k = 0; k < NC; k++) s.c[k] = Fr::zero();
    return s;
}
template <int NC>
static __device__ __forceinline__ void sh_store(fe* a, fe* b, size_t i, const Sh<NC>& s) {
    fe_store(a + i, s.c[0]);
    if (NC == 2) fe_store(b + i, s.c[NC - 1]);
}
template <int NC>
static FF_HD Sh<NC> sh_add(const Sh<NC>& x, const Sh<NC>& y) {
    Sh<NC> r;
    for (int k = 0; k < NC; k++) r.c[k] = Fr::add(x.c[k], y.c[k]);
    return r;
}
template <int NC>
static FF_HD Sh<NC> sh_sub(const Sh<NC>& x, const Sh<NC>& y) {
    Sh<NC> r;
    for (int k = 0; k < NC; k++) r.c[k] = Fr::sub(x.c[k], y.c[k]);
    return r;
}
// Share x public (ops.rs:80-101)
template <int NC>
static FF_HD Sh<NC> sh_mul_public(const Sh<NC>& x, const fe& p) {
    Sh<NC> r;
    for (int k = 0; k < NC; k++) r.c[k] = Fr::mul(x.c[k], p);
    return r;
}
// lo + r * (hi - lo)   (`add_mul_public`, and the bind formula of dense_mlpoly.rs:316-374)
template <int NC>
static FF_HD Sh<NC> sh_lerp(const Sh<NC>& lo, const Sh<NC>& hi, const fe& r) {
    Sh<NC> o;
    for (int k = 0; k < NC; k++) o.c[k] = Fr::add(lo.c[k], Fr::mul(Fr::sub(hi.c[k], lo.c[k]), r));
    return o;
}
// Share x Share -> additive: a.a*b.a + a.a*b.b + a.b*b.a (ops.rs:71-78), evaluated as
// a.a*(b.a + b.b) + a.b*b.a: the same field element with 2 multiplications instead of 3.
template <int NC>
static FF_HD fe sh_local_mul(const Sh<NC>& x, const Sh<NC>& y) {
    if (NC == 1) return Fr::mul(x.c[0], y.c[0]);
    return Fr::add(Fr::mul(x.c[0], Fr::add(y.c[0], y.c[NC - 1])), Fr::mul(x.c[NC - 1], y.c[0]));
}
// into_additive *without* the TWO_INV factor: callers fold the constant in once per sum
template <int NC>
static FF_HD fe sh_ab_sum(const Sh<NC>& x) {
    if (NC == 1) return x.c[0];
    return Fr::add(x.c[0], x.c[NC - 1]);
}

// ---- lazy dot products: sum_i x_i * y_i with ONE Montgomery reduction at the end.
// A Montgomery product is 128 multiply-adds (64 for x * y, 64 for the reduction) + their carries; a dot product only needs
// the reduction once.  FrWide keeps the 15 columns of the 8 x 8 limb products as 96-bit accumulators (v_mad_u64_u32's
// carry-out folded into `hi`, ff_macc.inc): 64 mads + 64 addc per term, good for 2^29 terms; fr_wide_reduce turns the
// 544-bit total T = T0 + T1 R + T2 R^2 into T / R mod r = from_mont(T0) + (T1 mod r) + to_mont(T2): the same field element
// as the sum of the individually reduced products.  This is what makes evaluate_at_chi / linear_combination HBM-bound
// instead of multiplier-bound.
struct FrWide {
    uint64_t lo[15];
    uint32_t hi[15];
};
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ void fr_wide_zero(FrWide& w) {
#pragma unroll
    for (int k = 0; k < 15; k++) {
        w.lo[k] = 0;
        w.hi[k] = 0;
    }
}
static __device__ __forceinline__ void fr_wide_mac(FrWide& w, const fe& a, const fe& b) {
    const uint32_t* A = a.l;
    const uint32_t* B = b.l;
    MACC1_VV(w.lo[0], w.hi[0], A[0], B[0]);
    MACC2_VV(w.lo[1], w.hi[1], A[0], A[1], B[1], B[0]);
    MACC3_VV(w.lo[2], w.hi[2], A[0], A[1], A[2], B[2], B[1], B[0]);
    MACC4_VV(w.lo[3], w.hi[3], A[0], A[1], A[2], A[3], B[3], B[2], B[1], B[0]);
    MACC5_VV(w.lo[4], w.hi[4], A[0], A[1], A[2], A[3], A[4], B[4], B[3], B[2], B[1], B[0]);
    MACC6_VV(w.lo[5], w.hi[5], A[0], A[1], A[2], A[3], A[4], A[5], B[5], B[4], B[3], B[2], B[1], B[0]);
    MACC7_VV(w.lo[6], w.hi[6], A[0], A[1], A[2], A[3], A[4], A[5], A[6], B[6], B[5], B[4], B[3], B[2], B[1], B[0]);
    MACC8_VV(w.lo[7], w.hi[7], A[0], A[1], A[2], A[3], A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4], B[3], B[2], B[1], B[0]);
    MACC7_VV(w.lo[8], w.hi[8], A[1], A[2], A[3], A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4], B[3], B[2], B[1]);
    MACC6_VV(w.lo[9], w.hi[9], A[2], A[3], A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4], B[3], B[2]);
    MACC5_VV(w.lo[10], w.hi[10], A[3], A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4], B[3]);
    MACC4_VV(w.lo[11], w.hi[11], A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4]);
    MACC3_VV(w.lo[12], w.hi[12], A[5], A[6], A[7], B[7], B[6], B[5]);
    MACC2_VV(w.lo[13], w.hi[13], A[6], A[7], B[7], B[6]);
    MACC1_VV(w.lo[14], w.hi[14], A[7], B[7]);
}
static __device__ __forceinline__ fe fr_wide_reduce(const FrWide& w) {
    uint32_t t[18];
    uint64_t carry = 0;  // < 2^64: a column is < 2^96, so (column + carry) >> 32 < 2^64
#pragma unroll
    for (int k = 0; k < 15; k++) {
        uint64_t s = w.lo[k] + carry;
        uint32_t h = w.hi[k] + (s < carry ? 1u : 0u);
        t[k] = (uint32_t)s;
        carry = (s >> 32) | ((uint64_t)h << 32);
    }
    t[15] = (uint32_t)carry;
    t[16] = (uint32_t)(carry >> 32);
    t[17] = 0;
    fe t0, t1, t2 = Fr::zero();
#pragma unroll
    for (int i = 0; i < 8; i++) {
        t0.l[i] = t[i];
        t1.l[i] = t[8 + i];
    }
    t2.l[0] = t[16];
    t2.l[1] = t[17];
    fe r = Fr::add(Fr::from_mont(t0), Fr::to_mont(Fr::from_mont(t1)));
    return Fr::add(r, Fr::to_mont(t2));
}
#else  // host pass of the kernels' bodies: declarations only (the multiply-add chains are device assembly)
__device__ void fr_wide_zero(FrWide& w);
__device__ void fr_wide_mac(FrWide& w, const fe& a, const fe& b);
__device__ fe fr_wide_reduce(const FrWide& w);
#endif

// ---- reductions of field elements: wave shuffle tree, then LDS across the 4 waves of a block
static __device__ __forceinline__ fe fr_wave_sum(fe v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        fe o;
#pragma unroll
        for (int k = 0; k < 8; k++) o.l[k] = __shfl_down(v.l[k], off);
        v = Fr::add(v, o);
    }
    return v;
}
// all threads of a 256-thread block call this; thread 0 gets the block sum
static __device__ __forceinline__ fe fr_block_sum(fe v, fe* sh4) {
    v = fr_wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh4[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = sh4[0];
        for (int i = 1; i < (int)(blockDim.x >> 6); i++) v = Fr::add(v, sh4[i]);
    }
    return v;
}

// host <-> ABI conversions for Fr
static inline fe fe_from_u64x4(const uint64_t v[4]) {
    fe r;
    for (int i = 0; i < 4; i++) {
        r.l[2 * i] = (uint32_t)v[i];
        r.l[2 * i + 1] = (uint32_t)(v[i] >> 32);
    }
    return r;
}
static inline void fe_to_u64x4(const fe& a, uint64_t v[4]) {
    for (int i = 0; i < 4; i++) v[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
}

// unique polynomial through (i, evals[i]), i = 0..n-1 (UniPoly::from_evals), n in {3, 4}
static inline void unipoly_from_evals(const fe* ev, int n, fe* coeffs) {
    // Lagrange on the fixed nodes 0..n-1 with small-integer inverses
    fe inv2 = fr_two_inv();
    if (n == 3) {
        // c0 = e0; c2 = (e2 - 2 e1 + e0)/2; c1 = e1 - e0 - c2
        fe c2 = Fr::mul(Fr::add(Fr::sub(ev[2], Fr::dbl(ev[1])), ev[0]), inv2);
        coeffs[0] = ev[0];
        coeffs[2] = c2;
        coeffs[1] = Fr::sub(Fr::sub(ev[1], ev[0]), c2);
        return;
    }
    // n == 4: finite differences: d1 = e1-e0, d2 = e2-2e1+e0, d3 = e3-3e2+3e1-e0
    fe inv6;  // 6^{-1} in Montgomery form
    {
        const uint32_t v6[8] = {0x0aaaaaaau, 0x7d695c48u, 0xaed9b4f4u, 0x3a880fcfu,
                                0xa9a9c517u, 0xda7526dbu, 0x69deea8eu, 0x0a67cbb3u};
        for (int i = 0; i < 8; i++) inv6.l[i] = v6[i];
    }
    fe d1 = Fr::sub(ev[1], ev[0]);
    fe d2 = Fr::add(Fr::sub(ev[2], Fr::dbl(ev[1])), ev[0]);
    fe three_e2 = Fr::add(Fr::dbl(ev[2]), ev[2]), three_e1 = Fr::add(Fr::dbl(ev[1]), ev[1]);
    fe d3 = Fr::sub(Fr::add(Fr::sub(ev[3], three_e2), three_e1), ev[0]);
    // p(x) = e0 + d1 x + d2 x(x-1)/2 + d3 x(x-1)(x-2)/6
    fe a3 = Fr::mul(d3, inv6);
    fe h2 = Fr::mul(d2, inv2);
    // x(x-1)/2 -> h2 (x^2 - x); x(x-1)(x-2)/6 -> a3 (x^3 - 3x^2 + 2x)
    fe three_a3 = Fr::add(Fr::dbl(a3), a3);
    coeffs[0] = ev[0];
    coeffs[1] = Fr::add(Fr::sub(d1, h2), Fr::dbl(a3));
    coeffs[2] = Fr::sub(h2, three_a3);
    coeffs[3] = a3;
}

