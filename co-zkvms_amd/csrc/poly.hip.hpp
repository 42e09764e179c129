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

