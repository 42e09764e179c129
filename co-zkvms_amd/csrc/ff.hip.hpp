// BN254 prime-field arithmetic for gfx950 (CDNA4) -- device + host.
//
// Representation: 8 x u32 little-endian limbs in Montgomery form (R = 2^256).  Byte-identical to
// arkworks' Fp256<MontBackend<_,4>> (4 x u64 LE limbs), so reference buffers upload unchanged
// (SURVEY.md 8 conventions; constants pinned by snarks-core/src/field.rs:5-7 and
// co-noir-spartan/noir-r1cs/noir_proof_scheme.json:7).
//
// CDNA4 has no 64x64 multiplier: the work-horse is v_mad_u64_u32 (32x32+64 -> 64).  The
// multiplier below is the "no-carry" merged CIOS (valid because the top limb of both moduli is
// 0x30644e72 < 2^31): 2 mads per limb pair, fully unrolled so all 8 limbs stay in VGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FF_HD __host__ __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
#include "ff_macc.inc"
#include "ff_mul2.inc"
#endif

struct alignas(16) fe {
    uint32_t l[8];
};

struct FrParams {
    static constexpr uint32_t MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                        0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV = 0xefffffffu;  // -r^{-1} mod 2^32
    static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                        0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                       0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

struct FqParams {
    static constexpr uint32_t MOD[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                        0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV = 0xe4866389u;  // -p^{-1} mod 2^32
    static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                        0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                       0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};

template <class Pm>
struct Field {
    static FF_HD fe zero() {
        fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = 0;
        return r;
    }
    static FF_HD fe one() {
        fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = Pm::ONE[i];
        return r;
    }
    static FF_HD fe r2() {
        fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = Pm::R2[i];
        return r;
    }
    static FF_HD bool is_zero(const fe& a) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= a.l[i];
        return o == 0;
    }
    static FF_HD bool eq(const fe& a, const fe& b) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= a.l[i] ^ b.l[i];
        return o == 0;
    }
    // a >= MOD ?
    static FF_HD bool geq_mod(const fe& a) {
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)a.l[i] - Pm::MOD[i] - borrow;
            borrow = (d >> 63) & 1;
        }
        return borrow == 0;
    }
    // r = a - MOD if a >= MOD else a   (a < 2*MOD)
    static FF_HD fe reduce_once(const fe& a) {
        fe d;
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)a.l[i] - Pm::MOD[i] - borrow;
            d.l[i] = (uint32_t)t;
            borrow = (t >> 63) & 1;
        }
        fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = borrow ? a.l[i] : d.l[i];
        return r;
    }
    static FF_HD fe add(const fe& a, const fe& b) {
        fe s;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)a.l[i] + b.l[i] + c;
            s.l[i] = (uint32_t)t;
            c = t >> 32;
        }
        return reduce_once(s);  // 2*MOD < 2^256: no carry out of limb 7
    }
    static FF_HD fe sub(const fe& a, const fe& b) {
        fe d;
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)a.l[i] - b.l[i] - borrow;
            d.l[i] = (uint32_t)t;
            borrow = (t >> 63) & 1;
        }
        uint32_t mask = borrow ? 0xffffffffu : 0u;
        fe r;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)d.l[i] + (Pm::MOD[i] & mask) + c;
            r.l[i] = (uint32_t)t;
            c = t >> 32;
        }
        return r;
    }
    static FF_HD fe neg(const fe& a) { return is_zero(a) ? a : sub(zero(), a); }
    static FF_HD fe dbl(const fe& a) { return add(a, a); }

    // Montgomery product a*b*R^-1 mod MOD, result fully reduced.
    // Host path: merged ("no-carry") CIOS in portable C++.
    // Device path: finely-integrated product scanning with a 96-bit column accumulator driven by
    // v_mad_u64_u32's carry-out (ff_macc.inc): 128 mads + 128 addc + ~46 moves per product, versus
    // 128 mads + 122 64-bit adds + 333 moves for what hipcc makes of the portable CIOS.
    static FF_HD fe mul(const fe& a, const fe& b) {
#if defined(__HIP_DEVICE_COMPILE__)
        return reduce_once(mul_nr(a, b));
#else
        return mul_host(a, b);
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // the product before its final conditional subtraction: for inputs < 2 MOD the result is < MOD (1 + 4 MOD / R)
    // < 1.76 MOD (R = 2^256, MOD < 2^254), i.e. again < 2 MOD -- the "lazy" range the EC formulas work in
    static __device__ __forceinline__ fe mul_nr(const fe& a, const fe& b) {
        uint64_t lo = 0;
        uint32_t hi = 0;
        uint32_t m[8], r[8];
        const uint32_t* A = a.l;
        const uint32_t* B = b.l;
#define P_(j) Pm::MOD[j]
#define SHIFT_() lo = (lo >> 32) | ((uint64_t)hi << 32); hi = 0
#define MSTEP_(k) m[k] = (uint32_t)lo * Pm::INV; MACC1_VS(lo, hi, m[k], P_(0)); SHIFT_()
        MACC1_VV(lo, hi, A[0], B[0]);
        MSTEP_(0);
        MACC2_VV(lo, hi, A[0], A[1], B[1], B[0]);
        MACC1_VS(lo, hi, m[0], P_(1));
        MSTEP_(1);
        MACC3_VV(lo, hi, A[0], A[1], A[2], B[2], B[1], B[0]);
        MACC2_VS(lo, hi, m[0], m[1], P_(2), P_(1));
        MSTEP_(2);
        MACC4_VV(lo, hi, A[0], A[1], A[2], A[3], B[3], B[2], B[1], B[0]);
        MACC3_VS(lo, hi, m[0], m[1], m[2], P_(3), P_(2), P_(1));
        MSTEP_(3);
        MACC5_VV(lo, hi, A[0], A[1], A[2], A[3], A[4], B[4], B[3], B[2], B[1], B[0]);
        MACC4_VS(lo, hi, m[0], m[1], m[2], m[3], P_(4), P_(3), P_(2), P_(1));
        MSTEP_(4);
        MACC6_VV(lo, hi, A[0], A[1], A[2], A[3], A[4], A[5], B[5], B[4], B[3], B[2], B[1], B[0]);
        MACC5_VS(lo, hi, m[0], m[1], m[2], m[3], m[4], P_(5), P_(4), P_(3), P_(2), P_(1));
        MSTEP_(5);
        MACC7_VV(lo, hi, A[0], A[1], A[2], A[3], A[4], A[5], A[6], B[6], B[5], B[4], B[3], B[2], B[1], B[0]);
        MACC6_VS(lo, hi, m[0], m[1], m[2], m[3], m[4], m[5], P_(6), P_(5), P_(4), P_(3), P_(2), P_(1));
        MSTEP_(6);
        MACC8_VV(lo, hi, A[0], A[1], A[2], A[3], A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4], B[3], B[2], B[1], B[0]);
        MACC7_VS(lo, hi, m[0], m[1], m[2], m[3], m[4], m[5], m[6], P_(7), P_(6), P_(5), P_(4), P_(3), P_(2), P_(1));
        MSTEP_(7);
        MACC7_VV(lo, hi, A[1], A[2], A[3], A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4], B[3], B[2], B[1]);
        MACC7_VS(lo, hi, m[1], m[2], m[3], m[4], m[5], m[6], m[7], P_(7), P_(6), P_(5), P_(4), P_(3), P_(2), P_(1));
        r[0] = (uint32_t)lo; SHIFT_();
        MACC6_VV(lo, hi, A[2], A[3], A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4], B[3], B[2]);
        MACC6_VS(lo, hi, m[2], m[3], m[4], m[5], m[6], m[7], P_(7), P_(6), P_(5), P_(4), P_(3), P_(2));
        r[1] = (uint32_t)lo; SHIFT_();
        MACC5_VV(lo, hi, A[3], A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4], B[3]);
        MACC5_VS(lo, hi, m[3], m[4], m[5], m[6], m[7], P_(7), P_(6), P_(5), P_(4), P_(3));
        r[2] = (uint32_t)lo; SHIFT_();
        MACC4_VV(lo, hi, A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4]);
        MACC4_VS(lo, hi, m[4], m[5], m[6], m[7], P_(7), P_(6), P_(5), P_(4));
        r[3] = (uint32_t)lo; SHIFT_();
        MACC3_VV(lo, hi, A[5], A[6], A[7], B[7], B[6], B[5]);
        MACC3_VS(lo, hi, m[5], m[6], m[7], P_(7), P_(6), P_(5));
        r[4] = (uint32_t)lo; SHIFT_();
        MACC2_VV(lo, hi, A[6], A[7], B[7], B[6]);
        MACC2_VS(lo, hi, m[6], m[7], P_(7), P_(6));
        r[5] = (uint32_t)lo; SHIFT_();
        MACC1_VV(lo, hi, A[7], B[7]);
        MACC1_VS(lo, hi, m[7], P_(7));
        r[6] = (uint32_t)lo;
        r[7] = (uint32_t)(lo >> 32);
#undef P_
#undef SHIFT_
#undef MSTEP_
        fe o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.l[i] = r[i];
        return o;
    }
#endif
    // portable 8 x 32 CIOS (kept as the reference the 64-bit host path is checked against: tests/host_field_selftest)
    static inline fe mul_host32(const fe& a, const fe& b) {
        uint32_t t[8];
        for (int i = 0; i < 8; i++) t[i] = 0;
        for (int i = 0; i < 8; i++) {
            uint64_t A = (uint64_t)a.l[0] * b.l[i] + t[0];
            uint32_t m = (uint32_t)A * Pm::INV;
            uint64_t C = (uint64_t)m * Pm::MOD[0] + (uint32_t)A;
            A >>= 32;
            C >>= 32;
            for (int j = 1; j < 8; j++) {
                A += (uint64_t)a.l[j] * b.l[i] + t[j];
                C += (uint64_t)m * Pm::MOD[j] + (uint32_t)A;
                t[j - 1] = (uint32_t)C;
                A >>= 32;
                C >>= 32;
            }
            t[7] = (uint32_t)(A + C);
        }
        fe r;
        for (int i = 0; i < 8; i++) r.l[i] = t[i];
        return reduce_once(r);
    }
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
    // -MOD^-1 mod 2^64 from the 32-bit constant (one Newton step on the inverse)
    static constexpr uint64_t ninv64() {
        const uint64_t m0 = (uint64_t)Pm::MOD[0] | ((uint64_t)Pm::MOD[1] << 32);
        const uint64_t x32 = (uint64_t)(uint32_t)(0u - Pm::INV);  // MOD^-1 mod 2^32
        const uint64_t x64 = x32 * (2ull - m0 * x32);             // MOD^-1 mod 2^64
        return 0ull - x64;
    }
    // Host path: CIOS on 4 x 64-bit limbs with 128-bit products (the round callbacks, the coordinator and the verifier run ~25 field
    // products per sumcheck round on the host; the 8 x 32 loop above cost ~4 us of every round's critical path)
    static inline fe mul_host(const fe& a, const fe& b) {
        typedef unsigned __int128 u128;
        uint64_t A[4], B[4], M[4], t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            A[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
            B[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
            M[i] = (uint64_t)Pm::MOD[2 * i] | ((uint64_t)Pm::MOD[2 * i + 1] << 32);
        }
        constexpr uint64_t ninv = ninv64();
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)A[j] * B[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (uint64_t)c;
            t[5] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * ninv;
            c = (u128)m * M[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; j++) {
                c += (u128)m * M[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = t[5] + (uint64_t)(c >> 64);
        }
        fe r;
        for (int i = 0; i < 4; i++) {
            r.l[2 * i] = (uint32_t)t[i];
            r.l[2 * i + 1] = (uint32_t)(t[i] >> 32);
        }
        return reduce_once(r);  // inputs < MOD: t < 2 MOD and t[4] = 0
    }
#else
    static inline fe mul_host(const fe& a, const fe& b) { return mul_host32(a, b); }
#endif
    static FF_HD fe sqr(const fe& a) { return mul(a, a); }
    // (a*b + c*d) * R^-1 mod MOD with ONE Montgomery reduction: both products accumulate into the same column
    // sums (16 products + 8 reduction terms per column stay far below the 96-bit accumulator), the result is
    // < p (2p/R + 1) < 1.38 p, so a single conditional subtraction finishes it.  200 mads instead of the 272 of
    // two separate products + an addition: used for Y3 = R (Q - X3) - Y1 PPP of the XYZZ additions.
    static FF_HD fe mul_add2(const fe& a, const fe& b, const fe& c, const fe& d) {
#if defined(__HIP_DEVICE_COMPILE__)
        return reduce_once(mul_add2_nr(a, b, c, d));
#else
        return add(mul(a, b), mul(c, d));
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // before the final subtraction: < MOD (1 + 2 MOD / R) for canonical inputs, < MOD (1 + 8 MOD / R) < 2.52 MOD for
    // inputs < 2 MOD
    static __device__ __forceinline__ fe mul_add2_nr(const fe& a, const fe& b, const fe& c, const fe& d) {
        uint64_t lo = 0;
        uint32_t hi = 0;
        uint32_t m[8], r[8];
        const uint32_t* A = a.l;
        const uint32_t* B = b.l;
        const uint32_t* C = c.l;
        const uint32_t* D = d.l;
#define P_(j) Pm::MOD[j]
#define SHIFT_() lo = (lo >> 32) | ((uint64_t)hi << 32); hi = 0
#define MSTEP_(k) m[k] = (uint32_t)lo * Pm::INV; MACC1_VS(lo, hi, m[k], P_(0)); SHIFT_()
        MACC1_VV(lo, hi, A[0], B[0]);
        MACC1_VV(lo, hi, C[0], D[0]);
        MSTEP_(0);
        MACC2_VV(lo, hi, A[0], A[1], B[1], B[0]);
        MACC2_VV(lo, hi, C[0], C[1], D[1], D[0]);
        MACC1_VS(lo, hi, m[0], P_(1));
        MSTEP_(1);
        MACC3_VV(lo, hi, A[0], A[1], A[2], B[2], B[1], B[0]);
        MACC3_VV(lo, hi, C[0], C[1], C[2], D[2], D[1], D[0]);
        MACC2_VS(lo, hi, m[0], m[1], P_(2), P_(1));
        MSTEP_(2);
        MACC4_VV(lo, hi, A[0], A[1], A[2], A[3], B[3], B[2], B[1], B[0]);
        MACC4_VV(lo, hi, C[0], C[1], C[2], C[3], D[3], D[2], D[1], D[0]);
        MACC3_VS(lo, hi, m[0], m[1], m[2], P_(3), P_(2), P_(1));
        MSTEP_(3);
        MACC5_VV(lo, hi, A[0], A[1], A[2], A[3], A[4], B[4], B[3], B[2], B[1], B[0]);
        MACC5_VV(lo, hi, C[0], C[1], C[2], C[3], C[4], D[4], D[3], D[2], D[1], D[0]);
        MACC4_VS(lo, hi, m[0], m[1], m[2], m[3], P_(4), P_(3), P_(2), P_(1));
        MSTEP_(4);
        MACC6_VV(lo, hi, A[0], A[1], A[2], A[3], A[4], A[5], B[5], B[4], B[3], B[2], B[1], B[0]);
        MACC6_VV(lo, hi, C[0], C[1], C[2], C[3], C[4], C[5], D[5], D[4], D[3], D[2], D[1], D[0]);
        MACC5_VS(lo, hi, m[0], m[1], m[2], m[3], m[4], P_(5), P_(4), P_(3), P_(2), P_(1));
        MSTEP_(5);
        MACC7_VV(lo, hi, A[0], A[1], A[2], A[3], A[4], A[5], A[6], B[6], B[5], B[4], B[3], B[2], B[1], B[0]);
        MACC7_VV(lo, hi, C[0], C[1], C[2], C[3], C[4], C[5], C[6], D[6], D[5], D[4], D[3], D[2], D[1], D[0]);
        MACC6_VS(lo, hi, m[0], m[1], m[2], m[3], m[4], m[5], P_(6), P_(5), P_(4), P_(3), P_(2), P_(1));
        MSTEP_(6);
        MACC8_VV(lo, hi, A[0], A[1], A[2], A[3], A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4], B[3], B[2], B[1], B[0]);
        MACC8_VV(lo, hi, C[0], C[1], C[2], C[3], C[4], C[5], C[6], C[7], D[7], D[6], D[5], D[4], D[3], D[2], D[1], D[0]);
        MACC7_VS(lo, hi, m[0], m[1], m[2], m[3], m[4], m[5], m[6], P_(7), P_(6), P_(5), P_(4), P_(3), P_(2), P_(1));
        MSTEP_(7);
        MACC7_VV(lo, hi, A[1], A[2], A[3], A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4], B[3], B[2], B[1]);
        MACC7_VV(lo, hi, C[1], C[2], C[3], C[4], C[5], C[6], C[7], D[7], D[6], D[5], D[4], D[3], D[2], D[1]);
        MACC7_VS(lo, hi, m[1], m[2], m[3], m[4], m[5], m[6], m[7], P_(7), P_(6), P_(5), P_(4), P_(3), P_(2), P_(1));
        r[0] = (uint32_t)lo; SHIFT_();
        MACC6_VV(lo, hi, A[2], A[3], A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4], B[3], B[2]);
        MACC6_VV(lo, hi, C[2], C[3], C[4], C[5], C[6], C[7], D[7], D[6], D[5], D[4], D[3], D[2]);
        MACC6_VS(lo, hi, m[2], m[3], m[4], m[5], m[6], m[7], P_(7), P_(6), P_(5), P_(4), P_(3), P_(2));
        r[1] = (uint32_t)lo; SHIFT_();
        MACC5_VV(lo, hi, A[3], A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4], B[3]);
        MACC5_VV(lo, hi, C[3], C[4], C[5], C[6], C[7], D[7], D[6], D[5], D[4], D[3]);
        MACC5_VS(lo, hi, m[3], m[4], m[5], m[6], m[7], P_(7), P_(6), P_(5), P_(4), P_(3));
        r[2] = (uint32_t)lo; SHIFT_();
        MACC4_VV(lo, hi, A[4], A[5], A[6], A[7], B[7], B[6], B[5], B[4]);
        MACC4_VV(lo, hi, C[4], C[5], C[6], C[7], D[7], D[6], D[5], D[4]);
        MACC4_VS(lo, hi, m[4], m[5], m[6], m[7], P_(7), P_(6), P_(5), P_(4));
        r[3] = (uint32_t)lo; SHIFT_();
        MACC3_VV(lo, hi, A[5], A[6], A[7], B[7], B[6], B[5]);
        MACC3_VV(lo, hi, C[5], C[6], C[7], D[7], D[6], D[5]);
        MACC3_VS(lo, hi, m[5], m[6], m[7], P_(7), P_(6), P_(5));
        r[4] = (uint32_t)lo; SHIFT_();
        MACC2_VV(lo, hi, A[6], A[7], B[7], B[6]);
        MACC2_VV(lo, hi, C[6], C[7], D[7], D[6]);
        MACC2_VS(lo, hi, m[6], m[7], P_(7), P_(6));
        r[5] = (uint32_t)lo; SHIFT_();
        MACC1_VV(lo, hi, A[7], B[7]);
        MACC1_VV(lo, hi, C[7], D[7]);
        MACC1_VS(lo, hi, m[7], P_(7));
        r[6] = (uint32_t)lo;
        r[7] = (uint32_t)(lo >> 32);
#undef P_
#undef SHIFT_
#undef MSTEP_
        fe o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.l[i] = r[i];
        return o;
    }
#endif
    // a*b - c*d with one reduction
    static FF_HD fe mul_sub2(const fe& a, const fe& b, const fe& c, const fe& d) { return mul_add2(a, b, neg(c), d); }

    // Two independent products r1 = a*b, r2 = c*d.  On the device their column-accumulator chains are
    // interleaved instruction by instruction (ff_mul2.inc): at the 4 waves/SIMD the EC kernels run at, one
    // chain per wave reaches ~105 G mul/s while two interleaved chains reach the ~128 G mul/s peak.
    static FF_HD void mul2(const fe& a, const fe& b, const fe& c, const fe& d, fe& o1, fe& o2) {
#if defined(__HIP_DEVICE_COMPILE__)
        uint64_t lo1 = 0, lo2 = 0;
        uint32_t hi1 = 0, hi2 = 0;
        uint32_t m1[8], m2[8], r1[8], r2[8];
        const uint32_t* A = a.l;
        const uint32_t* B = b.l;
        const uint32_t* C = c.l;
        const uint32_t* D = d.l;
        FF_MUL2_BODY
        fe t1, t2;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            t1.l[i] = r1[i];
            t2.l[i] = r2[i];
        }
        o1 = reduce_once(t1);
        o2 = reduce_once(t2);
#else
        o1 = mul(a, b);
        o2 = mul(c, d);
#endif
    }

    // ---- lazy range [0, 2 MOD) for the EC formulas (device): products skip their final conditional subtraction
    // (16 of ~330 instructions each), additions / subtractions correct by 2 MOD instead of MOD, zero tests accept
    // 0 and MOD.  4 MOD < 2^256, so nothing overflows.  On the host these are the canonical operations (values
    // agree mod MOD; only canonical results -- to_affine -- are ever compared or serialised).
    static FF_HD fe two_mod() {
        fe r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            r.l[i] = (Pm::MOD[i] << 1) | c;
            c = Pm::MOD[i] >> 31;
        }
        return r;
    }
    static FF_HD fe lcanon(const fe& a) {
#if defined(__HIP_DEVICE_COMPILE__)
        return reduce_once(a);
#else
        return a;
#endif
    }
    static FF_HD fe lmul(const fe& a, const fe& b) {
#if defined(__HIP_DEVICE_COMPILE__)
        return mul_nr(a, b);
#else
        return mul(a, b);
#endif
    }
    static FF_HD void lmul2(const fe& a, const fe& b, const fe& c, const fe& d, fe& o1, fe& o2) {
#if defined(__HIP_DEVICE_COMPILE__)
        uint64_t lo1 = 0, lo2 = 0;
        uint32_t hi1 = 0, hi2 = 0;
        uint32_t m1[8], m2[8], r1[8], r2[8];
        const uint32_t* A = a.l;
        const uint32_t* B = b.l;
        const uint32_t* C = c.l;
        const uint32_t* D = d.l;
        FF_MUL2_BODY
#pragma unroll
        for (int i = 0; i < 8; i++) {
            o1.l[i] = r1[i];
            o2.l[i] = r2[i];
        }
#else
        o1 = mul(a, b);
        o2 = mul(c, d);
#endif
    }
    static FF_HD fe ladd(const fe& a, const fe& b) {
#if defined(__HIP_DEVICE_COMPILE__)
        fe s;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)a.l[i] + b.l[i] + c;
            s.l[i] = (uint32_t)t;
            c = t >> 32;
        }
        const fe tm = two_mod();
        fe d;
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)s.l[i] - tm.l[i] - borrow;
            d.l[i] = (uint32_t)t;
            borrow = (t >> 63) & 1;
        }
        fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = borrow ? s.l[i] : d.l[i];
        return r;
#else
        return add(a, b);
#endif
    }
    static FF_HD fe lsub(const fe& a, const fe& b) {
#if defined(__HIP_DEVICE_COMPILE__)
        fe d;
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)a.l[i] - b.l[i] - borrow;
            d.l[i] = (uint32_t)t;
            borrow = (t >> 63) & 1;
        }
        const fe tm = two_mod();
        uint32_t mask = borrow ? 0xffffffffu : 0u;
        fe r;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)d.l[i] + (tm.l[i] & mask) + c;
            r.l[i] = (uint32_t)t;
            c = t >> 32;
        }
        return r;
#else
        return sub(a, b);
#endif
    }
    static FF_HD fe ldbl(const fe& a) { return ladd(a, a); }
    static FF_HD fe lneg(const fe& a) { return lsub(zero(), a); }  // 0 -> 0, otherwise 2 MOD - a
    static FF_HD bool lis_zero(const fe& a) {
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t z = 0, m = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            z |= a.l[i];
            m |= a.l[i] ^ Pm::MOD[i];
        }
        return z == 0 || m == 0;
#else
        return is_zero(a);
#endif
    }
    // a*b - c*d in the lazy range: one reduction, < 2.52 MOD before, one conditional subtraction of MOD after
    static FF_HD fe lmul_sub2(const fe& a, const fe& b, const fe& c, const fe& d) {
#if defined(__HIP_DEVICE_COMPILE__)
        fe x = mul_add2_nr(a, b, lneg(c), d);
        const fe tm = two_mod();
        // x >= 2 MOD ? x - MOD : x
        fe t;
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t u = (uint64_t)x.l[i] - tm.l[i] - borrow;
            t.l[i] = (uint32_t)u;
            borrow = (u >> 63) & 1;
        }
        uint32_t mask = borrow ? 0u : 0xffffffffu;  // no borrow: x >= 2 MOD
        fe r;
        uint64_t bb = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t u = (uint64_t)x.l[i] - (Pm::MOD[i] & mask) - bb;
            r.l[i] = (uint32_t)u;
            bb = (u >> 63) & 1;
        }
        return r;
#else
        return mul_sub2(a, b, c, d);
#endif
    }

    static FF_HD fe to_mont(const fe& a) { return mul(a, r2()); }
    static FF_HD fe from_mont(const fe& a) {
#if defined(__HIP_DEVICE_COMPILE__)
        // REDC of the 256-bit value itself: the product-scanning Montgomery product above with the a * b columns replaced
        // by a's limbs -- 36 + 28 multiply-adds instead of 128, and no second operand to hold in registers
        uint64_t lo = 0;
        uint32_t hi = 0;
        uint32_t m[8], r[8];
        const uint32_t* A = a.l;
#define P_(j) Pm::MOD[j]
#define SHIFT_() lo = (lo >> 32) | ((uint64_t)hi << 32); hi = 0
#define ADDA_(k) { uint64_t t_ = lo + A[k]; hi += t_ < lo ? 1u : 0u; lo = t_; }
#define MSTEP_(k) m[k] = (uint32_t)lo * Pm::INV; MACC1_VS(lo, hi, m[k], P_(0)); SHIFT_()
        ADDA_(0);
        MSTEP_(0);
        ADDA_(1);
        MACC1_VS(lo, hi, m[0], P_(1));
        MSTEP_(1);
        ADDA_(2);
        MACC2_VS(lo, hi, m[0], m[1], P_(2), P_(1));
        MSTEP_(2);
        ADDA_(3);
        MACC3_VS(lo, hi, m[0], m[1], m[2], P_(3), P_(2), P_(1));
        MSTEP_(3);
        ADDA_(4);
        MACC4_VS(lo, hi, m[0], m[1], m[2], m[3], P_(4), P_(3), P_(2), P_(1));
        MSTEP_(4);
        ADDA_(5);
        MACC5_VS(lo, hi, m[0], m[1], m[2], m[3], m[4], P_(5), P_(4), P_(3), P_(2), P_(1));
        MSTEP_(5);
        ADDA_(6);
        MACC6_VS(lo, hi, m[0], m[1], m[2], m[3], m[4], m[5], P_(6), P_(5), P_(4), P_(3), P_(2), P_(1));
        MSTEP_(6);
        ADDA_(7);
        MACC7_VS(lo, hi, m[0], m[1], m[2], m[3], m[4], m[5], m[6], P_(7), P_(6), P_(5), P_(4), P_(3), P_(2), P_(1));
        MSTEP_(7);
        MACC7_VS(lo, hi, m[1], m[2], m[3], m[4], m[5], m[6], m[7], P_(7), P_(6), P_(5), P_(4), P_(3), P_(2), P_(1));
        r[0] = (uint32_t)lo; SHIFT_();
        MACC6_VS(lo, hi, m[2], m[3], m[4], m[5], m[6], m[7], P_(7), P_(6), P_(5), P_(4), P_(3), P_(2));
        r[1] = (uint32_t)lo; SHIFT_();
        MACC5_VS(lo, hi, m[3], m[4], m[5], m[6], m[7], P_(7), P_(6), P_(5), P_(4), P_(3));
        r[2] = (uint32_t)lo; SHIFT_();
        MACC4_VS(lo, hi, m[4], m[5], m[6], m[7], P_(7), P_(6), P_(5), P_(4));
        r[3] = (uint32_t)lo; SHIFT_();
        MACC3_VS(lo, hi, m[5], m[6], m[7], P_(7), P_(6), P_(5));
        r[4] = (uint32_t)lo; SHIFT_();
        MACC2_VS(lo, hi, m[6], m[7], P_(7), P_(6));
        r[5] = (uint32_t)lo; SHIFT_();
        MACC1_VS(lo, hi, m[7], P_(7));
        r[6] = (uint32_t)lo;
        r[7] = (uint32_t)(lo >> 32);
#undef P_
#undef SHIFT_
#undef ADDA_
#undef MSTEP_
        fe o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.l[i] = r[i];
        return reduce_once(o);
#else
        fe o = zero();
        o.l[0] = 1;
        return mul(a, o);
#endif
    }
    static FF_HD fe from_u64(uint64_t v) {
        fe o = zero();
        o.l[0] = (uint32_t)v;
        o.l[1] = (uint32_t)(v >> 32);
        return to_mont(o);
    }
    // a^(e) with e given as 8 u32 limbs (non-Montgomery exponent)
    static FF_HD fe pow(const fe& a, const uint32_t e[8]) {
        fe acc = one();
        for (int i = 7; i >= 0; i--) {
            for (int b = 31; b >= 0; b--) {
                acc = sqr(acc);
                if ((e[i] >> b) & 1) acc = mul(acc, a);
            }
        }
        return acc;
    }
    // Fermat inverse a^(MOD-2); inv(0) = 0
    static FF_HD fe inv(const fe& a) {
        uint32_t e[8];
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = Pm::MOD[i];
        e[0] -= 2;  // low limb of both moduli is >= 2
        return pow(a, e);
    }
};

typedef Field<FrParams> Fr;
typedef Field<FqParams> Fq;

// 16-byte vector load/store of one field element (2 x dwordx4)
static __device__ __forceinline__ fe fe_load(const fe* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    fe r;
    r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
    r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
    return r;
}
static __device__ __forceinline__ void fe_store(fe* p, const fe& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
