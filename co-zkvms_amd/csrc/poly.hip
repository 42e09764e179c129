// Polynomial seam: device-resident multilinear polynomials over shares.
//
//   cozk_poly     <-> Rep3DensePolynomial (co-jolt/src/poly/dense_mlpoly.rs:23-32) / plain DensePolynomial
//   cozk_layer    <-> Rep3DenseInterleavedPolynomial (co-jolt/src/poly/dense_interleaved_poly.rs:35-48)
//   cozk_spliteq  <-> SplitEqPolynomial (jolt-core; used dense_interleaved_poly.rs:218-303)
//
// All kernels are streaming kernels on 32-byte Montgomery elements (16-byte vector loads per lane);
// round results (3-4 field elements) are reduced in-kernel (wave shuffle tree + LDS), finished by a
// one-block kernel, and returned through pinned host memory.
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <string.h>
#include "poly.hip.hpp"
#include "prf.hip.hpp"
#include "fr9.hip.hpp"

static constexpr int PT = 256;      // threads per block
static constexpr int MAXBLK = 2048; // grid cap for reducing kernels (>= 8 blocks per CU)

struct cozk_poly {
    cozk_ctx* ctx;
    int mode;            // 1 plain, 2 rep3
    fe* a0; fe* b0;      // unbound coefficients (`coeffs`, immutable; possibly a borrowed chunk view)
    bool own0;
    fe* buf[2][2];       // ping-pong bound buffers [which][component]
    size_t cap[2];
    int cur;             // -1: unbound, else index of the live bound buffer
    size_t len;
    size_t orig_len;
};

struct cozk_layer {
    cozk_ctx* ctx;
    int mode;
    fe* buf[2][2];       // ping-pong [which][component]
    size_t cap[2];
    int cur;
    size_t len;
};

struct cozk_spliteq {
    cozk_ctx* ctx;
    fe* E1[2]; fe* E2[2];   // ping-pong
    int c1, c2;
    size_t E1_len, E2_len, E1_cap, E2_cap;
    int num_vars;
};

static inline unsigned grid_for(size_t n) { return (unsigned)((n + PT - 1) / PT); }
static inline unsigned grid_capped(size_t n) {
    size_t g = (n + PT - 1) / PT;
    return (unsigned)(g > MAXBLK ? MAXBLK : (g ? g : 1));
}

// Grid of a grid-stride kernel whose waves are long (thousands of instructions per element): exactly the workgroups the chip
// holds at once (occupancy query x CUs), so that every wave is resident from the start and the last residency round is not a
// partial one.  Measured on the 2^24-element probe with the LDS-staged 9 x 29 kernels (compute-bound): 0.311 vs 0.334 ms plain,
// 0.486 vs 0.505 ms Rep3 against the former fixed cap of 1024 workgroups (COZK_RESIDENT_GRID=0 restores it for A/B runs).
static unsigned resident_grid(const void* kernel, size_t nblocks_needed, int device) {
    static const int legacy = getenv("COZK_RESIDENT_GRID") && atoi(getenv("COZK_RESIDENT_GRID")) == 0;  // A/B switch: the old cap
    if (legacy) return (unsigned)(nblocks_needed < 1024 ? (nblocks_needed ? nblocks_needed : 1) : 1024);
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, unsigned> cache;
    unsigned cap;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find({kernel, device});
        if (it == cache.end()) {
            int per_cu = 0, cus = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, PT, 0));
            HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
            unsigned v = (unsigned)(per_cu > 0 ? per_cu : 1) * (unsigned)(cus > 0 ? cus : 1);
            if (v > MAXBLK) v = MAXBLK;
            it = cache.emplace(std::make_pair(kernel, device), v).first;
        }
        cap = it->second;
    }
    return (unsigned)(nblocks_needed < cap ? (nblocks_needed ? nblocks_needed : 1) : cap);
}

static const fe* poly_a(const cozk_poly* p) { return p->cur < 0 ? p->a0 : p->buf[p->cur][0]; }
static const fe* poly_b(const cozk_poly* p) { return p->cur < 0 ? p->b0 : p->buf[p->cur][1]; }

// ------------------------------------------------------------------ kernels: dense polynomial
// bind: out[i] = lo + r (hi - lo); LowToHigh pairs (2i, 2i+1), HighToLow pairs (i, i + n)
// (dense_mlpoly.rs:310-378).  Algorithmic traffic: read n shares, write n/2 = 96 n bytes (REP3).
template <int NC, int ORDER>
__global__ void __launch_bounds__(PT) k_poly_bind(const fe* __restrict__ ia, const fe* __restrict__ ib, fe* oa, fe* ob,
                                               size_t n_out, fe r) {
    size_t i = (size_t)blockIdx.x * PT + threadIdx.x;
    if (i >= n_out) return;
    size_t lo_i = ORDER == COZK_LOW_TO_HIGH ? 2 * i : i;
    size_t hi_i = ORDER == COZK_LOW_TO_HIGH ? 2 * i + 1 : i + n_out;
    Sh<NC> lo = sh_load<NC>(ia, ib, lo_i), hi = sh_load<NC>(ia, ib, hi_i);
    sh_store<NC>(oa, ob, i, sh_lerp<NC>(lo, hi, r));
}

// partial[blockIdx.y * gridDim.x + blockIdx.x] = sum over this block's stride of (a+b) * chi
// (evaluate_at_chi, dense_mlpoly.rs:160-181; TWO_INV folded in by the finishing kernel)
template <int NC>
__global__ void __launch_bounds__(PT) k_poly_eval_chi(const fe* const* __restrict__ pa, const fe* const* __restrict__ pb,
                                                   const size_t* __restrict__ lens, const fe* __restrict__ chi, int k,
                                                   fe* __restrict__ partial) {
    __shared__ fe sh4[4];
    // a workgroup row serves TWO polynomials, so chi[i] is fetched once per pair (chi is 32 MiB at 2^20: it does not stay in
    // L2 between the polynomials' passes); each product goes into its polynomial's wide accumulator (poly.hip.hpp: lazy dot
    // products, one reduction per lane at the end)
    const int p0 = 2 * blockIdx.y, p1 = p0 + 1;
    const bool two = p1 < k;
    const fe* a0 = pa[p0];
    const fe* b0 = NC == 2 ? pb[p0] : nullptr;
    const fe* a1 = two ? pa[p1] : nullptr;
    const fe* b1 = (two && NC == 2) ? pb[p1] : nullptr;
    const size_t n0 = lens[p0], n1 = two ? lens[p1] : 0;
    const size_t n = n0 > n1 ? n0 : n1;
    FrWide w0, w1;
    fr_wide_zero(w0);
    fr_wide_zero(w1);
    for (size_t i = (size_t)blockIdx.x * PT + threadIdx.x; i < n; i += (size_t)gridDim.x * PT) {
        fe c = fe_load(chi + i);
        if (i < n0) fr_wide_mac(w0, sh_ab_sum<NC>(sh_load<NC>(a0, b0, i)), c);
        if (i < n1) fr_wide_mac(w1, sh_ab_sum<NC>(sh_load<NC>(a1, b1, i)), c);
    }
    fe acc = fr_block_sum(fr_wide_reduce(w0), sh4);
    if (threadIdx.x == 0) fe_store(partial + (size_t)p0 * gridDim.x + blockIdx.x, acc);
    if (two) {  // uniform per workgroup
        acc = fr_block_sum(fr_wide_reduce(w1), sh4);
        if (threadIdx.x == 0) fe_store(partial + (size_t)p1 * gridDim.x + blockIdx.x, acc);
    }
}

// out[y] = scale * sum_x partial[y * nper + x]
// ... and, when `pub` is armed, the LAST of its (few) workgroups publishes the round's sequence number into the pinned word the host
// spins on: no stream memory write behind the kernel (~560 per chained proof, 3.5 us of GPU time + a launch gap each).
struct RoundPublish {
    unsigned* ticket;  // device counter, zero between launches; nullptr: nothing to publish
    uint32_t* flag;    // pinned word
    uint32_t seq;
};
__global__ void __launch_bounds__(PT) k_finish_sums(const fe* __restrict__ partial, unsigned nper, fe scale, int apply_scale,
                                                 fe* __restrict__ out, RoundPublish pub) {
    __shared__ fe sh4[4];
    fe acc = Fr::zero();
    for (unsigned x = threadIdx.x; x < nper; x += PT) acc = Fr::add(acc, fe_load(partial + (size_t)blockIdx.x * nper + x));
    acc = fr_block_sum(acc, sh4);
    if (threadIdx.x == 0) {
        fe_store(out + blockIdx.x, apply_scale ? Fr::mul(acc, scale) : acc);
        if (pub.ticket) {
            __threadfence_system();  // this workgroup's result is in host memory before its ticket
            if (atomicAdd(pub.ticket, 1u) == gridDim.x - 1) {
                *pub.ticket = 0;
                __threadfence_system();
                __hip_atomic_store(pub.flag, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// per-component dot product with a public vector (dot_product_with_public, dense_mlpoly.rs:228-234)
template <int NC>
__global__ void __launch_bounds__(PT) k_poly_dot_public(const fe* __restrict__ a, const fe* __restrict__ b,
                                                     const fe* __restrict__ pub, size_t n, fe* __restrict__ partial) {
    __shared__ fe sh4[4];
    FrWide w[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) fr_wide_zero(w[k]);
    for (size_t i = (size_t)blockIdx.x * PT + threadIdx.x; i < n; i += (size_t)gridDim.x * PT) {
        Sh<NC> s = sh_load<NC>(a, b, i);
        fe p = fe_load(pub + i);
#pragma unroll
        for (int k = 0; k < NC; k++) fr_wide_mac(w[k], s.c[k], p);
    }
#pragma unroll
    for (int k = 0; k < NC; k++) {
        fe v = fr_block_sum(fr_wide_reduce(w[k]), sh4);
        if (threadIdx.x == 0) fe_store(partial + (size_t)k * gridDim.x + blockIdx.x, v);
    }
}

// out[i] = sum_k coeff[k] * poly_k[i] (i < len_k)   (linear_combination, dense_mlpoly.rs:195-226;
// public polynomials enter through their trivial share, multilinear_polynomial.rs:196-296)
template <int NC>
__global__ void __launch_bounds__(PT) k_poly_lincomb(const fe* const* __restrict__ pa, const fe* const* __restrict__ pb,
                                                  const size_t* __restrict__ lens, const fe* __restrict__ coeffs, int k,
                                                  fe* oa, fe* ob, size_t n) {
    size_t i = (size_t)blockIdx.x * PT + threadIdx.x;
    if (i >= n) return;
    Sh<NC> acc;
    if (k >= 8) {
        // long combinations (the rho-RLC of all committed polynomials): one reduction per output element instead of k
        FrWide w[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) fr_wide_zero(w[c]);
        for (int j = 0; j < k; j++) {
            if (i < lens[j]) {
                fe cf = fe_load(coeffs + j);
                if (pa[j]) fr_wide_mac(w[0], fe_load(pa[j] + i), cf);
                if (NC == 2 && pb[j]) fr_wide_mac(w[NC - 1], fe_load(pb[j] + i), cf);
            }
        }
#pragma unroll
        for (int c = 0; c < NC; c++) acc.c[c] = fr_wide_reduce(w[c]);
        sh_store<NC>(oa, ob, i, acc);
        return;
    }
    for (int c = 0; c < NC; c++) acc.c[c] = Fr::zero();
    for (int j = 0; j < k; j++) {
        if (i < lens[j]) {
            fe cf = fe_load(coeffs + j);
            if (pa[j]) acc.c[0] = Fr::add(acc.c[0], Fr::mul(fe_load(pa[j] + i), cf));
            if (NC == 2 && pb[j]) acc.c[NC - 1] = Fr::add(acc.c[NC - 1], Fr::mul(fe_load(pb[j] + i), cf));
        }
    }
    sh_store<NC>(oa, ob, i, acc);
}

// the same with the (<= 16) operand pointers passed by value in the kernel arguments: the opening-reduction
// sumcheck calls this once per round, and four metadata uploads + a drain per round cost more than the kernel
struct OpenQuadArgs {
    const fe* a[16];
    const fe* b[16];
    const fe* eq[16];
    size_t half[16];
};
template <int NC>
__global__ void __launch_bounds__(PT) k_open_quadratic_args(OpenQuadArgs args, fe* __restrict__ partial) {
    __shared__ fe sh4[4];
    const fe* a = args.a[blockIdx.y];
    const fe* b = NC == 2 ? args.b[blockIdx.y] : nullptr;
    const fe* eq = args.eq[blockIdx.y];
    size_t h = args.half[blockIdx.y];
    fe e0 = Fr::zero(), e2 = Fr::zero();
    for (size_t i = (size_t)blockIdx.x * PT + threadIdx.x; i < h; i += (size_t)gridDim.x * PT) {
        fe q0 = fe_load(eq + i), q1 = fe_load(eq + i + h);
        fe p0 = sh_ab_sum<NC>(sh_load<NC>(a, b, i)), p1 = sh_ab_sum<NC>(sh_load<NC>(a, b, i + h));
        e0 = Fr::add(e0, Fr::mul(p0, q0));
        fe pb2 = Fr::sub(Fr::dbl(p1), p0), qb2 = Fr::sub(Fr::dbl(q1), q0);
        e2 = Fr::add(e2, Fr::mul(pb2, qb2));
    }
    e0 = fr_block_sum(e0, sh4);
    if (threadIdx.x == 0) fe_store(partial + (size_t)(2 * blockIdx.y) * gridDim.x + blockIdx.x, e0);
    e2 = fr_block_sum(e2, sh4);
    if (threadIdx.x == 0) fe_store(partial + (size_t)(2 * blockIdx.y + 1) * gridDim.x + blockIdx.x, e2);
}

// K11 leaf fingerprints (compute_leaves of the three memory-checking instances, e.g.
// co-jolt/src/jolt/vm/bytecode/worker.rs:57-100, read_write_memory/worker.rs:207-260):
//   leaf[i] = sum_k c_k * col_k[i]  (compact public columns, CompactPolynomial::field_mul)
//           + sum_j d_j * poly_j[i]  (shared or plain Fr polynomials, mul_public)
//           + constant               (e.g. -tau, or gamma^7 - tau for the write leaves)
// The public part enters a Rep3 leaf through add_public: party 0's a, party 1's b.  `cs` holds c_k * R (the
// Montgomery form of the Montgomery form), so that one product with the plain small integer gives c_k * v in
// Montgomery form.
struct SmallCol {
    const void* p;
    int kind;
};
static __device__ __forceinline__ uint64_t small_load(const SmallCol& c, size_t i) {
    switch (c.kind) {
        case COZK_SCALAR_U8: return reinterpret_cast<const uint8_t*>(c.p)[i];
        case COZK_SCALAR_U16: return reinterpret_cast<const uint16_t*>(c.p)[i];
        case COZK_SCALAR_U32: return reinterpret_cast<const uint32_t*>(c.p)[i];
        default: return reinterpret_cast<const uint64_t*>(c.p)[i];
    }
}
template <int NC>
__global__ void __launch_bounds__(PT) k_fingerprint_leaves(const SmallCol* __restrict__ cols, const fe* __restrict__ cs, int ks,
                                                        const fe* const* __restrict__ pa, const fe* const* __restrict__ pb,
                                                        const fe* __restrict__ ds, int kp, fe constant, int pub_a, int pub_b, fe* oa, fe* ob,
                                                        size_t n) {
    size_t i = (size_t)blockIdx.x * PT + threadIdx.x;
    if (i >= n) return;
    fe pub = constant;
    for (int k = 0; k < ks; k++) {
        uint64_t v = small_load(cols[k], i);
        fe x = Fr::zero();
        x.l[0] = (uint32_t)v;
        x.l[1] = (uint32_t)(v >> 32);
        pub = Fr::add(pub, Fr::mul(fe_load(cs + k), x));
    }
    Sh<NC> acc;
    for (int c = 0; c < NC; c++) acc.c[c] = Fr::zero();
    for (int j = 0; j < kp; j++) {
        fe d = fe_load(ds + j);
        if (pa[j]) acc.c[0] = Fr::add(acc.c[0], Fr::mul(fe_load(pa[j] + i), d));
        if (NC == 2 && pb[j]) acc.c[NC - 1] = Fr::add(acc.c[NC - 1], Fr::mul(fe_load(pb[j] + i), d));
    }
    if (pub_a) acc.c[0] = Fr::add(acc.c[0], pub);
    if (NC == 2 && pub_b) acc.c[NC - 1] = Fr::add(acc.c[NC - 1], pub);
    sh_store<NC>(oa, ob, i, acc);
}

// the usual shapes (<= 16 columns, <= 8 polynomials) travel by value in the kernel arguments: no staging, no drain
struct FingerprintArgs {
    SmallCol cols[16];
    fe cs[16];
    const fe* pa[8];
    const fe* pb[8];
    fe ds[8];
};
template <int NC>
__global__ void __launch_bounds__(PT) k_fingerprint_leaves_args(FingerprintArgs a, int ks, int kp, fe constant, int pub_a, int pub_b, fe* oa, fe* ob,
                                                             size_t n) {
    size_t i = (size_t)blockIdx.x * PT + threadIdx.x;
    if (i >= n) return;
    fe pub = constant;
    for (int k = 0; k < ks; k++) {
        uint64_t v = small_load(a.cols[k], i);
        fe x = Fr::zero();
        x.l[0] = (uint32_t)v;
        x.l[1] = (uint32_t)(v >> 32);
        pub = Fr::add(pub, Fr::mul(a.cs[k], x));
    }
    Sh<NC> acc;
    for (int c = 0; c < NC; c++) acc.c[c] = Fr::zero();
    for (int j = 0; j < kp; j++) {
        if (a.pa[j]) acc.c[0] = Fr::add(acc.c[0], Fr::mul(fe_load(a.pa[j] + i), a.ds[j]));
        if (NC == 2 && a.pb[j]) acc.c[NC - 1] = Fr::add(acc.c[NC - 1], Fr::mul(fe_load(a.pb[j] + i), a.ds[j]));
    }
    if (pub_a) acc.c[0] = Fr::add(acc.c[0], pub);
    if (NC == 2 && pub_b) acc.c[NC - 1] = Fr::add(acc.c[NC - 1], pub);
    sh_store<NC>(oa, ob, i, acc);
}

// quadratic opening-reduction round (compute_quadratic, opening_proof.rs:374-414): per opening
// eval_0 = sum_i poly[i]*eq[i], eval_2 = sum_i (2 poly[i+h] - poly[i]) * (2 eq[i+h] - eq[i]);
// partial[(2*y + e) * gridDim.x + x]; the (a+b)*TWO_INV conversion is folded into the finisher.
template <int NC>
__global__ void __launch_bounds__(PT) k_open_quadratic(const fe* const* __restrict__ pa, const fe* const* __restrict__ pb,
                                                    const fe* const* __restrict__ peq, const size_t* __restrict__ halves,
                                                    fe* __restrict__ partial) {
    __shared__ fe sh4[4];
    const fe* a = pa[blockIdx.y];
    const fe* b = NC == 2 ? pb[blockIdx.y] : nullptr;
    const fe* eq = peq[blockIdx.y];
    size_t h = halves[blockIdx.y];
    fe e0 = Fr::zero(), e2 = Fr::zero();
    for (size_t i = (size_t)blockIdx.x * PT + threadIdx.x; i < h; i += (size_t)gridDim.x * PT) {
        fe q0 = fe_load(eq + i), q1 = fe_load(eq + i + h);
        fe p0 = sh_ab_sum<NC>(sh_load<NC>(a, b, i)), p1 = sh_ab_sum<NC>(sh_load<NC>(a, b, i + h));
        e0 = Fr::add(e0, Fr::mul(p0, q0));
        fe pb2 = Fr::sub(Fr::dbl(p1), p0), qb2 = Fr::sub(Fr::dbl(q1), q0);
        e2 = Fr::add(e2, Fr::mul(pb2, qb2));
    }
    e0 = fr_block_sum(e0, sh4);
    if (threadIdx.x == 0) fe_store(partial + (size_t)(2 * blockIdx.y) * gridDim.x + blockIdx.x, e0);
    e2 = fr_block_sum(e2, sh4);
    if (threadIdx.x == 0) fe_store(partial + (size_t)(2 * blockIdx.y + 1) * gridDim.x + blockIdx.x, e2);
}

// PST13 `open` fold (pst13.rs:445-459): q[b] = r[2b+1] - r[2b]; r'[b] = r[2b](1-p) + r[2b+1] p
__global__ void __launch_bounds__(PT) k_pst_fold(const fe* __restrict__ r, fe* __restrict__ q, fe* __restrict__ rn, size_t half, fe p) {
    size_t i = (size_t)blockIdx.x * PT + threadIdx.x;
    if (i >= half) return;
    fe lo = fe_load(r + 2 * i), hi = fe_load(r + 2 * i + 1);
    fe d = Fr::sub(hi, lo);
    fe_store(q + i, d);
    fe_store(rn + i, Fr::add(lo, Fr::mul(d, p)));
}

// ------------------------------------------------------------------ kernels: interleaved GKR layer
// bind 4 -> 2 with a zero-padded ragged tail (dense_interleaved_poly.rs:155-195)
template <int NC>
__global__ void __launch_bounds__(PT) k_layer_bind(const fe* __restrict__ ia, const fe* __restrict__ ib, fe* oa, fe* ob,
                                                size_t len, fe r) {
    size_t c = (size_t)blockIdx.x * PT + threadIdx.x;
    size_t nch = (len + 3) / 4;
    if (c >= nch) return;
    Sh<NC> u0 = sh_load_or_zero<NC>(ia, ib, 4 * c, len), u1 = sh_load_or_zero<NC>(ia, ib, 4 * c + 1, len);
    Sh<NC> u2 = sh_load_or_zero<NC>(ia, ib, 4 * c + 2, len), u3 = sh_load_or_zero<NC>(ia, ib, 4 * c + 3, len);
    sh_store<NC>(oa, ob, 2 * c, sh_lerp<NC>(u0, u2, r));
    sh_store<NC>(oa, ob, 2 * c + 1, sh_lerp<NC>(u1, u3, r));
}

// eq evaluations at 0, 2, 3 of the linear factor through (e0, e1)
static __device__ __forceinline__ void eq3(const fe& e0, const fe& e1, fe out[3]) {
    fe m = Fr::sub(e1, e0);
    out[0] = e0;
    out[1] = Fr::add(e1, m);
    out[2] = Fr::add(out[1], m);
}

// cubic round evaluations g(0), g(2), g(3) of sum eq * L * R  (compute_cubic,
// dense_interleaved_poly.rs:210-356).  NESTED = 0: E1 fully bound, eq pairs come from E2 (:218-268);
// NESTED = 1: Dao-Thaler split, chunk k belongs to x2 = k / (E1_len/2), x1 = k % (E1_len/2) (:269-356).
template <int NC, int NESTED>
__global__ void __launch_bounds__(PT) k_layer_cubic(const fe* __restrict__ a, const fe* __restrict__ b, size_t len,
                                                 const fe* __restrict__ E1, size_t E1_half, const fe* __restrict__ E2,
                                                 size_t E2_len, fe* __restrict__ partial) {
    __shared__ fe sh4[4];
    size_t nch = (len + 3) / 4;
    size_t limit = NESTED ? E1_half * E2_len : E2_len / 2;
    if (nch > limit) nch = limit;  // zip() stops at the shorter side
    fe s0 = Fr::zero(), s2 = Fr::zero(), s3 = Fr::zero();
    for (size_t c = (size_t)blockIdx.x * PT + threadIdx.x; c < nch; c += (size_t)gridDim.x * PT) {
        fe e[3];
        fe scale;
        if (NESTED) {
            size_t x2 = c >> (__ffsll((long long)E1_half) - 1), x1 = c & (E1_half - 1);  // E1_half is a power of two
            eq3(fe_load(E1 + 2 * x1), fe_load(E1 + 2 * x1 + 1), e);
            scale = fe_load(E2 + x2);
        } else {
            eq3(fe_load(E2 + 2 * c), fe_load(E2 + 2 * c + 1), e);
        }
        Sh<NC> l0 = sh_load_or_zero<NC>(a, b, 4 * c, len), r0 = sh_load_or_zero<NC>(a, b, 4 * c + 1, len);
        Sh<NC> l1 = sh_load_or_zero<NC>(a, b, 4 * c + 2, len), r1 = sh_load_or_zero<NC>(a, b, 4 * c + 3, len);
        Sh<NC> ml = sh_sub<NC>(l1, l0), mr = sh_sub<NC>(r1, r0);
        Sh<NC> l2 = sh_add<NC>(l1, ml), r2 = sh_add<NC>(r1, mr);
        Sh<NC> l3 = sh_add<NC>(l2, ml), r3 = sh_add<NC>(r2, mr);
        fe t0 = Fr::mul(sh_local_mul<NC>(l0, r0), e[0]);
        fe t2 = Fr::mul(sh_local_mul<NC>(l2, r2), e[1]);
        fe t3 = Fr::mul(sh_local_mul<NC>(l3, r3), e[2]);
        if (NESTED) {
            t0 = Fr::mul(t0, scale);
            t2 = Fr::mul(t2, scale);
            t3 = Fr::mul(t3, scale);
        }
        s0 = Fr::add(s0, t0);
        s2 = Fr::add(s2, t2);
        s3 = Fr::add(s3, t3);
    }
    s0 = fr_block_sum(s0, sh4);
    if (threadIdx.x == 0) fe_store(partial + blockIdx.x, s0);
    s2 = fr_block_sum(s2, sh4);
    if (threadIdx.x == 0) fe_store(partial + gridDim.x + blockIdx.x, s2);
    s3 = fr_block_sum(s3, sh4);
    if (threadIdx.x == 0) fe_store(partial + 2 * gridDim.x + blockIdx.x, s3);
}

// bind + the next round's cubic sums in ONE pass over a large layer: a lane reads 8 unbound elements, writes the 4
// bound ones and adds their chunk's three terms (eq tables already folded by the caller).  The separate kernels
// read the layer twice and write it once per round (80 B per element, plain); this reads once and writes once (48).
template <int NC, int NESTED>
__global__ void __launch_bounds__(PT) k_layer_bind_cubic(const fe* __restrict__ ia, const fe* __restrict__ ib, fe* oa, fe* ob, size_t len_in, fe r,
                                                      const fe* __restrict__ E1, size_t E1_half, const fe* __restrict__ E2, size_t E2_len,
                                                      fe* __restrict__ partial) {
    __shared__ fe sh4[4];
    const size_t nch_in = (len_in + 3) / 4;      // input chunks of 4 -> 2 bound elements each
    const size_t len_out = 2 * nch_in;
    const size_t nch_out = (len_out + 3) / 4;    // output chunks of 4 bound elements = 2 input chunks
    size_t limit = NESTED ? E1_half * E2_len : E2_len / 2;
    fe s0 = Fr::zero(), s2 = Fr::zero(), s3 = Fr::zero();
    for (size_t c = (size_t)blockIdx.x * PT + threadIdx.x; c < nch_out; c += (size_t)gridDim.x * PT) {
        Sh<NC> v[4];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            size_t ci = 2 * c + h;
            if (ci < nch_in) {
                Sh<NC> u0 = sh_load_or_zero<NC>(ia, ib, 4 * ci, len_in), u1 = sh_load_or_zero<NC>(ia, ib, 4 * ci + 1, len_in);
                Sh<NC> u2 = sh_load_or_zero<NC>(ia, ib, 4 * ci + 2, len_in), u3 = sh_load_or_zero<NC>(ia, ib, 4 * ci + 3, len_in);
                v[2 * h] = sh_lerp<NC>(u0, u2, r);
                v[2 * h + 1] = sh_lerp<NC>(u1, u3, r);
                sh_store<NC>(oa, ob, 2 * ci, v[2 * h]);
                sh_store<NC>(oa, ob, 2 * ci + 1, v[2 * h + 1]);
            } else {
                for (int k = 0; k < NC; k++) v[2 * h].c[k] = v[2 * h + 1].c[k] = Fr::zero();
            }
        }
        if (c < limit) {  // zip() stops at the shorter side (k_layer_cubic)
            fe e[3];
            fe scale;
            if (NESTED) {
                size_t x2 = c >> (__ffsll((long long)E1_half) - 1), x1 = c & (E1_half - 1);  // E1_half is a power of two
                eq3(fe_load(E1 + 2 * x1), fe_load(E1 + 2 * x1 + 1), e);
                scale = fe_load(E2 + x2);
            } else {
                eq3(fe_load(E2 + 2 * c), fe_load(E2 + 2 * c + 1), e);
            }
            Sh<NC> l0 = v[0], r0 = v[1], l1 = v[2], r1 = v[3];
            Sh<NC> ml = sh_sub<NC>(l1, l0), mr = sh_sub<NC>(r1, r0);
            Sh<NC> l2 = sh_add<NC>(l1, ml), r2 = sh_add<NC>(r1, mr);
            Sh<NC> l3 = sh_add<NC>(l2, ml), r3 = sh_add<NC>(r2, mr);
            fe t0 = Fr::mul(sh_local_mul<NC>(l0, r0), e[0]);
            fe t2 = Fr::mul(sh_local_mul<NC>(l2, r2), e[1]);
            fe t3 = Fr::mul(sh_local_mul<NC>(l3, r3), e[2]);
            if (NESTED) {
                t0 = Fr::mul(t0, scale);
                t2 = Fr::mul(t2, scale);
                t3 = Fr::mul(t3, scale);
            }
            s0 = Fr::add(s0, t0);
            s2 = Fr::add(s2, t2);
            s3 = Fr::add(s3, t3);
        }
    }
    s0 = fr_block_sum(s0, sh4);
    if (threadIdx.x == 0) fe_store(partial + blockIdx.x, s0);
    s2 = fr_block_sum(s2, sh4);
    if (threadIdx.x == 0) fe_store(partial + gridDim.x + blockIdx.x, s2);
    s3 = fr_block_sum(s3, sh4);
    if (threadIdx.x == 0) fe_store(partial + 2 * gridDim.x + blockIdx.x, s3);
}

// ---- the layer kernels on the 9 x 29 multiplier (round 3, fr9.hip.hpp): same thread mapping and tails as k_layer_cubic /
// k_layer_bind_cubic, every product on the fused 29-bit chains, additions lazy, one lambda-removing product per lane at the end.
// terms of one output chunk: s_k += (l_k x r_k) e_k at X = 0, 2, 3.  NESTED = 1: the E2 factor is folded into the eq pair first (two
// products per chunk).  NESTED = 2 (E1 tables of >= 512 pairs; the launcher walks the chunks block-contiguously): the inner sums of
// the split-eq form -- the terms of the chunks that share one E2 entry are summed with the raw E1 pair (lambda^2) in `grp` and
// multiplied by E2[x2] once per group (three products per E1_half / 256 chunks instead of two per chunk), as the reference's
// nested loops do (dense_interleaved_poly.rs:230-300).
struct Layer9Group {
    f9 g0, g2, g3;
    size_t x2;
};
static __device__ __forceinline__ void layer9_group_flush(Layer9Group& grp, const fe* __restrict__ E2, f9& s0, f9& s2, f9& s3) {
    if (grp.x2 == (size_t)-1) return;
    const f9 sc = f9_from_fe(fe_load(E2 + grp.x2));
    s0 = f9_norm(fr9_add(s0, fr9_mul(grp.g0, sc)));
    s2 = f9_norm(fr9_add(s2, fr9_mul(grp.g2, sc)));
    s3 = f9_norm(fr9_add(s3, fr9_mul(grp.g3, sc)));
}
template <int NC, int NESTED>
static __device__ __forceinline__ void layer9_terms(const Sh9<NC>& l0, const Sh9<NC>& r0, const Sh9<NC>& l1, const Sh9<NC>& r1, const fe* __restrict__ E1,
                                                    size_t E1_half, int e1_shift, const fe* __restrict__ E2, size_t c, f9& s0, f9& s2, f9& s3,
                                                    Layer9Group& grp) {
    f9 e0, e1;
    if (NESTED == 2) {
        const size_t x2 = c >> e1_shift, x1 = c & (E1_half - 1);
        if (x2 != grp.x2) {
            layer9_group_flush(grp, E2, s0, s2, s3);
            grp.g0 = grp.g2 = grp.g3 = fr9_zero();
            grp.x2 = x2;
        }
        e0 = f9_from_fe(fe_load(E1 + 2 * x1));
        e1 = f9_from_fe(fe_load(E1 + 2 * x1 + 1));
    } else if (NESTED) {
        const size_t x2 = c >> e1_shift, x1 = c & (E1_half - 1);
        const f9 sc = f9_from_fe(fe_load(E2 + x2));
        e0 = fr9_mul(f9_from_fe(fe_load(E1 + 2 * x1)), sc);
        e1 = fr9_mul(f9_from_fe(fe_load(E1 + 2 * x1 + 1)), sc);
    } else {
        e0 = f9_from_fe(fe_load(E2 + 2 * c));
        e1 = f9_from_fe(fe_load(E2 + 2 * c + 1));
    }
    const f9 me = f9_norm(f9_sub(e1, FR9_C2, e0));  // e0, e1 < 1.01 r normalised: me < 3.01 r
    const f9 e2 = fr9_add(e1, me), e3 = fr9_add(e2, me);  // limbs < 2^30, 1.5 * 2^30
    const Sh9<NC> ml = sh9_diff<NC>(l1, l0), mr = sh9_diff<NC>(r1, r0);
    const Sh9<NC> l2 = sh9_add_norm<NC>(l1, ml), r2 = sh9_add_norm<NC>(r1, mr);
    const Sh9<NC> l3 = sh9_add_norm<NC>(l2, ml), r3 = sh9_add_norm<NC>(r2, mr);
    f9& a0 = NESTED == 2 ? grp.g0 : s0;
    f9& a2 = NESTED == 2 ? grp.g2 : s2;
    f9& a3 = NESTED == 2 ? grp.g3 : s3;
    a0 = f9_norm(fr9_add(a0, fr9_mul(e0, sh9_local_mul<NC>(l0, r0))));
    a2 = f9_norm(fr9_add(a2, fr9_mul(e2, sh9_local_mul<NC>(l2, r2))));
    a3 = f9_norm(fr9_add(a3, fr9_mul(e3, sh9_local_mul<NC>(l3, r3))));
}
// the chunk walk of a workgroup: grid-stride, or (NESTED = 2) one contiguous range per workgroup so that a lane stays inside an E2
// group for E1_half / 256 iterations; `first` is the wave's first chunk of the first iteration
static __device__ __forceinline__ void layer9_walk(int nested, size_t nch, size_t& first, size_t& end, size_t& step) {
    if (nested == 2) {
        const size_t per = ((nch + gridDim.x - 1) / gridDim.x + PT - 1) / PT * PT;
        first = (size_t)blockIdx.x * per + (threadIdx.x & ~63u);
        end = (size_t)(blockIdx.x + 1) * per < nch ? (size_t)(blockIdx.x + 1) * per : nch;
        step = PT;
    } else {
        first = (size_t)blockIdx.x * PT + (threadIdx.x & ~63u);
        end = nch;
        step = (size_t)gridDim.x * PT;
    }
}
// lane sums -> canonical R-form block sums in partial[k * gridDim.x + blockIdx.x]
template <int NESTED>
static __device__ __forceinline__ void layer9_finish(const f9& s0, const f9& s2, const f9& s3, fe* __restrict__ partial, fe* sh4) {
    const f9 K = f9_const(NESTED ? FR9_K3 : FR9_K2);
    fe v = fr_block_sum(fr9_to_canonical(fr9_mul(s0, K)), sh4);
    if (threadIdx.x == 0) fe_store(partial + blockIdx.x, v);
    v = fr_block_sum(fr9_to_canonical(fr9_mul(s2, K)), sh4);
    if (threadIdx.x == 0) fe_store(partial + gridDim.x + blockIdx.x, v);
    v = fr_block_sum(fr9_to_canonical(fr9_mul(s3, K)), sh4);
    if (threadIdx.x == 0) fe_store(partial + 2 * gridDim.x + blockIdx.x, v);
}


// ---- wave-private LDS staging of layer data (round 3).  A lane owns 128 consecutive bytes (4 field elements) of a row; read or
// written straight from the lane, a wave's dwordx4 instruction touches 64 different 128-byte lines 16 bytes at a time and every line
// is re-referenced by eight instructions (nontemporal accesses, which forbid that re-use, slowed the unstaged round 1.5-2 x).  Staged,
// instruction k moves rows 8k .. 8k+7 as eight WHOLE lines (lane l: row 8k + l / 8, 16-byte piece l % 8) and the lanes trade pieces
// through a wave-private LDS window of 64 rows x 144 bytes (128 + 16: a b128 access of 16 lanes then covers 16 distinct bank quads).
// A wave's LDS instructions execute in order; the fences keep the compiler from reordering them across the exchange.
static constexpr int ST_ROW = 144;
static constexpr int ST_WAVE_BYTES = 64 * ST_ROW;
static __device__ __forceinline__ void stage_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// out[0..3] <- the 4 elements at g + lane * row_stride (bytes); the wave's 64 rows are all in range
static __device__ __forceinline__ void stage_in_rows(const char* __restrict__ g, size_t row_stride, char* st, fe out[4]) {
    const int lane = threadIdx.x & 63, piece = lane & 7, sub = lane >> 3;
    uint4 t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) t[k] = *reinterpret_cast<const uint4*>(g + (size_t)(8 * k + sub) * row_stride + 16 * piece);
#pragma unroll
    for (int k = 0; k < 8; k++) *reinterpret_cast<uint4*>(st + (8 * k + sub) * ST_ROW + 16 * piece) = t[k];
    stage_fence();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint4 lo = *reinterpret_cast<const uint4*>(st + lane * ST_ROW + 32 * j);
        const uint4 hi = *reinterpret_cast<const uint4*>(st + lane * ST_ROW + 32 * j + 16);
        out[j].l[0] = lo.x; out[j].l[1] = lo.y; out[j].l[2] = lo.z; out[j].l[3] = lo.w;
        out[j].l[4] = hi.x; out[j].l[5] = hi.y; out[j].l[6] = hi.z; out[j].l[7] = hi.w;
    }
    stage_fence();  // the window is free again once every lane has its row
}
// the 4 elements of this lane -> g + lane * row_stride (bytes)
static __device__ __forceinline__ void stage_out_rows(char* __restrict__ g, char* st, const fe in[4], size_t row_stride = 128) {
    const int lane = threadIdx.x & 63, piece = lane & 7, sub = lane >> 3;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        *reinterpret_cast<uint4*>(st + lane * ST_ROW + 32 * j) = make_uint4(in[j].l[0], in[j].l[1], in[j].l[2], in[j].l[3]);
        *reinterpret_cast<uint4*>(st + lane * ST_ROW + 32 * j + 16) = make_uint4(in[j].l[4], in[j].l[5], in[j].l[6], in[j].l[7]);
    }
    stage_fence();
#pragma unroll
    for (int k = 0; k < 8; k++)
        *reinterpret_cast<uint4*>(g + (size_t)(8 * k + sub) * row_stride + 16 * piece) = *reinterpret_cast<const uint4*>(st + (8 * k + sub) * ST_ROW + 16 * piece);
    stage_fence();
}

template <int NC, int NESTED>
__global__ void __launch_bounds__(PT) k_layer_cubic9(const fe* __restrict__ a, const fe* __restrict__ b, size_t len, const fe* __restrict__ E1, size_t E1_half,
                                                  const fe* __restrict__ E2, size_t E2_len, fe* __restrict__ partial) {
    __shared__ fe sh4[4];
    __shared__ __attribute__((aligned(16))) char stage[PT / 64][ST_WAVE_BYTES];
    char* st = stage[threadIdx.x >> 6];
    size_t nch = (len + 3) / 4;
    const size_t limit = NESTED ? E1_half * E2_len : E2_len / 2;
    if (nch > limit) nch = limit;  // zip() stops at the shorter side
    const int e1_shift = NESTED ? __ffsll((long long)E1_half) - 1 : 0;
    f9 s0 = fr9_zero(), s2 = fr9_zero(), s3 = fr9_zero();
    Layer9Group grp;
    grp.x2 = (size_t)-1;
    size_t c_first, c_end, c_step;
    layer9_walk(NESTED, nch, c_first, c_end, c_step);
    // the trip count is per WAVE (c0 = the wave's first chunk): a wave whose 64 chunks are all inside the layer takes the staged path
    for (size_t c0 = c_first; c0 < c_end; c0 += c_step) {
        const size_t c = c0 + (threadIdx.x & 63);
        if (c0 + 64 <= nch && 4 * (c0 + 64) <= len) {
            Sh9<NC> q[4];
            const fe* comp[2] = {a, b};
#pragma unroll
            for (int k = 0; k < NC; k++) {
                fe raw[4];
                stage_in_rows(reinterpret_cast<const char*>(comp[k] + 4 * c0), 128, st, raw);
#pragma unroll
                for (int j = 0; j < 4; j++) q[j].c[k] = f9_from_fe(raw[j]);
            }
            layer9_terms<NC, NESTED>(q[0], q[1], q[2], q[3], E1, E1_half, e1_shift, E2, c, s0, s2, s3, grp);
        } else if (c < nch) {
            const Sh9<NC> l0 = sh9_load_or_zero<NC>(a, b, 4 * c, len), r0 = sh9_load_or_zero<NC>(a, b, 4 * c + 1, len);
            const Sh9<NC> l1 = sh9_load_or_zero<NC>(a, b, 4 * c + 2, len), r1 = sh9_load_or_zero<NC>(a, b, 4 * c + 3, len);
            layer9_terms<NC, NESTED>(l0, r0, l1, r1, E1, E1_half, e1_shift, E2, c, s0, s2, s3, grp);
        }
    }
    if (NESTED == 2) layer9_group_flush(grp, E2, s0, s2, s3);
    layer9_finish<NESTED>(s0, s2, s3, partial, sh4);
}

template <int NC, int NESTED>
__global__ void __launch_bounds__(PT) k_layer_bind_cubic9(const fe* __restrict__ ia, const fe* __restrict__ ib, fe* oa, fe* ob, size_t len_in, fe r5,
                                                       const fe* __restrict__ E1, size_t E1_half, const fe* __restrict__ E2, size_t E2_len,
                                                       fe* __restrict__ partial) {
    __shared__ fe sh4[4];
    const size_t nch_in = (len_in + 3) / 4;
    const size_t len_out = 2 * nch_in;
    const size_t nch_out = (len_out + 3) / 4;
    const size_t limit = NESTED ? E1_half * E2_len : E2_len / 2;
    const int e1_shift = NESTED ? __ffsll((long long)E1_half) - 1 : 0;
    const f9 r9 = f9_from_fe(r5);
    f9 s0 = fr9_zero(), s2 = fr9_zero(), s3 = fr9_zero();
    __shared__ __attribute__((aligned(16))) char stage[PT / 64][ST_WAVE_BYTES];
    char* st = stage[threadIdx.x >> 6];
    Layer9Group grp;
    grp.x2 = (size_t)-1;
    size_t c_first, c_end, c_step;
    layer9_walk(NESTED, nch_out, c_first, c_end, c_step);
    for (size_t c0 = c_first; c0 < c_end; c0 += c_step) {
        const size_t c = c0 + (threadIdx.x & 63);
        Sh9<NC> v[4];
        if (c0 + 64 <= nch_out && 8 * (c0 + 64) <= len_in) {
            // staged: the wave's 64 x 8 inputs come in as whole lines (two halves of 4 elements per lane and component), its
            // 64 x 4 outputs leave as whole lines
            const fe* icomp[2] = {ia, ib};
            fe* ocomp[2] = {oa, ob};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                Sh9<NC> u[4];
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    fe raw[4];
                    stage_in_rows(reinterpret_cast<const char*>(icomp[k] + 8 * c0 + 4 * h), 256, st, raw);
#pragma unroll
                    for (int j = 0; j < 4; j++) u[j].c[k] = f9_from_fe(raw[j]);
                }
                v[2 * h] = sh9_lerp<NC>(u[0], u[2], r9);
                v[2 * h + 1] = sh9_lerp<NC>(u[1], u[3], r9);
            }
#pragma unroll
            for (int k = 0; k < NC; k++) {
                fe outv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) outv[j] = fr9_to_canonical(v[j].c[k]);
                stage_out_rows(reinterpret_cast<char*>(ocomp[k] + 4 * c0), st, outv);
            }
            if (c < limit) layer9_terms<NC, NESTED>(v[0], v[1], v[2], v[3], E1, E1_half, e1_shift, E2, c, s0, s2, s3, grp);
            continue;
        }
        if (c >= nch_out) continue;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const size_t ci = 2 * c + h;
            if (ci < nch_in) {
                const Sh9<NC> u0 = sh9_load_or_zero<NC>(ia, ib, 4 * ci, len_in), u1 = sh9_load_or_zero<NC>(ia, ib, 4 * ci + 1, len_in);
                const Sh9<NC> u2 = sh9_load_or_zero<NC>(ia, ib, 4 * ci + 2, len_in), u3 = sh9_load_or_zero<NC>(ia, ib, 4 * ci + 3, len_in);
                v[2 * h] = sh9_lerp<NC>(u0, u2, r9);
                v[2 * h + 1] = sh9_lerp<NC>(u1, u3, r9);
                sh9_store<NC>(oa, ob, 2 * ci, v[2 * h]);
                sh9_store<NC>(oa, ob, 2 * ci + 1, v[2 * h + 1]);
            } else {
                for (int k = 0; k < NC; k++) v[2 * h].c[k] = v[2 * h + 1].c[k] = fr9_zero();
            }
        }
        if (c < limit) layer9_terms<NC, NESTED>(v[0], v[1], v[2], v[3], E1, E1_half, e1_shift, E2, c, s0, s2, s3, grp);
    }
    if (NESTED == 2) layer9_group_flush(grp, E2, s0, s2, s3);
    layer9_finish<NESTED>(s0, s2, s3, partial, sh4);
}

// One whole sumcheck round of a SMALL layer in a single one-workgroup launch: bind the layer and the split-eq
// tables with the previous challenge (k_layer_bind + SplitEqPolynomial::bind), then the cubic sums of the new
// round (k_layer_cubic) and their block reduction straight into the pinned result slot.  The GKR proof is ~270
// rounds deep and ~200 of them touch <= 8192 elements: as five separate launches each costs more in launch
// gaps than in work.
static constexpr int RT = 1024;
static constexpr size_t ROUND_SMALL_MAX = 2048;  // layer length before the bind (measured: a single workgroup loses above 2^11)
template <int NC>
__global__ void __launch_bounds__(RT) k_layer_round_small(const fe* __restrict__ ia, const fe* __restrict__ ib, fe* oa, fe* ob, size_t len_in,
                                                       int do_bind, fe r, const fe* fold_in, fe* fold_out, size_t fold_n, fe* scale_vec,
                                                       size_t scale_n, const fe* ca, const fe* cb, size_t len, const fe* E1, size_t E1_half,
                                                       const fe* E2, size_t E2_len, int nested, fe* __restrict__ res) {
    __shared__ fe sh16[16];
    if (do_bind) {
        size_t nch_in = (len_in + 3) / 4;
        for (size_t c = threadIdx.x; c < nch_in; c += RT) {
            Sh<NC> u0 = sh_load_or_zero<NC>(ia, ib, 4 * c, len_in), u1 = sh_load_or_zero<NC>(ia, ib, 4 * c + 1, len_in);
            Sh<NC> u2 = sh_load_or_zero<NC>(ia, ib, 4 * c + 2, len_in), u3 = sh_load_or_zero<NC>(ia, ib, 4 * c + 3, len_in);
            sh_store<NC>(oa, ob, 2 * c, sh_lerp<NC>(u0, u2, r));
            sh_store<NC>(oa, ob, 2 * c + 1, sh_lerp<NC>(u1, u3, r));
        }
        for (size_t i = threadIdx.x; i < fold_n; i += RT) {
            fe lo = fe_load(fold_in + 2 * i), hi = fe_load(fold_in + 2 * i + 1);
            fe_store(fold_out + i, Fr::add(lo, Fr::mul(Fr::sub(hi, lo), r)));
        }
        __syncthreads();
        if (scale_n) {  // E1 just collapsed to one value: it multiplies E2 from now on
            fe s = fe_load(fold_out);
            for (size_t i = threadIdx.x; i < scale_n; i += RT) fe_store(scale_vec + i, Fr::mul(fe_load(scale_vec + i), s));
            __syncthreads();
        }
    }
    size_t nch = (len + 3) / 4;
    size_t limit = nested ? E1_half * E2_len : E2_len / 2;
    if (nch > limit) nch = limit;
    fe s0 = Fr::zero(), s2 = Fr::zero(), s3 = Fr::zero();
    for (size_t c = threadIdx.x; c < nch; c += RT) {
        fe e[3];
        fe scale = Fr::zero();
        if (nested) {
            size_t x2 = c >> (__ffsll((long long)E1_half) - 1), x1 = c & (E1_half - 1);  // E1_half is a power of two
            eq3(fe_load(E1 + 2 * x1), fe_load(E1 + 2 * x1 + 1), e);
            scale = fe_load(E2 + x2);
        } else {
            eq3(fe_load(E2 + 2 * c), fe_load(E2 + 2 * c + 1), e);
        }
        Sh<NC> l0 = sh_load_or_zero<NC>(ca, cb, 4 * c, len), r0 = sh_load_or_zero<NC>(ca, cb, 4 * c + 1, len);
        Sh<NC> l1 = sh_load_or_zero<NC>(ca, cb, 4 * c + 2, len), r1 = sh_load_or_zero<NC>(ca, cb, 4 * c + 3, len);
        Sh<NC> ml = sh_sub<NC>(l1, l0), mr = sh_sub<NC>(r1, r0);
        Sh<NC> l2 = sh_add<NC>(l1, ml), r2 = sh_add<NC>(r1, mr);
        Sh<NC> l3 = sh_add<NC>(l2, ml), r3 = sh_add<NC>(r2, mr);
        fe t0 = Fr::mul(sh_local_mul<NC>(l0, r0), e[0]);
        fe t2 = Fr::mul(sh_local_mul<NC>(l2, r2), e[1]);
        fe t3 = Fr::mul(sh_local_mul<NC>(l3, r3), e[2]);
        if (nested) {
            t0 = Fr::mul(t0, scale);
            t2 = Fr::mul(t2, scale);
            t3 = Fr::mul(t3, scale);
        }
        s0 = Fr::add(s0, t0);
        s2 = Fr::add(s2, t2);
        s3 = Fr::add(s3, t3);
    }
    s0 = fr_block_sum(s0, sh16);
    if (threadIdx.x == 0) fe_store(res, s0);
    s2 = fr_block_sum(s2, sh16);
    if (threadIdx.x == 0) fe_store(res + 1, s2);
    s3 = fr_block_sum(s3, sh16);
    if (threadIdx.x == 0) fe_store(res + 2, s3);
}

// ------------------------------------------------------------------ persistent round kernel (host mailbox)
// A launch-and-drain costs ~38 us on this platform however small the kernel (measured per round, 2^3 .. 2^19
// elements alike), and a grand product has ~170 rounds on layers of <= 2048 elements.  For those tails ONE
// single-workgroup kernel stays resident for all remaining rounds of the layer's sumcheck and talks to the host
// through a mailbox in fine-grained pinned memory: it publishes the three cubic sums of a round, spins (wave 0
// only; the other waves sit at the barrier) until the host has posted the challenge, binds layer + eq tables,
// and goes on -- a PCIe round trip (~3 us) per round instead of a launch.  After the last challenge it binds once
// more and publishes the final claims.  Every wait is bounded by the wall clock (resident_timeout_s() seconds of the
// 100 MHz counter): a host that never answers makes the kernel raise `status` and return, so the grid always drains.
struct alignas(64) RoundMailbox {
    // host -> device, ONE 64-byte line read by one load instruction (four lanes x 16 B = one PCIe read instead of a poll of the
    // sequence number followed by a read of the challenge): pieces 0..2 = three limbs of the challenge + the challenge's number
    // as a tag in their last word (a 16-byte piece is read whole, the host writes its tag last, so a piece whose tag matches
    // carries its payload), piece 3 = the abort word
    uint32_t cmd[16];
    uint32_t pad0[16];
    uint32_t res_seq;  // device -> host: result number k is ready
    uint32_t status;   // device -> host: 0 ok, 1 timed out waiting for the host, 2 aborted
    uint32_t pad2[14];
    fe res[4];         // device -> host: cubic sums g(0), g(2), g(3); final: left.a, left.b, right.a, right.b
    unsigned long long dbg[8];  // device -> host: 100 MHz ticks spent waiting / binding / in the cubic sums / publishing
};
static constexpr long long MB_TICKS_PER_S = 100000000ll;  // wall_clock64() runs at 100 MHz
// watchdog of the resident kernel, seconds (COZK_RESIDENT_TIMEOUT_S, default 10, 1..600): raise it when the round
// callback can legitimately take longer (a coordinator across a slow link)
static int resident_timeout_s() {
    const char* e = getenv("COZK_RESIDENT_TIMEOUT_S");  // read per call (a handful of times per layer): hosts and tests may change it
    int t = e ? atoi(e) : 10;
    return t < 1 ? 1 : (t > 600 ? 600 : t);
}
static constexpr size_t ROUND_PERSIST_MAX = 2048;

template <int NC, int TRACE>
__global__ void __launch_bounds__(RT) k_layer_rounds_persistent(fe* la0, fe* lb0, fe* la1, fe* lb1, int lcur, size_t len, fe* e1_0, fe* e1_1,
                                                             int c1, size_t E1_len, fe* e2_0, fe* e2_1, int c2, size_t E2_len, int nrounds,
                                                             int bind_first, fe r_first, RoundMailbox* mb, long long timeout_ticks) {
    __shared__ fe sh16[16];
    __shared__ fe sh_r;
    __shared__ int sh_ok;
    fe* la[2] = {la0, la1};
    fe* lb[2] = {lb0, lb1};
    fe* e1[2] = {e1_0, e1_1};
    fe* e2[2] = {e2_0, e2_1};
    uint32_t seq = 0;
    long long tk0 = TRACE ? wall_clock64() : 0, tk_wait = 0, tk_bind = 0, tk_cubic = 0, tk_pub = 0;  // TRACE: per-phase device times (COZK_TRACE_ROUNDS)
    // rounds 0 .. nrounds-1 publish cubic sums; "round" nrounds only binds and publishes the final claims
    for (int round = 0; round <= nrounds; round++) {
        bool do_bind = round > 0 || bind_first;
        fe r = r_first;
        if (TRACE) tk0 = wall_clock64();
        if (round > 0) {
            if (threadIdx.x < 64) {  // wave 0 polls: lanes 0..3 read one 16-byte piece of the command line each
                typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                const int lane = threadIdx.x;
                const long long t0 = wall_clock64();
                int ok = 1;
                v4u v = {0u, 0u, 0u, 0u};
                for (;;) {
                    if (lane < 4) v = *reinterpret_cast<volatile v4u*>(&mb->cmd[4 * lane]);
                    const unsigned long long ready = __ballot(lane < 3 && v.w == (uint32_t)round);
                    const unsigned long long aborted = __ballot(lane == 3 && v.x != 0u);
                    if ((ready & 7ull) == 7ull) break;
                    if (aborted) {
                        ok = 0;
                        if (lane == 0) __hip_atomic_store(&mb->status, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    if (__shfl((int)(wall_clock64() - t0 > timeout_ticks), 0)) {  // lane 0's clock decides for the wave
                        ok = 0;
                        if (lane == 0) __hip_atomic_store(&mb->status, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                fe rr;
                rr.l[0] = __shfl(v.x, 0); rr.l[1] = __shfl(v.y, 0); rr.l[2] = __shfl(v.z, 0);
                rr.l[3] = __shfl(v.x, 1); rr.l[4] = __shfl(v.y, 1); rr.l[5] = __shfl(v.z, 1);
                rr.l[6] = __shfl(v.x, 2); rr.l[7] = __shfl(v.y, 2);
                if (lane == 0) {
                    if (ok) sh_r = rr;
                    sh_ok = ok;
                }
            }
            __syncthreads();
            if (!sh_ok) return;  // uniform: every wave reads the same shared word after the barrier
            r = sh_r;
        }
        if (TRACE) {
            long long t = wall_clock64();
            tk_wait += t - tk0;
            tk0 = t;
        }
        if (do_bind) {
            const fe *ia = la[lcur], *ib = lb[lcur];
            fe *oa = la[1 - lcur], *ob = lb[1 - lcur];
            size_t nch_in = (len + 3) / 4;
            for (size_t c = threadIdx.x; c < nch_in; c += RT) {
                Sh<NC> u0 = sh_load_or_zero<NC>(ia, ib, 4 * c, len), u1 = sh_load_or_zero<NC>(ia, ib, 4 * c + 1, len);
                Sh<NC> u2 = sh_load_or_zero<NC>(ia, ib, 4 * c + 2, len), u3 = sh_load_or_zero<NC>(ia, ib, 4 * c + 3, len);
                sh_store<NC>(oa, ob, 2 * c, sh_lerp<NC>(u0, u2, r));
                sh_store<NC>(oa, ob, 2 * c + 1, sh_lerp<NC>(u1, u3, r));
            }
            lcur = 1 - lcur;
            len = 2 * nch_in;
            // SplitEqPolynomial::bind
            if (E1_len == 1) {
                size_t n = E2_len / 2;
                const fe* in = e2[c2];
                fe* out = e2[1 - c2];
                // from the LAST thread down: the fold runs on other waves than the layer's bind instead of behind it in the same
                // lanes (a tiny round's bind phase was two dependent load -> product -> store chains back to back: 4.9 us)
                for (size_t i = RT - 1 - threadIdx.x; i < n; i += RT) {
                    fe lo = fe_load(in + 2 * i), hi = fe_load(in + 2 * i + 1);
                    fe_store(out + i, Fr::add(lo, Fr::mul(Fr::sub(hi, lo), r)));
                }
                c2 = 1 - c2;
                E2_len = n;
                __syncthreads();
            } else {
                size_t n = E1_len / 2;
                const fe* in = e1[c1];
                fe* out = e1[1 - c1];
                for (size_t i = RT - 1 - threadIdx.x; i < n; i += RT) {
                    fe lo = fe_load(in + 2 * i), hi = fe_load(in + 2 * i + 1);
                    fe_store(out + i, Fr::add(lo, Fr::mul(Fr::sub(hi, lo), r)));
                }
                c1 = 1 - c1;
                E1_len = n;
                __syncthreads();
                if (n == 1) {
                    fe sc = fe_load(e1[c1]);
                    fe* v = e2[c2];
                    for (size_t i = threadIdx.x; i < E2_len; i += RT) fe_store(v + i, Fr::mul(fe_load(v + i), sc));
                    __syncthreads();
                }
            }
        }
        seq++;
        if (TRACE) {
            long long t = wall_clock64();
            tk_bind += t - tk0;
            tk0 = t;
        }
        if (round == nrounds) {  // final claims: coeffs[0], coeffs[1] of the fully bound layer
            if (threadIdx.x == 0) {
                mb->dbg[0] = (unsigned long long)tk_wait;
                mb->dbg[1] = (unsigned long long)tk_bind;
                mb->dbg[2] = (unsigned long long)tk_cubic;
                mb->dbg[3] = (unsigned long long)tk_pub;
                fe z = Fr::zero();
                fe v[4] = {fe_load(la[lcur]), NC == 2 ? fe_load(lb[lcur]) : z, fe_load(la[lcur] + 1), NC == 2 ? fe_load(lb[lcur] + 1) : z};
                for (int k = 0; k < 4; k++) fe_store(&mb->res[k], v[k]);
                __threadfence_system();
                __hip_atomic_store(&mb->res_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        // cubic sums of this round.  The serial chain is what a tiny round costs (one dependent Fr product is
        // ~1.3 us on a lone wave), so the three evaluation points go to three groups of five waves: a lane does
        // ONE (chunk, point) term -- l x r and e x scale side by side, then their product -- and the three sums
        // come out of one reduction pass.
        const fe *ca = la[lcur], *cb = lb[lcur];
        const fe *E1 = e1[c1], *E2 = e2[c2];
        const bool nested = E1_len != 1;
        const size_t E1_half = E1_len / 2;
        size_t nch = (len + 3) / 4;
        size_t limit = nested ? E1_half * E2_len : E2_len / 2;
        if (nch > limit) nch = limit;
        const int wave = threadIdx.x >> 6;
        const int pt = wave / 5;  // 0, 1, 2 = evaluation at 0, 2, 3; wave 15 idles
        fe acc = Fr::zero();
        if (pt < 3) {
            for (size_t c = threadIdx.x - 320 * pt; c < nch; c += 320) {
                fe e0, e1v, ek;
                if (nested) {
                    size_t x2 = c >> (__ffsll((long long)E1_half) - 1), x1 = c & (E1_half - 1);  // E1_half is a power of two
                    e0 = fe_load(E1 + 2 * x1);
                    e1v = fe_load(E1 + 2 * x1 + 1);
                } else {
                    e0 = fe_load(E2 + 2 * c);
                    e1v = fe_load(E2 + 2 * c + 1);
                }
                Sh<NC> l0 = sh_load_or_zero<NC>(ca, cb, 4 * c, len), r0 = sh_load_or_zero<NC>(ca, cb, 4 * c + 1, len);
                Sh<NC> lk = l0, rk = r0;
                ek = e0;
                if (pt > 0) {
                    Sh<NC> l1 = sh_load_or_zero<NC>(ca, cb, 4 * c + 2, len), r1 = sh_load_or_zero<NC>(ca, cb, 4 * c + 3, len);
                    Sh<NC> ml = sh_sub<NC>(l1, l0), mr = sh_sub<NC>(r1, r0);
                    fe me = Fr::sub(e1v, e0);
                    lk = sh_add<NC>(l1, ml);
                    rk = sh_add<NC>(r1, mr);
                    ek = Fr::add(e1v, me);
                    if (pt == 2) {
                        lk = sh_add<NC>(lk, ml);
                        rk = sh_add<NC>(rk, mr);
                        ek = Fr::add(ek, me);
                    }
                }
                fe lr = sh_local_mul<NC>(lk, rk);
                // same association as k_layer_cubic ((l x r) e) scale: field products are exact, any order agrees
                fe es = nested ? Fr::mul(ek, fe_load(E2 + (c >> (__ffsll((long long)E1_half) - 1)))) : ek;
                acc = Fr::add(acc, Fr::mul(lr, es));
            }
        }
        acc = fr_wave_sum(acc);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh16[wave] = acc;
        __syncthreads();
        if (TRACE) {
            long long t = wall_clock64();
            tk_cubic += t - tk0;
            tk0 = t;
        }
        if (threadIdx.x < 3) {
            fe v = sh16[5 * threadIdx.x];
            for (int i = 1; i < 5; i++) v = Fr::add(v, sh16[5 * threadIdx.x + i]);
            fe_store(&mb->res[threadIdx.x], v);
            __threadfence_system();
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&mb->res_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (TRACE) {
            long long t = wall_clock64();
            tk_pub += t - tk0;
            tk0 = t;
        }
    }
}

// out[j] = L[j] x R[j] (+ mask_j): local half of `mul_vec` (layer_output,
// dense_interleaved_poly.rs:122-141; local product ops.rs:71-78; zero-sharing mask
// mask_j = PRF(key_self, ctr+j) - PRF(key_prev, ctr+j), the keyed ChaCha12 PRF of prf.hip.hpp)
template <int NC>
__global__ void __launch_bounds__(PT) k_layer_output(const fe* __restrict__ a, const fe* __restrict__ b, size_t len,
                                                  fe* __restrict__ out, size_t n_out, int masked, prf_key key_self,
                                                  prf_key key_prev, uint64_t ctr) {
    size_t j = (size_t)blockIdx.x * PT + threadIdx.x;
    if (j >= n_out) return;
    Sh<NC> l = sh_load_or_zero<NC>(a, b, 2 * j, len), r = sh_load_or_zero<NC>(a, b, 2 * j + 1, len);
    fe v = sh_local_mul<NC>(l, r);
    if (masked) v = Fr::add(v, Fr::sub(prf_fr(key_self, ctr + j), prf_fr(key_prev, ctr + j)));
    fe_store(out + j, v);
}

// element-wise Rep3 mul_vec local part on two share vectors (rep3::arithmetic::mul_vec)
template <int NC>
__global__ void __launch_bounds__(PT) k_mul_vec_local(const fe* __restrict__ xa, const fe* __restrict__ xb,
                                                   const fe* __restrict__ ya, const fe* __restrict__ yb, size_t n,
                                                   fe* __restrict__ out, int masked, prf_key key_self, prf_key key_prev,
                                                   uint64_t ctr) {
    size_t j = (size_t)blockIdx.x * PT + threadIdx.x;
    if (j >= n) return;
    fe v = sh_local_mul<NC>(sh_load<NC>(xa, xb, j), sh_load<NC>(ya, yb, j));
    if (masked) v = Fr::add(v, Fr::sub(prf_fr(key_self, ctr + j), prf_fr(key_prev, ctr + j)));
    fe_store(out + j, v);
}

// ------------------------------------------------------------------ kernels: generic product rounds
// One round of `prove_arbitrary_worker` (co-jolt/src/subprotocols/sumcheck.rs:168-246) for the
// comb_funcs the reference uses -- a product of m polynomials of which at most one is shared
// (Spartan inner / shift sumchecks r1cs/spartan/worker.rs:162-235, output check
// read_write_memory/worker.rs:149-164): evaluations at x = 0, 2, .., degree of
// sum_i prod_j P_j(x; i) with HighToLow pairs (i, i + half) (sumcheck_evals, dense_mlpoly.rs:113-147).
// partial[e * gridDim.x + block]; shared factor enters as (a + b), TWO_INV applied by the finisher.
struct ProdRoundPtrs {  // <= 4 factors: the pointer table travels in the kernel arguments (no upload + stream sync per round)
    const fe* a[4];
    const fe* b[4];
};
template <int NC, int M>
__global__ void __launch_bounds__(PT) k_prod_round(ProdRoundPtrs tab, int shared_idx, size_t half, int degree, fe* __restrict__ partial) {
    const fe* const* pa = tab.a;
    const fe* const* pb = tab.b;
    __shared__ fe sh4[4];
    fe acc[4];
    for (int e = 0; e < 4; e++) acc[e] = Fr::zero();
    for (size_t i = (size_t)blockIdx.x * PT + threadIdx.x; i < half; i += (size_t)gridDim.x * PT) {
        fe cur[M], step[M];
        for (int j = 0; j < M; j++) {
            fe lo, hi;
            if (NC == 2 && j == shared_idx) {
                lo = Fr::add(fe_load(pa[j] + i), fe_load(pb[j] + i));
                hi = Fr::add(fe_load(pa[j] + i + half), fe_load(pb[j] + i + half));
            } else {
                lo = fe_load(pa[j] + i);
                hi = fe_load(pa[j] + i + half);
            }
            step[j] = Fr::sub(hi, lo);
            cur[j] = lo;
        }
        for (int e = 0; e < degree; e++) {
            // points 0, 2, 3, ...: after the first evaluation jump to x = 2
            if (e == 1)
                for (int j = 0; j < M; j++) cur[j] = Fr::add(Fr::add(cur[j], step[j]), step[j]);
            else if (e > 1)
                for (int j = 0; j < M; j++) cur[j] = Fr::add(cur[j], step[j]);
            fe prod = cur[0];
            for (int j = 1; j < M; j++) prod = Fr::mul(prod, cur[j]);
            acc[e] = Fr::add(acc[e], prod);
        }
    }
    for (int e = 0; e < degree; e++) {
        fe v = fr_block_sum(acc[e], sh4);
        if (threadIdx.x == 0) fe_store(partial + (size_t)e * gridDim.x + blockIdx.x, v);
    }
}

// co-spartan sumcheck #1 round (co-noir-spartan/co-spartan/src/sumcheck.rs:171-280): evaluations at
// X = 0..3 of sum_b [ (A x B)(X) * pub(X) - into_additive(C(X) * pub(X)) ], LowToHigh pairs (2b, 2b+1).
// The -C term is accumulated as (c.a + c.b) * pub separately so that TWO_INV is applied once per sum.
template <int NC>
__global__ void __launch_bounds__(PT) k_spartan_first(const fe* __restrict__ aa, const fe* __restrict__ ab, const fe* __restrict__ ba,
                                                   const fe* __restrict__ bb, const fe* __restrict__ ca, const fe* __restrict__ cb,
                                                   const fe* __restrict__ pub, size_t half, fe* __restrict__ partial) {
    __shared__ fe sh4[4];
    fe accp[4], accc[4];
    for (int e = 0; e < 4; e++) {
        accp[e] = Fr::zero();
        accc[e] = Fr::zero();
    }
    for (size_t b = (size_t)blockIdx.x * PT + threadIdx.x; b < half; b += (size_t)gridDim.x * PT) {
        Sh<NC> A = sh_load<NC>(aa, ab, 2 * b), B = sh_load<NC>(ba, bb, 2 * b), C = sh_load<NC>(ca, cb, 2 * b);
        Sh<NC> sA = sh_sub<NC>(sh_load<NC>(aa, ab, 2 * b + 1), A), sB = sh_sub<NC>(sh_load<NC>(ba, bb, 2 * b + 1), B),
               sC = sh_sub<NC>(sh_load<NC>(ca, cb, 2 * b + 1), C);
        fe P = fe_load(pub + 2 * b);
        fe sP = Fr::sub(fe_load(pub + 2 * b + 1), P);
        for (int e = 0; e < 4; e++) {
            accp[e] = Fr::add(accp[e], Fr::mul(sh_local_mul<NC>(A, B), P));
            accc[e] = Fr::add(accc[e], Fr::mul(sh_ab_sum<NC>(C), P));
            A = sh_add<NC>(A, sA);
            B = sh_add<NC>(B, sB);
            C = sh_add<NC>(C, sC);
            P = Fr::add(P, sP);
        }
    }
    for (int e = 0; e < 4; e++) {
        fe v = fr_block_sum(accp[e], sh4);
        if (threadIdx.x == 0) fe_store(partial + (size_t)e * gridDim.x + blockIdx.x, v);
        fe w = fr_block_sum(accc[e], sh4);
        if (threadIdx.x == 0) fe_store(partial + (size_t)(4 + e) * gridDim.x + blockIdx.x, w);
    }
}

// co-spartan sumcheck #2 round (sumcheck.rs:282-395): Rep3 evaluations at X = 0..2 of
// sum_b z(X) * (alpha a(X) + beta b(X) + gamma c(X)); component k of the share lands in partial rows 3k..3k+2
template <int NC>
__global__ void __launch_bounds__(PT) k_spartan_second(const fe* __restrict__ za, const fe* __restrict__ zb, const fe* __restrict__ pa,
                                                    const fe* __restrict__ pbv, const fe* __restrict__ pc, fe c0, fe c1, fe c2,
                                                    size_t half, fe* __restrict__ partial) {
    __shared__ fe sh4[4];
    fe acc[NC][3];
    for (int k = 0; k < NC; k++)
        for (int e = 0; e < 3; e++) acc[k][e] = Fr::zero();
    for (size_t b = (size_t)blockIdx.x * PT + threadIdx.x; b < half; b += (size_t)gridDim.x * PT) {
        Sh<NC> Z = sh_load<NC>(za, zb, 2 * b);
        Sh<NC> sZ = sh_sub<NC>(sh_load<NC>(za, zb, 2 * b + 1), Z);
        fe A = fe_load(pa + 2 * b), B = fe_load(pbv + 2 * b), C = fe_load(pc + 2 * b);
        fe sA = Fr::sub(fe_load(pa + 2 * b + 1), A), sB = Fr::sub(fe_load(pbv + 2 * b + 1), B), sC = Fr::sub(fe_load(pc + 2 * b + 1), C);
        for (int e = 0; e < 3; e++) {
            fe lin = Fr::add(Fr::add(Fr::mul(A, c0), Fr::mul(B, c1)), Fr::mul(C, c2));
            for (int k = 0; k < NC; k++) acc[k][e] = Fr::add(acc[k][e], Fr::mul(Z.c[k], lin));
            Z = sh_add<NC>(Z, sZ);
            A = Fr::add(A, sA);
            B = Fr::add(B, sB);
            C = Fr::add(C, sC);
        }
    }
    for (int k = 0; k < NC; k++)
        for (int e = 0; e < 3; e++) {
            fe v = fr_block_sum(acc[k][e], sh4);
            if (threadIdx.x == 0) fe_store(partial + (size_t)(3 * k + e) * gridDim.x + blockIdx.x, v);
        }
}

// zero_round (co-noir-spartan/co-spartan/src/worker.rs:153-182): za[row] = sum_nnz val_a * z[col] (same for
// b, c) on shares, CSR rows (field sums are order-independent, so COO vs CSR order cannot change a bit)
// Rows up to SPMV_LONG entries: one lane per row.  Longer rows (R1CS has them: the constant-1 column of the
// transposed matrices that `third_round` walks, worker.rs:241-249, holds an entry of almost every constraint)
// are cut into SPMV_CHUNK-entry work items queued on the device: one workgroup reduces one item
// (k_sparse_matvec3_items), then one workgroup per long row adds up its items' partial sums
// (k_sparse_matvec3_rows).  A dense column of 2^18 entries costs two short launches instead of a 2^18-step
// serial loop.  Queue layout in `q`: [0] number of long rows, [1] number of items, then rows[] = (row, first item,
// item count) and items[] = (row, chunk index).
static constexpr uint32_t SPMV_LONG = 64;
static constexpr uint32_t SPMV_CHUNK = 2048;
template <int NC>
__global__ void __launch_bounds__(PT) k_sparse_matvec3(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                                    const fe* __restrict__ va, const fe* __restrict__ vb, const fe* __restrict__ vc,
                                                    const fe* __restrict__ za, const fe* __restrict__ zb, size_t nrows,
                                                    fe* oa0, fe* oa1, fe* ob0, fe* ob1, fe* oc0, fe* oc1,
                                                    uint32_t* __restrict__ q, uint32_t* __restrict__ q_rows, uint32_t* __restrict__ q_items) {
    size_t r = (size_t)blockIdx.x * PT + threadIdx.x;
    if (r >= nrows) return;
    uint32_t e0 = row_ptr[r], e1 = row_ptr[r + 1];
    if (e1 - e0 > SPMV_LONG) {
        uint32_t nchunks = (e1 - e0 + SPMV_CHUNK - 1) / SPMV_CHUNK;
        uint32_t ri = atomicAdd(&q[0], 1u);
        uint32_t first = atomicAdd(&q[1], nchunks);
        q_rows[3 * ri] = (uint32_t)r;
        q_rows[3 * ri + 1] = first;
        q_rows[3 * ri + 2] = nchunks;
        for (uint32_t j = 0; j < nchunks; j++) {
            q_items[2 * (first + j)] = (uint32_t)r;
            q_items[2 * (first + j) + 1] = j;
        }
        return;
    }
    Sh<NC> A, B, C;
    for (int k = 0; k < NC; k++) A.c[k] = B.c[k] = C.c[k] = Fr::zero();
    for (uint32_t e = e0; e < e1; e++) {
        Sh<NC> z = sh_load<NC>(za, zb, col[e]);
        A = sh_add<NC>(A, sh_mul_public<NC>(z, fe_load(va + e)));
        B = sh_add<NC>(B, sh_mul_public<NC>(z, fe_load(vb + e)));
        C = sh_add<NC>(C, sh_mul_public<NC>(z, fe_load(vc + e)));
    }
    sh_store<NC>(oa0, oa1, r, A);
    sh_store<NC>(ob0, ob1, r, B);
    sh_store<NC>(oc0, oc1, r, C);
}
// partial[item][3][NC]: the sums of one SPMV_CHUNK-entry slice of a long row
template <int NC>
__global__ void __launch_bounds__(PT) k_sparse_matvec3_items(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                                          const fe* __restrict__ va, const fe* __restrict__ vb, const fe* __restrict__ vc,
                                                          const fe* __restrict__ za, const fe* __restrict__ zb, const uint32_t* __restrict__ q,
                                                          const uint32_t* __restrict__ q_items, fe* __restrict__ partial) {
    __shared__ fe sh4[4];
    uint32_t nitems = q[1];
    for (uint32_t it = blockIdx.x; it < nitems; it += gridDim.x) {  // every workgroup reaches the end of the queue
        uint32_t r = q_items[2 * it], j = q_items[2 * it + 1];
        uint32_t lo = row_ptr[r] + j * SPMV_CHUNK;
        uint32_t hi = min(lo + SPMV_CHUNK, row_ptr[r + 1]);
        Sh<NC> A, B, C;
        for (int k = 0; k < NC; k++) A.c[k] = B.c[k] = C.c[k] = Fr::zero();
        for (uint32_t e = lo + threadIdx.x; e < hi; e += PT) {
            Sh<NC> z = sh_load<NC>(za, zb, col[e]);
            A = sh_add<NC>(A, sh_mul_public<NC>(z, fe_load(va + e)));
            B = sh_add<NC>(B, sh_mul_public<NC>(z, fe_load(vb + e)));
            C = sh_add<NC>(C, sh_mul_public<NC>(z, fe_load(vc + e)));
        }
        fe* out = partial + (size_t)it * 3 * NC;
        for (int k = 0; k < NC; k++) {
            fe a = fr_block_sum(A.c[k], sh4), b = fr_block_sum(B.c[k], sh4), c = fr_block_sum(C.c[k], sh4);
            if (threadIdx.x == 0) {
                fe_store(out + k, a);
                fe_store(out + NC + k, b);
                fe_store(out + 2 * NC + k, c);
            }
        }
    }
}
template <int NC>
__global__ void __launch_bounds__(PT) k_sparse_matvec3_rows(const uint32_t* __restrict__ q, const uint32_t* __restrict__ q_rows,
                                                         const fe* __restrict__ partial, fe* oa0, fe* oa1, fe* ob0, fe* ob1, fe* oc0, fe* oc1) {
    __shared__ fe sh4[4];
    uint32_t nlong = q[0];
    for (uint32_t ri = blockIdx.x; ri < nlong; ri += gridDim.x) {
        uint32_t r = q_rows[3 * ri], first = q_rows[3 * ri + 1], n = q_rows[3 * ri + 2];
        fe acc[3 * NC];
        for (int k = 0; k < 3 * NC; k++) acc[k] = Fr::zero();
        for (uint32_t j = threadIdx.x; j < n; j += PT)
            for (int k = 0; k < 3 * NC; k++) acc[k] = Fr::add(acc[k], fe_load(partial + (size_t)(first + j) * 3 * NC + k));
        fe tot[3 * NC];
        for (int k = 0; k < 3 * NC; k++) tot[k] = fr_block_sum(acc[k], sh4);
        if (threadIdx.x == 0) {
            Sh<NC> A, B, C;
            for (int k = 0; k < NC; k++) {
                A.c[k] = tot[k];
                B.c[k] = tot[NC + k];
                C.c[k] = tot[2 * NC + k];
            }
            sh_store<NC>(oa0, oa1, r, A);
            sh_store<NC>(ob0, ob1, r, B);
            sh_store<NC>(oc0, oc1, r, C);
        }
    }
}

// ------------------------------------------------------------------ kernels: split-eq tables
__global__ void __launch_bounds__(PT) k_fold_pairs(const fe* __restrict__ in, fe* __restrict__ out, size_t n_out, fe r) {
    size_t i = (size_t)blockIdx.x * PT + threadIdx.x;
    if (i >= n_out) return;
    fe lo = fe_load(in + 2 * i), hi = fe_load(in + 2 * i + 1);
    fe_store(out + i, Fr::add(lo, Fr::mul(Fr::sub(hi, lo), r)));
}
__global__ void __launch_bounds__(PT) k_fold_halves(fe* __restrict__ v, size_t n_out, fe r) {
    size_t i = (size_t)blockIdx.x * PT + threadIdx.x;
    if (i >= n_out) return;
    fe lo = fe_load(v + i), hi = fe_load(v + i + n_out);
    fe_store(v + i, Fr::add(lo, Fr::mul(Fr::sub(hi, lo), r)));
}
__global__ void __launch_bounds__(PT) k_scale_by_first(fe* __restrict__ v, size_t n, const fe* __restrict__ s) {
    size_t i = (size_t)blockIdx.x * PT + threadIdx.x;
    if (i >= n) return;
    fe_store(v + i, Fr::mul(fe_load(v + i), fe_load(s)));
}
// EqPolynomial::evals(r) on device, big-endian (r[0] <-> MSB): level j doubles the table
__global__ void __launch_bounds__(PT) k_eq_expand(const fe* __restrict__ in, fe* __restrict__ out, size_t n_in, fe rj) {
    size_t i = (size_t)blockIdx.x * PT + threadIdx.x;
    if (i >= n_in) return;
    fe e = fe_load(in + i);
    fe hi = Fr::mul(e, rj);
    fe_store(out + 2 * i, Fr::sub(e, hi));
    fe_store(out + 2 * i + 1, hi);
}

// whole EqPolynomial::evals table in one launch (one workgroup; tables here are <= 2^13 entries:
// the two halves of a split-eq): level j doubles the table in place from the top index down
__global__ void k_store_fe(fe* out, fe v) { fe_store(out, v); }
__global__ void k_store_u32s(uint32_t* out, const uint32_t* __restrict__ in0, const uint32_t* __restrict__ in1, int n0, size_t stride0) {
    // out[i] = in0[i * stride0] for i < n0, then out[n0] = *in1 (small gathers into a pinned slot instead of strided device-to-host copies)
    int i = threadIdx.x;
    if (i < n0) out[i] = in0[(size_t)i * stride0];
    if (i == n0 && in1) out[n0] = *in1;
}
// The point travels in the kernel arguments (<= 13 x 32 B): a grand product builds ~220 of these tables per chained proof, and a
// host-to-device copy of the point plus the stream synchronisation that protects the caller's buffer cost more than the kernel.
struct EqBuildPoint {
    fe r[13];
};
__global__ void __launch_bounds__(1024) k_eq_build(EqBuildPoint pt, int nv, fe* __restrict__ out) {
    if (threadIdx.x == 0) fe_store(out, Fr::one());
    __syncthreads();
    size_t n = 1;
    for (int j = 0; j < nv; j++) {
        fe rj = pt.r[j];
        // read phase, then write phase: entry i expands to (2i, 2i+1)
        fe e[8];
        int cnt = 0;
        for (size_t i = threadIdx.x; i < n; i += 1024) e[cnt++] = fe_load(out + i);
        __syncthreads();
        cnt = 0;
        for (size_t i = threadIdx.x; i < n; i += 1024) {
            fe hi = Fr::mul(e[cnt], rj);
            fe_store(out + 2 * i, Fr::sub(e[cnt], hi));
            fe_store(out + 2 * i + 1, hi);
            cnt++;
        }
        __syncthreads();
        n *= 2;
    }
}

// ------------------------------------------------------------------ host helpers
// from the pool of the context whose ABI call is running on this thread (cozk_guard)
static fe* dev_alloc_fe(size_t n) { return (fe*)ctx_dev_alloc(t_cur_ctx, (n ? n : 1) * sizeof(fe)); }

// copy k small results from device scratch to host (sync)
// the stream's work up to here has finished: a stream memory write of a sequence number + a host spin on that pinned word replaces
// hipStreamSynchronize's event / interrupt path (tens of microseconds) with a cache-line hand-off (COZK_SYNC_ROUNDS: the blocking wait)
static void stream_drain_by_flag(cozk_ctx* ctx) {
    static const bool use_flag = getenv("COZK_SYNC_ROUNDS") == nullptr;
    if (!use_flag) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return;
    }
    if (!ctx->round_flag) HIP_TRY(hipHostMalloc((void**)&ctx->round_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    uint32_t seq = ++ctx->round_seq;
    HIP_TRY(hipStreamWriteValue32(ctx->stream, (void*)ctx->round_flag, seq, 0));
    volatile uint32_t* f = ctx->round_flag;
    uint64_t spins = 0;
    while (*f != seq) {
        __builtin_ia32_pause();
        if (++spins > (1ull << 22)) {  // ~10 ms without news: fall back to the blocking wait (also surfaces errors)
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            break;
        }
    }
}
// arm the finishing kernel's own publication of the round (COZK_FLAG_IN_FINISH=0 / COZK_SYNC_ROUNDS: not armed, fetch_fe drains the stream)
static RoundPublish arm_round_publish(cozk_ctx* ctx) {
    static const bool on = !(getenv("COZK_FLAG_IN_FINISH") && atoi(getenv("COZK_FLAG_IN_FINISH")) == 0) && getenv("COZK_SYNC_ROUNDS") == nullptr;
    if (!on) return RoundPublish{nullptr, nullptr, 0};
    if (!ctx->round_flag) HIP_TRY(hipHostMalloc((void**)&ctx->round_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    if (!ctx->finish_ticket) {
        HIP_TRY(hipMalloc((void**)&ctx->finish_ticket, 64));
        HIP_TRY(hipMemsetAsync(ctx->finish_ticket, 0, 64, ctx->stream));
    }
    ctx->armed_seq = ++ctx->round_seq;
    return RoundPublish{ctx->finish_ticket, ctx->round_flag, ctx->armed_seq};
}
// copy k small results to the host.  Round results are written by the finishing kernel straight into pinned (device-visible) host
// memory (d == the pinned slot): nothing to copy, only the stream to drain, ~270 times per grand product; anything else is copied
// into the pinned slot first.
static void fetch_fe(cozk_ctx* ctx, const fe* d, size_t k, fe* h) {
    fe* pin = (fe*)ctx_pinned(ctx, k * sizeof(fe));
    const uint32_t armed = ctx->armed_seq;
    ctx->armed_seq = 0;
    if (d == pin && armed) {  // the finishing kernel publishes the sequence number itself
        volatile uint32_t* f = ctx->round_flag;
        uint64_t spins = 0;
        while (*f != armed) {
            __builtin_ia32_pause();
            if (++spins > (1ull << 22)) {  // ~10 ms without news: the blocking wait also surfaces launch errors
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    } else {
        if (d != pin) HIP_TRY(hipMemcpyAsync(pin, d, k * sizeof(fe), hipMemcpyDeviceToHost, ctx->stream));
        stream_drain_by_flag(ctx);
    }
    for (size_t i = 0; i < k; i++) h[i] = pin[i];
}
// where a finishing kernel should put k round results (pinned host memory, mapped into the device)
static fe* result_slot(cozk_ctx* ctx, size_t k) { return (fe*)ctx_pinned(ctx, k * sizeof(fe)); }

// build an eq table on device: out (len 2^nv) from point r (host), big-endian
static void eq_evals_device(cozk_ctx* ctx, const fe* r, int nv, fe* out, fe* tmp) {
    if (nv <= 13) {
        // small table: one launch, the point in the kernel arguments
        (void)tmp;
        EqBuildPoint pt;
        for (int j = 0; j < nv; j++) pt.r[j] = r[j];
        for (int j = nv; j < 13; j++) pt.r[j] = Fr::zero();
        k_eq_build<<<1, 1024, 0, ctx->stream>>>(pt, nv, out);
        HIP_TRY(hipGetLastError());
        return;
    }
    fe* cur = (nv % 2 == 0) ? out : tmp;  // ping-pong so that the last level lands in `out`
    fe* nxt = (nv % 2 == 0) ? tmp : out;
    k_store_fe<<<1, 1, 0, ctx->stream>>>(cur, Fr::one());  // the value travels in the kernel arguments: no copy, no sync
    size_t n = 1;
    for (int j = 0; j < nv; j++) {
        k_eq_expand<<<grid_for(n), PT, 0, ctx->stream>>>(cur, nxt, n, r[j]);
        std::swap(cur, nxt);
        n *= 2;
    }
    HIP_TRY(hipGetLastError());
}

// ------------------------------------------------------------------ C ABI: dense polynomial
extern "C" {

int cozk_poly_create(cozk_ctx* ctx, int mode, const cozk_vec* a, const cozk_vec* b, cozk_poly** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && out && a && a->kind == COZK_SCALAR_FR && (mode == COZK_MODE_PLAIN || mode == COZK_MODE_REP3),
                     "poly_create: bad argument");
        COZK_REQUIRE(mode == COZK_MODE_PLAIN || (b && b->kind == COZK_SCALAR_FR && b->n == a->n), "poly_create: share b missing");
        cozk_poly* p = new cozk_poly();
        p->ctx = ctx;
        p->mode = mode;
        p->len = p->orig_len = a->n;
        p->cur = -1;
        p->own0 = true;
        p->a0 = dev_alloc_fe(a->n);
        p->b0 = nullptr;
        HIP_TRY(hipMemcpyAsync(p->a0, a->d, a->n * sizeof(fe), hipMemcpyDeviceToDevice, ctx->stream));
        if (mode == COZK_MODE_REP3) {
            p->b0 = dev_alloc_fe(a->n);
            HIP_TRY(hipMemcpyAsync(p->b0, b->d, a->n * sizeof(fe), hipMemcpyDeviceToDevice, ctx->stream));
        }
        *out = p;
    });
}

// zero-copy chunk view over the high variables (`split_poly`, dense_mlpoly.rs:275-301)
int cozk_poly_chunk(cozk_ctx* ctx, const cozk_poly* src, size_t offset, size_t len, cozk_poly** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && src && out && src->cur < 0 && offset + len <= src->orig_len, "poly_chunk: bad argument (source must be unbound)");
        cozk_poly* p = new cozk_poly();
        p->ctx = ctx;
        p->mode = src->mode;
        p->len = p->orig_len = len;
        p->cur = -1;
        p->own0 = false;
        p->a0 = src->a0 + offset;
        p->b0 = src->b0 ? src->b0 + offset : nullptr;
        *out = p;
    });
}

int cozk_poly_free(cozk_poly* p) {
    if (!p) return COZK_OK;
    if (p->own0) {
        if (p->a0) ctx_dev_free(p->ctx, p->a0);
        if (p->b0) ctx_dev_free(p->ctx, p->b0);
    }
    for (int w = 0; w < 2; w++)
        for (int c = 0; c < 2; c++)
            if (p->buf[w][c]) ctx_dev_free(p->ctx, p->buf[w][c]);
    delete p;
    return COZK_OK;
}

size_t cozk_poly_len(const cozk_poly* p) { return p ? p->len : 0; }
int cozk_poly_mode(const cozk_poly* p) { return p ? p->mode : 0; }

// current coefficients -> host (a: len x 4 u64; b likewise or NULL)
int cozk_poly_download(cozk_ctx* ctx, const cozk_poly* p, uint64_t* a, uint64_t* b) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && p && a, "poly_download: bad argument");
        HIP_TRY(hipMemcpyAsync(a, poly_a(p), p->len * sizeof(fe), hipMemcpyDeviceToHost, ctx->stream));
        if (p->mode == COZK_MODE_REP3 && b)
            HIP_TRY(hipMemcpyAsync(b, poly_b(p), p->len * sizeof(fe), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    });
}

// `copy_share_a` (dense_mlpoly.rs:103-110) as a zero-copy view of the current coefficients
int cozk_poly_share_view(cozk_ctx* ctx, const cozk_poly* p, int component, cozk_vec** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && p && out && (component == 0 || (component == 1 && p->mode == COZK_MODE_REP3)), "poly_share_view: bad argument");
        const fe* d = component == 0 ? poly_a(p) : poly_b(p);
        *out = new cozk_vec{ctx, p->len, COZK_SCALAR_FR, (void*)d, p->len * sizeof(fe), false};
    });
}

// PolynomialBinding::bind / bind_parallel (dense_mlpoly.rs:310-459; the correct `left +` formula
// is used for every case -- see SURVEY 9 on the reference's bind_parallel quirk)
int cozk_poly_bind(cozk_ctx* ctx, cozk_poly* p, const uint64_t r[4], int order) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && p && r && p->len >= 2 && (order == COZK_LOW_TO_HIGH || order == COZK_HIGH_TO_LOW), "poly_bind: bad argument");
        size_t n = p->len / 2;
        fe rr = fe_from_u64x4(r);
        const fe* ia = poly_a(p);
        const fe* ib = poly_b(p);
        int dst;
        if (p->cur >= 0 && order == COZK_HIGH_TO_LOW) {
            dst = p->cur;  // in place: lane i reads (i, i+n) and writes i
        } else {
            dst = p->cur < 0 ? 0 : 1 - p->cur;
            if (p->cap[dst] < n) {
                for (int c = 0; c < 2; c++) {
                    if (p->buf[dst][c]) ctx_dev_free(p->ctx, p->buf[dst][c]);
                    p->buf[dst][c] = nullptr;
                }
                p->buf[dst][0] = dev_alloc_fe(n);
                if (p->mode == COZK_MODE_REP3) p->buf[dst][1] = dev_alloc_fe(n);
                p->cap[dst] = n;
            }
        }
        fe* oa = p->buf[dst][0];
        fe* ob = p->buf[dst][1];
        if (p->mode == COZK_MODE_REP3) {
            if (order == COZK_LOW_TO_HIGH) k_poly_bind<2, COZK_LOW_TO_HIGH><<<grid_for(n), PT, 0, ctx->stream>>>(ia, ib, oa, ob, n, rr);
            else k_poly_bind<2, COZK_HIGH_TO_LOW><<<grid_for(n), PT, 0, ctx->stream>>>(ia, ib, oa, ob, n, rr);
        } else {
            if (order == COZK_LOW_TO_HIGH) k_poly_bind<1, COZK_LOW_TO_HIGH><<<grid_for(n), PT, 0, ctx->stream>>>(ia, ib, oa, ob, n, rr);
            else k_poly_bind<1, COZK_HIGH_TO_LOW><<<grid_for(n), PT, 0, ctx->stream>>>(ia, ib, oa, ob, n, rr);
        }
        HIP_TRY(hipGetLastError());
        p->cur = dst;
        p->len = n;
    });
}

// get_bound_coeff / final_sumcheck_claim (dense_mlpoly.rs:251-257,461-465)
int cozk_poly_get_coeff(cozk_ctx* ctx, const cozk_poly* p, size_t index, uint64_t a[4], uint64_t b[4]) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && p && a && index < p->len, "poly_get_coeff: bad argument");
        fe h[2];
        fetch_fe(ctx, poly_a(p) + index, 1, &h[0]);
        fe_to_u64x4(h[0], a);
        if (p->mode == COZK_MODE_REP3 && b) {
            fetch_fe(ctx, poly_b(p) + index, 1, &h[1]);
            fe_to_u64x4(h[1], b);
        }
    });
}

// build EqPolynomial::evals(r) on device (big-endian) -> cozk_vec of 2^nv elements
int cozk_eq_evals(cozk_ctx* ctx, const uint64_t* r, int nv, cozk_vec** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && out && nv >= 0 && nv < 40 && (r || nv == 0), "eq_evals: bad argument");
        size_t n = (size_t)1 << nv;
        std::vector<fe> rr(nv);
        for (int j = 0; j < nv; j++) rr[j] = fe_from_u64x4(r + 4 * j);
        fe* d = dev_alloc_fe(n);
        ctx->scratch2.reserve(n * sizeof(fe));
        eq_evals_device(ctx, rr.data(), nv, d, ctx->scratch2.as<fe>());
        *out = new cozk_vec{ctx, n, COZK_SCALAR_FR, d, n * sizeof(fe), true};
    });
}

// Rep3DensePolynomial::batch_evaluate (dense_mlpoly.rs:183-192): out[k] = additive share of poly_k(r)
// = TWO_INV * sum_i (a_i + b_i) chi_i (REP3) or sum_i a_i chi_i (PLAIN).  chi: FR vec, len >= max len.
int cozk_poly_batch_evaluate_at_chi(cozk_ctx* ctx, const cozk_poly* const* polys, size_t k, const cozk_vec* chi,
                                    uint64_t* out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && polys && k > 0 && chi && out && chi->kind == COZK_SCALAR_FR, "batch_evaluate: bad argument");
        int mode = polys[0]->mode;
        size_t maxlen = 0;
        std::vector<const fe*> ha(k), hb(k);
        std::vector<size_t> hl(k);
        for (size_t i = 0; i < k; i++) {
            COZK_REQUIRE(polys[i] && polys[i]->mode == mode, "batch_evaluate: mixed share modes");
            COZK_REQUIRE(polys[i]->len <= chi->n, "batch_evaluate: chi shorter than polynomial");
            ha[i] = poly_a(polys[i]);
            hb[i] = poly_b(polys[i]);
            hl[i] = polys[i]->len;
            if (hl[i] > maxlen) maxlen = hl[i];
        }
        unsigned gx = grid_capped(maxlen);
        // few, long lanes: a lane ends with one wide reduction per accumulator (8 products) + the block sums, which would outweigh
        // its multiply-add chains if it only had a handful of elements (env COZK_EVAL_GX for A/B runs)
        static const unsigned gx_cap = getenv("COZK_EVAL_GX") ? (unsigned)atoi(getenv("COZK_EVAL_GX")) : 192u;
        if (gx > gx_cap) gx = gx_cap;
        size_t meta = k * (2 * sizeof(void*) + sizeof(size_t));
        ctx->scratch.reserve(meta + (k * gx + k) * sizeof(fe) + 64);
        char* base = (char*)ctx->scratch.p;
        const fe** da = (const fe**)base;
        const fe** db = da + k;
        size_t* dl = (size_t*)(db + k);
        fe* partial = (fe*)(((uintptr_t)(dl + k) + 31) & ~(uintptr_t)31);
        fe* res = result_slot(ctx, k);  // pinned: no device-to-host copy behind the finishing kernel
        HIP_TRY(hipMemcpyAsync(da, ha.data(), k * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(db, hb.data(), k * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(dl, hl.data(), k * sizeof(size_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));  // host vectors go out of scope
        dim3 grid(gx, (unsigned)((k + 1) / 2));
        {
            // algorithmic bytes: chi read once for the batch, every polynomial once (96 n, then 64 n per extra polynomial: SURVEY 8d K6)
            uint64_t alg = (uint64_t)chi->n * 32;
            for (size_t i = 0; i < k; i++) alg += (uint64_t)hl[i] * (mode == COZK_MODE_REP3 ? 64 : 32);
            ProfScope prof(ctx, COZK_PROF_EVAL_CHI, alg);
            if (mode == COZK_MODE_REP3) k_poly_eval_chi<2><<<grid, PT, 0, ctx->stream>>>(da, db, dl, (const fe*)chi->d, (int)k, partial);
            else k_poly_eval_chi<1><<<grid, PT, 0, ctx->stream>>>(da, db, dl, (const fe*)chi->d, (int)k, partial);
        }
        k_finish_sums<<<(unsigned)k, PT, 0, ctx->stream>>>(partial, gx, fr_two_inv(), mode == COZK_MODE_REP3 ? 1 : 0, res, arm_round_publish(ctx));
        HIP_TRY(hipGetLastError());
        std::vector<fe> h(k);
        fetch_fe(ctx, res, k, h.data());
        for (size_t i = 0; i < k; i++) fe_to_u64x4(h[i], out + 4 * i);
    });
}

// dot_product_with_public (dense_mlpoly.rs:228-234) -> share (a[4], b[4])
int cozk_poly_dot_product_with_public(cozk_ctx* ctx, const cozk_poly* p, const cozk_vec* pub, uint64_t a[4], uint64_t b[4]) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && p && pub && a && pub->kind == COZK_SCALAR_FR && pub->n == p->len, "dot_product: length mismatch (zip_eq)");
        unsigned gx = grid_capped(p->len);
        if (gx > 512) gx = 512;
        ctx->scratch.reserve((2 * gx + 2) * sizeof(fe));
        fe* partial = ctx->scratch.as<fe>();
        fe* res = result_slot(ctx, 2);
        int nc = p->mode == COZK_MODE_REP3 ? 2 : 1;
        if (nc == 2) k_poly_dot_public<2><<<gx, PT, 0, ctx->stream>>>(poly_a(p), poly_b(p), (const fe*)pub->d, p->len, partial);
        else k_poly_dot_public<1><<<gx, PT, 0, ctx->stream>>>(poly_a(p), poly_b(p), (const fe*)pub->d, p->len, partial);
        k_finish_sums<<<nc, PT, 0, ctx->stream>>>(partial, gx, Fr::one(), 0, res, arm_round_publish(ctx));
        HIP_TRY(hipGetLastError());
        fe h[2];
        fetch_fe(ctx, res, nc, h);
        fe_to_u64x4(h[0], a);
        if (nc == 2 && b) fe_to_u64x4(h[1], b);
    });
}

// Rep3DensePolynomial::linear_combination / Rep3MultilinearPolynomial::linear_combination
// (dense_mlpoly.rs:195-226; multilinear_polynomial.rs:196-296).  A PLAIN polynomial inside a REP3
// combination is a public polynomial entering through its trivial share (P0: a = v, P1: b = v,
// P2: nothing; types.rs:90-96) -- `party_id` selects which.
int cozk_poly_linear_combination(cozk_ctx* ctx, const cozk_poly* const* polys, const uint64_t* coeffs, size_t k,
                                 int out_mode, int party_id, cozk_poly** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && polys && coeffs && k > 0 && out && (out_mode == COZK_MODE_PLAIN || out_mode == COZK_MODE_REP3) &&
                         party_id >= 0 && party_id < 3,
                     "linear_combination: bad argument");
        size_t maxlen = 0;
        std::vector<const fe*> ha(k), hb(k);
        std::vector<size_t> hl(k);
        std::vector<fe> hc(k);
        for (size_t i = 0; i < k; i++) {
            const cozk_poly* p = polys[i];
            COZK_REQUIRE(p, "linear_combination: null polynomial");
            hl[i] = p->len;
            hc[i] = fe_from_u64x4(coeffs + 4 * i);
            if (p->len > maxlen) maxlen = p->len;
            if (p->mode == COZK_MODE_REP3) {
                COZK_REQUIRE(out_mode == COZK_MODE_REP3, "linear_combination: shared input needs a shared output");
                ha[i] = poly_a(p);
                hb[i] = poly_b(p);
            } else if (out_mode == COZK_MODE_REP3) {
                // public polynomial: `add_public` puts it on P0's a / P1's b / nowhere on P2
                // (SharedOrPublic::add_public_assign, co-jolt/src/utils/shared_or_public.rs:150-152)
                ha[i] = party_id == 0 ? poly_a(p) : nullptr;
                hb[i] = party_id == 1 ? poly_a(p) : nullptr;
            } else {
                ha[i] = poly_a(p);
                hb[i] = nullptr;
            }
        }
        cozk_poly* o = new cozk_poly();
        o->ctx = ctx;
        o->mode = out_mode;
        o->len = o->orig_len = maxlen;
        o->cur = -1;
        o->own0 = true;
        o->a0 = dev_alloc_fe(maxlen);
        o->b0 = out_mode == COZK_MODE_REP3 ? dev_alloc_fe(maxlen) : nullptr;
        size_t meta = k * (2 * sizeof(void*) + sizeof(size_t)) + 64 + k * sizeof(fe);
        ctx->scratch.reserve(meta + 64);
        char* base = (char*)ctx->scratch.p;
        const fe** da = (const fe**)base;
        const fe** db = da + k;
        size_t* dl = (size_t*)(db + k);
        fe* dc = (fe*)(((uintptr_t)(dl + k) + 31) & ~(uintptr_t)31);
        HIP_TRY(hipMemcpyAsync(da, ha.data(), k * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(db, hb.data(), k * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(dl, hl.data(), k * sizeof(size_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(dc, hc.data(), k * sizeof(fe), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        {
            // algorithmic bytes: every operand component read once + the result written once ((m + 1) n S)
            uint64_t alg = (uint64_t)maxlen * (out_mode == COZK_MODE_REP3 ? 64 : 32);
            for (size_t i = 0; i < k; i++) alg += (uint64_t)hl[i] * ((ha[i] ? 32 : 0) + (out_mode == COZK_MODE_REP3 && hb[i] ? 32 : 0));
            ProfScope prof(ctx, COZK_PROF_LINCOMB, alg);
            if (out_mode == COZK_MODE_REP3) k_poly_lincomb<2><<<grid_for(maxlen), PT, 0, ctx->stream>>>(da, db, dl, dc, (int)k, o->a0, o->b0, maxlen);
            else k_poly_lincomb<1><<<grid_for(maxlen), PT, 0, ctx->stream>>>(da, db, dl, dc, (int)k, o->a0, o->b0, maxlen);
        }
        HIP_TRY(hipGetLastError());
        *out = o;
    });
}

// compute_leaves (K11): see k_fingerprint_leaves.  Writes n leaves at out_a/out_b[offset ..]; public polynomials
// (MODE_PLAIN) among `polys` count as part of the public term when the output is REP3.
int cozk_fingerprint_leaves(cozk_ctx* ctx, const cozk_vec* const* cols, const uint64_t* col_coeffs, size_t ks, const cozk_poly* const* polys,
                            const uint64_t* poly_coeffs, size_t kp, const uint64_t constant[4], int mode, int party_id, cozk_vec* out_a,
                            cozk_vec* out_b, size_t offset, size_t n) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && constant && out_a && out_a->kind == COZK_SCALAR_FR && (mode == COZK_MODE_PLAIN || mode == COZK_MODE_REP3) &&
                         party_id >= 0 && party_id < 3 && n > 0 && offset + n <= out_a->n && (ks == 0 || (cols && col_coeffs)) &&
                         (kp == 0 || (polys && poly_coeffs)) && ks <= 32 && kp <= 32,
                     "fingerprint_leaves: bad argument");
        COZK_REQUIRE(mode == COZK_MODE_PLAIN || (out_b && out_b->kind == COZK_SCALAR_FR && out_b->n == out_a->n), "fingerprint_leaves: share b output missing");
        std::vector<SmallCol> hcols(ks);
        std::vector<fe> hcs(ks), hds(kp);
        std::vector<const fe*> ha(kp), hb(kp);
        for (size_t k = 0; k < ks; k++) {
            const cozk_vec* v = cols[k];
            COZK_REQUIRE(v && v->n >= n && (v->kind == COZK_SCALAR_U8 || v->kind == COZK_SCALAR_U16 || v->kind == COZK_SCALAR_U32 || v->kind == COZK_SCALAR_U64),
                         "fingerprint_leaves: compact columns are U8 / U16 / U32 / U64 vectors of >= n entries");
            hcols[k] = SmallCol{v->d, v->kind};
            hcs[k] = Fr::to_mont(fe_from_u64x4(col_coeffs + 4 * k));  // c * R^2: one product with a plain integer -> Montgomery c * v
        }
        for (size_t j = 0; j < kp; j++) {
            const cozk_poly* p = polys[j];
            COZK_REQUIRE(p && p->len >= n, "fingerprint_leaves: polynomial shorter than n");
            hds[j] = fe_from_u64x4(poly_coeffs + 4 * j);
            if (p->mode == COZK_MODE_REP3) {
                COZK_REQUIRE(mode == COZK_MODE_REP3, "fingerprint_leaves: shared input needs a shared output");
                ha[j] = poly_a(p);
                hb[j] = poly_b(p);
            } else if (mode == COZK_MODE_REP3) {
                ha[j] = party_id == 0 ? poly_a(p) : nullptr;  // add_public: P0's a, P1's b
                hb[j] = party_id == 1 ? poly_a(p) : nullptr;
            } else {
                ha[j] = poly_a(p);
                hb[j] = nullptr;
            }
        }
        fe cst0 = fe_from_u64x4(constant);
        if (ks <= 16 && kp <= 8) {
            FingerprintArgs fa;
            memset(&fa, 0, sizeof fa);
            for (size_t k = 0; k < ks; k++) {
                fa.cols[k] = hcols[k];
                fa.cs[k] = hcs[k];
            }
            for (size_t j = 0; j < kp; j++) {
                fa.pa[j] = ha[j];
                fa.pb[j] = hb[j];
                fa.ds[j] = hds[j];
            }
            fe* oa0 = (fe*)out_a->d + offset;
            if (mode == COZK_MODE_REP3)
                k_fingerprint_leaves_args<2><<<grid_for(n), PT, 0, ctx->stream>>>(fa, (int)ks, (int)kp, cst0, party_id == 0, party_id == 1, oa0,
                                                                                 (fe*)out_b->d + offset, n);
            else
                k_fingerprint_leaves_args<1><<<grid_for(n), PT, 0, ctx->stream>>>(fa, (int)ks, (int)kp, cst0, 1, 0, oa0, nullptr, n);
            HIP_TRY(hipGetLastError());
            return;
        }
        size_t meta = ks * sizeof(SmallCol) + 64 + (ks + kp) * sizeof(fe) + 64 + 2 * kp * sizeof(void*) + 64;
        ctx->scratch.reserve(meta);
        char* base = (char*)ctx->scratch.p;
        SmallCol* dcols = (SmallCol*)base;
        fe* dcs = (fe*)(((uintptr_t)(dcols + ks) + 31) & ~(uintptr_t)31);
        fe* dds = dcs + ks;
        const fe** da = (const fe**)(((uintptr_t)(dds + kp) + 15) & ~(uintptr_t)15);
        const fe** db = da + kp;
        if (ks) {
            HIP_TRY(hipMemcpyAsync(dcols, hcols.data(), ks * sizeof(SmallCol), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipMemcpyAsync(dcs, hcs.data(), ks * sizeof(fe), hipMemcpyHostToDevice, ctx->stream));
        }
        if (kp) {
            HIP_TRY(hipMemcpyAsync(dds, hds.data(), kp * sizeof(fe), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipMemcpyAsync(da, ha.data(), kp * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipMemcpyAsync(db, hb.data(), kp * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));  // the staging vectors live on this stack frame
        fe cst = fe_from_u64x4(constant);
        fe* oa = (fe*)out_a->d + offset;
        if (mode == COZK_MODE_REP3)
            k_fingerprint_leaves<2><<<grid_for(n), PT, 0, ctx->stream>>>(dcols, dcs, (int)ks, da, db, dds, (int)kp, cst, party_id == 0, party_id == 1, oa,
                                                                        (fe*)out_b->d + offset, n);
        else
            k_fingerprint_leaves<1><<<grid_for(n), PT, 0, ctx->stream>>>(dcols, dcs, (int)ks, da, db, dds, (int)kp, cst, 1, 0, oa, nullptr, n);
        HIP_TRY(hipGetLastError());
    });
}

// one round of the batched opening-reduction sumcheck for the openings that are "live" this round
// (compute_quadratic, opening_proof.rs:374-414): out[2*i] = eval_0, out[2*i+1] = eval_2 as additive
// shares (REP3: TWO_INV * (a+b) already applied; PLAIN: the value).
int cozk_open_quadratic_evals(cozk_ctx* ctx, const cozk_poly* const* polys, const cozk_poly* const* eqs, size_t k,
                              uint64_t* out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && polys && eqs && k > 0 && out, "open_quadratic: bad argument");
        int mode = polys[0]->mode;
        std::vector<const fe*> ha(k), hb(k), he(k);
        std::vector<size_t> hh(k);
        size_t maxh = 0;
        for (size_t i = 0; i < k; i++) {
            COZK_REQUIRE(polys[i] && eqs[i] && polys[i]->mode == mode && eqs[i]->mode == COZK_MODE_PLAIN &&
                             eqs[i]->len == polys[i]->len && polys[i]->len >= 2,
                         "open_quadratic: polynomial / eq mismatch");
            ha[i] = poly_a(polys[i]);
            hb[i] = poly_b(polys[i]);
            he[i] = poly_a(eqs[i]);
            hh[i] = polys[i]->len / 2;
            if (hh[i] > maxh) maxh = hh[i];
        }
        unsigned gx = grid_capped(maxh);
        if (gx > 512) gx = 512;
        size_t meta = k * (3 * sizeof(void*) + sizeof(size_t));
        ctx->scratch.reserve(meta + (2 * k * gx + 2 * k) * sizeof(fe) + 64);
        char* base = (char*)ctx->scratch.p;
        const fe** da = (const fe**)base;
        const fe** db = da + k;
        const fe** de = db + k;
        size_t* dh = (size_t*)(de + k);
        fe* partial = (fe*)(((uintptr_t)(dh + k) + 31) & ~(uintptr_t)31);
        fe* res = result_slot(ctx, 2 * k);
        dim3 grid(gx, (unsigned)k);
        if (k <= 16) {
            OpenQuadArgs args;
            for (size_t i = 0; i < 16; i++) {
                args.a[i] = i < k ? ha[i] : nullptr;
                args.b[i] = i < k ? hb[i] : nullptr;
                args.eq[i] = i < k ? he[i] : nullptr;
                args.half[i] = i < k ? hh[i] : 0;
            }
            if (mode == COZK_MODE_REP3) k_open_quadratic_args<2><<<grid, PT, 0, ctx->stream>>>(args, partial);
            else k_open_quadratic_args<1><<<grid, PT, 0, ctx->stream>>>(args, partial);
        } else {
            HIP_TRY(hipMemcpyAsync(da, ha.data(), k * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipMemcpyAsync(db, hb.data(), k * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipMemcpyAsync(de, he.data(), k * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipMemcpyAsync(dh, hh.data(), k * sizeof(size_t), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (mode == COZK_MODE_REP3) k_open_quadratic<2><<<grid, PT, 0, ctx->stream>>>(da, db, de, dh, partial);
            else k_open_quadratic<1><<<grid, PT, 0, ctx->stream>>>(da, db, de, dh, partial);
        }
        k_finish_sums<<<(unsigned)(2 * k), PT, 0, ctx->stream>>>(partial, gx, fr_two_inv(), mode == COZK_MODE_REP3 ? 1 : 0, res, arm_round_publish(ctx));
        HIP_TRY(hipGetLastError());
        std::vector<fe> h(2 * k);
        fetch_fe(ctx, res, 2 * k, h.data());
        for (size_t i = 0; i < 2 * k; i++) fe_to_u64x4(h[i], out + 4 * i);
    });
}

// one round of prove_arbitrary_worker's evaluation loop (sumcheck.rs:189-215) for a product of m
// polynomials (<= 4), at most one of them REP3: out[e] = additive evaluation at x = 0, 2, .., degree
// (e = 0..degree-1).  All polynomials must have the same (current) length; HighToLow.
int cozk_prod_sumcheck_evals(cozk_ctx* ctx, const cozk_poly* const* polys, size_t m, int degree, uint64_t* out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && polys && out && m >= 1 && m <= 4 && degree >= 1 && degree <= 4, "prod_sumcheck: bad argument");
        size_t len = polys[0]->len;
        int shared = -1;
        const fe* ha[4] = {nullptr, nullptr, nullptr, nullptr};
        const fe* hb[4] = {nullptr, nullptr, nullptr, nullptr};
        for (size_t j = 0; j < m; j++) {
            COZK_REQUIRE(polys[j] && polys[j]->len == len && len >= 2, "prod_sumcheck: length mismatch");
            ha[j] = poly_a(polys[j]);
            hb[j] = poly_b(polys[j]);
            if (polys[j]->mode == COZK_MODE_REP3) {
                COZK_REQUIRE(shared < 0, "prod_sumcheck: at most one shared factor (a shared x shared product needs a reshare)");
                shared = (int)j;
            }
        }
        size_t half = len / 2;
        unsigned gx = grid_capped(half);
        if (gx > 512) gx = 512;
        ctx->scratch.reserve((4 * (size_t)gx + 4) * sizeof(fe) + 64);
        fe* partial = ctx->scratch.as<fe>();
        fe* res = result_slot(ctx, 4);
        ProdRoundPtrs tab;
        for (int j = 0; j < 4; j++) {
            tab.a[j] = ha[j];
            tab.b[j] = hb[j];
        }
#define PROD_LAUNCH(NCV, MV) k_prod_round<NCV, MV><<<gx, PT, 0, ctx->stream>>>(tab, shared, half, degree, partial)
        if (shared >= 0) {
            switch (m) { case 1: PROD_LAUNCH(2, 1); break; case 2: PROD_LAUNCH(2, 2); break; case 3: PROD_LAUNCH(2, 3); break; default: PROD_LAUNCH(2, 4); }
        } else {
            switch (m) { case 1: PROD_LAUNCH(1, 1); break; case 2: PROD_LAUNCH(1, 2); break; case 3: PROD_LAUNCH(1, 3); break; default: PROD_LAUNCH(1, 4); }
        }
#undef PROD_LAUNCH
        k_finish_sums<<<(unsigned)degree, PT, 0, ctx->stream>>>(partial, gx, fr_two_inv(), shared >= 0 ? 1 : 0, res, arm_round_publish(ctx));
        HIP_TRY(hipGetLastError());
        fe h[4];
        fetch_fe(ctx, res, (size_t)degree, h);
        for (int e = 0; e < degree; e++) fe_to_u64x4(h[e], out + 4 * e);
    });
}

// Rep3Sumcheck::first_sumcheck_prove_round evaluations (co-spartan/src/sumcheck.rs:171-280), before the
// additive zero-mask: out[t] for X = t = 0..3.  za, zb, zc share polys (same mode), pub a PLAIN poly.
int cozk_spartan_first_round(cozk_ctx* ctx, const cozk_poly* za, const cozk_poly* zb, const cozk_poly* zc, const cozk_poly* pub,
                             uint64_t out[16]) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && za && zb && zc && pub && out && pub->mode == COZK_MODE_PLAIN && za->mode == zb->mode && za->mode == zc->mode &&
                         za->len == zb->len && za->len == zc->len && za->len == pub->len && za->len >= 2,
                     "spartan_first_round: bad argument");
        size_t half = za->len / 2;
        unsigned gx = grid_capped(half);
        if (gx > 512) gx = 512;
        ctx->scratch.reserve((8 * (size_t)gx + 8) * sizeof(fe));
        fe* partial = ctx->scratch.as<fe>();
        fe* res = result_slot(ctx, 8);
        if (za->mode == COZK_MODE_REP3)
            k_spartan_first<2><<<gx, PT, 0, ctx->stream>>>(poly_a(za), poly_b(za), poly_a(zb), poly_b(zb), poly_a(zc), poly_b(zc), poly_a(pub), half, partial);
        else
            k_spartan_first<1><<<gx, PT, 0, ctx->stream>>>(poly_a(za), nullptr, poly_a(zb), nullptr, poly_a(zc), nullptr, poly_a(pub), half, partial);
        k_finish_sums<<<8, PT, 0, ctx->stream>>>(partial, gx, Fr::one(), 0, res, arm_round_publish(ctx));
        HIP_TRY(hipGetLastError());
        fe h[8];
        fetch_fe(ctx, res, 8, h);
        for (int e = 0; e < 4; e++) {
            fe c = za->mode == COZK_MODE_REP3 ? Fr::mul(h[4 + e], fr_two_inv()) : h[4 + e];
            fe_to_u64x4(Fr::sub(h[e], c), out + 4 * e);
        }
    });
}

// Rep3Sumcheck::second_sumcheck_prove_round evaluations (sumcheck.rs:282-395), before the Rep3 mask:
// out_a[t], out_b[t] for X = t = 0..2 (out_b = 0 for PLAIN).  coef = (alpha, beta, gamma), 12 u64.
int cozk_spartan_second_round(cozk_ctx* ctx, const cozk_poly* z, const cozk_poly* a, const cozk_poly* b, const cozk_poly* c,
                              const uint64_t coef[12], uint64_t out_a[12], uint64_t out_b[12]) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && z && a && b && c && coef && out_a && out_b && a->mode == COZK_MODE_PLAIN && b->mode == COZK_MODE_PLAIN &&
                         c->mode == COZK_MODE_PLAIN && z->len == a->len && z->len == b->len && z->len == c->len && z->len >= 2,
                     "spartan_second_round: bad argument");
        size_t half = z->len / 2;
        unsigned gx = grid_capped(half);
        if (gx > 512) gx = 512;
        ctx->scratch.reserve((6 * (size_t)gx + 6) * sizeof(fe));
        fe* partial = ctx->scratch.as<fe>();
        fe* res = result_slot(ctx, 6);
        int nc = z->mode == COZK_MODE_REP3 ? 2 : 1;
        fe c0 = fe_from_u64x4(coef), c1 = fe_from_u64x4(coef + 4), c2 = fe_from_u64x4(coef + 8);
        if (nc == 2) k_spartan_second<2><<<gx, PT, 0, ctx->stream>>>(poly_a(z), poly_b(z), poly_a(a), poly_a(b), poly_a(c), c0, c1, c2, half, partial);
        else k_spartan_second<1><<<gx, PT, 0, ctx->stream>>>(poly_a(z), nullptr, poly_a(a), poly_a(b), poly_a(c), c0, c1, c2, half, partial);
        k_finish_sums<<<(unsigned)(3 * nc), PT, 0, ctx->stream>>>(partial, gx, Fr::one(), 0, res, arm_round_publish(ctx));
        HIP_TRY(hipGetLastError());
        fe h[6];
        fetch_fe(ctx, res, (size_t)(3 * nc), h);
        for (int e = 0; e < 3; e++) {
            fe_to_u64x4(h[e], out_a + 4 * e);
            if (nc == 2) fe_to_u64x4(h[3 + e], out_b + 4 * e);
            else for (int k = 0; k < 4; k++) out_b[4 * e + k] = 0;
        }
    });
}

// SpartanProverWorker::zero_round (co-spartan/src/worker.rs:153-182): (za, zb, zc) = (A, B, C) . z on shares.
// CSR: row_ptr (U32, nrows + 1), col (U32, nnz), val_a/b/c (FR, nnz).
int cozk_sparse_matvec3(cozk_ctx* ctx, const cozk_vec* row_ptr, const cozk_vec* col, const cozk_vec* val_a, const cozk_vec* val_b,
                        const cozk_vec* val_c, const cozk_poly* z, cozk_poly** out_za, cozk_poly** out_zb, cozk_poly** out_zc) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && row_ptr && col && val_a && val_b && val_c && z && out_za && out_zb && out_zc &&
                         row_ptr->kind == COZK_SCALAR_U32 && col->kind == COZK_SCALAR_U32 && val_a->kind == COZK_SCALAR_FR &&
                         val_b->kind == COZK_SCALAR_FR && val_c->kind == COZK_SCALAR_FR && row_ptr->n >= 2 &&
                         val_a->n == col->n && val_b->n == col->n && val_c->n == col->n,
                     "sparse_matvec3: bad argument");
        size_t nrows = row_ptr->n - 1;
        cozk_poly* o[3];
        for (int i = 0; i < 3; i++) {
            o[i] = new cozk_poly();
            o[i]->ctx = ctx;
            o[i]->mode = z->mode;
            o[i]->len = o[i]->orig_len = nrows;
            o[i]->cur = -1;
            o[i]->own0 = true;
            o[i]->a0 = dev_alloc_fe(nrows);
            o[i]->b0 = z->mode == COZK_MODE_REP3 ? dev_alloc_fe(nrows) : nullptr;
        }
        // device-side queues of the long rows and of their SPMV_CHUNK-entry work items, and the items' partial sums
        const size_t nnz = col->n;
        const size_t max_rows = nnz / SPMV_LONG + 1, max_items = nnz / SPMV_CHUNK + max_rows;
        const int nc = z->mode == COZK_MODE_REP3 ? 2 : 1;
        const size_t q_words = 16 + 3 * max_rows + 2 * max_items;
        const size_t q_bytes = (q_words * sizeof(uint32_t) + 63) & ~(size_t)63;
        ctx->scratch.reserve(q_bytes + max_items * 3 * nc * sizeof(fe));
        uint32_t* q = ctx->scratch.as<uint32_t>();
        uint32_t* q_rows = q + 16;
        uint32_t* q_items = q_rows + 3 * max_rows;
        fe* partial = reinterpret_cast<fe*>(reinterpret_cast<char*>(ctx->scratch.p) + q_bytes);
        HIP_TRY(hipMemsetAsync(q, 0, 2 * sizeof(uint32_t), ctx->stream));
        const uint32_t* rp = (const uint32_t*)row_ptr->d;
        const uint32_t* ci = (const uint32_t*)col->d;
        const fe *pa = (const fe*)val_a->d, *pb = (const fe*)val_b->d, *pc = (const fe*)val_c->d;
        unsigned g_items = (unsigned)std::min<size_t>(max_items, 4096), g_rows = (unsigned)std::min<size_t>(max_rows, 1024);
        if (z->mode == COZK_MODE_REP3) {
            k_sparse_matvec3<2><<<grid_for(nrows), PT, 0, ctx->stream>>>(rp, ci, pa, pb, pc, poly_a(z), poly_b(z), nrows, o[0]->a0, o[0]->b0, o[1]->a0,
                                                                        o[1]->b0, o[2]->a0, o[2]->b0, q, q_rows, q_items);
            k_sparse_matvec3_items<2><<<g_items, PT, 0, ctx->stream>>>(rp, ci, pa, pb, pc, poly_a(z), poly_b(z), q, q_items, partial);
            k_sparse_matvec3_rows<2><<<g_rows, PT, 0, ctx->stream>>>(q, q_rows, partial, o[0]->a0, o[0]->b0, o[1]->a0, o[1]->b0, o[2]->a0, o[2]->b0);
        } else {
            k_sparse_matvec3<1><<<grid_for(nrows), PT, 0, ctx->stream>>>(rp, ci, pa, pb, pc, poly_a(z), nullptr, nrows, o[0]->a0, nullptr, o[1]->a0, nullptr,
                                                                        o[2]->a0, nullptr, q, q_rows, q_items);
            k_sparse_matvec3_items<1><<<g_items, PT, 0, ctx->stream>>>(rp, ci, pa, pb, pc, poly_a(z), nullptr, q, q_items, partial);
            k_sparse_matvec3_rows<1><<<g_rows, PT, 0, ctx->stream>>>(q, q_rows, partial, o[0]->a0, nullptr, o[1]->a0, nullptr, o[2]->a0, nullptr);
        }
        HIP_TRY(hipGetLastError());
        *out_za = o[0];
        *out_zb = o[1];
        *out_zc = o[2];
    });
}

// PST13 `open` fold step (pst13.rs:445-459): r (len 2h) -> q (len h), r' (len h)
int cozk_pst_fold(cozk_ctx* ctx, const cozk_vec* r, const uint64_t p[4], cozk_vec* q, cozk_vec* r_next) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && r && p && q && r_next && r->kind == COZK_SCALAR_FR && r->n >= 2 && r->n % 2 == 0 &&
                         q->n >= r->n / 2 && r_next->n >= r->n / 2,
                     "pst_fold: bad argument");
        size_t h = r->n / 2;
        k_pst_fold<<<grid_for(h), PT, 0, ctx->stream>>>((const fe*)r->d, (fe*)q->d, (fe*)r_next->d, h, fe_from_u64x4(p));
        HIP_TRY(hipGetLastError());
    });
}

// ------------------------------------------------------------------ C ABI: interleaved GKR layer
int cozk_layer_create(cozk_ctx* ctx, int mode, const cozk_vec* a, const cozk_vec* b, int take_ownership, cozk_layer** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && out && a && a->kind == COZK_SCALAR_FR && a->n % 2 == 0 && (mode == COZK_MODE_PLAIN || mode == COZK_MODE_REP3),
                     "layer_create: bad argument (coeffs.len() % 2 == 0, dense_interleaved_poly.rs:62)");
        COZK_REQUIRE(mode == COZK_MODE_PLAIN || (b && b->kind == COZK_SCALAR_FR && b->n == a->n), "layer_create: share b missing");
        cozk_layer* l = new cozk_layer();
        l->ctx = ctx;
        l->mode = mode;
        l->len = a->n;
        l->cur = 0;
        l->cap[0] = a->n;
        if (take_ownership) {
            COZK_REQUIRE(a->owned && (mode == COZK_MODE_PLAIN || b->owned), "layer_create: cannot take ownership of a view");
            l->buf[0][0] = (fe*)a->d;
            const_cast<cozk_vec*>(a)->owned = false;
            if (mode == COZK_MODE_REP3) {
                l->buf[0][1] = (fe*)b->d;
                const_cast<cozk_vec*>(b)->owned = false;
            }
        } else {
            l->buf[0][0] = dev_alloc_fe(a->n);
            HIP_TRY(hipMemcpyAsync(l->buf[0][0], a->d, a->n * sizeof(fe), hipMemcpyDeviceToDevice, ctx->stream));
            if (mode == COZK_MODE_REP3) {
                l->buf[0][1] = dev_alloc_fe(a->n);
                HIP_TRY(hipMemcpyAsync(l->buf[0][1], b->d, a->n * sizeof(fe), hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        *out = l;
    });
}

int cozk_layer_free(cozk_layer* l) {
    if (!l) return COZK_OK;
    for (int w = 0; w < 2; w++)
        for (int c = 0; c < 2; c++)
            if (l->buf[w][c]) ctx_dev_free(l->ctx, l->buf[w][c]);
    delete l;
    return COZK_OK;
}

size_t cozk_layer_len(const cozk_layer* l) { return l ? l->len : 0; }

int cozk_layer_download(cozk_ctx* ctx, const cozk_layer* l, uint64_t* a, uint64_t* b) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && l && a, "layer_download: bad argument");
        HIP_TRY(hipMemcpyAsync(a, l->buf[l->cur][0], l->len * sizeof(fe), hipMemcpyDeviceToHost, ctx->stream));
        if (l->mode == COZK_MODE_REP3 && b)
            HIP_TRY(hipMemcpyAsync(b, l->buf[l->cur][1], l->len * sizeof(fe), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    });
}

// deep copy (the GKR prover binds layers destructively; tests and the harness keep the originals)
int cozk_layer_clone(cozk_ctx* ctx, const cozk_layer* src, cozk_layer** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && src && out, "layer_clone: bad argument");
        cozk_layer* l = new cozk_layer();
        l->ctx = ctx;
        l->mode = src->mode;
        l->len = src->len;
        l->cur = 0;
        l->cap[0] = src->len;
        for (int c = 0; c < (src->mode == COZK_MODE_REP3 ? 2 : 1); c++) {
            l->buf[0][c] = dev_alloc_fe(src->len);
            HIP_TRY(hipMemcpyAsync(l->buf[0][c], src->buf[src->cur][c], src->len * sizeof(fe), hipMemcpyDeviceToDevice, ctx->stream));
        }
        *out = l;
    });
}

int cozk_layer_as_poly(cozk_ctx* ctx, const cozk_layer* l, cozk_poly** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && l && out, "layer_as_poly: bad argument");
        cozk_poly* p = new cozk_poly();
        p->ctx = ctx;
        p->mode = l->mode;
        p->len = p->orig_len = l->len;
        p->cur = -1;
        p->own0 = false;
        p->a0 = l->buf[l->cur][0];
        p->b0 = l->buf[l->cur][1];
        *out = p;
    });
}

// Rep3Bindable::bind (dense_interleaved_poly.rs:155-195)
int cozk_layer_bind(cozk_ctx* ctx, cozk_layer* l, const uint64_t r[4]) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && l && r && l->len >= 2, "layer_bind: bad argument");
        size_t nch = (l->len + 3) / 4;
        size_t nout = 2 * nch;
        int dst = 1 - l->cur;
        if (l->cap[dst] < nout) {
            for (int c = 0; c < 2; c++) {
                if (l->buf[dst][c]) ctx_dev_free(l->ctx, l->buf[dst][c]);
                l->buf[dst][c] = nullptr;
            }
            l->buf[dst][0] = dev_alloc_fe(nout);
            if (l->mode == COZK_MODE_REP3) l->buf[dst][1] = dev_alloc_fe(nout);
            l->cap[dst] = nout;
        }
        fe rr = fe_from_u64x4(r);
        if (l->mode == COZK_MODE_REP3)
            k_layer_bind<2><<<grid_for(nch), PT, 0, ctx->stream>>>(l->buf[l->cur][0], l->buf[l->cur][1], l->buf[dst][0], l->buf[dst][1], l->len, rr);
        else
            k_layer_bind<1><<<grid_for(nch), PT, 0, ctx->stream>>>(l->buf[l->cur][0], nullptr, l->buf[dst][0], nullptr, l->len, rr);
        HIP_TRY(hipGetLastError());
        l->cur = dst;
        l->len = nout;
    });
}

// SplitEqPolynomial::new(w) (jolt-core; m = len/2, E2 = evals(w[..m]), E1 = evals(w[m..]))
int cozk_spliteq_new(cozk_ctx* ctx, const uint64_t* w, int nv, cozk_spliteq** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && out && nv >= 0 && nv < 60 && (w || nv == 0), "spliteq_new: bad argument");
        int m = nv / 2;
        cozk_spliteq* e = new cozk_spliteq();
        e->ctx = ctx;
        e->num_vars = nv;
        e->E2_len = e->E2_cap = (size_t)1 << m;
        e->E1_len = e->E1_cap = (size_t)1 << (nv - m);
        for (int k = 0; k < 2; k++) {
            e->E1[k] = dev_alloc_fe(e->E1_cap);
            e->E2[k] = dev_alloc_fe(e->E2_cap);
        }
        std::vector<fe> rr(nv);
        for (int j = 0; j < nv; j++) rr[j] = fe_from_u64x4(w + 4 * j);
        eq_evals_device(ctx, rr.data(), m, e->E2[0], e->E2[1]);
        eq_evals_device(ctx, rr.data() + m, nv - m, e->E1[0], e->E1[1]);
        e->c1 = e->c2 = 0;
        *out = e;
    });
}

int cozk_spliteq_free(cozk_spliteq* e) {
    if (!e) return COZK_OK;
    for (int k = 0; k < 2; k++) {
        if (e->E1[k]) ctx_dev_free(e->ctx, e->E1[k]);
        if (e->E2[k]) ctx_dev_free(e->ctx, e->E2[k]);
    }
    delete e;
    return COZK_OK;
}

int cozk_spliteq_lens(const cozk_spliteq* e, size_t* e1_len, size_t* e2_len) {
    if (!e) return COZK_ERR_INVALID_ARG;
    if (e1_len) *e1_len = e->E1_len;
    if (e2_len) *e2_len = e->E2_len;
    return COZK_OK;
}

// SplitEqPolynomial::bind(r)
int cozk_spliteq_bind(cozk_ctx* ctx, cozk_spliteq* e, const uint64_t r[4]) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && e && r, "spliteq_bind: bad argument");
        fe rr = fe_from_u64x4(r);
        if (e->E1_len == 1) {
            COZK_REQUIRE(e->E2_len >= 2, "spliteq_bind: polynomial already fully bound");
            size_t n = e->E2_len / 2;
            k_fold_pairs<<<grid_for(n), PT, 0, ctx->stream>>>(e->E2[e->c2], e->E2[1 - e->c2], n, rr);
            e->c2 = 1 - e->c2;
            e->E2_len = n;
        } else {
            size_t n = e->E1_len / 2;
            k_fold_pairs<<<grid_for(n), PT, 0, ctx->stream>>>(e->E1[e->c1], e->E1[1 - e->c1], n, rr);
            e->c1 = 1 - e->c1;
            e->E1_len = n;
            if (n == 1) k_scale_by_first<<<grid_for(e->E2_len), PT, 0, ctx->stream>>>(e->E2[e->c2], e->E2_len, e->E1[e->c1]);
        }
        HIP_TRY(hipGetLastError());
    });
}

// Rep3BatchedCubicSumcheckWorker::compute_cubic (dense_interleaved_poly.rs:210-365): returns the 4
// additive coefficient shares of the round polynomial through evals [g0, claim - g0, g2, g3]
static void layer_cubic_sums(cozk_ctx* ctx, const cozk_layer* l, const cozk_spliteq* eq, fe s[3]) {
    size_t nch = (l->len + 3) / 4;
    unsigned gx = grid_capped(nch);
    if (gx > 1024) gx = 1024;
    ctx->scratch.reserve((3 * (size_t)MAXBLK + 3) * sizeof(fe));
    fe* partial = ctx->scratch.as<fe>();
    fe* res = result_slot(ctx, 3);
    const fe* a = l->buf[l->cur][0];
    const fe* b = l->buf[l->cur][1];
    const fe* E1 = eq->E1[eq->c1];
    const fe* E2 = eq->E2[eq->c2];
    bool nested = eq->E1_len != 1;
    static const int f9_env = getenv("COZK_LAYER_F9") ? atoi(getenv("COZK_LAYER_F9")) : 1;
    if (f9_env != 0 && (l->len + 3) / 4 >= 1024) {  // the 9 x 29 multiplier kernels (fr9.hip.hpp) for throughput-bound layers
        const size_t need = (nch + PT - 1) / PT;
        const int dev = ctx->device;
#define COZK_CUBIC9(NC_, NE_, B_, E1H_)                                                                                             \
    do {                                                                                                                            \
        gx = resident_grid((const void*)k_layer_cubic9<NC_, NE_>, need, dev);                                                        \
        k_layer_cubic9<NC_, NE_><<<gx, PT, 0, ctx->stream>>>(a, B_, l->len, E1, E1H_, E2, eq->E2_len, partial);                      \
    } while (0)
        static const bool group_env = !(getenv("COZK_LAYER_GROUPED") && atoi(getenv("COZK_LAYER_GROUPED")) == 0);
        const bool grouped = nested && group_env && eq->E1_len / 2 >= 512;  // split-eq inner sums (layer9_terms, NESTED = 2)
        if (l->mode == COZK_MODE_REP3) {  // (the Rep3 kernels lose with the three group accumulators: 0.58 vs 0.52 ms, register pressure)
            if (nested) COZK_CUBIC9(2, 1, b, eq->E1_len / 2);
            else COZK_CUBIC9(2, 0, b, 0);
        } else {
            if (grouped) COZK_CUBIC9(1, 2, b, eq->E1_len / 2);
            else if (nested) COZK_CUBIC9(1, 1, b, eq->E1_len / 2);
            else COZK_CUBIC9(1, 0, b, 0);
        }
#undef COZK_CUBIC9
    } else if (l->mode == COZK_MODE_REP3) {
        if (nested) k_layer_cubic<2, 1><<<gx, PT, 0, ctx->stream>>>(a, b, l->len, E1, eq->E1_len / 2, E2, eq->E2_len, partial);
        else k_layer_cubic<2, 0><<<gx, PT, 0, ctx->stream>>>(a, b, l->len, E1, 0, E2, eq->E2_len, partial);
    } else {
        if (nested) k_layer_cubic<1, 1><<<gx, PT, 0, ctx->stream>>>(a, b, l->len, E1, eq->E1_len / 2, E2, eq->E2_len, partial);
        else k_layer_cubic<1, 0><<<gx, PT, 0, ctx->stream>>>(a, b, l->len, E1, 0, E2, eq->E2_len, partial);
    }
    k_finish_sums<<<3, PT, 0, ctx->stream>>>(partial, gx, Fr::one(), 0, res, arm_round_publish(ctx));
    HIP_TRY(hipGetLastError());
    fetch_fe(ctx, res, 3, s);
}

int cozk_layer_compute_cubic(cozk_ctx* ctx, const cozk_layer* l, const cozk_spliteq* eq, const uint64_t prev_claim[4],
                             uint64_t out_coeffs[16]) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && l && eq && prev_claim && out_coeffs, "compute_cubic: bad argument");
        fe s[3];
        layer_cubic_sums(ctx, l, eq, s);
        fe ev[4] = {s[0], Fr::sub(fe_from_u64x4(prev_claim), s[0]), s[1], s[2]};
        fe cf[4];
        unipoly_from_evals(ev, 4, cf);
        for (int i = 0; i < 4; i++) fe_to_u64x4(cf[i], out_coeffs + 4 * i);
    });
}

// One round of prove_sumcheck (co-jolt/src/subprotocols/sumcheck.rs:107-122) per call: bind the layer and the
// split-eq polynomial with the previous round's challenge `r` (NULL in the first round), then compute_cubic.
// Layers of <= ROUND_SMALL_MAX elements take the single-launch kernel; larger ones the separate kernels.
int cozk_layer_round(cozk_ctx* ctx, cozk_layer* l, cozk_spliteq* e, const uint64_t* r, const uint64_t prev_claim[4],
                     uint64_t out_coeffs[16]) {
    if (!ctx || !l || !e || !prev_claim || !out_coeffs) return COZK_ERR_INVALID_ARG;
    if (l->len > ROUND_SMALL_MAX) {
        if (!r) return cozk_layer_compute_cubic(ctx, l, e, prev_claim, out_coeffs);
        // large layer: fold the eq tables, then bind and take the next round's sums in one pass (k_layer_bind_cubic)
        int rc = cozk_spliteq_bind(ctx, e, r);
        if (rc != COZK_OK) return rc;
        return cozk_guard(ctx, [&] {
            size_t nch_in = (l->len + 3) / 4;
            size_t nout = 2 * nch_in;
            int dst = 1 - l->cur;
            if (l->cap[dst] < nout) {
                for (int c = 0; c < 2; c++) {
                    if (l->buf[dst][c]) ctx_dev_free(l->ctx, l->buf[dst][c]);
                    l->buf[dst][c] = nullptr;
                }
                l->buf[dst][0] = dev_alloc_fe(nout);
                if (l->mode == COZK_MODE_REP3) l->buf[dst][1] = dev_alloc_fe(nout);
                l->cap[dst] = nout;
            }
            size_t nch_out = (nout + 3) / 4;
            unsigned gx = grid_capped(nch_out);
            if (gx > 1024) gx = 1024;
            ctx->scratch.reserve((3 * (size_t)MAXBLK + 3) * sizeof(fe));
            fe* partial = ctx->scratch.as<fe>();
            fe* res = result_slot(ctx, 3);
            fe rr = fe_from_u64x4(r);
            const fe *ia = l->buf[l->cur][0], *ib = l->buf[l->cur][1];
            fe *oa = l->buf[dst][0], *ob = l->buf[dst][1];
            const fe* E1 = e->E1[e->c1];
            const fe* E2 = e->E2[e->c2];
            bool nested = e->E1_len != 1;
            {
            // algorithmic bytes: the layer read once and its bound half written once (SURVEY 8d K3 + K4 fused), + the eq tables
            const uint64_t S = l->mode == COZK_MODE_REP3 ? 64 : 32;
            ProfScope prof(ctx, COZK_PROF_BIND_CUBIC, (uint64_t)l->len * S + (uint64_t)nout * S + (uint64_t)(e->E1_len + e->E2_len) * 32);
            // 9 x 29 multiplier kernels for layers large enough to be throughput-bound (COZK_LAYER_F9=0: the saturated kernels)
            static const int f9_env = getenv("COZK_LAYER_F9") ? atoi(getenv("COZK_LAYER_F9")) : 1;
            if (f9_env != 0 && nch_out >= 1024) {
                fe r5 = rr;
                for (int d = 0; d < 5; d++) r5 = Fr::dbl(r5);  // the challenge times 2^5 = 1 / lambda (fr9.hip.hpp)
                const size_t need = (nch_out + PT - 1) / PT;
                const int dev = ctx->device;
#define COZK_BIND_CUBIC9(NC_, NE_, IB_, OB_, E1H_)                                                                                  \
    do {                                                                                                                            \
        gx = resident_grid((const void*)k_layer_bind_cubic9<NC_, NE_>, need, dev);                                                   \
        k_layer_bind_cubic9<NC_, NE_><<<gx, PT, 0, ctx->stream>>>(ia, IB_, oa, OB_, l->len, r5, E1, E1H_, E2, e->E2_len, partial);   \
    } while (0)
                static const bool group_env = !(getenv("COZK_LAYER_GROUPED") && atoi(getenv("COZK_LAYER_GROUPED")) == 0);
                const bool grouped = nested && group_env && e->E1_len / 2 >= 512;  // split-eq inner sums (layer9_terms, NESTED = 2)
                if (l->mode == COZK_MODE_REP3) {  // (plain only: the Rep3 kernels lose with the group accumulators)
                    if (nested) COZK_BIND_CUBIC9(2, 1, ib, ob, e->E1_len / 2);
                    else COZK_BIND_CUBIC9(2, 0, ib, ob, 0);
                } else {
                    if (grouped) COZK_BIND_CUBIC9(1, 2, nullptr, nullptr, e->E1_len / 2);
                    else if (nested) COZK_BIND_CUBIC9(1, 1, nullptr, nullptr, e->E1_len / 2);
                    else COZK_BIND_CUBIC9(1, 0, nullptr, nullptr, 0);
                }
#undef COZK_BIND_CUBIC9
            } else if (l->mode == COZK_MODE_REP3) {
                if (nested) k_layer_bind_cubic<2, 1><<<gx, PT, 0, ctx->stream>>>(ia, ib, oa, ob, l->len, rr, E1, e->E1_len / 2, E2, e->E2_len, partial);
                else k_layer_bind_cubic<2, 0><<<gx, PT, 0, ctx->stream>>>(ia, ib, oa, ob, l->len, rr, E1, 0, E2, e->E2_len, partial);
            } else {
                if (nested) k_layer_bind_cubic<1, 1><<<gx, PT, 0, ctx->stream>>>(ia, nullptr, oa, nullptr, l->len, rr, E1, e->E1_len / 2, E2, e->E2_len, partial);
                else k_layer_bind_cubic<1, 0><<<gx, PT, 0, ctx->stream>>>(ia, nullptr, oa, nullptr, l->len, rr, E1, 0, E2, e->E2_len, partial);
            }
            }
            k_finish_sums<<<3, PT, 0, ctx->stream>>>(partial, gx, Fr::one(), 0, res, arm_round_publish(ctx));
            HIP_TRY(hipGetLastError());
            l->cur = dst;
            l->len = nout;
            fe sres[3];
            fetch_fe(ctx, res, 3, sres);
            fe ev[4] = {sres[0], Fr::sub(fe_from_u64x4(prev_claim), sres[0]), sres[1], sres[2]};
            fe cf[4];
            unipoly_from_evals(ev, 4, cf);
            for (int i = 0; i < 4; i++) fe_to_u64x4(cf[i], out_coeffs + 4 * i);
        });
    }
    return cozk_guard(ctx, [&] {
        const fe *ia = l->buf[l->cur][0], *ib = l->buf[l->cur][1];
        fe *oa = nullptr, *ob = nullptr;
        size_t len_in = l->len;
        const fe* fold_in = nullptr;
        fe* fold_out = nullptr;
        size_t fold_n = 0, scale_n = 0;
        fe* scale_vec = nullptr;
        fe rr = Fr::zero();
        if (r) {
            COZK_REQUIRE(l->len >= 2, "layer_round: layer already fully bound");
            rr = fe_from_u64x4(r);
            size_t nout = 2 * ((l->len + 3) / 4);
            int dst = 1 - l->cur;
            if (l->cap[dst] < nout) {
                for (int c = 0; c < 2; c++) {
                    if (l->buf[dst][c]) ctx_dev_free(l->ctx, l->buf[dst][c]);
                    l->buf[dst][c] = nullptr;
                }
                l->buf[dst][0] = dev_alloc_fe(nout);
                if (l->mode == COZK_MODE_REP3) l->buf[dst][1] = dev_alloc_fe(nout);
                l->cap[dst] = nout;
            }
            oa = l->buf[dst][0];
            ob = l->buf[dst][1];
            l->cur = dst;
            l->len = nout;
            // SplitEqPolynomial::bind bookkeeping (cozk_spliteq_bind)
            if (e->E1_len == 1) {
                COZK_REQUIRE(e->E2_len >= 2, "layer_round: eq polynomial already fully bound");
                fold_n = e->E2_len / 2;
                fold_in = e->E2[e->c2];
                fold_out = e->E2[1 - e->c2];
                e->c2 = 1 - e->c2;
                e->E2_len = fold_n;
            } else {
                fold_n = e->E1_len / 2;
                fold_in = e->E1[e->c1];
                fold_out = e->E1[1 - e->c1];
                e->c1 = 1 - e->c1;
                e->E1_len = fold_n;
                if (fold_n == 1) {
                    scale_vec = e->E2[e->c2];
                    scale_n = e->E2_len;
                }
            }
        }
        const fe* ca = l->buf[l->cur][0];
        const fe* cb = l->buf[l->cur][1];
        const fe* E1 = e->E1[e->c1];
        const fe* E2 = e->E2[e->c2];
        int nested = e->E1_len != 1;
        fe* res = result_slot(ctx, 3);
        if (l->mode == COZK_MODE_REP3)
            k_layer_round_small<2><<<1, RT, 0, ctx->stream>>>(ia, ib, oa, ob, len_in, r != nullptr, rr, fold_in, fold_out, fold_n, scale_vec, scale_n, ca, cb,
                                                             l->len, E1, e->E1_len / 2, E2, e->E2_len, nested, res);
        else
            k_layer_round_small<1><<<1, RT, 0, ctx->stream>>>(ia, nullptr, oa, nullptr, len_in, r != nullptr, rr, fold_in, fold_out, fold_n, scale_vec, scale_n,
                                                             ca, nullptr, l->len, E1, e->E1_len / 2, E2, e->E2_len, nested, res);
        HIP_TRY(hipGetLastError());
        fe sres[3];
        fetch_fe(ctx, res, 3, sres);
        fe ev[4] = {sres[0], Fr::sub(fe_from_u64x4(prev_claim), sres[0]), sres[1], sres[2]};
        fe cf[4];
        unipoly_from_evals(ev, 4, cf);
        for (int i = 0; i < 4; i++) fe_to_u64x4(cf[i], out_coeffs + 4 * i);
    });
}

// The whole round loop of Rep3BatchedCubicSumcheckWorker::prove_sumcheck (co-jolt/src/subprotocols/sumcheck.rs:
// 96-131) for one layer: per round compute_cubic -> `cb` (the host sends the coefficients, receives the challenge
// and the next claim) -> bind, then final_claims.  While the layer is large each round is its own launch
// (cozk_layer_round); once it is down to ROUND_PERSIST_MAX elements the remaining rounds, the last bind and the
// final claims run inside ONE resident kernel that exchanges sums and challenges with this thread through a
// pinned mailbox (k_layer_rounds_persistent).  cb returns 0 on success; `next_claim` must already be this party's
// additive share.  out_r = num_rounds x 4; final_claims = left.a, left.b, right.a, right.b.
int cozk_layer_prove_rounds(cozk_ctx* ctx, cozk_layer* l, cozk_spliteq* e, const uint64_t claim[4], int num_rounds, cozk_round_cb cb,
                            void* user, uint64_t* out_r, uint64_t final_claims[16]) {
    if (!ctx || !l || !e || !claim || !cb || !final_claims || num_rounds < 0 || (num_rounds > 0 && !out_r)) return COZK_ERR_INVALID_ARG;
    static const bool no_persist = getenv("COZK_NO_PERSIST") != nullptr;
    uint64_t pc[4], rr[4], coeffs[16], nc[4];
    memcpy(pc, claim, sizeof pc);
    bool have_r = false;
    int round = 0;
    // the resident kernel is used when the context allows it (cozk_ctx_set_resident_rounds; by default only while this
    // is the one live context on its device in the process) and is abandoned for the rest of the call once its watchdog
    // has fired (a round callback that took longer than COZK_RESIDENT_TIMEOUT_S): the loop then goes on with one launch
    // per round, from exactly the state the kernel left
    bool allow_resident = !no_persist && ctx_resident_rounds_enabled(ctx);
    for (;;) {
        for (; round < num_rounds; round++) {
            if (allow_resident && l->len <= ROUND_PERSIST_MAX && l->len >= 2) break;
            int rc = cozk_layer_round(ctx, l, e, have_r ? rr : nullptr, pc, coeffs);
            if (rc != COZK_OK) return rc;
            if (cb(user, round, coeffs, rr, nc) != 0) {
                ctx->last_error = "layer_prove_rounds: round callback failed";
                return COZK_ERR_INTERNAL;
            }
            memcpy(out_r + 4 * round, rr, sizeof rr);
            memcpy(pc, nc, sizeof pc);
            have_r = true;
        }
        if (round == num_rounds) {  // every remaining round ran as its own launch
            if (have_r) {
                int rc = cozk_layer_bind(ctx, l, rr);
                if (rc != COZK_OK) return rc;
                rc = cozk_spliteq_bind(ctx, e, rr);
                if (rc != COZK_OK) return rc;
            }
            return cozk_layer_final_claims(ctx, l, final_claims);
        }
        int rounds_done = -1;  // >= 0: the resident kernel timed out after that many of its rounds; fall back
        int rc = cozk_guard(ctx, [&] {
            const int nrem = num_rounds - round;
            // both ping-pong buffers of the layer must hold the current length
            for (int w = 0; w < 2; w++) {
                if (w != l->cur && l->cap[w] < l->len) {
                    for (int c = 0; c < 2; c++) {
                        if (l->buf[w][c]) ctx_dev_free(l->ctx, l->buf[w][c]);
                        l->buf[w][c] = nullptr;
                    }
                    l->buf[w][0] = dev_alloc_fe(l->len);
                    if (l->mode == COZK_MODE_REP3) l->buf[w][1] = dev_alloc_fe(l->len);
                    l->cap[w] = l->len;
                }
            }
            if (!ctx->mailbox) HIP_TRY(hipHostMalloc(&ctx->mailbox, sizeof(RoundMailbox), hipHostMallocMapped | hipHostMallocCoherent));
            RoundMailbox* mb = (RoundMailbox*)ctx->mailbox;
            memset(mb, 0, sizeof *mb);
            std::atomic_thread_fence(std::memory_order_seq_cst);
            fe r_first = have_r ? fe_from_u64x4(rr) : Fr::zero();
            const long long ticks = (long long)resident_timeout_s() * MB_TICKS_PER_S;
            static const bool trace = getenv("COZK_TRACE_ROUNDS") != nullptr;  // the traced variant reads the 100 MHz clock four times per round
#define COZK_RESIDENT(NC_, TR_, LB0_, LB1_)                                                                                                          \
    k_layer_rounds_persistent<NC_, TR_><<<1, RT, 0, ctx->stream>>>(l->buf[0][0], LB0_, l->buf[1][0], LB1_, l->cur, l->len, e->E1[0], e->E1[1], e->c1, \
                                                                   e->E1_len, e->E2[0], e->E2[1], e->c2, e->E2_len, nrem, have_r ? 1 : 0, r_first, mb, ticks)
            if (l->mode == COZK_MODE_REP3) {
                if (trace) COZK_RESIDENT(2, 1, l->buf[0][1], l->buf[1][1]);
                else COZK_RESIDENT(2, 0, l->buf[0][1], l->buf[1][1]);
            } else {
                if (trace) COZK_RESIDENT(1, 1, nullptr, nullptr);
                else COZK_RESIDENT(1, 0, nullptr, nullptr);
            }
#undef COZK_RESIDENT
            HIP_TRY(hipGetLastError());
            volatile uint32_t* res_seq = &mb->res_seq;
            volatile uint32_t* status = &mb->status;
            auto give_up = [&](const char* why) {
                *(volatile uint32_t*)&mb->cmd[12] = 1;
                std::atomic_thread_fence(std::memory_order_seq_cst);
                (void)hipStreamSynchronize(ctx->stream);
                throw CozkError(COZK_ERR_INTERNAL, why);
            };
            // false: the kernel's watchdog fired (it waited resident_timeout_s() for the host's challenge and left)
            auto wait_result = [&](uint32_t want) -> bool {
                auto t0 = std::chrono::steady_clock::now();
                uint64_t spins = 0;
                while (*res_seq != want) {
                    if (*status == 1) return false;
                    if (*status) give_up("layer_prove_rounds: the resident kernel stopped");
                    __builtin_ia32_pause();
                    if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(resident_timeout_s() + 5))
                        give_up("layer_prove_rounds: no result from the resident kernel");
                }
                std::atomic_thread_fence(std::memory_order_acquire);
                return true;
            };
            // mirror the kernel's bookkeeping: one bind per completed round (+ the pending one it started with)
            auto mirror_binds = [&](int binds) {
                for (int b = 0; b < binds; b++) {
                    l->len = 2 * ((l->len + 3) / 4);
                    l->cur = 1 - l->cur;
                    if (e->E1_len == 1) {
                        e->E2_len /= 2;
                        e->c2 = 1 - e->c2;
                    } else {
                        e->E1_len /= 2;
                        e->c1 = 1 - e->c1;
                    }
                }
            };
            const int bind_first = have_r ? 1 : 0;
            int j = 0;
            bool alive = true;
            for (; j < nrem; j++) {
                if (!(alive = wait_result((uint32_t)j + 1))) break;
                fe s0 = mb->res[0], s2 = mb->res[1], s3 = mb->res[2];
                fe ev[4] = {s0, Fr::sub(fe_from_u64x4(pc), s0), s2, s3};
                fe cf[4];
                unipoly_from_evals(ev, 4, cf);
                for (int i = 0; i < 4; i++) fe_to_u64x4(cf[i], coeffs + 4 * i);
                if (cb(user, round + j, coeffs, rr, nc) != 0) give_up("layer_prove_rounds: round callback failed");
                memcpy(out_r + 4 * (round + j), rr, sizeof rr);
                memcpy(pc, nc, sizeof pc);
                have_r = true;
                {
                    // three self-validating pieces: payload first, then the half that carries the tag (x86 stores stay in order)
                    const fe rv = fe_from_u64x4(rr);
                    const uint32_t tag = (uint32_t)j + 1;
                    const uint32_t pay[9] = {rv.l[0], rv.l[1], rv.l[2], rv.l[3], rv.l[4], rv.l[5], rv.l[6], rv.l[7], 0u};
                    for (int k = 0; k < 3; k++) {
                        volatile uint64_t* piece = reinterpret_cast<volatile uint64_t*>(&mb->cmd[4 * k]);
                        piece[0] = (uint64_t)pay[3 * k] | ((uint64_t)pay[3 * k + 1] << 32);
                        std::atomic_thread_fence(std::memory_order_release);
                        piece[1] = (uint64_t)pay[3 * k + 2] | ((uint64_t)tag << 32);
                    }
                }
            }
            if (alive) alive = wait_result((uint32_t)nrem + 1);
            if (!alive) {
                // the kernel left while the host was inside the callback of round j - 1 (or before the first result):
                // it published j results and bound j - 1 (+ bind_first) times; the challenge of round j - 1 is pending
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                mirror_binds(j > 0 ? j - 1 + bind_first : 0);
                rounds_done = j;
                return;
            }
            for (int k = 0; k < 4; k++) fe_to_u64x4(mb->res[k], final_claims + 4 * k);
            if (getenv("COZK_TRACE_ROUNDS"))
                fprintf(stderr, "[mailbox] %d rounds: device us/round: wait %.1f bind %.1f cubic %.1f publish %.1f\n", nrem,
                        mb->dbg[0] / 100.0 / (nrem + 1), mb->dbg[1] / 100.0 / (nrem + 1), mb->dbg[2] / 100.0 / nrem, mb->dbg[3] / 100.0 / nrem);
            mirror_binds(nrem + bind_first);
        });
        if (rc != COZK_OK || rounds_done < 0) return rc;
        round += rounds_done;
        allow_resident = false;
    }
}

// cozk_layer_prove_rounds for a worker sub-net, which cannot derive g(1) from the (global) previous claim and sends
// the raw sums g(0), g(2), g(3) instead (as cozk_layer_compute_cubic_evals).  Same machinery: the round polynomial
// through (g0, -g0, g2, g3) -- i.e. previous claim 0 -- is evaluated back at 0, 2, 3 for the callback.
namespace {
struct EvalsShim {
    cozk_round_evals_cb cb;
    void* user;
};
int evals_shim_cb(void* user, int round, const uint64_t coeffs[16], uint64_t r_out[4], uint64_t next_claim_out[4]) {
    EvalsShim* sh = static_cast<EvalsShim*>(user);
    std::vector<fe> cf(4);
    for (int i = 0; i < 4; i++) cf[i] = fe_from_u64x4(coeffs + 4 * i);
    uint64_t ev[12];
    fe_to_u64x4(cf[0], ev);
    auto horner = [&](const fe& x) { return Fr::add(cf[0], Fr::mul(x, Fr::add(cf[1], Fr::mul(x, Fr::add(cf[2], Fr::mul(x, cf[3])))))); };
    fe_to_u64x4(horner(Fr::from_u64(2)), ev + 4);
    fe_to_u64x4(horner(Fr::from_u64(3)), ev + 8);
    for (int i = 0; i < 4; i++) next_claim_out[i] = 0;
    return sh->cb(sh->user, round, ev, r_out);
}
}  // namespace
int cozk_layer_prove_rounds_evals(cozk_ctx* ctx, cozk_layer* l, cozk_spliteq* e, int num_rounds, cozk_round_evals_cb cb, void* user,
                                  uint64_t* out_r, uint64_t final_claims[16]) {
    if (!cb) return COZK_ERR_INVALID_ARG;
    EvalsShim sh{cb, user};
    const uint64_t zero[4] = {0, 0, 0, 0};
    return cozk_layer_prove_rounds(ctx, l, e, zero, num_rounds, evals_shim_cb, &sh, out_r, final_claims);
}

// the raw additive sums g(0), g(2), g(3) of compute_cubic, for a worker sub-net that cannot derive g(1) from
// the (global) previous claim: the coordinator inserts claim - g(0), as the reference does for the primary
// sumcheck (jolt/vm/instruction_lookups/worker.rs:593-597, coordinator.rs:131-132)
int cozk_layer_compute_cubic_evals(cozk_ctx* ctx, const cozk_layer* l, const cozk_spliteq* eq, uint64_t out_evals[12]) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && l && eq && out_evals, "compute_cubic_evals: bad argument");
        fe s[3];
        layer_cubic_sums(ctx, l, eq, s);
        for (int i = 0; i < 3; i++) fe_to_u64x4(s[i], out_evals + 4 * i);
    });
}

// final_claims (dense_interleaved_poly.rs:367-372): coeffs[0], coeffs[1]; out = L.a, L.b, R.a, R.b
int cozk_layer_final_claims(cozk_ctx* ctx, const cozk_layer* l, uint64_t out[16]) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && l && out && l->len == 2, "final_claims: layer must be fully bound (len == 2)");
        fe h[2];
        fetch_fe(ctx, l->buf[l->cur][0], 2, h);
        fe_to_u64x4(h[0], out);
        fe_to_u64x4(h[1], out + 8);
        if (l->mode == COZK_MODE_REP3) {
            fetch_fe(ctx, l->buf[l->cur][1], 2, h);
            fe_to_u64x4(h[0], out + 4);
            fe_to_u64x4(h[1], out + 12);
        } else {
            for (int i = 0; i < 4; i++) out[4 + i] = out[12 + i] = 0;
        }
    });
}

// local half of layer_output / mul_vec: out[j] = L[j] x R[j] + mask_j (additive share c.a);
// masked = 0 for the plain prover (then out IS the next layer).  The ring reshare that turns c.a
// into (c.a, c.b = prev's c.a) is the network seam (cozk_layer_create on the two buffers).
int cozk_layer_output_local(cozk_ctx* ctx, const cozk_layer* l, int masked, const uint8_t* key_self_b, const uint8_t* key_prev_b,
                            uint64_t counter, cozk_vec** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && l && out && l->len >= 2 && (!masked || (key_self_b && key_prev_b)), "layer_output_local: bad argument");
        prf_key seed_self{}, seed_prev{};
        if (masked) {
            seed_self = prf_key_from_bytes(key_self_b);
            seed_prev = prf_key_from_bytes(key_prev_b);
        }
        size_t n = (l->len + 1) / 2;
        fe* d = dev_alloc_fe(n);
        const fe* a = l->buf[l->cur][0];
        const fe* b = l->buf[l->cur][1];
        {
            // algorithmic bytes: 2 n shares read, n additive shares written (192 n for Rep3)
            ProfScope prof(ctx, COZK_PROF_LAYER_OUTPUT, (uint64_t)l->len * (l->mode == COZK_MODE_REP3 ? 64 : 32) + (uint64_t)n * 32);
            if (l->mode == COZK_MODE_REP3) k_layer_output<2><<<grid_for(n), PT, 0, ctx->stream>>>(a, b, l->len, d, n, masked, seed_self, seed_prev, counter);
            else k_layer_output<1><<<grid_for(n), PT, 0, ctx->stream>>>(a, b, l->len, d, n, masked, seed_self, seed_prev, counter);
        }
        HIP_TRY(hipGetLastError());
        *out = new cozk_vec{ctx, n, COZK_SCALAR_FR, d, n * sizeof(fe), true};
    });
}

// rep3::arithmetic::mul_vec local half on two share vectors (x, y given as SoA component vectors)
int cozk_rep3_mul_vec_local(cozk_ctx* ctx, int mode, const cozk_vec* xa, const cozk_vec* xb, const cozk_vec* ya,
                            const cozk_vec* yb, int masked, const uint8_t* key_self_b, const uint8_t* key_prev_b, uint64_t counter,
                            cozk_vec** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && xa && ya && out && xa->n == ya->n && (mode == COZK_MODE_PLAIN || (xb && yb && xb->n == xa->n && yb->n == xa->n)) &&
                         (!masked || (key_self_b && key_prev_b)),
                     "mul_vec_local: bad argument");
        prf_key seed_self{}, seed_prev{};
        if (masked) {
            seed_self = prf_key_from_bytes(key_self_b);
            seed_prev = prf_key_from_bytes(key_prev_b);
        }
        size_t n = xa->n;
        fe* d = dev_alloc_fe(n);
        if (mode == COZK_MODE_REP3)
            k_mul_vec_local<2><<<grid_for(n), PT, 0, ctx->stream>>>((const fe*)xa->d, (const fe*)xb->d, (const fe*)ya->d, (const fe*)yb->d, n, d, masked, seed_self, seed_prev, counter);
        else
            k_mul_vec_local<1><<<grid_for(n), PT, 0, ctx->stream>>>((const fe*)xa->d, nullptr, (const fe*)ya->d, nullptr, n, d, masked, seed_self, seed_prev, counter);
        HIP_TRY(hipGetLastError());
        *out = new cozk_vec{ctx, n, COZK_SCALAR_FR, d, n * sizeof(fe), true};
    });
}

// claimed_outputs (grand_product.rs:266-272): last layer chunks(2) -> L x R, additive; out: n/2 x 4 u64
int cozk_layer_claimed_outputs(cozk_ctx* ctx, const cozk_layer* l, uint64_t* out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && l && out && l->len >= 2 && l->len % 2 == 0, "claimed_outputs: bad argument");
        size_t n = l->len / 2;
        ctx->scratch.reserve(n * sizeof(fe));
        fe* d = ctx->scratch.as<fe>();
        const fe* a = l->buf[l->cur][0];
        const fe* b = l->buf[l->cur][1];
        if (l->mode == COZK_MODE_REP3) k_layer_output<2><<<grid_for(n), PT, 0, ctx->stream>>>(a, b, l->len, d, n, 0, prf_key{}, prf_key{}, 0);
        else k_layer_output<1><<<grid_for(n), PT, 0, ctx->stream>>>(a, b, l->len, d, n, 0, prf_key{}, prf_key{}, 0);
        HIP_TRY(hipGetLastError());
        std::vector<fe> h(n);
        fetch_fe(ctx, d, n, h.data());
        for (size_t i = 0; i < n; i++) fe_to_u64x4(h[i], out + 4 * i);
    });
}

}  // extern "C"

#include "toggle_layer.inc"
#include "primary_sumcheck.inc"
#include "spartan_outer.inc"
#include "spartan_inner.inc"
#include "logup.inc"
