// Host utilities: arkworks-compatible wire encoding, SHA-256 transcript, small Fr helpers.
//
// Wire = ark-serialize *uncompressed* encoding, as the reference's transport uses
// (mpc-net/src/rep3/quic/worker.rs:187-219): Fr = 32-byte little-endian canonical integer,
// Vec<T> = u64 LE length prefix + elements, tuples = concatenation, usize = u64 LE,
// G1Affine = x || y (32 B LE each) with ark-ec's SWFlags in the two spare top bits of the last byte: bit 6 = point at
// infinity (x = y = 0), bit 7 = "y is negative", i.e. y > -y as integers (short_weierstrass serialize_with_mode passes
// `to_flags()` to y.serialize_with_flags also when uncompressed; from upstream knowledge of ark-ec 0.5 -- the reference
// holds no serialized point to pin it, see tests/golden/wire_format.json).  Readers ignore bit 7 and validate like
// arkworks' Validate::Yes: canonical coordinates, on the curve (BN254 G1 has cofactor 1: no subgroup check needed).
#pragma once
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define COZK_HAVE_SHANI 1
#else
#define COZK_HAVE_SHANI 0
#endif

#include <string.h>

#include <string>
#include <vector>

#include "../poly.hip.hpp"

namespace cozk {

typedef std::vector<uint8_t> Bytes;

struct Writer {
    Bytes b;
    void u64(uint64_t v) {
        for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i)));
    }
    void fr(const fe& mont) {
        fe c = Fr::from_mont(mont);
        for (int i = 0; i < 8; i++)
            for (int k = 0; k < 4; k++) b.push_back((uint8_t)(c.l[i] >> (8 * k)));
    }
    void vec_fr(const std::vector<fe>& v) {
        u64(v.size());
        for (const fe& x : v) fr(x);
    }
    void g1(const g1_affine& p) {
        if (G1::is_inf(p)) {
            for (int i = 0; i < 63; i++) b.push_back(0);
            b.push_back(0x40);
            return;
        }
        fe x = Fq::from_mont(p.x), y = Fq::from_mont(p.y), ny = Fq::from_mont(Fq::neg(p.y));
        for (int i = 0; i < 8; i++)
            for (int k = 0; k < 4; k++) b.push_back((uint8_t)(x.l[i] >> (8 * k)));
        for (int i = 0; i < 8; i++)
            for (int k = 0; k < 4; k++) b.push_back((uint8_t)(y.l[i] >> (8 * k)));
        // SWFlags::from_y_coordinate: YIsNegative (bit 7) iff y > -y
        bool neg = false;
        for (int i = 7; i >= 0; i--) {
            if (y.l[i] != ny.l[i]) {
                neg = y.l[i] > ny.l[i];
                break;
            }
        }
        if (neg) b.back() |= 0x80;
    }
    void vec_g1(const std::vector<g1_affine>& v) {
        u64(v.size());
        for (const auto& p : v) g1(p);
    }
};

struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    explicit Reader(const Bytes& b) : p(b.data()), end(b.data() + b.size()) {}
    void need(size_t n) const {
        if ((size_t)(end - p) < n) throw CozkError(COZK_ERR_INTERNAL, "wire: truncated message");
    }
    uint64_t u64() {
        need(8);
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i);
        p += 8;
        return v;
    }
    fe fr() {
        need(32);
        fe c;
        for (int i = 0; i < 8; i++) c.l[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        p += 32;
        if (Fr::geq_mod(c)) throw CozkError(COZK_ERR_INTERNAL, "wire: non-canonical field element");
        return Fr::to_mont(c);
    }
    // a length prefix is checked against what the message still holds BEFORE it is multiplied (no overflow)
    void need_elems(uint64_t n, size_t elem) const {
        if (n > (uint64_t)(end - p) / elem) throw CozkError(COZK_ERR_INTERNAL, "wire: truncated message");
    }
    std::vector<fe> vec_fr() {
        uint64_t n = u64();
        need_elems(n, 32);
        std::vector<fe> v(n);
        for (auto& x : v) x = fr();
        return v;
    }
    g1_affine g1() {
        need(64);
        g1_affine a;
        fe x, y;
        for (int i = 0; i < 8; i++) {
            x.l[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
            y.l[i] = (uint32_t)p[32 + 4 * i] | ((uint32_t)p[33 + 4 * i] << 8) | ((uint32_t)p[34 + 4 * i] << 16) | ((uint32_t)p[35 + 4 * i] << 24);
        }
        const uint8_t flags = p[63] & 0xC0;
        y.l[7] &= 0x3fffffffu;  // the two flag bits are not part of y
        p += 64;
        // what arkworks' deserialize (Validate::Yes) rejects is rejected here too: a peer's bytes are not trusted
        if (flags == 0xC0) throw CozkError(COZK_ERR_INTERNAL, "wire: invalid point flags");
        if (Fq::geq_mod(x) || Fq::geq_mod(y)) throw CozkError(COZK_ERR_INTERNAL, "wire: non-canonical point coordinate");
        if (flags & 0x40) {
            if (!Fq::is_zero(x) || !Fq::is_zero(y)) throw CozkError(COZK_ERR_INTERNAL, "wire: point at infinity with non-zero coordinates");
            a.x = Fq::zero();
            a.y = Fq::zero();
            return a;
        }
        a.x = Fq::to_mont(x);
        a.y = Fq::to_mont(y);
        // y^2 = x^3 + 3
        fe rhs = Fq::add(Fq::mul(Fq::sqr(a.x), a.x), Fq::from_u64(3));
        if (!Fq::eq(Fq::sqr(a.y), rhs)) throw CozkError(COZK_ERR_INTERNAL, "wire: point is not on the curve");
        return a;
    }
    std::vector<g1_affine> vec_g1() {
        uint64_t n = u64();
        need_elems(n, 64);
        std::vector<g1_affine> v(n);
        for (auto& x : v) x = g1();
        return v;
    }
};

// ---------------------------------------------------------------- SHA-256 (FIPS 180-4)
struct Sha256 {
    uint32_t h[8];
    uint8_t buf[64];
    uint64_t len = 0;
    size_t fill = 0;
    Sha256() {
        static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        memcpy(h, iv, sizeof h);
    }
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
#if COZK_HAVE_SHANI
    // One compression with the x86 SHA extensions (53 ns instead of 302 ns per block on the boxes' EPYC hosts): the transcript hashes
    // ~4 blocks per sumcheck round and a chained proof has ~1660 rounds whose coordinator sits on the workers' critical path.  The
    // instructions keep the state as (ABEF, CDGH); checked against the portable code on random blocks (tests/test_wire_format.py
    // pins the digests through the proofs).
    __attribute__((target("sha,sse4.1,ssse3"))) static void block_shani(uint32_t h[8], const uint8_t* p, const uint32_t* K) {
        const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
        __m128i t = _mm_loadu_si128((const __m128i*)&h[0]);
        __m128i s1 = _mm_loadu_si128((const __m128i*)&h[4]);
        t = _mm_shuffle_epi32(t, 0xB1);
        s1 = _mm_shuffle_epi32(s1, 0x1B);
        __m128i s0 = _mm_alignr_epi8(t, s1, 8);
        s1 = _mm_blend_epi16(s1, t, 0xF0);
        const __m128i s0_save = s0, s1_save = s1;
        __m128i m[4];
        for (int i = 0; i < 4; i++) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * i)), bswap);
        for (int r = 0; r < 16; r++) {
            const __m128i wk = _mm_add_epi32(m[r & 3], _mm_loadu_si128((const __m128i*)&K[4 * r]));
            s1 = _mm_sha256rnds2_epu32(s1, s0, wk);
            s0 = _mm_sha256rnds2_epu32(s0, s1, _mm_shuffle_epi32(wk, 0x0E));
            if (r < 12) {  // the four message words 16 rounds ahead: W[t] = s1(W[t-2]) + W[t-7] + s0(W[t-15]) + W[t-16]
                __m128i x = _mm_sha256msg1_epu32(m[r & 3], m[(r + 1) & 3]);
                x = _mm_add_epi32(x, _mm_alignr_epi8(m[(r + 3) & 3], m[(r + 2) & 3], 4));
                m[r & 3] = _mm_sha256msg2_epu32(x, m[(r + 3) & 3]);
            }
        }
        s0 = _mm_add_epi32(s0, s0_save);
        s1 = _mm_add_epi32(s1, s1_save);
        t = _mm_shuffle_epi32(s0, 0x1B);
        s1 = _mm_shuffle_epi32(s1, 0xB1);
        s0 = _mm_blend_epi16(t, s1, 0xF0);
        s1 = _mm_alignr_epi8(s1, t, 8);
        _mm_storeu_si128((__m128i*)&h[0], s0);
        _mm_storeu_si128((__m128i*)&h[4], s1);
    }
#endif
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
            0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
            0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
            0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
            0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
            0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#if COZK_HAVE_SHANI
        static const bool shani = __builtin_cpu_supports("sha") && getenv("COZK_NO_SHANI") == nullptr;
        if (shani) {
            block_shani(h, p, K);
            return;
        }
#endif
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
            uint32_t ch = (e & f) ^ (~e & g);
            uint32_t t1 = hh + S1 + ch + K[i] + w[i];
            uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
            uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
            uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const uint8_t* p, size_t n) {
        len += n;
        while (n) {
            size_t k = 64 - fill < n ? 64 - fill : n;
            memcpy(buf + fill, p, k);
            fill += k;
            p += k;
            n -= k;
            if (fill == 64) {
                block(buf);
                fill = 0;
            }
        }
    }
    void final(uint8_t out[32]) {
        uint64_t bits = len * 8;
        uint8_t pad = 0x80;
        update(&pad, 1);
        uint8_t z = 0;
        while (fill != 56) update(&z, 1);
        uint8_t lb[8];
        for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(lb, 8);
        for (int i = 0; i < 8; i++) {
            out[4 * i] = (uint8_t)(h[i] >> 24);
            out[4 * i + 1] = (uint8_t)(h[i] >> 16);
            out[4 * i + 2] = (uint8_t)(h[i] >> 8);
            out[4 * i + 3] = (uint8_t)h[i];
        }
    }
};

// Harness transcript.  The reference's KeccakTranscript lives out of tree (jolt-core) and only the
// coordinator ever hashes; this SHA-256 sponge keeps its shape -- 32-byte state,
// u32 round counter, 128-bit challenges -- and is restated bit-for-bit in oracle/pyref.py.
struct Transcript {
    uint8_t state[32];
    uint32_t n_rounds = 0;
    explicit Transcript(const char* label = "cozk") {
        Sha256 s;
        s.update((const uint8_t*)label, strlen(label));
        s.final(state);
    }
    void absorb(const uint8_t* data, size_t n) {
        Sha256 s;
        s.update(state, 32);
        uint8_t c[4] = {(uint8_t)n_rounds, (uint8_t)(n_rounds >> 8), (uint8_t)(n_rounds >> 16), (uint8_t)(n_rounds >> 24)};
        s.update(c, 4);
        s.update(data, n);
        s.final(state);
        n_rounds++;
    }
    void append_scalar(const fe& x) {
        Writer w;
        w.fr(x);
        absorb(w.b.data(), w.b.size());
    }
    void append_scalars(const std::vector<fe>& xs) {
        Writer w;
        for (const fe& x : xs) w.fr(x);
        absorb(w.b.data(), w.b.size());
    }
    void append_point(const g1_affine& p) {
        // raw x || y without serialization flags (jolt's append_point hashes the two coordinates; zeros for infinity)
        Writer w;
        if (G1::is_inf(p)) {
            w.b.assign(64, 0);
        } else {
            w.g1(p);
            w.b.back() &= 0x3f;
        }
        absorb(w.b.data(), w.b.size());
    }
    fe challenge_scalar() {
        absorb((const uint8_t*)"challenge", 9);
        fe c = Fr::zero();
        for (int i = 0; i < 4; i++) c.l[i] = (uint32_t)state[4 * i] | ((uint32_t)state[4 * i + 1] << 8) | ((uint32_t)state[4 * i + 2] << 16) | ((uint32_t)state[4 * i + 3] << 24);
        return Fr::to_mont(c);
    }
    std::vector<fe> challenge_vector(size_t n) {
        std::vector<fe> v(n);
        for (auto& x : v) x = challenge_scalar();
        return v;
    }
};

// ---------------------------------------------------------------- small Fr helpers (host)
static inline fe fr_from_u64(uint64_t v) { return Fr::from_u64(v); }

static inline fe unipoly_eval(const std::vector<fe>& coeffs, const fe& x) {
    fe acc = Fr::zero();
    for (size_t i = coeffs.size(); i-- > 0;) acc = Fr::add(Fr::mul(acc, x), coeffs[i]);
    return acc;
}
// CompressedUniPoly: drop the linear term (subprotocols/sumcheck.rs:146-148)
static inline std::vector<fe> unipoly_compress(const std::vector<fe>& c) {
    std::vector<fe> o;
    o.push_back(c[0]);
    for (size_t i = 2; i < c.size(); i++) o.push_back(c[i]);
    return o;
}
// decompress given the claim e = g(0) + g(1): c1 = e - 2 c0 - sum(rest)
static inline std::vector<fe> unipoly_decompress(const std::vector<fe>& comp, const fe& e) {
    fe c1 = Fr::sub(e, Fr::dbl(comp[0]));
    for (size_t i = 1; i < comp.size(); i++) c1 = Fr::sub(c1, comp[i]);
    std::vector<fe> o;
    o.push_back(comp[0]);
    o.push_back(c1);
    for (size_t i = 1; i < comp.size(); i++) o.push_back(comp[i]);
    return o;
}
// EqPolynomial::evals(r) on the host (big-endian)
static inline std::vector<fe> eq_evals_host(const std::vector<fe>& r) {
    std::vector<fe> ev(1, Fr::one());
    for (const fe& rj : r) {
        std::vector<fe> nx(ev.size() * 2);
        for (size_t i = 0; i < ev.size(); i++) {
            fe hi = Fr::mul(ev[i], rj);
            nx[2 * i] = Fr::sub(ev[i], hi);
            nx[2 * i + 1] = hi;
        }
        ev.swap(nx);
    }
    return ev;
}
// additive::combine_additive_vec (mpc-core/src/protocols/additive.rs:103-114)
static inline std::vector<fe> combine_additive(const std::vector<std::vector<fe>>& parts) {
    std::vector<fe> o(parts[0].size(), Fr::zero());
    for (const auto& p : parts) {
        if (p.size() != o.size()) throw CozkError(COZK_ERR_INTERNAL, "combine_additive: length mismatch");
        for (size_t i = 0; i < o.size(); i++) o[i] = Fr::add(o[i], p[i]);
    }
    return o;
}

static inline void rc_check(int rc, cozk_ctx* ctx, const char* what) {
    if (rc != COZK_OK) throw CozkError(rc, std::string(what) + ": " + (ctx ? cozk_last_error(ctx) : "?"));
}

}  // namespace cozk
