// Keyed PRF for every piece of secret randomness the engine generates: the Rep3 sharing of a witness vector
// (rep3::share_field_element, mpc-core/src/protocols/rep3/arithmetic.rs:21-33) and the zero-sharing masks of
// mul / mul_vec and of the co-spartan round messages (Rep3Rand / get_mask_scalar_*, mpc-core/src/protocols/
// additive.rs:44-50, rep3/arithmetic.rs:39-48).  The reference draws these from ChaCha12 streams keyed with 32-byte
// seeds that neighbouring parties exchange once (IoContext::init; mpc-types/src/protocols/rep3.rs:29,177,
// mpc-core/src/protocols/rep3/network.rs:190-211).  A GPU wants random access instead of a sequential stream, so
// element j of a stream is its own ChaCha12 block:
//
//   state = "expand 32-byte k" | key[0..8] | counter = j (64 bit) | domain "COZK" | attempt
//   block = 12 rounds (6 double rounds) + feed-forward, as in RFC 8439 with the round count of rand_chacha's ChaCha12
//   element = the first of (words 0..7, words 8..15; top word masked to 30 bits, i.e. a 254-bit integer) that is
//             below r; if neither is (probability 6 %), attempt += 1 and the next block is drawn
//
// which is uniform on [0, r) and a PRF in (key, j) as long as ChaCha12 is.  The value is returned in Montgomery
// form.  Host and device run the same code (the per-round masks of the co-spartan sumchecks are made on the host).
// oracle/pyref.py `prf_fr` and oracle/c restate it; the masks cancel in every sum, so proofs do not depend on it.
#pragma once
#include "ff.hip.hpp"

struct prf_key {
    uint32_t k[8];
};

static inline prf_key prf_key_from_bytes(const uint8_t b[32]) {
    prf_key key;
    for (int i = 0; i < 8; i++)
        key.k[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    return key;
}

static FF_HD uint32_t prf_rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }

#define COZK_CHACHA_QR(a, b, c, d)  \
    a += b; d ^= a; d = prf_rotl(d, 16); \
    c += d; b ^= c; b = prf_rotl(b, 12); \
    a += b; d ^= a; d = prf_rotl(d, 8);  \
    c += d; b ^= c; b = prf_rotl(b, 7);

static FF_HD void chacha12_block(const prf_key& key, uint64_t counter, uint32_t attempt, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                      key.k[0], key.k[1], key.k[2], key.k[3], key.k[4], key.k[5], key.k[6], key.k[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), 0x4b5a4f43u, attempt};
    uint32_t x0 = s[0], x1 = s[1], x2 = s[2], x3 = s[3], x4 = s[4], x5 = s[5], x6 = s[6], x7 = s[7];
    uint32_t x8 = s[8], x9 = s[9], x10 = s[10], x11 = s[11], x12 = s[12], x13 = s[13], x14 = s[14], x15 = s[15];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        COZK_CHACHA_QR(x0, x4, x8, x12)
        COZK_CHACHA_QR(x1, x5, x9, x13)
        COZK_CHACHA_QR(x2, x6, x10, x14)
        COZK_CHACHA_QR(x3, x7, x11, x15)
        COZK_CHACHA_QR(x0, x5, x10, x15)
        COZK_CHACHA_QR(x1, x6, x11, x12)
        COZK_CHACHA_QR(x2, x7, x8, x13)
        COZK_CHACHA_QR(x3, x4, x9, x14)
    }
    out[0] = x0 + s[0]; out[1] = x1 + s[1]; out[2] = x2 + s[2]; out[3] = x3 + s[3];
    out[4] = x4 + s[4]; out[5] = x5 + s[5]; out[6] = x6 + s[6]; out[7] = x7 + s[7];
    out[8] = x8 + s[8]; out[9] = x9 + s[9]; out[10] = x10 + s[10]; out[11] = x11 + s[11];
    out[12] = x12 + s[12]; out[13] = x13 + s[13]; out[14] = x14 + s[14]; out[15] = x15 + s[15];
}

// PRF(key, j) -> uniform element of Fr, Montgomery form
static FF_HD fe prf_fr(const prf_key& key, uint64_t j) {
    for (uint32_t attempt = 0;; attempt++) {
        uint32_t w[16];
        chacha12_block(key, j, attempt, w);
        for (int half = 0; half < 2; half++) {
            fe v;
            for (int i = 0; i < 8; i++) v.l[i] = w[8 * half + i];
            v.l[7] &= 0x3fffffffu;
            if (!Fr::geq_mod(v)) return Fr::to_mont(v);
        }
    }
}
