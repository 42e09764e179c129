// libcozk C ABI: context, device vectors, synthetic data, micro-benchmarks.
#include "common.hpp"
#include "fq9.hip.hpp"
#include "prf.hip.hpp"

// ------------------------------------------------------------------ synthetic data
static __device__ __forceinline__ uint64_t splitmix_next(uint64_t& s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// SYNTHETIC DATA ONLY (never secret randomness -- that is prf.hip.hpp): element i draws from its own SplitMix64 stream
// seeded with seed + i * 0xD1342543DE82EF95 (oracle/pyref.py `synthetic_fr` restates this).  FR: rejection-sample a canonical value < r
// (top word masked to 62 bits), optionally masked to max_bits, stored in Montgomery form.
__global__ void k_fill_random_fr(fe* out, size_t n, uint64_t seed, int max_bits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = seed + (uint64_t)i * 0xD1342543DE82EF95ull;
    fe v;
    for (;;) {
        uint64_t w0 = splitmix_next(s), w1 = splitmix_next(s), w2 = splitmix_next(s);
        uint64_t w3 = splitmix_next(s) & ((1ull << 62) - 1ull);
        v.l[0] = (uint32_t)w0; v.l[1] = (uint32_t)(w0 >> 32);
        v.l[2] = (uint32_t)w1; v.l[3] = (uint32_t)(w1 >> 32);
        v.l[4] = (uint32_t)w2; v.l[5] = (uint32_t)(w2 >> 32);
        v.l[6] = (uint32_t)w3; v.l[7] = (uint32_t)(w3 >> 32);
        if (!Fr::geq_mod(v)) break;
    }
    if (max_bits > 0 && max_bits < 254) {
        for (int k = 0; k < 8; k++) {
            int lo = 32 * k;
            if (max_bits <= lo) v.l[k] = 0;
            else if (max_bits < lo + 32) v.l[k] &= (1u << (max_bits - lo)) - 1u;
        }
    }
    fe_store(out + i, Fr::to_mont(v));
}

template <class T>
__global__ void k_fill_random_small(T* out, size_t n, uint64_t seed, int max_bits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s = seed + (uint64_t)i * 0xD1342543DE82EF95ull;
    uint64_t v = splitmix_next(s);
    int bits = (int)sizeof(T) * 8;
    if (max_bits > 0 && max_bits < bits) bits = max_bits;
    if (bits < 64) v &= (1ull << bits) - 1ull;
    out[i] = (T)v;
}

// ------------------------------------------------------------------ element-wise field ops
template <class F, int OP>
__global__ void __launch_bounds__(256) k_fe_binop(const fe* __restrict__ a, const fe* __restrict__ b, fe* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe x = fe_load(a + i), y = fe_load(b + i), r;
    if (OP == COZK_OP_ADD) r = F::add(x, y);
    else if (OP == COZK_OP_SUB) r = F::sub(x, y);
    else r = F::mul(x, y);
    fe_store(out + i, r);
}

__global__ void __launch_bounds__(256) k_fe_scale(fe* __restrict__ v, size_t n, fe s) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store(v + i, Fr::mul(fe_load(v + i), s));
}

template <class F>
static void launch_binop(cozk_ctx* ctx, int op, const fe* a, const fe* b, fe* out, size_t n) {
    unsigned grid = (unsigned)((n + 255) / 256);
    switch (op) {
        case COZK_OP_ADD: k_fe_binop<F, COZK_OP_ADD><<<grid, 256, 0, ctx->stream>>>(a, b, out, n); break;
        case COZK_OP_SUB: k_fe_binop<F, COZK_OP_SUB><<<grid, 256, 0, ctx->stream>>>(a, b, out, n); break;
        case COZK_OP_MUL: k_fe_binop<F, COZK_OP_MUL><<<grid, 256, 0, ctx->stream>>>(a, b, out, n); break;
        default: throw CozkError(COZK_ERR_INVALID_ARG, "unknown field op");
    }
    HIP_TRY(hipGetLastError());
}

// ------------------------------------------------------------------ mont-mul micro benchmark
template <int VARIANT>
__global__ void __launch_bounds__(256) k_bench_montmul(fe* x, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe a = fe_load(x + i);
    fe b = a;
    b.l[0] ^= 0x5a5a5a5au & 0x0fffffffu;
    if (VARIANT == 0) {
        for (int k = 0; k < iters; k++) a = Fq::mul(a, b);
    } else if (VARIANT == 2) {
        // the 9 x 29-bit unsaturated multiplier of the MSM gather kernel (fq9.hip.hpp)
        f9 x9 = f9_from_fe(a), y9 = f9_from_fe(b);
        for (int k = 0; k < iters; k++) x9 = f9_mul(x9, y9);
        a = f9_to_fe(x9);
    } else if (VARIANT == 3) {
        // two 9 x 29 products with interleaved accumulator chains
        f9 x9 = f9_from_fe(a), y9 = f9_from_fe(b), z9 = y9;
        z9.l[1] ^= 0x155u;
        for (int k = 0; k < iters; k += 2) f9_mul_x2(x9, y9, z9, y9, x9, z9);
        a = Fq::add(f9_to_fe(x9), f9_to_fe(z9));
    } else {
        // two independent chains (ILP probe)
        fe c = b;
        for (int k = 0; k < iters; k += 2) {
            a = Fq::mul(a, b);
            c = Fq::mul(c, b);
        }
        a = Fq::add(a, c);
    }
    fe_store(x + i, a);
}

template <typename T>
__global__ void k_vec_narrow(const fe* __restrict__ in, T* __restrict__ out, size_t n, unsigned* __restrict__ bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe v = Fr::from_mont(fe_load(in + i));
    uint32_t high = 0;
    for (int k = (int)(sizeof(T) / 4); k < 8; k++) high |= v.l[k];
    if (high) atomicOr(bad, 1u);
    out[i] = sizeof(T) == 4 ? (T)v.l[0] : (T)((uint64_t)v.l[0] | ((uint64_t)v.l[1] << 32));
}

extern "C" {

int cozk_device_count(int* out) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        if (out) *out = 0;
        return COZK_ERR_NO_DEVICE;
    }
    if (out) *out = n;
    return COZK_OK;
}

int cozk_ctx_create(int device, cozk_ctx** out) {
    if (!out) return COZK_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return COZK_ERR_NO_DEVICE;  // no CPU fallback exists
    if (device < 0 || device >= n) return COZK_ERR_INVALID_ARG;
    cozk_ctx* ctx = new cozk_ctx();
    ctx->device = device;
    int rc = cozk_guard(ctx, [&] { HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); });
    if (rc != COZK_OK) {
        delete ctx;
        return rc;
    }
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        g_live_ctx.insert(ctx);
    }
    *out = ctx;
    return COZK_OK;
}

int cozk_ctx_destroy(cozk_ctx* ctx) {
    if (!ctx) return COZK_OK;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        g_live_ctx.erase(ctx);
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ring_comm) (void)cozk_ring_destroy(ctx);
    ctx->pool.destroy();
    for (auto& pr : ctx->prof_events) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (ProfSlot& sl : ctx->prof_slots) sl.reset();
    ctx->msm_ws.release();
    ctx->scratch.release();
    ctx->scratch2.release();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->round_flag) (void)hipHostFree(ctx->round_flag);
    if (ctx->finish_ticket) (void)hipFree(ctx->finish_ticket);
    if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
    if (ctx->stream2) {
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamDestroy(ctx->stream2);
    }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return COZK_OK;
}

// see cozk.h: the resident round kernel must not be used by provers whose progress depends on each other's GPU work;
// enable > 0 on, 0 off, < 0 back to the automatic default (on only while the context is alone on its device)
int cozk_ctx_set_resident_rounds(cozk_ctx* ctx, int enable) {
    if (!ctx) return COZK_ERR_INVALID_ARG;
    ctx->resident_rounds = enable > 0 ? 1 : (enable == 0 ? 0 : -1);
    return COZK_OK;
}

const char* cozk_last_error(cozk_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int cozk_ctx_synchronize(cozk_ctx* ctx) {
    return cozk_guard(ctx, [&] { HIP_TRY(hipStreamSynchronize(ctx->stream)); });
}

int cozk_ctx_stream(cozk_ctx* ctx, void** out_stream) {
    if (!ctx || !out_stream) return COZK_ERR_INVALID_ARG;
    *out_stream = (void*)ctx->stream;
    return COZK_OK;
}

int cozk_vec_alloc(cozk_ctx* ctx, size_t n, int kind, cozk_vec** out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && out && scalar_kind_bytes(kind), "vec_alloc: bad argument");
        size_t bytes = n * scalar_kind_bytes(kind);
        void* d = ctx_dev_alloc(ctx, bytes);
        *out = new cozk_vec{ctx, n, kind, d, bytes, true};
    });
}

int cozk_vec_upload(cozk_ctx* ctx, const void* host, size_t n, int kind, cozk_vec** out) {
    int rc = cozk_vec_alloc(ctx, n, kind, out);
    if (rc != COZK_OK) return rc;
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(host || n == 0, "vec_upload: null host pointer");
        if ((*out)->bytes) {
            HIP_TRY(hipMemcpyAsync((*out)->d, host, (*out)->bytes, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
    });
}

int cozk_vec_narrow(cozk_ctx* ctx, const cozk_vec* fr, int kind, cozk_vec** out) {
    if (!out) return COZK_ERR_INVALID_ARG;
    *out = nullptr;
    int rc = cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && fr && fr->kind == COZK_SCALAR_FR && (kind == COZK_SCALAR_U32 || kind == COZK_SCALAR_U64), "vec_narrow: an FR vector and U32 / U64");
    });
    if (rc != COZK_OK) return rc;
    rc = cozk_vec_alloc(ctx, fr->n, kind, out);
    if (rc != COZK_OK) return rc;
    unsigned bad = 0;
    rc = cozk_guard(ctx, [&] {
        if (fr->n == 0) return;
        ctx->scratch2.reserve(64);
        unsigned* d_bad = (unsigned*)ctx->scratch2.p;
        HIP_TRY(hipMemsetAsync(d_bad, 0, sizeof(unsigned), ctx->stream));
        const unsigned grid = (unsigned)((fr->n + 255) / 256);
        if (kind == COZK_SCALAR_U32) k_vec_narrow<uint32_t><<<grid, 256, 0, ctx->stream>>>((const fe*)fr->d, (uint32_t*)(*out)->d, fr->n, d_bad);
        else k_vec_narrow<uint64_t><<<grid, 256, 0, ctx->stream>>>((const fe*)fr->d, (uint64_t*)(*out)->d, fr->n, d_bad);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&bad, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    });
    if (rc == COZK_OK && bad) {
        ctx->last_error = "vec_narrow: a value does not fit the requested kind";
        rc = COZK_ERR_INVALID_ARG;
    }
    if (rc != COZK_OK) {
        cozk_vec_free(*out);
        *out = nullptr;
    }
    return rc;
}

int cozk_vec_download(cozk_ctx* ctx, const cozk_vec* v, void* host) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && v && (host || v->bytes == 0), "vec_download: bad argument");
        if (v->bytes) {
            HIP_TRY(hipMemcpyAsync(host, v->d, v->bytes, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
    });
}

int cozk_vec_free(cozk_vec* v) {
    if (!v) return COZK_OK;
    if (v->owned && v->d) ctx_dev_free(v->ctx, v->d);
    delete v;
    return COZK_OK;
}

size_t cozk_vec_len(const cozk_vec* v) { return v ? v->n : 0; }
void* cozk_vec_device_ptr(const cozk_vec* v) { return v ? v->d : nullptr; }

// Rep3 sharing of a secret vector on the device (rep3::share_field_element, mpc-core/src/protocols/rep3/arithmetic.rs:
// 21-33; the witness scatter of jolt/vm/*/witness.rs generate_poly_shares_rep3): t0 = PRF(key0, .), t1 = PRF(key1, .)
// (ChaCha12, prf.hip.hpp), t2 = v - t0 - t1; party 0 holds (t0, t2), party 1 (t1, t0), party 2 (t2, t1).  One fused pass.
// (The reference's generate_poly_shares_rep3 repeats ONE random element over the whole vector, SURVEY 9: not copied.)
__global__ void __launch_bounds__(256) k_rep3_share(const fe* __restrict__ v, size_t n, prf_key key0, prf_key key1, uint64_t ctr, int party,
                                                    fe* __restrict__ a, fe* __restrict__ b) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe t0 = prf_fr(key0, ctr + i), t1 = prf_fr(key1, ctr + i);
    fe t2 = Fr::sub(Fr::sub(fe_load(v + i), t0), t1);
    fe_store(a + i, party == 0 ? t0 : party == 1 ? t1 : t2);
    fe_store(b + i, party == 0 ? t2 : party == 1 ? t0 : t1);
}
__global__ void __launch_bounds__(256) k_fill_prf(fe* __restrict__ out, size_t n, prf_key key, uint64_t ctr) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store(out + i, prf_fr(key, ctr + i));
}
int cozk_vec_fill_prf(cozk_ctx* ctx, cozk_vec* v, const uint8_t key[COZK_PRF_KEY_BYTES], uint64_t counter) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && v && key && v->kind == COZK_SCALAR_FR, "vec_fill_prf: bad argument");
        if (v->n == 0) return;
        k_fill_prf<<<(unsigned)((v->n + 255) / 256), 256, 0, ctx->stream>>>((fe*)v->d, v->n, prf_key_from_bytes(key), counter);
        HIP_TRY(hipGetLastError());
    });
}
int cozk_rep3_share_vec(cozk_ctx* ctx, const cozk_vec* v, const uint8_t key0[COZK_PRF_KEY_BYTES], const uint8_t key1[COZK_PRF_KEY_BYTES],
                        uint64_t counter, int party, cozk_vec** out_a, cozk_vec** out_b) {
    if (!out_a || !out_b) return COZK_ERR_INVALID_ARG;
    *out_a = *out_b = nullptr;
    int rc = cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && v && key0 && key1 && v->kind == COZK_SCALAR_FR && party >= 0 && party < 3, "rep3_share_vec: bad argument");
    });
    if (rc != COZK_OK) return rc;
    rc = cozk_vec_alloc(ctx, v->n, COZK_SCALAR_FR, out_a);
    if (rc != COZK_OK) return rc;
    rc = cozk_vec_alloc(ctx, v->n, COZK_SCALAR_FR, out_b);
    if (rc != COZK_OK) {
        cozk_vec_free(*out_a);
        *out_a = nullptr;
        return rc;
    }
    return cozk_guard(ctx, [&] {
        if (v->n == 0) return;
        k_rep3_share<<<(unsigned)((v->n + 255) / 256), 256, 0, ctx->stream>>>((const fe*)v->d, v->n, prf_key_from_bytes(key0), prf_key_from_bytes(key1),
                                                                              counter, party, (fe*)(*out_a)->d, (fe*)(*out_b)->d);
        HIP_TRY(hipGetLastError());
    });
}

// The witness scatter, device to device (jolt/vm/jolt/coordinator.rs:72-91 + receive_witness_share, jolt/vm/*/witness.rs):
// the dealer's context holds the secret vector; `party`'s Rep3 components are generated on the dealer's GPU and land in
// vectors owned by the PARTY's context -- written in place when both contexts share a device, otherwise generated into the
// dealer's memory and moved with one peer copy per component (xGMI on an MI355X node).  Nothing crosses the host; the
// reference serialises 64-byte shares through QUIC (init = 71 s at 2^20).
int cozk_rep3_scatter(cozk_ctx* dealer, const cozk_vec* v, const uint8_t key0[COZK_PRF_KEY_BYTES], const uint8_t key1[COZK_PRF_KEY_BYTES],
                      uint64_t counter, cozk_ctx* party_ctx, int party, cozk_vec** out_a, cozk_vec** out_b) {
    if (!out_a || !out_b || !party_ctx) return COZK_ERR_INVALID_ARG;
    *out_a = *out_b = nullptr;
    int rc = cozk_guard(dealer, [&] {
        COZK_REQUIRE(dealer && v && key0 && key1 && v->kind == COZK_SCALAR_FR && party >= 0 && party < 3, "rep3_scatter: bad argument");
    });
    if (rc != COZK_OK) return rc;
    rc = cozk_vec_alloc(party_ctx, v->n, COZK_SCALAR_FR, out_a);
    if (rc == COZK_OK) rc = cozk_vec_alloc(party_ctx, v->n, COZK_SCALAR_FR, out_b);
    if (rc == COZK_OK)
        rc = cozk_guard(dealer, [&] {
            const size_t n = v->n;
            if (n == 0) return;
            const bool same = dealer->device == party_ctx->device;
            // out_a / out_b come from the PARTY's pool, whose blocks are ordered by the party's stream only (common.hpp):
            // a block the party has just freed may still be read by a kernel in flight there, so the dealer's stream must
            // not write it before the party's stream has drained
            HIP_TRY(hipStreamSynchronize(party_ctx->stream));
            fe *da = (fe*)(*out_a)->d, *db = (fe*)(*out_b)->d;
            fe *sa = da, *sb = db;
            if (!same) {
                sa = (fe*)ctx_dev_alloc(dealer, 2 * n * sizeof(fe));
                sb = sa + n;
            }
            k_rep3_share<<<(unsigned)((n + 255) / 256), 256, 0, dealer->stream>>>((const fe*)v->d, n, prf_key_from_bytes(key0), prf_key_from_bytes(key1), counter,
                                                                              party, sa, sb);
            HIP_TRY(hipGetLastError());
            if (!same) {
                HIP_TRY(hipMemcpyPeerAsync(da, party_ctx->device, sa, dealer->device, n * sizeof(fe), dealer->stream));
                HIP_TRY(hipMemcpyPeerAsync(db, party_ctx->device, sb, dealer->device, n * sizeof(fe), dealer->stream));
            }
            HIP_TRY(hipStreamSynchronize(dealer->stream));  // the party's stream may use the shares as soon as this returns
            if (!same) ctx_dev_free(dealer, sa);
        });
    if (rc != COZK_OK) {
        cozk_vec_free(*out_a);
        cozk_vec_free(*out_b);
        *out_a = *out_b = nullptr;
    }
    return rc;
}

int cozk_vec_fill_random(cozk_ctx* ctx, cozk_vec* v, uint64_t seed, int max_bits) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && v, "vec_fill_random: bad argument");
        if (v->n == 0) return;
        unsigned grid = (unsigned)((v->n + 255) / 256);
        switch (v->kind) {
            case COZK_SCALAR_FR:
                k_fill_random_fr<<<grid, 256, 0, ctx->stream>>>((fe*)v->d, v->n, seed, max_bits);
                break;
            case COZK_SCALAR_U8:
                k_fill_random_small<uint8_t><<<grid, 256, 0, ctx->stream>>>((uint8_t*)v->d, v->n, seed, max_bits);
                break;
            case COZK_SCALAR_U16:
                k_fill_random_small<uint16_t><<<grid, 256, 0, ctx->stream>>>((uint16_t*)v->d, v->n, seed, max_bits);
                break;
            case COZK_SCALAR_U32:
                k_fill_random_small<uint32_t><<<grid, 256, 0, ctx->stream>>>((uint32_t*)v->d, v->n, seed, max_bits);
                break;
            default:
                k_fill_random_small<uint64_t><<<grid, 256, 0, ctx->stream>>>((uint64_t*)v->d, v->n, seed, max_bits);
                break;
        }
        HIP_TRY(hipGetLastError());
    });
}

int cozk_vec_scale(cozk_ctx* ctx, cozk_vec* v, const uint64_t s[4]) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && v && s && v->kind == COZK_SCALAR_FR, "vec_scale: bad argument");
        if (v->n == 0) return;
        fe sc;
        for (int i = 0; i < 4; i++) {
            sc.l[2 * i] = (uint32_t)s[i];
            sc.l[2 * i + 1] = (uint32_t)(s[i] >> 32);
        }
        k_fe_scale<<<(unsigned)((v->n + 255) / 256), 256, 0, ctx->stream>>>((fe*)v->d, v->n, sc);
        HIP_TRY(hipGetLastError());
    });
}

int cozk_vec_binop(cozk_ctx* ctx, int op, int base_field, const cozk_vec* a, const cozk_vec* b, cozk_vec* out) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && a && b && out && a->kind == COZK_SCALAR_FR && b->kind == COZK_SCALAR_FR &&
                         out->kind == COZK_SCALAR_FR && a->n == b->n && a->n == out->n,
                     "vec_binop: bad argument");
        if (a->n == 0) return;
        if (base_field) launch_binop<Fq>(ctx, op, (const fe*)a->d, (const fe*)b->d, (fe*)out->d, a->n);
        else launch_binop<Fr>(ctx, op, (const fe*)a->d, (const fe*)b->d, (fe*)out->d, a->n);
    });
}

int cozk_bench_montmul(cozk_ctx* ctx, size_t lanes, int iters, int variant, double* out_ms) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && out_ms && lanes >= 256 && lanes % 256 == 0 && iters > 0, "bench_montmul: bad argument");
        ctx->scratch2.reserve(lanes * sizeof(fe));
        fe* x = ctx->scratch2.as<fe>();
        k_fill_random_fr<<<(unsigned)(lanes / 256), 256, 0, ctx->stream>>>(x, lanes, 12345, 0);
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        // warm-up launch, then the timed one
        auto launch = [&](int it) {
            // bits 8.. of `variant`: KiB of dynamic LDS per workgroup, an occupancy limiter for the probe (160 KiB per CU,
            // one wave per SIMD per workgroup: 40 KiB -> 4 waves per SIMD, 53 -> 3, 64 -> 2)
            const int v = variant & 0xff;
            const size_t lds = (size_t)(variant >> 8) * 1024;
            if (v == 0) k_bench_montmul<0><<<(unsigned)(lanes / 256), 256, lds, ctx->stream>>>(x, it);
            else if (v == 2) k_bench_montmul<2><<<(unsigned)(lanes / 256), 256, lds, ctx->stream>>>(x, it);
            else if (v == 3) k_bench_montmul<3><<<(unsigned)(lanes / 256), 256, lds, ctx->stream>>>(x, it);
            else k_bench_montmul<1><<<(unsigned)(lanes / 256), 256, lds, ctx->stream>>>(x, it);
        };
        launch(8);
        HIP_TRY(hipEventRecord(e0, ctx->stream));
        launch(iters);
        HIP_TRY(hipEventRecord(e1, ctx->stream));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        *out_ms = ms;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    });
}

}  // extern "C"
