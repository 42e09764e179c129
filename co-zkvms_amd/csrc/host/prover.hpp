// Worker-side and coordinator-side drivers of the hot path, written against the network seam
// (net.hpp) and the C ABI kernels.  One-to-one with the reference's round loops:
//
//   prove_sumcheck                    co-jolt/src/subprotocols/sumcheck.rs:96-131
//   coordinate_prove_arbitrary        co-jolt/src/subprotocols/sumcheck.rs:134-165
//   Rep3BatchedDenseGrandProduct      co-jolt/src/subprotocols/grand_product.rs:219-282
//   prove_grand_product_worker        co-jolt/src/subprotocols/grand_product.rs:111-130
//   prove_layer / coordinate_prove_layer   grand_product.rs:143-217
//   Rep3ProverOpeningAccumulator      co-jolt/src/poly/opening_proof.rs:63-438
//   PST13                             co-jolt/src/poly/commitment/pst13.rs
//
// `mode` = COZK_MODE_REP3 (three parties, shares {a,b}) or COZK_MODE_PLAIN (one party, the plain
// prover: same message schedule with a single worker whose additive share is the value itself).
#pragma once
#include <chrono>
#include <memory>

#include <string.h>

#include "net.hpp"
#include "../prf.hip.hpp"

namespace cozk {

// ---------------------------------------------------------------- RAII handles over the C ABI
struct VecH {
    cozk_vec* h = nullptr;
    VecH() {}
    explicit VecH(cozk_vec* v) : h(v) {}
    VecH(const VecH&) = delete;
    VecH& operator=(const VecH&) = delete;
    VecH(VecH&& o) noexcept : h(o.h) { o.h = nullptr; }
    VecH& operator=(VecH&& o) noexcept {
        if (this != &o) {
            cozk_vec_free(h);
            h = o.h;
            o.h = nullptr;
        }
        return *this;
    }
    ~VecH() { cozk_vec_free(h); }
};
struct PolyH {
    cozk_poly* h = nullptr;
    PolyH() {}
    explicit PolyH(cozk_poly* p) : h(p) {}
    PolyH(const PolyH&) = delete;
    PolyH& operator=(const PolyH&) = delete;
    PolyH(PolyH&& o) noexcept : h(o.h) { o.h = nullptr; }
    PolyH& operator=(PolyH&& o) noexcept {
        if (this != &o) {
            cozk_poly_free(h);
            h = o.h;
            o.h = nullptr;
        }
        return *this;
    }
    ~PolyH() { cozk_poly_free(h); }
};
struct LayerH {
    cozk_layer* h = nullptr;
    LayerH() {}
    explicit LayerH(cozk_layer* p) : h(p) {}
    LayerH(const LayerH&) = delete;
    LayerH& operator=(const LayerH&) = delete;
    LayerH(LayerH&& o) noexcept : h(o.h) { o.h = nullptr; }
    LayerH& operator=(LayerH&& o) noexcept {
        if (this != &o) {
            cozk_layer_free(h);
            h = o.h;
            o.h = nullptr;
        }
        return *this;
    }
    ~LayerH() { cozk_layer_free(h); }
};
struct EqH {
    cozk_spliteq* h = nullptr;
    EqH() {}
    EqH(const EqH&) = delete;
    EqH& operator=(const EqH&) = delete;
    ~EqH() { cozk_spliteq_free(h); }
};

static inline std::vector<uint64_t> to_abi(const std::vector<fe>& v) {
    std::vector<uint64_t> o(4 * v.size());
    for (size_t i = 0; i < v.size(); i++) fe_to_u64x4(v[i], o.data() + 4 * i);
    return o;
}

// a Rep3 share (or a plain value with b = 0)
struct Share {
    fe a, b;
};

struct WorkerEnv {
    cozk_ctx* ctx;
    int mode;   // COZK_MODE_PLAIN / COZK_MODE_REP3
    int party;  // PartyID 0..2 (0 for the plain prover)
    StarNetWorker* star;
    RingNet* ring;            // null for the plain prover
    uint8_t key_self[COZK_PRF_KEY_BYTES] = {0};  // 32-byte ChaCha12 PRF key shared with the next party (prf.hip.hpp)
    uint8_t key_prev[COZK_PRF_KEY_BYTES] = {0};  // ... with the previous party
    uint64_t mask_ctr = 0;    // zero-sharing counter (advances identically on all parties)
    void set_keys(const uint8_t* self_key, const uint8_t* prev_key) {
        memcpy(key_self, self_key, COZK_PRF_KEY_BYTES);
        memcpy(key_prev, prev_key, COZK_PRF_KEY_BYTES);
    }

    // additive::promote_to_trivial_share(value, id) (mpc-core/src/protocols/additive.rs:62-64)
    fe additive_trivial(const fe& v) const { return party == 0 ? v : Fr::zero(); }
    // rep3::arithmetic::promote_to_trivial_share(id, value) (types.rs:90-96)
    Share rep3_trivial(const fe& v) const {
        Share s{Fr::zero(), Fr::zero()};
        if (mode == COZK_MODE_PLAIN || party == 0) s.a = v;
        else if (party == 1) s.b = v;
        return s;
    }
    // Rep3PrimeFieldShare::into_additive (types.rs:76-81); identity for the plain prover
    fe into_additive(const Share& s) const {
        if (mode == COZK_MODE_PLAIN) return s.a;
        return Fr::mul(Fr::add(s.a, s.b), fr_two_inv());
    }
};

// Keys of the in-process HARNESSES only: a real host hands libcozk the 32-byte seeds its parties exchanged (from its
// CryptoRng); the synthetic runs expand (run seed, key index) to 32 bytes so that every party derives the same pairwise
// keys without a key exchange.  Key `idx` is the one party idx shares with party idx + 1.
static inline void harness_prf_key(uint64_t seed, uint64_t idx, uint8_t out[COZK_PRF_KEY_BYTES]) {
    uint64_t s = seed ^ (0xC0DEC0DEull + idx * 0x9E3779B97F4A7C15ull);
    for (int i = 0; i < 4; i++) {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(z >> (8 * b));
    }
}

// ================================================================= cubic sumcheck over one GKR layer
struct SumcheckResult {
    std::vector<fe> r;
    Share left, right;
};

// optional phase accounting of the round loop (COZK_TRACE_ROUNDS=1): kernel + drain vs star round trip
struct RoundTrace {
    double t_round = 0, t_star = 0, t_rest = 0;
    uint64_t rounds = 0;
};
inline thread_local RoundTrace t_round_trace;
static inline double trace_now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Rep3BatchedCubicSumcheckWorker::prove_sumcheck (sumcheck.rs:96-131).  The round loop itself lives behind the
// ABI (cozk_layer_prove_rounds: per-round launches while the layer is large, one resident kernel for its tail);
// the star exchange of each round is the callback.
struct RoundCtx {
    WorkerEnv* env;
    std::string error;
};
static int prove_sumcheck_round_cb(void* user, int /*round*/, const uint64_t coeffs[16], uint64_t r_out[4], uint64_t next_claim_out[4]) {
    RoundCtx* rc = static_cast<RoundCtx*>(user);
    try {
        WorkerEnv& env = *rc->env;
        double tb = trace_now_us();
        Writer w;
        std::vector<fe> cf(4);
        for (int i = 0; i < 4; i++) cf[i] = fe_from_u64x4(coeffs + 4 * i);
        w.vec_fr(cf);
        env.star->send_response(w.b);
        Bytes req = env.star->receive_request();
        Reader rd(req);
        fe r_j = rd.fr();
        fe next_claim = rd.fr();
        fe_to_u64x4(r_j, r_out);
        fe_to_u64x4(env.additive_trivial(next_claim), next_claim_out);
        t_round_trace.t_star += trace_now_us() - tb;
        t_round_trace.rounds++;
        return 0;
    } catch (const std::exception& e) {
        rc->error = e.what();
        return 1;
    }
}
static SumcheckResult prove_sumcheck(WorkerEnv& env, cozk_layer* layer, const fe& claim, cozk_spliteq* eq, int num_rounds) {
    SumcheckResult res;
    uint64_t pc[4], fc[16];
    fe_to_u64x4(claim, pc);
    std::vector<uint64_t> rs((size_t)4 * (num_rounds > 0 ? num_rounds : 1));
    RoundCtx rc{&env, {}};
    double ta = trace_now_us();
    int st = cozk_layer_prove_rounds(env.ctx, layer, eq, pc, num_rounds, prove_sumcheck_round_cb, &rc, rs.data(), fc);
    t_round_trace.t_round += trace_now_us() - ta;
    if (st != COZK_OK && !rc.error.empty()) throw CozkError(COZK_ERR_INTERNAL, rc.error);
    rc_check(st, env.ctx, "layer_prove_rounds");
    for (int j = 0; j < num_rounds; j++) res.r.push_back(fe_from_u64x4(rs.data() + 4 * j));
    res.left = Share{fe_from_u64x4(fc), fe_from_u64x4(fc + 4)};
    res.right = Share{fe_from_u64x4(fc + 8), fe_from_u64x4(fc + 12)};
    Writer w;
    w.fr(res.left.a);
    w.fr(res.left.b);
    w.fr(res.right.a);
    w.fr(res.right.b);
    env.star->send_response(w.b);
    return res;
}

struct SumcheckProof {
    std::vector<std::vector<fe>> compressed_polys;
};

// coordinate_prove_arbitrary (sumcheck.rs:134-165)
static std::vector<fe> coordinate_prove_arbitrary(StarNetCoordinator& net, Transcript& tr, int num_rounds, SumcheckProof& proof) {
    std::vector<fe> r;
    for (int round = 0; round < num_rounds; round++) {
        std::vector<std::vector<fe>> parts;
        for (Bytes& b : net.receive_responses()) {
            Reader rd(b);
            parts.push_back(rd.vec_fr());
        }
        std::vector<fe> poly = combine_additive(parts);
        std::vector<fe> comp = unipoly_compress(poly);
        tr.append_scalars(comp);
        fe r_j = tr.challenge_scalar();
        r.push_back(r_j);
        fe claim = unipoly_eval(poly, r_j);
        Writer w;
        w.fr(r_j);
        w.fr(claim);
        net.broadcast_request(w.b);
        proof.compressed_polys.push_back(comp);
    }
    return r;
}

// receive_final_claims (sumcheck.rs:53-75): combine_field_element = a0 + a1 + a2
static void receive_final_claims(StarNetCoordinator& net, fe& left, fe& right) {
    left = Fr::zero();
    right = Fr::zero();
    for (Bytes& b : net.receive_responses()) {
        Reader rd(b);
        fe la = rd.fr();
        (void)rd.fr();
        fe ra = rd.fr();
        (void)rd.fr();
        left = Fr::add(left, la);
        right = Fr::add(right, ra);
    }
}

// ================================================================= dense batched grand product (GKR)
struct GrandProductLayerProof {
    SumcheckProof proof;
    fe left_claim, right_claim;
};
struct GrandProductProof {
    std::vector<fe> outputs;
    std::vector<GrandProductLayerProof> gkr_layers;
};

struct Rep3BatchedDenseGrandProduct {
    std::vector<LayerH> layers;  // layers[0] = leaves ... layers.back() = top

    // construct (grand_product.rs:239-255): layer[i+1] = mul_vec(L(layer[i]), R(layer[i]))
    static Rep3BatchedDenseGrandProduct construct(WorkerEnv& env, LayerH leaves, size_t batch_size) {
        size_t n = cozk_layer_len(leaves.h);
        COZK_REQUIRE(batch_size > 0 && n % batch_size == 0, "grand product: leaves.len() % batch_size != 0");
        size_t per = n / batch_size;
        COZK_REQUIRE(per >= 2 && (per & (per - 1)) == 0, "grand product: leaves per circuit must be a power of two >= 2");
        int num_layers = 0;
        while (((size_t)1 << num_layers) < per) num_layers++;
        Rep3BatchedDenseGrandProduct gp;
        gp.layers.push_back(std::move(leaves));
        for (int i = 0; i < num_layers - 1; i++) {
            cozk_layer* prev = gp.layers[i].h;
            size_t n_out = (cozk_layer_len(prev) + 1) / 2;
            cozk_vec* ca = nullptr;
            rc_check(cozk_layer_output_local(env.ctx, prev, env.mode == COZK_MODE_REP3 ? 1 : 0, env.key_self, env.key_prev,
                                             env.mask_ctr, &ca),
                     env.ctx, "layer_output_local");
            VecH va(ca);
            env.mask_ctr += n_out;
            cozk_layer* nl = nullptr;
            if (env.mode == COZK_MODE_REP3) {
                cozk_vec* cb = nullptr;
                rc_check(cozk_vec_alloc(env.ctx, n_out, COZK_SCALAR_FR, &cb), env.ctx, "vec_alloc");
                VecH vb(cb);
                // ring reshare: own c.a -> next, c.b <- prev (arithmetic.rs:148-150)
                env.ring->reshare(env.ctx, (const fe*)cozk_vec_device_ptr(va.h), (fe*)cozk_vec_device_ptr(vb.h), n_out);
                rc_check(cozk_layer_create(env.ctx, COZK_MODE_REP3, va.h, vb.h, 1, &nl), env.ctx, "layer_create");
            } else {
                rc_check(cozk_layer_create(env.ctx, COZK_MODE_PLAIN, va.h, nullptr, 1, &nl), env.ctx, "layer_create");
            }
            gp.layers.push_back(LayerH(nl));
        }
        return gp;
    }

    size_t num_layers() const { return layers.size(); }

    // claimed_outputs (grand_product.rs:266-272)
    std::vector<fe> claimed_outputs(WorkerEnv& env) {
        cozk_layer* top = layers.back().h;
        size_t n = cozk_layer_len(top) / 2;
        std::vector<uint64_t> out(4 * n);
        rc_check(cozk_layer_claimed_outputs(env.ctx, top, out.data()), env.ctx, "claimed_outputs");
        std::vector<fe> v(n);
        for (size_t i = 0; i < n; i++) v[i] = fe_from_u64x4(out.data() + 4 * i);
        return v;
    }

    // prove_layer (grand_product.rs:186-217)
    static void prove_layer(WorkerEnv& env, cozk_layer* layer, fe& claim, std::vector<fe>& r_grand_product) {
        EqH eq;
        std::vector<uint64_t> w = to_abi(r_grand_product);
        rc_check(cozk_spliteq_new(env.ctx, w.data(), (int)r_grand_product.size(), &eq.h), env.ctx, "spliteq_new");
        int num_rounds = (int)r_grand_product.size();
        if (env.party == 0) {
            Writer wr;
            wr.u64((uint64_t)num_rounds);
            env.star->send_response(wr.b);
        }
        SumcheckResult sc = prove_sumcheck(env, layer, claim, eq.h, num_rounds);
        r_grand_product.assign(sc.r.rbegin(), sc.r.rend());
        Bytes req = env.star->receive_request();
        Reader rd(req);
        fe r_layer = rd.fr();
        // claim = add_mul_public(left, right - left, r_layer).into_additive()
        Share s;
        s.a = Fr::add(sc.left.a, Fr::mul(Fr::sub(sc.right.a, sc.left.a), r_layer));
        s.b = Fr::add(sc.left.b, Fr::mul(Fr::sub(sc.right.b, sc.left.b), r_layer));
        claim = env.into_additive(s);
        r_grand_product.push_back(r_layer);
    }

    // prove_grand_product_worker (grand_product.rs:111-130).  Layers are bound destructively.
    std::vector<fe> prove_grand_product_worker(WorkerEnv& env) {
        std::vector<fe> outputs = claimed_outputs(env);
        Writer w;
        w.vec_fr(outputs);
        env.star->send_response(w.b);
        Bytes req = env.star->receive_request();
        Reader rd(req);
        std::vector<fe> r = rd.vec_fr();
        fe claim = env.additive_trivial(rd.fr());
        for (size_t i = layers.size(); i-- > 0;) prove_layer(env, layers[i].h, claim, r);
        return r;
    }
};

// coordinate_prove_layer (grand_product.rs:143-183)
static GrandProductLayerProof coordinate_prove_layer(StarNetCoordinator& net, Transcript& tr, fe& claim, std::vector<fe>& r_grand_product) {
    GrandProductLayerProof lp;
    Bytes nb = net.receive_response(0);
    Reader rd(nb);
    int num_rounds = (int)rd.u64();
    std::vector<fe> r_sumcheck = coordinate_prove_arbitrary(net, tr, num_rounds, lp.proof);
    receive_final_claims(net, lp.left_claim, lp.right_claim);
    tr.append_scalar(lp.left_claim);
    tr.append_scalar(lp.right_claim);
    r_grand_product.assign(r_sumcheck.rbegin(), r_sumcheck.rend());
    fe r_layer = tr.challenge_scalar();
    Writer w;
    w.fr(r_layer);
    net.broadcast_request(w.b);
    claim = Fr::add(lp.left_claim, Fr::mul(r_layer, Fr::sub(lp.right_claim, lp.left_claim)));
    r_grand_product.push_back(r_layer);
    return lp;
}

// cooridinate_prove_grand_product (grand_product.rs:56-85) -> proof, final (claim, r)
static GrandProductProof coordinate_prove_grand_product(StarNetCoordinator& net, Transcript& tr, size_t num_layers, fe& claim_out,
                                                        std::vector<fe>& r_out) {
    GrandProductProof proof;
    std::vector<std::vector<fe>> parts;
    for (Bytes& b : net.receive_responses()) {
        Reader rd(b);
        parts.push_back(rd.vec_fr());
    }
    proof.outputs = combine_additive(parts);
    tr.append_scalars(proof.outputs);
    // DensePolynomial::new_padded(outputs).evaluate(r)
    std::vector<fe> padded = proof.outputs;
    while (padded.size() & (padded.size() - 1)) padded.push_back(Fr::zero());
    int nv = 0;
    while (((size_t)1 << nv) < padded.size()) nv++;
    std::vector<fe> r = tr.challenge_vector(nv);
    std::vector<fe> eq = eq_evals_host(r);
    fe claim = Fr::zero();
    for (size_t i = 0; i < padded.size(); i++) claim = Fr::add(claim, Fr::mul(eq[i], padded[i]));
    Writer w;
    w.vec_fr(r);
    w.fr(claim);
    net.broadcast_request(w.b);
    for (size_t i = 0; i < num_layers; i++) proof.gkr_layers.push_back(coordinate_prove_layer(net, tr, claim, r));
    claim_out = claim;
    r_out = r;
    return proof;
}

// plain verifier of the GKR proof (replays the transcript; checks g(0)+g(1) = claim implicitly via
// decompression and the layer reduction eq(r, r') L R = g(r')).  Returns false on any mismatch.
static bool verify_grand_product(const GrandProductProof& proof, Transcript& tr, fe& claim_out, std::vector<fe>& r_out) {
    tr.append_scalars(proof.outputs);
    std::vector<fe> padded = proof.outputs;
    while (padded.size() & (padded.size() - 1)) padded.push_back(Fr::zero());
    int nv = 0;
    while (((size_t)1 << nv) < padded.size()) nv++;
    std::vector<fe> r = tr.challenge_vector(nv);
    std::vector<fe> eqv = eq_evals_host(r);
    fe claim = Fr::zero();
    for (size_t i = 0; i < padded.size(); i++) claim = Fr::add(claim, Fr::mul(eqv[i], padded[i]));
    fe one = Fr::one();
    for (const GrandProductLayerProof& lp : proof.gkr_layers) {
        std::vector<fe> rs;
        fe e = claim;
        for (const auto& comp : lp.proof.compressed_polys) {
            std::vector<fe> poly = unipoly_decompress(comp, e);
            tr.append_scalars(comp);
            fe r_j = tr.challenge_scalar();
            rs.push_back(r_j);
            e = unipoly_eval(poly, r_j);
        }
        if (rs.size() != r.size()) return false;
        fe eq = one;
        for (size_t i = 0; i < r.size(); i++) {
            const fe& a = r[i];
            const fe& b = rs[rs.size() - 1 - i];
            fe ab = Fr::mul(a, b);
            // a b + (1-a)(1-b) = 1 - a - b + 2ab
            fe t = Fr::add(Fr::sub(Fr::sub(one, a), b), Fr::dbl(ab));
            eq = Fr::mul(eq, t);
        }
        if (!Fr::eq(Fr::mul(Fr::mul(eq, lp.left_claim), lp.right_claim), e)) return false;
        tr.append_scalar(lp.left_claim);
        tr.append_scalar(lp.right_claim);
        r.assign(rs.rbegin(), rs.rend());
        fe r_layer = tr.challenge_scalar();
        claim = Fr::add(lp.left_claim, Fr::mul(r_layer, Fr::sub(lp.right_claim, lp.left_claim)));
        r.push_back(r_layer);
    }
    claim_out = claim;
    r_out = r;
    return true;
}

// ================================================================= toggled (sparse) batched grand product
// Rep3ToggledBatchedGrandProduct (co-jolt/src/subprotocols/sparse_grand_product.rs:890-1020): one toggle layer (flags x
// fingerprints) under tree_depth Rep3SparseInterleavedPolynomial layers.  The sparse layers are kept dense on the device
// (toggle_layer.inc), so they are exactly a Rep3BatchedDenseGrandProduct over the toggle layer's output; the toggle layer
// has its own round function and, unlike a multiplication layer, no r_layer fold after its sumcheck (:850-873).
struct ToggleH {
    cozk_toggle* h = nullptr;
    ToggleH() {}
    explicit ToggleH(cozk_toggle* t) : h(t) {}
    ToggleH(const ToggleH&) = delete;
    ToggleH& operator=(const ToggleH&) = delete;
    ToggleH(ToggleH&& o) noexcept : h(o.h) { o.h = nullptr; }
    ToggleH& operator=(ToggleH&& o) noexcept {
        if (this != &o) {
            cozk_toggle_free(h);
            h = o.h;
            o.h = nullptr;
        }
        return *this;
    }
    ~ToggleH() { cozk_toggle_free(h); }
};

struct Rep3ToggledBatchedGrandProduct {
    ToggleH toggle_layer;
    Rep3BatchedDenseGrandProduct sparse_layers;

    // construct (sparse_grand_product.rs:905-930): sparse_layers[0] = toggle_layer.layer_output(), then layer_output up the tree
    static Rep3ToggledBatchedGrandProduct construct(WorkerEnv& env, ToggleH toggle) {
        Rep3ToggledBatchedGrandProduct gp;
        cozk_layer* l0 = nullptr;
        rc_check(cozk_toggle_layer_output(env.ctx, toggle.h, env.party, &l0), env.ctx, "toggle_layer_output");
        size_t batch = cozk_toggle_batch(toggle.h);
        gp.sparse_layers = Rep3BatchedDenseGrandProduct::construct(env, LayerH(l0), batch);
        gp.toggle_layer = std::move(toggle);
        return gp;
    }
    size_t num_layers() const { return sparse_layers.layers.size() + 1; }

    // prove_layer of the toggle layer (:850-873) with prove_sumcheck (sumcheck.rs:96-131) over cozk_toggle_round
    static void prove_toggle_layer(WorkerEnv& env, cozk_toggle* t, const fe& claim_in, std::vector<fe>& r_grand_product) {
        EqH eq;
        std::vector<uint64_t> w = to_abi(r_grand_product);
        int num_rounds = (int)r_grand_product.size();
        rc_check(cozk_spliteq_new(env.ctx, w.data(), num_rounds, &eq.h), env.ctx, "spliteq_new");
        if (env.party == 0) {
            Writer wr;
            wr.u64((uint64_t)num_rounds);
            env.star->send_response(wr.b);
        }
        fe previous_claim = claim_in;
        std::vector<fe> rs;
        uint64_t rr[4];
        for (int round = 0; round < num_rounds; round++) {
            uint64_t ev[12];
            rc_check(cozk_toggle_round(env.ctx, t, eq.h, round ? rr : nullptr, env.party, ev), env.ctx, "toggle_round");
            fe g0 = fe_from_u64x4(ev);
            fe pts[4] = {g0, Fr::sub(previous_claim, g0), fe_from_u64x4(ev + 4), fe_from_u64x4(ev + 8)};
            std::vector<fe> cf(4);
            unipoly_from_evals(pts, 4, cf.data());
            Writer wr;
            wr.vec_fr(cf);
            env.star->send_response(wr.b);
            Bytes req = env.star->receive_request();
            Reader rd(req);
            fe r_j = rd.fr();
            previous_claim = env.additive_trivial(rd.fr());
            rs.push_back(r_j);
            fe_to_u64x4(r_j, rr);
        }
        if (num_rounds > 0) rc_check(cozk_toggle_bind(env.ctx, t, rr), env.ctx, "toggle_bind");
        // final_claims (:825-835): (promote_to_trivial_share(flag), fingerprint)
        uint64_t fl[4], pa[4], pb[4];
        rc_check(cozk_toggle_final_claims(env.ctx, t, fl, pa, pb), env.ctx, "toggle_final_claims");
        Share flag = env.rep3_trivial(fe_from_u64x4(fl));
        Writer wr;
        wr.fr(flag.a);
        wr.fr(flag.b);
        wr.fr(fe_from_u64x4(pa));
        wr.fr(fe_from_u64x4(pb));
        env.star->send_response(wr.b);
        r_grand_product.assign(rs.rbegin(), rs.rend());
    }

    // prove_grand_product_worker (grand_product.rs:111-130) over layers() = [toggle, sparse...].rev() (:947-960)
    std::vector<fe> prove_grand_product_worker(WorkerEnv& env) {
        std::vector<fe> outputs = sparse_layers.claimed_outputs(env);
        Writer w;
        w.vec_fr(outputs);
        env.star->send_response(w.b);
        Bytes req = env.star->receive_request();
        Reader rd(req);
        std::vector<fe> r = rd.vec_fr();
        fe claim = env.additive_trivial(rd.fr());
        for (size_t i = sparse_layers.layers.size(); i-- > 0;) Rep3BatchedDenseGrandProduct::prove_layer(env, sparse_layers.layers[i].h, claim, r);
        prove_toggle_layer(env, toggle_layer.h, claim, r);
        return r;
    }
};

// coordinate_prove_layer of the toggle layer (sparse_grand_product.rs:876-903): no r_layer challenge, no claim fold
static GrandProductLayerProof coordinate_prove_toggle_layer(StarNetCoordinator& net, Transcript& tr, std::vector<fe>& r_grand_product) {
    GrandProductLayerProof lp;
    Bytes nb = net.receive_response(0);
    Reader rd(nb);
    int num_rounds = (int)rd.u64();
    std::vector<fe> r_sumcheck = coordinate_prove_arbitrary(net, tr, num_rounds, lp.proof);
    receive_final_claims(net, lp.left_claim, lp.right_claim);
    tr.append_scalar(lp.left_claim);
    tr.append_scalar(lp.right_claim);
    r_grand_product.assign(r_sumcheck.rbegin(), r_sumcheck.rend());
    return lp;
}

// cooridinate_prove_grand_product (grand_product.rs:56-85) for the toggled circuit: num_layers - 1 multiplication layers,
// then the toggle layer.  flag_claim / fingerprint_claim = the toggle layer's final claims at r_out.
static GrandProductProof coordinate_prove_toggled_grand_product(StarNetCoordinator& net, Transcript& tr, size_t num_layers, std::vector<fe>& r_out) {
    GrandProductProof proof;
    std::vector<std::vector<fe>> parts;
    for (Bytes& b : net.receive_responses()) {
        Reader rd(b);
        parts.push_back(rd.vec_fr());
    }
    proof.outputs = combine_additive(parts);
    tr.append_scalars(proof.outputs);
    std::vector<fe> padded = proof.outputs;
    while (padded.size() & (padded.size() - 1)) padded.push_back(Fr::zero());
    int nv = 0;
    while (((size_t)1 << nv) < padded.size()) nv++;
    std::vector<fe> r = tr.challenge_vector(nv);
    std::vector<fe> eq = eq_evals_host(r);
    fe claim = Fr::zero();
    for (size_t i = 0; i < padded.size(); i++) claim = Fr::add(claim, Fr::mul(eq[i], padded[i]));
    Writer w;
    w.vec_fr(r);
    w.fr(claim);
    net.broadcast_request(w.b);
    for (size_t i = 0; i + 1 < num_layers; i++) proof.gkr_layers.push_back(coordinate_prove_layer(net, tr, claim, r));
    proof.gkr_layers.push_back(coordinate_prove_toggle_layer(net, tr, r));
    r_out = r;
    return proof;
}

// plain verifier (jolt-core ToggledBatchedGrandProduct::verify_sumcheck_claim, out of tree): multiplication layers check
// eq * L * R and fold with r_layer, the toggle layer (last) checks eq * (flag * fingerprint + 1 - flag)
static bool verify_toggled_grand_product(const GrandProductProof& proof, Transcript& tr, fe& flag_claim, fe& fingerprint_claim, std::vector<fe>& r_out) {
    tr.append_scalars(proof.outputs);
    std::vector<fe> padded = proof.outputs;
    while (padded.size() & (padded.size() - 1)) padded.push_back(Fr::zero());
    int nv = 0;
    while (((size_t)1 << nv) < padded.size()) nv++;
    std::vector<fe> r = tr.challenge_vector(nv);
    std::vector<fe> eqv = eq_evals_host(r);
    fe claim = Fr::zero();
    for (size_t i = 0; i < padded.size(); i++) claim = Fr::add(claim, Fr::mul(eqv[i], padded[i]));
    fe one = Fr::one();
    if (proof.gkr_layers.empty()) return false;
    for (size_t li = 0; li < proof.gkr_layers.size(); li++) {
        const GrandProductLayerProof& lp = proof.gkr_layers[li];
        std::vector<fe> rs;
        fe e = claim;
        for (const auto& comp : lp.proof.compressed_polys) {
            std::vector<fe> poly = unipoly_decompress(comp, e);
            tr.append_scalars(comp);
            fe r_j = tr.challenge_scalar();
            rs.push_back(r_j);
            e = unipoly_eval(poly, r_j);
        }
        if (rs.size() != r.size()) return false;
        fe eq = one;
        for (size_t i = 0; i < r.size(); i++) {
            const fe& a = r[i];
            const fe& b = rs[rs.size() - 1 - i];
            eq = Fr::mul(eq, Fr::add(Fr::sub(Fr::sub(one, a), b), Fr::dbl(Fr::mul(a, b))));
        }
        tr.append_scalar(lp.left_claim);
        tr.append_scalar(lp.right_claim);
        r.assign(rs.rbegin(), rs.rend());
        if (li + 1 < proof.gkr_layers.size()) {
            if (!Fr::eq(Fr::mul(Fr::mul(eq, lp.left_claim), lp.right_claim), e)) return false;
            fe r_layer = tr.challenge_scalar();
            claim = Fr::add(lp.left_claim, Fr::mul(r_layer, Fr::sub(lp.right_claim, lp.left_claim)));
            r.push_back(r_layer);
        } else {
            fe node = Fr::add(Fr::mul(lp.left_claim, lp.right_claim), Fr::sub(one, lp.left_claim));
            if (!Fr::eq(Fr::mul(eq, node), e)) return false;
            flag_claim = lp.left_claim;
            fingerprint_claim = lp.right_claim;
        }
    }
    r_out = r;
    return true;
}

// ================================================================= Lasso primary sumcheck (instruction lookups)
struct PrimaryH {
    cozk_primary* h = nullptr;
    PrimaryH() {}
    explicit PrimaryH(cozk_primary* p) : h(p) {}
    PrimaryH(const PrimaryH&) = delete;
    PrimaryH& operator=(const PrimaryH&) = delete;
    PrimaryH(PrimaryH&& o) noexcept : h(o.h) { o.h = nullptr; }
    ~PrimaryH() { cozk_primary_free(h); }
};

// UniPoly::from_evals for any number of points 0, 1, .., n-1 (Lagrange with small-integer denominators)
static inline std::vector<fe> unipoly_from_evals_general(const std::vector<fe>& ev) {
    const size_t n = ev.size();
    std::vector<fe> coeffs(n, Fr::zero());
    for (size_t i = 0; i < n; i++) {
        std::vector<fe> num(1, Fr::one());  // prod_{j != i} (x - j), low -> high
        fe den = Fr::one();
        for (size_t j = 0; j < n; j++) {
            if (j == i) continue;
            std::vector<fe> nx(num.size() + 1, Fr::zero());
            fe jj = Fr::from_u64((uint64_t)j);
            for (size_t k = 0; k < num.size(); k++) {
                nx[k + 1] = Fr::add(nx[k + 1], num[k]);
                nx[k] = Fr::sub(nx[k], Fr::mul(jj, num[k]));
            }
            num.swap(nx);
            fe d = i > j ? Fr::from_u64((uint64_t)(i - j)) : Fr::neg(Fr::from_u64((uint64_t)(j - i)));
            den = Fr::mul(den, d);
        }
        fe sc = Fr::mul(ev[i], Fr::inv(den));
        for (size_t k = 0; k < n; k++) coeffs[k] = Fr::add(coeffs[k], Fr::mul(sc, num[k]));
    }
    return coeffs;
}

// prove_primary_sumcheck_inner (jolt/vm/instruction_lookups/worker.rs:375-452): per round the `degree` additive evaluations
// go up, r_j comes down; every multiplication level of the collations is one mul_vec over all items (local half on the
// device, ring reshare here).  Afterwards the final claims are sent as additive shares in the order E, flags, outputs
// (the primary_sumcheck_openings of worker.rs:128-141).  Returns the challenges in round order.
struct PrimaryFinals {
    std::vector<Share> E;      // E_m(r) as shares
    std::vector<fe> flags;     // flag_i(r), public
    Share outputs;
    fe eq;
};
// the rounds of prove_primary_sumcheck_inner over ONE polynomial set (a whole trace, a worker's chunk, or the 2^k gathered
// finals): sends the evaluations, receives r_j, returns the challenges and the set's final values
static std::vector<fe> prove_primary_rounds(WorkerEnv& env, StarNetWorker* star, cozk_primary* prim, int num_rounds, size_t n_mem, size_t n_instr,
                                            PrimaryFinals& fin) {
    const int D = cozk_primary_degree(prim);
    std::vector<fe> rs;
    uint64_t rr[4];
    for (int round = 0; round < num_rounds; round++) {
        size_t n_items = 0;
        int n_levels = 0;
        rc_check(cozk_primary_round_begin(env.ctx, prim, round ? rr : nullptr, &n_items, &n_levels), env.ctx, "primary_round_begin");
        for (int level = 1; level <= n_levels; level++) {
            const void* send = nullptr;
            void* recv = nullptr;
            size_t n = 0;
            rc_check(cozk_primary_level(env.ctx, prim, level, env.key_self, env.key_prev, env.mask_ctr, &send, &recv, &n), env.ctx, "primary_level");
            if (env.mode == COZK_MODE_REP3 && n) {  // n == 0: no active instruction has a value to reshare at this level (public knowledge)
                env.ring->reshare(env.ctx, (const fe*)send, (fe*)recv, n);
                env.mask_ctr += n;
            }
        }
        std::vector<uint64_t> ev(4 * (size_t)D);
        rc_check(cozk_primary_round_finish(env.ctx, prim, ev.data()), env.ctx, "primary_round_finish");
        std::vector<fe> msg((size_t)D);
        for (int k = 0; k < D; k++) msg[k] = fe_from_u64x4(ev.data() + 4 * k);
        Writer w;
        w.vec_fr(msg);
        star->send_response(w.b);
        Bytes req = star->receive_request();
        Reader rd(req);
        fe r_j = rd.fr();
        rs.push_back(r_j);
        fe_to_u64x4(r_j, rr);
    }
    std::vector<uint64_t> Ee(8 * n_mem), Fe(4 * n_instr);
    uint64_t oe[8], qe[4];
    rc_check(cozk_primary_final_evals(env.ctx, prim, rr, Ee.data(), Fe.data(), oe, qe), env.ctx, "primary_final_evals");
    fin.E.clear();
    fin.flags.clear();
    for (size_t m = 0; m < n_mem; m++) fin.E.push_back(Share{fe_from_u64x4(Ee.data() + 8 * m), fe_from_u64x4(Ee.data() + 8 * m + 4)});
    for (size_t i = 0; i < n_instr; i++) fin.flags.push_back(fe_from_u64x4(Fe.data() + 4 * i));
    fin.outputs = Share{fe_from_u64x4(oe), fe_from_u64x4(oe + 4)};
    fin.eq = fe_from_u64x4(qe);
    return rs;
}
// the final claims as additive shares in the order E, flags, outputs (the primary_sumcheck_openings of worker.rs:128-141)
static void send_primary_openings(WorkerEnv& env, StarNetWorker* star, const PrimaryFinals& fin) {
    std::vector<fe> openings;
    for (const Share& e : fin.E) openings.push_back(env.into_additive(e));
    for (const fe& f : fin.flags) openings.push_back(env.mode == COZK_MODE_REP3 ? env.additive_trivial(f) : f);
    openings.push_back(env.into_additive(fin.outputs));
    Writer w;
    w.vec_fr(openings);
    star->send_response(w.b);
}
static std::vector<fe> prove_primary_sumcheck_worker(WorkerEnv& env, cozk_primary* prim, int num_rounds, size_t n_mem, size_t n_instr) {
    PrimaryFinals fin;
    std::vector<fe> rs = prove_primary_rounds(env, env.star, prim, num_rounds, n_mem, n_instr, fin);
    send_primary_openings(env, env.star, fin);
    return rs;
}

struct PrimarySumcheckProof {
    std::vector<std::vector<fe>> compressed_polys;
    std::vector<fe> openings;  // E(r) (n_mem), flags(r) (n_instr), lookup_outputs(r)
};

// prove_primary_sumcheck_rep3 (jolt/vm/instruction_lookups/coordinator.rs:97-150) + the final claims.  n_participants = parties x
// workers (participant id = worker * parties + party); during the first num_rounds - log_workers rounds every worker
// sub-net responds and the coordinator adds all of them (:112-128), afterwards only worker 0 of every party does (:107-111)
static PrimarySumcheckProof coordinate_primary_sumcheck(StarNetCoordinator& net, Transcript& tr, int num_rounds, std::vector<fe>& r_out, int nparties = 0,
                                                        int log_workers = 0) {
    PrimarySumcheckProof proof;
    fe previous_claim = Fr::zero();
    r_out.clear();
    if (nparties <= 0) nparties = net.n_workers();
    const int all = net.n_workers();
    auto gather = [&](int count) {
        std::vector<std::vector<fe>> parts;
        for (int id = 0; id < count; id++) {
            Bytes b = net.receive_response(id);
            Reader rd(b);
            parts.push_back(rd.vec_fr());
        }
        return combine_additive(parts);
    };
    for (int round = 0; round < num_rounds; round++) {
        const bool split_round = round < num_rounds - log_workers;
        std::vector<fe> ev = gather(split_round ? all : nparties);
        ev.insert(ev.begin() + 1, Fr::sub(previous_claim, ev[0]));  // round_evals.insert(1, previous_claim - round_evals[0])
        std::vector<fe> poly = unipoly_from_evals_general(ev);
        std::vector<fe> comp = unipoly_compress(poly);
        tr.append_scalars(comp);
        proof.compressed_polys.push_back(comp);
        fe r_j = tr.challenge_scalar();
        Writer w;
        w.fr(r_j);
        for (int id = 0; id < (split_round ? all : nparties); id++) net.send_request(id, w.b);
        r_out.push_back(r_j);
        previous_claim = unipoly_eval(poly, r_j);
    }
    proof.openings = gather(nparties);
    tr.append_scalars(proof.openings);
    return proof;
}

// g_i on opened values (the verifier's side: the PLAIN combine_lookups of co-jolt/src/jolt/instruction/*.rs, written from the
// formulas there and independently of the device's collation programs)
static inline fe primary_g_plain(const cozk_primary_instr& in, const std::vector<fe>& E) {
    const fe one = Fr::one();
    const int n = in.n_mems;
    auto v = [&](int pos) { return E[in.mems[pos]]; };
    auto prod = [&](int p0, int cnt) {
        fe acc = one;
        for (int t = 0; t < cnt; t++) acc = Fr::mul(acc, v(p0 + t));
        return acc;
    };
    // sum_{i < C} ltu_i prod_{j < i} eq_j; *eq_prod = prod_{j < n_eq} eq_j  (sltu.rs:32-47, virtual_assert_lte.rs:33-50)
    auto ltu_sum = [&](int C, int l0, int e0, int n_eq, fe* eq_prod) {
        fe s = Fr::zero(), pr = one;
        for (int i = 0; i < C; i++) {
            s = Fr::add(s, Fr::mul(v(l0 + i), pr));
            if (i < n_eq) pr = Fr::mul(pr, v(e0 + i));
        }
        if (eq_prod) *eq_prod = pr;
        return s;
    };
    switch (in.form) {
        case COZK_G_CONCAT: {  // utils/instruction_utils.rs concatenate_lookups
            fe shift = one, two = Fr::from_u64(2);
            for (int b = 0; b < in.bits; b++) shift = Fr::mul(shift, two);
            fe acc = Fr::zero(), w = one;
            for (int t = n - 1; t >= 0; t--) {
                acc = Fr::add(acc, Fr::mul(v(t), w));
                w = Fr::mul(w, shift);
            }
            return acc;
        }
        case COZK_G_PRODUCT: return prod(0, n);                      // beq.rs:35-37
        case COZK_G_NOT_PRODUCT: return Fr::sub(one, prod(0, n));    // bne.rs:33-35
        case COZK_G_LTU: return ltu_sum((n + 1) / 2, 0, (n + 1) / 2, (n + 1) / 2 - 1, nullptr);
        case COZK_G_NOT_LTU: return Fr::sub(one, ltu_sum((n + 1) / 2, 0, (n + 1) / 2, (n + 1) / 2 - 1, nullptr));  // bgeu.rs:31-39
        case COZK_G_SLT:
        case COZK_G_NOT_SLT: {  // slt.rs:33-59
            const int C = (n - 1) / 2;
            const fe l = v(0), r = v(1), lt_abs = v(2 * C - 1), eq_abs = v(2 * C);
            fe s = lt_abs, pr = eq_abs;
            for (int i = 0; i <= C - 2; i++) {
                s = Fr::add(s, Fr::mul(v(2 + i), pr));
                if (i < C - 2) pr = Fr::mul(pr, v(C + 1 + i));
            }
            fe eq_s = Fr::add(Fr::mul(l, r), Fr::mul(Fr::sub(one, l), Fr::sub(one, r)));
            fe g = Fr::add(Fr::mul(l, Fr::sub(one, r)), Fr::mul(eq_s, s));
            return in.form == COZK_G_SLT ? g : Fr::sub(one, g);  // bge.rs:34-43
        }
        case COZK_G_LTE: {  // virtual_assert_lte.rs:33-50
            fe pr;
            fe s = ltu_sum(n / 2, 0, n / 2, n / 2, &pr);
            return Fr::add(s, pr);
        }
        case COZK_G_NOT_FIRST: return Fr::sub(one, v(0));  // virtual_assert_halfword_alignment.rs:33-37
        case COZK_G_DIV0: return Fr::add(Fr::sub(one, prod(0, n / 2)), prod(n / 2, n / 2));  // virtual_assert_valid_div0.rs:36-42
        case COZK_G_UNSIGNED_REM: {  // virtual_assert_valid_unsigned_remainder.rs:30-45
            const int C = (n + 1) / 3;
            return Fr::add(ltu_sum(C, 0, C, C - 1, nullptr), prod(2 * C - 1, C));
        }
        case COZK_G_SIGNED_REM: {  // virtual_assert_valid_signed_remainder.rs:40-67
            const int C = (n - 2) / 4;
            const fe l = v(0), r = v(1);
            fe s = v(2 * C + 1), pr = v(2 * C);  // lt_abs, eq_abs
            for (int i = 0; i <= C - 2; i++) {
                s = Fr::add(s, Fr::mul(v(C + 1 + i), pr));
                pr = Fr::mul(pr, v(2 + i));
            }
            fe rem_zero = prod(2 * C + 2, C), div_zero = prod(3 * C + 2, C);
            fe g = Fr::mul(Fr::sub(Fr::sub(one, l), r), s);
            g = Fr::add(g, Fr::mul(Fr::mul(l, r), Fr::sub(one, pr)));
            g = Fr::add(g, Fr::mul(Fr::mul(Fr::sub(one, l), r), rem_zero));
            return Fr::add(g, div_zero);
        }
        default: return Fr::zero();  // COZK_G_ZERO: virtual_pow2.rs:20-22
    }
}

// g_poly_degree of a form (co-jolt/src/jolt/instruction/*.rs; SLT / BGE: C + 2, the true degree -- slt.rs:61-63 says C + 1,
// which the RV32I set's maximum C + 2 of the signed remainder covers) and sumcheck_poly_degree (worker.rs:701-708)
static inline int primary_g_degree(const cozk_primary_instr& in) {
    const int n = in.n_mems;
    switch (in.form) {
        case COZK_G_CONCAT:
        case COZK_G_NOT_FIRST:
        case COZK_G_ZERO: return 1;
        case COZK_G_PRODUCT:
        case COZK_G_NOT_PRODUCT: return n;
        case COZK_G_LTU:
        case COZK_G_NOT_LTU: return (n + 1) / 2;
        case COZK_G_LTE:
        case COZK_G_DIV0: return n / 2;
        case COZK_G_UNSIGNED_REM: return (n + 1) / 3;
        case COZK_G_SLT:
        case COZK_G_NOT_SLT: return (n - 1) / 2 + 2;
        default: return (n - 2) / 4 + 2;  // COZK_G_SIGNED_REM
    }
}
static inline int primary_sumcheck_degree(const std::vector<cozk_primary_instr>& instrs) {
    int g = 1;
    for (const auto& in : instrs) g = std::max(g, primary_g_degree(in));
    return g + 2;  // eq and flag
}

// plain verifier of the primary sumcheck (jolt-core verify_primary_sumcheck, out of tree): replay, then
// claim == eq(r_eq, r) (sum_i flag_i(r) g_i(E(r)) - outputs(r)); the point is the reversed challenge list (LowToHigh)
static bool verify_primary_sumcheck(const PrimarySumcheckProof& proof, const std::vector<cozk_primary_instr>& instrs, size_t n_mem, int degree,
                                    const std::vector<fe>& r_eq, Transcript& tr, std::vector<fe>& r_out) {
    fe claim = Fr::zero();
    r_out.clear();
    for (const auto& comp : proof.compressed_polys) {
        if ((int)comp.size() != degree) return false;
        std::vector<fe> poly = unipoly_decompress(comp, claim);
        tr.append_scalars(comp);
        fe r_j = tr.challenge_scalar();
        r_out.push_back(r_j);
        claim = unipoly_eval(poly, r_j);
    }
    if (proof.openings.size() != n_mem + instrs.size() + 1 || r_out.size() != r_eq.size()) return false;
    tr.append_scalars(proof.openings);
    std::vector<fe> E(proof.openings.begin(), proof.openings.begin() + n_mem);
    fe one = Fr::one(), eq = one;
    for (size_t i = 0; i < r_eq.size(); i++) {
        const fe& a = r_eq[i];
        const fe& b = r_out[r_out.size() - 1 - i];
        eq = Fr::mul(eq, Fr::add(Fr::sub(Fr::sub(one, a), b), Fr::dbl(Fr::mul(a, b))));
    }
    fe acc = Fr::zero();
    for (size_t i = 0; i < instrs.size(); i++) acc = Fr::add(acc, Fr::mul(proof.openings[n_mem + i], primary_g_plain(instrs[i], E)));
    return Fr::eq(Fr::mul(eq, Fr::sub(acc, proof.openings.back())), claim);
}

// ================================================================= co-jolt Spartan outer sumcheck
struct OuterH {
    cozk_outer* h = nullptr;
    OuterH() {}
    explicit OuterH(cozk_outer* p) : h(p) {}
    OuterH(const OuterH&) = delete;
    OuterH& operator=(const OuterH&) = delete;
    ~OuterH() { cozk_outer_free(h); }
};

// prove_spartan_cubic_sumcheck (co-jolt/src/r1cs/spartan/worker.rs:277-300): per round the cubic goes up
// (process_eq_sumcheck_round_worker, subprotocols/sumcheck_spartan.rs:44-79), (next claim, r_i) comes down; at the end the
// three final evaluations.  Returns the challenges in round order.
static std::vector<fe> prove_spartan_cubic_sumcheck_worker(WorkerEnv& env, cozk_outer* st, int num_rounds) {
    std::vector<fe> rs;
    fe claim = Fr::zero();
    uint64_t rr[4];
    for (int round = 0; round < num_rounds; round++) {
        uint64_t cl[4], cf[16];
        fe_to_u64x4(claim, cl);
        rc_check(cozk_outer_round(env.ctx, st, round ? rr : nullptr, cl, cf), env.ctx, "outer_round");
        std::vector<fe> poly(4);
        for (int i = 0; i < 4; i++) poly[i] = fe_from_u64x4(cf + 4 * i);
        Writer w;
        w.vec_fr(poly);
        env.star->send_response(w.b);
        Bytes req = env.star->receive_request();
        Reader rd(req);
        fe next_claim = rd.fr();
        fe r_i = rd.fr();
        rs.push_back(r_i);
        claim = env.additive_trivial(next_claim);
        fe_to_u64x4(r_i, rr);
    }
    uint64_t fin[12];
    rc_check(cozk_outer_final_evals(env.ctx, st, rr, fin), env.ctx, "outer_final_evals");
    std::vector<fe> ev(3);
    for (int i = 0; i < 3; i++) ev[i] = fe_from_u64x4(fin + 4 * i);
    Writer w;
    w.vec_fr(ev);
    env.star->send_response(w.b);
    return rs;
}

struct OuterSumcheckProof {
    std::vector<std::vector<fe>> compressed_polys;
    std::vector<fe> claims;  // Az(r), Bz(r), Cz(r)
};

// coordinate_eq_sumcheck_round (subprotocols/sumcheck_spartan.rs:14-42) over all rounds + the outer claims
// (r1cs/spartan/coordinator.rs:41-63)
static OuterSumcheckProof coordinate_outer_sumcheck(StarNetCoordinator& net, Transcript& tr, int num_rounds, std::vector<fe>& r_out) {
    OuterSumcheckProof proof;
    r_out.clear();
    for (int round = 0; round < num_rounds; round++) {
        std::vector<std::vector<fe>> parts;
        for (Bytes& b : net.receive_responses()) {
            Reader rd(b);
            parts.push_back(rd.vec_fr());
        }
        std::vector<fe> poly = combine_additive(parts);
        std::vector<fe> comp = unipoly_compress(poly);
        tr.append_scalars(comp);
        proof.compressed_polys.push_back(comp);
        fe r_i = tr.challenge_scalar();
        r_out.push_back(r_i);
        fe claim = unipoly_eval(poly, r_i);
        Writer w;
        w.fr(claim);
        w.fr(r_i);
        net.broadcast_request(w.b);
    }
    std::vector<std::vector<fe>> parts;
    for (Bytes& b : net.receive_responses()) {
        Reader rd(b);
        parts.push_back(rd.vec_fr());
    }
    proof.claims = combine_additive(parts);
    tr.append_scalars(proof.claims);
    return proof;
}

// outer-sumcheck part of the plain verifier (jolt-core UniformSpartanProof::verify, out of tree)
static bool verify_outer_sumcheck(const OuterSumcheckProof& proof, const std::vector<fe>& tau, Transcript& tr, std::vector<fe>& r_out) {
    fe claim = Fr::zero();
    r_out.clear();
    for (const auto& comp : proof.compressed_polys) {
        if (comp.size() != 3) return false;
        std::vector<fe> poly = unipoly_decompress(comp, claim);
        tr.append_scalars(comp);
        fe r_i = tr.challenge_scalar();
        r_out.push_back(r_i);
        claim = unipoly_eval(poly, r_i);
    }
    if (proof.claims.size() != 3 || r_out.size() != tau.size()) return false;
    tr.append_scalars(proof.claims);
    fe one = Fr::one(), eq = one;
    for (size_t i = 0; i < tau.size(); i++) {
        const fe& a = tau[i];
        const fe& b = r_out[r_out.size() - 1 - i];
        eq = Fr::mul(eq, Fr::add(Fr::sub(Fr::sub(one, a), b), Fr::dbl(Fr::mul(a, b))));
    }
    return Fr::eq(Fr::mul(eq, Fr::sub(Fr::mul(proof.claims[0], proof.claims[1]), proof.claims[2])), claim);
}

// ================================================================= PST13
struct PST13Commitment {
    uint64_t nv;
    g1_affine g_product;
};

// SRS levels `ck.powers_of_g[i]` (size 2^(nv-i)) live concatenated in one device handle; `halves`
// holds G[2b] + G[2b+1] of the same array, i.e. the level-i bases for the duplicated scalars
// q[x >> 1] of `open` (pst13.rs:459) at offset level_offset(i) / 2.
struct PST13Setup {
    int nv = 0;
    cozk_bases* powers_all = nullptr;  // levels 0..nv-1 concatenated: N, N/2, ..., 2  (2N - 2 points)
    cozk_bases* halves = nullptr;      // pair sums of powers_all (N - 1 points)
    g1_affine g;                       // generator
    std::vector<fe> trapdoor;          // t (harness / tests only: pairing-free `check`)
    size_t level_offset(int i) const { return ((size_t)1 << (nv + 1)) - ((size_t)1 << (nv - i + 1)); }
    ~PST13Setup() {
        cozk_bases_free(powers_all);
        cozk_bases_free(halves);
    }
};

static inline g1_affine abi_to_g1(const uint64_t* xy, int inf) {
    g1_affine a;
    a.x = fe_from_u64x4(xy);
    a.y = fe_from_u64x4(xy + 4);
    if (inf) {
        a.x = Fq::zero();
        a.y = Fq::zero();
    }
    return a;
}

struct PST13 {
    // MultilinearPC::setup(nv, rng) with the trapdoor t supplied (PST13::setup, pst13.rs:49-62).
    // powers_of_g[i][b] = g^{eq_le(t[i..], b)}: index bit j of b pairs with t[i + j].
    static std::unique_ptr<PST13Setup> setup(cozk_ctx* ctx, const std::vector<fe>& t, int precompute = 1, const g1_affine* generator = nullptr) {
        std::unique_ptr<PST13Setup> s(new PST13Setup());
        int nv = (int)t.size();
        COZK_REQUIRE(nv >= 1 && nv <= 26, "PST13::setup: nv out of range");
        s->nv = nv;
        s->trapdoor = t;
        s->g.x = Fq::one();
        s->g.y = Fq::from_u64(2);
        // a worker sub-net's slice of a larger SRS is the local table over its low variables with the
        // generator pre-multiplied by eq(t_high, worker index)
        if (generator) s->g = *generator;
        size_t total = ((size_t)1 << (nv + 1)) - 2;
        cozk_vec* sc = nullptr;
        rc_check(cozk_vec_alloc(ctx, total, COZK_SCALAR_FR, &sc), ctx, "vec_alloc");
        VecH scv(sc);
        // level i table = evals_be(reverse(t[i..])), built on device straight into its slice
        for (int i = 0; i < nv; i++) {
            std::vector<fe> rev(t.rbegin(), t.rend() - i);
            std::vector<uint64_t> w = to_abi(rev);
            cozk_vec* ev = nullptr;
            rc_check(cozk_eq_evals(ctx, w.data(), nv - i, &ev), ctx, "eq_evals");
            VecH evh(ev);
            HIP_TRY(hipMemcpyAsync((fe*)cozk_vec_device_ptr(scv.h) + s->level_offset(i), cozk_vec_device_ptr(evh.h),
                                   ((size_t)1 << (nv - i)) * sizeof(fe), hipMemcpyDeviceToDevice, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        uint64_t gxy[8];
        fe_to_u64x4(s->g.x, gxy);
        fe_to_u64x4(s->g.y, gxy + 4);
        rc_check(cozk_bases_from_scalars(ctx, scv.h, gxy, precompute, &s->powers_all), ctx, "bases_from_scalars");
        rc_check(cozk_bases_pair_sums(ctx, s->powers_all, precompute, &s->halves), ctx, "bases_pair_sums");
        return s;
    }

    // batch_commit (pst13.rs:299-331): one MSM per polynomial over powers_of_g[0][..len]
    static std::vector<PST13Commitment> batch_commit(cozk_ctx* ctx, const PST13Setup& s, const std::vector<cozk_vec*>& polys) {
        size_t k = polys.size();
        std::vector<PST13Commitment> out(k);
        if (k == 0) return out;
        std::vector<uint64_t> xy(8 * k);
        std::vector<int> inf(k);
        std::vector<size_t> offs(k, 0);
        rc_check(cozk_batch_msm_slices(ctx, s.powers_all, offs.data(), polys.data(), nullptr, k, xy.data(), inf.data()), ctx, "batch_msm");
        for (size_t i = 0; i < k; i++) {
            size_t len = cozk_vec_len(polys[i]);
            uint64_t nv = 0;
            while (((size_t)1 << nv) < len) nv++;
            out[i].nv = nv;
            out[i].g_product = abi_to_g1(xy.data() + 8 * i, inf[i]);
        }
        return out;
    }

    // `open` (pst13.rs:428-474) on the share-a evaluations; point already reversed by the caller.
    // All nv MSMs go out as one launch set over `halves`.
    static std::vector<g1_affine> open(cozk_ctx* ctx, const PST13Setup& s, const cozk_vec* evals, const std::vector<fe>& point,
                                       fe* final_value = nullptr) {
        int nv = s.nv;
        COZK_REQUIRE(cozk_vec_len(evals) == ((size_t)1 << nv), "PST13::open: invalid size of polynomial");
        COZK_REQUIRE((int)point.size() == nv, "PST13::open: point length");
        std::vector<VecH> q(nv), r(nv);
        const cozk_vec* cur = evals;
        for (int i = 0; i < nv; i++) {
            size_t half = (size_t)1 << (nv - i - 1);
            cozk_vec *qv = nullptr, *rv = nullptr;
            rc_check(cozk_vec_alloc(ctx, half, COZK_SCALAR_FR, &qv), ctx, "vec_alloc");
            q[i] = VecH(qv);
            rc_check(cozk_vec_alloc(ctx, half, COZK_SCALAR_FR, &rv), ctx, "vec_alloc");
            r[i] = VecH(rv);
            uint64_t p[4];
            fe_to_u64x4(point[i], p);
            // the fold reads `cur` (len 2*half): for i > 0 that is r[i-1]
            cozk_vec view = *cur;
            view.n = 2 * half;
            rc_check(cozk_pst_fold(ctx, &view, p, q[i].h, r[i].h), ctx, "pst_fold");
            cur = r[i].h;
        }
        if (final_value) {  // the fully folded value r[0][0] (worker sub-nets hand it on for the last k folds)
            uint64_t v[4];
            rc_check(cozk_vec_download(ctx, nv > 0 ? r[nv - 1].h : evals, v), ctx, "vec_download");
            *final_value = fe_from_u64x4(v);
        }
        std::vector<const cozk_vec*> qs(nv);
        std::vector<size_t> offs(nv);
        for (int i = 0; i < nv; i++) {
            qs[i] = q[i].h;
            offs[i] = s.level_offset(i) / 2;
        }
        std::vector<uint64_t> xy(8 * nv);
        std::vector<int> inf(nv);
        rc_check(cozk_batch_msm_slices(ctx, s.halves, offs.data(), qs.data(), nullptr, nv, xy.data(), inf.data()), ctx, "open msm");
        std::vector<g1_affine> proofs(nv);
        for (int i = 0; i < nv; i++) proofs[i] = abi_to_g1(xy.data() + 8 * i, inf[i]);
        return proofs;
    }

    // prove_rep3 (pst13.rs:125-137): open share-a at the reversed point, send the proof points
    static void prove_rep3(WorkerEnv& env, const PST13Setup& s, cozk_poly* joint, const std::vector<fe>& opening_point) {
        std::vector<fe> rev(opening_point.rbegin(), opening_point.rend());
        cozk_vec* av = nullptr;
        rc_check(cozk_poly_share_view(env.ctx, joint, 0, &av), env.ctx, "share_view");
        VecH a(av);
        std::vector<g1_affine> pf = open(env.ctx, s, a.h, rev);
        Writer w;
        w.vec_g1(pf);
        env.star->send_response(w.b);
    }

    // coordinate_prove (pst13.rs:110-122): point-wise sum of the parties' proofs
    static std::vector<g1_affine> coordinate_prove(StarNetCoordinator& net) {
        std::vector<g1_xyzz> acc;
        for (Bytes& b : net.receive_responses()) {
            Reader rd(b);
            std::vector<g1_affine> pf = rd.vec_g1();
            if (acc.empty()) acc.assign(pf.size(), G1::identity());
            if (acc.size() != pf.size()) throw CozkError(COZK_ERR_INTERNAL, "coordinate_prove: proof length mismatch");
            for (size_t i = 0; i < pf.size(); i++) acc[i] = G1::add_mixed(acc[i], pf[i]);
        }
        std::vector<g1_affine> out(acc.size());
        for (size_t i = 0; i < acc.size(); i++) out[i] = G1::to_affine(acc[i]);
        return out;
    }

    // combine_commitment_shares (pst13.rs:72-108): sum of the parties' share commitments
    static PST13Commitment combine_commitment_shares(const std::vector<PST13Commitment>& shares) {
        g1_xyzz acc = G1::identity();
        for (const auto& c : shares) {
            if (c.nv != shares[0].nv) throw CozkError(COZK_ERR_INTERNAL, "combine_commitment_shares: nv mismatch");
            acc = G1::add_mixed(acc, c.g_product);
        }
        return PST13Commitment{shares[0].nv, G1::to_affine(acc)};
    }

    static g1_xyzz scalar_mul(const g1_affine& p, const fe& s_mont) {
        fe sc = Fr::from_mont(s_mont);
        g1_xyzz acc = G1::identity();
        for (int k = 7; k >= 0; k--)
            for (int b = 31; b >= 0; b--) {
                acc = G1::dbl(acc);
                if ((sc.l[k] >> b) & 1u) acc = G1::add_mixed(acc, p);
            }
        return acc;
    }

    // combine_commitments (pst13.rs:333-348): sum_i coeff_i * C_i
    static g1_affine combine_commitments(const std::vector<g1_affine>& cs, const std::vector<fe>& coeffs) {
        g1_xyzz acc = G1::identity();
        for (size_t i = 0; i < cs.size(); i++) acc = G1::add(acc, scalar_mul(cs[i], coeffs[i]));
        return G1::to_affine(acc);
    }

    // pairing-free restatement of MultilinearPC::check with the trapdoor known (verify, pst13.rs:367-385):
    // C - v g == sum_i (t_i - p_i) pi_i, point in PST (reversed) order
    static bool check_with_trapdoor(const PST13Setup& s, const g1_affine& commitment, const std::vector<fe>& point_rev, const fe& value,
                                    const std::vector<g1_affine>& proofs) {
        if ((int)proofs.size() != s.nv || (int)point_rev.size() != s.nv) return false;
        g1_xyzz lhs = G1::add_mixed(G1::neg(scalar_mul(s.g, value)), commitment);
        g1_xyzz rhs = G1::identity();
        for (int i = 0; i < s.nv; i++) rhs = G1::add(rhs, scalar_mul(proofs[i], Fr::sub(s.trapdoor[i], point_rev[i])));
        g1_affine l = G1::to_affine(lhs), r = G1::to_affine(rhs);
        return Fq::eq(l.x, r.x) && Fq::eq(l.y, r.y);
    }
};

// ================================================================= opening accumulator
struct Rep3ProverOpening {
    PolyH polynomial;  // RLC of the polynomials opened at one point
    PolyH eq_poly;     // EQ(x, opening_point) as a plain polynomial
    std::vector<fe> opening_point;
    Share claim;
    size_t num_vars;
};

struct ReducedOpeningProof {
    SumcheckProof sumcheck_proof;
    std::vector<fe> sumcheck_claims;
    std::vector<g1_affine> joint_opening_proof;
};

struct Rep3ProverOpeningAccumulator {
    std::vector<Rep3ProverOpening> openings;

    // append (opening_proof.rs:77-106)
    void append(WorkerEnv& env, const std::vector<cozk_poly*>& polynomials, const cozk_vec* eq_evals, const std::vector<fe>& opening_point,
                const std::vector<fe>& claims) {
        COZK_REQUIRE(polynomials.size() == claims.size() && !polynomials.empty(), "append: polynomials / claims mismatch");
        Writer w;
        w.vec_fr(claims);
        env.star->send_response(w.b);
        Bytes req = env.star->receive_request();
        Reader rd(req);
        fe rho = rd.fr();
        fe batched_claim = rd.fr();
        std::vector<fe> rho_powers(1, Fr::one());
        for (size_t i = 1; i < polynomials.size(); i++) rho_powers.push_back(Fr::mul(rho_powers[i - 1], rho));
        std::vector<uint64_t> cf = to_abi(rho_powers);
        cozk_poly* batched = nullptr;
        rc_check(cozk_poly_linear_combination(env.ctx, polynomials.data(), cf.data(), polynomials.size(), env.mode, env.party, &batched),
                 env.ctx, "linear_combination");
        Rep3ProverOpening op;
        op.polynomial = PolyH(batched);
        cozk_poly* eqp = nullptr;
        rc_check(cozk_poly_create(env.ctx, COZK_MODE_PLAIN, eq_evals, nullptr, &eqp), env.ctx, "poly_create(eq)");
        op.eq_poly = PolyH(eqp);
        op.opening_point = opening_point;
        op.claim = env.rep3_trivial(batched_claim);
        op.num_vars = opening_point.size();
        openings.push_back(std::move(op));
    }

    // receive_claims (opening_proof.rs:108-128), coordinator side
    static std::vector<fe> receive_claims(StarNetCoordinator& net, Transcript& tr) {
        std::vector<std::vector<fe>> parts;
        for (Bytes& b : net.receive_responses()) {
            Reader rd(b);
            parts.push_back(rd.vec_fr());
        }
        std::vector<fe> claims = combine_additive(parts);
        fe rho = tr.challenge_scalar();
        fe pw = Fr::one(), batched = Fr::zero();
        for (size_t i = 0; i < claims.size(); i++) {
            batched = Fr::add(batched, Fr::mul(pw, claims[i]));
            pw = Fr::mul(pw, rho);
        }
        Writer w;
        w.fr(rho);
        w.fr(batched);
        net.broadcast_request(w.b);
        return claims;
    }

    // compute_quadratic (opening_proof.rs:364-437) -> 3 coefficients
    std::vector<fe> compute_quadratic(WorkerEnv& env, const std::vector<fe>& coeffs, size_t remaining_rounds, const fe& previous_claim) {
        size_t k = openings.size();
        std::vector<fe> e0(k), e2(k);
        std::vector<const cozk_poly*> lp, le;
        std::vector<size_t> live;
        for (size_t i = 0; i < k; i++) {
            if (remaining_rounds <= openings[i].opening_point.size()) {
                live.push_back(i);
                lp.push_back(openings[i].polynomial.h);
                le.push_back(openings[i].eq_poly.h);
            } else {
                size_t rem = remaining_rounds - openings[i].opening_point.size() - 1;
                fe sc = Fr::mul(env.into_additive(openings[i].claim), fr_from_u64((uint64_t)1 << rem));
                e0[i] = sc;
                e2[i] = sc;
            }
        }
        if (!live.empty()) {
            std::vector<uint64_t> out(8 * live.size());
            rc_check(cozk_open_quadratic_evals(env.ctx, lp.data(), le.data(), live.size(), out.data()), env.ctx, "open_quadratic");
            for (size_t j = 0; j < live.size(); j++) {
                e0[live[j]] = fe_from_u64x4(out.data() + 8 * j);
                e2[live[j]] = fe_from_u64x4(out.data() + 8 * j + 4);
            }
        }
        fe c0 = Fr::zero(), c2 = Fr::zero();
        for (size_t i = 0; i < k; i++) {
            c0 = Fr::add(c0, Fr::mul(e0[i], coeffs[i]));
            c2 = Fr::add(c2, Fr::mul(e2[i], coeffs[i]));
        }
        fe ev[3] = {c0, Fr::sub(previous_claim, c0), c2};
        std::vector<fe> cf(3);
        unipoly_from_evals(ev, 3, cf.data());
        return cf;
    }

    // prove_batch_opening_reduction (opening_proof.rs:293-361)
    std::vector<fe> prove_batch_opening_reduction(WorkerEnv& env, const std::vector<fe>& coeffs, std::vector<fe>& claims_out) {
        size_t max_num_vars = 0;
        for (auto& o : openings) max_num_vars = std::max(max_num_vars, o.num_vars);
        if (env.party == 0) {
            Writer w;
            w.u64(max_num_vars);
            env.star->send_response(w.b);
        }
        fe e = Fr::zero();
        for (size_t i = 0; i < openings.size(); i++) {
            Share cl = openings[i].claim;
            if (openings[i].num_vars != max_num_vars) {
                fe sc = fr_from_u64((uint64_t)1 << (max_num_vars - openings[i].num_vars));
                cl.a = Fr::mul(cl.a, sc);
                cl.b = Fr::mul(cl.b, sc);
            }
            e = Fr::add(e, Fr::mul(env.into_additive(cl), coeffs[i]));
        }
        std::vector<fe> r;
        for (size_t round = 0; round < max_num_vars; round++) {
            size_t remaining = max_num_vars - round;
            std::vector<fe> uni = compute_quadratic(env, coeffs, remaining, e);
            Writer w;
            w.vec_fr(uni);
            env.star->send_response(w.b);
            Bytes req = env.star->receive_request();
            Reader rd(req);
            fe r_j = rd.fr();
            fe new_claim = rd.fr();
            r.push_back(r_j);
            e = env.additive_trivial(new_claim);
            uint64_t rr[4];
            fe_to_u64x4(r_j, rr);
            for (auto& o : openings) {
                if (remaining <= o.opening_point.size()) {
                    rc_check(cozk_poly_bind(env.ctx, o.eq_poly.h, rr, COZK_HIGH_TO_LOW), env.ctx, "bind eq");
                    rc_check(cozk_poly_bind(env.ctx, o.polynomial.h, rr, COZK_HIGH_TO_LOW), env.ctx, "bind poly");
                }
            }
        }
        claims_out.clear();
        for (auto& o : openings) {
            uint64_t a[4], b[4] = {0, 0, 0, 0};
            rc_check(cozk_poly_get_coeff(env.ctx, o.polynomial.h, 0, a, b), env.ctx, "get_coeff");
            Share s{fe_from_u64x4(a), fe_from_u64x4(b)};
            claims_out.push_back(env.into_additive(s));
        }
        return r;
    }

    // reduce_and_prove_worker (opening_proof.rs:238-291)
    void reduce_and_prove_worker(WorkerEnv& env, const PST13Setup& pcs_setup) {
        Bytes req = env.star->receive_request();
        Reader rd(req);
        fe rho = rd.fr();
        std::vector<fe> rho_powers(1, Fr::one());
        for (size_t i = 1; i < openings.size(); i++) rho_powers.push_back(Fr::mul(rho_powers[i - 1], rho));
        // "unbound_polys": binds never touch a polynomial's original coefficients on the device, so a
        // zero-copy chunk view taken before the sumcheck stands in for the reference's clones
        std::vector<PolyH> unbound;
        for (auto& o : openings) {
            cozk_poly* v = nullptr;
            rc_check(cozk_poly_chunk(env.ctx, o.polynomial.h, 0, cozk_poly_len(o.polynomial.h), &v), env.ctx, "poly_chunk");
            unbound.push_back(PolyH(v));
        }
        std::vector<fe> sumcheck_claims;
        std::vector<fe> r_sumcheck = prove_batch_opening_reduction(env, rho_powers, sumcheck_claims);
        Writer w;
        w.vec_fr(sumcheck_claims);
        env.star->send_response(w.b);
        Bytes greq = env.star->receive_request();
        Reader grd(greq);
        fe gamma = grd.fr();
        std::vector<fe> gamma_powers(1, Fr::one());
        for (size_t i = 1; i < openings.size(); i++) gamma_powers.push_back(Fr::mul(gamma_powers[i - 1], gamma));
        std::vector<const cozk_poly*> up;
        for (auto& u : unbound) up.push_back(u.h);
        std::vector<uint64_t> cf = to_abi(gamma_powers);
        cozk_poly* joint = nullptr;
        rc_check(cozk_poly_linear_combination(env.ctx, up.data(), cf.data(), up.size(), env.mode, env.party, &joint), env.ctx, "joint poly");
        PolyH jp(joint);
        PST13::prove_rep3(env, pcs_setup, jp.h, r_sumcheck);
    }

    // reduce_and_prove (opening_proof.rs:181-235), coordinator side
    static ReducedOpeningProof reduce_and_prove(StarNetCoordinator& net, Transcript& tr, std::vector<fe>& r_out, fe& rho_out, fe& gamma_out) {
        ReducedOpeningProof proof;
        fe rho = tr.challenge_scalar();
        Writer w;
        w.fr(rho);
        net.broadcast_request(w.b);
        Bytes mb = net.receive_response(0);
        Reader mr(mb);
        size_t max_num_vars = (size_t)mr.u64();
        std::vector<fe> r;
        for (size_t round = 0; round < max_num_vars; round++) {
            std::vector<std::vector<fe>> parts;
            for (Bytes& b : net.receive_responses()) {
                Reader rd(b);
                parts.push_back(rd.vec_fr());
            }
            std::vector<fe> uni = combine_additive(parts);
            std::vector<fe> comp = unipoly_compress(uni);
            tr.append_scalars(comp);
            fe r_j = tr.challenge_scalar();
            r.push_back(r_j);
            fe new_claim = unipoly_eval(uni, r_j);
            Writer ww;
            ww.fr(r_j);
            ww.fr(new_claim);
            net.broadcast_request(ww.b);
            proof.sumcheck_proof.compressed_polys.push_back(comp);
        }
        std::vector<std::vector<fe>> parts;
        for (Bytes& b : net.receive_responses()) {
            Reader rd(b);
            parts.push_back(rd.vec_fr());
        }
        proof.sumcheck_claims = combine_additive(parts);
        tr.append_scalars(proof.sumcheck_claims);
        fe gamma = tr.challenge_scalar();
        Writer gw;
        gw.fr(gamma);
        net.broadcast_request(gw.b);
        proof.joint_opening_proof = PST13::coordinate_prove(net);
        r_out = r;
        rho_out = rho;
        gamma_out = gamma;
        return proof;
    }
};

// ================================================================= generic sumcheck (prove_arbitrary_worker)
// co-jolt/src/subprotocols/sumcheck.rs:168-246 for comb_func = product of the polynomials (at most one shared
// factor; Spartan inner/shift sumchecks r1cs/spartan/worker.rs:162-235, output check read_write_memory/worker.rs:149-164).
// Binds HighToLow; returns r and the polynomials' final claims as additive shares.
struct ArbitraryResult {
    std::vector<fe> r;
    std::vector<fe> final_evals;
};
static ArbitraryResult prove_arbitrary_worker(WorkerEnv& env, const fe& claim, int num_rounds, std::vector<cozk_poly*>& polys,
                                              int combined_degree) {
    COZK_REQUIRE(combined_degree >= 1 && combined_degree <= 3 && !polys.empty() && polys.size() <= 4, "prove_arbitrary: unsupported shape");
    ArbitraryResult res;
    fe previous_claim = claim;
    bool any_shared = false;
    for (auto* p : polys) any_shared |= cozk_poly_mode(p) == COZK_MODE_REP3;
    for (int round = 0; round < num_rounds; round++) {
        std::vector<uint64_t> ev(4 * (size_t)combined_degree);
        rc_check(cozk_prod_sumcheck_evals(env.ctx, polys.data(), polys.size(), combined_degree, ev.data()), env.ctx, "prod_sumcheck_evals");
        std::vector<fe> pts((size_t)combined_degree + 1);
        for (int e = 0; e < combined_degree; e++) {
            fe v = fe_from_u64x4(ev.data() + 4 * e);
            // a product of public polynomials is a public value: additive share held by P0 only
            if (!any_shared && env.mode == COZK_MODE_REP3) v = env.additive_trivial(v);
            pts[e == 0 ? 0 : e + 1] = v;
        }
        pts[1] = Fr::sub(previous_claim, pts[0]);  // eval_points.insert(1, previous_claim - eval_points[0])
        std::vector<fe> cf((size_t)combined_degree + 1);
        if (combined_degree == 1) {
            cf[0] = pts[0];
            cf[1] = Fr::sub(pts[1], pts[0]);
        } else {
            unipoly_from_evals(pts.data(), combined_degree + 1, cf.data());
        }
        Writer w;
        w.vec_fr(cf);
        env.star->send_response(w.b);
        Bytes req = env.star->receive_request();
        Reader rd(req);
        fe r_j = rd.fr();
        fe next_claim = rd.fr();
        res.r.push_back(r_j);
        uint64_t rr[4];
        fe_to_u64x4(r_j, rr);
        for (auto* p : polys) rc_check(cozk_poly_bind(env.ctx, p, rr, COZK_HIGH_TO_LOW), env.ctx, "bind");
        previous_claim = env.additive_trivial(next_claim);
    }
    for (auto* p : polys) {
        uint64_t a[4], b[4] = {0, 0, 0, 0};
        rc_check(cozk_poly_get_coeff(env.ctx, p, 0, a, b), env.ctx, "get_coeff");
        if (cozk_poly_mode(p) == COZK_MODE_REP3) res.final_evals.push_back(env.into_additive(Share{fe_from_u64x4(a), fe_from_u64x4(b)}));
        else res.final_evals.push_back(env.mode == COZK_MODE_REP3 ? env.additive_trivial(fe_from_u64x4(a)) : fe_from_u64x4(a));
    }
    return res;
}

// ================================================================= co-noir-spartan sumcheck workers
// rep3_first_sumcheck_worker (co-noir-spartan/co-spartan/src/worker.rs:593-639): per round send the 4
// evaluations (+ additive zero-mask, co-spartan/src/sumcheck.rs:271-273), receive r, fix_variables.
// The mask stream is PRF(key_self) - PRF(key_prev) (get_mask_scalar_additive, mpc-core/src/protocols/additive.rs:44-50);
// returns the point; `finals` = (za, zb, zc share_0[0], eq[0]) as sent at the end.
static fe spartan_mask_additive(WorkerEnv& env);
static std::vector<fe> rep3_first_sumcheck_worker(WorkerEnv& env, cozk_poly* za, cozk_poly* zb, cozk_poly* zc, cozk_poly* eq, std::vector<fe>& finals) {
    size_t len = cozk_poly_len(eq);
    int num_vars = 0;
    while (((size_t)1 << num_vars) < len) num_vars++;
    std::vector<fe> point;
    for (int round = 0; round < num_vars; round++) {
        uint64_t ev[16];
        rc_check(cozk_spartan_first_round(env.ctx, za, zb, zc, eq, ev), env.ctx, "spartan_first_round");
        std::vector<fe> msg(4);
        for (int t = 0; t < 4; t++) msg[t] = Fr::add(fe_from_u64x4(ev + 4 * t), spartan_mask_additive(env));
        Writer w;
        w.vec_fr(msg);
        env.star->send_response(w.b);
        Bytes req = env.star->receive_request();
        Reader rd(req);
        fe r = rd.fr();
        point.push_back(r);
        uint64_t rr[4];
        fe_to_u64x4(r, rr);
        for (cozk_poly* p : {za, zb, zc, eq}) rc_check(cozk_poly_bind(env.ctx, p, rr, COZK_LOW_TO_HIGH), env.ctx, "fix_variables");
    }
    finals.clear();
    Writer w;
    for (cozk_poly* p : {za, zb, zc, eq}) {
        uint64_t a[4], b[4];
        rc_check(cozk_poly_get_coeff(env.ctx, p, 0, a, b), env.ctx, "get_coeff");
        finals.push_back(fe_from_u64x4(a));
        w.fr(finals.back());
    }
    env.star->send_response(w.b);
    return point;
}

// rep3_second_sumcheck_worker (co-spartan/src/sumcheck.rs:282-395 round function): 3 Rep3 evaluations per round
static std::vector<fe> rep3_second_sumcheck_worker(WorkerEnv& env, cozk_poly* z, cozk_poly* a, cozk_poly* b, cozk_poly* c, const fe coef[3],
                                                   std::vector<fe>& finals) {
    size_t len = cozk_poly_len(z);
    int num_vars = 0;
    while (((size_t)1 << num_vars) < len) num_vars++;
    uint64_t cf[12];
    for (int i = 0; i < 3; i++) fe_to_u64x4(coef[i], cf + 4 * i);
    std::vector<fe> point;
    for (int round = 0; round < num_vars; round++) {
        uint64_t ea[12], eb[12];
        rc_check(cozk_spartan_second_round(env.ctx, z, a, b, c, cf, ea, eb), env.ctx, "spartan_second_round");
        Writer w;
        w.u64(3);
        for (int t = 0; t < 3; t++) {
            // get_mask_scalar_rep3 (mpc-core/src/protocols/rep3/arithmetic.rs:39-48): a zero-sharing per component
            fe m0 = spartan_mask_additive(env), m1 = spartan_mask_additive(env);
            w.fr(Fr::add(fe_from_u64x4(ea + 4 * t), m0));
            w.fr(env.mode == COZK_MODE_REP3 ? Fr::add(fe_from_u64x4(eb + 4 * t), m1) : Fr::zero());
        }
        env.star->send_response(w.b);
        Bytes req = env.star->receive_request();
        Reader rd(req);
        fe r = rd.fr();
        point.push_back(r);
        uint64_t rr[4];
        fe_to_u64x4(r, rr);
        for (cozk_poly* p : {z, a, b, c}) rc_check(cozk_poly_bind(env.ctx, p, rr, COZK_LOW_TO_HIGH), env.ctx, "fix_variables");
    }
    finals.clear();
    Writer w;
    for (cozk_poly* p : {z, a, b, c}) {
        uint64_t x[4], y[4];
        rc_check(cozk_poly_get_coeff(env.ctx, p, 0, x, y), env.ctx, "get_coeff");
        finals.push_back(fe_from_u64x4(x));
        w.fr(finals.back());
    }
    env.star->send_response(w.b);
    return point;
}

// the tiny per-round masks are made on the host with the same keyed ChaCha12 PRF as the device masks (prf.hip.hpp)
static fe spartan_mask_additive(WorkerEnv& env) {
    if (env.mode != COZK_MODE_REP3) return Fr::zero();  // a single party has nothing to hide from itself
    fe m = Fr::sub(prf_fr(prf_key_from_bytes(env.key_self), env.mask_ctr), prf_fr(prf_key_from_bytes(env.key_prev), env.mask_ctr));
    env.mask_ctr++;
    return m;
}

}  // namespace cozk
