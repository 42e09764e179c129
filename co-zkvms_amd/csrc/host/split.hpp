// Worker sub-net ("split") form of the pipeline: every party is 2^k workers, each holding one chunk of
// every polynomial over the HIGH variables (split_poly, co-jolt/src/poly/dense_mlpoly.rs:275-301;
// co-jolt/README.md:44) and gp_batch / 2^k whole grand-product circuits.  Included by harness.hip (uses its
// PartyState / ProofBundle).  What changes against the single-worker drivers of prover.hpp:
//   * a worker cannot derive g(1) = claim - g(0) (the claim is global), so it sends raw evaluations and the
//     coordinator inserts it -- exactly what the reference does for its only split sumcheck
//     (jolt/vm/instruction_lookups/worker.rs:593-597, coordinator.rs:131-132);
//   * after the chunk-local rounds the 2^k per-worker finals are gathered and the last k rounds run on them
//     (reference: worker 0 of the party, instruction_lookups/worker.rs:317-355; here: the coordinator, who in
//     the delegated-proving model owns the witness anyway) -- the round polynomials are mathematically the
//     same, so the GKR part of the proof is bit-identical to the single-worker proof;
//   * the opening-reduction sumcheck binds the chunk-local variables first (HighToLow inside the chunk) and
//     the k worker variables last, so its point is (worker challenges || local challenges); PST13 `open`
//     folds chunk-locally against the worker's SRS slice (co-spartan `distributed_open`,
//     co-noir-spartan/co-spartan/src/worker.rs:774-809) and the coordinator finishes the last k folds.
// participant id = worker * nparties + party (global_worker_id, mpc-net/src/rep3/mod.rs:29-32).
#pragma once

namespace cozk {

static inline fe eq1(const fe& a, const fe& b) {
    fe one = Fr::one();
    fe ab = Fr::mul(a, b);
    return Fr::add(Fr::sub(Fr::sub(one, a), b), Fr::dbl(ab));
}
// eq(r[off .. off+k), bits of w) with r[off] pairing with the MOST significant of the k bits (big-endian)
static inline fe eq_index_be(const std::vector<fe>& r, size_t off, int k, uint32_t w) {
    fe acc = Fr::one(), one = Fr::one();
    for (int j = 0; j < k; j++) {
        uint32_t bit = (w >> (k - 1 - j)) & 1u;
        acc = Fr::mul(acc, bit ? r[off + j] : Fr::sub(one, r[off + j]));
    }
    return acc;
}

struct SplitEnv {
    WorkerEnv env;
    int worker = 0;
    int k = 0;  // log2(workers)
    int participant() const { return worker * (env.mode == COZK_MODE_REP3 ? 3 : 1) + env.party; }
    bool is_lead() const { return env.party == 0 && worker == 0; }  // holds "trivial" shares of public values
};

// ---------------------------------------------------------------- worker: GKR over its own circuits
static void prove_layer_split(SplitEnv& se, cozk_layer* layer, std::vector<fe>& r_gp) {
    WorkerEnv& env = se.env;
    int k = se.k;
    std::vector<fe> local(r_gp.begin() + k, r_gp.end());
    int n_loc = (int)local.size();
    fe s_w = eq_index_be(r_gp, 0, k, (uint32_t)se.worker);
    EqH eq;
    std::vector<uint64_t> w = to_abi(local);
    rc_check(cozk_spliteq_new(env.ctx, w.data(), n_loc, &eq.h), env.ctx, "spliteq_new");
    // the local rounds run behind the ABI (per-round launches, then the resident kernel for the tail of the chunk)
    struct Cb {
        WorkerEnv* env;
        fe s_w;
        std::string error;
        static int fn(void* user, int, const uint64_t ev[12], uint64_t r_out[4]) {
            Cb* c = static_cast<Cb*>(user);
            try {
                std::vector<fe> msg(3);
                for (int i = 0; i < 3; i++) msg[i] = Fr::mul(fe_from_u64x4(ev + 4 * i), c->s_w);
                Writer wr;
                wr.vec_fr(msg);
                c->env->star->send_response(wr.b);
                Bytes req = c->env->star->receive_request();
                Reader rd(req);
                fe_to_u64x4(rd.fr(), r_out);
                return 0;
            } catch (const std::exception& e) {
                c->error = e.what();
                return 1;
            }
        }
    } cb{&env, s_w, {}};
    std::vector<uint64_t> rbuf((size_t)4 * (n_loc > 0 ? n_loc : 1));
    uint64_t fc[16];
    int st = cozk_layer_prove_rounds_evals(env.ctx, layer, eq.h, n_loc, Cb::fn, &cb, rbuf.data(), fc);
    if (st != COZK_OK && !cb.error.empty()) throw CozkError(COZK_ERR_INTERNAL, cb.error);
    rc_check(st, env.ctx, "layer_prove_rounds_evals");
    std::vector<fe> rs;
    for (int j = 0; j < n_loc; j++) rs.push_back(fe_from_u64x4(rbuf.data() + 4 * j));
    Writer wf;
    for (int i = 0; i < 4; i++) wf.fr(fe_from_u64x4(fc + 4 * i));
    env.star->send_response(wf.b);
    Bytes req = env.star->receive_request();
    Reader rd(req);
    std::vector<fe> extra = rd.vec_fr();  // the k worker-bit challenges
    fe r_layer = rd.fr();
    std::vector<fe> all = rs;
    all.insert(all.end(), extra.begin(), extra.end());
    r_gp.assign(all.rbegin(), all.rend());
    r_gp.push_back(r_layer);
}

static std::vector<fe> prove_grand_product_split(SplitEnv& se, Rep3BatchedDenseGrandProduct& gp) {
    WorkerEnv& env = se.env;
    std::vector<fe> outputs = gp.claimed_outputs(env);
    Writer w;
    w.vec_fr(outputs);
    env.star->send_response(w.b);
    Bytes req = env.star->receive_request();
    Reader rd(req);
    std::vector<fe> r = rd.vec_fr();
    (void)rd.fr();
    for (size_t i = gp.layers.size(); i-- > 0;) prove_layer_split(se, gp.layers[i].h, r);
    return r;
}

// ---------------------------------------------------------------- coordinator: GKR
struct SplitTopo {
    int np, W, k;
    int participants() const { return np * W; }
};

// host rounds over the 2^k gathered (L, R) finals: interleaved-layer cubic sumcheck, LowToHigh
static void host_cubic_evals(const std::vector<fe>& L, const std::vector<fe>& R, const std::vector<fe>& E, fe g[3]) {
    g[0] = g[1] = g[2] = Fr::zero();
    for (size_t j = 0; j + 1 < L.size(); j += 2) {
        fe el = E[j], l = L[j], r = R[j];
        fe se_ = Fr::sub(E[j + 1], E[j]), sl = Fr::sub(L[j + 1], L[j]), sr = Fr::sub(R[j + 1], R[j]);
        g[0] = Fr::add(g[0], Fr::mul(Fr::mul(l, r), el));
        // X = 2
        el = Fr::add(Fr::add(el, se_), se_);
        l = Fr::add(Fr::add(l, sl), sl);
        r = Fr::add(Fr::add(r, sr), sr);
        g[1] = Fr::add(g[1], Fr::mul(Fr::mul(l, r), el));
        el = Fr::add(el, se_);
        l = Fr::add(l, sl);
        r = Fr::add(r, sr);
        g[2] = Fr::add(g[2], Fr::mul(Fr::mul(l, r), el));
    }
}
static void host_fold_pairs(std::vector<fe>& v, const fe& r) {
    size_t n = v.size() / 2;
    for (size_t i = 0; i < n; i++) v[i] = Fr::add(v[2 * i], Fr::mul(Fr::sub(v[2 * i + 1], v[2 * i]), r));
    v.resize(n);
}
static void host_fold_halves(std::vector<fe>& v, const fe& r) {
    size_t n = v.size() / 2;
    for (size_t i = 0; i < n; i++) v[i] = Fr::add(v[i], Fr::mul(Fr::sub(v[i + n], v[i]), r));
    v.resize(n);
}

static GrandProductProof coordinate_prove_grand_product_split(StarNetCoordinator& net, Transcript& tr, const SplitTopo& tp, size_t num_layers,
                                                              fe& claim_out, std::vector<fe>& r_out) {
    GrandProductProof proof;
    // outputs: per worker the sum over its parties, concatenated in worker order (= global circuit order)
    {
        std::vector<Bytes> msgs = net.receive_responses();
        for (int w = 0; w < tp.W; w++) {
            std::vector<std::vector<fe>> parts;
            for (int p = 0; p < tp.np; p++) {
                Reader rd(msgs[(size_t)w * tp.np + p]);
                parts.push_back(rd.vec_fr());
            }
            std::vector<fe> o = combine_additive(parts);
            proof.outputs.insert(proof.outputs.end(), o.begin(), o.end());
        }
    }
    tr.append_scalars(proof.outputs);
    std::vector<fe> padded = proof.outputs;
    while (padded.size() & (padded.size() - 1)) padded.push_back(Fr::zero());
    int nv = 0;
    while (((size_t)1 << nv) < padded.size()) nv++;
    std::vector<fe> r = tr.challenge_vector(nv);
    std::vector<fe> eqv = eq_evals_host(r);
    fe claim = Fr::zero();
    for (size_t i = 0; i < padded.size(); i++) claim = Fr::add(claim, Fr::mul(eqv[i], padded[i]));
    {
        Writer w;
        w.vec_fr(r);
        w.fr(claim);
        net.broadcast_request(w.b);
    }
    for (size_t layer = 0; layer < num_layers; layer++) {
        GrandProductLayerProof lp;
        int k = tp.k;
        int n_loc = (int)r.size() - k;
        std::vector<fe> c;  // round challenges in binding order
        fe e = claim;
        for (int round = 0; round < n_loc; round++) {
            fe G[3] = {Fr::zero(), Fr::zero(), Fr::zero()};
            for (Bytes& b : net.receive_responses()) {
                Reader rd(b);
                std::vector<fe> v = rd.vec_fr();
                for (int i = 0; i < 3; i++) G[i] = Fr::add(G[i], v[i]);
            }
            fe ev[4] = {G[0], Fr::sub(e, G[0]), G[1], G[2]};
            std::vector<fe> cf(4);
            unipoly_from_evals(ev, 4, cf.data());
            std::vector<fe> comp = unipoly_compress(cf);
            tr.append_scalars(comp);
            fe r_j = tr.challenge_scalar();
            c.push_back(r_j);
            e = unipoly_eval(cf, r_j);
            Writer w;
            w.fr(r_j);
            w.fr(e);
            net.broadcast_request(w.b);
            lp.proof.compressed_polys.push_back(comp);
        }
        // gathered finals -> L[w], R[w]
        std::vector<fe> L(tp.W, Fr::zero()), R(tp.W, Fr::zero());
        {
            std::vector<Bytes> msgs = net.receive_responses();
            for (int w = 0; w < tp.W; w++)
                for (int p = 0; p < tp.np; p++) {
                    Reader rd(msgs[(size_t)w * tp.np + p]);
                    fe la = rd.fr();
                    (void)rd.fr();
                    fe ra = rd.fr();
                    (void)rd.fr();
                    L[w] = Fr::add(L[w], la);
                    R[w] = Fr::add(R[w], ra);
                }
        }
        // eq over the worker bits, times the fully bound local eq
        fe e_loc = Fr::one();
        for (int i = 0; i < n_loc; i++) e_loc = Fr::mul(e_loc, eq1(r[k + i], c[n_loc - 1 - i]));
        std::vector<fe> E(tp.W);
        for (int w = 0; w < tp.W; w++) E[w] = Fr::mul(eq_index_be(r, 0, k, (uint32_t)w), e_loc);
        std::vector<fe> extra;
        for (int t = 0; t < k; t++) {
            fe g[3];
            host_cubic_evals(L, R, E, g);
            fe ev[4] = {g[0], Fr::sub(e, g[0]), g[1], g[2]};
            std::vector<fe> cf(4);
            unipoly_from_evals(ev, 4, cf.data());
            std::vector<fe> comp = unipoly_compress(cf);
            tr.append_scalars(comp);
            fe r_t = tr.challenge_scalar();
            extra.push_back(r_t);
            e = unipoly_eval(cf, r_t);
            host_fold_pairs(L, r_t);
            host_fold_pairs(R, r_t);
            host_fold_pairs(E, r_t);
            lp.proof.compressed_polys.push_back(comp);
        }
        lp.left_claim = L[0];
        lp.right_claim = R[0];
        tr.append_scalar(lp.left_claim);
        tr.append_scalar(lp.right_claim);
        std::vector<fe> all = c;
        all.insert(all.end(), extra.begin(), extra.end());
        r.assign(all.rbegin(), all.rend());
        fe r_layer = tr.challenge_scalar();
        Writer w;
        w.vec_fr(extra);
        w.fr(r_layer);
        net.broadcast_request(w.b);
        claim = Fr::add(lp.left_claim, Fr::mul(r_layer, Fr::sub(lp.right_claim, lp.left_claim)));
        r.push_back(r_layer);
        proof.gkr_layers.push_back(lp);
    }
    claim_out = claim;
    r_out = r;
    return proof;
}

// ---------------------------------------------------------------- worker: openings on chunks
struct SplitOpening {
    PolyH polynomial;  // RLC of the chunk polynomials
    PolyH eq_poly;     // this worker's chunk of EQ(x, point): eq(point[:k], w) * evals(point[k:])
    PolyH unbound;     // zero-copy view of `polynomial` taken before binding
};

static VecH chunk_chi(SplitEnv& se, const std::vector<fe>& point) {
    WorkerEnv& env = se.env;
    std::vector<fe> local(point.begin() + se.k, point.end());
    std::vector<uint64_t> rr = to_abi(local);
    cozk_vec* chi = nullptr;
    rc_check(cozk_eq_evals(env.ctx, rr.data(), (int)local.size(), &chi), env.ctx, "eq_evals");
    VecH h(chi);
    if (se.k > 0) {
        uint64_t s[4];
        fe_to_u64x4(eq_index_be(point, 0, se.k, (uint32_t)se.worker), s);
        rc_check(cozk_vec_scale(env.ctx, chi, s), env.ctx, "vec_scale");
    }
    return h;
}

// batch_evaluate + append on the worker's chunks (opening_proof.rs:77-106 with chunked polynomials)
static void split_open_group(SplitEnv& se, std::vector<SplitOpening>& acc, const std::vector<cozk_poly*>& polys, const std::vector<fe>& point) {
    WorkerEnv& env = se.env;
    VecH chi = chunk_chi(se, point);
    std::vector<fe> claims(polys.size());
    std::vector<const cozk_poly*> shp, pbp;
    std::vector<size_t> shi, pbi;
    for (size_t i = 0; i < polys.size(); i++) {
        if (cozk_poly_mode(polys[i]) == env.mode) {
            shp.push_back(polys[i]);
            shi.push_back(i);
        } else {
            pbp.push_back(polys[i]);
            pbi.push_back(i);
        }
    }
    if (!shp.empty()) {
        std::vector<uint64_t> out(4 * shp.size());
        rc_check(cozk_poly_batch_evaluate_at_chi(env.ctx, shp.data(), shp.size(), chi.h, out.data()), env.ctx, "batch_evaluate");
        for (size_t j = 0; j < shp.size(); j++) claims[shi[j]] = fe_from_u64x4(out.data() + 4 * j);
    }
    if (!pbp.empty()) {
        std::vector<uint64_t> out(4 * pbp.size());
        rc_check(cozk_poly_batch_evaluate_at_chi(env.ctx, pbp.data(), pbp.size(), chi.h, out.data()), env.ctx, "batch_evaluate(public)");
        // a public chunk evaluation is contributed by party 0's worker only
        for (size_t j = 0; j < pbp.size(); j++) claims[pbi[j]] = env.party == 0 ? fe_from_u64x4(out.data() + 4 * j) : Fr::zero();
    }
    Writer w;
    w.vec_fr(claims);
    env.star->send_response(w.b);
    Bytes req = env.star->receive_request();
    Reader rd(req);
    fe rho = rd.fr();
    (void)rd.fr();  // batched claim: tracked by the coordinator in the split form
    std::vector<fe> pw(1, Fr::one());
    for (size_t i = 1; i < polys.size(); i++) pw.push_back(Fr::mul(pw[i - 1], rho));
    std::vector<uint64_t> cf = to_abi(pw);
    cozk_poly* batched = nullptr;
    rc_check(cozk_poly_linear_combination(env.ctx, polys.data(), cf.data(), polys.size(), env.mode, env.party, &batched), env.ctx, "linear_combination");
    SplitOpening op;
    op.polynomial = PolyH(batched);
    cozk_poly* eqp = nullptr;
    rc_check(cozk_poly_create(env.ctx, COZK_MODE_PLAIN, chi.h, nullptr, &eqp), env.ctx, "poly_create(eq)");
    op.eq_poly = PolyH(eqp);
    cozk_poly* view = nullptr;
    rc_check(cozk_poly_chunk(env.ctx, batched, 0, cozk_poly_len(batched), &view), env.ctx, "poly_chunk");
    op.unbound = PolyH(view);
    acc.push_back(std::move(op));
}

// reduce_and_prove_worker on chunks: local rounds send (eval_0, eval_2) sums; finals; gamma; chunk-local PST open
static void split_reduce_and_prove_worker(SplitEnv& se, std::vector<SplitOpening>& acc, const PST13Setup& setup) {
    WorkerEnv& env = se.env;
    Bytes req = env.star->receive_request();
    Reader rd0(req);
    fe rho = rd0.fr();
    std::vector<fe> coeffs(1, Fr::one());
    for (size_t i = 1; i < acc.size(); i++) coeffs.push_back(Fr::mul(coeffs[i - 1], rho));
    size_t len = cozk_poly_len(acc[0].polynomial.h);
    int n_loc = 0;
    while (((size_t)1 << n_loc) < len) n_loc++;
    std::vector<const cozk_poly*> lp, le;
    for (auto& o : acc) {
        lp.push_back(o.polynomial.h);
        le.push_back(o.eq_poly.h);
    }
    std::vector<fe> c;
    for (int round = 0; round < n_loc; round++) {
        std::vector<uint64_t> out(8 * acc.size());
        rc_check(cozk_open_quadratic_evals(env.ctx, lp.data(), le.data(), acc.size(), out.data()), env.ctx, "open_quadratic");
        fe c0 = Fr::zero(), c2 = Fr::zero();
        for (size_t i = 0; i < acc.size(); i++) {
            c0 = Fr::add(c0, Fr::mul(fe_from_u64x4(out.data() + 8 * i), coeffs[i]));
            c2 = Fr::add(c2, Fr::mul(fe_from_u64x4(out.data() + 8 * i + 4), coeffs[i]));
        }
        Writer w;
        w.vec_fr({c0, c2});
        env.star->send_response(w.b);
        Bytes rq = env.star->receive_request();
        Reader rd(rq);
        fe r_j = rd.fr();
        c.push_back(r_j);
        uint64_t rr[4];
        fe_to_u64x4(r_j, rr);
        for (auto& o : acc) {
            rc_check(cozk_poly_bind(env.ctx, o.eq_poly.h, rr, COZK_HIGH_TO_LOW), env.ctx, "bind eq");
            rc_check(cozk_poly_bind(env.ctx, o.polynomial.h, rr, COZK_HIGH_TO_LOW), env.ctx, "bind poly");
        }
    }
    {
        // finals: per opening the additive share of poly(w, c) and the public eq(w, c)
        Writer w;
        for (auto& o : acc) {
            uint64_t a[4], b[4] = {0, 0, 0, 0}, q[4], qb[4];
            rc_check(cozk_poly_get_coeff(env.ctx, o.polynomial.h, 0, a, b), env.ctx, "get_coeff");
            rc_check(cozk_poly_get_coeff(env.ctx, o.eq_poly.h, 0, q, qb), env.ctx, "get_coeff(eq)");
            w.fr(env.into_additive(Share{fe_from_u64x4(a), fe_from_u64x4(b)}));
            w.fr(fe_from_u64x4(q));
        }
        env.star->send_response(w.b);
    }
    Bytes greq = env.star->receive_request();
    Reader grd(greq);
    (void)grd.vec_fr();  // the k worker-bit challenges (used by the coordinator's folds only)
    fe gamma = grd.fr();
    std::vector<fe> gp(1, Fr::one());
    for (size_t i = 1; i < acc.size(); i++) gp.push_back(Fr::mul(gp[i - 1], gamma));
    std::vector<const cozk_poly*> up;
    for (auto& o : acc) up.push_back(o.unbound.h);
    std::vector<uint64_t> cf = to_abi(gp);
    cozk_poly* joint = nullptr;
    rc_check(cozk_poly_linear_combination(env.ctx, up.data(), cf.data(), up.size(), env.mode, env.party, &joint), env.ctx, "joint poly");
    PolyH jp(joint);
    cozk_vec* av = nullptr;
    rc_check(cozk_poly_share_view(env.ctx, jp.h, 0, &av), env.ctx, "share_view");
    VecH a(av);
    // chunk-local PST folds use the local challenges in reverse (little-endian variable order)
    std::vector<fe> rev(c.rbegin(), c.rend());
    fe final_value;
    std::vector<g1_affine> pf = PST13::open(env.ctx, setup, a.h, rev, &final_value);
    Writer w;
    w.vec_g1(pf);
    w.fr(final_value);
    env.star->send_response(w.b);
}

}  // namespace cozk
