// Split (worker sub-net) form of the harness pipeline; included by harness.hip after the single-worker
// worker_main / coordinator_main.  See host/split.hpp for the protocol.
#pragma once
#include "split.hpp"

namespace {

// SplitMix stream chunk: element i of the chunk is element (offset + i) of stream `seed`
static inline uint64_t chunk_seed(uint64_t seed, uint64_t offset) { return seed + offset * 0xD1342543DE82EF95ull; }

static fe eq_index_le(const std::vector<fe>& t, size_t off, int k, uint32_t w) {
    // little-endian: bit j of w pairs with t[off + j]
    fe acc = Fr::one(), one = Fr::one();
    for (int j = 0; j < k; j++) acc = Fr::mul(acc, ((w >> j) & 1u) ? t[off + j] : Fr::sub(one, t[off + j]));
    return acc;
}

static std::vector<fe> harness_trapdoor(const cozk_harness_config& c) {
    std::vector<fe> t(c.log_n);
    for (int i = 0; i < c.log_n; i++) t[i] = synthetic_fr_host(c.seed ^ 0x7A7A7A7Aull, (uint64_t)i);
    return t;
}

// one participant (party p, worker w): chunk w of every polynomial, circuits [w B', (w+1) B') of the grand product
static void setup_participant_split(cozk_harness* h, PartyState& ps, int worker) {
    const cozk_harness_config& c = h->cfg;
    cozk_ctx* ctx = ps.ctx;
    int k = c.log_workers;
    int W = 1 << k;
    size_t n_chunk = h->N >> k;
    int n_loc = c.log_n - k;
    COZK_REQUIRE(n_loc >= 1, "split: more workers than polynomial entries");
    COZK_REQUIRE(c.n_small == 0, "split: n_small must be 0 (shorter polynomials live in worker 0's range only)");
    COZK_REQUIRE(c.gp_batch % W == 0, "split: gp_batch must be a multiple of the worker count");
    std::vector<fe> t = harness_trapdoor(c);
    std::vector<fe> t_loc(t.begin(), t.begin() + n_loc);
    // generator of this worker's SRS slice: eq_le(t[n_loc..], w) * g
    g1_affine g;
    g.x = Fq::one();
    g.y = Fq::from_u64(2);
    g1_affine gw = G1::to_affine(PST13::scalar_mul(g, eq_index_le(t, (size_t)n_loc, k, (uint32_t)worker)));
    ps.setup = PST13::setup(ctx, t_loc, c.precompute, &gw);
    uint64_t off = (uint64_t)worker * n_chunk;
    int j = 0;
    for (int q = 0; q < c.n_fr; q++, j++) {
        VecH a, b;
        make_share_vectors(ctx, n_chunk, chunk_seed(c.seed + 1000ull * (uint64_t)(j + 1), off), ps.party, c.mode, a, b);
        cozk_poly* p = nullptr;
        rc_check(cozk_poly_create(ctx, c.mode, a.h, b.h, &p), ctx, "poly_create");
        ps.polys.push_back(PolyH(p));
        cozk_vec* view = nullptr;
        rc_check(cozk_poly_share_view(ctx, p, 0, &view), ctx, "share_view");
        ps.commit_vecs.push_back(VecH(view));
        ps.is_public.push_back(0);
    }
    struct SmallSpec { int count, kind, bits; };
    SmallSpec specs[3] = {{c.n_u16, COZK_SCALAR_U16, 0}, {c.n_u32, COZK_SCALAR_U32, 0}, {c.n_flags, COZK_SCALAR_U8, 1}};
    for (auto& sp : specs) {
        for (int q = 0; q < sp.count; q++, j++) {
            VecH sv = make_vec_random(ctx, n_chunk, sp.kind, chunk_seed(c.seed + 1000ull * (uint64_t)(j + 1), off), sp.bits);
            cozk_vec* frv = nullptr;
            rc_check(cozk_vec_alloc(ctx, n_chunk, COZK_SCALAR_FR, &frv), ctx, "vec_alloc");
            VecH fr(frv);
            unsigned grid = (unsigned)((n_chunk + 255) / 256);
            if (sp.kind == COZK_SCALAR_U8) k_small_to_fr_u8<<<grid, 256, 0, ctx->stream>>>((const uint8_t*)cozk_vec_device_ptr(sv.h), (fe*)cozk_vec_device_ptr(fr.h), n_chunk);
            else if (sp.kind == COZK_SCALAR_U16) k_small_to_fr_u16<<<grid, 256, 0, ctx->stream>>>((const uint16_t*)cozk_vec_device_ptr(sv.h), (fe*)cozk_vec_device_ptr(fr.h), n_chunk);
            else k_small_to_fr_u32<<<grid, 256, 0, ctx->stream>>>((const uint32_t*)cozk_vec_device_ptr(sv.h), (fe*)cozk_vec_device_ptr(fr.h), n_chunk);
            HIP_TRY(hipGetLastError());
            cozk_poly* p = nullptr;
            rc_check(cozk_poly_create(ctx, COZK_MODE_PLAIN, fr.h, nullptr, &p), ctx, "poly_create");
            ps.polys.push_back(PolyH(p));
            ps.commit_vecs.push_back(std::move(sv));
            ps.is_public.push_back(1);
        }
    }
    size_t leaves_per_circuit = (size_t)1 << c.gp_log_leaves;
    size_t bprime = (size_t)c.gp_batch / W;
    size_t nleaves = bprime * leaves_per_circuit;
    VecH la, lb;
    make_share_vectors(ctx, nleaves, chunk_seed(c.seed + 500000ull, (uint64_t)worker * nleaves), ps.party, c.mode, la, lb);
    cozk_layer* lv = nullptr;
    rc_check(cozk_layer_create(ctx, c.mode, la.h, lb.h, 1, &lv), ctx, "layer_create");
    ps.leaves = LayerH(lv);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
}

static void worker_main_split(cozk_harness* h, PartyState& ps, int worker, StarNetWorker* star, RingNet* ring) {
    const cozk_harness_config& c = h->cfg;
    SplitEnv se;
    se.env.ctx = ps.ctx;
    se.env.mode = c.mode;
    se.env.party = ps.party;
    se.env.star = star;
    se.env.ring = ring;
    harness_prf_key(c.seed, (uint64_t)ps.party + 16ull * (uint64_t)worker, se.env.key_self);
    harness_prf_key(c.seed, (uint64_t)((ps.party + 2) % 3) + 16ull * (uint64_t)worker, se.env.key_prev);
    se.worker = worker;
    se.k = c.log_workers;
    WorkerEnv& env = se.env;
    HIP_TRY(hipSetDevice(ps.ctx->device));
    double t_start = now_ms();
    {
        std::vector<cozk_vec*> vs;
        for (auto& v : ps.commit_vecs) vs.push_back(v.h);
        std::vector<PST13Commitment> cm = PST13::batch_commit(ps.ctx, *ps.setup, vs);
        Writer w;
        put_commitments(w, cm, ps.is_public, ps.party);
        star->send_response(w.b);
    }
    double t1 = now_ms();
    ps.t_commit = t1 - t_start;
    cozk_layer* lv = nullptr;
    rc_check(cozk_layer_clone(ps.ctx, ps.leaves.h, &lv), ps.ctx, "layer_clone");
    size_t W = (size_t)1 << se.k;
    Rep3BatchedDenseGrandProduct gp = Rep3BatchedDenseGrandProduct::construct(env, LayerH(lv), (size_t)c.gp_batch / W);
    HIP_TRY(hipStreamSynchronize(ps.ctx->stream));
    double t2 = now_ms();
    ps.t_construct = t2 - t1;
    std::vector<fe> r_gp = prove_grand_product_split(se, gp);
    double t3 = now_ms();
    ps.t_gp = t3 - t2;
    {
        // harness check message: this participant's additive share of leaves(r_gp) restricted to its circuits
        VecH chi = chunk_chi(se, r_gp);
        cozk_poly* lp = nullptr;
        rc_check(cozk_layer_as_poly(ps.ctx, ps.leaves.h, &lp), ps.ctx, "layer_as_poly");
        PolyH lph(lp);
        uint64_t ev[4];
        const cozk_poly* arr[1] = {lph.h};
        rc_check(cozk_poly_batch_evaluate_at_chi(ps.ctx, arr, 1, chi.h, ev), ps.ctx, "leaf evaluation");
        Writer w;
        w.fr(fe_from_u64x4(ev));
        star->send_response(w.b);
    }
    std::vector<SplitOpening> acc;
    size_t K = ps.polys.size();
    size_t half = (K + 1) / 2;
    int nv = c.log_n;
    {
        std::vector<fe> p1(r_gp.end() - nv, r_gp.end()), p2(r_gp.begin(), r_gp.begin() + nv);
        std::vector<cozk_poly*> g1v, g2v;
        for (size_t i = 0; i < half; i++) g1v.push_back(ps.polys[i].h);
        for (size_t i = half; i < K; i++) g2v.push_back(ps.polys[i].h);
        split_open_group(se, acc, g1v, p1);
        if (!g2v.empty()) split_open_group(se, acc, g2v, p2);
    }
    double t4 = now_ms();
    ps.t_eval = t4 - t3;
    split_reduce_and_prove_worker(se, acc, *ps.setup);
    double t5 = now_ms();
    ps.t_open = t5 - t4;
    ps.t_total = t5 - t_start;
    ps.star_up = star->bytes_up;
    ps.star_down = star->bytes_down;
    ps.star_msgs = star->n_msgs;
    ps.ring_bytes = ring ? ring->bytes_sent : 0;
}

// commitments: entry i of every participant; shared (tag 2) entries and P0's public (tag 1) entries each sum over
// the workers (a chunk commitment against an SRS slice is a partial commitment: co-spartan/src/utils.rs:38-83)
static std::vector<PST13Commitment> combine_commitments_split(std::vector<Reader>& rds, uint64_t full_nv) {
    std::vector<PST13Commitment> out;
    uint64_t n = 0;
    for (size_t p = 0; p < rds.size(); p++) {
        uint64_t m = rds[p].u64();
        if (p == 0) n = m;
        if (m != n) throw CozkError(COZK_ERR_INTERNAL, "commitment count mismatch");
    }
    for (uint64_t i = 0; i < n; i++) {
        g1_xyzz pub = G1::identity(), sh = G1::identity();
        bool have_pub = false;
        for (size_t p = 0; p < rds.size(); p++) {
            rds[p].need(1);
            uint8_t tag = *rds[p].p++;
            if (tag == 0) continue;
            (void)rds[p].u64();
            g1_affine g = rds[p].g1();
            if (tag == 1) {
                have_pub = true;
                pub = G1::add_mixed(pub, g);
            } else {
                sh = G1::add_mixed(sh, g);
            }
        }
        out.push_back(PST13Commitment{full_nv, G1::to_affine(have_pub ? pub : sh)});
    }
    return out;
}

static int coordinator_main_split(cozk_harness* h, StarNetCoordinator& net, ProofBundle& proof, bool verify, std::string& why) {
    const cozk_harness_config& c = h->cfg;
    SplitTopo tp{h->nparties, 1 << c.log_workers, c.log_workers};
    int nv = c.log_n, k = tp.k, n_loc = nv - k;
    Transcript tr("cozk-harness");
    {
        std::vector<Bytes> msgs = net.receive_responses();
        std::vector<Reader> rds;
        for (auto& m : msgs) rds.emplace_back(m);
        proof.commitments = combine_commitments_split(rds, (uint64_t)nv);
        for (auto& cm : proof.commitments) tr.append_point(cm.g_product);
    }
    fe gp_claim;
    std::vector<fe> r_gp;
    proof.gp = coordinate_prove_grand_product_split(net, tr, tp, (size_t)c.gp_log_leaves, gp_claim, r_gp);
    fe leaf_eval = Fr::zero();
    for (Bytes& b : net.receive_responses()) {
        Reader rd(b);
        leaf_eval = Fr::add(leaf_eval, rd.fr());
    }
    size_t K = proof.commitments.size();
    size_t half = (K + 1) / 2;
    int nappend = half < K ? 2 : 1;
    std::vector<fe> batched_claims;
    for (int a = 0; a < nappend; a++) {
        std::vector<std::vector<fe>> parts;
        for (Bytes& b : net.receive_responses()) {
            Reader rd(b);
            parts.push_back(rd.vec_fr());
        }
        std::vector<fe> claims = combine_additive(parts);
        fe rho = tr.challenge_scalar();
        fe pw = Fr::one(), batched = Fr::zero();
        for (auto& cl : claims) {
            batched = Fr::add(batched, Fr::mul(pw, cl));
            pw = Fr::mul(pw, rho);
        }
        Writer w;
        w.fr(rho);
        w.fr(batched);
        net.broadcast_request(w.b);
        proof.opening_claims.push_back(claims);
        batched_claims.push_back(batched);
    }
    // ---- reduction sumcheck: local rounds on the workers, last k rounds here
    fe rho2 = tr.challenge_scalar();
    {
        Writer w;
        w.fr(rho2);
        net.broadcast_request(w.b);
    }
    std::vector<fe> coeffs(1, Fr::one());
    for (int i = 1; i < nappend; i++) coeffs.push_back(Fr::mul(coeffs[i - 1], rho2));
    fe e = Fr::zero();
    for (int i = 0; i < nappend; i++) e = Fr::add(e, Fr::mul(coeffs[i], batched_claims[i]));
    std::vector<fe> cl;  // local challenges
    for (int round = 0; round < n_loc; round++) {
        fe C0 = Fr::zero(), C2 = Fr::zero();
        for (Bytes& b : net.receive_responses()) {
            Reader rd(b);
            std::vector<fe> v = rd.vec_fr();
            C0 = Fr::add(C0, v[0]);
            C2 = Fr::add(C2, v[1]);
        }
        fe ev[3] = {C0, Fr::sub(e, C0), C2};
        std::vector<fe> cf(3);
        unipoly_from_evals(ev, 3, cf.data());
        std::vector<fe> comp = unipoly_compress(cf);
        tr.append_scalars(comp);
        fe r_j = tr.challenge_scalar();
        cl.push_back(r_j);
        e = unipoly_eval(cf, r_j);
        Writer w;
        w.fr(r_j);
        w.fr(e);
        net.broadcast_request(w.b);
        proof.reduced.sumcheck_proof.compressed_polys.push_back(comp);
    }
    std::vector<std::vector<fe>> Pv(nappend, std::vector<fe>(tp.W, Fr::zero())), Qv(nappend, std::vector<fe>(tp.W, Fr::zero()));
    {
        std::vector<Bytes> msgs = net.receive_responses();
        for (int w = 0; w < tp.W; w++)
            for (int p = 0; p < tp.np; p++) {
                Reader rd(msgs[(size_t)w * tp.np + p]);
                for (int o = 0; o < nappend; o++) {
                    fe pf = rd.fr();
                    fe qf = rd.fr();
                    Pv[o][w] = Fr::add(Pv[o][w], pf);
                    if (p == 0) Qv[o][w] = qf;
                }
            }
    }
    std::vector<fe> extra;
    for (int t = 0; t < k; t++) {
        fe C0 = Fr::zero(), C2 = Fr::zero();
        for (int o = 0; o < nappend; o++) {
            size_t hf = Pv[o].size() / 2;
            fe e0 = Fr::zero(), e2 = Fr::zero();
            for (size_t i = 0; i < hf; i++) {
                e0 = Fr::add(e0, Fr::mul(Pv[o][i], Qv[o][i]));
                fe pb = Fr::sub(Fr::dbl(Pv[o][i + hf]), Pv[o][i]), qb = Fr::sub(Fr::dbl(Qv[o][i + hf]), Qv[o][i]);
                e2 = Fr::add(e2, Fr::mul(pb, qb));
            }
            C0 = Fr::add(C0, Fr::mul(coeffs[o], e0));
            C2 = Fr::add(C2, Fr::mul(coeffs[o], e2));
        }
        fe ev[3] = {C0, Fr::sub(e, C0), C2};
        std::vector<fe> cf(3);
        unipoly_from_evals(ev, 3, cf.data());
        std::vector<fe> comp = unipoly_compress(cf);
        tr.append_scalars(comp);
        fe r_t = tr.challenge_scalar();
        extra.push_back(r_t);
        e = unipoly_eval(cf, r_t);
        for (int o = 0; o < nappend; o++) {
            host_fold_halves(Pv[o], r_t);
            host_fold_halves(Qv[o], r_t);
        }
        proof.reduced.sumcheck_proof.compressed_polys.push_back(comp);
    }
    for (int o = 0; o < nappend; o++) proof.reduced.sumcheck_claims.push_back(Pv[o][0]);
    tr.append_scalars(proof.reduced.sumcheck_claims);
    fe gamma = tr.challenge_scalar();
    {
        Writer w;
        w.vec_fr(extra);
        w.fr(gamma);
        net.broadcast_request(w.b);
    }
    // effective (big-endian) opening point: worker challenges || local challenges
    std::vector<fe> point_be = extra;
    point_be.insert(point_be.end(), cl.begin(), cl.end());
    // ---- PST13: sum the chunk-local proofs, then the last k folds over the gathered finals
    std::vector<fe> t = harness_trapdoor(c);
    std::vector<g1_xyzz> acc((size_t)n_loc, G1::identity());
    std::vector<fe> v(tp.W, Fr::zero());
    {
        std::vector<Bytes> msgs = net.receive_responses();
        for (int w = 0; w < tp.W; w++)
            for (int p = 0; p < tp.np; p++) {
                Reader rd(msgs[(size_t)w * tp.np + p]);
                std::vector<g1_affine> pf = rd.vec_g1();
                if ((int)pf.size() != n_loc) throw CozkError(COZK_ERR_INTERNAL, "split open: proof length mismatch");
                for (int i = 0; i < n_loc; i++) acc[i] = G1::add_mixed(acc[i], pf[i]);
                v[w] = Fr::add(v[w], rd.fr());
            }
    }
    for (int i = 0; i < n_loc; i++) proof.reduced.joint_opening_proof.push_back(G1::to_affine(acc[i]));
    g1_affine g;
    g.x = Fq::one();
    g.y = Fq::from_u64(2);
    for (int tt = 0; tt < k; tt++) {
        int level = n_loc + tt;                 // global PST level; table over t[level..nv)
        int m = nv - level;                     // variables left
        const fe& coord = extra[k - 1 - tt];    // point_rev[level] = point_be[nv - 1 - level]
        size_t hf = (size_t)1 << (m - 1);
        g1_xyzz pi = G1::identity();
        std::vector<fe> nvv(hf);
        for (size_t b = 0; b < hf; b++) {
            fe q = Fr::sub(v[2 * b + 1], v[2 * b]);
            nvv[b] = Fr::add(v[2 * b], Fr::mul(q, coord));
            // G_level[2b] + G_level[2b+1] = g^{eq_le(t[level+1..], b)} (the eq over the lowest variable sums to 1)
            fe sc = eq_index_le(t, (size_t)level + 1, m - 1, (uint32_t)b);
            pi = G1::add(pi, PST13::scalar_mul(g, Fr::mul(q, sc)));
        }
        v.swap(nvv);
        proof.reduced.joint_opening_proof.push_back(G1::to_affine(pi));
    }
    if (!verify) return -1;

    // ------------------------------------------------ plain verifier
    Transcript vt("cozk-harness");
    for (auto& cm : proof.commitments) vt.append_point(cm.g_product);
    fe v_claim;
    std::vector<fe> v_r;
    if (!verify_grand_product(proof.gp, vt, v_claim, v_r)) {
        why = "GKR proof rejected";
        return 0;
    }
    if (!Fr::eq(v_claim, gp_claim) || !Fr::eq(v_claim, leaf_eval)) {
        why = "final GKR claim != direct evaluation of the leaves";
        return 0;
    }
    std::vector<std::vector<fe>> points;
    points.push_back(std::vector<fe>(v_r.end() - nv, v_r.end()));
    if (nappend == 2) points.push_back(std::vector<fe>(v_r.begin(), v_r.begin() + nv));
    std::vector<fe> vb;
    std::vector<g1_affine> bc;
    for (int a = 0; a < nappend; a++) {
        fe rho = vt.challenge_scalar();
        std::vector<fe> pw(1, Fr::one());
        const auto& claims = proof.opening_claims[a];
        for (size_t i = 1; i < claims.size(); i++) pw.push_back(Fr::mul(pw[i - 1], rho));
        fe b = Fr::zero();
        for (size_t i = 0; i < pw.size(); i++) b = Fr::add(b, Fr::mul(pw[i], claims[i]));
        vb.push_back(b);
        std::vector<g1_affine> cs;
        size_t lo = a == 0 ? 0 : half, hi = a == 0 ? half : K;
        for (size_t i = lo; i < hi; i++) cs.push_back(proof.commitments[i].g_product);
        if (cs.size() != claims.size()) {
            why = "claims / commitments mismatch";
            return 0;
        }
        bc.push_back(PST13::combine_commitments(cs, pw));
    }
    fe vrho2 = vt.challenge_scalar();
    std::vector<fe> vco(1, Fr::one());
    for (int i = 1; i < nappend; i++) vco.push_back(Fr::mul(vco[i - 1], vrho2));
    fe ve = Fr::zero();
    for (int i = 0; i < nappend; i++) ve = Fr::add(ve, Fr::mul(vco[i], vb[i]));
    std::vector<fe> rs;
    if ((int)proof.reduced.sumcheck_proof.compressed_polys.size() != nv) {
        why = "reduction sumcheck: wrong number of rounds";
        return 0;
    }
    for (auto& comp : proof.reduced.sumcheck_proof.compressed_polys) {
        std::vector<fe> poly = unipoly_decompress(comp, ve);
        vt.append_scalars(comp);
        fe r_j = vt.challenge_scalar();
        rs.push_back(r_j);
        ve = unipoly_eval(poly, r_j);
    }
    // variable order of the split sumcheck: rounds 0..n_loc-1 bind the local bits (HighToLow), then the worker bits
    std::vector<fe> vpoint(rs.begin() + n_loc, rs.end());
    vpoint.insert(vpoint.end(), rs.begin(), rs.begin() + n_loc);
    fe expect = Fr::zero();
    for (int i = 0; i < nappend; i++) expect = Fr::add(expect, Fr::mul(vco[i], Fr::mul(eq_eval(points[i], vpoint), proof.reduced.sumcheck_claims[i])));
    if (!Fr::eq(expect, ve)) {
        why = "reduction sumcheck: final check failed";
        return 0;
    }
    vt.append_scalars(proof.reduced.sumcheck_claims);
    fe vgamma = vt.challenge_scalar();
    std::vector<fe> gpw(1, Fr::one());
    for (int i = 1; i < nappend; i++) gpw.push_back(Fr::mul(gpw[i - 1], vgamma));
    g1_affine joint_c = PST13::combine_commitments(bc, gpw);
    fe joint_claim = Fr::zero();
    for (int i = 0; i < nappend; i++) joint_claim = Fr::add(joint_claim, Fr::mul(gpw[i], proof.reduced.sumcheck_claims[i]));
    // pairing-free PST check with the FULL trapdoor
    PST13Setup full;
    full.nv = nv;
    full.trapdoor = t;
    full.g = g;
    std::vector<fe> rev(vpoint.rbegin(), vpoint.rend());
    if (!PST13::check_with_trapdoor(full, joint_c, rev, joint_claim, proof.reduced.joint_opening_proof)) {
        why = "PST13 opening check failed";
        return 0;
    }
    return 1;
}

}  // namespace
