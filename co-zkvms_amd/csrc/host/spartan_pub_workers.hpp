// The public lookup round of co-noir-spartan over 2^k public workers (cfg.log_pub_workers = k > 0); included by
// spartan_harness.hpp.  What the reference deals and merges:
//   setup.rs split_ipk / split_ck   worker j holds chunk j (the high k variables = j) of rows, cols, val_a/b/c and the
//                                   multiplicities, and the slice of ck_index over the low qv - k variables
//   worker.rs:296-345, 398-575      every public worker builds the FULL eq_tilde_rx / eq_tilde_ry (the lookup tables index
//                                   them) and proves on its chunk of everything else
//   coordinator.rs:425-475          val_a, val_b, val_c and the commitments are sums over the workers
//   coordinator.rs:748-811          distributed_sumcheck_coordinator: the workers' messages of the first qv - k rounds are
//                                   summed; the last k rounds run at the coordinator on the workers' final states
//   coordinator.rs rep3_poly_commit / batch open with Some(log_num_pub_workers): chunk-local opening proofs are summed and the
//                                   last k quotients are committed from the gathered finals
// The schedule on the public star (K = 2^k workers; the Rep3 parties take no part):
//   req   rx, ry, (v_0, v_1, v_2)
//   resp  partial val_a, val_b, val_c, C(eq_tilde_rx chunk), C(eq_tilde_ry chunk)
//   req   v, x_r, x_c                  resp  C(h_0), C(h_1) of both lookups on the chunk
//   req   z_r, lambda_r, z_c, lambda_c
//   (qv - k) x  resp 4 evaluations / req r;   resp the 15 final values
//   req   eta;  resp the chunk-local opening (qv - k points), its final value, 15 partial evaluations
// Every sum the coordinator forms equals the one-worker value, so the proof bytes do not depend on k (tests compare them).
#pragma once

namespace {

// non-owning window [off, off + n) of a resident device vector
static VecH vec_window(const VecH& v, size_t off, size_t n) {
    COZK_REQUIRE(v.h && off + n <= v.h->n, "vec_window: out of range");
    cozk_vec* w = new cozk_vec(*v.h);
    size_t sz = scalar_kind_bytes(v.h->kind);
    w->d = (char*)v.h->d + off * sz;
    w->n = n;
    w->bytes = n * sz;
    w->owned = false;
    return VecH(w);
}

static void spartan_setup_pub_worker(cozk_spartan* h, SpartanPubWorker& pw) {
    const cozk_spartan_config& c = h->cfg;
    const int qv = h->qv, k = c.log_pub_workers, ql = qv - k;
    const size_t Cn = (size_t)1 << ql, off = (size_t)pw.id * Cn, real = h->h_col.size();
    cozk_ctx* ctx = pw.ctx;
    // split_ck: the chunk's commitment key is ck_index restricted to the low variables, scaled by eq(t_high, id)
    const std::vector<fe>& t = h->parties[0].setup_idx->trapdoor;
    std::vector<fe> t_loc(t.begin(), t.begin() + ql);
    g1_affine g = h->parties[0].setup_idx->g;
    g1_affine gw = G1::to_affine(PST13::scalar_mul(g, eq_index_le(t, (size_t)ql, k, (uint32_t)pw.id)));
    pw.setup_slice = PST13::setup(ctx, t_loc, c.precompute, &gw);
    // rows / cols of the chunk with the padding (first term) written out: hash_tuple's own padding repeats ITS entry 0,
    // which for a chunk other than the first is not the first term of the whole vector
    std::vector<uint32_t> rp(Cn), cp(Cn);
    for (size_t i = 0; i < Cn; i++) {
        size_t e = off + i;
        rp[i] = e < real ? (uint32_t)(e / 3) : 0u;
        cp[i] = e < real ? h->h_col[e] : h->h_col[0];
    }
    pw.rows_pad = upload_u32(ctx, rp);
    pw.cols_pad = upload_u32(ctx, cp);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
}

static void spartan_pub_worker_main(cozk_spartan* h, SpartanPubWorker& pw, StarNetWorker* star) {
    const int qv = h->qv, k = h->cfg.log_pub_workers, ql = qv - k;
    const size_t NZ = (size_t)1 << qv, Cn = (size_t)1 << ql, off = (size_t)pw.id * Cn;
    SpartanParty& p0 = h->parties[0];  // the resident index (read only)
    cozk_ctx* ctx = pw.ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<fe> rx, ry, coef;
    {
        Bytes req = star->receive_request();
        Reader rd(req);
        rx = rd.vec_fr();
        ry = rd.vec_fr();
        coef = rd.vec_fr();
        COZK_REQUIRE(coef.size() == 3, "spartan pub worker: v_msg length");
    }
    double t0 = now_ms();
    auto hash = [&](const VecH& idx, const VecH& eq, const fe& v) {
        uint64_t vv[4];
        fe_to_u64x4(v, vv);
        cozk_vec* g = nullptr;
        rc_check(cozk_hash_tuple(ctx, idx.h, eq.h, vv, Cn, &g), ctx, "hash_tuple");
        return VecH(g);
    };
    // ---- third_round's public tail (worker.rs:296-343): the full eq_tilde tables, then everything on the chunk
    VecH eqrx = eq_le_device(ctx, rx), eqry = eq_le_device(ctx, ry);
    VecH erx_full, ery_full;
    {
        cozk_vec *a = nullptr, *b = nullptr;
        rc_check(cozk_vec_gather(ctx, p0.rows_u32.h, eqrx.h, NZ, &a), ctx, "vec_gather");
        erx_full = VecH(a);
        rc_check(cozk_vec_gather(ctx, p0.cols_u32.h, eqry.h, NZ, &b), ctx, "vec_gather");
        ery_full = VecH(b);
    }
    VecH erx = vec_window(erx_full, off, Cn), ery = vec_window(ery_full, off, Cn);
    VecH val_w[3] = {vec_window(p0.val_pad[0], off, Cn), vec_window(p0.val_pad[1], off, Cn), vec_window(p0.val_pad[2], off, Cn)};
    VecH freq_r = vec_window(p0.freq_r, off, Cn), freq_c = vec_window(p0.freq_c, off, Cn);
    VecH dom = vec_window(p0.domain_u32, off, Cn);
    PolyH val_poly[3];
    for (int i = 0; i < 3; i++) val_poly[i] = plain_poly_from(ctx, val_w[i]);
    {
        cozk_vec* wv = nullptr;
        rc_check(cozk_vec_alloc(ctx, Cn, COZK_SCALAR_FR, &wv), ctx, "vec_alloc");
        VecH w(wv);
        rc_check(cozk_vec_binop(ctx, COZK_OP_MUL, 0, erx.h, ery.h, w.h), ctx, "eq_rx * eq_ry");
        std::vector<fe> part(3);
        for (int i = 0; i < 3; i++) {
            uint64_t a[4], b[4];
            rc_check(cozk_poly_dot_product_with_public(ctx, val_poly[i].h, w.h, a, b), ctx, "val . eq eq");
            part[(size_t)i] = fe_from_u64x4(a);
        }
        std::vector<PST13Commitment> cm = PST13::batch_commit(ctx, *pw.setup_slice, {erx.h, ery.h});
        Writer wr;
        wr.vec_fr(part);
        wr.g1(cm[0].g_product);
        wr.g1(cm[1].g_product);
        star->send_response(wr.b);
    }
    PolyH val_m;
    {
        const cozk_poly* arr[3] = {val_poly[0].h, val_poly[1].h, val_poly[2].h};
        uint64_t cf[12];
        for (int i = 0; i < 3; i++) fe_to_u64x4(coef[(size_t)i], cf + 4 * i);
        cozk_poly* vm = nullptr;
        rc_check(cozk_poly_linear_combination(ctx, arr, cf, 3, COZK_MODE_PLAIN, 0, &vm), ctx, "val_m");
        val_m = PolyH(vm);
    }
    cozk_vec* vmv = nullptr;
    rc_check(cozk_poly_share_view(ctx, val_m.h, 0, &vmv), ctx, "share_view");
    VecH val_m_vec(vmv);
    // ---- fourth_round (worker.rs:398-575) on the chunk
    fe v, x_r, x_c;
    {
        Bytes req = star->receive_request();
        Reader rd(req);
        v = rd.fr();
        x_r = rd.fr();
        x_c = rd.fr();
    }
    VecH q_row = hash(pw.rows_pad, erx_full, v), q_col = hash(pw.cols_pad, ery_full, v);
    VecH t_row = hash(dom, erx_full, v), t_col = hash(dom, ery_full, v);
    auto prove = [&](const VecH& query, const VecH& table, const VecH& m, const fe& x, VecH out[4]) {
        uint64_t xx[4];
        fe_to_u64x4(x, xx);
        cozk_vec *phi0 = nullptr, *h0 = nullptr, *phi1 = nullptr, *h1 = nullptr;
        rc_check(cozk_logup_h(ctx, table.h, m.h, xx, &phi0, &h0), ctx, "logup_h(table)");
        out[0] = VecH(h0);
        out[1] = VecH(phi0);
        rc_check(cozk_logup_h(ctx, query.h, nullptr, xx, &phi1, &h1), ctx, "logup_h(query)");
        out[2] = VecH(h1);
        out[3] = VecH(phi1);
    };
    VecH lr[4], lc[4];  // h_0, phi_0, h_1, phi_1
    prove(q_row, t_row, freq_r, x_r, lr);
    prove(q_col, t_col, freq_c, x_c, lc);
    {
        std::vector<PST13Commitment> cm = PST13::batch_commit(ctx, *pw.setup_slice, {lr[0].h, lr[2].h, lc[0].h, lc[2].h});
        Writer w;
        for (int i = 0; i < 4; i++) w.g1(cm[(size_t)i].g_product);
        star->send_response(w.b);
    }
    std::vector<fe> z_r, z_c;
    fe lam_r, lam_c;
    {
        Bytes req = star->receive_request();
        Reader rd(req);
        z_r = rd.vec_fr();
        lam_r = rd.fr();
        z_c = rd.vec_fr();
        lam_c = rd.fr();
        COZK_REQUIRE((int)z_r.size() == qv && (int)z_c.size() == qv, "spartan pub worker: lookup z length");
    }
    // partial_generate_eq: the chunk of eq(z, .) (start_eq = id * 2^log_chunk, worker.rs:507-525)
    VecH lag_r_full = eq_le_device(ctx, z_r), lag_c_full = eq_le_device(ctx, z_c);
    VecH lag_r = vec_window(lag_r_full, off, Cn), lag_c = vec_window(lag_c_full, off, Cn);
    const cozk_vec* polys[15] = {erx.h,   ery.h,   val_m_vec.h, lag_r.h, lr[0].h, lr[1].h, freq_r.h, lr[2].h,
                                 lr[3].h, lag_c.h, lc[0].h,     lc[1].h, freq_c.h, lc[2].h, lc[3].h};
    LookupProducts lp(lam_r, lam_c);
    std::vector<uint64_t> cabi = to_abi(lp.coefs);
    cozk_prodlist* pl = nullptr;
    rc_check(cozk_prodlist_create(ctx, polys, 15, cabi.data(), lp.counts.data(), lp.factors.data(), lp.coefs.size(), &pl), ctx, "prodlist_create");
    struct PlGuard {
        cozk_prodlist* p;
        ~PlGuard() { cozk_prodlist_free(p); }
    } plg{pl};
    // distributed_sumcheck_worker (worker.rs:694-724): the first qv - k rounds, then the prover state to the coordinator
    std::vector<fe> point;
    {
        uint64_t rr[4];
        for (int j = 0; j < ql; j++) {
            uint64_t ev[16];
            rc_check(cozk_prodlist_round(ctx, pl, j ? rr : nullptr, ev), ctx, "prodlist_round");
            Writer w;
            w.vec_fr({fe_from_u64x4(ev), fe_from_u64x4(ev + 4), fe_from_u64x4(ev + 8), fe_from_u64x4(ev + 12)});
            star->send_response(w.b);
            Bytes req = star->receive_request();
            Reader rd(req);
            fe r = rd.fr();
            point.push_back(r);
            fe_to_u64x4(r, rr);
        }
        std::vector<uint64_t> fv(4 * 15);
        rc_check(cozk_prodlist_final(ctx, pl, rr, fv.data()), ctx, "prodlist_final");
        std::vector<fe> finals(15);
        for (int i = 0; i < 15; i++) finals[(size_t)i] = fe_from_u64x4(fv.data() + 4 * i);
        Writer w;
        w.vec_fr(finals);
        star->send_response(w.b);
    }
    fe eta;
    {
        Bytes req = star->receive_request();
        Reader rd(req);
        eta = rd.fr();
    }
    // distributed_batch_open_poly_worker (worker.rs:745-772) on the chunk: the local quotients, the folded value, the chunk's
    // share of the 15 evaluations (the coordinator weighs it with eq(point_high, id))
    {
        const cozk_vec* all[15] = {lr[0].h, lr[2].h, lc[0].h, lc[2].h, erx.h, ery.h, val_w[0].h, val_w[1].h, val_w[2].h,
                                   freq_r.h, q_row.h, t_row.h, freq_c.h, q_col.h, t_col.h};
        std::vector<PolyH> ph;
        std::vector<const cozk_poly*> pp;
        for (int i = 0; i < 15; i++) {
            cozk_poly* p = nullptr;
            rc_check(cozk_poly_create(ctx, COZK_MODE_PLAIN, all[i], nullptr, &p), ctx, "poly_create");
            ph.emplace_back(p);
            pp.push_back(p);
        }
        std::vector<fe> pwr(9);
        pwr[0] = Fr::one();
        for (int i = 1; i < 9; i++) pwr[(size_t)i] = Fr::mul(pwr[(size_t)i - 1], eta);
        std::vector<uint64_t> pabi = to_abi(pwr);
        cozk_poly* agg = nullptr;
        rc_check(cozk_poly_linear_combination(ctx, pp.data(), pabi.data(), 9, COZK_MODE_PLAIN, 0, &agg), ctx, "aggregate_poly");
        PolyH aggh(agg);
        cozk_vec* av = nullptr;
        rc_check(cozk_poly_share_view(ctx, agg, 0, &av), ctx, "share_view");
        VecH aggv(av);
        fe folded;
        std::vector<g1_affine> pf = PST13::open(ctx, *pw.setup_slice, aggv.h, point, &folded);
        VecH chi = eq_le_device(ctx, point);
        std::vector<uint64_t> ev(4 * 15);
        rc_check(cozk_poly_batch_evaluate_at_chi(ctx, pp.data(), 15, chi.h, ev.data()), ctx, "evaluations at the point");
        std::vector<fe> evals(15);
        for (int i = 0; i < 15; i++) evals[(size_t)i] = fe_from_u64x4(ev.data() + 4 * i);
        Writer w;
        w.vec_g1(pf);
        w.fr(folded);
        w.vec_fr(evals);
        star->send_response(w.b);
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    pw.t_lookup = now_ms() - t0;
    pw.star_up = star->bytes_up;
    pw.star_down = star->bytes_down;
    pw.star_msgs = star->n_msgs;
}

// the coordinator's side: fills the lookup part of the proof exactly as the one-worker path does
static void spartan_coordinate_lookup_split(cozk_spartan* h, StarNetCoordinator& pnet, Transcript& tr, SpartanProof& pf, const std::vector<fe>& rx,
                                            const std::vector<fe>& ry, const std::vector<fe>& abc) {
    const int qv = h->qv, k = h->cfg.log_pub_workers, ql = qv - k, K = 1 << k;
    {
        Writer w;
        w.vec_fr(rx);
        w.vec_fr(ry);
        w.vec_fr(abc);
        pnet.broadcast_request(w.b);
    }
    {  // coordinator.rs:425-475: sums over the public workers
        pf.val_abc.assign(3, Fr::zero());
        g1_xyzz crx = G1::identity(), cry = G1::identity();
        for (Bytes& b : pnet.receive_responses()) {
            Reader rd(b);
            std::vector<fe> part = rd.vec_fr();
            if (part.size() != 3) throw CozkError(COZK_ERR_INTERNAL, "spartan: val_a, val_b, val_c expected");
            for (int i = 0; i < 3; i++) pf.val_abc[(size_t)i] = Fr::add(pf.val_abc[(size_t)i], part[(size_t)i]);
            crx = G1::add_mixed(crx, rd.g1());
            cry = G1::add_mixed(cry, rd.g1());
        }
        pf.c_rx = G1::to_affine(crx);
        pf.c_ry = G1::to_affine(cry);
    }
    tr.append_scalars(pf.val_abc);
    tr.append_point(pf.c_rx);
    tr.append_point(pf.c_ry);
    {
        fe v = tr.challenge_scalar(), x_r = tr.challenge_scalar(), x_c = tr.challenge_scalar();
        Writer w;
        w.fr(v);
        w.fr(x_r);
        w.fr(x_c);
        pnet.broadcast_request(w.b);
    }
    {
        g1_xyzz acc[4] = {G1::identity(), G1::identity(), G1::identity(), G1::identity()};
        for (Bytes& b : pnet.receive_responses()) {
            Reader rd(b);
            for (int i = 0; i < 4; i++) acc[i] = G1::add_mixed(acc[i], rd.g1());
        }
        for (int i = 0; i < 4; i++) {
            pf.h_comms.push_back(G1::to_affine(acc[i]));
            tr.append_point(pf.h_comms.back());
        }
    }
    fe lam_r, lam_c;
    {
        std::vector<fe> z_r = tr.challenge_vector(qv);
        lam_r = tr.challenge_scalar();
        std::vector<fe> z_c = tr.challenge_vector(qv);
        lam_c = tr.challenge_scalar();
        Writer w;
        w.vec_fr(z_r);
        w.fr(lam_r);
        w.vec_fr(z_c);
        w.fr(lam_c);
        pnet.broadcast_request(w.b);
    }
    // distributed_sumcheck_coordinator (coordinator.rs:748-811): rounds 0 .. ql - 1 are sums of the workers' messages
    std::vector<fe> point;
    for (int j = 0; j < ql; j++) {
        std::vector<fe> ev(4, Fr::zero());
        for (Bytes& b : pnet.receive_responses()) {
            Reader rd(b);
            std::vector<fe> m = rd.vec_fr();
            if (m.size() != 4) throw CozkError(COZK_ERR_INTERNAL, "spartan: lookup sumcheck message length");
            for (int t = 0; t < 4; t++) ev[(size_t)t] = Fr::add(ev[(size_t)t], m[(size_t)t]);
        }
        tr.append_scalars(ev);
        fe r = tr.challenge_scalar();
        point.push_back(r);
        pf.lk_msgs.push_back(ev);
        Writer w;
        w.fr(r);
        pnet.broadcast_request(w.b);
    }
    // ... and the last k rounds run here on the gathered prover states: polynomial i over the worker index
    {
        std::vector<std::vector<fe>> F(15, std::vector<fe>((size_t)K));
        {
            std::vector<Bytes> msgs = pnet.receive_responses();
            for (int j = 0; j < K; j++) {
                Reader rd(msgs[(size_t)j]);
                std::vector<fe> fin = rd.vec_fr();
                if (fin.size() != 15) throw CozkError(COZK_ERR_INTERNAL, "spartan: lookup sumcheck prover state length");
                for (int i = 0; i < 15; i++) F[(size_t)i][(size_t)j] = fin[(size_t)i];
            }
        }
        LookupProducts lp(lam_r, lam_c);
        for (int m = 0; m < k; m++) {
            size_t half = F[0].size() / 2;
            std::vector<fe> ev(4, Fr::zero());
            for (size_t b = 0; b < half; b++) {
                fe at[15][4];  // polynomial i at t = 0..3 of the variable being bound
                for (int i = 0; i < 15; i++) {
                    const fe &lo = F[(size_t)i][2 * b], &hi = F[(size_t)i][2 * b + 1];
                    fe d = Fr::sub(hi, lo);
                    at[i][0] = lo;
                    at[i][1] = hi;
                    at[i][2] = Fr::add(hi, d);
                    at[i][3] = Fr::add(at[i][2], d);
                }
                size_t f = 0;
                for (size_t q = 0; q < lp.coefs.size(); q++) {
                    for (int t = 0; t < 4; t++) {
                        fe prod = lp.coefs[q];
                        for (int u = 0; u < lp.counts[q]; u++) prod = Fr::mul(prod, at[lp.factors[f + (size_t)u]][t]);
                        ev[(size_t)t] = Fr::add(ev[(size_t)t], prod);
                    }
                    f += (size_t)lp.counts[q];
                }
            }
            tr.append_scalars(ev);
            fe r = tr.challenge_scalar();
            point.push_back(r);
            pf.lk_msgs.push_back(ev);
            for (int i = 0; i < 15; i++) {
                std::vector<fe> nx(half);
                for (size_t b = 0; b < half; b++) {
                    const fe &lo = F[(size_t)i][2 * b], &hi = F[(size_t)i][2 * b + 1];
                    nx[b] = Fr::add(lo, Fr::mul(r, Fr::sub(hi, lo)));
                }
                F[(size_t)i].swap(nx);
            }
        }
    }
    {
        fe eta = tr.challenge_scalar();
        Writer w;
        w.fr(eta);
        pnet.broadcast_request(w.b);
    }
    // the batched opening: chunk-local quotient commitments add up (each worker's slice carries its eq(t_high, id)); the last
    // k quotients come from the workers' folded values, committed under g^{eq(t[level + 1 ..], b)} = G_level[2b] + G_level[2b+1]
    {
        std::vector<g1_xyzz> acc((size_t)ql, G1::identity());
        std::vector<fe> folded((size_t)K);
        pf.lk_evals.assign(15, Fr::zero());
        std::vector<Bytes> msgs = pnet.receive_responses();
        for (int j = 0; j < K; j++) {
            Reader rd(msgs[(size_t)j]);
            std::vector<g1_affine> part = rd.vec_g1();
            if ((int)part.size() != ql) throw CozkError(COZK_ERR_INTERNAL, "spartan: chunk opening length");
            for (int i = 0; i < ql; i++) acc[(size_t)i] = G1::add_mixed(acc[(size_t)i], part[(size_t)i]);
            folded[(size_t)j] = rd.fr();
            std::vector<fe> ev = rd.vec_fr();
            if (ev.size() != 15) throw CozkError(COZK_ERR_INTERNAL, "spartan: chunk evaluations length");
            fe wj = eq_index_le(point, (size_t)ql, k, (uint32_t)j);
            for (int i = 0; i < 15; i++) pf.lk_evals[(size_t)i] = Fr::add(pf.lk_evals[(size_t)i], Fr::mul(wj, ev[(size_t)i]));
        }
        for (int i = 0; i < ql; i++) pf.lk_opening.push_back(G1::to_affine(acc[(size_t)i]));
        const PST13Setup& full = *h->parties[0].setup_idx;
        const std::vector<fe>& t = full.trapdoor;
        std::vector<fe> vcur = folded;
        for (int tt = 0; tt < k; tt++) {
            int level = ql + tt, m = qv - level;
            size_t hf = (size_t)1 << (m - 1);
            g1_xyzz pi = G1::identity();
            std::vector<fe> nx(hf);
            for (size_t b = 0; b < hf; b++) {
                fe q = Fr::sub(vcur[2 * b + 1], vcur[2 * b]);
                nx[b] = Fr::add(vcur[2 * b], Fr::mul(q, point[(size_t)level]));
                pi = G1::add(pi, PST13::scalar_mul(full.g, Fr::mul(q, eq_index_le(t, (size_t)level + 1, m - 1, (uint32_t)b))));
            }
            vcur.swap(nx);
            pf.lk_opening.push_back(G1::to_affine(pi));
        }
    }
    pf.has_lookup = true;
}

}  // namespace
