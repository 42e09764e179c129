// In-process harness for BASELINE config 4 (co-noir-spartan): the worker side of
// `SpartanProverWorker::prove` (co-noir-spartan/co-spartan/src/worker.rs:119-149) restricted to the hot
// path -- zero_round (sparse Az, Bz, Cz on shares, :153-182), first_round (PST commit of share_0,
// :185-193,577-590), second_round (rep3_first_sumcheck_worker, :195-233,593-639), third_round (A(rx,.)
// build :235-249, rep3_second_sumcheck_worker :641-688, rep3_eval_poly_worker, distributed_open :774-809) --
// with the calling thread as coordinator + verifier.  fourth_round (public logup over pub_ipk) is SURVEY
// 8(f) scope and is not part of this harness.  Included by harness.hip (same translation unit).
//
// Synthetic instance (no Noir front-end on the box), everything derived from `seed`, n = 2^nv rows/columns:
//   z[i]      = stream(seed + 1000)[i], z[0] = 1 (the constant column of R1CS)
//   entry e = 3 i + k of row i (k = 0, 1, 2), one index set shared by A, B, C as in the reference's
//   `rows_indexed / cols_indexed / val_{a,b,c}_indexed`:
//     col_e = small(seed + 7000, nv bits)[e] for k < 2,  0 for k = 2
//     a_e = stream(seed + 7100)[e],  b_e = stream(seed + 7200)[e]
//     c_e = stream(seed + 7300)[e] for k < 2;  for k = 2 the value that makes (Az)_i (Bz)_i = (Cz)_i
// so the instance is satisfied and the first sumcheck starts from the claim 0.
#pragma once

namespace {

static inline uint32_t synthetic_small_host(uint64_t seed, uint64_t i, int bits) {
    uint64_t s = seed + i * 0xD1342543DE82EF95ull;
    return (uint32_t)(sm_next_host(s) & (((uint64_t)1 << bits) - 1ull));
}

struct SpartanParty {
    cozk_ctx* ctx = nullptr;
    bool own_ctx = false;
    int party = 0;
    PolyH z;                               // witness shares (PLAIN: the witness itself)
    VecH row_ptr, col, va, vb, vc;         // CSR by row (zero_round)
    VecH t_ptr, t_row, t_va, t_vb, t_vc;   // CSR of the transpose: per column, the rows it touches (third_round)
    std::unique_ptr<PST13Setup> setup;
    double t_zero = 0, t_commit = 0, t_sc1 = 0, t_build = 0, t_sc2 = 0, t_open = 0, t_total = 0;
    uint64_t star_up = 0, star_down = 0, star_msgs = 0;
    std::string error;
};

struct SpartanProof {
    PST13Commitment cz;
    std::vector<std::vector<fe>> sc1;  // nv x 4 evaluations at X = 0..3
    std::vector<fe> sc1_finals;        // za(rx), zb(rx), zc(rx), eq(tau, rx)
    std::vector<std::vector<fe>> sc2;  // nv x 3 evaluations at X = 0..2
    std::vector<fe> sc2_finals;        // z(ry), A(rx, ry), B(rx, ry), C(rx, ry)
    fe z_eval;                         // rep3_eval_poly_worker's z(ry)
    std::vector<g1_affine> opening;    // nv quotient commitments
    Bytes serialize() const {
        Writer w;
        w.u64(cz.nv);
        w.g1(cz.g_product);
        w.u64(sc1.size());
        for (auto& r : sc1) w.vec_fr(r);
        w.vec_fr(sc1_finals);
        w.u64(sc2.size());
        for (auto& r : sc2) w.vec_fr(r);
        w.vec_fr(sc2_finals);
        w.fr(z_eval);
        w.vec_g1(opening);
        return w.b;
    }
};

}  // namespace

struct cozk_spartan {
    cozk_spartan_config cfg;
    int nparties = 1;
    size_t n = 0;
    std::vector<SpartanParty> parties;
    // host copy of the instance for the verifier (entry e = 3 row + k)
    std::vector<uint32_t> h_col;
    std::vector<fe> h_va, h_vb, h_vc;
    std::string error;
    Bytes last_proof;
};

namespace {

// little-endian eq table: out[idx] = prod_i (bit i of idx ? r_i : 1 - r_i)  (generate_eq, co-spartan/src/utils.rs)
static std::vector<fe> eq_table_le_host(const std::vector<fe>& r) {
    std::vector<fe> t(1, Fr::one());
    for (size_t i = 0; i < r.size(); i++) {
        size_t m = t.size();
        t.resize(2 * m);
        for (size_t j = 0; j < m; j++) {
            fe hi = Fr::mul(t[j], r[i]);
            t[j + m] = hi;
            t[j] = Fr::sub(t[j], hi);
        }
    }
    return t;
}

static VecH eq_le_device(cozk_ctx* ctx, const std::vector<fe>& r) {
    std::vector<fe> rev(r.rbegin(), r.rend());  // EqPolynomial::evals is big-endian: reversing the point flips the bit order
    std::vector<uint64_t> w = to_abi(rev);
    cozk_vec* v = nullptr;
    rc_check(cozk_eq_evals(ctx, w.data(), (int)r.size(), &v), ctx, "eq_evals");
    return VecH(v);
}

static PolyH plain_poly_from(cozk_ctx* ctx, const VecH& v) {
    cozk_poly* p = nullptr;
    rc_check(cozk_poly_create(ctx, COZK_MODE_PLAIN, v.h, nullptr, &p), ctx, "poly_create");
    return PolyH(p);
}

static VecH upload_fe(cozk_ctx* ctx, const std::vector<fe>& v) {
    std::vector<uint64_t> w = to_abi(v);
    cozk_vec* d = nullptr;
    rc_check(cozk_vec_upload(ctx, w.data(), v.size(), COZK_SCALAR_FR, &d), ctx, "vec_upload");
    return VecH(d);
}
static VecH upload_u32(cozk_ctx* ctx, const std::vector<uint32_t>& v) {
    cozk_vec* d = nullptr;
    rc_check(cozk_vec_upload(ctx, v.data(), v.size(), COZK_SCALAR_U32, &d), ctx, "vec_upload");
    return VecH(d);
}

// builds the instance on the host (every participant derives the same one), fills h->h_*
static void spartan_build_instance(cozk_spartan* h, std::vector<fe>& z_plain) {
    const cozk_spartan_config& c = h->cfg;
    size_t n = h->n, nnz = 3 * n;
    int nv = c.log_n;
    z_plain.resize(n);
    for (size_t i = 0; i < n; i++) z_plain[i] = synthetic_fr_host(c.seed + 1000ull, i);
    z_plain[0] = Fr::one();
    h->h_col.resize(nnz);
    h->h_va.resize(nnz);
    h->h_vb.resize(nnz);
    h->h_vc.resize(nnz);
    for (size_t i = 0; i < n; i++) {
        fe az = Fr::zero(), bz = Fr::zero(), cz = Fr::zero();
        for (int k = 0; k < 3; k++) {
            size_t e = 3 * i + k;
            uint32_t cj = k < 2 ? synthetic_small_host(c.seed + 7000ull, e, nv) : 0u;
            h->h_col[e] = cj;
            h->h_va[e] = synthetic_fr_host(c.seed + 7100ull, e);
            h->h_vb[e] = synthetic_fr_host(c.seed + 7200ull, e);
            az = Fr::add(az, Fr::mul(h->h_va[e], z_plain[cj]));
            bz = Fr::add(bz, Fr::mul(h->h_vb[e], z_plain[cj]));
            if (k < 2) {
                h->h_vc[e] = synthetic_fr_host(c.seed + 7300ull, e);
                cz = Fr::add(cz, Fr::mul(h->h_vc[e], z_plain[cj]));
            } else {
                h->h_vc[e] = Fr::sub(Fr::mul(az, bz), cz);  // z[0] = 1
            }
        }
    }
}

static void spartan_setup_party(cozk_spartan* h, SpartanParty& ps, const std::vector<fe>& z_plain) {
    const cozk_spartan_config& c = h->cfg;
    cozk_ctx* ctx = ps.ctx;
    size_t n = h->n, nnz = 3 * n;
    int nv = c.log_n;
    std::vector<fe> t(nv);
    for (int i = 0; i < nv; i++) t[i] = synthetic_fr_host(c.seed ^ 0x7A7A7A7Aull, (uint64_t)i);
    ps.setup = PST13::setup(ctx, t, c.precompute);
    // witness shares: P0 = (t0, t2), P1 = (t1, t0), P2 = (t2, t1), t2 = z - t0 - t1 (arithmetic.rs:21-33)
    VecH zv = upload_fe(ctx, z_plain);
    if (c.mode == COZK_MODE_PLAIN) {
        ps.z = plain_poly_from(ctx, zv);
    } else {
        // the witness scatter of the engine: t0 = PRF(k0, i), t1 = PRF(k1, i) (cozk_rep3_share_vec, ChaCha12)
        uint8_t k0[COZK_PRF_KEY_BYTES], k1[COZK_PRF_KEY_BYTES];
        harness_prf_key(c.seed + 1000ull, 101, k0);
        harness_prf_key(c.seed + 1000ull, 102, k1);
        cozk_vec *sa = nullptr, *sb = nullptr;
        rc_check(cozk_rep3_share_vec(ctx, zv.h, k0, k1, 0, ps.party, &sa, &sb), ctx, "rep3_share_vec");
        VecH a(sa), b(sb);
        cozk_poly* p = nullptr;
        rc_check(cozk_poly_create(ctx, COZK_MODE_REP3, a.h, b.h, &p), ctx, "poly_create");
        ps.z = PolyH(p);
    }
    // CSR by row: three entries per row
    std::vector<uint32_t> rp(n + 1);
    for (size_t i = 0; i <= n; i++) rp[i] = (uint32_t)(3 * i);
    ps.row_ptr = upload_u32(ctx, rp);
    ps.col = upload_u32(ctx, h->h_col);
    ps.va = upload_fe(ctx, h->h_va);
    ps.vb = upload_fe(ctx, h->h_vb);
    ps.vc = upload_fe(ctx, h->h_vc);
    // transpose (counting sort by column; field sums are order-independent)
    std::vector<uint32_t> tp(n + 1, 0), trow(nnz);
    for (size_t e = 0; e < nnz; e++) tp[h->h_col[e] + 1]++;
    for (size_t j = 0; j < n; j++) tp[j + 1] += tp[j];
    std::vector<uint32_t> cur(tp.begin(), tp.end() - 1);
    std::vector<fe> ta(nnz), tb(nnz), tc(nnz);
    for (size_t e = 0; e < nnz; e++) {
        uint32_t pos = cur[h->h_col[e]]++;
        trow[pos] = (uint32_t)(e / 3);
        ta[pos] = h->h_va[e];
        tb[pos] = h->h_vb[e];
        tc[pos] = h->h_vc[e];
    }
    ps.t_ptr = upload_u32(ctx, tp);
    ps.t_row = upload_u32(ctx, trow);
    ps.t_va = upload_fe(ctx, ta);
    ps.t_vb = upload_fe(ctx, tb);
    ps.t_vc = upload_fe(ctx, tc);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
}

// --------------------------------------------------------------------------- worker
static void spartan_worker_main(cozk_spartan* h, SpartanParty& ps, StarNetWorker* star) {
    const cozk_spartan_config& c = h->cfg;
    cozk_ctx* ctx = ps.ctx;
    int nv = c.log_n;
    WorkerEnv env;
    env.ctx = ctx;
    env.mode = c.mode;
    env.party = ps.party;
    env.star = star;
    env.ring = nullptr;  // neither sumcheck multiplies two secrets across parties: no reshare on this path
    harness_prf_key(c.seed, (uint64_t)ps.party, env.key_self);
    harness_prf_key(c.seed, (uint64_t)((ps.party + 2) % 3), env.key_prev);
    HIP_TRY(hipSetDevice(ctx->device));
    double t0 = now_ms();
    // ---- zero_round (worker.rs:153-182)
    cozk_poly *za = nullptr, *zb = nullptr, *zc = nullptr;
    rc_check(cozk_sparse_matvec3(ctx, ps.row_ptr.h, ps.col.h, ps.va.h, ps.vb.h, ps.vc.h, ps.z.h, &za, &zb, &zc), ctx, "zero_round");
    PolyH zah(za), zbh(zb), zch(zc);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    double t1 = now_ms();
    ps.t_zero = t1 - t0;
    // ---- first_round: poly_commit_worker on share_0 (worker.rs:185-193,577-590)
    cozk_vec* zview = nullptr;
    rc_check(cozk_poly_share_view(ctx, ps.z.h, 0, &zview), ctx, "share_view");
    VecH zviewh(zview);
    {
        std::vector<PST13Commitment> cm = PST13::batch_commit(ctx, *ps.setup, {zviewh.h});
        Writer w;
        w.u64(cm[0].nv);
        w.g1(cm[0].g_product);
        star->send_response(w.b);
    }
    double t2 = now_ms();
    ps.t_commit = t2 - t1;
    // ---- second_round (worker.rs:195-233): eq(tau, .) then the degree-3 sumcheck
    std::vector<fe> tau;
    {
        Bytes req = star->receive_request();
        Reader rd(req);
        tau = rd.vec_fr();
        COZK_REQUIRE((int)tau.size() == nv, "spartan: tau length");
    }
    std::vector<fe> rx, finals1;
    {
        VecH eqv = eq_le_device(ctx, tau);
        PolyH eq = plain_poly_from(ctx, eqv);
        rx = rep3_first_sumcheck_worker(env, zah.h, zbh.h, zch.h, eq.h, finals1);
    }
    double t3 = now_ms();
    ps.t_sc1 = t3 - t2;
    // ---- third_round (worker.rs:235-300)
    fe coef[3];
    {
        Bytes req = star->receive_request();
        Reader rd(req);
        std::vector<fe> v = rd.vec_fr();
        COZK_REQUIRE(v.size() == 3, "spartan: v_msg length");
        for (int i = 0; i < 3; i++) coef[i] = v[i];
    }
    // A(rx, .), B(rx, .), C(rx, .): per column, sum of val * eq_rx[row] = transposed mat-vec with the public table
    cozk_poly *arx = nullptr, *brx = nullptr, *crx = nullptr;
    {
        VecH eqrx = eq_le_device(ctx, rx);
        PolyH eqp = plain_poly_from(ctx, eqrx);
        rc_check(cozk_sparse_matvec3(ctx, ps.t_ptr.h, ps.t_row.h, ps.t_va.h, ps.t_vb.h, ps.t_vc.h, eqp.h, &arx, &brx, &crx), ctx, "A(rx,.) build");
    }
    PolyH arxh(arx), brxh(brx), crxh(crx);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    double t4 = now_ms();
    ps.t_build = t4 - t3;
    std::vector<fe> ry, finals2;
    {
        // the sumcheck binds its operands: work on a copy of z, the witness stays for the opening
        cozk_vec* zb_view = nullptr;
        if (c.mode == COZK_MODE_REP3) rc_check(cozk_poly_share_view(ctx, ps.z.h, 1, &zb_view), ctx, "share_view");
        VecH zbv(zb_view);
        cozk_poly* zw = nullptr;
        rc_check(cozk_poly_create(ctx, c.mode, zviewh.h, zb_view, &zw), ctx, "poly_create");
        PolyH zwh(zw);
        ry = rep3_second_sumcheck_worker(env, zwh.h, arxh.h, brxh.h, crxh.h, coef, finals2);
    }
    double t5 = now_ms();
    ps.t_sc2 = t5 - t4;
    // rep3_eval_poly_worker: additive share of z(ry)
    {
        VecH chi = eq_le_device(ctx, ry);
        uint64_t ev[4];
        const cozk_poly* arr[1] = {ps.z.h};
        rc_check(cozk_poly_batch_evaluate_at_chi(ctx, arr, 1, chi.h, ev), ctx, "eval z(ry)");
        Writer w;
        w.fr(fe_from_u64x4(ev));
        star->send_response(w.b);
    }
    // distributed_open (worker.rs:774-809): point[i] folds variable i (no reversal: ry is already LSB-first)
    {
        std::vector<g1_affine> pf = PST13::open(ctx, *ps.setup, zviewh.h, ry);
        Writer w;
        w.vec_g1(pf);
        star->send_response(w.b);
    }
    double t6 = now_ms();
    ps.t_open = t6 - t5;
    ps.t_total = t6 - t0;
    ps.star_up = star->bytes_up;
    ps.star_down = star->bytes_down;
    ps.star_msgs = star->n_msgs;
}

// --------------------------------------------------------------------------- coordinator + verifier
static fe eval_points(const std::vector<fe>& ev, const fe& r) {
    std::vector<fe> cf(ev.size());
    unipoly_from_evals(ev.data(), (int)ev.size(), cf.data());
    return unipoly_eval(cf, r);
}

static bool spartan_verify(cozk_spartan* h, const SpartanProof& pf, std::string& why) {
    const cozk_spartan_config& c = h->cfg;
    int nv = c.log_n;
    Transcript tr("cozk-spartan");
    tr.append_point(pf.cz.g_product);
    std::vector<fe> tau = tr.challenge_vector(nv);
    if ((int)pf.sc1.size() != nv || (int)pf.sc2.size() != nv || pf.sc1_finals.size() != 4 || pf.sc2_finals.size() != 4 || (int)pf.opening.size() != nv) {
        why = "malformed proof";
        return false;
    }
    fe claim = Fr::zero();  // a satisfied instance: sum_x eq(tau, x) (Az Bz - Cz)(x) = 0
    std::vector<fe> rx;
    for (int j = 0; j < nv; j++) {
        const std::vector<fe>& ev = pf.sc1[j];
        if (ev.size() != 4 || !Fr::eq(Fr::add(ev[0], ev[1]), claim)) {
            why = "first sumcheck: round " + std::to_string(j) + " g(0) + g(1) != claim";
            return false;
        }
        tr.append_scalars(ev);
        fe r = tr.challenge_scalar();
        rx.push_back(r);
        claim = eval_points(ev, r);
    }
    const fe &va = pf.sc1_finals[0], &vb = pf.sc1_finals[1], &vc = pf.sc1_finals[2], &veq = pf.sc1_finals[3];
    if (!Fr::eq(veq, eq_eval(tau, rx))) {
        why = "first sumcheck: eq(tau, rx) mismatch";
        return false;
    }
    if (!Fr::eq(claim, Fr::mul(veq, Fr::sub(Fr::mul(va, vb), vc)))) {
        why = "first sumcheck: final check failed";
        return false;
    }
    tr.append_scalars({va, vb, vc});
    std::vector<fe> abc = tr.challenge_vector(3);
    fe claim2 = Fr::add(Fr::add(Fr::mul(abc[0], va), Fr::mul(abc[1], vb)), Fr::mul(abc[2], vc));
    std::vector<fe> ry;
    for (int j = 0; j < nv; j++) {
        const std::vector<fe>& ev = pf.sc2[j];
        if (ev.size() != 3 || !Fr::eq(Fr::add(ev[0], ev[1]), claim2)) {
            why = "second sumcheck: round " + std::to_string(j) + " g(0) + g(1) != claim";
            return false;
        }
        tr.append_scalars(ev);
        fe r = tr.challenge_scalar();
        ry.push_back(r);
        claim2 = eval_points(ev, r);
    }
    const fe &vz = pf.sc2_finals[0], &ar = pf.sc2_finals[1], &br = pf.sc2_finals[2], &cr = pf.sc2_finals[3];
    if (!Fr::eq(claim2, Fr::mul(vz, Fr::add(Fr::add(Fr::mul(abc[0], ar), Fr::mul(abc[1], br)), Fr::mul(abc[2], cr))))) {
        why = "second sumcheck: final check failed";
        return false;
    }
    // the verifier's own A(rx, ry), B(rx, ry), C(rx, ry) from the public matrices
    {
        std::vector<fe> ex = eq_table_le_host(rx), ey = eq_table_le_host(ry);
        fe sa = Fr::zero(), sb = Fr::zero(), sc = Fr::zero();
        for (size_t e = 0; e < h->h_col.size(); e++) {
            fe w = Fr::mul(ex[e / 3], ey[h->h_col[e]]);
            sa = Fr::add(sa, Fr::mul(h->h_va[e], w));
            sb = Fr::add(sb, Fr::mul(h->h_vb[e], w));
            sc = Fr::add(sc, Fr::mul(h->h_vc[e], w));
        }
        if (!Fr::eq(sa, ar) || !Fr::eq(sb, br) || !Fr::eq(sc, cr)) {
            why = "matrix evaluation A/B/C(rx, ry) mismatch";
            return false;
        }
    }
    if (!Fr::eq(pf.z_eval, vz)) {
        why = "z(ry) from the evaluation round != the sumcheck's final z";
        return false;
    }
    if (!PST13::check_with_trapdoor(*h->parties[0].setup, pf.cz.g_product, ry, vz, pf.opening)) {
        why = "PST13 opening check failed";
        return false;
    }
    return true;
}

static int spartan_coordinator_main(cozk_spartan* h, StarNetCoordinator& net, SpartanProof& pf, bool verify, std::string& why) {
    const cozk_spartan_config& c = h->cfg;
    int nv = c.log_n;
    Transcript tr("cozk-spartan");
    {
        std::vector<PST13Commitment> shares;
        for (Bytes& b : net.receive_responses()) {
            Reader rd(b);
            PST13Commitment cm;
            cm.nv = rd.u64();
            cm.g_product = rd.g1();
            shares.push_back(cm);
        }
        pf.cz = PST13::combine_commitment_shares(shares);
        tr.append_point(pf.cz.g_product);
    }
    std::vector<fe> tau = tr.challenge_vector(nv);
    {
        Writer w;
        w.vec_fr(tau);
        net.broadcast_request(w.b);
    }
    auto sum_finals = [&](std::vector<fe>& out, int n_secret) {
        // four values per party: the first n_secret are additive (share_0) and sum, the rest are public and equal
        std::vector<Bytes> msgs = net.receive_responses();
        out.assign(4, Fr::zero());
        for (size_t p = 0; p < msgs.size(); p++) {
            Reader rd(msgs[p]);
            for (int i = 0; i < 4; i++) {
                fe v = rd.fr();
                if (i < n_secret) out[i] = Fr::add(out[i], v);
                else if (p == 0) out[i] = v;
            }
        }
    };
    // first sumcheck: 4 additive evaluations per party per round
    for (int j = 0; j < nv; j++) {
        std::vector<fe> ev(4, Fr::zero());
        for (Bytes& b : net.receive_responses()) {
            Reader rd(b);
            std::vector<fe> m = rd.vec_fr();
            if (m.size() != 4) throw CozkError(COZK_ERR_INTERNAL, "spartan: first sumcheck message length");
            for (int t = 0; t < 4; t++) ev[t] = Fr::add(ev[t], m[t]);
        }
        tr.append_scalars(ev);
        fe r = tr.challenge_scalar();
        pf.sc1.push_back(ev);
        Writer w;
        w.fr(r);
        net.broadcast_request(w.b);
    }
    sum_finals(pf.sc1_finals, 3);  // (za, zb, zc secret; eq public)
    tr.append_scalars({pf.sc1_finals[0], pf.sc1_finals[1], pf.sc1_finals[2]});
    std::vector<fe> abc = tr.challenge_vector(3);
    {
        Writer w;
        w.vec_fr(abc);
        net.broadcast_request(w.b);
    }
    // second sumcheck: 3 Rep3 evaluations (a, b) per party per round; the a components are additive
    for (int j = 0; j < nv; j++) {
        std::vector<fe> ev(3, Fr::zero());
        for (Bytes& b : net.receive_responses()) {
            Reader rd(b);
            if (rd.u64() != 3) throw CozkError(COZK_ERR_INTERNAL, "spartan: second sumcheck message length");
            for (int t = 0; t < 3; t++) {
                ev[t] = Fr::add(ev[t], rd.fr());
                (void)rd.fr();
            }
        }
        tr.append_scalars(ev);
        fe r = tr.challenge_scalar();
        pf.sc2.push_back(ev);
        Writer w;
        w.fr(r);
        net.broadcast_request(w.b);
    }
    sum_finals(pf.sc2_finals, 1);  // (z secret; A, B, C public)
    pf.z_eval = Fr::zero();
    for (Bytes& b : net.receive_responses()) {
        Reader rd(b);
        pf.z_eval = Fr::add(pf.z_eval, rd.fr());
    }
    pf.opening = PST13::coordinate_prove(net);
    if (!verify) return -1;
    return spartan_verify(h, pf, why) ? 1 : 0;
}

}  // namespace

extern "C" {

int cozk_spartan_create(const cozk_spartan_config* cfg, cozk_spartan** out) {
    if (!cfg || !out) return COZK_ERR_INVALID_ARG;
    *out = nullptr;
    cozk_spartan* h = new cozk_spartan();
    h->cfg = *cfg;
    try {
        COZK_REQUIRE(cfg->mode == COZK_MODE_PLAIN || cfg->mode == COZK_MODE_REP3, "spartan: bad mode");
        COZK_REQUIRE(cfg->log_n >= 2 && cfg->log_n <= 24, "spartan: log_n out of range");
        h->nparties = cfg->mode == COZK_MODE_REP3 ? 3 : 1;
        h->n = (size_t)1 << cfg->log_n;
        std::vector<fe> z_plain;
        spartan_build_instance(h, z_plain);
        h->parties.resize((size_t)h->nparties);
        for (int p = 0; p < h->nparties; p++) {
            SpartanParty& ps = h->parties[p];
            ps.party = p;
            int rc = cozk_ctx_create(cfg->devices[p], &ps.ctx);
            if (rc != COZK_OK) throw CozkError(rc, "spartan: cannot create a context (no HIP device?)");
            ps.own_ctx = true;
            HIP_TRY(hipSetDevice(ps.ctx->device));
            spartan_setup_party(h, ps, z_plain);
        }
    } catch (const CozkError& e) {
        h->error = e.what();
        *out = h;
        return e.code;
    } catch (const std::exception& e) {
        h->error = e.what();
        *out = h;
        return COZK_ERR_INTERNAL;
    }
    *out = h;
    return COZK_OK;
}

const char* cozk_spartan_error(const cozk_spartan* h) { return h ? h->error.c_str() : "null harness"; }

int cozk_spartan_destroy(cozk_spartan* h) {
    if (!h) return COZK_OK;
    for (auto& ps : h->parties) {
        if (ps.ctx) (void)hipSetDevice(ps.ctx->device);
        ps.z = PolyH();
        for (VecH* v : {&ps.row_ptr, &ps.col, &ps.va, &ps.vb, &ps.vc, &ps.t_ptr, &ps.t_row, &ps.t_va, &ps.t_vb, &ps.t_vc}) *v = VecH();
        ps.setup.reset();
        if (ps.own_ctx && ps.ctx) cozk_ctx_destroy(ps.ctx);
    }
    delete h;
    return COZK_OK;
}

int cozk_spartan_prove(cozk_spartan* h, int verify, cozk_spartan_result* res) {
    if (!h || !res) return COZK_ERR_INVALID_ARG;
    memset(res, 0, sizeof *res);
    res->verified = -1;
    int np = h->nparties;
    InProcStar star(np);
    std::vector<std::unique_ptr<InProcStarWorker>> sw;
    for (int p = 0; p < np; p++) {
        sw.emplace_back(new InProcStarWorker(&star, p));
        h->parties[p].error.clear();
    }
    std::vector<std::thread> threads;
    double t0 = now_ms();
    for (int p = 0; p < np; p++) {
        threads.emplace_back([&, p] {
            try {
                spartan_worker_main(h, h->parties[p], sw[p].get());
            } catch (const std::exception& e) {
                h->parties[p].error = e.what();
                star.abort.flag.store(true);
            }
        });
    }
    SpartanProof proof;
    std::string why;
    int verified = -1;
    int rc = COZK_OK;
    double t_prove_end = 0;
    try {
        InProcStarCoordinator coord(&star);
        verified = spartan_coordinator_main(h, coord, proof, verify != 0, why);
    } catch (const std::exception& e) {
        h->error = std::string("coordinator: ") + e.what();
        star.abort.flag.store(true);
        rc = COZK_ERR_INTERNAL;
    }
    for (auto& t : threads) t.join();
    t_prove_end = now_ms();
    for (int p = 0; p < np; p++) {
        if (!h->parties[p].error.empty()) {
            h->error = "party " + std::to_string(p) + ": " + h->parties[p].error;
            rc = COZK_ERR_INTERNAL;
        }
    }
    if (rc != COZK_OK) return rc;
    if (verified == 0) h->error = "verification failed: " + why;
    res->verified = verified;
    res->wall_ms = t_prove_end - t0;
    for (int p = 0; p < np; p++) {
        SpartanParty& ps = h->parties[p];
        res->t_zero_round_ms = std::max(res->t_zero_round_ms, ps.t_zero);
        res->t_commit_ms = std::max(res->t_commit_ms, ps.t_commit);
        res->t_sumcheck1_ms = std::max(res->t_sumcheck1_ms, ps.t_sc1);
        res->t_matrix_build_ms = std::max(res->t_matrix_build_ms, ps.t_build);
        res->t_sumcheck2_ms = std::max(res->t_sumcheck2_ms, ps.t_sc2);
        res->t_open_ms = std::max(res->t_open_ms, ps.t_open);
        res->t_worker_ms = std::max(res->t_worker_ms, ps.t_total);
        res->bytes_star_up += ps.star_up;
        res->bytes_star_down += ps.star_down;
        res->star_messages += ps.star_msgs;
    }
    h->last_proof = proof.serialize();
    res->proof_len = h->last_proof.size();
    Sha256 s;
    s.update(h->last_proof.data(), h->last_proof.size());
    s.final(res->proof_digest);
    return COZK_OK;
}

int cozk_spartan_proof_bytes(const cozk_spartan* h, uint8_t* out, size_t cap) {
    if (!h || !out || cap < h->last_proof.size()) return COZK_ERR_INVALID_ARG;
    memcpy(out, h->last_proof.data(), h->last_proof.size());
    return COZK_OK;
}

}  // extern "C"
