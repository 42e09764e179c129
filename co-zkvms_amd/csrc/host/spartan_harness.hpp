// In-process harness for BASELINE config 4 (co-noir-spartan): the worker side of
// `SpartanProverWorker::prove` (co-noir-spartan/co-spartan/src/worker.rs:119-149) restricted to the hot
// path -- zero_round (sparse Az, Bz, Cz on shares, :153-182), first_round (PST commit of share_0,
// :185-193,577-590), second_round (rep3_first_sumcheck_worker, :195-233,593-639), third_round (A(rx,.)
// build :235-249, rep3_second_sumcheck_worker :641-688, rep3_eval_poly_worker, distributed_open :774-809) --
// with the calling thread as coordinator + verifier.  cfg.lookup_round adds the PUBLIC part with one public
// worker on party 0's GPU: third_round's tail (:296-343) and fourth_round (:398-575) -- two logup lookups, the distributed
// sumcheck over the 13 products, the batch opening of 15 polynomials under ck_index -- verified as spartan/src/logup.rs:117-190
// does.  Included by harness.hip (same translation unit).
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
    // public lookup round (party 0, cfg.lookup_round): the index under ck_index (spartan/src/indexer.rs:176-231)
    std::unique_ptr<PST13Setup> setup_idx;
    VecH rows_u32, cols_u32, domain_u32;  // entry -> row / column (the real entries); the domain 0 .. 2^qv - 1
    VecH val_pad[3];                      // val_a, val_b, val_c padded to 2^qv entries
    PolyH val_poly[3];
    VecH freq_r, freq_c;                  // normalized_multiplicities of the padded rows / cols against the domain
    double t_lookup = 0;
    double t_zero = 0, t_commit = 0, t_sc1 = 0, t_build = 0, t_sc2 = 0, t_open = 0, t_total = 0;
    uint64_t star_up = 0, star_down = 0, star_msgs = 0;
    std::string error;
};

// cfg.log_pub_workers > 0: public worker j of 2^k with chunk j of the index (setup.rs split_ipk); own context and stream on
// party 0's device, reading party 0's resident index through chunk views
struct SpartanPubWorker {
    cozk_ctx* ctx = nullptr;
    int id = 0;
    std::unique_ptr<PST13Setup> setup_slice;  // split_ck: ck_index over the low qv - k variables, generator g^{eq(t_high, j)}
    VecH rows_pad, cols_pad;                  // rows / cols of the chunk, the padding spelled out with the first term (q_row, q_col)
    double t_lookup = 0;
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
    // cfg.lookup_round (LogLookupProof + the third round's public claims; coordinator.rs:475-591)
    bool has_lookup = false;
    std::vector<fe> val_abc;               // val_a, val_b, val_c = sum val * eq_rx[row] * eq_ry[col]
    g1_affine c_rx, c_ry;                  // commitments of eq_tilde_rx_chunk, eq_tilde_ry_chunk
    std::vector<g1_affine> h_comms;        // h_0, h_1 of the row lookup, h_0, h_1 of the column lookup
    std::vector<std::vector<fe>> lk_msgs;  // qv x 4 evaluations at t = 0..3
    std::vector<fe> lk_evals;              // the 9 committed + 6 public polynomials at the sumcheck's point
    std::vector<g1_affine> lk_opening;     // qv quotient commitments of the eta-batched opening
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
        if (has_lookup) {
            w.vec_fr(val_abc);
            w.g1(c_rx);
            w.g1(c_ry);
            for (const auto& c : h_comms) w.g1(c);
            w.u64(lk_msgs.size());
            for (auto& r : lk_msgs) w.vec_fr(r);
            w.vec_fr(lk_evals);
            w.vec_g1(lk_opening);
        }
        return w.b;
    }
};

}  // namespace

struct cozk_spartan {
    cozk_spartan_config cfg;
    int nparties = 1;
    size_t n = 0;
    std::vector<SpartanParty> parties;
    std::vector<SpartanPubWorker> pub;  // cfg.log_pub_workers > 0
    // host copy of the instance for the verifier (entry e = 3 row + k)
    std::vector<uint32_t> h_col;
    std::vector<fe> h_va, h_vb, h_vc;
    int qv = 0;                          // num_variables_val: entries padded to 2^qv (lookup round)
    std::vector<g1_affine> val_oracles;  // IndexVerifierKey::val_{a,b,c}_oracle (indexer.rs:205-207)
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
    if (c.lookup_round && ps.party == 0) {
        // the index (spartan/src/indexer.rs:176-231): entries padded to 2^qv, ck_index with qv variables, the val oracles,
        // multiplicities of the rows / cols padded with their first term against the domain 0 .. 2^qv - 1
        const int qv = h->qv;
        const size_t NZ = (size_t)1 << qv;
        std::vector<fe> ti((size_t)qv);
        for (int i = 0; i < qv; i++) ti[(size_t)i] = synthetic_fr_host(c.seed ^ 0x1D1D1D1Dull, (uint64_t)i);
        ps.setup_idx = PST13::setup(ctx, ti, c.precompute);
        std::vector<uint32_t> rows(nnz), dom(NZ), fr_(NZ, 0), fc_(NZ, 0);
        for (size_t e = 0; e < nnz; e++) rows[e] = (uint32_t)(e / 3);
        for (size_t i = 0; i < NZ; i++) dom[i] = (uint32_t)i;
        for (size_t e = 0; e < NZ; e++) {
            fr_[e < nnz ? rows[e] : rows[0]]++;
            fc_[e < nnz ? h->h_col[e] : h->h_col[0]]++;
        }
        ps.rows_u32 = upload_u32(ctx, rows);
        ps.cols_u32 = upload_u32(ctx, h->h_col);
        ps.domain_u32 = upload_u32(ctx, dom);
        std::vector<fe> f1(NZ), f2(NZ);
        for (size_t i = 0; i < NZ; i++) {
            f1[i] = Fr::from_u64(fr_[i]);
            f2[i] = Fr::from_u64(fc_[i]);
        }
        ps.freq_r = upload_fe(ctx, f1);
        ps.freq_c = upload_fe(ctx, f2);
        const std::vector<fe>* vals[3] = {&h->h_va, &h->h_vb, &h->h_vc};
        std::vector<cozk_vec*> vv;
        for (int k = 0; k < 3; k++) {
            std::vector<fe> pad(*vals[k]);
            pad.resize(NZ, Fr::zero());
            ps.val_pad[k] = upload_fe(ctx, pad);
            ps.val_poly[k] = plain_poly_from(ctx, ps.val_pad[k]);
            vv.push_back(ps.val_pad[k].h);
        }
        h->val_oracles.clear();
        for (const PST13Commitment& cm : PST13::batch_commit(ctx, *ps.setup_idx, vv)) h->val_oracles.push_back(cm.g_product);
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
}

// ListOfProductsOfPolynomials of the lookup sumcheck (worker.rs:469-541; append_sumcheck_polys, sumcheck.rs:459-500) over the
// 15 polynomials eq_rx, eq_ry, val_m, then per lookup (base 3: rows, base 9: columns) lagrange, h_0, phi_0, m, h_1, phi_1
struct LookupProducts {
    std::vector<fe> coefs;
    std::vector<int> counts, factors;
    void add(const fe& cf, std::initializer_list<int> idx) {
        coefs.push_back(cf);
        counts.push_back((int)idx.size());
        for (int i : idx) factors.push_back(i);
    }
    void append(int b, const fe& lam) {
        fe eta = lam;
        add(lam, {b + 1});
        eta = Fr::mul(eta, lam);
        add(eta, {b, b + 1, b + 2});
        add(Fr::neg(eta), {b, b + 3});  // degree_diff = 0: 2^-0 = 1
        add(Fr::neg(lam), {b + 4});
        eta = Fr::mul(eta, lam);
        add(eta, {b, b + 4, b + 5});
        add(Fr::neg(eta), {b});
    }
    LookupProducts(const fe& lam_r, const fe& lam_c) {
        add(Fr::one(), {0, 1, 2});
        append(3, lam_r);
        append(9, lam_c);
    }
};

// --------------------------------------------------------------------------- public lookup round (cfg.lookup_round)
// The schedule every participant follows (only party 0 computes; the others answer with empty messages, as the reference's
// inactive workers answer with defaults, worker.rs:344-361):
//   resp  val_a, val_b, val_c, C(eq_tilde_rx), C(eq_tilde_ry)          third_round's public tail (worker.rs:296-343)
//   req   v, x_r, x_c                                                   fourth_round (worker.rs:415, 478-480)
//   resp  C(h_0), C(h_1) of the row lookup, of the column lookup        (:482-505)
//   req   z_r, lambda_r, z_c, lambda_c                                  (:507-541)
//   qv x  resp 4 evaluations / req r                                    distributed_sumcheck_worker (:694-724)
//   req   eta;  resp the batched opening + the 15 evaluations           distributed_batch_open_poly_worker (:745-772)
static void spartan_lookup_worker(cozk_spartan* h, SpartanParty& ps, StarNetWorker* star, const std::vector<fe>& rx, const std::vector<fe>& ry,
                                  const fe coef[3]) {
    const int qv = h->qv;
    const size_t NZ = (size_t)1 << qv;
    cozk_ctx* ctx = ps.ctx;
    if (ps.party != 0) {
        star->send_response(Bytes());
        (void)star->receive_request();
        star->send_response(Bytes());
        (void)star->receive_request();
        for (int j = 0; j < qv; j++) {
            star->send_response(Bytes());
            (void)star->receive_request();
        }
        (void)star->receive_request();
        star->send_response(Bytes());
        return;
    }
    auto gather = [&](const VecH& idx, const VecH& src) {
        cozk_vec* g = nullptr;
        rc_check(cozk_vec_gather(ctx, idx.h, src.h, NZ, &g), ctx, "vec_gather");
        return VecH(g);
    };
    auto hash = [&](const VecH& idx, const VecH& eq, const fe& v) {
        uint64_t vv[4];
        fe_to_u64x4(v, vv);
        cozk_vec* g = nullptr;
        rc_check(cozk_hash_tuple(ctx, idx.h, eq.h, vv, NZ, &g), ctx, "hash_tuple");
        return VecH(g);
    };
    // ---- third_round's public tail: eq_tilde_{rx,ry}(_chunk), val_a, val_b, val_c, the two commitments
    VecH eqrx = eq_le_device(ctx, rx), eqry = eq_le_device(ctx, ry);
    VecH erx = gather(ps.rows_u32, eqrx), ery = gather(ps.cols_u32, eqry);
    fe val_abc[3];
    {
        cozk_vec* wv = nullptr;
        rc_check(cozk_vec_alloc(ctx, NZ, COZK_SCALAR_FR, &wv), ctx, "vec_alloc");
        VecH w(wv);
        rc_check(cozk_vec_binop(ctx, COZK_OP_MUL, 0, erx.h, ery.h, w.h), ctx, "eq_rx * eq_ry");
        for (int k = 0; k < 3; k++) {
            uint64_t a[4], b[4];
            rc_check(cozk_poly_dot_product_with_public(ctx, ps.val_poly[k].h, w.h, a, b), ctx, "val . eq eq");
            val_abc[k] = fe_from_u64x4(a);
        }
    }
    {
        std::vector<PST13Commitment> cm = PST13::batch_commit(ctx, *ps.setup_idx, {erx.h, ery.h});
        Writer w;
        w.vec_fr({val_abc[0], val_abc[1], val_abc[2]});
        w.g1(cm[0].g_product);
        w.g1(cm[1].g_product);
        star->send_response(w.b);
    }
    // val_m = v_0 val_a + v_1 val_b + v_2 val_c (worker.rs:334-337)
    PolyH val_m;
    {
        const cozk_poly* arr[3] = {ps.val_poly[0].h, ps.val_poly[1].h, ps.val_poly[2].h};
        uint64_t cf[12];
        for (int k = 0; k < 3; k++) fe_to_u64x4(coef[k], cf + 4 * k);
        cozk_poly* vm = nullptr;
        rc_check(cozk_poly_linear_combination(ctx, arr, cf, 3, COZK_MODE_PLAIN, 0, &vm), ctx, "val_m");
        val_m = PolyH(vm);
    }
    cozk_vec* vmv = nullptr;
    rc_check(cozk_poly_share_view(ctx, val_m.h, 0, &vmv), ctx, "share_view");
    VecH val_m_vec(vmv);
    // ---- fourth_round
    fe v, x_r, x_c;
    {
        Bytes req = star->receive_request();
        Reader rd(req);
        v = rd.fr();
        x_r = rd.fr();
        x_c = rd.fr();
    }
    VecH q_row = hash(ps.rows_u32, erx, v), q_col = hash(ps.cols_u32, ery, v);
    VecH t_row = hash(ps.domain_u32, erx, v), t_col = hash(ps.domain_u32, ery, v);
    // LogLookupProof::prove (logup.rs:31-80): phi_0 = x + t, h_0 = m / phi_0, phi_1 = x + q, h_1 = 1 / phi_1 (boost_degree is
    // the identity here: the table and the query have qv variables each)
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
    prove(q_row, t_row, ps.freq_r, x_r, lr);
    prove(q_col, t_col, ps.freq_c, x_c, lc);
    {
        std::vector<PST13Commitment> cm = PST13::batch_commit(ctx, *ps.setup_idx, {lr[0].h, lr[2].h, lc[0].h, lc[2].h});
        Writer w;
        for (int i = 0; i < 4; i++) w.g1(cm[i].g_product);
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
        COZK_REQUIRE((int)z_r.size() == qv && (int)z_c.size() == qv, "spartan: lookup z length");
    }
    VecH lag_r = eq_le_device(ctx, z_r), lag_c = eq_le_device(ctx, z_c);  // partial_generate_eq over the whole domain
    const cozk_vec* polys[15] = {erx.h,   ery.h,   val_m_vec.h, lag_r.h, lr[0].h, lr[1].h, ps.freq_r.h, lr[2].h,
                                 lr[3].h, lag_c.h, lc[0].h,     lc[1].h, ps.freq_c.h, lc[2].h, lc[3].h};
    LookupProducts lp(lam_r, lam_c);
    const std::vector<fe>& coefs = lp.coefs;
    const std::vector<int>&counts = lp.counts, &factors = lp.factors;
    std::vector<uint64_t> cabi = to_abi(coefs);
    cozk_prodlist* pl = nullptr;
    rc_check(cozk_prodlist_create(ctx, polys, 15, cabi.data(), counts.data(), factors.data(), coefs.size(), &pl), ctx, "prodlist_create");
    struct PlGuard {
        cozk_prodlist* p;
        ~PlGuard() { cozk_prodlist_free(p); }
    } plg{pl};
    // distributed_sumcheck_worker (worker.rs:694-724)
    std::vector<fe> point;
    {
        uint64_t rr[4];
        for (int j = 0; j < qv; j++) {
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
    }
    fe eta;
    {
        Bytes req = star->receive_request();
        Reader rd(req);
        eta = rd.fr();
    }
    // distributed_batch_open_poly_worker (worker.rs:745-772): the 9 committed polynomials batched with powers of eta and opened
    // at the sumcheck's point, the evaluations of all 15
    {
        const cozk_vec* all[15] = {lr[0].h, lr[2].h, lc[0].h, lc[2].h, erx.h, ery.h, ps.val_pad[0].h, ps.val_pad[1].h, ps.val_pad[2].h,
                                   ps.freq_r.h, q_row.h, t_row.h, ps.freq_c.h, q_col.h, t_col.h};
        std::vector<PolyH> ph;
        std::vector<const cozk_poly*> pp;
        for (int i = 0; i < 15; i++) {
            cozk_poly* p = nullptr;
            rc_check(cozk_poly_create(ctx, COZK_MODE_PLAIN, all[i], nullptr, &p), ctx, "poly_create");
            ph.emplace_back(p);
            pp.push_back(p);
        }
        std::vector<fe> pw(9);
        pw[0] = Fr::one();
        for (int i = 1; i < 9; i++) pw[i] = Fr::mul(pw[i - 1], eta);
        std::vector<uint64_t> pabi = to_abi(pw);
        cozk_poly* agg = nullptr;
        rc_check(cozk_poly_linear_combination(ctx, pp.data(), pabi.data(), 9, COZK_MODE_PLAIN, 0, &agg), ctx, "aggregate_poly");
        PolyH aggh(agg);
        cozk_vec* av = nullptr;
        rc_check(cozk_poly_share_view(ctx, agg, 0, &av), ctx, "share_view");
        VecH aggv(av);
        std::vector<g1_affine> pf = PST13::open(ctx, *ps.setup_idx, aggv.h, point);
        VecH chi = eq_le_device(ctx, point);
        std::vector<uint64_t> ev(4 * 15);
        rc_check(cozk_poly_batch_evaluate_at_chi(ctx, pp.data(), 15, chi.h, ev.data()), ctx, "evaluations at the point");
        std::vector<fe> evals(15);
        for (int i = 0; i < 15; i++) evals[i] = fe_from_u64x4(ev.data() + 4 * i);
        Writer w;
        w.vec_g1(pf);
        w.vec_fr(evals);
        star->send_response(w.b);
    }
}

}  // namespace
#include "spartan_pub_workers.hpp"
namespace {

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
    if (c.lookup_round && c.log_pub_workers == 0) {
        spartan_lookup_worker(h, ps, star, rx, ry, coef);
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ps.t_lookup = now_ms() - t6;
        t6 = now_ms();
    }
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
    if (c.lookup_round) {
        // LogLookupProof::verify (spartan/src/logup.rs:117-190) as verifier.rs:124-150 calls it: sumcheck from the claimed sum
        // val_m, the final identity from the opened evaluations, the eta-batched opening under vk_index (with the trapdoor)
        const int qv = h->qv;
        const size_t NZ = (size_t)1 << qv, real = h->h_col.size();
        if (!pf.has_lookup || pf.val_abc.size() != 3 || pf.h_comms.size() != 4 || (int)pf.lk_msgs.size() != qv || pf.lk_evals.size() != 15 ||
            (int)pf.lk_opening.size() != qv) {
            why = "lookup round: malformed proof";
            return false;
        }
        if (!Fr::eq(pf.val_abc[0], ar) || !Fr::eq(pf.val_abc[1], br) || !Fr::eq(pf.val_abc[2], cr)) {
            why = "lookup round: val_a, val_b, val_c != the second sumcheck's A, B, C(rx, ry)";
            return false;
        }
        tr.append_scalars(pf.val_abc);
        tr.append_point(pf.c_rx);
        tr.append_point(pf.c_ry);
        fe v = tr.challenge_scalar(), x[2];
        x[0] = tr.challenge_scalar();
        x[1] = tr.challenge_scalar();
        for (const auto& cm : pf.h_comms) tr.append_point(cm);
        std::vector<fe> z[2];
        fe lam[2];
        for (int i = 0; i < 2; i++) {
            z[i] = tr.challenge_vector(qv);
            lam[i] = tr.challenge_scalar();
        }
        fe expected = Fr::add(Fr::add(Fr::mul(abc[0], pf.val_abc[0]), Fr::mul(abc[1], pf.val_abc[1])), Fr::mul(abc[2], pf.val_abc[2]));
        std::vector<fe> point;
        for (int j = 0; j < qv; j++) {
            const std::vector<fe>& ev = pf.lk_msgs[j];
            if (ev.size() != 4 || !Fr::eq(Fr::add(ev[0], ev[1]), expected)) {
                why = "lookup sumcheck: round " + std::to_string(j) + " g(0) + g(1) != claim";
                return false;
            }
            tr.append_scalars(ev);
            fe r = tr.challenge_scalar();
            point.push_back(r);
            expected = eval_points(ev, r);
        }
        fe eta = tr.challenge_scalar();
        const std::vector<fe>& E = pf.lk_evals;
        const fe one = Fr::one();
        fe res = Fr::mul(Fr::mul(E[4], E[5]), Fr::add(Fr::add(Fr::mul(E[6], abc[0]), Fr::mul(E[7], abc[1])), Fr::mul(E[8], abc[2])));  // aux_eval
        for (int i = 0; i < 2; i++) {
            const fe &h0 = E[2 * i], &h1 = E[2 * i + 1], &m_e = E[9 + 3 * i], &q_e = E[10 + 3 * i], &t_e = E[11 + 3 * i];
            fe eqv = eq_eval(point, z[i]);
            fe l2 = Fr::mul(lam[i], lam[i]), l3 = Fr::mul(l2, lam[i]);
            fe q0 = Fr::add(Fr::mul(h0, lam[i]), Fr::mul(Fr::mul(eqv, l2), Fr::sub(Fr::mul(h0, Fr::add(t_e, x[i])), m_e)));
            fe q1 = Fr::add(Fr::neg(Fr::mul(h1, lam[i])), Fr::mul(Fr::mul(eqv, l3), Fr::sub(Fr::mul(h1, Fr::add(q_e, x[i])), one)));
            res = Fr::add(res, Fr::add(q0, q1));
        }
        if (!Fr::eq(res, expected)) {
            why = "lookup sumcheck: final evaluation mismatch";
            return false;
        }
        // the six public polynomials at the point, recomputed (BatchOracleEval.debug_val is not trusted here)
        {
            std::vector<fe> ex = eq_table_le_host(rx), ey = eq_table_le_host(ry), chi = eq_table_le_host(point);
            std::vector<fe> erx(NZ, Fr::zero()), ery(NZ, Fr::zero());
            std::vector<uint32_t> fr_(NZ, 0), fc_(NZ, 0);
            for (size_t e = 0; e < real; e++) {
                erx[e] = ex[e / 3];
                ery[e] = ey[h->h_col[e]];
            }
            for (size_t e = 0; e < NZ; e++) {
                fr_[e < real ? e / 3 : 0]++;
                fc_[e < real ? h->h_col[e] : h->h_col[0]]++;
            }
            auto idx_fe = [](size_t i) { return Fr::from_u64((uint64_t)i); };
            fe pub[6] = {Fr::zero(), Fr::zero(), Fr::zero(), Fr::zero(), Fr::zero(), Fr::zero()};
            for (size_t e = 0; e < NZ; e++) {
                size_t r_e = e < real ? e / 3 : 0, c_e = e < real ? h->h_col[e] : h->h_col[0];
                fe qr = Fr::add(idx_fe(r_e), Fr::mul(v, erx[r_e])), qc = Fr::add(idx_fe(c_e), Fr::mul(v, ery[c_e]));
                fe tr_ = Fr::add(idx_fe(e), Fr::mul(v, erx[e])), tc = Fr::add(idx_fe(e), Fr::mul(v, ery[e]));
                pub[0] = Fr::add(pub[0], Fr::mul(Fr::from_u64(fr_[e]), chi[e]));
                pub[1] = Fr::add(pub[1], Fr::mul(qr, chi[e]));
                pub[2] = Fr::add(pub[2], Fr::mul(tr_, chi[e]));
                pub[3] = Fr::add(pub[3], Fr::mul(Fr::from_u64(fc_[e]), chi[e]));
                pub[4] = Fr::add(pub[4], Fr::mul(qc, chi[e]));
                pub[5] = Fr::add(pub[5], Fr::mul(tc, chi[e]));
            }
            for (int i = 0; i < 6; i++)
                if (!Fr::eq(pub[i], E[9 + i])) {
                    why = "lookup round: public evaluation " + std::to_string(i) + " (freq / query / table) mismatch";
                    return false;
                }
        }
        // batch_verify_poly (verifier.rs:256-272)
        std::vector<g1_affine> comms(pf.h_comms);
        comms.push_back(pf.c_rx);
        comms.push_back(pf.c_ry);
        for (const auto& o : h->val_oracles) comms.push_back(o);
        std::vector<fe> pw(9);
        pw[0] = one;
        for (int i = 1; i < 9; i++) pw[i] = Fr::mul(pw[i - 1], eta);
        fe batch_eval = Fr::zero();
        for (int i = 0; i < 9; i++) batch_eval = Fr::add(batch_eval, Fr::mul(pw[i], E[i]));
        g1_affine batch_comm = PST13::combine_commitments(comms, pw);
        if (!PST13::check_with_trapdoor(*h->parties[0].setup_idx, batch_comm, point, batch_eval, pf.lk_opening)) {
            why = "lookup round: batched PST13 opening check failed";
            return false;
        }
    }
    return true;
}

static int spartan_coordinator_main(cozk_spartan* h, StarNetCoordinator& net, StarNetCoordinator* pnet, SpartanProof& pf, bool verify,
                                    std::string& why) {
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
    std::vector<fe> rx, ry;
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
        rx.push_back(r);
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
        ry.push_back(r);
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
    if (c.lookup_round && c.log_pub_workers > 0) {
        COZK_REQUIRE(pnet, "spartan: the public workers' star is missing");
        spartan_coordinate_lookup_split(h, *pnet, tr, pf, rx, ry, abc);
    } else if (c.lookup_round) {
        // fourth_round, coordinator side (coordinator.rs:475-591) after third_round's public claims (:430-470)
        const int qv = h->qv;
        {
            std::vector<Bytes> m = net.receive_responses();
            Reader rd(m[0]);
            pf.val_abc = rd.vec_fr();
            pf.c_rx = rd.g1();
            pf.c_ry = rd.g1();
            if (pf.val_abc.size() != 3) throw CozkError(COZK_ERR_INTERNAL, "spartan: val_a, val_b, val_c expected");
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
            net.broadcast_request(w.b);
        }
        {
            std::vector<Bytes> m = net.receive_responses();
            Reader rd(m[0]);
            for (int i = 0; i < 4; i++) {
                pf.h_comms.push_back(rd.g1());
                tr.append_point(pf.h_comms.back());
            }
        }
        {
            std::vector<fe> z_r = tr.challenge_vector(qv);
            fe lam_r = tr.challenge_scalar();
            std::vector<fe> z_c = tr.challenge_vector(qv);
            fe lam_c = tr.challenge_scalar();
            Writer w;
            w.vec_fr(z_r);
            w.fr(lam_r);
            w.vec_fr(z_c);
            w.fr(lam_c);
            net.broadcast_request(w.b);
        }
        for (int j = 0; j < qv; j++) {  // distributed_sumcheck_coordinator (coordinator.rs:748-811), one public worker
            std::vector<Bytes> m = net.receive_responses();
            Reader rd(m[0]);
            std::vector<fe> ev = rd.vec_fr();
            if (ev.size() != 4) throw CozkError(COZK_ERR_INTERNAL, "spartan: lookup sumcheck message length");
            tr.append_scalars(ev);
            fe r = tr.challenge_scalar();
            pf.lk_msgs.push_back(ev);
            Writer w;
            w.fr(r);
            net.broadcast_request(w.b);
        }
        {
            fe eta = tr.challenge_scalar();
            Writer w;
            w.fr(eta);
            net.broadcast_request(w.b);
        }
        {
            std::vector<Bytes> m = net.receive_responses();
            Reader rd(m[0]);
            pf.lk_opening = rd.vec_g1();
            pf.lk_evals = rd.vec_fr();
        }
        pf.has_lookup = true;
    }
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
        COZK_REQUIRE(cfg->log_pub_workers >= 0 && cfg->log_pub_workers <= 3, "spartan: log_pub_workers out of range (0..3)");
        COZK_REQUIRE(cfg->log_pub_workers == 0 || cfg->lookup_round, "spartan: log_pub_workers needs lookup_round");
        h->nparties = cfg->mode == COZK_MODE_REP3 ? 3 : 1;
        h->n = (size_t)1 << cfg->log_n;
        h->qv = cfg->log_n + 2;  // 3 n entries padded to 4 n (indexer.rs:176-178)
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
        if (cfg->log_pub_workers > 0) {
            int K = 1 << cfg->log_pub_workers;
            h->pub.resize((size_t)K);
            for (int j = 0; j < K; j++) {
                SpartanPubWorker& pw = h->pub[(size_t)j];
                pw.id = j;
                // party 0's device: the workers read its resident index in place
                int rc = cozk_ctx_create(cfg->devices[0], &pw.ctx);
                if (rc != COZK_OK) throw CozkError(rc, "spartan: cannot create a public worker's context");
                HIP_TRY(hipSetDevice(pw.ctx->device));
                spartan_setup_pub_worker(h, pw);
            }
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
    for (auto& pw : h->pub) {
        if (pw.ctx) (void)hipSetDevice(pw.ctx->device);
        pw.rows_pad = VecH();
        pw.cols_pad = VecH();
        pw.setup_slice.reset();
        if (pw.ctx) cozk_ctx_destroy(pw.ctx);
    }
    for (auto& ps : h->parties) {
        if (ps.ctx) (void)hipSetDevice(ps.ctx->device);
        ps.z = PolyH();
        for (VecH* v : {&ps.row_ptr, &ps.col, &ps.va, &ps.vb, &ps.vc, &ps.t_ptr, &ps.t_row, &ps.t_va, &ps.t_vb, &ps.t_vc}) *v = VecH();
        ps.setup.reset();
        for (VecH* v : {&ps.rows_u32, &ps.cols_u32, &ps.domain_u32, &ps.val_pad[0], &ps.val_pad[1], &ps.val_pad[2], &ps.freq_r, &ps.freq_c}) *v = VecH();
        for (int k = 0; k < 3; k++) ps.val_poly[k] = PolyH();
        ps.setup_idx.reset();
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
    // the public workers' own star (log_pub_workers > 0); one abort flag for both
    const int K = (int)h->pub.size();
    InProcStar pstar(K > 0 ? K : 1);
    for (auto& ch : pstar.up) ch.abort = &star.abort;
    for (auto& ch : pstar.down) ch.abort = &star.abort;
    std::vector<std::unique_ptr<InProcStarWorker>> psw;
    for (int j = 0; j < K; j++) {
        psw.emplace_back(new InProcStarWorker(&pstar, j));
        h->pub[(size_t)j].error.clear();
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
    for (int j = 0; j < K; j++) {
        threads.emplace_back([&, j] {
            try {
                spartan_pub_worker_main(h, h->pub[(size_t)j], psw[(size_t)j].get());
            } catch (const std::exception& e) {
                h->pub[(size_t)j].error = e.what();
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
        InProcStarCoordinator coord(&star), pcoord(&pstar);
        verified = spartan_coordinator_main(h, coord, K > 0 ? &pcoord : nullptr, proof, verify != 0, why);
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
    for (int j = 0; j < K; j++) {
        if (!h->pub[(size_t)j].error.empty()) {
            h->error = "public worker " + std::to_string(j) + ": " + h->pub[(size_t)j].error;
            rc = COZK_ERR_INTERNAL;
        }
    }
    if (rc != COZK_OK) return rc;
    if (verified == 0) h->error = "verification failed: " + why;
    res->verified = verified;
    res->wall_ms = t_prove_end - t0;
    res->pub_workers = K > 0 ? K : (h->cfg.lookup_round ? 1 : 0);
    for (auto& pw : h->pub) {
        res->t_lookup_ms = std::max(res->t_lookup_ms, pw.t_lookup);
        res->pub_star_messages += pw.star_msgs;
        res->pub_bytes_up += pw.star_up;
        res->pub_bytes_down += pw.star_down;
    }
    for (int p = 0; p < np; p++) {
        SpartanParty& ps = h->parties[p];
        res->t_zero_round_ms = std::max(res->t_zero_round_ms, ps.t_zero);
        res->t_commit_ms = std::max(res->t_commit_ms, ps.t_commit);
        res->t_sumcheck1_ms = std::max(res->t_sumcheck1_ms, ps.t_sc1);
        res->t_matrix_build_ms = std::max(res->t_matrix_build_ms, ps.t_build);
        res->t_sumcheck2_ms = std::max(res->t_sumcheck2_ms, ps.t_sc2);
        res->t_open_ms = std::max(res->t_open_ms, ps.t_open);
        res->t_worker_ms = std::max(res->t_worker_ms, ps.t_total);
        res->t_lookup_ms = std::max(res->t_lookup_ms, ps.t_lookup);
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
