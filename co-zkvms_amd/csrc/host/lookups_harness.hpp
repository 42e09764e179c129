// In-process harness of the instruction-lookups memory checking: the toggled / sparse batched grand
// product of Lasso's read / write circuits (co-jolt/src/jolt/vm/instruction_lookups/worker.rs:763-859 compute_leaves +
// co-jolt/src/subprotocols/sparse_grand_product.rs), worker side on the GPU(s), coordinator + plain verifier on the calling
// thread.  Synthetic restatement (no Jolt tracer on the box): n_pairs "memories", each with one public 0/1 flag column
// of N = 2^log_n cycles (density_pct % of the entries set) shared by its read and its write circuit, and one shared
// fingerprint vector per circuit:
//   flag column q, entry i  = (byte(seed + 4000 (q + 1), i) * 100 < density_pct * 256)      (SplitMix64 stream bytes)
//   fingerprints of circuit b = stream(seed + 7000 (b + 1)), shared through cozk_rep3_share_vec (harness keys)
// The verifier replays the transcript, checks every round and layer reduction, and checks the toggle layer's final
// claims against direct evaluations of the flag and fingerprint polynomials (flags padded with ones, fingerprints with
// zeros, as sparse_grand_product.rs:118-131 pads them).  oracle/pylookups.py restates the pipeline over the SPARSE oracle.
#pragma once

struct LookupsParty {
    cozk_ctx* ctx = nullptr;
    bool own_ctx = false;
    int party = 0;
    std::vector<VecH> flags;  // n_pairs U8 columns
    VecH fp_a, fp_b;          // 2 * n_pairs circuits x N, circuit-major
    // primary sumcheck (cfg.primary): instruction flags, E polynomials, lookup_outputs
    std::vector<VecH> instr_flags;
    std::vector<PolyH> E;
    PolyH outputs;
    double t_primary = 0, t_construct = 0, t_prove = 0, t_total = 0;
    uint64_t star_up = 0, star_down = 0, star_msgs = 0, ring_bytes = 0;
    std::string error;
};

struct cozk_lookups {
    cozk_lookups_config cfg;
    int nparties = 1;
    size_t N = 0, batch = 0;
    std::vector<LookupsParty> parties;
    // the verifier's own view: plain flags (as Fr polynomials) and plain fingerprints, on its own context
    cozk_ctx* vctx = nullptr;
    std::vector<PolyH> v_flags, v_fps;
    std::vector<cozk_primary_instr> instrs;
    std::vector<uint8_t> which;            // instruction of every cycle
    std::vector<fe> outputs_plain;         // lookup_outputs in the clear (the dealer's view)
    std::vector<PolyH> v_E, v_iflags;
    PolyH v_outputs;
    std::string error;
    Bytes last_proof;
};

namespace {

std::vector<uint8_t> lookups_flag_column(const cozk_lookups_config& c, int q, size_t n) {
    std::vector<uint8_t> col(n);
    const uint64_t seed = c.seed + 4000ull * (uint64_t)(q + 1);
    for (size_t i = 0; i < n; i++) col[i] = (uint32_t)synthetic_small_host(seed, i, 8) * 100u < (uint32_t)c.density_pct * 256u ? 1 : 0;
    return col;
}

// the synthetic instruction table of the primary sumcheck: three CONCAT (AND / OR / XOR-like), one PRODUCT (BEQ-like) and
// one LTU (SLTU-like) instruction with C = 4 chunks over memory indices taken modulo n_mem
std::vector<cozk_primary_instr> lookups_instr_table(int n_mem) {
    auto mk = [&](int form, std::initializer_list<int> mems, int bits) {
        cozk_primary_instr in{};
        in.form = form;
        in.bits = bits;
        in.n_mems = 0;
        for (int m : mems) in.mems[in.n_mems++] = m % n_mem;
        return in;
    };
    return {mk(COZK_G_CONCAT, {0, 1, 2, 3}, 8), mk(COZK_G_CONCAT, {4, 5, 6, 7}, 8), mk(COZK_G_CONCAT, {4, 1, 6, 3}, 4), mk(COZK_G_PRODUCT, {8, 9, 10, 11}, 0),
            mk(COZK_G_LTU, {12, 13, 14, 15, 16, 17, 18}, 0)};
}

// the dealer's view: which instruction every cycle runs and lookup_outputs(x) = g_{which(x)}(E(x)) in the clear
void lookups_setup_primary_clear(cozk_lookups* h) {
    const cozk_lookups_config& c = h->cfg;
    h->instrs = lookups_instr_table(c.n_pairs);
    h->which.resize(h->N);
    h->outputs_plain.resize(h->N);
    std::vector<fe> E((size_t)c.n_pairs);
    for (size_t x = 0; x < h->N; x++) {
        uint8_t w = (uint8_t)(synthetic_small_host(c.seed + 1234567ull, x, 8) % h->instrs.size());
        h->which[x] = w;
        const cozk_primary_instr& in = h->instrs[w];
        for (int t = 0; t < in.n_mems; t++) E[in.mems[t]] = synthetic_fr_host(c.seed + 9000ull * (uint64_t)(in.mems[t] + 1), x);
        h->outputs_plain[x] = primary_g_plain(in, E);
    }
}

void lookups_setup_primary_party(cozk_lookups* h, LookupsParty& ps) {
    const cozk_lookups_config& c = h->cfg;
    cozk_ctx* ctx = ps.ctx;
    for (size_t i = 0; i < h->instrs.size(); i++) {
        std::vector<uint8_t> col(h->N);
        for (size_t x = 0; x < h->N; x++) col[x] = h->which[x] == i ? 1 : 0;
        cozk_vec* v = nullptr;
        rc_check(cozk_vec_upload(ctx, col.data(), h->N, COZK_SCALAR_U8, &v), ctx, "vec_upload(instruction flags)");
        ps.instr_flags.push_back(VecH(v));
    }
    for (int m = 0; m < c.n_pairs; m++) {
        VecH a, b;
        make_share_vectors(ctx, h->N, c.seed + 9000ull * (uint64_t)(m + 1), ps.party, c.mode, a, b);
        cozk_poly* p = nullptr;
        rc_check(cozk_poly_create(ctx, c.mode, a.h, b.h, &p), ctx, "poly_create(E)");
        ps.E.push_back(PolyH(p));
    }
    cozk_vec* ov = nullptr;
    rc_check(cozk_vec_upload(ctx, h->outputs_plain.data(), h->N, COZK_SCALAR_FR, &ov), ctx, "vec_upload(outputs)");
    VecH ovh(ov);
    cozk_poly* op = nullptr;
    if (c.mode == COZK_MODE_REP3) {
        uint8_t k0[COZK_PRF_KEY_BYTES], k1[COZK_PRF_KEY_BYTES];
        harness_prf_key(c.seed + 555ull, 101, k0);
        harness_prf_key(c.seed + 555ull, 102, k1);
        cozk_vec *sa = nullptr, *sb = nullptr;
        rc_check(cozk_rep3_share_vec(ctx, ovh.h, k0, k1, 0, ps.party, &sa, &sb), ctx, "rep3_share_vec(outputs)");
        VecH a(sa), b(sb);
        rc_check(cozk_poly_create(ctx, COZK_MODE_REP3, a.h, b.h, &op), ctx, "poly_create(outputs)");
    } else {
        rc_check(cozk_poly_create(ctx, COZK_MODE_PLAIN, ovh.h, nullptr, &op), ctx, "poly_create(outputs)");
    }
    ps.outputs = PolyH(op);
}

void lookups_setup_primary_verifier(cozk_lookups* h) {
    const cozk_lookups_config& c = h->cfg;
    cozk_ctx* ctx = h->vctx;
    auto plain_poly = [&](const VecH& v) {
        cozk_poly* p = nullptr;
        rc_check(cozk_poly_create(ctx, COZK_MODE_PLAIN, v.h, nullptr, &p), ctx, "poly_create");
        return PolyH(p);
    };
    for (int m = 0; m < c.n_pairs; m++) h->v_E.push_back(plain_poly(make_vec_random(ctx, h->N, COZK_SCALAR_FR, c.seed + 9000ull * (uint64_t)(m + 1), 0)));
    for (size_t i = 0; i < h->instrs.size(); i++) {
        std::vector<fe> f(h->N);
        for (size_t x = 0; x < h->N; x++) f[x] = h->which[x] == i ? Fr::one() : Fr::zero();
        cozk_vec* v = nullptr;
        rc_check(cozk_vec_upload(ctx, f.data(), h->N, COZK_SCALAR_FR, &v), ctx, "vec_upload");
        VecH vh(v);
        h->v_iflags.push_back(plain_poly(vh));
    }
    cozk_vec* ov = nullptr;
    rc_check(cozk_vec_upload(ctx, h->outputs_plain.data(), h->N, COZK_SCALAR_FR, &ov), ctx, "vec_upload");
    VecH ovh(ov);
    h->v_outputs = plain_poly(ovh);
}

void lookups_setup_party(cozk_lookups* h, LookupsParty& ps) {
    const cozk_lookups_config& c = h->cfg;
    cozk_ctx* ctx = ps.ctx;
    for (int q = 0; q < c.n_pairs; q++) {
        std::vector<uint8_t> col = lookups_flag_column(c, q, h->N);
        cozk_vec* v = nullptr;
        rc_check(cozk_vec_upload(ctx, col.data(), h->N, COZK_SCALAR_U8, &v), ctx, "vec_upload(flags)");
        ps.flags.push_back(VecH(v));
    }
    cozk_vec *fa = nullptr, *fb = nullptr;
    rc_check(cozk_vec_alloc(ctx, h->batch * h->N, COZK_SCALAR_FR, &fa), ctx, "vec_alloc");
    ps.fp_a = VecH(fa);
    if (c.mode == COZK_MODE_REP3) {
        rc_check(cozk_vec_alloc(ctx, h->batch * h->N, COZK_SCALAR_FR, &fb), ctx, "vec_alloc");
        ps.fp_b = VecH(fb);
    }
    for (size_t b = 0; b < h->batch; b++) {
        VecH a, bb;
        make_share_vectors(ctx, h->N, c.seed + 7000ull * (uint64_t)(b + 1), ps.party, c.mode, a, bb);
        HIP_TRY(hipMemcpyAsync((fe*)cozk_vec_device_ptr(ps.fp_a.h) + b * h->N, cozk_vec_device_ptr(a.h), h->N * sizeof(fe), hipMemcpyDeviceToDevice, ctx->stream));
        if (c.mode == COZK_MODE_REP3)
            HIP_TRY(hipMemcpyAsync((fe*)cozk_vec_device_ptr(ps.fp_b.h) + b * h->N, cozk_vec_device_ptr(bb.h), h->N * sizeof(fe), hipMemcpyDeviceToDevice,
                                   ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
}

void lookups_setup_verifier(cozk_lookups* h) {
    const cozk_lookups_config& c = h->cfg;
    cozk_ctx* ctx = h->vctx;
    for (int q = 0; q < c.n_pairs; q++) {
        std::vector<uint8_t> col = lookups_flag_column(c, q, h->N);
        std::vector<fe> f(h->N);
        for (size_t i = 0; i < h->N; i++) f[i] = col[i] ? Fr::one() : Fr::zero();
        cozk_vec* v = nullptr;
        rc_check(cozk_vec_upload(ctx, f.data(), h->N, COZK_SCALAR_FR, &v), ctx, "vec_upload");
        VecH vh(v);
        cozk_poly* p = nullptr;
        rc_check(cozk_poly_create(ctx, COZK_MODE_PLAIN, vh.h, nullptr, &p), ctx, "poly_create");
        h->v_flags.push_back(PolyH(p));
    }
    for (size_t b = 0; b < h->batch; b++) {
        VecH v = make_vec_random(ctx, h->N, COZK_SCALAR_FR, c.seed + 7000ull * (uint64_t)(b + 1), 0);
        cozk_poly* p = nullptr;
        rc_check(cozk_poly_create(ctx, COZK_MODE_PLAIN, v.h, nullptr, &p), ctx, "poly_create");
        h->v_fps.push_back(PolyH(p));
    }
}

struct LookupsProof {
    bool has_primary = false;
    PrimarySumcheckProof primary;
    GrandProductProof gp;
    Bytes serialize() const {
        Writer w;
        if (has_primary) {
            w.u64(primary.compressed_polys.size());
            for (auto& p : primary.compressed_polys) w.vec_fr(p);
            w.vec_fr(primary.openings);
        }
        w.vec_fr(gp.outputs);
        w.u64(gp.gkr_layers.size());
        for (auto& l : gp.gkr_layers) {
            w.u64(l.proof.compressed_polys.size());
            for (auto& p : l.proof.compressed_polys) w.vec_fr(p);
            w.fr(l.left_claim);
            w.fr(l.right_claim);
        }
        return w.b;
    }
};

void lookups_worker_main(cozk_lookups* h, LookupsParty& ps, StarNetWorker* star, RingNet* ring) {
    const cozk_lookups_config& c = h->cfg;
    WorkerEnv env;
    env.ctx = ps.ctx;
    env.mode = c.mode;
    env.party = ps.party;
    env.star = star;
    env.ring = ring;
    harness_prf_key(c.seed, (uint64_t)ps.party, env.key_self);
    harness_prf_key(c.seed, (uint64_t)((ps.party + 2) % 3), env.key_prev);
    HIP_TRY(hipSetDevice(ps.ctx->device));
    double tp0 = now_ms();
    if (c.primary) {
        // ---- Lasso primary sumcheck (jolt/vm/instruction_lookups/worker.rs:95-141): r_eq from the coordinator, eq table,
        //      then prove_primary_sumcheck_inner
        Bytes req = env.star->receive_request();
        Reader rd(req);
        std::vector<fe> r_eq = rd.vec_fr();
        std::vector<uint64_t> w = to_abi(r_eq);
        cozk_vec* eqv = nullptr;
        rc_check(cozk_eq_evals(env.ctx, w.data(), (int)r_eq.size(), &eqv), env.ctx, "eq_evals");
        VecH eqh(eqv);
        std::vector<const cozk_vec*> fl;
        for (auto& f : ps.instr_flags) fl.push_back(f.h);
        std::vector<const cozk_poly*> E;
        for (auto& e : ps.E) E.push_back(e.h);
        cozk_primary* pr = nullptr;
        rc_check(cozk_primary_create(env.ctx, c.mode, ps.party, h->instrs.data(), h->instrs.size(), fl.data(), E.data(), E.size(), ps.outputs.h, eqh.h, &pr),
                 env.ctx, "primary_create");
        PrimaryH prh(pr);
        (void)prove_primary_sumcheck_worker(env, pr, c.log_n, E.size(), h->instrs.size());
    }
    double t0 = now_ms();
    ps.t_primary = t0 - tp0;
    // the leaves are consumed by the prover: work on copies of the resident fingerprints
    cozk_vec *fa = nullptr, *fb = nullptr;
    size_t total = h->batch * h->N;
    rc_check(cozk_vec_alloc(env.ctx, total, COZK_SCALAR_FR, &fa), env.ctx, "vec_alloc");
    VecH a(fa), b;
    HIP_TRY(hipMemcpyAsync(cozk_vec_device_ptr(a.h), cozk_vec_device_ptr(ps.fp_a.h), total * sizeof(fe), hipMemcpyDeviceToDevice, env.ctx->stream));
    if (c.mode == COZK_MODE_REP3) {
        rc_check(cozk_vec_alloc(env.ctx, total, COZK_SCALAR_FR, &fb), env.ctx, "vec_alloc");
        b = VecH(fb);
        HIP_TRY(hipMemcpyAsync(cozk_vec_device_ptr(b.h), cozk_vec_device_ptr(ps.fp_b.h), total * sizeof(fe), hipMemcpyDeviceToDevice, env.ctx->stream));
    }
    std::vector<const cozk_vec*> fl;
    for (auto& f : ps.flags) fl.push_back(f.h);
    cozk_toggle* t = nullptr;
    rc_check(cozk_toggle_create(env.ctx, c.mode, fl.data(), fl.size(), a.h, b.h, 1, &t), env.ctx, "toggle_create");
    Rep3ToggledBatchedGrandProduct gp = Rep3ToggledBatchedGrandProduct::construct(env, ToggleH(t));
    rc_check(cozk_ctx_synchronize(env.ctx), env.ctx, "sync");
    double t1 = now_ms();
    ps.t_construct = t1 - t0;
    (void)gp.prove_grand_product_worker(env);
    double t2 = now_ms();
    ps.t_prove = t2 - t1;
    ps.t_total = t2 - tp0;
    ps.star_up = star->bytes_up;
    ps.star_down = star->bytes_down;
    ps.star_msgs = star->n_msgs;
    ps.ring_bytes = ring ? ring->bytes_sent : 0;
}

// MLE of `polys` (per circuit, length N each, circuit-major, padded to L circuits with `pad`) at the big-endian point r
fe lookups_eval_circuit_major(cozk_lookups* h, const std::vector<const cozk_poly*>& polys, const std::vector<fe>& r, bool pad_with_ones) {
    size_t L = 1;
    while (L < polys.size()) L <<= 1;
    int hi = 0;
    while (((size_t)1 << hi) < L) hi++;
    COZK_REQUIRE(r.size() == (size_t)hi + (size_t)h->cfg.log_n, "lookups verifier: point length");
    std::vector<fe> r_hi(r.begin(), r.begin() + hi), r_lo(r.begin() + hi, r.end());
    std::vector<uint64_t> w = to_abi(r_lo);
    cozk_vec* chi = nullptr;
    rc_check(cozk_eq_evals(h->vctx, w.data(), (int)r_lo.size(), &chi), h->vctx, "eq_evals");
    VecH chih(chi);
    std::vector<uint64_t> out(4 * polys.size());
    rc_check(cozk_poly_batch_evaluate_at_chi(h->vctx, polys.data(), polys.size(), chih.h, out.data()), h->vctx, "batch_evaluate");
    std::vector<fe> eq_hi = eq_evals_host(r_hi);
    fe acc = Fr::zero();
    for (size_t c = 0; c < L; c++) {
        fe v = c < polys.size() ? fe_from_u64x4(out.data() + 4 * c) : (pad_with_ones ? Fr::one() : Fr::zero());  // sum_i eq_lo(i) = 1
        acc = Fr::add(acc, Fr::mul(eq_hi[c], v));
    }
    return acc;
}

int lookups_coordinator_main(cozk_lookups* h, StarNetCoordinator& net, LookupsProof& proof, bool verify, std::string& why) {
    Transcript tr("cozk-lookups");
    std::vector<fe> r_eq, r_primary;
    if (h->cfg.primary) {
        r_eq = tr.challenge_vector((size_t)h->cfg.log_n);
        Writer w;
        w.vec_fr(r_eq);
        net.broadcast_request(w.b);
        proof.has_primary = true;
        proof.primary = coordinate_primary_sumcheck(net, tr, h->cfg.log_n, r_primary);
    }
    size_t num_layers = (size_t)h->cfg.log_n + 1;  // tree_depth sparse layers + the toggle layer
    std::vector<fe> r;
    proof.gp = coordinate_prove_toggled_grand_product(net, tr, num_layers, r);
    if (!verify) return -1;
    Transcript vt("cozk-lookups");
    if (h->cfg.primary) {
        std::vector<fe> vr_eq = vt.challenge_vector((size_t)h->cfg.log_n), rs;
        const int degree = 6;  // max g degree (C = 4) + 2
        if (!verify_primary_sumcheck(proof.primary, h->instrs, (size_t)h->cfg.n_pairs, degree, vr_eq, vt, rs)) {
            why = "primary sumcheck: a round or the final claim does not hold";
            return 0;
        }
        // the opened evaluations against direct evaluations of the polynomials (stand-in for the PCS opening)
        std::vector<fe> pt(rs.rbegin(), rs.rend());
        std::vector<uint64_t> w = to_abi(pt);
        HIP_TRY(hipSetDevice(h->vctx->device));
        cozk_vec* chi = nullptr;
        rc_check(cozk_eq_evals(h->vctx, w.data(), (int)pt.size(), &chi), h->vctx, "eq_evals");
        VecH chih(chi);
        std::vector<const cozk_poly*> ps;
        for (auto& e : h->v_E) ps.push_back(e.h);
        for (auto& f : h->v_iflags) ps.push_back(f.h);
        ps.push_back(h->v_outputs.h);
        std::vector<uint64_t> out(4 * ps.size());
        rc_check(cozk_poly_batch_evaluate_at_chi(h->vctx, ps.data(), ps.size(), chih.h, out.data()), h->vctx, "batch_evaluate");
        for (size_t i = 0; i < ps.size(); i++)
            if (!Fr::eq(fe_from_u64x4(out.data() + 4 * i), proof.primary.openings[i])) {
                why = "primary sumcheck: opening " + std::to_string(i) + " != polynomial(r)";
                return 0;
            }
    }
    fe flag_claim, fp_claim;
    std::vector<fe> rv;
    if (!verify_toggled_grand_product(proof.gp, vt, flag_claim, fp_claim, rv)) {
        why = "toggled grand product: a sumcheck round or a layer reduction does not hold";
        return 0;
    }
    if (rv.size() != r.size()) {
        why = "toggled grand product: point length";
        return 0;
    }
    for (size_t i = 0; i < r.size(); i++)
        if (!Fr::eq(r[i], rv[i])) {
            why = "toggled grand product: verifier derived a different point";
            return 0;
        }
    // the outputs are the products of the toggled fingerprints: checked through the final claims below (GKR soundness);
    // final claims against direct evaluations of the leaf polynomials
    std::vector<const cozk_poly*> fl, fp;
    for (size_t b = 0; b < h->batch; b++) {
        fl.push_back(h->v_flags[b / 2].h);
        fp.push_back(h->v_fps[b].h);
    }
    HIP_TRY(hipSetDevice(h->vctx->device));
    if (!Fr::eq(lookups_eval_circuit_major(h, fl, rv, true), flag_claim)) {
        why = "toggle layer: flag claim != flags(r)";
        return 0;
    }
    if (!Fr::eq(lookups_eval_circuit_major(h, fp, rv, false), fp_claim)) {
        why = "toggle layer: fingerprint claim != fingerprints(r)";
        return 0;
    }
    return 1;
}

}  // namespace

extern "C" {

int cozk_lookups_create(const cozk_lookups_config* cfg, cozk_lookups** out) {
    if (!cfg || !out) return COZK_ERR_INVALID_ARG;
    *out = nullptr;
    cozk_lookups* h = new cozk_lookups();
    h->cfg = *cfg;
    try {
        COZK_REQUIRE(cfg->mode == COZK_MODE_PLAIN || cfg->mode == COZK_MODE_REP3, "lookups: mode");
        COZK_REQUIRE(cfg->log_n >= 1 && cfg->log_n <= 24 && cfg->n_pairs >= 1 && cfg->n_pairs <= 128 && cfg->density_pct >= 0 && cfg->density_pct <= 100,
                     "lookups: log_n in 1..24, n_pairs in 1..128, density_pct in 0..100");
        h->nparties = cfg->mode == COZK_MODE_REP3 ? 3 : 1;
        h->N = (size_t)1 << cfg->log_n;
        h->batch = 2 * (size_t)cfg->n_pairs;
        h->parties.resize(h->nparties);
        if (cfg->primary) lookups_setup_primary_clear(h);
        for (int p = 0; p < h->nparties; p++) {
            LookupsParty& ps = h->parties[p];
            ps.party = p;
            int rc = cozk_ctx_create(cfg->devices[p], &ps.ctx);
            if (rc != COZK_OK) throw CozkError(rc, "lookups: cannot create a context (no HIP device?)");
            ps.own_ctx = true;
            cozk_ctx_set_resident_rounds(ps.ctx, h->nparties > 1 ? 0 : 1);
            HIP_TRY(hipSetDevice(ps.ctx->device));
            lookups_setup_party(h, ps);
            if (cfg->primary) lookups_setup_primary_party(h, ps);
        }
        int rc = cozk_ctx_create(cfg->devices[0], &h->vctx);
        if (rc != COZK_OK) throw CozkError(rc, "lookups: cannot create the verifier's context");
        HIP_TRY(hipSetDevice(h->vctx->device));
        lookups_setup_verifier(h);
        if (cfg->primary) lookups_setup_primary_verifier(h);
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

const char* cozk_lookups_error(const cozk_lookups* h) { return h ? h->error.c_str() : "null harness"; }

int cozk_lookups_destroy(cozk_lookups* h) {
    if (!h) return COZK_OK;
    for (auto& ps : h->parties) {
        if (ps.ctx) (void)hipSetDevice(ps.ctx->device);
        ps.flags.clear();
        ps.fp_a = VecH();
        ps.fp_b = VecH();
        ps.instr_flags.clear();
        ps.E.clear();
        ps.outputs = PolyH();
        if (ps.own_ctx && ps.ctx) cozk_ctx_destroy(ps.ctx);
    }
    if (h->vctx) {
        (void)hipSetDevice(h->vctx->device);
        h->v_flags.clear();
        h->v_fps.clear();
        h->v_E.clear();
        h->v_iflags.clear();
        h->v_outputs = PolyH();
        cozk_ctx_destroy(h->vctx);
    }
    delete h;
    return COZK_OK;
}

int cozk_lookups_prove(cozk_lookups* h, int verify, cozk_lookups_result* res) {
    if (!h || !res) return COZK_ERR_INVALID_ARG;
    memset(res, 0, sizeof *res);
    res->verified = -1;
    int np = h->nparties;
    InProcStar star(np);
    InProcRing ring(&star.abort);
    std::vector<std::unique_ptr<InProcStarWorker>> sw;
    std::vector<std::unique_ptr<InProcRingNet>> rn;
    for (int p = 0; p < np; p++) {
        sw.emplace_back(new InProcStarWorker(&star, p));
        rn.emplace_back(np == 3 ? new InProcRingNet(&ring, p) : nullptr);
        h->parties[p].error.clear();
    }
    std::vector<std::thread> threads;
    double t0 = now_ms();
    for (int p = 0; p < np; p++) {
        threads.emplace_back([&, p] {
            try {
                lookups_worker_main(h, h->parties[p], sw[p].get(), rn[p].get());
            } catch (const std::exception& e) {
                h->parties[p].error = e.what();
                star.abort.flag.store(true);
            }
        });
    }
    LookupsProof proof;
    std::string why;
    int verified = -1;
    int rc = COZK_OK;
    try {
        InProcStarCoordinator coord(&star);
        verified = lookups_coordinator_main(h, coord, proof, verify != 0, why);
    } catch (const std::exception& e) {
        h->error = std::string("coordinator: ") + e.what();
        star.abort.flag.store(true);
        rc = COZK_ERR_INTERNAL;
    }
    for (auto& t : threads) t.join();
    double t1 = now_ms();
    for (int p = 0; p < np; p++) {
        if (!h->parties[p].error.empty()) {
            h->error = "party " + std::to_string(p) + ": " + h->parties[p].error;
            rc = COZK_ERR_INTERNAL;
        }
    }
    if (rc != COZK_OK) return rc;
    if (verified == 0) h->error = "verification failed: " + why;
    res->verified = verified;
    res->wall_ms = t1 - t0;
    for (int p = 0; p < np; p++) {
        LookupsParty& ps = h->parties[p];
        res->t_primary_ms = std::max(res->t_primary_ms, ps.t_primary);
        res->t_construct_ms = std::max(res->t_construct_ms, ps.t_construct);
        res->t_prove_ms = std::max(res->t_prove_ms, ps.t_prove);
        res->t_worker_ms = std::max(res->t_worker_ms, ps.t_total);
        res->bytes_star_up += ps.star_up;
        res->bytes_star_down += ps.star_down;
        res->bytes_ring += ps.ring_bytes;
        res->star_messages += ps.star_msgs;
    }
    h->last_proof = proof.serialize();
    res->proof_len = h->last_proof.size();
    Sha256 s;
    s.update(h->last_proof.data(), h->last_proof.size());
    s.final(res->proof_digest);
    return COZK_OK;
}

int cozk_lookups_proof_bytes(const cozk_lookups* h, uint8_t* out, size_t cap) {
    if (!h || !out || cap < h->last_proof.size()) return COZK_ERR_INVALID_ARG;
    memcpy(out, h->last_proof.data(), h->last_proof.size());
    return COZK_OK;
}

}  // extern "C"
