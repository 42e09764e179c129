// In-process harness of co-jolt's Spartan outer sumcheck: Az / Bz / Cz from a synthetic, SATISFIED
// constraint system over 14 witness columns and the cubic sumcheck of co-jolt/src/r1cs/spartan/worker.rs:63-100,277-300;
// workers on the GPU(s), coordinator + plain verifier on the calling thread.  The system (the concrete Jolt constraints
// live out of tree in jolt-core) exercises every case of Rep3SpartanInterleavedPolynomial::new
// (co-jolt/src/poly/spartan_interleaved_poly.rs:40-172): shared x shared, public x shared, public x public, an LC that is
// often zero, an empty LC, and two cross-step constraints (one with constant terms on both sides):
//   columns  0..3  s0..s3   shared   stream(seed + 100 (v + 1))
//            4     p0       public   8-bit values   small(seed + 501)
//            5     p1       public   0/1            small(seed + 502)
//            6     prod0    shared   (s0 + 2 s1 + 3)(s2 - s3)
//            7     prod1    shared   (p0 + 1)(s1 + s2)
//            8     pp       public   p0 (p1 - 1)
//            9     prod3    shared   p1 s0
//            10    pc       shared   pc[x] = next_pc[x - 1], pc[0] = 7
//            11    next_pc  shared   stream(seed + 600)
//            12    flagc    public   0/1, small(seed + 503), 0 at the last step
//            13    q        shared   q[x] = s0[x - 1], q[0] = 9
//   uniform  0: (s0 + 2 s1 + 3) (s2 - s3) = prod0     1: (p0 + 1) (s1 + s2) = prod1     2: p0 (p1 - 1) = pp
//            3: p1 s0 = prod3                          4: (empty) s1 = (empty)
//   cross    0: flagc (next_pc - pc') = 0              1: flagc ((s0 + 5) - (q' + 5)) = 0            (' = next step)
// 8 rows per step.  oracle/pyspartan_outer.py restates the reference's SPARSE algorithm on the same instance.
#pragma once
// cfg.system = 1 replaces the toy system by the reference's own constraint SET (co-jolt/src/r1cs/constraints.rs:39-257 over the 78
// inputs of r1cs/inputs.rs: 70 uniform + 2 cross-step constraints, 128 rows per step) on a synthetic trace that satisfies it
// (csrc/host/jolt_r1cs.hpp); cfg.full = 1 runs the WHOLE Spartan worker (outer + inner + shift sumchecks and the two opening
// appends, r1cs/spartan/worker.rs:63-273; csrc/host/spartan_jolt.hpp) instead of the outer sumcheck alone.
#include "jolt_r1cs.hpp"
#include "spartan_jolt.hpp"

struct OuterParty {
    cozk_ctx* ctx = nullptr;
    bool own_ctx = false;
    int party = 0;
    std::vector<PolyH> cols;
    double t_build = 0, t_prove = 0, t_total = 0;
    SpartanTimes times;
    uint64_t star_up = 0, star_down = 0, star_msgs = 0;
    std::string error;
};

typedef jolt::System OuterSystem;

struct cozk_outer_harness {
    cozk_outer_config cfg;
    int nparties = 1;
    size_t N = 0;
    int ncols = 14;
    std::vector<OuterParty> parties;
    OuterSystem sys;
    std::vector<std::vector<fe>> clear;  // the dealer's view of the columns
    std::vector<int> is_public;
    // the verifier's own view (cfg.full): the clear columns as plain polynomials on its own context, made at the first verify
    cozk_ctx* vctx = nullptr;
    std::vector<PolyH> v_cols;
    std::string error;
    Bytes last_proof;
};

namespace {

void outer_build_system(OuterSystem& s) {
    using jolt::LC;
    const int C = -1;  // the constant "variable"
    auto L = [](std::initializer_list<std::pair<int, int64_t>> t) { return LC(t); };
    // uniform constraints: a, b, c
    s.add_uniform(L({{0, 1}, {1, 2}, {C, 3}}), L({{2, 1}, {3, -1}}), L({{6, 1}}));
    s.add_uniform(L({{4, 1}, {C, 1}}), L({{1, 1}, {2, 1}}), L({{7, 1}}));
    s.add_uniform(L({{4, 1}}), L({{5, 1}, {C, -1}}), L({{8, 1}}));
    s.add_uniform(L({{5, 1}}), L({{0, 1}}), L({{9, 1}}));
    s.add_uniform(L({}), L({{1, 1}}), L({}));
    // cross-step constraints: a, b, cond
    s.add_cross(L({{11, 1}}), 0, L({{10, 1}}), 1, L({{12, 1}}), 0);
    s.add_cross(L({{0, 1}, {C, 5}}), 0, L({{13, 1}, {C, 5}}), 1, L({{12, 1}}), 0);
    s.finish(14);
}

void outer_build_clear(cozk_outer_harness* h) {
    const uint64_t seed = h->cfg.seed;
    const size_t N = h->N;
    auto& c = h->clear;
    if (h->cfg.system == 1) {
        jolt::build_clear(seed, N, c);
        return;
    }
    c.assign(14, std::vector<fe>(N));
    const fe one = Fr::one();
    for (size_t x = 0; x < N; x++) {
        for (int v = 0; v < 4; v++) c[v][x] = synthetic_fr_host(seed + 100ull * (uint64_t)(v + 1), x);
        c[4][x] = Fr::from_u64(synthetic_small_host(seed + 501ull, x, 8));
        c[5][x] = Fr::from_u64(synthetic_small_host(seed + 502ull, x, 1));
        c[12][x] = x + 1 == N ? Fr::zero() : Fr::from_u64(synthetic_small_host(seed + 503ull, x, 1));
        c[11][x] = synthetic_fr_host(seed + 600ull, x);
    }
    for (size_t x = 0; x < N; x++) {
        c[6][x] = Fr::mul(Fr::add(Fr::add(c[0][x], Fr::dbl(c[1][x])), Fr::from_u64(3)), Fr::sub(c[2][x], c[3][x]));
        c[7][x] = Fr::mul(Fr::add(c[4][x], one), Fr::add(c[1][x], c[2][x]));
        c[8][x] = Fr::mul(c[4][x], Fr::sub(c[5][x], one));
        c[9][x] = Fr::mul(c[5][x], c[0][x]);
        c[10][x] = x == 0 ? Fr::from_u64(7) : c[11][x - 1];
        c[13][x] = x == 0 ? Fr::from_u64(9) : c[0][x - 1];
    }
}

void outer_setup_party(cozk_outer_harness* h, OuterParty& ps) {
    cozk_ctx* ctx = ps.ctx;
    for (int v = 0; v < h->ncols; v++) {
        cozk_vec* pv = nullptr;
        rc_check(cozk_vec_upload(ctx, h->clear[v].data(), h->N, COZK_SCALAR_FR, &pv), ctx, "vec_upload(column)");
        VecH plain(pv);
        cozk_poly* p = nullptr;
        if (h->is_public[v] || h->cfg.mode == COZK_MODE_PLAIN) {
            rc_check(cozk_poly_create(ctx, COZK_MODE_PLAIN, plain.h, nullptr, &p), ctx, "poly_create");
        } else {
            uint8_t k0[COZK_PRF_KEY_BYTES], k1[COZK_PRF_KEY_BYTES];
            harness_prf_key(h->cfg.seed + 100ull * (uint64_t)(v + 1), 101, k0);
            harness_prf_key(h->cfg.seed + 100ull * (uint64_t)(v + 1), 102, k1);
            cozk_vec *sa = nullptr, *sb = nullptr;
            rc_check(cozk_rep3_share_vec(ctx, plain.h, k0, k1, 0, ps.party, &sa, &sb), ctx, "rep3_share_vec");
            VecH a(sa), b(sb);
            rc_check(cozk_poly_create(ctx, COZK_MODE_REP3, a.h, b.h, &p), ctx, "poly_create");
        }
        ps.cols.push_back(PolyH(p));
    }
}

struct OuterProofBundle {
    bool full = false;
    OuterSumcheckProof outer;
    JoltSpartanProof spartan;
    Bytes serialize() const {
        Writer w;
        if (full) {
            spartan.write(w);
            return w.b;
        }
        w.u64(outer.compressed_polys.size());
        for (auto& p : outer.compressed_polys) w.vec_fr(p);
        w.vec_fr(outer.claims);
        return w.b;
    }
};

void outer_worker_main(cozk_outer_harness* h, OuterParty& ps, StarNetWorker* star) {
    WorkerEnv env;
    env.ctx = ps.ctx;
    env.mode = h->cfg.mode;
    env.party = ps.party;
    env.star = star;
    env.ring = nullptr;  // Az (x) Bz is summed, never reshared: no ring on this path
    HIP_TRY(hipSetDevice(ps.ctx->device));
    double t0 = now_ms();
    if (h->cfg.full) {
        std::vector<cozk_poly*> cols;
        for (auto& c : ps.cols) cols.push_back(c.h);
        Rep3ProverOpeningAccumulator acc;  // the two openings stay here (a whole Jolt proof reduces them with all the others)
        ps.times = SpartanTimes();
        prove_spartan_worker(env, h->sys, cols, h->N, acc, &ps.times);
        double t2 = now_ms();
        ps.t_build = ps.times.t_build;
        ps.t_prove = t2 - t0 - ps.times.t_build;
        ps.t_total = t2 - t0;
    } else {
        Bytes req = env.star->receive_request();
        Reader rd(req);
        std::vector<fe> tau = rd.vec_fr();
        std::vector<uint64_t> w = to_abi(tau);
        std::vector<const cozk_poly*> cols;
        for (auto& c : ps.cols) cols.push_back(c.h);
        cozk_outer* st = nullptr;
        rc_check(cozk_outer_create(env.ctx, env.mode, ps.party, &h->sys.desc, cols.data(), cols.size(), w.data(), tau.size(), &st), env.ctx, "outer_create");
        OuterH sth(st);
        double t1 = now_ms();
        ps.t_build = t1 - t0;
        (void)prove_spartan_cubic_sumcheck_worker(env, st, (int)tau.size());
        double t2 = now_ms();
        ps.t_prove = t2 - t1;
        ps.t_total = t2 - t0;
    }
    ps.star_up = star->bytes_up;
    ps.star_down = star->bytes_down;
    ps.star_msgs = star->n_msgs;
}

// the dealer's Az, Bz, Cz in the clear, row by row, and their multilinear extensions at the big-endian point pt
void outer_clear_claims(cozk_outer_harness* h, const std::vector<fe>& pt, fe out[3]) {
    const size_t N = h->N, P = h->sys.padded;
    const OuterSystem& s = h->sys;
    std::vector<fe> eq = eq_evals_host(pt);
    auto lc_eval = [&](const cozk_lc& lc, size_t step) {
        fe acc = Fr::zero();
        size_t row = step;
        bool const_only = false;
        if (lc.offset) {
            if (step + 1 < N) row = step + 1;
            else const_only = true;
        }
        for (int t = 0; t < lc.n_terms; t++) {
            int v = s.term_var[lc.first_term + t];
            int64_t c = s.term_coeff[lc.first_term + t];
            fe cf = Fr::from_u64((uint64_t)(c < 0 ? -c : c));
            if (c < 0) cf = Fr::neg(cf);
            if (v < 0) acc = Fr::add(acc, cf);
            else if (!const_only) acc = Fr::add(acc, Fr::mul(h->clear[v][row], cf));
        }
        return acc;
    };
    out[0] = out[1] = out[2] = Fr::zero();
    for (size_t step = 0; step < N; step++) {
        for (size_t ci = 0; ci < s.desc.n_uniform + s.desc.n_cross; ci++) {
            fe az, bz, cz;
            if (ci < s.desc.n_uniform) {
                az = lc_eval(s.uniform[3 * ci], step);
                bz = lc_eval(s.uniform[3 * ci + 1], step);
                cz = lc_eval(s.uniform[3 * ci + 2], step);
            } else {
                size_t j = ci - s.desc.n_uniform;
                az = Fr::sub(lc_eval(s.cross[3 * j], step), lc_eval(s.cross[3 * j + 1], step));
                bz = lc_eval(s.cross[3 * j + 2], step);
                cz = Fr::zero();
            }
            const fe& e = eq[step * P + ci];
            out[0] = Fr::add(out[0], Fr::mul(e, az));
            out[1] = Fr::add(out[1], Fr::mul(e, bz));
            out[2] = Fr::add(out[2], Fr::mul(e, cz));
        }
    }
}

// the verifier's columns: the dealer's clear values as plain polynomials on a context of their own
void outer_setup_verifier(cozk_outer_harness* h) {
    if (h->vctx) return;
    int rc = cozk_ctx_create(h->cfg.devices[0], &h->vctx);
    if (rc != COZK_OK) throw CozkError(rc, "outer harness: cannot create the verifier's context");
    HIP_TRY(hipSetDevice(h->vctx->device));
    for (int v = 0; v < h->ncols; v++) {
        cozk_vec* pv = nullptr;
        rc_check(cozk_vec_upload(h->vctx, h->clear[v].data(), h->N, COZK_SCALAR_FR, &pv), h->vctx, "vec_upload(column)");
        VecH plain(pv);
        cozk_poly* p = nullptr;
        rc_check(cozk_poly_create(h->vctx, COZK_MODE_PLAIN, plain.h, nullptr, &p), h->vctx, "poly_create");
        h->v_cols.push_back(PolyH(p));
    }
}

// claims[i] == column_i(point) for every column (stand-in for the PCS opening of the flattened witness)
bool outer_check_openings(cozk_outer_harness* h, const std::vector<fe>& point, const std::vector<fe>& claims) {
    HIP_TRY(hipSetDevice(h->vctx->device));
    std::vector<uint64_t> w = to_abi(point);
    cozk_vec* chi = nullptr;
    rc_check(cozk_eq_evals(h->vctx, w.data(), (int)point.size(), &chi), h->vctx, "eq_evals");
    VecH chih(chi);
    std::vector<const cozk_poly*> ps;
    for (auto& c : h->v_cols) ps.push_back(c.h);
    std::vector<uint64_t> out(4 * ps.size());
    rc_check(cozk_poly_batch_evaluate_at_chi(h->vctx, ps.data(), ps.size(), chih.h, out.data()), h->vctx, "batch_evaluate");
    if (claims.size() != ps.size()) return false;
    for (size_t i = 0; i < ps.size(); i++)
        if (!Fr::eq(fe_from_u64x4(out.data() + 4 * i), claims[i])) return false;
    return true;
}

int outer_coordinator_main(cozk_outer_harness* h, StarNetCoordinator& net, OuterProofBundle& proof, bool verify, std::string& why) {
    int constr_bits = 0;
    while (((size_t)1 << constr_bits) < h->sys.padded) constr_bits++;
    if (h->cfg.full) {
        proof.full = true;
        Transcript tr("cozk-spartan");
        proof.spartan = coordinate_spartan(net, tr, h->sys, h->N);
        if (!verify) return -1;
        Transcript vt("cozk-spartan");
        std::vector<fe> rx_step, shift_r;
        fe rho[2];
        if (!verify_spartan(proof.spartan, h->sys, h->N, vt, rx_step, shift_r, rho, why)) return 0;
        outer_setup_verifier(h);
        if (!outer_check_openings(h, rx_step, proof.spartan.witness_evals)) {
            why = "spartan: claimed_witness_evals != the columns at rx_step";
            return 0;
        }
        if (!outer_check_openings(h, shift_r, proof.spartan.shift_witness_evals)) {
            why = "spartan: shift_sumcheck_witness_evals != the columns at the shift point";
            return 0;
        }
        return 1;
    }
    Transcript tr("cozk-spartan-outer");
    int num_rounds = h->cfg.log_steps + constr_bits;  // log2(steps * padded rows per step)
    std::vector<fe> tau = tr.challenge_vector((size_t)num_rounds);
    {
        Writer w;
        w.vec_fr(tau);
        net.broadcast_request(w.b);
    }
    std::vector<fe> r;
    proof.outer = coordinate_outer_sumcheck(net, tr, num_rounds, r);
    if (!verify) return -1;
    Transcript vt("cozk-spartan-outer");
    std::vector<fe> vtau = vt.challenge_vector((size_t)num_rounds), rv;
    if (!verify_outer_sumcheck(proof.outer, vtau, vt, rv)) {
        why = "outer sumcheck: a round or the final claim does not hold";
        return 0;
    }
    if ((h->N << constr_bits) > ((size_t)1 << 22)) return 1;  // the row-by-row host evaluation below is for test sizes
    std::vector<fe> pt(rv.rbegin(), rv.rend());
    fe direct[3];
    outer_clear_claims(h, pt, direct);
    for (int i = 0; i < 3; i++)
        if (!Fr::eq(direct[i], proof.outer.claims[i])) {
            why = std::string("outer sumcheck: claim ") + "ABC"[i] + "z(r) != its multilinear extension";
            return 0;
        }
    return 1;
}

}  // namespace

extern "C" {

int cozk_outer_harness_create(const cozk_outer_config* cfg, cozk_outer_harness** out) {
    if (!cfg || !out) return COZK_ERR_INVALID_ARG;
    *out = nullptr;
    cozk_outer_harness* h = new cozk_outer_harness();
    h->cfg = *cfg;
    try {
        COZK_REQUIRE(cfg->mode == COZK_MODE_PLAIN || cfg->mode == COZK_MODE_REP3, "outer harness: mode");
        COZK_REQUIRE(cfg->log_steps >= 0 && cfg->log_steps <= 22, "outer harness: log_steps in 0..22");
        COZK_REQUIRE((cfg->system == 0 || cfg->system == 1) && (cfg->full == 0 || cfg->full == 1), "outer harness: system / full are 0 or 1");
        h->nparties = cfg->mode == COZK_MODE_REP3 ? 3 : 1;
        h->N = (size_t)1 << cfg->log_steps;
        if (cfg->system == 1) {
            jolt::build_system(h->sys);
            h->ncols = jolt::NUM_INPUTS;
            for (int v = 0; v < h->ncols; v++) h->is_public.push_back(jolt::public_bytes(v) ? 1 : 0);
        } else {
            outer_build_system(h->sys);
            h->ncols = 14;
            h->is_public = {0, 0, 0, 0, 1, 1, 0, 0, 1, 0, 0, 0, 1, 0};
        }
        outer_build_clear(h);
        h->parties.resize(h->nparties);
        for (int p = 0; p < h->nparties; p++) {
            OuterParty& ps = h->parties[p];
            ps.party = p;
            int rc = cozk_ctx_create(cfg->devices[p], &ps.ctx);
            if (rc != COZK_OK) throw CozkError(rc, "outer harness: cannot create a context (no HIP device?)");
            ps.own_ctx = true;
            HIP_TRY(hipSetDevice(ps.ctx->device));
            outer_setup_party(h, ps);
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

const char* cozk_outer_harness_error(const cozk_outer_harness* h) { return h ? h->error.c_str() : "null harness"; }

int cozk_outer_harness_destroy(cozk_outer_harness* h) {
    if (!h) return COZK_OK;
    for (auto& ps : h->parties) {
        if (ps.ctx) (void)hipSetDevice(ps.ctx->device);
        ps.cols.clear();
        if (ps.own_ctx && ps.ctx) cozk_ctx_destroy(ps.ctx);
    }
    if (h->vctx) {
        (void)hipSetDevice(h->vctx->device);
        h->v_cols.clear();
        cozk_ctx_destroy(h->vctx);
    }
    delete h;
    return COZK_OK;
}

int cozk_outer_harness_prove(cozk_outer_harness* h, int verify, cozk_outer_result* res) {
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
                outer_worker_main(h, h->parties[p], sw[p].get());
            } catch (const std::exception& e) {
                h->parties[p].error = e.what();
                star.abort.flag.store(true);
            }
        });
    }
    OuterProofBundle proof;
    std::string why;
    int verified = -1;
    int rc = COZK_OK;
    try {
        InProcStarCoordinator coord(&star);
        verified = outer_coordinator_main(h, coord, proof, verify != 0, why);
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
        OuterParty& ps = h->parties[p];
        res->t_build_ms = std::max(res->t_build_ms, ps.t_build);
        res->t_prove_ms = std::max(res->t_prove_ms, ps.t_prove);
        res->t_worker_ms = std::max(res->t_worker_ms, ps.t_total);
        res->t_outer_ms = std::max(res->t_outer_ms, ps.times.t_outer);
        res->t_inner_ms = std::max(res->t_inner_ms, ps.times.t_inner);
        res->t_shift_ms = std::max(res->t_shift_ms, ps.times.t_shift);
        res->t_openings_ms = std::max(res->t_openings_ms, ps.times.t_openings);
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

int cozk_outer_harness_proof_bytes(const cozk_outer_harness* h, uint8_t* out, size_t cap) {
    if (!h || !out || cap < h->last_proof.size()) return COZK_ERR_INVALID_ARG;
    memcpy(out, h->last_proof.data(), h->last_proof.size());
    return COZK_OK;
}

}  // extern "C"
