// The WHOLE co-jolt Spartan worker: Rep3UniformSpartanProver::prove (co-jolt/src/r1cs/spartan/worker.rs:63-273)
// with its coordinator (r1cs/spartan/coordinator.rs:27-136) and the plain verifier's sumcheck checks:
//   outer cubic sumcheck over Az / Bz / Cz  (prover.hpp prove_spartan_cubic_sumcheck_worker, csrc/spartan_outer.inc)
//   inner sumcheck over y = (shift bit | constant bit | variable): MixedPolynomials of 4 * num_vars_padded entries
//       (co-jolt/src/poly/mixed_polynomial.rs:12-193; host -- 512 entries), bind_z / bind_shift_z by ONE pass of dot products
//       over the flattened witness (cozk_poly_batch_dot_public), poly_ABC from the constraint table
//   shift sumcheck over the steps: prove_arbitrary_worker over (sum_i eq_ry[i] poly_i, eq_plus_one(rx_step, .))
//   the two batch_evaluate + opening_accumulator.append (worker.rs:243-272)
// Out of tree (jolt-core; restated, parity unpinned): EqPlusOnePolynomial, UniformSpartanKey::evaluate_matrix_mle_partial,
// UniformSpartanProof::verify -- fixed by the identity the in-tree worker relies on (see oracle/pyspartan_outer.py prove_full).
#pragma once

namespace cozk {

// ---------------------------------------------------------------- SharedOrPublic (co-jolt/src/utils/shared_or_public.rs:16-290)
// the two kinds a MixedPolynomial of the inner sumcheck holds: Public(F) and Shared(Rep3 share); the plain prover's values are
// Shared with b = 0 (into_additive = the value)
struct SoP {
    bool shared = false;
    fe a = Fr::zero(), b = Fr::zero();
};
struct SoPOps {
    int mode, party;
    SoP pub(const fe& v) const { return SoP{false, v, Fr::zero()}; }
    // rep3::arithmetic::add_public: party 0's a, party 1's b
    SoP add_public(SoP s, const fe& c) const {
        if (mode == COZK_MODE_PLAIN || party == 0) s.a = Fr::add(s.a, c);
        else if (party == 1) s.b = Fr::add(s.b, c);
        return s;
    }
    SoP neg(SoP x) const {
        x.a = Fr::neg(x.a);
        x.b = Fr::neg(x.b);
        return x;
    }
    SoP add(const SoP& x, const SoP& y) const {
        if (x.shared && y.shared) return SoP{true, Fr::add(x.a, y.a), Fr::add(x.b, y.b)};
        if (x.shared) return add_public(x, y.a);
        if (y.shared) return add_public(y, x.a);
        return pub(Fr::add(x.a, y.a));
    }
    SoP sub(const SoP& x, const SoP& y) const { return add(x, neg(y)); }  // sub_shared_by_public / sub_public_by_shared
    SoP mul_public(SoP x, const fe& c) const {
        x.a = Fr::mul(x.a, c);
        if (x.shared) x.b = Fr::mul(x.b, c);
        return x;
    }
    // (x * y).into_additive(party) for a public x (comb_func of worker.rs:162-165 with poly_ABC public)
    fe mul_public_into_additive(const fe& x, const SoP& y) const {
        if (!y.shared) return (mode == COZK_MODE_PLAIN || party == 0) ? Fr::mul(x, y.a) : Fr::zero();
        if (mode == COZK_MODE_PLAIN) return Fr::mul(x, y.a);
        return Fr::mul(Fr::mul(Fr::add(y.a, y.b), x), fr_two_inv());
    }
    fe into_additive(const SoP& y) const { return mul_public_into_additive(Fr::one(), y); }
};

// EqPlusOnePolynomial::evaluate at points (x, y), big-endian
static inline fe eq_plus_one_point(const std::vector<fe>& x, const std::vector<fe>& y) {
    const int l = (int)x.size();
    const fe one = Fr::one();
    fe acc = Fr::zero();
    for (int k = 0; k < l; k++) {
        fe v = Fr::mul(Fr::sub(one, x[l - 1 - k]), y[l - 1 - k]);
        for (int j = 0; j < k; j++) v = Fr::mul(v, Fr::mul(x[l - 1 - j], Fr::sub(one, y[l - 1 - j])));
        for (int j = 0; j < l - k - 1; j++) v = Fr::mul(v, Fr::add(Fr::sub(Fr::sub(one, x[j]), y[j]), Fr::dbl(Fr::mul(x[j], y[j]))));
        acc = Fr::add(acc, v);
    }
    return acc;
}

static inline size_t spartan_vars_padded(const jolt::System& s) {
    size_t V = 1;
    while (V < s.num_vars) V <<= 1;
    return V;
}

// key.evaluate_matrix_mle_partial(rx_constr, rx_step, rlc) (used worker.rs:123-126): ABC(rx_constr, .) = A + rlc B + rlc^2 C as
// 4 V entries [variables | constant at V | shifted variables | unused].  An offset LC of a cross-step constraint reads the next
// step: its variables land in the shifted half; its constant stays in the constant column (sum_t eq(rx_step, t) = 1, and at the
// last step an offset LC IS its constant, spartan_interleaved_poly.rs:666-684, while eq_plus_one has no wrap-around).
static inline std::vector<fe> spartan_matrix_mle_partial(const jolt::System& s, const std::vector<fe>& rx_constr, const fe& rlc) {
    const size_t V = spartan_vars_padded(s);
    std::vector<fe> eq = eq_evals_host(rx_constr), out(4 * V, Fr::zero());
    const fe one = Fr::one(), r2 = Fr::mul(rlc, rlc);
    auto add = [&](const jolt::LC& lc, size_t row, const fe& w, bool shifted, bool negate) {
        for (auto& t : lc) {
            const size_t col = t.first < 0 ? V : (size_t)t.first + (shifted ? 2 * V : 0);
            fe v = Fr::mul(Fr::mul(jolt::fr_i64(t.second), w), eq[row]);
            out[col] = negate ? Fr::sub(out[col], v) : Fr::add(out[col], v);
        }
    };
    for (size_t ci = 0; ci < s.u_rows.size(); ci++) {
        add(s.u_rows[ci].a, ci, one, false, false);
        add(s.u_rows[ci].b, ci, rlc, false, false);
        add(s.u_rows[ci].c, ci, r2, false, false);
    }
    for (size_t ci = 0; ci < s.x_rows.size(); ci++) {
        const size_t row = s.u_rows.size() + ci;
        const auto& x = s.x_rows[ci];  // Az = a - b, Bz = cond (field c), Cz = 0
        add(x.a, row, one, x.off_a != 0, false);
        add(x.b, row, one, x.off_b != 0, true);
        add(x.c, row, rlc, x.off_c != 0, false);
    }
    return out;
}

struct JoltSpartanProof {
    OuterSumcheckProof outer;
    SumcheckProof inner, shift;
    fe shift_claim;
    std::vector<fe> witness_evals, shift_witness_evals;
    void write(Writer& w) const {
        w.u64(outer.compressed_polys.size());
        for (auto& p : outer.compressed_polys) w.vec_fr(p);
        w.vec_fr(outer.claims);
        w.u64(inner.compressed_polys.size());
        for (auto& p : inner.compressed_polys) w.vec_fr(p);
        w.fr(shift_claim);
        w.u64(shift.compressed_polys.size());
        for (auto& p : shift.compressed_polys) w.vec_fr(p);
        w.vec_fr(witness_evals);
        w.vec_fr(shift_witness_evals);
    }
};

struct SpartanTimes {
    double t_build = 0, t_outer = 0, t_inner = 0, t_shift = 0, t_openings = 0;
};

static inline double spartan_now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Rep3MultilinearPolynomial::batch_evaluate + into_additive per claim (worker.rs:243-272): shared polynomials give additive
// shares, public ones a value held by party 0
static inline std::vector<fe> spartan_batch_evaluate(WorkerEnv& env, const std::vector<cozk_poly*>& polys, const cozk_vec* chi) {
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
        rc_check(cozk_poly_batch_evaluate_at_chi(env.ctx, shp.data(), shp.size(), chi, out.data()), env.ctx, "batch_evaluate");
        for (size_t k = 0; k < shp.size(); k++) claims[shi[k]] = fe_from_u64x4(out.data() + 4 * k);
    }
    if (!pbp.empty()) {
        std::vector<uint64_t> out(4 * pbp.size());
        rc_check(cozk_poly_batch_evaluate_at_chi(env.ctx, pbp.data(), pbp.size(), chi, out.data()), env.ctx, "batch_evaluate(public)");
        for (size_t k = 0; k < pbp.size(); k++) claims[pbi[k]] = env.additive_trivial(fe_from_u64x4(out.data() + 4 * k));
    }
    return claims;
}

// Rep3UniformSpartanProver::prove (worker.rs:63-273).  `cols` = I::flatten() (REP3 polynomials are shared columns, PLAIN ones
// public); the two openings land in `acc`.
static inline void prove_spartan_worker(WorkerEnv& env, const jolt::System& sys, const std::vector<cozk_poly*>& cols, size_t num_steps,
                                        Rep3ProverOpeningAccumulator& acc, SpartanTimes* times = nullptr) {
    const size_t V = spartan_vars_padded(sys);
    int steps_bits = 0;
    while (((size_t)1 << steps_bits) < num_steps) steps_bits++;
    int constr_bits = 0;
    while (((size_t)1 << constr_bits) < sys.padded) constr_bits++;
    double t0 = spartan_now_ms();
    // ---- Sumcheck 1: outer
    std::vector<fe> tau;
    {
        Bytes req = env.star->receive_request();
        Reader rd(req);
        tau = rd.vec_fr();
    }
    std::vector<fe> outer_r;
    double t1;
    {
        std::vector<uint64_t> w = to_abi(tau);
        std::vector<const cozk_poly*> cc(cols.begin(), cols.end());
        cozk_outer* st = nullptr;
        rc_check(cozk_outer_create(env.ctx, env.mode, env.party, &sys.desc, cc.data(), cc.size(), w.data(), tau.size(), &st), env.ctx, "outer_create");
        OuterH sth(st);
        t1 = spartan_now_ms();
        std::vector<fe> rs = prove_spartan_cubic_sumcheck_worker(env, st, (int)tau.size());
        outer_r.assign(rs.rbegin(), rs.rend());
    }
    double t2 = spartan_now_ms();
    // ---- Sumcheck 2: inner
    fe rlc, claim_inner_joint;
    {
        Bytes req = env.star->receive_request();
        Reader rd(req);
        rlc = rd.fr();
        claim_inner_joint = rd.fr();
    }
    std::vector<fe> rx_step(outer_r.begin(), outer_r.begin() + steps_bits), rx_constr(outer_r.begin() + steps_bits, outer_r.end());
    std::vector<uint64_t> rxs = to_abi(rx_step);
    cozk_vec *eqv = nullptr, *eqp1v = nullptr;
    rc_check(cozk_eq_evals(env.ctx, rxs.data(), steps_bits, &eqv), env.ctx, "eq_evals(rx_step)");
    VecH eq_step(eqv);
    rc_check(cozk_eq_plus_one_evals(env.ctx, rxs.data(), steps_bits, &eqp1v), env.ctx, "eq_plus_one_evals(rx_step)");
    VecH eqp1_step(eqp1v);
    std::vector<fe> abc = spartan_matrix_mle_partial(sys, rx_constr, rlc);
    const SoPOps ops{env.mode, env.party};
    std::vector<SoP> z(4 * V);  // bind_z (2 V) then bind_shift_z (2 V); the entries past the inputs stay zero_public
    {
        const cozk_vec* pubs[2] = {eq_step.h, eqp1_step.h};
        std::vector<const cozk_poly*> cc(cols.begin(), cols.end());
        std::vector<uint64_t> dots(cc.size() * 2 * 8);
        rc_check(cozk_poly_batch_dot_public(env.ctx, cc.data(), cc.size(), pubs, 2, dots.data()), env.ctx, "batch_dot_public");
        for (size_t i = 0; i < cc.size(); i++) {
            const bool shared = cozk_poly_mode(cc[i]) == COZK_MODE_REP3 || env.mode == COZK_MODE_PLAIN;
            for (int q = 0; q < 2; q++) {
                const uint64_t* d = dots.data() + (i * 2 + q) * 8;
                z[(size_t)q * 2 * V + i] = SoP{shared, fe_from_u64x4(d), fe_from_u64x4(d + 4)};
            }
        }
        z[V] = ops.pub(Fr::one());  // bind_z[num_vars_uniform] = 1 (worker.rs:154)
    }
    std::vector<fe> inner_r;
    {
        fe previous_claim = env.additive_trivial(claim_inner_joint);
        int rounds = 0;
        while (((size_t)1 << rounds) < 4 * V) rounds++;
        for (int round = 0; round < rounds; round++) {
            const size_t half = abc.size() / 2;
            fe e0 = Fr::zero(), e2 = Fr::zero();
            for (size_t i = 0; i < half; i++) {  // sumcheck_evals(i, 2, HighToLow) of both polynomials, comb_func, sum
                e0 = Fr::add(e0, ops.mul_public_into_additive(abc[i], z[i]));
                fe a2 = Fr::sub(Fr::dbl(abc[i + half]), abc[i]);
                SoP m = ops.sub(z[i + half], z[i]);
                SoP z2 = ops.add(z[i + half], m);
                e2 = Fr::add(e2, ops.mul_public_into_additive(a2, z2));
            }
            fe pts[3] = {e0, Fr::sub(previous_claim, e0), e2};
            std::vector<fe> cf(3);
            unipoly_from_evals(pts, 3, cf.data());
            Writer w;
            w.vec_fr(cf);
            env.star->send_response(w.b);
            Bytes req = env.star->receive_request();
            Reader rd(req);
            fe r_j = rd.fr();
            previous_claim = env.additive_trivial(rd.fr());
            inner_r.push_back(r_j);
            for (size_t i = 0; i < half; i++) {  // bound_poly_var_top (mixed_polynomial.rs:78-89)
                abc[i] = Fr::add(abc[i], Fr::mul(Fr::sub(abc[i + half], abc[i]), r_j));
                z[i] = ops.add(z[i], ops.mul_public(ops.sub(z[i + half], z[i]), r_j));
            }
            abc.resize(half);
            z.resize(half);
        }
    }
    double t3 = spartan_now_ms();
    // ---- Sumcheck 3: shift
    std::vector<fe> ry_var(inner_r.begin() + 1, inner_r.end());
    std::vector<fe> eq_ry = eq_evals_host(ry_var);
    std::vector<fe> shift_r;
    {
        std::vector<fe> cf(eq_ry.begin(), eq_ry.begin() + cols.size());
        std::vector<uint64_t> cfa = to_abi(cf);
        std::vector<const cozk_poly*> cc(cols.begin(), cols.end());
        cozk_poly* zry = nullptr;  // bind_z_ry_var: scale_coeff + sum_for per step (worker.rs:196-205)
        rc_check(cozk_poly_linear_combination(env.ctx, cc.data(), cfa.data(), cc.size(), env.mode, env.party, &zry), env.ctx, "bind_z_ry_var");
        PolyH zryh(zry);
        cozk_poly* ep = nullptr;
        rc_check(cozk_poly_create(env.ctx, COZK_MODE_PLAIN, eqp1_step.h, nullptr, &ep), env.ctx, "poly_create(eq_plus_one)");
        PolyH eph(ep);
        uint64_t cl[4];
        const cozk_poly* one_poly[1] = {zryh.h};
        rc_check(cozk_poly_batch_evaluate_at_chi(env.ctx, one_poly, 1, eqp1_step.h, cl), env.ctx, "shift_sumcheck_claim");
        fe shift_claim = fe_from_u64x4(cl);
        Writer w;
        w.fr(shift_claim);
        env.star->send_response(w.b);
        std::vector<cozk_poly*> sp = {zryh.h, eph.h};
        ArbitraryResult ar = prove_arbitrary_worker(env, shift_claim, steps_bits, sp, 2);
        shift_r = ar.r;
    }
    double t4 = spartan_now_ms();
    // ---- the two openings of the flattened witness (worker.rs:243-272)
    {
        std::vector<fe> claims = spartan_batch_evaluate(env, cols, eq_step.h);
        acc.append(env, cols, eq_step.h, rx_step, claims);
        std::vector<uint64_t> sr = to_abi(shift_r);
        cozk_vec* chi2 = nullptr;
        rc_check(cozk_eq_evals(env.ctx, sr.data(), (int)shift_r.size(), &chi2), env.ctx, "eq_evals(shift_r)");
        VecH chi2h(chi2);
        claims = spartan_batch_evaluate(env, cols, chi2h.h);
        acc.append(env, cols, chi2h.h, shift_r, claims);
    }
    double t5 = spartan_now_ms();
    if (times) {
        times->t_build += t1 - t0;
        times->t_outer += t2 - t1;
        times->t_inner += t3 - t2;
        times->t_shift += t4 - t3;
        times->t_openings += t5 - t4;
    }
}

// Rep3UniformSpartanCoordinator::prove_rep3 (r1cs/spartan/coordinator.rs:27-136)
static inline JoltSpartanProof coordinate_spartan(StarNetCoordinator& net, Transcript& tr, const jolt::System& sys, size_t num_steps) {
    JoltSpartanProof proof;
    const size_t V = spartan_vars_padded(sys);
    int steps_bits = 0;
    while (((size_t)1 << steps_bits) < num_steps) steps_bits++;
    int constr_bits = 0;
    while (((size_t)1 << constr_bits) < sys.padded) constr_bits++;
    std::vector<fe> tau = tr.challenge_vector((size_t)(steps_bits + constr_bits));
    {
        Writer w;
        w.vec_fr(tau);
        net.broadcast_request(w.b);
    }
    std::vector<fe> r;
    proof.outer = coordinate_outer_sumcheck(net, tr, steps_bits + constr_bits, r);
    fe rlc = tr.challenge_scalar();
    fe claim_inner = Fr::add(proof.outer.claims[0], Fr::add(Fr::mul(rlc, proof.outer.claims[1]), Fr::mul(Fr::mul(rlc, rlc), proof.outer.claims[2])));
    {
        Writer w;
        w.fr(rlc);
        w.fr(claim_inner);
        net.broadcast_request(w.b);
    }
    int inner_rounds = 0;
    while (((size_t)1 << inner_rounds) < 4 * V) inner_rounds++;
    (void)coordinate_prove_arbitrary(net, tr, inner_rounds, proof.inner);
    proof.shift_claim = Fr::zero();  // combine_additive_share; not appended to the transcript (coordinator.rs:113-117)
    for (Bytes& b : net.receive_responses()) {
        Reader rd(b);
        proof.shift_claim = Fr::add(proof.shift_claim, rd.fr());
    }
    (void)coordinate_prove_arbitrary(net, tr, steps_bits, proof.shift);
    proof.witness_evals = Rep3ProverOpeningAccumulator::receive_claims(net, tr);
    proof.shift_witness_evals = Rep3ProverOpeningAccumulator::receive_claims(net, tr);
    return proof;
}

static inline bool spartan_verify_rounds(const SumcheckProof& p, size_t rounds, size_t degree, fe& claim, Transcript& tr, std::vector<fe>& rs) {
    if (p.compressed_polys.size() != rounds) return false;
    rs.clear();
    for (const auto& comp : p.compressed_polys) {
        if (comp.size() != degree) return false;
        std::vector<fe> poly = unipoly_decompress(comp, claim);
        tr.append_scalars(comp);
        fe r_j = tr.challenge_scalar();
        rs.push_back(r_j);
        claim = unipoly_eval(poly, r_j);
    }
    return true;
}

// the sumcheck checks of the plain verifier (jolt-core UniformSpartanProof::verify, out of tree).  On success rx_step and
// shift_r are the two opening points and rho[2] the batching challenges of the two claim exchanges.
static inline bool verify_spartan(const JoltSpartanProof& proof, const jolt::System& sys, size_t num_steps, Transcript& tr, std::vector<fe>& rx_step,
                                  std::vector<fe>& shift_r, fe rho[2], std::string& why) {
    const size_t V = spartan_vars_padded(sys), nvars = sys.num_vars;
    int steps_bits = 0;
    while (((size_t)1 << steps_bits) < num_steps) steps_bits++;
    int constr_bits = 0;
    while (((size_t)1 << constr_bits) < sys.padded) constr_bits++;
    std::vector<fe> tau = tr.challenge_vector((size_t)(steps_bits + constr_bits)), rs;
    if (!verify_outer_sumcheck(proof.outer, tau, tr, rs)) {
        why = "spartan: outer sumcheck";
        return false;
    }
    std::vector<fe> outer_r(rs.rbegin(), rs.rend());
    rx_step.assign(outer_r.begin(), outer_r.begin() + steps_bits);
    std::vector<fe> rx_constr(outer_r.begin() + steps_bits, outer_r.end());
    fe rlc = tr.challenge_scalar();
    fe claim = Fr::add(proof.outer.claims[0], Fr::add(Fr::mul(rlc, proof.outer.claims[1]), Fr::mul(Fr::mul(rlc, rlc), proof.outer.claims[2])));
    int inner_rounds = 0;
    while (((size_t)1 << inner_rounds) < 4 * V) inner_rounds++;
    std::vector<fe> inner_r;
    if (!spartan_verify_rounds(proof.inner, (size_t)inner_rounds, 2, claim, tr, inner_r) || proof.witness_evals.size() != nvars ||
        proof.shift_witness_evals.size() != nvars) {
        why = "spartan: inner sumcheck shape";
        return false;
    }
    std::vector<fe> ry_var(inner_r.begin() + 1, inner_r.end());
    std::vector<fe> eq_ry = eq_evals_host(ry_var);
    fe z_eval = eq_ry[V];  // the constant column
    for (size_t i = 0; i < nvars; i++) z_eval = Fr::add(z_eval, Fr::mul(eq_ry[i], proof.witness_evals[i]));
    fe z_comb = Fr::add(Fr::mul(Fr::sub(Fr::one(), inner_r[0]), z_eval), Fr::mul(inner_r[0], proof.shift_claim));
    std::vector<fe> abc = spartan_matrix_mle_partial(sys, rx_constr, rlc), eq_y = eq_evals_host(inner_r);
    fe abc_eval = Fr::zero();
    for (size_t i = 0; i < abc.size(); i++) abc_eval = Fr::add(abc_eval, Fr::mul(abc[i], eq_y[i]));
    if (!Fr::eq(Fr::mul(abc_eval, z_comb), claim)) {
        why = "spartan: inner sumcheck final claim != ABC(r) z(r)";
        return false;
    }
    claim = proof.shift_claim;
    if (!spartan_verify_rounds(proof.shift, (size_t)steps_bits, 2, claim, tr, shift_r)) {
        why = "spartan: shift sumcheck shape";
        return false;
    }
    fe z_shift = Fr::zero();
    for (size_t i = 0; i < nvars; i++) z_shift = Fr::add(z_shift, Fr::mul(eq_ry[i], proof.shift_witness_evals[i]));
    if (!Fr::eq(Fr::mul(z_shift, eq_plus_one_point(rx_step, shift_r)), claim)) {
        why = "spartan: shift sumcheck final claim != z(ry_var, r) eq_plus_one(rx_step, r)";
        return false;
    }
    rho[0] = tr.challenge_scalar();  // receive_claims x 2
    rho[1] = tr.challenge_scalar();
    return true;
}

}  // namespace cozk
