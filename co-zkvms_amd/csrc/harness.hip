// In-process harness: the counterpart of the reference's runner (co-jolt/examples/rep3_jolt.rs:118-317
// + run_3_party_jolt.sh) for the hot path.  From a seed it (i) synthesises the witness polynomials of
// one trace, (ii) shares them Rep3 (mpc-core/.../arithmetic.rs:21-33) or
// keeps them plain, (iii) runs on one worker thread per party (one ctx / HIP stream each):
//       commit -> dense grand product (construct + prove) -> batch_evaluate + append -> reduce_and_prove
//     while the calling thread plays the coordinator (transcript owner), and (iv) verifies:
//     GKR proof, final GKR claim == direct leaf evaluation, opening-reduction sumcheck, PST opening
//     (pairing-free, trapdoor known).  Setup (SRS + witness resident in HBM) is separate from the
//     timed `prove` step.
#include <chrono>
#include <thread>

#include "host/prover.hpp"

using namespace cozk;

static uint64_t sm_next_host(uint64_t& s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// element i of the stream `seed` (same generator as cozk_vec_fill_random / oracle synthetic_fr)
// max_bits in 1..253 keeps the low bits of the canonical value (cozk_vec_fill_random's max_bits, pyref.synthetic_fr's)
static fe synthetic_fr_host(uint64_t seed, uint64_t i, int max_bits = 0) {
    uint64_t s = seed + i * 0xD1342543DE82EF95ull;
    fe v;
    for (;;) {
        uint64_t w0 = sm_next_host(s), w1 = sm_next_host(s), w2 = sm_next_host(s), w3 = sm_next_host(s) & ((1ull << 62) - 1ull);
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
    return Fr::to_mont(v);
}

__global__ void k_small_to_fr_u8(const uint8_t* in, fe* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store(out + i, Fr::from_u64(in[i]));
}
__global__ void k_small_to_fr_u16(const uint16_t* in, fe* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store(out + i, Fr::from_u64(in[i]));
}
__global__ void k_small_to_fr_u32(const uint32_t* in, fe* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store(out + i, Fr::from_u64(in[i]));
}

namespace {

struct PartyState {
    cozk_ctx* ctx = nullptr;
    bool own_ctx = false;
    int party = 0;
    // committed polynomials in commit order: first the shared (FR) ones, then the public ones
    std::vector<PolyH> polys;        // evaluation view (REP3 shares, or PLAIN Fr values for public polys)
    std::vector<VecH> commit_vecs;   // what the MSM consumes: share-a view (FR) or the compact small-scalar vector
    std::vector<int> is_public;
    std::vector<PolyH> small_polys;  // shared polynomials of length N / 16 ("final_cts"-like)
    std::vector<VecH> small_commit_vecs;
    LayerH leaves;                   // grand-product leaves (kept intact; each prove works on a clone)
    std::unique_ptr<PST13Setup> setup;
    // per-phase wall times of the last prove (ms)
    double t_commit = 0, t_construct = 0, t_gp = 0, t_eval = 0, t_open = 0, t_total = 0;
    uint64_t star_up = 0, star_down = 0, ring_bytes = 0, star_msgs = 0;
    std::string error;
};

struct ProofBundle {
    std::vector<PST13Commitment> commitments, small_commitments;
    GrandProductProof gp;
    std::vector<std::vector<fe>> opening_claims;  // per append
    ReducedOpeningProof reduced;
    Bytes serialize() const {
        Writer w;
        w.u64(commitments.size());
        for (auto& c : commitments) {
            w.u64(c.nv);
            w.g1(c.g_product);
        }
        w.u64(small_commitments.size());
        for (auto& c : small_commitments) {
            w.u64(c.nv);
            w.g1(c.g_product);
        }
        w.vec_fr(gp.outputs);
        w.u64(gp.gkr_layers.size());
        for (auto& l : gp.gkr_layers) {
            w.u64(l.proof.compressed_polys.size());
            for (auto& p : l.proof.compressed_polys) w.vec_fr(p);
            w.fr(l.left_claim);
            w.fr(l.right_claim);
        }
        w.u64(opening_claims.size());
        for (auto& c : opening_claims) w.vec_fr(c);
        w.u64(reduced.sumcheck_proof.compressed_polys.size());
        for (auto& p : reduced.sumcheck_proof.compressed_polys) w.vec_fr(p);
        w.vec_fr(reduced.sumcheck_claims);
        w.vec_g1(reduced.joint_opening_proof);
        return w.b;
    }
};

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct cozk_harness {
    cozk_harness_config cfg;
    int local_party = -1;  // >= 0: distributed form, only this (party, worker) participant lives in this process
    int local_worker = 0;
    int nparties = 1;
    size_t N = 0;
    std::vector<PartyState> parties;
    std::string error;
    Bytes last_proof;
};

// --------------------------------------------------------------------------- setup
static VecH make_vec_random(cozk_ctx* ctx, size_t n, int kind, uint64_t seed, int bits) {
    cozk_vec* v = nullptr;
    rc_check(cozk_vec_alloc(ctx, n, kind, &v), ctx, "vec_alloc");
    VecH h(v);
    rc_check(cozk_vec_fill_random(ctx, v, seed, bits), ctx, "fill_random");
    return h;
}
static VecH vec_binop(cozk_ctx* ctx, int op, const VecH& a, const VecH& b) {
    cozk_vec* o = nullptr;
    rc_check(cozk_vec_alloc(ctx, cozk_vec_len(a.h), COZK_SCALAR_FR, &o), ctx, "vec_alloc");
    VecH h(o);
    rc_check(cozk_vec_binop(ctx, op, 0, a.h, b.h, o), ctx, "vec_binop");
    return h;
}

// share components of the synthetic secret vector stream(seed) through the engine's witness scatter
// (cozk_rep3_share_vec: t0 = PRF(k0, i), t1 = PRF(k1, i), t2 = v - t0 - t1; P0 = (t0, t2), P1 = (t1, t0), P2 = (t2, t1);
// arithmetic.rs:21-33) with the harness keys (seed, 101) and (seed, 102)
static void make_share_vectors(cozk_ctx* ctx, size_t n, uint64_t seed, int party, int mode, VecH& a, VecH& b, int max_bits = 0) {
    if (mode == COZK_MODE_PLAIN) {
        a = make_vec_random(ctx, n, COZK_SCALAR_FR, seed, max_bits);
        return;
    }
    VecH v = make_vec_random(ctx, n, COZK_SCALAR_FR, seed, max_bits);
    uint8_t k0[COZK_PRF_KEY_BYTES], k1[COZK_PRF_KEY_BYTES];
    harness_prf_key(seed, 101, k0);
    harness_prf_key(seed, 102, k1);
    cozk_vec *sa = nullptr, *sb = nullptr;
    rc_check(cozk_rep3_share_vec(ctx, v.h, k0, k1, 0, party, &sa, &sb), ctx, "rep3_share_vec");
    a = VecH(sa);
    b = VecH(sb);
}

static void setup_party(cozk_harness* h, PartyState& ps) {
    const cozk_harness_config& c = h->cfg;
    cozk_ctx* ctx = ps.ctx;
    size_t N = h->N;
    int nv = c.log_n;
    // SRS from the seeded trapdoor
    std::vector<fe> t(nv);
    for (int i = 0; i < nv; i++) t[i] = synthetic_fr_host(c.seed ^ 0x7A7A7A7Aull, (uint64_t)i);
    ps.setup = PST13::setup(ctx, t, c.precompute);
    int j = 0;
    for (int k = 0; k < c.n_fr; k++, j++) {
        VecH a, b;
        make_share_vectors(ctx, N, c.seed + 1000ull * (uint64_t)(j + 1), ps.party, c.mode, a, b);
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
        for (int k = 0; k < sp.count; k++, j++) {
            VecH sv = make_vec_random(ctx, N, sp.kind, c.seed + 1000ull * (uint64_t)(j + 1), sp.bits);
            cozk_vec* frv = nullptr;
            rc_check(cozk_vec_alloc(ctx, N, COZK_SCALAR_FR, &frv), ctx, "vec_alloc");
            VecH fr(frv);
            unsigned grid = (unsigned)((N + 255) / 256);
            if (sp.kind == COZK_SCALAR_U8) k_small_to_fr_u8<<<grid, 256, 0, ctx->stream>>>((const uint8_t*)cozk_vec_device_ptr(sv.h), (fe*)cozk_vec_device_ptr(fr.h), N);
            else if (sp.kind == COZK_SCALAR_U16) k_small_to_fr_u16<<<grid, 256, 0, ctx->stream>>>((const uint16_t*)cozk_vec_device_ptr(sv.h), (fe*)cozk_vec_device_ptr(fr.h), N);
            else k_small_to_fr_u32<<<grid, 256, 0, ctx->stream>>>((const uint32_t*)cozk_vec_device_ptr(sv.h), (fe*)cozk_vec_device_ptr(fr.h), N);
            HIP_TRY(hipGetLastError());
            cozk_poly* p = nullptr;
            rc_check(cozk_poly_create(ctx, COZK_MODE_PLAIN, fr.h, nullptr, &p), ctx, "poly_create");
            ps.polys.push_back(PolyH(p));
            ps.commit_vecs.push_back(std::move(sv));
            ps.is_public.push_back(1);
        }
    }
    if (c.n_small > 0) {
        COZK_REQUIRE(c.log_n >= 5, "n_small needs log_n >= 5");
        size_t M = N >> 4;
        for (int k = 0; k < c.n_small; k++) {
            VecH a, b;
            make_share_vectors(ctx, M, c.seed + 300000ull + 1000ull * (uint64_t)k, ps.party, c.mode, a, b);
            cozk_poly* p = nullptr;
            rc_check(cozk_poly_create(ctx, c.mode, a.h, b.h, &p), ctx, "poly_create");
            ps.small_polys.push_back(PolyH(p));
            cozk_vec* view = nullptr;
            rc_check(cozk_poly_share_view(ctx, p, 0, &view), ctx, "share_view");
            ps.small_commit_vecs.push_back(VecH(view));
        }
    }
    // grand-product leaves: gp_batch circuits x 2^gp_log_leaves interleaved entries (seeded random shares, unless
    // they are computed per prove as fingerprints of the committed columns)
    if (c.leaf_fingerprints) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return;
    }
    size_t nleaves = (size_t)c.gp_batch << c.gp_log_leaves;
    VecH la, lb;
    make_share_vectors(ctx, nleaves, c.seed + 500000ull, ps.party, c.mode, la, lb);
    cozk_layer* lv = nullptr;
    rc_check(cozk_layer_create(ctx, c.mode, la.h, lb.h, 1, &lv), ctx, "layer_create");
    ps.leaves = LayerH(lv);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
}

// --------------------------------------------------------------------------- worker
static void put_commitments(Writer& w, const std::vector<PST13Commitment>& cs, const std::vector<int>& is_public, int party) {
    w.u64(cs.size());
    for (size_t i = 0; i < cs.size(); i++) {
        // MaybeShared tag: 0 Public(None), 1 Public(Some), 2 Shared (pst13.rs:140-162)
        uint8_t tag = is_public[i] ? (party == 0 ? 1 : 0) : 2;
        w.b.push_back(tag);
        if (tag != 0) {
            w.u64(cs[i].nv);
            w.g1(cs[i].g_product);
        }
    }
}

static void worker_main(cozk_harness* h, PartyState& ps, StarNetWorker* star, RingNet* ring) {
    const cozk_harness_config& c = h->cfg;
    WorkerEnv env;
    env.ctx = ps.ctx;
    env.mode = c.mode;
    env.party = ps.party;
    env.star = star;
    env.ring = ring;
    harness_prf_key(c.seed, (uint64_t)ps.party, env.key_self);
    harness_prf_key(c.seed, (uint64_t)((ps.party + 2) % 3), env.key_prev);
    HIP_TRY(hipSetDevice(ps.ctx->device));
    double t_start = now_ms();
    // ---- 1. commit (Rep3JoltPolynomials::commit, jolt/vm/jolt/witness.rs:304-382): every party MSMs
    //         every polynomial; only P0's public commitments are kept (pst13.rs:165-229)
    {
        std::vector<cozk_vec*> vs;
        for (auto& v : ps.commit_vecs) vs.push_back(v.h);
        std::vector<PST13Commitment> cm = PST13::batch_commit(ps.ctx, *ps.setup, vs);
        std::vector<cozk_vec*> sv;
        for (auto& v : ps.small_commit_vecs) sv.push_back(v.h);
        std::vector<PST13Commitment> scm = PST13::batch_commit(ps.ctx, *ps.setup, sv);
        Writer w;
        put_commitments(w, cm, ps.is_public, ps.party);
        std::vector<int> none(scm.size(), 0);
        put_commitments(w, scm, none, ps.party);
        star->send_response(w.b);
    }
    double t1 = now_ms();
    ps.t_commit = t1 - t_start;
    // ---- 2. dense grand product (memory-checking shape, lasso/memory_checking/worker.rs:77-127)
    cozk_layer* lv = nullptr;
    LayerH fresh_leaves;  // fingerprint mode: this prove's leaves (kept for the leaf-evaluation check below)
    if (c.leaf_fingerprints) {
        // compute_leaves (K11): receive (gamma, tau), fingerprint the committed columns circuit by circuit
        Bytes req = star->receive_request();
        Reader rd(req);
        fe gamma = rd.fr(), tau = rd.fr();
        size_t N = h->N, nleaves = (size_t)c.gp_batch << c.gp_log_leaves;
        cozk_vec *la = nullptr, *lb = nullptr;
        rc_check(cozk_vec_alloc(ps.ctx, nleaves, COZK_SCALAR_FR, &la), ps.ctx, "vec_alloc");
        VecH lah(la), lbh;
        if (c.mode == COZK_MODE_REP3) {
            rc_check(cozk_vec_alloc(ps.ctx, nleaves, COZK_SCALAR_FR, &lb), ps.ctx, "vec_alloc");
            lbh = VecH(lb);
        }
        const int first_u16 = c.n_fr, first_u32 = c.n_fr + c.n_u16, first_flag = c.n_fr + c.n_u16 + c.n_u32;
        for (int circ = 0; circ < c.gp_batch; circ++) {
            std::vector<const cozk_vec*> cols;
            std::vector<fe> cc;
            fe g = gamma;
            auto take = [&](int first, int count) {
                if (count <= 0) return;
                cols.push_back(ps.commit_vecs[(size_t)(first + circ % count)].h);
                cc.push_back(g);
                g = Fr::mul(g, gamma);
            };
            take(first_u16, c.n_u16);
            take(first_u32, c.n_u32);
            take(first_flag, c.n_flags);
            const cozk_poly* shared[1] = {ps.polys[(size_t)(circ % c.n_fr)].h};
            fe g_sh = g, g_w = Fr::mul(g, gamma);
            std::vector<uint64_t> ccw = to_abi(cc);
            uint64_t dsh[4], cst[4];
            fe_to_u64x4(g_sh, dsh);
            size_t base = (size_t)circ << c.gp_log_leaves;
            fe_to_u64x4(Fr::neg(tau), cst);  // read leaves
            rc_check(cozk_fingerprint_leaves(ps.ctx, cols.data(), ccw.data(), cols.size(), shared, dsh, 1, cst, c.mode, ps.party, lah.h, lbh.h, base, N),
                     ps.ctx, "fingerprint_leaves");
            fe_to_u64x4(Fr::sub(g_w, tau), cst);  // write leaves: + gamma^(k+1)
            rc_check(cozk_fingerprint_leaves(ps.ctx, cols.data(), ccw.data(), cols.size(), shared, dsh, 1, cst, c.mode, ps.party, lah.h, lbh.h, base + N, N),
                     ps.ctx, "fingerprint_leaves");
        }
        cozk_layer* fl = nullptr;
        rc_check(cozk_layer_create(ps.ctx, c.mode, lah.h, lbh.h, 1, &fl), ps.ctx, "layer_create");
        fresh_leaves = LayerH(fl);
        rc_check(cozk_layer_clone(ps.ctx, fl, &lv), ps.ctx, "layer_clone");
    } else {
        rc_check(cozk_layer_clone(ps.ctx, ps.leaves.h, &lv), ps.ctx, "layer_clone");  // stand-in for compute_leaves
    }
    const cozk_layer* check_leaves = c.leaf_fingerprints ? fresh_leaves.h : ps.leaves.h;
    Rep3BatchedDenseGrandProduct gp = Rep3BatchedDenseGrandProduct::construct(env, LayerH(lv), (size_t)c.gp_batch);
    HIP_TRY(hipStreamSynchronize(ps.ctx->stream));
    double t2 = now_ms();
    ps.t_construct = t2 - t1;
    t_round_trace = RoundTrace();
    std::vector<fe> r_gp = gp.prove_grand_product_worker(env);
    double t3 = now_ms();
    ps.t_gp = t3 - t2;
    if (getenv("COZK_TRACE_ROUNDS"))
        fprintf(stderr, "[rounds] party %d: %llu rounds, round loop %.2f ms (of which star round trips %.2f ms), gp_prove %.2f ms\n", ps.party,
                (unsigned long long)t_round_trace.rounds, t_round_trace.t_round / 1e3, t_round_trace.t_star / 1e3, ps.t_gp);
    // harness check message (not part of the proof): additive share of leaves(r_gp)
    {
        // the original leaves as a dense polynomial over (circuit, position, l/r) variables
        std::vector<uint64_t> rr = to_abi(r_gp);
        cozk_vec* chi = nullptr;
        rc_check(cozk_eq_evals(ps.ctx, rr.data(), (int)r_gp.size(), &chi), ps.ctx, "eq_evals");
        VecH chih(chi);
        cozk_poly* lp = nullptr;
        rc_check(cozk_layer_as_poly(ps.ctx, check_leaves, &lp), ps.ctx, "layer_as_poly");
        PolyH lph(lp);
        uint64_t ev[4];
        const cozk_poly* arr[1] = {lph.h};
        rc_check(cozk_poly_batch_evaluate_at_chi(ps.ctx, arr, 1, chih.h, ev), ps.ctx, "leaf evaluation");
        Writer w;
        w.fr(fe_from_u64x4(ev));
        star->send_response(w.b);
    }
    // ---- 3. openings: batch_evaluate + append (compute_openings -> opening_accumulator.append)
    Rep3ProverOpeningAccumulator acc;
    size_t K = ps.polys.size();
    size_t half = (K + 1) / 2;
    int nv = c.log_n;
    auto open_group = [&](size_t lo, size_t hi, const std::vector<fe>& point, std::vector<PolyH>& src) {
        if (hi <= lo) return;
        std::vector<uint64_t> rr = to_abi(point);
        cozk_vec* chi = nullptr;
        rc_check(cozk_eq_evals(ps.ctx, rr.data(), (int)point.size(), &chi), ps.ctx, "eq_evals");
        VecH chih(chi);
        // shared and public polynomials evaluate in their own batches (additive vs trivial share)
        std::vector<cozk_poly*> ps_all;
        std::vector<fe> claims(hi - lo);
        std::vector<const cozk_poly*> shp, pbp;
        std::vector<size_t> shi, pbi;
        for (size_t i = lo; i < hi; i++) {
            ps_all.push_back(src[i].h);
            if (cozk_poly_mode(src[i].h) == c.mode) {
                shp.push_back(src[i].h);
                shi.push_back(i - lo);
            } else {
                pbp.push_back(src[i].h);
                pbi.push_back(i - lo);
            }
        }
        if (!shp.empty()) {
            std::vector<uint64_t> out(4 * shp.size());
            rc_check(cozk_poly_batch_evaluate_at_chi(ps.ctx, shp.data(), shp.size(), chih.h, out.data()), ps.ctx, "batch_evaluate");
            for (size_t k = 0; k < shp.size(); k++) claims[shi[k]] = fe_from_u64x4(out.data() + 4 * k);
        }
        if (!pbp.empty()) {
            std::vector<uint64_t> out(4 * pbp.size());
            rc_check(cozk_poly_batch_evaluate_at_chi(ps.ctx, pbp.data(), pbp.size(), chih.h, out.data()), ps.ctx, "batch_evaluate(public)");
            // a public evaluation is held by P0 only (additive::promote_to_trivial_share)
            for (size_t k = 0; k < pbp.size(); k++) claims[pbi[k]] = env.additive_trivial(fe_from_u64x4(out.data() + 4 * k));
        }
        acc.append(env, ps_all, chih.h, point, claims);
    };
    {
        std::vector<fe> p1(r_gp.end() - nv, r_gp.end());
        std::vector<fe> p2(r_gp.begin(), r_gp.begin() + nv);
        open_group(0, half, p1, ps.polys);
        open_group(half, K, p2, ps.polys);
        if (!ps.small_polys.empty()) {
            std::vector<fe> p3(r_gp.end() - (nv - 4), r_gp.end());
            open_group(0, ps.small_polys.size(), p3, ps.small_polys);
        }
    }
    double t4 = now_ms();
    ps.t_eval = t4 - t3;
    // ---- 4. reduce_and_prove_worker (opening_proof.rs:238-291)
    acc.reduce_and_prove_worker(env, *ps.setup);
    double t5 = now_ms();
    ps.t_open = t5 - t4;
    ps.t_total = t5 - t_start;
    ps.star_up = star->bytes_up;
    ps.star_down = star->bytes_down;
    ps.star_msgs = star->n_msgs;
    ps.ring_bytes = ring ? ring->bytes_sent : 0;
}

// --------------------------------------------------------------------------- coordinator + verifier
static std::vector<PST13Commitment> combine_commitments_from_parties(std::vector<Reader>& rds, int nparties) {
    std::vector<PST13Commitment> out;
    uint64_t n = 0;
    for (int p = 0; p < nparties; p++) {
        uint64_t m = rds[p].u64();
        if (p == 0) n = m;
        if (m != n) throw CozkError(COZK_ERR_INTERNAL, "commitment count mismatch");
    }
    for (uint64_t i = 0; i < n; i++) {
        std::vector<PST13Commitment> shares;
        bool have_public = false;
        PST13Commitment pub{};
        for (int p = 0; p < nparties; p++) {
            rds[p].need(1);
            uint8_t tag = *rds[p].p++;
            if (tag == 0) continue;
            PST13Commitment c;
            c.nv = rds[p].u64();
            c.g_product = rds[p].g1();
            if (tag == 1) {
                have_public = true;
                pub = c;
            } else {
                shares.push_back(c);
            }
        }
        if (have_public) out.push_back(pub);
        else out.push_back(PST13::combine_commitment_shares(shares));
    }
    return out;
}

static fe eq_eval(const std::vector<fe>& a, const std::vector<fe>& b) {
    fe one = Fr::one(), acc = one;
    for (size_t i = 0; i < a.size(); i++) {
        fe ab = Fr::mul(a[i], b[i]);
        acc = Fr::mul(acc, Fr::add(Fr::sub(Fr::sub(one, a[i]), b[i]), Fr::dbl(ab)));
    }
    return acc;
}

static int coordinator_main(cozk_harness* h, StarNetCoordinator& net, ProofBundle& proof, bool verify, std::string& why) {
    const cozk_harness_config& c = h->cfg;
    int np = h->nparties;
    int nv = c.log_n;
    Transcript tr("cozk-harness");
    // receive_commitments (jolt/vm/jolt/witness.rs:221-285)
    {
        std::vector<Bytes> msgs = net.receive_responses();
        std::vector<Reader> rds;
        for (auto& m : msgs) rds.emplace_back(m);
        proof.commitments = combine_commitments_from_parties(rds, np);
        proof.small_commitments = combine_commitments_from_parties(rds, np);
        for (auto& cm : proof.commitments) tr.append_point(cm.g_product);
        for (auto& cm : proof.small_commitments) tr.append_point(cm.g_product);
    }
    if (c.leaf_fingerprints) {  // receive_gamma_tau (lasso/memory_checking/worker.rs:85)
        fe gamma = tr.challenge_scalar(), tau = tr.challenge_scalar();
        Writer w;
        w.fr(gamma);
        w.fr(tau);
        net.broadcast_request(w.b);
    }
    fe gp_claim;
    std::vector<fe> r_gp;
    size_t num_layers = (size_t)c.gp_log_leaves;
    proof.gp = coordinate_prove_grand_product(net, tr, num_layers, gp_claim, r_gp);
    // harness check: sum of the parties' additive leaf evaluations
    fe leaf_eval = Fr::zero();
    for (Bytes& b : net.receive_responses()) {
        Reader rd(b);
        leaf_eval = Fr::add(leaf_eval, rd.fr());
    }
    size_t K = proof.commitments.size();
    size_t half = (K + 1) / 2;
    int nappend = 2 + (c.n_small > 0 ? 1 : 0);
    if (half >= K) nappend -= 1;
    Transcript tr_open_start = tr;  // the verifier replays from here
    std::vector<fe> rhos;
    for (int a = 0; a < nappend; a++) {
        Transcript before = tr;
        proof.opening_claims.push_back(Rep3ProverOpeningAccumulator::receive_claims(net, tr));
        (void)before;
    }
    std::vector<fe> r_red;
    fe rho_red, gamma;
    proof.reduced = Rep3ProverOpeningAccumulator::reduce_and_prove(net, tr, r_red, rho_red, gamma);
    if (!verify) return -1;

    // ------------------------------------------------ plain verifier (replays its own transcript)
    Transcript vt("cozk-harness");
    for (auto& cm : proof.commitments) vt.append_point(cm.g_product);
    for (auto& cm : proof.small_commitments) vt.append_point(cm.g_product);
    if (c.leaf_fingerprints) {
        (void)vt.challenge_scalar();  // gamma, tau
        (void)vt.challenge_scalar();
    }
    fe v_claim;
    std::vector<fe> v_r;
    if (!verify_grand_product(proof.gp, vt, v_claim, v_r)) {
        why = "GKR proof rejected";
        return 0;
    }
    if (v_r.size() != r_gp.size() || !Fr::eq(v_claim, gp_claim)) {
        why = "GKR verifier / coordinator disagree";
        return 0;
    }
    if (!Fr::eq(v_claim, leaf_eval)) {
        why = "final GKR claim != direct evaluation of the leaves";
        return 0;
    }
    // openings: points as the workers chose them
    struct VOpen { std::vector<fe> point; std::vector<g1_affine> cs; std::vector<fe> claims; fe rho; };
    std::vector<VOpen> vo;
    {
        std::vector<fe> p1(v_r.end() - nv, v_r.end()), p2(v_r.begin(), v_r.begin() + nv);
        VOpen o1;
        o1.point = p1;
        for (size_t i = 0; i < half; i++) o1.cs.push_back(proof.commitments[i].g_product);
        vo.push_back(o1);
        if (half < K) {
            VOpen o2;
            o2.point = p2;
            for (size_t i = half; i < K; i++) o2.cs.push_back(proof.commitments[i].g_product);
            vo.push_back(o2);
        }
        if (c.n_small > 0) {
            VOpen o3;
            o3.point.assign(v_r.end() - (nv - 4), v_r.end());
            for (auto& cm : proof.small_commitments) o3.cs.push_back(cm.g_product);
            vo.push_back(o3);
        }
    }
    if (vo.size() != proof.opening_claims.size()) {
        why = "opening count mismatch";
        return 0;
    }
    std::vector<fe> batched_claims;
    std::vector<g1_affine> batched_commitments;
    for (size_t a = 0; a < vo.size(); a++) {
        vo[a].claims = proof.opening_claims[a];
        if (vo[a].claims.size() != vo[a].cs.size()) {
            why = "claims / commitments mismatch";
            return 0;
        }
        vo[a].rho = vt.challenge_scalar();
        std::vector<fe> pw(1, Fr::one());
        for (size_t i = 1; i < vo[a].claims.size(); i++) pw.push_back(Fr::mul(pw[i - 1], vo[a].rho));
        fe bc = Fr::zero();
        for (size_t i = 0; i < pw.size(); i++) bc = Fr::add(bc, Fr::mul(pw[i], vo[a].claims[i]));
        batched_claims.push_back(bc);
        batched_commitments.push_back(PST13::combine_commitments(vo[a].cs, pw));
    }
    // reduction sumcheck
    fe rho2 = vt.challenge_scalar();
    size_t max_nv = 0;
    for (auto& o : vo) max_nv = std::max(max_nv, o.point.size());
    std::vector<fe> coeffs(1, Fr::one());
    for (size_t i = 1; i < vo.size(); i++) coeffs.push_back(Fr::mul(coeffs[i - 1], rho2));
    fe e = Fr::zero();
    for (size_t i = 0; i < vo.size(); i++)
        e = Fr::add(e, Fr::mul(coeffs[i], Fr::mul(batched_claims[i], fr_from_u64((uint64_t)1 << (max_nv - vo[i].point.size())))));
    std::vector<fe> rs;
    if (proof.reduced.sumcheck_proof.compressed_polys.size() != max_nv) {
        why = "reduction sumcheck: wrong number of rounds";
        return 0;
    }
    for (auto& comp : proof.reduced.sumcheck_proof.compressed_polys) {
        std::vector<fe> poly = unipoly_decompress(comp, e);
        vt.append_scalars(comp);
        fe r_j = vt.challenge_scalar();
        rs.push_back(r_j);
        e = unipoly_eval(poly, r_j);
    }
    fe expect = Fr::zero();
    for (size_t i = 0; i < vo.size(); i++) {
        std::vector<fe> slice(rs.end() - vo[i].point.size(), rs.end());
        expect = Fr::add(expect, Fr::mul(coeffs[i], Fr::mul(eq_eval(vo[i].point, slice), proof.reduced.sumcheck_claims[i])));
    }
    if (!Fr::eq(expect, e)) {
        why = "reduction sumcheck: final check failed";
        return 0;
    }
    vt.append_scalars(proof.reduced.sumcheck_claims);
    fe vgamma = vt.challenge_scalar();
    // joint commitment / claim
    std::vector<fe> gp_pw(1, Fr::one());
    for (size_t i = 1; i < vo.size(); i++) gp_pw.push_back(Fr::mul(gp_pw[i - 1], vgamma));
    g1_affine joint_c = PST13::combine_commitments(batched_commitments, gp_pw);
    fe joint_claim = Fr::zero();
    fe one = Fr::one();
    for (size_t i = 0; i < vo.size(); i++) {
        fe sc = one;
        for (size_t j = 0; j + vo[i].point.size() < max_nv; j++) sc = Fr::mul(sc, Fr::sub(one, rs[j]));
        joint_claim = Fr::add(joint_claim, Fr::mul(gp_pw[i], Fr::mul(sc, proof.reduced.sumcheck_claims[i])));
    }
    std::vector<fe> rev(rs.rbegin(), rs.rend());
    const PST13Setup& vsetup = *h->parties[h->local_party >= 0 ? h->local_worker * h->nparties + h->local_party : 0].setup;  // same SRS everywhere
    if (!PST13::check_with_trapdoor(vsetup, joint_c, rev, joint_claim, proof.reduced.joint_opening_proof)) {
        why = "PST13 opening check failed";
        return 0;
    }
    (void)tr_open_start;
    (void)rho_red;
    (void)gamma;
    (void)r_red;
    return 1;
}

#include "host/split_harness.hpp"

// --------------------------------------------------------------------------- C ABI
extern "C" {

int cozk_harness_create(const cozk_harness_config* cfg, cozk_harness** out) {
    if (!cfg || !out) return COZK_ERR_INVALID_ARG;
    *out = nullptr;
    cozk_harness* h = new cozk_harness();
    h->cfg = *cfg;
    try {
        COZK_REQUIRE(cfg->mode == COZK_MODE_PLAIN || cfg->mode == COZK_MODE_REP3, "harness: bad mode");
        COZK_REQUIRE(cfg->log_n >= 2 && cfg->log_n <= 24, "harness: log_n out of range");
        COZK_REQUIRE(cfg->gp_batch >= 1 && cfg->gp_log_leaves >= 1, "harness: bad grand-product shape");
        int gbits = 0;
        while ((1 << gbits) < cfg->gp_batch) gbits++;
        COZK_REQUIRE(gbits + cfg->gp_log_leaves >= cfg->log_n, "harness: grand-product point shorter than the opening point");
        COZK_REQUIRE(cfg->n_fr + cfg->n_u16 + cfg->n_u32 + cfg->n_flags >= 1, "harness: no polynomials");
        if (cfg->leaf_fingerprints)
            COZK_REQUIRE(cfg->gp_log_leaves == cfg->log_n + 1 && cfg->n_fr >= 1 && cfg->log_workers == 0,
                         "harness: leaf_fingerprints needs gp_log_leaves == log_n + 1, n_fr >= 1, log_workers == 0");
        h->nparties = cfg->mode == COZK_MODE_REP3 ? 3 : 1;
        h->N = (size_t)1 << cfg->log_n;
        COZK_REQUIRE(cfg->log_workers >= 0 && cfg->log_workers <= 3, "harness: log_workers must be 0..3");
        int W = 1 << cfg->log_workers;
        if (W > 1) COZK_REQUIRE((cfg->gp_batch & (cfg->gp_batch - 1)) == 0, "split: gp_batch must be a power of two");
        h->parties.resize((size_t)h->nparties * W);
        for (int w = 0; w < W; w++)
            for (int p = 0; p < h->nparties; p++) {
                PartyState& ps = h->parties[(size_t)w * h->nparties + p];
                ps.party = p;
                int dev = cfg->mode == COZK_MODE_REP3 ? cfg->devices[p] : (W > 1 ? cfg->worker_devices[w] : cfg->devices[0]);
                int rc = cozk_ctx_create(dev, &ps.ctx);
                if (rc != COZK_OK) throw CozkError(rc, "harness: cannot create a context (no HIP device?)");
                ps.own_ctx = true;
                // several participants driven from this one process wait for each other's round messages: no resident
                // round kernels (cozk_ctx_set_resident_rounds); a lone participant depends on nobody's GPU work, so it
                // keeps them even when the host has other, unrelated contexts open on the device
                cozk_ctx_set_resident_rounds(ps.ctx, h->nparties * W > 1 ? 0 : 1);
                HIP_TRY(hipSetDevice(ps.ctx->device));
                if (W > 1) setup_participant_split(h, ps, w);
                else setup_party(h, ps);
            }
    } catch (const CozkError& e) {
        h->error = e.what();
        *out = h;  // caller reads the error, then destroys
        return e.code;
    } catch (const std::exception& e) {
        h->error = e.what();
        *out = h;
        return COZK_ERR_INTERNAL;
    }
    *out = h;
    return COZK_OK;
}

const char* cozk_harness_error(const cozk_harness* h) { return h ? h->error.c_str() : "null harness"; }

int cozk_harness_destroy(cozk_harness* h) {
    if (!h) return COZK_OK;
    for (auto& ps : h->parties) {
        if (ps.ctx) (void)hipSetDevice(ps.ctx->device);
        ps.polys.clear();
        ps.commit_vecs.clear();
        ps.small_polys.clear();
        ps.small_commit_vecs.clear();
        ps.leaves = LayerH();
        ps.setup.reset();
        if (ps.own_ctx && ps.ctx) cozk_ctx_destroy(ps.ctx);
    }
    delete h;
    return COZK_OK;
}

// one full pass of the hot path (the bench "step"); verify != 0 also runs the plain verifier
int cozk_harness_prove(cozk_harness* h, int verify, cozk_harness_result* res) {
    if (!h || !res) return COZK_ERR_INVALID_ARG;
    memset(res, 0, sizeof *res);
    res->verified = -1;
    if (h->local_party >= 0) return COZK_ERR_INVALID_ARG;  // distributed harness: use cozk_harness_prove_distributed
    const int W = 1 << h->cfg.log_workers;
    const int nparty = h->nparties;
    int np = nparty * W;  // participants: id = worker * nparty + party
    InProcStar star(np);
    std::vector<std::unique_ptr<InProcRing>> rings;  // one ring per worker index (its three parties)
    for (int w = 0; w < W; w++) rings.emplace_back(new InProcRing(&star.abort));
    std::vector<std::unique_ptr<InProcStarWorker>> sw;
    std::vector<std::unique_ptr<InProcRingNet>> rn;
    for (int p = 0; p < np; p++) {
        sw.emplace_back(new InProcStarWorker(&star, p));
        rn.emplace_back(nparty == 3 ? new InProcRingNet(rings[p / nparty].get(), p % nparty) : nullptr);
        h->parties[p].error.clear();
    }
    std::vector<std::thread> threads;
    double t0 = now_ms();
    for (int p = 0; p < np; p++) {
        threads.emplace_back([&, p] {
            try {
                if (W > 1) worker_main_split(h, h->parties[p], p / nparty, sw[p].get(), rn[p].get());
                else worker_main(h, h->parties[p], sw[p].get(), rn[p].get());
            } catch (const std::exception& e) {
                h->parties[p].error = e.what();
                star.abort.flag.store(true);
            }
        });
    }
    ProofBundle proof;
    std::string why;
    int verified = -1;
    int rc = COZK_OK;
    try {
        InProcStarCoordinator coord(&star);
        verified = W > 1 ? coordinator_main_split(h, coord, proof, verify != 0, why) : coordinator_main(h, coord, proof, verify != 0, why);
    } catch (const std::exception& e) {
        h->error = std::string("coordinator: ") + e.what();
        star.abort.flag.store(true);
        rc = COZK_ERR_INTERNAL;
    }
    for (auto& t : threads) t.join();
    double t1 = now_ms();
    for (int p = 0; p < np; p++) {
        if (!h->parties[p].error.empty()) {
            h->error = "participant " + std::to_string(p) + ": " + h->parties[p].error;
            rc = COZK_ERR_INTERNAL;
        }
    }
    if (rc != COZK_OK) return rc;
    if (verified == 0) h->error = "verification failed: " + why;
    res->verified = verified;
    res->wall_ms = t1 - t0;
    for (int p = 0; p < np; p++) {
        PartyState& ps = h->parties[p];
        res->t_commit_ms = std::max(res->t_commit_ms, ps.t_commit);
        res->t_gp_construct_ms = std::max(res->t_gp_construct_ms, ps.t_construct);
        res->t_gp_prove_ms = std::max(res->t_gp_prove_ms, ps.t_gp);
        res->t_eval_ms = std::max(res->t_eval_ms, ps.t_eval);
        res->t_open_ms = std::max(res->t_open_ms, ps.t_open);
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

// ---- distributed form: ONE party per process / GPU (BASELINE config 3: "one MI355X per party").
// Every process runs its own copy of the (deterministic) coordinator: a "star" gather is an all-gather of
// the parties' messages through the host's transport (`cozk_hub_net`, e.g. torch.distributed / RCCL), after
// which each copy derives the same challenge -- no coordinator process, no extra hop.  The ring reshare goes
// through `cozk_ring_net` on device pointers (an RCCL send/recv pair over xGMI).
int cozk_harness_create_participant(const cozk_harness_config* cfg, int local_party, int local_worker, cozk_harness** out) {
    if (!cfg || !out) return COZK_ERR_INVALID_ARG;
    *out = nullptr;
    cozk_harness* h = new cozk_harness();
    h->cfg = *cfg;
    try {
        COZK_REQUIRE(cfg->mode == COZK_MODE_PLAIN || cfg->mode == COZK_MODE_REP3, "harness: bad mode");
        int np = cfg->mode == COZK_MODE_REP3 ? 3 : 1;
        COZK_REQUIRE(cfg->log_workers >= 0 && cfg->log_workers <= 3, "harness: log_workers must be 0..3");
        int W = 1 << cfg->log_workers;
        COZK_REQUIRE(local_party >= 0 && local_party < np && local_worker >= 0 && local_worker < W, "harness_create_participant: bad (party, worker)");
        COZK_REQUIRE(cfg->log_n >= 2 && cfg->log_n <= 24 && cfg->gp_batch >= 1 && cfg->gp_log_leaves >= 1, "harness: bad shape");
        if (cfg->leaf_fingerprints)
            COZK_REQUIRE(cfg->gp_log_leaves == cfg->log_n + 1 && cfg->n_fr >= 1 && cfg->log_workers == 0,
                         "harness: leaf_fingerprints needs gp_log_leaves == log_n + 1, n_fr >= 1, log_workers == 0");
        int gbits = 0;
        while ((1 << gbits) < cfg->gp_batch) gbits++;
        COZK_REQUIRE(gbits + cfg->gp_log_leaves >= cfg->log_n, "harness: grand-product point shorter than the opening point");
        if (W > 1) COZK_REQUIRE((cfg->gp_batch & (cfg->gp_batch - 1)) == 0, "split: gp_batch must be a power of two");
        h->nparties = np;
        h->local_party = local_party;
        h->local_worker = local_worker;
        h->N = (size_t)1 << cfg->log_n;
        h->parties.resize((size_t)np * W);
        for (int w = 0; w < W; w++)
            for (int p = 0; p < np; p++) h->parties[(size_t)w * np + p].party = p;
        PartyState& ps = h->parties[(size_t)local_worker * np + local_party];
        int dev = cfg->mode == COZK_MODE_REP3 ? cfg->devices[local_party] : (W > 1 ? cfg->worker_devices[local_worker] : cfg->devices[0]);
        int rc = cozk_ctx_create(dev, &ps.ctx);
        if (rc != COZK_OK) throw CozkError(rc, "harness: cannot create a context (no HIP device?)");
        ps.own_ctx = true;
        HIP_TRY(hipSetDevice(ps.ctx->device));
        if (W > 1) setup_participant_split(h, ps, local_worker);
        else setup_party(h, ps);
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

int cozk_harness_create_party(const cozk_harness_config* cfg, int local_party, cozk_harness** out) {
    return cozk_harness_create_participant(cfg, local_party, 0, out);
}

namespace {
struct HubStarCoordinator : StarNetCoordinator {
    InProcStar* local;  // only slot `me` is used: the local worker's up/down channels
    cozk_hub_net hub;
    int me;
    std::unique_ptr<uint8_t[]> recv_buf;  // n x 256 KiB landing slots, allocated once (rounds are ~1 us apart)
    std::vector<size_t> lens;
    double t_wait_ms = 0;   // time inside the hub's all_gather (= waiting for the slowest participant of each exchange)
    uint64_t n_exchanges = 0;
    HubStarCoordinator(InProcStar* l, const cozk_hub_net& h, int me_) : local(l), hub(h), me(me_) {}
    std::vector<Bytes> gather(const Bytes& mine) {
        const size_t cap = 1 << 18;
        if (!recv_buf) recv_buf.reset(new uint8_t[(size_t)hub.n_participants * cap]);
        uint8_t* recv = recv_buf.get();
        lens.assign((size_t)hub.n_participants, 0);
        if (mine.size() > cap) throw CozkError(COZK_ERR_INTERNAL, "hub all_gather: message larger than the 256 KiB slot");
        const double tw0 = now_ms();
        if (hub.all_gather(hub.user, mine.data(), mine.size(), recv, cap, lens.data()) != 0)
            throw CozkError(COZK_ERR_INTERNAL, "hub all_gather callback failed");
        t_wait_ms += now_ms() - tw0;
        n_exchanges++;
        std::vector<Bytes> out;
        for (int p = 0; p < hub.n_participants; p++) {
            if (lens[p] > cap) throw CozkError(COZK_ERR_INTERNAL, "hub all_gather: bad length");
            out.emplace_back(recv + (size_t)p * cap, recv + (size_t)p * cap + lens[p]);
        }
        return out;
    }
    int n_workers() const override { return hub.n_participants; }
    std::vector<Bytes> receive_responses() override { return gather(local->up[me].pop()); }
    Bytes receive_response(int party) override {
        Bytes mine;
        if (party == me) mine = local->up[me].pop();
        return gather(mine)[party];
    }
    void broadcast_request(const Bytes& b) override { local->down[me].push(b); }
    void send_request(int party, const Bytes& b) override {
        if (party == me) local->down[me].push(b);
    }
};
}  // namespace

int cozk_harness_prove_distributed(cozk_harness* h, const cozk_hub_net* hub, const cozk_ring_net* ring, int verify,
                                   cozk_harness_result* res) {
    if (!h || !hub || !res || h->local_party < 0) return COZK_ERR_INVALID_ARG;
    const int W = 1 << h->cfg.log_workers;
    const int nparty = h->nparties;
    int me = h->local_worker * nparty + h->local_party;
    if (hub->n_participants != nparty * W || hub->my_index != me || (nparty == 3 && !ring)) return COZK_ERR_INVALID_ARG;
    memset(res, 0, sizeof *res);
    res->verified = -1;
    InProcStar star(nparty * W);
    InProcStarWorker sw(&star, me);
    std::unique_ptr<CallbackRingNet> rnp(nparty == 3 ? new CallbackRingNet(*ring) : nullptr);
    PartyState& ps = h->parties[me];
    ps.error.clear();
    double t0 = now_ms();
    std::thread worker([&] {
        try {
            if (W > 1) worker_main_split(h, ps, h->local_worker, &sw, rnp.get());
            else worker_main(h, ps, &sw, rnp.get());
        } catch (const std::exception& e) {
            ps.error = e.what();
            star.abort.flag.store(true);
        }
    });
    ProofBundle proof;
    std::string why;
    int verified = -1;
    int rc = COZK_OK;
    double hub_wait = 0;
    uint64_t hub_n = 0;
    try {
        HubStarCoordinator coord(&star, *hub, me);
        verified = W > 1 ? coordinator_main_split(h, coord, proof, verify != 0, why) : coordinator_main(h, coord, proof, verify != 0, why);
        hub_wait = coord.t_wait_ms;
        hub_n = coord.n_exchanges;
    } catch (const std::exception& e) {
        h->error = std::string("coordinator: ") + e.what();
        star.abort.flag.store(true);
        rc = COZK_ERR_INTERNAL;
    }
    worker.join();
    double t1 = now_ms();
    if (!ps.error.empty()) {
        h->error = "participant " + std::to_string(me) + ": " + ps.error;
        rc = COZK_ERR_INTERNAL;
    }
    if (rc != COZK_OK) return rc;
    if (verified == 0) h->error = "verification failed: " + why;
    res->verified = verified;
    res->wall_ms = t1 - t0;
    res->t_commit_ms = ps.t_commit;
    res->t_gp_construct_ms = ps.t_construct;
    res->t_gp_prove_ms = ps.t_gp;
    res->t_eval_ms = ps.t_eval;
    res->t_open_ms = ps.t_open;
    res->t_worker_ms = ps.t_total;
    res->bytes_star_up = ps.star_up;
    res->bytes_star_down = ps.star_down;
    res->bytes_ring = ps.ring_bytes;
    res->star_messages = ps.star_msgs;
    res->t_hub_wait_ms = hub_wait;
    res->hub_exchanges = hub_n;
    h->last_proof = proof.serialize();
    res->proof_len = h->last_proof.size();
    Sha256 s;
    s.update(h->last_proof.data(), h->last_proof.size());
    s.final(res->proof_digest);
    return COZK_OK;
}

// raw copy between any two pointers (device or host) on the context's stream, synchronous: lets a host
// transport stage ring payloads in its own buffers (torch tensors) without knowing the engine's types
int cozk_copy(cozk_ctx* ctx, void* dst, const void* src, size_t nbytes) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && (nbytes == 0 || (dst && src)), "cozk_copy: bad argument");
        if (nbytes) {
            HIP_TRY(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDefault, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
    });
}

// serialized proof of the last prove (cozk_harness_result.proof_len bytes)
int cozk_harness_proof_bytes(const cozk_harness* h, uint8_t* out, size_t cap) {
    if (!h || !out || cap < h->last_proof.size()) return COZK_ERR_INVALID_ARG;
    memcpy(out, h->last_proof.data(), h->last_proof.size());
    return COZK_OK;
}

// context of party p (profiling hooks: cozk_prof_enable / cozk_prof_read)
cozk_ctx* cozk_harness_ctx(cozk_harness* h, int party) {
    if (!h || party < 0 || party >= (int)h->parties.size()) return nullptr;
    return h->parties[party].ctx;
}

}  // extern "C"

// --------------------------------------------------------------------------- wire format (host only)
extern "C" {
int cozk_wire_g1_encode(const uint64_t xy[8], int infinity, uint8_t out[64]) {
    if (!xy || !out) return COZK_ERR_INVALID_ARG;
    try {
        Writer w;
        w.g1(abi_to_g1(xy, infinity));
        memcpy(out, w.b.data(), 64);
        return COZK_OK;
    } catch (const std::exception&) {
        return COZK_ERR_INTERNAL;
    }
}
int cozk_wire_g1_decode(const uint8_t in[64], uint64_t xy[8], int* infinity) {
    if (!in || !xy || !infinity) return COZK_ERR_INVALID_ARG;
    try {
        Bytes b(in, in + 64);
        Reader rd(b);
        g1_affine p = rd.g1();
        *infinity = G1::is_inf(p) ? 1 : 0;
        fe_to_u64x4(p.x, xy);
        fe_to_u64x4(p.y, xy + 4);
        return COZK_OK;
    } catch (const std::exception&) {
        return COZK_ERR_INVALID_ARG;
    }
}
}  // extern "C"

// --------------------------------------------------------------------------- worker drivers over host nets
// The C++ round loops of csrc/host/prover.hpp with the host's own transport plugged in (cozk_star_net /
// cozk_ring_net): what a Rust host calls when it wants the loop, not just the per-round kernels.
static WorkerEnv make_env(cozk_ctx* ctx, const cozk_worker_params* wp, StarNetWorker* star, RingNet* ring) {
    WorkerEnv env;
    env.ctx = ctx;
    env.mode = wp->mode;
    env.party = wp->party;
    env.star = star;
    env.ring = ring;
    env.set_keys(wp->key_self, wp->key_prev);
    env.mask_ctr = wp->mask_counter;
    return env;
}
static void copy_out_fr(const std::vector<fe>& v, uint64_t* out, size_t cap, size_t* n_out, const char* what) {
    if (n_out) *n_out = v.size();
    if (!out) return;
    if (v.size() > cap) throw CozkError(COZK_ERR_INVALID_ARG, std::string(what) + ": output buffer too small");
    for (size_t i = 0; i < v.size(); i++) fe_to_u64x4(v[i], out + 4 * i);
}

extern "C" {

int cozk_worker_prove_grand_product(cozk_ctx* ctx, const cozk_worker_params* wp, const cozk_star_net* star, const cozk_ring_net* ring,
                                    cozk_layer* leaves, size_t batch_size, uint64_t* out_r, size_t r_cap, size_t* out_r_len) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && wp && star && leaves && (wp->mode == COZK_MODE_PLAIN || ring), "worker_prove_grand_product: bad argument");
        CallbackStarWorker sw(*star);
        std::unique_ptr<CallbackRingNet> rn(ring ? new CallbackRingNet(*ring) : nullptr);
        WorkerEnv env = make_env(ctx, wp, &sw, rn.get());
        cozk_layer* lv = nullptr;
        rc_check(cozk_layer_clone(ctx, leaves, &lv), ctx, "layer_clone");
        Rep3BatchedDenseGrandProduct gp = Rep3BatchedDenseGrandProduct::construct(env, LayerH(lv), batch_size);
        std::vector<fe> r = gp.prove_grand_product_worker(env);
        copy_out_fr(r, out_r, r_cap, out_r_len, "worker_prove_grand_product");
    });
}

int cozk_worker_prove_arbitrary(cozk_ctx* ctx, const cozk_worker_params* wp, const cozk_star_net* star, cozk_poly* const* polys, size_t m,
                                int combined_degree, const uint64_t claim[4], int num_rounds, uint64_t* out_r, uint64_t* out_final_evals) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && wp && star && polys && claim && m >= 1, "worker_prove_arbitrary: bad argument");
        CallbackStarWorker sw(*star);
        WorkerEnv env = make_env(ctx, wp, &sw, nullptr);
        std::vector<cozk_poly*> ps(polys, polys + m);
        ArbitraryResult res = prove_arbitrary_worker(env, fe_from_u64x4(claim), num_rounds, ps, combined_degree);
        copy_out_fr(res.r, out_r, (size_t)num_rounds, nullptr, "worker_prove_arbitrary");
        copy_out_fr(res.final_evals, out_final_evals, m, nullptr, "worker_prove_arbitrary");
    });
}

int cozk_worker_spartan_first_sumcheck(cozk_ctx* ctx, const cozk_worker_params* wp, const cozk_star_net* star, cozk_poly* za, cozk_poly* zb,
                                       cozk_poly* zc, cozk_poly* eq, uint64_t* out_point, uint64_t out_finals[16]) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && wp && star && za && zb && zc && eq, "worker_spartan_first_sumcheck: bad argument");
        CallbackStarWorker sw(*star);
        WorkerEnv env = make_env(ctx, wp, &sw, nullptr);
        std::vector<fe> finals;
        std::vector<fe> pt = rep3_first_sumcheck_worker(env, za, zb, zc, eq, finals);
        copy_out_fr(pt, out_point, pt.size(), nullptr, "spartan_first");
        copy_out_fr(finals, out_finals, 4, nullptr, "spartan_first");
    });
}

int cozk_worker_spartan_second_sumcheck(cozk_ctx* ctx, const cozk_worker_params* wp, const cozk_star_net* star, cozk_poly* z, cozk_poly* a,
                                        cozk_poly* b, cozk_poly* c, const uint64_t coef[12], uint64_t* out_point, uint64_t out_finals[16]) {
    return cozk_guard(ctx, [&] {
        COZK_REQUIRE(ctx && wp && star && z && a && b && c && coef, "worker_spartan_second_sumcheck: bad argument");
        CallbackStarWorker sw(*star);
        WorkerEnv env = make_env(ctx, wp, &sw, nullptr);
        fe cf[3] = {fe_from_u64x4(coef), fe_from_u64x4(coef + 4), fe_from_u64x4(coef + 8)};
        std::vector<fe> finals;
        std::vector<fe> pt = rep3_second_sumcheck_worker(env, z, a, b, c, cf, finals);
        copy_out_fr(pt, out_point, pt.size(), nullptr, "spartan_second");
        copy_out_fr(finals, out_finals, 4, nullptr, "spartan_second");
    });
}

}  // extern "C"

// --------------------------------------------------------------------------- co-noir-spartan harness (config 4)
#include "host/spartan_harness.hpp"
#include "host/lookups_harness.hpp"
#include "host/outer_harness.hpp"
#include "host/flow_harness.hpp"
