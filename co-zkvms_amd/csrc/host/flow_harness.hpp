// ONE chained co-jolt worker flow: JoltRep3Prover::prove (co-jolt/src/jolt/vm/jolt/worker.rs:175-266) against its
// coordinator (jolt/vm/jolt/coordinator.rs:118-222) under ONE transcript and ONE Rep3ProverOpeningAccumulator, every polynomial
// committed once:
//
//   1 commit-all                                  Rep3JoltPolynomials::commit, jolt/vm/jolt/witness.rs:304-382 (PST13 batch commit)
//   2 bytecode memory checking                    lasso/memory_checking/worker.rs:40-237 + jolt/vm/bytecode/worker.rs:43-142
//   3 instruction lookups                         jolt/vm/instruction_lookups/worker.rs:95-176: primary sumcheck, then memory checking with
//                                                 the TOGGLED read / write grand product and the dense init / final one (:742-860)
//   4 read-write memory checking + output check   jolt/vm/read_write_memory/worker.rs:54-180 (prove_outputs), :196-344 (leaves)
//   5 Spartan                                     r1cs/spartan/worker.rs:63-273 on the reference's constraint set (spartan_jolt.hpp)
//   6 reduce_and_prove over ALL openings          poly/opening_proof.rs:181-291 -> one PST13 opening
//
// Every grand product's leaves are K11 fingerprints of COMMITTED columns (cozk_fingerprint_leaves) with the reference's formulas.
// The witness is synthetic but Jolt-shaped: the 78 R1CS inputs of a satisfying synthetic trace (jolt_r1cs.hpp) double as
// the bytecode / register / RAM / lookup columns exactly as inputs.rs:269-300 maps them; read_cts / E / final_cts per memory, the
// timestamps, v_init / v_final are seeded streams.  NOT part of the flow (out of this path's scope): party 0's public
// TimestampValidityProof (jolt-core, read_write_memory/worker.rs:78-104) and the verifier's multiset-equality check of the hashes (the
// synthetic counters are not a consistent offline-memory-checking instance; every sumcheck, GKR layer, fingerprint-vs-opening relation and
// the batched PST13 opening ARE verified).  The Lasso lookup_outputs polynomial is generated from the E polynomials (sum_i flag_i g_i(E)),
// independently of the R1CS LookupOutput column (a real trace identifies the two).  oracle/pyflow.py restates the flow; proofs are compared
// byte for byte.
#pragma once
#include <functional>

struct FlowIdx {
    int n_mem = 0;
    int r1cs = 0, bc_t_read = 0, rw_t_read = 0, read_cts = 0, E = 0, lasso_out = 0, final_cts = 0, bc_t_final = 0, rw_v_init = 0, rw_v_final = 0, rw_t_final = 0,
        count = 0;
    explicit FlowIdx(int nm = 0) : n_mem(nm) {
        r1cs = 0;
        bc_t_read = jolt::NUM_INPUTS;
        rw_t_read = bc_t_read + 1;
        read_cts = rw_t_read + 4;
        E = read_cts + nm;
        lasso_out = E + nm;
        final_cts = lasso_out + 1;
        bc_t_final = final_cts + nm;
        rw_v_init = bc_t_final + 1;
        rw_v_final = rw_v_init + 1;
        rw_t_final = rw_v_final + 1;
        count = rw_t_final + 1;
    }
};

struct FlowParty {
    cozk_ctx* ctx = nullptr;
    int party = 0;
    std::vector<PolyH> polys;       // evaluation view (REP3 shares, or PLAIN Fr values for public polynomials), commit order
    std::vector<VecH> commit_vecs;  // what the MSM and the fingerprints consume: share-a view, or the compact public column
    std::vector<VecH> msm_vecs;     // PLAIN mode: a narrow (U32 / U64) copy of a secret column whose values fit, for the MSM only
                                    // (msm_field_elements' dispatch on the scalars' bit length); empty handle = commit_vecs[i]
    std::unique_ptr<PST13Setup> setup;
    VecH iota;                         // 0, 1, 2, .. as U32 (identity / index columns of the leaves)
    std::vector<VecH> bc_table;        // the bytecode table (preprocessing.v_init_final): 6 compact columns of B entries
    std::vector<VecH> subtables;       // materialized subtables: U32 columns of M entries
    std::vector<VecH> mem_flags;       // memory_flag_indices as dense U8 columns
    VecH io_range, v_io;               // FR vectors of MEM entries (output check)
    double t_commit = 0, t_bytecode = 0, t_primary = 0, t_lookups_gp = 0, t_rw = 0, t_spartan = 0, t_open = 0, t_total = 0;
    SpartanTimes sp_times;
    uint64_t star_up = 0, star_down = 0, star_msgs = 0, ring_bytes = 0;
    std::string error;
};

struct cozk_flow {
    cozk_flow_config cfg;
    int nparties = 1;
    size_t N = 0, M = 0, B = 0, MEM = 0;
    FlowIdx ix;
    std::vector<FlowParty> parties;
    jolt::System sys;
    std::vector<cozk_primary_instr> instrs;
    // the dealer's / verifier's clear view
    std::vector<std::vector<fe>> clear;  // per committed polynomial
    std::vector<int> is_public, pub_bytes;
    std::vector<uint64_t> device_stream;  // != 0: the column is the seeded stream of this seed, generated on the device (no host copy)
    std::vector<int> device_bits;         // ... masked to this many bits (cfg.small_witness), 0 = full width
    std::vector<size_t> lens;
    std::vector<std::vector<uint64_t>> bc_table_clear, subtables_clear;
    std::vector<fe> io_range_clear, v_io_clear;
    std::string error;
    Bytes last_proof;
};

namespace {

struct MemCheckProof {
    std::vector<fe> rw_hashes, if_hashes;
    GrandProductProof rw, init_final;
    std::vector<fe> rw_claims, if_claims;
    void write(Writer& w) const {
        w.vec_fr(rw_hashes);
        w.vec_fr(if_hashes);
        for (const GrandProductProof* g : {&rw, &init_final}) {
            w.vec_fr(g->outputs);
            w.u64(g->gkr_layers.size());
            for (auto& l : g->gkr_layers) {
                w.u64(l.proof.compressed_polys.size());
                for (auto& p : l.proof.compressed_polys) w.vec_fr(p);
                w.fr(l.left_claim);
                w.fr(l.right_claim);
            }
        }
        w.vec_fr(rw_claims);
        w.vec_fr(if_claims);
    }
};

struct FlowProof {
    std::vector<PST13Commitment> commitments;
    MemCheckProof bytecode, lookups, rw;
    PrimarySumcheckProof primary;
    std::vector<fe> primary_claims;
    SumcheckProof outputs;
    std::vector<fe> outputs_claims;
    JoltSpartanProof spartan;
    ReducedOpeningProof reduced;
    Bytes serialize() const {
        Writer w;
        w.u64(commitments.size());
        for (auto& c : commitments) {
            w.u64(c.nv);
            w.g1(c.g_product);
        }
        bytecode.write(w);
        w.u64(primary.compressed_polys.size());
        for (auto& p : primary.compressed_polys) w.vec_fr(p);
        w.vec_fr(primary.openings);
        w.vec_fr(primary_claims);
        lookups.write(w);
        rw.write(w);
        w.u64(outputs.compressed_polys.size());
        for (auto& p : outputs.compressed_polys) w.vec_fr(p);
        w.vec_fr(outputs_claims);
        spartan.write(w);
        w.u64(reduced.sumcheck_proof.compressed_polys.size());
        for (auto& p : reduced.sumcheck_proof.compressed_polys) w.vec_fr(p);
        w.vec_fr(reduced.sumcheck_claims);
        w.vec_g1(reduced.joint_opening_proof);
        return w.b;
    }
};

inline bool flow_vec_eq(const std::vector<fe>& a, const std::vector<fe>& b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++)
        if (!Fr::eq(a[i], b[i])) return false;
    return true;
}
inline int flow_log2(size_t n) {
    int k = 0;
    while (((size_t)1 << k) < n) k++;
    return k;
}

// ------------------------------------------------------------------------------------------------ the dealer's view
void flow_build_clear(cozk_flow* h) {
    const cozk_flow_config& c = h->cfg;
    const uint64_t seed = c.seed;
    const FlowIdx& ix = h->ix;
    const size_t N = h->N, M = h->M, B = h->B, MEM = h->MEM;
    h->clear.assign((size_t)ix.count, {});
    h->is_public.assign((size_t)ix.count, 0);
    h->pub_bytes.assign((size_t)ix.count, 0);
    h->device_stream.assign((size_t)ix.count, 0);
    h->lens.assign((size_t)ix.count, 0);
    std::vector<std::vector<fe>> r1;
    jolt::build_clear(seed, N, r1);
    for (int v = 0; v < jolt::NUM_INPUTS; v++) {
        h->clear[(size_t)v] = std::move(r1[(size_t)v]);
        h->pub_bytes[(size_t)v] = jolt::public_bytes(v);
        h->is_public[(size_t)v] = h->pub_bytes[(size_t)v] ? 1 : 0;
    }
    auto pub = [&](int idx, uint64_t off, size_t n, int bits) {
        std::vector<fe>& col = h->clear[(size_t)idx];
        col.resize(n);
        for (size_t i = 0; i < n; i++) col[i] = Fr::from_u64(synthetic_small_host(seed + off, i, bits));
        h->is_public[(size_t)idx] = 1;
        h->pub_bytes[(size_t)idx] = 4;
    };
    // host threads over index ranges (the setup of a 2^20-cycle trace generates ~10^8 field elements)
    auto par_for = [&](size_t n, const std::function<void(size_t, size_t)>& body) {
        unsigned nt = std::thread::hardware_concurrency();
        nt = nt == 0 ? 1 : (nt > 16 ? 16 : nt);
        if (n < 4096 || nt == 1) {
            body(0, n);
            return;
        }
        std::vector<std::thread> ts;
        for (unsigned k = 0; k < nt; k++) ts.emplace_back([&, k] { body(n * k / nt, n * (k + 1) / nt); });
        for (auto& t : ts) t.join();
    };
    // cfg.small_witness: the widths a real trace gives these columns (0 = uniform field elements, see include/cozk.h)
    const int cnt_bits = c.small_witness ? c.log_n : 0, val_bits = c.small_witness ? 32 : 0;
    h->device_bits.assign((size_t)ix.count, 0);
    auto sh = [&](int idx, uint64_t s, size_t n, int bits) {
        std::vector<fe>& col = h->clear[(size_t)idx];
        col.resize(n);
        par_for(n, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) col[i] = synthetic_fr_host(s, i, bits);
        });
    };
    auto dev = [&](int idx, uint64_t s, size_t n, int bits) {
        h->device_stream[(size_t)idx] = s;
        h->device_bits[(size_t)idx] = bits;
        h->lens[(size_t)idx] = n;
    };
    pub(ix.bc_t_read, 31, N, 20);
    pub(ix.bc_t_final, 32, B, 20);
    for (int k = 0; k < 4; k++) pub(ix.rw_t_read + k, 41 + (uint64_t)k, N, 20);
    sh(ix.rw_v_init, seed + 45000, MEM, val_bits);
    sh(ix.rw_v_final, seed + 46000, MEM, val_bits);
    pub(ix.rw_t_final, 47, MEM, 20);
    for (int m = 0; m < c.n_mem; m++) {
        dev(ix.read_cts + m, seed + 11000ull * (uint64_t)(m + 1), N, cnt_bits);
        sh(ix.E + m, seed + 9000ull * (uint64_t)(m + 1), N, val_bits);
        dev(ix.final_cts + m, seed + 13000ull * (uint64_t)(m + 1), M, cnt_bits);
    }
    static const int tab_bits[6] = {20, 32, 6, 6, 6, 12};
    h->bc_table_clear.assign(6, std::vector<uint64_t>(B));
    for (int k = 0; k < 6; k++)
        for (size_t i = 0; i < B; i++) h->bc_table_clear[(size_t)k][i] = synthetic_small_host(seed + 33 + (uint64_t)k, i, tab_bits[k]);
    h->subtables_clear.assign((size_t)c.n_subtables, std::vector<uint64_t>(M));
    for (int s = 0; s < c.n_subtables; s++)
        for (size_t i = 0; i < M; i++) h->subtables_clear[(size_t)s][i] = synthetic_small_host(seed + 15000ull * (uint64_t)(s + 1), i, 32);
    // lookup_outputs = sum_i flag_i g_i(E) in the clear
    h->instrs = lookups_instr_table(c.n_mem);
    std::vector<fe>& outs = h->clear[(size_t)ix.lasso_out];
    outs.assign(N, Fr::zero());
    par_for(N, [&](size_t lo, size_t hi) {
        std::vector<fe> E((size_t)c.n_mem);
        for (size_t t = lo; t < hi; t++) {
            for (int i = 0; i < jolt::N_INSTR; i++) {
                if (Fr::is_zero(h->clear[(size_t)(jolt::V_INSTR + i)][t])) continue;
                const cozk_primary_instr& in = h->instrs[(size_t)i];
                for (int j = 0; j < in.n_mems; j++) E[(size_t)in.mems[j]] = h->clear[(size_t)(ix.E + in.mems[j])][t];
                outs[t] = primary_g_plain(in, E);
                break;
            }
        }
    });
    h->io_range_clear.assign(MEM, Fr::zero());
    h->v_io_clear.assign(MEM, Fr::zero());
    for (size_t i = MEM / 4; i < MEM / 2; i++) {
        h->io_range_clear[i] = Fr::one();
        h->v_io_clear[i] = h->clear[(size_t)ix.rw_v_final][i];
    }
    for (int i = 0; i < ix.count; i++)
        if (!h->device_stream[(size_t)i]) h->lens[(size_t)i] = h->clear[(size_t)i].size();
}

uint64_t flow_share_seed(const cozk_flow* h, int idx) {
    const FlowIdx& ix = h->ix;
    const uint64_t seed = h->cfg.seed;
    if (idx < jolt::NUM_INPUTS) return seed + 100ull * (uint64_t)(idx + 1);
    if (idx == ix.rw_v_init) return seed + 45000;
    if (idx == ix.rw_v_final) return seed + 46000;
    if (idx == ix.lasso_out) return seed + 555;
    if (idx >= ix.read_cts && idx < ix.E) return seed + 11000ull * (uint64_t)(idx - ix.read_cts + 1);
    if (idx >= ix.E && idx < ix.lasso_out) return seed + 9000ull * (uint64_t)(idx - ix.E + 1);
    return seed + 13000ull * (uint64_t)(idx - ix.final_cts + 1);
}

template <typename T>
VecH flow_upload_ints(cozk_ctx* ctx, const std::vector<uint64_t>& v, int kind) {
    std::vector<T> t(v.size());
    for (size_t i = 0; i < v.size(); i++) t[i] = (T)v[i];
    cozk_vec* d = nullptr;
    rc_check(cozk_vec_upload(ctx, t.data(), t.size(), kind, &d), ctx, "vec_upload(compact column)");
    return VecH(d);
}
VecH flow_upload_compact(cozk_ctx* ctx, const std::vector<uint64_t>& v, int bytes) {
    if (bytes == 1) return flow_upload_ints<uint8_t>(ctx, v, COZK_SCALAR_U8);
    if (bytes == 4) return flow_upload_ints<uint32_t>(ctx, v, COZK_SCALAR_U32);
    return flow_upload_ints<uint64_t>(ctx, v, COZK_SCALAR_U64);
}

void flow_setup_party(cozk_flow* h, FlowParty& ps) {
    const cozk_flow_config& c = h->cfg;
    cozk_ctx* ctx = ps.ctx;
    std::vector<fe> t((size_t)c.log_n);
    for (int i = 0; i < c.log_n; i++) t[(size_t)i] = synthetic_fr_host(c.seed ^ 0x7A7A7A7Aull, (uint64_t)i);
    ps.setup = PST13::setup(ctx, t, c.precompute);
    for (int idx = 0; idx < h->ix.count; idx++) {
        if (h->device_stream[(size_t)idx]) {  // a seeded stream: generated and shared on the device (harness.hip make_share_vectors)
            VecH a, b;
            make_share_vectors(ctx, h->lens[(size_t)idx], h->device_stream[(size_t)idx], ps.party, c.mode, a, b, h->device_bits[(size_t)idx]);
            cozk_poly* p = nullptr;
            rc_check(cozk_poly_create(ctx, c.mode, a.h, b.h, &p), ctx, "poly_create");
            ps.polys.push_back(PolyH(p));
            cozk_vec* view = nullptr;
            rc_check(cozk_poly_share_view(ctx, p, 0, &view), ctx, "share_view");
            ps.commit_vecs.push_back(VecH(view));
            ps.msm_vecs.emplace_back();
            const int bits = h->device_bits[(size_t)idx];
            if (c.mode == COZK_MODE_PLAIN && bits > 0 && bits <= 64) {  // masked by construction; cozk_vec_narrow checks it anyway
                cozk_vec* nv = nullptr;
                rc_check(cozk_vec_narrow(ctx, view, bits <= 32 ? COZK_SCALAR_U32 : COZK_SCALAR_U64, &nv), ctx, "vec_narrow");
                ps.msm_vecs.back() = VecH(nv);
            }
            continue;
        }
        const std::vector<fe>& col = h->clear[(size_t)idx];
        cozk_vec* pv = nullptr;
        rc_check(cozk_vec_upload(ctx, col.data(), col.size(), COZK_SCALAR_FR, &pv), ctx, "vec_upload(column)");
        VecH plain(pv);
        cozk_poly* p = nullptr;
        if (h->is_public[(size_t)idx]) {
            rc_check(cozk_poly_create(ctx, COZK_MODE_PLAIN, plain.h, nullptr, &p), ctx, "poly_create");
            ps.polys.push_back(PolyH(p));
            std::vector<uint64_t> ints(col.size());
            for (size_t i = 0; i < col.size(); i++) {
                fe v = Fr::from_mont(col[i]);
                ints[i] = (uint64_t)v.l[0] | ((uint64_t)v.l[1] << 32);
            }
            ps.commit_vecs.push_back(flow_upload_compact(ctx, ints, h->pub_bytes[(size_t)idx]));
            ps.msm_vecs.emplace_back();
        } else {
            if (c.mode == COZK_MODE_PLAIN) {
                rc_check(cozk_poly_create(ctx, COZK_MODE_PLAIN, plain.h, nullptr, &p), ctx, "poly_create");
            } else {
                uint8_t k0[COZK_PRF_KEY_BYTES], k1[COZK_PRF_KEY_BYTES];
                const uint64_t s = flow_share_seed(h, idx);
                harness_prf_key(s, 101, k0);
                harness_prf_key(s, 102, k1);
                cozk_vec *sa = nullptr, *sb = nullptr;
                rc_check(cozk_rep3_share_vec(ctx, plain.h, k0, k1, 0, ps.party, &sa, &sb), ctx, "rep3_share_vec");
                VecH a(sa), b(sb);
                rc_check(cozk_poly_create(ctx, COZK_MODE_REP3, a.h, b.h, &p), ctx, "poly_create");
            }
            ps.polys.push_back(PolyH(p));
            cozk_vec* view = nullptr;
            rc_check(cozk_poly_share_view(ctx, p, 0, &view), ctx, "share_view");
            ps.commit_vecs.push_back(VecH(view));
            ps.msm_vecs.emplace_back();
            if (c.mode == COZK_MODE_PLAIN) {  // a secret column of a plain prover: as wide as its values (the host has them in the clear)
                uint32_t hi32 = 0, hi64 = 0;
                std::vector<uint64_t> ints(col.size());
                for (size_t i = 0; i < col.size(); i++) {
                    const fe v = Fr::from_mont(col[i]);
                    hi32 |= v.l[1];
                    for (int k = 2; k < 8; k++) hi64 |= v.l[k];
                    ints[i] = (uint64_t)v.l[0] | ((uint64_t)v.l[1] << 32);
                }
                if (!hi64) ps.msm_vecs.back() = flow_upload_compact(ctx, ints, (hi32 | hi64) ? 8 : 4);
            }
        }
    }
    size_t imax = std::max(std::max(h->N, h->M), std::max(h->B, h->MEM));
    std::vector<uint64_t> io(imax);
    for (size_t i = 0; i < imax; i++) io[i] = i;
    ps.iota = flow_upload_compact(ctx, io, 4);
    static const int tab_bytes[6] = {4, 8, 1, 1, 1, 4};
    for (int k = 0; k < 6; k++) ps.bc_table.push_back(flow_upload_compact(ctx, h->bc_table_clear[(size_t)k], tab_bytes[k]));
    for (auto& s : h->subtables_clear) ps.subtables.push_back(flow_upload_compact(ctx, s, 4));
    // memory_flag_indices: memory m is live at step t iff the step's instruction uses it
    for (int m = 0; m < c.n_mem; m++) {
        std::vector<uint64_t> col(h->N, 0);
        for (int i = 0; i < jolt::N_INSTR; i++) {
            const cozk_primary_instr& in = h->instrs[(size_t)i];
            bool uses = false;
            for (int j = 0; j < in.n_mems; j++) uses |= in.mems[j] == m;
            if (!uses) continue;
            const std::vector<fe>& fl = h->clear[(size_t)(jolt::V_INSTR + i)];
            for (size_t t2 = 0; t2 < h->N; t2++)
                if (!Fr::is_zero(fl[t2])) col[t2] = 1;
        }
        ps.mem_flags.push_back(flow_upload_compact(ctx, col, 1));
    }
    cozk_vec* v = nullptr;
    rc_check(cozk_vec_upload(ctx, h->io_range_clear.data(), h->MEM, COZK_SCALAR_FR, &v), ctx, "vec_upload");
    ps.io_range = VecH(v);
    rc_check(cozk_vec_upload(ctx, h->v_io_clear.data(), h->MEM, COZK_SCALAR_FR, &v), ctx, "vec_upload");
    ps.v_io = VecH(v);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
}

// ------------------------------------------------------------------------------------------------ worker
struct LeafTerm {
    const cozk_vec* col = nullptr;    // a compact public column ...
    const cozk_poly* poly = nullptr;  // ... or a polynomial (shared, or public as Fr)
    fe coeff;
};
struct LeafBuf {
    VecH a, b;
};

LeafBuf flow_leaf_alloc(WorkerEnv& env, size_t n) {
    LeafBuf lb;
    cozk_vec* v = nullptr;
    rc_check(cozk_vec_alloc(env.ctx, n, COZK_SCALAR_FR, &v), env.ctx, "vec_alloc(leaves)");
    lb.a = VecH(v);
    if (env.mode == COZK_MODE_REP3) {
        rc_check(cozk_vec_alloc(env.ctx, n, COZK_SCALAR_FR, &v), env.ctx, "vec_alloc(leaves)");
        lb.b = VecH(v);
    }
    return lb;
}
void flow_fingerprint(WorkerEnv& env, const std::vector<LeafTerm>& terms, const fe& constant, LeafBuf& out, size_t offset, size_t n) {
    std::vector<const cozk_vec*> cols;
    std::vector<const cozk_poly*> polys;
    std::vector<fe> cc, pc;
    for (auto& t : terms) {
        if (t.col) {
            cols.push_back(t.col);
            cc.push_back(t.coeff);
        } else {
            polys.push_back(t.poly);
            pc.push_back(t.coeff);
        }
    }
    std::vector<uint64_t> ccw = to_abi(cc), pcw = to_abi(pc);
    uint64_t cst[4];
    fe_to_u64x4(constant, cst);
    rc_check(cozk_fingerprint_leaves(env.ctx, cols.data(), ccw.data(), cols.size(), polys.data(), pcw.data(), polys.size(), cst, env.mode, env.party, out.a.h,
                                     out.b.h, offset, n),
             env.ctx, "fingerprint_leaves");
}
LayerH flow_leaves_to_layer(WorkerEnv& env, LeafBuf& lb) {
    cozk_layer* l = nullptr;
    rc_check(cozk_layer_create(env.ctx, env.mode, lb.a.h, lb.b.h, 1, &l), env.ctx, "layer_create");
    return LayerH(l);
}

// batch_evaluate + append of one opening group (compute_openings, lasso/memory_checking/worker.rs:132-176)
void flow_open(WorkerEnv& env, Rep3ProverOpeningAccumulator& acc, const std::vector<cozk_poly*>& polys, const std::vector<fe>& point) {
    std::vector<uint64_t> rr = to_abi(point);
    cozk_vec* chi = nullptr;
    rc_check(cozk_eq_evals(env.ctx, rr.data(), (int)point.size(), &chi), env.ctx, "eq_evals");
    VecH chih(chi);
    std::vector<fe> claims = spartan_batch_evaluate(env, polys, chih.h);
    acc.append(env, polys, chih.h, point, claims);
}

// prove_memory_checking after compute_leaves (lasso/memory_checking/worker.rs:40-127): construct both circuits, send the hashes,
// prove both, return (r_read_write_opening, r_init_final_opening).  toggle != null: the read / write circuit is toggled.
void flow_memory_checking(WorkerEnv& env, LayerH rw_leaves, size_t rw_batch, ToggleH toggle, LayerH if_leaves, size_t if_batch, std::vector<fe>& r_rw_open,
                          std::vector<fe>& r_if_open) {
    Rep3ToggledBatchedGrandProduct tgp;
    Rep3BatchedDenseGrandProduct rw;
    std::vector<fe> rw_hashes;
    if (toggle.h) {
        rw_batch = cozk_toggle_batch(toggle.h);
        tgp = Rep3ToggledBatchedGrandProduct::construct(env, std::move(toggle));
        rw_hashes = tgp.sparse_layers.claimed_outputs(env);
    } else {
        rw = Rep3BatchedDenseGrandProduct::construct(env, std::move(rw_leaves), rw_batch);
        rw_hashes = rw.claimed_outputs(env);
    }
    Rep3BatchedDenseGrandProduct inf = Rep3BatchedDenseGrandProduct::construct(env, std::move(if_leaves), if_batch);
    std::vector<fe> if_hashes = inf.claimed_outputs(env);
    {
        Writer w;
        w.vec_fr(rw_hashes);
        w.vec_fr(if_hashes);
        env.star->send_response(w.b);
    }
    std::vector<fe> r_rw = tgp.toggle_layer.h ? tgp.prove_grand_product_worker(env) : rw.prove_grand_product_worker(env);
    std::vector<fe> r_if = inf.prove_grand_product_worker(env);
    r_rw_open.assign(r_rw.begin() + flow_log2(rw_batch), r_rw.end());
    r_if_open.assign(r_if.begin() + flow_log2(if_batch), r_if.end());
}

void flow_worker_main(cozk_flow* h, FlowParty& ps, StarNetWorker* star, RingNet* ring) {
    const cozk_flow_config& c = h->cfg;
    const FlowIdx& ix = h->ix;
    const size_t N = h->N, M = h->M, B = h->B, MEM = h->MEM;
    WorkerEnv env;
    env.ctx = ps.ctx;
    env.mode = c.mode;
    env.party = ps.party;
    env.star = star;
    env.ring = ring;
    harness_prf_key(c.seed, (uint64_t)ps.party, env.key_self);
    harness_prf_key(c.seed, (uint64_t)((ps.party + 2) % 3), env.key_prev);
    HIP_TRY(hipSetDevice(ps.ctx->device));
    auto P = [&](int idx) { return ps.polys[(size_t)idx].h; };
    auto CV = [&](int idx) { return (const cozk_vec*)ps.commit_vecs[(size_t)idx].h; };
    // a committed polynomial as a leaf term: public ones through their compact column, shared ones through their shares
    auto T = [&](int idx, const fe& coeff) {
        LeafTerm t;
        if (h->is_public[(size_t)idx]) t.col = CV(idx);
        else t.poly = P(idx);
        t.coeff = coeff;
        return t;
    };
    auto TC = [&](const VecH& col, const fe& coeff) {
        LeafTerm t;
        t.col = col.h;
        t.coeff = coeff;
        return t;
    };
    auto gammas = [&](int k, std::vector<fe>& g, fe& tau) {  // receive (gamma, tau); g[j] = gamma^j
        Bytes req = star->receive_request();
        Reader rd(req);
        fe gamma = rd.fr();
        tau = rd.fr();
        g.assign(1, Fr::one());
        for (int j = 0; j < k; j++) g.push_back(Fr::mul(g.back(), gamma));
    };
    const fe one = Fr::one();
    double t0 = now_ms();
    // ---- 1. commit-all: every party MSMs every polynomial; only P0's public commitments are kept (pst13.rs:165-229)
    {
        std::vector<cozk_vec*> vs;
        for (size_t i = 0; i < ps.commit_vecs.size(); i++) vs.push_back(ps.msm_vecs[i].h ? ps.msm_vecs[i].h : ps.commit_vecs[i].h);
        std::vector<PST13Commitment> cm = PST13::batch_commit(ps.ctx, *ps.setup, vs);
        Writer w;
        put_commitments(w, cm, h->is_public, ps.party);
        star->send_response(w.b);
    }
    double t1 = now_ms();
    ps.t_commit = t1 - t0;
    Rep3ProverOpeningAccumulator acc;
    std::vector<fe> g, r_rw, r_if;
    fe tau;
    // ---- 2. bytecode (bytecode/worker.rs:43-142): read = a g + v_address g^2 + .. + v_rs2 g^6 + t g^7 + imm - tau, write = read + g^7
    {
        gammas(7, g, tau);
        static const int bcv[5] = {jolt::V_ELF, jolt::V_BITFLAGS, jolt::V_BC_RD, jolt::V_BC_RS1, jolt::V_BC_RS2};
        std::vector<LeafTerm> read = {T(jolt::V_BYTECODE_A, g[1])};
        for (int k = 0; k < 5; k++) read.push_back(T(bcv[k], g[(size_t)(2 + k)]));
        read.push_back(T(ix.bc_t_read, g[7]));
        read.push_back(T(jolt::V_IMM, one));
        LeafBuf rw = flow_leaf_alloc(env, 2 * N), inf = flow_leaf_alloc(env, 2 * B);
        flow_fingerprint(env, read, Fr::neg(tau), rw, 0, N);
        flow_fingerprint(env, read, Fr::sub(g[7], tau), rw, N, N);
        std::vector<LeafTerm> init = {TC(ps.iota, g[1])};
        for (int k = 0; k < 5; k++) init.push_back(TC(ps.bc_table[(size_t)k], g[(size_t)(2 + k)]));
        init.push_back(TC(ps.bc_table[5], one));
        flow_fingerprint(env, init, Fr::neg(tau), inf, 0, B);
        init.push_back(T(ix.bc_t_final, g[7]));
        flow_fingerprint(env, init, Fr::neg(tau), inf, B, B);
        flow_memory_checking(env, flow_leaves_to_layer(env, rw), 2, ToggleH(), flow_leaves_to_layer(env, inf), 2, r_rw, r_if);
        std::vector<cozk_poly*> rwp = {P(jolt::V_BYTECODE_A)};
        for (int k = 0; k < 5; k++) rwp.push_back(P(bcv[k]));
        rwp.push_back(P(jolt::V_IMM));
        rwp.push_back(P(ix.bc_t_read));
        flow_open(env, acc, rwp, r_rw);
        flow_open(env, acc, {P(ix.bc_t_final)}, r_if);
    }
    double t2 = now_ms();
    ps.t_bytecode = t2 - t1;
    // ---- 3. instruction lookups (instruction_lookups/worker.rs:95-176)
    std::vector<cozk_poly*> iflag_polys, E_polys, rc_polys, fc_polys, dims;
    for (int i = 0; i < jolt::N_INSTR; i++) iflag_polys.push_back(P(jolt::V_INSTR + i));
    for (int m = 0; m < c.n_mem; m++) {
        E_polys.push_back(P(ix.E + m));
        rc_polys.push_back(P(ix.read_cts + m));
        fc_polys.push_back(P(ix.final_cts + m));
    }
    for (int i = 0; i < 4; i++) dims.push_back(P(jolt::V_QUERY + i));
    {
        Bytes req = star->receive_request();
        Reader rd(req);
        std::vector<fe> r_eq = rd.vec_fr();
        std::vector<uint64_t> wabi = to_abi(r_eq);
        cozk_vec* eqv = nullptr;
        rc_check(cozk_eq_evals(env.ctx, wabi.data(), (int)r_eq.size(), &eqv), env.ctx, "eq_evals");
        VecH eqh(eqv);
        std::vector<const cozk_vec*> fl;
        for (int i = 0; i < jolt::N_INSTR; i++) fl.push_back(CV(jolt::V_INSTR + i));
        std::vector<const cozk_poly*> Ec(E_polys.begin(), E_polys.end());
        cozk_primary* pr = nullptr;
        rc_check(cozk_primary_create(env.ctx, c.mode, ps.party, h->instrs.data(), h->instrs.size(), fl.data(), Ec.data(), (size_t)c.n_mem, P(ix.lasso_out), eqh.h, &pr),
                 env.ctx, "primary_create");
        PrimaryH prh(pr);
        std::vector<fe> rs = prove_primary_sumcheck_worker(env, pr, c.log_n, (size_t)c.n_mem, h->instrs.size());
        std::vector<fe> r_primary(rs.rbegin(), rs.rend());
        std::vector<cozk_poly*> pp(E_polys);
        pp.insert(pp.end(), iflag_polys.begin(), iflag_polys.end());
        pp.push_back(P(ix.lasso_out));
        flow_open(env, acc, pp, r_primary);
    }
    double t3 = now_ms();
    ps.t_primary = t3 - t2;
    {
        gammas(2, g, tau);
        LeafBuf rw = flow_leaf_alloc(env, 2 * (size_t)c.n_mem * N), inf = flow_leaf_alloc(env, (size_t)(c.n_subtables + c.n_mem) * M);
        for (int m = 0; m < c.n_mem; m++) {  // read = t g^2 + v g + a - tau, write = read + g^2 (:776-800)
            std::vector<LeafTerm> terms = {T(ix.read_cts + m, g[2]), T(ix.E + m, g[1]), T(jolt::V_QUERY + m % 4, one)};
            flow_fingerprint(env, terms, Fr::neg(tau), rw, (size_t)(2 * m) * N, N);
            flow_fingerprint(env, terms, Fr::sub(g[2], tau), rw, (size_t)(2 * m + 1) * N, N);
        }
        size_t off = 0;
        for (int s = 0; s < c.n_subtables; s++) {  // per subtable: init, then the finals of its memories (:803-833)
            std::vector<LeafTerm> init = {TC(ps.subtables[(size_t)s], g[1]), TC(ps.iota, one)};
            flow_fingerprint(env, init, Fr::neg(tau), inf, off, M);
            off += M;
            for (int m = 0; m < c.n_mem; m++) {
                if (m % c.n_subtables != s) continue;
                std::vector<LeafTerm> fin(init);
                fin.push_back(T(ix.final_cts + m, g[2]));
                flow_fingerprint(env, fin, Fr::neg(tau), inf, off, M);
                off += M;
            }
        }
        std::vector<const cozk_vec*> mf;
        for (auto& f : ps.mem_flags) mf.push_back(f.h);
        cozk_toggle* tg = nullptr;
        rc_check(cozk_toggle_create(env.ctx, c.mode, mf.data(), mf.size(), rw.a.h, rw.b.h, 1, &tg), env.ctx, "toggle_create");
        flow_memory_checking(env, LayerH(), 0, ToggleH(tg), flow_leaves_to_layer(env, inf), (size_t)(c.n_subtables + c.n_mem), r_rw, r_if);
        std::vector<cozk_poly*> rwp(dims);
        rwp.insert(rwp.end(), rc_polys.begin(), rc_polys.end());
        rwp.insert(rwp.end(), E_polys.begin(), E_polys.end());
        rwp.insert(rwp.end(), iflag_polys.begin(), iflag_polys.end());
        rwp.push_back(P(ix.lasso_out));
        flow_open(env, acc, rwp, r_rw);
        flow_open(env, acc, fc_polys, r_if);
    }
    double t4 = now_ms();
    ps.t_lookups_gp = t4 - t3;
    // ---- 4. read-write memory (read_write_memory/worker.rs:196-344, :109-180)
    {
        gammas(2, g, tau);
        struct Reg {
            int a, v_read, v_write, t;
        };
        static const Reg regs[4] = {{jolt::V_BC_RS1, jolt::V_RS1, jolt::V_RS1, 1}, {jolt::V_BC_RS2, jolt::V_RS2, jolt::V_RS2, 2}, {jolt::V_BC_RD, jolt::V_RD_READ, jolt::V_RD_WRITE, 0},
                                    {jolt::V_RAM_ADDR, jolt::V_RAM_READ, jolt::V_RAM_WRITE, 3}};
        LeafBuf rw = flow_leaf_alloc(env, 8 * N), inf = flow_leaf_alloc(env, 2 * MEM);
        for (int k = 0; k < 4; k++) {
            flow_fingerprint(env, {T(regs[k].v_read, g[1]), T(ix.rw_t_read + regs[k].t, g[2]), T(regs[k].a, one)}, Fr::neg(tau), rw, (size_t)(2 * k) * N, N);
            flow_fingerprint(env, {T(regs[k].v_write, g[1]), TC(ps.iota, g[2]), T(regs[k].a, one)}, Fr::neg(tau), rw, (size_t)(2 * k + 1) * N, N);
        }
        flow_fingerprint(env, {T(ix.rw_v_init, g[1]), TC(ps.iota, one)}, Fr::neg(tau), inf, 0, MEM);
        flow_fingerprint(env, {T(ix.rw_v_final, g[1]), T(ix.rw_t_final, g[2]), TC(ps.iota, one)}, Fr::neg(tau), inf, MEM, MEM);
        flow_memory_checking(env, flow_leaves_to_layer(env, rw), 8, ToggleH(), flow_leaves_to_layer(env, inf), 2, r_rw, r_if);
        std::vector<cozk_poly*> rwp = {P(jolt::V_RAM_ADDR), P(jolt::V_RD_READ), P(jolt::V_RS1), P(jolt::V_RS2), P(jolt::V_RAM_READ), P(jolt::V_RD_WRITE), P(jolt::V_RAM_WRITE)};
        for (int k = 0; k < 4; k++) rwp.push_back(P(ix.rw_t_read + k));
        rwp.push_back(P(jolt::V_BC_RD));
        rwp.push_back(P(jolt::V_BC_RS1));
        rwp.push_back(P(jolt::V_BC_RS2));
        flow_open(env, acc, rwp, r_rw);
        flow_open(env, acc, {P(ix.rw_v_final), P(ix.rw_t_final)}, r_if);
        // prove_outputs: sum_x eq(r_eq, x) io_range(x) (v_final(x) - v_io(x)) = 0, degree 3, HighToLow
        Bytes req = star->receive_request();
        Reader rd(req);
        std::vector<fe> r_eq = rd.vec_fr();
        std::vector<uint64_t> wabi = to_abi(r_eq);
        cozk_vec* eqv = nullptr;
        rc_check(cozk_eq_evals(env.ctx, wabi.data(), (int)r_eq.size(), &eqv), env.ctx, "eq_evals");
        VecH eqh(eqv);
        cozk_poly *pe = nullptr, *pr = nullptr, *pio = nullptr, *pd = nullptr;
        rc_check(cozk_poly_create(env.ctx, COZK_MODE_PLAIN, eqh.h, nullptr, &pe), env.ctx, "poly_create");
        PolyH peh(pe);
        rc_check(cozk_poly_create(env.ctx, COZK_MODE_PLAIN, ps.io_range.h, nullptr, &pr), env.ctx, "poly_create");
        PolyH prh(pr);
        rc_check(cozk_poly_create(env.ctx, COZK_MODE_PLAIN, ps.v_io.h, nullptr, &pio), env.ctx, "poly_create");
        PolyH pioh(pio);
        const cozk_poly* two[2] = {P(ix.rw_v_final), pioh.h};
        fe cf[2] = {one, Fr::neg(one)};
        std::vector<uint64_t> cfa = to_abi(std::vector<fe>(cf, cf + 2));
        rc_check(cozk_poly_linear_combination(env.ctx, two, cfa.data(), 2, env.mode, env.party, &pd), env.ctx, "v_final - v_io");
        PolyH pdh(pd);
        std::vector<cozk_poly*> sp = {peh.h, prh.h, pdh.h};
        ArbitraryResult ar = prove_arbitrary_worker(env, Fr::zero(), c.log_mem, sp, 3);
        flow_open(env, acc, {P(ix.rw_v_final)}, ar.r);
    }
    double t5 = now_ms();
    ps.t_rw = t5 - t4;
    // ---- 5. Spartan
    {
        std::vector<cozk_poly*> cols;
        for (int v = 0; v < jolt::NUM_INPUTS; v++) cols.push_back(P(v));
        ps.sp_times = SpartanTimes();
        prove_spartan_worker(env, h->sys, cols, N, acc, &ps.sp_times);
    }
    double t6 = now_ms();
    ps.t_spartan = t6 - t5;
    // ---- 6. one reduce_and_prove over everything
    acc.reduce_and_prove_worker(env, *ps.setup);
    double t7 = now_ms();
    ps.t_open = t7 - t6;
    ps.t_total = t7 - t0;
    ps.star_up = star->bytes_up;
    ps.star_down = star->bytes_down;
    ps.star_msgs = star->n_msgs;
    ps.ring_bytes = ring ? ring->bytes_sent : 0;
}

// ------------------------------------------------------------------------------------------------ coordinator
void flow_coordinate_memory_checking(StarNetCoordinator& net, Transcript& tr, MemCheckProof& mp, size_t rw_layers, bool toggled, size_t if_layers) {
    std::vector<std::vector<fe>> a, b;
    for (Bytes& m : net.receive_responses()) {
        Reader rd(m);
        a.push_back(rd.vec_fr());
        b.push_back(rd.vec_fr());
    }
    mp.rw_hashes = combine_additive(a);
    mp.if_hashes = combine_additive(b);
    tr.append_scalars(mp.rw_hashes);
    tr.append_scalars(mp.if_hashes);
    fe claim;
    std::vector<fe> r;
    if (toggled) mp.rw = coordinate_prove_toggled_grand_product(net, tr, rw_layers, r);
    else mp.rw = coordinate_prove_grand_product(net, tr, rw_layers, claim, r);
    mp.init_final = coordinate_prove_grand_product(net, tr, if_layers, claim, r);
    mp.rw_claims = Rep3ProverOpeningAccumulator::receive_claims(net, tr);
    mp.if_claims = Rep3ProverOpeningAccumulator::receive_claims(net, tr);
}

void flow_broadcast_gamma_tau(StarNetCoordinator& net, Transcript& tr) {
    fe gamma = tr.challenge_scalar(), tau = tr.challenge_scalar();
    Writer w;
    w.fr(gamma);
    w.fr(tau);
    net.broadcast_request(w.b);
}

void flow_coordinate(cozk_flow* h, StarNetCoordinator& net, FlowProof& proof) {
    const cozk_flow_config& c = h->cfg;
    Transcript tr("cozk-jolt");
    {
        std::vector<Bytes> msgs = net.receive_responses();
        std::vector<Reader> rds;
        for (auto& m : msgs) rds.emplace_back(m);
        proof.commitments = combine_commitments_from_parties(rds, h->nparties);
        for (auto& cm : proof.commitments) tr.append_point(cm.g_product);
    }
    flow_broadcast_gamma_tau(net, tr);
    flow_coordinate_memory_checking(net, tr, proof.bytecode, (size_t)c.log_n, false, (size_t)c.log_b);
    {
        std::vector<fe> r_eq = tr.challenge_vector((size_t)c.log_n), r_primary;
        Writer w;
        w.vec_fr(r_eq);
        net.broadcast_request(w.b);
        proof.primary = coordinate_primary_sumcheck(net, tr, c.log_n, r_primary, h->nparties, 0);
        proof.primary_claims = Rep3ProverOpeningAccumulator::receive_claims(net, tr);
    }
    flow_broadcast_gamma_tau(net, tr);
    flow_coordinate_memory_checking(net, tr, proof.lookups, (size_t)c.log_n + 1, true, (size_t)c.log_m);
    flow_broadcast_gamma_tau(net, tr);
    flow_coordinate_memory_checking(net, tr, proof.rw, (size_t)c.log_n, false, (size_t)c.log_mem);
    {
        std::vector<fe> r_eq = tr.challenge_vector((size_t)c.log_mem);
        Writer w;
        w.vec_fr(r_eq);
        net.broadcast_request(w.b);
        (void)coordinate_prove_arbitrary(net, tr, c.log_mem, proof.outputs);
        proof.outputs_claims = Rep3ProverOpeningAccumulator::receive_claims(net, tr);
    }
    proof.spartan = coordinate_spartan(net, tr, h->sys, h->N);
    std::vector<fe> r_red;
    fe rho, gamma;
    proof.reduced = Rep3ProverOpeningAccumulator::reduce_and_prove(net, tr, r_red, rho, gamma);
}

// ------------------------------------------------------------------------------------------------ plain verifier
struct FlowVOpen {
    std::vector<int> polys;  // indices into the commitments
    std::vector<fe> point, claims;
    fe rho;
};

fe flow_mle_host(const std::vector<fe>& vals, const std::vector<fe>& r) {
    std::vector<fe> eq = eq_evals_host(r);
    fe acc = Fr::zero();
    for (size_t i = 0; i < vals.size(); i++) acc = Fr::add(acc, Fr::mul(eq[i], vals[i]));
    return acc;
}
// the identity column 0, 1, 2, .. at a big-endian point
fe flow_iota_eval(const std::vector<fe>& r) {
    fe acc = Fr::zero();
    for (size_t j = 0; j < r.size(); j++) acc = Fr::add(Fr::dbl(acc), r[j]);
    return acc;
}
// MLE of circuit-major leaves at r = (r_hi over the circuits, padded to a power of two with `pad`) x r_lo
fe flow_batch_eval(const std::vector<fe>& per_circuit, const std::vector<fe>& r_hi, const fe& pad) {
    std::vector<fe> eq = eq_evals_host(r_hi);
    fe acc = Fr::zero();
    for (size_t cidx = 0; cidx < eq.size(); cidx++) acc = Fr::add(acc, Fr::mul(eq[cidx], cidx < per_circuit.size() ? per_circuit[cidx] : pad));
    return acc;
}

bool flow_verify(cozk_flow* h, const FlowProof& proof, std::string& why) {
    const cozk_flow_config& c = h->cfg;
    const FlowIdx& ix = h->ix;
    const size_t N = h->N, M = h->M, B = h->B, MEM = h->MEM;
    const fe one = Fr::one();
    Transcript vt("cozk-jolt");
    if ((int)proof.commitments.size() != ix.count) {
        why = "commitment count";
        return false;
    }
    for (auto& cm : proof.commitments) vt.append_point(cm.g_product);
    std::vector<FlowVOpen> opens;
    // one claim exchange (receive_claims): the batching challenge is drawn right after it (rho given: already drawn)
    auto take_claims = [&](const std::vector<int>& polys, const std::vector<fe>& point, const std::vector<fe>& claims, const fe* rho = nullptr) {
        FlowVOpen o;
        o.polys = polys;
        o.point = point;
        o.claims = claims;
        o.rho = rho ? *rho : vt.challenge_scalar();
        opens.push_back(o);
        return polys.size() == claims.size();
    };
    std::vector<fe> g;
    fe tau;
    auto gammas = [&](int k) {
        fe gamma = vt.challenge_scalar();
        tau = vt.challenge_scalar();
        g.assign(1, one);
        for (int j = 0; j < k; j++) g.push_back(Fr::mul(g.back(), gamma));
    };
    // dense memory checking: both GKR proofs; returns the final claims and points
    auto verify_dense = [&](const MemCheckProof& mp, bool toggled, size_t rw_batch, size_t if_batch, fe& rw_claim, fe& flag_claim, std::vector<fe>& r_rw, fe& if_claim,
                            std::vector<fe>& r_if) {
        vt.append_scalars(mp.rw_hashes);
        vt.append_scalars(mp.if_hashes);
        if (!flow_vec_eq(mp.rw.outputs, mp.rw_hashes) || !flow_vec_eq(mp.init_final.outputs, mp.if_hashes) || mp.rw_hashes.size() != rw_batch || mp.if_hashes.size() != if_batch) return false;
        if (toggled) {
            if (!verify_toggled_grand_product(mp.rw, vt, flag_claim, rw_claim, r_rw)) return false;
        } else if (!verify_grand_product(mp.rw, vt, rw_claim, r_rw)) {
            return false;
        }
        return verify_grand_product(mp.init_final, vt, if_claim, r_if);
    };
    auto split = [&](const std::vector<fe>& r, size_t batch, std::vector<fe>& hi, std::vector<fe>& lo) {
        int k = flow_log2(batch);
        hi.assign(r.begin(), r.begin() + k);
        lo.assign(r.begin() + k, r.end());
    };
    auto fr_u64 = [](uint64_t v) { return Fr::from_u64(v); };
    fe rw_claim, flag_claim, if_claim;
    std::vector<fe> r_rw, r_if, hi, lo, hi2, lo2;
    // ---- 2. bytecode
    {
        gammas(7);
        if (!verify_dense(proof.bytecode, false, 2, 2, rw_claim, flag_claim, r_rw, if_claim, r_if)) {
            why = "bytecode: grand products";
            return false;
        }
        split(r_rw, 2, hi, lo);
        split(r_if, 2, hi2, lo2);
        std::vector<int> rwp = {jolt::V_BYTECODE_A, jolt::V_ELF, jolt::V_BITFLAGS, jolt::V_BC_RD, jolt::V_BC_RS1, jolt::V_BC_RS2, jolt::V_IMM, ix.bc_t_read};
        if (!take_claims(rwp, lo, proof.bytecode.rw_claims) || !take_claims({ix.bc_t_final}, lo2, proof.bytecode.if_claims)) {
            why = "bytecode: claims";
            return false;
        }
        const std::vector<fe>& cl = proof.bytecode.rw_claims;
        fe read = Fr::sub(cl[6], tau);
        for (int k = 0; k < 6; k++) read = Fr::add(read, Fr::mul(g[(size_t)(1 + k)], cl[(size_t)k]));
        read = Fr::add(read, Fr::mul(g[7], cl[7]));
        if (!Fr::eq(flow_batch_eval({read, Fr::add(read, g[7])}, hi, Fr::zero()), rw_claim)) {
            why = "bytecode: read / write fingerprints != the GKR claim";
            return false;
        }
        std::vector<fe> initv(B);
        for (size_t i = 0; i < B; i++) {
            fe v = Fr::sub(Fr::add(fr_u64(h->bc_table_clear[5][i]), Fr::mul(g[1], fr_u64(i))), tau);
            for (int k = 0; k < 5; k++) v = Fr::add(v, Fr::mul(g[(size_t)(2 + k)], fr_u64(h->bc_table_clear[(size_t)k][i])));
            initv[i] = v;
        }
        fe init = flow_mle_host(initv, lo2);
        if (!Fr::eq(flow_batch_eval({init, Fr::add(init, Fr::mul(g[7], proof.bytecode.if_claims[0]))}, hi2, Fr::zero()), if_claim)) {
            why = "bytecode: init / final fingerprints != the GKR claim";
            return false;
        }
    }
    // ---- 3. instruction lookups
    {
        std::vector<fe> r_eq = vt.challenge_vector((size_t)c.log_n), rs;
        const int degree = primary_sumcheck_degree(h->instrs);
        if (!verify_primary_sumcheck(proof.primary, h->instrs, (size_t)c.n_mem, degree, r_eq, vt, rs)) {
            why = "primary sumcheck";
            return false;
        }
        if (!flow_vec_eq(proof.primary_claims, proof.primary.openings)) {
            why = "primary sumcheck: openings != the accumulator's claims";
            return false;
        }
        std::vector<int> pp;
        for (int m = 0; m < c.n_mem; m++) pp.push_back(ix.E + m);
        for (int i = 0; i < jolt::N_INSTR; i++) pp.push_back(jolt::V_INSTR + i);
        pp.push_back(ix.lasso_out);
        take_claims(pp, std::vector<fe>(rs.rbegin(), rs.rend()), proof.primary_claims);
        gammas(2);
        const size_t rw_batch = 2 * (size_t)c.n_mem, if_batch = (size_t)(c.n_subtables + c.n_mem);
        if (!verify_dense(proof.lookups, true, rw_batch, if_batch, rw_claim, flag_claim, r_rw, if_claim, r_if)) {
            why = "lookups: grand products";
            return false;
        }
        split(r_rw, rw_batch, hi, lo);
        split(r_if, if_batch, hi2, lo2);
        std::vector<int> rwp;
        for (int i = 0; i < 4; i++) rwp.push_back(jolt::V_QUERY + i);
        for (int m = 0; m < c.n_mem; m++) rwp.push_back(ix.read_cts + m);
        for (int m = 0; m < c.n_mem; m++) rwp.push_back(ix.E + m);
        for (int i = 0; i < jolt::N_INSTR; i++) rwp.push_back(jolt::V_INSTR + i);
        rwp.push_back(ix.lasso_out);
        std::vector<int> fcp;
        for (int m = 0; m < c.n_mem; m++) fcp.push_back(ix.final_cts + m);
        if (!take_claims(rwp, lo, proof.lookups.rw_claims) || !take_claims(fcp, lo2, proof.lookups.if_claims)) {
            why = "lookups: claims";
            return false;
        }
        const std::vector<fe>& cl = proof.lookups.rw_claims;
        const size_t o_rc = 4, o_E = 4 + (size_t)c.n_mem, o_fl = 4 + 2 * (size_t)c.n_mem;
        std::vector<fe> fps, fls;
        for (int m = 0; m < c.n_mem; m++) {
            fe read = Fr::sub(Fr::add(Fr::add(Fr::mul(g[2], cl[o_rc + (size_t)m]), Fr::mul(g[1], cl[o_E + (size_t)m])), cl[(size_t)(m % 4)]), tau);
            fps.push_back(read);
            fps.push_back(Fr::add(read, g[2]));
            fe fl = Fr::zero();  // the memory's flag = sum of the (one-hot) instruction flags that use it
            for (int i = 0; i < jolt::N_INSTR; i++) {
                const cozk_primary_instr& in = h->instrs[(size_t)i];
                bool uses = false;
                for (int j = 0; j < in.n_mems; j++) uses |= in.mems[j] == m;
                if (uses) fl = Fr::add(fl, cl[o_fl + (size_t)i]);
            }
            fls.push_back(fl);
            fls.push_back(fl);
        }
        if (!Fr::eq(flow_batch_eval(fls, hi, one), flag_claim) || !Fr::eq(flow_batch_eval(fps, hi, Fr::zero()), rw_claim)) {
            why = "lookups: toggle layer's flag / fingerprint claims != the opened values";
            return false;
        }
        std::vector<fe> leaves;
        fe io = flow_iota_eval(lo2);
        for (int s = 0; s < c.n_subtables; s++) {
            std::vector<fe> sv(M);
            for (size_t i = 0; i < M; i++) sv[i] = fr_u64(h->subtables_clear[(size_t)s][i]);
            fe init = Fr::sub(Fr::add(Fr::mul(g[1], flow_mle_host(sv, lo2)), io), tau);
            leaves.push_back(init);
            for (int m = 0; m < c.n_mem; m++)
                if (m % c.n_subtables == s) leaves.push_back(Fr::add(init, Fr::mul(g[2], proof.lookups.if_claims[(size_t)m])));
        }
        if (!Fr::eq(flow_batch_eval(leaves, hi2, Fr::zero()), if_claim)) {
            why = "lookups: init / final fingerprints != the GKR claim";
            return false;
        }
    }
    // ---- 4. read-write memory
    {
        gammas(2);
        if (!verify_dense(proof.rw, false, 8, 2, rw_claim, flag_claim, r_rw, if_claim, r_if)) {
            why = "read-write memory: grand products";
            return false;
        }
        split(r_rw, 8, hi, lo);
        split(r_if, 2, hi2, lo2);
        std::vector<int> rwp = {jolt::V_RAM_ADDR, jolt::V_RD_READ, jolt::V_RS1, jolt::V_RS2, jolt::V_RAM_READ, jolt::V_RD_WRITE, jolt::V_RAM_WRITE,
                                ix.rw_t_read, ix.rw_t_read + 1, ix.rw_t_read + 2, ix.rw_t_read + 3, jolt::V_BC_RD, jolt::V_BC_RS1, jolt::V_BC_RS2};
        if (!take_claims(rwp, lo, proof.rw.rw_claims) || !take_claims({ix.rw_v_final, ix.rw_t_final}, lo2, proof.rw.if_claims)) {
            why = "read-write memory: claims";
            return false;
        }
        const std::vector<fe>& cl = proof.rw.rw_claims;
        // claims: a_ram 0, v_read_rd 1, v_read_rs1 2, v_read_rs2 3, v_read_ram 4, v_write_rd 5, v_write_ram 6, t_read rd 7 rs1 8 rs2 9 ram 10, a_rd 11, a_rs1 12, a_rs2 13
        struct R4 {
            int a, vr, vw, t;
        };
        static const R4 regs[4] = {{12, 2, 2, 8}, {13, 3, 3, 9}, {11, 1, 5, 7}, {0, 4, 6, 10}};
        fe ident = flow_iota_eval(lo);
        std::vector<fe> leaves;
        for (int k = 0; k < 4; k++) {
            leaves.push_back(Fr::sub(Fr::add(Fr::add(Fr::mul(g[1], cl[(size_t)regs[k].vr]), Fr::mul(g[2], cl[(size_t)regs[k].t])), cl[(size_t)regs[k].a]), tau));
            leaves.push_back(Fr::sub(Fr::add(Fr::add(Fr::mul(g[1], cl[(size_t)regs[k].vw]), Fr::mul(g[2], ident)), cl[(size_t)regs[k].a]), tau));
        }
        if (!Fr::eq(flow_batch_eval(leaves, hi, Fr::zero()), rw_claim)) {
            why = "read-write memory: read / write fingerprints != the GKR claim";
            return false;
        }
        fe io = flow_iota_eval(lo2);
        fe init = Fr::sub(Fr::add(Fr::mul(g[1], flow_mle_host(h->clear[(size_t)ix.rw_v_init], lo2)), io), tau);  // VerifierComputedOpening of v_init
        fe fin = Fr::sub(Fr::add(Fr::add(Fr::mul(g[1], proof.rw.if_claims[0]), Fr::mul(g[2], proof.rw.if_claims[1])), io), tau);
        if (!Fr::eq(flow_batch_eval({init, fin}, hi2, Fr::zero()), if_claim)) {
            why = "read-write memory: init / final fingerprints != the GKR claim";
            return false;
        }
        std::vector<fe> r_eq = vt.challenge_vector((size_t)c.log_mem), r_out;
        fe claim = Fr::zero();
        if (!spartan_verify_rounds(proof.outputs, (size_t)c.log_mem, 3, claim, vt, r_out) || proof.outputs_claims.size() != 1) {
            why = "output check: shape";
            return false;
        }
        fe e = one;
        for (size_t i = 0; i < r_eq.size(); i++) e = Fr::mul(e, Fr::add(Fr::sub(Fr::sub(one, r_eq[i]), r_out[i]), Fr::dbl(Fr::mul(r_eq[i], r_out[i]))));
        fe want = Fr::mul(Fr::mul(e, flow_mle_host(h->io_range_clear, r_out)), Fr::sub(proof.outputs_claims[0], flow_mle_host(h->v_io_clear, r_out)));
        if (!Fr::eq(want, claim)) {
            why = "output check: final claim";
            return false;
        }
        take_claims({ix.rw_v_final}, r_out, proof.outputs_claims);
    }
    // ---- 5. Spartan
    {
        std::vector<fe> rx_step, shift_r;
        fe rho[2];
        if (!verify_spartan(proof.spartan, h->sys, N, vt, rx_step, shift_r, rho, why)) return false;
        std::vector<int> cols;
        for (int v = 0; v < jolt::NUM_INPUTS; v++) cols.push_back(v);
        take_claims(cols, rx_step, proof.spartan.witness_evals, &rho[0]);
        take_claims(cols, shift_r, proof.spartan.shift_witness_evals, &rho[1]);
    }
    // ---- 6. the batched opening (opening_proof.rs:181-235 + the verifier's reduce_and_verify, out of tree): reduction sumcheck over
    //      all openings, then ONE PST13 check of the joint polynomial (pairing-free, trapdoor known)
    {
        std::vector<fe> batched_claims;
        std::vector<std::vector<fe>> pws;
        for (auto& o : opens) {
            std::vector<fe> pw(1, one);
            for (size_t i = 1; i < o.claims.size(); i++) pw.push_back(Fr::mul(pw[i - 1], o.rho));
            fe bc = Fr::zero();
            for (size_t i = 0; i < pw.size(); i++) bc = Fr::add(bc, Fr::mul(pw[i], o.claims[i]));
            batched_claims.push_back(bc);
            pws.push_back(pw);
        }
        fe rho2 = vt.challenge_scalar();
        size_t max_nv = 0;
        for (auto& o : opens) max_nv = std::max(max_nv, o.point.size());
        std::vector<fe> coeffs(1, one);
        for (size_t i = 1; i < opens.size(); i++) coeffs.push_back(Fr::mul(coeffs[i - 1], rho2));
        fe e = Fr::zero();
        for (size_t i = 0; i < opens.size(); i++)
            e = Fr::add(e, Fr::mul(coeffs[i], Fr::mul(batched_claims[i], fr_from_u64((uint64_t)1 << (max_nv - opens[i].point.size())))));
        std::vector<fe> rs;
        if (!spartan_verify_rounds(proof.reduced.sumcheck_proof, max_nv, 2, e, vt, rs) || proof.reduced.sumcheck_claims.size() != opens.size()) {
            why = "opening reduction: shape";
            return false;
        }
        fe expect = Fr::zero();
        for (size_t i = 0; i < opens.size(); i++) {
            std::vector<fe> slice(rs.end() - (long)opens[i].point.size(), rs.end());
            fe eqv = one;
            for (size_t j = 0; j < slice.size(); j++)
                eqv = Fr::mul(eqv, Fr::add(Fr::sub(Fr::sub(one, opens[i].point[j]), slice[j]), Fr::dbl(Fr::mul(opens[i].point[j], slice[j]))));
            expect = Fr::add(expect, Fr::mul(coeffs[i], Fr::mul(eqv, proof.reduced.sumcheck_claims[i])));
        }
        if (!Fr::eq(expect, e)) {
            why = "opening reduction: final check";
            return false;
        }
        vt.append_scalars(proof.reduced.sumcheck_claims);
        fe vgamma = vt.challenge_scalar();
        // joint commitment = sum_i gamma^i sum_k rho_i^k C_(i,k): one scalar per commitment
        std::vector<fe> scal((size_t)ix.count, Fr::zero());
        fe gp = one, joint_claim = Fr::zero();
        for (size_t i = 0; i < opens.size(); i++) {
            for (size_t k = 0; k < opens[i].polys.size(); k++) {
                fe& sc = scal[(size_t)opens[i].polys[k]];
                sc = Fr::add(sc, Fr::mul(gp, pws[i][k]));
            }
            fe sc = one;
            for (size_t j = 0; j + opens[i].point.size() < max_nv; j++) sc = Fr::mul(sc, Fr::sub(one, rs[j]));
            joint_claim = Fr::add(joint_claim, Fr::mul(gp, Fr::mul(sc, proof.reduced.sumcheck_claims[i])));
            gp = Fr::mul(gp, vgamma);
        }
        std::vector<g1_affine> cs;
        std::vector<fe> ss;
        for (int i = 0; i < ix.count; i++)
            if (!Fr::is_zero(scal[(size_t)i])) {
                cs.push_back(proof.commitments[(size_t)i].g_product);
                ss.push_back(scal[(size_t)i]);
            }
        g1_affine joint_c = PST13::combine_commitments(cs, ss);
        std::vector<fe> rev(rs.rbegin(), rs.rend());
        if (!PST13::check_with_trapdoor(*h->parties[0].setup, joint_c, rev, joint_claim, proof.reduced.joint_opening_proof)) {
            why = "PST13 opening check failed";
            return false;
        }
    }
    (void)M;
    return true;
}

}  // namespace

extern "C" {

int cozk_flow_create(const cozk_flow_config* cfg, cozk_flow** out) {
    if (!cfg || !out) return COZK_ERR_INVALID_ARG;
    *out = nullptr;
    cozk_flow* h = new cozk_flow();
    h->cfg = *cfg;
    try {
        COZK_REQUIRE(cfg->mode == COZK_MODE_PLAIN || cfg->mode == COZK_MODE_REP3, "flow: mode");
        COZK_REQUIRE(cfg->log_n >= 1 && cfg->log_n <= 22, "flow: log_n in 1..22");
        COZK_REQUIRE(cfg->log_m >= 1 && cfg->log_m <= cfg->log_n && cfg->log_b >= 1 && cfg->log_b <= cfg->log_n && cfg->log_mem >= 2 && cfg->log_mem <= cfg->log_n,
                     "flow: log_m, log_b in 1..log_n, log_mem in 2..log_n");
        COZK_REQUIRE(cfg->n_mem >= 1 && cfg->n_mem <= 128 && cfg->n_subtables >= 1 && cfg->n_subtables <= cfg->n_mem, "flow: n_mem in 1..128, n_subtables in 1..n_mem");
        h->nparties = cfg->mode == COZK_MODE_REP3 ? 3 : 1;
        h->N = (size_t)1 << cfg->log_n;
        h->M = (size_t)1 << cfg->log_m;
        h->B = (size_t)1 << cfg->log_b;
        h->MEM = (size_t)1 << cfg->log_mem;
        h->ix = FlowIdx(cfg->n_mem);
        jolt::build_system(h->sys);
        flow_build_clear(h);
        h->parties.resize((size_t)h->nparties);
        for (int p = 0; p < h->nparties; p++) {
            FlowParty& ps = h->parties[(size_t)p];
            ps.party = p;
            int rc = cozk_ctx_create(cfg->devices[p], &ps.ctx);
            if (rc != COZK_OK) throw CozkError(rc, "flow: cannot create a context (no HIP device?)");
            cozk_ctx_set_resident_rounds(ps.ctx, h->nparties > 1 ? 0 : 1);
            HIP_TRY(hipSetDevice(ps.ctx->device));
            flow_setup_party(h, ps);
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

const char* cozk_flow_error(const cozk_flow* h) { return h ? h->error.c_str() : "null harness"; }

int cozk_flow_destroy(cozk_flow* h) {
    if (!h) return COZK_OK;
    for (auto& ps : h->parties) {
        if (ps.ctx) (void)hipSetDevice(ps.ctx->device);
        ps.polys.clear();
        ps.commit_vecs.clear();
        ps.msm_vecs.clear();
        ps.setup.reset();
        ps.iota = VecH();
        ps.bc_table.clear();
        ps.subtables.clear();
        ps.mem_flags.clear();
        ps.io_range = VecH();
        ps.v_io = VecH();
        if (ps.ctx) cozk_ctx_destroy(ps.ctx);
    }
    delete h;
    return COZK_OK;
}

size_t cozk_flow_num_polys(const cozk_flow* h) { return h ? (size_t)h->ix.count : 0; }

cozk_ctx* cozk_flow_ctx(cozk_flow* h, int party) {
    if (!h || party < 0 || party >= (int)h->parties.size()) return nullptr;
    return h->parties[(size_t)party].ctx;
}

int cozk_flow_prove(cozk_flow* h, int verify, cozk_flow_result* res) {
    if (!h || !res) return COZK_ERR_INVALID_ARG;
    memset(res, 0, sizeof *res);
    res->verified = -1;
    const int np = h->nparties;
    InProcStar star(np);
    InProcRing ring(&star.abort);
    std::vector<std::unique_ptr<InProcStarWorker>> sw;
    std::vector<std::unique_ptr<InProcRingNet>> rn;
    for (int p = 0; p < np; p++) {
        sw.emplace_back(new InProcStarWorker(&star, p));
        rn.emplace_back(np == 3 ? new InProcRingNet(&ring, p) : nullptr);
        h->parties[(size_t)p].error.clear();
    }
    std::vector<std::thread> threads;
    double t0 = now_ms();
    for (int p = 0; p < np; p++) {
        threads.emplace_back([&, p] {
            try {
                flow_worker_main(h, h->parties[(size_t)p], sw[(size_t)p].get(), rn[(size_t)p].get());
            } catch (const std::exception& e) {
                h->parties[(size_t)p].error = e.what();
                star.abort.flag.store(true);
            }
        });
    }
    FlowProof proof;
    int rc = COZK_OK;
    try {
        InProcStarCoordinator coord(&star);
        flow_coordinate(h, coord, proof);
    } catch (const std::exception& e) {
        h->error = std::string("coordinator: ") + e.what();
        star.abort.flag.store(true);
        rc = COZK_ERR_INTERNAL;
    }
    for (auto& t : threads) t.join();
    double t1 = now_ms();
    for (int p = 0; p < np; p++)
        if (!h->parties[(size_t)p].error.empty()) {
            h->error = "party " + std::to_string(p) + ": " + h->parties[(size_t)p].error;
            rc = COZK_ERR_INTERNAL;
        }
    if (rc != COZK_OK) return rc;
    res->wall_ms = t1 - t0;
    if (verify) {
        std::string why;
        try {
            HIP_TRY(hipSetDevice(h->parties[0].ctx->device));
            res->verified = flow_verify(h, proof, why) ? 1 : 0;
        } catch (const std::exception& e) {
            why = e.what();
            res->verified = 0;
        }
        if (res->verified == 0) h->error = "verification failed: " + why;
    }
    for (int p = 0; p < np; p++) {
        FlowParty& ps = h->parties[(size_t)p];
        res->t_commit_ms = std::max(res->t_commit_ms, ps.t_commit);
        res->t_bytecode_ms = std::max(res->t_bytecode_ms, ps.t_bytecode);
        res->t_primary_ms = std::max(res->t_primary_ms, ps.t_primary);
        res->t_lookups_gp_ms = std::max(res->t_lookups_gp_ms, ps.t_lookups_gp);
        res->t_rw_ms = std::max(res->t_rw_ms, ps.t_rw);
        res->t_spartan_ms = std::max(res->t_spartan_ms, ps.t_spartan);
        res->t_open_ms = std::max(res->t_open_ms, ps.t_open);
        res->t_worker_ms = std::max(res->t_worker_ms, ps.t_total);
        res->t_spartan_build_ms = std::max(res->t_spartan_build_ms, ps.sp_times.t_build);
        res->bytes_star_up += ps.star_up;
        res->bytes_star_down += ps.star_down;
        res->bytes_ring += ps.ring_bytes;
        res->star_messages += ps.star_msgs;
    }
    res->n_polys = (uint64_t)h->ix.count;
    res->n_openings = 10;
    h->last_proof = proof.serialize();
    res->proof_len = h->last_proof.size();
    Sha256 s;
    s.update(h->last_proof.data(), h->last_proof.size());
    s.final(res->proof_digest);
    return COZK_OK;
}

int cozk_flow_proof_bytes(const cozk_flow* h, uint8_t* out, size_t cap) {
    if (!h || !out || cap < h->last_proof.size()) return COZK_ERR_INVALID_ARG;
    memcpy(out, h->last_proof.data(), h->last_proof.size());
    return COZK_OK;
}

}  // extern "C"
