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

// the instruction table of the primary sumcheck: the 27 RV32I instructions in the order of jolt/vm/rv32i_vm.rs:41-70, each
// with the collation form and memory count of its combine_lookups at C = 4, M = 2^16 (co-jolt/src/jolt/instruction/*.rs).
// WHICH memory an instruction's subtable lands in comes from jolt-core's preprocessing (out of tree): the harness
// assigns them synthetically, instruction t using memories 3 t, 3 t + 1, .. modulo n_mem (instructions share memories,
// as they share subtables in the VM).
std::vector<cozk_primary_instr> lookups_instr_table(int n_mem) {
    struct Row {
        int form, n_mems, bits, repeat;
    };
    const int C = 4;
    static const Row rows[27] = {
        {COZK_G_CONCAT, C / 2, 16, 0},              // ADD   add.rs:29-33
        {COZK_G_CONCAT, C / 2, 16, 0},              // SUB   sub.rs:31-36
        {COZK_G_CONCAT, C, 8, 0},                   // AND   and.rs:34-36
        {COZK_G_CONCAT, C, 8, 0},                   // OR
        {COZK_G_CONCAT, C, 8, 0},                   // XOR
        {COZK_G_PRODUCT, C, 0, 0},                  // BEQ   beq.rs:35-37
        {COZK_G_NOT_SLT, 2 * C + 1, 0, 0},          // BGE   bge.rs:34-43
        {COZK_G_NOT_LTU, 2 * C - 1, 0, 0},          // BGEU
        {COZK_G_NOT_PRODUCT, C, 0, 0},              // BNE
        {COZK_G_SLT, 2 * C + 1, 0, 0},              // SLT   slt.rs:33-59
        {COZK_G_LTU, 2 * C - 1, 0, 0},              // SLTU  sltu.rs:32-47
        {COZK_G_CONCAT, C, 8, 0},                   // SLL   sll.rs:32-35
        {COZK_G_CONCAT, C + 1, 0, 0},               // SRA   sra.rs:32-36 (sum)
        {COZK_G_CONCAT, C, 0, 0},                   // SRL   srl.rs (sum)
        {COZK_G_CONCAT, 2, 16, 1},                  // MOVSIGN  virtual_movsign.rs:36-42: the one value repeated WORD_SIZE / 16 times
        {COZK_G_CONCAT, C / 2, 16, 0},              // MUL
        {COZK_G_CONCAT, C / 2, 16, 0},              // MULU
        {COZK_G_CONCAT, C / 2, 16, 0},              // MULHU
        {COZK_G_CONCAT, C / 2, 16, 0},              // VIRTUAL_ADVICE
        {COZK_G_CONCAT, C, 16, 0},                  // VIRTUAL_MOVE  virtual_move.rs:26-28
        {COZK_G_LTE, 2 * C, 0, 0},                  // VIRTUAL_ASSERT_LTE
        {COZK_G_SIGNED_REM, 4 * C + 2, 0, 0},       // VIRTUAL_ASSERT_VALID_SIGNED_REMAINDER
        {COZK_G_UNSIGNED_REM, 3 * C - 1, 0, 0},     // VIRTUAL_ASSERT_VALID_UNSIGNED_REMAINDER
        {COZK_G_DIV0, 2 * C, 0, 0},                 // VIRTUAL_ASSERT_VALID_DIV0
        {COZK_G_NOT_FIRST, 1, 0, 0},                // VIRTUAL_ASSERT_HALFWORD_ALIGNMENT
        {COZK_G_ZERO, 1, 0, 0},                     // VIRTUAL_POW2
        {COZK_G_ZERO, 1, 0, 0},                     // VIRTUAL_SRA_PADDING
    };
    std::vector<cozk_primary_instr> out;
    for (int t = 0; t < 27; t++) {
        cozk_primary_instr in{};
        in.form = rows[t].form;
        in.bits = rows[t].bits;
        in.n_mems = rows[t].n_mems;
        for (int j = 0; j < in.n_mems; j++) in.mems[j] = (3 * t + (rows[t].repeat ? 0 : j)) % n_mem;
        out.push_back(in);
    }
    return out;
}

// cfg.mix = 1: a trace-shaped instruction mix -- the proportions of a sha2-chain guest (the reference's benchmark program,
// co-jolt/README.md:20-31): per 256 cycles, in the order of jolt/vm/rv32i_vm.rs:41-70.  ADD (address arithmetic and the loads /
// stores' lookups), XOR, AND, OR, SLL, SRL dominate; 15 of 256 cycles (6 %) run a multiplicative collation against 37 % uniformly.
static const uint8_t LOOKUPS_SHA2_MIX[27] = {72, 4, 20, 20, 56, 2, 1, 1, 3, 1, 6, 24, 2, 30, 1, 1, 1, 1, 2, 4, 1, 0, 0, 0, 2, 1, 0};
static inline int lookups_mix_instr(int mix, uint32_t byte) {
    if (mix == 0) return (int)(byte % 27);
    uint32_t acc = 0;
    for (int i = 0; i < 27; i++) {
        acc += LOOKUPS_SHA2_MIX[i];
        if (byte < acc) return i;
    }
    return 0;
}

// the dealer's view: which instruction every cycle runs and lookup_outputs(x) = g_{which(x)}(E(x)) in the clear
void lookups_setup_primary_clear(cozk_lookups* h) {
    const cozk_lookups_config& c = h->cfg;
    h->instrs = lookups_instr_table(c.n_pairs);
    h->which.resize(h->N);
    h->outputs_plain.resize(h->N);
    std::vector<fe> E((size_t)c.n_pairs);
    for (size_t x = 0; x < h->N; x++) {
        uint8_t w = (uint8_t)lookups_mix_instr(c.mix, synthetic_small_host(c.seed + 1234567ull, x, 8));
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

// star: this party's channel of the grand-product phase; pstars: its 2^log_workers channels of the primary sumcheck
void lookups_worker_main(cozk_lookups* h, LookupsParty& ps, StarNetWorker* star, const std::vector<StarNetWorker*>& pstars, RingNet* ring) {
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
        //      then prove_primary_sumcheck (:180-373), split over 2^log_workers worker sub-nets
        const int W = 1 << c.log_workers;
        std::vector<fe> r_eq;
        for (int w = 0; w < W; w++) {
            Bytes req = pstars[w]->receive_request();
            Reader rd(req);
            r_eq = rd.vec_fr();
        }
        std::vector<uint64_t> wabi = to_abi(r_eq);
        cozk_vec* eqv = nullptr;
        rc_check(cozk_eq_evals(env.ctx, wabi.data(), (int)r_eq.size(), &eqv), env.ctx, "eq_evals");
        VecH eqh(eqv);
        const size_t n_mem = ps.E.size(), n_instr = h->instrs.size();
        const size_t chunk = h->N >> c.log_workers;
        // split_poly (worker.rs:213-235): worker w gets the w-th high-variable chunk of every polynomial -- zero-copy views
        std::vector<PrimaryH> prims;
        std::vector<std::vector<PolyH>> views((size_t)W);
        for (int w = 0; w < W; w++) {
            const size_t off = (size_t)w * chunk;
            std::vector<cozk_vec> fv(n_instr);
            std::vector<const cozk_vec*> fl;
            for (size_t i = 0; i < n_instr; i++) {
                fv[i] = *ps.instr_flags[i].h;
                fv[i].d = (uint8_t*)fv[i].d + off;
                fv[i].n = chunk;
                fv[i].bytes = chunk;
                fv[i].owned = false;
                fl.push_back(&fv[i]);
            }
            std::vector<const cozk_poly*> E;
            for (auto& e : ps.E) {
                cozk_poly* v = nullptr;
                rc_check(cozk_poly_chunk(env.ctx, e.h, off, chunk, &v), env.ctx, "poly_chunk");
                views[w].push_back(PolyH(v));
                E.push_back(v);
            }
            cozk_poly* ov = nullptr;
            rc_check(cozk_poly_chunk(env.ctx, ps.outputs.h, off, chunk, &ov), env.ctx, "poly_chunk");
            views[w].push_back(PolyH(ov));
            cozk_vec ev = *eqh.h;
            ev.d = (fe*)ev.d + off;
            ev.n = chunk;
            ev.bytes = chunk * sizeof(fe);
            ev.owned = false;
            cozk_primary* pr = nullptr;
            rc_check(cozk_primary_create(env.ctx, c.mode, ps.party, h->instrs.data(), n_instr, fl.data(), E.data(), n_mem, ov, &ev, &pr), env.ctx, "primary_create");
            prims.emplace_back(pr);
        }
        if (W == 1) {
            env.star = pstars[0];
            (void)prove_primary_sumcheck_worker(env, prims[0].h, c.log_n, n_mem, n_instr);
        } else {
            // the first log_n - log_workers rounds: every sub-net answers each round (worker.rs:237-300); the party's
            // workers are time-sliced on this context, one GPU each in a real deployment
            const int D = cozk_primary_degree(prims[0].h);
            const int split_rounds = c.log_n - c.log_workers;
            uint64_t rr[4];
            for (int round = 0; round < split_rounds; round++) {
                for (int w = 0; w < W; w++) {
                    size_t n_items = 0;
                    int n_levels = 0;
                    rc_check(cozk_primary_round_begin(env.ctx, prims[w].h, round ? rr : nullptr, &n_items, &n_levels), env.ctx, "primary_round_begin");
                    for (int level = 1; level <= n_levels; level++) {
                        const void* send = nullptr;
                        void* recv = nullptr;
                        size_t n = 0;
                        rc_check(cozk_primary_level(env.ctx, prims[w].h, level, env.key_self, env.key_prev, env.mask_ctr, &send, &recv, &n), env.ctx, "primary_level");
                        if (env.mode == COZK_MODE_REP3 && n) {
                            env.ring->reshare(env.ctx, (const fe*)send, (fe*)recv, n);
                            env.mask_ctr += n;
                        }
                    }
                    std::vector<uint64_t> ev(4 * (size_t)D);
                    rc_check(cozk_primary_round_finish(env.ctx, prims[w].h, ev.data()), env.ctx, "primary_round_finish");
                    std::vector<fe> msg((size_t)D);
                    for (int k = 0; k < D; k++) msg[k] = fe_from_u64x4(ev.data() + 4 * k);
                    Writer wr;
                    wr.vec_fr(msg);
                    pstars[w]->send_response(wr.b);
                }
                for (int w = 0; w < W; w++) {
                    Bytes req = pstars[w]->receive_request();
                    Reader rd(req);
                    fe_to_u64x4(rd.fr(), rr);
                }
            }
            // every worker's final values become entry w of the 2^log_workers-long polynomials of the remaining rounds
            // (worker.rs:301-345), which worker 0 proves (:347-372)
            std::vector<fe> eq_rem((size_t)W);
            std::vector<std::vector<fe>> fl_rem(n_instr, std::vector<fe>((size_t)W));
            std::vector<std::vector<fe>> Ea(n_mem, std::vector<fe>((size_t)W)), Eb(n_mem, std::vector<fe>((size_t)W));
            std::vector<fe> oa((size_t)W), ob((size_t)W);
            for (int w = 0; w < W; w++) {
                std::vector<uint64_t> Ee(8 * n_mem), Fe(4 * n_instr);
                uint64_t oe[8], qe[4];
                rc_check(cozk_primary_final_evals(env.ctx, prims[w].h, rr, Ee.data(), Fe.data(), oe, qe), env.ctx, "primary_final_evals");
                eq_rem[w] = fe_from_u64x4(qe);
                for (size_t i = 0; i < n_instr; i++) fl_rem[i][w] = fe_from_u64x4(Fe.data() + 4 * i);
                for (size_t m = 0; m < n_mem; m++) {
                    Ea[m][w] = fe_from_u64x4(Ee.data() + 8 * m);
                    Eb[m][w] = fe_from_u64x4(Ee.data() + 8 * m + 4);
                }
                oa[w] = fe_from_u64x4(oe);
                ob[w] = fe_from_u64x4(oe + 4);
            }
            auto up = [&](const std::vector<fe>& v) {
                cozk_vec* d = nullptr;
                rc_check(cozk_vec_upload(env.ctx, v.data(), v.size(), COZK_SCALAR_FR, &d), env.ctx, "vec_upload");
                return VecH(d);
            };
            auto mkpoly = [&](const std::vector<fe>& a, const std::vector<fe>& b) {
                VecH va = up(a), vb;
                if (c.mode == COZK_MODE_REP3) vb = up(b);
                cozk_poly* p = nullptr;
                rc_check(cozk_poly_create(env.ctx, c.mode, va.h, vb.h, &p), env.ctx, "poly_create");
                return PolyH(p);
            };
            std::vector<VecH> flv;
            std::vector<const cozk_vec*> flp;
            for (size_t i = 0; i < n_instr; i++) {
                flv.push_back(up(fl_rem[i]));
                flp.push_back(flv.back().h);
            }
            std::vector<PolyH> Er;
            std::vector<const cozk_poly*> Ep;
            for (size_t m = 0; m < n_mem; m++) {
                Er.push_back(mkpoly(Ea[m], Eb[m]));
                Ep.push_back(Er.back().h);
            }
            PolyH outr = mkpoly(oa, ob);
            VecH eqr = up(eq_rem);
            cozk_primary* pr = nullptr;
            rc_check(cozk_primary_create(env.ctx, c.mode, ps.party, h->instrs.data(), n_instr, flp.data(), Ep.data(), n_mem, outr.h, eqr.h, &pr), env.ctx,
                     "primary_create(remaining rounds)");
            PrimaryH prh(pr);
            PrimaryFinals fin;
            (void)prove_primary_rounds(env, pstars[0], pr, c.log_workers, n_mem, n_instr, fin);
            send_primary_openings(env, pstars[0], fin);
        }
        env.star = star;
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

int lookups_coordinator_main(cozk_lookups* h, StarNetCoordinator& net, StarNetCoordinator& pnet, LookupsProof& proof, bool verify, std::string& why) {
    Transcript tr("cozk-lookups");
    std::vector<fe> r_eq, r_primary;
    if (h->cfg.primary) {
        r_eq = tr.challenge_vector((size_t)h->cfg.log_n);
        Writer w;
        w.vec_fr(r_eq);
        pnet.broadcast_request(w.b);
        proof.has_primary = true;
        proof.primary = coordinate_primary_sumcheck(pnet, tr, h->cfg.log_n, r_primary, h->nparties, h->cfg.log_workers);
    }
    size_t num_layers = (size_t)h->cfg.log_n + 1;  // tree_depth sparse layers + the toggle layer
    std::vector<fe> r;
    proof.gp = coordinate_prove_toggled_grand_product(net, tr, num_layers, r);
    if (!verify) return -1;
    Transcript vt("cozk-lookups");
    if (h->cfg.primary) {
        std::vector<fe> vr_eq = vt.challenge_vector((size_t)h->cfg.log_n), rs;
        const int degree = primary_sumcheck_degree(h->instrs);
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
        COZK_REQUIRE(cfg->log_workers >= 0 && cfg->log_workers <= 3 && cfg->log_workers < cfg->log_n, "lookups: log_workers in 0..3, below log_n");
        COZK_REQUIRE(cfg->log_n >= 1 && cfg->log_n <= 24 && cfg->n_pairs >= 1 && cfg->n_pairs <= 128 && cfg->density_pct >= 0 && cfg->density_pct <= 100,
                     "lookups: log_n in 1..24, n_pairs in 1..128, density_pct in 0..100");
        COZK_REQUIRE(cfg->mix == 0 || cfg->mix == 1, "lookups: mix is 0 (uniform) or 1 (sha2-shaped)");
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
    const int W = 1 << h->cfg.log_workers;
    InProcStar star(np);
    InProcStar pstar(np * W);  // the primary sumcheck's worker sub-nets: participant = worker * parties + party
    pstar.abort.flag.store(false);
    InProcRing ring(&star.abort);
    std::vector<std::unique_ptr<InProcStarWorker>> sw, psw;
    std::vector<std::unique_ptr<InProcRingNet>> rn;
    for (int p = 0; p < np; p++) {
        sw.emplace_back(new InProcStarWorker(&star, p));
        rn.emplace_back(np == 3 ? new InProcRingNet(&ring, p) : nullptr);
        h->parties[p].error.clear();
    }
    for (int id = 0; id < np * W; id++) psw.emplace_back(new InProcStarWorker(&pstar, id));
    std::vector<std::thread> threads;
    double t0 = now_ms();
    for (int p = 0; p < np; p++) {
        threads.emplace_back([&, p] {
            try {
                std::vector<StarNetWorker*> mine;
                for (int w = 0; w < W; w++) mine.push_back(psw[(size_t)w * np + p].get());
                lookups_worker_main(h, h->parties[p], sw[p].get(), mine, rn[p].get());
            } catch (const std::exception& e) {
                h->parties[p].error = e.what();
                star.abort.flag.store(true);
                pstar.abort.flag.store(true);
            }
        });
    }
    LookupsProof proof;
    std::string why;
    int verified = -1;
    int rc = COZK_OK;
    try {
        InProcStarCoordinator coord(&star), pcoord(&pstar);
        verified = lookups_coordinator_main(h, coord, pcoord, proof, verify != 0, why);
    } catch (const std::exception& e) {
        h->error = std::string("coordinator: ") + e.what();
        star.abort.flag.store(true);
        pstar.abort.flag.store(true);
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
