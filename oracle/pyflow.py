"""ORACLE (test infrastructure -- never imported by the product): the ONE chained co-jolt worker flow of
csrc/host/flow_harness.hpp in pure Python -- JoltRep3Prover::prove (co-jolt/src/jolt/vm/jolt/worker.rs:175-266) with its
coordinator (jolt/vm/jolt/coordinator.rs:118-222): one transcript, every polynomial committed once, one opening accumulator:

    commit-all                                     jolt/vm/jolt/witness.rs:304-382
    bytecode memory checking                       lasso/memory_checking/worker.rs:40-237, jolt/vm/bytecode/worker.rs:43-142
    instruction lookups: primary sumcheck,         jolt/vm/instruction_lookups/worker.rs:95-176, :742-860
      toggled read/write + dense init/final GPs
    read-write memory checking + output check      jolt/vm/read_write_memory/worker.rs:54-180, :196-344
    Spartan (outer + inner + shift)                r1cs/spartan/worker.rs:63-273
    reduce_and_prove over ALL accumulated openings poly/opening_proof.rs:181-291

over a synthetic Jolt-shaped witness (oracle/pyjolt_r1cs.py trace + the memory-checking polynomials of SURVEY App. A).
Small sizes only.  Not restated (out of scope, documented in DESIGN): the public TimestampValidityProof of party 0
(jolt-core, read_write_memory/worker.rs:78-104) and the verifier's multiset-equality check of the hashes (the synthetic
counters are random, not a consistent offline-memory-checking instance).  Masks of mul_vec are zero here: they cancel in
every value the coordinator sees, so the proof bytes do not depend on them."""
import hashlib

import pyharness as H
import pyjolt_r1cs as J
import pylookups as L
import pyprimary as P
import pyref as O
import pysparse as SP
import pyspartan_outer as S

R = O.R


# ------------------------------------------------------------------------------------------------ polynomials
class Poly:
    """one committed polynomial: clear values + per-party parts (ints for a public polynomial / the plain prover, (a, b) shares)"""

    def __init__(self, name, clear, public, share_seed, np_):
        self.name, self.clear, self.public = name, [v % R for v in clear], public
        if public or np_ == 1:
            self.parts = [self.clear] * np_
        else:
            self.parts = O.rep3_share_vec(self.clear, O.harness_prf_key(share_seed, 101), O.harness_prf_key(share_seed, 102))
        self.commit = [[c[0] if isinstance(c, tuple) else c for c in p] for p in self.parts]

    def col(self, q):
        """('P' | 'S', list) for the SharedOrPublic oracles"""
        return ("P" if self.public else "S", self.parts[q])


def default_cfg(**kw):
    cfg = dict(mode="plain", log_n=4, log_m=3, log_b=3, log_mem=3, n_mem=6, n_subtables=3, seed=1)
    cfg.update(kw)
    return cfg


def memory_flag_columns(instrs, iflags, n_mem, n):
    """memory_flag_indices: memory m is live at step t iff the step's instruction uses it"""
    cols = [[0] * n for _ in range(n_mem)]
    for i, ins in enumerate(instrs):
        for m in set(ins.mems):
            for t in range(n):
                if iflags[i][t]:
                    cols[m][t] = 1
    return cols


def build_witness(cfg):
    np_ = 1 if cfg["mode"] == "plain" else 3
    seed = cfg["seed"]
    N, M, B, MEM = 1 << cfg["log_n"], 1 << cfg["log_m"], 1 << cfg["log_b"], 1 << cfg["log_mem"]
    n_mem, n_sub = cfg["n_mem"], cfg["n_subtables"]
    W = {}
    r1 = J.synthetic_columns(seed, N)
    W["r1cs"] = [Poly(J.NAMES[v], r1[v], J.IS_PUBLIC[v], seed + 100 * (v + 1), np_) for v in range(J.NUM_INPUTS)]
    pub = lambda name, off, n, bits: Poly(name, O.synthetic_small(seed + off, n, bits), True, 0, np_)
    # cfg["small_witness"]: counters < 2^log_n, subtable entries and memory words 32 bits (include/cozk.h); default: uniform
    small = int(cfg.get("small_witness", 0))
    cnt_bits, val_bits = (cfg["log_n"], 32) if small else (0, 0)
    sh = lambda name, s, n, bits=0: Poly(name, O.synthetic_fr(s, n, bits), False, s, np_)
    W["bc_t_read"] = pub("bc_t_read", 31, N, 20)
    W["bc_t_final"] = pub("bc_t_final", 32, B, 20)
    # the bytecode table (preprocessing.v_init_final): address, bitflags, rd, rs1, rs2, imm
    W["bc_table"] = [O.synthetic_small(seed + 33, B, 20), O.synthetic_small(seed + 34, B, 32), O.synthetic_small(seed + 35, B, 6),
                     O.synthetic_small(seed + 36, B, 6), O.synthetic_small(seed + 37, B, 6), O.synthetic_small(seed + 38, B, 12)]
    W["rw_t_read"] = [pub("rw_t_read_%s" % k, 41 + i, N, 20) for i, k in enumerate(("rd", "rs1", "rs2", "ram"))]
    W["rw_v_init"] = sh("rw_v_init", seed + 45000, MEM, val_bits)
    W["rw_v_final"] = sh("rw_v_final", seed + 46000, MEM, val_bits)
    W["rw_t_final"] = pub("rw_t_final", 47, MEM, 20)
    W["read_cts"] = [sh("read_cts_%d" % m, seed + 11000 * (m + 1), N, cnt_bits) for m in range(n_mem)]
    W["E"] = [sh("E_%d" % m, seed + 9000 * (m + 1), N, val_bits) for m in range(n_mem)]
    W["final_cts"] = [sh("final_cts_%d" % m, seed + 13000 * (m + 1), M, cnt_bits) for m in range(n_mem)]
    W["subtables"] = [O.synthetic_small(seed + 15000 * (s + 1), M, 32) for s in range(n_sub)]
    instrs = L.instr_table(n_mem)
    iflags = [r1[J.IDX["I_" + nm]] for nm in J.INSTRUCTIONS]
    outs = []
    for t in range(N):
        which = [i for i in range(len(instrs)) if iflags[i][t]]
        outs.append(P.g_plain(instrs[which[0]], [W["E"][m].clear[t] for m in instrs[which[0]].mems]) if which else 0)
    W["lasso_outputs"] = Poly("lasso_outputs", outs, False, seed + 555, np_)
    W["instrs"], W["iflags"] = instrs, iflags
    W["mem_flags"] = memory_flag_columns(instrs, iflags, n_mem, N)
    # program outputs: the io range of the memory and v_io = v_final there (read_write_memory/worker.rs:118-147)
    rng = [1 if MEM // 4 <= i < MEM // 2 else 0 for i in range(MEM)]
    W["io_range"] = rng
    W["v_io"] = [W["rw_v_final"].clear[i] if rng[i] else 0 for i in range(MEM)]
    return W


def commit_order(W):
    """the order of the commitments in the transcript and the proof"""
    return (W["r1cs"] + [W["bc_t_read"]] + W["rw_t_read"] + W["read_cts"] + W["E"] + [W["lasso_outputs"]] + W["final_cts"] + [W["bc_t_final"]]
            + [W["rw_v_init"], W["rw_v_final"], W["rw_t_final"]])


# ------------------------------------------------------------------------------------------------ pieces
def fingerprint(terms, constant, n, q, np_):
    """sum_k coeff_k poly_k + constant as a share vector of party q (add_public placement): terms = [(Poly | list of public ints, coeff)]"""
    cols, cc, polys, pc = [], [], [], []
    for p, c in terms:
        if isinstance(p, Poly) and not p.public:
            polys.append(p.parts[q])
            pc.append(c % R)
        else:
            cols.append(p.clear if isinstance(p, Poly) else p)
            cc.append(c % R)
    party = None if np_ == 1 else q
    if not polys:  # an all-public leaf: promote_to_trivial_share
        vals = [(sum(col[i] * c for col, c in zip(cols, cc)) + constant) % R for i in range(n)]
        return vals if np_ == 1 else [O.rep3_promote_from_trivial(v, q) for v in vals]
    return O.fingerprint_leaves([c[:n] for c in cols], cc, [p[:n] for p in polys], pc, constant % R, party)


def dense_gp(leaves_pp, batch, tr):
    layers = O.gp_construct(leaves_pp, batch, O.SplitMix64(0)) if len(leaves_pp) == 1 else _gp_construct_zero_masks(leaves_pp, batch)
    hashes = O.combine_additive(O.gp_claimed_outputs(layers))
    return layers, hashes


def _gp_construct_zero_masks(leaves_pp, batch):
    n = len(leaves_pp[0])
    per = n // batch
    layers = [[list(l) for l in leaves_pp]]
    for _ in range(per.bit_length() - 2):
        prev = layers[-1]
        lr = [O.interleaved_uninterleave(prev[p]) for p in range(3)]
        zero = [[0] * len(lr[0][0]) for _ in range(3)]
        layers.append(O.rep3_mul_vec([x[0] for x in lr], [x[1] for x in lr], zero))
    return layers


def ser_gp(proof):
    out = O.ser_vec_fr(proof["outputs"]) + O.ser_u64(len(proof["layers"]))
    for lp in proof["layers"]:
        out += O.ser_u64(len(lp["round_polys"])) + b"".join(O.ser_vec_fr(c) for c in lp["round_polys"]) + O.ser_fr(lp["left"]) + O.ser_fr(lp["right"])
    return out


class Accumulator:
    """Rep3ProverOpeningAccumulator: append (opening_proof.rs:77-106) with the coordinator's receive_claims (:108-128)"""

    def __init__(self, np_, tr):
        self.np_, self.tr = np_, tr
        self.openings = [[] for _ in range(np_)]
        self.claims = []
        self.meta = []  # (polys, point)

    def append(self, polys, point):
        np_ = self.np_
        eq = O.eq_evals(point)
        parts = []
        for q in range(np_):
            cl = []
            for p in polys:
                if p.public:
                    cl.append(O.additive_promote_from_trivial(sum(a * b for a, b in zip(p.parts[q], eq)) % R, q))
                else:
                    cl.append(O.dense_evaluate_at_chi(p.parts[q], eq))
            parts.append(cl)
        claims, rho, batched = S.receive_claims(parts, self.tr)
        self.claims.append(claims)
        self.meta.append((polys, list(point)))
        pw = [1]
        for _ in range(1, len(polys)):
            pw.append(pw[-1] * rho % R)
        n = max(len(p.clear) for p in polys)
        for q in range(np_):
            if np_ == 1:
                acc = [0] * n
                for c, p in zip(pw, polys):
                    for i, v in enumerate(p.parts[q]):
                        acc[i] = (acc[i] + v * c) % R
                claim_share = batched
            else:
                acc = [(0, 0)] * n
                for c, p in zip(pw, polys):
                    for i, v in enumerate(p.parts[q]):
                        if p.public:
                            a, b = acc[i]
                            if q == 0:
                                a = (a + v * c) % R
                            elif q == 1:
                                b = (b + v * c) % R
                            acc[i] = (a, b)
                        else:
                            acc[i] = O.rep3_add(acc[i], O.rep3_mul_public(v, c))
                claim_share = O.rep3_promote_from_trivial(batched, q)
            self.openings[q].append({"poly": acc, "eq": list(eq), "point": list(point), "claim": claim_share})
        return claims


def memory_checking(tr, acc, np_, rw_leaves_pp, rw_batch, if_leaves_pp, if_batch, rw_polys, if_polys, toggled=None):
    """prove_memory_checking (lasso/memory_checking/worker.rs:40-127) + coordinator (mod.rs:69-160) after (gamma, tau):
    construct both circuits, hashes, both grand-product proofs, the two opening appends.  toggled = flag index lists when the
    read / write circuit is a Rep3ToggledBatchedGrandProduct.  Returns the serialized section."""
    if toggled is not None:
        toggles, sparse = SP.toggled_construct(toggled, rw_leaves_pp)
        rw_hashes = O.combine_additive(SP.toggled_claimed_outputs(sparse))
    else:
        rw_layers, rw_hashes = dense_gp(rw_leaves_pp, rw_batch, tr)
    if_layers, if_hashes = dense_gp(if_leaves_pp, if_batch, tr)
    tr.append_scalars(rw_hashes)
    tr.append_scalars(if_hashes)
    if toggled is not None:
        rw_proof, r_rw = SP.toggled_prove(toggles, sparse, tr)
        rw_blob = L.serialize(rw_proof)
    else:
        rw_proof, r_rw = O.gp_prove(rw_layers, tr)
        rw_blob = ser_gp(rw_proof)
    if_proof, r_if = O.gp_prove(if_layers, tr)

    def batch_bits(b):
        k = 0
        while (1 << k) < b:
            k += 1
        return k
    r_rw_open = r_rw[batch_bits(rw_batch):]
    r_if_open = r_if[batch_bits(if_batch):]
    c1 = acc.append(rw_polys, r_rw_open)
    c2 = acc.append(if_polys, r_if_open)
    return O.ser_vec_fr(rw_hashes) + O.ser_vec_fr(if_hashes) + rw_blob + ser_gp(if_proof) + O.ser_vec_fr(c1) + O.ser_vec_fr(c2)


def prove_arbitrary_product(polys_pp, claims, num_rounds, degree, tr):
    """prove_arbitrary_worker (subprotocols/sumcheck.rs:168-246) for comb_func = product of dense polynomials (at most one
    shared), HighToLow, with coordinate_prove_arbitrary.  Returns (compressed polys, r)."""
    np_ = len(polys_pp)
    prev = list(claims)
    comps, rs = [], []
    for _ in range(num_rounds):
        msgs = []
        for q in range(np_):
            ev = O.prod_round_evals(polys_pp[q], degree)
            any_shared = any(isinstance(p[0], tuple) for p in polys_pp[q])
            if np_ == 3 and not any_shared:
                ev = [O.additive_promote_from_trivial(v, q) for v in ev]
            pts = [ev[0], (prev[q] - ev[0]) % R] + ev[1:]
            msgs.append(O.unipoly_from_evals(pts))
        poly = O.combine_additive(msgs)
        comp = O.unipoly_compress(poly)
        tr.append_scalars(comp)
        r_j = tr.challenge_scalar()
        rs.append(r_j)
        comps.append(comp)
        nxt = O.unipoly_eval(poly, r_j)
        for q in range(np_):
            prev[q] = O.additive_promote_from_trivial(nxt, q)
            polys_pp[q] = [O.dense_bind(p, r_j, O.HIGH_TO_LOW) if isinstance(p[0], tuple) else O.public_bind(p, r_j, O.HIGH_TO_LOW) for p in polys_pp[q]]
    return comps, rs


# ------------------------------------------------------------------------------------------------ the flow
def run(cfg):
    cfg = default_cfg(**cfg)
    np_ = 1 if cfg["mode"] == "plain" else 3
    seed = cfg["seed"]
    nv = cfg["log_n"]
    N, M, B, MEM = 1 << nv, 1 << cfg["log_m"], 1 << cfg["log_b"], 1 << cfg["log_mem"]
    n_mem, n_sub = cfg["n_mem"], cfg["n_subtables"]
    assert cfg["log_m"] <= nv and cfg["log_b"] <= nv and cfg["log_mem"] <= nv
    W = build_witness(cfg)
    ck = H._pst_setup(seed, nv)
    tr = O.Transcript(b"cozk-jolt")
    sections = {}
    # ---- 1. commit-all
    order = commit_order(W)
    commitments = []
    for p in order:
        if p.public:
            commitments.append(H._commit(ck, p.commit[0]))
        else:
            c = None
            for q in range(np_):
                c = O.g1_add(c, H._commit(ck, p.commit[q]))
            commitments.append(c)
    for c in commitments:
        tr.append_point(c)
    blob = O.ser_u64(len(order))
    for p, c in zip(order, commitments):
        blob += O.ser_u64(len(p.clear).bit_length() - 1) + O.ser_g1(c)
    sections["commit"] = blob
    acc = Accumulator(np_, tr)
    r1 = {p.name: p for p in W["r1cs"]}

    def gammas(k):
        g, tau = tr.challenge_scalar(), tr.challenge_scalar()
        pw = [1]
        for _ in range(k):
            pw.append(pw[-1] * g % R)
        return pw, tau
    # ---- 2. bytecode memory checking (bytecode/worker.rs:43-142)
    g, tau = gammas(7)
    bc_v = [r1["Bytecode_ELFAddress"], r1["Bytecode_Bitflags"], r1["Bytecode_RD"], r1["Bytecode_RS1"], r1["Bytecode_RS2"]]
    read_terms = [(r1["Bytecode_A"], g[1])] + [(bc_v[k], g[2 + k]) for k in range(5)] + [(W["bc_t_read"], g[7]), (r1["Bytecode_Imm"], 1)]
    tab = W["bc_table"]
    init_terms = [(list(range(B)), g[1])] + [(tab[k], g[2 + k]) for k in range(5)] + [(tab[5], 1)]
    rw_pp, if_pp = [], []
    for q in range(np_):
        rw_pp.append(fingerprint(read_terms, -tau, N, q, np_) + fingerprint(read_terms, g[7] - tau, N, q, np_))
        if_pp.append(fingerprint(init_terms, -tau, B, q, np_) + fingerprint(init_terms + [(W["bc_t_final"], g[7])], -tau, B, q, np_))
    sections["bytecode"] = memory_checking(tr, acc, np_, rw_pp, 2, if_pp, 2,
                                           [r1["Bytecode_A"]] + bc_v + [r1["Bytecode_Imm"], W["bc_t_read"]], [W["bc_t_final"]])
    # ---- 3. instruction lookups (instruction_lookups/worker.rs:95-176)
    instrs, iflags = W["instrs"], W["iflags"]
    r_eq = tr.challenge_vector(nv)
    E_pp = [[W["E"][m].parts[q] for m in range(n_mem)] for q in range(np_)]
    outs_pp = [W["lasso_outputs"].parts[q] for q in range(np_)]
    pproof, prs, _fin = P.prove(instrs, r_eq, iflags, E_pp, outs_pp, tr)
    blob = O.ser_u64(len(pproof["round_polys"])) + b"".join(O.ser_vec_fr(c) for c in pproof["round_polys"]) + O.ser_vec_fr(pproof["openings"])
    iflag_polys = [r1["I_" + nm] for nm in J.INSTRUCTIONS]
    c0 = acc.append(W["E"] + iflag_polys + [W["lasso_outputs"]], list(reversed(prs)))
    assert c0 == pproof["openings"]
    blob += O.ser_vec_fr(c0)
    g, tau = gammas(2)
    dims = [r1["ChunksQuery%d" % i] for i in range(4)]
    rw_pp, if_pp = [], []
    for q in range(np_):
        per = []
        for m in range(n_mem):
            terms = [(W["read_cts"][m], g[2]), (W["E"][m], g[1]), (dims[m % 4], 1)]
            per.append(fingerprint(terms, -tau, N, q, np_))
            per.append(fingerprint(terms, g[2] - tau, N, q, np_))
        rw_pp.append(per)  # per circuit lists (the toggled construct takes them circuit by circuit)
        leaves = []
        for s in range(n_sub):
            init_terms = [(W["subtables"][s], g[1]), (list(range(M)), 1)]
            leaves += fingerprint(init_terms, -tau, M, q, np_)
            for m in range(n_mem):
                if m % n_sub == s:
                    leaves += fingerprint(init_terms + [(W["final_cts"][m], g[2])], -tau, M, q, np_)
        if_pp.append(leaves)
    flag_indices = [[t for t in range(N) if W["mem_flags"][m][t]] for m in range(n_mem)]
    sections["lookups"] = blob + memory_checking(tr, acc, np_, rw_pp, 2 * n_mem, if_pp, n_sub + n_mem,
                                                 dims + W["read_cts"] + W["E"] + iflag_polys + [W["lasso_outputs"]], W["final_cts"], toggled=flag_indices)
    # ---- 4. read-write memory checking + output check (read_write_memory/worker.rs:54-180, 196-344)
    g, tau = gammas(2)
    ident = list(range(N))
    regs = [("Bytecode_RS1", "RS1_Read", "RS1_Read", 1), ("Bytecode_RS2", "RS2_Read", "RS2_Read", 2), ("Bytecode_RD", "RD_Read", "RD_Write", 0),
            ("RAM_Address", "RAM_Read", "RAM_Write", 3)]
    rw_pp, if_pp = [], []
    for q in range(np_):
        leaves = []
        for a, vr, vw, ti in regs:
            leaves += fingerprint([(r1[vr], g[1]), (W["rw_t_read"][ti], g[2]), (r1[a], 1)], -tau, N, q, np_)
            leaves += fingerprint([(r1[vw], g[1]), (ident, g[2]), (r1[a], 1)], -tau, N, q, np_)
        rw_pp.append(leaves)
        if_pp.append(fingerprint([(W["rw_v_init"], g[1]), (list(range(MEM)), 1)], -tau, MEM, q, np_)
                     + fingerprint([(W["rw_v_final"], g[1]), (W["rw_t_final"], g[2]), (list(range(MEM)), 1)], -tau, MEM, q, np_))
    rw_polys = [r1["RAM_Address"], r1["RD_Read"], r1["RS1_Read"], r1["RS2_Read"], r1["RAM_Read"], r1["RD_Write"], r1["RAM_Write"]] + W["rw_t_read"] + \
               [r1["Bytecode_RD"], r1["Bytecode_RS1"], r1["Bytecode_RS2"]]
    blob = memory_checking(tr, acc, np_, rw_pp, 8, if_pp, 2, rw_polys, [W["rw_v_final"], W["rw_t_final"]])
    r_eq = tr.challenge_vector(cfg["log_mem"])
    eqv = O.eq_evals(r_eq)
    polys_pp = []
    for q in range(np_):
        vf = W["rw_v_final"].parts[q]
        if np_ == 1:
            d = [(a - b) % R for a, b in zip(vf, W["v_io"])]
        else:
            d = [SP.rep3_sub_shared_by_public(a, b, q) for a, b in zip(vf, W["v_io"])]
        polys_pp.append([list(eqv), list(W["io_range"]), d])
    comps, r_out = prove_arbitrary_product(polys_pp, [0] * np_, cfg["log_mem"], 3, tr)
    blob += O.ser_u64(len(comps)) + b"".join(O.ser_vec_fr(c) for c in comps)
    blob += O.ser_vec_fr(acc.append([W["rw_v_final"]], r_out))
    sections["rw"] = blob
    # ---- 5. Spartan (r1cs/spartan/worker.rs:63-273)
    uniform, cross, padded = J.build_system()
    cols_pp = [[p.col(q) for p in W["r1cs"]] for q in range(np_)]
    sp = prove_spartan_with_acc(uniform, cross, padded, cols_pp, W["r1cs"], N, tr, acc)
    sections["spartan"] = S.serialize_full(sp)
    # ---- 6. reduce_and_prove (opening_proof.rs:181-291)
    r_red, red_claims, red = O.opening_reduce(acc.openings, tr)
    gamma = tr.challenge_scalar()
    gpw = [1]
    for _ in range(1, len(acc.meta)):
        gpw.append(gpw[-1] * gamma % R)
    proofs = None
    point_rev = list(reversed(r_red))
    for q in range(np_):
        joint = [0] * N
        for c, op in zip(gpw, acc.openings[q]):
            for i, v in enumerate(op["poly"]):
                a = v[0] if isinstance(v, tuple) else v
                joint[i] = (joint[i] + a * c) % R
        pf, _ = O.pst_open(ck, joint, point_rev)
        proofs = pf if proofs is None else [O.g1_add(x, y) for x, y in zip(proofs, pf)]
    blob = O.ser_u64(len(red["round_polys"])) + b"".join(O.ser_vec_fr(c) for c in red["round_polys"]) + O.ser_vec_fr(red_claims)
    blob += O.ser_u64(len(proofs)) + b"".join(O.ser_g1(p) for p in proofs)
    sections["open"] = blob
    out = b"".join(sections[k] for k in ("commit", "bytecode", "lookups", "rw", "spartan", "open"))
    return {"proof_bytes": out, "digest": hashlib.sha256(out).hexdigest(), "sections": sections, "n_openings": len(acc.meta),
            "n_commitments": len(order)}


def prove_spartan_with_acc(uniform, cross, padded, cols_pp, polys, num_steps, tr, acc):
    """pyspartan_outer.prove_full with the two claim exchanges going through the flow's accumulator"""
    proof = S.prove_full(uniform, cross, padded, cols_pp, num_steps, tr, appends=lambda point: acc.append(polys, point))
    return proof
