"""ORACLE (test infrastructure): the synthetic instruction-lookups pipeline of csrc/host/lookups_harness.hpp in pure
Python over the SPARSE oracle (oracle/pysparse.py): same synthetic flags and fingerprints, same transcript, same proof
serialization -- for small sizes the engine's proof must be bit-identical (plain prover and 3-party Rep3 run)."""
import hashlib

import pyprimary as P
import pyref as O
import pysparse as S

R = O.R


def flag_column(seed, q, n, density_pct):
    b = O.synthetic_small(seed + 4000 * (q + 1), n, 8)
    return [1 if v * 100 < density_pct * 256 else 0 for v in b]


def fingerprints_of(seed, b, n, nparties):
    s = seed + 7000 * (b + 1)
    v = O.synthetic_fr(s, n)
    if nparties == 1:
        return [v]
    return O.rep3_share_vec(v, O.harness_prf_key(s, 101), O.harness_prf_key(s, 102))


def serialize(proof):
    out = O.ser_vec_fr(proof["outputs"]) + O.ser_u64(len(proof["layers"]))
    for lp in proof["layers"]:
        out += O.ser_u64(len(lp["round_polys"]))
        for comp in lp["round_polys"]:
            out += O.ser_vec_fr(comp)
        out += O.ser_fr(lp["left"]) + O.ser_fr(lp["right"])
    return out


def instr_table(n_mem):
    """lookups_harness.hpp lookups_instr_table: the 27 RV32I instructions (jolt/vm/rv32i_vm.rs:41-70) with the collation form and
    memory count of their combine_lookups at C = 4, M = 2^16; instruction t uses memories 3 t, 3 t + 1, .. modulo n_mem"""
    C = 4
    rows = [(P.CONCAT, C // 2, 16, 0), (P.CONCAT, C // 2, 16, 0), (P.CONCAT, C, 8, 0), (P.CONCAT, C, 8, 0), (P.CONCAT, C, 8, 0),
            (P.PRODUCT, C, 0, 0), (P.NOT_SLT, 2 * C + 1, 0, 0), (P.NOT_LTU, 2 * C - 1, 0, 0), (P.NOT_PRODUCT, C, 0, 0),
            (P.SLT, 2 * C + 1, 0, 0), (P.LTU, 2 * C - 1, 0, 0), (P.CONCAT, C, 8, 0), (P.CONCAT, C + 1, 0, 0), (P.CONCAT, C, 0, 0),
            (P.CONCAT, 2, 16, 1), (P.CONCAT, C // 2, 16, 0), (P.CONCAT, C // 2, 16, 0), (P.CONCAT, C // 2, 16, 0),
            (P.CONCAT, C // 2, 16, 0), (P.CONCAT, C, 16, 0), (P.LTE, 2 * C, 0, 0), (P.SIGNED_REM, 4 * C + 2, 0, 0),
            (P.UNSIGNED_REM, 3 * C - 1, 0, 0), (P.DIV0, 2 * C, 0, 0), (P.NOT_FIRST, 1, 0, 0), (P.ZERO, 1, 0, 0), (P.ZERO, 1, 0, 0)]
    return [P.Instr(form, [(3 * t + (0 if rep else j)) % n_mem for j in range(n)], bits) for t, (form, n, bits, rep) in enumerate(rows)]


SHA2_MIX = [72, 4, 20, 20, 56, 2, 1, 1, 3, 1, 6, 24, 2, 30, 1, 1, 1, 1, 2, 4, 1, 0, 0, 0, 2, 1, 0]  # lookups_harness.hpp LOOKUPS_SHA2_MIX: cycles per 256 of each RV32I instruction in a sha2-chain-shaped trace


def mix_instr(mix, byte):
    if not mix:
        return byte % 27
    acc = 0
    for i, c in enumerate(SHA2_MIX):
        acc += c
        if byte < acc:
            return i
    return 0


def run_primary(cfg, tr, vt):
    """the primary-sumcheck phase: returns (serialized part, verified)"""
    nparties = 1 if cfg["mode"] == "plain" else 3
    n = 1 << cfg["log_n"]
    n_mem = cfg["n_pairs"]
    seed = cfg["seed"]
    instrs = instr_table(n_mem)
    which = [mix_instr(cfg.get("mix", 0), v) for v in O.synthetic_small(seed + 1234567, n, 8)]
    flags = [[1 if which[x] == i else 0 for x in range(n)] for i in range(len(instrs))]
    Ep = [O.synthetic_fr(seed + 9000 * (m + 1), n) for m in range(n_mem)]
    out = [P.g_plain(instrs[which[x]], [Ep[m][x] for m in instrs[which[x]].mems]) for x in range(n)]
    if nparties == 1:
        E, outs = [Ep], [out]
    else:
        Es = [O.rep3_share_vec(Ep[m], O.harness_prf_key(seed + 9000 * (m + 1), 101), O.harness_prf_key(seed + 9000 * (m + 1), 102)) for m in range(n_mem)]
        E = [[Es[m][p] for m in range(n_mem)] for p in range(3)]
        outs = O.rep3_share_vec(out, O.harness_prf_key(seed + 555, 101), O.harness_prf_key(seed + 555, 102))
    r_eq = tr.challenge_vector(cfg["log_n"])
    proof, rs, _fin = P.prove(instrs, r_eq, flags, E, outs, tr)
    vr_eq = vt.challenge_vector(cfg["log_n"])
    v = P.verify(instrs, vr_eq, proof, n_mem, vt)
    ok = v == rs
    if ok:  # openings == direct evaluations at the reversed challenge list
        eq = O.eq_evals(list(reversed(rs)))
        direct = [sum(a * b for a, b in zip(eq, poly)) % R for poly in Ep + flags + [out]]
        ok = direct == proof["openings"]
    blob = O.ser_u64(len(proof["round_polys"])) + b"".join(O.ser_vec_fr(c) for c in proof["round_polys"]) + O.ser_vec_fr(proof["openings"])
    return blob, ok


def run(cfg):
    nparties = 1 if cfg["mode"] == "plain" else 3
    n = 1 << cfg["log_n"]
    batch = 2 * cfg["n_pairs"]
    seed = cfg["seed"]
    tr = O.Transcript(b"cozk-lookups")
    vt = O.Transcript(b"cozk-lookups")
    head, ok_primary = (run_primary(cfg, tr, vt) if cfg.get("primary") else (b"", True))
    cols = [flag_column(seed, q, n, cfg["density_pct"]) for q in range(cfg["n_pairs"])]
    flag_indices = [[i for i, f in enumerate(c) if f] for c in cols]
    per_circuit = [fingerprints_of(seed, b, n, nparties) for b in range(batch)]
    fps = [[per_circuit[b][p] for b in range(batch)] for p in range(nparties)]
    toggles, sparse = S.toggled_construct(flag_indices, fps)
    proof, r = S.toggled_prove(toggles, sparse, tr)
    v = S.toggled_verify(proof, vt)
    ok = v is not None and ok_primary
    if ok:
        plain = [O.synthetic_fr(seed + 7000 * (b + 1), n) for b in range(batch)]
        ok = (v[0], v[1]) == S.toggled_leaf_mles(flag_indices, plain, v[2]) and v[2] == r
    blob = head + serialize(proof)
    return {"proof_bytes": blob, "digest": hashlib.sha256(blob).hexdigest(), "verified": ok, "proof": proof, "r": r}
