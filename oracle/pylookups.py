"""ORACLE (test infrastructure): the synthetic instruction-lookups pipeline of csrc/host/lookups_harness.hpp in pure
Python over the SPARSE oracle (oracle/pysparse.py): same synthetic flags and fingerprints, same transcript, same proof
serialization -- for small sizes the engine's proof must be bit-identical (plain prover and 3-party Rep3 run)."""
import hashlib

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


def run(cfg):
    nparties = 1 if cfg["mode"] == "plain" else 3
    n = 1 << cfg["log_n"]
    batch = 2 * cfg["n_pairs"]
    seed = cfg["seed"]
    cols = [flag_column(seed, q, n, cfg["density_pct"]) for q in range(cfg["n_pairs"])]
    flag_indices = [[i for i, f in enumerate(c) if f] for c in cols]
    per_circuit = [fingerprints_of(seed, b, n, nparties) for b in range(batch)]
    fps = [[per_circuit[b][p] for b in range(batch)] for p in range(nparties)]
    toggles, sparse = S.toggled_construct(flag_indices, fps)
    tr = O.Transcript(b"cozk-lookups")
    proof, r = S.toggled_prove(toggles, sparse, tr)
    v = S.toggled_verify(proof, O.Transcript(b"cozk-lookups"))
    ok = v is not None
    if ok:
        plain = [O.synthetic_fr(seed + 7000 * (b + 1), n) for b in range(batch)]
        ok = (v[0], v[1]) == S.toggled_leaf_mles(flag_indices, plain, v[2]) and v[2] == r
    blob = serialize(proof)
    return {"proof_bytes": blob, "digest": hashlib.sha256(blob).hexdigest(), "verified": ok, "proof": proof, "r": r}
