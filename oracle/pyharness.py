"""
ORACLE (test infrastructure, NOT product code) -- pure-Python restatement of the whole synthetic
prove pipeline that co-zkvms_amd/csrc/harness.hip runs on the GPU: same seeded witness, same Rep3
sharing, same message schedule, same SHA-256 transcript, same proof serialisation.  Small sizes
only (Python loops).  The SHA-256 of the serialized proof is the parity handle: the HIP path must
reproduce it bit for bit, for the plain prover and for the 3-party Rep3 run (which must also equal
each other, because shares and masks cancel in the coordinator's sums -- SURVEY.md 0).

Reference call chain restated: co-jolt/src/jolt/vm/jolt/worker.rs:175-266 (commit -> grand products ->
openings -> reduce_and_prove) restricted to the dense path of SURVEY.md 8d config 2/3.
"""
import hashlib

import pyref as O

R = O.R


def _ser_fr(x):
    return (x % R).to_bytes(32, "little")


def _ser_u64(x):
    return int(x).to_bytes(8, "little")


def _ser_vec(v):
    return _ser_u64(len(v)) + b"".join(_ser_fr(x) for x in v)


def _ser_g1(pt):
    return O.ser_g1(pt)


def _shares_of(seed, n, mode):
    """per-party coefficient lists of the secret stream(seed) (harness.hip make_share_vectors)"""
    v = O.synthetic_fr(seed, n)
    if mode == "plain":
        return [v]
    return O.rep3_share_vec(v, O.harness_prf_key(seed, 101), O.harness_prf_key(seed, 102))


def _pst_setup(seed, nv):
    t = [O.synthetic_fr(seed ^ 0x7A7A7A7A, nv)[i] for i in range(nv)]
    powers = []
    for i in range(nv):
        ev = [1]
        for tj in t[i:]:
            ev = [e * (1 - tj) % R for e in ev] + [e * tj % R for e in ev]
        powers.append([O.g1_mul(O.G1_GEN, e) for e in ev])
    return {"nv": nv, "t": t, "g": O.G1_GEN, "powers_of_g": powers}


def _commit(ck, scalars):
    return O.msm_naive(ck["powers_of_g"][0][:len(scalars)], scalars)


def _a_of(coeffs):
    return [c[0] if isinstance(c, tuple) else c for c in coeffs]


def run(cfg):
    """cfg: dict(mode, log_n, n_fr, n_u16, n_u32, n_flags, n_small, gp_batch, gp_log_leaves, seed).
    Returns dict(proof_bytes, digest, verified)."""
    mode = cfg["mode"]
    np_ = 1 if mode == "plain" else 3
    nv = cfg["log_n"]
    N = 1 << nv
    seed = cfg["seed"]
    ck = _pst_setup(seed, nv)
    # ---- witness
    polys = []  # per poly: {"public": bool, "parts": per-party coeff lists, "commit": per-party scalar lists}
    j = 0
    for _ in range(cfg["n_fr"]):
        parts = _shares_of(seed + 1000 * (j + 1), N, mode)
        polys.append({"public": False, "parts": parts, "commit": [_a_of(p) for p in parts]})
        j += 1
    for count, bits in ((cfg["n_u16"], 16), (cfg["n_u32"], 32), (cfg["n_flags"], 1)):
        for _ in range(count):
            vals = O.synthetic_small(seed + 1000 * (j + 1), N, bits)
            polys.append({"public": True, "parts": [vals] * np_, "commit": [vals] * np_})
            j += 1
    small = []
    for k in range(cfg["n_small"]):
        parts = _shares_of(seed + 300000 + 1000 * k, N >> 4, mode)
        small.append({"public": False, "parts": parts, "commit": [_a_of(p) for p in parts]})
    nleaves = cfg["gp_batch"] << cfg["gp_log_leaves"]
    fingerprints = bool(cfg.get("leaf_fingerprints", False))
    leaves = None if fingerprints else _shares_of(seed + 500000, nleaves, mode)
    tr = O.Transcript(b"cozk-harness")
    # ---- commit: shared = sum of parties' share commitments; public = P0's commitment
    def combine(plist):
        out = []
        for p in plist:
            if p["public"]:
                out.append(_commit(ck, p["commit"][0]))
            else:
                acc = None
                for q in range(np_):
                    acc = O.g1_add(acc, _commit(ck, p["commit"][q]))
                out.append(acc)
        return out
    commitments = combine(polys)
    small_commitments = combine(small)
    for c in commitments + small_commitments:
        tr.append_point(c)
    if fingerprints:
        # compute_leaves (K11; harness.hip worker_main): per circuit the read leaves gamma u16 + gamma^2 u32 + gamma^3 flag
        # + gamma^k shared - tau over the N cycles, then the write leaves (+ gamma^(k+1))
        assert cfg["gp_log_leaves"] == nv + 1 and cfg["n_fr"] >= 1
        gamma, tau = tr.challenge_scalar(), tr.challenge_scalar()
        first_u16, first_u32 = cfg["n_fr"], cfg["n_fr"] + cfg["n_u16"]
        first_flag = first_u32 + cfg["n_u32"]
        leaves = [[] for _ in range(np_)]
        for circ in range(cfg["gp_batch"]):
            cols, cc, g = [], [], gamma
            for first, count in ((first_u16, cfg["n_u16"]), (first_u32, cfg["n_u32"]), (first_flag, cfg["n_flags"])):
                if count > 0:
                    cols.append(polys[first + circ % count]["parts"][0])
                    cc.append(g)
                    g = g * gamma % R
            g_sh, g_w = g, g * gamma % R
            for q in range(np_):
                shared = polys[circ % cfg["n_fr"]]["parts"][q]
                party = None if mode == "plain" else q
                leaves[q] += O.fingerprint_leaves(cols, cc, [shared], [g_sh], (-tau) % R, party)
                leaves[q] += O.fingerprint_leaves(cols, cc, [shared], [g_sh], (g_w - tau) % R, party)
    # ---- grand product
    mask_ctr = 0

    class _MaskRng:
        """zero-sum masks exactly as the engine derives them: mask_p[j] = PRF(seed_p, ctr+j) - PRF(seed_{p-1}, ctr+j)"""
    layers = [[list(l) for l in leaves]]
    per = 1 << cfg["gp_log_leaves"]
    num_layers = cfg["gp_log_leaves"]
    for _ in range(num_layers - 1):
        prev = layers[-1]
        if np_ == 1:
            layers.append([O.interleaved_layer_output_local(prev[0])])
        else:
            lr = [O.interleaved_uninterleave(prev[p]) for p in range(3)]
            n_out = len(lr[0][0])
            prf = [O.prf_fr_vec(O.harness_prf_key(seed, p), mask_ctr, n_out) for p in range(3)]
            masks = [[(prf[p][jj] - prf[(p + 2) % 3][jj]) % R for jj in range(n_out)] for p in range(3)]
            layers.append(O.rep3_mul_vec([x[0] for x in lr], [x[1] for x in lr], masks))
            mask_ctr += n_out
    gp_proof, r_gp = O.gp_prove(layers, tr)
    # ---- openings
    K = len(polys)
    half = (K + 1) // 2
    groups = [(polys[:half], r_gp[len(r_gp) - nv:])]
    if half < K:
        groups.append((polys[half:], r_gp[:nv]))
    if small:
        groups.append((small, r_gp[len(r_gp) - (nv - 4):]))
    opening_claims = []
    openings_pp = [[] for _ in range(np_)]
    for plist, point in groups:
        eq = O.eq_evals(point)
        claims_pp = []
        for q in range(np_):
            cl = []
            for p in plist:
                if p["public"]:
                    v = sum(a * b for a, b in zip(p["parts"][q], eq)) % R
                    cl.append(O.additive_promote_from_trivial(v, q))
                else:
                    cl.append(O.dense_evaluate_at_chi(p["parts"][q], eq))
            claims_pp.append(cl)
        claims = O.combine_additive(claims_pp)
        opening_claims.append(claims)
        rho = tr.challenge_scalar()
        pw = [1]
        for _ in range(1, len(plist)):
            pw.append(pw[-1] * rho % R)
        batched_claim = sum(a * b for a, b in zip(pw, claims)) % R
        for q in range(np_):
            n = max(len(p["parts"][q]) for p in plist)
            if mode == "plain":
                acc = [0] * n
                for c, p in zip(pw, plist):
                    for i, v in enumerate(p["parts"][q]):
                        acc[i] = (acc[i] + v * c) % R
                claim_share = batched_claim
            else:
                acc = [(0, 0)] * n
                acc = list(acc)
                for c, p in zip(pw, plist):
                    for i, v in enumerate(p["parts"][q]):
                        if p["public"]:
                            # add_public: P0 -> a, P1 -> b, P2 -> nothing
                            a, b = acc[i]
                            if q == 0:
                                a = (a + v * c) % R
                            elif q == 1:
                                b = (b + v * c) % R
                            acc[i] = (a, b)
                        else:
                            acc[i] = O.rep3_add(acc[i], O.rep3_mul_public(v, c))
                claim_share = O.rep3_promote_from_trivial(batched_claim, q)
            openings_pp[q].append({"poly": acc, "eq": list(eq), "point": list(point), "claim": claim_share})
    r_red, red_claims, red = O.opening_reduce(openings_pp, tr)
    gamma = tr.challenge_scalar()
    gpw = [1]
    for _ in range(1, len(groups)):
        gpw.append(gpw[-1] * gamma % R)
    proofs = None
    point_rev = list(reversed(r_red))
    for q in range(np_):
        joint = [0] * N
        for c, op in zip(gpw, openings_pp[q]):
            for i, v in enumerate(op["poly"]):
                a = v[0] if isinstance(v, tuple) else v
                joint[i] = (joint[i] + a * c) % R
        pf, _ = O.pst_open(ck, joint, point_rev)
        proofs = pf if proofs is None else [O.g1_add(x, y) for x, y in zip(proofs, pf)]
    # ---- serialise exactly like harness.hip ProofBundle::serialize
    out = b""
    out += _ser_u64(len(commitments))
    for p, c in zip(polys, commitments):
        out += _ser_u64(nv) + _ser_g1(c)
    out += _ser_u64(len(small_commitments))
    for c in small_commitments:
        out += _ser_u64(nv - 4) + _ser_g1(c)
    out += _ser_vec(gp_proof["outputs"])
    out += _ser_u64(len(gp_proof["layers"]))
    for lp in gp_proof["layers"]:
        out += _ser_u64(len(lp["round_polys"]))
        for comp in lp["round_polys"]:
            out += _ser_vec(comp)
        out += _ser_fr(lp["left"]) + _ser_fr(lp["right"])
    out += _ser_u64(len(opening_claims))
    for cl in opening_claims:
        out += _ser_vec(cl)
    out += _ser_u64(len(red["round_polys"]))
    for comp in red["round_polys"]:
        out += _ser_vec(comp)
    out += _ser_vec(red_claims)
    out += _ser_u64(len(proofs)) + b"".join(_ser_g1(p) for p in proofs)
    # ---- verification of the PST opening with the trapdoor (joint commitment / claim)
    batched_c = []
    for (plist, point), claims in zip(groups, opening_claims):
        pass
    return {"proof_bytes": out, "digest": hashlib.sha256(out).hexdigest(), "r_gp": r_gp, "gp_proof": gp_proof,
            "commitments": commitments, "proofs": proofs}
