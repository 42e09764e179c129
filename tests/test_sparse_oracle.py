"""CPU tests (no GPU) of the toggled / sparse grand-product oracle (oracle/pysparse.py): the literal
restatement of the reference's SPARSE algorithms (sparse_interleaved_poly.rs, sparse_grand_product.rs) must
  * verify and open to the right products, with final claims equal to direct evaluations of the leaf polynomials;
  * give the same proof for the plain prover and the 3-party Rep3 run (shares cancel in the coordinator's sums);
  * agree, party by party and round by round, with the DENSE formulation the engine uses (missing entries stored as
    shares of one; flags and fingerprints as dense vectors) -- the property the device kernels rely on."""
import pytest

import pylookups
import pyref as O
import pysparse as S

R = O.R


def _instance(batch, n, density, seed, nparties):
    rng = O.SplitMix64(seed)
    flags = [sorted(i for i in range(n) if rng.next() % 100 < density) for _ in range(batch // 2)]
    vals = [[rng.field() for _ in range(n)] for _ in range(batch)]
    if nparties == 1:
        fps = [vals]
    else:
        sh = [[O.rep3_share(v, rng) for v in row] for row in vals]
        fps = [[[s[p] for s in row] for row in sh] for p in range(3)]
    return flags, vals, fps


@pytest.mark.parametrize("batch,n,density", [(2, 8, 50), (4, 16, 30), (6, 8, 60), (2, 4, 100), (4, 8, 0), (6, 32, 10), (10, 16, 25)])
def test_toggled_grand_product_verifies_and_plain_equals_rep3(batch, n, density):
    proofs = []
    for nparties in (1, 3):
        flags, vals, fps = _instance(batch, n, density, 7, nparties)
        toggles, sparse = S.toggled_construct(flags, fps)
        proof, r = S.toggled_prove(toggles, sparse, O.Transcript(b"t"))
        v = S.toggled_verify(proof, O.Transcript(b"t"))
        assert v is not None and v[2] == r
        assert (v[0], v[1]) == S.toggled_leaf_mles(flags, vals, r)
        for b in range(batch):
            s, p = set(flags[b // 2]), 1
            for i in s:
                p = p * vals[b][i] % R
            assert proof["outputs"][b] == p
        proofs.append(proof)
    assert proofs[0] == proofs[1]


def _toggle_dense_evals(flags_dense, fps_dense, half_n, eq, party, nparties):
    """the engine's formulation of the toggle layer's round sums (csrc/toggle_layer.inc k_toggle_cubic): dense flags and
    fingerprints, pair j of circuit b uses the flag pair of circuit b >> 1 (half_n = pairs per circuit; 0: coalesced)"""
    npairs = len(fps_dense) // 2
    s = [0, 0, 0]
    nested = eq.E1_len != 1
    E1h = eq.E1_len // 2
    limit = E1h * eq.E2_len if nested else eq.E2_len // 2
    for j in range(min(npairs, limit)):
        if nested:
            x2, x1 = divmod(j, E1h)
            e = [v * eq.E2[x2] % R for v in S._eq3(eq.E1[2 * x1], eq.E1[2 * x1 + 1])]
        else:
            e = S._eq3(eq.E2[2 * j], eq.E2[2 * j + 1])
        fj = j
        if half_n:
            b, i = divmod(j, half_n)
            fj = (b >> 1) * half_n + i
        f = S._eq3(flags_dense[2 * fj], flags_dense[2 * fj + 1])
        p0, p1 = O.sh_into_additive(fps_dense[2 * j]), O.sh_into_additive(fps_dense[2 * j + 1])
        p = S._eq3(p0, p1)
        for k in range(3):
            s[k] = (s[k] + e[k] * (p[k] * f[k] + ((1 - f[k]) if party == 0 else 0))) % R
    return s


@pytest.mark.parametrize("batch,n,density,nparties", [(4, 16, 30, 3), (6, 8, 60, 3), (6, 8, 60, 1), (2, 32, 15, 3), (10, 4, 50, 1)])
def test_sparse_algorithms_equal_the_dense_formulation_party_by_party(batch, n, density, nparties):
    flags, vals, fps = _instance(batch, n, density, 11, nparties)
    toggles, sparse = S.toggled_construct(flags, fps)
    rng = O.SplitMix64(99)
    # ---- a sparse layer: compute_cubic == dense compute_cubic of coalesce(), bind == dense bind, every round
    for layer_set in (sparse[0], sparse[1] if len(sparse) > 1 else sparse[0]):
        nv = max(1, (layer_set[0].dense_len // 2 - 1).bit_length())
        w = [rng.field() for _ in range(nv)]
        for p in range(nparties):
            import copy
            lay = copy.deepcopy(layer_set[p])
            dense = lay.coalesce()
            eq_s, eq_d = O.SplitEq(w), O.SplitEq(w)
            claim = rng.field()
            for _ in range(nv):
                assert lay.compute_cubic_evals(eq_s, claim) == O.interleaved_compute_cubic_evals(dense, eq_d, claim)
                r = rng.field()
                lay.bind(r)
                dense = O.interleaved_bind(dense, r)
                eq_s.bind(r)
                eq_d.bind(r)
                assert lay.coalesce() == dense
    # ---- the toggle layer: the four cases == the dense sums over (flags, fingerprints)
    L = 1 << (batch - 1).bit_length()
    nv = (L * n).bit_length() - 1
    w = [rng.field() for _ in range(nv)]
    for p in range(nparties):
        import copy
        t = copy.deepcopy(toggles[p])
        fl_dense = [1 if i in set(flags[q]) else 0 for q in range(batch // 2) for i in range(n)]
        fp_dense = [v for row in fps[p] for v in row]
        half_n, coalesced = n // 2, False
        eq_s, eq_d = O.SplitEq(w), O.SplitEq(w)
        claim = rng.field()
        for _ in range(nv):
            ev = t.compute_cubic_evals(eq_s, claim)
            ds = _toggle_dense_evals(fl_dense, fp_dense, 0 if coalesced else half_n, eq_d, p, nparties)
            assert [ev[0], ev[2], ev[3]] == ds
            r = rng.field()
            t.bind(r)
            eq_s.bind(r)
            eq_d.bind(r)
            fl_dense = O.public_bind(fl_dense, r, O.LOW_TO_HIGH)
            fp_dense = O.dense_bind(fp_dense, r, O.LOW_TO_HIGH)
            if not coalesced:
                half_n //= 2
                if half_n == 0:  # one entry per circuit left: coalesce (sparse_grand_product.rs:104-134)
                    zero = S.zero_share(nparties)
                    fl_dense = [fl_dense[c >> 1] if c < batch else 1 for c in range(L)]
                    fp_dense = [fp_dense[c] if c < batch else zero for c in range(L)]
                    coalesced = True
        fl, fp = t.final_claims()
        assert fp == fp_dense[0]
        assert (fl if nparties == 1 else O.rep3_open([O.rep3_promote_from_trivial(fl_dense[0], q) for q in range(3)])) == fl_dense[0]


def test_pipeline_oracle_small():
    for mode in ("plain", "rep3"):
        out = pylookups.run(dict(mode=mode, log_n=3, n_pairs=2, density_pct=40, seed=5))
        assert out["verified"]
    a = pylookups.run(dict(mode="plain", log_n=3, n_pairs=3, density_pct=30, seed=9))
    b = pylookups.run(dict(mode="rep3", log_n=3, n_pairs=3, density_pct=30, seed=9))
    assert a["verified"] and a["digest"] == b["digest"]


def test_primary_sumcheck_oracle_plain_equals_rep3_and_verifies():
    """oracle/pyprimary.py: the three collation forms, 3-party lock-step with mul_vec reshares == the plain
    prover; the verifier's final check and the openings hold; pipeline bytes of the two modes are identical"""
    for log_n, n_pairs in ((1, 3), (3, 19), (4, 8)):
        a = pylookups.run(dict(mode="plain", log_n=log_n, n_pairs=n_pairs, density_pct=30, seed=4, primary=1))
        b = pylookups.run(dict(mode="rep3", log_n=log_n, n_pairs=n_pairs, density_pct=30, seed=4, primary=1))
        assert a["verified"] and b["verified"]
        assert a["proof_bytes"] == b["proof_bytes"]


def test_every_rv32i_collation_rep3_schedule_opens_to_the_plain_formula():
    """all 27 instruction rows of the harness table (13 collation forms): the Rep3 schedule of combine_lookups_rep3_batched,
    run by three lock-step parties, opens to the plain combine_lookups on the same values -- also for C = 2, 3 shapes"""
    import pyprimary as P
    rng = O.SplitMix64(77)
    table = pylookups.instr_table(64)
    extra = [P.Instr(P.SLT, range(5)), P.Instr(P.SLT, range(7)), P.Instr(P.NOT_SLT, range(7)), P.Instr(P.SIGNED_REM, range(10)),
             P.Instr(P.SIGNED_REM, range(14)), P.Instr(P.LTE, range(2)), P.Instr(P.LTE, range(4)), P.Instr(P.DIV0, range(2)),
             P.Instr(P.UNSIGNED_REM, range(2)), P.Instr(P.UNSIGNED_REM, range(5)), P.Instr(P.LTU, range(1)), P.Instr(P.NOT_LTU, range(3)),
             P.Instr(P.PRODUCT, range(1)), P.Instr(P.NOT_PRODUCT, range(2))]
    assert len(table) == 27 and {i.form for i in table} == set(range(13))
    assert P.sumcheck_degree(table) == 8
    for instr in table + extra:
        nm, items = len(instr.mems), 5
        plain = [[rng.field() for _ in range(items)] for _ in range(nm)]
        shares = [[O.rep3_share(v, rng) for v in row] for row in plain]
        vals = [[[shares[m][j][p] for j in range(items)] for m in range(nm)] for p in range(3)]
        out3 = P.combine_lookups_batched(instr, vals)
        out1 = P.combine_lookups_batched(instr, [plain])
        for j in range(items):
            want = P.g_plain(instr, [plain[m][j] for m in range(nm)])
            assert sum(O.sh_into_additive(out3[p][j]) for p in range(3)) % R == want, (instr.form, nm)
            assert out1[0][j] % R == want, (instr.form, nm)


def test_primary_sumcheck_oracle_covers_every_instruction_at_2p7():
    a = pylookups.run(dict(mode="plain", log_n=7, n_pairs=54, density_pct=30, seed=6, primary=1))
    b = pylookups.run(dict(mode="rep3", log_n=7, n_pairs=54, density_pct=30, seed=6, primary=1))
    assert a["verified"] and b["verified"] and a["proof_bytes"] == b["proof_bytes"]


def test_spartan_outer_oracle_verifies_and_plain_equals_rep3():
    """oracle/pyspartan_outer.py: the sparse SharedOrPublic walk with the Gruen split-eq verifies, its claims
    are the multilinear extensions of the clear Az, Bz, Cz, and the plain and the 3-party run give the same bytes"""
    import pyspartan_outer as SO
    for ls in (0, 1, 3, 4):
        a = SO.run(dict(mode="plain", log_steps=ls, seed=3))
        b = SO.run(dict(mode="rep3", log_steps=ls, seed=3))
        assert a["verified"] and b["verified"] and a["proof_bytes"] == b["proof_bytes"]


def test_round2_oracles_against_the_committed_golden_fixtures():
    """tests/golden/round2_pipelines.json (made by tests/golden/make_golden.py): proof digests of the lookups / spartan (+ lookup
    round) / outer-sumcheck oracles at small sizes, and one known-answer row per RV32I instruction's collation"""
    import hashlib
    import json
    import os
    import pyprimary as P
    import pyspartan
    import pyspartan_outer as SO
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "round2_pipelines.json")))
    for row in G["lookups"]:
        r = pylookups.run(dict(row["cfg"]))
        assert r["verified"] and r["digest"] == row["digest"] and len(r["proof_bytes"]) == row["proof_len"], row["cfg"]
    for row in G["spartan"]:
        r = pyspartan.run(dict(row["cfg"]))
        assert r["verified"] and r["digest"] == row["digest"] and len(r["proof_bytes"]) == row["proof_len"], row["cfg"]
    for row in G["outer"]:
        r = SO.run(dict(row["cfg"]))
        assert r["verified"] and hashlib.sha256(r["proof_bytes"]).hexdigest() == row["digest"], row["cfg"]
    table = pylookups.instr_table(64)
    assert len(G["collations"]) == len(table) == 27
    for row, instr in zip(G["collations"], table):
        assert (row["form"], row["n_mems"], row["bits"]) == (instr.form, len(instr.mems), instr.bits)
        assert P.g_plain(instr, [int(v, 16) for v in row["E"]]) == int(row["g"], 16), row["instruction"]
