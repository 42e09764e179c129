"""GPU parity of co-noir-spartan's PUBLIC lookup round (co-noir-spartan/co-spartan/src/worker.rs:400-575,
694-724; spartan/src/logup.rs:31-80; co-spartan/src/sumcheck.rs:434-500) against oracle/pylogup.py:
  * hash_tuple, LogLookupProof::prove's field work (phi, h with one inversion per element), boost_degree;
  * the whole round the way fourth_round composes it: two lookups (row / column) -> 13 products over 13 flattened polynomials
    -> IPForMLSumcheck rounds (4 evaluations each, degree 3) with Python playing the coordinator: every round message equals the
    oracle's, the verifier's checks hold (claimed sum = the constant-term product's sum), and the final values
    (obtain_distrbuted_sumcheck_prover_state) are the polynomials at the point."""
import importlib

import pytest

import pylogup as G
import pyref as O

pytestmark = pytest.mark.gpu
R = O.R


def test_hash_tuple_logup_h_and_boost_degree(cozk, ctx):
    LG = importlib.import_module("co-zkvms_amd.logup")
    rng = O.SplitMix64(77)
    eq = [rng.field() for _ in range(64)]
    idx = [rng.next() % 64 for _ in range(37)]
    v = rng.field()
    got = LG.hash_tuple(ctx, idx, cozk.Vec.from_ints(ctx, eq), v, 64).to_ints()
    assert got == G.hash_tuple(idx, eq, v)
    gidx = idx + [0xFFFFFFFF, 3]
    assert LG.gather(ctx, gidx, cozk.Vec.from_ints(ctx, eq), 64).to_ints() == [eq[i] for i in idx] + [0, eq[3]] + [0] * (64 - len(gidx))
    vals = [rng.field() for _ in range(100)] + [0, 1, R - 1]
    m = [rng.next() % 5 for _ in range(len(vals))]
    x = rng.field()
    phi, h = LG.logup_h(ctx, cozk.Vec.from_ints(ctx, vals), cozk.Vec.from_ints(ctx, m), x)
    assert phi.to_ints() == [(x + t) % R for t in vals]
    assert h.to_ints() == [mv * pow((x + t) % R, -1, R) % R for mv, t in zip(m, vals)]
    _phi1, h1 = LG.logup_h(ctx, cozk.Vec.from_ints(ctx, vals), None, x)
    assert h1.to_ints() == [pow((x + t) % R, -1, R) for t in vals]
    g = [rng.field() for _ in range(8)]
    assert LG.boost_degree(ctx, cozk.Vec.from_ints(ctx, g), 5).to_ints() == G.boost_degree(g, 5)
    assert LG.boost_degree(ctx, cozk.Vec.from_ints(ctx, g), 3).to_ints() == g


@pytest.mark.parametrize("qv,tv", [(4, 2), (6, 4), (9, 6)])
def test_public_lookup_round_matches_oracle(cozk, ctx, qv, tv):
    """two logup lookups (queries into a table, with multiplicities) + the eq . eq . val product, as fourth_round builds its
    ListOfProductsOfPolynomials, proven with the device's IPForMLSumcheck rounds"""
    LG = importlib.import_module("co-zkvms_amd.logup")
    rng = O.SplitMix64(1000 + qv)
    V = cozk.Vec.from_ints
    polys_ref, products = [], []
    dev_polys = []
    # the first product of fourth_round: eq_rx_chunk * eq_ry_chunk * val_m_chunk (worker.rs:461-467)
    trio = [[rng.field() for _ in range(1 << qv)] for _ in range(3)]
    for p in trio:
        polys_ref.append(p)
    products.append((1, [0, 1, 2]))
    claimed = sum(a * b % R * c for a, b, c in zip(*trio)) % R
    for _lookup in range(2):
        table = [rng.field() for _ in range(1 << tv)]
        qidx = [rng.next() % (1 << tv) for _ in range(1 << qv)]
        query = [table[i] for i in qidx]
        mult = [0] * (1 << tv)
        for i in qidx:
            mult[i] += 1
        x = rng.field()
        z = [rng.field() for _ in range(qv)]
        lam = rng.field()
        # device: LogLookupProof::prove
        phi0_d, h0_d = LG.logup_h(ctx, V(ctx, table), V(ctx, mult), x)
        phi1_d, h1_d = LG.logup_h(ctx, V(ctx, query), None, x)
        h0_b, phi0_b = LG.boost_degree(ctx, h0_d, qv), LG.boost_degree(ctx, phi0_d, qv)
        h, phi = G.loglookup_prove(query, table, mult, x)
        assert [h0_b.to_ints(), h1_d.to_ints()] == h and [phi0_b.to_ints(), phi1_d.to_ints()] == phi
        m_b = G.boost_degree(mult, qv)  # boost_degree(freq, q.num_vars) (worker.rs:528)
        G.append_sumcheck_polys(polys_ref, products, h, phi, m_b, qv - tv, z, lam)
    dev_polys = [V(ctx, p) for p in polys_ref]
    pl = LG.ProdList(ctx, dev_polys, products)
    assert pl.degree == 3
    # rounds: Python is the coordinator
    tr = O.Transcript(b"logup")
    ref_msgs, ref_point, ref_finals = G.distributed_sumcheck(polys_ref, products, O.Transcript(b"logup"))
    msgs, point = [], []
    r = None
    for j in range(qv):
        ev = pl.round(r)
        assert ev == ref_msgs[j], j
        tr.append_scalars(ev)
        r = tr.challenge_scalar()
        msgs.append(ev)
        point.append(r)
    finals = pl.final(r)
    assert point == ref_point and finals == ref_finals
    # the logup identities make each lookup's six products sum to zero; the trio contributes its inner product
    assert G.verify_sumcheck(msgs, point, finals, products, claimed)
    # finals are the polynomials at the point (ark's little-endian order: variable j <-> index bit j)
    for p, f in zip(polys_ref[:4], finals[:4]):
        assert O.pst_evaluate_le(p, point) == f
    pl.free()
