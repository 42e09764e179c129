"""GPU tests of the C++ worker round loops (`cozk_worker_*`) run against a host-supplied transport:
Python plays the coordinator inside the cozk_star_net callbacks and checks every message with the oracle.
Covers prove_arbitrary_worker (co-jolt/src/subprotocols/sumcheck.rs:168-246), the dense grand-product worker
(grand_product.rs:111-130) and co-noir-spartan's two sumcheck workers (co-spartan/src/worker.rs:593-639)."""
import importlib

import pytest

import pyref as O

pytestmark = pytest.mark.gpu


def _fr_vec(b):
    n = int.from_bytes(b[:8], "little")
    return [int.from_bytes(b[8 + 32 * i:40 + 32 * i], "little") for i in range(n)]


def _ser(*xs):
    return b"".join((x % O.R).to_bytes(32, "little") for x in xs)


@pytest.mark.parametrize("degree,m", [(2, 2), (3, 3)])
def test_prove_arbitrary_worker_callbacks(cozk, ctx, degree, m):
    W = importlib.import_module("co-zkvms_amd.workers")
    rng = O.SplitMix64(90 + degree)
    n = 128
    ref = [[rng.field() for _ in range(n)] for _ in range(m)]
    polys = [cozk.Rep3DensePolynomial.new(ctx, c) for c in ref]
    claim = 0
    for i in range(n):
        t = 1
        for j in range(m):
            t = t * ref[j][i] % O.R
        claim = (claim + t) % O.R
    state = {"ref": ref, "claim": claim, "msgs": 0, "pending": None}

    def on_send(b):
        cf = _fr_vec(b)
        evs = O.prod_round_evals(state["ref"], degree)
        pts = [evs[0], (state["claim"] - evs[0]) % O.R] + evs[1:]
        assert cf == O.unipoly_from_evals(pts)  # the worker's message is the oracle's round polynomial
        assert (O.unipoly_eval(cf, 0) + O.unipoly_eval(cf, 1)) % O.R == state["claim"]  # g(0)+g(1) = claim
        r = rng.field()
        state["claim"] = O.unipoly_eval(cf, r)
        state["ref"] = [O.dense_bind(c, r, O.HIGH_TO_LOW) for c in state["ref"]]
        state["pending"] = _ser(r, state["claim"])
        state["msgs"] += 1

    star = W.CallbackStar(on_send, lambda: state["pending"])
    nv = n.bit_length() - 1
    r, finals = W.prove_arbitrary(ctx, polys, degree, claim, nv, star)
    assert state["msgs"] == nv and len(r) == nv
    assert finals == [c[0] for c in state["ref"]]
    prod = 1
    for f in finals:
        prod = prod * f % O.R
    assert prod == state["claim"]  # final sumcheck check


def test_grand_product_worker_callbacks_matches_oracle_proof(cozk, ctx):
    """whole GKR prove through the callback transport == the oracle's proof (plain prover)"""
    W = importlib.import_module("co-zkvms_amd.workers")
    rng = O.SplitMix64(123)
    batch, per = 2, 16
    leaves = [rng.field() for _ in range(batch * per)]
    layers = O.gp_construct([leaves], batch, None)
    ref_proof, ref_r = O.gp_prove(layers, O.Transcript(b"t"))
    tr = O.Transcript(b"t")
    st = {"phase": "outputs", "claim": None, "rounds_left": 0, "layer": 0, "pending": None, "rs": [], "r": None, "polys": []}

    def on_send(b):
        if st["phase"] == "outputs":
            outs = _fr_vec(b)
            assert outs == ref_proof["outputs"]
            tr.append_scalars(outs)
            nvo = (len(outs) - 1).bit_length()
            r = tr.challenge_vector(nvo)
            claim = sum(e * v for e, v in zip(O.eq_evals(r), outs)) % O.R
            st["r"], st["claim"] = r, claim
            st["pending"] = len(r).to_bytes(8, "little") + _ser(*r) + _ser(claim)
            st["phase"] = "num_rounds"
        elif st["phase"] == "num_rounds":
            st["rounds_left"] = int.from_bytes(b[:8], "little")
            st["rs"] = []
            st["phase"] = "round" if st["rounds_left"] else "finals"
        elif st["phase"] == "round":
            cf = _fr_vec(b)
            comp = O.unipoly_compress(cf)
            assert comp == ref_proof["layers"][st["layer"]]["round_polys"][len(st["rs"])]
            tr.append_scalars(comp)
            rj = tr.challenge_scalar()
            st["rs"].append(rj)
            st["claim"] = O.unipoly_eval(cf, rj)
            st["pending"] = _ser(rj, st["claim"])
            st["rounds_left"] -= 1
            if st["rounds_left"] == 0:
                st["phase"] = "finals"
        elif st["phase"] == "finals":
            la, _lb, ra, _rb = (int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(4))
            lp = ref_proof["layers"][st["layer"]]
            assert (la, ra) == (lp["left"], lp["right"])
            tr.append_scalar(la)
            tr.append_scalar(ra)
            r_layer = tr.challenge_scalar()
            st["r"] = list(reversed(st["rs"])) + [r_layer]
            st["claim"] = (la + r_layer * (ra - la)) % O.R
            st["pending"] = _ser(r_layer)
            st["layer"] += 1
            st["phase"] = "num_rounds"

    star = W.CallbackStar(on_send, lambda: st["pending"])
    layer = cozk.Rep3DenseInterleavedPolynomial.new(ctx, leaves)
    r = W.prove_grand_product(ctx, layer, batch, star)
    assert r == ref_r and st["layer"] == len(ref_proof["layers"])
    assert layer.coeffs() == leaves  # the driver works on a clone


def test_spartan_sumcheck_workers_callbacks(cozk, ctx):
    W = importlib.import_module("co-zkvms_amd.workers")
    rng = O.SplitMix64(31337)
    nv = 6
    n = 1 << nv
    ref = [[rng.field() for _ in range(n)] for _ in range(4)]  # za, zb, zc, eq (plain prover)
    P = [cozk.Rep3DensePolynomial.new(ctx, c) for c in ref]
    st = {"ref": ref, "pending": None, "rounds": 0, "claim": None}

    def on_send(b):
        if len(b) == 8 + 4 * 32:
            ev = _fr_vec(b)
            assert ev == O.spartan_first_round_evals(*st["ref"])
            if st["claim"] is not None:
                assert (ev[0] + ev[1]) % O.R == st["claim"]
            r = rng.field()
            # degree-3 interpolation through X = 0..3 gives the next claim
            st["claim"] = O.unipoly_eval(O.unipoly_from_evals(ev), r)
            st["ref"] = [O.dense_bind(c, r, O.LOW_TO_HIGH) for c in st["ref"]]
            st["pending"] = _ser(r)
            st["rounds"] += 1
        else:
            fin = [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(4)]
            assert fin == [c[0] for c in st["ref"]]

    star = W.CallbackStar(on_send, lambda: st["pending"])
    pt, fin = W.spartan_first_sumcheck(ctx, *P, star)
    assert st["rounds"] == nv and len(pt) == nv
    a, b, c, e = fin
    assert (a * b - c) * e % O.R == st["claim"]
    # second sumcheck
    ref2 = [[rng.field() for _ in range(n)] for _ in range(4)]  # z, A, B, C
    coef = [rng.field() for _ in range(3)]
    P2 = [cozk.Rep3DensePolynomial.new(ctx, c) for c in ref2]
    st2 = {"ref": ref2, "pending": None, "rounds": 0}

    def on_send2(b):
        if len(b) == 8 + 6 * 32:
            vals = [int.from_bytes(b[8 + 32 * i:40 + 32 * i], "little") for i in range(6)]
            assert vals[0::2] == O.spartan_second_round_evals(*st2["ref"], coef) and vals[1::2] == [0, 0, 0]
            r = rng.field()
            st2["ref"] = [O.dense_bind(c, r, O.LOW_TO_HIGH) for c in st2["ref"]]
            st2["pending"] = _ser(r)
            st2["rounds"] += 1

    star2 = W.CallbackStar(on_send2, lambda: st2["pending"])
    pt2, fin2 = W.spartan_second_sumcheck(ctx, *P2, coef, star2)
    assert st2["rounds"] == nv and fin2 == [c[0] for c in st2["ref"]]
