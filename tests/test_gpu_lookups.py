"""GPU parity of the toggled / sparse batched grand product (co-jolt/src/subprotocols/sparse_grand_product.rs,
co-jolt/src/poly/sparse_interleaved_poly.rs) through the C ABI:
  * the toggle-layer kernels (cozk_toggle_*) against the oracle's LITERAL sparse restatement (oracle/pysparse.py), every
    round of the layer's sumcheck, per party: layer_output, compute_cubic (all four cases: coalesced or not x E1 bound or
    not), bind incl. the coalesce step, final_claims; ragged batches (6, 10 circuits), densities 0 % and 100 %;
  * the sparse layers, kept dense on the device, against the oracle's sparse layers round by round;
  * the whole pipeline (csrc/host/lookups_harness.hpp): proofs bit-identical to oracle/pylookups.py for the plain prover
    and the 3-party Rep3 run at small sizes; at larger sizes the built-in verifier (every round, every layer reduction,
    final claims == direct evaluations of the flag and fingerprint polynomials) and Rep3 proof == plain proof."""
import copy
import hashlib
import importlib

import pytest

import pylookups
import pyref as O
import pysparse as S

pytestmark = pytest.mark.gpu
R = O.R


def _instance(batch, n, density, seed, nparties):
    rng = O.SplitMix64(seed)
    cols = [[1 if rng.next() % 100 < density else 0 for _ in range(n)] for _ in range(batch // 2)]
    vals = [[rng.field() for _ in range(n)] for _ in range(batch)]
    if nparties == 1:
        fps = [vals]
    else:
        sh = [[O.rep3_share(v, rng) for v in row] for row in vals]
        fps = [[[s[p] for s in row] for row in sh] for p in range(3)]
    return cols, vals, fps


@pytest.mark.parametrize("batch,n,density,nparties", [(4, 16, 30, 3), (6, 8, 60, 3), (6, 8, 60, 1), (2, 64, 15, 3), (10, 4, 50, 1),
                                                       (2, 2, 100, 3), (4, 8, 0, 1), (8, 256, 10, 3)])
def test_toggle_layer_all_rounds_match_the_sparse_oracle(cozk, ctx, batch, n, density, nparties):
    LK = importlib.import_module("co-zkvms_amd.lookups")
    cols, vals, fps = _instance(batch, n, density, 100 + batch + n, nparties)
    flag_indices = [[i for i, f in enumerate(c) if f] for c in cols]
    rng = O.SplitMix64(5)
    L = 1 << (batch - 1).bit_length()
    nv = (L * n).bit_length() - 1
    w = [rng.field() for _ in range(nv)]
    rs = [rng.field() for _ in range(nv)]
    for p in range(nparties):
        ref = S.ToggleLayer(flag_indices, fps[p], p, nparties)
        dev = LK.ToggleLayer(ctx, cols, fps[p])
        # layer_output: the sparse layer's dense form (missing entries = this party's share of one)
        out = dev.layer_output(party=p)
        assert out.coeffs() == ref.layer_output().coalesce()
        out.free()
        eq_ref = O.SplitEq(w)
        eq_dev = cozk.SplitEqPolynomial(ctx, w)
        claim = rng.field()
        for j in range(nv):
            ev = ref.compute_cubic_evals(eq_ref, claim)
            got = dev.round(eq_dev, rs[j - 1] if j else None, party=p)
            assert got == [ev[0], ev[2], ev[3]], (p, j)
            ref.bind(rs[j])
            eq_ref.bind(rs[j])
        dev.bind(rs[-1])
        fl, fp = dev.final_claims()
        rfl, rfp = ref.final_claims()
        assert fp == rfp and fl == (rfl if nparties == 1 else ref.coalesced_flags[0])
        dev.free()


@pytest.mark.parametrize("batch,n,nparties", [(6, 16, 3), (4, 32, 1)])
def test_sparse_layers_kept_dense_match_the_sparse_oracle(cozk, ctx, batch, n, nparties):
    """a Rep3SparseInterleavedPolynomial on the device is a cozk_layer with the ones stored: cubic round messages and
    binds equal the oracle's sparse bind / compute_cubic (20 neighbour cases) in every round"""
    cols, vals, fps = _instance(batch, n, 35, 77, nparties)
    flag_indices = [[i for i, f in enumerate(c) if f] for c in cols]
    toggles, sparse = S.toggled_construct(flag_indices, fps)
    rng = O.SplitMix64(8)
    for li in (0, 1):
        for p in range(nparties):
            ref = copy.deepcopy(sparse[li][p])
            dev = cozk.Rep3DenseInterleavedPolynomial.new(ctx, ref.coalesce())
            nv = max(1, (ref.dense_len // 2 - 1).bit_length())
            w = [rng.field() for _ in range(nv)]
            eq_ref, eq_dev = O.SplitEq(w), cozk.SplitEqPolynomial(ctx, w)
            claim = rng.field()
            for _ in range(nv):
                assert dev.compute_cubic(eq_dev, claim) == O.unipoly_from_evals(ref.compute_cubic_evals(eq_ref, claim))
                r = rng.field()
                ref.bind(r)
                dev.bind(r)
                eq_ref.bind(r)
                eq_dev.bind(r)
                assert dev.coeffs() == ref.coalesce()
            dev.free()


@pytest.mark.parametrize("mode", ["plain", "rep3"])
@pytest.mark.parametrize("cfg", [dict(log_n=3, n_pairs=2, density_pct=40, seed=5), dict(log_n=4, n_pairs=3, density_pct=25, seed=9),
                                 dict(log_n=1, n_pairs=1, density_pct=100, seed=2), dict(log_n=3, n_pairs=5, density_pct=0, seed=3)])
def test_small_proofs_bit_identical_to_the_sparse_oracle(cozk, mode, cfg):
    LK = importlib.import_module("co-zkvms_amd.lookups")
    h = LK.LookupsHarness(mode=mode, **cfg)
    res = h.prove(verify=True)
    assert res.verified == 1, h.last_error()
    ref = pylookups.run(dict(cfg, mode=mode))
    assert ref["verified"]
    got = h.proof_bytes(res)
    assert hashlib.sha256(got).hexdigest() == bytes(res.proof_digest).hex()
    assert got == ref["proof_bytes"]
    h.close()


def test_2p14_54_memories_verifies_and_rep3_equals_plain(cozk):
    """Jolt's shape at a reduced trace length: 54 memories (108 circuits, padded to 128 by the split-eq point), 10 % flag
    density, 2^14 cycles; the prove leaves the resident leaves untouched (a second prove gives the same proof)"""
    LK = importlib.import_module("co-zkvms_amd.lookups")
    digs = {}
    for mode in ("plain", "rep3"):
        h = LK.LookupsHarness(mode=mode, log_n=14, n_pairs=54, density_pct=10, seed=2026)
        r = h.prove(verify=True)
        assert r.verified == 1, h.last_error()
        assert bytes(h.prove(verify=False).proof_digest) == bytes(r.proof_digest)
        if mode == "rep3":
            assert r.bytes_ring > 0
        digs[mode] = bytes(r.proof_digest)
        h.close()
    assert digs["plain"] == digs["rep3"]


@pytest.mark.parametrize("mode", ["plain", "rep3"])
@pytest.mark.parametrize("cfg", [dict(log_n=1, n_pairs=3, density_pct=50, seed=4), dict(log_n=3, n_pairs=19, density_pct=30, seed=4),
                                 dict(log_n=4, n_pairs=8, density_pct=30, seed=11), dict(log_n=5, n_pairs=21, density_pct=20, seed=12),
                                 dict(log_n=7, n_pairs=54, density_pct=30, seed=6), dict(log_n=6, n_pairs=54, density_pct=10, seed=9, mix="sha2")])
def test_primary_sumcheck_proofs_bit_identical_to_the_oracle(cozk, mode, cfg):
    """Lasso's primary sumcheck + the toggled grand product in one proof: the 27 RV32I instructions' collation
    forms (13 of them), degree-8 round polynomials (8 evaluations per round), up to three mul_vec / reshare levels per round in
    the Rep3 run; bytes equal the oracle (oracle/pyprimary.py restates primary_sumcheck_prover_message and every
    combine_lookups_rep3_batched)"""
    LK = importlib.import_module("co-zkvms_amd.lookups")
    h = LK.LookupsHarness(mode=mode, primary=True, **cfg)
    res = h.prove(verify=True)
    assert res.verified == 1, h.last_error()
    ref = pylookups.run(dict(cfg, mode=mode, primary=1, mix=1 if cfg.get("mix") == "sha2" else 0))
    assert ref["verified"]
    assert h.proof_bytes(res) == ref["proof_bytes"]
    if mode == "rep3" and cfg["log_n"] >= 3:
        assert res.bytes_ring > 0
    h.close()


def test_primary_sumcheck_2p14_verifies_and_rep3_equals_plain(cozk):
    LK = importlib.import_module("co-zkvms_amd.lookups")
    digs = {}
    for mode in ("plain", "rep3"):
        h = LK.LookupsHarness(mode=mode, log_n=14, n_pairs=54, density_pct=10, seed=2026, primary=True)
        r = h.prove(verify=True)
        assert r.verified == 1, h.last_error()
        assert bytes(h.prove(verify=False).proof_digest) == bytes(r.proof_digest)
        digs[mode] = bytes(r.proof_digest)
        h.close()
    assert digs["plain"] == digs["rep3"]


@pytest.mark.parametrize("mode", ["plain", "rep3"])
@pytest.mark.parametrize("logw", [1, 2])
def test_primary_sumcheck_worker_subnets_give_the_same_proof(cozk, mode, logw):
    """the worker sub-net split of the primary sumcheck (jolt/vm/instruction_lookups/worker.rs:194-372, coordinator.rs:97-150):
    every worker proves the first log_n - log_workers rounds on its high-variable chunk (the coordinator adds the sub-nets'
    evaluations), worker 0 finishes on the 2^log_workers gathered finals; the proof is byte-identical to the unsplit one
    and to the oracle's"""
    LK = importlib.import_module("co-zkvms_amd.lookups")
    cfg = dict(log_n=6, n_pairs=19, density_pct=30, seed=21)
    h = LK.LookupsHarness(mode=mode, primary=True, log_workers=logw, **cfg)
    res = h.prove(verify=True)
    assert res.verified == 1, h.last_error()
    ref = pylookups.run(dict(cfg, mode=mode, primary=1))
    assert h.proof_bytes(res) == ref["proof_bytes"]
    h.close()


def test_primary_sumcheck_split_2p14_equals_unsplit(cozk):
    LK = importlib.import_module("co-zkvms_amd.lookups")
    digs = []
    for logw in (0, 3):
        h = LK.LookupsHarness(mode="rep3", log_n=14, n_pairs=54, density_pct=10, seed=2026, primary=True, log_workers=logw)
        r = h.prove(verify=True)
        assert r.verified == 1, h.last_error()
        digs.append(bytes(r.proof_digest))
        h.close()
    assert digs[0] == digs[1]


def test_hip_pipelines_reproduce_the_committed_golden_digests(cozk):
    """tests/golden/round2_pipelines.json: the HIP harnesses give the committed digests of the lookups, spartan (+ public lookup
    round) and outer-sumcheck pipelines"""
    import json
    import os
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "round2_pipelines.json")))
    LK = importlib.import_module("co-zkvms_amd.lookups")
    OU = importlib.import_module("co-zkvms_amd.outer")
    for row in G["lookups"]:
        cfg = dict(row["cfg"])
        h = LK.LookupsHarness(mode=cfg.pop("mode"), primary=bool(cfg.pop("primary", 0)), **cfg)
        r = h.prove(verify=True)
        assert r.verified == 1 and bytes(r.proof_digest).hex() == row["digest"] and r.proof_len == row["proof_len"], row["cfg"]
        h.close()
    for row in G["spartan"]:
        cfg = dict(row["cfg"])
        for mode in ("plain", "rep3"):
            h = cozk.SpartanHarness(mode=mode, log_n=cfg["log_n"], seed=cfg["seed"], lookup_round=bool(cfg.get("lookup_round", 0)))
            r = h.prove(verify=True)
            assert r.verified == 1 and bytes(r.proof_digest).hex() == row["digest"], (row["cfg"], mode)
            h.close()
    for row in G["outer"]:
        cfg = dict(row["cfg"])
        h = OU.OuterHarness(mode=cfg["mode"], log_steps=cfg["log_steps"], seed=cfg["seed"])
        r = h.prove(verify=True)
        assert r.verified == 1 and bytes(r.proof_digest).hex() == row["digest"], row["cfg"]
        h.close()
