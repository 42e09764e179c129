"""GPU parity of the ONE chained co-jolt worker flow (csrc/host/flow_harness.hpp; JoltRep3Prover::prove,
co-jolt/src/jolt/vm/jolt/worker.rs:175-266): commit-all -> bytecode -> instruction lookups (primary sumcheck, toggled + dense grand
products) -> read-write memory + output check -> Spartan -> ONE reduce_and_prove, under one transcript and one opening accumulator.
  * whole proofs byte-identical to oracle/pyflow.py (plain prover and 3-party Rep3) at small sizes -- section by section, so a
    mismatch names the phase;
  * the harness's own plain verifier accepts (every sumcheck / GKR layer, fingerprints of the OPENED values == the GKR claims,
    reduction sumcheck, the one batched PST13 opening with the trapdoor);
  * at 2^10 / 2^14 with Jolt's 54 memories: verified, Rep3 == plain, deterministic across steps."""
import hashlib
import importlib

import pytest

import pyflow as F

pytestmark = pytest.mark.gpu

SECTIONS = ("commit", "bytecode", "lookups", "rw", "spartan", "open")


def _split(blob, ref):
    out, off = {}, 0
    for k in SECTIONS:
        n = len(ref["sections"][k])
        out[k] = blob[off:off + n]
        off += n
    return out


@pytest.mark.parametrize("mode", ["plain", "rep3"])
@pytest.mark.parametrize("cfg", [dict(log_n=2, log_m=1, log_b=1, log_mem=2, n_mem=3, n_subtables=2, seed=3),
                                 dict(log_n=3, log_m=3, log_b=2, log_mem=3, n_mem=6, n_subtables=3, seed=5),
                                 dict(log_n=5, log_m=3, log_b=4, log_mem=4, n_mem=9, n_subtables=4, seed=7),
                                 dict(log_n=4, log_m=3, log_b=3, log_mem=3, n_mem=6, n_subtables=3, seed=11, small_witness=1)])
def test_flow_proof_bit_identical_to_the_oracle(cozk, mode, cfg):
    FL = importlib.import_module("co-zkvms_amd.flow")
    h = FL.FlowHarness(mode=mode, **cfg)
    res = h.prove(verify=True)
    got = h.proof_bytes(res)
    ref = F.run(dict(cfg, mode=mode))
    sec = _split(got, ref)
    for k in SECTIONS:
        assert sec[k] == ref["sections"][k], "section %s differs" % k
    assert got == ref["proof_bytes"]
    assert res.verified == 1, h.last_error()
    assert hashlib.sha256(got).hexdigest() == bytes(res.proof_digest).hex()
    assert h.num_polys() == ref["n_commitments"]
    h.close()


@pytest.mark.parametrize("log_n,n_mem,n_sub,log_m", [(10, 54, 26, 8), (14, 54, 26, 12)])
def test_flow_jolt_shape_verifies_and_rep3_equals_plain(cozk, log_n, n_mem, n_sub, log_m):
    FL = importlib.import_module("co-zkvms_amd.flow")
    digs = {}
    for mode in ("plain", "rep3"):
        h = FL.FlowHarness(mode=mode, log_n=log_n, log_m=log_m, log_b=log_n - 2, log_mem=log_n - 1, n_mem=n_mem, n_subtables=n_sub, seed=2026)
        r = h.prove(verify=True)
        assert r.verified == 1, h.last_error()
        assert bytes(h.prove(verify=False).proof_digest) == bytes(r.proof_digest)
        digs[mode] = bytes(r.proof_digest)
        h.close()
    assert digs["plain"] == digs["rep3"]


def test_golden_round3_digests(cozk):
    """tests/golden/round3_pipelines.json: the HIP harnesses give the committed digests of the whole Spartan worker and of the flow"""
    import json
    import os
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "round3_pipelines.json")))
    FL = importlib.import_module("co-zkvms_amd.flow")
    OU = importlib.import_module("co-zkvms_amd.outer")
    for row in G["flow"]:
        cfg = dict(row["cfg"])
        h = FL.FlowHarness(**cfg)
        res = h.prove(verify=True)
        assert res.verified == 1, h.last_error()
        assert bytes(res.proof_digest).hex() == row["digest"] and int(res.proof_len) == row["proof_len"]
        h.close()
    for row in G["spartan_full"]:
        cfg = dict(row["cfg"])
        h = OU.OuterHarness(mode=cfg["mode"], log_steps=cfg["log_steps"], seed=cfg["seed"], system=cfg.get("system", "jolt"), full=True)
        res = h.prove(verify=True)
        assert res.verified == 1, h.last_error()
        assert bytes(res.proof_digest).hex() == row["digest"]
        h.close()
