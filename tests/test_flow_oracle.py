"""CPU tier for the chained-flow oracle (oracle/pyflow.py) and the round-3 golden fixtures (tests/golden/round3_pipelines.json, made by
tests/golden/make_golden.py from the Python oracles; parity unpinned by reference outputs -- the reference holds none):
the 3-party Rep3 run and the plain prover give the same proof, section by section (shares and masks cancel in everything the
coordinator sees), and the oracles still reproduce the committed digests."""
import hashlib
import json
import os

import pyflow as F
import pyjolt_r1cs as J
import pyref as O
import pyspartan_outer as S

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "round3_pipelines.json")))


def test_flow_plain_equals_rep3_section_by_section():
    cfg = dict(log_n=3, log_m=2, log_b=2, log_mem=3, n_mem=5, n_subtables=2, seed=21)
    a, b = F.run(dict(cfg, mode="plain")), F.run(dict(cfg, mode="rep3"))
    for k in a["sections"]:
        assert a["sections"][k] == b["sections"][k], k
    assert a["n_openings"] == 10
    # 78 R1CS inputs + bytecode t_read + 4 timestamps + read_cts / E / final_cts per memory + lookup outputs + t_final + v_init / v_final / t_final
    assert a["n_commitments"] == 78 + 1 + 4 + 3 * 5 + 1 + 1 + 3


def test_golden_flow_digests():
    for row in G["flow"]:
        r = F.run(row["cfg"])
        assert r["digest"] == row["digest"] and len(r["proof_bytes"]) == row["proof_len"]
        assert {k: hashlib.sha256(v).hexdigest() for k, v in r["sections"].items()} == row["sections"]


def test_golden_spartan_digests():
    for row in G["spartan_full"]:
        r = S.run_full(row["cfg"])
        assert r["verified"] and r["digest"] == row["digest"] and len(r["proof_bytes"]) == row["proof_len"]


def test_golden_constraint_table():
    uniform, cross, padded = J.build_system()
    g = G["jolt_r1cs"]
    assert (len(uniform), len(cross), padded, J.NUM_INPUTS) == (g["n_uniform"], g["n_cross"], g["padded"], g["n_inputs"])
    assert hashlib.sha256(repr((uniform, cross)).encode()).hexdigest() == g["table_sha256"]
    cols = J.synthetic_columns(5, 16)
    assert hashlib.sha256(b"".join(O.ser_vec_fr(c) for c in cols)).hexdigest() == g["trace_seed5_n16_sha256"]
