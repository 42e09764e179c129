"""CPU: the co-noir-spartan public lookup round over 2^k public workers restated with Python integers (oracle/pyspartan.py
lookup_round(log_pub_workers=k): co-spartan/src/setup.rs split_ipk / split_ck, coordinator.rs:425-475,748-811) gives the one-worker
proof byte for byte -- the coordinator's merges are linear.  The GPU harness is checked against the same bytes in
tests/test_gpu_spartan.py."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import pyspartan  # noqa: E402
import pylogup  # noqa: E402
import pyref as O  # noqa: E402


@pytest.mark.parametrize("log_n,seed,ks", [(2, 3, (1, 2, 3)), (3, 5, (1, 2))])
def test_split_lookup_round_equals_one_worker(log_n, seed, ks):
    base = pyspartan.run(dict(log_n=log_n, seed=seed, lookup_round=1))
    assert base["verified"]
    for k in ks:
        got = pyspartan.run(dict(log_n=log_n, seed=seed, lookup_round=1, log_pub_workers=k))
        assert got["verified"]
        assert got["proof_bytes"] == base["proof_bytes"], f"k = {k}"


def test_key_slices_add_up_to_the_index_key():
    ck = pyspartan.index_ck(11, 4)
    v = O.synthetic_fr(77, 16)
    want = O.pst_commit(ck, v)
    for k in (1, 2, 3):
        assert pyspartan.split_commit(ck, v, k) == want
    point = O.synthetic_fr(78, 4)
    proofs, val = O.pst_open(ck, v, point)
    for k in (1, 2, 3):
        p2, v2 = pyspartan.split_open(ck, v, point, k)
        assert p2 == proofs and v2 == val


def test_split_sumcheck_messages_equal_unsplit():
    polys = [O.synthetic_fr(100 + i, 16) for i in range(4)]
    products = [(3, [0, 1, 2]), (O.R - 1, [0, 3]), (5, [1])]
    t1 = O.Transcript(b"t")
    a = pylogup.distributed_sumcheck(polys, products, t1)
    for k in (1, 2, 3):
        t2 = O.Transcript(b"t")
        assert pylogup.distributed_sumcheck_split(polys, products, t2, k) == a
