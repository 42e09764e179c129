"""Worker sub-net ("split") form: every polynomial chunked over the high variables across 2^k workers
(split_poly, co-jolt/src/poly/dense_mlpoly.rs:275-301), grand-product circuits divided among the workers, the
last k sumcheck rounds and PST folds finished on the gathered finals.  In-process here (one thread per worker
on the single GPU).  Checks: the built-in verifier accepts (GKR, leaf evaluation, reduction sumcheck, PST13
with the full trapdoor), and the commitment + GKR sections of the proof are BYTE-IDENTICAL to the single-worker
proof of the same witness (partial commitments against SRS slices add up; round polynomials are unique)."""
import pytest

pytestmark = pytest.mark.gpu


def _u64(b, o):
    return int.from_bytes(b[o:o + 8], "little"), o + 8


def _skip_vec(b, o):
    n, o = _u64(b, o)
    return o + 32 * n


def _gp_end(b):
    """offset of the end of the commitments + GKR sections of a serialized ProofBundle"""
    n, o = _u64(b, 0)
    o += n * 72
    n, o = _u64(b, o)
    o += n * 72
    o = _skip_vec(b, o)  # outputs
    nl, o = _u64(b, o)
    for _ in range(nl):
        nr, o = _u64(b, o)
        for _ in range(nr):
            o = _skip_vec(b, o)
        o += 64
    return o


@pytest.mark.parametrize("mode", ["plain", "rep3"])
@pytest.mark.parametrize("log_workers", [1, 2])
def test_split_verifies_and_matches_single_worker_prefix(cozk, mode, log_workers):
    cfg = dict(log_n=10, n_fr=5, n_u16=2, n_u32=1, n_flags=2, n_small=0, gp_batch=4, gp_log_leaves=11, seed=77)
    h1 = cozk.Harness(mode=mode, **cfg)
    r1 = h1.prove(verify=True)
    assert r1.verified == 1, h1.last_error()
    p1 = h1.proof_bytes(r1)
    h1.close()
    hs = cozk.Harness(mode=mode, log_workers=log_workers, **cfg)
    rs = hs.prove(verify=True)
    assert rs.verified == 1, hs.last_error()
    ps = hs.proof_bytes(rs)
    # deterministic across steps
    assert bytes(hs.prove(verify=False).proof_digest) == bytes(rs.proof_digest)
    hs.close()
    e1, es = _gp_end(p1), _gp_end(ps)
    assert e1 == es and p1[:e1] == ps[:es]
    assert len(p1) == len(ps)  # same proof shape (same number of rounds / points)


def test_split_eight_workers_medium(cozk):
    h = cozk.Harness(mode="plain", log_workers=3, log_n=14, n_fr=8, n_u16=4, n_u32=2, n_flags=2, n_small=0, gp_batch=8, seed=3)
    r = h.prove(verify=True)
    assert r.verified == 1, h.last_error()
    h.close()
