"""GPU tests of the native Rep3 ring (csrc/ring.hip: RCCL ncclSend / ncclRecv on the context's stream behind
cozk_ring_init / cozk_reshare / cozk_rep3_mul_vec -- the exchange of rep3::arithmetic::mul_vec and reshare_additive,
mpc-core/src/protocols/rep3/arithmetic.rs:144-164).
  * on ONE GPU the ring degenerates to a single rank whose next and previous party are itself: that still runs the whole
    RCCL path for real (ncclGetUniqueId, ncclCommInitRank, a grouped ncclSend + ncclRecv on the context's stream,
    ncclCommDestroy), and what comes back must be what was sent;
  * with >= 3 GPUs (skipped below): three processes, one party per GPU, the whole distributed proof over the native
    ring, bit-identical to the in-process three-party run."""
import json
import os
import subprocess
import sys

import pytest
import torch  # noqa: F401 - a torch host maps its own librccl first; libcozk then reuses that copy (one RCCL per process)

import pyref as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_ring_single_rank_self_loop(cozk):
    c = cozk.Context(0)
    try:
        c.ring_init(cozk.Context.ring_unique_id(), 0, 1)
        assert c.ring_info()[:2] == (0, 1)
        n = 1000
        vals = O.synthetic_fr(31, n)
        got = c.reshare(cozk.Vec.from_ints(c, vals))
        assert got.to_ints() == vals  # next == prev == self
        # mul_vec, whole: c.a = local product + mask, c.b = previous (= own) c.a
        rng = O.SplitMix64(5)
        X = [(rng.field(), rng.field()) for _ in range(n)]
        Y = [(rng.field(), rng.field()) for _ in range(n)]
        ks, kp = O.harness_prf_key(1, 0), O.harness_prf_key(1, 2)
        V = cozk.Vec.from_ints
        ca, cb = c.rep3_mul_vec(V(c, [x[0] for x in X]), V(c, [x[1] for x in X]), V(c, [y[0] for y in Y]), V(c, [y[1] for y in Y]), ks, kp, counter=17)
        ms, mp = O.prf_fr_vec(ks, 17, n), O.prf_fr_vec(kp, 17, n)
        exp = [(O.rep3_local_mul(x, y) + a - b) % O.R for x, y, a, b in zip(X, Y, ms, mp)]
        assert ca.to_ints() == exp and cb.to_ints() == exp
        assert c.ring_info()[2] == 2 * n * 32
        # a large exchange (64 MiB) and an empty one
        big = cozk.Vec.random(c, 1 << 21, seed=9)
        back = c.reshare(big)
        assert back.binop(cozk.OP_SUB, big).to_numpy().any() == False  # noqa: E712 - numpy bool
        assert len(c.reshare(cozk.Vec.alloc(c, 0))) == 0
        c.ring_destroy()
        with pytest.raises(cozk.CozkError):
            c.reshare(big)  # no ring any more
    finally:
        c.close()


def test_native_ring_three_parties_one_gpu_each(cozk):
    import torch
    if torch.cuda.device_count() < 3:
        pytest.skip("needs 3 GPUs: RCCL refuses two ranks of one communicator on the same device")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", "29561", os.path.join(ROOT, "tools", "dist_prove.py"), "--ring", "native", "--log-n", "14", "--steps", "1", "--polys", "16"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rep = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["verified"] == 1 and rep["ring"] == "native" and rep["ring_bytes_per_party"] > 0
    h = cozk.Harness(mode="rep3", log_n=14, n_fr=8, n_u16=4, n_u32=2, n_flags=2, n_small=0, gp_batch=8, seed=2026)
    r = h.prove(verify=True)
    assert r.verified == 1 and bytes(r.proof_digest).hex()[:16] == rep["proof_sha256"]
    h.close()
