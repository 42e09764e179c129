"""Multi-process GPU test (3 ranks <= the 6-process guard): a 3-party Rep3 proof with ONE PARTY PER PROCESS
(BASELINE config 3's mapping), replicated coordinator over a torch.distributed all-gather hub and the ring
reshare through torch.distributed P2P.  On the 1-GPU box the three ranks share the GPU and the ring is staged
through host memory (gloo); with >= 3 GPUs the same code rides RCCL.  The proof must verify on every rank
and be bit-identical to the in-process 3-party run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_three_processes_one_party_each(cozk):
    env = dict(os.environ, COZK_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tools", "rep3_dist.py"), "--log-n", "12", "--steps", "1", "--polys", "16"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rep = json.loads(line)
    assert rep["verified"] == 1 and rep["ring_bytes_per_party"] > 0
    h = cozk.Harness(mode="rep3", log_n=12, n_fr=8, n_u16=4, n_u32=2, n_flags=2, n_small=0, gp_batch=8, seed=2026)
    r = h.prove(verify=True)
    assert r.verified == 1
    assert bytes(r.proof_digest).hex()[:16] == rep["proof_sha256"]
    h.close()
