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
           "--master-port", "29533", os.path.join(ROOT, "tools", "dist_prove.py"), "--log-n", "12", "--steps", "1", "--polys", "16"]
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


@pytest.mark.parametrize("mode,logw,hub", [("plain", 1, "shm"), ("plain", 2, "shm"), ("plain", 1, "gloo")])
def test_worker_subnets_one_process_each(cozk, mode, logw, hub):
    """the worker sub-net (split) form with one (party, worker) participant per process: 2 / 4 plain workers sharing the one GPU
    (3 parties x 2 workers would be 6 ranks + this process, over the box's 6-process guard: that combination is
    covered in-process by test_gpu_split.py); star messages through the shared-memory hub or gloo.  Every rank assembles and verifies the same proof, bit-identical to the
    in-process run of the same configuration."""
    nproc = (3 if mode == "rep3" else 1) << logw
    env = dict(os.environ, COZK_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(29540 + logw + (4 if hub == "gloo" else 0)), os.path.join(ROOT, "tools", "dist_prove.py"),
           "--mode", mode, "--log-workers", str(logw), "--hub", hub, "--log-n", "12", "--steps", "1", "--polys", "16"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rep = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["verified"] == 1
    h = cozk.Harness(mode=mode, log_n=12, n_fr=8, n_u16=4, n_u32=2, n_flags=2, n_small=0, gp_batch=8, seed=2026, log_workers=logw)
    r = h.prove(verify=True)
    assert r.verified == 1
    assert bytes(r.proof_digest).hex()[:16] == rep["proof_sha256"]
    h.close()
