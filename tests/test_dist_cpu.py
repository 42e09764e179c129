"""world_size-2 gloo test of bench.py's multi-process plumbing (barrier, max-over-ranks timing,
digest gather, per-rank segment seeds).  No GPU, no field arithmetic."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import importlib, os, sys
    sys.path.insert(0, %r)
    d = importlib.import_module("co-zkvms_amd.dist")
    g = d.Group(backend="gloo", device=None)
    g.barrier()
    m = g.max_over_ranks(10.0 + g.rank)
    s = g.sum_over_ranks(1 << 20)
    digs = g.all_gather_bytes(bytes([g.rank]) * 32)
    assert m == 10.0 + g.world - 1, m
    assert s == g.world * (1 << 20)
    assert digs == [bytes([r]) * 32 for r in range(g.world)]
    assert d.shard_seed(5, g.rank) == 5 + 7919 * g.rank
    g.barrier()
    g.close()
    print("rank", g.rank, "ok")
""") % ROOT


def test_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()
        assert b"ok" in out


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` without a launcher starts its own two rank processes (fresh children, before anything
    touches a GPU), they rendezvous over gloo on 127.0.0.1 and the parent relays exactly rank 0's JSON line"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only"], env=env, capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == [0, 1] and out["max_over_ranks"] == 2.0


def test_bench_self_launch_reports_a_failed_rank():
    """a rank that fails (here: no GPU for the proving) makes the parent exit non-zero instead of hanging or printing a line"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    try:
        import torch
        if torch.cuda.is_available():
            import pytest
            pytest.skip("a GPU is visible: the ranks would really prove")
    except ImportError:
        pass
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, timeout=300)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]


def test_bench_self_launch_one_late_rank_fails_fast():
    """rank 1 dies before the rendezvous while rank 0 waits in it -- the parent polls every child, ends rank 0 and
    returns rank 1's exit code at once instead of waiting for the rendezvous to time out"""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["COZK_BENCH_TEST_FAIL_RANK"] = "1"
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only"], env=env, capture_output=True, timeout=300)
    assert p.returncode == 3, (p.returncode, p.stderr.decode())
    assert time.time() - t0 < 120
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]


def test_bench_self_launch_eight_ranks_cpu_rehearsal():
    """the 8-rank shape of the driver's scaling run, rehearsed where it is allowed to run -- on the CPU (a GPU box admits
    at most 6 processes on its card, so 8 GPU ranks are the driver's to start on an 8-GPU node): `bench.py --gpus 8` starts its 8
    ranks itself, they rendezvous over gloo on 127.0.0.1, run the barrier / max-over-ranks / digest-gather plumbing of the timed
    region and 64 rounds of the worker sub-nets' star exchange through libcozk's shared-memory hub, every rank checking every
    other rank's bytes"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plumbing-only"], env=env, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == list(range(8)) and out["max_over_ranks"] == 8.0 and out["shm_hub_64_rounds_ok"] is True


def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--plumbing-only"], env=env, capture_output=True, timeout=120)
    assert p.returncode != 0 and b"WORLD_SIZE" in p.stderr
