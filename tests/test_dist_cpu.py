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
