"""gloo world-size-3 CPU test of the variable-length byte all-gather that carries the star messages of the
distributed (replicated-coordinator) form: fast path (<= 2040 B in one collective) and long-message path."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = textwrap.dedent("""
    import importlib, os, sys
    sys.path.insert(0, %r)
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    P = importlib.import_module("co-zkvms_amd.party_dist")
    g = dist.new_group(backend="gloo")
    for sizes in ([0, 5, 300], [2040, 2040, 2040], [10, 5000, 0], [70000, 1, 2]):
        mine = bytes([(rank * 7 + i) %% 251 for i in range(sizes[rank])])
        got = P.all_gather_bytes(g, world, mine)
        assert [len(x) for x in got] == sizes, (sizes, [len(x) for x in got])
        for r in range(world):
            assert got[r] == bytes([(r * 7 + i) %% 251 for i in range(sizes[r])])
    dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def test_all_gather_bytes_gloo_world3(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="29588")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()
        assert b"ok" in out
